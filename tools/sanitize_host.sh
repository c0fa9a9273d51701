#!/bin/bash
# Host loaders (image decoder, glTF / OBJ / scene-file parsers, Renderer edit flows) under AddressSanitizer +
# UBSan on valid, truncated and byte-flipped inputs.  CPU only.  Usage: tools/sanitize_host.sh [workdir]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WORK=${1:-/tmp/rt_fuzz}
python3 "$ROOT/tools/make_fuzz_inputs.py" "$WORK"
cd "$ROOT/raytracing_amd/csrc/host"
g++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -I. \
    -o "$WORK/driver" "$ROOT/tools/sanitize_host_main.cpp" renderer.cpp accel_build.cpp meshes.cpp scene_file.cpp \
    image_decode.cpp image_jpeg.cpp image_encode.cpp gltf_file.cpp -L../.. -lrt_hip -Wl,-rpath,"$ROOT/raytracing_amd"
cd "$ROOT"
"$WORK/driver" "$WORK"/*.png "$WORK"/*.jpg "$WORK"/*.gltf "$WORK"/*.glb "$WORK"/*.obj "$WORK"/*.json assets/scenes/*.json 2>&1 | grep -v "rta warn"
# scene validation + flattening of librt_hip.so on scenes with randomly overwritten indices / counts / offsets
cd "$ROOT/raytracing_amd/csrc/host"
g++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -I. \
    -o "$WORK/flatten" "$ROOT/tools/sanitize_flatten_main.cpp" renderer.cpp accel_build.cpp meshes.cpp scene_file.cpp \
    image_decode.cpp image_jpeg.cpp image_encode.cpp gltf_file.cpp ../rt_flatten.cpp -L../.. -lrt_hip -Wl,-rpath,"$ROOT/raytracing_amd"
cd "$ROOT"
ASAN_OPTIONS=detect_leaks=0 "$WORK/flatten" assets/scenes/cornell.json assets/scenes/multi.json 2>&1 | grep -v "rta warn"
