#!/usr/bin/env python3
"""Inputs for tools/sanitize_host.sh: valid, truncated and byte-flipped PNG / glTF / GLB / OBJ / scene files."""
import json, os, pathlib, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_image_decode import encode_png, _rand, CASES, CHANNELS
from tests import test_gltf_import as G

out = pathlib.Path(sys.argv[1] if len(sys.argv) > 1 else "/tmp/rt_fuzz")
out.mkdir(parents=True, exist_ok=True)
rng = np.random.default_rng(0)
n = 0
for color, depth, inter in CASES:
    smp = _rand(rng, 17, 33, CHANNELS[color], depth)
    pal = rng.integers(0, 256, (1 << depth, 3), dtype=np.uint8) if color == 3 else None
    data = encode_png(smp, depth, color, pal, None, inter, 6)
    (out / f"ok_{n}.png").write_bytes(data)
    for k in range(6):
        b = bytearray(data)
        if k < 3:
            b = b[:len(b) * (k + 1) // 5]
        else:
            for _ in range(4):
                b[rng.integers(8, len(b))] = rng.integers(0, 256)
        (out / f"bad_{n}_{k}.png").write_bytes(bytes(b))
    n += 1


def flips(raw, name, count, lo, hi, first=0, per=3):
    for k in range(count):
        b = bytearray(raw)
        for _ in range(per):
            b[rng.integers(first, len(b))] = rng.integers(lo, hi)
        (out / name.format(k)).write_bytes(bytes(b))


# JPEG: valid files over subsampling modes and restart intervals, truncations, flips in headers and in the scan
try:
    import io
    from PIL import Image as _PIL
    from tests.test_jpeg_decode import _picture
    for k, kw in enumerate([dict(subsampling=0), dict(subsampling=1), dict(subsampling=2), dict(subsampling=2, optimize=True),
                            dict(subsampling=2, restart_marker_blocks=2), dict(subsampling=2, progressive=True),
                            dict(subsampling=0, progressive=True, optimize=True)]):
        buf = io.BytesIO()
        try:
            _PIL.fromarray(_picture(45, 31, k)).save(buf, "JPEG", quality=80, **kw)
        except TypeError:
            continue
        raw = buf.getvalue()
        (out / f"ok_{k}.jpg").write_bytes(raw)
        for t in range(1, 6):
            (out / f"cut_{k}_{t}.jpg").write_bytes(raw[: len(raw) * t // 6])
        flips(raw, f"flip_{k}_{{}}.jpg", 24, 0, 256, first=2, per=2)
        # crafted: quantisers at their maximum so that DC * q and the IDCT sums leave 16 / 31 bits; scan data
        # replaced by long runs of one-bits and of random bytes (unbounded DC deltas); 0xFF fill bytes before RSTn
        b = bytearray(raw)
        pos = 2
        while pos + 4 <= len(b) and b[pos] == 0xFF and b[pos + 1] != 0xDA:
            ln = (b[pos + 2] << 8) | b[pos + 3]
            if b[pos + 1] == 0xDB:
                for q in range(pos + 5, pos + 2 + ln):
                    b[q] = 0xFF
            pos += 2 + ln
        (out / f"maxq_{k}.jpg").write_bytes(bytes(b))
        sos = raw.find(b"\xff\xda")
        if sos > 0:
            hdr = sos + 2 + ((raw[sos + 2] << 8) | raw[sos + 3])
            for t, fill in enumerate((b"\xfe" * 4000, bytes(rng.integers(0, 255, 4000, dtype=np.uint8).tolist()), b"\x7f\xfe" * 2000)):
                (out / f"scan_{k}_{t}.jpg").write_bytes(bytes(b[:hdr]) + fill + b"\xff\xd9")
            (out / f"fill_{k}.jpg").write_bytes(raw.replace(b"\xff\xd0", b"\xff\xff\xff\xd0").replace(b"\xff\xd1", b"\xff\xff\xd1"))
except ImportError:
    pass


flips(open(G._author(out, glb=False, embed=True)[0], "rb").read(), "fuzz_{}.gltf", 40, 32, 127)
flips(open(G._author(out, glb=True, embed=True)[0], "rb").read(), "fuzz_{}.glb", 40, 0, 256, first=12)
flips(open(os.path.join(ROOT, "assets", "teapot.obj"), "rb").read()[:20000], "fuzz_{}.obj", 40, 32, 127, per=20)
flips(open(os.path.join(ROOT, "assets", "scenes", "multi.json"), "rb").read(), "fuzz_{}.json", 40, 32, 127)
# crafted strides / counts whose product wraps size_t (gltf_file.cpp accessor bounds check)
_doc = json.load(open(G._author(out, glb=False, embed=True)[0]))
for k, (stride, count) in enumerate([(1 << 40, (1 << 24) + 1), (1 << 62, 5), (252, 1 << 28), (0xFFFFFFFFFFFF, 1 << 20)]):
    d = json.loads(json.dumps(_doc))
    d["bufferViews"][d["accessors"][0]["bufferView"]]["byteStride"] = stride
    d["accessors"][0]["count"] = count
    (out / f"stride_{k}.gltf").write_text(json.dumps(d))
(out / "neg.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf -1 -2 -3\nf 1 2 9\nf 1/5/7 2/2/2 3/3/3\nf 0 0 0\nvt 1\nvn\nf 1 2\n")
print(len(list(out.iterdir())), "files in", out)
