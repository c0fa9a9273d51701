"""Scene files and the command line: the reference's JSON scene
description (README.md:53-68, Scenes/TestScene.json; loader semantics of the dead
src/Scene.cpp:134-233) loaded by raytracing_amd/csrc/host/scene_file.cpp, and
`bin/renderer <scene>.json -hip -o out.png`."""
import json
import math
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from raytracing_amd import renderer as R
from raytracing_amd import _types as T
from . import _oracle as O
from . import _scenes as S

SCENES = os.path.join(os.path.dirname(S.ASSETS.rstrip("/")), "assets", "scenes") \
    if not os.path.isdir(os.path.join(S.ASSETS, "scenes")) else os.path.join(S.ASSETS, "scenes")
CLI = os.path.join(os.path.dirname(os.path.abspath(R.__file__)), "bin", "renderer")
F = np.float32


def _expected_transform(model):
    """Scene.cpp:195-223: each factor pre-multiplied in the order scale, rotate, translate."""
    L = O.lib()
    t = np.zeros(16, F); L.orc_mat4_identity(T.ptr(t))
    ident = t.copy()

    def mul(a, b):
        out = np.zeros(16, F)
        L.orc_mat4_mul(T.ptr(a), T.ptr(b), T.ptr(out))
        return out
    if "scale" in model:
        s = np.zeros(16, F); sc = F(model["scale"])
        L.orc_mat4_scale(T.ptr(ident), T.ptr(O.f32a([sc, sc, sc])), T.ptr(s))
        t = mul(s, t)
    if "rotation" in model:
        x, y, z, deg = [F(v) for v in model["rotation"]]
        ln = F(np.sqrt(F(F(x * x) + F(y * y)) + F(z * z)))
        inv = F(1.0) / ln
        q = O.angle_axis(F(deg) * F(0.01745329251994329576923690768489), [x * inv, y * inv, z * inv])
        rot = O.trs((0, 0, 0), q, (1, 1, 1))  # translate(I,0) * toMat4(q) * scale(I,1) == toMat4(q) exactly
        t = mul(rot, t)
    if "translation" in model:
        tr = np.zeros(16, F)
        L.orc_mat4_translate(T.ptr(ident), T.ptr(O.f32a(model["translation"])), T.ptr(tr))
        t = mul(tr, t)
    return t


def _expected_scene(json_path):
    """The scene the file describes, assembled through the ORACLE's renderer."""
    doc = json.load(open(json_path))
    o = O.OracleRenderer()
    for which in (0, 1, 2):  # plane, cube, sphere of Renderer::init (Renderer.cpp:332-334); the product's
        o.add_blas(*R.Mesh.primitive(which).arrays())  # meshes, since the sphere goes through libm sinf/cosf
    fallback = o.add_lambert_material((0.8, 0.0, 0.8))
    for m in doc.get("models", []):
        path = os.path.normpath(os.path.join(os.path.dirname(json_path), m["model_path"]))
        t = _expected_transform(m)
        for mesh, kd, has in R.Mesh.load_obj(path):
            mat = o.add_lambert_material(tuple(kd)) if has else fallback
            b = o.add_blas(*mesh.arrays())
            o.add_blas_instance(b, t, mat)
    return doc, o.build()


def _same_arrays(a, b):
    for name in ("tri_geom", "tri_shading", "tri_ids", "bvh_nodes", "blas", "instances", "lamberts", "textures", "texels"):
        x, y = getattr(a, name), getattr(b, name)
        assert x.tobytes() == y.tobytes(), name
    assert a.tlas_node_count == b.tlas_node_count
    n = a.tlas_node_count
    assert a.tlas_nodes[:n].tobytes() == b.tlas_nodes[:n].tobytes()


@pytest.mark.parametrize("name", ["cornell.json", "teapot.json", "multi.json"])
def test_scene_file_assembles_the_scene_the_oracle_builds(name):
    path = os.path.join(SCENES, name)
    doc, want = _expected_scene(path)
    r = R.Renderer(8, 8, device=None)  # host-only: scene assembly and builders
    cam, info = r.load_scene_file(path)
    c = doc["camera"]
    assert info["screen_width"] == c["screen_width"]
    assert info["screen_height"] == max(1, int(F(c["screen_width"]) / F(c["aspect_ratio"])))
    assert info["samples_per_pixel"] == c["samples_per_pixel"] and info["max_depth"] == c["max_depth"]
    assert info["model_count"] == len(doc["models"])
    assert list(cam.c.position) == [F(v) for v in c["center"]] and list(cam.c.look_at) == [F(v) for v in c["lookat"]]
    assert cam.c.fov == F(c["vfov_deg"]) and cam.c.defocus_angle == F(c["defocus_angle"])
    got = r.scene_arrays()
    assert info["triangles"] > 0
    _same_arrays(got, want)


def test_scene_file_errors_are_reported(tmp_path):
    r = R.Renderer(8, 8, device=None)
    with pytest.raises(R.RendererError, match="not found"):
        r.load_scene_file(str(tmp_path / "missing.json"))
    bad = tmp_path / "bad.json"
    bad.write_text('{ "camera": { "center": [0, 0, 1], }')
    with pytest.raises(R.RendererError, match="line 1"):
        r.load_scene_file(str(bad))
    nocam = tmp_path / "nocam.json"
    nocam.write_text('{ "models": [] }')
    with pytest.raises(R.RendererError, match="camera"):
        r.load_scene_file(str(nocam))
    doc = json.load(open(os.path.join(SCENES, "teapot.json")))
    del doc["camera"]["vfov_deg"]  # every camera key is required (Scene.cpp:152-179)
    part = tmp_path / "part.json"
    part.write_text(json.dumps(doc))
    with pytest.raises(R.RendererError, match="vfov_deg"):
        r.load_scene_file(str(part))
    doc = json.load(open(os.path.join(SCENES, "teapot.json")))
    doc["models"][0]["model_path"] = "nope.obj"
    nomodel = tmp_path / "nomodel.json"
    nomodel.write_text(json.dumps(doc))
    with pytest.raises(R.RendererError, match="nope.obj"):
        r.load_scene_file(str(nomodel))
    onlycam = tmp_path / "onlycam.json"  # `models` is optional (Scene.cpp:185-186)
    del doc["models"]
    onlycam.write_text(json.dumps(doc))
    cam, info = r.load_scene_file(str(onlycam))
    assert info["model_count"] == 0 and info["triangles"] == 0


def test_cli_refuses_other_renderers_and_bad_usage():
    assert os.path.exists(CLI), "bin/renderer is missing: run __graft_entry__.build()"
    scene = os.path.join(SCENES, "teapot.json")
    for flag, word in (("-cpu", "no CPU renderer"), ("-gpu", "Vulkan")):
        p = subprocess.run([CLI, scene, flag], capture_output=True, text=True)
        assert p.returncode == 2 and word in p.stderr
    p = subprocess.run([CLI, scene], capture_output=True, text=True)
    assert p.returncode == 2 and "usage" in p.stderr
    p = subprocess.run([CLI, "/nonexistent.json", "-hip"], capture_output=True, text=True)
    assert p.returncode == 1 and "not found" in p.stderr


def _read_png_rgb8(path):
    """Decoded by Pillow (independent of the product) and by the product's own decoder: both must agree."""
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB"))
    assert np.array_equal(img, R.decode_image(path)[..., :3])
    return img


def _tone_map(img):  # FullScreen.slang:38-55 + Image.hpp:62-66
    c = np.maximum(img[..., :3].astype(F), F(0))
    a, b, cc, d, e = F(2.51), F(0.03), F(2.43), F(0.59), F(0.14)
    m = (c * (a * c + b)) / (c * (cc * c + d) + e)
    out = (np.clip(m, 0, 1) * F(255)).astype(np.uint8)
    out[m >= 1] = 255
    out[~(m > 0)] = 0
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name,width,spp,depth", [("cornell.json", 96, 4, 4), ("multi.json", 120, 4, 5)])
def test_cli_renders_the_scene_file_bit_exactly(tmp_path, name, width, spp, depth):
    path = os.path.join(SCENES, name)
    png, raw = str(tmp_path / "out.png"), str(tmp_path / "out.f32")
    p = subprocess.run([CLI, path, "-hip", "-o", png, "--raw", raw, "--width", str(width), "--spp", str(spp),
                        "--depth", str(depth)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    doc, sc = _expected_scene(path)
    c = doc["camera"]
    aspect = c["screen_width"] / max(1, int(F(c["screen_width"]) / F(c["aspect_ratio"])))
    h = max(1, int(width / aspect))
    got = np.fromfile(raw, np.float32).reshape(h, width, 4)
    host = R.Renderer(width, h, device=None)
    cam = R.Camera(c["center"], c["lookat"], c["vup"], c["vfov_deg"], c["defocus_angle"], c["focus_dist"])
    camu = host.camera_uniforms(cam)
    want = O.render(sc, camu, O.render_params(width, h, 0, int(math.sqrt(spp)), depth), nthreads=8)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(_read_png_rgb8(png), _tone_map(got))
