"""Image files for Renderer::add_lambert_material(path) (src/Renderer.cpp:705-716: stbi_load(path, .., 4)) and for
`map_Kd` of OBJ materials (dead loader src/Scene.cpp:62-65).

The host library decodes PNG and binary PNM itself (raytracing_amd/csrc/host/image_decode.cpp).  PNG is lossless, so
the expected bytes are known exactly: the tests encode images with an encoder of their own (all colour types, bit
depths, filters, Adam7, colour keys; Python's zlib at several levels, i.e. stored, fixed and dynamic DEFLATE blocks),
compute the RGBA8 result by stb_image's published conversion rules in numpy, and cross-check against Pillow where
Pillow's conversion means the same thing (8-bit modes).  stb_image itself is not available here (third-party header,
not copied), so equality with its output is by construction of the format, not by running it: parity unpinned."""
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

F = np.float32
CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


# ---- a PNG encoder for the tests ---------------------------------------------------------------------------------
def _chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def _pack_rows(samples, depth):
    """samples [h, w, c] (uint16) -> list of packed scanlines (bytes)."""
    h, w, c = samples.shape
    rows = []
    for y in range(h):
        flat = samples[y].reshape(-1).astype(np.uint32)
        if depth == 16:
            rows.append(flat.astype(">u2").tobytes())
        elif depth == 8:
            rows.append(flat.astype(np.uint8).tobytes())
        else:
            per = 8 // depth
            pad = (-len(flat)) % per
            v = np.concatenate([flat, np.zeros(pad, np.uint32)]).reshape(-1, per)
            shifts = np.arange(per - 1, -1, -1, dtype=np.uint32) * depth
            rows.append((v << shifts).sum(1).astype(np.uint8).tobytes())
    return rows


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def _filter_rows(rows, bpp, filters):
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for y, row in enumerate(rows):
        f = filters[y % len(filters)]
        enc = bytearray(len(row))
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[f]
            enc[i] = (v - pred) & 0xFF
        out.append(f)
        out += enc
        prev = row
    return bytes(out)


ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def encode_png(samples, depth, color, palette=None, trns=None, interlace=False, level=6, filters=(0, 1, 2, 3, 4),
               idat_split=None):
    samples = np.asarray(samples, np.uint16)
    h, w, c = samples.shape
    assert c == CHANNELS[color]
    bpp = max(1, c * depth // 8)
    if interlace:
        raw = b""
        for x0, y0, dx, dy in ADAM7:
            sub = samples[y0::dy, x0::dx]
            if sub.size:
                raw += _filter_rows(_pack_rows(sub, depth), bpp, filters)
    else:
        raw = _filter_rows(_pack_rows(samples, depth), bpp, filters)
    z = zlib.compress(raw, level)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, int(interlace)))
    out += _chunk(b"tEXt", b"Comment\0ancillary chunks are skipped")
    if palette is not None:
        out += _chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        out += _chunk(b"tRNS", trns)
    if idat_split:
        for i in range(0, len(z), idat_split):
            out += _chunk(b"IDAT", z[i:i + idat_split])
    else:
        out += _chunk(b"IDAT", z)
    return out + _chunk(b"IEND", b"")


def expected_rgba(samples, depth, color, palette=None, trns=None):
    """stb_image's conversion to 4 x 8 bit (stbi__convert_format, stbi__compute_transparency, depth scaling)."""
    s = np.asarray(samples, np.uint32)
    h, w, c = s.shape
    out = np.zeros((h, w, 4), np.uint8)
    if color == 3:
        pal = np.zeros((256, 4), np.uint8)
        pal[:, 3] = 255
        p = np.asarray(palette, np.uint8).reshape(-1, 3)
        pal[:len(p), :3] = p
        if trns is not None:
            pal[:len(trns), 3] = np.frombuffer(trns, np.uint8)
        return pal[s[..., 0]]
    scale = {1: 0xFF, 2: 0x55, 4: 0x11, 8: 1}.get(depth, 1)
    to8 = (s >> 8) if depth == 16 else (s * scale if depth < 8 else s)
    to8 = to8.astype(np.uint8)
    keyed = np.zeros((h, w), bool)
    if trns is not None:
        key = np.frombuffer(trns, ">u2").astype(np.uint32)
        if depth == 16:
            keyed = (s[..., :len(key)] == key).all(-1)
        else:
            keyed = (to8[..., :len(key)] == ((key & 255) * scale).astype(np.uint8)).all(-1)
    if color == 0:
        out[..., 0] = out[..., 1] = out[..., 2] = to8[..., 0]
        out[..., 3] = np.where(keyed, 0, 255)
    elif color == 4:
        out[..., 0] = out[..., 1] = out[..., 2] = to8[..., 0]
        out[..., 3] = to8[..., 1]
    elif color == 2:
        out[..., :3] = to8
        out[..., 3] = np.where(keyed, 0, 255)
    else:
        out[...] = to8
    return out


def _rand(rng, h, w, c, depth):
    return rng.integers(0, 1 << depth, (h, w, c), dtype=np.uint32).astype(np.uint16)


CASES = []
for _color, _depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))):
    for _depth in _depths:
        for _interlace in (False, True):
            CASES.append((_color, _depth, _interlace))


@pytest.mark.parametrize("color,depth,interlace", CASES)
def test_png_decodes_to_the_expected_rgba(tmp_path, color, depth, interlace):
    rng = np.random.default_rng(color * 100 + depth * 2 + interlace)
    for k, (h, w) in enumerate([(1, 1), (5, 7), (17, 33), (64, 50)]):
        c = CHANNELS[color]
        smp = _rand(rng, h, w, c, depth)
        palette = trns = None
        if color == 3:
            n = 1 << depth
            palette = rng.integers(0, 256, (n, 3), dtype=np.uint8)
            trns = rng.integers(0, 256, max(1, n // 2), dtype=np.uint8).tobytes() if k % 2 else None
        elif color in (0, 2) and k % 2:  # colour key = the first pixel's value
            trns = b"".join(struct.pack(">H", int(v)) for v in smp[0, 0])
            smp[h // 2, w // 2] = smp[0, 0]
        level = (0, 1, 6, 9)[k]
        data = encode_png(smp, depth, color, palette, trns, interlace, level, idat_split=(None, 7, 100, 4096)[k])
        path = tmp_path / f"c{color}_d{depth}_{k}.png"
        path.write_bytes(data)
        got = R.decode_image(str(path))
        want = expected_rgba(smp, depth, color, palette, trns)
        assert got.shape == want.shape and np.array_equal(got, want), (color, depth, interlace, h, w)


def test_png_fixed_huffman_and_long_matches(tmp_path):
    """Z_FIXED blocks and long back-references (distance and length codes with extra bits)."""
    rng = np.random.default_rng(3)
    tile = rng.integers(0, 256, (8, 300, 3), dtype=np.uint16)
    smp = np.tile(tile, (40, 4, 1))  # repeats 2400 and 900 bytes apart
    rows = _filter_rows(_pack_rows(smp, 8), 3, (0,))
    for strategy in (zlib.Z_FIXED, zlib.Z_DEFAULT_STRATEGY, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY):
        co = zlib.compressobj(9, zlib.DEFLATED, 15, 9, strategy)
        z = co.compress(rows) + co.flush()
        h, w, _ = smp.shape
        data = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + \
            _chunk(b"IDAT", z) + _chunk(b"IEND", b"")
        path = tmp_path / f"s{strategy}.png"
        path.write_bytes(data)
        assert np.array_equal(R.decode_image(str(path))[..., :3], smp.astype(np.uint8))


def test_png_agrees_with_pillow_on_8_bit_modes(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(11)
    for mode, c in (("L", 1), ("LA", 2), ("RGB", 3), ("RGBA", 4)):
        a = rng.integers(0, 256, (37, 53, c), dtype=np.uint8)
        im = PIL.fromarray(a[..., 0] if c == 1 else a, mode)
        for level in (0, 9):
            path = tmp_path / f"{mode}_{level}.png"
            im.save(path, compress_level=level)
            assert np.array_equal(R.decode_image(str(path)), np.asarray(PIL.open(path).convert("RGBA")))
    pim = PIL.fromarray(rng.integers(0, 256, (20, 31, 3), dtype=np.uint8), "RGB").quantize(16)
    path = tmp_path / "pal.png"
    pim.save(path, bits=4)
    assert np.array_equal(R.decode_image(str(path)), np.asarray(PIL.open(path).convert("RGBA")))
    one = PIL.fromarray(rng.integers(0, 2, (9, 13), dtype=np.uint8) * 255, "L").convert("1")
    path = tmp_path / "one.png"
    one.save(path)
    assert np.array_equal(R.decode_image(str(path)), np.asarray(PIL.open(path).convert("RGBA")))


@pytest.mark.parametrize("name", ["Screenshot.png", "Screenshot2.png", "Screenshot3.png"])
def test_png_files_of_the_reference_repository(name):
    """Real-world encoder output (read in place; only in the build container, where the reference is mounted)."""
    PIL = pytest.importorskip("PIL.Image")
    path = os.path.join("/root/reference", name)
    if not os.path.isfile(path):
        pytest.skip("reference checkout not present")
    assert np.array_equal(R.decode_image(path), np.asarray(PIL.open(path).convert("RGBA")))


def test_png_writer_round_trips(tmp_path):
    """The command line's PNG writer (image_encode.cpp: row filters + fixed-Huffman DEFLATE with LZ77 matches):
    Pillow and the product's decoder read back the pixels that were written; smooth images shrink."""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(8)
    yy, xx = np.mgrid[0:90, 0:160]
    images = {
        "noise": rng.integers(0, 256, (37, 53, 3), dtype=np.uint8),
        "gradient": np.stack([xx * 255 // 159, yy * 255 // 89, (xx + yy) % 256], -1).astype(np.uint8),
        "flat": np.full((64, 64, 3), 200, np.uint8),
        "tiles": np.tile(rng.integers(0, 256, (8, 8, 3), dtype=np.uint8), (20, 30, 1)),
        "one": rng.integers(0, 256, (1, 1, 3), dtype=np.uint8),
        "row": rng.integers(0, 256, (1, 300, 3), dtype=np.uint8),
    }
    for name, img in images.items():
        path = str(tmp_path / (name + ".png"))
        R.write_png_rgb8(path, img)
        assert np.array_equal(np.asarray(PIL.open(path).convert("RGB")), img), name
        assert np.array_equal(R.decode_image(path)[..., :3], img), name
        PIL.open(path).verify()
    for name in ("gradient", "flat", "tiles"):
        assert os.path.getsize(tmp_path / (name + ".png")) < images[name].size // 4, name


def test_pnm(tmp_path):
    rng = np.random.default_rng(5)
    g = rng.integers(0, 256, (6, 9), dtype=np.uint8)
    (tmp_path / "g.pgm").write_bytes(b"P5\n# a comment\n9 6\n255\n" + g.tobytes())
    got = R.decode_image(str(tmp_path / "g.pgm"))
    assert np.array_equal(got[..., 0], g) and np.array_equal(got[..., 2], g) and (got[..., 3] == 255).all()
    c = rng.integers(0, 65536, (4, 5, 3), dtype=np.uint16)
    (tmp_path / "c.ppm").write_bytes(b"P6 5 4 65535\n" + c.astype(">u2").tobytes())
    got = R.decode_image(str(tmp_path / "c.ppm"))
    assert np.array_equal(got[..., :3], (c >> 8).astype(np.uint8)) and (got[..., 3] == 255).all()


def test_bad_images_are_reported(tmp_path):
    good = encode_png(_rand(np.random.default_rng(0), 8, 8, 3, 8), 8, 2)
    cases = {"trunc.png": good[:len(good) // 2], "sig.png": b"\x89PNG\r\n\x1a\n", "jpeg.jpg": b"\xff\xd8\xff\xe0" + b"\0" * 64,
             "text.txt": b"hello", "pnm.ppm": b"P6 4 4 255\n\0\0", "empty.png": b""}
    bad_idat = bytearray(good)
    i = bad_idat.index(b"IDAT") + 10
    bad_idat[i:i + 8] = b"\xff" * 8
    cases["corrupt.png"] = bytes(bad_idat)
    cases["depth.png"] = good.replace(struct.pack(">IIBB", 8, 8, 8, 2), struct.pack(">IIBB", 8, 8, 3, 2))
    for name, data in cases.items():
        (tmp_path / name).write_bytes(data)
        with pytest.raises(R.RendererError):
            R.decode_image(str(tmp_path / name))
    with pytest.raises(R.RendererError, match="cannot open"):
        R.decode_image(str(tmp_path / "missing.png"))
    r = R.Renderer(8, 8, device=None)
    assert r.add_lambert_image(str(tmp_path / "jpeg.jpg")).type == T.MATERIAL_NONE


# ---- through the renderer ----------------------------------------------------------------------------------------
def _textured_scene(tmp_path):
    """Two quads with map_Kd textures (one PNG with texture options in the .mtl, one PGM) and one Kd-only quad."""
    rng = np.random.default_rng(21)
    tex = rng.integers(0, 256, (16, 8, 3), dtype=np.uint16)
    (tmp_path / "checker tex.png").write_bytes(encode_png(tex, 8, 2, interlace=True))
    g = rng.integers(0, 256, (4, 4), dtype=np.uint8)
    (tmp_path / "g.pgm").write_bytes(b"P5 4 4 255\n" + g.tobytes())
    (tmp_path / "quads.mtl").write_text(
        "newmtl a\nKd 0.1 0.2 0.3\nmap_Kd -s 1 1 1 -clamp on checker tex.png\n"
        "newmtl b\nKd 0.9 0.8 0.1\n"
        "newmtl c\nmap_Kd g.pgm\n")
    obj = ["mtllib quads.mtl"]
    for k, name in enumerate("abc"):
        x = 2.2 * k - 2.2
        obj += [f"v {x - 1} 0 0", f"v {x + 1} 0 0", f"v {x + 1} 2 0", f"v {x - 1} 2 0"]
    obj += ["vt 0 0", "vt 1 0", "vt 1 1", "vt 0 1", "vn 0 0 1"]
    for k, name in enumerate("abc"):
        b = 4 * k
        obj += [f"usemtl {name}", f"f {b + 1}/1/1 {b + 2}/2/1 {b + 3}/3/1 {b + 4}/4/1"]
    (tmp_path / "quads.obj").write_text("\n".join(obj) + "\n")
    scene = {"camera": {"screen_width": 64, "aspect_ratio": 2.0, "samples_per_pixel": 4, "max_depth": 3,
                        "vfov_deg": 40.0, "center": [0, 1, 6], "lookat": [0, 1, 0], "vup": [0, 1, 0],
                        "defocus_angle": 0.0, "focus_dist": 1.0},
             "models": [{"model_path": "quads.obj"}]}
    import json
    (tmp_path / "scene.json").write_text(json.dumps(scene))
    rgba_a = expected_rgba(tex, 8, 2)
    rgba_c = np.stack([g, g, g, np.full_like(g, 255)], -1)
    return str(tmp_path / "scene.json"), scene, rgba_a, rgba_c


def _expected_textured(tmp_path, rgba_a, rgba_c):
    o = O.OracleRenderer()
    for which in (0, 1, 2):
        o.add_blas(*R.Mesh.primitive(which).arrays())
    o.add_lambert_material((0.8, 0.0, 0.8))  # the loader's fallback, Scene.cpp:182
    ident = np.eye(4, dtype=F).reshape(-1)
    groups = R.Mesh.load_obj(str(tmp_path / "quads.obj"))
    assert len(groups) == 3
    for k, (mesh, kd, has) in enumerate(groups):
        if k == 0:
            mat = o.add_lambert_texture(rgba_a.shape[1], rgba_a.shape[0], rgba_a.reshape(-1).tolist())
        elif k == 2:
            mat = o.add_lambert_texture(4, 4, rgba_c.reshape(-1).tolist())
        else:
            mat = o.add_lambert_material(tuple(kd))
        o.add_blas_instance(o.add_blas(*mesh.arrays()), ident, mat)
    return o.build()


def test_map_kd_textures_reach_the_scene_arrays(tmp_path):
    path, scene, rgba_a, rgba_c = _textured_scene(tmp_path)
    r = R.Renderer(8, 8, device=None)
    r.load_scene_file(path)
    got, want = r.scene_arrays(), _expected_textured(tmp_path, rgba_a, rgba_c)
    for name in ("lamberts", "textures", "texels", "instances", "tri_shading"):
        assert getattr(got, name).tobytes() == getattr(want, name).tobytes(), name
    # a missing texture is an error of the loader, not a silent fallback
    os.remove(tmp_path / "g.pgm")
    with pytest.raises(R.RendererError, match="g.pgm"):
        R.Renderer(8, 8, device=None).load_scene_file(path)


@pytest.mark.gpu
def test_cli_renders_map_kd_textures_bit_exactly(tmp_path):
    import math
    path, scene, rgba_a, rgba_c = _textured_scene(tmp_path)
    cli = os.path.join(os.path.dirname(os.path.abspath(R.__file__)), "bin", "renderer")
    raw = str(tmp_path / "out.f32")
    p = subprocess.run([cli, path, "-hip", "-o", str(tmp_path / "out.png"), "--raw", raw], capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    c = scene["camera"]
    w, h = 64, 32
    got = np.fromfile(raw, np.float32).reshape(h, w, 4)
    host = R.Renderer(w, h, device=None)
    camu = host.camera_uniforms(R.Camera(c["center"], c["lookat"], c["vup"], c["vfov_deg"], 0.0, 1.0))
    sc = _expected_textured(tmp_path, rgba_a, rgba_c)
    want = O.render(sc, camu, O.render_params(w, h, 0, int(math.sqrt(c["samples_per_pixel"])), c["max_depth"]), nthreads=8)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert len(np.unique(got[..., :3].reshape(-1, 3), axis=0)) > 50  # the textures are visible
