"""CPU tests of the C++ host tool: its PNG boundary code and the argument handling
it mirrors from the reference's main (cube.cpp:394-453)."""
import os
import subprocess

import numpy as np
import pytest

from helpers import read_png, write_png, write_png_general

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "vulkan-renderer_amd", "bin")
CUBE = os.path.join(BIN, "cube")
ROUNDTRIP = os.path.join(BIN, "png_roundtrip")


def _run(*args):
    return subprocess.run(list(args), capture_output=True, text=True, timeout=60)


@pytest.mark.parametrize("color_type,filters", [(6, (0,)), (6, (0, 1, 2, 3, 4)), (2, (4, 3, 1)), (6, (4,))])
def test_png_io_roundtrip(tmp_path, color_type, filters):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(37, 23, 4), dtype=np.uint8)
    img[5:20, 3:9] = img[5, 3]          # flat region, so the filters see repeats
    src, dst = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    write_png(src, img, color_type=color_type, filters=filters)
    r = _run(ROUNDTRIP, src, dst)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["23", "37"]
    got = read_png(dst)
    want = img.copy()
    if color_type == 2:
        want[..., 3] = 255
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("color_type,depth", [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4),
                                              (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)])
@pytest.mark.parametrize("interlace", [False, True])
def test_png_reader_takes_every_format_stb_takes(tmp_path, color_type, depth, interlace):
    """cube.cpp:458 loads through stb_image with STBI_rgb_alpha: any bit depth, colour type and
    Adam7.  Expected pixels by stb's rules: 16-bit samples keep their high byte, low-depth grey is
    scaled to 0..255, a palette is looked up (with its tRNS alphas), a tRNS colour key turns the
    matching pixels transparent."""
    import importlib
    png = importlib.import_module("vulkan_renderer_amd.pngio")
    rng = np.random.default_rng(depth * 10 + color_type)
    h, w = 13, 11                                           # not multiples of 8: short Adam7 passes
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    smp = rng.integers(0, 1 << depth, size=(h, w, ch), dtype=np.uint16)
    palette = trns = None
    if color_type == 3:
        palette = rng.integers(0, 256, size=(1 << depth, 3), dtype=np.uint8)
        trns = bytes(rng.integers(0, 256, size=min(1 << depth, 5), dtype=np.uint8))
    elif color_type in (0, 2) and depth in (8, 16):
        smp[3, 4] = smp[0, 0]                               # a colour key that two pixels match
        trns = b"".join(int(v).to_bytes(2, "big") for v in smp[0, 0])
    path = str(tmp_path / "x.png")
    write_png_general(path, smp, depth, color_type, interlace, palette, trns)

    def to8(v):
        return (v >> 8).astype(np.uint8) if depth == 16 else (v * {1: 255, 2: 85, 4: 17, 8: 1}[depth]).astype(np.uint8)
    want = np.zeros((h, w, 4), np.uint8)
    if color_type == 3:
        want[..., :3] = palette[smp[..., 0]]
        alpha = np.full(1 << depth, 255, np.uint8)
        alpha[:len(trns)] = np.frombuffer(trns, np.uint8)
        want[..., 3] = alpha[smp[..., 0]]
    else:
        if color_type in (0, 4):
            want[..., :3] = to8(smp[..., :1])
        else:
            want[..., :3] = to8(smp[..., :3])
        if color_type in (4, 6):
            want[..., 3] = to8(smp[..., -1])
        elif trns is not None:
            want[..., 3] = np.where((smp == smp[0, 0]).all(axis=-1), 0, 255)
        else:
            want[..., 3] = 255
    np.testing.assert_array_equal(png.read_rgba8(path), want)
    r = _run(ROUNDTRIP, path, str(tmp_path / "y.png"))      # and through the tool's own binary
    assert r.returncode == 0, r.stderr
    np.testing.assert_array_equal(read_png(str(tmp_path / "y.png")), want)


def test_png_io_rejects_garbage(tmp_path):
    p = tmp_path / "x.png"
    p.write_bytes(b"not a png at all")
    r = _run(ROUNDTRIP, str(p), str(tmp_path / "y.png"))
    assert r.returncode == 1 and "not a PNG" in r.stderr
    # a crafted header: 60000 x 60000 texels of 16-bit RGBA would wrap the reader's size arithmetic
    import struct
    import zlib

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data))
    big = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 60000, 60000, 16, 6, 0, 0, 0))
           + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b""))
    p.write_bytes(big)
    r = _run(ROUNDTRIP, str(p), str(tmp_path / "y.png"))
    assert r.returncode == 1 and "too large" in r.stderr
    import importlib
    png = importlib.import_module("vulkan_renderer_amd.pngio")
    with pytest.raises(Exception, match="too large"):
        png.read_rgba8(str(p))


@pytest.mark.parametrize("args,message", [
    ([], "command line arguments do not have an input file listed"),
    (["--cubemap-size"], "missing argument to --cubemap-size"),
    (["--cubemap-size", "abc", "x.png"], "invalid argument to --cubemap-size"),
    (["--input-downscale"], "missing argument to --input-downscale"),
    (["--input-downscale", "3", "x.png"], "scale factor for --input-downscale should be a power of 2 >= 1"),
    (["--input-downscale", "0", "x.png"], "scale factor for --input-downscale should be a power of 2 >= 1"),
    (["a.png", "b.png"], "invalid command line argument: b.png"),
    (["--lambertian", "/nonexistent/file.png"], "failed to load texture image!"),
])
def test_cli_argument_errors_match_reference_messages(args, message):
    r = _run(CUBE, *args)
    assert r.returncode != 0
    assert message in r.stderr


def test_cli_without_work_flags_only_loads(tmp_path):
    # cube.cpp:509-516: neither --lambertian nor --ggx -> load, convert, exit 0, no files
    img = np.zeros((6 * 4, 4, 4), np.uint8)
    src = tmp_path / "env.png"
    write_png(str(src), img)
    r = _run(CUBE, str(src))
    assert r.returncode == 0, r.stderr
    assert sorted(os.listdir(tmp_path)) == ["env.png"]


SCENE = """["s72-v1",
 {"type":"MESH","name":"m","topology":"TRIANGLE_LIST","count":3},
 {"type":"ENVIRONMENT","name":"sky","radiance":{"src":"env/sky.png","type":"cube","format":"rgbe"}},
 {"type":"NODE","name":"n \\"quoted\\"","environment":2,"children":[]},
 {"type":"ENVIRONMENT","name":"room","local":{"minCorner":[0,0,0],"maxCorner":[1,1,1.5e0]}},
 {"type":"ENVIRONMENT","name":"mirror","local":{"mesh":1,"distance":2.5}}
]"""


def test_scene_mode_lists_environments_like_the_loader(tmp_path):
    """scene_loader.cpp:113-116,509-535: environments are numbered in order of
    appearance; a radiance cube is <scene dir>/<src>, a local one
    <scene>.localIBL_<number>.png."""
    scene = tmp_path / "demo.s72"
    scene.write_text(SCENE)
    (tmp_path / "env").mkdir()
    (tmp_path / "env" / "sky.png").write_bytes(b"")
    r = _run(CUBE, "--scene", str(scene), "--list-environments")
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == [
        'Environment 0 "sky": %s' % (tmp_path / "env" / "sky.png"),
        'Environment 1 "room": %s (local) (missing)' % (tmp_path / "demo.localIBL_1.png"),
        'Environment 2 "mirror": %s (local) (missing)' % (tmp_path / "demo.localIBL_2.png"),
    ]


@pytest.mark.parametrize("text,message", [
    ('["s72-v2"]', "first element of scene JSON is not correct magic string"),
    ('{"a":1}', "scene JSON is not an array"),
    ('["s72-v1", {"type":"ENVIRONMENT","name":"e","radiance":{"src":"a.png"}}]', "texture has an unexpected type"),
    ('["s72-v1", {"type":"ENVIRONMENT","name":"e","radiance":{"type":"cube"}}]', "texture does not have a source"),
    ('["s72-v1", {"type":"ENVIRONMENT"', "Scene loading error: JSON"),
])
def test_scene_mode_errors(tmp_path, text, message):
    scene = tmp_path / "bad.s72"
    scene.write_text(text)
    r = _run(CUBE, "--scene", str(scene), "--list-environments")
    assert r.returncode != 0
    assert message in r.stderr


def test_scene_mode_argument_errors():
    r = _run(CUBE, "--scene")
    assert r.returncode != 0 and "missing argument to --scene" in r.stderr
    r = _run(CUBE, "--scene", "/nonexistent.s72")
    assert r.returncode != 0 and "cannot open" in r.stderr


def test_multi_gpu_bake_argument_handling():
    """vulkan_renderer_amd.bake mirrors the tool's flags (cube.cpp:394-453)."""
    import sys
    r = subprocess.run([sys.executable, "-m", "vulkan_renderer_amd.bake", "--input-downscale", "3", "x.png"],
                       capture_output=True, text=True, timeout=60, cwd=ROOT)
    assert r.returncode != 0
    assert "scale factor for --input-downscale should be a power of 2 >= 1" in r.stderr
    r = subprocess.run([sys.executable, "-m", "vulkan_renderer_amd.bake"], capture_output=True, text=True, timeout=60, cwd=ROOT)
    assert r.returncode != 0 and "file" in r.stderr


def test_python_png_module_roundtrip(tmp_path):
    import importlib
    png = importlib.import_module("vulkan_renderer_amd.pngio")
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, size=(30, 17, 4), dtype=np.uint8)
    img[5:20, 3:9] = img[5, 3]
    for ct, filters in ((6, (0, 1, 2, 3, 4)), (2, (4, 3, 1))):
        p = str(tmp_path / "a.png")
        write_png(p, img, color_type=ct, filters=filters)
        want = img.copy()
        if ct == 2:
            want[..., 3] = 255
        np.testing.assert_array_equal(png.read_rgba8(p), want)                    # the C++ reader behind ctypes
        np.testing.assert_array_equal(png.read_rgba8(p, native=False), want)      # the numpy fallback
    assert png._native() is not None, "libibl_png.so is missing: run __graft_entry__.build()"
    p = str(tmp_path / "b.png")
    png.write_rgba8(p, img)
    np.testing.assert_array_equal(read_png(p), img)
    r = _run(ROUNDTRIP, p, str(tmp_path / "c.png"))          # and the C++ reader takes it
    assert r.returncode == 0, r.stderr
    np.testing.assert_array_equal(read_png(str(tmp_path / "c.png")), img)
