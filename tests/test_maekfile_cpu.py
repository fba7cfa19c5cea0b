"""The node build description (Maekfile.js): the rules it holds and a real build of the host targets.

The reference is built by `node Maekfile.js [targets]` (Maekfile.js:38,82-84); this repository's file of
the same name adds the HIP rule and the `bin/cube` target that links the tool against the HIP library
instead of the renderer."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")


def _maek(args, out, **env):
    e = dict(os.environ, OUT=str(out), **env)
    return subprocess.run([NODE, os.path.join(ROOT, "Maekfile.js")] + args, cwd=ROOT, env=e, capture_output=True,
                          text=True, timeout=600)


def test_dry_run_lists_every_rule(tmp_path):
    r = _maek(["-n"], tmp_path / "out")
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    hip = [l for l in lines if "--offload-arch=gfx950" in l and " -c " in l]
    sources = sorted(f for f in os.listdir(os.path.join(ROOT, "vulkan-renderer_amd", "csrc")) if f.endswith(".hip"))
    assert len(hip) == len(sources) == 14
    for s in sources:
        assert any(l.rstrip().split(" -o ")[0].endswith(s) for l in hip), s
    assert any("lambertian_split.hip" in l and "-fno-slp-vectorize" in l for l in hip)
    assert any("lambertian_hex.hip" in l and "-fno-slp-vectorize" in l for l in hip)
    assert any("lambertian_o48.hip" in l and "-fno-slp-vectorize" in l for l in hip)
    assert sum("-fno-slp-vectorize" in l for l in hip) == 3
    link_lib = [l for l in lines if "-shared" in l and l.split(" -o ")[1].split()[0].endswith("libibl_hip.so")]
    assert len(link_lib) == 1 and link_lib[0].count(".o") >= 12
    cube = [l for l in lines if " -o " in l and l.split(" -o ")[1].split()[0].endswith("bin/cube")]
    assert len(cube) == 1 and "-libl_hip" in cube[0] and "-lz" in cube[0]
    assert "vulkan" not in cube[0].replace("vulkan-renderer_amd", "")        # none of the renderer's objects
    assert not (tmp_path / "out").exists()                                   # a dry run writes nothing


def test_builds_the_host_targets_and_is_idempotent(tmp_path):
    lib = os.path.join(ROOT, "vulkan-renderer_amd", "libibl_hip.so")
    if not os.path.exists(lib):
        pytest.skip("libibl_hip.so is not built")
    out = tmp_path / "out"
    goals = ["bin/cube", "bin/png_roundtrip", "libibl_png.so"]
    r = _maek(goals, out, MAEK_LINK_TREE_LIB="1")
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("[maek] CPP") == 6 and r.stdout.count("[maek] LINK") == 3
    for g in goals:
        assert (out / g).exists(), g
    # the tool runs (its first complaint is the reference's, cube.cpp:451-453) without a GPU
    t = subprocess.run([str(out / "bin" / "cube")], capture_output=True, text=True, timeout=60)
    assert "input file" in (t.stdout + t.stderr)
    # nothing is stale afterwards; touching a header rebuilds what includes it and relinks
    r = _maek(goals, out, MAEK_LINK_TREE_LIB="1")
    assert r.returncode == 0 and "[maek]" not in r.stdout
    hdr = os.path.join(ROOT, "vulkan-renderer_amd", "host", "s72_scene.hpp")
    st = os.stat(hdr)
    try:
        os.utime(hdr, None)
        r = _maek(["-n"] + goals, out, MAEK_LINK_TREE_LIB="1")
        assert r.returncode == 0
        rebuilt = r.stdout
        assert "s72_scene.cpp" in rebuilt and "cube.cpp" in rebuilt and "bin/cube" in rebuilt
        assert "png_io.cpp" not in rebuilt
    finally:
        os.utime(hdr, (st.st_atime, st.st_mtime))


def test_unknown_target_fails(tmp_path):
    r = _maek(["bin/viewer"], tmp_path / "out")
    assert r.returncode != 0 and "no rule to make" in r.stderr
