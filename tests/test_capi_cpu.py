"""CPU-side checks of the C-ABI library: it loads, exports everything the header
declares, does its host-only arithmetic, and refuses to run without a GPU."""
import ctypes
import os

import numpy as np
import pytest

import vulkan_renderer_amd as vra


def test_library_exports_every_declared_symbol():
    lib = vra.load_library()
    names = vra.exported_symbols()
    assert len(names) >= 23, names
    for name in names:
        assert hasattr(lib, name), "libibl_hip.so does not export %s" % name
    # and the binding table covers exactly the header
    from vulkan_renderer_amd import capi
    assert sorted(capi._SIGNATURES) == names


def test_level_geometry_matches_oracle(orc):
    lib = vra.load_library()
    for (w, m) in [(512, 16), (2048, 16), (64, 16), (64, 8), (1024, 128), (16, 16), (16, 32)]:
        want = orc.ggx_mip_levels(w, w, m) if w >= m else 0
        assert lib.ibl_ggx_mip_levels(w, w, m) == want, (w, m)
    for (mip, mips, w) in [(1, 5, 512), (3, 5, 512), (5, 5, 512), (1, 7, 2048), (2, 7, 2048), (2, 2, 64), (1, 1, 16)]:
        r, a2, c = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        ps = ctypes.c_uint32()
        assert lib.ibl_ggx_level_params(mip, mips, w, w, ctypes.byref(r), ctypes.byref(a2), ctypes.byref(c),
                                        ctypes.byref(ps)) == 0
        want = orc.ggx_params(mip, mips, w, w)
        assert (np.float32(r.value), np.float32(a2.value), np.float32(c.value), ps.value) == (
            np.float32(want["roughness"]), np.float32(want["alpha2"]), np.float32(want["cutoff"]), want["patch"])
    assert lib.ibl_ggx_level_params(0, 5, 64, 64, None, None, None, None) != 0
    assert b"out of range" in lib.ibl_last_error()


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path cannot be exercised")
    lib = vra.load_library()
    assert lib.ibl_device_count() == 0
    assert lib.ibl_init(0) == 2          # IBL_ENODEV
    assert b"no CPU path" in lib.ibl_last_error()
    buf = (ctypes.c_float * 4)()
    assert lib.ibl_lambertian(buf, 1, 1, 1, buf) == 2
    with pytest.raises(vra.IblError):
        vra.Ibl(0)


def test_header_is_plain_c(tmp_path):
    """include/ibl_hip.h is the C ABI: it must compile as strict C99 and as C++ with nothing but <stdint.h>."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "use.c"
    src.write_text('#include "ibl_hip.h"\nint main(void) { return IBL_OK; }\n')
    for cmd in (["gcc", "-std=c99", "-pedantic"], ["g++", "-std=c++20", "-x", "c++"]):
        r = subprocess.run(cmd + ["-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), "-fsyntax-only", str(src)],
                           capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr


def test_ggx_level_plans_stay_within_four_gib():
    """The launch plan of every level of the chains the benchmarks use, and of inputs up to 8192
    texels per face edge: the partial sums never need more than 4 GiB, the plan is found (a
    2 GB-per-run level once sent the search into an endless loop), and large levels spread their
    runs over several launches instead of taking fewer (the largest add every launch into one set of FP64 sums)."""
    from vulkan_renderer_amd import capi
    for in_w in (64, 512, 2048, 4096, 8192):
        mips = capi.load_library().ibl_ggx_mip_levels(in_w, in_w, 16)
        for m in range(1, mips + 1):
            p = capi.ggx_level_plan(in_w, in_w >> m, m, mips)
            assert 1 <= p["runs"] < 32768 and p["passes"] >= 1 and p["runs"] % p["passes"] == 0
            assert p["sums_bytes"] <= 4 << 30, (in_w, m, p)
    p = capi.ggx_level_plan(2048, 1024, 1, 7)
    assert p["passes"] > 1 and p["runs"] >= 80                # not the 48 runs that fit one launch
    p = capi.ggx_level_plan(8192, 4096, 1, 9)
    assert p["passes"] == 32 and p["runs"] == 32              # 1.3 GB per run: one at a time, added into one FP64 set
    with pytest.raises(capi.IblError):
        capi.ggx_level_plan(100, 50, 1, 2)                    # the general kernel's sizes
