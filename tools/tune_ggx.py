"""Times the GGX level kernel for several builds of the library (extra builds of
csrc/ under tools/variants/), interleaved in one process.

    python tools/tune_ggx.py [in_w] [min_size]
"""
import ctypes
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import vulkan_renderer_amd as vra  # noqa: E402
from vulkan_renderer_amd import capi  # noqa: E402


def load(path):
    lib = ctypes.CDLL(path)
    for name, (argtypes, restype) in capi._SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, restype
    assert lib.ibl_init(0) == 0, lib.ibl_last_error()
    lib.ibl_set_timing(1)
    return lib


def main():
    in_w = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    min_size = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    ibl = vra.Ibl(0)
    strip = vra.synth.hdr_environment(in_w, seed=1234)
    env = ibl.rgbe_decode(torch.from_numpy(strip).cuda()).reshape(6, in_w, in_w, 4).contiguous()
    mips = ibl.ggx_mip_levels(in_w, in_w, min_size)
    libs = {"main": load(capi.LIB_PATH)}
    for p in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "*.so"))):
        libs[os.path.basename(p)] = load(p)
    outs = {m: torch.zeros((6, in_w >> m, in_w >> m, 4), device="cuda") for m in range(1, mips + 1)}
    res = {}
    for rnd in range(3):
        for name, lib in libs.items():
            for m in range(1, mips + 1):
                ow = in_w >> m
                rc = lib.ibl_ggx_level_dev(env.data_ptr(), in_w, in_w, ow, ow, m, mips, 0, 6 * ow * ow,
                                           outs[m].data_ptr(), None)
                assert rc == 0, lib.ibl_last_error()
                ms = ctypes.c_float()
                assert lib.ibl_last_kernel_ms(ctypes.byref(ms)) == 0
                res.setdefault((name, m), []).append(ms.value)
    for name in libs:
        row = [min(res[(name, m)]) for m in range(1, mips + 1)]
        print("%-18s " % name + " ".join("L%d %8.3f" % (m + 1, v) for m, v in enumerate(row)) + "  total %9.3f ms" % sum(row))


if __name__ == "__main__":
    main()
