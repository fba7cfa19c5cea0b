"""Times the split Lambertian kernel (ls_pairs) for several builds of the library
(tools/variants/*.so, see the Makefile's EXTRA/BUILD/LIB variables) and several launch shapes,
interleaved in one process.

    python tools/tune_lamb_split.py [in_w out_w]
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
    in_w = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    out_w = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    ibl = vra.Ibl(0)
    strip = vra.synth.hdr_environment(in_w, seed=1234)
    env = ibl.rgbe_decode(torch.from_numpy(strip).cuda()).reshape(6, in_w, in_w, 4).contiguous()
    out = torch.zeros((6, out_w, out_w, 4), device="cuda")
    libs = {"main": load(capi.LIB_PATH)}
    for p in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "*.so"))):
        libs[os.path.basename(p)] = load(p)
    n_out = 6 * out_w * out_w
    results = {}
    for rnd in range(3):
        for name, lib in libs.items():
            for slices in (152, 304, 608, 1216):
                for affine in (0, 1):
                    lib.ibl_set_tuning(b"lamb_oct_slices", float(slices))
                    lib.ibl_set_tuning(b"lamb_oct_affine", float(affine))
                    rc = lib.ibl_lambertian_dev(env.data_ptr(), in_w, in_w, out_w, 0, n_out, out.data_ptr(), None)
                    assert rc == 0, lib.ibl_last_error()
                    ms = ctypes.c_float()
                    assert lib.ibl_last_kernel_ms(ctypes.byref(ms)) == 0
                    torch.cuda.synchronize()
                    results.setdefault((name, slices, affine), []).append(ms.value)
    for (name, slices, affine), v in sorted(results.items()):
        print("%-20s slices<=%4d affine=%d  kernel ms min %.3f med %.3f" % (name, slices, affine, min(v), sorted(v)[len(v) // 2]))


if __name__ == "__main__":
    main()
