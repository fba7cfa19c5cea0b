"""Prints how the deterministic (midpoint) BRDF-table quadrature converges with the
number of strata per axis.  Run on the GPU box: python tools/lut_convergence.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_renderer_amd as vra  # noqa: E402


def main():
    ibl = vra.Ibl(0)
    prev = None
    for s in (256, 512, 1024, 2048, 4096, 8192):
        cur = ibl.brdf_lut(64, 64, s, mode=vra.capi.LUT_JITTER_MIDPOINT).cpu().numpy().astype(np.float64)
        if prev is not None:
            d = np.abs(cur - prev)
            i = np.unravel_index(d.argmax(), d.shape)
            print("strata %5d vs %5d: max %.3e at %s  rms %.3e  p99 %.3e" % (s, s // 2, d.max(), i, np.sqrt((d ** 2).mean()), np.quantile(d, 0.99)))
        prev = cur
    jit = ibl.brdf_lut(64, 64, 2048, seed=5).cpu().numpy().astype(np.float64)
    d = np.abs(jit - prev)
    print("jittered 2048 vs midpoint 8192: max %.3e rms %.3e" % (d.max(), np.sqrt((d ** 2).mean())))


if __name__ == "__main__":
    main()
