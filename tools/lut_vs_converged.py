"""Statistics of BRDF-table runs against the converged quadrature fixture
(tests/golden/ggx_lut_converged.npy): mean and RMS of the difference per channel and
for a few roughness rows, for several seeds / strata counts.

    python tools/lut_vs_converged.py [samples_sqrt ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vulkan_renderer_amd as vra  # noqa: E402


def main():
    conv = np.load(os.path.join(ROOT, "tests", "golden", "ggx_lut_converged.npy")).astype(np.float64)
    ibl = vra.Ibl()
    for ss in [int(a) for a in sys.argv[1:]] or [32, 256]:
        for seed in (1, 2024):
            got = ibl.brdf_lut(256, 256, ss, seed=seed).cpu().numpy().astype(np.float64)
            d = got - conv
            print("strata %d^2 seed %d: mean %.3g %.3g  rms %.3g %.3g  max %.3g  within 1e-4 %.3f" % (
                ss, seed, d[..., 0].mean(), d[..., 1].mean(), np.sqrt((d[..., 0] ** 2).mean()),
                np.sqrt((d[..., 1] ** 2).mean()), np.abs(d).max(), (np.abs(d) < 1e-4).all(-1).mean()))
            print("   row rms:", {r: float("%.3g" % np.sqrt((d[r, :, 0] ** 2).mean())) for r in (0, 16, 33, 64, 128, 192, 224, 255)})
            print("   row max:", {r: float("%.3g" % np.abs(d[r]).max()) for r in (0, 16, 33, 64, 128, 192, 224, 255)})
    for ss in (1024, 2048):
        got = ibl.brdf_lut(256, 256, ss, mode=vra.capi.LUT_JITTER_MIDPOINT).cpu().numpy().astype(np.float64)
        d = got - conv
        print("midpoint %d^2: mean %.3g %.3g rms %.3g %.3g max %.3g" % (ss, d[..., 0].mean(), d[..., 1].mean(),
              np.sqrt((d[..., 0] ** 2).mean()), np.sqrt((d[..., 1] ** 2).mean()), np.abs(d).max()))


if __name__ == "__main__":
    main()
