"""Narrowest GGX level of a very large input (default 8192 x 8192) on 8 output texels against the CPU
oracle, for the forms of the octet kernel and several widths of the exact core (the oracle is the checker here).

    python tests/probes/ggx_huge_probe.py [in_w]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import oracle  # noqa: E402
import vulkan_renderer_amd as vra  # noqa: E402


def main():
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    rep = w // 512
    small = vra.synth.hdr_environment(512, seed=9).reshape(6, 512, 512, 4)
    strip = np.repeat(np.repeat(small, rep, axis=1), rep, axis=2)
    yy, xx = np.meshgrid(np.arange(w), np.arange(w), indexing="ij")
    strip[..., 0] ^= ((xx + 3 * yy) & 3).astype(np.uint8)[None]
    del yy, xx
    env = oracle.rgbe_decode(strip.reshape(-1, 4)).reshape(6, w, w, 4)
    ibl = vra.Ibl(0)
    env_d = ibl.rgbe_decode(torch.from_numpy(strip.reshape(-1, 4)).cuda()).reshape(6, w, w, 4)
    mips = ibl.ggx_mip_levels(w, w, 16)
    half = w // 2
    begin = 6 * half * half // 2 + half * 77 + 1000
    want = oracle.ggx_level(env, half, half, 1, mips, out_begin=begin, out_count=8).reshape(-1, 4)[begin:begin + 8, :3]
    want_px = want
    for name, matrix, rsq, core in (("matrix rsq", 1, 2, 2), ("matrix sqrt+rcp", 1, 0, 2), ("vector", 0, 0, 2),
                                    ("matrix rsq, core 4", 1, 2, 4), ("matrix rsq, core 8", 1, 2, 8),
                                    ("matrix rsq, core 16", 1, 2, 16), ("vector, core 8", 0, 0, 8)):
        ibl.set_tuning("ggx_matrix", matrix)
        ibl.set_tuning("ggx_rsq", rsq)
        ibl.set_tuning("ggx_core", core)
        runs = vra.capi.ggx_level_plan(w, half, 1, mips)["runs"]
        got = ibl.ggx_level(env_d, half, 1, mips, out_begin=begin, out_count=8).cpu().numpy().reshape(-1, 4)[begin:begin + 8, :3]
        err = (got.astype(np.float64) - want) / want
        ibl.set_timing(True)
        print("%-28s runs %s  rel err mean %+.3e  max |.| %.3e" % (name, runs, err.mean(), np.abs(err).max()), flush=True)
    # ... and for several run counts (the fewer the runs, the longer a wave's FP32 running sums)
    ibl.set_tuning("ggx_matrix", 1)
    ibl.set_tuning("ggx_rsq", 1)
    ibl.set_tuning("ggx_core", 2)
    units = (3 * (half // 2) ** 2 + 63) // 64
    for want in (1, 3, 8, 24, 96):
        ibl.set_tuning("ggx_waves", float(want * units))
        plan = vra.capi.ggx_level_plan(w, half, 1, mips)
        got = ibl.ggx_level(env_d, half, 1, mips, out_begin=begin, out_count=8).cpu().numpy().reshape(-1, 4)[begin:begin + 8, :3]
        err = (got.astype(np.float64) - want_px) / want_px
        print("asked for %3d runs: plan %s  rel err mean %+.3e  max |.| %.3e" % (want, plan, err.mean(), np.abs(err).max()), flush=True)
    ibl.set_tuning("ggx_waves", 0)


if __name__ == "__main__":
    main()
