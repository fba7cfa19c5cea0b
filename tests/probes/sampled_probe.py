"""Sampled GGX mode: GPU against its CPU statement and against the quadrature, with times.

    python tests/probes/sampled_probe.py [in_w] [n_samples]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import oracle  # noqa: E402
import vulkan_renderer_amd as vra  # noqa: E402
from helpers import decode_strip  # noqa: E402


def main():
    in_w = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    ibl = vra.Ibl(0)
    ibl.set_timing(True)
    env = decode_strip(oracle, vra.synth.hdr_environment(in_w, seed=1234))
    d_env = torch.from_numpy(env).cuda()
    pyr = ibl.env_pyramid(d_env)
    want_pyr = oracle.env_pyramid(env)
    padded = oracle.env_pyramid_padded(env)
    print("pyramid bit-exact:", bool((pyr.cpu().numpy().reshape(-1).view(np.uint32) == padded.view(np.uint32)).all()))
    mips = ibl.ggx_mip_levels(in_w, in_w, 16)
    for m in range(1, mips + 1):
        ow = in_w >> m
        n_out = 6 * ow * ow
        cnt = min(n_out, 1536)
        beg = (n_out - cnt) // 2
        got_full = ibl.ggx_level_sampled(pyr, in_w, ow, m, mips, n, 1.0)
        ms = ibl.last_kernel_ms()
        got = got_full.cpu().numpy().reshape(-1, 4)[beg:beg + cnt]
        want = oracle.ggx_level_sampled(want_pyr, in_w, ow, m, mips, n, 1.0, beg, cnt).reshape(-1, 4)[beg:beg + cnt]
        err = np.abs(got[:, :3] - want[:, :3]) / np.abs(want[:, :3])
        quad = oracle.ggx_level(env, ow, ow, m, mips, out_begin=beg, out_count=cnt).reshape(-1, 4)[beg:beg + cnt]
        qerr = np.abs(got[:, :3] - quad[:, :3]) / np.abs(quad[:, :3])
        print("level %d/%d out %4d N=%d: %.3f ms (%.1f Gsamples/s) | vs CPU statement mean %.2e max %.2e | vs quadrature mean %.2e p95 %.2e"
              % (m, mips, ow, n, ms, n_out * n / ms / 1e6, err.mean(), err.max(), qerr.mean(), np.percentile(qerr, 95)), flush=True)


if __name__ == "__main__":
    main()
