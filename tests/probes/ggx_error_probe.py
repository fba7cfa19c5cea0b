"""GGX level error against the CPU oracle on a small range of output texels, for
several settings of the exact-core threshold multiplier (env IBL_GGX_CORE).

    python tests/probes/ggx_error_probe.py in_w mips level [count] [mult ...]
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
from helpers import decode_strip  # noqa: E402


def main():
    in_w, mips, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    count = int(sys.argv[4]) if len(sys.argv) > 4 else 192
    mults = sys.argv[5:] or ["0", "2", "8"]
    ow = in_w >> m
    n_out = 6 * ow * ow
    begin = int(os.environ.get("PROBE_BEGIN", (n_out - count) // 2 + 3))
    env = decode_strip(oracle, vra.synth.hdr_environment(in_w, seed=5))
    mask = os.environ.get("PROBE_MASK")
    if mask:
        lo, hi = (float(v) for v in mask.split(","))
        peak = env[..., :3].max(axis=-1, keepdims=True)
        env = np.where((peak >= lo) & (peak < hi), env, 0.0).astype(np.float32)
    ref, ref_counts = oracle.ggx_level(env, ow, ow, m, mips, out_begin=begin, out_count=count, want_counts=True)
    ref = ref.reshape(-1, 4)[begin:begin + count]
    print("oracle counts", ref_counts)
    ref_b = oracle.rgbe_encode(ref)
    if os.environ.get("PROBE_LIB"):
        vra.capi.LIB_PATH = os.path.abspath(os.environ["PROBE_LIB"])
    ibl = vra.Ibl(0)
    d_env = torch.from_numpy(env).cuda()
    ibl.set_timing(True)
    for mult in mults:
        os.environ["IBL_GGX_CORE"] = mult
        got = ibl.ggx_level(d_env, ow, m, mips, out_begin=begin, out_count=count).cpu().numpy().reshape(-1, 4)[begin:begin + count]
        ibl.set_counting(True)
        ibl.ggx_level(d_env, ow, m, mips, out_begin=begin, out_count=count)
        print("    gpu counts", ibl.last_counts())
        ibl.set_counting(False)
        ibl.ggx_level(d_env, ow, m, mips)
        ms = ibl.last_kernel_ms()
        err = np.abs(got[:, :3].astype(np.float64) - ref[:, :3]) / np.abs(ref[:, :3])
        same = (oracle.rgbe_encode(got) == ref_b).all(axis=-1).mean()
        print("in %d mips %d level %d core %-3s  mean %.2e max %.2e  rgbe identical %.4f  full level %.3f ms"
              % (in_w, mips, m, mult, err.mean(), err.max(), same, ms), flush=True)
        print("    abs diff of first texel", got[0, :3].astype(np.float64) - ref[0, :3], "ref", ref[0, :3])
        worst = np.argsort(err.max(axis=1))[-4:][::-1]
        for i in worst:
            print("    texel %d  err %s  ref %s  got %s" % (begin + i, np.array2string(err[i], precision=2),
                                                            np.array2string(ref[i, :3], precision=9),
                                                            np.array2string(got[i, :3], precision=9)))


if __name__ == "__main__":
    main()
