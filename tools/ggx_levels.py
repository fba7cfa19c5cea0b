"""Per-level kernel times and work counts of a GGX chain, for both kernel families.

    python tools/ggx_levels.py [in_w] [min_size] [downscale]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import vulkan_renderer_amd as vra  # noqa: E402


def main():
    in_w = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    min_size = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    down = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    ibl = vra.Ibl(0)
    strip = vra.synth.hdr_environment(in_w, seed=1234)
    env = ibl.rgbe_decode(torch.from_numpy(strip).cuda()).reshape(6, in_w, in_w, 4).contiguous()
    if down > 1:
        env = ibl.downscale(env, down)
    mips = ibl.ggx_mip_levels(in_w, in_w, min_size)
    ibl.set_timing(True)
    for mode in (("octets",) if os.environ.get("OCTETS_ONLY") else ("octets", "general")):
        ibl.set_kernel_mode(mode == "general")
        total = 0.0
        for m in range(1, mips + 1):
            ow = in_w >> m
            ibl.set_counting(True)
            ibl.ggx_level(env, ow, m, mips)
            c = ibl.last_counts()
            ibl.set_counting(False)
            best = 1e9
            for _ in range(2):
                ibl.ggx_level(env, ow, m, mips)
                best = min(best, ibl.last_kernel_ms())
            total += best
            p = ibl.ggx_level_params(m, mips, env.shape[2], env.shape[1])
            flop = 30.0 * c["accepted"] + 8.0 * (c["tested"] - c["accepted"])
            lanes = c["tested"] / (512.0 * c["wave_records"]) if c["wave_records"] else float("nan")
            print("%-8s level %d out %4d r=%.3f cutoff=%.3f tested %.3e accepted %.3e  %.3f ms  %.1f Gtested/s  %.1f TFLOP/s(alg)"
                  "  wave-records %.3e (lanes in use %.3f, %.1f ns each per SIMD, %.2f patterns per record visited)"
                  % (mode, m, ow, p["roughness"], p["cutoff"], c["tested"], c["accepted"], best,
                     c["tested"] / best / 1e6, flop / best / 1e9, c["wave_records"], lanes,
                     best * 1e6 * 1024 / max(c["wave_records"], 1), c["wave_records"] / max(c.get("wave_visits", 0), 1)))
        print("%-8s chain kernel total %.3f ms" % (mode, total))


if __name__ == "__main__":
    main()
