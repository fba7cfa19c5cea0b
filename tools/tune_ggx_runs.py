"""Times ggx_oct_pairs on chosen levels of a chain for several wave-count targets (the knob
that fixes the number of runs per level) and grid orders.

    python tools/tune_ggx_runs.py [in_w levels waves...]   e.g.  2048 1,2 0 65536 262144 2e6
    SHARD=r/n in the environment times shard r of n instead of the whole level.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import vulkan_renderer_amd as vra  # noqa: E402


def main():
    in_w = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    levels = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2").split(",")]
    waves = [float(x) for x in sys.argv[3:]] or [0.0]
    ibl = vra.Ibl(0)
    ibl.set_timing(True)
    strip = vra.synth.hdr_environment(in_w, seed=1234)
    env = ibl.rgbe_decode(torch.from_numpy(strip).cuda()).reshape(6, in_w, in_w, 4).contiguous()
    mips = ibl.ggx_mip_levels(in_w, in_w, 16)
    for m in levels:
        ow = in_w >> m
        out = torch.zeros((6, ow, ow, 4), device="cuda")
        for w in waves:
            for xcd in (-1, 0, 1):
                ibl.set_tuning("ggx_waves", w)
                ibl.set_tuning("ggx_xcd_slices", xcd)
                ms = []
                for _ in range(2):
                    if os.environ.get("SHARD"):
                        r, n = (int(v) for v in os.environ["SHARD"].split("/"))
                        ibl.ggx_level_shard(env, ow, m, mips, r, n, out=out)
                    else:
                        ibl.ggx_level(env, ow, m, mips, out=out)
                    ms.append(ibl.last_kernel_ms())
                print("level %d waves %-9g xcd %2d: %s ms" % (m, w, xcd, " ".join("%.2f" % x for x in ms)), flush=True)


if __name__ == "__main__":
    main()
