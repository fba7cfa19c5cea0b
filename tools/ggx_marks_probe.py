"""What the marked-patch launch of the GGX kernels walks and costs: per level of a chain, the wave-records
and the kernel time with the marking on (ggx_few 16), off (0) and with every patch marked.

    python tools/ggx_marks_probe.py [in_w]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import vulkan_renderer_amd as vra  # noqa: E402


def main():
    in_w = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    ibl = vra.Ibl(0)
    strip = vra.synth.hdr_environment(in_w, seed=1234)
    env = ibl.rgbe_decode(torch.from_numpy(strip).cuda()).reshape(6, in_w, in_w, 4).contiguous()
    mips = ibl.ggx_mip_levels(in_w, in_w, 16)
    ibl.set_timing(True)
    for m in range(1, mips + 1):
        ow = in_w >> m
        row = []
        for few in (0, 16, 1e9):
            ibl.set_tuning("ggx_few", few)
            ibl.set_counting(True)
            ibl.ggx_level(env, ow, m, mips)
            c = ibl.last_counts()
            ibl.set_counting(False)
            best = 1e9
            for _ in range(3):
                ibl.ggx_level(env, ow, m, mips)
                best = min(best, ibl.last_kernel_ms())
            row.append("few=%g: %.4g wave-records %.3f ms" % (few, c["wave_records"], best))
        print("level %d out %d: %s" % (m, ow, " | ".join(row)))
    ibl.set_tuning("ggx_few", 16)


if __name__ == "__main__":
    main()
