"""Matrix form of the GGX octet kernel against the vector form: times per level and how far the texels differ.

    python tools/ggx_matrix_check.py [in_w] [min_size] [first_level] [last_level]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import vulkan_renderer_amd as vra  # noqa: E402


def main():
    in_w = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    min_size = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    ibl = vra.Ibl(0)
    strip = vra.synth.hdr_environment(in_w, seed=1234)
    env = ibl.rgbe_decode(torch.from_numpy(strip).cuda()).reshape(6, in_w, in_w, 4).contiguous()
    mips = ibl.ggx_mip_levels(in_w, in_w, min_size)
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    last = int(sys.argv[4]) if len(sys.argv) > 4 else mips
    modes = [int(v) for v in os.environ.get("MODES", "0,1,2").split(",")]
    ibl.set_timing(True)
    totals = {m: 0.0 for m in modes}
    for m in range(first, last + 1):
        ow = in_w >> m
        outs = {}
        for mode in modes:
            ibl.set_tuning("ggx_matrix", mode)
            best = 1e9
            for _ in range(2):
                out = ibl.ggx_level(env, ow, m, mips)
                best = min(best, ibl.last_kernel_ms())
            outs[mode] = out.clone()
            totals[mode] += best
            line = "level %d out %4d  ggx_matrix=%d  %9.3f ms" % (m, ow, mode, best)
            if mode != modes[0]:
                ref = outs[modes[0]]
                a, b = out[..., :3].double(), ref[..., :3].double()
                rel = (a - b).abs() / b.abs().clamp_min(1e-30)
                enc_a = ibl.rgbe_encode(out.reshape(-1, 4).contiguous())
                enc_b = ibl.rgbe_encode(ref.reshape(-1, 4).contiguous())
                same = (enc_a == enc_b).all(dim=-1).double().mean().item()
                line += "   vs mode %d: max rel %.3e  mean rel %.3e  RGBE texels identical %.5f" % (
                    modes[0], rel.max().item(), rel.mean().item(), same)
            print(line, flush=True)
    for mode in modes:
        print("ggx_matrix=%d  levels %d-%d kernel total %.3f ms" % (mode, first, last, totals[mode]))


if __name__ == "__main__":
    main()
