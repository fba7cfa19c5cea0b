"""The forms of the GGX octet kernel side by side: time per level and how far the texels differ
from the FP32 vector form.

    python tools/ggx_forms_check.py [in_w] [min_size] [first_level] [last_level]

Forms: vector (ggx_matrix 0), matrix with v_sqrt_f32 + v_rcp_f32 weights everywhere (ggx_rsq 0),
matrix with the v_rsq_f32 weights at narrow lobes (the default).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import vulkan_renderer_amd as vra  # noqa: E402

FORMS = (("vector", 0, 0), ("matrix sqrt+rcp", 1, 0), ("matrix rsq", 1, 1))


def main():
    in_w = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    min_size = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    ibl = vra.Ibl(0)
    strip = vra.synth.hdr_environment(in_w, seed=1234)
    env = ibl.rgbe_decode(torch.from_numpy(strip).cuda()).reshape(6, in_w, in_w, 4).contiguous()
    mips = ibl.ggx_mip_levels(in_w, in_w, min_size)
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    last = int(sys.argv[4]) if len(sys.argv) > 4 else mips
    skip = os.environ.get("SKIP", "").split(",")
    forms = [f for f in FORMS if f[0].split()[0] not in skip]
    ibl.set_timing(True)
    totals = {f[0]: 0.0 for f in forms}
    for m in range(first, last + 1):
        ow = in_w >> m
        ref = None
        for name, matrix, rsq in forms:
            ibl.set_tuning("ggx_matrix", matrix)
            ibl.set_tuning("ggx_rsq", rsq)
            best = 1e9
            for _ in range(2):
                out = ibl.ggx_level(env, ow, m, mips)
                best = min(best, ibl.last_kernel_ms())
            totals[name] += best
            line = "level %d out %4d  %-16s %9.3f ms  bits %016x" % (
                m, ow, name, best, int(out.view(torch.int32).to(torch.int64).sum().item()) & 0xffffffffffffffff)
            if ref is None:
                ref = out.clone()
            else:
                a, b = out[..., :3].double(), ref[..., :3].double()
                rel = (a - b).abs() / b.abs().clamp_min(1e-30)
                enc_a = ibl.rgbe_encode(out.reshape(-1, 4).contiguous())
                enc_b = ibl.rgbe_encode(ref.reshape(-1, 4).contiguous())
                same = (enc_a == enc_b).all(dim=-1).double().mean().item()
                line += "   vs %s: max rel %.3e  mean rel %.3e  RGBE texels identical %.5f" % (
                    forms[0][0], rel.max().item(), rel.mean().item(), same)
            print(line, flush=True)
    for name, _, _ in forms:
        print("%-16s levels %d-%d kernel total %.3f ms" % (name, first, last, totals[name]))


if __name__ == "__main__":
    main()
