"""Debug aid: matrix form against vector form on special environments (constant; one hot texel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import vulkan_renderer_amd as vra  # noqa: E402


def both(ibl, env, ow, m, mips):
    outs = []
    for mode in (0, 2):
        ibl.set_tuning("ggx_matrix", mode)
        outs.append(ibl.ggx_level(env, ow, m, mips)[..., :3].double().clone())
    return outs


def main():
    in_w = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    ibl = vra.Ibl(0)
    mips = ibl.ggx_mip_levels(in_w, in_w, 16)
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ow = in_w >> m
    env = torch.ones((6, in_w, in_w, 4), device="cuda")
    a, b = both(ibl, env, ow, m, mips)
    print("constant: vector min/max %.8f %.8f   matrix min/max %.8f %.8f" % (a.min(), a.max(), b.min(), b.max()))
    g = torch.Generator().manual_seed(5)
    env = (torch.rand((6, in_w, in_w, 4), generator=g) + 0.5).cuda()
    a, b = both(ibl, env, ow, m, mips)
    rel = ((a - b).abs() / a.abs()).max().item()
    print("uniform random [0.5,1.5]: max rel %.3e" % rel)
    for (f, y, x) in [(0, 0, 0), (0, 0, 1), (0, 0, 2), (0, 0, 3), (0, 1, 0), (0, 2, 0), (0, 3, 0), (0, 3, 3), (0, 5, 9), (2, 40, 17), (5, 63, 63)]:
        env = torch.full((6, in_w, in_w, 4), 1e-6, device="cuda")
        env[f, y, x, :3] = 1.0
        a, b = both(ibl, env, ow, m, mips)
        big = a > 1e-3 * a.max()
        rel = ((a - b).abs() / a.abs())[big]
        print("hot texel face %d y %2d x %2d: texels lit %6d  max rel %.3e  mean rel %.3e   sum vector %.6e matrix %.6e" % (
            f, y, x, int(big.sum()), rel.max().item(), rel.mean().item(), a.sum().item(), b.sum().item()))


if __name__ == "__main__":
    main()
