"""Kernel time of one shard of the Lambertian cube (ls_pairs) for several slice counts and grid
orders.

    python tools/tune_lamb_shard.py [n_shards]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import vulkan_renderer_amd as vra  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ibl = vra.Ibl(0)
    ibl.set_timing(True)
    strip = vra.synth.hdr_environment(1024, seed=1234)
    env = ibl.rgbe_decode(torch.from_numpy(strip).cuda()).reshape(6, 1024, 1024, 4).contiguous()
    out = torch.zeros((6, 128, 128, 4), device="cuda")
    for slices in (256, 304, 320, 341, 352, 400, 512, 683):
        for affine in (0, 1):
            ibl.set_tuning("lamb_oct_slices", slices)
            ibl.set_tuning("lamb_oct_affine", affine)
            ms = []
            for _ in range(5):
                ibl.lambertian_shard(env, 128, 3, n, out=out)
                ms.append(ibl.last_kernel_ms())
            print("shards %d  slices<=%4d affine=%d  kernel ms min %.4f med %.4f" % (n, slices, affine, min(ms), sorted(ms)[2]), flush=True)


if __name__ == "__main__":
    main()
