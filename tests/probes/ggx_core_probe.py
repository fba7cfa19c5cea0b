import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
import oracle
import vulkan_renderer_amd as vra
from helpers import decode_strip
ibl = vra.Ibl(0); ibl.set_timing(True)
for in_w, mips in ((512, 5), (128, 3), (256, 4)):
    env = decode_strip(oracle, vra.synth.hdr_environment(in_w, seed=5))
    env_d = torch.from_numpy(env).cuda()
    ow = in_w >> 1
    n_out = 6 * ow * ow
    count = 512 if in_w == 512 else n_out
    begin = (n_out - count) // 2
    ts = []
    for _ in range(3):
        got = ibl.ggx_level(env_d, ow, 1, mips); ts.append(ibl.last_kernel_ms())
    want = oracle.ggx_level(env, ow, ow, 1, mips, out_begin=begin, out_count=count).reshape(-1, 4)[begin:begin + count, :3]
    g = got.cpu().numpy().reshape(-1, 4)[begin:begin + count, :3].astype(np.float64)
    err = (g - want) / want
    print("%d level 1: %.3f ms, vs oracle on %d texels: mean %+.3e, mean |.| %.3e, max |.| %.3e, bits %016x" % (
        in_w, min(ts), count, err.mean(), np.abs(err).mean(), np.abs(err).max(),
        int(got.view(torch.int32).to(torch.int64).sum().item()) & 0xffffffffffffffff), flush=True)
