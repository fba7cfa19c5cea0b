import os, sys
sys.path.insert(0, os.getcwd())
import torch
import vulkan_renderer_amd as vra
ibl = vra.Ibl(0); ibl.set_timing(True)
for in_w, mips in ((512, 5), (1024, 6), (2048, 7)):
    strip = vra.synth.hdr_environment(in_w, seed=1234)
    env = ibl.rgbe_decode(torch.from_numpy(strip).cuda()).reshape(6, in_w, in_w, 4).contiguous()
    for core in (2.0, 0.0):
        ibl.set_tuning("ggx_core", core)
        ts = []
        for _ in range(3):
            ibl.ggx_level(env, in_w >> 1, 1, mips); ts.append(ibl.last_kernel_ms())
        print(in_w, "level 1 core multiplier", core, "kernel ms", min(ts), flush=True)
