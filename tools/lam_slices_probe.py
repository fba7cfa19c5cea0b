import sys, os
sys.path.insert(0, '/root/repo')
import torch, vulkan_renderer_amd as vra
ibl = vra.Ibl(0); ibl.set_timing(True)
strip = vra.synth.hdr_environment(1024, seed=1234)
env = ibl.rgbe_decode(torch.from_numpy(strip).cuda()).reshape(6,1024,1024,4).contiguous()
out = torch.zeros((6,128,128,4), device='cuda')
for sl in (512, 384, 256, 128):
    ibl.set_tuning("lamb_oct_slices", sl)
    for n in (1, 2, 4, 8):
        ms=[]
        for _ in range(4):
            ibl.lambertian_shard(env, 128, 0, n, out=out); ms.append(ibl.last_kernel_ms())
        print("slices %d  shard 0 of %d: %.3f ms  (x%d = %.2f)" % (sl, n, min(ms), n, min(ms)*n), flush=True)
