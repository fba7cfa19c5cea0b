import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vulkan_renderer_amd as vra
few = float(sys.argv[1])
ibl = vra.Ibl(0)
strip = vra.synth.hdr_environment(512, seed=1234)
env = ibl.rgbe_decode(torch.from_numpy(strip).cuda()).reshape(6, 512, 512, 4).contiguous()
ibl.set_tuning("ggx_few", few)
for _ in range(3):
    ibl.ggx_level(env, 256, 1, 5)
torch.cuda.synchronize()
