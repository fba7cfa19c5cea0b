import sys, torch
sys.path.insert(0, '/root/repo')
import vulkan_renderer_amd as vra
ibl = vra.Ibl(0)
for w in (512, 2048):
    env = torch.rand((6, w, w, 4), device='cuda')
    for d in (2, 4, 8, 16, 32, 64):
        for _ in range(2): ibl.downscale(env, d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ibl.downscale(env, d)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("w %d d %d: %.3f ms  (%.1f GB/s read)" % (w, d, ms, 6 * w * w * 16 / ms / 1e6))
