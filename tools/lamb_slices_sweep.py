"""Lambertian cube 1024 -> 128 for several slice counts: kernel time and whole-call time of the slowest
of the first, middle and last shard at 1, 2, 4 and 8 shards (medians).

    python tools/lamb_slices_sweep.py [max slice counts...]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vulkan_renderer_amd as vra
ibl = vra.Ibl(0)
ibl.set_timing(True)
strip = vra.synth.hdr_environment(1024, seed=1234)
env = ibl.rgbe_decode(torch.from_numpy(strip).cuda()).reshape(6, 1024, 1024, 4).contiguous()
out = torch.zeros((6, 128, 128, 4), device="cuda")
def whole(fn, reps=15):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
for slices in [int(v) for v in (sys.argv[1:] or '341 400 512 683'.split())]:
    ibl.set_tuning("lamb_oct_slices", slices)
    line = "slices<=%4d " % slices
    for n in (1, 2, 4, 8):
        worst_k, worst_t = 0.0, 0.0
        for sh in sorted({0, n // 2, n - 1}):
            t = whole(lambda: ibl.lambertian_shard(env, 128, sh, n, out=out))
            k = ibl.last_kernel_ms()
            worst_k, worst_t = max(worst_k, k), max(worst_t, t)
        line += " N=%d kernel %.3f call %.3f |" % (n, worst_k, worst_t)
    print(line, flush=True)
