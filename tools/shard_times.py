"""Kernel time of one rank's shard for N = 1, 2, 4, 8 ranks, measured on one GPU
(every shard of a split is run in turn; the slowest one bounds the job).

    python tools/shard_times.py [c4|c3|c5]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import vulkan_renderer_amd as vra  # noqa: E402


def timed(fn, reps=3, burst=1):
    """Best of `reps` timings of `burst` calls in a row (short calls are timed in bursts, after two untimed ones:
    a lone millisecond call between two synchronisations runs at whatever clock the idle GPU had)."""
    best = 1e9
    for _ in range(reps):
        if burst > 1:
            fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(burst):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / burst)
    return best


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
    ibl = vra.Ibl(0)
    ibl.set_timing(True)
    in_w = {"c4": 1024, "c3": 512, "c5": 2048}[wl]
    strip = vra.synth.hdr_environment(in_w, seed=1234)
    env = ibl.rgbe_decode(torch.from_numpy(strip).cuda()).reshape(6, in_w, in_w, 4).contiguous()
    if wl == "c4":
        out = torch.zeros((6, 128, 128, 4), device="cuda")
        jobs = [("lambertian", lambda r, n: ibl.lambertian_shard(env, 128, r, n, out=out))]
    else:
        mips = ibl.ggx_mip_levels(in_w, in_w, 16)
        jobs = []
        for m in range(1, mips + 1):
            ow = in_w >> m
            o = torch.zeros((6, ow, ow, 4), device="cuda")
            jobs.append(("ggx level %d" % m, lambda r, n, ow=ow, m=m, o=o: ibl.ggx_level_shard(env, ow, m, mips, r, n, out=o)))
    only = os.environ.get("SHARD_LEVELS")          # e.g. "1,2": restrict a GGX chain to these levels
    if only and wl != "c4":
        jobs = [jobs[int(v) - 1] for v in only.split(",")]
    ns = tuple(int(v) for v in os.environ.get("SHARD_NS", "1,2,4,8").split(","))
    totals = {}
    for name, fn in jobs:
        base = None
        for n in ns:
            ranks = range(n) if (n <= 4 or os.environ.get("SHARD_ALL")) else (0, 3, 7)
            ts, ks = [], []
            for r in ranks:
                ts.append(timed(lambda: fn(r, n), reps=2 if wl == "c5" else 3, burst=8 if wl == "c4" else 1))
                try:
                    ks.append(ibl.last_kernel_ms())
                except vra.capi.IblError:      # this rank owns no block of the level
                    ks.append(0.0)
            worst = max(ts)
            base = worst * n if base is None else base
            totals[n] = totals.get(n, 0.0) + worst
            print("%-14s N=%d  slowest shard %9.3f ms  (ideal %9.3f)  efficiency %.3f   main kernel alone %9.3f ms"
                  % (name, n, worst, base / n, base / n / worst, max(ks)), flush=True)
            if os.environ.get("SHARD_ALL"):
                print("    per shard:", " ".join("%.2f" % t for t in ts), " sum %.2f" % sum(ts), flush=True)
    for n in ns:
        print("total N=%d  %9.3f ms  efficiency %.3f" % (n, totals[n], totals[ns[0]] * ns[0] / n / totals[n]))


if __name__ == "__main__":
    main()
