"""Captures a small bake (downscale -> 5 GGX levels -> RGBE encode) in a HIP graph through
torch.cuda.graph and compares replay with eager execution: bits and time per bake.

    python tools/graph_probe.py [face_width] [input_downscale]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import vulkan_renderer_amd as vra  # noqa: E402


def main():
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    ibl = vra.Ibl(0)
    strip = torch.from_numpy(vra.synth.hdr_environment(w, seed=1234)).cuda()
    env_full = ibl.rgbe_decode(strip.reshape(-1, 4)).reshape(6, w, w, 4).contiguous()
    mips = ibl.ggx_mip_levels(w, w, 16)

    def bake():
        env = ibl.downscale(env_full, d)
        return [ibl.rgbe_encode(ibl.ggx_level(env, w >> m, m, mips).reshape(-1, 4)) for m in range(1, mips + 1)]

    for _ in range(3):          # warm up: the library's workspaces reach their final size and stop allocating
        want = bake()
    torch.cuda.synchronize()
    want = [e.clone() for e in want]

    def timed(fn, n=50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    eager_ms = timed(bake)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        enc = bake()                      # tensors of the graph's own pool: rewritten by every replay
    for e in enc:
        e.zero_()
    graph.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(enc, want))
    graph_ms = timed(graph.replay)
    print("bake of a %d/face environment at downscale %d, %d levels: eager %.3f ms, graph replay %.3f ms, identical %s"
          % (w, d, mips, eager_ms, graph_ms, same))


if __name__ == "__main__":
    main()
