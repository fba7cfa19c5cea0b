"""Times of the streaming kernels around the integrals (RGBE codec, PNG scanline filter) on cube strips.

    python tools/stream_kernels_time.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import vulkan_renderer_amd as vra  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ibl = vra.Ibl(0)
    for w in (256, 1024, 2048):
        n = 6 * w * w
        rgbe = torch.randint(0, 256, (6 * w, w, 4), dtype=torch.uint8, device="cuda")
        flt = ibl.rgbe_decode(rgbe.reshape(-1, 4))
        for name, fn, nbytes in (
                ("rgbe_decode", lambda: ibl.rgbe_decode(rgbe.reshape(-1, 4)), n * 20),
                ("rgbe_encode", lambda: ibl.rgbe_encode(flt), n * 20),
                ("png_filter_rows", lambda: ibl.png_filter_rows(rgbe), n * 8 + 6 * w)):
            ms = timed(fn)
            print("%-16s %4d/face: %.3f ms  %.0f GB/s" % (name, w, ms, nbytes / ms / 1e6))


if __name__ == "__main__":
    main()
