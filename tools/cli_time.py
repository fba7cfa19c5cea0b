"""Wall time of the cube tool end to end (PNG in, PNGs out) on a synthetic strip.

    python tools/cli_time.py [face_width] [extra cube arguments...]
"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vulkan_renderer_amd as vra  # noqa: E402
from helpers import write_png  # noqa: E402


def main():
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    extra = sys.argv[2:]
    cube = os.path.join(ROOT, "vulkan-renderer_amd", "bin", "cube")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "sky.png")
        write_png(src, vra.synth.hdr_environment(w, seed=1234))
        for _ in range(2):      # second run: warm page cache and GPU driver
            t0 = time.time()
            r = subprocess.run([cube, "--lambertian", "--ggx", "--cubemap-size", "16"] + extra + [src],
                               capture_output=True, text=True, env=dict(os.environ, IBL_CUBE_TIMING="1"))
            dt = time.time() - t0
            assert r.returncode == 0, r.stderr
        print(r.stderr.strip())
        files = sorted(f for f in os.listdir(d) if f != "sky.png")
        print("cube %s on a %dx%d/face strip: %.2f s wall, wrote %d files (%.1f MB)"
              % (" ".join(extra) or "(quadrature)", w, w, dt, len(files),
                 sum(os.path.getsize(os.path.join(d, f)) for f in files) / 1e6))


if __name__ == "__main__":
    main()
