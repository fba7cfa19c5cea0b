"""Regression pins of the CPU oracle on a seeded 16x16/face environment.

These are outputs of THIS repo's oracle (oracle/ibl_oracle.c), not of the
reference: the reference cannot be built in this image (its cube.cpp needs the
Vulkan and GLFW headers), so the pins guard the checker against silent edits and
nothing more.

    python tests/golden/make_oracle_pins.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import vulkan_renderer_amd as vra  # noqa: E402


def main():
    rgbe = vra.synth.hdr_environment(16, seed=1234)
    env = oracle.rgbe_decode(rgbe).reshape(6, 16, 16, 4)
    lam = oracle.lambertian(env, 8)
    ggx = oracle.ggx_level(env, 8, 8, 1, 1)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_pins_16.npz")
    np.savez_compressed(dst, rgbe=rgbe, lambertian8=lam, ggx8_r1=ggx, lambertian8_rgbe=oracle.rgbe_encode(lam))
    print("wrote", dst)


if __name__ == "__main__":
    main()
