"""Mirror-local re-projection (src/main.cpp:523-604, shaders/mirrorLocal.vert/.frag) as a
per-texel computation, HIP against the CPU statement in oracle/ibl_mirror_oracle.c.

Parity unpinned: the reference runs this step through the raster pipeline and the hardware cube
sampler and holds no output of it; the two sides here share one stated arithmetic (no fused
multiply-adds, correctly rounded divisions and square roots), so they are compared bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ibl():
    import vulkan_renderer_amd as vra
    return vra.Ibl()


def _cube(w, seed):
    rng = np.random.default_rng(seed)
    c = rng.random((6, w, w, 4), dtype=np.float32) * 4.0
    c[..., 3] = 1.0
    return c


CASES = [
    (16, [0.0, 0.0, 0.0], [-50, -50, -50], [50, 50, 50]),          # far walls: almost the identity
    (16, [0.3, -0.2, 0.1], [-1, -1, -1], [1, 1, 1]),
    (64, [1.5, 0.25, -0.75], [-2, -1, -3], [2, 1.5, 1]),
    (32, [5.0, 0.0, 0.0], [-1, -1, -1], [1, 1, 1]),                # camera outside the box: most rays miss
    (128, [-0.9, 0.9, 0.9], [-1, -1, -1], [1, 1, 1]),               # camera in a corner
    (2, [0.1, 0.1, 0.1], [-1, -1, -1], [1, 1, 1]),
    (1, [0.0, 0.0, 0.0], [-1, -1, -1], [1, 1, 1]),
    (64, [0.0, 0.0, 0.995], [-1, -1, -1], [1, 1, 1]),              # the +z wall is inside the near plane
]


@pytest.mark.parametrize("w,cam,lo,hi", CASES)
def test_mirror_reproject_matches_cpu_statement(ibl, w, cam, lo, hi):
    import torch
    import oracle
    src = _cube(w, w)
    want = oracle.mirror_reproject(src, cam, lo, hi)
    got = ibl.mirror_reproject(torch.from_numpy(src).cuda(), cam, lo, hi).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_mirror_reproject_properties(ibl):
    import torch
    import oracle
    # a constant cube stays constant wherever the camera stands inside the box
    src = np.empty((6, 32, 32, 4), np.float32)
    src[...] = [0.25, 2.0, 7.5, 1.0]
    got = ibl.mirror_reproject(torch.from_numpy(src).cuda(), [0.4, -0.3, 0.2], [-1, -1, -1], [1, 1, 1]).cpu().numpy()
    assert np.abs(got - src).max() <= 1e-6
    # a camera at the centre of a cube-shaped box looks every texel up in its own direction
    src = _cube(64, 3)
    got = ibl.mirror_reproject(torch.from_numpy(src).cuda(), [0, 0, 0], [-1, -1, -1], [1, 1, 1]).cpu().numpy()
    assert np.abs(got - src).max() <= 2e-5
    # the reflected camera, mirrorLocal.vert:81-82
    t = np.eye(4, dtype=np.float32)
    t[:3, :3] = [[0, 1, 0], [-1, 0, 0], [0, 0, 1]]
    t[3, :3] = [0.5, -1.0, 2.0]                                   # column-major: the last column is the translation
    a = ibl.mirror_reflect_camera(t, [1.0, 2.0, 3.0], 0.75)
    b = oracle.mirror_reflect_camera(t, [1.0, 2.0, 3.0], 0.75)
    assert np.array_equal(a, b) and a[2] == -(3.0 + 2.0 + 0.75) - 0.75


def test_mirror_reproject_chain_level_sizes(ibl):
    """Every level of a 512 chain, as the caller's loop over mips runs it (main.cpp:533)."""
    import torch
    import oracle
    for w in (256, 128, 64, 32, 16):
        src = _cube(w, 100 + w)
        want = oracle.mirror_reproject(src, [0.2, 0.1, -1.7], [-3, -1, -2], [3, 2, 2])
        got = ibl.mirror_reproject(torch.from_numpy(src).cuda(), [0.2, 0.1, -1.7], [-3, -1, -2], [3, 2, 2]).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_mirror_reproject_rejects_bad_arguments(ibl):
    import torch
    src = torch.zeros(6, 8, 8, 4, device="cuda")
    with pytest.raises(Exception):
        ibl.mirror_reproject(src, [0, 0, 0], [1, -1, -1], [-1, 1, 1])
    with pytest.raises(Exception):
        ibl.mirror_reproject(src, [0, 0, 0], [-1, -1, -1], [1, 1, 1], out=src)
    with pytest.raises(Exception):
        ibl.mirror_reproject(src, [float("nan"), 0, 0], [-1, -1, -1], [1, 1, 1])
