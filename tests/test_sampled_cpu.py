"""CPU tests of the sampled GGX mode's own CPU statement (oracle/ibl_sampled_oracle.c).
The mode has no counterpart in the reference; what anchors it is the reference's
quadrature (oracle.ggx_level), which it must approach as the sample count grows."""
import numpy as np
import pytest

import vulkan_renderer_amd as vra
from helpers import decode_strip


def test_pyramid_layout(orc):
    assert [orc.pyramid_levels(w) for w in (1, 2, 30, 48, 64)] == [1, 2, 2, 5, 7]
    assert orc.pyramid_offset(64, 0) == 0
    assert orc.pyramid_offset(64, 2) == 6 * (64 * 64 + 32 * 32)
    env = decode_strip(orc, vra.synth.hdr_environment(48, seed=3))
    pyr = orc.env_pyramid(env).reshape(-1, 4)
    assert len(pyr) == orc.pyramid_offset(48, 5) == 6 * (48 ** 2 + 24 ** 2 + 12 ** 2 + 6 ** 2 + 3 ** 2)
    np.testing.assert_array_equal(pyr[:6 * 48 * 48], env.reshape(-1, 4))
    lvl1 = pyr[orc.pyramid_offset(48, 1):orc.pyramid_offset(48, 2)]
    np.testing.assert_array_equal(lvl1, orc.downscale(env, 2).reshape(-1, 4))
    # every level keeps the mean radiance
    top = pyr[orc.pyramid_offset(48, 4):]
    np.testing.assert_allclose(top[:, :3].mean(axis=0), env[..., :3].reshape(-1, 3).mean(axis=0), rtol=1e-5)


def test_sampled_constant_environment_is_exact(orc):
    env = decode_strip(orc, vra.synth.constant_environment(32, (1.0, 2.0, 0.5)))
    pyr = orc.env_pyramid(env)
    got = orc.ggx_level_sampled(pyr, 32, 16, 1, 1, 64)
    want = env[0, 0, 0, :3]
    assert np.abs(got[..., :3] / want - 1.0).max() < 1e-6
    assert (got[..., 3] == 0).all()


def test_sampled_approaches_the_quadrature(orc):
    """On a 64-texel HDR environment (4000:1 hot spot) the difference to the reference's
    quadrature falls with the sample count; with 16384 samples and no bias it is a
    fraction of a per cent on average."""
    w = 64
    env = decode_strip(orc, vra.synth.hdr_environment(w, seed=1234))
    pyr = orc.env_pyramid(env)
    mips = orc.ggx_mip_levels(w, w, 16)
    for m, bound in ((1, 2e-3), (2, 5e-3)):
        ow = w >> m
        ref = orc.ggx_level(env, ow, ow, m, mips)[..., :3]
        errs = []
        for n in (256, 4096, 16384):
            got = orc.ggx_level_sampled(pyr, w, ow, m, mips, n, 0.0)[..., :3]
            errs.append(float((np.abs(got - ref) / ref).mean()))
        assert errs[0] > errs[1] > errs[2], errs
        assert errs[2] < bound, errs


def test_sampled_ranges_are_independent(orc):
    env = decode_strip(orc, vra.synth.hdr_environment(32, seed=9))
    pyr = orc.env_pyramid(env)
    full = orc.ggx_level_sampled(pyr, 32, 16, 1, 1, 128)
    part = orc.ggx_level_sampled(pyr, 32, 16, 1, 1, 128, out_begin=100, out_count=300)
    np.testing.assert_array_equal(part.reshape(-1, 4)[100:400], full.reshape(-1, 4)[100:400])
    assert (part.reshape(-1, 4)[:100] == 0).all() and (part.reshape(-1, 4)[400:] == 0).all()


def test_sampled_rejects_bad_arguments(orc):
    env = decode_strip(orc, vra.synth.constant_environment(16))
    pyr = orc.env_pyramid(env)
    with pytest.raises(RuntimeError):
        orc.ggx_level_sampled(pyr, 16, 8, 1, 1, 0)
    with pytest.raises(RuntimeError):
        orc.ggx_level_sampled(pyr, 16, 8, 1, 1, 16, out_begin=0, out_count=6 * 8 * 8 + 1)
