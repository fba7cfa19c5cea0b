"""ibl_lambertian_static_dev: the Lambertian entry point that allocates nothing (its
workspace is a file-scope device array of the library), against ibl_lambertian_dev."""
import numpy as np
import pytest

import vulkan_renderer_amd as vra
from helpers import decode_strip

pytestmark = pytest.mark.gpu


def _env(ibl, orc, w, seed):
    env = decode_strip(orc, vra.synth.hdr_environment(w, seed=seed))
    return ibl.torch.from_numpy(np.ascontiguousarray(env)).to(ibl.device)


def test_static_scratch_lambertian_equals_the_allocating_entry_point(ibl, orc):
    """Two calls in a row, the second with a smaller input (the array keeps the first call's
    contents): both must give the bits of ibl_lambertian_dev, which runs the same kernels in
    allocated workspaces."""
    t = ibl.torch
    for in_w, out_w, seed in ((256, 32, 3), (64, 16, 4)):
        env = _env(ibl, orc, in_w, seed)
        want = ibl.lambertian(env, out_w)
        got = t.full((6, out_w, out_w, 4), -1.0, dtype=t.float32, device=ibl.device)
        free_before = t.cuda.mem_get_info()[0]
        ibl.lambertian_static(env, out_w, got)
        t.cuda.synchronize()
        assert t.cuda.mem_get_info()[0] == free_before          # nothing was allocated or freed
        assert t.equal(got, want)
        ref = orc.lambertian(env.cpu().numpy(), out_w)
        err = np.abs(got.cpu().numpy()[..., :3] - ref[..., :3]) / np.maximum(np.abs(ref[..., :3]), 1e-30)
        assert err.max() < 5e-6
    # a light bright enough that some octets go through the second, exact-weights pass
    env = _env(ibl, orc, 64, 5)
    env[0, 20:24, 30:34, :3] = 1.0e4
    got = t.full((6, 16, 16, 4), -1.0, dtype=t.float32, device=ibl.device)
    ibl.lambertian_static(env, 16, got)
    assert t.equal(got, ibl.lambertian(env, 16))
    ibl.set_counting(True)
    try:
        ibl.lambertian(env, 16)
        assert 0 < ibl.last_exact_octets() <= 6 * 16 * 16 // 8
    finally:
        ibl.set_counting(False)


@pytest.mark.parametrize("o48", [1, 0])
@pytest.mark.parametrize("in_w,out_w", [(512, 32), (256, 64), (128, 64)])
def test_static_scratch_holds_the_sizes_the_header_names(ibl, orc, in_w, out_w, o48):
    """(In the 48-fold form, the default for these sizes, and in the hexadecet form behind it.)"""
    t = ibl.torch
    env = _env(ibl, orc, in_w, 6)
    got = t.full((6, out_w, out_w, 4), -1.0, dtype=t.float32, device=ibl.device)
    ibl.set_tuning("lamb_o48", o48)
    try:
        ibl.lambertian_static(env, out_w, got)
        assert t.equal(got, ibl.lambertian(env, out_w))
    finally:
        ibl.set_tuning("lamb_o48", 1)


def test_static_scratch_lambertian_rejects_what_does_not_fit(ibl, orc):
    t = ibl.torch
    env = t.ones((6, 1024, 1024, 4), dtype=t.float32, device=ibl.device)
    out = t.zeros((6, 128, 128, 4), dtype=t.float32, device=ibl.device)
    with pytest.raises(vra.IblError, match="static array holds"):
        ibl.lambertian_static(env, 128, out)
    assert float(out.abs().sum()) == 0.0
    odd = t.ones((6, 30, 30, 4), dtype=t.float32, device=ibl.device)
    with pytest.raises(vra.IblError, match="mirror-octet"):
        ibl.lambertian_static(odd, 5, t.zeros((6, 5, 5, 4), dtype=t.float32, device=ibl.device))
