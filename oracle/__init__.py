"""ctypes front-end of the CPU oracle (``oracle/ibl_oracle.c``).

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product path
never does.  Parity status of each function is stated in the header of
``ibl_oracle.c`` (Lambertian/GGX: parity unpinned; LUT: statistical pin against
the reference's committed ``include/ggx_lut.inl``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libibl_oracle.so")
_lib = None

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u32 = ctypes.c_uint32
_u64 = ctypes.c_uint64


def build():
    """Compile the oracle with gcc (idempotent)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_normal_from.argtypes = [_u32, _u32, _u32, _u32, _u32, _f32p]
        L.orc_normal_from.restype = None
        L.orc_texel_directions.argtypes = [_u32, _u32, ctypes.c_int, _f32p]
        L.orc_texel_directions.restype = None
        L.orc_pyramid_levels.argtypes = [_u32]
        L.orc_pyramid_levels.restype = _u32
        L.orc_pyramid_offset.argtypes = [_u32, _u32]
        L.orc_pyramid_offset.restype = ctypes.c_uint64
        L.orc_pyramid_padded_offset.argtypes = [_u32, _u32]
        L.orc_pyramid_padded_offset.restype = ctypes.c_uint64
        L.orc_pyramid_pad.argtypes = [_f32p, _u32, _f32p]
        L.orc_pyramid_pad.restype = None
        L.orc_ggx_level_sampled.argtypes = [_f32p, _u32, _u32, _u32, _u32, _u32, ctypes.c_float, _u32, _u32, _f32p]
        L.orc_ggx_level_sampled.restype = ctypes.c_int
        L.orc_mirror_reproject.argtypes = [_f32p, _u32, _f32p, _f32p, _f32p, _f32p]
        L.orc_mirror_reproject.restype = None
        L.orc_mirror_reflect_camera.argtypes = [_f32p, _f32p, ctypes.c_float, _f32p]
        L.orc_mirror_reflect_camera.restype = None
        L.orc_lambertian.argtypes = [_f32p, _u32, _u32, _u32, _u32, _u32, _f32p, ctypes.c_int]
        L.orc_lambertian.restype = ctypes.c_int
        L.orc_ggx_mip_levels.argtypes = [_u32, _u32, _u32]
        L.orc_ggx_mip_levels.restype = _u32
        L.orc_ggx_params.argtypes = [_u32, _u32, _u32, _u32] + [ctypes.POINTER(ctypes.c_float)] * 3 + [ctypes.POINTER(_u32)]
        L.orc_ggx_params.restype = None
        L.orc_ggx_level.argtypes = [_f32p, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _f32p,
                                    ctypes.POINTER(_u64)]
        L.orc_ggx_level.restype = ctypes.c_int
        L.orc_brdf_lut.argtypes = [_u32, _u32, _u32, _u64, ctypes.c_int, _u32, _u32, _f64p]
        L.orc_brdf_lut.restype = ctypes.c_int
        L.orc_mt19937_nth.argtypes = [_u32, _u32]
        L.orc_mt19937_nth.restype = _u32
        L.orc_lut_jitter.argtypes = [_u64, _u32, _u32, ctypes.c_double,
                                     ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        L.orc_lut_jitter.restype = None
        L.orc_rgbe_decode.argtypes = [_u8p, _f32p, _u32]
        L.orc_rgbe_decode.restype = None
        L.orc_rgbe_encode.argtypes = [_f32p, _u8p, _u32]
        L.orc_rgbe_encode.restype = None
        L.orc_downscale.argtypes = [_f32p, _u32, _u32, _u32, _f32p]
        L.orc_downscale.restype = None
        L.orc_png_filter_rows.argtypes = [_u8p, _u32, _u32, _u8p]
        L.orc_png_filter_rows.restype = None
        L.orc_vec_ops.argtypes = [_f32p, _f32p, _u32, _f32p]
        L.orc_vec_ops.restype = None
        _lib = L
    return _lib


def normal_from(face, x, y, w, h):
    out = np.zeros(3, np.float32)
    lib().orc_normal_from(face, x, y, w, h, out)
    return out


def vec_ops(a, b):
    """The oracle's vector helpers on n pairs of xyz triples: (n, 11) float32 rows of
    dot(a,b), length2(a), normalize(a), a + b, 2*dot(a,b)*b - a."""
    a = np.ascontiguousarray(a, np.float32).reshape(-1, 3)
    b = np.ascontiguousarray(b, np.float32).reshape(-1, 3)
    assert a.shape == b.shape
    out = np.zeros((a.shape[0], 11), np.float32)
    lib().orc_vec_ops(a.reshape(-1), b.reshape(-1), a.shape[0], out.reshape(-1))
    return out


def texel_directions(w, h=None, normalized=True):
    """(6, h, w, 4) float32 direction of every texel (w component 0)."""
    h = w if h is None else h
    out = np.zeros((6, h, w, 4), np.float32)
    lib().orc_texel_directions(w, h, int(bool(normalized)), out.reshape(-1))
    return out


def _env(env):
    env = np.ascontiguousarray(env, dtype=np.float32)
    assert env.ndim == 4 and env.shape[0] == 6 and env.shape[3] == 4, env.shape
    return env


def lambertian(env, out_size, out_begin=0, out_count=None, faithful=False):
    """env: (6, H, W, 4) float32 -> (6, out_size, out_size, 4) float32 (alpha 0)."""
    env = _env(env)
    _, h, w, _ = env.shape
    n_out = 6 * out_size * out_size
    if out_count is None:
        out_count = n_out - out_begin
    out = np.zeros((6, out_size, out_size, 4), np.float32)
    rc = lib().orc_lambertian(env, w, h, out_size, out_begin, out_count, out, int(bool(faithful)))
    if rc:
        raise RuntimeError("orc_lambertian failed: %d" % rc)
    return out


def ggx_mip_levels(top_w, top_h, min_size):
    return int(lib().orc_ggx_mip_levels(top_w, top_h, min_size))


def ggx_params(mip, mips, in_w, in_h):
    r, a2, c = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
    ps = _u32()
    lib().orc_ggx_params(mip, mips, in_w, in_h, ctypes.byref(r), ctypes.byref(a2), ctypes.byref(c), ctypes.byref(ps))
    return dict(roughness=r.value, alpha2=a2.value, cutoff=c.value, patch=ps.value)


def ggx_level(env, out_w, out_h, mip, mips, out_begin=0, out_count=None, want_counts=False):
    env = _env(env)
    _, h, w, _ = env.shape
    n_out = 6 * out_w * out_h
    if out_count is None:
        out_count = n_out - out_begin
    out = np.zeros((6, out_h, out_w, 4), np.float32)
    counts = (_u64 * 3)()
    rc = lib().orc_ggx_level(env, w, h, out_w, out_h, mip, mips, out_begin, out_count, out, counts)
    if rc:
        raise RuntimeError("orc_ggx_level failed: %d" % rc)
    if want_counts:
        return out, dict(patch_tests=counts[0], tested=counts[1], accepted=counts[2])
    return out


def pyramid_levels(w):
    return lib().orc_pyramid_levels(w)


def pyramid_offset(w, level):
    """float4 offset of a level inside the packed pyramid."""
    return lib().orc_pyramid_offset(w, level)


def env_pyramid(env):
    """Box pyramid of a square cube: level k+1 is the 2x2 box mean of level k (the tool's
    --input-downscale filter applied with factor 2).  -> flat float32 array of float4 texels,
    levels back to back."""
    env = _env(env)
    w = env.shape[2]
    levels = pyramid_levels(w)
    parts = [env]
    for _ in range(1, levels):
        parts.append(downscale(parts[-1], 2))
    return np.concatenate([p.reshape(-1) for p in parts]).astype(np.float32)


def env_pyramid_padded(env):
    """The pyramid in the layout of the GPU library: one-texel apron around every face."""
    env = _env(env)
    w = env.shape[2]
    flat = env_pyramid(env)
    out = np.zeros(4 * lib().orc_pyramid_padded_offset(w, pyramid_levels(w)), np.float32)
    lib().orc_pyramid_pad(flat, w, out)
    return out


def ggx_level_sampled(pyramid, in_w, out_w, mip, mips, n_samples, lod_bias=1.0, out_begin=0, out_count=None):
    """The sampled GGX mode (an extension without a counterpart in the reference)."""
    n_out = 6 * out_w * out_w
    if out_count is None:
        out_count = n_out - out_begin
    out = np.zeros((6, out_w, out_w, 4), np.float32)
    pyramid = np.ascontiguousarray(pyramid, dtype=np.float32)
    rc = lib().orc_ggx_level_sampled(pyramid, in_w, out_w, mip, mips, n_samples, lod_bias, out_begin, out_count,
                                     out.reshape(-1))
    if rc:
        raise RuntimeError("orc_ggx_level_sampled failed: %d" % rc)
    return out


MODE_MT19937, MODE_COUNTER, MODE_MIDPOINT = 0, 1, 2


def brdf_lut(n_rough, n_cos, samples_sqrt, seed=0, mode=MODE_COUNTER, cell_begin=0, cell_count=None):
    """-> (n_rough, n_cos, 2) float64; cells outside the range are 0."""
    n = n_rough * n_cos
    if cell_count is None:
        cell_count = n - cell_begin
    out = np.zeros((n_rough, n_cos, 2), np.float64)
    rc = lib().orc_brdf_lut(n_rough, n_cos, samples_sqrt, seed, mode, cell_begin, cell_count, out)
    if rc:
        raise RuntimeError("orc_brdf_lut failed: %d" % rc)
    return out


def mt19937_nth(seed, n):
    return int(lib().orc_mt19937_nth(seed, n))


def rgbe_decode(rgbe):
    rgbe = np.ascontiguousarray(rgbe, dtype=np.uint8)
    out = np.zeros(rgbe.shape, np.float32)
    lib().orc_rgbe_decode(rgbe.reshape(-1), out.reshape(-1), rgbe.size // 4)
    return out


def rgbe_encode(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.zeros(rgb.shape, np.uint8)
    lib().orc_rgbe_encode(rgb.reshape(-1), out.reshape(-1), rgb.size // 4)
    return out


def downscale(env, d):
    env = _env(env)
    _, h, w, _ = env.shape
    out = np.zeros((6, h // d, w // d, 4), np.float32)
    lib().orc_downscale(env, w, h, d, out)
    return out


def png_filter_rows(rgba8):
    """(h, w, 4) uint8 -> (h, 4w+1) uint8 PNG scanlines (filter type first) with the
    writer's per-row filter choice."""
    rgba8 = np.ascontiguousarray(rgba8, np.uint8)
    h, w, c = rgba8.shape
    assert c == 4
    rows = np.zeros((h, 4 * w + 1), np.uint8)
    lib().orc_png_filter_rows(rgba8.reshape(-1), w, h, rows.reshape(-1))
    return rows


def mirror_reflect_camera(transform, camera, mirror_dist):
    """shaders/mirrorLocal.vert:81-82 on the CPU."""
    out = np.zeros(3, np.float32)
    lib().orc_mirror_reflect_camera(np.ascontiguousarray(transform, np.float32).reshape(16),
                                    np.ascontiguousarray(camera, np.float32).reshape(3), float(mirror_dist), out)
    return out


def mirror_reproject(src, camera, box_min, box_max):
    """CPU statement of the mirror-local re-projection of one cube level (ibl_mirror_oracle.c)."""
    src = np.ascontiguousarray(src, np.float32)
    w = src.shape[1]
    out = np.zeros_like(src)
    lib().orc_mirror_reproject(src.reshape(-1), w, np.ascontiguousarray(camera, np.float32).reshape(3),
                               np.ascontiguousarray(box_min, np.float32).reshape(3),
                               np.ascontiguousarray(box_max, np.float32).reshape(3), out.reshape(-1))
    return out
