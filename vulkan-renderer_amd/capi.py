"""ctypes binding of ``libibl_hip.so`` (C ABI: ``include/ibl_hip.h``).

This is plumbing for the tests and the benchmark: torch supplies device memory
and streams, every computation happens inside the HIP library.  Nothing here
falls back to a CPU implementation: if the library is missing, or there is no
gfx950 device, construction of :class:`Ibl` raises.
"""
import ctypes
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
# (IBL_HIP_LIB: another build of the same library, e.g. a kernel variant under tools/variants/)
LIB_PATH = os.environ.get("IBL_HIP_LIB") or os.path.join(_HERE, "libibl_hip.so")
HEADER_PATH = os.path.join(_ROOT, "include", "ibl_hip.h")

SHARD_LAMBERTIAN = 0
SHARD_GGX = 1
LUT_JITTER_COUNTER = 1
LUT_JITTER_MIDPOINT = 2

_u32 = ctypes.c_uint32
_u64 = ctypes.c_uint64
_vp = ctypes.c_void_p
_int = ctypes.c_int

_SIGNATURES = {
    "ibl_init": ([_int], _int),
    "ibl_init_all": ([_int], _int),
    "ibl_initialized_devices": ([], _int),
    "ibl_set_kernel_mode": ([_int], _int),
    "ibl_set_tuning": ([ctypes.c_char_p, ctypes.c_double], _int),
    "ibl_get_tuning": ([ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)], _int),
    "ibl_last_exact_octets": ([ctypes.POINTER(_u64)], _int),
    "ibl_last_wave_visits": ([ctypes.POINTER(_u64)], _int),
    "ibl_shutdown": ([], _int),
    "ibl_last_error": ([], ctypes.c_char_p),
    "ibl_device_count": ([], _int),
    "ibl_reserve": ([_u32, _u32, _u32, _u32], _int),
    "ibl_ggx_mip_levels": ([_u32, _u32, _u32], _u32),
    "ibl_ggx_level_params": ([_u32, _u32, _u32, _u32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                              ctypes.POINTER(ctypes.c_float), ctypes.POINTER(_u32)], _int),
    "ibl_lambertian_dev": ([_vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp], _int),
    "ibl_lambertian_shard_dev": ([_vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp], _int),
    "ibl_lambertian_static_dev": ([_vp, _u32, _u32, _u32, _vp, _vp], _int),
    "ibl_ggx_level_dev": ([_vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp], _int),
    "ibl_ggx_level_shard_dev": ([_vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp], _int),
    "ibl_shard_texels": ([_int, _u32, _u32, _u32, _u32, _u32], _u32),
    "ibl_shard_indices": ([_int, _u32, _u32, _u32, _u32, _u32, _u32, _vp], _int),
    "ibl_shard_pack_dev": ([_int, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp], _int),
    "ibl_shard_unpack_dev": ([_int, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp], _int),
    "ibl_mirror_reproject_dev": ([_vp, _u32, _vp, _vp, _vp, _vp, _vp], _int),
    "ibl_mirror_reflect_camera": ([_vp, _vp, ctypes.c_float, _vp], _int),
    "ibl_brdf_lut_dev": ([_u32, _u32, _u32, _u64, _int, _u32, _u32, _vp, _vp], _int),
    "ibl_rgbe_decode_dev": ([_vp, _vp, _u32, _vp], _int),
    "ibl_rgbe_encode_dev": ([_vp, _vp, _u32, _vp], _int),
    "ibl_downscale_dev": ([_vp, _u32, _u32, _u32, _vp, _vp], _int),
    "ibl_texel_directions_dev": ([_u32, _u32, _int, _vp, _vp], _int),
    "ibl_env_pyramid_levels": ([_u32], _u32),
    "ibl_env_pyramid_texels": ([_u32], _u64),
    "ibl_env_pyramid_dev": ([_vp, _u32, _vp, _vp], _int),
    "ibl_ggx_level_sampled_dev": ([_vp, _u32, _u32, _u32, _u32, _u32, ctypes.c_float, _u32, _u32, _vp, _vp], _int),
    "ibl_ggx_level_sampled": ([_vp, _u32, _u32, _u32, _u32, _u32, ctypes.c_float, _vp], _int),
    "ibl_lambertian": ([_vp, _u32, _u32, _u32, _vp], _int),
    "ibl_ggx_level": ([_vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp], _int),
    "ibl_brdf_lut": ([_u32, _u32, _u32, _u64, _int, _vp], _int),
    "ibl_rgbe_decode": ([_vp, _vp, _u32], _int),
    "ibl_rgbe_encode": ([_vp, _vp, _u32], _int),
    "ibl_downscale": ([_vp, _u32, _u32, _u32, _vp], _int),
    "ibl_set_progress": ([_vp, _vp], _int),
    "ibl_bake_rgbe": ([_vp, _u32, _u32, _u32, _vp, _vp], _int),
    "ibl_bake_rgbe_sampled": ([_vp, _u32, _u32, _u32, _u32, ctypes.c_float, _vp, _vp], _int),
    "ibl_png_rows_bytes": ([_u32, _u32], _u64),
    "ibl_png_filter_rows_dev": ([_vp, _u32, _u32, _vp, _vp], _int),
    "ibl_png_filter_rows": ([_vp, _u32, _u32, _vp], _int),
    "ibl_bake_png_rows": ([_vp, _u32, _u32, _u32, _u32, ctypes.c_float, _vp, _vp], _int),
    "ibl_set_timing": ([_int], _int),
    "ibl_last_kernel_ms": ([ctypes.POINTER(ctypes.c_float)], _int),
    "ibl_set_counting": ([_int], _int),
    "ibl_last_counts": ([ctypes.POINTER(_u64)] * 3, _int),
    "ibl_ggx_level_plan": ([_u32, _u32, _u32, _u32, ctypes.POINTER(_u32), ctypes.POINTER(_u32), ctypes.POINTER(_u64),
                            ctypes.POINTER(_u64)], _int),
    "ibl_kernel_ms_back": ([_u32, ctypes.POINTER(ctypes.c_float)], _int),
    "ibl_last_wave_work": ([ctypes.POINTER(_u64)] * 2, _int),
}


class IblError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("ibl error %d: %s" % (code, message))
        self.code = code


def build():
    """Compile libibl_hip.so for gfx950 with hipcc (works without a GPU)."""
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc")])


def shard_indices(kind, in_w, out_w, mip, mips, shard, n_shards):
    """Texel index of every slot of a shard (host arithmetic in the library; needs no GPU):
    uint32 array, 0xffffffff for padding."""
    lib = load_library()
    per = lib.ibl_shard_texels(kind, in_w, out_w, mip, mips, n_shards)
    idx = np.empty(per, np.uint32)
    rc = lib.ibl_shard_indices(kind, in_w, out_w, mip, mips, shard, n_shards, idx.ctypes.data)
    if rc != 0:
        raise IblError(rc, lib.ibl_last_error().decode())
    return idx


def ggx_level_plan(in_w, out_w, mip, mips):
    """Runs, launches and workspace bytes of one whole GGX level on the mirror-octet kernel (host
    arithmetic of the library: no GPU needed)."""
    lib = load_library()
    runs, passes, sums, recs = _u32(), _u32(), _u64(), _u64()
    rc = lib.ibl_ggx_level_plan(in_w, out_w, mip, mips, ctypes.byref(runs), ctypes.byref(passes), ctypes.byref(sums),
                                ctypes.byref(recs))
    if rc != 0:
        raise IblError(rc, lib.ibl_last_error().decode())
    return dict(runs=runs.value, passes=passes.value, sums_bytes=sums.value, records_bytes=recs.value)


def exported_symbols():
    """Names of the functions declared in include/ibl_hip.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ibl_[a-z0-9_]+)\s*\(", text)))


_lib = None


def load_library():
    """dlopen the library and attach the prototypes.  torch is imported first so
    that both share one HIP runtime (torch ships its own libamdhip64.so.7 and a
    process must not hold two)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IblError(2, "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (argtypes, restype) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def _ptr(t):
    return _vp(t.data_ptr())


class Ibl:
    """One GPU's view of the library.  Device methods take and return torch CUDA
    tensors shaped (6, H, W, 4); host methods take numpy arrays."""

    def __init__(self, device=None):
        import torch
        self.torch = torch
        self.lib = load_library()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        if self.lib.ibl_device_count() == 0:
            raise IblError(2, "no HIP device visible; the IBL library has no CPU path")
        torch.cuda.set_device(device)
        self.device = torch.device("cuda", device)
        self._check(self.lib.ibl_init(device))

    # -- plumbing -----------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise IblError(rc, self.lib.ibl_last_error().decode())

    def _stream(self, stream):
        if stream is None:
            stream = self.torch.cuda.current_stream(self.device)
        return _vp(stream.cuda_stream)

    def _env(self, env):
        t = self.torch
        assert env.is_cuda and env.dtype == t.float32 and env.is_contiguous(), "env must be a contiguous CUDA float32 tensor"
        assert env.dim() == 4 and env.shape[0] == 6 and env.shape[3] == 4, tuple(env.shape)
        return int(env.shape[2]), int(env.shape[1])

    def close(self):
        self.lib.ibl_shutdown()

    def reserve(self, in_w, lambertian_size=0, ggx_top_w=0, ggx_min_size=16):
        self._check(self.lib.ibl_reserve(in_w, lambertian_size, ggx_top_w, ggx_min_size))

    # -- geometry ---------------------------------------------------------------------
    def ggx_mip_levels(self, top_w, top_h, min_size):
        return int(self.lib.ibl_ggx_mip_levels(top_w, top_h, min_size))

    def ggx_level_params(self, mip, mips, in_w, in_h):
        r, a2, c = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        ps = _u32()
        self._check(self.lib.ibl_ggx_level_params(mip, mips, in_w, in_h, ctypes.byref(r), ctypes.byref(a2),
                                                  ctypes.byref(c), ctypes.byref(ps)))
        return dict(roughness=r.value, alpha2=a2.value, cutoff=c.value, patch=ps.value)

    # -- device-resident entry points ------------------------------------------------
    def lambertian(self, env, out_size, out_begin=0, out_count=None, out=None, stream=None):
        w, h = self._env(env)
        n_out = 6 * out_size * out_size
        if out_count is None:
            out_count = n_out - out_begin
        if out is None:
            out = self.torch.zeros((6, out_size, out_size, 4), dtype=self.torch.float32, device=self.device)
        self._check(self.lib.ibl_lambertian_dev(_ptr(env), w, h, out_size, out_begin, out_count, _ptr(out),
                                                self._stream(stream)))
        return out

    def lambertian_static(self, env, out_size, out, stream=None):
        """ibl_lambertian_static_dev: the whole cube into `out`, with no allocation anywhere in the call."""
        w, h = self._env(env)
        self._check(self.lib.ibl_lambertian_static_dev(_ptr(env), w, h, out_size, _ptr(out), self._stream(stream)))
        return out

    def lambertian_shard(self, env, out_size, shard, n_shards, out=None, stream=None):
        w, h = self._env(env)
        if out is None:
            out = self.torch.zeros((6, out_size, out_size, 4), dtype=self.torch.float32, device=self.device)
        self._check(self.lib.ibl_lambertian_shard_dev(_ptr(env), w, h, out_size, shard, n_shards, _ptr(out),
                                                      self._stream(stream)))
        return out

    def ggx_level(self, env, out_w, mip, mips, out_begin=0, out_count=None, out=None, stream=None, out_h=None):
        w, h = self._env(env)
        out_h = out_w if out_h is None else out_h
        n_out = 6 * out_w * out_h
        if out_count is None:
            out_count = n_out - out_begin
        if out is None:
            out = self.torch.zeros((6, out_h, out_w, 4), dtype=self.torch.float32, device=self.device)
        self._check(self.lib.ibl_ggx_level_dev(_ptr(env), w, h, out_w, out_h, mip, mips, out_begin, out_count,
                                               _ptr(out), self._stream(stream)))
        return out

    def ggx_level_shard(self, env, out_w, mip, mips, shard, n_shards, out=None, stream=None):
        w, h = self._env(env)
        if out is None:
            out = self.torch.zeros((6, out_w, out_w, 4), dtype=self.torch.float32, device=self.device)
        self._check(self.lib.ibl_ggx_level_shard_dev(_ptr(env), w, h, out_w, out_w, mip, mips, shard, n_shards,
                                                     _ptr(out), self._stream(stream)))
        return out

    # -- mirror-local re-projection ----------------------------------------------------------
    def mirror_reflect_camera(self, transform, camera, mirror_dist):
        t = np.ascontiguousarray(transform, np.float32).reshape(16)
        c = np.ascontiguousarray(camera, np.float32).reshape(3)
        out = np.zeros(3, np.float32)
        self._check(self.lib.ibl_mirror_reflect_camera(t.ctypes.data, c.ctypes.data, float(mirror_dist), out.ctypes.data))
        return out

    def mirror_reproject(self, src, camera, box_min, box_max, out=None, stream=None):
        """One level of a mirror-local environment's chain, re-projected through the reflected camera."""
        w, h = self._env(src)
        assert w == h
        if out is None:
            out = self.torch.empty_like(src)
        c, lo, hi = (np.ascontiguousarray(v, np.float32).reshape(3) for v in (camera, box_min, box_max))
        self._check(self.lib.ibl_mirror_reproject_dev(_ptr(src), w, c.ctypes.data, lo.ctypes.data, hi.ctypes.data, _ptr(out),
                                                      self._stream(stream)))
        return out

    # -- shards: the exchange step between GPUs ----------------------------------------------
    def shard_texels(self, kind, in_w, out_w, mip=1, mips=1, n_shards=1):
        return int(self.lib.ibl_shard_texels(kind, in_w, out_w, mip, mips, n_shards))

    def shard_pack(self, kind, in_w, cube, mip, mips, shard, n_shards, stream=None):
        """The shard's texels of `cube` ((6, W, W, 4) float32 or uint8) as a packed (slots, 4) tensor."""
        out_w = int(cube.shape[1])
        per = self.shard_texels(kind, in_w, out_w, mip, mips, n_shards)
        packed = self.torch.empty((per, 4), dtype=cube.dtype, device=self.device)
        self._check(self.lib.ibl_shard_pack_dev(kind, in_w, out_w, mip, mips, shard, n_shards, cube.element_size() * 4,
                                                _ptr(cube), _ptr(packed), self._stream(stream)))
        return packed

    def shard_unpack(self, kind, in_w, packed_all, cube, mip, mips, n_shards, stream=None):
        """Scatters the n_shards packed shards (back to back) into `cube`."""
        out_w = int(cube.shape[1])
        assert packed_all.is_contiguous() and packed_all.dtype == cube.dtype
        self._check(self.lib.ibl_shard_unpack_dev(kind, in_w, out_w, mip, mips, n_shards, cube.element_size() * 4,
                                                  _ptr(packed_all), _ptr(cube), self._stream(stream)))
        return cube

    def brdf_lut(self, n_rough=256, n_cos=256, samples_sqrt=32, seed=0, mode=LUT_JITTER_COUNTER,
                 cell_begin=0, cell_count=None, out=None, stream=None):
        n = n_rough * n_cos
        if cell_count is None:
            cell_count = n - cell_begin
        if out is None:
            out = self.torch.zeros((n_rough, n_cos, 2), dtype=self.torch.float32, device=self.device)
        self._check(self.lib.ibl_brdf_lut_dev(n_rough, n_cos, samples_sqrt, seed, mode, cell_begin, cell_count,
                                              _ptr(out), self._stream(stream)))
        return out

    def rgbe_decode(self, rgbe, stream=None):
        t = self.torch
        assert rgbe.is_cuda and rgbe.dtype == t.uint8 and rgbe.is_contiguous() and rgbe.shape[-1] == 4
        out = t.empty(tuple(rgbe.shape), dtype=t.float32, device=self.device)
        self._check(self.lib.ibl_rgbe_decode_dev(_ptr(rgbe), _ptr(out), rgbe.numel() // 4, self._stream(stream)))
        return out

    def rgbe_encode(self, rgba, stream=None):
        t = self.torch
        assert rgba.is_cuda and rgba.dtype == t.float32 and rgba.is_contiguous() and rgba.shape[-1] == 4
        out = t.empty(tuple(rgba.shape), dtype=t.uint8, device=self.device)
        self._check(self.lib.ibl_rgbe_encode_dev(_ptr(rgba), _ptr(out), rgba.numel() // 4, self._stream(stream)))
        return out

    def downscale(self, env, d, stream=None):
        w, h = self._env(env)
        out = self.torch.empty((6, max(h // max(d, 1), 1), max(w // max(d, 1), 1), 4), dtype=self.torch.float32,
                               device=self.device)
        self._check(self.lib.ibl_downscale_dev(_ptr(env), w, h, d, _ptr(out), self._stream(stream)))
        return out

    # -- sampled GGX mode (an addition; the reference has only the quadrature) ----------
    def env_pyramid(self, env, stream=None):
        """Box pyramid of a square cube as one flat (texels, 4) float32 tensor."""
        w, h = self._env(env)
        out = self.torch.empty((self.lib.ibl_env_pyramid_texels(w), 4), dtype=self.torch.float32, device=self.device)
        self._check(self.lib.ibl_env_pyramid_dev(_ptr(env), w, _ptr(out), self._stream(stream)))
        return out

    def ggx_level_sampled(self, pyramid, in_w, out_w, mip, mips, n_samples, lod_bias=1.0, out_begin=0, out_count=None,
                          out=None, stream=None):
        t = self.torch
        n_out = 6 * out_w * out_w
        if out_count is None:
            out_count = n_out - out_begin
        if out is None:
            out = t.zeros((6, out_w, out_w, 4), dtype=t.float32, device=self.device)
        assert pyramid.is_cuda and pyramid.dtype == t.float32 and pyramid.is_contiguous()
        self._check(self.lib.ibl_ggx_level_sampled_dev(_ptr(pyramid), in_w, out_w, mip, mips, n_samples, float(lod_bias),
                                                       out_begin, out_count, _ptr(out), self._stream(stream)))
        return out

    def ggx_level_sampled_host(self, env, out_w, mip, mips, n_samples, lod_bias=1.0):
        env = self._np_env(env)
        out = np.zeros((6, out_w, out_w, 4), np.float32)
        self._check(self.lib.ibl_ggx_level_sampled(env.ctypes.data, env.shape[2], out_w, mip, mips, n_samples,
                                                   float(lod_bias), out.ctypes.data))
        return out

    def texel_directions(self, w, h=None, normalized=True, stream=None):
        """(6, h, w, 4) float32: the direction of every texel (getNormalFrom, normalised or raw)."""
        h = w if h is None else h
        out = self.torch.empty((6, h, w, 4), dtype=self.torch.float32, device=self.device)
        self._check(self.lib.ibl_texel_directions_dev(w, h, int(bool(normalized)), _ptr(out), self._stream(stream)))
        return out

    def png_filter_rows(self, rgba8, stream=None):
        """(h, w, 4) uint8 device tensor -> (h, 4w+1) uint8 PNG scanlines (filter type first),
        filtered per row as the tool's PNG writer chooses (cube.cpp:141,322)."""
        assert rgba8.dtype == self.torch.uint8 and rgba8.is_contiguous() and rgba8.shape[-1] == 4
        h, w = rgba8.shape[0], rgba8.shape[1]
        rows = self.torch.empty((h, 4 * w + 1), dtype=self.torch.uint8, device=self.device)
        self._check(self.lib.ibl_png_filter_rows_dev(_ptr(rgba8), w, h, _ptr(rows), self._stream(stream)))
        return rows

    # -- measurement --------------------------------------------------------------------
    def set_kernel_mode(self, general_only):
        self._check(self.lib.ibl_set_kernel_mode(int(bool(general_only))))

    def set_tuning(self, name, value):
        self._check(self.lib.ibl_set_tuning(name.encode(), float(value)))

    def get_tuning(self, name):
        v = ctypes.c_double()
        self._check(self.lib.ibl_get_tuning(name.encode(), ctypes.byref(v)))
        return v.value

    def last_exact_octets(self):
        v = _u64()
        self._check(self.lib.ibl_last_exact_octets(ctypes.byref(v)))
        return v.value

    def set_timing(self, on):
        self._check(self.lib.ibl_set_timing(int(bool(on))))

    def last_kernel_ms(self):
        ms = ctypes.c_float()
        self._check(self.lib.ibl_last_kernel_ms(ctypes.byref(ms)))
        return ms.value

    def kernel_ms_back(self, calls_back):
        ms = ctypes.c_float()
        self._check(self.lib.ibl_kernel_ms_back(int(calls_back), ctypes.byref(ms)))
        return ms.value

    def set_counting(self, on):
        self._check(self.lib.ibl_set_counting(int(bool(on))))

    def last_counts(self):
        a, b, c = _u64(), _u64(), _u64()
        self._check(self.lib.ibl_last_counts(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        d, e = _u64(), _u64()
        self._check(self.lib.ibl_last_wave_work(ctypes.byref(d), ctypes.byref(e)))
        f = _u64()
        self._check(self.lib.ibl_last_wave_visits(ctypes.byref(f)))
        return dict(patch_tests=a.value, tested=b.value, accepted=c.value, wave_records=d.value, wave_patterns=e.value,
                    wave_visits=f.value)

    # -- host-buffer entry points ---------------------------------------------------------
    @staticmethod
    def _np_env(env):
        env = np.ascontiguousarray(env, dtype=np.float32)
        assert env.ndim == 4 and env.shape[0] == 6 and env.shape[3] == 4, env.shape
        return env

    def lambertian_host(self, env, out_size):
        env = self._np_env(env)
        out = np.zeros((6, out_size, out_size, 4), np.float32)
        self._check(self.lib.ibl_lambertian(env.ctypes.data, env.shape[2], env.shape[1], out_size, out.ctypes.data))
        return out

    def ggx_level_host(self, env, out_w, mip, mips):
        env = self._np_env(env)
        out = np.zeros((6, out_w, out_w, 4), np.float32)
        self._check(self.lib.ibl_ggx_level(env.ctypes.data, env.shape[2], env.shape[1], out_w, out_w, mip, mips,
                                           out.ctypes.data))
        return out

    def brdf_lut_host(self, n_rough=256, n_cos=256, samples_sqrt=32, seed=0, mode=LUT_JITTER_COUNTER):
        out = np.zeros((n_rough, n_cos, 2), np.float32)
        self._check(self.lib.ibl_brdf_lut(n_rough, n_cos, samples_sqrt, seed, mode, out.ctypes.data))
        return out

    def rgbe_decode_host(self, rgbe):
        rgbe = np.ascontiguousarray(rgbe, dtype=np.uint8)
        out = np.zeros(rgbe.shape, np.float32)
        self._check(self.lib.ibl_rgbe_decode(rgbe.ctypes.data, out.ctypes.data, rgbe.size // 4))
        return out

    def rgbe_encode_host(self, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        out = np.zeros(rgba.shape, np.uint8)
        self._check(self.lib.ibl_rgbe_encode(rgba.ctypes.data, out.ctypes.data, rgba.size // 4))
        return out

    def downscale_host(self, env, d):
        env = self._np_env(env)
        out = np.zeros((6, env.shape[1] // d, env.shape[2] // d, 4), np.float32)
        self._check(self.lib.ibl_downscale(env.ctypes.data, env.shape[2], env.shape[1], d, out.ctypes.data))
        return out

    def bake_rgbe(self, rgbe, input_downscale=1, cubemap_size=16, lambertian=True, ggx=True, ggx_samples=0,
                  lod_bias=1.0):
        """rgbe: (6*W, W, 4) uint8 strip.  Returns (lambertian strip or None, [ggx level strips]).
        ggx_samples > 0 selects the sampled GGX mode (an addition), 0 the reference's quadrature."""
        rgbe = np.ascontiguousarray(rgbe, dtype=np.uint8)
        w = rgbe.shape[1]
        assert rgbe.shape == (6 * w, w, 4), rgbe.shape
        lam = np.zeros((6 * cubemap_size, cubemap_size, 4), np.uint8) if lambertian else None
        mips = self.ggx_mip_levels(w, w, cubemap_size) if ggx else 0
        sizes = [w >> m for m in range(1, mips + 1)]
        total = sum(6 * s * s * 4 for s in sizes)
        gbuf = np.zeros(max(total, 1), np.uint8) if ggx else None
        self._check(self.lib.ibl_bake_rgbe_sampled(rgbe.ctypes.data, w, input_downscale, cubemap_size, ggx_samples,
                                                   float(lod_bias), lam.ctypes.data if lambertian else None,
                                                   gbuf.ctypes.data if ggx else None))
        levels, off = [], 0
        for s in sizes:
            n = 6 * s * s * 4
            levels.append(gbuf[off:off + n].reshape(6 * s, s, 4))
            off += n
        return lam, levels

    def png_filter_rows_host(self, rgba8):
        rgba8 = np.ascontiguousarray(rgba8, dtype=np.uint8)
        h, w, c = rgba8.shape
        assert c == 4
        rows = np.zeros((h, 4 * w + 1), np.uint8)
        assert rows.size == self.lib.ibl_png_rows_bytes(w, h)
        self._check(self.lib.ibl_png_filter_rows(rgba8.ctypes.data, w, h, rows.ctypes.data))
        return rows

    def bake_png_rows(self, rgbe, input_downscale=1, cubemap_size=16, lambertian=True, ggx=True, ggx_samples=0,
                      lod_bias=1.0, _expose=None):
        """bake_rgbe, but every result is the (6*S, 4*S+1) uint8 PNG scanline stream of its strip.
        _expose: a dict that receives the output buffers ("lam", "ggx") before the call starts, so that a
        progress callback can look at a cube the moment it is reported."""
        rgbe = np.ascontiguousarray(rgbe, dtype=np.uint8)
        w = rgbe.shape[1]
        assert rgbe.shape == (6 * w, w, 4), rgbe.shape
        lam = np.zeros((6 * cubemap_size, 4 * cubemap_size + 1), np.uint8) if lambertian else None
        mips = self.ggx_mip_levels(w, w, cubemap_size) if ggx else 0
        sizes = [w >> m for m in range(1, mips + 1)]
        total = sum(6 * s * (4 * s + 1) for s in sizes)
        gbuf = np.zeros(max(total, 1), np.uint8) if ggx else None
        levels, off = [], 0
        for s in sizes:
            n = 6 * s * (4 * s + 1)
            levels.append(gbuf[off:off + n].reshape(6 * s, 4 * s + 1))
            off += n
        if _expose is not None:
            _expose["lam"], _expose["ggx"] = lam, levels
        self._check(self.lib.ibl_bake_png_rows(rgbe.ctypes.data, w, input_downscale, cubemap_size, ggx_samples,
                                               float(lod_bias), lam.ctypes.data if lambertian else None,
                                               gbuf.ctypes.data if ggx else None))
        return lam, levels
