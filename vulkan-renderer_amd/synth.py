"""Seeded procedural HDR environments, produced directly as RGBE cube strips (the
tool's input format: a W x 6W RGBA8 image, faces +X,-X,+Y,-Y,+Z,-Z stacked
vertically, shared-exponent radiance, src/util.cpp:93-111).

Smooth per-direction gradients in roughly [0.2, 5], plus one small "sun" disc of
radiance ~4e3 so the dynamic range exceeds 10^3:1 and the cutoff-weight path of
the GGX prefilter (cube.cpp:284-290) matters, plus low-amplitude texel noise so
that neighbouring texels never carry identical bytes.
"""
import numpy as np


def cube_directions(w):
    """(6, w, w, 3) float64 unit directions of texel centres (cube.cpp:31-50)."""
    c = (np.arange(w, dtype=np.float64) + 0.5) / w * 2.0 - 1.0
    t, s = np.meshgrid(c, c, indexing="ij")        # t: rows (y), s: columns (x)
    one = np.ones_like(s)
    faces = [
        (one, -t, -s), (-one, -t, s), (s, one, t), (s, -one, -t), (s, -t, one), (-s, -t, -one),
    ]
    d = np.stack([np.stack(f, axis=-1) for f in faces], axis=0)
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def encode_rgbe(rgb):
    """float (..., 3) -> uint8 (..., 4), truncating shared-exponent encoding."""
    rgb = np.asarray(rgb, dtype=np.float32)
    m = rgb.max(axis=-1)
    out = np.zeros(rgb.shape[:-1] + (4,), np.uint8)
    ok = m > 1e-32
    frac, exp = np.frexp(np.where(ok, m, 1.0).astype(np.float32))
    scale = (256.0 * frac / np.where(ok, m, 1.0)).astype(np.float32)
    mant = np.clip((rgb * scale[..., None]).astype(np.int32), 0, 255).astype(np.uint8)
    out[..., :3] = np.where(ok[..., None], mant, 0)
    out[..., 3] = np.where(ok, np.clip(exp + 128, 0, 255), 0).astype(np.uint8)
    return out


def hdr_environment(w, seed=1234, sun=True):
    """-> (6*w, w, 4) uint8 RGBE strip."""
    rng = np.random.default_rng(seed)
    d = cube_directions(w)
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    k = rng.uniform(0.5, 2.5, size=(3, 3))
    ph = rng.uniform(0.0, 2.0 * np.pi, size=(3, 3))
    rgb = np.empty(d.shape, np.float64)
    for c in range(3):
        g = (np.sin(k[c, 0] * np.pi * x + ph[c, 0]) + np.sin(k[c, 1] * np.pi * y + ph[c, 1])
             + np.sin(k[c, 2] * np.pi * z + ph[c, 2])) / 3.0
        rgb[..., c] = 0.2 + 4.8 * (0.5 + 0.5 * g) ** 2
    rgb *= 1.0 + 0.05 * rng.standard_normal(size=rgb.shape[:-1] + (1,))
    rgb = np.maximum(rgb, 0.05)
    if sun:
        sd = np.array([0.45, 0.8, 0.39])
        sd /= np.linalg.norm(sd)
        cosang = x * sd[0] + y * sd[1] + z * sd[2]
        disc = np.clip((cosang - np.cos(0.06)) / (1.0 - np.cos(0.06)), 0.0, 1.0)
        rgb += disc[..., None] * np.array([4000.0, 3600.0, 3000.0])
    return encode_rgbe(rgb).reshape(6 * w, w, 4)


def constant_environment(w, rgb=(1.0, 2.0, 0.5)):
    """-> (6*w, w, 4) uint8 RGBE strip of one colour."""
    px = encode_rgbe(np.asarray(rgb, np.float32)[None, :])[0]
    return np.broadcast_to(px, (6 * w, w, 4)).copy()


def single_texel_environment(w, face=2, x=None, y=None, value=1000.0):
    """All-zero strip with one bright texel."""
    out = np.zeros((6, w, w, 4), np.uint8)
    x = w // 3 if x is None else x
    y = w // 2 if y is None else y
    out[face, y, x] = encode_rgbe(np.asarray([[value, value * 0.5, value * 0.25]], np.float32))[0]
    return out.reshape(6 * w, w, 4)


def sun_on_dim_sky_environment(w, ratio=1e5, seed=3):
    """-> (6*w, w, 4) uint8 RGBE strip: a smooth, dim sky (radiance 0.02-0.08) and a sun of ONE
    texel `ratio` times as bright as the sky around it."""
    rng = np.random.default_rng(seed)
    d = cube_directions(w)
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    rgb = np.empty(d.shape, np.float64)
    for c, (a, b) in enumerate(((1.3, 0.4), (0.9, 1.1), (1.7, 2.0))):
        rgb[..., c] = 0.05 + 0.03 * np.sin(a * np.pi * x + b) * np.cos(a * np.pi * y - b) + 0.01 * z
    rgb *= 1.0 + 0.02 * rng.standard_normal(size=rgb.shape[:-1] + (1,))
    face, sy, sx = 2, (5 * w) // 8 + 1, w // 3 + 2
    rgb[face, sy, sx] = rgb[face, sy, sx] * ratio * np.array([1.0, 0.9, 0.8])
    return encode_rgbe(rgb).reshape(6 * w, w, 4)
