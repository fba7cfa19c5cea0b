"""Shared comparison helpers for the parity tests."""
import numpy as np


def rel_err(got, want):
    """Per-texel, per-channel relative error on RGB, guarded for exact zeros."""
    got = np.asarray(got, np.float64)[..., :3]
    want = np.asarray(want, np.float64)[..., :3]
    den = np.maximum(np.abs(want), 1e-30)
    err = np.abs(got - want) / den
    err[(want == 0) & (got == 0)] = 0.0
    return err


def byte_identical_fraction(a_rgbe, b_rgbe):
    a = np.asarray(a_rgbe).reshape(-1, 4)
    b = np.asarray(b_rgbe).reshape(-1, 4)
    return float((a == b).all(axis=1).mean())


def decode_strip(orc, strip):
    """(6W, W, 4) uint8 -> (6, W, W, 4) float32 through the oracle's decoder."""
    w = strip.shape[1]
    return orc.rgbe_decode(strip).reshape(6, w, w, 4)
