"""Converged values of the split-sum table the reference's generateGGXLUT estimates by
Monte Carlo (cube.cpp:346-385), as a fixture: tests/golden/ggx_lut_converged.npy
(256 x 256 x 2 float32, row = roughness, column = cos(theta_v), as the .inl).

The reference draws u = (ux, uy) uniformly, maps it to a half vector h with
phi = 2 pi ux, cos^2(theta_h) = (1 - uy) / (1 + (alpha^2 - 1) uy) (:367-371), reflects
v about h (:372) and, where l.z > 0 (:373), adds G1(l) G1(v) {1 - F, F} (v.h)/(h.z v.z)
with F = (1 - v.h)^5 and G1(x) = x.z / (x.z (1 - k) + k), k = alpha / 2 (:338-341,374-379).
The table entry is the mean over the unit square, i.e. a double integral.  This script
evaluates that integral in float64 by deterministic quadrature, independently of the
sampling code in oracle/:

  * outer variable t = tan(theta_h) / alpha, for which uy = t^2 / (1 + t^2);
  * l.z > 0 holds for all phi when theta_h < pi/4 - theta_v/2, for no phi when
    theta_h > pi/4 + theta_v/2, and in between for |phi| < phi0 with
    cos(phi0) = cos(theta_v) (1 - 2 cos^2(theta_h)) / (2 cos(theta_h) sin(theta_v) sin(theta_h));
    the integrand is even in phi and vanishes at phi0, so the inner integral runs over
    [0, phi0] with Gauss-Legendre nodes and no discontinuity is ever straddled;
  * the outer integral uses Gauss-Legendre panels that are geometric in t (the GGX weight
    lives at t ~ 1, the rest of the integrand at t ~ 1/alpha), with an s^2 substitution in
    the panels that touch the two kinks.

Run with --check to compare two node counts (the committed file used the larger one and
differed from the smaller by < 1e-7 everywhere).
"""
import os
import sys
from multiprocessing import Pool

import numpy as np

N_ROUGH = N_COS = 256
HERE = os.path.dirname(os.path.abspath(__file__))


def _gl(n):
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w          # on [0, 1]


def _panels(a, b, xs, ws, kind):
    """Nodes and weights of one panel [a, b] per cell (a, b: (C,)); kind 0 linear,
    +1 clusters nodes at a (t = a + (b-a) s^2), -1 at b."""
    d = (b - a)[:, None]
    if kind == 0:
        return a[:, None] + d * xs, d * ws
    if kind > 0:
        return a[:, None] + d * xs ** 2, d * 2.0 * xs * ws
    return b[:, None] - d * xs ** 2, d * 2.0 * xs * ws


def row_values(args):
    ri, n_outer, n_inner, k_full, k_part = args
    roughness = (ri + 0.5) / N_ROUGH
    alpha = roughness * roughness
    k = 0.5 * alpha
    cv = (np.arange(N_COS) + 0.5) / N_COS
    sv = np.sqrt(1.0 - cv * cv)
    theta_v = np.arccos(cv)
    t1 = np.tan(np.pi / 4 - theta_v / 2) / alpha
    t2 = np.tan(np.pi / 4 + theta_v / 2) / alpha
    xs, ws = _gl(n_outer)
    xi, wi = _gl(n_inner)

    nodes, weights, partial = [], [], []
    # all phi pass: [0, t1], panels [0, t1 2^-K], [t1 2^-K, t1 2^-(K-1)], ..., [t1/2, t1]
    edges = [np.zeros_like(t1)] + [t1 * 2.0 ** (-j) for j in range(k_full, -1, -1)]
    for j in range(len(edges) - 1):
        t, w = _panels(edges[j], edges[j + 1], xs, ws, -1 if j == len(edges) - 2 else 0)
        nodes.append(t); weights.append(w); partial.append(np.zeros(t.shape, bool))
    # some phi pass: [t1, t2] in k_part panels of equal ratio
    ratio = t2 / t1
    for j in range(k_part):
        a, b = t1 * ratio ** (j / k_part), t1 * ratio ** ((j + 1) / k_part)
        t, w = _panels(a, b, xs, ws, 1 if j == 0 else (-1 if j == k_part - 1 else 0))
        nodes.append(t); weights.append(w); partial.append(np.ones(t.shape, bool))
    t = np.concatenate(nodes, axis=1)                 # (C, T)
    w = np.concatenate(weights, axis=1)
    part = np.concatenate(partial, axis=1)

    w = w * 2.0 * t / (1.0 + t * t) ** 2              # d(uy)
    theta = np.arctan(alpha * t)
    c, s = np.cos(theta), np.sin(theta)
    with np.errstate(divide="ignore", invalid="ignore"):
        c0 = cv[:, None] * (1.0 - 2.0 * c * c) / (2.0 * c * sv[:, None] * s)
    phi0 = np.where(part, np.arccos(np.clip(c0, -1.0, 1.0)), np.pi)

    phi = phi0[:, :, None] * xi                        # (C, T, P)
    wphi = phi0[:, :, None] * wi / np.pi
    vh = sv[:, None, None] * s[:, :, None] * np.cos(phi) + (cv[:, None] * c)[:, :, None]
    lz = np.maximum(2.0 * vh * c[:, :, None] - cv[:, None, None], 0.0)
    g = lz / (lz * (1.0 - k) + k) * vh / c[:, :, None] / (cv * (1.0 - k) + k)[:, None, None]
    fres = (1.0 - vh) ** 5
    slope = ((g * (1.0 - fres) * wphi).sum(axis=2) * w).sum(axis=1)
    bias = ((g * fres * wphi).sum(axis=2) * w).sum(axis=1)
    return ri, np.stack([slope, bias], axis=-1)


def table(n_outer, n_inner, k_full=40, k_part=24, rows=None, workers=8):
    rows = list(range(N_ROUGH)) if rows is None else list(rows)
    out = np.zeros((N_ROUGH, N_COS, 2))
    with Pool(workers) as pool:
        for ri, vals in pool.imap_unordered(row_values, [(r, n_outer, n_inner, k_full, k_part) for r in rows]):
            out[ri] = vals
    return out


def main():
    if "--check" in sys.argv:
        rows = [0, 1, 7, 16, 33, 41, 64, 128, 200, 255]
        a = table(12, 32, rows=rows)
        b = table(20, 64, k_full=44, k_part=40, rows=rows)
        d = np.abs(a - b)[rows]
        print("max |coarse - fine| per row:", dict(zip(rows, d.max(axis=(1, 2)))))
        return
    full = table(20, 64, k_full=44, k_part=40)
    np.save(os.path.join(HERE, "ggx_lut_converged.npy"), full.astype(np.float32))
    print("wrote ggx_lut_converged.npy", full.shape, float(full[..., 0].mean()), float(full[..., 1].mean()))


if __name__ == "__main__":
    main()
