"""The algebra of csrc/lambertian_o48.hip in numpy: the sum of |n . w| L over the 48 x 48 texel pairs of an
output and an input orbit of the cube's symmetry group, once pair by pair and once through the transforms the
kernel uses (Walsh-Hadamard over the sign flips, Fourier over the axis permutations with the kernel's tables and
register pairings), in float64 for the identities and in float32 -- the kernel's arithmetic, record by record --
for the accuracy.

    python tools/o48_identities.py        # prints the three residuals

tests/test_o48_identities_cpu.py runs the same checks."""
import numpy as np

PERMS = [(0, 1, 2), (1, 0, 2), (2, 1, 0), (2, 0, 1), (0, 2, 1), (1, 2, 0)]      # (P_p v)_x = v[PERMS[p][x]]
PAR = [0, 1, 1, 0, 1, 0]
INV = [0, 1, 2, 5, 4, 3]
EVEN = [[0, 3, 5], [5, 0, 3], [3, 5, 0]]       # the even permutation that carries character a to character b
ODD = [[1, 4, 2], [4, 2, 1], [2, 1, 4]]
R2I = [[[1, 0], [0, 1]], [[-1, 1], [0, 1]], [[0, -1], [-1, 0]], [[0, -1], [1, -1]], [[1, 0], [1, -1]], [[-1, 1], [-1, 0]]]
KS = [0, 3, 5, 6]                              # even characters of the sign group: 1, xy, xz, yz


def parity(a):
    return sum(1 for i in range(3) for j in range(i) if a[j] > a[i]) % 2


def compose(a, b):
    return tuple(a[b[i]] for i in range(3))


def inverse(a):
    r = [0] * 3
    for i in range(3):
        r[a[i]] = i
    return tuple(r)


def act(rho, k):
    """The character k carried through the permutation rho (a set of axes)."""
    r = 0
    for i in range(3):
        if k >> i & 1:
            r |= 1 << rho[i]
    return r


def chi(k, s):
    return -1.0 if bin(k & s).count("1") & 1 else 1.0


def signs(s):
    return np.array([-1.0 if s >> i & 1 else 1.0 for i in range(3)])


def member(v, s, p):
    return signs(s) * np.array([v[PERMS[p][i]] for i in range(3)])


def tables_are_consistent():
    """The kernel's tables against their definitions."""
    pidx = {p: i for i, p in enumerate(PERMS)}
    ok = all(PAR[i] == parity(p) and INV[i] == pidx[inverse(p)] for i, p in enumerate(PERMS))
    for a, k in enumerate(KS[1:]):
        for b, kp in enumerate(KS[1:]):
            for table, par in ((EVEN, 0), (ODD, 1)):
                p = PERMS[table[a][b]]
                ok = ok and act(p, k) == kp and parity(p) == par
    # R2I[p] = R2(P_p^-1), R2 the permutation representation on x + y + z = 0 in the basis (e_x - e_y, e_y - e_z)
    basis = np.array([[1, -1, 0], [0, 1, -1]], float).T
    for i, p in enumerate(PERMS):
        q = inverse(p)
        m = np.zeros((3, 3))
        for x in range(3):
            m[q[x], x] = 1
        r2 = np.linalg.lstsq(basis, m @ basis, rcond=None)[0]
        ok = ok and np.allclose(r2, np.array(R2I[i], float))
    return ok


def brute_force(n0, g, L):
    """E[t, r] = sum over the records and their members of |member(n0, t, r) . member(g, s, p)| L[s, p] (float64)."""
    E = np.zeros((8, 6))
    G = [[np.stack([signs(s)[i] * g[:, PERMS[p][i]].astype(float) for i in range(3)], 1) for p in range(6)] for s in range(8)]
    for t in range(8):
        for r in range(6):
            nn = member(n0.astype(float), t, r)
            E[t, r] = sum((np.abs(G[s][p] @ nn) * L[:, s, p]).sum() for s in range(8) for p in range(6))
    return E


def records(L):
    """What lo_prep stores per record: (Lts, LR, LA) in float64."""
    n = L.shape[0]
    Lh = np.zeros((n, 4, 6))
    for q, k in enumerate(KS):
        for p in range(6):
            Lh[:, q, p] = sum(chi(k, s) * L[:, s, p].astype(float) for s in range(8))
    lts = np.stack([Lh[:, 0, :].sum(1), sum((-1) ** PAR[p] * Lh[:, 0, p] for p in range(6))], 1)
    lr = sum(Lh[:, 0, p][:, None, None] * np.array(R2I[INV[p]], float)[None] for p in range(6))
    la = np.zeros((n, 3, 3, 2))
    for a in range(3):
        for c in range(3):
            e, o = Lh[:, a + 1, EVEN[a][c]], Lh[:, a + 1, ODD[a][c]]
            la[:, a, c, 0], la[:, a, c, 1] = e + o, e - o
    return lts, lr, la


def through_the_transforms(n0, g, L, dtype=np.float64, chunk=256):
    """The kernel's arithmetic in `dtype`, sums over chunks of records in dtype and over the chunks in float64."""
    f = dtype
    n = g.shape[0]
    lts, lr, la = (x.astype(f) for x in records(L))
    M = (n0.astype(f)[:, None, None] * g.astype(f).T[None, :, :]).astype(f)        # M[j][l][record] = n0_j g_l

    def coefficients(A, B, C):
        s1, s2 = A + B, A - B
        p, q, t = np.maximum(s1, C), np.minimum(s1, C), np.maximum(C, np.abs(s2))
        m = np.median(np.stack([s2, -C, C]), 0).astype(f)
        return {0: p + t, 3: p - t, 5: q + m, 6: q - m}                            # a quarter of the coefficients

    w = [coefficients(M[0][t[0]], M[1][t[1]], M[2][t[2]]) for t in PERMS]

    def total(x):
        return sum(float(np.add.reduce(x[s:s + chunk].astype(f), dtype=f)) for s in range(0, n, chunk))

    accA = np.zeros((3, 3, 2))
    for a in range(3):
        for b, kp in enumerate(KS[1:]):
            vp, vm = np.zeros(n, f), np.zeros(n, f)
            for c in range(3):
                te, to = EVEN[b][c], ODD[b][c]
                vp = vp + (w[te][kp] + w[to][kp]) * la[:, a, c, 0]
                vm = vm + (w[te][kp] - w[to][kp]) * la[:, a, c, 1]
            accA[a, b] = total(vp), total(vm)
    f0, f1, f2, f3, f4, f5 = (w[p][0] for p in range(6))
    et = total(((f0 + f3 + f5) + (f1 + f2 + f4)) * lts[:, 0])
    es = total(((f0 + f3 + f5) - (f1 + f2 + f4)) * lts[:, 1])
    fc = [[f0 - f1 + f4 - f5, f1 - f2 - f3 + f5], [-f2 + f3 + f4 - f5, f0 + f1 - f3 - f4]]
    X = np.array([[total(fc[i][0] * lr[:, 0, j] + fc[i][1] * lr[:, 1, j]) for j in range(2)] for i in range(2)])
    # lo_final
    E = np.zeros((8, 6))
    for r in range(6):
        e0 = 4.0 / 6.0 * (et + (-1) ** PAR[r] * es + 2.0 * np.trace(np.array(R2I[r], float) @ X))
        ek = []
        for a in range(3):
            b = [bb for bb in range(3) if EVEN[a][bb] == r or ODD[a][bb] == r][0]
            ek.append(2.0 * (accA[a, b, 0] + (-1) ** PAR[r] * accA[a, b, 1]))
        for t in range(8):
            E[t, r] = 0.125 * (e0 + sum(chi(k, t) * ek[a] for a, k in enumerate(KS[1:])))
    return E


def sample(n, hw=64, seed=3):
    rng = np.random.default_rng(seed)
    c = lambda i: 1 - (2 * i + 1) / (2 * hw)
    d = np.stack([c(rng.integers(0, hw, n)), c(rng.integers(0, hw, n)), np.ones(n)], 1)
    l2 = (d * d).sum(1)
    g = (d / (l2 * l2)[:, None]).astype(np.float32)
    L = (rng.uniform(0.2, 5, (n, 8, 6)) * np.where(rng.uniform(size=(n, 8, 6)) < 0.001, 1000, 1)).astype(np.float32)
    n0 = np.array([c(5), c(20), 1.0])
    n0 = (n0 / np.linalg.norm(n0)).astype(np.float32)
    return n0, g, L


def residuals(n=2048):
    n0, g, L = sample(n)
    want = brute_force(n0, g, L)
    exact = through_the_transforms(n0, g, L, np.float64)
    single = through_the_transforms(n0, g, L, np.float32)
    return float(np.abs(exact / want - 1).max()), float(np.abs(single / want - 1).max())


if __name__ == "__main__":
    print("tables consistent:", tables_are_consistent())
    e64, e32 = residuals()
    print("through the transforms against pair by pair: float64 %.3g, float32 %.3g (relative)" % (e64, e32))
