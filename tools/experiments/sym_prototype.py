"""numpy prototype of the layer matrices in symmetrised coordinates (round 3): R, T, source,
int_flux, int_flux_source (LW) and S_up, S_dn (SW) from the Cholesky factors of C and M and the SVD of
W = F^T L, compared with the oracle in fp64 and in binary128.  Usage: python tools/experiments/sym_prototype.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import checker                                  # noqa: E402
from tests.test_hip_units import _physical_gammas           # noqa: E402


SWEEPS = []


def jacobi_svd(W):
    """one-sided (Hestenes) Jacobi on the columns of W: returns U, sigma with W Y = U diag(sigma)"""
    A = W.copy()
    n = A.shape[1]
    for sweep in range(30):
        rotated = False
        for p in range(n - 1):
            for q in range(p + 1, n):
                a = A[:, p] @ A[:, p]
                b = A[:, q] @ A[:, q]
                g = A[:, p] @ A[:, q]
                if abs(g) <= 2.3e-16 * np.sqrt(a * b):
                    continue
                rotated = True
                zeta = (b - a) / (2.0 * g)
                t = np.sign(zeta) / (abs(zeta) + np.sqrt(1.0 + zeta * zeta)) if zeta != 0 else 1.0
                c = 1.0 / np.sqrt(1.0 + t * t)
                sn = c * t
                ap = c * A[:, p] - sn * A[:, q]
                aq = sn * A[:, p] + c * A[:, q]
                A[:, p], A[:, q] = ap, aq
        if not rotated:
            break
    SWEEPS.append(sweep + 1)
    sig = np.sqrt((A * A).sum(axis=0))
    return A / sig, sig


def layer_lw_sym(g1, g2, b, d, dz):
    n = g1.shape[0]
    rd = 1.0 / d
    C = -(g1 - g2) * rd[:, None] * d[None, :]
    S = g2 * rd[:, None] * d[None, :]
    C = 0.5 * (C + C.T)
    S = 0.5 * (S + S.T)
    M = C - 2.0 * S
    L = np.linalg.cholesky(C)
    F = np.linalg.cholesky(M)
    W = F.T @ L
    U, sig = jacobi_svd(W)
    h = 0.5 * dz
    th = np.tanh(sig * h)
    tp = th / sig
    tm = 1.0 / (sig * th)
    B = F @ U
    G = np.linalg.solve(F.T, U)
    Zp = (B * tp) @ B.T
    Zm = (B * tm) @ B.T
    I = np.eye(n)
    Wp = np.linalg.inv(I + Zp)
    Wm = np.linalg.inv(I + Zm)
    Rh = Wm - Zp @ Wp
    Th = Wp @ ((B * (2.0 / (sig * np.sinh(2.0 * sig * h)))) @ B.T) @ Wm
    bh = rd * b
    gh = np.linalg.solve(F.T, np.linalg.solve(F, bh))
    src = 2.0 * Zp @ (Wp @ gh)
    iflux = 2.0 * (G * tp) @ (B.T @ Wp)
    isrc = 2.0 * (gh * dz - np.linalg.solve(F.T, np.linalg.solve(F, src)))
    Dm = lambda A: A * d[:, None] * rd[None, :]             # noqa: E731
    return dict(reflectance=Dm(Rh), transmittance=Dm(Th), source=d * src, int_flux=Dm(iflux),
                int_flux_source=d * isrc)


def main():
    rng = np.random.default_rng(7)
    for (nreg, ns, lo, hi, label) in ((3, 2, 1e-2, 1.0, "moderate"), (3, 2, 1e-6, 1.0, "graded"),
                                      (3, 2, 1e-9, 3.0, "very graded"), (2, 2, 1e-5, 1.0, "n=4 graded")):
        g0, g1, g2, g3, d0, d1 = _physical_gammas(rng, 200, nreg, ns, ext_lo=lo, ext_hi=hi)
        n = nreg * ns
        b = rng.uniform(0.1, 1.0, (200, n)) * np.abs(np.einsum("mii->mi", g1 + g2))
        for dz in (0.5, 4.0, 30.0):
            o = checker.calc_matrices_lw_eig(dz, g1, g2, b, "oracle")
            q = checker.calc_matrices_lw_eig(dz, g1, g2, b, "quad")
            worst = {}
            for m in range(g1.shape[0]):
                s = layer_lw_sym(g1[m], g2[m], b[m], d1[m], dz)
                for k in s:
                    sc = max(1e-300, np.max(np.abs(q[k][m])))
                    e_new = np.max(np.abs(s[k] - q[k][m])) / sc
                    e_ref = np.max(np.abs(o[k][m] - q[k][m])) / sc
                    w = worst.setdefault(k, [0.0, 0.0])
                    w[0] = max(w[0], e_new)
                    w[1] = max(w[1], e_ref)
            print("sweeps mean %.2f max %d" % (np.mean(SWEEPS), max(SWEEPS)))
            SWEEPS.clear()
            print("%-12s dz=%5.1f " % (label, dz) + "  ".join("%s new %.1e ref %.1e" % (k[:8], v[0], v[1]) for k, v in worst.items()))


if __name__ == "__main__":
    main()
