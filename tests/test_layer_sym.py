"""Kernel-level known answers of the symmetrised layer route (csrc/ss_layer_sym.hpp) on the CPU: the unit
bodies compiled for the host (tests/emul, TEST infrastructure) against the oracle's calc_matrices_*_eig in fp64
and in binary128, on Gamma sets with the structure of the solver's (regions exchanging with detailed balance,
Legendre-Gauss streams):

* moderate and strongly graded extinctions, identical regions (degenerate spectrum), sub-problem orders 1-6;
* the transparent, nearly conservative layer that pinned the design (both regions clear air of extinction 1e-5
  and single-scattering albedo 0.999, eigenvalues over 13 decades): the layer's absorption 1 - R - T, which the
  downward sweep turns into absorbed fluxes through -(Gamma1 + Gamma2)^-1, must come out at least as accurately
  as in the reference's own route;
* the direct-beam sources next to (not at) a resonance of a direct with a diffuse eigenvalue.

The bar is the distance from the binary128 evaluation: the symmetrised route is a different algorithm, so it is
not compared with the reference's fp64 rounding but required to be no farther from exact arithmetic than the
reference's route (and within 1e-9 where that is easy).
"""
import ctypes as C

import numpy as np
import pytest

from tests.test_hip_units import _physical_gammas


@pytest.fixture(scope="module")
def units():
    from oracle import checker
    from tests import emul
    if not checker.have_oracle():
        checker.build(ref=False)
    emul.build()
    return C.CDLL(emul.LIB), checker


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _lw(lib, g1, g2, b, d, dz):
    n = len(d)
    off = (C.c_int * 6)()
    assert lib.emul_lw_rec_offsets(n, off) == 0
    rec = np.zeros(off[5] + 2)
    k = [np.asfortranarray(x, dtype=np.float64) for x in (g1, g2, b, d)]
    rc = lib.emul_lw_sym_unit(n, 0, _p(k[0]), _p(k[1]), _p(k[2]), _p(k[3]), C.c_double(dz), _p(rec))
    m = lambda i, r, c: rec[off[i]:off[i] + r * c].reshape(r, c, order="F")   # noqa: E731
    return rc, dict(reflectance=m(0, n, n), transmittance=m(1, n, n), int_flux=m(2, n, n),
                    source=rec[off[3]:off[3] + n], int_flux_source=rec[off[4]:off[4] + n])


def _sw(lib, g0, g1, g2, g3, d0, d, dz):
    n, nd = len(d), len(d0)
    off = (C.c_int * 7)()
    assert lib.emul_sw_rec_offsets(n, nd, off) == 0
    rec = np.zeros(off[6] + 2)
    k = [np.asfortranarray(x, dtype=np.float64) for x in (g0, g1, g2, g3, d0, d)]
    rc = lib.emul_sw_sym_unit(n, nd, *[_p(x) for x in k], C.c_double(dz), _p(rec))
    m = lambda i, r, c: rec[off[i]:off[i] + r * c].reshape(r, c, order="F")   # noqa: E731
    return rc, dict(reflectance=m(0, n, n), transmittance=m(1, n, n), s_up=m(2, n, nd), s_dn=m(3, n, nd),
                    trans_dir=m(4, nd, nd), int_dir=m(5, nd, nd))


def _dist(a, exact):
    return float(np.max(np.abs(np.asarray(a) - exact))) / max(1.0e-300, float(np.max(np.abs(exact))))


@pytest.mark.parametrize("nreg,ns,lo,hi,identical", [(3, 2, 1e-2, 1.0, False), (3, 2, 1e-6, 1.0, False),
                                                      (2, 2, 1e-5, 1.0, False), (2, 3, 1e-3, 2.0, True),
                                                      (3, 1, 1e-4, 1.0, False), (2, 1, 1e-2, 1.0, False),
                                                      # order 24: the same route with rolled loops (matrices in scratch)
                                                      (3, 8, 1e-3, 1.0, False)])
def test_longwave_unit_vs_exact(units, nreg, ns, lo, hi, identical):
    lib, chk = units
    rng = np.random.default_rng(100 * nreg + ns)
    nset = 40 if nreg * ns <= 6 else 6
    g0, g1, g2, g3, d0, d1 = _physical_gammas(rng, nset, nreg, ns, ext_lo=lo, ext_hi=hi, identical=identical)
    n = nreg * ns
    b = rng.uniform(0.1, 1.0, (nset, n)) * np.abs(np.einsum("mii->mi", g1 + g2))
    for dz in (0.5, 6.0):
        o = chk.calc_matrices_lw_eig(dz, g1, g2, b, "oracle")
        q = chk.calc_matrices_lw_eig(dz, g1, g2, b, "quad")
        worst = {}
        for m in range(g1.shape[0]):
            rc, s = _lw(lib, g1[m], g2[m], b[m], d1[m], dz)
            assert rc == 0
            for k in s:
                w = worst.setdefault(k, [0.0, 0.0])
                w[0] = max(w[0], _dist(s[k], q[k][m]))
                w[1] = max(w[1], _dist(o[k][m], q[k][m]))
        for k, (e_new, e_ref) in worst.items():
            assert e_new <= max(1.0e-9, 2.0 * e_ref), (k, dz, e_new, e_ref)


@pytest.mark.parametrize("nreg,ns,lo,hi", [(3, 2, 1e-2, 1.0), (2, 2, 1e-4, 1.0), (2, 3, 1e-3, 1.0), (2, 1, 1e-2, 1.0),
                                           (3, 8, 1e-3, 1.0)])
def test_shortwave_unit_vs_exact(units, nreg, ns, lo, hi):
    lib, chk = units
    rng = np.random.default_rng(300 + 10 * nreg + ns)
    g0, g1, g2, g3, d0, d1 = _physical_gammas(rng, 40 if nreg * ns <= 6 else 6, nreg, ns, ext_lo=lo, ext_hi=hi)
    for dz in (0.5, 6.0):
        # (mu0 only enters the reference's routine through unused arguments: the Gamma matrices carry it)
        o = chk.calc_matrices_sw_eig(dz, 0.6, g0, g1, g2, g3, "oracle")
        q = chk.calc_matrices_sw_eig(dz, 0.6, g0, g1, g2, g3, "quad")
        worst = {}
        for m in range(g1.shape[0]):
            rc, s = _sw(lib, g0[m], g1[m], g2[m], g3[m], d0[m], d1[m], dz)
            assert rc == 0
            for k in s:
                w = worst.setdefault(k, [0.0, 0.0])
                w[0] = max(w[0], _dist(s[k], q[k][m]))
                w[1] = max(w[1], _dist(o[k][m], q[k][m]))
        for k, (e_new, e_ref) in worst.items():
            assert e_new <= max(1.0e-8, 2.0 * e_ref), (k, dz, e_new, e_ref)


def _transparent_layer(mu0):
    """Two regions of clear air (extinction 1e-5, single-scattering albedo 0.999) that exchange laterally: the
    trunk layer of the reference's test/rami4pilps canopy seen by 3 streams."""
    from spartacus_surface_amd.legendre_gauss import LegendreGauss
    ns, nreg = 3, 2
    lg = LegendreGauss(ns)
    mu, w, tan = np.array(lg.mu), np.array(lg.weight), np.array(lg.tan_ang)
    n = nreg * ns
    f = np.array([0.7, 0.3])
    ext = np.array([1e-5, 1e-5])
    ssa = np.array([0.999, 0.999])
    perim = 4 * 0.3 / 10.0
    fx = np.zeros((2, 2))
    fx[1, 0] = perim / (np.pi * f[0])
    fx[0, 1] = perim / (np.pi * f[1])
    tan0 = np.sqrt(1 - mu0 * mu0) / mu0
    g0 = np.zeros((nreg, nreg)); g1 = np.zeros((n, n)); g2 = np.zeros((n, n)); g3 = np.zeros((n, nreg))
    for jf in range(nreg):
        for jt in range(nreg):
            if jf == jt:
                continue
            g0[jf, jf] -= tan0 * fx[jt, jf]
            g0[jt, jf] = tan0 * fx[jt, jf]
            for s in range(ns):
                g1[s + jf * ns, s + jf * ns] -= tan[s] * fx[jt, jf]
                g1[s + jt * ns, s + jf * ns] = tan[s] * fx[jt, jf]
    for r in range(nreg):
        g0[r, r] -= ext[r] / mu0
        for s in range(ns):
            g1[s + r * ns, s + r * ns] -= ext[r] / mu[s]
            g3[s + r * ns, r] = 0.5 * w[s] * ext[r] * ssa[r]
            for s2 in range(ns):
                g2[s + r * ns, s2 + r * ns] = 0.5 * w[s] * ext[r] * ssa[r] / mu[s2]
    g1 += g2
    d1 = np.repeat(np.sqrt(f), ns) * np.tile(np.sqrt(w * mu), nreg)
    v = np.tile(np.array(lg.hweight), nreg) * np.repeat(f, ns)
    return g0, g1, g2, g3, np.sqrt(f), d1, v


def test_absorption_of_a_transparent_conservative_layer(units):
    """1 - R - T of a layer whose absorption sits 13 decades below its largest eigenvalue: integrated flux by
    the conservation identity the downward sweep uses, -(Gamma1 + Gamma2)^-1 (1 - R - T) v."""
    lib, chk = units
    g0, g1, g2, g3, d0, d1, v = _transparent_layer(0.8)
    n = len(d1)
    b = np.full(n, 1.0e-6)
    dz = 4.0
    o = chk.calc_matrices_lw_eig(dz, g1[None], g2[None], b[None], "oracle")
    q = chk.calc_matrices_lw_eig(dz, g1[None], g2[None], b[None], "quad")
    rc, s = _lw(lib, g1, g2, b, d1, dz)
    assert rc == 0
    minv = -np.linalg.inv(g1 + g2)
    absorb = lambda r, t: minv @ ((np.eye(n) - r - t) @ v)      # noqa: E731
    exact = absorb(q["reflectance"][0], q["transmittance"][0])
    e_ref = float(np.max(np.abs(absorb(o["reflectance"][0], o["transmittance"][0]) - exact) / np.abs(exact)))
    e_new = float(np.max(np.abs(absorb(s["reflectance"], s["transmittance"]) - exact) / np.abs(exact)))
    assert e_new <= 1.0e-5 and e_new <= e_ref, (e_new, e_ref)


@pytest.mark.parametrize("mu0", [1.0, 0.8, 0.3])
def test_direct_sources_of_a_transparent_layer(units, mu0):
    """S_up / S_dn where the unattenuated direct mode (mu^2 ~ 1e-10) sits inside the cluster of tiny diffuse
    eigenvalues: needs their eigenVECTORS accurate (graded order of the QL iteration), not only the eigenvalues."""
    lib, chk = units
    g0, g1, g2, g3, d0, d1, v = _transparent_layer(mu0)
    dz = 4.0
    o = chk.calc_matrices_sw_eig(dz, mu0, g0[None], g1[None], g2[None], g3[None], "oracle")
    q = chk.calc_matrices_sw_eig(dz, mu0, g0[None], g1[None], g2[None], g3[None], "quad")
    rc, s = _sw(lib, g0, g1, g2, g3, d0, d1, dz)
    assert rc == 0
    for k in ("s_up", "s_dn", "reflectance", "transmittance"):
        e_new, e_ref = _dist(s[k], q[k][0]), _dist(o[k][0], q[k][0])
        assert e_new <= max(1.0e-9, 2.0 * e_ref), (k, e_new, e_ref)
