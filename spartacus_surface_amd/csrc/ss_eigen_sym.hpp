// ss_eigen_sym.hpp - eigen-decomposition of P = (G1 - G2)(G1 + G2) through its
// hidden symmetry (no counterpart in the reference, which calls its general
// real eigen-solver, radtool_calc_matrices_sw_eig.F90:180-186).
//
// With d_i = sqrt(frac_region(i) * weight_stream(i) * mu_stream(i)) and
// D = diag(d) both D^-1 (G1 - G2) D = -C and D^-1 G2 D = S are symmetric:
//   * lateral exchange obeys detailed balance, rate(b <- a) frac_a = rate(a <- b) frac_b
//     (exchange_rates, ss_geometry.hpp; reference radsurf_forest_sw.F90:301-321);
//   * in-region scattering between streams is ~ weight_to / mu_from
//     (reference radsurf_forest_sw.F90:357-366);
//   * wall scattering is ~ vweight_to * tan_from, and vweight ~ weight * sin, so
//     vweight / tan ~ weight * mu again (radsurf_urban_sw.F90:470-473,
//     radtool_legendre_gauss.F90:73-77).
// C is positive definite (extinction and exchange losses on the diagonal), so
//   D^-1 P D = (-C)(-C + 2 S) = C M,   M = C - 2 S symmetric,
// and with the Cholesky factor C = L L^T the matrix K = L^T M L is symmetric and
// similar to P:  K Y = Y Lambda  =>  P (D L Y) = (D L Y) Lambda.
// A symmetric tridiagonal-QL solver (EISPACK tred2 + tql2) costs about half the
// arithmetic of the general Hessenberg-QR, has no complex-pair / balancing
// logic, and returns orthogonal eigenvectors also for the degenerate spectra of
// identical regions, where the general solver's eigenvector matrix is close to
// singular (that is the rounding noise of the reference documented in DESIGN.md).
//
// Everything below is register resident (fully unrolled, 3 <= N <= SS_UNROLL_MAX);
// the caller falls back to the general solver when C is not numerically positive
// definite or the symmetry check fails.
#pragma once

#include "ss_common.hpp"

namespace ss {

// Householder reduction of the symmetric matrix a (N x N, full storage, row i
// column j at a[i + j*N]; only the lower triangle is referenced) to tridiagonal
// form, accumulating the transformation: on return a = Q, d = diagonal,
// e[1..N-1] = sub-diagonal (e[0] = 0).  EISPACK tred2.
template <int N>
SS_HD void sym_tridiagonalise(double* a, double* d, double* e) {
#define A_(i, j) a[(i) + (j) * N]
  SS_SYM_UNROLL(N)
  for (int i = N - 1; i >= 1; --i) {
    const int l = i - 1;
    double h = 0.0;
    if (l > 0) {
      double scale = 0.0;
      SS_SYM_UNROLL(N)
      for (int k = 0; k <= l; ++k) scale += fabs(A_(i, k));
      if (scale == 0.0) {
        e[i] = A_(i, l);
      } else {
        const double rs = 1.0 / scale;
        SS_SYM_UNROLL(N)
        for (int k = 0; k <= l; ++k) {
          A_(i, k) *= rs;
          h += A_(i, k) * A_(i, k);
        }
        double f = A_(i, l);
        double g = -ss_sign(sqrt(h), f);
        e[i] = scale * g;
        h -= f * g;
        A_(i, l) = f - g;
        f = 0.0;
        const double rh = 1.0 / h;
        SS_SYM_UNROLL(N)
        for (int j = 0; j <= l; ++j) {
          A_(j, i) = A_(i, j) * rh;
          g = 0.0;
          SS_SYM_UNROLL(N)
          for (int k = 0; k <= j; ++k) g += A_(j, k) * A_(i, k);
          SS_SYM_UNROLL(N)
          for (int k = j + 1; k <= l; ++k) g += A_(k, j) * A_(i, k);
          e[j] = g * rh;
          f += e[j] * A_(i, j);
        }
        const double hh = f / (h + h);
        SS_SYM_UNROLL(N)
        for (int j = 0; j <= l; ++j) {
          f = A_(i, j);
          g = e[j] - hh * f;
          e[j] = g;
          SS_SYM_UNROLL(N)
          for (int k = 0; k <= j; ++k) A_(j, k) -= f * e[k] + g * A_(i, k);
        }
      }
    } else {
      e[i] = A_(i, l);
    }
    d[i] = h;
  }
  d[0] = 0.0;
  e[0] = 0.0;
  SS_SYM_UNROLL(N)
  for (int i = 0; i < N; ++i) {
    const int l = i - 1;
    if (d[i] != 0.0) {
      SS_SYM_UNROLL(N)
      for (int j = 0; j <= l; ++j) {
        double g = 0.0;
        SS_SYM_UNROLL(N)
        for (int k = 0; k <= l; ++k) g += A_(i, k) * A_(k, j);
        SS_SYM_UNROLL(N)
        for (int k = 0; k <= l; ++k) A_(k, j) -= g * A_(k, i);
      }
    }
    d[i] = A_(i, i);
    A_(i, i) = 1.0;
    SS_SYM_UNROLL(N)
    for (int j = 0; j <= l; ++j) {
      A_(j, i) = 0.0;
      A_(i, j) = 0.0;
    }
  }
#undef A_
}

// QL iteration with implicit shifts on the tridiagonal (d, e) with accumulation
// into z (= Q on entry, eigenvectors in columns on return).  EISPACK tql2; the
// run-time bounds (m, the rotation index) are predicates on static iterations.
// Returns 0, or 1 when an eigenvalue needed more than 30 iterations.
template <int N>
SS_HD int sym_tridiagonal_ql(double* d, double* e, double* z) {
#define Z_(i, j) z[(i) + (j) * N]
  int ierr = 0;
  SS_SYM_UNROLL(N)
  for (int i = 1; i < N; ++i) e[i - 1] = e[i];
  e[N - 1] = 0.0;
  SS_SYM_UNROLL(N)
  for (int l = 0; l < N; ++l) {
    int iter = 0;
    for (;;) {
      // first m >= l with a negligible sub-diagonal element e[m] (e[N-1] = 0)
      int m = N - 1;
      {
        bool found = false;
        SS_SYM_UNROLL(N)
        for (int mm = l; mm < N - 1; ++mm) {
          if (!found) {
            const double dd = fabs(d[mm]) + fabs(d[mm + 1]);
            if (fabs(e[mm]) <= DBL_EPSILON * dd) { m = mm; found = true; }
          }
        }
      }
      if (m == l) break;
      if (iter == 30) { ierr = 1; break; }
      ++iter;
      // shift
      double g = (d[l + 1 < N ? l + 1 : l] - d[l]) / (2.0 * e[l]);
      // (|g| < 1 / (2 eps) by the test above: g^2 + 1 is far from the ends of the exponent range, and so are
      // the sums of squares below unless they are exactly zero - the fast square root needs no scaling)
      double r, rr_;
      fast_sqrt_rsqrt(g * g + 1.0, r, rr_);
      // d[m] with run-time m
      double dm = 0.0;
      SS_SYM_UNROLL(N)
      for (int k = l; k < N; ++k) dm = (k == m) ? d[k] : dm;
      g = dm - d[l] + e[l] / (g + ss_sign(r, g));
      double s = 1.0, c = 1.0, p = 0.0;
      bool broke = false;
      SS_SYM_UNROLL(N)
      for (int i = N - 2; i >= l; --i) {
        if (i <= m - 1 && !broke) {
          double f = s * e[i];
          const double b = c * e[i];
          const double h2 = f * f + g * g;
          double rr;
          fast_sqrt_rsqrt(h2 > 0.0 ? h2 : 1.0, r, rr);
          r = h2 > 0.0 ? r : 0.0;
          e[i + 1] = r;
          if (r == 0.0) {
            d[i + 1] -= p;
            // e[m] = 0
            SS_SYM_UNROLL(N)
            for (int k = l; k < N; ++k) e[k] = (k == m) ? 0.0 : e[k];
            broke = true;
          } else {
            s = f * rr;
            c = g * rr;
            g = d[i + 1] - p;
            r = (d[i] - g) * s + 2.0 * c * b;
            p = s * r;
            d[i + 1] = g + p;
            g = c * r - b;
            SS_SYM_UNROLL(N)
            for (int k = 0; k < N; ++k) {
              f = Z_(k, i + 1);
              Z_(k, i + 1) = s * Z_(k, i) + c * f;
              Z_(k, i) = c * Z_(k, i) - s * f;
            }
          }
        }
      }
      if (!broke) {
        d[l] -= p;
        e[l] = g;
        SS_SYM_UNROLL(N)
        for (int k = l; k < N; ++k) e[k] = (k == m) ? 0.0 : e[k];
      }
    }
  }
#undef Z_
  return ierr;
}

// Eigenvalues and eigenvectors of P = A (A + 2 G2) where, with the scaling d,
// D^-1 A D = -C (C symmetric positive definite) and D^-1 G2 D symmetric.
// Returns 0 on success; -1 when the structure is not there (the caller then
// uses the general solver); > 0 when the QL iteration did not converge.
// ainv_x (N x N) receives A^-1 X = -D L^-T Y, which the caller needs next and
// which costs one triangular solve here instead of an LU factorisation of A.
template <int N>
SS_HD int eigen_decomposition_gamma_sym(const double* A, const double* G2, const double* dscale,
                                        double* eigenvalue, double* eigenvector, double* ainv_x) {
  double rd[N];
  SS_SYM_UNROLL(N)
  for (int i = 0; i < N; ++i) {
    if (!(dscale[i] > 0.0)) return -1;
    rd[i] = 1.0 / dscale[i];
  }
  // lower triangles of C = -D^-1 A D and M = C - 2 D^-1 G2 D (upper ones mirrored
  // from the lower: the rounding-level asymmetry is dropped), and the symmetry check
  double C[N * N], M[N * N];
  double asym = 0.0, size = 0.0;
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) {
    SS_SYM_UNROLL(N)
    for (int i = j; i < N; ++i) {
      const double c_ij = -SS_E(A, i, j, N) * (dscale[j] * rd[i]);
      const double s_ij = SS_E(G2, i, j, N) * (dscale[j] * rd[i]);
      if (i != j) {
        const double c_ji = -SS_E(A, j, i, N) * (dscale[i] * rd[j]);
        const double s_ji = SS_E(G2, j, i, N) * (dscale[i] * rd[j]);
        asym = ss_max(asym, ss_max(fabs(c_ij - c_ji), fabs(s_ij - s_ji)));
      }
      size = ss_max(size, ss_max(fabs(c_ij), fabs(s_ij)));
      SS_E(C, i, j, N) = c_ij;
      SS_E(C, j, i, N) = c_ij;
      SS_E(M, i, j, N) = c_ij - 2.0 * s_ij;
      SS_E(M, j, i, N) = c_ij - 2.0 * s_ij;
    }
  }
  if (!(asym <= 1.0e-10 * size)) return -1;
  // Cholesky C = L L^T (L in the lower triangle of C, reciprocal diagonal in rl)
  double rl[N];
  bool ok = true;
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) {
    double s = SS_E(C, j, j, N);
    SS_SYM_UNROLL(N)
    for (int k = 0; k < j; ++k) s -= SS_E(C, j, k, N) * SS_E(C, j, k, N);
    if (!(s > 1.0e-14 * size)) ok = false;
    const double ljj = sqrt(ok ? s : 1.0);
    SS_E(C, j, j, N) = ljj;
    rl[j] = 1.0 / ljj;
    SS_SYM_UNROLL(N)
    for (int i = j + 1; i < N; ++i) {
      double t = SS_E(C, i, j, N);
      SS_SYM_UNROLL(N)
      for (int k = 0; k < j; ++k) t -= SS_E(C, i, k, N) * SS_E(C, j, k, N);
      SS_E(C, i, j, N) = t * rl[j];
    }
  }
  if (!ok) return -1;
  // K = L^T M L  (W = M L, then K = L^T W; K symmetric, lower triangle computed)
  double W[N * N], K[N * N];
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) {
    SS_SYM_UNROLL(N)
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
      SS_SYM_UNROLL(N)
      for (int k = j; k < N; ++k) s += SS_E(M, i, k, N) * SS_E(C, k, j, N);
      SS_E(W, i, j, N) = s;
    }
  }
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) {
    SS_SYM_UNROLL(N)
    for (int i = j; i < N; ++i) {
      double s = 0.0;
      SS_SYM_UNROLL(N)
      for (int k = i; k < N; ++k) s += SS_E(C, k, i, N) * SS_E(W, k, j, N);
      SS_E(K, i, j, N) = s;
      SS_E(K, j, i, N) = s;
    }
  }
  double e[N];
  sym_tridiagonalise<N>(K, eigenvalue, e);
  const int ierr = sym_tridiagonal_ql<N>(eigenvalue, e, K);
  // eigenvectors of P: X = D L Y;   A^-1 X = -D L^-T Y (back-substitution with L^T)
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) {
    SS_SYM_UNROLL(N)
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
      SS_SYM_UNROLL(N)
      for (int k = 0; k <= i; ++k) s += SS_E(C, i, k, N) * SS_E(K, k, j, N);
      SS_E(eigenvector, i, j, N) = dscale[i] * s;
    }
    double v[N];
    SS_SYM_UNROLL(N)
    for (int i = N - 1; i >= 0; --i) {
      double s = SS_E(K, i, j, N);
      SS_SYM_UNROLL(N)
      for (int k = i + 1; k < N; ++k) s -= SS_E(C, k, i, N) * v[k];
      v[i] = s * rl[i];
      SS_E(ainv_x, i, j, N) = -dscale[i] * v[i];
    }
  }
  return ierr;
}

// The symmetric QL iteration delivers eigenvalues with an ABSOLUTE accuracy of
// eps * (largest eigenvalue).  Optically thin regions next to dense ones (clear
// air of extinction 1e-5 beside vegetation) give spectra that span ten or more
// decades, and the longwave emission of the thin region is proportional to its
// (tiny) eigenvalue: such spectra go to the general solver, whose balancing keeps
// the relative accuracy of the small eigenvalues.  1e-6 leaves every accepted
// eigenvalue a relative accuracy of ~1e-10; the longwave balance of a thin region
// amplifies an eigenvalue error a few hundred times (measured: 5e-4 -> 14 % of a
// flux), so 1e-8 (all parity tests still pass, 2 % faster) would leave no margin.
template <int N>
SS_HD bool sym_spectrum_is_well_graded(const double* ev) {
  double lo = fabs(ev[0]), hi = fabs(ev[0]);
  SS_SYM_UNROLL(N)
  for (int i = 1; i < N; ++i) {
    lo = ss_min(lo, fabs(ev[i]));
    hi = ss_max(hi, fabs(ev[i]));
  }
#ifndef SS_SYM_GRADING
#define SS_SYM_GRADING 1.0e-6
#endif
  return lo > SS_SYM_GRADING * hi;
}

// Eigen-decomposition of a matrix G that the scaling d makes symmetric
// (D^-1 G D = K symmetric; the direct-beam exchange matrix Gamma0 with
// d = sqrt(region fraction)).  Same return convention as above.
template <int N>
SS_HD int eigen_decomposition_scaled_sym(const double* G, const double* dscale, double* eigenvalue,
                                         double* eigenvector) {
  double rd[N];
  SS_SYM_UNROLL(N)
  for (int i = 0; i < N; ++i) {
    if (!(dscale[i] > 0.0)) return -1;
    rd[i] = 1.0 / dscale[i];
  }
  double K[N * N];
  double asym = 0.0, size = 0.0;
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) {
    SS_SYM_UNROLL(N)
    for (int i = j; i < N; ++i) {
      const double k_ij = SS_E(G, i, j, N) * (dscale[j] * rd[i]);
      if (i != j) asym = ss_max(asym, fabs(k_ij - SS_E(G, j, i, N) * (dscale[i] * rd[j])));
      size = ss_max(size, fabs(k_ij));
      SS_E(K, i, j, N) = k_ij;
      SS_E(K, j, i, N) = k_ij;
    }
  }
  if (!(asym <= 1.0e-10 * size)) return -1;
  double e[N];
  sym_tridiagonalise<N>(K, eigenvalue, e);
  const int ierr = sym_tridiagonal_ql<N>(eigenvalue, e, K);
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) {
    SS_SYM_UNROLL(N)
    for (int i = 0; i < N; ++i) SS_E(eigenvector, i, j, N) = dscale[i] * SS_E(K, i, j, N);
  }
  return ierr;
}

}  // namespace ss
