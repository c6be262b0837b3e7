// ss_eigen_coop.hpp - cooperative eigen-decomposition for matrix orders that do not fit in
// the registers of one lane (7 <= n <= 16; n = 12 is 3 regions x 4 streams, test/rami5).
//
// What is solved.  The layer matrices need the eigen-decomposition of
//   P = (G1 - G2)(G1 + G2)            (radtool_calc_matrices_sw_eig.F90:180-188, _lw_eig :142-148),
// which the reference hands to its general solver (radtool_eigen_decomposition.F90:51-828).
// With the scaling D of ss_eigen_sym.hpp, D^-1 (G1 - G2) D = -C and D^-1 (G1 + G2) D = -M with
// C and M symmetric positive definite, so P = D C M D^-1: a PRODUCT of two s.p.d. matrices.
// With the Cholesky factors C = L L^T, M = F F^T
//   L^-1 (C M) L = L^T M L = (F^T L)^T (F^T L) = W^T W,
// i.e. the eigenvalues of P are the squared singular values of W = F^T L and its eigenvectors
// are X = D L Y with Y the right singular vectors of W.  The layer matrices use
// lambda = sqrt(eigenvalue) = the singular values themselves.
//
// How.  One-sided (Hestenes) Jacobi on W: rotate pairs of COLUMNS of W (and of Y, which starts
// as the identity) until all columns are orthogonal; their norms are the singular values.
//  * It delivers every singular value to high RELATIVE accuracy - optically thin regions next
//    to dense ones give spectra that span ten decades, which is why the symmetric QL route of
//    the register-resident orders has to fall back to the general solver for 17 % of its units
//    (sym_spectrum_is_well_graded); here nothing falls back.
//  * It only ever combines columns: with the ROWS of W and Y dealt out to the lanes of a group
//    (row i on lane i mod G) every rotation is lane-local; the three inner products that
//    define a rotation are a local part plus a butterfly over the G lanes of the group.
//  * So a 12 x 12 problem lives in the registers of 4 lanes (3 rows of W and of Y each: 144
//    VGPRs) instead of 17 KB of scratch memory per lane, 16 problems per wave.
// The kernel (k_eig_coop, ss_inst_body.hpp) sits between two passes of the layer kernel: the
// first builds W (coop_eig_pre, one lane per problem), the second takes eigenvalues and
// vectors from the record (coop_eig_post) where the one-kernel form calls the eigen-solver.
//
// Group size G = 4 (a DPP quad).  Orders above 16 (8 streams x 3 regions) keep the one-kernel
// form: they would need 8 lanes per problem and twice the unrolled code.
#pragma once

#include "ss_common.hpp"
#include "ss_layer_matrices.hpp"

namespace ss {

constexpr int kCoopLanes = 4;
constexpr int kCoopMaxOrder = 16;
constexpr bool coop_order(int n) {
#if defined(SS_NO_COOP_EIGEN)
  return false && n > 0;
#else
  return n > SS_UNROLL_MAX && n <= kCoopMaxOrder;
#endif
}

// Eigen record of one (column, band, layer) unit in the tiling of the scratch records:
//   [0] status (1 = W valid / eigen-decomposition valid, 0 = not a cooperative unit), [1] unused,
//   [EV .. EV + N) squared singular values, [MAT .. MAT + N*N) W, later Y, ROW-major
template <int N>
struct EigRec {
  static constexpr int STATUS = 0;
  static constexpr int EV = 2;
  static constexpr int MAT = EV + rec_even(N);
  static constexpr int NELEM = MAT + rec_even(N * N);
};

// ---- pass 1 of the layer kernel: W = F^T L from the Gamma matrices (one lane per problem) ----
// A = G1 - G2, g2 = G2, dscale = D.  Returns false (status 0) when the structure is not there
// numerically (not symmetric after scaling, not positive definite): the second pass then uses
// the general solver for this unit.
template <int N>
SS_HD bool coop_eig_pre(const double* A, const double* G2, const double* dscale, double* rec, long stride) {
  double C[N * N], M[N * N], rd[N];
  bool ok = true;
  for (int i = 0; i < N; ++i) {
    if (!(dscale[i] > 0.0)) ok = false;
    rd[i] = 1.0 / (ok ? dscale[i] : 1.0);
  }
  double asym = 0.0, size = 0.0;
  for (int j = 0; j < N; ++j) {
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
  if (!(asym <= 1.0e-10 * size)) ok = false;
  // Cholesky factors in the lower triangles: C = L L^T, M = F F^T
  auto chol = [&](double* S) {
    for (int j = 0; j < N; ++j) {
      double s = SS_E(S, j, j, N);
      for (int k = 0; k < j; ++k) s -= SS_E(S, j, k, N) * SS_E(S, j, k, N);
      if (!(s > 1.0e-14 * size)) ok = false;
      const double ljj = sqrt(ok ? s : 1.0);
      SS_E(S, j, j, N) = ljj;
      const double r = 1.0 / ljj;
      for (int i = j + 1; i < N; ++i) {
        double t = SS_E(S, i, j, N);
        for (int k = 0; k < j; ++k) t -= SS_E(S, i, k, N) * SS_E(S, j, k, N);
        SS_E(S, i, j, N) = t * r;
      }
    }
  };
  chol(C);
  chol(M);
  rec_pair_store(rec + (long)EigRec<N>::STATUS * stride, ok ? 1.0 : 0.0, 0.0);
  if (!ok) return false;
  // W = F^T L: W(i, j) = sum_{k >= max(i, j)} F(k, i) L(k, j); stored row-major, pairs along a row
  for (int i = 0; i < N; ++i) {
    for (int j = 0; j < N; j += 2) {
      double w0 = 0.0, w1 = 0.0;
      for (int k = (i > j ? i : j); k < N; ++k) w0 += SS_E(M, k, i, N) * SS_E(C, k, j, N);
      if (j + 1 < N)
        for (int k = (i > j + 1 ? i : j + 1); k < N; ++k) w1 += SS_E(M, k, i, N) * SS_E(C, k, j + 1, N);
      if (((i * N + j) & 1) == 0) {
        rec_pair_store(rec + (long)(EigRec<N>::MAT + i * N + j) * stride, w0, w1);
      } else {
        // (odd N: a row does not start at an even element - element-wise through the pair it shares)
        double* p = rec + (long)(EigRec<N>::MAT + i * N + j - 1) * stride;
        p[1] = w0;
        if (j + 1 < N) (rec + (long)(EigRec<N>::MAT + i * N + j + 1) * stride)[0] = w1;
      }
    }
  }
  return true;
}

// element e of a record, read as a scalar (pairs: element e sits at pair e/2, half e%2)
SS_HD double rec_elem_load(const double* rec, long stride, int e) {
  return rec[(long)(e & ~1) * stride + (e & 1)];
}

// ---- pass 2 of the layer kernel: eigenvalues / vectors from the record ----
// Same outputs as eigen_decomposition_gamma_sym: eigenvalue, eigenvector X = D L Y, ainv_x =
// A^-1 X = -D L^-T Y.  Returns -1 when the record says the unit is not a cooperative one.
template <int N>
SS_HD int coop_eig_post(const double* A, const double* dscale, const double* rec, long stride,
                        double* eigenvalue, double* eigenvector, double* ainv_x) {
  if (!(rec_elem_load(rec, stride, EigRec<N>::STATUS) > 0.5)) return -1;
  // L again (a third of an n^3 - cheaper than carrying it through memory)
  double C[N * N], rd[N], rl[N];
  for (int i = 0; i < N; ++i) rd[i] = 1.0 / dscale[i];
  for (int j = 0; j < N; ++j)
    for (int i = j; i < N; ++i) SS_E(C, i, j, N) = -SS_E(A, i, j, N) * (dscale[j] * rd[i]);
  for (int j = 0; j < N; ++j) {
    double s = SS_E(C, j, j, N);
    for (int k = 0; k < j; ++k) s -= SS_E(C, j, k, N) * SS_E(C, j, k, N);
    const double ljj = sqrt(s);
    SS_E(C, j, j, N) = ljj;
    rl[j] = 1.0 / ljj;
    for (int i = j + 1; i < N; ++i) {
      double t = SS_E(C, i, j, N);
      for (int k = 0; k < j; ++k) t -= SS_E(C, i, k, N) * SS_E(C, j, k, N);
      SS_E(C, i, j, N) = t * rl[j];
    }
  }
  for (int j = 0; j < N; ++j) eigenvalue[j] = rec_elem_load(rec, stride, EigRec<N>::EV + j);
  for (int j = 0; j < N; ++j) {
    double y[N], v[N];
    for (int i = 0; i < N; ++i) y[i] = rec_elem_load(rec, stride, EigRec<N>::MAT + i * N + j);   // Y(i, j), row-major
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
      for (int k = 0; k <= i; ++k) s += SS_E(C, i, k, N) * y[k];
      SS_E(eigenvector, i, j, N) = dscale[i] * s;
    }
    for (int i = N - 1; i >= 0; --i) {
      double s = y[i];
      for (int k = i + 1; k < N; ++k) s -= SS_E(C, k, i, N) * v[k];
      v[i] = s * rl[i];
      SS_E(ainv_x, i, j, N) = -dscale[i] * v[i];
    }
  }
  return 0;
}

// ---- the cooperative kernel body ----
#if defined(__HIP_DEVICE_COMPILE__)
// sum over the G lanes of a group (butterfly: every lane ends with the same bits)
template <int G>
__device__ __forceinline__ double group_sum(double x) {
#pragma unroll
  for (int m = 1; m < G; m <<= 1) x += __shfl_xor(x, m);
  return x;
}
#endif

// One group of G lanes = one problem; "q" = lane within the group; rows q, q + G, ... of W and Y.
// rec = eigen record of the problem (all lanes of the group pass the same).  Host builds run the
// same arithmetic with G = 1 (tests/emul), there being no lanes to share with.
template <int N, int G>
SS_HD void coop_eig_unit(double* rec, long stride, int q) {
  constexpr int NR = (N + G - 1) / G;
  using Rec = EigRec<N>;
  if (!(rec_elem_load(rec, stride, Rec::STATUS) > 0.5)) return;   // same for the whole group
  double w[NR][N], y[NR][N];
  SS_UNROLL_FULL
  for (int r = 0; r < NR; ++r) {
    const int i = q + r * G;
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      w[r][j] = (i < N) ? rec_elem_load(rec, stride, Rec::MAT + i * N + j) : 0.0;
      y[r][j] = (i == j) ? 1.0 : 0.0;
    }
  }
  auto gsum = [](double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return group_sum<G>(x);
#else
    return x;
#endif
  };
  // cyclic sweeps; a sweep without a rotation ends the iteration (the decision is taken on
  // group-wide sums, so the lanes of a group agree on it)
  constexpr double tol = 4.0 * DBL_EPSILON;
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool rotated = false;
    SS_UNROLL_FULL
    for (int p = 0; p < N - 1; ++p) {
      SS_UNROLL_FULL
      for (int c = p + 1; c < N; ++c) {
        double app = 0.0, acc = 0.0, apc = 0.0;
        SS_UNROLL_FULL
        for (int r = 0; r < NR; ++r) {
          app += w[r][p] * w[r][p];
          acc += w[r][c] * w[r][c];
          apc += w[r][p] * w[r][c];
        }
        app = gsum(app);
        acc = gsum(acc);
        apc = gsum(apc);
        if (apc * apc > (tol * tol) * (app * acc)) {
          rotated = true;
          // Rutishauser's formulas: the smaller rotation angle
          const double zeta = (acc - app) / (2.0 * apc);
          const double t = ss_sign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
          SS_UNROLL_FULL
          for (int r = 0; r < NR; ++r) {
            const double wp = w[r][p], wc = w[r][c], yp = y[r][p], yc = y[r][c];
            w[r][p] = cs * wp - sn * wc;
            w[r][c] = sn * wp + cs * wc;
            y[r][p] = cs * yp - sn * yc;
            y[r][c] = sn * yp + cs * yc;
          }
        }
      }
    }
    if (!rotated) break;
  }
  // squared singular values, and Y in place of W
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    double s = 0.0;
    SS_UNROLL_FULL
    for (int r = 0; r < NR; ++r) s += w[r][j] * w[r][j];
    s = gsum(s);
    if (q == (j % G)) (rec + (long)((Rec::EV + j) & ~1) * stride)[(Rec::EV + j) & 1] = s;
  }
  SS_UNROLL_FULL
  for (int r = 0; r < NR; ++r) {
    const int i = q + r * G;
    if (i < N) {
      SS_UNROLL_FULL
      for (int j = 0; j < N; ++j) {
        const int e = Rec::MAT + i * N + j;
        (rec + (long)(e & ~1) * stride)[e & 1] = y[r][j];
      }
    }
  }
}

}  // namespace ss
