// ss_linalg.hpp - per-lane small dense linear algebra (one matrix per lane).
//
// Covers the live subset of the reference's radtool_matrix (SURVEY.md section
// 2.1): products, unpivoted LU, triangular substitution, inverse.  Matrices are
// column-major with compile-time shapes; for orders <= SS_UNROLL_MAX every
// loop is fully unrolled, so each element is a named register and an n x n
// product is n^3 back-to-back v_fma_f64 with n^2-way ILP.
//
// Differences from the reference that change results only at rounding level:
//  * the diagonal of U is inverted once per factorisation and substitution
//    multiplies by it (the reference divides per right-hand side,
//    radtool_matrix.F90:1047) - fp64 division costs ~15 VALU ops on gfx950;
//  * one LU factorisation is reused for every solve with the same matrix
//    (the reference refactors, e.g. radsurf_forest_sw.F90:485,490,575,703);
//  * 3x3 systems use the generic unpivoted LU (the reference hard-codes the
//    same elimination, radtool_matrix.F90:827-906); 2x2 systems use the
//    adjugate (Cramer, as :779-821).
#pragma once

#include "ss_common.hpp"
#include "ss_linalg_blocked.hpp"

namespace ss {

// orders above SS_UNROLL_MAX go to the register-tiled routines of ss_linalg_blocked.hpp
template <int A, int B = 0, int C = 0>
constexpr bool kBigOrder = (A > SS_UNROLL_MAX) || (B > SS_UNROLL_MAX) || (C > SS_UNROLL_MAX);

template <int N>
SS_HD void vec_zero(double* x) {
  SS_UNROLL_N(N)
  for (int i = 0; i < N; ++i) x[i] = 0.0;
}

template <int N>
SS_HD void vec_copy(const double* a, double* b) {
  SS_UNROLL_N(N)
  for (int i = 0; i < N; ++i) b[i] = a[i];
}
template <int N>
SS_HD double vec_sum(const double* a) {
  double s = 0.0;
  SS_UNROLL_FULL
  for (int i = 0; i < N; ++i) s += a[i];
  return s;
}

template <int R, int C>
SS_HD void mat_zero(double* A) {
  SS_UNROLL_N(C)
  for (int j = 0; j < C; ++j) {
    SS_UNROLL_N(R)
    for (int i = 0; i < R; ++i) SS_E(A, i, j, R) = 0.0;
  }
}

template <int R, int C>
SS_HD void mat_copy(const double* A, double* B) {
  SS_UNROLL_N(C)
  for (int j = 0; j < C; ++j) {
    SS_UNROLL_N(R)
    for (int i = 0; i < R; ++i) SS_E(B, i, j, R) = SS_E(A, i, j, R);
  }
}

// C(M x N) = A(M x K) * B(K x N)   (mat_x_mat :248, rect_mat_x_mat :335)
template <int M, int K, int N>
SS_HD void mat_mul(const double* A, const double* B, double* C) {
  if constexpr (kBigOrder<M, K, N>) {
    blk::mat_mul<M, K, N, false, false>(A, nullptr, B, C);
    return;
  }
  SS_UNROLL_N(N)
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_N(M)
    for (int i = 0; i < M; ++i) {
      double s = 0.0;
      SS_UNROLL_N(K)
      for (int k = 0; k < K; ++k) s += SS_E(A, i, k, M) * SS_E(B, k, j, K);
      SS_E(C, i, j, M) = s;
    }
  }
}

// C += A*B
template <int M, int K, int N>
SS_HD void mat_mul_acc(const double* A, const double* B, double* C) {
  if constexpr (kBigOrder<M, K, N>) {
    blk::mat_mul<M, K, N, true, false>(A, nullptr, B, C);
    return;
  }
  SS_UNROLL_N(N)
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_N(M)
    for (int i = 0; i < M; ++i) {
      double s = SS_E(C, i, j, M);
      SS_UNROLL_N(K)
      for (int k = 0; k < K; ++k) s += SS_E(A, i, k, M) * SS_E(B, k, j, K);
      SS_E(C, i, j, M) = s;
    }
  }
}

// C (+)= A diag(e) B without forming A diag(e) or diag(e) B
template <int M, int K, int N, bool ACC>
SS_HD void mat_mul_rs(const double* A, const double* e, const double* B, double* C) {
  if constexpr (kBigOrder<M, K, N>) {
    blk::mat_mul<M, K, N, ACC, true>(A, e, B, C);
    return;
  }
  SS_UNROLL_N(N)
  for (int j = 0; j < N; ++j) {
    double eb[K];
    SS_UNROLL_N(K)
    for (int k = 0; k < K; ++k) eb[k] = e[k] * SS_E(B, k, j, K);
    SS_UNROLL_N(M)
    for (int i = 0; i < M; ++i) {
      double s = ACC ? SS_E(C, i, j, M) : 0.0;
      SS_UNROLL_N(K)
      for (int k = 0; k < K; ++k) s += SS_E(A, i, k, M) * eb[k];
      SS_E(C, i, j, M) = s;
    }
  }
}

// y(M) = A(M x K) x(K)   (mat_x_vec :71, rect_mat_x_vec :125)
template <int M, int K>
SS_HD void mat_vec(const double* A, const double* x, double* y) {
  SS_UNROLL_N(M)
  for (int i = 0; i < M; ++i) {
    double s = 0.0;
    SS_UNROLL_N(K)
    for (int k = 0; k < K; ++k) s += SS_E(A, i, k, M) * x[k];
    y[i] = s;
  }
}

// y += A x
template <int M, int K>
SS_HD void mat_vec_acc(const double* A, const double* x, double* y) {
  SS_UNROLL_N(M)
  for (int i = 0; i < M; ++i) {
    double s = y[i];
    SS_UNROLL_N(K)
    for (int k = 0; k < K; ++k) s += SS_E(A, i, k, M) * x[k];
    y[i] = s;
  }
}

// In-place unpivoted LU (lu_factorization :982-1017, Crout ordering) with the
// reciprocals of U's diagonal stored ON the diagonal.  N = 1 and N = 2 are
// closed forms: the "factor" of a 2x2 matrix is its inverse (adjugate / det).
template <int N>
SS_HD void lu_factor(double* A) {
  if (N == 1) {
    A[0] = 1.0 / A[0];
    return;
  }
  if (N == 2) {
    const double a = A[0], c = A[1], b = A[2], d = A[3];
    const double inv_det = 1.0 / (a * d - b * c);
    A[0] = inv_det * d;
    A[1] = -inv_det * c;
    A[2] = -inv_det * b;
    A[3] = inv_det * a;
    return;
  }
  if constexpr (kBigOrder<N>) {
    blk::lu_factor<N>(A);
    return;
  }
  SS_UNROLL_N(N)
  for (int j2 = 0; j2 < N; ++j2) {
    SS_UNROLL_N(N)
    for (int j1 = 0; j1 < N; ++j1) {
      double s = SS_E(A, j1, j2, N);
      const int kmax = (j1 < j2) ? j1 : j2;
      SS_UNROLL_N(N)
      for (int j3 = 0; j3 < kmax; ++j3) s -= SS_E(A, j1, j3, N) * SS_E(A, j3, j2, N);
      SS_E(A, j1, j2, N) = s;
    }
    const double r = 1.0 / SS_E(A, j2, j2, N);
    SS_E(A, j2, j2, N) = r;
    SS_UNROLL_N(N)
    for (int j1 = j2 + 1; j1 < N; ++j1) SS_E(A, j1, j2, N) *= r;
  }
}

// Solve A x = b in place given lu_factor(A)  (lu_substitution :1024-1050)
template <int N>
SS_HD void lu_solve(const double* LU, double* x) {
  if (N == 1) {
    x[0] *= LU[0];
    return;
  }
  if (N == 2) {
    const double x0 = LU[0] * x[0] + LU[2] * x[1];
    const double x1 = LU[1] * x[0] + LU[3] * x[1];
    x[0] = x0;
    x[1] = x1;
    return;
  }
  SS_UNROLL_N(N)
  for (int j2 = 1; j2 < N; ++j2) {
    double s = x[j2];
    SS_UNROLL_N(N)
    for (int j1 = 0; j1 < j2; ++j1) s -= x[j1] * SS_E(LU, j2, j1, N);
    x[j2] = s;
  }
  SS_UNROLL_N(N)
  for (int jj = 0; jj < N; ++jj) {
    const int j2 = N - 1 - jj;
    double s = x[j2];
    SS_UNROLL_N(N)
    for (int j1 = j2 + 1; j1 < N; ++j1) s -= x[j1] * SS_E(LU, j2, j1, N);
    x[j2] = s * SS_E(LU, j2, j2, N);
  }
}

// Solve A X = B in place, B is N x M  (solve_mat :1175, solve_rect_mat :1119)
template <int N, int M>
SS_HD void lu_solve_mat(const double* LU, double* B) {
  if constexpr (kBigOrder<N> && M >= 2) {
    blk::lu_solve_mat<N, M>(LU, B);
    return;
  }
  SS_UNROLL_N(M)
  for (int j = 0; j < M; ++j) lu_solve<N>(LU, B + j * N);
}

// X = A^-1 given lu_factor(A)  (lu_invert :1057-1090)
template <int N>
SS_HD void lu_invert(const double* LU, double* X) {
  if (N <= 2) {
    SS_UNROLL_FULL
    for (int k = 0; k < N * N; ++k) X[k] = LU[k];
    return;
  }
  if constexpr (kBigOrder<N>) {
    for (int k = 0; k < N * N; ++k) X[k] = 0.0;
    for (int k = 0; k < N; ++k) X[k + k * N] = 1.0;
    blk::lu_solve_mat<N, N>(LU, X);
    return;
  }
  SS_UNROLL_N(N)
  for (int j3 = 0; j3 < N; ++j3) {
    double* x = X + j3 * N;
    SS_UNROLL_N(N)
    for (int j2 = 0; j2 < N; ++j2) x[j2] = (j2 == j3) ? 1.0 : 0.0;
    lu_solve<N>(LU, x);
  }
}

}  // namespace ss
