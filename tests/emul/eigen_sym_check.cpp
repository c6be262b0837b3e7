// TEST-ONLY: the symmetric route (ss_eigen_sym.hpp) against the general solver
// (ss_eigen.hpp) on random Gamma-like matrices with the hidden symmetry.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <algorithm>
#include "../../spartacus_surface_amd/csrc/ss_eigen.hpp"
#include "../../spartacus_surface_amd/csrc/ss_eigen_sym.hpp"

template <int N>
int check(int ncase, unsigned seed) {
  int bad = 0;
  double worst_val = 0, worst_res = 0;
  srand(seed);
  auto rnd = []() { return (double)rand() / RAND_MAX; };
  for (int c = 0; c < ncase; ++c) {
    // symmetric C (SPD, diagonally dominant), symmetric S >= 0, scaling d
    double C[N * N], S[N * N], d[N];
    for (int i = 0; i < N; ++i) d[i] = 0.05 + rnd();
    for (int i = 0; i < N; ++i) for (int j = 0; j <= i; ++j) {
      double c_ij = (i == j) ? 0.0 : -0.2 * rnd() * ((c % 3 == 0 && ((i < N / 2) != (j < N / 2))) ? 0.0 : 1.0);
      double s_ij = 0.15 * rnd();
      C[i + j * N] = C[j + i * N] = c_ij;
      S[i + j * N] = S[j + i * N] = s_ij;
    }
    for (int i = 0; i < N; ++i) {
      double s = 0;
      for (int j = 0; j < N; ++j) if (j != i) s += fabs(C[i + j * N]) + 2 * fabs(S[i + j * N]);
      C[i + i * N] = s + 2 * S[i + i * N] + 0.05 + rnd();
      if (c % 5 == 4) C[i + i * N] = s + 2 * S[i + i * N] + 0.3;   // near-degenerate diagonal
    }
    double A[N * N], G2[N * N], P[N * N];
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) {
      A[i + j * N] = -C[i + j * N] * d[i] / d[j];
      G2[i + j * N] = S[i + j * N] * d[i] / d[j];
    }
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) {
      double s = 0;
      for (int k = 0; k < N; ++k) s += A[i + k * N] * (A[k + j * N] + 2 * G2[k + j * N]);
      P[i + j * N] = s;
    }
    double v1[N], x1[N * N], v2[N], x2[N * N], ax[N * N];
    ss::eigen_decomposition_real_dynamic<N>(P, v1, x1);
    int rc = ss::eigen_decomposition_gamma_sym<N>(A, G2, d, v2, x2, ax);
    if (rc != 0) { if (bad < 3) printf("N=%d case %d rc %d\n", N, c, rc); ++bad; continue; }
    double s1[N], s2[N];
    for (int i = 0; i < N; ++i) { s1[i] = v1[i]; s2[i] = v2[i]; }
    std::sort(s1, s1 + N); std::sort(s2, s2 + N);
    double scale = fabs(s1[N - 1]);
    for (int i = 0; i < N; ++i) worst_val = std::max(worst_val, fabs(s1[i] - s2[i]) / scale);
    // residual || P X - X Lambda || / (||P|| ||X||) column by column
    for (int j = 0; j < N; ++j) {
      double rn = 0, xn = 0;
      for (int i = 0; i < N; ++i) {
        double s = 0;
        for (int k = 0; k < N; ++k) s += P[i + k * N] * x2[k + j * N];
        rn = std::max(rn, fabs(s - v2[j] * x2[i + j * N]));
        xn = std::max(xn, fabs(x2[i + j * N]));
      }
      worst_res = std::max(worst_res, rn / (scale * xn));
      // A (A^-1 X) = X
      double an = 0;
      for (int i = 0; i < N; ++i) {
        double s = 0;
        for (int k = 0; k < N; ++k) s += A[i + k * N] * ax[k + j * N];
        an = std::max(an, fabs(s - x2[i + j * N]));
      }
      worst_res = std::max(worst_res, an / xn * 1e-3);   // (conditioning of A up to ~1e3 here)
    }
  }
  printf("N=%d: %d cases, %d failures, worst eigenvalue diff %.2e, worst residual %.2e\n", N, ncase, bad,
         worst_val, worst_res);
  return bad + (worst_val > 1e-11) + (worst_res > 1e-12);
}

int main() {
  int bad = 0;
  bad += check<3>(3000, 1);
  bad += check<4>(3000, 2);
  bad += check<5>(3000, 3);
  bad += check<6>(3000, 4);
  return bad ? 1 : 0;
}
