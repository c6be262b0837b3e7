// TEST-ONLY: the register-resident eigen routine must be bit-identical to the
// run-time-indexed one (same arithmetic, different addressing).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "../../spartacus_surface_amd/csrc/ss_eigen.hpp"

template <int N>
int check(int ncase, unsigned seed) {
  int bad = 0;
  srand(seed);
  for (int c = 0; c < ncase; ++c) {
    double a[N * N], v1[N], v2[N], x1[N * N], x2[N * N];
    // gamma-like matrices: product of two diagonally dominant matrices; some
    // cases block diagonal / triangular / with zero rows to exercise isolation
    double g1[N * N], g2[N * N];
    for (int k = 0; k < N * N; ++k) { g1[k] = 0.05 * rand() / RAND_MAX; g2[k] = 0.03 * rand() / RAND_MAX; }
    for (int i = 0; i < N; ++i) g1[i + i * N] = -(0.3 + 2.0 * rand() / RAND_MAX);
    int mode = c % 8;
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) {
      if (mode == 1 && ((i < N / 2) != (j < N / 2))) { g1[i + j * N] = 0; g2[i + j * N] = 0; }
      if (mode == 2 && i > j) { g1[i + j * N] = 0; g2[i + j * N] = 0; }
      if (mode == 3 && i != j) { g1[i + j * N] = 0; g2[i + j * N] = 0; }
      if (mode == 4 && i < j) { g1[i + j * N] = 0; g2[i + j * N] = 0; }
      if (mode == 5 && (i == 1 || j == N - 1) && i != j) { g1[i + j * N] = 0; g2[i + j * N] = 0; }
    }
    if (mode == 6) for (int i = 0; i < N; ++i) g1[i + i * N] = -1.0;   // degenerate diagonal
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) {
      double s = 0;
      for (int k = 0; k < N; ++k) s += (g1[i + k * N] - g2[i + k * N]) * (g1[k + j * N] + g2[k + j * N]);
      a[i + j * N] = s;
    }
    if (mode == 7) for (int k = 0; k < N * N; ++k) a[k] = g1[k];   // direct gamma0-like
    memset(x1, 0, sizeof x1); memset(x2, 0, sizeof x2);
    int e1 = ss::eigen_decomposition_real_dynamic<N>(a, v1, x1);
    int e2 = ss::eigen_decomposition_real_static<N>(a, v2, x2);
    if (e1 != e2 || memcmp(v1, v2, sizeof v1) || memcmp(x1, x2, sizeof x1)) {
      if (bad < 3) {
        printf("N=%d case %d mode %d MISMATCH err %d %d\n", N, c, mode, e1, e2);
        for (int i = 0; i < N; ++i) printf("  val %.17g %.17g\n", v1[i], v2[i]);
      }
      ++bad;
    }
  }
  printf("N=%d: %d cases, %d mismatches\n", N, ncase, bad);
  return bad;
}

int main() {
  int bad = 0;
  bad += check<3>(4000, 1);
  bad += check<4>(4000, 2);
  bad += check<5>(4000, 3);
  bad += check<6>(4000, 4);
  return bad ? 1 : 0;
}
