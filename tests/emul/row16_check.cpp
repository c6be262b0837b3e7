// row16_check.cpp - TEST-ONLY: the dense-algebra primitives of the row-of-16 kernels
// (spartacus_surface_amd/csrc/ss_row16.hpp, cholesky_rows of ss_layer16.hpp) on packs of 16 "lanes"
// against plain loops: product, unpivoted LU + substitution, symmetric elimination (both Cholesky
// factors), sums over lanes, for the orders 7, 12 and 16.  Prints one line per order and check:
//   <order> <name> <max abs error relative to the largest element>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../spartacus_surface_amd/csrc/ss_layer16.hpp"

using namespace ss::r16;

static double rnd() { return rand() / (double)RAND_MAX - 0.5; }

template <int N>
static void to_rows(const std::vector<double>& a, rv* A) {   // a row-major N x N -> rows on lanes
  for (int j = 0; j < N; ++j) A[j] = gen([&](int i) { return i < N ? a[i * N + j] : 0.0; });
}
template <int N>
static double diff_rows(const rv* A, const std::vector<double>& a) {
  double e = 0.0, s = 0.0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) { e = std::fmax(e, std::fabs(A[j].v[i] - a[i * N + j])); s = std::fmax(s, std::fabs(a[i * N + j])); }
  return e / s;
}

template <int N>
static void check() {
  std::vector<double> a(N * N), b(N * N), c(N * N), x(N * N);
  for (auto& v : a) v = rnd();
  for (auto& v : b) v = rnd();
  for (int i = 0; i < N; ++i) a[i * N + i] += N;       // diagonally dominant: unpivoted LU is safe
  rv A[N], B[N], C[N];
  to_rows<N>(a, A);
  to_rows<N>(b, B);
  // product
  mul<N, N, false, false>(A, nullptr, B, C);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) { double s = 0; for (int k = 0; k < N; ++k) s += a[i * N + k] * b[k * N + j]; c[i * N + j] = s; }
  std::printf("%d mul %.3e\n", N, diff_rows<N>(C, c));
  // LU + substitution: X = A^-1 B, checked through A X = B
  LU<N> f;
  for (int j = 0; j < N; ++j) f.a[j] = A[j];
  lu_factor<N>(f);
  rv X[N];
  for (int j = 0; j < N; ++j) X[j] = B[j];
  lu_solve<N, N>(f, X);
  mul<N, N, false, false>(A, nullptr, X, C);
  std::printf("%d lu_solve %.3e\n", N, diff_rows<N>(C, b));
  // matrix-vector product and sum over the lanes
  rv xv = gen([&](int i) { return i < N ? b[i] : 0.0; }), y;
  mat_vec<N, false>(A, xv, y);
  double e = 0.0, tot = 0.0;
  for (int i = 0; i < N; ++i) { double s = 0; for (int k = 0; k < N; ++k) s += a[i * N + k] * b[k]; e = std::fmax(e, std::fabs(s - y.v[i])); tot += b[i]; }
  std::printf("%d mat_vec %.3e\n", N, e);
  std::printf("%d all_sum %.3e\n", N, std::fabs(all_sum<N>(xv) - tot));
  // symmetric elimination: S = B B^T + N I = L L^T; rows of L and of L^T
  std::vector<double> s(N * N), l(N * N, 0.0);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) { double t = (i == j) ? N : 0.0; for (int k = 0; k < N; ++k) t += b[i * N + k] * b[j * N + k]; s[i * N + j] = t; }
  for (int j = 0; j < N; ++j) {
    double d = s[j * N + j];
    for (int k = 0; k < j; ++k) d -= l[j * N + k] * l[j * N + k];
    l[j * N + j] = std::sqrt(d);
    for (int i = j + 1; i < N; ++i) {
      double t = s[i * N + j];
      for (int k = 0; k < j; ++k) t -= l[i * N + k] * l[j * N + k];
      l[i * N + j] = t / l[j * N + j];
    }
  }
  rv S[N], Lr[N];
  double rs[N];
  to_rows<N>(s, S);
  const bool ok = cholesky_rows<N>(S, Lr, rs, 0.0);
  std::printf("%d cholesky_ok %d\n", N, ok ? 1 : 0);
  std::printf("%d cholesky_L %.3e\n", N, diff_rows<N>(Lr, l));
  // rows of L^T: S[k] * rs_own for k >= i
  std::vector<double> lt(N * N, 0.0);
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) lt[i * N + j] = l[j * N + i];
  const rv rs_own = own<N>(rs);
  rv LT[N];
  for (int k = 0; k < N; ++k) LT[k] = select(lane_gt(k), rv(0.0), S[k] * rs_own);
  std::printf("%d cholesky_LT %.3e\n", N, diff_rows<N>(LT, lt));
}

int main() {
  srand(7);
  check<7>();
  check<12>();
  check<16>();
  return 0;
}
