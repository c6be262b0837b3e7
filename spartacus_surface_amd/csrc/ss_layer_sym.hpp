// ss_layer_sym.hpp - layer matrices in SYMMETRISED coordinates (orders that live in registers).
//
// What the reference computes (calc_matrices_lw_eig, radtool/radtool_calc_matrices_lw_eig.F90:142-228;
// calc_matrices_sw_eig, radtool/radtool_calc_matrices_sw_eig.F90:180-296: eigenvectors G1/G2 of the
// two-stream operator, the coefficient matrices C'_lower / C'_upper by Schur complement, R and T as
// sums of four products, three LU factorisations and an inverse) is here obtained from symmetric
// positive definite algebra only - no LU, no general eigen-solver:
//
// With D = diag(sqrt(frac_region weight_stream mu_stream)) (ss_eigen_sym.hpp)
//     D^-1 (G1 - G2) D = -C,   D^-1 (G1 + G2) D = -M,   C, M symmetric positive definite,
// and scaled fluxes u^ = D^-1 u the sum s = u^ + v^ and difference d = u^ - v^ of the upwelling and
// downwelling streams obey  s' = C d,  d' = M s  (from d(u+v)/dz = -(G1-G2)(u-v), d(u-v)/dz = -(G1+G2)(u+v)).
// A homogeneous layer is mirror symmetric about its mid-plane, so illumination that is symmetric /
// antisymmetric about the mid-plane decouples: with h = dz/2 and the spd "impedances"
//     Z+ = L^-T  kappa tanh(kappa h) L^-1,   Z- = L^-T  kappa coth(kappa h) L^-1,
//     C = L L^T,  K = L^T M L = Y sigma^2 Y^T,  kappa = Y sigma Y^T,
// the scaled reflectance and transmittance are Cayley transforms
//     R^ + T^ = (I - Z+)(I + Z+)^-1,   R^ - T^ = (I - Z-)(I + Z-)^-1,
// i.e. with W+- = (I + Z+-)^-1 (inverses of spd matrices, all eigenvalues in (0, 1]):
//     R^ = W+ + W- - I,   T^ = W+ - W-          (both SYMMETRIC: reciprocity).
// Emission (longwave): the particular solution is u = v = g^ = M^-1 b^ and the layer's source is
//     s^ = (I - R^ - T^) g^ = 2 Z+ W+ g^;
// the layer-integrated flux per unit incoming flux is 2 M^-1 Z+ W+ = 2 A diag(tanh(sigma h)/sigma) A^-T W+
// (A = L Y; the conservation identity "integrated flux = M^-1 x net flux convergence" without its
// cancellation), and the integrated flux of the emission 2 (g^ dz - IF g^).
// Direct beam (shortwave): for a source s e^{mu z} the particular solution is p^ = g3 + g4 = A c,
// q^ = g3 - g4 = mu A^-T c with c = (sigma^2 - mu^2)^-1 A^T 2 s^, and
//     S_up = g3 - R g4 - T g3 e,   S_dn = g4 e - T g4 - R g3 e      (e = e^{mu dz}).
//
// Arithmetic: two Cholesky factorisations, one symmetric eigen-decomposition, two triangular products,
// two symmetric rank-n updates, two spd inverses - about a third of the multiply-adds of the Schur
// route after the eigen-decomposition, half the live registers (symmetric matrices are held packed,
// 21 doubles at n = 6), and better conditioned: every inverse is of I + (spd), where the reference
// inverts eigenvector matrices (its rounding noise on degenerate and graded spectra).
// The unit returns -1 BEFORE writing anything when the structure is not there numerically (asymmetric
// exchange, C or M not positive definite); the caller then takes the general route (ss_layer_matrices.hpp).
#pragma once

#include "ss_common.hpp"
#include "ss_eigen_sym.hpp"
#include "ss_layer_matrices.hpp"

namespace ss {
namespace sym {

template <int N> constexpr int kTri = N * (N + 1) / 2;
// index of element (i, j), i >= j, of a packed lower triangle (column by column)
template <int N> constexpr int lo(int i, int j) { return i + j * N - j * (j + 1) / 2; }
// element (i, j) of a packed symmetric matrix, any order of i and j
template <int N> constexpr int sy(int i, int j) { return i >= j ? lo<N>(i, j) : lo<N>(j, i); }

SS_HD void sqrt_rsqrt(double x, double& s, double& rs) {
#if defined(SS_SYM_IEEE)
  s = sqrt(x);
  rs = 1.0 / s;
#else
  fast_sqrt_rsqrt(x, s, rs);
#endif
}

// In-place Cholesky factor of a packed spd matrix: A = L L^T, L in the packed lower triangle, rd = 1 / diag(L).
// Returns false when a pivot is not positive (relative to "size"); the factor is then meaningless.
template <int N>
SS_HD bool cholesky(double* A, double* rd, double size) {
  bool ok = true;
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) {
    double s = A[lo<N>(j, j)];
    SS_SYM_UNROLL(N)
    for (int k = 0; k < j; ++k) s -= A[lo<N>(j, k)] * A[lo<N>(j, k)];
    if (!(s > 1.0e-14 * size)) ok = false;
    double ljj, rl;
    sqrt_rsqrt(ok ? s : 1.0, ljj, rl);
    A[lo<N>(j, j)] = ljj;
    rd[j] = rl;
    SS_SYM_UNROLL(N)
    for (int i = j + 1; i < N; ++i) {
      double t = A[lo<N>(i, j)];
      SS_SYM_UNROLL(N)
      for (int k = 0; k < j; ++k) t -= A[lo<N>(i, k)] * A[lo<N>(j, k)];
      A[lo<N>(i, j)] = t * rl;
    }
  }
  return ok;
}

// x := L^-1 x, x := L^-T x (L packed lower with reciprocal diagonal rd)
template <int N>
SS_HD void solve_L(const double* L, const double* rd, double* x) {
  SS_SYM_UNROLL(N)
  for (int i = 0; i < N; ++i) {
    double s = x[i];
    SS_SYM_UNROLL(N)
    for (int k = 0; k < i; ++k) s -= L[lo<N>(i, k)] * x[k];
    x[i] = s * rd[i];
  }
}
template <int N>
SS_HD void solve_LT(const double* L, const double* rd, double* x) {
  SS_SYM_UNROLL(N)
  for (int i = N - 1; i >= 0; --i) {
    double s = x[i];
    SS_SYM_UNROLL(N)
    for (int k = i + 1; k < N; ++k) s -= L[lo<N>(k, i)] * x[k];
    x[i] = s * rd[i];
  }
}

// y = S x for a packed symmetric S
template <int N>
SS_HD void symv(const double* S, const double* x, double* y) {
  SS_SYM_UNROLL(N)
  for (int i = 0; i < N; ++i) {
    double s = 0.0;
    SS_SYM_UNROLL(N)
    for (int k = 0; k < N; ++k) s += S[sy<N>(i, k)] * x[k];
    y[i] = s;
  }
}

// Z (packed) = X diag(f) X^T for a full N x N matrix X (column-major)
template <int N>
SS_HD void rank_update(const double* X, const double* f, double* Z) {
  SS_SYM_UNROLL(N)
  for (int i = 0; i < N; ++i) {
    double xf[N];
    SS_SYM_UNROLL(N)
    for (int k = 0; k < N; ++k) xf[k] = SS_E(X, i, k, N) * f[k];
    SS_SYM_UNROLL(N)
    for (int j = 0; j <= i; ++j) {
      double s = 0.0;
      SS_SYM_UNROLL(N)
      for (int k = 0; k < N; ++k) s += xf[k] * SS_E(X, j, k, N);
      Z[lo<N>(i, j)] = s;
    }
  }
}

// Z (packed spd) := (I + Z)^-1, in place: Cholesky I + Z = R R^T, R^-1 in place, then R^-T R^-1.
// I + Z with Z positive semi-definite has all pivots >= 1: no pivot test.
template <int N>
SS_HD void inverse_of_identity_plus(double* Z) {
  double rd[N];
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) Z[lo<N>(j, j)] += 1.0;
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) {
    double s = Z[lo<N>(j, j)];
    SS_SYM_UNROLL(N)
    for (int k = 0; k < j; ++k) s -= Z[lo<N>(j, k)] * Z[lo<N>(j, k)];
    double ljj, rl;
    sqrt_rsqrt(s, ljj, rl);
    rd[j] = rl;
    SS_SYM_UNROLL(N)
    for (int i = j + 1; i < N; ++i) {
      double t = Z[lo<N>(i, j)];
      SS_SYM_UNROLL(N)
      for (int k = 0; k < j; ++k) t -= Z[lo<N>(i, k)] * Z[lo<N>(j, k)];
      Z[lo<N>(i, j)] = t * rl;
    }
  }
  // Q = R^-1 (lower), column by column: Q(j,j) = 1/R(j,j), Q(i,j) = -Q(i,i) sum_{k=j}^{i-1} R(i,k) Q(k,j)
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) {
    double q[N];
    q[j] = rd[j];
    SS_SYM_UNROLL(N)
    for (int i = j + 1; i < N; ++i) {
      double s = 0.0;
      SS_SYM_UNROLL(N)
      for (int k = j; k < i; ++k) s += Z[lo<N>(i, k)] * q[k];
      q[i] = -s * rd[i];
    }
    // column j of R is not needed by the later columns of Q (they use rows i > j', k >= j' > j)
    SS_SYM_UNROLL(N)
    for (int i = j; i < N; ++i) Z[lo<N>(i, j)] = q[i];
  }
  // (I + Z)^-1 = Q^T Q: element (i, j), i >= j, = sum_{k >= i} Q(k,i) Q(k,j); rows top to bottom in place
  // (element (i, j) reads Q(k, .) with k >= i only, and of row i itself the entries not yet overwritten)
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) {
    SS_SYM_UNROLL(N)
    for (int i = j; i < N; ++i) {
      double s = 0.0;
      SS_SYM_UNROLL(N)
      for (int k = i; k < N; ++k) s += Z[lo<N>(k, i)] * Z[lo<N>(k, j)];
      Z[lo<N>(i, j)] = s;
    }
  }
}

// Packed C = -D^-1 (G1 - G2) D and M = C - 2 D^-1 G2 D from the full Gamma matrices (drop-in form; the
// fused kernels build them from the layer's scalars), with the symmetry test of ss_eigen_sym.hpp.
// GD = G1 - G2.  Returns the size (largest magnitude) or a negative value when the structure is missing.
template <int N>
SS_HD double build_CM(const double* GD, const double* G2, const double* dscale, double* C, double* M) {
  double rd[N];
  bool ok = true;
  SS_SYM_UNROLL(N)
  for (int i = 0; i < N; ++i) {
    if (!(dscale[i] > 0.0)) ok = false;
    rd[i] = 1.0 / dscale[i];
  }
  double asym = 0.0, size = 0.0;
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) {
    SS_SYM_UNROLL(N)
    for (int i = j; i < N; ++i) {
      const double c_ij = -SS_E(GD, i, j, N) * (dscale[j] * rd[i]);
      const double s_ij = SS_E(G2, i, j, N) * (dscale[j] * rd[i]);
      if (i != j) {
        const double c_ji = -SS_E(GD, j, i, N) * (dscale[i] * rd[j]);
        const double s_ji = SS_E(G2, j, i, N) * (dscale[i] * rd[j]);
        asym = ss_max(asym, ss_max(fabs(c_ij - c_ji), fabs(s_ij - s_ji)));
      }
      size = ss_max(size, ss_max(fabs(c_ij), fabs(s_ij)));
      C[lo<N>(i, j)] = c_ij;
      M[lo<N>(i, j)] = c_ij - 2.0 * s_ij;
    }
  }
  if (!ok || !(asym <= 1.0e-10 * size)) return -1.0;
  return size;
}

// Spectral part shared by longwave and shortwave: from packed C, M (both destroyed) to
//   sig (the eigenvalues sigma^2 of K, clamped at zero), A = L Y, Ainv_T = L^-T Y (= A^-T).
// Returns 0, > 0 when the QL iteration did not converge, -1 when C is not positive definite or an
// eigenvalue is not positive.
// Spectral part shared by longwave and shortwave.  In: packed C (overwritten by its Cholesky factor L, rl =
// 1 / diag L) and M.  Out: ev = the eigenvalues sigma^2 of K = L^T M L (Rayleigh quotients with M, clamped at
// zero) and AiT = L^-T Y = A^-T (A = L Y is never stored: A x = C (A^-T x) = L (L^T (A^-T x))).
// Returns 0, > 0 when the QL iteration did not converge, -1 when C is not positive definite or an
// eigenvalue is negative beyond rounding.
template <int N, class Lds>
SS_HD int spectral(double* C, double* rl, const double* M, double size, double* ev, double* AiT, const Lds& lds) {
  if (!cholesky<N>(C, rl, size)) return -1;
  // K = L^T M L column by column: t = M L(:, j), K(i, j) = L(:, i)^T t for i >= j (tred2 reads the lower triangle)
  double* K = AiT;      // the eigenvectors, then A^-T, replace K in place
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) {
    double t[N];
    SS_SYM_UNROLL(N)
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
      SS_SYM_UNROLL(N)
      for (int k = j; k < N; ++k) s += M[sy<N>(i, k)] * C[lo<N>(k, j)];
      t[i] = s;
    }
    SS_SYM_UNROLL(N)
    for (int i = j; i < N; ++i) {
      double s = 0.0;
      SS_SYM_UNROLL(N)
      for (int k = i; k < N; ++k) s += C[lo<N>(k, i)] * t[k];
      SS_E(K, i, j, N) = s;
      SS_E(K, j, i, N) = s;
    }
  }
  int ierr = 0;
  if constexpr (N == 1) {
    ev[0] = K[0];
    K[0] = 1.0;
  } else if constexpr (N == 2) {
    // one Jacobi rotation: exact eigenvectors of a symmetric 2 x 2 matrix
    const double a = K[0], b = K[1], c = K[3];
    double cs = 1.0, sn = 0.0, t = 0.0;
    if (b != 0.0) {
      const double zeta = (c - a) / (2.0 * b);
      t = ss_sign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
      cs = 1.0 / sqrt(1.0 + t * t);
      sn = cs * t;
    }
    ev[0] = a - t * b;
    ev[1] = c + t * b;
    K[0] = cs; K[1] = -sn; K[2] = sn; K[3] = cs;
  } else {
  // Order of the components handed to the QL iteration: by ascending diagonal of K.  EISPACK's tred2 + tql2
  // keep the small eigenvalues AND their eigenvectors of a graded matrix accurate when its elements grow
  // towards the bottom right (measured on the transparent trunk layer of rami4pilps, eigenvalues over 13
  // decades: S_up off by 3e-6 in the solver's order, where the oblique streams - the largest rates - come
  // first; 4e-9 graded; the reference's general route 2.5e-8).  The eigenvectors matter, not only the
  // eigenvalues: the particular solutions of the shortwave divide by sigma^2 - mu^2, and the unattenuated
  // direct mode of a transparent layer has mu^2 inside the cluster of tiny eigenvalues.
  // The permutation is data dependent, so it goes through the lane's LDS slots (N^2 <= 36 of them).
  int rank[N];
  SS_SYM_UNROLL(N)
  for (int i = 0; i < N; ++i) {
    int r = 0;
    SS_SYM_UNROLL(N)
    for (int j = 0; j < N; ++j) {
      const double di = SS_E(K, i, i, N), dj = SS_E(K, j, j, N);
      if (j != i && (dj < di || (dj == di && j < i))) ++r;
    }
    rank[i] = r;
  }
  if constexpr (N * N <= Lds::kSlots) {
    SS_SYM_UNROLL(N)
    for (int j = 0; j < N; ++j) {
      SS_SYM_UNROLL(N)
      for (int i = 0; i < N; ++i) lds.put(rank[i] + rank[j] * N, SS_E(K, i, j, N));
    }
    SS_SYM_UNROLL(N)
    for (int k = 0; k < N * N; ++k) K[k] = lds.get(k);
  } else {
    // orders whose matrices live in scratch memory anyway (above 16): a copy takes the place of the LDS slots
    double Kc[N * N];
    for (int k = 0; k < N * N; ++k) Kc[k] = K[k];
    for (int j = 0; j < N; ++j)
      for (int i = 0; i < N; ++i) K[rank[i] + rank[j] * N] = Kc[i + j * N];
  }
  double e[N];
  sym_tridiagonalise<N>(K, ev, e);
  ierr = sym_tridiagonal_ql<N>(ev, e, K);
  // rows of the eigenvector matrix back into the solver's order
  if constexpr (N * N <= Lds::kSlots) {
    SS_SYM_UNROLL(N)
    for (int k = 0; k < N * N; ++k) lds.put(k, K[k]);
    SS_SYM_UNROLL(N)
    for (int j = 0; j < N; ++j) {
      SS_SYM_UNROLL(N)
      for (int i = 0; i < N; ++i) SS_E(K, i, j, N) = lds.get(rank[i] + j * N);
    }
  } else {
    double Kc[N * N];
    for (int k = 0; k < N * N; ++k) Kc[k] = K[k];
    for (int j = 0; j < N; ++j)
      for (int i = 0; i < N; ++i) SS_E(K, i, j, N) = Kc[rank[i] + j * N];
  }
  }
  // The QL iteration has absolute accuracy eps * (largest eigenvalue): an eigenvalue far below that
  // (transparent, conservative layers: 13 decades) may come out as rounding noise of either sign.  A negative
  // eigenvalue that is NOT noise declines the unit.
  double evmax = 0.0;
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) evmax = ss_max(evmax, ev[j]);
  bool pos = evmax > 0.0;
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j)
    if (!(ev[j] > -1.0e-10 * evmax)) pos = false;
  if (!pos) return -1;
  // Column by column: y -> A^-T(:, j) = L^-T y (back substitution), and the eigenvalue as the Rayleigh quotient
  // with M itself, sigma^2 = y^T K y = a^T M a with a = L y: the eigenvector error (~ eps x largest / gap)
  // enters it squared, what is left is the rounding of M a, ~1e-17 absolute: eigenvalues 10 decades down keep
  // 6-8 digits.  They carry the absorption of a nearly conservative layer (reflect_transmit) and meet mu^2 of
  // the unattenuated direct mode in the particular solutions of the shortwave.
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) {
    double y[N], a[N], ma[N];
    SS_SYM_UNROLL(N)
    for (int i = 0; i < N; ++i) y[i] = SS_E(K, i, j, N);
    SS_SYM_UNROLL(N)
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
      SS_SYM_UNROLL(N)
      for (int k = 0; k <= i; ++k) s += C[lo<N>(i, k)] * y[k];
      a[i] = s;
    }
    symv<N>(M, a, ma);
    double rq = 0.0;
    SS_SYM_UNROLL(N)
    for (int i = 0; i < N; ++i) rq += a[i] * ma[i];
    ev[j] = ss_max(rq, 0.0);
    solve_LT<N>(C, rl, y);
    SS_SYM_UNROLL(N)
    for (int i = 0; i < N; ++i) SS_E(AiT, i, j, N) = y[i];
  }
  return ierr;
}

// x := C x = L (L^T x) with the packed Cholesky factor
template <int N>
SS_HD void apply_C(const double* L, double* x) {
  double t[N];
  SS_SYM_UNROLL(N)
  for (int k = 0; k < N; ++k) {
    double s = 0.0;
    SS_SYM_UNROLL(N)
    for (int i = k; i < N; ++i) s += L[lo<N>(i, k)] * x[i];
    t[k] = s;
  }
  SS_SYM_UNROLL(N)
  for (int i = 0; i < N; ++i) {
    double s = 0.0;
    SS_SYM_UNROLL(N)
    for (int k = 0; k <= i; ++k) s += L[lo<N>(i, k)] * t[k];
    x[i] = s;
  }
}

// phi(x) = tanh(x) / x for x >= 0: smooth, in (0, 1], phi(0) = 1
SS_HD double phi_tanh(double x) {
  const double x2 = x * x;
  if (x < 0.03) return 1.0 + x2 * (-1.0 / 3.0 + x2 * (2.0 / 15.0 + x2 * (-17.0 / 315.0)));
  const double em1 = expm1(-2.0 * x);      // in (-1, 0): tanh x = -em1 / (2 + em1)
  return -em1 / ((2.0 + em1) * x);
}

// The two impedances of a layer of half-thickness h and their Cayley transforms W+- = (I + Z+-)^-1 (packed):
//   Z- = A^-T diag(1 / (h phi)) A^-1
//   Z+ = M Phi,  Phi = C Psi,  Psi = A^-T diag(h phi) A^-1 = L^-T (tanh(kappa h) / kappa) L^-1 (symmetric)
// (Z+ = L^-T kappa tanh(kappa h) L^-1, since M L = L^-T K).  An eigenvalue enters only through phi(sigma h),
// smooth at zero.  This matters: the absorption of a layer is 1 - R - T = 2 Z+ W+, and for a nearly
// conservative layer (clear air: extinction 1e-5, single-scattering albedo 0.999) it lives in eigenvalues 13
// decades below the largest one, which a QL iteration delivers with NO correct digit.  Z+ in its spectral form
// A^-T diag(sigma tanh(sigma h)) A^-1 inherits that error (measured: layer absorption off by 1e-3 ... 1e-6
// depending on the index order handed to the QL iteration); M Phi takes the small numbers from M - input
// data - and only a smooth factor from the eigen-decomposition (2e-7 whatever the order; LAPACK dsyevd and
// one-sided Jacobi in the spectral form: 6e-7; the reference: 2e-5).
// KEEP_PHI: Phi (full, N x N) is returned (longwave: integrated flux = 2 Phi W+); else Phi may be nullptr.
template <int N, bool KEEP_PHI>
SS_HD void impedances(const double* M, const double* L, const double* AiT, const double* ev, double h,
                      double* Wp, double* Wm, double* Phi) {
  double t[N], fm[N];
  SS_SYM_UNROLL(N)
  for (int k = 0; k < N; ++k) {
    t[k] = h * phi_tanh(sqrt(ev[k]) * h);
    fm[k] = 1.0 / t[k];
  }
  rank_update<N>(AiT, fm, Wm);
  inverse_of_identity_plus<N>(Wm);
  double Psi[kTri<N>];
  rank_update<N>(AiT, t, Psi);
  // column j of Phi = C Psi(:, j); Z+(i, j) = M(i, :) Phi(:, j) for i >= j (symmetric in exact arithmetic)
  SS_SYM_UNROLL(N)
  for (int j = 0; j < N; ++j) {
    double c[N];
    SS_SYM_UNROLL(N)
    for (int i = 0; i < N; ++i) c[i] = Psi[sy<N>(i, j)];
    apply_C<N>(L, c);
    if constexpr (KEEP_PHI) {
      SS_SYM_UNROLL(N)
      for (int i = 0; i < N; ++i) SS_E(Phi, i, j, N) = c[i];
    }
    SS_SYM_UNROLL(N)
    for (int i = j; i < N; ++i) {
      double s = 0.0;
      SS_SYM_UNROLL(N)
      for (int k = 0; k < N; ++k) s += M[sy<N>(i, k)] * c[k];
      Wp[lo<N>(i, j)] = s;
    }
  }
  inverse_of_identity_plus<N>(Wp);
}

// R^ = W+ + W- - I and T^ = W+ - W- into the record: packed and scaled where the record is
// (SwLayRec / LwLayRec, ss_layer_matrices.hpp), else un-scaled (R = D R^ D^-1) into the full blocks
template <int NF, int N, int I0, bool PACKED, class W>
SS_HD void put_reflect_transmit(const W& w, int offR, int offT, const double* Wp, const double* Wm,
                                const double* dsc, const double* rd) {
  if constexpr (PACKED) {
    put_packed<NF, N, I0>(w, offR, [&](auto ic, auto jc) {
      constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
      return Wp[lo<N>(i, j)] + Wm[lo<N>(i, j)] - (i == j ? 1.0 : 0.0);
    });
    put_packed<NF, N, I0>(w, offT, [&](auto ic, auto jc) {
      constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
      return Wp[lo<N>(i, j)] - Wm[lo<N>(i, j)];
    });
  } else {
    double t[N * N];
    SS_SYM_UNROLL(N)
    for (int j = 0; j < N; ++j) {
      SS_SYM_UNROLL(N)
      for (int i = 0; i < N; ++i)
        SS_E(t, i, j, N) = (dsc[i] * rd[j]) * (Wp[sy<N>(i, j)] + Wm[sy<N>(i, j)] - (i == j ? 1.0 : 0.0));
    }
    put_embedded<NF, NF, N, N, I0, I0>(w, offR, t);
    SS_SYM_UNROLL(N)
    for (int j = 0; j < N; ++j) {
      SS_SYM_UNROLL(N)
      for (int i = 0; i < N; ++i) SS_E(t, i, j, N) = (dsc[i] * rd[j]) * (Wp[sy<N>(i, j)] - Wm[sy<N>(i, j)]);
    }
    put_embedded<NF, NF, N, N, I0, I0>(w, offT, t);
  }
}

// calc_matrices_lw_eig in symmetrised coordinates for an N sub-problem embedded at stream offset I0 of
// the (NF x NF) record; same outputs as calc_matrices_lw, in the layout of the pipeline's layer record.
template <int NF, int N, int I0, class W, class Lds>
SS_HD int calc_matrices_lw_sym(const double* g1, const double* g2, const double* b, const double* dsc, double dz,
                               const W& w, const Lds& lds) {
  using Rec = LwLayRec<NF>;
  double C[kTri<N>], M[kTri<N>], size;
  {
    double GD[N * N];
    SS_UNROLL_NN(N)
    for (int k = 0; k < N * N; ++k) GD[k] = g1[k] - g2[k];
    size = build_CM<N>(GD, g2, dsc, C, M);
  }
  if (!(size > 0.0)) return -1;
  // particular solution g^ = M^-1 b^ (Cholesky of a copy of M: M itself is needed later)
  double gh[N];
  {
    double F[kTri<N>], rf[N];
    SS_SYM_UNROLL(N)
    for (int k = 0; k < kTri<N>; ++k) F[k] = M[k];
    if (!cholesky<N>(F, rf, size)) return -1;
    SS_SYM_UNROLL(N)
    for (int i = 0; i < N; ++i) gh[i] = b[i] / dsc[i];
    solve_L<N>(F, rf, gh);
    solve_LT<N>(F, rf, gh);
  }
  double ev[N], rl[N], AiT[N * N];
  const int ierr = spectral<N>(C, rl, M, size, ev, AiT, lds);
  if (ierr < 0) return -1;
  double Wp[kTri<N>], Wm[kTri<N>], Phi[N * N];
  impedances<N, true>(M, C, AiT, ev, 0.5 * dz, Wp, Wm, Phi);
  double rd[N];
  SS_SYM_UNROLL(N)
  for (int i = 0; i < N; ++i) rd[i] = 1.0 / dsc[i];
  put_reflect_transmit<NF, N, I0, Rec::PACKED>(w, Rec::R, Rec::T, Wp, Wm, dsc, rd);
  // source = 2 Z+ W+ g^ = 2 M (Phi (W+ g^)): the small numbers come from M
  double y[N], py[N], srch[N];
  symv<N>(Wp, gh, y);
  mat_vec<N, N>(Phi, y, py);
  symv<N>(M, py, srch);
  put_embedded_vec<NF, N, I0>(w, Rec::SRC, [&](int i) { return dsc[i] * (2.0 * srch[i]); });
  // integrated flux matrix IF = 2 M^-1 Z+ W+ = 2 Phi W+ and integrated source 2 (g^ dz - IF g^) = 2 (g^ dz - 2 Phi y)
  {
    double IF[N * N];
    SS_SYM_UNROLL(N)
    for (int j = 0; j < N; ++j) {
      SS_SYM_UNROLL(N)
      for (int i = 0; i < N; ++i) {
        double s2 = 0.0;
        SS_SYM_UNROLL(N)
        for (int k = 0; k < N; ++k) s2 += SS_E(Phi, i, k, N) * Wp[sy<N>(k, j)];
        SS_E(IF, i, j, N) = (2.0 * dsc[i] * rd[j]) * s2;
      }
    }
    put_embedded<NF, NF, N, N, I0, I0>(w, Rec::IFLUX, IF);
    put_embedded_vec<NF, N, I0>(w, Rec::ISRC, [&](int i) { return dsc[i] * (2.0 * (gh[i] * dz - 2.0 * py[i])); });
  }
  return ierr;
}

// calc_matrices_sw_eig in symmetrised coordinates for an (N, D) sub-problem embedded at (I0, R0) of the
// (NF, DF) record; same outputs as calc_matrices_sw, in the layout of the pipeline's layer record (the blocks
// behind R and T sit at the same distances as in SwRec).  The direct-beam part (Gamma0: D x D)
// is shared with the general route (sw_direct_embedded); the direct-diffuse part, a (2n + d) square system
// in the reference (direct_diffuse_part, sw_eig :303-386), is two symmetric matrix-vector products per
// direct eigenvalue here.
template <int NF, int DF, int N, int D, int I0, int R0, class W, class Lds>
SS_HD int calc_matrices_sw_sym(const double* g0, const double* g1, const double* g2, const double* g3,
                               const double* dsc0, const double* dsc, double dz, const W& w, const Lds& lds) {
  using Rec = SwLayRec<NF, DF>;
  double C[kTri<N>], M[kTri<N>], size;
  {
    double GD[N * N];
    SS_UNROLL_NN(N)
    for (int k = 0; k < N * N; ++k) GD[k] = g1[k] - g2[k];
    size = build_CM<N>(GD, g2, dsc, C, M);
  }
  if (!(size > 0.0)) return -1;
  double ev[N], rl[N], AiT[N * N];
  int ierr = spectral<N>(C, rl, M, size, ev, AiT, lds);
  if (ierr < 0) return -1;

  double mu[D], G0[D * D], G0inv[D * D], emu[D];
  ierr += sw_direct_embedded<NF, DF, D, R0, Rec>(g0, dsc0, dz, w, mu, G0, G0inv, emu);
  double rd[N];
  SS_SYM_UNROLL(N)
  for (int i = 0; i < N; ++i) rd[i] = 1.0 / dsc[i];

  // Particular solutions of the direct sources (sw_eig :231-253), one per direct eigenvalue mu, scaled: with
  // s^ = D^-1 (Gamma3 G0)(:, jd), p^ = g3^ + g4^ and q^ = g3^ - g4^ obey (C M - mu^2) p^ = 2 C s^, q^ = mu C^-1 p^:
  //     p^ = C r,  q^ = mu r,  r = A^-T (sigma^2 - mu^2)^-1 A^-1 C 2 s^
  // (the eigenvalues are the Rayleigh quotients of spectral(): in a transparent layer the unattenuated direct
  // mode, mu^2 ~ 1e-10, meets eigenvalues of the same size in this division).  Kept: r (N per mode).
  double rv[N * D];
  SS_SYM_UNROLL(N)
  for (int jd = 0; jd < D; ++jd) {
    double sh[N];
    SS_SYM_UNROLL(N)
    for (int i = 0; i < N; ++i) {
      double t = 0.0;
      SS_SYM_UNROLL(N)
      for (int k = 0; k < D; ++k) t += SS_E(g3, i, k, N) * SS_E(G0, k, jd, D);
      sh[i] = 2.0 * t * rd[i];
    }
    apply_C<N>(C, sh);
    double c[N];
    SS_SYM_UNROLL(N)
    for (int k = 0; k < N; ++k) {
      double t = 0.0;
      SS_SYM_UNROLL(N)
      for (int i = 0; i < N; ++i) t += SS_E(AiT, i, k, N) * sh[i];
      c[k] = t / (ev[k] - mu[jd] * mu[jd]);
    }
    SS_SYM_UNROLL(N)
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
      SS_SYM_UNROLL(N)
      for (int k = 0; k < N; ++k) s += SS_E(AiT, i, k, N) * c[k];
      SS_E(rv, i, jd, N) = s;
    }
  }
  double Wp[kTri<N>], Wm[kTri<N>];
  impedances<N, false>(M, C, AiT, ev, 0.5 * dz, Wp, Wm, nullptr);
  put_reflect_transmit<NF, N, I0, Rec::PACKED>(w, Rec::R, Rec::T, Wp, Wm, dsc, rd);
  // The fluxes that leave the layer per unit amplitude of a direct mode: with g3 = (p + q) / 2, g4 = (p - q) / 2,
  // e = exp(mu dz), R^ = W+ + W- - I, T^ = W+ - W-:
  //   S_up = g3 - R g4 - T g3 e = p - W+ x+ - W- x-,   S_dn = g4 e - T g4 - R g3 e = e p - W+ x+ + W- x-,
  //   x+ = g4 + g3 e,  x- = g4 - g3 e
  double su[N * D], sd[N * D];
  SS_SYM_UNROLL(N)
  for (int jd = 0; jd < D; ++jd) {
    double p[N], xp[N], xm[N], wp[N], wm[N];
    SS_SYM_UNROLL(N)
    for (int i = 0; i < N; ++i) p[i] = SS_E(rv, i, jd, N);
    apply_C<N>(C, p);
    const double e = emu[jd];
    SS_SYM_UNROLL(N)
    for (int i = 0; i < N; ++i) {
      const double q = mu[jd] * SS_E(rv, i, jd, N);
      const double g3h = 0.5 * (p[i] + q), g4h = 0.5 * (p[i] - q);
      xp[i] = g4h + g3h * e;
      xm[i] = g4h - g3h * e;
    }
    symv<N>(Wp, xp, wp);
    symv<N>(Wm, xm, wm);
    SS_SYM_UNROLL(N)
    for (int i = 0; i < N; ++i) {
      // (rows scaled where the record is packed: the upward sweep then works in the scaled coordinates)
      const double un = Rec::PACKED ? 1.0 : dsc[i];
      SS_E(su, i, jd, N) = un * (p[i] - wp[i] - wm[i]);
      SS_E(sd, i, jd, N) = un * (e * p[i] - wp[i] + wm[i]);
    }
  }
  {
    double t[N * D];
    mat_mul<N, D, D>(su, G0inv, t);
    put_embedded<NF, DF, N, D, I0, R0>(w, Rec::SUP, t);
    mat_mul<N, D, D>(sd, G0inv, t);
    put_embedded<NF, DF, N, D, I0, R0>(w, Rec::SDN, t);
  }
  return ierr;
}

}  // namespace sym
}  // namespace ss
