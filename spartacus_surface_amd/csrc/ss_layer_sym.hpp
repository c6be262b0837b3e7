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
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    double s = A[lo<N>(j, j)];
    SS_UNROLL_FULL
    for (int k = 0; k < j; ++k) s -= A[lo<N>(j, k)] * A[lo<N>(j, k)];
    if (!(s > 1.0e-14 * size)) ok = false;
    double ljj, rl;
    sqrt_rsqrt(ok ? s : 1.0, ljj, rl);
    A[lo<N>(j, j)] = ljj;
    rd[j] = rl;
    SS_UNROLL_FULL
    for (int i = j + 1; i < N; ++i) {
      double t = A[lo<N>(i, j)];
      SS_UNROLL_FULL
      for (int k = 0; k < j; ++k) t -= A[lo<N>(i, k)] * A[lo<N>(j, k)];
      A[lo<N>(i, j)] = t * rl;
    }
  }
  return ok;
}

// x := L^-1 x, x := L^-T x (L packed lower with reciprocal diagonal rd)
template <int N>
SS_HD void solve_L(const double* L, const double* rd, double* x) {
  SS_UNROLL_FULL
  for (int i = 0; i < N; ++i) {
    double s = x[i];
    SS_UNROLL_FULL
    for (int k = 0; k < i; ++k) s -= L[lo<N>(i, k)] * x[k];
    x[i] = s * rd[i];
  }
}
template <int N>
SS_HD void solve_LT(const double* L, const double* rd, double* x) {
  SS_UNROLL_FULL
  for (int i = N - 1; i >= 0; --i) {
    double s = x[i];
    SS_UNROLL_FULL
    for (int k = i + 1; k < N; ++k) s -= L[lo<N>(k, i)] * x[k];
    x[i] = s * rd[i];
  }
}

// y = S x for a packed symmetric S
template <int N>
SS_HD void symv(const double* S, const double* x, double* y) {
  SS_UNROLL_FULL
  for (int i = 0; i < N; ++i) {
    double s = 0.0;
    SS_UNROLL_FULL
    for (int k = 0; k < N; ++k) s += S[sy<N>(i, k)] * x[k];
    y[i] = s;
  }
}

// Z (packed) = X diag(f) X^T for a full N x N matrix X (column-major)
template <int N>
SS_HD void rank_update(const double* X, const double* f, double* Z) {
  SS_UNROLL_FULL
  for (int i = 0; i < N; ++i) {
    double xf[N];
    SS_UNROLL_FULL
    for (int k = 0; k < N; ++k) xf[k] = SS_E(X, i, k, N) * f[k];
    SS_UNROLL_FULL
    for (int j = 0; j <= i; ++j) {
      double s = 0.0;
      SS_UNROLL_FULL
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
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) Z[lo<N>(j, j)] += 1.0;
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    double s = Z[lo<N>(j, j)];
    SS_UNROLL_FULL
    for (int k = 0; k < j; ++k) s -= Z[lo<N>(j, k)] * Z[lo<N>(j, k)];
    double ljj, rl;
    sqrt_rsqrt(s, ljj, rl);
    rd[j] = rl;
    SS_UNROLL_FULL
    for (int i = j + 1; i < N; ++i) {
      double t = Z[lo<N>(i, j)];
      SS_UNROLL_FULL
      for (int k = 0; k < j; ++k) t -= Z[lo<N>(i, k)] * Z[lo<N>(j, k)];
      Z[lo<N>(i, j)] = t * rl;
    }
  }
  // Q = R^-1 (lower), column by column: Q(j,j) = 1/R(j,j), Q(i,j) = -Q(i,i) sum_{k=j}^{i-1} R(i,k) Q(k,j)
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    double q[N];
    q[j] = rd[j];
    SS_UNROLL_FULL
    for (int i = j + 1; i < N; ++i) {
      double s = 0.0;
      SS_UNROLL_FULL
      for (int k = j; k < i; ++k) s += Z[lo<N>(i, k)] * q[k];
      q[i] = -s * rd[i];
    }
    // column j of R is not needed by the later columns of Q (they use rows i > j', k >= j' > j)
    SS_UNROLL_FULL
    for (int i = j; i < N; ++i) Z[lo<N>(i, j)] = q[i];
  }
  // (I + Z)^-1 = Q^T Q: element (i, j), i >= j, = sum_{k >= i} Q(k,i) Q(k,j); rows top to bottom in place
  // (element (i, j) reads Q(k, .) with k >= i only, and of row i itself the entries not yet overwritten)
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_FULL
    for (int i = j; i < N; ++i) {
      double s = 0.0;
      SS_UNROLL_FULL
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
  SS_UNROLL_FULL
  for (int i = 0; i < N; ++i) {
    if (!(dscale[i] > 0.0)) ok = false;
    rd[i] = 1.0 / dscale[i];
  }
  double asym = 0.0, size = 0.0;
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_FULL
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
template <int N, class Lds>
SS_HD int spectral(double* C, const double* M, double size, double* sig, double* A, double* AiT, const Lds& lds) {
  double rl[N];
  if (!cholesky<N>(C, rl, size)) return -1;
  // K = L^T M L: W = M L (full), then the lower triangle of L^T W, mirrored (tred2 reads the lower one)
  double K[N * N];
  {
    double Wm[N * N];
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) {
        double s = 0.0;
        SS_UNROLL_FULL
        for (int k = j; k < N; ++k) s += M[sy<N>(i, k)] * C[lo<N>(k, j)];
        SS_E(Wm, i, j, N) = s;
      }
    }
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      SS_UNROLL_FULL
      for (int i = j; i < N; ++i) {
        double s = 0.0;
        SS_UNROLL_FULL
        for (int k = i; k < N; ++k) s += C[lo<N>(k, i)] * SS_E(Wm, k, j, N);
        SS_E(K, i, j, N) = s;
        SS_E(K, j, i, N) = s;
      }
    }
  }
  // Order of the components handed to the QL iteration: by ascending diagonal of K.  EISPACK's tred2 + tql2
  // keep the small eigenvalues AND their eigenvectors of a graded matrix accurate when its elements grow
  // towards the bottom right (measured on the transparent trunk layer of rami4pilps, eigenvalues over 13
  // decades: S_up off by 3e-6 in the solver's order, where the oblique streams - the largest rates - come
  // first; 4e-9 graded; the reference's general route 2.5e-8).  The eigenvectors matter, not only the
  // eigenvalues: the particular solutions of the shortwave divide by sigma^2 - mu^2, and the unattenuated
  // direct mode of a transparent layer has mu^2 inside the cluster of tiny eigenvalues.
  // The permutation is data dependent, so it goes through the lane's LDS slots (N^2 <= 36 of them).
  int rank[N];
  SS_UNROLL_FULL
  for (int i = 0; i < N; ++i) {
    int r = 0;
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      const double di = SS_E(K, i, i, N), dj = SS_E(K, j, j, N);
      if (j != i && (dj < di || (dj == di && j < i))) ++r;
    }
    rank[i] = r;
  }
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) lds.put(rank[i] + rank[j] * N, SS_E(K, i, j, N));
  }
  SS_UNROLL_FULL
  for (int k = 0; k < N * N; ++k) K[k] = lds.get(k);
  double ev[N], e[N];
  sym_tridiagonalise<N>(K, ev, e);
  const int ierr = sym_tridiagonal_ql<N>(ev, e, K);
  // rows of the eigenvector matrix back into the solver's order
  SS_UNROLL_FULL
  for (int k = 0; k < N * N; ++k) lds.put(k, K[k]);
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) SS_E(K, i, j, N) = lds.get(rank[i] + j * N);
  }
  // sig receives the EIGENVALUES sigma^2.  The QL iteration has absolute accuracy eps * (largest eigenvalue):
  // an eigenvalue far below that (transparent, conservative layers: 13 decades) may come out as rounding
  // noise of either sign - harmless, because everything downstream depends on it through functions that
  // are smooth at zero (phi, below); clamped.  A negative eigenvalue that is NOT noise declines the unit.
  double evmax = 0.0;
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) evmax = ss_max(evmax, ev[j]);
  bool pos = true;
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    if (!(ev[j] > -1.0e-10 * evmax)) pos = false;
    sig[j] = ss_max(ev[j], 0.0);
  }
  if (!pos || !(evmax > 0.0)) return -1;
  // A = L Y, A^-T = L^-T Y (back substitution with L^T), column by column
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
      SS_UNROLL_FULL
      for (int k = 0; k <= i; ++k) s += C[lo<N>(i, k)] * SS_E(K, k, j, N);
      SS_E(A, i, j, N) = s;
    }
    double v[N];
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) v[i] = SS_E(K, i, j, N);
    solve_LT<N>(C, rl, v);
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) SS_E(AiT, i, j, N) = v[i];
#if !defined(SS_SYM_NO_RAYLEIGH)
    // Eigenvalue as the Rayleigh quotient with M itself: sigma^2 = y^T K y = (L y)^T M (L y).  The QL
    // iteration returns an eigenvalue with the absolute accuracy eps x (largest eigenvalue), the eigenvector
    // with an error ~ eps x largest / gap that enters the quotient squared; what is left is the rounding of
    // M a, ~1e-17 absolute: eigenvalues 10 decades down keep 6-8 digits.  They carry the absorption of a
    // nearly conservative layer (see reflect_transmit) and meet mu^2 of the unattenuated direct mode in
    // the particular solutions of the shortwave.
    {
      double ma[N];
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) v[i] = SS_E(A, i, j, N);
      symv<N>(M, v, ma);
      double rq = 0.0;
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) rq += v[i] * ma[i];
      sig[j] = ss_max(rq, 0.0);
    }
#endif
  }
  return ierr;
}

// phi(x) = tanh(x) / x for x >= 0: smooth, in (0, 1], phi(0) = 1
SS_HD double phi_tanh(double x) {
  const double x2 = x * x;
  if (x < 0.03) return 1.0 + x2 * (-1.0 / 3.0 + x2 * (2.0 / 15.0 + x2 * (-17.0 / 315.0)));
  const double em1 = expm1(-2.0 * x);      // in (-1, 0): tanh x = -em1 / (2 + em1)
  return -em1 / ((2.0 + em1) * x);
}

// Scaled reflectance and transmittance (packed symmetric) of a layer of half-thickness h, W+, and
//   Phi = L (tanh(kappa h) / kappa) L^-1 = A diag(h phi(sigma h)) A^-1   (full, N x N).
// The impedances are formed as
//   Z+ = M Phi          (= L^-T kappa tanh(kappa h) L^-1, since M L = L^-T K)
//   Z- = A^-T diag(1 / (h phi)) A^-1
// so that an eigenvalue enters only through phi, which is smooth at zero.  This matters: the absorption of
// a layer is 1 - R - T = 2 Z+ W+, and for a nearly conservative layer (clear air: extinction 1e-5, single-
// scattering albedo 0.999) it lives in eigenvalues 13 decades below the largest one, which a QL iteration
// delivers with NO correct digit.  Z+ = A^-T diag(sigma tanh(sigma h)) A^-1 inherits that error (measured:
// layer absorption off by 1e-3, or 1e-4 ... 1e-6 with other index orders handed to the QL iteration);
// M Phi takes the small numbers from M - input data - and only a smooth factor from the eigen-decomposition
// (8e-7 whatever the order; LAPACK dsyevd and one-sided Jacobi in the old form: 6e-7; the reference: 2e-5).
template <int N>
SS_HD void reflect_transmit(const double* M, const double* A, const double* AiT, const double* ev, double h,
                            double* Rh, double* Th, double* Wp, double* Phi) {
  double t[N], fm[N];
  SS_UNROLL_FULL
  for (int k = 0; k < N; ++k) {
    const double ph = phi_tanh(sqrt(ev[k]) * h);
    t[k] = h * ph;
    fm[k] = 1.0 / t[k];
  }
  // Phi(i, j) = sum_k A(i,k) t_k AiT(j,k)
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    double at[N];
    SS_UNROLL_FULL
    for (int k = 0; k < N; ++k) at[k] = t[k] * SS_E(AiT, j, k, N);
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
      SS_UNROLL_FULL
      for (int k = 0; k < N; ++k) s += SS_E(A, i, k, N) * at[k];
      SS_E(Phi, i, j, N) = s;
    }
  }
  // Z+ = M Phi (lower triangle; symmetric in exact arithmetic), W+ = (I + Z+)^-1
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_FULL
    for (int i = j; i < N; ++i) {
      double s = 0.0;
      SS_UNROLL_FULL
      for (int k = 0; k < N; ++k) s += M[sy<N>(i, k)] * SS_E(Phi, k, j, N);
      Wp[lo<N>(i, j)] = s;
    }
  }
  inverse_of_identity_plus<N>(Wp);
  double Wm[kTri<N>];
  rank_update<N>(AiT, fm, Wm);
  inverse_of_identity_plus<N>(Wm);
  SS_UNROLL_FULL
  for (int k = 0; k < kTri<N>; ++k) {
    Rh[k] = Wp[k] + Wm[k];
    Th[k] = Wp[k] - Wm[k];
  }
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) Rh[lo<N>(j, j)] -= 1.0;
}

// calc_matrices_lw_eig in symmetrised coordinates for an N sub-problem embedded at stream offset I0 of
// the (NF x NF) record; same outputs, same (unscaled) record layout as calc_matrices_lw.
template <int NF, int N, int I0, class W, class Lds>
SS_HD int calc_matrices_lw_sym(const double* g1, const double* g2, const double* b, const double* dsc, double dz,
                               const W& w, const Lds& lds) {
  using Rec = LwRec<NF>;
  double C[kTri<N>], M[kTri<N>], size;
  {
    double GD[N * N];
    SS_UNROLL_NN(N)
    for (int k = 0; k < N * N; ++k) GD[k] = g1[k] - g2[k];
    size = build_CM<N>(GD, g2, dsc, C, M);
  }
  if (!(size > 0.0)) return -1;
  // particular solution g^ = M^-1 b^ (Cholesky of a copy of M: M itself is needed for K)
  double gh[N];
  {
    double F[kTri<N>], rf[N];
    SS_UNROLL_FULL
    for (int k = 0; k < kTri<N>; ++k) F[k] = M[k];
    if (!cholesky<N>(F, rf, size)) return -1;
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) gh[i] = b[i] / dsc[i];
    solve_L<N>(F, rf, gh);
    solve_LT<N>(F, rf, gh);
  }
  double sig[N], A[N * N], AiT[N * N];
  const int ierr = spectral<N>(C, M, size, sig, A, AiT, lds);
  if (ierr < 0) return -1;
  double Rh[kTri<N>], Th[kTri<N>], Wp[kTri<N>], Phi[N * N];
  reflect_transmit<N>(M, A, AiT, sig, 0.5 * dz, Rh, Th, Wp, Phi);
  double rd[N];
  SS_UNROLL_FULL
  for (int i = 0; i < N; ++i) rd[i] = 1.0 / dsc[i];
  {
    double t[N * N];
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) SS_E(t, i, j, N) = (dsc[i] * rd[j]) * Rh[sy<N>(i, j)];
    }
    put_embedded<NF, NF, N, N, I0, I0>(w, Rec::R, t);
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) SS_E(t, i, j, N) = (dsc[i] * rd[j]) * Th[sy<N>(i, j)];
    }
    put_embedded<NF, NF, N, N, I0, I0>(w, Rec::T, t);
  }
  // source = 2 Z+ W+ g^ = 2 M (Phi (W+ g^)): the small numbers come from M
  double y[N], py[N], srch[N];
  symv<N>(Wp, gh, y);
  mat_vec<N, N>(Phi, y, py);
  symv<N>(M, py, srch);
  put_embedded_vec<NF, N, I0>(w, Rec::SRC, [&](int i) { return dsc[i] * (2.0 * srch[i]); });
  // integrated flux matrix IF = 2 M^-1 Z+ W+ = 2 Phi W+ and integrated source 2 (g^ dz - IF g^) = 2 (g^ dz - 2 Phi y)
  {
    double IF[N * N];
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) {
        double s2 = 0.0;
        SS_UNROLL_FULL
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
// (NF, DF) record; same outputs and record layout as calc_matrices_sw.  The direct-beam part (Gamma0: D x D)
// is shared with the general route (sw_direct_embedded); the direct-diffuse part, a (2n + d) square system
// in the reference (direct_diffuse_part, sw_eig :303-386), is four symmetric matrix-vector products per
// direct eigenvalue here.
template <int NF, int DF, int N, int D, int I0, int R0, class W, class Lds>
SS_HD int calc_matrices_sw_sym(const double* g0, const double* g1, const double* g2, const double* g3,
                               const double* dsc0, const double* dsc, double dz, const W& w, const Lds& lds) {
  using Rec = SwRec<NF, DF>;
  double C[kTri<N>], M[kTri<N>], size;
  {
    double GD[N * N];
    SS_UNROLL_NN(N)
    for (int k = 0; k < N * N; ++k) GD[k] = g1[k] - g2[k];
    size = build_CM<N>(GD, g2, dsc, C, M);
  }
  if (!(size > 0.0)) return -1;
  double sig[N], A[N * N], AiT[N * N];
  int ierr = spectral<N>(C, M, size, sig, A, AiT, lds);
  if (ierr < 0) return -1;

  double mu[D], G0[D * D], G0inv[D * D], emu[D];
  ierr += sw_direct_embedded<NF, DF, D, R0>(g0, dsc0, dz, w, mu, G0, G0inv, emu);

  double Rh[kTri<N>], Th[kTri<N>];
  {
    double Wp[kTri<N>], Phi[N * N];
    reflect_transmit<N>(M, A, AiT, sig, 0.5 * dz, Rh, Th, Wp, Phi);
  }
  double rd[N];
  SS_UNROLL_FULL
  for (int i = 0; i < N; ++i) rd[i] = 1.0 / dsc[i];
  {
    double t[N * N];
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) SS_E(t, i, j, N) = (dsc[i] * rd[j]) * Rh[sy<N>(i, j)];
    }
    put_embedded<NF, NF, N, N, I0, I0>(w, Rec::R, t);
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) SS_E(t, i, j, N) = (dsc[i] * rd[j]) * Th[sy<N>(i, j)];
    }
    put_embedded<NF, NF, N, N, I0, I0>(w, Rec::T, t);
  }
  // Particular solutions of the direct sources (sw_eig :231-253), one per direct eigenvalue mu, scaled:
  //   p^ = g3^ + g4^ = A c,  q^ = g3^ - g4^ = mu A^-T c,  c = (sigma^2 - mu^2)^-1 A^T 2 s^,  s^ = D^-1 (Gamma3 G0)(:, jd)
  // (the eigenvalues sigma^2 are the Rayleigh quotients with M of spectral(): in a transparent layer the
  // unattenuated direct mode, mu^2 ~ 1e-10, meets eigenvalues of the same size in this division)
  double su[N * D], sd[N * D];
  SS_UNROLL_FULL
  for (int jd = 0; jd < D; ++jd) {
    double sh[N];
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) {
      double t = 0.0;
      SS_UNROLL_FULL
      for (int k = 0; k < D; ++k) t += SS_E(g3, i, k, N) * SS_E(G0, k, jd, D);
      sh[i] = 2.0 * t * rd[i];
    }
    double c[N];
    SS_UNROLL_FULL
    for (int k = 0; k < N; ++k) {
      double t = 0.0;
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) t += SS_E(A, i, k, N) * sh[i];
      c[k] = t / (sig[k] - mu[jd] * mu[jd]);       // sig = sigma^2
    }
    double g3h[N], g4h[N];
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) {
      double pv = 0.0, qv = 0.0;
      SS_UNROLL_FULL
      for (int k = 0; k < N; ++k) {
        pv += SS_E(A, i, k, N) * c[k];
        qv += SS_E(AiT, i, k, N) * c[k];
      }
      qv *= mu[jd];
      g3h[i] = 0.5 * (pv + qv);
      g4h[i] = 0.5 * (pv - qv);
    }
    // S_up = g3 - R g4 - T g3 e,  S_dn = g4 e - T g4 - R g3 e   (e = exp(mu dz))
    double g3e[N], r4[N], t3[N], t4[N], r3[N];
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) g3e[i] = g3h[i] * emu[jd];
    symv<N>(Rh, g4h, r4);
    symv<N>(Th, g3e, t3);
    symv<N>(Th, g4h, t4);
    symv<N>(Rh, g3e, r3);
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) {
      SS_E(su, i, jd, N) = dsc[i] * (g3h[i] - r4[i] - t3[i]);
      SS_E(sd, i, jd, N) = dsc[i] * (g4h[i] * emu[jd] - t4[i] - r3[i]);
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
