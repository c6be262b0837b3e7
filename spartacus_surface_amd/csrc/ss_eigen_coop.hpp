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
//    (row i on lane i mod G) every rotation is lane-local; the inner product that defines a
//    rotation is a local part plus a butterfly over the G lanes of the group (the squared column
//    norms are carried along and recomputed once per sweep).
//  * So a 12 x 12 problem lives in the registers of 2 lanes (6 rows of W and of Y each)
//    instead of 17 KB of scratch memory per lane, 32 problems per wave.
// The kernel (k_eig_coop, ss_inst_body.hpp) sits between the kernels that build W and the factors
// D L, -D L^-T (k_pre16, r16::eig_pre_scalars in ss_layer16.hpp: Cholesky by symmetric elimination
// with the matrix rows on the 16 lanes of a DPP row) and the layer kernel that forms X = (D L) Y
// and the layer matrices from them (k_layer16).  The one-lane forms of the two ends (coop_eig_pre,
// coop_eig_post) serve builds without the row-of-16 kernels (-DSS_NO_R16_LAYER).
//
// Group size G: 2 lanes up to order 12, 4 above (coop_lanes).  Orders above 16 (8 streams x 3
// regions) keep the one-kernel, one-lane form.
#pragma once

#include "ss_common.hpp"
#include "ss_layer_matrices.hpp"

namespace ss {

// lanes per problem of the Jacobi kernel: 2 up to order 12 (W and Y are 2 x 72 doubles per lane: the
// update of a rotation is lane-local, the angle is computed once per 2 lanes instead of once per 4;
// 0.72 ms per 200 000 12 x 12 problems against 1.13 ms with 4 lanes), 4 above (registers)
#ifdef SS_COOP_LANES
constexpr int coop_lanes(int) { return SS_COOP_LANES; }
#else
constexpr int coop_lanes(int n) { return n <= 12 ? 2 : 4; }
#endif
// which spectral regions run the layer matrices of the cooperative units on the 16 lanes of a DPP row (ss_layer16.hpp)
constexpr bool r16_layer(int n, bool lw);
constexpr int kCoopMaxOrder = 16;
constexpr bool coop_order(int n) {
#if defined(SS_NO_COOP_EIGEN)
  return false && n > 0;
#else
  return n >= SS_COOP_MIN_ORDER && n <= kCoopMaxOrder;
#endif
}

constexpr bool r16_layer(int n, bool lw) {
#if defined(SS_NO_R16_LAYER)
  return false && n > 0 && lw;
#else
  return coop_order(n) && (lw || !lw);
#endif
}

// Eigen record of one (column, band, layer) unit in the tiling of the scratch records (all
// matrices ROW-major, so that a lane of the cooperative kernels loads its rows):
//   [0] status (1 = a cooperative unit, 0 = not), [1] layer thickness dz,
//   EV    squared singular values (out)
//   W     W = F^T L in, its right singular vectors Y out
//   X     D L:      the eigenvectors are X = (D L) Y
//   AX    -D L^-T:  (G1 - G2)^-1 X = (-D L^-T) Y
//   G1, G2, VEC: the Gamma matrices and vectors of the unit for the layer kernel in the row-of-16
//   layout (ss_layer16.hpp); VEC = [dscale (N)] then, longwave: [b (N)]; shortwave: [g3 (N x D, row-major)]
//   [mu (D)] [eigenvectors of Gamma0 (D x D)] [their inverse (D x D)] [exp(mu dz) (D)]
constexpr int eig_rec_nvec(int n) { return 5 * n + 18; }     // room for either spectral region (D <= 3, n >= 7)
constexpr int eig_rec_nelem(int n) {
  return 2 + rec_even(n) + 5 * rec_even(n * n) + rec_even(eig_rec_nvec(n));
}
template <int N>
struct EigRec {
  static constexpr int STATUS = 0;
  static constexpr int EV = 2;
  static constexpr int W = EV + rec_even(N);
  static constexpr int X = W + rec_even(N * N);
  static constexpr int AX = X + rec_even(N * N);
  static constexpr int G1 = AX + rec_even(N * N);
  static constexpr int G2 = G1 + rec_even(N * N);
  static constexpr int VEC = G2 + rec_even(N * N);
  static constexpr int NVEC = eig_rec_nvec(N);
  static constexpr int NELEM = VEC + rec_even(NVEC);
  static_assert(NELEM == eig_rec_nelem(N), "eig_rec_nelem is what the plan sizes the scratch with");
  // shortwave members of VEC for D direct streams
  static constexpr int vg3() { return VEC + N; }
  static constexpr int vmu(int D) { return VEC + N + N * D; }
  static constexpr int vg0(int D) { return vmu(D) + D; }
  static constexpr int vg0inv(int D) { return vg0(D) + D * D; }
  static constexpr int vemu(int D) { return vg0inv(D) + D * D; }
  static_assert(N < 7 || N + N * 3 + 3 + 9 + 9 + 3 <= NVEC, "VEC too small");
};

// element e of a record, read as a scalar (pairs: element e sits at pair e/2, half e%2)
SS_HD double rec_elem_load(const double* rec, long stride, int e) {
  return rec[(long)(e & ~1) * stride + (e & 1)];
}
// store element e of a record (scalar; elements 2p, 2p+1 share a 16-byte pair)
SS_HD void rec_elem_store(double* rec, long stride, int e, double v) {
  rec[(long)(e & ~1) * stride + (e & 1)] = v;
}

// A column-major N x N matrix of a lane into the ROW-major block "off" (even) of a record: 16-byte
// pairs along a row where the order is even (every row then starts at an even element)
template <int N>
SS_HD void rec_store_rowmajor(double* rec, long stride, int off, const double* a) {
  for (int i = 0; i < N; ++i) {
    if constexpr (N % 2 == 0) {
      for (int j = 0; j < N; j += 2)
        rec_pair_store(rec + (long)(off + i * N + j) * stride, SS_E(a, i, j, N), SS_E(a, i, j + 1, N));
    } else {
      for (int j = 0; j < N; ++j) rec_elem_store(rec, stride, off + i * N + j, SS_E(a, i, j, N));
    }
  }
}

// ---- pass 1 of the layer kernel, one lane per problem: is the structure there? ----
// dscale = D: both D^-1 (G1 - G2) D and D^-1 G2 D must be symmetric.  Returns the
// largest element of the two scaled matrices in "size" (the scale of the pivot test of the Cholesky
// factorisations, which the row-of-16 kernel does: r16::eig_pre, ss_layer16.hpp).
template <int N>
SS_HD bool coop_eig_check(const double* G1, const double* G2, const double* dscale, double& size) {
  bool ok = true;
  double rd[N];
  for (int i = 0; i < N; ++i) {
    if (!(dscale[i] > 0.0)) ok = false;
    rd[i] = 1.0 / (ok ? dscale[i] : 1.0);
  }
  double asym = 0.0;
  size = 0.0;
  for (int j = 0; j < N; ++j) {
    for (int i = j; i < N; ++i) {
      const double c_ij = -(SS_E(G1, i, j, N) - SS_E(G2, i, j, N)) * (dscale[j] * rd[i]);
      const double s_ij = SS_E(G2, i, j, N) * (dscale[j] * rd[i]);
      if (i != j) {
        const double c_ji = -(SS_E(G1, j, i, N) - SS_E(G2, j, i, N)) * (dscale[i] * rd[j]);
        const double s_ji = SS_E(G2, j, i, N) * (dscale[i] * rd[j]);
        asym = ss_max(asym, ss_max(fabs(c_ij - c_ji), fabs(s_ij - s_ji)));
      }
      size = ss_max(size, ss_max(fabs(c_ij), fabs(s_ij)));
    }
  }
  return ok && (asym <= 1.0e-10 * size);
}

// ---- the same plus the factors, one lane per problem (kernel-level entries, builds without the row-of-16 kernels) ----
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
  rec_elem_store(rec, stride, EigRec<N>::STATUS, ok ? 1.0 : 0.0);
  if (!ok) return false;
  // W = F^T L: W(i, j) = sum_{k >= max(i, j)} F(k, i) L(k, j)
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      double w0 = 0.0;
      for (int k = (i > j ? i : j); k < N; ++k) w0 += SS_E(M, k, i, N) * SS_E(C, k, j, N);
      rec_elem_store(rec, stride, EigRec<N>::W + i * N + j, w0);
    }
  // the two matrices the rotations are accumulated into: D L, and -D L^-T (M is free: L^-1 into it)
  for (int j = 0; j < N; ++j) {
    for (int i = 0; i < j; ++i) SS_E(M, i, j, N) = 0.0;
    SS_E(M, j, j, N) = 1.0 / SS_E(C, j, j, N);
    for (int i = j + 1; i < N; ++i) {
      double t = 0.0;
      for (int k = j; k < i; ++k) t -= SS_E(C, i, k, N) * SS_E(M, k, j, N);
      SS_E(M, i, j, N) = t / SS_E(C, i, i, N);
    }
  }
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      rec_elem_store(rec, stride, EigRec<N>::X + i * N + j, (j <= i) ? dscale[i] * SS_E(C, i, j, N) : 0.0);
      rec_elem_store(rec, stride, EigRec<N>::AX + i * N + j, (j >= i) ? -dscale[i] * SS_E(M, j, i, N) : 0.0);
    }
  return true;
}


// ---- pass 2 of the layer kernel: eigenvalues / vectors from the record ----
// Same outputs as eigen_decomposition_gamma_sym: eigenvalue, eigenvector X = (D L) Y, ainv_x =
// A^-1 X = (-D L^-T) Y.  Returns -1 when the record says the unit is not a cooperative one.
// (One lane per problem: the kernel-level entries and builds without the row-of-16 kernels; the
// production path forms the two products in ss_layer16.hpp.)
template <int N>
SS_HD int coop_eig_post(const double* A, const double* dscale, const double* rec, long stride,
                        double* eigenvalue, double* eigenvector, double* ainv_x) {
  (void)A; (void)dscale;
  if (!(rec_elem_load(rec, stride, EigRec<N>::STATUS) > 0.5)) return -1;
  for (int j = 0; j < N; ++j) eigenvalue[j] = rec_elem_load(rec, stride, EigRec<N>::EV + j);
  for (int j = 0; j < N; ++j) {
    double y[N];
    for (int k = 0; k < N; ++k) y[k] = rec_elem_load(rec, stride, EigRec<N>::W + k * N + j);   // Y(k, j)
    for (int i = 0; i < N; ++i) {
      double s = 0.0, t = 0.0;
      for (int k = 0; k <= i; ++k) s += rec_elem_load(rec, stride, EigRec<N>::X + i * N + k) * y[k];
      for (int k = i; k < N; ++k) t += rec_elem_load(rec, stride, EigRec<N>::AX + i * N + k) * y[k];
      SS_E(eigenvector, i, j, N) = s;
      SS_E(ainv_x, i, j, N) = t;
    }
  }
  return 0;
}

// ---- the dense list of cooperative units ----
// In a vegetated-urban column set only part of the unit-layers are full problems (C3 with 4 streams: 60 %; the
// layers without vegetation or without clear air are sub-problems of a smaller order, done by the first pass on
// one lane).  Found by position, the cooperative units leave 40 % of the lanes of the factor, Jacobi and layer
// kernels idle.  The first pass therefore appends every cooperative unit-layer to a list - one atomic per wave, the
// units of a wave in lane order, so that neighbours in the list are neighbours in the record tiling and a
// workgroup still reads whole 128-byte lines of the records - and those three kernels take problem m of the list.
// Launch grids stay what they were (the count is only known on the device): waves beyond the list exit at once.
SS_HD void coop_list_push(unsigned long long* list, bool coop, int jlay, int cu) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (!list) return;
  const unsigned long long mask = __ballot(coop);
  if (mask == 0ULL) return;
  const int lane = (int)(threadIdx.x & 63u);
  const int leader = __ffsll((long long)mask) - 1;
  unsigned long long base = 0ULL;
  if (lane == leader) base = atomicAdd(list, (unsigned long long)__popcll(mask));
  base = __shfl(base, leader);
  if (coop)
    list[1ULL + base + (unsigned long long)__popcll(mask & ((1ULL << lane) - 1ULL))] =
        ((unsigned long long)(unsigned)jlay << 32) | (unsigned long long)(unsigned)cu;
#else
  (void)list; (void)coop; (void)jlay; (void)cu;
#endif
}
// problem m of the list -> (layer, column unit); false: beyond its end
SS_HD bool coop_list_get(const unsigned long long* list, unsigned long long m, int& jlay, int& cu) {
  if (m >= list[0]) return false;
  const unsigned long long e = list[1ULL + m];
  jlay = (int)(e >> 32);
  cu = (int)(e & 0xffffffffULL);
  return true;
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
// One-sided (Hestenes) Jacobi, cyclic by rows; per rotation one inner product over the group (the
// squared column norms are carried along: a' = a - t g, b' = b + t g, and recomputed at the start of
// every sweep), the angle from fast reciprocals, the update of the own rows of W and Y.
// A problem whose Jacobi iteration has not converged after the last sweep, or whose singular values are not
// finite and positive (the fast reciprocals have no fix-up for zero norms), is handed to the second pass of
// the layer kernel like a declined unit: status 0 and one more in the counter "one_lane_units" that wakes that
// pass.  force_redo (diagnostic, SPARTACUS_DECLINE_COOP=2) does that to every problem AFTER the iteration, so
// that the hand-over itself is covered by the goldens.
#if defined(__HIP_DEVICE_COMPILE__)
// the partner lane's value in a group of 2 (quad_perm [1, 0, 3, 2]): two DPP moves instead of two ds_bpermute
// round trips through the LDS pipe
__device__ __forceinline__ double pair_swap(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0xB1, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <>
__device__ __forceinline__ double group_sum<2>(double x) { return x + pair_swap(x); }
#endif

// One-sided Jacobi WITHOUT an accumulated rotation matrix (round 4, second session).  K = W^T W = Y S^2 Y^T is what
// the layer matrices need.  Rotating the columns of W needs Y accumulated beside it (Y = the product of the
// rotations); rotating the columns of G = W^T = L^T F does not: G V = Q S with orthonormal Q gives K = G G^T =
// Q S^2 Q^T, i.e. Y = Q = the rotated columns themselves, normalised.  Half the multiply-adds per rotation, and
// the state of a lane drops from 156 to 84 doubles (order 12, 2 lanes): the kernel fits two waves per SIMD, where
// the scalar instructions of one wave issue beside the vector instructions of the other
// (tools/experiments/issue_probe.hip: one wave alone issues ANY instruction every ~4.8 cycles).
#ifndef SS_JACOBI_ACCUMULATE
#define SS_JACOBI_ACCUMULATE 0      // 1: the form of rounds 2-3 (columns of W, Y accumulated)
#endif
constexpr bool kJacobiAccumulate = SS_JACOBI_ACCUMULATE != 0;

template <int N, int G>
SS_HD void coop_eig_unit(double* rec, long stride, int q, unsigned long long* one_lane_units = nullptr,
                         bool force_redo = false) {
  constexpr int NR = (N + G - 1) / G;
  constexpr int NY = kJacobiAccumulate ? NR : 1;
  using Rec = EigRec<N>;
  if (!(rec_elem_load(rec, stride, Rec::STATUS) > 0.5)) return;   // same for the whole group
  double w[NR][N], y[NY][N], nrm[N];
  SS_UNROLL_FULL
  for (int r = 0; r < NR; ++r) {
    const int i = q + r * G;
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      if constexpr (kJacobiAccumulate) {
        w[r][j] = (i < N) ? rec_elem_load(rec, stride, Rec::W + i * N + j) : 0.0;
        y[r][j] = (i == j) ? 1.0 : 0.0;
      } else {
        w[r][j] = (i < N) ? rec_elem_load(rec, stride, Rec::W + j * N + i) : 0.0;      // G = W^T
      }
    }
  }
  auto gsum = [](double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return group_sum<G>(v);
#else
    return v;
#endif
  };
  // A rotation is applied where |g| > tol sqrt(a b).  The iteration ends after the first sweep in which
  // no pair was above 1e-8 sqrt(a b) BEFORE its rotation: convergence is quadratic, so that sweep
  // left every pair at the 1e-16 level and a further one would only confirm it (it was a seventh of
  // the kernel).  The decision is taken on group-wide sums, so the lanes of a group agree on it.
  constexpr double tol = 4.0 * DBL_EPSILON;
  constexpr double tol_last = 1.0e-8;
  bool still_rotating = true;
  for (int sweep = 0; sweep < 30; ++sweep) {
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      double s = 0.0;
      SS_UNROLL_FULL
      for (int r = 0; r < NR; ++r) s += w[r][j] * w[r][j];
      nrm[j] = gsum(s);
    }
    bool rotated = false;
    SS_UNROLL_FULL
    for (int p = 0; p < N - 1; ++p) {
      SS_UNROLL_FULL
      for (int c = p + 1; c < N; ++c) {
        double apc = 0.0;
        SS_UNROLL_FULL
        for (int r = 0; r < NR; ++r) apc += w[r][p] * w[r][c];
        apc = gsum(apc);
        const double app = nrm[p], acc = nrm[c];
        if (apc * apc > (tol * tol) * (app * acc)) {
          if (apc * apc > (tol_last * tol_last) * (app * acc)) rotated = true;
          // the smaller rotation angle (Rutishauser): t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)),
          // zeta = (acc - app) / (2 apc), written without the division by apc
          const double d = acc - app, g2 = apc + apc;
          double sq, rsq_unused;
          fast_sqrt_rsqrt(d * d + g2 * g2, sq, rsq_unused);
          const double t = ((d >= 0.0) ? g2 : -g2) * fast_rcp(fabs(d) + sq);
          double cs_inv, cs;
          fast_sqrt_rsqrt(1.0 + t * t, cs_inv, cs);
          const double sn = cs * t;
          nrm[p] = app - t * apc;
          nrm[c] = acc + t * apc;
          SS_UNROLL_FULL
          for (int r = 0; r < NR; ++r) {
            const double wp = w[r][p], wc = w[r][c];
            w[r][p] = cs * wp - sn * wc;
            w[r][c] = sn * wp + cs * wc;
            if constexpr (kJacobiAccumulate) {
              const double yp = y[r][p], yc = y[r][c];
              y[r][p] = cs * yp - sn * yc;
              y[r][c] = sn * yp + cs * yc;
            }
          }
        }
      }
    }
    if (!rotated) { still_rotating = false; break; }
  }
  // squared singular values; Y in place of W
  bool bad = still_rotating || force_redo;
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    double s = 0.0;
    SS_UNROLL_FULL
    for (int r = 0; r < NR; ++r) s += w[r][j] * w[r][j];
    s = gsum(s);
    if (!(s > 0.0 && s < 1.0e300)) bad = true;      // (the same on every lane of the group)
    if (q == (j % G)) rec_elem_store(rec, stride, Rec::EV + j, s);
    nrm[j] = s;
  }
  if (bad) {
    if (q == 0) {
      rec_elem_store(rec, stride, Rec::STATUS, 0.0);
      if (one_lane_units) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicAdd(one_lane_units, 1ULL);
#else
        *one_lane_units += 1ULL;
#endif
      }
    }
    return;
  }
  if constexpr (!kJacobiAccumulate) {
    // Y = the rotated columns, normalised (IEEE square root and division: once per column)
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) nrm[j] = 1.0 / sqrt(nrm[j]);
  }
  SS_UNROLL_FULL
  for (int r = 0; r < NR; ++r) {
    const int i = q + r * G;
    if (i < N) {
      SS_UNROLL_FULL
      for (int j = 0; j < N; ++j) {
        if constexpr (kJacobiAccumulate) rec_elem_store(rec, stride, Rec::W + i * N + j, y[r][j]);
        else rec_elem_store(rec, stride, Rec::W + i * N + j, w[r][j] * nrm[j]);
      }
    }
  }
}

}  // namespace ss
