// ss_layer16.hpp - layer matrices of the orders 7..16 with one problem on the 16 lanes of a DPP
// row (ss_row16.hpp): calc_matrices_lw_eig / calc_matrices_sw_eig of the reference
// (radtool/radtool_calc_matrices_lw_eig.F90:32-230, radtool_calc_matrices_sw_eig.F90:30-386) in the
// formulation of ss_layer_matrices.hpp (Hogan 2019 Eqs. 45-49; block elimination of the direct
// system; particular solutions from the eigen-decomposition), with the eigen-decomposition taken
// from the eigen record as the cooperative Jacobi kernel left it (ss_eigen_coop.hpp).
// A unit = the full problem of one (column, band, layer); the sub-problems of layers without a
// vegetated or without a clear region stay with the one-lane kernels (status 0 in the record).
#pragma once

#include "ss_common.hpp"
#include "ss_eigen_coop.hpp"
#include "ss_layer_matrices.hpp"
#include "ss_row16.hpp"

namespace ss {
namespace r16 {

// Addresses.  Element e of a record sits at rec[(e & ~1) * stride + (e & 1)] (16-byte pairs, tiles of 64
// units).  For the blocks the row-of-16 kernels move, the lane's row index i and the column j separate:
//   column-major N x M, N and off even:  e = off + j N + i  ->  [(i & ~1) stride + (i & 1)]  +  (off + j N) stride
//   row-major N x M,    M and off even:  e = off + i M + j  ->  [i M stride]  +  (off + (j & ~1)) stride + (j & 1)
// i.e. ONE lane-dependent pointer per block plus compile-time offsets.  Left to itself the compiler keeps
// a lane-dependent 64-bit offset per ELEMENT as a loop invariant (dozens of register pairs, spilled and
// reloaded one dependent scratch round trip after the other in the sweeps); the device versions below
// therefore form the lane pointer first and hide it from the optimiser (SS_PIN).  Rows beyond the
// order load the last row's data (no branch) and are zeroed.
#if defined(__HIP_DEVICE_COMPILE__)
#define SS_R16_FAST_ADDR 1
#endif

// own row of a ROW-major N x M block of a record (rows beyond N: zero)
template <int N, int M>
SS_HD void load_rowmajor(const double* rec, long stride, int off, rv* A) {
#ifdef SS_R16_FAST_ADDR
  if constexpr (M % 2 == 0) {
    const int i = lane(), ic = i < N ? i : N - 1;
    const double* p = rec + (long)(ic * M) * stride;
    SS_PIN(p);
    SS_UNROLL_FULL
    for (int j = 0; j < M; ++j) {
      const double v = p[(long)(off + (j & ~1)) * stride + (j & 1)];
      A[j] = i < N ? v : 0.0;
    }
    return;
  }
#endif
  SS_UNROLL_FULL
  for (int j = 0; j < M; ++j)
    A[j] = gen([&](int i) { return (i < N) ? rec_elem_load(rec, stride, off + i * M + j) : 0.0; });
}
// own row of a COLUMN-major N x M block (element (i, j) at off + i + j N)
template <int N, int M>
SS_HD void load_colmajor(const double* rec, long stride, int off, rv* A) {
#ifdef SS_R16_FAST_ADDR
  if constexpr (N % 2 == 0) {
    const int i = lane(), ic = i < N ? i : N - 1;
    const double* p = rec + (long)(ic & ~1) * stride + (ic & 1);
    SS_PIN(p);
    SS_UNROLL_FULL
    for (int j = 0; j < M; ++j) {
      const double v = p[(long)(off + j * N) * stride];
      A[j] = i < N ? v : 0.0;
    }
    return;
  }
#endif
  SS_UNROLL_FULL
  for (int j = 0; j < M; ++j)
    A[j] = gen([&](int i) { return (i < N) ? rec_elem_load(rec, stride, off + i + j * N) : 0.0; });
}
template <int N, int M>
SS_HD void store_colmajor(double* rec, long stride, int off, const rv* A) {
#ifdef SS_R16_FAST_ADDR
  if constexpr (N % 2 == 0) {
    const int i = lane();
    if (i < N) {
      double* p = rec + (long)(i & ~1) * stride + (i & 1);
      SS_PIN(p);
      SS_UNROLL_FULL
      for (int j = 0; j < M; ++j) p[(long)(off + j * N) * stride] = A[j];
    }
    return;
  }
#endif
  SS_UNROLL_FULL
  for (int j = 0; j < M; ++j)
    each(A[j], [&](int i, double x) { if (i < N) rec_elem_store(rec, stride, off + i + j * N, x); });
}
// own row of an N x M matrix -> ROW-major block (16-byte pairs along the row where M is even)
template <int N, int M>
SS_HD void store_rowmajor(double* rec, long stride, int off, const rv* A) {
#ifdef SS_R16_FAST_ADDR
  if constexpr (M % 2 == 0) {
    const int i = lane();
    if (i < N) {
      double* p = rec + (long)(i * M) * stride;
      SS_PIN(p);
      SS_UNROLL_FULL
      for (int j = 0; j < M; j += 2) rec_pair_store(p + (long)(off + j) * stride, A[j], A[j + 1]);
    }
    return;
  }
#endif
  SS_UNROLL_FULL
  for (int j = 0; j < M; ++j)
    each(A[j], [&](int i, double x) { if (i < N) rec_elem_store(rec, stride, off + i * M + j, x); });
}
template <int N>
SS_HD rv load_vec(const double* rec, long stride, int off) {
  return gen([&](int i) { const int ic = i < N ? i : N - 1; const double v = rec_elem_load(rec, stride, off + ic); return (i < N) ? v : 0.0; });
}
template <int N>
SS_HD void store_vec(double* rec, long stride, int off, const rv& x) {
  each(x, [&](int i, double v) { if (i < N) rec_elem_store(rec, stride, off + i, v); });
}

// what a lane knows about its row i = (region, stream)
template <int NREG, int NS>
struct RowGeom {
  rm sm[NS];       // stream of the own row is s (rows beyond the order: never)
  rm rg[NREG];     // region of the own row is r
  rv hw;           // hweight of the own stream
  SS_HD explicit RowGeom(const LG& lg) {
    constexpr int N = NREG * NS;
    SS_UNROLL_FULL
    for (int s = 0; s < NS; ++s) sm[s] = gen_mask([=](int i) { return i < N && (i % NS) == s; });
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) rg[r] = gen_mask([=](int i) { return i < N && (i / NS) == r; });
    hw = 0.0;
    SS_UNROLL_FULL
    for (int s = 0; s < NS; ++s) hw = select(sm[s], rv(lg.hweight[s]), hw);
  }
  // own row of an NREG-row matrix given on every lane (column-major, NREG rows, NC columns)
  template <int NC>
  SS_HD void own_row(const double* U, rv* urow) const {
    SS_UNROLL_FULL
    for (int c = 0; c < NC; ++c) {
      rv u = 0.0;
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) u = select(rg[r], rv(SS_E(U, r, c, NREG)), u);
      urow[c] = u;
    }
  }
};

// Scalars of a layer the Gamma matrices are built from, as the first pass of the layer kernel leaves
// them in the eigen record (slot G1 of the record, which holds no matrix any more)
template <int NREG>
struct GammaScalars {
  static constexpr int EXT = 0, SSA = NREG, FW = 2 * NREG, FX = 3 * NREG, WEXT = 3 * NREG + NREG * NREG,
                       WFAC = WEXT + 1, COUNT = WFAC + 1;
  double ext[NREG], ssa[NREG], fw[NREG], fx[NREG * NREG], wall_ext, wall_factor;
  SS_HD void load(const double* rec, long stride, int off) {
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) {
      ext[r] = rec_elem_load(rec, stride, off + EXT + r);
      ssa[r] = rec_elem_load(rec, stride, off + SSA + r);
      fw[r] = rec_elem_load(rec, stride, off + FW + r);
    }
    SS_UNROLL_FULL
    for (int k = 0; k < NREG * NREG; ++k) fx[k] = rec_elem_load(rec, stride, off + FX + k);
    wall_ext = rec_elem_load(rec, stride, off + WEXT);
    wall_factor = rec_elem_load(rec, stride, off + WFAC);
  }
  SS_HD void store(double* rec, long stride, int off) const {
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) {
      rec_elem_store(rec, stride, off + EXT + r, ext[r]);
      rec_elem_store(rec, stride, off + SSA + r, ssa[r]);
      rec_elem_store(rec, stride, off + FW + r, fw[r]);
    }
    SS_UNROLL_FULL
    for (int k = 0; k < NREG * NREG; ++k) rec_elem_store(rec, stride, off + FX + k, fx[k]);
    rec_elem_store(rec, stride, off + WEXT, wall_ext);
    rec_elem_store(rec, stride, off + WFAC, wall_factor);
  }
};

// Own rows of the exchange / extinction part E = Gamma1 - Gamma2 and of Gamma2 (forest :329-376 /
// urban :383-440 of the longwave solvers, :329-387 / :424-497 of the shortwave ones; the expressions
// of lw_gamma_elem / sw_gamma_elem, evaluated for the own row).  LWFORM: the association of the forest
// longwave scattering term, (0.5 w / mu) ext ssa, instead of 0.5 w ext ssa / mu.
template <bool URBAN, int NREG, int NS, bool LWFORM>
SS_HD void gamma_rows(const LG& lg, const RowGeom<NREG, NS>& g, const GammaScalars<NREG>& q, rv* E, rv* G2) {
  constexpr int N = NREG * NS;
  rv tan_i = 0.0, w_i = 0.0, vw_i = 0.0, mu_i = 1.0, ext_i = 0.0, ssa_i = 0.0, fw_i = 0.0;
  SS_UNROLL_FULL
  for (int s = 0; s < NS; ++s) {
    tan_i = select(g.sm[s], rv(lg.tan_ang[s]), tan_i);
    w_i = select(g.sm[s], rv(lg.weight[s]), w_i);
    vw_i = select(g.sm[s], rv(lg.vweight[s]), vw_i);
    mu_i = select(g.sm[s], rv(lg.mu[s]), mu_i);
  }
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) {
    ext_i = select(g.rg[r], rv(q.ext[r]), ext_i);
    ssa_i = select(g.rg[r], rv(q.ssa[r]), ssa_i);
    fw_i = select(g.rg[r], rv(q.fw[r]), fw_i);
  }
  rv fxrow[NREG], fxcol[NREG];
  SS_UNROLL_FULL
  for (int k = 0; k < NREG; ++k) {
    rv a = 0.0, b = 0.0;
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) {
      a = select(g.rg[r], rv(SS_E(q.fx, r, k, NREG)), a);     // fx(jto = own, jfr = k)
      b = select(g.rg[r], rv(SS_E(q.fx, k, r, NREG)), b);     // fx(jto = k, jfr = own)
    }
    fxrow[k] = a;
    fxcol[k] = b;
  }
  rv diag = 0.0;
  SS_UNROLL_FULL
  for (int k = 0; k < NREG; ++k) diag = select(g.rg[k], diag, diag - tan_i * fxcol[k]);
  if (URBAN) diag = diag - ext_i / mu_i - tan_i * fw_i * q.wall_ext;
  else diag = diag - ext_i / mu_i;
  rv g2s[NS];
  SS_UNROLL_FULL
  for (int sj = 0; sj < NS; ++sj) {
    if (URBAN) g2s[sj] = 0.5 * (w_i * ext_i * ssa_i / lg.mu[sj] + vw_i * lg.tan_ang[sj] * fw_i * q.wall_factor);
    else if (LWFORM) g2s[sj] = (0.5 * w_i / lg.mu[sj]) * ext_i * ssa_i;
    else g2s[sj] = 0.5 * w_i * ext_i * ssa_i / lg.mu[sj];
  }
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    const int rj = j / NS, sj = j % NS;
    E[j] = select(lane_eq(j), diag, select(g.sm[sj], select(g.rg[rj], rv(0.0), tan_i * fxrow[rj]), rv(0.0)));
    G2[j] = select(g.rg[rj], g2s[sj], rv(0.0));
  }
}

// Symmetric elimination of an s.p.d. matrix with its rows on the lanes (both triangles are kept up
// to date, so that the multiplier of row i in step j is its own element j): on return Lr = own row of
// the Cholesky factor L (A = L L^T), A = own row of L^T scaled by the own pivot root (row i of L^T is
// A[k] * rs_own for k >= i), rs[j] = 1 / sqrt(pivot j) on every lane.  Returns false when a pivot is
// not above "thr" (the lanes of a row agree).
template <int N>
SS_HD bool cholesky_rows(rv* A, rv* Lr, double* rs, double thr) {
  bool ok = true;
  static_for<0, N>([&](auto jc) {
    constexpr int J = decltype(jc)::value;
    const double piv = bc<J>(A[J]);
    if (!(piv > thr)) ok = false;
    const double r = 1.0 / sqrt(ok ? piv : 1.0);
    rs[J] = r;
    const rv l = A[J] * r;
    Lr[J] = select(lane_lt(J), rv(0.0), l);
    row_fma_from<J, J + 1, N>(A, A, select(lane_gt(J), -(l * r), rv(0.0)));
  });
  return ok;
}

// The factors the cooperative Jacobi kernel and the layer kernel need (what coop_eig_pre does on one
// lane): with C = -D^-1 (G1 - G2) D = L L^T and M = C - 2 D^-1 G2 D = F F^T: W = F^T L, D L, -D L^-T.
// In: the own rows of G1 - G2 and G2, dscale (and b) from the record.  A unit whose matrices are not positive definite gets status 0 (second pass, general solver).
// Longwave: also inv_gamma_b = -(G1 + G2)^-1 b (lw_eig :178-195), in place of b: G1 + G2 = -D M D^-1, so it
// is D F^-T F^-1 D^-1 b - two substitutions with the factor that is at hand here, where the layer kernel
// used to factorise G1 + G2.
// (E = own row of G1 - G2, g2 = own row of G2; both are consumed)
// The scaled matrices C = -D^-1 (G1 - G2) D and M = C - 2 D^-1 G2 D (own rows; rows beyond the order zero), the scale
// of the pivot test and d = the scaling on every lane, rd = the own reciprocal.  (E = own row of G1 - G2, g2 of G2)
template <int N>
SS_HD void scaled_CM(const rv& dsc, const rv* E, const rv* g2, rv* C, rv* M, double& size, rv& rd) {
  double d[N];
  all_of<N>(dsc, d);
  rd = select(lane_lt(N), 1.0 / (dsc + select(lane_lt(N), rv(0.0), rv(1.0))), rv(0.0));
  size = 0.0;             // scale of the pivot test: the largest element of the scaled matrices
  rv big = 0.0;           // largest |element| of the two scaled matrices in the own row
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    const rv sc = rd * d[j];
    const rv sj = g2[j] * sc;
    C[j] = -(E[j] * sc);
    M[j] = C[j] - 2.0 * sj;
    big = r_max(big, r_max(r_abs(C[j]), r_abs(sj)));
  }
  static_for<0, N>([&](auto kc) { size = ss_max(size, bc<decltype(kc)::value>(big)); });
  // rows beyond the order: identity (their pivots are never used, but keep the arithmetic finite)
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    C[j] = select(lane_lt(N), C[j], rv(0.0));
    M[j] = select(lane_lt(N), M[j], rv(0.0));
  }
}
// Own rows of D L and of -D L^-T from the factorisation of C by cholesky_rows (C: what it left of the matrix -
// the own row of L^T scaled by the own pivot root; L: the own row of the factor; rsL: the reciprocal pivot roots)
template <int N>
SS_HD void factor_rows(const rv& dsc, const rv* C, const rv* L, const double* rsL, rv* DL, rv* AX) {
  const rv rsL_own = own<N>(rsL);
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) DL[j] = L[j] * dsc;
  // -D L^-T: R = L^T is upper triangular with its rows on the lanes; R X = I by back substitution
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) AX[j] = select(lane_eq(j), rv(1.0), rv(0.0));
  static_for_down<1, N>([&](auto kc) {
    constexpr int K = decltype(kc)::value;
    // R(i, K) = C[K] * rsL_own for i <= K; x_K = (row K of X) / R(K, K) = (row K) * rsL[K]
    const rv u = select(lane_lt(K), (C[K] * rsL_own) * rsL[K], rv(0.0));
    row_fma<K, N>(AX, AX, -u);
  });
  const rv sc = -(dsc * rsL_own);
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) AX[j] = AX[j] * sc;
}

// STORE_FACTORS: D L and -D L^-T go to the record (the kernel-level entries, whose layer step reads them back);
// the solver's layer kernel recomputes them from the layer's scalars instead (load_eigenvectors_scalars): k_pre16 is
// bound by its writes and k_layer16 by its 8-byte record reads, and a Cholesky factorisation on the lanes costs
// neither of them anything.
template <int N, bool LW, bool STORE_FACTORS = true>
SS_HD void eig_pre_rows(double* erec, long stride, const rv* E, const rv* g2, unsigned long long* one_lane_units) {
  using ER = EigRec<N>;
  const rv dsc = load_vec<N>(erec, stride, ER::VEC);
  rv rd;
  rv C[N], M[N];
  double size;
  scaled_CM<N>(dsc, E, g2, C, M, size, rd);
  rv L[N], F[N];
  double rsL[N], rsF[N];
  bool ok = cholesky_rows<N>(C, L, rsL, 1.0e-14 * size);
  ok = cholesky_rows<N>(M, F, rsF, 1.0e-14 * size) && ok;
  if (!ok) {
    if (lane0()) {
      rec_elem_store(erec, stride, ER::STATUS, 0.0);
#if defined(__HIP_DEVICE_COMPILE__)
      if (one_lane_units) atomicAdd(one_lane_units, 1ULL);
#else
      if (one_lane_units) *one_lane_units += 1ULL;
#endif
    }
    return;
  }
  // own rows of F^T (M holds them scaled by the own pivot root) and of L^T
  const rv rsF_own = own<N>(rsF);
  {
    rv FT[N], W[N];
    SS_UNROLL_FULL
    for (int k = 0; k < N; ++k) FT[k] = select(lane_gt(k), rv(0.0), M[k] * rsF_own);
    if (LW) {
      // F y = D^-1 b downward, F^T z = y upward (row k is used unscaled: x_k = (row k) / F_kk), igb = D z
      rv v = load_vec<N>(erec, stride, ER::VEC + N) * rd;
      static_for<0, N - 1>([&](auto kc) {
        constexpr int K = decltype(kc)::value;
        fma_bc<K>(v, v, select(lane_gt(K), -(F[K] * rsF[K]), rv(0.0)));
      });
      v = v * rsF_own;
      static_for_down<1, N>([&](auto kc) {
        constexpr int K = decltype(kc)::value;
        fma_bc<K>(v, v, select(lane_lt(K), -(FT[K] * rsF[K]), rv(0.0)));
      });
      store_vec<N>(erec, stride, ER::VEC + N, (v * rsF_own) * dsc);
    }
    mul<N, N, false, false>(FT, nullptr, L, W);
    store_rowmajor<N, N>(erec, stride, ER::W, W);
  }
  if constexpr (STORE_FACTORS) {
    rv DL[N], X[N];
    factor_rows<N>(dsc, C, L, rsL, DL, X);
    store_rowmajor<N, N>(erec, stride, ER::X, DL);
    store_rowmajor<N, N>(erec, stride, ER::AX, X);
  }
}

// ... from Gamma1, Gamma2 in the record (the kernel-level entries, ss_units_capi.hip)
template <int N, bool LW>
SS_HD void eig_pre(double* erec, long stride, unsigned long long* one_lane_units) {
  using ER = EigRec<N>;
  rv E[N], g2[N];
  load_rowmajor<N, N>(erec, stride, ER::G1, E);
  load_rowmajor<N, N>(erec, stride, ER::G2, g2);
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) E[j] = E[j] - g2[j];
  eig_pre_rows<N, LW>(erec, stride, E, g2, one_lane_units);
}
// ... from the layer's scalars in the record (the solver: first pass of the layer kernel, layer_first_sw / _lw)
template <bool URBAN, int NREG, int NS, bool LW>
SS_HD void eig_pre_scalars(const LG& lg, double* erec, long stride, unsigned long long* one_lane_units) {
  constexpr int N = NREG * NS;
  using ER = EigRec<N>;
  GammaScalars<NREG> q;
  q.load(erec, stride, ER::G1);
  const RowGeom<NREG, NS> g(lg);
  rv E[N], g2[N];
  gamma_rows<URBAN, NREG, NS, LW>(lg, g, q, E, g2);
#if defined(SS_LAYER16_FACTORS_FROM_RECORD)
  eig_pre_rows<N, LW, true>(erec, stride, E, g2, one_lane_units);
#else
  eig_pre_rows<N, LW, false>(erec, stride, E, g2, one_lane_units);
#endif
}

// Eigenvectors X = (D L) Y and (G1 - G2)^-1 X = (-D L^-T) Y from the factors and the rotations of the record
template <int N>
SS_HD void load_eigenvectors(const double* erec, long stride, rv* X, rv* AX) {
  using ER = EigRec<N>;
  rv Y[N], F[N];
  load_rowmajor<N, N>(erec, stride, ER::W, Y);
  load_rowmajor<N, N>(erec, stride, ER::X, F);
  mul<N, N, false, false>(F, nullptr, Y, X);
  load_rowmajor<N, N>(erec, stride, ER::AX, F);
  mul<N, N, false, false>(F, nullptr, Y, AX);
}

// The same with the factors recomputed from the layer's scalars (the solver: k_pre16 stores W only): Gamma rows,
// C, its Cholesky factor, D L and -D L^-T - the arithmetic of eig_pre_rows, operation for operation, so the factor
// is the one W was formed with
template <bool URBAN, int NREG, int NS, bool LW>
struct EigenvectorsFromScalars {
  const LG* lg;
  template <int N>
  SS_HD void operator()(const double* erec, long stride, rv* X, rv* AX) const {
    static_assert(N == NREG * NS, "order");
    using ER = EigRec<N>;
#if defined(SS_LAYER16_FACTORS_FROM_RECORD)
    load_eigenvectors<N>(erec, stride, X, AX);
#else
    rv DL[N], AXr[N];
    {
      GammaScalars<NREG> q;
      q.load(erec, stride, ER::G1);
      const RowGeom<NREG, NS> g(*lg);
      rv E[N], g2[N], C[N], M[N], L[N];
      gamma_rows<URBAN, NREG, NS, LW>(*lg, g, q, E, g2);
      const rv dsc = load_vec<N>(erec, stride, ER::VEC);
      rv rd;
      double size, rsL[N];
      scaled_CM<N>(dsc, E, g2, C, M, size, rd);
      cholesky_rows<N>(C, L, rsL, 1.0e-14 * size);       // (status 1: k_pre16 ran the same factorisation and it held)
      factor_rows<N>(dsc, C, L, rsL, DL, AXr);
    }
    rv Y[N];
    load_rowmajor<N, N>(erec, stride, ER::W, Y);
    mul<N, N, false, false>(DL, nullptr, Y, X);
    mul<N, N, false, false>(AXr, nullptr, Y, AX);
#endif
  }
};
// (the kernel-level entries: factors from the record)
struct EigenvectorsFromRecord {
  template <int N>
  SS_HD void operator()(const double* erec, long stride, rv* X, rv* AX) const { load_eigenvectors<N>(erec, stride, X, AX); }
};

// Diffuse part (sw_eig :180-221, lw_eig :142-176): from eigenvalues, X and (G1 - G2)^-1 X of the
// record to the eigenvector blocks G1, G2, lam, e = exp(-lam dz), LU(G1), C'_lower, C'_upper; R and T
// go to the layer record (column-major blocks at offR, offT).
template <int N>
struct Diffuse {
  rv G1[N], G2[N], Clo[N], Cup[N];
  LU<N> luG1;
  rv lam_d, e_d;
};

// B = diag(e) A (e one element per lane: the own row scaled)
template <int M>
SS_HD void scale_rows(const rv* A, const rv& e_d, rv* B) {
  SS_UNROLL_FULL
  for (int j = 0; j < M; ++j) B[j] = A[j] * e_d;
}

template <int N>
SS_HD void diffuse_part(const rv* X, const rv* AX, const rv& ev_d, double dz, Diffuse<N>& w, double* lrec,
                        long stride, int offR, int offT) {
  w.lam_d = r_sqrt(r_max(ev_d, 0.0));
  w.e_d = r_exp(-(w.lam_d * dz));
  rv t1[N];
  {
    // tmp = -((G1 - G2)^-1 X) diag(lam);  G1 = X + tmp, G2 = X - tmp
    double lam[N];
    all_of<N>(w.lam_d, lam);
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      const rv t = -(AX[j] * lam[j]);
      w.G1[j] = X[j] + t;
      w.G2[j] = X[j] - t;
      w.luG1.a[j] = w.G1[j];
      t1[j] = w.G2[j];
    }
  }
  // Eq. 48 by Schur complement; G1D = G1 diag(e), G2D = G2 diag(e) are never formed: the scaling
  // goes to the rows of the right-hand factor (one multiplication by the own element of e)
  lu_factor<N>(w.luG1);
  lu_solve<N, N>(w.luG1, t1);                               // G1^-1 G2
  {
    double e[N];
    all_of<N>(w.e_d, e);
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) t1[j] = t1[j] * (w.e_d * e[j]);   // diag(e) G1^-1 G2 diag(e)
  }
  {
    LU<N> S;
    mul<N, N, false, false>(w.G2, nullptr, t1, S.a);        // G2D G1^-1 G2D
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      S.a[j] = w.G1[j] - S.a[j];
      w.Clo[j] = select(lane_eq(j), rv(1.0), rv(0.0));
    }
    lu_factor<N>(S);
    lu_solve<N, N>(S, w.Clo);                               // C'_lower = S^-1
  }
  scale_rows<N>(w.Clo, w.e_d, t1);
  mul<N, N, false, false>(w.G2, nullptr, t1, w.Cup);        // G2D Clo
  lu_solve<N, N>(w.luG1, w.Cup);
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) w.Cup[j] = -w.Cup[j];         // C'_upper
  // Eq. 49: R = G1D Cup + G2 Clo ; T = G2 Cup + G1D Clo
  {
    rv acc[N];
    mul<N, N, false, false>(w.G1, nullptr, t1, acc);        // G1D Clo (t1 = diag(e) Clo)
    mul<N, N, true, false>(w.G2, nullptr, w.Cup, acc);
    store_colmajor<N, N>(lrec, stride, offT, acc);
    scale_rows<N>(w.Cup, w.e_d, t1);
    mul<N, N, false, false>(w.G1, nullptr, t1, acc);        // G1D Cup
    mul<N, N, true, false>(w.G2, nullptr, w.Clo, acc);
    store_colmajor<N, N>(lrec, stride, offR, acc);
  }
}

// calc_matrices_lw_eig of a full problem: layer record of the unit from its eigen record
template <int N, class LoadEV = EigenvectorsFromRecord>
SS_HD void layer_lw(const double* erec, long stride, double* lrec, const LoadEV& load_ev = LoadEV{}) {
  using ER = EigRec<N>;
  using LR = LwRec<N>;
  const double dz = rec_elem_load(erec, stride, 1);
  // inv_gamma_b = -(G1 + G2)^-1 b (lw_eig :178-195): left in the record by eig_pre
  const rv igb = load_vec<N>(erec, stride, ER::VEC + N);
  Diffuse<N> df;
  {
    rv X[N], AX[N];
    load_ev.template operator()<N>(erec, stride, X, AX);
    diffuse_part<N>(X, AX, load_vec<N>(erec, stride, ER::EV), dz, df, lrec, stride, LR::R, LR::T);
  }
  // Section 4 (:198-209): cb' and the source
  rv cb;
  {
    rv t[1] = {igb};
    lu_solve<N, 1>(df.luG1, t);                              // g1^-1 inv_gamma_b
    rv v;
    mat_vec<N, false>(df.G2, t[0] * df.e_d, v);              // G2D g1^-1 igb
    mat_vec<N, false>(df.Clo, igb - v, cb);
    cb = -cb;
    // source = (G1D + G2) cb' + inv_gamma_b
    mat_vec<N, false>(df.G2, cb, v);
    mat_vec<N, true>(df.G1, df.e_d * cb, v);
    store_vec<N>(lrec, stride, LR::SRC, v + igb);
  }
  // Section 5 (:211-227): integrated fluxes
  {
    double z[N];
    all_of<N>((1.0 - df.e_d) / df.lam_d, z);
    rv GZ[N], IF[N];
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      GZ[j] = df.G1[j] * z[j] + df.G2[j] * z[j];
      df.Clo[j] = df.Clo[j] + df.Cup[j];
    }
    mul<N, N, false, false>(GZ, nullptr, df.Clo, IF);
    store_colmajor<N, N>(lrec, stride, LR::IFLUX, IF);
    rv v;
    mat_vec<N, false>(GZ, cb, v);
    store_vec<N>(lrec, stride, LR::ISRC, 2.0 * (v + igb * dz));
  }
}

// calc_matrices_sw_eig of a full problem (N diffuse, D direct streams): R, T, S_up, S_dn of the layer
// record from the eigen record; int_dir and E were written by the first pass of the layer kernel
// (sw_direct_part), which also left mu, G0, G0^-1, exp(mu dz) of Gamma0 in the eigen record.
template <int N, int D, class LoadEV = EigenvectorsFromRecord>
SS_HD void layer_sw(const double* erec, long stride, double* lrec, const LoadEV& load_ev = LoadEV{}) {
  using ER = EigRec<N>;
  using LR = SwRec<N, D>;
  const double dz = rec_elem_load(erec, stride, 1);
  const rv ev_d = load_vec<N>(erec, stride, ER::EV);
  rv g3v[D], g4v[D];   // particular solutions, one column per direct eigenvalue
  Diffuse<N> df;
  {
    rv X[N], AX[N];
    load_ev.template operator()<N>(erec, stride, X, AX);
    // Section 4 (:231-253) through the eigen-decomposition (see ss_layer_matrices.hpp): sources
    // s = g3 G0 (one column per direct eigenvalue), c = 2 (Lambda - mu^2)^-1 (D^-2 X)^T s,
    // p = X c, q = -mu ((G1 - G2)^-1 X) c, g3 = (p + q) / 2, g4 = (p - q) / 2
    double mu[D], G0[D * D];
    SS_UNROLL_FULL
    for (int r = 0; r < D; ++r) mu[r] = rec_elem_load(erec, stride, ER::vmu(D) + r);
    SS_UNROLL_FULL
    for (int k = 0; k < D * D; ++k) G0[k] = rec_elem_load(erec, stride, ER::vg0(D) + k);
    rv g3[D];
    load_rowmajor<N, D>(erec, stride, ER::vg3(), g3);
    const rv dsc = load_vec<N>(erec, stride, ER::VEC);
    const rv rd2 = select(lane_lt(N), 1.0 / (dsc * dsc), rv(0.0));
    const rv one = 1.0;
    SS_UNROLL_FULL
    for (int jd = 0; jd < D; ++jd) {
      rv s = 0.0;
      SS_UNROLL_FULL
      for (int r = 0; r < D; ++r) s = s + g3[r] * SS_E(G0, r, jd, D);
      s = s * rd2;
      rv t[N], h[N];
      SS_UNROLL_FULL
      for (int k = 0; k < N; ++k) {
        t[k] = X[k] * s;
        h[k] = 0.0;
      }
      // sums over the lanes: h[k] = sum_i X(i, k) s_i / d_i^2, on every lane
      static_for<0, N>([&](auto ic) { row_fma<decltype(ic)::value, N>(h, t, one); });
      // 2 / (lambda_k - mu^2): one division on lane k, then to every lane
      double rden[N];
      all_of<N>(2.0 / (ev_d - mu[jd] * mu[jd]), rden);
      rv pv = 0.0, qv = 0.0;
      SS_UNROLL_FULL
      for (int k = 0; k < N; ++k) {
        const rv c = h[k] * rden[k];
        pv = pv + X[k] * c;
        qv = qv + AX[k] * c;
      }
      const rv q = -(qv * mu[jd]);
      g3v[jd] = 0.5 * (pv + q);
      g4v[jd] = 0.5 * (pv - q);
    }
    // (the particular solutions wait in the S_up / S_dn slots of the record while the diffuse part
    // needs the registers)
    store_colmajor<N, D>(lrec, stride, LR::SUP, g3v);
    store_colmajor<N, D>(lrec, stride, LR::SDN, g4v);
    // Sections 1-2: diffuse R, T
    diffuse_part<N>(X, AX, ev_d, dz, df, lrec, stride, LR::R, LR::T);
  }
  load_colmajor<N, D>(lrec, stride, LR::SUP, g3v);
  load_colmajor<N, D>(lrec, stride, LR::SDN, g4v);
  // direct_diffuse_part (:303-386) by block elimination (header of ss_layer_matrices.hpp)
  {
    double G0inv[D * D], emu[D];
    SS_UNROLL_FULL
    for (int r = 0; r < D; ++r) emu[r] = rec_elem_load(erec, stride, ER::vemu(D) + r);
    SS_UNROLL_FULL
    for (int k = 0; k < D * D; ++k) G0inv[k] = rec_elem_load(erec, stride, ER::vg0inv(D) + k);
    rv b1[D], b2[D], c1[D], c2[D], t[D], te[D], su[D];
    SS_UNROLL_FULL
    for (int j = 0; j < D; ++j) {
      rv s1 = 0.0, s2 = 0.0;
      SS_UNROLL_FULL
      for (int r = 0; r < D; ++r) {
        s1 = s1 + (g3v[r] * emu[r]) * SS_E(G0inv, r, j, D);      // (g3 e) G0^-1 = -(rhs of block 1)
        s2 = s2 + g4v[r] * SS_E(G0inv, r, j, D);                 // g4 G0^-1     = -(rhs of block 2)
      }
      b1[j] = s1;
      b2[j] = s2;
      t[j] = s1;
    }
    // c2 = S^-1 (-b2 + G2D G1^-1 b1)
    lu_solve<N, D>(df.luG1, t);
    SS_UNROLL_FULL
    for (int j = 0; j < D; ++j) te[j] = t[j] * df.e_d;
    mul<N, D, false, false>(df.G2, nullptr, te, c1);
    SS_UNROLL_FULL
    for (int j = 0; j < D; ++j) c1[j] = c1[j] - b2[j];
    mul<N, D, false, false>(df.Clo, nullptr, c1, c2);
    // c1 = G1^-1 (-b1 - G2D c2)
    SS_UNROLL_FULL
    for (int j = 0; j < D; ++j) te[j] = c2[j] * df.e_d;           // diag(e) c2, reused for S_dn
    mul<N, D, false, false>(df.G2, nullptr, te, c1);
    SS_UNROLL_FULL
    for (int j = 0; j < D; ++j) c1[j] = -b1[j] - c1[j];
    lu_solve<N, D>(df.luG1, c1);
    // S_up = G1D c1 + G2 c2 + g3 G0^-1 ; S_dn = G2 c1 + G1D c2 + (g4 e) G0^-1
    mul<N, D, false, false>(df.G2, nullptr, c1, t);
    mul<N, D, true, false>(df.G1, nullptr, te, t);               // S_dn without the particular part
    SS_UNROLL_FULL
    for (int j = 0; j < D; ++j) te[j] = c1[j] * df.e_d;           // diag(e) c1
    mul<N, D, false, false>(df.G1, nullptr, te, su);
    mul<N, D, true, false>(df.G2, nullptr, c2, su);
    SS_UNROLL_FULL
    for (int j = 0; j < D; ++j) {
      rv s1 = su[j], s2 = t[j];
      SS_UNROLL_FULL
      for (int r = 0; r < D; ++r) {
        s1 = s1 + g3v[r] * SS_E(G0inv, r, j, D);
        s2 = s2 + (g4v[r] * emu[r]) * SS_E(G0inv, r, j, D);
      }
      su[j] = s1;
      t[j] = s2;
    }
    store_colmajor<N, D>(lrec, stride, LR::SUP, su);
    store_colmajor<N, D>(lrec, stride, LR::SDN, t);
  }
}

}  // namespace r16
}  // namespace ss
