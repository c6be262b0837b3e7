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

// own row of a ROW-major N x M block of a record (rows beyond N: zero)
template <int N, int M>
SS_HD void load_rowmajor(const double* rec, long stride, int off, rv* A) {
  SS_UNROLL_FULL
  for (int j = 0; j < M; ++j)
    A[j] = gen([&](int i) { return (i < N) ? rec_elem_load(rec, stride, off + i * M + j) : 0.0; });
}
// own row of a COLUMN-major N x M block (element (i, j) at off + i + j N)
template <int N, int M>
SS_HD void load_colmajor(const double* rec, long stride, int off, rv* A) {
  SS_UNROLL_FULL
  for (int j = 0; j < M; ++j)
    A[j] = gen([&](int i) { return (i < N) ? rec_elem_load(rec, stride, off + i + j * N) : 0.0; });
}
template <int N, int M>
SS_HD void store_colmajor(double* rec, long stride, int off, const rv* A) {
  SS_UNROLL_FULL
  for (int j = 0; j < M; ++j)
    each(A[j], [&](int i, double x) { if (i < N) rec_elem_store(rec, stride, off + i + j * N, x); });
}
template <int N>
SS_HD rv load_vec(const double* rec, long stride, int off) {
  return gen([&](int i) { return (i < N) ? rec_elem_load(rec, stride, off + i) : 0.0; });
}
template <int N>
SS_HD void store_vec(double* rec, long stride, int off, const rv& x) {
  each(x, [&](int i, double v) { if (i < N) rec_elem_store(rec, stride, off + i, v); });
}

// Diffuse part (sw_eig :180-221, lw_eig :142-176): from eigenvalues, X and (G1 - G2)^-1 X of the
// record to the eigenvector blocks G1, G2, lam, e = exp(-lam dz), LU(G1), C'_lower, C'_upper; R and T
// go to the layer record (column-major blocks at offR, offT).
template <int N>
struct Diffuse {
  rv G1[N], G2[N], Clo[N], Cup[N];
  LU<N> luG1;
  double lam[N], e[N];
  rv lam_d, e_d;
};

template <int N>
SS_HD void diffuse_part(const double* erec, long stride, double dz, Diffuse<N>& w, double* lrec, int offR, int offT) {
  using ER = EigRec<N>;
  w.lam_d = r_sqrt(r_max(load_vec<N>(erec, stride, ER::EV), 0.0));
  w.e_d = r_exp(-(w.lam_d * dz));
  all_of<N>(w.lam_d, w.lam);
  all_of<N>(w.e_d, w.e);
  rv t1[N];
  {
    // tmp = -((G1 - G2)^-1 X) diag(lam);  G1 = X + tmp, G2 = X - tmp
    rv X[N], AX[N];
    load_rowmajor<N, N>(erec, stride, ER::X, X);
    load_rowmajor<N, N>(erec, stride, ER::AX, AX);
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      const rv t = -(AX[j] * w.lam[j]);
      w.G1[j] = X[j] + t;
      w.G2[j] = X[j] - t;
      w.luG1.a[j] = w.G1[j];
      t1[j] = w.G2[j];
    }
  }
  // Eq. 48 by Schur complement; G1D = G1 diag(e), G2D = G2 diag(e) are never formed
  lu_factor<N>(w.luG1);
  lu_solve<N, N>(w.luG1, t1);                               // G1^-1 G2
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) t1[j] = t1[j] * (w.e_d * w.e[j]);   // diag(e) G1^-1 G2 diag(e)
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
  mul<N, N, false, true>(w.G2, w.e, w.Clo, w.Cup);          // G2D Clo
  lu_solve<N, N>(w.luG1, w.Cup);
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) w.Cup[j] = -w.Cup[j];         // C'_upper
  // Eq. 49: R = G1D Cup + G2 Clo ; T = G2 Cup + G1D Clo
  mul<N, N, false, false>(w.G2, nullptr, w.Clo, t1);
  mul<N, N, true, true>(w.G1, w.e, w.Cup, t1);
  store_colmajor<N, N>(lrec, stride, offR, t1);
  mul<N, N, false, false>(w.G2, nullptr, w.Cup, t1);
  mul<N, N, true, true>(w.G1, w.e, w.Clo, t1);
  store_colmajor<N, N>(lrec, stride, offT, t1);
}

// calc_matrices_lw_eig of a full problem: layer record of the unit from its eigen record
template <int N>
SS_HD void layer_lw(const double* erec, long stride, double* lrec) {
  using ER = EigRec<N>;
  using LR = LwRec<N>;
  const double dz = rec_elem_load(erec, stride, 1);
  // inv_gamma_b = -(G1 + G2)^-1 b (lw_eig :178-195 through the identity of ss_layer_matrices.hpp)
  rv igb;
  {
    LU<N> B;
    rv g2[N];
    load_rowmajor<N, N>(erec, stride, ER::G1, B.a);
    load_rowmajor<N, N>(erec, stride, ER::G2, g2);
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) B.a[j] = B.a[j] + g2[j];
    lu_factor<N>(B);
    rv v[1] = {-load_vec<N>(erec, stride, ER::VEC + N)};
    lu_solve<N, 1>(B, v);
    igb = v[0];
  }
  Diffuse<N> df;
  diffuse_part<N>(erec, stride, dz, df, lrec, LR::R, LR::T);
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

}  // namespace r16
}  // namespace ss
