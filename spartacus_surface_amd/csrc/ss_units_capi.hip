// ss_units_capi.hip - kernel-level entry points of the C-ABI: the layer-matrix
// units of the hot path on their own, with the argument lists of the reference's
//   calc_matrices_sw_eig   radtool/radtool_calc_matrices_sw_eig.F90:30-298
//   calc_matrices_lw_eig   radtool/radtool_calc_matrices_lw_eig.F90:32-230
//   eigen_decomposition_real radtool/radtool_eigen_decomposition.F90:51-828
// (stacks of nmat independent problems, Fortran layout x(nmat, i, j)).  They launch
// the SAME __device__ bodies the solver kernels inline (ss_layer_matrices.hpp,
// ss_eigen*.hpp: one lane per problem; orders 7..16 with a scaling vector: the row-of-16
// kernels of ss_layer16.hpp around the cooperative Jacobi kernel, as the solver runs them),
// results through the scratch records the layer kernel writes.  A host model that keeps radtool's interface can call
// them; the test-suite uses them for known answers on the device (the reference's
// driver/test_sw.F90 and seeded Gamma sets for every matrix order).
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/spartacus_surface_hip.h"
#include "ss_common.hpp"
#include "ss_eigen.hpp"
#include "ss_eigen_sym.hpp"
#include "ss_eigen_coop.hpp"
#include "ss_layer_matrices.hpp"
#include "ss_linalg.hpp"
#include "ss_layer16.hpp"

namespace {

struct UnitArgs {
  int nmat;
  double dz;
  const double *g0, *g1, *g2, *g3, *b, *dsc0, *dsc;
  double* rec;          // scratch records [tile][1 layer][elem pair][lane][2]
  double *o0, *o1, *o2, *o3, *o4, *o5, *o6, *o7;
  int32_t* route;
  int force_dynamic;
  double* eig;          // eigen records of the cooperative eigen-solver (orders 7..16 with dscale), else nullptr
};

// element (i, j) of problem m in a stack x(nmat, R, C)
__device__ inline long sidx(int nmat, int m, int i, int j, int R) { return (long)m + (long)nmat * (i + (long)R * j); }

template <int R, int C>
__device__ inline void load_stack(const double* x, int nmat, int m, double* a) {
  for (int j = 0; j < C; ++j)
    for (int i = 0; i < R; ++i) a[i + j * R] = x[sidx(nmat, m, i, j, R)];
}
template <int R, int C>
__device__ inline void store_stack(double* x, int nmat, int m, const double* a) {
  if (!x) return;
  for (int j = 0; j < C; ++j)
    for (int i = 0; i < R; ++i) x[sidx(nmat, m, i, j, R)] = a[i + j * R];
}
// block of NEL elements starting at the even element "off" of this lane's record -> stack
template <int R, int C>
__device__ inline void rec_to_stack(const double* rec, int off, double* x, int nmat, int m) {
  if (!x) return;
  double a[R * C];
  ss::rec_load<R * C>(rec, ss::kRecStride, off, a);
  store_stack<R, C>(x, nmat, m, a);
}

template <int N, int D>
__global__ void __launch_bounds__(64) k_unit_sw(const UnitArgs A) {
  const int m = blockIdx.x * 64 + threadIdx.x;
  if (m >= A.nmat) return;
  using Rec = ss::SwRec<N, D>;
  double g0[D * D], g1[N * N], g2[N * N], g3[N * D], dsc0[D], dsc[N];
  load_stack<D, D>(A.g0, A.nmat, m, g0);
  load_stack<N, N>(A.g1, A.nmat, m, g1);
  load_stack<N, N>(A.g2, A.nmat, m, g2);
  load_stack<N, D>(A.g3, A.nmat, m, g3);
  if (A.dsc0) load_stack<D, 1>(A.dsc0, A.nmat, m, dsc0);
  if (A.dsc) load_stack<N, 1>(A.dsc, A.nmat, m, dsc);
  double* rec = A.rec + ss::rec_base(0, 1, Rec::NELEM, m);
  ss::RecWriter w{rec, ss::kRecStride, nullptr};
  // which eigen route the unit takes (the decision of diffuse_part, repeated for the record)
  if (A.route) {
    int r = 0;
    if constexpr (N >= 3 && N <= SS_UNROLL_MAX) {
      if (A.dsc) {
        double GD[N * N], ev[N], X[N * N], P[N * N];
        for (int k = 0; k < N * N; ++k) GD[k] = g1[k] - g2[k];
        const int ierr = ss::eigen_decomposition_gamma_sym<N>(GD, g2, dsc, ev, X, P);
        r = (ierr >= 0 && ss::sym_spectrum_is_well_graded<N>(ev)) ? 1 : 0;
      }
    }
    A.route[m] = r;
  }
  const ss::CoopEig ce{A.eig ? A.eig + ss::rec_base(0, 1, ss::EigRec<N>::NELEM, m) : nullptr, ss::kRecStride};
  if (A.route && ss::coop_order(N) && A.eig)
    A.route[m] = ss::rec_elem_load(ce.rec, ce.stride, ss::EigRec<N>::STATUS) > 0.5 ? 2 : 0;   // 2 = cooperative Jacobi
  bool done16 = false;
  if constexpr (ss::r16_layer(N, false))
    done16 = A.eig && ss::rec_elem_load(ce.rec, ce.stride, ss::EigRec<N>::STATUS) > 0.5;   // k_unit_layer16 wrote the record
  if (!done16)
    ss::calc_matrices_sw<N, D, N, D, 0, 0>(g0, g1, g2, g3, A.dsc0 ? dsc0 : nullptr, A.dsc ? dsc : nullptr, A.dz, w,
                                           A.eig ? &ce : nullptr);
  __threadfence();
  rec_to_stack<N, N>(rec, Rec::R, A.o0, A.nmat, m);
  rec_to_stack<N, N>(rec, Rec::T, A.o1, A.nmat, m);
  rec_to_stack<N, D>(rec, Rec::SUP, A.o2, A.nmat, m);
  rec_to_stack<N, D>(rec, Rec::SDN, A.o3, A.nmat, m);
  rec_to_stack<D, D>(rec, Rec::EDIR, A.o4, A.nmat, m);
  rec_to_stack<D, D>(rec, Rec::IDIR, A.o5, A.nmat, m);
  // int_diff and int_dir_diff are not records of the layer kernel: the downward sweep applies
  // int_diff = -(G1 + G2)^-1 and int_dir_diff = -2 int_diff G3 G0^-1 as one solve
  // (sw_int_flux_system, ss_solver_sw.hpp); the same identities, as matrices, here
  if (A.o6 || A.o7) {
    double B[N * N], Bi[N * N];
    for (int k = 0; k < N * N; ++k) B[k] = g1[k] + g2[k];
    ss::lu_factor<N>(B);
    ss::lu_invert<N>(B, Bi);
    for (int k = 0; k < N * N; ++k) Bi[k] = -Bi[k];
    store_stack<N, N>(A.o6, A.nmat, m, Bi);
    if (A.o7) {
      double L0[D * D], g0i[D * D], t[N * D], idd[N * D];
      ss::mat_copy<D, D>(g0, L0);
      ss::lu_factor<D>(L0);
      ss::lu_invert<D>(L0, g0i);
      ss::mat_mul<N, D, D>(g3, g0i, t);
      ss::mat_mul<N, N, D>(Bi, t, idd);
      for (int k = 0; k < N * D; ++k) idd[k] = -2.0 * idd[k];
      store_stack<N, D>(A.o7, A.nmat, m, idd);
    }
  }
}

template <int N>
__global__ void __launch_bounds__(64) k_unit_lw(const UnitArgs A) {
  const int m = blockIdx.x * 64 + threadIdx.x;
  if (m >= A.nmat) return;
  using Rec = ss::LwRec<N>;
  double g1[N * N], g2[N * N], b[N], dsc[N];
  load_stack<N, N>(A.g1, A.nmat, m, g1);
  load_stack<N, N>(A.g2, A.nmat, m, g2);
  load_stack<N, 1>(A.b, A.nmat, m, b);
  if (A.dsc) load_stack<N, 1>(A.dsc, A.nmat, m, dsc);
  double* rec = A.rec + ss::rec_base(0, 1, Rec::NELEM, m);
  ss::RecWriter w{rec, ss::kRecStride, nullptr};
  const ss::CoopEig ce{A.eig ? A.eig + ss::rec_base(0, 1, ss::EigRec<N>::NELEM, m) : nullptr, ss::kRecStride};
  bool done16 = false;
  if constexpr (ss::r16_layer(N, true))
    done16 = A.eig && ss::rec_elem_load(ce.rec, ce.stride, ss::EigRec<N>::STATUS) > 0.5;   // k_unit_layer16 wrote the record
  if (!done16) ss::calc_matrices_lw<N, N, 0>(g1, g2, b, A.dsc ? dsc : nullptr, A.dz, w, A.eig ? &ce : nullptr);
  __threadfence();
  rec_to_stack<N, N>(rec, Rec::R, A.o0, A.nmat, m);
  rec_to_stack<N, N>(rec, Rec::T, A.o1, A.nmat, m);
  rec_to_stack<N, 1>(rec, Rec::SRC, A.o2, A.nmat, m);
  rec_to_stack<N, N>(rec, Rec::IFLUX, A.o3, A.nmat, m);
  rec_to_stack<N, 1>(rec, Rec::ISRC, A.o4, A.nmat, m);
}

// The launches of the cooperative route (orders 7..16), as launch_ns (ss_inst_body.hpp) runs them:
// first pass (one lane per problem: structure check, the Gamma matrices and - shortwave - the
// direct-beam part into the records), Cholesky factors on DPP rows, Jacobi, layer matrices on DPP rows
template <int N, int D, bool LW>
__global__ void __launch_bounds__(64) k_unit_coop_first(const UnitArgs A) {
  const int m = blockIdx.x * 64 + threadIdx.x;
  if (m >= A.nmat) return;
  using ER = ss::EigRec<N>;
  double g1[N * N], g2[N * N], dsc[N];
  load_stack<N, N>(A.g1, A.nmat, m, g1);
  load_stack<N, N>(A.g2, A.nmat, m, g2);
  load_stack<N, 1>(A.dsc, A.nmat, m, dsc);
  double* erec = A.eig + ss::rec_base(0, 1, ER::NELEM, m);
  constexpr long st = ss::kRecStride;
  double size;
  bool ok = ss::coop_eig_check<N>(g1, g2, dsc, size);
  if (!LW && !A.dsc0) ok = false;
  ss::rec_elem_store(erec, st, ER::STATUS, ok ? 1.0 : 0.0);
  if (!ok) return;
  ss::rec_elem_store(erec, st, 1, A.dz);
  ss::rec_elem_store(erec, st, ER::EV, size);
  for (int i = 0; i < N; ++i) {
    for (int j = 0; j < N; ++j) {
      ss::rec_elem_store(erec, st, ER::G1 + i * N + j, g1[i + j * N]);
      ss::rec_elem_store(erec, st, ER::G2 + i * N + j, g2[i + j * N]);
    }
    ss::rec_elem_store(erec, st, ER::VEC + i, dsc[i]);
  }
  if constexpr (LW) {
    double b[N];
    load_stack<N, 1>(A.b, A.nmat, m, b);
    for (int i = 0; i < N; ++i) ss::rec_elem_store(erec, st, ER::VEC + N + i, b[i]);
  } else {
    double g0[D * D], g3[N * D], dsc0[D], mu[D], G0[D * D], G0inv[D * D], emu[D];
    load_stack<D, D>(A.g0, A.nmat, m, g0);
    load_stack<N, D>(A.g3, A.nmat, m, g3);
    load_stack<D, 1>(A.dsc0, A.nmat, m, dsc0);
    const ss::RecWriter w{A.rec + ss::rec_base(0, 1, ss::SwRec<N, D>::NELEM, m), st, nullptr};
    ss::sw_direct_part<N, D>(g0, dsc0, A.dz, w, mu, G0, G0inv, emu);
    for (int i = 0; i < N; ++i)
      for (int r = 0; r < D; ++r) ss::rec_elem_store(erec, st, ER::vg3() + i * D + r, g3[i + r * N]);
    for (int r = 0; r < D; ++r) {
      ss::rec_elem_store(erec, st, ER::vmu(D) + r, mu[r]);
      ss::rec_elem_store(erec, st, ER::vemu(D) + r, emu[r]);
    }
    for (int k = 0; k < D * D; ++k) {
      ss::rec_elem_store(erec, st, ER::vg0(D) + k, G0[k]);
      ss::rec_elem_store(erec, st, ER::vg0inv(D) + k, G0inv[k]);
    }
  }
}
// STEP 0: Cholesky factors (r16::eig_pre), STEP 1: layer matrices (r16::layer_lw / layer_sw); 4 problems per wave
template <int N, int D, bool LW, int STEP>
__global__ void __launch_bounds__(64, 2) k_unit_row16(const UnitArgs A) {
  const int m = blockIdx.x * (64 / ss::r16::kRow) + (int)threadIdx.x / ss::r16::kRow;
  if (m >= A.nmat) return;
  if constexpr (ss::coop_order(N)) {
    double* erec = A.eig + ss::rec_base(0, 1, ss::EigRec<N>::NELEM, m);
    if (!(ss::rec_elem_load(erec, ss::kRecStride, ss::EigRec<N>::STATUS) > 0.5)) return;
    if constexpr (STEP == 0) ss::r16::eig_pre<N, LW>(erec, ss::kRecStride, nullptr);
    else if constexpr (LW) ss::r16::layer_lw<N>(erec, ss::kRecStride, A.rec + ss::rec_base(0, 1, ss::LwRec<N>::NELEM, m));
    else ss::r16::layer_sw<N, D>(erec, ss::kRecStride, A.rec + ss::rec_base(0, 1, ss::SwRec<N, D>::NELEM, m));
  }
}
template <int N>
__global__ void __launch_bounds__(64) k_unit_coop_eig(const UnitArgs A) {
  constexpr int G = ss::coop_lanes(N);
  const int m = blockIdx.x * (64 / G) + (int)threadIdx.x / G;
  if (m >= A.nmat) return;
  if constexpr (ss::coop_order(N))
    ss::coop_eig_unit<N, G>(A.eig + ss::rec_base(0, 1, ss::EigRec<N>::NELEM, m), ss::kRecStride, (int)threadIdx.x % G);
}

template <int N>
__global__ void __launch_bounds__(64) k_unit_eigen(const UnitArgs A) {
  const int m = blockIdx.x * 64 + threadIdx.x;
  if (m >= A.nmat) return;
  double a[N * N], ev[N], X[N * N];
  load_stack<N, N>(A.g1, A.nmat, m, a);
  int ierr;
  if (A.force_dynamic) ierr = ss::eigen_decomposition_real_dynamic<N>(a, ev, X);
  else ierr = ss::eigen_decomposition_real<N>(a, ev, X);
  store_stack<N, 1>(A.o0, A.nmat, m, ev);
  store_stack<N, N>(A.o1, A.nmat, m, X);
  if (A.route) A.route[m] = ierr;
}

thread_local std::string g_units_error;

// device staging of one call: inputs up, outputs down, everything freed on scope exit
struct Stage {
  std::vector<void*> bufs;
  hipError_t err = hipSuccess;
  ~Stage() { for (void* p : bufs) hipFree(p); }
  void* raw(size_t bytes) {
    void* p = nullptr;
    if (err == hipSuccess) err = hipMalloc(&p, bytes ? bytes : 8);
    if (p) bufs.push_back(p);
    return p;
  }
  const double* up(const double* h, size_t n) {
    if (!h) return nullptr;
    double* d = static_cast<double*>(raw(n * sizeof(double)));
    if (d && err == hipSuccess) err = hipMemcpy(d, h, n * sizeof(double), hipMemcpyHostToDevice);
    return d;
  }
  double* out(double* h, size_t n) { return h ? static_cast<double*>(raw(n * sizeof(double))) : nullptr; }
  void down(double* h, const double* d, size_t n) {
    if (h && d && err == hipSuccess) err = hipMemcpy(h, d, n * sizeof(double), hipMemcpyDeviceToHost);
  }
};

int finish(Stage& s, const char* what) {
  if (s.err == hipSuccess) s.err = hipGetLastError();
  if (s.err == hipSuccess) s.err = hipDeviceSynchronize();
  if (s.err != hipSuccess) {
    g_units_error = std::string(what) + ": " + hipGetErrorString(s.err);
    return SS_ERR_HIP;
  }
  return SS_OK;
}

#define UNIT_COOP(NN, DD, LWF)                                                                       \
  if (ss::coop_order(NN) && A.dsc) {                                                                \
    const unsigned g16 = (unsigned)((nmat + 3) / 4);                                                \
    A.eig = static_cast<double*>(s.raw(tiles * 64 * (size_t)ss::EigRec<NN>::NELEM * sizeof(double)));   \
    if (s.err == hipSuccess) hipLaunchKernelGGL((k_unit_coop_first<NN, DD, LWF>), dim3(tiles), dim3(64), 0, 0, A); \
    if (s.err == hipSuccess) hipLaunchKernelGGL((k_unit_row16<NN, DD, LWF, 0>), dim3(g16), dim3(64), 0, 0, A);      \
    if (s.err == hipSuccess) hipLaunchKernelGGL((k_unit_coop_eig<NN>), dim3((nmat + 64 / ss::coop_lanes(NN) - 1) / (64 / ss::coop_lanes(NN))), dim3(64), 0, 0, A); \
    if (s.err == hipSuccess) hipLaunchKernelGGL((k_unit_row16<NN, DD, LWF, 1>), dim3(g16), dim3(64), 0, 0, A);      \
  }
#define UNIT_SW_CASE(NN, DD)                                                                        \
  if (ndiff == NN && ndir == DD) {                                                                  \
    A.rec = static_cast<double*>(s.raw(tiles * 64 * (size_t)ss::SwRec<NN, DD>::NELEM * sizeof(double))); \
    UNIT_COOP(NN, DD, false)                                                                        \
    if (s.err == hipSuccess) hipLaunchKernelGGL((k_unit_sw<NN, DD>), dim3(tiles), dim3(64), 0, 0, A);   \
    launched = true;                                                                                \
  }
#define UNIT_LW_CASE(NN)                                                                            \
  if (norder == NN) {                                                                               \
    A.rec = static_cast<double*>(s.raw(tiles * 64 * (size_t)ss::LwRec<NN>::NELEM * sizeof(double)));   \
    UNIT_COOP(NN, 1, true)                                                                          \
    if (s.err == hipSuccess) hipLaunchKernelGGL((k_unit_lw<NN>), dim3(tiles), dim3(64), 0, 0, A);       \
    launched = true;                                                                                \
  }
#define UNIT_EIG_CASE(NN)                                                                           \
  if (norder == NN) {                                                                               \
    if (s.err == hipSuccess) hipLaunchKernelGGL((k_unit_eigen<NN>), dim3(tiles), dim3(64), 0, 0, A);    \
    launched = true;                                                                                \
  }

}  // namespace

extern "C" {

const char* ss_units_last_error(void) { return g_units_error.c_str(); }

int ss_calc_matrices_sw_eig(int32_t nmat, int32_t ndiff, int32_t ndir, double dz, double mu0,
                            const double* gamma0, const double* gamma1, const double* gamma2,
                            const double* gamma3, const double* dscale0, const double* dscale,
                            double* reflectance, double* transmittance, double* s_up, double* s_dn,
                            double* trans_dir, double* int_dir, double* int_diff, double* int_dir_diff,
                            int32_t* sym_route) {
  (void)mu0;   // radtool_calc_matrices_sw_eig.F90:377,384: the mu0 scaling of s_up / s_dn is commented out
  if (nmat <= 0 || !gamma0 || !gamma1 || !gamma2 || !gamma3) { g_units_error = "bad argument"; return SS_ERR_BAD_ARGUMENT; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_units_error = "no HIP device"; return SS_ERR_NO_DEVICE; }
  Stage s;
  const size_t nm = (size_t)nmat, n2 = nm * ndiff * ndiff, nd = nm * ndiff * ndir, d2 = nm * ndir * ndir;
  const unsigned tiles = (unsigned)((nmat + 63) / 64);
  UnitArgs A{};
  A.nmat = nmat; A.dz = dz;
  A.g0 = s.up(gamma0, d2); A.g1 = s.up(gamma1, n2); A.g2 = s.up(gamma2, n2); A.g3 = s.up(gamma3, nd);
  A.dsc0 = s.up(dscale0, nm * ndir); A.dsc = s.up(dscale, nm * ndiff);
  A.o0 = s.out(reflectance, n2); A.o1 = s.out(transmittance, n2); A.o2 = s.out(s_up, nd); A.o3 = s.out(s_dn, nd);
  A.o4 = s.out(trans_dir, d2); A.o5 = s.out(int_dir, d2); A.o6 = s.out(int_diff, n2); A.o7 = s.out(int_dir_diff, nd);
  A.route = sym_route ? static_cast<int32_t*>(s.raw(nm * sizeof(int32_t))) : nullptr;
  bool launched = false;
  UNIT_SW_CASE(1, 1) UNIT_SW_CASE(2, 1) UNIT_SW_CASE(2, 2) UNIT_SW_CASE(3, 3) UNIT_SW_CASE(4, 2)
  UNIT_SW_CASE(6, 3) UNIT_SW_CASE(8, 2) UNIT_SW_CASE(12, 3)
  if (!launched) { g_units_error = "(ndiff, ndir) not instantiated: (1,1) (2,1) (2,2) (3,3) (4,2) (6,3) (8,2) (12,3)"; return SS_ERR_UNSUPPORTED; }
  int rc = finish(s, "ss_calc_matrices_sw_eig");
  if (rc) return rc;
  s.down(reflectance, A.o0, n2); s.down(transmittance, A.o1, n2); s.down(s_up, A.o2, nd); s.down(s_dn, A.o3, nd);
  s.down(trans_dir, A.o4, d2); s.down(int_dir, A.o5, d2); s.down(int_diff, A.o6, n2); s.down(int_dir_diff, A.o7, nd);
  if (sym_route && s.err == hipSuccess) s.err = hipMemcpy(sym_route, A.route, nm * sizeof(int32_t), hipMemcpyDeviceToHost);
  return finish(s, "ss_calc_matrices_sw_eig (download)");
}

int ss_calc_matrices_lw_eig(int32_t nmat, int32_t norder, double dz, const double* gamma1,
                            const double* gamma2, const double* emiss_rate, const double* dscale,
                            double* reflectance, double* transmittance, double* source, double* int_flux,
                            double* int_flux_source) {
  if (nmat <= 0 || !gamma1 || !gamma2 || !emiss_rate) { g_units_error = "bad argument"; return SS_ERR_BAD_ARGUMENT; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_units_error = "no HIP device"; return SS_ERR_NO_DEVICE; }
  Stage s;
  const size_t nm = (size_t)nmat, n2 = nm * norder * norder, n1 = nm * norder;
  const unsigned tiles = (unsigned)((nmat + 63) / 64);
  UnitArgs A{};
  A.nmat = nmat; A.dz = dz;
  A.g1 = s.up(gamma1, n2); A.g2 = s.up(gamma2, n2); A.b = s.up(emiss_rate, n1); A.dsc = s.up(dscale, n1);
  A.o0 = s.out(reflectance, n2); A.o1 = s.out(transmittance, n2); A.o2 = s.out(source, n1);
  A.o3 = s.out(int_flux, n2); A.o4 = s.out(int_flux_source, n1);
  bool launched = false;
  UNIT_LW_CASE(1) UNIT_LW_CASE(2) UNIT_LW_CASE(3) UNIT_LW_CASE(4) UNIT_LW_CASE(6) UNIT_LW_CASE(8) UNIT_LW_CASE(12)
  if (!launched) { g_units_error = "order not instantiated: 1 2 3 4 6 8 12"; return SS_ERR_UNSUPPORTED; }
  int rc = finish(s, "ss_calc_matrices_lw_eig");
  if (rc) return rc;
  s.down(reflectance, A.o0, n2); s.down(transmittance, A.o1, n2); s.down(source, A.o2, n1);
  s.down(int_flux, A.o3, n2); s.down(int_flux_source, A.o4, n1);
  return finish(s, "ss_calc_matrices_lw_eig (download)");
}

int ss_eigen_decomposition_real(int32_t norder, int32_t nmat, const double* amat, double* eigenvalue,
                                double* eigenvector, int32_t force_general_loops, int32_t* ierr) {
  if (nmat <= 0 || !amat || !eigenvalue || !eigenvector) { g_units_error = "bad argument"; return SS_ERR_BAD_ARGUMENT; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_units_error = "no HIP device"; return SS_ERR_NO_DEVICE; }
  Stage s;
  const size_t nm = (size_t)nmat, n2 = nm * norder * norder, n1 = nm * norder;
  const unsigned tiles = (unsigned)((nmat + 63) / 64);
  UnitArgs A{};
  A.nmat = nmat;
  A.force_dynamic = force_general_loops;
  A.g1 = s.up(amat, n2);
  A.o0 = s.out(eigenvalue, n1); A.o1 = s.out(eigenvector, n2);
  A.route = ierr ? static_cast<int32_t*>(s.raw(nm * sizeof(int32_t))) : nullptr;
  bool launched = false;
  UNIT_EIG_CASE(1) UNIT_EIG_CASE(2) UNIT_EIG_CASE(3) UNIT_EIG_CASE(4) UNIT_EIG_CASE(5) UNIT_EIG_CASE(6)
  UNIT_EIG_CASE(8) UNIT_EIG_CASE(12)
  if (!launched) { g_units_error = "order not instantiated: 1 2 3 4 5 6 8 12"; return SS_ERR_UNSUPPORTED; }
  int rc = finish(s, "ss_eigen_decomposition_real");
  if (rc) return rc;
  s.down(eigenvalue, A.o0, n1); s.down(eigenvector, A.o1, n2);
  if (ierr && s.err == hipSuccess) s.err = hipMemcpy(ierr, A.route, nm * sizeof(int32_t), hipMemcpyDeviceToHost);
  return finish(s, "ss_eigen_decomposition_real (download)");
}

}  // extern "C"
