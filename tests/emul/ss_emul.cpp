#include <cstdio>
#include <cstdlib>
// ss_emul.cpp - TEST-ONLY host build of the kernel bodies.
//
// The per-unit functions of spartacus_surface_amd/csrc (layer_unit_*,
// column_unit_*, flat/zero/spectrum units) are __host__ __device__; this file
// compiles them with g++ and drives them through the SAME host-side plan logic
// as the HIP library (ss_plan_impl.hpp) with plain loops instead of kernel
// launches.  Purpose: debug / sanitise the kernel arithmetic and the host logic
// in a container without a GPU.  It is NOT a fallback: nothing in the product
// path (libspartacus_surface_hip.so, spartacus_surface_amd.radsurf) links or
// loads it, and it lives under tests/.
#include <cstdlib>
#include <cstring>

#include "../../spartacus_surface_amd/csrc/ss_plan_impl.hpp"
#include "../../spartacus_surface_amd/csrc/ss_layer16.hpp"
#include "../../spartacus_surface_amd/csrc/ss_column16.hpp"
#include "../../spartacus_surface_amd/csrc/ss_down16.hpp"

namespace {

struct EmulBackend {
  struct Timers {
    void release() {}
  };
  static void init() {}
  static void release() {}
  static void begin_solve(Timers&) {}
  static void end_solve(Timers&) {}
  static void* alloc(size_t n) { return std::malloc(n ? n : 1); }
  static void free(void* p) { std::free(p); }
  static void memset(void* p, int v, size_t n) { std::memset(p, v, n); }
  static void memset_async(void* p, int v, size_t n) { std::memset(p, v, n); }
  static void h2d(void* d, const void* h, size_t n) { std::memcpy(d, h, n); }
  static void h2d_staged(void* d, const void* h, size_t n) { std::memcpy(d, h, n); }
  static void arena_reset() {}
  static void gate_copies() {}
  static void downloads_queued() {}
  static void d2h_batch_begin() {}
  static void d2h_batch_end() {}
  static void d2h(void* h, const void* d, size_t n) { std::memcpy(h, d, n); }
  static void sync() {}

  static void launch_flat(const ss::MiscLaunch& M, Timers&) {
    for (int i = 0; i < M.ncols; ++i) ss::flat_unit(M, i);
  }
  static void launch_night(const ss::MiscLaunch& M, Timers&) {
    for (int i = 0; i < M.ncols; ++i) {
      ss::zero_unit(M.in, M.sw_dir, M.nsw, M.cols[i]);
      ss::zero_unit(M.in, M.sw_diff, M.nsw, M.cols[i]);
    }
  }
  static void launch_simple_urban(const ss::MiscLaunch& M, Timers&) {
    const int nmax = M.nsw > M.nlw ? M.nsw : M.nlw;
    for (int u = 0; u < M.ncols * nmax; ++u) ss::simple_urban_unit(M, u);
  }
  static void launch_pack(const ss::PackLaunch& P, Timers&) {
    for (int cu = 0; cu < P.ncu; ++cu)
      for (int jlay = 0; jlay < P.maxlay; ++jlay) ss::pack_unit(P, cu, jlay);
  }
  static void launch_spectrum(const ss::SpectrumLaunch& S, Timers&) {
    for (int i = 0; i < S.ncols; ++i) ss::spectrum_col_unit(S, i);
    for (int i = 0; i < S.nlays; ++i) ss::spectrum_lay_unit(S, i);
  }
  template <bool U, int R, int NS>
  static void run(bool lw, const ss::Launch& L) {
    // orders of the fused kernel do the layer matrices inside column_unit_* (layer_up_*)
    if (!ss::fused_order(R * NS, lw)) {
      constexpr bool coop = ss::coop_order(R * NS);
      constexpr int NN = R * NS;
      if constexpr (coop) {
        // the two passes of the layer kernel around the cooperative eigen-solver, which a host
        // build runs with one "lane" per problem (same arithmetic, no lanes to share the rows with)
        for (int jlay = 0; jlay < L.maxlay; ++jlay)
          for (int cu = 0; cu < L.ncu; ++cu) {
            if (lw) ss::layer_unit_lw<U, R, NS, 1>(L, cu, jlay);
            else ss::layer_unit_sw<U, R, NS, 1>(L, cu, jlay);
            if (jlay < L.in.nlay[L.cols[cu / L.nspec]]) {
              double* erec = L.eig + ss::rec_base(jlay, L.maxlay, ss::EigRec<NN>::NELEM, cu);
              if (ss::r16_layer(NN, lw) && ss::rec_elem_load(erec, ss::kRecStride, ss::EigRec<NN>::STATUS) > 0.5)
                lw ? ss::r16::eig_pre_scalars<U, R, NS, true>(L.lg, erec, ss::kRecStride, L.eigen_fail ? L.eigen_fail + ss::kSlotOneLaneUnits : nullptr)
                   : ss::r16::eig_pre_scalars<U, R, NS, false>(L.lg, erec, ss::kRecStride, L.eigen_fail ? L.eigen_fail + ss::kSlotOneLaneUnits : nullptr);
              if (const char* dump = getenv("SS_EMUL_DUMP_W")) {   // prototyping aid: the Jacobi kernel's input matrices
                if (ss::rec_elem_load(erec, ss::kRecStride, ss::EigRec<NN>::STATUS) > 0.5) {
                  FILE* f = fopen(dump, "ab");
                  double hdr = NN;
                  fwrite(&hdr, 8, 1, f);
                  for (int e = 0; e < NN * NN; ++e) {
                    const double v = ss::rec_elem_load(erec, ss::kRecStride, ss::EigRec<NN>::W + e);
                    fwrite(&v, 8, 1, f);
                  }
                  fclose(f);
                }
              }
              ss::coop_eig_unit<NN, 1>(erec, ss::kRecStride, 0, L.eigen_fail ? L.eigen_fail + ss::kSlotOneLaneUnits : nullptr,
                                       L.cfg.decline_cooperative == 2);
              if (ss::r16_layer(NN, lw) && ss::rec_elem_load(erec, ss::kRecStride, ss::EigRec<NN>::STATUS) > 0.5) {
                if (lw) ss::r16::layer_lw<NN>(erec, ss::kRecStride, L.lay + ss::rec_base(jlay, L.maxlay, ss::LwRec<NN>::NELEM, cu),
                                              ss::r16::EigenvectorsFromScalars<U, R, NS, true>{&L.lg});
                else ss::r16::layer_sw<NN, R>(erec, ss::kRecStride, L.lay + ss::rec_base(jlay, L.maxlay, ss::SwRec<NN, R>::NELEM, cu),
                                              ss::r16::EigenvectorsFromScalars<U, R, NS, false>{&L.lg});
              }
            }
          }
      }
      for (int jlay = 0; jlay < L.maxlay; ++jlay)
        for (int cu = 0; cu < L.ncu; ++cu) {
          if (lw) ss::layer_unit_lw<U, R, NS, coop ? 2 : 0>(L, cu, jlay);
          else ss::layer_unit_sw<U, R, NS, coop ? 2 : 0>(L, cu, jlay);
        }
    }
    for (int cu = 0; cu < L.ncu; ++cu) {
      if constexpr (ss::r16_layer(R * NS, true) && ss::r16_layer(R * NS, false)) {
        // the cooperative orders: both sweeps on a pack of 16 "lanes"
        if (lw) {
          ss::r16::column_up_lw<U, R, NS>(L, cu);
          ss::r16::column_down_lw<U, R, NS>(L, cu);
        } else {
          ss::r16::column_up_sw<U, R, NS>(L, cu);
          ss::r16::column_down_sw<U, R, NS>(L, cu);
        }
      } else {
        if (lw) ss::column_unit_lw<U, R, NS>(L, cu);
        else ss::column_unit_sw<U, R, NS>(L, cu);
      }
    }
  }
  static void launch_scale_sum(const ss::ScaleSumLaunch& S, Timers&) {
    for (int kind = 0; kind < 4; ++kind) {
      long first;
      const long n = ss::scale_sum_pairs(S, kind, &first);
      for (long i = 0; i < n; ++i) ss::scale_sum_unit(S, kind, i);
    }
  }
  static void fork(int) {}
  static void join(int) {}
  static void launch_solver(const ss::ClassKey& key, bool lw, const ss::Launch& L, Timers&, int) {
#define EMUL_CALL(U, R, NS) run<U, R, NS>(lw, L)
    SS_DISPATCH(key, EMUL_CALL);
#undef EMUL_CALL
  }
};

}  // namespace

// solve + canopy_flux%scale/%sum: returns the scaled objects and the two sums
extern "C" int spartacus_scale_sum_emul(const ss_config* config, const ss_canopy* canopy,
                                        const ss_sw_spectral* sw, const ss_lw_spectral* lw,
                                        ss_bc_out* bc_out, int32_t istartcol, int32_t iendcol,
                                        ss_flux* sw_norm_dir, ss_flux* sw_norm_diff,
                                        ss_flux* lw_internal, ss_flux* lw_norm, int32_t want_profiles,
                                        const double* top_sw, const double* top_sw_dir, const double* top_lw,
                                        ss_flux* sw_flux, ss_flux* lw_flux) {
  ss::Plan<EmulBackend> plan;
  int rc = plan.create(config, canopy, sw, lw, istartcol, iendcol, want_profiles);
  if (rc) return rc;
  plan.upload_outputs(bc_out, sw_norm_dir, sw_norm_diff, lw_internal, lw_norm);
  rc = plan.solve();
  if (rc) return rc;
  rc = plan.scale_and_sum(top_sw, top_sw_dir, top_lw);
  if (rc) return rc;
  rc = plan.download(bc_out, sw_norm_dir, sw_norm_diff, lw_internal, lw_norm);
  if (rc) return rc;
  return plan.download_sums(sw_flux, lw_flux);
}

extern "C" int spartacus_radsurf_emul(const ss_config* config, const ss_canopy* canopy,
                                      const ss_sw_spectral* sw, const ss_lw_spectral* lw,
                                      ss_bc_out* bc_out, int32_t istartcol, int32_t iendcol,
                                      ss_flux* sw_norm_dir, ss_flux* sw_norm_diff,
                                      ss_flux* lw_internal, ss_flux* lw_norm,
                                      int32_t do_simple_spectrum, int32_t want_profiles,
                                      long long* eigen_failures) {
  ss::Plan<EmulBackend> plan;
  int rc = plan.create(config, canopy, sw, lw, istartcol, iendcol, want_profiles);
  if (rc) return rc;
  plan.upload_outputs(bc_out, sw_norm_dir, sw_norm_diff, lw_internal, lw_norm);
  if (do_simple_spectrum) {
    rc = plan.simple_spectrum_lw();
    if (rc) return rc;
    plan.download_spectrum(lw);
  }
  rc = plan.solve();
  if (rc) return rc;
  if (eigen_failures) *eigen_failures = plan.eigen_failures();
  return plan.download(bc_out, sw_norm_dir, sw_norm_diff, lw_internal, lw_norm);
}

// ---- unit-level entries of the symmetrised layer route (ss_layer_sym.hpp), TEST-ONLY ----
// The units write the pipeline's layer record (R^, T^ packed and scaled for the small orders); the entries hand
// back the layout of LwRec / SwRec (full, un-scaled), which is what the tests compare with the oracle.
namespace {
template <int N>
void unpack_rt(const double* in, int offR, int offT, const double* dsc, double* out, int outR, int outT) {
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < N; ++i) {
      const int e = ss::tri_index(N, i >= j ? i : j, i >= j ? j : i);
      out[outR + i + j * N] = dsc[i] * in[offR + e] / dsc[j];
      out[outT + i + j * N] = dsc[i] * in[offT + e] / dsc[j];
    }
}
template <int N>
int lw_sym_unit(const double* g1, const double* g2, const double* b, const double* dsc, double dz, double* rec) {
  using LR = ss::LwLayRec<N>;
  using FR = ss::LwRec<N>;
  double tmp[LR::NELEM + 2] = {0.0};
  const ss::ArrayRecWriter w{tmp, nullptr};
  const ss::LaneLds<(N <= SS_UNROLL_MAX)> lds{};   // orders above 16: rolled loops, nothing staged (ss_solver_lw.hpp)
  const int rc = ss::sym::calc_matrices_lw_sym<N, N, 0>(g1, g2, b, dsc, dz, w, lds);
  if (LR::PACKED) unpack_rt<N>(tmp, LR::R, LR::T, dsc, rec, FR::R, FR::T);
  else for (int k = 0; k < FR::IFLUX; ++k) rec[k] = tmp[k];
  for (int k = 0; k < FR::NELEM - FR::IFLUX; ++k) rec[FR::IFLUX + k] = tmp[LR::IFLUX + k];
  return rc;
}
template <int N, int D>
int sw_sym_unit(const double* g0, const double* g1, const double* g2, const double* g3, const double* dsc0,
                const double* dsc, double dz, double* rec) {
  using LR = ss::SwLayRec<N, D>;
  using FR = ss::SwRec<N, D>;
  double tmp[LR::NELEM + 2] = {0.0};
  const ss::ArrayRecWriter w{tmp, nullptr};
  const ss::LaneLds<(N <= SS_UNROLL_MAX)> lds{};
  const int rc = ss::sym::calc_matrices_sw_sym<N, D, N, D, 0, 0>(g0, g1, g2, g3, dsc0, dsc, dz, w, lds);
  if (LR::PACKED) unpack_rt<N>(tmp, LR::R, LR::T, dsc, rec, FR::R, FR::T);
  else for (int k = 0; k < FR::SUP; ++k) rec[k] = tmp[k];
  for (int k = 0; k < FR::NELEM - FR::SUP; ++k) rec[FR::SUP + k] = tmp[LR::SUP + k];
  if (LR::PACKED)
    for (int j = 0; j < D; ++j)
      for (int i = 0; i < N; ++i) { rec[FR::SUP + i + j * N] *= dsc[i]; rec[FR::SDN + i + j * N] *= dsc[i]; }
  return rc;
}
}  // namespace

// matrices column-major (n x n); rec receives the LwRec<n> layout (R, T, IFLUX, SRC, ISRC)
extern "C" int emul_lw_sym_unit(int n, int nstr, const double* g1, const double* g2, const double* b, const double* dsc,
                                double dz, double* rec) {
  (void)nstr;
  switch (n) {
    case 1: return lw_sym_unit<1>(g1, g2, b, dsc, dz, rec);
    case 2: return lw_sym_unit<2>(g1, g2, b, dsc, dz, rec);
    case 3: return lw_sym_unit<3>(g1, g2, b, dsc, dz, rec);
    case 4: return lw_sym_unit<4>(g1, g2, b, dsc, dz, rec);
    case 5: return lw_sym_unit<5>(g1, g2, b, dsc, dz, rec);
    case 6: return lw_sym_unit<6>(g1, g2, b, dsc, dz, rec);
    case 24: return lw_sym_unit<24>(g1, g2, b, dsc, dz, rec);
    default: return -99;
  }
}
extern "C" int emul_lw_rec_offsets(int n, int* off) {
  switch (n) {
#define OFFS(N) case N: off[0] = ss::LwRec<N>::R; off[1] = ss::LwRec<N>::T; off[2] = ss::LwRec<N>::IFLUX; off[3] = ss::LwRec<N>::SRC; off[4] = ss::LwRec<N>::ISRC; off[5] = ss::LwRec<N>::NELEM; return 0;
    OFFS(1) OFFS(2) OFFS(3) OFFS(4) OFFS(5) OFFS(6) OFFS(24)
#undef OFFS
    default: return -99;
  }
}
// rec receives the SwRec<n, d> layout (R, T, SUP, SDN, EDIR, IDIR)
extern "C" int emul_sw_sym_unit(int n, int d, const double* g0, const double* g1, const double* g2, const double* g3,
                                const double* dsc0, const double* dsc, double dz, double* rec) {
  if (n == 6 && d == 3) return sw_sym_unit<6, 3>(g0, g1, g2, g3, dsc0, dsc, dz, rec);
  if (n == 4 && d == 2) return sw_sym_unit<4, 2>(g0, g1, g2, g3, dsc0, dsc, dz, rec);
  if (n == 6 && d == 2) return sw_sym_unit<6, 2>(g0, g1, g2, g3, dsc0, dsc, dz, rec);
  if (n == 3 && d == 3) return sw_sym_unit<3, 3>(g0, g1, g2, g3, dsc0, dsc, dz, rec);
  if (n == 2 && d == 1) return sw_sym_unit<2, 1>(g0, g1, g2, g3, dsc0, dsc, dz, rec);
  if (n == 2 && d == 2) return sw_sym_unit<2, 2>(g0, g1, g2, g3, dsc0, dsc, dz, rec);
  if (n == 24 && d == 3) return sw_sym_unit<24, 3>(g0, g1, g2, g3, dsc0, dsc, dz, rec);
  return -99;
}
extern "C" int emul_sw_rec_offsets(int n, int d, int* off) {
#define OFFS(N, D) if (n == N && d == D) { using R_ = ss::SwRec<N, D>; off[0] = R_::R; off[1] = R_::T; off[2] = R_::SUP; off[3] = R_::SDN; off[4] = R_::EDIR; off[5] = R_::IDIR; off[6] = R_::NELEM; return 0; }
  OFFS(6, 3) OFFS(4, 2) OFFS(6, 2) OFFS(3, 3) OFFS(2, 1) OFFS(2, 2) OFFS(24, 3)
#undef OFFS
  return -99;
}
