// ss_solver_sw.hpp - shortwave solver units.
//
//   layer_unit_sw  : one (column, band, layer): region optics, exchange rates,
//                    Gamma matrices and the layer matrices R, T, S_up, S_dn, E,
//                    int_* (sections 3a-3c of spartacus_forest_sw,
//                    radsurf_forest_sw.F90:270-457, and of spartacus_urban_sw,
//                    radsurf_urban_sw.F90:335-585) -> HBM layer record.
//   column_unit_sw : one (column, band): adding method upward (section 4,
//                    forest :463-528 / urban :591-674) and the two downward
//                    flux passes (section 5, forest :530-760 / urban :676-984)
//                    -> bc_out and canopy_flux arrays.
//
// Template parameters: URBAN (walls + exposed-roof pseudo region), NREG
// (regions: 1 clear + NREG-1 vegetated), NS (streams per hemisphere).
#pragma once

#include "ss_common.hpp"
#include "ss_linalg.hpp"
#include "ss_geometry.hpp"
#include "ss_layer_matrices.hpp"
#include "ss_layer_sym.hpp"
#include "ss_eigen_coop.hpp"
#include "ss_pack.hpp"
#include "ss_layer16.hpp"

namespace ss {

// Copy the (RS x CS) sub-block at (I0, J0) of an (RF x CF) matrix
template <int RF, int RS, int CS, int I0, int J0>
SS_HD void get_block(const double* A, double* B) {
  SS_UNROLL_FULL
  for (int j = 0; j < CS; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < RS; ++i) SS_E(B, i, j, RS) = SS_E(A, I0 + i, J0 + j, RF);
  }
}

// Region extinction and single-scattering albedo (radsurf_forest_sw.F90:277-297)
template <int NREG>
SS_HD void region_optics(double air_ext, double air_ssa, double veg_ext, double veg_ssa,
                         const double* od, double* ext_reg, double* ssa_reg) {
  ext_reg[0] = air_ext;
  ssa_reg[0] = air_ssa;
  if (NREG == 2) {
    ext_reg[1] = air_ext + veg_ext;
    ssa_reg[1] = (ext_reg[0] * ssa_reg[0] + veg_ext * veg_ssa) / ss_max(ext_reg[1], 1.0e-8);
  }
  if (NREG == 3) {
    SS_UNROLL_FULL
    for (int r = 1; r < NREG; ++r) {
      ext_reg[r] = air_ext + od[r] * veg_ext;
      ssa_reg[r] = (ext_reg[0] * ssa_reg[0] + od[r] * veg_ext * veg_ssa) / ss_max(ext_reg[r], 1.0e-8);
    }
  }
}

// Everything of a layer the shortwave Gamma matrices are built from
template <bool URBAN, int NREG>
struct SwLayerProps {
  LayerGeom g;
  double cos_sza, sin0, tan0;
  double frac[NREG], f_wall[NREG], ext_reg[NREG], ssa_reg[NREG], fx[NREG * NREG];
  double wall_ext, wall_factor;
};
template <bool URBAN, int NREG>
SS_HD void sw_layer_props(const Launch& L, double cos_in, const double* pk, SwLayerProps<URBAN, NREG>& p) {
  p.cos_sza = URBAN ? ss_max(cos_in, 1.0e-6) : cos_in;
  p.sin0 = 0.0;
  if (URBAN) {
    p.sin0 = sqrt(1.0 - p.cos_sza * p.cos_sza);
    p.tan0 = p.sin0 / p.cos_sza;
  } else {
    p.tan0 = sqrt(1.0 - p.cos_sza * p.cos_sza) / ss_max(p.cos_sza, 1.0e-6);
  }
  LayerGeom& g = p.g;
  g.dz = pk[0]; g.veg_fraction = pk[1]; g.veg_scale = pk[2]; g.veg_ext = pk[3];
  g.veg_fsd = pk[4]; g.building_fraction = pk[5]; g.building_scale = pk[6]; g.veg_contact_fraction = pk[7];
  double np[NREG], npw[NREG], od[NREG];
  region_fractions<URBAN, false, NREG>(g.veg_fraction, g.building_fraction, p.frac);
  norm_perimeters<URBAN, NREG>(L.cfg, g, np, npw);
  exchange_rates<NREG>(p.frac, np, L.cfg.min_vegetation_fraction, p.fx);
  od_scaling<NREG>(g.veg_fsd, od);
  p.wall_ext = 0.0;
  p.wall_factor = 0.0;
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) p.f_wall[r] = 0.0;
  if (URBAN) {
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r)
      p.f_wall[r] = (p.frac[r] <= L.cfg.min_vegetation_fraction) ? 0.0 : npw[r] / (kPi * p.frac[r]);
    const double wa = pk[10], wsf = pk[14];
    p.wall_ext = 1.0 - wa * wsf;
    p.wall_factor = wa * (1.0 - wsf);
  }
  region_optics<NREG>(pk[11], pk[12], g.veg_ext, pk[13], od, p.ext_reg, p.ssa_reg);
}

// Element (I, J) of Gamma1 (as used: + Gamma2) and Gamma2, the expressions of layer_unit_sw below
// element by element with the indices at compile time (forest :329-387, urban :424-497)
template <bool URBAN, int NREG, int NS, int I, int J>
SS_HD void sw_gamma_elem(const LG& lg, const SwLayerProps<URBAN, NREG>& p, double& g1, double& g2) {
  constexpr int ri = I / NS, si = I % NS, rj = J / NS, sj = J % NS;
  double e = 0.0;
  if constexpr (I == J) {
    SS_UNROLL_FULL
    for (int jto = 0; jto < NREG; ++jto)
      if (jto != ri) e = e - lg.tan_ang[si] * SS_E(p.fx, jto, ri, NREG);
    if (URBAN) e = e - p.ext_reg[ri] / lg.mu[si] - lg.tan_ang[si] * p.f_wall[ri] * p.wall_ext;
    else e = e - p.ext_reg[ri] / lg.mu[si];
  } else if constexpr (si == sj && ri != rj) {
    e = lg.tan_ang[si] * SS_E(p.fx, ri, rj, NREG);
  }
  g2 = 0.0;
  if constexpr (ri == rj) {
    if (URBAN)
      g2 = 0.5 * (lg.weight[si] * p.ext_reg[ri] * p.ssa_reg[ri] / lg.mu[sj]
                  + lg.vweight[si] * lg.tan_ang[sj] * p.f_wall[ri] * p.wall_factor);
    else
      g2 = 0.5 * lg.weight[si] * p.ext_reg[ri] * p.ssa_reg[ri] / lg.mu[sj];
  }
  g1 = e + g2;
}
// Gamma0 (D x D) and the own-region column of Gamma3 (element (i, region of i); the rest is zero)
template <bool URBAN, int NREG>
SS_HD void sw_gamma0(const SwLayerProps<URBAN, NREG>& p, double* g0) {
  constexpr int D = NREG;
  mat_zero<D, D>(g0);
  SS_UNROLL_FULL
  for (int jfr = 0; jfr < NREG; ++jfr) {
    SS_UNROLL_FULL
    for (int jto = 0; jto < NREG; ++jto) {
      if (jfr != jto) {
        SS_E(g0, jfr, jfr, D) = SS_E(g0, jfr, jfr, D) - p.tan0 * SS_E(p.fx, jto, jfr, NREG);
        SS_E(g0, jto, jfr, D) = p.tan0 * SS_E(p.fx, jto, jfr, NREG);
      }
    }
  }
  SS_UNROLL_FULL
  for (int jreg = 0; jreg < NREG; ++jreg) {
    if (URBAN)
      SS_E(g0, jreg, jreg, D) = SS_E(g0, jreg, jreg, D) - p.ext_reg[jreg] / p.cos_sza
                                - p.tan0 * p.f_wall[jreg] * p.wall_ext;
    else
      SS_E(g0, jreg, jreg, D) = SS_E(g0, jreg, jreg, D) - p.ext_reg[jreg] / p.cos_sza;
  }
}
template <bool URBAN, int NREG>
SS_HD double sw_gamma3_own(const LG& lg, const SwLayerProps<URBAN, NREG>& p, int jreg, int js) {
  return URBAN ? 0.5 * (lg.weight[js] * p.ext_reg[jreg] * p.ssa_reg[jreg]
                        + lg.vweight[js] * p.sin0 * p.f_wall[jreg] * p.wall_factor)
               : 0.5 * lg.weight[js] * p.ext_reg[jreg] * p.ssa_reg[jreg];
}

template <int NF, int D, class W>
SS_HD int sw_direct_part(const double* g0, const double* dsc0, double dz, const W& w, double* mu, double* G0,
                         double* G0inv, double* emu);
template <int NF, int DF, int N, int D, int I0, int R0, class W>
SS_HD int calc_matrices_sw(const double* g0, const double* g1, const double* g2, const double* g3, const double* dsc0,
                           const double* dsc, double dz, const W& w, const CoopEig* coop);
template <int NREG>
SS_HD bool gamma_structure_ok(const double* frac, const double* fx);

// Layer matrices of an (N, D) sub-problem: in symmetrised coordinates (ss_layer_sym.hpp) where the orders
// live in registers - returns -1, nothing written, when the unit declines (the caller then sends the whole
// unit to the out-of-line general route, layer_unit_sw_cold) -, else the general route (ss_layer_matrices.hpp).
template <int NF, int DF, int N, int D, int I0, int R0, bool GENERAL, class W>
SS_HD int calc_matrices_sw_route(const double* g0, const double* g1, const double* g2, const double* g3,
                                 const double* dsc0, const double* dsc, double dz, const W& w,
                                 const CoopEig* coop = nullptr, bool decline_sym = false) {
#if !defined(SS_NO_SYM_LAYER)
  // the register-resident orders, and (rolled loops, matrices in scratch) the orders above those of the
  // row-of-16 kernels: 3 regions x 8 streams = 24 - the general eigen-solver then only sees declined units
  if constexpr (!GENERAL && NF >= 3 && (N <= SS_UNROLL_MAX || sym_scratch_order(N))) {
    int rc = -1;
    if (dsc && !(decline_sym && NF <= SS_UNROLL_MAX)) {
      if constexpr (N <= SS_UNROLL_MAX) rc = sym::calc_matrices_sw_sym<NF, DF, N, D, I0, R0>(g0, g1, g2, g3, dsc0, dsc, dz, w, w.scratch());
      else rc = sym::calc_matrices_sw_sym<NF, DF, N, D, I0, R0>(g0, g1, g2, g3, dsc0, dsc, dz, w, LaneLds<false>{});
    }
    if (rc >= 0 || NF <= SS_UNROLL_MAX) return rc;      // (see calc_matrices_lw_route)
  }
#endif
  return calc_matrices_sw<NF, DF, N, D, I0, R0>(g0, g1, g2, g3, dsc0, dsc, dz, w, coop);
}

template <bool URBAN, int NREG, int NS>
SS_HD_NOINLINE void layer_unit_sw_cold(const Launch* Lp, int cu, int jlay, double* rec);

// First pass of the layer kernel for the orders the row-of-16 kernels take (see layer_first_lw)
template <bool URBAN, int NREG, int NS, class W>
SS_HD void layer_first_sw(const Launch& L, int cu, int jlay, const W& w) {
  constexpr int N = NREG * NS, D = NREG;
  using ER = EigRec<N>;
  const Inputs& in = L.in;
  const LG& lg = L.lg;
  const int col = L.cols[cu / L.nspec];
  if (jlay >= in.nlay[col]) return;
  double pk[kPackSw];
  rec_load<kPackSw>(pack_rec(L, jlay, cu, kPackSw), kRecStride, 0, pk);
  SwLayerProps<URBAN, NREG> p;
  sw_layer_props<URBAN, NREG>(L, in.cos_sza[col], pk, p);
  double* erec = L.eig + rec_base(jlay, L.maxlay, ER::NELEM, cu);
  constexpr long st = kRecStride;
  const bool clear_only = (NREG == 1) || (p.g.veg_fraction <= L.cfg.min_vegetation_fraction);
  const bool full = !clear_only && !(p.frac[0] <= L.cfg.min_vegetation_fraction);
  double status = 0.0;
  int ierr = 0;
  if (full) {
    if (gamma_structure_ok<NREG>(p.frac, p.fx) && L.cfg.decline_cooperative != 1) {
      status = 1.0;
      // The direct-beam part of the unit (D x D: one lane does it for its unit here, the 16 lanes of
      // the row-of-16 kernel would each repeat it): int_dir and E go to the layer record, the
      // eigen-decomposition of Gamma0 and what else ss_layer16.hpp needs to the eigen record
      double g0[D * D], dsc0[D], mu[D], G0[D * D], G0inv[D * D], emu[D];
      sw_gamma0<URBAN, NREG>(p, g0);
      SS_UNROLL_FULL
      for (int r = 0; r < D; ++r) dsc0[r] = sqrt(p.frac[r]);
      ierr = sw_direct_part<N, D>(g0, dsc0, p.g.dz, w, mu, G0, G0inv, emu);
      rec_elem_store(erec, st, 1, p.g.dz);
      // the scalars Gamma1 and Gamma2 are built from (r16::gamma_rows builds the rows in k_pre16)
      r16::GammaScalars<NREG> q;
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) { q.ext[r] = p.ext_reg[r]; q.ssa[r] = p.ssa_reg[r]; q.fw[r] = p.f_wall[r]; }
      SS_UNROLL_FULL
      for (int k = 0; k < NREG * NREG; ++k) q.fx[k] = p.fx[k];
      q.wall_ext = p.wall_ext;
      q.wall_factor = p.wall_factor;
      q.store(erec, st, ER::G1);
      r16::static_for<0, N>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        rec_elem_store(erec, st, ER::VEC + I, dsc0[I / NS] * lg.sqrt_wmu[I % NS]);
        SS_UNROLL_FULL
        for (int r = 0; r < D; ++r)
          rec_elem_store(erec, st, ER::vg3() + I * D + r, (r == I / NS) ? sw_gamma3_own<URBAN, NREG>(lg, p, I / NS, I % NS) : 0.0);
      });
      SS_UNROLL_FULL
      for (int r = 0; r < D; ++r) {
        rec_elem_store(erec, st, ER::vmu(D) + r, mu[r]);
        rec_elem_store(erec, st, ER::vemu(D) + r, emu[r]);
      }
      SS_UNROLL_FULL
      for (int k = 0; k < D * D; ++k) {
        rec_elem_store(erec, st, ER::vg0(D) + k, G0[k]);
        rec_elem_store(erec, st, ER::vg0inv(D) + k, G0inv[k]);
      }
    }
  } else if (clear_only && NS <= SS_UNROLL_MAX) {
    status = -1.0;
    // clear-sky region only: (1:ns, 1:1)
    double g0[D * D], s0[1], s1[NS * NS], s2[NS * NS], s3[NS], dsc0[1], dsc[NS];
    sw_gamma0<URBAN, NREG>(p, g0);
    s0[0] = g0[0];
    dsc0[0] = sqrt(p.frac[0]);
    r16::static_for<0, NS>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      r16::static_for<0, NS>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        sw_gamma_elem<URBAN, NREG, NS, I, J>(lg, p, SS_E(s1, I, J, NS), SS_E(s2, I, J, NS));
      });
      s3[I] = sw_gamma3_own<URBAN, NREG>(lg, p, 0, I);
      dsc[I] = dsc0[0] * lg.sqrt_wmu[I];
    });
    ierr = calc_matrices_sw_route<N, D, NS, 1, 0, 0, true>(s0, s1, s2, s3, dsc0, dsc, p.g.dz, w, nullptr);
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if (ierr != 0 && L.eigen_fail) atomicAdd(L.eigen_fail, 1ULL);
#else
  if (ierr != 0 && L.eigen_fail) *L.eigen_fail += 1ULL;
#endif
  rec_elem_store(erec, st, ER::STATUS, status);
  coop_list_push(L.coop_list, status > 0.5, jlay, cu);
  if (status == 0.0 && L.eigen_fail) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(L.eigen_fail + kSlotOneLaneUnits, 1ULL);
#else
    L.eigen_fail[kSlotOneLaneUnits] += 1ULL;
#endif
  }
}

// W: where the layer matrices go - RecWriter (HBM layer record, three-kernel form) or the
// FusedWriter of k_layer_up (LDS / registers / column record).
// STAGE (orders of the cooperative eigen-solver, ss_eigen_coop.hpp): 0 = everything here;
// 1 = first pass: only the factor W of the full problem goes to the eigen record;
// 2 = second pass: everything, with the eigen-decomposition taken from the record.
template <bool URBAN, int NREG, int NS, class W, int STAGE = 0, bool GENERAL = false>
SS_HD void layer_unit_sw(const Launch& L, int cu, int jlay, const W& w) {
  SS_STAMP_DECL((unsigned long long*)L.eigen_fail);
  constexpr int N = NREG * NS, D = NREG;
  const Inputs& in = L.in;
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  // The pack record of (cu, jlay) exists for every layer slot of the tile, whether the column has that layer or
  // not, and its address does not depend on the column: its loads are requested together with the column
  // index (it used to wait for the column index and then for the layer count: three dependent round trips at
  // the start of every wave).
  double pk[kPackSw];
  constexpr bool kEarlyPack = (STAGE == 0) || !r16_layer(NREG * NS, false);
  if constexpr (kEarlyPack) {
    rec_load<kPackSw>(pack_rec(L, jlay, cu, kPackSw), kRecStride, 0, pk);
    SS_LOAD_FENCE();
  }
  double cos_in;
  if constexpr (kEarlyPack) {
    // ... and the pack record itself says whether the layer exists and carries the column's cos_sza (ss_pack.hpp):
    // ONE round trip before the arithmetic starts
    if (pk[15] == kPackNoLayer) return;
    cos_in = pk[15];
  } else {
    const int nlay_col = in.nlay[col];
    cos_in = in.cos_sza[col];
    SS_LOAD_FENCE();
    if (jlay >= nlay_col) return;
  }
  if constexpr (STAGE == 1 && r16_layer(NREG * NS, false)) {
    layer_first_sw<URBAN, NREG, NS>(L, cu, jlay, w);
    return;
  }
  if constexpr (STAGE == 2 && r16_layer(NREG * NS, false)) {
    // units of the cooperative route had their layer matrices computed by the row-of-16 kernel
    if (rec_elem_load(L.eig + rec_base(jlay, L.maxlay, EigRec<NREG * NS>::NELEM, cu), kRecStride,
                      EigRec<NREG * NS>::STATUS) != 0.0) return;      // 1: row-of-16 kernels, -1: first pass
  }

  const double cos_sza = URBAN ? ss_max(cos_in, 1.0e-6) : cos_in;
  double sin0 = 0.0, tan0;
  if (URBAN) {
    sin0 = sqrt(1.0 - cos_sza * cos_sza);
    tan0 = sin0 / cos_sza;
  } else {
    tan0 = sqrt(1.0 - cos_sza * cos_sza) / ss_max(cos_sza, 1.0e-6);
  }

  // per-layer inputs from the input pack (ss_pack.hpp)
  if constexpr (!kEarlyPack) rec_load<kPackSw>(pack_rec(L, jlay, cu, kPackSw), kRecStride, 0, pk);
  LayerGeom g;
  g.dz = pk[0]; g.veg_fraction = pk[1]; g.veg_scale = pk[2]; g.veg_ext = pk[3];
  g.veg_fsd = pk[4]; g.building_fraction = pk[5]; g.building_scale = pk[6]; g.veg_contact_fraction = pk[7];
  double frac[NREG], np[NREG], npw[NREG], fx[NREG * NREG], od[NREG], f_wall[NREG];
  region_fractions<URBAN, false, NREG>(g.veg_fraction, g.building_fraction, frac);
  norm_perimeters<URBAN, NREG>(L.cfg, g, np, npw);
  exchange_rates<NREG>(frac, np, L.cfg.min_vegetation_fraction, fx);
  od_scaling<NREG>(g.veg_fsd, od);

  double wall_ext = 0.0, wall_factor = 0.0;
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) f_wall[r] = 0.0;
  if (URBAN) {
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r)
      f_wall[r] = (frac[r] <= L.cfg.min_vegetation_fraction) ? 0.0 : npw[r] / (kPi * frac[r]);
    const double wa = pk[10], wsf = pk[14];
    wall_ext = 1.0 - wa * wsf;
    wall_factor = wa * (1.0 - wsf);
  }

  double ext_reg[NREG], ssa_reg[NREG];
  region_optics<NREG>(pk[11], pk[12], g.veg_ext, pk[13], od, ext_reg, ssa_reg);

  // ---- Gamma matrices (forest :329-387, urban :424-497) ----
  double g0[D * D], g1[N * N], g2[N * N], g3[N * D];
  mat_zero<D, D>(g0);
  mat_zero<N, N>(g1);
  mat_zero<N, N>(g2);
  mat_zero<N, D>(g3);
  SS_UNROLL_FULL
  for (int jfr = 0; jfr < NREG; ++jfr) {
    SS_UNROLL_FULL
    for (int jto = 0; jto < NREG; ++jto) {
      if (jfr != jto) {
        SS_E(g0, jfr, jfr, D) = SS_E(g0, jfr, jfr, D) - tan0 * SS_E(fx, jto, jfr, NREG);
        SS_E(g0, jto, jfr, D) = tan0 * SS_E(fx, jto, jfr, NREG);
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) {
          const int ifr = js + jfr * NS, ito = js + jto * NS;
          SS_E(g1, ifr, ifr, N) = SS_E(g1, ifr, ifr, N) - L.lg.tan_ang[js] * SS_E(fx, jto, jfr, NREG);
          SS_E(g1, ito, ifr, N) = L.lg.tan_ang[js] * SS_E(fx, jto, jfr, NREG);
        }
      }
    }
  }
  SS_UNROLL_FULL
  for (int jreg = 0; jreg < NREG; ++jreg) {
    if (URBAN)
      SS_E(g0, jreg, jreg, D) = SS_E(g0, jreg, jreg, D) - ext_reg[jreg] / cos_sza
                                - tan0 * f_wall[jreg] * wall_ext;
    else
      SS_E(g0, jreg, jreg, D) = SS_E(g0, jreg, jreg, D) - ext_reg[jreg] / cos_sza;
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) {
      const int ifr = js + jreg * NS;
      if (URBAN)
        SS_E(g1, ifr, ifr, N) = SS_E(g1, ifr, ifr, N) - ext_reg[jreg] / L.lg.mu[js]
                                - L.lg.tan_ang[js] * f_wall[jreg] * wall_ext;
      else
        SS_E(g1, ifr, ifr, N) = SS_E(g1, ifr, ifr, N) - ext_reg[jreg] / L.lg.mu[js];
    }
  }
  SS_UNROLL_FULL
  for (int jsfr = 0; jsfr < NS; ++jsfr) {
    SS_UNROLL_FULL
    for (int jsto = 0; jsto < NS; ++jsto) {
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg) {
        const int ifr = jsfr + jreg * NS, ito = jsto + jreg * NS;
        if (URBAN)
          SS_E(g2, ito, ifr, N) = 0.5 * (L.lg.weight[jsto] * ext_reg[jreg] * ssa_reg[jreg] / L.lg.mu[jsfr]
                                         + L.lg.vweight[jsto] * L.lg.tan_ang[jsfr] * f_wall[jreg] * wall_factor);
        else
          SS_E(g2, ito, ifr, N) = 0.5 * L.lg.weight[jsto] * ext_reg[jreg] * ssa_reg[jreg] / L.lg.mu[jsfr];
      }
    }
  }
  SS_UNROLL_NN(N)
  for (int k = 0; k < N * N; ++k) g1[k] = g1[k] + g2[k];
  SS_UNROLL_FULL
  for (int jreg = 0; jreg < NREG; ++jreg) {
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) {
      const int ito = js + jreg * NS;
      if (URBAN)
        SS_E(g3, ito, jreg, N) = 0.5 * (L.lg.weight[js] * ext_reg[jreg] * ssa_reg[jreg]
                                        + L.lg.vweight[js] * sin0 * f_wall[jreg] * wall_factor);
      else
        SS_E(g3, ito, jreg, N) = 0.5 * L.lg.weight[js] * ext_reg[jreg] * ssa_reg[jreg];
    }
  }

  // scaling that makes the Gamma matrices symmetric (ss_eigen_sym.hpp)
  double dsc0[D], dsc[N];
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) {
    dsc0[r] = sqrt(frac[r]);
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) dsc[js + r * NS] = dsc0[r] * L.lg.sqrt_wmu[js];
  }

  // ---- layer matrices, 3-way branch (forest :408-455, urban :520-583) ----
  SS_STAMP(0);
  int ierr;
  double* erec = nullptr;
  if constexpr (STAGE != 0) erec = L.eig + rec_base(jlay, L.maxlay, EigRec<N>::NELEM, cu);
  if constexpr (STAGE == 1) {
    const bool full = !(NREG == 1 || g.veg_fraction <= L.cfg.min_vegetation_fraction)
                      && !(frac[0] <= L.cfg.min_vegetation_fraction);
    if (full) {
      if constexpr (r16_layer(N, false)) {
        double size;
        const bool ok = coop_eig_check<N>(g1, g2, dsc, size);
        rec_elem_store(erec, kRecStride, EigRec<N>::STATUS, ok ? 1.0 : 0.0);
        if (ok) {
          // The direct-beam part of the unit (D x D: one lane does it for its unit here, the 16 lanes
          // of the row-of-16 kernel would each repeat it): int_dir and E go to the layer record, the
          // eigen-decomposition of Gamma0 and what else ss_layer16.hpp needs to the eigen record
          using ER = EigRec<N>;
          double mu[D], G0[D * D], G0inv[D * D], emu[D];
          const int rc = sw_direct_part<N, D>(g0, dsc0, g.dz, w, mu, G0, G0inv, emu);
#if defined(__HIP_DEVICE_COMPILE__)
          if (rc != 0 && L.eigen_fail) atomicAdd(L.eigen_fail, 1ULL);
#else
          if (rc != 0 && L.eigen_fail) *L.eigen_fail += 1ULL;
#endif
          rec_elem_store(erec, kRecStride, 1, g.dz);
          rec_elem_store(erec, kRecStride, ER::EV, size);
          rec_store_rowmajor<N>(erec, kRecStride, ER::G1, g1);
          rec_store_rowmajor<N>(erec, kRecStride, ER::G2, g2);
          for (int i = 0; i < N; ++i) {
            rec_elem_store(erec, kRecStride, ER::VEC + i, dsc[i]);
            for (int r = 0; r < D; ++r) rec_elem_store(erec, kRecStride, ER::vg3() + i * D + r, SS_E(g3, i, r, N));
          }
          for (int r = 0; r < D; ++r) {
            rec_elem_store(erec, kRecStride, ER::vmu(D) + r, mu[r]);
            rec_elem_store(erec, kRecStride, ER::vemu(D) + r, emu[r]);
          }
          for (int k = 0; k < D * D; ++k) {
            rec_elem_store(erec, kRecStride, ER::vg0(D) + k, G0[k]);
            rec_elem_store(erec, kRecStride, ER::vg0inv(D) + k, G0inv[k]);
          }
        }
      } else {
        double GD[N * N];
        for (int k = 0; k < N * N; ++k) GD[k] = g1[k] - g2[k];
        coop_eig_pre<N>(GD, g2, dsc, erec, kRecStride);
      }
    } else {
      // a sub-problem: the clear-sky one (order NS, registers) is done right here when the row-of-16
      // kernels take the full problems (status -1: nothing left for anybody); the vegetation-only one
      // (order N - NS, scratch-resident matrices) and everything else waits for the second pass
      const bool clear_only = (NREG == 1) || (g.veg_fraction <= L.cfg.min_vegetation_fraction);
      rec_elem_store(erec, kRecStride, EigRec<N>::STATUS, (r16_layer(N, false) && clear_only && NS <= SS_UNROLL_MAX) ? -1.0 : 0.0);
    }
    const double status = rec_elem_load(erec, kRecStride, EigRec<N>::STATUS);
    if (L.eigen_fail && status > -0.5 && !(status > 0.5)) {
#if defined(__HIP_DEVICE_COMPILE__)
      atomicAdd(L.eigen_fail + kSlotOneLaneUnits, 1ULL);
#else
      L.eigen_fail[kSlotOneLaneUnits] += 1ULL;
#endif
    }
    if (!(status < -0.5)) return;
  }
  const CoopEig ce{erec, kRecStride};
  if (NREG == 1 || g.veg_fraction <= L.cfg.min_vegetation_fraction) {
    // clear-sky region only: (1:ns, 1:1)
    double s0[1], s1[NS * NS], s2[NS * NS], s3[NS];
    get_block<D, 1, 1, 0, 0>(g0, s0);
    get_block<N, NS, NS, 0, 0>(g1, s1);
    get_block<N, NS, NS, 0, 0>(g2, s2);
    get_block<N, NS, 1, 0, 0>(g3, s3);
    ierr = calc_matrices_sw_route<N, D, NS, 1, 0, 0, GENERAL>(s0, s1, s2, s3, dsc0, dsc, g.dz, w, nullptr, SS_DECLINE_SYM(L));
  } else if (STAGE != 1 && frac[0] <= L.cfg.min_vegetation_fraction) {
    // vegetated regions only: (ns+1:n, 2:nreg)
    constexpr int NV = (NREG > 1) ? (NREG - 1) * NS : NS;
    constexpr int DV = (NREG > 1) ? NREG - 1 : 1;
    constexpr int IV = (NREG > 1) ? NS : 0;
    constexpr int RV = (NREG > 1) ? 1 : 0;
    if constexpr (STAGE != 1) {
      double s0[DV * DV], s1[NV * NV], s2[NV * NV], s3[NV * DV];
      get_block<D, DV, DV, RV, RV>(g0, s0);
      get_block<N, NV, NV, IV, IV>(g1, s1);
      get_block<N, NV, NV, IV, IV>(g2, s2);
      get_block<N, NV, DV, IV, RV>(g3, s3);
      ierr = calc_matrices_sw_route<N, D, NV, DV, IV, RV, GENERAL>(s0, s1, s2, s3, dsc0 + RV, dsc + IV, g.dz, w, nullptr, SS_DECLINE_SYM(L));
    } else {
      ierr = 0;
    }
  } else {
    if constexpr (STAGE == 1) ierr = 0;     // (full problems of the first pass went the cooperative way above)
    else ierr = calc_matrices_sw_route<N, D, N, D, 0, 0, GENERAL>(g0, g1, g2, g3, dsc0, dsc, g.dz, w, STAGE == 2 ? &ce : nullptr, SS_DECLINE_SYM(L));
  }
  if constexpr (!GENERAL && STAGE == 0 && N <= SS_UNROLL_MAX) {
    if (ierr < 0) {
      // declined by the symmetrised route (nothing written yet): the whole unit again, out of line
      using FRec = SwRec<N, D>;          // what the general route writes
      using LRec = SwLayRec<N, D>;       // what the pipeline reads
      double rec[FRec::NELEM];
      for (int k = 0; k < FRec::NELEM; ++k) rec[k] = 0.0;
      // (device: the cold unit is handed the kernel-argument segment, which is the descriptor: ss_common.hpp)
      layer_unit_sw_cold<URBAN, NREG, NS>(SS_LAUNCH_PTR_FOR_CALLEE(L), cu, jlay, rec);
      if constexpr (LRec::PACKED) {
        // R^ = D^-1 R D, T^ likewise: lower triangles (reciprocity makes them symmetric)
        double rdv[N];
        for (int i = 0; i < N; ++i) rdv[i] = dsc[i] > 0.0 ? 1.0 / dsc[i] : 0.0;
        put_packed<N, N, 0>(w, LRec::R, [&](auto ic, auto jc) {
          constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
          return rec[FRec::R + i + j * N] * (dsc[j] * rdv[i]);
        });
        put_packed<N, N, 0>(w, LRec::T, [&](auto ic, auto jc) {
          constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
          return rec[FRec::T + i + j * N] * (dsc[j] * rdv[i]);
        });
        // S_up, S_dn: rows scaled; E, int_dir as they are
        for (int j = 0; j < D; ++j)
          for (int i = 0; i < N; ++i) { rec[FRec::SUP + i + j * N] *= rdv[i]; rec[FRec::SDN + i + j * N] *= rdv[i]; }
        SS_UNROLL_FULL
        for (int e = 0; e < FRec::NELEM - FRec::SUP; e += 2)
          w.put2(LRec::SUP + e, rec[FRec::SUP + e], rec[FRec::SUP + e + 1]);
      } else {
        SS_UNROLL_FULL
        for (int e = 0; e < FRec::NELEM; e += 2) w.put2(e, rec[e], rec[e + 1]);
      }
      return;
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if (ierr != 0 && L.eigen_fail) atomicAdd(L.eigen_fail, 1ULL);
#else
  if (ierr != 0 && L.eigen_fail) *L.eigen_fail += 1ULL;
#endif
}

template <bool URBAN, int NREG, int NS>
SS_HD_NOINLINE void layer_unit_sw_cold(const Launch* Lp, int cu, int jlay, double* rec) {
  // (Lp: the kernel-argument segment = the descriptor, formed by the caller IN the kernel: SS_LAUNCH_PTR_FOR_CALLEE)
  const Launch& L = *Lp;
  const ArrayRecWriter w{rec, nullptr};
  layer_unit_sw<URBAN, NREG, NS, ArrayRecWriter, 0, true>(L, cu, jlay, w);
}

template <bool URBAN, int NREG, int NS, int STAGE = 0>
SS_HD void layer_unit_sw(const Launch& L, int cu, int jlay) {
  using Rec = SwLayRec<NREG * NS, NREG>;
  const RecWriter w{L.lay + rec_base(jlay, L.maxlay, Rec::NELEM, cu), kRecStride, (unsigned long long*)L.eigen_fail};
  layer_unit_sw<URBAN, NREG, NS, RecWriter, STAGE>(L, cu, jlay, w);
}

// Gamma1 + Gamma2 (B, N x N) and Gamma3 (g3, N x D) of one layer from its inputs, as
// layer_unit_sw builds them (forest :329-387, urban :424-497; there Gamma1 already
// includes + Gamma2, so B = the exchange / extinction part + 2 Gamma2), with the rows
// and columns outside the layer's sub-problem (3-way branch of layer_unit_sw:
// clear-sky only / vegetated only / full) replaced by the identity.  The downward
// sweep uses them for the integrated fluxes: the reference's int_diff is
// -(Gamma1 + Gamma2)^-1 and int_dir_diff = -2 int_diff Gamma3 Gamma0^-1 (see
// calc_matrices_sw), so int_diff conv + int_dir_diff dir_conv =
// -(Gamma1 + Gamma2)^-1 (conv + 2 Gamma3 int_flux_dir): one solve instead of two
// stored matrices per layer.  inc[i] tells which components belong to the sub-problem.
template <bool URBAN, int NREG, int NS>
SS_HD void sw_int_flux_system(const Launch& L, const LayerGeom& g, double wall_albedo, double wall_specular_frac,
                              double air_ext, double air_ssa, double veg_ssa, double sin0,
                              double* B, double* g3, bool* inc) {
  constexpr int N = NREG * NS, D = NREG;
  double frac[NREG], np[NREG], npw[NREG], fx[NREG * NREG], od[NREG], f_wall[NREG];
  region_fractions<URBAN, false, NREG>(g.veg_fraction, g.building_fraction, frac);
  norm_perimeters<URBAN, NREG>(L.cfg, g, np, npw);
  exchange_rates<NREG>(frac, np, L.cfg.min_vegetation_fraction, fx);
  od_scaling<NREG>(g.veg_fsd, od);
  double wall_ext = 0.0, wall_factor = 0.0;
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) f_wall[r] = 0.0;
  if (URBAN) {
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r)
      f_wall[r] = (frac[r] <= L.cfg.min_vegetation_fraction) ? 0.0 : npw[r] / (kPi * frac[r]);
    wall_ext = 1.0 - wall_albedo * wall_specular_frac;
    wall_factor = wall_albedo * (1.0 - wall_specular_frac);
  }
  double ext_reg[NREG], ssa_reg[NREG];
  region_optics<NREG>(air_ext, air_ssa, g.veg_ext, veg_ssa, od, ext_reg, ssa_reg);

  mat_zero<N, N>(B);
  mat_zero<N, D>(g3);
  SS_UNROLL_FULL
  for (int jfr = 0; jfr < NREG; ++jfr) {
    SS_UNROLL_FULL
    for (int jto = 0; jto < NREG; ++jto) {
      if (jfr != jto) {
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) {
          const int ifr = js + jfr * NS, ito = js + jto * NS;
          SS_E(B, ifr, ifr, N) = SS_E(B, ifr, ifr, N) - L.lg.tan_ang[js] * SS_E(fx, jto, jfr, NREG);
          SS_E(B, ito, ifr, N) = L.lg.tan_ang[js] * SS_E(fx, jto, jfr, NREG);
        }
      }
    }
  }
  SS_UNROLL_FULL
  for (int jreg = 0; jreg < NREG; ++jreg) {
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) {
      const int ifr = js + jreg * NS;
      if (URBAN)
        SS_E(B, ifr, ifr, N) = SS_E(B, ifr, ifr, N) - ext_reg[jreg] / L.lg.mu[js]
                               - L.lg.tan_ang[js] * f_wall[jreg] * wall_ext;
      else
        SS_E(B, ifr, ifr, N) = SS_E(B, ifr, ifr, N) - ext_reg[jreg] / L.lg.mu[js];
    }
  }
  SS_UNROLL_FULL
  for (int jsfr = 0; jsfr < NS; ++jsfr) {
    SS_UNROLL_FULL
    for (int jsto = 0; jsto < NS; ++jsto) {
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg) {
        const int ifr = jsfr + jreg * NS, ito = jsto + jreg * NS;
        double g2;
        if (URBAN)
          g2 = 0.5 * (L.lg.weight[jsto] * ext_reg[jreg] * ssa_reg[jreg] / L.lg.mu[jsfr]
                      + L.lg.vweight[jsto] * L.lg.tan_ang[jsfr] * f_wall[jreg] * wall_factor);
        else
          g2 = 0.5 * L.lg.weight[jsto] * ext_reg[jreg] * ssa_reg[jreg] / L.lg.mu[jsfr];
        // Gamma1 (as used) = exchange / extinction part + Gamma2, so Gamma1 + Gamma2 adds it twice
        SS_E(B, ito, ifr, N) = (SS_E(B, ito, ifr, N) + g2) + g2;
      }
    }
  }
  SS_UNROLL_FULL
  for (int jreg = 0; jreg < NREG; ++jreg) {
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) {
      const int ito = js + jreg * NS;
      if (URBAN)
        SS_E(g3, ito, jreg, N) = 0.5 * (L.lg.weight[js] * ext_reg[jreg] * ssa_reg[jreg]
                                        + L.lg.vweight[js] * sin0 * f_wall[jreg] * wall_factor);
      else
        SS_E(g3, ito, jreg, N) = 0.5 * L.lg.weight[js] * ext_reg[jreg] * ssa_reg[jreg];
    }
  }
  // the sub-problem of this layer (the 3-way branch of layer_unit_sw)
  const bool clear_only = (NREG == 1) || (g.veg_fraction <= L.cfg.min_vegetation_fraction);
  const bool veg_only = !clear_only && (frac[0] <= L.cfg.min_vegetation_fraction);
  SS_UNROLL_FULL
  for (int i = 0; i < N; ++i) inc[i] = clear_only ? (i < NS) : (veg_only ? (i >= NS) : true);
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i)
      if (!(inc[i] && inc[j])) SS_E(B, i, j, N) = (i == j) ? 1.0 : 0.0;
  }
}

// ---- overlap operators (U (x) I_NS), (V (x) I_NS) on the canopy + roof blocks ----

// Interface geometry above layer jlay (0-based; jlay = -1 is the ground):
// U (NREG x DB), V (DB x NREG).
template <bool URBAN, bool LW, int NREG>
SS_HD void interface_overlap_vals(double thr, bool upper_is_sky, double vf_up, double bf_up,
                                  bool lower_is_ground, double vf_lo, double bf_lo, double* U, double* V) {
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  double fu[NREG], fl[DB];
  // upper layer, or the free atmosphere
  if (upper_is_sky) {
    fu[0] = 1.0;
    SS_UNROLL_FULL
    for (int r = 1; r < NREG; ++r) fu[r] = 0.0;
  } else {
    region_fractions<URBAN, LW, NREG>((NREG > 1) ? vf_up : 0.0, URBAN ? bf_up : 0.0, fu);
  }
  if (!URBAN) {
    if (lower_is_ground) {
      fl[0] = 1.0;
      SS_UNROLL_FULL
      for (int r = 1; r < NREG; ++r) fl[r] = 0.0;
    } else {
      region_fractions<URBAN, LW, NREG>((NREG > 1) ? vf_lo : 0.0, 0.0, fl);
    }
    overlap_forest<NREG>(fu, fl, thr, U, V);
  } else {
    if (lower_is_ground) {
      // ground = one "roof" region the size of the non-building part of layer 1
      double s = 0.0;
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) { fl[r] = 0.0; s += fu[r]; }
      fl[DB - 1] = s;
    } else {
      double flo[NREG], sum_up = 0.0;
      region_fractions<URBAN, LW, NREG>((NREG > 1) ? vf_lo : 0.0, bf_lo, flo);
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) sum_up += fu[r];
      urban_frac_lower<NREG>(flo, sum_up, upper_is_sky, fl);
    }
    overlap_urban<NREG>(fu, fl, thr, U, V);
  }
}

// The same from the column's packed arrays: interface between layer jl_lower
// (or the ground, -1) and layer jl_lower + 1 (or the free atmosphere)
template <bool URBAN, bool LW, int NREG>
SS_HD void interface_overlap(const Launch& L, int col, int jl_lower, double* U, double* V) {
  const Inputs& in = L.in;
  const int nlay = in.nlay[col];
  const int l0 = in.istartlay[col] - 1;
  const bool sky = jl_lower + 1 >= nlay, ground = jl_lower < 0;
  double vf_up = 0.0, bf_up = 0.0, vf_lo = 0.0, bf_lo = 0.0;
  if (!sky) {
    if (NREG > 1) vf_up = in.veg_fraction[l0 + jl_lower + 1];
    if (URBAN) bf_up = in.building_fraction[l0 + jl_lower + 1];
  }
  if (!ground) {
    if (NREG > 1) vf_lo = in.veg_fraction[l0 + jl_lower];
    if (URBAN) bf_lo = in.building_fraction[l0 + jl_lower];
  }
  interface_overlap_vals<URBAN, LW, NREG>(L.cfg.min_vegetation_fraction, sky, vf_up, bf_up, ground,
                                          vf_lo, bf_lo, U, V);
}

// y (DB*NS) = (V (x) I) x (NREG*NS)      rect_expandedmat_x_vec :608
template <int NREG, int DB, int NS>
SS_HD void apply_V_vec(const double* V, const double* x, double* y) {
  SS_UNROLL_FULL
  for (int jl = 0; jl < DB; ++jl) {
    SS_UNROLL_FULL
    for (int s = 0; s < NS; ++s) {
      double acc = 0.0;
      SS_UNROLL_FULL
      for (int ju = 0; ju < NREG; ++ju) acc += SS_E(V, jl, ju, DB) * x[ju * NS + s];
      y[jl * NS + s] = acc;
    }
  }
}

// Anew (N x C) = (U (x) I) [ Bc (N x C) ; Broof (NS x C) ]   rect_expandedmat_x_mat :505
template <int NREG, int DB, int NS, int C>
SS_HD void apply_U_mat(const double* U, const double* Bc, const double* Broof, double* Anew) {
  constexpr int N = NREG * NS;
  SS_UNROLL_N(C)
  for (int c = 0; c < C; ++c) {
    SS_UNROLL_FULL
    for (int ju = 0; ju < NREG; ++ju) {
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) {
        double acc = 0.0;
        SS_UNROLL_FULL
        for (int jl = 0; jl < NREG; ++jl) acc += SS_E(U, ju, jl, NREG) * SS_E(Bc, jl * NS + s, c, N);
        if (DB > NREG) acc += SS_E(U, ju, DB - 1, NREG) * SS_E(Broof, s, c, NS);
        SS_E(Anew, ju * NS + s, c, N) = acc;
      }
    }
  }
}

}  // namespace ss
