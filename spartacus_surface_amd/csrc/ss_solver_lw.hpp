// ss_solver_lw.hpp - longwave solver units.
//
//   layer_unit_lw  : one (column, band, layer): region optics + Planck, exchange
//                    and wall rates, Gamma1/Gamma2, emission rates and the layer
//                    matrices R, T, source, int_flux, int_source (sections 3a-3c
//                    of spartacus_forest_lw, radsurf_forest_lw.F90:270-446, and
//                    spartacus_urban_lw, radsurf_urban_lw.F90:281-545).
//   column_unit_lw : one (column, band): adding method with sources (forest
//                    :452-506, urban :551-626) and the internal-emission and
//                    incoming-flux passes (forest :508-694, urban :628-858).
#pragma once

#include "ss_common.hpp"
#include "ss_linalg.hpp"
#include "ss_geometry.hpp"
#include "ss_eigen_coop.hpp"
#include "ss_layer_matrices.hpp"
#include "ss_layer_sym.hpp"
#include "ss_solver_sw.hpp"
#include "ss_pack.hpp"
#include "ss_layer16.hpp"

namespace ss {

// Everything of a layer that both LW units need: geometry, optics, emission
template <bool URBAN, int NREG>
struct LwLayerProps {
  LayerGeom g;
  double frac[NREG], np[NREG], npw[NREG], f_wall[NREG], od[NREG];
  double ext_reg[NREG], ssa_reg[NREG], planck_reg[NREG];
  double vssa, vplanck, vaplanck;
};

// Raw per-layer inputs of the longwave units (loads only, so that they can be
// issued ahead of the arithmetic)
template <bool URBAN, int NREG>
struct LwLayerIn {
  LayerGeom g;
  double air_ext, air_ssa, veg_ssa, veg_planck, veg_air_planck, clear_air_planck;
  double roof_emissivity, roof_emission, wall_emissivity, wall_emission;
  SS_HD void load(const Inputs& in, int lay, long sidx) {
    g = load_layer_geom<URBAN, NREG>(in, lay);
    air_ext = in.air_ext[sidx];
    air_ssa = in.air_ssa[sidx];
    veg_ssa = (NREG > 1) ? in.veg_ssa[sidx] : 0.0;
    veg_planck = (NREG > 1) ? in.veg_planck[sidx] : 0.0;
    veg_air_planck = (NREG > 1) ? in.veg_air_planck[sidx] : 0.0;
    clear_air_planck = in.clear_air_planck[sidx];
    roof_emissivity = roof_emission = wall_emissivity = wall_emission = 0.0;
    if (URBAN) {
      roof_emissivity = in.roof_emissivity[sidx];
      roof_emission = in.roof_emission[sidx];
      wall_emissivity = in.wall_emissivity[sidx];
      wall_emission = in.wall_emission[sidx];
    }
  }
  // The same from the input pack (ss_pack.hpp; rec = pack record of the layer)
  SS_HD void load_pack(const double* rec) {
    double v[kPackLw];
    rec_load<kPackLw>(rec, kRecStride, 0, v);
    g.dz = v[0]; g.veg_fraction = v[1]; g.veg_scale = v[2]; g.veg_ext = v[3];
    g.veg_fsd = v[4]; g.building_fraction = v[5]; g.building_scale = v[6]; g.veg_contact_fraction = v[7];
    air_ext = v[8]; air_ssa = v[9]; veg_ssa = v[10]; veg_planck = v[11];
    veg_air_planck = v[12]; clear_air_planck = v[13];
    roof_emissivity = v[14]; roof_emission = v[15]; wall_emissivity = v[16]; wall_emission = v[17];
  }
};

constexpr double kLwMaxSsa = 1.0 - 1.0e-12;

template <bool URBAN, int NREG>
SS_HD void lw_layer_props(const Launch& L, const LwLayerIn<URBAN, NREG>& r, LwLayerProps<URBAN, NREG>& p) {
  p.g = r.g;
  region_fractions<URBAN, true, NREG>(p.g.veg_fraction, p.g.building_fraction, p.frac);
  norm_perimeters<URBAN, NREG>(L.cfg, p.g, p.np, p.npw);
  od_scaling<NREG>(p.g.veg_fsd, p.od);
  p.vssa = r.veg_ssa;
  p.vplanck = r.veg_planck;
  p.vaplanck = r.veg_air_planck;
  region_optics<NREG>(r.air_ext, r.air_ssa, p.g.veg_ext, p.vssa, p.od, p.ext_reg, p.ssa_reg);
  // A longwave region that does not absorb at all (single-scattering albedo of exactly 1) is taken at 1 - 1e-12: a
  // layer in which NO region absorbs has a singular Gamma1 + Gamma2, the symmetrised route declines it, and the
  // general route that takes it over divides by zero - NaN for the whole column on the device, where the reference
  // returns NaN or a finite number depending on the column (tools/fuzz_parity.py --edge, profiles/round4_edge_kinds.txt).
  // With the clamp the unit stays on the symmetrised route and gets the limit; inputs below 1 - 1e-12 are untouched.
  SS_UNROLL_FULL
  for (int k = 0; k < NREG; ++k) p.ssa_reg[k] = ss_min(p.ssa_reg[k], kLwMaxSsa);
  p.planck_reg[0] = r.clear_air_planck;
  // (radsurf_forest_lw.F90:283-300)
  SS_UNROLL_FULL
  for (int k = 1; k < NREG; ++k)
    p.planck_reg[k] = (p.ext_reg[0] * (1.0 - p.ssa_reg[0]) * p.vaplanck
                       + ((NREG == 2) ? p.g.veg_ext * (1.0 - p.vssa) * p.vplanck
                                      : p.od[k] * p.g.veg_ext * (1.0 - p.vssa) * p.vplanck))
                      / ss_max(p.ext_reg[k] * (1.0 - p.ssa_reg[k]), 1.0e-8);
  SS_UNROLL_FULL
  for (int k = 0; k < NREG; ++k) {
    if (URBAN)   // radsurf_urban_lw.F90:372-381 (note vadjustment2)
      p.f_wall[k] = (p.frac[k] <= L.cfg.min_vegetation_fraction)
                        ? 0.0 : p.npw[k] * L.lg.vadjustment2 / (kPi * p.frac[k]);
    else
      p.f_wall[k] = 0.0;
  }
}

template <bool URBAN, int NREG>
SS_HD void lw_layer_props(const Launch& L, int lay, long sidx, LwLayerProps<URBAN, NREG>& p) {
  LwLayerIn<URBAN, NREG> r;
  r.load(L.in, lay, sidx);
  lw_layer_props<URBAN, NREG>(L, r, p);
}

// Element (I, J) of Gamma1 (as used: exchange / extinction part + Gamma2) and Gamma2 of a layer, the
// expressions of layer_unit_lw below element by element, indices at compile time (forest :329-376,
// urban :383-440): the first pass of the cooperative orders writes the matrices straight from these,
// no N x N array of a lane in between (those arrays lived in scratch memory).
template <bool URBAN, int NREG, int NS, int I, int J>
SS_HD void lw_gamma_elem(const LG& lg, const LwLayerProps<URBAN, NREG>& p, const double* fx, double wall_ext,
                         double wall_factor, double& g1, double& g2) {
  constexpr int ri = I / NS, si = I % NS, rj = J / NS, sj = J % NS;
  double e = 0.0;
  if constexpr (I == J) {
    SS_UNROLL_FULL
    for (int jto = 0; jto < NREG; ++jto)
      if (jto != ri) e = e - lg.tan_ang[si] * SS_E(fx, jto, ri, NREG);
    if (URBAN) e = e - p.ext_reg[ri] / lg.mu[si] - lg.tan_ang[si] * p.f_wall[ri] * wall_ext;
    else e = e - p.ext_reg[ri] / lg.mu[si];
  } else if constexpr (si == sj && ri != rj) {
    e = lg.tan_ang[si] * SS_E(fx, ri, rj, NREG);
  }
  g2 = 0.0;
  if constexpr (ri == rj) {
    if (URBAN)
      g2 = 0.5 * (lg.weight[si] * p.ext_reg[ri] * p.ssa_reg[ri] / lg.mu[sj]
                  + lg.vweight[si] * lg.tan_ang[sj] * p.f_wall[ri] * wall_factor);
    else
      g2 = (0.5 * lg.weight[si] / lg.mu[sj]) * p.ext_reg[ri] * p.ssa_reg[ri];
  }
  g1 = e + g2;
}

// Is the symmetric structure there?  D^-1 Gamma D is symmetric (D = sqrt(region fraction x weight x
// mu)) when the lateral exchange is in detailed balance, fx(a, b) frac(b) = fx(b, a) frac(a), and every
// region has a positive fraction: the in-region and wall scattering terms are symmetric by the
// Legendre-Gauss table alone (ss_eigen_sym.hpp).  The element-wise test of the matrices
// (coop_eig_check) says the same with N^2 loads.
template <int NREG>
SS_HD bool gamma_structure_ok(const double* frac, const double* fx) {
  bool ok = true;
  SS_UNROLL_FULL
  for (int a = 0; a < NREG; ++a) {
    if (!(frac[a] > 0.0)) ok = false;
    SS_UNROLL_FULL
    for (int b = a + 1; b < NREG; ++b) {
      const double x = SS_E(fx, a, b, NREG) * frac[b], y = SS_E(fx, b, a, NREG) * frac[a];
      if (!(fabs(x - y) <= 1.0e-10 * ss_max(fabs(x), fabs(y)))) ok = false;
    }
  }
  return ok;
}

// First pass of the layer kernel for the orders the row-of-16 kernels take (k_layer<..., 1>): full
// problems leave Gamma1, Gamma2, the scaling and the emission rates in the eigen record (status 1),
// clear-sky sub-problems (order NS) are done here (status -1), the rest waits for the second pass (0).
template <bool URBAN, int NREG, int NS, class W>
SS_HD void layer_first_lw(const Launch& L, int cu, int jlay, const W& w) {
  constexpr int N = NREG * NS;
  using ER = EigRec<N>;
  const Inputs& in = L.in;
  const LG& lg = L.lg;
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  if (jlay >= in.nlay[col]) return;
  LwLayerIn<URBAN, NREG> raw;
  raw.load_pack(pack_rec(L, jlay, cu, kPackLw));
  LwLayerProps<URBAN, NREG> p;
  lw_layer_props<URBAN, NREG>(L, raw, p);
  double fx[NREG * NREG];
  exchange_rates<NREG>(p.frac, p.np, L.cfg.min_vegetation_fraction, fx);
  const double wall_ext = 1.0;
  const double wall_factor =
      URBAN ? 1.0 - rec_pair_load(pack_rec(L, jlay, cu - jb, kPackLw) + 16 * kRecStride).x : 0.0;
  double* erec = L.eig + rec_base(jlay, L.maxlay, ER::NELEM, cu);
  constexpr long st = kRecStride;
  // emission rate "b" and the symmetrising scaling, element i
  auto b_of = [&](int jreg, int js) {
    const double volume_emiss = p.frac[jreg] * (p.ext_reg[jreg] * (1.0 - p.ssa_reg[jreg]) * p.planck_reg[jreg]);
    const double wall_emiss = URBAN ? p.npw[jreg] * lg.vadjustment * raw.wall_emission : 0.0;
    return URBAN ? (lg.hweight[js] / lg.mu[js]) * volume_emiss + (0.5 * lg.vweight[js]) * wall_emiss
                 : (lg.hweight[js] / lg.mu[js]) * volume_emiss;
  };
  const bool clear_only = (NREG == 1) || (p.g.veg_fraction <= L.cfg.min_vegetation_fraction);
  const bool full = !clear_only && !(p.frac[0] <= L.cfg.min_vegetation_fraction);
  double status = 0.0;
  if (full) {
    if (gamma_structure_ok<NREG>(p.frac, fx) && L.cfg.decline_cooperative != 1) {
      status = 1.0;
      rec_elem_store(erec, st, 1, p.g.dz);
      // the scalars Gamma1 and Gamma2 are built from (r16::gamma_rows builds the rows in k_pre16)
      r16::GammaScalars<NREG> q;
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) { q.ext[r] = p.ext_reg[r]; q.ssa[r] = p.ssa_reg[r]; q.fw[r] = p.f_wall[r]; }
      SS_UNROLL_FULL
      for (int k = 0; k < NREG * NREG; ++k) q.fx[k] = fx[k];
      q.wall_ext = wall_ext;
      q.wall_factor = wall_factor;
      q.store(erec, st, ER::G1);
      r16::static_for<0, N>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        rec_elem_store(erec, st, ER::VEC + I, sqrt(p.frac[I / NS]) * lg.sqrt_wmu[I % NS]);
        rec_elem_store(erec, st, ER::VEC + N + I, b_of(I / NS, I % NS));
      });
    }
  } else if (clear_only && NS <= SS_UNROLL_MAX) {
    status = -1.0;
    double s1[NS * NS], s2[NS * NS], b[NS], dsc[NS];
    r16::static_for<0, NS>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      r16::static_for<0, NS>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        lw_gamma_elem<URBAN, NREG, NS, I, J>(lg, p, fx, wall_ext, wall_factor, SS_E(s1, I, J, NS), SS_E(s2, I, J, NS));
      });
      b[I] = b_of(0, I);
      dsc[I] = sqrt(p.frac[0]) * lg.sqrt_wmu[I];
    });
    const int ierr = calc_matrices_lw<N, NS, 0>(s1, s2, b, dsc, p.g.dz, w);
#if defined(__HIP_DEVICE_COMPILE__)
    if (ierr != 0 && L.eigen_fail) atomicAdd(L.eigen_fail, 1ULL);
#else
    if (ierr != 0 && L.eigen_fail) *L.eigen_fail += 1ULL;
#endif
  }
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

// Layer matrices of an N sub-problem: in symmetrised coordinates (ss_layer_sym.hpp) where the orders live
// in registers - returns -1, nothing written, when the unit declines (the caller then sends the whole unit
// to the out-of-line general route, layer_unit_lw_cold) -, else the general route (ss_layer_matrices.hpp).
// GENERAL: general route whatever the order (the cold unit itself; -DSS_NO_SYM_LAYER: always, the round-2 form).
template <int NF, int N, int I0, bool GENERAL, class W>
SS_HD int calc_matrices_lw_route(const double* g1, const double* g2, const double* b, const double* dsc, double dz,
                                 const W& w, const CoopEig* coop = nullptr, bool decline_sym = false) {
#if !defined(SS_NO_SYM_LAYER)
  // the register-resident orders, and (rolled loops, matrices in scratch) the orders above those of the
  // row-of-16 kernels: 3 regions x 8 streams = 24 - the general eigen-solver then only sees declined units
  if constexpr (!GENERAL && NF >= 3 && (N <= SS_UNROLL_MAX || sym_scratch_order(N))) {
    int rc = -1;
    if (dsc && !(decline_sym && NF <= SS_UNROLL_MAX)) {
      if constexpr (N <= SS_UNROLL_MAX) rc = sym::calc_matrices_lw_sym<NF, N, I0>(g1, g2, b, dsc, dz, w, w.scratch());
      else rc = sym::calc_matrices_lw_sym<NF, N, I0>(g1, g2, b, dsc, dz, w, LaneLds<false>{});
    }
    // sub-problems of the orders above SS_UNROLL_MAX (second pass of the layer kernel, itself the slow
    // path): general route in line; the small orders hand the decline to their caller
    if (rc >= 0 || NF <= SS_UNROLL_MAX) return rc;
  }
#endif
  return calc_matrices_lw<NF, N, I0>(g1, g2, b, dsc, dz, w, coop);
}

template <bool URBAN, int NREG, int NS>
SS_HD_NOINLINE void layer_unit_lw_cold(const Launch* Lp, int cu, int jlay, double* rec);

// W: RecWriter (HBM layer record) or the FusedWriter of k_layer_up (ss_layer_matrices.hpp)
// (STAGE: see layer_unit_sw)
template <bool URBAN, int NREG, int NS, class W, int STAGE = 0, bool GENERAL = false>
SS_HD void layer_unit_lw(const Launch& L, int cu, int jlay, const W& w) {
  SS_STAMP_DECL((unsigned long long*)L.eigen_fail);
  constexpr int N = NREG * NS;
  const Inputs& in = L.in;
  const LG& lg = L.lg;
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  // (the pack record is requested together with the column index, before the layer count can say whether this
  // unit exists: see layer_unit_sw)
  LwLayerIn<URBAN, NREG> raw;
  double wall_emis1 = 0.0;
  constexpr bool kEarlyPack = (STAGE == 0) || !r16_layer(NREG * NS, true);
  if constexpr (kEarlyPack) {
    raw.load_pack(pack_rec(L, jlay, cu, kPackLw));
    if (URBAN) wall_emis1 = rec_pair_load(pack_rec(L, jlay, cu - jb, kPackLw) + 16 * kRecStride).x;
    SS_LOAD_FENCE();
    if (raw.g.dz == kPackNoLayer) return;      // (the pack record says whether the column has this layer: ss_pack.hpp)
  } else {
    if (jlay >= in.nlay[col]) return;
  }
  if constexpr (STAGE == 1 && r16_layer(NREG * NS, true)) {
    layer_first_lw<URBAN, NREG, NS>(L, cu, jlay, w);
    return;
  }
  if constexpr (STAGE == 2 && r16_layer(NREG * NS, true)) {
    // units of the cooperative route had their layer matrices computed by the row-of-16 kernel
    if (rec_elem_load(L.eig + rec_base(jlay, L.maxlay, EigRec<NREG * NS>::NELEM, cu), kRecStride,
                      EigRec<NREG * NS>::STATUS) != 0.0) return;      // 1: row-of-16 kernels, -1: first pass
  }

  // per-layer inputs from the input pack (ss_pack.hpp)
  if constexpr (!kEarlyPack) {
    raw.load_pack(pack_rec(L, jlay, cu, kPackLw));
    if (URBAN) wall_emis1 = rec_pair_load(pack_rec(L, jlay, cu - jb, kPackLw) + 16 * kRecStride).x;
  }
  LwLayerProps<URBAN, NREG> p;
  lw_layer_props<URBAN, NREG>(L, raw, p);
  double fx[NREG * NREG];
  exchange_rates<NREG>(p.frac, p.np, L.cfg.min_vegetation_fraction, fx);

  // wall_ext = 1, wall_factor = 1 - wall_emissivity(1,jlay) - band 1 for all
  // bands (radsurf_urban_lw.F90:390-392)
  const double wall_ext = 1.0;
  // (band 1 = the pack record of the column's first unit, element 16 = wall_emissivity)
  const double wall_factor = URBAN ? 1.0 - wall_emis1 : 0.0;

  double g1[N * N], g2[N * N], b[N];
  mat_zero<N, N>(g1);
  mat_zero<N, N>(g2);
  SS_UNROLL_FULL
  for (int jfr = 0; jfr < NREG; ++jfr) {
    SS_UNROLL_FULL
    for (int jto = 0; jto < NREG; ++jto) {
      if (jfr != jto) {
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) {
          const int ifr = js + jfr * NS, ito = js + jto * NS;
          SS_E(g1, ifr, ifr, N) = SS_E(g1, ifr, ifr, N) - lg.tan_ang[js] * SS_E(fx, jto, jfr, NREG);
          SS_E(g1, ito, ifr, N) = lg.tan_ang[js] * SS_E(fx, jto, jfr, NREG);
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
        SS_E(g1, ifr, ifr, N) = SS_E(g1, ifr, ifr, N) - p.ext_reg[jreg] / lg.mu[js]
                                - lg.tan_ang[js] * p.f_wall[jreg] * wall_ext;
      else
        SS_E(g1, ifr, ifr, N) = SS_E(g1, ifr, ifr, N) - p.ext_reg[jreg] / lg.mu[js];
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
          SS_E(g2, ito, ifr, N) = 0.5 * (lg.weight[jsto] * p.ext_reg[jreg] * p.ssa_reg[jreg] / lg.mu[jsfr]
                                         + lg.vweight[jsto] * lg.tan_ang[jsfr] * p.f_wall[jreg] * wall_factor);
        else
          SS_E(g2, ito, ifr, N) = (0.5 * lg.weight[jsto] / lg.mu[jsfr]) * p.ext_reg[jreg] * p.ssa_reg[jreg];
      }
    }
  }
  SS_UNROLL_NN(N)
  for (int k = 0; k < N * N; ++k) g1[k] = g1[k] + g2[k];

  // emission rate "b" (forest :377-386, urban :441-456)
  SS_UNROLL_FULL
  for (int jreg = 0; jreg < NREG; ++jreg) {
    const double volume_emiss = p.frac[jreg] * (p.ext_reg[jreg] * (1.0 - p.ssa_reg[jreg]) * p.planck_reg[jreg]);
    const double wall_emiss = URBAN ? p.npw[jreg] * lg.vadjustment * raw.wall_emission : 0.0;
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) {
      const int ifr = js + jreg * NS;
      if (URBAN)
        b[ifr] = (lg.hweight[js] / lg.mu[js]) * volume_emiss + (0.5 * lg.vweight[js]) * wall_emiss;
      else
        b[ifr] = (lg.hweight[js] / lg.mu[js]) * volume_emiss;
    }
  }

  // scaling that makes the Gamma matrices symmetric (ss_eigen_sym.hpp)
  double dsc[N];
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) {
    const double dr = sqrt(p.frac[r]);
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) dsc[js + r * NS] = dr * lg.sqrt_wmu[js];
  }

  SS_STAMP(10);
  int ierr;
  double* erec = nullptr;
  if constexpr (STAGE != 0) erec = L.eig + rec_base(jlay, L.maxlay, EigRec<N>::NELEM, cu);
  if constexpr (STAGE == 1) {
    const bool full = !(NREG == 1 || p.g.veg_fraction <= L.cfg.min_vegetation_fraction)
                      && !(p.frac[0] <= L.cfg.min_vegetation_fraction);
    if (full) {
      if constexpr (r16_layer(N, true)) {
        // what the row-of-16 kernels (ss_layer16.hpp: eig_pre, layer_lw) need of this unit, matrices row-major
        double size;
        const bool ok = coop_eig_check<N>(g1, g2, dsc, size);
        rec_elem_store(erec, kRecStride, EigRec<N>::STATUS, ok ? 1.0 : 0.0);
        if (ok) {
          rec_elem_store(erec, kRecStride, 1, p.g.dz);
          rec_elem_store(erec, kRecStride, EigRec<N>::EV, size);
          rec_store_rowmajor<N>(erec, kRecStride, EigRec<N>::G1, g1);
          rec_store_rowmajor<N>(erec, kRecStride, EigRec<N>::G2, g2);
          for (int i = 0; i < N; ++i) {
            rec_elem_store(erec, kRecStride, EigRec<N>::VEC + i, dsc[i]);
            rec_elem_store(erec, kRecStride, EigRec<N>::VEC + N + i, b[i]);
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
      const bool clear_only = (NREG == 1) || (p.g.veg_fraction <= L.cfg.min_vegetation_fraction);
      rec_elem_store(erec, kRecStride, EigRec<N>::STATUS, (r16_layer(N, true) && clear_only && NS <= SS_UNROLL_MAX) ? -1.0 : 0.0);
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
  if (NREG == 1 || p.g.veg_fraction <= L.cfg.min_vegetation_fraction) {
    double s1[NS * NS], s2[NS * NS];
    get_block<N, NS, NS, 0, 0>(g1, s1);
    get_block<N, NS, NS, 0, 0>(g2, s2);
    ierr = calc_matrices_lw_route<N, NS, 0, GENERAL>(s1, s2, b, dsc, p.g.dz, w, nullptr, SS_DECLINE_SYM(L));
  } else if (STAGE != 1 && p.frac[0] <= L.cfg.min_vegetation_fraction) {
    constexpr int NV = (NREG > 1) ? (NREG - 1) * NS : NS;
    constexpr int IV = (NREG > 1) ? NS : 0;
    if constexpr (STAGE != 1) {
      double s1[NV * NV], s2[NV * NV];
      get_block<N, NV, NV, IV, IV>(g1, s1);
      get_block<N, NV, NV, IV, IV>(g2, s2);
      ierr = calc_matrices_lw_route<N, NV, IV, GENERAL>(s1, s2, b + IV, dsc + IV, p.g.dz, w, nullptr, SS_DECLINE_SYM(L));
    } else {
      ierr = 0;
    }
  } else {
    if constexpr (STAGE == 1) ierr = 0;     // (full problems of the first pass went the cooperative way above)
    else ierr = calc_matrices_lw_route<N, N, 0, GENERAL>(g1, g2, b, dsc, p.g.dz, w, STAGE == 2 ? &ce : nullptr, SS_DECLINE_SYM(L));
  }
  if constexpr (!GENERAL && STAGE == 0 && N <= SS_UNROLL_MAX) {
    if (ierr < 0) {
      // declined by the symmetrised route (nothing written yet): the whole unit again, general route, out of
      // line - the cold path must not keep the Gamma matrices of this path alive, nor set its register budget
      using FRec = LwRec<N>;          // what the general route writes
      using LRec = LwLayRec<N>;       // what the pipeline reads
      double rec[FRec::NELEM];
      for (int k = 0; k < FRec::NELEM; ++k) rec[k] = 0.0;
      // (device: the cold unit is handed the kernel-argument segment, which is the descriptor: ss_common.hpp)
      layer_unit_lw_cold<URBAN, NREG, NS>(SS_LAUNCH_PTR_FOR_CALLEE(L), cu, jlay, rec);
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
        SS_UNROLL_FULL
        for (int e = 0; e < FRec::NELEM - FRec::IFLUX; e += 2)
          w.put2(LRec::IFLUX + e, rec[FRec::IFLUX + e], rec[FRec::IFLUX + e + 1]);
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
SS_HD_NOINLINE void layer_unit_lw_cold(const Launch* Lp, int cu, int jlay, double* rec) {
  // (Lp: the kernel-argument segment = the descriptor, formed by the caller IN the kernel: SS_LAUNCH_PTR_FOR_CALLEE)
  const Launch& L = *Lp;
  const ArrayRecWriter w{rec, nullptr};
  layer_unit_lw<URBAN, NREG, NS, ArrayRecWriter, 0, true>(L, cu, jlay, w);
}

template <bool URBAN, int NREG, int NS, int STAGE = 0>
SS_HD void layer_unit_lw(const Launch& L, int cu, int jlay) {
  using Rec = LwLayRec<NREG * NS>;
  const RecWriter w{L.lay + rec_base(jlay, L.maxlay, Rec::NELEM, cu), kRecStride, (unsigned long long*)L.eigen_fail};
  layer_unit_lw<URBAN, NREG, NS, RecWriter, STAGE>(L, cu, jlay, w);
}

}  // namespace ss
