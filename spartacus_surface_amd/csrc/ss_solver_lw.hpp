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
#include "ss_layer_matrices.hpp"
#include "ss_solver_sw.hpp"

namespace ss {

template <int N>
struct LwColRec {
  static constexpr int AAB = 0;               // a_above(l)          N x N
  static constexpr int SAB = AAB + N * N;     // source_above(l)     N
  static constexpr int DEN = SAB + N;         // lu(denominator(l))  N x N
  static constexpr int ABE = DEN + N * N;     // a_below(l+1)        N x N (canopy part)
  static constexpr int SBE = ABE + N * N;     // source_below(l+1)   N     (canopy part)
  static constexpr int NELEM = SBE + N;
};

// Everything of a layer that both LW units need: geometry, optics, emission
template <bool URBAN, int NREG>
struct LwLayerProps {
  LayerGeom g;
  double frac[NREG], np[NREG], npw[NREG], f_wall[NREG], od[NREG];
  double ext_reg[NREG], ssa_reg[NREG], planck_reg[NREG];
  double vssa, vplanck, vaplanck;
};

template <bool URBAN, int NREG>
SS_HD void lw_layer_props(const Launch& L, int lay, long sidx, LwLayerProps<URBAN, NREG>& p) {
  const Inputs& in = L.in;
  p.g = load_layer_geom<URBAN, NREG>(in, lay);
  region_fractions<URBAN, true, NREG>(p.g.veg_fraction, p.g.building_fraction, p.frac);
  norm_perimeters<URBAN, NREG>(L.cfg, p.g, p.np, p.npw);
  od_scaling<NREG>(p.g.veg_fsd, p.od);
  p.vssa = (NREG > 1) ? in.veg_ssa[sidx] : 0.0;
  p.vplanck = (NREG > 1) ? in.veg_planck[sidx] : 0.0;
  p.vaplanck = (NREG > 1) ? in.veg_air_planck[sidx] : 0.0;
  region_optics<NREG>(in.air_ext[sidx], in.air_ssa[sidx], p.g.veg_ext, p.vssa, p.od,
                      p.ext_reg, p.ssa_reg);
  p.planck_reg[0] = in.clear_air_planck[sidx];
  // (radsurf_forest_lw.F90:283-300)
  SS_UNROLL_FULL
  for (int r = 1; r < NREG; ++r)
    p.planck_reg[r] = (p.ext_reg[0] * (1.0 - p.ssa_reg[0]) * p.vaplanck
                       + ((NREG == 2) ? p.g.veg_ext * (1.0 - p.vssa) * p.vplanck
                                      : p.od[r] * p.g.veg_ext * (1.0 - p.vssa) * p.vplanck))
                      / ss_max(p.ext_reg[r] * (1.0 - p.ssa_reg[r]), 1.0e-8);
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) {
    if (URBAN)   // radsurf_urban_lw.F90:372-381 (note vadjustment2)
      p.f_wall[r] = (p.frac[r] <= L.cfg.min_vegetation_fraction)
                        ? 0.0 : p.npw[r] * L.lg.vadjustment2 / (kPi * p.frac[r]);
    else
      p.f_wall[r] = 0.0;
  }
}

template <bool URBAN, int NREG, int NS>
SS_HD void layer_unit_lw(const Launch& L, int cu, int jlay) {
  constexpr int N = NREG * NS;
  using Rec = LwRec<N>;
  const Inputs& in = L.in;
  const LG& lg = L.lg;
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  if (jlay >= in.nlay[col]) return;
  const int lay = in.istartlay[col] - 1 + jlay;
  const long sidx = (long)jb + (long)L.nspec * lay;

  LwLayerProps<URBAN, NREG> p;
  lw_layer_props<URBAN, NREG>(L, lay, sidx, p);
  double fx[NREG * NREG];
  exchange_rates<NREG>(p.frac, p.np, L.cfg.min_vegetation_fraction, fx);

  // wall_ext = 1, wall_factor = 1 - wall_emissivity(1,jlay) - band 1 for all
  // bands (radsurf_urban_lw.F90:390-392)
  const double wall_ext = 1.0;
  const double wall_factor = URBAN ? 1.0 - in.wall_emissivity[(long)L.nspec * lay] : 0.0;

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
    const double wall_emiss = URBAN ? p.npw[jreg] * lg.vadjustment * in.wall_emission[sidx] : 0.0;
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) {
      const int ifr = js + jreg * NS;
      if (URBAN)
        b[ifr] = (lg.hweight[js] / lg.mu[js]) * volume_emiss + (0.5 * lg.vweight[js]) * wall_emiss;
      else
        b[ifr] = (lg.hweight[js] / lg.mu[js]) * volume_emiss;
    }
  }

  RecWriter w{L.lay + ((long)jlay * Rec::NELEM) * L.ncu + cu, (long)L.ncu};
  int ierr;
  if (NREG == 1 || p.g.veg_fraction <= L.cfg.min_vegetation_fraction) {
    double s1[NS * NS], s2[NS * NS];
    get_block<N, NS, NS, 0, 0>(g1, s1);
    get_block<N, NS, NS, 0, 0>(g2, s2);
    ierr = calc_matrices_lw<N, NS, 0>(s1, s2, b, p.g.dz, w);
  } else if (p.frac[0] <= L.cfg.min_vegetation_fraction) {
    constexpr int NV = (NREG > 1) ? (NREG - 1) * NS : NS;
    constexpr int IV = (NREG > 1) ? NS : 0;
    double s1[NV * NV], s2[NV * NV];
    get_block<N, NV, NV, IV, IV>(g1, s1);
    get_block<N, NV, NV, IV, IV>(g2, s2);
    ierr = calc_matrices_lw<N, NV, IV>(s1, s2, b + IV, p.g.dz, w);
  } else {
    ierr = calc_matrices_lw<N, N, 0>(g1, g2, b, p.g.dz, w);
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if (ierr != 0 && L.eigen_fail) atomicAdd(L.eigen_fail, 1ULL);
#else
  if (ierr != 0 && L.eigen_fail) *L.eigen_fail += 1ULL;
#endif
}

template <bool URBAN, int NREG, int NS>
SS_HD void column_unit_lw(const Launch& L, int cu) {
  constexpr int N = NREG * NS;
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  constexpr int NB = DB * NS;
  using LRec = LwRec<N>;
  using CRec = LwColRec<N>;
  const Inputs& in = L.in;
  const LG& lg = L.lg;
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  const int nspec = L.nspec;
  const int nlay = in.nlay[col];
  const int l0 = in.istartlay[col] - 1;
  const long stride = L.ncu;
  const long cidx = (long)jb + (long)nspec * col;

  // ---- section 4: adding method with sources ----
  double Aab[N * N], Sab[N];
  mat_zero<N, N>(Aab);
  {
    const double gem = in.ground_emissivity[cidx], gemission = in.ground_emission[cidx];
    double frac1[NREG];
    region_fractions<URBAN, true, NREG>((NREG > 1) ? in.veg_fraction[l0] : 0.0,
                                        URBAN ? in.building_fraction[l0] : 0.0, frac1);
    SS_UNROLL_FULL
    for (int jreg = 0; jreg < NREG; ++jreg) {
      SS_UNROLL_FULL
      for (int jsto = 0; jsto < NS; ++jsto) {
        SS_UNROLL_FULL
        for (int jsfr = 0; jsfr < NS; ++jsfr)
          SS_E(Aab, jsto + jreg * NS, jsfr + jreg * NS, N) = (1.0 - gem) * lg.hweight[jsto];
      }
      SS_UNROLL_FULL
      for (int js = 0; js < NS; ++js) Sab[js + jreg * NS] = (lg.hweight[js] * frac1[jreg]) * gemission;
    }
  }

  for (int jlay = 0; jlay < nlay; ++jlay) {
    const double* lrec = L.lay + ((long)jlay * LRec::NELEM) * stride + cu;
    double* crec = L.col + ((long)jlay * CRec::NELEM) * stride + cu;
    const int lay = l0 + jlay;
    const long sidx = (long)jb + (long)nspec * lay;
    double Abe[N * N], Sbe[N];
    {
      double R[N * N], T[N * N], Den[N * N], X[N * N], Sl[N];
      rec_load<N * N>(lrec, stride, LRec::R, R);
      rec_load<N * N>(lrec, stride, LRec::T, T);
      rec_load<N>(lrec, stride, LRec::SRC, Sl);
      rec_store<N * N>(crec, stride, CRec::AAB, Aab);
      rec_store<N>(crec, stride, CRec::SAB, Sab);
      mat_mul<N, N, N>(Aab, R, Den);
      SS_UNROLL_NN(N)
      for (int k = 0; k < N * N; ++k) Den[k] = -Den[k];
      SS_UNROLL_N(N)
      for (int j = 0; j < N; ++j) SS_E(Den, j, j, N) = 1.0 + SS_E(Den, j, j, N);
      lu_factor<N>(Den);
      rec_store<N * N>(crec, stride, CRec::DEN, Den);
      mat_mul<N, N, N>(Aab, T, X);
      lu_solve_mat<N, N>(Den, X);
      mat_mul<N, N, N>(T, X, Abe);
      SS_UNROLL_NN(N)
      for (int k = 0; k < N * N; ++k) Abe[k] = R[k] + Abe[k];
      rec_store<N * N>(crec, stride, CRec::ABE, Abe);
      // Eq. 34: source_below = source_lay + T Den^-1 (source_above + a_above source_lay)
      double v[N];
      mat_vec<N, N>(Aab, Sl, v);
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) v[i] = Sab[i] + v[i];
      lu_solve<N>(Den, v);
      mat_vec<N, N>(T, v, Sbe);
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) Sbe[i] = Sl[i] + Sbe[i];
      rec_store<N>(crec, stride, CRec::SBE, Sbe);
    }
    {
      double U[NREG * DB], V[DB * NREG];
      interface_overlap<URBAN, true, NREG>(L, col, jlay, U, V);
      double roof_a = 0.0, roof_emis = 0.0, exposed_roof_frac = 0.0;
      if (URBAN) {
        // urban :587-605
        if (jlay < nlay - 1)
          exposed_roof_frac = ss_max(0.0, in.building_fraction[lay] - in.building_fraction[lay + 1]);
        else
          exposed_roof_frac = in.building_fraction[lay];
        roof_a = 1.0 - in.roof_emissivity[sidx];
        roof_emis = in.roof_emission[sidx];
      }
      double tA[N * N], tAroof[NS * N];
      SS_UNROLL_FULL
      for (int ju = 0; ju < NREG; ++ju) {
        SS_UNROLL_FULL
        for (int s = 0; s < NS; ++s) {
          const int c = ju * NS + s;
          SS_UNROLL_N(N)
          for (int i = 0; i < N; ++i) {
            double acc = 0.0;
            SS_UNROLL_FULL
            for (int jl = 0; jl < NREG; ++jl) acc += SS_E(Abe, i, jl * NS + s, N) * SS_E(V, jl, ju, DB);
            SS_E(tA, i, c, N) = acc;
          }
          SS_UNROLL_FULL
          for (int sr = 0; sr < NS; ++sr)
            SS_E(tAroof, sr, c, NS) = URBAN ? (roof_a * lg.hweight[sr]) * SS_E(V, DB - 1, ju, DB) : 0.0;
        }
      }
      apply_U_mat<NREG, DB, NS, N>(U, tA, tAroof, Aab);
      // source_above(l+1) = (U (x) I) source_below(l+1)   (incl. the roof part)
      double sroof[NS];
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) sroof[s] = URBAN ? lg.hweight[s] * roof_emis * exposed_roof_frac : 0.0;
      apply_U_mat<NREG, DB, NS, 1>(U, Sbe, sroof, Sab);
    }
  }

  double top_emissivity, top_emission;
  {
    double s = 0.0;
    SS_UNROLL_FULL
    for (int j1 = 0; j1 < NS; ++j1) {
      double acc = 0.0;
      SS_UNROLL_FULL
      for (int j2 = 0; j2 < NS; ++j2) acc += SS_E(Aab, j1, j2, N) * lg.hweight[j2];
      s += acc;
    }
    top_emissivity = 1.0 - s;
    s = 0.0;
    SS_UNROLL_FULL
    for (int j1 = 0; j1 < NS; ++j1) s += Sab[j1];
    top_emission = s;
    L.top1[cidx] = top_emissivity;
    L.top2[cidx] = top_emission;
  }

#define PUTC(fd, F, v) do { if ((fd).f[F]) (fd).f[F][cidx] = (v); } while (0)
#define PUTL(fd, F, v) do { if ((fd).f[F]) (fd).f[F][sidx] = (v); } while (0)

  // ---- section 5: kind 0 = internal emission, kind 1 = incoming at canopy top ----
  double gvd_int = 0.0;   // forest: both passes go to lw_internal (Appendix B item 2)
  SS_UNROLL_FULL
  for (int kind = 0; kind < 2; ++kind) {
    const FluxDev& f = (kind == 0) ? L.flux_a : L.flux_b;
    double dn_above[N], up_above[N];
    vec_zero<N>(dn_above);
    vec_zero<N>(up_above);
    if (kind == 0) {
      PUTC(f, F_TOP_DN, 0.0);
      PUTC(f, F_TOP_NET, -top_emission);
    } else {
      SS_UNROLL_FULL
      for (int js = 0; js < NS; ++js) dn_above[js] = lg.hweight[js];
      PUTC(f, F_TOP_DN, 1.0);
      PUTC(f, F_TOP_NET, top_emissivity);
    }
    for (int jlay = nlay - 1; jlay >= 0; --jlay) {
      const double* lrec = L.lay + ((long)jlay * LRec::NELEM) * stride + cu;
      const double* crec = L.col + ((long)jlay * CRec::NELEM) * stride + cu;
      const int lay = l0 + jlay;
      const long sidx = (long)jb + (long)nspec * lay;
      LwLayerProps<URBAN, NREG> p;
      lw_layer_props<URBAN, NREG>(L, lay, sidx, p);
      double U[NREG * DB], V[DB * NREG];
      interface_overlap<URBAN, true, NREG>(L, col, jlay, U, V);

      double dn_below[NB], up_below[NB];
      apply_V_vec<NREG, DB, NS>(V, dn_above, dn_below);
      double M[N * N];
      rec_load<N * N>(crec, stride, CRec::ABE, M);
      mat_vec<N, N>(M, dn_below, up_below);
      if (kind == 0) {
        double Sbe[N];
        rec_load<N>(crec, stride, CRec::SBE, Sbe);
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) up_below[i] = up_below[i] + Sbe[i];
      }
      if (URBAN) {
        const double roof_a = 1.0 - in.roof_emissivity[sidx];
        double exposed_roof_frac;
        if (jlay < nlay - 1)
          exposed_roof_frac = ss_max(0.0, in.building_fraction[lay] - in.building_fraction[lay + 1]);
        else
          exposed_roof_frac = in.building_fraction[lay];
        double sdn = 0.0, sup = 0.0;
        SS_UNROLL_FULL
        for (int s = 0; s < NS; ++s) sdn += dn_below[N + s];
        SS_UNROLL_FULL
        for (int s = 0; s < NS; ++s) {
          double acc = 0.0;
          SS_UNROLL_FULL
          for (int s2 = 0; s2 < NS; ++s2) acc += (roof_a * lg.hweight[s]) * dn_below[N + s2];
          if (kind == 0) acc = acc + lg.hweight[s] * in.roof_emission[sidx] * exposed_roof_frac;
          up_below[N + s] = acc;
          sup += acc;
        }
        PUTL(f, F_ROOF_IN, sdn);
        PUTL(f, F_ROOF_NET, sdn - sup);
      } else {
        PUTL(f, F_ROOF_IN, 0.0);
        PUTL(f, F_ROOF_NET, 0.0);
      }

      double rhs[N], Sab_l[N];
      rec_load<N * N>(lrec, stride, LRec::T, M);
      mat_vec<N, N>(M, dn_below, rhs);
      if (kind == 0) {
        rec_load<N>(crec, stride, CRec::SAB, Sab_l);
        rec_load<N * N>(lrec, stride, LRec::R, M);
        double t2[N], Sl[N];
        mat_vec<N, N>(M, Sab_l, t2);
        rec_load<N>(lrec, stride, LRec::SRC, Sl);
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) rhs[i] = (rhs[i] + t2[i] + Sl[i]);
      }
      rec_load<N * N>(crec, stride, CRec::DEN, M);
      lu_solve<N>(M, rhs);
      vec_copy<N>(rhs, dn_above);
      rec_load<N * N>(crec, stride, CRec::AAB, M);
      mat_vec<N, N>(M, dn_above, up_above);
      if (kind == 0) {
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) up_above[i] = up_above[i] + Sab_l[i];
      }

      if (f.f[F_FLUX_DN_LAYER_TOP]) {
        double s = 0.0;
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) s += dn_below[i];
        PUTL(f, F_FLUX_DN_LAYER_TOP, s);
        s = 0.0;
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) s += up_below[i];
        PUTL(f, F_FLUX_UP_LAYER_TOP, s);
        s = 0.0;
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) s += dn_above[i];
        PUTL(f, F_FLUX_DN_LAYER_BASE, s);
        s = 0.0;
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) s += up_above[i];
        PUTL(f, F_FLUX_UP_LAYER_BASE, s);
      }

      double conv[N], int_flux[N];
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) conv[i] = dn_below[i] + up_above[i];
      rec_load<N * N>(lrec, stride, LRec::IFLUX, M);
      mat_vec<N, N>(M, conv, int_flux);
      if (kind == 0) {
        double Is[N];
        rec_load<N>(lrec, stride, LRec::ISRC, Is);
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) int_flux[i] = int_flux[i] + Is[i];
      }

      // absorption (forest :571-590 / :658-672, urban :700-737 / :811-835)
      const double air_ext = in.air_ext[sidx], air_ssa = in.air_ssa[sidx];
      double emiss_factor = 0.0;
      SS_UNROLL_FULL
      for (int js = 0; js < NS; ++js) emiss_factor += lg.hweight[js] / lg.mu[js];
      emiss_factor = 2.0 * emiss_factor;
      {
        double s = 0.0;
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) s += int_flux[js] * (1.0 / lg.mu[js]);
        if (kind == 0) {
          const double volume_emiss = p.frac[0] * (p.ext_reg[0] * (1.0 - p.ssa_reg[0]) * p.planck_reg[0]);
          PUTL(f, F_CLEAR_AIR_ABS, 0.0 + air_ext * (1.0 - air_ssa) * s - (emiss_factor * volume_emiss) * p.g.dz);
        } else {
          PUTL(f, F_CLEAR_AIR_ABS, 0.0 + air_ext * (1.0 - air_ssa) * s);
        }
      }
      if (NREG > 1) {
        double vaa = 0.0, va = 0.0;
        SS_UNROLL_FULL
        for (int jreg = 1; jreg < NREG; ++jreg) {
          double s = 0.0;
          SS_UNROLL_FULL
          for (int js = 0; js < NS; ++js) s += int_flux[jreg * NS + js] * (1.0 / lg.mu[js]);
          if (kind == 0) {
            const double emiss_air = emiss_factor * p.frac[jreg] * p.ext_reg[0] * (1.0 - p.ssa_reg[0]) * p.vaplanck;
            const double emiss_veg = emiss_factor * p.frac[jreg] * p.g.veg_ext * (1.0 - p.vssa) * p.vplanck * p.od[jreg];
            vaa = vaa + air_ext * (1.0 - air_ssa) * s - emiss_air * p.g.dz;
            va = va + p.g.veg_ext * (1.0 - p.vssa) * s * p.od[jreg] - emiss_veg * p.g.dz;
          } else {
            vaa = vaa + air_ext * (1.0 - air_ssa) * s;
            va = va + p.g.veg_ext * (1.0 - p.vssa) * s * p.od[jreg];
          }
        }
        PUTL(f, F_VEG_AIR_ABS, vaa);
        PUTL(f, F_VEG_ABS, va);
      } else {
        PUTL(f, F_VEG_AIR_ABS, 0.0);
        PUTL(f, F_VEG_ABS, 0.0);
      }
      if (URBAN) {
        double wall_in = 0.0, npw_sum = 0.0;
        SS_UNROLL_FULL
        for (int jreg = 0; jreg < NREG; ++jreg) {
          double s = 0.0;
          SS_UNROLL_FULL
          for (int js = 0; js < NS; ++js) s += int_flux[jreg * NS + js] * lg.tan_ang[js];
          wall_in = wall_in + p.f_wall[jreg] * s;
          npw_sum += p.npw[jreg];
        }
        PUTL(f, F_WALL_IN, wall_in);
        if (kind == 0) {
          const double emiss_wall = (npw_sum * lg.vadjustment) * in.wall_emission[sidx];
          PUTL(f, F_WALL_NET, wall_in * in.wall_emissivity[sidx] - emiss_wall * p.g.dz);
        } else {
          PUTL(f, F_WALL_NET, wall_in * in.wall_emissivity[sidx]);
        }
      } else {
        PUTL(f, F_WALL_IN, 0.0);
        PUTL(f, F_WALL_NET, 0.0);
      }
    }
    {
      double s2 = 0.0, s3 = 0.0, gvd = 0.0;
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) s2 += dn_above[i];
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) s3 += up_above[i];
      PUTC(f, F_GROUND_DN, s2);
      PUTC(f, F_GROUND_NET, s2 - s3);
      if (URBAN) {
        SS_UNROLL_FULL
        for (int jreg = 0; jreg < NREG; ++jreg) {
          SS_UNROLL_FULL
          for (int js = 0; js < NS; ++js) {
            const int ifr = js + jreg * NS;
            gvd = gvd + (dn_above[ifr] + up_above[ifr]) * lg.tan_ang[js] / kPi;
          }
        }
        PUTC(f, F_GROUND_VERTICAL_DIFF, gvd);
      } else {
        // forest: the incoming pass also accumulates into lw_internal and
        // lw_norm%ground_vertical_diff stays zero (radsurf_forest_lw.F90:691)
        SS_UNROLL_FULL
        for (int jreg = 0; jreg < NREG; ++jreg) {
          SS_UNROLL_FULL
          for (int js = 0; js < NS; ++js) {
            const int ifr = js + jreg * NS;
            gvd_int = gvd_int + (dn_above[ifr] + up_above[ifr]) * lg.tan_ang[js] / kPi;
          }
        }
        if (kind == 1) {
          PUTC(L.flux_a, F_GROUND_VERTICAL_DIFF, gvd_int);
          PUTC(L.flux_b, F_GROUND_VERTICAL_DIFF, 0.0);
        }
      }
    }
  }
#undef PUTC
#undef PUTL
}

}  // namespace ss
