// ss_down16.hpp - the downward flux passes of the orders 7..16 with one column unit on the 16 lanes
// of a DPP row (ss_row16.hpp).  Same steps, same records, same outputs as column_down_sw /
// column_down_lw (ss_column_sw.hpp, ss_column_lw.hpp; reference radsurf_forest_sw.F90:530-760,
// radsurf_urban_sw.F90:676-984, radsurf_forest_lw.F90:508-694, radsurf_urban_lw.F90:628-858).
// Flux vectors are one element per lane, BOTH passes of a spectral region side by side (pass 0 =
// direct / internal emission, pass 1 = diffuse / incoming flux: every broadcast-multiply-add
// serves the two); the exposed-roof components sit in a second vector on the lanes 0 .. NS-1;
// direct-beam vectors (NREG elements) and every per-layer scalar are held by all lanes.
// Sums over regions and streams are sums over lanes: multiply-adds with a broadcast operand.
#pragma once

#include "ss_column16.hpp"

namespace ss {
namespace r16 {

// s[p] = sum over the lanes 0 .. NL-1 of x[p], on every lane (both passes)
template <int NL>
SS_HD void all_sum2(const rv* x, double* s) {
  rv acc[2] = {0.0, 0.0};
  const rv one = 1.0;
  static_for<0, NL>([&](auto kc) { row_fma<decltype(kc)::value, 2>(acc, x, one); });
  s[0] = uniform(acc[0]);
  s[1] = uniform(acc[1]);
}
// per region: s[r][p] = sum_js x[p](r, js) * w[js]
template <int NREG, int NS>
SS_HD void region_sums(const rv* x, const double* w, double (*s)[2]) {
  rv acc[NREG][2];
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) acc[r][0] = acc[r][1] = 0.0;
  static_for<0, NREG * NS>([&](auto kc) {
    constexpr int K = decltype(kc)::value;
    row_fma<K, 2>(acc[K / NS], x, rv(w[K % NS]));
  });
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) {
    s[r][0] = uniform(acc[r][0]);
    s[r][1] = uniform(acc[r][1]);
  }
}
// y[p] (+)= A x[p]
template <int N, bool ACC>
SS_HD void mat_vec2(const rv* A, const rv* x, rv* y) {
  if (!ACC) y[0] = y[1] = 0.0;
  static_for<0, N>([&](auto kc) { constexpr int K = decltype(kc)::value; row_fma<K, 2>(y, x, A[K]); });
}

// Crossing an interface downward (see the header of "cross" in ss_column_sw.hpp): fluxes just
// below it through V, the upward flux just below it from a_below (and the exposed roofs), the
// upward flux just above it through U.
template <bool URBAN, int NREG, int NS>
struct Cross {
  static constexpr int N = NREG * NS;
  static constexpr int DB = URBAN ? NREG + 1 : NREG;
  const RowGeom<NREG, NS>& g;
  double U[NREG * DB], V[DB * NREG];
  SS_HD explicit Cross(const RowGeom<NREG, NS>& g_) : g(g_) {}
  // dn_below (canopy rows dc, roof rows dr on the lanes 0 .. NS-1) = (V (x) I) dn_ab
  SS_HD void down(const rv* dn_ab, rv* dc, rv* dr) const {
    rv vrow[NREG];
    SS_UNROLL_FULL
    for (int ju = 0; ju < NREG; ++ju) {
      rv v = 0.0;
      SS_UNROLL_FULL
      for (int jl = 0; jl < NREG; ++jl) v = select(g.rg[jl], rv(SS_E(V, jl, ju, DB)), v);
      vrow[ju] = v;
    }
    dc[0] = dc[1] = 0.0;
    dr[0] = dr[1] = 0.0;
    static_for<0, N>([&](auto kc) {
      constexpr int K = decltype(kc)::value;
      row_fma<K, 2>(dc, dn_ab, select(g.sm[K % NS], vrow[K / NS], rv(0.0)));
      if (URBAN) row_fma<K, 2>(dr, dn_ab, select(lane_eq(K % NS), rv(SS_E(V, DB - 1, K / NS, DB)), rv(0.0)));
    });
  }
  // up_ab = (U (x) I) [uc; ur]
  SS_HD void up(const rv* uc, const rv* ur, rv* up_ab) const {
    rv urow[DB];
    g.template own_row<DB>(U, urow);
    apply_U<NREG, NS, 2>(g, urow, uc, up_ab);
    if (URBAN) {
      static_for<0, NS>([&](auto kc) {
        constexpr int K = decltype(kc)::value;
        row_fma<K, 2>(up_ab, ur, select(g.sm[K], urow[DB - 1], rv(0.0)));
      });
    }
  }
};

// ---------------------------------------------------------------- longwave
template <bool URBAN, int NREG, int NS>
SS_HD void column_down_lw(const Launch& L, int cu) {
  constexpr int N = NREG * NS;
  using LRec = LwRec<N>;
  using CRec = LwColRec<N>;
  constexpr long st = kRecStride;
  const Inputs& in = L.in;
  const LG& lg = L.lg;
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  const int nspec = L.nspec;
  const int nlay = in.nlay[col];
  const int l0 = in.istartlay[col] - 1;
  const long cidx = (long)jb + (long)nspec * col;
  const RowGeom<NREG, NS> g(lg);
  const rv hw_r = select(lane_lt(NS), g.hw, rv(0.0));      // hweight on the roof lanes
  const double top_emissivity = L.top1[cidx], top_emission = L.top2[cidx];
  double* const* FA = L.flux_a.f;     // lw_internal (pass 0)
  double* const* FB = L.flux_b.f;     // lw_norm     (pass 1)
// (the member pointers of the two flux objects are read from the kernel-argument segment where they are used:
// held from the kernel entry they are 108 scalar registers, spilled lane by lane into VGPRs - SS_LAUNCH_HERE)
#if !defined(SS_DOWN16_POINTERS_AT_ENTRY)
#define PUT2(F, idx, v0, v1) do { if (lane0()) { SS_LAUNCH_HERE(L, Lo_); double* a_ = Lo_.flux_a.f[F]; double* b_ = Lo_.flux_b.f[F]; \
    if (a_) a_[idx] = (v0); if (b_) b_[idx] = (v1); } } while (0)
#else
#define PUT2(F, idx, v0, v1) do { if (lane0()) { if (FA[F]) FA[F][idx] = (v0); if (FB[F]) FB[F][idx] = (v1); } } while (0)
#endif

  rv dn_above[2] = {rv(0.0), select(g.rg[0], g.hw, rv(0.0))}, up_above[2] = {0.0, 0.0};
  PUT2(F_TOP_DN, cidx, 0.0, 1.0);
  PUT2(F_TOP_NET, cidx, -top_emission, top_emissivity);
  double emiss_factor = 0.0;
  SS_UNROLL_FULL
  for (int js = 0; js < NS; ++js) emiss_factor += lg.hweight[js] / lg.mu[js];
  emiss_factor = 2.0 * emiss_factor;
  double rmu[NS];
  SS_UNROLL_FULL
  for (int js = 0; js < NS; ++js) rmu[js] = 1.0 / lg.mu[js];

  Cross<URBAN, NREG, NS> X(g);
  constexpr int DB = Cross<URBAN, NREG, NS>::DB;
  rv dbc[2], dbr[2] = {0.0, 0.0}, ubc[2], ubr[2] = {0.0, 0.0};   // fluxes just below the interface above the current layer
  auto cross = [&](const LwLayerIn<URBAN, NREG>& dlo, bool upper_is_sky, double vf_up, double bf_up,
                   const rv* ABE, const rv& SBE, rv* up_ab) {
    interface_overlap_vals<URBAN, true, NREG>(L.cfg.min_vegetation_fraction, upper_is_sky, vf_up, bf_up,
                                              false, dlo.g.veg_fraction, dlo.g.building_fraction, X.U, X.V);
    X.down(dn_above, dbc, dbr);
    ubc[0] = SBE;
    ubc[1] = 0.0;
    mat_vec2<N, true>(ABE, dbc, ubc);
    if (URBAN) {
      const double roof_a = 1.0 - dlo.roof_emissivity;
      const double exposed = upper_is_sky ? dlo.g.building_fraction : ss_max(0.0, dlo.g.building_fraction - bf_up);
      double sr[2];
      all_sum2<NS>(dbr, sr);
      ubr[0] = (roof_a * hw_r) * sr[0] + hw_r * (dlo.roof_emission * exposed);
      ubr[1] = (roof_a * hw_r) * sr[1];
    }
    if (up_ab) X.up(ubc, ubr, up_ab);
  };

  LwLayerIn<URBAN, NREG> din;
  if (nlay > 0) {
    const double* crec = L.col + rec_base(nlay - 1, L.maxlay, CRec::NELEM, cu);
    rv ABE[N];
    din.load_pack(pack_rec(L, nlay - 1, cu, kPackLw));
    load_colmajor<N, N>(crec, st, CRec::ABE, ABE);
    cross(din, true, 0.0, 0.0, ABE, load_vec<N>(crec, st, CRec::SBE), nullptr);
  }
  const bool profiles = FA[F_FLUX_DN_LAYER_TOP] || FB[F_FLUX_DN_LAYER_TOP];

  for (int jlay = nlay - 1; jlay >= 0; --jlay) {
    const double* lrec = L.lay + rec_base(jlay, L.maxlay, LRec::NELEM, cu);
    const double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    const long sidx = (long)jb + (long)nspec * (l0 + jlay);
    const LwLayerIn<URBAN, NREG> dcur = din;

    if (URBAN) {
      double sdn[2], sup[2];
      all_sum2<NS>(dbr, sdn);
      all_sum2<NS>(ubr, sup);
      PUT2(F_ROOF_IN, sidx, sdn[0], sdn[1]);
      PUT2(F_ROOF_NET, sidx, sdn[0] - sup[0], sdn[1] - sup[1]);
    } else {
      PUT2(F_ROOF_IN, sidx, 0.0, 0.0);
      PUT2(F_ROOF_NET, sidx, 0.0, 0.0);
    }
    rv conv[2] = {dbc[0], dbc[1]};
    double s_top_dn[2] = {0.0, 0.0}, s_top_up[2] = {0.0, 0.0};
    if (profiles) {
      all_sum2<N>(dbc, s_top_dn);
      all_sum2<N>(ubc, s_top_up);
    }
    // flux at the base of the layer (forest :547-553, :633-636): P flux_top + Den^-1 (R source_above + source_lay)
    {
      rv P[N];
      load_colmajor<N, N>(crec, st, CRec::PDN, P);
      dn_above[0] = load_vec<N>(crec, st, CRec::RSD);
      dn_above[1] = 0.0;
      mat_vec2<N, true>(P, dbc, dn_above);
    }
    if (jlay > 0) {
      const double* crec_n = L.col + rec_base(jlay - 1, L.maxlay, CRec::NELEM, cu);
      rv ABE[N];
      din.load_pack(pack_rec(L, jlay - 1, cu, kPackLw));
      load_colmajor<N, N>(crec_n, st, CRec::ABE, ABE);
      cross(din, false, dcur.g.veg_fraction, dcur.g.building_fraction, ABE, load_vec<N>(crec_n, st, CRec::SBE), up_above);
    } else {
      // the ground (forest :464-470, urban :560-567): grey Lambertian emitter, per region
      const double gem = in.ground_emissivity[cidx], gemission = in.ground_emission[cidx];
      double frac1[NREG], ones[NS], sd[NREG][2];
      region_fractions<URBAN, true, NREG>(dcur.g.veg_fraction, dcur.g.building_fraction, frac1);
      SS_UNROLL_FULL
      for (int js = 0; js < NS; ++js) ones[js] = 1.0;
      region_sums<NREG, NS>(dn_above, ones, sd);
      rv fr = 0.0, sd0 = 0.0, sd1 = 0.0;
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) {
        fr = select(g.rg[r], rv(frac1[r]), fr);
        sd0 = select(g.rg[r], rv(sd[r][0]), sd0);
        sd1 = select(g.rg[r], rv(sd[r][1]), sd1);
      }
      up_above[0] = ((1.0 - gem) * g.hw) * sd0 + (g.hw * fr) * gemission;
      up_above[1] = ((1.0 - gem) * g.hw) * sd1;
    }
    if (profiles) {
      double sb_dn[2], sb_up[2];
      all_sum2<N>(dn_above, sb_dn);
      all_sum2<N>(up_above, sb_up);
      PUT2(F_FLUX_DN_LAYER_TOP, sidx, s_top_dn[0], s_top_dn[1]);
      PUT2(F_FLUX_UP_LAYER_TOP, sidx, s_top_up[0], s_top_up[1]);
      PUT2(F_FLUX_DN_LAYER_BASE, sidx, sb_dn[0], sb_dn[1]);
      PUT2(F_FLUX_UP_LAYER_BASE, sidx, sb_up[0], sb_up[1]);
    }
    conv[0] = conv[0] + up_above[0];
    conv[1] = conv[1] + up_above[1];
    rv ifl[2];
    {
      rv IFLUX[N];
      load_colmajor<N, N>(lrec, st, LRec::IFLUX, IFLUX);
      ifl[0] = load_vec<N>(lrec, st, LRec::ISRC);
      ifl[1] = 0.0;
      mat_vec2<N, true>(IFLUX, conv, ifl);
    }
    // absorption (forest :571-590 / :658-672, urban :700-737 / :811-835)
    LwLayerProps<URBAN, NREG> p;
    lw_layer_props<URBAN, NREG>(L, dcur, p);
    const double air_ext = dcur.air_ext, air_ssa = dcur.air_ssa;
    double sa[NREG][2];
    region_sums<NREG, NS>(ifl, rmu, sa);
    {
      const double volume_emiss = p.frac[0] * (p.ext_reg[0] * (1.0 - p.ssa_reg[0]) * p.planck_reg[0]);
      PUT2(F_CLEAR_AIR_ABS, sidx, 0.0 + air_ext * (1.0 - air_ssa) * sa[0][0] - ((emiss_factor * volume_emiss) * p.g.dz),
           0.0 + air_ext * (1.0 - air_ssa) * sa[0][1]);
    }
    if (NREG > 1) {
      double vaa[2] = {0.0, 0.0}, va[2] = {0.0, 0.0};
      SS_UNROLL_FULL
      for (int jreg = 1; jreg < NREG; ++jreg) {
        const double emiss_air = emiss_factor * p.frac[jreg] * p.ext_reg[0] * (1.0 - p.ssa_reg[0]) * p.vaplanck;
        const double emiss_veg = emiss_factor * p.frac[jreg] * p.g.veg_ext * (1.0 - p.vssa) * p.vplanck * p.od[jreg];
        vaa[0] = vaa[0] + air_ext * (1.0 - air_ssa) * sa[jreg][0] - (emiss_air * p.g.dz);
        vaa[1] = vaa[1] + air_ext * (1.0 - air_ssa) * sa[jreg][1];
        va[0] = va[0] + p.g.veg_ext * (1.0 - p.vssa) * sa[jreg][0] * p.od[jreg] - (emiss_veg * p.g.dz);
        va[1] = va[1] + p.g.veg_ext * (1.0 - p.vssa) * sa[jreg][1] * p.od[jreg];
      }
      PUT2(F_VEG_AIR_ABS, sidx, vaa[0], vaa[1]);
      PUT2(F_VEG_ABS, sidx, va[0], va[1]);
    } else {
      PUT2(F_VEG_AIR_ABS, sidx, 0.0, 0.0);
      PUT2(F_VEG_ABS, sidx, 0.0, 0.0);
    }
    if (URBAN) {
      double st_[NREG][2], wall_in[2] = {0.0, 0.0}, npw_sum = 0.0;
      region_sums<NREG, NS>(ifl, lg.tan_ang, st_);
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg) {
        wall_in[0] = wall_in[0] + p.f_wall[jreg] * st_[jreg][0];
        wall_in[1] = wall_in[1] + p.f_wall[jreg] * st_[jreg][1];
        npw_sum += p.npw[jreg];
      }
      const double emiss_wall = (npw_sum * lg.vadjustment) * dcur.wall_emission;
      PUT2(F_WALL_IN, sidx, wall_in[0], wall_in[1]);
      PUT2(F_WALL_NET, sidx, wall_in[0] * dcur.wall_emissivity - (emiss_wall * p.g.dz), wall_in[1] * dcur.wall_emissivity);
    } else {
      PUT2(F_WALL_IN, sidx, 0.0, 0.0);
      PUT2(F_WALL_NET, sidx, 0.0, 0.0);
    }
  }

  // ground (forest :599-610 / :680-694, urban :741-752 / :845-858)
  {
    double s2[2], s3[2], gv[NREG][2], gvd[2] = {0.0, 0.0};
    all_sum2<N>(dn_above, s2);
    all_sum2<N>(up_above, s3);
    PUT2(F_GROUND_DN, cidx, s2[0], s2[1]);
    PUT2(F_GROUND_NET, cidx, s2[0] - s3[0], s2[1] - s3[1]);
    double tpi[NS];
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) tpi[js] = lg.tan_ang[js] / kPi;
    const rv du[2] = {dn_above[0] + up_above[0], dn_above[1] + up_above[1]};
    region_sums<NREG, NS>(du, tpi, gv);
    SS_UNROLL_FULL
    for (int jreg = 0; jreg < NREG; ++jreg) {
      gvd[0] += gv[jreg][0];
      gvd[1] += gv[jreg][1];
    }
    if (URBAN) {
      PUT2(F_GROUND_VERTICAL_DIFF, cidx, gvd[0], gvd[1]);
    } else {
      // forest: the incoming pass accumulates on top of the internal pass's value, into lw_internal,
      // and lw_norm%ground_vertical_diff stays zero (radsurf_forest_lw.F90:691)
      PUT2(F_GROUND_VERTICAL_DIFF, cidx, gvd[0] + gvd[1], 0.0);
    }
  }
#undef PUT2
}

// ---------------------------------------------------------------- shortwave
// Own row of Gamma1 + Gamma2 (B) of one layer and the own non-zero element of Gamma3, as
// sw_int_flux_system builds them (ss_solver_sw.hpp), with the rows and columns outside the layer's
// sub-problem replaced by the identity; inc_i / inc_r: which rows / regions belong to it.
template <bool URBAN, int NREG, int NS>
SS_HD void sw_int_flux_rows(const Launch& L, const RowGeom<NREG, NS>& g, const LayerGeom& geo, double wall_albedo,
                            double wall_specular_frac, double air_ext, double air_ssa, double veg_ssa, double sin0,
                            rv* B, rv& g3own, rm& inc_i, bool* inc_r) {
  constexpr int N = NREG * NS;
  const LG& lg = L.lg;
  double frac[NREG], np[NREG], npw[NREG], fx[NREG * NREG], od[NREG], f_wall[NREG];
  region_fractions<URBAN, false, NREG>(geo.veg_fraction, geo.building_fraction, frac);
  norm_perimeters<URBAN, NREG>(L.cfg, geo, np, npw);
  exchange_rates<NREG>(frac, np, L.cfg.min_vegetation_fraction, fx);
  od_scaling<NREG>(geo.veg_fsd, od);
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
  region_optics<NREG>(air_ext, air_ssa, geo.veg_ext, veg_ssa, od, ext_reg, ssa_reg);
  // own rows of the exchange / extinction part and of Gamma2, own non-zero element of Gamma3
  GammaScalars<NREG> q;
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) { q.ext[r] = ext_reg[r]; q.ssa[r] = ssa_reg[r]; q.fw[r] = f_wall[r]; }
  SS_UNROLL_FULL
  for (int k = 0; k < NREG * NREG; ++k) q.fx[k] = fx[k];
  q.wall_ext = wall_ext;
  q.wall_factor = wall_factor;
  rv E[N], G2[N];
  gamma_rows<URBAN, NREG, NS, false>(lg, g, q, E, G2);
  {
    rv w_i = 0.0, vw_i = 0.0, es_i = 0.0, fw_i = 0.0;
    SS_UNROLL_FULL
    for (int s = 0; s < NS; ++s) {
      w_i = select(g.sm[s], rv(lg.weight[s]), w_i);
      vw_i = select(g.sm[s], rv(lg.vweight[s]), vw_i);
    }
    rv ext_i = 0.0, ssa_i = 0.0;
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) {
      ext_i = select(g.rg[r], rv(ext_reg[r]), ext_i);
      ssa_i = select(g.rg[r], rv(ssa_reg[r]), ssa_i);
      fw_i = select(g.rg[r], rv(f_wall[r]), fw_i);
    }
    (void)es_i;
    if (URBAN) g3own = 0.5 * (w_i * ext_i * ssa_i + vw_i * sin0 * fw_i * wall_factor);
    else g3own = 0.5 * w_i * ext_i * ssa_i;
  }
  // the sub-problem of this layer (the 3-way branch of layer_unit_sw)
  const bool clear_only = (NREG == 1) || (geo.veg_fraction <= L.cfg.min_vegetation_fraction);
  const bool veg_only = !clear_only && (frac[0] <= L.cfg.min_vegetation_fraction);
  inc_i = gen_mask([=](int i) { return i < N && (clear_only ? (i < NS) : (veg_only ? (i >= NS) : true)); });
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) inc_r[r] = clear_only ? (r == 0) : (veg_only ? (r >= 1) : true);
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    const int rj = j / NS;
    // Gamma1 (as used) = exchange / extinction part + Gamma2, so Gamma1 + Gamma2 adds it twice
    const rv v = (E[j] + G2[j]) + G2[j];
    const rv ident = select(lane_eq(j), rv(1.0), rv(0.0));
    B[j] = inc_r[rj] ? select(inc_i, v, ident) : ident;
  }
}

template <bool URBAN, int NREG, int NS>
SS_HD void column_down_sw(const Launch& L, int cu) {
  constexpr int N = NREG * NS, D = NREG;
  using LRec = SwRec<N, D>;
  using CRec = SwColRec<N, D>;
  constexpr long st = kRecStride;
  const Inputs& in = L.in;
  const LG& lg = L.lg;
  const SwColCtx<URBAN, NREG, NS> c(L, cu, true);
  const int col = c.col, jb = c.jb, nspec = c.nspec, nlay = c.nlay, l0 = c.l0, itransp = c.itransp;
  const long cidx = c.cidx;
  const double cos_sza = c.cos_sza, sin0 = c.sin0;
  const RowGeom<NREG, NS> g(lg);
  const rv hw_r = select(lane_lt(NS), g.hw, rv(0.0));
  double* const* FA = L.flux_a.f;     // sw_norm_dir  (pass 0)
  double* const* FB = L.flux_b.f;     // sw_norm_diff (pass 1)
// (the member pointers of the two flux objects are read from the kernel-argument segment where they are used:
// held from the kernel entry they are 108 scalar registers, spilled lane by lane into VGPRs - SS_LAUNCH_HERE)
#if !defined(SS_DOWN16_POINTERS_AT_ENTRY)
#define PUT2(F, idx, v0, v1) do { if (lane0()) { SS_LAUNCH_HERE(L, Lo_); double* a_ = Lo_.flux_a.f[F]; double* b_ = Lo_.flux_b.f[F]; \
    if (a_) a_[idx] = (v0); if (b_) b_[idx] = (v1); } } while (0)
#else
#define PUT2(F, idx, v0, v1) do { if (lane0()) { if (FA[F]) FA[F][idx] = (v0); if (FB[F]) FB[F][idx] = (v1); } } while (0)
#endif

  double dir_above[D];
  dir_above[0] = 1.0 / cos_sza;
  SS_UNROLL_FULL
  for (int r = 1; r < D; ++r) dir_above[r] = 0.0;
  rv diff_above[2] = {rv(0.0), select(g.rg[0], g.hw, rv(0.0))}, up_above[2] = {0.0, 0.0};
  PUT2(F_TOP_DN_DIR, cidx, 1.0, 0.0);
  PUT2(F_TOP_DN, cidx, 1.0, 1.0);
  PUT2(F_TOP_NET, cidx, 1.0 * (1.0 - L.top2[cidx]), 1.0 - L.top1[cidx]);
  double flux_dn_dir_clear = 1.0 / cos_sza;   // clear-sky direct beam of the direct pass
  double rmu[NS];
  SS_UNROLL_FULL
  for (int js = 0; js < NS; ++js) rmu[js] = 1.0 / lg.mu[js];

  Cross<URBAN, NREG, NS> X(g);
  constexpr int DB = Cross<URBAN, NREG, NS>::DB;
  double dir_below[DB];
  rv dbc[2], dbr[2] = {0.0, 0.0}, ubc[2], ubr[2] = {0.0, 0.0};
  // lo = veg_fraction, building_fraction, roof_albedo, roof_albedo_dir of the lower layer
  auto cross = [&](const double* lo, bool upper_is_sky, double vf_up, double bf_up, const rv* ABE, const rv* DBE,
                   rv* up_ab) {
    interface_overlap_vals<URBAN, false, NREG>(L.cfg.min_vegetation_fraction, upper_is_sky, vf_up, bf_up,
                                               false, lo[0], lo[1], X.U, X.V);
    SS_UNROLL_FULL
    for (int jl = 0; jl < DB; ++jl) {
      double acc = 0.0;
      SS_UNROLL_FULL
      for (int ju = 0; ju < NREG; ++ju) acc += SS_E(X.V, jl, ju, DB) * dir_above[ju];
      dir_below[jl] = acc;
    }
    X.down(diff_above, dbc, dbr);
    mat_vec2<N, false>(ABE, dbc, ubc);
    SS_UNROLL_FULL
    for (int r = 0; r < D; ++r) ubc[0] = ubc[0] + DBE[r] * dir_below[r];
    if (URBAN) {
      const double roof_a = lo[2], roof_d = cos_sza * lo[3];
      double sr[2];
      all_sum2<NS>(dbr, sr);
      ubr[0] = (roof_a * hw_r) * sr[0] + (roof_d * hw_r) * dir_below[DB - 1];
      ubr[1] = (roof_a * hw_r) * sr[1];
    }
    if (up_ab) X.up(ubc, ubr, up_ab);
  };

  SwDownIn<URBAN, NREG> din;
  double bf_up = 0.0;          // building fraction of the layer above (roof / sunlit bookkeeping)
  if (nlay > 0) {
    const double* crec = L.col + rec_base(nlay - 1, L.maxlay, CRec::NELEM, cu);
    rv ABE[N], DBE[D];
    din.load_pack(pack_rec(L, nlay - 1, cu, kPackSw));
    load_colmajor<N, N>(crec, st, CRec::ABE, ABE);
    load_colmajor<N, D>(crec, st, CRec::DBE, DBE);
    const double lo[4] = {din.g.veg_fraction, din.g.building_fraction, din.roof_albedo, din.roof_albedo_dir};
    cross(lo, true, 0.0, 0.0, ABE, DBE, nullptr);
  }
  const bool profiles = FA[F_FLUX_DN_LAYER_TOP] || FB[F_FLUX_DN_LAYER_TOP];

  for (int jlay = nlay - 1; jlay >= 0; --jlay) {
    const double* lrec = L.lay + rec_base(jlay, L.maxlay, LRec::NELEM, cu);
    const double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    const int lay = l0 + jlay;
    const long sidx = (long)jb + (long)nspec * lay;
    const SwDownIn<URBAN, NREG> dcur = din;
    const LayerGeom geo = dcur.g;

    if (URBAN) {
      double sdn[2], sup[2];
      all_sum2<NS>(dbr, sdn);
      all_sum2<NS>(ubr, sup);
      const double roof_in_dir = cos_sza * dir_below[DB - 1];
      PUT2(F_ROOF_IN_DIR, sidx, roof_in_dir, 0.0);
      PUT2(F_ROOF_IN, sidx, roof_in_dir + sdn[0], 0.0 + sdn[1]);
      PUT2(F_ROOF_NET, sidx, (roof_in_dir + sdn[0]) - sup[0], (0.0 + sdn[1]) - sup[1]);
      if (jb == itransp) {
        // urban :805-812
        const double nbf_up = (jlay + 1 < nlay) ? 1.0 - bf_up : 1.0;
        const double nbf = 1.0 - geo.building_fraction;
        double roof_fraction;
        if (jlay == nlay - 1) roof_fraction = geo.building_fraction;
        else roof_fraction = ss_max(0.0, geo.building_fraction - bf_up);
        const double rsf = roof_in_dir * nbf_up
            / (cos_sza * flux_dn_dir_clear * ss_max(L.cfg.min_building_fraction, roof_fraction));
        PUT2(F_ROOF_SUNLIT_FRAC, lay, rsf, 0.0);
        flux_dn_dir_clear = flux_dn_dir_clear * nbf / nbf_up;
      }
    }
    // direct beam through the layer, and its contribution to the diffuse flux at the base
    double dir_conv[D];
    rv t3;
    // (-DSS_DOWN16_PIPE requests the record loads of a stage one stage ahead: P before the direct-beam stage,
    // a_below / d_below of the layer below before the product with P, int_dir before the interface.  Measured
    // and OFF: the kernel is at its 256 registers, 22 -> 53 spilled dwords, 4.34 -> 5.55 ms per step of C3 with
    // 4 streams.)
    rv P[N];
    {
      double E[D * D];
      SS_UNROLL_FULL
      for (int k = 0; k < D * D; ++k) E[k] = rec_elem_load(lrec, st, LRec::EDIR + k);
      rv Q[D];
      load_colmajor<N, D>(crec, st, CRec::QDN, Q);
#if defined(SS_DOWN16_PIPE)
      SS_LOAD_FENCE(); SS_SCHED_FENCE();
      load_colmajor<N, N>(crec, st, CRec::PDN, P);
      SS_LOAD_FENCE(); SS_SCHED_FENCE();
#endif
      t3 = 0.0;
      SS_UNROLL_FULL
      for (int r = 0; r < D; ++r) t3 = t3 + Q[r] * dir_below[r];
      SS_UNROLL_FULL
      for (int r = 0; r < D; ++r) {
        double t = 0.0;
        SS_UNROLL_FULL
        for (int k = 0; k < D; ++k) t += SS_E(E, r, k, D) * dir_below[k];
        dir_above[r] = t;
        dir_conv[r] = dir_below[r] - t;
      }
    }
    rv conv[2] = {dbc[0] - ubc[0], dbc[1] - ubc[1]};
    double s_top_dir = 0.0, s_top_dn[2] = {0.0, 0.0}, s_top_up[2] = {0.0, 0.0};
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) s_top_dir += dir_below[r];
    if (profiles) {
      all_sum2<N>(dbc, s_top_dn);
      all_sum2<N>(ubc, s_top_up);
    }
    // flux at the base of the layer: diff = P diff_top + Q dir_top
    rv ABE[N], DBE[D];
    double lo[4] = {0.0, 0.0, 0.0, 0.0};
    double IDIR[D * D];
#if defined(SS_DOWN16_PIPE)
    if (jlay > 0) {
      const double* crec_n = L.col + rec_base(jlay - 1, L.maxlay, CRec::NELEM, cu);
      const double* pn = pack_rec(L, jlay - 1, cu, kPackSw);
      pack_fractions(pn, lo[0], lo[1]);
      const RecPair ra = rec_pair_load(pn + 8 * kRecStride);   // roof_albedo, roof_albedo_dir
      lo[2] = ra.x;
      lo[3] = ra.y;
      load_colmajor<N, N>(crec_n, st, CRec::ABE, ABE);
      load_colmajor<N, D>(crec_n, st, CRec::DBE, DBE);
    }
    SS_LOAD_FENCE(); SS_SCHED_FENCE();
#endif
    {
#if !defined(SS_DOWN16_PIPE)
      load_colmajor<N, N>(crec, st, CRec::PDN, P);
#endif
      diff_above[0] = t3;
      diff_above[1] = 0.0;
      mat_vec2<N, true>(P, dbc, diff_above);
    }
#if defined(SS_DOWN16_PIPE)
    SS_SCHED_FENCE();
    SS_UNROLL_FULL
    for (int k = 0; k < D * D; ++k) IDIR[k] = rec_elem_load(lrec, st, LRec::IDIR + k);
    SS_LOAD_FENCE(); SS_SCHED_FENCE();
#endif
    if (jlay > 0) {
#if !defined(SS_DOWN16_PIPE)
      const double* crec_n = L.col + rec_base(jlay - 1, L.maxlay, CRec::NELEM, cu);
      const double* pn = pack_rec(L, jlay - 1, cu, kPackSw);
      pack_fractions(pn, lo[0], lo[1]);
      const RecPair ra = rec_pair_load(pn + 8 * kRecStride);   // roof_albedo, roof_albedo_dir
      lo[2] = ra.x;
      lo[3] = ra.y;
      load_colmajor<N, N>(crec_n, st, CRec::ABE, ABE);
      load_colmajor<N, D>(crec_n, st, CRec::DBE, DBE);
#endif
      cross(lo, false, geo.veg_fraction, geo.building_fraction, ABE, DBE, up_above);
    } else {
      // the ground (forest :479-484, urban :612-617): Lambertian, per region
      const double ga = in.ground_albedo[cidx], gad = in.ground_albedo_dir[cidx];
      double ones[NS], sd[NREG][2];
      SS_UNROLL_FULL
      for (int js = 0; js < NS; ++js) ones[js] = 1.0;
      region_sums<NREG, NS>(diff_above, ones, sd);
      rv sd0 = 0.0, sd1 = 0.0, da = 0.0;
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) {
        sd0 = select(g.rg[r], rv(sd[r][0]), sd0);
        sd1 = select(g.rg[r], rv(sd[r][1]), sd1);
        da = select(g.rg[r], rv(dir_above[r]), da);
      }
      up_above[0] = (ga * g.hw) * sd0 + ((cos_sza * gad) * g.hw) * da;
      up_above[1] = (ga * g.hw) * sd1;
    }
    if (profiles) {
      double sb_dn[2], sb_up[2], s = 0.0;
      all_sum2<N>(diff_above, sb_dn);
      all_sum2<N>(up_above, sb_up);
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) s += dir_above[r];
      const double ddt = cos_sza * s_top_dir, ddb = cos_sza * s;
      PUT2(F_FLUX_DN_DIR_LAYER_TOP, sidx, ddt, 0.0);
      PUT2(F_FLUX_DN_LAYER_TOP, sidx, ddt + s_top_dn[0], 0.0 + s_top_dn[1]);
      PUT2(F_FLUX_UP_LAYER_TOP, sidx, s_top_up[0], s_top_up[1]);
      PUT2(F_FLUX_DN_DIR_LAYER_BASE, sidx, ddb, 0.0);
      PUT2(F_FLUX_DN_LAYER_BASE, sidx, ddb + sb_dn[0], 0.0 + sb_dn[1]);
      PUT2(F_FLUX_UP_LAYER_BASE, sidx, sb_up[0], sb_up[1]);
    }
    double int_flux_dir[D];
    {
#if !defined(SS_DOWN16_PIPE)
      SS_UNROLL_FULL
      for (int k = 0; k < D * D; ++k) IDIR[k] = rec_elem_load(lrec, st, LRec::IDIR + k);
#endif
      SS_UNROLL_FULL
      for (int r = 0; r < D; ++r) {
        double t = 0.0;
        SS_UNROLL_FULL
        for (int k = 0; k < D; ++k) t += SS_E(IDIR, r, k, D) * dir_conv[k];
        int_flux_dir[r] = t;
      }
    }
    conv[0] = conv[0] - diff_above[0] + up_above[0];
    conv[1] = conv[1] - diff_above[1] + up_above[1];
    if (jlay > 0) din.load_pack(pack_rec(L, jlay - 1, cu, kPackSw));
    bf_up = geo.building_fraction;

    // integrated fluxes (forest :598-606): int_diff conv + int_dir_diff dir_conv
    //   = -(Gamma1 + Gamma2)^-1 (conv + 2 Gamma3 int_flux_dir), see sw_int_flux_system
    rv ifd[2];
    {
      LU<N> B;
      rv g3own;
      rm inc_i;
      bool inc_r[NREG];
      sw_int_flux_rows<URBAN, NREG, NS>(L, g, geo, dcur.wall_albedo, dcur.wall_specular_frac, dcur.air_ext,
                                        dcur.air_ssa, dcur.veg_ssa, sin0, B.a, g3own, inc_i, inc_r);
      rv ifd_own = 0.0;       // int_flux_dir of the own region (zero outside the sub-problem)
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) ifd_own = select(g.rg[r], rv(inc_r[r] ? int_flux_dir[r] : 0.0), ifd_own);
      ifd[0] = select(inc_i, -(conv[0] + 2.0 * g3own * ifd_own), rv(0.0));
      ifd[1] = select(inc_i, -conv[1], rv(0.0));
      lu_factor<N>(B);
      lu_solve<N, 2>(B, ifd);
    }

    // absorption (forest :608-628, urban :758-788)
    const double air_ext = dcur.air_ext, air_ssa = dcur.air_ssa;
    double sa[NREG][2];
    region_sums<NREG, NS>(ifd, rmu, sa);
    PUT2(F_CLEAR_AIR_ABS, sidx, 0.0 + air_ext * (1.0 - air_ssa) * (int_flux_dir[0] + sa[0][0]),
         0.0 + air_ext * (1.0 - air_ssa) * (0.0 + sa[0][1]));
    double veg_abs_dir = 0.0;
    if (NREG > 1) {
      double od[NREG];
      od_scaling<NREG>(geo.veg_fsd, od);
      const double vssa = dcur.veg_ssa;
      double vaa[2] = {0.0, 0.0}, va[2] = {0.0, 0.0};
      SS_UNROLL_FULL
      for (int jreg = 1; jreg < NREG; ++jreg) {
        vaa[0] = vaa[0] + air_ext * (1.0 - air_ssa) * (int_flux_dir[jreg] + sa[jreg][0]);
        vaa[1] = vaa[1] + air_ext * (1.0 - air_ssa) * (0.0 + sa[jreg][1]);
        veg_abs_dir = veg_abs_dir + geo.veg_ext * (1.0 - vssa) * int_flux_dir[jreg] * od[jreg];
        va[0] = va[0] + geo.veg_ext * (1.0 - vssa) * (int_flux_dir[jreg] + sa[jreg][0]) * od[jreg];
        va[1] = va[1] + geo.veg_ext * (1.0 - vssa) * (0.0 + sa[jreg][1]) * od[jreg];
      }
      PUT2(F_VEG_AIR_ABS, sidx, vaa[0], vaa[1]);
      PUT2(F_VEG_ABS_DIR, sidx, veg_abs_dir, 0.0);
      PUT2(F_VEG_ABS, sidx, va[0], va[1]);
    } else {
      PUT2(F_VEG_AIR_ABS, sidx, 0.0, 0.0);
      PUT2(F_VEG_ABS_DIR, sidx, 0.0, 0.0);
      PUT2(F_VEG_ABS, sidx, 0.0, 0.0);
    }

    double wall_in_dir = 0.0, npw_sum = 0.0;
    if (URBAN) {
      // walls (urban :790-802)
      double frac[NREG], np[NREG], npw[NREG], f_wall[NREG], stn[NREG][2];
      region_fractions<URBAN, false, NREG>(geo.veg_fraction, geo.building_fraction, frac);
      norm_perimeters<URBAN, NREG>(L.cfg, geo, np, npw);
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) {
        f_wall[r] = (frac[r] <= L.cfg.min_vegetation_fraction) ? 0.0 : npw[r] / (kPi * frac[r]);
        npw_sum += npw[r];
      }
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg)
        wall_in_dir = wall_in_dir + f_wall[jreg] * sin0 * int_flux_dir[jreg];
      region_sums<NREG, NS>(ifd, lg.tan_ang, stn);
      double wall_in[2] = {wall_in_dir, 0.0};
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg) {
        wall_in[0] = wall_in[0] + f_wall[jreg] * stn[jreg][0];
        wall_in[1] = wall_in[1] + f_wall[jreg] * stn[jreg][1];
      }
      PUT2(F_WALL_IN_DIR, sidx, wall_in_dir, 0.0);
      PUT2(F_WALL_IN, sidx, wall_in[0], wall_in[1]);
      PUT2(F_WALL_NET, sidx, wall_in[0] * (1.0 - dcur.wall_albedo), wall_in[1] * (1.0 - dcur.wall_albedo));
    } else {
      // members the reference leaves at zero after zero() for forest tiles
      PUT2(F_ROOF_IN, sidx, 0.0, 0.0); PUT2(F_ROOF_IN_DIR, sidx, 0.0, 0.0); PUT2(F_ROOF_NET, sidx, 0.0, 0.0);
      PUT2(F_WALL_IN, sidx, 0.0, 0.0); PUT2(F_WALL_IN_DIR, sidx, 0.0, 0.0); PUT2(F_WALL_NET, sidx, 0.0, 0.0);
    }

    // sunlit fractions and the clear-sky direct beam (forest :630-649, urban :405-411, :814-852);
    // band itransp only, zero in the diffuse pass
    if (jb == itransp) {
      const double ae = dcur.air_ext_t;
      const double trans_dir_clear = exp(-ae * geo.dz / cos_sza);
      const double ifdc = (ae > 0.0) ? flux_dn_dir_clear * (1.0 - trans_dir_clear) * cos_sza / ae
                                     : flux_dn_dir_clear * geo.dz;
      if (URBAN) {
        const double nbf = 1.0 - geo.building_fraction;
        const double f_wall_dir_clear =
            (nbf <= L.cfg.min_building_fraction) ? 0.0 : npw_sum / (kPi * nbf);
        const double wsf = 0.5 * wall_in_dir / ss_max(DBL_EPSILON, (f_wall_dir_clear * sin0 * ifdc));
        PUT2(F_WALL_SUNLIT_FRAC, lay, wsf, 0.0);
      } else {
        PUT2(F_ROOF_SUNLIT_FRAC, lay, 0.0, 0.0);
        PUT2(F_WALL_SUNLIT_FRAC, lay, 0.0, 0.0);
      }
      double vsf = 0.0;
      if (NREG > 1) {
        const double vadc = ifdc * geo.veg_ext * (1.0 - dcur.veg_ssa_t) * geo.veg_fraction;
        vsf = veg_abs_dir / ss_max(DBL_EPSILON, vadc);
      }
      PUT2(F_VEG_SUNLIT_FRAC, lay, vsf, 0.0);
      flux_dn_dir_clear = flux_dn_dir_clear * trans_dir_clear;
    }
  }

  // ground (forest :661-676, urban :861-874)
  {
    double s = 0.0, s2[2], s3[2], gv[NREG][2], gvd[2] = {0.0, 0.0}, tpi[NS];
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) s += dir_above[r];
    const double gdd = cos_sza * s;
    all_sum2<N>(diff_above, s2);
    all_sum2<N>(up_above, s3);
    PUT2(F_GROUND_DN_DIR, cidx, gdd, 0.0);
    PUT2(F_GROUND_DN, cidx, gdd + s2[0], 0.0 + s2[1]);
    PUT2(F_GROUND_NET, cidx, (gdd + s2[0]) - s3[0], (0.0 + s2[1]) - s3[1]);
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) tpi[js] = lg.tan_ang[js] / kPi;
    const rv du[2] = {diff_above[0] + up_above[0], diff_above[1] + up_above[1]};
    region_sums<NREG, NS>(du, tpi, gv);
    SS_UNROLL_FULL
    for (int jreg = 0; jreg < NREG; ++jreg) {
      gvd[0] += gv[jreg][0];
      gvd[1] += gv[jreg][1];
    }
    PUT2(F_GROUND_VERTICAL_DIFF, cidx, gvd[0], gvd[1]);
    if (jb == itransp && lane0()) {
      if (FA[F_GROUND_SUNLIT_FRAC]) FA[F_GROUND_SUNLIT_FRAC][col] = gdd / (cos_sza * flux_dn_dir_clear);
      if (FB[F_GROUND_SUNLIT_FRAC]) FB[F_GROUND_SUNLIT_FRAC][col] = 0.0;
    }
  }
#undef PUT2
}

}  // namespace r16
}  // namespace ss
