// ss_column16.hpp - the adding method (upward sweep) of the orders 7..16 with one column unit on the
// 16 lanes of a DPP row (ss_row16.hpp).  Same steps, same records as column_up_sw / column_up_lw
// (ss_column_sw.hpp, ss_column_lw.hpp; reference radsurf_forest_sw.F90:463-528, radsurf_urban_sw.F90:591-674,
// radsurf_forest_lw.F90:452-506, radsurf_urban_lw.F90:551-626) with the albedo matrices of the column in
// registers: row i of a_above, R, T, Den ... on lane i, vectors one element per lane.
// The overlap operators: right-multiplication by (V (x) I) mixes columns and is lane-local;
// left-multiplication by (U (x) I) mixes rows and is a product with the sparse N x N matrix U (x) I,
// whose own row a lane builds from the NREG x DB matrix U.
#pragma once

#include "ss_common.hpp"
#include "ss_column_lw.hpp"
#include "ss_column_sw.hpp"
#include "ss_layer16.hpp"
#include "ss_row16.hpp"

namespace ss {
namespace r16 {

// A (N x M) = (U (x) I) B for the canopy rows of B (N x M): urow = own row of U
template <int NREG, int NS, int M>
SS_HD void apply_U(const RowGeom<NREG, NS>& g, const rv* urow, const rv* B, rv* A) {
  SS_UNROLL_FULL
  for (int j = 0; j < M; ++j) A[j] = 0.0;
  static_for<0, NREG * NS>([&](auto kc) {
    constexpr int K = decltype(kc)::value;
    row_fma<K, M>(A, B, select(g.sm[K % NS], urow[K / NS], rv(0.0)));
  });
}
// tA (N x N) = A (V (x) I): column (ju, s) = sum_jl A(:, (jl, s)) V(jl, ju)
template <int NREG, int NS, int DB>
SS_HD void apply_V_right(const rv* A, const double* V, rv* tA) {
  SS_UNROLL_FULL
  for (int ju = 0; ju < NREG; ++ju) {
    SS_UNROLL_FULL
    for (int s = 0; s < NS; ++s) {
      rv acc = 0.0;
      SS_UNROLL_FULL
      for (int jl = 0; jl < NREG; ++jl) acc = acc + A[jl * NS + s] * SS_E(V, jl, ju, DB);
      tA[ju * NS + s] = acc;
    }
  }
}

// ---------------------------------------------------------------- longwave
template <bool URBAN, int NREG, int NS>
SS_HD void column_up_lw(const Launch& L, int cu) {
  constexpr int N = NREG * NS;
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  using LRec = LwRec<N>;
  using CRec = LwColRec<N>;
  constexpr long st = kRecStride;
  const Inputs& in = L.in;
  const LG& lg = L.lg;
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  const int nlay = in.nlay[col];
  const int l0 = in.istartlay[col] - 1;
  const long cidx = (long)jb + (long)L.nspec * col;
  const RowGeom<NREG, NS> g(lg);

  // a_above, source_above at the ground (forest :455-466, urban :554-565)
  rv Aab[N], Sab;
  {
    const double gem = in.ground_emissivity[cidx], gemission = in.ground_emission[cidx];
    double frac1[NREG];
    region_fractions<URBAN, true, NREG>((NREG > 1) ? in.veg_fraction[l0] : 0.0,
                                        URBAN ? in.building_fraction[l0] : 0.0, frac1);
    rv fr = 0.0;
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) fr = select(g.rg[r], rv(frac1[r]), fr);
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) Aab[j] = select(g.rg[j / NS], (1.0 - gem) * g.hw, rv(0.0));
    Sab = (g.hw * fr) * gemission;
  }
  // Software pipeline of the record loads (see sym::sw_adding_step, ss_column_sw_sym.hpp): R, T and the source of
  // a layer are requested by the previous step - after its last use of T, ahead of its overlap stage -, the
  // fractions and roof properties of the overlap stage at the top of the step.  Loads complete in the order
  // of issue, so every wait only covers what was requested before it.  Measured (C3 with 4 streams, ms per
  // step on one stream): shortwave sweep 5.16 -> 3.83, longwave sweep 3.33 -> 3.47 - the longwave step has a
  // single load point and two waves per SIMD cover it; so the shortwave sweep is pipelined
  // (-DSS_NO_COLUMN16_PREFETCH: off) and the longwave sweep is not (-DSS_COLUMN16_PREFETCH_LW: on).
  rv R[N], T[N], Sl, Rn[N], Sn;
#if defined(SS_COLUMN16_PREFETCH_LW)
  if (nlay > 0) {
    const double* lrec = L.lay + rec_base(0, L.maxlay, LRec::NELEM, cu);
    load_colmajor<N, N>(lrec, st, LRec::R, R);
    load_colmajor<N, N>(lrec, st, LRec::T, T);
    Sl = load_vec<N>(lrec, st, LRec::SRC);
  }
#endif
  for (int jlay = 0; jlay < nlay; ++jlay) {
    const double* lrec = L.lay + rec_base(jlay, L.maxlay, LRec::NELEM, cu);
    double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    // (R turns into a_below in place and T into P: five matrices live at the peak)
    rv Sbe;
    const double* prec = pack_rec(L, jlay, cu, kPackLw);
    double vf_lo, bf_lo, vf_up = 0.0, bf_up = 0.0;
    RecPair re{0.0, 0.0};
#if !defined(SS_COLUMN16_PREFETCH_LW)
    load_colmajor<N, N>(lrec, st, LRec::R, R);
    load_colmajor<N, N>(lrec, st, LRec::T, T);
    Sl = load_vec<N>(lrec, st, LRec::SRC);
#else
    pack_fractions(prec, vf_lo, bf_lo);
    if (jlay + 1 < nlay) pack_fractions(pack_rec(L, jlay + 1, cu, kPackLw), vf_up, bf_up);
    if (URBAN) re = rec_pair_load(prec + 14 * kRecStride);   // roof_emissivity, roof_emission
    SS_LOAD_FENCE(); SS_SCHED_FENCE();
#endif
    {
      // denominator = I - a_above R
      LU<N> Den;
      mul<N, N, false, false>(Aab, nullptr, R, Den.a);
      SS_UNROLL_FULL
      for (int j = 0; j < N; ++j) Den.a[j] = select(lane_eq(j), rv(1.0), rv(0.0)) - Den.a[j];
      lu_factor<N>(Den);
      // Den^-1 (source_above + a_above source_lay) and Den^-1 (R source_above + source_lay)
      rv v[2];
      v[0] = Sab;
      mat_vec<N, true>(Aab, Sl, v[0]);
      v[1] = Sl;
      mat_vec<N, true>(R, Sab, v[1]);
      lu_solve<N, 2>(Den, v);
      store_vec<N>(crec, st, CRec::RSD, v[1]);
      {
        // a_below = R + T Den^-1 (a_above T)
        rv X[N];
        mul<N, N, false, false>(Aab, nullptr, T, X);
        lu_solve<N, N>(Den, X);
        mul<N, N, true, false>(T, nullptr, X, R);
        store_colmajor<N, N>(crec, st, CRec::ABE, R);
      }
      // Eq. 34: source_below = source_lay + T Den^-1 (source_above + a_above source_lay)
      Sbe = Sl;
      mat_vec<N, true>(T, v[0], Sbe);
      store_vec<N>(crec, st, CRec::SBE, Sbe);
      // P = Den^-1 T
      lu_solve<N, N>(Den, T);
      store_colmajor<N, N>(crec, st, CRec::PDN, T);
    }
    const rv* Abe = R;
#if defined(SS_COLUMN16_PREFETCH_LW)
    if (jlay + 1 < nlay) {
      const double* lnext = L.lay + rec_base(jlay + 1, L.maxlay, LRec::NELEM, cu);
      SS_SCHED_FENCE();
      load_colmajor<N, N>(lnext, st, LRec::R, Rn);
      load_colmajor<N, N>(lnext, st, LRec::T, T);
      Sn = load_vec<N>(lnext, st, LRec::SRC);
      SS_LOAD_FENCE(); SS_SCHED_FENCE();
    }
#endif
    // overlap to the base of the layer above, or to the free atmosphere (forest :495-506, urban :583-626)
    {
      double U[NREG * DB], V[DB * NREG];
#if !defined(SS_COLUMN16_PREFETCH_LW)
      pack_fractions(prec, vf_lo, bf_lo);
      if (jlay + 1 < nlay) pack_fractions(pack_rec(L, jlay + 1, cu, kPackLw), vf_up, bf_up);
      if (URBAN) re = rec_pair_load(prec + 14 * kRecStride);   // roof_emissivity, roof_emission
#endif
      interface_overlap_vals<URBAN, true, NREG>(L.cfg.min_vegetation_fraction, jlay + 1 >= nlay, vf_up, bf_up,
                                                false, vf_lo, bf_lo, U, V);
      double roof_a = 0.0, roof_emis = 0.0, exposed_roof_frac = 0.0;
      if (URBAN) {
        if (jlay < nlay - 1) exposed_roof_frac = ss_max(0.0, bf_lo - bf_up);
        else exposed_roof_frac = bf_lo;
        roof_a = 1.0 - re.x;
        roof_emis = re.y;
      }
      rv urow[DB], tA[N];
      g.template own_row<DB>(U, urow);
      apply_V_right<NREG, NS, DB>(Abe, V, tA);
      apply_U<NREG, NS, N>(g, urow, tA, Aab);
      apply_U<NREG, NS, 1>(g, urow, &Sbe, &Sab);
      if (URBAN) {
        // roof rows: a_below(roof s, (ju, s2)) = roof_a hweight(s) V(roof, ju); source: hweight(s) emission x exposed fraction
        const rv ur = urow[DB - 1] * g.hw;
        SS_UNROLL_FULL
        for (int j = 0; j < N; ++j) Aab[j] = Aab[j] + ur * (roof_a * SS_E(V, DB - 1, j / NS, DB));
        Sab = Sab + ur * (roof_emis * exposed_roof_frac);
      }
    }
#if defined(SS_COLUMN16_PREFETCH_LW)
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) R[j] = Rn[j];
    Sl = Sn;
#endif
  }
  // top-of-canopy emissivity and emission (lw_top)
  {
    rv acc = 0.0;
    SS_UNROLL_FULL
    for (int j2 = 0; j2 < NS; ++j2) acc = acc + Aab[j2] * lg.hweight[j2];
    const double s1 = all_sum<NS>(acc), s2 = all_sum<NS>(Sab);
    if (lane0()) {
      L.top1[cidx] = 1.0 - s1;
      L.top2[cidx] = s2;
    }
  }
}

// ---------------------------------------------------------------- shortwave
template <bool URBAN, int NREG, int NS>
SS_HD void column_up_sw(const Launch& L, int cu) {
  constexpr int N = NREG * NS, D = NREG;
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  using LRec = SwRec<N, D>;
  using CRec = SwColRec<N, D>;
  constexpr long st = kRecStride;
  const Inputs& in = L.in;
  const LG& lg = L.lg;
  const SwColCtx<URBAN, NREG, NS> c(L, cu, false);
  const RowGeom<NREG, NS> g(lg);

  // a_above, d_above at the ground (forest :466-477, urban :594-606)
  rv Aab[N], Dab[D];
  {
    const double ga = in.ground_albedo[c.cidx], gad = in.ground_albedo_dir[c.cidx];
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) Aab[j] = select(g.rg[j / NS], ga * g.hw, rv(0.0));
    SS_UNROLL_FULL
    for (int r = 0; r < D; ++r) Dab[r] = select(g.rg[r], (c.cos_sza * gad) * g.hw, rv(0.0));
  }
  // (software pipeline of the record loads: see column_up_lw above; the small direct-beam blocks E, S_dn, S_up
  // of a layer are requested at the top of its step, a factorisation ahead of their use)
  rv R[N], T[N], Rn[N];
#if !defined(SS_NO_COLUMN16_PREFETCH)
  if (c.nlay > 0) {
    const double* lrec = L.lay + rec_base(0, L.maxlay, LRec::NELEM, cu);
    load_colmajor<N, N>(lrec, st, LRec::R, R);
    load_colmajor<N, N>(lrec, st, LRec::T, T);
  }
#endif
  for (int jlay = 0; jlay < c.nlay; ++jlay) {
    const double* lrec = L.lay + rec_base(jlay, L.maxlay, LRec::NELEM, cu);
    double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    // (R turns into a_below in place and T into P: five matrices live at the peak)
    rv Dbe[D];
    const double* prec = pack_rec(L, jlay, cu, kPackSw);
    double vf_lo, bf_lo, vf_up = 0.0, bf_up = 0.0;
    RecPair ra{0.0, 0.0};
    double E[D * D];
    rv S[D];
#if defined(SS_NO_COLUMN16_PREFETCH)
    load_colmajor<N, N>(lrec, st, LRec::R, R);
    load_colmajor<N, N>(lrec, st, LRec::T, T);
#else
    SS_UNROLL_FULL
    for (int k = 0; k < D * D; ++k) E[k] = rec_elem_load(lrec, st, LRec::EDIR + k);
    load_colmajor<N, D>(lrec, st, LRec::SDN, S);
    load_colmajor<N, D>(lrec, st, LRec::SUP, Dbe);
    pack_fractions(prec, vf_lo, bf_lo);
    if (jlay + 1 < c.nlay) pack_fractions(pack_rec(L, jlay + 1, cu, kPackSw), vf_up, bf_up);
    if (URBAN) ra = rec_pair_load(prec + 8 * kRecStride);   // roof_albedo, roof_albedo_dir
    SS_LOAD_FENCE(); SS_SCHED_FENCE();
#endif
    {
      // denominator = I - a_above R
      LU<N> Den;
      mul<N, N, false, false>(Aab, nullptr, R, Den.a);
      SS_UNROLL_FULL
      for (int j = 0; j < N; ++j) Den.a[j] = select(lane_eq(j), rv(1.0), rv(0.0)) - Den.a[j];
      lu_factor<N>(Den);
      rv Y[D];
      {
        // Q = Den^-1 (R (d_above E) + S_dn);  Y = Den^-1 (d_above E + a_above S_dn)
        rv Q[D];
#if defined(SS_NO_COLUMN16_PREFETCH)
        SS_UNROLL_FULL
        for (int k = 0; k < D * D; ++k) E[k] = rec_elem_load(lrec, st, LRec::EDIR + k);
        load_colmajor<N, D>(lrec, st, LRec::SDN, S);
#endif
        SS_UNROLL_FULL
        for (int j = 0; j < D; ++j) {
          rv acc = 0.0;
          SS_UNROLL_FULL
          for (int r = 0; r < D; ++r) acc = acc + Dab[r] * SS_E(E, r, j, D);
          Y[j] = acc;
          Q[j] = S[j];
        }
        mul<N, D, true, false>(R, nullptr, Y, Q);
        lu_solve<N, D>(Den, Q);
        store_colmajor<N, D>(crec, st, CRec::QDN, Q);
        mul<N, D, true, false>(Aab, nullptr, S, Y);
        lu_solve<N, D>(Den, Y);
      }
      {
        // a_below = R + T Den^-1 (a_above T)
        rv X[N];
        mul<N, N, false, false>(Aab, nullptr, T, X);
        lu_solve<N, N>(Den, X);
        mul<N, N, true, false>(T, nullptr, X, R);
        store_colmajor<N, N>(crec, st, CRec::ABE, R);
      }
      // d_below = S_up + T Y
#if defined(SS_NO_COLUMN16_PREFETCH)
      load_colmajor<N, D>(lrec, st, LRec::SUP, Dbe);
#endif
      mul<N, D, true, false>(T, nullptr, Y, Dbe);
      store_colmajor<N, D>(crec, st, CRec::DBE, Dbe);
      // P = Den^-1 T
      lu_solve<N, N>(Den, T);
      store_colmajor<N, N>(crec, st, CRec::PDN, T);
    }
    const rv* Abe = R;
#if !defined(SS_NO_COLUMN16_PREFETCH)
    if (jlay + 1 < c.nlay) {
      const double* lnext = L.lay + rec_base(jlay + 1, L.maxlay, LRec::NELEM, cu);
      SS_SCHED_FENCE();
      load_colmajor<N, N>(lnext, st, LRec::R, Rn);
      load_colmajor<N, N>(lnext, st, LRec::T, T);
      SS_LOAD_FENCE(); SS_SCHED_FENCE();
    }
#endif
    // overlap to the base of the layer above, or to the free atmosphere (forest :495-503, urban :627-653)
    {
      double U[NREG * DB], V[DB * NREG];
#if defined(SS_NO_COLUMN16_PREFETCH)
      pack_fractions(prec, vf_lo, bf_lo);
      if (jlay + 1 < c.nlay) pack_fractions(pack_rec(L, jlay + 1, cu, kPackSw), vf_up, bf_up);
      if (URBAN) ra = rec_pair_load(prec + 8 * kRecStride);   // roof_albedo, roof_albedo_dir
#endif
      interface_overlap_vals<URBAN, false, NREG>(L.cfg.min_vegetation_fraction, jlay + 1 >= c.nlay, vf_up, bf_up,
                                                 false, vf_lo, bf_lo, U, V);
      double roof_a = 0.0, roof_d = 0.0;
      if (URBAN) {
        roof_a = ra.x;
        roof_d = c.cos_sza * ra.y;
      }
      rv urow[DB], tA[N], tD[D];
      g.template own_row<DB>(U, urow);
      apply_V_right<NREG, NS, DB>(Abe, V, tA);
      apply_U<NREG, NS, N>(g, urow, tA, Aab);
      // tD = d_below V
      SS_UNROLL_FULL
      for (int ju = 0; ju < NREG; ++ju) {
        rv acc = 0.0;
        SS_UNROLL_FULL
        for (int jl = 0; jl < NREG; ++jl) acc = acc + Dbe[jl] * SS_E(V, jl, ju, DB);
        tD[ju] = acc;
      }
      apply_U<NREG, NS, D>(g, urow, tD, Dab);
      if (URBAN) {
        const rv ur = urow[DB - 1] * g.hw;
        SS_UNROLL_FULL
        for (int j = 0; j < N; ++j) Aab[j] = Aab[j] + ur * (roof_a * SS_E(V, DB - 1, j / NS, DB));
        SS_UNROLL_FULL
        for (int ju = 0; ju < D; ++ju) Dab[ju] = Dab[ju] + ur * (roof_d * SS_E(V, DB - 1, ju, DB));
      }
    }
#if !defined(SS_NO_COLUMN16_PREFETCH)
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) R[j] = Rn[j];
#endif
  }
  // top-of-canopy albedos (forest :520-528, urban :666-674)
  {
    rv acc = 0.0;
    SS_UNROLL_FULL
    for (int j2 = 0; j2 < NS; ++j2) acc = acc + Aab[j2] * lg.hweight[j2];
    const double s1 = all_sum<NS>(acc), s2 = all_sum<NS>(Dab[0]);
    if (lane0()) {
      L.top1[c.cidx] = s1;
      L.top2[c.cidx] = s2 / c.cos_sza;
    }
  }
}

}  // namespace r16
}  // namespace ss
