// ss_column_lw_sym.hpp - the longwave upward sweep (adding method with sources, radsurf_forest_lw.F90:452-506 /
// radsurf_urban_lw.F90:551-626) in SYMMETRISED coordinates, for the orders whose records are packed
// (LwLayRec<N>::PACKED, LwColRec<N>::PACKED).  Same construction as ss_column_sw_sym.hpp: u^ = D_l^-1 u with
// D_l = diag(sqrt(region fraction of layer l) x sqrt(weight mu)); R^, T^, a_above^, a_below^ symmetric and
// packed; the source vectors scaled by D^-1; P = Den^-1 T and the vectors of the column record restored on
// their way out (the downward sweep reads them unscaled), a_below^ stored as it is.
#pragma once

#include "ss_column_lw.hpp"
#include "ss_column_sw_sym.hpp"

namespace ss {
namespace sym {

// a_above^, source_above^ at the ground (forest :455-466, urban :554-565) in the scaling of the lowest layer
template <int NREG, int NS>
SS_HD void lw_ground(const LG& lg, double gem, double gemission, const double* sf, double* Aab, double* Sab) {
  constexpr int N = NREG * NS;
  double swm = 0.0;
  SS_UNROLL_FULL
  for (int s = 0; s < NS; ++s) swm += lg.sqrt_wmu[s] * lg.sqrt_wmu[s];
  const double rs = 1.0 / swm;
  SS_UNROLL_FULL
  for (int k = 0; k < kTri<N>; ++k) Aab[k] = 0.0;
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) {
    SS_UNROLL_FULL
    for (int sto = 0; sto < NS; ++sto) {
      // source_above = hweight(s) frac(r) emission;  / (sqrt(frac) sqrt_wmu(s))
      Sab[sto + r * NS] = (lg.hweight[sto] * lg.rsqrt_wmu[sto]) * sf[r] * gemission;
      SS_UNROLL_FULL
      for (int sfr = 0; sfr <= sto; ++sfr)
        Aab[lo<N>(sto + r * NS, sfr + r * NS)] = (1.0 - gem) * (lg.sqrt_wmu[sto] * lg.sqrt_wmu[sfr]) * rs;
    }
  }
}

// One layer: consumes R^, T^ (packed) and source_lay (unscaled) of the layer record; stores P, RSD, source_below
// unscaled and a_below^ packed into the column record; returns a_below^ and source_below^ (scaled).
template <int N, class Src>
SS_HD void lw_up_step(const Src& lrec, const double* Aab, const double* Sab, double* crec, const double* dsc,
                      const double* rd, double* Abe, double* Sbe) {
  using LRec = LwLayRec<N>;
  using CRec = LwColRec<N>;
  static_assert(LRec::PACKED && CRec::PACKED, "packed records");
  constexpr long st = kRecStride;
  double R[rec_even(kTri<N>)], T[rec_even(kTri<N>)], Sl[N];
  lrec.template load<rec_even(kTri<N>)>(LRec::R, R);
  lrec.template load<rec_even(kTri<N>)>(LRec::T, T);
  lrec.template load<N>(LRec::SRC, Sl);
  SS_LOAD_FENCE(); SS_SCHED_FENCE();
  SS_UNROLL_FULL
  for (int i = 0; i < N; ++i) Sl[i] *= rd[i];
  double Den[N * N];
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
      SS_UNROLL_FULL
      for (int k = 0; k < N; ++k) s -= Aab[sy<N>(i, k)] * R[sy<N>(k, j)];
      SS_E(Den, i, j, N) = s;
    }
  }
  lu_factor<N>(Den);
  double X[N * N];
  // a_below^ = R^ + T^ Den^-1 (a_above^ T^): lower triangle
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
      SS_UNROLL_FULL
      for (int k = 0; k < N; ++k) s += Aab[sy<N>(i, k)] * T[sy<N>(k, j)];
      SS_E(X, i, j, N) = s;
    }
  }
  lu_solve_mat<N, N>(Den, X);
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_FULL
    for (int i = j; i < N; ++i) {
      double s = R[lo<N>(i, j)];
      SS_UNROLL_FULL
      for (int k = 0; k < N; ++k) s += T[sy<N>(i, k)] * SS_E(X, k, j, N);
      Abe[lo<N>(i, j)] = s;
    }
  }
  static_loop<0, rec_even(kTri<N>) / 2>([&](auto pc) {
    constexpr int e = 2 * decltype(pc)::value;
    rec_pair_store(crec + (long)(CRec::ABE + e) * st, Abe[e], (e + 1 < kTri<N>) ? Abe[e + 1 < kTri<N> ? e + 1 : 0] : 0.0);
  });
  // source_below^ = source_lay^ + T^ Den^-1 (source_above^ + a_above^ source_lay^)   (Eq. 34)
  {
    double v[N];
    symv<N>(Aab, Sl, v);
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) v[i] = Sab[i] + v[i];
    lu_solve<N>(Den, v);
    symv<N>(T, v, Sbe);
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) Sbe[i] = Sl[i] + Sbe[i];
    double t[N];
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) t[i] = dsc[i] * Sbe[i];
    rec_store<N>(crec, st, CRec::SBE, t);
  }
  // operators of the downward sweep: P = D Den^-1 T^ D^-1, RSD = D Den^-1 (R^ source_above^ + source_lay^)
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) SS_E(X, i, j, N) = T[sy<N>(i, j)];
  }
  lu_solve_mat<N, N>(Den, X);
  SS_UNROLL_FULL
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) SS_E(X, i, j, N) *= dsc[i] * rd[j];
  }
  rec_store<N * N>(crec, st, CRec::PDN, X);
  {
    double rs[N];
    symv<N>(R, Sab, rs);
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) rs[i] = rs[i] + Sl[i];
    lu_solve<N>(Den, rs);
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) rs[i] *= dsc[i];
    rec_store<N>(crec, st, CRec::RSD, rs);
  }
}

// Overlap to the base of the layer above layer jlay, or to the free atmosphere (forest :495-506, urban :583-626)
template <bool URBAN, int NREG, int NS>
SS_HD void lw_overlap_above(const Launch& L, int cu, int jlay, int nlay, const double* Abe, const double* Sbe,
                            const double* sf_lo, const double* rsf_lo, double* sf_up, double* rsf_up,
                            double* Aab, double* Sab) {
  constexpr int N = NREG * NS;
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  const LG& lg = L.lg;
  double U[NREG * DB], V[DB * NREG];
  const double* prec = pack_rec(L, jlay, cu, kPackLw);
  double vf_lo, bf_lo, vf_up = 0.0, bf_up = 0.0;
  pack_fractions(prec, vf_lo, bf_lo);
  const bool sky = jlay + 1 >= nlay;
  if (!sky) pack_fractions(pack_rec(L, jlay + 1, cu, kPackLw), vf_up, bf_up);
  interface_overlap_vals<URBAN, true, NREG>(L.cfg.min_vegetation_fraction, sky, vf_up, bf_up, false, vf_lo, bf_lo, U, V);
  {
    double fu[NREG];
    if (sky) {
      fu[0] = 1.0;
      SS_UNROLL_FULL
      for (int r = 1; r < NREG; ++r) fu[r] = 0.0;
    } else {
      region_fractions<URBAN, true, NREG>((NREG > 1) ? vf_up : 0.0, URBAN ? bf_up : 0.0, fu);
    }
    region_roots<NREG>(fu, sf_up, rsf_up);
  }
  double roof_a = 0.0, roof_emis = 0.0, exposed_roof_frac = 0.0;
  if (URBAN) {
    // urban :587-605
    if (jlay < nlay - 1) exposed_roof_frac = ss_max(0.0, bf_lo - bf_up);
    else exposed_roof_frac = bf_lo;
    const RecPair re = rec_pair_load(prec + 14 * kRecStride);   // roof_emissivity, roof_emission
    roof_a = 1.0 - re.x;
    roof_emis = re.y;
  }
  double Vs[DB * NREG], Us[NREG * DB];
  SS_UNROLL_FULL
  for (int ju = 0; ju < NREG; ++ju) {
    SS_UNROLL_FULL
    for (int jl = 0; jl < DB; ++jl) {
      const double lo_s = (jl < NREG) ? sf_lo[jl < NREG ? jl : 0] : 1.0, lo_r = (jl < NREG) ? rsf_lo[jl < NREG ? jl : 0] : 1.0;
      SS_E(Vs, jl, ju, DB) = SS_E(V, jl, ju, DB) * (sf_up[ju] * lo_r);
      SS_E(Us, ju, jl, NREG) = SS_E(U, ju, jl, NREG) * (lo_s * rsf_up[ju]);
    }
  }
  double tA[N * N], tAroof[NS * N];
  SS_UNROLL_FULL
  for (int ju = 0; ju < NREG; ++ju) {
    SS_UNROLL_FULL
    for (int s = 0; s < NS; ++s) {
      const int c = ju * NS + s;
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) {
        double acc = 0.0;
        SS_UNROLL_FULL
        for (int jl = 0; jl < NREG; ++jl) acc += Abe[sy<N>(i, jl * NS + s)] * SS_E(Vs, jl, ju, DB);
        SS_E(tA, i, c, N) = acc;
      }
      SS_UNROLL_FULL
      for (int sr = 0; sr < NS; ++sr)
        SS_E(tAroof, sr, c, NS) = URBAN ? (roof_a * lg.hweight[sr] * lg.rsqrt_wmu[sr]) * (SS_E(Vs, DB - 1, ju, DB) * lg.sqrt_wmu[s]) : 0.0;
    }
  }
  {
    double Af[N * N];
    apply_U_mat<NREG, DB, NS, N>(Us, tA, tAroof, Af);
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      SS_UNROLL_FULL
      for (int i = j; i < N; ++i) Aab[lo<N>(i, j)] = 0.5 * (SS_E(Af, i, j, N) + SS_E(Af, j, i, N));
    }
  }
  // source_above^(l+1) = U^ [source_below^; roof emission / sqrt(weight mu)]
  double sroof[NS];
  SS_UNROLL_FULL
  for (int s = 0; s < NS; ++s) sroof[s] = URBAN ? (lg.hweight[s] * lg.rsqrt_wmu[s]) * roof_emis * exposed_roof_frac : 0.0;
  apply_U_mat<NREG, DB, NS, 1>(Us, Sbe, sroof, Sab);
}

// top-of-canopy emissivity and emission (clear-sky region of the free atmosphere: scaling sqrt(weight mu))
template <int NREG, int NS>
SS_HD void lw_top(const Launch& L, long cidx, const double* Aab, const double* Sab) {
  constexpr int N = NREG * NS;
  const LG& lg = L.lg;
  double s = 0.0;
  SS_UNROLL_FULL
  for (int j1 = 0; j1 < NS; ++j1) {
    double acc = 0.0;
    SS_UNROLL_FULL
    for (int j2 = 0; j2 < NS; ++j2) acc += Aab[sy<N>(j1, j2)] * (lg.rsqrt_wmu[j2] * lg.hweight[j2]);
    s += lg.sqrt_wmu[j1] * acc;
  }
  L.top1[cidx] = 1.0 - s;           // top_emissivity
  s = 0.0;
  SS_UNROLL_FULL
  for (int j1 = 0; j1 < NS; ++j1) s += lg.sqrt_wmu[j1] * Sab[j1];
  L.top2[cidx] = s;                 // top_emission
}

template <bool URBAN, int NREG, int NS>
SS_HD void column_up_lw(const Launch& L, int cu) {
  constexpr int N = NREG * NS;
  using LRec = LwLayRec<N>;
  using CRec = LwColRec<N>;
  const Inputs& in = L.in;
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  const int nlay = in.nlay[col];
  const long cidx = (long)jb + (long)L.nspec * col;
  double Aab[kTri<N>], Sab[N];
  double sf[NREG], rsf[NREG];
  {
    double vf = 0.0, bf = 0.0, frac[NREG];
    if (nlay > 0) pack_fractions(pack_rec(L, 0, cu, kPackLw), vf, bf);
    region_fractions<URBAN, true, NREG>(vf, bf, frac);
    region_roots<NREG>(frac, sf, rsf);
  }
  sym::lw_ground<NREG, NS>(L.lg, in.ground_emissivity[cidx], in.ground_emission[cidx], sf, Aab, Sab);
  for (int jlay = 0; jlay < nlay; ++jlay) {
    const RecSource lrec{L.lay + rec_base(jlay, L.maxlay, LRec::NELEM, cu), kRecStride};
    double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    double dsc[N], rd[N];
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) {
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) {
        dsc[s + r * NS] = sf[r] * L.lg.sqrt_wmu[s];
        rd[s + r * NS] = rsf[r] * L.lg.rsqrt_wmu[s];
      }
    }
    double Abe[kTri<N>], Sbe[N];
    sym::lw_up_step<N>(lrec, Aab, Sab, crec, dsc, rd, Abe, Sbe);
    double sfu[NREG], rsfu[NREG];
    sym::lw_overlap_above<URBAN, NREG, NS>(L, cu, jlay, nlay, Abe, Sbe, sf, rsf, sfu, rsfu, Aab, Sab);
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) { sf[r] = sfu[r]; rsf[r] = rsfu[r]; }
  }
  sym::lw_top<NREG, NS>(L, cidx, Aab, Sab);
}

// ---- layer matrices AND adding step of every layer of a column unit on one lane (k_layer_up), both in the
// scaled coordinates (-DSS_SYM_FUSE_LW) ----
// The layer unit writes R^, T^ (2 x 22 slots) and the layer's source (6) into the lane's LDS slots; the
// integrated-flux matrix and source, which only the downward sweep reads, go straight to the column record.
// No layer record exists.  Nothing is carried across the layer stage: a_above^ / source_above^ of a layer are
// formed from a_below^ / source_below of the layer below, which the previous step wrote to the column record
// for the downward sweep anyway (cache), and the scaling of either layer follows from its two fractions.
template <int N, class Lds>
struct LwSymFusedWriter {
  using LRec = LwLayRec<N>;
  using CRec = LwColRec<N>;
  static constexpr int kSrcSlot = LRec::IFLUX;      // the source follows R^, T^ in the slots
  const Lds* lds;
  double* crec;
  long stride;
  unsigned long long* stamps;
  SS_HD void put2(int e, double v0, double v1) const {
    if (e < LRec::IFLUX) {
      lds->put(e, v0);
      lds->put(e + 1, v1);
    } else if (e < LRec::SRC) {
      rec_pair_store(crec + (long)(CRec::IFLUX + e - LRec::IFLUX) * stride, v0, v1);
    } else if (e < LRec::ISRC) {
      lds->put(kSrcSlot + e - LRec::SRC, v0);
      lds->put(kSrcSlot + e + 1 - LRec::SRC, v1);
    } else {
      rec_pair_store(crec + (long)(CRec::ISRC + e - LRec::ISRC) * stride, v0, v1);
    }
  }
  SS_HD const Lds& scratch() const { return *lds; }   // the slots are free until R^ and T^ are put
};
template <int N, class Lds>
struct LwSymFusedSource {
  using LRec = LwLayRec<N>;
  const Lds* lds;
  template <int NEL>
  SS_HD void load(int off, double* dst) const {
    const int at = (off < LRec::IFLUX) ? off : LwSymFusedWriter<N, Lds>::kSrcSlot + off - LRec::SRC;
    SS_UNROLL_FULL
    for (int k = 0; k < NEL; ++k) dst[k] = lds->get(at + k);
  }
};

// square roots of the region fractions of layer jlay (and their reciprocals)
template <bool URBAN, int NREG>
SS_HD void lw_layer_roots(const Launch& L, int cu, int jlay, double* sf, double* rsf) {
  double vf, bf, frac[NREG];
  pack_fractions(pack_rec(L, jlay, cu, kPackLw), vf, bf);
  region_fractions<URBAN, true, NREG>(vf, bf, frac);
  region_roots<NREG>(frac, sf, rsf);
}

template <bool URBAN, int NREG, int NS>
SS_HD void layer_up_lw(const Launch& L, int cu) {
  constexpr int N = NREG * NS;
  using LRec = LwLayRec<N>;
  using CRec = LwColRec<N>;
  static_assert(CRec::FUSED && LRec::PACKED && CRec::PACKED && LRec::IFLUX + N <= kLaneLdsSlots,
                "R^, T^ and the source must fit the lane's LDS slots");
  using Lds = LaneLds<true>;
  constexpr long st = kRecStride;
  const Inputs& in = L.in;
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  const int nlay = in.nlay[col];
  const long cidx = (long)jb + (long)L.nspec * col;
  const Lds lds{};
  // a_above^ / source_above^ at the top of layer jlo, in the scaling of the layer above it (or of the sky),
  // from the column record of layer jlo
  auto above = [&](int jlo, double* sf_up, double* rsf_up, double* Aab, double* Sab) {
    const double* prev = L.col + rec_base(jlo, L.maxlay, CRec::NELEM, cu);
    double Abe[rec_even(kTri<N>)], Sbe[N];
    rec_load<rec_even(kTri<N>)>(prev, st, CRec::ABE, Abe);
    rec_load<N>(prev, st, CRec::SBE, Sbe);
    double sfl[NREG], rsfl[NREG];
    lw_layer_roots<URBAN, NREG>(L, cu, jlo, sfl, rsfl);
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) {
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) Sbe[s + r * NS] *= rsfl[r] * L.lg.rsqrt_wmu[s];   // stored unscaled
    }
    sym::lw_overlap_above<URBAN, NREG, NS>(L, cu, jlo, nlay, Abe, Sbe, sfl, rsfl, sf_up, rsf_up, Aab, Sab);
  };
  double Aab[kTri<N>], Sab[N], sf[NREG], rsf[NREG];
  for (int jlay = 0; jlay < nlay; ++jlay) {
    double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    const LwSymFusedWriter<N, Lds> w{&lds, crec, st, (unsigned long long*)L.eigen_fail};
    layer_unit_lw<URBAN, NREG, NS>(L, cu, jlay, w);
    if (jlay == 0) {
      lw_layer_roots<URBAN, NREG>(L, cu, 0, sf, rsf);
      sym::lw_ground<NREG, NS>(L.lg, in.ground_emissivity[cidx], in.ground_emission[cidx], sf, Aab, Sab);
    } else {
      above(jlay - 1, sf, rsf, Aab, Sab);
    }
    double dsc[N], rd[N];
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) {
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) {
        dsc[s + r * NS] = sf[r] * L.lg.sqrt_wmu[s];
        rd[s + r * NS] = rsf[r] * L.lg.rsqrt_wmu[s];
      }
    }
    const LwSymFusedSource<N, Lds> src{&lds};
    double Abe[kTri<N>], Sbe[N];
    sym::lw_up_step<N>(src, Aab, Sab, crec, dsc, rd, Abe, Sbe);
  }
  if (nlay > 0) {
    above(nlay - 1, sf, rsf, Aab, Sab);
  } else {
    double frac[NREG];
    region_fractions<URBAN, true, NREG>(0.0, 0.0, frac);
    region_roots<NREG>(frac, sf, rsf);
    sym::lw_ground<NREG, NS>(L.lg, in.ground_emissivity[cidx], in.ground_emission[cidx], sf, Aab, Sab);
  }
  sym::lw_top<NREG, NS>(L, cidx, Aab, Sab);
}

// Second form (-DSS_SYM_FUSE_LW2, round 4): the loop-carried a_above^ / source_above^ wait in the lane's LDS slots
// behind R^, T^ and the layer's source (50 + 22 + 6 of the 80) while the layer stage runs, instead of being
// re-derived from the column record of the layer below (a round trip through memory and the overlap step again);
// what the step reads of the launch descriptor it reads inside the step (SS_LAUNCH_HERE_REF: see ss_layer_up_sw2.hpp).
template <bool URBAN, int NREG, int NS>
SS_HD void layer_up_lw2(const Launch& L, int cu) {
  constexpr int N = NREG * NS;
  using LRec = LwLayRec<N>;
  using CRec = LwColRec<N>;
  using Lds = LaneLds<true>;
  constexpr int kA = LRec::IFLUX + rec_even(N);                  // a_above^ (packed) behind R^, T^, source
  constexpr int kS = kA + rec_even(kTri<N>);                     // source_above^
  static_assert(CRec::FUSED && LRec::PACKED && CRec::PACKED && kS + N <= kLaneLdsSlots, "LDS slots of a lane");
  constexpr long st = kRecStride;
  const Inputs& in = L.in;
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  const int nlay = in.nlay[col];
  const long cidx = (long)jb + (long)L.nspec * col;
  const Lds lds{};
  {
    // ground, in the scaling of the lowest layer (or of the sky for a column without layers)
    double Aab[kTri<N>], Sab[N], sf[NREG], rsf[NREG];
    if (nlay > 0) {
      lw_layer_roots<URBAN, NREG>(L, cu, 0, sf, rsf);
    } else {
      double frac[NREG];
      region_fractions<URBAN, true, NREG>(0.0, 0.0, frac);
      region_roots<NREG>(frac, sf, rsf);
    }
    sym::lw_ground<NREG, NS>(L.lg, in.ground_emissivity[cidx], in.ground_emission[cidx], sf, Aab, Sab);
    if (nlay == 0) {
      sym::lw_top<NREG, NS>(L, cidx, Aab, Sab);
      return;
    }
    SS_UNROLL_FULL
    for (int k = 0; k < kTri<N>; ++k) lds.put(kA + k, Aab[k]);
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) lds.put(kS + i, Sab[i]);
  }
  for (int jlay = 0; jlay < nlay; ++jlay) {
    SS_LAUNCH_HERE_REF(L, Lq);
    double* crec = Lq.col + rec_base(jlay, Lq.maxlay, CRec::NELEM, cu);
    {
      const LwSymFusedWriter<N, Lds> w{&lds, crec, st, (unsigned long long*)Lq.eigen_fail};
      layer_unit_lw<URBAN, NREG, NS>(Lq, cu, jlay, w);
    }
    double sf[NREG], rsf[NREG];
    lw_layer_roots<URBAN, NREG>(Lq, cu, jlay, sf, rsf);
    double dsc[N], rd[N];
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) {
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) {
        dsc[s + r * NS] = sf[r] * Lq.lg.sqrt_wmu[s];
        rd[s + r * NS] = rsf[r] * Lq.lg.rsqrt_wmu[s];
      }
    }
    double Aab[kTri<N>], Sab[N];
    SS_UNROLL_FULL
    for (int k = 0; k < kTri<N>; ++k) Aab[k] = lds.get(kA + k);
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) Sab[i] = lds.get(kS + i);
    const LwSymFusedSource<N, Lds> src{&lds};
    double Abe[kTri<N>], Sbe[N];
    sym::lw_up_step<N>(src, Aab, Sab, crec, dsc, rd, Abe, Sbe);
    double sfu[NREG], rsfu[NREG];
    sym::lw_overlap_above<URBAN, NREG, NS>(Lq, cu, jlay, nlay, Abe, Sbe, sf, rsf, sfu, rsfu, Aab, Sab);
    if (jlay + 1 < nlay) {
      SS_UNROLL_FULL
      for (int k = 0; k < kTri<N>; ++k) lds.put(kA + k, Aab[k]);
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) lds.put(kS + i, Sab[i]);
    } else {
      sym::lw_top<NREG, NS>(Lq, cidx, Aab, Sab);
    }
  }
}

}  // namespace sym

// layer + upward sweep on one lane (k_layer_up)
template <bool URBAN, int NREG, int NS>
SS_HD void layer_up_lw(const Launch& L, int cu) {
  if constexpr (LwLayRec<NREG * NS>::PACKED && LwColRec<NREG * NS>::PACKED) {
#if defined(SS_SYM_FUSE_LW2)
    sym::layer_up_lw2<URBAN, NREG, NS>(L, cu);
#else
    sym::layer_up_lw<URBAN, NREG, NS>(L, cu);
#endif
  } else {
    layer_up_lw_full<URBAN, NREG, NS>(L, cu);
  }
}

// The upward sweep of one column unit (k_column): in symmetrised coordinates where the records are packed
template <bool URBAN, int NREG, int NS>
SS_HD void column_up_lw(const Launch& L, int cu) {
  if constexpr (LwLayRec<NREG * NS>::PACKED && LwColRec<NREG * NS>::PACKED) sym::column_up_lw<URBAN, NREG, NS>(L, cu);
  else column_up_lw_full<URBAN, NREG, NS>(L, cu);
}

// Whole column on one lane (host emulation and small problems)
template <bool URBAN, int NREG, int NS>
SS_HD void column_unit_lw(const Launch& L, int cu) {
  double slot = 0.0;
  if constexpr (LwColRec<NREG * NS>::FUSED) layer_up_lw<URBAN, NREG, NS>(L, cu);
  else column_up_lw<URBAN, NREG, NS>(L, cu);
  column_down_lw<URBAN, NREG, NS>(L, cu, 0, &slot);
  column_down_lw<URBAN, NREG, NS>(L, cu, 1, &slot);
}

}  // namespace ss
