// ss_column_sw_sym.hpp - the shortwave upward sweep (adding method, radsurf_forest_sw.F90:463-528 /
// radsurf_urban_sw.F90:591-674) in SYMMETRISED coordinates, for the orders whose layer record holds R^ and T^
// packed (SwLayRec<N, D>::PACKED, ss_layer_matrices.hpp).
//
// With u^ = D_l^-1 u, D_l = diag(sqrt(region fraction of layer l) x sqrt(weight mu)) (ss_layer_sym.hpp), the
// recursion of the reference keeps its form - it is a similarity transform layer by layer - and gains the
// symmetry the unscaled one hides: R^, T^ and the albedo of everything below an interface, a_above^ / a_below^,
// are symmetric (reciprocity), so they live as 21 instead of 36 doubles (n = 6), the layer record is read
// without restoring R = D R^ D^-1, and a_below^ = R^ + T^ X needs its lower triangle only.  The live state of
// the step drops from ~200 to ~150 doubles: the unscaled form spilled 90 dwords per lane, a quarter of the
// bytes this bandwidth-bound kernel moved.  The interface to the layer above changes the scaling (its region
// fractions differ): a_above^ = U^ [a_below^ V^; roof rows] with U^ = D_up^-1 (U x I) D_lo, V^ = D_lo^-1 (V x I) D_up.
// Direct-beam columns (one per region) stay unscaled; the rows of d_above / d_below / S_up / S_dn are scaled.
// The column record keeps the layout and the (unscaled) contents the downward sweep reads: P, Q, a_below,
// d_below are restored on their way out (a multiplication per element, no registers).
#pragma once

#include "ss_column_sw.hpp"
#include "ss_down_sw_both.hpp"
#include "ss_layer_sym.hpp"

namespace ss {
namespace sym {

// sqrt of the region fractions and their reciprocals (0 for a region without area)
template <int NREG>
SS_HD void region_roots(const double* frac, double* sf, double* rsf) {
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) {
    sf[r] = sqrt(frac[r]);
    rsf[r] = frac[r] > 0.0 ? 1.0 / sf[r] : 0.0;
  }
}

// a_above^, d_above^ at the ground (forest :466-477, urban :594-606) in the scaling of the lowest layer
template <int NREG, int NS>
SS_HD void sw_ground(const LG& lg, double ga, double gad, double cos_sza, const double* rsf, double* Aab, double* Dab) {
  constexpr int N = NREG * NS, D = NREG;
  double swm = 0.0;
  SS_UNROLL_FULL
  for (int s = 0; s < NS; ++s) swm += lg.sqrt_wmu[s] * lg.sqrt_wmu[s];
  const double rs = 1.0 / swm;
  SS_UNROLL_FULL
  for (int k = 0; k < kTri<N>; ++k) Aab[k] = 0.0;
  mat_zero<N, D>(Dab);
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) {
    SS_UNROLL_FULL
    for (int sto = 0; sto < NS; ++sto) {
      // hweight(s) / sqrt_wmu(s) = sqrt_wmu(s) / sum(w mu)
      SS_E(Dab, sto + r * NS, r, N) = (cos_sza * gad) * (lg.hweight[sto] * lg.rsqrt_wmu[sto]) * rsf[r];
      SS_UNROLL_FULL
      for (int sfr = 0; sfr <= sto; ++sfr)
        Aab[lo<N>(sto + r * NS, sfr + r * NS)] = ga * (lg.sqrt_wmu[sto] * lg.sqrt_wmu[sfr]) * rs;
    }
  }
}

// One layer: consumes R^, T^ (packed), S_up^, S_dn^ (rows scaled), E of the layer record; stores P, Q, a_below,
// d_below UNSCALED into the column record; returns a_below^ (packed) and d_below^ (rows scaled).
// src: the layer record in HBM (RecSource) or what the fused layer stage left in LDS / registers (FusedSource)
template <int N, int D, class Src>
SS_HD void sw_adding_step(const Src& src, const Src* next, double* R, double* T, const double* Aab,
                          const double* Dab, double* crec, const double* dsc, const double* rd, double* Abe,
                          double* Dbe, const double* dtop = nullptr) {
  using LRec = SwLayRec<N, D>;
  using CRec = SwColRec<N, D>;
  static_assert(LRec::PACKED, "packed layer record");
  constexpr long st = kRecStride;
  // Software pipeline of the record loads (every one of them used to be a full memory round trip in the
  // middle of a layer step: 62 % of the sweep's cycles waited).  R^ and T^ arrive in R, T - requested by the
  // caller for the first layer, by the previous step for the others.  E and S_dn are requested here, a matrix
  // factorisation ahead of their use; S_up takes the place of S_dn before the last solve; R^, T^ of the next
  // layer are requested as soon as T^ has been used for the last time, ahead of the overlap step.  Loads
  // complete in the order of issue (one counter), so each wait only covers what was requested before it.
  double E[D * D], S[N * D];
#if !defined(SS_NO_COLUMN_PREFETCH)
  src.template load<D * D>(LRec::EDIR, E);
  src.template load<N * D>(LRec::SDN, S);
#endif
  SS_LOAD_FENCE(); SS_SCHED_FENCE();
  // denominator = I - a_above^ R^
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
  // P^ = Den^-1 T^, stored as P = D P^ D^-1
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
  if constexpr (CRec::PACKED) {
    // ... stored as it is: packed and scaled (the downward sweep applies it as D (a_below^ (D^-1 x)))
    static_loop<0, rec_even(kTri<N>) / 2>([&](auto pc) {
      constexpr int e = 2 * decltype(pc)::value;
      rec_pair_store(crec + (long)(CRec::ABE + e) * st, Abe[e], (e + 1 < kTri<N>) ? Abe[e + 1 < kTri<N> ? e + 1 : 0] : 0.0);
    });
  } else {
    // ... stored unscaled and full: a_below = D a_below^ D^-1 (X is free)
    SS_UNROLL_FULL
    for (int j = 0; j < N; ++j) {
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) SS_E(X, i, j, N) = (dsc[i] * rd[j]) * Abe[sy<N>(i, j)];
    }
    rec_store<N * N>(crec, st, CRec::ABE, X);
  }
  // d_below^ = S_up^ + T^ Den^-1 (d_above^ E + a_above^ S_dn^);  Q^ = Den^-1 (R^ (d_above^ E) + S_dn^)
  double Y[N * D];
#if defined(SS_NO_COLUMN_PREFETCH)
  src.template load<D * D>(LRec::EDIR, E);
  src.template load<N * D>(LRec::SDN, S);
#endif
  mat_mul<N, D, D>(Dab, E, Y);
  {
    double Q[N * D];
    SS_UNROLL_FULL
    for (int j = 0; j < D; ++j) {
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) {
        double s = SS_E(S, i, j, N);
        SS_UNROLL_FULL
        for (int k = 0; k < N; ++k) s += R[sy<N>(i, k)] * SS_E(Y, k, j, N);
        SS_E(Q, i, j, N) = s;
      }
    }
    lu_solve_mat<N, D>(Den, Q);
    SS_UNROLL_FULL
    for (int j = 0; j < D; ++j) {
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) SS_E(Q, i, j, N) *= dsc[i];
    }
    if constexpr (CRec::DIRVEC) {
      // Q dir_top (dtop: the direct flux at the top of this layer, column_up_sw's first stage)
      double qd[N];
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) {
        double a = 0.0;
        SS_UNROLL_FULL
        for (int j = 0; j < D; ++j) a += SS_E(Q, i, j, N) * dtop[j];
        qd[i] = a;
      }
      rec_store<N>(crec, st, CRec::QDN, qd);
    } else {
      rec_store<N * D>(crec, st, CRec::QDN, Q);
    }
  }
  SS_UNROLL_FULL
  for (int j = 0; j < D; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) {
      double s = SS_E(Y, i, j, N);
      SS_UNROLL_FULL
      for (int k = 0; k < N; ++k) s += Aab[sy<N>(i, k)] * SS_E(S, k, j, N);
      SS_E(Y, i, j, N) = s;
    }
  }
#if !defined(SS_NO_COLUMN_PREFETCH)
  src.template load<N * D>(LRec::SUP, S);
  SS_LOAD_FENCE(); SS_SCHED_FENCE();
  lu_solve_mat<N, D>(Den, Y);
#else
  lu_solve_mat<N, D>(Den, Y);
  src.template load<N * D>(LRec::SUP, S);
#endif
  SS_UNROLL_FULL
  for (int j = 0; j < D; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) {
      double s = SS_E(S, i, j, N);
      SS_UNROLL_FULL
      for (int k = 0; k < N; ++k) s += T[sy<N>(i, k)] * SS_E(Y, k, j, N);
      SS_E(Dbe, i, j, N) = s;
    }
  }
#if !defined(SS_NO_COLUMN_PREFETCH)
  if (next) {
    SS_SCHED_FENCE();
    next->template load<rec_even(kTri<N>)>(LRec::R, R);
    next->template load<rec_even(kTri<N>)>(LRec::T, T);
    SS_LOAD_FENCE(); SS_SCHED_FENCE();
  }
#endif
  SS_UNROLL_FULL
  for (int j = 0; j < D; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) SS_E(S, i, j, N) = dsc[i] * SS_E(Dbe, i, j, N);
  }
  if constexpr (CRec::DIRVEC) {
    double dd[N];
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) {
      double a = 0.0;
      SS_UNROLL_FULL
      for (int j = 0; j < D; ++j) a += SS_E(S, i, j, N) * dtop[j];
      dd[i] = a;
    }
    rec_store<N>(crec, st, CRec::DBE, dd);
  } else {
    rec_store<N * D>(crec, st, CRec::DBE, S);
  }
}

// First stage of the upward kernel where the column record is in its DIRVEC form: the direct flux at the top of
// every layer for a unit direct beam on the horizontal at canopy top (the pass the downward kernel normalises
// by the direct flux), top down - through the overlap rules at every interface and the direct transmittance E
// of every layer, exactly the recursion of column_down_sw -, into element DTP of the column records.
template <bool URBAN, int NREG, int NS>
SS_HD void sw_direct_tops(const Launch& L, int cu, int nlay, double cos_sza) {
  constexpr int N = NREG * NS, D = NREG;
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  using LRec = SwLayRec<N, D>;
  using CRec = SwColRec<N, D>;
  constexpr long st = kRecStride;
  double dir_ab[D];
  dir_ab[0] = 1.0 / cos_sza;
  SS_UNROLL_FULL
  for (int r = 1; r < D; ++r) dir_ab[r] = 0.0;
  double vf_up = 0.0, bf_up = 0.0;
  double E[D * D], vf_lo = 0.0, bf_lo = 0.0;
  if (nlay > 0) {
    rec_load<D * D>(L.lay + rec_base(nlay - 1, L.maxlay, LRec::NELEM, cu), st, LRec::EDIR, E);
    pack_fractions(pack_rec(L, nlay - 1, cu, kPackSw), vf_lo, bf_lo);
  }
  for (int jl = nlay - 1; jl >= 0; --jl) {
    // (what the next layer down needs is requested a step ahead)
    double En[D * D], vf_n = 0.0, bf_n = 0.0;
    if (jl > 0) {
      rec_load<D * D>(L.lay + rec_base(jl - 1, L.maxlay, LRec::NELEM, cu), st, LRec::EDIR, En);
      pack_fractions(pack_rec(L, jl - 1, cu, kPackSw), vf_n, bf_n);
    }
    SS_LOAD_FENCE(); SS_SCHED_FENCE();
    double U[NREG * DB], V[DB * NREG];
    interface_overlap_vals<URBAN, false, NREG>(L.cfg.min_vegetation_fraction, jl + 1 >= nlay, vf_up, bf_up, false,
                                               vf_lo, bf_lo, U, V);
    double db[D];
    SS_UNROLL_FULL
    for (int k = 0; k < D; ++k) {
      double a = 0.0;
      SS_UNROLL_FULL
      for (int ju = 0; ju < NREG; ++ju) a += SS_E(V, k, ju, DB) * dir_ab[ju];
      db[k] = a;
    }
    rec_store<D>(L.col + rec_base(jl, L.maxlay, CRec::NELEM, cu), st, CRec::DTP, db);
    SS_UNROLL_FULL
    for (int r = 0; r < D; ++r) {
      double a = 0.0;
      SS_UNROLL_FULL
      for (int k = 0; k < D; ++k) a += SS_E(E, r, k, D) * db[k];
      dir_ab[r] = a;
    }
    SS_UNROLL_FULL
    for (int k = 0; k < D * D; ++k) E[k] = En[k];
    vf_up = vf_lo;
    bf_up = bf_lo;
    vf_lo = vf_n;
    bf_lo = bf_n;
  }
}

// Overlap to the base of the layer above layer jlay, or to the free atmosphere (forest :495-503, urban
// :627-653), from the scaling of layer jlay (sf_lo, rsf_lo) into that of the layer above (returned in
// sf_up, rsf_up): a_above^ = U^ [a_below^ V^; roof rows], d_above^ likewise.
// What the overlap step reads of the input packs (requested at the top of the layer step, see sw_adding_step)
struct SwOverlapIn {
  double vf_lo, bf_lo, vf_up, bf_up, roof_albedo, roof_albedo_dir;
  template <bool URBAN>
  SS_HD void load(const Launch& L, int cu, int jlay, int nlay) {
    const double* prec = pack_rec(L, jlay, cu, kPackSw);
    pack_fractions(prec, vf_lo, bf_lo);
    vf_up = bf_up = 0.0;
    if (jlay + 1 < nlay) pack_fractions(pack_rec(L, jlay + 1, cu, kPackSw), vf_up, bf_up);
    roof_albedo = roof_albedo_dir = 0.0;
    if (URBAN) {
      const RecPair ra = rec_pair_load(prec + 8 * kRecStride);   // roof_albedo, roof_albedo_dir
      roof_albedo = ra.x;
      roof_albedo_dir = ra.y;
    }
  }
};

template <bool URBAN, int NREG, int NS>
SS_HD void sw_overlap_above(const Launch& L, const SwOverlapIn& oi, int jlay, int nlay, double cos_sza, const double* Abe,
                            const double* Dbe, const double* sf_lo, const double* rsf_lo, double* sf_up,
                            double* rsf_up, double* Aab, double* Dab) {
  constexpr int N = NREG * NS, D = NREG;
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  const LG& lg = L.lg;
  double U[NREG * DB], V[DB * NREG];
  const double vf_lo = oi.vf_lo, bf_lo = oi.bf_lo, vf_up = oi.vf_up, bf_up = oi.bf_up;
  const bool sky = jlay + 1 >= nlay;
  interface_overlap_vals<URBAN, false, NREG>(L.cfg.min_vegetation_fraction, sky, vf_up, bf_up, false, vf_lo, bf_lo, U, V);
  {
    double fu[NREG];
    if (sky) {
      fu[0] = 1.0;
      SS_UNROLL_FULL
      for (int r = 1; r < NREG; ++r) fu[r] = 0.0;
    } else {
      region_fractions<URBAN, false, NREG>((NREG > 1) ? vf_up : 0.0, URBAN ? bf_up : 0.0, fu);
    }
    region_roots<NREG>(fu, sf_up, rsf_up);
  }
  const double roof_a = URBAN ? oi.roof_albedo : 0.0, roof_d = URBAN ? cos_sza * oi.roof_albedo_dir : 0.0;
  // Vs = V^ (diffuse columns: scaled on both sides), Us = U^
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
  // tA = a_below^ (V^ (x) I): canopy rows (N x N) and roof rows (NS x N; roof components scaled by sqrt(weight mu))
  double tA[N * N], tAroof[NS * N];
  SS_UNROLL_FULL
  for (int ju = 0; ju < NREG; ++ju) {
    SS_UNROLL_FULL
    for (int s = 0; s < NS; ++s) {
      const int cc = ju * NS + s;
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) {
        double acc = 0.0;
        SS_UNROLL_FULL
        for (int jl = 0; jl < NREG; ++jl) acc += Abe[sy<N>(i, jl * NS + s)] * SS_E(Vs, jl, ju, DB);
        SS_E(tA, i, cc, N) = acc;
      }
      SS_UNROLL_FULL
      for (int sr = 0; sr < NS; ++sr)
        SS_E(tAroof, sr, cc, NS) = URBAN ? (roof_a * lg.hweight[sr] * lg.rsqrt_wmu[sr]) * (SS_E(Vs, DB - 1, ju, DB) * lg.sqrt_wmu[s]) : 0.0;
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
  // tD = d_below^ V (direct-beam columns unscaled): canopy rows and roof rows
  double tD[N * D], tDroof[NS * D];
  SS_UNROLL_FULL
  for (int ju = 0; ju < NREG; ++ju) {
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) {
      double acc = 0.0;
      SS_UNROLL_FULL
      for (int jl = 0; jl < NREG; ++jl) acc += SS_E(Dbe, i, jl, N) * SS_E(V, jl, ju, DB);
      SS_E(tD, i, ju, N) = acc;
    }
    SS_UNROLL_FULL
    for (int sr = 0; sr < NS; ++sr)
      SS_E(tDroof, sr, ju, NS) = URBAN ? (roof_d * lg.hweight[sr] * lg.rsqrt_wmu[sr]) * SS_E(V, DB - 1, ju, DB) : 0.0;
  }
  apply_U_mat<NREG, DB, NS, D>(Us, tD, tDroof, Dab);
}

// top-of-canopy albedos (forest :520-528, urban :666-674): the clear-sky region of the free atmosphere
// (fraction 1: its scaling is sqrt(weight mu))
template <int NREG, int NS>
SS_HD void sw_top_albedos(const Launch& L, long cidx, double cos_sza, const double* Aab, const double* Dab) {
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
  L.top1[cidx] = s;
  s = 0.0;
  SS_UNROLL_FULL
  for (int j1 = 0; j1 < NS; ++j1) s += lg.sqrt_wmu[j1] * SS_E(Dab, j1, 0, N);
  L.top2[cidx] = s / cos_sza;
}

// The upward sweep of one column unit (k_column)
template <bool URBAN, int NREG, int NS>
SS_HD void column_up_sw(const Launch& L, int cu) {
  constexpr int N = NREG * NS, D = NREG;
  using LRec = SwLayRec<N, D>;
  using CRec = SwColRec<N, D>;
  const Inputs& in = L.in;
  const SwColCtx<URBAN, NREG, NS> c(L, cu, false);
  double Aab[kTri<N>], Dab[N * D];
  double sf[NREG], rsf[NREG];
  {
    double vf = 0.0, bf = 0.0, frac[NREG];
    if (c.nlay > 0) pack_fractions(pack_rec(L, 0, cu, kPackSw), vf, bf);
    region_fractions<URBAN, false, NREG>(vf, bf, frac);
    region_roots<NREG>(frac, sf, rsf);
  }
  sym::sw_ground<NREG, NS>(L.lg, in.ground_albedo[c.cidx], in.ground_albedo_dir[c.cidx], c.cos_sza, rsf, Aab, Dab);
  if constexpr (CRec::DIRVEC) sw_direct_tops<URBAN, NREG, NS>(L, cu, c.nlay, c.cos_sza);
  double R[rec_even(kTri<N>)], T[rec_even(kTri<N>)];
#if !defined(SS_NO_COLUMN_PREFETCH)
  if (c.nlay > 0) {
    const RecSource lrec{L.lay + rec_base(0, L.maxlay, LRec::NELEM, cu), kRecStride};
    lrec.template load<rec_even(kTri<N>)>(LRec::R, R);
    lrec.template load<rec_even(kTri<N>)>(LRec::T, T);
  }
#endif
  for (int jlay = 0; jlay < c.nlay; ++jlay) {
    const RecSource lrec{L.lay + rec_base(jlay, L.maxlay, LRec::NELEM, cu), kRecStride};
    const RecSource lnext{L.lay + rec_base(jlay + 1, L.maxlay, LRec::NELEM, cu), kRecStride};
    double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    SwOverlapIn oi;
    double dtop[rec_even(D)] = {0.0};
    if constexpr (CRec::DIRVEC) rec_load<D>(crec, kRecStride, CRec::DTP, dtop);
#if defined(SS_NO_COLUMN_PREFETCH)
    lrec.template load<rec_even(kTri<N>)>(LRec::R, R);
    lrec.template load<rec_even(kTri<N>)>(LRec::T, T);
#else
    oi.template load<URBAN>(L, cu, jlay, c.nlay);
#endif
    double dsc[N], rd[N];
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) {
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) {
        dsc[s + r * NS] = sf[r] * L.lg.sqrt_wmu[s];
        rd[s + r * NS] = rsf[r] * L.lg.rsqrt_wmu[s];
      }
    }
    double Abe[kTri<N>], Dbe[N * D];
    sym::sw_adding_step<N, D>(lrec, jlay + 1 < c.nlay ? &lnext : nullptr, R, T, Aab, Dab, crec, dsc, rd, Abe, Dbe, dtop);
#if defined(SS_NO_COLUMN_PREFETCH)
    oi.template load<URBAN>(L, cu, jlay, c.nlay);
#endif
    double sfu[NREG], rsfu[NREG];
    sym::sw_overlap_above<URBAN, NREG, NS>(L, oi, jlay, c.nlay, c.cos_sza, Abe, Dbe, sf, rsf, sfu, rsfu, Aab, Dab);
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) { sf[r] = sfu[r]; rsf[r] = rsfu[r]; }
  }
  sym::sw_top_albedos<NREG, NS>(L, c.cidx, c.cos_sza, Aab, Dab);
}

// Layer matrices AND adding step of every layer of a column unit on one lane (k_layer_up), both in the scaled
// coordinates: the layer unit writes R^, T^ (2 x 22 slots) and S_up^, S_dn^ (2 x 18) into the lane's 80 LDS
// slots - the whole packed layer record up to E -, E stays in registers and goes, with int_dir, to the column
// record for the downward sweep; the adding step reads them back from LDS element by element.  No layer
// record exists.  a_above^ / d_above^ are carried across the layer stage by the compiler (VGPRs, or AGPRs as
// first-level spill space at one wave per SIMD).
template <bool URBAN, int NREG, int NS>
SS_HD void layer_up_sw(const Launch& L, int cu) {
  constexpr int N = NREG * NS, D = NREG;
  using LRec = SwLayRec<N, D>;
  using CRec = SwColRec<N, D>;
  static_assert(CRec::FUSED && LRec::PACKED && LRec::EDIR <= kLaneLdsSlots, "R^, T^, S_up^, S_dn^ must fit the lane's LDS slots");
  using Lds = LaneLds<true>;
  const Inputs& in = L.in;
  const SwColCtx<URBAN, NREG, NS> c(L, cu, false);
  const Lds lds{};
  double Aab[kTri<N>], Dab[N * D];
  double sf[NREG], rsf[NREG];
  {
    double vf = 0.0, bf = 0.0, frac[NREG];
    if (c.nlay > 0) pack_fractions(pack_rec(L, 0, cu, kPackSw), vf, bf);
    region_fractions<URBAN, false, NREG>(vf, bf, frac);
    region_roots<NREG>(frac, sf, rsf);
  }
  sym::sw_ground<NREG, NS>(L.lg, in.ground_albedo[c.cidx], in.ground_albedo_dir[c.cidx], c.cos_sza, rsf, Aab, Dab);
  for (int jlay = 0; jlay < c.nlay; ++jlay) {
    double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    double hold[LRec::IDIR - LRec::EDIR];       // E
    const FusedWriter<LRec::EDIR, LRec::IDIR, LRec::EDIR, CRec::EDIR, Lds> w{
        &lds, hold, crec, kRecStride, (unsigned long long*)L.eigen_fail};
    layer_unit_sw<URBAN, NREG, NS>(L, cu, jlay, w);
    double dsc[N], rd[N];
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) {
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) {
        dsc[s + r * NS] = sf[r] * L.lg.sqrt_wmu[s];
        rd[s + r * NS] = rsf[r] * L.lg.rsqrt_wmu[s];
      }
    }
    const FusedSource<LRec::EDIR, Lds> src{&lds, hold};
    double Abe[kTri<N>], Dbe[N * D];
    double R[rec_even(kTri<N>)], T[rec_even(kTri<N>)];
    src.template load<rec_even(kTri<N>)>(LRec::R, R);
    src.template load<rec_even(kTri<N>)>(LRec::T, T);
    sym::sw_adding_step<N, D>(src, (const decltype(src)*)nullptr, R, T, Aab, Dab, crec, dsc, rd, Abe, Dbe);
    SwOverlapIn oi;
    oi.template load<URBAN>(L, cu, jlay, c.nlay);
    double sfu[NREG], rsfu[NREG];
    sym::sw_overlap_above<URBAN, NREG, NS>(L, oi, jlay, c.nlay, c.cos_sza, Abe, Dbe, sf, rsf, sfu, rsfu, Aab, Dab);
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) { sf[r] = sfu[r]; rsf[r] = rsfu[r]; }
  }
  sym::sw_top_albedos<NREG, NS>(L, c.cidx, c.cos_sza, Aab, Dab);
}

}  // namespace sym

// The upward sweep of one column unit (k_column): in symmetrised coordinates where the layer record is packed
template <bool URBAN, int NREG, int NS>
SS_HD void column_up_sw(const Launch& L, int cu) {
  if constexpr (SwLayRec<NREG * NS, NREG>::PACKED) sym::column_up_sw<URBAN, NREG, NS>(L, cu);
  else column_up_sw_full<URBAN, NREG, NS>(L, cu);
}

}  // namespace ss
#include "ss_layer_up_sw2.hpp"
namespace ss {

// Whole column on one lane (host emulation and small problems)
// layer + upward sweep on one lane (k_layer_up): scaled coordinates where the layer record is packed
template <bool URBAN, int NREG, int NS>
SS_HD void layer_up_sw_any(const Launch& L, int cu) {
  if constexpr (SwLayRec<NREG * NS, NREG>::PACKED) {
    if constexpr (SwColRec<NREG * NS, NREG>::DIRVEC) sym::layer_up_sw2<URBAN, NREG, NS>(L, cu);
    else sym::layer_up_sw<URBAN, NREG, NS>(L, cu);
  } else {
    layer_up_sw<URBAN, NREG, NS>(L, cu);
  }
}

template <bool URBAN, int NREG, int NS>
SS_HD void column_unit_sw(const Launch& L, int cu) {
  if constexpr (SwColRec<NREG * NS, NREG>::FUSED) layer_up_sw_any<URBAN, NREG, NS>(L, cu);
  else column_up_sw<URBAN, NREG, NS>(L, cu);
  if constexpr (sw_down_both(NREG * NS)) {
    column_down_sw_both<URBAN, NREG, NS>(L, cu);
  } else {
    column_down_sw<URBAN, NREG, NS>(L, cu, 0);
    column_down_sw<URBAN, NREG, NS>(L, cu, 1);
  }
}

}  // namespace ss
