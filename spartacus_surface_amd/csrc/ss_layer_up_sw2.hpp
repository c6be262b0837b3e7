// ss_layer_up_sw2.hpp - layer matrices AND adding method of every layer of a shortwave column unit on one lane
// (k_layer_up, -DSS_SYM_FUSE_SW2), second form.  No layer record exists: 17 KB per column of the 66 KB the
// step moves (DESIGN.md section 7.0 a).
//
// The first fused form in symmetrised coordinates (sym::layer_up_sw, -DSS_SYM_FUSE_SW) is correct and slow - 9.5 ms
// against 3.3 + 2.9 as two kernels - because the adding method's loop-carried state (a_above^ 21 + d_above^ 18
// doubles) has to live through the layer stage, which fills the register file on its own: 670 spilled dwords.
// What is different here:
//  * the carried state does not stay in registers: it waits in LDS slots while the layer stage runs.  The
//    stage's own scratch (36 slots of the eigen stage) and, later, its results R^, T^ (2 x 22) share the slots
//    0 .. 43; a_above^ (22) and the carried direct-beam vector (6) sit at 44 .. 71 of the lane's 80;
//  * the direct beam is a VECTOR throughout.  Of the blocks d_above (N x D), S_up, S_dn (N x D) the solution
//    only ever needs the products with ONE vector - the direct flux at the top of the layer for a unit beam at
//    canopy top, which follows from the direct transmittances and the overlap rules alone (top down, before
//    anything else: sw_direct_tops2).  The layer stage contracts S_up, S_dn with it as they are produced
//    (FuseWriter2), the adding step solves for N-vectors where it solved for N x D blocks, and the carried
//    d_above^ shrinks from 18 to 6 doubles:
//        d_above(l) (E_l dir_top(l)) = U [ d_below(l-1) dir_top(l-1) ; roof rows x direct flux on the roofs ]
//    because V E_l dir_top(l) IS dir_top(l-1) (and the roof component) by the recursion that defines them.
// Reference lines: the layer matrices radtool_calc_matrices_sw_eig.F90:180-296, the adding method
// radsurf_forest_sw.F90:463-528 / radsurf_urban_sw.F90:591-674.
#pragma once

#include "ss_column_sw_sym.hpp"

namespace ss {
namespace sym {

// LDS map of the fused kernel (doubles per lane)
template <int N>
struct Fuse2Map {
  static constexpr int RT = 0;                                   // R^ at 0, T^ at rec_even(tri): = SwLayRec offsets
  static constexpr int A = 2 * rec_even(kTri<N>);                // a_above^, packed
  static constexpr int DV = A + rec_even(kTri<N>);               // d_above^ (E dir_top), N
  static constexpr int DG = DV + N;                              // direct flux at the ground, D (first stage)
  static constexpr int DT = DG + 4;                              // direct flux at the top of the current layer, D + 1
  static constexpr int END = DT + 4;
  static_assert(N * N <= A, "the eigen stage's scratch fits below the carried state");
  static_assert(END <= kLaneLdsSlots, "LDS slots of a lane");
};

// Writer of the layer stage: R^, T^ -> LDS; S_up, S_dn contracted with the direct flux at the layer top (LDS slots
// DT: nothing the stage does not produce itself stays in registers while it runs) -> su, sd (N each, accumulated:
// zero them first); E and int_dir -> column record
template <int N, int D, class Lds>
struct FuseWriter2 {
  using LRec = SwLayRec<N, D>;
  using CRec = SwColRec<N, D>;
  const Lds* lds;
  double* su;
  double* sd;
  double* crec;
  long stride;
  unsigned long long* stamps;
  SS_HD void put1(int e, double v) const {
    if (e < LRec::SUP) {
      lds->put(Fuse2Map<N>::RT + e, v);
    } else if (e < LRec::SDN) {
      const int k = e - LRec::SUP;
      if (k < N * D) su[k % N] += v * lds->get(Fuse2Map<N>::DT + k / N);
    } else if (e < LRec::EDIR) {
      const int k = e - LRec::SDN;
      if (k < N * D) sd[k % N] += v * lds->get(Fuse2Map<N>::DT + k / N);
    }
  }
  SS_HD void put2(int e, double v0, double v1) const {
    put1(e, v0);
    put1(e + 1, v1);
    if (e >= LRec::EDIR) rec_pair_store(crec + (long)(CRec::EDIR + e - LRec::EDIR) * stride, v0, v1);
  }
  SS_HD const Lds& scratch() const { return *lds; }   // slots 0 .. N N - 1: free until R^ and T^ are put
};

// E of one layer from its input pack, with the 3-way branch of layer_unit_sw (clear-sky only / vegetated only /
// full): what the top-down recursion of the direct beam needs before any layer matrix exists
template <int D>
struct CaptureE {
  double* E;
  int edir;
  unsigned long long* stamps;
  SS_HD void put2(int e, double v0, double v1) const {
    if (e >= edir && e < edir + D * D) {
      E[e - edir] = v0;
      if (e + 1 - edir < D * D) E[e + 1 - edir] = v1;
    }
  }
};
template <bool URBAN, int NREG, int NS>
SS_HD void layer_direct_transmittance(const Launch& L, const double* pk, double cos_in, double* E) {
  constexpr int N = NREG * NS, D = NREG;
  using LRec = SwLayRec<N, D>;
  SwLayerProps<URBAN, NREG> p;
  sw_layer_props<URBAN, NREG>(L, cos_in, pk, p);
  double g0[D * D], dsc0[D];
  sw_gamma0<URBAN, NREG>(p, g0);
  SS_UNROLL_FULL
  for (int r = 0; r < D; ++r) dsc0[r] = sqrt(p.frac[r]);
  SS_UNROLL_FULL
  for (int k = 0; k < D * D; ++k) E[k] = 0.0;
  const CaptureE<D> w{E, LRec::EDIR, nullptr};
  double mu[D], G0[D * D], G0inv[D * D], emu[D];
  if (NREG == 1 || p.g.veg_fraction <= L.cfg.min_vegetation_fraction) {
    double s0[1];
    s0[0] = g0[0];
    sw_direct_embedded<N, D, 1, 0, LRec>(s0, dsc0, p.g.dz, w, mu, G0, G0inv, emu);
  } else if (p.frac[0] <= L.cfg.min_vegetation_fraction) {
    constexpr int DV = (NREG > 1) ? NREG - 1 : 1;
    constexpr int RV = (NREG > 1) ? 1 : 0;
    double s0[DV * DV];
    get_block<D, DV, DV, RV, RV>(g0, s0);
    sw_direct_embedded<N, D, DV, RV, LRec>(s0, dsc0 + RV, p.g.dz, w, mu, G0, G0inv, emu);
  } else {
    sw_direct_embedded<N, D, D, 0, LRec>(g0, dsc0, p.g.dz, w, mu, G0, G0inv, emu);
  }
}

// First stage: the direct flux just below the top interface of every layer (its regions and, urban, the exposed
// roofs) for a unit direct beam on the horizontal at canopy top, top down, into element DTP of the column records
template <bool URBAN, int NREG, int NS, class Lds>
SS_HD void sw_direct_tops2(const Launch& L, int cu, int nlay, double cos_in, double cos_sza, const Lds& lds) {
  constexpr int N = NREG * NS, D = NREG;
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  using CRec = SwColRec<N, D>;
  constexpr long st = kRecStride;
  double dir_ab[D];
  dir_ab[0] = 1.0 / cos_sza;
  SS_UNROLL_FULL
  for (int r = 1; r < D; ++r) dir_ab[r] = 0.0;
  double vf_up = 0.0, bf_up = 0.0;
  for (int jl = nlay - 1; jl >= 0; --jl) {
    double pk[kPackSw];
    rec_load<kPackSw>(pack_rec(L, jl, cu, kPackSw), st, 0, pk);
    const double vf_lo = pk[1], bf_lo = pk[5];
    double U[NREG * DB], V[DB * NREG];
    interface_overlap_vals<URBAN, false, NREG>(L.cfg.min_vegetation_fraction, jl + 1 >= nlay, vf_up, bf_up, false,
                                               vf_lo, bf_lo, U, V);
    double db[rec_even(D + 1)];
    SS_UNROLL_FULL
    for (int k = 0; k < rec_even(D + 1); ++k) db[k] = 0.0;
    SS_UNROLL_FULL
    for (int k = 0; k < DB; ++k) {
      double a = 0.0;
      SS_UNROLL_FULL
      for (int ju = 0; ju < NREG; ++ju) a += SS_E(V, k, ju, DB) * dir_ab[ju];
      db[k] = a;
    }
    rec_store<rec_even(D + 1)>(L.col + rec_base(jl, L.maxlay, CRec::NELEM, cu), st, CRec::DTP, db);
    double E[D * D];
    layer_direct_transmittance<URBAN, NREG, NS>(L, pk, cos_in, E);
    SS_UNROLL_FULL
    for (int r = 0; r < D; ++r) {
      double a = 0.0;
      SS_UNROLL_FULL
      for (int k = 0; k < D; ++k) a += SS_E(E, r, k, D) * db[k];
      dir_ab[r] = a;
    }
    vf_up = vf_lo;
    bf_up = bf_lo;
  }
  // the direct flux at the ground (the adding step of the lowest layer starts from d_above of the ground x this)
  SS_UNROLL_FULL
  for (int r = 0; r < D; ++r) lds.put(Fuse2Map<N>::DG + r, dir_ab[r]);
}

// The adding step of one layer with the direct beam as a vector.  R^, T^ in the lane's LDS slots (src), a_above^
// (Aab, packed) and dab = d_above^ (E dir_top) in registers, su / sd = S_up^ dir_top, S_dn^ dir_top (rows scaled).
// Stores P, Q dir_top, a_below^ (packed, scaled), d_below dir_top into the column record; returns a_below^ and
// d_below^ dir_top (rows scaled).
template <int N, int D, class Lds>
SS_HD void sw_adding_step_vec(const Lds& lds, const double* Aab, const double* dab, const double* su, const double* sd,
                              double* crec, const double* dsc, const double* rd, double* Abe, double* dbe) {
  using CRec = SwColRec<N, D>;
  static_assert(CRec::PACKED && CRec::DIRVEC, "packed column record with the direct beam as a vector");
  constexpr long st = kRecStride;
  constexpr int TOFF = rec_even(kTri<N>);
  double R[kTri<N>], T[kTri<N>];
  SS_UNROLL_FULL
  for (int k = 0; k < kTri<N>; ++k) {
    R[k] = lds.get(Fuse2Map<N>::RT + k);
    T[k] = lds.get(Fuse2Map<N>::RT + TOFF + k);
  }
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
  static_loop<0, rec_even(kTri<N>) / 2>([&](auto pc) {
    constexpr int e = 2 * decltype(pc)::value;
    rec_pair_store(crec + (long)(CRec::ABE + e) * st, Abe[e], (e + 1 < kTri<N>) ? Abe[e + 1 < kTri<N> ? e + 1 : 0] : 0.0);
  });
  // Q^ dir_top = Den^-1 (R^ dab + sd);  d_below^ dir_top = su + T^ Den^-1 (dab + a_above^ sd)
  double q[N], y[N];
  SS_UNROLL_FULL
  for (int i = 0; i < N; ++i) {
    double a = sd[i], b = dab[i];
    SS_UNROLL_FULL
    for (int k = 0; k < N; ++k) {
      a += R[sy<N>(i, k)] * dab[k];
      b += Aab[sy<N>(i, k)] * sd[k];
    }
    q[i] = a;
    y[i] = b;
  }
  lu_solve<N>(Den, q);
  lu_solve<N>(Den, y);
  double t[N];
  SS_UNROLL_FULL
  for (int i = 0; i < N; ++i) {
    double a = su[i];
    SS_UNROLL_FULL
    for (int k = 0; k < N; ++k) a += T[sy<N>(i, k)] * y[k];
    dbe[i] = a;
    q[i] *= dsc[i];
    t[i] = dsc[i] * a;
  }
  rec_store<N>(crec, st, CRec::QDN, q);
  rec_store<N>(crec, st, CRec::DBE, t);
}

// Overlap to the base of the layer above (or to the free atmosphere) with the direct beam as a vector: the matrix
// part is sw_overlap_above's, the vector is U^ [d_below^ dir_top ; roof rows x direct flux on the exposed roofs]
template <bool URBAN, int NREG, int NS>
SS_HD void sw_overlap_above_vec(const Launch& L, const SwOverlapIn& oi, int jlay, int nlay, double cos_sza,
                                const double* Abe, const double* dbe, double dir_roof, const double* sf_lo,
                                const double* rsf_lo, double* sf_up, double* rsf_up, double* Aab, double* dab) {
  constexpr int N = NREG * NS;
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  const LG& lg = L.lg;
  double U[NREG * DB], V[DB * NREG];
  const bool sky = jlay + 1 >= nlay;
  interface_overlap_vals<URBAN, false, NREG>(L.cfg.min_vegetation_fraction, sky, oi.vf_up, oi.bf_up, false, oi.vf_lo,
                                             oi.bf_lo, U, V);
  {
    double fu[NREG];
    if (sky) {
      fu[0] = 1.0;
      SS_UNROLL_FULL
      for (int r = 1; r < NREG; ++r) fu[r] = 0.0;
    } else {
      region_fractions<URBAN, false, NREG>((NREG > 1) ? oi.vf_up : 0.0, URBAN ? oi.bf_up : 0.0, fu);
    }
    region_roots<NREG>(fu, sf_up, rsf_up);
  }
  const double roof_a = URBAN ? oi.roof_albedo : 0.0, roof_d = URBAN ? cos_sza * oi.roof_albedo_dir : 0.0;
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
  double droof[NS];
  SS_UNROLL_FULL
  for (int sr = 0; sr < NS; ++sr) droof[sr] = URBAN ? (roof_d * lg.hweight[sr] * lg.rsqrt_wmu[sr]) * dir_roof : 0.0;
  apply_U_mat<NREG, DB, NS, 1>(Us, dbe, droof, dab);
}

// top-of-canopy albedos (forest :520-528, urban :666-674) from a_above^ and d_above^ x (direct beam at canopy top)
template <int NREG, int NS>
SS_HD void sw_top_albedos_vec(const Launch& L, long cidx, const double* Aab, const double* dab) {
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
  for (int j1 = 0; j1 < NS; ++j1) s += lg.sqrt_wmu[j1] * dab[j1];
  L.top2[cidx] = s;
}

// The fused kernel body: one column unit
template <bool URBAN, int NREG, int NS>
SS_HD void layer_up_sw2(const Launch& L, int cu) {
  constexpr int N = NREG * NS, D = NREG;
  using LRec = SwLayRec<N, D>;
  using CRec = SwColRec<N, D>;
  static_assert(CRec::FUSED && CRec::DIRVEC && LRec::PACKED, "fused, packed, direct beam as a vector");
  using Lds = LaneLds<true>;
  using Map = Fuse2Map<N>;
  const Inputs& in = L.in;
  const LG& lg = L.lg;
  const SwColCtx<URBAN, NREG, NS> c(L, cu, false);
  const double cos_in = in.cos_sza[c.col];
  const Lds lds{};
  constexpr long st = kRecStride;
  // direct flux below the top interface of every layer (element DTP of the column records)
#if !defined(SS_F2_NO_PRE)       // (diagnostic builds: which stage owns the register pressure)
  sw_direct_tops2<URBAN, NREG, NS>(L, cu, c.nlay, cos_in, c.cos_sza, lds);
#endif

  // ground (forest :466-477, urban :594-606): a_above^ in the scaling of the lowest layer -> LDS; the direct part
  // needs the direct flux at the ground, which the first stage left in LDS
  const double ga = in.ground_albedo[c.cidx], gad = in.ground_albedo_dir[c.cidx];
  {
    double sf[NREG], rsf[NREG];
    double vf = 0.0, bf = 0.0, frac[NREG];
    if (c.nlay > 0) pack_fractions(pack_rec(L, 0, cu, kPackSw), vf, bf);
    region_fractions<URBAN, false, NREG>(vf, bf, frac);
    region_roots<NREG>(frac, sf, rsf);
    double Aab[kTri<N>], Dab[N * D];
    sw_ground<NREG, NS>(lg, ga, gad, c.cos_sza, rsf, Aab, Dab);
    SS_UNROLL_FULL
    for (int k = 0; k < kTri<N>; ++k) lds.put(Map::A + k, Aab[k]);
    // d_above^ of the ground x the direct flux at the ground
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) {
      double a = 0.0;
      SS_UNROLL_FULL
      for (int r = 0; r < D; ++r) a += SS_E(Dab, i, r, N) * (c.nlay > 0 ? lds.get(Map::DG + r) : (r == 0 ? 1.0 / c.cos_sza : 0.0));
      lds.put(Map::DV + i, a);
    }
    if (c.nlay == 0) {
      // no layer: the canopy-top albedos are the ground's (direct beam 1 / cos_sza in the clear region)
      double dab[N];
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) dab[i] = lds.get(Map::DV + i);
      sw_top_albedos_vec<NREG, NS>(L, c.cidx, Aab, dab);
      return;
    }
  }
  for (int jlay = 0; jlay < c.nlay; ++jlay) {
    // (what the step reads of the launch descriptor it reads HERE: left to itself the compiler hoists every
    // product of Legendre-Gauss weights and configuration scalars out of this loop and spills them)
    SS_LAUNCH_HERE_REF(L, Lq);
    const LG& lg = Lq.lg;
    double* crec = Lq.col + rec_base(jlay, Lq.maxlay, CRec::NELEM, cu);
    // the direct flux at the top of this layer (regions) and on its exposed roofs -> LDS
    {
      double dtop[rec_even(D + 1)];
      rec_load<rec_even(D + 1)>(crec, st, CRec::DTP, dtop);
      SS_UNROLL_FULL
      for (int k = 0; k < rec_even(D + 1); ++k) lds.put(Map::DT + k, dtop[k]);
    }
    // ---- layer stage: R^, T^ -> LDS, S_up^ dir_top, S_dn^ dir_top -> su, sd, E and int_dir -> column record ----
    double su[N], sd[N];
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) { su[i] = 0.0; sd[i] = 0.0; }
    {
#if defined(SS_F2_RECWRITER)     // (diagnostic: the stand-alone kernel's writer inside this loop)
      const RecWriter w{Lq.col + rec_base(jlay, Lq.maxlay, LRec::NELEM, cu), kRecStride, (unsigned long long*)Lq.eigen_fail};
#else
      const FuseWriter2<N, D, Lds> w{&lds, su, sd, crec, st, (unsigned long long*)Lq.eigen_fail};
#endif
      layer_unit_sw<URBAN, NREG, NS>(Lq, cu, jlay, w);
    }
#if defined(SS_F2_NO_ADD)
    if (su[0] + sd[1] == 1.2345) lds.put(Map::DV, su[2]);
    continue;
#endif
    // ---- adding step: the carried state comes back from LDS, the scaling of the layer from its fractions ----
    SwOverlapIn oi;
    oi.template load<URBAN>(Lq, cu, jlay, c.nlay);
    double sf[NREG], rsf[NREG];
    {
      double frac[NREG];
      region_fractions<URBAN, false, NREG>(oi.vf_lo, oi.bf_lo, frac);
      region_roots<NREG>(frac, sf, rsf);
    }
    double dsc[N], rd[N];
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) {
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) {
        dsc[s + r * NS] = sf[r] * lg.sqrt_wmu[s];
        rd[s + r * NS] = rsf[r] * lg.rsqrt_wmu[s];
      }
    }
    double Aab[kTri<N>], dab[N];
    SS_UNROLL_FULL
    for (int k = 0; k < kTri<N>; ++k) Aab[k] = lds.get(Map::A + k);
    SS_UNROLL_FULL
    for (int i = 0; i < N; ++i) dab[i] = lds.get(Map::DV + i);
    double Abe[kTri<N>], dbe[N];
    sw_adding_step_vec<N, D>(lds, Aab, dab, su, sd, crec, dsc, rd, Abe, dbe);
    // ---- overlap to the base of the layer above; the new carried state goes to LDS ----
    double sfu[NREG], rsfu[NREG];
    sw_overlap_above_vec<URBAN, NREG, NS>(Lq, oi, jlay, c.nlay, c.cos_sza, Abe, dbe, URBAN ? lds.get(Map::DT + D) : 0.0, sf, rsf,
                                          sfu, rsfu, Aab, dab);
    if (jlay + 1 < c.nlay) {
      SS_UNROLL_FULL
      for (int k = 0; k < kTri<N>; ++k) lds.put(Map::A + k, Aab[k]);
      SS_UNROLL_FULL
      for (int i = 0; i < N; ++i) lds.put(Map::DV + i, dab[i]);
    } else {
      sw_top_albedos_vec<NREG, NS>(Lq, c.cidx, Aab, dab);
    }
  }
}

}  // namespace sym
}  // namespace ss
