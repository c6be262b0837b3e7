// ss_down_sw_both.hpp - the shortwave downward sweep with BOTH flux passes of a (column, band) unit on one lane
// (section 5 of spartacus_forest_sw / spartacus_urban_sw: the pass normalised by the direct flux at canopy top,
// radsurf_forest_sw.F90:530-676 / radsurf_urban_sw.F90:676-874, and the pass normalised by the diffuse flux,
// forest :684-760 / urban :876-984).
//
// Rounds 1-3 ran the two passes on paired lanes (column_down_sw, ss_column_sw.hpp): one instruction stream, the
// same memory requests.  But most of what a layer step computes does not depend on the pass - the overlap
// matrices of the interface, the scaling of a_below^, region optics, Gamma1 + Gamma2, Gamma3 and the LU factors of
// the integrated-flux solve: ~1650 of the ~2135 VALU instructions per layer - and the sweep is bound by the
// dependent record loads of a layer at one wave per SIMD, which a wave then paid for 32 columns.  Here a lane
// carries both flux sets: everything pass-independent once, every record element fetched once per column and
// applied to two vectors (twice the independent arithmetic behind each load), and the pass without a direct
// beam compiled without its identically-zero terms.  Per pass the expressions are those of column_down_sw, in
// the same order, so results are equal to it bit for bit under a compiler that fuses nothing differently
// (the host emulation asserts that; on the device to rounding).
//
// Per-layer outputs: 10 + 7 fields per layer do not fit the 80 LDS slots of a lane at 10 layers, so the
// fields are staged for a GROUP of layers (16 slots x 5 layers for vegetated urban tiles, 7 x 11 for forests) and
// written out - every field's layers of the group back to back, 40 bytes of a line at a time - whenever the group
// is full; wall_net = wall_in (1 - wall_albedo) is formed at that point from the staged wall_in of the pass and
// one staged 1 - wall_albedo instead of being staged twice.
//
// Loads: the sweep runs one wave per SIMD, so every record load a layer step waits for is a memory round trip
// (~2 us under load) nobody hides.  E, Q and P of the next layer down are requested when the integrated-flux
// solve of this layer is done (its 54-double working set is dead by then), a_below, d_below of the layer below,
// int_dir and the layer's input pack at the top of the step - their first use comes after the P products -:
// no step requests something and uses it right away.  What the loop reads of the launch descriptor it reads
// through SS_LAUNCH_HERE_REF inside the step: left to itself the compiler hoists every product of Legendre-Gauss
// weights out of the loop, runs out of registers and reloads them from scratch memory in the middle of the
// step, each reload a round trip that also waits for every record load in flight.
#pragma once

#include "ss_column_sw.hpp"

namespace ss {

// Off by default: measured on C3 (10^6 columns, one stream) the sweep itself runs in 2.6 ms against ~2.9 for the
// paired-lane form, but its per-layer outputs leave in half-column groups - lines written half now, half 50 us
// later, evicted in between: 12 KB per column of write traffic against 3.7 - and the kernel takes 4.5 ms against
// 3.55 (DESIGN.md section 7.0 d).  -DSS_DOWN_BOTH builds it; the host emulation checks it against the paired form.
constexpr bool sw_down_both(int n) {
#if defined(SS_DOWN_BOTH)
  return n <= SS_UNROLL_MAX;
#else
  return false && n > 0;
#endif
}

// Staging plan: (pass, field) pairs in slot order; a pair is staged if its field exists for the tile kind
// (vegetation fields: NREG > 1, roof / wall fields: urban tiles); every other per-layer field of the sweep is
// zero (or, wall_net, derived) when the group is written out.
struct Sw2Stage { int pass, field; };
constexpr Sw2Stage kSw2Cand[] = {
    {0, F_CLEAR_AIR_ABS}, {0, F_VEG_ABS}, {0, F_VEG_AIR_ABS}, {0, F_VEG_ABS_DIR}, {0, F_WALL_IN}, {0, F_WALL_IN_DIR},
    {0, F_ROOF_IN}, {0, F_ROOF_IN_DIR}, {0, F_ROOF_NET},
    {1, F_CLEAR_AIR_ABS}, {1, F_VEG_ABS}, {1, F_VEG_AIR_ABS}, {1, F_WALL_IN}, {1, F_ROOF_IN}, {1, F_ROOF_NET},
    {2, F_WALL_NET}};    // pass "2": 1 - wall_albedo of the layer, shared by the wall_net of both passes
constexpr int kSw2NCand = (int)(sizeof(kSw2Cand) / sizeof(kSw2Cand[0]));
constexpr bool sw2_field_exists(bool urban, int nreg, int field) {
  switch (field) {
    case F_CLEAR_AIR_ABS: return true;
    case F_VEG_ABS: case F_VEG_AIR_ABS: case F_VEG_ABS_DIR: return nreg > 1;
    case F_WALL_IN: case F_WALL_IN_DIR: case F_WALL_NET: case F_ROOF_IN: case F_ROOF_IN_DIR: case F_ROOF_NET: return urban;
    default: return false;
  }
}
// slot of (pass, field), -1 = not staged
constexpr int sw2_slot(bool urban, int nreg, int pass, int field) {
  int s = 0;
  for (int k = 0; k < kSw2NCand; ++k) {
    if (!sw2_field_exists(urban, nreg, kSw2Cand[k].field)) continue;
    if (kSw2Cand[k].pass == pass && kSw2Cand[k].field == field) return s;
    ++s;
  }
  return -1;
}
constexpr int sw2_nslots(bool urban, int nreg) {
  int s = 0;
  for (int k = 0; k < kSw2NCand; ++k) s += sw2_field_exists(urban, nreg, kSw2Cand[k].field) ? 1 : 0;
  return s;
}
// the per-layer members of the two flux objects this sweep owns (everything PUTL of column_down_sw writes)
constexpr int kSw2LayerFields[] = {F_CLEAR_AIR_ABS, F_VEG_ABS, F_VEG_AIR_ABS, F_VEG_ABS_DIR, F_WALL_IN, F_WALL_IN_DIR,
                                   F_WALL_NET, F_ROOF_IN, F_ROOF_IN_DIR, F_ROOF_NET};
constexpr int kSw2NLayerFields = (int)(sizeof(kSw2LayerFields) / sizeof(kSw2LayerFields[0]));

#ifndef SS_DB_ABE_AT
#define SS_DB_ABE_AT 2
#endif
#if defined(SS_EXP_NOSCHED)
#define SS_DB_SCHED() do {} while (0)
#else
#define SS_DB_SCHED() SS_SCHED_FENCE()
#endif
template <bool URBAN, int NREG, int NS>
SS_HD void column_down_sw_both(const Launch& L, int cu) {
  constexpr int N = NREG * NS, D = NREG;
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  constexpr int NB = DB * NS;
  using LRec = SwLayRec<N, D>;
  using CRec = SwColRec<N, D>;
  static_assert(sw_down_both(N), "orders whose vectors live in registers");
  const SwColCtx<URBAN, NREG, NS> c(L, cu, true);
  constexpr int NABE = CRec::PACKED ? rec_even(tri_count(N)) : N * N;
  const int col = c.col, jb = c.jb, nspec = c.nspec, nlay = c.nlay, l0 = c.l0, itransp = c.itransp;
  const long cidx = c.cidx;
  const double cos_sza = c.cos_sza, sin0 = c.sin0;
  // the unit's records: one buffer resource + one 32-bit lane offset per scratch family
  const RecView colv(L.col, L.maxlay, CRec::NELEM, cu), layv(L.lay, L.maxlay, LRec::NELEM, cu);
  const RecView packv(L.pack, L.pack_maxlay, kPackSw, L.pack_cu0 + cu);

  // flux objects: a = sw_norm_dir (pass 0), b = sw_norm_diff (pass 1); Lo: a descriptor read where it is used
#define PUTC0(F, v) do { double* p_ = Lo.flux_a.f[F]; if (p_) p_[cidx] = (v); } while (0)
#define PUTC1(F, v) do { double* p_ = Lo.flux_b.f[F]; if (p_) p_[cidx] = (v); } while (0)
  constexpr int NSLOT = sw2_nslots(URBAN, NREG);
  constexpr int G = kLaneLdsSlots / NSLOT;          // layers per staging group
  static_assert(G >= 1, "staging group");
  const LaneLds<true> obuf;
  int kst = 0;                                      // layers staged since the last write-out
#define STAGE(P, F, v) do { constexpr int s_ = sw2_slot(URBAN, NREG, P, F); \
    if constexpr (s_ >= 0) obuf.put(s_ * G + kst, (v)); } while (0)
  // profile members are stored as they are computed
#define PUTP0(F, v) do { double* p_ = Lo.flux_a.f[F]; if (p_) p_[sidx] = (v); } while (0)
#define PUTP1(F, v) do { double* p_ = Lo.flux_b.f[F]; if (p_) p_[sidx] = (v); } while (0)

  // write the staged layers jlo .. jlo + cnt - 1 out (the layer jlo + i was staged at position cnt - 1 - i)
  auto write_group = [&](int jlo, int cnt) {
    SS_LAUNCH_HERE(L, Lo);
    const long e0 = (long)jb + (long)nspec * (l0 + jlo);
    for (int i = 0; i < cnt; ++i) {
      const int k = cnt - 1 - i;
      const long e = e0 + (long)nspec * i;
      static_loop<0, 2 * kSw2NLayerFields>([&](auto kc) {
        constexpr int P = decltype(kc)::value / kSw2NLayerFields;
        constexpr int F = kSw2LayerFields[decltype(kc)::value % kSw2NLayerFields];
        constexpr int s_ = sw2_slot(URBAN, NREG, P, F);
        constexpr int sw_ = sw2_slot(URBAN, NREG, P, F_WALL_IN), so_ = sw2_slot(URBAN, NREG, 2, F_WALL_NET);
        double* p_ = (P == 0 ? Lo.flux_a.f[F] : Lo.flux_b.f[F]);
        if (p_) {
          double v_ = 0.0;
          if constexpr (F == F_WALL_NET) {
            if constexpr (URBAN) v_ = obuf.get(sw_ * G + k) * obuf.get(so_ * G + k);
          } else if constexpr (s_ >= 0) {
            v_ = obuf.get(s_ * G + k);
          }
          p_[e] = v_;
        }
      });
    }
  };

  {
    SS_LAUNCH_HERE(L, Lo);
    PUTC0(F_TOP_DN_DIR, 1.0);
    PUTC1(F_TOP_DN_DIR, 0.0);
    PUTC0(F_TOP_DN, 1.0);
    PUTC1(F_TOP_DN, 1.0);
    PUTC0(F_TOP_NET, 1.0 * (1.0 - Lo.top2[cidx]));
    PUTC1(F_TOP_NET, 1.0 - Lo.top1[cidx]);
  }
  double flux_dn_dir_clear = 1.0 / cos_sza;   // clear-sky direct beam of the direct pass

  double dir_below[DB], diff_below0[NB], up_below0[NB], diff_below1[NB], up_below1[NB];

  // Crossing the interface at the top of the lower layer "lo" (see column_down_sw): downward fluxes through V,
  // upward flux just below the interface from a_below / d_below (and the exposed roofs), upward flux just above
  // it through U - the overlap matrices, the scaling of a_below^ and the roof factors once for both passes.
  auto cross = [&](const Launch& Lq, const double* lo, bool upper_is_sky, double vf_up, double bf_up, const double* ABE,
                   const double* DBE, const double* dir_above, const double* diff_above0, const double* diff_above1,
                   double* up_above0, double* up_above1) {
    const LG& lg = Lq.lg;
    double U[NREG * DB], V[DB * NREG];
    interface_overlap_vals<URBAN, false, NREG>(Lq.cfg.min_vegetation_fraction, upper_is_sky, vf_up, bf_up, false, lo[0],
                                               lo[1], U, V);
    SS_UNROLL_FULL
    for (int jl = 0; jl < DB; ++jl) {
      double acc = 0.0;
      SS_UNROLL_FULL
      for (int ju = 0; ju < NREG; ++ju) acc += SS_E(V, jl, ju, DB) * dir_above[ju];
      dir_below[jl] = acc;
    }
    apply_V_vec<NREG, DB, NS>(V, diff_above0, diff_below0);
    apply_V_vec<NREG, DB, NS>(V, diff_above1, diff_below1);
    if constexpr (CRec::PACKED) {
      double frac_[NREG], dsc_[N], rd_[N];
      region_fractions<URBAN, false, NREG>(lo[0], lo[1], frac_);
      layer_scaling<NREG, NS>(lg, frac_, dsc_, rd_);
      packed_scaled_matvec<N>(ABE, dsc_, rd_, diff_below0, up_below0);
      packed_scaled_matvec<N>(ABE, dsc_, rd_, diff_below1, up_below1);
    } else {
      mat_vec<N, N>(ABE, diff_below0, up_below0);
      mat_vec<N, N>(ABE, diff_below1, up_below1);
    }
    if constexpr (CRec::DIRVEC) {
      // d_below dir_below, formed by the upward kernel
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) up_below0[i] = up_below0[i] + DBE[i];
    } else {
      mat_vec_acc<N, D>(DBE, dir_below, up_below0);
    }
    if (URBAN) {
      // exposed roof (urban :717-721): a_below / d_below roof blocks
      const double roof_a = lo[2];
      const double roof_d = cos_sza * lo[3];
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) {
        double acc0 = 0.0, acc1 = 0.0;
        SS_UNROLL_FULL
        for (int s2 = 0; s2 < NS; ++s2) {
          acc0 += (roof_a * lg.hweight[s]) * diff_below0[N + s2];
          acc1 += (roof_a * lg.hweight[s]) * diff_below1[N + s2];
        }
        up_below0[N + s] = acc0 + (roof_d * lg.hweight[s]) * dir_below[DB - 1];
        up_below1[N + s] = acc1;
      }
    }
    if (up_above0) {
      apply_U_mat<NREG, DB, NS, 1>(U, up_below0, up_below0 + N, up_above0);
      apply_U_mat<NREG, DB, NS, 1>(U, up_below1, up_below1 + N, up_above1);
    }
  };
  // the four scalars of a layer the interface at its top needs: veg_fraction, building_fraction, roof_albedo,
  // roof_albedo_dir
  auto load_lo = [&](int jl, double* lo) {
    lo[0] = packv.pair(jl, 0).y;
    lo[1] = packv.pair(jl, 4).y;
    const RecPair ra = packv.pair(jl, 8);
    lo[2] = ra.x;
    lo[3] = ra.y;
  };

  // ground (forest :661-676, urban :861-874).  Written by the layer step of the lowest layer, inside the loop,
  // as soon as the fluxes at the ground exist: lanes leave the loop at different trip counts, so whatever is
  // read after it has to stay in registers, for the lanes that have left, through every step of the others
  auto ground_out = [&](const double* dir_above, const double* diff_above0, const double* up_above0,
                        const double* diff_above1, const double* up_above1) -> double {
    SS_LAUNCH_HERE(L, Lo);
    double s = 0.0;
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) s += dir_above[r];
    const double gdd = cos_sza * s;
    const double s20 = vec_sum<N>(diff_above0), s30 = vec_sum<N>(up_above0);
    const double s21 = vec_sum<N>(diff_above1), s31 = vec_sum<N>(up_above1);
    PUTC0(F_GROUND_DN_DIR, gdd);
    PUTC0(F_GROUND_DN, gdd + s20);
    PUTC0(F_GROUND_NET, (gdd + s20) - s30);
    PUTC1(F_GROUND_DN_DIR, 0.0);
    PUTC1(F_GROUND_DN, 0.0 + s21);
    PUTC1(F_GROUND_NET, (0.0 + s21) - s31);
    double gvd0 = 0.0, gvd1 = 0.0;
    SS_UNROLL_FULL
    for (int jreg = 0; jreg < NREG; ++jreg) {
      SS_UNROLL_FULL
      for (int js = 0; js < NS; ++js) {
        const int ifr = js + jreg * NS;
        gvd0 = gvd0 + (diff_above0[ifr] + up_above0[ifr]) * Lo.lg.tan_ang[js] / kPi;
        gvd1 = gvd1 + (diff_above1[ifr] + up_above1[ifr]) * Lo.lg.tan_ang[js] / kPi;
      }
    }
    PUTC0(F_GROUND_VERTICAL_DIFF, gvd0);
    PUTC1(F_GROUND_VERTICAL_DIFF, gvd1);
    return gdd;
  };
  // ... and the sunlit fraction of the ground once the clear-sky beam has passed the lowest layer
  auto ground_sunlit = [&](double gdd, double flux_dn_dir_clear) {
    if (jb == itransp) {
      SS_LAUNCH_HERE(L, Lo);
      if (Lo.flux_a.f[F_GROUND_SUNLIT_FRAC]) Lo.flux_a.f[F_GROUND_SUNLIT_FRAC][col] = gdd / (cos_sza * flux_dn_dir_clear);
      if (Lo.flux_b.f[F_GROUND_SUNLIT_FRAC]) Lo.flux_b.f[F_GROUND_SUNLIT_FRAC][col] = 0.0;
    }
  };

  // records in flight across the layer steps: E, Q, P of the layer about to be entered
  double E[D * D], Q[CRec::NQ], P[N * N];
  double vf_cur = 0.0, bf_cur = 0.0;   // fractions of the layer about to be entered
  double bf_up = 0.0;                  // building fraction of the layer above (roof / sunlit bookkeeping)
  {
    // fluxes at canopy top: 0 = direct beam on (sw_norm_dir), 1 = diffuse illumination (sw_norm_diff)
    double dir_top[D], diff_top0[N], diff_top1[N];
    dir_top[0] = 1.0 / cos_sza;
    SS_UNROLL_FULL
    for (int r = 1; r < D; ++r) dir_top[r] = 0.0;
    vec_zero<N>(diff_top0);
    vec_zero<N>(diff_top1);
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) diff_top1[js] = L.lg.hweight[js];
    if (nlay > 0) {
      // top layer: a_below, d_below and its scalars, then the interface to the free atmosphere
      SS_LAUNCH_HERE_REF(L, Lq);
      const int jlay = nlay - 1;
      double ABE[NABE], DBE[CRec::NQ], lo[4];
      load_lo(jlay, lo);
      colv.template load<NABE>(jlay, CRec::ABE, ABE);
      colv.template load<CRec::NQ>(jlay, CRec::DBE, DBE);
#if defined(SS_DB_EQP_AHEAD)
      if (CRec::FUSED) colv.template load<D * D>(jlay, CRec::EDIR, E);
      else layv.template load<D * D>(jlay, LRec::EDIR, E);
      colv.template load<CRec::NQ>(jlay, CRec::QDN, Q);
      colv.template load<N * N>(jlay, CRec::PDN, P);
#endif
      SS_LOAD_FENCE(); SS_DB_SCHED();
      cross(Lq, lo, true, 0.0, 0.0, ABE, DBE, dir_top, diff_top0, diff_top1, nullptr, nullptr);
      vf_cur = lo[0];
      bf_cur = lo[1];
    } else {
      // a column without layers: the ground sees the canopy-top fluxes
      double up0[N], up1[N];
      vec_zero<N>(up0);
      vec_zero<N>(up1);
      ground_sunlit(ground_out(dir_top, diff_top0, up0, diff_top1, up1), flux_dn_dir_clear);
    }
  }

  for (int jlay = nlay - 1; jlay >= 0; --jlay) {
    SS_LAUNCH_HERE_REF(L, Lq);
    const LG& lg = Lq.lg;
    const int lay = l0 + jlay;
    const long sidx = (long)jb + (long)nspec * lay;

    double ABE[NABE], DBE[CRec::NQ];
    double lo[4] = {0.0, 0.0, 0.0, 0.0};
#define SS_DB_REQUEST_ABE() do { if (jlay > 0) { load_lo(jlay - 1, lo); \
      colv.template load<NABE>(jlay - 1, CRec::ABE, ABE); colv.template load<CRec::NQ>(jlay - 1, CRec::DBE, DBE); } } while (0)
    // ---- request: int_dir and the input pack of this layer - first used after the P products ----
    double IDIR[D * D];
    SwDownIn<URBAN, NREG> dcur;
    if (CRec::FUSED) colv.template load<D * D>(jlay, CRec::IDIR, IDIR);
    else layv.template load<D * D>(jlay, LRec::IDIR, IDIR);
    dcur.load_pack(packv, jlay);
#if !defined(SS_DB_EQP_AHEAD)
    if (CRec::FUSED) colv.template load<D * D>(jlay, CRec::EDIR, E);
    else layv.template load<D * D>(jlay, LRec::EDIR, E);
    colv.template load<CRec::NQ>(jlay, CRec::QDN, Q);
    colv.template load<N * N>(jlay, CRec::PDN, P);
#endif
#if SS_DB_ABE_AT == 1
    SS_DB_REQUEST_ABE();
#endif
    SS_LOAD_FENCE(); SS_DB_SCHED();

    // ---- the fluxes just below the top interface: exposed roofs ----
    double roof_in_dir = 0.0;
    if (URBAN) {
      double sdn0 = 0.0, sup0 = 0.0, sdn1 = 0.0, sup1 = 0.0;
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) {
        sdn0 += diff_below0[N + s];
        sup0 += up_below0[N + s];
        sdn1 += diff_below1[N + s];
        sup1 += up_below1[N + s];
      }
      roof_in_dir = cos_sza * dir_below[DB - 1];
      STAGE(0, F_ROOF_IN_DIR, roof_in_dir);
      STAGE(0, F_ROOF_IN, roof_in_dir + sdn0);
      STAGE(0, F_ROOF_NET, (roof_in_dir + sdn0) - sup0);
      STAGE(1, F_ROOF_IN, 0.0 + sdn1);
      STAGE(1, F_ROOF_NET, (0.0 + sdn1) - sup1);
    }

    // ---- E, Q (requested a step ago): the direct beam through the layer ----
    double dir_above[D], dir_conv[D];   // direct flux at the base; dir_below - dir_above
    {
      double t[D];
      mat_vec<D, D>(E, dir_below, t);
      SS_UNROLL_FULL
      for (int r = 0; r < D; ++r) { dir_above[r] = t[r]; dir_conv[r] = dir_below[r] - t[r]; }
    }
    double t3[N];
    if constexpr (CRec::DIRVEC) {
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) t3[i] = Q[i];     // Q dir_below, formed by the upward kernel
    } else {
      mat_vec<N, D>(Q, dir_below, t3);
    }

#if SS_DB_ABE_AT == 2
    SS_DB_REQUEST_ABE();
    SS_LOAD_FENCE(); SS_DB_SCHED();
#endif
    // ---- P: flux at the base of the layer, diff = P diff_top + Q dir_top ----
    double conv0[N], conv1[N];       // diff_below - up_below at the top, completed below
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) {
      conv0[i] = diff_below0[i] - up_below0[i];
      conv1[i] = diff_below1[i] - up_below1[i];
    }
    double s_top_dir = 0.0;
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) s_top_dir += dir_below[r];
    const double s_top_dn0 = vec_sum<N>(diff_below0), s_top_up0 = vec_sum<N>(up_below0);
    const double s_top_dn1 = vec_sum<N>(diff_below1), s_top_up1 = vec_sum<N>(up_below1);
    double diff_above0[N], diff_above1[N], up_above0[N], up_above1[N];
    mat_vec<N, N>(P, diff_below0, diff_above0);
    mat_vec<N, N>(P, diff_below1, diff_above1);
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) diff_above0[i] = diff_above0[i] + t3[i];

    // ---- request a_below, d_below and the scalars of the layer below (the interface at the base) into the
    // registers E, Q, P have left; their latency is covered by the part of the integrated-flux solve that does not
    // depend on the fluxes: Gamma1 + Gamma2, Gamma3 of the layer from its scalars and the LU factorisation
    // (forest :598-606: int_diff conv + int_dir_diff dir_conv = -(Gamma1 + Gamma2)^-1 (conv + 2 Gamma3 int_flux_dir),
    // see sw_int_flux_system) ----
#if SS_DB_ABE_AT == 0
    SS_DB_REQUEST_ABE();
    SS_LOAD_FENCE(); SS_DB_SCHED();
#endif
    const LayerGeom g = dcur.g;
#if defined(SS_DB_LU_EARLY)
    double B[N * N], g3[N * D];
    bool inc[N];
    sw_int_flux_system<URBAN, NREG, NS>(Lq, g, dcur.wall_albedo, dcur.wall_specular_frac, dcur.air_ext,
                                        dcur.air_ssa, dcur.veg_ssa, sin0, B, g3, inc);
    lu_factor<N>(B);
#endif

    // ---- the interface at the base (a_below, d_below of the layer below), or the ground ----
    double gdd = 0.0;
    if (jlay > 0) {
      cross(Lq, lo, false, vf_cur, bf_cur, ABE, DBE, dir_above, diff_above0, diff_above1, up_above0, up_above1);
    } else {
      // the ground (forest :479-484, urban :612-617): Lambertian, per region
      const double ga = Lq.in.ground_albedo[cidx], gad = Lq.in.ground_albedo_dir[cidx];
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg) {
        double sd0 = 0.0, sd1 = 0.0;
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) {
          sd0 += diff_above0[js + jreg * NS];
          sd1 += diff_above1[js + jreg * NS];
        }
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) {
          up_above0[js + jreg * NS] = (ga * lg.hweight[js]) * sd0 + (cos_sza * gad * lg.hweight[js]) * dir_above[jreg];
          up_above1[js + jreg * NS] = (ga * lg.hweight[js]) * sd1;
        }
      }
      gdd = ground_out(dir_above, diff_above0, up_above0, diff_above1, up_above1);
    }

    {
      SS_LAUNCH_HERE(L, Lo);
      if (Lo.flux_a.f[F_FLUX_DN_LAYER_TOP]) {
        const double ddt = cos_sza * s_top_dir;
        PUTP0(F_FLUX_DN_DIR_LAYER_TOP, ddt);
        PUTP0(F_FLUX_DN_LAYER_TOP, ddt + s_top_dn0);
        PUTP0(F_FLUX_UP_LAYER_TOP, s_top_up0);
        double s = 0.0;
        SS_UNROLL_FULL
        for (int r = 0; r < NREG; ++r) s += dir_above[r];
        const double ddb = cos_sza * s;
        PUTP0(F_FLUX_DN_DIR_LAYER_BASE, ddb);
        PUTP0(F_FLUX_DN_LAYER_BASE, ddb + vec_sum<N>(diff_above0));
        PUTP0(F_FLUX_UP_LAYER_BASE, vec_sum<N>(up_above0));
      }
      if (Lo.flux_b.f[F_FLUX_DN_LAYER_TOP]) {
        PUTP1(F_FLUX_DN_DIR_LAYER_TOP, 0.0);
        PUTP1(F_FLUX_DN_LAYER_TOP, 0.0 + s_top_dn1);
        PUTP1(F_FLUX_UP_LAYER_TOP, s_top_up1);
        PUTP1(F_FLUX_DN_DIR_LAYER_BASE, 0.0);
        PUTP1(F_FLUX_DN_LAYER_BASE, 0.0 + vec_sum<N>(diff_above1));
        PUTP1(F_FLUX_UP_LAYER_BASE, vec_sum<N>(up_above1));
      }
    }
    double int_flux_dir[D];
    mat_vec<D, D>(IDIR, dir_conv, int_flux_dir);
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) {
      conv0[i] = conv0[i] - diff_above0[i] + up_above0[i];
      conv1[i] = conv1[i] - diff_above1[i] + up_above1[i];
    }

    if (URBAN) {
      if (jb == itransp) {
        // urban :805-812
        SS_LAUNCH_HERE(L, Lo);
        const double nbf_up = (jlay + 1 < nlay) ? 1.0 - bf_up : 1.0;
        const double nbf = 1.0 - g.building_fraction;
        double roof_fraction;
        if (jlay == nlay - 1) roof_fraction = g.building_fraction;
        else roof_fraction = ss_max(0.0, g.building_fraction - bf_up);
        const double rsf = roof_in_dir * nbf_up
            / (cos_sza * flux_dn_dir_clear * ss_max(Lq.cfg.min_building_fraction, roof_fraction));
        if (Lo.flux_a.f[F_ROOF_SUNLIT_FRAC]) Lo.flux_a.f[F_ROOF_SUNLIT_FRAC][lay] = rsf;
        if (Lo.flux_b.f[F_ROOF_SUNLIT_FRAC]) Lo.flux_b.f[F_ROOF_SUNLIT_FRAC][lay] = 0.0;
        flux_dn_dir_clear = flux_dn_dir_clear * nbf / nbf_up;
      }
    }
    bf_up = g.building_fraction;

    // integrated fluxes: one factorisation, two solves
#if !defined(SS_DB_LU_EARLY)
    double B[N * N], g3[N * D];
    bool inc[N];
    sw_int_flux_system<URBAN, NREG, NS>(Lq, g, dcur.wall_albedo, dcur.wall_specular_frac, dcur.air_ext,
                                        dcur.air_ssa, dcur.veg_ssa, sin0, B, g3, inc);
    lu_factor<N>(B);
#endif
    double int_flux_diff0[N], int_flux_diff1[N];
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) {
      double s = conv0[i];
      SS_UNROLL_FULL
      for (int r = 0; r < D; ++r) s += 2.0 * SS_E(g3, i, r, N) * (inc[r * NS] ? int_flux_dir[r] : 0.0);
      int_flux_diff0[i] = inc[i] ? -s : 0.0;
      int_flux_diff1[i] = inc[i] ? -conv1[i] : 0.0;
    }
    lu_solve<N>(B, int_flux_diff0);
    lu_solve<N>(B, int_flux_diff1);

    // ---- request E, Q, P of the layer below: the working set of the solve is dead, what follows covers
    // part of their latency, the top of the next step the rest ----
#if defined(SS_DB_EQP_AHEAD)
    if (jlay > 0) {
      if (CRec::FUSED) colv.template load<D * D>(jlay - 1, CRec::EDIR, E);
      else layv.template load<D * D>(jlay - 1, LRec::EDIR, E);
      colv.template load<CRec::NQ>(jlay - 1, CRec::QDN, Q);
      colv.template load<N * N>(jlay - 1, CRec::PDN, P);
    }
    SS_LOAD_FENCE(); SS_DB_SCHED();
#endif
    vf_cur = lo[0];
    bf_cur = lo[1];

    // absorption (forest :608-628, urban :758-788)
    const double air_ext = dcur.air_ext, air_ssa = dcur.air_ssa;
    {
      double s0 = 0.0, s1 = 0.0;
      SS_UNROLL_FULL
      for (int js = 0; js < NS; ++js) {
        s0 += int_flux_diff0[js] * (1.0 / lg.mu[js]);
        s1 += int_flux_diff1[js] * (1.0 / lg.mu[js]);
      }
      STAGE(0, F_CLEAR_AIR_ABS, 0.0 + air_ext * (1.0 - air_ssa) * (int_flux_dir[0] + s0));
      STAGE(1, F_CLEAR_AIR_ABS, 0.0 + air_ext * (1.0 - air_ssa) * (0.0 + s1));
    }
    double veg_abs_dir = 0.0;
    if (NREG > 1) {
      double od[NREG];
      od_scaling<NREG>(g.veg_fsd, od);
      const double vssa = dcur.veg_ssa;
      double vaa0 = 0.0, va0 = 0.0, vaa1 = 0.0, va1 = 0.0;
      SS_UNROLL_FULL
      for (int jreg = 1; jreg < NREG; ++jreg) {
        double s0 = 0.0, s1 = 0.0;
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) {
          s0 += int_flux_diff0[jreg * NS + js] * (1.0 / lg.mu[js]);
          s1 += int_flux_diff1[jreg * NS + js] * (1.0 / lg.mu[js]);
        }
        vaa0 = vaa0 + air_ext * (1.0 - air_ssa) * (int_flux_dir[jreg] + s0);
        veg_abs_dir = veg_abs_dir + g.veg_ext * (1.0 - vssa) * int_flux_dir[jreg] * od[jreg];
        va0 = va0 + g.veg_ext * (1.0 - vssa) * (int_flux_dir[jreg] + s0) * od[jreg];
        vaa1 = vaa1 + air_ext * (1.0 - air_ssa) * (0.0 + s1);
        va1 = va1 + g.veg_ext * (1.0 - vssa) * (0.0 + s1) * od[jreg];
      }
      STAGE(0, F_VEG_AIR_ABS, vaa0);
      STAGE(0, F_VEG_ABS_DIR, veg_abs_dir);
      STAGE(0, F_VEG_ABS, va0);
      STAGE(1, F_VEG_AIR_ABS, vaa1);
      STAGE(1, F_VEG_ABS, va1);
    }

    double wall_in_dir = 0.0;
    double npw_sum = 0.0;
    if (URBAN) {
      // walls (urban :790-802)
      double frac[NREG], np[NREG], npw[NREG], f_wall[NREG];
      region_fractions<URBAN, false, NREG>(g.veg_fraction, g.building_fraction, frac);
      norm_perimeters<URBAN, NREG>(Lq.cfg, g, np, npw);
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) {
        f_wall[r] = (frac[r] <= Lq.cfg.min_vegetation_fraction) ? 0.0 : npw[r] / (kPi * frac[r]);
        npw_sum += npw[r];
      }
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg)
        wall_in_dir = wall_in_dir + f_wall[jreg] * sin0 * int_flux_dir[jreg];
      double wall_in0 = wall_in_dir, wall_in1 = 0.0;
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg) {
        double s0 = 0.0, s1 = 0.0;
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) {
          s0 += int_flux_diff0[jreg * NS + js] * lg.tan_ang[js];
          s1 += int_flux_diff1[jreg * NS + js] * lg.tan_ang[js];
        }
        wall_in0 = wall_in0 + f_wall[jreg] * s0;
        wall_in1 = wall_in1 + f_wall[jreg] * s1;
      }
      STAGE(0, F_WALL_IN_DIR, wall_in_dir);
      STAGE(0, F_WALL_IN, wall_in0);
      STAGE(1, F_WALL_IN, wall_in1);
      STAGE(2, F_WALL_NET, 1.0 - dcur.wall_albedo);   // wall_net = wall_in (1 - wall_albedo): write_group
    }

    // sunlit fractions and the clear-sky direct beam (forest :630-649,
    // urban :405-411, :814-852); band itransp only, zero in the diffuse pass
    if (jb == itransp) {
      SS_LAUNCH_HERE(L, Lo);
      const double ae = dcur.air_ext_t;
      const double trans_dir_clear = exp(-ae * g.dz / cos_sza);
      const double ifdc = (ae > 0.0) ? flux_dn_dir_clear * (1.0 - trans_dir_clear) * cos_sza / ae
                                     : flux_dn_dir_clear * g.dz;
      if (URBAN) {
        const double nbf = 1.0 - g.building_fraction;
        const double f_wall_dir_clear =
            (nbf <= Lq.cfg.min_building_fraction) ? 0.0 : npw_sum / (kPi * nbf);
        const double wsf = 0.5 * wall_in_dir / ss_max(DBL_EPSILON, (f_wall_dir_clear * sin0 * ifdc));
        if (Lo.flux_a.f[F_WALL_SUNLIT_FRAC]) Lo.flux_a.f[F_WALL_SUNLIT_FRAC][lay] = wsf;
        if (Lo.flux_b.f[F_WALL_SUNLIT_FRAC]) Lo.flux_b.f[F_WALL_SUNLIT_FRAC][lay] = 0.0;
      } else {
        if (Lo.flux_a.f[F_ROOF_SUNLIT_FRAC]) Lo.flux_a.f[F_ROOF_SUNLIT_FRAC][lay] = 0.0;
        if (Lo.flux_b.f[F_ROOF_SUNLIT_FRAC]) Lo.flux_b.f[F_ROOF_SUNLIT_FRAC][lay] = 0.0;
        if (Lo.flux_a.f[F_WALL_SUNLIT_FRAC]) Lo.flux_a.f[F_WALL_SUNLIT_FRAC][lay] = 0.0;
        if (Lo.flux_b.f[F_WALL_SUNLIT_FRAC]) Lo.flux_b.f[F_WALL_SUNLIT_FRAC][lay] = 0.0;
      }
      if (Lo.flux_a.f[F_VEG_SUNLIT_FRAC]) {
        double vsf = 0.0;
        if (NREG > 1) {
          const double vadc = ifdc * g.veg_ext * (1.0 - dcur.veg_ssa_t) * g.veg_fraction;
          vsf = veg_abs_dir / ss_max(DBL_EPSILON, vadc);
        }
        Lo.flux_a.f[F_VEG_SUNLIT_FRAC][lay] = vsf;
      }
      if (Lo.flux_b.f[F_VEG_SUNLIT_FRAC]) Lo.flux_b.f[F_VEG_SUNLIT_FRAC][lay] = 0.0;
      flux_dn_dir_clear = flux_dn_dir_clear * trans_dir_clear;
    }

    // the staging group is full, or the column is finished: write it out
    ++kst;
    if (kst == G || jlay == 0) {
#if !defined(SS_EXP_NO_FLUSH)      // timing experiment: the per-layer outputs are computed and staged, never written
      write_group(jlay, kst);
#endif
      kst = 0;
    }
    if (jlay == 0) ground_sunlit(gdd, flux_dn_dir_clear);
  }

#undef SS_DB_REQUEST_ABE
#undef PUTC0
#undef PUTC1
#undef PUTP0
#undef PUTP1
#undef STAGE
}

}  // namespace ss
