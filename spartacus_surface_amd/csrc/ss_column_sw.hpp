// ss_column_sw.hpp - shortwave column unit: one (column, band) lane does the
// adding method upward (section 4 of spartacus_forest_sw,
// radsurf_forest_sw.F90:463-528 / spartacus_urban_sw, radsurf_urban_sw.F90:591-674)
// and the two downward flux passes (section 5, forest :530-760 / urban
// :676-984).  What the downward sweep needs of the adding method (a_below,
// d_below and the operators P, Q that take the fluxes from the top to the base
// of a layer) goes through the column scratch [tile][layer][elem pair][lane].
//
// Two kernels: k_column (upward, matrix-matrix work) and k_down (both flux
// passes in one sweep, matrix-vector work only), one wave per SIMD each.
// Measured alternative: recomputing denominator /
// a_below / d_below in the downward sweep instead of storing them loads fewer
// doubles but needs > 256 live fp64 values per lane at n = 6 and spills more
// than it saves (sw_column 18 -> 27..37 ms per 1e6 columns).
#pragma once

#include "ss_common.hpp"
#include "ss_geometry.hpp"
#include "ss_layer_matrices.hpp"
#include "ss_linalg.hpp"
#include "ss_solver_sw.hpp"
#include "ss_pack.hpp"

namespace ss {

// Adding-method record of one layer in the column scratch [layer][elem][cu]
template <int N, int D>
struct SwColRec {
  static constexpr int PDN = 0;                           // P = Den^-1 T      N x N
  // (sw_dirvec, ss_layer_matrices.hpp: Q and d_below already applied to the direct flux at the top of the layer)
  static constexpr bool DIRVEC = sw_dirvec(N);
  static constexpr int NQ = DIRVEC ? N : N * D;
  static constexpr int QDN = PDN + rec_even(N * N);       // Q = Den^-1 (R d_above E + S_dn)  N x D  [or Q dir_top]
  static constexpr int ABE = QDN + rec_even(NQ);          // a_below(l+1)      N x N (canopy part)
  // a_below is symmetric in the scaled coordinates of its layer (reciprocity): the small orders store
  // D^-1 a_below D as a packed lower triangle (sym_packed, ss_layer_matrices.hpp)
  static constexpr bool PACKED = sym_packed_col(N, false);
  static constexpr int NAB = PACKED ? tri_count(N) : N * N;
  static constexpr int DBE = ABE + rec_even(NAB);         // d_below(l+1)      N x D (canopy part)  [or d_below dir_top]
  static constexpr int DTP = DBE + rec_even(NQ);          // DIRVEC: direct flux at the top of the layer, D (written by
                                                          // the first, downward stage of the upward kernel)
  // fused layer + upward kernel (fused_order(N)): no layer record exists, and what the downward
  // sweep reads of it travels in the column record: trans_dir_dir (E) and int_dir, D x D each
  static constexpr bool FUSED = fused_order(N, false);
#if defined(SS_FUSE_VIA_RECORD)
  // variant: S_up, S_dn also pass through the column record (re-read by the adding step) instead
  // of staying in registers across the layer stage; tail = [S_up S_dn E int_dir] as in SwRec
  static constexpr int TAIL = DBE + rec_even(N * D);
  static constexpr int EDIR = TAIL + 2 * rec_even(N * D);
#else
  // (fused AND direct beam as a vector, ss_layer_up_sw2.hpp: DTP holds the D + 1 components just below the top
  // interface of the layer - its regions and the exposed roofs -, E and int_dir follow)
  static constexpr int NDTP = (FUSED && DIRVEC) ? rec_even(D + 1) : rec_even(D);
  static constexpr int EDIR = (FUSED && DIRVEC) ? DTP + NDTP : DBE + rec_even(N * D);
#endif
  static constexpr int IDIR = EDIR + rec_even(D * D);
  static constexpr int NELEM = FUSED ? IDIR + rec_even(D * D) : DIRVEC ? DTP + rec_even(D) : DBE + rec_even(N * D);
  // Den = I - a_above R.  The downward sweep needs the flux at the layer base,
  //   diff_base = Den^-1 (T diff_top + R d_above E dir_top + S_dn dir_top) = P diff_top + Q dir_top,
  // so the two operators are stored instead of LU(Den): the sweep then reads
  // neither T, R nor S_dn of the layer record.
};

// Common per-unit context of the two column kernels
template <bool URBAN, int NREG, int NS>
struct SwColCtx {
  int col, jb, nspec, nlay, l0, itransp;
  long cidx;
  double cos_sza, sin0;
  SS_HD SwColCtx(const Launch& L, int cu, bool need_itransp) {
    const Inputs& in = L.in;
    nspec = L.nspec;
    col = L.cols[cu / nspec];
    jb = cu % nspec;
    nlay = in.nlay[col];
    l0 = in.istartlay[col] - 1;
    cidx = (long)jb + (long)nspec * col;
    const double cos_in = in.cos_sza[col];
    cos_sza = URBAN ? ss_max(cos_in, 1.0e-6) : cos_in;
    sin0 = URBAN ? sqrt(1.0 - cos_sza * cos_sza) : 0.0;
    // most transparent band (forest :253, urban :310): minloc of the canopy
    // optical depth - the only cross-band quantity; recomputed by every lane
    itransp = 0;
    if (need_itransp && nspec > 1) {
      double best = 0.0;
      for (int b = 0; b < nspec; ++b) {
        double odsum = 0.0;
        for (int l = 0; l < nlay; ++l) odsum += in.air_ext[(long)b + (long)nspec * (l0 + l)] * in.dz[l0 + l];
        if (b == 0 || odsum < best) { best = odsum; itransp = b; }
      }
    }
  }
};

// ---- section 4: adding method upward ----
// a_above, d_above at the ground (forest :466-477, urban :594-606)
template <int NREG, int NS>
SS_HD void sw_ground_albedo(const LG& lg, double ga, double gad, double cos_sza, double* Aab, double* Dab) {
  constexpr int N = NREG * NS, D = NREG;
  mat_zero<N, N>(Aab);
  mat_zero<N, D>(Dab);
  SS_UNROLL_FULL
  for (int jreg = 0; jreg < NREG; ++jreg) {
    SS_UNROLL_FULL
    for (int jsto = 0; jsto < NS; ++jsto) {
      SS_E(Dab, jsto + jreg * NS, jreg, N) = cos_sza * gad * lg.hweight[jsto];
      SS_UNROLL_FULL
      for (int jsfr = 0; jsfr < NS; ++jsfr)
        SS_E(Aab, jsto + jreg * NS, jsfr + jreg * NS, N) = ga * lg.hweight[jsto];
    }
  }
}

// Adding step of one layer (forest :486-493, urban :619-626): consumes R, T, S_up, S_dn, E of
// the layer from "src" (HBM layer record or the fused layer stage), stores the operators of
// the downward sweep and a_below / d_below into the column record, returns the latter.
// dsc, rd: scaling of the layer's components and its reciprocal (layer_scaling; read where the records are packed)
template <int N, int D, class Src>
SS_HD void sw_adding_step(const Src& src, const double* Aab, const double* Dab, double* crec,
                          double* Abe, double* Dbe, const double* dsc, const double* rd) {
  using LRec = SwLayRec<N, D>;
  using CRec = SwColRec<N, D>;
  constexpr long stride = kRecStride;
  double R[N * N], T[N * N], Den[N * N], X[N * N];
  if constexpr (LRec::PACKED) {
    load_packed_unscaled<N>(src, LRec::R, dsc, rd, R);
    load_packed_unscaled<N>(src, LRec::T, dsc, rd, T);
  } else {
    src.template load<N * N>(LRec::R, R);
    src.template load<N * N>(LRec::T, T);
  }
  SS_LOAD_FENCE(); SS_SCHED_FENCE();   // the loads of E, S_dn, S_up stay where they are written (register pressure)
  // denominator = I - a_above R
  mat_mul<N, N, N>(Aab, R, Den);
  SS_UNROLL_NN(N)
  for (int k = 0; k < N * N; ++k) Den[k] = -Den[k];
  SS_UNROLL_N(N)
  for (int j = 0; j < N; ++j) SS_E(Den, j, j, N) = 1.0 + SS_E(Den, j, j, N);
  lu_factor<N>(Den);
  // P = Den^-1 T
  mat_copy<N, N>(T, X);
  lu_solve_mat<N, N>(Den, X);
  rec_store<N * N>(crec, stride, CRec::PDN, X);
  // a_below = R + T Den^-1 (a_above T)
  mat_mul<N, N, N>(Aab, T, X);
  lu_solve_mat<N, N>(Den, X);
  mat_mul<N, N, N>(T, X, Abe);
  SS_UNROLL_NN(N)
  for (int k = 0; k < N * N; ++k) Abe[k] = R[k] + Abe[k];
  if constexpr (CRec::PACKED) {
    store_packed_scaled<N>(crec, stride, CRec::ABE, Abe, dsc, rd);
  } else {
    rec_store<N * N>(crec, stride, CRec::ABE, Abe);
  }
  // d_below = S_up + T Den^-1 (d_above E + a_above S_dn)
  double E[D * D], S[N * D], Y[N * D];
  src.template load<D * D>(LRec::EDIR, E);
  src.template load<N * D>(LRec::SDN, S);
  mat_mul<N, D, D>(Dab, E, Y);
  {
    // Q = Den^-1 (R (d_above E) + S_dn)
    double Q[N * D];
    mat_mul<N, N, D>(R, Y, Q);
    SS_UNROLL_FULL
    for (int k = 0; k < N * D; ++k) Q[k] = Q[k] + S[k];
    lu_solve_mat<N, D>(Den, Q);
    rec_store<N * D>(crec, stride, CRec::QDN, Q);
  }
  mat_mul_acc<N, N, D>(Aab, S, Y);
  lu_solve_mat<N, D>(Den, Y);
  src.template load<N * D>(LRec::SUP, S);
  mat_mul<N, N, D>(T, Y, Dbe);
  SS_UNROLL_FULL
  for (int k = 0; k < N * D; ++k) Dbe[k] = S[k] + Dbe[k];
  rec_store<N * D>(crec, stride, CRec::DBE, Dbe);
}

// Overlap to the base of the layer above layer jlay, or to the free atmosphere
// (forest :495-503, urban :627-653): a_above = (U x I) [a_below (V x I); roof rows], same for d
template <bool URBAN, int NREG, int NS>
SS_HD void sw_overlap_above(const Launch& L, int cu, int jlay, int nlay, double cos_sza, const double* Abe,
                            const double* Dbe, double* Aab, double* Dab) {
  constexpr int N = NREG * NS, D = NREG;
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  const LG& lg = L.lg;
  double U[NREG * DB], V[DB * NREG];
  const double* prec = pack_rec(L, jlay, cu, kPackSw);
  double vf_lo, bf_lo, vf_up = 0.0, bf_up = 0.0;
  pack_fractions(prec, vf_lo, bf_lo);
  if (jlay + 1 < nlay) pack_fractions(pack_rec(L, jlay + 1, cu, kPackSw), vf_up, bf_up);
  interface_overlap_vals<URBAN, false, NREG>(L.cfg.min_vegetation_fraction, jlay + 1 >= nlay, vf_up, bf_up,
                                             false, vf_lo, bf_lo, U, V);
  double roof_a = 0.0, roof_d = 0.0;
  if (URBAN) {
    const RecPair ra = rec_pair_load(prec + 8 * kRecStride);   // roof_albedo, roof_albedo_dir
    roof_a = ra.x;
    roof_d = cos_sza * ra.y;
  }
  // tA = a_below (V (x) I): canopy rows (N x N) and roof rows (NS x N)
  double tA[N * N], tAroof[NS * N];
  SS_UNROLL_FULL
  for (int ju = 0; ju < NREG; ++ju) {
    SS_UNROLL_FULL
    for (int s = 0; s < NS; ++s) {
      const int cc = ju * NS + s;
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) {
        double acc = 0.0;
        SS_UNROLL_FULL
        for (int jl = 0; jl < NREG; ++jl) acc += SS_E(Abe, i, jl * NS + s, N) * SS_E(V, jl, ju, DB);
        SS_E(tA, i, cc, N) = acc;
      }
      SS_UNROLL_FULL
      for (int sr = 0; sr < NS; ++sr)
        SS_E(tAroof, sr, cc, NS) = URBAN ? (roof_a * lg.hweight[sr]) * SS_E(V, DB - 1, ju, DB) : 0.0;
    }
  }
  apply_U_mat<NREG, DB, NS, N>(U, tA, tAroof, Aab);
  // tD = d_below V (rect_mat_x_singlemat :462): canopy rows and roof rows
  double tD[N * D], tDroof[NS * D];
  SS_UNROLL_FULL
  for (int ju = 0; ju < NREG; ++ju) {
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) {
      double acc = 0.0;
      SS_UNROLL_FULL
      for (int jl = 0; jl < NREG; ++jl) acc += SS_E(Dbe, i, jl, N) * SS_E(V, jl, ju, DB);
      SS_E(tD, i, ju, N) = acc;
    }
    SS_UNROLL_FULL
    for (int sr = 0; sr < NS; ++sr)
      SS_E(tDroof, sr, ju, NS) = URBAN ? (roof_d * lg.hweight[sr]) * SS_E(V, DB - 1, ju, DB) : 0.0;
  }
  apply_U_mat<NREG, DB, NS, D>(U, tD, tDroof, Dab);
}

// top-of-canopy albedos (forest :520-528, urban :666-674)
template <int NREG, int NS>
SS_HD void sw_top_albedos(const Launch& L, long cidx, double cos_sza, const double* Aab, const double* Dab) {
  constexpr int N = NREG * NS;
  const LG& lg = L.lg;
  double s = 0.0;
  SS_UNROLL_FULL
  for (int j1 = 0; j1 < NS; ++j1) {
    double acc = 0.0;
    SS_UNROLL_FULL
    for (int j2 = 0; j2 < NS; ++j2) acc += SS_E(Aab, j1, j2, N) * lg.hweight[j2];
    s += acc;
  }
  L.top1[cidx] = s;
  s = 0.0;
  SS_UNROLL_FULL
  for (int j1 = 0; j1 < NS; ++j1) s += SS_E(Dab, j1, 0, N);
  L.top2[cidx] = s / cos_sza;
}

// The upward sweep in the reference's (unscaled) coordinates: orders whose layer record is not packed
// (k_column dispatches in ss_column_sw_sym.hpp)
template <bool URBAN, int NREG, int NS>
SS_HD void column_up_sw_full(const Launch& L, int cu) {
  constexpr int N = NREG * NS, D = NREG;
  using LRec = SwLayRec<N, D>;
  using CRec = SwColRec<N, D>;
  const Inputs& in = L.in;
  const SwColCtx<URBAN, NREG, NS> c(L, cu, false);
  double Aab[N * N], Dab[N * D];
  sw_ground_albedo<NREG, NS>(L.lg, in.ground_albedo[c.cidx], in.ground_albedo_dir[c.cidx], c.cos_sza, Aab, Dab);
  for (int jlay = 0; jlay < c.nlay; ++jlay) {
    const RecSource src{L.lay + rec_base(jlay, L.maxlay, LRec::NELEM, cu), kRecStride};
    double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    double Abe[N * N], Dbe[N * D];
    double dsc[N], rd[N];
    if constexpr (LRec::PACKED || CRec::PACKED) {
      double vf, bf, frac[NREG];
      pack_fractions(pack_rec(L, jlay, cu, kPackSw), vf, bf);
      region_fractions<URBAN, false, NREG>(vf, bf, frac);
      layer_scaling<NREG, NS>(L.lg, frac, dsc, rd);
    }
    sw_adding_step<N, D>(src, Aab, Dab, crec, Abe, Dbe, dsc, rd);
    sw_overlap_above<URBAN, NREG, NS>(L, cu, jlay, c.nlay, c.cos_sza, Abe, Dbe, Aab, Dab);
  }
  sw_top_albedos<NREG, NS>(L, c.cidx, c.cos_sza, Aab, Dab);
}

// Layer matrices and adding step of every layer of a column unit on one lane (k_layer_up, see
// the header of the fused kernels in ss_layer_matrices.hpp)
template <bool URBAN, int NREG, int NS>
SS_HD void layer_up_sw(const Launch& L, int cu) {
  constexpr int N = NREG * NS, D = NREG;
  using LRec = SwRec<N, D>;
  using CRec = SwColRec<N, D>;
  static_assert(CRec::FUSED && LRec::SUP <= kLaneLdsSlots, "fused form needs R and T in the lane's LDS slots");
  using Lds = LaneLds<true>;
  constexpr long stride = kRecStride;
  const Inputs& in = L.in;
  const SwColCtx<URBAN, NREG, NS> c(L, cu, false);
  const Lds lds{};
  for (int jlay = 0; jlay < c.nlay; ++jlay) {
    double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    // layer stage: R, T -> LDS; S_up, S_dn, E -> registers; E, int_dir -> column record
#if defined(SS_FUSE_VIA_RECORD)
    double hold[2];
    const FusedWriter<LRec::SUP, LRec::SUP, LRec::SUP, CRec::TAIL, Lds> w{
        &lds, hold, crec, stride, (unsigned long long*)L.eigen_fail};
#else
    double hold[LRec::IDIR - LRec::SUP];
    const FusedWriter<LRec::SUP, LRec::IDIR, LRec::EDIR, CRec::EDIR, Lds> w{
        &lds, hold, crec, stride, (unsigned long long*)L.eigen_fail};
#endif
    layer_unit_sw<URBAN, NREG, NS>(L, cu, jlay, w);
    // a_above, d_above: from the ground, or from a_below / d_below of the layer below (written a
    // layer ago) through the overlap matrices of the interface in between
    double Aab[N * N], Dab[N * D];
    if (jlay == 0) {
      sw_ground_albedo<NREG, NS>(L.lg, in.ground_albedo[c.cidx], in.ground_albedo_dir[c.cidx], c.cos_sza, Aab, Dab);
    } else {
      const double* prev = L.col + rec_base(jlay - 1, L.maxlay, CRec::NELEM, cu);
      double Abe[N * N], Dbe[N * D];
      rec_load<N * N>(prev, stride, CRec::ABE, Abe);
      rec_load<N * D>(prev, stride, CRec::DBE, Dbe);
      sw_overlap_above<URBAN, NREG, NS>(L, cu, jlay - 1, c.nlay, c.cos_sza, Abe, Dbe, Aab, Dab);
    }
#if defined(SS_FUSE_VIA_RECORD)
    const LdsRecSource<LRec::SUP, CRec::TAIL, Lds> src{&lds, crec, stride};
#else
    const FusedSource<LRec::SUP, Lds> src{&lds, hold};
#endif
    double Abe[N * N], Dbe[N * D];
    sw_adding_step<N, D>(src, Aab, Dab, crec, Abe, Dbe, nullptr, nullptr);
  }
  if (c.nlay > 0) {
    const double* prev = L.col + rec_base(c.nlay - 1, L.maxlay, CRec::NELEM, cu);
    double Abe[N * N], Dbe[N * D], Aab[N * N], Dab[N * D];
    rec_load<N * N>(prev, stride, CRec::ABE, Abe);
    rec_load<N * D>(prev, stride, CRec::DBE, Dbe);
    sw_overlap_above<URBAN, NREG, NS>(L, cu, c.nlay - 1, c.nlay, c.cos_sza, Abe, Dbe, Aab, Dab);
    sw_top_albedos<NREG, NS>(L, c.cidx, c.cos_sza, Aab, Dab);
  }
}

// Scalars of one layer used by the downward sweep, fetched one layer ahead
template <bool URBAN, int NREG>
struct SwDownIn {
  LayerGeom g;
  double roof_albedo, roof_albedo_dir, wall_albedo, air_ext, air_ssa, veg_ssa, air_ext_t, veg_ssa_t;
  double wall_specular_frac;
  SS_HD void load(const Inputs& in, int lay, int jb, int itransp, int nspec) {
    const long sidx = (long)jb + (long)nspec * lay, tidx = (long)itransp + (long)nspec * lay;
    g = load_layer_geom<URBAN, NREG>(in, lay);
    roof_albedo = roof_albedo_dir = wall_albedo = 0.0;
    if (URBAN) {
      roof_albedo = in.roof_albedo[sidx];
      roof_albedo_dir = in.roof_albedo_dir[sidx];
      wall_albedo = in.wall_albedo[sidx];
    }
    air_ext = in.air_ext[sidx];
    air_ssa = in.air_ssa[sidx];
    veg_ssa = (NREG > 1) ? in.veg_ssa[sidx] : 0.0;
    air_ext_t = in.air_ext[tidx];
    veg_ssa_t = (NREG > 1) ? in.veg_ssa[tidx] : 0.0;
    wall_specular_frac = URBAN ? in.wall_specular_frac[sidx] : 0.0;
  }
  // The same from the input pack (ss_pack.hpp).  The *_t members (band itransp) are
  // only used by the lane whose band IS itransp (sunlit fractions), for which they
  // equal its own air_ext / veg_ssa.
  SS_HD void load_pack(const RecView& pv, int jlay) {
    double v[kPackSw];
    pv.template load<kPackSw>(jlay, 0, v);
    set_pack(v);
  }
  SS_HD void load_pack(const double* prec) {
    double v[kPackSw];
    rec_load<kPackSw>(prec, kRecStride, 0, v);
    set_pack(v);
  }
  SS_HD void set_pack(const double* v) {
    wall_specular_frac = v[14];
    g.dz = v[0]; g.veg_fraction = v[1]; g.veg_scale = v[2]; g.veg_ext = v[3];
    g.veg_fsd = v[4]; g.building_fraction = v[5]; g.building_scale = v[6]; g.veg_contact_fraction = v[7];
    roof_albedo = v[8]; roof_albedo_dir = v[9]; wall_albedo = v[10];
    air_ext = v[11]; air_ssa = v[12]; veg_ssa = v[13];
    air_ext_t = air_ext;
    veg_ssa_t = veg_ssa;
  }
};

// Staging plan of the per-layer outputs of the downward kernel (see PUTL there).  Ten fields x maxlay
// layers per lane do not fit the 80 LDS slots of a lane at 10 layers, but the two lanes of a column (direct
// pass on the even lane, diffuse pass on the odd one) do not need the same: the three direct-beam fields are
// identically zero in the diffuse pass (written as zeros at the end, never staged there).  So the diffuse
// lane stages the fields of slots 0..6, the direct lane those of slots 0..7 in its own slots and the field
// of slot 8 in its neighbour's unused slots 7 maxlay .. 8 maxlay - 1; the field of slot 9 is stored as it
// is computed (direct pass only).
constexpr int kSwDownFields = 10, kSwDownDiffuseFields = 7, kSwDownOwnFields = 8;
constexpr int sw_down_slot(int F) {
  switch (F) {
    case F_CLEAR_AIR_ABS: return 0;
    case F_VEG_ABS: return 1;
    case F_VEG_AIR_ABS: return 2;
    case F_WALL_IN: return 3;
    case F_WALL_NET: return 4;
    case F_ROOF_IN: return 5;
    case F_ROOF_NET: return 6;
    case F_VEG_ABS_DIR: return 7;
    case F_WALL_IN_DIR: return 8;
    case F_ROOF_IN_DIR: return 9;
    default: return -1;
  }
}
constexpr int kSwDownSlotField[kSwDownFields] = {F_CLEAR_AIR_ABS, F_VEG_ABS, F_VEG_AIR_ABS, F_WALL_IN, F_WALL_NET,
                                                 F_ROOF_IN, F_ROOF_NET, F_VEG_ABS_DIR, F_WALL_IN_DIR, F_ROOF_IN_DIR};
constexpr bool sw_down_slots_consistent() {
  for (int s = 0; s < kSwDownFields; ++s)
    if (sw_down_slot(kSwDownSlotField[s]) != s) return false;
  return true;
}
static_assert(sw_down_slots_consistent(), "kSwDownSlotField must be the inverse of sw_down_slot");

// ---- section 5: the downward flux passes (k_down) ----
// One lane = (column, band, pass): pass 0 is normalised by the direct flux at
// canopy top (sw_norm_dir, forest :530-676 / urban :676-874), pass 1 by the
// diffuse flux (sw_norm_diff, forest :684-760 / urban :876-984).  The diffuse
// pass is the direct pass with a zero direct beam, so both run the same
// instruction stream (no divergence) and the two lanes of a column read the
// same scratch addresses (one memory request).  Only vectors are carried.  The
// sweep is software-pipelined by hand: the matrices of a layer are fetched in
// groups, each issued (SS_LOAD_FENCE) one stage before the arithmetic that uses
// it, instead of ~90 dependent load-use round trips per layer.
template <bool URBAN, int NREG, int NS>
SS_HD void column_down_sw(const Launch& L, int cu, int pass) {
  constexpr int N = NREG * NS, D = NREG;
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  constexpr int NB = DB * NS;
  using LRec = SwLayRec<N, D>;
  using CRec = SwColRec<N, D>;
  constexpr long st = kRecStride;
  const Inputs& in = L.in;
  const LG& lg = L.lg;
  const SwColCtx<URBAN, NREG, NS> c(L, cu, true);
  // a_below of a layer: the column record holds it packed and scaled for the small orders; NABE doubles are
  // loaded as they lie and applied through abe_matvec (vf, bf: the fractions of the layer it belongs to)
  constexpr int NABE = CRec::PACKED ? rec_even(tri_count(N)) : N * N;
  auto abe_matvec = [&](const double* ABE_, double vf_, double bf_, const double* x_, double* y_) {
    if constexpr (CRec::PACKED) {
      double frac_[NREG], dsc_[N], rd_[N];
      region_fractions<URBAN, false, NREG>(vf_, bf_, frac_);
      layer_scaling<NREG, NS>(lg, frac_, dsc_, rd_);
      packed_scaled_matvec<N>(ABE_, dsc_, rd_, x_, y_);
    } else {
      mat_vec<N, N>(ABE_, x_, y_);
    }
  };
  const int col = c.col, jb = c.jb, nspec = c.nspec, nlay = c.nlay, l0 = c.l0, itransp = c.itransp;
  const long cidx = c.cidx;
  const double cos_sza = c.cos_sza, sin0 = c.sin0;
  const bool dirpass = (pass == 0);
  const double top_albedo = dirpass ? L.top2[cidx] : L.top1[cidx];

  // flux object of this pass: per-lane selection of the array base pointers
  // (both pointers are wave-uniform scalars; the select is one v_cndmask pair)
#define FSEL(F) (dirpass ? L.flux_a.f[F] : L.flux_b.f[F])
#define PUTC(F, v) do { double* p_ = FSEL(F); if (p_) p_[cidx] = (v); } while (0)
// Per-layer outputs: a lane writes one 8-byte element of a field per layer, ~20 us
  // apart, into lines it shares with nobody until the column is finished; at
  // 66 KB of such lines per wave L2 evicts them half-written (measured: the
  // stores are 2.2 of the 7.1 ms of this kernel, three times their bytes' worth).
  // The fields that fit are therefore staged in LDS and written out back to back
  // after the sweep, so that every line is completed while it is still in L2.
  LaneLds<(N <= SS_UNROLL_MAX)> obuf;
  const int nbuf = obuf.kSlots / (L.maxlay > 0 ? L.maxlay : 1);   // fields that fit a lane's own slots
  // is the field of slot s_ staged (by the lane that has something to stage)?
  auto staged = [&](int s_) { return s_ < kSwDownOwnFields ? s_ < nbuf : (s_ == kSwDownOwnFields && nbuf >= kSwDownOwnFields); };
  // (a staged field is staged whether or not its member is allocated - that is looked at when the fields are
  // written out: inside the sweep an output pointer is only touched by the fields that are stored directly,
  // which keeps two dozen wave-uniform pointers out of the scalar registers of the loop)
#define PUTL(F, v) do { constexpr int s_ = sw_down_slot(F); \
    if (s_ >= kSwDownDiffuseFields && !dirpass) { /* zero, written after the sweep */ } \
    else if (s_ >= 0 && staged(s_)) { \
      if (s_ < kSwDownOwnFields) obuf.put(s_ * L.maxlay + jlay, (v)); \
      else obuf.put_next(kSwDownDiffuseFields * L.maxlay + jlay, (v)); } \
    else { double* p_ = FSEL(F); if (p_) p_[sidx] = (v); } } while (0)

  double dir_above[D], diff_above[N], up_above[N];
  dir_above[0] = dirpass ? 1.0 / cos_sza : 0.0;
  SS_UNROLL_FULL
  for (int r = 1; r < D; ++r) dir_above[r] = 0.0;
  vec_zero<N>(diff_above);
  vec_zero<N>(up_above);
  if (!dirpass) {
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) diff_above[js] = lg.hweight[js];
  }
  PUTC(F_TOP_DN_DIR, dirpass ? 1.0 : 0.0);
  PUTC(F_TOP_DN, 1.0);
  PUTC(F_TOP_NET, dirpass ? 1.0 * (1.0 - top_albedo) : 1.0 - top_albedo);
  double flux_dn_dir_clear = 1.0 / cos_sza;   // clear-sky direct beam of the direct pass

  // Crossing the interface at the top of layer jl (lower), coming from the base
  // of the layer above (or the free atmosphere): downward fluxes through V, the
  // upward flux just below the interface from a_below / d_below of layer jl (and
  // its exposed roofs), and the upward flux just above it through U.  The last
  // one is what the reference gets as a_above flux + d_above dir at the base of the
  // layer above: a_above = (U x I) [a_below (V x I); roof rows], so the product
  // with the flux vector is formed instead of the matrix, and a_above / d_above
  // are neither stored by the upward sweep nor read here.
  // (lo = the four scalars of the lower layer this needs: veg_fraction, building_fraction,
  // roof_albedo, roof_albedo_dir)
  auto cross = [&](const double* lo, bool upper_is_sky, double vf_up, double bf_up,
                   const double* ABE, const double* DBE, const double* dir_ab, const double* diff_ab,
                   double* dir_below, double* diff_below, double* up_below, double* up_ab) {
    double U[NREG * DB], V[DB * NREG];
    interface_overlap_vals<URBAN, false, NREG>(L.cfg.min_vegetation_fraction, upper_is_sky, vf_up, bf_up,
                                               false, lo[0], lo[1], U, V);
    SS_UNROLL_FULL
    for (int jl = 0; jl < DB; ++jl) {
      double acc = 0.0;
      SS_UNROLL_FULL
      for (int ju = 0; ju < NREG; ++ju) acc += SS_E(V, jl, ju, DB) * dir_ab[ju];
      dir_below[jl] = acc;
    }
    apply_V_vec<NREG, DB, NS>(V, diff_ab, diff_below);
    abe_matvec(ABE, lo[0], lo[1], diff_below, up_below);
    if constexpr (CRec::DIRVEC) {
      // d_below dir_below, formed by the upward kernel (zero in the pass without a direct beam)
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) up_below[i] = up_below[i] + (dirpass ? DBE[i] : 0.0);
    } else {
      mat_vec_acc<N, D>(DBE, dir_below, up_below);
    }
    if (URBAN) {
      // exposed roof (urban :717-721): a_below / d_below roof blocks
      const double roof_a = lo[2];
      const double roof_d = cos_sza * lo[3];
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) {
        // row n+s of a_below is roof_albedo*hweight(s) in every roof column
        double acc = 0.0;
        SS_UNROLL_FULL
        for (int s2 = 0; s2 < NS; ++s2) acc += (roof_a * lg.hweight[s]) * diff_below[N + s2];
        up_below[N + s] = acc + (roof_d * lg.hweight[s]) * dir_below[DB - 1];
      }
    }
    if (up_ab) apply_U_mat<NREG, DB, NS, 1>(U, up_below, up_below + N, up_ab);
  };

  // top layer: its scalars and group 1 (a_below, d_below), then the interface to the free atmosphere
  SwDownIn<URBAN, NREG> din;
  double dir_below[DB], diff_below[NB], up_below[NB];
  double bf_up = 0.0;          // building fraction of the layer above (roof / sunlit bookkeeping)
  if (nlay > 0) {
    const int jlay = nlay - 1;
    const double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    double ABE[NABE], DBE[CRec::NQ];
    din.load_pack(pack_rec(L, jlay, cu, kPackSw));
    rec_load<NABE>(crec, st, CRec::ABE, ABE);
    rec_load<CRec::NQ>(crec, st, CRec::DBE, DBE);
    SS_LOAD_FENCE();
    const double lo[4] = {din.g.veg_fraction, din.g.building_fraction, din.roof_albedo, din.roof_albedo_dir};
    cross(lo, true, 0.0, 0.0, ABE, DBE, dir_above, diff_above, dir_below, diff_below, up_below, nullptr);
  }

  for (int jlay = nlay - 1; jlay >= 0; --jlay) {
    const double* lrec = L.lay + rec_base(jlay, L.maxlay, LRec::NELEM, cu);
    const double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    const int lay = l0 + jlay;
    const long sidx = (long)jb + (long)nspec * lay;
    const SwDownIn<URBAN, NREG> dcur = din;
    const LayerGeom g = dcur.g;

    // ---- issue group 2 (E, Q), work on the fluxes just below the top interface ----
    double E[D * D], Q[CRec::NQ];
    if (CRec::FUSED) rec_load<D * D>(crec, st, CRec::EDIR, E);
    else rec_load<D * D>(lrec, st, LRec::EDIR, E);
    rec_load<CRec::NQ>(crec, st, CRec::QDN, Q);
    SS_LOAD_FENCE();

    if (URBAN) {
      double sdn = 0.0, sup = 0.0;
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) { sdn += diff_below[N + s]; sup += up_below[N + s]; }
      const double roof_in_dir = cos_sza * dir_below[DB - 1];
      PUTL(F_ROOF_IN_DIR, roof_in_dir);
      PUTL(F_ROOF_IN, roof_in_dir + sdn);
      PUTL(F_ROOF_NET, (roof_in_dir + sdn) - sup);
      if (jb == itransp) {
        // urban :805-812
        const double nbf_up = (jlay + 1 < nlay) ? 1.0 - bf_up : 1.0;
        const double nbf = 1.0 - g.building_fraction;
        double roof_fraction;
        if (jlay == nlay - 1) roof_fraction = g.building_fraction;
        else roof_fraction = ss_max(0.0, g.building_fraction - bf_up);
        const double rsf = roof_in_dir * nbf_up
            / (cos_sza * flux_dn_dir_clear * ss_max(L.cfg.min_building_fraction, roof_fraction));
        if (FSEL(F_ROOF_SUNLIT_FRAC)) FSEL(F_ROOF_SUNLIT_FRAC)[lay] = dirpass ? rsf : 0.0;
        flux_dn_dir_clear = flux_dn_dir_clear * nbf / nbf_up;
      }
    }

    // ---- issue group 3 (P), work on group 2 ----
    double P[N * N];
    rec_load<N * N>(crec, st, CRec::PDN, P);
    SS_LOAD_FENCE();

    double dir_conv[D];   // dir_below - dir_above (new)
    {
      double t[D];
      mat_vec<D, D>(E, dir_below, t);
      SS_UNROLL_FULL
      for (int r = 0; r < D; ++r) { dir_above[r] = t[r]; dir_conv[r] = dir_below[r] - t[r]; }
    }
    double t3[N];
    if constexpr (CRec::DIRVEC) {
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) t3[i] = dirpass ? Q[i] : 0.0;     // Q dir_below, formed by the upward kernel
    } else {
      mat_vec<N, D>(Q, dir_below, t3);
    }

    // ---- issue group 4 (a_below, d_below and the scalars of the layer below; int_dir), work on group 3 ----
    double ABE[NABE], DBE[CRec::NQ], IDIR[D * D];
    double lo[4] = {0.0, 0.0, 0.0, 0.0};
    if (jlay > 0) {
      const double* crec_n = L.col + rec_base(jlay - 1, L.maxlay, CRec::NELEM, cu);
      const double* pn = pack_rec(L, jlay - 1, cu, kPackSw);
      pack_fractions(pn, lo[0], lo[1]);
      const RecPair ra = rec_pair_load(pn + 8 * kRecStride);   // roof_albedo, roof_albedo_dir
      lo[2] = ra.x;
      lo[3] = ra.y;
      rec_load<NABE>(crec_n, st, CRec::ABE, ABE);
      rec_load<CRec::NQ>(crec_n, st, CRec::DBE, DBE);
    }
    if (CRec::FUSED) rec_load<D * D>(crec, st, CRec::IDIR, IDIR);
    else rec_load<D * D>(lrec, st, LRec::IDIR, IDIR);
    SS_LOAD_FENCE();

    // flux at the base of the layer: diff = P diff_top + Q dir_top
    double conv[N];       // diff_below - up_below at the top, completed below
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) conv[i] = diff_below[i] - up_below[i];
    double s_top_dir = 0.0;
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) s_top_dir += dir_below[r];
    const double s_top_dn = vec_sum<N>(diff_below), s_top_up = vec_sum<N>(up_below);
    mat_vec<N, N>(P, diff_below, diff_above);
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) diff_above[i] = diff_above[i] + t3[i];

    // ---- group 4: the interface at the base ----
    if (jlay > 0) {
      cross(lo, false, g.veg_fraction, g.building_fraction, ABE, DBE, dir_above, diff_above,
            dir_below, diff_below, up_below, up_above);
    } else {
      // the ground (forest :479-484, urban :612-617): Lambertian, per region
      const double ga = in.ground_albedo[cidx], gad = in.ground_albedo_dir[cidx];
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg) {
        double sd = 0.0;
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) sd += diff_above[js + jreg * NS];
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js)
          up_above[js + jreg * NS] = (ga * lg.hweight[js]) * sd + (cos_sza * gad * lg.hweight[js]) * dir_above[jreg];
      }
    }

    if (FSEL(F_FLUX_DN_LAYER_TOP)) {
      const double ddt = cos_sza * s_top_dir;
      PUTL(F_FLUX_DN_DIR_LAYER_TOP, ddt);
      PUTL(F_FLUX_DN_LAYER_TOP, ddt + s_top_dn);
      PUTL(F_FLUX_UP_LAYER_TOP, s_top_up);
      double s = 0.0;
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) s += dir_above[r];
      const double ddb = cos_sza * s;
      PUTL(F_FLUX_DN_DIR_LAYER_BASE, ddb);
      PUTL(F_FLUX_DN_LAYER_BASE, ddb + vec_sum<N>(diff_above));
      PUTL(F_FLUX_UP_LAYER_BASE, vec_sum<N>(up_above));
    }
    double int_flux_dir[D];
    mat_vec<D, D>(IDIR, dir_conv, int_flux_dir);
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) conv[i] = conv[i] - diff_above[i] + up_above[i];

    // ---- issue: the scalars of the layer below ----
    if (jlay > 0) din.load_pack(pack_rec(L, jlay - 1, cu, kPackSw));
    SS_LOAD_FENCE();
    bf_up = g.building_fraction;

    // integrated fluxes (forest :598-606): int_diff conv + int_dir_diff dir_conv
    //   = -(Gamma1 + Gamma2)^-1 (conv + 2 Gamma3 int_flux_dir), see sw_int_flux_system
    double int_flux_diff[N];
    {
      double B[N * N], g3[N * D];
      bool inc[N];
      sw_int_flux_system<URBAN, NREG, NS>(L, g, dcur.wall_albedo, dcur.wall_specular_frac, dcur.air_ext,
                                          dcur.air_ssa, dcur.veg_ssa, sin0, B, g3, inc);
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) {
        double s = conv[i];
        SS_UNROLL_FULL
        for (int r = 0; r < D; ++r) s += 2.0 * SS_E(g3, i, r, N) * (inc[r * NS] ? int_flux_dir[r] : 0.0);
        int_flux_diff[i] = inc[i] ? -s : 0.0;
      }
      lu_factor<N>(B);
      lu_solve<N>(B, int_flux_diff);
    }

    // absorption (forest :608-628, urban :758-788)
    const double air_ext = dcur.air_ext, air_ssa = dcur.air_ssa;
    {
      double s = 0.0;
      SS_UNROLL_FULL
      for (int js = 0; js < NS; ++js) s += int_flux_diff[js] * (1.0 / lg.mu[js]);
      PUTL(F_CLEAR_AIR_ABS, 0.0 + air_ext * (1.0 - air_ssa) * (int_flux_dir[0] + s));
    }
    double veg_abs_dir = 0.0;
    if (NREG > 1) {
      double od[NREG];
      od_scaling<NREG>(g.veg_fsd, od);
      const double vssa = dcur.veg_ssa;
      double vaa = 0.0, va = 0.0;
      SS_UNROLL_FULL
      for (int jreg = 1; jreg < NREG; ++jreg) {
        double s = 0.0;
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) s += int_flux_diff[jreg * NS + js] * (1.0 / lg.mu[js]);
        vaa = vaa + air_ext * (1.0 - air_ssa) * (int_flux_dir[jreg] + s);
        veg_abs_dir = veg_abs_dir + g.veg_ext * (1.0 - vssa) * int_flux_dir[jreg] * od[jreg];
        va = va + g.veg_ext * (1.0 - vssa) * (int_flux_dir[jreg] + s) * od[jreg];
      }
      PUTL(F_VEG_AIR_ABS, vaa);
      PUTL(F_VEG_ABS_DIR, veg_abs_dir);
      PUTL(F_VEG_ABS, va);
    } else {
      PUTL(F_VEG_AIR_ABS, 0.0);
      PUTL(F_VEG_ABS_DIR, 0.0);
      PUTL(F_VEG_ABS, 0.0);
    }

    double wall_in_dir = 0.0;
    double npw_sum = 0.0;
    if (URBAN) {
      // walls (urban :790-802)
      double frac[NREG], np[NREG], npw[NREG], f_wall[NREG];
      region_fractions<URBAN, false, NREG>(g.veg_fraction, g.building_fraction, frac);
      norm_perimeters<URBAN, NREG>(L.cfg, g, np, npw);
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) {
        f_wall[r] = (frac[r] <= L.cfg.min_vegetation_fraction) ? 0.0 : npw[r] / (kPi * frac[r]);
        npw_sum += npw[r];
      }
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg)
        wall_in_dir = wall_in_dir + f_wall[jreg] * sin0 * int_flux_dir[jreg];
      double wall_in = wall_in_dir;
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg) {
        double s = 0.0;
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) s += int_flux_diff[jreg * NS + js] * lg.tan_ang[js];
        wall_in = wall_in + f_wall[jreg] * s;
      }
      PUTL(F_WALL_IN_DIR, wall_in_dir);
      PUTL(F_WALL_IN, wall_in);
      PUTL(F_WALL_NET, wall_in * (1.0 - dcur.wall_albedo));
    } else {
      // members the reference leaves at zero after zero() for forest tiles
      PUTL(F_ROOF_IN, 0.0); PUTL(F_ROOF_IN_DIR, 0.0); PUTL(F_ROOF_NET, 0.0);
      PUTL(F_WALL_IN, 0.0); PUTL(F_WALL_IN_DIR, 0.0); PUTL(F_WALL_NET, 0.0);
    }

    // sunlit fractions and the clear-sky direct beam (forest :630-649,
    // urban :405-411, :814-852); band itransp only, zero in the diffuse pass
    if (jb == itransp) {
      const double ae = dcur.air_ext_t;
      const double trans_dir_clear = exp(-ae * g.dz / cos_sza);
      const double ifdc = (ae > 0.0) ? flux_dn_dir_clear * (1.0 - trans_dir_clear) * cos_sza / ae
                                     : flux_dn_dir_clear * g.dz;
      if (URBAN) {
        const double nbf = 1.0 - g.building_fraction;
        const double f_wall_dir_clear =
            (nbf <= L.cfg.min_building_fraction) ? 0.0 : npw_sum / (kPi * nbf);
        const double wsf = 0.5 * wall_in_dir / ss_max(DBL_EPSILON, (f_wall_dir_clear * sin0 * ifdc));
        if (FSEL(F_WALL_SUNLIT_FRAC)) FSEL(F_WALL_SUNLIT_FRAC)[lay] = dirpass ? wsf : 0.0;
      } else {
        if (FSEL(F_ROOF_SUNLIT_FRAC)) FSEL(F_ROOF_SUNLIT_FRAC)[lay] = 0.0;
        if (FSEL(F_WALL_SUNLIT_FRAC)) FSEL(F_WALL_SUNLIT_FRAC)[lay] = 0.0;
      }
      if (FSEL(F_VEG_SUNLIT_FRAC)) {
        double vsf = 0.0;
        if (NREG > 1) {
          const double vadc = ifdc * g.veg_ext * (1.0 - dcur.veg_ssa_t) * g.veg_fraction;
          vsf = veg_abs_dir / ss_max(DBL_EPSILON, vadc);
        }
        FSEL(F_VEG_SUNLIT_FRAC)[lay] = dirpass ? vsf : 0.0;
      }
      flux_dn_dir_clear = flux_dn_dir_clear * trans_dir_clear;
    }
  }

  // ground (forest :661-676, urban :861-874)
  {
    double s = 0.0;
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r) s += dir_above[r];
    const double gdd = cos_sza * s;
    const double s2 = vec_sum<N>(diff_above), s3 = vec_sum<N>(up_above);
    PUTC(F_GROUND_DN_DIR, gdd);
    PUTC(F_GROUND_DN, gdd + s2);
    PUTC(F_GROUND_NET, (gdd + s2) - s3);
    double gvd = 0.0;
    SS_UNROLL_FULL
    for (int jreg = 0; jreg < NREG; ++jreg) {
      SS_UNROLL_FULL
      for (int js = 0; js < NS; ++js) {
        const int ifr = js + jreg * NS;
        gvd = gvd + (diff_above[ifr] + up_above[ifr]) * lg.tan_ang[js] / kPi;
      }
    }
    PUTC(F_GROUND_VERTICAL_DIFF, gvd);
    if (jb == itransp && FSEL(F_GROUND_SUNLIT_FRAC))
      FSEL(F_GROUND_SUNLIT_FRAC)[col] = dirpass ? gdd / (cos_sza * flux_dn_dir_clear) : 0.0;
  }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SS_NO_TRANSPOSED_OUT)
  // The staged per-layer fields of a wave that owns one contiguous block of the caller's arrays: transposed
  // (wave_owns_contiguous_block, ss_common.hpp) - element e of the block is layer e % nlay of column unit e / nlay
  // of the wave, i.e. of lane 2 (e / nlay) + pass.
  if constexpr (N <= SS_UNROLL_MAX) {
    int first_l0 = 0;
    if (wave_owns_contiguous_block(col, nlay, l0, nspec, L.maxlay, &first_l0)) {
      const int lane = (int)(threadIdx.x & 63), nelem = 32 * nlay;
      SS_UNROLL_FULL
      for (int pz = 0; pz < 2; ++pz) {
        SS_UNROLL_FULL
        for (int s_ = 0; s_ < kSwDownFields; ++s_) {
          double* p_ = (pz == 0 ? L.flux_a : L.flux_b).f[kSwDownSlotField[s_]];
          const bool zero_ = s_ >= kSwDownDiffuseFields && pz == 1;
          if (p_ && (zero_ || staged(s_))) {
            p_ += first_l0;
            for (int e = lane; e < nelem; e += 64) {
              const int cr = e / nlay, jl = e - cr * nlay;
              double v_ = 0.0;
              if (!zero_)
                v_ = s_ < kSwDownOwnFields ? obuf.get_at(s_ * L.maxlay + jl, 2 * cr + pz)
                                           : obuf.get_at(kSwDownDiffuseFields * L.maxlay + jl, 2 * cr + 1);
              p_[e] = v_;
            }
          }
        }
      }
      return;
    }
  }
#endif
  // the staged per-layer fields, all layers of a field back to back
  SS_UNROLL_FULL
  for (int s_ = 0; s_ < kSwDownFields; ++s_) {
    double* p_ = FSEL(kSwDownSlotField[s_]);
    const bool zero_ = s_ >= kSwDownDiffuseFields && !dirpass;
    if (p_ && (zero_ || staged(s_))) {
      for (int jlay = 0; jlay < nlay; ++jlay) {
        double v_ = 0.0;
        if (!zero_) v_ = s_ < kSwDownOwnFields ? obuf.get(s_ * L.maxlay + jlay) : obuf.get_next(kSwDownDiffuseFields * L.maxlay + jlay);
        p_[(long)jb + (long)nspec * (l0 + jlay)] = v_;
      }
    }
  }
#undef PUTC
#undef PUTL
#undef FSEL
}

}  // namespace ss
