// ss_column_lw.hpp - longwave column unit: adding method with sources (forest
// radsurf_forest_lw.F90:452-506, urban radsurf_urban_lw.F90:551-626) and the
// internal-emission and incoming-flux passes (forest :508-694, urban :628-858)
// of one (column, band): k_up stores the adding-method matrices of every layer,
// k_down runs the two passes on paired lanes (same structure as ss_column_sw.hpp).
#pragma once

#include "ss_common.hpp"
#include "ss_geometry.hpp"
#include "ss_layer_matrices.hpp"
#include "ss_linalg.hpp"
#include "ss_solver_lw.hpp"
#include "ss_pack.hpp"

namespace ss {

template <int N>
struct LwColRec {
  static constexpr int PDN = 0;                       // P = Den^-1 T      N x N   (Den = I - a_above R)
  static constexpr int ABE = PDN + rec_even(N * N);   // a_below(l+1)      N x N (canopy part)
  // (the small orders store D^-1 a_below D as a packed lower triangle: see SwColRec)
  static constexpr bool PACKED = sym_packed_col(N, true);
  static constexpr int NAB = PACKED ? tri_count(N) : N * N;
  static constexpr int SBE = ABE + rec_even(NAB);     // source_below(l+1) N (canopy part)
  static constexpr int RSD = SBE + rec_even(N);       // Den^-1 (R source_above + source_lay)  N
  // fused layer + upward kernel (fused_order(N)): what the downward sweep reads of the layer
  // matrices travels here - int_flux_mat (N x N) and int_source (N)
  static constexpr bool FUSED = fused_order(N, true);
  static constexpr int IFLUX = RSD + rec_even(N);
  static constexpr int ISRC = IFLUX + rec_even(N * N);
  static constexpr int NELEM = FUSED ? ISRC + rec_even(N) : IFLUX;
  // flux at the layer base in the downward sweep = P flux_top + RSD (forest :547-553)
};

// denominator LU, a_below(l+1), source_below(l+1) (canopy parts) of layer l
template <int N>
SS_HD void lw_adding_step(const double* Aab, const double* Sab, const double* R, const double* T,
                          const double* Sl, double* Den, double* Abe, double* Sbe) {
  mat_mul<N, N, N>(Aab, R, Den);
  SS_UNROLL_NN(N)
  for (int k = 0; k < N * N; ++k) Den[k] = -Den[k];
  SS_UNROLL_N(N)
  for (int j = 0; j < N; ++j) SS_E(Den, j, j, N) = 1.0 + SS_E(Den, j, j, N);
  lu_factor<N>(Den);
  {
    double X[N * N];
    mat_mul<N, N, N>(Aab, T, X);
    lu_solve_mat<N, N>(Den, X);
    mat_mul<N, N, N>(T, X, Abe);
    SS_UNROLL_NN(N)
    for (int k = 0; k < N * N; ++k) Abe[k] = R[k] + Abe[k];
  }
  // Eq. 34: source_below = source_lay + T Den^-1 (source_above + a_above source_lay)
  {
    double v[N];
    mat_vec<N, N>(Aab, Sl, v);
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) v[i] = Sab[i] + v[i];
    lu_solve<N>(Den, v);
    mat_vec<N, N>(T, v, Sbe);
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) Sbe[i] = Sl[i] + Sbe[i];
  }
}

// ---- section 4: adding method with sources, upward ----
// a_above, source_above at the ground (forest :455-466, urban :554-565)
template <bool URBAN, int NREG, int NS>
SS_HD void lw_ground(const Launch& L, long cidx, int l0, double* Aab, double* Sab) {
  constexpr int N = NREG * NS;
  const Inputs& in = L.in;
  const LG& lg = L.lg;
  mat_zero<N, N>(Aab);
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

// Adding step of one layer: R, T, source_lay from "src"; stores a_below, source_below and the
// operators of the downward sweep into the column record
// dsc, rd: scaling of the layer's components and its reciprocal (layer_scaling; read where the records are packed)
template <int N, class Src>
SS_HD void lw_up_step(const Src& src, const double* Aab, const double* Sab, double* crec, double* Abe, double* Sbe,
                      const double* dsc, const double* rd) {
  using LRec = LwLayRec<N>;
  using CRec = LwColRec<N>;
  constexpr long stride = kRecStride;
  double R[N * N], T[N * N], Den[N * N], Sl[N];
  if constexpr (LRec::PACKED) {
    load_packed_unscaled<N>(src, LRec::R, dsc, rd, R);
    load_packed_unscaled<N>(src, LRec::T, dsc, rd, T);
  } else {
    src.template load<N * N>(LRec::R, R);
    src.template load<N * N>(LRec::T, T);
  }
  src.template load<N>(LRec::SRC, Sl);
  lw_adding_step<N>(Aab, Sab, R, T, Sl, Den, Abe, Sbe);
  if constexpr (CRec::PACKED) {
    store_packed_scaled<N>(crec, stride, CRec::ABE, Abe, dsc, rd);
  } else
  rec_store<N * N>(crec, stride, CRec::ABE, Abe);
  rec_store<N>(crec, stride, CRec::SBE, Sbe);
  // operators of the downward sweep
  lu_solve_mat<N, N>(Den, T);                 // P = Den^-1 T (T is dead afterwards)
  rec_store<N * N>(crec, stride, CRec::PDN, T);
  double rs[N];
  mat_vec<N, N>(R, Sab, rs);
  SS_UNROLL_N(N)
  for (int i = 0; i < N; ++i) rs[i] = rs[i] + Sl[i];
  lu_solve<N>(Den, rs);
  rec_store<N>(crec, stride, CRec::RSD, rs);
}

// Overlap to the base of the layer above layer jlay, or to the free atmosphere (forest :495-506,
// urban :583-626)
template <bool URBAN, int NREG, int NS>
SS_HD void lw_overlap_above(const Launch& L, int cu, int jlay, int nlay, const double* Abe, const double* Sbe,
                            double* Aab, double* Sab) {
  constexpr int N = NREG * NS;
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  const LG& lg = L.lg;
  double U[NREG * DB], V[DB * NREG];
  const double* prec = pack_rec(L, jlay, cu, kPackLw);
  double vf_lo, bf_lo, vf_up = 0.0, bf_up = 0.0;
  pack_fractions(prec, vf_lo, bf_lo);
  if (jlay + 1 < nlay) pack_fractions(pack_rec(L, jlay + 1, cu, kPackLw), vf_up, bf_up);
  interface_overlap_vals<URBAN, true, NREG>(L.cfg.min_vegetation_fraction, jlay + 1 >= nlay, vf_up, bf_up,
                                            false, vf_lo, bf_lo, U, V);
  double roof_a = 0.0, roof_emis = 0.0, exposed_roof_frac = 0.0;
  if (URBAN) {
    // urban :587-605
    if (jlay < nlay - 1) exposed_roof_frac = ss_max(0.0, bf_lo - bf_up);
    else exposed_roof_frac = bf_lo;
    const RecPair re = rec_pair_load(prec + 14 * kRecStride);   // roof_emissivity, roof_emission
    roof_a = 1.0 - re.x;
    roof_emis = re.y;
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

template <int NREG, int NS>
SS_HD void lw_top(const Launch& L, long cidx, const double* Aab, const double* Sab) {
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
  L.top1[cidx] = 1.0 - s;           // top_emissivity
  s = 0.0;
  SS_UNROLL_FULL
  for (int j1 = 0; j1 < NS; ++j1) s += Sab[j1];
  L.top2[cidx] = s;                 // top_emission
}

// The upward sweep in the reference's (unscaled) coordinates: orders whose records are not packed
// (k_column dispatches in ss_column_lw_sym.hpp)
template <bool URBAN, int NREG, int NS>
SS_HD void column_up_lw_full(const Launch& L, int cu) {
  constexpr int N = NREG * NS;
  using LRec = LwLayRec<N>;
  using CRec = LwColRec<N>;
  const Inputs& in = L.in;
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  const int nlay = in.nlay[col];
  const int l0 = in.istartlay[col] - 1;
  const long cidx = (long)jb + (long)L.nspec * col;
  double Aab[N * N], Sab[N];
  lw_ground<URBAN, NREG, NS>(L, cidx, l0, Aab, Sab);
  for (int jlay = 0; jlay < nlay; ++jlay) {
    const RecSource src{L.lay + rec_base(jlay, L.maxlay, LRec::NELEM, cu), kRecStride};
    double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    double Abe[N * N], Sbe[N];
    double dsc[N], rd[N];
    if constexpr (LRec::PACKED || CRec::PACKED) {
      double vf, bf, frac[NREG];
      pack_fractions(pack_rec(L, jlay, cu, kPackLw), vf, bf);
      region_fractions<URBAN, true, NREG>(vf, bf, frac);
      layer_scaling<NREG, NS>(L.lg, frac, dsc, rd);
    }
    lw_up_step<N>(src, Aab, Sab, crec, Abe, Sbe, dsc, rd);
    lw_overlap_above<URBAN, NREG, NS>(L, cu, jlay, nlay, Abe, Sbe, Aab, Sab);
  }
  lw_top<NREG, NS>(L, cidx, Aab, Sab);
}

// Writer of the fused longwave layer stage: R, T -> LDS slots; source_lay -> registers (hold);
// int_flux_mat and int_source -> column record (read by the downward sweep only)
template <int N, class Lds>
struct LwFusedWriter {
  using LRec = LwRec<N>;
  using CRec = LwColRec<N>;
  const Lds* lds;
  double* hold;           // rec_even(N) elements: source_lay
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
      hold[e - LRec::SRC] = v0;
      hold[e + 1 - LRec::SRC] = v1;
    } else {
      rec_pair_store(crec + (long)(CRec::ISRC + e - LRec::ISRC) * stride, v0, v1);
    }
  }
  SS_HD const Lds& scratch() const { return *lds; }   // the slots are free until R and T are put
};

// Layer matrices and adding step of every layer of a column unit on one lane (k_layer_up, see
// the header of the fused kernels in ss_layer_matrices.hpp)
template <bool URBAN, int NREG, int NS>
SS_HD void layer_up_lw_full(const Launch& L, int cu) {
  constexpr int N = NREG * NS;
  using LRec = LwRec<N>;
  using CRec = LwColRec<N>;
  static_assert(CRec::FUSED && LRec::IFLUX <= kLaneLdsSlots, "fused form needs R and T in the lane's LDS slots");
  using Lds = LaneLds<true>;
  constexpr long stride = kRecStride;
  const Inputs& in = L.in;
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  const int nlay = in.nlay[col];
  const int l0 = in.istartlay[col] - 1;
  const long cidx = (long)jb + (long)L.nspec * col;
  const Lds lds{};
  for (int jlay = 0; jlay < nlay; ++jlay) {
    double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    double hold[LRec::ISRC - LRec::SRC];
    const LwFusedWriter<N, Lds> w{&lds, hold, crec, stride, (unsigned long long*)L.eigen_fail};
    layer_unit_lw<URBAN, NREG, NS>(L, cu, jlay, w);
    double Aab[N * N], Sab[N];
    if (jlay == 0) {
      lw_ground<URBAN, NREG, NS>(L, cidx, l0, Aab, Sab);
    } else {
      const double* prev = L.col + rec_base(jlay - 1, L.maxlay, CRec::NELEM, cu);
      double Abe[N * N], Sbe[N];
      rec_load<N * N>(prev, stride, CRec::ABE, Abe);
      rec_load<N>(prev, stride, CRec::SBE, Sbe);
      lw_overlap_above<URBAN, NREG, NS>(L, cu, jlay - 1, nlay, Abe, Sbe, Aab, Sab);
    }
    const FusedSource<LRec::SRC, Lds> src{&lds, hold};
    double Abe[N * N], Sbe[N];
    lw_up_step<N>(src, Aab, Sab, crec, Abe, Sbe, nullptr, nullptr);
  }
  if (nlay > 0) {
    const double* prev = L.col + rec_base(nlay - 1, L.maxlay, CRec::NELEM, cu);
    double Abe[N * N], Sbe[N], Aab[N * N], Sab[N];
    rec_load<N * N>(prev, stride, CRec::ABE, Abe);
    rec_load<N>(prev, stride, CRec::SBE, Sbe);
    lw_overlap_above<URBAN, NREG, NS>(L, cu, nlay - 1, nlay, Abe, Sbe, Aab, Sab);
    lw_top<NREG, NS>(L, cidx, Aab, Sab);
  }
}

// Value held by the partner lane (the other pass of the same column unit).  On
// the host the two passes run one after the other and hand the value over
// through "slot".
SS_HD double lw_partner_value(double v, int pass, double* slot) {
#if defined(__HIP_DEVICE_COMPILE__)
  (void)pass; (void)slot;
  return __shfl_xor(v, 1);
#else
  if (pass == 0) { *slot = v; return 0.0; }
  return *slot;
#endif
}

// ---- section 5: downward sweep (k_down) ----
// One lane = (column, band, pass): pass 0 = internal emission (lw_internal,
// forest :508-610 / urban :628-752), pass 1 = incoming flux at canopy top
// (lw_norm, forest :612-694 / urban :754-858).  Pass 1 is pass 0 with every
// source switched off and a unit incoming flux, so both run one instruction
// stream and share their memory requests; software-pipelined like the
// shortwave sweep (ss_column_sw.hpp).
// Staging priority of the per-layer outputs of the downward kernel (see PUTL there)
constexpr int lw_down_slot(int F) {
  switch (F) {
    case F_CLEAR_AIR_ABS: return 0;
    case F_VEG_ABS: return 1;
    case F_VEG_AIR_ABS: return 2;
    case F_WALL_IN: return 3;
    case F_WALL_NET: return 4;
    case F_ROOF_IN: return 5;
    case F_ROOF_NET: return 6;
    default: return -1;
  }
}
constexpr int kLwDownSlotField[7] = {F_CLEAR_AIR_ABS, F_VEG_ABS, F_VEG_AIR_ABS, F_WALL_IN, F_WALL_NET,
                                     F_ROOF_IN, F_ROOF_NET};
constexpr bool lw_down_slots_consistent() {
  for (int s = 0; s < 7; ++s)
    if (lw_down_slot(kLwDownSlotField[s]) != s) return false;
  return true;
}
static_assert(lw_down_slots_consistent(), "kLwDownSlotField must be the inverse of lw_down_slot");

template <bool URBAN, int NREG, int NS>
SS_HD void column_down_lw(const Launch& L, int cu, int pass, double* slot) {
  constexpr int N = NREG * NS;
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  constexpr int NB = DB * NS;
  using LRec = LwLayRec<N>;
  using CRec = LwColRec<N>;
  constexpr long st = kRecStride;
  const Inputs& in = L.in;
  const LG& lg = L.lg;
  // a_below of a layer: the column record holds it packed and scaled for the small orders; NABE doubles are
  // loaded as they lie and applied through abe_matvec (vf, bf: the fractions of the layer it belongs to)
  constexpr int NABE = CRec::PACKED ? rec_even(tri_count(N)) : N * N;
  auto abe_matvec = [&](const double* ABE_, double vf_, double bf_, const double* x_, double* y_) {
    if constexpr (CRec::PACKED) {
      double frac_[NREG], dsc_[N], rd_[N];
      region_fractions<URBAN, true, NREG>(vf_, bf_, frac_);
      layer_scaling<NREG, NS>(lg, frac_, dsc_, rd_);
      packed_scaled_matvec<N>(ABE_, dsc_, rd_, x_, y_);
    } else {
      mat_vec<N, N>(ABE_, x_, y_);
    }
  };
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  const int nspec = L.nspec;
  const int nlay = in.nlay[col];
  const int l0 = in.istartlay[col] - 1;
  const long cidx = (long)jb + (long)nspec * col;
  const bool internal = (pass == 0);
  const double src = internal ? 1.0 : 0.0;     // switches the emission terms
  const double top_emissivity = L.top1[cidx], top_emission = L.top2[cidx];

#define FSEL(F) (internal ? L.flux_a.f[F] : L.flux_b.f[F])
#define PUTC(F, v) do { double* p_ = FSEL(F); if (p_) p_[cidx] = (v); } while (0)
  // per-layer outputs staged in LDS and written out after the sweep (see ss_column_sw.hpp)
  LaneLds<(N <= SS_UNROLL_MAX)> obuf;
  const int nbuf = obuf.kSlots / (L.maxlay > 0 ? L.maxlay : 1);
#define PUTL(F, v) do { double* p_ = FSEL(F); if (p_) { constexpr int s_ = lw_down_slot(F); \
    if (s_ >= 0 && s_ < nbuf) obuf.put(s_ * L.maxlay + jlay, (v)); else p_[sidx] = (v); } } while (0)

  double dn_above[N], up_above[N];
  vec_zero<N>(dn_above);
  vec_zero<N>(up_above);
  if (!internal) {
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) dn_above[js] = lg.hweight[js];
  }
  PUTC(F_TOP_DN, internal ? 0.0 : 1.0);
  PUTC(F_TOP_NET, internal ? -top_emission : top_emissivity);

  double emiss_factor = 0.0;
  SS_UNROLL_FULL
  for (int js = 0; js < NS; ++js) emiss_factor += lg.hweight[js] / lg.mu[js];
  emiss_factor = 2.0 * emiss_factor;

  // Crossing the interface at the top of layer jl (see ss_column_sw.hpp): downward
  // flux through V, upward flux just below it from a_below / source_below of layer
  // jl and its exposed roofs, upward flux just above it through U - which is the
  // reference's a_above flux + source_above at the base of the layer above, with
  // a_above = (U x I) [a_below (V x I); roof rows] applied to the vector instead of
  // being stored as a matrix by the upward sweep.
  auto cross = [&](const LwLayerIn<URBAN, NREG>& dlo, bool upper_is_sky, double vf_up, double bf_up,
                   const double* ABE, const double* SBE, const double* dn_ab, double* dn_below,
                   double* up_below, double* up_ab) {
    double U[NREG * DB], V[DB * NREG];
    interface_overlap_vals<URBAN, true, NREG>(L.cfg.min_vegetation_fraction, upper_is_sky, vf_up, bf_up,
                                              false, dlo.g.veg_fraction, dlo.g.building_fraction, U, V);
    apply_V_vec<NREG, DB, NS>(V, dn_ab, dn_below);
    abe_matvec(ABE, dlo.g.veg_fraction, dlo.g.building_fraction, dn_below, up_below);
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) up_below[i] = up_below[i] + src * SBE[i];
    if (URBAN) {
      const double roof_a = 1.0 - dlo.roof_emissivity;
      double exposed_roof_frac;
      if (!upper_is_sky) exposed_roof_frac = ss_max(0.0, dlo.g.building_fraction - bf_up);
      else exposed_roof_frac = dlo.g.building_fraction;
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) {
        double acc = 0.0;
        SS_UNROLL_FULL
        for (int s2 = 0; s2 < NS; ++s2) acc += (roof_a * lg.hweight[s]) * dn_below[N + s2];
        up_below[N + s] = acc + src * (lg.hweight[s] * dlo.roof_emission * exposed_roof_frac);
      }
    }
    if (up_ab) apply_U_mat<NREG, DB, NS, 1>(U, up_below, up_below + N, up_ab);
  };

  // top layer: its scalars and group 1 (a_below, source_below), then the interface to the free atmosphere
  LwLayerIn<URBAN, NREG> din;
  double dn_below[NB], up_below[NB];
  if (nlay > 0) {
    const int jlay = nlay - 1;
    const double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    double ABE[NABE], SBE[N];
    din.load_pack(pack_rec(L, jlay, cu, kPackLw));
    rec_load<NABE>(crec, st, CRec::ABE, ABE);
    rec_load<N>(crec, st, CRec::SBE, SBE);
    SS_LOAD_FENCE();
    cross(din, true, 0.0, 0.0, ABE, SBE, dn_above, dn_below, up_below, nullptr);
  }

  for (int jlay = nlay - 1; jlay >= 0; --jlay) {
    const double* lrec = L.lay + rec_base(jlay, L.maxlay, LRec::NELEM, cu);
    const double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    const int lay = l0 + jlay;
    const long sidx = (long)jb + (long)nspec * lay;
    const LwLayerIn<URBAN, NREG> dcur = din;

    // ---- issue group 2 (P, RSD), work on the fluxes just below the top interface ----
    double P[N * N], RSD[N];
    rec_load<N * N>(crec, st, CRec::PDN, P);
    rec_load<N>(crec, st, CRec::RSD, RSD);
    SS_LOAD_FENCE();

    if (URBAN) {
      double sdn = 0.0, sup = 0.0;
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) { sdn += dn_below[N + s]; sup += up_below[N + s]; }
      PUTL(F_ROOF_IN, sdn);
      PUTL(F_ROOF_NET, sdn - sup);
    } else {
      PUTL(F_ROOF_IN, 0.0);
      PUTL(F_ROOF_NET, 0.0);
    }

    // ---- issue group 3 (a_below, source_below and the scalars of the layer below), work on group 2 ----
    double ABE[NABE], SBE[N];
    if (jlay > 0) {
      const double* crec_n = L.col + rec_base(jlay - 1, L.maxlay, CRec::NELEM, cu);
      din.load_pack(pack_rec(L, jlay - 1, cu, kPackLw));
      rec_load<NABE>(crec_n, st, CRec::ABE, ABE);
      rec_load<N>(crec_n, st, CRec::SBE, SBE);
    }
    SS_LOAD_FENCE();

    // flux at the base of the layer (forest :547-553, :633-636)
    double conv[N];
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) conv[i] = dn_below[i];
    const double s_top_dn = vec_sum<N>(dn_below), s_top_up = vec_sum<N>(up_below);
    mat_vec<N, N>(P, dn_below, dn_above);
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) dn_above[i] = dn_above[i] + src * RSD[i];

    // ---- issue group 4 (int_flux), work on group 3: the interface at the base ----
    double IFLUX[N * N], ISRC[N];
    if (CRec::FUSED) {
      rec_load<N * N>(crec, st, CRec::IFLUX, IFLUX);
      rec_load<N>(crec, st, CRec::ISRC, ISRC);
    } else {
      rec_load<N * N>(lrec, st, LRec::IFLUX, IFLUX);
      rec_load<N>(lrec, st, LRec::ISRC, ISRC);
    }
    SS_LOAD_FENCE();

    if (jlay > 0) {
      cross(din, false, dcur.g.veg_fraction, dcur.g.building_fraction, ABE, SBE, dn_above, dn_below,
            up_below, up_above);
    } else {
      // the ground (forest :464-470, urban :560-567): grey Lambertian emitter, per region
      const double gem = in.ground_emissivity[cidx], gemission = in.ground_emission[cidx];
      double frac1[NREG];
      region_fractions<URBAN, true, NREG>(dcur.g.veg_fraction, dcur.g.building_fraction, frac1);
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg) {
        double sd = 0.0;
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) sd += dn_above[js + jreg * NS];
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js)
          up_above[js + jreg * NS] = ((1.0 - gem) * lg.hweight[js]) * sd
                                     + src * ((lg.hweight[js] * frac1[jreg]) * gemission);
      }
    }

    if (FSEL(F_FLUX_DN_LAYER_TOP)) {
      PUTL(F_FLUX_DN_LAYER_TOP, s_top_dn);
      PUTL(F_FLUX_UP_LAYER_TOP, s_top_up);
      PUTL(F_FLUX_DN_LAYER_BASE, vec_sum<N>(dn_above));
      PUTL(F_FLUX_UP_LAYER_BASE, vec_sum<N>(up_above));
    }
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) conv[i] = conv[i] + up_above[i];

    double ifl[N];
    mat_vec<N, N>(IFLUX, conv, ifl);
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) ifl[i] = ifl[i] + src * ISRC[i];

    // absorption (forest :571-590 / :658-672, urban :700-737 / :811-835)
    LwLayerProps<URBAN, NREG> p;
    lw_layer_props<URBAN, NREG>(L, dcur, p);
    const double air_ext = dcur.air_ext, air_ssa = dcur.air_ssa;
    {
      double sa = 0.0;
      SS_UNROLL_FULL
      for (int js = 0; js < NS; ++js) sa += ifl[js] * (1.0 / lg.mu[js]);
      const double volume_emiss = p.frac[0] * (p.ext_reg[0] * (1.0 - p.ssa_reg[0]) * p.planck_reg[0]);
      PUTL(F_CLEAR_AIR_ABS, 0.0 + air_ext * (1.0 - air_ssa) * sa - src * ((emiss_factor * volume_emiss) * p.g.dz));
    }
    if (NREG > 1) {
      double vaa = 0.0, va = 0.0;
      SS_UNROLL_FULL
      for (int jreg = 1; jreg < NREG; ++jreg) {
        double sa = 0.0;
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) sa += ifl[jreg * NS + js] * (1.0 / lg.mu[js]);
        const double emiss_air = emiss_factor * p.frac[jreg] * p.ext_reg[0] * (1.0 - p.ssa_reg[0]) * p.vaplanck;
        const double emiss_veg = emiss_factor * p.frac[jreg] * p.g.veg_ext * (1.0 - p.vssa) * p.vplanck * p.od[jreg];
        vaa = vaa + air_ext * (1.0 - air_ssa) * sa - src * (emiss_air * p.g.dz);
        va = va + p.g.veg_ext * (1.0 - p.vssa) * sa * p.od[jreg] - src * (emiss_veg * p.g.dz);
      }
      PUTL(F_VEG_AIR_ABS, vaa);
      PUTL(F_VEG_ABS, va);
    } else {
      PUTL(F_VEG_AIR_ABS, 0.0);
      PUTL(F_VEG_ABS, 0.0);
    }
    if (URBAN) {
      double wall_in = 0.0, npw_sum = 0.0;
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg) {
        double sa = 0.0;
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) sa += ifl[jreg * NS + js] * lg.tan_ang[js];
        wall_in = wall_in + p.f_wall[jreg] * sa;
        npw_sum += p.npw[jreg];
      }
      const double emiss_wall = (npw_sum * lg.vadjustment) * dcur.wall_emission;
      PUTL(F_WALL_IN, wall_in);
      PUTL(F_WALL_NET, wall_in * dcur.wall_emissivity - src * (emiss_wall * p.g.dz));
    } else {
      PUTL(F_WALL_IN, 0.0);
      PUTL(F_WALL_NET, 0.0);
    }
  }

  // ground (forest :599-610 / :680-694, urban :741-752 / :845-858)
  {
    const double s2 = vec_sum<N>(dn_above), s3 = vec_sum<N>(up_above);
    PUTC(F_GROUND_DN, s2);
    PUTC(F_GROUND_NET, s2 - s3);
    // own accumulation (urban: starts from zero in both passes)
    double gvd_own = 0.0;
    SS_UNROLL_FULL
    for (int jreg = 0; jreg < NREG; ++jreg) {
      SS_UNROLL_FULL
      for (int js = 0; js < NS; ++js) {
        const int ifr = js + jreg * NS;
        gvd_own = gvd_own + (dn_above[ifr] + up_above[ifr]) * lg.tan_ang[js] / kPi;
      }
    }
    if (URBAN) {
      PUTC(F_GROUND_VERTICAL_DIFF, gvd_own);
    } else {
      // forest: the incoming pass accumulates ON TOP of the internal pass's value,
      // into lw_internal, and lw_norm%ground_vertical_diff stays zero
      // (radsurf_forest_lw.F90:691): the incoming-pass lane takes the internal
      // value from its partner lane and writes both members
      const double gvda = lw_partner_value(gvd_own, pass, slot);
      if (!internal) {
        double gvd2 = gvda;
        SS_UNROLL_FULL
        for (int jreg = 0; jreg < NREG; ++jreg) {
          SS_UNROLL_FULL
          for (int js = 0; js < NS; ++js) {
            const int ifr = js + jreg * NS;
            gvd2 = gvd2 + (dn_above[ifr] + up_above[ifr]) * lg.tan_ang[js] / kPi;
          }
        }
        if (L.flux_a.f[F_GROUND_VERTICAL_DIFF]) L.flux_a.f[F_GROUND_VERTICAL_DIFF][cidx] = gvd2;
        if (L.flux_b.f[F_GROUND_VERTICAL_DIFF]) L.flux_b.f[F_GROUND_VERTICAL_DIFF][cidx] = 0.0;
      }
    }
  }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SS_NO_TRANSPOSED_OUT)
  // a wave that owns one contiguous block of the caller's arrays writes it with transposed stores (see column_down_sw)
  if constexpr (N <= SS_UNROLL_MAX) {
    int first_l0 = 0;
    if (wave_owns_contiguous_block(col, nlay, l0, nspec, L.maxlay, &first_l0)) {
      const int lane = (int)(threadIdx.x & 63), nelem = 32 * nlay;
      SS_UNROLL_FULL
      for (int pz = 0; pz < 2; ++pz) {
        SS_UNROLL_FULL
        for (int s_ = 0; s_ < 7; ++s_) {
          double* p_ = (pz == 0 ? L.flux_a : L.flux_b).f[kLwDownSlotField[s_]];
          if (s_ < nbuf && p_) {
            p_ += first_l0;
            for (int e = lane; e < nelem; e += 64) {
              const int cr = e / nlay, jl = e - cr * nlay;
              p_[e] = obuf.get_at(s_ * L.maxlay + jl, 2 * cr + pz);
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
  for (int s_ = 0; s_ < 7; ++s_) {
    double* p_ = FSEL(kLwDownSlotField[s_]);
    if (s_ < nbuf && p_) {
      for (int jlay = 0; jlay < nlay; ++jlay)
        p_[(long)jb + (long)nspec * (l0 + jlay)] = obuf.get(s_ * L.maxlay + jlay);
    }
  }
#undef PUTC
#undef PUTL
#undef FSEL
}

}  // namespace ss
