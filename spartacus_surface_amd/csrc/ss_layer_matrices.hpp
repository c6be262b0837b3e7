// ss_layer_matrices.hpp - per-lane layer matrices of one (column, band, layer)
// unit: calc_matrices_sw_eig (radtool/radtool_calc_matrices_sw_eig.F90:30-298)
// and calc_matrices_lw_eig (radtool/radtool_calc_matrices_lw_eig.F90:32-230),
// with schur_invert_sw (radtool/radtool_schur.F90:32-53).
//
// Same mathematics as the reference (Hogan 2019 Eqs. 45-49, 56-59), re-ordered
// for a register-resident lane and using structure the reference does not:
//  * the eigen-decomposition of P = (G1 - G2)(G1 + G2) goes through the hidden
//    symmetry of the Gamma matrices where it can (ss_eigen_sym.hpp), which also
//    yields (G1 - G2)^-1 X from the Cholesky factor;
//  * the direct-beam particular solutions g3, g4 (sw_eig :231-253) come from that
//    eigen-decomposition ((P - mu^2) p = -2 (G1 - G2) s, diffuse_part) instead of
//    two LU factorisations, an inverse and two products per direct eigenvalue;
//  * int_diff = -(G1 + G2)^-1 and inv_gamma_b = -(G1 + G2)^-1 b by the identity
//    G1 - G2 G1^-1 G2 = (G1 - G2) G1^-1 (G1 + G2) (radtool_schur.F90:45-51,
//    lw_eig :178-195);
//  * the (2n+d) x (2n+d) system of direct_diffuse_part (:303-386) is never
//    formed: its matrix is block upper-triangular,
//        [ G1   G2D  g3*e ]   [c1]   [0]
//        [ G2D  G1   g4   ] * [c2] = [0]      => c3 = G0^-1
//        [ 0    0    G0   ]   [c3]   [I]
//    and the remaining 2n x 2n block has exactly the structure already solved
//    for the diffuse R/T (Schur complement S = G1 - G2D G1^-1 G2D, whose
//    inverse is C'_lower).  Unpivoted LU of the full matrix (what the reference
//    does) performs this very elimination, so results agree to rounding while
//    the 15x15 (n=6) / 27x27 (n=12) matrix and its 2/3 N^3 flops disappear;
//  * every LU factorisation is computed once and reused.
// Results are written straight to the HBM scratch record of the unit through a
// writer, embedded in the full (NF x NF) layout with zero fill when the unit is
// a clear-only or vegetation-only sub-problem (radsurf_forest_sw.F90:408-455).
#pragma once

#include "ss_common.hpp"
#include "ss_linalg.hpp"
#include "ss_eigen.hpp"
#include "ss_eigen_sym.hpp"

namespace ss {

// Element offsets inside one layer record of the scratch.  Scratch layout:
// [tile of 64 column-units][layer][element pair][lane][2]: everything one
// wavefront reads or writes for its 64 units is one contiguous block (all of a
// wave's accesses fall into a few DRAM/TLB pages instead of one page per
// element), and elements 2p and 2p+1 of a unit are adjacent, so that every lane
// moves 16 bytes per access (global_load/store_dwordx4).  Every block of a
// record therefore starts at an even element.
constexpr int rec_even(int n) { return (n + 1) & ~1; }

template <int NF, int DF>
struct SwRec {
  static constexpr int R = 0;                              // ref_diff        NF x NF
  static constexpr int T = R + rec_even(NF * NF);          // trans_diff      NF x NF
  static constexpr int SUP = T + rec_even(NF * NF);        // ref_dir         NF x DF
  static constexpr int SDN = SUP + rec_even(NF * DF);      // trans_dir_diff  NF x DF
  static constexpr int EDIR = SDN + rec_even(NF * DF);     // trans_dir_dir   DF x DF
  static constexpr int IDIR = EDIR + rec_even(DF * DF);    // int_dir         DF x DF
  // int_diff (NF x NF) and int_dir_diff (NF x DF) are not stored: the downward sweep
  // applies them as one solve with Gamma1 + Gamma2 (sw_int_flux_system, ss_solver_sw.hpp)
  static constexpr int NELEM = IDIR + rec_even(DF * DF);
};

template <int NF>
struct LwRec {
  static constexpr int R = 0;                              // ref       NF x NF
  static constexpr int T = R + rec_even(NF * NF);          // trans     NF x NF
  static constexpr int IFLUX = T + rec_even(NF * NF);      // int_flux_mat NF x NF
  static constexpr int SRC = IFLUX + rec_even(NF * NF);    // source_lay   NF
  static constexpr int ISRC = SRC + rec_even(NF);          // int_source   NF
  static constexpr int NELEM = ISRC + rec_even(NF);
};

// orders above those of the row-of-16 kernels (3 regions x 7 or 8 streams): the symmetrised route with rolled
// loops and its matrices in scratch memory, in place of the general eigen-solver (-DSS_NO_SYM_SCRATCH: off)
constexpr bool sym_scratch_order(int n) {
#if defined(SS_NO_SYM_SCRATCH) || defined(SS_NO_SYM_LAYER)
  return false && n > 0;
#else
  return n > 16;   // kCoopMaxOrder of ss_eigen_coop.hpp
#endif
}

// Layer records of the solver pipeline.  Orders whose layer matrices are computed in symmetrised coordinates
// (3 <= NF <= SS_UNROLL_MAX: ss_layer_sym.hpp) store R^ = D^-1 R D and T^ = D^-1 T D - symmetric by
// reciprocity - as packed lower triangles, NF (NF + 1) / 2 doubles instead of NF^2 (n = 6: the shortwave
// record 128 -> 98 doubles, the longwave one 120 -> 90); the upward sweep restores R = D R^ D^-1 with the
// scaling of its layer.  Every other order keeps the layout of SwRec / LwRec, which also remains the
// layout of the general route (ss_layer_matrices.hpp, the kernel-level entries, the cold path).
// Layer record (R^, T^): packed for both spectral regions.  The shortwave upward sweep then runs in the
// scaled coordinates (ss_column_sw_sym.hpp); restoring R = D R^ D^-1 into full matrices next to a_above,
// d_above, S_up, S_dn was measured and costs more than the bytes save (88 -> 121 spilled dwords, upward sweep
// 3.81 -> 5.56 ms per step).  -DSS_NO_SYM_PACK_SW: full shortwave records, unscaled sweep.
constexpr bool sym_packed(int nf, bool lw) {
#if defined(SS_NO_SYM_LAYER) || defined(SS_FUSE_SW) || defined(SS_FUSE_LW)
  return false && nf > 0 && lw;
#elif defined(SS_NO_SYM_PACK_SW)
  return lw && nf >= 3 && nf <= SS_UNROLL_MAX;
#else
  return nf >= 3 && nf <= SS_UNROLL_MAX && (lw || true);
#endif
}
// Column record (a_below): packed and scaled.  The downward sweeps apply it to their flux vector as
// D (a_below^ (D^-1 x)) without restoring the matrix (packed_scaled_matvec): restoring it into 36 doubles first
// was measured on the shortwave sweep and costs more than the bytes save (4.13 -> 4.41 ms per step).
#if defined(SS_NO_SYM_PACK_COL_SW)
constexpr bool sym_packed_col(int nf, bool lw) { return lw && sym_packed(nf, lw); }
#else
constexpr bool sym_packed_col(int nf, bool lw) { return sym_packed(nf, lw); }
#endif
// Shortwave column record of the small orders: the two direct-beam blocks of a layer, Q (N x D: diffuse flux at the
// base per unit of direct flux in each region at the top) and d_below (N x D), are only ever applied to ONE
// vector in the downward sweep - the direct flux at the top of that layer in the pass normalised by the direct
// beam -, and that vector follows from the direct transmittances E and the overlap rules alone.  The upward
// sweep therefore runs that recursion first (top down, 3 x 3 work) and stores Q dir_top and d_below dir_top,
// N numbers each, instead of the 2 N D of the blocks: 94 -> 74 doubles per unit-layer written and read
// (-DSS_NO_SW_DIRVEC: the blocks).
constexpr bool fused_order(int n, bool lw);
constexpr bool sw_dirvec(int n) {
#if defined(SS_NO_SW_DIRVEC)
  return false && n > 0;
#elif defined(SS_SYM_FUSE_SW2)
  return sym_packed_col(n, false);      // the fused layer + upward kernel of round 4 carries the direct beam as a vector too
#else
  return sym_packed_col(n, false) && !fused_order(n, false);
#endif
}
constexpr int tri_count(int n) { return n * (n + 1) / 2; }
// packed lower triangle, column by column: index of (i, j), i >= j, and back
constexpr int tri_index(int n, int i, int j) { return i + j * n - j * (j + 1) / 2; }
constexpr int tri_col(int n, int e) {
  int j = 0;
  while (e >= n - j) { e -= n - j; ++j; }
  return j;
}
constexpr int tri_row(int n, int e) {
  int j = 0;
  while (e >= n - j) { e -= n - j; ++j; }
  return j + e;
}

template <int NF, int DF>
struct SwLayRec {
  static constexpr bool PACKED = sym_packed(NF, false);
  static constexpr int NRT = PACKED ? tri_count(NF) : NF * NF;
  static constexpr int R = 0;
  static constexpr int T = R + rec_even(NRT);
  static constexpr int SUP = T + rec_even(NRT);
  static constexpr int SDN = SUP + rec_even(NF * DF);
  static constexpr int EDIR = SDN + rec_even(NF * DF);
  static constexpr int IDIR = EDIR + rec_even(DF * DF);
  static constexpr int NELEM = IDIR + rec_even(DF * DF);
};
template <int NF>
struct LwLayRec {
  static constexpr bool PACKED = sym_packed(NF, true);
  static constexpr int NRT = PACKED ? tri_count(NF) : NF * NF;
  static constexpr int R = 0;
  static constexpr int T = R + rec_even(NRT);
  static constexpr int IFLUX = T + rec_even(NRT);
  static constexpr int SRC = IFLUX + rec_even(NF * NF);
  static constexpr int ISRC = SRC + rec_even(NF);
  static constexpr int NELEM = ISRC + rec_even(NF);
};

// 16-byte element pair of the scratch
#ifndef SS_RECPAIR_DEFINED
#define SS_RECPAIR_DEFINED
struct alignas(16) RecPair { double x, y; };
#endif

// Lanes per scratch tile (= wavefront size) and the element stride inside a tile
constexpr long kRecStride = 64;

// Offset of the record of column-unit cu in layer slot jlay (records of nelem
// elements, maxlay layer slots per tile); element e (even) lives at + e * kRecStride
SS_HD long rec_base(int jlay, int maxlay, int nelem, int cu) {
  const long tile = cu >> 6, lane = cu & 63;
  return ((tile * maxlay + jlay) * nelem) * kRecStride + 2 * lane;
}

// Records are written once by one kernel and read once by a later one, GBs
// later: they move with the non-temporal hint, so that they do not displace what
// L2 is useful for here - per-layer inputs shared by neighbouring blocks and flux
// outputs still being merged into full lines (C3: 31.0 -> 32.2e6 columns/s;
// -DSS_NO_NT switches the hint off).
#if defined(__HIP_DEVICE_COMPILE__)
typedef double ss_double2 __attribute__((ext_vector_type(2)));
#endif
SS_HD RecPair rec_pair_load(const double* p) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SS_NO_NT)
  const ss_double2 t = __builtin_nontemporal_load(reinterpret_cast<const ss_double2*>(p));
  RecPair v;
  v.x = t.x;
  v.y = t.y;
  return v;
#else
  return *reinterpret_cast<const RecPair*>(p);
#endif
}
SS_HD void rec_pair_store(double* p, double v0, double v1) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SS_NO_NT)
  ss_double2 t;
  t.x = v0;
  t.y = v1;
  __builtin_nontemporal_store(t, reinterpret_cast<ss_double2*>(p));
#else
  RecPair v;
  v.x = v0;
  v.y = v1;
  *reinterpret_cast<RecPair*>(p) = v;
#endif
}

// NEL consecutive elements starting at the EVEN element "off" of a record
template <int NEL>
SS_HD void rec_load(const double* rec, long stride, int off, double* dst) {
  SS_UNROLL_FULL
  for (int k = 0; k < NEL; k += 2) {
    const RecPair v = rec_pair_load(rec + (long)(off + k) * stride);
    dst[k] = v.x;
    if (k + 1 < NEL) dst[k + 1] = v.y;
  }
}
template <int NEL>
SS_HD void rec_store(double* rec, long stride, int off, const double* src) {
  SS_UNROLL_FULL
  for (int k = 0; k < NEL; k += 2) {
    rec_pair_store(rec + (long)(off + k) * stride, src[k], (k + 1 < NEL) ? src[k + 1] : 0.0);
  }
}

// The records of ONE column unit in a scratch family (layer record, column record, input pack), addressed by
// layer and element: on the device a buffer resource whose base is the wave's tile - uniform, four scalar
// registers - plus ONE 32-bit byte offset per lane; the element offset of an access is a scalar operand.  With
// flat 64-bit addresses a sweep that touches ~25 windows of 4 KB per layer keeps as many 64-bit pointers per lane
// alive (and advances each of them every layer with a 64-bit add): ~50 VGPRs and ~200 VALU instructions per layer
// of pure address arithmetic in the downward kernel.  Loads carry the non-temporal hint like rec_pair_load.
struct RecView {
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t r;
  int voff0;        // byte offset of element 0 of layer 0 of this lane's unit from the resource base
  int lay_bytes;    // bytes per layer slot of a tile
  SS_HD RecView(const double* scratch, int maxlay, int nelem, int cu_abs) {
    const int tile = cu_abs >> 6;
    const int tile0 = __builtin_amdgcn_readfirstlane(tile);     // lanes ascend in cu: the smallest tile of the wave
    lay_bytes = nelem * (int)kRecStride * 8;
    const long tile_bytes = (long)maxlay * lay_bytes;
    const char* base = reinterpret_cast<const char*>(scratch) + (long)tile0 * tile_bytes;
    voff0 = (tile - tile0) * (int)tile_bytes + 16 * (cu_abs & 63);
    const long range = 2 * tile_bytes;      // a wave spans at most two tiles (unaligned first unit of a chunk)
    r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, range > 0x7fffffffL ? 0x7fffffff : (int)range, 0x00020000);
  }
  SS_HD RecPair pair(int jlay, int e) const {
    typedef int ss_int4 __attribute__((ext_vector_type(4)));
    const ss_int4 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff0 + jlay * lay_bytes, e * ((int)kRecStride * 8), 2);
    const ss_double2 d = __builtin_bit_cast(ss_double2, t);
    RecPair v;
    v.x = d.x;
    v.y = d.y;
    return v;
  }
#else
  const double* base;
  long lay_stride;
  SS_HD RecView(const double* scratch, int maxlay, int nelem, int cu_abs)
      : base(scratch + rec_base(0, maxlay, nelem, cu_abs)), lay_stride((long)nelem * kRecStride) {}
  SS_HD RecPair pair(int jlay, int e) const { return rec_pair_load(base + jlay * lay_stride + (long)e * kRecStride); }
#endif
  // NEL consecutive elements of layer jlay starting at the EVEN element "off"
  template <int NEL>
  SS_HD void load(int jlay, int off, double* dst) const {
    SS_UNROLL_FULL
    for (int k = 0; k < NEL; k += 2) {
      const RecPair v = pair(jlay, off + k);
      dst[k] = v.x;
      if (k + 1 < NEL) dst[k + 1] = v.y;
    }
  }
};

// Writer of one scratch record (rec = scratch + rec_base(...))
struct RecWriter {
  double* rec;
  long stride;
  unsigned long long* stamps;   // profiling builds (SS_STAMPS): section counters, else unused
  // elements e (even) and e+1
  SS_HD void put2(int e, double v0, double v1) const {
    rec_pair_store(rec + (long)e * stride, v0, v1);
  }
  // per-lane LDS scratch for the unit that writes through this writer (ss_layer_sym.hpp)
  SS_HD LaneLdsN<36> scratch() const { return LaneLdsN<36>{}; }
};
// Writer into a plain array (the cold general route of the small orders: its record is replayed through
// the real writer afterwards, see calc_matrices_*_route)
struct ArrayRecWriter {
  double* rec;
  unsigned long long* stamps;
  SS_HD void put2(int e, double v0, double v1) const { rec[e] = v0; rec[e + 1] = v1; }
  SS_HD LaneLdsN<36> scratch() const { return LaneLdsN<36>{}; }
};

// ---- fused layer + upward-sweep kernels (orders that live in registers) ----
// The layer kernel used to hand its matrices to the upward sweep through an HBM record of
// 128 (SW) / 120 (LW) doubles per (column, band, layer): written once, read once, 2 x 20 KB per
// column in the headline configuration - as much traffic as everything else together.  For
// orders up to SS_UNROLL_MAX one lane now does the layer matrices of a layer AND the adding
// step that consumes them before it goes on to the next layer (k_layer_up): R and T wait in the
// lane's LDS slots (2 n^2 <= 72 of kLaneLdsSlots = 80 doubles), the direct-beam / source
// blocks in registers, and what only the downward sweep reads (E, int_dir; int_flux, int_source)
// goes straight into the column record.  The loop-carried albedo matrices are not kept across
// the layer stage (they would be spilled): the adding step re-reads a_below / d_below of the
// layer below, which it wrote to the column record a layer earlier (L2), and applies the
// overlap matrices first.  -DSS_NO_FUSE builds the three-kernel form for every order.
// Measured on C3, 1e6 columns, one stream (ms per step; tools/run_variants.sh):
//                      layer+up SW   layer+up LW   4 streams, columns/s
//   three kernels      6.3 + 4.1     6.3 + 3.2     42.4e6
//   fused, both        11.5 (10.8 with S_up/S_dn through the record, -DSS_FUSE_VIA_RECORD)
//                                    9.4           40.1e6
//   fused, LW only     6.3 + 4.1     9.4           42.7e6     <- the default
// The fused kernel is the sum of its parts, not less: with one wave per SIMD the adding step
// was bound by exposed latencies, not by the bytes of the layer record, and the shortwave form
// spills three times as much as its two halves (831 against 270 + 94 dwords).  The longwave
// form costs nothing and takes 20 KB per column off the HBM traffic, so it is on; the
// shortwave form stays available (-DSS_FUSE_SW) and correct (all parity tests pass with it).
constexpr bool fused_order(int n, bool lw) {
  // Round 3: with the layer matrices in symmetrised coordinates (ss_layer_sym.hpp) the layer kernel of the
  // small orders no longer spills, and the three-kernel form is the faster one for the longwave too
  // (measured, C3, ms per step on one stream: fused 7.06; layer 3.29 + upward 3.11).  The fused kernels
  // remain for builds with -DSS_FUSE_LW / -DSS_FUSE_SW, which imply the full (unpacked) records.
#if defined(SS_FUSE_SW)
  return n <= SS_UNROLL_MAX || (false && lw);
#elif defined(SS_FUSE_LW)
  return lw && n <= SS_UNROLL_MAX && n < SS_COOP_MIN_ORDER;
#elif defined(SS_SYM_FUSE_LW) || defined(SS_SYM_FUSE_LW2) || defined(SS_SYM_FUSE_SW) || defined(SS_SYM_FUSE_SW2)
  // layer + upward sweep on one lane, both in symmetrised coordinates; no layer record exists.
  //  SS_SYM_FUSE_LW  (ss_column_lw_sym.hpp: sym::layer_up_lw): R^, T^ and the layer's source wait in LDS slots;
  //  SS_SYM_FUSE_SW  (ss_column_sw_sym.hpp: sym::layer_up_sw): R^, T^, S_up^, S_dn^ wait in the 80 LDS slots of the lane;
  //  SS_SYM_FUSE_SW2 (ss_layer_up_sw2.hpp): the loop-carried albedo of everything below waits in LDS slots too and
  //                  the direct beam is a vector throughout.
  // The spectral regions are switched independently.
  bool f = false;
#if defined(SS_SYM_FUSE_LW) || defined(SS_SYM_FUSE_LW2)
  if (lw && n >= 3 && n <= SS_UNROLL_MAX) f = true;
#endif
#if defined(SS_SYM_FUSE_SW) || defined(SS_SYM_FUSE_SW2)
  if (!lw && n >= 3 && n <= SS_UNROLL_MAX) f = true;
#endif
  return f;
#else
  return false && n > 0 && lw;
#endif
}

// Writer of the fused layer stage: elements below KEEP0 (R, T) go to the lane's LDS slots,
// [KEEP0, KEEP1) stay in registers (hold), [FWD0, ...) are stored to the column record at
// element FWD_AT + (e - FWD0); an element range may be both kept and forwarded.
template <int KEEP0, int KEEP1, int FWD0, int FWD_AT, class Lds>
struct FusedWriter {
  const Lds* lds;
  double* hold;           // KEEP1 - KEEP0 elements
  double* crec;           // column record of this (unit, layer)
  long stride;
  unsigned long long* stamps;
  SS_HD void put2(int e, double v0, double v1) const {
    if (e < KEEP0) {
      lds->put(e, v0);
      lds->put(e + 1, v1);
    } else {
      if (e < KEEP1) {
        hold[e - KEEP0] = v0;
        hold[e + 1 - KEEP0] = v1;
      }
      if (e >= FWD0) rec_pair_store(crec + (long)(FWD_AT + e - FWD0) * stride, v0, v1);
    }
  }
  SS_HD const Lds& scratch() const { return *lds; }   // the slots are free until R and T are put
};

// Sources of the adding step: the HBM layer record (three-kernel form) ...
struct RecSource {
  const double* rec;
  long stride;
  template <int NEL>
  SS_HD void load(int off, double* dst) const { rec_load<NEL>(rec, stride, off, dst); }
};
// ... or what the fused layer stage left in LDS / registers
template <int KEEP0, class Lds>
struct FusedSource {
  const Lds* lds;
  const double* hold;
  template <int NEL>
  SS_HD void load(int off, double* dst) const {
    SS_UNROLL_FULL
    for (int k = 0; k < NEL; ++k) dst[k] = (off < KEEP0) ? lds->get(off + k) : hold[off + k - KEEP0];
  }
};

// ... or LDS for the elements below KEEP0 and a record (element e at AT + e - KEEP0) for the rest
template <int KEEP0, int AT, class Lds>
struct LdsRecSource {
  const Lds* lds;
  const double* rec;
  long stride;
  template <int NEL>
  SS_HD void load(int off, double* dst) const {
    if (off < KEEP0) {
      SS_UNROLL_FULL
      for (int k = 0; k < NEL; ++k) dst[k] = lds->get(off + k);
    } else {
      rec_load<NEL>(rec, stride, AT + off - KEEP0, dst);
    }
  }
};

// Write an (RS x CS) block held in registers into the (RF x CF) matrix that
// starts at the even element "off", at block position (I0, J0); everything else zero.
template <int RF, int CF, int RS, int CS, int I0, int J0, class W>
SS_HD void put_embedded(const W& w, int off, const double* blk) {
  auto val = [&](int idx) -> double {
    const int i = idx % RF, j = idx / RF;
    const bool inside = (idx < RF * CF) && (i >= I0) && (i < I0 + RS) && (j >= J0) && (j < J0 + CS);
    return inside ? SS_E(blk, i - I0, j - J0, RS) : 0.0;
  };
  SS_UNROLL_FULL
  for (int e = 0; e < RF * CF; e += 2) w.put2(off + e, val(e), val(e + 1));
}
// Same for a vector of NF elements holding an N-vector at I0
template <int NF, int N, int I0, class W, typename F>
SS_HD void put_embedded_vec(const W& w, int off, F&& elem) {
  auto val = [&](int i) -> double {
    const bool inside = (i < NF) && (i >= I0) && (i < I0 + N);
    return inside ? elem(i - I0) : 0.0;
  };
  SS_UNROLL_FULL
  for (int e = 0; e < NF; e += 2) w.put2(off + e, val(e), val(e + 1));
}

// Packed symmetric block of an N sub-problem at offset I0 into the packed NF record block at "off":
// elem(ic, jc) with integral constants i >= j gives the sub-problem's element; everything else zero.
template <int NF, int N, int I0, int E, typename F>
SS_HD double packed_value(F&& elem) {
  if constexpr (E >= tri_count(NF)) {
    return 0.0;
  } else {
    constexpr int i = tri_row(NF, E), j = tri_col(NF, E);
    if constexpr ((i >= I0) && (i < I0 + N) && (j >= I0) && (j < I0 + N))
      return elem(std::integral_constant<int, i - I0>{}, std::integral_constant<int, j - I0>{});
    else
      return 0.0;
  }
}
template <int NF, int N, int I0, class W, typename F>
SS_HD void put_packed(const W& w, int off, F&& elem) {
  static_loop<0, (tri_count(NF) + 1) / 2>([&](auto pc) {
    constexpr int E = 2 * decltype(pc)::value;
    w.put2(off + E, packed_value<NF, N, I0, E>(elem), packed_value<NF, N, I0, E + 1>(elem));
  });
}
// Full N x N matrix from a packed scaled block of a record source: out(i, j) = dsc_i X^(i, j) / dsc_j
template <int N, class Src>
SS_HD void load_packed_unscaled(const Src& src, int off, const double* dsc, const double* rd, double* out) {
  double t[rec_even(tri_count(N))];
  src.template load<rec_even(tri_count(N))>(off, t);
  static_loop<0, tri_count(N)>([&](auto ec) {
    constexpr int E = decltype(ec)::value;
    constexpr int i = tri_row(N, E), j = tri_col(N, E);
    const double v = t[E];
    SS_E(out, i, j, N) = (dsc[i] * rd[j]) * v;
    if constexpr (i != j) SS_E(out, j, i, N) = (dsc[j] * rd[i]) * v;
  });
}
// The lower triangle of D^-1 X D into a packed block of a record (X full, N x N)
template <int N>
SS_HD void store_packed_scaled(double* rec, long stride, int off, const double* X, const double* dsc, const double* rd) {
  static_loop<0, (tri_count(N) + 1) / 2>([&](auto pc) {
    constexpr int E = 2 * decltype(pc)::value;
    constexpr int i0 = tri_row(N, E), j0 = tri_col(N, E);
    double v1 = 0.0;
    if constexpr (E + 1 < tri_count(N)) {
      constexpr int i1 = tri_row(N, E + 1), j1 = tri_col(N, E + 1);
      v1 = SS_E(X, i1, j1, N) * (dsc[j1] * rd[i1]);
    }
    rec_pair_store(rec + (long)(off + E) * stride, SS_E(X, i0, j0, N) * (dsc[j0] * rd[i0]), v1);
  });
}
// y = D X^ D^-1 x for a packed symmetric X^ (the product with the unscaled matrix without restoring it)
template <int N>
SS_HD void packed_scaled_matvec(const double* Xp, const double* dsc, const double* rd, const double* x, double* y) {
  double t[N];
  SS_UNROLL_FULL
  for (int k = 0; k < N; ++k) t[k] = rd[k] * x[k];
  static_loop<0, N>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    double acc = 0.0;
    static_loop<0, N>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      acc += Xp[tri_index(N, i >= k ? i : k, i >= k ? k : i)] * t[k];
    });
    y[i] = dsc[i] * acc;
  });
}
// scaling of the components of a layer: dsc = sqrt(region fraction) x sqrt(weight mu), rd = 1 / dsc
// (0 for a region without area: its components carry no flux)
template <int NREG, int NS>
SS_HD void layer_scaling(const LG& lg, const double* frac, double* dsc, double* rd) {
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) {
    const double sf = sqrt(frac[r]);
    const double rsf = frac[r] > 0.0 ? 1.0 / sf : 0.0;
    SS_UNROLL_FULL
    for (int s = 0; s < NS; ++s) {
      dsc[s + r * NS] = sf * lg.sqrt_wmu[s];
      rd[s + r * NS] = rsf * lg.rsqrt_wmu[s];
    }
  }
}

template <int N>
SS_HD void scale_cols(const double* A, const double* v, double* B) {
  SS_UNROLL_N(N)
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) SS_E(B, i, j, N) = SS_E(A, i, j, N) * v[j];
  }
}

// B = diag(v) A  (A is R x C)
template <int R, int C>
SS_HD void scale_rows(const double* A, const double* v, double* B) {
  SS_UNROLL_FULL
  for (int j = 0; j < C; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < R; ++i) SS_E(B, i, j, R) = v[i] * SS_E(A, i, j, R);
  }
}

// Diffuse part shared by SW and LW (sw_eig :180-221, lw_eig :142-176).
// On return: G1, G2 = eigenvector blocks, lam, e = exp(-lam dz),
// LUG1 = lu_factor(G1), Clo = C'_lower, Cup = C'_upper; R and T are written.
template <int N>
struct Diffuse {
  double G1[N * N], G2[N * N], LUG1[N * N], Clo[N * N], Cup[N * N];
  double lam[N], e[N];
};

// Optionally (ND > 0) also the particular solutions of the direct-beam sources
// (sw_eig :231-253): for every direct eigenvalue mu and source s = srhs(:, jd)
//   (G1 + mu) g3 + G2 g4 = -s,   G2 g3 + (G1 - mu) g4 = -s.
// With p = g3 + g4, q = g3 - g4 the sum and the difference of the two read
//   (P - mu^2) p = -2 (G1 - G2) s,   q = -mu (G1 - G2)^-1 p,   P = (G1 - G2)(G1 + G2),
// so the eigen-decomposition P = X Lambda X^-1 that is needed anyway gives
//   p = X c,  q = -mu ((G1 - G2)^-1 X) c,  c = -2 (Lambda - mu^2)^-1 X^-1 (G1 - G2) s
// for three matrix-vector products per mu, where the reference factorises two
// N x N matrices, inverts one and multiplies two per mu.  X^-1 (G1 - G2) s is
// -(D^-2 X)^T s on the symmetric route and an LU solve with X on the general one.
// Eigen record of the unit for the orders of the cooperative eigen-solver (ss_eigen_coop.hpp):
// nullptr = solve here.
struct CoopEig {
  const double* rec;
  long stride;
};
template <int N>
SS_HD int coop_eig_post(const double* A, const double* dscale, const double* rec, long stride,
                        double* eigenvalue, double* eigenvector, double* ainv_x);

template <int N, int ND = 0, class PutR, class PutT>
SS_HD int diffuse_part(const double* g1, const double* g2, const double* dscale, double dz, Diffuse<N>& w,
                       const PutR& putR, const PutT& putT, unsigned long long* stamps = nullptr,
                       int stamp0 = 0, const double* srhs = nullptr, const double* mu = nullptr,
                       double* g3v = nullptr, double* g4v = nullptr, const CoopEig* coop = nullptr) {
  SS_STAMP_DECL(stamps);
  double GD[N * N], P[N * N], X[N * N], ev[N];
  double hc[N * (ND > 0 ? ND : 1)];   // X^-1 (G1 - G2) s, one column per direct eigenvalue
  int ierr = -1;
  SS_UNROLL_NN(N)
  for (int k = 0; k < N * N; ++k) GD[k] = g1[k] - g2[k];
  SS_STAMP(stamp0);
  // eigen-decomposition of (G1 - G2)(G1 + G2): through its hidden symmetry where
  // the register-resident solver exists (ss_eigen_sym.hpp), else - or when the
  // structure is not there numerically - with the general solver of the reference
  if constexpr (N >= 3 && N <= SS_UNROLL_MAX) {
    if (dscale) ierr = eigen_decomposition_gamma_sym<N>(GD, g2, dscale, ev, X, P);
    if (ierr >= 0 && !sym_spectrum_is_well_graded<N>(ev)) ierr = -1;
    if (ND > 0 && ierr >= 0) {
      // X^-1 (G1 - G2) s = -(D^-2 X)^T s
      double rd2[N];
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) rd2[i] = 1.0 / (dscale[i] * dscale[i]);
      SS_UNROLL_N(ND)
      for (int jd = 0; jd < ND; ++jd) {
        SS_UNROLL_N(N)
        for (int k = 0; k < N; ++k) {
          double h = 0.0;
          SS_UNROLL_N(N)
          for (int i = 0; i < N; ++i) h -= SS_E(X, i, k, N) * (srhs[i + jd * N] * rd2[i]);
          hc[k + jd * N] = h;
        }
      }
    }
  }
  if constexpr (N > SS_UNROLL_MAX) {
    // eigenvalues and vectors solved by the cooperative kernel between the two passes of the layer kernel
    if (coop && dscale) ierr = coop_eig_post<N>(GD, dscale, coop->rec, coop->stride, ev, X, P);
    if (ND > 0 && ierr >= 0) {
      // X^-1 (G1 - G2) s = -(D^-2 X)^T s, as on the symmetric route of the small orders
      for (int jd = 0; jd < ND; ++jd)
        for (int k = 0; k < N; ++k) {
          double h = 0.0;
          for (int i = 0; i < N; ++i) h -= SS_E(X, i, k, N) * (srhs[i + jd * N] / (dscale[i] * dscale[i]));
          hc[k + jd * N] = h;
        }
    }
  }
#if defined(SS_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
  // profiling builds: slots 30 / 31 count the units that took the symmetric / the general solver
  if (stamps && N >= 3 && N <= SS_UNROLL_MAX) atomicAdd(stamps + 1 + (ierr < 0 ? 30 : 29), 1ULL);
#endif
  if (ierr < 0) {
    double GS[N * N];
    SS_UNROLL_NN(N)
    for (int k = 0; k < N * N; ++k) GS[k] = g1[k] + g2[k];
    mat_mul<N, N, N>(GD, GS, P);
    ierr = eigen_decomposition_real<N>(P, ev, X);
    if (ND > 0) {
      // X^-1 ((G1 - G2) s) by an LU solve with X (GS is free: reuse it for LU(X))
      mat_mul<N, N, (ND > 0 ? ND : 1)>(GD, srhs, hc);
      mat_copy<N, N>(X, GS);
      lu_factor<N>(GS);
      lu_solve_mat<N, (ND > 0 ? ND : 1)>(GS, hc);
    }
    // P = GD^-1 X (the symmetric route returns it from its Cholesky factor)
    lu_factor<N>(GD);
    mat_copy<N, N>(X, P);
    lu_solve_mat<N, N>(GD, P);
  }
  if (ND > 0) {
    SS_UNROLL_N(ND)
    for (int jd = 0; jd < ND; ++jd) {
      double c[N], pv[N], qv[N];
      SS_UNROLL_N(N)
      for (int k = 0; k < N; ++k) c[k] = -2.0 * hc[k + jd * N] / (ev[k] - mu[jd] * mu[jd]);
      mat_vec<N, N>(X, c, pv);
      mat_vec<N, N>(P, c, qv);
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) {
        const double q = -mu[jd] * qv[i];
        g3v[i + jd * N] = 0.5 * (pv[i] + q);
        g4v[i + jd * N] = 0.5 * (pv[i] - q);
      }
    }
  }
  SS_STAMP(stamp0 + 1);
  SS_UNROLL_N(N)
  for (int j = 0; j < N; ++j) {
    w.lam[j] = sqrt(ss_max(0.0, ev[j]));
    w.e[j] = exp(-w.lam[j] * dz);
  }
  // tmp = -(GD^-1 X) diag(lam);  G1 = X + tmp, G2 = X - tmp   (P = GD^-1 X)
  SS_UNROLL_N(N)
  for (int j = 0; j < N; ++j) {
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) {
      const double t = -SS_E(P, i, j, N) * w.lam[j];
      SS_E(w.G1, i, j, N) = SS_E(X, i, j, N) + t;
      SS_E(w.G2, i, j, N) = SS_E(X, i, j, N) - t;
    }
  }
  // Eq. 48 by Schur complement.  G1D = G1 diag(e) and G2D = G2 diag(e) are never
  // formed: the scaling is applied to the rows of the right-hand factor.
  mat_copy<N, N>(w.G1, w.LUG1);
  lu_factor<N>(w.LUG1);
  {
    double t1[N * N];
    mat_copy<N, N>(w.G2, t1);
    lu_solve_mat<N, N>(w.LUG1, t1);             // G1^-1 G2
    SS_UNROLL_N(N)
    for (int j = 0; j < N; ++j) {
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) SS_E(t1, i, j, N) *= w.e[i] * w.e[j];   // diag(e) G1^-1 G2 diag(e)
    }
    mat_mul<N, N, N>(w.G2, t1, P);               // G2D G1^-1 G2D
    SS_UNROLL_NN(N)
    for (int k = 0; k < N * N; ++k) P[k] = w.G1[k] - P[k];
    lu_factor<N>(P);
    lu_invert<N>(P, w.Clo);                      // C'_lower = S^-1
    mat_mul_rs<N, N, N, false>(w.G2, w.e, w.Clo, w.Cup);    // G2D Clo
    lu_solve_mat<N, N>(w.LUG1, w.Cup);
    SS_UNROLL_NN(N)
    for (int k = 0; k < N * N; ++k) w.Cup[k] = -w.Cup[k];   // C'_upper
    // Eq. 49: R = G1D Cup + G2 Clo ; T = G2 Cup + G1D Clo
    mat_mul<N, N, N>(w.G2, w.Clo, t1);
    mat_mul_rs<N, N, N, true>(w.G1, w.e, w.Cup, t1);
    putR(t1);
    mat_mul<N, N, N>(w.G2, w.Cup, t1);
    mat_mul_rs<N, N, N, true>(w.G1, w.e, w.Clo, t1);
    putT(t1);
  }
  SS_STAMP(stamp0 + 2);
  return ierr;
}

// The direct-beam part of calc_matrices_sw_eig for a full problem: int_dir = -Gamma0^-1
// (radtool_schur.F90:45-51) and E = G0 diag(exp(mu dz)) G0^-1 (sw_eig :223-229) into the layer
// record; returns the eigenvalues mu and eigenvectors G0 of Gamma0, G0^-1 and exp(mu dz).
template <int NF, int D, class W>
SS_HD int sw_direct_part(const double* g0, const double* dsc0, double dz, const W& w, double* mu, double* G0,
                         double* G0inv, double* emu) {
  using Rec = SwRec<NF, D>;
  {
    double LU0[D * D], g0i[D * D], idir[D * D];
    mat_copy<D, D>(g0, LU0);
    lu_factor<D>(LU0);
    lu_invert<D>(LU0, g0i);
    SS_UNROLL_FULL
    for (int k = 0; k < D * D; ++k) idir[k] = -g0i[k];
    put_embedded<D, D, D, D, 0, 0>(w, Rec::IDIR, idir);
  }
  int rc = -1;
  if constexpr (D >= 3 && D <= SS_UNROLL_MAX) {
    if (dsc0) rc = eigen_decomposition_scaled_sym<D>(g0, dsc0, mu, G0);
    if (rc >= 0 && !sym_spectrum_is_well_graded<D>(mu)) rc = -1;
  }
  if (rc < 0) rc = eigen_decomposition_real<D>(g0, mu, G0);
  double LU0[D * D], t[D * D], Ed[D * D];
  mat_copy<D, D>(G0, LU0);
  lu_factor<D>(LU0);
  lu_invert<D>(LU0, G0inv);
  SS_UNROLL_N(D)
  for (int j = 0; j < D; ++j) emu[j] = exp(mu[j] * dz);
  scale_cols<D>(G0, emu, t);
  mat_mul<D, D, D>(t, G0inv, Ed);
  put_embedded<D, D, D, D, 0, 0>(w, Rec::EDIR, Ed);
  return rc;
}

// The direct-beam part of a (D) sub-problem embedded at region offset R0 of the (DF x DF) blocks of the
// record: int_dir = -Gamma0^-1 (schur_invert_sw, radtool_schur.F90:45-51; int_diff = -(G1 + G2)^-1 and
// int_dir_diff = -2 int_diff (g3 g0i) of the same routine are not formed: the downward sweep applies both
// to its flux convergences with one solve with G1 + G2) and E = G0 diag(exp(mu dz)) G0^-1 (sw_eig :223-229).
// Returns the eigen-solver's code; mu, G0, G0^-1 and exp(mu dz) for the particular solutions.
template <int NF, int DF, int D, int R0, class Rec = SwRec<NF, DF>, class W>
SS_HD int sw_direct_embedded(const double* g0, const double* dsc0, double dz, const W& w, double* mu, double* G0,
                             double* G0inv, double* emu) {
  SS_STAMP_DECL(w.stamps);
  {
    double LU0[D * D], g0i[D * D], idir[D * D];
    mat_copy<D, D>(g0, LU0);
    lu_factor<D>(LU0);
    lu_invert<D>(LU0, g0i);
    SS_UNROLL_FULL
    for (int k = 0; k < D * D; ++k) idir[k] = -g0i[k];
    put_embedded<DF, DF, D, D, R0, R0>(w, Rec::IDIR, idir);
  }
  SS_STAMP(1);
  // Gamma0 is symmetric after scaling with sqrt(region fraction)
  int rc = -1;
  if constexpr (D >= 3 && D <= SS_UNROLL_MAX) {
    if (dsc0) rc = eigen_decomposition_scaled_sym<D>(g0, dsc0, mu, G0);
    if (rc >= 0 && !sym_spectrum_is_well_graded<D>(mu)) rc = -1;
  }
  if (rc < 0) rc = eigen_decomposition_real<D>(g0, mu, G0);
  {
    double LU0[D * D], t[D * D], Ed[D * D];
    mat_copy<D, D>(G0, LU0);
    lu_factor<D>(LU0);
    lu_invert<D>(LU0, G0inv);
    SS_UNROLL_N(D)
    for (int j = 0; j < D; ++j) emu[j] = exp(mu[j] * dz);
    scale_cols<D>(G0, emu, t);
    mat_mul<D, D, D>(t, G0inv, Ed);
    put_embedded<DF, DF, D, D, R0, R0>(w, Rec::EDIR, Ed);
  }
  return rc;
}

// calc_matrices_sw_eig for an (N, D) sub-problem embedded at (I0 = stream
// offset, R0 = region offset) of the full (NF, DF) record.
template <int NF, int DF, int N, int D, int I0, int R0, class W>
SS_HD int calc_matrices_sw(const double* g0, const double* g1, const double* g2,
                           const double* g3, const double* dsc0, const double* dsc, double dz,
                           const W& w, const CoopEig* coop = nullptr) {
  using Rec = SwRec<NF, DF>;
  SS_STAMP_DECL(w.stamps);
  int ierr = 0;
  double mu[D], G0[D * D], G0inv[D * D], emu[D];
  double g3v[N * D], g4v[N * D];   // particular solutions, one column per direct eigenvalue

  ierr += sw_direct_embedded<NF, DF, D, R0>(g0, dsc0, dz, w, mu, G0, G0inv, emu);

  // Section 4 (:231-253): the sources of the particular solutions g3, g4, one column
  // per direct eigenvalue; solved inside diffuse_part from the eigen-decomposition
  double g3g0[N * D];
  mat_mul<N, D, D>(g3, G0, g3g0);

  SS_STAMP(2);
  // Sections 1-2: diffuse R, T
  Diffuse<N> df;
  ierr += diffuse_part<N, D>(
      g1, g2, dsc, dz, df,
      [&](const double* R) { put_embedded<NF, NF, N, N, I0, I0>(w, Rec::R, R); },
      [&](const double* T) { put_embedded<NF, NF, N, N, I0, I0>(w, Rec::T, T); }, w.stamps, 3,
      g3g0, mu, g3v, g4v, coop);
  SS_STAMP(20);   // (re-arm)

  // direct_diffuse_part (:303-386) by block elimination (see header comment)
  {
    double b1[N * D], b2[N * D], c1[N * D], c2[N * D], t[N * D];
    {
      double g3e[N * D];
      SS_UNROLL_N(D)
      for (int j = 0; j < D; ++j) {
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) SS_E(g3e, i, j, N) = SS_E(g3v, i, j, N) * emu[j];
      }
      mat_mul<N, D, D>(g3e, G0inv, b1);    // = -(rhs of block 1)
      mat_mul<N, D, D>(g4v, G0inv, b2);    // = -(rhs of block 2)
    }
    // c2 = S^-1 (-b2 + G2D G1^-1 b1)   (G2D x = G2 (diag(e) x))
    double te[N * D];
    mat_copy<N, D>(b1, t);
    lu_solve_mat<N, D>(df.LUG1, t);
    scale_rows<N, D>(t, df.e, te);
    mat_mul<N, N, D>(df.G2, te, c1);
    SS_UNROLL_FULL
    for (int k = 0; k < N * D; ++k) c1[k] = c1[k] - b2[k];
    mat_mul<N, N, D>(df.Clo, c1, c2);
    // c1 = G1^-1 (-b1 - G2D c2)
    scale_rows<N, D>(c2, df.e, te);              // diag(e) c2, reused for S_dn
    mat_mul<N, N, D>(df.G2, te, c1);
    SS_UNROLL_FULL
    for (int k = 0; k < N * D; ++k) c1[k] = -b1[k] - c1[k];
    lu_solve_mat<N, D>(df.LUG1, c1);
    // S_up = G1D c1 + G2 c2 + g3 G0^-1 ; S_dn = G2 c1 + G1D c2 + (g4 e) G0^-1
    mat_mul<N, N, D>(df.G2, c1, t);
    mat_mul_acc<N, N, D>(df.G1, te, t);          // S_dn without the particular part
    scale_rows<N, D>(c1, df.e, te);              // diag(e) c1
    double su[N * D];
    mat_mul<N, N, D>(df.G1, te, su);
    mat_mul_acc<N, N, D>(df.G2, c2, su);
    mat_mul_acc<N, D, D>(g3v, G0inv, su);
    put_embedded<NF, DF, N, D, I0, R0>(w, Rec::SUP, su);
    {
      double g4e[N * D];
      SS_UNROLL_N(D)
      for (int j = 0; j < D; ++j) {
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) SS_E(g4e, i, j, N) = SS_E(g4v, i, j, N) * emu[j];
      }
      mat_mul_acc<N, D, D>(g4e, G0inv, t);
    }
    put_embedded<NF, DF, N, D, I0, R0>(w, Rec::SDN, t);
  }
  SS_STAMP(6);

  return ierr;
}

// calc_matrices_lw_eig for an N sub-problem embedded at stream offset I0.
template <int NF, int N, int I0, class W>
SS_HD int calc_matrices_lw(const double* g1, const double* g2, const double* b, const double* dsc,
                           double dz, const W& w, const CoopEig* coop = nullptr) {
  using Rec = LwRec<NF>;
  SS_STAMP_DECL(w.stamps);
  // Section 3 (:178-195): inv_gamma_b (first, while only the Gamma matrices are live)
  // (G1 - G2 G1^-1 G2) igb = (G2 G1^-1 - I) b  <=>  igb = -(G1 + G2)^-1 b  (same identity as
  // for int_diff of the shortwave: G1 - G2 G1^-1 G2 = (G1 - G2) G1^-1 (G1 + G2))
  double igb[N];
  {
    double B[N * N];
    SS_UNROLL_NN(N)
    for (int k = 0; k < N * N; ++k) B[k] = g1[k] + g2[k];
    lu_factor<N>(B);
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) igb[i] = -b[i];
    lu_solve<N>(B, igb);
  }
  SS_STAMP(12);
  Diffuse<N> df;
  int ierr = diffuse_part<N>(
      g1, g2, dsc, dz, df,
      [&](const double* R) { put_embedded<NF, NF, N, N, I0, I0>(w, Rec::R, R); },
      [&](const double* T) { put_embedded<NF, NF, N, N, I0, I0>(w, Rec::T, T); }, w.stamps, 13,
      nullptr, nullptr, nullptr, nullptr, coop);
  SS_STAMP(20);   // (re-arm)

  // Section 4 (:198-209): cb' and the source
  double cb[N], v[N];
  {
    double t[N];
    vec_copy<N>(igb, t);
    lu_solve<N>(df.LUG1, t);                    // g1^-1 inv_gamma_b
    double te[N];
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) te[i] = df.e[i] * t[i];
    mat_vec<N, N>(df.G2, te, v);                // G2D g1^-1 igb
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) v[i] = igb[i] - v[i];
    mat_vec<N, N>(df.Clo, v, cb);
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) cb[i] = -cb[i];
    // source = (G1D + G2) cb' + inv_gamma_b
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) te[i] = df.e[i] * cb[i];
    mat_vec<N, N>(df.G2, cb, v);
    mat_vec_acc<N, N>(df.G1, te, v);
    put_embedded_vec<NF, N, I0>(w, Rec::SRC, [&](int i) { return v[i] + igb[i]; });
  }
  // Section 5 (:211-227): integrated fluxes
  {
    double z[N], GZ[N * N], CS[N * N], IF[N * N];
    SS_UNROLL_N(N)
    for (int j = 0; j < N; ++j) z[j] = (1.0 - df.e[j]) / df.lam[j];
    SS_UNROLL_N(N)
    for (int j = 0; j < N; ++j) {
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i)
        SS_E(GZ, i, j, N) = SS_E(df.G1, i, j, N) * z[j] + SS_E(df.G2, i, j, N) * z[j];
    }
    SS_UNROLL_NN(N)
    for (int k = 0; k < N * N; ++k) CS[k] = df.Clo[k] + df.Cup[k];
    mat_mul<N, N, N>(GZ, CS, IF);
    put_embedded<NF, NF, N, N, I0, I0>(w, Rec::IFLUX, IF);
    mat_vec<N, N>(GZ, cb, v);
    put_embedded_vec<NF, N, I0>(w, Rec::ISRC, [&](int i) { return 2.0 * (v[i] + igb[i] * dz); });
  }
  SS_STAMP(16);
  return ierr;
}

}  // namespace ss
