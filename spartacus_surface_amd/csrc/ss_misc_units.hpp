// ss_misc_units.hpp - the cheap per-column paths of radsurf():
//   flat_unit       : flat tiles, closed form (radsurf_interface.F90:122-173)
//   zero_unit       : zero_canopy_flux of one column (radsurf_canopy_flux.F90:286-341),
//                     used for night-time columns (radsurf_interface.F90:193-196)
//   simple_urban_unit: single-layer Harman tiles, types 4 / 5
//                     (radsurf_simple_urban_sw.F90, radsurf_simple_urban_lw.F90,
//                     radsurf_view_factor.F90; dispatch radsurf_interface.F90:271-308)
//   simple_spectrum_lw units: calc_simple_spectrum_lw
//                     (radsurf_simple_spectrum.F90:41-66), the pre-step of the LW path
#pragma once

#include "ss_common.hpp"

namespace ss {

#ifndef SS_RECPAIR_DEFINED
#define SS_RECPAIR_DEFINED
// 16-byte element pair
struct alignas(16) RecPair { double x, y; };
#endif

struct MiscLaunch {
  Inputs in;
  const int32_t* cols;
  int ncols;
  int nsw, nlw;
  int do_sw, do_lw;
  double* sw_albedo;
  double* sw_albedo_dir;
  double* lw_emissivity;
  double* lw_emission;
  FluxDev sw_dir, sw_diff, lw_int, lw_norm;
  // simple-urban tiles only
  double min_building_fraction;
  int infinite_street;     // 1: type 5 (infinite street), 0: type 4 (exponential model)
};

// one flat column
SS_HD void flat_unit(const MiscLaunch& M, int i) {
  const int col = M.cols[i];
#define PUT(fd, F, idx, v) do { if ((fd).f[F]) (fd).f[F][idx] = (v); } while (0)
  if (M.do_sw) {
    for (int b = 0; b < M.nsw; ++b) {
      const long c = (long)b + (long)M.nsw * col;
      const double ga = M.in.ground_albedo[c], gad = M.in.ground_albedo_dir[c];
      M.sw_albedo[c] = ga;
      M.sw_albedo_dir[c] = gad;
      PUT(M.sw_dir, F_GROUND_DN_DIR, c, 1.0);
      PUT(M.sw_dir, F_GROUND_DN, c, 1.0);
      PUT(M.sw_dir, F_GROUND_NET, c, 1.0 - gad);
      PUT(M.sw_dir, F_GROUND_VERTICAL_DIFF, c, 0.5 * gad);
      PUT(M.sw_dir, F_TOP_DN_DIR, c, 1.0);
      PUT(M.sw_dir, F_TOP_DN, c, 1.0);
      PUT(M.sw_dir, F_TOP_NET, c, 1.0 - gad);
      PUT(M.sw_diff, F_GROUND_DN_DIR, c, 0.0);
      PUT(M.sw_diff, F_GROUND_DN, c, 1.0);
      PUT(M.sw_diff, F_GROUND_NET, c, 1.0 - ga);
      PUT(M.sw_diff, F_GROUND_VERTICAL_DIFF, c, 0.5 * (1.0 + ga));
      PUT(M.sw_diff, F_TOP_DN_DIR, c, 0.0);
      PUT(M.sw_diff, F_TOP_DN, c, 1.0);
      PUT(M.sw_diff, F_TOP_NET, c, 1.0 - ga);
    }
  }
  if (M.do_lw) {
    for (int b = 0; b < M.nlw; ++b) {
      const long c = (long)b + (long)M.nlw * col;
      const double em = M.in.ground_emissivity[c], emission = M.in.ground_emission[c];
      M.lw_emissivity[c] = em;
      M.lw_emission[c] = emission;
      PUT(M.lw_int, F_GROUND_DN, c, 0.0);
      PUT(M.lw_int, F_GROUND_NET, c, -emission);
      PUT(M.lw_int, F_GROUND_VERTICAL_DIFF, c, 0.5 * emission);
      PUT(M.lw_int, F_TOP_DN, c, 0.0);
      PUT(M.lw_int, F_TOP_NET, c, -emission);
      PUT(M.lw_norm, F_GROUND_DN, c, 1.0);
      PUT(M.lw_norm, F_GROUND_NET, c, em);
      PUT(M.lw_norm, F_GROUND_VERTICAL_DIFF, c, 0.5 * (2.0 - em));
      PUT(M.lw_norm, F_TOP_DN, c, 1.0);
      PUT(M.lw_norm, F_TOP_NET, c, em);
    }
  }
#undef PUT
}

// Shape class of every canopy_flux member: 0 (nspec,ncol), 1 (nspec,ntotlay),
// 2 (ncol), 3 (ntotlay)
SS_HD int flux_field_shape(int F) {
  switch (F) {
    case F_GROUND_DN: case F_GROUND_DN_DIR: case F_GROUND_NET: case F_TOP_DN:
    case F_TOP_DN_DIR: case F_TOP_NET: case F_GROUND_VERTICAL_DIFF:
      return 0;
    case F_GROUND_SUNLIT_FRAC:
      return 2;
    case F_ROOF_SUNLIT_FRAC: case F_WALL_SUNLIT_FRAC: case F_VEG_SUNLIT_FRAC:
      return 3;
    default:
      return 1;
  }
}

// zero() of one column in one flux object
SS_HD void zero_unit(const Inputs& in, const FluxDev& f, int nspec, int col) {
  const int nlay = in.nlay[col];
  const int l0 = in.istartlay[col] - 1;
  for (int F = 0; F < F_NFIELDS; ++F) {
    double* a = f.f[F];
    if (!a) continue;
    const int shape = flux_field_shape(F);
    if (shape == 0) {
      for (int b = 0; b < nspec; ++b) a[(long)b + (long)nspec * col] = 0.0;
    } else if (shape == 2) {
      a[col] = 0.0;
    } else if (shape == 1) {
      for (int l = l0; l < l0 + nlay; ++l)
        for (int b = 0; b < nspec; ++b) a[(long)b + (long)nspec * l] = 0.0;
    } else {
      for (int l = l0; l < l0 + nlay; ++l) a[l] = 0.0;
    }
  }
}

// zero() of one band of one column (the per-column / per-layer scalars go with
// band 0); nlay layers from l0
SS_HD void zero_band_unit(const FluxDev& f, int nspec, int col, int l0, int nlay, int b) {
  for (int F = 0; F < F_NFIELDS; ++F) {
    double* a = f.f[F];
    if (!a) continue;
    const int shape = flux_field_shape(F);
    if (shape == 0) {
      a[(long)b + (long)nspec * col] = 0.0;
    } else if (shape == 2) {
      if (b == 0) a[col] = 0.0;
    } else if (shape == 1) {
      for (int l = l0; l < l0 + nlay; ++l) a[(long)b + (long)nspec * l] = 0.0;
    } else if (b == 0) {
      for (int l = l0; l < l0 + nlay; ++l) a[l] = 0.0;
    }
  }
}

// ---- single-layer urban tiles (types 4 / 5) ----

// View factors of radsurf_view_factor.F90: infinite street (:28-69) or the
// exponential model (:75-136, 8-point Gauss-Legendre on (0,1)).  vdg is only
// meaningful when cos_sza > 0.
SS_HD void simple_view_factors(bool infinite, double ratio, double cos_sza, bool want_dir,
                               double& vgs, double& vww, double& vdg) {
  vdg = 0.0;
  if (infinite) {
    vgs = sqrt(ratio * ratio + 1.0) - ratio;
    vww = sqrt(1.0 / (ratio * ratio) + 1.0) - 1.0 / ratio;
    if (want_dir) {
      const double norm_x0 = (kPi * 0.5) * ratio * sqrt(1.0 / (cos_sza * cos_sza) - 1.0);
      const double y_over_w = sqrt(ss_max(norm_x0 * norm_x0 - 1.0, 0.0));
      if (y_over_w > 0.0) vdg = (2.0 / kPi) * (y_over_w - norm_x0 + atan(1.0 / y_over_w));
      else vdg = 1.0 - 2.0 * norm_x0 / kPi;
    }
  } else {
    const double weights[8] = {0.0506142681451884, 0.111190517226687, 0.156853322938944,
                               0.181341891689181,  0.181341891689181, 0.156853322938944,
                               0.111190517226687,  0.0506142681451884};
    const double nodes[8] = {0.0198550717512319, 0.101666761293187, 0.237233795041836,
                             0.408282678752175,  0.591717321247825, 0.762766204958164,
                             0.898333238706813,  0.980144928248768};
    double sh = 0.0, sv = 0.0, vw[8];
    for (int k = 0; k < 8; ++k) sh += weights[k] * nodes[k];
    for (int k = 0; k < 8; ++k) { vw[k] = weights[k] * sqrt(1.0 - nodes[k] * nodes[k]); }
    for (int k = 0; k < 8; ++k) sv += vw[k];
    double a = 0.0, b = 0.0;
    for (int k = 0; k < 8; ++k) {
      const double tk = ratio * sqrt(1.0 / (nodes[k] * nodes[k]) - 1.0);
      const double e = exp(-tk);
      a += (weights[k] * nodes[k] / sh) * e;
      b += (vw[k] / sv) * (1.0 - e) / tk;
    }
    vgs = a;
    vww = 1.0 - b;
    if (want_dir) vdg = exp(-(ratio * sqrt(1.0 / (cos_sza * cos_sza) - 1.0)));
  }
}

// x = A^-1 b for the 2x2 interaction matrix, Cramer as radtool_matrix.F90:779-794
SS_HD void solve_vec_2(double a11, double a12, double a21, double a22, double b1, double b2,
                       double& x1, double& x2) {
  const double inv_det = 1.0 / (a11 * a22 - a12 * a21);
  x1 = inv_det * (a22 * b1 - a12 * b2);
  x2 = inv_det * (a11 * b2 - a21 * b1);
}

// One (column, band) unit of a type-4/5 tile; u = column-in-list * nmax + band,
// nmax = max(nsw, nlw).  Per-column members are indexed with the COLUMN: the
// reference uses the layer index for several of them
// (radsurf_simple_urban_sw.F90:200,255-275; _lw.F90:181-182,230-249), which is
// the same element whenever istartlay(col) == col (the reference's own
// test/single_layer) and a wrong one otherwise.  Like the reference, bc_out is
// not touched (its simple_urban_* never assign top_albedo / top_emissivity).
SS_HD void simple_urban_unit(const MiscLaunch& M, int u) {
  const int nmax = M.nsw > M.nlw ? M.nsw : M.nlw;
  const int col = M.cols[u / nmax];
  const int b = u % nmax;
  const int lay = M.in.istartlay[col] - 1;
#define PUT(fd, F, idx, v) do { if ((fd).f[F]) (fd).f[F][idx] = (v); } while (0)
  // geometry (sw:124-152, lw:118-143): calc_norm_perim_urban with one region
  const double dz = M.in.dz[lay];
  const double bf = M.in.building_fraction[lay];
  const double npw = (bf > M.min_building_fraction) ? 4.0 * bf / M.in.building_scale[lay] : 0.0;
  const double open = 1.0 - bf;
  const bool infinite = M.infinite_street != 0;
  const double length = infinite ? 2.0 * open / npw : kPi * open / npw;
  const double cos_sza = M.in.cos_sza ? M.in.cos_sza[col] : 0.0;
  const bool sun = M.do_sw && cos_sza > 0.0;
  double vgs, vww, vdg;
  simple_view_factors(infinite, dz / length, cos_sza, sun, vgs, vww, vdg);
  const double vdw = 1.0 - vdg, vwg = 0.5 * (1.0 - vww), vgw = 1.0 - vgs;

  if (M.do_sw && b < M.nsw) {
    zero_band_unit(M.sw_diff, M.nsw, col, lay, 1, b);
    zero_band_unit(M.sw_dir, M.nsw, col, lay, 1, b);
    if (sun) {
      const long c = (long)b + (long)M.nsw * col, l = (long)b + (long)M.nsw * lay;
      const double wall_albedo = M.in.wall_albedo[l], roof_albedo = M.in.roof_albedo[l];
      const double ga = M.in.ground_albedo[c], gad = M.in.ground_albedo_dir[c];
      const double a11 = 1.0, a12 = -vwg * wall_albedo, a21 = -vgw * ga, a22 = 1.0 - vww * wall_albedo;
      double x1, x2;
      // direct (sw:154-233)
      solve_vec_2(a11, a12, a21, a22, 0.0, (vdw + gad * vdg * vgw) * open, x1, x2);
      {
        const double gdd = vdg * open, gdn = gdd + x1;
        const double gnet = gdd * (1.0 - gad) + x1 * (1.0 - ga);
        const double wnet = x2 * (1.0 - wall_albedo);
        PUT(M.sw_dir, F_GROUND_DN_DIR, c, gdd);
        PUT(M.sw_dir, F_GROUND_DN, c, gdn);
        PUT(M.sw_dir, F_GROUND_NET, c, gnet);
        PUT(M.sw_dir, F_ROOF_IN_DIR, l, bf);
        PUT(M.sw_dir, F_ROOF_IN, l, bf);
        PUT(M.sw_dir, F_ROOF_NET, l, bf * (1.0 - roof_albedo));
        PUT(M.sw_dir, F_WALL_IN_DIR, l, vdw * open);
        PUT(M.sw_dir, F_WALL_IN, l, x2);
        PUT(M.sw_dir, F_WALL_NET, l, wnet);
        PUT(M.sw_dir, F_TOP_DN_DIR, c, 1.0);
        PUT(M.sw_dir, F_TOP_DN, c, 1.0);
        PUT(M.sw_dir, F_TOP_NET, c, 1.0 - bf * roof_albedo - (gdn - gnet) * vgs - (x2 - wnet) * vwg);
        PUT(M.sw_dir, F_FLUX_DN_DIR_LAYER_TOP, l, open);
        PUT(M.sw_dir, F_FLUX_DN_LAYER_TOP, l, open);
        PUT(M.sw_dir, F_FLUX_UP_LAYER_TOP, l, (gdn - gnet) * vgs + (x2 - wnet) * vwg);
        PUT(M.sw_dir, F_FLUX_DN_DIR_LAYER_BASE, l, gdd);
        PUT(M.sw_dir, F_FLUX_DN_LAYER_BASE, l, gdn);
        PUT(M.sw_dir, F_FLUX_UP_LAYER_BASE, l, gdn - gnet);
      }
      if (b == 0) {
        const double tan_sza = sqrt(1.0 / (cos_sza * cos_sza) - 1.0);
        PUT(M.sw_dir, F_GROUND_SUNLIT_FRAC, col, vdg);
        PUT(M.sw_dir, F_ROOF_SUNLIT_FRAC, lay, 1.0);
        PUT(M.sw_dir, F_WALL_SUNLIT_FRAC, lay,
            0.5 * vdw / (ss_max(tan_sza, 1.0e-6) * npw * dz / (kPi * open)));
      }
      // diffuse (sw:235-284)
      solve_vec_2(a11, a12, a21, a22, vgs * open, vgw * open, x1, x2);
      {
        const double gdn = x1, gnet = gdn * (1.0 - ga), wnet = x2 * (1.0 - wall_albedo);
        PUT(M.sw_diff, F_GROUND_DN, c, gdn);
        PUT(M.sw_diff, F_GROUND_NET, c, gnet);
        PUT(M.sw_diff, F_ROOF_IN, l, bf);
        PUT(M.sw_diff, F_ROOF_NET, l, bf * (1.0 - roof_albedo));
        PUT(M.sw_diff, F_WALL_IN, l, x2);
        PUT(M.sw_diff, F_WALL_NET, l, wnet);
        PUT(M.sw_diff, F_TOP_DN, c, 1.0);
        PUT(M.sw_diff, F_TOP_NET, c, 1.0 - bf * roof_albedo - (gdn - gnet) * vgs - (x2 - wnet) * vwg);
        PUT(M.sw_diff, F_FLUX_DN_LAYER_TOP, l, open);
        PUT(M.sw_diff, F_FLUX_UP_LAYER_TOP, l, (gdn - gnet) * vgs + (x2 - wnet) * vwg);
        PUT(M.sw_diff, F_FLUX_DN_LAYER_BASE, l, gdn);
        PUT(M.sw_diff, F_FLUX_UP_LAYER_BASE, l, gdn - gnet);
      }
    }
  }

  if (M.do_lw && b < M.nlw) {
    zero_band_unit(M.lw_norm, M.nlw, col, lay, 1, b);
    zero_band_unit(M.lw_int, M.nlw, col, lay, 1, b);
    const long c = (long)b + (long)M.nlw * col, l = (long)b + (long)M.nlw * lay;
    const double gem = M.in.ground_emissivity[c], gemission = M.in.ground_emission[c];
    const double rem = M.in.roof_emissivity[l], remission = M.in.roof_emission[l];
    const double wem = M.in.wall_emissivity[l], wemission = M.in.wall_emission[l];
    // (2,2) with the GROUND emissivity, as the reference has it (lw:157)
    const double a11 = 1.0, a12 = -vwg * (1.0 - wem), a21 = -vgw * (1.0 - gem), a22 = 1.0 - vww * (1.0 - gem);
    double x1, x2;
    // internal emission (lw:159-204)
    solve_vec_2(a11, a12, a21, a22, vwg * wemission * npw * dz,
                vgw * gemission * open + vww * wemission * npw * dz, x1, x2);
    {
      const double gnet = x1 * gem - gemission * open;
      const double wnet = x2 * wem - wemission * npw * dz;
      PUT(M.lw_int, F_GROUND_DN, c, x1);
      PUT(M.lw_int, F_GROUND_NET, c, gnet);
      PUT(M.lw_int, F_ROOF_NET, l, -bf * remission);
      PUT(M.lw_int, F_WALL_IN, l, x2);
      PUT(M.lw_int, F_WALL_NET, l, wnet);
      PUT(M.lw_int, F_TOP_NET, c, -bf * remission - (x1 - gnet) * vgs - (x2 - wnet) * vwg);
      PUT(M.lw_int, F_FLUX_UP_LAYER_TOP, l, (x1 - gnet) * vgs + (x2 - wnet) * vwg);
      PUT(M.lw_int, F_FLUX_DN_LAYER_BASE, l, x1);
      PUT(M.lw_int, F_FLUX_UP_LAYER_BASE, l, x1 - gnet);
    }
    // incoming diffuse (lw:206-247)
    solve_vec_2(a11, a12, a21, a22, vgs * open, vgw * open, x1, x2);
    {
      const double gnet = x1 * gem, wnet = x2 * wem;
      PUT(M.lw_norm, F_GROUND_DN, c, x1);
      PUT(M.lw_norm, F_GROUND_NET, c, gnet);
      PUT(M.lw_norm, F_ROOF_IN, l, bf);
      PUT(M.lw_norm, F_ROOF_NET, l, bf * rem);
      PUT(M.lw_norm, F_WALL_IN, l, x2);
      PUT(M.lw_norm, F_WALL_NET, l, wnet);
      PUT(M.lw_norm, F_TOP_DN, c, 1.0);
      PUT(M.lw_norm, F_TOP_NET, c, 1.0 - bf * (1.0 - rem) - (x1 - gnet) * vgs - (x2 - wnet) * vwg);
      PUT(M.lw_norm, F_FLUX_DN_LAYER_TOP, l, 1.0 - bf);
      PUT(M.lw_norm, F_FLUX_UP_LAYER_TOP, l, (x1 - gnet) * vgs + (x2 - wnet) * vwg);
      PUT(M.lw_norm, F_FLUX_DN_LAYER_BASE, l, x1);
      PUT(M.lw_norm, F_FLUX_UP_LAYER_BASE, l, x1 - gnet);
    }
  }
#undef PUT
}

// ---- canopy_flux%scale + %sum (radsurf_canopy_flux.F90:212-282, 399-460) as
// the driver applies them after radsurf (spartacus_surface_driver.F90:250-261):
//   a = fa * a,  b = fb * b  (in place; sunlit fractions are not scaled),
//   sum = a + b   (a member missing in an operand counts as zero).
struct ScaleSumLaunch {
  static constexpr int kMaxMembers = 16;   // members of one shape class (16 of shape (nspec, ntotlay))
  // members compacted per kind (filled by compact()): nm[kind] triplets
  double* ma[4][kMaxMembers];
  double* mb[4][kMaxMembers];
  double* ms[4][kMaxMembers];
  int nm[4];
  FluxDev a, b, sum;
  const double* fa;        // (nspec, ncol) factor of a, nullptr = a is not scaled
  const double* fb;        // (nspec, ncol) factor of b, nullptr = b is not scaled
  const int32_t* indcol;   // (ntotlay) 0-based column of every packed layer
  int nspec;
  long col0, ncols;        // column range (0-based first, count)
  long lay0, nlays;        // packed layer range
};

// Group the members by shape class so that the kernel's loop over them is dense
// (all loads of a thread are then issued back to back)
inline void scale_sum_compact(ScaleSumLaunch& S) {
  for (int k = 0; k < 4; ++k) S.nm[k] = 0;
  for (int F = 0; F < F_NFIELDS; ++F) {
    if (!S.a.f[F] && !S.b.f[F] && !S.sum.f[F]) continue;
    const int k = flux_field_shape(F);
    if (S.nm[k] >= ScaleSumLaunch::kMaxMembers) continue;   // cannot happen: 16 is the largest class
    const int i = S.nm[k]++;
    S.ma[k][i] = S.a.f[F];
    S.mb[k][i] = S.b.f[F];
    S.ms[k][i] = S.sum.f[F];
  }
}

// kind 0: (band, column) elements of every (nspec, ncol) member;
// kind 1: (band, layer) elements of every (nspec, ntotlay) member;
// kind 2 / 3: columns / layers of the unscaled (ncol) / (ntotlay) members.
// Unit i handles the aligned element pair (2i, 2i+1) of the kind's index space
// (16-byte accesses; arrays are 16-byte aligned), restricted to the plan's range.
SS_HD long scale_sum_pairs(const ScaleSumLaunch& S, int kind, long* first) {
  const long lo = kind == 0 ? (long)S.nspec * S.col0 : kind == 1 ? (long)S.nspec * S.lay0
                : kind == 2 ? S.col0 : S.lay0;
  const long n = kind == 0 ? (long)S.nspec * S.ncols : kind == 1 ? (long)S.nspec * S.nlays
               : kind == 2 ? S.ncols : S.nlays;
  *first = lo & ~1L;
  return n > 0 ? ((lo + n + 1) / 2 - lo / 2) : 0;
}

SS_HD void scale_sum_unit(const ScaleSumLaunch& S, int kind, long i) {
#if defined(__clang__)
#pragma clang fp contract(off)   // the reference rounds the product before the sum
#endif
  const long lo = kind == 0 ? (long)S.nspec * S.col0 : kind == 1 ? (long)S.nspec * S.lay0
                : kind == 2 ? S.col0 : S.lay0;
  const long hi = lo + (kind == 0 ? (long)S.nspec * S.ncols : kind == 1 ? (long)S.nspec * S.nlays
                        : kind == 2 ? S.ncols : S.nlays);
  const long e0 = (lo & ~1L) + 2 * i;
  const bool in0 = e0 >= lo && e0 < hi, in1 = e0 + 1 >= lo && e0 + 1 < hi;
  if (!in0 && !in1) return;
  const bool scaled = (kind <= 1);
  double fa0 = 1.0, fa1 = 1.0, fb0 = 1.0, fb1 = 1.0;
  if (scaled) {
    long f0 = e0, f1 = e0 + 1;
    if (kind == 1) {
      const long l0 = e0 / S.nspec, l1 = (e0 + 1) / S.nspec;
      f0 = (e0 - l0 * S.nspec) + (long)S.nspec * S.indcol[in0 ? l0 : l1];
      f1 = (e0 + 1 - l1 * S.nspec) + (long)S.nspec * S.indcol[in1 ? l1 : l0];
    }
    if (S.fa) { if (in0) fa0 = S.fa[f0]; if (in1) fa1 = S.fa[f1]; }
    if (S.fb) { if (in0) fb0 = S.fb[f0]; if (in1) fb1 = S.fb[f1]; }
  }
  const bool both = in0 && in1;
  const int nm = S.nm[kind];
  // loads of all members first, then the arithmetic and the stores
  RecPair va[ScaleSumLaunch::kMaxMembers], vb[ScaleSumLaunch::kMaxMembers];
  SS_UNROLL_FULL
  for (int m = 0; m < ScaleSumLaunch::kMaxMembers; ++m) {
    va[m].x = va[m].y = vb[m].x = vb[m].y = 0.0;
    if (m < nm) {
      const double* pa = S.ma[kind][m];
      const double* pb = S.mb[kind][m];
      if (pa) {
        if (both) va[m] = *reinterpret_cast<const RecPair*>(pa + e0);
        else if (in0) va[m].x = pa[e0];
        else va[m].y = pa[e0 + 1];
      }
      if (pb) {
        if (both) vb[m] = *reinterpret_cast<const RecPair*>(pb + e0);
        else if (in0) vb[m].x = pb[e0];
        else vb[m].y = pb[e0 + 1];
      }
    }
  }
  SS_UNROLL_FULL
  for (int m = 0; m < ScaleSumLaunch::kMaxMembers; ++m) {
    if (m < nm) {
      double* pa = S.ma[kind][m];
      double* pb = S.mb[kind][m];
      double* ps = S.ms[kind][m];
      if (pa && scaled && S.fa) {
        va[m].x = fa0 * va[m].x;
        va[m].y = fa1 * va[m].y;
        if (both) *reinterpret_cast<RecPair*>(pa + e0) = va[m];
        else if (in0) pa[e0] = va[m].x;
        else pa[e0 + 1] = va[m].y;
      }
      if (pb && scaled && S.fb) {
        vb[m].x = fb0 * vb[m].x;
        vb[m].y = fb1 * vb[m].y;
        if (both) *reinterpret_cast<RecPair*>(pb + e0) = vb[m];
        else if (in0) pb[e0] = vb[m].x;
        else pb[e0 + 1] = vb[m].y;
      }
      if (ps) {
        RecPair vs;
        vs.x = va[m].x + vb[m].x;
        vs.y = va[m].y + vb[m].y;
        if (both) *reinterpret_cast<RecPair*>(ps + e0) = vs;
        else if (in0) ps[e0] = vs.x;
        else ps[e0 + 1] = vs.y;
      }
    }
  }
}

// calc_simple_spectrum_lw: per-column and per-layer parts
struct SpectrumLaunch {
  const double* ground_temperature;
  const double* roof_temperature;
  const double* wall_temperature;
  const double* clear_air_temperature;
  const double* veg_temperature;
  const double* veg_air_temperature;
  const double* ground_emissivity;
  const double* roof_emissivity;
  const double* wall_emissivity;
  double* ground_emission;
  double* roof_emission;
  double* wall_emission;
  double* clear_air_planck;
  double* veg_planck;
  double* veg_air_planck;
  int col0, ncols;     // 0-based first column, count
  int lay0, nlays;     // 0-based first layer, count
  int do_vegetation;
};

SS_HD double pow4(double t) { return (t * t) * (t * t); }

SS_HD void spectrum_col_unit(const SpectrumLaunch& S, int i) {
  const int c = S.col0 + i;
  S.ground_emission[c] = kStefanBoltzmann * S.ground_emissivity[c] * pow4(S.ground_temperature[c]);
}

SS_HD void spectrum_lay_unit(const SpectrumLaunch& S, int i) {
  const int l = S.lay0 + i;
  if (S.roof_emission) {
    S.roof_emission[l] = kStefanBoltzmann * S.roof_emissivity[l] * pow4(S.roof_temperature[l]);
    S.wall_emission[l] = kStefanBoltzmann * S.wall_emissivity[l] * pow4(S.wall_temperature[l]);
  }
  S.clear_air_planck[l] = kStefanBoltzmann * pow4(S.clear_air_temperature[l]);
  if (S.do_vegetation) {
    S.veg_planck[l] = kStefanBoltzmann * pow4(S.veg_temperature[l]);
    S.veg_air_planck[l] = kStefanBoltzmann * pow4(S.veg_air_temperature[l]);
  }
}

}  // namespace ss
