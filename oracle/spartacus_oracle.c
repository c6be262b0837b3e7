/* spartacus_oracle.c - plain-C restatement of the SPARTACUS-Surface column
 * solver (radsurf layer of the reference).
 *
 * TEST INFRASTRUCTURE ONLY (see spartacus_oracle.h): the CPU checker that the
 * HIP path is compared against in tests/, smoke() and bench.py's cpu_baseline
 * leg.  Parity status: PINNED - tests/test_oracle_vs_reference.py compares
 * this file against the real reference compiled from /root/reference
 * (oracle/_ref) on every test input the reference ships, and
 * tests/test_golden.py against the committed golden vectors.
 *
 * One spectral band at a time (the reference vectorises over bands; bands are
 * independent except for the index "itransp" of the most transparent band,
 * radsurf_forest_sw.F90:253).  Forest and urban solvers share one routine per
 * spectral region, switched by is_urban, with the arithmetic of each
 * reference routine kept where the two differ.
 */
#include <stdlib.h>
#include <stdio.h>
#include <omp.h>

#include "spartacus_oracle.h"
#include "oracle_radtool.h"
#include "oracle_geometry.h"

static const or_real Pi = 3.14159265358979323846;             /* radiation_constants.F90:24 */
static const or_real StefanBoltzmann = 5.67037321e-8;         /* radiation_constants.F90:26 */

#define MAXREG 3
#define MAXS SS_MAX_STREAMS

/* Per-column workspace: the reference's automatic arrays
 * (radsurf_forest_sw.F90:107-199, radsurf_urban_sw.F90:112-210). */
typedef struct {
  int n, nb, nlay;             /* n = nreg*ns, nb = "below" order */
  or_real *trans_diff, *ref_diff, *ref_dir, *trans_dir_diff, *trans_dir_dir;
  or_real *int_diff, *int_dir, *int_dir_diff;
  or_real *a_above, *d_above, *a_below, *d_below, *denominator;
  or_real *source_lay, *int_source, *source_above, *source_below;   /* LW */
  or_real *frac, *norm_perim, *norm_perim_wall, *u_overlap, *v_overlap;
  or_real *od_scaling, *f_wall, *f_wall_dir_clear;
  or_real *emiss_reg, *emiss_air, *emiss_veg, *emiss_wall;
} or_work_t;

static or_real *dalloc(size_t n) {
  or_real *p = (or_real *)calloc(n ? n : 1, sizeof(or_real));
  if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
  return p;
}

static void work_alloc(or_work_t *w, int nreg, int ns, int nlay, int is_urban) {
  const int n = nreg * ns, d = nreg;
  const int nb = is_urban ? (nreg + 1) * ns : n;
  const int db = is_urban ? nreg + 1 : nreg;
  w->n = n; w->nb = nb; w->nlay = nlay;
  w->trans_diff = dalloc((size_t)n * n * nlay);
  w->ref_diff = dalloc((size_t)n * n * nlay);
  w->ref_dir = dalloc((size_t)n * d * nlay);
  w->trans_dir_diff = dalloc((size_t)n * d * nlay);
  w->trans_dir_dir = dalloc((size_t)d * d * nlay);
  w->int_diff = dalloc((size_t)n * n * nlay);
  w->int_dir = dalloc((size_t)d * d * nlay);
  w->int_dir_diff = dalloc((size_t)n * d * nlay);
  w->a_above = dalloc((size_t)n * n * (nlay + 1));
  w->d_above = dalloc((size_t)n * d * (nlay + 1));
  w->a_below = dalloc((size_t)nb * nb * (nlay + 1));
  w->d_below = dalloc((size_t)nb * db * (nlay + 1));
  w->denominator = dalloc((size_t)n * n * nlay);
  w->source_lay = dalloc((size_t)n * nlay);
  w->int_source = dalloc((size_t)n * nlay);
  w->source_above = dalloc((size_t)n * (nlay + 1));
  w->source_below = dalloc((size_t)nb * (nlay + 1));
  w->frac = dalloc((size_t)d * (nlay + 1));
  w->norm_perim = dalloc((size_t)d * nlay);
  w->norm_perim_wall = dalloc((size_t)d * nlay);
  w->u_overlap = dalloc((size_t)d * db * (nlay + 1));
  w->v_overlap = dalloc((size_t)db * d * (nlay + 1));
  w->od_scaling = dalloc((size_t)(MAXREG + 1) * nlay);
  w->f_wall = dalloc((size_t)d * nlay);
  w->f_wall_dir_clear = dalloc((size_t)nlay);
  w->emiss_reg = dalloc((size_t)d * nlay);
  w->emiss_air = dalloc((size_t)(MAXREG + 1) * nlay);
  w->emiss_veg = dalloc((size_t)(MAXREG + 1) * nlay);
  w->emiss_wall = dalloc((size_t)nlay);
}

static void work_free(or_work_t *w) {
  free(w->trans_diff); free(w->ref_diff); free(w->ref_dir); free(w->trans_dir_diff);
  free(w->trans_dir_dir); free(w->int_diff); free(w->int_dir); free(w->int_dir_diff);
  free(w->a_above); free(w->d_above); free(w->a_below); free(w->d_below);
  free(w->denominator); free(w->source_lay); free(w->int_source); free(w->source_above);
  free(w->source_below); free(w->frac); free(w->norm_perim); free(w->norm_perim_wall);
  free(w->u_overlap); free(w->v_overlap); free(w->od_scaling); free(w->f_wall);
  free(w->f_wall_dir_clear); free(w->emiss_reg); free(w->emiss_air); free(w->emiss_veg);
  free(w->emiss_wall);
}

/* zero_canopy_flux (radsurf_canopy_flux.F90:286-341); icol, l0 are 0-based */
static void flux_zero(ss_flux *f, int icol, int l0, int nlay) {
  const int ns = f->nspec;
#define ZC(a) do { if (f->a) for (int b = 0; b < ns; ++b) f->a[b + (size_t)ns * icol] = 0.0; } while (0)
#define ZL(a) do { if (f->a) for (int l = l0; l < l0 + nlay; ++l) for (int b = 0; b < ns; ++b) f->a[b + (size_t)ns * l] = 0.0; } while (0)
#define ZL1(a) do { if (f->a) for (int l = l0; l < l0 + nlay; ++l) f->a[l] = 0.0; } while (0)
  ZC(ground_dn); ZC(ground_net); ZC(ground_vertical_diff); ZC(top_dn); ZC(top_net);
  if (f->ground_dn_dir) {
    ZC(ground_dn_dir); ZC(top_dn_dir);
    if (f->ground_sunlit_frac) f->ground_sunlit_frac[icol] = 0.0;
  }
  if (nlay >= 1) {
    ZL(roof_in); ZL(roof_net); ZL(wall_in); ZL(wall_net);
    ZL(roof_in_dir); ZL(wall_in_dir); ZL1(roof_sunlit_frac); ZL1(wall_sunlit_frac);
    ZL(clear_air_abs); ZL(veg_abs); ZL(veg_air_abs); ZL(veg_abs_dir); ZL1(veg_sunlit_frac);
    ZL(flux_dn_layer_top); ZL(flux_up_layer_top); ZL(flux_dn_layer_base); ZL(flux_up_layer_base);
    ZL(flux_dn_dir_layer_top); ZL(flux_dn_dir_layer_base);
  }
#undef ZC
#undef ZL
#undef ZL1
}

/* Region fractions: forest SW radsurf_forest_sw.F90:244-248, forest LW
 * radsurf_forest_lw.F90:244-248, urban radsurf_urban_sw.F90:284-291 */
static void region_fractions(int is_urban, int is_lw, int nreg, int nlay,
                             const double *veg_fraction, const double *building_fraction,
                             or_real *frac) {
  for (int l = 0; l < nlay; ++l) {
    or_real *fr = frac + (size_t)l * nreg;
    if (!is_urban) {
      fr[0] = 1.0 - veg_fraction[l];
      for (int r = 1; r < nreg; ++r)
        fr[r] = (is_lw ? (1.0 - fr[0]) : veg_fraction[l]) / (or_real)(nreg - 1);
    } else {
      fr[0] = 1.0 - building_fraction[l];
      if (nreg > 1) {
        fr[0] = OR_FMAX(0.0, fr[0] - veg_fraction[l]);
        for (int r = 1; r < nreg; ++r)
          fr[r] = OR_FMAX(0.0, 1.0 - building_fraction[l] - fr[0]) / (or_real)(nreg - 1);
      }
    }
  }
  frac[(size_t)nlay * nreg] = 1.0;
  for (int r = 1; r < nreg; ++r) frac[(size_t)nlay * nreg + r] = 0.0;
}

/* f_exchange (radsurf_forest_sw.F90:299-321 and the identical blocks of the
 * other three solvers).  Note the use of norm_perim(jreg) after the loop, i.e.
 * jreg == nreg. */
static void exchange_rates(int nreg, const or_real *frac, const or_real *norm_perim,
                           or_real min_veg, or_real *f_exchange) {
  for (int k = 0; k < nreg * nreg; ++k) f_exchange[k] = 0.0;
  for (int jreg = 0; jreg < nreg - 1; ++jreg) {
    if (frac[jreg] <= min_veg || frac[jreg + 1] <= min_veg) {
      E(f_exchange, jreg + 1, jreg, nreg) = 0.0;
      E(f_exchange, jreg, jreg + 1, nreg) = 0.0;
    } else {
      E(f_exchange, jreg + 1, jreg, nreg) = norm_perim[jreg] / (Pi * frac[jreg]);
      E(f_exchange, jreg, jreg + 1, nreg) = norm_perim[jreg] / (Pi * frac[jreg + 1]);
    }
  }
  if (nreg > 2 && norm_perim[nreg - 1] > 0.0) {
    if (frac[2] <= min_veg || frac[0] <= min_veg) {
      E(f_exchange, 0, 2, nreg) = 0.0;
      E(f_exchange, 2, 0, nreg) = 0.0;
    } else {
      E(f_exchange, 0, 2, nreg) = norm_perim[nreg - 1] / (Pi * frac[2]);
      E(f_exchange, 2, 0, nreg) = norm_perim[nreg - 1] / (Pi * frac[0]);
    }
  }
}

/* (U (x) I_ns) applied from the left: rect_expandedmat_x_mat
 * (radtool_matrix.F90:505-550).  A is m x o, B is (o*s) x p with leading
 * dimension ldb, C is (m*s) x p. */
static void expandedmat_x_mat(int m, int o, int s, int p, const or_real *A, const or_real *B,
                              int ldb, or_real *C) {
  const int ldc = m * s;
  for (int k = 0; k < ldc * p; ++k) C[k] = 0.0;
  for (int j1 = 0; j1 < m; ++j1)
    for (int j3 = 0; j3 < o; ++j3)
      if (E(A, j1, j3, m) != 0.0) {
        const int offset2 = (j3 - j1) * s;
        for (int jj2 = 0; jj2 < p; ++jj2)
          for (int jj1 = j1 * s; jj1 < (j1 + 1) * s; ++jj1)
            E(C, jj1, jj2, ldc) += E(A, j1, j3, m) * E(B, jj1 + offset2, jj2, ldb);
      }
}

/* rect_mat_x_expandedmat (radtool_matrix.F90:556-601): A is p x (m*s),
 * B is m x o, C is p x (o*s) */
static void mat_x_expandedmat(int m, int o, int s, int p, const or_real *A, const or_real *B,
                              or_real *C) {
  for (int k = 0; k < p * o * s; ++k) C[k] = 0.0;
  for (int j2 = 0; j2 < o; ++j2)
    for (int j3 = 0; j3 < m; ++j3)
      if (E(B, j3, j2, m) != 0.0) {
        const int offset3 = (j3 - j2) * s;
        for (int jj1 = 0; jj1 < p; ++jj1)
          for (int jj2 = j2 * s; jj2 < (j2 + 1) * s; ++jj2)
            E(C, jj1, jj2, p) += E(A, jj1, jj2 + offset3, p) * E(B, j3, j2, m);
      }
}

/* rect_expandedmat_x_vec (radtool_matrix.F90:608-649): A m x k, b (k*s), y (m*s) */
static void expandedmat_x_vec(int m, int k, int s, const or_real *A, const or_real *b, or_real *y) {
  for (int i = 0; i < m * s; ++i) y[i] = 0.0;
  for (int j1 = 0; j1 < m; ++j1)
    for (int j3 = 0; j3 < k; ++j3)
      if (E(A, j1, j3, m) != 0.0) {
        const int offset2 = (j3 - j1) * s;
        for (int jj1 = j1 * s; jj1 < (j1 + 1) * s; ++jj1) y[jj1] += E(A, j1, j3, m) * b[jj1 + offset2];
      }
}

/* Copy an (nsub x msub) result into the sub-block starting at (r0,c0) of a
 * zeroed (ld x *) matrix */
static void put_block(or_real *dst, int ld, int r0, int c0, const or_real *src, int nsub, int msub) {
  for (int j = 0; j < msub; ++j)
    for (int i = 0; i < nsub; ++i) E(dst, r0 + i, c0 + j, ld) = E(src, i, j, nsub);
}
static void get_block(const or_real *src, int ld, int r0, int c0, or_real *dst, int nsub, int msub) {
  for (int j = 0; j < msub; ++j)
    for (int i = 0; i < nsub; ++i) E(dst, i, j, nsub) = E(src, r0 + i, c0 + j, ld);
}

/* od_scaling / ext_reg / ssa_reg [/ planck_reg] for one layer and band
 * (radsurf_forest_sw.F90:277-297, radsurf_forest_lw.F90:278-300 and the urban
 * equivalents).  od[r] valid for r = 1..nreg-1. */
static void region_optics(int nreg, or_real air_ext, or_real air_ssa, or_real veg_ext, or_real veg_ssa,
                          or_real veg_fsd, or_real *od, or_real *ext_reg, or_real *ssa_reg) {
  ext_reg[0] = air_ext;
  ssa_reg[0] = air_ssa;
  if (nreg == 2) {
    ext_reg[1] = air_ext + veg_ext;
    ssa_reg[1] = (ext_reg[0] * ssa_reg[0] + veg_ext * veg_ssa) / OR_FMAX(ext_reg[1], 1.0e-8);
    od[1] = 1.0;
  } else if (nreg == 3) {
    od[1] = OR_EXP(-veg_fsd * (1.0 + 0.5 * veg_fsd * (1.0 + 0.5 * veg_fsd)));
    od[2] = 2.0 - od[1];
    ext_reg[1] = air_ext + od[1] * veg_ext;
    ext_reg[2] = air_ext + od[2] * veg_ext;
    ssa_reg[1] = (ext_reg[0] * ssa_reg[0] + od[1] * veg_ext * veg_ssa) / OR_FMAX(ext_reg[1], 1.0e-8);
    ssa_reg[2] = (ext_reg[0] * ssa_reg[0] + od[2] * veg_ext * veg_ssa) / OR_FMAX(ext_reg[2], 1.0e-8);
  }
}

/* =======================================================================
 * Shortwave: spartacus_forest_sw (radsurf_forest_sw.F90:35-783) and
 * spartacus_urban_sw (radsurf_urban_sw.F90:35-1007), one column.
 * icol, l0 0-based (l0 = istartlay-1).
 * ======================================================================= */
static int solve_sw(const ss_config *cfg, int is_urban, int nsw, int ns, int nreg, int nlay,
                    int icol, int l0, const ss_legendre_gauss *lg, or_real cos_sza_in,
                    const ss_canopy *cp, const ss_sw_spectral *sp, const double *ground_albedo_diff,
                    const double *ground_albedo_dir, double *top_albedo_diff, double *top_albedo_dir,
                    ss_flux *fdir, ss_flux *fdiff) {
  const int n = nreg * ns, d = nreg;
  const int db = is_urban ? nreg + 1 : nreg; /* regions just below an interface */
  const int nb = db * ns;
  const int do_vegetation = (nreg > 1);
  or_work_t w;
  int ierr = 0;

  const double *dz = cp->dz + l0;
  const double *veg_fraction = cp->veg_fraction ? cp->veg_fraction + l0 : NULL;
  const double *veg_scale = cp->veg_scale ? cp->veg_scale + l0 : NULL;
  const double *veg_ext = cp->veg_ext ? cp->veg_ext + l0 : NULL;
  const double *veg_fsd = cp->veg_fsd ? cp->veg_fsd + l0 : NULL;
  const double *building_fraction = cp->building_fraction ? cp->building_fraction + l0 : NULL;
  const double *building_scale = cp->building_scale ? cp->building_scale + l0 : NULL;
  const double *veg_contact = cp->veg_contact_fraction ? cp->veg_contact_fraction + l0 : NULL;
#define SPEC(a, b, l) ((a)[(b) + (size_t)nsw * (l0 + (l))])

  /* urban: corrected cosine used everywhere (radsurf_urban_sw.F90:268);
   * forest: clamp only inside tan0 (radsurf_forest_sw.F90:241) */
  const or_real cos_sza = is_urban ? OR_FMAX(cos_sza_in, 1.0e-6) : cos_sza_in;
  or_real sin0 = 0.0, tan0;
  or_real roof_fraction[512], non_building_fraction[513];
  int itransp = 0;

  if (is_urban && do_vegetation && !cfg->do_vegetation) return SS_ERR_VEG_DISABLED;
  if (nlay > 511) return SS_ERR_BAD_ARGUMENT;

  if (is_urban) {
    sin0 = OR_SQRT(1.0 - cos_sza * cos_sza);
    tan0 = sin0 / cos_sza;
  } else {
    tan0 = OR_SQRT(1.0 - cos_sza * cos_sza) / OR_FMAX(cos_sza, 1.0e-6);
  }

  work_alloc(&w, nreg, ns, nlay, is_urban);
  region_fractions(is_urban, 0, nreg, nlay, veg_fraction, building_fraction, w.frac);

  if (is_urban) {
    /* radsurf_urban_sw.F90:293-298 */
    roof_fraction[nlay] = 0.0;
    roof_fraction[nlay - 1] = building_fraction[nlay - 1];
    for (int l = 0; l < nlay - 1; ++l)
      roof_fraction[l] = OR_FMAX(0.0, building_fraction[l] - building_fraction[l + 1]);
    non_building_fraction[nlay] = 1.0;
    for (int l = 0; l < nlay; ++l) non_building_fraction[l] = 1.0 - building_fraction[l];
  }

  /* most transparent band: minloc(sum(air_ext*dz)) :253 */
  {
    or_real best = 0.0;
    for (int b = 0; b < nsw; ++b) {
      or_real od = 0.0;
      for (int l = 0; l < nlay; ++l) od += SPEC(sp->air_ext, b, l) * dz[l];
      if (b == 0 || od < best) { best = od; itransp = b; }
    }
  }

  if (is_urban) {
    or_calc_overlap_matrices_urban(nlay, nreg, w.frac, nreg, w.u_overlap, w.v_overlap,
                                   cfg->min_vegetation_fraction);
    or_calc_norm_perim_urban(cfg, nlay, nreg, building_fraction, building_scale, veg_fraction,
                             veg_scale, veg_contact, w.norm_perim, w.norm_perim_wall);
  } else {
    or_calc_overlap_matrices(nlay, nreg, w.frac, nreg, w.u_overlap, w.v_overlap,
                             cfg->min_vegetation_fraction);
    or_calc_norm_perim_forest(cfg, nlay, nreg, veg_fraction, veg_scale, w.norm_perim);
  }

  /* Section 5 preamble: zero the outputs of this column (:535-536) */
  flux_zero(fdiff, icol, l0, nlay);
  flux_zero(fdir, icol, l0, nlay);

  /* clear-sky direct flux chain for the sunlit fractions is band independent
   * apart from air_ext(itransp); it is evaluated inside band itransp */

  for (int jb = 0; jb < nsw; ++jb) {
    or_real gamma0[9], gamma1[OR_MAXN * OR_MAXN], gamma2[OR_MAXN * OR_MAXN], gamma3[OR_MAXN * 3];
    or_real f_exchange[9], ext_reg[3], ssa_reg[3];
    or_real sub0[9], sub1[OR_MAXN * OR_MAXN], sub2[OR_MAXN * OR_MAXN], sub3[OR_MAXN * 3];
    or_real oR[OR_MAXN * OR_MAXN], oT[OR_MAXN * OR_MAXN], oSu[OR_MAXN * 3], oSd[OR_MAXN * 3];
    or_real oE[9], oId[9], oIf[OR_MAXN * OR_MAXN], oIdf[OR_MAXN * 3];

    /* ---- Section 3: first loop over layers ---- */
    for (int jlay = 0; jlay < nlay; ++jlay) {
      const or_real *frac = w.frac + (size_t)jlay * nreg;
      const or_real *norm_perim = w.norm_perim + (size_t)jlay * nreg;
      or_real *od = w.od_scaling + (size_t)jlay * (MAXREG + 1);
      or_real *f_wall = w.f_wall + (size_t)jlay * nreg;
      or_real wall_ext = 0.0, wall_factor = 0.0;
      const or_real air_ext = SPEC(sp->air_ext, jb, jlay), air_ssa = SPEC(sp->air_ssa, jb, jlay);

      region_optics(nreg, air_ext, air_ssa, do_vegetation ? veg_ext[jlay] : 0.0,
                    do_vegetation ? SPEC(sp->veg_ssa, jb, jlay) : 0.0,
                    (nreg == 3) ? veg_fsd[jlay] : 0.0, od, ext_reg, ssa_reg);
      exchange_rates(nreg, frac, norm_perim, cfg->min_vegetation_fraction, f_exchange);

      for (int r = 0; r < nreg; ++r) f_wall[r] = 0.0;
      if (is_urban) {
        /* radsurf_urban_sw.F90:395-418 */
        const or_real *npw = w.norm_perim_wall + (size_t)jlay * nreg;
        for (int r = 0; r < nreg; ++r) {
          if (frac[r] <= cfg->min_vegetation_fraction) f_wall[r] = 0.0;
          else f_wall[r] = npw[r] / (Pi * frac[r]);
        }
        if (non_building_fraction[jlay] <= cfg->min_building_fraction) {
          w.f_wall_dir_clear[jlay] = 0.0;
        } else {
          or_real s = 0.0;
          for (int r = 0; r < nreg; ++r) s += npw[r];
          w.f_wall_dir_clear[jlay] = s / (Pi * non_building_fraction[jlay]);
        }
        wall_ext = 1.0 - SPEC(sp->wall_albedo, jb, jlay) * SPEC(sp->wall_specular_frac, jb, jlay);
        wall_factor = SPEC(sp->wall_albedo, jb, jlay) * (1.0 - SPEC(sp->wall_specular_frac, jb, jlay));
      }

      /* ---- Section 3b: Gamma matrices (:329-387 / urban :424-497) ---- */
      for (int k = 0; k < d * d; ++k) gamma0[k] = 0.0;
      for (int k = 0; k < n * n; ++k) { gamma1[k] = 0.0; gamma2[k] = 0.0; }
      for (int k = 0; k < n * d; ++k) gamma3[k] = 0.0;
      for (int jfr = 0; jfr < nreg; ++jfr)
        for (int jto = 0; jto < nreg; ++jto)
          if (jfr != jto) {
            E(gamma0, jfr, jfr, d) = E(gamma0, jfr, jfr, d) - tan0 * E(f_exchange, jto, jfr, nreg);
            E(gamma0, jto, jfr, d) = +tan0 * E(f_exchange, jto, jfr, nreg);
            for (int js = 0; js < ns; ++js) {
              const int ifr = js + jfr * ns, ito = js + jto * ns;
              E(gamma1, ifr, ifr, n) = E(gamma1, ifr, ifr, n) - lg->tan_ang[js] * E(f_exchange, jto, jfr, nreg);
              E(gamma1, ito, ifr, n) = +lg->tan_ang[js] * E(f_exchange, jto, jfr, nreg);
            }
          }
      for (int jreg = 0; jreg < nreg; ++jreg) {
        if (is_urban)
          E(gamma0, jreg, jreg, d) = E(gamma0, jreg, jreg, d) - ext_reg[jreg] / cos_sza
                                     - tan0 * f_wall[jreg] * wall_ext;
        else
          E(gamma0, jreg, jreg, d) = E(gamma0, jreg, jreg, d) - ext_reg[jreg] / cos_sza;
        for (int js = 0; js < ns; ++js) {
          const int ifr = js + jreg * ns;
          if (is_urban)
            E(gamma1, ifr, ifr, n) = E(gamma1, ifr, ifr, n) - ext_reg[jreg] / lg->mu[js]
                                     - lg->tan_ang[js] * f_wall[jreg] * wall_ext;
          else
            E(gamma1, ifr, ifr, n) = E(gamma1, ifr, ifr, n) - ext_reg[jreg] / lg->mu[js];
        }
      }
      for (int jsfr = 0; jsfr < ns; ++jsfr)
        for (int jsto = 0; jsto < ns; ++jsto)
          for (int jreg = 0; jreg < nreg; ++jreg) {
            const int ifr = jsfr + jreg * ns, ito = jsto + jreg * ns;
            if (is_urban)
              E(gamma2, ito, ifr, n) = 0.5 * (lg->weight[jsto] * ext_reg[jreg] * ssa_reg[jreg] / lg->mu[jsfr]
                                              + lg->vweight[jsto] * lg->tan_ang[jsfr] * f_wall[jreg] * wall_factor);
            else
              E(gamma2, ito, ifr, n) = 0.5 * lg->weight[jsto] * ext_reg[jreg] * ssa_reg[jreg] / lg->mu[jsfr];
          }
      for (int k = 0; k < n * n; ++k) gamma1[k] = gamma1[k] + gamma2[k];
      for (int jreg = 0; jreg < nreg; ++jreg)
        for (int js = 0; js < ns; ++js) {
          const int ito = js + jreg * ns;
          if (is_urban)
            E(gamma3, ito, jreg, n) = 0.5 * (lg->weight[js] * ext_reg[jreg] * ssa_reg[jreg]
                                             + lg->vweight[js] * sin0 * f_wall[jreg] * wall_factor);
          else
            E(gamma3, ito, jreg, n) = 0.5 * lg->weight[js] * ext_reg[jreg] * ssa_reg[jreg];
        }

      /* ---- Section 3c: layer matrices, 3-way branch (:408-455 / :520-583) ---- */
      {
        or_real *Rl = w.ref_diff + (size_t)jlay * n * n, *Tl = w.trans_diff + (size_t)jlay * n * n;
        or_real *Sul = w.ref_dir + (size_t)jlay * n * d, *Sdl = w.trans_dir_diff + (size_t)jlay * n * d;
        or_real *El = w.trans_dir_dir + (size_t)jlay * d * d;
        or_real *Idl = w.int_dir + (size_t)jlay * d * d, *Ifl = w.int_diff + (size_t)jlay * n * n;
        or_real *Idfl = w.int_dir_diff + (size_t)jlay * n * d;
        int r0, nr; /* first region and number of regions in the sub-problem */
        if (!do_vegetation || veg_fraction[jlay] <= cfg->min_vegetation_fraction) {
          r0 = 0; nr = 1;
        } else if (frac[0] <= cfg->min_vegetation_fraction) {
          r0 = 1; nr = nreg - 1;
        } else {
          r0 = 0; nr = nreg;
        }
        for (int k = 0; k < n * n; ++k) { Rl[k] = 0.0; Tl[k] = 0.0; Ifl[k] = 0.0; }
        for (int k = 0; k < n * d; ++k) { Sul[k] = 0.0; Sdl[k] = 0.0; Idfl[k] = 0.0; }
        for (int k = 0; k < d * d; ++k) { El[k] = 0.0; Idl[k] = 0.0; }
        {
          const int nsub = nr * ns, i0 = r0 * ns;
          get_block(gamma0, d, r0, r0, sub0, nr, nr);
          get_block(gamma1, n, i0, i0, sub1, nsub, nsub);
          get_block(gamma2, n, i0, i0, sub2, nsub, nsub);
          get_block(gamma3, n, i0, r0, sub3, nsub, nr);
          ierr += or_calc_matrices_sw_eig(nsub, nr, dz[jlay], cos_sza, sub0, sub1, sub2, sub3,
                                          oR, oT, oSu, oSd, oE, oId, oIf, oIdf);
          put_block(Rl, n, i0, i0, oR, nsub, nsub);
          put_block(Tl, n, i0, i0, oT, nsub, nsub);
          put_block(Sul, n, i0, r0, oSu, nsub, nr);
          put_block(Sdl, n, i0, r0, oSd, nsub, nr);
          put_block(El, d, r0, r0, oE, nr, nr);
          put_block(Idl, d, r0, r0, oId, nr, nr);
          put_block(Ifl, n, i0, i0, oIf, nsub, nsub);
          put_block(Idfl, n, i0, r0, oIdf, nsub, nr);
        }
      }
    } /* layers */

    /* ---- Section 4: albedo of the scene at each interface (:463-504 / :591-654) ---- */
    for (size_t k = 0; k < (size_t)n * n * (nlay + 1); ++k) w.a_above[k] = 0.0;
    for (size_t k = 0; k < (size_t)n * d * (nlay + 1); ++k) w.d_above[k] = 0.0;
    for (size_t k = 0; k < (size_t)nb * nb * (nlay + 1); ++k) w.a_below[k] = 0.0;
    for (size_t k = 0; k < (size_t)nb * db * (nlay + 1); ++k) w.d_below[k] = 0.0;
    for (int jreg = 0; jreg < nreg; ++jreg)
      for (int jsto = 0; jsto < ns; ++jsto) {
        E(w.d_above, jsto + jreg * ns, jreg, n) = cos_sza * ground_albedo_dir[jb] * lg->hweight[jsto];
        for (int jsfr = 0; jsfr < ns; ++jsfr)
          E(w.a_above, jsto + jreg * ns, jsfr + jreg * ns, n) = ground_albedo_diff[jb] * lg->hweight[jsto];
      }

    for (int jlay = 0; jlay < nlay; ++jlay) {
      const or_real *Rl = w.ref_diff + (size_t)jlay * n * n, *Tl = w.trans_diff + (size_t)jlay * n * n;
      const or_real *Sul = w.ref_dir + (size_t)jlay * n * d, *Sdl = w.trans_dir_diff + (size_t)jlay * n * d;
      const or_real *El = w.trans_dir_dir + (size_t)jlay * d * d;
      const or_real *Aab = w.a_above + (size_t)jlay * n * n, *Dab = w.d_above + (size_t)jlay * n * d;
      or_real *Den = w.denominator + (size_t)jlay * n * n;
      or_real *Abe = w.a_below + (size_t)(jlay + 1) * nb * nb, *Dbe = w.d_below + (size_t)(jlay + 1) * nb * db;
      or_real *Aab1 = w.a_above + (size_t)(jlay + 1) * n * n, *Dab1 = w.d_above + (size_t)(jlay + 1) * n * d;
      const or_real *U = w.u_overlap + (size_t)(jlay + 1) * d * db;
      const or_real *V = w.v_overlap + (size_t)(jlay + 1) * db * d;
      or_real t1[OR_MAXN * OR_MAXN], t2[OR_MAXN * OR_MAXN], t3[OR_MAXN * OR_MAXN];
      or_real r1[OR_MAXN * 3], r2[OR_MAXN * 3], r3[OR_MAXN * 3];

      /* denominator = I - a_above*R  (identity_minus_mat_x_mat :655-690) */
      or_mat_x_mat(n, Aab, Rl, Den);
      for (int k = 0; k < n * n; ++k) Den[k] = -Den[k];
      for (int j = 0; j < n; ++j) E(Den, j, j, n) = 1.0 + E(Den, j, j, n);
      /* a_below = R + T * solve_mat(Den, a_above*T) */
      or_mat_x_mat(n, Aab, Tl, t1);
      or_solve_mat(n, Den, t1, t2);
      or_mat_x_mat(n, Tl, t2, t3);
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) E(Abe, i, j, nb) = E(Rl, i, j, n) + E(t3, i, j, n);
      /* d_below = S_up + T * solve_rect_mat(Den, d_above*E + a_above*S_dn) */
      or_rect_mat_x_mat(n, d, d, Dab, El, r1);
      or_rect_mat_x_mat(n, n, d, Aab, Sdl, r2);
      for (int k = 0; k < n * d; ++k) r1[k] = r1[k] + r2[k];
      or_solve_rect_mat(n, d, Den, r1, r2);
      or_rect_mat_x_mat(n, n, d, Tl, r2, r3);
      for (int j = 0; j < d; ++j)
        for (int i = 0; i < n; ++i) E(Dbe, i, j, nb) = E(Sul, i, j, n) + E(r3, i, j, n);
      if (is_urban) {
        /* exposed roofs (radsurf_urban_sw.F90:626-643) */
        const double *roof_dir = sp->roof_albedo_dir ? sp->roof_albedo_dir : sp->roof_albedo;
        for (int js = 0; js < ns; ++js) {
          for (int js2 = 0; js2 < ns; ++js2)
            E(Abe, n + js, n + js2, nb) = SPEC(sp->roof_albedo, jb, jlay) * lg->hweight[js];
          E(Dbe, n + js, nreg, nb) = cos_sza * SPEC(roof_dir, jb, jlay) * lg->hweight[js];
        }
      }
      /* overlap (:495-503 / :645-653) */
      {
        or_real tA[OR_MAXN * OR_MAXN], tD[OR_MAXN * 3];
        mat_x_expandedmat(db, d, ns, nb, Abe, V, tA);           /* nb x n  */
        expandedmat_x_mat(d, db, ns, n, U, tA, nb, Aab1);       /* n  x n  */
        /* rect_mat_x_singlemat :462-499: d_below (nb x db) * V (db x d) */
        or_rect_mat_x_mat(nb, db, d, Dbe, V, tD);               /* nb x d  */
        expandedmat_x_mat(d, db, ns, d, U, tD, nb, Dab1);       /* n  x d  */
      }
    }

    /* top-of-canopy boundary conditions (:520-528 / :666-674) */
    {
      const or_real *At = w.a_above + (size_t)nlay * n * n;
      const or_real *Dt = w.d_above + (size_t)nlay * n * d;
      or_real y[MAXS], s = 0.0;
      /* mat_x_vec on the ns x ns top-left block, then sum over rows */
      for (int j1 = 0; j1 < ns; ++j1) {
        or_real acc = 0.0;
        for (int j2 = 0; j2 < ns; ++j2) acc += E(At, j1, j2, n) * lg->hweight[j2];
        y[j1] = acc;
      }
      for (int j1 = 0; j1 < ns; ++j1) s += y[j1];
      top_albedo_diff[jb] = s;
      s = 0.0;
      for (int j1 = 0; j1 < ns; ++j1) s += E(Dt, j1, 0, n);
      top_albedo_dir[jb] = s / cos_sza;
    }

    /* ---- Section 5: normalized flux profiles ---- */
#define FC(f, a) ((f)->a[jb + (size_t)nsw * icol])
#define FL(f, a, l) ((f)->a[jb + (size_t)nsw * (l0 + (l))])
    /* ===== fluxes normalized by the direct flux at canopy top (:538-676 / :686-859) ===== */
    {
      or_real dir_above[4], dir_below[4], diff_above[OR_MAXN], diff_below[OR_MAXN];
      or_real up_above[OR_MAXN], up_below[OR_MAXN], refl_dir[OR_MAXN];
      or_real int_flux_dir[3], int_flux_diff[OR_MAXN], v1[OR_MAXN], v2[OR_MAXN], v3[OR_MAXN];
      or_real flux_dn_dir_clear = 1.0 / cos_sza;
      dir_above[0] = 1.0 / cos_sza;
      for (int r = 1; r < nreg; ++r) dir_above[r] = 0.0;
      for (int i = 0; i < n; ++i) { diff_above[i] = 0.0; up_above[i] = 0.0; }
      FC(fdir, top_dn_dir) = 1.0;
      FC(fdir, top_dn) = FC(fdir, top_dn_dir);
      FC(fdir, top_net) = FC(fdir, top_dn_dir) * (1.0 - top_albedo_dir[jb]);
      if (is_urban && jb == 0 && fdir->roof_sunlit_frac) fdir->roof_sunlit_frac[l0 + nlay - 1] = 1.0;

      for (int jlay = nlay - 1; jlay >= 0; --jlay) {
        const or_real *Rl = w.ref_diff + (size_t)jlay * n * n, *Tl = w.trans_diff + (size_t)jlay * n * n;
        const or_real *Sdl = w.trans_dir_diff + (size_t)jlay * n * d;
        const or_real *El = w.trans_dir_dir + (size_t)jlay * d * d;
        const or_real *Idl = w.int_dir + (size_t)jlay * d * d, *Ifl = w.int_diff + (size_t)jlay * n * n;
        const or_real *Idfl = w.int_dir_diff + (size_t)jlay * n * d;
        const or_real *Aab = w.a_above + (size_t)jlay * n * n, *Dab = w.d_above + (size_t)jlay * n * d;
        const or_real *Den = w.denominator + (size_t)jlay * n * n;
        const or_real *Abe = w.a_below + (size_t)(jlay + 1) * nb * nb, *Dbe = w.d_below + (size_t)(jlay + 1) * nb * db;
        const or_real *V = w.v_overlap + (size_t)(jlay + 1) * db * d;
        const or_real *od = w.od_scaling + (size_t)jlay * (MAXREG + 1);
        const or_real *f_wall = w.f_wall + (size_t)jlay * nreg;
        const or_real air_ext = SPEC(sp->air_ext, jb, jlay), air_ssa = SPEC(sp->air_ssa, jb, jlay);

        /* across the interface at the top of the layer */
        or_rect_mat_x_vec(db, d, V, db, dir_above, dir_below);   /* singlemat_x_vec / rect_singlemat_x_vec */
        expandedmat_x_vec(db, d, ns, V, diff_above, diff_below);
        or_mat_x_vec(nb, Abe, diff_below, v1);
        or_rect_mat_x_vec(nb, db, Dbe, nb, dir_below, v2);
        for (int i = 0; i < nb; ++i) up_below[i] = v1[i] + v2[i];

        if (is_urban) {
          or_real s1 = 0.0, s2 = 0.0;
          FL(fdir, roof_in_dir, jlay) = cos_sza * dir_below[nreg];
          for (int i = n; i < nb; ++i) s1 += diff_below[i];
          FL(fdir, roof_in, jlay) = FL(fdir, roof_in_dir, jlay) + s1;
          for (int i = n; i < nb; ++i) s2 += up_below[i];
          FL(fdir, roof_net, jlay) = FL(fdir, roof_in, jlay) - s2;
        }

        /* fluxes at base of layer */
        or_mat_x_vec(d, El, dir_below, dir_above);
        or_rect_mat_x_vec(n, d, Dab, n, dir_above, refl_dir);
        or_mat_x_vec(n, Tl, diff_below, v1);
        or_mat_x_vec(n, Rl, refl_dir, v2);
        or_rect_mat_x_vec(n, d, Sdl, n, dir_below, v3);
        for (int i = 0; i < n; ++i) v1[i] = v1[i] + v2[i] + v3[i];
        or_solve_vec(n, Den, v1, diff_above);
        or_mat_x_vec(n, Aab, diff_above, v1);
        for (int i = 0; i < n; ++i) up_above[i] = v1[i] + refl_dir[i];

        if (fdir->flux_dn_layer_top) {
          or_real s;
          s = 0.0; for (int r = 0; r < nreg; ++r) s += dir_below[r];
          FL(fdir, flux_dn_dir_layer_top, jlay) = cos_sza * s;
          s = 0.0; for (int i = 0; i < n; ++i) s += diff_below[i];
          FL(fdir, flux_dn_layer_top, jlay) = FL(fdir, flux_dn_dir_layer_top, jlay) + s;
          s = 0.0; for (int i = 0; i < n; ++i) s += up_below[i];
          FL(fdir, flux_up_layer_top, jlay) = s;
          s = 0.0; for (int r = 0; r < nreg; ++r) s += dir_above[r];
          FL(fdir, flux_dn_dir_layer_base, jlay) = cos_sza * s;
          s = 0.0; for (int i = 0; i < n; ++i) s += diff_above[i];
          FL(fdir, flux_dn_layer_base, jlay) = FL(fdir, flux_dn_dir_layer_base, jlay) + s;
          s = 0.0; for (int i = 0; i < n; ++i) s += up_above[i];
          FL(fdir, flux_up_layer_base, jlay) = s;
        }

        /* integrated fluxes (:598-606) */
        for (int r = 0; r < nreg; ++r) v1[r] = dir_below[r] - dir_above[r];
        or_mat_x_vec(d, Idl, v1, int_flux_dir);
        for (int i = 0; i < n; ++i) v2[i] = diff_below[i] - diff_above[i] - up_below[i] + up_above[i];
        or_mat_x_vec(n, Ifl, v2, int_flux_diff);
        or_rect_mat_x_vec(n, d, Idfl, n, v1, v3);
        for (int i = 0; i < n; ++i) int_flux_diff[i] = int_flux_diff[i] + v3[i];

        /* absorption (:608-628) */
        {
          or_real s = 0.0;
          for (int js = 0; js < ns; ++js) s += int_flux_diff[js] * (1.0 / lg->mu[js]);
          FL(fdir, clear_air_abs, jlay) = FL(fdir, clear_air_abs, jlay)
                                          + air_ext * (1.0 - air_ssa) * (int_flux_dir[0] + s);
        }
        if (do_vegetation) {
          const or_real vssa = SPEC(sp->veg_ssa, jb, jlay);
          for (int jreg = 1; jreg < nreg; ++jreg) {
            or_real s = 0.0;
            for (int js = 0; js < ns; ++js) s += int_flux_diff[jreg * ns + js] * (1.0 / lg->mu[js]);
            FL(fdir, veg_air_abs, jlay) = FL(fdir, veg_air_abs, jlay)
                                          + air_ext * (1.0 - air_ssa) * (int_flux_dir[jreg] + s);
            FL(fdir, veg_abs_dir, jlay) = FL(fdir, veg_abs_dir, jlay)
                                          + veg_ext[jlay] * (1.0 - vssa) * int_flux_dir[jreg] * od[jreg];
            FL(fdir, veg_abs, jlay) = FL(fdir, veg_abs, jlay)
                                      + veg_ext[jlay] * (1.0 - vssa) * (int_flux_dir[jreg] + s) * od[jreg];
          }
        }

        if (is_urban) {
          /* walls (radsurf_urban_sw.F90:790-802) */
          for (int jreg = 0; jreg < nreg; ++jreg)
            FL(fdir, wall_in_dir, jlay) = FL(fdir, wall_in_dir, jlay) + f_wall[jreg] * sin0 * int_flux_dir[jreg];
          FL(fdir, wall_in, jlay) = FL(fdir, wall_in_dir, jlay);
          for (int jreg = 0; jreg < nreg; ++jreg) {
            or_real s = 0.0;
            for (int js = 0; js < ns; ++js) s += int_flux_diff[jreg * ns + js] * lg->tan_ang[js];
            FL(fdir, wall_in, jlay) = FL(fdir, wall_in, jlay) + f_wall[jreg] * s;
          }
          FL(fdir, wall_net, jlay) = FL(fdir, wall_in, jlay) * (1.0 - SPEC(sp->wall_albedo, jb, jlay));
        }

        /* sunlit fractions: only in the most transparent band */
        if (jb == itransp) {
          const or_real ae = SPEC(sp->air_ext, itransp, jlay);
          or_real trans_dir_clear, int_flux_dir_clear, veg_abs_dir_clear;
          if (is_urban) {
            /* :804-812 */
            fdir->roof_sunlit_frac[l0 + jlay] = FL(fdir, roof_in_dir, jlay) * non_building_fraction[jlay + 1]
                / (cos_sza * flux_dn_dir_clear * OR_FMAX(cfg->min_building_fraction, roof_fraction[jlay]));
            flux_dn_dir_clear = flux_dn_dir_clear * non_building_fraction[jlay] / non_building_fraction[jlay + 1];
          }
          trans_dir_clear = OR_EXP(-ae * dz[jlay] / cos_sza);
          if (ae > 0.0) int_flux_dir_clear = flux_dn_dir_clear * (1.0 - trans_dir_clear) * cos_sza / ae;
          else int_flux_dir_clear = flux_dn_dir_clear * dz[jlay];
          if (do_vegetation) {
            veg_abs_dir_clear = int_flux_dir_clear * veg_ext[jlay]
                                * (1.0 - SPEC(sp->veg_ssa, itransp, jlay)) * veg_fraction[jlay];
            fdir->veg_sunlit_frac[l0 + jlay] = FL(fdir, veg_abs_dir, jlay) / OR_FMAX(DBL_EPSILON, veg_abs_dir_clear);
          }
          if (is_urban)
            fdir->wall_sunlit_frac[l0 + jlay] = 0.5 * FL(fdir, wall_in_dir, jlay)
                / OR_FMAX(DBL_EPSILON, (w.f_wall_dir_clear[jlay] * sin0 * int_flux_dir_clear));
          flux_dn_dir_clear = flux_dn_dir_clear * trans_dir_clear;
        }
      } /* layers downward */

      {
        or_real s = 0.0, s2 = 0.0, s3 = 0.0;
        for (int r = 0; r < nreg; ++r) s += dir_above[r];
        FC(fdir, ground_dn_dir) = cos_sza * s;
        for (int i = 0; i < n; ++i) s2 += diff_above[i];
        FC(fdir, ground_dn) = FC(fdir, ground_dn_dir) + s2;
        for (int i = 0; i < n; ++i) s3 += up_above[i];
        FC(fdir, ground_net) = FC(fdir, ground_dn) - s3;
        for (int jreg = 0; jreg < nreg; ++jreg)
          for (int js = 0; js < ns; ++js) {
            const int ifr = js + jreg * ns;
            FC(fdir, ground_vertical_diff) = FC(fdir, ground_vertical_diff)
                + (diff_above[ifr] + up_above[ifr]) * lg->tan_ang[js] / Pi;
          }
        if (jb == itransp)
          fdir->ground_sunlit_frac[icol] = FC(fdir, ground_dn_dir) / (cos_sza * flux_dn_dir_clear);
      }
    }

    /* ===== fluxes normalized by the diffuse flux at canopy top (:678-760 / :876-984) ===== */
    {
      or_real diff_above[OR_MAXN], diff_below[OR_MAXN], up_above[OR_MAXN], up_below[OR_MAXN];
      or_real int_flux_diff[OR_MAXN], v1[OR_MAXN], v2[OR_MAXN];
      for (int i = 0; i < n; ++i) { diff_above[i] = 0.0; up_above[i] = 0.0; }
      for (int js = 0; js < ns; ++js) diff_above[js] = lg->hweight[js];
      FC(fdiff, top_dn_dir) = 0.0;
      FC(fdiff, top_dn) = 1.0;
      FC(fdiff, top_net) = 1.0 - top_albedo_diff[jb];

      for (int jlay = nlay - 1; jlay >= 0; --jlay) {
        const or_real *Tl = w.trans_diff + (size_t)jlay * n * n;
        const or_real *Ifl = w.int_diff + (size_t)jlay * n * n;
        const or_real *Aab = w.a_above + (size_t)jlay * n * n;
        const or_real *Den = w.denominator + (size_t)jlay * n * n;
        const or_real *Abe = w.a_below + (size_t)(jlay + 1) * nb * nb;
        const or_real *V = w.v_overlap + (size_t)(jlay + 1) * db * d;
        const or_real *od = w.od_scaling + (size_t)jlay * (MAXREG + 1);
        const or_real *f_wall = w.f_wall + (size_t)jlay * nreg;
        const or_real air_ext = SPEC(sp->air_ext, jb, jlay), air_ssa = SPEC(sp->air_ssa, jb, jlay);

        expandedmat_x_vec(db, d, ns, V, diff_above, diff_below);
        or_mat_x_vec(nb, Abe, diff_below, up_below);
        if (is_urban) {
          or_real s1 = 0.0, s2 = 0.0;
          for (int i = n; i < nb; ++i) s1 += diff_below[i];
          FL(fdiff, roof_in, jlay) = +s1;
          for (int i = n; i < nb; ++i) s2 += up_below[i];
          FL(fdiff, roof_net, jlay) = FL(fdiff, roof_in, jlay) - s2;
        }
        or_mat_x_vec(n, Tl, diff_below, v1);
        or_solve_vec(n, Den, v1, diff_above);
        or_mat_x_vec(n, Aab, diff_above, up_above);

        if (fdiff->flux_dn_layer_top) {
          or_real s;
          s = 0.0; for (int i = 0; i < n; ++i) s += diff_below[i];
          FL(fdiff, flux_dn_layer_top, jlay) = s;
          s = 0.0; for (int i = 0; i < n; ++i) s += up_below[i];
          FL(fdiff, flux_up_layer_top, jlay) = s;
          s = 0.0; for (int i = 0; i < n; ++i) s += diff_above[i];
          FL(fdiff, flux_dn_layer_base, jlay) = s;
          s = 0.0; for (int i = 0; i < n; ++i) s += up_above[i];
          FL(fdiff, flux_up_layer_base, jlay) = s;
        }

        for (int i = 0; i < n; ++i) v2[i] = diff_below[i] - diff_above[i] - up_below[i] + up_above[i];
        or_mat_x_vec(n, Ifl, v2, int_flux_diff);

        {
          or_real s = 0.0;
          for (int js = 0; js < ns; ++js) s += int_flux_diff[js] * (1.0 / lg->mu[js]);
          FL(fdiff, clear_air_abs, jlay) = FL(fdiff, clear_air_abs, jlay) + air_ext * (1.0 - air_ssa) * s;
        }
        if (do_vegetation) {
          const or_real vssa = SPEC(sp->veg_ssa, jb, jlay);
          for (int jreg = 1; jreg < nreg; ++jreg) {
            or_real s = 0.0;
            for (int js = 0; js < ns; ++js) s += int_flux_diff[jreg * ns + js] * (1.0 / lg->mu[js]);
            FL(fdiff, veg_air_abs, jlay) = FL(fdiff, veg_air_abs, jlay) + air_ext * (1.0 - air_ssa) * s;
            FL(fdiff, veg_abs, jlay) = FL(fdiff, veg_abs, jlay) + veg_ext[jlay] * (1.0 - vssa) * s * od[jreg];
          }
        }
        if (is_urban) {
          for (int jreg = 0; jreg < nreg; ++jreg) {
            or_real s = 0.0;
            for (int js = 0; js < ns; ++js) s += int_flux_diff[jreg * ns + js] * lg->tan_ang[js];
            FL(fdiff, wall_in, jlay) = FL(fdiff, wall_in, jlay) + f_wall[jreg] * (s);
          }
          FL(fdiff, wall_net, jlay) = FL(fdiff, wall_in, jlay) * (1.0 - SPEC(sp->wall_albedo, jb, jlay));
        }
      }
      {
        or_real s2 = 0.0, s3 = 0.0;
        FC(fdiff, ground_dn_dir) = 0.0;
        for (int i = 0; i < n; ++i) s2 += diff_above[i];
        FC(fdiff, ground_dn) = s2;
        for (int i = 0; i < n; ++i) s3 += up_above[i];
        FC(fdiff, ground_net) = FC(fdiff, ground_dn) - s3;
        for (int jreg = 0; jreg < nreg; ++jreg)
          for (int js = 0; js < ns; ++js) {
            const int ifr = js + jreg * ns;
            FC(fdiff, ground_vertical_diff) = FC(fdiff, ground_vertical_diff)
                + (diff_above[ifr] + up_above[ifr]) * lg->tan_ang[js] / Pi;
          }
      }
    }
  } /* bands */

  work_free(&w);
  (void)ierr;
  return SS_OK;
#undef SPEC
#undef FC
#undef FL
}

/* =======================================================================
 * Longwave: spartacus_forest_lw (radsurf_forest_lw.F90:35-715) and
 * spartacus_urban_lw (radsurf_urban_lw.F90:35-883), one column.
 * ======================================================================= */
static int solve_lw(const ss_config *cfg, int is_urban, int nlw, int ns, int nreg, int nlay,
                    int icol, int l0, const ss_legendre_gauss *lg, const ss_canopy *cp,
                    const ss_lw_spectral *sp, double *top_emissivity, double *top_emission,
                    ss_flux *fint, ss_flux *fnorm) {
  const int n = nreg * ns, d = nreg;
  const int db = is_urban ? nreg + 1 : nreg;
  const int nb = db * ns;
  const int do_vegetation = (nreg > 1);
  or_work_t w;
  int ierr = 0;
  const double *dz = cp->dz + l0;
  const double *veg_fraction = cp->veg_fraction ? cp->veg_fraction + l0 : NULL;
  const double *veg_scale = cp->veg_scale ? cp->veg_scale + l0 : NULL;
  const double *veg_ext = cp->veg_ext ? cp->veg_ext + l0 : NULL;
  const double *veg_fsd = cp->veg_fsd ? cp->veg_fsd + l0 : NULL;
  const double *building_fraction = cp->building_fraction ? cp->building_fraction + l0 : NULL;
  const double *building_scale = cp->building_scale ? cp->building_scale + l0 : NULL;
  const double *veg_contact = cp->veg_contact_fraction ? cp->veg_contact_fraction + l0 : NULL;
#define SPEC(a, b, l) ((a)[(b) + (size_t)nlw * (l0 + (l))])
#define FC(f, a) ((f)->a[jb + (size_t)nlw * icol])
#define FL(f, a, l) ((f)->a[jb + (size_t)nlw * (l0 + (l))])

  if (is_urban && do_vegetation && !cfg->do_vegetation) return SS_ERR_VEG_DISABLED;

  work_alloc(&w, nreg, ns, nlay, is_urban);
  region_fractions(is_urban, 1, nreg, nlay, veg_fraction, building_fraction, w.frac);

  if (is_urban) {
    or_calc_overlap_matrices_urban(nlay, nreg, w.frac, nreg, w.u_overlap, w.v_overlap,
                                   cfg->min_vegetation_fraction);
    or_calc_norm_perim_urban(cfg, nlay, nreg, building_fraction, building_scale, veg_fraction,
                             veg_scale, veg_contact, w.norm_perim, w.norm_perim_wall);
  } else {
    or_calc_overlap_matrices(nlay, nreg, w.frac, nreg, w.u_overlap, w.v_overlap,
                             cfg->min_vegetation_fraction);
    or_calc_norm_perim_forest(cfg, nlay, nreg, veg_fraction, veg_scale, w.norm_perim);
  }

  flux_zero(fint, icol, l0, nlay);
  flux_zero(fnorm, icol, l0, nlay);

  for (int jb = 0; jb < nlw; ++jb) {
    or_real gamma1[OR_MAXN * OR_MAXN], gamma2[OR_MAXN * OR_MAXN], emiss_rate[OR_MAXN];
    or_real f_exchange[9], ext_reg[3], ssa_reg[3], planck_reg[3];
    or_real sub1[OR_MAXN * OR_MAXN], sub2[OR_MAXN * OR_MAXN];
    or_real oR[OR_MAXN * OR_MAXN], oT[OR_MAXN * OR_MAXN], oS[OR_MAXN], oIf[OR_MAXN * OR_MAXN], oIs[OR_MAXN];
    or_real emiss_factor;
    const or_real ground_emissivity = sp->ground_emissivity[jb + (size_t)nlw * icol];
    const or_real ground_emission = sp->ground_emission[jb + (size_t)nlw * icol];

    for (int jlay = 0; jlay < nlay; ++jlay) {
      const or_real *frac = w.frac + (size_t)jlay * nreg;
      const or_real *norm_perim = w.norm_perim + (size_t)jlay * nreg;
      or_real *od = w.od_scaling + (size_t)jlay * (MAXREG + 1);
      or_real *f_wall = w.f_wall + (size_t)jlay * nreg;
      or_real wall_ext = 1.0, wall_factor = 0.0;
      const or_real air_ext = SPEC(sp->air_ext, jb, jlay), air_ssa = SPEC(sp->air_ssa, jb, jlay);
      const or_real vext = do_vegetation ? veg_ext[jlay] : 0.0;
      const or_real vssa = do_vegetation ? SPEC(sp->veg_ssa, jb, jlay) : 0.0;
      const or_real vplanck = do_vegetation ? SPEC(sp->veg_planck, jb, jlay) : 0.0;
      const or_real vaplanck = do_vegetation ? SPEC(sp->veg_air_planck, jb, jlay) : 0.0;

      region_optics(nreg, air_ext, air_ssa, vext, vssa, (nreg == 3) ? veg_fsd[jlay] : 0.0,
                    od, ext_reg, ssa_reg);
      planck_reg[0] = SPEC(sp->clear_air_planck, jb, jlay);
      if (nreg == 2) {
        planck_reg[1] = (ext_reg[0] * (1.0 - ssa_reg[0]) * vaplanck + vext * (1.0 - vssa) * vplanck)
                        / OR_FMAX(ext_reg[1] * (1.0 - ssa_reg[1]), 1.0e-8);
      } else if (nreg == 3) {
        planck_reg[1] = (ext_reg[0] * (1.0 - ssa_reg[0]) * vaplanck + od[1] * vext * (1.0 - vssa) * vplanck)
                        / OR_FMAX(ext_reg[1] * (1.0 - ssa_reg[1]), 1.0e-8);
        planck_reg[2] = (ext_reg[0] * (1.0 - ssa_reg[0]) * vaplanck + od[2] * vext * (1.0 - vssa) * vplanck)
                        / OR_FMAX(ext_reg[2] * (1.0 - ssa_reg[2]), 1.0e-8);
      }
      exchange_rates(nreg, frac, norm_perim, cfg->min_vegetation_fraction, f_exchange);

      for (int r = 0; r < nreg; ++r) f_wall[r] = 0.0;
      if (is_urban) {
        /* radsurf_urban_lw.F90:372-392; wall_factor uses band 1 (Appendix B item 3) */
        const or_real *npw = w.norm_perim_wall + (size_t)jlay * nreg;
        for (int r = 0; r < nreg; ++r) {
          if (frac[r] <= cfg->min_vegetation_fraction) f_wall[r] = 0.0;
          else f_wall[r] = npw[r] * lg->vadjustment2 / (Pi * frac[r]);
        }
        wall_ext = 1.0;
        wall_factor = 1.0 - SPEC(sp->wall_emissivity, 0, jlay);
      }

      for (int k = 0; k < n * n; ++k) { gamma1[k] = 0.0; gamma2[k] = 0.0; }
      for (int jfr = 0; jfr < nreg; ++jfr)
        for (int jto = 0; jto < nreg; ++jto)
          if (jfr != jto)
            for (int js = 0; js < ns; ++js) {
              const int ifr = js + jfr * ns, ito = js + jto * ns;
              E(gamma1, ifr, ifr, n) = E(gamma1, ifr, ifr, n) - lg->tan_ang[js] * E(f_exchange, jto, jfr, nreg);
              E(gamma1, ito, ifr, n) = +lg->tan_ang[js] * E(f_exchange, jto, jfr, nreg);
            }
      for (int jreg = 0; jreg < nreg; ++jreg)
        for (int js = 0; js < ns; ++js) {
          const int ifr = js + jreg * ns;
          if (is_urban)
            E(gamma1, ifr, ifr, n) = E(gamma1, ifr, ifr, n) - ext_reg[jreg] / lg->mu[js]
                                     - lg->tan_ang[js] * f_wall[jreg] * wall_ext;
          else
            E(gamma1, ifr, ifr, n) = E(gamma1, ifr, ifr, n) - ext_reg[jreg] / lg->mu[js];
        }
      for (int jsfr = 0; jsfr < ns; ++jsfr)
        for (int jsto = 0; jsto < ns; ++jsto)
          for (int jreg = 0; jreg < nreg; ++jreg) {
            const int ifr = jsfr + jreg * ns, ito = jsto + jreg * ns;
            if (is_urban)
              E(gamma2, ito, ifr, n) = 0.5 * (lg->weight[jsto] * ext_reg[jreg] * ssa_reg[jreg] / lg->mu[jsfr]
                                              + lg->vweight[jsto] * lg->tan_ang[jsfr] * f_wall[jreg] * wall_factor);
            else
              E(gamma2, ito, ifr, n) = (0.5 * lg->weight[jsto] / lg->mu[jsfr]) * ext_reg[jreg] * ssa_reg[jreg];
          }
      for (int k = 0; k < n * n; ++k) gamma1[k] = gamma1[k] + gamma2[k];

      /* emission rates (radsurf_forest_lw.F90:377-395 / radsurf_urban_lw.F90:441-476) */
      emiss_factor = 0.0;
      for (int js = 0; js < ns; ++js) emiss_factor += lg->hweight[js] / lg->mu[js];
      emiss_factor = 2.0 * emiss_factor;
      for (int jreg = 0; jreg < nreg; ++jreg) {
        const or_real volume_emiss = frac[jreg] * (ext_reg[jreg] * (1.0 - ssa_reg[jreg]) * planck_reg[jreg]);
        or_real wall_emiss = 0.0;
        if (is_urban)
          wall_emiss = w.norm_perim_wall[(size_t)jlay * nreg + jreg] * lg->vadjustment
                       * SPEC(sp->wall_emission, jb, jlay);
        for (int js = 0; js < ns; ++js) {
          const int ifr = js + jreg * ns;
          if (is_urban)
            emiss_rate[ifr] = (lg->hweight[js] / lg->mu[js]) * volume_emiss + (0.5 * lg->vweight[js]) * wall_emiss;
          else
            emiss_rate[ifr] = (lg->hweight[js] / lg->mu[js]) * volume_emiss;
        }
        w.emiss_reg[(size_t)jlay * nreg + jreg] = emiss_factor * volume_emiss;
        if (jreg > 0) {
          w.emiss_air[(size_t)jlay * (MAXREG + 1) + jreg] = emiss_factor * frac[jreg] * ext_reg[0]
                                                           * (1.0 - ssa_reg[0]) * vaplanck;
          w.emiss_veg[(size_t)jlay * (MAXREG + 1) + jreg] = emiss_factor * frac[jreg] * vext
                                                           * (1.0 - vssa) * vplanck * od[jreg];
        }
      }
      if (is_urban) {
        or_real s = 0.0;
        for (int r = 0; r < nreg; ++r) s += w.norm_perim_wall[(size_t)jlay * nreg + r];
        w.emiss_wall[jlay] = (s * lg->vadjustment) * SPEC(sp->wall_emission, jb, jlay);
      }

      /* layer matrices, 3-way branch */
      {
        or_real *Rl = w.ref_diff + (size_t)jlay * n * n, *Tl = w.trans_diff + (size_t)jlay * n * n;
        or_real *Sl = w.source_lay + (size_t)jlay * n, *Ifl = w.int_diff + (size_t)jlay * n * n;
        or_real *Isl = w.int_source + (size_t)jlay * n;
        int r0, nr;
        if (!do_vegetation || veg_fraction[jlay] <= cfg->min_vegetation_fraction) {
          r0 = 0; nr = 1;
        } else if (frac[0] <= cfg->min_vegetation_fraction) {
          r0 = 1; nr = nreg - 1;
        } else {
          r0 = 0; nr = nreg;
        }
        for (int k = 0; k < n * n; ++k) { Rl[k] = 0.0; Tl[k] = 0.0; Ifl[k] = 0.0; }
        for (int k = 0; k < n; ++k) { Sl[k] = 0.0; Isl[k] = 0.0; }
        {
          const int nsub = nr * ns, i0 = r0 * ns;
          get_block(gamma1, n, i0, i0, sub1, nsub, nsub);
          get_block(gamma2, n, i0, i0, sub2, nsub, nsub);
          ierr += or_calc_matrices_lw_eig(nsub, dz[jlay], sub1, sub2, emiss_rate + i0, oR, oT, oS, oIf, oIs);
          put_block(Rl, n, i0, i0, oR, nsub, nsub);
          put_block(Tl, n, i0, i0, oT, nsub, nsub);
          put_block(Ifl, n, i0, i0, oIf, nsub, nsub);
          for (int i = 0; i < nsub; ++i) { Sl[i0 + i] = oS[i]; Isl[i0 + i] = oIs[i]; }
        }
      }
    } /* layers */

    /* Section 4: adding (radsurf_forest_lw.F90:452-494 / radsurf_urban_lw.F90:551-614) */
    for (size_t k = 0; k < (size_t)n * n * (nlay + 1); ++k) w.a_above[k] = 0.0;
    for (size_t k = 0; k < (size_t)n * (nlay + 1); ++k) w.source_above[k] = 0.0;
    for (size_t k = 0; k < (size_t)nb * nb * (nlay + 1); ++k) w.a_below[k] = 0.0;
    for (size_t k = 0; k < (size_t)nb * (nlay + 1); ++k) w.source_below[k] = 0.0;
    for (int jreg = 0; jreg < nreg; ++jreg) {
      for (int jsto = 0; jsto < ns; ++jsto)
        for (int jsfr = 0; jsfr < ns; ++jsfr)
          E(w.a_above, jsto + jreg * ns, jsfr + jreg * ns, n) = (1.0 - ground_emissivity) * lg->hweight[jsto];
      for (int js = 0; js < ns; ++js)
        w.source_above[js + jreg * ns] = (lg->hweight[js] * w.frac[jreg]) * ground_emission;
    }
    for (int jlay = 0; jlay < nlay; ++jlay) {
      const or_real *Rl = w.ref_diff + (size_t)jlay * n * n, *Tl = w.trans_diff + (size_t)jlay * n * n;
      const or_real *Sl = w.source_lay + (size_t)jlay * n;
      const or_real *Aab = w.a_above + (size_t)jlay * n * n, *Sab = w.source_above + (size_t)jlay * n;
      or_real *Den = w.denominator + (size_t)jlay * n * n;
      or_real *Abe = w.a_below + (size_t)(jlay + 1) * nb * nb, *Sbe = w.source_below + (size_t)(jlay + 1) * nb;
      or_real *Aab1 = w.a_above + (size_t)(jlay + 1) * n * n, *Sab1 = w.source_above + (size_t)(jlay + 1) * n;
      const or_real *U = w.u_overlap + (size_t)(jlay + 1) * d * db;
      const or_real *V = w.v_overlap + (size_t)(jlay + 1) * db * d;
      or_real t1[OR_MAXN * OR_MAXN], t2[OR_MAXN * OR_MAXN], t3[OR_MAXN * OR_MAXN];
      or_real v1[OR_MAXN], v2[OR_MAXN];

      or_mat_x_mat(n, Aab, Rl, Den);
      for (int k = 0; k < n * n; ++k) Den[k] = -Den[k];
      for (int j = 0; j < n; ++j) E(Den, j, j, n) = 1.0 + E(Den, j, j, n);
      or_mat_x_mat(n, Aab, Tl, t1);
      or_solve_mat(n, Den, t1, t2);
      or_mat_x_mat(n, Tl, t2, t3);
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) E(Abe, i, j, nb) = E(Rl, i, j, n) + E(t3, i, j, n);
      /* Eq. 34: source_below = source_lay + T*solve_vec(Den, source_above + a_above*source_lay) */
      or_mat_x_vec(n, Aab, Sl, v1);
      for (int i = 0; i < n; ++i) v1[i] = Sab[i] + v1[i];
      or_solve_vec(n, Den, v1, v2);
      or_mat_x_vec(n, Tl, v2, v1);
      for (int i = 0; i < n; ++i) Sbe[i] = Sl[i] + v1[i];
      if (is_urban) {
        or_real exposed_roof_frac;
        if (jlay < nlay - 1) exposed_roof_frac = OR_FMAX(0.0, building_fraction[jlay] - building_fraction[jlay + 1]);
        else exposed_roof_frac = building_fraction[jlay];
        for (int js = 0; js < ns; ++js) {
          for (int js2 = 0; js2 < ns; ++js2)
            E(Abe, n + js, n + js2, nb) = (1.0 - SPEC(sp->roof_emissivity, jb, jlay)) * lg->hweight[js];
          Sbe[n + js] = lg->hweight[js] * SPEC(sp->roof_emission, jb, jlay) * exposed_roof_frac;
        }
      }
      {
        or_real tA[OR_MAXN * OR_MAXN];
        mat_x_expandedmat(db, d, ns, nb, Abe, V, tA);
        expandedmat_x_mat(d, db, ns, n, U, tA, nb, Aab1);
        expandedmat_x_vec(d, db, ns, U, Sbe, Sab1);
      }
    }
    {
      const or_real *At = w.a_above + (size_t)nlay * n * n;
      const or_real *St = w.source_above + (size_t)nlay * n;
      or_real y[MAXS], s = 0.0;
      for (int j1 = 0; j1 < ns; ++j1) {
        or_real acc = 0.0;
        for (int j2 = 0; j2 < ns; ++j2) acc += E(At, j1, j2, n) * lg->hweight[j2];
        y[j1] = acc;
      }
      for (int j1 = 0; j1 < ns; ++j1) s += y[j1];
      top_emissivity[jb] = 1.0 - s;
      s = 0.0;
      for (int j1 = 0; j1 < ns; ++j1) s += St[j1];
      top_emission[jb] = s;
    }

    /* Section 5: two downward passes: kind 0 = internal emission, 1 = incoming */
    for (int kind = 0; kind < 2; ++kind) {
      ss_flux *f = kind == 0 ? fint : fnorm;
      or_real dn_above[OR_MAXN], dn_below[OR_MAXN], up_above[OR_MAXN], up_below[OR_MAXN];
      or_real int_flux[OR_MAXN], v1[OR_MAXN], v2[OR_MAXN];
      for (int i = 0; i < n; ++i) { dn_above[i] = 0.0; up_above[i] = 0.0; }
      if (kind == 0) {
        FC(f, top_dn) = 0.0;
        FC(f, top_net) = -top_emission[jb];
      } else {
        for (int js = 0; js < ns; ++js) dn_above[js] = lg->hweight[js];
        FC(f, top_dn) = 1.0;
        FC(f, top_net) = top_emissivity[jb];
      }
      for (int jlay = nlay - 1; jlay >= 0; --jlay) {
        const or_real *Rl = w.ref_diff + (size_t)jlay * n * n, *Tl = w.trans_diff + (size_t)jlay * n * n;
        const or_real *Sl = w.source_lay + (size_t)jlay * n, *Ifl = w.int_diff + (size_t)jlay * n * n;
        const or_real *Isl = w.int_source + (size_t)jlay * n;
        const or_real *Aab = w.a_above + (size_t)jlay * n * n, *Sab = w.source_above + (size_t)jlay * n;
        const or_real *Den = w.denominator + (size_t)jlay * n * n;
        const or_real *Abe = w.a_below + (size_t)(jlay + 1) * nb * nb, *Sbe = w.source_below + (size_t)(jlay + 1) * nb;
        const or_real *V = w.v_overlap + (size_t)(jlay + 1) * db * d;
        const or_real *od = w.od_scaling + (size_t)jlay * (MAXREG + 1);
        const or_real *f_wall = w.f_wall + (size_t)jlay * nreg;
        const or_real air_ext = SPEC(sp->air_ext, jb, jlay), air_ssa = SPEC(sp->air_ssa, jb, jlay);

        expandedmat_x_vec(db, d, ns, V, dn_above, dn_below);
        or_mat_x_vec(nb, Abe, dn_below, up_below);
        if (kind == 0) for (int i = 0; i < nb; ++i) up_below[i] = up_below[i] + Sbe[i];
        if (is_urban) {
          or_real s1 = 0.0, s2 = 0.0;
          for (int i = n; i < nb; ++i) s1 += dn_below[i];
          FL(f, roof_in, jlay) = s1;
          for (int i = n; i < nb; ++i) s2 += up_below[i];
          FL(f, roof_net, jlay) = FL(f, roof_in, jlay) - s2;
        }
        or_mat_x_vec(n, Tl, dn_below, v1);
        if (kind == 0) {
          or_mat_x_vec(n, Rl, Sab, v2);
          for (int i = 0; i < n; ++i) v1[i] = (v1[i] + v2[i] + Sl[i]);
        }
        or_solve_vec(n, Den, v1, dn_above);
        or_mat_x_vec(n, Aab, dn_above, up_above);
        if (kind == 0) for (int i = 0; i < n; ++i) up_above[i] = up_above[i] + Sab[i];

        if (f->flux_dn_layer_top) {
          or_real s;
          s = 0.0; for (int i = 0; i < n; ++i) s += dn_below[i];
          FL(f, flux_dn_layer_top, jlay) = s;
          s = 0.0; for (int i = 0; i < n; ++i) s += up_below[i];
          FL(f, flux_up_layer_top, jlay) = s;
          s = 0.0; for (int i = 0; i < n; ++i) s += dn_above[i];
          FL(f, flux_dn_layer_base, jlay) = s;
          s = 0.0; for (int i = 0; i < n; ++i) s += up_above[i];
          FL(f, flux_up_layer_base, jlay) = s;
        }

        for (int i = 0; i < n; ++i) v2[i] = dn_below[i] + up_above[i];
        or_mat_x_vec(n, Ifl, v2, int_flux);
        if (kind == 0) for (int i = 0; i < n; ++i) int_flux[i] = int_flux[i] + Isl[i];

        {
          or_real s = 0.0;
          for (int js = 0; js < ns; ++js) s += int_flux[js] * (1.0 / lg->mu[js]);
          if (kind == 0)
            FL(f, clear_air_abs, jlay) = FL(f, clear_air_abs, jlay) + air_ext * (1.0 - air_ssa) * s
                                         - w.emiss_reg[(size_t)jlay * nreg] * dz[jlay];
          else
            FL(f, clear_air_abs, jlay) = FL(f, clear_air_abs, jlay) + air_ext * (1.0 - air_ssa) * s;
        }
        if (do_vegetation) {
          const or_real vssa = SPEC(sp->veg_ssa, jb, jlay);
          for (int jreg = 1; jreg < nreg; ++jreg) {
            or_real s = 0.0;
            for (int js = 0; js < ns; ++js) s += int_flux[jreg * ns + js] * (1.0 / lg->mu[js]);
            if (kind == 0) {
              FL(f, veg_air_abs, jlay) = FL(f, veg_air_abs, jlay) + air_ext * (1.0 - air_ssa) * s
                                         - w.emiss_air[(size_t)jlay * (MAXREG + 1) + jreg] * dz[jlay];
              FL(f, veg_abs, jlay) = FL(f, veg_abs, jlay) + veg_ext[jlay] * (1.0 - vssa) * s * od[jreg]
                                     - w.emiss_veg[(size_t)jlay * (MAXREG + 1) + jreg] * dz[jlay];
            } else {
              FL(f, veg_air_abs, jlay) = FL(f, veg_air_abs, jlay) + air_ext * (1.0 - air_ssa) * s;
              FL(f, veg_abs, jlay) = FL(f, veg_abs, jlay) + veg_ext[jlay] * (1.0 - vssa) * s * od[jreg];
            }
          }
        }
        if (is_urban) {
          for (int jreg = 0; jreg < nreg; ++jreg) {
            or_real s = 0.0;
            for (int js = 0; js < ns; ++js) s += int_flux[jreg * ns + js] * lg->tan_ang[js];
            FL(f, wall_in, jlay) = FL(f, wall_in, jlay) + f_wall[jreg] * s;
          }
          if (kind == 0)
            FL(f, wall_net, jlay) = FL(f, wall_in, jlay) * SPEC(sp->wall_emissivity, jb, jlay)
                                    - w.emiss_wall[jlay] * dz[jlay];
          else
            FL(f, wall_net, jlay) = FL(f, wall_in, jlay) * SPEC(sp->wall_emissivity, jb, jlay);
        }
      }
      {
        or_real s2 = 0.0, s3 = 0.0;
        /* forest LW accumulates BOTH passes into lw_internal%ground_vertical_diff
         * (radsurf_forest_lw.F90:691); urban uses lw_norm */
        ss_flux *fv = (kind == 1 && !is_urban) ? fint : f;
        for (int i = 0; i < n; ++i) s2 += dn_above[i];
        FC(f, ground_dn) = s2;
        for (int i = 0; i < n; ++i) s3 += up_above[i];
        FC(f, ground_net) = FC(f, ground_dn) - s3;
        for (int jreg = 0; jreg < nreg; ++jreg)
          for (int js = 0; js < ns; ++js) {
            const int ifr = js + jreg * ns;
            FC(fv, ground_vertical_diff) = FC(fv, ground_vertical_diff)
                + (dn_above[ifr] + up_above[ifr]) * lg->tan_ang[js] / Pi;
          }
      }
    }
  } /* bands */

  work_free(&w);
  (void)ierr;
  return SS_OK;
#undef SPEC
#undef FC
#undef FL
}

/* calc_simple_spectrum_lw (radsurf/radsurf_simple_spectrum.F90:41-66) */
static int simple_spectrum_lw(const ss_config *cfg, const ss_canopy *cp, const ss_lw_spectral *lw,
                              int istartcol, int iendcol) {
  int ilay1 = cp->istartlay[istartcol - 1];
  int ilay2 = cp->istartlay[iendcol - 1] + cp->nlay[iendcol - 1] - 1;
  if (lw->nspec > 1) return SS_ERR_BAD_ARGUMENT;
  for (int c = istartcol - 1; c < iendcol; ++c) {
    or_real t = cp->ground_temperature[c];
    lw->ground_emission[c] = StefanBoltzmann * lw->ground_emissivity[c] * (t * t * t * t);
  }
  for (int l = ilay1 - 1; l < ilay2; ++l) {
    or_real t;
    if (lw->roof_emission) {
      t = cp->roof_temperature[l];
      lw->roof_emission[l] = StefanBoltzmann * lw->roof_emissivity[l] * (t * t * t * t);
      t = cp->wall_temperature[l];
      lw->wall_emission[l] = StefanBoltzmann * lw->wall_emissivity[l] * (t * t * t * t);
    }
    t = cp->clear_air_temperature[l];
    lw->clear_air_planck[l] = StefanBoltzmann * (t * t * t * t);
    if (cfg->do_vegetation) {
      t = cp->veg_temperature[l];
      lw->veg_planck[l] = StefanBoltzmann * (t * t * t * t);
      t = cp->veg_air_temperature[l];
      lw->veg_air_planck[l] = StefanBoltzmann * (t * t * t * t);
    }
  }
  return SS_OK;
}

/* ==========================================================================
 * Single-layer "simple urban" tiles (types 4, 5): Harman et al. 2x2 problem
 * with infinite-street or exponential view factors.
 * ========================================================================== */

/* calc_view_factors_inf (radsurf/radsurf_view_factor.F90:28-69); cos_sza <= 0
 * means "direct part not requested" */
static void view_factors_inf(or_real hw, or_real *vgs, or_real *vww, or_real cos_sza, or_real *vdg) {
  *vgs = OR_SQRT(hw * hw + 1.0) - hw;
  *vww = OR_SQRT(1.0 / (hw * hw) + 1.0) - 1.0 / hw;
  if (vdg) {
    or_real norm_x0 = (Pi * 0.5) * hw * OR_SQRT(1.0 / (cos_sza * cos_sza) - 1.0);
    or_real y_over_w = OR_SQRT(OR_FMAX(norm_x0 * norm_x0 - 1.0, 0.0));
    if (y_over_w > 0.0) *vdg = (2.0 / Pi) * (y_over_w - norm_x0 + OR_ATAN(1.0 / y_over_w));
    else *vdg = 1.0 - 2.0 * norm_x0 / Pi;
  }
}

/* calc_view_factors_exp (radsurf_view_factor.F90:75-136) */
static void view_factors_exp(or_real hx, or_real *vgs, or_real *vww, or_real cos_sza, or_real *vdg) {
  static const double weights[8] = {0.0506142681451884, 0.111190517226687, 0.156853322938944,
                                    0.181341891689181,  0.181341891689181, 0.156853322938944,
                                    0.111190517226687,  0.0506142681451884};
  static const double nodes[8] = {0.0198550717512319, 0.101666761293187, 0.237233795041836,
                                  0.408282678752175,  0.591717321247825, 0.762766204958164,
                                  0.898333238706813,  0.980144928248768};
  or_real hweight[8], vweight[8], sh = 0.0, sv = 0.0, a = 0.0, b = 0.0;
  for (int k = 0; k < 8; ++k) sh += (or_real)weights[k] * nodes[k];
  for (int k = 0; k < 8; ++k) hweight[k] = (or_real)weights[k] * nodes[k] / sh;
  for (int k = 0; k < 8; ++k) vweight[k] = weights[k] * OR_SQRT(1.0 - (or_real)nodes[k] * nodes[k]);
  for (int k = 0; k < 8; ++k) sv += vweight[k];
  for (int k = 0; k < 8; ++k) vweight[k] = vweight[k] / sv;
  for (int k = 0; k < 8; ++k) {
    or_real tk = hx * OR_SQRT(1.0 / ((or_real)nodes[k] * nodes[k]) - 1.0);
    or_real e = OR_EXP(-tk);
    a += hweight[k] * e;
    b += vweight[k] * (1.0 - e) / tk;
  }
  *vgs = a;
  *vww = 1.0 - b;
  if (vdg) *vdg = OR_EXP(-(hx * OR_SQRT(1.0 / (cos_sza * cos_sza) - 1.0)));
}

/* Geometry shared by the SW and LW variants
 * (radsurf_simple_urban_sw.F90:124-152, _lw.F90:118-143) */
typedef struct {
  or_real vgs, vww, vwg, vgw, vdg, npw, bf, dz;
} simple_geom;

static void simple_urban_geometry(const ss_config *cfg, int infinite, const ss_canopy *cp, int ilay,
                                  or_real cos_sza, int want_dir, simple_geom *g) {
  const double zero = 0.0, one = 1.0;
  or_real np[1], npw[1];
  g->dz = cp->dz[ilay];
  g->bf = cp->building_fraction[ilay];
  or_calc_norm_perim_urban(cfg, 1, 1, cp->building_fraction + ilay, cp->building_scale + ilay,
                           &zero, &one, &zero, np, npw);
  g->npw = npw[0];
  g->vdg = 0.0;
  if (infinite) {
    or_real street_width = 2.0 * (1.0 - g->bf) / g->npw;
    view_factors_inf(g->dz / street_width, &g->vgs, &g->vww, cos_sza, want_dir ? &g->vdg : NULL);
  } else {
    or_real sep = Pi * (1.0 - g->bf) / g->npw;
    view_factors_exp(g->dz / sep, &g->vgs, &g->vww, cos_sza, want_dir ? &g->vdg : NULL);
  }
  g->vwg = 0.5 * (1.0 - g->vww);
  g->vgw = 1.0 - g->vgs;
}

/* simple_urban_sw (radsurf_simple_urban_sw.F90:28-294).  The reference indexes
 * several (nspec,ncol) members with the LAYER index (e.g. :200, :255-257,
 * :263-275); that is the same element whenever istartlay(icol) == icol - every
 * column before it has exactly one layer - which is the only arrangement its
 * own test (test/single_layer) uses.  This restatement uses the column index
 * throughout.  top_albedo_diff/dir are never assigned by the reference
 * (bc_out keeps its previous contents), so they are not touched here. */
static void simple_urban_sw(const ss_config *cfg, int infinite, int nsw, int icol, int ilay,
                            or_real cos_sza, const ss_canopy *cp, const ss_sw_spectral *sp,
                            const double *galb, const double *galb_dir, ss_flux *fdir, ss_flux *fdiff) {
  simple_geom g;
  simple_urban_geometry(cfg, infinite, cp, ilay, cos_sza, 1, &g);
  const or_real vdw = 1.0 - g.vdg, open = 1.0 - g.bf;
  flux_zero(fdiff, icol, ilay, 1);
  flux_zero(fdir, icol, ilay, 1);
  for (int jb = 0; jb < nsw; ++jb) {
    const size_t c = jb + (size_t)nsw * icol, l = jb + (size_t)nsw * ilay;
    const or_real wall_albedo = sp->wall_albedo[l], roof_albedo = sp->roof_albedo[l];
    const or_real ga = galb[jb], gad = galb_dir[jb];
    or_real A[4], src[2], x[2], up;
    E(A, 0, 0, 2) = 1.0;
    E(A, 0, 1, 2) = -g.vwg * wall_albedo;
    E(A, 1, 0, 2) = -g.vgw * ga;
    E(A, 1, 1, 2) = 1.0 - g.vww * wall_albedo;
    /* direct */
    src[0] = 0.0;
    src[1] = (vdw + gad * g.vdg * g.vgw) * open;
    or_solve_vec(2, A, src, x);
    {
      const or_real gdd = g.vdg * open, gdn = gdd + x[0];
      const or_real gnet = gdd * (1.0 - gad) + x[0] * (1.0 - ga);
      const or_real wnet = x[1] * (1.0 - wall_albedo);
      fdir->ground_dn_dir[c] = gdd;
      fdir->ground_dn[c] = gdn;
      fdir->ground_net[c] = gnet;
      fdir->roof_in_dir[l] = g.bf;
      fdir->roof_in[l] = g.bf;
      fdir->roof_net[l] = g.bf * (1.0 - roof_albedo);
      fdir->wall_in_dir[l] = vdw * open;
      fdir->wall_in[l] = x[1];
      fdir->wall_net[l] = wnet;
      fdir->top_dn_dir[c] = 1.0;
      fdir->top_dn[c] = 1.0;
      up = (gdn - gnet) * g.vgs + (x[1] - wnet) * g.vwg;
      fdir->top_net[c] = 1.0 - g.bf * roof_albedo - (gdn - gnet) * g.vgs - (x[1] - wnet) * g.vwg;
      if (fdir->flux_dn_layer_top) {
        fdir->flux_dn_dir_layer_top[l] = open;
        fdir->flux_dn_layer_top[l] = open;
        fdir->flux_up_layer_top[l] = up;
        fdir->flux_dn_dir_layer_base[l] = gdd;
        fdir->flux_dn_layer_base[l] = gdn;
        fdir->flux_up_layer_base[l] = gdn - gnet;
      }
    }
    /* diffuse */
    src[0] = g.vgs * open;
    src[1] = g.vgw * open;
    or_solve_vec(2, A, src, x);
    {
      const or_real gdn = x[0], gnet = gdn * (1.0 - ga), wnet = x[1] * (1.0 - wall_albedo);
      fdiff->ground_dn_dir[c] = 0.0;
      fdiff->ground_dn[c] = gdn;
      fdiff->ground_net[c] = gnet;
      fdiff->roof_in[l] = g.bf;
      fdiff->roof_net[l] = g.bf * (1.0 - roof_albedo);
      fdiff->wall_in[l] = x[1];
      fdiff->wall_net[l] = wnet;
      fdiff->top_dn_dir[c] = 0.0;
      fdiff->top_dn[c] = 1.0;
      up = (gdn - gnet) * g.vgs + (x[1] - wnet) * g.vwg;
      fdiff->top_net[c] = 1.0 - g.bf * roof_albedo - (gdn - gnet) * g.vgs - (x[1] - wnet) * g.vwg;
      if (fdiff->flux_dn_layer_top) {
        fdiff->flux_dn_layer_top[l] = open;
        fdiff->flux_up_layer_top[l] = up;
        fdiff->flux_dn_layer_base[l] = gdn;
        fdiff->flux_up_layer_base[l] = gdn - gnet;
      }
    }
  }
  fdir->ground_sunlit_frac[icol] = g.vdg;
  fdir->roof_sunlit_frac[ilay] = 1.0;
  {
    const or_real tan_sza = OR_SQRT(1.0 / (cos_sza * cos_sza) - 1.0);
    fdir->wall_sunlit_frac[ilay] = 0.5 * vdw / (OR_FMAX(tan_sza, 1.0e-6) * g.npw * g.dz / (Pi * open));
  }
}

/* simple_urban_lw (radsurf_simple_urban_lw.F90:28-257); same remark on the
 * column index; top_emissivity / top_emission are never assigned by the
 * reference.  interaction_matrix(2,2) uses the GROUND emissivity (:157), as the
 * reference does. */
static void simple_urban_lw(const ss_config *cfg, int infinite, int nlw, int icol, int ilay,
                            const ss_canopy *cp, const ss_lw_spectral *sp, ss_flux *fint, ss_flux *fnorm) {
  simple_geom g;
  simple_urban_geometry(cfg, infinite, cp, ilay, 1.0, 0, &g);
  const or_real open = 1.0 - g.bf;
  flux_zero(fnorm, icol, ilay, 1);
  flux_zero(fint, icol, ilay, 1);
  for (int jb = 0; jb < nlw; ++jb) {
    const size_t c = jb + (size_t)nlw * icol, l = jb + (size_t)nlw * ilay;
    const or_real gem = sp->ground_emissivity[c], gemission = sp->ground_emission[c];
    const or_real rem = sp->roof_emissivity[l], remission = sp->roof_emission[l];
    const or_real wem = sp->wall_emissivity[l], wemission = sp->wall_emission[l];
    or_real A[4], src[2], x[2], up;
    E(A, 0, 0, 2) = 1.0;
    E(A, 0, 1, 2) = -g.vwg * (1.0 - wem);
    E(A, 1, 0, 2) = -g.vgw * (1.0 - gem);
    E(A, 1, 1, 2) = 1.0 - g.vww * (1.0 - gem);
    /* internal emission */
    src[0] = g.vwg * wemission * g.npw * g.dz;
    src[1] = g.vgw * gemission * open + g.vww * wemission * g.npw * g.dz;
    or_solve_vec(2, A, src, x);
    {
      const or_real gnet = x[0] * gem - gemission * open;
      const or_real wnet = x[1] * wem - wemission * g.npw * g.dz;
      fint->ground_dn[c] = x[0];
      fint->ground_net[c] = gnet;
      fint->roof_in[l] = 0.0;
      fint->roof_net[l] = -g.bf * remission;
      fint->wall_in[l] = x[1];
      fint->wall_net[l] = wnet;
      fint->top_dn[c] = 0.0;
      up = (x[0] - gnet) * g.vgs + (x[1] - wnet) * g.vwg;
      fint->top_net[c] = -g.bf * remission - (x[0] - gnet) * g.vgs - (x[1] - wnet) * g.vwg;
      if (fint->flux_dn_layer_top) {
        fint->flux_dn_layer_top[l] = 0.0;
        fint->flux_up_layer_top[l] = up;
        fint->flux_dn_layer_base[l] = x[0];
        fint->flux_up_layer_base[l] = x[0] - gnet;
      }
    }
    /* incoming diffuse */
    src[0] = g.vgs * open;
    src[1] = g.vgw * open;
    or_solve_vec(2, A, src, x);
    {
      const or_real gnet = x[0] * gem, wnet = x[1] * wem;
      fnorm->ground_dn[c] = x[0];
      fnorm->ground_net[c] = gnet;
      fnorm->roof_in[l] = g.bf;
      fnorm->roof_net[l] = g.bf * rem;
      fnorm->wall_in[l] = x[1];
      fnorm->wall_net[l] = wnet;
      fnorm->top_dn[c] = 1.0;
      up = (x[0] - gnet) * g.vgs + (x[1] - wnet) * g.vwg;
      fnorm->top_net[c] = 1.0 - g.bf * (1.0 - rem) - (x[0] - gnet) * g.vgs - (x[1] - wnet) * g.vwg;
      if (fnorm->flux_dn_layer_top) {
        fnorm->flux_dn_layer_top[l] = 1.0 - g.bf;
        fnorm->flux_up_layer_top[l] = up;
        fnorm->flux_dn_layer_base[l] = x[0];
        fnorm->flux_up_layer_base[l] = x[0] - gnet;
      }
    }
  }
}

/* One column of radsurf (radsurf/radsurf_interface.F90:105-311); jcol 0-based */
static int radsurf_column(const ss_config *cfg, const ss_canopy *cp, const ss_sw_spectral *sw,
                          const ss_lw_spectral *lw, ss_bc_out *bc, int jcol, ss_flux *sw_dir,
                          ss_flux *sw_diff, ss_flux *lw_int, ss_flux *lw_norm) {
  const int irep = cp->i_representation[jcol];
  const int nsw = cfg->nsw, nlw = cfg->nlw;
  const int nlay = cp->nlay[jcol];
  const int l0 = (irep == SS_TILE_FLAT) ? 0 : cp->istartlay[jcol] - 1;
  const double *galb_dir = NULL;
  int rc = SS_OK;
  if (cfg->do_sw) galb_dir = cfg->use_sw_direct_albedo ? sw->ground_albedo_dir : sw->ground_albedo;
#define C(a, b) ((a)[(b) + (size_t)nsw * jcol])
#define CL(a, b) ((a)[(b) + (size_t)nlw * jcol])

  switch (irep) {
  case SS_TILE_FLAT:
    if (cfg->do_sw)
      for (int b = 0; b < nsw; ++b) {
        C(bc->sw_albedo, b) = C(sw->ground_albedo, b);
        C(bc->sw_albedo_dir, b) = C(galb_dir, b);
        C(sw_dir->ground_dn_dir, b) = 1.0;
        C(sw_dir->ground_dn, b) = 1.0;
        C(sw_dir->ground_net, b) = 1.0 - C(galb_dir, b);
        C(sw_dir->ground_vertical_diff, b) = 0.5 * C(galb_dir, b);
        C(sw_dir->top_dn_dir, b) = 1.0;
        C(sw_dir->top_dn, b) = 1.0;
        C(sw_dir->top_net, b) = 1.0 - C(galb_dir, b);
        C(sw_diff->ground_dn_dir, b) = 0.0;
        C(sw_diff->ground_dn, b) = 1.0;
        C(sw_diff->ground_net, b) = 1.0 - C(sw->ground_albedo, b);
        C(sw_diff->ground_vertical_diff, b) = 0.5 * (1.0 + C(sw->ground_albedo, b));
        C(sw_diff->top_dn_dir, b) = 0.0;
        C(sw_diff->top_dn, b) = 1.0;
        C(sw_diff->top_net, b) = 1.0 - C(sw->ground_albedo, b);
      }
    if (cfg->do_lw)
      for (int b = 0; b < nlw; ++b) {
        CL(bc->lw_emissivity, b) = CL(lw->ground_emissivity, b);
        CL(bc->lw_emission, b) = CL(lw->ground_emission, b);
        CL(lw_int->ground_dn, b) = 0.0;
        CL(lw_int->ground_net, b) = -CL(lw->ground_emission, b);
        CL(lw_int->ground_vertical_diff, b) = 0.5 * CL(lw->ground_emission, b);
        CL(lw_int->top_dn, b) = 0.0;
        CL(lw_int->top_net, b) = -CL(lw->ground_emission, b);
        CL(lw_norm->ground_dn, b) = 1.0;
        CL(lw_norm->ground_net, b) = CL(lw->ground_emissivity, b);
        CL(lw_norm->ground_vertical_diff, b) = 0.5 * (2.0 - CL(lw->ground_emissivity, b));
        CL(lw_norm->top_dn, b) = 1.0;
        CL(lw_norm->top_net, b) = CL(lw->ground_emissivity, b);
      }
    break;
  case SS_TILE_FOREST:
  case SS_TILE_URBAN:
  case SS_TILE_VEGETATED_URBAN: {
    const int is_urban = (irep != SS_TILE_FOREST);
    const int nreg = (irep == SS_TILE_FOREST) ? cfg->n_vegetation_region_forest + 1
                     : (irep == SS_TILE_URBAN) ? 1 : cfg->n_vegetation_region_urban + 1;
    const ss_legendre_gauss *lgs = is_urban ? &cfg->lg_sw_urban : &cfg->lg_sw_forest;
    const ss_legendre_gauss *lgl = is_urban ? &cfg->lg_lw_urban : &cfg->lg_lw_forest;
    if (cfg->do_sw) {
      /* type 2 zeroes before solving, the others only when the sun is down
       * (radsurf_interface.F90:215-216 vs :193-196) */
      if (irep == SS_TILE_URBAN) {
        flux_zero(sw_dir, jcol, l0, nlay);
        flux_zero(sw_diff, jcol, l0, nlay);
      }
      if (cp->cos_sza[jcol] > 0.0) {
        rc = solve_sw(cfg, is_urban, nsw, lgs->nstream, nreg, nlay, jcol, l0, lgs, cp->cos_sza[jcol],
                      cp, sw, sw->ground_albedo + (size_t)nsw * jcol, galb_dir + (size_t)nsw * jcol,
                      bc->sw_albedo + (size_t)nsw * jcol, bc->sw_albedo_dir + (size_t)nsw * jcol,
                      sw_dir, sw_diff);
        if (rc) return rc;
      } else if (irep != SS_TILE_URBAN) {
        flux_zero(sw_dir, jcol, l0, nlay);
        flux_zero(sw_diff, jcol, l0, nlay);
      }
    }
    if (cfg->do_lw) {
      if (irep == SS_TILE_URBAN) {
        flux_zero(lw_int, jcol, l0, nlay);
        flux_zero(lw_norm, jcol, l0, nlay);
      }
      rc = solve_lw(cfg, is_urban, nlw, lgl->nstream, nreg, nlay, jcol, l0, lgl, cp, lw,
                    bc->lw_emissivity + (size_t)nlw * jcol, bc->lw_emission + (size_t)nlw * jcol,
                    lw_int, lw_norm);
      if (rc) return rc;
    }
    break;
  }
  default: {
    /* simple-urban tiles (types 4, 5), radsurf_interface.F90:271-308 */
    const int infinite = (irep == SS_TILE_INFINITE_STREET);
    if (nlay > 1) return SS_ERR_SIMPLE_URBAN_LAYERS;
    if (nlay < 1) return SS_ERR_BAD_ARGUMENT;
    if (cfg->do_sw) {
      if (cp->cos_sza[jcol] > 0.0) {
        simple_urban_sw(cfg, infinite, nsw, jcol, l0, cp->cos_sza[jcol], cp, sw,
                        sw->ground_albedo + (size_t)nsw * jcol, galb_dir + (size_t)nsw * jcol,
                        sw_dir, sw_diff);
      } else {
        flux_zero(sw_dir, jcol, l0, nlay);
        flux_zero(sw_diff, jcol, l0, nlay);
      }
    }
    if (cfg->do_lw) simple_urban_lw(cfg, infinite, nlw, jcol, l0, cp, lw, lw_int, lw_norm);
    break;
  }
  }
  return SS_OK;
#undef C
#undef CL
}

int spartacus_radsurf_oracle(const ss_config *config, const ss_canopy *canopy,
                             const ss_sw_spectral *sw, const ss_lw_spectral *lw,
                             ss_bc_out *bc_out, int32_t istartcol, int32_t iendcol,
                             ss_flux *sw_norm_dir, ss_flux *sw_norm_diff,
                             ss_flux *lw_internal, ss_flux *lw_norm,
                             int32_t do_simple_spectrum, int32_t nthreads, double *elapsed) {
  int status = SS_OK;
  double t0;
  if (nthreads < 1) nthreads = 1;
  t0 = omp_get_wtime();
  if (config->do_lw && do_simple_spectrum) {
    status = simple_spectrum_lw(config, canopy, lw, istartcol, iendcol);
    if (status) return status;
  }
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
  for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
    int rc = radsurf_column(config, canopy, sw, lw, bc_out, jcol, sw_norm_dir, sw_norm_diff,
                            lw_internal, lw_norm);
    if (rc) {
#pragma omp critical
      status = rc;
    }
  }
  if (elapsed) *elapsed = omp_get_wtime() - t0;
  return status;
}

/* ---- kernel-level entries on (nmat, i, j) stacked Fortran-layout arrays ---- */

static void unstack(const double *src, int nmat, int jmat, int rows, int cols, or_real *dst) {
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < rows; ++i) dst[i + (size_t)j * rows] = src[jmat + (size_t)nmat * (i + (size_t)rows * j)];
}
static void restack(const or_real *src, int nmat, int jmat, int rows, int cols, double *dst) {
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < rows; ++i) dst[jmat + (size_t)nmat * (i + (size_t)rows * j)] = src[i + (size_t)j * rows];
}

void oracle_eigen_decomposition_real(int32_t norder, int32_t nmat, const double *amat,
                                     double *eigenvalue, double *eigenvector) {
  or_real a[OR_MAXN * OR_MAXN], val[OR_MAXN], vec[OR_MAXN * OR_MAXN];
  for (int m = 0; m < nmat; ++m) {
    unstack(amat, nmat, m, norder, norder, a);
    memset(vec, 0, sizeof(vec));
    or_eigen_decomposition_real(norder, a, val, vec);
    restack(val, nmat, m, norder, 1, eigenvalue);
    restack(vec, nmat, m, norder, norder, eigenvector);
  }
}

void oracle_calc_matrices_sw_eig(int32_t nmat, int32_t ndiff, int32_t ndir, double dz, double mu0,
                                 const double *gamma0, const double *gamma1, const double *gamma2,
                                 const double *gamma3, double *reflectance, double *transmittance,
                                 double *s_up, double *s_dn, double *trans_dir, double *int_dir,
                                 double *int_diff, double *int_dir_diff) {
  or_real g0[9], g1[OR_MAXN * OR_MAXN], g2[OR_MAXN * OR_MAXN], g3[OR_MAXN * 3];
  or_real oR[OR_MAXN * OR_MAXN], oT[OR_MAXN * OR_MAXN], oSu[OR_MAXN * 3], oSd[OR_MAXN * 3];
  or_real oE[9], oId[9], oIf[OR_MAXN * OR_MAXN], oIdf[OR_MAXN * 3];
  for (int m = 0; m < nmat; ++m) {
    unstack(gamma0, nmat, m, ndir, ndir, g0);
    unstack(gamma1, nmat, m, ndiff, ndiff, g1);
    unstack(gamma2, nmat, m, ndiff, ndiff, g2);
    unstack(gamma3, nmat, m, ndiff, ndir, g3);
    or_calc_matrices_sw_eig(ndiff, ndir, dz, mu0, g0, g1, g2, g3, oR, oT, oSu, oSd, oE, oId, oIf, oIdf);
    restack(oR, nmat, m, ndiff, ndiff, reflectance);
    restack(oT, nmat, m, ndiff, ndiff, transmittance);
    restack(oSu, nmat, m, ndiff, ndir, s_up);
    restack(oSd, nmat, m, ndiff, ndir, s_dn);
    restack(oE, nmat, m, ndir, ndir, trans_dir);
    restack(oId, nmat, m, ndir, ndir, int_dir);
    restack(oIf, nmat, m, ndiff, ndiff, int_diff);
    restack(oIdf, nmat, m, ndiff, ndir, int_dir_diff);
  }
}

void oracle_calc_matrices_lw_eig(int32_t nmat, int32_t norder, double dz, const double *gamma1,
                                 const double *gamma2, const double *emiss_rate, double *reflectance,
                                 double *transmittance, double *source, double *int_flux,
                                 double *int_flux_source) {
  or_real g1[OR_MAXN * OR_MAXN], g2[OR_MAXN * OR_MAXN], b[OR_MAXN];
  or_real oR[OR_MAXN * OR_MAXN], oT[OR_MAXN * OR_MAXN], oS[OR_MAXN], oIf[OR_MAXN * OR_MAXN], oIs[OR_MAXN];
  for (int m = 0; m < nmat; ++m) {
    unstack(gamma1, nmat, m, norder, norder, g1);
    unstack(gamma2, nmat, m, norder, norder, g2);
    unstack(emiss_rate, nmat, m, norder, 1, b);
    or_calc_matrices_lw_eig(norder, dz, g1, g2, b, oR, oT, oS, oIf, oIs);
    restack(oR, nmat, m, norder, norder, reflectance);
    restack(oT, nmat, m, norder, norder, transmittance);
    restack(oS, nmat, m, norder, 1, source);
    restack(oIf, nmat, m, norder, norder, int_flux);
    restack(oIs, nmat, m, norder, 1, int_flux_source);
  }
}
