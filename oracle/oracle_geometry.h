/* oracle_geometry.h - band-independent per-column geometry of the reference:
 * overlap matrices (radsurf/radsurf_overlap.F90) and normalized perimeter
 * lengths (radsurf/radsurf_norm_perim.F90).  TEST INFRASTRUCTURE ONLY.
 * Matrices column-major, 0-based; u_overlap is (nup x nlo x nlay+1) etc.
 */
#ifndef ORACLE_GEOMETRY_H
#define ORACLE_GEOMETRY_H

#include "oracle_radtool.h"

/* calc_overlap_matrix_max_ran :28-73 (nreg = 2 or 3; also called with nreg=1
 * never - forests always have >= 2 regions) */
static void or_overlap_matrix_max_ran(int nreg, or_real f_upper, or_real f_lower, or_real *om) {
  or_real pair_cover = OR_FMAX(f_upper, f_lower);
  E(om, 0, 0, nreg) = 1.0 - pair_cover;
  if (nreg == 2) {
    E(om, 0, 1, nreg) = pair_cover - f_upper;
    E(om, 1, 0, nreg) = pair_cover - f_lower;
    E(om, 1, 1, nreg) = f_upper + f_lower - pair_cover;
  } else if (nreg == 3) {
    E(om, 0, 1, nreg) = 0.5 * (pair_cover - f_upper);
    E(om, 0, 2, nreg) = E(om, 0, 1, nreg);
    E(om, 1, 0, nreg) = 0.5 * (pair_cover - f_lower);
    E(om, 2, 0, nreg) = E(om, 1, 0, nreg);
    E(om, 1, 1, nreg) = 0.5 * (f_upper + f_lower - pair_cover);
    E(om, 2, 2, nreg) = E(om, 1, 1, nreg);
    E(om, 1, 2, nreg) = 0.0;
    E(om, 2, 1, nreg) = 0.0;
  }
}

/* calc_overlap_matrices :85-171.  frac is (nreg x nlay) [the caller's
 * (nreg x nlay+1) array works too: only the first nlay columns are read].
 * u_overlap(jupper,jlower,jlay): nreg x nreg x (nlay+1);
 * v_overlap(jlower,jupper,jlay): nreg x nreg x (nlay+1). */
static void or_calc_overlap_matrices(int nlay, int nreg, const or_real *frac, int ldfrac,
                                     or_real *u_overlap, or_real *v_overlap,
                                     or_real frac_threshold) {
  or_real om[9], frac_upper[3], frac_lower[3];
  frac_lower[0] = 1.0;
  for (int r = 1; r < nreg; ++r) frac_lower[r] = 0.0;
  for (int jlay = 0; jlay <= nlay; ++jlay) {
    or_real *u = u_overlap + (size_t)jlay * nreg * nreg;
    or_real *v = v_overlap + (size_t)jlay * nreg * nreg;
    if (jlay >= nlay) {
      frac_upper[0] = 1.0;
      for (int r = 1; r < nreg; ++r) frac_upper[r] = 0.0;
    } else {
      for (int r = 0; r < nreg; ++r) frac_upper[r] = E(frac, r, jlay, ldfrac);
    }
    or_overlap_matrix_max_ran(nreg, 1.0 - frac_upper[0], 1.0 - frac_lower[0], om);
    for (int jupper = 0; jupper < nreg; ++jupper)
      for (int jlower = 0; jlower < nreg; ++jlower) {
        if (frac_lower[jlower] >= frac_threshold)
          E(u, jupper, jlower, nreg) = E(om, jupper, jlower, nreg) / frac_lower[jlower];
        else
          E(u, jupper, jlower, nreg) = 0.0;
        if (frac_upper[jupper] >= frac_threshold)
          E(v, jlower, jupper, nreg) = E(om, jupper, jlower, nreg) / frac_upper[jupper];
        else
          E(v, jlower, jupper, nreg) = 0.0;
      }
    for (int r = 0; r < nreg; ++r) frac_lower[r] = frac_upper[r];
  }
}

/* calc_overlap_matrix_max_ran_urban :178-280.  om is nreg x (nreg+1),
 * first index upper layer, second lower layer (last = exposed roof). */
static void or_overlap_matrix_max_ran_urban(int nreg, const or_real *fu, const or_real *fl,
                                            or_real *om) {
  or_real pair_cover;
  if (nreg == 1) {
    E(om, 0, 0, 1) = fl[0];
    E(om, 0, 1, 1) = fl[1];
  } else if (nreg == 2) {
    pair_cover = OR_FMAX(fu[1], fl[1]);
    if (pair_cover <= fl[0] + fl[1]) {
      E(om, 1, 2, 2) = 0.0;
      E(om, 0, 2, 2) = fl[2];
      E(om, 0, 0, 2) = fl[0] + fl[1] - pair_cover;
      E(om, 0, 1, 2) = pair_cover - fu[1];
      E(om, 1, 0, 2) = pair_cover - fl[1];
      E(om, 1, 1, 2) = fu[1] + fl[1] - pair_cover;
    } else {
      E(om, 0, 0, 2) = 0.0;
      E(om, 0, 1, 2) = 0.0;
      E(om, 1, 0, 2) = fl[0];
      E(om, 1, 1, 2) = fl[1];
      E(om, 1, 2, 2) = fu[1] - fl[0] - fl[1];
      E(om, 0, 2, 2) = fu[0];
    }
  } else {
    E(om, 1, 2, 3) = 0.0;
    E(om, 2, 1, 3) = 0.0;
    pair_cover = OR_FMAX(fu[1] + fu[2], fl[1] + fl[2]);
    if (pair_cover <= fl[0] + fl[1] + fl[2]) {
      E(om, 1, 3, 3) = 0.0;
      E(om, 2, 3, 3) = 0.0;
      E(om, 0, 3, 3) = fl[3];
      E(om, 0, 0, 3) = fl[0] + fl[1] + fl[2] - pair_cover;
      if (pair_cover > fu[1] + fu[2]) {
        E(om, 1, 0, 3) = 0.0;
        E(om, 2, 0, 3) = 0.0;
        E(om, 1, 1, 3) = fu[1];
        E(om, 2, 2, 3) = fu[2];
        E(om, 0, 1, 3) = fl[1] - fu[1];
        E(om, 0, 2, 3) = fl[2] - fu[2];
      } else {
        E(om, 0, 1, 3) = 0.0;
        E(om, 0, 2, 3) = 0.0;
        E(om, 1, 1, 3) = fl[1];
        E(om, 2, 2, 3) = fl[2];
        E(om, 1, 0, 3) = fu[1] - fl[1];
        E(om, 2, 0, 3) = fu[2] - fl[2];
      }
    } else {
      E(om, 0, 0, 3) = 0.0;
      E(om, 0, 1, 3) = 0.0;
      E(om, 0, 2, 3) = 0.0;
      E(om, 1, 1, 3) = fl[1];
      E(om, 2, 2, 3) = fl[2];
      E(om, 1, 0, 3) = fl[0] * 0.5;
      /* :271 as written: overlap_matrix(3,1) = overlap_matrix(1,2), which was
       * just zeroed at :264 */
      E(om, 2, 0, 3) = E(om, 0, 1, 3);
      E(om, 1, 3, 3) = (fl[3] - fu[0]) * 0.5;
      E(om, 2, 3, 3) = E(om, 1, 3, 3);
      E(om, 0, 3, 3) = fu[0];
    }
  }
}

/* calc_overlap_matrices_urban :289-394.
 * u_overlap(jupper,jlower,jlay): nreg x (nreg+1) x (nlay+1)
 * v_overlap(jlower,jupper,jlay): (nreg+1) x nreg x (nlay+1) */
static void or_calc_overlap_matrices_urban(int nlay, int nreg, const or_real *frac, int ldfrac,
                                           or_real *u_overlap, or_real *v_overlap,
                                           or_real frac_threshold) {
  const int nlo = nreg + 1;
  or_real om[12], frac_upper[3], frac_lower[4];
  or_real s0 = 0.0;
  for (int r = 0; r < nreg; ++r) { frac_lower[r] = 0.0; s0 += E(frac, r, 0, ldfrac); }
  frac_lower[nreg] = s0;
  for (int jlay = 0; jlay <= nlay; ++jlay) {
    or_real *u = u_overlap + (size_t)jlay * nreg * nlo;
    or_real *v = v_overlap + (size_t)jlay * nlo * nreg;
    if (jlay >= nlay) {
      frac_upper[0] = 1.0;
      for (int r = 1; r < nreg; ++r) frac_upper[r] = 0.0;
    } else {
      for (int r = 0; r < nreg; ++r) frac_upper[r] = E(frac, r, jlay, ldfrac);
    }
    or_overlap_matrix_max_ran_urban(nreg, frac_upper, frac_lower, om);
    for (int jupper = 0; jupper < nreg; ++jupper)
      for (int jlower = 0; jlower < nlo; ++jlower) {
        if (frac_lower[jlower] >= frac_threshold)
          E(u, jupper, jlower, nreg) = E(om, jupper, jlower, nreg) / frac_lower[jlower];
        else
          E(u, jupper, jlower, nreg) = 0.0;
        if (frac_upper[jupper] >= frac_threshold)
          E(v, jlower, jupper, nlo) = E(om, jupper, jlower, nreg) / frac_upper[jupper];
        else
          E(v, jlower, jupper, nlo) = 0.0;
      }
    for (int r = 0; r < nreg; ++r) frac_lower[r] = frac_upper[r];
    /* jlay is 0-based here: Fortran "jlay < nlay" <=> jlay+1 < nlay */
    if (jlay + 1 < nlay) {
      or_real sup = 0.0, slo = 0.0;
      for (int r = 0; r < nreg; ++r) { sup += E(frac, r, jlay + 1, ldfrac); slo += E(frac, r, jlay, ldfrac); }
      frac_lower[nreg] = sup - slo;
      if (frac_lower[nreg] < 0.0) {
        for (int r = 0; r < nreg; ++r) frac_lower[r] = frac_lower[r] * sup / slo;
        frac_lower[nreg] = 0.0;
      }
    } else if (jlay + 1 == nlay) {
      or_real slo = 0.0;
      for (int r = 0; r < nreg; ++r) slo += E(frac, r, jlay, ldfrac);
      frac_lower[nreg] = 1.0 - slo;
    }
  }
}

/* calc_norm_perim_forest :30-116.  norm_perim is nreg x nlay. */
static void or_calc_norm_perim_forest(const ss_config *config, int nlay, int nreg,
                                      const double *veg_fraction, const double *veg_scale,
                                      or_real *norm_perim) {
  const or_real iso = config->vegetation_isolation_factor_forest;
  for (int k = 0; k < nreg * nlay; ++k) norm_perim[k] = 0.0;
  for (int jlay = 0; jlay < nlay; ++jlay) {
    or_real *np = norm_perim + (size_t)jlay * nreg;
    if (nreg > 1) {
      if (veg_fraction[jlay] > config->min_vegetation_fraction) {
        if (config->use_symmetric_vegetation_scale_forest)
          np[0] = 4.0 * veg_fraction[jlay] * OR_FMAX(0.0, 1.0 - veg_fraction[jlay]) / veg_scale[jlay];
        else
          np[0] = 4.0 * veg_fraction[jlay] / veg_scale[jlay];
        if (nreg > 2) {
          np[nreg - 1] = 0.5 * iso * np[0];
          np[0] = (1.0 - 0.5 * iso) * np[0];
          if (config->use_symmetric_vegetation_scale_forest)
            np[1] = (1.0 - iso) * 4.0 * (0.5 * veg_fraction[jlay])
                    * (1.0 - (0.5 * veg_fraction[jlay])) / veg_scale[jlay];
          else
            np[1] = (1.0 - iso) * 4.0 * veg_fraction[jlay] / (OR_SQRT(2.0) * veg_scale[jlay]);
        } else {
          for (int r = 1; r < nreg; ++r) np[r] = 0.0;
        }
      }
    }
  }
}

/* calc_norm_perim_urban :131-281.  veg_* pointers are not touched when
 * nreg == 1. */
static void or_calc_norm_perim_urban(const ss_config *config, int nlay, int nreg,
                                     const double *building_fraction,
                                     const double *building_scale, const double *veg_fraction,
                                     const double *veg_scale, const double *veg_contact_fraction,
                                     or_real *norm_perim, or_real *norm_perim_wall) {
  const or_real iso = config->vegetation_isolation_factor_urban;
  for (int k = 0; k < nreg * nlay; ++k) { norm_perim[k] = 0.0; norm_perim_wall[k] = 0.0; }
  for (int jlay = 0; jlay < nlay; ++jlay) {
    or_real *np = norm_perim + (size_t)jlay * nreg;
    or_real *npw = norm_perim_wall + (size_t)jlay * nreg;
    if (nreg > 1) {
      if (veg_fraction[jlay] > config->min_vegetation_fraction) {
        if (config->use_symmetric_vegetation_scale_urban)
          np[0] = 4.0 * veg_fraction[jlay]
                  * OR_FMAX(0.0, 1.0 - veg_fraction[jlay] - building_fraction[jlay])
                  / (OR_FMAX(config->min_building_fraction, 1.0 - building_fraction[jlay])
                     * veg_scale[jlay]);
        else
          np[0] = 4.0 * veg_fraction[jlay] / veg_scale[jlay];
        if (nreg > 2) {
          np[nreg - 1] = 0.5 * iso * np[0];
          np[0] = (1.0 - 0.5 * iso) * np[0];
          if (config->use_symmetric_vegetation_scale_urban)
            np[1] = (1.0 - iso) * 4.0 * (0.5 * veg_fraction[jlay])
                    * (1.0 - (0.5 * veg_fraction[jlay]) - building_fraction[jlay])
                    / (OR_FMAX(config->min_building_fraction, 1.0 - building_fraction[jlay])
                       * veg_scale[jlay]);
          else
            np[1] = (1.0 - iso) * 4.0 * veg_fraction[jlay] / (OR_SQRT(2.0) * veg_scale[jlay]);
        } else {
          for (int r = 1; r < nreg; ++r) np[r] = 0.0;
        }
      }
    }
    if (building_fraction[jlay] > config->min_building_fraction) {
      npw[0] = 4.0 * building_fraction[jlay] / building_scale[jlay];
      if (nreg > 1) {
        if (1.0 - veg_fraction[jlay] - building_fraction[jlay] <= config->min_vegetation_fraction) {
          if (nreg == 2) {
            npw[1] = npw[0];
          } else {
            npw[1] = npw[0] * (1.0 - iso);
            npw[2] = npw[0] * iso;
          }
          npw[0] = 0.0;
        } else if (veg_fraction[jlay] > config->min_vegetation_fraction) {
          if (veg_contact_fraction[jlay] > 0.0) {
            if (nreg == 2) {
              npw[1] = npw[0] * veg_contact_fraction[jlay];
            } else {
              npw[1] = npw[0] * veg_contact_fraction[jlay] * (1.0 - iso);
              npw[2] = npw[0] * veg_contact_fraction[jlay] * iso;
            }
            npw[0] = npw[0] * (1.0 - veg_contact_fraction[jlay]);
          }
        }
      }
    }
  }
}

#endif /* ORACLE_GEOMETRY_H */
