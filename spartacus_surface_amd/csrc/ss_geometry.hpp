// ss_geometry.hpp - band-independent geometry of one column, evaluated locally
// per layer / per interface so that no per-column geometry arrays are needed:
// region fractions, normalized perimeter lengths (radsurf/radsurf_norm_perim.F90),
// exchange and wall-interception rates, and the directional overlap matrices of
// one interface (radsurf/radsurf_overlap.F90).  All of it depends only on the
// canopy properties of the layer itself and of its lower neighbour.
#pragma once

#include "ss_common.hpp"

namespace ss {

// Geometry inputs of one layer
struct LayerGeom {
  double dz, veg_fraction, veg_scale, veg_ext, veg_fsd;
  double building_fraction, building_scale, veg_contact_fraction;
};

template <bool URBAN, int NREG>
SS_HD LayerGeom load_layer_geom(const Inputs& in, int lay) {
  LayerGeom g;
  g.dz = in.dz[lay];
  g.veg_fraction = g.veg_scale = g.veg_ext = g.veg_fsd = 0.0;
  g.building_fraction = g.building_scale = g.veg_contact_fraction = 0.0;
  if (NREG > 1) {   // vegetation arrays are never touched for NREG == 1
    g.veg_fraction = in.veg_fraction[lay];
    g.veg_scale = in.veg_scale[lay];
    g.veg_ext = in.veg_ext[lay];
    if (NREG == 3) g.veg_fsd = in.veg_fsd[lay];
  }
  if (URBAN) {
    g.building_fraction = in.building_fraction[lay];
    g.building_scale = in.building_scale[lay];
    if (NREG > 1) g.veg_contact_fraction = in.veg_contact_fraction[lay];
  }
  return g;
}

// Region fractions of one layer: forest SW radsurf_forest_sw.F90:244-245,
// forest LW radsurf_forest_lw.F90:245-247 (1-(1-v) instead of v),
// urban radsurf_urban_sw.F90:284-291.
template <bool URBAN, bool LW, int NREG>
SS_HD void region_fractions(double veg_fraction, double building_fraction, double* frac) {
  if (!URBAN) {
    frac[0] = 1.0 - veg_fraction;
    SS_UNROLL_FULL
    for (int r = 1; r < NREG; ++r)
      frac[r] = (LW ? (1.0 - frac[0]) : veg_fraction) / (double)(NREG - 1);
  } else {
    frac[0] = 1.0 - building_fraction;
    if (NREG > 1) {
      frac[0] = ss_max(0.0, frac[0] - veg_fraction);
      SS_UNROLL_FULL
      for (int r = 1; r < NREG; ++r)
        frac[r] = ss_max(0.0, 1.0 - building_fraction - frac[0]) / (double)(NREG - 1);
    }
  }
}

// calc_norm_perim_forest (norm_perim.F90:30-116) / calc_norm_perim_urban
// (:131-281) for one layer.
template <bool URBAN, int NREG>
SS_HD void norm_perimeters(const Cfg& cfg, const LayerGeom& g, double* np, double* npw) {
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) { np[r] = 0.0; npw[r] = 0.0; }
  const double iso = cfg.vegetation_isolation_factor;
  const double vf = g.veg_fraction, bf = g.building_fraction;
  if (NREG > 1) {
    if (vf > cfg.min_vegetation_fraction) {
      if (cfg.use_symmetric_vegetation_scale) {
        if (URBAN)
          np[0] = 4.0 * vf * ss_max(0.0, 1.0 - vf - bf)
                  / (ss_max(cfg.min_building_fraction, 1.0 - bf) * g.veg_scale);
        else
          np[0] = 4.0 * vf * ss_max(0.0, 1.0 - vf) / g.veg_scale;
      } else {
        np[0] = 4.0 * vf / g.veg_scale;
      }
      if (NREG > 2) {
        np[NREG - 1] = 0.5 * iso * np[0];
        np[0] = (1.0 - 0.5 * iso) * np[0];
        if (cfg.use_symmetric_vegetation_scale) {
          if (URBAN)
            np[1] = (1.0 - iso) * 4.0 * (0.5 * vf) * (1.0 - (0.5 * vf) - bf)
                    / (ss_max(cfg.min_building_fraction, 1.0 - bf) * g.veg_scale);
          else
            np[1] = (1.0 - iso) * 4.0 * (0.5 * vf) * (1.0 - (0.5 * vf)) / g.veg_scale;
        } else {
          np[1] = (1.0 - iso) * 4.0 * vf / (sqrt(2.0) * g.veg_scale);
        }
      }
    }
  }
  if (URBAN) {
    if (bf > cfg.min_building_fraction) {
      npw[0] = 4.0 * bf / g.building_scale;
      if (NREG > 1) {
        if (1.0 - vf - bf <= cfg.min_vegetation_fraction) {
          if (NREG == 2) {
            npw[1] = npw[0];
          } else {
            npw[1] = npw[0] * (1.0 - iso);
            npw[NREG - 1] = npw[0] * iso;
          }
          npw[0] = 0.0;
        } else if (vf > cfg.min_vegetation_fraction) {
          if (g.veg_contact_fraction > 0.0) {
            if (NREG == 2) {
              npw[1] = npw[0] * g.veg_contact_fraction;
            } else {
              npw[1] = npw[0] * g.veg_contact_fraction * (1.0 - iso);
              npw[NREG - 1] = npw[0] * g.veg_contact_fraction * iso;
            }
            npw[0] = npw[0] * (1.0 - g.veg_contact_fraction);
          }
        }
      }
    }
  }
}

// f_exchange(to,from) (radsurf_forest_sw.F90:299-321; uses norm_perim(nreg)
// for the 1<->3 pair)
template <int NREG>
SS_HD void exchange_rates(const double* frac, const double* np, double min_veg, double* fx) {
  SS_UNROLL_FULL
  for (int k = 0; k < NREG * NREG; ++k) fx[k] = 0.0;
  SS_UNROLL_FULL
  for (int r = 0; r < NREG - 1; ++r) {
    if (!(frac[r] <= min_veg || frac[r + 1] <= min_veg)) {
      SS_E(fx, r + 1, r, NREG) = np[r] / (kPi * frac[r]);
      SS_E(fx, r, r + 1, NREG) = np[r] / (kPi * frac[r + 1]);
    }
  }
  if (NREG > 2) {
    if (np[NREG - 1] > 0.0) {
      if (!(frac[NREG - 1] <= min_veg || frac[0] <= min_veg)) {
        SS_E(fx, 0, NREG - 1, NREG) = np[NREG - 1] / (kPi * frac[NREG - 1]);
        SS_E(fx, NREG - 1, 0, NREG) = np[NREG - 1] / (kPi * frac[0]);
      }
    }
  }
}

// od_scaling (radsurf_forest_sw.F90:283-288); od[0] unused
template <int NREG>
SS_HD void od_scaling(double veg_fsd, double* od) {
  od[0] = 0.0;
  if (NREG == 2) od[1] = 1.0;
  if (NREG == 3) {
    od[1] = exp(-veg_fsd * (1.0 + 0.5 * veg_fsd * (1.0 + 0.5 * veg_fsd)));
    od[NREG - 1] = 2.0 - od[1];
  }
}

// ---- overlap at one interface -------------------------------------------
// U is NREG x DB (upper x lower), V is DB x NREG (lower x upper), column-major,
// DB = NREG (forest) or NREG+1 (urban, last lower region = exposed roof).

// Forest: calc_overlap_matrix_max_ran :28-73 + directional part of
// calc_overlap_matrices :85-171.  fu / fl: region fractions of the upper /
// lower layer (ground: (1,0,..), above canopy: (1,0,..)).
template <int NREG>
SS_HD void overlap_forest(const double* fu, const double* fl, double thr, double* U, double* V) {
  double om[NREG * NREG];
  const double f_upper = 1.0 - fu[0], f_lower = 1.0 - fl[0];
  const double pair_cover = ss_max(f_upper, f_lower);
  SS_UNROLL_FULL
  for (int k = 0; k < NREG * NREG; ++k) om[k] = 0.0;
  SS_E(om, 0, 0, NREG) = 1.0 - pair_cover;
  if (NREG == 2) {
    SS_E(om, 0, 1, NREG) = pair_cover - f_upper;
    SS_E(om, 1, 0, NREG) = pair_cover - f_lower;
    SS_E(om, 1, 1, NREG) = f_upper + f_lower - pair_cover;
  } else if (NREG == 3) {
    SS_E(om, 0, 1, NREG) = 0.5 * (pair_cover - f_upper);
    SS_E(om, 0, NREG - 1, NREG) = SS_E(om, 0, 1, NREG);
    SS_E(om, 1, 0, NREG) = 0.5 * (pair_cover - f_lower);
    SS_E(om, NREG - 1, 0, NREG) = SS_E(om, 1, 0, NREG);
    SS_E(om, 1, 1, NREG) = 0.5 * (f_upper + f_lower - pair_cover);
    SS_E(om, NREG - 1, NREG - 1, NREG) = SS_E(om, 1, 1, NREG);
  }
  SS_UNROLL_FULL
  for (int ju = 0; ju < NREG; ++ju) {
    SS_UNROLL_FULL
    for (int jl = 0; jl < NREG; ++jl) {
      SS_E(U, ju, jl, NREG) = (fl[jl] >= thr) ? SS_E(om, ju, jl, NREG) / fl[jl] : 0.0;
      SS_E(V, jl, ju, NREG) = (fu[ju] >= thr) ? SS_E(om, ju, jl, NREG) / fu[ju] : 0.0;
    }
  }
}

// Urban: calc_overlap_matrix_max_ran_urban :178-280 + calc_overlap_matrices_urban
// :289-394.  fl has NREG+1 entries (last = exposed roof / ground).
template <int NREG>
SS_HD void overlap_urban(const double* fu, const double* fl, double thr, double* U, double* V) {
  constexpr int DB = NREG + 1;
  double om[NREG * DB];
  SS_UNROLL_FULL
  for (int k = 0; k < NREG * DB; ++k) om[k] = 0.0;
  if (NREG == 1) {
    SS_E(om, 0, 0, NREG) = fl[0];
    SS_E(om, 0, 1, NREG) = fl[1];
  } else if (NREG == 2) {
    const double pair_cover = ss_max(fu[1], fl[1]);
    if (pair_cover <= fl[0] + fl[1]) {
      SS_E(om, 0, 2, NREG) = fl[2];
      SS_E(om, 0, 0, NREG) = fl[0] + fl[1] - pair_cover;
      SS_E(om, 0, 1, NREG) = pair_cover - fu[1];
      SS_E(om, 1, 0, NREG) = pair_cover - fl[1];
      SS_E(om, 1, 1, NREG) = fu[1] + fl[1] - pair_cover;
    } else {
      SS_E(om, 1, 0, NREG) = fl[0];
      SS_E(om, 1, 1, NREG) = fl[1];
      SS_E(om, 1, 2, NREG) = fu[1] - fl[0] - fl[1];
      SS_E(om, 0, 2, NREG) = fu[0];
    }
  } else if (NREG == 3) {
    const int R2 = NREG - 1;   // index of region 3 (keeps NREG<3 instantiations in bounds)
    const double pair_cover = ss_max(fu[1] + fu[R2], fl[1] + fl[R2]);
    if (pair_cover <= fl[0] + fl[1] + fl[R2]) {
      SS_E(om, 0, DB - 1, NREG) = fl[DB - 1];
      SS_E(om, 0, 0, NREG) = fl[0] + fl[1] + fl[R2] - pair_cover;
      if (pair_cover > fu[1] + fu[R2]) {
        SS_E(om, 1, 1, NREG) = fu[1];
        SS_E(om, R2, R2, NREG) = fu[R2];
        SS_E(om, 0, 1, NREG) = fl[1] - fu[1];
        SS_E(om, 0, R2, NREG) = fl[R2] - fu[R2];
      } else {
        SS_E(om, 1, 1, NREG) = fl[1];
        SS_E(om, R2, R2, NREG) = fl[R2];
        SS_E(om, 1, 0, NREG) = fu[1] - fl[1];
        SS_E(om, R2, 0, NREG) = fu[R2] - fl[R2];
      }
    } else {
      SS_E(om, 1, 1, NREG) = fl[1];
      SS_E(om, R2, R2, NREG) = fl[R2];
      SS_E(om, 1, 0, NREG) = fl[0] * 0.5;
      // overlap_matrix(3,1) = overlap_matrix(1,2) which is zero here
      // (radsurf_overlap.F90:264,271)
      SS_E(om, 1, DB - 1, NREG) = (fl[DB - 1] - fu[0]) * 0.5;
      SS_E(om, R2, DB - 1, NREG) = SS_E(om, 1, DB - 1, NREG);
      SS_E(om, 0, DB - 1, NREG) = fu[0];
    }
  }
  SS_UNROLL_FULL
  for (int ju = 0; ju < NREG; ++ju) {
    SS_UNROLL_FULL
    for (int jl = 0; jl < DB; ++jl) {
      SS_E(U, ju, jl, NREG) = (fl[jl] >= thr) ? SS_E(om, ju, jl, NREG) / fl[jl] : 0.0;
      SS_E(V, jl, ju, DB) = (fu[ju] >= thr) ? SS_E(om, ju, jl, NREG) / fu[ju] : 0.0;
    }
  }
}

// Fractions "just below" interface j (1 <= j <= nlay, 0-based layer j-1 below,
// layer j above) of an urban column, from the fractions of the two layers:
// calc_overlap_matrices_urban :373-391 (incl. the overhanging-building rescale).
// frac_lo: fractions of the lower layer; sum_up: sum of fractions of the upper
// layer (or 1 at the canopy top).
template <int NREG>
SS_HD void urban_frac_lower(const double* frac_lo, double sum_up, bool is_top, double* fl) {
  double sum_lo = 0.0;
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) { fl[r] = frac_lo[r]; sum_lo += frac_lo[r]; }
  if (is_top) {
    fl[NREG] = 1.0 - sum_lo;
  } else {
    fl[NREG] = sum_up - sum_lo;
    if (fl[NREG] < 0.0) {
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) fl[r] = fl[r] * sum_up / sum_lo;
      fl[NREG] = 0.0;
    }
  }
}

}  // namespace ss
