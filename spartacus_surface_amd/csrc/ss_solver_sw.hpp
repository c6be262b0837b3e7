// ss_solver_sw.hpp - shortwave solver units.
//
//   layer_unit_sw  : one (column, band, layer): region optics, exchange rates,
//                    Gamma matrices and the layer matrices R, T, S_up, S_dn, E,
//                    int_* (sections 3a-3c of spartacus_forest_sw,
//                    radsurf_forest_sw.F90:270-457, and of spartacus_urban_sw,
//                    radsurf_urban_sw.F90:335-585) -> HBM layer record.
//   column_unit_sw : one (column, band): adding method upward (section 4,
//                    forest :463-528 / urban :591-674) and the two downward
//                    flux passes (section 5, forest :530-760 / urban :676-984)
//                    -> bc_out and canopy_flux arrays.
//
// Template parameters: URBAN (walls + exposed-roof pseudo region), NREG
// (regions: 1 clear + NREG-1 vegetated), NS (streams per hemisphere).
#pragma once

#include "ss_common.hpp"
#include "ss_linalg.hpp"
#include "ss_geometry.hpp"
#include "ss_layer_matrices.hpp"

namespace ss {

// Adding-method record of one layer in the column scratch [layer][elem][cu]
template <int N, int D>
struct SwColRec {
  static constexpr int AAB = 0;                 // a_above(l)        N x N
  static constexpr int DAB = AAB + N * N;       // d_above(l)        N x D
  static constexpr int DEN = DAB + N * D;       // lu(denominator(l)) N x N
  static constexpr int ABE = DEN + N * N;       // a_below(l+1)      N x N (canopy part)
  static constexpr int DBE = ABE + N * N;       // d_below(l+1)      N x D (canopy part)
  static constexpr int NELEM = DBE + N * D;
};

template <int NEL>
SS_HD void rec_load(const double* rec, long stride, int off, double* dst) {
  SS_UNROLL_FULL
  for (int k = 0; k < NEL; ++k) dst[k] = rec[(long)(off + k) * stride];
}
template <int NEL>
SS_HD void rec_store(double* rec, long stride, int off, const double* src) {
  SS_UNROLL_FULL
  for (int k = 0; k < NEL; ++k) rec[(long)(off + k) * stride] = src[k];
}

// Copy the (RS x CS) sub-block at (I0, J0) of an (RF x CF) matrix
template <int RF, int RS, int CS, int I0, int J0>
SS_HD void get_block(const double* A, double* B) {
  SS_UNROLL_FULL
  for (int j = 0; j < CS; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < RS; ++i) SS_E(B, i, j, RS) = SS_E(A, I0 + i, J0 + j, RF);
  }
}

// Region extinction and single-scattering albedo (radsurf_forest_sw.F90:277-297)
template <int NREG>
SS_HD void region_optics(double air_ext, double air_ssa, double veg_ext, double veg_ssa,
                         const double* od, double* ext_reg, double* ssa_reg) {
  ext_reg[0] = air_ext;
  ssa_reg[0] = air_ssa;
  if (NREG == 2) {
    ext_reg[1] = air_ext + veg_ext;
    ssa_reg[1] = (ext_reg[0] * ssa_reg[0] + veg_ext * veg_ssa) / ss_max(ext_reg[1], 1.0e-8);
  }
  if (NREG == 3) {
    SS_UNROLL_FULL
    for (int r = 1; r < NREG; ++r) {
      ext_reg[r] = air_ext + od[r] * veg_ext;
      ssa_reg[r] = (ext_reg[0] * ssa_reg[0] + od[r] * veg_ext * veg_ssa) / ss_max(ext_reg[r], 1.0e-8);
    }
  }
}

template <bool URBAN, int NREG, int NS>
SS_HD void layer_unit_sw(const Launch& L, int cu, int jlay) {
  constexpr int N = NREG * NS, D = NREG;
  using Rec = SwRec<N, D>;
  const Inputs& in = L.in;
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  if (jlay >= in.nlay[col]) return;
  const int lay = in.istartlay[col] - 1 + jlay;
  const long sidx = (long)jb + (long)L.nspec * lay;

  const double cos_in = in.cos_sza[col];
  const double cos_sza = URBAN ? ss_max(cos_in, 1.0e-6) : cos_in;
  double sin0 = 0.0, tan0;
  if (URBAN) {
    sin0 = sqrt(1.0 - cos_sza * cos_sza);
    tan0 = sin0 / cos_sza;
  } else {
    tan0 = sqrt(1.0 - cos_sza * cos_sza) / ss_max(cos_sza, 1.0e-6);
  }

  const LayerGeom g = load_layer_geom<URBAN, NREG>(in, lay);
  double frac[NREG], np[NREG], npw[NREG], fx[NREG * NREG], od[NREG], f_wall[NREG];
  region_fractions<URBAN, false, NREG>(g.veg_fraction, g.building_fraction, frac);
  norm_perimeters<URBAN, NREG>(L.cfg, g, np, npw);
  exchange_rates<NREG>(frac, np, L.cfg.min_vegetation_fraction, fx);
  od_scaling<NREG>(g.veg_fsd, od);

  double wall_ext = 0.0, wall_factor = 0.0;
  SS_UNROLL_FULL
  for (int r = 0; r < NREG; ++r) f_wall[r] = 0.0;
  if (URBAN) {
    SS_UNROLL_FULL
    for (int r = 0; r < NREG; ++r)
      f_wall[r] = (frac[r] <= L.cfg.min_vegetation_fraction) ? 0.0 : npw[r] / (kPi * frac[r]);
    const double wa = in.wall_albedo[sidx], wsf = in.wall_specular_frac[sidx];
    wall_ext = 1.0 - wa * wsf;
    wall_factor = wa * (1.0 - wsf);
  }

  double ext_reg[NREG], ssa_reg[NREG];
  region_optics<NREG>(in.air_ext[sidx], in.air_ssa[sidx], g.veg_ext,
                      (NREG > 1) ? in.veg_ssa[sidx] : 0.0, od, ext_reg, ssa_reg);

  // ---- Gamma matrices (forest :329-387, urban :424-497) ----
  double g0[D * D], g1[N * N], g2[N * N], g3[N * D];
  mat_zero<D, D>(g0);
  mat_zero<N, N>(g1);
  mat_zero<N, N>(g2);
  mat_zero<N, D>(g3);
  SS_UNROLL_FULL
  for (int jfr = 0; jfr < NREG; ++jfr) {
    SS_UNROLL_FULL
    for (int jto = 0; jto < NREG; ++jto) {
      if (jfr != jto) {
        SS_E(g0, jfr, jfr, D) = SS_E(g0, jfr, jfr, D) - tan0 * SS_E(fx, jto, jfr, NREG);
        SS_E(g0, jto, jfr, D) = tan0 * SS_E(fx, jto, jfr, NREG);
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) {
          const int ifr = js + jfr * NS, ito = js + jto * NS;
          SS_E(g1, ifr, ifr, N) = SS_E(g1, ifr, ifr, N) - L.lg.tan_ang[js] * SS_E(fx, jto, jfr, NREG);
          SS_E(g1, ito, ifr, N) = L.lg.tan_ang[js] * SS_E(fx, jto, jfr, NREG);
        }
      }
    }
  }
  SS_UNROLL_FULL
  for (int jreg = 0; jreg < NREG; ++jreg) {
    if (URBAN)
      SS_E(g0, jreg, jreg, D) = SS_E(g0, jreg, jreg, D) - ext_reg[jreg] / cos_sza
                                - tan0 * f_wall[jreg] * wall_ext;
    else
      SS_E(g0, jreg, jreg, D) = SS_E(g0, jreg, jreg, D) - ext_reg[jreg] / cos_sza;
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) {
      const int ifr = js + jreg * NS;
      if (URBAN)
        SS_E(g1, ifr, ifr, N) = SS_E(g1, ifr, ifr, N) - ext_reg[jreg] / L.lg.mu[js]
                                - L.lg.tan_ang[js] * f_wall[jreg] * wall_ext;
      else
        SS_E(g1, ifr, ifr, N) = SS_E(g1, ifr, ifr, N) - ext_reg[jreg] / L.lg.mu[js];
    }
  }
  SS_UNROLL_FULL
  for (int jsfr = 0; jsfr < NS; ++jsfr) {
    SS_UNROLL_FULL
    for (int jsto = 0; jsto < NS; ++jsto) {
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg) {
        const int ifr = jsfr + jreg * NS, ito = jsto + jreg * NS;
        if (URBAN)
          SS_E(g2, ito, ifr, N) = 0.5 * (L.lg.weight[jsto] * ext_reg[jreg] * ssa_reg[jreg] / L.lg.mu[jsfr]
                                         + L.lg.vweight[jsto] * L.lg.tan_ang[jsfr] * f_wall[jreg] * wall_factor);
        else
          SS_E(g2, ito, ifr, N) = 0.5 * L.lg.weight[jsto] * ext_reg[jreg] * ssa_reg[jreg] / L.lg.mu[jsfr];
      }
    }
  }
  SS_UNROLL_NN(N)
  for (int k = 0; k < N * N; ++k) g1[k] = g1[k] + g2[k];
  SS_UNROLL_FULL
  for (int jreg = 0; jreg < NREG; ++jreg) {
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) {
      const int ito = js + jreg * NS;
      if (URBAN)
        SS_E(g3, ito, jreg, N) = 0.5 * (L.lg.weight[js] * ext_reg[jreg] * ssa_reg[jreg]
                                        + L.lg.vweight[js] * sin0 * f_wall[jreg] * wall_factor);
      else
        SS_E(g3, ito, jreg, N) = 0.5 * L.lg.weight[js] * ext_reg[jreg] * ssa_reg[jreg];
    }
  }

  // ---- layer matrices, 3-way branch (forest :408-455, urban :520-583) ----
  RecWriter w{L.lay + ((long)jlay * Rec::NELEM) * L.ncu + cu, (long)L.ncu};
  int ierr;
  if (NREG == 1 || g.veg_fraction <= L.cfg.min_vegetation_fraction) {
    // clear-sky region only: (1:ns, 1:1)
    double s0[1], s1[NS * NS], s2[NS * NS], s3[NS];
    get_block<D, 1, 1, 0, 0>(g0, s0);
    get_block<N, NS, NS, 0, 0>(g1, s1);
    get_block<N, NS, NS, 0, 0>(g2, s2);
    get_block<N, NS, 1, 0, 0>(g3, s3);
    ierr = calc_matrices_sw<N, D, NS, 1, 0, 0>(s0, s1, s2, s3, g.dz, w);
  } else if (frac[0] <= L.cfg.min_vegetation_fraction) {
    // vegetated regions only: (ns+1:n, 2:nreg)
    constexpr int NV = (NREG > 1) ? (NREG - 1) * NS : NS;
    constexpr int DV = (NREG > 1) ? NREG - 1 : 1;
    constexpr int IV = (NREG > 1) ? NS : 0;
    constexpr int RV = (NREG > 1) ? 1 : 0;
    double s0[DV * DV], s1[NV * NV], s2[NV * NV], s3[NV * DV];
    get_block<D, DV, DV, RV, RV>(g0, s0);
    get_block<N, NV, NV, IV, IV>(g1, s1);
    get_block<N, NV, NV, IV, IV>(g2, s2);
    get_block<N, NV, DV, IV, RV>(g3, s3);
    ierr = calc_matrices_sw<N, D, NV, DV, IV, RV>(s0, s1, s2, s3, g.dz, w);
  } else {
    ierr = calc_matrices_sw<N, D, N, D, 0, 0>(g0, g1, g2, g3, g.dz, w);
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if (ierr != 0 && L.eigen_fail) atomicAdd(L.eigen_fail, 1ULL);
#else
  if (ierr != 0 && L.eigen_fail) *L.eigen_fail += 1ULL;
#endif
}

// ---- overlap operators (U (x) I_NS), (V (x) I_NS) on the canopy + roof blocks ----

// Interface geometry above layer jlay (0-based; jlay = -1 is the ground):
// U (NREG x DB), V (DB x NREG).
template <bool URBAN, bool LW, int NREG>
SS_HD void interface_overlap(const Launch& L, int col, int jl_lower, double* U, double* V) {
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  const Inputs& in = L.in;
  const int nlay = in.nlay[col];
  const int l0 = in.istartlay[col] - 1;
  double fu[NREG], fl[DB];
  // upper layer (jl_lower + 1), or the free atmosphere
  if (jl_lower + 1 >= nlay) {
    fu[0] = 1.0;
    SS_UNROLL_FULL
    for (int r = 1; r < NREG; ++r) fu[r] = 0.0;
  } else {
    const int lay = l0 + jl_lower + 1;
    region_fractions<URBAN, LW, NREG>((NREG > 1) ? in.veg_fraction[lay] : 0.0,
                                      URBAN ? in.building_fraction[lay] : 0.0, fu);
  }
  if (!URBAN) {
    if (jl_lower < 0) {
      fl[0] = 1.0;
      SS_UNROLL_FULL
      for (int r = 1; r < NREG; ++r) fl[r] = 0.0;
    } else {
      const int lay = l0 + jl_lower;
      region_fractions<URBAN, LW, NREG>((NREG > 1) ? in.veg_fraction[lay] : 0.0, 0.0, fl);
    }
    overlap_forest<NREG>(fu, fl, L.cfg.min_vegetation_fraction, U, V);
  } else {
    if (jl_lower < 0) {
      // ground = one "roof" region the size of the non-building part of layer 1
      double s = 0.0;
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) { fl[r] = 0.0; s += fu[r]; }
      fl[DB - 1] = s;
    } else {
      const int lay = l0 + jl_lower;
      double flo[NREG], sum_up = 0.0;
      region_fractions<URBAN, LW, NREG>((NREG > 1) ? in.veg_fraction[lay] : 0.0,
                                        in.building_fraction[lay], flo);
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) sum_up += fu[r];
      urban_frac_lower<NREG>(flo, sum_up, jl_lower + 1 >= nlay, fl);
    }
    overlap_urban<NREG>(fu, fl, L.cfg.min_vegetation_fraction, U, V);
  }
}

// y (DB*NS) = (V (x) I) x (NREG*NS)      rect_expandedmat_x_vec :608
template <int NREG, int DB, int NS>
SS_HD void apply_V_vec(const double* V, const double* x, double* y) {
  SS_UNROLL_FULL
  for (int jl = 0; jl < DB; ++jl) {
    SS_UNROLL_FULL
    for (int s = 0; s < NS; ++s) {
      double acc = 0.0;
      SS_UNROLL_FULL
      for (int ju = 0; ju < NREG; ++ju) acc += SS_E(V, jl, ju, DB) * x[ju * NS + s];
      y[jl * NS + s] = acc;
    }
  }
}

// Anew (N x C) = (U (x) I) [ Bc (N x C) ; Broof (NS x C) ]   rect_expandedmat_x_mat :505
template <int NREG, int DB, int NS, int C>
SS_HD void apply_U_mat(const double* U, const double* Bc, const double* Broof, double* Anew) {
  constexpr int N = NREG * NS;
  SS_UNROLL_N(C)
  for (int c = 0; c < C; ++c) {
    SS_UNROLL_FULL
    for (int ju = 0; ju < NREG; ++ju) {
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) {
        double acc = 0.0;
        SS_UNROLL_FULL
        for (int jl = 0; jl < NREG; ++jl) acc += SS_E(U, ju, jl, NREG) * SS_E(Bc, jl * NS + s, c, N);
        if (DB > NREG) acc += SS_E(U, ju, DB - 1, NREG) * SS_E(Broof, s, c, NS);
        SS_E(Anew, ju * NS + s, c, N) = acc;
      }
    }
  }
}

template <bool URBAN, int NREG, int NS>
SS_HD void column_unit_sw(const Launch& L, int cu) {
  constexpr int N = NREG * NS, D = NREG;
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  constexpr int NB = DB * NS;
  using LRec = SwRec<N, D>;
  using CRec = SwColRec<N, D>;
  const Inputs& in = L.in;
  const LG& lg = L.lg;
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  const int nspec = L.nspec;
  const int nlay = in.nlay[col];
  const int l0 = in.istartlay[col] - 1;
  const long stride = L.ncu;
  const long cidx = (long)jb + (long)nspec * col;

  const double cos_in = in.cos_sza[col];
  const double cos_sza = URBAN ? ss_max(cos_in, 1.0e-6) : cos_in;
  const double sin0 = URBAN ? sqrt(1.0 - cos_sza * cos_sza) : 0.0;

  // most transparent band (forest :253, urban :310): minloc of the canopy
  // optical depth - the only cross-band quantity; recomputed by every lane
  int itransp = 0;
  {
    double best = 0.0;
    for (int b = 0; b < nspec; ++b) {
      double odsum = 0.0;
      for (int l = 0; l < nlay; ++l) odsum += in.air_ext[(long)b + (long)nspec * (l0 + l)] * in.dz[l0 + l];
      if (b == 0 || odsum < best) { best = odsum; itransp = b; }
    }
  }

  // ---- section 4: adding method upward ----
  double Aab[N * N], Dab[N * D];
  mat_zero<N, N>(Aab);
  mat_zero<N, D>(Dab);
  {
    const double ga = in.ground_albedo[cidx], gad = in.ground_albedo_dir[cidx];
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

  for (int jlay = 0; jlay < nlay; ++jlay) {
    const double* lrec = L.lay + ((long)jlay * LRec::NELEM) * stride + cu;
    double* crec = L.col + ((long)jlay * CRec::NELEM) * stride + cu;
    const long sidx = (long)jb + (long)nspec * (l0 + jlay);
    double Abe[N * N], Dbe[N * D];
    {
      double R[N * N], T[N * N], Den[N * N], X[N * N];
      rec_load<N * N>(lrec, stride, LRec::R, R);
      rec_load<N * N>(lrec, stride, LRec::T, T);
      rec_store<N * N>(crec, stride, CRec::AAB, Aab);
      rec_store<N * D>(crec, stride, CRec::DAB, Dab);
      // denominator = I - a_above R
      mat_mul<N, N, N>(Aab, R, Den);
      SS_UNROLL_NN(N)
      for (int k = 0; k < N * N; ++k) Den[k] = -Den[k];
      SS_UNROLL_N(N)
      for (int j = 0; j < N; ++j) SS_E(Den, j, j, N) = 1.0 + SS_E(Den, j, j, N);
      lu_factor<N>(Den);
      rec_store<N * N>(crec, stride, CRec::DEN, Den);
      // a_below = R + T Den^-1 (a_above T)
      mat_mul<N, N, N>(Aab, T, X);
      lu_solve_mat<N, N>(Den, X);
      mat_mul<N, N, N>(T, X, Abe);
      SS_UNROLL_NN(N)
      for (int k = 0; k < N * N; ++k) Abe[k] = R[k] + Abe[k];
      rec_store<N * N>(crec, stride, CRec::ABE, Abe);
      // d_below = S_up + T Den^-1 (d_above E + a_above S_dn)
      double E[D * D], S[N * D], Y[N * D];
      rec_load<D * D>(lrec, stride, LRec::EDIR, E);
      rec_load<N * D>(lrec, stride, LRec::SDN, S);
      mat_mul<N, D, D>(Dab, E, Y);
      mat_mul_acc<N, N, D>(Aab, S, Y);
      lu_solve_mat<N, D>(Den, Y);
      rec_load<N * D>(lrec, stride, LRec::SUP, S);
      mat_mul<N, N, D>(T, Y, Dbe);
      SS_UNROLL_FULL
      for (int k = 0; k < N * D; ++k) Dbe[k] = S[k] + Dbe[k];
      rec_store<N * D>(crec, stride, CRec::DBE, Dbe);
    }
    // overlap to the base of the layer above (forest :495-503, urban :645-653)
    {
      double U[NREG * DB], V[DB * NREG];
      interface_overlap<URBAN, false, NREG>(L, col, jlay, U, V);
      double roof_a = 0.0, roof_d = 0.0;
      if (URBAN) {
        roof_a = in.roof_albedo[sidx];
        roof_d = cos_sza * in.roof_albedo_dir[sidx];
      }
      // tA = a_below (V (x) I): canopy rows (N x N) and roof rows (NS x N)
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
  }

  // top-of-canopy albedos (forest :520-528, urban :666-674)
  double top_albedo_diff, top_albedo_dir;
  {
    double s = 0.0;
    SS_UNROLL_FULL
    for (int j1 = 0; j1 < NS; ++j1) {
      double acc = 0.0;
      SS_UNROLL_FULL
      for (int j2 = 0; j2 < NS; ++j2) acc += SS_E(Aab, j1, j2, N) * lg.hweight[j2];
      s += acc;
    }
    top_albedo_diff = s;
    s = 0.0;
    SS_UNROLL_FULL
    for (int j1 = 0; j1 < NS; ++j1) s += SS_E(Dab, j1, 0, N);
    top_albedo_dir = s / cos_sza;
    L.top1[cidx] = top_albedo_diff;
    L.top2[cidx] = top_albedo_dir;
  }

  const FluxDev& fa = L.flux_a;   // sw_norm_dir
  const FluxDev& fb = L.flux_b;   // sw_norm_diff
#define PUTC(fd, F, v) do { if ((fd).f[F]) (fd).f[F][cidx] = (v); } while (0)
#define PUTL(fd, F, v) do { if ((fd).f[F]) (fd).f[F][sidx] = (v); } while (0)

  // ---- section 5a: fluxes normalized by the direct flux at canopy top ----
  {
    double dir_above[D], diff_above[N], up_above[N];
    dir_above[0] = 1.0 / cos_sza;
    SS_UNROLL_FULL
    for (int r = 1; r < D; ++r) dir_above[r] = 0.0;
    vec_zero<N>(diff_above);
    vec_zero<N>(up_above);
    PUTC(fa, F_TOP_DN_DIR, 1.0);
    PUTC(fa, F_TOP_DN, 1.0);
    PUTC(fa, F_TOP_NET, 1.0 * (1.0 - top_albedo_dir));
    double flux_dn_dir_clear = 1.0 / cos_sza;
    double gvd = 0.0;

    for (int jlay = nlay - 1; jlay >= 0; --jlay) {
      const double* lrec = L.lay + ((long)jlay * LRec::NELEM) * stride + cu;
      const double* crec = L.col + ((long)jlay * CRec::NELEM) * stride + cu;
      const int lay = l0 + jlay;
      const long sidx = (long)jb + (long)nspec * lay;
      const LayerGeom g = load_layer_geom<URBAN, NREG>(in, lay);
      double U[NREG * DB], V[DB * NREG];
      interface_overlap<URBAN, false, NREG>(L, col, jlay, U, V);

      // across the interface at the top of the layer
      double dir_below[DB], diff_below[NB], up_below[NB];
      SS_UNROLL_FULL
      for (int jl = 0; jl < DB; ++jl) {
        double acc = 0.0;
        SS_UNROLL_FULL
        for (int ju = 0; ju < NREG; ++ju) acc += SS_E(V, jl, ju, DB) * dir_above[ju];
        dir_below[jl] = acc;
      }
      apply_V_vec<NREG, DB, NS>(V, diff_above, diff_below);
      {
        double M[N * N];
        rec_load<N * N>(crec, stride, CRec::ABE, M);
        mat_vec<N, N>(M, diff_below, up_below);
        double Dm[N * D];
        rec_load<N * D>(crec, stride, CRec::DBE, Dm);
        mat_vec_acc<N, D>(Dm, dir_below, up_below);
      }
      if (URBAN) {
        // exposed roof (urban :717-721): a_below / d_below roof blocks
        const double roof_a = in.roof_albedo[sidx];
        const double roof_d = cos_sza * in.roof_albedo_dir[sidx];
        double sdn = 0.0;
        SS_UNROLL_FULL
        for (int s = 0; s < NS; ++s) sdn += diff_below[N + s];
        double sup = 0.0;
        SS_UNROLL_FULL
        for (int s = 0; s < NS; ++s) {
          // row n+s of a_below is roof_albedo*hweight(s) in every roof column
          double acc = 0.0;
          SS_UNROLL_FULL
          for (int s2 = 0; s2 < NS; ++s2) acc += (roof_a * lg.hweight[s]) * diff_below[N + s2];
          up_below[N + s] = acc + (roof_d * lg.hweight[s]) * dir_below[DB - 1];
          sup += up_below[N + s];
        }
        const double roof_in_dir = cos_sza * dir_below[DB - 1];
        PUTL(fa, F_ROOF_IN_DIR, roof_in_dir);
        PUTL(fa, F_ROOF_IN, roof_in_dir + sdn);
        PUTL(fa, F_ROOF_NET, (roof_in_dir + sdn) - sup);
        if (jb == itransp && fa.f[F_ROOF_SUNLIT_FRAC]) {
          // urban :805-812
          const double nbf_up = (jlay + 1 < nlay) ? 1.0 - in.building_fraction[lay + 1] : 1.0;
          const double nbf = 1.0 - g.building_fraction;
          double roof_fraction;
          if (jlay == nlay - 1) roof_fraction = g.building_fraction;
          else roof_fraction = ss_max(0.0, g.building_fraction - in.building_fraction[lay + 1]);
          fa.f[F_ROOF_SUNLIT_FRAC][lay] = roof_in_dir * nbf_up
              / (cos_sza * flux_dn_dir_clear * ss_max(L.cfg.min_building_fraction, roof_fraction));
          flux_dn_dir_clear = flux_dn_dir_clear * nbf / nbf_up;
        }
      }

      // fluxes at the base of the layer
      double refl_dir[N], rhs[N];
      {
        double E[D * D];
        rec_load<D * D>(lrec, stride, LRec::EDIR, E);
        double t[D];
        mat_vec<D, D>(E, dir_below, t);
        double dir_conv[D];   // dir_below - dir_above (new), needed below
        SS_UNROLL_FULL
        for (int r = 0; r < D; ++r) { dir_above[r] = t[r]; dir_conv[r] = dir_below[r] - t[r]; }
        double Dm[N * D];
        rec_load<N * D>(crec, stride, CRec::DAB, Dm);
        mat_vec<N, D>(Dm, dir_above, refl_dir);
        double M[N * N];
        rec_load<N * N>(lrec, stride, LRec::T, M);
        mat_vec<N, N>(M, diff_below, rhs);
        rec_load<N * N>(lrec, stride, LRec::R, M);
        double t2[N];
        mat_vec<N, N>(M, refl_dir, t2);
        rec_load<N * D>(lrec, stride, LRec::SDN, Dm);
        double t3[N];
        mat_vec<N, D>(Dm, dir_below, t3);
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) rhs[i] = rhs[i] + t2[i] + t3[i];
        rec_load<N * N>(crec, stride, CRec::DEN, M);
        lu_solve<N>(M, rhs);
        vec_copy<N>(rhs, diff_above);
        rec_load<N * N>(crec, stride, CRec::AAB, M);
        mat_vec<N, N>(M, diff_above, up_above);
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) up_above[i] = up_above[i] + refl_dir[i];

        if (fa.f[F_FLUX_DN_LAYER_TOP]) {
          double s = 0.0;
          SS_UNROLL_FULL
          for (int r = 0; r < NREG; ++r) s += dir_below[r];
          const double ddt = cos_sza * s;
          PUTL(fa, F_FLUX_DN_DIR_LAYER_TOP, ddt);
          s = 0.0;
          SS_UNROLL_N(N)
          for (int i = 0; i < N; ++i) s += diff_below[i];
          PUTL(fa, F_FLUX_DN_LAYER_TOP, ddt + s);
          s = 0.0;
          SS_UNROLL_N(N)
          for (int i = 0; i < N; ++i) s += up_below[i];
          PUTL(fa, F_FLUX_UP_LAYER_TOP, s);
          s = 0.0;
          SS_UNROLL_FULL
          for (int r = 0; r < NREG; ++r) s += dir_above[r];
          const double ddb = cos_sza * s;
          PUTL(fa, F_FLUX_DN_DIR_LAYER_BASE, ddb);
          s = 0.0;
          SS_UNROLL_N(N)
          for (int i = 0; i < N; ++i) s += diff_above[i];
          PUTL(fa, F_FLUX_DN_LAYER_BASE, ddb + s);
          s = 0.0;
          SS_UNROLL_N(N)
          for (int i = 0; i < N; ++i) s += up_above[i];
          PUTL(fa, F_FLUX_UP_LAYER_BASE, s);
        }

        // integrated fluxes (forest :598-606)
        double int_flux_dir[D], int_flux_diff[N], conv[N];
        rec_load<D * D>(lrec, stride, LRec::IDIR, E);
        mat_vec<D, D>(E, dir_conv, int_flux_dir);
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) conv[i] = diff_below[i] - diff_above[i] - up_below[i] + up_above[i];
        rec_load<N * N>(lrec, stride, LRec::IDIFF, M);
        mat_vec<N, N>(M, conv, int_flux_diff);
        rec_load<N * D>(lrec, stride, LRec::IDD, Dm);
        mat_vec<N, D>(Dm, dir_conv, t3);
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) int_flux_diff[i] = int_flux_diff[i] + t3[i];

        // absorption (forest :608-628, urban :758-788)
        const double air_ext = in.air_ext[sidx], air_ssa = in.air_ssa[sidx];
        {
          double s = 0.0;
          SS_UNROLL_FULL
          for (int js = 0; js < NS; ++js) s += int_flux_diff[js] * (1.0 / lg.mu[js]);
          PUTL(fa, F_CLEAR_AIR_ABS, 0.0 + air_ext * (1.0 - air_ssa) * (int_flux_dir[0] + s));
        }
        double veg_abs_dir = 0.0;
        if (NREG > 1) {
          double od[NREG];
          od_scaling<NREG>(g.veg_fsd, od);
          const double vssa = in.veg_ssa[sidx];
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
          PUTL(fa, F_VEG_AIR_ABS, vaa);
          PUTL(fa, F_VEG_ABS_DIR, veg_abs_dir);
          PUTL(fa, F_VEG_ABS, va);
        } else {
          PUTL(fa, F_VEG_AIR_ABS, 0.0);
          PUTL(fa, F_VEG_ABS_DIR, 0.0);
          PUTL(fa, F_VEG_ABS, 0.0);
        }

        double wall_in_dir = 0.0;
        if (URBAN) {
          // walls (urban :790-802)
          double frac[NREG], np[NREG], npw[NREG], f_wall[NREG];
          region_fractions<URBAN, false, NREG>(g.veg_fraction, g.building_fraction, frac);
          norm_perimeters<URBAN, NREG>(L.cfg, g, np, npw);
          double npw_sum = 0.0;
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
          PUTL(fa, F_WALL_IN_DIR, wall_in_dir);
          PUTL(fa, F_WALL_IN, wall_in);
          PUTL(fa, F_WALL_NET, wall_in * (1.0 - in.wall_albedo[sidx]));
          if (jb == itransp) {
            // wall sunlit fraction needs f_wall_dir_clear (urban :405-411)
            const double nbf = 1.0 - g.building_fraction;
            const double f_wall_dir_clear =
                (nbf <= L.cfg.min_building_fraction) ? 0.0 : npw_sum / (kPi * nbf);
            const double ae = in.air_ext[(long)itransp + (long)nspec * lay];
            const double trans_dir_clear = exp(-ae * g.dz / cos_sza);
            const double ifdc = (ae > 0.0) ? flux_dn_dir_clear * (1.0 - trans_dir_clear) * cos_sza / ae
                                           : flux_dn_dir_clear * g.dz;
            if (fa.f[F_WALL_SUNLIT_FRAC])
              fa.f[F_WALL_SUNLIT_FRAC][lay] = 0.5 * wall_in_dir / ss_max(DBL_EPSILON, (f_wall_dir_clear * sin0 * ifdc));
          }
        }

        // vegetation sunlit fraction and the clear-sky direct beam
        // (forest :630-649, urban :814-852); band itransp only
        if (jb == itransp) {
          const double ae = in.air_ext[(long)itransp + (long)nspec * lay];
          const double trans_dir_clear = exp(-ae * g.dz / cos_sza);
          const double ifdc = (ae > 0.0) ? flux_dn_dir_clear * (1.0 - trans_dir_clear) * cos_sza / ae
                                         : flux_dn_dir_clear * g.dz;
          if (NREG > 1 && fa.f[F_VEG_SUNLIT_FRAC]) {
            const double vadc = ifdc * g.veg_ext * (1.0 - in.veg_ssa[(long)itransp + (long)nspec * lay])
                                * g.veg_fraction;
            fa.f[F_VEG_SUNLIT_FRAC][lay] = veg_abs_dir / ss_max(DBL_EPSILON, vadc);
          }
          flux_dn_dir_clear = flux_dn_dir_clear * trans_dir_clear;
        }
      }
    }

    // ground (forest :661-676, urban :861-874)
    {
      double s = 0.0;
      SS_UNROLL_FULL
      for (int r = 0; r < NREG; ++r) s += dir_above[r];
      const double gdd = cos_sza * s;
      double s2 = 0.0, s3 = 0.0;
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) s2 += diff_above[i];
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) s3 += up_above[i];
      PUTC(fa, F_GROUND_DN_DIR, gdd);
      PUTC(fa, F_GROUND_DN, gdd + s2);
      PUTC(fa, F_GROUND_NET, (gdd + s2) - s3);
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg) {
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) {
          const int ifr = js + jreg * NS;
          gvd = gvd + (diff_above[ifr] + up_above[ifr]) * lg.tan_ang[js] / kPi;
        }
      }
      PUTC(fa, F_GROUND_VERTICAL_DIFF, gvd);
      if (jb == itransp && fa.f[F_GROUND_SUNLIT_FRAC])
        fa.f[F_GROUND_SUNLIT_FRAC][col] = gdd / (cos_sza * flux_dn_dir_clear);
    }
  }

  // ---- section 5b: fluxes normalized by the diffuse flux at canopy top ----
  {
    double diff_above[N], up_above[N];
    vec_zero<N>(diff_above);
    vec_zero<N>(up_above);
    SS_UNROLL_FULL
    for (int js = 0; js < NS; ++js) diff_above[js] = lg.hweight[js];
    PUTC(fb, F_TOP_DN_DIR, 0.0);
    PUTC(fb, F_TOP_DN, 1.0);
    PUTC(fb, F_TOP_NET, 1.0 - top_albedo_diff);

    for (int jlay = nlay - 1; jlay >= 0; --jlay) {
      const double* lrec = L.lay + ((long)jlay * LRec::NELEM) * stride + cu;
      const double* crec = L.col + ((long)jlay * CRec::NELEM) * stride + cu;
      const int lay = l0 + jlay;
      const long sidx = (long)jb + (long)nspec * lay;
      const LayerGeom g = load_layer_geom<URBAN, NREG>(in, lay);
      double U[NREG * DB], V[DB * NREG];
      interface_overlap<URBAN, false, NREG>(L, col, jlay, U, V);

      double diff_below[NB], up_below[NB];
      apply_V_vec<NREG, DB, NS>(V, diff_above, diff_below);
      double M[N * N];
      rec_load<N * N>(crec, stride, CRec::ABE, M);
      mat_vec<N, N>(M, diff_below, up_below);
      if (URBAN) {
        const double roof_a = in.roof_albedo[sidx];
        double sdn = 0.0, sup = 0.0;
        SS_UNROLL_FULL
        for (int s = 0; s < NS; ++s) sdn += diff_below[N + s];
        SS_UNROLL_FULL
        for (int s = 0; s < NS; ++s) {
          double acc = 0.0;
          SS_UNROLL_FULL
          for (int s2 = 0; s2 < NS; ++s2) acc += (roof_a * lg.hweight[s]) * diff_below[N + s2];
          up_below[N + s] = acc;
          sup += acc;
        }
        PUTL(fb, F_ROOF_IN, sdn);
        PUTL(fb, F_ROOF_NET, sdn - sup);
        PUTL(fb, F_ROOF_IN_DIR, 0.0);
      }
      double rhs[N];
      rec_load<N * N>(lrec, stride, LRec::T, M);
      mat_vec<N, N>(M, diff_below, rhs);
      rec_load<N * N>(crec, stride, CRec::DEN, M);
      lu_solve<N>(M, rhs);
      vec_copy<N>(rhs, diff_above);
      rec_load<N * N>(crec, stride, CRec::AAB, M);
      mat_vec<N, N>(M, diff_above, up_above);

      if (fb.f[F_FLUX_DN_LAYER_TOP]) {
        double s = 0.0;
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) s += diff_below[i];
        PUTL(fb, F_FLUX_DN_LAYER_TOP, s);
        s = 0.0;
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) s += up_below[i];
        PUTL(fb, F_FLUX_UP_LAYER_TOP, s);
        s = 0.0;
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) s += diff_above[i];
        PUTL(fb, F_FLUX_DN_LAYER_BASE, s);
        s = 0.0;
        SS_UNROLL_N(N)
        for (int i = 0; i < N; ++i) s += up_above[i];
        PUTL(fb, F_FLUX_UP_LAYER_BASE, s);
        PUTL(fb, F_FLUX_DN_DIR_LAYER_TOP, 0.0);
        PUTL(fb, F_FLUX_DN_DIR_LAYER_BASE, 0.0);
      }

      double conv[N], int_flux_diff[N];
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) conv[i] = diff_below[i] - diff_above[i] - up_below[i] + up_above[i];
      rec_load<N * N>(lrec, stride, LRec::IDIFF, M);
      mat_vec<N, N>(M, conv, int_flux_diff);

      const double air_ext = in.air_ext[sidx], air_ssa = in.air_ssa[sidx];
      {
        double s = 0.0;
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) s += int_flux_diff[js] * (1.0 / lg.mu[js]);
        PUTL(fb, F_CLEAR_AIR_ABS, 0.0 + air_ext * (1.0 - air_ssa) * s);
      }
      if (NREG > 1) {
        double od[NREG];
        od_scaling<NREG>(g.veg_fsd, od);
        const double vssa = in.veg_ssa[sidx];
        double vaa = 0.0, va = 0.0;
        SS_UNROLL_FULL
        for (int jreg = 1; jreg < NREG; ++jreg) {
          double s = 0.0;
          SS_UNROLL_FULL
          for (int js = 0; js < NS; ++js) s += int_flux_diff[jreg * NS + js] * (1.0 / lg.mu[js]);
          vaa = vaa + air_ext * (1.0 - air_ssa) * s;
          va = va + g.veg_ext * (1.0 - vssa) * s * od[jreg];
        }
        PUTL(fb, F_VEG_AIR_ABS, vaa);
        PUTL(fb, F_VEG_ABS, va);
      } else {
        PUTL(fb, F_VEG_AIR_ABS, 0.0);
        PUTL(fb, F_VEG_ABS, 0.0);
      }
      PUTL(fb, F_VEG_ABS_DIR, 0.0);
      if (jb == itransp) {
        if (fb.f[F_VEG_SUNLIT_FRAC]) fb.f[F_VEG_SUNLIT_FRAC][lay] = 0.0;
        if (fb.f[F_ROOF_SUNLIT_FRAC]) fb.f[F_ROOF_SUNLIT_FRAC][lay] = 0.0;
        if (fb.f[F_WALL_SUNLIT_FRAC]) fb.f[F_WALL_SUNLIT_FRAC][lay] = 0.0;
      }
      if (URBAN) {
        double frac[NREG], np[NREG], npw[NREG];
        region_fractions<URBAN, false, NREG>(g.veg_fraction, g.building_fraction, frac);
        norm_perimeters<URBAN, NREG>(L.cfg, g, np, npw);
        double wall_in = 0.0;
        SS_UNROLL_FULL
        for (int jreg = 0; jreg < NREG; ++jreg) {
          const double f_wall = (frac[jreg] <= L.cfg.min_vegetation_fraction) ? 0.0 : npw[jreg] / (kPi * frac[jreg]);
          double s = 0.0;
          SS_UNROLL_FULL
          for (int js = 0; js < NS; ++js) s += int_flux_diff[jreg * NS + js] * lg.tan_ang[js];
          wall_in = wall_in + f_wall * (s);
        }
        PUTL(fb, F_WALL_IN, wall_in);
        PUTL(fb, F_WALL_NET, wall_in * (1.0 - in.wall_albedo[sidx]));
        PUTL(fb, F_WALL_IN_DIR, 0.0);
      } else {
        // members the reference leaves at zero after zero() for forest tiles
        PUTL(fa, F_ROOF_IN, 0.0); PUTL(fa, F_ROOF_IN_DIR, 0.0); PUTL(fa, F_ROOF_NET, 0.0);
        PUTL(fa, F_WALL_IN, 0.0); PUTL(fa, F_WALL_IN_DIR, 0.0); PUTL(fa, F_WALL_NET, 0.0);
        PUTL(fb, F_ROOF_IN, 0.0); PUTL(fb, F_ROOF_IN_DIR, 0.0); PUTL(fb, F_ROOF_NET, 0.0);
        PUTL(fb, F_WALL_IN, 0.0); PUTL(fb, F_WALL_IN_DIR, 0.0); PUTL(fb, F_WALL_NET, 0.0);
        if (jb == itransp) {
          if (fa.f[F_ROOF_SUNLIT_FRAC]) fa.f[F_ROOF_SUNLIT_FRAC][lay] = 0.0;
          if (fa.f[F_WALL_SUNLIT_FRAC]) fa.f[F_WALL_SUNLIT_FRAC][lay] = 0.0;
        }
      }
      if (NREG == 1 && jb == itransp && fa.f[F_VEG_SUNLIT_FRAC]) fa.f[F_VEG_SUNLIT_FRAC][lay] = 0.0;
    }
    {
      double s2 = 0.0, s3 = 0.0, gvd = 0.0;
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) s2 += diff_above[i];
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) s3 += up_above[i];
      PUTC(fb, F_GROUND_DN_DIR, 0.0);
      PUTC(fb, F_GROUND_DN, s2);
      PUTC(fb, F_GROUND_NET, s2 - s3);
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg) {
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) {
          const int ifr = js + jreg * NS;
          gvd = gvd + (diff_above[ifr] + up_above[ifr]) * lg.tan_ang[js] / kPi;
        }
      }
      PUTC(fb, F_GROUND_VERTICAL_DIFF, gvd);
      if (jb == itransp && fb.f[F_GROUND_SUNLIT_FRAC]) fb.f[F_GROUND_SUNLIT_FRAC][col] = 0.0;
    }
  }
#undef PUTC
#undef PUTL
}

}  // namespace ss
