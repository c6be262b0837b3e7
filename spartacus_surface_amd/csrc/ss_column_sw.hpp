// ss_column_sw.hpp - shortwave column unit: one (column, band) lane does the
// adding method upward (section 4 of spartacus_forest_sw,
// radsurf_forest_sw.F90:463-528 / spartacus_urban_sw, radsurf_urban_sw.F90:591-674)
// and the two downward flux passes (section 5, forest :530-760 / urban
// :676-984).  The adding-method matrices of every layer (a_above, d_above,
// LU of the denominator, a_below, d_below) go through the column scratch
// [layer][elem][cu].
//
// Measured alternatives (profiles/round1_column_variants.md): recomputing
// denominator / a_below / d_below in a single merged downward sweep loads
// fewer doubles but needs > 256 live fp64 values per lane at n = 6 and spills
// more than it saves (sw_column 18 -> 27..37 ms per 1e6 columns); the stored
// form below is the fastest measured for the shortwave.
#pragma once

#include "ss_common.hpp"
#include "ss_geometry.hpp"
#include "ss_layer_matrices.hpp"
#include "ss_linalg.hpp"
#include "ss_solver_sw.hpp"

namespace ss {

// Adding-method record of one layer in the column scratch [layer][elem][cu]
template <int N, int D>
struct SwColRec {
  static constexpr int AAB = 0;                           // a_above(l)        N x N
  static constexpr int DAB = AAB + rec_even(N * N);       // d_above(l)        N x D
  static constexpr int DEN = DAB + rec_even(N * D);       // lu(denominator(l)) N x N
  static constexpr int ABE = DEN + rec_even(N * N);       // a_below(l+1)      N x N (canopy part)
  static constexpr int DBE = ABE + rec_even(N * N);       // d_below(l+1)      N x D (canopy part)
  static constexpr int NELEM = DBE + rec_even(N * D);
};

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
  constexpr long stride = kRecStride;
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
    const double* lrec = L.lay + rec_base(jlay, L.maxlay, LRec::NELEM, cu);
    double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
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
      const double* lrec = L.lay + rec_base(jlay, L.maxlay, LRec::NELEM, cu);
      const double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
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
      const double* lrec = L.lay + rec_base(jlay, L.maxlay, LRec::NELEM, cu);
      const double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
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
