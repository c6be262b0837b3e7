// ss_column_lw.hpp - longwave column unit: adding method with sources (forest
// radsurf_forest_lw.F90:452-506, urban radsurf_urban_lw.F90:551-626) and the
// internal-emission and incoming-flux passes (forest :508-694, urban :628-858)
// of one (column, band), both passes in ONE downward sweep.  Only a_above(l),
// source_above(l) are stored per layer; the sweep recomputes LU(denominator),
// a_below and source_below from them and from the layer record it loads anyway
// (bit-identical to the upward values) - measured 17.9 -> 13.6 ms per 1e6
// columns on MI355X against storing them (profiles/round1_column_variants.md).
#pragma once

#include "ss_common.hpp"
#include "ss_geometry.hpp"
#include "ss_layer_matrices.hpp"
#include "ss_linalg.hpp"
#include "ss_solver_lw.hpp"

namespace ss {

template <int N>
struct LwColRec {
  static constexpr int AAB = 0;                       // a_above(l)       N x N
  static constexpr int SAB = AAB + rec_even(N * N);   // source_above(l)  N
  static constexpr int NELEM = SAB + rec_even(N);
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

template <bool URBAN, int NREG, int NS>
SS_HD void column_unit_lw(const Launch& L, int cu) {
  constexpr int N = NREG * NS;
  constexpr int DB = URBAN ? NREG + 1 : NREG;
  constexpr int NB = DB * NS;
  using LRec = LwRec<N>;
  using CRec = LwColRec<N>;
  const Inputs& in = L.in;
  const LG& lg = L.lg;
  const int col = L.cols[cu / L.nspec];
  const int jb = cu % L.nspec;
  const int nspec = L.nspec;
  const int nlay = in.nlay[col];
  const int l0 = in.istartlay[col] - 1;
  constexpr long stride = kRecStride;
  const long cidx = (long)jb + (long)nspec * col;

  // ---- section 4: adding method with sources ----
  double Aab[N * N], Sab[N];
  mat_zero<N, N>(Aab);
  {
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

  for (int jlay = 0; jlay < nlay; ++jlay) {
    const double* lrec = L.lay + rec_base(jlay, L.maxlay, LRec::NELEM, cu);
    double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    const int lay = l0 + jlay;
    const long sidx = (long)jb + (long)nspec * lay;
    double Abe[N * N], Sbe[N];
    {
      double R[N * N], T[N * N], Den[N * N], Sl[N];
      rec_load<N * N>(lrec, stride, LRec::R, R);
      rec_load<N * N>(lrec, stride, LRec::T, T);
      rec_load<N>(lrec, stride, LRec::SRC, Sl);
      rec_store<N * N>(crec, stride, CRec::AAB, Aab);
      rec_store<N>(crec, stride, CRec::SAB, Sab);
      lw_adding_step<N>(Aab, Sab, R, T, Sl, Den, Abe, Sbe);
    }
    {
      double U[NREG * DB], V[DB * NREG];
      interface_overlap<URBAN, true, NREG>(L, col, jlay, U, V);
      double roof_a = 0.0, roof_emis = 0.0, exposed_roof_frac = 0.0;
      if (URBAN) {
        // urban :587-605
        if (jlay < nlay - 1)
          exposed_roof_frac = ss_max(0.0, in.building_fraction[lay] - in.building_fraction[lay + 1]);
        else
          exposed_roof_frac = in.building_fraction[lay];
        roof_a = 1.0 - in.roof_emissivity[sidx];
        roof_emis = in.roof_emission[sidx];
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
  }

  double top_emissivity, top_emission;
  {
    double s = 0.0;
    SS_UNROLL_FULL
    for (int j1 = 0; j1 < NS; ++j1) {
      double acc = 0.0;
      SS_UNROLL_FULL
      for (int j2 = 0; j2 < NS; ++j2) acc += SS_E(Aab, j1, j2, N) * lg.hweight[j2];
      s += acc;
    }
    top_emissivity = 1.0 - s;
    s = 0.0;
    SS_UNROLL_FULL
    for (int j1 = 0; j1 < NS; ++j1) s += Sab[j1];
    top_emission = s;
    L.top1[cidx] = top_emissivity;
    L.top2[cidx] = top_emission;
  }

  const FluxDev& fa = L.flux_a;   // lw_internal
  const FluxDev& fb = L.flux_b;   // lw_norm
#define PUTC(fd, F, v) do { if ((fd).f[F]) (fd).f[F][cidx] = (v); } while (0)
#define PUTL(fd, F, v) do { if ((fd).f[F]) (fd).f[F][sidx] = (v); } while (0)

  // ---- section 5: "a" = internal emission, "b" = incoming at canopy top ----
  double dna_above[N], upa_above[N], dnb_above[N], upb_above[N];
  vec_zero<N>(dna_above);
  vec_zero<N>(upa_above);
  vec_zero<N>(dnb_above);
  vec_zero<N>(upb_above);
  SS_UNROLL_FULL
  for (int js = 0; js < NS; ++js) dnb_above[js] = lg.hweight[js];
  PUTC(fa, F_TOP_DN, 0.0);
  PUTC(fa, F_TOP_NET, -top_emission);
  PUTC(fb, F_TOP_DN, 1.0);
  PUTC(fb, F_TOP_NET, top_emissivity);

  double emiss_factor = 0.0;
  SS_UNROLL_FULL
  for (int js = 0; js < NS; ++js) emiss_factor += lg.hweight[js] / lg.mu[js];
  emiss_factor = 2.0 * emiss_factor;

  for (int jlay = nlay - 1; jlay >= 0; --jlay) {
    const double* lrec = L.lay + rec_base(jlay, L.maxlay, LRec::NELEM, cu);
    const double* crec = L.col + rec_base(jlay, L.maxlay, CRec::NELEM, cu);
    const int lay = l0 + jlay;
    const long sidx = (long)jb + (long)nspec * lay;
    LwLayerProps<URBAN, NREG> p;
    lw_layer_props<URBAN, NREG>(L, lay, sidx, p);
    double U[NREG * DB], V[DB * NREG];
    interface_overlap<URBAN, true, NREG>(L, col, jlay, U, V);

    double dna_below[NB], dnb_below[NB], upa_below[NB], upb_below[NB];
    apply_V_vec<NREG, DB, NS>(V, dna_above, dna_below);
    apply_V_vec<NREG, DB, NS>(V, dnb_above, dnb_below);

    // LU(denominator), a_below and source_below recomputed with the arithmetic
    // of the upward sweep (bit-identical), see the header comment
    double R[N * N], T[N * N], Den[N * N], AabL[N * N], Sl[N], SabL[N];
    rec_load<N * N>(lrec, stride, LRec::R, R);
    rec_load<N * N>(lrec, stride, LRec::T, T);
    rec_load<N>(lrec, stride, LRec::SRC, Sl);
    rec_load<N * N>(crec, stride, CRec::AAB, AabL);
    rec_load<N>(crec, stride, CRec::SAB, SabL);
    {
      double Abe[N * N], Sbe[N];
      lw_adding_step<N>(AabL, SabL, R, T, Sl, Den, Abe, Sbe);
      mat_vec<N, N>(Abe, dna_below, upa_below);
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) upa_below[i] = upa_below[i] + Sbe[i];
      mat_vec<N, N>(Abe, dnb_below, upb_below);
    }
    if (URBAN) {
      const double roof_a = 1.0 - in.roof_emissivity[sidx];
      double exposed_roof_frac;
      if (jlay < nlay - 1)
        exposed_roof_frac = ss_max(0.0, in.building_fraction[lay] - in.building_fraction[lay + 1]);
      else
        exposed_roof_frac = in.building_fraction[lay];
      double sdna = 0.0, supa = 0.0, sdnb = 0.0, supb = 0.0;
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) { sdna += dna_below[N + s]; sdnb += dnb_below[N + s]; }
      SS_UNROLL_FULL
      for (int s = 0; s < NS; ++s) {
        double acca = 0.0, accb = 0.0;
        SS_UNROLL_FULL
        for (int s2 = 0; s2 < NS; ++s2) {
          acca += (roof_a * lg.hweight[s]) * dna_below[N + s2];
          accb += (roof_a * lg.hweight[s]) * dnb_below[N + s2];
        }
        acca = acca + lg.hweight[s] * in.roof_emission[sidx] * exposed_roof_frac;
        upa_below[N + s] = acca;
        upb_below[N + s] = accb;
        supa += acca;
        supb += accb;
      }
      PUTL(fa, F_ROOF_IN, sdna);
      PUTL(fa, F_ROOF_NET, sdna - supa);
      PUTL(fb, F_ROOF_IN, sdnb);
      PUTL(fb, F_ROOF_NET, sdnb - supb);
    } else {
      PUTL(fa, F_ROOF_IN, 0.0);
      PUTL(fa, F_ROOF_NET, 0.0);
      PUTL(fb, F_ROOF_IN, 0.0);
      PUTL(fb, F_ROOF_NET, 0.0);
    }

    {
      // fluxes at the base of the layer (forest :547-553, :633-636)
      double rhs[N], t2[N];
      mat_vec<N, N>(T, dna_below, rhs);
      mat_vec<N, N>(R, SabL, t2);
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) rhs[i] = (rhs[i] + t2[i] + Sl[i]);
      lu_solve<N>(Den, rhs);
      vec_copy<N>(rhs, dna_above);
      mat_vec<N, N>(AabL, dna_above, upa_above);
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) upa_above[i] = upa_above[i] + SabL[i];
      mat_vec<N, N>(T, dnb_below, rhs);
      lu_solve<N>(Den, rhs);
      vec_copy<N>(rhs, dnb_above);
      mat_vec<N, N>(AabL, dnb_above, upb_above);
    }

    if (fa.f[F_FLUX_DN_LAYER_TOP]) {
      double s = 0.0;
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) s += dna_below[i];
      PUTL(fa, F_FLUX_DN_LAYER_TOP, s);
      s = 0.0;
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) s += upa_below[i];
      PUTL(fa, F_FLUX_UP_LAYER_TOP, s);
      s = 0.0;
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) s += dna_above[i];
      PUTL(fa, F_FLUX_DN_LAYER_BASE, s);
      s = 0.0;
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) s += upa_above[i];
      PUTL(fa, F_FLUX_UP_LAYER_BASE, s);
    }
    if (fb.f[F_FLUX_DN_LAYER_TOP]) {
      double s = 0.0;
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) s += dnb_below[i];
      PUTL(fb, F_FLUX_DN_LAYER_TOP, s);
      s = 0.0;
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) s += upb_below[i];
      PUTL(fb, F_FLUX_UP_LAYER_TOP, s);
      s = 0.0;
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) s += dnb_above[i];
      PUTL(fb, F_FLUX_DN_LAYER_BASE, s);
      s = 0.0;
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) s += upb_above[i];
      PUTL(fb, F_FLUX_UP_LAYER_BASE, s);
    }

    double ifa[N], ifb[N];
    {
      double M[N * N], conv[N], Is[N];
      rec_load<N * N>(lrec, stride, LRec::IFLUX, M);
      rec_load<N>(lrec, stride, LRec::ISRC, Is);
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) conv[i] = dna_below[i] + upa_above[i];
      mat_vec<N, N>(M, conv, ifa);
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) ifa[i] = ifa[i] + Is[i];
      SS_UNROLL_N(N)
      for (int i = 0; i < N; ++i) conv[i] = dnb_below[i] + upb_above[i];
      mat_vec<N, N>(M, conv, ifb);
    }

    // absorption (forest :571-590 / :658-672, urban :700-737 / :811-835)
    const double air_ext = in.air_ext[sidx], air_ssa = in.air_ssa[sidx];
    {
      double sa = 0.0, sb = 0.0;
      SS_UNROLL_FULL
      for (int js = 0; js < NS; ++js) {
        sa += ifa[js] * (1.0 / lg.mu[js]);
        sb += ifb[js] * (1.0 / lg.mu[js]);
      }
      const double volume_emiss = p.frac[0] * (p.ext_reg[0] * (1.0 - p.ssa_reg[0]) * p.planck_reg[0]);
      PUTL(fa, F_CLEAR_AIR_ABS, 0.0 + air_ext * (1.0 - air_ssa) * sa - (emiss_factor * volume_emiss) * p.g.dz);
      PUTL(fb, F_CLEAR_AIR_ABS, 0.0 + air_ext * (1.0 - air_ssa) * sb);
    }
    if (NREG > 1) {
      double vaa_a = 0.0, va_a = 0.0, vaa_b = 0.0, va_b = 0.0;
      SS_UNROLL_FULL
      for (int jreg = 1; jreg < NREG; ++jreg) {
        double sa = 0.0, sb = 0.0;
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) {
          sa += ifa[jreg * NS + js] * (1.0 / lg.mu[js]);
          sb += ifb[jreg * NS + js] * (1.0 / lg.mu[js]);
        }
        const double emiss_air = emiss_factor * p.frac[jreg] * p.ext_reg[0] * (1.0 - p.ssa_reg[0]) * p.vaplanck;
        const double emiss_veg = emiss_factor * p.frac[jreg] * p.g.veg_ext * (1.0 - p.vssa) * p.vplanck * p.od[jreg];
        vaa_a = vaa_a + air_ext * (1.0 - air_ssa) * sa - emiss_air * p.g.dz;
        va_a = va_a + p.g.veg_ext * (1.0 - p.vssa) * sa * p.od[jreg] - emiss_veg * p.g.dz;
        vaa_b = vaa_b + air_ext * (1.0 - air_ssa) * sb;
        va_b = va_b + p.g.veg_ext * (1.0 - p.vssa) * sb * p.od[jreg];
      }
      PUTL(fa, F_VEG_AIR_ABS, vaa_a);
      PUTL(fa, F_VEG_ABS, va_a);
      PUTL(fb, F_VEG_AIR_ABS, vaa_b);
      PUTL(fb, F_VEG_ABS, va_b);
    } else {
      PUTL(fa, F_VEG_AIR_ABS, 0.0);
      PUTL(fa, F_VEG_ABS, 0.0);
      PUTL(fb, F_VEG_AIR_ABS, 0.0);
      PUTL(fb, F_VEG_ABS, 0.0);
    }
    if (URBAN) {
      double wall_in_a = 0.0, wall_in_b = 0.0, npw_sum = 0.0;
      SS_UNROLL_FULL
      for (int jreg = 0; jreg < NREG; ++jreg) {
        double sa = 0.0, sb = 0.0;
        SS_UNROLL_FULL
        for (int js = 0; js < NS; ++js) {
          sa += ifa[jreg * NS + js] * lg.tan_ang[js];
          sb += ifb[jreg * NS + js] * lg.tan_ang[js];
        }
        wall_in_a = wall_in_a + p.f_wall[jreg] * sa;
        wall_in_b = wall_in_b + p.f_wall[jreg] * (sb);
        npw_sum += p.npw[jreg];
      }
      const double emiss_wall = (npw_sum * lg.vadjustment) * in.wall_emission[sidx];
      PUTL(fa, F_WALL_IN, wall_in_a);
      PUTL(fa, F_WALL_NET, wall_in_a * in.wall_emissivity[sidx] - emiss_wall * p.g.dz);
      PUTL(fb, F_WALL_IN, wall_in_b);
      PUTL(fb, F_WALL_NET, wall_in_b * in.wall_emissivity[sidx]);
    } else {
      PUTL(fa, F_WALL_IN, 0.0);
      PUTL(fa, F_WALL_NET, 0.0);
      PUTL(fb, F_WALL_IN, 0.0);
      PUTL(fb, F_WALL_NET, 0.0);
    }
  }

  // ground (forest :599-610 / :680-694, urban :741-752 / :845-858)
  {
    double s2a = 0.0, s3a = 0.0, s2b = 0.0, s3b = 0.0, gvda = 0.0, gvdb = 0.0;
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) { s2a += dna_above[i]; s2b += dnb_above[i]; }
    SS_UNROLL_N(N)
    for (int i = 0; i < N; ++i) { s3a += upa_above[i]; s3b += upb_above[i]; }
    PUTC(fa, F_GROUND_DN, s2a);
    PUTC(fa, F_GROUND_NET, s2a - s3a);
    PUTC(fb, F_GROUND_DN, s2b);
    PUTC(fb, F_GROUND_NET, s2b - s3b);
    SS_UNROLL_FULL
    for (int jreg = 0; jreg < NREG; ++jreg) {
      SS_UNROLL_FULL
      for (int js = 0; js < NS; ++js) {
        const int ifr = js + jreg * NS;
        gvda = gvda + (dna_above[ifr] + upa_above[ifr]) * lg.tan_ang[js] / kPi;
      }
    }
    // second accumulation separately, in the reference's order
    double gvd2 = URBAN ? 0.0 : gvda;
    SS_UNROLL_FULL
    for (int jreg = 0; jreg < NREG; ++jreg) {
      SS_UNROLL_FULL
      for (int js = 0; js < NS; ++js) {
        const int ifr = js + jreg * NS;
        gvd2 = gvd2 + (dnb_above[ifr] + upb_above[ifr]) * lg.tan_ang[js] / kPi;
      }
    }
    gvdb = gvd2;
    if (URBAN) {
      PUTC(fa, F_GROUND_VERTICAL_DIFF, gvda);
      PUTC(fb, F_GROUND_VERTICAL_DIFF, gvdb);
    } else {
      // forest: the incoming pass also accumulates into lw_internal and
      // lw_norm%ground_vertical_diff stays zero (radsurf_forest_lw.F90:691)
      PUTC(fa, F_GROUND_VERTICAL_DIFF, gvdb);
      PUTC(fb, F_GROUND_VERTICAL_DIFF, 0.0);
    }
  }
#undef PUTC
#undef PUTL
}

}  // namespace ss
