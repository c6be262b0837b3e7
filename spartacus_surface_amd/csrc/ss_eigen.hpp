// ss_eigen.hpp - per-lane eigen-decomposition of a small real matrix with a
// real spectrum: the algorithm of eigen_decomposition_real
// (radtool/radtool_eigen_decomposition.F90:51-828; EISPACK/DISORT ASYMTX):
// Parlett-Reinsch balancing with isolation of trivially known eigenvalues
// (:182-310), Householder reduction to Hessenberg form (:312-395), shifted
// double-step QR with exceptional shifts at iterations 10 and 20 and a cap of
// 30 (:410-635), back-substitution (:636-717); closed forms for orders 2 and 1
// (:770-816).
//
// The iteration is data dependent (deflation, early exit), so each lane runs
// its own trip counts; the work arrays are addressed with run-time indices and
// therefore live in the lane's scratch (order > 2 only).  The same sequence of
// arithmetic as the reference is kept on purpose: nearly degenerate spectra
// are normal here (identical regions), eigenvectors are then individually
// ill-conditioned and only an identical algorithm reproduces the reference's
// R/T matrices to ~1e-12 (SURVEY.md section 7 "hard parts").
#pragma once

#include "ss_common.hpp"
#include "ss_eigen_static.hpp"

namespace ss {

template <int N>
SS_HD int eigen_decomposition_real_dynamic(const double* amat, double* eigenvalue, double* eigenvector);

// Dispatch: closed forms for N <= 2 (inside the dynamic routine), the
// register-resident form for 3 <= N <= SS_UNROLL_MAX, run-time indexed arrays
// (scratch) above that.
template <int N>
SS_HD int eigen_decomposition_real(const double* amat, double* eigenvalue, double* eigenvector) {
#if defined(SS_TIMING_NO_EIGEN)   // timing experiment only: wrong results
  if constexpr (N >= 3) {
    for (int i = 0; i < N; ++i) {
      eigenvalue[i] = amat[i + i * N];
      for (int j = 0; j < N; ++j) eigenvector[i + j * N] = (i == j) ? 1.0 : 0.01 * amat[i + j * N];
    }
    return 0;
  } else
#elif !defined(SS_EIGEN_DYNAMIC_ONLY)
  if constexpr (N >= 3 && N <= SS_UNROLL_MAX) {
    return eigen_decomposition_real_static<N>(amat, eigenvalue, eigenvector);
  } else
#endif
  {
    return eigen_decomposition_real_dynamic<N>(amat, eigenvalue, eigenvector);
  }
}

// amat: N x N column-major (not modified).  eigenvalue: N.  eigenvector: N x N
// (columns).  Returns 0 or the index of the eigenvalue at which the QR
// iteration gave up (:498-510; the reference's callers ignore it).
template <int N>
SS_HD int eigen_decomposition_real_dynamic(const double* amat, double* eigenvalue, double* eigenvector) {
  if (N == 1) {
    eigenvalue[0] = amat[0];
    eigenvector[0] = 1.0;
    return 0;
  }
#ifdef SS_TIMING_NO_EIGEN   // timing experiment only: wrong results
  for (int i = 0; i < N; ++i) {
    eigenvalue[i] = amat[i + i * N];
    for (int j = 0; j < N; ++j) eigenvector[i + j * N] = (i == j) ? 1.0 : 0.01 * amat[i + j * N];
  }
  return 0;
#endif
  const double Tol = DBL_EPSILON;
  if (N == 2) {
    const double a11 = amat[0], a21 = amat[1], a12 = amat[2], a22 = amat[3];
    const double discriminant = (a11 - a22) * (a11 - a22) + 4.0 * a12 * a21;
    const int ierr = (discriminant < 0.0) ? 2 : 0;
    const double half_sqrt_disc = 0.5 * sqrt(discriminant);
    double e1 = 0.5 * (a11 + a22), e2 = e1;
    if (a11 >= a22) {
      e1 = e1 + half_sqrt_disc;
      e2 = e2 - half_sqrt_disc;
    } else {
      e1 = e1 - half_sqrt_disc;
      e2 = e2 + half_sqrt_disc;
    }
    eigenvalue[0] = e1;
    eigenvalue[1] = e2;
    eigenvector[0] = 1.0;
    eigenvector[3] = 1.0;
    if (a11 == a22 && (a21 == 0.0 || a12 == 0.0)) {
      // operator precedence exactly as written in the reference :795-797
      const double rnorm = 1.0 / (Tol * fabs(a11) + fabs(a21) + fabs(a12) + fabs(a22));
      eigenvector[1] = a21 * rnorm;
      eigenvector[2] = a12 * rnorm;
    } else {
      eigenvector[1] = a21 / (e1 - a22);
      eigenvector[2] = a12 / (e2 - a11);
    }
    return ierr;
  }

  constexpr int n = N;
  const double C1 = 0.4375, C2 = 0.5, C3 = 0.75, C4 = 0.95, C5 = 16.0, C6 = 256.0;
  double abal[(N > 2 ? N : 1) * (N > 2 ? N : 1)];
  double wkd[2 * (N > 2 ? N : 1)];
  double* eigenvec = eigenvector;
  double* eigenval = eigenvalue;
#define AB(i, j) abal[((i)-1) + ((j)-1) * n]
#define EV(i, j) eigenvec[((i)-1) + ((j)-1) * n]
#define WK(i) wkd[(i)-1]
#define EVAL(i) eigenval[(i)-1]
  int ji, jj, jn, ka, kk, ll, lb = 0, lll, n1, n2, in, kkk;
  double column, ff, gg, hh, pp = 0.0, qq = 0.0, rr = 0.0, tmp, rnorm, row, ss, scale, tt;
  double uu, vv, ww, xx, yy, zz;
  bool not_finished, not_last, is_error = false;
  int ierr = 0;

  for (ji = 1; ji <= n; ++ji) EVAL(ji) = 0.0;
  for (ji = 0; ji < n * n; ++ji) {
    eigenvec[ji] = 0.0;
    abal[ji] = amat[ji];
  }
  for (ji = 1; ji <= n; ++ji) EV(ji, ji) = 1.0;
  for (ji = 0; ji < 2 * n; ++ji) wkd[ji] = 0.0;

  rnorm = 0.0;
  ll = 1;
  kk = n;

  // rows isolating an eigenvalue are pushed down (:182-222)
  not_finished = true;
  while (not_finished) {
    not_finished = false;
    kkk = kk;
    for (jj = kkk; jj >= 1; --jj) {
      row = 0.0;
      for (ji = 1; ji <= kk; ++ji)
        if (ji != jj) row += fabs(AB(jj, ji));
      if (row == 0.0) {
        WK(kk) = jj;
        // ":201 if (ji /= kk)" is always true in the reference
        for (ji = 1; ji <= kk; ++ji) {
          tmp = AB(ji, jj); AB(ji, jj) = AB(ji, kk); AB(ji, kk) = tmp;
        }
        for (ji = ll; ji <= n; ++ji) {
          tmp = AB(jj, ji); AB(jj, ji) = AB(kk, ji); AB(kk, ji) = tmp;
        }
        kk = kk - 1;
        not_finished = true;
        break;
      }
    }
  }

  // columns isolating an eigenvalue are pushed left (:227-262)
  not_finished = true;
  while (not_finished) {
    not_finished = false;
    lll = ll;
    for (jj = lll; jj <= kk; ++jj) {
      column = 0.0;
      for (ji = ll; ji <= kk; ++ji)
        if (ji != jj) column += fabs(AB(ji, jj));
      if (column == 0.0) {
        WK(ll) = jj;
        if (jj != ll) {
          for (ji = 1; ji <= kk; ++ji) {
            tmp = AB(ji, jj); AB(ji, jj) = AB(ji, ll); AB(ji, ll) = tmp;
          }
          for (ji = ll; ji <= n; ++ji) {
            tmp = AB(jj, ji); AB(jj, ji) = AB(ll, ji); AB(ll, ji) = tmp;
          }
        }
        ll = ll + 1;
        not_finished = true;
        break;
      }
    }
  }

  // balance rows ll..kk (:264-310)
  for (ji = ll; ji <= kk; ++ji) WK(ji) = 1.0;
  not_finished = true;
  for (int sweep = 0; not_finished && sweep < 1000; ++sweep) {
    not_finished = false;
    for (ji = ll; ji <= kk; ++ji) {
      column = 0.0;
      row = 0.0;
      for (jj = ll; jj <= kk; ++jj)
        if (jj != ji) {
          column += fabs(AB(jj, ji));
          row += fabs(AB(ji, jj));
        }
      ff = 1.0;
      gg = row / C5;
      hh = column + row;
      // (the caps only matter for a zero/NaN column, where the reference would
      // spin forever; a GPU wave must always drain)
      for (int guard = 0; column < gg && guard < 1200; ++guard) { ff = ff * C5; column = column * C6; }
      gg = row * C5;
      for (int guard = 0; column > gg && guard < 1200; ++guard) { ff = ff / C5; column = column / C6; }
      if ((column + row) / ff < C4 * hh) {
        WK(ji) = WK(ji) * ff;
        not_finished = true;
        for (jj = ll; jj <= n; ++jj) AB(ji, jj) = AB(ji, jj) / ff;
        for (jj = 1; jj <= kk; ++jj) AB(jj, ji) = AB(jj, ji) * ff;
      }
    }
  }

  // Householder reduction to Hessenberg form (:312-395)
  if (kk - 1 >= ll + 1) {
    for (jn = ll + 1; jn <= kk - 1; ++jn) {
      hh = 0.0;
      WK(jn + n) = 0.0;
      scale = 0.0;
      for (ji = jn; ji <= kk; ++ji) scale += fabs(AB(ji, jn - 1));
      if (scale != 0.0) {
        for (ji = kk; ji >= jn; --ji) {
          WK(ji + n) = AB(ji, jn - 1) / scale;
          hh += WK(ji + n) * WK(ji + n);
        }
        gg = -ss_sign(sqrt(hh), WK(jn + n));
        hh = hh - WK(jn + n) * gg;
        WK(jn + n) = WK(jn + n) - gg;
        hh = 1.0 / hh;
        for (jj = jn; jj <= n; ++jj) {
          ff = 0.0;
          for (ji = kk; ji >= jn; --ji) ff += WK(ji + n) * AB(ji, jj);
          for (ji = jn; ji <= kk; ++ji) AB(ji, jj) = AB(ji, jj) - WK(ji + n) * ff * hh;
        }
        for (ji = 1; ji <= kk; ++ji) {
          ff = 0.0;
          for (jj = kk; jj >= jn; --jj) ff += WK(jj + n) * AB(ji, jj);
          for (jj = jn; jj <= kk; ++jj) AB(ji, jj) = AB(ji, jj) - WK(jj + n) * ff * hh;
        }
        WK(jn + n) = scale * WK(jn + n);
        AB(jn, jn - 1) = scale * gg;
      }
    }
    for (jn = kk - 2; jn >= ll; --jn) {
      n1 = jn + 1;
      n2 = jn + 2;
      ff = AB(n1, jn);
      if (ff != 0.0) {
        ff = ff * WK(jn + 1 + n);
        for (ji = n2; ji <= kk; ++ji) WK(ji + n) = AB(ji, jn);
        if (n1 < kk) {
          for (jj = 1; jj <= n; ++jj) {
            gg = 0.0;
            for (ji = n1; ji <= kk; ++ji) gg += WK(ji + n) * EV(ji, jj);
            gg = gg / ff;
            for (ji = n1; ji <= kk; ++ji) EV(ji, jj) = EV(ji, jj) + gg * WK(ji + n);
          }
        }
      }
    }
  }

  // :397-408
  jn = 1;
  for (ji = 1; ji <= n; ++ji) {
    for (jj = jn; jj <= n; ++jj) rnorm += fabs(AB(ji, jj));
    jn = ji;
    if (ji < ll || ji > kk) EVAL(ji) = AB(ji, ji);
  }
  jn = kk;
  tt = 0.0;

  // shifted QR iteration (:410-635)
  not_finished = true;
  while (not_finished && jn >= ll) {
    not_finished = false;
    in = 0;
    n1 = jn - 1;
    n2 = jn - 2;
    for (;;) {
      for (ji = ll; ji <= jn; ++ji) {
        lb = jn + ll - ji;
        if (lb == ll) break;
        ss = fabs(AB(lb - 1, lb - 1)) + fabs(AB(lb, lb));
        if (ss == 0.0) ss = rnorm;
        if (fabs(AB(lb, lb - 1)) < Tol * ss) break;
      }
      xx = AB(jn, jn);
      if (lb == jn) {
        AB(jn, jn) = xx + tt;
        EVAL(jn) = AB(jn, jn);
        jn = n1;
        not_finished = true;
        break;
      }
      yy = AB(n1, n1);
      ww = AB(jn, n1) * AB(n1, jn);
      if (lb == n1) {
        pp = (yy - xx) * C2;
        qq = pp * pp + ww;
        zz = sqrt(fabs(qq));
        AB(jn, jn) = xx + tt;
        xx = AB(jn, jn);
        AB(n1, n1) = yy + tt;
        zz = pp + ss_sign(zz, pp);
        EVAL(n1) = xx + zz;
        EVAL(jn) = EVAL(n1);
        if (zz != 0.0) EVAL(jn) = xx - ww / zz;
        xx = AB(jn, n1);
        rr = 1.0 / sqrt(xx * xx + zz * zz);
        pp = xx * rr;
        qq = zz * rr;
        for (jj = n1; jj <= n; ++jj) {
          zz = AB(n1, jj);
          AB(n1, jj) = qq * zz + pp * AB(jn, jj);
          AB(jn, jj) = qq * AB(jn, jj) - pp * zz;
        }
        for (ji = 1; ji <= jn; ++ji) {
          zz = AB(ji, n1);
          AB(ji, n1) = qq * zz + pp * AB(ji, jn);
          AB(ji, jn) = qq * AB(ji, jn) - pp * zz;
        }
        for (ji = ll; ji <= kk; ++ji) {
          zz = EV(ji, n1);
          EV(ji, n1) = qq * zz + pp * EV(ji, jn);
          EV(ji, jn) = qq * EV(ji, jn) - pp * zz;
        }
        jn = n2;
        not_finished = true;
        break;
      }
      if (in == 30) {
        ierr = jn;
        is_error = true;
        not_finished = false;
        break;
      }
      if (in == 10 || in == 20) {
        tt = tt + xx;
        for (ji = ll; ji <= jn; ++ji) AB(ji, ji) = AB(ji, ji) - xx;
        ss = fabs(AB(jn, n1)) + fabs(AB(n1, n2));
        xx = C3 * ss;
        yy = xx;
        ww = -C1 * ss * ss;
      }
      in = in + 1;
      for (jj = lb; jj <= n2; ++jj) {
        ji = n2 + lb - jj;
        zz = AB(ji, ji);
        rr = xx - zz;
        ss = yy - zz;
        pp = (rr * ss - ww) / AB(ji + 1, ji) + AB(ji, ji + 1);
        qq = AB(ji + 1, ji + 1) - zz - rr - ss;
        rr = AB(ji + 2, ji + 1);
        ss = 1.0 / (fabs(pp) + fabs(qq) + fabs(rr));
        pp = pp * ss;
        qq = qq * ss;
        rr = rr * ss;
        if (ji == lb) break;
        uu = fabs(AB(ji, ji - 1)) * (fabs(qq) + fabs(rr));
        vv = fabs(pp) * (fabs(AB(ji - 1, ji - 1)) + fabs(zz) + fabs(AB(ji + 1, ji + 1)));
        if (uu <= Tol * vv) break;
      }
      AB(ji + 2, ji) = 0.0;
      for (jj = ji + 3; jj <= jn; ++jj) {
        AB(jj, jj - 2) = 0.0;
        AB(jj, jj - 3) = 0.0;
      }
      {
        const int ka_first = ji;
        for (ka = ka_first; ka <= n1; ++ka) {
          not_last = (ka != n1);
          if (ka == ka_first) {
            ss = ss_sign(sqrt(pp * pp + qq * qq + rr * rr), pp);
            if (lb != ka_first) AB(ka, ka - 1) = -AB(ka, ka - 1);
          } else {
            pp = AB(ka, ka - 1);
            qq = AB(ka + 1, ka - 1);
            rr = 0.0;
            if (not_last) rr = AB(ka + 2, ka - 1);
            xx = fabs(pp) + fabs(qq) + fabs(rr);
            if (xx == 0.0) continue;
            pp = pp / xx;
            qq = qq / xx;
            rr = rr / xx;
            ss = ss_sign(sqrt(pp * pp + qq * qq + rr * rr), pp);
            AB(ka, ka - 1) = -ss * xx;
          }
          pp = pp + ss;
          ss = 1.0 / ss;
          xx = pp * ss;
          yy = qq * ss;
          zz = rr * ss;
          pp = 1.0 / pp;
          qq = qq * pp;
          rr = rr * pp;
          for (jj = ka; jj <= n; ++jj) {
            pp = AB(ka, jj) + qq * AB(ka + 1, jj);
            if (not_last) {
              pp = pp + rr * AB(ka + 2, jj);
              AB(ka + 2, jj) = AB(ka + 2, jj) - pp * zz;
            }
            AB(ka + 1, jj) = AB(ka + 1, jj) - pp * yy;
            AB(ka, jj) = AB(ka, jj) - pp * xx;
          }
          {
            const int top = (jn < ka + 3) ? jn : ka + 3;
            for (ji = 1; ji <= top; ++ji) {
              pp = xx * AB(ji, ka) + yy * AB(ji, ka + 1);
              if (not_last) {
                pp = pp + zz * AB(ji, ka + 2);
                AB(ji, ka + 2) = AB(ji, ka + 2) - pp * rr;
              }
              AB(ji, ka + 1) = AB(ji, ka + 1) - pp * qq;
              AB(ji, ka) = AB(ji, ka) - pp;
            }
          }
          for (ji = ll; ji <= kk; ++ji) {
            pp = xx * EV(ji, ka) + yy * EV(ji, ka + 1);
            if (not_last) {
              pp = pp + zz * EV(ji, ka + 2);
              EV(ji, ka + 2) = EV(ji, ka + 2) - pp * rr;
            }
            EV(ji, ka + 1) = EV(ji, ka + 1) - pp * qq;
            EV(ji, ka) = EV(ji, ka) - pp;
          }
        }
      }
    }
  }

  // back-substitution (:636-717)
  if (!is_error) {
    if (rnorm != 0.0) {
      for (jn = n; jn >= 1; --jn) {
        n2 = jn;
        AB(jn, jn) = 1.0;
        for (ji = jn - 1; ji >= 1; --ji) {
          ww = AB(ji, ji) - EVAL(jn);
          if (fabs(ww) < fabs(Tol * rnorm)) ww = ss_sign(Tol * rnorm, ww);
          rr = AB(ji, jn);
          for (jj = n2; jj <= jn - 1; ++jj) rr += AB(ji, jj) * AB(jj, jn);
          AB(ji, jn) = -rr / ww;
          n2 = ji;
        }
      }
      for (ji = 1; ji <= n; ++ji)
        if (ji < ll || ji > kk)
          for (jj = ji; jj <= n; ++jj) EV(ji, jj) = AB(ji, jj);
      if (kk != 0) {
        for (jj = n; jj >= ll; --jj) {
          for (ji = ll; ji <= kk; ++ji) {
            const int top = (jj < kk) ? jj : kk;
            zz = 0.0;
            for (jn = ll; jn <= top; ++jn) zz += EV(ji, jn) * AB(jn, jj);
            EV(ji, jj) = zz;
          }
        }
      }
    }
    for (ji = ll; ji <= kk; ++ji)
      for (jj = 1; jj <= n; ++jj) EV(ji, jj) = EV(ji, jj) * WK(ji);
    for (ji = ll - 1; ji >= 1; --ji) {
      jj = (int)(WK(ji) + 0.5);
      if (ji < jj)
        for (jn = 1; jn <= n; ++jn) {
          tmp = EV(ji, jn); EV(ji, jn) = EV(jj, jn); EV(jj, jn) = tmp;
        }
    }
    for (ji = kk + 1; ji <= n; ++ji) {
      jj = (int)(WK(ji) + 0.5);
      if (ji != jj)
        for (jn = 1; jn <= n; ++jn) {
          tmp = EV(ji, jn); EV(ji, jn) = EV(jj, jn); EV(jj, jn) = tmp;
        }
    }
  }
  return ierr;
#undef AB
#undef EV
#undef WK
#undef EVAL
}

}  // namespace ss
