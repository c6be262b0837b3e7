// ss_eigen_static.hpp - register-resident form of eigen_decomposition_real for
// small orders (3 <= N <= SS_UNROLL_MAX).
//
// Same algorithm, same arithmetic and the same operation order on every
// matrix element as ss_eigen.hpp / radtool_eigen_decomposition.F90:51-828 -
// on the CPU the two produce bit-identical results (tests/test_emul_eigen.py).
// The difference is purely how elements are addressed: every loop over matrix
// indices runs over its full compile-time range and is fully unrolled, and the
// run-time bounds of the reference (the deflation window ll..kk, the active
// eigenvalue jn, the bulge position ka, ...) become PREDICATES on those static
// iterations.  Every element of the balanced matrix, of the eigenvector matrix
// and of the work vector then has a fixed register, and nothing of the
// iteration touches memory.  (The run-time-indexed version keeps those arrays
// in scratch memory: measured on MI355X it spends ~85 % of the layer kernels in
// memory-latency stalls of the QR sweeps, profiles/round1_v0_pmc.json.)
//
// Run-time-indexed scalar reads (a handful per sweep: A(jn,jn), A(jn,n1), ...)
// are selects over the static diagonal / sub-diagonals.
#pragma once

#include "ss_common.hpp"

namespace ss {

template <int N>
struct EigStatic {
  static constexpr int n = N;
  double ab[N * N];   // balanced matrix / Hessenberg / triangular form
  double wk[2 * N];   // permutation + scaling (1..N), Householder vector (N+1..2N)

  // 1-based element access with compile-time-foldable indices
  SS_HD double& A(int i, int j) { return ab[(i - 1) + (j - 1) * N]; }
  SS_HD double& W(int i) { return wk[i - 1]; }

  // run-time index reads as selects over static positions
  SS_HD double diag(int k) {
    double v = 0.0;
    SS_UNROLL_FULL
    for (int i = 1; i <= N; ++i) v = (i == k) ? A(i, i) : v;
    return v;
  }
  SS_HD double subdiag(int k) {   // A(k, k-1), 2 <= k <= N
    double v = 0.0;
    SS_UNROLL_FULL
    for (int i = 2; i <= N; ++i) v = (i == k) ? A(i, i - 1) : v;
    return v;
  }
  SS_HD double superdiag(int k) {   // A(k-1, k), 2 <= k <= N
    double v = 0.0;
    SS_UNROLL_FULL
    for (int i = 2; i <= N; ++i) v = (i == k) ? A(i - 1, i) : v;
    return v;
  }
};

template <int N>
SS_HD int eigen_decomposition_real_static(const double* amat, double* eigenval, double* eigenvec) {
  static_assert(N >= 3, "orders 1 and 2 use the closed forms");
  const double Tol = DBL_EPSILON;
  const double C1 = 0.4375, C2 = 0.5, C3 = 0.75, C4 = 0.95, C5 = 16.0, C6 = 256.0;
  EigStatic<N> S;
#define AB(i, j) S.A((i), (j))
#define EV(i, j) eigenvec[((i)-1) + ((j)-1) * N]
#define WK(i) S.W((i))
#define EVAL(i) eigenval[(i)-1]
#define FOR(v) SS_UNROLL_FULL for (int v = 1; v <= N; ++v)
#define FORDOWN(v) SS_UNROLL_FULL for (int v = N; v >= 1; --v)

  FOR(i) EVAL(i) = 0.0;
  SS_UNROLL_FULL
  for (int k = 0; k < N * N; ++k) {
    eigenvec[k] = 0.0;
    S.ab[k] = amat[k];
  }
  FOR(i) EV(i, i) = 1.0;
  SS_UNROLL_FULL
  for (int k = 0; k < 2 * N; ++k) S.wk[k] = 0.0;

  double rnorm = 0.0;
  int ll = 1, kk = N;

  // ---- rows isolating an eigenvalue are pushed down (:182-222) ----
  {
    bool not_finished = true;
    while (not_finished) {
      not_finished = false;
      bool done = false;
      FORDOWN(jj) {
        if (!done && jj <= kk) {
          double row = 0.0;
          FOR(ji) if (ji <= kk && ji != jj) row += fabs(AB(jj, ji));
          if (row == 0.0) {
            // WK(kk) = jj; swap column jj <-> kk (rows 1..kk), row jj <-> kk (cols ll..N)
            FOR(c) {
              if (c == kk) {
                WK(c) = (double)jj;
                FOR(ji) if (ji <= kk) { const double t = AB(ji, jj); AB(ji, jj) = AB(ji, c); AB(ji, c) = t; }
                FOR(ji) if (ji >= ll) { const double t = AB(jj, ji); AB(jj, ji) = AB(c, ji); AB(c, ji) = t; }
              }
            }
            kk = kk - 1;
            not_finished = true;
            done = true;
          }
        }
      }
    }
  }

  // ---- columns isolating an eigenvalue are pushed left (:227-262) ----
  {
    bool not_finished = true;
    while (not_finished) {
      not_finished = false;
      bool done = false;
      FOR(jj) {
        if (!done && jj >= ll && jj <= kk) {
          double column = 0.0;
          FOR(ji) if (ji >= ll && ji <= kk && ji != jj) column += fabs(AB(ji, jj));
          if (column == 0.0) {
            FOR(c) {
              if (c == ll) {
                WK(c) = (double)jj;
                if (jj != c) {
                  FOR(ji) if (ji <= kk) { const double t = AB(ji, jj); AB(ji, jj) = AB(ji, c); AB(ji, c) = t; }
                  FOR(ji) if (ji >= ll) { const double t = AB(jj, ji); AB(jj, ji) = AB(c, ji); AB(c, ji) = t; }
                }
              }
            }
            ll = ll + 1;
            not_finished = true;
            done = true;
          }
        }
      }
    }
  }

  // ---- balance rows ll..kk (:264-310) ----
  FOR(ji) if (ji >= ll && ji <= kk) WK(ji) = 1.0;
  {
    bool not_finished = true;
    for (int sweep = 0; not_finished && sweep < 1000; ++sweep) {
      not_finished = false;
      FOR(ji) {
        if (ji >= ll && ji <= kk) {
          double column = 0.0, row = 0.0;
          FOR(jj) if (jj >= ll && jj <= kk && jj != ji) {
            column += fabs(AB(jj, ji));
            row += fabs(AB(ji, jj));
          }
          double ff = 1.0;
          double gg = row / C5;
          const double hh = column + row;
          for (int guard = 0; column < gg && guard < 1200; ++guard) { ff = ff * C5; column = column * C6; }
          gg = row * C5;
          for (int guard = 0; column > gg && guard < 1200; ++guard) { ff = ff / C5; column = column / C6; }
          if ((column + row) / ff < C4 * hh) {
            WK(ji) = WK(ji) * ff;
            not_finished = true;
            FOR(jj) if (jj >= ll) AB(ji, jj) = AB(ji, jj) / ff;
            FOR(jj) if (jj <= kk) AB(jj, ji) = AB(jj, ji) * ff;
          }
        }
      }
    }
  }

  // ---- Householder reduction to Hessenberg form (:312-395) ----
  if (kk - 1 >= ll + 1) {
    SS_UNROLL_FULL
    for (int jn = 2; jn <= N - 1; ++jn) {
      if (jn >= ll + 1 && jn <= kk - 1) {
        double hh = 0.0;
        WK(jn + N) = 0.0;
        double scale = 0.0;
        SS_UNROLL_FULL
        for (int ji = jn; ji <= N; ++ji) if (ji <= kk) scale += fabs(AB(ji, jn - 1));
        if (scale != 0.0) {
          SS_UNROLL_FULL
          for (int ji = N; ji >= jn; --ji) if (ji <= kk) {
            WK(ji + N) = AB(ji, jn - 1) / scale;
            hh += WK(ji + N) * WK(ji + N);
          }
          const double gg = -ss_sign(sqrt(hh), WK(jn + N));
          hh = hh - WK(jn + N) * gg;
          WK(jn + N) = WK(jn + N) - gg;
          hh = 1.0 / hh;
          SS_UNROLL_FULL
          for (int jj = jn; jj <= N; ++jj) {
            double ff = 0.0;
            SS_UNROLL_FULL
            for (int ji = N; ji >= jn; --ji) if (ji <= kk) ff += WK(ji + N) * AB(ji, jj);
            SS_UNROLL_FULL
            for (int ji = jn; ji <= N; ++ji) if (ji <= kk) AB(ji, jj) = AB(ji, jj) - WK(ji + N) * ff * hh;
          }
          FOR(ji) if (ji <= kk) {
            double ff = 0.0;
            SS_UNROLL_FULL
            for (int jj = N; jj >= jn; --jj) if (jj <= kk) ff += WK(jj + N) * AB(ji, jj);
            SS_UNROLL_FULL
            for (int jj = jn; jj <= N; ++jj) if (jj <= kk) AB(ji, jj) = AB(ji, jj) - WK(jj + N) * ff * hh;
          }
          WK(jn + N) = scale * WK(jn + N);
          AB(jn, jn - 1) = scale * gg;
        }
      }
    }
    SS_UNROLL_FULL
    for (int jn = N - 2; jn >= 1; --jn) {
      if (jn <= kk - 2 && jn >= ll) {
        const int n1 = jn + 1;   // static
        double ff = AB(n1, jn);
        if (ff != 0.0) {
          ff = ff * WK(jn + 1 + N);
          SS_UNROLL_FULL
          for (int ji = jn + 2; ji <= N; ++ji) if (ji <= kk) WK(ji + N) = AB(ji, jn);
          if (n1 < kk) {
            FOR(jj) {
              double gg = 0.0;
              SS_UNROLL_FULL
              for (int ji = n1; ji <= N; ++ji) if (ji <= kk) gg += WK(ji + N) * EV(ji, jj);
              gg = gg / ff;
              SS_UNROLL_FULL
              for (int ji = n1; ji <= N; ++ji) if (ji <= kk) EV(ji, jj) = EV(ji, jj) + gg * WK(ji + N);
            }
          }
        }
      }
    }
  }

  // ---- :397-408 ----
  FOR(ji) {
    // jn of the reference: 1 for ji = 1, else ji-1
    SS_UNROLL_FULL
    for (int jj = (ji == 1 ? 1 : ji - 1); jj <= N; ++jj) rnorm += fabs(AB(ji, jj));
    if (ji < ll || ji > kk) EVAL(ji) = AB(ji, ji);
  }
  int jn = kk;
  double tt = 0.0;
  bool is_error = false;
  int ierr = 0;

  // ---- shifted QR iteration (:410-635) ----
  bool not_finished = true;
  while (not_finished && jn >= ll) {
    not_finished = false;
    int in = 0;
    int n1 = jn - 1;
    int n2 = jn - 2;
    for (;;) {
      // look for a single small sub-diagonal element, from jn down to ll+1
      int lb = ll;
      {
        bool stop = false;
        SS_UNROLL_FULL
        for (int l = N; l >= 2; --l) {
          if (!stop && l <= jn && l > ll) {
            double ss = fabs(AB(l - 1, l - 1)) + fabs(AB(l, l));
            if (ss == 0.0) ss = rnorm;
            if (fabs(AB(l, l - 1)) < Tol * ss) { lb = l; stop = true; }
          }
        }
      }
      double xx = S.diag(jn);
      if (lb == jn) {
        // one eigenvalue found
        FOR(k) if (k == jn) { AB(k, k) = xx + tt; EVAL(k) = AB(k, k); }
        jn = n1;
        not_finished = true;
        break;
      }
      double yy = S.diag(n1);
      double ww = S.subdiag(jn) * S.superdiag(jn);
      if (lb == n1) {
        // two eigenvalues found (:449-496); rows/cols (n1, jn) = (r, r+1)
        double pp = (yy - xx) * C2;
        double qq = pp * pp + ww;
        double zz = sqrt(fabs(qq));
        const double xjn = xx + tt;   // new A(jn,jn)
        xx = xjn;
        zz = pp + ss_sign(zz, pp);
        const double e_n1 = xx + zz;
        double e_jn = e_n1;
        if (zz != 0.0) e_jn = xx - ww / zz;
        SS_UNROLL_FULL
        for (int r = 1; r <= N - 1; ++r) {
          if (r == n1) {
            AB(r + 1, r + 1) = xjn;
            AB(r, r) = yy + tt;
            EVAL(r) = e_n1;
            EVAL(r + 1) = e_jn;
            const double x2 = AB(r + 1, r);
            const double rr = 1.0 / sqrt(x2 * x2 + zz * zz);
            const double p2 = x2 * rr;
            const double q2 = zz * rr;
            // row modification, jj = n1..N
            SS_UNROLL_FULL
            for (int jj = r; jj <= N; ++jj) {
              const double z = AB(r, jj);
              AB(r, jj) = q2 * z + p2 * AB(r + 1, jj);
              AB(r + 1, jj) = q2 * AB(r + 1, jj) - p2 * z;
            }
            // column modification, ji = 1..jn
            SS_UNROLL_FULL
            for (int ji = 1; ji <= r + 1; ++ji) {
              const double z = AB(ji, r);
              AB(ji, r) = q2 * z + p2 * AB(ji, r + 1);
              AB(ji, r + 1) = q2 * AB(ji, r + 1) - p2 * z;
            }
            // accumulate transformations, ji = ll..kk
            FOR(ji) if (ji >= ll && ji <= kk) {
              const double z = EV(ji, r);
              EV(ji, r) = q2 * z + p2 * EV(ji, r + 1);
              EV(ji, r + 1) = q2 * EV(ji, r + 1) - p2 * z;
            }
          }
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
        // exceptional shift (:513-523)
        tt = tt + xx;
        FOR(ji) if (ji >= ll && ji <= jn) AB(ji, ji) = AB(ji, ji) - xx;
        const double ss = fabs(S.subdiag(jn)) + fabs(S.subdiag(n1));
        xx = C3 * ss;
        yy = xx;
        ww = -C1 * ss * ss;
      }
      in = in + 1;
      // look for two consecutive small sub-diagonal elements: m from n2 down to lb
      double pp = 0.0, qq = 0.0, rr = 0.0;
      int jstart = lb;
      {
        bool stop = false;
        SS_UNROLL_FULL
        for (int m = N - 2; m >= 1; --m) {
          if (!stop && m <= n2 && m >= lb) {
            const double zz = AB(m, m);
            double r2 = xx - zz;
            double s2 = yy - zz;
            double p2 = (r2 * s2 - ww) / AB(m + 1, m) + AB(m, m + 1);
            double q2 = AB(m + 1, m + 1) - zz - r2 - s2;
            r2 = AB(m + 2, m + 1);
            s2 = 1.0 / (fabs(p2) + fabs(q2) + fabs(r2));
            p2 = p2 * s2;
            q2 = q2 * s2;
            r2 = r2 * s2;
            pp = p2; qq = q2; rr = r2;
            jstart = m;
            if (m == lb) {
              stop = true;
            } else {
              // m > lb >= 1, so m - 1 >= 1
              const double am1 = (m >= 2) ? AB((m >= 2 ? m : 2), (m >= 2 ? m - 1 : 1)) : 0.0;
              const double dm1 = (m >= 2) ? AB((m >= 2 ? m - 1 : 1), (m >= 2 ? m - 1 : 1)) : 0.0;
              const double uu = fabs(am1) * (fabs(q2) + fabs(r2));
              const double vv = fabs(p2) * (fabs(dm1) + fabs(zz) + fabs(AB(m + 1, m + 1)));
              if (uu <= Tol * vv) stop = true;
            }
          }
        }
      }
      // AB(ji+2,ji) = 0; AB(jj,jj-2) = AB(jj,jj-3) = 0 for jj = ji+3..jn
      SS_UNROLL_FULL
      for (int m = 1; m <= N - 2; ++m) if (m == jstart) AB(m + 2, m) = 0.0;
      SS_UNROLL_FULL
      for (int jj = 4; jj <= N; ++jj) if (jj >= jstart + 3 && jj <= jn) {
        AB(jj, jj - 2) = 0.0;
        AB(jj, jj - 3) = 0.0;
      }
      // double QR step: ka = jstart .. n1  (:557-633)
      SS_UNROLL_FULL
      for (int ka = 1; ka <= N - 1; ++ka) {
        if (ka >= jstart && ka <= n1) {
          const bool not_last = (ka != n1);
          double ss;
          bool skip = false;
          if (ka == jstart) {
            ss = ss_sign(sqrt(pp * pp + qq * qq + rr * rr), pp);
            if (ka >= 2) {
              if (lb != jstart) AB(ka, (ka >= 2 ? ka - 1 : 1)) = -AB(ka, (ka >= 2 ? ka - 1 : 1));
            }
          } else {
            // ka > jstart >= 1  =>  ka >= 2
            const int km1 = (ka >= 2) ? ka - 1 : 1;
            pp = AB(ka, km1);
            qq = AB(ka + 1, km1);
            rr = 0.0;
            if (ka + 2 <= N) { if (not_last) rr = AB((ka + 2 <= N ? ka + 2 : N), km1); }
            xx = fabs(pp) + fabs(qq) + fabs(rr);
            if (xx == 0.0) {
              skip = true;
              ss = 0.0;
            } else {
              pp = pp / xx;
              qq = qq / xx;
              rr = rr / xx;
              ss = ss_sign(sqrt(pp * pp + qq * qq + rr * rr), pp);
              AB(ka, km1) = -ss * xx;
            }
          }
          if (!skip) {
            pp = pp + ss;
            ss = 1.0 / ss;
            xx = pp * ss;
            yy = qq * ss;
            const double zz = rr * ss;
            pp = 1.0 / pp;
            qq = qq * pp;
            rr = rr * pp;
            const int k2 = (ka + 2 <= N) ? ka + 2 : N;   // only dereferenced when not_last
            // row modification, jj = ka..N
            SS_UNROLL_FULL
            for (int jj = ka; jj <= N; ++jj) {
              double p = AB(ka, jj) + qq * AB(ka + 1, jj);
              if (ka + 2 <= N) {
                if (not_last) {
                  p = p + rr * AB(k2, jj);
                  AB(k2, jj) = AB(k2, jj) - p * zz;
                }
              }
              AB(ka + 1, jj) = AB(ka + 1, jj) - p * yy;
              AB(ka, jj) = AB(ka, jj) - p * xx;
            }
            // column modification, ji = 1..min(jn, ka+3)
            SS_UNROLL_FULL
            for (int ji = 1; ji <= (ka + 3 < N ? ka + 3 : N); ++ji) {
              if (ji <= jn) {
                double p = xx * AB(ji, ka) + yy * AB(ji, ka + 1);
                if (ka + 2 <= N) {
                  if (not_last) {
                    p = p + zz * AB(ji, k2);
                    AB(ji, k2) = AB(ji, k2) - p * rr;
                  }
                }
                AB(ji, ka + 1) = AB(ji, ka + 1) - p * qq;
                AB(ji, ka) = AB(ji, ka) - p;
              }
            }
            // accumulate transformations, ji = ll..kk
            FOR(ji) if (ji >= ll && ji <= kk) {
              double p = xx * EV(ji, ka) + yy * EV(ji, ka + 1);
              if (ka + 2 <= N) {
                if (not_last) {
                  p = p + zz * EV(ji, k2);
                  EV(ji, k2) = EV(ji, k2) - p * rr;
                }
              }
              EV(ji, ka + 1) = EV(ji, ka + 1) - p * qq;
              EV(ji, ka) = EV(ji, ka) - p;
            }
          }
        }
      }
    }
  }

  // ---- back-substitution (:636-717) ----
  if (!is_error) {
    if (rnorm != 0.0) {
      FORDOWN(jn2) {
        AB(jn2, jn2) = 1.0;
        SS_UNROLL_FULL
        for (int ji = jn2 - 1; ji >= 1; --ji) {
          double ww = AB(ji, ji) - EVAL(jn2);
          if (fabs(ww) < fabs(Tol * rnorm)) ww = ss_sign(Tol * rnorm, ww);
          double rr = AB(ji, jn2);
          // n2 of the reference equals ji+1 here
          SS_UNROLL_FULL
          for (int jj = ji + 1; jj <= jn2 - 1; ++jj) rr += AB(ji, jj) * AB(jj, jn2);
          AB(ji, jn2) = -rr / ww;
        }
      }
      FOR(ji) if (ji < ll || ji > kk) {
        SS_UNROLL_FULL
        for (int jj = ji; jj <= N; ++jj) EV(ji, jj) = AB(ji, jj);
      }
      if (kk != 0) {
        FORDOWN(jj) if (jj >= ll) {
          FOR(ji) if (ji >= ll && ji <= kk) {
            double zz = 0.0;
            SS_UNROLL_FULL
            for (int m = 1; m <= jj; ++m) if (m >= ll && m <= kk) zz += EV(ji, m) * AB(m, jj);
            EV(ji, jj) = zz;
          }
        }
      }
    }
    FOR(ji) if (ji >= ll && ji <= kk) {
      FOR(jj) EV(ji, jj) = EV(ji, jj) * WK(ji);
    }
    // interchange rows if permutations occurred (rare)
    FORDOWN(ji) if (ji <= ll - 1) {
      const int jj = (int)(WK(ji) + 0.5);
      if (ji < jj) {
        FOR(c) if (c == jj) {
          FOR(m) { const double t = EV(ji, m); EV(ji, m) = EV(c, m); EV(c, m) = t; }
        }
      }
    }
    FOR(ji) if (ji >= kk + 1) {
      const int jj = (int)(WK(ji) + 0.5);
      if (ji != jj) {
        FOR(c) if (c == jj) {
          FOR(m) { const double t = EV(ji, m); EV(ji, m) = EV(c, m); EV(c, m) = t; }
        }
      }
    }
  }
  return ierr;
#undef AB
#undef EV
#undef WK
#undef EVAL
#undef FOR
#undef FORDOWN
}

}  // namespace ss
