/* oracle_radtool.h - plain-C restatement of the reference's radtool layer.
 *
 * TEST INFRASTRUCTURE ONLY (see spartacus_oracle.h).  One matrix at a time
 * (the reference's leading "nmat" spectral dimension is looped by the caller);
 * matrices are column-major like the Fortran, 0-based.  Each function cites
 * the reference lines it restates.  Operation ORDER inside each routine is the
 * reference's, so results agree with the compiled reference to rounding.
 */
#ifndef ORACLE_RADTOOL_H
#define ORACLE_RADTOOL_H

#include <math.h>
#include <string.h>
#include <float.h>

/* Arithmetic type of the restatement.  Default: or_real = double (jprb = jprd,
 * utilities/parkind1.F90:42-49) - the checker proper.  With -DORACLE_QUAD the
 * same source is compiled in IEEE binary128 (liboracle_quad.so): an
 * "exact-arithmetic" rendering of the reference's algorithm on the same fp64
 * inputs and constants, used by the tests to measure how much of a difference
 * between two fp64 implementations is the reference's own rounding noise. */
#ifdef ORACLE_QUAD
#include <quadmath.h>
typedef __float128 or_real;
#define OR_SQRT sqrtq
#define OR_EXP expq
#define OR_ATAN atanq
#define OR_FABS fabsq
#define OR_FMAX fmaxq
#define OR_COPYSIGN copysignq
#define OR_EPS FLT128_EPSILON
#define OR_QR_CAP 120   /* converging to 1e-34 takes more sweeps than the fp64 cap of 30 */
#else
typedef double or_real;
#define OR_SQRT sqrt
#define OR_EXP exp
#define OR_ATAN atan
#define OR_FABS fabs
#define OR_FMAX fmax
#define OR_COPYSIGN copysign
#define OR_EPS DBL_EPSILON
#define OR_QR_CAP 30    /* radtool_eigen_decomposition.F90:498 */
#endif

/* element (i,j) of a column-major matrix with leading dimension ld */
#define E(a, i, j, ld) ((a)[(i) + (size_t)(j) * (ld)])

/* largest diffuse order: (nreg+1)*ns with nreg = 3, ns = 16 */
#define OR_MAXN 64
/* largest system in direct_diffuse_part: 2*nreg*ns + nreg */
#define OR_MAXBIG (2 * 48 + 3)

/* Fortran sign(a,b) */
static inline or_real or_sign(or_real a, or_real b) { return OR_COPYSIGN(OR_FABS(a), b); }

/* ---- matrix products (radtool/radtool_matrix.F90) ---- */

/* mat_x_mat :248-331 (dense branch): C = A*B, all m x m.  Loop order j2,j3,j1. */
static void or_mat_x_mat(int m, const or_real *A, const or_real *B, or_real *C) {
  for (int k = 0; k < m * m; ++k) C[k] = 0.0;
  for (int j2 = 0; j2 < m; ++j2)
    for (int j3 = 0; j3 < m; ++j3)
      for (int j1 = 0; j1 < m; ++j1)
        E(C, j1, j2, m) += E(A, j1, j3, m) * E(B, j3, j2, m);
}

/* rect_mat_x_mat :335-374: C(k x m) = A(k x l) * B(l x m).  Order j2,j1,j3. */
static void or_rect_mat_x_mat(int k, int l, int m, const or_real *A, const or_real *B, or_real *C) {
  for (int j2 = 0; j2 < m; ++j2)
    for (int j1 = 0; j1 < k; ++j1) {
      or_real s = 0.0;
      for (int j3 = 0; j3 < l; ++j3) s += E(A, j1, j3, k) * E(B, j3, j2, l);
      E(C, j1, j2, k) = s;
    }
}

/* mat_x_vec :71-119 / rect_mat_x_vec :125-160: y(k) = A(k x l) x(l), lda given */
static void or_rect_mat_x_vec(int k, int l, const or_real *A, int lda, const or_real *x, or_real *y) {
  for (int j1 = 0; j1 < k; ++j1) {
    or_real s = 0.0;
    for (int j2 = 0; j2 < l; ++j2) s += E(A, j1, j2, lda) * x[j2];
    y[j1] = s;
  }
}
static void or_mat_x_vec(int m, const or_real *A, const or_real *x, or_real *y) {
  or_rect_mat_x_vec(m, m, A, m, x, y);
}

/* ---- LU (no pivoting) (radtool_matrix.F90:982-1090) ---- */

/* lu_factorization :982-1017 */
static void or_lu_factorization(int m, const or_real *A, or_real *LU) {
  memcpy(LU, A, sizeof(or_real) * (size_t)m * m);
  for (int j2 = 0; j2 < m; ++j2) {
    for (int j1 = 0; j1 < j2; ++j1) {
      or_real s = E(LU, j1, j2, m);
      for (int j3 = 0; j3 < j1; ++j3) s -= E(LU, j1, j3, m) * E(LU, j3, j2, m);
      E(LU, j1, j2, m) = s;
    }
    for (int j1 = j2; j1 < m; ++j1) {
      or_real s = E(LU, j1, j2, m);
      for (int j3 = 0; j3 < j2; ++j3) s -= E(LU, j1, j3, m) * E(LU, j3, j2, m);
      E(LU, j1, j2, m) = s;
    }
    if (j2 != m - 1) {
      or_real s = 1.0 / E(LU, j2, j2, m);
      for (int j1 = j2 + 1; j1 < m; ++j1) E(LU, j1, j2, m) *= s;
    }
  }
}

/* lu_substitution :1024-1050 */
static void or_lu_substitution(int m, const or_real *LU, const or_real *b, or_real *x) {
  for (int j = 0; j < m; ++j) x[j] = b[j];
  for (int j2 = 1; j2 < m; ++j2)
    for (int j1 = 0; j1 < j2; ++j1) x[j2] -= x[j1] * E(LU, j2, j1, m);
  for (int j2 = m - 1; j2 >= 0; --j2) {
    for (int j1 = j2 + 1; j1 < m; ++j1) x[j2] -= x[j1] * E(LU, j2, j1, m);
    x[j2] /= E(LU, j2, j2, m);
  }
}

/* invert :1203-1231 = lu_factorization + lu_invert :1057-1090 (always generic) */
static void or_invert(int m, const or_real *A, or_real *X) {
  or_real LU[OR_MAXN * OR_MAXN];
  or_lu_factorization(m, A, LU);
  for (int j3 = 0; j3 < m; ++j3) {
    or_real *x = X + (size_t)j3 * m;
    for (int j2 = 0; j2 < m; ++j2) x[j2] = 0.0;
    x[j3] = 1.0;
    for (int j2 = 1; j2 < m; ++j2)
      for (int j1 = 0; j1 < j2; ++j1) x[j2] -= x[j1] * E(LU, j2, j1, m);
    for (int j2 = m - 1; j2 >= 0; --j2) {
      for (int j1 = j2 + 1; j1 < m; ++j1) x[j2] -= x[j1] * E(LU, j2, j1, m);
      x[j2] /= E(LU, j2, j2, m);
    }
  }
}

/* solve_rect_mat :1119-1137: always generic LU; B, X are m x l */
static void or_solve_rect_mat(int m, int l, const or_real *A, const or_real *B, or_real *X) {
  or_real LU[OR_MAXBIG * OR_MAXBIG];
  or_lu_factorization(m, A, LU);
  for (int j = 0; j < l; ++j) or_lu_substitution(m, LU, B + (size_t)j * m, X + (size_t)j * m);
}

/* solve_vec :1143-1168 with solve_vec_2 (Cramer) :779-795 and solve_vec_3
 * (explicit LU) :827-860 */
static void or_solve_vec(int m, const or_real *A, const or_real *b, or_real *x) {
  if (m == 2) {
    or_real inv_det = 1.0 / (E(A, 0, 0, 2) * E(A, 1, 1, 2) - E(A, 0, 1, 2) * E(A, 1, 0, 2));
    or_real x0 = inv_det * (E(A, 1, 1, 2) * b[0] - E(A, 0, 1, 2) * b[1]);
    or_real x1 = inv_det * (E(A, 0, 0, 2) * b[1] - E(A, 1, 0, 2) * b[0]);
    x[0] = x0; x[1] = x1;
  } else if (m == 3) {
    or_real L21 = E(A, 1, 0, 3) / E(A, 0, 0, 3);
    or_real L31 = E(A, 2, 0, 3) / E(A, 0, 0, 3);
    or_real U22 = E(A, 1, 1, 3) - L21 * E(A, 0, 1, 3);
    or_real U23 = E(A, 1, 2, 3) - L21 * E(A, 0, 2, 3);
    or_real L32 = (E(A, 2, 1, 3) - L31 * E(A, 0, 1, 3)) / U22;
    or_real U33 = E(A, 2, 2, 3) - L31 * E(A, 0, 2, 3) - L32 * U23;
    or_real y2 = b[1] - L21 * b[0];
    or_real y3 = b[2] - L31 * b[0] - L32 * y2;
    or_real x3 = y3 / U33;
    or_real x2 = (y2 - U23 * x3) / U22;
    or_real x1 = (b[0] - E(A, 0, 1, 3) * x2 - E(A, 0, 2, 3) * x3) / E(A, 0, 0, 3);
    x[0] = x1; x[1] = x2; x[2] = x3;
  } else {
    or_real LU[OR_MAXN * OR_MAXN];
    or_real tmp[OR_MAXN];
    or_lu_factorization(m, A, LU);
    or_lu_substitution(m, LU, b, tmp);
    for (int j = 0; j < m; ++j) x[j] = tmp[j];
  }
}

/* solve_mat :1175-1197 (m=2 Cramer :801-821, m=3 explicit LU :866-906, else
 * solve_mat_n :1096-1112).  B, X are m x m; X may not alias A or B. */
static void or_solve_mat(int m, const or_real *A, const or_real *B, or_real *X) {
  if (m == 2 || m == 3) {
    for (int j = 0; j < m; ++j) or_solve_vec(m, A, B + (size_t)j * m, X + (size_t)j * m);
  } else {
    or_real LU[OR_MAXN * OR_MAXN];
    or_lu_factorization(m, A, LU);
    for (int j = 0; j < m; ++j) or_lu_substitution(m, LU, B + (size_t)j * m, X + (size_t)j * m);
  }
}

/* ---- eigen_decomposition_real (radtool/radtool_eigen_decomposition.F90:51-828)
 * One matrix, always fp64 (the reference's locals are jprd).  1-based index
 * macros keep the transcription line-for-line checkable.  Returns 0, or the
 * index of the eigenvalue at which the 30-iteration cap was hit (:498-510). */
static int or_eigen_decomposition_real(int norder, const or_real *amat, or_real *eigenvalue,
                                       or_real *eigenvector) {
  const or_real Tol = OR_EPS;
  const or_real C1 = 0.4375, C2 = 0.5, C3 = 0.75, C4 = 0.95, C5 = 16.0, C6 = 256.0;
  const int n = norder;
#define AB(i, j) abal[((i)-1) + (size_t)((j)-1) * n]
#define EV(i, j) eigenvec[((i)-1) + (size_t)((j)-1) * n]
#define WK(i) wkd[(i)-1]
#define EVAL(i) eigenval[(i)-1]

  if (n > 2) {
    or_real abal[OR_MAXN * OR_MAXN], eigenvec[OR_MAXN * OR_MAXN];
    or_real eigenval[OR_MAXN], wkd[2 * OR_MAXN];
    int ji, jj, jn, ka, kk, ll, lb = 0, lll, n1, n2, in, kkk;
    or_real column, ff, gg, hh, pp = 0, qq = 0, rr = 0, tmp, rnorm, row, ss, scale, tt;
    or_real uu, vv, ww, xx, yy, zz;
    int not_finished, not_last, is_error = 0, ierr = 0;

    for (ji = 1; ji <= n; ++ji) EVAL(ji) = 0.0;
    for (ji = 0; ji < n * n; ++ji) eigenvec[ji] = 0.0;
    for (ji = 1; ji <= n; ++ji) EV(ji, ji) = 1.0;
    memcpy(abal, amat, sizeof(or_real) * (size_t)n * n);
    for (ji = 0; ji < 2 * n; ++ji) wkd[ji] = 0.0;

    rnorm = 0.0;
    ll = 1;
    kk = n;

    /* :182-222 search for rows isolating an eigenvalue and push them down */
    not_finished = 1;
    while (not_finished) {
      not_finished = 0;
      kkk = kk;
      for (jj = kkk; jj >= 1; --jj) {
        row = 0.0;
        for (ji = 1; ji <= kk; ++ji)
          if (ji != jj) row += OR_FABS(AB(jj, ji));
        if (row == 0.0) {
          WK(kk) = jj;
          /* :201 "if (ji /= kk)" is always true (ji == kk+1 after the loop) */
          for (ji = 1; ji <= kk; ++ji) {
            tmp = AB(ji, jj); AB(ji, jj) = AB(ji, kk); AB(ji, kk) = tmp;
          }
          for (ji = ll; ji <= n; ++ji) {
            tmp = AB(jj, ji); AB(jj, ji) = AB(kk, ji); AB(kk, ji) = tmp;
          }
          kk = kk - 1;
          not_finished = 1;
          break;
        } else {
          not_finished = 0;
        }
      }
    }

    /* :227-262 search for columns isolating an eigenvalue and push them left */
    not_finished = 1;
    while (not_finished) {
      not_finished = 0;
      lll = ll;
      for (jj = lll; jj <= kk; ++jj) {
        column = 0.0;
        for (ji = ll; ji <= kk; ++ji)
          if (ji != jj) column += OR_FABS(AB(ji, jj));
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
          not_finished = 1;
          break;
        } else {
          not_finished = 0;
        }
      }
    }

    /* :264-310 balance the submatrix in rows ll through kk */
    for (ji = ll; ji <= kk; ++ji) WK(ji) = 1.0;
    not_finished = 1;
    while (not_finished) {
      not_finished = 0;
      for (ji = ll; ji <= kk; ++ji) {
        column = 0.0;
        row = 0.0;
        for (jj = ll; jj <= kk; ++jj)
          if (jj != ji) {
            column += OR_FABS(AB(jj, ji));
            row += OR_FABS(AB(ji, jj));
          }
        ff = 1.0;
        gg = row / C5;
        hh = column + row;
        while (column < gg) { ff = ff * C5; column = column * C6; }
        gg = row * C5;
        while (column > gg) { ff = ff / C5; column = column / C6; }
        if ((column + row) / ff < C4 * hh) {
          WK(ji) = WK(ji) * ff;
          not_finished = 1;
          for (jj = ll; jj <= n; ++jj) AB(ji, jj) = AB(ji, jj) / ff;
          for (jj = 1; jj <= kk; ++jj) AB(jj, ji) = AB(jj, ji) * ff;
        }
      }
    }

    /* :312-395 reduce to Hessenberg form */
    if (kk - 1 >= ll + 1) {
      for (jn = ll + 1; jn <= kk - 1; ++jn) {
        hh = 0.0;
        WK(jn + n) = 0.0;
        scale = 0.0;
        for (ji = jn; ji <= kk; ++ji) scale += OR_FABS(AB(ji, jn - 1));
        if (scale != 0.0) {
          for (ji = kk; ji >= jn; --ji) {
            WK(ji + n) = AB(ji, jn - 1) / scale;
            hh += WK(ji + n) * WK(ji + n);
          }
          gg = -or_sign(OR_SQRT(hh), WK(jn + n));
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

    /* :397-408 */
    jn = 1;
    for (ji = 1; ji <= n; ++ji) {
      for (jj = jn; jj <= n; ++jj) rnorm += OR_FABS(AB(ji, jj));
      jn = ji;
      if (ji < ll || ji > kk) EVAL(ji) = AB(ji, ji);
    }
    jn = kk;
    tt = 0.0;

    /* :410-635 search for the eigenvalues */
    not_finished = 1;
    while (not_finished && jn >= ll) {
      not_finished = 0;
      in = 0;
      n1 = jn - 1;
      n2 = jn - 2;
      for (;;) { /* "do while (not_found)" :420 - left only through exits */
        /* :423-435 look for single small sub-diagonal element */
        for (ji = ll; ji <= jn; ++ji) {
          lb = jn + ll - ji;
          if (lb == ll) break;
          ss = OR_FABS(AB(lb - 1, lb - 1)) + OR_FABS(AB(lb, lb));
          if (ss == 0.0) ss = rnorm;
          if (OR_FABS(AB(lb, lb - 1)) < Tol * ss) break;
        }
        xx = AB(jn, jn);
        if (lb == jn) {
          /* one eigenvalue found */
          AB(jn, jn) = xx + tt;
          EVAL(jn) = AB(jn, jn);
          jn = n1;
          not_finished = 1;
          break;
        }
        yy = AB(n1, n1);
        ww = AB(jn, n1) * AB(n1, jn);
        if (lb == n1) {
          /* two eigenvalues found :449-496 */
          pp = (yy - xx) * C2;
          qq = pp * pp + ww;
          zz = OR_SQRT(OR_FABS(qq));
          AB(jn, jn) = xx + tt;
          xx = AB(jn, jn);
          AB(n1, n1) = yy + tt;
          zz = pp + or_sign(zz, pp);
          EVAL(n1) = xx + zz;
          EVAL(jn) = EVAL(n1);
          if (zz != 0.0) EVAL(jn) = xx - ww / zz;
          xx = AB(jn, n1);
          rr = 1.0 / OR_SQRT(xx * xx + zz * zz);
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
          not_finished = 1;
          break;
        }
        if (in == OR_QR_CAP) {
          /* :498-510 no convergence */
          ierr = jn;
          is_error = 1;
          not_finished = 0;
          break;
        }
        /* :512-523 exceptional shift */
        if (in == 10 || in == 20) {
          tt = tt + xx;
          for (ji = ll; ji <= jn; ++ji) AB(ji, ji) = AB(ji, ji) - xx;
          ss = OR_FABS(AB(jn, n1)) + OR_FABS(AB(n1, n2));
          xx = C3 * ss;
          yy = xx;
          ww = -C1 * ss * ss;
        }
        in = in + 1;
        /* :526-549 look for two consecutive small sub-diagonal elements */
        for (jj = lb; jj <= n2; ++jj) {
          ji = n2 + lb - jj;
          zz = AB(ji, ji);
          rr = xx - zz;
          ss = yy - zz;
          pp = (rr * ss - ww) / AB(ji + 1, ji) + AB(ji, ji + 1);
          qq = AB(ji + 1, ji + 1) - zz - rr - ss;
          rr = AB(ji + 2, ji + 1);
          ss = 1.0 / (OR_FABS(pp) + OR_FABS(qq) + OR_FABS(rr));
          pp = pp * ss;
          qq = qq * ss;
          rr = rr * ss;
          if (ji == lb) break;
          uu = OR_FABS(AB(ji, ji - 1)) * (OR_FABS(qq) + OR_FABS(rr));
          vv = OR_FABS(pp) * (OR_FABS(AB(ji - 1, ji - 1)) + OR_FABS(zz) + OR_FABS(AB(ji + 1, ji + 1)));
          if (uu <= Tol * vv) break;
        }
        AB(ji + 2, ji) = 0.0;
        for (jj = ji + 3; jj <= jn; ++jj) {
          AB(jj, jj - 2) = 0.0;
          AB(jj, jj - 3) = 0.0;
        }
        /* :557-633 or_real QR step on rows ji..jn; note the loop bound is the
         * value of ji on entry, while ji itself is reused inside */
        {
          const int ka_first = ji;
          for (ka = ka_first; ka <= n1; ++ka) {
            not_last = (ka != n1);
            if (ka == ka_first) {
              ss = or_sign(OR_SQRT(pp * pp + qq * qq + rr * rr), pp);
              if (lb != ka_first) AB(ka, ka - 1) = -AB(ka, ka - 1);
            } else {
              pp = AB(ka, ka - 1);
              qq = AB(ka + 1, ka - 1);
              rr = 0.0;
              if (not_last) rr = AB(ka + 2, ka - 1);
              xx = OR_FABS(pp) + OR_FABS(qq) + OR_FABS(rr);
              if (xx == 0.0) continue;
              pp = pp / xx;
              qq = qq / xx;
              rr = rr / xx;
              ss = or_sign(OR_SQRT(pp * pp + qq * qq + rr * rr), pp);
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
            /* row modification */
            for (jj = ka; jj <= n; ++jj) {
              pp = AB(ka, jj) + qq * AB(ka + 1, jj);
              if (not_last) {
                pp = pp + rr * AB(ka + 2, jj);
                AB(ka + 2, jj) = AB(ka + 2, jj) - pp * zz;
              }
              AB(ka + 1, jj) = AB(ka + 1, jj) - pp * yy;
              AB(ka, jj) = AB(ka, jj) - pp * xx;
            }
            /* column modification */
            {
              int top = (jn < ka + 3) ? jn : ka + 3;
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
            /* accumulate transformations */
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
      } /* do while not_found */
    }   /* do while not_finished */

    /* :636-717 back-substitute */
    if (!is_error) {
      if (rnorm != 0.0) {
        for (jn = n; jn >= 1; --jn) {
          n2 = jn;
          AB(jn, jn) = 1.0;
          for (ji = jn - 1; ji >= 1; --ji) {
            ww = AB(ji, ji) - EVAL(jn);
            if (OR_FABS(ww) < OR_FABS(Tol * rnorm)) ww = or_sign(Tol * rnorm, ww);
            rr = AB(ji, jn);
            for (jj = n2; jj <= jn - 1; ++jj) rr += AB(ji, jj) * AB(jj, jn);
            AB(ji, jn) = -rr / ww;
            n2 = ji;
          }
        }
        for (ji = 1; ji <= n; ++ji)
          if (ji < ll || ji > kk)
            for (jj = ji; jj <= n; ++jj) EV(ji, jj) = AB(ji, jj);
        /* :672 "if (kk /= 0.0)": for kk == 0 the loops below are empty anyway */
        if (kk != 0) {
          for (jj = n; jj >= ll; --jj) {
            for (ji = ll; ji <= kk; ++ji) {
              int top = (jj < kk) ? jj : kk;
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
    for (ji = 0; ji < n; ++ji) eigenvalue[ji] = eigenval[ji];
    memcpy(eigenvector, eigenvec, sizeof(or_real) * (size_t)n * n);
    return ierr;
  } else if (n == 2) {
    /* :770-803 closed form */
    const or_real a11 = E(amat, 0, 0, 2), a21 = E(amat, 1, 0, 2);
    const or_real a12 = E(amat, 0, 1, 2), a22 = E(amat, 1, 1, 2);
    or_real discriminant = (a11 - a22) * (a11 - a22) + 4.0 * a12 * a21;
    or_real half_sqrt_disc, rnorm;
    int ierr = (discriminant < 0.0) ? 2 : 0;
    eigenvalue[0] = 0.5 * (a11 + a22);
    eigenvalue[1] = eigenvalue[0];
    half_sqrt_disc = 0.5 * OR_SQRT(discriminant);
    if (a11 >= a22) {
      eigenvalue[0] = eigenvalue[0] + half_sqrt_disc;
      eigenvalue[1] = eigenvalue[1] - half_sqrt_disc;
    } else {
      eigenvalue[0] = eigenvalue[0] - half_sqrt_disc;
      eigenvalue[1] = eigenvalue[1] + half_sqrt_disc;
    }
    E(eigenvector, 0, 0, 2) = 1.0;
    E(eigenvector, 1, 1, 2) = 1.0;
    if (a11 == a22 && (a21 == 0.0 || a12 == 0.0)) {
      /* precedence exactly as written in the reference :795-797 */
      rnorm = 1.0 / (Tol * OR_FABS(a11) + OR_FABS(a21) + OR_FABS(a12) + OR_FABS(a22));
      E(eigenvector, 1, 0, 2) = a21 * rnorm;
      E(eigenvector, 0, 1, 2) = a12 * rnorm;
    } else {
      E(eigenvector, 1, 0, 2) = a21 / (eigenvalue[0] - a22);
      E(eigenvector, 0, 1, 2) = a12 / (eigenvalue[1] - a11);
    }
    return ierr;
  } else if (n == 1) {
    eigenvalue[0] = amat[0];
    eigenvector[0] = 1.0;
    return 0;
  }
  return 1;
#undef AB
#undef EV
#undef WK
#undef EVAL
}

/* ---- schur_invert_sw (radtool/radtool_schur.F90:32-53) ---- */
static void or_schur_invert_sw(int n0, int n1, const or_real *g0, const or_real *g1,
                               const or_real *g2, const or_real *g3, or_real *g0i, or_real *g1i,
                               or_real *g2i, or_real *g3i) {
  or_real t1[OR_MAXN * OR_MAXN], t2[OR_MAXN * OR_MAXN], t3[OR_MAXN * OR_MAXN];
  or_invert(n0, g0, g0i);
  /* g1i = invert(g1 - g2 * solve_mat(g1, g2)) */
  or_solve_mat(n1, g1, g2, t1);
  or_mat_x_mat(n1, g2, t1, t2);
  for (int k = 0; k < n1 * n1; ++k) t3[k] = g1[k] - t2[k];
  or_invert(n1, t3, g1i);
  /* g2i = g1i * (g2 * invert(g1)) */
  or_invert(n1, g1, t1);
  or_mat_x_mat(n1, g2, t1, t2);
  or_mat_x_mat(n1, g1i, t2, g2i);
  /* g3i = (g1i - g2i) * (g3 * g0i) */
  or_rect_mat_x_mat(n1, n0, n0, g3, g0i, t1);
  for (int k = 0; k < n1 * n1; ++k) t2[k] = g1i[k] - g2i[k];
  or_rect_mat_x_mat(n1, n1, n0, t2, t1, g3i);
}

/* ---- direct_diffuse_part (radtool_calc_matrices_sw_eig.F90:303-386) ---- */
static void or_direct_diffuse_part(int ndiff, int ndir, const or_real *exp_lambda_dz,
                                   const or_real *exp_eigenval_dir_dz, const or_real *g0,
                                   const or_real *g1, const or_real *g2, const or_real *g3,
                                   const or_real *g4, or_real *s_up, or_real *s_dn) {
  const int nb = 2 * ndiff + ndir;
  static __thread or_real g_d[OR_MAXBIG * OR_MAXBIG];
  or_real g_d2[OR_MAXN * OR_MAXBIG], cprime_dir[OR_MAXBIG * 3], rhs[OR_MAXBIG * 3];
  for (int k = 0; k < nb * nb; ++k) g_d[k] = 0.0;
  for (int j = 0; j < ndiff; ++j)
    for (int i = 0; i < ndiff; ++i) {
      E(g_d, i, j, nb) = E(g1, i, j, ndiff);
      E(g_d, ndiff + i, j, nb) = E(g2, i, j, ndiff) * exp_lambda_dz[j];
      E(g_d, i, ndiff + j, nb) = E(g_d, ndiff + i, j, nb);
      E(g_d, ndiff + i, ndiff + j, nb) = E(g1, i, j, ndiff);
    }
  for (int j = 0; j < ndir; ++j)
    for (int i = 0; i < ndir; ++i) E(g_d, 2 * ndiff + i, 2 * ndiff + j, nb) = E(g0, i, j, ndir);
  for (int j = 0; j < ndir; ++j)
    for (int i = 0; i < ndiff; ++i) {
      E(g_d, i, 2 * ndiff + j, nb) = E(g3, i, j, ndiff) * exp_eigenval_dir_dz[j];
      E(g_d, ndiff + i, 2 * ndiff + j, nb) = E(g4, i, j, ndiff);
    }
  for (int k = 0; k < nb * ndir; ++k) rhs[k] = 0.0;
  for (int jj = 0; jj < ndir; ++jj) E(rhs, 2 * ndiff + jj, jj, nb) = 1.0;

  or_solve_rect_mat(nb, ndir, g_d, rhs, cprime_dir);

  for (int j = 0; j < ndiff; ++j)
    for (int i = 0; i < ndiff; ++i) {
      E(g_d2, i, j, ndiff) = E(g1, i, j, ndiff) * exp_lambda_dz[j];
      E(g_d2, i, ndiff + j, ndiff) = E(g2, i, j, ndiff);
    }
  for (int j = 0; j < ndir; ++j)
    for (int i = 0; i < ndiff; ++i) E(g_d2, i, 2 * ndiff + j, ndiff) = E(g3, i, j, ndiff);
  or_rect_mat_x_mat(ndiff, nb, ndir, g_d2, cprime_dir, s_up);

  for (int j = 0; j < ndiff; ++j)
    for (int i = 0; i < ndiff; ++i) {
      E(g_d2, i, j, ndiff) = E(g2, i, j, ndiff);
      E(g_d2, i, ndiff + j, ndiff) = E(g1, i, j, ndiff) * exp_lambda_dz[j];
    }
  for (int j = 0; j < ndir; ++j)
    for (int i = 0; i < ndiff; ++i)
      E(g_d2, i, 2 * ndiff + j, ndiff) = E(g4, i, j, ndiff) * exp_eigenval_dir_dz[j];
  or_rect_mat_x_mat(ndiff, nb, ndir, g_d2, cprime_dir, s_dn);
}

/* Shared first half of calc_matrices_{sw,lw}_eig: eigen-decomposition of
 * (gamma1-gamma2)(gamma1+gamma2) and the diffuse R, T
 * (sw :180-221, lw :142-176).  Outputs kept for the callers. */
typedef struct {
  or_real lambda[OR_MAXN], exp_lambda_dz[OR_MAXN];
  or_real g1[OR_MAXN * OR_MAXN], g2[OR_MAXN * OR_MAXN];
  or_real g1_d[OR_MAXN * OR_MAXN], g2_d[OR_MAXN * OR_MAXN];
  or_real cprime_upper[OR_MAXN * OR_MAXN], cprime_lower[OR_MAXN * OR_MAXN];
} or_diffuse_t;

static int or_diffuse_part(int n, or_real dz, const or_real *gamma1, const or_real *gamma2,
                           or_real *reflectance, or_real *transmittance, or_diffuse_t *w) {
  or_real gamma_diff[OR_MAXN * OR_MAXN], gamma_sum[OR_MAXN * OR_MAXN];
  or_real gamma_product[OR_MAXN * OR_MAXN], eigenval_prod[OR_MAXN];
  or_real eigenvec_prod[OR_MAXN * OR_MAXN], tmp_mat[OR_MAXN * OR_MAXN];
  or_real t1[OR_MAXN * OR_MAXN], t2[OR_MAXN * OR_MAXN];
  const int nn = n * n;
  int ierr;
  for (int k = 0; k < nn; ++k) {
    gamma_diff[k] = gamma1[k] - gamma2[k];
    gamma_sum[k] = gamma1[k] + gamma2[k];
  }
  or_mat_x_mat(n, gamma_diff, gamma_sum, gamma_product);
  ierr = or_eigen_decomposition_real(n, gamma_product, eigenval_prod, eigenvec_prod);
  for (int j = 0; j < n; ++j) {
    w->lambda[j] = OR_SQRT(OR_FMAX(0.0, eigenval_prod[j]));
    w->exp_lambda_dz[j] = OR_EXP(-w->lambda[j] * dz);
  }
  /* tmp_mat = -solve_mat(gamma_diff, eigenvec_prod) * diag(lambda) */
  or_solve_mat(n, gamma_diff, eigenvec_prod, tmp_mat);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      or_real t = -E(tmp_mat, i, j, n);
      t = t * w->lambda[j];
      E(w->g1, i, j, n) = E(eigenvec_prod, i, j, n) + t;
      E(w->g2, i, j, n) = E(eigenvec_prod, i, j, n) - t;
    }
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      E(w->g1_d, i, j, n) = E(w->g1, i, j, n) * w->exp_lambda_dz[j];
      E(w->g2_d, i, j, n) = E(w->g2, i, j, n) * w->exp_lambda_dz[j];
    }
  /* cprime_lower = invert(g1 - g2_d * solve_mat(g1, g2_d)) */
  or_solve_mat(n, w->g1, w->g2_d, t1);
  or_mat_x_mat(n, w->g2_d, t1, t2);
  for (int k = 0; k < nn; ++k) t1[k] = w->g1[k] - t2[k];
  or_invert(n, t1, w->cprime_lower);
  /* cprime_upper = -solve_mat(g1, g2_d * cprime_lower) */
  or_mat_x_mat(n, w->g2_d, w->cprime_lower, t1);
  or_solve_mat(n, w->g1, t1, t2);
  for (int k = 0; k < nn; ++k) w->cprime_upper[k] = -t2[k];
  /* Eq. 49 */
  or_mat_x_mat(n, w->g1_d, w->cprime_upper, t1);
  or_mat_x_mat(n, w->g2, w->cprime_lower, t2);
  for (int k = 0; k < nn; ++k) reflectance[k] = t1[k] + t2[k];
  or_mat_x_mat(n, w->g2, w->cprime_upper, t1);
  or_mat_x_mat(n, w->g1_d, w->cprime_lower, t2);
  for (int k = 0; k < nn; ++k) transmittance[k] = t1[k] + t2[k];
  return ierr;
}

/* ---- calc_matrices_sw_eig (radtool_calc_matrices_sw_eig.F90:30-298) ---- */
static int or_calc_matrices_sw_eig(int ndiff, int ndir, or_real dz, or_real mu0,
                                   const or_real *gamma0, const or_real *gamma1,
                                   const or_real *gamma2, const or_real *gamma3,
                                   or_real *reflectance, or_real *transmittance, or_real *s_up,
                                   or_real *s_dn, or_real *trans_dir, or_real *int_dir,
                                   or_real *int_diff, or_real *int_dir_diff) {
  static __thread or_diffuse_t w;
  const int n = ndiff, nn = ndiff * ndiff;
  or_real eigenval_dir[3], g0[9], g0_inv[9], g0e[9], exp_dir[3];
  or_real gamma3_g0[OR_MAXN * 3], g3[OR_MAXN * 3], g4[OR_MAXN * 3];
  or_real gamma1_d[OR_MAXN * OR_MAXN], gamma2_inv_gamma1_d[OR_MAXN * OR_MAXN];
  or_real tmp_mat[OR_MAXN * OR_MAXN], t1[OR_MAXN * OR_MAXN];
  or_real v1[OR_MAXN], v2[OR_MAXN];
  or_real gamma0i[9], gamma1i[OR_MAXN * OR_MAXN], gamma2i[OR_MAXN * OR_MAXN];
  or_real gamma3i[OR_MAXN * 3];
  int ierr;
  (void)mu0;

  /* Sections 1-2 */
  ierr = or_diffuse_part(n, dz, gamma1, gamma2, reflectance, transmittance, &w);

  /* Section 3 :223-229: trans_dir = g0 diag(OR_EXP(eigenval_dir*dz)) g0^-1 */
  ierr += or_eigen_decomposition_real(ndir, gamma0, eigenval_dir, g0);
  or_invert(ndir, g0, g0_inv);
  for (int j = 0; j < ndir; ++j) {
    exp_dir[j] = OR_EXP(eigenval_dir[j] * dz);
    for (int i = 0; i < ndir; ++i) E(g0e, i, j, ndir) = E(g0, i, j, ndir) * exp_dir[j];
  }
  or_mat_x_mat(ndir, g0e, g0_inv, trans_dir);

  /* Section 4 :231-253 */
  or_rect_mat_x_mat(n, ndir, ndir, gamma3, g0, gamma3_g0);
  memcpy(gamma1_d, gamma1, sizeof(or_real) * (size_t)nn);
  for (int jd = 0; jd < ndir; ++jd) {
    for (int jo = 0; jo < n; ++jo) E(gamma1_d, jo, jo, n) = E(gamma1, jo, jo, n) + eigenval_dir[jd];
    or_invert(n, gamma1_d, t1);
    or_mat_x_mat(n, gamma2, t1, gamma2_inv_gamma1_d);
    or_mat_x_mat(n, gamma2_inv_gamma1_d, gamma2, t1);
    for (int k = 0; k < nn; ++k) tmp_mat[k] = gamma1[k] - t1[k];
    for (int jo = 0; jo < n; ++jo) E(tmp_mat, jo, jo, n) -= eigenval_dir[jd];
    for (int jo = 0; jo < n; ++jo) E(gamma2_inv_gamma1_d, jo, jo, n) -= 1.0;
    or_mat_x_vec(n, gamma2_inv_gamma1_d, gamma3_g0 + (size_t)jd * n, v1);
    or_solve_vec(n, tmp_mat, v1, g4 + (size_t)jd * n);
    or_mat_x_vec(n, gamma2, g4 + (size_t)jd * n, v1);
    for (int i = 0; i < n; ++i) v1[i] = gamma3_g0[(size_t)jd * n + i] + v1[i];
    or_solve_vec(n, gamma1_d, v1, v2);
    for (int i = 0; i < n; ++i) g3[(size_t)jd * n + i] = -v2[i];
  }

  or_direct_diffuse_part(n, ndir, w.exp_lambda_dz, exp_dir, g0, w.g1, w.g2, g3, g4, s_up, s_dn);

  /* :289-296 integrated-flux matrices */
  or_schur_invert_sw(ndir, n, gamma0, gamma1, gamma2, gamma3, gamma0i, gamma1i, gamma2i, gamma3i);
  for (int k = 0; k < ndir * ndir; ++k) int_dir[k] = -gamma0i[k];
  for (int k = 0; k < nn; ++k) int_diff[k] = gamma2i[k] - gamma1i[k];
  for (int k = 0; k < n * ndir; ++k) int_dir_diff[k] = 2.0 * gamma3i[k];
  return ierr;
}

/* ---- calc_matrices_lw_eig (radtool_calc_matrices_lw_eig.F90:32-230) ---- */
static int or_calc_matrices_lw_eig(int n, or_real dz, const or_real *gamma1, const or_real *gamma2,
                                   const or_real *emiss_rate, or_real *reflectance,
                                   or_real *transmittance, or_real *source, or_real *int_flux,
                                   or_real *int_flux_source) {
  static __thread or_diffuse_t w;
  const int nn = n * n;
  or_real gamma2_inv_gamma1[OR_MAXN * OR_MAXN], tmp_mat[OR_MAXN * OR_MAXN];
  or_real t1[OR_MAXN * OR_MAXN];
  or_real inv_gamma_b[OR_MAXN], inv_g1_inv_gamma_b[OR_MAXN], tmp_vec[OR_MAXN];
  or_real cb_prime[OR_MAXN], v1[OR_MAXN];
  int ierr = or_diffuse_part(n, dz, gamma1, gamma2, reflectance, transmittance, &w);

  /* Section 3 :178-195 */
  or_invert(n, gamma1, t1);
  or_mat_x_mat(n, gamma2, t1, gamma2_inv_gamma1);
  or_mat_x_mat(n, gamma2_inv_gamma1, gamma2, t1);
  for (int k = 0; k < nn; ++k) tmp_mat[k] = gamma1[k] - t1[k];
  for (int jo = 0; jo < n; ++jo) E(gamma2_inv_gamma1, jo, jo, n) -= 1.0;
  or_mat_x_vec(n, gamma2_inv_gamma1, emiss_rate, v1);
  or_solve_vec(n, tmp_mat, v1, inv_gamma_b);

  /* Section 4 :198-209 */
  or_solve_vec(n, w.g1, inv_gamma_b, inv_g1_inv_gamma_b);
  or_mat_x_vec(n, w.g2_d, inv_g1_inv_gamma_b, v1);
  for (int i = 0; i < n; ++i) tmp_vec[i] = inv_gamma_b[i] - v1[i];
  or_mat_x_vec(n, w.cprime_lower, tmp_vec, v1);
  for (int i = 0; i < n; ++i) cb_prime[i] = -v1[i];
  for (int k = 0; k < nn; ++k) t1[k] = w.g1_d[k] + w.g2[k];
  or_mat_x_vec(n, t1, cb_prime, v1);
  for (int i = 0; i < n; ++i) source[i] = v1[i] + inv_gamma_b[i];

  /* Section 5 :211-227 */
  for (int j = 0; j < n; ++j) w.exp_lambda_dz[j] = (1.0 - w.exp_lambda_dz[j]) / w.lambda[j];
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      E(w.g1, i, j, n) = E(w.g1, i, j, n) * w.exp_lambda_dz[j];
      E(w.g2, i, j, n) = E(w.g2, i, j, n) * w.exp_lambda_dz[j];
    }
  for (int k = 0; k < nn; ++k) {
    tmp_mat[k] = w.g1[k] + w.g2[k];
    t1[k] = w.cprime_lower[k] + w.cprime_upper[k];
  }
  or_mat_x_mat(n, tmp_mat, t1, int_flux);
  or_mat_x_vec(n, tmp_mat, cb_prime, v1);
  for (int i = 0; i < n; ++i) int_flux_source[i] = 2.0 * (v1[i] + inv_gamma_b[i] * dz);
  return ierr;
}

#endif /* ORACLE_RADTOOL_H */
