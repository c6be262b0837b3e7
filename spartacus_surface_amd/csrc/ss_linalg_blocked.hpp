// ss_linalg_blocked.hpp - the same per-lane dense linear algebra as ss_linalg.hpp
// for orders above SS_UNROLL_MAX (4 or 8 streams per hemisphere: n = 8 ... 24),
// where a matrix no longer fits in the registers of its lane and lives in
// scratch memory.
//
// The rolled loops of the small-order code would fetch both operands of every
// multiply-add from scratch (2 loads per FMA; with ~1000 waves in flight the
// scratch working set exceeds every cache, so that is HBM traffic).  Here the
// matrices are processed in 4 x 4 tiles held in registers: a tile product is 64
// FMAs for 32 loads, the trailing update of the block LU 64 FMAs for 32 loads
// and 16 stores.  Results differ from the rolled loops only in the order of the
// partial sums.  Unpivoted, like everything in the reference's radtool_matrix;
// the factor layout is that of ss_linalg.hpp (unit L below the diagonal, U above,
// RECIPROCALS of U's diagonal on the diagonal), so the two are interchangeable.
#pragma once

#include "ss_common.hpp"

namespace ss {
namespace blk {

constexpr int kT = 4;   // tile edge

// Tile (i0.., j0..) of the column-major R x C matrix A; elements outside the
// matrix read as zero (as one on the diagonal when `unit_pad`: the padding of a
// matrix that is being factorised must stay an identity block).
template <int R, int C>
SS_HD void tile_load(const double* A, int i0, int j0, double* t, bool unit_pad = false) {
  SS_UNROLL_FULL
  for (int j = 0; j < kT; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < kT; ++i) {
      const int r = i0 + i, c = j0 + j;
      const bool inside = ((R % kT == 0) || r < R) && ((C % kT == 0) || c < C);
      t[i + kT * j] = inside ? A[r + (long)c * R] : ((unit_pad && r == c) ? 1.0 : 0.0);
    }
  }
}

template <int R, int C>
SS_HD void tile_store(double* A, int i0, int j0, const double* t) {
  SS_UNROLL_FULL
  for (int j = 0; j < kT; ++j) {
    SS_UNROLL_FULL
    for (int i = 0; i < kT; ++i) {
      const int r = i0 + i, c = j0 + j;
      if (((R % kT == 0) || r < R) && ((C % kT == 0) || c < C)) A[r + (long)c * R] = t[i + kT * j];
    }
  }
}

// acc (+/-)= a * b
template <bool NEG>
SS_HD void tile_mma(double* acc, const double* a, const double* b) {
  SS_UNROLL_FULL
  for (int j = 0; j < kT; ++j) {
    SS_UNROLL_FULL
    for (int k = 0; k < kT; ++k) {
      const double bkj = NEG ? -b[k + kT * j] : b[k + kT * j];
      SS_UNROLL_FULL
      for (int i = 0; i < kT; ++i) acc[i + kT * j] += a[i + kT * k] * bkj;
    }
  }
}

constexpr int tiles(int n) { return (n + kT - 1) / kT; }

// C (+)= A diag(e)? B     (A: M x K, B: K x N, C: M x N)
template <int M, int K, int N, bool ACC, bool SCALED>
SS_HD void mat_mul(const double* A, const double* e, const double* B, double* C) {
  SS_PRAGMA(unroll(1))
  for (int jb = 0; jb < tiles(N); ++jb) {
    SS_PRAGMA(unroll(1))
    for (int ib = 0; ib < tiles(M); ++ib) {
      double acc[kT * kT];
      if (ACC) {
        tile_load<M, N>(C, ib * kT, jb * kT, acc);
      } else {
        SS_UNROLL_FULL
        for (int q = 0; q < kT * kT; ++q) acc[q] = 0.0;
      }
      SS_PRAGMA(unroll(1))
      for (int kb = 0; kb < tiles(K); ++kb) {
        double a[kT * kT], b[kT * kT];
        tile_load<M, K>(A, ib * kT, kb * kT, a);
        tile_load<K, N>(B, kb * kT, jb * kT, b);
        if (SCALED) {
          SS_UNROLL_FULL
          for (int k = 0; k < kT; ++k) {
            const int kk = kb * kT + k;
            const double ek = ((K % kT == 0) || kk < K) ? e[kk] : 0.0;
            SS_UNROLL_FULL
            for (int j = 0; j < kT; ++j) b[k + kT * j] *= ek;
          }
        }
        tile_mma<false>(acc, a, b);
      }
      tile_store<M, N>(C, ib * kT, jb * kT, acc);
    }
  }
}

// In-register unpivoted LU of one tile (reciprocal diagonal stored)
SS_HD void tile_lu(double* d) {
  SS_UNROLL_FULL
  for (int k = 0; k < kT; ++k) {
    const double r = 1.0 / d[k + kT * k];
    d[k + kT * k] = r;
    SS_UNROLL_FULL
    for (int i = k + 1; i < kT; ++i) d[i + kT * k] *= r;
    SS_UNROLL_FULL
    for (int j = k + 1; j < kT; ++j) {
      SS_UNROLL_FULL
      for (int i = k + 1; i < kT; ++i) d[i + kT * j] -= d[i + kT * k] * d[k + kT * j];
    }
  }
}

// x <- L^-1 x, L = unit lower triangle of the factored tile d (x: 4 x 4, columns independent)
SS_HD void tile_solve_lower(const double* d, double* x) {
  SS_UNROLL_FULL
  for (int j = 0; j < kT; ++j) {
    SS_UNROLL_FULL
    for (int i = 1; i < kT; ++i) {
      SS_UNROLL_FULL
      for (int k = 0; k < i; ++k) x[i + kT * j] -= d[i + kT * k] * x[k + kT * j];
    }
  }
}

// x <- U^-1 x, U = upper triangle of the factored tile d with reciprocal diagonal
SS_HD void tile_solve_upper(const double* d, double* x) {
  SS_UNROLL_FULL
  for (int j = 0; j < kT; ++j) {
    SS_UNROLL_FULL
    for (int ii = 0; ii < kT; ++ii) {
      const int i = kT - 1 - ii;
      SS_UNROLL_FULL
      for (int k = i + 1; k < kT; ++k) x[i + kT * j] -= d[i + kT * k] * x[k + kT * j];
      x[i + kT * j] *= d[i + kT * i];
    }
  }
}

// x <- x U^-1 (rows independent)
SS_HD void tile_solve_upper_right(const double* d, double* x) {
  SS_UNROLL_FULL
  for (int j = 0; j < kT; ++j) {
    SS_UNROLL_FULL
    for (int k = 0; k < j; ++k) {
      SS_UNROLL_FULL
      for (int i = 0; i < kT; ++i) x[i + kT * j] -= x[i + kT * k] * d[k + kT * j];
    }
    SS_UNROLL_FULL
    for (int i = 0; i < kT; ++i) x[i + kT * j] *= d[j + kT * j];
  }
}

// Right-looking block LU, in place
template <int N>
SS_HD void lu_factor(double* A) {
  constexpr int nb = tiles(N);
  SS_PRAGMA(unroll(1))
  for (int kb = 0; kb < nb; ++kb) {
    double d[kT * kT];
    tile_load<N, N>(A, kb * kT, kb * kT, d, true);
    tile_lu(d);
    tile_store<N, N>(A, kb * kT, kb * kT, d);
    SS_PRAGMA(unroll(1))
    for (int jb = kb + 1; jb < nb; ++jb) {       // U_kj = L_kk^-1 A_kj
      double x[kT * kT];
      tile_load<N, N>(A, kb * kT, jb * kT, x);
      tile_solve_lower(d, x);
      tile_store<N, N>(A, kb * kT, jb * kT, x);
    }
    SS_PRAGMA(unroll(1))
    for (int ib = kb + 1; ib < nb; ++ib) {       // L_ik = A_ik U_kk^-1, then the trailing row of tiles
      double l[kT * kT];
      tile_load<N, N>(A, ib * kT, kb * kT, l);
      tile_solve_upper_right(d, l);
      tile_store<N, N>(A, ib * kT, kb * kT, l);
      SS_PRAGMA(unroll(1))
      for (int jb = kb + 1; jb < nb; ++jb) {
        double u[kT * kT], t[kT * kT];
        tile_load<N, N>(A, kb * kT, jb * kT, u);
        tile_load<N, N>(A, ib * kT, jb * kT, t, true);
        tile_mma<true>(t, l, u);
        tile_store<N, N>(A, ib * kT, jb * kT, t);
      }
    }
  }
}

// Solve A X = B in place (B: N x M) given the factor
template <int N, int M>
SS_HD void lu_solve_mat(const double* LU, double* B) {
  constexpr int nb = tiles(N);
  SS_PRAGMA(unroll(1))
  for (int jb = 0; jb < tiles(M); ++jb) {
    SS_PRAGMA(unroll(1))
    for (int kb = 0; kb < nb; ++kb) {            // forward: L y = b
      double x[kT * kT], d[kT * kT];
      tile_load<N, M>(B, kb * kT, jb * kT, x);
      SS_PRAGMA(unroll(1))
      for (int pb = 0; pb < kb; ++pb) {
        double l[kT * kT], y[kT * kT];
        tile_load<N, N>(LU, kb * kT, pb * kT, l);
        tile_load<N, M>(B, pb * kT, jb * kT, y);
        tile_mma<true>(x, l, y);
      }
      tile_load<N, N>(LU, kb * kT, kb * kT, d, true);
      tile_solve_lower(d, x);
      tile_store<N, M>(B, kb * kT, jb * kT, x);
    }
    SS_PRAGMA(unroll(1))
    for (int kk = 0; kk < nb; ++kk) {            // backward: U x = y
      const int kb = nb - 1 - kk;
      double x[kT * kT], d[kT * kT];
      tile_load<N, M>(B, kb * kT, jb * kT, x);
      SS_PRAGMA(unroll(1))
      for (int pb = kb + 1; pb < nb; ++pb) {
        double u[kT * kT], y[kT * kT];
        tile_load<N, N>(LU, kb * kT, pb * kT, u);
        tile_load<N, M>(B, pb * kT, jb * kT, y);
        tile_mma<true>(x, u, y);
      }
      tile_load<N, N>(LU, kb * kT, kb * kT, d, true);
      tile_solve_upper(d, x);
      tile_store<N, M>(B, kb * kT, jb * kT, x);
    }
  }
}

}  // namespace blk
}  // namespace ss
