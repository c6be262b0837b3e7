// ss_row16.hpp - dense algebra of ONE problem on the 16 lanes of a DPP row (orders 7..16).
//
// Layout: row i of every matrix lives on lane i of a 16-lane row of the wavefront (4 problems
// per wave), as an array of registers indexed by the column; a vector is one double per lane.
// gfx950 executes   v_fmac_f64_dpp  d, s0, s1  row_newbcast:K   ( d += [s0 of lane K of the row] * s1 )
// at the rate of a plain v_fma_f64 (measured: tools/experiments/dpp_probe.hip, 6.5 against 5.9 cycles
// per wave instruction, bit-identical results), so "broadcast row k of B and update the own row"
// costs nothing for the broadcast:
//   product     C = A B      for k:  c_i[j] += bcast_k(b[j]) * a_i[k]                  N*M  fmac per lane
//   LU          (unpivoted)  for k:  l_i = a_i[k] / a_k[k];  a_i[j] -= bcast_k(a[j]) l_i  (i > k)
//   L U X = B                for k:  x_i[j] -= bcast_k(x[j]) * l_i[k]   (i > k), then upward
//   y = A x                  for k:  y_i += bcast_k(x) * a_i[k]
//   sum over the lanes       for k:  s += bcast_k(x) * 1
// A 12 x 12 matrix is 24 VGPRs per lane: the layer matrices of a unit and the adding method of a
// column live in registers at two waves per SIMD, where the one-lane form of these orders keeps
// 17 KB per lane in scratch memory.
//
// The DPP instructions go in as inline asm (the compiler folds no 64-bit DPP move into an FMA).
// The hazard "VALU writes a VGPR, DPP reads it within 2 wait states" is not tracked through
// inline asm, so every asm block opens with s_nop 1; the blocks are volatile: they keep their
// program order and are never sunk into divergent control flow (a DPP read of a lane that EXEC
// has switched off does not happen: all 16 lanes of a row always take the same branches).
// (A "memory" clobber on them, which pins loads and stores between the blocks, was needed while the
// record addresses cost dozens of registers - ss_layer16.hpp, "Addresses" - and is off since:
// -DSS_R16_MEMCLOBBER; without it the longwave downward passes overlap their loads with the
// arithmetic, 0.62 -> 0.44 ms.)
//
// Host builds (tests/emul, TEST-ONLY) run the same code on a pack of 16 doubles, which they bring
// along themselves (tests/emul/row16_pack.hpp through -DSS_ROW16_PACK_HEADER).
#pragma once

#include <type_traits>
#include <utility>

#include "ss_common.hpp"

namespace ss {
namespace r16 {

constexpr int kRow = 16;

template <int B, int E, class F>
SS_HD void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
// E-1, E-2, ..., B
template <int B, int E, class F>
SS_HD void static_for_down(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, E - 1>{});
    static_for_down<B, E - 1>(f);
  }
}

#if defined(__HIP_DEVICE_COMPILE__)
// ------------------------------------------------------------------ device: one value per lane
using rv = double;
using rm = bool;
SS_HD int lane() { return (int)threadIdx.x & (kRow - 1); }
SS_HD rm lane_lt(int k) { return lane() < k; }
SS_HD rm lane_gt(int k) { return lane() > k; }
SS_HD rm lane_eq(int k) { return lane() == k; }
SS_HD rv select(rm m, rv a, rv b) { return m ? a : b; }
SS_HD rv r_sqrt(rv x) { return sqrt(x); }
SS_HD rv r_exp(rv x) { return exp(x); }
SS_HD rv r_max(rv a, double b) { return a > b ? a : b; }
SS_HD rv r_abs(rv x) { return fabs(x); }
template <class F>
SS_HD rv gen(F&& f) { return f(lane()); }
template <class F>
SS_HD rm gen_mask(F&& f) { return f(lane()); }
SS_HD bool lane0() { return lane() == 0; }
template <class F>
SS_HD void each(const rv& x, F&& f) { f(lane(), x); }
SS_HD double uniform(rv x) { return x; }

#define SS_R16_DPP " row_mask:0xf bank_mask:0xf\n"
#if defined(SS_R16_MEMCLOBBER)      // experiment: pin loads and stores between the asm blocks
#define SS_R16_CLOBBER : "memory"
#else
#define SS_R16_CLOBBER
#endif
// value of lane K of the row, on every lane
template <int K>
SS_HD double bc(rv x) {
  double r;
  asm volatile("s_nop 1\nv_mov_b64_dpp %0, %1 row_newbcast:%2" SS_R16_DPP : "=v"(r) : "v"(x), "n"(K));
  return r;
}
// d += bcast_K(b) * a
#define SS_R16_L(o, i) "v_fmac_f64_dpp %" #o ", %" #i ", %[a] row_newbcast:%[k]" SS_R16_DPP
template <int K>
SS_HD void fma1(rv& c0, rv b0, rv a) {
  asm volatile("s_nop 1\n" SS_R16_L(0, 1) : "+v"(c0) : "v"(b0), [a] "v"(a), [k] "n"(K) SS_R16_CLOBBER);
}
template <int K>
SS_HD void fma2(rv& c0, rv& c1, rv b0, rv b1, rv a) {
  asm volatile("s_nop 1\n" SS_R16_L(0, 2) SS_R16_L(1, 3) : "+v"(c0), "+v"(c1) : "v"(b0), "v"(b1), [a] "v"(a), [k] "n"(K) SS_R16_CLOBBER);
}
template <int K>
SS_HD void fma3(rv* c, const rv* b, rv a) {
  asm volatile("s_nop 1\n" SS_R16_L(0, 3) SS_R16_L(1, 4) SS_R16_L(2, 5)
               : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]) : "v"(b[0]), "v"(b[1]), "v"(b[2]), [a] "v"(a), [k] "n"(K) SS_R16_CLOBBER);
}
template <int K>
SS_HD void fma4(rv* c, const rv* b, rv a) {
  asm volatile("s_nop 1\n" SS_R16_L(0, 4) SS_R16_L(1, 5) SS_R16_L(2, 6) SS_R16_L(3, 7)
               : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
               : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), [a] "v"(a), [k] "n"(K) SS_R16_CLOBBER);
}
template <int K>
SS_HD void fma6(rv* c, const rv* b, rv a) {
  asm volatile("s_nop 1\n" SS_R16_L(0, 6) SS_R16_L(1, 7) SS_R16_L(2, 8) SS_R16_L(3, 9) SS_R16_L(4, 10) SS_R16_L(5, 11)
               : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5])
               : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), [a] "v"(a), [k] "n"(K) SS_R16_CLOBBER);
}
template <int K>
SS_HD void fma12(rv* c, const rv* b, rv a) {
  asm volatile("s_nop 1\n" SS_R16_L(0, 12) SS_R16_L(1, 13) SS_R16_L(2, 14) SS_R16_L(3, 15) SS_R16_L(4, 16)
               SS_R16_L(5, 17) SS_R16_L(6, 18) SS_R16_L(7, 19) SS_R16_L(8, 20) SS_R16_L(9, 21) SS_R16_L(10, 22)
               SS_R16_L(11, 23)
               : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]),
                 "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11])
               : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]),
                 "v"(b[9]), "v"(b[10]), "v"(b[11]), [a] "v"(a), [k] "n"(K) SS_R16_CLOBBER);
}
#undef SS_R16_L
// c[j] += bcast_K(b[j]) * a for j = J0 .. M-1, in as few asm blocks as the operand limit allows
template <int K, int J0, int M>
SS_HD void row_fma_from(rv* c, const rv* b, rv a) {
  constexpr int W = M - J0;
  if constexpr (W >= 12) {
    fma12<K>(c + J0, b + J0, a);
    row_fma_from<K, J0 + 12, M>(c, b, a);
  } else if constexpr (W >= 6) {
    fma6<K>(c + J0, b + J0, a);
    row_fma_from<K, J0 + 6, M>(c, b, a);
  } else if constexpr (W >= 4) {
    fma4<K>(c + J0, b + J0, a);
    row_fma_from<K, J0 + 4, M>(c, b, a);
  } else if constexpr (W == 3) {
    fma3<K>(c + J0, b + J0, a);
  } else if constexpr (W >= 2) {
    fma2<K>(c[J0], c[J0 + 1], b[J0], b[J0 + 1], a);
    row_fma_from<K, J0 + 2, M>(c, b, a);
  } else if constexpr (W == 1) {
    fma1<K>(c[J0], b[J0], a);
  }
}

#elif defined(__HIPCC__)
// ------------------------------------------------------------------ host pass of hipcc: the same
// one-value-per-lane types so that the __host__ __device__ kernel bodies parse; never executed (the
// product launches them as kernels only, and there is no CPU fallback)
using rv = double;
using rm = bool;
SS_HD int lane() { return 0; }
SS_HD rm lane_lt(int k) { return 0 < k; }
SS_HD rm lane_gt(int) { return false; }
SS_HD rm lane_eq(int k) { return 0 == k; }
SS_HD rv select(rm m, rv a, rv b) { return m ? a : b; }
SS_HD rv r_sqrt(rv x) { return sqrt(x); }
SS_HD rv r_exp(rv x) { return exp(x); }
SS_HD rv r_max(rv a, double b) { return a > b ? a : b; }
SS_HD rv r_abs(rv x) { return fabs(x); }
template <class F>
SS_HD rv gen(F&& f) { return f(0); }
template <class F>
SS_HD rm gen_mask(F&& f) { return f(0); }
SS_HD bool lane0() { return true; }
template <class F>
SS_HD void each(const rv& x, F&& f) { f(0, x); }
SS_HD double uniform(rv x) { return x; }
template <int K>
SS_HD double bc(rv x) { return x; }
template <int K, int J0, int M>
SS_HD void row_fma_from(rv* c, const rv* b, rv a) {
  for (int j = J0; j < M; ++j) c[j] += b[j] * a;
}
#elif defined(SS_ROW16_PACK_HEADER)
}  // namespace r16
}  // namespace ss
#include SS_ROW16_PACK_HEADER      // TEST-ONLY host emulation (tests/emul/row16_pack.hpp): rv = a pack of 16 doubles
namespace ss {
namespace r16 {
#else
#error "ss_row16.hpp is device code (hipcc); host builds of the test-suite supply the lane pack with -DSS_ROW16_PACK_HEADER"
#endif

template <int K, int M>
SS_HD void row_fma(rv* c, const rv* b, const rv& a) { row_fma_from<K, 0, M>(c, b, a); }
// d += bcast_K(b) * a
template <int K>
SS_HD void fma_bc(rv& d, const rv& b, const rv& a) { row_fma_from<K, 0, 1>(&d, &b, a); }

// the own element of a vector every lane holds whole: x[lane]
template <int N>
SS_HD rv own(const double* x) {
  rv r = 0.0;
  static_for<0, N>([&](auto kc) { constexpr int K = decltype(kc)::value; r = select(lane_eq(K), rv(x[K]), r); });
  return r;
}
// a distributed vector on every lane
template <int N>
SS_HD void all_of(const rv& x, double* out) {
  static_for<0, N>([&](auto kc) { constexpr int K = decltype(kc)::value; out[K] = bc<K>(x); });
}
// sum of the elements of a distributed vector (lanes 0 .. N-1), on every lane
template <int N>
SS_HD double all_sum(const rv& x) {
  rv s0 = 0.0, s1 = 0.0;
  const rv one = 1.0;
  static_for<0, N>([&](auto kc) {
    constexpr int K = decltype(kc)::value;
    if (K & 1) fma_bc<K>(s1, x, one);
    else fma_bc<K>(s0, x, one);
  });
  return uniform(s0 + s1);
}

// C (+)= A [diag(e)] B:  A with NK columns (own row), B NK x M (rows 0 .. NK-1 on the lanes), e on every lane
template <int NK, int M, bool ACC, bool SCALE>
SS_HD void mul(const rv* A, const double* e, const rv* B, rv* C) {
  if (!ACC) {
    SS_UNROLL_FULL
    for (int j = 0; j < M; ++j) C[j] = 0.0;
  }
  static_for<0, NK>([&](auto kc) {
    constexpr int K = decltype(kc)::value;
    rv a = A[K];
    if (SCALE) a = a * e[K];
    row_fma<K, M>(C, B, a);
  });
}
// y (+)= A x, x distributed (two chains of dependent multiply-adds)
template <int NK, bool ACC>
SS_HD void mat_vec(const rv* A, const rv& x, rv& y) {
  rv y0 = ACC ? y : rv(0.0), y1 = 0.0;
  static_for<0, NK>([&](auto kc) {
    constexpr int K = decltype(kc)::value;
    if (K & 1) fma_bc<K>(y1, x, A[K]);
    else fma_bc<K>(y0, x, A[K]);
  });
  y = y0 + y1;
}

// unpivoted LU in place: below the diagonal the unit lower factor, above it U with column k scaled
// by 1 / u_kk (the form the upward substitution wants), on the diagonal the pivots; rdd = reciprocal
// of the own pivot
template <int N>
struct LU {
  rv a[N];
  rv rdd;
};
template <int N>
SS_HD void lu_factor(LU<N>& f) {
  f.rdd = 0.0;
  static_for<0, N>([&](auto kc) {
    constexpr int K = decltype(kc)::value;
    const double rp = 1.0 / bc<K>(f.a[K]);      // (fast_rcp here: no measurable gain, kept IEEE)
    f.rdd = select(lane_eq(K), rv(rp), f.rdd);
    const rv s = f.a[K] * rp;                       // rows above: u_ik / u_kk (final), rows below: l_ik
    const rv l = select(lane_gt(K), s, rv(0.0));
    f.a[K] = select(lane_eq(K), f.a[K], s);
    row_fma_from<K, K + 1, N>(f.a, f.a, -l);
  });
}
// X <- (L U)^-1 X, X with M columns.  Upward: row k is used unscaled, x_k = (row k) / u_kk, so the
// update of row i < k is  -(u_ik / u_kk) * (row k), and every row is scaled by its own pivot at the end.
template <int N, int M>
SS_HD void lu_solve(const LU<N>& f, rv* X) {
  static_for<0, N - 1>([&](auto kc) {
    constexpr int K = decltype(kc)::value;
    row_fma<K, M>(X, X, select(lane_gt(K), -f.a[K], rv(0.0)));
  });
  static_for_down<1, N>([&](auto kc) {
    constexpr int K = decltype(kc)::value;
    row_fma<K, M>(X, X, select(lane_lt(K), -f.a[K], rv(0.0)));
  });
  SS_UNROLL_FULL
  for (int j = 0; j < M; ++j) X[j] = X[j] * f.rdd;
}

}  // namespace r16
}  // namespace ss
