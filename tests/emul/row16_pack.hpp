// row16_pack.hpp - TEST-ONLY: the "lane" type of the row-of-16 kernels for host builds.
// spartacus_surface_amd/csrc/ss_row16.hpp is device code: a value of type rv is one double per lane
// and its broadcast-multiply-add is v_fmac_f64_dpp.  The host emulation (tests/emul, g++) runs the same
// kernel source on a PACK of 16 doubles, one "lane" each; this header supplies that type and the few
// primitives the device header defines with DPP.  It is included by ss_row16.hpp through
// -DSS_ROW16_PACK_HEADER when the compiler is not hipcc, and by nothing in the product.
#pragma once

#include <cmath>

namespace ss {
namespace r16 {

// ------------------------------------------------------------------ host: a pack of 16 lanes
struct rv {
  double v[kRow];
  rv() = default;
  rv(double x) { for (int l = 0; l < kRow; ++l) v[l] = x; }
};
struct rm { bool v[kRow]; };
#define SS_R16_OP(op)                                                                                   \
  inline rv operator op(const rv& a, const rv& b) { rv r; for (int l = 0; l < kRow; ++l) r.v[l] = a.v[l] op b.v[l]; return r; } \
  inline rv operator op(const rv& a, double b) { rv r; for (int l = 0; l < kRow; ++l) r.v[l] = a.v[l] op b; return r; }       \
  inline rv operator op(double a, const rv& b) { rv r; for (int l = 0; l < kRow; ++l) r.v[l] = a op b.v[l]; return r; }
SS_R16_OP(+) SS_R16_OP(-) SS_R16_OP(*) SS_R16_OP(/)
#undef SS_R16_OP
inline rv operator-(const rv& a) { rv r; for (int l = 0; l < kRow; ++l) r.v[l] = -a.v[l]; return r; }
inline rv& operator*=(rv& a, const rv& b) { a = a * b; return a; }
inline rv& operator*=(rv& a, double b) { a = a * b; return a; }
inline rv& operator+=(rv& a, const rv& b) { a = a + b; return a; }
inline rv& operator-=(rv& a, const rv& b) { a = a - b; return a; }
inline rm lane_lt(int k) { rm m; for (int l = 0; l < kRow; ++l) m.v[l] = l < k; return m; }
inline rm lane_gt(int k) { rm m; for (int l = 0; l < kRow; ++l) m.v[l] = l > k; return m; }
inline rm lane_eq(int k) { rm m; for (int l = 0; l < kRow; ++l) m.v[l] = l == k; return m; }
inline rv select(const rm& m, const rv& a, const rv& b) { rv r; for (int l = 0; l < kRow; ++l) r.v[l] = m.v[l] ? a.v[l] : b.v[l]; return r; }
inline rv r_sqrt(const rv& x) { rv r; for (int l = 0; l < kRow; ++l) r.v[l] = std::sqrt(x.v[l]); return r; }
inline rv r_exp(const rv& x) { rv r; for (int l = 0; l < kRow; ++l) r.v[l] = std::exp(x.v[l]); return r; }
inline rv r_max(const rv& a, double b) { rv r; for (int l = 0; l < kRow; ++l) r.v[l] = a.v[l] > b ? a.v[l] : b; return r; }
inline rv r_max(const rv& a, const rv& b) { rv r; for (int l = 0; l < kRow; ++l) r.v[l] = a.v[l] > b.v[l] ? a.v[l] : b.v[l]; return r; }
inline rv r_abs(const rv& x) { rv r; for (int l = 0; l < kRow; ++l) r.v[l] = std::fabs(x.v[l]); return r; }
template <class F>
inline rv gen(F&& f) { rv r; for (int l = 0; l < kRow; ++l) r.v[l] = f(l); return r; }
template <class F>
inline rm gen_mask(F&& f) { rm m; for (int l = 0; l < kRow; ++l) m.v[l] = f(l); return m; }
inline bool lane0() { return true; }
template <class F>
inline void each(const rv& x, F&& f) { for (int l = 0; l < kRow; ++l) f(l, x.v[l]); }
inline double uniform(const rv& x) { return x.v[0]; }
template <int K>
inline double bc(const rv& x) { return x.v[K]; }
template <int K, int J0, int M>
inline void row_fma_from(rv* c, const rv* b, const rv& a) {
  for (int j = J0; j < M; ++j) {
    const double s = b[j].v[K];
    for (int l = 0; l < kRow; ++l) c[j].v[l] = std::fma(s, a.v[l], c[j].v[l]);
  }
}

}  // namespace r16
}  // namespace ss
