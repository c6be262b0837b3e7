// Probe: v_fmac_f64_dpp row_newbcast on gfx950 - correctness of a 12x12 product with matrix rows
// on the lanes of a 16-lane DPP row, and its issue rate against plain v_fma_f64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <utility>

template <int K>
__device__ __forceinline__ void row_fma12(double (&c)[12], const double (&b)[12], double a) {
  asm volatile("s_nop 1\n"
      "v_fmac_f64_dpp %0, %12, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %1, %13, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %2, %14, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %3, %15, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %4, %16, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %5, %17, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %6, %18, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %7, %19, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %8, %20, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %9, %21, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %10, %22, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %11, %23, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n"
      : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]),
        "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11])
      : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]),
        "v"(b[9]), "v"(b[10]), "v"(b[11]), "v"(a), "n"(K));
}
template <int... K>
__device__ __forceinline__ void prod12(double (&c)[12], const double (&a)[12], const double (&b)[12],
                                       std::integer_sequence<int, K...>) {
  (row_fma12<K>(c, b, a[K]), ...);
}

// rows of A, B (row-major 16 x 12 per problem, rows 12..15 zero) on lanes; C = A B; REP times C = C B + A
__global__ void __launch_bounds__(64) k_dpp(const double* A, const double* B, double* C, int rep) {
  const long t = (long)blockIdx.x * 64 + threadIdx.x;
  double a[12], b[12], c[12];
  for (int j = 0; j < 12; ++j) { a[j] = A[t * 12 + j]; b[j] = B[t * 12 + j]; c[j] = 0.0; }
  prod12(c, a, b, std::make_integer_sequence<int, 12>{});
  for (int r = 1; r < rep; ++r) {
    double d[12];
    for (int j = 0; j < 12; ++j) d[j] = a[j];
    prod12(d, c, b, std::make_integer_sequence<int, 12>{});
    for (int j = 0; j < 12; ++j) c[j] = d[j];
  }
  for (int j = 0; j < 12; ++j) C[t * 12 + j] = c[j];
}
// the same arithmetic volume with plain FMAs (lane-local: c_j += a_k * b_j)
__global__ void __launch_bounds__(64) k_plain(const double* A, const double* B, double* C, int rep) {
  const long t = (long)blockIdx.x * 64 + threadIdx.x;
  double a[12], b[12], c[12];
  for (int j = 0; j < 12; ++j) { a[j] = A[t * 12 + j]; b[j] = B[t * 12 + j]; c[j] = 0.0; }
  for (int r = 0; r < rep; ++r) {
    double d[12];
    for (int j = 0; j < 12; ++j) d[j] = (r == 0) ? 0.0 : a[j];
#pragma unroll
    for (int k = 0; k < 12; ++k)
#pragma unroll
      for (int j = 0; j < 12; ++j) d[j] = fma((r == 0) ? a[k] : c[k], b[j], d[j]);
    for (int j = 0; j < 12; ++j) c[j] = d[j];
  }
  for (int j = 0; j < 12; ++j) C[t * 12 + j] = c[j];
}

int main(int argc, char** argv) {
  const int nblk = argc > 1 ? atoi(argv[1]) : 8192, rep = argc > 2 ? atoi(argv[2]) : 64;
  const long nl = (long)nblk * 64;
  std::vector<double> A(nl * 12), B(nl * 12), C(nl * 12);
  srand(1);
  for (long t = 0; t < nl; ++t)
    for (int j = 0; j < 12; ++j) {
      const bool pad = (t % 16) >= 12;
      A[t * 12 + j] = pad ? 0.0 : (rand() / (double)RAND_MAX - 0.5);
      B[t * 12 + j] = pad ? 0.0 : (rand() / (double)RAND_MAX - 0.5) * 0.5;
    }
  double *dA, *dB, *dC;
  hipMalloc(&dA, nl * 96); hipMalloc(&dB, nl * 96); hipMalloc(&dC, nl * 96);
  hipMemcpy(dA, A.data(), nl * 96, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), nl * 96, hipMemcpyHostToDevice);
  // correctness: one product
  hipLaunchKernelGGL(k_dpp, dim3(nblk), dim3(64), 0, 0, dA, dB, dC, 1);
  hipMemcpy(C.data(), dC, nl * 96, hipMemcpyDeviceToHost);
  double maxerr = 0.0;
  for (long p = 0; p < nl / 16; ++p)
    for (int i = 0; i < 12; ++i)
      for (int j = 0; j < 12; ++j) {
        double s = 0.0;
        for (int k = 0; k < 12; ++k) s = fma(B[(p * 16 + k) * 12 + j], A[(p * 16 + i) * 12 + k], s);
        maxerr = fmax(maxerr, fabs(s - C[(p * 16 + i) * 12 + j]));
      }
  printf("dpp product: max abs error vs host fma chain %.3e (%ld problems)\n", maxerr, nl / 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    for (int it = 0; it < 2; ++it) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k_dpp, dim3(nblk), dim3(64), 0, 0, dA, dB, dC, rep);
      else hipLaunchKernelGGL(k_plain, dim3(nblk), dim3(64), 0, 0, dA, dB, dC, rep);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fma_wave = (double)nblk * rep * 144.0;       // wave-level FMA instructions
      printf("%s: %.3f ms, %.2f cycles per wave-FMA per SIMD at 2.4 GHz (1024 SIMDs), %.1f TFLOP/s lane-raw\n",
             which == 0 ? "dpp  " : "plain", ms, ms * 1e-3 * 2.4e9 * 1024 / fma_wave, fma_wave * 128 / (ms * 1e-3) / 1e12);
    }
  }
  return maxerr < 1e-12 ? 0 : 1;
}
