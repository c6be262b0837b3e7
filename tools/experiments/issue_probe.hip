// Probe: what ONE SIMD of the MI355X sustains for the instruction classes of the fp64 kernels, as a function of the
// number of resident waves per SIMD - the yardstick behind "fraction of the fp64 peak" in DESIGN.md section 4.
//   issue_probe [reps]
// Every wave runs the same unrolled body (12 independent v_fma_f64 chains, optionally interleaved with one filler
// instruction per FMA) and stamps s_memtime (shader clock) and s_memrealtime (100 MHz) at both ends, so the cycles per
// instruction and the clock the chip really ran at are separated.  Grid = 1024 * W one-wave blocks; the resident
// waves per SIMD are capped at W by the LDS each block asks for.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define FMA12 \
  "v_fma_f64 %0, %14, %15, %0\n v_fma_f64 %1, %14, %15, %1\n v_fma_f64 %2, %14, %15, %2\n v_fma_f64 %3, %14, %15, %3\n" \
  "v_fma_f64 %4, %14, %15, %4\n v_fma_f64 %5, %14, %15, %5\n v_fma_f64 %6, %14, %15, %6\n v_fma_f64 %7, %14, %15, %7\n" \
  "v_fma_f64 %8, %14, %15, %8\n v_fma_f64 %9, %14, %15, %9\n v_fma_f64 %10, %14, %15, %10\n v_fma_f64 %11, %14, %15, %11\n"
#define WITH(F) \
  "v_fma_f64 %0, %14, %15, %0\n" F "v_fma_f64 %1, %14, %15, %1\n" F "v_fma_f64 %2, %14, %15, %2\n" F "v_fma_f64 %3, %14, %15, %3\n" F \
  "v_fma_f64 %4, %14, %15, %4\n" F "v_fma_f64 %5, %14, %15, %5\n" F "v_fma_f64 %6, %14, %15, %6\n" F "v_fma_f64 %7, %14, %15, %7\n" F \
  "v_fma_f64 %8, %14, %15, %8\n" F "v_fma_f64 %9, %14, %15, %9\n" F "v_fma_f64 %10, %14, %15, %10\n" F "v_fma_f64 %11, %14, %15, %11\n" F
#define OPS : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), \
              "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(t32), "+v"(m0) : "v"(a), "v"(b), "v"(u32), "v"(m1)

// MODE 0 fma only; 1 + v_mov_b32; 2 + s_mov_b32; 3 + v_cndmask_b32; 4 + v_accvgpr_write; 5 + v_mul_f64 (independent);
// 6 + v_add_f64; 7 + v_mov_b64; 8 + s_nop 0; 9 + v_add_u32; 10: fmac with a DEPENDENT chain of length 2 (6 chains x 2)
template <int MODE>
__global__ void __launch_bounds__(64) k_issue(double* out, long long* stamps, int reps, double a, double b) {
  extern __shared__ char lds[];
  double c[12];
  for (int j = 0; j < 12; ++j) c[j] = 1.0 + threadIdx.x * 1e-3 + j;
  int t32 = threadIdx.x, u32 = threadIdx.x + 1;
  double m0 = a, m1 = b;
  const long long t0 = __builtin_readcyclecounter();
  const long long w0 = wall_clock64();
  for (int r = 0; r < reps; ++r) {
    if constexpr (MODE == 0) asm volatile(FMA12 FMA12 FMA12 FMA12 OPS);
    if constexpr (MODE == 1) asm volatile(WITH("v_mov_b32 %12, %16\n") WITH("v_mov_b32 %12, %16\n") OPS);
    if constexpr (MODE == 2) asm volatile(WITH("s_mov_b32 s20, 5\n") WITH("s_mov_b32 s20, 5\n") OPS : "s20");
    if constexpr (MODE == 3) asm volatile(WITH("v_cndmask_b32 %12, %16, %12, vcc\n") WITH("v_cndmask_b32 %12, %16, %12, vcc\n") OPS : "vcc");
    if constexpr (MODE == 4) asm volatile(WITH("v_accvgpr_write_b32 a0, %16\n") WITH("v_accvgpr_write_b32 a0, %16\n") OPS : "a0");
    if constexpr (MODE == 5) asm volatile(WITH("v_mul_f64 %13, %14, %15\n") WITH("v_mul_f64 %13, %14, %15\n") OPS);
    if constexpr (MODE == 6) asm volatile(WITH("v_add_f64 %13, %14, %15\n") WITH("v_add_f64 %13, %14, %15\n") OPS);
    if constexpr (MODE == 7) asm volatile(WITH("v_mov_b64 %13, %17\n") WITH("v_mov_b64 %13, %17\n") OPS);
    if constexpr (MODE == 8) asm volatile(WITH("s_nop 0\n") WITH("s_nop 0\n") OPS);
    if constexpr (MODE == 9) asm volatile(WITH("v_add_u32 %12, %16, %12\n") WITH("v_add_u32 %12, %16, %12\n") OPS);
    if constexpr (MODE == 10)
      asm volatile(
          "v_fma_f64 %0, %14, %15, %0\n v_fma_f64 %0, %14, %15, %0\n v_fma_f64 %1, %14, %15, %1\n v_fma_f64 %1, %14, %15, %1\n"
          "v_fma_f64 %2, %14, %15, %2\n v_fma_f64 %2, %14, %15, %2\n v_fma_f64 %3, %14, %15, %3\n v_fma_f64 %3, %14, %15, %3\n"
          "v_fma_f64 %4, %14, %15, %4\n v_fma_f64 %4, %14, %15, %4\n v_fma_f64 %5, %14, %15, %5\n v_fma_f64 %5, %14, %15, %5\n"
          "v_fma_f64 %6, %14, %15, %6\n v_fma_f64 %6, %14, %15, %6\n v_fma_f64 %7, %14, %15, %7\n v_fma_f64 %7, %14, %15, %7\n"
          "v_fma_f64 %8, %14, %15, %8\n v_fma_f64 %8, %14, %15, %8\n v_fma_f64 %9, %14, %15, %9\n v_fma_f64 %9, %14, %15, %9\n"
          "v_fma_f64 %10, %14, %15, %10\n v_fma_f64 %10, %14, %15, %10\n v_fma_f64 %11, %14, %15, %11\n v_fma_f64 %11, %14, %15, %11\n"
          OPS);
  }
  const long long t1 = __builtin_readcyclecounter();
  const long long w1 = wall_clock64();
  double s = 0.0;
  for (int j = 0; j < 12; ++j) s += c[j];
  out[(long)blockIdx.x * 64 + threadIdx.x] = s + t32 + m0;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = w1 - w0; }
  if (reps < 0) lds[threadIdx.x] = 1;
}

template <int MODE>
void run(const char* name, int fma_per_rep, int fill_per_rep, int reps, double* out, long long* stamps) {
  printf("%-28s", name);
  for (int W : {1, 2, 4, 8}) {
    const int nblk = 1024 * W;
    const size_t lds = (size_t)(160 * 1024) / (4 * W) - 512;       // at most 4 W blocks per CU
    hipFuncSetAttribute((const void*)k_issue<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e30f;
    std::vector<long long> st(2 * nblk);
    for (int it = 0; it < 3; ++it) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_issue<MODE>, dim3(nblk), dim3(64), lds, 0, out, stamps, reps, 1.0000001, 1e-9);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      best = std::min(best, ms);
      hipEventDestroy(e0); hipEventDestroy(e1);
    }
    hipMemcpy(st.data(), stamps, sizeof(long long) * 2 * nblk, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < nblk; ++i) { cyc += st[2 * i]; wall += st[2 * i + 1]; }
    cyc /= nblk; wall /= nblk;
    const double mhz = cyc / (wall / 100.0);          // wall_clock64: 100 MHz
    const double inst = (double)reps * (fma_per_rep + fill_per_rep);
    // cycles of the SIMD per instruction of ANY wave on it: the wave's cycles / its instructions / W
    printf(" | W=%d %5.2f cyc/inst/SIMD  %4.0f MHz  %5.1f TF", W, cyc / inst / W, mhz,
           (double)nblk * reps * fma_per_rep * 128.0 / (best * 1e-3) / 1e12);
  }
  printf("\n");
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 2000;
  double* out; long long* stamps;
  hipMalloc(&out, 8192L * 64 * 8); hipMalloc(&stamps, 8192L * 16);
  printf("cyc/inst/SIMD = shader-clock cycles of a wave / (its instructions x resident waves per SIMD); TF = fp64 FMA flops of the launch / event time\n");
  run<0>("v_fma_f64 only", 48, 0, reps, out, stamps);
  run<10>("v_fma_f64 dependent pairs", 24, 0, reps, out, stamps);
  run<1>("fma + v_mov_b32", 24, 24, reps, out, stamps);
  run<7>("fma + v_mov_b64", 24, 24, reps, out, stamps);
  run<3>("fma + v_cndmask_b32", 24, 24, reps, out, stamps);
  run<4>("fma + v_accvgpr_write", 24, 24, reps, out, stamps);
  run<9>("fma + v_add_u32", 24, 24, reps, out, stamps);
  run<5>("fma + v_mul_f64", 24, 24, reps, out, stamps);
  run<6>("fma + v_add_f64", 24, 24, reps, out, stamps);
  run<2>("fma + s_mov_b32", 24, 24, reps, out, stamps);
  run<8>("fma + s_nop 0", 24, 24, reps, out, stamps);
  return 0;
}
