// ss_capi.hip - the C-ABI of include/spartacus_surface_hip.h on the HIP backend.
//
// Everything here runs on the GPU or fails: there is no CPU fallback.  If no
// HIP device is usable every entry point returns SS_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/spartacus_surface_hip.h"
#include "ss_kernels.hpp"
#include "ss_plan_impl.hpp"

namespace {

thread_local std::string g_last_error;
std::mutex g_device_mutex;   // serialises host-entry calls per process (concurrent OpenMP callers)

bool hip_ok(hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}

#define HIP_CHECK_THROW(call)                                                       \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                         \
      g_last_error = std::string(#call) + ": " + hipGetErrorString(e_);             \
      throw e_;                                                                     \
    }                                                                               \
  } while (0)

hipStream_t g_stream = nullptr;

hipStream_t stream() {
  if (!g_stream) HIP_CHECK_THROW(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
  return g_stream;
}

// Worker streams of the solve ("pipes" of ss_plan_impl.hpp): pipe 0 is the
// library's main stream, further pipes fork from it and join it again.
constexpr int kMaxPipes = 8;
hipStream_t g_pipe[kMaxPipes] = {};
hipEvent_t g_fork_ev = nullptr, g_join_ev[kMaxPipes] = {};

hipStream_t pipe_stream(int p) {
  if (p <= 0) return stream();
  if (!g_pipe[p]) HIP_CHECK_THROW(hipStreamCreateWithFlags(&g_pipe[p], hipStreamNonBlocking));
  return g_pipe[p];
}

// ---- misc kernels ----
__global__ void __launch_bounds__(256) k_flat(const ss::MiscLaunch M) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M.ncols) ss::flat_unit(M, i);
}
__global__ void __launch_bounds__(256) k_night(const ss::MiscLaunch M) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M.ncols) {
    ss::zero_unit(M.in, M.sw_dir, M.nsw, M.cols[i]);
    ss::zero_unit(M.in, M.sw_diff, M.nsw, M.cols[i]);
  }
}
__global__ void __launch_bounds__(256) k_simple_urban(const ss::MiscLaunch M) {
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nmax = M.nsw > M.nlw ? M.nsw : M.nlw;
  if (u < (long)M.ncols * nmax) ss::simple_urban_unit(M, (int)u);
}
// input pack (ss_pack.hpp): block -> (tile, layer) with the XCD-aware order of k_layer, so that the
// layers of a tile, which share the input lines, run in the same L2 at the same time
__global__ void __launch_bounds__(64) k_pack(const ss::PackLaunch P) {
  const int per_group = 8 * P.maxlay;
  const int group = blockIdx.x / per_group, r = blockIdx.x % per_group;
  const int tile = group * 8 + (r & 7), jlay = r >> 3;
  const int cu = tile * 64 + threadIdx.x;
  if (cu < P.ncu) ss::pack_unit(P, cu, jlay);
}
__global__ void __launch_bounds__(256) k_spectrum_col(const ss::SpectrumLaunch S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < S.ncols) ss::spectrum_col_unit(S, i);
}
__global__ void __launch_bounds__(256) k_spectrum_lay(const ss::SpectrumLaunch S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < S.nlays) ss::spectrum_lay_unit(S, i);
}

__global__ void __launch_bounds__(256) k_scale_sum(const ss::ScaleSumLaunch S) {
  // blockIdx.y = member kind (ss::scale_sum_unit); one aligned element pair per thread
  const int kind = blockIdx.y;
  long first;
  const long n = ss::scale_sum_pairs(S, kind, &first);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    ss::scale_sum_unit(S, kind, i);
}

enum Family { FAM_SW_LAYER = 0, FAM_SW_COLUMN, FAM_LW_LAYER, FAM_LW_COLUMN, FAM_MISC, FAM_SW_DOWN, FAM_LW_DOWN, FAM_COUNT };

struct HipBackend {
  // HIP-event timing of every launch of the last solve, on the plan's stream
  struct Timers {
    struct Span { int family; hipEvent_t a, b; };
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    std::vector<Span> spans;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool have_total = false;

    hipEvent_t next() {
      if (used == pool.size()) {
        hipEvent_t e;
        HIP_CHECK_THROW(hipEventCreate(&e));
        pool.push_back(e);
      }
      return pool[used++];
    }
    void begin_solve() {
      used = 0;
      spans.clear();
      t0 = next();
      t1 = next();
      have_total = false;
      HIP_CHECK_THROW(hipEventRecord(t0, stream()));
    }
    void end_solve() {
      HIP_CHECK_THROW(hipEventRecord(t1, stream()));
      have_total = true;
    }
    void release() {
      for (auto e : pool) hipEventDestroy(e);
      pool.clear();
      used = 0;
      spans.clear();
    }
    // sum (ms) and launch count of one family after synchronisation
    void query(int family, double* ms, int* count) const {
      double s = 0.0;
      int n = 0;
      for (const auto& sp : spans)
        if (sp.family == family) {
          float t = 0.f;
          if (hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess) { s += t; ++n; }
        }
      *ms = s;
      *count = n;
    }
    double total_ms() const {
      float t = 0.f;
      if (have_total && hipEventElapsedTime(&t, t0, t1) == hipSuccess) return t;
      return 0.0;
    }
  };

  static void* alloc(size_t n) {
    void* p = nullptr;
    HIP_CHECK_THROW(hipMalloc(&p, n ? n : 8));
    return p;
  }
  static void free(void* p) { if (p) hipFree(p); }
  static void memset(void* p, int v, size_t n) {
    HIP_CHECK_THROW(hipMemsetAsync(p, v, n, stream()));
  }
  static void memset_async(void* p, int v, size_t n) {
    HIP_CHECK_THROW(hipMemsetAsync(p, v, n, stream()));
  }
  static void h2d(void* d, const void* h, size_t n) {
    if (n) {
      HIP_CHECK_THROW(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, stream()));
      HIP_CHECK_THROW(hipStreamSynchronize(stream()));   // host buffer may be reused at once
    }
  }
  static void d2h(void* h, const void* d, size_t n) {
    if (n) {
      HIP_CHECK_THROW(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, stream()));
      HIP_CHECK_THROW(hipStreamSynchronize(stream()));
    }
  }
  static void sync() { HIP_CHECK_THROW(hipStreamSynchronize(stream())); }

  static void launch_flat(const ss::MiscLaunch& M, Timers& t) {
    hipEvent_t a = t.next(), b = t.next();
    hipEventRecord(a, stream());
    hipLaunchKernelGGL(k_flat, dim3((M.ncols + 255) / 256), dim3(256), 0, stream(), M);
    hipEventRecord(b, stream());
    t.spans.push_back({FAM_MISC, a, b});
  }
  static void launch_night(const ss::MiscLaunch& M, Timers& t) {
    hipEvent_t a = t.next(), b = t.next();
    hipEventRecord(a, stream());
    hipLaunchKernelGGL(k_night, dim3((M.ncols + 255) / 256), dim3(256), 0, stream(), M);
    hipEventRecord(b, stream());
    t.spans.push_back({FAM_MISC, a, b});
  }
  static void launch_simple_urban(const ss::MiscLaunch& M, Timers& t) {
    const long units = (long)M.ncols * std::max(M.nsw, M.nlw);
    hipEvent_t a = t.next(), b = t.next();
    hipEventRecord(a, stream());
    hipLaunchKernelGGL(k_simple_urban, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, stream(), M);
    hipEventRecord(b, stream());
    t.spans.push_back({FAM_MISC, a, b});
  }
  static void launch_pack(const ss::PackLaunch& P, Timers& t) {
    if (P.ncu <= 0 || P.maxlay <= 0) return;
    const unsigned tiles = (unsigned)((P.ncu + 63) / 64);
    hipEvent_t a = t.next(), b = t.next();
    hipEventRecord(a, stream());
    hipLaunchKernelGGL(k_pack, dim3(((tiles + 7) / 8) * 8 * (unsigned)P.maxlay), dim3(64), 0, stream(), P);
    hipEventRecord(b, stream());
    t.spans.push_back({FAM_MISC, a, b});
  }
  static void launch_scale_sum(const ss::ScaleSumLaunch& S, Timers& t) {
    const long nmax = ((long)S.nspec * std::max(S.ncols, S.nlays)) / 2 + 1;
    if (S.ncols <= 0) return;
    const unsigned gx = (unsigned)std::min<long>((nmax + 255) / 256, 1L << 20);
    hipEvent_t a = t.next(), b = t.next();
    hipEventRecord(a, stream());
    hipLaunchKernelGGL(k_scale_sum, dim3(gx, 4), dim3(256), 0, stream(), S);
    hipEventRecord(b, stream());
    t.spans.push_back({FAM_MISC, a, b});
  }
  static void launch_spectrum(const ss::SpectrumLaunch& S, Timers&) {
    if (S.ncols > 0)
      hipLaunchKernelGGL(k_spectrum_col, dim3((S.ncols + 255) / 256), dim3(256), 0, stream(), S);
    if (S.nlays > 0)
      hipLaunchKernelGGL(k_spectrum_lay, dim3((S.nlays + 255) / 256), dim3(256), 0, stream(), S);
  }
  // the pipes start after everything queued on the main stream so far ...
  static void fork(int npipes) {
    if (npipes <= 1) return;
    if (!g_fork_ev) HIP_CHECK_THROW(hipEventCreateWithFlags(&g_fork_ev, hipEventDisableTiming));
    HIP_CHECK_THROW(hipEventRecord(g_fork_ev, stream()));
    for (int p = 1; p < npipes; ++p) HIP_CHECK_THROW(hipStreamWaitEvent(pipe_stream(p), g_fork_ev, 0));
  }
  // ... and the main stream continues after all of them
  static void join(int npipes) {
    for (int p = 1; p < npipes; ++p) {
      if (!g_join_ev[p]) HIP_CHECK_THROW(hipEventCreateWithFlags(&g_join_ev[p], hipEventDisableTiming));
      HIP_CHECK_THROW(hipEventRecord(g_join_ev[p], pipe_stream(p)));
      HIP_CHECK_THROW(hipStreamWaitEvent(stream(), g_join_ev[p], 0));
    }
  }
  static void launch_solver(const ss::ClassKey& key, bool lw, const ss::Launch& L, Timers& t, int pipe) {
    ss::SolverLauncher fn = nullptr;
    if (!lw) {
      if (!key.urban) fn = key.nreg == 2 ? ss::launch_sw_forest_r2 : ss::launch_sw_forest_r3;
      else fn = key.nreg == 1 ? ss::launch_sw_urban_r1 : key.nreg == 2 ? ss::launch_sw_urban_r2 : ss::launch_sw_urban_r3;
    } else {
      if (!key.urban) fn = key.nreg == 2 ? ss::launch_lw_forest_r2 : ss::launch_lw_forest_r3;
      else fn = key.nreg == 1 ? ss::launch_lw_urban_r1 : key.nreg == 2 ? ss::launch_lw_urban_r2 : ss::launch_lw_urban_r3;
    }
    hipEvent_t ev[4] = {t.next(), t.next(), t.next(), t.next()};
    fn(key.ns, L, pipe_stream(pipe), ev);
    t.spans.push_back({lw ? FAM_LW_LAYER : FAM_SW_LAYER, ev[0], ev[1]});
    t.spans.push_back({lw ? FAM_LW_COLUMN : FAM_SW_COLUMN, ev[1], ev[2]});
    t.spans.push_back({lw ? FAM_LW_DOWN : FAM_SW_DOWN, ev[2], ev[3]});
    HIP_CHECK_THROW(hipGetLastError());
  }
};

using HipPlan = ss::Plan<HipBackend>;

int device_ready() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    g_last_error = "no HIP device available (this library has no CPU fallback)";
    return SS_ERR_NO_DEVICE;
  }
  return SS_OK;
}

template <class F>
int guarded(F&& f) {
  try {
    return f();
  } catch (hipError_t) {
    return SS_ERR_HIP;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return SS_ERR_BAD_ARGUMENT;
  }
}

}  // namespace

struct ss_plan {
  HipPlan impl;
};

extern "C" {

int ss_device_count(int32_t* count) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  if (count) *count = n;
  return n > 0 ? SS_OK : SS_ERR_NO_DEVICE;
}

int ss_set_device(int32_t device) {
  int rc = device_ready();
  if (rc) return rc;
  if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return SS_ERR_HIP;
  if (g_stream) { hipStreamDestroy(g_stream); g_stream = nullptr; }
  return SS_OK;
}

const char* ss_last_error(void) { return g_last_error.c_str(); }
const char* ss_version(void) { return "spartacus_surface_hip 0.1 (gfx950)"; }

int ss_plan_create(ss_plan** plan, const ss_config* config, const ss_canopy* canopy,
                   const ss_sw_spectral* sw, const ss_lw_spectral* lw, int32_t istartcol,
                   int32_t iendcol, int32_t want_profiles) {
  if (!plan || !config || !canopy) return SS_ERR_BAD_ARGUMENT;
  int rc = device_ready();
  if (rc) return rc;
  return guarded([&]() {
    ss_plan* p = new ss_plan();
    int r = p->impl.create(config, canopy, sw, lw, istartcol, iendcol, want_profiles);
    if (r) { g_last_error = p->impl.error; delete p; return r; }
    *plan = p;
    return (int)SS_OK;
  });
}

int ss_plan_upload(ss_plan* plan, const ss_canopy* canopy, const ss_sw_spectral* sw,
                   const ss_lw_spectral* lw) {
  if (!plan || !canopy) return SS_ERR_BAD_ARGUMENT;
  return guarded([&]() {
    int r = plan->impl.upload(canopy, sw, lw);
    if (r) g_last_error = plan->impl.error;
    return r;
  });
}

int ss_plan_calc_simple_spectrum_lw(ss_plan* plan) {
  if (!plan) return SS_ERR_BAD_ARGUMENT;
  return guarded([&]() {
    int r = plan->impl.simple_spectrum_lw();
    if (r) g_last_error = plan->impl.error;
    return r;
  });
}

int ss_plan_solve(ss_plan* plan) {
  if (!plan) return SS_ERR_BAD_ARGUMENT;
  return guarded([&]() {
    int r = plan->impl.solve();
    if (r) g_last_error = plan->impl.error;
    return r;
  });
}

int ss_plan_synchronize(ss_plan* plan) {
  if (!plan) return SS_ERR_BAD_ARGUMENT;
  return guarded([&]() { HipBackend::sync(); return (int)SS_OK; });
}

int ss_plan_download(ss_plan* plan, ss_bc_out* bc_out, ss_flux* sw_norm_dir, ss_flux* sw_norm_diff,
                     ss_flux* lw_internal, ss_flux* lw_norm) {
  if (!plan) return SS_ERR_BAD_ARGUMENT;
  return guarded([&]() { return plan->impl.download(bc_out, sw_norm_dir, sw_norm_diff, lw_internal, lw_norm); });
}

int ss_plan_last_kernel_ms(ss_plan* plan, const char* name, double* ms) {
  if (!plan || !name || !ms) return SS_ERR_BAD_ARGUMENT;
  return guarded([&]() {
    HipBackend::sync();
    const std::string n(name);
    int count = 0;
    // "<family>" = summed device time of the family in the last solve,
    // "<family>.avg" = average per launch, "<family>.launches" = launch count
    std::string fam = n, suffix;
    const size_t dot = n.find('.');
    if (dot != std::string::npos) { fam = n.substr(0, dot); suffix = n.substr(dot + 1); }
    if (fam == "total") { *ms = plan->impl.timers.total_ms(); return (int)SS_OK; }
    int f = -1;
    if (fam == "sw_layer") f = FAM_SW_LAYER;
    else if (fam == "sw_column") f = FAM_SW_COLUMN;
    else if (fam == "lw_layer") f = FAM_LW_LAYER;
    else if (fam == "lw_column") f = FAM_LW_COLUMN;
    else if (fam == "sw_down") f = FAM_SW_DOWN;
    else if (fam == "lw_down") f = FAM_LW_DOWN;
    else if (fam == "misc") f = FAM_MISC;
    if (f < 0) { g_last_error = "unknown kernel family"; return (int)SS_ERR_BAD_ARGUMENT; }
    double s = 0.0;
    plan->impl.timers.query(f, &s, &count);
    if (suffix == "avg") *ms = count ? s / count : 0.0;
    else if (suffix == "launches") *ms = (double)count;
    else *ms = s;
    return (int)SS_OK;
  });
}

int ss_plan_eigen_failures(ss_plan* plan, int64_t* count) {
  if (!plan || !count) return SS_ERR_BAD_ARGUMENT;
  return guarded([&]() { *count = plan->impl.eigen_failures(); return (int)SS_OK; });
}

int ss_plan_scale_and_sum(ss_plan* plan, const double* top_flux_dn_sw, const double* top_flux_dn_direct_sw,
                          const double* top_flux_dn_lw) {
  if (!plan) return SS_ERR_BAD_ARGUMENT;
  return guarded([&]() {
    int rc = plan->impl.scale_and_sum(top_flux_dn_sw, top_flux_dn_direct_sw, top_flux_dn_lw);
    if (rc) g_last_error = plan->impl.error;
    return rc;
  });
}

int ss_plan_download_sums(ss_plan* plan, ss_flux* sw_flux, ss_flux* lw_flux) {
  if (!plan) return SS_ERR_BAD_ARGUMENT;
  return guarded([&]() { return plan->impl.download_sums(sw_flux, lw_flux); });
}

int ss_plan_set_pipes(ss_plan* plan, int32_t npipes) {
  if (!plan || npipes < 1) return SS_ERR_BAD_ARGUMENT;
  return guarded([&]() { return plan->impl.set_pipes(npipes); });
}

int ss_plan_device_ptr(ss_plan* plan, const char* name, void** ptr, int64_t* nelem) {
  if (!plan || !name || !ptr) return SS_ERR_BAD_ARGUMENT;
  auto it = plan->impl.dev.find(name);
  if (it == plan->impl.dev.end()) { g_last_error = "no such device array"; return SS_ERR_BAD_ARGUMENT; }
  *ptr = it->second;
  if (nelem) *nelem = (int64_t)plan->impl.dev_elems[name];
  return SS_OK;
}

int ss_plan_destroy(ss_plan* plan) {
  if (!plan) return SS_OK;
  return guarded([&]() { HipBackend::sync(); delete plan; return (int)SS_OK; });
}

int spartacus_radsurf_hip(const ss_config* config, const ss_canopy* canopy, const ss_sw_spectral* sw,
                          const ss_lw_spectral* lw, ss_bc_out* bc_out, int32_t istartcol,
                          int32_t iendcol, ss_flux* sw_norm_dir, ss_flux* sw_norm_diff,
                          ss_flux* lw_internal, ss_flux* lw_norm) {
  if (!config || !canopy || !bc_out) return SS_ERR_BAD_ARGUMENT;
  int rc = device_ready();
  if (rc) return rc;
  // The reference is called concurrently on disjoint column ranges
  // (spartacus_surface_driver.F90:206-233); one device, one stream: serialise.
  std::lock_guard<std::mutex> lock(g_device_mutex);
  return guarded([&]() {
    const bool profiles = (sw_norm_dir && sw_norm_dir->flux_dn_layer_top)
                          || (lw_internal && lw_internal->flux_dn_layer_top);
    // A host model calls this entry once per time step with the same configuration and
    // shapes: the plan (device mirrors, scratch: GiBs of hipMalloc) of the previous call is
    // kept and only re-filled, unless SPARTACUS_NO_PLAN_CACHE is set.
    static HipPlan* cached = nullptr;   // deliberately never destroyed at process exit (HIP may be gone)
    static ss_config cached_cfg;
    static int cached_dims[5] = {-1, -1, -1, -1, -1};
    const int dims[5] = {canopy->ncol, canopy->ntotlay, istartcol, iendcol, profiles ? 1 : 0};
    const bool reuse = cached && !std::getenv("SPARTACUS_NO_PLAN_CACHE")
                       && std::memcmp(&cached_cfg, config, sizeof(ss_config)) == 0
                       && std::memcmp(cached_dims, dims, sizeof(dims)) == 0;
    int r;
    if (reuse) {
      r = cached->upload(canopy, sw, lw);
    } else {
      delete cached;
      cached = new HipPlan;
      cached->pipes_cap = 2;   // less scratch to allocate for what may be a one-shot call
      r = cached->create(config, canopy, sw, lw, istartcol, iendcol, profiles ? 1 : 0);
      cached_cfg = *config;
      std::memcpy(cached_dims, dims, sizeof(dims));
    }
    HipPlan& plan = *cached;
    if (r) { g_last_error = plan.error; delete cached; cached = nullptr; return r; }
    plan.upload_outputs(bc_out, sw_norm_dir, sw_norm_diff, lw_internal, lw_norm);
    r = plan.solve();
    if (r) { g_last_error = plan.error; delete cached; cached = nullptr; return r; }
    r = plan.download(bc_out, sw_norm_dir, sw_norm_diff, lw_internal, lw_norm);
    if (std::getenv("SPARTACUS_NO_PLAN_CACHE")) { delete cached; cached = nullptr; }
    return r;
  });
}

}  // extern "C"
