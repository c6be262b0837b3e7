// ss_capi.hip - the C-ABI of include/spartacus_surface_hip.h on the HIP backend.
//
// Everything here runs on the GPU or fails: there is no CPU fallback.  If no
// HIP device is usable every entry point returns SS_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
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

constexpr int kMaxPipes = 8;

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

// Page-locked caller arrays (hipHostRegister): copies then move by DMA at PCIe rate without blocking
// the host thread, so uploads, kernels and downloads of successive chunks overlap in both directions;
// pageable arrays (the default) go through the runtime's staging path, which blocks per copy.
//
// OWNERSHIP - who may keep a registration, and why (round 2 lost a process here).
// A registration describes an ADDRESS RANGE, and the runtime keeps using it after the memory behind it
// is gone: when the caller frees an array (munmap for anything above malloc's mmap threshold) and the
// allocator maps NEW pages at the same address, a copy "through" the stale registration is a DMA to
// pages that no longer exist - the GPU raises a memory access fault and the runtime kills the process
// (tools/experiments/pin_probe.hip, scenario munmap_remap_copy, reproduces exactly that; every other
// staleness scenario - heap reuse, unregistering unmapped ranges, overlapping registrations - is benign).
// The library cannot see a munmap, so it never keeps a registration it was not explicitly given:
//  * ss_host_pin(ptr, bytes): the CALLER owns the registration and promises the range stays mapped
//    until ss_host_unpin(ptr) / ss_host_unpin_all() (a host model with allocatable arrays that live for
//    the run: pin once after allocation);
//  * SPARTACUS_PIN_HOST=1: the CALL owns its registrations - made on entry for the arrays of this call
//    (alive by the contract of the call), dropped before the entry returns.  Nothing of it survives.
// Ranges never overlap in the registry (a request that partly overlaps an entry is refused and that array
// stays pageable); a copy whose host range crosses the end of a registration is split there, because the
// runtime rejects a copy that starts inside a registration and ends outside it (probe: copy_straddles).
struct PinRegistry {
  struct Entry { size_t bytes; bool call_owned; };
  mutable std::mutex mu;                    // plan-API threads copy while host-entry threads pin
  std::map<const char*, Entry> ranges;      // base -> entry, disjoint
  size_t bytes = 0;

  // end of the registration that contains p, or nullptr (caller holds mu)
  const char* end_of_registration_at(const char* p) const {
    auto it = ranges.upper_bound(p);
    if (it == ranges.begin()) return nullptr;
    --it;
    const char* end = it->first + it->second.bytes;
    return p < end ? end : nullptr;
  }
  bool covered(const void* p, size_t n) const {
    std::lock_guard<std::mutex> lock(mu);
    const char* end = end_of_registration_at((const char*)p);
    return end && (const char*)p + n <= end;
  }
  // how many leading bytes of [p, p + n) one asynchronous copy may take
  size_t copy_piece(const void* p, size_t n) const {
    std::lock_guard<std::mutex> lock(mu);
    if (ranges.empty()) return n;
    const char* end = end_of_registration_at((const char*)p);
    if (!end) return n;                     // starts outside every registration: the runtime stages it
    return std::min(n, (size_t)(end - (const char*)p));
  }
  bool pin(const void* p, size_t n, bool call_owned) {
    if (!p || n == 0) return true;
    std::lock_guard<std::mutex> lock(mu);
    const char *a = (const char*)p, *b = a + n;
    const char* end = end_of_registration_at(a);
    if (end && b <= end) return true;       // already page-locked
    // partial overlap with an entry: refuse (never unregister what someone else owns and may be using)
    auto it = ranges.lower_bound(a);
    if (end || (it != ranges.end() && it->first < b)) return false;
    // memory the caller page-locked itself (hipHostRegister / hipHostMalloc outside this library): not ours - a
    // second registration of the same base would succeed, and our release would then take the caller's lock (or
    // the device pointer it mapped) away.  It moves by DMA as it is; nothing is recorded.
    {
      hipPointerAttribute_t attr;
      if (hipPointerGetAttributes(&attr, p) == hipSuccess) {
        if (attr.type == hipMemoryTypeHost) return true;
      } else {
        (void)hipGetLastError();            // plain pageable memory: "invalid value"
      }
    }
    if (hipHostRegister(const_cast<void*>(p), n, hipHostRegisterPortable) != hipSuccess) {   // every device of the call
      (void)hipGetLastError();              // stays pageable: slower, still correct
      return false;
    }
    ranges[a] = {n, call_owned};
    bytes += n;
    return true;
  }
  // caller: nothing queued may still use the ranges (the host entry synchronises first)
  // returns the number of registrations dropped
  int release(bool call_owned_only, const void* only_base = nullptr) {
    std::lock_guard<std::mutex> lock(mu);
    int dropped = 0;
    for (auto it = ranges.begin(); it != ranges.end();) {
      if ((call_owned_only && !it->second.call_owned) || (only_base && it->first != (const char*)only_base)) { ++it; continue; }
      if (hipHostUnregister(const_cast<char*>(it->first)) != hipSuccess) (void)hipGetLastError();
      bytes -= it->second.bytes;
      it = ranges.erase(it);
      ++dropped;
    }
    return dropped;
  }
  size_t pinned_bytes() const { std::lock_guard<std::mutex> lock(mu); return bytes; }
};
PinRegistry g_pins;

// Launch check: a kernel launch reports configuration errors through hipGetLastError
#define SS_LAUNCH_CHECK() HIP_CHECK_THROW(hipGetLastError())

// Device context of ONE plan: the device it was created on, its main stream and
// the worker streams ("pipes" of ss_plan_impl.hpp: pipe 0 is the main stream,
// further pipes fork from it and join it again) with their events.  Nothing here
// is process-global: two plans never share a stream, and a plan created on device
// d keeps working after the process makes another device current.
struct HipBackend {
  int device = -1;
  hipStream_t main_ = nullptr;
  hipStream_t pipe_[kMaxPipes] = {};
  hipEvent_t fork_ev_ = nullptr, join_ev_[kMaxPipes] = {};

  // HIP-event timing of every launch of the last solve, on the plan's streams
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
    void release() {
      for (auto e : pool) hipEventDestroy(e);
      pool.clear();
      used = 0;
      spans.clear();
      have_total = false;
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

  void init() {
    if (device < 0) HIP_CHECK_THROW(hipGetDevice(&device));
  }
  // make the plan's device current (every entry point of a plan starts here)
  void activate() const {
    int cur = -1;
    if (device >= 0 && (hipGetDevice(&cur) != hipSuccess || cur != device)) HIP_CHECK_THROW(hipSetDevice(device));
  }
  void release() {
    if (device < 0) return;
    int cur = -1;
    const bool sw = hipGetDevice(&cur) == hipSuccess && cur != device && hipSetDevice(device) == hipSuccess;
    for (int p = 0; p < kMaxPipes; ++p) {
      if (pipe_[p]) hipStreamDestroy(pipe_[p]);
      if (join_ev_[p]) hipEventDestroy(join_ev_[p]);
      pipe_[p] = nullptr;
      join_ev_[p] = nullptr;
    }
    if (fork_ev_) hipEventDestroy(fork_ev_);
    for (hipEvent_t* e : {&up_ev_, &gate_ev_, &down_ev_, &ord_ev_}) { if (*e) hipEventDestroy(*e); *e = nullptr; }
    if (main_) hipStreamDestroy(main_);
    if (arena_) hipHostFree(arena_);
    arena_ = nullptr;
    arena_cap_ = arena_used_ = 0;
    fork_ev_ = nullptr;
    main_ = nullptr;
    if (sw) hipSetDevice(cur);
  }
  hipStream_t stream() {
    if (!main_) HIP_CHECK_THROW(hipStreamCreateWithFlags(&main_, hipStreamNonBlocking));
    return main_;
  }

  // Optional transfer streams that are NOT the plan's (the host entry lends its two sub-plans
  // one upload and one download stream).  Measured on this platform (tools/pipeline_probe*.py):
  // host-to-device and device-to-host copies only run at the same time when each direction has
  // a stream of its own - two streams that each carry upload, kernels, download of a chunk do
  // not overlap their copies at all (17.3 ms = the sum, for 4 x (102 MB up + 146 MB down)),
  // a dedicated upload stream, the compute stream and a dedicated download stream with events
  // between them do (11.9 ms).  Without them (up_ = down_ = nullptr) everything stays on the
  // main stream, in order.
  hipStream_t up_ = nullptr, down_ = nullptr;
  hipEvent_t up_ev_ = nullptr, gate_ev_ = nullptr, down_ev_ = nullptr, ord_ev_ = nullptr;
  bool up_pending_ = false, down_pending_ = false, main_dirty_ = true;
  void set_transfer_streams(hipStream_t up, hipStream_t down) { up_ = up; down_ = down; }
  hipStream_t up_stream() { return up_ ? up_ : stream(); }
  // kernels of the main stream start after the uploads queued so far
  void join_uploads() {
    main_dirty_ = true;
    if (!up_ || !up_pending_) return;
    if (!up_ev_) HIP_CHECK_THROW(hipEventCreateWithFlags(&up_ev_, hipEventDisableTiming));
    HIP_CHECK_THROW(hipEventRecord(up_ev_, up_));
    HIP_CHECK_THROW(hipStreamWaitEvent(stream(), up_ev_, 0));
    up_pending_ = false;
  }
  // uploads queued from now on start after everything queued on the main stream so far
  // (a chunk's inputs must not be overwritten while the previous chunk's kernels read them:
  // the host entry waits for the previous chunk's DOWNLOADS before it re-targets a plan, which
  // implies that; this is for callers of the plan API that upload while a solve is in flight)
  void order_uploads_after_main() {
    if (!up_) return;
    if (!ord_ev_) HIP_CHECK_THROW(hipEventCreateWithFlags(&ord_ev_, hipEventDisableTiming));
    HIP_CHECK_THROW(hipEventRecord(ord_ev_, stream()));
    HIP_CHECK_THROW(hipStreamWaitEvent(up_, ord_ev_, 0));
  }
  hipStream_t pipe_stream(int p) {
    if (p <= 0) return stream();
    if (!pipe_[p]) HIP_CHECK_THROW(hipStreamCreateWithFlags(&pipe_[p], hipStreamNonBlocking));
    return pipe_[p];
  }

  void begin_solve(Timers& t) {
    t.used = 0;
    t.spans.clear();
    t.t0 = t.next();
    t.t1 = t.next();
    t.have_total = false;
    join_uploads();
    HIP_CHECK_THROW(hipEventRecord(t.t0, stream()));
  }
  void end_solve(Timers& t) {
    HIP_CHECK_THROW(hipEventRecord(t.t1, stream()));
    t.have_total = true;
  }

  void* alloc(size_t n) {
    void* p = nullptr;
    HIP_CHECK_THROW(hipMalloc(&p, n ? n : 8));
    return p;
  }
  // One asynchronous copy per piece of the host range that lies inside ONE registration or outside all
  // of them (PinRegistry::copy_piece): the runtime rejects a copy that leaves the registration it starts in.
  static void copy_async(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s) {
    const bool host_is_src = kind == hipMemcpyHostToDevice;
    char* d = static_cast<char*>(dst);
    const char* h = static_cast<const char*>(src);
    while (n > 0) {
      const size_t m = g_pins.copy_piece(host_is_src ? (const void*)h : (const void*)d, n);
      const hipError_t e = hipMemcpyAsync(d, h, m, kind, s);
      if (e != hipSuccess) {
        g_last_error = std::string("hipMemcpyAsync: ") + hipGetErrorString(e);
        throw e;
      }
      d += m;
      h += m;
      n -= m;
    }
  }
  void free(void* p) { if (p) hipFree(p); }
  // initialisation of freshly allocated arrays: same stream as the uploads that may follow into them
  void memset(void* p, int v, size_t n) {
    HIP_CHECK_THROW(hipMemsetAsync(p, v, n, up_stream()));
    up_pending_ = true;
  }
  void memset_async(void* p, int v, size_t n) { HIP_CHECK_THROW(hipMemsetAsync(p, v, n, stream())); }
  // Copies are queued on the plan's main stream without a synchronisation of their
  // own: from pageable memory the runtime has staged the host buffer when the call
  // returns, and the plan synchronises once at the end of upload() / download()
  // (one wait per call instead of one per array).
  void h2d(void* d, const void* h, size_t n) {
    if (!n) return;
    copy_async(d, h, n, hipMemcpyHostToDevice, up_stream());
    up_pending_ = true;
  }
  // Downloads: on the download stream if one was lent (it first waits for everything queued on
  // the main stream), else on the main stream behind the kernels.  Extra copy streams per plan
  // were tried and are worse: HIP maps streams onto a handful of hardware queues and a queue
  // executes in order, so they end up sharing a queue with another plan's kernels.
  void d2h(void* h, const void* d, size_t n) {
    if (!n) return;
    if (down_) {
      if (main_dirty_) {
        if (!gate_ev_) HIP_CHECK_THROW(hipEventCreateWithFlags(&gate_ev_, hipEventDisableTiming));
        HIP_CHECK_THROW(hipEventRecord(gate_ev_, stream()));
        HIP_CHECK_THROW(hipStreamWaitEvent(down_, gate_ev_, 0));
        main_dirty_ = false;
      }
      copy_async(h, d, n, hipMemcpyDeviceToHost, down_);
      down_pending_ = true;
      return;
    }
    copy_async(h, d, n, hipMemcpyDeviceToHost, stream());
  }
  // The stream downloads of this plan's results go to, ordered behind everything queued on the main stream so far
  hipStream_t gate_downloads() {
    if (!down_) return stream();
    if (!gate_ev_) HIP_CHECK_THROW(hipEventCreateWithFlags(&gate_ev_, hipEventDisableTiming));
    HIP_CHECK_THROW(hipEventRecord(gate_ev_, stream()));
    HIP_CHECK_THROW(hipStreamWaitEvent(down_, gate_ev_, 0));
    main_dirty_ = false;
    return down_;
  }
  void gate_copies() {}      // copies queued now already run after everything queued before them
  // (One copy KERNEL per chunk that stores straight into mapped host memory, instead of ~35
  // DMA copies with ~30 us between them, was measured slower - 1.18e7 -> 0.90e7 columns/s: its
  // workgroups wait for compute units that the solver's one-wave-per-SIMD kernels hold.)
  void d2h_batch_begin() {}
  void d2h_batch_end() {}
  // wait for this plan's work: uploads, kernels, and ITS downloads (not the other plan's on a lent stream)
  void sync() {
    if (up_ && up_pending_) join_uploads();
    HIP_CHECK_THROW(hipStreamSynchronize(stream()));
    if (down_ && down_pending_) {
      if (!down_marked_) downloads_queued();
      HIP_CHECK_THROW(hipEventSynchronize(down_ev_));
      down_pending_ = down_marked_ = false;
    }
  }
  // mark the end of a batch of downloads without waiting (the later sync() waits for exactly these)
  bool down_marked_ = false;
  void downloads_queued() {
    if (!down_ || !down_pending_) return;
    if (!down_ev_) HIP_CHECK_THROW(hipEventCreateWithFlags(&down_ev_, hipEventDisableTiming));
    HIP_CHECK_THROW(hipEventRecord(down_ev_, down_));
    down_marked_ = true;
  }

  // The plan's OWN small host vectors (layer counts, column index lists) go through a
  // page-locked arena of the plan: a copy from pageable memory would make the call wait for
  // everything queued on the stream before it - with page-locked caller arrays that is the
  // whole upload of the chunk, and the host thread could not go on to fetch the previous
  // chunk's results meanwhile.  arena_reset() at the start of an upload: the copies of the
  // previous upload of this plan have completed by then (upload / download synchronise).
  char* arena_ = nullptr;
  size_t arena_cap_ = 0, arena_used_ = 0, arena_want_ = 0;
  void arena_reset() {
    // grow between uploads, to twice what the last upload asked for (never while copies from it may be pending
    // within an upload: a request that does not fit falls back to a plain copy for that one upload)
    if (arena_want_ > arena_cap_) {
      if (arena_) hipHostFree(arena_);
      arena_ = nullptr;
      arena_cap_ = 0;
      const size_t cap = std::max(arena_want_ * 2, (size_t)1 << 20);
      if (hipHostMalloc(reinterpret_cast<void**>(&arena_), cap, hipHostMallocDefault) == hipSuccess) arena_cap_ = cap;
      else { (void)hipGetLastError(); arena_ = nullptr; }
    }
    arena_used_ = arena_want_ = 0;
  }
  void h2d_staged(void* d, const void* h, size_t n) {
    if (!n) return;
    const size_t need = (arena_used_ + n + 63) & ~(size_t)63;
    arena_want_ = std::max(arena_want_, (arena_want_ + n + 63) & ~(size_t)63);
    if (need > arena_cap_ || !arena_) { h2d(d, h, n); return; }
    std::memcpy(arena_ + arena_used_, h, n);
    HIP_CHECK_THROW(hipMemcpyAsync(d, arena_ + arena_used_, n, hipMemcpyHostToDevice, up_stream()));
    up_pending_ = true;
    arena_used_ = need;
  }

  template <class K, class A>
  void timed_launch(Timers& t, K kernel, dim3 grid, dim3 block, const A& arg) {
    hipEvent_t a = t.next(), b = t.next();
    join_uploads();
    HIP_CHECK_THROW(hipEventRecord(a, stream()));
    hipLaunchKernelGGL(kernel, grid, block, 0, stream(), arg);
    SS_LAUNCH_CHECK();
    HIP_CHECK_THROW(hipEventRecord(b, stream()));
    t.spans.push_back({FAM_MISC, a, b});
  }
  void launch_flat(const ss::MiscLaunch& M, Timers& t) {
    if (M.ncols <= 0) return;
    timed_launch(t, k_flat, dim3((M.ncols + 255) / 256), dim3(256), M);
  }
  void launch_night(const ss::MiscLaunch& M, Timers& t) {
    if (M.ncols <= 0) return;
    timed_launch(t, k_night, dim3((M.ncols + 255) / 256), dim3(256), M);
  }
  void launch_simple_urban(const ss::MiscLaunch& M, Timers& t) {
    const long units = (long)M.ncols * std::max(M.nsw, M.nlw);
    if (units <= 0) return;
    timed_launch(t, k_simple_urban, dim3((unsigned)((units + 255) / 256)), dim3(256), M);
  }
  void launch_pack(const ss::PackLaunch& P, Timers& t) {
    if (P.ncu <= 0 || P.maxlay <= 0) return;
    const unsigned tiles = (unsigned)((P.ncu + 63) / 64);
    timed_launch(t, k_pack, dim3(((tiles + 7) / 8) * 8 * (unsigned)P.maxlay), dim3(64), P);
  }
  void launch_scale_sum(const ss::ScaleSumLaunch& S, Timers& t) {
    const long nmax = ((long)S.nspec * std::max(S.ncols, S.nlays)) / 2 + 1;
    if (S.ncols <= 0) return;
    const unsigned gx = (unsigned)std::min<long>((nmax + 255) / 256, 1L << 20);
    timed_launch(t, k_scale_sum, dim3(gx, 4), dim3(256), S);
  }
  void launch_spectrum(const ss::SpectrumLaunch& S, Timers&) {
    join_uploads();
    if (S.ncols > 0) {
      hipLaunchKernelGGL(k_spectrum_col, dim3((S.ncols + 255) / 256), dim3(256), 0, stream(), S);
      SS_LAUNCH_CHECK();
    }
    if (S.nlays > 0) {
      hipLaunchKernelGGL(k_spectrum_lay, dim3((S.nlays + 255) / 256), dim3(256), 0, stream(), S);
      SS_LAUNCH_CHECK();
    }
  }
  // the pipes start after everything queued on the main stream so far ...
  void fork(int npipes) {
    if (npipes <= 1) return;
    if (!fork_ev_) HIP_CHECK_THROW(hipEventCreateWithFlags(&fork_ev_, hipEventDisableTiming));
    HIP_CHECK_THROW(hipEventRecord(fork_ev_, stream()));
    for (int p = 1; p < npipes; ++p) HIP_CHECK_THROW(hipStreamWaitEvent(pipe_stream(p), fork_ev_, 0));
  }
  // ... and the main stream continues after all of them
  void join(int npipes) {
    for (int p = 1; p < npipes; ++p) {
      if (!join_ev_[p]) HIP_CHECK_THROW(hipEventCreateWithFlags(&join_ev_[p], hipEventDisableTiming));
      HIP_CHECK_THROW(hipEventRecord(join_ev_[p], pipe_stream(p)));
      HIP_CHECK_THROW(hipStreamWaitEvent(stream(), join_ev_[p], 0));
    }
  }
  void launch_solver(const ss::ClassKey& key, bool lw, const ss::Launch& L, Timers& t, int pipe) {
    ss::SolverLauncher fn = nullptr;
    if (!lw) {
      if (!key.urban) fn = key.nreg == 2 ? ss::launch_sw_forest_r2 : ss::launch_sw_forest_r3;
      else fn = key.nreg == 1 ? ss::launch_sw_urban_r1 : key.nreg == 2 ? ss::launch_sw_urban_r2 : ss::launch_sw_urban_r3;
    } else {
      if (!key.urban) fn = key.nreg == 2 ? ss::launch_lw_forest_r2 : ss::launch_lw_forest_r3;
      else fn = key.nreg == 1 ? ss::launch_lw_urban_r1 : key.nreg == 2 ? ss::launch_lw_urban_r2 : ss::launch_lw_urban_r3;
    }
    // four span marks + two events for the fallback-stream hand-over of the cooperative orders
    hipEvent_t ev[6] = {t.next(), t.next(), t.next(), t.next(), t.next(), t.next()};
    const int rc = fn(key.ns, L, pipe_stream(pipe), ev, pipe_stream(0));
    if (rc != 0) {
      // SS_LAUNCH_BAD_RECORD: the plan's record sizes disagree with the kernels' (a build
      // inconsistency); anything else is the hipError_t of the failing launch / event record
      g_last_error = rc == ss::kLaunchBadRecord
                         ? std::string("scratch record size mismatch between plan and kernels")
                         : std::string("solver kernel launch: ") + hipGetErrorString((hipError_t)rc);
      throw (hipError_t)(rc == ss::kLaunchBadRecord ? hipErrorInvalidValue : (hipError_t)rc);
    }
    t.spans.push_back({lw ? FAM_LW_LAYER : FAM_SW_LAYER, ev[0], ev[1]});
    t.spans.push_back({lw ? FAM_LW_COLUMN : FAM_SW_COLUMN, ev[1], ev[2]});
    t.spans.push_back({lw ? FAM_LW_DOWN : FAM_SW_DOWN, ev[2], ev[3]});
  }
};

using HipPlan = ss::Plan<HipBackend>;

int device_ready() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    g_last_error = "no HIP device available (this library has no CPU fallback)";
    return SS_ERR_NO_DEVICE;
  }
  // an error the HOST left behind in this thread (a failed call of its own) is not ours: the launch checks of
  // this library read hipGetLastError and would report it as a failed kernel launch
  (void)hipGetLastError();
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

namespace {

// ---- host entry: column chunks through three alternating device plans ----
//
// spartacus_radsurf_hip gets host arrays and must hand host arrays back, so its
// time is PCIe time: ~2 KB up and ~3.4 KB down per column (C3) against 24 ns of
// kernels.  The columns of a call are therefore cut into chunks and the chunks
// alternate between three sub-plans with their own device arrays and compute stream: while
// chunk k is solved, chunk k+1 is uploaded and chunk k-1 downloaded, and the host
// thread only ever waits for copies.  A sub-plan sees a chunk as a column set of
// its own: every pointer of the caller's structs is advanced to the chunk's first
// column / first packed layer and istartlay is re-based.  The sub-plans persist
// between calls (key: configuration, profile flag, device); their device arrays
// only grow, so a driver that calls with a different column block every time
// (nblocksize << ncol, driver/spartacus_surface_driver.F90:206-233) re-uses them
// instead of re-allocating - only the slices of the block cross the bus.
struct HostEntry {
  static constexpr int kSlots = 3;
  HipPlan* slot[kSlots] = {nullptr, nullptr, nullptr};
  // one upload and one download stream for both sub-plans: each direction of the bus needs a
  // stream of its own to run at the same time as the other (HipBackend::set_transfer_streams)
  hipStream_t up = nullptr, down = nullptr;
  // call-level device arrays of the per-column outputs (bc_out, the (nspec, ncol) and (ncol)
  // members of the flux objects): the sub-plans write their chunk's slice of them, and they come
  // back in ONE copy per member at the end of the call instead of one per member and chunk -
  // these copies are 200-400 KB per chunk, all start-up latency on the download engine
  std::map<std::string, std::pair<double*, size_t>> arena;
  double* arena_get(const std::string& name, size_t elems) {
    auto& a = arena[name];
    if (a.second < elems) {
      if (a.first) hipFree(a.first);
      a.first = nullptr;
      HIP_CHECK_THROW(hipMalloc(reinterpret_cast<void**>(&a.first), std::max<size_t>(elems, 1) * sizeof(double)));
      a.second = elems;
      // never hand uninitialised device memory towards the caller's arrays (blocking: new arrays are rare)
      HIP_CHECK_THROW(hipMemset(a.first, 0, std::max<size_t>(elems, 1) * sizeof(double)));
    }
    return a.first;
  }
  ss_config cfg;
  int profiles = -1, device = -1;

  void streams() {
    if (!up) HIP_CHECK_THROW(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    if (!down) HIP_CHECK_THROW(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
  }
  void drop() {
    for (auto& p : slot) { delete p; p = nullptr; }
    for (auto& kv : arena)
      if (kv.second.first) hipFree(kv.second.first);
    arena.clear();
    if (up) hipStreamDestroy(up);
    if (down) hipStreamDestroy(down);
    up = down = nullptr;
    profiles = -1;
    device = -1;
  }
};
// One HostEntry per shard of a large call: shard 0 is the calling thread's current device (the only one unless
// devices were configured, ss_set_devices / SPARTACUS_DEVICES); deliberately never destroyed at process exit
// (HIP may be gone).  Guarded by g_device_mutex (large calls are exclusive).
std::vector<HostEntry*>& g_hosts = *new std::vector<HostEntry*>();
HostEntry& host_entry(size_t shard) {
  while (g_hosts.size() <= shard) g_hosts.push_back(new HostEntry);
  return *g_hosts[shard];
}
std::atomic<long> g_plan_creations{0};   // sub-plans built by the host entry so far, small-call slots included

// Devices of the host entry (row e of the scope: columns shard over the GPUs of a node; the reference's only
// parallel construct is one process looping over column blocks, driver/spartacus_surface_driver.F90:203-234, so
// the fan-out lives HERE, inside the one call its untouched driver makes).  Empty = the calling thread's current
// device.  A device may be listed more than once (several shards on one GPU: how the one-GPU test box
// exercises the path).  Set by ss_set_devices or, if that was never called, read once from SPARTACUS_DEVICES
// ("all", a count, or a comma-separated list).
std::vector<int> g_devices;
bool g_devices_set = false;
void devices_from_env() {
  if (g_devices_set) return;
  g_devices_set = true;
  const char* e = std::getenv("SPARTACUS_DEVICES");
  if (!e || !*e) return;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  const std::string v(e);
  if (v == "all") {
    for (int d = 0; d < n; ++d) g_devices.push_back(d);
  } else if (v.find(',') == std::string::npos) {
    const int k = std::atoi(v.c_str());
    for (int d = 0; d < k && d < n; ++d) g_devices.push_back(d);
  } else {
    size_t a = 0;
    while (a <= v.size()) {
      const size_t b = v.find(',', a);
      const std::string tok = v.substr(a, b == std::string::npos ? std::string::npos : b - a);
      if (!tok.empty()) {
        const int d = std::atoi(tok.c_str());
        if (d >= 0 && d < n) g_devices.push_back(d);
      }
      if (b == std::string::npos) break;
      a = b + 1;
    }
  }
}


// SPARTACUS_PIN_HOST=1: page-lock the slices of the caller's arrays this call touches - columns
// [c0, c1), packed layers [l0, l1) - as registrations OWNED BY THE CALL (dropped before it returns).
void pin_call_arrays(const ss_config* cfg, const ss_canopy* can, const ss_sw_spectral* sw, const ss_lw_spectral* lw,
                     ss_bc_out* bc, ss_flux* const* fl, long c0, long c1, long l0, long l1) {
  const size_t nc = (size_t)(c1 - c0), nl = (size_t)(l1 - l0), D = sizeof(double);
  auto cols = [&](const double* p, size_t w) { if (p) g_pins.pin(p + w * (size_t)c0, w * nc * D, true); };
  auto lays = [&](const double* p, size_t w) { if (p && nl) g_pins.pin(p + w * (size_t)l0, w * nl * D, true); };
  cols(can->cos_sza, 1);
  const double* lay[] = {can->dz, can->building_fraction, can->veg_fraction, can->building_scale, can->veg_scale,
                         can->veg_ext, can->veg_fsd, can->veg_contact_fraction};
  for (const double* p : lay) lays(p, 1);
  if (sw && cfg->do_sw) {
    const size_t S = (size_t)cfg->nsw;
    const double* a[] = {sw->air_ext, sw->air_ssa, sw->veg_ssa, sw->roof_albedo, sw->wall_albedo, sw->roof_albedo_dir,
                         sw->wall_specular_frac};
    for (const double* p : a) lays(p, S);
    cols(sw->ground_albedo, S);
    cols(sw->ground_albedo_dir, S);
  }
  if (lw && cfg->do_lw) {
    const size_t W = (size_t)cfg->nlw;
    const double* a[] = {lw->air_ext, lw->air_ssa, lw->veg_ssa, lw->clear_air_planck, lw->veg_planck,
                         lw->veg_air_planck, lw->roof_emissivity, lw->wall_emissivity, lw->roof_emission,
                         lw->wall_emission};
    for (const double* p : a) lays(p, W);
    cols(lw->ground_emissivity, W);
    cols(lw->ground_emission, W);
  }
  cols(bc->sw_albedo, (size_t)cfg->nsw);
  cols(bc->sw_albedo_dir, (size_t)cfg->nsw);
  cols(bc->lw_emissivity, (size_t)cfg->nlw);
  cols(bc->lw_emission, (size_t)cfg->nlw);
  for (int k = 0; k < 4; ++k) {
    if (!fl[k]) continue;
    const size_t ns = (size_t)(k < 2 ? cfg->nsw : cfg->nlw);
    double* const* m = &fl[k]->ground_dn;
    for (int F = 0; F < ss::F_NFIELDS; ++F) {
      const int shape = ss::flux_field_shape(F);
      if (shape == 0) cols(m[F], ns);
      else if (shape == 1) lays(m[F], ns);
      else if (shape == 2) cols(m[F], 1);
      else lays(m[F], 1);
    }
  }
}

// packed-layer range [l0, l1) (0-based) of the columns [c0, c1)
void layer_range(const ss_canopy* can, int c0, int c1, long* l0_out, long* l1_out) {
  long l0 = -1, l1 = 0;
  for (int j = c0; j < c1; ++j) {
    if (can->nlay[j] <= 0) continue;
    const long a = can->istartlay[j] - 1;
    if (l0 < 0 || a < l0) l0 = a;
    l1 = std::max(l1, a + can->nlay[j]);
  }
  if (l0 < 0) l0 = l1 = 0;
  *l0_out = l0;
  *l1_out = l1;
}

struct ChunkView {
  ss_canopy can;
  ss_sw_spectral sw;
  ss_lw_spectral lw;
  ss_bc_out bc;
  ss_flux fl[4];
  ss_flux fl_layers[4];     // fl with the per-column members removed (what is downloaded per chunk)
  bool have_fl[4];
  std::vector<int32_t> istartlay;
};

template <class T>
T* adv(T* p, long off) { return p ? p + off : nullptr; }

// columns [c0, c1) (0-based) of the caller's objects as objects of their own
void make_view(const ss_canopy* can, const ss_sw_spectral* sw, const ss_lw_spectral* lw, ss_bc_out* bc,
               ss_flux* const* fl, int nsw, int nlw, int c0, int c1, ChunkView& v) {
  long l0, l1;
  layer_range(can, c0, c1, &l0, &l1);
  v.istartlay.resize(c1 - c0);
  for (int j = c0; j < c1; ++j)
    v.istartlay[j - c0] = can->nlay[j] > 0 ? (int32_t)(can->istartlay[j] - l0) : 1;
  v.can = *can;
  v.can.ncol = c1 - c0;
  v.can.ntotlay = (int32_t)(l1 - l0);
  v.can.nlay = can->nlay + c0;
  v.can.istartlay = v.istartlay.data();
  v.can.i_representation = can->i_representation + c0;
  v.can.cos_sza = adv(can->cos_sza, c0);
  v.can.ground_temperature = adv(can->ground_temperature, c0);
#define LAYF(m) v.can.m = adv(can->m, l0)
  LAYF(dz); LAYF(building_fraction); LAYF(veg_fraction); LAYF(building_scale); LAYF(veg_scale);
  LAYF(veg_ext); LAYF(veg_fsd); LAYF(veg_contact_fraction); LAYF(roof_temperature);
  LAYF(wall_temperature); LAYF(clear_air_temperature); LAYF(veg_temperature); LAYF(veg_air_temperature);
#undef LAYF
  if (sw) {
    v.sw = *sw;
    const long cl = (long)nsw * l0, cc = (long)nsw * c0;
    v.sw.air_ext = adv(sw->air_ext, cl); v.sw.air_ssa = adv(sw->air_ssa, cl); v.sw.veg_ssa = adv(sw->veg_ssa, cl);
    v.sw.ground_albedo = adv(sw->ground_albedo, cc); v.sw.ground_albedo_dir = adv(sw->ground_albedo_dir, cc);
    v.sw.roof_albedo = adv(sw->roof_albedo, cl); v.sw.wall_albedo = adv(sw->wall_albedo, cl);
    v.sw.roof_albedo_dir = adv(sw->roof_albedo_dir, cl); v.sw.wall_specular_frac = adv(sw->wall_specular_frac, cl);
  }
  if (lw) {
    v.lw = *lw;
    const long cl = (long)nlw * l0, cc = (long)nlw * c0;
    v.lw.air_ext = adv(lw->air_ext, cl); v.lw.air_ssa = adv(lw->air_ssa, cl); v.lw.veg_ssa = adv(lw->veg_ssa, cl);
    v.lw.clear_air_planck = adv(lw->clear_air_planck, cl); v.lw.veg_planck = adv(lw->veg_planck, cl);
    v.lw.veg_air_planck = adv(lw->veg_air_planck, cl);
    v.lw.ground_emissivity = adv(lw->ground_emissivity, cc); v.lw.ground_emission = adv(lw->ground_emission, cc);
    v.lw.roof_emissivity = adv(lw->roof_emissivity, cl); v.lw.wall_emissivity = adv(lw->wall_emissivity, cl);
    v.lw.roof_emission = adv(lw->roof_emission, cl); v.lw.wall_emission = adv(lw->wall_emission, cl);
  }
  v.bc.sw_albedo = adv(bc->sw_albedo, (long)nsw * c0);
  v.bc.sw_albedo_dir = adv(bc->sw_albedo_dir, (long)nsw * c0);
  v.bc.lw_emissivity = adv(bc->lw_emissivity, (long)nlw * c0);
  v.bc.lw_emission = adv(bc->lw_emission, (long)nlw * c0);
  for (int k = 0; k < 4; ++k) {
    v.have_fl[k] = fl[k] != nullptr;
    if (!fl[k]) continue;
    v.fl[k] = *fl[k];
    const int nspec = k < 2 ? nsw : nlw;
    v.fl[k].ncol = c1 - c0;
    v.fl[k].ntotlay = (int32_t)(l1 - l0);
    double** m = &v.fl[k].ground_dn;
    double* const* src = &fl[k]->ground_dn;
    for (int F = 0; F < ss::F_NFIELDS; ++F) {
      const int shape = ss::flux_field_shape(F);
      const long off = shape == 0 ? (long)nspec * c0 : shape == 1 ? (long)nspec * l0 : shape == 2 ? c0 : l0;
      m[F] = adv(src[F], off);
    }
    v.fl_layers[k] = v.fl[k];
    double** ml = &v.fl_layers[k].ground_dn;
    for (int F = 0; F < ss::F_NFIELDS; ++F)
      if (ss::flux_field_shape(F) == 0 || ss::flux_field_shape(F) == 2) ml[F] = nullptr;
  }
}

// the per-column output members of a call: name of the plan's device array, the caller's array,
// doubles per column
struct ColumnOutput {
  std::string name;
  double* host;
  int width;
};
// the per-layer output members of a call (same record: width = doubles per layer)
std::vector<ColumnOutput> layer_outputs(const ss_config* cfg, int profiles, ss_flux* const* fl);
// Only the members a plan of this configuration allocates and writes (flux_member_present): a caller
// may pass more (an LW flux object with its *_dir members allocated, roof members without do_urban);
// those stay untouched, exactly as the per-chunk download skips device arrays the plan does not have.
std::vector<ColumnOutput> column_outputs(const ss_config* cfg, int profiles, ss_bc_out* bc, ss_flux* const* fl) {
  static const char* obj[4] = {"sw_norm_dir", "sw_norm_diff", "lw_internal", "lw_norm"};
  std::vector<ColumnOutput> out;
  if (cfg->do_sw) {
    if (bc->sw_albedo) out.push_back({"bc_out.sw_albedo", bc->sw_albedo, cfg->nsw});
    if (bc->sw_albedo_dir) out.push_back({"bc_out.sw_albedo_dir", bc->sw_albedo_dir, cfg->nsw});
  }
  if (cfg->do_lw) {
    if (bc->lw_emissivity) out.push_back({"bc_out.lw_emissivity", bc->lw_emissivity, cfg->nlw});
    if (bc->lw_emission) out.push_back({"bc_out.lw_emission", bc->lw_emission, cfg->nlw});
  }
  for (int k = 0; k < 4; ++k) {
    if (!fl[k]) continue;
    double* const* m = &fl[k]->ground_dn;
    for (int F = 0; F < ss::F_NFIELDS; ++F) {
      const int shape = ss::flux_field_shape(F);
      if ((shape == 0 || shape == 2) && m[F] && ss::flux_member_present_for(*cfg, profiles != 0, F, k < 2))
        out.push_back({std::string(obj[k]) + "." + HipPlan::flux_field_name(F), m[F],
                       shape == 0 ? (k < 2 ? cfg->nsw : cfg->nlw) : 1});
    }
  }
  return out;
}

std::vector<ColumnOutput> layer_outputs(const ss_config* cfg, int profiles, ss_flux* const* fl) {
  static const char* obj[4] = {"sw_norm_dir", "sw_norm_diff", "lw_internal", "lw_norm"};
  std::vector<ColumnOutput> out;
  for (int k = 0; k < 4; ++k) {
    if (!fl[k]) continue;
    double* const* m = &fl[k]->ground_dn;
    for (int F = 0; F < ss::F_NFIELDS; ++F) {
      const int shape = ss::flux_field_shape(F);
      if ((shape == 1 || shape == 3) && m[F] && ss::flux_member_present_for(*cfg, profiles != 0, F, k < 2))
        out.push_back({std::string(obj[k]) + "." + HipPlan::flux_field_name(F), m[F],
                       shape == 1 ? (k < 2 ? cfg->nsw : cfg->nlw) : 1});
    }
  }
  return out;
}

// Columns of the next chunk.  The downloads are the longest stage of the pipeline (C3: 2.9 KB
// down against 2.0 KB up per column), so the download engine must start early and never run
// dry: a first chunk of a sixteenth of the call, then equal chunks of a sixth (every chunk
// costs ~60 copies whose start-up latencies add up on the engine: not smaller than 16k
// columns; not larger than 128k: the last chunk's download has nothing to overlap with),
// whole waves.  SPARTACUS_HOST_CHUNK=<columns> fixes the size.
// Pageable caller arrays: every copy blocks the calling thread, nothing overlaps, and what
// counts is the number of copies - chunks as large as the cap (measured, 200k columns: one
// chunk 27 ms, six chunks 38 ms).
int next_chunk_columns(int total, int previous, bool pinned) {
  const char* e = std::getenv("SPARTACUS_HOST_CHUNK");
  if (e && std::atoi(e) > 0) return std::atoi(e);
  if (!pinned) return 262144;
  int c = previous > 0 ? std::max(16384, total / 6) : std::max(8192, total / 16);
  c = std::min(c, 131072);
  return ((c + 63) / 64) * 64;
}

// ---- small calls: the driver's block loop ----
//
// The reference's driver calls radsurf from OpenMP threads with blocks of nblocksize = 16 columns
// (driver/spartacus_surface_driver.F90:206-234, driver/spartacus_surface_config.F90:25).  Through the
// chunk pipeline above such a call is ~70 blocking copies of a few hundred bytes (1.9 ms: 8e3 columns/s,
// five times slower than the reference on the host cores), and one mutex made the threads queue up.
// Calls of up to kSmallColumns columns take this path instead:
//  * a pool of kSlots sub-plans, each with a stream, a device slab and a page-locked host slab of its own
//    and a mutex of its own: concurrent callers work on different slots at the same time (the entry-level
//    lock is shared among them; large calls and re-configurations take it exclusively);
//  * every output member of the call is a slice of the slot's device slab, so the results come home in
//    ONE copy into the page-locked mirror and are scattered into the caller's arrays by the calling thread.
struct SmallSlot {
  std::mutex mu;
  HipPlan* plan = nullptr;
  double* slab = nullptr;       // device
  double* hslab = nullptr;      // page-locked host mirror
  size_t cap = 0;               // elements of both
  void ensure(size_t elems) {
    if (elems <= cap) return;
    if (slab) hipFree(slab);
    if (hslab) hipHostFree(hslab);
    slab = hslab = nullptr;
    cap = 0;
    const size_t n = std::max<size_t>(elems * 2, 1 << 16);
    HIP_CHECK_THROW(hipMalloc(reinterpret_cast<void**>(&slab), n * sizeof(double)));
    HIP_CHECK_THROW(hipMemset(slab, 0, n * sizeof(double)));
    HIP_CHECK_THROW(hipHostMalloc(reinterpret_cast<void**>(&hslab), n * sizeof(double), hipHostMallocDefault));
    cap = n;
  }
  void drop() {
    delete plan;
    plan = nullptr;
    if (slab) hipFree(slab);
    if (hslab) hipHostFree(hslab);
    slab = hslab = nullptr;
    cap = 0;
  }
};
struct SmallPool {
  static constexpr int kSlots = 8;
  SmallSlot slot[kSlots];
  std::atomic<unsigned> next{0};
  ss_config cfg;
  int profiles = -1, device = -1;
  bool matches(const ss_config* c, int prof, int dev) const {
    return profiles == prof && device == dev && std::memcmp(&cfg, c, sizeof(ss_config)) == 0;
  }
};
SmallPool g_small;                 // never destroyed at process exit (HIP may be gone)
std::shared_mutex g_entry_mutex;   // shared: small calls; exclusive: large calls, (re)configuration of the pool
constexpr int kSmallColumns = 2048;

int small_call(SmallSlot& sl, const ss_config* config, const ss_canopy* canopy, const ss_sw_spectral* sw,
               const ss_lw_spectral* lw, ss_bc_out* bc_out, int c_begin, int c_end, ss_flux* const* fl, int profiles) {
  ChunkView v;
  make_view(canopy, sw, lw, bc_out, fl, config->nsw, config->nlw, c_begin, c_end, v);
  long l0, l1;
  layer_range(canopy, c_begin, c_end, &l0, &l1);
  const std::vector<ColumnOutput> col_out = column_outputs(config, profiles, bc_out, fl);
  const std::vector<ColumnOutput> lay_out = layer_outputs(config, profiles, fl);
  const size_t ncols = (size_t)(c_end - c_begin), nlays = (size_t)(l1 - l0);
  // layout of the slab: every member at an even element (16-byte pairs)
  std::vector<size_t> off(col_out.size() + lay_out.size() + 1, 0);
  for (size_t i = 0; i < col_out.size(); ++i) off[i + 1] = off[i] + (((size_t)col_out[i].width * ncols + 1) & ~(size_t)1);
  for (size_t i = 0; i < lay_out.size(); ++i) {
    const size_t j = col_out.size() + i;
    off[j + 1] = off[j] + (((size_t)lay_out[i].width * std::max<size_t>(1, nlays) + 1) & ~(size_t)1);
  }
  const size_t total = off.back();
  sl.ensure(total);
  const bool fresh = !sl.plan;
  if (fresh) sl.plan = new HipPlan;
  HipPlan* p = sl.plan;
  p->unbind_externals();
  for (size_t i = 0; i < col_out.size(); ++i)
    p->bind_external(col_out[i].name, sl.slab + off[i], (size_t)col_out[i].width * ncols);
  for (size_t i = 0; i < lay_out.size(); ++i)
    p->bind_external(lay_out[i].name, sl.slab + off[col_out.size() + i], (size_t)lay_out[i].width * std::max<size_t>(1, nlays));
  int r;
  if (fresh) {
    p->pipes_cap = 1;          // everything of a slot on its one stream, in order: no events between streams
    ++g_plan_creations;
    r = p->create(config, &v.can, sw ? &v.sw : nullptr, lw ? &v.lw : nullptr, 1, c_end - c_begin, profiles);
  } else {
    r = p->reset(&v.can, sw ? &v.sw : nullptr, lw ? &v.lw : nullptr, 1, c_end - c_begin, false);
  }
  if (r) { g_last_error = p->error; return r; }
  auto fl_of = [&](int k) { return v.have_fl[k] ? &v.fl[k] : nullptr; };
  const bool always_seed = std::getenv("SPARTACUS_ALWAYS_SEED") != nullptr;
  p->upload_outputs(&v.bc, fl_of(0), fl_of(1), fl_of(2), fl_of(3), always_seed ? (int)HipPlan::SEED_ALL : p->seed_needed());
  r = p->solve();
  if (r) { g_last_error = p->error; return r; }
  hipStream_t st = p->be.stream();
  HIP_CHECK_THROW(hipMemcpyAsync(sl.hslab, sl.slab, total * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_CHECK_THROW(hipStreamSynchronize(st));
  for (size_t i = 0; i < col_out.size(); ++i) {
    const size_t w = (size_t)col_out[i].width;
    std::memcpy(col_out[i].host + w * (size_t)c_begin, sl.hslab + off[i], w * ncols * sizeof(double));
  }
  if (nlays > 0)
    for (size_t i = 0; i < lay_out.size(); ++i) {
      const size_t w = (size_t)lay_out[i].width;
      std::memcpy(lay_out[i].host + w * (size_t)l0, sl.hslab + off[col_out.size() + i], w * nlays * sizeof(double));
    }
  return SS_OK;
}

// One shard of a large call - columns [c_begin, c_end) of the caller's arrays - through the chunk pipeline of the
// HostEntry H on the calling thread's current device (= "device").  The caller's arrays are page-locked (or not)
// by the entry before the shards start; a shard never returns while its device may still touch them.
int large_call_shard(HostEntry& H, int device, const ss_config* config, const ss_canopy* canopy, const ss_sw_spectral* sw,
                     const ss_lw_spectral* lw, ss_bc_out* bc_out, int c_begin, int c_end, ss_flux* const* fl,
                     int profiles) {
  auto body = [&]() -> int {
    const bool nocache = std::getenv("SPARTACUS_NO_PLAN_CACHE") != nullptr;
    if (H.slot[0] && (nocache || H.profiles != profiles || H.device != device
                           || std::memcmp(&H.cfg, config, sizeof(ss_config)) != 0))
      H.drop();
    H.cfg = *config;
    H.profiles = profiles;
    H.device = device;

    long call_l0 = 0, call_l1 = 0;
    layer_range(canopy, c_begin, c_end, &call_l0, &call_l1);
    // are the caller's (large) arrays page-locked - by the lines above or by ss_host_pin?
    const bool pinned = canopy->dz && call_l1 > call_l0
                        && g_pins.covered(canopy->dz + call_l0, sizeof(double) * (size_t)(call_l1 - call_l0));
    const bool always_seed = std::getenv("SPARTACUS_ALWAYS_SEED") != nullptr;
    // per-column outputs: call-level device arrays, one download per member at the end
    const std::vector<ColumnOutput> col_out = column_outputs(config, profiles, bc_out, fl);
    std::vector<double*> col_dev(col_out.size());
    for (size_t i = 0; i < col_out.size(); ++i)
      col_dev[i] = H.arena_get(col_out[i].name, (size_t)col_out[i].width * (size_t)(c_end - c_begin));
    // per-layer outputs: call-level device arrays too - the sub-plans write their chunk's layer range of them,
    // and they come home in copies that do NOT follow the chunking: one per member for every group of chunks
    // (the download engine starts ~30 us late on every copy; 34 members x 7 chunks of such gaps were a third
    // of a pinned 200 000-column call)
    const std::vector<ColumnOutput> lay_out = layer_outputs(config, profiles, fl);
    std::vector<double*> lay_dev(lay_out.size());
    for (size_t i = 0; i < lay_out.size(); ++i)
      lay_dev[i] = H.arena_get(lay_out[i].name, (size_t)lay_out[i].width * (size_t)std::max<long>(1, call_l1 - call_l0));
    long col_home = c_begin;       // columns [c_begin, col_home): per-column members on their way to the host
    constexpr int NS = HostEntry::kSlots;
    ChunkView views[NS];
    bool inflight[NS] = {};     // kernels of the slot's chunk may still be running
    int r = SS_OK;
    auto fail = [&](HipPlan* p, int code) {
      if (p) g_last_error = p->error;
      return code;              // leave() synchronises, then drops the sub-plans
    };
    auto fl_of = [](ChunkView& v, int k) { return v.have_fl[k] ? &v.fl[k] : nullptr; };
    // SPARTACUS_HOST_TIMING=1: wall-clock milliseconds of the host thread per chunk and phase (stderr)
    const bool timing = std::getenv("SPARTACUS_HOST_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    int k = 0, chunk = 0;
    for (int c0 = c_begin; c0 < c_end; c0 += chunk, ++k) {
      chunk = next_chunk_columns(c_end - c_begin, chunk, pinned);
      const int c1 = std::min(c_end, c0 + chunk), s = k % NS;
      const double ta = now();
      HipPlan*& p = H.slot[s];
      if (inflight[s]) {          // the chunk NS back used this slot: its kernels must be done before its inputs are replaced
        p->be.sync();
        inflight[s] = false;
      }
      const double tw = now();
      ChunkView& v = views[s];
      make_view(canopy, sw, lw, bc_out, fl, config->nsw, config->nlw, c0, c1, v);
      const bool fresh = !p;
      if (fresh) p = new HipPlan;
      p->unbind_externals();      // bindings of an earlier call (other members, other sizes) end here
      for (size_t i = 0; i < col_out.size(); ++i)
        p->bind_external(col_out[i].name, col_dev[i] + (size_t)col_out[i].width * (size_t)(c0 - c_begin),
                         (size_t)col_out[i].width * (size_t)(c1 - c0));
      long chunk_l0, chunk_l1;
      layer_range(canopy, c0, c1, &chunk_l0, &chunk_l1);
      if (chunk_l1 <= chunk_l0) chunk_l0 = chunk_l1 = call_l0;
      for (size_t i = 0; i < lay_out.size(); ++i)
        p->bind_external(lay_out[i].name, lay_dev[i] + (size_t)lay_out[i].width * (size_t)(chunk_l0 - call_l0),
                         (size_t)lay_out[i].width * (size_t)std::max<long>(1, chunk_l1 - chunk_l0));
      if (fresh) {
        // one stream per sub-plan: the two sub-plans then sit on two hardware queues of their own, and
        // what this entry has to hide is PCIe time, not the (shorter) kernel time
        p->pipes_cap = 1;
        ++g_plan_creations;
        H.streams();
        // (a second download stream taken in turn by successive chunks, to fill the start-up gaps
        // between a chunk's ~60 copies, was measured slower: 1.04e7 -> 0.84e7 columns/s - more
        // streams than hardware queues)
        if (!std::getenv("SPARTACUS_HOST_ONE_STREAM")) p->be.set_transfer_streams(H.up, H.down);
        r = p->create(config, &v.can, sw ? &v.sw : nullptr, lw ? &v.lw : nullptr, 1, c1 - c0, profiles);
      } else {
        // no wait: the copies of this chunk stay queued in front of its kernels (the caller's
        // arrays are ours until this entry returns); page-locked arrays then move by DMA while
        // the host thread goes on
        r = p->reset(&v.can, sw ? &v.sw : nullptr, lw ? &v.lw : nullptr, 1, c1 - c0, false);
      }
      if (r) return fail(p, r);
      const double tb = now();
      p->upload_outputs(&v.bc, fl_of(v, 0), fl_of(v, 1), fl_of(v, 2), fl_of(v, 3),
                        always_seed ? (int)HipPlan::SEED_ALL : p->seed_needed());
      const double tc = now();
      r = p->solve();
      if (r) return fail(p, r);
      const double td = now();
      inflight[s] = true;
      // The per-layer results of this chunk come home behind its kernels: its OWN layer range (nothing is assumed
      // about the order of the columns' layers in the packed arrays), gated on its own stream.  (Fewer, larger
      // copies - every second / third chunk - were measured and are not faster: the transfer is bound by its bytes.)
      if (chunk_l1 > chunk_l0) {
        hipStream_t ds = p->be.gate_downloads();      // the download stream waits for everything queued on this plan's stream
        for (size_t i = 0; i < lay_out.size(); ++i) {
          const size_t w = (size_t)lay_out[i].width;
          HipBackend::copy_async(lay_out[i].host + w * (size_t)chunk_l0, lay_dev[i] + w * (size_t)(chunk_l0 - call_l0),
                                 w * (size_t)(chunk_l1 - chunk_l0) * sizeof(double), hipMemcpyDeviceToHost, ds);
        }
      }
      if (timing)
        std::fprintf(stderr, "host entry chunk %d (%d cols): wait for slot %.2f, view+upload %.2f, seed %.2f, solve queued %.2f, download queued %.2f ms\n",
                     k, c1 - c0, tw - ta, tb - tw, tc - tb, td - tc, now() - td);
    }
    for (int s = 0; s < NS; ++s) {
      if (!inflight[s]) continue;
      const double te = now();
      H.slot[s]->be.sync();
      if (timing) std::fprintf(stderr, "host entry: slot %d done after %.2f ms\n", s, now() - te);
    }
    // every kernel of the call has completed: the per-column outputs, one copy per member.  (Measured, 200 000
    // columns: sending them with every chunk - 27 more ~1 MB copies per download point - 17.1 -> 20.2 ms; once
    // at two thirds of the call and once at the end 18.7 ms; all at the end 17.3 ms.  Likewise the per-layer
    // members: with every chunk 17.3, every second 17.5, every third 18.1, all at the end 22.3 ms: the transfer
    // is bound by the bytes, ~45 GB/s down while ~30 GB/s go up, not by the number of copies.)
    {
      const double te = now();
      hipStream_t ds = H.down ? H.down : nullptr;
      for (size_t i = 0; i < col_out.size(); ++i) {
        const size_t w = (size_t)col_out[i].width;
        HipBackend::copy_async(col_out[i].host + w * (size_t)col_home, col_dev[i] + w * (size_t)(col_home - c_begin),
                               w * (size_t)(c_end - col_home) * sizeof(double), hipMemcpyDeviceToHost, ds);
      }
      HIP_CHECK_THROW(hipStreamSynchronize(ds));
      if (timing) std::fprintf(stderr, "host entry: %zu per-column members home after %.2f ms\n", col_out.size(), now() - te);
    }
    if (timing) std::fprintf(stderr, "host entry total %.2f ms\n", now() - t_start);
    if (nocache) H.drop();
    return r;
  };
  auto leave = [&](bool failed) {
    if (failed) {
      (void)hipDeviceSynchronize();
      (void)hipGetLastError();
      H.drop();            // sub-plans with half-done state (pending downloads, bindings) are not reused
    }
  };
  return guarded([&]() -> int {
    int r;
    try {
      r = body();
    } catch (...) {
      leave(true);
      throw;
    }
    leave(r != SS_OK);
    return r;
  });
}

// Contiguous column shards of [c_begin, c_end) balanced by the number of layers (+ 1 per column: tiles without
// layers still cost an upload and a kernel lane); shard k = [cut[k], cut[k + 1])
std::vector<int> shard_cuts(const ss_canopy* can, int c_begin, int c_end, int nshards) {
  std::vector<int> cut(nshards + 1, c_end);
  cut[0] = c_begin;
  double total = 0.0;
  for (int j = c_begin; j < c_end; ++j) total += 1.0 + std::max(0, can->nlay[j]);
  double acc = 0.0;
  int k = 1;
  for (int j = c_begin; j < c_end && k < nshards; ++j) {
    acc += 1.0 + std::max(0, can->nlay[j]);
    while (k < nshards && acc >= total * k / nshards) cut[k++] = j + 1;
  }
  return cut;
}

}  // namespace

extern "C" {

int ss_device_count(int32_t* count) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  if (count) *count = n;
  return n > 0 ? SS_OK : SS_ERR_NO_DEVICE;
}

// Makes "device" current for plans created from now on.  Existing plans stay on
// the device they were created on (each owns its streams); the host entry's cached
// sub-plans are dropped when the device they live on is no longer the current one
// at their next use.
int ss_set_device(int32_t device) {
  int rc = device_ready();
  if (rc) return rc;
  if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return SS_ERR_HIP;
  return SS_OK;
}

const char* ss_last_error(void) { return g_last_error.c_str(); }
const char* ss_version(void) { return "spartacus_surface_hip 0.3 (gfx950)"; }
#ifndef SS_BUILD_ID
#define SS_BUILD_ID "unstamped"
#endif
const char* ss_build_id(void) { return SS_BUILD_ID; }

// Devices of the host entry: large calls of spartacus_radsurf_hip cut [istartcol, iendcol] into one contiguous
// shard per listed device (balanced by layer count), one host thread per shard.  ndevices = 0: back to the
// calling thread's current device.  A device may be listed more than once.
int ss_set_devices(const int32_t* devices, int32_t ndevices) {
  int rc = device_ready();
  if (rc) return rc;
  if (ndevices < 0 || (ndevices > 0 && !devices)) return SS_ERR_BAD_ARGUMENT;
  int n = 0;
  if (!hip_ok(hipGetDeviceCount(&n), "hipGetDeviceCount")) return SS_ERR_HIP;
  for (int k = 0; k < ndevices; ++k)
    if (devices[k] < 0 || devices[k] >= n) {
      g_last_error = "ss_set_devices: no such device";
      return SS_ERR_BAD_ARGUMENT;
    }
  std::unique_lock<std::shared_mutex> entry_lock(g_entry_mutex);
  std::lock_guard<std::mutex> lock(g_device_mutex);
  g_devices.assign(devices, devices + ndevices);
  g_devices_set = true;
  return SS_OK;
}

int ss_get_devices(int32_t* devices, int32_t capacity, int32_t* ndevices) {
  if (!ndevices) return SS_ERR_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lock(g_device_mutex);
  devices_from_env();
  *ndevices = (int32_t)g_devices.size();
  for (int k = 0; devices && k < capacity && k < (int)g_devices.size(); ++k) devices[k] = g_devices[k];
  return SS_OK;
}

int ss_plan_create(ss_plan** plan, const ss_config* config, const ss_canopy* canopy,
                   const ss_sw_spectral* sw, const ss_lw_spectral* lw, int32_t istartcol,
                   int32_t iendcol, int32_t want_profiles) {
  if (!plan || !config || !canopy) return SS_ERR_BAD_ARGUMENT;
  int rc = device_ready();
  if (rc) return rc;
  return guarded([&]() {
    ss_plan* p = new ss_plan();
    int r;
    try {
      r = p->impl.create(config, canopy, sw, lw, istartcol, iendcol, want_profiles);
    } catch (...) {
      delete p;
      throw;
    }
    if (r) { g_last_error = p->impl.error; delete p; return r; }
    *plan = p;
    return (int)SS_OK;
  });
}

// Every plan entry makes the plan's own device current first.  A plan is not
// re-entrant: calls on ONE plan must come from one thread at a time (different
// plans are independent - they share no stream or event).
#define SS_PLAN_ENTRY(plan) \
  if (!(plan)) return SS_ERR_BAD_ARGUMENT; \
  (void)hipGetLastError()          /* an error the host left behind in this thread is not ours */

int ss_plan_upload(ss_plan* plan, const ss_canopy* canopy, const ss_sw_spectral* sw,
                   const ss_lw_spectral* lw) {
  if (!plan || !canopy) return SS_ERR_BAD_ARGUMENT;
  return guarded([&]() {
    plan->impl.be.activate();
    int r = plan->impl.upload(canopy, sw, lw);
    if (r) g_last_error = plan->impl.error;
    return r;
  });
}

int ss_plan_calc_simple_spectrum_lw(ss_plan* plan) {
  SS_PLAN_ENTRY(plan);
  return guarded([&]() {
    plan->impl.be.activate();
    int r = plan->impl.simple_spectrum_lw();
    if (r) g_last_error = plan->impl.error;
    return r;
  });
}

int ss_plan_solve(ss_plan* plan) {
  SS_PLAN_ENTRY(plan);
  return guarded([&]() {
    plan->impl.be.activate();
    int r = plan->impl.solve();
    if (r) g_last_error = plan->impl.error;
    return r;
  });
}

int ss_plan_synchronize(ss_plan* plan) {
  SS_PLAN_ENTRY(plan);
  return guarded([&]() { plan->impl.be.activate(); plan->impl.be.sync(); return (int)SS_OK; });
}

int ss_plan_download(ss_plan* plan, ss_bc_out* bc_out, ss_flux* sw_norm_dir, ss_flux* sw_norm_diff,
                     ss_flux* lw_internal, ss_flux* lw_norm) {
  SS_PLAN_ENTRY(plan);
  return guarded([&]() {
    plan->impl.be.activate();
    return plan->impl.download(bc_out, sw_norm_dir, sw_norm_diff, lw_internal, lw_norm);
  });
}

int ss_plan_last_kernel_ms(ss_plan* plan, const char* name, double* ms) {
  if (!plan || !name || !ms) return SS_ERR_BAD_ARGUMENT;
  return guarded([&]() {
    plan->impl.be.activate();
    plan->impl.be.sync();
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
  return guarded([&]() { plan->impl.be.activate(); *count = plan->impl.eigen_failures(); return (int)SS_OK; });
}

int ss_plan_scale_and_sum(ss_plan* plan, const double* top_flux_dn_sw, const double* top_flux_dn_direct_sw,
                          const double* top_flux_dn_lw) {
  SS_PLAN_ENTRY(plan);
  return guarded([&]() {
    plan->impl.be.activate();
    int rc = plan->impl.scale_and_sum(top_flux_dn_sw, top_flux_dn_direct_sw, top_flux_dn_lw);
    if (rc) g_last_error = plan->impl.error;
    return rc;
  });
}

int ss_plan_download_sums(ss_plan* plan, ss_flux* sw_flux, ss_flux* lw_flux) {
  SS_PLAN_ENTRY(plan);
  return guarded([&]() { plan->impl.be.activate(); return plan->impl.download_sums(sw_flux, lw_flux); });
}

int ss_plan_set_pipes(ss_plan* plan, int32_t npipes) {
  if (!plan || npipes < 1) return SS_ERR_BAD_ARGUMENT;
  return guarded([&]() { plan->impl.be.activate(); return plan->impl.set_pipes(npipes); });
}

int ss_plan_device_ptr(ss_plan* plan, const char* name, void** ptr, int64_t* nelem) {
  if (!plan || !name || !ptr) return SS_ERR_BAD_ARGUMENT;
  auto it = plan->impl.dev.find(name);
  if (it == plan->impl.dev.end()) { g_last_error = "no such device array"; return SS_ERR_BAD_ARGUMENT; }
  *ptr = it->second;
  if (nelem) *nelem = (int64_t)plan->impl.dev_elems[name];
  return SS_OK;
}

int ss_plan_device(ss_plan* plan, int32_t* device) {
  if (!plan || !device) return SS_ERR_BAD_ARGUMENT;
  *device = plan->impl.be.device;
  return SS_OK;
}

int ss_plan_destroy(ss_plan* plan) {
  if (!plan) return SS_OK;
  return guarded([&]() {
    plan->impl.be.activate();
    try { plan->impl.be.sync(); } catch (...) {}
    delete plan;
    return (int)SS_OK;
  });
}

// ---- one column set on several devices: a plan per contiguous shard ----
struct ss_multi_plan {
  struct Part { ss_plan* plan; int device, istartcol, iendcol; };
  std::vector<Part> parts;
};

int ss_plan_create_multi(ss_multi_plan** mplan, const ss_config* config, const ss_canopy* canopy,
                         const ss_sw_spectral* sw, const ss_lw_spectral* lw, int32_t istartcol, int32_t iendcol,
                         int32_t want_profiles, const int32_t* devices, int32_t ndevices) {
  if (!mplan || !config || !canopy || ndevices < 1 || !devices) return SS_ERR_BAD_ARGUMENT;
  int rc = device_ready();
  if (rc) return rc;
  if (istartcol < 1 || iendcol > canopy->ncol || iendcol < istartcol) {
    g_last_error = "column range out of bounds";
    return SS_ERR_BAD_ARGUMENT;
  }
  int prev = 0;
  if (!hip_ok(hipGetDevice(&prev), "hipGetDevice")) return SS_ERR_HIP;
  const int nparts = std::min<int>(ndevices, std::max(1, (iendcol - istartcol + 1) / 64));
  const std::vector<int> cut = shard_cuts(canopy, istartcol - 1, iendcol, nparts);
  ss_multi_plan* mp = new ss_multi_plan();
  int r = SS_OK;
  for (int k = 0; k < nparts && r == SS_OK; ++k) {
    if (cut[k + 1] <= cut[k]) continue;
    if (!hip_ok(hipSetDevice(devices[k]), "hipSetDevice")) { r = SS_ERR_HIP; break; }
    ss_plan* p = nullptr;
    r = ss_plan_create(&p, config, canopy, sw, lw, cut[k] + 1, cut[k + 1], want_profiles);
    if (r == SS_OK) mp->parts.push_back({p, devices[k], cut[k] + 1, cut[k + 1]});
  }
  (void)hipSetDevice(prev);
  if (r != SS_OK) {
    for (auto& q : mp->parts) ss_plan_destroy(q.plan);
    delete mp;
    return r;
  }
  *mplan = mp;
  return SS_OK;
}

int ss_multi_plan_count(ss_multi_plan* mplan, int32_t* nparts) {
  if (!mplan || !nparts) return SS_ERR_BAD_ARGUMENT;
  *nparts = (int32_t)mplan->parts.size();
  return SS_OK;
}

int ss_multi_plan_part(ss_multi_plan* mplan, int32_t k, ss_plan** plan, int32_t* device, int32_t* istartcol,
                       int32_t* iendcol) {
  if (!mplan || k < 0 || k >= (int32_t)mplan->parts.size()) return SS_ERR_BAD_ARGUMENT;
  const auto& q = mplan->parts[k];
  if (plan) *plan = q.plan;
  if (device) *device = q.device;
  if (istartcol) *istartcol = q.istartcol;
  if (iendcol) *iendcol = q.iendcol;
  return SS_OK;
}

// queues the solve of every part on its device and returns (like ss_plan_solve)
int ss_multi_plan_solve(ss_multi_plan* mplan) {
  if (!mplan) return SS_ERR_BAD_ARGUMENT;
  for (auto& q : mplan->parts) {
    const int r = ss_plan_solve(q.plan);
    if (r != SS_OK) return r;
  }
  return SS_OK;
}

int ss_multi_plan_synchronize(ss_multi_plan* mplan) {
  if (!mplan) return SS_ERR_BAD_ARGUMENT;
  int rc = SS_OK;
  for (auto& q : mplan->parts) {
    const int r = ss_plan_synchronize(q.plan);
    if (r != SS_OK && rc == SS_OK) rc = r;
  }
  return rc;
}

// every part writes its own column / layer slices of the caller's (full-size) objects
int ss_multi_plan_download(ss_multi_plan* mplan, ss_bc_out* bc_out, ss_flux* sw_norm_dir, ss_flux* sw_norm_diff,
                           ss_flux* lw_internal, ss_flux* lw_norm) {
  if (!mplan) return SS_ERR_BAD_ARGUMENT;
  for (auto& q : mplan->parts) {
    const int r = ss_plan_download(q.plan, bc_out, sw_norm_dir, sw_norm_diff, lw_internal, lw_norm);
    if (r != SS_OK) return r;
  }
  return SS_OK;
}

int ss_multi_plan_destroy(ss_multi_plan* mplan) {
  if (!mplan) return SS_OK;
  int rc = SS_OK;
  for (auto& q : mplan->parts) {
    const int r = ss_plan_destroy(q.plan);
    if (r != SS_OK && rc == SS_OK) rc = r;
  }
  delete mplan;
  return rc;
}

int ss_host_pin(const void* ptr, int64_t bytes) {
  if (!ptr || bytes <= 0) return SS_ERR_BAD_ARGUMENT;
  int rc = device_ready();
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(g_device_mutex);
  if (!g_pins.pin(ptr, (size_t)bytes, /*call_owned=*/false)) {
    g_last_error = "ss_host_pin: range overlaps another registration or cannot be page-locked";
    return SS_ERR_HIP;
  }
  return SS_OK;
}

// The caller's side of the contract of ss_host_pin: before freeing a pinned array.  Waits for the
// device first (a queued copy must not lose its registration).
int ss_host_unpin(const void* ptr) {
  if (!ptr) return SS_ERR_BAD_ARGUMENT;
  int rc = device_ready();
  if (rc) return rc;
  // no host-entry call - small ones included, which only share this lock - may be in flight while a
  // registration is dropped: a DMA queued from the range would lose it
  std::unique_lock<std::shared_mutex> entry_lock(g_entry_mutex);
  std::lock_guard<std::mutex> lock(g_device_mutex);
  if (!hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize")) return SS_ERR_HIP;
  if (g_pins.release(false, ptr) == 0) {
    g_last_error = "ss_host_unpin: not the base of a registration made by ss_host_pin";
    return SS_ERR_BAD_ARGUMENT;
  }
  return SS_OK;
}

int ss_host_unpin_all(void) {
  int rc = device_ready();
  if (rc) return rc;
  std::unique_lock<std::shared_mutex> entry_lock(g_entry_mutex);
  std::lock_guard<std::mutex> lock(g_device_mutex);
  if (!hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize")) return SS_ERR_HIP;
  g_pins.release(false);
  return SS_OK;
}

int ss_host_pinned_bytes(int64_t* bytes) {
  if (!bytes) return SS_ERR_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lock(g_device_mutex);
  *bytes = (int64_t)g_pins.pinned_bytes();
  return SS_OK;
}

int ss_host_entry_plan_creations(int64_t* count) {
  if (!count) return SS_ERR_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lock(g_device_mutex);
  *count = g_plan_creations;
  return SS_OK;
}

int spartacus_radsurf_hip(const ss_config* config, const ss_canopy* canopy, const ss_sw_spectral* sw,
                          const ss_lw_spectral* lw, ss_bc_out* bc_out, int32_t istartcol,
                          int32_t iendcol, ss_flux* sw_norm_dir, ss_flux* sw_norm_diff,
                          ss_flux* lw_internal, ss_flux* lw_norm) {
  if (!config || !canopy || !bc_out) return SS_ERR_BAD_ARGUMENT;
  int rc = device_ready();
  if (rc) return rc;
  if (istartcol < 1 || iendcol > canopy->ncol || iendcol < istartcol - 1) {
    g_last_error = "column range out of bounds";
    return SS_ERR_BAD_ARGUMENT;
  }
  if (iendcol < istartcol) return SS_OK;
  // The reference is called concurrently on disjoint column ranges (spartacus_surface_driver.F90:206-233).
  // Small calls - its default blocks of 16 columns - run side by side on the slots of g_small ...
  {
    const char* se = std::getenv("SPARTACUS_SMALL_CALL_COLUMNS");
    const int small_max = se ? std::atoi(se) : kSmallColumns;
    if (iendcol - istartcol + 1 <= small_max) {
      const int profiles = ((sw_norm_dir && sw_norm_dir->flux_dn_layer_top)
                            || (lw_internal && lw_internal->flux_dn_layer_top)) ? 1 : 0;
      int device = 0;
      if (!hip_ok(hipGetDevice(&device), "hipGetDevice")) return SS_ERR_HIP;
      ss_flux* fls[4] = {config->do_sw ? sw_norm_dir : nullptr, config->do_sw ? sw_norm_diff : nullptr,
                         config->do_lw ? lw_internal : nullptr, config->do_lw ? lw_norm : nullptr};
      for (;;) {
        {
          std::shared_lock<std::shared_mutex> shared(g_entry_mutex);
          if (g_small.matches(config, profiles, device)) {
            // a free slot (one that already holds a plan first: a sequential caller stays on one slot), else queue up
            SmallSlot* sl = nullptr;
            std::unique_lock<std::mutex> own;
            // (a slot's members are only read under its mutex: another thread may be building its plan)
            for (int pass = 0; pass < 2 && !sl; ++pass)
              for (int i = 0; i < SmallPool::kSlots && !sl; ++i) {
                std::unique_lock<std::mutex> t(g_small.slot[i].mu, std::try_to_lock);
                if (!t.owns_lock()) continue;
                if (pass == 0 && !g_small.slot[i].plan) continue;      // (t unlocks)
                sl = &g_small.slot[i];
                own = std::move(t);
              }
            if (!sl) {
              sl = &g_small.slot[g_small.next++ % SmallPool::kSlots];
              own = std::unique_lock<std::mutex>(sl->mu);
            }
            const int rcs = guarded([&]() -> int {
              try {
                return small_call(*sl, config, canopy, sw, lw, bc_out, istartcol - 1, iendcol, fls, profiles);
              } catch (...) {
                if (sl->plan) { try { sl->plan->be.sync(); } catch (...) {} }
                sl->drop();
                throw;
              }
            });
            if (rcs != SS_OK && sl->plan) { try { sl->plan->be.sync(); } catch (...) {} sl->drop(); }
            return rcs;
          }
        }
        // (re)configure the pool: exclusive
        std::unique_lock<std::shared_mutex> excl(g_entry_mutex);
        if (!g_small.matches(config, profiles, device)) {
          for (auto& sl : g_small.slot) { std::lock_guard<std::mutex> l(sl.mu); sl.drop(); }
          g_small.cfg = *config;
          g_small.profiles = profiles;
          g_small.device = device;
        }
      }
    }
  }
  // ... large calls have the device(s) to themselves.
  std::unique_lock<std::shared_mutex> entry_lock(g_entry_mutex);
  std::lock_guard<std::mutex> lock(g_device_mutex);
  const int profiles = ((sw_norm_dir && sw_norm_dir->flux_dn_layer_top)
                        || (lw_internal && lw_internal->flux_dn_layer_top)) ? 1 : 0;
  ss_flux* fl[4] = {config->do_sw ? sw_norm_dir : nullptr, config->do_sw ? sw_norm_diff : nullptr,
                    config->do_lw ? lw_internal : nullptr, config->do_lw ? lw_norm : nullptr};
  const int c_begin = istartcol - 1, c_end = iendcol;
  int device = 0;
  if (!hip_ok(hipGetDevice(&device), "hipGetDevice")) return SS_ERR_HIP;
  devices_from_env();
  // shards: one per configured device (not more than there are blocks of 64 columns), else the current device
  std::vector<int> devs = g_devices;
  if (devs.empty()) devs.push_back(device);
  // (SPARTACUS_SHARD_MIN_COLUMNS: smallest shard worth a host thread and a pipeline of its own, default 64 columns;
  // the tests lower it to send the reference's small inputs through the fan-out)
  const char* sme = std::getenv("SPARTACUS_SHARD_MIN_COLUMNS");
  const int shard_min = std::max(1, sme ? std::atoi(sme) : 64);
  const int max_shards = std::max(1, (c_end - c_begin) / shard_min);
  if ((int)devs.size() > max_shards) devs.resize(max_shards);
  // SPARTACUS_PIN_HOST: registrations owned by this call (PinRegistry), made once for all shards and released
  // when every shard has returned.  Call-owned locks are the DEFAULT for calls of 32768 columns and more
  // (measured on this box: locking and releasing ~1 GB per call costs ~1 ms from the second call on, the pipeline
  // it enables takes 27 -> 17 ms off a 200 000-column call); SPARTACUS_PIN_HOST=1 lowers the threshold to
  // SPARTACUS_PIN_HOST_MIN_COLUMNS (default 4096), SPARTACUS_PIN_HOST=0 switches locking off.
  bool pins_made = false;
  {
    const char* pe = std::getenv("SPARTACUS_PIN_HOST");
    const char* me = std::getenv("SPARTACUS_PIN_HOST_MIN_COLUMNS");
    const bool off = pe && pe[0] == '0';
    const int min_cols = me ? std::atoi(me) : (pe ? 4096 : 32768);
    if (!off && c_end - c_begin >= min_cols) {
      long call_l0 = 0, call_l1 = 0;
      layer_range(canopy, c_begin, c_end, &call_l0, &call_l1);
      pins_made = true;
      pin_call_arrays(config, canopy, sw, lw, bc_out, fl, c_begin, c_end, call_l0, call_l1);
    }
  }
  int result = SS_OK;
  if (devs.size() == 1) {
    if (devs[0] != device && !hip_ok(hipSetDevice(devs[0]), "hipSetDevice")) result = SS_ERR_HIP;
    if (result == SS_OK)
      result = large_call_shard(host_entry(0), devs[0], config, canopy, sw, lw, bc_out, c_begin, c_end, fl, profiles);
    if (devs[0] != device) (void)hipSetDevice(device);
  } else {
    // one host thread per shard: its own device, its own HostEntry (sub-plans, streams, call-level arrays), its
    // own contiguous slices of the caller's arrays - uploads and downloads of different devices share nothing
    // but the host's memory bandwidth, and no result ever crosses from one device to another
    const std::vector<int> cut = shard_cuts(canopy, c_begin, c_end, (int)devs.size());
    for (size_t k = 0; k < devs.size(); ++k) (void)host_entry(k);          // (the vector must not grow under the threads)
    std::vector<int> rcs(devs.size(), SS_OK);
    std::vector<std::string> errs(devs.size());
    std::vector<std::thread> workers;
    for (size_t k = 0; k < devs.size(); ++k) {
      workers.emplace_back([&, k]() {
        if (cut[k + 1] <= cut[k]) return;
        if (hipSetDevice(devs[k]) != hipSuccess) {
          rcs[k] = SS_ERR_HIP;
          errs[k] = "hipSetDevice failed for a device of the host entry";
          return;
        }
        rcs[k] = large_call_shard(*g_hosts[k], devs[k], config, canopy, sw, lw, bc_out, cut[k], cut[k + 1], fl, profiles);
        if (rcs[k] != SS_OK) errs[k] = g_last_error;      // (thread-local in the worker)
      });
    }
    for (auto& w : workers) w.join();
    for (size_t k = 0; k < devs.size(); ++k)
      if (rcs[k] != SS_OK && result == SS_OK) {
        result = rcs[k];
        g_last_error = "shard " + std::to_string(k) + " (device " + std::to_string(devs[k]) + "): " + errs[k];
      }
  }
  if (pins_made) g_pins.release(/*call_owned_only=*/true);
  return result;
}

}  // extern "C"
