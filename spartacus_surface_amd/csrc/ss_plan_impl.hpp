// ss_plan_impl.hpp - host-side plan: column classification, HBM layout, chunked
// launches.  Templated on a Backend so that the identical host logic drives
//   * HipBackend  (ss_capi.hip)      - the product: HIP kernels on gfx950
//   * EmulBackend (tests/emul)       - TEST-ONLY host loops over the same
//                                      per-unit bodies (sanitizers, gdb)
//
// Column classes.  radsurf() dispatches per column on the tile type
// (radsurf_interface.F90:105-311).  Here the columns of the requested range are
// sorted once into classes (tile kind, number of regions, number of streams);
// every class is one template instance of the solver kernels, so a wavefront
// never mixes matrix orders.  Within a class columns keep their original order
// (stable) sorted by descending layer count, so that lanes of a wave have equal
// trip counts and idle layer-units fall into whole waves that exit at once.
//
// HBM layout.  Inputs and outputs are byte-for-byte mirrors of the host
// (Fortran) arrays, (nspec, ncol) / (nspec, ntotlay) with the band index
// fastest.  The per-layer matrices live in two scratch buffers
// [tile of 64 column-units][layer][element pair][lane][2] (one contiguous block
// per wavefront, 16 bytes per lane and access, ss_layer_matrices.hpp);
// column-unit = (column of the chunk, band) with
// the band fastest.  A class is processed in chunks of columns so that the
// scratch stays bounded (default 4 GiB, env SPARTACUS_SCRATCH_MB).
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/spartacus_surface_hip.h"
#include "ss_common.hpp"
#include "ss_pack.hpp"
#include "ss_layer_matrices.hpp"
#include "ss_eigen_coop.hpp"
#include "ss_misc_units.hpp"
#include "ss_column_lw.hpp"
#include "ss_column_sw.hpp"
#include "ss_column_sw_sym.hpp"
#include "ss_column_lw_sym.hpp"
#include "ss_solver_lw.hpp"
#include "ss_solver_sw.hpp"

namespace ss {

struct ClassKey {
  bool urban;
  int nreg;
  int ns;
  bool operator<(const ClassKey& o) const {
    if (urban != o.urban) return urban < o.urban;
    if (nreg != o.nreg) return nreg < o.nreg;
    return ns < o.ns;
  }
};

struct ColumnClass {
  ClassKey key;
  std::vector<int32_t> cols;   // 0-based, sorted by nlay descending (stable)
  size_t dev_offset = 0;       // offset of cols in the device index buffer
  int maxlay = 0;
  double* d_pack = nullptr;    // per-layer inputs in record tiling (ss_pack.hpp), whole class
  size_t pack_elems = 0;
};

inline bool key_supported(const ClassKey& k) {
  const bool ns_ok = (k.ns >= 1 && k.ns <= 8);
  const bool nreg_ok = k.urban ? (k.nreg >= 1 && k.nreg <= 3) : (k.nreg == 2 || k.nreg == 3);
  return ns_ok && nreg_ok;
}

// Number of scratch elements per layer of a class
// (orders of the fused layer + upward kernel have no layer record, and their column record
// also carries what the downward sweep used to read from it: fused_order, ss_layer_matrices.hpp)
inline int lay_nelem(bool lw, const ClassKey& k) {
  const int n = k.nreg * k.ns, d = k.nreg;
  if (fused_order(n, lw)) return 0;
  auto ev = [](int x) { return (x + 1) & ~1; };   // blocks start at even elements (16-byte pairs)
  // = LwLayRec<n>::NELEM / SwLayRec<n, d>::NELEM (ss_layer_matrices.hpp): R^ and T^ packed for the small orders
  const int rt = sym_packed(n, lw) ? tri_count(n) : n * n;
  return lw ? 2 * ev(rt) + ev(n * n) + 2 * ev(n) : 2 * ev(rt) + 2 * ev(n * d) + 2 * ev(d * d);
}
inline int col_nelem(bool lw, const ClassKey& k) {
  const int n = k.nreg * k.ns, d = k.nreg;
  // = LwColRec<n>::NELEM / SwColRec<n,d>::NELEM (ss_column_lw.hpp, ss_column_sw.hpp)
  auto ev = [](int x) { return (x + 1) & ~1; };
  // a_below is packed (scaled, symmetric) for the small orders
  const int ab = sym_packed_col(n, lw) ? tri_count(n) : n * n;
  // (shortwave, small orders: Q and d_below applied to the direct flux at the layer top, + that flux: sw_dirvec)
  const int base = lw ? ev(n * n) + ev(ab) + 2 * ev(n)
                      : sw_dirvec(n) ? ev(n * n) + ev(ab) + 2 * ev(n) + ev(d) : ev(n * n) + ev(ab) + 2 * ev(n * d);
  if (!fused_order(n, lw)) return base;
#if defined(SS_FUSE_VIA_RECORD)
  return base + (lw ? ev(n * n) + ev(n) : 2 * ev(d * d) + 2 * ev(n * d));
#else
  // (shortwave fused with the direct beam as a vector: the direct flux below the top interface has d + 1 components)
  if (!lw && sw_dirvec(n)) return base - ev(d) + ev(d + 1) + 2 * ev(d * d);
  return base + (lw ? ev(n * n) + ev(n) : 2 * ev(d * d));
#endif
}

// eigen record of the cooperative eigen-solver (= EigRec<n>::NELEM, ss_eigen_coop.hpp); 0 = order not cooperative
inline int eig_nelem(const ClassKey& k) {
  const int n = k.nreg * k.ns;
  return coop_order(n) ? eig_rec_nelem(n) : 0;
}

// Runtime (urban, nreg, ns) -> template instance
#define SS_DISPATCH_NS(URB, NREG, ns, CALL)                 \
  switch (ns) {                                             \
    case 1: { constexpr int NS_ = 1; CALL(URB, NREG, NS_); } break; \
    case 2: { constexpr int NS_ = 2; CALL(URB, NREG, NS_); } break; \
    case 3: { constexpr int NS_ = 3; CALL(URB, NREG, NS_); } break; \
    case 4: { constexpr int NS_ = 4; CALL(URB, NREG, NS_); } break; \
    case 5: { constexpr int NS_ = 5; CALL(URB, NREG, NS_); } break; \
    case 6: { constexpr int NS_ = 6; CALL(URB, NREG, NS_); } break; \
    case 7: { constexpr int NS_ = 7; CALL(URB, NREG, NS_); } break; \
    case 8: { constexpr int NS_ = 8; CALL(URB, NREG, NS_); } break; \
    default: break;                                         \
  }
#define SS_DISPATCH(key, CALL)                                         \
  do {                                                                 \
    if ((key).urban) {                                                 \
      if ((key).nreg == 1) { SS_DISPATCH_NS(true, 1, (key).ns, CALL) } \
      else if ((key).nreg == 2) { SS_DISPATCH_NS(true, 2, (key).ns, CALL) } \
      else if ((key).nreg == 3) { SS_DISPATCH_NS(true, 3, (key).ns, CALL) } \
    } else {                                                           \
      if ((key).nreg == 2) { SS_DISPATCH_NS(false, 2, (key).ns, CALL) } \
      else if ((key).nreg == 3) { SS_DISPATCH_NS(false, 3, (key).ns, CALL) } \
    }                                                                  \
  } while (0)

// which canopy_flux members exist (allocate_canopy_flux, radsurf_canopy_flux.F90:96-163)
inline bool flux_member_present_for(const ss_config& cfg, bool want_profiles, int F, bool use_direct) {
  switch (F) {
    case F_GROUND_DN_DIR: case F_TOP_DN_DIR: case F_GROUND_SUNLIT_FRAC:
      return use_direct;
    case F_ROOF_IN: case F_ROOF_NET: case F_WALL_IN: case F_WALL_NET:
      return cfg.do_urban != 0;
    case F_ROOF_IN_DIR: case F_WALL_IN_DIR: case F_ROOF_SUNLIT_FRAC: case F_WALL_SUNLIT_FRAC:
      return cfg.do_urban != 0 && use_direct;
    case F_VEG_ABS: case F_VEG_AIR_ABS:
      return cfg.do_vegetation != 0;
    case F_VEG_ABS_DIR: case F_VEG_SUNLIT_FRAC:
      return cfg.do_vegetation != 0 && use_direct;
    case F_FLUX_DN_LAYER_TOP: case F_FLUX_UP_LAYER_TOP: case F_FLUX_DN_LAYER_BASE:
    case F_FLUX_UP_LAYER_BASE:
      return want_profiles;
    case F_FLUX_DN_DIR_LAYER_TOP: case F_FLUX_DN_DIR_LAYER_BASE:
      return want_profiles && use_direct;
    default:
      return true;
  }
}

template <class Backend>
class Plan {
 public:
  ss_config cfg;
  int ncol = 0, ntotlay = 0, nsw = 0, nlw = 0;
  int istartcol = 1, iendcol = 0;
  int lay_first = 0, lay_count = 0;   // packed layer range of the column range (0-based)
  bool want_profiles = true;

  // host copies of the small integer arrays (classification, slices)
  std::vector<int32_t> h_nlay, h_istartlay, h_irep;
  std::vector<double> h_cos_sza;

  // device mirrors by name
  std::map<std::string, void*> dev;
  std::map<std::string, size_t> dev_elems, dev_cap;

  std::vector<ColumnClass> sw_classes, lw_classes;
  std::vector<int32_t> flat_cols, night_cols, lwzero_urban_cols;
  std::vector<int32_t> simple_cols[2];   // [0] exponential model (type 4), [1] infinite street (type 5)
  std::vector<int32_t> h_indcol;
  int32_t* d_index = nullptr;
  size_t flat_off = 0, night_off = 0, simple_off[2] = {0, 0};
  size_t index_elems = 0;

  // Scratch of the layer / adding-method records, one set per "pipe": successive
  // chunks (and the SW and LW families) alternate between the pipes, which the
  // backend maps to concurrent streams, so that the memory-bound column kernels
  // of one chunk overlap with the arithmetic-bound layer kernel of the next.
  static constexpr int kMaxPipes = 8;
  int npipes = 1;
  double* d_lay[kMaxPipes] = {};
  double* d_col[kMaxPipes] = {};
  double* d_eig[kMaxPipes] = {};
  size_t coop_list_bytes = 0;      // tail of every eigen buffer: the dense list of cooperative unit-layers
  bool use_coop_list = true;       // env SPARTACUS_NO_COOP_LIST=1: the kernels find their units by position
  size_t scratch_budget_bytes = 0;
  static constexpr int kDefaultPipes = 4;
  int pipes_wanted = kDefaultPipes;
  int chunks_per_pipe = 4;
  int pipes_cap = kMaxPipes;     // set before create(): upper limit for this plan
  size_t lay_bytes = 0, col_bytes = 0, eig_bytes = 0;
  // slot 0: eigen failures of the last solve; the rest: section stamps of profiling builds
  static constexpr int kCounters = 32;
  unsigned long long* d_eigen_fail = nullptr;

  Backend be;                    // device context of this plan: its own streams and events
  typename Backend::Timers timers;
  std::string error;

  ~Plan() { release(); }

  void release() {
    for (auto& kv : dev)
      if (!dev_external.count(kv.first) || !dev_external[kv.first]) be.free(kv.second);
    dev_external.clear();
    dev.clear();
    dev_elems.clear();
    dev_cap.clear();
    if (d_index) be.free(d_index);
    free_packs(sw_classes);
    free_packs(lw_classes);
    for (int p = 0; p < kMaxPipes; ++p) {
      if (d_lay[p]) be.free(d_lay[p]);
      if (d_col[p]) be.free(d_col[p]);
      if (d_eig[p]) be.free(d_eig[p]);
      d_lay[p] = d_col[p] = d_eig[p] = nullptr;
    }
    lay_bytes = col_bytes = eig_bytes = 0;
    coop_list_bytes = 0;
    if (d_eigen_fail) be.free(d_eigen_fail);
    d_index = nullptr; d_eigen_fail = nullptr;
    timers.release();
    be.release();
  }

  void free_packs(std::vector<ColumnClass>& cls) {
    for (auto& c : cls) {
      if (c.d_pack) be.free(c.d_pack);
      c.d_pack = nullptr;
      c.pack_elems = 0;
    }
  }

  // (Re)build the input packs of all classes from the device mirrors of the inputs
  bool packs_valid = false;
  int pack_inputs() {
    for (int lw = 0; lw < 2; ++lw) {
      auto& classes = lw ? lw_classes : sw_classes;
      const int nspec = lw ? nlw : nsw;
      for (auto& c : classes) {
        int rc = check_inputs(c.key, lw != 0);
        if (rc) return rc;
        const size_t units = c.cols.size() * (size_t)nspec;
        const size_t tiles = (units + 63) / 64;
        const size_t need = tiles * 64 * (size_t)c.maxlay * (size_t)(lw ? kPackLw : kPackSw);
        if (!c.d_pack || c.pack_elems < need) {
          if (c.d_pack) be.free(c.d_pack);
          c.d_pack = static_cast<double*>(be.alloc(need * sizeof(double)));
          c.pack_elems = need;
        }
        PackLaunch P;
        P.in = make_inputs(lw != 0);
        P.cols = d_index + c.dev_offset;
        P.ncu = (int)units;
        P.nspec = nspec;
        P.maxlay = c.maxlay;
        P.lw = lw;
        P.urban = c.key.urban ? 1 : 0;
        P.nreg = c.key.nreg;
        P.pack = c.d_pack;
        be.launch_pack(P, timers);
      }
    }
    packs_valid = true;
    return SS_OK;
  }

  // ---- device mirrors ----
  double* darr(const char* name) const {
    auto it = dev.find(name);
    return it == dev.end() ? nullptr : static_cast<double*>(it->second);
  }
  // Device arrays only grow: a plan that is re-targeted to another column set
  // (reset(), the host entry's chunk loop and block calls) keeps its allocations;
  // dev_elems is the logical size, dev_cap the allocated one (both in elements).
  // An array may live outside the plan (bind_external: the host entry keeps the per-column
  // outputs of a whole call in call-level arrays, so that they come back in one copy per member
  // instead of one per member and chunk): the plan then neither sizes nor frees it.
  std::map<std::string, bool> dev_external;
  void bind_external(const std::string& name, void* p, size_t elems) {
    auto it = dev.find(name);
    if (it != dev.end() && !dev_external[name]) be.free(it->second);
    dev[name] = p;
    dev_elems[name] = elems;
    dev_cap[name] = elems;
    dev_external[name] = true;
  }
  // forget every external binding (the owner keeps the memory): the next ensure() of such a name
  // allocates an array of the plan's own again.  The host entry does this at the start of every
  // chunk, so that a binding never outlives the call it was made for.
  void unbind_externals() {
    for (auto& kv : dev_external)
      if (kv.second) { dev.erase(kv.first); dev_elems.erase(kv.first); dev_cap.erase(kv.first); }
    dev_external.clear();
  }
  void ensure(const std::string& name, size_t elems, size_t elsize = sizeof(double)) {
    auto ex = dev_external.find(name);
    if (ex != dev_external.end() && ex->second) return;
    auto it = dev.find(name);
    if (it != dev.end() && dev_cap[name] >= elems) { dev_elems[name] = elems; return; }
    if (it != dev.end()) be.free(it->second);
    void* p = be.alloc(elems * elsize);
    be.memset(p, 0, elems * elsize);
    dev[name] = p;
    dev_elems[name] = elems;
    dev_cap[name] = elems;
  }
  void mirror(const std::string& name, const void* host, size_t elems, size_t elsize = sizeof(double),
              bool staged = false) {
    if (!host) {
      auto it = dev.find(name);
      if (it != dev.end()) { be.free(it->second); dev.erase(it); dev_elems.erase(name); dev_cap.erase(name); }
      return;
    }
    ensure(name, elems, elsize);
    if (staged) be.h2d_staged(dev[name], host, elems * elsize);
    else be.h2d(dev[name], host, elems * elsize);
  }

  // which canopy_flux members exist (allocate_canopy_flux, radsurf_canopy_flux.F90:96-163)
  bool flux_member_present(int F, bool use_direct) const {
    return flux_member_present_for(cfg, want_profiles, F, use_direct);
  }
  static const char* flux_field_name(int F) {
    static const char* names[F_NFIELDS] = {
        "ground_dn", "ground_dn_dir", "ground_net", "top_dn", "top_dn_dir", "top_net",
        "roof_in", "roof_in_dir", "roof_net", "wall_in", "wall_in_dir", "wall_net",
        "roof_sunlit_frac", "wall_sunlit_frac", "ground_sunlit_frac", "ground_vertical_diff",
        "clear_air_abs", "veg_abs", "veg_air_abs", "veg_abs_dir", "veg_sunlit_frac",
        "flux_dn_layer_top", "flux_dn_dir_layer_top", "flux_up_layer_top",
        "flux_dn_layer_base", "flux_dn_dir_layer_base", "flux_up_layer_base"};
    return names[F];
  }
  size_t flux_field_elems(int F, int nspec) const {
    switch (flux_field_shape(F)) {
      case 0: return (size_t)nspec * ncol;
      case 1: return (size_t)nspec * ntotlay;
      case 2: return (size_t)ncol;
      default: return (size_t)ntotlay;
    }
  }
  FluxDev flux_dev(const char* obj) const {
    FluxDev f;
    for (int F = 0; F < F_NFIELDS; ++F)
      f.f[F] = darr((std::string(obj) + "." + flux_field_name(F)).c_str());
    return f;
  }
  void alloc_flux(const char* obj, int nspec, bool use_direct) {
    for (int F = 0; F < F_NFIELDS; ++F)
      if (flux_member_present(F, use_direct))
        ensure(std::string(obj) + "." + flux_field_name(F), std::max<size_t>(1, flux_field_elems(F, nspec)));
  }

  // ---- creation ----
  int create(const ss_config* c, const ss_canopy* can, const ss_sw_spectral* sw,
             const ss_lw_spectral* lw, int i0, int i1, int profiles) {
    cfg = *c;
    nsw = c->nsw;
    nlw = c->nlw;
    want_profiles = profiles != 0;
    const char* mb = std::getenv("SPARTACUS_SCRATCH_MB");
    scratch_budget_bytes = (mb ? (size_t)std::atol(mb) : (size_t)4096) << 20;
    if (const char* ncl = std::getenv("SPARTACUS_NO_COOP_LIST")) use_coop_list = !(ncl[0] == '1');
    if (const char* mcu = std::getenv("SPARTACUS_MIN_CHUNK_UNITS")) min_chunk_units = (size_t)std::max(64L, std::atol(mcu));
    const char* cpp = std::getenv("SPARTACUS_CHUNKS_PER_PIPE");
    chunks_per_pipe = cpp ? std::max(0, std::atoi(cpp)) : 4;
    const char* np = std::getenv("SPARTACUS_PIPES");
    pipes_wanted = std::min(pipes_cap, np ? std::max(1, std::atoi(np)) : kDefaultPipes);
    be.init();
    d_eigen_fail = static_cast<unsigned long long*>(be.alloc(kCounters * sizeof(unsigned long long)));
    be.memset(d_eigen_fail, 0, kCounters * sizeof(unsigned long long));
    return reset(can, sw, lw, i0, i1);
  }

  // Point the plan at a column set (same configuration, any shape): output arrays
  // for (ncol, ntotlay), inputs uploaded, columns of [i0, i1] classified.
  int reset(const ss_canopy* can, const ss_sw_spectral* sw, const ss_lw_spectral* lw, int i0, int i1,
            bool wait = true) {
    ncol = can->ncol;
    ntotlay = can->ntotlay;
    istartcol = i0;
    iendcol = i1;
    if (i0 < 1 || i1 > ncol || i1 < i0 - 1) { error = "column range out of bounds"; return SS_ERR_BAD_ARGUMENT; }
    if (cfg.do_sw && (!sw || sw->nspec != nsw)) { error = "sw spectral properties missing / nspec mismatch"; return SS_ERR_BAD_ARGUMENT; }
    if (cfg.do_lw && (!lw || lw->nspec != nlw)) { error = "lw spectral properties missing / nspec mismatch"; return SS_ERR_BAD_ARGUMENT; }
    ensure("bc_out.sw_albedo", std::max<size_t>(1, (size_t)nsw * ncol));
    ensure("bc_out.sw_albedo_dir", std::max<size_t>(1, (size_t)nsw * ncol));
    ensure("bc_out.lw_emissivity", std::max<size_t>(1, (size_t)nlw * ncol));
    ensure("bc_out.lw_emission", std::max<size_t>(1, (size_t)nlw * ncol));
    if (cfg.do_sw) { alloc_flux("sw_norm_dir", nsw, true); alloc_flux("sw_norm_diff", nsw, true); }
    if (cfg.do_lw) { alloc_flux("lw_internal", nlw, false); alloc_flux("lw_norm", nlw, false); }
    return upload(can, sw, lw, wait);
  }

  // ---- upload inputs + classify ----
  // wait = false: the copies stay queued on the plan's stream (host arrays that are pinned are
  // read by the DMA engine later: the caller keeps them unchanged until it has synchronised -
  // the host entry does, its arrays are the caller's for the whole call)
  int upload(const ss_canopy* can, const ss_sw_spectral* sw, const ss_lw_spectral* lw, bool wait = true) {
    if (can->ncol != ncol || can->ntotlay != ntotlay) { error = "shape changed since plan creation"; return SS_ERR_BAD_ARGUMENT; }
    const bool timing_ = std::getenv("SPARTACUS_HOST_TIMING") != nullptr;
    auto now_ = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0_ = now_();
    h_nlay.assign(can->nlay, can->nlay + ncol);
    h_indcol.clear();
    h_istartlay.assign(can->istartlay, can->istartlay + ncol);
    h_irep.assign(can->i_representation, can->i_representation + ncol);
    if (can->cos_sza) h_cos_sza.assign(can->cos_sza, can->cos_sza + ncol);
    else h_cos_sza.assign(ncol, 0.0);

    be.arena_reset();
    mirror("canopy.nlay", can->nlay, ncol, sizeof(int32_t), true);
    mirror("canopy.istartlay", can->istartlay, ncol, sizeof(int32_t), true);
    mirror("canopy.cos_sza", can->cos_sza, ncol);
    const size_t nl = std::max(1, ntotlay);
#define MIRL(obj, member) mirror(#obj "." #member, (obj)->member, (obj)->member ? nl : 0)
    mirror("canopy.dz", can->dz, nl);
    mirror("canopy.building_fraction", can->building_fraction, nl);
    mirror("canopy.veg_fraction", can->veg_fraction, nl);
    mirror("canopy.building_scale", can->building_scale, nl);
    mirror("canopy.veg_scale", can->veg_scale, nl);
    mirror("canopy.veg_ext", can->veg_ext, nl);
    mirror("canopy.veg_fsd", can->veg_fsd, nl);
    mirror("canopy.veg_contact_fraction", can->veg_contact_fraction, nl);
    mirror("canopy.ground_temperature", can->ground_temperature, ncol);
    mirror("canopy.roof_temperature", can->roof_temperature, nl);
    mirror("canopy.wall_temperature", can->wall_temperature, nl);
    mirror("canopy.clear_air_temperature", can->clear_air_temperature, nl);
    mirror("canopy.veg_temperature", can->veg_temperature, nl);
    mirror("canopy.veg_air_temperature", can->veg_air_temperature, nl);
#undef MIRL
    if (cfg.do_sw) {
      const size_t sl = (size_t)nsw * nl, sc = (size_t)nsw * ncol;
      mirror("sw.air_ext", sw->air_ext, sl);
      mirror("sw.air_ssa", sw->air_ssa, sl);
      mirror("sw.veg_ssa", sw->veg_ssa, sl);
      mirror("sw.ground_albedo", sw->ground_albedo, sc);
      mirror("sw.roof_albedo", sw->roof_albedo, sl);
      mirror("sw.wall_albedo", sw->wall_albedo, sl);
      mirror("sw.ground_albedo_dir", sw->ground_albedo_dir, sc);
      mirror("sw.roof_albedo_dir", sw->roof_albedo_dir, sl);
      mirror("sw.wall_specular_frac", sw->wall_specular_frac, sl);
    }
    if (cfg.do_lw) {
      const size_t ll = (size_t)nlw * nl, lc = (size_t)nlw * ncol;
      mirror("lw.air_ext", lw->air_ext, ll);
      mirror("lw.air_ssa", lw->air_ssa, ll);
      mirror("lw.veg_ssa", lw->veg_ssa, ll);
      mirror("lw.clear_air_planck", lw->clear_air_planck, ll);
      mirror("lw.veg_planck", lw->veg_planck, ll);
      mirror("lw.veg_air_planck", lw->veg_air_planck, ll);
      mirror("lw.ground_emissivity", lw->ground_emissivity, lc);
      mirror("lw.roof_emissivity", lw->roof_emissivity, ll);
      mirror("lw.wall_emissivity", lw->wall_emissivity, ll);
      mirror("lw.ground_emission", lw->ground_emission, lc);
      mirror("lw.roof_emission", lw->roof_emission, ll);
      mirror("lw.wall_emission", lw->wall_emission, ll);
    }
    const double t1_ = now_();
    const int rc = classify();
    if (timing_) std::fprintf(stderr, "  upload: mirrors %.2f ms, classify %.2f ms\n", t1_ - t0_, now_() - t1_);
    if (wait) be.sync();     // the caller may reuse (or free) its host arrays as soon as this returns
    return rc;
  }

  int classify() {
    // the pack buffers survive a re-classification with the same classes (a host
    // that calls every time step): park them by class key, take them back below
    std::map<std::pair<int, ClassKey>, std::pair<double*, size_t>> pool;
    for (int lw = 0; lw < 2; ++lw)
      for (auto& c : (lw ? lw_classes : sw_classes))
        if (c.d_pack) { pool[{lw, c.key}] = {c.d_pack, c.pack_elems}; c.d_pack = nullptr; }
    packs_valid = false;
    sw_classes.clear(); lw_classes.clear();
    flat_cols.clear(); night_cols.clear(); simple_cols[0].clear(); simple_cols[1].clear();
    std::map<ClassKey, std::vector<int32_t>> swm, lwm;
    // neighbouring columns are mostly of one class: remember the last bucket of each family
    struct Last { ClassKey k{false, -1, -1}; std::vector<int32_t>* v = nullptr; } last_sw, last_lw;
    auto bucket = [](std::map<ClassKey, std::vector<int32_t>>& m, Last& l, const ClassKey& k) -> std::vector<int32_t>& {
      if (!l.v || l.k < k || k < l.k) { l.v = &m[k]; l.k = k; }
      return *l.v;
    };
    lay_first = 0; lay_count = 0;
    for (int j = istartcol - 1; j < iendcol; ++j) {
      const int irep = h_irep[j];
      if (irep == SS_TILE_FLAT) { flat_cols.push_back(j); continue; }
      if (irep == SS_TILE_SIMPLE_URBAN || irep == SS_TILE_INFINITE_STREET) {
        // radsurf_interface.F90:271-308: one layer only; solved by the misc kernel family
        if (h_nlay[j] > 1) { error = "simple urban representations must have only one layer"; return SS_ERR_SIMPLE_URBAN_LAYERS; }
        if (h_nlay[j] < 1) { error = "canopy column without layers"; return SS_ERR_BAD_ARGUMENT; }
        simple_cols[irep == SS_TILE_INFINITE_STREET].push_back(j);
        continue;
      }
      if (irep < 0 || irep > 5) { error = "bad surface type"; return SS_ERR_BAD_ARGUMENT; }
      const bool urban = irep != SS_TILE_FOREST;
      const int nreg = (irep == SS_TILE_FOREST) ? cfg.n_vegetation_region_forest + 1
                       : (irep == SS_TILE_URBAN) ? 1 : cfg.n_vegetation_region_urban + 1;
      if (urban && nreg > 1 && !cfg.do_vegetation) { error = "urban tile with >1 region but do_vegetation off"; return SS_ERR_VEG_DISABLED; }
      if (h_nlay[j] < 1) { error = "canopy column without layers"; return SS_ERR_BAD_ARGUMENT; }
      if (cfg.do_sw) {
        if (h_cos_sza[j] > 0.0) {
          ClassKey k{urban, nreg, urban ? cfg.lg_sw_urban.nstream : cfg.lg_sw_forest.nstream};
          if (!key_supported(k)) { error = "SW (nstream,nregion) combination not instantiated"; return SS_ERR_UNSUPPORTED; }
          bucket(swm, last_sw, k).push_back(j);
        } else {
          night_cols.push_back(j);
        }
      }
      if (cfg.do_lw) {
        ClassKey k{urban, nreg, urban ? cfg.lg_lw_urban.nstream : cfg.lg_lw_forest.nstream};
        if (!key_supported(k)) { error = "LW (nstream,nregion) combination not instantiated"; return SS_ERR_UNSUPPORTED; }
        bucket(lwm, last_lw, k).push_back(j);
      }
    }
    if (iendcol >= istartcol) {
      lay_first = h_istartlay[istartcol - 1] - 1;
      // last canopy column of the range defines the end of the layer slice
      int last = lay_first;
      for (int j = istartcol - 1; j < iendcol; ++j)
        if (h_irep[j] != SS_TILE_FLAT) last = std::max(last, h_istartlay[j] - 1 + h_nlay[j]);
      lay_count = last - lay_first;
    }
    auto finish = [&](std::map<ClassKey, std::vector<int32_t>>& m, std::vector<ColumnClass>& out) {
      for (auto& kv : m) {
        ColumnClass cc;
        cc.key = kv.first;
        // descending layer count, stable: a counting sort over the layer counts that occur
        // (linear; nothing to do when all columns of the class have the same count)
        int lo = h_nlay[kv.second.front()], hi = lo;
        for (int32_t j : kv.second) { lo = std::min(lo, h_nlay[j]); hi = std::max(hi, h_nlay[j]); }
        if (lo == hi) {
          cc.cols = std::move(kv.second);
        } else {
          std::vector<size_t> start(hi - lo + 2, 0);
          for (int32_t j : kv.second) ++start[hi - h_nlay[j] + 1];
          for (size_t b = 1; b < start.size(); ++b) start[b] += start[b - 1];
          cc.cols.resize(kv.second.size());
          for (int32_t j : kv.second) cc.cols[start[hi - h_nlay[j]]++] = j;
        }
        cc.maxlay = hi;
        out.push_back(std::move(cc));
      }
    };
    finish(swm, sw_classes);
    finish(lwm, lw_classes);
    for (int lw = 0; lw < 2; ++lw)
      for (auto& c : (lw ? lw_classes : sw_classes)) {
        auto it = pool.find({lw, c.key});
        if (it != pool.end()) { c.d_pack = it->second.first; c.pack_elems = it->second.second; pool.erase(it); }
      }
    for (auto& kv : pool) be.free(kv.second.first);

    // one device index buffer: [sw classes][lw classes][flat][night][simple exp][simple inf]
    std::vector<int32_t> idx;
    for (auto& c : sw_classes) { c.dev_offset = idx.size(); idx.insert(idx.end(), c.cols.begin(), c.cols.end()); }
    for (auto& c : lw_classes) { c.dev_offset = idx.size(); idx.insert(idx.end(), c.cols.begin(), c.cols.end()); }
    flat_off = idx.size(); idx.insert(idx.end(), flat_cols.begin(), flat_cols.end());
    night_off = idx.size(); idx.insert(idx.end(), night_cols.begin(), night_cols.end());
    for (int k = 0; k < 2; ++k) { simple_off[k] = idx.size(); idx.insert(idx.end(), simple_cols[k].begin(), simple_cols[k].end()); }
    if (idx.empty()) idx.push_back(0);
    if (!d_index || index_elems < idx.size()) {
      if (d_index) be.free(d_index);
      d_index = static_cast<int32_t*>(be.alloc(idx.size() * sizeof(int32_t)));
      index_elems = idx.size();
    }
    be.h2d_staged(d_index, idx.data(), idx.size() * sizeof(int32_t));

    // scratch sizing: the largest per-column need over all classes
    size_t need_lay = 0, need_col = 0, need_eig = 0, need_list = 0;
    auto account = [&](const std::vector<ColumnClass>& cls, bool lw, int nspec) {
      for (auto& c : cls) {
        const size_t per_unit = (size_t)c.maxlay * sizeof(double);
        const size_t chunk = chunk_columns(c, lw, nspec);
        const size_t units = ((chunk * nspec + 63) / 64) * 64;     // whole tiles of 64 units
        need_lay = std::max(need_lay, units * per_unit * lay_nelem(lw, c.key));
        need_col = std::max(need_col, units * per_unit * col_nelem(lw, c.key));
        need_eig = std::max(need_eig, units * per_unit * eig_nelem(c.key));
        // (the dense list of cooperative unit-layers, at the END of the eigen buffer of a pipe: Launch::coop_list)
        if (eig_nelem(c.key)) need_list = std::max(need_list, ((units * (size_t)c.maxlay + 8) * sizeof(unsigned long long) + 255) / 256 * 256);
      }
    };
    account(sw_classes, false, nsw);
    account(lw_classes, true, nlw);
    coop_list_bytes = std::max(coop_list_bytes, need_list);
    if (need_eig) need_eig += coop_list_bytes;
    // a second pipe only pays off when there is more than one launch to overlap
    size_t nlaunch = 0;
    for (auto& c : sw_classes) nlaunch += (c.cols.size() + chunk_columns(c, false, nsw) - 1) / chunk_columns(c, false, nsw);
    for (auto& c : lw_classes) nlaunch += (c.cols.size() + chunk_columns(c, true, nlw) - 1) / chunk_columns(c, true, nlw);
    npipes = std::max(1, std::min<int>({kMaxPipes, pipes_wanted, (int)nlaunch}));
    if (need_lay > lay_bytes || need_col > col_bytes || need_eig > eig_bytes) {
      for (int p = 0; p < kMaxPipes; ++p) {
        if (d_lay[p]) be.free(d_lay[p]);
        if (d_col[p]) be.free(d_col[p]);
        if (d_eig[p]) be.free(d_eig[p]);
        d_lay[p] = d_col[p] = d_eig[p] = nullptr;
      }
      lay_bytes = std::max(lay_bytes, need_lay);
      col_bytes = std::max(col_bytes, need_col);
      eig_bytes = std::max(eig_bytes, need_eig);
    }
    for (int p = 0; p < npipes; ++p) {
      if (!d_lay[p]) d_lay[p] = static_cast<double*>(be.alloc(lay_bytes));
      if (!d_col[p]) d_col[p] = static_cast<double*>(be.alloc(col_bytes));
      if (!d_eig[p] && eig_bytes) d_eig[p] = static_cast<double*>(be.alloc(eig_bytes));
    }
    return SS_OK;
  }

  int set_pipes(int n) {
    pipes_wanted = std::max(1, std::min(n, (int)kMaxPipes));
    npipes = std::min(npipes, pipes_wanted);
    if (pipes_wanted > npipes && lay_bytes) {
      for (int p = npipes; p < pipes_wanted; ++p) {
        if (!d_lay[p]) d_lay[p] = static_cast<double*>(be.alloc(lay_bytes));
        if (!d_col[p]) d_col[p] = static_cast<double*>(be.alloc(col_bytes));
        if (!d_eig[p] && eig_bytes) d_eig[p] = static_cast<double*>(be.alloc(eig_bytes));
      }
      npipes = pipes_wanted;
    }
    return SS_OK;
  }

  // Columns of class c solved per launch so that both scratch buffers stay
  // within the budget
  // ... and so that a class gives every pipe several chunks (chunks_per_pipe, default 4, env
  // SPARTACUS_CHUNKS_PER_PIPE; 0: the budget alone decides): the layer kernels are compute-bound and the sweep
  // kernels bandwidth-bound, and the pipes only overlap the two kinds when they are in different phases -
  // with one chunk per pipe all of them run their layer kernel together and then their sweeps together
  // (C3, 10^6 columns: 62.9 -> 68.5e6 columns/s from 1 to 4 chunks per pipe and family).  Not below
  // kMinChunkUnits column-units = one wave on every SIMD of the device for the one-lane kernels (256 CUs x 4 SIMDs
  // x 64 lanes): smaller launches leave SIMDs idle and the step becomes a string of launch latencies.  Measured
  // against the 16 384 of rounds 2-3: C2 (10^5 two-layer columns) 3.6e8 -> 6.8e8 columns/s, C3 with 2 10^5 columns
  // 5.3e7 -> 6.4e7, the 10^6-column workloads unchanged (gpurun_out/r4z_min_chunk.txt -> profiles/).
  static constexpr size_t kMinChunkUnits = 65536;
  size_t min_chunk_units = kMinChunkUnits;     // env SPARTACUS_MIN_CHUNK_UNITS
  size_t chunk_columns(const ColumnClass& c, bool lw, int nspec) const {
    const size_t per_col = (size_t)nspec * c.maxlay * sizeof(double)
                           * (lay_nelem(lw, c.key) + col_nelem(lw, c.key) + eig_nelem(c.key));
    size_t chunk = std::max<size_t>(1, scratch_budget_bytes / std::max<size_t>(1, per_col));
    // (classes that stay on one pipe - orders above 16, whose kernels keep their matrices in the hardware scratch
    // and want launches as large as the budget allows - are cut by the budget alone: measured on the order-24
    // workload, 8.7e5 columns/s against 3.2e5 with 16 384-unit chunks)
    const int order = c.key.nreg * c.key.ns;
    const bool one_pipe = order > SS_UNROLL_MAX && !coop_order(order);
    if (chunks_per_pipe > 0 && !one_pipe) {
      // (counted for the default number of pipes, so that the chunk size - and with it the size of the
      // scratch buffers - does not depend on set_pipes())
      const size_t parts = (size_t)chunks_per_pipe * (size_t)kDefaultPipes;
      const size_t floor_cols = (min_chunk_units + (size_t)nspec - 1) / (size_t)std::max(1, nspec);
      const size_t even = std::max(floor_cols, (c.cols.size() + parts - 1) / parts);
      chunk = std::min(chunk, (even + 63) / 64 * 64);
    }
    if (chunk >= 128) chunk -= chunk % 64;     // whole waves where possible
    return std::max<size_t>(1, std::min(chunk, c.cols.size()));
  }

  // ---- building the launch descriptors ----
  static LG make_lg(const ss_legendre_gauss& g) {
    LG l;
    for (int i = 0; i < kDevMaxStreams; ++i) {
      const bool on = i < g.nstream;
      l.mu[i] = on ? g.mu[i] : 1.0;
      l.tan_ang[i] = on ? g.tan_ang[i] : 0.0;
      l.weight[i] = on ? g.weight[i] : 0.0;
      l.hweight[i] = on ? g.hweight[i] : 0.0;
      l.vweight[i] = on ? g.vweight[i] : 0.0;
      l.sqrt_wmu[i] = on ? std::sqrt(g.weight[i] * g.mu[i]) : 1.0;
      l.rsqrt_wmu[i] = 1.0 / l.sqrt_wmu[i];
    }
    l.vadjustment = g.vadjustment;
    l.vadjustment2 = g.vadjustment2;
    return l;
  }

  Inputs make_inputs(bool lw) const {
    Inputs in;
    std::memset(&in, 0, sizeof(in));
    in.nlay = static_cast<const int32_t*>(dev.at("canopy.nlay"));
    in.istartlay = static_cast<const int32_t*>(dev.at("canopy.istartlay"));
    in.cos_sza = darr("canopy.cos_sza");
    in.dz = darr("canopy.dz");
    in.building_fraction = darr("canopy.building_fraction");
    in.veg_fraction = darr("canopy.veg_fraction");
    in.building_scale = darr("canopy.building_scale");
    in.veg_scale = darr("canopy.veg_scale");
    in.veg_ext = darr("canopy.veg_ext");
    in.veg_fsd = darr("canopy.veg_fsd");
    in.veg_contact_fraction = darr("canopy.veg_contact_fraction");
    const std::string p = lw ? "lw." : "sw.";
    in.air_ext = darr((p + "air_ext").c_str());
    in.air_ssa = darr((p + "air_ssa").c_str());
    in.veg_ssa = darr((p + "veg_ssa").c_str());
    // both spectral families are filled (the flat / misc kernels need both)
    in.ground_albedo = darr("sw.ground_albedo");
    // ground_sw_albedo_dir pointer choice of radsurf_interface.F90:98-102
    in.ground_albedo_dir = cfg.use_sw_direct_albedo ? darr("sw.ground_albedo_dir") : darr("sw.ground_albedo");
    in.roof_albedo = darr("sw.roof_albedo");
    // radsurf_urban_sw.F90:631-643: roof_albedo_dir if allocated, else roof_albedo
    in.roof_albedo_dir = darr("sw.roof_albedo_dir") ? darr("sw.roof_albedo_dir") : darr("sw.roof_albedo");
    in.wall_albedo = darr("sw.wall_albedo");
    in.wall_specular_frac = darr("sw.wall_specular_frac");
    in.clear_air_planck = darr("lw.clear_air_planck");
    in.veg_planck = darr("lw.veg_planck");
    in.veg_air_planck = darr("lw.veg_air_planck");
    in.ground_emissivity = darr("lw.ground_emissivity");
    in.roof_emissivity = darr("lw.roof_emissivity");
    in.wall_emissivity = darr("lw.wall_emissivity");
    in.ground_emission = darr("lw.ground_emission");
    in.roof_emission = darr("lw.roof_emission");
    in.wall_emission = darr("lw.wall_emission");
    return in;
  }

  // Inputs a class needs must be present (the reference would dereference an
  // unallocated array)
  int check_inputs(const ClassKey& k, bool lw) {
    auto need = [&](const char* n) {
      if (!darr(n)) { error = std::string("required input array not provided: ") + n; return false; }
      return true;
    };
    bool ok = need("canopy.dz");
    const std::string p = lw ? "lw." : "sw.";
    ok = ok && need((p + "air_ext").c_str()) && need((p + "air_ssa").c_str());
    if (k.nreg > 1) {
      ok = ok && need("canopy.veg_fraction") && need("canopy.veg_scale") && need("canopy.veg_ext")
           && need((p + "veg_ssa").c_str());
      if (k.nreg == 3) ok = ok && need("canopy.veg_fsd");
      if (lw) ok = ok && need("lw.veg_planck") && need("lw.veg_air_planck");
    }
    if (k.urban) {
      ok = ok && need("canopy.building_fraction") && need("canopy.building_scale");
      if (k.nreg > 1) ok = ok && need("canopy.veg_contact_fraction");
      if (lw) ok = ok && need("lw.roof_emissivity") && need("lw.wall_emissivity") && need("lw.roof_emission") && need("lw.wall_emission");
      else ok = ok && need("sw.roof_albedo") && need("sw.wall_albedo") && need("sw.wall_specular_frac");
    }
    if (lw) ok = ok && need("lw.clear_air_planck") && need("lw.ground_emissivity") && need("lw.ground_emission");
    else ok = ok && need("sw.ground_albedo") && need("canopy.cos_sza")
              && (!cfg.use_sw_direct_albedo || need("sw.ground_albedo_dir"));
    return ok ? SS_OK : SS_ERR_BAD_ARGUMENT;
  }

  Launch make_launch(const ColumnClass& c, bool lw, size_t first, size_t count, int chunk_maxlay,
                     int pipe) const {
    Launch L;
    L.in = make_inputs(lw);
    const bool urban = c.key.urban;
    L.lg = make_lg(lw ? (urban ? cfg.lg_lw_urban : cfg.lg_lw_forest)
                      : (urban ? cfg.lg_sw_urban : cfg.lg_sw_forest));
    L.cfg.min_vegetation_fraction = cfg.min_vegetation_fraction;
    L.cfg.min_building_fraction = cfg.min_building_fraction;
    L.cfg.vegetation_isolation_factor = urban ? cfg.vegetation_isolation_factor_urban
                                              : cfg.vegetation_isolation_factor_forest;
    L.cfg.use_symmetric_vegetation_scale = urban ? cfg.use_symmetric_vegetation_scale_urban
                                                 : cfg.use_symmetric_vegetation_scale_forest;
    L.cfg.do_vegetation = cfg.do_vegetation;
    {
      const char* e = std::getenv("SPARTACUS_DECLINE_COOP");
      L.cfg.decline_cooperative = (e && e[0] == '1') ? 1 : (e && e[0] == '2') ? 2 : 0;
      const char* ds = std::getenv("SPARTACUS_DECLINE_SYM");
      L.cfg.decline_symmetric = (ds && ds[0] == '1') ? 1 : 0;
    }
    L.cols = d_index + c.dev_offset + first;
    L.ncols = (int)count;
    L.nspec = lw ? nlw : nsw;
    L.ncu = L.ncols * L.nspec;
    L.maxlay = chunk_maxlay;
    L.lay_nelem = lay_nelem(lw, c.key);
    L.col_nelem = col_nelem(lw, c.key);
    L.lay = d_lay[pipe];
    L.col = d_col[pipe];
    L.eig = d_eig[pipe];
    L.eig_nelem = eig_nelem(c.key);
    L.coop_list = (L.eig_nelem && d_eig[pipe] && coop_list_bytes && use_coop_list)
                      ? reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(d_eig[pipe]) + (eig_bytes - coop_list_bytes))
                      : nullptr;
    L.pack = c.d_pack;
    L.pack_maxlay = c.maxlay;
    L.pack_cu0 = (int)(first * (size_t)L.nspec);
    L.top1 = darr(lw ? "bc_out.lw_emissivity" : "bc_out.sw_albedo");
    L.top2 = darr(lw ? "bc_out.lw_emission" : "bc_out.sw_albedo_dir");
    L.flux_a = flux_dev(lw ? "lw_internal" : "sw_norm_dir");
    L.flux_b = flux_dev(lw ? "lw_norm" : "sw_norm_diff");
    L.eigen_fail = d_eigen_fail;
    return L;
  }

  MiscLaunch make_misc(size_t off, size_t count) const {
    MiscLaunch M;
    M.in = make_inputs(false);
    M.cols = d_index + off;
    M.ncols = (int)count;
    M.nsw = nsw; M.nlw = nlw;
    M.do_sw = cfg.do_sw; M.do_lw = cfg.do_lw;
    M.sw_albedo = darr("bc_out.sw_albedo");
    M.sw_albedo_dir = darr("bc_out.sw_albedo_dir");
    M.lw_emissivity = darr("bc_out.lw_emissivity");
    M.lw_emission = darr("bc_out.lw_emission");
    M.sw_dir = flux_dev("sw_norm_dir");
    M.sw_diff = flux_dev("sw_norm_diff");
    M.lw_int = flux_dev("lw_internal");
    M.lw_norm = flux_dev("lw_norm");
    M.min_building_fraction = cfg.min_building_fraction;
    M.infinite_street = 0;
    return M;
  }

  // ---- one pass of the hot path ----
  // Everything that can fail on the host side is checked before the first launch,
  // so that nothing returns between fork() and join(); a failing launch throws and
  // the worker streams are still joined (the caller may solve again).
  int validate_inputs() {
    auto need = [&](const char* n) {
      if (!darr(n)) { error = std::string("required input array not provided: ") + n; return false; }
      return true;
    };
    if (!flat_cols.empty()) {
      if (cfg.do_sw && !need("sw.ground_albedo")) return SS_ERR_BAD_ARGUMENT;
      if (cfg.do_lw && (!need("lw.ground_emissivity") || !need("lw.ground_emission"))) return SS_ERR_BAD_ARGUMENT;
    }
    if (!simple_cols[0].empty() || !simple_cols[1].empty()) {
      bool ok = need("canopy.dz") && need("canopy.building_fraction") && need("canopy.building_scale");
      if (cfg.do_sw) ok = ok && need("canopy.cos_sza") && need("sw.ground_albedo") && need("sw.roof_albedo") && need("sw.wall_albedo")
                          && (!cfg.use_sw_direct_albedo || need("sw.ground_albedo_dir"));
      if (cfg.do_lw) ok = ok && need("lw.ground_emissivity") && need("lw.ground_emission") && need("lw.roof_emissivity")
                          && need("lw.roof_emission") && need("lw.wall_emissivity") && need("lw.wall_emission");
      if (!ok) return SS_ERR_BAD_ARGUMENT;
    }
    for (int lw = 0; lw < 2; ++lw)
      for (const auto& c : (lw ? lw_classes : sw_classes)) {
        int rc = check_inputs(c.key, lw != 0);
        if (rc) return rc;
      }
    return SS_OK;
  }

  int solve() {
    int rc = validate_inputs();
    if (rc) return rc;
    be.begin_solve(timers);
    bool forked = false;
    try {
      be.memset_async(d_eigen_fail, 0, kCounters * sizeof(unsigned long long));
      if (!packs_valid) {      // first solve after an upload (or after the device spectrum changed the LW inputs)
        rc = pack_inputs();
        if (rc) { be.end_solve(timers); return rc; }
      }
      if (!flat_cols.empty()) be.launch_flat(make_misc(flat_off, flat_cols.size()), timers);
      if (!night_cols.empty()) be.launch_night(make_misc(night_off, night_cols.size()), timers);
      for (int k = 0; k < 2; ++k) {
        if (simple_cols[k].empty()) continue;
        MiscLaunch M = make_misc(simple_off[k], simple_cols[k].size());
        M.infinite_street = k;
        be.launch_simple_urban(M, timers);
      }
      be.fork(npipes);
      forked = true;
      // the chunks of the two spectral regions, launched alternately (a pipe then mostly stays with one
      // region, and the regions' kernels have different lengths: the pipes drift out of phase)
      struct Item { int lw; const ColumnClass* c; size_t first, count; };
      std::vector<Item> items[2];
      for (int lw = 0; lw < 2; ++lw) {
        const auto& classes = lw ? lw_classes : sw_classes;
        const int nspec = lw ? nlw : nsw;
        for (const auto& c : classes) {
          const size_t chunk = chunk_columns(c, lw != 0, nspec);
          for (size_t first = 0; first < c.cols.size(); first += chunk)
            items[lw].push_back(Item{lw, &c, first, std::min(chunk, c.cols.size() - first)});
        }
      }
      std::vector<Item> sequence;
      for (size_t k = 0; k < std::max(items[0].size(), items[1].size()); ++k)
        for (int lw = 0; lw < 2; ++lw)
          if (k < items[lw].size()) sequence.push_back(items[lw][k]);
      // every chunk goes to the pipe with the least work so far (a shortwave chunk costs about 4/3 of a
      // longwave chunk of the same size: 9.7 against 7.5 ms per 10^6 columns of C3)
      double pipe_load[kMaxPipes] = {0.0};
      for (const Item& it : sequence) {
        const int lw = it.lw;
        const ColumnClass& c = *it.c;
        const size_t first = it.first, count = it.count;
        const int chunk_maxlay = h_nlay[c.cols[first]];   // sorted descending
        // Orders above SS_UNROLL_MAX that the row-of-16 kernels do not take (above 16) keep
        // their matrices in the hardware scratch (tens of KB per lane, GBs per queue):
        // concurrent queues then fight over it (measured 5.5e5 -> 3.2e5 columns/s with four
        // streams at n = 12 in the one-lane form), so those classes stay on the first stream.
        // (The cooperative orders, 7..16, keep their one-lane second pass - 17 KB of scratch per lane
        // for the units the cooperative route declines - on the first stream whatever their pipe:
        // launch_ns, "fallback".)
        const int order = c.key.nreg * c.key.ns;
        const bool big_order = order > SS_UNROLL_MAX && !coop_order(order);
        int pipe = 0;
        if (!big_order) {
          for (int q = 1; q < npipes; ++q)
            if (pipe_load[q] < pipe_load[pipe]) pipe = q;
          pipe_load[pipe] += (double)count * (lw ? nlw : nsw) * chunk_maxlay * (lw ? 3.0 : 4.0);
        }
        Launch L = make_launch(c, lw != 0, first, count, chunk_maxlay, pipe);
        be.launch_solver(c.key, lw != 0, L, timers, pipe);
      }
      be.join(npipes);
      forked = false;
      be.end_solve(timers);
    } catch (...) {
      if (forked) { try { be.join(npipes); } catch (...) {} }
      try { be.end_solve(timers); } catch (...) {}
      throw;
    }
    return SS_OK;
  }

  int simple_spectrum_lw() {
    if (!cfg.do_lw) return SS_OK;
    if (nlw > 1) { error = "Simple longwave spectrum only possible with one input spectral interval"; return SS_ERR_BAD_ARGUMENT; }
    if (iendcol < istartcol) return SS_OK;
    SpectrumLaunch S;
    std::memset(&S, 0, sizeof(S));
    S.ground_temperature = darr("canopy.ground_temperature");
    S.roof_temperature = darr("canopy.roof_temperature");
    S.wall_temperature = darr("canopy.wall_temperature");
    S.clear_air_temperature = darr("canopy.clear_air_temperature");
    S.veg_temperature = darr("canopy.veg_temperature");
    S.veg_air_temperature = darr("canopy.veg_air_temperature");
    S.ground_emissivity = darr("lw.ground_emissivity");
    S.roof_emissivity = darr("lw.roof_emissivity");
    S.wall_emissivity = darr("lw.wall_emissivity");
    S.ground_emission = darr("lw.ground_emission");
    S.roof_emission = (S.roof_emissivity && S.roof_temperature) ? darr("lw.roof_emission") : nullptr;
    S.wall_emission = darr("lw.wall_emission");
    S.clear_air_planck = darr("lw.clear_air_planck");
    S.veg_planck = darr("lw.veg_planck");
    S.veg_air_planck = darr("lw.veg_air_planck");
    S.do_vegetation = cfg.do_vegetation && S.veg_planck && S.veg_temperature && S.veg_air_temperature;
    if (!S.ground_temperature || !S.ground_emissivity || !S.ground_emission) { error = "ground temperature/emissivity/emission arrays missing"; return SS_ERR_BAD_ARGUMENT; }
    S.col0 = istartcol - 1;
    S.ncols = iendcol - istartcol + 1;
    // calc_simple_spectrum_lw: ilay1 = istartlay(istartcol), ilay2 = istartlay(iendcol)+nlay(iendcol)-1
    S.lay0 = h_istartlay[istartcol - 1] - 1;
    S.nlays = (h_istartlay[iendcol - 1] - 1 + h_nlay[iendcol - 1]) - S.lay0;
    if (S.nlays > 0 && (!S.clear_air_temperature || !S.clear_air_planck)) { error = "air temperature / planck arrays missing"; return SS_ERR_BAD_ARGUMENT; }
    be.launch_spectrum(S, timers);
    packs_valid = false;     // emission / Planck arrays of the LW packs are stale now
    return SS_OK;
  }

  // ---- download: only the slices the reference would have written ----
  void get_cols(const char* name, double* host, int nspec) {
    if (!host || iendcol < istartcol) return;
    const double* d = darr(name);
    if (!d) return;
    const size_t off = (size_t)nspec * (istartcol - 1), n = (size_t)nspec * (iendcol - istartcol + 1);
    be.d2h(host + off, d + off, n * sizeof(double));
  }
  void get_lays(const char* name, double* host, int nspec) {
    if (!host || lay_count <= 0) return;
    const double* d = darr(name);
    if (!d) return;
    const size_t off = (size_t)nspec * lay_first, n = (size_t)nspec * lay_count;
    be.d2h(host + off, d + off, n * sizeof(double));
  }
  void put_cols(const char* name, const double* host, int nspec) {
    if (!host || iendcol < istartcol) return;
    double* d = darr(name);
    if (!d) return;
    const size_t off = (size_t)nspec * (istartcol - 1), n = (size_t)nspec * (iendcol - istartcol + 1);
    be.h2d(d + off, host + off, n * sizeof(double));
  }
  void put_lays(const char* name, const double* host, int nspec) {
    if (!host || lay_count <= 0) return;
    double* d = darr(name);
    if (!d) return;
    const size_t off = (size_t)nspec * lay_first, n = (size_t)nspec * lay_count;
    be.h2d(d + off, host + off, n * sizeof(double));
  }
  static double* const* flux_members(const ss_flux* f) { return &f->ground_dn; }

  void xfer_flux(const char* obj, ss_flux* f, int nspec, bool to_host) {
    if (!f) return;
    double* const* m = &f->ground_dn;
    for (int F = 0; F < F_NFIELDS; ++F) {
      const std::string name = std::string(obj) + "." + flux_field_name(F);
      const int shape = flux_field_shape(F);
      const int ns = (shape == 0 || shape == 1) ? nspec : 1;
      if (shape == 0 || shape == 2) { if (to_host) get_cols(name.c_str(), m[F], ns); else put_cols(name.c_str(), m[F], ns); }
      else if (to_host) get_lays(name.c_str(), m[F], ns);
      // (per-layer members are not seeded from the host: every layer of every column of the
      //  range is written by the solver or zeroed by the night kernel)
    }
  }

  // Two-phase form for callers that overlap the download with other work: _begin queues every
  // copy behind the kernels that are queued now and returns; _end waits until the host arrays
  // are complete.
  int download_begin(ss_bc_out* bc, ss_flux* sw_dir, ss_flux* sw_diff, ss_flux* lw_int, ss_flux* lw_nrm) {
    be.gate_copies();
    be.d2h_batch_begin();
    const int rc = download_queue(bc, sw_dir, sw_diff, lw_int, lw_nrm);
    be.d2h_batch_end();
    be.downloads_queued();
    return rc;
  }
  int download_end() {
    be.sync();
    return SS_OK;
  }
  int download(ss_bc_out* bc, ss_flux* sw_dir, ss_flux* sw_diff, ss_flux* lw_int, ss_flux* lw_nrm) {
    be.sync();
    download_queue(bc, sw_dir, sw_diff, lw_int, lw_nrm);
    be.sync();     // all copies were queued asynchronously: complete before the host reads
    return SS_OK;
  }
  int download_queue(ss_bc_out* bc, ss_flux* sw_dir, ss_flux* sw_diff, ss_flux* lw_int, ss_flux* lw_nrm) {
    if (bc) {
      if (cfg.do_sw) { get_cols("bc_out.sw_albedo", bc->sw_albedo, nsw); get_cols("bc_out.sw_albedo_dir", bc->sw_albedo_dir, nsw); }
      if (cfg.do_lw) { get_cols("bc_out.lw_emissivity", bc->lw_emissivity, nlw); get_cols("bc_out.lw_emission", bc->lw_emission, nlw); }
    }
    if (cfg.do_sw) { xfer_flux("sw_norm_dir", sw_dir, nsw, true); xfer_flux("sw_norm_diff", sw_diff, nsw, true); }
    if (cfg.do_lw) { xfer_flux("lw_internal", lw_int, nlw, true); xfer_flux("lw_norm", lw_nrm, nlw, true); }
    return SS_OK;
  }

  // Seed the device outputs with the caller's current contents, so members the
  // reference leaves untouched (e.g. bc_out SW of night-time columns,
  // per-column members of flat tiles) round-trip unchanged.
  // What has to be seeded for the classified column range: canopy columns in
  // daylight write every element of every member (a test holds this against the
  // seeded path with sentinel-filled outputs), night-time columns leave bc_out's
  // shortwave members alone, flat and single-layer urban tiles leave most members.
  enum { SEED_NONE = 0, SEED_BC_SW = 1, SEED_ALL = 2 };
  int seed_needed() const {
    if (!flat_cols.empty() || !simple_cols[0].empty() || !simple_cols[1].empty()) return SEED_ALL;
    return night_cols.empty() ? SEED_NONE : SEED_BC_SW;
  }
  int upload_outputs(const ss_bc_out* bc, ss_flux* sw_dir, ss_flux* sw_diff, ss_flux* lw_int, ss_flux* lw_nrm,
                     int mode = SEED_ALL) {
    if (mode == SEED_NONE) return SS_OK;
    if (bc) {
      if (cfg.do_sw) { put_cols("bc_out.sw_albedo", bc->sw_albedo, nsw); put_cols("bc_out.sw_albedo_dir", bc->sw_albedo_dir, nsw); }
      if (cfg.do_lw && mode == SEED_ALL) { put_cols("bc_out.lw_emissivity", bc->lw_emissivity, nlw); put_cols("bc_out.lw_emission", bc->lw_emission, nlw); }
    }
    if (mode != SEED_ALL) return SS_OK;
    if (cfg.do_sw) { xfer_flux("sw_norm_dir", sw_dir, nsw, false); xfer_flux("sw_norm_diff", sw_diff, nsw, false); }
    if (cfg.do_lw) { xfer_flux("lw_internal", lw_int, nlw, false); xfer_flux("lw_norm", lw_nrm, nlw, false); }
    return SS_OK;
  }

  // ---- canopy_flux%scale + %sum on the device-resident results ----
  // top_flux_dn_sw, top_flux_dn_direct_sw: (nsw, ncol); top_flux_dn_lw: (nlw, ncol)
  // (driver/spartacus_surface_driver.F90:250-261).  Works on the plan's column range.
  int scale_and_sum(const double* top_sw, const double* top_sw_dir, const double* top_lw) {
    if (iendcol < istartcol) return SS_OK;
    if (h_indcol.size() != (size_t)ntotlay) {
      h_indcol.assign(std::max(1, ntotlay), 0);
      for (int c = 0; c < ncol; ++c)
        for (int l = 0; l < h_nlay[c]; ++l) h_indcol[h_istartlay[c] - 1 + l] = c;
      mirror("canopy.indcol", h_indcol.data(), h_indcol.size(), sizeof(int32_t));
    }
    auto launch = [&](const char* oa, const char* ob, const char* osum, const double* fa, const double* fb,
                      int nspec, bool use_direct) {
      alloc_flux(osum, nspec, use_direct);
      ScaleSumLaunch S;
      S.a = flux_dev(oa);
      S.b = flux_dev(ob);
      S.sum = flux_dev(osum);
      S.fa = fa;
      S.fb = fb;
      S.indcol = static_cast<const int32_t*>(dev["canopy.indcol"]);
      S.nspec = nspec;
      S.col0 = istartcol - 1;
      S.ncols = iendcol - istartcol + 1;
      S.lay0 = lay_first;
      S.nlays = lay_count;
      scale_sum_compact(S);
      be.launch_scale_sum(S, timers);
    };
    if (cfg.do_sw) {
      if (!top_sw || !top_sw_dir) { error = "top-of-canopy shortwave fluxes missing"; return SS_ERR_BAD_ARGUMENT; }
      const size_t n = (size_t)nsw * ncol;
      std::vector<double> diff(n);
      for (size_t k = 0; k < n; ++k) diff[k] = top_sw[k] - top_sw_dir[k];
      mirror("factor.sw_dir", top_sw_dir, n);
      mirror("factor.sw_diff", diff.data(), n);
      launch("sw_norm_dir", "sw_norm_diff", "sw_flux", darr("factor.sw_dir"), darr("factor.sw_diff"), nsw, true);
    }
    if (cfg.do_lw) {
      if (!top_lw) { error = "top-of-canopy longwave flux missing"; return SS_ERR_BAD_ARGUMENT; }
      mirror("factor.lw", top_lw, (size_t)nlw * ncol);
      launch("lw_internal", "lw_norm", "lw_flux", nullptr, darr("factor.lw"), nlw, false);
    }
    return SS_OK;
  }
  int download_sums(ss_flux* sw_flux, ss_flux* lw_flux) {
    be.sync();
    if (cfg.do_sw) xfer_flux("sw_flux", sw_flux, nsw, true);
    if (cfg.do_lw) xfer_flux("lw_flux", lw_flux, nlw, true);
    be.sync();
    return SS_OK;
  }

  // emission / Planck arrays computed by simple_spectrum_lw go back to the host
  void download_spectrum(const ss_lw_spectral* lw) {
    if (!lw) return;
    be.sync();
    get_cols("lw.ground_emission", lw->ground_emission, nlw);
    get_lays("lw.roof_emission", lw->roof_emission, nlw);
    get_lays("lw.wall_emission", lw->wall_emission, nlw);
    get_lays("lw.clear_air_planck", lw->clear_air_planck, nlw);
    get_lays("lw.veg_planck", lw->veg_planck, nlw);
    get_lays("lw.veg_air_planck", lw->veg_air_planck, nlw);
    be.sync();
  }

  long long eigen_failures() {
    be.sync();
    unsigned long long v[kCounters] = {0};
    be.d2h(v, d_eigen_fail, sizeof(v));
    be.sync();
    if (std::getenv("SPARTACUS_PRINT_STAMPS")) {
      std::fprintf(stderr, "stamps:");
      for (int k = 1; k < kCounters; ++k) std::fprintf(stderr, " %llu", v[k]);
      std::fprintf(stderr, "\n");
    }
    return (long long)v[0];
  }
};

}  // namespace ss
