/* spartacus_surface_hip.h - C-ABI of the MI355X SPARTACUS-Surface column solver.
 *
 * This is the drop-in boundary for ONE path of ecmwf/spartacus-surface: the
 * per-column flux solver reached through
 *     subroutine radsurf(config, canopy_props, sw_spectral_props,
 *                        lw_spectral_props, bc_out, istartcol, iendcol,
 *                        sw_norm_dir, sw_norm_diff, lw_internal, lw_norm)
 * (reference: radsurf/radsurf_interface.F90:20-25, intents :48-59).
 *
 * Every struct below is a flat, plain-C view of one of the reference's Fortran
 * derived types: dimensions plus raw pointers to the caller's own contiguous
 * (allocatable) Fortran arrays.  A NULL pointer means "component not
 * allocated", which is how the reference signals that a feature is off
 * (e.g. flux_dn_layer_top => no flux profiles, radsurf_forest_sw.F90:583;
 * roof_albedo_dir, radsurf_urban_sw.F90:631).  All reals are fp64
 * (jprb = jprd, utilities/parkind1.F90:42-49), all integers 32 bit (jpim).
 * Array index order is Fortran column-major with the spectral index first:
 *     x(nspec, ncol)      ->  x[ispec + nspec*icol]
 *     x(nspec, ntotlay)   ->  x[ispec + nspec*ilay]
 * Layers are packed: column j owns layers istartlay[j] .. istartlay[j]+nlay[j]-1
 * (1-based, exactly the Fortran values; radsurf_canopy_properties.F90:53-110).
 *
 * The Fortran side of the boundary is spartacus_surface_amd/fortran/
 * radsurf_interface_hip.F90: a replacement "module radsurf_interface" with the
 * identical radsurf signature that fills these structs with c_loc() pointers.
 *
 * No torch types, no C++ types, no device pointers in the host entry points.
 */
#ifndef SPARTACUS_SURFACE_HIP_H
#define SPARTACUS_SURFACE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Maximum streams per hemisphere carried in the Legendre-Gauss tables.  The
 * reference's shipped tests use 1, 2, 4 and 8 (test/.../config.nam). */
#define SS_MAX_STREAMS 16

/* Tile representation codes (radsurf_canopy_properties.F90:26-33). */
enum {
  SS_TILE_FLAT = 0,
  SS_TILE_FOREST = 1,
  SS_TILE_URBAN = 2,
  SS_TILE_VEGETATED_URBAN = 3,
  SS_TILE_SIMPLE_URBAN = 4,
  SS_TILE_INFINITE_STREET = 5
};

/* Status codes returned by every entry point (0 = success).  The Fortran shim
 * turns non-zero into radiation_abort, matching the reference's error
 * behaviour (utilities/radiation_io.F90:46-54). */
enum {
  SS_OK = 0,
  SS_ERR_NO_DEVICE = 1,          /* HIP runtime / MI355X not available        */
  SS_ERR_UNSUPPORTED = 2,        /* (nstream, nregion) combination not built  */
  SS_ERR_SIMPLE_URBAN_LAYERS = 3,/* radsurf_interface.F90:281-284             */
  SS_ERR_VEG_DISABLED = 4,       /* radsurf_urban_sw.F90:275-277              */
  SS_ERR_BAD_ARGUMENT = 5,
  SS_ERR_HIP = 6,                /* a HIP call failed; see ss_last_error()    */
  SS_ERR_TILE_NOT_ON_DEVICE = 7  /* no longer returned: tile types 4,5 are solved
                                    on the device; value kept for ABI stability */
};

/* legendre_gauss_type (radtool/radtool_legendre_gauss.F90:25-48). */
typedef struct ss_legendre_gauss {
  int32_t nstream;
  int32_t pad_;
  double mu[SS_MAX_STREAMS];
  double sin_ang[SS_MAX_STREAMS];
  double tan_ang[SS_MAX_STREAMS];
  double weight[SS_MAX_STREAMS];
  double hweight[SS_MAX_STREAMS];
  double vweight[SS_MAX_STREAMS];
  double vadjustment;
  double vadjustment2;
} ss_legendre_gauss;

/* config_type: the members the solver reads (radsurf_config.F90:32-113). */
typedef struct ss_config {
  int32_t do_sw;
  int32_t do_lw;
  int32_t use_sw_direct_albedo;
  int32_t do_vegetation;
  int32_t do_urban;
  int32_t n_vegetation_region_forest;
  int32_t n_vegetation_region_urban;
  int32_t nsw;                 /* nswinternal */
  int32_t nlw;                 /* nlwinternal */
  int32_t use_symmetric_vegetation_scale_forest;
  int32_t use_symmetric_vegetation_scale_urban;
  int32_t iverbose;
  double vegetation_isolation_factor_forest;
  double vegetation_isolation_factor_urban;
  double min_vegetation_fraction;
  double min_building_fraction;
  ss_legendre_gauss lg_sw_forest;
  ss_legendre_gauss lg_sw_urban;
  ss_legendre_gauss lg_lw_forest;
  ss_legendre_gauss lg_lw_urban;
} ss_config;

/* canopy_properties_type (radsurf_canopy_properties.F90:53-110).  The
 * temperature members are not read by radsurf itself; they feed the optional
 * calc_simple_spectrum_lw pre-step (radsurf_simple_spectrum.F90:41-66). */
typedef struct ss_canopy {
  int32_t ncol;
  int32_t ntotlay;
  const int32_t *nlay;             /* (ncol) */
  const int32_t *istartlay;        /* (ncol) 1-based */
  const int32_t *i_representation; /* (ncol) SS_TILE_* */
  const double *cos_sza;           /* (ncol) */
  const double *dz;                /* (ntotlay) */
  const double *building_fraction; /* (ntotlay) */
  const double *veg_fraction;      /* (ntotlay) */
  const double *building_scale;    /* (ntotlay) */
  const double *veg_scale;         /* (ntotlay) */
  const double *veg_ext;           /* (ntotlay) */
  const double *veg_fsd;           /* (ntotlay) */
  const double *veg_contact_fraction; /* (ntotlay) */
  const double *ground_temperature;    /* (ncol)    */
  const double *roof_temperature;      /* (ntotlay) */
  const double *wall_temperature;      /* (ntotlay) */
  const double *clear_air_temperature; /* (ntotlay) */
  const double *veg_temperature;       /* (ntotlay) */
  const double *veg_air_temperature;   /* (ntotlay) */
} ss_canopy;

/* sw_spectral_properties_type (radsurf_sw_spectral_properties.F90:24-49). */
typedef struct ss_sw_spectral {
  int32_t nspec;
  int32_t pad_;
  const double *air_ext;            /* (nspec,ntotlay) */
  const double *air_ssa;            /* (nspec,ntotlay) */
  const double *veg_ssa;            /* (nspec,ntotlay) */
  const double *ground_albedo;      /* (nspec,ncol)    */
  const double *roof_albedo;        /* (nspec,ntotlay) */
  const double *wall_albedo;        /* (nspec,ntotlay) */
  const double *ground_albedo_dir;  /* (nspec,ncol)    */
  const double *roof_albedo_dir;    /* (nspec,ntotlay) */
  const double *wall_specular_frac; /* (nspec,ntotlay) */
} ss_sw_spectral;

/* lw_spectral_properties_type (radsurf_lw_spectral_properties.F90:24-57).
 * The emission / Planck members are written by ss_calc_simple_spectrum_lw. */
typedef struct ss_lw_spectral {
  int32_t nspec;
  int32_t pad_;
  const double *air_ext;           /* (nspec,ntotlay) */
  const double *air_ssa;           /* (nspec,ntotlay) */
  const double *veg_ssa;           /* (nspec,ntotlay) */
  double *clear_air_planck;        /* (nspec,ntotlay) */
  double *veg_planck;              /* (nspec,ntotlay) */
  double *veg_air_planck;          /* (nspec,ntotlay) */
  const double *ground_emissivity; /* (nspec,ncol)    */
  const double *roof_emissivity;   /* (nspec,ntotlay) */
  const double *wall_emissivity;   /* (nspec,ntotlay) */
  double *ground_emission;         /* (nspec,ncol)    */
  double *roof_emission;           /* (nspec,ntotlay) */
  double *wall_emission;           /* (nspec,ntotlay) */
} ss_lw_spectral;

/* boundary_conds_out_type (radsurf_boundary_conds_out.F90:24-39). */
typedef struct ss_bc_out {
  double *sw_albedo;     /* (nsw,ncol) */
  double *sw_albedo_dir; /* (nsw,ncol) */
  double *lw_emissivity; /* (nlw,ncol) */
  double *lw_emission;   /* (nlw,ncol) */
} ss_bc_out;

/* canopy_flux_type (radsurf_canopy_flux.F90:27-91).  Member order is the
 * order of declaration in the reference. */
typedef struct ss_flux {
  int32_t nspec;
  int32_t ncol;
  int32_t ntotlay;
  int32_t pad_;
  double *ground_dn;              /* (nspec,ncol)    */
  double *ground_dn_dir;          /* (nspec,ncol)    */
  double *ground_net;             /* (nspec,ncol)    */
  double *top_dn;                 /* (nspec,ncol)    */
  double *top_dn_dir;             /* (nspec,ncol)    */
  double *top_net;                /* (nspec,ncol)    */
  double *roof_in;                /* (nspec,ntotlay) */
  double *roof_in_dir;            /* (nspec,ntotlay) */
  double *roof_net;               /* (nspec,ntotlay) */
  double *wall_in;                /* (nspec,ntotlay) */
  double *wall_in_dir;            /* (nspec,ntotlay) */
  double *wall_net;               /* (nspec,ntotlay) */
  double *roof_sunlit_frac;       /* (ntotlay)       */
  double *wall_sunlit_frac;       /* (ntotlay)       */
  double *ground_sunlit_frac;     /* (ncol)          */
  double *ground_vertical_diff;   /* (nspec,ncol)    */
  double *clear_air_abs;          /* (nspec,ntotlay) */
  double *veg_abs;                /* (nspec,ntotlay) */
  double *veg_air_abs;            /* (nspec,ntotlay) */
  double *veg_abs_dir;            /* (nspec,ntotlay) */
  double *veg_sunlit_frac;        /* (ntotlay)       */
  double *flux_dn_layer_top;      /* (nspec,ntotlay) */
  double *flux_dn_dir_layer_top;  /* (nspec,ntotlay) */
  double *flux_up_layer_top;      /* (nspec,ntotlay) */
  double *flux_dn_layer_base;     /* (nspec,ntotlay) */
  double *flux_dn_dir_layer_base; /* (nspec,ntotlay) */
  double *flux_up_layer_base;     /* (nspec,ntotlay) */
} ss_flux;

#define SS_FLUX_NFIELDS 27

/* --------------------------------------------------------------------------
 * Host-buffer entry: replaces radsurf_interface::radsurf
 * (radsurf/radsurf_interface.F90:20-313).  Same argument order and meaning.
 * istartcol/iendcol are 1-based inclusive; pass 1 and canopy->ncol for the
 * Fortran "optional absent" case (radsurf_interface.F90:86-96).  Flux objects
 * may be NULL when the corresponding do_sw / do_lw is off.  Only slices
 * (:,istartcol:iendcol) and the matching packed layer ranges of the outputs
 * are written, exactly as the reference does, so disjoint column ranges may be
 * processed by concurrent callers (driver/spartacus_surface_driver.F90:206-233);
 * calls are serialised per device inside the library.
 * The pointers are HOST pointers; the library stages them to HBM, runs the HIP
 * kernels on the current device and copies the results back.
 * The columns of a call are cut into chunks that alternate between three device
 * plans, so that the upload of one chunk, the kernels of the previous one and
 * the download of the one before overlap (PCIe is what bounds this entry; uploads
 * and downloads have a stream each, shared by the plans).  The
 * plans persist between calls while the configuration, the profile flag and
 * the current device stay the same - whatever the shapes and the column range:
 * their device arrays only grow, and only the slices of [istartcol, iendcol]
 * cross the bus, so the reference driver's block loop (a different column block
 * per call, nblocksize << ncol) re-uses them.  SPARTACUS_NO_PLAN_CACHE=1 drops
 * them after every call, SPARTACUS_HOST_CHUNK=<columns> sets the chunk size.
 * -------------------------------------------------------------------------- */
int spartacus_radsurf_hip(const ss_config *config, const ss_canopy *canopy,
                          const ss_sw_spectral *sw, const ss_lw_spectral *lw,
                          ss_bc_out *bc_out, int32_t istartcol, int32_t iendcol,
                          ss_flux *sw_norm_dir, ss_flux *sw_norm_diff,
                          ss_flux *lw_internal, ss_flux *lw_norm);

/* Optional page-locking of the caller's arrays for the host entry: copies of a
 * locked array are direct DMA at PCIe rate that does not block the calling
 * thread, so uploads, kernels and downloads of successive column chunks overlap
 * in both directions.  Pageable arrays (the default) are always correct, just slower.
 *
 * ss_host_pin(ptr, bytes): the CALLER owns the lock.  CONTRACT: the range stays
 * allocated until ss_host_unpin(ptr) / ss_host_unpin_all() - a freed array whose
 * address the allocator re-uses for new pages makes the next copy a DMA to pages
 * that are gone: the GPU faults and the runtime ends the process (the library
 * cannot see a free).  Meant for a host model whose allocatable arrays live for
 * the whole run: pin once after allocation, unpin before deallocation.  A range
 * that partly overlaps an existing lock is refused (SS_ERR_HIP).
 *
 * Call-owned locks (the default for calls of 32768 columns and more): spartacus_radsurf_hip
 * locks the slices of [istartcol, iendcol] it is handed on entry and releases them
 * before it returns - safe with any allocation pattern, nothing survives the call.
 * SPARTACUS_PIN_HOST=1 lowers the threshold to SPARTACUS_PIN_HOST_MIN_COLUMNS (default
 * 4096) columns, SPARTACUS_PIN_HOST=0 switches the locking off (every copy is then
 * staged by the runtime and blocks the calling thread). */
int ss_host_pin(const void *ptr, int64_t bytes);
int ss_host_unpin(const void *ptr);
int ss_host_unpin_all(void);
int ss_host_pinned_bytes(int64_t *bytes);

/* --------------------------------------------------------------------------
 * Device-resident plan API (what bench.py times and what a host model that
 * keeps its fields on the GPU would call).  A plan owns device mirrors of all
 * input and output arrays of one column set; upload / solve / download are
 * separate so the timed region can start with inputs already in HBM.
 * -------------------------------------------------------------------------- */
typedef struct ss_plan ss_plan;

/* Create a plan for the column set described by (config, canopy, sw, lw):
 * classifies columns by tile type / layer count, allocates device mirrors,
 * per-layer matrix scratch and the output arrays.  want_profiles != 0
 * allocates the six SW / four LW layer-profile arrays (the stock driver always
 * does: radsurf_canopy_flux.F90:114-118). */
int ss_plan_create(ss_plan **plan, const ss_config *config,
                   const ss_canopy *canopy, const ss_sw_spectral *sw,
                   const ss_lw_spectral *lw, int32_t istartcol,
                   int32_t iendcol, int32_t want_profiles);
/* Copy (again) the input arrays host -> HBM. */
int ss_plan_upload(ss_plan *plan, const ss_canopy *canopy,
                   const ss_sw_spectral *sw, const ss_lw_spectral *lw);
/* Optional pre-step calc_simple_spectrum_lw on device
 * (radsurf/radsurf_simple_spectrum.F90:41-66). */
int ss_plan_calc_simple_spectrum_lw(ss_plan *plan);
/* One pass of the hot path over every column of the plan (asynchronous on the
 * plan's stream). */
int ss_plan_solve(ss_plan *plan);
/* Block until the plan's stream is idle. */
int ss_plan_synchronize(ss_plan *plan);
/* Copy results HBM -> host arrays (only the plan's column / layer slices). */
int ss_plan_download(ss_plan *plan, ss_bc_out *bc_out, ss_flux *sw_norm_dir,
                     ss_flux *sw_norm_diff, ss_flux *lw_internal,
                     ss_flux *lw_norm);
/* Average device time (ms) of the kernels launched by the last ss_plan_solve,
 * measured with HIP events on the plan's own stream; name selects a kernel
 * family ("sw_layer", "lw_layer", "sw_column", "lw_column", "total"). */
int ss_plan_last_kernel_ms(ss_plan *plan, const char *name, double *ms);
/* Number of eigenproblems whose shifted-QR iteration hit the 30-iteration cap
 * in the last solve (the reference silently ignores these:
 * radtool_eigen_decomposition.F90:498-510, calc_matrices_sw_eig.F90:183). */
int ss_plan_eigen_failures(ss_plan *plan, int64_t *count);
/* canopy_flux%scale and %sum (radsurf/radsurf_canopy_flux.F90:212-282, 399-460)
 * exactly as the offline driver applies them after radsurf
 * (driver/spartacus_surface_driver.F90:250-261), on the device-resident results
 * of the last ss_plan_solve, for the plan's column range:
 *     sw_norm_dir  *= top_flux_dn_direct_sw
 *     sw_norm_diff *= top_flux_dn_sw - top_flux_dn_direct_sw
 *     sw_flux       = sw_norm_dir + sw_norm_diff
 *     lw_norm      *= top_flux_dn_lw
 *     lw_flux       = lw_internal + lw_norm
 * The factors are host arrays (nsw, ncol) / (nlw, ncol); sunlit fractions are
 * not scaled.  A later ss_plan_download returns the scaled objects,
 * ss_plan_download_sums the two sums (members laid out like any ss_flux). */
int ss_plan_scale_and_sum(ss_plan *plan, const double *top_flux_dn_sw,
                          const double *top_flux_dn_direct_sw,
                          const double *top_flux_dn_lw);
int ss_plan_download_sums(ss_plan *plan, ss_flux *sw_flux, ss_flux *lw_flux);
/* Number of concurrent HIP streams ("pipes") the next ss_plan_solve spreads its
 * chunk launches over: 1 = strictly serial kernels (what per-kernel profiling
 * wants), up to 8 (default 4 - more did not pay on MI355X -, env SPARTACUS_PIPES) = the memory-bound column
 * kernels of one chunk overlap with the arithmetic-bound layer kernels of others.  No
 * counterpart in the reference (its concurrency is the OpenMP block loop,
 * driver/spartacus_surface_driver.F90:206-233). */
int ss_plan_set_pipes(ss_plan *plan, int32_t npipes);
/* Raw device pointer of an output / input mirror by name, e.g.
 * "sw_norm_dir.ground_dn", "bc_out.sw_albedo" (for RCCL gathers). */
int ss_plan_device_ptr(ss_plan *plan, const char *name, void **ptr,
                       int64_t *nelem);
/* Device the plan lives on (the device that was current at ss_plan_create).  A
 * plan owns its HIP streams and events; every ss_plan_* call makes the plan's
 * device current.  Calls on one plan must not overlap in time (different plans
 * are independent). */
int ss_plan_device(ss_plan *plan, int32_t *device);
int ss_plan_destroy(ss_plan *plan);
/* Number of device plans the host entry spartacus_radsurf_hip has had to build
 * so far in this process (diagnostic: block-wise calls must not rebuild them). */
int ss_host_entry_plan_creations(int64_t *count);

/* --------------------------------------------------------------------------
 * Several GPUs of one node from ONE process (scope row e).  The reference's only
 * parallel construct is one process looping over column blocks
 * (driver/spartacus_surface_driver.F90:203-234), so the fan-out lives inside the
 * library: nothing of the host model changes, no MPI, no second process.
 *
 * ss_set_devices: large calls of spartacus_radsurf_hip (more than 2048 columns) cut
 * [istartcol, iendcol] into one contiguous shard per listed device, balanced by
 * layer count; one host thread per shard drives its device's chunk pipeline, and
 * every device moves its own slices of the caller's arrays - no result crosses
 * from one device to another, no gather.  ndevices = 0 returns to the calling
 * thread's current device.  A device may be listed more than once (several shards on
 * one GPU).  Without a call the list is read once from the environment:
 * SPARTACUS_DEVICES = "all" | <count> | <d0>,<d1>,...
 *
 * ss_plan_create_multi: the device-resident form - one plan per shard, each on its
 * device, over disjoint column ranges of the SAME caller objects.  solve queues the
 * work of every part and returns; download lets every part write its slices of the
 * full-size objects.  ss_multi_plan_part hands out the parts (plan, device, column
 * range) for every other plan entry (upload, scale_and_sum, device pointers ...).
 * -------------------------------------------------------------------------- */
int ss_set_devices(const int32_t *devices, int32_t ndevices);
int ss_get_devices(int32_t *devices, int32_t capacity, int32_t *ndevices);
typedef struct ss_multi_plan ss_multi_plan;
int ss_plan_create_multi(ss_multi_plan **mplan, const ss_config *config, const ss_canopy *canopy,
                         const ss_sw_spectral *sw, const ss_lw_spectral *lw, int32_t istartcol,
                         int32_t iendcol, int32_t want_profiles, const int32_t *devices,
                         int32_t ndevices);
int ss_multi_plan_count(ss_multi_plan *mplan, int32_t *nparts);
int ss_multi_plan_part(ss_multi_plan *mplan, int32_t k, ss_plan **plan, int32_t *device,
                       int32_t *istartcol, int32_t *iendcol);
int ss_multi_plan_solve(ss_multi_plan *mplan);
int ss_multi_plan_synchronize(ss_multi_plan *mplan);
int ss_multi_plan_download(ss_multi_plan *mplan, ss_bc_out *bc_out, ss_flux *sw_norm_dir,
                           ss_flux *sw_norm_diff, ss_flux *lw_internal, ss_flux *lw_norm);
int ss_multi_plan_destroy(ss_multi_plan *mplan);

/* --------------------------------------------------------------------------
 * Kernel-level entry points: the layer-matrix units of the hot path on their
 * own, with the argument lists of radtool's routines (stacks of nmat independent
 * problems, HOST arrays in Fortran layout x(nmat, i, j) -> x[m + nmat*(i + n*j)]).
 * They launch the same device bodies the solver kernels inline, one lane per
 * problem.  Instantiated orders: (ndiff, ndir) in {(1,1) (2,1) (2,2) (3,3) (4,2)
 * (6,3) (8,2) (12,3)} for the shortwave, 1 2 3 4 6 8 12 for the longwave and
 * 1 2 3 4 5 6 8 12 for the eigen-decomposition; others return SS_ERR_UNSUPPORTED.
 * Any output pointer may be NULL.
 * -------------------------------------------------------------------------- */
/* calc_matrices_sw_eig (radtool/radtool_calc_matrices_sw_eig.F90:30-298, with
 * schur_invert_sw radtool/radtool_schur.F90:32-53 for the three int_* outputs).
 * dscale0 (nmat, ndir) / dscale (nmat, ndiff): optional scaling vectors d with
 * diag(d)^-1 Gamma diag(d) symmetric (the solver passes sqrt(region fraction) and
 * sqrt(fraction * weight * mu)); with them orders 3..6 take the symmetric
 * eigen-route and orders 7..16 the cooperative Jacobi route (two more launches),
 * NULL = the reference's general route.  sym_route (nmat, optional) receives 1
 * where the symmetric route was taken, 2 for the cooperative route, else 0.  mu0 is accepted for the
 * reference's signature and unused, as in the reference (:377, :384). */
int ss_calc_matrices_sw_eig(int32_t nmat, int32_t ndiff, int32_t ndir, double dz, double mu0,
                            const double *gamma0, const double *gamma1, const double *gamma2,
                            const double *gamma3, const double *dscale0, const double *dscale,
                            double *reflectance, double *transmittance, double *s_up, double *s_dn,
                            double *trans_dir, double *int_dir, double *int_diff,
                            double *int_dir_diff, int32_t *sym_route);
/* calc_matrices_lw_eig (radtool/radtool_calc_matrices_lw_eig.F90:32-230). */
int ss_calc_matrices_lw_eig(int32_t nmat, int32_t norder, double dz, const double *gamma1,
                            const double *gamma2, const double *emiss_rate, const double *dscale,
                            double *reflectance, double *transmittance, double *source,
                            double *int_flux, double *int_flux_source);
/* eigen_decomposition_real (radtool/radtool_eigen_decomposition.F90:51-828; note
 * its argument order norder, nmat).  force_general_loops != 0 runs the run-time
 * loop form for every order instead of the register-resident form of orders 3..6
 * (same arithmetic, must agree bit for bit).  ierr (nmat, optional): 0 or the
 * index at which the QR iteration gave up (:498-510). */
int ss_eigen_decomposition_real(int32_t norder, int32_t nmat, const double *amat,
                                double *eigenvalue, double *eigenvector,
                                int32_t force_general_loops, int32_t *ierr);
const char *ss_units_last_error(void);

/* Library / device information. */
int ss_device_count(int32_t *count);
/* hipSetDevice for plans created from now on and for the host entry; existing
 * plans stay on their device. */
int ss_set_device(int32_t device);
const char *ss_last_error(void);
const char *ss_version(void);
/* hash of the sources this library was built from (csrc/Makefile): stamps the committed counter summaries */
const char *ss_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* SPARTACUS_SURFACE_HIP_H */
