/* spartacus_oracle.h - CPU restatement of the SPARTACUS-Surface column solver.
 *
 * TEST INFRASTRUCTURE ONLY: the checker the HIP path is compared against.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call this.  Parity status: PINNED against the real reference compiled from
 * /root/reference (oracle/_ref, see tests/test_oracle_vs_reference.py) and
 * against the committed golden vectors in tests/golden/.
 */
#ifndef SPARTACUS_ORACLE_H
#define SPARTACUS_ORACLE_H

#include "spartacus_surface_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Same argument list as spartacus_radsurf_hip plus: do_simple_spectrum (run
 * calc_simple_spectrum_lw first), nthreads (OpenMP threads over columns),
 * elapsed (seconds spent in the column loop). */
int spartacus_radsurf_oracle(const ss_config *config, const ss_canopy *canopy,
                             const ss_sw_spectral *sw, const ss_lw_spectral *lw,
                             ss_bc_out *bc_out, int32_t istartcol, int32_t iendcol,
                             ss_flux *sw_norm_dir, ss_flux *sw_norm_diff,
                             ss_flux *lw_internal, ss_flux *lw_norm,
                             int32_t do_simple_spectrum, int32_t nthreads,
                             double *elapsed);

/* Kernel-level entries on nmat stacked matrices in the reference's layout
 * (nmat, i, j) column-major, for direct comparison with the reference's
 * radtool routines. */
void oracle_eigen_decomposition_real(int32_t norder, int32_t nmat, const double *amat,
                                     double *eigenvalue, double *eigenvector);
void oracle_calc_matrices_sw_eig(int32_t nmat, int32_t ndiff, int32_t ndir, double dz,
                                 double mu0, const double *gamma0, const double *gamma1,
                                 const double *gamma2, const double *gamma3,
                                 double *reflectance, double *transmittance, double *s_up,
                                 double *s_dn, double *trans_dir, double *int_dir,
                                 double *int_diff, double *int_dir_diff);
void oracle_calc_matrices_lw_eig(int32_t nmat, int32_t norder, double dz,
                                 const double *gamma1, const double *gamma2,
                                 const double *emiss_rate, double *reflectance,
                                 double *transmittance, double *source, double *int_flux,
                                 double *int_flux_source);

#ifdef __cplusplus
}
#endif
#endif
