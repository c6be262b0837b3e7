﻿!mod$ v1 sum:136c1997a309e43f
module radsurf_interface
contains
subroutine radsurf(config,canopy_props,sw_spectral_props,lw_spectral_props,bc_out,istartcol,iendcol,sw_norm_dir,sw_norm_diff,lw_internal,lw_norm)
use radsurf_boundary_conds_out,only:boundary_conds_out_type
use radsurf_canopy_flux,only:canopy_flux_type
use radsurf_canopy_properties,only:canopy_properties_type
use radsurf_config,only:config_type
use radsurf_lw_spectral_properties,only:lw_spectral_properties_type
use radsurf_sw_spectral_properties,only:sw_spectral_properties_type
type(config_type),intent(in)::config
type(canopy_properties_type),intent(in)::canopy_props
type(sw_spectral_properties_type),intent(in),target::sw_spectral_props
type(lw_spectral_properties_type),intent(in),target::lw_spectral_props
type(boundary_conds_out_type),intent(inout)::bc_out
integer(4),intent(in),optional::istartcol
integer(4),intent(in),optional::iendcol
type(canopy_flux_type),intent(inout),optional::sw_norm_dir
type(canopy_flux_type),intent(inout),optional::sw_norm_diff
type(canopy_flux_type),intent(inout),optional::lw_internal
type(canopy_flux_type),intent(inout),optional::lw_norm
end
end
