﻿!mod$ v1 sum:d1eff9a3268a4359
module radsurf_urban_lw
contains
subroutine spartacus_urban_lw(config,nlw,ns,nreg,nlay,icol,ilay1,ilay2,lg,canopy_props,lw_spectral_props,top_emissivity,top_emission,lw_internal,lw_norm)
use radsurf_canopy_flux,only:canopy_flux_type
use radsurf_canopy_properties,only:canopy_properties_type
use radsurf_config,only:config_type
use radtool_legendre_gauss,only:legendre_gauss_type
use radsurf_lw_spectral_properties,only:lw_spectral_properties_type
type(config_type),intent(in)::config
integer(4),intent(in)::nlw
integer(4),intent(in)::ns
integer(4),intent(in)::nreg
integer(4),intent(in)::nlay
integer(4),intent(in)::icol
integer(4),intent(in)::ilay1
integer(4),intent(in)::ilay2
type(legendre_gauss_type),intent(in)::lg
type(canopy_properties_type),intent(in)::canopy_props
type(lw_spectral_properties_type),intent(in)::lw_spectral_props
real(8),intent(out)::top_emissivity(1_8:int(nlw,kind=8))
real(8),intent(out)::top_emission(1_8:int(nlw,kind=8))
type(canopy_flux_type),intent(inout),optional::lw_internal
type(canopy_flux_type),intent(inout),optional::lw_norm
end
end
