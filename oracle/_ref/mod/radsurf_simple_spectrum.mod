﻿!mod$ v1 sum:e70c07596ae20559
module radsurf_simple_spectrum
contains
subroutine calc_simple_spectrum_lw(config,canopy_props,lw_spectral_props,istartcol,iendcol)
use radsurf_canopy_properties,only:canopy_properties_type
use radsurf_config,only:config_type
use radsurf_lw_spectral_properties,only:lw_spectral_properties_type
type(config_type),intent(in)::config
type(canopy_properties_type),intent(in)::canopy_props
type(lw_spectral_properties_type),intent(inout)::lw_spectral_props
integer(4),intent(in)::istartcol
integer(4),intent(in)::iendcol
end
end
