﻿!mod$ v1 sum:7159225ee16a948e
module radsurf_norm_perim
contains
subroutine calc_norm_perim_forest(config,nlay,nreg,veg_fraction,veg_scale,norm_perim)
use radsurf_config,only:config_type
type(config_type),intent(in)::config
integer(4),intent(in)::nlay
integer(4),intent(in)::nreg
real(8),intent(in)::veg_fraction(1_8:int(nlay,kind=8))
real(8),intent(in)::veg_scale(1_8:int(nlay,kind=8))
real(8),intent(out)::norm_perim(1_8:int(nreg,kind=8),1_8:int(nlay,kind=8))
end
subroutine calc_norm_perim_urban(config,nlay,nreg,building_fraction,building_scale,veg_fraction,veg_scale,veg_contact_fraction,norm_perim,norm_perim_wall)
use radsurf_config,only:config_type
type(config_type),intent(in)::config
integer(4),intent(in)::nlay
integer(4),intent(in)::nreg
real(8),intent(in)::building_fraction(1_8:int(nlay,kind=8))
real(8),intent(in)::building_scale(1_8:int(nlay,kind=8))
real(8),intent(in)::veg_fraction(1_8:int(nlay,kind=8))
real(8),intent(in)::veg_scale(1_8:int(nlay,kind=8))
real(8),intent(in)::veg_contact_fraction(1_8:int(nlay,kind=8))
real(8),intent(out)::norm_perim(1_8:int(nreg,kind=8),1_8:int(nlay,kind=8))
real(8),intent(out)::norm_perim_wall(1_8:int(nreg,kind=8),1_8:int(nlay,kind=8))
end
end
