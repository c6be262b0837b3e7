﻿!mod$ v1 sum:077dde184e79c98f
!need$ a63c8333ed1677c5 n parkind1
module radsurf_lw_spectral_properties
use parkind1,only:jpim
use parkind1,only:jprb
type::lw_spectral_properties_type
real(8),allocatable::air_ext(:,:)
real(8),allocatable::air_ssa(:,:)
real(8),allocatable::veg_ssa(:,:)
real(8),allocatable::clear_air_planck(:,:)
real(8),allocatable::veg_planck(:,:)
real(8),allocatable::veg_air_planck(:,:)
real(8),allocatable::ground_emissivity(:,:)
real(8),allocatable::roof_emissivity(:,:)
real(8),allocatable::wall_emissivity(:,:)
real(8),allocatable::ground_emission(:,:)
real(8),allocatable::roof_emission(:,:)
real(8),allocatable::wall_emission(:,:)
integer(4)::nspec
integer(4)::ncol
integer(4)::ntotlay
contains
procedure::allocate=>allocate_spectral
procedure::deallocate=>deallocate_spectral
procedure::calc_monochromatic_emission
end type
contains
subroutine allocate_spectral(this,config,nspec,ncol,ntotlay,i_representation)
use radsurf_config,only:config_type
class(lw_spectral_properties_type),intent(inout)::this
type(config_type),intent(in)::config
integer(4),intent(in)::nspec
integer(4),intent(in)::ncol
integer(4),intent(in)::ntotlay
integer(4),intent(in),optional::i_representation(:)
end
subroutine deallocate_spectral(this)
class(lw_spectral_properties_type),intent(inout)::this
end
subroutine calc_monochromatic_emission(this,canopy_props)
use radsurf_canopy_properties,only:canopy_properties_type
class(lw_spectral_properties_type),intent(inout)::this
type(canopy_properties_type),intent(in)::canopy_props
end
end
