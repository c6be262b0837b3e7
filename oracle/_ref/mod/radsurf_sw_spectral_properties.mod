﻿!mod$ v1 sum:809a08002cb35c69
!need$ a63c8333ed1677c5 n parkind1
module radsurf_sw_spectral_properties
use parkind1,only:jpim
use parkind1,only:jprb
type::sw_spectral_properties_type
real(8),allocatable::air_ext(:,:)
real(8),allocatable::air_ssa(:,:)
real(8),allocatable::veg_ssa(:,:)
real(8),allocatable::ground_albedo(:,:)
real(8),allocatable::roof_albedo(:,:)
real(8),allocatable::wall_albedo(:,:)
real(8),allocatable::ground_albedo_dir(:,:)
real(8),allocatable::roof_albedo_dir(:,:)
real(8),allocatable::wall_specular_frac(:,:)
integer(4)::nspec
integer(4)::ncol
integer(4)::ntotlay
contains
procedure::allocate=>allocate_spectral
procedure::deallocate=>deallocate_spectral
end type
contains
subroutine allocate_spectral(this,config,ncol,ntotlay,nspec,i_representation)
use radsurf_config,only:config_type
class(sw_spectral_properties_type),intent(inout)::this
type(config_type),intent(in)::config
integer(4),intent(in)::ncol
integer(4),intent(in)::ntotlay
integer(4),intent(in)::nspec
integer(4),intent(in),optional::i_representation(:)
end
subroutine deallocate_spectral(this)
class(sw_spectral_properties_type),intent(inout)::this
end
end
