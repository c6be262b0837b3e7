﻿!mod$ v1 sum:6844f395c5e29291
!need$ a63c8333ed1677c5 n parkind1
module radsurf_canopy_flux
use parkind1,only:jpim
use parkind1,only:jprb
type::canopy_flux_type
real(8),allocatable::ground_dn(:,:)
real(8),allocatable::ground_dn_dir(:,:)
real(8),allocatable::ground_net(:,:)
real(8),allocatable::top_dn(:,:)
real(8),allocatable::top_dn_dir(:,:)
real(8),allocatable::top_net(:,:)
real(8),allocatable::roof_in(:,:)
real(8),allocatable::roof_in_dir(:,:)
real(8),allocatable::roof_net(:,:)
real(8),allocatable::wall_in(:,:)
real(8),allocatable::wall_in_dir(:,:)
real(8),allocatable::wall_net(:,:)
real(8),allocatable::roof_sunlit_frac(:)
real(8),allocatable::wall_sunlit_frac(:)
real(8),allocatable::ground_sunlit_frac(:)
real(8),allocatable::ground_vertical_diff(:,:)
real(8),allocatable::clear_air_abs(:,:)
real(8),allocatable::veg_abs(:,:)
real(8),allocatable::veg_air_abs(:,:)
real(8),allocatable::veg_abs_dir(:,:)
real(8),allocatable::veg_sunlit_frac(:)
real(8),allocatable::flux_dn_layer_top(:,:)
real(8),allocatable::flux_dn_dir_layer_top(:,:)
real(8),allocatable::flux_up_layer_top(:,:)
real(8),allocatable::flux_dn_layer_base(:,:)
real(8),allocatable::flux_dn_dir_layer_base(:,:)
real(8),allocatable::flux_up_layer_base(:,:)
integer(4)::nspec
integer(4)::ncol
integer(4)::ntotlay
contains
procedure::allocate=>allocate_canopy_flux
procedure::deallocate=>deallocate_canopy_flux
procedure::scale=>scale_canopy_flux
procedure::zero=>zero_canopy_flux
procedure::zero_all=>zero_all_canopy_flux
procedure::sum=>sum_canopy_flux
procedure::check=>check_canopy_flux
end type
contains
subroutine allocate_canopy_flux(this,config,ncol,ntotlay,nspec,use_direct,do_save_flux_profile)
use radsurf_config,only:config_type
class(canopy_flux_type),intent(inout)::this
type(config_type),intent(in)::config
integer(4),intent(in)::ncol
integer(4),intent(in)::ntotlay
integer(4),intent(in)::nspec
logical(4),intent(in),optional::use_direct
logical(4),intent(in),optional::do_save_flux_profile
end
subroutine deallocate_canopy_flux(this)
class(canopy_flux_type),intent(inout)::this
end
subroutine scale_canopy_flux(this,nlay,factor)
class(canopy_flux_type),intent(inout)::this
integer(4),intent(in)::nlay(:)
real(8),intent(in)::factor(:,:)
end
subroutine zero_canopy_flux(this,icol,ilay1,ilay2)
class(canopy_flux_type),intent(inout)::this
integer(4),intent(in)::icol
integer(4),intent(in),optional::ilay1
integer(4),intent(in),optional::ilay2
end
subroutine zero_all_canopy_flux(this)
class(canopy_flux_type),intent(inout)::this
end
subroutine sum_canopy_flux(this,flux1,flux2)
class(canopy_flux_type),intent(inout)::this
type(canopy_flux_type),intent(in)::flux1
type(canopy_flux_type),intent(in)::flux2
end
subroutine check_canopy_flux(this,canopy_props,istartcol,iendcol,iverbose)
use radsurf_canopy_properties,only:canopy_properties_type
class(canopy_flux_type),intent(inout)::this
type(canopy_properties_type),intent(in)::canopy_props
integer(4),intent(in),optional::istartcol
integer(4),intent(in),optional::iendcol
integer(4),intent(in),optional::iverbose
end
end
