﻿!mod$ v1 sum:426d6ee2724ff0dc
!need$ a63c8333ed1677c5 n parkind1
module radsurf_canopy_properties
use parkind1,only:jpim
use parkind1,only:jprb
integer(4),parameter::ntiletypes=6_4
integer(4),parameter::itileflat=0_4
integer(4),parameter::itileforest=1_4
integer(4),parameter::itileurban=2_4
integer(4),parameter::itilevegetatedurban=3_4
integer(4),parameter::itilesimpleurban=4_4
integer(4),parameter::itileinfinitestreet=5_4
character(*,1),parameter::tilerepresentationname(0_8:5_8)=[CHARACTER(KIND=1,LEN=14)::"Flat          ","Forest        ","Urban         ","VegetatedUrban","SimpleUrban   ","InfiniteStreet"]
type::canopy_properties_type
integer(4),allocatable::nlay(:)
integer(4),allocatable::istartlay(:)
real(8),allocatable::dz(:)
real(8),allocatable::cos_sza(:)
real(8),allocatable::ground_temperature(:)
real(8),allocatable::roof_temperature(:)
real(8),allocatable::wall_temperature(:)
real(8),allocatable::clear_air_temperature(:)
real(8),allocatable::veg_temperature(:)
real(8),allocatable::veg_air_temperature(:)
real(8),allocatable::building_fraction(:)
real(8),allocatable::veg_fraction(:)
real(8),allocatable::building_scale(:)
real(8),allocatable::veg_scale(:)
real(8),allocatable::veg_ext(:)
real(8),allocatable::veg_fsd(:)
real(8),allocatable::veg_contact_fraction(:)
integer(4),allocatable::i_representation(:)
integer(4)::ncol
integer(4)::ntotlay
contains
procedure::allocate=>allocate_canopy
procedure::deallocate=>deallocate_canopy
end type
contains
subroutine allocate_canopy(this,config,ncol,ntotlay,i_representation)
use radsurf_config,only:config_type
class(canopy_properties_type),intent(inout)::this
type(config_type),intent(in)::config
integer(4),intent(in)::ncol
integer(4),intent(in)::ntotlay
integer(4),intent(in),optional::i_representation(:)
end
subroutine deallocate_canopy(this)
class(canopy_properties_type),intent(inout)::this
end
subroutine print_surface_representation(i_representation)
integer(4),allocatable,intent(in)::i_representation(:)
end
end
