﻿!mod$ v1 sum:7e14fb4b785524cf
!need$ a63c8333ed1677c5 n parkind1
module radtool_legendre_gauss
use parkind1,only:jpim
use parkind1,only:jprb
type::legendre_gauss_type
real(8),allocatable::mu(:)
real(8),allocatable::sin_ang(:)
real(8),allocatable::tan_ang(:)
real(8),allocatable::weight(:)
real(8),allocatable::hweight(:)
real(8),allocatable::vweight(:)
real(8)::vadjustment
real(8)::vadjustment2
integer(4)::nstream=0_4
contains
procedure::initialize=>initialize_legendre_gauss
procedure::deallocate=>deallocate_legendre_gauss
end type
contains
subroutine initialize_legendre_gauss(this,nstream)
class(legendre_gauss_type),intent(inout)::this
integer(4),intent(in)::nstream
end
subroutine deallocate_legendre_gauss(this)
class(legendre_gauss_type),intent(inout)::this
end
subroutine calc_legendre_gauss(nnode,x1,x2,xnode,weight)
integer(4),intent(in)::nnode
real(8),intent(in)::x1
real(8),intent(in)::x2
real(8),intent(out)::xnode(1_8:int(nnode,kind=8))
real(8),intent(out)::weight(1_8:int(nnode,kind=8))
end
end
