﻿!mod$ v1 sum:7b824a3df9d4df01
!need$ a63c8333ed1677c5 n parkind1
module radsurf_boundary_conds_out
use parkind1,only:jprb
type::boundary_conds_out_type
real(8),allocatable::sw_albedo(:,:)
real(8),allocatable::sw_albedo_dir(:,:)
real(8),allocatable::lw_emissivity(:,:)
real(8),allocatable::lw_emission(:,:)
contains
procedure::allocate=>allocate_boundary_conds_out
procedure::deallocate=>deallocate_boundary_conds_out
end type
contains
subroutine allocate_boundary_conds_out(this,ncol,nsw,nlw)
class(boundary_conds_out_type),intent(inout)::this
integer(4),intent(in)::ncol
integer(4),intent(in)::nsw
integer(4),intent(in)::nlw
end
subroutine deallocate_boundary_conds_out(this)
class(boundary_conds_out_type),intent(inout)::this
end
end
