﻿!mod$ v1 sum:2d00fb526bfcacf9
!need$ a63c8333ed1677c5 n parkind1
module radtool_matrix
use parkind1,only:jprb
integer(4),parameter::imatrixpatterndense=0_4
integer(4),parameter::imatrixpatternshortwave=1_4
private::solve_vec_2
private::solve_mat_2
private::solve_vec_3
private::solve_mat_3
private::diag_mat_right_divide_3
private::lu_factorization
private::lu_substitution
private::solve_mat_n
interface fast_expm_exchange
procedure::fast_expm_exchange_2
procedure::fast_expm_exchange_3
end interface
contains
function mat_x_vec(n,iend,m,a,b,do_top_left_only_in)
integer(4),intent(in)::n
integer(4),intent(in)::iend
integer(4),intent(in)::m
real(8),intent(in)::a(:,:,:)
real(8),intent(in)::b(:,:)
logical(4),intent(in),optional::do_top_left_only_in
real(8)::mat_x_vec(1_8:int(iend,kind=8),1_8:int(m,kind=8))
end
function rect_mat_x_vec(n,k,l,a,b)
integer(4),intent(in)::n
integer(4),intent(in)::k
integer(4),intent(in)::l
real(8),intent(in)::a(1_8:int(n,kind=8),1_8:int(k,kind=8),1_8:int(l,kind=8))
real(8),intent(in)::b(1_8:int(n,kind=8),1_8:int(l,kind=8))
real(8)::rect_mat_x_vec(1_8:int(n,kind=8),1_8:int(k,kind=8))
end
function singlemat_x_vec(n,iend,m,a,b)
integer(4),intent(in)::n
integer(4),intent(in)::iend
integer(4),intent(in)::m
real(8),intent(in)::a(1_8:int(m,kind=8),1_8:int(m,kind=8))
real(8),intent(in)::b(:,:)
real(8)::singlemat_x_vec(1_8:int(iend,kind=8),1_8:int(m,kind=8))
end
function rect_singlemat_x_vec(n,m,k,a,b)
integer(4),intent(in)::n
integer(4),intent(in)::m
integer(4),intent(in)::k
real(8),intent(in)::a(1_8:int(m,kind=8),1_8:int(k,kind=8))
real(8),intent(in)::b(:,:)
real(8)::rect_singlemat_x_vec(1_8:int(n,kind=8),1_8:int(m,kind=8))
end
function mat_x_mat(n,iend,m,a,b,i_matrix_pattern)
integer(4),intent(in)::n
integer(4),intent(in)::iend
integer(4),intent(in)::m
real(8),intent(in)::a(:,:,:)
real(8),intent(in)::b(:,:,:)
integer(4),intent(in),optional::i_matrix_pattern
real(8)::mat_x_mat(1_8:int(iend,kind=8),1_8:int(m,kind=8),1_8:int(m,kind=8))
end
function rect_mat_x_mat(n,k,l,m,a,b)
integer(4),intent(in)::n
integer(4),intent(in)::k
integer(4),intent(in)::l
integer(4),intent(in)::m
real(8),intent(in)::a(1_8:int(n,kind=8),1_8:int(k,kind=8),1_8:int(l,kind=8))
real(8),intent(in)::b(1_8:int(n,kind=8),1_8:int(l,kind=8),1_8:int(m,kind=8))
real(8)::rect_mat_x_mat(1_8:int(n,kind=8),1_8:int(k,kind=8),1_8:int(m,kind=8))
end
function singlemat_x_mat(n,iend,m,a,b)
integer(4),intent(in)::n
integer(4),intent(in)::iend
integer(4),intent(in)::m
real(8),intent(in)::a(1_8:int(m,kind=8),1_8:int(m,kind=8))
real(8),intent(in)::b(:,:,:)
real(8)::singlemat_x_mat(1_8:int(iend,kind=8),1_8:int(m,kind=8),1_8:int(m,kind=8))
end
function mat_x_singlemat(n,iend,m,a,b)
integer(4),intent(in)::n
integer(4),intent(in)::iend
integer(4),intent(in)::m
real(8),intent(in)::a(:,:,:)
real(8),intent(in)::b(1_8:int(m,kind=8),1_8:int(m,kind=8))
real(8)::mat_x_singlemat(1_8:int(iend,kind=8),1_8:int(m,kind=8),1_8:int(m,kind=8))
end
function rect_mat_x_singlemat(n,m,o,p,a,b)
integer(4),intent(in)::n
integer(4),intent(in)::m
integer(4),intent(in)::o
integer(4),intent(in)::p
real(8),intent(in)::a(:,:,:)
real(8),intent(in)::b(1_8:int(o,kind=8),1_8:int(p,kind=8))
real(8)::rect_mat_x_singlemat(1_8:int(n,kind=8),1_8:int(m,kind=8),1_8:int(p,kind=8))
end
function rect_expandedmat_x_mat(n,m,o,s,p,a,b)
integer(4),intent(in)::n
integer(4),intent(in)::m
integer(4),intent(in)::o
integer(4),intent(in)::s
integer(4),intent(in)::p
real(8),intent(in)::a(1_8:int(m,kind=8),1_8:int(o,kind=8))
real(8),intent(in)::b(:,:,:)
real(8)::rect_expandedmat_x_mat(1_8:int(n,kind=8),1_8:int(m*s,kind=8),1_8:int(p,kind=8))
end
function rect_mat_x_expandedmat(n,m,o,s,p,a,b)
integer(4),intent(in)::n
integer(4),intent(in)::m
integer(4),intent(in)::o
integer(4),intent(in)::s
integer(4),intent(in)::p
real(8),intent(in)::a(:,:,:)
real(8),intent(in)::b(1_8:int(m,kind=8),1_8:int(o,kind=8))
real(8)::rect_mat_x_expandedmat(1_8:int(n,kind=8),1_8:int(p,kind=8),1_8:int(o*s,kind=8))
end
function rect_expandedmat_x_vec(n,m,k,s,a,b)
integer(4),intent(in)::n
integer(4),intent(in)::m
integer(4),intent(in)::k
integer(4),intent(in)::s
real(8),intent(in)::a(1_8:int(m,kind=8),1_8:int(k,kind=8))
real(8),intent(in)::b(:,:)
real(8)::rect_expandedmat_x_vec(1_8:int(n,kind=8),1_8:int(m*s,kind=8))
end
function identity_minus_mat_x_mat(n,iend,m,a,b,i_matrix_pattern)
integer(4),intent(in)::n
integer(4),intent(in)::iend
integer(4),intent(in)::m
real(8),intent(in)::a(:,:,:)
real(8),intent(in)::b(:,:,:)
integer(4),intent(in),optional::i_matrix_pattern
real(8)::identity_minus_mat_x_mat(1_8:int(iend,kind=8),1_8:int(m,kind=8),1_8:int(m,kind=8))
end
function repeated_square(m,a,nrepeat,i_matrix_pattern)
integer(4),intent(in)::m
real(8),intent(inout)::a(1_8:int(m,kind=8),1_8:int(m,kind=8))
integer(4),intent(in)::nrepeat
integer(4),intent(in),optional::i_matrix_pattern
real(8)::repeated_square(1_8:int(m,kind=8),1_8:int(m,kind=8))
end
pure subroutine solve_vec_2(n,iend,a,b,x)
integer(4),intent(in)::n
integer(4),intent(in)::iend
real(8),intent(in)::a(:,:,:)
real(8),intent(in)::b(:,:)
real(8),intent(out)::x(:,:)
end
pure subroutine solve_mat_2(n,iend,a,b,x)
integer(4),intent(in)::n
integer(4),intent(in)::iend
real(8),intent(in)::a(:,:,:)
real(8),intent(in)::b(:,:,:)
real(8),intent(out)::x(:,:,:)
end
pure subroutine solve_vec_3(n,iend,a,b,x)
integer(4),intent(in)::n
integer(4),intent(in)::iend
real(8),intent(in)::a(:,:,:)
real(8),intent(in)::b(:,:)
real(8),intent(out)::x(:,:)
end
pure subroutine solve_mat_3(n,iend,a,b,x)
integer(4),intent(in)::n
integer(4),intent(in)::iend
real(8),intent(in)::a(:,:,:)
real(8),intent(in)::b(:,:,:)
real(8),intent(out)::x(:,:,:)
end
pure subroutine diag_mat_right_divide_3(n,iend,a,b,x)
integer(4),intent(in)::n
integer(4),intent(in)::iend
real(8),intent(in)::a(1_8:int(iend,kind=8),1_8:3_8,1_8:3_8)
real(8),intent(in)::b(1_8:int(iend,kind=8),1_8:3_8)
real(8),intent(out)::x(1_8:int(n,kind=8),1_8:3_8,1_8:3_8)
end
pure subroutine lu_factorization(n,iend,m,a,lu)
integer(4),intent(in)::n
integer(4),intent(in)::iend
integer(4),intent(in)::m
real(8),intent(in)::a(:,:,:)
real(8),intent(out)::lu(1_8:int(iend,kind=8),1_8:int(m,kind=8),1_8:int(m,kind=8))
end
pure subroutine lu_substitution(n,iend,m,lu,b,x)
integer(4),intent(in)::n
integer(4),intent(in)::iend
integer(4),intent(in)::m
real(8),intent(in)::lu(1_8:int(iend,kind=8),1_8:int(m,kind=8),1_8:int(m,kind=8))
real(8),intent(in)::b(:,:)
real(8),intent(out)::x(1_8:int(iend,kind=8),1_8:int(m,kind=8))
end
pure subroutine lu_invert(n,iend,m,lu,x)
integer(4),intent(in)::n
integer(4),intent(in)::iend
integer(4),intent(in)::m
real(8),intent(in)::lu(1_8:int(iend,kind=8),1_8:int(m,kind=8),1_8:int(m,kind=8))
real(8),intent(out)::x(1_8:int(iend,kind=8),1_8:int(m,kind=8),1_8:int(m,kind=8))
end
pure subroutine solve_mat_n(n,iend,m,a,b,x)
integer(4),intent(in)::n
integer(4),intent(in)::iend
integer(4),intent(in)::m
real(8),intent(in)::a(:,:,:)
real(8),intent(in)::b(:,:,:)
real(8),intent(out)::x(1_8:int(iend,kind=8),1_8:int(m,kind=8),1_8:int(m,kind=8))
end
function solve_rect_mat(n,m,l,a,b)
integer(4),intent(in)::n
integer(4),intent(in)::m
integer(4),intent(in)::l
real(8),intent(in)::a(1_8:int(n,kind=8),1_8:int(m,kind=8),1_8:int(m,kind=8))
real(8),intent(in)::b(1_8:int(n,kind=8),1_8:int(m,kind=8),1_8:int(l,kind=8))
real(8)::solve_rect_mat(1_8:int(n,kind=8),1_8:int(m,kind=8),1_8:int(l,kind=8))
end
function solve_vec(n,iend,m,a,b)
integer(4),intent(in)::n
integer(4),intent(in)::iend
integer(4),intent(in)::m
real(8),intent(in)::a(:,:,:)
real(8),intent(in)::b(:,:)
real(8)::solve_vec(1_8:int(iend,kind=8),1_8:int(m,kind=8))
end
function solve_mat(n,iend,m,a,b)
integer(4),intent(in)::n
integer(4),intent(in)::iend
integer(4),intent(in)::m
real(8),intent(in)::a(:,:,:)
real(8),intent(in)::b(:,:,:)
real(8)::solve_mat(1_8:int(iend,kind=8),1_8:int(m,kind=8),1_8:int(m,kind=8))
end
function invert(n,iend,m,a)
integer(4),intent(in)::n
integer(4),intent(in)::iend
integer(4),intent(in)::m
real(8),intent(in)::a(:,:,:)
real(8)::invert(1_8:int(iend,kind=8),1_8:int(m,kind=8),1_8:int(m,kind=8))
end
subroutine expm(n,iend,m,a,i_matrix_pattern)
integer(4),intent(in)::n
integer(4),intent(in)::iend
integer(4),intent(in)::m
real(8),intent(inout)::a(1_8:int(n,kind=8),1_8:int(m,kind=8),1_8:int(m,kind=8))
integer(4),intent(in)::i_matrix_pattern
end
subroutine fast_expm_exchange_2(n,iend,a,b,r)
integer(4),intent(in)::n
integer(4),intent(in)::iend
real(8),intent(in)::a(1_8:int(n,kind=8))
real(8),intent(in)::b(1_8:int(n,kind=8))
real(8),intent(out)::r(1_8:int(n,kind=8),1_8:2_8,1_8:2_8)
end
subroutine fast_expm_exchange_3(n,iend,a,b,c,d,r)
integer(4),intent(in)::n
integer(4),intent(in)::iend
real(8),intent(in)::a(1_8:int(n,kind=8))
real(8),intent(in)::b(1_8:int(n,kind=8))
real(8),intent(in)::c(1_8:int(n,kind=8))
real(8),intent(in)::d(1_8:int(n,kind=8))
real(8),intent(out)::r(1_8:int(n,kind=8),1_8:3_8,1_8:3_8)
end
end
