﻿!mod$ v1 sum:42cc4aa19b898f19
module radtool_calc_matrices_sw_eig
contains
subroutine calc_matrices_sw_eig(nmat,ndiff,ndir,dz,mu0,gamma0,gamma1,gamma2,gamma3,reflectance,transmittance,s_up,s_dn,trans_dir,int_dir,int_diff,int_dir_diff)
integer(4),intent(in)::nmat
integer(4),intent(in)::ndiff
integer(4),intent(in)::ndir
real(8),intent(in)::dz
real(8),intent(in)::mu0
real(8),intent(in)::gamma0(1_8:int(nmat,kind=8),1_8:int(ndir,kind=8),1_8:int(ndir,kind=8))
real(8),intent(in)::gamma1(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndiff,kind=8))
real(8),intent(in)::gamma2(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndiff,kind=8))
real(8),intent(in)::gamma3(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndir,kind=8))
real(8),intent(out)::reflectance(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndiff,kind=8))
real(8),intent(out)::transmittance(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndiff,kind=8))
real(8),intent(out)::s_up(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndir,kind=8))
real(8),intent(out)::s_dn(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndir,kind=8))
real(8),intent(out)::trans_dir(1_8:int(nmat,kind=8),1_8:int(ndir,kind=8),1_8:int(ndir,kind=8))
real(8),intent(out),optional::int_dir(1_8:int(nmat,kind=8),1_8:int(ndir,kind=8),1_8:int(ndir,kind=8))
real(8),intent(out),optional::int_diff(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndiff,kind=8))
real(8),intent(out),optional::int_dir_diff(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndir,kind=8))
end
subroutine direct_diffuse_part(nmat,ndiff,ndir,mu0,exp_lambda_dz,exp_eigenval_dir_dz,g0,g1,g2,g3,g4,s_up,s_dn)
integer(4),intent(in)::nmat
integer(4),intent(in)::ndiff
integer(4),intent(in)::ndir
real(8),intent(in)::mu0
real(8),intent(in)::exp_lambda_dz(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8))
real(8),intent(in)::exp_eigenval_dir_dz(1_8:int(nmat,kind=8),1_8:int(ndir,kind=8))
real(8),intent(in)::g0(1_8:int(nmat,kind=8),1_8:int(ndir,kind=8),1_8:int(ndir,kind=8))
real(8),intent(in)::g1(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndiff,kind=8))
real(8),intent(in)::g2(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndiff,kind=8))
real(8),intent(in)::g3(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndir,kind=8))
real(8),intent(in)::g4(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndir,kind=8))
real(8),intent(out)::s_up(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndir,kind=8))
real(8),intent(out)::s_dn(1_8:int(nmat,kind=8),1_8:int(ndiff,kind=8),1_8:int(ndir,kind=8))
end
end
