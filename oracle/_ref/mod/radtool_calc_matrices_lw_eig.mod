﻿!mod$ v1 sum:8199304d9d2ab4cc
module radtool_calc_matrices_lw_eig
contains
subroutine calc_matrices_lw_eig(nmat,norder,dz,gamma1,gamma2,emiss_rate,reflectance,transmittance,source,int_flux,int_flux_source)
integer(4),intent(in)::nmat
integer(4),intent(in)::norder
real(8),intent(in)::dz
real(8),intent(in)::gamma1(1_8:int(nmat,kind=8),1_8:int(norder,kind=8),1_8:int(norder,kind=8))
real(8),intent(in)::gamma2(1_8:int(nmat,kind=8),1_8:int(norder,kind=8),1_8:int(norder,kind=8))
real(8),intent(in)::emiss_rate(1_8:int(nmat,kind=8),1_8:int(norder,kind=8))
real(8),intent(out)::reflectance(1_8:int(nmat,kind=8),1_8:int(norder,kind=8),1_8:int(norder,kind=8))
real(8),intent(out)::transmittance(1_8:int(nmat,kind=8),1_8:int(norder,kind=8),1_8:int(norder,kind=8))
real(8),intent(out)::source(1_8:int(nmat,kind=8),1_8:int(norder,kind=8))
real(8),intent(out),optional::int_flux(1_8:int(nmat,kind=8),1_8:int(norder,kind=8),1_8:int(norder,kind=8))
real(8),intent(out),optional::int_flux_source(1_8:int(nmat,kind=8),1_8:int(norder,kind=8))
end
end
