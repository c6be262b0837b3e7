﻿!mod$ v1 sum:2bf14d589c89d1f2
module radtool_eigen_decomposition
contains
subroutine eigen_decomposition_real(norder,nmat,amat,eigenvalue,eigenvector,nerror,ierror)
integer(4),intent(in)::norder
integer(4),intent(in)::nmat
real(8),intent(in)::amat(1_8:int(nmat,kind=8),1_8:int(norder,kind=8),1_8:int(norder,kind=8))
real(8),intent(out)::eigenvalue(1_8:int(nmat,kind=8),1_8:int(norder,kind=8))
real(8),intent(out)::eigenvector(1_8:int(nmat,kind=8),1_8:int(norder,kind=8),1_8:int(norder,kind=8))
integer(4),intent(out),optional::nerror
integer(4),intent(out),optional::ierror(1_8:int(nmat,kind=8))
end
end
