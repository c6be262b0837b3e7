﻿!mod$ v1 sum:4cc8b62dc649a088
module radtool_schur
contains
subroutine schur_invert_sw(nmat,n0,n1,g0,g1,g2,g3,g0i,g1i,g2i,g3i)
integer(4),intent(in)::nmat
integer(4),intent(in)::n0
integer(4),intent(in)::n1
real(8),intent(in)::g0(1_8:int(nmat,kind=8),1_8:int(n0,kind=8),1_8:int(n0,kind=8))
real(8),intent(in)::g1(1_8:int(nmat,kind=8),1_8:int(n1,kind=8),1_8:int(n1,kind=8))
real(8),intent(in)::g2(1_8:int(nmat,kind=8),1_8:int(n1,kind=8),1_8:int(n1,kind=8))
real(8),intent(in)::g3(1_8:int(nmat,kind=8),1_8:int(n1,kind=8),1_8:int(n0,kind=8))
real(8),intent(out)::g0i(1_8:int(nmat,kind=8),1_8:int(n0,kind=8),1_8:int(n0,kind=8))
real(8),intent(out)::g1i(1_8:int(nmat,kind=8),1_8:int(n1,kind=8),1_8:int(n1,kind=8))
real(8),intent(out)::g2i(1_8:int(nmat,kind=8),1_8:int(n1,kind=8),1_8:int(n1,kind=8))
real(8),intent(out)::g3i(1_8:int(nmat,kind=8),1_8:int(n1,kind=8),1_8:int(n0,kind=8))
end
end
