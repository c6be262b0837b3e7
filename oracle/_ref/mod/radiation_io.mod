﻿!mod$ v1 sum:660b02b068a15f16
!need$ a63c8333ed1677c5 n parkind1
module radiation_io
use parkind1,only:jpim
integer(4),parameter::nulout=6_4
integer(4),parameter::nulerr=0_4
integer(4),parameter::nulrad=25_4
contains
subroutine radiation_abort(text)
character(*,1),intent(in),optional::text
end
end
