﻿!mod$ v1 sum:be8b8ffc7bcf2bf1
module yomhook
logical(4)::lhook
contains
subroutine dr_hook(proc_name,iswitch,proc_key)
character(*,1),intent(in)::proc_name
integer(4),intent(in)::iswitch
real(8),intent(inout)::proc_key
end
end
