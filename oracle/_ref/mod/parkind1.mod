﻿!mod$ v1 sum:a63c8333ed1677c5
module parkind1
integer(4),parameter::jpit=1_4
intrinsic::selected_int_kind
integer(4),parameter::jpis=2_4
integer(4),parameter::jpim=4_4
integer(4),parameter::jpib=8_4
integer(4),parameter::jpia=4_4
integer(4),parameter::jprt=2_4
intrinsic::selected_real_kind
integer(4),parameter::jprs=4_4
integer(4),parameter::jprm=4_4
integer(4),parameter::jprd=8_4
integer(4),parameter::jprb=8_4
integer(4),parameter::jplm=4_4
end
