﻿!mod$ v1 sum:16f862c6ea0067f6
module radsurf_overlap
contains
pure function calc_overlap_matrix_max_ran(nreg,f_upper,f_lower) result(overlap_matrix)
integer(4),intent(in)::nreg
real(8),intent(in)::f_upper
real(8),intent(in)::f_lower
real(8)::overlap_matrix(1_8:int(nreg,kind=8),1_8:int(nreg,kind=8))
end
subroutine calc_overlap_matrices(nlay,nreg,region_fracs,u_overlap,v_overlap,frac_threshold)
integer(4),intent(in)::nlay
integer(4),intent(in)::nreg
real(8),intent(in)::region_fracs(1_8:int(nreg,kind=8),1_8:int(nlay,kind=8))
real(8),intent(out)::u_overlap(1_8:int(nreg,kind=8),1_8:int(nreg,kind=8),1_8:int(nlay+1_4,kind=8))
real(8),intent(out)::v_overlap(1_8:int(nreg,kind=8),1_8:int(nreg,kind=8),1_8:int(nlay+1_4,kind=8))
real(8),intent(in),optional::frac_threshold
end
pure function calc_overlap_matrix_max_ran_urban(nreg,frac_upper,frac_lower) result(overlap_matrix)
integer(4),intent(in)::nreg
real(8),intent(in)::frac_upper(1_8:int(nreg,kind=8))
real(8),intent(in)::frac_lower(1_8:int(nreg+1_4,kind=8))
real(8)::overlap_matrix(1_8:int(nreg,kind=8),1_8:int(nreg+1_4,kind=8))
end
subroutine calc_overlap_matrices_urban(nlay,nreg,region_fracs,u_overlap,v_overlap,frac_threshold)
integer(4),intent(in)::nlay
integer(4),intent(in)::nreg
real(8),intent(in)::region_fracs(1_8:int(nreg,kind=8),1_8:int(nlay,kind=8))
real(8),intent(out)::u_overlap(1_8:int(nreg,kind=8),1_8:int(nreg+1_4,kind=8),1_8:int(nlay+1_4,kind=8))
real(8),intent(out)::v_overlap(1_8:int(nreg+1_4,kind=8),1_8:int(nreg,kind=8),1_8:int(nlay+1_4,kind=8))
real(8),intent(in),optional::frac_threshold
end
end
