﻿!mod$ v1 sum:23a767086ebdf319
module radsurf_view_factor
contains
subroutine calc_view_factors_inf(height_width_ratio,view_ground_sky,view_wall_wall,cos_sza,view_dir_ground)
real(8),intent(in)::height_width_ratio
real(8),intent(out)::view_ground_sky
real(8),intent(out)::view_wall_wall
real(8),intent(in),optional::cos_sza
real(8),intent(out),optional::view_dir_ground
end
elemental subroutine calc_view_factors_exp(height_separation_ratio,view_ground_sky,view_wall_wall,cos_sza,view_dir_ground)
real(8),intent(in)::height_separation_ratio
real(8),intent(out)::view_ground_sky
real(8),intent(out)::view_wall_wall
real(8),intent(in),optional::cos_sza
real(8),intent(out),optional::view_dir_ground
end
end
