﻿!mod$ v1 sum:9749695ac68ebc76
module print_matrix_mod
contains
subroutine print_vector(name,vec)
character(*,1),intent(in)::name
real(8),intent(in)::vec(:)
end
subroutine print_matrix(name,mat)
character(*,1),intent(in)::name
real(8),intent(in)::mat(:,:)
end
subroutine print_array3(name,mat)
character(*,1),intent(in)::name
real(8),intent(in)::mat(:,:,:)
end
end
