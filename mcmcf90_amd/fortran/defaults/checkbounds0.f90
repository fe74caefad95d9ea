!!! Default checkbounds: everything is in bounds (checkbounds0.f90:10-15 of the reference).
function checkbounds(theta)
  implicit none
  real*8 theta(:)
  logical checkbounds
  logical, save :: first = .true.
  if (first) then
     write(*,*) 'note: using the default checkbounds (no bounds)'
     first = .false.
  end if
  checkbounds = .true.
end function checkbounds
