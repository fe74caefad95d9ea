!!! ssfunction_er0.f90 -- default link-time ssfunction_er (reference ssfunction_er0.f90:12-38): no early rejection
!!! inside the sum of squares, the plain ssfunction is called.  A user object defining ssfunction_er replaces it.
function ssfunction_er(theta,npar,ny,sscrit) result(ss)
  use mcmcmod, only : verbosity
  implicit none
  integer(kind=4), intent(in) :: npar, ny
  real(kind=8), intent(in) :: theta(npar), sscrit
  real(kind=8) :: ss(ny)
  logical, save :: once = .true.
  interface
     function ssfunction(theta,npar,ny)
       integer(kind=4) :: npar, ny
       real(kind=8) theta(npar)
       real(kind=8) ssfunction(ny)
     end function ssfunction
  end interface
  if (once) then
     if (verbosity > 0) write(*,*) 'note: the default er ssfunction, so no er for ss'
     once = .false.
  end if
  ss = ssfunction(theta, npar, ny)
end function ssfunction_er
