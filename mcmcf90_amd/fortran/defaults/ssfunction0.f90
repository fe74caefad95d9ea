!!! Default ssfunction: the user must supply one (same behaviour as ssfunction0.f90:10-14 of the reference).
!!! Archive member of libmcmcxf.a, overridden by a user object earlier on the link line.
function ssfunction(theta,npar,ny) result(ss)
  implicit none
  integer*4 npar, ny
  real*8 theta(npar)
  real*8 ss(ny)
  write(*,*) 'ERROR: no ssfunction: write your own and link it before the library'
  ss = 0.0d0
  stop 1
end function ssfunction
