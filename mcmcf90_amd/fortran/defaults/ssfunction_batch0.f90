!!! ssfunction_batch0.f90 -- default link-time ssfunction_batch (engine extension, &mcmcx hostbatch = 1): the
!!! candidates of all chains in one call, theta(npar,n) -> ss(ny,n).  This default is a loop over the user's
!!! ssfunction (external_inc.h:12-19); a user object defining ssfunction_batch replaces it, exactly like the other
!!! archive members (a vectorised / threaded / offloaded likelihood belongs there).
subroutine ssfunction_batch(theta, npar, n, ny, ss)
  implicit none
  integer(kind=4), intent(in) :: npar, n, ny
  real(kind=8), intent(in) :: theta(npar, n)
  real(kind=8), intent(out) :: ss(ny, n)
  integer :: i
  interface
     function ssfunction(theta,npar,ny)
       integer(kind=4) :: npar, ny
       real(kind=8) theta(npar)
       real(kind=8) ssfunction(ny)
     end function ssfunction
  end interface
  do i = 1, n
     ss(:, i) = ssfunction(theta(:, i), npar, ny)
  end do
end subroutine ssfunction_batch
