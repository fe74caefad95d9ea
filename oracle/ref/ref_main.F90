!!! oracle/ref/ref_main.F90 -- TEST INFRASTRUCTURE. Not part of the product path.
!!!
!!! Driver program linked against the real mcmcf90 library built from
!!! /root/reference (oracle/Makefile).  Same shape as the reference's own example
!!! programs (testcases/mcmcrun.F90:48-52, mcmcrun4.F90:5-15): a main that calls
!!! mcmc_main() plus the user callbacks ssfunction and checkbounds
!!! (external_inc.h:12-33), which forward to oracle/ref/user_target.c.
program mcxref
  implicit none
#ifdef MCX_MAIN_ONE
  call mcmc_main_one()      ! one evaluation per invocation: MCMC_run1 / MCMC_run1_er (mcmc_main.F90:49-70)
#else
  call mcmc_main()
#endif
end program mcxref

function ssfunction(theta,npar,ny) result(ss)
  use iso_c_binding
  implicit none
  integer(4) :: npar, ny
  real(8) :: theta(npar)
  real(8) :: ss(ny)
  interface
     subroutine mcxref_ss_cols(theta,npar,ny,ss) bind(C,name='mcxref_ss_cols')
       use iso_c_binding
       real(c_double) :: theta(*), ss(*)
       integer(c_int), value :: npar, ny
     end subroutine mcxref_ss_cols
  end interface
  call mcxref_ss_cols(theta,npar,ny,ss)
end function ssfunction

function checkbounds(theta)
  use iso_c_binding
  implicit none
  real(8) :: theta(:)
  logical :: checkbounds
  real(8) :: t(size(theta))
  interface
     function mcxref_inbounds(theta,npar) bind(C,name='mcxref_inbounds') result(v)
       use iso_c_binding
       real(c_double) :: theta(*)
       integer(c_int), value :: npar
       integer(c_int) :: v
     end function mcxref_inbounds
  end interface
  t = theta
  checkbounds = (mcxref_inbounds(t,size(t)) /= 0)
end function checkbounds
