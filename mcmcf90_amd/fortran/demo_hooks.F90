!!! demo_hooks.F90 -- a user program that overrides every optional link-time hook of the library:
!!! its own `initialize` (no input files), `dump_init` / `dump` / `dump_end` (a trace file written every
!!! dumpint iterations), next to ssfunction and checkbounds.  Same model as testcases/mcmcrun.F90.
program mcmcmain
  implicit none
  call mcmc_main()
end program mcmcmain

subroutine initialize(par0,npar,cmat0,initcmatn,sigma2,nobs,nycol)
  use mcmcprec
  implicit none
  integer, intent(inout) :: npar, initcmatn, nycol
  real(kind=dbl), intent(inout), allocatable :: par0(:), cmat0(:,:)
  real(kind=dbl), intent(inout), allocatable :: sigma2(:)
  integer, intent(inout), allocatable :: nobs(:)
  npar = 2; nycol = 1
  allocate(par0(2), cmat0(2,2), sigma2(1), nobs(1))
  par0 = (/10.0_dbl, 0.1_dbl/)
  cmat0 = reshape((/0.2_dbl, 0.0_dbl, 0.0_dbl, 0.001_dbl/), (/2,2/))
  sigma2 = 0.5_dbl; nobs = 11
end subroutine initialize

subroutine dump_init()
  implicit none
  open(91, file='dump_trace.dat', status='replace')
  write(91,'(A)') '# dump_init'
end subroutine dump_init

subroutine dump(oldpar)
  use mcmcprec
  implicit none
  real(kind=dbl), intent(in) :: oldpar(:)
  write(91,'(*(ES24.16E3,1X))') oldpar
end subroutine dump

subroutine dump_end()
  implicit none
  write(91,'(A)') '# dump_end'
  close(91)
end subroutine dump_end

function ssfunction(theta,npar,ny) result(ss)
  use mcmcprec
  use matutils, only : loaddata
  implicit none
  integer(kind=ik4) :: npar, ny
  real(kind=dbl) :: theta(npar)
  real(kind=dbl) :: ss(ny)
  real(kind=dbl), save, pointer :: data(:,:)
  logical, save :: first = .true.
  if (first) then
     call loaddata('data.dat', data)
     first = .false.
  end if
  ss(1) = sum((data(:,2) - theta(1)*exp(-theta(2)*data(:,1)))**2)
end function ssfunction

function checkbounds(theta)
  implicit none
  real*8 theta(:)
  logical checkbounds
  checkbounds = all(theta > 0.0d0)
end function checkbounds
