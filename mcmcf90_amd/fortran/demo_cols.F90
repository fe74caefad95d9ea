!!! demo_cols.F90 -- a user program with TWO response columns (nycol = 2): ssfunction returns one sum of squares per
!!! column, MCMC_setsigma2nobs gets vectors, and the library keeps one error variance per column (mcmc.F90:30-33,
!!! MCMC_DRAM.F90:100-118, 192-206).  Model: column j of data2.dat (x, y1, y2) is theta(1)*exp(-theta(1+j)*x).
program mcmcmain
  use mcmcprec
  use mcmcmod, only : MCMC_setpar0, MCMC_setsigma2nobs, MCMC_setcmat0
  implicit none
  real(kind=dbl) :: c0(3,3)
  call MCMC_setpar0((/9.0_dbl, 0.1_dbl, 0.2_dbl/))
  c0 = 0.0_dbl
  c0(1,1) = 0.02_dbl; c0(2,2) = 0.0001_dbl; c0(3,3) = 0.0002_dbl
  call MCMC_setcmat0(c0)
  call MCMC_setsigma2nobs((/0.5_dbl, 0.3_dbl/), (/11, 13/))
  call mcmc_main()
end program mcmcmain

function ssfunction(theta,npar,ny) result(ss)
  use mcmcprec
  use matutils, only : loaddata
  implicit none
  integer(kind=ik4) :: npar, ny
  real(kind=dbl) :: theta(npar)
  real(kind=dbl) :: ss(ny)
  real(kind=dbl), save, pointer :: data(:,:)
  logical, save :: first = .true.
  integer :: j
  if (first) then
     call loaddata('data2.dat', data)
     first = .false.
  end if
  do j = 1, ny
     ss(j) = sum((data(:,1+j) - theta(1)*exp(-theta(1+j)*data(:,1)))**2)
  end do
end function ssfunction

function checkbounds(theta)
  implicit none
  real*8 theta(:)
  logical checkbounds
  checkbounds = all(theta > 0.0d0)
end function checkbounds
