!!! initialize0.F90 -- the default link-time `initialize` (reference initialize.F90:41-119): initial values from
!!! files.  A user object that defines `initialize` replaces this archive member, like with libmcmcrun.a.
!!!   mcmcnycol.dat (optional)  number of sigma2 columns (the device engine supports 1)
!!!   parfile                   par0, any shape, flattened
!!!   cov0file                  npar x npar proposal covariance
!!!   covnfile (if named)       weight of cov0 -> initcmatn
!!!   sigma2file                row 1 sigma2, row 2 nobs; missing file: sigma2 = 1, nobs = 1
subroutine initialize(par0,npar,cmat0,initcmatn,sigma2,nobs,nycol)
  use mcmcprec
  use matutils, only : loaddata, loaddata2, doerror
  use mcmcmod, only : parfile, cov0file, covnfile, sigma2file
  implicit none
  integer, intent(inout) :: npar, initcmatn, nycol
  real(kind=dbl), intent(inout), allocatable :: par0(:), cmat0(:,:)
  real(kind=dbl), intent(inout), allocatable :: sigma2(:)
  integer, intent(inout), allocatable :: nobs(:)
  real(kind=dbl), pointer :: sm(:,:)
  real(kind=dbl) :: x
  integer :: stat

  call loaddata('mcmcnycol.dat', nycol, stat)
  if (stat /= 0) nycol = 1
  call loaddata2(parfile, par0)
  npar = size(par0)
  call loaddata2(cov0file, cmat0, stat, uselock=.true.)          ! initialize.F90:77
  if (stat /= 0) call doerror('Error reading file '//trim(cov0file))
  if (size(cmat0,1) /= npar .or. size(cmat0,2) /= npar) call doerror('Error reading file mcmccov.dat')
  if (len_trim(covnfile) > 0) then
     call loaddata(covnfile, x, stat, uselock=.true.)            ! initialize.F90:94
     if (stat == 0) initcmatn = int(x)
  end if
  allocate(sigma2(nycol), nobs(nycol))
  call loaddata(sigma2file, sm, stat)
  if (stat /= 0) then
     sigma2 = 1.0_dbl; nobs = 1
  else
     if (size(sm,1) < 2 .or. size(sm,2) < nycol) call doerror('error in mcmcsigma2.dat (obs: new format 4.8.2006)')
     sigma2 = sm(1,1:nycol); nobs = int(sm(2,1:nycol))
     deallocate(sm)
  end if
end subroutine initialize
