!!! Default priorfun: independent Gaussian priors read once from `priorsfile` of namelist /mcmc/
!!! (two rows: means, standard deviations; sigma <= 0 = flat), -2 log p(theta) = sum(((theta-mu)/sig)**2).
!!! Same behaviour as priorfun.f90:31-103 of the reference; empty priorsfile = no prior.
function priorfun(theta,len) result(priss)
  use mcmcmod, only : priorsfile, loadnumbers
  implicit none
  real*8 priss
  integer*4 len
  real*8 theta(len)
  real*8, allocatable, save :: pmus(:), psig(:)
  logical, save :: first = .true.
  real*8, allocatable :: v(:)
  integer :: nr, nc, stat, i
  priss = 0.0d0
  if (len_trim(priorsfile) <= 0) return
  if (first) then
     call loadnumbers(priorsfile, v, nr, nc, stat)
     if (stat /= 0) then
        write(*,*) 'could not open file ', trim(priorsfile)
        stop 1
     end if
     if (size(v) /= 2*len) then
        write(*,*) 'priors.dat should have  2*npar elements'
        stop 1
     end if
     allocate(pmus(len), psig(len))
     pmus = v(1:len); psig = v(len+1:2*len)
     first = .false.
  end if
  do i = 1, len
     if (psig(i) > 0.0d0) priss = priss + ((theta(i)-pmus(i))/psig(i))**2
  end do
end function priorfun
