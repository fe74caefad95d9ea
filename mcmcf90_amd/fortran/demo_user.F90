!!! demo_user.F90 -- a complete user program in the style of the reference's testcases/mcmcrun.F90:
!!! main program + the user's own ssfunction (reads data.dat on first call) + checkbounds (all
!!! parameters positive).  Linked against libmcmcxf.a + libmcmcx.so instead of libmcmcrun.a; nothing in
!!! this file knows about the GPU.
program mcmcmain
  implicit none
  call mcmc_main()
end program mcmcmain

function ssfunction(theta,npar,ny) result(ss)
  implicit none
  integer*4 npar, ny
  real*8 theta(npar)
  real*8 ss(ny)
  real*8, save :: x(1000), y(1000)
  integer, save :: n = 0
  logical, save :: first = .true.
  character(len=256) :: line
  integer :: ios
  if (first) then
     open(77, file='data.dat', status='old')
     do
        read(77, '(A)', iostat=ios) line
        if (ios /= 0) exit
        line = adjustl(line)
        if (len_trim(line) == 0 .or. line(1:1) == '%' .or. line(1:1) == '#') cycle
        n = n + 1
        read(line, *) x(n), y(n)
     end do
     close(77)
     first = .false.
  end if
  ss(1) = sum((y(1:n) - theta(1)*exp(-theta(2)*x(1:n)))**2)
end function ssfunction

function checkbounds(theta)
  implicit none
  real*8 theta(:)
  logical checkbounds
  checkbounds = .true.
  if (any(theta <= 0.0d0)) checkbounds = .false.
end function checkbounds
