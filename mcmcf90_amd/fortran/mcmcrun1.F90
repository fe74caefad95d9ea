!!! mcmcrun1.F90 -- the control variables MCMC_run1 / MCMC_run1_er carry from one program invocation to the next
!!! (reference mcmcrun1.F90:11-74: module mcmcrun1, namelist /mcmcrun/ in the file mcmcrun.nml).  Same module name, public
!!! variables and three subroutines, so a driver that prepares or inspects mcmcrun.nml through the module keeps its source.
module mcmcrun1
  use mcmcprec
  implicit none
  public
  real(kind=dbl) :: alpha12, sscrit                  ! last acceptance probability; early rejection's threshold
  integer :: drstage, isimu, nrej, ieval             ! DR stage, chain index so far, tries since the last move, evaluations so far
  namelist /mcmcrun/ drstage, isimu, ieval, nrej, alpha12, sscrit
contains

  subroutine init_mcmcrun_namelist()                 ! mcmcrun1.F90:27-37
    drstage = 1; isimu = 1; nrej = 0; ieval = 0
    alpha12 = 0.0_dbl; sscrit = -1.0_dbl
  end subroutine init_mcmcrun_namelist

  subroutine read_mcmcrun_namelist()                 ! mcmcrun1.F90:40-60: a missing or unreadable file stops the program
    integer :: u, ios
    open(newunit=u, file='mcmcrun.nml', status='old', iostat=ios)
    if (ios /= 0) then
       write(*,*) 'ERROR: File mcmcrun.nml not found'
       stop
    end if
    read(u, nml=mcmcrun, iostat=ios)
    close(u)
    if (ios /= 0) then
       write(*,*) 'ERROR: Error reading parameters namelist from file mcmcrun.nml'
       write(*,*) '   status:', ios
       stop
    end if
  end subroutine read_mcmcrun_namelist

  subroutine write_mcmcrun_namelist()                ! mcmcrun1.F90:64-72
    integer :: u, ios
    open(newunit=u, file='mcmcrun.nml', status='replace', iostat=ios)
    if (ios /= 0) return
    write(u, nml=mcmcrun)
    close(u)
  end subroutine write_mcmcrun_namelist

end module mcmcrun1
