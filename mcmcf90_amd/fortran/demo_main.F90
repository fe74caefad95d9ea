!!! demo_main.F90 -- a driver program exactly like the reference's testcases/mcmcrun.F90:48-52:
!!! everything comes from mcmcinit.nml and the .dat files in the working directory.
program mcmcmain
  implicit none
  call mcmc_main()
end program mcmcmain
