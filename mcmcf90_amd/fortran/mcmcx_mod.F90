!!! mcmcx_mod.F90 -- Fortran host side of the MI355X engine: the same user surface as mcmcf90.
!!!
!!! An existing driver program keeps its source:
!!!
!!!     use mcmcmod, only : MCMC_setpar0, MCMC_setcmat0, MCMC_setsigma2nobs
!!!     call MCMC_setpar0(5, 0.0d0); call MCMC_setcmat0(0.1d0)      ! testcases/mcmcrun4.F90:11-12
!!!     call mcmc_main()                                            ! testcases/mcmcrun.F90:51
!!!
!!! and is linked with  mcmcx_mod.o  -L<repo>/mcmcf90_amd -lmcmcx  instead of  -lmcmcrun.
!!!
!!!   module mcmcmod      same name and generic setters as mcmc.F90:12,62-71 / MCMC_init.F90:168-347;
!!!                       public chain, sschain, s2chain, npar, nycol, sigma2, simuind, chainind
!!!                       (mcmc.F90:28-52); namelist /mcmc/ with the reference's variable list
!!!                       (mcmcinit.F90:74-82) and defaults (:184-230), read with real namelist I/O
!!!                       from mcmcinit.nml, else the file named in mcmcnml.txt (:101-117)
!!!   mcmc_main()         external subroutine, no arguments (mcmc_main.F90:12-44): init, run, write
!!!
!!! The sampling itself runs in libmcmcx.so (HIP) through the bind(C) interfaces below
!!! (include/mcmcx.h).  By default the user's link-time ssfunction / priorfun / checkbounds are called on
!!! the host exactly where the reference calls them (candidates make a D2H/H2D round trip: the plumbing
!!! path).  The fast path keeps the likelihood on the GPU: pick a device-resident target by
!!! MCMC_settarget_* calls before mcmc_main or by an extra namelist group in the same file (ignored by
!!! the reference, which reads only &mcmc):
!!!
!!!     &mcmcx  devtarget='gauss'  nchains=65536  mufile='mcmctest_mu.dat' lamfile='mcmctest_lam.dat' /
!!!
!!! Input files follow initialize.F90:41-119 (parfile, cov0file, sigma2file; whitespace/comma
!!! separated numbers, lines starting with # % ! are comments, matutils.F90:1019-1021); outputs follow
!!! MCMC_writechains (MCMC_aux.F90:25-69) in ASCII for chain 0.
module mcmcmod
  use iso_c_binding
  use mcmcprec
  use matutils
  implicit none
  public

  character(len=*), parameter :: Mcmc_Code_Version = 'mcmcx 0.1 (MI355X engine behind the mcmcf90 1.2.8 surface)'

  !! namelist /mcmc/ variables, mcmcinit.F90:20-62
  integer, save :: nsimu, doadapt, doburnin, adaptint, adapthist, initcmatn, burnintime, badaptint
  integer, save :: adaptend, greedy, svddim, sstype, filepars, printint, updatesigma, usrfunlen, dumpint
  integer, save :: verbosity
  real(kind=dbl), save :: scalelimit, scalefactor, drscale, condmax, condmaxini, N0, S02, sstrans
  real(kind=dbl), save :: alphatarget, nuparam
  character(len=256), save :: chainfile, s2file, ssfile, priorsfile, cov0file, covffile, covnfile
  character(len=256), save :: meanfile, nmlffile, parfile, parffile, sigma2file, sigma2ffile
  character(len=10), save :: method
  namelist /mcmc/ nsimu, doadapt, doburnin, adaptint, adapthist, &
       scalelimit, scalefactor, drscale, &
       badaptint, adaptend, initcmatn, N0, S02, &
       filepars, burnintime, greedy, printint, updatesigma, &
       usrfunlen, chainfile, s2file, ssfile, svddim, condmax, &
       cov0file, covffile, covnfile, meanfile, &
       nmlffile, parfile, parffile, sigma2file, sigma2ffile, &
       condmaxini, sstype, sstrans, dumpint, priorsfile, verbosity, &
       method, alphatarget, nuparam

  !! engine extension group &mcmcx (same file)
  character(len=16), save :: devtarget = 'host'       ! 'host' = the user's link-time ssfunction/priorfun/checkbounds
  integer, save :: nchains = 1, seed = 1835232611     ! 0x6D636D63
  integer, save :: pooled = 0                          ! 1: one shared proposal factor from the pooled covariance
  real(kind=dbl), save :: banana_b = 0.1_dbl
  character(len=256), save :: mufile = 'mcmctest_mu.dat', lamfile = 'mcmctest_lam.dat'
  character(len=256), save :: datafile = 'data.dat', lowerfile = '', upperfile = ''
  integer, save :: ngpus = 1                           ! GPUs of this node: nchains/ngpus chains each, one RCCL communicator
  integer, save :: usecomm = 0                         ! 1: form the RCCL communicator for ngpus = 1 too (the N-GPU code path on one GPU)
  integer, save :: scamfast = 0                        ! 1 (method = 'scam'): componentwise proposals as oldpar + delta U(:,j) (include/mcmcx.h: scam_fast)
  !! the user's own functions faster than one host call per chain:
  !!   hostbatch = 1: ssfunction_batch(theta(npar,n), npar, n, ny, ss(ny,n)) once per stage (default member: a loop over
  !!                  ssfunction), from hostthreads threads on disjoint column blocks when hostthreads > 1
  !!   devtarget = 'module': device code built with include/mcmcx_target.h -- modulefile (hipcc --genco output),
  !!                  modulekernel (the MCMCX_DEFINE_TARGET name), moduledatafile (numbers handed to the functions)
  integer, save :: hostbatch = 0, hostthreads = 1
  character(len=256), save :: modulefile = '', modulekernel = 'user_target', moduledatafile = ''
  namelist /mcmcx/ devtarget, nchains, seed, pooled, banana_b, mufile, lamfile, datafile, lowerfile, upperfile, ngpus, &
       hostbatch, hostthreads, modulefile, modulekernel, moduledatafile, usecomm, scamfast

  !! public state, mcmc.F90:28-52
  integer, save :: npar = 0, nycol = 1, simuind = 0, chainind = 0, MCMC_running = 0
  real(kind=dbl), allocatable, save :: chain(:,:), sschain(:,:), s2chain(:,:), sigma2(:)
  real(kind=dbl), allocatable, save :: chaincmat(:,:), chainmean(:)
  real(kind=dbl), allocatable, save :: laststates(:,:), pooledmean(:), pooledcov(:,:)   ! nchains > 1 (engine extension)
  real(kind=dbl), save :: chainwsum = 0.0_dbl
  integer, allocatable, save :: nobs(:)
  integer, save :: stayed = 0, bndstayed = 0, draccepted = 0, drtries = 0

  !! set by the MCMC_set* calls
  real(kind=dbl), allocatable, save, private :: par0(:), cmat0(:,:)
  real(kind=dbl), allocatable, save, private :: tmu(:), tlam(:,:), tx(:), ty(:), tlo(:), thi(:), pmu(:), psig(:), moddata(:)
  logical, save, private :: par0ok = .false., cmat0ok = .false., sigma2ok = .false., nparok = .false.
  logical, save, private :: has_lo = .false., has_hi = .false., interrupted = .false.
  character(len=32), save, private :: seedfile_used = ''
  type(c_ptr), save, private :: handle = c_null_ptr                  ! the engine of GPU 0 (= handles(1)): chain 1 lives there
  type(c_ptr), allocatable, save, private :: handles(:), comms(:)    ! one engine + one communicator rank per GPU

  interface MCMC_setpar0
     module procedure MCMC_setpar0_vec, MCMC_setpar0_n, MCMC_setpar0_file
  end interface
  interface MCMC_setcmat0
     module procedure MCMC_setcmat0_mat, MCMC_setcmat0_pct, MCMC_setcmat0_std, MCMC_setcmat0_file
  end interface
  interface MCMC_setsigma2nobs
     module procedure MCMC_setsigma2nobs_sca, MCMC_setsigma2nobs_vec
  end interface

  !! plain-C mirror of mcmcx_config (include/mcmcx.h)
  type, bind(C) :: mcmcx_config
     integer(c_int32_t) :: npar, nchains, method, nsimu
     integer(c_int32_t) :: doadapt, doburnin, adaptint, adapthist, badaptint, adaptend, initcmatn
     integer(c_int32_t) :: burnintime, greedy, updatesigma
     real(c_double) :: scalelimit, scalefactor, drscale, N0, S02, condmax, alphatarget, nuparam
     integer(c_int32_t) :: seed, chain_id0, record_accept, record_chain, device, pooled, scam_fast
  end type mcmcx_config

  interface
     subroutine mcmcx_config_defaults(cfg) bind(C, name='mcmcx_config_defaults')
       import :: mcmcx_config
       type(mcmcx_config), intent(inout) :: cfg
     end subroutine
     function mcmcx_create(cfg, h) bind(C, name='mcmcx_create') result(rc)
       import :: mcmcx_config, c_ptr, c_int
       type(mcmcx_config), intent(in) :: cfg
       type(c_ptr), intent(out) :: h
       integer(c_int) :: rc
     end function
     function mcmcx_destroy(h) bind(C, name='mcmcx_destroy') result(rc)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int) :: rc
     end function
     function mcmcx_last_error() bind(C, name='mcmcx_last_error') result(p)
       import :: c_ptr
       type(c_ptr) :: p
     end function
     function mcmcx_set_par0(h, p, n) bind(C, name='mcmcx_set_par0') result(rc)
       import :: c_ptr, c_int, c_double, c_int32_t
       type(c_ptr), value :: h
       real(c_double), intent(in) :: p(*)
       integer(c_int32_t), value :: n
       integer(c_int) :: rc
     end function
     function mcmcx_set_cmat0(h, c, n) bind(C, name='mcmcx_set_cmat0') result(rc)
       import :: c_ptr, c_int, c_double, c_int32_t
       type(c_ptr), value :: h
       real(c_double), intent(in) :: c(*)
       integer(c_int32_t), value :: n
       integer(c_int) :: rc
     end function
     function mcmcx_set_sigma2nobs(h, s2, nobs, ny) bind(C, name='mcmcx_set_sigma2nobs') result(rc)
       import :: c_ptr, c_int, c_double, c_int32_t
       type(c_ptr), value :: h
       real(c_double), intent(in) :: s2(*)
       integer(c_int32_t), intent(in) :: nobs(*)
       integer(c_int32_t), value :: ny
       integer(c_int) :: rc
     end function
     function mcmcx_set_target_gauss(h, mu, lam) bind(C, name='mcmcx_set_target_gauss') result(rc)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       real(c_double), intent(in) :: mu(*), lam(*)
       integer(c_int) :: rc
     end function
     function mcmcx_set_target_banana(h, b) bind(C, name='mcmcx_set_target_banana') result(rc)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       real(c_double), value :: b
       integer(c_int) :: rc
     end function
     function mcmcx_set_target_expdata(h, n, x, y) bind(C, name='mcmcx_set_target_expdata') result(rc)
       import :: c_ptr, c_int, c_double, c_int32_t
       type(c_ptr), value :: h
       integer(c_int32_t), value :: n
       real(c_double), intent(in) :: x(*), y(*)
       integer(c_int) :: rc
     end function
     function mcmcx_set_target_expdata_cols(h, n, ny, x, y) bind(C, name='mcmcx_set_target_expdata_cols') result(rc)
       import :: c_ptr, c_int, c_double, c_int32_t
       type(c_ptr), value :: h
       integer(c_int32_t), value :: n, ny
       real(c_double), intent(in) :: x(*), y(*)
       integer(c_int) :: rc
     end function
     function mcmcx_set_target_host(h, ss, pri, cb, user) bind(C, name='mcmcx_set_target_host') result(rc)
       import :: c_ptr, c_funptr, c_int
       type(c_ptr), value :: h, user
       type(c_funptr), value :: ss, pri, cb
       integer(c_int) :: rc
     end function
     function mcmcx_set_target_host_er(h, ss_er) bind(C, name='mcmcx_set_target_host_er') result(rc)
       import :: c_ptr, c_funptr, c_int
       type(c_ptr), value :: h
       type(c_funptr), value :: ss_er
       integer(c_int) :: rc
     end function
     function mcmcx_set_bounds(h, lo, hi) bind(C, name='mcmcx_set_bounds') result(rc)
       import :: c_ptr, c_int
       type(c_ptr), value :: h, lo, hi
       integer(c_int) :: rc
     end function
     function mcmcx_set_priors(h, mu, sig) bind(C, name='mcmcx_set_priors') result(rc)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       real(c_double), intent(in) :: mu(*), sig(*)
       integer(c_int) :: rc
     end function
     function mcmcx_init(h) bind(C, name='mcmcx_init') result(rc)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int) :: rc
     end function
     function mcmcx_run(h, upto) bind(C, name='mcmcx_run') result(rc)
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: h
       integer(c_int32_t), value :: upto
       integer(c_int) :: rc
     end function
     function mcmcx_sync(h) bind(C, name='mcmcx_sync') result(rc)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int) :: rc
     end function
     function mcmcx_get_counters(h, chain, c8) bind(C, name='mcmcx_get_counters') result(rc)
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), value :: h
       integer(c_int32_t), value :: chain
       integer(c_int32_t), intent(out) :: c8(8)
       integer(c_int) :: rc
     end function
     function mcmcx_get_chain(h, chain, ch, ss, s2, nrows) bind(C, name='mcmcx_get_chain') result(rc)
       import :: c_ptr, c_int, c_int32_t, c_double
       type(c_ptr), value :: h
       integer(c_int32_t), value :: chain
       real(c_double), intent(out) :: ch(*), ss(*), s2(*)
       integer(c_int32_t), intent(out) :: nrows
       integer(c_int) :: rc
     end function
     function mcmcx_get_chaincov(h, chain, cm, mean, wsum) bind(C, name='mcmcx_get_chaincov') result(rc)
       import :: c_ptr, c_int, c_int32_t, c_double
       type(c_ptr), value :: h
       integer(c_int32_t), value :: chain
       real(c_double), intent(out) :: cm(*), mean(*), wsum
       integer(c_int) :: rc
     end function
     function mcmcx_get_theta(h, out) bind(C, name='mcmcx_get_theta') result(rc)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       real(c_double), intent(out) :: out(*)
       integer(c_int) :: rc
     end function
     function mcmcx_pooled_moments(h, out) bind(C, name='mcmcx_pooled_moments') result(rc)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       real(c_double), intent(out) :: out(*)
       integer(c_int) :: rc
     end function
     function mcmcx_simuind(h) bind(C, name='mcmcx_simuind') result(n)
       import :: c_ptr, c_int32_t
       type(c_ptr), value :: h
       integer(c_int32_t) :: n
     end function
     function mcmcx_install_signal_handlers() bind(C, name='mcmcx_install_signal_handlers') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function
     function mcmcx_get_scalars(h, out) bind(C, name='mcmcx_get_scalars') result(rc)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       real(c_double), intent(out) :: out(*)
       integer(c_int) :: rc
     end function
     function mcmcx_set_target_host_batch(h, ssb, pri, cb, user, nthreads) bind(C, name='mcmcx_set_target_host_batch') result(rc)
       import :: c_ptr, c_funptr, c_int, c_int32_t
       type(c_ptr), value :: h, user
       type(c_funptr), value :: ssb, pri, cb
       integer(c_int32_t), value :: nthreads
       integer(c_int) :: rc
     end function
     function mcmcx_set_target_module(h, path, kname, udata, nbytes) bind(C, name='mcmcx_set_target_module') result(rc)
       import :: c_ptr, c_char, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       character(kind=c_char), intent(in) :: path(*), kname(*)
       real(c_double), intent(in) :: udata(*)
       integer(c_int64_t), value :: nbytes
       integer(c_int) :: rc
     end function
     function mcmcx_get_totals_n(h, t, n) bind(C, name='mcmcx_get_totals_n') result(rc)
       import :: c_ptr, c_int, c_int64_t, c_int32_t
       type(c_ptr), value :: h
       integer(c_int64_t), intent(out) :: t(*)
       integer(c_int32_t), value :: n
       integer(c_int) :: rc
     end function
     function mcmcx_comm_create_all(ndev, devices, out) bind(C, name='mcmcx_comm_create_all') result(rc)
       import :: c_ptr, c_int, c_int32_t
       integer(c_int32_t), value :: ndev
       type(c_ptr), value :: devices
       type(c_ptr), intent(out) :: out(*)
       integer(c_int) :: rc
     end function
     function mcmcx_comm_destroy(c) bind(C, name='mcmcx_comm_destroy') result(rc)
       import :: c_ptr, c_int
       type(c_ptr), value :: c
       integer(c_int) :: rc
     end function
     function mcmcx_set_comm(h, c) bind(C, name='mcmcx_set_comm') result(rc)
       import :: c_ptr, c_int
       type(c_ptr), value :: h, c
       integer(c_int) :: rc
     end function
     function mcmcx_run_all(hs, n, upto) bind(C, name='mcmcx_run_all') result(rc)
       import :: c_ptr, c_int, c_int32_t
       type(c_ptr), intent(in) :: hs(*)
       integer(c_int32_t), value :: n, upto
       integer(c_int) :: rc
     end function
     function mcmcx_allreduce_moments_all(hs, n, out) bind(C, name='mcmcx_allreduce_moments_all') result(rc)
       import :: c_ptr, c_int, c_int32_t, c_double
       type(c_ptr), intent(in) :: hs(*)
       integer(c_int32_t), value :: n
       real(c_double), intent(out) :: out(*)
       integer(c_int) :: rc
     end function
     function mcmcx_set_target_external(h) bind(C, name='mcmcx_set_target_external') result(rc)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int) :: rc
     end function
     function mcmcx_run1_decide(h, drstage, oldpar2, ssprev2, sspri2, oldpar1, ssprev1, sspri1, alpha12, newpar, ss, sspri, &
          alpha, reject) bind(C, name='mcmcx_run1_decide') result(rc)
       import :: c_ptr, c_int, c_int32_t, c_double
       type(c_ptr), value :: h
       integer(c_int32_t), value :: drstage
       real(c_double), intent(in) :: oldpar2(*), ssprev2(*), sspri2(*), oldpar1(*), ssprev1(*), sspri1(*), alpha12(*)
       real(c_double), intent(in) :: newpar(*), ss(*), sspri(*)
       real(c_double), intent(out) :: alpha(*)
       integer(c_int32_t), intent(out) :: reject(*)
       integer(c_int) :: rc
     end function
     function mcmcx_run1_propose(h, stage, from, newpar) bind(C, name='mcmcx_run1_propose') result(rc)
       import :: c_ptr, c_int, c_int32_t, c_double
       type(c_ptr), value :: h
       integer(c_int32_t), value :: stage
       real(c_double), intent(in) :: from(*)
       real(c_double), intent(out) :: newpar(*)
       integer(c_int) :: rc
     end function
     function mcmcx_run1_sscrit(h, ssprev1, sspri1, crit) bind(C, name='mcmcx_run1_sscrit') result(rc)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       real(c_double), intent(in) :: ssprev1(*), sspri1(*)
       real(c_double), intent(out) :: crit(*)
       integer(c_int) :: rc
     end function
  end interface

contains

  subroutine chk(rc)
    integer(c_int), intent(in) :: rc
    character(kind=c_char), pointer :: s(:)
    character(len=512) :: msg
    integer :: i
    if (rc >= 0) return
    call c_f_pointer(mcmcx_last_error(), s, [512])
    msg = ''
    do i = 1, 512
       if (s(i) == c_null_char) exit
       msg(i:i) = s(i)
    end do
    call doerror(msg)
  end subroutine chk

  !! MCMC_init_namelist, mcmcinit.F90:184-230
  subroutine MCMC_init_namelist()
    nsimu = 0; doadapt = 1; doburnin = 0; burnintime = 0; badaptint = -1; greedy = 0
    scalelimit = 0.05_dbl; scalefactor = 2.5_dbl; drscale = 0.0_dbl; adaptint = 100; adapthist = 0
    adaptend = 0; initcmatn = 0; N0 = 1.0_dbl; S02 = 0.0_dbl; filepars = 1; printint = 500
    updatesigma = 1; usrfunlen = 0; dumpint = 0
    chainfile = 'chain.dat'; s2file = 's2chain.dat'; ssfile = 'sschain.dat'; priorsfile = ''
    cov0file = 'mcmccov.dat'; covffile = 'mcmccovf.dat'; covnfile = ''; meanfile = 'mcmcmean.dat'
    nmlffile = ''; parfile = 'mcmcpar.dat'; parffile = 'mcmcparf.dat'
    sigma2file = 'mcmcsigma2.dat'; sigma2ffile = 'mcmcsigma2f.dat'
    svddim = 0; condmax = 0.0_dbl; condmaxini = 1.0e15_dbl; sstype = 0; sstrans = -1.0_dbl
    verbosity = 1; method = 'dram'; alphatarget = 0.234_dbl; nuparam = 0.7_dbl
  end subroutine MCMC_init_namelist

  !! read_mcmcinit_namelist, mcmcinit.F90:86-142
  subroutine read_mcmcinit_namelist(status)
    integer, intent(out) :: status
    integer :: fstat
    character(len=256) :: nmlfile
    logical :: fexist
    status = 0
    call MCMC_init_namelist()
    inquire(file='mcmcinit.nml', exist=fexist)
    if (fexist) then
       nmlfile = 'mcmcinit.nml'
    else
       inquire(file='mcmcnml.txt', exist=fexist)
       nmlfile = 'mcmcinit.nml'
       if (fexist) then
          open(unit=10, file='mcmcnml.txt', status='old', iostat=fstat)
          read(10,*) nmlfile
          close(10)
          if (len_trim(nmlfile) == 0) nmlfile = 'mcmcinit.nml'
       end if
    end if
    open(unit=10, file=nmlfile, status='old', iostat=fstat)
    if (fstat /= 0) then
       write(*,*) 'File ', trim(nmlfile), ' not found, no MCMC run'
       status = -1
       return
    end if
    read(10, nml=mcmc, iostat=fstat)
    if (fstat /= 0) then
       write(*,*) 'Error reading mcmc namelist from file ', trim(nmlfile)
       write(*,*) ' status:', fstat
       write(*,*) 'No mcmc run'
       close(10)
       status = -2
       return
    end if
    rewind(10)
    read(10, nml=mcmcx, iostat=fstat)      ! optional engine group; absent -> defaults / MCMC_settarget_*
    close(10)
    if (verbosity > 0) write(*,*) 'note: using nmlfile ', trim(nmlfile)
  end subroutine read_mcmcinit_namelist

  !! ---------------------------------------------------------------- setters, MCMC_init.F90:168-347
  subroutine MCMC_setpar0_vec(par)
    real(kind=dbl), intent(in) :: par(:)
    if (nparok .and. npar /= size(par)) call doerror('par0 and npar dont match')
    npar = size(par)
    if (allocated(par0)) deallocate(par0)
    allocate(par0(npar)); par0 = par
    nparok = .true.; par0ok = .true.
  end subroutine MCMC_setpar0_vec
  subroutine MCMC_setpar0_n(n, par)
    integer, intent(in) :: n
    real(kind=dbl), intent(in) :: par
    real(kind=dbl) :: v(n)
    v = par
    call MCMC_setpar0_vec(v)
  end subroutine MCMC_setpar0_n
  subroutine MCMC_setpar0_file(file)
    character(len=*), intent(in) :: file
    real(kind=dbl), allocatable :: v(:)
    integer :: nr, nc, stat
    call loadnumbers(file, v, nr, nc, stat)
    if (stat /= 0) call doerror('Error reading file '//trim(file))
    call MCMC_setpar0_vec(v)
  end subroutine MCMC_setpar0_file
  subroutine MCMC_setcmat0_mat(cmat)
    real(kind=dbl), intent(in) :: cmat(:,:)
    if (nparok .and. npar /= size(cmat,1)) call doerror('cmat0 and npar dont match')
    if (size(cmat,2) /= size(cmat,1)) call doerror('cmat0 must be square')
    npar = size(cmat,1)
    if (allocated(cmat0)) deallocate(cmat0)
    allocate(cmat0(npar,npar)); cmat0 = cmat
    nparok = .true.; cmat0ok = .true.
  end subroutine MCMC_setcmat0_mat
  subroutine MCMC_setcmat0_pct(pct)                       ! MCMC_init.F90:240-262
    real(kind=dbl), intent(in) :: pct
    integer :: i
    if (.not.nparok .or. .not.par0ok) call doerror('par0 not defined when calling setcmat0')
    if (allocated(cmat0)) deallocate(cmat0)
    allocate(cmat0(npar,npar)); cmat0 = 0.0_dbl
    do i = 1, npar
       if (par0(i) /= 0.0_dbl) then
          cmat0(i,i) = abs(pct*par0(i))**2
       else
          cmat0(i,i) = abs(pct)**2
       end if
    end do
    cmat0ok = .true.
  end subroutine MCMC_setcmat0_pct
  subroutine MCMC_setcmat0_std(std)
    real(kind=dbl), intent(in) :: std(:)
    integer :: i
    if (.not.nparok) call doerror('npar not defined when calling setcmat0')
    if (size(std) /= npar) call doerror('std size does not macth when calling setcmat0')
    if (allocated(cmat0)) deallocate(cmat0)
    allocate(cmat0(npar,npar)); cmat0 = 0.0_dbl
    do i = 1, npar
       cmat0(i,i) = std(i)**2
    end do
    cmat0ok = .true.
  end subroutine MCMC_setcmat0_std
  subroutine MCMC_setcmat0_file(file)
    character(len=*), intent(in) :: file
    real(kind=dbl), allocatable :: v(:)
    integer :: nr, nc, stat
    call loadnumbers(file, v, nr, nc, stat)
    if (stat /= 0 .or. nr /= nc) call doerror('Error reading file '//trim(file))
    call MCMC_setcmat0_mat(transpose(reshape(v, (/nc, nr/))))
  end subroutine MCMC_setcmat0_file
  subroutine MCMC_setsigma2nobs_vec(sig2, n)
    real(kind=dbl), intent(in) :: sig2(:)
    integer, intent(in) :: n(:)
    if (size(sig2) /= size(n)) call doerror('sig2 and nobs sizes')
    nycol = size(sig2)
    if (allocated(sigma2)) deallocate(sigma2)
    if (allocated(nobs)) deallocate(nobs)
    allocate(sigma2(nycol), nobs(nycol))
    sigma2 = sig2; nobs = n
    sigma2ok = .true.
  end subroutine MCMC_setsigma2nobs_vec
  subroutine MCMC_setsigma2nobs_sca(sig2, n)
    real(kind=dbl), intent(in) :: sig2
    integer, intent(in) :: n
    call MCMC_setsigma2nobs_vec((/sig2/), (/n/))
  end subroutine MCMC_setsigma2nobs_sca

  !! ---------------------------------------------------------------- device-resident user callbacks
  subroutine MCMC_settarget_gauss(mu, lam)              ! ss = (theta-mu)' lam (theta-mu), testcases/mcmcrun4.F90:47
    real(kind=dbl), intent(in) :: mu(:), lam(:,:)
    if (allocated(tmu)) deallocate(tmu, tlam)
    allocate(tmu(size(mu)), tlam(size(lam,1), size(lam,2)))
    tmu = mu; tlam = lam; devtarget = 'gauss'
  end subroutine MCMC_settarget_gauss
  subroutine MCMC_settarget_banana(b)
    real(kind=dbl), intent(in) :: b
    banana_b = b; devtarget = 'banana'
  end subroutine MCMC_settarget_banana
  subroutine MCMC_settarget_expdata(x, y)               ! ss = sum((y - th1*exp(-th2*x))**2), testcases/mcmcrun.F90:89,104
    real(kind=dbl), intent(in) :: x(:), y(:)
    if (allocated(tx)) deallocate(tx, ty)
    allocate(tx(size(x)), ty(size(y)))
    tx = x; ty = y; devtarget = 'expdata'
  end subroutine MCMC_settarget_expdata
  subroutine MCMC_settarget_expdata_cols(x, y)          ! nycol columns: ss(j) = sum((y(:,j) - th1*exp(-th(1+j)*x))**2)
    real(kind=dbl), intent(in) :: x(:), y(:,:)
    if (allocated(tx)) deallocate(tx, ty)
    allocate(tx(size(x)), ty(size(y)))
    tx = x; ty = reshape(y, (/size(y)/)); devtarget = 'expcols'
  end subroutine MCMC_settarget_expdata_cols
  subroutine MCMC_setbounds(lo, hi)                     ! box form of checkbounds, external_inc.h:29-32
    real(kind=dbl), intent(in), optional :: lo(:), hi(:)
    if (present(lo)) then
       if (allocated(tlo)) deallocate(tlo)
       allocate(tlo(size(lo))); tlo = lo; has_lo = .true.
    end if
    if (present(hi)) then
       if (allocated(thi)) deallocate(thi)
       allocate(thi(size(hi))); thi = hi; has_hi = .true.
    end if
  end subroutine MCMC_setbounds
  subroutine MCMC_setnchains(n)
    integer, intent(in) :: n
    nchains = n
  end subroutine MCMC_setnchains

  !! ---------------------------------------------------------------- adapters for the user's link-time callbacks
  !! (external_inc.h:4-33: array-valued result and assumed-shape dummies need the Fortran compiler's own ABI,
  !! so the C engine calls these bind(C) wrappers, which call the user's functions)
  subroutine mcx_ss_adapter(theta, n, ny, ss_out, user) bind(C)
    real(c_double), intent(in) :: theta(*)
    integer(c_int32_t), value :: n, ny
    real(c_double), intent(out) :: ss_out(*)
    type(c_ptr), value :: user
    integer(kind=4) :: n4, ny4
    interface
       function ssfunction(theta,npar,ny)
         integer*4 npar, ny
         real*8 theta(npar)
         real*8 ssfunction(ny)
       end function ssfunction
    end interface
    n4 = n; ny4 = ny
    ss_out(1:ny) = ssfunction(theta(1:n), n4, ny4)
  end subroutine mcx_ss_adapter
  subroutine mcx_ss_batch_adapter(theta, n, nb, ny, ss_out, user) bind(C)
    real(c_double), intent(in) :: theta(*)
    integer(c_int32_t), value :: n, nb, ny
    real(c_double), intent(out) :: ss_out(*)
    type(c_ptr), value :: user
    integer(kind=4) :: n4, nb4, ny4
    interface
       subroutine ssfunction_batch(theta,npar,n,ny,ss)
         integer*4 npar, n, ny
         real*8 theta(npar,n)
         real*8 ss(ny,n)
       end subroutine ssfunction_batch
    end interface
    n4 = n; nb4 = nb; ny4 = ny
    call ssfunction_batch(theta, n4, nb4, ny4, ss_out)
  end subroutine mcx_ss_batch_adapter
  subroutine mcx_ss_er_adapter(theta, n, ny, sscrit, ss_out, user) bind(C)
    real(c_double), intent(in) :: theta(*)
    integer(c_int32_t), value :: n, ny
    real(c_double), value :: sscrit
    real(c_double), intent(out) :: ss_out(*)
    type(c_ptr), value :: user
    integer(kind=4) :: n4, ny4
    real(kind=8) :: crit
    interface
       function ssfunction_er(theta,npar,ny,sscrit)
         integer*4 npar, ny
         real*8 theta(npar), sscrit
         real*8 ssfunction_er(ny)
       end function ssfunction_er
    end interface
    n4 = n; ny4 = ny; crit = sscrit
    ss_out(1:ny) = ssfunction_er(theta(1:n), n4, ny4, crit)
  end subroutine mcx_ss_er_adapter
  function mcx_prior_adapter(theta, n, user) bind(C) result(p)
    real(c_double), intent(in) :: theta(*)
    integer(c_int32_t), value :: n
    type(c_ptr), value :: user
    real(c_double) :: p
    integer(kind=4) :: n4
    interface
       function priorfun(theta,len)
         real*8 priorfun
         integer*4 len
         real*8 theta(len)
       end function priorfun
    end interface
    n4 = n
    p = priorfun(theta(1:n), n4)
  end function mcx_prior_adapter
  function mcx_bounds_adapter(theta, n, user) bind(C) result(ok)
    real(c_double), intent(in) :: theta(*)
    integer(c_int32_t), value :: n
    type(c_ptr), value :: user
    integer(c_int32_t) :: ok
    real(kind=dbl) :: t(n)
    interface
       function checkbounds(theta)
         real*8 theta(:)
         logical checkbounds
       end function checkbounds
    end interface
    t = theta(1:n)
    ok = 0
    if (checkbounds(t)) ok = 1
  end function mcx_bounds_adapter

  !! ---------------------------------------------------------------- initial values, MCMC_init.F90:44-70
  !! After MCMC_setpar0 the missing pieces get the reference's defaults (cmat0 = MCMC_initcmat0: unit matrix;
  !! sigma2 = 1, nobs = 1); otherwise the link-time `initialize` (external_inc.h:35-43; default = files,
  !! defaults/initialize0.F90, overridable like in libmcmcrun.a) allocates and fills everything.
  subroutine MCMC_initial_values()
    real(kind=dbl), allocatable :: p0(:), c0(:,:), s2(:)
    integer, allocatable :: nob(:)
    integer :: np, i
    interface
       subroutine initialize(par0,npar,cmat0,initcmatn,sigma2,nobs,nycol)
         use mcmcprec
         implicit none
         integer, intent(inout) :: npar, initcmatn, nycol
         real(kind=dbl), intent(inout), allocatable :: par0(:), cmat0(:,:)
         real(kind=dbl), intent(inout), allocatable :: sigma2(:)
         integer, intent(inout), allocatable :: nobs(:)
       end subroutine initialize
    end interface
    if (nparok) then
       if (verbosity > 0) write(*,*) 'note: user init for par0'
       if (.not.par0ok) call doerror('user initialization error')
       if (.not.cmat0ok) then
          allocate(c0(npar,npar)); c0 = 0.0_dbl
          do i = 1, npar
             c0(i,i) = 1.0_dbl
          end do
          call MCMC_setcmat0_mat(c0)
       end if
       if (.not.sigma2ok) call MCMC_setsigma2nobs_sca(1.0_dbl, 1)
    else
       np = npar
       call initialize(p0, np, c0, initcmatn, s2, nob, nycol)
       if (.not.allocated(p0) .or. .not.allocated(c0) .or. .not.allocated(s2) .or. .not.allocated(nob)) &
            call doerror('initialize did not allocate par0, cmat0, sigma2, nobs')
       if (size(s2) /= nycol .or. size(nob) /= nycol) call doerror('initialize: sigma2 / nobs are not of length nycol')
       call MCMC_setpar0_vec(p0)
       if (size(c0,1) /= npar .or. size(c0,2) /= npar) call doerror('initialize: cmat0 is not npar x npar')
       call MCMC_setcmat0_mat(c0)
       call MCMC_setsigma2nobs_vec(s2, nob)
    end if
  end subroutine MCMC_initial_values

  subroutine load_target_from_files()
    real(kind=dbl), allocatable :: v(:), w(:), a2(:,:)
    integer :: nr, nc, stat
    select case (trim(devtarget))
    case ('gauss')
       if (.not.allocated(tmu)) then
          call loadnumbers(mufile, v, nr, nc, stat)
          if (stat /= 0 .or. size(v) /= npar) call doerror('error in sizes: '//trim(mufile))
          call loadnumbers(lamfile, w, nr, nc, stat)
          if (stat /= 0 .or. size(w) /= npar*npar) call doerror('error in sizes: '//trim(lamfile))
          call MCMC_settarget_gauss(v, transpose(reshape(w, (/npar, npar/))))
       end if
    case ('expdata')
       if (.not.allocated(tx)) then
          call loadnumbers(datafile, v, nr, nc, stat)
          if (stat /= 0 .or. nc /= 2) call doerror('error reading '//trim(datafile))
          call MCMC_settarget_expdata(v(1::2), v(2::2))
       end if
    case ('expcols')                                   ! datafile rows: x, y_1 .. y_nycol
       if (.not.allocated(tx)) then
          call loadnumbers(datafile, v, nr, nc, stat)
          if (stat /= 0 .or. nc < 2) call doerror('error reading '//trim(datafile))
          allocate(a2(nc, nr)); a2 = reshape(v, (/nc, nr/))          ! a2(column, row)
          call MCMC_settarget_expdata_cols(a2(1,:), transpose(a2(2:nc,:)))
          deallocate(a2)
       end if
    case ('banana')
    case ('host')
    case ('module')                                    ! the numbers the user's device functions get as `data`
       if (len_trim(moduledatafile) > 0 .and. .not.allocated(moddata)) then
          call loadnumbers(moduledatafile, v, nr, nc, stat)
          if (stat /= 0) call doerror('error reading '//trim(moduledatafile))
          allocate(moddata(size(v))); moddata = v
       end if
    case default
       call doerror('unknown devtarget in &mcmcx: '//trim(devtarget))
    end select
    if (len_trim(lowerfile) > 0 .and. .not.has_lo) then
       call loadnumbers(lowerfile, v, nr, nc, stat)
       if (stat /= 0 .or. size(v) /= npar) call doerror('error reading '//trim(lowerfile))
       call MCMC_setbounds(lo=v)
    end if
    if (len_trim(upperfile) > 0 .and. .not.has_hi) then
       call loadnumbers(upperfile, v, nr, nc, stat)
       if (stat /= 0 .or. size(v) /= npar) call doerror('error reading '//trim(upperfile))
       call MCMC_setbounds(hi=v)
    end if
    if (len_trim(priorsfile) > 0 .and. trim(devtarget) /= 'host' .and. trim(devtarget) /= 'module') then    ! priorfun.f90:52-80: two rows, mu and sigma
       call loadnumbers(priorsfile, v, nr, nc, stat)
       if (stat /= 0 .or. size(v) /= 2*npar) call doerror('priors.dat should have  2*npar elements')
       allocate(pmu(npar), psig(npar)); pmu = v(1:npar); psig = v(npar+1:2*npar)
    end if
  end subroutine load_target_from_files

  !! ---------------------------------------------------------------- MCMC_init + MCMC_run* + results
  subroutine MCMC_engine_run()
    type(mcmcx_config) :: cfg
    integer(c_int32_t) :: c8(8), nrows
    integer(c_int64_t) :: t7(7)
    integer(c_int32_t), allocatable :: nob(:)
    real(kind=dbl), allocatable :: ch(:), ss(:), s2(:), cm(:), sc(:)
    real(kind=dbl), target, allocatable :: lo(:), hi(:)
    real(kind=dbl) :: lamrow(npar*npar)
    type(c_ptr) :: plo, phi
    real(kind=dbl), allocatable :: th(:), pm(:)
    integer :: i, j, g, upto, nxt, nloc, stbits
    integer(c_int) :: rc
    interface
       subroutine dump_init()
       end subroutine dump_init
       subroutine dump(oldpar)
         use mcmcprec
         real(kind=dbl), intent(in) :: oldpar(:)
       end subroutine dump
    end interface
    call chk(mcmcx_install_signal_handlers())           ! signal_handler_init, MCMC_signal_handler.F90:21-60
    call read_seed_file()
    !! &mcmcx ngpus = G: the nchains chains are split into G contiguous blocks, one engine per GPU, chain c keyed
    !! (seed, c) whatever G is; the GPUs meet in one RCCL communicator for the pooled moments (SURVEY 8e)
    if (ngpus < 1) ngpus = 1
    if (mod(nchains, ngpus) /= 0) call doerror('&mcmcx: nchains must be a multiple of ngpus')
    nloc = nchains / ngpus
    allocate(handles(ngpus), comms(ngpus))
    handles = c_null_ptr; comms = c_null_ptr
    if (ngpus > 1 .or. usecomm /= 0) call chk(mcmcx_comm_create_all(int(ngpus, c_int32_t), c_null_ptr, comms))
    call mcmcx_config_defaults(cfg)
    cfg%npar = npar; cfg%nchains = nloc; cfg%nsimu = nsimu
    select case (trim(method))
    case ('ram');  cfg%method = 1
    case ('scam'); cfg%method = 2
    case ('er');   cfg%method = 3
    case default;  cfg%method = 0
    end select
    cfg%doadapt = doadapt; cfg%doburnin = doburnin; cfg%adaptint = adaptint; cfg%adapthist = adapthist
    cfg%badaptint = badaptint; cfg%adaptend = adaptend; cfg%initcmatn = initcmatn
    cfg%burnintime = burnintime; cfg%greedy = greedy; cfg%updatesigma = updatesigma
    cfg%scalelimit = scalelimit; cfg%scalefactor = scalefactor; cfg%drscale = drscale
    cfg%N0 = N0; cfg%S02 = S02; cfg%condmax = condmax; cfg%alphatarget = alphatarget; cfg%nuparam = nuparam
    cfg%seed = seed; cfg%record_accept = 0; cfg%pooled = pooled; cfg%scam_fast = scamfast
    allocate(nob(nycol)); nob = nobs                    ! nycol response columns: one sigma2 / nobs each (host callbacks when > 1)
    if (trim(devtarget) == 'gauss') then
       do i = 1, npar                                  ! row-major lam(i,j) for the C side
          do j = 1, npar
             lamrow((i-1)*npar + j) = tlam(i,j)
          end do
       end do
    end if
    plo = c_null_ptr; phi = c_null_ptr
    if (has_lo) then
       allocate(lo(npar)); lo = tlo; plo = c_loc(lo)
    end if
    if (has_hi) then
       allocate(hi(npar)); hi = thi; phi = c_loc(hi)
    end if
    do g = 1, ngpus
       cfg%device = g - 1; cfg%chain_id0 = (g - 1) * nloc
       cfg%record_chain = merge(1, 0, g == 1)           ! the reference's chain arrays hold chain 1, which lives on GPU 0
       call chk(mcmcx_create(cfg, handles(g)))
       handle = handles(g)
       if (ngpus > 1 .or. usecomm /= 0) call chk(mcmcx_set_comm(handle, comms(g)))
       call chk(mcmcx_set_par0(handle, par0, int(npar, c_int32_t)))
       call chk(mcmcx_set_cmat0(handle, cmat0, int(npar, c_int32_t)))
       call chk(mcmcx_set_sigma2nobs(handle, sigma2, nob, int(nycol, c_int32_t)))
       select case (trim(devtarget))
       case ('gauss')
          call chk(mcmcx_set_target_gauss(handle, tmu, lamrow))
       case ('banana')
          call chk(mcmcx_set_target_banana(handle, banana_b))
       case ('expdata')
          call chk(mcmcx_set_target_expdata(handle, int(size(tx), c_int32_t), tx, ty))
       case ('expcols')
          if (size(ty) /= size(tx)*nycol) call doerror('devtarget expcols: the data file needs 1 + nycol columns')
          call chk(mcmcx_set_target_expdata_cols(handle, int(size(tx), c_int32_t), int(nycol, c_int32_t), tx, ty))
       case ('host')
          if (hostbatch /= 0) then
             call chk(mcmcx_set_target_host_batch(handle, c_funloc(mcx_ss_batch_adapter), c_funloc(mcx_prior_adapter), &
                  c_funloc(mcx_bounds_adapter), c_null_ptr, int(hostthreads, c_int32_t)))
          else
             call chk(mcmcx_set_target_host(handle, c_funloc(mcx_ss_adapter), c_funloc(mcx_prior_adapter), &
                  c_funloc(mcx_bounds_adapter), c_null_ptr))
             call chk(mcmcx_set_target_host_er(handle, c_funloc(mcx_ss_er_adapter)))     ! used by method = 'er' only
          end if
       case ('module')
          if (len_trim(modulefile) == 0) call doerror('devtarget module: &mcmcx modulefile is not set')
          if (.not. allocated(moddata)) then
             allocate(moddata(1)); moddata = 0.0_dbl
          end if
          call chk(mcmcx_set_target_module(handle, trim(modulefile)//c_null_char, trim(modulekernel)//c_null_char, &
               moddata, int(8*size(moddata), c_int64_t)))
       end select
       if (has_lo .or. has_hi) call chk(mcmcx_set_bounds(handle, plo, phi))
       if (allocated(pmu)) call chk(mcmcx_set_priors(handle, pmu, psig))
       call chk(mcmcx_init(handle))
    end do
    handle = handles(1)
    call dump_init()                                    ! MCMC_dump_init, MCMC_run.F90:38
    MCMC_running = 1
    interrupted = .false.
    upto = 1
    do while (upto < nsimu)
       !! The run is cut where the host has something to do: every printint iterations the progress line of
       !! MCMC_adapt.F90:22-37 (chain 1's counters), every dumpint iterations MCMC_dump(oldpar) (MCMC_run.F90:102: the
       !! user's dump sees the current point of chain 1; the reference calls it at every iteration)
       nxt = nsimu
       if (dumpint > 0) nxt = min(nxt, (upto / dumpint + 1) * dumpint)
       if (printint > 0) nxt = min(nxt, (upto / printint + 1) * printint)
       upto = nxt
       rc = mcmcx_run_all(handles, int(ngpus, c_int32_t), int(upto, c_int32_t))
       call chk(rc)
       if (rc == 2) then                                ! MCMCX_INTERRUPTED: cc_handler, MCMC_signal_handler.F90:95-107
          interrupted = .true.
          exit
       end if
       if (printint > 0) then
          if (mod(upto, printint) == 0) call progress_line(upto)
       end if
       if (dumpint > 0) then
          if (mod(upto, dumpint) == 0 .or. upto == nsimu) then
             allocate(th(npar*nloc))
             call chk(mcmcx_get_theta(handle, th))
             call dump(th(1:npar))
             deallocate(th)
          end if
       end if
    end do
    do g = 1, ngpus
       call chk(mcmcx_sync(handles(g)))
    end do
    MCMC_running = 0
    simuind = mcmcx_simuind(handle)
    if (interrupted) write(*,*) 'Saving chain upto ', simuind
    !! chain 0 in the reference's arrays
    allocate(ch(nsimu*(npar+1)), ss(nsimu*(nycol+1)), s2(nsimu*nycol), cm(npar*npar), sc(4*nloc))
    call chk(mcmcx_get_chain(handle, 0_c_int32_t, ch, ss, s2, nrows))
    chainind = nrows
    if (allocated(chain)) deallocate(chain, sschain)
    allocate(chain(nsimu, npar+1), sschain(nsimu, nycol+1))             ! MCMC_init.F90:126-131
    chain = 0.0_dbl; sschain = 0.0_dbl
    do i = 1, chainind
       chain(i,:) = ch((i-1)*(npar+1)+1 : i*(npar+1))
       sschain(i,:) = ss((i-1)*(nycol+1)+1 : i*(nycol+1))
    end do
    if (updatesigma /= 0) then
       if (allocated(s2chain)) deallocate(s2chain)
       allocate(s2chain(nsimu,nycol)); s2chain = 0.0_dbl                ! MCMC_init.F90:119-122
       do i = 1, simuind
          s2chain(i,:) = s2((i-1)*nycol+1 : i*nycol)
       end do
    end if
    if (allocated(chaincmat)) deallocate(chaincmat, chainmean)
    allocate(chaincmat(npar,npar), chainmean(npar))
    call chk(mcmcx_get_chaincov(handle, 0_c_int32_t, cm, chainmean, chainwsum))
    chaincmat = reshape(cm, (/npar, npar/))
    call chk(mcmcx_get_counters(handle, 0_c_int32_t, c8))
    stayed = c8(1); bndstayed = c8(2); draccepted = c8(3); drtries = c8(4)
    call chk(mcmcx_get_scalars(handle, sc))
    sigma2(1) = sc(3)
    if (updatesigma /= 0 .and. simuind >= 1) sigma2 = s2chain(simuind,:)
    !! several chains: what the reference has no counterpart for -- the last state of every chain and the moments of
    !! those states over all chains (about par0: count, sum, upper second moments)
    !! where the reference would have stopped or warned (the engine records it per chain and carries on): a failed
    !! choldowndate stops (matutils.F90:719-722), so does a covariance that cannot be inverted for DR
    !! (MCMC_adapt.F90:221-224); a failed Cholesky / SVD keeps the old factor with a warning (MCMC_adapt.F90:168-171)
    stbits = 0
    do g = 1, ngpus
       call chk(mcmcx_get_totals_n(handles(g), t7, 7_c_int32_t))
       stbits = ior(stbits, int(t7(7)))
    end do
    if (iand(stbits, 2) /= 0) write(*,*) 'Warning: error in Chol/SVD, not adapting (some chain and tick)'
    if (iand(c8(6), 1) /= 0) call doerror('error in coldowndate, info: -1')
    if (iand(c8(6), 4) /= 0) call doerror('ERROR: cannot invert cmat')
    if (iand(stbits, 1) /= 0) write(*,*) 'Warning: choldowndate failed (info = -1) in some chain other than chain 1; its factor was left unchanged'
    if (iand(stbits, 4) /= 0) write(*,*) 'Warning: cannot invert cmat in some chain other than chain 1'
    if (nchains > 1) then
       allocate(th(npar*nloc), pm(1 + npar + npar*(npar+1)/2))
       if (allocated(laststates)) deallocate(laststates, pooledmean, pooledcov)
       allocate(laststates(nchains, npar), pooledmean(npar), pooledcov(npar, npar))
       do g = 1, ngpus
          call chk(mcmcx_get_theta(handles(g), th))
          laststates((g-1)*nloc+1 : g*nloc, :) = transpose(reshape(th, (/npar, nloc/)))
       end do
       if (ngpus > 1 .or. usecomm /= 0) then
          call chk(mcmcx_allreduce_moments_all(handles, int(ngpus, c_int32_t), pm))      ! RCCL: all GPUs' chains
       else
          call chk(mcmcx_pooled_moments(handle, pm))
       end if
       do j = 1, npar
          pooledmean(j) = pm(1 + j) / pm(1)
       end do
       do j = 1, npar
          do i = 1, j
             pooledcov(i,j) = (pm(1 + npar + j*(j-1)/2 + i) - pm(1) * pooledmean(i) * pooledmean(j)) / (pm(1) - 1.0_dbl)
             pooledcov(j,i) = pooledcov(i,j)
          end do
       end do
       pooledmean = pooledmean + par0
    end if
  end subroutine MCMC_engine_run

  !! ---------------------------------------------------------------- mcmc_main_one: MCMC_run1 / MCMC_run1_er
  !! One evaluation of ssfunction per program invocation; the chain's state lives in files between invocations
  !! (MCMC_run1.F90:14-29).  The files, the namelist /mcmcrun/ and the user's callbacks are the host's; the arithmetic
  !! of an invocation -- MCMC_alpha / MCMC_DR_alpha13, MCMC_reject, MCMC_propose, MCMC_sscrit -- runs in the engine
  !! (mcmcx_run1_*), one chain, on the factors of mcmcx_init (MCMC_init.F90:81-160: R, R2, iC of cmat0).
  subroutine MCMC_engine_create_one()
    type(mcmcx_config) :: cfg
    integer(c_int32_t), allocatable :: nob(:)
    call chk(mcmcx_install_signal_handlers())
    call read_seed_file()
    allocate(handles(1), comms(1))
    handles = c_null_ptr; comms = c_null_ptr
    call mcmcx_config_defaults(cfg)
    cfg%npar = npar; cfg%nchains = 1; cfg%nsimu = nsimu
    select case (trim(method))
    case ('ram');  cfg%method = 1
    case ('scam'); cfg%method = 2
    case ('er');   cfg%method = 3
    case default;  cfg%method = 0
    end select
    cfg%doadapt = doadapt; cfg%doburnin = doburnin; cfg%adaptint = adaptint; cfg%adapthist = adapthist
    cfg%badaptint = badaptint; cfg%adaptend = adaptend; cfg%initcmatn = initcmatn
    cfg%burnintime = burnintime; cfg%greedy = greedy; cfg%updatesigma = 0
    cfg%scalelimit = scalelimit; cfg%scalefactor = scalefactor; cfg%drscale = drscale
    cfg%N0 = N0; cfg%S02 = S02; cfg%condmax = condmax; cfg%alphatarget = alphatarget; cfg%nuparam = nuparam
    cfg%seed = seed; cfg%record_accept = 0; cfg%record_chain = 0; cfg%pooled = 0; cfg%device = 0; cfg%chain_id0 = 0
    allocate(nob(nycol)); nob = nobs
    call chk(mcmcx_create(cfg, handles(1)))
    handle = handles(1)
    call chk(mcmcx_set_par0(handle, par0, int(npar, c_int32_t)))
    call chk(mcmcx_set_cmat0(handle, cmat0, int(npar, c_int32_t)))
    call chk(mcmcx_set_sigma2nobs(handle, sigma2, nob, int(nycol, c_int32_t)))
    call chk(mcmcx_set_target_external(handle))          ! every evaluation is made here, by the user's functions
    call chk(mcmcx_init(handle))
    if (allocated(chaincmat)) deallocate(chaincmat, chainmean)
    allocate(chaincmat(npar,npar), chainmean(npar))
    chaincmat = cmat0; chainmean = par0; chainwsum = dble(initcmatn)      ! MCMC_init.F90:99-102
    if (allocated(chain)) deallocate(chain, sschain)
    allocate(chain(nsimu, npar+1), sschain(nsimu, nycol+1))
    chain = 0.0_dbl; sschain = 0.0_dbl
    stayed = 0; bndstayed = 0; draccepted = 0; drtries = 0; chainind = 0; simuind = 1
  end subroutine MCMC_engine_create_one

  subroutine touch_file(file)
    character(len=*), intent(in) :: file
    integer :: u, ios
    open(newunit=u, file=file, status='replace', iostat=ios)
    if (ios == 0) close(u)
  end subroutine touch_file
  subroutine remove_file(file)
    character(len=*), intent(in) :: file
    integer :: u, ios
    logical :: ex
    inquire(file=file, exist=ex)
    if (.not. ex) return
    open(newunit=u, file=file, status='old', iostat=ios)
    if (ios == 0) close(u, status='delete')
  end subroutine remove_file

  !! MCMC_adapt(1) at the end of an invocation (MCMC_run1.F90:223): with simuind = 1 the schedule of
  !! MCMC_adapt.F90:42-46 fires only for adaptint = 1 or badaptint = 1, on a "chain" of one row
  subroutine run1_adapt_guard()
    integer :: bad
    bad = badaptint
    if (bad <= 0) bad = adaptint
    if ((doadapt /= 0 .or. doburnin /= 0) .and. (adaptint == 1 .or. bad == 1)) &
         call doerror('mcmc_main_one: adaptint = 1 / badaptint = 1 (adaptation from a one-row chain) is not supported')
  end subroutine run1_adapt_guard

  !! MCMC_run1, MCMC_run1.F90:31-256
  subroutine MCMC_run1()
    use mcmcrun1
    real(kind=dbl) :: ss(nycol), newpar(npar), sspri(1), alpha(1), a12(1)
    real(kind=dbl) :: oldpar1(npar), oldpar2(npar), ssprev1(nycol), ssprev2(nycol), sspri1(1), sspri2(1), pad(npar)
    integer(c_int32_t) :: rej(1)
    logical :: reject, inbounds, dodr
    integer :: fstat
    integer(kind=4) :: n4, ny4
    interface
       function ssfunction(theta,npar,ny)
         integer*4 npar, ny
         real*8 theta(npar)
         real*8 ssfunction(ny)
       end function ssfunction
       function priorfun(theta,len)
         real*8 priorfun
         integer*4 len
         real*8 theta(len)
       end function priorfun
       function checkbounds(theta)
         real*8 theta(:)
         logical checkbounds
       end function checkbounds
    end interface
    n4 = npar; ny4 = nycol
    dodr = (drscale > 0.0_dbl) .and. trim(method) /= 'ram' .and. trim(method) /= 'scam'      ! mcmcinit.F90:303-322
    if (updatesigma == 1) then
       write(*,*) 'Warning: do not use updatesigma in ER!'
       updatesigma = 0
    end if
    call run1_adapt_guard()
    call read_mcmcrun_namelist()
    reject = .false.
    if (isimu == 1) then
       if (verbosity >= 0) write(*,*) 'first simulation'
       newpar = par0; oldpar1 = par0; oldpar2 = par0
       sspri(1) = priorfun(newpar, n4)
       ss = ssfunction(newpar, n4, ny4)
       ieval = ieval + 1
       if (verbosity >= 0) write(*,*) 'ss1 = ', real(ss), ' sspri1 = ', real(sspri(1))
       ssprev1 = ss; ssprev2 = ss
       isimu = isimu + 1
       nrej = 1
       call readdata(meanfile, par0, stat=fstat, uselock=.true.)            ! MCMC_run1.F90:86: continuing an older run
       if (fstat == 0 .and. verbosity > 0) write(*,*) 'note: read ', trim(meanfile), ' for the chain mean'
    else
       if (drstage > 1 .and. dodr) then
          call readdata('mcmcoldpar2.dat', oldpar2)
          call readdata('mcmcoldpar1.dat', oldpar1)
          call readdata('mcmcssprev2.dat', ssprev2)
          call readdata('mcmcssprev1.dat', ssprev1)
       else
          call readdata('mcmcparf.dat', oldpar1)
          call readdata('mcmcssprev1.dat', ssprev1)
          !! the reference leaves oldpar2 / ssprev2 unset here (locals, MCMC_run1.F90:44-45): set before use after an accept,
          !! but after a reject without DR its next proposal starts from whatever the stack holds; stated as what the
          !! protocol means, the last accepted point
          oldpar2 = oldpar1; ssprev2 = ssprev1
       end if
       call readdata(meanfile, par0, uselock=.true.)                        ! MCMC_run1.F90:108: the point to evaluate
       simuind = isimu
       newpar = par0
       sspri(1) = priorfun(newpar, n4)
       ss = ssfunction(newpar, n4, ny4)
       ieval = ieval + 1
       if (verbosity >= 0) write(*,*) 'isimu:', isimu, ' drstage:', drstage
       if (verbosity >= 0) write(*,*) 'oldpar1:', real(oldpar1), ' ssprev1:', real(ssprev1)
       !! MCMC_run1.F90:126,133: the old priors are recomputed -- from the ss vectors, as the reference writes it
       !! (MCMC_priorfun(ssprev1)); here the first nycol entries of an npar-vector, the rest zero
       pad = 0.0_dbl; pad(1:min(npar,nycol)) = ssprev1(1:min(npar,nycol))
       sspri1(1) = priorfun(pad, n4)
       sspri2(1) = 0.0_dbl
       a12(1) = alpha12
       if (drstage > 1 .and. dodr) then
          pad = 0.0_dbl; pad(1:min(npar,nycol)) = ssprev2(1:min(npar,nycol))
          sspri2(1) = priorfun(pad, n4)
       end if
       call chk(mcmcx_run1_decide(handle, int(drstage, c_int32_t), oldpar2, ssprev2, sspri2, oldpar1, ssprev1, sspri1, a12, &
            newpar, ss, sspri, alpha, rej))
       reject = (rej(1) /= 0)
       if (reject) then                                                      ! MCMC_run1.F90:145-172
          if (dodr) then
             if (drstage == 1) then
                drstage = 2
                ssprev2 = ssprev1; ssprev1 = ss
                oldpar2 = oldpar1; oldpar1 = newpar
                alpha12 = alpha(1)
             else
                isimu = isimu + 1
                drstage = 1
                ssprev1 = ssprev2; oldpar1 = oldpar2
             end if
          else
             isimu = isimu + 1
             drstage = 1
          end if
       else
          drstage = 1
          isimu = isimu + 1
          nrej = 1
          oldpar1 = newpar; ssprev1 = ss
          alpha12 = alpha(1)
          ssprev2 = ssprev1; oldpar2 = oldpar1
       end if
       if (verbosity >= 0) write(*,*) 'newpar:', real(newpar), 'ssnew:', real(ss), ' ', .not.reject, ' ', real(alpha(1))
    end if
    !! the next value: proposed until one falls inside the bounds (MCMC_run1.F90:180-212)
    inbounds = .false.
    nrej = 1
    do while (.not. inbounds)
       if (drstage > 1 .and. dodr) then
          call chk(mcmcx_run1_propose(handle, 2_c_int32_t, oldpar2, newpar))
       else
          call chk(mcmcx_run1_propose(handle, 1_c_int32_t, oldpar2, newpar))
       end if
       inbounds = checkbounds(newpar)
       if (.not. inbounds) then
          if (drstage == 1 .and. dodr) then
             drstage = 2
             oldpar1 = newpar
             ssprev1 = huge(ssprev1)
             alpha12 = 0.0_dbl
             if (verbosity >= 0) write(*,*) 'outbound, stage 1, par:', real(newpar)
          else
             drstage = 1
             isimu = isimu + 1
             nrej = nrej + 1
             oldpar1 = oldpar2; ssprev1 = ssprev2
             if (verbosity >= 0) write(*,*) 'outbound, rej:', nrej, 'par:', real(newpar)
          end if
       end if
    end do
    chain(1,1:npar) = oldpar2; sschain(1,1:nycol) = ssprev2
    chain(1,npar+1) = dble(nrej); sschain(1,nycol+1) = dble(nrej)
    chainind = 1; simuind = 1
    call write_mcmcrun_namelist()
    call writedata('mcmcparnew.dat', reshape(newpar, (/1, npar/)))
    call writedata('mcmcssprev1.dat', ssprev1)
    if (dodr) call writedata('mcmcoldpar2.dat', oldpar2)
    if (dodr) call writedata('mcmcoldpar1.dat', oldpar1)
    if (dodr) call writedata('mcmcssprev2.dat', ssprev2)
    if (reject) then
       call remove_file('mcmc_accepted'); call touch_file('mcmc_rejected')
    else
       call remove_file('mcmc_rejected'); call touch_file('mcmc_accepted')
    end if
    if (ieval >= nsimu) call touch_file('mcmc_run_done')
  end subroutine MCMC_run1

  !! MCMC_run1_er, MCMC_run1_er.F90:28-234: early rejection -- the threshold for the next point is drawn with the proposal
  subroutine MCMC_run1_er()
    use mcmcrun1
    real(kind=dbl) :: ss(nycol), newpar(npar), sspri, alpha
    real(kind=dbl) :: oldpar1(npar), ssprev1(nycol), sspri1(1), pad(npar), crit(1)
    logical :: reject, inbounds
    integer :: fstat
    integer(kind=4) :: n4, ny4
    interface
       function ssfunction(theta,npar,ny)
         integer*4 npar, ny
         real*8 theta(npar)
         real*8 ssfunction(ny)
       end function ssfunction
       function priorfun(theta,len)
         real*8 priorfun
         integer*4 len
         real*8 theta(len)
       end function priorfun
       function checkbounds(theta)
         real*8 theta(:)
         logical checkbounds
       end function checkbounds
    end interface
    n4 = npar; ny4 = nycol
    alpha = 0.0_dbl; sspri1 = 0.0_dbl
    drstage = 1
    if (updatesigma == 1) then
       write(*,*) 'Warning: do not use updatesigma in ER!'
       updatesigma = 0
    end if
    call run1_adapt_guard()
    call read_mcmcrun_namelist()
    drstage = 1
    reject = .false.
    if (isimu == 1) then
       if (verbosity >= 0) write(*,*) 'first simulation'
       if (verbosity >= 0) write(*,*) 'Early rejection version'
       newpar = par0; oldpar1 = par0
       sspri = priorfun(newpar, n4)
       ss = ssfunction(newpar, n4, ny4)
       ieval = ieval + 1
       if (verbosity >= 0) write(*,*) 'ss1 = ', real(ss), ' sspri1 = ', real(sspri)
       ssprev1 = ss
       sspri1(1) = sspri                          ! unset in the reference's first invocation (MCMC_run1_er.F90:43); the prior of the point just evaluated
       isimu = isimu + 1
       nrej = 1
       call readdata(meanfile, par0, stat=fstat, uselock=.true.)
       if (fstat == 0 .and. verbosity > 0) write(*,*) 'note: read ', trim(meanfile), ' for the chain mean'
    else
       call readdata('mcmcparf.dat', oldpar1)
       call readdata('mcmcssprev1.dat', ssprev1)
       call readdata(meanfile, par0, uselock=.true.)
       simuind = isimu
       newpar = par0
       sspri = priorfun(newpar, n4)
       ss = ssfunction(newpar, n4, ny4)
       ieval = ieval + 1
       if (verbosity >= 0) write(*,*) 'isimu:', isimu
       if (verbosity >= 0) write(*,*) 'oldpar1:', real(oldpar1), ' ssprev1:', real(ssprev1), ' sscrit', real(sscrit)
       pad = 0.0_dbl; pad(1:min(npar,nycol)) = ssprev1(1:min(npar,nycol))
       sspri1(1) = priorfun(pad, n4)                                         ! MCMC_run1_er.F90:126, as written there
       reject = (sum(ss/sigma2) >= sscrit)                                   ! :131-137
       if (reject) then
          isimu = isimu + 1
       else
          isimu = isimu + 1
          nrej = 1
          oldpar1 = newpar; ssprev1 = ss
          alpha12 = alpha
          sspri1(1) = sspri
       end if
       if (verbosity >= 0) write(*,*) 'newpar:', real(newpar), 'ssnew:', real(ss), ' ', .not.reject, ' ', real(alpha)
    end if
    inbounds = .false.
    nrej = 1
    do while (.not. inbounds)                                                ! :162-190
       call chk(mcmcx_run1_propose(handle, 1_c_int32_t, oldpar1, newpar))
       inbounds = checkbounds(newpar)
       if (.not. inbounds) then
          isimu = isimu + 1
          nrej = nrej + 1
          if (verbosity >= 0) write(*,*) 'outbound, rej:', nrej, 'par:', real(newpar)
       else
          sspri = priorfun(newpar, n4)
          call chk(mcmcx_run1_sscrit(handle, ssprev1, sspri1, crit))
          sscrit = crit(1)
          if (sspri >= sscrit) then
             inbounds = .false.
             isimu = isimu + 1
             nrej = nrej + 1
             if (verbosity >= 0) write(*,*) 'er, rej by prior:', nrej, 'par:', real(newpar)
          else
             sscrit = sscrit - sspri
          end if
       end if
    end do
    chain(1,1:npar) = oldpar1; sschain(1,1:nycol) = ssprev1
    chain(1,npar+1) = dble(nrej); sschain(1,nycol+1) = dble(nrej)
    chainind = 1; simuind = 1
    call write_mcmcrun_namelist()
    call writedata('mcmcparnew.dat', reshape(newpar, (/1, npar/)))
    call writedata('mcmcssprev1.dat', ssprev1)
    call writedata('mcmcsscrit.dat', sscrit)
    if (reject) then
       call remove_file('mcmc_accepted'); call touch_file('mcmc_rejected')
    else
       call remove_file('mcmc_rejected'); call touch_file('mcmc_accepted')
    end if
    if (ieval >= nsimu) call touch_file('mcmc_run_done')
  end subroutine MCMC_run1_er

  !! the progress line of MCMC_adapt.F90:22-37 / MCMC_run_ram.F90:118-122, from chain 1's counters
  subroutine progress_line(it)
    integer, intent(in) :: it
    integer(c_int32_t) :: c8(8)
    call chk(mcmcx_get_counters(handle, 0_c_int32_t, c8))
    if (drscale > 0.0_dbl .and. trim(method) == 'dram') then
       write(*,'(A,I10,A,F5.1,A,F5.1,A,F5.1,A)') ' simu i =', it, &
            ', stayed % = ', real(c8(1))/real(it)*100.0, &
            ' /', (1.0-real(c8(3))/real(max(c8(4),1)))*100.0, &
            '  [', real(c8(2))/real(it)*100.0, ']'
    else if (c8(7) > 0) then
       write(*,'(A,I10,A,F5.1,A,F5.1,A)') ' simu i =', it, &
            ', stayed % = ', real(c8(1))/real(it)*100.0, &
            '  [', real(c8(7))/real(it)*100.0, '](er)'
    else
       write(*,'(A,I10,A,F5.1,A,F5.1,A)') ' simu i =', it, &
            ', stayed % = ', real(c8(1))/real(it)*100.0, &
            '  [', real(c8(2))/real(it)*100.0, ']'
    end if
  end subroutine progress_line

  !! MAT-v4 file like writemat4_mat (matfiles.F90:66-126): 5 x int32 header (type 0, mrows, ncols, imagf 0, namelen),
  !! the name with a trailing NUL, the data column by column as 8-byte reals
  subroutine writemat4(file, a, aname)
    character(len=*), intent(in) :: file, aname
    real(kind=dbl), intent(in) :: a(:,:)
    integer :: u
    open(newunit=u, file=file, status='replace', form='unformatted', access='stream')
    write(u) 0_4, int(size(a,1),4), int(size(a,2),4), 0_4, int(len_trim(aname)+1,4)
    write(u) trim(aname)//achar(0)
    write(u) a
    close(u)
  end subroutine writemat4

  logical function ismat(file)
    character(len=*), intent(in) :: file
    integer :: n
    n = len_trim(file)
    ismat = .false.
    if (n >= 4) ismat = (file(n-3:n) == '.mat')
  end function ismat

  !! MCMC_writechains, MCMC_aux.F90:17-85
  subroutine MCMC_writechains()
    integer :: u
    if (ismat(chainfile)) then
       call writemat4(chainfile, chain(1:chainind,:), 'chain')
    else
       call writedata(chainfile, chain(1:chainind,:))
    end if
    if (ismat(ssfile)) then
       call writemat4(ssfile, sschain(1:chainind,:), 'sschain')
    else
       call writedata(ssfile, sschain(1:chainind,:))
    end if
    if (updatesigma /= 0) then
       if (ismat(s2file)) then
          call writemat4(s2file, s2chain(1:simuind,:), 's2chain')
       else
          call writedata(s2file, s2chain(1:simuind,:))
       end if
    end if
    !! the restart files, in the reference's order, shapes and locking (MCMC_aux.F90:46-62): covffile, covnfile and meanfile
    !! under their lock files; the mean as an npar x 1 matrix, the last point as a vector (one value per line)
    call writedata(covffile, chaincmat, uselock=.true.)
    if (len_trim(covnfile) > 0) then
       initcmatn = int(chainwsum)
       call writedata(covnfile, reshape((/dble(initcmatn)/), (/1,1/)), uselock=.true.)
    end if
    call writedata(meanfile, reshape(chainmean, (/npar, 1/)), uselock=.true.)
    call writedata(parffile, chain(chainind, 1:npar))
    if (updatesigma /= 0) call writedata(sigma2ffile, &
         transpose(reshape((/s2chain(simuind,1:nycol), dble(nobs)/), (/nycol, 2/))))
    if (nchains > 1) then                               ! engine extension: all chains' last states, pooled mean / covariance
       call writedata('mcmclaststates.dat', laststates)
       call writedata('mcmcpooledmean.dat', reshape(pooledmean, (/npar, 1/)))
       call writedata('mcmcpooledcov.dat', pooledcov)
    end if
    if (verbosity > 0) write(*,*) 'note: saved results in ', trim(chainfile), ' and ', trim(ssfile), '.'
    !! restart namelist, MCMC_aux.F90:78-83
    initcmatn = initcmatn + simuind
    burnintime = 0
    if (len_trim(nmlffile) > 0) then
       open(newunit=u, file=nmlffile, status='replace', delim='APOSTROPHE')
       write(u, nml=mcmc)
       close(u)
    end if
  end subroutine MCMC_writechains

  !! the stream's key lives in the reference's seed file (mcmcrand.F90:28-40, 214-238): an integer there replaces
  !! the &mcmcx seed, and the end of job (random_eoj, :317-342) leaves the next run's key, so reruns are
  !! reproducible and successive runs continue with fresh streams
  subroutine read_seed_file()
    integer :: u, ios, v
    logical :: ex
    seedfile_used = ''
    inquire(file='gfortran_seed.dat', exist=ex)
    if (ex) then
       seedfile_used = 'gfortran_seed.dat'
    else
       inquire(file='random_seed.dat', exist=ex)
       if (ex) seedfile_used = 'random_seed.dat'
    end if
    if (len_trim(seedfile_used) == 0) return
    open(newunit=u, file=seedfile_used, status='old', iostat=ios)
    if (ios /= 0) return
    read(u, *, iostat=ios) v
    close(u)
    if (ios == 0) then
       seed = v
       if (verbosity > 0) write(*,*) 'note: read the seed from '//trim(seedfile_used)
    end if
  end subroutine read_seed_file

  subroutine write_seed_file()
    integer :: u, ios
    integer(kind=8) :: nx
    if (len_trim(seedfile_used) == 0) return
    nx = mod(int(seed, 8) * 6364136223846793005_8 + 1442695040888963407_8 + int(simuind, 8), 2147483647_8)
    if (nx < 0) nx = -nx
    open(newunit=u, file=seedfile_used, status='replace', iostat=ios)
    if (ios /= 0) return
    write(u, *) int(nx)
    close(u)
  end subroutine write_seed_file

  logical function was_interrupted()
    was_interrupted = interrupted
  end function was_interrupted

  subroutine MCMC_cleanup()
    integer(c_int) :: rc
    integer :: g
    interface
       subroutine dump_end()
       end subroutine dump_end
    end interface
    call write_seed_file()                              ! random_eoj, MCMC_aux.F90:115
    call dump_end()                                     ! MCMC_dump_end, MCMC_aux.F90:116
    if (allocated(handles)) then
       do g = 1, size(handles)
          rc = mcmcx_destroy(handles(g))
          if (c_associated(comms(g))) rc = mcmcx_comm_destroy(comms(g))
       end do
       deallocate(handles, comms)
    end if
    handle = c_null_ptr
    nparok = .false.; par0ok = .false.; cmat0ok = .false.; sigma2ok = .false.
  end subroutine MCMC_cleanup

end module mcmcmod

!!! mcmc_main, mcmc_main.F90:12-44: external, no arguments
subroutine mcmc_main()
  use mcmcmod
  implicit none
  integer :: status
  write(*,*) 'MCMC code version: ', Mcmc_Code_Version
  call read_mcmcinit_namelist(status)
  if (status /= 0) call doerror('error in mcmcinit namelist')          ! MCMC_init.F90:35
  if (nsimu <= 0) then
     write(*,*) 'nsimu <= 0 stopping'
     stop
  end if
  call MCMC_initial_values()
  call load_target_from_files()
  call MCMC_engine_run()
  call MCMC_writechains()
  call MCMC_cleanup()
  if (was_interrupted()) stop 'Coltrol-c interrupt'      ! [sic] MCMC_signal_handler.F90:105
end subroutine mcmc_main

!!! mcmc_main_one, mcmc_main.F90:49-70: consecutive one-evaluation invocations through the file protocol of MCMC_run1 /
!!! MCMC_run1_er (state in mcmcrun.nml and the mcmc*.dat files between program runs).  Same sequence as the reference:
!!! MCMC_init, one MCMC_run1[_er], MCMC_writechains, MCMC_cleanup.
subroutine mcmc_main_one()
  use mcmcmod
  implicit none
  integer :: status
  write(*,*) 'MCMC code version: ', Mcmc_Code_Version
  call read_mcmcinit_namelist(status)
  if (status /= 0) call doerror('error in mcmcinit namelist')
  if (nsimu <= 0) stop 'nsimu <= 0'
  call MCMC_initial_values()
  if (trim(method) == 'er') then                         ! MCMC_run1_er.F90:62-66, before the engine takes sigma2
     if (any(abs(sigma2 - 1.0_dbl) > 1.0e-6_dbl)) then
        write(*,*) 'Warning: in ER sigma2 should be exactly 1'
        sigma2 = 1.0_dbl
     end if
  end if
  call MCMC_engine_create_one()
  if (trim(method) == 'er') then
     call MCMC_run1_er()
  else
     call MCMC_run1()
  end if
  call MCMC_writechains()
  call MCMC_cleanup()
end subroutine mcmc_main_one
