!!! matutils.F90 -- the file I/O helpers of mcmcf90's `matutils` that user programs call
!!! (testcases/mcmcrun*.F90: loaddata, readdata, sizecheck_mat), under the same generic names and argument
!!! lists (matutils.F90:32-47, 764, 841, 1007, 1338, 1829), written for the engine's shim.
!!! File format as the reference reads it (matutils.F90:1019-1056): numbers separated by blanks, commas,
!!! semicolons or tabs, one matrix row per line; blank lines and lines starting with # % ! c C are skipped.
!!! Written as the reference writes it (matutils.F90:57, 889-899, 950): every element through the edit descriptor
!!! G0 -- a matrix row as '(G0,x)' per element and a newline, a vector one '(G0)' per line -- so that, compiled by
!!! the same compiler, a given double leaves the same bytes.  `uselock` is the reference's lock-file protocol
!!! (matutils.F90:1544-1680): wait up to 10 s for FILE.lock to disappear, create it, work on FILE, remove it.
module matutils
  use mcmcprec
  implicit none
  private
  public :: loaddata, loaddata2, readdata, writedata, sizecheck_mat, sizecheck_vec, doerror
  public :: loadnumbers, lock_acquire, lock_release
  real, parameter :: lock_timeout = 10.0               ! seconds (matutils.F90:1573)

  interface loaddata
     module procedure loaddata_mat, loaddata_vec, loaddata_x, loaddata_int, loaddata_ints
  end interface
  interface loaddata2
     module procedure loaddata_mata, loaddata_veca
  end interface
  interface readdata
     module procedure readdata_vec, readdata_mat, readdata_n, readdata_x
  end interface
  interface writedata
     module procedure writedata_mat, writedata_vec, writedata_scal
  end interface

contains

  !! message + stop, like doerror (matutils.F90:764-789); elevel / action kept for source compatibility
  subroutine doerror(creason, elevel, action)
    character(len=*), intent(in) :: creason
    integer, intent(in), optional :: elevel, action
    write(*,*) 'ERROR: ', trim(creason)
    stop 1
  end subroutine doerror

  !! FILE.lock: wait for another process's lock to go (polling once a second, with the reference's message, for at
  !! most lock_timeout seconds), then take it.  stat: 0 = taken, -1 = still locked after the timeout, > 0 = the lock file
  !! could not be created.  (matutils.F90:1585-1629)
  subroutine lock_acquire(file, stat)
    character(len=*), intent(in) :: file
    integer, intent(out) :: stat
    integer :: t0, t1, rate, u, ios
    real :: waited
    logical :: there
    stat = 0
    call system_clock(count_rate=rate)
    call system_clock(count=t0)
    do
       inquire(file=trim(file)//'.lock', exist=there)
       if (.not. there) exit
       call system_clock(count=t1)
       waited = real(t1 - t0) / real(rate)
       if (waited > lock_timeout) then
          stat = -1
          return
       end if
       write(*,*) 'waiting for lock file ...'//'time left:', lock_timeout - waited
       call sleep(1)
    end do
    open(newunit=u, file=trim(file)//'.lock', status='replace', iostat=ios)
    if (ios /= 0) then
       stat = ios
       return
    end if
    close(u)
  end subroutine lock_acquire

  subroutine lock_release(file)
    character(len=*), intent(in) :: file
    integer :: u, ios
    open(newunit=u, file=trim(file)//'.lock', status='old', iostat=ios)
    if (ios == 0) close(u, status='delete', iostat=ios)
  end subroutine lock_release

  !! all numbers of a text file, row by row; nrows / ncols as found (ncols of the first row); uselock: under FILE.lock
  subroutine loadnumbers(file, v, nrows, ncols, stat, uselock)
    character(len=*), intent(in) :: file
    real(kind=dbl), allocatable, intent(out) :: v(:)
    integer, intent(out) :: nrows, ncols, stat
    logical, intent(in), optional :: uselock
    logical :: locked
    character(len=8192) :: line
    real(kind=dbl) :: tmp(4096)
    real(kind=dbl), allocatable :: buf(:), nb(:)
    integer :: u, ios, n, i, k, ntot
    character(len=1) :: c
    stat = 0; nrows = 0; ncols = 0; ntot = 0
    locked = .false.
    if (present(uselock)) locked = uselock
    if (locked) then
       call lock_acquire(file, ios)
       if (ios /= 0) then
          stat = ios; allocate(v(0)); return
       end if
    end if
    allocate(buf(1024))
    open(newunit=u, file=file, status='old', iostat=ios)
    if (ios /= 0) then
       if (locked) call lock_release(file)
       stat = -1; allocate(v(0)); return
    end if
    do
       read(u, '(A)', iostat=ios) line
       if (ios /= 0) exit
       line = adjustl(line)
       if (len_trim(line) == 0) cycle
       c = line(1:1)
       if (c == '#' .or. c == '%' .or. c == '!' .or. c == 'C' .or. c == 'c') cycle
       do i = 1, len_trim(line)
          if (line(i:i) == ',' .or. line(i:i) == ';' .or. line(i:i) == achar(9)) line(i:i) = ' '
       end do
       n = 0
       do k = 1, 4096                               ! count the numbers on this line
          read(line, *, iostat=ios) tmp(1:k)
          if (ios /= 0) exit
          n = k
       end do
       if (n == 0) cycle
       if (ntot + n > size(buf)) then
          allocate(nb(2*size(buf) + n)); nb(1:ntot) = buf(1:ntot); call move_alloc(nb, buf)
       end if
       buf(ntot+1:ntot+n) = tmp(1:n)
       ntot = ntot + n
       nrows = nrows + 1
       if (nrows == 1) ncols = n
    end do
    close(u)
    if (locked) call lock_release(file)
    allocate(v(ntot)); v = buf(1:ntot)
    if (nrows > 0 .and. ncols*nrows /= ntot) ncols = ntot / nrows
  end subroutine loadnumbers

  subroutine failed(file, status)
    character(len=*), intent(in) :: file
    integer, intent(out), optional :: status
    if (present(status)) then
       status = -1
    else
       call doerror('Error reading file, '//trim(file))
    end if
  end subroutine failed

  !! ---- loaddata: the array is allocated here (pointer arguments like the reference's)
  subroutine loaddata_mat(file, xmat, status, uselock)
    character(len=*), intent(in) :: file
    real(kind=dbl), pointer :: xmat(:,:)
    integer, intent(out), optional :: status
    logical, intent(in), optional :: uselock
    real(kind=dbl), allocatable :: v(:)
    integer :: nr, nc, st
    if (present(status)) status = 0
    call loadnumbers(file, v, nr, nc, st, uselock)
    if (st /= 0 .or. nr < 1) then
       nullify(xmat); call failed(file, status); return
    end if
    allocate(xmat(nr, nc))
    xmat = transpose(reshape(v(1:nr*nc), (/nc, nr/)))
  end subroutine loaddata_mat

  subroutine loaddata_vec(file, xvec, status, uselock)
    character(len=*), intent(in) :: file
    real(kind=dbl), pointer :: xvec(:)
    integer, intent(out), optional :: status
    logical, intent(in), optional :: uselock
    real(kind=dbl), allocatable :: v(:)
    integer :: nr, nc, st
    if (present(status)) status = 0
    call loadnumbers(file, v, nr, nc, st, uselock)
    if (st /= 0 .or. size(v) < 1) then
       nullify(xvec); call failed(file, status); return
    end if
    allocate(xvec(size(v)))
    xvec = v                                         ! any shape, flattened row by row
  end subroutine loaddata_vec

  subroutine loaddata_x(file, x, status, uselock)
    character(len=*), intent(in) :: file
    real(kind=dbl), intent(out) :: x
    integer, intent(out), optional :: status
    logical, intent(in), optional :: uselock
    real(kind=dbl), allocatable :: v(:)
    integer :: nr, nc, st
    if (present(status)) status = 0
    x = 0.0_dbl
    call loadnumbers(file, v, nr, nc, st, uselock)
    if (st /= 0 .or. size(v) < 1) then
       call failed(file, status); return
    end if
    x = v(1)
  end subroutine loaddata_x

  subroutine loaddata_ints(file, n, status, uselock)
    character(len=*), intent(in) :: file
    integer, intent(out) :: n
    integer, intent(out), optional :: status
    logical, intent(in), optional :: uselock
    real(kind=dbl) :: x
    call loaddata_x(file, x, status, uselock)
    n = int(x)
  end subroutine loaddata_ints

  subroutine loaddata_int(file, nvec, status, uselock)
    character(len=*), intent(in) :: file
    integer, pointer :: nvec(:)
    integer, intent(out), optional :: status
    logical, intent(in), optional :: uselock
    real(kind=dbl), pointer :: xv(:)
    call loaddata_vec(file, xv, status, uselock)
    if (.not.associated(xv)) then
       nullify(nvec); return
    end if
    allocate(nvec(size(xv))); nvec = int(xv); deallocate(xv)
  end subroutine loaddata_int

  subroutine loaddata_mata(file, xmat, status, uselock)
    character(len=*), intent(in) :: file
    real(kind=dbl), intent(inout), allocatable :: xmat(:,:)
    integer, intent(out), optional :: status
    logical, intent(in), optional :: uselock
    real(kind=dbl), pointer :: p(:,:)
    call loaddata_mat(file, p, status, uselock)
    if (.not.associated(p)) return
    if (allocated(xmat)) deallocate(xmat)
    allocate(xmat(size(p,1), size(p,2))); xmat = p; deallocate(p)
  end subroutine loaddata_mata

  subroutine loaddata_veca(file, xvec, status, uselock)
    character(len=*), intent(in) :: file
    real(kind=dbl), intent(inout), allocatable :: xvec(:)
    integer, intent(out), optional :: status
    logical, intent(in), optional :: uselock
    real(kind=dbl), pointer :: p(:)
    call loaddata_vec(file, p, status, uselock)
    if (.not.associated(p)) return
    if (allocated(xvec)) deallocate(xvec)
    allocate(xvec(size(p))); xvec = p; deallocate(p)
  end subroutine loaddata_veca

  !! ---- readdata: into an array the caller has sized; a size mismatch is an error
  subroutine readdata_vec(file, par, stat, uselock)
    real(kind=dbl), intent(out) :: par(:)
    character(*), intent(in) :: file
    integer, intent(out), optional :: stat
    logical, intent(in), optional :: uselock
    real(kind=dbl), allocatable :: v(:)
    integer :: nr, nc, st
    if (present(stat)) stat = 0
    call loadnumbers(file, v, nr, nc, st, uselock)
    if (st /= 0 .or. size(v) /= size(par)) then
       call failed(file, stat); return
    end if
    par = v
  end subroutine readdata_vec

  subroutine readdata_mat(file, xmat, stat, uselock)
    real(kind=dbl), intent(out) :: xmat(:,:)
    character(*), intent(in) :: file
    integer, intent(out), optional :: stat
    logical, intent(in), optional :: uselock
    real(kind=dbl), allocatable :: v(:)
    integer :: nr, nc, st
    if (present(stat)) stat = 0
    call loadnumbers(file, v, nr, nc, st, uselock)
    if (st /= 0 .or. nr /= size(xmat,1) .or. nc /= size(xmat,2)) then
       call failed(file, stat); return
    end if
    xmat = transpose(reshape(v, (/nc, nr/)))
  end subroutine readdata_mat

  subroutine readdata_n(file, n, stat, uselock)
    integer, intent(out) :: n
    character(*), intent(in) :: file
    integer, intent(out), optional :: stat
    logical, intent(in), optional :: uselock
    call loaddata_ints(file, n, stat, uselock)
  end subroutine readdata_n

  subroutine readdata_x(file, x, stat, uselock)
    real(kind=dbl), intent(out) :: x
    character(*), intent(in) :: file
    integer, intent(out), optional :: stat
    logical, intent(in), optional :: uselock
    call loaddata_x(file, x, stat, uselock)
  end subroutine readdata_x

  !! ---- writedata (matutils.F90:841-967): a matrix row by row, every element as '(G0,x)' without advancing, then the end
  !! of the record; a vector one element per line as '(G0)'.  An empty file name sets stat = -1 and writes nothing; an
  !! open error is returned in stat when the caller asked for it and stops the program otherwise.
  subroutine writedata_mat(file, xmat, stat, uselock)
    character(len=*), intent(in) :: file
    real(kind=dbl), intent(in) :: xmat(:,:)
    integer, intent(out), optional :: stat
    logical, intent(in), optional :: uselock
    integer :: u, i, j, ios
    logical :: locked
    if (len_trim(file) == 0) then
       if (present(stat)) stat = -1
       return
    end if
    if (size(xmat,1) < 1 .or. size(xmat,2) < 1) then
       write(*,*) 'Error in writedata, empty matrix, file:', trim(file)
       return
    end if
    locked = .false.
    if (present(uselock)) locked = uselock
    ios = 0
    if (locked) call lock_acquire(file, ios)
    if (ios == 0) then
       open(newunit=u, file=file, status='replace', iostat=ios)
       if (ios /= 0 .and. locked) call lock_release(file)
    end if
    if (ios /= 0) then
       if (present(stat)) then
          stat = ios
          return
       end if
       write(*,*) 'Error opening file ', trim(file)
       stop
    end if
    do i = 1, size(xmat,1)
       do j = 1, size(xmat,2)
          write(u, '(G0,x)', iostat=ios, advance='NO') xmat(i,j)
          if (ios /= 0) then
             write(*,*) 'Write error on file ', trim(file)
             stop
          end if
       end do
       write(u, *)
    end do
    close(u, iostat=ios)
    if (locked) call lock_release(file)
    if (present(stat)) stat = ios
  end subroutine writedata_mat

  subroutine writedata_vec(file, xvec, stat)
    character(len=*), intent(in) :: file
    real(kind=dbl), intent(in) :: xvec(:)
    integer, intent(out), optional :: stat
    integer :: u, i, ios
    if (len_trim(file) == 0) then
       if (present(stat)) stat = -1
       return
    end if
    open(newunit=u, file=file, status='replace', iostat=ios)
    if (ios /= 0) then
       if (present(stat)) then
          stat = ios
          return
       end if
       write(*,*) 'Error opening file ', trim(file)
       stop
    end if
    do i = 1, size(xvec)
       write(u, '(G0)', iostat=ios) xvec(i)
       if (ios /= 0) then
          write(*,*) 'Write error on file ', trim(file)
          stop
       end if
    end do
    close(u)
    if (present(stat)) stat = ios
  end subroutine writedata_vec

  subroutine writedata_scal(file, x, stat)
    character(len=*), intent(in) :: file
    real(kind=dbl), intent(in) :: x
    integer, intent(out), optional :: stat
    call writedata_vec(file, (/x/), stat)
  end subroutine writedata_scal

  !! ---- shape checks (matutils.F90:1829-1860): wrong size is fatal
  subroutine sizecheck_mat(xmat, n1, n2, txt)
    real(kind=dbl), intent(in) :: xmat(:,:)
    integer, intent(in) :: n1, n2
    character(len=*), intent(in), optional :: txt
    if (size(xmat,1) /= n1 .or. size(xmat,2) /= n2) then
       if (present(txt)) then
          call doerror('wrong matrix size: '//trim(txt))
       else
          call doerror('wrong matrix size')
       end if
    end if
  end subroutine sizecheck_mat

  subroutine sizecheck_vec(xvec, n1, txt)
    real(kind=dbl), intent(in) :: xvec(:)
    integer, intent(in) :: n1
    character(len=*), intent(in), optional :: txt
    if (size(xvec) /= n1) then
       if (present(txt)) then
          call doerror('wrong vector size: '//trim(txt))
       else
          call doerror('wrong vector size')
       end if
    end if
  end subroutine sizecheck_vec

end module matutils
