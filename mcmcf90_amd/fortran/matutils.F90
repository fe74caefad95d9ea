!!! matutils.F90 -- the file I/O helpers of mcmcf90's `matutils` that user programs call
!!! (testcases/mcmcrun*.F90: loaddata, readdata, sizecheck_mat), under the same generic names and argument
!!! lists (matutils.F90:32-47, 764, 841, 1007, 1338, 1829), written for the engine's shim.
!!! File format as the reference reads it (matutils.F90:1019-1056): numbers separated by blanks, commas,
!!! semicolons or tabs, one matrix row per line; blank lines and lines starting with # % ! c C are skipped.
!!! `uselock` (the reference's .lock protocol) is accepted and ignored.
module matutils
  use mcmcprec
  implicit none
  private
  public :: loaddata, loaddata2, readdata, writedata, sizecheck_mat, sizecheck_vec, doerror
  public :: loadnumbers, writenumbers

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

  !! all numbers of a text file, row by row; nrows / ncols as found (ncols of the first row)
  subroutine loadnumbers(file, v, nrows, ncols, stat)
    character(len=*), intent(in) :: file
    real(kind=dbl), allocatable, intent(out) :: v(:)
    integer, intent(out) :: nrows, ncols, stat
    character(len=8192) :: line
    real(kind=dbl) :: tmp(4096)
    real(kind=dbl), allocatable :: buf(:), nb(:)
    integer :: u, ios, n, i, k, ntot
    character(len=1) :: c
    stat = 0; nrows = 0; ncols = 0; ntot = 0
    allocate(buf(1024))
    open(newunit=u, file=file, status='old', iostat=ios)
    if (ios /= 0) then
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
    allocate(v(ntot)); v = buf(1:ntot)
    if (nrows > 0 .and. ncols*nrows /= ntot) ncols = ntot / nrows
  end subroutine loadnumbers

  subroutine writenumbers(file, a, stat)
    character(len=*), intent(in) :: file
    real(kind=dbl), intent(in) :: a(:,:)
    integer, intent(out), optional :: stat
    integer :: u, i, ios
    open(newunit=u, file=file, status='replace', iostat=ios)
    if (present(stat)) stat = ios
    if (ios /= 0) return
    do i = 1, size(a,1)
       write(u, '(*(ES24.16E3,1X))') a(i,:)
    end do
    close(u)
  end subroutine writenumbers

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
    call loadnumbers(file, v, nr, nc, st)
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
    call loadnumbers(file, v, nr, nc, st)
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
    call loadnumbers(file, v, nr, nc, st)
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
    call loadnumbers(file, v, nr, nc, st)
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
    call loadnumbers(file, v, nr, nc, st)
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

  !! ---- writedata
  subroutine writedata_mat(file, xmat, stat, uselock)
    character(len=*), intent(in) :: file
    real(kind=dbl), intent(in) :: xmat(:,:)
    integer, intent(out), optional :: stat
    logical, intent(in), optional :: uselock
    call writenumbers(file, xmat, stat)
  end subroutine writedata_mat

  subroutine writedata_vec(file, xvec, stat)
    character(len=*), intent(in) :: file
    real(kind=dbl), intent(in) :: xvec(:)
    integer, intent(out), optional :: stat
    call writenumbers(file, reshape(xvec, (/size(xvec), 1/)), stat)
  end subroutine writedata_vec

  subroutine writedata_scal(file, x, stat)
    character(len=*), intent(in) :: file
    real(kind=dbl), intent(in) :: x
    integer, intent(out), optional :: stat
    call writenumbers(file, reshape((/x/), (/1, 1/)), stat)
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
