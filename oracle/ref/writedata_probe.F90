!!! writedata_probe.F90 -- TEST INFRASTRUCTURE.  A user-side program of module matutils' ASCII writers: reads
!!! wd_in.bin (int32 m, int32 n, m*n doubles, row by row) and writes the same numbers with writedata as a matrix
!!! (wd_mat.dat), under the lock-file protocol (wd_lock.dat), as a vector (wd_vec.dat) and as a scalar (wd_scal.dat).
!!! Compiled against the REAL reference's matutils (oracle/_ref/wd_ref, oracle/Makefile) to produce the byte fixture
!!! tests/golden/io/writedata.npz, and against the engine's shim (mcmcf90_amd/fortran/demo_writedata) to be compared with it.
program writedata_probe
  use matutils, only : writedata, loaddata
  implicit none
  integer(kind=4) :: m, n
  integer :: i, j, u, stat
  real(kind=8), allocatable :: flat(:), a(:,:)
  real(kind=8), pointer :: back(:,:)
  open(newunit=u, file='wd_in.bin', access='stream', form='unformatted', status='old')
  read(u) m, n
  allocate(flat(m*n), a(m,n))
  read(u) flat
  close(u)
  do i = 1, m
     do j = 1, n
        a(i,j) = flat((i-1)*n + j)
     end do
  end do
  call writedata('wd_mat.dat', a)
  call writedata('wd_lock.dat', a, stat, uselock=.true.)
  call writedata('wd_vec.dat', flat)
  call writedata('wd_scal.dat', flat(1))
  !! read back under the lock protocol: leaves wd_back.bin with what loaddata parsed
  call loaddata('wd_lock.dat', back, stat, uselock=.true.)
  open(newunit=u, file='wd_back.bin', access='stream', form='unformatted', status='replace')
  write(u) int(size(back,1), 4), int(size(back,2), 4)
  do i = 1, size(back,1)
     write(u) back(i,:)
  end do
  close(u)
end program writedata_probe
