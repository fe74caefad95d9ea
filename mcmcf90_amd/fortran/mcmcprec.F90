!!! mcmcprec.F90 -- kind parameters under the names user code of mcmcf90 imports (`use mcmcprec`,
!!! reference mcmcprec.F90:14-36): dbl / ik4 are what the callback interfaces (external_inc.h) are written in.
module mcmcprec
  use iso_c_binding, only : c_double, c_int32_t, c_int8_t, c_int16_t, c_float
  implicit none
  private
  integer, parameter, public :: Byte = c_int8_t, Short = c_int16_t, Long = c_int32_t
  integer, parameter, public :: Single = c_float, Double = c_double
  integer, parameter, public :: ik = Short, rk = Double
  integer, parameter, public :: dbl = Double, ik4 = Long
  real(kind=dbl), parameter, public :: log_realmin = -708.396418532264_dbl      ! log(tiny(0d0))
end module mcmcprec
