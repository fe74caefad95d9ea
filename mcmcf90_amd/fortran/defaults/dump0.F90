!!! dump0.F90 -- default (empty) dump hooks, overridable at link time (reference dump.F90:13-26)
subroutine dump_init()
  implicit none
end subroutine dump_init

subroutine dump_end()
  implicit none
end subroutine dump_end

subroutine dump(oldpar)
  use mcmcprec
  implicit none
  real(kind=dbl), intent(in) :: oldpar(:)
end subroutine dump
