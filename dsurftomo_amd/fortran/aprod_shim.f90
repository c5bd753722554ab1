! Optional second drop-in: `aprod_` (reference src/aprod.f90:7, called by LSMR at lsmrModule.f90:390,
! :486, :497) forwarded to dsa_aprod, which keeps the matrix on the GPU across the hundreds of products
! of one LSMR solve and adds every output element's entries in the reference's storage order (same
! bits).  Link it INSTEAD of aprod.o:
!   OBJS = ... calsurfg_shim.o aprod_shim.o main.o      (drop CalSurfG.o, surfdisp96.o and aprod.o)
subroutine aprod(mode, m, n, x, y, leniw, lenrw, iw, rw)
  use iso_c_binding
  implicit none
  integer mode, m, n, leniw, lenrw
  real x(n), y(m)
  integer iw(leniw)
  real rw(lenrw)
  interface
    integer(c_int) function dsa_aprod(mode, m, n, x, y, leniw, lenrw, iw, rw) bind(C, name='dsa_aprod')
      import :: c_int, c_float
      integer(c_int) :: mode, m, n, leniw, lenrw, iw(*)
      real(c_float) :: x(*), y(*), rw(*)
    end function
  end interface
  integer rc
  rc = dsa_aprod(mode, m, n, x, y, leniw, lenrw, iw, rw)
  if (rc /= 0) then
    write(6,*) 'aprod: the device matrix-vector product failed'
    stop 1
  endif
end subroutine
