! Optional third drop-in: module lsmrModule with subroutine LSMR (reference src/lsmrModule.f90:36, called at
! main.f90:487 through `use lsmrModule, only: lsmr`) forwarded to dsa_lsmr_dropin: the whole LSMR solve runs on the
! GPU with the matrix and all vectors resident, sums in the reference's order (same bits as the reference's LSMR).
! Build it INSTEAD of lsmrModule.f90 (it provides the same module name and the same argument list), before main.f90:
!   OBJS = lsmrDataModule.o lsmrblasInterface.o lsmrblas.o lsmr_shim.o ... calsurfg_shim.o main.o
! (lsmrblas stays: main.f90:22 uses dnrm2 from it; aprod.o is no longer referenced by LSMR.)
module lsmrModule
  use iso_c_binding
  implicit none
  private
  public :: LSMR
contains
  subroutine LSMR(m, n, leniw, lenrw, iw, rw, b, damp, atol, btol, conlim, itnlim, localSize, nout, &
                  x, istop, itn, normA, condA, normr, normAr, normx)
    integer, intent(in) :: leniw, lenrw
    integer, intent(in) :: iw(leniw)
    real, intent(in) :: rw(lenrw)
    integer, intent(in) :: m, n, itnlim, localSize, nout
    integer, intent(out) :: istop, itn
    real, intent(in) :: b(m)
    real, intent(out) :: x(n)
    real, intent(in) :: atol, btol, conlim, damp
    real, intent(out) :: normA, condA, normr, normAr, normx
    interface
      integer(c_int) function dsa_lsmr_dropin(m, n, leniw, lenrw, iw, rw, b, damp, atol, btol, conlim, itnlim, &
                                              localSize, nout, x, istop, itn, normA, condA, normr, normAr, normx) &
          bind(C, name='dsa_lsmr_dropin')
        import :: c_int, c_float
        integer(c_int) :: m, n, leniw, lenrw, iw(*), itnlim, localSize, nout, istop, itn
        real(c_float) :: rw(*), b(*), damp, atol, btol, conlim, x(*), normA, condA, normr, normAr, normx
      end function
    end interface
    integer :: rc
    rc = dsa_lsmr_dropin(m, n, leniw, lenrw, iw, rw, b, damp, atol, btol, conlim, itnlim, localSize, nout, &
                         x, istop, itn, normA, condA, normr, normAr, normx)
    if (rc /= 0) then
      write(6,*) 'LSMR: the device solve failed'
      stop 1
    endif
  end subroutine LSMR
end module lsmrModule
