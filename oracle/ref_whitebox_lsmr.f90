! TEST INFRASTRUCTURE ONLY.  C-callable handle on the reference's own LSMR (lsmrModule.f90:36, a module procedure,
! hence this wrapper); everything numerical is executed by the reference's objects.  nout = 0: no printing,
! like main.f90:47,107 where the unit is never opened.
subroutine ref_wb_lsmr(m, n, leniw, lenrw, iw, rw, b, damp, atol, btol, conlim, itnlim, localSize, &
                       x, istop, itn, normA, condA, normr, normAr, normx) bind(C, name='ref_wb_lsmr')
  use iso_c_binding
  use lsmrModule, only: lsmr
  implicit none
  integer(c_int) :: m, n, leniw, lenrw, itnlim, localSize, istop, itn
  integer(c_int) :: iw(leniw)
  real(c_float) :: rw(lenrw), b(m), x(n), damp, atol, btol, conlim, normA, condA, normr, normAr, normx
  integer :: nout
  nout = 0
  call LSMR(m, n, leniw, lenrw, iw, rw, b, damp, atol, btol, conlim, itnlim, localSize, nout, &
            x, istop, itn, normA, condA, normr, normAr, normx)
end subroutine
