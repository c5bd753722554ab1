! Drop-in replacement for the reference's CalSurfG.o: exports the link symbols `calsurfg_` and
! `synthetic_` with the reference's argument lists (reference src/CalSurfG.f90:939-943 and
! :2412-2415; called from main.f90:338-342 and :355-359) and forwards them over ISO_C_BINDING to
! dsa_calsurfg / dsa_synthetic in libdsurftomo_amd.so (include/dsurftomo_amd.h).
!
! Link it instead of CalSurfG.o (surfdisp96.o is then unused as well):
!   flang ... main.o calsurfg_shim.o <other reference objects> -L<repo>/dsurftomo_amd -ldsurftomo_amd
!
! Where the reference prints a message and STOPs (a source or receiver outside the model), the
! engine returns an error; this shim prints the engine's text -- the same words -- and stops too.
module dsa_bindings
  use iso_c_binding
  implicit none
  interface
    integer(c_int) function dsa_calsurfg(nx,ny,nz,nparpi,vels,iw,rw,col,dsurf, &
        goxdf,gozdf,dvxdf,dvzdf,kmaxRc,kmaxRg,kmaxLc,kmaxLg,tRc,tRg,tLc,tLg, &
        wavetype,igrt,periods,depz,minthk,scxf,sczf,rcxf,rczf,nrc1,nsrcsurf1, &
        kmax,nsrcsurf,nrcf,nar) bind(C, name='dsa_calsurfg')
      import :: c_int, c_float, c_double
      integer(c_int) :: nx,ny,nz,nparpi,kmaxRc,kmaxRg,kmaxLc,kmaxLg,kmax,nsrcsurf,nrcf,nar
      real(c_float) :: vels(*),rw(*),dsurf(*),goxdf,gozdf,dvxdf,dvzdf,depz(*),minthk
      real(c_float) :: scxf(*),sczf(*),rcxf(*),rczf(*)
      integer(c_int) :: iw(*),col(*),wavetype(*),igrt(*),periods(*),nrc1(*),nsrcsurf1(*)
      real(c_double) :: tRc(*),tRg(*),tLc(*),tLg(*)
    end function
    integer(c_int) function dsa_synthetic(nx,ny,nz,nparpi,vels,obst, &
        goxdf,gozdf,dvxdf,dvzdf,kmaxRc,kmaxRg,kmaxLc,kmaxLg,tRc,tRg,tLc,tLg, &
        wavetype,igrt,periods,depz,minthk,scxf,sczf,rcxf,rczf,nrc1,nsrcsurf1, &
        kmax,nsrcsurf,nrcf,noiselevel) bind(C, name='dsa_synthetic')
      import :: c_int, c_float, c_double
      integer(c_int) :: nx,ny,nz,nparpi,kmaxRc,kmaxRg,kmaxLc,kmaxLg,kmax,nsrcsurf,nrcf
      real(c_float) :: vels(*),obst(*),goxdf,gozdf,dvxdf,dvzdf,depz(*),minthk,noiselevel
      real(c_float) :: scxf(*),sczf(*),rcxf(*),rczf(*)
      integer(c_int) :: wavetype(*),igrt(*),periods(*),nrc1(*),nsrcsurf1(*)
      real(c_double) :: tRc(*),tRg(*),tLc(*),tLg(*)
    end function
    integer(c_int) function dsa_dropin_velocity_maps(which, pv) bind(C, name='dsa_dropin_velocity_maps')
      import :: c_int, c_double
      integer(c_int) :: which
      real(c_double) :: pv(*)
    end function
    integer(c_int) function dsa_dropin_diagnostics(rbint_notes, disp_count, disp_first, disp_period) bind(C, name='dsa_dropin_diagnostics')
      import :: c_int, c_long_long, c_double
      integer(c_int) :: rbint_notes, disp_first(5)
      integer(c_long_long) :: disp_count
      real(c_double) :: disp_period
    end function
    integer(c_int) function dsa_dropin_tie_diagnostics(flagged, left, marched, influence) bind(C, name='dsa_dropin_tie_diagnostics')
      import :: c_int, c_long_long, c_float
      integer(c_long_long) :: flagged, left, marched
      real(c_float) :: influence
    end function
    integer(c_int) function dsa_dropin_tie_census(tied, prone, bymap) bind(C, name='dsa_dropin_tie_census')
      import :: c_int, c_long_long
      integer(c_long_long) :: tied, prone, bymap
    end function
    integer(c_int) function dsa_dropin_dispersion_failure(index, info, vals, table, c) bind(C, name='dsa_dropin_dispersion_failure')
      import :: c_int, c_double, c_float
      integer(c_int), value :: index
      integer(c_int) :: info(8)
      real(c_double) :: vals(4), c(60)
      real(c_float) :: table(200,4)
    end function
    function dsa_dropin_error() bind(C, name='dsa_dropin_error') result(p)
      import :: c_ptr
      type(c_ptr) :: p
    end function
    integer(c_size_t) function c_strlen(s) bind(C, name='strlen')
      import :: c_ptr, c_size_t
      type(c_ptr), value :: s
    end function
  end interface
contains
  subroutine dsa_stop(where)
    character(len=*), intent(in) :: where
    type(c_ptr) :: p
    character(kind=c_char), pointer :: msg(:)
    integer :: n, i
    p = dsa_dropin_error()
    n = int(c_strlen(p))
    call c_f_pointer(p, msg, [n])
    write(6,*) (msg(i), i = 1, n)
    write(6,*) 'TERMINATING PROGRAM!!!! (', where, ')'
    stop 1
  end subroutine
end module

subroutine CalSurfG(nx,ny,nz,nparpi,vels,iw,rw,col,dsurf, &
    goxdf,gozdf,dvxdf,dvzdf,kmaxRc,kmaxRg,kmaxLc,kmaxLg, &
    tRc,tRg,tLc,tLg,wavetype,igrt,periods,depz,minthk, &
    scxf,sczf,rcxf,rczf,nrc1,nsrcsurf1,kmax,nsrcsurf,nrcf,nar)
  use dsa_bindings
  implicit none
  integer nx,ny,nz,nparpi,kmax,nsrcsurf,nrcf,nar
  real vels(nx,ny,nz),rw(*),dsurf(*),goxdf,gozdf,dvxdf,dvzdf,depz(nz),minthk
  integer iw(*),col(*),kmaxRc,kmaxRg,kmaxLc,kmaxLg
  real*8 tRc(*),tRg(*),tLc(*),tLg(*)
  integer wavetype(nsrcsurf,kmax),periods(nsrcsurf,kmax),nrc1(nsrcsurf,kmax),nsrcsurf1(kmax),igrt(nsrcsurf,kmax)
  real scxf(nsrcsurf,kmax),sczf(nsrcsurf,kmax),rcxf(nrcf,nsrcsurf,kmax),rczf(nrcf,nsrcsurf,kmax)
  integer rc
  rc = dsa_calsurfg(nx,ny,nz,nparpi,vels,iw,rw,col,dsurf,goxdf,gozdf,dvxdf,dvzdf, &
       kmaxRc,kmaxRg,kmaxLc,kmaxLg,tRc,tRg,tLc,tLg,wavetype,igrt,periods,depz,minthk, &
       scxf,sczf,rcxf,rczf,nrc1,nsrcsurf1,kmax,nsrcsurf,nrcf,nar)
  if (rc /= 0) call dsa_stop('CalSurfG')
  call dsa_report()
  call dsa_report_ties('CalSurfG')
end subroutine

! Not messages of the reference: one line on unit 6 when a call left travel-time fields with exact time ties above the threshold to the
! fixed-point solve (DSA_EXACT_TIES=0 in the environment; with the default such fields are solved again by the reference's own march), and one
! line when fields with ties BELOW the threshold kept the fixed-point times (the default mode's caveat; DSA_TIE_NOTE=0 silences it).
subroutine dsa_report_ties(where)
  use dsa_bindings
  implicit none
  character(len=*), intent(in) :: where
  integer(c_long_long) :: nflag, nleft, nmarch, ntied, nprone, nbymap
  real(c_float) :: infl
  integer :: rc, stat
  character(len=16) :: note
  rc = dsa_dropin_tie_diagnostics(nflag, nleft, nmarch, infl)
  if (rc /= 0) return
  if (nleft > 0) then
    write(6,'(a,a,a,i0,a,es9.2,a)') ' dsurftomo_amd (', where, '): ', nleft, &
      ' (period, source) travel-time fields hold exact time ties (largest influence ', infl, &
      ' s) and were left to the fixed-point solve (DSA_EXACT_TIES=0): they may differ from Fast Marching by more than 1e-4 s'
  endif
  ! (round 6) the default mode's own caveat: fields whose ties all stay below the threshold keep the fixed-point solve's times -- the reference's
  ! to 1e-4 s by measurement, not by construction (one-ulp differences can grow along a ridge of the field).  One line, unless DSA_TIE_NOTE=0.
  rc = dsa_dropin_tie_census(ntied, nprone, nbymap)
  if (rc /= 0 .or. ntied <= 0) return
  call get_environment_variable('DSA_TIE_NOTE', note, status=stat)
  if (stat == 0 .and. trim(note) == '0') return
  write(6,'(a,a,a,i0,a)') ' dsurftomo_amd (', where, '): ', ntied, &
    ' (period, source) travel-time fields hold exact time ties of small influence and keep the fixed-point times: within 1e-4 s of' // &
    ' Fast Marching by measurement, not by construction (DSA_EXACT_TIES=2: the reference''s bits; DSA_TIE_NOTE=0: no note)'
end subroutine

! The reference's non-fatal messages, written where and how it writes them.
!  * unit 6, CalSurfG.f90:1447-1454: the six-line note about a ray tracked along the model boundary, once after every (period,
!    source) iteration from the first such ray on -- the count comes from the engine.
!  * unit 66 (the host program's log file, main.f90:156), surfdisp96.f:308-339: "improper initial value in disper - no zero found".
!    The reference writes the block once per failing surfdisp96 call together with that call's layer table.  With DSA_DISP_FAILURE_LOG=N
!    in the environment the engine keeps the first N failing calls in the reference's single-thread call order and this routine writes
!    the reference's block for each of them, line by line, layer table included (dsa_dropin_dispersion_failure replays the curve on
!    the host for the numbers); the last list holds c(1..k-1) and 0 for c(k), an element the reference prints without ever having
!    assigned it.  Without the variable (default): the text once per CalSurfG call for the first failing curve, followed by the number
!    of curves that ended this way.
subroutine dsa_report()
  use dsa_bindings
  implicit none
  integer :: notes, first(5), rc, i, j, nlog, info(8), ifunc, k, is, ie, mmax
  integer(c_long_long) :: ndisp
  real*8 :: period, vals(4), c(60), cc, cm, c1
  real*4 :: table(200,4)
  rc = dsa_dropin_diagnostics(notes, ndisp, first, period)
  if (rc /= 0) return
  do i = 1, notes
    WRITE(6,*)'Note that at least one two-point ray path'
    WRITE(6,*)'tracked along the boundary of the model.'
    WRITE(6,*)'This class of path is unlikely to be'
    WRITE(6,*)'a true path, and it is STRONGLY RECOMMENDED'
    WRITE(6,*)'that you adjust the dimensions of your grid'
    WRITE(6,*)'to prevent this from occurring.'
  enddo
  ! the failing calls one by one, as the reference writes them (surfdisp96.f:308-339), when the engine kept a log
  nlog = 0
  if (ndisp > 0) then
    rc = dsa_dropin_dispersion_failure(0, info, vals, table, c)
    if (rc == 0) nlog = info(8)
  endif
  do j = 1, nlog
    rc = dsa_dropin_dispersion_failure(j - 1, info, vals, table, c)
    if (rc /= 0) exit
    ifunc = info(1); k = info(5); is = 1; ie = info(6); mmax = info(7)
    cc = vals(2); cm = vals(3); c1 = vals(4)
    write(66,*)'improper initial value in disper - no zero found'
    write(66,*)'in fundamental mode '
    write(66,*)'This may be due to low velocity zone '
    write(66,*)'causing reverse phase velocity dispersion, '
    write(66,*)'and mode jumping.'
    write(66,*)'due to looking for Love waves in a halfspace'
    write(66,*)'which is OK if there are Rayleigh data.'
    write(66,*)'If reverse dispersion is the problem,'
    write(66,*)'Get present model using OPTION 28, edit sobs.d,'
    write(66,*)'Rerun with onel large than 2'
    write(66,*)'which is the default '
    write(66,*)'ifunc = ',ifunc ,' (1=L, 2=R)'
    write(66,*)'mode  = ',0
    write(66,*)'period= ',vals(1), ' for k,is,ie=',k,is,ie
    write(66,*)'cc,cm = ',cc,cm
    write(66,*)'c1    = ',c1
    write(66,*)'d,a,b,rho (d(mmax)=control ignore)'
    write(66,'(4f15.5)')(table(i,1),table(i,2),table(i,3),table(i,4),i=1,mmax)
    write(66,*)' c(i),i=1,k (NOTE may be part)'
    write(66,*)(c(i),i=1,k)
  enddo
  if (nlog > 0 .and. ndisp > nlog) write(66,*)'curves of this call that ended this way: ',ndisp,' (the first ',nlog,' written above)'
  if (ndisp > 0 .and. nlog == 0) then
    write(66,*)'improper initial value in disper - no zero found'
    write(66,*)'in fundamental mode '
    write(66,*)'This may be due to low velocity zone '
    write(66,*)'causing reverse phase velocity dispersion, '
    write(66,*)'and mode jumping.'
    write(66,*)'due to looking for Love waves in a halfspace'
    write(66,*)'which is OK if there are Rayleigh data.'
    write(66,*)'If reverse dispersion is the problem,'
    write(66,*)'Get present model using OPTION 28, edit sobs.d,'
    write(66,*)'Rerun with onel large than 2'
    write(66,*)'which is the default '
    write(66,*)'ifunc = ',first(1) ,' (1=L, 2=R)'
    write(66,*)'mode  = ',0
    write(66,*)'period= ',period, ' for k=',first(5)
    write(66,*)'grid column (jj-1)*nx+ii = ',first(3),' depth-kernel perturbation = ',first(4),' group velocity = ',first(2)
    write(66,*)'curves of this call that ended this way: ',ndisp
  endif
end subroutine

subroutine synthetic(nx,ny,nz,nparpi,vels,obst, &
    goxdf,gozdf,dvxdf,dvzdf,kmaxRc,kmaxRg,kmaxLc,kmaxLg, &
    tRc,tRg,tLc,tLg,wavetype,igrt,periods,depz,minthk, &
    scxf,sczf,rcxf,rczf,nrc1,nsrcsurf1,kmax,nsrcsurf,nrcf,noiselevel)
  use dsa_bindings
  implicit none
  integer nx,ny,nz,nparpi,kmax,nsrcsurf,nrcf
  real vels(nx,ny,nz),obst(*),goxdf,gozdf,dvxdf,dvzdf,depz(nz),minthk,noiselevel
  integer kmaxRc,kmaxRg,kmaxLc,kmaxLg
  real*8 tRc(*),tRg(*),tLc(*),tLg(*)
  integer wavetype(nsrcsurf,kmax),periods(nsrcsurf,kmax),nrc1(nsrcsurf,kmax),nsrcsurf1(kmax),igrt(nsrcsurf,kmax)
  real scxf(nsrcsurf,kmax),sczf(nsrcsurf,kmax),rcxf(nrcf,nsrcsurf,kmax),rczf(nrcf,nsrcsurf,kmax)
  real gaussian, zero
  external gaussian
  integer rc, knumi, srcnum, istep, count1, which, kk, i, j, k
  integer kper(4)
  real*8, allocatable :: pv(:,:)
  real*8 tper
  character(len=16) :: fname(4)
  ! noise-free times from the device, then the reference's own noise statement in the reference's
  ! loop order (CalSurfG.f90:2840), drawing from the host program's gaussian() like the original
  zero = 0.0
  rc = dsa_synthetic(nx,ny,nz,nparpi,vels,obst,goxdf,gozdf,dvxdf,dvzdf, &
       kmaxRc,kmaxRg,kmaxLc,kmaxLg,tRc,tRg,tLc,tLg,wavetype,igrt,periods,depz,minthk, &
       scxf,sczf,rcxf,rczf,nrc1,nsrcsurf1,kmax,nsrcsurf,nrcf,zero)
  if (rc /= 0) call dsa_stop('synthetic')
  call dsa_report_ties('synthetic')
  ! the 2-D velocity maps the reference writes for plotting (CalSurfG.f90:2559-2617), same statements
  kper = (/ kmaxRc, kmaxRg, kmaxLc, kmaxLg /)
  fname = (/ 'velmap2dRc.dat  ', 'velmap2dRg.dat  ', 'velmap2dLc.dat  ', 'velmap2dLg.dat  ' /)
  do kk = 1, 4
    if (kper(kk) <= 0) cycle
    allocate(pv(nx*ny, kper(kk)))
    which = kk - 1
    rc = dsa_dropin_velocity_maps(which, pv)
    if (rc /= 0) call dsa_stop('synthetic')
    open(62, file=trim(fname(kk)))
    do k = 1, kper(kk)
      if (kk == 1) tper = tRc(k)
      if (kk == 2) tper = tRg(k)
      if (kk == 3) tper = tLc(k)
      if (kk == 4) tper = tLg(k)
      do j = 1, ny-2
        do i = 1, nx-2
          write(62,'(5f8.4)') gozdf+(j-1)*dvzdf, goxdf-(i-1)*dvxdf, tper, pv((j+1)*nx+i+1,k)
        enddo
      enddo
    enddo
    close(62)
    deallocate(pv)
  enddo
  count1 = 0
  do knumi = 1, kmax
    do srcnum = 1, nsrcsurf1(knumi)
      do istep = 1, nrc1(srcnum,knumi)
        count1 = count1 + 1
        obst(count1) = obst(count1) + obst(count1)*gaussian()*noiselevel
      enddo
    enddo
  enddo
end subroutine
