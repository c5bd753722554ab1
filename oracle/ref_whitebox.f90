! TEST INFRASTRUCTURE ONLY -- never linked into the product library.
!
! White-box C-callable handles onto the *reference's own* compiled objects
! (built from /root/reference/src by oracle/Makefile into oracle/_ref/).
! Everything numerical here is executed by the reference's subroutines
! (gridder, bsplrefine, travel, srtimes, rpaths; CalSurfG.f90:1460,1562,288,
! 1636,1771); this file only sequences them the way the CalSurfG driver does
! (CalSurfG.f90:1032-1096 set-up, :1186-1356 per-source stage sequence) and
! copies module arrays in and out so that tests can look at full fields.
module ref_whitebox
  use iso_c_binding
  use globalp
  use traveltime
  implicit none
  integer, save :: c_nnx = 0, c_nnz = 0          ! coarse dims kept across the refined stage
  real, allocatable, save :: inj_t(:,:)          ! coarse field right before travel(urg=2)
  integer, allocatable, save :: inj_s(:,:)
contains

  subroutine wb_release() bind(C, name='ref_wb_release')
    if (allocated(velv)) deallocate(velv)
    if (allocated(veln)) deallocate(veln)
    if (allocated(velnb)) deallocate(velnb)
    if (allocated(ttn)) deallocate(ttn)
    if (allocated(ttnr)) deallocate(ttnr)
    if (allocated(nsts)) deallocate(nsts)
    if (allocated(nstsr)) deallocate(nstsr)
    if (allocated(btg)) deallocate(btg)
    if (allocated(inj_t)) deallocate(inj_t)
    if (allocated(inj_s)) deallocate(inj_s)
  end subroutine

  ! grid set-up with the driver's constants; dicing = 8 for CalSurfG, 5 for synthetic
  subroutine wb_init(nx, ny, oxd, ozd, sxd, szd, dicing) bind(C, name='ref_wb_init')
    integer(c_int), value :: nx, ny, dicing
    real(c_float), value :: oxd, ozd, sxd, szd
    call wb_release()
    gdx = dicing; gdz = dicing
    asgr = 1; sgdl = 8; earth = 6371.0; fom = 1; snb = 0.5
    goxd = oxd; gozd = ozd; dvxd = sxd; dvzd = szd
    nvx = nx - 2; nvz = ny - 2
    allocate(velv(0:nvz+1, 0:nvx+1))
    dvx = dvxd*pi/180.0
    dvz = dvzd*pi/180.0
    gox = (90.0 - goxd)*pi/180.0
    goz = gozd*pi/180.0
    nnx = (nvx - 1)*gdx + 1
    nnz = (nvz - 1)*gdz + 1
    dnx = dvx/gdx
    dnz = dvz/gdz
    dnxd = dvxd/gdx
    dnzd = dvzd/gdz
    c_nnx = nnx; c_nnz = nnz
    allocate(veln(nnz, nnx), ttn(nnz, nnx), nsts(nnz, nnx))
    allocate(btg(nint(snb*nnx*nnz)))
    rbint = 0
  end subroutine

  subroutine wb_dims(onnx, onnz, ogox, ogoz, odnx, odnz) bind(C, name='ref_wb_dims')
    integer(c_int) :: onnx, onnz
    real(c_float) :: ogox, ogoz, odnx, odnz
    onnx = nnx; onnz = nnz; ogox = gox; ogoz = goz; odnx = dnx; odnz = dnz
  end subroutine

  subroutine wb_gridder(pv) bind(C, name='ref_wb_gridder')
    real(c_double) :: pv(*)
    call gridder(pv)
  end subroutine

  ! out is (nnz,nnx) column-major == C row-major [ix][iz]
  subroutine wb_get_veln(out) bind(C, name='ref_wb_get_veln')
    real(c_float) :: out(c_nnz, c_nnx)
    out = veln(1:c_nnz, 1:c_nnx)
  end subroutine

  subroutine wb_set_veln(inp) bind(C, name='ref_wb_set_veln')
    real(c_float) :: inp(c_nnz, c_nnx)
    veln(1:c_nnz, 1:c_nnx) = inp
  end subroutine

  ! one source: refined stage, hand-off, coarse stage (sequence of CalSurfG.f90:1192-1356)
  subroutine wb_solve(sx, sz) bind(C, name='ref_wb_solve')
    real(c_float), value :: sx, sz
    real :: x, z, kgox, kgoz, kdnx, kdnz
    integer :: knnx, knnz, isx, isz, sgs, m1, m2, k, l, a, b
    x = sx; z = sz; sgs = 8
    if (allocated(ttnr)) deallocate(ttnr)
    if (allocated(nstsr)) deallocate(nstsr)
    if (allocated(velnb)) deallocate(velnb)
    allocate(velnb(nnz, nnx))
    velnb(1:nnz, 1:nnx) = veln(1:nnz, 1:nnx)
    knnx = nnx; knnz = nnz; kdnx = dnx; kdnz = dnz; kgox = gox; kgoz = goz
    isx = int((x - gox)/dnx) + 1
    isz = int((z - goz)/dnz) + 1
    if (isx < 1 .or. isx > nnx .or. isz < 1 .or. isz > nnz) stop 'ref_wb_solve: source outside grid'
    if (isx == nnx) isx = isx - 1
    if (isz == nnz) isz = isz - 1
    vnl = max(isx - sgs, 1); vnr = min(isx + sgs, nnx)
    vnt = max(isz - sgs, 1); vnb = min(isz + sgs, nnz)
    nrnx = (vnr - vnl)*sgdl + 1
    nrnz = (vnb - vnt)*sgdl + 1
    drnx = dvx/real(gdx*sgdl)
    drnz = dvz/real(gdz*sgdl)
    gorx = gox + dnx*(vnl - 1)
    gorz = goz + dnz*(vnt - 1)
    nnx = nrnx; nnz = nrnz; dnx = drnx; dnz = drnz; gox = gorx; goz = gorz
    m1 = max(nnx, knnx); m2 = max(nnz, knnz)
    if (nnx > knnx .or. nnz > knnz) then
      deallocate(veln, ttn, nsts, btg)
      allocate(veln(m2, m1), ttn(m2, m1), nsts(m2, m1))
      allocate(btg(nint(snb*m1*m2)))
    end if
    call bsplrefine
    call travel(x, z, 1)
    allocate(ttnr(m2, m1), nstsr(m2, m1))
    ttnr = ttn
    nstsr = nsts
    nsts = -1
    do k = 1, nnz, sgdl
      a = vnt + (k - 1)/sgdl
      do l = 1, nnx, sgdl
        b = vnl + (l - 1)/sgdl
        nsts(a, b) = nstsr(k, l)
        if (nsts(a, b) >= 0) ttn(a, b) = ttnr(k, l)
      end do
    end do
    nnxr = nnx; nnzr = nnz; goxr = gox; gozr = goz; dnxr = dnx; dnzr = dnz
    nnx = knnx; nnz = knnz; dnx = kdnx; dnz = kdnz; gox = kgox; goz = kgoz
    veln(1:nnz, 1:nnx) = velnb(1:nnz, 1:nnx)
    do k = 1, nnx
      do l = 1, nnz
        if (nsts(l, k) == 0) then
          if (l > 1) then
            if (nsts(l-1, k) == -1) nsts(l, k) = 1
          end if
          if (l < nnz) then
            if (nsts(l+1, k) == -1) nsts(l, k) = 1
          end if
          if (k > 1) then
            if (nsts(l, k-1) == -1) nsts(l, k) = 1
          end if
          if (k < nnx) then
            if (nsts(l, k+1) == -1) nsts(l, k) = 1
          end if
        end if
      end do
    end do
    if (allocated(inj_t)) deallocate(inj_t, inj_s)
    allocate(inj_t(nnz, nnx), inj_s(nnz, nnx))
    inj_t = ttn(1:nnz, 1:nnx)
    inj_s = nsts(1:nnz, 1:nnx)
    call travel(x, z, 2)
  end subroutine

  subroutine wb_get_ttn(out) bind(C, name='ref_wb_get_ttn')
    real(c_float) :: out(c_nnz, c_nnx)
    out = ttn(1:c_nnz, 1:c_nnx)
  end subroutine

  subroutine wb_get_injected(ot, os) bind(C, name='ref_wb_get_injected')
    real(c_float) :: ot(c_nnz, c_nnx)
    integer(c_int) :: os(c_nnz, c_nnx)
    ot = inj_t
    os = inj_s
  end subroutine

  subroutine wb_refined_dims(onx, onz, ogox, ogoz, odnx, odnz, ovnl, ovnr, ovnt, ovnb) &
      bind(C, name='ref_wb_refined_dims')
    integer(c_int) :: onx, onz, ovnl, ovnr, ovnt, ovnb
    real(c_float) :: ogox, ogoz, odnx, odnz
    onx = nnxr; onz = nnzr; ogox = goxr; ogoz = gozr; odnx = dnxr; odnz = dnzr
    ovnl = vnl; ovnr = vnr; ovnt = vnt; ovnb = vnb
  end subroutine

  ! refined snapshot, (nnzr,nnxr) column-major
  subroutine wb_get_refined(ot, os) bind(C, name='ref_wb_get_refined')
    real(c_float) :: ot(nnzr, nnxr)
    integer(c_int) :: os(nnzr, nnxr)
    ot = ttnr(1:nnzr, 1:nnxr)
    os = nstsr(1:nnzr, 1:nnxr)
  end subroutine

  function wb_srtimes(sx, sz, rx, rz) result(t) bind(C, name='ref_wb_srtimes')
    real(c_float), value :: sx, sz, rx, rz
    real(c_float) :: t
    real :: a, b, c, d, e
    a = sx; b = sz; c = rx; d = rz
    call srtimes(a, b, c, d, e)
    t = e
  end function

  ! fdm is (0:nvz+1, 0:nvx+1) column-major
  subroutine wb_rpaths(sx, sz, rx, rz, fdm) bind(C, name='ref_wb_rpaths')
    real(c_float), value :: sx, sz, rx, rz
    real(c_float) :: fdm(0:nvz+1, 0:nvx+1)
    real :: a, b, c, d
    a = sx; b = sz; c = rx; d = rz
    call rpaths(a, b, fdm, c, d)
  end subroutine

  function wb_rbint() result(r) bind(C, name='ref_wb_rbint')
    integer(c_int) :: r
    r = rbint
  end function
end module
