! Test driver for dsurftomo_amd/fortran/calsurfg_shim.f90: reads one CalSurfG argument set from a
! flat binary file (written by tests/test_gpu_boundary.py), calls CalSurfG and synthetic through the
! shim's link symbols exactly as the reference's host program does (main.f90:338-359), and writes
! the outputs back.  Own code; no reference source involved.
program shim_driver
  use lsmrModule, only: lsmr          ! dsurftomo_amd/fortran/lsmr_shim.f90, like main.f90:21
  implicit none
  integer :: nx,ny,nz,nparpi,kmaxRc,kmaxRg,kmaxLc,kmaxLg,kmax,nsrcsurf,nrcf,ndata,maxnar,nar,i,leniw
  real, allocatable :: xv(:), yv(:), bv(:), dv(:)
  integer :: istop, itn, nout
  real :: damp, atol, btol, conlim, anorm, acond, rnorm, arnorm, xnorm
  real :: goxd,gozd,dvxd,dvzd,minthk,noiselevel
  real, allocatable :: vels(:,:,:),depz(:),scxf(:,:),sczf(:,:),rcxf(:,:,:),rczf(:,:,:),rw(:),dsurf(:),obst(:)
  real*8, allocatable :: tRc(:),tRg(:),tLc(:),tLg(:)
  integer, allocatable :: wavetype(:,:),igrt(:,:),periods(:,:),nrc1(:,:),nsrcsurf1(:),iw(:),col(:)
  character(len=512) :: fin, fout
  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(21,file=trim(fin),access='stream',form='unformatted',status='old')
  read(21) nx,ny,nz,kmaxRc,kmaxRg,kmaxLc,kmaxLg,kmax,nsrcsurf,nrcf,ndata,maxnar
  read(21) goxd,gozd,dvxd,dvzd,minthk
  nparpi = (nx-2)*(ny-2)*(nz-1)
  allocate(vels(nx,ny,nz),depz(nz),tRc(max(kmaxRc,1)),tRg(max(kmaxRg,1)),tLc(max(kmaxLc,1)),tLg(max(kmaxLg,1)))
  allocate(wavetype(nsrcsurf,kmax),igrt(nsrcsurf,kmax),periods(nsrcsurf,kmax),nrc1(nsrcsurf,kmax),nsrcsurf1(kmax))
  allocate(scxf(nsrcsurf,kmax),sczf(nsrcsurf,kmax),rcxf(nrcf,nsrcsurf,kmax),rczf(nrcf,nsrcsurf,kmax))
  allocate(rw(maxnar),iw(2*maxnar+1),col(maxnar),dsurf(ndata),obst(ndata))
  read(21) vels, depz
  if (kmaxRc > 0) read(21) tRc(1:kmaxRc)
  if (kmaxRg > 0) read(21) tRg(1:kmaxRg)
  if (kmaxLc > 0) read(21) tLc(1:kmaxLc)
  if (kmaxLg > 0) read(21) tLg(1:kmaxLg)
  read(21) wavetype, igrt, periods, nrc1, nsrcsurf1, scxf, sczf, rcxf, rczf
  close(21)
  write(6,*) 'inputs', nx,ny,nz,kmax,nsrcsurf,nrcf,ndata,maxnar,sum(nrc1),sum(nsrcsurf1),sum(scxf),sum(rcxf),minthk,depz
  iw = 0
  rw = 0.0
  col = 0
  noiselevel = 0.0
  call synthetic(nx,ny,nz,nparpi,vels,obst,goxd,gozd,dvxd,dvzd,kmaxRc,kmaxRg,kmaxLc,kmaxLg, &
       tRc,tRg,tLc,tLg,wavetype,igrt,periods,depz,minthk,scxf,sczf,rcxf,rczf,nrc1,nsrcsurf1,kmax, &
       nsrcsurf,nrcf,noiselevel)
  call CalSurfG(nx,ny,nz,nparpi,vels,iw,rw,col,dsurf,goxd,gozd,dvxd,dvzd,kmaxRc,kmaxRg,kmaxLc,kmaxLg, &
       tRc,tRg,tLc,tLg,wavetype,igrt,periods,depz,minthk,scxf,sczf,rcxf,rczf,nrc1,nsrcsurf1,kmax, &
       nsrcsurf,nrcf,nar)
  open(22,file=trim(fout),access='stream',form='unformatted',status='replace')
  write(22) nar
  write(22) dsurf, obst
  write(22) rw(1:nar), (iw(1+i), i=1,nar), col(1:nar)
  ! the matrix-vector products the way the host program sets them up (main.f90:457-461, lsmrModule.f90:390-497)
  iw(1) = nar
  do i = 1, nar
    iw(1+nar+i) = col(i)
  enddo
  leniw = 2*nar + 1
  allocate(xv(nparpi), yv(ndata))
  do i = 1, nparpi
    xv(i) = real(mod(i*7, 13) - 6) * 0.125
  enddo
  do i = 1, ndata
    yv(i) = real(mod(i*5, 11) - 5) * 0.25
  enddo
  call aprod(1, ndata, nparpi, xv, yv, leniw, nar, iw, rw)
  call aprod(2, ndata, nparpi, xv, yv, leniw, nar, iw, rw)
  write(22) xv, yv
  ! LSMR the way main.f90:470-489 calls it (module procedure of lsmrModule), on the same matrix
  allocate(bv(ndata), dv(nparpi))
  do i = 1, ndata
    bv(i) = real(mod(i*3, 17) - 8) * 0.01
  enddo
  damp = 1.0; atol = 1e-6; btol = 1e-6; conlim = 100; nout = 0
  dv = 0
  call LSMR(ndata, nparpi, leniw, nar, iw, rw, bv, damp, atol, btol, conlim, 400, 10, nout, &
            dv, istop, itn, anorm, acond, rnorm, arnorm, xnorm)
  write(22) istop, itn
  write(22) dv, anorm, acond, rnorm, arnorm, xnorm
  close(22)
end program

! the reference's host program provides gaussian() (gaussian.f90); the shim's synthetic calls it
real function gaussian()
  gaussian = 0.0
end function
