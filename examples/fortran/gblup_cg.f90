! GBLUP-style ridge system (Zc Zc^T + lambda I) x = y solved by conjugate gradients from Fortran, on the ADDITIVE entry points of libmiraculix_amd.so
! (miraculix_amd/bindings/fortran/modmiraculix_amd.f90): the library stages the .bed file itself (mxa_bed2compressed) and every iteration is ONE fused call
! (mxa_gram_matvec: the snps-long intermediate never leaves the device).  The reference's counterpart is examples/iterative_solver/grm_solve_cg.jl:74-84, which
! calls dgemm_compressed 'T' then 'N' per iteration; the solution is verified here exactly that way, through the reference's own entries, and -- for small data --
! against a dense product on genotypes decoded in this file.
!   gblup_cg.out <file.bed> [lambda / snps (default 1.0)] [max iterations (default 200)]
! .bim / .fam must sit next to the .bed (their line counts give the dimensions).  Exit status 0 = every check passed.
program gblup_cg
 use, intrinsic :: iso_c_binding, only: c_int, c_long, c_double, c_ptr, c_null_ptr, c_null_char, c_associated
 use, intrinsic :: iso_fortran_env, only: int8, int64, real64
 use modmiraculix_amd
 implicit none
 character(len=512) :: bedfile, arg
 real(c_double) :: lam_rel, lambda, rr, rr_new, pap, alpha, beta, ynorm, relres, t_iter
 real(c_double), allocatable :: f(:), y(:,:), x(:,:), r(:,:), p(:,:), ap(:,:), t(:,:), gx(:,:), gx2(:,:)
 integer(c_int) :: rc, snps, indiv, maxit, it, i
 integer(int64) :: c0, c1, crate
 type(c_ptr) :: obj
 logical :: ok

 if (command_argument_count() < 1) then
  print '(a)', 'usage: gblup_cg.out <file.bed> [lambda / snps] [max iterations]'
  error stop 2
 end if
 call get_command_argument(1, bedfile)
 lam_rel = 1.0_c_double
 maxit = 200
 if (command_argument_count() >= 2) then
  call get_command_argument(2, arg); read(arg, *) lam_rel
 end if
 if (command_argument_count() >= 3) then
  call get_command_argument(3, arg); read(arg, *) maxit
 end if

 ! options as the reference's GPU harness sets them (utils/benchmark/benchmark.f90:222), centred, quiet
 call c_setOptions_compressed(1_c_int, 0_c_int, 0_c_int, 0_c_int, 1_c_int, 0_c_int, 0_c_int, 0_c_int, 32_c_int, 0_c_int)
 obj = c_null_ptr
 rc = mxa_bed2compressed(trim(bedfile)//c_null_char, 0_c_int, 0_c_int, 1_c_int, obj, c_null_ptr, snps, indiv)
 if (rc /= 0 .or. .not. c_associated(obj)) then
  print '(a,i0)', 'mxa_bed2compressed failed, mxa_last_error = ', mxa_last_error()
  error stop 1
 end if
 print '(a,i0,a,i0,a,i0,a,i0)', 'object: ', snps, ' SNPs x ', indiv, ' individuals, packed copies kept: ', merge(1, 2, mxa_single_orientation(obj) == 1), &
       ', devices visible: ', mxa_device_count()
 allocate(f(snps)); call c_get_compressed_freq(obj, f)
 lambda = lam_rel * real(snps, c_double)

 allocate(y(indiv,1), x(indiv,1), r(indiv,1), p(indiv,1), ap(indiv,1), gx(indiv,1), gx2(indiv,1), t(snps,1))
 do i = 1, indiv
  y(i,1) = sin(0.37_c_double * i) + 0.1_c_double * cos(1.3_c_double * i)
 end do
 ynorm = sqrt(sum(y * y))

 ! ---- conjugate gradients, x0 = 0
 x = 0; r = y; p = r; rr = sum(r * r)
 call system_clock(c0, crate)
 do it = 1, maxit
  if (mxa_gram_matvec(obj, 1_c_int, p, int(indiv, c_long), ap, int(indiv, c_long)) /= 0) error stop 1
  ap = ap + lambda * p
  pap = sum(p * ap)
  alpha = rr / pap
  x = x + alpha * p
  r = r - alpha * ap
  rr_new = sum(r * r)
  if (sqrt(rr_new) <= 1e-11_c_double * ynorm) exit
  beta = rr_new / rr
  p = r + beta * p
  rr = rr_new
 end do
 call system_clock(c1)
 it = min(it, maxit)
 t_iter = real(c1 - c0, c_double) / real(crate, c_double) / real(it, c_double)
 print '(a,i0,a,es10.3,a,f8.3,a)', 'CG: ', it, ' iterations, recurrence residual / |y| = ', sqrt(rr_new) / ynorm, ', ', 1e3_c_double * t_iter, ' ms per iteration (host vectors)'
 print '(a,i0)', 'kernel family of the last product (2 = exact int8 route): ', mxa_last_path()

 ok = .true.
 ! ---- check 1: the true residual through the REFERENCE entries, the way grm_solve_cg.jl multiplies: 'T' then 'N'
 call c_dgemm_compressed('T', obj, 1_c_int, x, indiv, t, snps)
 call c_dgemm_compressed('N', obj, 1_c_int, t, snps, gx, indiv)
 relres = sqrt(sum((y - gx - lambda * x)**2)) / ynorm
 print '(a,es10.3)', 'true residual / |y| through dgemm_compressed T then N: ', relres
 if (.not. (relres <= 1e-8_c_double)) then
  print '(a)', 'FAIL: residual'; ok = .false.
 end if
 ! ---- check 2: the fused step is bit-identical to its two products
 if (mxa_gram_matvec(obj, 1_c_int, x, int(indiv, c_long), gx2, int(indiv, c_long)) /= 0) error stop 1
 if (any(gx2 /= gx)) then
  print '(a,es10.3)', 'FAIL: mxa_gram_matvec differs from T then N, max |diff| = ', maxval(abs(gx2 - gx)); ok = .false.
 else
  print '(a)', 'mxa_gram_matvec == dgemm_compressed T then N, bit for bit'
 end if
 ! ---- check 3 (small data): dense product on genotypes decoded here
 if (real(snps, real64) * real(indiv, real64) <= 6.0e7_real64) call dense_check()

 call c_free_compressed(obj)
 if (c_associated(obj)) then
  print '(a)', 'FAIL: free_compressed left the handle set'; ok = .false.
 end if
 if (.not. ok) error stop 1
 print '(a)', 'PASS'

contains

 ! Zc = Z - 2 1 f^T with the decode of the multiply (00 -> 0, 10 -> 1, 11 -> 2, missing 01 -> 0 before centring); G x = Zc (Zc^T x)
 subroutine dense_check()
  integer(int8), allocatable :: raw(:,:)
  real(real64), allocatable :: zc(:,:), tt(:), gd(:)
  integer(int8) :: magic(3)
  integer :: un, s, j, bps, code
  real(real64), parameter :: val(0:3) = [0._real64, 0._real64, 1._real64, 2._real64]
  real(real64) :: err
  bps = (indiv + 3) / 4
  allocate(raw(bps, snps), zc(indiv, snps), tt(snps), gd(indiv))
  open(newunit=un, file=trim(bedfile), access='stream', status='old', action='read')
  read(un) magic
  read(un) raw
  close(un)
  do s = 1, snps
   do j = 1, indiv
    code = ibits(int(raw((j - 1) / 4 + 1, s)), 2 * mod(j - 1, 4), 2)
    zc(j, s) = val(code) - 2._real64 * f(s)
   end do
  end do
  tt = matmul(transpose(zc), x(:,1))
  gd = matmul(zc, tt)
  err = maxval(abs(gd - gx(:,1))) / maxval(abs(gd))
  print '(a,es10.3)', 'dense check of G x (genotypes decoded in Fortran): max relative difference ', err
  if (.not. (err <= 1e-11_real64)) then
   print '(a)', 'FAIL: dense check'; ok = .false.
  end if
 end subroutine

end program gblup_cg
