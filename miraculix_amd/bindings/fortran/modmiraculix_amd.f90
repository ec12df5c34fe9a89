! Fortran binding of libmiraculix_amd.so: what a maintainer of the reference would put beside src/bindings/Fortran/mod5codesapi.f90 to reach the ADDITIVE
! entry points of include/miraculix_amd.h (part 2) from Fortran.  The five reference entries keep the interfaces of mod5codesapi.f90:22-82 (same C
! symbols, same argument kinds) and are repeated here under the same names so that a program needs this one module only.
! Conventions: `compressed` is the opaque object (type(c_ptr)); leading dimensions of the mxa_* entries are C long (integer(c_long)); file names are
! NUL-terminated character arrays (trim(name)//c_null_char); optional C pointers are passed as type(c_ptr) by value (c_loc(x) or c_null_ptr).
! Built and run by tests/test_fortran_binding_gpu.py through examples/fortran/gblup_cg.f90.
module modmiraculix_amd
 use, intrinsic :: iso_c_binding, only: c_int, c_long, c_double, c_char, c_ptr
 implicit none
 private
 ! reference entries (5codesAPI.c:37-161)
 public :: c_setOptions_compressed, c_plink2compressed, c_dgemm_compressed, c_get_compressed_freq, c_free_compressed
 ! additive entries
 public :: mxa_last_error, mxa_device_count, mxa_bed2compressed, mxa_gram_matvec, mxa_set_engine, mxa_get_engine, mxa_last_path
 public :: mxa_single_orientation, mxa_num_shards, mxa_allele_freq, mxa_transpose_2bit
 public :: mxa_plink2compressed_begin, mxa_plink2compressed_rows, mxa_plink2compressed_end

 interface
  subroutine c_setOptions_compressed(use_gpu, cores, floatLoop, meanSubstract, ignore_missings, do_not_center, do_normalize, use_miraculix_freq, variant, print_details) &
             bind(C, name='setOptions_compressed')
   import c_int
   integer(c_int), value, intent(in) :: use_gpu, cores, floatLoop, meanSubstract, ignore_missings, do_not_center, do_normalize, use_miraculix_freq, variant, print_details
  end subroutine

  subroutine c_plink2compressed(plink, plink_transposed, snps, indiv, f, n, compressed) bind(C, name='plink2compressed')
   import c_int, c_ptr, c_double
   type(c_ptr), value, intent(in) :: plink, plink_transposed      ! plink_transposed may be c_null_ptr: one packed copy is kept anyway
   integer(c_int), value, intent(in) :: snps, indiv, n
   real(c_double), intent(in) :: f(*)
   type(c_ptr), intent(out) :: compressed
  end subroutine

  subroutine c_dgemm_compressed(trans, compressed, n, B, ldb, C, ldc) bind(C, name='dgemm_compressed')
   import c_char, c_int, c_double, c_ptr
   character(c_char), intent(in) :: trans(*)
   type(c_ptr), value, intent(in) :: compressed
   integer(c_int), value, intent(in) :: n, ldb, ldc
   real(c_double), intent(in) :: B(ldb, *)
   real(c_double), intent(inout) :: C(ldc, *)
  end subroutine

  subroutine c_get_compressed_freq(compressed, freq) bind(C, name='get_compressed_freq')
   import c_double, c_ptr
   type(c_ptr), value, intent(in) :: compressed
   real(c_double), intent(out) :: freq(*)
  end subroutine

  subroutine c_free_compressed(compressed) bind(C, name='free_compressed')
   import c_ptr
   type(c_ptr), intent(inout) :: compressed                          ! c_null_ptr afterwards
  end subroutine

  ! ---- additive (include/miraculix_amd.h part 2)
  function mxa_last_error() bind(C, name='mxa_last_error') result(code)   ! 0 = the most recent fallible call succeeded
   import c_int
   integer(c_int) :: code
  end function

  function mxa_device_count() bind(C, name='mxa_device_count') result(n)
   import c_int
   integer(c_int) :: n
  end function

  ! .bed staging owned by the library; snps / indiv <= 0: from the line counts of the .bim / .fam next to the file.  f_out: c_loc of snps doubles, or c_null_ptr
  function mxa_bed2compressed(bed_path, snps, indiv, max_n, compressed, f_out, snps_out, indiv_out) bind(C, name='mxa_bed2compressed') result(rc)
   import c_char, c_int, c_ptr
   character(c_char), intent(in) :: bed_path(*)
   integer(c_int), value, intent(in) :: snps, indiv, max_n
   type(c_ptr), intent(out) :: compressed
   type(c_ptr), value, intent(in) :: f_out
   integer(c_int), intent(out) :: snps_out, indiv_out
   integer(c_int) :: rc
  end function

  ! out (indiv x n) = Zc (Zc^T V): one step of the GRM solvers, the snps x n intermediate stays on the device
  function mxa_gram_matvec(compressed, n, V, ldv, out, ldo) bind(C, name='mxa_gram_matvec') result(rc)
   import c_int, c_long, c_double, c_ptr
   type(c_ptr), value, intent(in) :: compressed
   integer(c_int), value, intent(in) :: n
   integer(c_long), value, intent(in) :: ldv, ldo
   real(c_double), intent(in) :: V(ldv, *)
   real(c_double), intent(inout) :: out(ldo, *)
   integer(c_int) :: rc
  end function

  function mxa_set_engine(engine) bind(C, name='mxa_set_engine') result(previous)   ! 0 default, 1 i8, 3 f64-strict, 4 i8-exact
   import c_int
   integer(c_int), value, intent(in) :: engine
   integer(c_int) :: previous
  end function

  function mxa_get_engine() bind(C, name='mxa_get_engine') result(engine)
   import c_int
   integer(c_int) :: engine
  end function

  function mxa_last_path() bind(C, name='mxa_last_path') result(path)               ! 0 k_gemm, 1 k_lut, 2 k_gemm_i8, 3 fp64 chains behind the int8 route
   import c_int
   integer(c_int) :: path
  end function

  function mxa_single_orientation(compressed) bind(C, name='mxa_single_orientation') result(single)
   import c_int, c_ptr
   type(c_ptr), value, intent(in) :: compressed
   integer(c_int) :: single
  end function

  function mxa_num_shards(compressed) bind(C, name='mxa_num_shards') result(shards)
   import c_int, c_ptr
   type(c_ptr), value, intent(in) :: compressed
   integer(c_int) :: shards
  end function

  function mxa_allele_freq(plink, snps, indiv, f) bind(C, name='mxa_allele_freq') result(rc)
   import c_int, c_long, c_double, c_ptr
   type(c_ptr), value, intent(in) :: plink
   integer(c_long), value, intent(in) :: snps, indiv
   real(c_double), intent(out) :: f(*)
   integer(c_int) :: rc
  end function

  function mxa_transpose_2bit(in, rows, cols, out) bind(C, name='mxa_transpose_2bit') result(rc)
   import c_int, c_long, c_ptr
   type(c_ptr), value, intent(in) :: in, out
   integer(c_long), value, intent(in) :: rows, cols
   integer(c_int) :: rc
  end function

  ! incremental staging: the object is filled by blocks of SNP rows (objects larger than any buffer the caller could hold)
  function mxa_plink2compressed_begin(snps, indiv, max_n, compressed) bind(C, name='mxa_plink2compressed_begin') result(rc)
   import c_int, c_long, c_ptr
   integer(c_long), value, intent(in) :: snps, indiv
   integer(c_int), value, intent(in) :: max_n
   type(c_ptr), intent(out) :: compressed
   integer(c_int) :: rc
  end function

  function mxa_plink2compressed_rows(compressed, plink_rows, snp_begin, nrows, f_rows) bind(C, name='mxa_plink2compressed_rows') result(rc)
   import c_int, c_long, c_ptr
   type(c_ptr), value, intent(in) :: compressed, plink_rows, f_rows   ! f_rows: c_loc of nrows doubles, or c_null_ptr (counted on the device)
   integer(c_long), value, intent(in) :: snp_begin, nrows              ! zero-based first row
   integer(c_int) :: rc
  end function

  function mxa_plink2compressed_end(compressed) bind(C, name='mxa_plink2compressed_end') result(rc)
   import c_int, c_ptr
   type(c_ptr), value, intent(in) :: compressed
   integer(c_int) :: rc
  end function
 end interface
end module modmiraculix_amd
