"""Host-side mirror of the reference's Julia module `miraculix.dgemm_compressed`
(src/bindings/Julia/dgemm_compressed.jl:26-156): same function names, argument meaning and error behaviour, over the
same C ABI.  Matrices are column-major like Julia's: a numpy array is converted with np.asfortranarray, a torch
tensor (host or cuda:k) must be 2-D; it is used in place when its memory is column-major (B.t() contiguous)."""
import ctypes

import numpy as np

from . import lib as _lib


def set_options(use_gpu=True, cores=0, not_center=False, variant=0, verbose=1):
    """dgemm_compressed.jl:42-58.  The Julia binding defaults to use_gpu=false (its CPU engine); this library is the GPU engine
    only, so the default here is True and use_gpu=False raises (the C entry reports the error, leaves the process alive and makes
    later plink2compressed calls return a NULL handle -- see include/miraculix_amd.h)."""
    L = _lib.check_library_handle()
    floatLoop, meanSubstract, ignore_missings, normalize, use_miraculix_freq = 0, 0, 1, 0, 0
    L.setOptions_compressed(int(use_gpu), int(cores), floatLoop, meanSubstract, ignore_missings, int(not_center), normalize, use_miraculix_freq, int(variant), int(verbose))
    if L.mxa_last_error():
        raise RuntimeError("setOptions_compressed failed: " + _lib.last_error()[1])


_ENGINES = {"f64": 0, "i8": 1, "small-n-i8": 2, "f64-strict": 3}


def set_engine(name):
    """Additive (no reference counterpart): 'f64' = default (fp64 matrix cores; for n <= 2 the exact int8 slicing when a per-call
    check proves it exact, else fp64 pair tables), 'i8' = int8 slicing of B for every n, 'small-n-i8' = 'i8' for n <= 4,
    'f64-strict' = fp64 arithmetic for every n (include/miraculix_amd.h, mxa_set_engine).  Returns the previous engine's name."""
    L = _lib.check_library_handle()
    if name not in _ENGINES:
        raise ValueError("engine must be one of " + ", ".join(_ENGINES))
    prev = L.mxa_set_engine(_ENGINES[name])
    return {v: k for k, v in _ENGINES.items()}[prev]


def last_path():
    """kernel family of the most recent product: 'k_gemm' (fp64 MFMA), 'k_lut' (fp64 pair tables) or 'k_gemm_i8' (int8 slicing)"""
    return {0: "k_gemm", 1: "k_lut", 2: "k_gemm_i8"}[_lib.check_library_handle().mxa_last_path()]


def check_dimensions(plink, snps, indiv):
    """miraculix.jl check_dimensions: rows of ceil(indiv/4) bytes, one row per SNP (row-major here)."""
    nbytes = int(np.prod(plink.shape))
    if nbytes != snps * ((indiv + 3) // 4):
        raise ValueError(f"Matrix has wrong dimensions: {tuple(plink.shape)} for snps={snps} indiv={indiv}")


def check_storage_object(obj_ref):
    if obj_ref is None or not obj_ref.value:
        raise RuntimeError("The storage object points to an uninitialized pointer.")


def init_compressed(plink, plink_transposed, snps, indiv, freq, max_ncol):
    """dgemm_compressed.jl:82-93.  plink: snps rows x ceil(indiv/4) bytes, plink_transposed: indiv rows x ceil(snps/4) bytes
    (uint8, C-contiguous numpy arrays or torch uint8 tensors, host or device).  Returns the opaque object reference."""
    obj_ref = ctypes.c_void_p(None)
    check_dimensions(plink, snps, indiv)
    check_dimensions(plink_transposed, indiv, snps)
    L = _lib.check_library_handle()
    if not _lib.is_torch_tensor(freq):
        freq = np.ascontiguousarray(freq, dtype=np.float64)
    L.plink2compressed(_lib.ptr(plink), _lib.ptr(plink_transposed), int(snps), int(indiv), _lib.ptr(freq), int(max_ncol), ctypes.byref(obj_ref))
    if not obj_ref.value:
        raise RuntimeError("plink2compressed failed: " + _lib.last_error()[1])
    return obj_ref


def init_compressed_from_bed(bed_path, max_ncol, snps=0, indiv=0):
    """Stage straight from a PLINK .bed file (C entry mxa_bed2compressed: host C++ reads the file, the transposed copy and the
    allele frequencies are produced on the device).  Returns (obj_ref, freq, snps, indiv)."""
    L = _lib.check_library_handle()
    obj_ref = ctypes.c_void_p(None)
    s_out, i_out = ctypes.c_int(0), ctypes.c_int(0)
    # frequencies: dimension may be unknown before the call -> query with a first pass over .bim if needed
    if snps <= 0:
        base = bed_path[:-4] if bed_path.endswith(".bed") else bed_path
        with open(base + ".bim") as fh:
            snps = sum(1 for _ in fh)
    f = np.zeros(snps, dtype=np.float64)
    rc = L.mxa_bed2compressed(bed_path.encode(), int(snps), int(indiv), int(max_ncol), ctypes.byref(obj_ref), _lib.ptr(f), ctypes.byref(s_out), ctypes.byref(i_out))
    if rc != 0 or not obj_ref.value:
        raise RuntimeError("mxa_bed2compressed failed: " + _lib.last_error()[1])
    return obj_ref, f, s_out.value, i_out.value


def init_compressed_from_bed_range(bed_path, snp_begin, snp_end, max_ncol, snps=0, indiv=0):
    """SNP rows [snp_begin, snp_end) of a .bed file only (C entry mxa_bed2compressed_range): a rank of a one-process-per-GPU job
    stages its own block without anybody holding the full matrices.  Returns (obj_ref, freq of the range)."""
    L = _lib.check_library_handle()
    obj_ref = ctypes.c_void_p(None)
    f = np.zeros(snp_end - snp_begin, dtype=np.float64)
    rc = L.mxa_bed2compressed_range(bed_path.encode(), int(snps), int(indiv), int(snp_begin), int(snp_end), int(max_ncol), ctypes.byref(obj_ref), _lib.ptr(f))
    if rc != 0 or not obj_ref.value:
        raise RuntimeError("mxa_bed2compressed_range failed: " + _lib.last_error()[1])
    return obj_ref, f


def num_shards(obj_ref):
    """number of per-device objects behind a handle: > 1 for one created under MIRACULIX_NUM_GPUS > 1 (include/miraculix_amd.h)"""
    return _lib.check_library_handle().mxa_num_shards(obj_ref)


def init_compressed_shard(plink, plink_transposed, snps_total, indiv, snp_begin, snp_end, freq, max_ncol):
    """SNP-sharded variant (mxa_plink2compressed_shard): this object holds SNPs [snp_begin, snp_end) only."""
    obj_ref = ctypes.c_void_p(None)
    L = _lib.check_library_handle()
    if not _lib.is_torch_tensor(freq):
        freq = np.ascontiguousarray(freq, dtype=np.float64)
    L.mxa_plink2compressed_shard(_lib.ptr(plink), _lib.ptr(plink_transposed), int(snps_total), int(indiv), int(snp_begin), int(snp_end), _lib.ptr(freq), int(max_ncol), ctypes.byref(obj_ref))
    if not obj_ref.value:
        raise RuntimeError("mxa_plink2compressed_shard failed: " + _lib.last_error()[1])
    return obj_ref


def _colmajor(B):
    """returns (object to keep alive, leading dimension) with column-major storage"""
    if _lib.is_torch_tensor(B):
        if B.dim() != 2:
            raise ValueError("B must be 2-D")
        if B.shape[1] == 1 and B.stride(0) == 1:
            return B, B.shape[0]
        if B.stride(0) == 1 and B.stride(1) >= B.shape[0]:
            return B, B.stride(1)
        Bc = B.t().contiguous().t()
        return Bc, Bc.stride(1) if Bc.shape[1] > 1 else Bc.shape[0]
    Bf = np.asfortranarray(B, dtype=np.float64)
    return Bf, Bf.shape[0]


def dgemm_compressed_main(transpose, obj_ref, B, snps, indiv, out=None):
    """dgemm_compressed.jl:114-135.  B: (snps x n) for transpose=False, (indiv x n) for transpose=True.
    Returns C (indiv x n resp. snps x n): numpy Fortran-ordered for numpy input, torch (column-major) on B's device for
    torch input.  `out` optionally supplies the (column-major) result buffer."""
    trans = b"T" if transpose else b"N"
    n_row, n_col = B.shape
    if n_row != (indiv if transpose else snps):
        raise ValueError(f"Matrix B is not compatible with genotype matrix of {snps} SNPs and {indiv} individuals when operation = {trans.decode()} ")
    check_storage_object(obj_ref)
    Bc, ldb = _colmajor(B)
    m = snps if transpose else indiv
    if out is not None:
        C = out
    elif _lib.is_torch_tensor(B):
        import torch
        C = torch.zeros((n_col, m), dtype=torch.float64, device=B.device).t()
    else:
        C = np.zeros((m, n_col), dtype=np.float64, order="F")
    Cc, ldc = _colmajor(C)
    if Cc is not C:
        raise ValueError("out must be column-major")
    L = _lib.check_library_handle()
    L.dgemm_compressed(trans, obj_ref, int(n_col), _lib.ptr(Bc), int(ldb), _lib.ptr(C), int(ldc))
    err = L.mxa_last_error()
    if err:
        raise RuntimeError("dgemm_compressed failed: " + _lib.last_error()[1])
    return C


def sparse_times_plink(transcompressed, plink, plink_transposed, snps, indiv, row_idx, col_idx, values, ldc=None, out=None):
    """Mirror of the Fortran binding c_sparse_times_plink (src/bindings/Fortran/mod5codesapi.f90:84-100, caller
    tests/sparse_plink/test_sparse_plink.f90:99) with transsparse = 'N'.  (row_idx, col_idx, values): ZERO-based CSR of the sparse
    matrix S with nIdx = len(row_idx) - 1 rows.
      transcompressed False ('N'): C (nIdx x indiv) = S (nIdx x snps)  * Z^T, reads `plink`;
      transcompressed True  ('T'): C (nIdx x snps)  = S (nIdx x indiv) * Z,   reads `plink_transposed`.
    Returns the column-major C (Fortran-ordered numpy, or `out`: numpy F-ordered / torch column-major, host or device)."""
    L = _lib.check_library_handle()
    ia = np.ascontiguousarray(row_idx, dtype=np.int32)
    ja = np.ascontiguousarray(col_idx, dtype=np.int32)
    a = np.ascontiguousarray(values, dtype=np.float64)
    nidx = len(ia) - 1
    entries = snps if transcompressed else indiv
    ldc = int(ldc or nidx)
    if out is None:
        C = np.zeros((ldc, entries), dtype=np.float64, order="F")
    else:
        C = out
        Cc, ld = _colmajor(C)
        if Cc is not C or ld != ldc:
            raise ValueError("out must be column-major with leading dimension ldc")
    L.sparse_times_plink(b"N", b"T" if transcompressed else b"N", _lib.ptr(plink), _lib.ptr(plink_transposed), int(snps), int(indiv), int(nidx),
                         _lib.ptr(ia), _lib.ptr(ja), _lib.ptr(a), _lib.ptr(C), ldc)
    if L.mxa_last_error():
        raise RuntimeError("sparse_times_plink failed: " + _lib.last_error()[1])
    return C[:nidx, :] if out is None else C


def gram_matvec(obj_ref, V, snps, indiv, out=None):
    """Additive: out (indiv x n) = Zc (Zc^T V) in one call (the 'T' + 'N' pair of a CG / GBLUP step,
    examples/iterative_solver/grm_solve_cg.jl:74-84) with the snps x n intermediate kept on the device."""
    check_storage_object(obj_ref)
    if V.shape[0] != indiv:
        raise ValueError(f"Matrix V must have {indiv} rows")
    n = V.shape[1]
    Vc, ldv = _colmajor(V)
    if out is not None:
        C = out
    elif _lib.is_torch_tensor(V):
        import torch
        C = torch.zeros((n, indiv), dtype=torch.float64, device=V.device).t()
    else:
        C = np.zeros((indiv, n), dtype=np.float64, order="F")
    Cc, ldo = _colmajor(C)
    if Cc is not C:
        raise ValueError("out must be column-major")
    L = _lib.check_library_handle()
    if L.mxa_gram_matvec(obj_ref, int(n), _lib.ptr(Vc), int(ldv), _lib.ptr(C), int(ldo)):
        raise RuntimeError("mxa_gram_matvec failed: " + _lib.last_error()[1])
    return C


def free_compressed(obj_ref):
    """dgemm_compressed.jl:149-156"""
    check_storage_object(obj_ref)
    L = _lib.check_library_handle()
    L.free_compressed(ctypes.byref(obj_ref))
