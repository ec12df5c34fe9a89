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


_ENGINES = {"f64": 0, "i8": 1, "f64-strict": 3, "i8-exact": 4}


def set_engine(name):
    """Additive (no reference counterpart): 'f64' = default (fp64 matrix cores; for n <= 6 and for odd columns the exact int8 slicing when a
    device-side check proves it exact, else fp64), 'i8' = int8 slicing of B for every n (7 digits, no exactness check), 'f64-strict' = fp64 arithmetic
    for every n, 'i8-exact' = the int8 slicing for every n with the digit count chosen per call so that B is represented without error (else the fp64
    path; the host reads three integers per call).  Table: include/miraculix_amd.h, mxa_set_engine.  Returns the previous engine's name."""
    L = _lib.check_library_handle()
    if name not in _ENGINES:
        raise ValueError("engine must be one of " + ", ".join(_ENGINES))
    prev = L.mxa_set_engine(_ENGINES[name])
    return {v: k for k, v in _ENGINES.items()}[prev]


def last_path():
    """kernel family of the most recent product: 'k_gemm' (fp64 MFMA), 'k_lut' (fp64 pair tables: engine f64-strict, K < 128), 'k_gemm_i8' (exact int8
    slicing) or 'k_small_n_fp64' (the fp64 path behind the exactness guard of the int8 route -- gated fp64 kernels, plain chains at n = 1 --: the device-side verdict declined it)"""
    return {0: "k_gemm", 1: "k_lut", 2: "k_gemm_i8", 3: "k_small_n_fp64"}[_lib.check_library_handle().mxa_last_path()]


def single_orientation(obj):
    """1: the object keeps ONE packed copy (SNP-major, the default), 0: both copies (multi-device object: of its shards); -1: not an object"""
    return int(_lib.check_library_handle().mxa_single_orientation(obj))


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
    (uint8, C-contiguous numpy arrays or torch uint8 tensors, host or device).  plink_transposed may be None (or plink itself, the call
    shape of the reference's CPU benchmark, utils/benchmark/benchmark.f90:185): the library then builds the individual-major copy on the
    device.  Returns the opaque object reference."""
    obj_ref = ctypes.c_void_p(None)
    check_dimensions(plink, snps, indiv)
    if plink_transposed is not None and plink_transposed is not plink:
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


def init_compressed_begin(snps, indiv, max_ncol):
    """Additive (mxa_plink2compressed_begin): allocate a single-orientation object to be filled by SNP-row blocks (append_rows) and sealed
    (init_compressed_end) -- staging without the whole PLINK matrix behind one pointer (include/miraculix_amd.h)."""
    obj_ref = ctypes.c_void_p(None)
    L = _lib.check_library_handle()
    if L.mxa_plink2compressed_begin(int(snps), int(indiv), int(max_ncol), ctypes.byref(obj_ref)) or not obj_ref.value:
        raise RuntimeError("mxa_plink2compressed_begin failed: " + _lib.last_error()[1])
    return obj_ref


def append_rows(obj_ref, plink_rows, snp_begin, freq=None):
    """SNP rows [snp_begin, snp_begin + plink_rows.shape[0]) of the object: compact PLINK rows (uint8, C-contiguous numpy array or torch tensor, host or
    device); freq: their allele frequencies, or None to have them counted on the device."""
    check_storage_object(obj_ref)
    L = _lib.check_library_handle()
    if freq is not None and not _lib.is_torch_tensor(freq):
        freq = np.ascontiguousarray(freq, dtype=np.float64)
    if L.mxa_plink2compressed_rows(obj_ref, _lib.ptr(plink_rows), int(snp_begin), int(plink_rows.shape[0]), _lib.ptr(freq)):
        raise RuntimeError("mxa_plink2compressed_rows failed: " + _lib.last_error()[1])


def init_compressed_end(obj_ref, snps):
    """seal an incrementally staged object; returns its allele frequencies (numpy)"""
    check_storage_object(obj_ref)
    L = _lib.check_library_handle()
    if L.mxa_plink2compressed_end(obj_ref):
        raise RuntimeError("mxa_plink2compressed_end failed: " + _lib.last_error()[1])
    f = np.zeros(snps, dtype=np.float64)
    L.get_compressed_freq(obj_ref, _lib.ptr(f))
    return f


def num_shards(obj_ref):
    """number of per-device objects behind a handle: > 1 for one created under MIRACULIX_NUM_GPUS > 1 (include/miraculix_amd.h)"""
    return _lib.check_library_handle().mxa_num_shards(obj_ref)


def shard_bounds(obj_ref, snps):
    """SNP blocks [(begin, end), ...] of the per-device objects behind a handle (mxa_shard_bounds)"""
    L = _lib.check_library_handle()
    G = L.mxa_num_shards(obj_ref)
    out = []
    for g in range(G):
        b, e = ctypes.c_long(0), ctypes.c_long(0)
        L.mxa_shard_bounds(int(snps), G, g, ctypes.byref(b), ctypes.byref(e))
        out.append((b.value, e.value))
    return out


def dgemm_compressed_multi(transpose, obj_ref, B_per_shard, C_per_shard, sync=True):
    """Additive (mxa_dgemm_compressed_multi): one product on a multi-device object with the operands handed over per shard -- lists of
    column-major torch tensors (entries may be None where the header allows it).  'N': B_per_shard[g] = rows of shard g's SNP block,
    C_per_shard[0] = the reduced indiv x n result; 'T': B_per_shard[g] = B as shard g reads it, C_per_shard[g] = shard g's row block.
    All B slices share one leading dimension, all C slices too.  sync=False returns when the work is enqueued (multi_synchronize waits)."""
    check_storage_object(obj_ref)
    L = _lib.check_library_handle()
    G = L.mxa_num_shards(obj_ref)
    if len(B_per_shard) != G or len(C_per_shard) != G:
        raise ValueError(f"need {G} entries per operand list")
    n = next(b for b in B_per_shard if b is not None).shape[1]
    lds = []
    for ops in (B_per_shard, C_per_shard):
        ld = None
        for t in ops:
            if t is None:
                continue
            tc, l = _colmajor(t)
            if tc is not t:
                raise ValueError("per-shard operands must be column-major")
            if ld is not None and l != ld and t.shape[1] > 1:
                raise ValueError("per-shard operands must share one leading dimension")
            ld = l if ld is None or t.shape[1] > 1 else max(ld, l)
        lds.append(ld)
    Bp = (ctypes.c_void_p * G)(*[None if t is None else t.data_ptr() for t in B_per_shard])
    Cp = (ctypes.c_void_p * G)(*[None if t is None else t.data_ptr() for t in C_per_shard])
    if L.mxa_dgemm_compressed_multi(b"T" if transpose else b"N", obj_ref, int(n), Bp, int(lds[0]), Cp, int(lds[1]), int(bool(sync))):
        raise RuntimeError("mxa_dgemm_compressed_multi failed: " + _lib.last_error()[1])


def multi_synchronize(obj_ref):
    if _lib.check_library_handle().mxa_multi_synchronize(obj_ref):
        raise RuntimeError("mxa_multi_synchronize failed: " + _lib.last_error()[1])


def multi_set_reduction(obj_ref, kind):
    """'p2p' (peer-to-peer pushes + one fixed-order addition kernel, the default) or 'rccl' (ncclReduce).  Returns False when RCCL is not
    applicable (several shards share a device); raises on errors."""
    rc = _lib.check_library_handle().mxa_multi_set_reduction(obj_ref, {"p2p": 0, "rccl": 1}[kind])
    if rc == 1:
        raise RuntimeError("mxa_multi_set_reduction failed: " + _lib.last_error()[1])
    return rc == 0


def multi_info(obj_ref, reset=False):
    """dict with the object's layout and what it has done since the last reset: per shard the device, the SNP block, the peer-access
    verdicts, the dominant-kernel launches / ms, operand-distribution, result-gather and partial-push copies / ms; the reduction kind,
    count and ms; the RCCL-vs-peer-to-peer cross-check (mxa_multi_get_info / mxa_multi_shard_info)."""
    L = _lib.check_library_handle()
    mi = _lib.MultiInfo()
    L.mxa_multi_synchronize(obj_ref)          # everything issued has finished and its events have been read
    if L.mxa_multi_get_info(obj_ref, ctypes.byref(mi)):
        raise RuntimeError("mxa_multi_get_info: not a multi-device object")
    out = {k: getattr(mi, k) for k, _ in _lib.MultiInfo._fields_}
    out["reduction"] = "rccl" if mi.reduction else "p2p-fixed-order"
    out["per_shard"] = []
    for g in range(mi.shards):
        si = _lib.ShardInfo()
        L.mxa_multi_shard_info(obj_ref, g, ctypes.byref(si))
        out["per_shard"].append({k: getattr(si, k) for k, _ in _lib.ShardInfo._fields_})
    if reset:
        L.mxa_multi_reset_profile(obj_ref)
    return out


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


def dgemm_plink(transpose, plink, plink_transposed, snps, indiv, f, B):
    """The reference's dgemm_plink (src/miraculix/5codesAPI.c:112-130): one product straight from the PLINK matrices, no object kept.  transpose False ('N'):
    C (indiv x n) = Zc B, B snps x n; True ('T'): C (snps x n) = Zc^T B, B indiv x n.  f None: uncentred, else centred with f.  Only the matrix the reference
    would read has to be given ('N': plink_transposed, 'T': plink).  Parity unpinned: the reference aborts unconditionally there (include/miraculix_amd.h)."""
    n_row, n_col = B.shape
    if n_row != (indiv if transpose else snps):
        raise ValueError(f"Matrix B is not compatible with genotype matrix of {snps} SNPs and {indiv} individuals")
    Bc, ldb = _colmajor(B)
    m = snps if transpose else indiv
    if _lib.is_torch_tensor(B):
        import torch
        C = torch.zeros((n_col, m), dtype=torch.float64, device=B.device).t()
    else:
        C = np.zeros((m, n_col), dtype=np.float64, order="F")
    _, ldc = _colmajor(C)
    if f is not None and not _lib.is_torch_tensor(f):
        f = np.ascontiguousarray(f, dtype=np.float64)
    L = _lib.check_library_handle()
    L.dgemm_plink(b"T" if transpose else b"N", _lib.ptr(plink), _lib.ptr(plink_transposed), int(snps), int(indiv), _lib.ptr(f), int(n_col), _lib.ptr(Bc), int(ldb),
                  _lib.ptr(C), int(ldc))
    if L.mxa_last_error():
        raise RuntimeError("dgemm_plink failed: " + _lib.last_error()[1])
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


def gram_matvec(obj_ref, V, snps, indiv, out=None, sync=True):
    """Additive: out (indiv x n) = Zc (Zc^T V) in one call (the 'T' + 'N' pair of a CG / GBLUP step,
    examples/iterative_solver/grm_solve_cg.jl:74-84) with the snps x n intermediate kept on the device.
    sync=False (torch tensors on the object's device, single-device object): returns when the products are enqueued on the object's blocking
    stream; later work on the device's default stream -- the torch ops of the loop -- is ordered behind them (mxa_gram_matvec_device)."""
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
    if not sync and _lib.is_torch_tensor(V) and V.is_cuda and L.mxa_num_shards(obj_ref) == 1:
        if L.mxa_gram_matvec_device(obj_ref, int(n), _lib.ptr(Vc), int(ldv), _lib.ptr(C), int(ldo), 0):
            raise RuntimeError("mxa_gram_matvec_device failed: " + _lib.last_error()[1])
        return C
    if L.mxa_gram_matvec(obj_ref, int(n), _lib.ptr(Vc), int(ldv), _lib.ptr(C), int(ldo)):
        raise RuntimeError("mxa_gram_matvec failed: " + _lib.last_error()[1])
    return C


def free_compressed(obj_ref):
    """dgemm_compressed.jl:149-156"""
    check_storage_object(obj_ref)
    L = _lib.check_library_handle()
    L.free_compressed(ctypes.byref(obj_ref))
