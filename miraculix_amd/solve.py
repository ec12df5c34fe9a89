"""Host-side mirror of the reference's Julia module `miraculix.solve` (src/bindings/Julia/solve.jl:45-180): sparse_init,
sparse_solve, sparse_free, dense_solve over the C symbols sparse2gpu, dcsrtrsv_solve_gpu, free_sparse_gpu, potrs_solve_gpu."""
import ctypes

import numpy as np

from . import lib as _lib


def sparse_init(V, I, J, nnz, m, max_ncol, is_lower):
    """solve.jl:45-65.  One-based COO triplets of a triangular m x m matrix (as Julia's findnz returns them)."""
    if (len(V), len(I), len(J)) != (nnz, nnz, nnz):
        raise ValueError("Unexpected length of vectors in COO format.")
    L = _lib.check_library_handle()
    V = np.ascontiguousarray(V, dtype=np.float64)
    I = np.ascontiguousarray(I, dtype=np.int64)
    J = np.ascontiguousarray(J, dtype=np.int64)
    obj_ref = ctypes.c_void_p(None)
    status = ctypes.c_int(0)
    L.sparse2gpu(_lib.ptr(V), _lib.ptr(I), _lib.ptr(J), int(nnz), int(m), int(max_ncol), int(bool(is_lower)), ctypes.byref(obj_ref), ctypes.byref(status))
    if status.value != 0:
        raise RuntimeError("Routine not successful: " + _lib.last_error()[1])
    if not obj_ref.value:
        raise RuntimeError("Encountered uninitialized pointer.")
    return obj_ref


def sparse_solve(obj_ref, transA, B, m):
    """solve.jl:84-105: X = op(A)^-1 B, transA 'n' or 't'."""
    if not obj_ref.value:
        raise RuntimeError("Encountered uninitialized pointer.")
    if B.shape[0] != m:
        raise ValueError(f"B must have {m} rows to be compatible with the sparse matrix.")
    L = _lib.check_library_handle()
    Bf = np.asfortranarray(B, dtype=np.float64)
    ncol = Bf.shape[1]
    X = np.zeros((m, ncol), dtype=np.float64, order="F")
    status = ctypes.c_int(0)
    L.dcsrtrsv_solve_gpu(obj_ref, transA.encode()[0:1], _lib.ptr(Bf), int(ncol), _lib.ptr(X), ctypes.byref(status))
    if status.value != 0:
        raise RuntimeError("Routine not successful: " + _lib.last_error()[1])
    return X


def sparse_free(obj_ref):
    """solve.jl:120-130; obj_ref is NULL afterwards and a second call raises like the Julia binding's check_storage_object."""
    if not obj_ref.value:
        raise RuntimeError("Encountered uninitialized pointer.")
    L = _lib.check_library_handle()
    status = ctypes.c_int(0)
    L.free_sparse_gpu(ctypes.byref(obj_ref), ctypes.byref(status))
    if status.value != 0:
        raise RuntimeError("Routine not successful: " + _lib.last_error()[1])


def dense_solve(M, B, calc_logdet=True, oversubscribe=False):
    """solve.jl:154-180: X = M^-1 B for a symmetric positive definite M by Cholesky; returns X or (X, logdet)."""
    n = M.shape[0]
    if M.shape[1] != n or B.shape[0] != n:
        raise ValueError("Incompatible dimensions of M and B.")
    L = _lib.check_library_handle()
    Mf = np.asfortranarray(M, dtype=np.float64)
    Bf = np.asfortranarray(B, dtype=np.float64)
    ncol = Bf.shape[1]
    X = np.zeros((n, ncol), dtype=np.float64, order="F")
    logdet = np.zeros(1, dtype=np.float64)
    status = ctypes.c_int(0)
    L.potrs_solve_gpu(_lib.ptr(Mf), int(n), _lib.ptr(Bf), int(ncol), _lib.ptr(X), _lib.ptr(logdet) if calc_logdet else None, int(bool(oversubscribe)), ctypes.byref(status))
    if status.value != 0:
        raise RuntimeError("Routine not successful: " + _lib.last_error()[1])
    return (X, float(logdet[0])) if calc_logdet else X
