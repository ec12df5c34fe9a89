"""Solver twin: potrs_solve_gpu (blocked Cholesky: own diagonal-block kernel + own fp64 MFMA trsm / syrk building blocks, mxa_dense.hip; no vendor library is loaded) and the sparse
triangular solves (own synchronisation-free kernel) through the C ABI, shaped like the reference's
tests/solve/test.jl:67-140 (upper triangular, strictly diagonally dominant sparse matrix; dense exp(-|i-j|/n) matrix; B = randn + 5;
X_sp = M^-1 (M^-T B); residual norms), with scipy / numpy as the CPU check and a far tighter tolerance than the reference's."""
import numpy as np
import pytest
import scipy.linalg
import scipy.sparse
import scipy.sparse.linalg

pytestmark = pytest.mark.gpu
TOL = 1e-10   # relative residual; the reference's test accepts 1e-1 ... 1e-3


@pytest.fixture(scope="module")
def sv():
    import miraculix_amd as m
    m.load_shared_library()
    return m.solve


def simulate_sparse_triangular(n, density, rng, lower=False):
    """tests/solve/test.jl:67-81 restated: diagonal max(randn + 2, 0.1), random strictly upper entries kept diagonally dominant"""
    diag = np.maximum(rng.standard_normal(n) + 2.0, 0.1)
    M = scipy.sparse.lil_matrix((n, n))
    M.setdiag(diag)
    rowsum = np.zeros(n)
    cnt = int(2 * np.ceil(density * n)) * 40
    for i, j in rng.integers(0, n, size=(cnt, 2)):
        if i >= j or M[i, j] != 0:
            continue
        v = rng.random() * max(0.0, 0.9 * diag[i] - rowsum[i])
        if v > 0:
            M[i, j] = v
            rowsum[i] += v
    M = M.tocsr()
    return M.T.tocsr() if lower else M


@pytest.mark.parametrize("n,ncol,density,lower", [(100, 1, 0.05, False), (100, 5, 0.7, True), (3000, 5, 0.05, False), (3000, 20, 0.02, True)])
def test_sparse_triangular_solves(sv, n, ncol, density, lower):
    rng = np.random.default_rng(n + ncol)
    M = simulate_sparse_triangular(n, density, rng, lower)
    coo = M.tocoo()
    # column-major order like Julia's findnz (i.e. NOT sorted by row), one-based
    order = np.lexsort((coo.row, coo.col))
    I, J, V = coo.row[order] + 1, coo.col[order] + 1, coo.data[order]
    B = rng.standard_normal((n, ncol)) + 5.0
    obj = sv.sparse_init(V, I.astype(np.int64), J.astype(np.int64), len(V), n, ncol, lower)
    try:
        Y = sv.sparse_solve(obj, "t", B, n)          # M^T Y = B
        X = sv.sparse_solve(obj, "n", Y, n)          # M X = Y
        with pytest.raises(RuntimeError):            # ncol is fixed at init (solve_cuda.cu:771-777)
            sv.sparse_solve(obj, "n", B[:, :1] if ncol > 1 else np.hstack([B, B]), n)
    finally:
        sv.sparse_free(obj)
    with pytest.raises(RuntimeError, match="uninitialized pointer"):
        sv.sparse_free(obj)                          # tests/solve/test.jl:129
    Md = M.toarray()
    Y_ref = scipy.linalg.solve_triangular(Md, B, trans="T", lower=lower)
    X_ref = scipy.linalg.solve_triangular(Md, Y_ref, lower=lower)
    assert np.abs(Y - Y_ref).max() <= TOL * np.abs(Y_ref).max()
    assert np.abs(X - X_ref).max() <= TOL * np.abs(X_ref).max()
    D = M.T @ (M @ X) - B
    assert np.linalg.norm(D) / np.linalg.norm(B) < TOL


@pytest.mark.parametrize("n,ncol", [(100, 1), (1000, 5), (4000, 20)])
def test_dense_cholesky_solve_and_logdet(sv, n, ncol):
    rng = np.random.default_rng(n)
    idx = np.arange(n, dtype=np.float64)
    M = np.exp(-np.abs(idx[:, None] - idx[None, :]) / n)          # simulate_dense_pd, tests/solve/test.jl:94-98
    M += 1e-3 * np.eye(n)                                        # keep the condition number within what 1e-10 can show
    B = rng.standard_normal((n, ncol)) + 5.0
    X, logdet = sv.dense_solve(M, B, calc_logdet=True)
    X_ref = scipy.linalg.cho_solve(scipy.linalg.cho_factor(M, lower=True), B)
    assert np.linalg.norm(M @ X - B) / np.linalg.norm(B) < TOL
    assert np.abs(X - X_ref).max() <= 1e-8 * np.abs(X_ref).max()
    sign, ld_ref = np.linalg.slogdet(M)
    assert sign > 0 and abs(logdet - ld_ref) <= 1e-10 * abs(ld_ref)
    if n <= 100:   # managed-memory variant (without XNACK every access to it crosses PCIe: small case only)
        X2 = sv.dense_solve(M, B, calc_logdet=False, oversubscribe=True)
        assert np.array_equal(X2, X)


def test_dense_not_positive_definite_and_bad_arguments(sv):
    import ctypes
    import miraculix_amd as m
    L = m.lib.check_library_handle()
    M = np.array([[1.0, 2.0], [2.0, 1.0]])
    with pytest.raises(RuntimeError, match="Cholesky factorization failed at minor 2"):
        sv.dense_solve(M, np.ones((2, 1)))
    X = np.zeros((2, 1))
    assert L.potrs_solve(m.lib.ptr(np.eye(2)), 2, m.lib.ptr(np.ones((2, 1))), 1, m.lib.ptr(X), None, 7) == 1   # oversubscribe not 0/1
    assert L.potrs_solve(m.lib.ptr(np.asfortranarray(2.0 * np.eye(2))), 2, m.lib.ptr(np.ones((2, 1))), 1, m.lib.ptr(X), None, 0) == 0
    assert np.allclose(X, 0.5, rtol=1e-15, atol=0)
    # an entry outside the selected triangle is ignored, as cuSPARSE's fill mode does in the reference (solve_cuda.cu:306-308):
    # upper triangle of [[1, 0], [3, 1]] is the identity
    obj = sv.sparse_init(np.array([1.0, 1.0, 3.0]), np.array([1, 2, 2]), np.array([1, 2, 1]), 3, 2, 1, False)
    try:
        assert np.array_equal(sv.sparse_solve(obj, "n", np.array([[5.0], [7.0]]), 2)[:, 0], np.array([5.0, 7.0]))
    finally:
        sv.sparse_free(obj)
    with pytest.raises(RuntimeError):                 # nothing at all in the selected triangle
        sv.sparse_init(np.array([3.0]), np.array([2]), np.array([1]), 1, 2, 1, False)
    st = ctypes.c_int(5)
    L.dcsrtrsv_solve_gpu(None, b"n", m.lib.ptr(X), 1, m.lib.ptr(X), ctypes.byref(st))
    assert st.value == 1


def test_sparse_solve_single_dependency_chain(sv):
    """Lower bidiagonal matrix: every row depends on the previous one -- one chain through all 25 000 workgroups, far more than are
    resident at a time; the flag-based kernel must drain (measured ~2.6 us per dependent row) and give the exact recurrence."""
    n = 100_000
    I = np.concatenate([np.arange(1, n + 1), np.arange(2, n + 1)]).astype(np.int64)
    J = np.concatenate([np.arange(1, n + 1), np.arange(1, n)]).astype(np.int64)
    V = np.concatenate([np.full(n, 2.0), np.full(n - 1, -1.0)])
    B = np.ones((n, 1))
    obj = sv.sparse_init(V, I, J, len(V), n, 1, True)
    try:
        X = sv.sparse_solve(obj, "n", B, n)[:, 0]          # x_i = (1 + x_{i-1}) / 2
        Xt = sv.sparse_solve(obj, "t", B, n)[:, 0]         # x_i = (1 + x_{i+1}) / 2, from the last row upwards
    finally:
        sv.sparse_free(obj)
    ref = np.empty(n); acc = 0.0
    for i in range(n):
        acc = (1.0 + acc) / 2.0
        ref[i] = acc
    assert np.array_equal(X, ref)
    assert np.array_equal(Xt, ref[::-1])
