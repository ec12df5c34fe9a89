"""Shared test helpers: synthetic PLINK data, the oracle (ctypes), the reference driver (build container only).

Everything under oracle/ is TEST INFRASTRUCTURE: loaded here, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg only.
"""
import ctypes
import os
import struct
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.environ.get("ORACLE_SO", os.path.join(ORACLE_DIR, "liboracle.so"))   # ORACLE_SO: the sanitizer build (tools/run_sanitizers.sh)
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libmiraculix_ref.so")
REF_DRIVER = os.path.join(ORACLE_DIR, "_ref", "ref_driver")

c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_i32p = ctypes.POINTER(ctypes.c_int32)


def synth_genotypes(snps, indiv, seed=42, missing_frac=0.0):
    """Synthetic genotypes per SURVEY.md 8(d): p_s ~ U(0.1,0.6), g ~ Binomial(2,p_s); returns Z (indiv x snps, int8)
    and a missing mask (or None)."""
    rng = np.random.default_rng(seed)
    p = rng.uniform(0.1, 0.6, size=snps)
    Z = rng.binomial(2, p[None, :], size=(indiv, snps)).astype(np.int8)
    miss = None
    if missing_frac > 0:
        miss = rng.random((indiv, snps)) < missing_frac
    return Z, miss


def pack_plink(Z, miss=None):
    """Z (rows x cols int, values 0/1/2) -> PLINK 2-bit rows: 0->00, 1->10, 2->11, missing->01; 4 per byte, low bits first,
    rows padded with zero bits (read_plink.jl:152, SURVEY.md terminology)."""
    rows, cols = Z.shape
    code = np.where(Z == 0, 0, Z + 1).astype(np.uint8)
    if miss is not None:
        code = np.where(miss, 1, code).astype(np.uint8)
    pad = (-cols) % 4
    if pad:
        code = np.concatenate([code, np.zeros((rows, pad), np.uint8)], axis=1)
    c4 = code.reshape(rows, -1, 4)
    return (c4[:, :, 0] | (c4[:, :, 1] << 2) | (c4[:, :, 2] << 4) | (c4[:, :, 3] << 6)).astype(np.uint8)


def make_problem(snps, indiv, n, seed=42, missing_frac=0.0, ldb_pad=0):
    """Returns dict with plink (snps x ceil(indiv/4)), plink_t (indiv x ceil(snps/4)), f, Z (with missing as 0)."""
    Z, miss = synth_genotypes(snps, indiv, seed, missing_frac)
    plink = pack_plink(Z.T.copy(), None if miss is None else miss.T.copy())
    plink_t = pack_plink(Z, miss)
    Zeff = Z.copy()
    if miss is not None:
        Zeff[miss] = 0
    f = Zeff.astype(np.float64).mean(axis=0) / 2.0
    return dict(snps=snps, indiv=indiv, plink=np.ascontiguousarray(plink), plink_t=np.ascontiguousarray(plink_t), f=f, Z=Zeff)


def make_B(k, n, seed=43, ldb=None):
    rng = np.random.default_rng(seed)
    ldb = ldb or k
    B = np.zeros((n, ldb))  # row j = column j of the column-major matrix
    B[:, :k] = rng.standard_normal((n, k))
    if ldb > k:
        B[:, k:] = 1e300  # poison the ld padding: must never be read as data
    return B


def elementwise_bound(o, trans, prob, B, centered, factor=4.0):
    """The hard element-wise error bound of SURVEY.md 8(d) for an fp64 chain of length K: factor * K * 2^-53 * sum_k |z_ik| |b_kj|, per output (n x m, like
    Oracle.dgemm_dense).  Centred: the product computes sum z b and the rank-1 term 2 sum f b separately, so the magnitudes add: 'N' + 2 sum_k f_k |b_kj| (per
    column), 'T' + 2 f_s sum_i |b_ij|.  The norm-wise tolerance (1e-11 of the largest entry) leaves output rows far below max|C| unprotected; this one does not."""
    snps, indiv = prob["snps"], prob["indiv"]
    k, m = (indiv, snps) if trans else (snps, indiv)
    Ba = np.ascontiguousarray(np.abs(B))
    mag = o.dgemm_dense(trans, prob, Ba, 0)[:, :m]                                   # sum |z| |b|  (z >= 0)
    if centered:
        f = np.asarray(prob["f"], dtype=np.float64)
        if trans:
            mag = mag + 2.0 * Ba[:, :k].sum(axis=1)[:, None] * f[None, :m]
        else:
            mag = mag + 2.0 * (Ba[:, :k] * f[None, :k]).sum(axis=1)[:, None]
    return factor * k * 2.0 ** -53 * mag


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
        L = ctypes.CDLL(ORACLE_SO)
        self.L = L
        L.oracle_dgemm_dense.argtypes = [ctypes.c_int, c_u8p, ctypes.c_long, ctypes.c_long, c_f64p, ctypes.c_int, ctypes.c_long, c_f64p, ctypes.c_long, c_f64p, ctypes.c_long]
        L.oracle_dgemm_gpuorder.argtypes = [ctypes.c_int, c_u8p, c_u8p, ctypes.c_long, ctypes.c_long, c_f64p, ctypes.c_int, ctypes.c_long, c_f64p, ctypes.c_long, c_f64p, ctypes.c_long]
        L.oracle5_create.restype = ctypes.c_void_p
        L.oracle5_create.argtypes = [c_u8p, ctypes.c_long, ctypes.c_long, c_f64p, ctypes.c_int]
        L.oracle5_dgemm.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_long, c_f64p, ctypes.c_long, c_f64p, ctypes.c_long]
        L.oracle5_free.argtypes = [ctypes.c_void_p]
        L.oracle_crossprod_i32.argtypes = [c_u8p, ctypes.c_long, ctypes.c_long, ctypes.c_int, c_i32p]
        L.oracle_crossprod_f64.argtypes = [c_u8p, ctypes.c_long, ctypes.c_long, ctypes.c_int, c_f64p]
        L.oracle_transpose_2bit.argtypes = [c_u8p, ctypes.c_long, ctypes.c_long, c_u8p]
        L.oracle_allele_freq.argtypes = [c_u8p, ctypes.c_long, ctypes.c_long, c_f64p]
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_sparse_times_plink.argtypes = [c_u8p, ctypes.c_long, ctypes.c_long, ctypes.c_int, c_i32p, c_i32p, c_f64p, c_f64p, ctypes.c_long]

    @staticmethod
    def _u8(a):
        return a.ctypes.data_as(c_u8p)

    @staticmethod
    def _f64(a):
        return a.ctypes.data_as(c_f64p)

    def dgemm_dense(self, trans, prob, B, centered, ldc=None):
        snps, indiv = prob["snps"], prob["indiv"]
        n, ldb = B.shape
        m = snps if trans else indiv
        ldc = ldc or m
        C = np.full((n, ldc), -777.0)
        self.L.oracle_dgemm_dense(int(trans), self._u8(prob["plink"]), snps, indiv, self._f64(prob["f"]), int(centered), n, self._f64(B), ldb, self._f64(C), ldc)
        return C

    def dgemm_gpuorder(self, trans, prob, B, centered, ldc=None):
        snps, indiv = prob["snps"], prob["indiv"]
        n, ldb = B.shape
        m = snps if trans else indiv
        ldc = ldc or m
        C = np.full((n, ldc), -777.0)
        self.L.oracle_dgemm_gpuorder(int(trans), self._u8(prob["plink"]), self._u8(prob["plink_t"]), snps, indiv, self._f64(prob["f"]), int(centered), n, self._f64(B), ldb, self._f64(C), ldc)
        return C

    def five_create(self, prob, cores=0):
        cores = cores or self.L.oracle_num_threads()
        return self.L.oracle5_create(self._u8(prob["plink"]), prob["snps"], prob["indiv"], self._f64(prob["f"]), cores)

    def five_dgemm(self, h, trans, prob, B, centered, ldc=None):
        n, ldb = B.shape
        m = prob["snps"] if trans else prob["indiv"]
        ldc = ldc or m
        C = np.full((n, ldc), -777.0)
        self.L.oracle5_dgemm(h, int(trans), int(centered), n, self._f64(B), ldb, self._f64(C), ldc)
        return C

    def five_free(self, h):
        self.L.oracle5_free(h)

    def crossprod_i32(self, X, k, is_plink):
        rows = X.shape[0]
        out = np.zeros((rows, rows), np.int32)
        self.L.oracle_crossprod_i32(self._u8(X), k, rows, int(is_plink), out.ctypes.data_as(c_i32p))
        return out

    def transpose_2bit(self, X, rows, cols):
        out = np.zeros((cols, (rows + 3) // 4), np.uint8)
        self.L.oracle_transpose_2bit(self._u8(X), rows, cols, self._u8(out))
        return out

    def sparse_times_plink(self, P, rows, entries, ia, ja, a, ldc=None):
        """P: rows x ceil(entries/4) PLINK bytes; (ia, ja, a): zero-based CSR with len(ia)-1 sparse rows.  Returns the column-major
        C as numpy (entries, ldc)."""
        nidx = len(ia) - 1
        ldc = ldc or nidx
        C = np.full((entries, ldc), -777.0)
        ia = np.ascontiguousarray(ia, np.int32); ja = np.ascontiguousarray(ja, np.int32); a = np.ascontiguousarray(a, np.float64)
        self.L.oracle_sparse_times_plink(self._u8(P), rows, entries, nidx, ia.ctypes.data_as(c_i32p), ja.ctypes.data_as(c_i32p), self._f64(a), self._f64(C), ldc)
        return C

    def allele_freq(self, plink, snps, indiv):
        f = np.zeros(snps)
        self.L.oracle_allele_freq(self._u8(plink), snps, indiv, self._f64(f))
        return f


def random_csr(nidx, ncols, max_nnz, seed, empty_rows=True):
    """Zero-based CSR (ia, ja, a) with 0..max_nnz sorted distinct columns per row and N(0,1) values."""
    rng = np.random.default_rng(seed)
    ia, ja, a = [0], [], []
    for j in range(nidx):
        k = int(rng.integers(0 if empty_rows else 1, min(max_nnz, ncols) + 1))
        cols = np.sort(rng.choice(ncols, size=k, replace=False))
        ja += cols.tolist()
        a += rng.standard_normal(k).tolist()
        ia.append(len(ja))
    return np.array(ia, np.int32), np.array(ja, np.int32), np.array(a, np.float64)


def have_reference():
    return os.path.exists(REF_SO) and os.path.exists(REF_DRIVER)


def run_reference(prob, trans, B, centered, ldc=None, variant=256, cores=8, reps=1):
    """Run the reference's own CPU 5codes library (oracle/_ref, build container or wherever it was built)."""
    snps, indiv = prob["snps"], prob["indiv"]
    n, ldb = B.shape
    m = snps if trans else indiv
    ldc = ldc or m
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as fh:
            fh.write(struct.pack("9i", snps, indiv, n, ldb, ldc, int(trans), int(not centered), variant, cores))
            fh.write(prob["plink"].tobytes())
            fh.write(prob["f"].astype(np.float64).tobytes())
            fh.write(np.ascontiguousarray(B).tobytes())
        env = dict(os.environ, OMP_NUM_THREADS=str(cores))
        subprocess.run([REF_DRIVER, REF_SO, fin, fout, str(reps)], check=True, env=env, stdout=subprocess.DEVNULL, timeout=600)
        raw = np.fromfile(fout, dtype=np.float64)
    C = raw[: n * ldc].reshape(n, ldc)
    return C, float(raw[n * ldc])
