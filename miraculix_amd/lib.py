"""Loading of the C-ABI shared library (mirrors src/bindings/Julia/miraculix.jl:60-110: set_library_path /
load_shared_library / check_library_handle).  There is no fallback: a missing library is an error."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT = os.path.join(_HERE, "lib", "libmiraculix_amd.so")
_LIBRARY_PATH = [os.environ.get("MIRACULIX_AMD_LIBRARY", _DEFAULT)]
_LIBRARY_HANDLE = [None]

c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_f64p = ctypes.POINTER(ctypes.c_double)


class MultiInfo(ctypes.Structure):   # include/miraculix_amd.h: mxa_multi_info
    _fields_ = [("shards", ctypes.c_int), ("devices", ctypes.c_int), ("root_device", ctypes.c_int), ("reduction", ctypes.c_int),
                ("reductions", ctypes.c_int), ("reduce_ms", ctypes.c_double), ("rccl_checked", ctypes.c_int), ("rccl_vs_p2p_max_rel_diff", ctypes.c_double)]


class ShardInfo(ctypes.Structure):   # include/miraculix_amd.h: mxa_shard_info
    _fields_ = [("device", ctypes.c_int), ("snp_begin", ctypes.c_long), ("snp_end", ctypes.c_long), ("peer_to_root", ctypes.c_int), ("peer_from_root", ctypes.c_int),
                ("kernel_launches", ctypes.c_int), ("kernel_ms", ctypes.c_double), ("in_copies", ctypes.c_int), ("in_ms", ctypes.c_double),
                ("out_copies", ctypes.c_int), ("out_ms", ctypes.c_double), ("pushes", ctypes.c_int), ("push_ms", ctypes.c_double)]


def set_library_path(path):
    _LIBRARY_PATH[0] = path
    _LIBRARY_HANDLE[0] = None


def load_shared_library():
    path = _LIBRARY_PATH[0]
    if not os.path.exists(path):
        raise RuntimeError(
            f"miraculix_amd: shared library {path} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C miraculix_amd/csrc`. There is no CPU fallback."
        )
    # RTLD_GLOBAL on purpose, for this PYTHON mirror only: PyTorch bundles its own libamdhip64 and the library links the system one.  Whichever HIP runtime
    # enters the global scope first serves both, so device pointers of torch tensors are valid in the library whatever the import order (with RTLD_LOCAL and
    # the library loaded first, torch's runtime fails to initialise: "No HIP GPUs are available").  What becomes global is the C ABI only -- the library
    # exports nothing else (csrc/exports.map).  A Julia / R / Fortran host dlopen()s it the ordinary way.
    L = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    L.setOptions_compressed.argtypes = [ctypes.c_int] * 10
    L.setOptions_compressed.restype = None
    L.plink2compressed.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.plink2compressed.restype = None
    L.mxa_plink2compressed_shard.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.mxa_plink2compressed_shard.restype = None
    L.dgemm_compressed.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.dgemm_compressed.restype = None
    L.mxa_dgemm_compressed_device.argtypes = [ctypes.c_char, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int]
    L.mxa_dgemm_compressed_device.restype = ctypes.c_int
    L.free_compressed.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    L.free_compressed.restype = None
    L.get_compressed_freq.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.get_compressed_freq.restype = None
    L.snp_multiply_gpu.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_bool]
    L.snp_multiply_gpu.restype = ctypes.c_int
    L.mxa_bed2compressed.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.mxa_bed2compressed.restype = ctypes.c_int
    L.mxa_bed2compressed_range.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]
    L.mxa_bed2compressed_range.restype = ctypes.c_int
    L.mxa_plink2compressed_begin.argtypes = [ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.mxa_plink2compressed_begin.restype = ctypes.c_int
    L.mxa_plink2compressed_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_void_p]
    L.mxa_plink2compressed_rows.restype = ctypes.c_int
    L.mxa_plink2compressed_end.argtypes = [ctypes.c_void_p]
    L.mxa_plink2compressed_end.restype = ctypes.c_int
    L.mxa_num_shards.argtypes = [ctypes.c_void_p]
    L.mxa_num_shards.restype = ctypes.c_int
    L.mxa_shard_bounds.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]
    L.mxa_shard_bounds.restype = ctypes.c_int
    L.mxa_dgemm_compressed_multi.argtypes = [ctypes.c_char, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_long, ctypes.POINTER(ctypes.c_void_p), ctypes.c_long, ctypes.c_int]
    L.mxa_dgemm_compressed_multi.restype = ctypes.c_int
    L.mxa_multi_synchronize.argtypes = [ctypes.c_void_p]
    L.mxa_multi_synchronize.restype = ctypes.c_int
    L.mxa_multi_set_reduction.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.mxa_multi_set_reduction.restype = ctypes.c_int
    L.mxa_multi_get_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(MultiInfo)]
    L.mxa_multi_get_info.restype = ctypes.c_int
    L.mxa_multi_shard_info.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ShardInfo)]
    L.mxa_multi_shard_info.restype = ctypes.c_int
    L.mxa_multi_reset_profile.argtypes = [ctypes.c_void_p]
    L.mxa_multi_reset_profile.restype = ctypes.c_int
    L.mxa_grm.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.mxa_grm.restype = ctypes.c_int
    L.mxa_ld.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.mxa_ld.restype = ctypes.c_int
    L.mxa_last_error.restype = ctypes.c_int
    L.mxa_last_error_string.restype = ctypes.c_char_p
    L.mxa_device_count.restype = ctypes.c_int
    L.mxa_transpose_2bit.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_void_p]
    L.mxa_transpose_2bit.restype = ctypes.c_int
    L.mxa_allele_freq.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_void_p]
    L.mxa_allele_freq.restype = ctypes.c_int
    L.sparse_times_plink.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.sparse_times_plink.restype = None
    L.dgemm_plink.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                              ctypes.c_void_p, ctypes.c_int]
    L.dgemm_plink.restype = None
    L.mxa_gram_matvec.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long]
    L.mxa_gram_matvec.restype = ctypes.c_int
    L.mxa_gram_matvec_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int]
    L.mxa_gram_matvec_device.restype = ctypes.c_int
    L.mxa_snp_multiply_panel.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int]
    L.mxa_snp_multiply_panel.restype = ctypes.c_int
    L.potrs_solve_gpu.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.potrs_solve_gpu.restype = None
    L.potrs_solve.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.potrs_solve.restype = ctypes.c_int
    L.sparse2gpu.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]
    L.sparse2gpu.restype = None
    L.dcsrtrsv_solve_gpu.argtypes = [ctypes.c_void_p, ctypes.c_char, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    L.dcsrtrsv_solve_gpu.restype = None
    L.free_sparse_gpu.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]
    L.free_sparse_gpu.restype = None
    L.mxa_set_engine.argtypes = [ctypes.c_int]
    L.mxa_set_engine.restype = ctypes.c_int
    L.mxa_get_engine.restype = ctypes.c_int
    L.mxa_last_path.restype = ctypes.c_int
    L.mxa_last_range_fallback.argtypes = [ctypes.c_void_p]
    L.mxa_last_range_fallback.restype = ctypes.c_int
    L.mxa_profile_reset.restype = None
    L.mxa_profile_get.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]
    L.mxa_profile_get.restype = None
    L.mxa_last_geometry.argtypes = [ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.mxa_last_geometry.restype = None
    L.mxa_single_orientation.argtypes = [ctypes.c_void_p]
    L.mxa_single_orientation.restype = ctypes.c_int
    L.mxa_partial_capacity.argtypes = [ctypes.c_void_p]
    L.mxa_partial_capacity.restype = ctypes.c_long
    L.mxa_plan_partial_doubles.argtypes = [ctypes.c_long, ctypes.c_long, ctypes.c_int]
    L.mxa_plan_partial_doubles.restype = ctypes.c_long
    _LIBRARY_HANDLE[0] = L
    return L


def check_library_handle():
    """miraculix.jl:29-35 -- error if the library has not been loaded; here it is loaded on first use."""
    if _LIBRARY_HANDLE[0] is None:
        load_shared_library()
    return _LIBRARY_HANDLE[0]


def last_error():
    L = check_library_handle()
    return L.mxa_last_error(), (L.mxa_last_error_string() or b"").decode()


def is_torch_tensor(x):
    return type(x).__module__.startswith("torch")


def ptr(x):
    """address of a numpy array or a torch tensor (host or device); None -> NULL"""
    if x is None:
        return None
    if is_torch_tensor(x):
        return ctypes.c_void_p(x.data_ptr())
    return ctypes.c_void_p(x.ctypes.data)
