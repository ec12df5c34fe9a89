"""miraculix_amd -- MI355X-native compressed-genotype GEMM engine behind the miraculix C ABI.

The product is the shared library miraculix_amd/lib/libmiraculix_amd.so (C ABI: include/miraculix_amd.h, sources
miraculix_amd/csrc).  These Python modules are the host-side mirror of the reference's Julia binding modules
(src/bindings/Julia/miraculix.jl:114-118: dgemm_compressed, crossproduct, solve, read_plink, compressed_operations), used by
the tests and bench.py exactly as the reference's tests use the Julia modules.  No CPU fallback exists.
"""
from . import lib  # noqa: F401
from . import dgemm_compressed  # noqa: F401
from . import crossproduct  # noqa: F401
from . import compressed_operations  # noqa: F401
from . import read_plink  # noqa: F401
from . import solve  # noqa: F401
from .lib import load_shared_library, set_library_path, check_library_handle  # noqa: F401
