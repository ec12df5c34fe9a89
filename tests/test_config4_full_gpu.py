"""BASELINE config 4 as ONE product at its stated size on ONE MI355X: 5 000 000 SNPs x 200 000 individuals, ncol = 128,
allele-frequency centred -- 250 GB of packed genotypes in a single-orientation object staged incrementally (mxa_plink2compressed_begin / _rows /
_end) from SNP blocks generated on the device, 'N' and 'T' once each under the checker: sampled rows against the centred long-double oracle on the
extracted packed rows (<= 1e-11), the centred adjoint identity, bitwise repeatability.  The reference cannot hold this problem on one device
(src/cuda/dgemm_compressed_cuda.cu:93-100).  The test body is bench.py's leg `config4_full_one_copy`.  A file of its own, run early in the
alphabetical order: it needs the whole device."""
import sys

import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-11


def test_config4_full_5M_x_200k_x_128_one_product_one_copy():
    import torch
    import bench
    import miraculix_amd as mx
    L = mx.load_shared_library()
    torch.cuda.empty_cache()
    dev = torch.device("cuda", 0)
    leg = bench.config4_full_one_copy_leg(torch, mx, L, dev, 5_000_000, 200_000, 128, log=lambda m: print(m, file=sys.stderr, flush=True))
    print(leg, file=sys.stderr)
    assert bench.leg_checks_ok(leg), leg
    ck = leg["check"]
    assert ck["N_16_sampled_rows_vs_dense_oracle_max_rel_err"] <= RTOL and ck["T_16_sampled_rows_vs_dense_oracle_max_rel_err"] <= RTOL
    assert ck["N_max_err_over_elementwise_bound"] <= 1.0 and ck["T_max_err_over_elementwise_bound"] <= 1.0     # per element: 4 K 2^-53 (sum |z||b| + centring magnitude)
    assert ck["centred_adjoint_identity_max_rel_err"] <= RTOL and ck["N_bitwise_repeatable"]
    assert leg["staging"]["single_orientation"] == 1
    # the stated size ran (a device with less free memory runs the largest SNP count that fits and says so: then this fails loudly, with the budget)
    assert leg["snps_run"] == 5_000_000, leg["byte_budget_GB"]
    assert leg["N"]["TFLOPs_call"] > 60.0 and leg["T"]["TFLOPs_call"] > 60.0, leg
