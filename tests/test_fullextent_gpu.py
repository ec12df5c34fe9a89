"""BASELINE.json configs 5 and 4 at their full extent on ONE MI355X (288 GB of HBM hold what the reference spreads over 8 GPUs), through the
8 SNP shards of the 8-GPU run behind the plain C symbols (MIRACULIX_NUM_GPUS=8, all shards on this device) and as one plain object.  The
oracle cannot run at these sizes: sampled rows against the long-double dense oracle on the extracted packed rows, plus size-independent
properties (adjoint identity, exact-integer partition independence, bitwise repeatability, the CG residual recomputed).  The test bodies
are bench.py's legs `config5_full_8_virtual_shards` / `config4_full_extent_8_virtual_shards`, so GPUTEST and the bench line check the
same thing.  (A file of its own: the module-scoped fixtures of test_fullsize_configs_gpu.py hold ~240 GB until that module ends.)"""
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-11   # stated fp64 tolerance of the path


def _mods():
    import torch
    import miraculix_amd as mx
    mx.load_shared_library()
    torch.cuda.empty_cache()
    return torch, mx, torch.device("cuda", 0)


def _assert_leg(leg):
    from bench import leg_checks_ok
    assert "failed" not in leg and leg_checks_ok(leg), leg


def test_c5_full_extent_8_virtual_shards_and_one_object():
    """BASELINE config 5 at its stated size on one GPU: 2 000 000 SNPs x 100 000 individuals (2 x 50 GB packed), >= 20 CG iterations
    (examples/grm_solve_cg.py = the reference's examples/iterative_solver/grm_solve_cg.jl:74-84,108-134) through mxa_gram_matvec, on the object
    cut into the 8 SNP shards of the 8-GPU run (MIRACULIX_NUM_GPUS=8, all on this device) and on one plain object: sampled rows of both
    products vs the long-double oracle <= 1e-11, sharded == single object bit for bit on an integer-valued vector, the CG loop bitwise
    repeatable and its reported residual confirmed by a separately computed one.  The same function is bench.py's leg
    `config5_full_8_virtual_shards`."""
    import bench
    torch, mx, dev = _mods()
    L = mx.load_shared_library()
    leg = bench.config5_full_leg(torch, mx, L, dev, 2_000_000, 100_000, shards=8, iters=20)
    _assert_leg(leg)
    for name in ("8_virtual_shards", "one_object", "one_object_single_orientation"):
        ck = leg[name]["check"]
        assert leg[name]["cg_iterations"] == 20
        assert ck["T_32_sampled_rows_vs_dense_oracle_max_rel_err"] <= RTOL and ck["N_32_sampled_rows_vs_dense_oracle_max_rel_err"] <= RTOL
        assert ck["gram_matvec_vs_T_then_N_max_rel_err"] <= RTOL and ck["cg_bitwise_repeatable"] and ck["cg_residual_consistent_ok"] and ck["cg_converging_ok"]
    assert leg["check"]["sharded_equals_one_object_bitwise_on_integer_vector"] and leg["check"]["integer_gram_equals_T_then_N_bitwise"]
    assert leg["check"]["single_orientation_equals_two_copies_bitwise_on_integer_vector"]
    held = [leg[k]["device_memory_held_by_the_object_GB"] for k in ("one_object", "one_object_single_orientation")]
    assert held[1] <= 0.56 * held[0]                                   # one packed copy instead of two (plus the same workspace)


def test_c4_full_snp_extent_8_virtual_shards_and_one_object():
    """BASELINE config 4's full 5 000 000-SNP extent (individuals reduced to 25 000 so that 2 x 31 GB fit one GPU), ncol = 128, centred,
    through the 8 SNP shards of the 8-GPU run (625 000 SNPs each) behind the plain dgemm_compressed symbol, and as one object: sampled rows
    vs the centred long-double oracle, centred adjoint identity, repeatability, sharded == single object bit for bit on integer-valued B.
    bench.py's leg `config4_full_extent_8_virtual_shards`."""
    import bench
    torch, mx, dev = _mods()
    L = mx.load_shared_library()
    leg = bench.config4_full_extent_leg(torch, mx, L, dev, 5_000_000, 25_000, 128, shards=8)
    _assert_leg(leg)
    for name in ("8_virtual_shards", "one_object"):
        ck = leg[name]["check"]
        assert ck["N_16_sampled_rows_vs_dense_oracle_max_rel_err"] <= RTOL and ck["T_16_sampled_rows_vs_dense_oracle_max_rel_err"] <= RTOL
        assert ck["centred_adjoint_identity_max_rel_err"] <= RTOL and ck["N_bitwise_repeatable"]
    assert leg["check"]["sharded_equals_one_object_bitwise_on_integer_B"]
