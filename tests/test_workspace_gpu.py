"""The partial-sum workspace after a column peel (advisor finding of round 3).  ensure_workspace sizes d_P for the plan of all n columns; with
n = 4q + 1 / 4q + 2 / 4q + 3 the odd columns go through the exact int8 route and the fp64 MFMA kernel is launched with the plan of the remaining
4q columns, whose tile and K pieces differ -- at 100 000 SNPs x 30 000 individuals, 'N', n = 9..11 that plan writes 12.81 M doubles where the
unpeeled one needs 11.96 M.  The launch must find a workspace that holds its plan (grown on demand), and the result must match the oracle."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [10, 11])
def test_peeled_plan_fits_the_workspace_and_matches_the_oracle(n):
    import torch
    import miraculix_amd as mx
    from bench import sampled_rows_vs_oracle, stage_object
    L = mx.load_shared_library()
    L.mxa_plan_partial_doubles.restype = ctypes.c_long
    L.mxa_plan_partial_doubles.argtypes = [ctypes.c_long, ctypes.c_long, ctypes.c_int]
    L.mxa_partial_capacity.restype = ctypes.c_long
    L.mxa_partial_capacity.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    snps, indiv = 100_000, 30_000
    S = stage_object(torch, mx, dev, snps, indiv, n, 91, centered=True)
    dg = S["dg"]
    try:
        n4 = n - n % 4
        need_full, need_peeled = L.mxa_plan_partial_doubles(indiv, snps, n), L.mxa_plan_partial_doubles(indiv, snps, n4)
        assert need_peeled > need_full, "the shape no longer exercises the case: pick another from tests/test_abi_cpu.py's sweep"
        g = torch.Generator(device=dev); g.manual_seed(5)
        for trans in (False, True):
            k = indiv if trans else snps
            B = torch.randn((n, k), dtype=torch.float64, device=dev, generator=g).t()
            C = dg.dgemm_compressed_main(trans, S["obj"], B, snps, indiv)
            torch.cuda.synchronize()
            assert dg.last_path() == "k_gemm"
            gn = ctypes.c_int()
            L.mxa_last_geometry(None, None, ctypes.byref(gn), None, None, None)
            assert gn.value == n4                                    # the MFMA launch saw the multiple of 4: the odd columns were peeled
            assert L.mxa_partial_capacity(S["obj"]) >= (need_peeled if not trans else 0)
            err = sampled_rows_vs_oracle(torch, S, int(trans), B, C, list(range(n)), 1, nsample=32, seed=3)
            assert err <= 1e-11, (trans, err)
            assert torch.equal(C, dg.dgemm_compressed_main(trans, S["obj"], B, snps, indiv))
    finally:
        dg.free_compressed(S["obj"])
        S.clear()
        torch.cuda.empty_cache()
