"""Launch-geometry regressions: output dimensions beyond the 65535 limit of gridDim.y (GRM post-processing with > 65535
columns, 2-bit transpose with > 65535 * 64 rows).  Device-resident operands only; checked on sampled rows / by round trip."""
import pytest

pytestmark = pytest.mark.gpu


def test_grm_postprocess_more_than_65535_columns():
    import torch
    import miraculix_amd as mx
    mx.load_shared_library()
    dev = torch.device("cuda", 0)
    n, k = 66000, 64
    g = torch.Generator(device=dev); g.manual_seed(1)
    b = torch.randint(0, 256, (n, k // 4), dtype=torch.uint8, device=dev, generator=g)
    miss = (b & 0x55) & ~((b >> 1) & 0x55)
    X = b ^ miss
    f = torch.full((k,), 0.3, dtype=torch.float64, device=dev)
    G = mx.crossproduct.grm(X, k, n, is_plink_format=True, do_scale=True, allele_freq=f)
    codes = torch.stack([(X >> (2 * q)) & 3 for q in range(4)], dim=2).reshape(n, -1)[:, :k].to(torch.float64)
    Z = torch.clamp(codes - 1, min=0)
    Zc = Z - Z.mean(dim=0, keepdim=True)
    idx = torch.tensor([0, 1, 65535, 65536, n - 1], device=dev)
    ref = (Zc[idx] @ Zc.T) / (2 * float((f * (1 - f)).sum()))
    assert float((G[idx] - ref).abs().max() / ref.abs().max()) < 1e-9
    assert torch.equal(G[idx][:, idx], G[:, idx][idx].t())


def test_transpose_more_than_4m_rows():
    import torch
    import miraculix_amd as mx
    mx.load_shared_library()
    dev = torch.device("cuda", 0)
    rows, cols = 4_500_000, 8
    g = torch.Generator(device=dev); g.manual_seed(2)
    P = torch.randint(0, 256, (rows, cols // 4), dtype=torch.uint8, device=dev, generator=g)
    T = mx.compressed_operations.transpose_genotype_matrix(P, rows, cols)
    back = mx.compressed_operations.transpose_genotype_matrix(T, cols, rows)
    assert torch.equal(back, P)
