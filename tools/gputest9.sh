cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_staging_gpu.py tests/test_crossprod_gpu.py -x -q -m gpu 2>&1 | tail -3
python - <<'PY'
import torch, numpy as np, miraculix_amd as mx
# GRM post-processing with more than 65535 columns and a transpose with more than 65535*64 rows: grid-limit regression checks
mx.load_shared_library()
dev = torch.device("cuda", 0)
n, k = 70000, 64
g = torch.Generator(device=dev); g.manual_seed(1)
b = torch.randint(0, 256, (n, k // 4), dtype=torch.uint8, device=dev, generator=g)
miss = (b & 0x55) & ~((b >> 1) & 0x55); X = b ^ miss
f = torch.full((k,), 0.3, dtype=torch.float64, device=dev)
G = mx.crossproduct.grm(X, k, n, is_plink_format=True, do_scale=True, allele_freq=f)
codes = torch.stack([(X >> (2 * q)) & 3 for q in range(4)], dim=2).reshape(n, -1)[:, :k].to(torch.float64)
Z = torch.clamp(codes - 1, min=0)
Zc = Z - Z.mean(dim=0, keepdim=True)
idx = torch.tensor([0, 1, 65535, 65536, 69999], device=dev)
ref = (Zc[idx] @ Zc.T) / (2 * float((f * (1 - f)).sum()))
err = float((G[idx] - ref).abs().max() / ref.abs().max())
print("grm 70000 cols rel err", err); assert err < 1e-9
rows, cols = 4_500_000, 8
P = torch.randint(0, 256, (rows, cols // 4), dtype=torch.uint8, device=dev, generator=g)
T = mx.compressed_operations.transpose_genotype_matrix(P, rows, cols)
back = mx.compressed_operations.transpose_genotype_matrix(T, cols, rows)
assert torch.equal(back, P); print("transpose 4.5M rows round trip ok")
PY
