cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_staging_gpu.py tests/test_largegrid_gpu.py -x -q -m gpu 2>&1 | tail -2
python - <<'PY'
import time, torch, miraculix_amd as mx
from bench import synth_plink_device
mx.load_shared_library()
dev = torch.device("cuda", 0)
P = synth_plink_device(torch, 1_000_000, 12500, 1, dev)
T = mx.compressed_operations.transpose_genotype_matrix(P, 1_000_000, 50_000)
torch.cuda.synchronize(); t0 = time.perf_counter()
T = mx.compressed_operations.transpose_genotype_matrix(P, 1_000_000, 50_000)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"transpose 1M x 50k (12.5 GB in, 12.5 GB out): {dt*1e3:.1f} ms = {25.0/dt/1e3:.2f} TB/s")
t0 = time.perf_counter(); f = mx.read_plink.calc_freq(P, 1_000_000, 50_000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"allele freq 1M x 50k: {dt*1e3:.1f} ms = {12.5/dt/1e3:.2f} TB/s")
PY
