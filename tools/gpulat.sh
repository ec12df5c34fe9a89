cd $GRAFT_REPO_ROOT
python - <<'PY'
import time, numpy as np, sys
sys.path.insert(0, "tests")
import miraculix_amd as mx
from _util import make_problem, make_B
mx.load_shared_library()
dg = mx.dgemm_compressed
prob = make_problem(1000, 500, 1, seed=1)
dg.set_options(use_gpu=True, not_center=False, verbose=0)
obj = dg.init_compressed(prob["plink"], prob["plink_t"], 1000, 500, prob["f"], 10)
for n in (1, 10):
    for trans in (False, True):
        k = 500 if trans else 1000
        B = np.asfortranarray(make_B(k, n, seed=2).T)
        C = dg.dgemm_compressed_main(trans, obj, B, 1000, 500)
        t0 = time.perf_counter()
        for _ in range(200):
            dg.dgemm_compressed_main(trans, obj, B, 1000, 500, out=C)
        dt = (time.perf_counter() - t0) / 200
        print(f"config 1 (1000 x 500) n={n} {'T' if trans else 'N'}: {dt*1e6:.0f} us per call through the host-pointer ABI")
dg.free_compressed(obj)
PY
timeout -k 10 600 python -m pytest tests/test_dgemm_gpu.py tests/test_shard_gpu.py -x -q -m gpu 2>&1 | tail -2
