"""bench.py's N = 2 control flow on a one-GPU box: two ranks on cuda:0 with gloo in place of RCCL (MXA_BENCH_SINGLE_DEVICE /
MXA_BENCH_BACKEND rehearsal knobs).  Checks the contract of the JSON line and the cross-rank adjoint identity; not a measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_gpu_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MXA_BENCH_SINGLE_DEVICE="1", MXA_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--snps", "60002", "--indiv", "8000", "--ncol", "32"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "strong" and out["dtype"] == "f64"
    assert out["unit"] == "GFLOP/s" and out["value"] > 0 and out["higher_is_better"] is True
    assert out["check"]["adjoint_identity_max_rel_err"] <= 1e-10          # 'N' (all-reduced over the ranks) against 'T' (sharded)
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert "cpu_baseline" not in out                                       # rank 0 at N = 1 only
    assert out["opt_in_engine"]["max_colwise_rel_diff_vs_f64_engine"] <= 1e-11
