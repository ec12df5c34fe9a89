"""bench.py's control flow for every form the driver runs, on a one-GPU box: N = 1, N = 2 in-process (virtual shards behind the C ABI) and N = 2 under the
launcher (two ranks on cuda:0 with gloo in place of RCCL: MXA_BENCH_SINGLE_DEVICE / MXA_BENCH_BACKEND rehearsal knobs).  Checks the contract of the JSON line
(<= 4 KB, the compact keys), that every form validates itself against the oracle before timing counts (`check.oracle_*`), that a deliberately mis-cut shard
boundary ends the run WITHOUT a number, and the detail file bench_detail.json.  Not a measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--snps", "60002", "--indiv", "8000", "--ncol", "32"]


def _line_and_detail(r):
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-1500:]
    assert len(lines[0]) <= 4096, len(lines[0])                          # the driver's record keeps the whole line
    out = json.loads(lines[0])
    assert out["detail"] == "bench_detail.json"
    return out, json.load(open(os.path.join(ROOT, "bench_detail.json")))


def _check_self_validation(out, n_ranks):
    ck = out["check"]
    assert ck["oracle_T_max_rel_err"] <= ck["tol"] == 1e-11 and ck["oracle_N_max_rel_err"] <= 1e-11      # both products against the oracle, before timing counts
    assert ck["rows_N"] == 64 and ck["rows_T"] >= 32 * min(n_ranks, 2)
    assert ck["adjoint_identity_max_rel_err"] <= 1e-10
    assert "rccl_ranks" in out and "world_size" in out


def _launch(extra_env=None, extra_args=()):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MXA_BENCH_SINGLE_DEVICE="1", MXA_BENCH_BACKEND="gloo", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL + list(extra_args)
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)


def test_bench_two_ranks_one_gpu_gloo():
    out, det = _line_and_detail(_launch())
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "strong" and out["dtype"] == "f64"
    assert "all-reduce" in out["reduction"] and out["predicted_ms_per_step"] > 0
    assert out["world_size"] == 2 and out["backend"] == "gloo" and out["rccl_ranks"] == 0          # the line records what ran the collective (rehearsal: gloo)
    assert out["peer_access_to_rank0"] == [-1, -1]                                                   # both ranks on one device here
    assert out["unit"] == "GFLOP/s" and out["value"] > 0 and out["higher_is_better"] is True
    _check_self_validation(out, 2)
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert "cpu_baseline" not in out                                       # rank 0 at N = 1 only
    assert len(out["k_gemm_ms_per_rank"]) == 2 and out["allreduce_alone_ms"] > 0
    assert det["opt_in_engine"]["max_colwise_rel_diff_vs_f64_engine"] <= 1e-11
    ex = det["opt_in_engine_exact"]
    assert ex["max_colwise_rel_diff_vs_f64_engine"] <= 1e-12 and ex["kernel_family_of_last_product"] == "k_gemm_i8" and 7 <= ex["digits_per_column"] <= 24


def test_bench_two_ranks_miscut_shard_boundary_aborts_without_a_number():
    """MXA_BENCH_TEST_MISCUT=1: rank 1 takes its rows of B four SNPs too early -- both products share the error, the adjoint identity holds, the oracle check
    must end the run on every rank without a JSON line"""
    r = _launch({"MXA_BENCH_TEST_MISCUT": "1"}, ["--no-alt-engine"])
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout[-800:]
    assert "differ from the oracle" in r.stderr, r.stderr[-1500:]


def _run_bench(extra, env=None, check=True):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--configs-scale", "0.02"] + extra
    r = subprocess.run(cmd, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=900, cwd=ROOT)
    return _line_and_detail(r) if check else r


def test_bench_single_gpu_line_is_complete():
    """the N = 1 line at a small size: roofline (with traffic measured by the run's own rocprofv3 --pmc children, or null with the
    reason), cpu_baseline with the core count, the ABI end-to-end leg (host B / C, bitwise the device-resident results), the in-run parity checks
    against the oracle (before timing) and against the CPU library, one compact key per BASELINE config leg; the details in bench_detail.json"""
    line, out = _run_bench([])
    assert line["n_gpus"] == 1 and line["unit"] == "GFLOP/s" and line["dtype"] == "f64" and line["vs_baseline"] is None
    assert line["world_size"] == 1 and line["rccl_ranks"] == 0
    _check_self_validation(line, 1)
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    rfd = out["roofline"]
    assert (isinstance(rf["traffic"], float) and rf["traffic"] > 0 and rfd["traffic_detail"]["launches"] == 2) or "skipped" in rfd["traffic_detail"]
    # both baselines, always: `cpu_baseline` = the reference's own library where oracle/_ref travelled with the push, else the port; `cpu_baseline_port` = the port
    cb, cp = line["cpu_baseline"], line["cpu_baseline_port"]
    assert cb["cores"] >= 1 and cb["value"] > 0 and cp["kind"] == "port" and cp["value"] > 0 and cb["kind"] in ("reference", "port")
    if cb["kind"] == "reference":
        assert out["cpu_baseline"]["port_T_output_bitwise_equal_to_reference_build"] is True
    else:
        assert cb["value"] == cp["value"]
    assert line["abi_bitwise_equal"] is True and line["abi_GFLOPs"] > 0
    ab = out["abi_end_to_end"]
    assert ab["max_GFLOPs"] >= ab["mean_GFLOPs"] > 0
    assert ab["plink2compressed_host_staging_s"] > 0 and ab["plink2compressed_snp_major_only_s"] > 0 and ab["staged_objects_reproduce_the_T_result_bitwise"] is True
    ck = line["check"]
    assert ck["gpu_T_rows_vs_cpu_library_max_rel_err"] <= 1e-11 and ck["gpu_N_64_sampled_rows_vs_dense_oracle_max_rel_err"] <= 1e-11
    # one compact key per leg in the line
    assert line["legs_parity_ok"] is True and "legs_failed" not in line
    for k in ("c3_ms_f4", "c3_ms_i8", "c5_step_ms", "c5_TBps"):
        assert line[k] > 0, k
    assert 0 < line["c3_frac_f4"] < 1 and 0 < line["c3_frac_i8"] < 1 and line["c3_exact"] is True and line["c5_bitwise_T_then_N"] is True
    assert len(line["c4_shard_frac"]) == 2 and len(line["c4_full_TFLOPs"]) == 2 and 0 < line["c5_frac_hbm"] < 1
    # the legs for BASELINE configs 3, 4 (shard) and 5 (shard), here at 2 % of their sizes: each carries its rates, its roofline fraction and its checks
    for k in ("config5_cg_step", "config4_shard", "config3_crossprod", "config5_full_8_virtual_shards", "config4_full_extent_8_virtual_shards", "config4_full_one_copy"):
        assert "failed" not in out[k], out[k]
    c5, c4, c3 = out["config5_cg_step"], out["config4_shard"], out["config3_crossprod"]
    assert c5["ms_per_cg_step"] > 0 and c5["check"]["gram_matvec_bitwise_equals_T_then_N"] is True and 0 < c5["frac_of_8_TBs_spec"] < 1
    assert c5["check"]["T_32_sampled_rows_vs_dense_oracle_max_rel_err"] <= 1e-11 and c5["check"]["N_32_sampled_rows_vs_dense_oracle_max_rel_err"] <= 1e-11
    for t in ("N", "T"):
        assert c4[t]["k_gemm_ms"] > 0 and 0 < c4[t]["frac_of_fp64_mfma_peak_kernel"] < 1
    assert c4["check"]["N_16_sampled_rows_vs_dense_oracle_max_rel_err"] <= 1e-11 and c4["check"]["T_16_sampled_rows_vs_dense_oracle_max_rel_err"] <= 1e-11
    for eng in ("k_crossprod_f4 (FP4 MFMA, default)", "k_crossprod_i8 (int8 MFMA)"):
        assert c3[eng]["kernel_ms"] > 0 and c3[eng]["check"]["four_256x256_tiles_and_mirrors_bit_exact_vs_int32_oracle"] is True
    # configs 5 and 4 at their "full extent" legs (8 virtual shards + one object), here at 2 % of the sizes
    c5f, c4f = out["config5_full_8_virtual_shards"], out["config4_full_extent_8_virtual_shards"]
    assert c5f["check"]["sharded_equals_one_object_bitwise_on_integer_vector"] is True and c4f["check"]["sharded_equals_one_object_bitwise_on_integer_B"] is True
    assert c5f["check"]["single_orientation_equals_two_copies_bitwise_on_integer_vector"] is True
    assert c5f["one_object_single_orientation"]["ms_per_gram_matvec"] > 0
    for name in ("8_virtual_shards", "one_object"):
        assert c5f[name]["ms_per_gram_matvec"] > 0 and c5f[name]["check"]["cg_bitwise_repeatable"] is True
        assert c5f[name]["check"]["T_32_sampled_rows_vs_dense_oracle_max_rel_err"] <= 1e-11 and c5f[name]["check"]["N_32_sampled_rows_vs_dense_oracle_max_rel_err"] <= 1e-11
        assert c4f[name]["N"]["ms_per_call"] > 0 and c4f[name]["check"]["N_16_sampled_rows_vs_dense_oracle_max_rel_err"] <= 1e-11


def test_bench_inprocess_two_shards_behind_the_c_abi():
    """python bench.py --gpus 2 without a launcher: the SNP shards live behind the C ABI (MIRACULIX_NUM_GPUS); on a one-GPU box the
    two shards share the device (virtual shards)"""
    line, out = _run_bench(["--gpus", "2", "--no-alt-engine"])
    assert line["n_gpus"] == 2 and "behind the C ABI" in out["config"]["workload"] and line["config"]["form"].startswith("in-process")
    _check_self_validation(line, 2)
    assert line["roofline"]["launches"] == 8
    assert line["abi_bitwise_equal"] is True
    # the line explains itself: who reduced, on how many devices, peer access per shard, the kernel time per shard; the tables are in the detail file
    assert line["reduction"].startswith("p2p (rccl not applicable") and line["rccl_ranks"] == 0 and line["devices"] == 1 and line["shards"] == 2
    assert line["peer_access_to_root"] == [-1, -1] and len(line["k_gemm_ms_per_shard"]) == 2 and line["reduce_kernel_ms"] > 0
    assert line["predicted_ms_per_step"] > 0 and line["hub_GFLOPs"] > 0
    ps = out["per_shard"]
    assert ps["reduction"] == "p2p-fixed-order" and ps["reductions"] == 2 and ps["avg_reduce_kernel_ms"] > 0 and len(ps["shards"]) == 2
    for s in ps["shards"]:
        assert s["k_gemm_launches"] == 4 and s["avg_k_gemm_ms"] > 0 and s["operand_copies_in"] == 0 and s["result_copies_out"] == 0
        assert s["peer_access_to_root"] == -1 and s["partial_pushes"] == 0      # one device: nothing to push
    assert "skipped" in out["rccl_reduction"]
    hub = out["hub_operands_on_first_device"]
    assert hub["value"] > 0 and len(hub["avg_copy_in_ms_per_shard"]) == 2


def test_bench_inprocess_miscut_shard_boundary_aborts_without_a_number():
    r = _run_bench(["--gpus", "2", "--no-alt-engine", "--no-abi", "--no-configs"], env={"MXA_BENCH_TEST_MISCUT": "1"}, check=False)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout[-800:]
    assert "differ from the oracle" in r.stderr, r.stderr[-1500:]
