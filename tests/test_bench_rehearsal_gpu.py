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
    assert "all-reduce" in out["reduction"] and out["predicted_ms_per_step_from_per_shard"]["from_this_run"] > 0
    assert out["world_size"] == 2 and out["backend"] == "gloo" and out["rccl_ranks"] == 0          # the line records what ran the collective (rehearsal: gloo)
    assert out["unit"] == "GFLOP/s" and out["value"] > 0 and out["higher_is_better"] is True
    assert out["check"]["adjoint_identity_max_rel_err"] <= 1e-10          # 'N' (all-reduced over the ranks) against 'T' (sharded)
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert "cpu_baseline" not in out                                       # rank 0 at N = 1 only
    assert out["opt_in_engine"]["max_colwise_rel_diff_vs_f64_engine"] <= 1e-11
    gd = out["opt_in_engine_guarded"]
    assert gd["max_colwise_rel_diff_vs_f64_engine"] <= 1e-12 and gd["kernel_family_of_last_product"] == "k_gemm_i8"
    ex = out["opt_in_engine_exact"]
    assert ex["max_colwise_rel_diff_vs_f64_engine"] <= 1e-12 and ex["kernel_family_of_last_product"] == "k_gemm_i8" and 7 <= ex["digits_per_column"] <= 24


def _run_bench(extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--snps", "60002", "--indiv", "8000", "--ncol", "32", "--configs-scale", "0.02"] + extra
    r = subprocess.run(cmd, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-1500:]
    return json.loads(lines[0])


def test_bench_single_gpu_line_is_complete():
    """the N = 1 line at a small size: roofline (with traffic measured by the run's own rocprofv3 --pmc children, or null with the
    reason), cpu_baseline with the core count and how it was obtained, the ABI end-to-end leg (host B / C, bitwise the
    device-resident results) and the in-run parity checks against the CPU library and the dense oracle"""
    out = _run_bench([])
    assert out["n_gpus"] == 1 and out["unit"] == "GFLOP/s" and out["dtype"] == "f64" and out["vs_baseline"] is None
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert (isinstance(rf["traffic"], float) and rf["traffic"] > 0 and rf["traffic_detail"]["launches"] == 2) or "skipped" in rf["traffic_detail"]
    cb = out["cpu_baseline"]
    assert cb["cores"] >= 1 and cb["value"] > 0 and "cores_counted_as" in cb
    if "cpu_baseline_port" in out:            # oracle/_ref travelled with the push: the baseline is the reference's own CPU library, the port the extra
        assert cb["kind"] == "reference" and cb["port_T_output_bitwise_equal_to_reference_build"] is True and out["cpu_baseline_port"]["kind"].startswith("port")
    else:                                     # clean checkout: the tracked port, labelled as faster than what it restates
        assert cb["kind"].startswith("port (")
    ab = out["abi_end_to_end"]
    assert ab["bitwise_equal_to_device_resident_results"] is True and ab["max_GFLOPs"] >= ab["mean_GFLOPs"] > 0
    assert ab["plink2compressed_host_staging_s"] > 0 and ab["plink2compressed_snp_major_only_s"] > 0 and ab["staged_objects_reproduce_the_T_result_bitwise"] is True
    ck = out["check"]
    assert ck["gpu_T_rows_vs_cpu_library_max_rel_err"] <= 1e-11 and ck["gpu_N_64_sampled_rows_vs_dense_oracle_max_rel_err"] <= 1e-11
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
    out = _run_bench(["--gpus", "2", "--no-alt-engine"])
    assert out["n_gpus"] == 2 and "behind the C ABI" in out["config"]["workload"]
    assert out["check"]["adjoint_identity_max_rel_err"] <= 1e-10 and out["roofline"]["launches"] == 8
    assert out["abi_end_to_end"]["bitwise_equal_to_device_resident_results"] is True
    # the line explains itself: per shard the kernel time, the copies and the pushes; the reduction that ran and its kernel time; the
    # RCCL variant (not applicable with two shards on one device) and the hub variant beside it
    ps = out["per_shard"]
    assert ps["reduction"] == "p2p-fixed-order" and ps["reductions"] == 2 and ps["avg_reduce_kernel_ms"] > 0 and len(ps["shards"]) == 2
    for s in ps["shards"]:
        assert s["k_gemm_launches"] == 4 and s["avg_k_gemm_ms"] > 0 and s["operand_copies_in"] == 0 and s["result_copies_out"] == 0
        assert s["peer_access_to_root"] == -1 and s["partial_pushes"] == 0      # one device: nothing to push
    assert "skipped" in out["rccl_reduction"]
    # --reduce auto: RCCL is the timed reduction wherever every shard has a device of its own; here the two shards share one, and the line says so
    assert out["reduction"].startswith("p2p (rccl not applicable") and out["rccl_ranks"] == 0 and out["devices"] == 1
    assert out["predicted_ms_per_step_from_per_shard"]["from_this_run"] > 0
    hub = out["hub_operands_on_first_device"]
    assert hub["value"] > 0 and len(hub["avg_copy_in_ms_per_shard"]) == 2
