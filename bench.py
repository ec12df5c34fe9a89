#!/usr/bin/env python3
"""bench.py -- headline benchmark: effective GFLOP/s of dgemm_compressed (2-bit SNP x fp64) on N MI355X.

Workload (BASELINE.json configs[1]): 1M SNPs x 50k individuals, n=32, dgemm_compressed 'N' and 'T', uncentred,
synthetic PLINK data generated on the device.  One step = one 'N' multiply (+ the fp64 all-reduce of the indiv x n
result when N > 1) and one 'T' multiply; flops per step = 2 * (2 * snps * indiv * n).  Inputs (packed genotypes, B, C)
are resident in HBM when the timed region starts.  N > 1: the SNP dimension is sharded over the ranks (strong scaling,
total work fixed), one process per GPU, RCCL all-reduce.

python bench.py --gpus N --steps K --warmup W   (N > 1: launched by torch.distributed.run, one rank per GPU)
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (spec); measured bare-MFMA ceiling 75.8


def synth_plink_device(torch, rows, row_bytes, seed, device):
    """random PLINK bytes without the missing code 01: fields 00 (p=1/2), 10 (1/4), 11 (1/4); generated in chunks"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((rows, row_bytes), dtype=torch.uint8, device=device)
    chunk = max(1, (256 << 20) // max(1, row_bytes))
    for r0 in range(0, rows, chunk):
        r1 = min(rows, r0 + chunk)
        b = torch.randint(0, 256, (r1 - r0, row_bytes), dtype=torch.uint8, device=device, generator=g)
        miss = (b & 0x55) & ~((b >> 1) & 0x55)  # low bit set, high bit clear -> 01
        out[r0:r1] = b ^ miss
    return out


def synth_genotypes_device(torch, rows, cols, seed, device, p_along="rows"):
    """PLINK 2-bit rows (rows x ceil(cols/4) bytes) with the distribution of SURVEY.md 8(d) (mirrors the reference's
    create_sim_file.jl:12): allele frequency p ~ U(0.1, 0.6) per SNP, genotype g ~ Binomial(2, p) i.i.d., no missings, codes
    0 -> 00, 1 -> 10, 2 -> 11, row padding bits zero.  p_along = "rows": one p per row (SNP-major matrix); "cols": one p per
    column (individual-major matrix).  Generated on the device in chunks."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rb = (cols + 3) // 4
    out = torch.empty((rows, rb), dtype=torch.uint8, device=device)
    p_all = torch.rand(rows if p_along == "rows" else cols, device=device, generator=g) * 0.5 + 0.1
    w = torch.tensor([1, 4, 16, 64], dtype=torch.uint8, device=device)
    chunk = max(1, (128 << 20) // max(1, 4 * rb))
    for r0 in range(0, rows, chunk):
        r1 = min(rows, r0 + chunk)
        u = torch.rand((r1 - r0, 4 * rb), device=device, generator=g)
        p = p_all[r0:r1, None] if p_along == "rows" else torch.nn.functional.pad(p_all, (0, 4 * rb - cols))[None, :]
        q0 = (1.0 - p) ** 2                       # P(g = 0)
        q1 = q0 + 2.0 * p * (1.0 - p)             # P(g <= 1)
        code = (u >= q0).to(torch.uint8) * 2 + (u >= q1).to(torch.uint8)   # 0 -> 00, 1 -> 10 (2), 2 -> 11 (3)
        if 4 * rb > cols:
            code[:, cols:] = 0
        out[r0:r1] = (code.view(r1 - r0, rb, 4) * w).sum(dim=2, dtype=torch.uint8)
    return out


def cpu_baseline(seconds_budget=20.0):
    """CPU 5codes baseline on a bounded sample of the same workload (same n, both ops), timed on this host's cores.
    kind 'reference' when oracle/_ref (the reference's own library built from its sources) travelled with the repo,
    else 'port' (oracle/oracle.c, bit-exact with the reference on the pinned fixtures)."""
    import numpy as np
    from _util import Oracle, have_reference, make_B, make_problem, run_reference
    snps, indiv, n = 40000, 10000, 32
    cores = min(os.cpu_count() or 1, 16)
    prob = make_problem(snps, indiv, n, seed=42)
    flops = 2.0 * snps * indiv * n
    times = {}
    if have_reference():
        kind = "reference"
        for trans in (0, 1):
            B = make_B(indiv if trans else snps, n, seed=43)
            _, t = run_reference(prob, trans, B, centered=False, variant=256, cores=cores, reps=3)
            times[trans] = t
    else:
        kind = "port"
        os.environ.setdefault("OMP_NUM_THREADS", str(cores))
        o = Oracle()
        h = o.five_create(prob, cores)
        for trans in (0, 1):
            B = make_B(indiv if trans else snps, n, seed=43)
            best = 1e30
            for _ in range(3):
                t0 = time.perf_counter()
                o.five_dgemm(h, trans, prob, B, 0)
                best = min(best, time.perf_counter() - t0)
            times[trans] = best
        o.five_free(h)
    gflops = 2 * flops / (times[0] + times[1]) * 1e-9
    return {"value": round(gflops, 2), "unit": "GFLOP/s", "cores": cores, "kind": kind,
            "sample": f"{snps} SNPs x {indiv} indiv, n={n}, one 'N' + one 'T' multiply, uncentred, best of 3 (N {times[0]:.3f}s, T {times[1]:.3f}s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--snps", type=int, default=1_000_000)
    ap.add_argument("--indiv", type=int, default=50_000)
    ap.add_argument("--ncol", type=int, default=32)
    ap.add_argument("--centered", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-engine", action="store_true", help="skip the extra (untimed, informational) pass with the opt-in int8 engine")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # rehearsal knobs (never used by the driver): MXA_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and MXA_BENCH_BACKEND=gloo
    # replaces RCCL, so the N > 1 control flow can be exercised on a one-GPU box
    if os.environ.get("MXA_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("MXA_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    os.environ["HIP_DEVICE"] = str(local_rank)
    force_dist = os.environ.get("MXA_BENCH_FORCE_DIST") == "1"   # exercise the RCCL path with a 1-rank group
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import miraculix_amd as mx
    from miraculix_amd.distributed import HipLocalEngine, ShardedGenotypeOperator, shard_bounds
    L = mx.load_shared_library()
    L.mxa_set_engine(0)   # the headline number is the fp64 engine, whatever MXA_ENGINE says

    snps, indiv, n = args.snps, args.indiv, args.ncol
    b, e = shard_bounds(snps, world, rank)
    snps_loc = e - b
    # ---- synthetic data, generated on the device (no PLINK binary needed)
    plink = synth_plink_device(torch, snps_loc, (indiv + 3) // 4, 42 + rank, device)           # SNP-major
    plink_t = torch.empty((indiv, (snps_loc + 3) // 4), dtype=torch.uint8, device=device)        # individual-major
    assert L.mxa_transpose_2bit(mx.lib.ptr(plink), snps_loc, indiv, mx.lib.ptr(plink_t)) == 0
    freq = torch.empty(snps_loc, dtype=torch.float64, device=device)
    assert L.mxa_allele_freq(mx.lib.ptr(plink), snps_loc, indiv, mx.lib.ptr(freq)) == 0
    eng = HipLocalEngine(plink, plink_t, snps_loc, indiv, freq, n, centered=bool(args.centered))
    del plink, plink_t
    torch.cuda.empty_cache()
    op = ShardedGenotypeOperator(eng, snps, indiv)
    op.force_collective = force_dist

    g = torch.Generator(device=device); g.manual_seed(43)
    B_N = torch.randn((n, snps), dtype=torch.float64, device=device, generator=g)[:, b:e].contiguous().t()   # snps_loc x n, column-major
    B_T = torch.randn((n, indiv), dtype=torch.float64, device=device, generator=g).t()                       # indiv x n
    C_N = torch.zeros((n, indiv), dtype=torch.float64, device=device).t()
    C_T = torch.zeros((n, snps_loc), dtype=torch.float64, device=device).t()

    def step():
        _, work = op.matmul_N(B_N, out=C_N, async_op=True)
        op.matmul_T(B_T, out=C_T)
        if work is not None:
            work.wait()

    def sync():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    L.mxa_profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    launches, total_ms = ctypes.c_int(0), ctypes.c_double(0.0)
    L.mxa_profile_get(ctypes.byref(launches), ctypes.byref(total_ms))

    # parity check on every run (size-independent property; the oracle cannot run at this size): the adjoint identity
    # <B_T[:,j], Z B_N[:,j]> == <Z^T B_T[:,j], B_N[:,j]> ties the 'N' result (individual-major copy, all-reduced over the
    # SNP shards) to the 'T' result (SNP-major copy) column by column.
    lhs = (B_T * C_N).sum(dim=0)
    rhs = (C_T * B_N).sum(dim=0)
    if dist.is_initialized():
        dist.all_reduce(rhs, op=dist.ReduceOp.SUM)
    adj_err = float(((lhs - rhs).abs() / lhs.abs().clamp_min(1.0)).max())
    if not (adj_err <= 1e-10):
        raise SystemExit(f"bench.py: adjoint identity violated (rel err {adj_err:.3e}): results are wrong, no number reported")
    flops_step = 2 * 2.0 * snps * indiv * n
    # informational second pass, outside the timed region: the same steps with the opt-in int8 engine (exact 7 x 8-bit slicing of
    # B, include/miraculix_amd.h mxa_set_engine).  Reported beside the headline, never as `value`.
    alt = None
    if not args.no_alt_engine:
        C_N64, C_T64 = C_N.clone(), C_T.clone()
        L.mxa_set_engine(1)
        step(); sync()
        L.mxa_profile_reset()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dt8 = time.perf_counter() - t1
        la8, ms8 = ctypes.c_int(0), ctypes.c_double(0.0)
        L.mxa_profile_get(ctypes.byref(la8), ctypes.byref(ms8))
        L.mxa_set_engine(0)
        if dist.is_initialized():
            t = torch.tensor([dt8], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt8 = float(t.item())
        dN = float(((C_N - C_N64).abs().amax(dim=0) / C_N64.abs().amax(dim=0)).max())
        dT = float(((C_T - C_T64).abs().amax(dim=0) / C_T64.abs().amax(dim=0)).max())
        alt = {"engine": "i8: B split exactly into 7 radix-256 digits per column, v_mfma_i32_32x32x32_i8, exact int32 sums, fp64 recombination",
               "value": round(flops_step * args.steps / dt8 * 1e-9, 1), "unit": "GFLOP/s (fp64-equivalent: same 2*snps*indiv*ncol count)",
               "ms_per_step": round(dt8 / args.steps * 1e3, 3), "avg_kernel_ms": round(ms8.value / max(1, la8.value), 3),
               "int8_ops_per_s_P": round(2.0 * snps_loc * indiv * n * 7 / (ms8.value / max(1, la8.value) * 1e-3) * 1e-15, 3),
               "max_colwise_rel_diff_vs_f64_engine": max(dN, dT)}
        del C_N64, C_T64
    value = flops_step * args.steps / dt * 1e-9
    ms_per_step = dt / args.steps * 1e3
    # dominant kernel: k_gemm; algorithmic flops per launch on this rank = 2 * snps_loc * indiv * n
    flops_launch = 2.0 * snps_loc * indiv * n
    avg_ms = total_ms.value / max(1, launches.value)
    achieved = flops_launch / (avg_ms * 1e-3) * 1e-12 if avg_ms > 0 else 0.0

    # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a
    # pass, and bench.py cannot run under its own profiler): tools/pmc_traffic.py stores the per-launch figure measured on
    # this exact workload under profiles/; it is reported only when the workload matches, else null.
    traffic = None
    try:
        if world == 1 and (snps, indiv, n) == (1_000_000, 50_000, 32):
            cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))
            if cands:
                traffic = round(json.load(open(os.path.join(ROOT, "profiles", cands[-1])))["traffic_bytes_per_launch_avg"] / 1e9, 3)
    except Exception:
        traffic = None
    if rank == 0:
        out = {
            "metric": "effective GFLOP/s for dgemm_compressed (2-bit SNP x fp64)",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{snps} SNPs x {indiv} indiv, ncol={n}, dgemm_compressed 'N' + 'T' per step, "
                                   f"{'centred' if args.centered else 'uncentred'}, SNP-sharded over {world} GPU(s)",
                       "snps": snps, "indiv": indiv, "ncol": n, "parallelism": f"snp-shard{world}"},
            "check": {"adjoint_identity_max_rel_err": adj_err, "tolerance": 1e-10},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "GB per launch (rocprofv3 PMC, profiles/)",
                         "algorithmic_bytes_per_launch_GB": round((snps_loc * ((indiv + 3) // 4) + 8.0 * (snps_loc + indiv) * n) / 1e9, 3),
                         "kernel": "k_gemm<8,8> (v_mfma_f64_4x4x4_4b_f64)", "launches": launches.value, "avg_launch_ms": round(avg_ms, 3)},
        }
        if alt is not None:
            out["opt_in_engine"] = alt
        if not args.no_cpu_baseline and world == 1:   # CPU baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    eng.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
