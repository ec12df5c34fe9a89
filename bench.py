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


def physical_cores():
    """(threads to use, how they were counted): the physical cores this process may use -- unique (physical id, core id) pairs of
    /proc/cpuinfo restricted to the scheduler affinity mask, capped by the cgroup CPU quota of the container (a 256-thread host
    hands a one-GPU job a 16-CPU share; more OpenMP threads than that only fight for the same quota)."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        return os.cpu_count() or 1, "os.cpu_count()"
    n, how = len(allowed), f"len(os.sched_getaffinity(0)) = {len(allowed)}"
    try:
        cores, cur = set(), {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif not line.strip() and cur:
                if int(cur.get("processor", -1)) in allowed:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
        if cores:
            n, how = len(cores), f"{len(cores)} unique (physical id, core id) pairs of /proc/cpuinfo within the affinity mask of {len(allowed)} logical CPUs"
    except OSError:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                   # cgroup v2
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())              # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n, how = max(1, int(quota)), how + f", capped by the cgroup CPU quota of {quota:g} CPUs"
    return n, how


def cpu_baseline_and_check(torch, mx, plink_dev, freq_dev, snps, indiv, n, B_T, C_T, B_N, C_N, plink_t_rows_fn):
    """CPU 5codes baseline on a bounded sample of the SAME workload -- the first SAMPLE_SNPS SNPs of the bench matrix, all
    individuals, same n, one 'N' + one 'T' multiply -- timed on this host's physical cores (reference harness
    utils/benchmark/benchmark.f90:185-209).  kind 'reference' when oracle/_ref (the reference's own library built from its
    sources) travelled with the repo, else 'port' (oracle/oracle.c, bit-exact with it on the pinned fixtures).
    The same leg is the in-run parity check against the checker: rows [0, SAMPLE_SNPS) of the GPU's 'T' result are compared with
    the CPU library's own output on the sample (same packed rows, same B), and 64 sampled individuals of the GPU's 'N' result with
    the long-double dense oracle.  Everything under oracle/ is used here as baseline / checker only."""
    import numpy as np
    from _util import Oracle, have_reference, run_reference
    sample = min(snps, 100_000)
    cores, how = physical_cores()
    plink = plink_dev[:sample].cpu().numpy()
    f = freq_dev[:sample].cpu().numpy()
    prob = dict(snps=sample, indiv=indiv, plink=np.ascontiguousarray(plink), plink_t=None, f=f)
    Bt = np.ascontiguousarray(B_T.t().cpu().numpy())                                   # n x indiv
    Bn = np.ascontiguousarray(B_N[:sample].t().cpu().numpy())                          # n x sample
    times, Ct_cpu = {}, None
    o = Oracle()
    if have_reference():
        kind = "reference"
        _, times[0] = run_reference(prob, 0, Bn, centered=False, variant=256, cores=cores, reps=2)
        Ct_cpu, times[1] = run_reference(prob, 1, Bt, centered=False, variant=256, cores=cores, reps=2)
    else:
        kind = "port"
        os.environ["OMP_NUM_THREADS"] = str(cores)
        h = o.five_create(prob, cores)
        for trans, B in ((0, Bn), (1, Bt)):
            best = 1e30
            for _ in range(2):
                t0 = time.perf_counter()
                C = o.five_dgemm(h, trans, prob, B, 0)
                best = min(best, time.perf_counter() - t0)
            times[trans] = best
            if trans:
                Ct_cpu = C
        o.five_free(h)
    flops = 2.0 * sample * indiv * n
    gflops = 2 * flops / (times[0] + times[1]) * 1e-9
    # parity: GPU 'T' rows of the sample against the CPU library
    got = C_T[:sample].t().cpu().numpy()
    err_t = float(np.abs(got - Ct_cpu[:, :sample]).max() / np.abs(Ct_cpu[:, :sample]).max())
    # parity: 64 sampled individuals of the GPU 'N' result against the dense long-double oracle on their extracted rows
    rng = np.random.default_rng(1)
    ii = np.sort(rng.choice(indiv, 64, replace=False))
    rows = plink_t_rows_fn(ii)                                                         # 64 x ceil(snps/4) PLINK bytes
    sub_plink = o.transpose_2bit(np.ascontiguousarray(rows), 64, snps)
    cols = [0, n - 1] if n > 1 else [0]
    ref = o.dgemm_dense(0, dict(snps=snps, indiv=64, plink=sub_plink, plink_t=rows, f=np.zeros(snps)), np.ascontiguousarray(B_N[:, cols].t().cpu().numpy()), 0)
    got_n = C_N[torch.from_numpy(ii).to(C_N.device)][:, cols].t().cpu().numpy()
    err_n = float(np.abs(got_n - ref).max() / np.abs(ref).max())
    base = {"value": round(gflops, 2), "unit": "GFLOP/s", "cores": cores, "cores_counted_as": how, "kind": kind,
            "sample": f"first {sample} SNPs of the bench matrix x {indiv} indiv, n={n}, one 'N' + one 'T' multiply, uncentred, AVX2 variant 256, "
                      f"best of 2 (N {times[0]:.3f}s, T {times[1]:.3f}s)"}
    check = {"gpu_T_rows_vs_cpu_library_max_rel_err": err_t, "gpu_N_64_sampled_rows_vs_dense_oracle_max_rel_err": err_n, "checker_tolerance": 1e-11}
    return base, check


def pmc_child(args):
    """child of measure_traffic(): stage the workload and run one 'N' and one 'T' product, nothing else (run under rocprofv3 --pmc)"""
    import torch
    import miraculix_amd as mx
    L = mx.load_shared_library()
    L.mxa_set_engine(0)
    dev = torch.device("cuda", 0)
    snps, indiv, n = args.snps, args.indiv, args.ncol
    plink = synth_genotypes_device(torch, snps, indiv, 42, dev)
    plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, snps, indiv)
    f = mx.read_plink.calc_freq(plink, snps, indiv)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=not args.centered, verbose=0)
    obj = dg.init_compressed(plink, plink_t, snps, indiv, f, n)
    del plink, plink_t
    g = torch.Generator(device=dev); g.manual_seed(43)
    for trans in (False, True):
        B = torch.randn((n, indiv if trans else snps), dtype=torch.float64, device=dev, generator=g).t()
        dg.dgemm_compressed_main(trans, obj, B, snps, indiv)
    torch.cuda.synchronize()
    dg.free_compressed(obj)


def measure_traffic(args):
    """HBM traffic of the dominant kernel for THIS workload, measured now: two child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (the two counters do not fit one pass; --kernel-trace only, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes), one 'N' and one 'T' launch each.  Units: the counters are KiB.  Read side
    x `read_factor`: the guide's gfx950 correction is x2 for wide coalesced streams; profiles/r02_pmc_calibration.json holds the factor
    calibrated here on a known-size stream through the same 1-KiB LDS-DMA units (tools/pmc_calibrate.*), used when present.
    Returns (GB per launch or None, detail dict)."""
    import collections, csv, glob, shutil, subprocess, tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, {"skipped": "rocprofv3 not found"}
    if any("rocprof" in (os.environ.get(k) or "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")) or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD"):
        return None, {"skipped": "this run is itself under a profiler (the PMC passes cannot be nested); run bench.py plainly for roofline.traffic"}
    factor, calib = 2.0, "guide default x2 (uncalibrated for this access pattern)"
    try:
        cj = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_calibration.json")))
        factor, calib = float(cj["fetch_size_factor_lds_dma_1KiB_units"]), "profiles/r02_pmc_calibration.json"
    except Exception:
        pass
    vals = {}
    tmp = tempfile.mkdtemp(prefix="mxa_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [rocprof, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
                   "--snps", str(args.snps), "--indiv", str(args.indiv), "--ncol", str(args.ncol), "--centered", str(args.centered)]
            r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True, timeout=600, env=dict(os.environ, TMPDIR=tmp))
            src = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if r.returncode != 0 or not src:
                return None, {"skipped": f"rocprofv3 --pmc {ctr} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}"}
            agg = collections.OrderedDict()
            for row in csv.DictReader(open(src[0])):
                if "k_gemm<" in row["Kernel_Name"] and row.get("Counter_Name", ctr) == ctr:
                    agg[row["Dispatch_Id"]] = agg.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            vals[ctr] = list(agg.values())
    except Exception as e:   # a profiler problem must not take the benchmark down
        return None, {"skipped": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if not vals.get("FETCH_SIZE") or len(vals["FETCH_SIZE"]) != len(vals.get("WRITE_SIZE", [])):
        return None, {"skipped": "no k_gemm dispatch in the counter output"}
    reads = [factor * x * 1024 for x in vals["FETCH_SIZE"]]
    writes = [x * 1024 for x in vals["WRITE_SIZE"]]
    per_launch = (sum(reads) + sum(writes)) / len(reads)
    return round(per_launch / 1e9, 3), {"launches": len(reads), "raw_FETCH_SIZE_GB": [round(x * 1024 / 1e9, 3) for x in vals["FETCH_SIZE"]],
                                        "WRITE_SIZE_GB": [round(w / 1e9, 3) for w in writes], "read_factor": factor, "read_factor_source": calib}


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
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic")
    ap.add_argument("--no-abi", action="store_true", help="skip the ABI end-to-end leg (host B / C through dgemm_compressed)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # single process asked for several GPUs (no torch.distributed launcher): the SNP shards live BEHIND the C ABI
    # (MIRACULIX_NUM_GPUS, mxa_multi.cpp) -- the path a Julia / Fortran caller gets.  The driver's N > 1 runs use one rank per GPU.
    inprocess = world == 1 and args.gpus > 1
    # roofline.traffic is measured first, before this process touches the GPU (N = 1 only; the children profile the same workload)
    traffic, traffic_detail = None, {"skipped": "N > 1" if (world > 1 or inprocess) else "--no-pmc"}
    if world == 1 and not inprocess and not args.no_pmc:
        traffic, traffic_detail = measure_traffic(args)

    import torch
    import torch.distributed as dist
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
    b, e = (0, snps) if inprocess else shard_bounds(snps, world, rank)
    snps_loc = e - b
    if snps_loc <= 0:
        raise SystemExit(f"bench.py: rank {rank} of {world} has no SNPs ({snps} SNPs): use fewer ranks")
    # ---- synthetic data, generated on the device (p_s ~ U(0.1, 0.6), g ~ Binomial(2, p_s), no missings)
    plink = synth_genotypes_device(torch, snps_loc, indiv, 42 + rank, device)                    # SNP-major
    plink_t = torch.empty((indiv, (snps_loc + 3) // 4), dtype=torch.uint8, device=device)        # individual-major
    assert L.mxa_transpose_2bit(mx.lib.ptr(plink), snps_loc, indiv, mx.lib.ptr(plink_t)) == 0
    freq = torch.empty(snps_loc, dtype=torch.float64, device=device)
    assert L.mxa_allele_freq(mx.lib.ptr(plink), snps_loc, indiv, mx.lib.ptr(freq)) == 0
    if inprocess:
        os.environ["MIRACULIX_NUM_GPUS"] = str(args.gpus)
    eng = HipLocalEngine(plink, plink_t, snps_loc, indiv, freq, n, centered=bool(args.centered))
    os.environ.pop("MIRACULIX_NUM_GPUS", None)
    n_shards = mx.dgemm_compressed.num_shards(eng.obj)
    keep_raw = world == 1 and not inprocess and not args.no_cpu_baseline       # the CPU-baseline / parity leg samples the raw matrices
    if not keep_raw:
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
    value = flops_step * args.steps / dt * 1e-9
    ms_per_step = dt / args.steps * 1e3
    # dominant kernel: k_gemm; algorithmic flops per launch on one device = 2 * (SNPs of the shard) * indiv * n
    flops_launch = 2.0 * snps_loc / n_shards * indiv * n
    avg_ms = total_ms.value / max(1, launches.value)
    achieved = flops_launch / (avg_ms * 1e-3) * 1e-12 if avg_ms > 0 else 0.0

    # N > 1: what every rank's dominant kernel took, and what the all-reduce of the indiv x n result costs on its own (untimed
    # extra pass), so that a scaling curve explains itself
    per_rank = None
    if dist.is_initialized():
        km = torch.tensor([avg_ms], dtype=torch.float64, device=device)
        allk = [torch.zeros_like(km) for _ in range(dist.get_world_size())]
        dist.all_gather(allk, km)
        flat = C_N.t()
        sync()
        t1 = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        sync()
        ar_ms = (time.perf_counter() - t1) / 10 * 1e3
        per_rank = {"avg_k_gemm_launch_ms": [round(float(x.item()), 3) for x in allk], "launches_per_step_per_rank": 2,
                    "allreduce_alone_ms": round(ar_ms, 3), "allreduce_bytes": int(8 * indiv * n),
                    "note": "the all-reduce of the 'N' result runs concurrently with the collective-free 'T' product of the same step"}
        step(); sync()   # C_N holds the product again (the extra all-reduces summed it up repeatedly)

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
               "int8_ops_per_s_P": round(2.0 * snps_loc / n_shards * indiv * n * 7 / (ms8.value / max(1, la8.value) * 1e-3) * 1e-15, 3),
               "max_colwise_rel_diff_vs_f64_engine": max(dN, dT)}
        C_N.copy_(C_N64); C_T.copy_(C_T64)
        del C_N64, C_T64

    # ABI end-to-end (reference harness utils/benchmark/benchmark.f90:192-209): the same two products with HOST
    # B and C through the plain reference symbol dgemm_compressed -- what a Julia / Fortran caller sees, PCIe included.
    # 1 warm-up + 10 repetitions, mean and min.  Never `value`.
    abi = None
    if world == 1 and not args.no_abi:
        import numpy as np
        hB_N, hB_T = np.asfortranarray(B_N.cpu().numpy()), np.asfortranarray(B_T.cpu().numpy())
        hC_N, hC_T = np.zeros((indiv, n), order="F"), np.zeros((snps, n), order="F")
        dg = mx.dgemm_compressed

        def abi_step():
            dg.dgemm_compressed_main(False, eng.obj, hB_N, snps, indiv, out=hC_N)
            dg.dgemm_compressed_main(True, eng.obj, hB_T, snps, indiv, out=hC_T)
        abi_step()
        ts = []
        for _ in range(10):
            t1 = time.perf_counter()
            abi_step()
            ts.append(time.perf_counter() - t1)
        same = bool(np.array_equal(hC_N, C_N.cpu().numpy()) and np.array_equal(hC_T, C_T.cpu().numpy()))
        abi = {"what": "dgemm_compressed 'N' + 'T' with host (pageable) B and C, PCIe transfers inside the call; 1 warm-up + 10 repetitions",
               "mean_GFLOPs": round(flops_step / (sum(ts) / len(ts)) * 1e-9, 1), "max_GFLOPs": round(flops_step / min(ts) * 1e-9, 1),
               "mean_ms_per_step": round(sum(ts) / len(ts) * 1e3, 3), "min_ms_per_step": round(min(ts) * 1e3, 3),
               "host_bytes_per_step": int(8 * 2 * (snps + indiv) * n), "bitwise_equal_to_device_resident_results": same}
        del hB_N, hB_T, hC_N, hC_T

    if rank == 0:
        out = {
            "metric": "effective GFLOP/s for dgemm_compressed (2-bit SNP x fp64)",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": args.gpus if inprocess else world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{snps} SNPs x {indiv} indiv, ncol={n}, dgemm_compressed 'N' + 'T' per step, "
                                   f"{'centred' if args.centered else 'uncentred'}, SNP-sharded over {args.gpus if inprocess else world} GPU(s)"
                                   + (" inside one process behind the C ABI (MIRACULIX_NUM_GPUS)" if inprocess else ""),
                       "snps": snps, "indiv": indiv, "ncol": n, "parallelism": f"snp-shard{args.gpus if inprocess else world}",
                       "genotypes": "p_s ~ U(0.1, 0.6), g ~ Binomial(2, p_s), no missings; B ~ N(0, 1)"},
            "check": {"adjoint_identity_max_rel_err": adj_err, "adjoint_tolerance": 1e-10},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "GB per launch, measured by this run (two rocprofv3 --pmc child passes of the same workload)", "traffic_detail": traffic_detail,
                         "algorithmic_bytes_per_launch_GB": round((snps_loc / n_shards * ((indiv + 3) // 4) + 8.0 * (snps_loc / n_shards + indiv) * n) / 1e9, 3),
                         "kernel": "k_gemm<8,8> (v_mfma_f64_4x4x4_4b_f64)", "launches": launches.value, "avg_launch_ms": round(avg_ms, 3)},
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
        if alt is not None:
            out["opt_in_engine"] = alt
        if abi is not None:
            out["abi_end_to_end"] = abi
        if keep_raw:   # CPU baseline + parity against the checker: rank 0 at N = 1 only
            def rows_of_plink_t(ii):
                return plink_t[torch.from_numpy(ii).to(device)].cpu().numpy()
            base, chk = cpu_baseline_and_check(torch, mx, plink, freq, snps, indiv, n, B_T, C_T, B_N, C_N, rows_of_plink_t)
            out["cpu_baseline"] = base
            out["check"].update(chk)
            if not (chk["gpu_T_rows_vs_cpu_library_max_rel_err"] <= 1e-11 and chk["gpu_N_64_sampled_rows_vs_dense_oracle_max_rel_err"] <= 1e-11):
                raise SystemExit(f"bench.py: GPU results differ from the checker ({chk}): no number reported")
        print(json.dumps(out), flush=True)
    eng.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
