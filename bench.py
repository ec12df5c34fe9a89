#!/usr/bin/env python3
"""bench.py -- headline benchmark: effective GFLOP/s of dgemm_compressed (2-bit SNP x fp64) on N MI355X.

python bench.py --gpus N --steps K --warmup W   (N > 1: launched by torch.distributed.run, one rank per GPU; without a launcher: SNP shards behind the C ABI)

Workload (BASELINE.json configs[1]): 1M SNPs x 50k individuals, n = 32, dgemm_compressed 'N' and 'T', uncentred, synthetic PLINK data generated on
the device.  One step = one 'N' multiply (+ the fp64 reduction of the indiv x n result when N > 1) and one 'T' multiply; flops per step =
2 * (2 * snps * indiv * n).  Inputs (packed genotypes, B, C) are resident in HBM when the timed region starts.  N > 1: the SNP dimension is sharded
(strong scaling, total work fixed).

Order of a run: setup_process / stage_headline -> ONE untimed step checked against the oracle (oracle_check: every N, every form; a violation ends the run
without a number) -> warm-up -> the timed region (run_timed: `value`) -> legs that never feed `value`: per_rank (launcher) or per_shard + the other
reduction + hub operands (in-process N > 1), opt-in engines, abi_end_to_end, cpu_baseline (+ the check against the CPU library), and at N = 1 the other
BASELINE configs under their own checkers (config5_cg_step, config4_shard, config3_crossprod, the full-extent legs).

Output: ONE compact JSON line on stdout (compact_line: <= 4 KB -- the contract's keys, roofline, cpu_baseline + cpu_baseline_port, check, who reduced,
one number per leg: c3_* crossproduct, c4_* config 4, c5_* CG step) and everything measured in bench_detail.json beside this file."""
import argparse
import ctypes
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL, CUDA-tensor sharing): keep the setting if the launcher's environment lost it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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
    utils/benchmark/benchmark.f90:185-209).
    `cpu_baseline` says what the REFERENCE's CPU path does on these cores (round 5): where oracle/_ref -- the reference's own library, compiled
    from /root/reference in the build container by oracle/Makefile.ref, git-ignored but carried along by gpurun pushes -- is present, it is the
    baseline (kind "reference") and the tracked port (oracle/oracle.c: oracle5_dgemm, bit-exact with that library on the pinned fixtures, but
    parallelised differently and about twice as fast) is reported beside it as `cpu_baseline_port`.  A clean checkout has no oracle/_ref: the port
    is then the baseline and its `kind` says that it is faster than the reference build it restates.
    The same leg is the in-run parity check against the checker: rows [0, SAMPLE_SNPS) of the GPU's 'T' result are compared with
    the CPU engine's own output on the sample (same packed rows, same B), and 64 sampled individuals of the GPU's 'N' result with
    the long-double dense oracle.  Everything under oracle/ is used here as baseline / checker only.
    Returns (cpu_baseline, cpu_baseline_port or None, check)."""
    import numpy as np
    from _util import Oracle, have_reference, run_reference
    sample = min(snps, 100_000)
    cores, how = physical_cores()
    plink = plink_dev[:sample].cpu().numpy()
    f = freq_dev[:sample].cpu().numpy()
    prob = dict(snps=sample, indiv=indiv, plink=np.ascontiguousarray(plink), plink_t=None, f=f)
    Bt = np.ascontiguousarray(B_T.t().cpu().numpy())                                   # n x indiv
    Bn = np.ascontiguousarray(B_N[:sample].t().cpu().numpy())                          # n x sample
    o = Oracle()
    flops = 2.0 * sample * indiv * n
    what = f"first {sample} SNPs of the bench matrix x {indiv} indiv, n={n}, one 'N' + one 'T' multiply, uncentred, best of 2"
    # the tracked port
    os.environ["OMP_NUM_THREADS"] = str(cores)
    times, Ct_cpu = {}, None
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
    port = {"value": round(2 * flops / (times[0] + times[1]) * 1e-9, 2), "unit": "GFLOP/s", "cores": cores, "cores_counted_as": how,
            "kind": "port (about 2x faster than the reference build it restates: 735 against 373 GFLOP/s on 16 cores, BENCH_r04; not tuned further)",
            "engine": "oracle/oracle.c oracle5_dgemm: 5-codes tables + lookup-add, OpenMP over the cores above (tracked source, built by __graft_entry__.build())",
            "sample": what + f" (N {times[0]:.3f}s, T {times[1]:.3f}s)"}
    base, extra = port, port          # both keys are always in the line: `cpu_baseline` = the reference's own library where it travelled, else the port; `cpu_baseline_port` = the port
    if have_reference():
        rt = {}
        _, rt[0] = run_reference(prob, 0, Bn, centered=False, variant=256, cores=cores, reps=2)
        Ct_ref, rt[1] = run_reference(prob, 1, Bt, centered=False, variant=256, cores=cores, reps=2)
        base = {"value": round(2 * flops / (rt[0] + rt[1]) * 1e-9, 2), "unit": "GFLOP/s", "cores": cores, "cores_counted_as": how, "kind": "reference",
                "engine": "oracle/_ref/libmiraculix_ref.so: the reference's own CPU library (5codes engine, AVX2 variant 256, OpenMP) compiled from /root/reference by "
                          "oracle/Makefile.ref; git-ignored build output that travelled with this push -- a clean checkout reports the port instead",
                "sample": what + f" (N {rt[0]:.3f}s, T {rt[1]:.3f}s)",
                "port_T_output_bitwise_equal_to_reference_build": bool(np.array_equal(Ct_ref[:, :sample], Ct_cpu[:, :sample]))}
    # parity: GPU 'T' rows of the sample against the CPU engine
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
    check = {"gpu_T_rows_vs_cpu_library_max_rel_err": err_t, "gpu_N_64_sampled_rows_vs_dense_oracle_max_rel_err": err_n, "checker_tolerance": 1e-11}
    return base, extra, check


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
    import collections, csv, glob, re, shutil, subprocess, tempfile
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
            r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True, timeout=600, env=dict(os.environ, TMPDIR=tmp, MXA_WARMUP="0"))   # (no warm-up products inside plink2compressed: the counters see the two launches of the workload only)
            src = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if r.returncode != 0 or not src:
                return None, {"skipped": f"rocprofv3 --pmc {ctr} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}"}
            agg = collections.OrderedDict()
            for row in csv.DictReader(open(src[0])):
                # the shipped instantiations only (MODE 2 / 3); the MODE 0 launch behind it is the range-guard fallback and returns at once
                if re.search(r"k_gemm<\d+, ?\d+, ?[23],", row["Kernel_Name"]) and row.get("Counter_Name", ctr) == ctr:
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


# ====================================================================================================== helpers shared by the legs
def extract_sample(torch, S, nsample=64, seed=1, seed_rows=None):
    """the packed rows a sampled check needs, copied to the host: nsample individuals (rows of the individual-major matrix) for 'N' and
    nsample SNPs (rows of the SNP-major matrix) for 'T'.  After this the raw device matrices may be released (the full-extent legs need
    the memory for the staged objects).  seed_rows: the SNP rows from a generator of their own (the launcher's ranks sample the same individuals
    and different SNP rows)."""
    import numpy as np
    dev, snps, indiv = S["dev"], S["snps"], S["indiv"]
    rng = np.random.default_rng(seed)
    ii = np.sort(rng.choice(indiv, min(nsample, indiv), replace=False))
    if seed_rows is not None:
        rng = np.random.default_rng(seed_rows)
    ss = np.sort(rng.choice(snps, min(nsample, snps), replace=False))
    f = S["f"].cpu().numpy()
    return dict(snps=snps, indiv=indiv, dev=dev, f=f, ii=ii, ss=ss,
                rows_t=np.ascontiguousarray(S["plink_t"][torch.from_numpy(ii).to(dev)].cpu().numpy()),      # nsample x ceil(snps/4)
                rows_s=np.ascontiguousarray(S["plink"][torch.from_numpy(ss).to(dev)].cpu().numpy()))       # nsample x ceil(indiv/4)


def check_sample(torch, sample, trans, Bdev, Cdev, cols, centered, row_offset=0, want_bound=False):
    """sampled rows of a result (individuals for 'N', SNPs for 'T'), columns `cols`, against the long-double dense oracle on the extracted
    packed rows.  Cdev may be a row block of the result starting at row_offset (per-shard results).  Returns max|C - ref| / max|ref|; with want_bound
    (err, max over the sampled ELEMENTS of |C - ref| / (4 K 2^-53 sum_k |z_ik| |b_kj|)) -- the hard element-wise bound of SURVEY.md 8(d), <= 1 passes;
    it protects output rows far below max|ref|, which the norm-wise tolerance does not.
    (Checker use of oracle/: tests and this file's parity legs only.)"""
    import numpy as np
    from _util import Oracle, elementwise_bound
    o = Oracle()
    dev, snps, indiv = sample["dev"], sample["snps"], sample["indiv"]
    Bs = np.ascontiguousarray(Bdev[:, cols].t().cpu().numpy())                        # len(cols) x k, row j = column cols[j]
    if not trans:
        ii, rows = sample["ii"], sample["rows_t"]
        sub_plink = o.transpose_2bit(rows, len(ii), snps)                              # snps x ceil(nsample/4)
        prob = dict(snps=snps, indiv=len(ii), plink=sub_plink, plink_t=rows, f=sample["f"])
        ref = o.dgemm_dense(0, prob, Bs, centered)                                     # len(cols) x nsample
        got = Cdev[torch.from_numpy(ii - row_offset).to(Cdev.device)][:, cols].t().cpu().numpy()
    else:
        ss, srows = sample["ss"], sample["rows_s"]
        prob = dict(snps=len(ss), indiv=indiv, plink=srows, plink_t=None, f=np.ascontiguousarray(sample["f"][ss]))
        ref = o.dgemm_dense(1, prob, Bs, centered)
        got = Cdev[torch.from_numpy(ss - row_offset).to(Cdev.device)][:, cols].t().cpu().numpy()
    err = float(np.abs(got - ref).max() / np.abs(ref).max())
    if not want_bound:
        return err
    bound = elementwise_bound(o, trans, prob, Bs, centered)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(bound > 0, np.abs(got - ref) / bound, np.where(got == ref, 0.0, np.inf))
    return err, float(ratio.max())


def sampled_rows_vs_oracle(torch, S, trans, Bdev, Cdev, cols, centered, nsample=64, seed=1, want_bound=False):
    """nsample rows of a result (individuals for 'N', SNPs for 'T'), columns `cols`, against the long-double dense oracle on the
    extracted rows of the packed matrix.  S: dict(dev, snps, indiv, plink (SNP-major, device), plink_t (individual-major, device),
    f (device)).  Returns max|C - ref| / max|ref|."""
    import numpy as np
    dev, snps, indiv = S["dev"], S["snps"], S["indiv"]
    rng = np.random.default_rng(seed)
    sample = dict(snps=snps, indiv=indiv, dev=dev, f=S["f"].cpu().numpy())
    if not trans:
        sample["ii"] = np.sort(rng.choice(indiv, nsample, replace=False))
        sample["rows_t"] = np.ascontiguousarray(S["plink_t"][torch.from_numpy(sample["ii"]).to(dev)].cpu().numpy())
    else:
        sample["ss"] = np.sort(rng.choice(snps, nsample, replace=False))
        sample["rows_s"] = np.ascontiguousarray(S["plink"][torch.from_numpy(sample["ss"]).to(dev)].cpu().numpy())
    return check_sample(torch, sample, trans, Bdev, Cdev, cols, centered, want_bound=want_bound)


def stage_object(torch, mx, dev, snps, indiv, n, seed, centered):
    """synthetic genotypes generated on the device, transposed and counted there, staged through plink2compressed"""
    plink = synth_genotypes_device(torch, snps, indiv, seed, dev)                       # SNP-major
    plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, snps, indiv)
    f = mx.read_plink.calc_freq(plink, snps, indiv)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
    obj = dg.init_compressed(plink, plink_t, snps, indiv, f, n)
    return dict(torch=torch, mx=mx, dev=dev, plink=plink, plink_t=plink_t, f=f, obj=obj, dg=dg, snps=snps, indiv=indiv, n=n)


def kernel_profile(L):
    la, ms = ctypes.c_int(0), ctypes.c_double(0.0)
    L.mxa_profile_get(ctypes.byref(la), ctypes.byref(ms))
    return la.value, ms.value


def timed(fn, sync, reps):
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


# ====================================================================================================== the other configs (N = 1, rank 0, never `value`)
def config5_cg_step_leg(torch, mx, L, dev, snps=250_000, indiv=100_000, reps=20):
    """BASELINE config 5 at its per-GPU shard (2M SNPs x 100k over 8 GPUs): one CG step G v = Zc (Zc^T v), n = 1, centred, vectors
    resident in HBM (reference loop: examples/iterative_solver/grm_solve_cg.jl:74-84).  HBM-bound: a step reads both packed
    orientations once."""
    S = stage_object(torch, mx, dev, snps, indiv, 1, 45, centered=True)
    dg = S["dg"]
    try:
        g = torch.Generator(device=dev); g.manual_seed(7)
        v = torch.randn((1, indiv), dtype=torch.float64, device=dev, generator=g).t()
        out = torch.zeros((1, indiv), dtype=torch.float64, device=dev).t()
        sync = torch.cuda.synchronize
        dg.gram_matvec(S["obj"], v, snps, indiv, out=out)
        t_step = timed(lambda: dg.gram_matvec(S["obj"], v, snps, indiv, out=out), sync, reps)
        # the loop as a device-resident caller runs it: no host wait between the steps (mxa_gram_matvec_device, sync = 0)
        t_async = timed(lambda: dg.gram_matvec(S["obj"], v, snps, indiv, out=out, sync=False), sync, reps)
        L.mxa_profile_reset()
        for _ in range(5):                                  # (one launch is a noisy sample: the mean of five)
            T = dg.dgemm_compressed_main(True, S["obj"], v, snps, indiv)
        la_t, ms_t = kernel_profile(L)
        path = dg.last_path()
        L.mxa_profile_reset()
        for _ in range(5):
            N = dg.dgemm_compressed_main(False, S["obj"], T, snps, indiv)
        la_n, ms_n = kernel_profile(L)
        err_t = sampled_rows_vs_oracle(torch, S, 1, v, T, [0], 1, nsample=32)
        err_n = sampled_rows_vs_oracle(torch, S, 0, T, N, [0], 1, nsample=32)
        same = bool(torch.equal(out, N))
        bytes_step = 2.0 * snps * ((indiv + 3) // 4)      # the packed matrix is read twice per step (one copy twice, or each of two copies once)
        tbs = bytes_step / t_step * 1e-12
        # the same step on an object that stores BOTH packed copies (MXA_SINGLE_ORIENTATION=0, the opt-in since round 5): 'N' on the plain int8 kernel
        old_so = os.environ.get("MXA_SINGLE_ORIENTATION")
        os.environ["MXA_SINGLE_ORIENTATION"] = "0"
        try:
            obj2 = dg.init_compressed(S["plink"], S["plink_t"], snps, indiv, S["f"], 1)
        finally:
            if old_so is None:
                os.environ.pop("MXA_SINGLE_ORIENTATION", None)
            else:
                os.environ["MXA_SINGLE_ORIENTATION"] = old_so
        try:
            out2 = torch.zeros((1, indiv), dtype=torch.float64, device=dev).t()
            dg.gram_matvec(obj2, v, snps, indiv, out=out2)
            t_two = timed(lambda: dg.gram_matvec(obj2, v, snps, indiv, out=out2), sync, reps)
            two_vs_one = float((out2 - out).abs().max() / out.abs().max())
        finally:
            dg.free_compressed(obj2)
        return {"workload": f"{snps} SNPs x {indiv} indiv (per-GPU shard of config 5), n=1, centred, one mxa_gram_matvec = 'T' + 'N'",
                "object": "one packed copy (default): 'T' on k_gemm_i8, 'N' on k_gemm_i8_tn, both reading the SNP-major copy",
                "ms_per_cg_step_two_copies": round(t_two * 1e3, 4), "two_copies_vs_one_copy_max_rel_err": two_vs_one,
                "ms_per_cg_step": round(t_step * 1e3, 4), "ms_per_cg_step_back_to_back_no_host_wait": round(t_async * 1e3, 4), "kernel_path": path, "dominant_kernel_ms": {"T": round(ms_t / max(1, la_t), 4), "N": round(ms_n / max(1, la_n), 4)},
                "algorithmic_TB_per_s": round(tbs, 3), "frac_of_8_TBs_spec": round(tbs / 8.0, 4), "frac_of_7.0_TBs_read_ceiling": round(tbs / 7.0, 4),   # 7.0: what one MI355X reads with non-temporal loads (tools/hbm_read_probe.hip; 6.3-6.5 with the default policy)
                "check": {"T_32_sampled_rows_vs_dense_oracle_max_rel_err": err_t, "N_32_sampled_rows_vs_dense_oracle_max_rel_err": err_n,
                          "gram_matvec_bitwise_equals_T_then_N": same, "checker_tolerance": 1e-11}}
    finally:
        dg.free_compressed(S["obj"])
        S.clear()
        torch.cuda.empty_cache()


def config4_shard_leg(torch, mx, L, dev, snps=625_000, indiv=200_000, n=128, reps=3):
    """BASELINE config 4 at its per-GPU shard (5M SNPs x 200k over 8 GPUs): ncol = 128, allele-frequency centred, 'N' + 'T'"""
    S = stage_object(torch, mx, dev, snps, indiv, n, 44, centered=True)
    dg = S["dg"]
    try:
        g = torch.Generator(device=dev); g.manual_seed(3)
        Y = torch.randn((n, snps), dtype=torch.float64, device=dev, generator=g).t()
        X = torch.randn((n, indiv), dtype=torch.float64, device=dev, generator=g).t()
        CN = torch.zeros((n, indiv), dtype=torch.float64, device=dev).t()
        CT = torch.zeros((n, snps), dtype=torch.float64, device=dev).t()
        sync = torch.cuda.synchronize
        res = {}
        for name, trans, B, C in (("N", False, Y, CN), ("T", True, X, CT)):
            dg.dgemm_compressed_main(trans, S["obj"], B, snps, indiv, out=C)
            L.mxa_profile_reset()
            t = timed(lambda: dg.dgemm_compressed_main(trans, S["obj"], B, snps, indiv, out=C), sync, reps)
            la, ms = kernel_profile(L)
            flops = 2.0 * snps * indiv * n
            res[name] = {"ms_per_call": round(t * 1e3, 3), "k_gemm_ms": round(ms / max(1, la), 3), "TFLOPs_call": round(flops / t * 1e-12, 2),
                         "TFLOPs_kernel": round(flops / (ms / max(1, la) * 1e-3) * 1e-12, 2), "frac_of_fp64_mfma_peak_kernel": round(flops / (ms / max(1, la) * 1e-3) * 1e-12 / FP64_MFMA_PEAK_TFLOPS, 4)}
        cols = [0, 31, 32, 127]
        err_n, br_n = sampled_rows_vs_oracle(torch, S, 0, Y, CN, cols, 1, nsample=16, want_bound=True)
        err_t, br_t = sampled_rows_vs_oracle(torch, S, 1, X, CT, cols, 1, nsample=16, want_bound=True)
        lhs, rhs = (X * CN).sum(dim=0), (CT * Y).sum(dim=0)
        adj = float(((lhs - rhs).abs() / (X.abs() * CN.abs()).sum(dim=0)).max())
        return {"workload": f"{snps} SNPs x {indiv} indiv (per-GPU shard of config 4), ncol={n}, centred", "N": res["N"], "T": res["T"],
                "check": {"N_16_sampled_rows_vs_dense_oracle_max_rel_err": err_n, "T_16_sampled_rows_vs_dense_oracle_max_rel_err": err_t,
                          "N_max_err_over_elementwise_bound": br_n, "T_max_err_over_elementwise_bound": br_t,
                          "centred_adjoint_identity_max_rel_err": adj, "checker_tolerance": 1e-11}}
    finally:
        dg.free_compressed(S["obj"])
        S.clear()
        torch.cuda.empty_cache()


def config3_crossprod_leg(torch, mx, L, dev, snps=500_000, indiv=100_000):
    """BASELINE config 3: GRM crossproduct M = X X^T, 500k SNPs x 100k individuals, device-resident 80 GB fp64 result, with both engines
    (FP4 MFMA = default while exact, int8 MFMA = the path BASELINE names).  Ops: 2 * snps * indiv^2 in the full-matrix count
    (SURVEY.md 8d); executed = the upper-triangular 256 x 256 tiles only."""
    import numpy as np
    from _util import Oracle
    X = synth_genotypes_device(torch, indiv, snps, 46, dev, p_along="cols")              # individual-major, 12.5 GB
    M = torch.empty((indiv, indiv), dtype=torch.float64, device=dev)                     # 80 GB
    o = Oracle()
    nb = (indiv + 255) // 256
    tiles = [(0, 0), (nb - 1, nb - 1), (0, nb - 1), (nb // 3, nb // 2)]
    refs = []
    for ti, tj in tiles:
        ri = np.arange(ti * 256, min(indiv, ti * 256 + 256)); rj = np.arange(tj * 256, min(indiv, tj * 256 + 256))
        rows = np.concatenate([ri, rj]) if ti != tj else ri
        sub = np.ascontiguousarray(X[torch.from_numpy(rows).to(dev)].cpu().numpy())
        ref = o.crossprod_i32(sub, snps, True).astype(np.float64)
        refs.append((ri, rj, ref if ti == tj else ref[: len(ri), len(ri):]))
    full_ops = 2.0 * snps * float(indiv) ** 2
    exec_ops = 2.0 * snps * 256.0 * 256.0 * (nb * (nb + 1) // 2)
    out = {"workload": f"GRM crossproduct, {snps} SNPs x {indiv} indiv, device-resident fp64 result ({indiv * indiv * 8 / 1e9:.0f} GB)"}
    old = os.environ.get("MXA_XPROD_ENGINE")
    try:
        for eng, peak in (("f4", 10.0), ("i8", 5.0)):
            if eng == "i8":
                os.environ["MXA_XPROD_ENGINE"] = "i8"
            else:
                os.environ.pop("MXA_XPROD_ENGINE", None)
            M.fill_(-1.0)
            torch.cuda.synchronize()
            L.mxa_profile_reset()
            t0 = time.perf_counter()
            mx.crossproduct.snp_crossprod(X, snps, indiv, is_snpmajor=False, is_plink_format=True, out=M)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            la, ms = kernel_profile(L)
            exact = True
            for ri, rj, ref in refs:
                exact &= bool(np.array_equal(M[ri[0]:ri[-1] + 1, rj[0]:rj[-1] + 1].cpu().numpy(), ref))
                exact &= bool(np.array_equal(M[rj[0]:rj[-1] + 1, ri[0]:ri[-1] + 1].cpu().numpy(), ref.T))
            sym = bool(torch.equal(M[:2048, :], M[:, :2048].t())) and float(M[-4096:].min()) >= 0.0
            k_ms = ms / max(1, la)
            # kernel names in a rocprofv3 trace: k_crossprod_gang<false, 0> (FP4) / <true, 0> (int8) -- the gang-synchronised persistent form of the tile pipeline
            out["k_crossprod_f4 (FP4 MFMA, default)" if eng == "f4" else "k_crossprod_i8 (int8 MFMA)"] = {
                "kernel_ms": round(k_ms, 2), "call_ms_incl_staging": round(wall * 1e3, 1), "Pop_s_full_matrix_count": round(full_ops / (k_ms * 1e-3) * 1e-15, 3),
                "Pop_s_executed": round(exec_ops / (k_ms * 1e-3) * 1e-15, 3), "dense_peak_Pop_s": peak, "frac_of_peak_executed": round(exec_ops / (k_ms * 1e-3) * 1e-15 / peak, 4),
                "check": {"four_256x256_tiles_and_mirrors_bit_exact_vs_int32_oracle": exact, "first_panel_symmetric_and_all_written": sym}}
    finally:
        if old is None:
            os.environ.pop("MXA_XPROD_ENGINE", None)
        else:
            os.environ["MXA_XPROD_ENGINE"] = old
        del X, M
        torch.cuda.empty_cache()
    return out


def _stage_full(torch, mx, dev, snps, indiv, n, seed, shards):
    """raw synthetic matrices on the device + one object over `shards` SNP blocks behind the plain symbols (MIRACULIX_NUM_GPUS; on a
    one-GPU box the shards share the device: the real SNP partition, worker threads, per-shard streams and the fixed-order reduction)"""
    plink = synth_genotypes_device(torch, snps, indiv, seed, dev)
    plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, snps, indiv)
    f = mx.read_plink.calc_freq(plink, snps, indiv)
    return dict(torch=torch, mx=mx, dev=dev, plink=plink, plink_t=plink_t, f=f, dg=mx.dgemm_compressed, snps=snps, indiv=indiv, n=n)


def _make_object(S, shards, centered, single=False):
    dg = S["dg"]
    dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
    old = {k: os.environ.get(k) for k in ("MIRACULIX_NUM_GPUS", "MXA_SINGLE_ORIENTATION")}
    try:
        if shards > 1:
            os.environ["MIRACULIX_NUM_GPUS"] = str(shards)
        else:
            os.environ.pop("MIRACULIX_NUM_GPUS", None)
        os.environ["MXA_SINGLE_ORIENTATION"] = "1" if single else "0"      # one packed copy (SNP-major) serves both products; plink_transposed is not read
        obj = dg.init_compressed(S["plink"], None if single else S["plink_t"], S["snps"], S["indiv"], S["f"], S["n"])
    finally:
        for k, v in old.items():                                            # a user-set value survives the leg
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert dg.num_shards(obj) == shards
    return obj


def config5_full_leg(torch, mx, L, dev, snps=2_000_000, indiv=100_000, shards=8, iters=20):
    """BASELINE config 5 at its FULL extent on one GPU: 2M SNPs x 100k individuals (2 x 50 GB packed), the GBLUP / CG loop of the reference's
    examples/iterative_solver/grm_solve_cg.jl:74-84,108-134 -- `iters` iterations of (Zc Zc^T + lambda I) x = b, one mxa_gram_matvec each --
    (i) on the object cut into 8 SNP shards behind the plain symbols (MIRACULIX_NUM_GPUS=8: the partition, staging, per-shard products and the
    fixed-order reduction of the 8-GPU run, here with all shards on one device), (ii) on one plain object and (iii) on one SINGLE-ORIENTATION object
    (MXA_SINGLE_ORIENTATION=1: only the SNP-major copy, 50 GB instead of 100; 'N' by the transposed-operand kernel k_gemm_i8_tn).  Checks: sampled rows of both
    products against the long-double dense oracle, the residual the loop reports against a separately computed one, bitwise repeatability,
    and sharded == single object bit for bit on an integer-valued vector (uncentred: every partial sum is an exact integer)."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    from grm_solve_cg import cg
    S = _stage_full(torch, mx, dev, snps, indiv, 1, 48, shards)
    dg = S["dg"]
    sample = extract_sample(torch, S, nsample=32, seed=5)
    res = {"workload": f"{snps} SNPs x {indiv} indiv (config 5 at full extent, {2 * snps * ((indiv + 3) // 4) / 1e9:.0f} GB packed in two orientations), n=1, centred, "
                       f"{iters} CG iterations, one mxa_gram_matvec each"}
    g = torch.Generator(device=dev); g.manual_seed(11)
    b = torch.randn((1, indiv), dtype=torch.float64, device=dev, generator=g).t()
    vint = torch.randint(-3, 4, (1, indiv), device=dev, generator=g).to(torch.float64).t()
    lam = float(snps)
    sync = torch.cuda.synchronize
    keep = {}
    for name, nsh in ((f"{shards}_virtual_shards", shards), ("one_object", 1), ("one_object_single_orientation", 1)):
        single = name.endswith("single_orientation")
        if single:                       # the individual-major raw matrix is not needed for this object: release it before staging
            S["plink_t"] = None
            torch.cuda.empty_cache()
        free0 = torch.cuda.mem_get_info()[0]
        obj = _make_object(S, nsh, centered=True, single=single)
        held_gb = (free0 - torch.cuda.mem_get_info()[0]) / 1e9
        if single:                       # the raw matrices are no longer needed: the samples are on the host
            S["plink"] = None
            torch.cuda.empty_cache()
        try:
            class Op:
                def gram(self, v):
                    return dg.gram_matvec(obj, v, snps, indiv)
            x0 = torch.zeros_like(b)
            cg(Op(), b, x0, lam, max_iter=2, conv_crit=0.0, verbose=False)            # warm-up (workspace growth, first launches)
            sync(); t0 = time.perf_counter()
            crit = 1e-30                                                                 # never met: the loop runs all its iterations (config 5 asks for >= 20)
            x, resid, it = cg(Op(), b, x0, lam, max_iter=iters, conv_crit=crit, verbose=False)
            sync(); dt = time.perf_counter() - t0
            v = x / torch.linalg.vector_norm(x)
            out = torch.zeros_like(v)
            dg.gram_matvec(obj, v, snps, indiv, out=out)
            t_step = timed(lambda: dg.gram_matvec(obj, v, snps, indiv, out=out), sync, 5)
            # the two products of the step on their own, under the checker
            T = dg.dgemm_compressed_main(True, obj, v, snps, indiv)
            N = dg.dgemm_compressed_main(False, obj, T, snps, indiv)
            err_t = check_sample(torch, sample, 1, v, T, [0], 1)
            err_n = check_sample(torch, sample, 0, T, N, [0], 1)
            gram_vs_pair = float((out - N).abs().max() / N.abs().max())
            Ax = dg.gram_matvec(obj, x, snps, indiv) + lam * x
            bnorm = float(torch.linalg.vector_norm(b))
            res_gap = abs(float(torch.linalg.vector_norm(b - Ax)) - resid) / bnorm
            x2, resid2, _ = cg(Op(), b, x0, lam, max_iter=iters, conv_crit=crit, verbose=False)
            repeat = bool(torch.equal(x, x2)) and resid == resid2
            # integer-valued vector, uncentred: exact integers throughout -> independent of the partition
            dg.set_options(use_gpu=True, not_center=True, verbose=0)
            Ti = dg.dgemm_compressed_main(True, obj, vint, snps, indiv)
            Ni = dg.dgemm_compressed_main(False, obj, Ti, snps, indiv)
            Gi = dg.gram_matvec(obj, vint, snps, indiv)
            dg.set_options(use_gpu=True, not_center=False, verbose=0)
            keep[name] = (Ti, Ni, Gi)
            bytes_step = 2.0 * snps * ((indiv + 3) // 4)
            res[name] = {"device_memory_held_by_the_object_GB": round(held_gb, 1),
                         "ms_per_cg_iteration_incl_vector_ops": round(dt / (it + 1) * 1e3, 3), "ms_per_gram_matvec": round(t_step * 1e3, 3),
                         "algorithmic_TB_per_s": round(bytes_step / t_step * 1e-12, 3), "frac_of_8_TBs_spec": round(bytes_step / t_step * 1e-12 / 8.0, 4),
                         "cg_iterations": it, "cg_residual": resid, "rhs_norm": bnorm,
                         "check": {"T_32_sampled_rows_vs_dense_oracle_max_rel_err": err_t, "N_32_sampled_rows_vs_dense_oracle_max_rel_err": err_n,
                                   "gram_matvec_vs_T_then_N_max_rel_err": gram_vs_pair,
                                   "residual_recomputed_minus_reported_over_rhs_norm": res_gap, "cg_residual_consistent_ok": bool(res_gap <= 1e-6 + 10 * resid / bnorm),
                                   "cg_converging_ok": bool(resid < 1e-3 * bnorm), "cg_bitwise_repeatable": repeat, "checker_tolerance": 1e-11}}
        finally:
            dg.free_compressed(obj)
            torch.cuda.empty_cache()
    a, c, d = keep[f"{shards}_virtual_shards"], keep["one_object"], keep["one_object_single_orientation"]
    res["check"] = {"sharded_equals_one_object_bitwise_on_integer_vector": bool(torch.equal(a[0], c[0]) and torch.equal(a[1], c[1]) and torch.equal(a[2], c[2])),
                    "single_orientation_equals_two_copies_bitwise_on_integer_vector": bool(torch.equal(d[0], c[0]) and torch.equal(d[1], c[1]) and torch.equal(d[2], c[2])),
                    "integer_gram_equals_T_then_N_bitwise": bool(torch.equal(c[1], c[2]))}
    S.clear(); keep.clear()
    torch.cuda.empty_cache()
    return res


def config4_full_extent_leg(torch, mx, L, dev, snps=5_000_000, indiv=25_000, n=128, shards=8, reps=2):
    """BASELINE config 4's FULL 5M-SNP extent (at a reduced individual count so that both orientations, 2 x 31 GB, fit one GPU), ncol = 128,
    allele-frequency centred, cut into 8 SNP shards behind the plain symbols (MIRACULIX_NUM_GPUS=8: 625 000 SNPs per shard, as on the 8-GPU
    node; here all shards on one device), 'N' (with the fixed-order reduction of the indiv x 128 partials) and 'T'.  Checks: sampled rows against
    the centred long-double oracle, the centred adjoint identity, bitwise repeatability, and sharded == single object bit for bit on an
    integer-valued B (uncentred)."""
    S = _stage_full(torch, mx, dev, snps, indiv, n, 49, shards)
    dg = S["dg"]
    sample = extract_sample(torch, S, nsample=16, seed=6)
    g = torch.Generator(device=dev); g.manual_seed(3)
    Y = torch.randn((n, snps), dtype=torch.float64, device=dev, generator=g).t()
    X = torch.randn((n, indiv), dtype=torch.float64, device=dev, generator=g).t()
    Yi = torch.randint(-2, 3, (8, snps), device=dev, generator=g).to(torch.float64).t()      # 8 integer columns for the partition-independence check
    Xi = torch.randint(-2, 3, (8, indiv), device=dev, generator=g).to(torch.float64).t()
    CN = torch.zeros((n, indiv), dtype=torch.float64, device=dev).t()
    CT = torch.zeros((n, snps), dtype=torch.float64, device=dev).t()
    sync = torch.cuda.synchronize
    res = {"workload": f"{snps} SNPs x {indiv} indiv (config 4's full SNP extent at reduced indiv, {2 * snps * ((indiv + 3) // 4) / 1e9:.0f} GB packed), ncol={n}, centred, "
                       f"SNP-sharded into {shards} blocks behind dgemm_compressed"}
    keep = {}
    cols = [0, 31, 32, 127] if n >= 128 else [0, n - 1]
    flops = 2.0 * snps * indiv * n
    for name, nsh in ((f"{shards}_virtual_shards", shards), ("one_object", 1)):
        obj = _make_object(S, nsh, centered=True)
        if name == "one_object":
            S["plink"] = S["plink_t"] = None
            torch.cuda.empty_cache()
        try:
            r = {}
            for tname, trans, B, C in (("N", False, Y, CN), ("T", True, X, CT)):
                dg.dgemm_compressed_main(trans, obj, B, snps, indiv, out=C)
                t = timed(lambda: dg.dgemm_compressed_main(trans, obj, B, snps, indiv, out=C), sync, reps)
                r[tname] = {"ms_per_call": round(t * 1e3, 2), "TFLOPs_call": round(flops / t * 1e-12, 2), "frac_of_fp64_mfma_peak_call": round(flops / t * 1e-12 / FP64_MFMA_PEAK_TFLOPS, 4)}
            err_n = check_sample(torch, sample, 0, Y, CN, cols, 1)
            err_t = check_sample(torch, sample, 1, X, CT, cols, 1)
            lhs, rhs = (X * CN).sum(dim=0), (CT * Y).sum(dim=0)
            adj = float(((lhs - rhs).abs() / (X.abs() * CN.abs()).sum(dim=0)).max())
            CN2 = dg.dgemm_compressed_main(False, obj, Y, snps, indiv)
            rep = bool(torch.equal(CN2, CN))
            del CN2
            dg.set_options(use_gpu=True, not_center=True, verbose=0)
            keep[name] = (dg.dgemm_compressed_main(False, obj, Yi, snps, indiv), dg.dgemm_compressed_main(True, obj, Xi, snps, indiv))
            dg.set_options(use_gpu=True, not_center=False, verbose=0)
            r["check"] = {"N_16_sampled_rows_vs_dense_oracle_max_rel_err": err_n, "T_16_sampled_rows_vs_dense_oracle_max_rel_err": err_t,
                          "centred_adjoint_identity_max_rel_err": adj, "N_bitwise_repeatable": rep, "checker_tolerance": 1e-11}
            res[name] = r
        finally:
            dg.free_compressed(obj)
            torch.cuda.empty_cache()
    a, c = keep[f"{shards}_virtual_shards"], keep["one_object"]
    res["check"] = {"sharded_equals_one_object_bitwise_on_integer_B": bool(torch.equal(a[0], c[0]) and torch.equal(a[1], c[1]))}
    S.clear(); keep.clear()
    del Y, X, CN, CT, Yi, Xi
    torch.cuda.empty_cache()
    return res


def config4_full_one_copy_leg(torch, mx, L, dev, snps=5_000_000, indiv=200_000, n=128, block_snps=50_000, seed=50, log=None):
    """BASELINE config 4 as ONE product at its stated size on ONE MI355X: 5 000 000 SNPs x 200 000 individuals, ncol = 128, allele-frequency centred.
    250 GB packed: the object holds ONE packed copy (SNP-major; 'T' in the plain form, 'N' in the transposed-operand form of k_gemm) and is staged
    INCREMENTALLY (mxa_plink2compressed_begin / _rows / _end) from SNP blocks generated on the device, so nothing but the object and one block is ever
    resident -- where the reference's pre-flight gives up (src/cuda/dgemm_compressed_cuda.cu:93-100; it needs both copies AND the host matrices).
    The split-K partial sums of these products (18 x 5.1 GB for 'T', 407 x 0.2 GB for 'N') do not fit beside the matrix: the K splits run in groups
    with the running sum kept in C (bit-identical to one pass: tests/test_grouped_and_incremental_gpu.py).
    Checks: sampled rows of both products against the centred long-double oracle on the extracted packed rows (<= 1e-11), the centred adjoint identity,
    bitwise repeatability of 'N'.  If the byte budget does not fit this device the budget is reported and the largest SNP count that fits is run."""
    import numpy as np
    dg = mx.dgemm_compressed
    say = log or (lambda *a: None)
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    torch.cuda.empty_cache()
    free0, total = torch.cuda.mem_get_info()
    rb = (indiv + 3) // 4

    def budget(s):
        packed = (s + 512) * ((indiv + 128) // 4)
        big = 8 * s * n                                   # one snps x n fp64 array
        return {"packed_one_copy_GB": packed / 1e9, "B_of_N_and_C_of_T_GB": 2 * big / 1e9, "B_in_fragment_order_GB": big / 1e9,
                "partial_sums_one_group_GB": max(big, min(16 * 2 ** 30, 18 * big)) / 1e9,   # the library's soft cap: 16 GiB, at least one K split
                "small_operands_and_frequencies_GB": (4 * 8 * indiv * n + 8 * s) / 1e9,
                "generator_block_and_margin_GB": (block_snps * rb * 2 + (4 << 30)) / 1e9}
    want = snps
    while sum(budget(snps).values()) * 1e9 > free0 and snps > 500_000:
        snps -= 250_000
    bud = budget(snps)
    res = {"workload": f"{snps} SNPs x {indiv} indiv, ncol={n}, centred: BASELINE config 4 as ONE product on one device, one packed copy staged incrementally",
           "device_memory_GB": {"total": round(total / 1e9, 1), "free_at_start": round(free0 / 1e9, 1)},
           "byte_budget_GB": {k: round(v, 2) for k, v in bud.items()} | {"sum": round(sum(bud.values()), 1)},
           "snps_requested": want, "snps_run": snps}
    rng = np.random.default_rng(seed)
    nsample = 16
    ii = np.sort(rng.choice(indiv, nsample, replace=False))
    ss = np.sort(rng.choice(snps, nsample, replace=False))
    rows_s = np.zeros((nsample, rb), dtype=np.uint8)
    rows_t = np.zeros((nsample, (snps + 3) // 4), dtype=np.uint8)
    ii_dev = torch.from_numpy(ii).to(dev)
    w4 = torch.tensor([1, 4, 16, 64], dtype=torch.uint8, device=dev)
    t0 = time.perf_counter()
    obj = dg.init_compressed_begin(snps, indiv, n)
    try:
        t_gen = t_app = 0.0
        for bi, b0 in enumerate(range(0, snps, block_snps)):
            nb = min(block_snps, snps - b0)
            ta = time.perf_counter()
            blk = synth_genotypes_device(torch, nb, indiv, seed + 1 + bi, dev)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            dg.append_rows(obj, blk, b0)                                             # recoded into the tiled layout, frequencies counted on the device
            t_app += time.perf_counter() - tb
            t_gen += tb - ta
            for q in np.nonzero((ss >= b0) & (ss < b0 + nb))[0]:
                rows_s[q] = blk[int(ss[q] - b0)].cpu().numpy()
            codes = (blk[:, ii_dev // 4] >> (2 * (ii_dev % 4)).to(torch.uint8)) & 3   # nb x nsample PLINK codes of the sampled individuals
            if nb % 4:
                codes = torch.nn.functional.pad(codes, (0, 0, 0, 4 - nb % 4))
            packed = (codes.view(-1, 4, nsample) * w4[None, :, None]).sum(dim=1, dtype=torch.uint8)    # 4 SNPs per byte
            rows_t[:, b0 // 4: b0 // 4 + packed.shape[0]] = packed.t().cpu().numpy()
            del blk, codes, packed
            if bi % 10 == 9:
                say(f"config4_full_one_copy: staged {b0 + nb} of {snps} SNPs ({time.perf_counter() - t0:.0f} s)")
        f = dg.init_compressed_end(obj, snps)
        torch.cuda.empty_cache()
        held = (free0 - torch.cuda.mem_get_info()[0]) / 1e9
        res["staging"] = {"seconds_total": round(time.perf_counter() - t0, 1), "seconds_generating_blocks_torch": round(t_gen, 1), "seconds_mxa_plink2compressed_rows": round(t_app, 1),
                          "device_memory_held_by_the_object_GB": round(held, 1), "single_orientation": int(L.mxa_single_orientation(obj))}
        say(f"config4_full_one_copy: object holds {held:.1f} GB")
        sample = dict(snps=snps, indiv=indiv, dev=dev, f=f, ii=ii, ss=ss, rows_t=rows_t, rows_s=rows_s)
        g = torch.Generator(device=dev); g.manual_seed(3)
        Y = torch.randn((n, snps), dtype=torch.float64, device=dev, generator=g).t()
        X = torch.randn((n, indiv), dtype=torch.float64, device=dev, generator=g).t()
        CN = torch.zeros((n, indiv), dtype=torch.float64, device=dev).t()
        CT = torch.zeros((n, snps), dtype=torch.float64, device=dev).t()
        sync = torch.cuda.synchronize
        flops = 2.0 * snps * indiv * n
        sp = ctypes.c_int(0)
        for tname, trans, B, C in (("N", False, Y, CN), ("T", True, X, CT)):
            dg.dgemm_compressed_main(trans, obj, B, snps, indiv, out=C)             # first call: workspace growth, clock ramp
            if tname == "N":
                CN1 = CN.clone()
            L.mxa_profile_reset()
            t = timed(lambda: dg.dgemm_compressed_main(trans, obj, B, snps, indiv, out=C), sync, 1)
            la, ms = kernel_profile(L)
            L.mxa_last_geometry(None, None, None, ctypes.byref(sp), None, None)
            res[tname] = {"ms_per_call": round(t * 1e3, 1), "TFLOPs_call": round(flops / t * 1e-12, 2), "frac_of_fp64_mfma_peak_call": round(flops / t * 1e-12 / FP64_MFMA_PEAK_TFLOPS, 4),
                          "k_gemm_and_group_finishes_ms": round(ms / max(1, la), 1), "k_splits": sp.value, "partial_sum_workspace_GB": round(L.mxa_partial_capacity(obj) * 8 / 1e9, 2)}
            say(f"config4_full_one_copy: {tname} {t * 1e3:.0f} ms = {flops / t * 1e-12:.1f} TFLOP/s")
        rep = bool(torch.equal(CN1, CN))
        del CN1
        cols = [0, 31, 32, 127] if n >= 128 else [0, n - 1]
        err_n, br_n = check_sample(torch, sample, 0, Y, CN, cols, 1, want_bound=True)
        err_t, br_t = check_sample(torch, sample, 1, X, CT, cols, 1, want_bound=True)
        lhs = (X * CN).sum(dim=0)
        den = (X.abs() * CN.abs()).sum(dim=0)
        rhs = torch.stack([(CT[:, j] * Y[:, j]).sum() for j in range(n)])             # column by column: a 5 GB temporary would not fit
        adj = float(((lhs - rhs).abs() / den).max())
        res["held_while_multiplying_GB"] = round((free0 - torch.cuda.mem_get_info()[0]) / 1e9, 1)
        res["check"] = {"N_16_sampled_rows_vs_dense_oracle_max_rel_err": err_n, "T_16_sampled_rows_vs_dense_oracle_max_rel_err": err_t,
                        "N_max_err_over_elementwise_bound": br_n, "T_max_err_over_elementwise_bound": br_t,
                        "centred_adjoint_identity_max_rel_err": adj, "N_bitwise_repeatable": rep, "checker_tolerance": 1e-11}
        del Y, X, CN, CT
    finally:
        dg.free_compressed(obj)
        torch.cuda.empty_cache()
    return res


# ====================================================================================================== the headline workload
class Workload:
    """config 2 staged for this process: the SNP block of this rank (one process per GPU) or the whole matrix behind a multi-device
    object (in-process, MIRACULIX_NUM_GPUS), B and C resident in HBM"""


def setup_process(args):
    import torch
    import torch.distributed as dist
    W = Workload()
    W.world = int(os.environ.get("WORLD_SIZE", "1"))
    W.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # single process asked for several GPUs (no torch.distributed launcher): the SNP shards live BEHIND the C ABI
    # (MIRACULIX_NUM_GPUS, mxa_multi.cpp) -- the path a Julia / Fortran caller gets.  Under the launcher: one rank per GPU, RCCL.
    W.inprocess = W.world == 1 and args.gpus > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # rehearsal knobs (never used by the driver): MXA_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and MXA_BENCH_BACKEND=gloo
    # replaces RCCL, so the N > 1 control flow can be exercised on a one-GPU box
    if os.environ.get("MXA_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("MXA_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    W.device = torch.device("cuda", local_rank)
    os.environ["HIP_DEVICE"] = str(local_rank)
    W.force_dist = os.environ.get("MXA_BENCH_FORCE_DIST") == "1"   # exercise the RCCL path with a 1-rank group
    if W.world > 1 or W.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=W.device)
        else:
            dist.init_process_group(backend)
    W.torch, W.dist = torch, dist
    W.n_gpus = args.gpus if W.inprocess else W.world
    return W


def stage_headline(W, args, mx, L):
    torch, dist, device = W.torch, W.dist, W.device
    from miraculix_amd.distributed import HipLocalEngine, ShardedGenotypeOperator, shard_bounds
    snps, indiv, n = args.snps, args.indiv, args.ncol
    b, e = (0, snps) if W.inprocess else shard_bounds(snps, W.world, W.rank)
    W.begin, W.end, W.snps_loc = b, e, e - b
    if W.snps_loc <= 0:
        raise SystemExit(f"bench.py: rank {W.rank} of {W.world} has no SNPs ({snps} SNPs): use fewer ranks")
    # ---- synthetic data, generated on the device (p_s ~ U(0.1, 0.6), g ~ Binomial(2, p_s), no missings)
    plink = synth_genotypes_device(torch, W.snps_loc, indiv, 42 + W.rank, device)                  # SNP-major
    plink_t = torch.empty((indiv, (W.snps_loc + 3) // 4), dtype=torch.uint8, device=device)        # individual-major
    assert L.mxa_transpose_2bit(mx.lib.ptr(plink), W.snps_loc, indiv, mx.lib.ptr(plink_t)) == 0
    freq = torch.empty(W.snps_loc, dtype=torch.float64, device=device)
    assert L.mxa_allele_freq(mx.lib.ptr(plink), W.snps_loc, indiv, mx.lib.ptr(freq)) == 0
    if W.inprocess:
        os.environ["MIRACULIX_NUM_GPUS"] = str(args.gpus)
    W.eng = HipLocalEngine(plink, plink_t, W.snps_loc, indiv, freq, n, centered=bool(args.centered))
    os.environ.pop("MIRACULIX_NUM_GPUS", None)
    W.n_shards = mx.dgemm_compressed.num_shards(W.eng.obj)
    # the packed rows the in-run oracle check needs (oracle_check): 64 individuals -- the SAME on every rank -- and 64 SNP rows of this process's block
    # (one process: of the whole matrix), copied to the host before the raw matrices are released
    W.sample = extract_sample(torch, dict(dev=device, snps=W.snps_loc, indiv=indiv, plink=plink, plink_t=plink_t, f=freq), nsample=64, seed=1, seed_rows=101 + W.rank)
    W.keep_raw = W.world == 1 and not W.inprocess and not args.no_cpu_baseline       # the CPU-baseline / parity leg samples the raw matrices
    W.plink, W.plink_t, W.freq = (plink, plink_t, freq) if W.keep_raw else (None, None, None)
    del plink, plink_t
    torch.cuda.empty_cache()
    W.op = ShardedGenotypeOperator(W.eng, snps, indiv)
    W.op.force_collective = W.force_dist
    g = torch.Generator(device=device); g.manual_seed(43)
    # MXA_BENCH_TEST_MISCUT=1 (tests only): ranks / shards > 0 take their rows of B four SNPs too early -- an error BOTH products share, which the adjoint
    # identity cannot see and the oracle check must (tests/test_bench_rehearsal_gpu.py)
    W.miscut = 4 if os.environ.get("MXA_BENCH_TEST_MISCUT") == "1" else 0
    bb, ee = (b - W.miscut, e - W.miscut) if (W.rank > 0 and W.miscut) else (b, e)
    W.B_N = torch.randn((n, snps), dtype=torch.float64, device=device, generator=g)[:, bb:ee].contiguous().t()   # snps_loc x n, column-major
    W.B_T = torch.randn((n, indiv), dtype=torch.float64, device=device, generator=g).t()                       # indiv x n
    W.C_N = torch.zeros((n, indiv), dtype=torch.float64, device=device).t()
    W.C_T = torch.zeros((n, W.snps_loc), dtype=torch.float64, device=device).t()
    if W.inprocess:
        # operands PER SHARD, each on its shard's device: 'N' reads B[s_g, :] and 'T' writes C[s_g, :] where they live; nothing but the
        # indiv x n partial sums crosses a device boundary (mxa_dgemm_compressed_multi).  The hub variant (all of B / C on the first
        # device, through the plain dgemm_compressed symbol) is timed beside it as `hub_operands_on_first_device`.
        dg = mx.dgemm_compressed
        W.info0 = dg.multi_info(W.eng.obj)
        W.bounds = dg.shard_bounds(W.eng.obj, snps)
        W.devs = [torch.device("cuda", s["device"]) for s in W.info0["per_shard"]]
        ld = max(e1 - b1 for b1, e1 in W.bounds)
        W.BN_s, W.CT_s, W.BT_s = [], [], []
        for (b1, e1), d in zip(W.bounds, W.devs):
            buf = torch.zeros((n, ld), dtype=torch.float64, device=d)
            mc = W.miscut if b1 > 0 else 0
            buf[:, : e1 - b1] = W.B_N[b1 - mc:e1 - mc].t().to(d)
            W.BN_s.append(buf.t()[: e1 - b1])
            W.CT_s.append(torch.zeros((n, ld), dtype=torch.float64, device=d).t()[: e1 - b1])
            W.BT_s.append(W.B_T if d == device else W.B_T.t().to(d).t())
        W.CN_list = [W.C_N] + [None] * (W.n_shards - 1)


def make_step_and_sync(W, mx, hub=False):
    torch, dist = W.torch, W.dist
    dg = mx.dgemm_compressed
    if W.inprocess and not hub:
        def step():
            dg.dgemm_compressed_multi(False, W.eng.obj, W.BN_s, W.CN_list, sync=False)      # 'N': products, pushes and the addition are enqueued ...
            dg.dgemm_compressed_multi(True, W.eng.obj, W.BT_s, W.CT_s, sync=False)          # ... and 'T' runs on the shard streams beside the reduction

        def sync():
            dg.multi_synchronize(W.eng.obj)
            for d in set(W.devs):
                torch.cuda.synchronize(d)
    else:
        def step():
            _, work = W.op.matmul_N(W.B_N, out=W.C_N, async_op=True)
            W.op.matmul_T(W.B_T, out=W.C_T)
            if work is not None:
                work.wait()

        def sync():
            if dist.is_initialized():
                dist.barrier()
            torch.cuda.synchronize()
    return step, sync


def run_timed(W, step, sync, warmup, steps):
    torch, dist = W.torch, W.dist
    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device=W.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def adjoint_check(W):
    """parity check on every run (size-independent property; the oracle cannot run at this size): the adjoint identity
    <B_T[:,j], Z B_N[:,j]> == <Z^T B_T[:,j], B_N[:,j]> ties the 'N' result (individual-major copy, reduced over the SNP shards) to the
    'T' result (SNP-major copy) column by column"""
    torch, dist = W.torch, W.dist
    lhs = (W.B_T * W.C_N).sum(dim=0)
    if W.inprocess:
        rhs = sum((c * b).sum(dim=0).to(W.device) for c, b in zip(W.CT_s, W.BN_s))
    else:
        rhs = (W.C_T * W.B_N).sum(dim=0)
        if dist.is_initialized():
            dist.all_reduce(rhs, op=dist.ReduceOp.SUM)
    err = float(((lhs - rhs).abs() / lhs.abs().clamp_min(1.0)).max())
    if not (err <= 1e-10):
        raise SystemExit(f"bench.py: adjoint identity violated (rel err {err:.3e}): results are wrong, no number reported")
    return err


def oracle_check(W, args):
    """Parity against the checker on EVERY run and for every N, before timing counts: sampled rows of both products of one untimed step against
    the long-double dense oracle (oracle/oracle.c) on the extracted packed rows; tolerance 1e-11 of the largest reference entry; a violation ends the run
    WITHOUT a number, on every rank.  Independent of the partition logic it checks: the global matrix is the concatenation of the ranks' blocks in rank
    order, the global B of 'N' is regenerated whole from its seed.
      'T' (no exchange): 64 SNP rows of every rank's / of the whole matrix's result -- in-process: read from the shard that owns the row.
      'N' (the reduced result): 64 individuals; under the launcher their packed bytes (64 x snps_loc / 4 per rank; the blocks are cut at multiples of 4) and
           the ranks' allele frequencies are gathered on rank 0, which checks the all-reduced C_N.
    Returns the `check` keys of the line."""
    import numpy as np
    torch, dist = W.torch, W.dist
    snps, indiv, n = args.snps, args.indiv, args.ncol
    cols = [0, n - 1] if n > 1 else [0]
    cen = int(bool(args.centered))
    sm = W.sample
    out = {"tol": 1e-11, "rows_N": int(len(sm["ii"])), "cols": cols}
    if not dist.is_initialized() or dist.get_world_size() == 1:
        err_n, br_n = check_sample(torch, sm, 0, W.B_N, W.C_N, cols, cen, want_bound=True)
        out["N_max_err_over_elementwise_bound"] = br_n          # <= 1: every sampled element within 4 K 2^-53 sum |z||b| of the oracle
        if W.inprocess:
            err_t, rows_t = 0.0, 0
            for g, (b1, e1) in enumerate(W.bounds):
                sel = (sm["ss"] >= b1) & (sm["ss"] < e1)
                if not sel.any():
                    continue
                sub = dict(sm, ss=sm["ss"][sel], rows_s=np.ascontiguousarray(sm["rows_s"][sel]))
                err_t = max(err_t, check_sample(torch, sub, 1, W.BT_s[g], W.CT_s[g], cols, cen, row_offset=b1))
                rows_t += int(sel.sum())
        else:
            err_t, rows_t = check_sample(torch, sm, 1, W.B_T, W.C_T, cols, cen), int(len(sm["ss"]))
    else:
        world, rank = dist.get_world_size(), dist.get_rank()
        on_host = dist.get_backend() != "nccl"                       # the gloo rehearsal gathers host tensors
        cdev = torch.device("cpu") if on_host else W.device
        err_t = check_sample(torch, sm, 1, W.B_T, W.C_T, cols, cen)
        rows_t = int(len(sm["ss"])) * world
        sizes = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([W.snps_loc], dtype=torch.int64, device=cdev))
        sizes = [int(x.item()) for x in sizes]
        maxs = max(sizes)
        rb = (W.snps_loc + 3) // 4
        buf = torch.zeros((len(sm["ii"]), (maxs + 3) // 4), dtype=torch.uint8, device=cdev)
        buf[:, :rb] = torch.from_numpy(sm["rows_t"]).to(cdev)
        fb = torch.zeros(maxs, dtype=torch.float64, device=cdev)
        fb[: W.snps_loc] = torch.from_numpy(sm["f"]).to(cdev)
        gb = [torch.zeros_like(buf) for _ in range(world)]
        gf = [torch.zeros_like(fb) for _ in range(world)]
        dist.all_gather(gb, buf)
        dist.all_gather(gf, fb)
        err_n = 0.0
        if rank == 0:
            if sum(sizes) != snps or any(x % 4 for x in sizes[:-1]):
                raise SystemExit(f"bench.py: the ranks' SNP blocks {sizes} do not tile {snps} SNPs at multiples of 4: no number reported")
            rows = np.ascontiguousarray(np.concatenate([g[:, : (x + 3) // 4].cpu().numpy() for g, x in zip(gb, sizes)], axis=1))
            f_all = np.concatenate([g[:x].cpu().numpy() for g, x in zip(gf, sizes)])
            gen = torch.Generator(device=W.device); gen.manual_seed(43)
            B_all = torch.randn((n, snps), dtype=torch.float64, device=W.device, generator=gen)          # the call of stage_headline, whole
            glob = dict(snps=snps, indiv=indiv, dev=W.device, f=f_all, ii=sm["ii"], rows_t=rows)
            err_n = check_sample(torch, glob, 0, B_all.t(), W.C_N, cols, cen)
            del B_all
        e = torch.tensor([err_t, err_n], dtype=torch.float64, device=cdev)
        e = torch.where(torch.isnan(e), torch.full_like(e, 1e300), e)
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        err_t, err_n = float(e[0].item()), float(e[1].item())
    out.update({"oracle_T_max_rel_err": err_t, "oracle_N_max_rel_err": err_n, "rows_T": rows_t})
    if not (err_t <= 1e-11 and err_n <= 1e-11 and out.get("N_max_err_over_elementwise_bound", 0.0) <= 1.0):
        raise SystemExit(f"bench.py: GPU results differ from the oracle ({out}): no number reported")
    return out


def per_shard_report(W, mx, args, flops_shard):
    """in-process N > 1: what every shard did in the timed region (HIP events on the streams the work ran on)"""
    info = mx.dgemm_compressed.multi_info(W.eng.obj)
    rep = {"reduction": info["reduction"], "root_device": info["root_device"], "devices": info["devices"], "reductions": info["reductions"],
           "avg_reduce_kernel_ms": round(info["reduce_ms"] / max(1, info["reductions"]), 4), "reduce_bytes_per_shard": int(8 * args.indiv * args.ncol),
           "shards": []}
    for s in info["per_shard"]:
        k = s["kernel_ms"] / max(1, s["kernel_launches"])
        rep["shards"].append({"device": s["device"], "snps": s["snp_end"] - s["snp_begin"], "peer_access_to_root": s["peer_to_root"], "peer_access_from_root": s["peer_from_root"],
                              "k_gemm_launches": s["kernel_launches"], "avg_k_gemm_ms": round(k, 4), "k_gemm_TFLOPs": round(flops_shard / (k * 1e-3) * 1e-12, 2) if k > 0 else None,
                              "operand_copies_in": s["in_copies"], "avg_copy_in_ms": round(s["in_ms"] / max(1, s["in_copies"]), 4),
                              "result_copies_out": s["out_copies"], "avg_copy_out_ms": round(s["out_ms"] / max(1, s["out_copies"]), 4),
                              "partial_pushes": s["pushes"], "avg_push_ms": round(s["push_ms"] / max(1, s["pushes"]), 4)})
    return rep, info


def predicted_step_ms(n_gpus, snps, indiv, n, slowest_avg_kgemm_ms):
    """What a step SHOULD take on N physical GPUs if the design holds, so that a measured scaling curve can be judged
    against it: the 'N' and the 'T' product of the slowest shard back to back, with the reduction of the indiv x n partials (p2p pushes + one
    addition kernel, or the RCCL collective) hidden behind the collective-free 'T' product of the same step.
      from_this_run    : 2 x the slowest shard's average k_gemm launch measured in THIS run + 0.3 ms for the small kernels around the two
                         launches (pack_B, colexp, finish: 0.13-0.16 ms per call, profiles/r03_bench_n1_kernel_stats.csv)
      single_gpu_table : the same sum measured on ONE MI355X at the per-shard shape of the default workload (1M x 50k x 32; round 3:
                         1 GPU 88.6 ms, 2 x 500k 45.0, 4 x 250k 23.1, 8 x 125k 11.6), None for other workloads
      if_reduction_not_hidden : + 8 indiv n bytes over one xGMI link at ~50 GB/s effective + the addition kernel"""
    table = {1: 88.6, 2: 45.0, 4: 23.1, 8: 11.6} if (snps, indiv, n) == (1_000_000, 50_000, 32) else {}
    this = 2.0 * slowest_avg_kgemm_ms + 0.3
    exposed = 8.0 * indiv * n / 50e9 * 1e3 + 0.06
    return {"from_this_run": round(this, 3), "single_gpu_table": table.get(n_gpus), "if_reduction_not_hidden": round(this + exposed, 3),
            "model": "2 x slowest shard's avg k_gemm launch + 0.3 ms of small kernels; reduction hidden behind the 'T' product (DESIGN.md section 6)"}


def opt_in_engine_leg(W, L, args, step, sync, engine, flops_step, description):
    """the headline steps once more under another multiply engine: fp64-equivalent rate, kernel time, executed int8 rate, and the
    column-wise difference from the fp64 engine's results (which the caller has checked against the oracle)"""
    C_N64, C_T64 = W.C_N.clone(), W.C_T.clone()
    L.mxa_set_engine(engine)
    try:
        step(); sync()
        L.mxa_profile_reset()
        dt8 = run_timed(W, step, sync, 0, args.steps)
        la8, ms8 = kernel_profile(L)
        gm, gk, gn, gs, ga, gc = ctypes.c_long(), ctypes.c_long(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        L.mxa_last_geometry(ctypes.byref(gm), ctypes.byref(gk), ctypes.byref(gn), ctypes.byref(gs), ctypes.byref(ga), ctypes.byref(gc))
        path = L.mxa_last_path()
    finally:
        L.mxa_set_engine(0)
    digits = ga.value if engine == 4 else 7          # of the last product of the step ('T'); engine 4 reports its per-call choice
    one_copy = bool(L.mxa_single_orientation(W.eng.obj) == 1)
    dN = float(((W.C_N - C_N64).abs().amax(dim=0) / C_N64.abs().amax(dim=0)).max())
    dT = float(((W.C_T - C_T64).abs().amax(dim=0) / C_T64.abs().amax(dim=0)).max())
    avg_ms = ms8 / max(1, la8)
    if one_copy:   # the headline object keeps one packed copy (the default): 'N' then runs the transposed-operand int8 kernel in column chunks
        description += ("; one-copy object (default): 'T' on k_gemm_i8, 'N' on k_gemm_i8_tn in column chunks of <= 6 digit tiles, one pass over the packed matrix per two "
                        "tiles (both copies, MXA_SINGLE_ORIENTATION=0: 'N' on the plain kernel in one launch)")
    out = {"engine": description, "kernel_family_of_last_product": {0: "k_gemm", 1: "k_lut", 2: "k_gemm_i8", 3: "k_small_n_fp64"}.get(path, str(path)),
           "value": round(flops_step * args.steps / dt8 * 1e-9, 1), "unit": "GFLOP/s (fp64-equivalent: same 2*snps*indiv*ncol count)",
           "ms_per_step": round(dt8 / args.steps * 1e3, 3), "avg_kernel_ms": round(avg_ms, 3), "digits_per_column": digits,
           "int8_ops_per_s_P": round(2.0 * W.snps_loc / W.n_shards * args.indiv * args.ncol * digits / (avg_ms * 1e-3) * 1e-15, 3),
           "max_colwise_rel_diff_vs_f64_engine": max(dN, dT)}
    W.C_N.copy_(C_N64); W.C_T.copy_(C_T64)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--snps", type=int, default=1_000_000)
    ap.add_argument("--indiv", type=int, default=50_000)
    ap.add_argument("--ncol", type=int, default=32)
    ap.add_argument("--centered", type=int, default=0)
    ap.add_argument("--reduce", choices=["auto", "p2p", "rccl"], default="auto",
                    help="in-process N > 1 (no launcher): the reduction of the 'N' partials that is TIMED and reported as `value` -- rccl = ncclReduce over xGMI (the "
                         "collective north_star names), p2p = peer-to-peer pushes + one fixed-order addition kernel (bitwise reproducible); auto (default) = rccl when "
                         "every shard has a device of its own, else p2p, and p2p with the reason in the line if RCCL fails.  The other one is reported beside it.  "
                         "Under the torch.distributed launcher the reduction is always RCCL's all-reduce.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-engine", action="store_true", help="skip the extra (untimed, informational) pass with the opt-in int8 engine")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic")
    ap.add_argument("--no-abi", action="store_true", help="skip the ABI end-to-end leg (host B / C through dgemm_compressed)")
    ap.add_argument("--no-configs", action="store_true", help="skip the legs for BASELINE configs 3, 4 (shard) and 5 (shard)")
    ap.add_argument("--configs-scale", type=float, default=1.0, help=argparse.SUPPRESS)   # rehearsals shrink the config legs
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner when its communicator comes up, i.e. on every
    # N > 1 run): keep the real stdout aside for the line and send everything else that reaches fd 1 to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver only supports dmabuf IPC (RCCL and peer access between processes need it); set before HIP comes up

    world = int(os.environ.get("WORLD_SIZE", "1"))
    inprocess = world == 1 and args.gpus > 1
    # roofline.traffic is measured first, before this process touches the GPU (N = 1 only; the children profile the same workload)
    traffic, traffic_detail = None, {"skipped": "N > 1" if (world > 1 or inprocess) else "--no-pmc"}
    if world == 1 and not inprocess and not args.no_pmc:
        traffic, traffic_detail = measure_traffic(args)

    W = setup_process(args)
    torch, dist = W.torch, W.dist
    import miraculix_amd as mx
    L = mx.load_shared_library()
    L.mxa_set_engine(0)   # the headline number is the fp64 engine, whatever MXA_ENGINE says
    stage_headline(W, args, mx, L)
    snps, indiv, n = args.snps, args.indiv, args.ncol
    dg = mx.dgemm_compressed

    # ---- the timed region: W untimed steps, then exactly K steps between barrier + synchronize on both sides
    step, sync = make_step_and_sync(W, mx)
    reduction = "none (one GPU)"
    reduction_info = {"world_size": 1, "rccl_ranks": 0}
    if dist.is_initialized():
        reduction = "rccl all-reduce (torch.distributed, one process per GPU)" if os.environ.get("MXA_BENCH_BACKEND", "nccl") == "nccl" else "gloo all-reduce (rehearsal)"
        reduction_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_ranks": dist.get_world_size() if dist.get_backend() == "nccl" else 0}
        # can this rank's device reach rank 0's memory directly (xGMI peer access)?  -1: same device (one-GPU rehearsal)
        me, root = torch.cuda.current_device(), 0
        pa = -1 if (os.environ.get("MXA_BENCH_SINGLE_DEVICE") == "1" or me == root) else int(torch.cuda.can_device_access_peer(me, root))
        cdev = torch.device("cpu") if dist.get_backend() != "nccl" else W.device
        allpa = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(dist.get_world_size())]
        dist.all_gather(allpa, torch.tensor([pa], dtype=torch.int64, device=cdev))
        reduction_info["peer_access_to_rank0"] = [int(x.item()) for x in allpa]
    elif W.inprocess:
        reduction = "p2p"
        if args.reduce in ("auto", "rccl"):   # the timed reduction is RCCL's wherever RCCL applies; any failure falls back to p2p and is said in the line
            try:
                if dg.multi_set_reduction(W.eng.obj, "rccl"):
                    reduction = "rccl"
                else:
                    reduction = "p2p (rccl not applicable: several shards share a device, RCCL needs one rank per device)"
            except RuntimeError as ex:
                reduction = f"p2p (rccl failed: {ex})"
                try:
                    dg.multi_set_reduction(W.eng.obj, "p2p")
                except RuntimeError:
                    pass
    # ---- parity first ("parity check on every run ... before timing counts"): one untimed step, both results against the oracle, for every N
    step()
    sync()
    chk = oracle_check(W, args)
    for _ in range(args.warmup):
        step()
    sync()
    L.mxa_profile_reset()
    if W.inprocess:
        L.mxa_multi_reset_profile(W.eng.obj)
    try:
        dt = run_timed(W, step, sync, 0, args.steps)
    except RuntimeError as ex:
        if not (W.inprocess and reduction == "rccl"):
            raise
        # RCCL came up but a reduction failed (the first one is cross-checked against the peer-to-peer sum inside the library): p2p, said in the line
        reduction = f"p2p (rccl failed in the timed region: {ex})"
        dg.multi_set_reduction(W.eng.obj, "p2p")
        for _ in range(max(1, args.warmup)):
            step()
        sync()
        L.mxa_profile_reset()
        L.mxa_multi_reset_profile(W.eng.obj)
        dt = run_timed(W, step, sync, 0, args.steps)
    if W.inprocess:
        info0 = dg.multi_info(W.eng.obj)
        reduction_info = {"world_size": 1, "shards": info0["shards"], "rccl_ranks": info0["shards"] if reduction == "rccl" else 0, "rccl_checked": bool(info0["rccl_checked"]),
                          "rccl_vs_p2p_max_rel_diff": info0["rccl_vs_p2p_max_rel_diff"], "devices": info0["devices"],
                          "peer_access_to_root": [s["peer_to_root"] for s in info0["per_shard"]]}
    launches, total_ms = kernel_profile(L)
    adj_err = adjoint_check(W)
    flops_step = 2 * 2.0 * snps * indiv * n
    value = flops_step * args.steps / dt * 1e-9
    ms_per_step = dt / args.steps * 1e3
    # dominant kernel: k_gemm; algorithmic flops per launch on one device = 2 * (SNPs of the shard) * indiv * n
    flops_launch = 2.0 * W.snps_loc / W.n_shards * indiv * n
    avg_ms = total_ms / max(1, launches)
    per_shard = None
    if W.inprocess:
        per_shard, _ = per_shard_report(W, mx, args, flops_launch)
        slowest = max(s["avg_k_gemm_ms"] for s in per_shard["shards"])
        avg_ms = slowest if slowest > 0 else avg_ms        # the roofline line quotes the SLOWEST shard's average launch
        per_shard["note"] = ("operands per shard on the shard's own device (mxa_dgemm_compressed_multi, asynchronous): the partial sums of 'N' travel on copy "
                             "streams and are added on the root device while the 'T' products of the same step run")
    achieved = flops_launch / (avg_ms * 1e-3) * 1e-12 if avg_ms > 0 else 0.0

    # in-process N > 1, extra (never `value`): (i) the same steps with the RCCL reduction (its first product is cross-checked against the
    # peer-to-peer one inside the library), (ii) the hub variant: all of B / C on the first device through the plain dgemm_compressed symbol
    extra_multi = None
    if W.inprocess:
        extra_multi = {}
        other = "p2p" if reduction == "rccl" else "rccl"
        key = other + "_reduction"
        try:   # a failure of this optional leg is reported in its key; it must not take the headline number down
            if dg.multi_set_reduction(W.eng.obj, other):
                L.mxa_multi_reset_profile(W.eng.obj)
                dt_r = run_timed(W, step, sync, 1, args.steps)
                rep_r, info_r = per_shard_report(W, mx, args, flops_launch)
                adjoint_check(W)
                extra_multi[key] = {"value": round(flops_step * args.steps / dt_r * 1e-9, 1), "unit": "GFLOP/s", "ms_per_step": round(dt_r / args.steps * 1e3, 3),
                                    "rccl_vs_p2p_max_rel_diff": info_r["rccl_vs_p2p_max_rel_diff"], "rccl_checked": bool(info_r["rccl_checked"]),
                                    "avg_push_or_ncclReduce_ms_on_root_rank": rep_r["shards"][0]["avg_push_ms"]}
            else:
                extra_multi[key] = {"skipped": "several shards share a device (RCCL needs one rank per device)"}
        except RuntimeError as ex:
            extra_multi[key] = {"failed": str(ex)}
        try:
            dg.multi_set_reduction(W.eng.obj, "rccl" if reduction == "rccl" else "p2p")
        except RuntimeError:
            pass
        hub_step, hub_sync = make_step_and_sync(W, mx, hub=True)
        hub_steps = max(1, min(args.steps, 5))
        L.mxa_multi_reset_profile(W.eng.obj)
        dt_h = run_timed(W, hub_step, hub_sync, 1, hub_steps)
        rep_h, _ = per_shard_report(W, mx, args, flops_launch)
        extra_multi["hub_operands_on_first_device"] = {"value": round(flops_step * hub_steps / dt_h * 1e-9, 1), "unit": "GFLOP/s", "ms_per_step": round(dt_h / hub_steps * 1e3, 3),
                                                       "steps": hub_steps, "avg_copy_in_ms_per_shard": [s["avg_copy_in_ms"] for s in rep_h["shards"]],
                                                       "avg_copy_out_ms_per_shard": [s["avg_copy_out_ms"] for s in rep_h["shards"]],
                                                       "what": "B and C whole on the first device, plain dgemm_compressed, synchronous calls: every shard copies its slice from / to that device"}
        step(); sync()     # the per-shard results are current again for the legs below

    # N > 1 under the launcher: what every rank's dominant kernel took, and what the all-reduce of the indiv x n result costs on its own
    # (untimed extra pass), so that a scaling curve explains itself
    per_rank = None
    if dist.is_initialized():
        km = torch.tensor([avg_ms], dtype=torch.float64, device=W.device)
        allk = [torch.zeros_like(km) for _ in range(dist.get_world_size())]
        dist.all_gather(allk, km)
        flat = W.C_N.t()
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

    # informational extra passes, outside the timed region: the same steps with the two opt-in int8 engines (include/miraculix_amd.h,
    # mxa_set_engine).  Reported beside the headline, never as `value`.
    alt = alt_exact = None
    if not args.no_alt_engine and not W.inprocess:
        alt = opt_in_engine_leg(W, L, args, step, sync, 1, flops_step,
                                "i8: B split into 7 radix-256 digits per column (to 2^-54 of the column maximum, no exactness check), v_mfma_i32_32x32x32_i8, exact int32 sums, fp64 recombination")
        alt_exact = opt_in_engine_leg(W, L, args, step, sync, 4, flops_step,
                                      "i8-exact: digit count chosen per call from the measured exponent span of B so that B is represented WITHOUT error "
                                      "(|error| <= 3.02 (S-1) 2^-53 sum|z b| per output, tighter than an fp64 FMA chain); fp64 MFMA path when that needs more than 24 digits")

    # ABI end-to-end (reference harness utils/benchmark/benchmark.f90:192-209): the same two products with HOST
    # B and C through the plain reference symbol dgemm_compressed -- what a Julia / Fortran caller sees, PCIe included.
    # 1 warm-up + 10 repetitions, mean and min.  Never `value`.
    abi = None
    if W.world == 1 and not args.no_abi:
        import numpy as np
        hB_N, hB_T = np.asfortranarray(W.B_N.cpu().numpy()), np.asfortranarray(W.B_T.cpu().numpy())
        hC_N, hC_T = np.zeros((indiv, n), order="F"), np.zeros((snps, n), order="F")

        def abi_step():
            dg.dgemm_compressed_main(False, W.eng.obj, hB_N, snps, indiv, out=hC_N)
            dg.dgemm_compressed_main(True, W.eng.obj, hB_T, snps, indiv, out=hC_T)
        abi_step()
        ts = []
        for _ in range(10):
            t1 = time.perf_counter()
            abi_step()
            ts.append(time.perf_counter() - t1)
        dev_T = np.concatenate([c.cpu().numpy() for c in W.CT_s]) if W.inprocess else W.C_T.cpu().numpy()
        same = bool(np.array_equal(hC_N, W.C_N.cpu().numpy()) and np.array_equal(hC_T, dev_T))
        abi = {"what": "dgemm_compressed 'N' + 'T' with host (pageable) B and C, PCIe transfers inside the call; 1 warm-up + 10 repetitions",
               "mean_GFLOPs": round(flops_step / (sum(ts) / len(ts)) * 1e-9, 1), "max_GFLOPs": round(flops_step / min(ts) * 1e-9, 1),
               "mean_ms_per_step": round(sum(ts) / len(ts) * 1e3, 3), "min_ms_per_step": round(min(ts) * 1e3, 3),
               "host_bytes_per_step": int(8 * 2 * (snps + indiv) * n), "bitwise_equal_to_device_resident_results": same}
        del hB_N, hB_T, hC_N, hC_T, dev_T
        if W.keep_raw:
            # row a2 of the path: plink2compressed itself with HOST matrices, as the reference harness times it (utils/benchmark/benchmark.f90:187):
            # both orientations over PCIe (the reference's call shape for its GPU path), and the SNP-major matrix alone with the individual-major copy
            # built on the device (plink_transposed = NULL / = plink, the call shape of its CPU path, benchmark.f90:185).  Never part of `value`.
            hp, hpt, hf = W.plink.cpu().numpy(), W.plink_t.cpu().numpy(), W.freq.cpu().numpy()
            stage = {}
            for key, second in (("two_host_pointers", hpt), ("snp_major_only_device_transpose", None)):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                o2 = dg.init_compressed(hp, second, snps, indiv, hf, n)
                stage[key] = time.perf_counter() - t1
                C2 = dg.dgemm_compressed_main(True, o2, W.B_T, snps, indiv)
                stage[key + "_same"] = bool(torch.equal(C2, W.C_T))
                del C2
                dg.free_compressed(o2)
            abi["plink2compressed_host_staging_s"] = round(stage["two_host_pointers"], 3)
            abi["plink2compressed_host_staging_GB"] = round((hp.nbytes + hpt.nbytes) / 1e9, 2)
            abi["plink2compressed_snp_major_only_s"] = round(stage["snp_major_only_device_transpose"], 3)
            abi["plink2compressed_snp_major_only_GB"] = round(hp.nbytes / 1e9, 2)
            abi["staged_objects_reproduce_the_T_result_bitwise"] = stage["two_host_pointers_same"] and stage["snp_major_only_device_transpose_same"]
            del hp, hpt, hf

    out = None
    if W.rank == 0:
        out = {
            "metric": "effective GFLOP/s for dgemm_compressed (2-bit SNP x fp64)",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": W.n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "reduction": reduction, **reduction_info,
            "config": {"workload": f"{snps} SNPs x {indiv} indiv, ncol={n}, dgemm_compressed 'N' + 'T' per step, "
                                   f"{'centred' if args.centered else 'uncentred'}, SNP-sharded over {W.n_gpus} GPU(s)"
                                   + (" inside one process behind the C ABI (MIRACULIX_NUM_GPUS), operands per shard on the shards' devices" if W.inprocess else ""),
                       "snps": snps, "indiv": indiv, "ncol": n, "parallelism": f"snp-shard{W.n_gpus}",
                       "genotypes": "p_s ~ U(0.1, 0.6), g ~ Binomial(2, p_s), no missings; B ~ N(0, 1)"},
            "check": {"adjoint_identity_max_rel_err": adj_err, "adjoint_tolerance": 1e-10, **chk},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "GB per launch, measured by this run (two rocprofv3 --pmc child passes of the same workload)", "traffic_detail": traffic_detail,
                         "algorithmic_bytes_per_launch_GB": round((W.snps_loc / W.n_shards * ((indiv + 3) // 4) + 8.0 * (W.snps_loc / W.n_shards + indiv) * n) / 1e9, 3),
                         "kernel": "k_gemm<8,8,3> on v_mfma_f64_4x4x4_4b_f64, two launches per step on the ONE stored (SNP-major) copy: 'N' = k_gemm<8, 8, 3, false, true> "
                                   "(transposed-operand form), 'T' = k_gemm<8, 8, 3, false, false> (plain form); same tile, same permuted K order, same rate", "launches": launches, "avg_launch_ms": round(avg_ms, 3)},
        }
        if W.n_gpus > 1:
            slowest = max(per_rank["avg_k_gemm_launch_ms"]) if per_rank is not None else avg_ms
            out["predicted_ms_per_step_from_per_shard"] = predicted_step_ms(W.n_gpus, snps, indiv, n, slowest)
        if per_rank is not None:
            out["per_rank"] = per_rank
        if per_shard is not None:
            out["per_shard"] = per_shard
            out.update(extra_multi)
        if alt is not None:
            out["opt_in_engine"] = alt
        if alt_exact is not None:
            out["opt_in_engine_exact"] = alt_exact
        if abi is not None:
            out["abi_end_to_end"] = abi
        if W.keep_raw:   # CPU baseline + parity against the checker: rank 0 at N = 1 only
            def rows_of_plink_t(ii):
                return W.plink_t[torch.from_numpy(ii).to(W.device)].cpu().numpy()
            base, port, chk = cpu_baseline_and_check(torch, mx, W.plink, W.freq, snps, indiv, n, W.B_T, W.C_T, W.B_N, W.C_N, rows_of_plink_t)
            out["cpu_baseline"] = base
            if port is not None:
                out["cpu_baseline_port"] = port
            out["check"].update(chk)
            if not (chk["gpu_T_rows_vs_cpu_library_max_rel_err"] <= 1e-11 and chk["gpu_N_64_sampled_rows_vs_dense_oracle_max_rel_err"] <= 1e-11):
                raise SystemExit(f"bench.py: GPU results differ from the checker ({chk}): no number reported")
    # the headline object and its operands are released before the other configs are staged
    W.eng.close()
    for name in ("plink", "plink_t", "freq", "B_N", "B_T", "C_N", "C_T", "BN_s", "CT_s", "BT_s", "CN_list", "op"):
        if hasattr(W, name):
            delattr(W, name)
    torch.cuda.empty_cache()

    # ---- BASELINE configs 3, 4 (per-GPU shard) and 5 (per-GPU shard), N = 1 only, each under its own checker; never `value`
    if W.rank == 0 and W.world == 1 and not W.inprocess and not args.no_configs:
        sc = args.configs_scale
        r = lambda x, q: x if sc == 1.0 else max(q, int(x * sc) // q * q)
        legs = (("config5_cg_step", lambda: config5_cg_step_leg(torch, mx, L, W.device, r(250_000, 4), r(100_000, 4))),
                ("config4_shard", lambda: config4_shard_leg(torch, mx, L, W.device, r(625_000, 4), r(200_000, 4))),
                ("config3_crossprod", lambda: config3_crossprod_leg(torch, mx, L, W.device, r(500_000, 4), r(100_000, 256))),
                ("config5_full_8_virtual_shards", lambda: config5_full_leg(torch, mx, L, W.device, r(2_000_000, 32), r(100_000, 4))),
                ("config4_full_extent_8_virtual_shards", lambda: config4_full_extent_leg(torch, mx, L, W.device, r(5_000_000, 32), r(25_000, 4))),
                ("config4_full_one_copy", lambda: config4_full_one_copy_leg(torch, mx, L, W.device, r(5_000_000, 32), r(200_000, 4), 128, r(50_000, 4),
                                                                            log=lambda m: print(m, file=sys.stderr, flush=True))))
        for name, fn in legs:
            t_leg = time.perf_counter()
            try:
                out[name] = fn()
                out[name]["leg_wall_s"] = round(time.perf_counter() - t_leg, 1)
            except Exception as ex:   # a leg must not take the headline number down: it reports its failure instead
                out[name] = {"failed": f"{type(ex).__name__}: {ex}"}
        dg.set_options(use_gpu=True, not_center=not args.centered, verbose=0)
        bad = [k for k, _ in legs if not leg_checks_ok(out[k])]
        if bad:
            raise SystemExit(f"bench.py: parity check failed in {bad}: {json.dumps({k: out[k] for k in bad})}")
    if W.rank == 0:
        # everything measured goes to bench_detail.json (next to this file, and to gpurun_out/ when that exists); the ONE line on stdout is its compact
        # summary (<= 4 KB, so that the driver's record keeps every leg)
        for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
            if os.path.isdir(d):
                try:
                    with open(os.path.join(d, "bench_detail.json"), "w") as fh:
                        json.dump(out, fh, indent=1)
                except OSError:
                    pass
        line = json.dumps(compact_line(out), separators=(",", ":"))
        assert len(line) <= 4096, len(line)
        sys.stdout.flush()
        os.write(real_stdout, (line + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


def compact_line(d):
    """the contract's JSON line: the headline, its roofline and CPU baseline, the in-run checks, who ran the reduction, and ONE number per
    leg of the other BASELINE configs (c3_* crossproduct, c4_* config 4, c5_* CG step); prose and per-shard tables stay in bench_detail.json"""
    g = lambda x, *ks: (g(x.get(ks[0]), *ks[1:]) if len(ks) > 1 else x.get(ks[0])) if isinstance(x, dict) and ks else x
    rf = d["roofline"]
    line = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    c = d["config"]
    line["config"] = {"workload": f"{c['snps']}x{c['indiv']} (SNPs x indiv), ncol={c['ncol']}, dgemm_compressed N+T per step, {'centred' if 'uncentred' not in c['workload'] else 'uncentred'}",
                      "snps": c["snps"], "indiv": c["indiv"], "ncol": c["ncol"], "parallelism": c["parallelism"], "form": "in-process (MIRACULIX_NUM_GPUS)" if "inside one process" in c["workload"] else "one process per GPU"}
    line["roofline"] = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")} | {"kernel": "k_gemm<8,8,3>", "avg_launch_ms": rf["avg_launch_ms"], "launches": rf["launches"],
                                                                                                       "algorithmic_GB": rf["algorithmic_bytes_per_launch_GB"]}
    for k in ("cpu_baseline", "cpu_baseline_port"):
        if k in d:
            b = d[k]
            line[k] = {"value": b["value"], "unit": b["unit"], "cores": b["cores"], "kind": b["kind"].split(" (")[0], "sample": b["sample"].split(", one 'N'")[0]}
    ck = d["check"]
    line["check"] = {k: ck[k] for k in ("oracle_T_max_rel_err", "oracle_N_max_rel_err", "rows_T", "rows_N", "tol", "adjoint_identity_max_rel_err",
                                        "gpu_T_rows_vs_cpu_library_max_rel_err", "gpu_N_64_sampled_rows_vs_dense_oracle_max_rel_err") if k in ck}
    line["reduction"] = d["reduction"][:90]
    for k in ("world_size", "backend", "shards", "rccl_ranks", "rccl_checked", "devices", "peer_access_to_rank0", "peer_access_to_root"):
        if k in d:
            line[k] = d[k]
    if "predicted_ms_per_step_from_per_shard" in d:
        line["predicted_ms_per_step"] = d["predicted_ms_per_step_from_per_shard"]["from_this_run"]
    if "per_rank" in d:
        line["k_gemm_ms_per_rank"] = d["per_rank"]["avg_k_gemm_launch_ms"]
        line["allreduce_alone_ms"] = d["per_rank"]["allreduce_alone_ms"]
    if "per_shard" in d:
        line["k_gemm_ms_per_shard"] = [x["avg_k_gemm_ms"] for x in d["per_shard"]["shards"]]
        line["reduce_kernel_ms"] = d["per_shard"]["avg_reduce_kernel_ms"]
    for k in ("p2p_reduction", "rccl_reduction", "hub_operands_on_first_device"):
        if k in d:
            line[k.split("_")[0] + "_GFLOPs"] = g(d[k], "value") if "value" in d[k] else next(iter(d[k].values()))[:60]
    if "abi_end_to_end" in d:
        line["abi_GFLOPs"] = d["abi_end_to_end"]["mean_GFLOPs"]
        line["abi_bitwise_equal"] = d["abi_end_to_end"]["bitwise_equal_to_device_resident_results"]
    for k, short in (("opt_in_engine", "i8"), ("opt_in_engine_exact", "i8_exact")):
        if k in d:
            line["engine_" + short + "_GFLOPs"] = d[k]["value"]
    failed = [k for k in d if isinstance(d[k], dict) and "failed" in d[k]]
    c3 = d.get("config3_crossprod")
    if c3 and "failed" not in c3:
        for key, e in (("k_crossprod_i8 (int8 MFMA)", "i8"), ("k_crossprod_f4 (FP4 MFMA, default)", "f4")):
            line[f"c3_ms_{e}"] = c3[key]["kernel_ms"]
            line[f"c3_frac_{e}"] = c3[key]["frac_of_peak_executed"]
        line["c3_exact"] = all(c3[key]["check"]["four_256x256_tiles_and_mirrors_bit_exact_vs_int32_oracle"] for key in c3 if key.startswith("k_crossprod"))
        line["c3_workload"] = c3["workload"].split(",")[1].strip() + ", int8->int32 MFMA path (c3_*_i8) = the engine BASELINE config 3 names; FP4 MFMA = default while exact"
    c5 = d.get("config5_cg_step")
    if c5 and "failed" not in c5:
        line.update({"c5_step_ms": c5["ms_per_cg_step"], "c5_step_ms_two_copies": c5["ms_per_cg_step_two_copies"], "c5_TBps": c5["algorithmic_TB_per_s"], "c5_frac_hbm": c5["frac_of_8_TBs_spec"],
                     "c5_bitwise_T_then_N": c5["check"]["gram_matvec_bitwise_equals_T_then_N"]})
    c4 = d.get("config4_shard")
    if c4 and "failed" not in c4:
        line["c4_shard_frac"] = [c4["N"]["frac_of_fp64_mfma_peak_kernel"], c4["T"]["frac_of_fp64_mfma_peak_kernel"]]
    c4f = d.get("config4_full_one_copy")
    if c4f and "failed" not in c4f:
        line["c4_full_TFLOPs"] = [c4f["N"]["TFLOPs_call"], c4f["T"]["TFLOPs_call"]]
        line["c4_full_snps"] = c4f["snps_run"]
    c5f = d.get("config5_full_8_virtual_shards")
    if c5f and "failed" not in c5f:
        line["c5_full_ms"] = {k: c5f[k]["ms_per_gram_matvec"] for k in c5f if isinstance(c5f[k], dict) and "ms_per_gram_matvec" in c5f[k]}
    c4e = d.get("config4_full_extent_8_virtual_shards")
    if c4e and "failed" not in c4e:
        line["c4_extent_TFLOPs_8_shards"] = [g(c4e, next(k for k in c4e if k.endswith("virtual_shards")), t, "TFLOPs_call") for t in ("N", "T")]
    if any(k.startswith("config") and k != "config" for k in d):
        line["legs_parity_ok"] = True          # (the legs ran; a violated leg check ends the run before this line is printed)
    if failed:
        line["legs_failed"] = failed
    line["detail"] = "bench_detail.json"
    return line


def leg_checks_ok(leg):
    """False when a PARITY check of the leg is violated (the run then ends without a number).  A leg that could not run at all (an exception: out of
    memory on a shared device, a profiler in the way) carries {"failed": ...} in the line instead and does not take the headline down."""
    if "failed" in leg:
        return True
    ok = True

    def walk(d):
        nonlocal ok
        for k, v in d.items():
            if isinstance(v, dict):
                walk(v)
            elif k.endswith("max_rel_err"):
                ok &= v <= 1e-11
            elif k.endswith("_over_elementwise_bound"):
                ok &= v <= 1.0
            elif k.startswith("four_256") or k.startswith("first_panel") or "bitwise" in k or k.endswith("_ok"):
                ok &= bool(v)
    walk(leg)
    return ok


if __name__ == "__main__":
    main()
