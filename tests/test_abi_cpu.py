"""CPU checks of the drop-in boundary: the shared library loads and exports every symbol include/miraculix_amd.h
declares; host-side option logic behaves like the reference's; no compute is attempted without a GPU and the
compute entries fail loudly (no CPU fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "miraculix_amd.h")
LIB = os.path.join(ROOT, "miraculix_amd", "lib", "libmiraculix_amd.so")


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "miraculix_amd", "csrc"), "-j4"])
    return LIB


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)          # preprocessor lines
    return sorted(set(re.findall(r"\b(\w+)\s*\([^;{]*\)\s*;", text)))


def test_header_declares_reference_abi():
    syms = _declared_symbols()
    for s in ["setOptions_compressed", "plink2compressed", "dgemm_compressed", "free_compressed", "get_compressed_freq", "snp_multiply_gpu"]:
        assert s in syms


def test_library_exports_every_declared_symbol(built):
    L = ctypes.CDLL(built)
    for s in _declared_symbols():
        assert hasattr(L, s), f"{s} declared in include/miraculix_amd.h but not exported"


def test_symbols_are_unmangled_c(built):
    out = subprocess.check_output(["nm", "-D", "--defined-only", built], text=True)
    names = {l.split()[-1] for l in out.splitlines() if l.strip()}
    for s in _declared_symbols():
        assert s in names


def test_library_exports_the_c_abi_and_nothing_else(built):
    """a drop-in .so is dlopen()ed into Julia / R / Fortran processes: no C++ symbol of the implementation (namespace mxa, kernel handles) may reach
    their namespace -- the dynamic symbol table is exactly what the header declares (-fvisibility=hidden + csrc/exports.map)"""
    out = subprocess.check_output(["nm", "-D", "--defined-only", built], text=True)
    names = {l.split()[-1] for l in out.splitlines() if l.strip()}
    assert not [x for x in names if x.startswith("_Z")]
    assert names == set(_declared_symbols()), names ^ set(_declared_symbols())


def _run_py(code):
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=300)


def test_use_gpu_zero_reports_and_survives(built):
    """GPU-only engine: setOptions_compressed(use_gpu=0, ...) must fail loudly, never fall back to a CPU path -- but the host
    process (a Julia / R session) survives: the error is remembered, later plink2compressed calls leave the handle NULL (the Julia
    binding throws on that, miraculix.jl:29-35), and a following use_gpu=1 call re-enables the engine."""
    r = _run_py(
        "import ctypes, numpy as np, miraculix_amd as m\n"
        "L = m.load_shared_library()\n"
        "L.setOptions_compressed(0,0,0,0,1,0,0,0,0,0)\n"
        "assert L.mxa_last_error() == 14, L.mxa_last_error()\n"
        "obj = ctypes.c_void_p(None)\n"
        "p = np.zeros((8, 1), np.uint8); pt = np.zeros((4, 2), np.uint8); f = np.zeros(8)\n"
        "L.plink2compressed(m.lib.ptr(p), m.lib.ptr(pt), 8, 4, m.lib.ptr(f), 1, ctypes.byref(obj))\n"
        "assert not obj.value and L.mxa_last_error() == 14\n"
        "try:\n    m.dgemm_compressed.set_options(use_gpu=False)\n    print('NOFAIL')\nexcept RuntimeError as e:\n    print('RAISED')\n"
        "L.setOptions_compressed(1,0,0,0,1,0,0,0,0,0)\n"
        "assert L.mxa_last_error() == 0\n"
        "print('ALIVE')\n")
    assert r.returncode == 0, r.stderr
    assert "ALIVE" in r.stdout and "RAISED" in r.stdout and "NOFAIL" not in r.stdout
    assert "MI355X" in r.stderr


def test_reference_fatal_option_combination(built):
    """5codesChar.cc:192-193: gpu && (use_miraculix_freq || !ignore_missings || do_normalize) -> error + exit"""
    code = ("import ctypes, miraculix_amd as m; L = m.load_shared_library(); "
            "L.setOptions_compressed(1,0,0,0,{im},0,{norm},{mf},0,0)")
    for im, norm, mf in [(0, 0, 0), (1, 1, 0), (1, 0, 1)]:
        r = _run_py(code.format(im=im, norm=norm, mf=mf))
        assert r.returncode != 0
        assert "Fortran frequency" in r.stderr
    r = _run_py(code.format(im=1, norm=0, mf=0))
    assert r.returncode == 0


def test_bad_trans_exits_99(built):
    """5codesAPI.c:73-77: any trans other than N n T t Y y -> exit(99)"""
    r = _run_py("import ctypes, miraculix_amd as m; L = m.load_shared_library(); L.dgemm_compressed(b'X', None, 1, None, 1, None, 1)")
    assert r.returncode == 99


def test_uninitialised_object_is_rejected(built):
    import miraculix_amd as m
    L = m.load_shared_library()
    L.dgemm_compressed(b"N", None, 1, None, 1, None, 1)   # prints, sets the error, does not crash
    assert L.mxa_last_error() != 0
    with pytest.raises(RuntimeError):
        m.dgemm_compressed.check_storage_object(ctypes.c_void_p(None))


def test_plink2compressed_without_gpu_fails_loudly(built):
    """on a machine without a HIP device the staging call must leave the handle NULL and report, not compute on the CPU"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = _run_py(
        "import numpy as np, miraculix_amd as m\n"
        "dg = m.dgemm_compressed; dg.set_options(use_gpu=True, verbose=0)\n"
        "p = np.zeros((8, 1), np.uint8); pt = np.zeros((4, 2), np.uint8)\n"
        "try:\n    dg.init_compressed(p, pt, 8, 4, np.zeros(8), 1)\n    print('NOFAIL')\nexcept RuntimeError as e:\n    print('RAISED', e)\n")
    assert "RAISED" in r.stdout and "NOFAIL" not in r.stdout


def test_missing_library_is_an_error(tmp_path):
    r = _run_py(f"import miraculix_amd as m; m.set_library_path({str(tmp_path / 'nope.so')!r}); m.load_shared_library()")
    assert r.returncode != 0 and "no cpu fallback" in r.stderr.lower()


def test_bed_reader_rejects_bad_magic(built, tmp_path):
    import miraculix_amd as m
    L = m.load_shared_library()
    p = tmp_path / "x.bed"
    p.write_bytes(bytes([0x6C, 0x1B, 0x00]) + bytes(10))          # individual-major flag: not supported, like the reference reader
    obj = ctypes.c_void_p(None)
    rc = L.mxa_bed2compressed(str(p).encode(), 5, 8, 1, ctypes.byref(obj), None, None, None)
    assert rc == 1 and not obj.value
    assert "magic" in (L.mxa_last_error_string() or b"").decode()


def test_shard_partition_covers_all_snps_at_multiples_of_4(built):
    """mxa_shard_bounds (the partition behind MIRACULIX_NUM_GPUS, mxa_multi.cpp): contiguous, disjoint, covering, boundaries at
    multiples of 4, empty blocks dropped, and identical to the Python-side rule (miraculix_amd.distributed.shard_bounds)"""
    import miraculix_amd as m
    from miraculix_amd.distributed import shard_bounds
    L = m.load_shared_library()
    for snps in (1, 3, 4, 10, 1003, 4096, 1_000_000, 5_000_001):
        for shards in (1, 2, 3, 7, 8, 64):
            b, e = ctypes.c_long(-1), ctypes.c_long(-1)
            cnt = L.mxa_shard_bounds(snps, shards, 0, ctypes.byref(b), ctypes.byref(e))
            assert 1 <= cnt <= shards
            pos = 0
            for g in range(cnt):
                assert L.mxa_shard_bounds(snps, shards, g, ctypes.byref(b), ctypes.byref(e)) == cnt
                assert b.value == pos and e.value > b.value and b.value % 4 == 0
                assert (b.value, e.value) == shard_bounds(snps, shards, g)
                pos = e.value
            assert pos == snps
            for g in range(cnt, shards):
                pb, pe = shard_bounds(snps, shards, g)
                assert pb == pe


def test_peeled_plan_can_need_more_partials_than_the_unpeeled_one(built):
    """The partial-sum workspace is sized for the plan of n columns, but a product that peels its odd columns into the exact int8 route launches
    the fp64 MFMA kernel with the plan of the REMAINING columns -- another tile width and other K pieces -- which may write more partial sums
    (advisor finding, round 3: 100 000 x 30 000, n = 9..11).  The planner runs without a device (256 CUs assumed): document that such shapes
    exist, so that the growth check in gemm_device (ensure_partials) is known to be load-bearing; the GPU test
    tests/test_workspace_gpu.py runs one of them."""
    L = ctypes.CDLL(built)
    L.mxa_plan_partial_doubles.restype = ctypes.c_long
    L.mxa_plan_partial_doubles.argtypes = [ctypes.c_long, ctypes.c_long, ctypes.c_int]
    worse = []
    for m, k in [(30_000, 100_000), (100_000, 30_000), (12_000, 50_000), (50_000, 12_000), (2050, 777), (50_000, 1_000_000)]:
        for n in range(5, 132):
            if n % 4 == 0:
                continue
            full, peeled = L.mxa_plan_partial_doubles(m, k, n), L.mxa_plan_partial_doubles(m, k, n - n % 4)
            assert full > 0 and peeled > 0
            if peeled > full:
                worse.append((m, k, n, full, peeled))
    assert any(m == 30_000 and k == 100_000 and n == 10 for m, k, n, _, _ in worse), worse[:5]


def test_planner_sweep_host_build():
    """the host-only planners (miraculix_amd/csrc/mxa_plan.h: tiles, K splits with the tapered tail, SNP shards) swept over shapes by a plain g++
    build of tests/host/plan_sweep.cpp; the same sweep runs under ASan + UBSan in tools/run_sanitizers.sh (profiles/r05_sanitizers.txt)"""
    d = os.path.join(ROOT, "tests", "host")
    subprocess.check_call(["make", "-C", d, "plan_sweep"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(d, "plan_sweep")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 failures" in r.stdout, r.stdout[-2000:]


def test_kernels_with_hand_counted_vmcnt_have_no_scratch_traffic(built, tmp_path):
    """k_gemm_i8_tn waits for its LDS DMA with hand-written `s_waitcnt vmcnt(N)` that count only what the source issues.  A register the compiler spills in
    such a kernel is scratch traffic on the same counter: the compiler then adds its own waits, counted WITHOUT the DMA it cannot see, and one of them in the
    stage loop serialises the ring (round 6: digit registers that lived across the epilogue were spilled -- 84..136 bytes of scratch per lane -- and the
    two-tile instantiation ran 1.86 ms where 1.45 had been measured; nothing failed, only the clock).  The code object's metadata states what was spilled:
    at most 16 bytes of scratch per lane and 4 spilled registers are allowed here (two address registers of the prologue), none in the one-tile form."""
    obj = os.path.join(ROOT, "miraculix_amd", "lib", "mxa_gemm_i8.o")
    if not os.path.exists(obj):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "miraculix_amd", "csrc"), "-j4"])
    llvm = "/opt/rocm/lib/llvm/bin"
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.check_call([f"{llvm}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([f"{llvm}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}", "--unbundle"])
    notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    seen = 0
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        if "k_gemm_i8_tn" not in name:
            continue
        seen += 1
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1))
        spilled = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", block).group(1))
        one_tile = "ILi4ELi1E" in name
        assert scratch <= (0 if one_tile else 16) and spilled <= (0 if one_tile else 4), (name, scratch, spilled)
    assert seen == 2


def test_reference_fortran_programs_bind_only_symbols_this_library_exports(built):
    """oracle/Makefile.ref_fortran links the reference's own (unmodified) Fortran tests and benchmark harness against libmiraculix_amd.so: every symbol their
    binding modules leave undefined (bind(C) names of mod5codesapi.f90 / modmiraculix_gpu.f90) must be one this library exports -- the link itself proves
    it; here the list is written out.  (No compute: the programs run in tests/test_reference_fortran_gpu.py.)  Skipped where they are not built."""
    bindir = os.path.join(ROOT, "oracle", "_ref", "fortran")
    progs = [os.path.join(bindir, p) for p in ("test_5codesapi.out", "test_5codesapi_t.out", "test_solve.out", "benchmark.out")]
    if not all(os.path.exists(p) for p in progs):
        pytest.skip("oracle/_ref/fortran not built (needs /root/reference and flang)")
    exported = {ln.split()[-1].split("@")[0] for ln in subprocess.check_output(["nm", "-D", "--defined-only", LIB], text=True).splitlines() if ln.strip()}
    api = {"setOptions_compressed", "plink2compressed", "dgemm_compressed", "free_compressed", "get_compressed_freq", "sparse_times_plink",
           "sparse2gpu", "dcsrtrsv_solve_gpu", "potrs_solve_gpu", "free_sparse_gpu", "snp_multiply_gpu"}
    bound = set()
    for p in progs:
        needed = subprocess.check_output(["readelf", "-d", p], text=True)
        assert "libmiraculix_amd.so" in needed
        undef = {ln.split()[-1].split("@")[0] for ln in subprocess.check_output(["nm", "-D", "--undefined-only", p], text=True).splitlines() if ln.strip()}
        bound |= undef & api
        assert (undef & api) <= exported
    assert {"setOptions_compressed", "plink2compressed", "dgemm_compressed", "free_compressed", "sparse2gpu", "dcsrtrsv_solve_gpu", "free_sparse_gpu"} <= bound


def test_rccl_bindings_are_typed_by_the_image_header_and_checked_at_compile_time(tmp_path):
    """mxa_multi.cpp binds RCCL with dlopen / dlsym (no link dependency).  Its pointers are typed by <rccl/rccl.h> (csrc/mxa_rccl.h: decltype(&ncclReduce), ...) and
    static_asserts pin the enumerators that cross the boundary (ncclFloat64 = 8, ncclSum = 0): the header of THIS image compiles, a differing expectation does not --
    i.e. a ROCm whose rccl.h changes a prototype or an enumerator stops the build instead of corrupting a reduction."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("hipcc / rccl.h not in this image")
    src = tmp_path / "rccl_check.cpp"
    src.write_text('#include "mxa_rccl.h"\nint main() { return mxa::rccl().ok ? 0 : 1; }\n')
    base = [hipcc, "-x", "hip", "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "miraculix_amd", "csrc"), str(src)]
    ok = subprocess.run(base, capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0 and "error" not in ok.stderr, ok.stderr[-2000:]
    for wrong in ("-DMXA_RCCL_EXPECT_FLOAT64=7", "-DMXA_RCCL_EXPECT_SUM=1"):
        bad = subprocess.run(base + [wrong], capture_output=True, text=True, timeout=300)
        assert "enumerator values differ" in bad.stderr, bad.stderr[-2000:]
    # and the product library does not link RCCL (bound at run time only)
    so = os.path.join(ROOT, "miraculix_amd", "lib", "libmiraculix_amd.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "librccl" not in needed and "libamdhip64" in needed
    multi = open(os.path.join(ROOT, "miraculix_amd", "csrc", "mxa_multi.cpp")).read()
    assert "ncclFloat64, ncclSum" in multi and "kNcclFloat64" not in multi          # the call site uses the header's enumerators, no magic numbers


def test_bench_compact_line_fits_the_driver_record():
    """bench.py prints ONE line that the driver keeps whole: compact_line() of the committed detail file of the last GPU run stays below 4 KB, carries the contract's keys,
    the roofline, both CPU baselines, the in-run oracle check and one key per BASELINE config leg"""
    import json
    sys.path.insert(0, ROOT)
    import bench
    detail = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_detail_n1.json")))
    line = bench.compact_line(detail)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= 4096, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline",
              "cpu_baseline_port", "check", "world_size", "rccl_ranks", "c3_ms_i8", "c3_frac_i8", "c3_ms_f4", "c3_frac_f4", "c5_step_ms", "c5_TBps", "c4_full_TFLOPs", "detail"):
        assert k in line, k
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-3
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and line["cpu_baseline_port"]["kind"] == "port"
    assert line["check"]["oracle_T_max_rel_err"] <= 1e-11 and line["check"]["oracle_N_max_rel_err"] <= 1e-11 and "model" not in line["config"]
