"""The reference's OWN Fortran acceptance tests against this library (drop-in check in the reference's host language).

oracle/Makefile.ref_fortran compiles, UNMODIFIED and from where they lie under /root/reference, the reference's Fortran binding modules
(src/bindings/Fortran/mod5codesapi.f90, modmiraculix_gpu.f90, modplink_miraculix.f90, modtestplink.f90) and its test programs
(tests/dgemm_compressed/test_5codesapi.f90, test_5codesapi_t.f90, tests/solve/test_solve.f90) with the image's
Fortran compiler (AMD flang) and links them against miraculix_amd/lib/libmiraculix_amd.so where the reference links libmiraculix + its CUDA library.
The binaries land in oracle/_ref/fortran/ (git-ignored build output that travels to the GPU box; a clean checkout without /root/reference has none: skipped).

What the programs check themselves (reference tolerances): dgemm_compressed 'n' / 't', ncol = 10, centred, three repetitions, against the compiler's
matmul on the decoded genotypes, |difference| <= 1e-4 absolute, `error stop 'Different outputs'` otherwise; the solver entries print "<case> OK".
(tests/sparse_plink/test_sparse_plink.f90 only prints, and its print-outs disagree on the reference's own CPU library -- one-based CSR arrays into a
zero-based C loop -- so it is no acceptance test; sparse_times_plink is pinned on the reference library's outputs in test_sparse_gpu.py.)  Inputs: synthetic .bed / .bim / .fam / .freq written by this file
(the reference's data/ directory is not part of its repository), no missing genotypes (the Fortran decode keeps a missing call as 3.0 in its dense check)."""
import os
import re
import subprocess

import pytest

from _util import make_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "fortran")


def _need(name):
    path = os.path.join(BIN, name)
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (oracle/Makefile.ref_fortran needs /root/reference and flang: build container only)")
    return path


def _write_dataset(base, snps, indiv, seed):
    """<base>.bed (magic 6c 1b 01, SNP-major), .bim / .fam (only their line counts are read), .freq (index, allele frequency = column mean / 2)"""
    import miraculix_amd as mx
    prob = make_problem(snps, indiv, 1, seed=seed)
    mx.read_plink.write_bed(base + ".bed", prob["plink"])
    with open(base + ".bim", "w") as fh:
        fh.write("".join(f"1 snp{i} 0 {i} A B\n" for i in range(snps)))
    with open(base + ".fam", "w") as fh:
        fh.write("".join(f"f{i} i{i} 0 0 0 -9\n" for i in range(indiv)))
    with open(base + ".freq", "w") as fh:
        fh.write("".join(f"{i + 1} {prob['f'][i]:.17g}\n" for i in range(snps)))
    return prob


def _run(cmd, cwd):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    p = subprocess.run(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    return p.returncode, p.stdout


@pytest.mark.parametrize("prog", ["test_5codesapi.out", "test_5codesapi_t.out"])
def test_reference_dgemm_compressed_tests_pass_on_this_library(tmp_path, prog):
    exe = _need(prog)
    _write_dataset(str(tmp_path / "small"), 5003, 1201, seed=5)
    rc, out = _run([exe, "small.bed", "small.freq"], str(tmp_path))   # relative names: the programs hold them in character(len=30) / (len=50) variables
    assert rc == 0, out[-3000:]
    assert "CUDA VERSION" in out                        # the GPU branch of the test (-DCUDA): use_gpu = 1, both packed copies handed over
    assert out.count("No MC error !") == 3, out[-3000:]  # one per repetition
    assert "Different outputs" not in out and "abs Difference" not in out
    assert "miraculix_amd - dgemm_compressed: using device" in out          # print_details = 2: the banner of THIS library


def test_reference_solver_test_reports_ok_for_every_case():
    exe = _need("test_solve.out")
    rc, out = _run([exe], BIN)
    assert rc == 0, out[-3000:]
    assert "wrong" not in out, out[-3000:]
    assert len(re.findall(r" OK\s*$", out, flags=re.M)) == 6, out[-3000:]   # U\B, U^T\B, U^T\U\B twice, L^T\L\B twice (without and with the permutation)


def test_reference_benchmark_harness_runs_in_gpu_mode(tmp_path):
    """utils/benchmark/benchmark.f90 (the timing protocol SURVEY.md 8d cites: 1 warm-up + 10 repetitions of 'n' and of 't', ncol = 10, centred, host B / C), mode GPU:
    runs to the end against this library and reports both averages (numbers at a size where they mean something: profiles/r05_reference_fortran_tests.txt)"""
    exe = _need("benchmark.out")
    _write_dataset(str(tmp_path / "small"), 5003, 1201, seed=6)
    rc, out = _run([exe, "GPU", "small.bed", "small.freq"], str(tmp_path))
    assert rc == 0, out[-3000:]
    assert out.count("Elapsed time - Z - GPU") == 10 and out.count("Elapsed time - Z^T - GPU") == 10, out[-3000:]
    for label in ("Average time - Z - GPU: ", "Average time - Z^T - GPU: "):
        m = re.search(re.escape(label) + r"(\S+)", out)
        assert m and 0.0 < float(m.group(1)) < 1.0, out[-3000:]
