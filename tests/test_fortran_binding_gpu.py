"""The Fortran binding of the ADDITIVE entry points (miraculix_amd/bindings/fortran/modmiraculix_amd.f90) exercised by examples/fortran/gblup_cg.f90 on the GPU:
the library stages the .bed itself (mxa_bed2compressed), conjugate gradients on (Zc Zc^T + lambda I) x = y run on the fused step mxa_gram_matvec, and the program
verifies the solution through the REFERENCE entries (dgemm_compressed 'T' then 'N', the loop of examples/iterative_solver/grm_solve_cg.jl:74-84), checks that the
fused step equals its two products bit for bit and -- small data -- compares with a dense product on genotypes it decodes itself.  Built by
__graft_entry__.build() (make -C examples/fortran) where a Fortran compiler exists (AMD flang in this image); the binary travels to the GPU box."""
import os
import re
import subprocess

import pytest

from test_reference_fortran_gpu import _write_dataset

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "fortran", "gblup_cg.out")


@pytest.mark.parametrize("snps,indiv,dense", [(3001, 803, True), (60000, 4001, False)])
def test_fortran_cg_on_the_additive_entries(tmp_path, snps, indiv, dense):
    if not os.path.exists(EXE):
        pytest.skip(f"{EXE} not built (make -C examples/fortran needs a Fortran compiler)")
    _write_dataset(str(tmp_path / "geno"), snps, indiv, seed=3)
    p = subprocess.run([EXE, "geno.bed", "1.0", "200"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    out = p.stdout
    assert p.returncode == 0 and out.rstrip().endswith("PASS"), out[-3000:]
    assert f"object: {snps} SNPs x {indiv} individuals, packed copies kept: 1" in out
    it = int(re.search(r"CG: (\d+) iterations", out).group(1))
    assert 3 <= it < 200
    assert "kernel family of the last product (2 = exact int8 route): 2" in out      # n = 1: the HBM-bound route of the CG step
    assert "mxa_gram_matvec == dgemm_compressed T then N, bit for bit" in out
    assert float(re.search(r"true residual / \|y\| through dgemm_compressed T then N:\s*(\S+)", out).group(1)) <= 1e-8
    if dense:
        assert float(re.search(r"dense check of G x .*: max relative difference\s*(\S+)", out).group(1)) <= 1e-11
    else:
        assert "dense check" not in out
