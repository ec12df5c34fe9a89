"""A compiled C caller (examples/c_driver.c: the reference's Fortran integration test restated in C, BASELINE config 1)
linked against the shared library only."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_driver(tmp_path):
    exe = str(tmp_path / "c_driver")
    lib = os.path.join(ROOT, "miraculix_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_driver.c"), "-o", exe,
                           "-L" + lib, "-lmiraculix_amd", "-Wl,-rpath," + lib, "-lm"])
    r = subprocess.run([exe, "1000", "500"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_driver ok" in r.stdout
    # the same binary, unchanged, with the SNPs sharded over three (virtual) devices behind the same symbols, and with the crossproduct in panels
    r = subprocess.run([exe, "1000", "500"], capture_output=True, text=True, timeout=300, env=dict(os.environ, MIRACULIX_NUM_GPUS="3"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_driver ok" in r.stdout
    # PRINT_LEVEL > 0 (reference: cuda_utils.cu:44-81): the compile banner once per process, the device line, the timings of debug_info
    r = subprocess.run([exe, "1000", "500"], capture_output=True, text=True, timeout=300, env=dict(os.environ, PRINT_LEVEL="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("Compiled on") == 1 and "git commit" in r.stdout and "using device" in r.stdout
    quiet = subprocess.run([exe, "1000", "500"], capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k != "PRINT_LEVEL"})
    assert "Compiled on" not in quiet.stdout
