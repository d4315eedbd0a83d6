"""CPU: the exhaustive proof behind `div255` in augcl_amd/csrc/aug_kernels.hip -- the 3-operation x/255 equals the
IEEE quotient for every non-negative finite float32 (tests/div255_check.c, ~2^31 inputs, a few seconds with OpenMP)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_div255_three_ops_equal_ieee_division_exhaustively(tmp_path):
    flags = open("/proc/cpuinfo").read()
    if " fma" not in flags:
        pytest.skip("host CPU has no FMA instruction: a software fmaf would take minutes")
    exe = str(tmp_path / "div255_check")
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", os.path.join(HERE, "div255_check.c"), "-o", exe, "-lm"],
                   check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip() == "0", f"{r.stdout.strip()} inputs where the 3-op form differs from x/255"
