"""The class tables and value codes that Y21 / Y22 / Y23 run from on the GPU (nhwcodec_amd/csrc/nhw_residual_rules.h, the text the HIP kernels
compile) compiled for the host and walked over their whole domains against the comparison chains they stand for (reference
encoder/nhw_encoder.c:970-1420): every (residual, next residual, third) triple's kind, every rule's effect on every LH1 coefficient,
Y23's (cell, coefficient) for every residual x coefficient x coefficient-before at every quality 13..23, Y21's cell rule for every
(cell, left, right) incl. the values the rule itself produces, and its four-cells-at-once "does anything fire" test.  No GPU needed."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tables_equal_the_chains(tmp_path):
    exe = str(tmp_path / "rules_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "residual_rules", "check.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "mismatches 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
