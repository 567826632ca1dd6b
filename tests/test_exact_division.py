"""Evidence behind the three-instruction exact division of the phi / beta kernels (csrc/ammsb_dev.h):
tests/cpp/div_check.c, compiled without FMA contraction, checks on the host's IEEE binary32 arithmetic that
  * a Newton-refined reciprocal from a seed within 1 ulp is NOT always the correctly rounded 1/d (33 known
    misses over all 2^23 significands x 3 seeds) -- the reason the kernels take y from a real division;
  * with y = RN(1/d), q0 = x*y; r = fma(-d, q0, x); q = fma(r, y, q0) equals x / d on 5.4e8 operand pairs."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_three_instruction_division(tmp_path):
    exe = str(tmp_path / "div_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "cpp", "div_check.c"), "-lm"],
                   check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, check=True).stdout
    m = re.search(r"rcp misses (\d+) of (\d+) ; div misses (\d+) of (\d+)", out)
    assert m, out
    rcp_miss, rcp_n, div_miss, div_n = map(int, m.groups())
    assert rcp_n == 3 * (1 << 23) and 0 < rcp_miss < 100   # refinement alone is not enough
    assert div_miss == 0 and div_n >= 500_000_000
