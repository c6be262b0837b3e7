"""The instruction the kernels of the orders 7..16 are built on, checked on the box the tests run on:
v_fmac_f64_dpp with row_newbcast (broadcast of one lane of a 16-lane row fused into a double-precision
multiply-add).  tools/experiments/dpp_probe (built by __graft_entry__.build()) multiplies 12 x 12
matrices whose rows sit on the lanes and compares with a host FMA chain: exit code 0 = bit-identical."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PROBE = os.path.join(HERE, "..", "tools", "experiments", "dpp_probe")


@pytest.mark.gpu
def test_dpp_row_newbcast_fma_is_exact_and_full_rate():
    if not os.path.exists(PROBE):
        pytest.fail("tools/experiments/dpp_probe is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([PROBE, "8192", "64"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "max abs error vs host fma chain 0.000e+00" in r.stdout, r.stdout
    # issue rate: within a factor of 1.5 of a plain v_fma_f64 (measured 6.5 against 5.9 cycles)
    rates = {}
    for line in r.stdout.splitlines():
        if "cycles per wave-FMA" in line:
            name = line.split(":")[0].strip()
            rates[name] = min(rates.get(name, 1e9), float(line.split(" ms,")[1].split("cycles")[0]))
    assert rates["dpp"] <= 1.5 * rates["plain"], rates
