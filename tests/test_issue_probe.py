"""What one SIMD of the box the tests run on sustains (tools/experiments/issue_probe, built by
__graft_entry__.build()): the facts DESIGN.md section 3 "What one SIMD sustains" prices the fused kernels with.
One wave alone on a SIMD reaches about half the nominal fp64 peak on a stream of independent v_fma_f64, a second
wave adds a good part of the rest, and with two waves scalar instructions issue beside the other wave's FMAs."""
import os
import re
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PROBE = os.path.join(HERE, "..", "tools", "experiments", "issue_probe")


@pytest.mark.gpu
def test_one_wave_per_simd_cannot_reach_the_fp64_peak():
    if not os.path.exists(PROBE):
        pytest.fail("tools/experiments/issue_probe is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([PROBE, "1500"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "Memory access fault" not in r.stdout + r.stderr, r.stdout + r.stderr
    rates = {}
    for line in r.stdout.splitlines():
        tf = [float(x) for x in re.findall(r"([0-9.]+) TF", line)]
        if len(tf) == 4:
            rates[line[:28].strip()] = tf          # waves per SIMD 1, 2, 4, 8
    fma = rates["v_fma_f64 only"]
    assert 25.0 < fma[0] < 55.0, fma                # measured 42: about half of 78.6
    assert fma[1] > 1.2 * fma[0] and max(fma) < 78.6, fma
    # a scalar instruction per FMA: costs a full slot to a lone wave, (nearly) nothing beside a second wave
    nop = rates["fma + s_nop 0"]
    assert nop[0] < 0.9 * fma[0] and nop[1] > 0.85 * fma[1], (nop, fma)
