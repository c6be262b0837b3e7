"""All golden cases through the host build of the kernels (TEST infrastructure), bar of the GPU suite (slack 1).
usage: python tools/experiments/emul_all_goldens.py [-DFLAG ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests import emul, fixture_io
from tests.conftest import golden_files
from tests.test_hip_parity import REFERENCE_SINGULAR
emul.build(force=True, extra_flags=[a for a in sys.argv[1:] if a.startswith("-D")])
nbad = 0
closer = farther = 0
t0 = time.time()
for path in golden_files():
    name = os.path.basename(path)[:-4]
    case, expected = fixture_io.load_case(path)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    n = max(cfg.n_vegetation_region_forest, cfg.n_vegetation_region_urban) + 1
    out, nfail = emul.run_emul(cfg, cp, sw, lw, want_profiles=case["want_profiles"], do_simple_spectrum=case["do_simple_spectrum"])
    noise = dict(case["ref_noise"])
    singular = name in REFERENCE_SINGULAR
    if singular:
        lim, bar = REFERENCE_SINGULAR[name]
        expected = {k: v for k, v in expected.items() if noise.get(k, 0.0) <= lim}
        noise = {k: bar / fixture_io.NOISE_FACTOR for k in noise}
    rep = {}
    bad = fixture_io.parity_violations(fixture_io.flatten_outputs(out), expected, noise, truth=None if singular else case["ref_truth"],
                                       noise_arr=None if singular else case["ref_noise_arr"], collect=rep, label="x")
    for r in rep["x"].values():
        if r.get("gpu_vs_truth") is not None and r.get("ref_vs_truth") is not None:
            if r["gpu_vs_truth"] <= r["ref_vs_truth"]: closer += 1
            else: farther += 1
    if bad or nfail:
        nbad += 1
        print("FAIL %-44s nfail %d %s" % (name, nfail, [(b[0], "%.2e" % b[1], "%.2e" % b[2]) for b in bad[:3]]))
print("%d cases, %d failing, %.0f s; variables with binary128 data: emulated kernels closer to exact than the reference in %d, farther in %d" % (len(golden_files()), nbad, time.time() - t0, closer, farther))
