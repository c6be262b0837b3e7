#!/usr/bin/env python3
"""bench.py - canopy columns/s of the SPARTACUS-Surface SW+LW column solver on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c2r3ns8|c3ns4|c4|c4ns4|rami5] [--scaling weak|strong]

A "step" is one pass of the hot path (radsurf: layer kernels + column kernels,
SW and LW) over one batch of synthetic columns that is already resident in HBM.
Default workload at every N: BASELINE.json configs[2] = C3 of SURVEY.md section 8(d)
(test/urban vegetated-urban column on 10 layers, 3 regions x 2 streams, 1 SW +
1 LW band, 10^6 jittered columns per GPU, seed 12345 + rank), weak scaling.
`--workload c4 --scaling strong` is BASELINE.json configs[3] as stated: 10^6 mixed
forest / urban columns with 14 SW + 16 LW bands IN TOTAL, sharded over the N GPUs.

With N > 1 every rank owns a contiguous column shard (one process per GPU, no
data-path collective).  The exchange step of the path is the FINAL gather of the
results: after the last step bc_out and every allocated member of the four flux
objects go from the plans' device memory to rank 0 over RCCL, device to device
(spartacus_surface_amd.distributed.DeviceGather), inside the timed region; the
four bc_out arrays (what a host atmosphere model needs every step) are gathered
after every step.

Rank 0 prints one JSON line with the throughput, the roofline of the dominant
kernel (HIP-event timing on the stream each kernel runs on, taken from steps with
strictly serial kernels so that the durations do not overlap) and the reference's
CPU path timed on the host cores of the same box.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6     # vector = matrix fp64 peak


def flops_per_column(case):
    """Leading-order flop model of the reference's operation sequence per
    (layer, band): F_sw = (74.4+8d)n^3 + 2/3(2n+d)^3 +
    30d^3 + 22n^2 + 6n^2 d, F_lw = 70.4 n^3 + 30 n^2."""
    cfg, cp = case["config"], case["canopy"]
    L = float(cp.ntotlay) / max(1, cp.ncol)
    tot = 0.0
    for urban, share in ((False, float(np.mean(cp.i_representation == 1))),
                         (True, float(np.mean(cp.i_representation != 1)))):
        if share == 0.0:
            continue
        d = (cfg.n_vegetation_region_urban if urban else cfg.n_vegetation_region_forest) + 1
        if cfg.do_sw:
            n = d * (cfg.n_stream_sw_urban if urban else cfg.n_stream_sw_forest)
            tot += share * cfg.nsw * L * ((74.4 + 8 * d) * n ** 3 + (2.0 / 3.0) * (2 * n + d) ** 3
                                          + 30 * d ** 3 + 22 * n ** 2 + 6 * n ** 2 * d)
        if cfg.do_lw:
            n = d * (cfg.n_stream_lw_urban if urban else cfg.n_stream_lw_forest)
            tot += share * cfg.nlw * L * (70.4 * n ** 3 + 30 * n ** 2)
    return tot


WORKLOAD_NAMES = {
    "c2": "C2: test/rami4pilps forest SW (vis-med-0.3), 2 regions x 1 stream x 2 layers, 1 band",
    "c2r3ns8": "C2 column set run as test/rami4pilps method '-2-8': forest SW, 3 regions x 8 streams (n = 24) x 2 layers, 1 band",
    "c3": "C3: test/urban vegetated-urban SW+LW, 3 regions x 2 streams x 10 layers, 1 SW + 1 LW band",
    "c3ns4": "C3 with 4 streams: vegetated-urban SW+LW, 3 regions x 4 streams x 10 layers (n = 12)",
    "c4": "C4: mixed forest / vegetated-urban SW+LW, 3 regions x 2 streams x 10 layers, 14 SW + 16 LW bands",
    "c4ns4": "C4 with 4 streams: mixed forest / vegetated-urban SW+LW, 3 regions x 4 streams x 10 layers (n = 12), "
             "14 SW + 16 LW bands",
    "rami5": "C5: test/rami5 scene HET09_JBS_SUM (41 deg, direct) replicated: vegetated tile, 3 regions x 4 streams "
             "(n = 12) x 62 layers, 14 SW bands, no LW",
}


def make_case(workload, ncol, seed):
    from spartacus_surface_amd import workloads
    if workload == "c2":
        return workloads.c2_rami4pilps(ncol=ncol)
    if workload == "c2r3ns8":
        return workloads.c2_rami4pilps(ncol=ncol, nstream=8, n_vegetation_region=2)
    if workload == "c3ns4":
        return workloads.c3_vegetated_urban(ncol=ncol, seed=seed, nstream=4)
    if workload == "c4":
        return workloads.c4_mixed(ncol=ncol, seed=seed)
    if workload == "c4ns4":
        return workloads.c4_mixed(ncol=ncol, seed=seed, nstream=4)
    if workload == "rami5":
        return workloads.rami5_replicated(ncol=ncol)
    return workloads.c3_vegetated_urban(ncol=ncol, seed=seed)


def cpu_baseline(workload, ncol_sample):
    """The REAL reference (oracle/_ref: Fortran compiled with flang -O2, OpenMP
    block loop of its driver, nblocksize 16, schedule dynamic) on a bounded
    sample of the same workload.  Falls back to the C port if _ref is absent."""
    from spartacus_surface_amd import workloads
    cores = min(16, os.cpu_count() or 1)
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    os.environ.setdefault("OMP_SCHEDULE", "dynamic")
    from oracle import checker
    case = make_case(workload, ncol_sample, 12345)
    a = (case["config"], case["canopy"], case["sw"], case["lw"])
    if checker.have_ref():
        ref_out, sec = checker.run_ref(*a, want_profiles=False, nblocksize=16, nrepeat=1)
        kind = "reference"
    else:
        ref_out, sec = checker.run_oracle(*a, want_profiles=False, nthreads=cores)
        kind = "port"
    base = {"value": ncol_sample / sec, "unit": "columns/s", "cores": cores, "kind": kind,
            "sample": "%d columns of the bench workload (%s), %.1f s, %s"
                      % (ncol_sample, workload, sec, "flang -O2 OpenMP dynamic nblocksize=16"
                         if kind == "reference" else "gcc -O2 OpenMP C port")}
    return base, parity_of_sample(case, ref_out, kind)


def parity_of_sample(case, ref_out, kind, ntruth=2048):
    """The second half of BASELINE.json's metric: max relative |flux difference| between
    the GPU path and the CPU checker on the sample the CPU baseline just computed (every
    member of bc_out and of the four flux objects; error metric of tests/fixture_io.py),
    and - on the first ntruth columns plus the column of every worst element - the
    distance of BOTH from exact arithmetic (the same algorithm in binary128)."""
    from spartacus_surface_amd import radsurf, workloads
    from oracle import checker
    from tests import fixture_io
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    plan = radsurf.Plan(cfg, cp, sw, lw, want_profiles=False)
    plan.solve()
    out = checker.alloc_outputs(cfg, cp, False)
    plan.download(out["bc_out"], out["sw_norm_dir"], out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
    plan.close()
    g, r = fixture_io.flatten_outputs(out), fixture_io.flatten_outputs(ref_out)
    errs = fixture_io.rel_errors(g, r)
    worst = max(errs, key=errs.get)
    over = {k: v for k, v in errs.items() if v > 1.0e-5}
    res = {"max_rel_error": errs[worst], "where": worst, "against": kind, "columns": int(cp.ncol),
           "variables": len(errs), "above_1e-5": over,
           "note": "relative to max(|ref|, 1e-3 max|ref| of the variable).  The GPU path is MORE ACCURATE THAN AND "
                   "NOT IDENTICAL TO the reference: it evaluates the same equations with other algebra "
                   "(symmetrised coordinates, Cholesky, symmetric eigen-solvers), so where the reference's own fp64 "
                   "rounding error exceeds 1e-5 the two differ by more than 1e-5 ('above_1e-5').  'vs_exact' "
                   "compares both with the binary128 evaluation of the reference's algorithm on a subset of the "
                   "sample; the CPU restatement oracle/spartacus_oracle.c, which keeps the reference's operation "
                   "order, matches it bit for bit.  Parity bar of the test-suite: tests/fixture_io.py (DESIGN.md "
                   "section 2)"}
    # distance from exact arithmetic: first ntruth columns of the sample
    nt = int(min(ntruth, cp.ncol))
    head = workloads.head_columns(case, nt)
    a = (head["config"], head["canopy"], head["sw"], head["lw"])
    t = fixture_io.flatten_outputs(checker.run_oracle(*a, want_profiles=False, quad=True,
                                                      nthreads=min(16, os.cpu_count() or 1))[0])
    nl = int(head["canopy"].ntotlay)
    cut = lambda x: x[:nt] if x.shape[0] == cp.ncol else x[:nl]   # noqa: E731
    gh, rh = {k: cut(v) for k, v in g.items()}, {k: cut(v) for k, v in r.items()}
    e_gpu = fixture_io.rel_errors({k: gh[k] - t[k] + rh[k] for k in rh}, rh)
    e_ref = fixture_io.rel_errors(t, rh)
    e_gr = fixture_io.rel_errors(gh, rh)
    wk = max(e_gpu, key=e_gpu.get)
    res["vs_exact"] = {"columns": nt, "gpu_vs_exact_max": e_gpu[wk], "gpu_vs_exact_where": wk,
                       "reference_vs_exact_max": max(e_ref.values()),
                       "reference_vs_exact_where": max(e_ref, key=e_ref.get),
                       "gpu_vs_reference_max": max(e_gr.values()),
                       "variables_where_gpu_is_farther_from_exact_than_reference_and_1e-5":
                           sorted(k for k in e_gpu if e_gpu[k] > max(1.0e-5, e_ref[k]))}
    if over:
        # where is the worst element of each such variable, and how far are GPU and
        # reference from exact arithmetic in that very column?
        detail = {}
        for k in over:
            den = np.maximum(np.abs(r[k]), 1.0e-3 * float(np.max(np.abs(r[k]))))
            e = np.abs(g[k] - r[k]) / den
            idx = np.unravel_index(int(np.argmax(e)), e.shape)
            per_col = r[k].shape[0] == cp.ncol
            col = int(idx[0]) if per_col else int(np.searchsorted(cp.istartlay - 1, idx[0], side="right") - 1)
            f128 = fixture_io.flatten_outputs(checker.run_oracle(cfg, cp, sw, lw, want_profiles=False,
                                                                 istartcol=col + 1, iendcol=col + 1, quad=True)[0])
            detail[k] = {"column": col, "error": float(e[idx]),
                         "reference_noise_there": float(abs(r[k][idx] - f128[k][idx]) / den[idx]),
                         "gpu_vs_exact_there": float(abs(g[k][idx] - f128[k][idx]) / den[idx])}
        res["worst_elements"] = detail
    return res


def library_build_id():
    from spartacus_surface_amd import radsurf
    return radsurf.build_id()


def pmc_summary(workload):
    """The committed PMC summary of the workload (profiles/traffic.json for the headline one,
    profiles/traffic_c3ns4.json for C3 with 4 streams; written by
    tools/pmc_to_profile.py from separate rocprofv3 --pmc passes; FETCH_SIZE doubled per
    MI355X_MICROARCH.md): bytes moved between L2 and the fabric and executed fp64 flops,
    per column and kernel family.  The file is stamped with the hash of the library it was measured on:
    counters of ANOTHER build are not used (returns (None, reason): the traffic- and flop-derived fractions
    of the line are then null instead of silently stale)."""
    name = {"c3": "traffic.json", "c3ns4": "traffic_c3ns4.json"}.get(workload)
    if name is None:
        return None, "no committed PMC summary for this workload"
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            t = json.load(f)
        t["bytes_per_column"], t["fp64_flops_per_column_executed"]     # noqa: B018
    except (OSError, KeyError, ValueError):
        return None, "profiles/%s missing or unreadable" % name
    have, want = t.get("library_build_id"), library_build_id()
    if have != want and os.environ.get("SPARTACUS_BENCH_ANY_PMC") != "1":
        return None, ("profiles/%s was measured on library build %s, the loaded library is %s: re-run "
                      "tools/profile_round.sh + tools/pmc_to_profile.py (SPARTACUS_BENCH_ANY_PMC=1 overrides)"
                      % (name, have, want))
    return t, None


def fp64_figure(flops_model, ncol, ms_per_step, pmc):
    """fp64 rate of the whole step (wall time of the timed region: with several
    streams the per-kernel durations overlap and must not be added up): in the
    operation count of the reference's algorithm (SURVEY 8d model) and, where the
    PMC summary is committed, in the flops the kernels actually execute."""
    ach = flops_model * ncol / (ms_per_step * 1e-3) / 1e12
    out = {"bound": "fp64-valu", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": ach / FP64_PEAK_TFLOPS, "flops_per_column_model": flops_model,
           "note": "model flops of the reference's operation sequence per column x columns / step time"}
    if pmc:
        ex = sum(pmc["fp64_flops_per_column_executed"].values())
        out["flops_per_column_executed"] = ex
        out["achieved_executed"] = ex * ncol / (ms_per_step * 1e-3) / 1e12
        out["frac_executed"] = out["achieved_executed"] / FP64_PEAK_TFLOPS
    return out


def roofline_block(fam_ms, fam_n, nsteps, columns, case, pmc, evidence, ms_per_step, fpc):
    """Roofline of the dominant kernel family from NON-overlapping durations: avg launch
    duration from HIP events of steps whose kernels ran strictly one after the other.
    The binding resource of this path is fp64 issue (57-77 flop per
    algorithmic byte against a machine balance of ~10), so the block is priced against the
    fp64 peak - vector and matrix fp64 peaks are the same number on MI355X - with the executed
    flops of that kernel from the committed PMC summary; the HBM view asked for by north_star
    (algorithmic bytes of the kernel's OWN spectral region, measured record traffic) sits in "hbm"."""
    from spartacus_surface_amd import workloads
    dom = max(fam_ms, key=lambda k: fam_ms[k])
    avg_ms = fam_ms[dom] / max(1, fam_n[dom])
    cols_per_launch = columns * nsteps / max(1, fam_n[dom])
    spectral = dom.split("_")[0]
    b_in, b_out = workloads.algorithmic_bytes_per_column(case, family=spectral)
    bpc = b_in + b_out
    hbm_ach = bpc * cols_per_launch / (avg_ms * 1e-3) / 1e9
    rl = {"bound": "fp64", "kernel": dom, "achieved": None, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
          "frac": None, "traffic": None, "evidence": evidence,
          "avg_launch_ms": avg_ms, "columns_per_launch": cols_per_launch,
          "launches_per_step": fam_n[dom] / nsteps,
          "serial_ms_per_step_all_kernels": sum(fam_ms.values()) / nsteps,
          "peak_note": "fp64 vector = fp64 matrix peak of MI355X; the kernels are scalar-per-lane fp64 FMA code "
                       "(6 x 6 problems, one per lane): MFMA does not apply (DESIGN.md section 3)",
          # what this chip SUSTAINS on streams of independent v_fma_f64 (tools/experiments/issue_probe.hip,
          # profiles/round4_issue_probe.txt; context for "frac", which stays priced against the nominal peak):
          # one wave per SIMD - what 512-register kernels like these run at - issues ANY instruction,
          # scalar ones included, every ~4.8 cycles of a clock that sits at ~2.2 GHz under this load
          "sustained_fma_only_TFLOPs": {"1_wave_per_simd": 42.1, "2": 62.7, "4": 65.1, "8": 69.3,
                                        "source": "profiles/round4_issue_probe.txt"},
          "hbm": {"algorithmic_bytes_per_column": bpc, "of": "%s inputs + outputs only" % spectral.upper(),
                  "achieved": hbm_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_ach / HBM_PEAK_GBS,
                  "note": "intensity ~%.0f flop per algorithmic byte of the whole step: the path cannot be "
                          "HBM-bound on them; its own intermediate records are what loads the memory system "
                          "('traffic')" % (fpc / sum(workloads.algorithmic_bytes_per_column(case)))}}
    if pmc:
        tb = pmc["bytes_per_column"]
        rl["traffic"] = tb.get(dom, 0.0) * cols_per_launch
        rl["traffic_source"] = pmc.get("source")
        rl["hbm"]["traffic_bytes_per_column_total"] = sum(tb.values())
        rl["hbm"]["traffic_over_algorithmic_whole_step"] = sum(tb.values()) / sum(workloads.algorithmic_bytes_per_column(case))
        # all bytes the kernels of a step move through L2 <-> fabric / the wall time of a step / HBM peak
        rl["hbm"]["traffic_frac_whole_step"] = sum(tb.values()) * columns / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS
        rl["hbm"]["traffic_frac_of_dominant_kernel"] = rl["traffic"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        rl["hbm"]["traffic_note"] = pmc.get("note")
        fl = pmc["fp64_flops_per_column_executed"].get(dom)
        if fl:
            rl["achieved"] = fl * cols_per_launch / (avg_ms * 1e-3) / 1e12
            rl["frac"] = rl["achieved"] / FP64_PEAK_TFLOPS
            rl["flops_per_column_executed_by_this_kernel"] = fl
        # every kernel family of the step against both ceilings (serial durations; traffic and flops of the
        # committed PMC summary): the layer kernels sit at the fp64 issue ceiling, the sweeps at the HBM one
        table = {}
        for k in fam_ms:
            if fam_n[k] == 0 or fam_ms[k] <= 0.0:
                continue
            sec = fam_ms[k] / nsteps * 1e-3
            f64 = pmc["fp64_flops_per_column_executed"].get(k, 0.0) * columns / sec / 1e12
            hbm = tb.get(k, 0.0) * columns / sec / 1e9
            table[k] = {"ms_per_step": 1e3 * sec, "fp64_TFLOPs": f64, "fp64_frac": f64 / FP64_PEAK_TFLOPS,
                        "traffic_GBs": hbm, "hbm_frac": hbm / HBM_PEAK_GBS,
                        "nearer_ceiling": "fp64" if f64 / FP64_PEAK_TFLOPS >= hbm / HBM_PEAK_GBS else "hbm"}
        rl["kernels"] = table
    return rl


def other_workload(workload, ncol, steps, ncheck):
    """One of the other BASELINE configurations inside the default run: `steps` timed steps on `ncol`
    columns resident in HBM (after one warm-up step), and the GPU results of the first `ncheck` columns
    against the CPU oracle (fp64 and binary128)."""
    from spartacus_surface_amd import radsurf, workloads
    from oracle import checker
    from tests import fixture_io
    import torch
    case = make_case(workload, ncol, 12345)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    plan = radsurf.Plan(cfg, cp, sw, lw, want_profiles=False)
    plan.solve()
    plan.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.solve()
    plan.synchronize()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / steps
    out = checker.alloc_outputs(cfg, cp, False)
    plan.download(out["bc_out"], out["sw_norm_dir"], out["sw_norm_diff"], out["lw_internal"], out["lw_norm"])
    nfail = plan.eigen_failures()
    plan.close()
    g = fixture_io.flatten_outputs(out)
    head = workloads.head_columns(case, ncheck)
    a = (head["config"], head["canopy"], head["sw"], head["lw"])
    cores = min(16, os.cpu_count() or 1)
    r = fixture_io.flatten_outputs(checker.run_oracle(*a, want_profiles=False, nthreads=cores)[0])
    t = fixture_io.flatten_outputs(checker.run_oracle(*a, want_profiles=False, nthreads=cores, quad=True)[0])
    nl = int(head["canopy"].ntotlay)
    gh = {k: (v[:ncheck] if v.shape[0] == cp.ncol else v[:nl]) for k, v in g.items()}
    e_gr = fixture_io.rel_errors(gh, r)
    e_gt = fixture_io.rel_errors({k: gh[k] - t[k] + r[k] for k in r}, r)
    e_rt = fixture_io.rel_errors(t, r)
    fpc = flops_per_column(case)
    return {"workload": WORKLOAD_NAMES[workload], "columns": ncol, "steps": steps,
            "value": ncol / sec, "unit": "columns/s", "ms_per_step": 1e3 * sec,
            "fp64_TFLOPs_reference_operation_count": fpc * ncol / sec / 1e12, "eigen_failures": nfail,
            "parity": {"columns": ncheck, "against": "oracle (C restatement, bit-identical to the reference on its test inputs)",
                       "max_rel_error": max(e_gr.values()), "where": max(e_gr, key=e_gr.get),
                       "gpu_vs_exact_max": max(e_gt.values()), "reference_vs_exact_max": max(e_rt.values())}}


def input_refresh_figure(case, steps):
    """The shape of the reference driver's timed loop (driver/spartacus_surface_driver.F90:195-265): every
    repeat derives the longwave emission / Planck arrays from the temperatures (calc_simple_spectrum_lw,
    :221-225) and calls radsurf on inputs it hands over again.  Two figures on a plan of its own, the same
    columns with the temperatures the workload's Planck values were made from:
      device_refresh: per step the device spectrum (ss_plan_calc_simple_spectrum_lw), the re-pack of the
                      longwave inputs it invalidates (k_pack) and the solve - inputs stay resident;
      host_refresh:   per step also ss_plan_upload of every input array from the host (PCIe-inclusive:
                      never the headline value), which re-packs both spectral regions."""
    import copy
    import torch
    from spartacus_surface_amd import radsurf
    cfg, cp, sw, lw = case["config"], copy.deepcopy(case["canopy"]), case["sw"], copy.deepcopy(case["lw"])
    t_air, t_surf = float(np.float32(278.15)), float(np.float32(283.15))
    cp.ground_temperature = np.full(cp.ncol, t_surf)
    for name, t in (("roof_temperature", t_surf), ("wall_temperature", t_surf), ("clear_air_temperature", t_air),
                    ("veg_temperature", t_air), ("veg_air_temperature", t_air)):
        setattr(cp, name, np.full(cp.ntotlay, t))
    plan = radsurf.Plan(cfg, cp, sw, lw, want_profiles=False)
    out = {}
    for label, upload in (("device_refresh", False), ("host_refresh", True)):
        def one():
            if upload:
                plan.upload()
            plan.calc_simple_spectrum_lw()
            plan.solve()
        one()
        plan.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        plan.synchronize()
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / steps
        out[label] = {"value": cp.ncol / sec, "unit": "columns/s", "ms_per_step": 1e3 * sec, "steps": steps}
    plan.close()
    out["note"] = ("per step: calc_simple_spectrum_lw on the device + re-pack of the inputs it changes + solve "
                   "(device_refresh); the same after re-uploading every input array from pageable host memory "
                   "(host_refresh, PCIe-inclusive).  The headline value times the solve on unchanged resident inputs.")
    return out


def scale_sum_rate(plan, case, ncol):
    """SURVEY 8f row 1 on the device: canopy_flux%scale / %sum over the resident
    results (one streaming kernel, HBM-bound); returns its HIP-event time and the
    bytes it has to move (reads + writes of every member touched)."""
    from spartacus_surface_amd import abi, types
    cfg, cp = case["config"], case["canopy"]
    plan.solve()
    plan.scale_and_sum(case.get("top_flux_dn_sw"), case.get("top_flux_dn_direct_sw"),
                       case.get("top_flux_dn_lw"))
    plan.synchronize()
    ms = plan.kernel_ms("misc")
    nbytes = 0
    for nspec, direct, scaled_a, on in ((cfg.nsw, True, 1, cfg.do_sw), (cfg.nlw, False, 0, cfg.do_lw)):
        if not on:
            continue
        f = types.CanopyFlux()
        f.allocate(cfg, 1, 1, nspec, use_direct=direct, do_save_flux_profile=False)
        for n, kind in abi.FLUX_FIELDS:
            if getattr(f, n) is None:
                continue
            per = {"sc": nspec * cp.ncol, "sl": nspec * cp.ntotlay, "c": cp.ncol, "l": cp.ntotlay}[kind]
            writes = 1 + ((scaled_a + 1) if kind in ("sc", "sl") else 0)
            nbytes += 8 * per * (2 + writes)
    return {"ms": ms, "bytes": nbytes, "GB/s": nbytes / (ms * 1e-3) / 1e9 if ms > 0 else None,
            "hbm_frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None,
            "note": "canopy_flux%scale + %sum of all four flux objects on the device (next row f1)"}


def host_entry_rate(ncol):
    """PCIe-inclusive rate of the host entry point (spartacus_radsurf_hip: host
    buffers in, host buffers out, one call for all columns) on a bounded sample.
    Reported next to the headline value, never as the value.  Two figures: the
    caller's arrays pageable (default) and page-locked once by the library
    (SPARTACUS_PIN_HOST=1: what a host model that calls every time step with the
    same arrays gets from its second call on; the first call pays the locking)."""
    from spartacus_surface_amd import radsurf, workloads, types
    case = make_case("c3", ncol, 999)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    bc = types.BoundaryCondsOut()
    bc.allocate(cp.ncol, cfg.nsw, cfg.nlw)
    fl = []
    for nspec, direct in ((cfg.nsw, True), (cfg.nsw, False), (cfg.nlw, False), (cfg.nlw, False)):
        f = types.CanopyFlux()
        f.allocate(cfg, cp.ncol, cp.ntotlay, nspec, use_direct=direct, do_save_flux_profile=False)
        fl.append(f)

    def best_of(n):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            radsurf.radsurf(cfg, cp, sw, lw, bc, 1, cp.ncol, *fl)
            ts.append(time.perf_counter() - t0)
        return ts

    bytes_in, bytes_out = workloads.algorithmic_bytes_per_column(case)
    # default: the entry page-locks the slices of a large call for the duration of the call (call-owned locks)
    os.environ.pop("SPARTACUS_PIN_HOST", None)
    t_def = best_of(4)
    # no locking at all: every copy is staged by the runtime and blocks the calling thread
    os.environ["SPARTACUS_PIN_HOST"] = "0"
    t_page = best_of(3)
    os.environ.pop("SPARTACUS_PIN_HOST", None)
    # the caller pins its arrays once (ss_host_pin) and unpins them before they are freed: the contract
    arrays = radsurf.host_arrays(cfg, cp, sw, lw, bc, *fl)
    t0 = time.perf_counter()
    for a in arrays:
        radsurf.host_pin(a)
    t_lock = time.perf_counter() - t0
    t_pin = best_of(4)
    pinned = radsurf.host_pinned_bytes()
    radsurf.host_unpin_all()
    return {"value": ncol / min(t_def[1:]), "unit": "columns/s",
            "sample": "%d columns, host buffers in/out, default behaviour (the entry page-locks the caller's slices "
                      "for the duration of a large call), best of 3 calls after the first" % ncol,
            "caller_locked": {"value": ncol / min(t_pin[1:]), "unit": "columns/s", "locking_s": t_lock,
                              "pinned_bytes": pinned,
                              "sample": "same, arrays page-locked once by the caller (ss_host_pin)"},
            "pageable": {"value": ncol / min(t_page[1:]), "unit": "columns/s",
                         "sample": "same, SPARTACUS_PIN_HOST=0: no locking, the runtime stages every copy"},
            "bytes_per_column_over_pcie": bytes_in + bytes_out,
            "GB/s_both_directions": (bytes_in + bytes_out) * ncol / min(t_def[1:]) / 1e9}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--columns", type=int, default=1_000_000,
                    help="columns per GPU (weak scaling) or in total (strong scaling)")
    ap.add_argument("--gather-bc-every-step", action="store_true",
                    help="N > 1: also gather the four bc_out arrays to rank 0 after every step (off: the only "
                         "exchange is the final gather of all flux members)")
    ap.add_argument("--workload", default="c3", choices=["c2", "c2r3ns8", "c3", "c3ns4", "c4", "c4ns4", "rami5"],
                    help="c3 (default) is the headline configuration; the others are extra lines")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --columns per GPU; strong: --columns in total, sharded over the GPUs "
                         "(BASELINE.json configs[3] is --workload c4 --scaling strong)")
    ap.add_argument("--cpu-sample", type=int, default=100_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="timed region only: no serial-kernel steps, no CPU baseline, no PCIe sample "
                         "(what the rocprofv3 runs use, so that the profiler sees the timed region only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # SPARTACUS_BENCH_REHEARSAL=1: N ranks on fewer GPUs (ranks share devices, gloo
    # instead of RCCL, gathers staged through the host) - only to rehearse the N > 1
    # code path on a one-GPU box; its numbers mean nothing.
    rehearsal = os.environ.get("SPARTACUS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    cdev = "cpu" if rehearsal else "cuda"

    from spartacus_surface_amd import distributed, radsurf, workloads
    radsurf.load_library()
    radsurf.set_device(local_rank)

    # this rank's columns: its own seeded batch (weak), or its share of the total (strong;
    # the columns are i.i.d. draws, so a shard is generated directly instead of cutting it
    # out of 10^6 x 66 KB of host arrays on every rank)
    if args.scaling == "strong":
        r = distributed.shard_ranges(np.full(args.columns, 1), world)[rank]
        my_cols = max(0, r[1] - r[0] + 1)
    else:
        my_cols = args.columns
    total_cols_per_step = args.columns if args.scaling == "strong" else args.columns * world
    case = make_case(args.workload, my_cols, 12345 + rank)
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    plan = radsurf.Plan(cfg, cp, sw, lw, want_profiles=False)   # inputs now resident in HBM

    # The path has ONE exchange step (north_star: "RCCL over xGMI only for the final flux gather"):
    # every flux member to rank 0 after the last step, inside the timed region.  The steps themselves involve no
    # collective and no synchronisation, exactly as at N = 1 (columns are independent).  --gather-bc-every-step
    # adds what a host atmosphere model that runs on rank 0 would need after every step (the four bc_out arrays).
    gather_bc = gather_all = None
    if world > 1:
        bc_names = ["bc_out." + n for n in ("sw_albedo", "sw_albedo_dir", "lw_emissivity", "lw_emission")]
        if args.gather_bc_every_step:
            gather_bc = distributed.DeviceGather(plan, dist, cp.ncol, cp.ntotlay, names=bc_names, stage_host=rehearsal)
        gather_all = distributed.DeviceGather(plan, dist, cp.ncol, cp.ntotlay, stage_host=rehearsal)

    def step():
        plan.solve()
        if gather_bc:
            plan.synchronize()                      # results of the plan's streams are complete
            gather_bc()                             # ... and on rank 0 before the next solve overwrites them

    for _ in range(args.warmup):
        step()
    if gather_all and args.warmup > 0:
        plan.synchronize()
        gather_all()                                # RCCL channels of the big gather set up outside the timed region
    plan.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    fam_ms = {k: 0.0 for k in ("sw_layer", "sw_column", "sw_down", "lw_layer", "lw_column", "lw_down")}
    fam_n = dict.fromkeys(fam_ms, 0)
    for _ in range(args.steps):
        step()
        # HIP-event times of this step's launches (the query waits for the plan's stream)
        for k in fam_ms:
            fam_ms[k] += plan.kernel_ms(k)
            fam_n[k] += int(plan.kernel_ms(k + ".launches"))
    final_gather_ms = None
    if gather_all:
        plan.synchronize()
        tg = time.perf_counter()
        gather_all()                                # the exchange step of the path: every flux member to rank 0
        final_gather_ms = 1e3 * (time.perf_counter() - tg)
    plan.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    nfail = plan.eigen_failures()

    # two extra, untimed steps with strictly serial kernels (one stream): per-kernel
    # durations without the chunk overlap of the timed region - the evidence the roofline
    # block is computed from
    pipes = int(os.environ.get("SPARTACUS_PIPES", "4"))
    serial_ms = serial_n = None
    if pipes > 1 and not args.no_extras and rank == 0:
        plan.set_pipes(1)
        serial_ms = {k: 0.0 for k in fam_ms}
        serial_n = dict.fromkeys(fam_ms, 0)
        for _ in range(2):
            plan.solve()
            for k in fam_ms:
                serial_ms[k] += plan.kernel_ms(k)
                serial_n[k] += int(plan.kernel_ms(k + ".launches"))
        plan.synchronize()

    if rank == 0:
        value = total_cols_per_step * args.steps / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        fpc = flops_per_column(case)
        pmc, pmc_stale = pmc_summary(args.workload)
        # per-GPU step time for the per-kernel figures (what one GPU's kernels see)
        if serial_ms is not None:
            roofline = roofline_block(serial_ms, serial_n, 2, my_cols, case, pmc,
                                      "2 untimed steps on one stream (kernels strictly serial)", ms_per_step, fpc)
        elif pipes == 1:
            roofline = roofline_block(fam_ms, fam_n, args.steps, my_cols, case, pmc,
                                      "timed region, one stream (kernels strictly serial)", ms_per_step, fpc)
        else:
            roofline = {"bound": "fp64", "kernel": None, "achieved": None, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": None, "traffic": None,
                        "note": "not computed: with %d streams the event-bracketed kernel durations of the timed "
                                "region overlap (they add up to more than the step), and --no-extras skipped the "
                                "serial steps; run without --no-extras or with SPARTACUS_PIPES=1" % pipes}
        line = {
            "metric": "canopy columns/s (SW+LW)", "value": value, "unit": "columns/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s, %d columns %s, seed 12345 + rank"
                                   % (WORKLOAD_NAMES[args.workload], args.columns,
                                      "per GPU" if args.scaling == "weak" else "in total, sharded over the GPUs"),
                       "columns_per_gpu": my_cols, "columns_per_step": total_cols_per_step,
                       "layers": float(cp.ntotlay) / max(1, cp.ncol), "nsw": cfg.nsw,
                       "nlw": cfg.nlw, "parallelism": "columns sharded, %d GPU(s)" % world},
            "roofline": roofline,
            "fp64": fp64_figure(fpc, my_cols, ms_per_step, pmc),
            "streams": pipes,
            "library": {"version": radsurf.load_library().ss_version().decode(), "build_id": library_build_id(),
                        "pmc_summary": "profiles counters of this build" if pmc else pmc_stale},
            "kernel_spans_ms_per_step": {k: v / args.steps for k, v in fam_ms.items()},
            "kernel_spans_note": "event-bracketed spans of the timed region; with streams > 1 they overlap in "
                                 "time and are NOT durations to add up or to divide bytes by",
            "eigen_failures": nfail,
        }
        if serial_ms is not None:
            line["kernel_ms_per_step_serial"] = {k: v / 2 for k, v in serial_ms.items()}
        if world > 1:
            line["final_gather"] = {"ms": final_gather_ms, "bytes_per_rank": gather_all.bytes_per_rank,
                                    "arrays": len(gather_all.names),
                                    "GB/s_into_rank0": (world - 1) * gather_all.bytes_per_rank
                                    / (final_gather_ms * 1e-3) / 1e9 if final_gather_ms else None,
                                    "note": "bc_out + every allocated member of the four flux objects, device to "
                                            "device over RCCL, once after the last step, inside the timed region"
                                            + ("; bc_out alone after every step" if args.gather_bc_every_step else
                                               "; no exchange and no synchronisation between the steps")}
        if world == 1 and not args.no_extras and case.get("top_flux_dn_sw") is not None:
            line["scale_sum"] = scale_sum_rate(plan, case, my_cols)
        if world == 1 and not args.no_cpu_baseline and not args.no_extras:
            line["cpu_baseline"], line["parity"] = cpu_baseline(args.workload, min(args.cpu_sample, my_cols))
            line["vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
            line["pcie_inclusive"] = host_entry_rate(min(my_cols, 200_000))
        if world == 1 and not args.no_extras and args.workload == "c3":
            # the driver's own timed step (spectrum + re-pack + solve), then every other configuration of
            # BASELINE.json, driver-observed: C2 (config 2), C4 (config 4, on one GPU), n = 12 on the C3 column
            # set, and config 5 (rami5) as a throughput workload
            plan.close()
            line["step_with_input_refresh"] = input_refresh_figure(case, 3)
            line["other_workloads"] = {"c2": other_workload("c2", 100_000, 3, 2048),
                                       "c3ns4": other_workload("c3ns4", 200_000, 3, 2048),
                                       "c4": other_workload("c4", 100_000, 3, 128),
                                       "rami5": other_workload("rami5", 20_000, 3, 128)}
        print(json.dumps(line), flush=True)
    plan.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
