#!/usr/bin/env python3
"""bench.py -- moment-matching rollouts/sec on BASELINE.json's configuration
(N=1000, D=10, E=10, H=40; SURVEY.md section 8(d) synthetic recipe, seed 1234).

One "step" = one rollout = one PILCO.predict(m0, S0, H=40) (pilco.py:118-136) with
the GP factorisation cached on the device ("R-fwd" of SURVEY.md 8(d)); inputs are
resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W

N > 1: launched under torch.distributed.run, one rank per GPU.  torch is used only
to rendezvous (gloo: broadcast of the RCCL unique id and of the peer-exchange
handles, barrier, max-over-ranks); the output pairs are sharded over the ranks
inside libpilco_hip.so ("scaling": "strong": one rollout is split) with one exchange
per horizon step: direct peer stores + flags into hipIpc-mapped exchange areas,
or one ncclAllGather when a rank cannot attach them (config.exchange says which).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N, D, E, H = 1000, 10, 10, 40
FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector == matrix peak (public spec; not in MI355X_MICROARCH.md)


def algorithmic_work(n, d, e):
    """SURVEY.md 8(d) 'Algorithmic work per unit' for one moment-matching step."""
    p = e * (e + 1) // 2
    exps = p * n * n + e * n
    flop = p * n * n * (2 * d + 4) + 2 * e * n * n + e * n * (2 * d * d + 4 * d)
    byts = 8 * (e * n * n + n * d + 2 * e * n + e * d + d * d + e * e + e * d)
    return exps, flop, byts


def cpu_baseline(cfg, budget_s=120.0):
    """Time the NumPy restatement of the GPflow CPU path (oracle/tf_path.py) on the SAME workload: one factorisation,
    then the benchmarked H = 40 rollout executed for real (reward, propagate, state carried over) with the
    factorisation cached -- the whole rollout unless `budget_s` runs out first (then: the steps done, extrapolated, and
    said so).  The host's best configuration is FOUND, not assumed: one step is timed under every thread setting of the
    sweep below (BLAS threads for the sequential pair loop; a pool of pair workers with one BLAS thread each, which is
    how a multi-threaded CPU runtime runs the element-wise work too) and the rollout runs at the fastest.
    Beside it ONE step exactly as the reference evaluates it (mgpr.py:77-79,120-147: re-factorise, all E^2
    output pairs, the (E,E,N,N) tensors materialised), timed, as the reference-faithful variant."""
    from oracle import tf_path as tp
    try:
        from threadpoolctl import threadpool_limits
    except Exception:   # (no control over the BLAS pool: the sweep degenerates to the pair workers)
        import contextlib
        threadpool_limits = lambda limits=None: contextlib.nullcontext()
    ncpu = os.cpu_count() or 1
    model = tp.Model(cfg["X"], cfg["Y"], cfg["lengthscales"], cfg["variance"], cfg["noise"], pairs=True)
    t0 = time.perf_counter()
    model._cache = model.factorize()
    t_fact = time.perf_counter() - t0

    def one_step(m, s):
        tp.exponential_reward(m, s)
        return tp.propagate(model, tp.no_controller, m, s, cache=True)

    counts = [c for c in (8, 16, 32, 64, 128) if c <= ncpu] or [ncpu]
    settings = [("blas", c, 0) for c in counts] + [("pair_workers", 1, c) for c in counts]
    sweep = []
    for kind, blas, workers in settings:
        model.workers = workers
        with threadpool_limits(limits=blas):
            t0 = time.perf_counter()
            one_step(cfg["m0"], cfg["S0"])
            sweep.append({"kind": kind, "blas_threads": blas, "pair_workers": workers, "step_s": time.perf_counter() - t0})
    best = min(sweep, key=lambda r: r["step_s"])
    model.workers = best["pair_workers"]
    m, s = cfg["m0"], cfg["S0"]
    ts = []
    with threadpool_limits(limits=best["blas_threads"]):
        t_begin = time.perf_counter()
        for _ in range(H):
            t0 = time.perf_counter()
            m, s = one_step(m, s)
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_begin > budget_s:
                break
    done = len(ts)
    t_rollout = float(np.sum(ts)) if done == H else float(np.median(ts)) * H
    faithful = None
    try:   # the reference's own evaluation order for one step: factorisation + E^2 pairs (several GB of temporaries)
        ref_model = tp.Model(cfg["X"], cfg["Y"], cfg["lengthscales"], cfg["variance"], cfg["noise"], pairs=False)
        with threadpool_limits(limits=max(r["blas_threads"] for r in sweep if r["kind"] == "blas" and
                                          r["step_s"] == min(q["step_s"] for q in sweep if q["kind"] == "blas"))):
            t0 = time.perf_counter()
            tp.exponential_reward(cfg["m0"], cfg["S0"])
            tp.propagate(ref_model, tp.no_controller, cfg["m0"], cfg["S0"], cache=False)
            faithful = time.perf_counter() - t0
    except Exception as exc:   # (memory): the symmetric-pair number stands on its own
        faithful = repr(exc)
    cores = best["pair_workers"] if best["pair_workers"] > 1 else best["blas_threads"]
    how = ("%d pair workers x 1 BLAS thread" % best["pair_workers"]) if best["pair_workers"] > 1 else ("sequential pair loop, %d BLAS threads" % best["blas_threads"])
    out = dict(value=1.0 / t_rollout, unit="rollouts/s", cores=cores, kind="port",
               host_cpu_count=ncpu, rollout_s=t_rollout, steps_executed=done, factorisation_s=t_fact,
               best_setting=best, thread_sweep_one_step=sweep,
               sample=("NumPy+OpenBLAS restatement of the GPflow path (oracle/tf_path.py, 55 symmetric pairs), fp64, at the fastest of %d "
                       "thread settings swept on one step (%s: %.3f s per step): 1 factorisation (%.2f s, cached) + %s"
                       % (len(sweep), how, best["step_s"], t_fact,
                          ("the full H = 40 benchmarked rollout executed for real: %.1f s" % t_rollout) if done == H else
                          ("%d of 40 steps executed for real within the %.0f s budget (median %.3f s per step), "
                           "extrapolated to 40" % (done, budget_s, float(np.median(ts)))))))
    if isinstance(faithful, float):
        out["reference_faithful_step_s"] = faithful
        out["reference_faithful_rollouts_per_s"] = 1.0 / (H * faithful)
        out["reference_faithful_note"] = ("ONE step as the reference evaluates it (mgpr.py:77-79: re-factorisation every step, all E^2 = 100 output "
                                          "pairs, (E,E,N,N) tensors materialised), timed; rollouts/s = 1 / (40 x that step)")
    else:
        out["reference_faithful_step_s"] = None
        out["reference_faithful_note"] = "not measured: %s" % faithful
    return out


def _sha16(path):
    import hashlib
    try:
        with open(path, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def newest_profile(suffix):
    """profiles/rNN<suffix> of the highest round that exists (committed rocprofv3 evidence)."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]" + suffix)):
        m = re.match(r"r(\d\d)", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


def rocprof_avg_us(kernel_substr, suffix="_kernel_stats.csv"):
    """(average launch duration in us, calls, file) of the first kernel whose name contains `kernel_substr` in the newest
    committed rocprofv3 --kernel-trace --stats summary."""
    import csv
    f = newest_profile(suffix)
    if not f:
        return None, None, None
    try:
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Name"]:
                return float(r["AverageNs"]) / 1e3, int(r["Calls"]), "profiles/" + os.path.basename(f)
    except Exception:
        pass
    return None, None, "profiles/" + os.path.basename(f)


KERNEL_SOURCES = ("pair.hip", "pair_device.h", "prep.hip", "prep_device.h", "prep_kernel.h", "glue_device.h", "mm_device.h",
                  "rollout.hip", "bwd.hip", "linalg.hip")


def kernel_source_hashes():
    return {n: _sha16(os.path.join(ROOT, "pilco_amd", "csrc", n)) for n in KERNEL_SOURCES}


def _median_ms(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def secondary_metrics(ctx, cfg, policy, rewards, ms_rollout, steps):
    """Everything else SURVEY.md 8(d) / BASELINE.md section 3 ask for, measured AFTER the timed region (1 GPU):
    back-to-back graph replays (no host sync between rollouts), R-fwd+fact, R-grad at C2u, config 4 (SMGPR)."""
    from pilco_amd import synthetic
    from pilco_amd.models import PILCO
    out = {}
    res = ctx.rollout_timed(policy, rewards, cfg["m0"], cfg["S0"], H, steps, time_pair=False)
    out["back_to_back_rollouts_per_s"] = steps * 1e3 / res["ms_total"]
    out["back_to_back_note"] = "hipGraph replays queued without a host sync or result download in between (hipEvent-timed)"
    # throughput mode: two independent rollouts in flight on ONE GPU (two contexts, two host threads): the serial head of
    # one rollout (a chain of latencies that leaves the chip idle) runs under the pair kernel of the other.  Not the
    # headline (PILCO.predict is called one at a time by the optimiser); restarts / several initial states can use it.
    try:
        import threading
        from pilco_amd import _lib
        ctxb = _lib.Context(device=ctx.device)
        ctxb.gp_set_data(0, cfg["X"], cfg["Y"])
        ctxb.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"])
        ctxb.gp_factorize(0)
        ctxb.rollout(policy, rewards, cfg["m0"], cfg["S0"], H)
        n_each = max(steps, 10)
        def worker(c):
            for _ in range(n_each):
                c.rollout(policy, rewards, cfg["m0"], cfg["S0"], H)
        th = [threading.Thread(target=worker, args=(c,)) for c in (ctx, ctxb)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        out["two_concurrent_rollouts_per_s"] = 2 * n_each / (time.perf_counter() - t0)
        out["two_concurrent_note"] = "two contexts of one process on one GPU, each calling pilco_rollout back to back from its own host thread"
        ctxb.close()
    except Exception as exc:
        out["two_concurrent_rollouts_per_s"] = repr(exc)
    # pilco_rollout_batch: B independent rollouts of this model in flight together (lanes borrow the model; the serial head of
    # one lane runs under the pair kernels of the others).  Throughput mode for multi-start policy search / several initial
    # states; NEVER the headline (`value` is one rollout at a time, as PILCO.predict is called).
    try:
        rs = np.random.RandomState(7)
        batched = {}
        for B in (4, 8):
            m0b = cfg["m0"] + 0.05 * rs.randn(B, E)
            S0b = np.stack([cfg["S0"]] * B)
            mB, SB, RB = ctx.rollout_batch([policy] * B, rewards, m0b, S0b, H)
            solo = ctx.rollout(policy, rewards, m0b[B - 1], S0b[B - 1], H)
            same = bool(np.array_equal(mB[B - 1], solo[0][0]) and np.array_equal(SB[B - 1], solo[1]) and RB[B - 1] == solo[2][0, 0])
            reps = max(3, steps // 2)
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.rollout_batch([policy] * B, rewards, m0b, S0b, H)
            dt = time.perf_counter() - t0
            batched["B=%d" % B] = {"rollouts_per_s": B * reps / dt, "ms_per_batch": dt / reps * 1e3, "last_lane_bit_identical_to_its_solo_run": same}
        batched["note"] = ("aggregate over B lanes, host-synchronised with every lane's result downloaded; bound by the pair kernel alone: "
                           "1 / (40 x its launch duration)")
        out["batched_rollouts"] = batched
    except Exception as exc:
        out["batched_rollouts"] = {"error": repr(exc)}
    # one exact factorisation (mgpr.py:81-89) = one replay of its launch sequence as a hipGraph + the read-back of the
    # not-positive-definite word: 20 single calls, event-timed each; the median is what the derived numbers use
    ctx.factorize_timed(0, 2)
    fts = sorted(ctx.factorize_timed(0, 1) for _ in range(20))
    fact_ms = float(np.median(fts))
    out["factorisation_ms"] = fact_ms
    out["factorisation_ms_min_median_max"] = [fts[0], fact_ms, fts[-1]]
    # one evaluation of the GP-training objective (mgpr.py:46-58 through GPflow's training_loss): exact NLML + analytic
    # gradient of all E outputs, hyper-parameters re-uploaded first as an optimiser step does (the factorisation is redone)
    def nlml_eval():
        ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"])
        ctx.gp_nlml(0, D, E)
    try:
        nlml_eval()
        out["nlml_eval_ms"] = _median_ms(nlml_eval, 7)
    except Exception as exc:
        out["nlml_eval_ms"] = None
        out["nlml_eval_error"] = repr(exc)
    ctx.gp_factorize(0)
    out["R_fwd_fact_rollouts_per_s"] = 1e3 / (fact_ms + ms_rollout)
    # FLOP_fact = E (N^2 (3D+3) + N^3/3 + 2 N^3/3 + 2 N^2) (triangular-inverse route), bound: f64 MFMA
    flop_fact = E * (N * N * (3 * D + 3) + N ** 3 / 3.0 + 2.0 * N ** 3 / 3.0 + 2.0 * N * N)
    _, flop_step, _ = algorithmic_work(N, D, E)
    out["R_fwd_fact_roofline"] = {
        "bound": "mfma", "unit": "TFLOP/s", "peak": FP64_PEAK_TFLOPS,
        "factorisation": {"achieved": flop_fact / (fact_ms * 1e-3) / 1e12, "frac": flop_fact / (fact_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                          "algorithmic_flop": flop_fact, "ms": fact_ms,
                          "what_bounds_it": "the blocked Cholesky's chain of 16 dependent steps (panel GEMM + trailing update whose first tile carries the next 64 x 64 block factorisation: latency), then the L^-1 levels and the iK GEMM (f64 MFMA, L2-bound at 64 x 64 tiles)"},
        "factorise_then_rollout": {"achieved": (flop_fact + H * flop_step) / ((fact_ms + ms_rollout) * 1e-3) / 1e12,
                                   "frac": (flop_fact + H * flop_step) / ((fact_ms + ms_rollout) * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                   "algorithmic_flop": flop_fact + H * flop_step, "ms": fact_ms + ms_rollout}}
    # ---- C2u: value + gradient w.r.t. a linear controller (state 10 + 1 control, D = 11)
    cu = synthetic.config_c2(N=N, D=D + 1, E=E)
    p = PILCO((cu["X"], cu["Y"]), horizon=H, ctx=ctx)
    for i, mdl in enumerate(p.mgpr.models):
        mdl.kernel.lengthscales.assign(cu["lengthscales"][i])
        mdl.kernel.variance.assign(cu["variance"][i])
        mdl.likelihood.variance.assign(cu["noise"][i])
    p.controller.W.assign(cu["W"])
    p.controller.b.assign(cu["b"])
    p.controller.max_action = 1.0
    p.m_init, p.S_init = cu["m0"], cu["S0"]
    p.value_and_gradient()
    g_ms = _median_ms(lambda: p.value_and_gradient(), 5)
    p.compute_reward()
    f_ms = _median_ms(p.compute_reward, 5)
    out["R_grad_C2u_ms"] = g_ms
    out["R_grad_over_R_fwd_C2u"] = g_ms / f_ms
    out["R_grad_note"] = ("value + gradient w.r.t. (W, b) by the Jacobian tape: the forward rollout runs the reverse sweep in place of the "
                          "forward pair kernel (one O(N^2) pass per step gives value and Jacobian), the reverse sweep is host algebra")
    out["R_grad_C2u_per_s"] = 1e3 / g_ms
    out["R_fwd_C2u_ms"] = f_ms
    # pilco_rollout_grad_batch: the restarts of optimize_policy (pilco.py:94-107) as lanes -- B value-and-gradient rollouts of this
    # model in flight, one controller each; every lane bit-identical to its solo call.  Throughput mode, never the headline.
    try:
        rsb = np.random.RandomState(11)
        spec = p._policy_spec()
        lanes = {}
        for B in (2, 3):
            pols = [dict(spec, W=cu["W"] + 0.05 * rsb.randn(*np.shape(cu["W"])), b=np.ravel(cu["b"])) for _ in range(B)]
            m0b, S0b = np.tile(np.ravel(cu["m0"]), (B, 1)), np.tile(cu["S0"], (B, 1, 1))
            rB, dWB, dbB = ctx.rollout_grad_batch(pols, p._reward_terms(), m0b, S0b, H)
            solo = ctx.rollout_grad(pols[B - 1], p._reward_terms(), m0b[B - 1], S0b[B - 1], H)
            same = bool(rB[B - 1] == solo[0] and np.array_equal(dWB[B - 1].ravel(), np.ravel(solo[1])) and np.array_equal(dbB[B - 1].ravel(), np.ravel(solo[2])))
            b_ms = _median_ms(lambda: ctx.rollout_grad_batch(pols, p._reward_terms(), m0b, S0b, H), 5)
            lanes["B=%d" % B] = {"ms_per_batch": b_ms, "ms_per_lane": b_ms / B, "per_lane_over_R_fwd_C2u": b_ms / B / f_ms,
                                 "last_lane_bit_identical_to_its_solo_call": same}
        lanes["note"] = "the sweep launches fill the chip by themselves: lanes hide the heads and the finish only (bound: 0.77 of the solo time per lane)"
        out["R_grad_C2u_lanes"] = lanes
    except Exception as exc:
        out["R_grad_C2u_lanes"] = {"error": repr(exc)}
    # roofline of the value-and-gradient rollout's dominant kernel (the reverse sweep k_mm_bwd_pair that stands in for the
    # forward pair kernel): its launches bracketed by HIP events in a separate, untimed pass
    try:
        ctx.set_pair_timing(True)
        p.value_and_gradient()
        sw_ms, sw_n = ctx.get_pair_timing()
    finally:
        ctx.set_pair_timing(False)
    if sw_n > 0:
        Du, Pn = D + 1, E * (E + 1) // 2
        sw_us = sw_ms * 1e3 / sw_n
        sw_flop = Pn * N * N * (2 * (Du + 1) + 2 * (Du + 1) + 4)   # exponent contraction + moment contraction + weights / sums
        rp, rpc, rpf = rocprof_avg_us("k_mm_bwd_pair", "_grad_kernel_stats.csv")
        out["R_grad_roofline"] = {
            "bound": "mfma", "unit": "TFLOP/s", "peak": FP64_PEAK_TFLOPS, "kernel": "k_mm_bwd_pair (reverse sweep: exponent MFMA, fp64 exp, moment MFMA, row / column sums)",
            "avg_launch_us": sw_us, "launches": sw_n, "achieved": sw_flop / (sw_us * 1e-6) / 1e12,
            "frac": sw_flop / (sw_us * 1e-6) / 1e12 / FP64_PEAK_TFLOPS, "algorithmic_flop_per_launch": sw_flop, "exp_per_launch": Pn * N * N,
            "rocprofv3_avg_launch_us": rp, "rocprofv3_calls": rpc, "rocprofv3_source": rpf,
            "rollout_level": {"algorithmic_flop": H * sw_flop, "ms": g_ms, "achieved": H * sw_flop / (g_ms * 1e-3) / 1e12,
                              "frac": H * sw_flop / (g_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}}
    # ---- config 4: SMGPR (M=200, N=5000, D=10, E=10), FITC factorisation + rollout
    c4 = synthetic.config_c4()
    ctx.gp_set_data(0, c4["X"], c4["Y"])
    ctx.gp_set_hyp(0, c4["lengthscales"], c4["variance"], c4["noise"])
    ctx.gp_set_inducing(0, c4["Z"])
    ctx.gp_factorize(0)
    ctx.factorize_timed(0, 2)
    f4 = sorted(ctx.factorize_timed(0, 1) for _ in range(20))
    out["config4_fitc_factorisation_ms"] = float(np.median(f4))
    out["config4_fitc_factorisation_ms_min_median_max"] = [f4[0], float(np.median(f4)), f4[-1]]
    try:   # GPRFITC objective + gradients w.r.t. the hyper-parameters and the 10 x 200 x 10 inducing inputs (smgpr.py:24-52 under GPflow's loss)
        Z4 = np.stack([c4["Z"]] * E)
        ctx.gp_fitc_nlml(0, Z4, D, E)
        out["config4_fitc_objective_eval_ms"] = _median_ms(lambda: ctx.gp_fitc_nlml(0, Z4, D, E), 7)
        ctx.gp_set_inducing(0, c4["Z"])
        ctx.gp_factorize(0)
    except Exception as exc:
        out["config4_fitc_objective_eval_ms"] = None
        out["config4_fitc_objective_error"] = repr(exc)
    ctx.rollout(policy, rewards, c4["m0"], c4["S0"], H)
    r4 = _median_ms(lambda: ctx.rollout(policy, rewards, c4["m0"], c4["S0"], H), 10)
    out["config4_rollout_ms"] = r4
    out["config4_rollouts_per_s"] = 1e3 / r4
    M4, N4 = c4["Z"].shape[0], c4["X"].shape[0]
    flop_fitc = 3.0 * E * M4 * M4 * N4   # ~3 E M^2 N (V = L^-1 Kmn, V V^T, the right-hand sides)
    _, flop_step4, _ = algorithmic_work(M4, D, E)
    try:
        ctx.set_pair_timing(True)
        ctx.rollout(policy, rewards, c4["m0"], c4["S0"], H)
        p4_ms, p4_n = ctx.get_pair_timing()
    finally:
        ctx.set_pair_timing(False)
    out["config4_roofline"] = {
        "bound": "mfma", "unit": "TFLOP/s", "peak": FP64_PEAK_TFLOPS,
        "fitc_factorisation": {"achieved": flop_fitc / (out["config4_fitc_factorisation_ms"] * 1e-3) / 1e12,
                               "frac": flop_fitc / (out["config4_fitc_factorisation_ms"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                               "algorithmic_flop": flop_fitc, "ms": out["config4_fitc_factorisation_ms"],
                               "hbm_GBps_Kmn_once": 8.0 * E * M4 * N4 / (out["config4_fitc_factorisation_ms"] * 1e-3) / 1e9},
        "rollout": {"achieved": H * flop_step4 / (r4 * 1e-3) / 1e12, "frac": H * flop_step4 / (r4 * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                    "algorithmic_flop": H * flop_step4, "ms": r4,
                    "pair_kernel_avg_launch_us": (p4_ms * 1e3 / p4_n) if p4_n else None,
                    "what_bounds_it": "launch / latency: at M = 200 a step has 55 x 200^2 = 2.2e6 exps (2 us of the fp64 pipe); the step is the "
                                      "serial head's chain of latencies plus two launch boundaries"}}
    # restore the benchmark model in slot 0
    ctx.gp_set_inducing(0, None)
    ctx.gp_set_data(0, cfg["X"], cfg["Y"])
    ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"])
    return out


def config5_metrics(ctx, with_cpu):
    """BASELINE config 5: the loop of the reference's examples/inverted_pendulum.py:13-39 (InvertedPendulum-v2-shaped
    built-in plant, 5 x 40 random steps, RbfController(bf=10), horizon 40, 3 x [optimize_models, optimize_policy
    (maxiter=50), 100-step rollout]) on the HIP path, and ONE iteration of the same loop on the CPU stand-in
    (oracle/cpu_loop.py; bounded sample) timed beside it."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import inverted_pendulum as ip
    from pilco_amd import _lib
    prev = _lib._default_ctx
    _lib.set_context(ctx)
    try:
        hip = ip.run_hip(verbose=False)
    finally:
        _lib.set_context(prev)
    out = {"hip_total_s": hip["total_s"], "hip_one_time_init_s": hip.get("init_s"), "hip_iterations": hip["iterations"]}
    if with_cpu:
        # in a child process with a hard time limit: a CPU stand-in must never stall the benchmark
        import subprocess
        code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r); import inverted_pendulum as ip; "
                "print('CPU5 ' + json.dumps(ip.run_cpu(iters=1, verbose=False)))" % (ROOT, os.path.join(ROOT, "examples")))
        try:
            pr = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
            line = [l for l in pr.stdout.splitlines() if l.startswith("CPU5 ")]
            cpu = json.loads(line[-1][5:]) if line else None
        except subprocess.TimeoutExpired:
            cpu = None
        if cpu is None:
            out["cpu_stand_in_first_iteration"] = "did not finish within 240 s"
            return out
        c0, h0 = cpu["iterations"][0], hip["iterations"][0]
        out["cpu_stand_in_first_iteration"] = c0
        out["cpu_stand_in_note"] = ("oracle/cpu_loop.py (NumPy/SciPy GP fit + torch-CPU reverse mode through the restated "
                                    "rollout, L-BFGS-B maxiter=50), 8 torch threads of %d host threads; same data, same objective; iteration 0 only" % os.cpu_count())
        out["speedup_first_iteration"] = ((c0["optimize_models_s"] + c0["optimize_policy_s"]) /
                                          max(h0["optimize_models_s"] + h0["optimize_policy_s"], 1e-9))
    return out


def engine_clock_under_pair_load(ctx, cfg, policy, rewards):
    """Engine clock while the pair kernel runs: shader-clock ticks against the 100 MHz wall clock inside wave 0 of one
    pair-kernel launch (developer stamps on a SEPARATE context with eager launches -- never inside the timed region)."""
    from pilco_amd import _lib
    cd = _lib.Context(device=ctx.device)
    try:
        cd.debug_timestamps(read=False)
        cd.gp_set_data(0, cfg["X"], cfg["Y"])
        cd.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"])
        cd.gp_factorize(0)
        ts, spans = None, []
        for _ in range(4):
            cd.rollout(policy, rewards, cfg["m0"], cfg["S0"], 3)
            ts = cd.debug_timestamps()
            ends = [x for x in cd.debug_blocks(960 + 256 + 8)[960:960 + 256] if x]
            if ends and ts[16]:
                spans.append((max(max(ends), ts[17], ts[18]) - ts[16]) / 100.0)   # first wave in -> last wave out, us
        return (ts[33] - ts[32]) / ((ts[17] - ts[16]) / 100.0), (float(np.median(spans[1:])) if len(spans) > 1 else None)
    finally:
        cd.close()


def verify_against_reference(mH, SH, reward, fatal=True):
    """The timed rollout's result against tests/golden/c2_rollout.npz: the reference's own source executed at this
    exact configuration (oracle/gen_golden_c2.py).  Raises when the 1e-5 relative tolerance of north_star is missed."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "c2_rollout.npz"))
    assert int(g["N"]) == N and int(g["D"]) == D and int(g["E"]) == E and int(g["H"]) == H
    Mr, Sr, Rr = g["M_traj"][:, -1], g["S_traj"][:, :, -1], float(g["R_traj"][-1])
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - b) / np.maximum(np.abs(b), 1e-300)))
    errs = {"m_H": rel(mH.ravel(), Mr), "S_H": rel(SH, Sr), "reward": rel([reward], [Rr])}
    bad = {k: v for k, v in errs.items() if not v <= 1e-5}
    if bad and not fatal:
        raise ValueError("rollout does not match the executed reference: %r" % bad)
    if bad:
        raise SystemExit("bench.py: the timed rollout does not match the executed reference (rtol 1e-5): %r" % bad)
    return dict(against="tests/golden/c2_rollout.npz (reference source executed, H=40)", rtol=1e-5, max_rel_err=errs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Invoked plainly (`python bench.py --gpus N ...`, the way the driver types the 1-GPU line): start the N ranks
        # ourselves -- the same launcher line the driver uses, one process per GPU, rendezvous on 127.0.0.1 -- and let
        # rank 0 of that job print the one JSON line.  exec, not spawn: no second Python sits above the ranks.
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or call it without a launcher)"
                         % (args.gpus, world))

    from pilco_amd import _lib, synthetic
    cfg = synthetic.config_c2(N=N, D=D, E=E)
    # PILCO_BENCH_SHARE_GPU=1 (test runs on a one-GPU box): all ranks use GPU 0.  RCCL refuses two ranks on one device, so
    # the once-per-model beta exchange goes over gloo and the per-step exchange is the peer exchange or nothing.
    share_gpu = os.environ.get("PILCO_BENCH_SHARE_GPU", "0") == "1"
    ctx = _lib.Context(device=0 if share_gpu else local_rank)

    dist = None
    rccl_ranks = None
    exchange = "none (one rank)"
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
        if share_gpu:
            ctx.shard_set(rank, world)
        else:
            id_t = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8)
            if rank == 0:
                id_t = torch.frombuffer(bytearray(ctx.comm_unique_id()), dtype=torch.uint8).clone()
            dist.broadcast(id_t, src=0)
            ctx.comm_init(bytes(id_t.numpy().tobytes()), rank, world)
            rccl_ranks = ctx.comm_count()       # what RCCL itself says (ncclCommCount), not what we asked for
        exchange = "ncclAllGather per horizon step (RCCL)"

    ctx.gp_set_data(0, cfg["X"], cfg["Y"])
    ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"])
    ctx.gp_factorize(0)        # N > 1: every rank factorises only the outputs it owns
    if world > 1 and share_gpu:   # no communicator: beta rows over the host transport
        import torch
        rows = torch.from_numpy(ctx.beta_export(0))
        allr = [torch.empty_like(rows) for _ in range(world)]
        dist.all_gather(allr, rows)
        ctx.beta_import(torch.stack(allr).numpy(), 0)
    if world > 1 and os.environ.get("PILCO_BENCH_PEER", "1") == "1":
        # Per-step exchange without a collective launch: every rank stores its segment into every rank's exchange area
        # (hipIpc-mapped device memory, xGMI between GPUs) and raises a flag; handles travel over gloo once.  Any rank
        # that cannot attach sends everybody back to the RCCL path.
        import torch
        ok, handle = 1, None
        try:
            handle = ctx.peer_export()
        except _lib.PilcoError:
            ok = 0
        handles = [None] * world
        dist.all_gather_object(handles, handle)
        if ok and all(h is not None for h in handles):
            try:
                ctx.peer_attach(handles, share_gpu=share_gpu)
            except _lib.PilcoError:
                ok = 0
        else:
            ok = 0
        t = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 1:
            exchange = "peer stores + flags into hipIpc-mapped exchange areas (no collective launch per step)"
        else:
            ctx.peer_detach()
    if world > 1 and share_gpu and not ctx.peer_attached():
        raise SystemExit("bench.py: ranks sharing one GPU need the peer exchange (RCCL refuses duplicate devices)")
    policy = dict(kind=_lib.POLICY_NONE, state_dim=E, control_dim=0)
    rewards = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(E), t=np.zeros(E))]

    # One step = ONE call of pilco_rollout -- the entry PILCO.predict (pilco.py:118-136) makes: host buffers in, the whole
    # H = 40 rollout on the device, (m_H, S_H, reward) downloaded, host-synchronised.  The model (X, beta, iK, ...) is
    # resident in HBM; per call only (m0, S0) go up and E + E*E + 1 doubles come down.
    def one_rollout():
        return ctx.rollout(policy, rewards, cfg["m0"], cfg["S0"], H)

    if dist is not None and ctx.peer_attached():
        # first rollout over the peer exchange: a rank whose flag wait gives up (or whose result is off) takes every rank
        # back to the RCCL path before anything is timed
        import torch
        ok = 1
        try:
            mH, SH, rew = one_rollout()
            verify_against_reference(mH, SH, float(rew[0, 0]), fatal=False)
        except Exception:
            ok = 0
        t = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) != 1:
            ctx.peer_detach()
            if share_gpu:
                raise SystemExit("bench.py: the peer exchange failed and ranks sharing one GPU have no RCCL path")
            exchange = "ncclAllGather per horizon step (RCCL; the peer exchange failed its first rollout)"
    def timed_leg():
        """args.warmup untimed + args.steps timed rollouts, barrier on both sides, MAX over ranks."""
        for _ in range(args.warmup):
            one_rollout()
        if dist is not None:
            dist.barrier()
        calls = []
        t_begin = time.perf_counter()
        res = None
        for _ in range(args.steps):
            t0 = time.perf_counter()
            res = one_rollout()          # returns after hipStreamSynchronize + download
            calls.append((time.perf_counter() - t0) * 1e3)
        wall = (time.perf_counter() - t_begin) * 1e3
        if dist is not None:
            dist.barrier()
            import torch
            t = torch.tensor([wall], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        return wall, calls, res

    wall_ms, per_call, (mH, SH, rew) = timed_leg()
    ms_per_rollout = wall_ms / args.steps
    verified = verify_against_reference(mH, SH, float(rew[0, 0])) if rank == 0 else None

    # dominant kernel (pair kernel): its own HIP-event pairs on the launch stream, separate pass
    prof = ctx.rollout_timed(policy, rewards, cfg["m0"], cfg["S0"], H, 1, time_pair=True)
    pair_ms = prof["ms_pair"] / max(prof["n_pair_launches"], 1)
    exps, flop, byts = algorithmic_work(N, D, E)
    flop_local = flop / world  # pairs are dealt over the ranks
    achieved = flop_local / (pair_ms * 1e-3) / 1e12 if pair_ms > 0 else 0.0
    # (N > 1: measured under the exchange `value` was timed with, before the other one is switched in)

    # N > 1: BOTH exchanges in this one invocation.  `value` is the leg above (the peer exchange when every rank attached it);
    # the other exchange is timed the same way and reported under secondary, each verified against the executed reference.
    other_exchange = None
    if dist is not None:
        import torch
        if ctx.peer_attached() and not share_gpu:
            ctx.peer_detach()                 # every rank: the same rollouts now take one ncclAllGather per horizon step
            ok, err = 1, None
            try:
                w2, calls2, (m2, S2, r2) = timed_leg()
                v2 = verify_against_reference(m2, S2, float(r2[0, 0]), fatal=False)
            except Exception as exc:          # (a rank that fails still reaches the all_reduce below)
                ok, err = 0, repr(exc)
            t = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) == 1:
                other_exchange = {"exchange": "ncclAllGather per horizon step (RCCL)", "rollouts_per_s": 1e3 / (w2 / args.steps),
                                  "ms_per_step": w2 / args.steps, "median_ms_per_call": float(np.median(calls2)), "verified": v2}
            else:
                other_exchange = {"exchange": "ncclAllGather per horizon step (RCCL)", "error": err or "another rank failed"}
        elif ctx.peer_attached():
            other_exchange = {"exchange": "ncclAllGather per horizon step (RCCL)",
                              "error": "not available: the ranks share one GPU (PILCO_BENCH_SHARE_GPU=1) and RCCL refuses duplicate devices"}
        else:
            other_exchange = {"exchange": "peer stores + flags", "error": "not attached on every rank (or disabled by PILCO_BENCH_PEER=0): `value` is the RCCL leg"}

    # the clock the engine actually runs at under this kernel's load, and the kernel's own instruction-issue bound there:
    # per 16-column step a wave issues 6 v_mfma_f64_16x16x4 (64 cycles) + 104 VALU ops (4 cycles) on the shared fp64 pipe
    clock_mhz, issue_us, span_us = None, None, None
    if world == 1:
        try:
            clock_mhz, span_us = engine_clock_under_pair_load(ctx, cfg, policy, rewards)
            issue_us = exps / 512.0 / 1024.0 * 800.0 / clock_mhz
        except Exception:
            clock_mhz = None
    # HBM bytes per pair-kernel launch: PMC counters (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes) of the committed
    # rocprofv3 profile of this same command, newest round first; a live run cannot collect counters on itself
    traffic, traffic_src, traffic_meta = None, None, {}
    f_pmc = newest_profile("_pmc_summary.json")
    if f_pmc:
        try:
            with open(f_pmc) as f:
                pj = json.load(f)
            traffic = pj["pair_kernel_hbm_bytes_per_launch"] / world
            traffic_src = "profiles/" + os.path.basename(f_pmc)
            now = kernel_source_hashes()
            then = pj.get("kernel_source_sha16") or {}
            changed = sorted(k for k in ("pair.hip", "pair_device.h", "mm_device.h") if then.get(k) != now.get(k))
            traffic_meta = {"profile_head": pj.get("git_head"), "profile_date": pj.get("date"),
                            "stale": bool(changed) or not then,
                            "stale_because": (("kernel sources changed since the profile: " + ", ".join(changed)) if changed and then else
                                              ("the profile predates source hashes (round <= 2)" if not then else None))}
            if traffic_meta["stale"]:
                print("bench.py: WARNING roofline.traffic comes from %s, which is older than the pair kernel's sources (%s): "
                      "re-run tools/profile_round.sh" % (traffic_src, traffic_meta["stale_because"]), file=sys.stderr)
        except Exception:
            traffic = None
    rp_us, rp_calls, rp_file = rocprof_avg_us("k_mm_pair_sk")

    # N > 1 only: the zero-communication alternative (SURVEY.md 8(e) "Alternative DP"): every rank also runs the whole
    # rollout on its own GPU (an unsharded second context); the aggregate is reported under "secondary", never as `value`.
    replicas = None
    if dist is not None:
        import torch
        local_ms, err = float("inf"), None
        try:   # no collective inside the try: a rank that fails must not leave the others waiting
            ctx2 = _lib.Context(device=0 if share_gpu else local_rank)
            ctx2.gp_set_data(0, cfg["X"], cfg["Y"])
            ctx2.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"])
            ctx2.gp_factorize(0)
            for _ in range(max(args.warmup, 1)):
                ctx2.rollout(policy, rewards, cfg["m0"], cfg["S0"], H)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ctx2.rollout(policy, rewards, cfg["m0"], cfg["S0"], H)
            local_ms = (time.perf_counter() - t0) * 1e3
            one_gpu = None
            if rank == 0:   # the 1-GPU phase times the Amdahl model below is built from (measured while the other ranks idle)
                pr1 = ctx2.rollout_timed(policy, rewards, cfg["m0"], cfg["S0"], H, 1, time_pair=True)
                one_gpu = (local_ms / args.steps, pr1["ms_pair"] * 1e3 / max(pr1["n_pair_launches"], 1))
        except Exception as exc:
            err = repr(exc)
        tt = torch.tensor([local_ms], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if np.isfinite(float(tt.item())):
            replicas = {"replica_rollouts_per_s": world * args.steps * 1e3 / float(tt.item()),
                        "note": "independent unsharded rollouts, one per GPU, concurrently, no communication (not the sharded metric)"}
            if rank == 0 and err is None and one_gpu:
                replicas["one_gpu_ms_per_rollout"], replicas["one_gpu_pair_us_per_launch"] = one_gpu
        else:
            replicas = {"error": err or "a rank failed"}

    per_rank, amdahl = None, None
    if dist is not None:
        # per rank: its pair kernel per launch (own share of the pairs) and what is left of a step (serial head + exchange)
        mine = {"rank": rank, "pair_us_per_launch": pair_ms * 1e3, "pair_launches": int(prof["n_pair_launches"]),
                "other_us_per_step": (prof["ms_total"] - prof["ms_pair"]) * 1e3 / H if prof["n_pair_launches"] else None}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank == 0:
        if dist is not None and replicas and "one_gpu_ms_per_rollout" in replicas:
            # what N ranks could reach if the exchange were free: the slowest rank's pair share + the serial head of a step
            # (which every rank repeats: Amdahl), from THIS run's 1-GPU phase times.  The measured point is read against it.
            t1, p1 = replicas["one_gpu_ms_per_rollout"], replicas["one_gpu_pair_us_per_launch"]
            head1 = t1 * 1e3 / H - p1
            slowest_pair = max(r["pair_us_per_launch"] for r in per_rank)
            pred_step = slowest_pair + head1
            amdahl = {"one_gpu_ms_per_rollout": t1, "one_gpu_pair_us_per_launch": p1, "one_gpu_head_us_per_step": head1,
                      "slowest_rank_pair_us_per_launch": slowest_pair,
                      "predicted_rollouts_per_s_with_a_free_exchange": 1e6 / (H * pred_step),
                      "measured_rollouts_per_s": 1e3 / ms_per_rollout,
                      "exchange_and_skew_us_per_step": ms_per_rollout * 1e3 / H - pred_step,
                      "ideal_linear_rollouts_per_s": world * 1e3 / t1}
        out = {
            "metric": "moment-matching rollouts/sec (N=1000,D=10,E=10,H=40)",
            "value": 1e3 / ms_per_rollout,
            "unit": "rollouts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_rollout,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: N=1000 D=10 E=10 H=40 full rollout + reward, "
                                   "control_dim=0 (D==E read literally), ExponentialReward(W=I,t=0), "
                                   "factorisation cached (R-fwd); one step = one host-synchronised pilco_rollout call "
                                   "(what PILCO.predict makes) with its result downloaded",
                       "parallelism": "pairs%d" % world, "exchange": exchange,
                       "median_ms_per_call": float(np.median(per_call)), "min_ms_per_call": float(np.min(per_call)),
                       "max_ms_per_call": float(np.max(per_call))},
            "verified": verified,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_mm_pair_sk (f64 MFMA exponent tiles + fp64 exp; MFMA and fp64 VALU share one pipe)",
                         "avg_launch_ms": pair_ms,
                         "in_kernel_span_us": span_us,
                         "engine_clock_mhz_under_load": clock_mhz, "issue_bound_us_at_that_clock": issue_us,
                         "frac_of_issue_bound": (issue_us / (pair_ms * 1e3)) if (issue_us and pair_ms > 0) else None,
                         "traffic_profile": traffic_meta,
                         "rocprofv3_avg_launch_us": rp_us, "rocprofv3_calls": rp_calls, "rocprofv3_source": rp_file,
                         "note": "avg_launch_ms: hipEvent pairs around each launch of an eager replay, measured live in this run (includes event "
                                 "latency: an upper bound); rocprofv3_avg_launch_us: the same kernel in the committed --kernel-trace --stats summary "
                                 "named beside it (read from that file, not typed in); in_kernel_span_us: first wave in to last wave out by "
                                 "the kernel's own 100 MHz stamps. frac = SURVEY 8(d) algorithmic FLOP (exp internals excluded) / spec peak at 2.4 GHz; the kernel's own "
                                 "bound is instruction issue: 800 cycles per 512 exps (6 f64 MFMA + 104 VALU ops) at the measured clock",
                         "algorithmic_flop_per_launch": flop_local, "exp_per_launch": exps / world,
                         "gexp_per_s": exps / world / (pair_ms * 1e-3) / 1e9 if pair_ms > 0 else 0.0,
                         "algorithmic_bytes_per_launch": byts / world,
                         "hbm_GBps_algorithmic": byts / world / (pair_ms * 1e-3) / 1e9 if pair_ms > 0 else 0.0},
        }
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary_metrics(ctx, cfg, policy, rewards, ms_per_rollout, args.steps)
            try:
                out["secondary"]["config5_inverted_pendulum"] = config5_metrics(ctx, not args.no_cpu_baseline)
            except Exception as exc:   # never lose the headline line to a secondary measurement
                out["secondary"]["config5_inverted_pendulum"] = {"error": repr(exc)}
            try:   # the reference's Safe-PILCO loop (examples/safe_cars_run.py:41-140) on the HIP path: wall-clock of the whole loop
                import safe_cars
                from pilco_amd import _lib as _l
                prev = _l._default_ctx
                _l.set_context(ctx)
                try:
                    sc = safe_cars.run(iters=5, verbose=False)
                finally:
                    _l.set_context(prev)
                out["secondary"]["safe_cars_linear_loop"] = {
                    "hip_total_s": sc["total_s"], "predicted_risk_per_iteration": [i["predicted_risk"] for i in sc["iterations"]],
                    "optimize_policy_s_per_iteration": [i["optimize_policy_s"] for i in sc["iterations"]],
                    "note": "SafePILCO, RbfController(bf=40), horizon 25, 5 x [optimize_models, optimize_policy(maxiter=20, restarts=2), risk check, rollout]; "
                            "the risk term enters the policy gradient as cotangent seeds of the native reverse sweep"}
            except Exception as exc:
                out["secondary"]["safe_cars_linear_loop"] = {"error": repr(exc)}
            ctx.gp_set_data(0, cfg["X"], cfg["Y"])
            ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"])
        if world == 1 and "secondary" in out:
            # Every BASELINE config's number as a flat dict of scalars under a key the driver's record keeps (its parser drops
            # `secondary` and `verified`): configs 2 (R-fwd+fact), 4, 5 and R-grad on the driver's own box.
            sec = out["secondary"]
            c5 = sec.get("config5_inverted_pendulum") or {}
            its = c5.get("hip_iterations") or []
            lanes3 = ((sec.get("R_grad_C2u_lanes") or {}).get("B=3") or {})
            rgr = sec.get("R_grad_roofline") or {}
            num = lambda v: float(v) if isinstance(v, (int, float)) and np.isfinite(v) else None
            out["roofline"].update({   # flat scalar keys of `roofline` itself: the driver's parser keeps scalars only (a nested dict is dropped)
                "step_us": num(ms_per_rollout * 1e3 / H), "pair_us": num(pair_ms * 1e3), "non_pair_us_per_step": num(ms_per_rollout * 1e3 / H - pair_ms * 1e3),
                "rollout_level_frac": num(H * flop / (ms_per_rollout * 1e-3) / 1e12 / FP64_PEAK_TFLOPS),
                "back_to_back_per_s": num(sec.get("back_to_back_rollouts_per_s")),
                "R_grad_C2u_ms": num(sec.get("R_grad_C2u_ms")), "R_fwd_C2u_ms": num(sec.get("R_fwd_C2u_ms")),
                "R_grad_over_R_fwd": num(sec.get("R_grad_over_R_fwd_C2u")),
                "sweep_us": num(rgr.get("avg_launch_us")), "sweep_frac": num(rgr.get("frac")),
                "lanes_B3_ms_per_lane": num(lanes3.get("ms_per_lane")),
                "factorisation_ms": num(sec.get("factorisation_ms")), "nlml_eval_ms": num(sec.get("nlml_eval_ms")),
                "R_fwd_fact_per_s": num(sec.get("R_fwd_fact_rollouts_per_s")),
                "config4_per_s": num(sec.get("config4_rollouts_per_s")), "config4_fitc_ms": num(sec.get("config4_fitc_factorisation_ms")),
                "config4_fitc_objective_ms": num(sec.get("config4_fitc_objective_eval_ms")),
                "config5_loop_s": num(c5.get("hip_total_s")),
                "config5_optimize_policy_s": num(np.median([i["optimize_policy_s"] for i in its])) if its else None,
                "config5_optimize_models_s": num(np.median([i["optimize_models_s"] for i in its])) if its else None,
                "config5_speedup_first_iteration_vs_cpu_stand_in": num(c5.get("speedup_first_iteration")),
                "verified_max_rel_err": num(max(verified["max_rel_err"].values())) if verified else None,
            })
        if replicas is not None:
            out["secondary"] = replicas
            out["secondary"]["other_exchange"] = other_exchange
            out["secondary"]["per_rank"] = per_rank
            out["secondary"]["amdahl_model"] = amdahl
            out["secondary"]["rccl_comm_count"] = rccl_ranks
            out["secondary"]["ranks_on_distinct_gpus"] = not share_gpu
            num = lambda v: float(v) if isinstance(v, (int, float)) and np.isfinite(v) else None
            out["roofline"].update({   # the multi-rank run's scalars as flat keys of `roofline` (the driver's record keeps scalars only)
                "replica_rollouts_per_s": num(replicas.get("replica_rollouts_per_s")),
                "other_exchange_rollouts_per_s": num((other_exchange or {}).get("rollouts_per_s")),
                "one_gpu_ms_per_rollout": num((amdahl or {}).get("one_gpu_ms_per_rollout")),
                "one_gpu_head_us_per_step": num((amdahl or {}).get("one_gpu_head_us_per_step")),
                "slowest_rank_pair_us_per_launch": num((amdahl or {}).get("slowest_rank_pair_us_per_launch")),
                "predicted_rollouts_per_s_with_a_free_exchange": num((amdahl or {}).get("predicted_rollouts_per_s_with_a_free_exchange")),
                "exchange_and_skew_us_per_step": num((amdahl or {}).get("exchange_and_skew_us_per_step")),
                "rccl_comm_count": num(rccl_ranks), "peer_exchange": 1.0 if exchange.startswith("peer") else 0.0,
                "verified_max_rel_err": num(max(verified["max_rel_err"].values())) if verified else None,
            })
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
