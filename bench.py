#!/usr/bin/env python3
"""bench.py -- moment-matching rollouts/sec on BASELINE.json's configuration
(N=1000, D=10, E=10, H=40; SURVEY.md section 8(d) synthetic recipe, seed 1234).

One "step" = one rollout = one PILCO.predict(m0, S0, H=40) (pilco.py:118-136) with
the GP factorisation cached on the device ("R-fwd" of SURVEY.md 8(d)); inputs are
resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W

N > 1: launched under torch.distributed.run, one rank per GPU.  torch is used only
to rendezvous (gloo: broadcast of the RCCL unique id, barrier, max-over-ranks);
the output pairs are sharded over the ranks inside libpilco_hip.so with one
ncclAllGather per horizon step ("scaling": "strong": one rollout is split).
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


def cpu_baseline(cfg):
    """Time the NumPy restatement of the GPflow CPU path (oracle/tf_path.py) on a
    bounded sample: one factorisation + 3 cached moment-matching steps."""
    from oracle import tf_path as tp
    t0 = time.perf_counter()
    iK, beta = tp.calculate_factorizations(cfg["X"], cfg["Y"], cfg["lengthscales"], cfg["variance"], cfg["noise"])
    t_fact = time.perf_counter() - t0
    m, s = cfg["m0"], cfg["S0"]
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        tp.predict_given_factorizations_pairs(cfg["X"], cfg["lengthscales"], cfg["variance"], m, s, iK, beta)
        ts.append(time.perf_counter() - t0)
    t_step = float(np.median(ts))
    return dict(value=1.0 / (H * t_step), unit="rollouts/s", cores=os.cpu_count(), kind="port",
                sample=("NumPy+OpenBLAS restatement of the GPflow path (oracle/tf_path.py, 55 symmetric pairs): "
                        "1 factorisation (%.2f s) + 3 moment-matching steps (median %.3f s); rollouts/s = "
                        "1/(40*t_step), factorisation cached; re-factorising every step as the reference does "
                        "(mgpr.py:77-79) would give %.4f rollouts/s" % (t_fact, t_step, 1.0 / (H * (t_step + t_fact)))))


def secondary_metrics(ctx, cfg, policy, rewards, ms_rollout):
    """The other two variants SURVEY.md 8(d) asks for, measured after the timed region (1 GPU):
    R-fwd+fact (factorise once, then roll out: what a fresh model pays) and R-grad (value + gradient of the
    rollout reward w.r.t. a linear controller at C2u: state 10 + 1 control, D=11)."""
    from pilco_amd import synthetic
    from pilco_amd.adjoint import rollout_value_and_grad
    from pilco_amd.models import PILCO
    out = {}
    fact_ms = ctx.factorize_timed(0, 5)
    out["factorisation_ms"] = fact_ms
    out["R_fwd_fact_rollouts_per_s"] = 1e3 / (fact_ms + ms_rollout)
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
    rollout_value_and_grad(p)
    t0 = time.perf_counter()
    for _ in range(3):
        rollout_value_and_grad(p)
    g_ms = (time.perf_counter() - t0) / 3 * 1e3
    p.compute_reward()
    t0 = time.perf_counter()
    for _ in range(3):
        p.compute_reward()
    f_ms = (time.perf_counter() - t0) / 3 * 1e3
    out["R_grad_C2u_ms"] = g_ms
    out["R_grad_C2u_per_s"] = 1e3 / g_ms
    out["R_fwd_C2u_ms"] = f_ms
    # restore the benchmark model in slot 0
    ctx.gp_set_data(0, cfg["X"], cfg["Y"])
    ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"])
    return out


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
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))

    from pilco_amd import _lib, synthetic
    cfg = synthetic.config_c2(N=N, D=D, E=E)
    ctx = _lib.Context(device=local_rank)

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
        id_t = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            id_t = torch.frombuffer(bytearray(ctx.comm_unique_id()), dtype=torch.uint8).clone()
        dist.broadcast(id_t, src=0)
        ctx.comm_init(bytes(id_t.numpy().tobytes()), rank, world)

    ctx.gp_set_data(0, cfg["X"], cfg["Y"])
    ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"])
    ctx.gp_factorize(0)
    policy = dict(kind=_lib.POLICY_NONE, state_dim=E, control_dim=0)
    rewards = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(E), t=np.zeros(E))]

    # warm-up (untimed)
    if args.warmup > 0:
        ctx.rollout_timed(policy, rewards, cfg["m0"], cfg["S0"], H, args.warmup, time_pair=False)

    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    res = ctx.rollout_timed(policy, rewards, cfg["m0"], cfg["S0"], H, args.steps, time_pair=False)  # syncs inside
    wall_ms = (time.perf_counter() - t0) * 1e3
    if dist is not None:
        dist.barrier()
        import torch
        t = torch.tensor([wall_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall_ms = float(t.item())
    ms_per_rollout = wall_ms / args.steps

    # dominant kernel (pair kernel): its own HIP-event pairs on the launch stream, separate pass
    prof = ctx.rollout_timed(policy, rewards, cfg["m0"], cfg["S0"], H, 1, time_pair=True)
    pair_ms = prof["ms_pair"] / max(prof["n_pair_launches"], 1)
    exps, flop, byts = algorithmic_work(N, D, E)
    flop_local = flop / world  # pairs are dealt over the ranks
    achieved = flop_local / (pair_ms * 1e-3) / 1e12 if pair_ms > 0 else 0.0

    # HBM bytes per pair-kernel launch: PMC counters of the committed rocprofv3 profile of this same
    # command (profiles/r01_pmc_summary.json: 2*FETCH_SIZE + WRITE_SIZE, see the calibration note there)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")) as f:
            traffic = json.load(f)["pair_kernel_hbm_bytes_per_launch"] / world
    except Exception:
        traffic = None

    # N > 1 only: the zero-communication alternative (SURVEY.md 8(e) "Alternative DP"): every rank also runs the whole
    # rollout on its own GPU (an unsharded second context); the aggregate is reported under "secondary", never as `value`.
    replicas = None
    if dist is not None:
        import torch
        local_ms, err = float("inf"), None
        try:   # no collective inside the try: a rank that fails must not leave the others waiting
            ctx2 = _lib.Context(device=local_rank)
            ctx2.gp_set_data(0, cfg["X"], cfg["Y"])
            ctx2.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"])
            ctx2.gp_factorize(0)
            ctx2.rollout_timed(policy, rewards, cfg["m0"], cfg["S0"], H, max(args.warmup, 1), time_pair=False)
            t0 = time.perf_counter()
            ctx2.rollout_timed(policy, rewards, cfg["m0"], cfg["S0"], H, args.steps, time_pair=False)
            local_ms = (time.perf_counter() - t0) * 1e3
        except Exception as exc:
            err = repr(exc)
        tt = torch.tensor([local_ms], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if np.isfinite(float(tt.item())):
            replicas = {"replica_rollouts_per_s": world * args.steps * 1e3 / float(tt.item()),
                        "note": "independent unsharded rollouts, one per GPU, concurrently, no communication (not the sharded metric)"}
        else:
            replicas = {"error": err or "a rank failed"}

    if rank == 0:
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
                                   "factorisation cached (R-fwd)",
                       "parallelism": "pairs%d" % world, "event_ms_per_rollout": res["ms_total"] / args.steps},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": "k_mm_pair_sk (f64 MFMA exponent tiles + fp64 exp; MFMA and fp64 VALU share one pipe)",
                         "avg_launch_ms": pair_ms,
                         "algorithmic_flop_per_launch": flop_local, "exp_per_launch": exps / world,
                         "gexp_per_s": exps / world / (pair_ms * 1e-3) / 1e9 if pair_ms > 0 else 0.0,
                         "algorithmic_bytes_per_launch": byts / world,
                         "hbm_GBps_algorithmic": byts / world / (pair_ms * 1e-3) / 1e9 if pair_ms > 0 else 0.0},
        }
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary_metrics(ctx, cfg, policy, rewards, ms_per_rollout)
        if replicas is not None:
            out["secondary"] = replicas
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
