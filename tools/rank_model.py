"""What ONE GPU can say about BASELINE config 3 (E = 10 outputs / 55 pairs sharded over W ranks): rank 0's OWN kernels of one
horizon step -- operand kernel, pair kernel, pack -- with pilco_shard_set(0, W), alone on the GPU (the other ranks' contexts
exist only to factorise their outputs and hand over their beta rows; they launch nothing in the measured loop).

    rocprofv3 --kernel-trace --output-format csv -d <dir>/W<W>v<variant> -o r -- python tools/rank_model.py run <W> <variant>
    python tools/rank_model.py summarise <dir-of-W1> <dir-of-W2> ... > profiles/r06_rank_model.json

The summary is a MODEL, not a measurement of xGMI: step(W) = [fused head of the 1-rank run] - [operand kernel(1) - operand
kernel(W)] + pair(W) + pack(W), with a free exchange."""
import csv, glob, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(W, variant):
    from pilco_amd import _lib, synthetic
    cfg = synthetic.config_c2()
    ctxs = []
    for r in range(W):
        cx = _lib.Context(device=0)
        if W > 1:
            cx.set_pair_kernel(variant)   # 0: stream-K (the fast one; its sums depend on the rank's pair count), 2: tiled (bit-identical across rank counts)
            cx.shard_set(r, W)
        cx.gp_set_data(0, cfg["X"], cfg["Y"]); cx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); cx.gp_factorize(0)
        ctxs.append(cx)
    if W > 1:
        _lib.group_sync_model(ctxs)
    m, s = cfg["m0"], cfg["S0"]
    for _ in range(30):
        ctxs[0].shard_pack(0, m, s, 10, 10, W, 0)
    if W == 1:   # the fused head and the stream-K pair kernel of the single-rank rollout, for the model's base line
        pol = dict(kind=_lib.POLICY_NONE, state_dim=10, control_dim=0)
        rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
        for _ in range(10):
            ctxs[0].rollout(pol, rw, m, s, 40)
    for cx in ctxs:
        cx.close()


def kernel_avgs(d):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0]
            acc.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc


def summarise(dirs):
    out = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "model": "step(W) = fused_head(1) - (operand(1) - operand(W)) + pair(W) + pack(W); exchange free; NOT a measurement of xGMI",
           "config": "N=1000 D=10 E=10 (55 pairs), rank 0 of W, both pair kernels on several ranks", "ranks": {}}
    base = None
    for d in dirs:
        tag = os.path.basename(os.path.normpath(d))
        W, variant = int(tag.split("W")[-1].split("v")[0]), int(tag.split("v")[-1])
        a = kernel_avgs(d)
        pick = lambda key, last: next((float(np.median(v[-last:])) for k, v in a.items() if key in k), None)
        row = {"operand_kernel_us": pick("k_mm_prep<10, false", 25), "pack_us": pick("k_glue", 25),
               "pair_kernel_us": pick("k_mm_pair_tiled", 25) if (W > 1 and variant == 2) else pick("k_mm_pair_sk", 25 if W > 1 else 400),
               "pair_kernel": "tiled (bit-identical across rank counts; the default on several ranks)" if (W > 1 and variant == 2) else "stream-K",
               "pairs_of_rank0": (55 - 0 + W - 1) // W}
        if W == 1:
            row["pair_kernel_tiled_us"] = None
            row["fused_head_us"] = pick("k_mm_prep<10, true", 400)
            base = row
        out["ranks"]["W=%d" % W if W == 1 else "W=%d %s" % (W, "tiled" if variant == 2 else "stream-K")] = row
    if base:
        s1 = base["fused_head_us"] + base["pair_kernel_us"]
        for k, row in out["ranks"].items():
            if k == "W=1":
                row["model_step_us"] = s1
                continue
            row["model_step_us"] = base["fused_head_us"] - (base["operand_kernel_us"] - row["operand_kernel_us"]) + row["pair_kernel_us"] + row["pack_us"]
            row["model_speedup_of_one_rollout"] = s1 / row["model_step_us"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 2)
    else:
        summarise(sys.argv[2:])
