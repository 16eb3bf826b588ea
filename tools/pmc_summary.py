#!/usr/bin/env python3
"""Build profiles/<round>_pmc_summary.json, _kernel_stats.csv and _bench.json from the rocprofv3 passes that
tools/profile_round.sh leaves under gpurun_out/profiles/.   usage: python tools/pmc_summary.py r01"""
import csv, json, os, shutil, sys
from collections import defaultdict

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = "gpurun_out/profiles"
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = os.path.join(src, d, "r_counter_collection.csv")
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
kern = {k: {c: v[1] / v[0] for c, v in cs.items()} for k, cs in acc.items()}
pair = next((k for k in kern if "k_mm_pair_sk" in k), None)
out = {
    "command": "python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary (rocprofv3 --kernel-trace --pmc <one counter group per pass>; tools/profile_round.sh)",
    "note": ("per-launch averages. FETCH_SIZE / WRITE_SIZE are KiB. Calibration on this access pattern (8 B/lane coalesced): "
             "k_matvec (non-transposed pass) reads 81920 KiB and reports FETCH_SIZE of about half that (the gfx950 under-count of "
             "MI355X_MICROARCH.md section HBM); k_gram writes 81920 KiB and reports WRITE_SIZE 81920 (1:1). "
             "Corrected HBM traffic = 2*FETCH_SIZE + WRITE_SIZE."),
    "kernels": kern,
}
if pair:
    p = kern[pair]
    out["pair_kernel"] = pair
    out["pair_kernel_hbm_bytes_per_launch"] = (2.0 * p.get("FETCH_SIZE", 0.0) + p.get("WRITE_SIZE", 0.0)) * 1024.0
    if p.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES and SQ_ACTIVE_INST_VALU (x4 cycles per
        # wave64 instruction) are summed over the 1024 SIMDs
        simd_cycles = p["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        out["pair_kernel_mfma_busy_frac"] = p.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles
        out["pair_kernel_valu_busy_frac"] = 4.0 * p.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles
    if "SQ_BUSY_CYCLES" in p and p.get("SQ_WAVE_CYCLES"):
        out["pair_kernel_valu_active_over_wave_cycles"] = p.get("SQ_ACTIVE_INST_VALU", 0.0) / p["SQ_WAVE_CYCLES"]
        out["pair_kernel_wait_inst_over_wave_cycles"] = p.get("SQ_WAIT_INST_ANY", 0.0) / p["SQ_WAVE_CYCLES"]
# provenance: which sources the profiled binary was built from (bench.py warns when the pair kernel has changed since)
import datetime, hashlib, subprocess
def _sha16(path):
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None
out["kernel_source_sha16"] = {n: _sha16(os.path.join("pilco_amd", "csrc", n))
                              for n in ("pair.hip", "pair_device.h", "prep.hip", "prep_device.h", "prep_kernel.h", "glue_device.h", "mm_device.h",
                                        "rollout.hip", "bwd.hip", "linalg.hip")}
out["date"] = datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%MZ")
try:
    out["git_head"] = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() + \
        ("+dirty" if subprocess.run(["git", "status", "--porcelain", "pilco_amd/csrc"], capture_output=True, text=True).stdout.strip() else "")
except Exception:
    out["git_head"] = None
os.makedirs("profiles", exist_ok=True)
json.dump(out, open("profiles/%s_pmc_summary.json" % rnd, "w"), indent=1)
shutil.copy(os.path.join(src, "trace", "r_kernel_stats.csv"), "profiles/%s_kernel_stats.csv" % rnd)
line = open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]
json.dump(json.loads(line), open("profiles/%s_bench.json" % rnd, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.startswith("pair_kernel")}, indent=1))
if pair:
    print(json.dumps(kern[pair], indent=1))
