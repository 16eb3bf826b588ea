#!/usr/bin/env python3
"""profiles/<round>_grad_kernel_stats.csv and _grad_pmc_summary.json from the passes tools/profile_grad.sh leaves under
gpurun_out/grad/ (value + gradient at C2u; the dominant kernel is the reverse sweep k_mm_bwd_pair).
usage: python tools/grad_pmc_summary.py r02"""
import csv, json, os, shutil, sys
from collections import defaultdict

rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = "gpurun_out/grad"
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for d in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write"):
    f = os.path.join(src, d, "r_counter_collection.csv")
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
kern = {k: {c: v[1] / v[0] for c, v in cs.items()} for k, cs in acc.items() if "bwd" in k or "jac" in k or "prep" in k or "rev_" in k}
out = {"command": "python tools/c2u_bench.py (rocprofv3 --kernel-trace --pmc <one counter group per pass>; tools/profile_grad.sh)",
       "note": "per-launch averages; GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_* over the 1024 SIMDs; FETCH_SIZE / WRITE_SIZE in KiB "
               "(corrected HBM traffic = 2*FETCH_SIZE + WRITE_SIZE, see the forward summary)",
       "kernels": kern}
sw = next((k for k in kern if "k_mm_bwd_pair" in k), None)
if sw:
    p = kern[sw]
    out["sweep_kernel"] = sw
    if p.get("GRBM_GUI_ACTIVE"):
        simd_cycles = p["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        out["sweep_mfma_busy_frac"] = p.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles
        out["sweep_valu_busy_frac"] = 4.0 * p.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles
    if p.get("SQ_WAVE_CYCLES"):
        out["sweep_wait_inst_over_wave_cycles"] = p.get("SQ_WAIT_INST_ANY", 0.0) / p["SQ_WAVE_CYCLES"]
        out["sweep_wait_lds_over_wave_cycles"] = p.get("SQ_WAIT_INST_LDS", 0.0) / p["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in p:
        out["sweep_hbm_bytes_per_launch"] = (2.0 * p.get("FETCH_SIZE", 0.0) + p.get("WRITE_SIZE", 0.0)) * 1024.0
os.makedirs("profiles", exist_ok=True)
json.dump(out, open("profiles/%s_grad_pmc_summary.json" % rnd, "w"), indent=1)
shutil.copy(os.path.join(src, "trace", "r_kernel_stats.csv"), "profiles/%s_grad_kernel_stats.csv" % rnd)
print(json.dumps({k: v for k, v in out.items() if k.startswith("sweep")}, indent=1))
