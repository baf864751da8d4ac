#!/bin/bash
# Matrix-core utilisation of the one kernel that uses it (k_kf_update: C <- A C A' on v_mfma_f64_16x16x4_f64, kf_core.h):
# one rocprofv3 --pmc pass over the KF-mode run (tools/kf_run.py), kernel-trace only.  Writes gpurun_out/mfma_k_kf_update.json.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/mfma
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p1 -- python3 $R/tools/kf_run.py 65536 30 > $OUT/p1.log 2>&1 || echo "pass failed (see $OUT/p1.log)"
python3 - "$OUT" "$R" <<'PY'
import csv, glob, json, sys, collections
out, root = sys.argv[1], sys.argv[2]
per, dur = collections.defaultdict(list), []
for f in glob.glob(f"{out}/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_kf_update"):
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{out}/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_kf_update"):
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {k: sum(v[-10:]) / len(v[-10:]) for k, v in per.items()}
mops = res.get("SQ_INSTS_VALU_MFMA_MOPS_F64")
avg_us = sum(dur[-10:]) / max(len(dur[-10:]), 1) if dur else None
o = {"kernel": "k_kf_update", "batch": 65536, "per_launch_mean_last10": res, "avg_launch_us": avg_us,
     "note": "six v_mfma_f64_16x16x4_f64 per instance and step (two 9x9x9 products zero-padded into 16x16 tiles, three K chunks each); "
             "SQ_INSTS_VALU_MFMA_MOPS_F64 counts 512-flop units; the kernel is bound by its global-memory round trips, not by the matrix core"}
if mops and avg_us:
    o["mfma_f64_tflops_achieved"] = mops * 512 / (avg_us * 1e-6) / 1e12
    o["mfma_f64_peak_tflops"] = 78.6
    o["mfma_utilisation"] = o["mfma_f64_tflops_achieved"] / 78.6
json.dump(o, open(f"{root}/gpurun_out/mfma_k_kf_update.json", "w"), indent=1)
print(json.dumps(o, indent=1))
PY
