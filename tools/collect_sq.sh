#!/bin/bash
# Issue/stall breakdown of the dominant kernel (k_mhe_solve_*) from rocprofv3 SQ counters: three
# separate --pmc passes (8 SQ slots each), kernel-trace only, program itself after `--`.
# Writes gpurun_out/sq_k_mhe_solve.json (mean over the last 6 launches = steady state).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES"
P3="SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 6 --warmup 24 --no-cpu-baseline > $OUT/p$i.log 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
done
python3 - "$OUT" "$R" <<'PY'
import csv, glob, json, sys, collections
out, root = sys.argv[1], sys.argv[2]
res = {}
kname = None
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_mhe_solve" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kname = r["Kernel_Name"]  # the last launches are the full-window kernel (the first N - 1 ticks fill the window)
    for k, v in per.items():
        res[k] = sum(v[-6:]) / len(v[-6:])
json.dump({"kernel": kname, "batch": 4096, "per_launch_mean_last6": res}, open(f"{root}/gpurun_out/sq_k_mhe_solve.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
