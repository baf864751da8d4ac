#!/bin/bash
# Issue/stall breakdown of the dominant kernel (k_mhe_solve_*) from rocprofv3 SQ counters: three
# separate --pmc passes (8 SQ slots each), kernel-trace only, program itself after `--`.
# Writes gpurun_out/sq_k_mhe_solve.json (mean over the last 6 launches = steady state).
#   tools/collect_sq.sh [shape]     shape: as tools/collect_traffic.sh (default: Go1 through bench.py); with a shape the file is
#                                   gpurun_out/sq_<kernel>.json
SHAPE=${1:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/sq
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES"
P3="SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  if [ -z "$SHAPE" ]; then
    rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 6 --warmup 24 --no-cpu-baseline > $OUT/p$i.log 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
  else
    rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_shapes.py $SHAPE 6 > $OUT/p$i.log 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
  fi
done
python3 - "$OUT" "$R" "$SHAPE" <<'PY'
import csv, glob, json, sys, collections
out, root, shape = sys.argv[1], sys.argv[2], sys.argv[3]
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
short = kname.split("(")[0].replace(".kd", "") if kname else "none"
w = res.get("SQ_WAVE_CYCLES", 0.0)
derived = {"wait_any_over_wave_cycles": res.get("SQ_WAIT_ANY", 0.0) / w if w else None,
           "lds_bank_conflict_over_lds_active": (res.get("SQ_LDS_BANK_CONFLICT", 0.0) / res["SQ_LDS_IDX_ACTIVE"]) if res.get("SQ_LDS_IDX_ACTIVE") else None}
json.dump({"kernel": kname, "shape": shape or "go1 (bench.py)", "per_launch_mean_last6": res, "derived": derived},
          open(f"{root}/gpurun_out/sq_{short if shape else 'k_mhe_solve'}.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
