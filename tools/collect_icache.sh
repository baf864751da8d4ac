#!/bin/bash
# Instruction-cache behaviour of the solve kernel (SQC counters), one --pmc pass, kernel-trace only.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/p -- python3 $R/bench.py --steps 6 --warmup 24 --no-cpu-baseline > $OUT/p.log 2>&1 || echo "pass failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(f"{out}/p/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_mhe_solve" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(per.items()):
        print(k, sum(v[-6:]) / len(v[-6:]))
PY
