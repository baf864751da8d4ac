#!/bin/bash
# Where do the operand requests of a solve kernel go?  L1 / L2 / fabric / DRAM request counters of k_mhe_solve_* in separate
# rocprofv3 --pmc passes (kernel-trace only, program itself after `--`), mean over the last 6 launches.
#   tools/collect_memsys.sh [harness.py args...]      default harness: tools/emu_foot.py 75 (leg_odom_type 1, 75 forced iterations)
# Writes gpurun_out/memsys_k_mhe_solve.json.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/memsys
rm -rf $OUT; mkdir -p $OUT
H=${@:-tools/emu_foot.py 75}
cd /tmp && export TMPDIR=/tmp
P1="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum"
P2="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"
# (a pass with TA_* counters — TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum — aborted
# inside rocprofv3 on this image and left the run hanging until the box's watchdog: not collected)
P3="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  ( cd $R && timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- python3 $H > $OUT/p$i.log 2>&1 ) || echo "pass $i failed (see $OUT/p$i.log)"
done
python3 - "$OUT" "$R" "$H" <<'PY'
import csv, glob, json, sys, collections
out, root, harness = sys.argv[1], sys.argv[2], sys.argv[3]
res, kname, dur = {}, None, []
for f in sorted(glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_mhe_solve" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kname = r["Kernel_Name"]
    for k, v in per.items():
        res[k] = sum(v[-6:]) / len(v[-6:])
for f in sorted(glob.glob(f"{out}/p1/**/*kernel_trace.csv", recursive=True)):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_mhe_solve" in r["Kernel_Name"]]
    dur = d[-6:]
json.dump({"kernel": kname, "harness": harness, "launch_ns_under_pmc_pass1": sum(dur) / max(len(dur), 1), "per_launch_mean_last6": res},
          open(f"{root}/gpurun_out/memsys_k_mhe_solve.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
