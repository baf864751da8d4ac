#!/bin/bash
# Regenerates every artefact under profiles/ that DESIGN.md quotes, on ONE GPU box, into gpurun_out/final/:
#   bench.json                    python bench.py (default flags, with cpu_baseline)
#   bench_with_pipelined_leg.json python bench.py --no-cpu-baseline --pipelined-leg (the line + the same steps with solve_pipeline = 1)
#   kernel_stats.csv              rocprofv3 --kernel-trace --stats of bench.py --steps 100 --warmup 30
#   bench_under_rocprof.json      bench.py's own line in that profiled run
#   kernel_trace_timed.json       average of the 100 timed solve launches from the kernel trace
#   sections.txt                  in-kernel section shares (needs csrc/libdekf_prof.so, see profile_sections.py)
#   sq_counters.json, traffic.json  PMC passes (tools/collect_sq.sh, tools/collect_traffic.sh)
#   mfma_k_kf_update.json         matrix-core counters of the KF kernel (tools/collect_mfma.sh)
# usage (from the repo root on the GPU box):  bash tools/final_profiles.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
timeout 900 python3 bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json
timeout 900 python3 bench.py --no-cpu-baseline --pipelined-leg 2> $OUT/bench_pl.err | tail -1 > $OUT/bench_with_pipelined_leg.json
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 100 --warmup 30 --no-cpu-baseline > $OUT/prof_stdout.log 2>&1 )
grep "^{\"metric\"" $OUT/prof_stdout.log | tail -1 > $OUT/bench_under_rocprof.json
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
st = glob.glob(f"{out}/prof/**/*kernel_stats.csv", recursive=True)
if st:
    open(f"{out}/kernel_stats.csv", "w").write(open(st[0]).read())
tr = glob.glob(f"{out}/prof/**/*kernel_trace.csv", recursive=True)
if tr:
    rows = list(csv.DictReader(open(tr[0])))
    def dur(name):
        return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if r["Kernel_Name"].startswith(name)]
    solve_name = [r["Kernel_Name"] for r in rows if r["Kernel_Name"].startswith("k_mhe_solve")][-1]
    sv, asm = dur("k_mhe_solve"), dur("k_mhe_assemble")
    res = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 30 --no-cpu-baseline",
           "kernel": solve_name, "launches_total": len(sv), "avg_ms_all_launches": sum(sv) / len(sv),
           "avg_ms_last_100_launches_(the_timed_region)": sum(sv[-100:]) / 100, "min_ms_timed": min(sv[-100:]), "max_ms_timed": max(sv[-100:]),
           "note": "the first launches fill the window (K < 20) and are shorter; bench.py's roofline.avg_launch_ms covers the timed launches only",
           "k_mhe_assemble_avg_ms_timed": sum(asm[-100:]) / 100}
    json.dump(res, open(f"{out}/kernel_trace_timed.json", "w"), indent=1)
    print(json.dumps(res))
PY
if [ -f decentralized_ekf_mhe_amd/csrc/libdekf_prof.so ]; then
  DEKF_LIB=$R/decentralized_ekf_mhe_amd/csrc/libdekf_prof.so timeout 600 python3 tools/profile_sections.py 4096 70 > $OUT/sections.txt 2>&1
fi
bash tools/collect_sq.sh > $OUT/sq_stdout.log 2>&1; cp gpurun_out/sq_k_mhe_solve.json $OUT/sq_counters.json 2>/dev/null
bash tools/collect_traffic.sh > $OUT/traffic_stdout.log 2>&1; cp gpurun_out/traffic_k_mhe_solve.json $OUT/traffic.json 2>/dev/null
bash tools/collect_mfma.sh > $OUT/mfma_stdout.log 2>&1; cp gpurun_out/mfma_k_kf_update.json $OUT/mfma_k_kf_update.json 2>/dev/null
timeout 900 python3 tools/stress_parity.py > $OUT/stress_parity.jsonl 2> $OUT/stress.err
timeout 900 python3 tools/bench_shapes.py > $OUT/bench_shapes.jsonl 2> $OUT/shapes.err
ls -la $OUT
