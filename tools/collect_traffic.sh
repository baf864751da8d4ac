#!/bin/bash
# HBM traffic of the dominant kernel (k_mhe_solve_*) from rocprofv3 PMC counters, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots),
# kernel-trace only, program itself after `--`; FETCH_SIZE doubled on gfx950 (it tallies 128-B
# requests at 64 B for wide coalesced reads — our 8-B accesses are outside the calibrated shape, so
# the figure is an upper bound on the read side).
#   tools/collect_traffic.sh            Go1 through bench.py        -> gpurun_out/traffic_k_mhe_solve.json
#   tools/collect_traffic.sh <shape>    another BASELINE shape through tools/bench_shapes.py <shape> (cassie | pogox | go1foot | go1_8192)
#                                       -> gpurun_out/traffic_<kernel>.json   (copy to profiles/ to have bench_shapes.py quote it)
set -e
SHAPE=${1:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/traffic
rm -rf $OUT          # (a second run in one call used to pick up the first run's counter file)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  if [ -z "$SHAPE" ]; then
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 6 --warmup 24 --no-cpu-baseline > $OUT/$c.log 2>&1
  else
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/tools/bench_shapes.py $SHAPE 6 > $OUT/$c.log 2>&1
  fi
done
python3 - "$OUT" "$R" "$SHAPE" <<'PY'
import csv, glob, json, os, sys
out, root, shape = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, root)
sys.path.insert(0, root + "/tools")
import bench
batch = 4096
if shape:
    import bench_shapes
    batch = bench_shapes.SHAPES[shape][2]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
    rows = [r for r in csv.DictReader(open(f)) if "k_mhe_solve" in r["Kernel_Name"] and r["Counter_Name"] == c]
    last = rows[-6:]                      # the timed, steady-state launches
    vals[c] = sum(float(r["Counter_Value"]) for r in last) / len(last)
fetch_b, write_b = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
kernel = last[-1]["Kernel_Name"].split("(")[0].replace(".kd", "")
res = {"kernel": kernel, "batch": batch, "shape": shape or "go1 (bench.py)", "csrc_sha1": bench.csrc_sha1(), "collected": "tools/collect_traffic.sh, rocprofv3 --pmc, separate passes", "FETCH_SIZE_KB_raw": vals["FETCH_SIZE"], "WRITE_SIZE_KB_raw": vals["WRITE_SIZE"],
       "fetch_bytes_corrected_x2": fetch_b, "write_bytes": write_b, "hbm_bytes_per_launch": fetch_b + write_b,
       "note": "separate --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950); 8-B accesses are uncalibrated"}
json.dump(res, open(f"{root}/gpurun_out/traffic_{kernel if shape else 'k_mhe_solve'}.json", "w"), indent=1)
print(json.dumps(res))
PY
