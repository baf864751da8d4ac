#!/bin/bash
# HBM traffic of the dominant kernel (k_mhe_solve_*) from rocprofv3 PMC counters, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots),
# kernel-trace only, program itself after `--`; FETCH_SIZE doubled on gfx950 (it tallies 128-B
# requests at 64 B for wide coalesced reads — our 8-B accesses are outside the calibrated shape, so
# the figure is an upper bound on the read side).  Writes profiles/traffic_k_mhe_solve.json.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/traffic
rm -rf $OUT          # (a second run in one call used to pick up the first run's counter file)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 6 --warmup 24 --no-cpu-baseline > $OUT/$c.log 2>&1
done
python3 - "$OUT" "$R" <<'PY'
import csv, glob, json, os, sys
out, root = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
import bench
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
    rows = [r for r in csv.DictReader(open(f)) if "k_mhe_solve" in r["Kernel_Name"] and r["Counter_Name"] == c]
    last = rows[-6:]                      # the timed, steady-state launches
    vals[c] = sum(float(r["Counter_Value"]) for r in last) / len(last)
fetch_b, write_b = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
res = {"kernel": last[-1]["Kernel_Name"].split("(")[0].replace(".kd", ""), "batch": 4096, "csrc_sha1": bench.csrc_sha1(), "collected": "tools/collect_traffic.sh, rocprofv3 --pmc, separate passes", "FETCH_SIZE_KB_raw": vals["FETCH_SIZE"], "WRITE_SIZE_KB_raw": vals["WRITE_SIZE"],
       "fetch_bytes_corrected_x2": fetch_b, "write_bytes": write_b, "hbm_bytes_per_launch": fetch_b + write_b,
       "note": "separate --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950); 8-B accesses are uncalibrated"}
json.dump(res, open(f"{root}/gpurun_out/traffic_k_mhe_solve.json", "w"), indent=1)
print(json.dumps(res))
PY
