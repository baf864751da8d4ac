#!/bin/bash
# A/B of any bench_shapes shape on ONE box: alternates candidate libraries (csrc/libdekf_<tag>.so).
# usage: tools/ab_shape.sh SHAPE ROUNDS tag1 tag2 ...    SHAPE: go1foot | pogox | cassie | go1 | go1pol   (prints: tag shape steps/s ms/step)
S=$1; R=$2; shift 2
for r in $(seq 1 $R); do for t in "$@"; do
  DEKF_LIB=decentralized_ekf_mhe_amd/csrc/libdekf_$t.so timeout -k 10 300 python - "$S" "$t" <<'PY'
import json, sys, io, contextlib
sys.path.insert(0, "tools")
import bench_shapes as bs
from decentralized_ekf_mhe_amd import cassie_params, go1_params, pogox_params
shape, tag = sys.argv[1], sys.argv[2]
cases = {"go1foot": ("go1foot", go1_params, 4096, 30, dict(leg_odom_type=1)), "pogox": ("pogox", pogox_params, 1024, 40, {}),
         "cassie": ("cassie", cassie_params, 4096, 60, {}), "go1": ("go1", go1_params, 4096, 60, {}), "go1pol": ("go1pol", go1_params, 4096, 60, dict(polish=1))}
name, maker, B, steps, kw = cases[shape]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bs.run(name, maker, B, steps, **kw)
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(tag, shape, round(d["estimator_steps_per_s"]), round(d["ms_per_step"], 4), d["mean_iters"], d["solved_frac"], flush=True)
PY
done; done
