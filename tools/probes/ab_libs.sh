#!/bin/bash
# A/B of two builds of the library on one box: bench.py at several batch sizes, alternating.   tools/probes/ab_libs.sh libA.so libB.so "4096 7680 8192" [reps]
A=$1; B=$2; SIZES=${3:-4096}; REPS=${4:-2}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for b in $SIZES; do for rep in $(seq $REPS); do for lib in $A $B; do
DEKF_LIB=$R/decentralized_ekf_mhe_amd/csrc/$lib python $R/bench.py --batch $b --steps 100 --warmup 50 --no-cpu-baseline --no-pipelined-leg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', $b, round(d['value']), round(d['ms_per_step'],4), round(d['kernel_ms_per_step']['solve'],4), round(d['kernel_ms_per_step']['assemble'],4))"
done; done; done
