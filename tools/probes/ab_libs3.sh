#!/bin/bash
# like ab_libs.sh for any number of libraries:  tools/probes/ab_libs3.sh "libA.so libB.so libC.so" [batch] [reps]
LIBS=$1; b=${2:-4096}; REPS=${3:-3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for rep in $(seq $REPS); do for lib in $LIBS; do
DEKF_LIB=$R/decentralized_ekf_mhe_amd/csrc/$lib python $R/bench.py --batch $b --steps 100 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', $b, round(d['value']), round(d['ms_per_step'],4), round(d['kernel_ms_per_step']['solve'],4), round(d['kernel_ms_per_step']['assemble'],4))"
done; done
