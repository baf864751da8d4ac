#!/bin/bash
# A/B on ONE box: alternate the candidate libraries (csrc/libdekf_<tag>.so), several rounds each;
# prints the solve-kernel ms per step of every run.  usage: tools/ab_bench.sh A B [rounds]
R=${3:-3}
for r in $(seq 1 $R); do for t in $1 $2; do
  DEKF_LIB=decentralized_ekf_mhe_amd/csrc/libdekf_$t.so timeout 300 python bench.py --steps 60 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', round(d['kernel_ms_per_step']['solve'],4), round(d['value']))"
done; done
