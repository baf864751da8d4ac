#!/bin/bash
# A/B/C... on ONE box: alternates any number of candidate libraries (csrc/libdekf_<tag>.so) through bench.py at a given batch.
# usage: tools/ab_variants.sh BATCH ROUNDS tag1 tag2 ...     (prints: tag batch solve-kernel-ms value)
B=$1; R=$2; shift 2
for r in $(seq 1 $R); do for t in "$@"; do
  DEKF_LIB=decentralized_ekf_mhe_amd/csrc/libdekf_$t.so timeout -k 10 300 python bench.py --batch $B --steps 40 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', $B, round(d['kernel_ms_per_step']['solve'],4), round(d['value']), flush=True)"
done; done
