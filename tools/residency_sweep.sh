#!/bin/bash
# Per-solve time of the three-workgroup kernel at a forced residency of 1, 2, 3 workgroups per CU and several batches.
# Needs an experiment build: DEKF_OUT=libdekf_r3x.so bash csrc/build.sh -DDEKF_GO1_ONLY -DDEKF_X_ALWAYS_R3
for cap in 1 2 3; do for b in 128 256 512 768 4096; do
DEKF_X_R3_CAP=$cap DEKF_LIB=$PWD/decentralized_ekf_mhe_amd/csrc/libdekf_r3x.so python bench.py --batch $b --steps 60 --warmup 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cap',$cap,'B',$b, round(d['value']), 'solve ms', round(d['kernel_ms_per_step']['solve'],4), 'wgs', d['roofline']['solve_workgroups'], d['roofline']['kernel'])"
done; done
