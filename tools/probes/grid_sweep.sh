#!/bin/bash
# Persistent grid of the Go1 three-workgroup solve kernel swept through an env override of a probe build (libdekf_grid.so: DEKF_X_GRID replaces
# solve_grid_full in dekf_create): fewer workgroups than the 768 slots, and oversubscribed grids (the hardware dispatcher then schedules dynamically).
for g in 768 760 752 736 704 640 576 1024 1536 2048 4096 768; do
DEKF_X_GRID=$g DEKF_LIB=$PWD/decentralized_ekf_mhe_amd/csrc/libdekf_grid.so python bench.py --steps 100 --warmup 50 --no-cpu-baseline --no-pipelined-leg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid', $g, d['roofline']['solve_workgroups'], round(d['value']), round(d['ms_per_step'],4), round(d['kernel_ms_per_step']['solve'],4), d['solver']['solved_frac'])"
done
