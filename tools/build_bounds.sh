#!/bin/bash
# Builds the bounds-checked diagnostic libraries (csrc/libdekf_bounds_<set>.so, -DDEKF_BOUNDS: wave.h BPtr), one kernel set per
# library, in parallel.  With checked pointers a single translation unit of all solve kernels takes hours to compile; one set takes
# minutes.  usage: tools/build_bounds.sh [set ...]   sets: go1 cassie legs1 legs2 legs3 legs4 foot1 foot2 foot3 foot4
cd "$(dirname "$0")/../decentralized_ekf_mhe_amd/csrc"
declare -A MASK=( [go1]=1 [cassie]=2 [legs1]=4 [legs2]=8 [legs3]=16 [legs4]=32 [foot1]=64 [foot2]=128 [foot3]=256 [foot4]=512 )
SETS=${@:-go1 cassie legs1 legs2 legs3 legs4 foot1 foot2 foot3 foot4}
for s in $SETS; do
  ( t0=$(date +%s); DEKF_OUT=libdekf_bounds_$s.so DEKF_UNITY=1 timeout -k 10 ${BOUNDS_BUILD_TIMEOUT:-1500} bash build.sh -DDEKF_BOUNDS -DDEKF_KSET=${MASK[$s]} > /tmp/bounds_$s.log 2>&1
    echo "$s: rc=$? $(( $(date +%s) - t0 )) s $(grep -c error /tmp/bounds_$s.log) errors" ) &
done
wait
