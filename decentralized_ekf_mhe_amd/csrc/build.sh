#!/bin/bash
# Builds libdekf.so (the C-ABI shared library, gfx950 only) in-tree.
#   bash build.sh                                            the product: every kernel, compiled as 12 translation units in parallel
#   DEKF_OUT=libdekf_prof.so DEKF_UNITY=1 bash build.sh -DDEKF_PROFILE -DDEKF_GO1_ONLY     diagnostic / A-B variants under another name
# Product build: kernels.hip is compiled once per kernel set (-DDEKF_KSET=<bit> -DDEKF_KSET_ONLY: the unit carries only its set;
# bit 1024 = the kernels that are not solves) next to dekf_capi.hip, DEKF_JOBS units at a time (default: all of them at once — they
# are independent, no -fgpu-rdc: a kernel is launched through its host stub, which any unit can reference), then linked.  One
# translation unit of everything takes 4.6 minutes, the slowest set 70 s.
# DEKF_UNITY=1 compiles both sources as ONE translation unit (variants that select their kernels with -DDEKF_KSET / -DDEKF_GO1_ONLY,
# the -DDEKF_BOUNDS build with its device-side counter).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${DEKF_OUT:-libdekf.so}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include -I$(pwd)"
if [ -n "$DEKF_UNITY" ]; then
    U=$(mktemp /tmp/dekf_unity_XXXXXX.hip)
    printf '#include "%s/kernels.hip"\n#include "%s/dekf_capi.hip"\n' "$(pwd)" "$(pwd)" > "$U"
    $HIPCC $FLAGS -shared -o "$OUT" "$U" -ldl "$@"
    rm -f "$U"
else
    T=$(mktemp -d /tmp/dekf_build_XXXXXX)
    trap 'rm -rf "$T"' EXIT
    JOBS=${DEKF_JOBS:-12}
    pids=()
    run() { "$@" & pids+=($!); while [ "$(jobs -rp | wc -l)" -ge "$JOBS" ]; do sleep 0.2; done; }
    # heaviest sets first (foot-state kernels, then the generic ones)
    # DEKF_NO_MLICM: the kernel sets compiled with LLVM's machine-level loop-invariant code motion off (default: every set of solve
    # kernels; the set of the small kernels, 1024, keeps the pass).  That pass hoists constant materialisations and address
    # arithmetic to the top of a kernel, where they stay live across the whole solve: at the 168 VGPRs of the three-workgroup
    # kernels the register allocator answered with 43 spilled VGPRs (7 without the pass), i.e. scratch traffic beyond L2 (DESIGN.md
    # section 6); the rows-in-registers kernel of set 4 (k_mhe_solve_rr_1, 256 VGPRs) spills 40 VGPRs inside its iteration loops with
    # the pass and none without; the foot-state kernels need 205 instead of 256 VGPRs and run 2 % faster.  Mid-round, hipcc 7.2 died
    # on two sets without the pass ("Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base"; it compiles all of them
    # now): a set that dies with the flag is retried without it below, and tests/test_resource_usage.py then says what that cost.
    # Default: the sets with MEASURED gains — 1, 2 (three-workgroup kernels), 4 (rows in registers), 64..512 (foot-state kernels); the
    # generic 2-4-leg sets (8, 16, 32: 179-207 VGPRs, no spills either way) keep the pass.  "none" for A/B builds, "all" for every set.
    NO_MLICM=${DEKF_NO_MLICM:-"1 2 4 64 128 256 512"}
    # Every unit is compiled with -Rpass-analysis=kernel-resource-usage: the compiler's own account of each kernel (VGPRs, spills,
    # scratch, occupancy) goes to resource_usage.txt next to the library, and tests/test_resource_usage.py fails when a benchmark
    # kernel leaves its design point (three workgroups per CU at <= 16 spilled VGPRs, the rows-in-registers kernel spill-free).
    # A set whose compile dies WITH the hidden flag is retried without it (and named): the build survives a compiler that rejects
    # the flag, and the test then says what that cost.
    unit() {  # unit <mask> <flags...>
        local m=$1; shift
        if ! $HIPCC $FLAGS "$@" -Rpass-analysis=kernel-resource-usage -c -DDEKF_KSET=$m -DDEKF_KSET_ONLY -o "$T/k_$m.o" kernels.hip $EXTRA 2> "$T/ru_$m.txt"; then
            # Retry without the hidden flag ONLY for the known compiler crash (its signature below); anything else — a genuine source
            # error that might happen to compile without the flag — fails the build with the diagnostics of THIS attempt.  The first
            # attempt's output is kept (first_$m.txt) and its non-remark lines go into the notes of the resource-usage file.
            case " $* " in *" -disable-machine-licm "*)
                if grep -q "Illegal instruction detected\|LLVM ERROR\|PLEASE submit a bug report" "$T/ru_$m.txt"; then
                    cp "$T/ru_$m.txt" "$T/first_$m.txt"
                    echo "build.sh: kernel set $m: compiler crash with -mllvm -disable-machine-licm, retrying without it" >&2
                    { echo "set $m: first attempt (with -disable-machine-licm) died:"; grep -v "remark:\|^ *[0-9]* | \|^ *| " "$T/first_$m.txt" | head -20; } >> "$T/ru_notes.txt"
                    $HIPCC $FLAGS -Rpass-analysis=kernel-resource-usage -c -DDEKF_KSET=$m -DDEKF_KSET_ONLY -o "$T/k_$m.o" kernels.hip $EXTRA 2> "$T/ru_$m.txt" \
                        && echo "set $m: compiled WITHOUT -disable-machine-licm (retry)" >> "$T/ru_notes.txt" && return 0
                fi;;
            esac
            grep -v "remark:" "$T/ru_$m.txt" >&2
            return 1
        fi
    }
    EXTRA="$*"
    for m in 512 256 128 64 32 16 8 4 1 2 1024; do
        X=""
        case " $NO_MLICM " in *" $m "*|*" all "*) X="-mllvm -disable-machine-licm";; esac
        run unit $m $X
    done
    run $HIPCC $FLAGS -c -o "$T/capi.o" dekf_capi.hip "$@"
    rc=0
    for p in "${pids[@]}"; do wait "$p" || rc=1; done
    [ $rc -eq 0 ] || { echo "build failed"; exit 1; }
    $HIPCC --offload-arch=gfx950 -fPIC -shared -o "$OUT" "$T"/*.o -ldl
    # (warnings of the units, if any, to the terminal; the remarks to the file)
    cat "$T"/ru_*.txt | grep -v "remark:\|^ *[0-9]* | \|^ *| " >&2 || true
    { echo "# compiler resource remarks of $(basename "$OUT"), $(date -u +%Y-%m-%dT%H:%MZ), flags: $FLAGS $EXTRA; no-machine-LICM sets: $NO_MLICM";
      cat "$T"/ru_notes.txt 2>/dev/null || true; cat "$T"/ru_*.txt | grep "remark:" | sed 's/^.*remark: *//; s/ *\[-Rpass-analysis=kernel-resource-usage\]//'; } > "${OUT%.so}_resource_usage.txt"
fi
echo "built $(pwd)/$OUT"
