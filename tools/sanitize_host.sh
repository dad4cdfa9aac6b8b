#!/bin/bash
# AddressSanitizer + UBSan over the host-side code that runs without a GPU: the index builder (kt_index.cpp, compiled into the
# replay test itself so that the instrumented copy is the one called) on the random suite and on dumped BASELINE programs, and the
# plugin mirror's unit tests.  GPU sanitizers are not available on the pool; the device side has KT_DEBUG_POISON (tests) instead.
#   bash tools/sanitize_host.sh [dumped program ...]      (tools/dump_program.py --config 4 /tmp/kt_cfg4.bin)
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/kt_asan; mkdir -p $OUT
SAN="-O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer"
CSRC=$REPO/kube_throttler_amd/csrc
g++ $SAN -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$CSRC -I$REPO/include -o $OUT/index_sim_test $REPO/tests/cpp/index_sim_test.cpp \
    $REPO/tools/study/kt_anchor.cpp $CSRC/kt_index.cpp -L$CSRC -lkt_engine -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$CSRC -lpthread 2> $OUT/build.log
g++ $SAN -I$REPO/kube_throttler_amd/host -I$REPO/include -o $OUT/host_unit_test $REPO/tests/cpp/host_unit_test.cpp $REPO/kube_throttler_amd/host/kt_host.cpp \
    -L$CSRC -lkt_engine -Wl,-rpath,$CSRC 2>> $OUT/build.log
export ASAN_OPTIONS=detect_leaks=0   # (the HIP runtime's start-up allocations are not ours)
fail=0
run() { "$@" > $OUT/run.log 2>&1 || fail=1; n=$(grep -c "runtime error\|AddressSanitizer" $OUT/run.log || true); echo "$(basename $1) ${*:2}: $(tail -1 $OUT/run.log | cut -c1-120) | sanitizer reports: $n"; [ "$n" == "0" ] || fail=1; }
run $OUT/index_sim_test
run $OUT/host_unit_test
for f in "$@"; do
  KT_SIM_CHK_WORD=816 KT_SIM_PACKED=40 run $OUT/index_sim_test $f
  KT_CUT_PLAN=grouped KT_SIM_CHK_WORD=816 KT_SIM_PACKED=40 run $OUT/index_sim_test $f
done
exit $fail
