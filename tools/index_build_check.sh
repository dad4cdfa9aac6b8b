#!/bin/bash
# Accepting a change to the index BUILDER (kube_throttler_amd/csrc/kt_index.cpp) without a GPU: the fingerprint of
# everything the device gets must not move — on the random suite of index_sim_test and on the real selector programs of
# BASELINE configs[2] and the configs[4] shard — and the phase times tell what the change bought (minimum of 12 builds,
# single-threaded unless THREADS is set).       usage: tools/index_build_check.sh
# The pinned values are those of the round-6 layout (veto plane of the words that have veto bits + a zero column, shared word lists as {begin, end} pairs: kt_index.h); a change of the index LAYOUT moves them on purpose (re-pin here and
# in tests/test_host_cpu.py after the GPU parity tests have passed on the new layout).
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
WANT="f610dd7cf5b008e7 472beb285d32e619 6a27f5073695a18e"
make -C kube_throttler_amd/csrc 2>&1 | grep -E "error|warning"
make -C kube_throttler_amd/host index_sim_test 2>&1 | grep -E "error|warning"
for c in 2 4; do
  [ -f /tmp/kt_cfg$c.bin ] || python tools/dump_program.py --config $c --pods 4096 /tmp/kt_cfg$c.bin > /dev/null
done
SIM=kube_throttler_amd/host/index_sim_test
a=$($SIM | tail -1 | grep -o "indexes [0-9a-f]*" | cut -d' ' -f2)
b=$($SIM /tmp/kt_cfg2.bin 2>/dev/null | grep fingerprint | awk '{print $3}')
c=$($SIM /tmp/kt_cfg4.bin 2>/dev/null | grep fingerprint | awk '{print $3}')
echo "fingerprints (random suite, configs[2], configs[4] shard): $a $b $c"
[ "$a $b $c" == "$WANT" ] && echo "IDENTICAL to the pinned layout" || echo "DIFFERENT from the pinned layout ($WANT)"
KT_INDEX_THREADS=${THREADS:-1} KT_DEBUG_COMPILE=1 KT_SIM_BUILD_REPS=12 $SIM /tmp/kt_cfg4.bin 2>&1 | grep "build_index" | python3 -c "
import sys, re, collections
m = collections.OrderedDict()
for l in sys.stdin:
    r = re.match(r'\s*build_index: (.*?)\s+([0-9.]+) ms', l)
    if r: m.setdefault(r.group(1).strip() or 'TOTAL', []).append(float(r.group(2)))
    else:
        r = re.match(r'build_index: ([0-9.]+) ms', l)
        if r: m.setdefault('TOTAL', []).append(float(r.group(1)))
print('configs[4] shard, ms: ' + ' | '.join('%s %.2f' % (k[:22], min(v)) for k, v in m.items()))"
