#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
KT_DEBUG_COMPILE=1 timeout 300 python - <<'PY' 2>&1 | tail -40
import sys, time, numpy as np
sys.path.insert(0, '.')
from kube_throttler_amd import engine as E, workload as W
cfg = W.preset(4).shard(0, 8)
snap = W.generate(cfg)
eng = E.Engine.for_snapshot(snap)
eng.check_atomic(rows=np.array([0], dtype=np.int64), want_status=False)
print("---- one throttle event", flush=True)
one = snap.throttle_batch(np.array([5], dtype=np.int32))
eng.upsert_throttles(one, rows=np.array([5], dtype=np.int32))
t0 = time.perf_counter(); eng.check_atomic(rows=np.array([0], dtype=np.int64), want_status=False); print("first check after the event: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
eng.close()
PY
for n in 1 4 16; do echo "== KT_INDEX_THREADS=$n"; KT_INDEX_THREADS=$n KT_DEBUG_COMPILE=1 timeout 120 kube_throttler_amd/host/index_sim_test tools/cfg4_program.bin 2>&1 | grep -i "build_index:" ; done
timeout 600 python -m pytest tests -m gpu -x -q -k "not config4" 2>&1 | tail -3
