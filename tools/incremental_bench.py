"""Times the incremental event path: a batch of pod updates followed by a reconcile, with an
incremental engine (delta scans + copy) and with a plain one (full rescan).  usage: python tools/incremental_bench.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from kube_throttler_amd import engine as E, workload as W  # noqa: E402
from test_engine_gpu import _permute_pods  # noqa: E402

cfg = W.preset(2)
snap = W.generate(cfg)
NOW = (cfg.now_s, 0)
out = {"config": 2, "pods": int(snap.n_pods), "throttles": int(snap.n_thr)}
rng = np.random.default_rng(5)
for name, variant in (("incremental", E.VARIANT_INDEXED | E.VARIANT_INCREMENTAL), ("full_rescan", E.VARIANT_INDEXED)):
    eng = E.Engine.for_snapshot(snap, variant)
    eng.reconcile(NOW, apply=True)
    res = {}
    for batch in (1, 100, 10000):
        rows = rng.choice(snap.n_pods, batch, replace=False).astype(np.int64)
        src = rng.choice(snap.n_pods, batch, replace=False)
        b = _permute_pods(snap, src)
        eng.upsert_pods(b, rows=rows)
        eng.reconcile(NOW, apply=True)  # warm
        t_up = t_rec = 0.0
        reps = 5
        for _ in range(reps):
            t0 = time.perf_counter()
            eng.upsert_pods(b, rows=rows)
            t1 = time.perf_counter()
            eng.reconcile_launch(NOW, True)
            eng.synchronize()
            t2 = time.perf_counter()
            t_up += t1 - t0
            t_rec += t2 - t1
        res[str(batch)] = {"upsert_ms": round(1e3 * t_up / reps, 3), "reconcile_ms": round(1e3 * t_rec / reps, 3)}
    out[name] = res
    eng.close()
print(json.dumps(out))
