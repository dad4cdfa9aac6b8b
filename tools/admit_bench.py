"""Times kt_admit_launch (sequential admission with reservation) on a BASELINE config.
usage: python tools/admit_bench.py [--config 2] [--queue 20000] [--pods 1000000]"""
import argparse
import json
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kube_throttler_amd import engine as E, snapshot as S, workload as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--queue", type=int, default=20000)
ap.add_argument("--pods", type=int, default=1000000)
a = ap.parse_args()
cfg = W.preset(a.config)
cfg.n_pods_total = cfg.n_pods = a.pods
snap = W.generate(cfg)
T = snap.n_thr
# head-room, as in tests/test_engine_gpu.py::test_admit_queue_*: the queue fills the throttles up on the way
snap.thr_spec.v[:T] = snap.thr_spec.v[:T] * 2 + 1
snap.thr_spec.count[:T] = snap.thr_spec.count[:T] * 2 + 3
eng = E.Engine.for_snapshot(snap)
eng.reconcile((1767225600, 0), apply=True)
fl = snap.pod_flags[:snap.n_pods]
pending = np.nonzero(((fl & S.POD_VALID) != 0) & ((fl & S.POD_SCHEDULED) == 0))[0][:a.queue].astype(np.int64)
eng.admit(pending[:256], commit=False, want_status=False)  # warm-up (allocations)
best = None
for _ in range(3):
    t0 = time.perf_counter()
    st, sm = eng.admit(pending, commit=False, want_status=False)
    dt = time.perf_counter() - t0
    best = dt if best is None or dt < best else best
verdict = S.summary_fields(sm)[0]
print(json.dumps({"config": a.config, "pods": int(snap.n_pods), "throttles": int(T), "queue": int(len(pending)),
                  "admitted": int((verdict == S.VERDICT_ALLOW).sum()), "blocked": int((verdict == S.VERDICT_BLOCK).sum()),
                  "seconds": best, "pods_per_s": len(pending) / best, "us_per_pod": 1e6 * best / len(pending),
                  "note": "one kt_admit_launch + kt_check_fetch (summaries only), dry run; wall clock incl. launch and D2H"}))
eng.close()
