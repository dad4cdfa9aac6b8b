#!/usr/bin/env python
"""Timing of the selector shapes that leave the bitmap fast path (the shapes of tests/test_engine_gpu.py's
test_throttles_with_more_than_64_terms / test_terms_with_four_and_five_positive_keys_100k, parity-checked there): ms per
reconcile + lean PreFilter sweep, beside the same cluster restricted to fast-path shapes.

    python tools/offpath_timing.py          # on an MI355X; prints one JSON line per shape
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kube_throttler_amd import engine as E, workload as W  # noqa: E402

NOW = (1_700_000_000, 0)


def time_step(cfg, name, steps=200):
    snap = W.generate(cfg)
    eng = E.Engine.for_snapshot(snap, E.VARIANT_INDEXED)
    n = snap.n_pods
    for _ in range(10):
        eng.reconcile(NOW, apply=True)
        eng.check(n=n, on_equal=False, want_status=False)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.reconcile_launch(NOW, True, None)
        eng.check_launch(n, None, False, False, None)
    eng.check_fetch(n, False)
    dt = (time.perf_counter() - t0) / steps
    terms = np.diff(snap.thr_term_off[:snap.n_thr + 1])
    out = {"shape": name, "pods": int(n), "throttles": int(snap.n_thr), "terms": int(snap.thr_term_off[snap.n_thr]),
           "max_terms_per_throttle": int(terms.max()), "ms_per_step": round(dt * 1e3, 4),
           "decisions_per_s": float(n) * float(snap.n_thr) / dt, "index": eng.index_stats() if hasattr(eng, "index_stats") else None,
           "kernels": {"check": eng.kernel_name(E.KERNEL_CHECK), "aggregate": eng.kernel_name(E.KERNEL_AGGREGATE)}}
    print(json.dumps(out), flush=True)
    eng.close()


def main():
    # (a) four and five positive keys per term (confirm() of `slow` terms) at 100k pods, and the same cluster with <= 3 requirements
    time_step(W.small(seed=4545, n_pods=100000, n_thr=160, n_cluster=80, n_ns=16, K=16, V=3, L=10, terms=(1, 3), reqs=(3, 5)), "4-5 positive keys, 100k pods")
    time_step(W.small(seed=4545, n_pods=100000, n_thr=160, n_cluster=80, n_ns=16, K=16, V=3, L=10, terms=(1, 3), reqs=(1, 3)), "same cluster, <= 3 requirements per term")
    # (b) throttles with 130 terms (slow list) beside ordinary ones, and the same cluster with <= 8 terms
    time_step(W.small(seed=840, n_pods=100000, n_thr=48, n_cluster=24, K=16, V=8, L=6, terms=(1, 130), reqs=(1, 3)), "1-130 terms per throttle, 100k pods")
    time_step(W.small(seed=840, n_pods=100000, n_thr=48, n_cluster=24, K=16, V=8, L=6, terms=(1, 8), reqs=(1, 3)), "same cluster, <= 8 terms per throttle")
    # ... and about the same NUMBER of terms (3170 there) spread over throttles of at most 64: the fair twin of (b)
    time_step(W.small(seed=840, n_pods=100000, n_thr=96, n_cluster=48, K=16, V=8, L=6, terms=(1, 64), reqs=(1, 3)), "twice the throttles, <= 64 terms each (same number of terms)")


if __name__ == "__main__":
    main()
