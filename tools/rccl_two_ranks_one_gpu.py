#!/usr/bin/env python
"""Can two engines on ONE GPU (two threads, two RCCL ranks) run the native exchange?  (RCCL normally refuses a duplicate
GPU; if it does, the world=2 transport can only be exercised on a box with two GPUs.)  Prints one line."""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kube_throttler_amd import engine as E, workload as W  # noqa: E402

NOW = (1767225600, 0)


def main():
    cfgs = [W.small(seed=5, n_pods=2048, n_thr=24, n_cluster=12).shard(r, 2) for r in range(2)]
    snaps = [W.generate(c) for c in cfgs]
    engs = [E.Engine.for_snapshot(s, device=0) for s in snaps]
    uid = E.Engine.comm_unique_id()
    res = [None, None]

    def run(r):
        try:
            engs[r].comm_init(r, 2, uid)
            engs[r].aggregate_launch()
            engs[r].comm_allreduce_partial()
            engs[r].finalize_launch(NOW, True)
            res[r] = engs[r].reconcile_fetch()
        except Exception as ex:  # noqa: BLE001
            res[r] = ex
    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    if any(isinstance(r, Exception) for r in res):
        print("two RCCL ranks on one GPU: refused:", [repr(r)[:200] for r in res if isinstance(r, Exception)][0])
        return
    T = snaps[0].n_thr
    same = np.array_equal(res[0].used.v[:T], res[1].used.v[:T])
    print("two RCCL ranks on one GPU: ok, ranks agree on used:", same)


if __name__ == "__main__":
    main()
