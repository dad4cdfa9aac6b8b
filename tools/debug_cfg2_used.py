"""debug: configs[2] reconcile vs the oracle with a fresh engine per round (the flow of test_config2_full_size); which
(throttle, dimension) entries differ and by how much."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from kube_throttler_amd import engine as E, workload as W, snapshot as S
import kt_oracle as O
cfg = W.preset(2)
snap = W.generate(cfg)
now = (cfg.now_s, 0)
o = O.Oracle(snap)
need = S.THR_VALID | S.THR_RESPONSIBLE
rows = np.nonzero((snap.thr_flags[:snap.n_thr] & need) == need)[0]
want = o.reconcile(now, rows=rows, nthreads=os.cpu_count() or 8)
want2 = o.reconcile(now, rows=rows, nthreads=1)
print("oracle threads vs 1 thread equal:", bool((want.used.v == want2.used.v).all()), "cpus", os.cpu_count())
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
junk = []
for rep in range(rounds):
    # dirty the allocator's free lists: stale bytes in whatever the next engine gets
    import ctypes
    eng = E.Engine.for_snapshot(snap)
    got = eng.reconcile(now, apply=True)
    g, w = got.used.v[rows], want.used.v[:len(rows)]
    bad = np.argwhere(g != w)
    print("rep", rep, "mismatches", len(bad), "kernel", eng.kernel_name(E.KERNEL_AGGREGATE), eng.kernel_name(E.KERNEL_FINALIZE), flush=True)
    if len(bad):
        for (i, d) in bad[:16]:
            print("   thr", int(rows[i]), "dim", int(d), "got", int(g[i, d]), "want", int(w[i, d]), "diff", int(g[i, d] - w[i, d]), "pods got/want", int(got.used.count[rows[i]]), int(want.used.count[i]))
        print("   dims:", np.bincount(bad[:, 1], minlength=snap.D), "diff values:", np.unique((g - w)[g != w])[:12])
        got2 = eng.reconcile(now, apply=True)
        print("   same engine again: mismatches", int((got2.used.v[rows] != w).sum()))
    eng.close()
