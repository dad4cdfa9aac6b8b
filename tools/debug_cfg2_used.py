"""debug: configs[2] reconcile vs the oracle with a fresh engine per round (the flow of test_config2_full_size), the
device allocator's free blocks filled with 0xFF garbage before every round (uninitialised reads show), and the smaller
random programs of the test suite the same way."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from kube_throttler_amd import engine as E, workload as W, snapshot as S
import kt_oracle as O
hip = ctypes.CDLL("libamdhip64.so")
rng = np.random.default_rng(5)

def garbage():
    """allocate, fill with 0xFF, free: blocks of many sizes"""
    ptrs = []
    for sz in list(rng.integers(1 << 10, 1 << 22, size=40)) + [1 << 26, 1 << 27, 3 << 24, 40 << 20]:
        p = ctypes.c_void_p()
        if hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(int(sz))) == 0:
            hip.hipMemset(p, 0xFF, ctypes.c_size_t(int(sz)))
            ptrs.append(p)
    hip.hipDeviceSynchronize()
    for p in ptrs:
        hip.hipFree(p)

def one(snap, now, want, rows, tag):
    garbage()
    eng = E.Engine.for_snapshot(snap)
    got = eng.reconcile(now, apply=True)
    g, w = got.used.v[rows], want.used.v[:len(rows)]
    bad = np.argwhere(g != w)
    if len(bad):
        print(tag, "MISMATCHES", len(bad), eng.kernel_name(E.KERNEL_AGGREGATE), eng.kernel_name(E.KERNEL_FINALIZE), flush=True)
        for (i, d) in bad[:16]:
            print("   thr", int(rows[i]), "cluster" if snap.thr_flags[rows[i]] & S.THR_CLUSTER else "namespaced", "dim", int(d), "got", int(g[i, d]), "want", int(w[i, d]),
                  "diff", int(g[i, d] - w[i, d]), "pods got/want", int(got.used.count[rows[i]]), int(want.used.count[i]))
        print("   dims:", np.bincount(bad[:, 1], minlength=snap.D), "diff values:", np.unique((g - w)[g != w])[:12])
        got2 = eng.reconcile(now, apply=True)
        print("   same engine again: mismatches", int((got2.used.v[rows] != w).sum()), flush=True)
    eng.close()
    return len(bad)

need = S.THR_VALID | S.THR_RESPONSIBLE
total = 0
cfg = W.preset(2)
snap = W.generate(cfg)
now = (cfg.now_s, 0)
rows = np.nonzero((snap.thr_flags[:snap.n_thr] & need) == need)[0]
want = O.Oracle(snap).reconcile(now, rows=rows, nthreads=os.cpu_count() or 8)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for rep in range(rounds):
    total += one(snap, now, want, rows, "cfg2 rep %d" % rep) != 0
print("cfg2:", rounds, "rounds,", total, "with mismatches", flush=True)
bad_small = 0
for seed in range(40):
    c = W.small(seed=200 + seed, n_pods=3000 + 37 * seed, n_thr=60 + seed, n_cluster=20 + seed // 2)
    sn = W.generate(c)
    nw = (c.now_s, 0)
    rw = np.nonzero((sn.thr_flags[:sn.n_thr] & need) == need)[0]
    wt = O.Oracle(sn).reconcile(nw, rows=rw, nthreads=8)
    ok = wt.error[:len(rw)] == 0
    garbage()
    eng = E.Engine.for_snapshot(sn)
    got = eng.reconcile(nw, apply=True)
    if (got.used.v[rw][ok] != wt.used.v[:len(rw)][ok]).any():
        bad_small += 1
        print("small seed", seed, "MISMATCH", int((got.used.v[rw][ok] != wt.used.v[:len(rw)][ok]).sum()), eng.kernel_name(E.KERNEL_AGGREGATE), eng.kernel_name(E.KERNEL_FINALIZE))
    eng.close()
print("small programs: 40 seeds,", bad_small, "with mismatches")
