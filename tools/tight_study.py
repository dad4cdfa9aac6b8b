"""Development aid: how many matched (pod, throttle) pairs of a configuration go through the comparison (kRecTight), and how
many of them only because a threshold / headroom is already exhausted (head < 1: every pod that requests the dimension is
over it — decidable from the pod's non-zero mask alone).   python tools/tight_study.py --config 4"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=4); args = ap.parse_args()
import numpy as np
from kube_throttler_amd import engine as E, workload as W, snapshot as S
cfg = W.preset(args.config)
per_gpu = cfg.n_pods_total // 8 if args.config == 4 else cfg.n_pods_total
cfg.n_pods_total = per_gpu; cfg.pod_begin = 0; cfg.n_pods = per_gpu
snap = W.generate(cfg)
eng = E.Engine.for_snapshot(snap, E.VARIANT_INDEXED, device=0)
now = (cfg.now_s, 0)
rec = eng.reconcile(now, apply=True)
v, pres = eng.fetch_pod_requests(n=per_gpu)
vmax = v.max(axis=0)
T, D = snap.n_thr, snap.D
thr = rec.calc.v[:T].astype(object); tp = rec.calc.present[:T]
used = rec.used.v[:T].astype(object); resv = snap.thr_reserved.v[:T].astype(object)
w = rec.used.count[:T].astype(np.float64)  # counted pods matched: the weight of a throttle's matches
n_b = n_c = n_any = 0.0; cnt = np.zeros(4)
for t in range(T):
    if w[t] == 0: continue
    has_b = has_c = False
    for d in range(D):
        if not (int(tp[t]) >> d) & 1: continue
        th = int(thr[t][d]); hd = th - int(used[t][d]) - int(resv[t][d])
        for x in (th, hd):
            if x < 1: has_b = True
            elif x < int(vmax[d]): has_c = True
    n_any += w[t]
    if has_c: n_c += w[t]
    elif has_b: n_b += w[t]
print("matched pairs (weights): all %.3g, tight by comparison (1 <= thr/head < vmax) %.1f %%, tight only by exhausted head/thr (< 1) %.1f %%" % (
    n_any, 100 * n_c / n_any, 100 * n_b / n_any))
eng.close()
