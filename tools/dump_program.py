#!/usr/bin/env python
"""Dumps the selector program + a pod sample of a BASELINE config in the flat form kt::build_index takes, for
tests/cpp/index_sim_test.cpp's file mode (CPU replay of the scan: step counts and lane occupancy of today's peel loop vs
the lane-parallel blueprint, on the REAL program of a config instead of random ones).

    python tools/dump_program.py --config 4 --pods 4096 /tmp/cfg4.bin
    kube_throttler_amd/host/index_sim_test /tmp/cfg4.bin
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kube_throttler_amd import snapshot as S, workload as W  # noqa: E402


def ns_selector_matches(op, key, val_off, val, r0, r1, ns_keys, ns_pairs):
    """AND of requirements against a namespace's labels (key ids / pair ids), clusterthrottle_selector.go:52-69."""
    for r in range(r0, r1):
        vals = val[val_off[r]:val_off[r + 1]]
        if op[r] == S.OP_IN:
            ok = any(v in ns_pairs for v in vals)
        elif op[r] == S.OP_NOT_IN:
            ok = not any(v in ns_pairs for v in vals)
        elif op[r] == S.OP_EXISTS:
            ok = key[r] in ns_keys
        else:
            ok = key[r] not in ns_keys
        if not ok:
            return False
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--pods", type=int, default=4096)
    ap.add_argument("out")
    a = ap.parse_args()
    cfg = W.preset(a.config)
    if a.config == 4:
        cfg.n_pods_total //= 8      # thresholds etc. as on one of the 8 shards
    cfg.pod_begin, cfg.n_pods = 0, a.pods
    snap = W.generate(cfg)
    T, G, NS = snap.n_thr, snap.n_term, snap.n_ns
    def flat(reqs):   # both pool classes keep one spare element behind op / key / val
        n = len(reqs)
        op, key, off, val = (reqs.op, reqs.key, reqs.val_off, reqs.val) if hasattr(reqs, "n") else reqs.arrays()
        off = np.asarray(off, np.uint32)
        return (np.asarray(op[:n], np.uint32).tolist(), np.asarray(key[:n], np.uint32).tolist(), off.tolist(),
                np.asarray(val[:int(off[n])], np.uint32).tolist())

    p_op, p_key, p_off, p_val = flat(snap.preq)
    n_op, n_key, n_off, n_val = flat(snap.nreq)
    need = S.THR_VALID | S.THR_RESPONSIBLE
    gw = (G + 31) // 32 + 1
    ok = np.zeros((NS, gw), np.uint32)
    ns_sets = [(set(snap.ns_label_key[snap.ns_label_off[n]:snap.ns_label_off[n + 1]].tolist()),
                set(snap.ns_label_pair[snap.ns_label_off[n]:snap.ns_label_off[n + 1]].tolist())) for n in range(NS)]
    for t in range(T):
        if (int(snap.thr_flags[t]) & need) != need:
            continue
        cluster = bool(int(snap.thr_flags[t]) & S.THR_CLUSTER)
        for g in range(int(snap.thr_term_off[t]), int(snap.thr_term_off[t + 1])):
            if not cluster:
                if snap.thr_ns[t] < NS:
                    ok[int(snap.thr_ns[t]), g >> 5] |= np.uint32(1 << (g & 31))
                continue
            if int(snap.term_flags[g]) & S.TERM_NS_SEL_INVALID:
                continue
            for n in range(NS):
                if snap.ns_valid[n] and ns_selector_matches(n_op, n_key, n_off, n_val, int(snap.term_nreq_off[g]),
                                                            int(snap.term_nreq_off[g + 1]), *ns_sets[n]):
                    ok[n, g >> 5] |= np.uint32(1 << (g & 31))
    term_thr = np.zeros(max(G, 1), np.uint32)
    for t in range(T):
        term_thr[int(snap.thr_term_off[t]):int(snap.thr_term_off[t + 1])] = t
    live = ((snap.thr_flags[:T] & need) == need).astype(np.uint32)
    cluster = ((snap.thr_flags[:T] & S.THR_CLUSTER) != 0).astype(np.uint32)
    arrays = [np.array([T, G, NS, gw, snap.n_pods, snap.D], np.uint32), snap.thr_term_off[:T + 1], term_thr[:G],
              snap.term_flags[:G].astype(np.uint32), snap.term_preq_off[:G + 1], np.asarray(p_op, np.uint32), np.asarray(p_key, np.uint32),
              np.asarray(p_off, np.uint32), np.asarray(p_val, np.uint32), live, cluster, snap.thr_ns[:T], ok.reshape(-1),
              snap.pod_ns[:snap.n_pods], snap.pod_label_off[:snap.n_pods + 1], snap.pod_label_key, snap.pod_label_pair]
    with open(a.out, "wb") as fh:
        for arr in arrays:
            arr = np.ascontiguousarray(arr, np.uint32)
            fh.write(np.uint32(arr.size).tobytes())
            fh.write(arr.tobytes())
    print(f"wrote {a.out}: T={T} terms={G} namespaces={NS} pods={snap.n_pods} requirements={len(p_op)}")


if __name__ == "__main__":
    main()
