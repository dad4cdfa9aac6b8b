#!/usr/bin/env python
"""Host-side study of the ANCHORED index layout (VERDICT r3 "do this" #3) on the real selector program of a BASELINE
config — the acceptance statistic before any kernel is touched: word visits per pod, copy blow-up, words that have to be
resident per namespace, and the match sets against brute force.

Layout studied (throttle-level anchoring; everything else as in kt_index.h):
  * a GROUP is a throttle's namespace cell: the terms whose namespace side admits the cell's namespaces (kt_index.cpp);
  * every term with a positive pair requirement has an ANCHOR requirement (its `In` with the fewest values); a group's
    anchors are the union of its terms' anchor atoms, in a fixed order a_0 < a_1 < ...; a group with a term that has no `In`
    stays unanchored (visited by every pod of its namespaces, as today);
  * the group is copied once per anchor: copy i = the group's terms that can still match a pod that carries a_i and none of
    a_0..a_{i-1} (a term whose anchor requirement is on a_i's key but does not name a_i, or whose anchor atoms are all
    among the earlier ones, cannot), with a veto on a_0..a_{i-1} — so for any pod exactly ONE copy of a throttle can fire
    (the first anchor it carries) and the "reported once" rule of the scans holds inside that copy;
  * copies are numbered by (anchor, admission class); a pod visits, per atom it carries, the words of that anchor that hold
    a copy admitted for its namespace — nothing else.

    python tools/study/anchor_study.py --config 4 --pods 4096
"""
import argparse
import collections
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kube_throttler_amd import snapshot as S, workload as W  # noqa: E402
from dump_program import ns_selector_matches  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--pods", type=int, default=4096)
    ap.add_argument("--brute", type=int, default=256, help="pods checked against brute force")
    a = ap.parse_args()
    cfg = W.preset(a.config)
    if a.config == 4:
        cfg.n_pods_total //= 8
    cfg.pod_begin, cfg.n_pods = 0, a.pods
    snap = W.generate(cfg)
    T, NS = snap.n_thr, snap.n_ns
    p, n_ = snap.preq, snap.nreq
    need = S.THR_VALID | S.THR_RESPONSIBLE
    ns_sets = [(set(snap.ns_label_key[snap.ns_label_off[n]:snap.ns_label_off[n + 1]].tolist()),
                set(snap.ns_label_pair[snap.ns_label_off[n]:snap.ns_label_off[n + 1]].tolist())) for n in range(NS)]
    n_op, n_key, n_off, n_val = n_.op.tolist(), n_.key.tolist(), n_.val_off.tolist(), n_.val.tolist()

    # ---- terms: requirements as (op, key, frozenset(values)); admission set as a python int over namespaces
    def term_reqs(g):
        out = []
        for r in range(int(snap.term_preq_off[g]), int(snap.term_preq_off[g + 1])):
            out.append((int(p.op[r]), int(p.key[r]), frozenset(p.val[int(p.val_off[r]):int(p.val_off[r + 1])].tolist())))
        return out

    groups = []  # (throttle, admission int, [term reqs])
    n_terms = 0
    for t in range(T):
        if (int(snap.thr_flags[t]) & need) != need:
            continue
        cluster = bool(int(snap.thr_flags[t]) & S.THR_CLUSTER)
        terms = []
        for g in range(int(snap.thr_term_off[t]), int(snap.thr_term_off[t + 1])):
            if cluster:
                if int(snap.term_flags[g]) & S.TERM_NS_SEL_INVALID:
                    continue
                adm = 0
                for n in range(NS):
                    if snap.ns_valid[n] and ns_selector_matches(n_op, n_key, n_off, n_val, int(snap.term_nreq_off[g]),
                                                                int(snap.term_nreq_off[g + 1]), *ns_sets[n]):
                        adm |= 1 << n
            else:
                adm = 1 << int(snap.thr_ns[t]) if snap.thr_ns[t] < NS else 0
            if adm:
                terms.append((adm, term_reqs(g)))
        n_terms += len(terms)
        # cells: namespaces partitioned by WHICH terms admit them
        cells = collections.defaultdict(int)
        union = 0
        for adm, _ in terms:
            union |= adm
        for n in range(NS):
            if (union >> n) & 1:
                pat = tuple(i for i, (adm, _) in enumerate(terms) if (adm >> n) & 1)
                cells[pat] |= 1 << n
        for pat, adm in cells.items():
            groups.append((t, adm, [terms[i][1] for i in pat]))

    # ---- anchors and copies
    def anchor_req(reqs):
        best = None
        for op, key, vals in reqs:
            if op == S.OP_IN and vals and (best is None or len(vals) < len(best[2])):
                best = (op, key, vals)
        return best

    pair_key = {}
    for r in range(len(p)):
        for v in p.val[int(p.val_off[r]):int(p.val_off[r + 1])].tolist():
            pair_key[v] = int(p.key[r])
    copies = collections.defaultdict(list)  # anchor atom (0 = unanchored) -> [(admission, n_terms, group index, copy terms, vetoes)]
    n_copy_terms = 0
    for gi, (t, adm, terms) in enumerate(groups):
        anch = [anchor_req(r) for r in terms]
        if any(x is None for x in anch):
            copies[0].append((adm, len(terms), gi, list(range(len(terms))), ()))
            n_copy_terms += len(terms)
            continue
        atoms = sorted(set(v for x in anch for v in x[2]))
        for i, ai in enumerate(atoms):
            earlier = set(atoms[:i])
            alive = []
            for k, x in enumerate(anch):
                if x[1] == pair_key[ai] and ai not in x[2]:
                    continue  # the pod carries a_i on that key: it cannot carry one of this term's anchor values
                if set(x[2]) <= earlier:
                    continue  # every anchor value of the term is vetoed in this copy
                alive.append(k)
            if alive:
                copies[ai].append((adm, len(alive), gi, alive, tuple(atoms[:i])))
                n_copy_terms += len(alive)

    # ---- numbering per anchor: copies sorted by admission class, a copy never straddles a word, words anchor-pure
    words_of = {}  # anchor -> list of words; word = list of (admission, n_terms)
    total_words = 0
    for ai, lst in copies.items():
        lst.sort(key=lambda c: c[0])
        words, cur, fill = [], [], 0
        for c in lst:
            if fill + c[1] > 64:
                words.append(cur)
                cur, fill = [], 0
            cur.append(c)
            fill += c[1]
        if cur:
            words.append(cur)
        words_of[ai] = words
        total_words += len(words)
    # per (anchor, namespace): the words a pod of that namespace carrying that anchor visits
    visit = {ai: [sum(1 for w in words if any((c[0] >> n) & 1 for c in w)) for n in range(NS)] for ai, words in words_of.items()}
    resident = [sum(visit[ai][n] for ai in visit) for n in range(NS)]  # words with any copy admitted for the namespace

    # ---- pods
    def pod_labels(i):
        l0, l1 = int(snap.pod_label_off[i]), int(snap.pod_label_off[i + 1])
        return snap.pod_label_key[l0:l1].tolist(), snap.pod_label_pair[l0:l1].tolist()

    def req_ok(req, keys, pairs):
        op, key, vals = req
        if op == S.OP_IN:
            return any(v in vals for v in pairs)
        if op == S.OP_NOT_IN:
            return not any(v in vals for v in pairs)
        if op == S.OP_EXISTS:
            return key in keys
        return key not in keys

    v_sum = v_max = m_sum = 0
    bad = 0
    for i in range(snap.n_pods):
        keys, pairs = pod_labels(i)
        n = int(snap.pod_ns[i])
        v = visit.get(0, [0] * NS)[n] + sum(visit[x][n] for x in pairs if x in visit)
        v_sum += v
        v_max = max(v_max, v)
        if i < a.brute:
            # the anchored walk: copies of the pod's atoms (+ unanchored) admitted for its namespace, with the vetoes
            got = collections.Counter()
            pset = set(pairs)
            for ai in [0] + [x for x in pairs if x in copies]:
                for adm, _, gi, alive, vetoes in copies[ai]:
                    if not (adm >> n) & 1 or any(x in pset for x in vetoes):
                        continue
                    terms = groups[gi][2]
                    if any(all(req_ok(r, keys, pairs) for r in terms[k]) for k in alive):
                        got[groups[gi][0]] += 1
            want = set()
            for t, adm, terms in groups:
                if (adm >> n) & 1 and any(all(req_ok(r, keys, pairs) for r in reqs) for reqs in terms):
                    want.add(t)
            m_sum += len(want)
            if set(got) != want or any(c != 1 for c in got.values()):
                bad += 1
    print(f"config {a.config}: {T} throttles, {n_terms} admitted terms, {len(groups)} groups (cells), {NS} namespaces")
    print(f"  anchored copies: {sum(len(v) for v in copies.values())} group copies, {n_copy_terms} term numbers "
          f"({n_copy_terms / max(n_terms, 1):.2f}x the terms; {len(copies.get(0, []))} groups stay unanchored), {total_words} words "
          f"over {len(words_of)} anchors")
    print(f"  word visits per pod: {v_sum / snap.n_pods:.2f} (max {v_max}) over {snap.n_pods} pods")
    print(f"  words with a copy admitted for a namespace (what a namespace-ordered workgroup keeps resident): "
          f"mean {np.mean(resident):.1f}, max {max(resident)}")
    print(f"  brute force on {min(a.brute, snap.n_pods)} pods: {m_sum / max(1, min(a.brute, snap.n_pods)):.2f} throttles matched per pod, "
          f"{bad} pods with a wrong or doubly reported throttle set")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
