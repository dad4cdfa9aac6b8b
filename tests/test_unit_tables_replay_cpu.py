"""Sanity of the unit-table replay itself (no GPU): the collector finds the states, and the oracle handles each of
them end to end at every probed instant — so that tests/test_parity_extended_gpu.py only adds the engine side."""
import numpy as np

from kube_throttler_amd import snapshot as S
from unit_table_states import collect, instants


def test_replay_states_on_oracle(oracle_mod):
    states = collect(oracle_mod)
    labels = [l for l, _ in states]
    assert len(states) >= 30 and len(set(labels)) == len(labels)
    assert any(l.startswith("invalid_selectors") for l in labels) and any(l.endswith("/bound") for l in labels)
    n_runs = n_used = n_active_override = 0
    for label, cs in states:
        for now in instants(cs):
            snap = cs.build().snapshot
            o = oracle_mod.Oracle(snap)
            need = S.THR_VALID | S.THR_RESPONSIBLE
            rows = np.nonzero((snap.thr_flags[:snap.n_thr] & need) == need)[0]
            assert len(rows) == snap.n_thr, label          # every throttle was made responsible
            o.pod_requests()
            want = o.reconcile(now, rows=rows)
            o.next_override(now)
            snap.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod,
                              want.error, rows=rows)
            for on_equal in (False, True):
                st, sm = o.check(on_equal=on_equal)
                assert st.shape == (snap.n_pods, snap.n_thr)
            n_runs += 1
            n_used += int((want.used.has_count[:len(rows)] != 0).any())
            n_active_override += int(want.calc_updated[:len(rows)].any())
    assert n_runs > len(states) and n_used >= 8 and n_active_override >= 10
