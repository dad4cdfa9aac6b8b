"""ThrottleSpecBase.NextOverrideHappensIn (throttle_types.go:37-63) on the oracle: the earliest begin / end instant
strictly after now; an unparsable `begin` skips the override, an unparsable `end` only the end.  The reference has no
test for it (parity unpinned): the table below is hand-derived from the code."""
from kube_throttler_amd.objects import ClusterState
from kube_throttler_amd.quantity import parse_rfc3339

NOW = "2026-01-01T00:00:00Z"


def _thr(name, overrides):
    return {"kind": "Throttle", "metadata": {"namespace": "default", "name": name},
            "spec": {"throttlerName": "kube-throttler", "threshold": {"resourceCounts": {"pod": 1}},
                     "selector": {"selectorTerms": [{"podSelector": {"matchLabels": {"a": "b"}}}]},
                     "temporaryThresholdOverrides": [dict(o, threshold={"resourceCounts": {"pod": 2}}) for o in overrides]}}


CASES = [
    ("none", [], None),
    ("past", [{"begin": "2025-01-01T00:00:00Z", "end": "2025-06-01T00:00:00Z"}], None),
    ("active-until", [{"begin": "2025-01-01T00:00:00Z", "end": "2026-03-01T00:00:00Z"}], "2026-03-01T00:00:00Z"),
    ("future", [{"begin": "2026-02-01T00:00:00Z", "end": "2026-03-01T00:00:00Z"}], "2026-02-01T00:00:00Z"),
    ("now-is-not-after-now", [{"begin": NOW, "end": "2026-01-01T00:00:01Z"}], "2026-01-01T00:00:01Z"),
    ("open-ended", [{"begin": "2026-05-01T00:00:00Z"}], "2026-05-01T00:00:00Z"),
    ("min-over-overrides", [{"begin": "2026-05-01T00:00:00Z", "end": "2026-06-01T00:00:00Z"},
                            {"begin": "2025-01-01T00:00:00Z", "end": "2026-04-01T00:00:00Z"}], "2026-04-01T00:00:00Z"),
    ("bad-begin-skips-the-override", [{"begin": "soon", "end": "2026-02-01T00:00:00Z"}], None),
    ("bad-end-keeps-the-begin", [{"begin": "2026-02-01T00:00:00Z", "end": "later"}], "2026-02-01T00:00:00Z"),
    ("bad-end-past-begin", [{"begin": "2025-02-01T00:00:00Z", "end": "later"},
                            {"begin": "2026-07-01T00:00:00Z", "end": "2026-08-01T00:00:00Z"}], "2026-07-01T00:00:00Z"),
    ("fractional", [{"begin": "2026-01-01T00:00:00.5Z", "end": "2026-01-02T00:00:00Z"}], "2026-01-01T00:00:00.5Z"),
]


def test_next_override_table(oracle_mod):
    cs = ClusterState()
    cs.add_namespace("default")
    for name, ovr, _ in CASES:
        cs.add(_thr(name, ovr))
    built = cs.build()
    o = oracle_mod.Oracle(built.snapshot)
    sec, nsec, has = o.next_override(parse_rfc3339(NOW))
    for i, (name, _, want) in enumerate(CASES):
        if want is None:
            assert not has[i], name
        else:
            assert has[i] and (int(sec[i]), int(nsec[i])) == parse_rfc3339(want), name
