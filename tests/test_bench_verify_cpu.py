"""bench.py's self-verification of a multi-rank run, driven on CPU with two fake ranks: the finalize is replicated,
so every rank's post-finalize per-throttle tables must hash equal (`ranks_agree`), and `--verify` holds a throttle sample of them
against the oracle on the UNSHARDED snapshot.  Tests the hashing / comparison code the first multi-GPU run will rely on — the
"engine results" here are the oracle's own (no GPU), perturbed where the check has to fire."""
import copy
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (bench.py imports torch inside main() only)
from kube_throttler_amd import snapshot as S, workload as W  # noqa: E402
from oracle import kt_oracle as O  # noqa: E402

NOW = (1767225600, 0)


def _replicated_result(snap):
    """What every rank holds after the replicated finalize, laid out by throttle row like Engine.reconcile_fetch()."""
    T, D = snap.n_thr, snap.D
    need = S.THR_VALID | S.THR_RESPONSIBLE
    rows = np.nonzero((snap.thr_flags[:T] & need) == need)[0]
    want = O.Oracle(snap).reconcile(NOW, rows=rows, nthreads=2)
    rec = types.SimpleNamespace(
        used=types.SimpleNamespace(v=np.zeros((T, D), np.int64), count=np.zeros(T, np.int64), present=np.zeros(T, np.uint32)),
        thrl_flag=np.zeros(T, np.uint32), thrl_has=np.zeros(T, np.uint8), thrl_pod=np.zeros(T, np.uint8), error=np.zeros(T, np.uint8))
    n = len(rows)
    rec.used.v[rows], rec.used.count[rows], rec.used.present[rows] = want.used.v[:n], want.used.count[:n], want.used.present[:n]
    rec.thrl_flag[rows], rec.thrl_has[rows], rec.thrl_pod[rows], rec.error[rows] = want.thrl_flag[:n], want.thrl_has[:n], want.thrl_pod[:n], want.error[:n]
    return rec, rows


def test_two_ranks_agree_and_a_diverging_rank_is_caught():
    snap = W.generate(W.small(seed=7, n_pods=2000, n_thr=64, n_cluster=32))
    rec, _ = _replicated_result(snap)
    # the ranks' own halves of the summary words differ, the replicated tables do not
    sm0, sm1 = np.arange(1000, dtype=np.uint64), np.arange(1000, 2000, dtype=np.uint64)
    h0, h1 = dict(bench.result_hashes(rec, sm0), rank=0), dict(bench.result_hashes(rec, sm1), rank=1)
    assert h0["replicated"] == h1["replicated"] and h0["own"] != h1["own"] and h0["both"] != h1["both"]
    assert bench.ranks_agree([h0, h1])
    # a world-1 run prints the same fields
    assert bench.ranks_agree([h0]) and set(h0) == {"replicated", "own", "both", "rank"}
    # one rank whose exchange went wrong: a single word of `used`, a single flag
    for field in ("used.v", "thrl_flag", "error"):
        bad = copy.deepcopy(rec)
        if field == "used.v":
            bad.used.v[5, 0] += 1
        else:
            getattr(bad, field)[5] ^= 1
        hb = dict(bench.result_hashes(bad, sm1), rank=1)
        assert not bench.ranks_agree([h0, hb]), field


def test_throttle_sample_against_the_oracle_on_the_unsharded_snapshot():
    cfg = W.small(seed=11, n_pods=3000, n_thr=48, n_cluster=24)
    full = W.generate(cfg)
    rec, rows = _replicated_result(full)
    sample = rows[:: max(1, len(rows) // 16)]
    assert bench.verify_against_oracle(full, NOW, rec, sample, nthreads=2) == []
    # the sum of two ranks' shards IS the unsharded result (integer sums are associative): a rank that dropped its shard's
    # contribution to one throttle shows up by name
    t = int(sample[len(sample) // 2])
    bad = copy.deepcopy(rec)
    bad.used.count[t] += 1
    got = bench.verify_against_oracle(full, NOW, bad, sample, nthreads=2)
    assert got == [("used.count", t)], got
