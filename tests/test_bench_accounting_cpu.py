"""bench.py's roofline numerator is the ALGORITHMIC byte count of SURVEY.md 8d — pinned here against the survey's own
figures (B_pod = 104 at D = 8 / L = 8; 112.3 MB for the 10^9 decisions of the headline config), without a GPU."""
import importlib.util
import os

from kube_throttler_amd import workload as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_match_the_survey():
    bench = _bench()
    cfg = W.preset(2)
    P_full = cfg.n_pods_total
    assert P_full == 10 ** 6
    cfg.n_pods_total = cfg.n_pods = 20000      # the throttle side does not depend on the pod count drawn here
    snap = W.generate(cfg)
    assert (snap.D, snap.L, snap.n_thr) == (8, 8, 1000)
    total, b_pod, thr_bytes = bench.algorithmic_bytes(snap, P_full, snap.L)
    assert b_pod == 8 + 4 * 8 + 8 * 8 == 104                       # B_pod
    n_req = len(snap.preq) + len(snap.nreq)
    assert thr_bytes == 1000 * (16 + (8 * 8 + 4 + 8) * 3 + 4) + 16 * n_req   # B_thr = 16 + (8D+4+8)*3 + 4 + 16*#req
    assert total == P_full * (104 + 8) + thr_bytes
    assert 112.0e6 < total < 112.6e6                               # "bytes_check ~ 112.3 MB for 10^9 decisions"
    assert abs(total / (P_full * snap.n_thr) - 0.112) < 0.001      # 0.112 B/decision
    # 100 % of the 8 TB/s peak <=> 14 us
    assert abs(total / (bench.HBM_PEAK_GBS * 1e9) - 14.0e-6) < 0.1e-6
    agg = bench.aggregate_bytes(snap, 600000, snap.L)
    assert agg == 600000 * 104 + 1000 * 16 + 16 * n_req + 1000 * 17 * 8


def test_workload_labels():
    bench = _bench()
    assert set(bench.WORKLOADS) == {1, 2, 3, 4}
    assert "1M pods x 1k" in bench.WORKLOADS[2]
