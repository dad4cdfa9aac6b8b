"""The single-object C-ABI and the native RCCL exchange, driven by a plain C99 program (tests/c/abi_flat_test.c): every
namespace, throttle and pod of a random rich workload goes through kt_upsert_namespace / kt_upsert_throttle /
kt_upsert_pod one call at a time (flat pointer arguments only — what a cgo shim on the reference's go 1.20 can pass
without pinning), the reconcile runs as kt_aggregate_launch -> kt_comm_allreduce_partial (RCCL, world = 1) ->
kt_finalize_launch, PreFilter as kt_check; everything the program writes back must equal the CPU oracle bit for bit."""
import os
import struct
import subprocess

import numpy as np
import pytest

from kube_throttler_amd import snapshot as S
from kube_throttler_amd import workload as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "kube_throttler_amd", "host")


def _emit(fh, a, dtype):
    b = np.ascontiguousarray(a, dtype=dtype).tobytes()
    fh.write(struct.pack("<I", len(b)))
    fh.write(b)


def write_scenario(path, snap, now_s):
    n, T, NS, D = snap.n_pods, snap.n_thr, snap.n_ns, snap.D
    with open(path, "wb") as fh:
        _emit(fh, [D, snap.L, NS, T, n, now_s & 0xFFFFFFFF, now_s >> 32], np.int32)
        _emit(fh, snap.ns_valid[:NS], np.uint8)
        _emit(fh, snap.ns_label_off[:NS + 1], np.uint32)
        nl = int(snap.ns_label_off[NS])
        _emit(fh, snap.ns_label_key[:nl], np.uint32)
        _emit(fh, snap.ns_label_pair[:nl], np.uint32)
        _emit(fh, snap.pod_ns[:n], np.uint32)
        _emit(fh, snap.pod_flags[:n], np.uint32)
        _emit(fh, snap.pod_label_off[:n + 1], np.uint32)
        nl = int(snap.pod_label_off[n])
        _emit(fh, snap.pod_label_key[:nl], np.uint32)
        _emit(fh, snap.pod_label_pair[:nl], np.uint32)
        _emit(fh, snap.pod_ctr_off[:n + 1], np.uint32)
        nc = int(snap.pod_ctr_off[n])
        _emit(fh, snap.ctr_init[:nc], np.uint8)
        _emit(fh, snap.ctr_present[:nc], np.uint32)
        _emit(fh, snap.ctr_req[:nc], np.int64)
        _emit(fh, snap.pod_ovh_present[:n], np.uint32)
        _emit(fh, snap.pod_ovh[:n], np.int64)
        _emit(fh, snap.thr_flags[:T], np.uint32)
        _emit(fh, snap.thr_ns[:T], np.uint32)
        for tab in (snap.thr_spec, snap.thr_calc, snap.thr_used, snap.thr_reserved):
            _emit(fh, tab.v[:T], np.int64)
            _emit(fh, tab.present[:T], np.uint32)
            _emit(fh, tab.count[:T], np.int64)
            _emit(fh, tab.has_count[:T], np.uint8)
        _emit(fh, snap.thr_thrl_flag[:T], np.uint32)
        _emit(fh, snap.thr_thrl_has[:T], np.uint32)
        _emit(fh, snap.thr_status_msgs_fp[:T], np.uint64)
        _emit(fh, snap.thr_spec_msgs_fp[:T], np.uint64)
        _emit(fh, snap.thr_ovr_off[:T + 1], np.uint32)
        no = int(snap.thr_ovr_off[T])
        _emit(fh, snap.ovr_begin_s[:no], np.int64)
        _emit(fh, snap.ovr_begin_ns[:no], np.int32)
        _emit(fh, snap.ovr_end_s[:no], np.int64)
        _emit(fh, snap.ovr_end_ns[:no], np.int32)
        _emit(fh, snap.ovr_flags[:no], np.uint8)
        _emit(fh, snap.ovr_thr.v[:no], np.int64)
        _emit(fh, snap.ovr_thr.present[:no], np.uint32)
        _emit(fh, snap.ovr_thr.count[:no], np.int64)
        _emit(fh, snap.ovr_thr.has_count[:no], np.uint8)
        _emit(fh, snap.thr_term_off[:T + 1], np.uint32)
        nt = int(snap.thr_term_off[T])
        _emit(fh, snap.term_flags[:nt], np.uint8)
        _emit(fh, snap.term_preq_off[:nt + 1], np.uint32)
        _emit(fh, snap.term_nreq_off[:nt + 1], np.uint32)
        for pool in (snap.preq, snap.nreq):
            _emit(fh, pool.op, np.uint8)
            _emit(fh, pool.key, np.uint32)
            _emit(fh, pool.val_off, np.uint32)
            _emit(fh, pool.val, np.uint32)


def read_results(path, n_pods, D):
    blob = open(path, "rb").read()
    pos = 0

    def take(dtype):
        nonlocal pos
        (n,) = struct.unpack_from("<I", blob, pos)
        pos += 4
        a = np.frombuffer(blob, dtype=dtype, count=n // np.dtype(dtype).itemsize, offset=pos)
        pos += n
        return a

    T = int(take(np.int32)[0])
    out = {}
    for tab in ("used", "calc"):
        out[tab + ".v"] = take(np.int64).reshape(T, D)
        out[tab + ".present"] = take(np.uint32)
        out[tab + ".count"] = take(np.int64)
        out[tab + ".has_count"] = take(np.uint8)
    for name, dt in (("calc_updated", np.uint8), ("thrl_flag", np.uint32), ("thrl_has", np.uint32), ("thrl_pod", np.uint8), ("error", np.uint8)):
        out[name] = take(dt)
    out["summary"] = take(np.uint64)
    out["status"] = take(np.uint8).reshape(n_pods, T)
    return T, out


@pytest.mark.parametrize("seed,kw,comm", [(31, {}, True), (32, dict(n_invalid_pod_sel=2, n_invalid_ns_sel=1, n_missing_ns=1), False)],
                         ids=["rccl-exchange", "selector-errors"])
def test_plain_c_host_through_flat_abi(seed, kw, comm, oracle_mod, tmp_path):
    subprocess.check_call(["make", "-C", HOST, "abi_flat_test"], stdout=subprocess.DEVNULL)
    cfg = W.small(seed=seed, n_pods=1500, n_thr=72, n_cluster=36, **kw)
    snap = W.generate(cfg)
    now = (cfg.now_s, 0)
    scen, res = tmp_path / "scenario.bin", tmp_path / "results.bin"
    write_scenario(scen, snap, cfg.now_s)
    run = subprocess.run([os.path.join(HOST, "abi_flat_test"), str(scen), str(res)] + (["comm"] if comm else []), capture_output=True,
                         text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    T, got = read_results(res, snap.n_pods, snap.D)
    assert T == snap.n_thr
    o = oracle_mod.Oracle(snap)
    need = S.THR_VALID | S.THR_RESPONSIBLE
    rows = np.nonzero((snap.thr_flags[:T] & need) == need)[0]
    want = o.reconcile(now, rows=rows, nthreads=8)
    np.testing.assert_array_equal(got["error"][rows] != 0, want.error[:len(rows)] != 0)
    ok = want.error[:len(rows)] == 0
    for f in ("v", "present", "count", "has_count"):
        np.testing.assert_array_equal(got["used." + f][rows][ok], getattr(want.used, f)[:len(rows)][ok], err_msg="used." + f)
        np.testing.assert_array_equal(got["calc." + f][rows][ok], getattr(want.calc, f)[:len(rows)][ok], err_msg="calc." + f)
    for f in ("calc_updated", "thrl_flag", "thrl_has", "thrl_pod"):
        np.testing.assert_array_equal(got[f][rows][ok], getattr(want, f)[:len(rows)][ok], err_msg=f)
    # the program applied the status (KT_RECONCILE_APPLY): PreFilter of every pod against it
    snap.apply_status(want.used, want.calc, want.calc_updated, want.thrl_flag, want.thrl_has, want.thrl_pod, want.error, rows=rows)
    st_w, sm_w = o.check(nthreads=8)
    np.testing.assert_array_equal(got["status"], st_w)
    np.testing.assert_array_equal(got["summary"], sm_w)
