"""N>1 path on CPU: world_size-2 gloo run of the sharding + all-reduce algebra that bench.py uses on RCCL.

The HIP kernels cannot run here, so each rank produces its partial-`used` buffer with the CPU oracle on ITS
pod shard; what is under test is everything around the kernels: the generator's shard determinism, the
partial-buffer layout (presence as counts), the sum all-reduce, and that the reduced buffer reproduces the
single-process result bit for bit.  In one sentence: this tests the ALGEBRA and the LAYOUT of the exchange (the layout comes
from the library's own kt_partial_layout), not the kernels — those are pinned per shard on one GPU by tests/test_sharded_gpu.py.
The same two ranks also run bench.py's self-verification of a multi-rank run (result_hashes / ranks_agree over
all_gather_object): agreement on the replicated tables, and a rank whose exchange went wrong is caught.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pods, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kube_throttler_amd import workload as W
    import partial_layout as KD
    from oracle import kt_oracle as O
    full_cfg = W.small(seed=77, n_pods=n_pods, n_thr=40, n_cluster=20)
    snap = W.generate(full_cfg.shard(rank, world))
    now = (full_cfg.now_s, 0)
    r = O.Oracle(snap).reconcile(now)
    buf = torch.from_numpy(KD.pack_partial(r.used.v, r.used.present, r.used.count, r.error, snap.D).copy())
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    # bench.py's self-verification of a multi-rank run, over the same transport primitives: every rank "finalizes" the reduced
    # buffer (replicated) and hashes it next to its own summary words; the ranks gather the hashes and must agree on the
    # replicated part (bench.result_hashes / ranks_agree) — and a rank that reduced something else must be caught
    import types
    import bench
    v, present, count, has_count, err = KD.unpack_partial(buf.numpy(), snap.D)
    T = snap.n_thr
    rec = types.SimpleNamespace(used=types.SimpleNamespace(v=v, count=count, present=present), thrl_flag=np.zeros(T, np.uint32),
                                thrl_has=np.zeros(T, np.uint8), thrl_pod=np.zeros(T, np.uint8), error=err)
    own = np.arange(snap.n_pods, dtype=np.uint64) + np.uint64(1000 * rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(bench.result_hashes(rec, own), rank=rank))
    agree = bench.ranks_agree(gathered)
    if rank == 1:
        rec.used.v = rec.used.v.copy()
        rec.used.v[3, 0] += 1  # this rank's exchange "went wrong"
    gathered_bad = [None] * world
    dist.all_gather_object(gathered_bad, dict(bench.result_hashes(rec, own), rank=rank))
    caught = not bench.ranks_agree(gathered_bad)
    if rank == 0:
        q.put((buf.numpy().copy(), agree, caught, [g["own"] for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pods", [2001])
def test_two_rank_allreduce_matches_single_process(n_pods, oracle_mod):
    from kube_throttler_amd import workload as W
    import partial_layout as KD
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pods, q)) for r in range(2)]
    for p in procs:
        p.start()
    reduced, agree, caught, own_hashes = q.get(timeout=900)  # the children re-import torch: minutes on a cold page cache
    assert agree and caught and own_hashes[0] != own_hashes[1]
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    full = W.generate(W.small(seed=77, n_pods=n_pods, n_thr=40, n_cluster=20))
    want = oracle_mod.Oracle(full).reconcile((full.cfg.now_s, 0))
    v, present, count, has_count, err = KD.unpack_partial(reduced, full.D)
    T = full.n_thr
    np.testing.assert_array_equal(v, want.used.v[:T])
    np.testing.assert_array_equal(present, want.used.present[:T])
    np.testing.assert_array_equal(count, want.used.count[:T])
    np.testing.assert_array_equal(has_count, want.used.has_count[:T] != 0)


def test_shards_are_rows_of_the_full_snapshot():
    from kube_throttler_amd import workload as W
    cfg = W.small(seed=5, n_pods=1001, n_thr=16, n_cluster=8)
    full = W.generate(cfg)
    lo = 0
    for r in range(3):
        sh = W.generate(cfg.shard(r, 3))
        n = sh.n_pods
        np.testing.assert_array_equal(sh.pod_ns[:n], full.pod_ns[lo:lo + n])
        np.testing.assert_array_equal(sh.pod_flags[:n], full.pod_flags[lo:lo + n])
        a, b = int(full.pod_label_off[lo]), int(full.pod_label_off[lo + n])
        np.testing.assert_array_equal(sh.pod_label_pair[:b - a], full.pod_label_pair[a:b])
        a, b = int(full.pod_ctr_off[lo]), int(full.pod_ctr_off[lo + n])
        np.testing.assert_array_equal(sh.ctr_req[:b - a], full.ctr_req[a:b])
        np.testing.assert_array_equal(sh.pod_ovh[:n], full.pod_ovh[lo:lo + n])
        # throttles are replicated bit for bit
        np.testing.assert_array_equal(sh.thr_spec.v, full.thr_spec.v)
        np.testing.assert_array_equal(sh.preq.val, full.preq.val)
        lo += n
    assert lo == 1001
