import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kube_throttler_amd import workload as W
from oracle import kt_oracle as O
import numpy as np
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as ex: pass
snap = W.generate(W.preset(4).shard(1, 8))
o = O.Oracle(snap)
sample = np.arange(0, 200000, dtype=np.int64)
o.check(rows=sample[:20000], want_status=False, nthreads=64)
for nt in (16, 32, 64, 128, 256):
    t0 = time.time(); o.check(rows=sample, want_status=False, nthreads=nt); print("check 200k pods, %3d threads: %.2fs" % (nt, time.time() - t0))
rows = np.arange(5000, 5400, dtype=np.int32)
for nt in (32, 128, 256):
    t0 = time.time(); o.reconcile((1767225600, 0), rows=rows, nthreads=nt); print("reconcile 400 cluster throttles, %3d threads: %.2fs" % (nt, time.time() - t0))
