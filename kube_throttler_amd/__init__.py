"""kube_throttler_amd — MI355X-native throttle-evaluation engine behind kube-throttler's plugin API.

The product path is ``libkt_engine.so`` (hand-written HIP for gfx950 behind the C-ABI of
include/kt_engine.h); :mod:`kube_throttler_amd.engine` is its ctypes binding and raises if the
library is missing — there is no CPU fallback.
"""
__all__ = ["snapshot", "quantity", "objects", "workload", "engine"]
