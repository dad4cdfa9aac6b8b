"""Development aid: per-phase cycle counters of kt_check_bitmap (a -DKT_PROFILE_PHASES build, tools/build_variant.sh).
   KT_ENGINE_LIB=tools/ab/libkt_engine_prof.so python tools/prof_phases.py --config 4"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=4)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
import numpy as np, torch
from kube_throttler_amd import engine as E, workload as W
cfg = W.preset(args.config)
per_gpu = cfg.n_pods_total // 8 if args.config == 4 else cfg.n_pods_total
cfg.n_pods_total = per_gpu; cfg.pod_begin = 0; cfg.n_pods = per_gpu
snap = W.generate(cfg)
eng = E.Engine.for_snapshot(snap, E.VARIANT_INDEXED, device=0)
now = (cfg.now_s, 0)
L = E.lib()
def step():
    eng.reconcile_launch(now, True, None); eng.check_launch(per_gpu, None, False, False, None)
for _ in range(3): step()
torch.cuda.synchronize(); eng.synchronize() if hasattr(eng, "synchronize") else None
out = (C.c_ulonglong * 16)()
L.kt_debug_prof_check(out, 1)
for _ in range(args.steps): step()
eng.check_fetch(8, False)
L.kt_debug_prof_check(out, 0)
v = list(out)
names = ["barrier(top)", "chunk open (stage+tables)", "barrier(open)", "tile fetch (wait)", "tile scan", "tile drain+write", "chunk opens", "tiles"]
waves = v[9]
print("waves %d (per launch %d), kernel cycles per wave %.0f" % (waves, waves // args.steps, v[8] / max(waves, 1)))
for k in range(6):
    print("  %-28s %10.0f cycles per wave  (%.1f %%)" % (names[k], v[k] / max(waves, 1), 100.0 * v[k] / max(v[8], 1)))
print("  chunk opens per wave %.1f, (tile, chunk) visits per wave %.1f" % (v[6] / max(waves, 1), v[7] / max(waves, 1)))
print("  per chunk open: top barrier %.0f, open %.0f, open barrier %.0f cycles; per (tile,chunk): fetch %.0f scan %.0f drain %.0f cycles" % (
    v[0] / max(v[6], 1), v[1] / max(v[6], 1), v[2] / max(v[6], 1), v[3] / max(v[7], 1), v[4] / max(v[7], 1), v[5] / max(v[7], 1)))
eng.close()
