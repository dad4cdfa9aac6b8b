#!/usr/bin/env python
"""Fills the @PLACEHOLDERS@ of DESIGN.md §7 from the bench lines under profiles/<round>_bench_*.json (tools/summarise_round.sh).
    python tools/fill_design_results.py r06
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]


def line(name):
    with open(os.path.join(ROOT, "profiles", f"{rnd}_bench_{name}.json")) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


def cells(d):
    k, r = d["roofline"]["per_kernel_ms"], d["roofline"]
    kk = "%.1f / %.1f / %.1f" % (k["check"] * 1e3, k["aggregate"] * 1e3, (k["reduce"] + k["finalize"]) * 1e3)
    if k.get("prepare"):
        kk += " (+ %.1f verdict images)" % (k["prepare"] * 1e3)
    return kk, "%.1f" % (d["ms_per_step"] * 1e3), ("%.2e" % d["value"]).replace("e+", "·10^"), \
        "%.2f / %.2f / %.2f" % (r["check"]["frac"], r["aggregate"]["frac"], r["step"]["frac"])


s = open(os.path.join(ROOT, "DESIGN.md")).read()
for tag, name in (("C2", "cfg2"), ("C3", "cfg3"), ("C24", "cfg2_4M"), ("C4", "cfg4"), ("D16", "cfg2_D16"), ("L16", "cfg2_L16")):
    kk, st, v, f = cells(line(name))
    s = s.replace(f"@{tag}K@", kk).replace(f"@{tag}S@", st).replace(f"@{tag}V@", v).replace(f"@{tag}F@", f)
d1 = line("cfg1")
s = s.replace("@C1S@", "%.1f" % (d1["ms_per_step"] * 1e3)).replace("@C1V@", ("%.2e" % d1["value"]).replace("e+", "·10^"))
drv = line("cfg2_driver")
lat = drv.get("latency") or {}
s = s.replace("@LAT1@", "%.1f" % lat.get("check1", {}).get("p50_us", float("nan")))
up = lat.get("upsert1", {}) or {}
s = s.replace("@LATU@", "%.1f" % up.get("p50_us", float("nan")))
uc = lat.get("upsert1_then_check1", {}) or {}
s = s.replace("@LATUC@", "%.1f" % uc.get("p50_us", float("nan")))
rc = (line("cfg4").get("latency") or {}).get("recompile", {}) or {}
s = s.replace("@LATRC@", "%.1f" % rc.get("mean_ms", float("nan")))
s = s.replace("@CPUB@", ("%.2e" % drv["cpu_baseline"]["value"]).replace("e+", "·10^"))
open(os.path.join(ROOT, "DESIGN.md"), "w").write(s)
print("filled")
