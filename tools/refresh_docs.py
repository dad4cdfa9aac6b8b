#!/usr/bin/env python
"""Rewrites the result figures the documents quote from the committed evidence (profiles/<round>_bench_*.json, pmc_summary.json,
the rocprofv3 kernel statistics) — run after tools/summarise_round.sh, so that the prose never trails the files:
    python tools/refresh_docs.py r06 r06q
Only the table rows and sentences matched below are touched; everything else in the documents is left alone.
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tag = sys.argv[1], sys.argv[2]


def line(name):
    with open(os.path.join(ROOT, "profiles", f"{rnd}_bench_{name}.json")) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


def parts(d):
    r = d["roofline"]
    k = r["per_kernel_ms"]
    return dict(ms=d["ms_per_step"], v=d["value"], c=k["check"], a=k["aggregate"], f=k["reduce"] + k["finalize"], p=k.get("prepare", 0.0),
                fc=r["check"]["frac"], fa=r["aggregate"]["frac"], fs=r["step"]["frac"])


def sci(v, sup=False):
    m, e = ("%.2e" % v).split("e")
    e = int(e)
    if sup:
        return m + "·10" + "".join("⁰¹²³⁴⁵⁶⁷⁸⁹"[int(ch)] for ch in str(e))
    return f"{m}·10^{e}"


def rocprof(cfg):
    out = {}
    for f in glob.glob(os.path.join(ROOT, "profiles", f"{rnd}_{tag}_{cfg}_kernel_stats.csv")):
        for row in csv.DictReader(open(f)):
            name = row["Name"]
            for key in ("kt_check_bitmap", "kt_aggregate_bitmap", "kt_reduce_finalize_packed", "kt_reduce_packed_slabs", "kt_finalize", "kt_build_verdict_images"):
                if key + "<" in name or name.startswith("kt::" + key + "(") or ("kt::" + key + "<") in name:
                    out.setdefault(key, float(row["AverageNs"]) / 1e3)
    return out


c2, c2d, c24, c3, c4, c1, d16, l16 = (parts(line(n)) for n in ("cfg2", "cfg2_driver", "cfg2_4M", "cfg3", "cfg4", "cfg1", "cfg2_D16", "cfg2_L16"))
pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
src = pmc["_source"]["config2_indexed"]["engine_version"].split("src=")[-1]
mb = lambda cfg, kn: pmc[cfg][kn]["hbm_bytes_per_launch"] / 1e6
alg4 = line("cfg4")["roofline"]["step"]["algorithmic_bytes"]
rp2, rp4 = rocprof("cfg2"), rocprof("cfg4")


def sub(path, rules):
    p = os.path.join(ROOT, path)
    s = open(p).read()
    for pat, rep in rules:
        s, n = re.subn(pat, rep, s, count=1, flags=re.S)
        if n != 1:
            print(f"{path}: no match for {pat[:60]!r}")
    open(p, "w").write(s)


us = lambda x: "%.1f" % (x * 1e3)
pc = lambda x: "%.0f %%" % (100 * x)
f2 = lambda x: "%.2f" % x
# ---- one row per config
sub("DESIGN.md", [
    (r"\| 2 \(1M × 1k\) \|[^\n]*\n", f"| 2 (1M × 1k) | {us(c2['c'])} / {us(c2['a'])} / {us(c2['f'])} | {us(c2['ms'])} | {sci(c2['v'])} | {f2(c2['fc'])} / {f2(c2['fa'])} / {f2(c2['fs'])} |\n"),
    (r"\| 3 \(overrides\) \|[^\n]*\n", f"| 3 (overrides) | {us(c3['c'])} / {us(c3['a'])} / {us(c3['f'])} | {us(c3['ms'])} | {sci(c3['v'])} | {f2(c3['fc'])} / {f2(c3['fa'])} / {f2(c3['fs'])} |\n"),
    (r"\| 2 with 4M pods \|[^\n]*\n", f"| 2 with 4M pods | {us(c24['c'])} / {us(c24['a'])} / {us(c24['f'])} | {us(c24['ms'])} | {sci(c24['v'])} | {f2(c24['fc'])} / {f2(c24['fa'])} / {f2(c24['fs'])} |\n"),
    (r"\| 1 \(10k × 100\) \| launch floors \| [0-9.]+ \| [^|]* \|", f"| 1 (10k × 100) | launch floors | {us(c1['ms'])} | {sci(c1['v'])} |"),
    (r"\| 4, one shard \(1\.25M × 10k\) \| [^|]*\| \*\*[0-9.]+\*\* \(r05: 882, r04: 1 257, r03: 1 612\) \| [^|]* \|",
     f"| 4, one shard (1.25M × 10k) | {us(c4['c'])} / {us(c4['a'])} / {us(c4['f'])} (+ {us(c4['p'])} verdict images) | **{us(c4['ms'])}** (r05: 882, r04: 1 257, r03: 1 612) | {sci(c4['v'])} |"),
    (r"\| 2 at D = 16 \(not BASELINE\) \| [^|]*\| \*\*[0-9.]+\*\* ([^|]*)\| [^|]* \| [^|]* \|",
     f"| 2 at D = 16 (not BASELINE) | {us(d16['c'])} / {us(d16['a'])} / {us(d16['f'])} | **{us(d16['ms'])}** \\1| {sci(d16['v'])} | {f2(d16['fc'])} / {f2(d16['fa'])} / {f2(d16['fs'])} |"),
    (r"\| 2 at L = 16 \(not BASELINE\) \| [^|]*\| [0-9.]+ \(r05: 96\.4\) \| [^|]* \| [^|]* \|",
     f"| 2 at L = 16 (not BASELINE) | {us(l16['c'])} / {us(l16['a'])} / {us(l16['f'])} | {us(l16['ms'])} (r05: 96.4) | {sci(l16['v'])} | {f2(l16['fc'])} / {f2(l16['fa'])} / {f2(l16['fs'])} |"),
    (r"\| r06 \| [0-9.]+ \| \*\*[0-9.]+\*\* \|", f"| r06 | {us(c2['ms'])} | **{us(c4['ms'])}** |"),
    (r"\(`profiles/r06_r06[a-z]_cfg4_pmc\.csv`\)", f"(`profiles/{rnd}_{tag}_cfg4_pmc.csv`)"),
])
# ---- BASELINE.md §4
ms3 = lambda x: "%.4f" % x
sub("BASELINE.md", [
    (r"`profiles/r06_r06[a-z]_\*_kernel_stats\.csv`", f"`profiles/{rnd}_{tag}_*_kernel_stats.csv`"),
    (r"\| 2 \(1M × 1k\) \| 1 \| \*\*[^*]*\*\* \([^)]*\) \| [^|]* \| \*\*[^*]*\*\* \| [0-9.]+ / [0-9.]+ MB",
     f"| 2 (1M × 1k) | 1 | **{sci(c2['v'], True)}** ({ms3(c2['ms'])} ms / 10⁹ decisions; the driver's command: {ms3(c2d['ms'])}) | {ms3(c2['c'])} / {ms3(c2['a'])} / {ms3(c2['f'])} | **{pc(c2['fc'])} / {pc(c2['fa'])} / {pc(c2['fs'])}** | {mb('config2_indexed', 'kt_check_bitmap'):.1f} / {mb('config2_indexed', 'kt_aggregate_bitmap'):.1f} MB"),
    (r"\| 2 with 4 M pods \| 1 \| [^|]* \| [^|]* \| [^|]* \|",
     f"| 2 with 4 M pods | 1 | {sci(c24['v'], True)} ({c24['ms']:.3f} ms) | {ms3(c24['c'])} / {ms3(c24['a'])} / {ms3(c24['f'])} | {pc(c24['fc'])} / {pc(c24['fa'])} / {pc(c24['fs'])} |"),
    (r"\| 3 \(overrides\) \| 1 \| [^|]* \| [^|]* \| [^|]* \|",
     f"| 3 (overrides) | 1 | {sci(c3['v'], True)} ({ms3(c3['ms'])} ms) | {ms3(c3['c'])} / {ms3(c3['a'])} / {ms3(c3['f'])} | {pc(c3['fc'])} / {pc(c3['fa'])} / {pc(c3['fs'])} |"),
    (r"\*\*[0-9.·¹²³⁴⁵⁶⁷⁸⁹⁰]+\*\* \([0-9.]+ ms; round 5: 0\.882\) \| [0-9.]+ / [0-9.]+ / [0-9.]+ \| [0-9.]+ % / [0-9.]+ % / [0-9.]+ %",
     f"**{sci(c4['v'], True)}** ({c4['ms']:.3f} ms; round 5: 0.882) | {c4['c']:.3f} / {c4['a']:.3f} / {c4['f']:.3f} | {100 * c4['fc']:.1f} % / {100 * c4['fa']:.1f} % / {100 * c4['fs']:.1f} %"),
    (r"[0-9.]+ / [0-9.]+ MB = [0-9.]+× / [0-9.]+× algorithmic \(round 5",
     f"{mb('config4_indexed', 'kt_check_bitmap'):.1f} / {mb('config4_indexed', 'kt_aggregate_bitmap'):.1f} MB = {mb('config4_indexed', 'kt_check_bitmap') * 1e6 / line('cfg4')['roofline']['check']['algorithmic_bytes_per_launch']:.1f}× / {mb('config4_indexed', 'kt_aggregate_bitmap') * 1e6 / line('cfg4')['roofline']['aggregate']['algorithmic_bytes_per_launch']:.1f}× algorithmic (round 5"),
    (r"\| 1 \(10k × 100\) \| 1 \| [^|]* \|", f"| 1 (10k × 100) | 1 | {sci(c1['v'], True)} |"),
    (r"\| 2 at D = 16 \(not BASELINE\) \| 1 \| [^|]* \| [^|]* \| [^|]* \|",
     f"| 2 at D = 16 (not BASELINE) | 1 | {sci(d16['v'], True)} ({ms3(d16['ms'])} ms; round 5: 0.140) | {ms3(d16['c'])} / {ms3(d16['a'])} / {ms3(d16['f'])} | {pc(d16['fc'])} / {pc(d16['fa'])} / {pc(d16['fs'])} |"),
])
# ---- README.md: the result paragraph
sub("README.md", [
    (r"\*\*[0-9.·¹²³⁴⁵⁶⁷⁸⁹⁰]+ pod×throttle decisions/s\*\*", f"**{sci(c2['v'], True)} pod×throttle decisions/s**"),
    (r"PreFilter pass \([0-9.]+ µs per 10⁹ decisions;", f"PreFilter pass ({us(c2['ms'])} µs per 10⁹ decisions;"),
    (r"`kt_check_bitmap` [0-9.]+ µs \([0-9.]+ by rocprofv3\) = [0-9]+ % of the 8 TB/s HBM peak in algorithmic bytes \([0-9]+ % at 4M pods\)",
     f"`kt_check_bitmap` {us(c2['c'])} µs ({rp2.get('kt_check_bitmap', 0):.1f} by rocprofv3) = {pc(c2['fc'])} of the 8 TB/s HBM peak in algorithmic bytes ({pc(c24['fc'])} at 4M pods)"),
    (r"with the packed fold [0-9.]+ µs = [0-9]+ % \([0-9]+ %\), `kt_reduce_finalize_packed` [0-9.]+ µs, the whole step [0-9]+ % \([0-9]+ %\)",
     f"with the packed fold {us(c2['a'])} µs = {pc(c2['fa'])} ({pc(c24['fa'])}), `kt_reduce_finalize_packed` {us(c2['f'])} µs, the whole step {pc(c2['fs'])} ({pc(c24['fs'])})"),
    (r"\*\*[0-9.]+ ms per 1\.25M-pod shard\*\*", f"**{c4['ms']:.3f} ms per 1.25M-pod shard**"),
    (r"\(step 0\.140 → [0-9.]+ ms, its check at [0-9]+ % of the HBM peak\)", f"(step 0.140 → {d16['ms']:.3f} ms, its check at {pc(d16['fc'])} of the HBM peak)"),
])
# ---- profiles/README.md: the round's header and file rows
k2 = "configs[2] check %.1f µs, aggregate %.1f, reduction + finalize %.1f" % (rp2.get("kt_check_bitmap", 0), rp2.get("kt_aggregate_bitmap", 0), rp2.get("kt_reduce_finalize_packed", 0))
k4 = "configs[4] shard %.0f / %.0f / %.1f + %.1f (+ %.1f verdict images) µs" % (rp4.get("kt_check_bitmap", 0), rp4.get("kt_aggregate_bitmap", 0), rp4.get("kt_reduce_packed_slabs", 0),
                                                                              rp4.get("kt_finalize", 0), rp4.get("kt_build_verdict_images", 0))
sub("profiles/README.md", [
    (r"Kernel sources `src=[0-9a-f]+` \(the final hash; its device code", f"Kernel sources `src={src}` (the final hash; its device code"),
    (r"`gpurun -- 'bash tools/round_evidence\.sh r06[a-z]'`,\nthen `bash tools/summarise_round\.sh r06[a-z] r06`", f"`gpurun -- 'bash tools/round_evidence.sh {tag}'`,\nthen `bash tools/summarise_round.sh {tag} {rnd}`"),
    (r"add up to [0-9.]+ µs against\n[0-9.]+ µs per un-instrumented step",
     "add up to %.1f µs against\n%s µs per un-instrumented step" % (rp2.get("kt_check_bitmap", 0) + rp2.get("kt_aggregate_bitmap", 0) + rp2.get("kt_reduce_finalize_packed", 0), us(c2["ms"]))),
    (r"`r06_r06[a-z]_cfg\{1,2,3,4,2_4M\}_kernel_stats\.csv` \| `rocprofv3 --kernel-trace --stats`, warm: configs\[2\] check [^;]*; configs\[4\] shard [^µ]*µs\.",
     f"`{rnd}_{tag}_cfg{{1,2,3,4,2_4M}}_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats`, warm: {k2}; {k4}."),
    (r"`r06_r06[a-z]_cfg\*_pmc\.csv`, `pmc_summary\.json` \| FETCH_SIZE / WRITE_SIZE per kernel \(separate passes\): configs\[2\] [0-9.]+ / [0-9.]+ MB per launch of check / aggregate, configs\[4\] shard [0-9.]+ / [0-9.]+ MB",
     f"`{rnd}_{tag}_cfg*_pmc.csv`, `pmc_summary.json` | FETCH_SIZE / WRITE_SIZE per kernel (separate passes): configs[2] {mb('config2_indexed', 'kt_check_bitmap'):.1f} / {mb('config2_indexed', 'kt_aggregate_bitmap'):.1f} MB per launch of check / aggregate, configs[4] shard {mb('config4_indexed', 'kt_check_bitmap'):.1f} / {mb('config4_indexed', 'kt_aggregate_bitmap'):.1f} MB"),
    (r"`r06_r06[a-z]_cfg\{2,4\}_sq_counters\.txt`", f"`{rnd}_{tag}_cfg{{2,4}}_sq_counters.txt`"),
])
q = os.path.join(ROOT, "profiles", f"{rnd}_quoted_bench_cfg2_driver.json")
if os.path.exists(q):
    d = json.loads(open(q).read().strip().splitlines()[-1])
    if src in d["config"]["engine_version"]:
        sub("profiles/README.md", [
            (r"behind the PMC passes: [0-9.]+ µs per step", "behind the PMC passes: %s µs per step" % us(d["ms_per_step"])),
            (r"`launch_gaps_ms` −?[0-9.]+", "`launch_gaps_ms` %s" % str(d["roofline"].get("launch_gaps_ms")).replace("-", "−")),
        ])
    else:
        print("profiles/%s_quoted_bench_cfg2_driver.json was taken on other sources: run `python bench.py --steps 20 --warmup 5` again" % rnd)
print("documents follow profiles/%s_bench_*.json (src=%s)" % (rnd, src))
