#!/usr/bin/env python
"""Summarise a tools/profile.sh output directory into profiles/ (tracked).

    python tools/pmc_summary.py gpurun_out/prof_<tag> <round> <key> [gpurun_out/sq_<tag>]

Writes profiles/<round>_<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim),
profiles/<round>_<tag>_pmc.csv (per-kernel FETCH_SIZE / WRITE_SIZE averages) and merges
profiles/pmc_summary.json[<key>] = {kernel: {hbm_bytes_per_launch, fetch_bytes, write_bytes, ...}}.

With the SQ counter directory of tools/pmc_sq.sh as a fourth argument the per-kernel averages of the SQ / LDS counters
are merged in as well (`sq`), and the rocprofv3 average duration of every kernel (`rocprof_avg_ns`, `rocprof_calls`) —
bench.py derives `hbm_actual_frac`, `valu_busy` and `lds_conflict_frac` from them when the library that runs reports the
same source hash.

Units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are KiB at the L2's memory-side;
on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced stream and is
uncalibrated for other widths, so both the raw and the x2-corrected read bytes are recorded;
hbm_bytes_per_launch uses the x2 correction (upper bound of the read side) + raw writes.
"""
import csv
import glob
import json
import os
import re
import shutil
import sys


def short(name):
    m = re.search(r"(kt_[a-z_]+)", name)
    return m.group(1) if m else name[:40]


def counter_avgs(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                k = short(row["Kernel_Name"])
                out.setdefault(k, []).append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in out.items()}


def kernel_stats(d):
    """{short kernel name: (average ns, calls)} from rocprofv3 --kernel-trace --stats (several instantiations of one
    kernel are weighted by their calls)"""
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "kt_" not in row["Name"]:
                    continue
                k = short(row["Name"])
                tot, n = acc.get(k, (0.0, 0))
                acc[k] = (tot + float(row["TotalDurationNs"]), n + int(row["Calls"]))
    return {k: (tot / n, n) for k, (tot, n) in acc.items() if n}


SQ_KEEP = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVES",
           "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_INSTS_LDS", "SQ_INSTS_LDS_ATOMIC", "SQ_INSTS_SALU")


def sq_counters(d):
    acc = {}
    for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] not in SQ_KEEP or "kt_" not in row["Kernel_Name"]:
                    continue
                acc.setdefault(short(row["Kernel_Name"]), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    src, rnd, key = sys.argv[1], sys.argv[2], sys.argv[3]
    sq_dir = sys.argv[4] if len(sys.argv) > 4 else None
    tag = os.path.basename(src.rstrip("/")).replace("prof_", "")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    os.makedirs(prof, exist_ok=True)
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(prof, f"{rnd}_{tag}_kernel_stats.csv"))
    for leg in ("trace", "fetch", "write"):
        j = os.path.join(src, leg + ".json")
        if os.path.exists(j) and os.path.getsize(j):
            shutil.copy(j, os.path.join(prof, f"{rnd}_{tag}_bench_{leg}.json"))
    fetch = counter_avgs(os.path.join(src, "fetch"), "FETCH_SIZE")
    write = counter_avgs(os.path.join(src, "write"), "WRITE_SIZE")
    rows, summ = [], {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("kt_"):
            continue
        fkb, fn = fetch.get(k, (0.0, 0))
        wkb, wn = write.get(k, (0.0, 0))
        fb, wb = fkb * 1024, wkb * 1024
        rows.append([k, fn, round(fb), round(2 * fb), wn, round(wb), round(2 * fb + wb)])
        summ[k] = {"fetch_bytes_raw": round(fb), "fetch_bytes_x2": round(2 * fb), "write_bytes": round(wb),
                   "hbm_bytes_per_launch": round(2 * fb + wb), "launches": fn}
    for k, (avg, n) in kernel_stats(os.path.join(src, "trace")).items():
        summ.setdefault(k, {}).update({"rocprof_avg_ns": round(avg, 1), "rocprof_calls": n})
    if sq_dir:
        for k, cs in sq_counters(sq_dir).items():
            summ.setdefault(k, {})["sq"] = {c: round(v, 1) for c, v in sorted(cs.items())}
        st = os.path.join(sq_dir, "summary.txt")
        if os.path.exists(st):
            shutil.copy(st, os.path.join(prof, f"{rnd}_{tag}_sq_counters.txt"))
    with open(os.path.join(prof, f"{rnd}_{tag}_pmc.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "fetch_launches", "FETCH_SIZE_bytes_raw", "FETCH_SIZE_bytes_x2_gfx950", "write_launches",
                    "WRITE_SIZE_bytes", "hbm_bytes_per_launch(x2 fetch + write)"])
        w.writerows(rows)
    path = os.path.join(prof, "pmc_summary.json")
    allp = {}
    if os.path.exists(path):
        with open(path) as fh:
            allp = json.load(fh)
    allp[key] = summ
    # which kernel sources the counters belong to: the engine_version the profiled bench.py run itself reported
    version = None
    for leg in ("fetch", "write", "trace"):
        j = os.path.join(src, leg + ".json")
        try:
            with open(j) as fh:
                version = json.loads(fh.read().strip().splitlines()[-1])["config"]["engine_version"]
            break
        except Exception:
            continue
    allp.setdefault("_source", {})[key] = {"tag": f"{rnd}_{tag}", "engine_version": version}
    with open(path, "w") as fh:
        json.dump(allp, fh, indent=1, sort_keys=True)
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
