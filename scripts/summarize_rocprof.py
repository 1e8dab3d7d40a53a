#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace / PMC passes) into the small summaries committed under profiles/.

  python scripts/summarize_rocprof.py <rocprof_out_dir> <out_prefix> [--pmc]
Kernel trace: per-kernel launches, total/avg/min/max duration.  PMC: per-kernel sums of each counter and the derived
HBM traffic per launch using the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE counts 64-B units for 128-B
requests -> x2; FETCH_SIZE / WRITE_SIZE are in KiB).
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    """Readable kernel name; itanium-mangled template instances are decoded far enough to tell them apart."""
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(.*?)EEv", name)
    if m:
        toks, rest = [], m.group(2)
        while rest:
            for pat, fn in ((r"DF16b", lambda g: "bf16"), (r"f", lambda g: "f32"), (r"Lb([01])E", lambda g: g.group(1)),
                            (r"Li(\d+)E", lambda g: g.group(1))):
                g = re.match(pat, rest)
                if g:
                    toks.append(fn(g))
                    rest = rest[g.end():]
                    break
            else:
                toks.append(rest)
                break
        return f"{m.group(1)}<{','.join(toks)}>"
    return name[:110]


def main():
    if sys.argv[1] == "--families":
        return merge_families(*sys.argv[2:])
    d, prefix = sys.argv[1], sys.argv[2]
    pmc = "--pmc" in sys.argv
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    cfiles = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    out = {}
    if files and not pmc:
        agg = collections.defaultdict(list)
        for f in files:
            for row in csv.DictReader(open(f)):
                agg[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        rows = sorted(((k, len(v), sum(v), sum(v) / len(v), min(v), max(v)) for k, v in agg.items()), key=lambda r: -r[2])
        tot = sum(r[2] for r in rows)
        with open(prefix + "_kernel_stats.md", "w") as fh:
            fh.write("| kernel | launches | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
            for k, n, t, a, mn, mx in rows[:40]:
                fh.write(f"| `{k}` | {n} | {t/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*t/tot:.1f} |\n")
        out["kernel_stats"] = [dict(kernel=k, launches=n, total_ms=t / 1e6, avg_us=a / 1e3) for k, n, t, a, mn, mx in rows[:40]]
    if cfiles:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(lambda: collections.defaultdict(int))
        for f in cfiles:
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k][row["Counter_Name"]] += 1
        res = []
        for k in agg:
            e = dict(kernel=k)
            for c, v in agg[k].items():
                e[c + "_sum"] = v
                e[c + "_launches"] = cnt[k][c]
            res.append(e)
        out["pmc"] = res
    json.dump(out, open(prefix + ".json", "w"), indent=1)
    print("wrote", prefix + ".json")


# kernels whose template instances are different operations of the step (forward transform vs the other transform's adjoint):
# their PMC traffic is kept per instance
PER_INSTANCE = ("synth_mfma_kernel", "analysis_mfma_kernel")


def family(name):
    """Kernel family = the __global__ function without template arguments (what bench.py's roofline object names); the STFT
    kernels keep their <ADJ> instance (synth_mfma_kernel<false> = iSTFT forward, <true> = STFT backward, ...)."""
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)(I|E)(.*)", name)
    if m:
        base = m.group(1)
        if base in PER_INSTANCE:
            b = re.match(r"Lb([01])E", m.group(3))
            return f"{base}<{'true' if b and b.group(1) == '1' else 'false'}>"
        return base
    name = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", name))
    base = re.split(r"[<(]", name)[0]
    if base in PER_INSTANCE:
        b = re.search(r"<\s*(true|false|\(bool\)[01]|[01])\s*[,>]", name)   # (<ADJ> or <ADJ, terms>)
        flag = b.group(1) if b else "false"
        return f"{base}<{'true' if flag in ('true', '1', '(bool)1') else 'false'}>"
    return base


def mfma_counters(mfma_dir):
    """Per family and for the whole run: SQ_VALU_MFMA_BUSY_CYCLES (32 per v_mfma_f32_32x32x16_bf16 wave-instruction, summed
    over the chip) against GRBM_GUI_ACTIVE (summed over the 8 XCDs) -> busy / (GUI_ACTIVE / 8 * 1024 SIMDs)
    (MI355X_MICROARCH.md: cycle constants, DVFS section)."""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(mfma_dir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            per[family(row["Kernel_Name"])][row["Counter_Name"]] += float(row["Counter_Value"])
    out, tb, tg = {}, 0.0, 0.0
    for k, c in per.items():
        busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        tb += busy
        tg += gui
        if busy > 0 and gui > 0:
            out[k] = {"mfma_busy_cycles": busy, "gui_active": gui, "util": busy / (gui / 8.0 * 1024.0),
                      "sq_busy_cycles": c.get("SQ_BUSY_CYCLES")}
    return {"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)", "by_family": out,
            "whole_step": (tb / (tg / 8.0 * 1024.0)) if tg else None}


def merge_families(trace_dir, fetch_dir, write_dir, prefix, mfma_dir=None, source_id=None):
    """Per kernel family: launches, total / average duration (kernel trace) and HBM bytes (separate FETCH_SIZE and
    WRITE_SIZE passes; FETCH_SIZE counts 64-B units for 128-B requests on gfx950 -> x2; both are KiB)."""
    fams = collections.defaultdict(lambda: dict(launches=0, total_ns=0, fetch_kib=None, write_kib=None, pmc_launches=0))
    for f in glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            e = fams[family(row["Kernel_Name"])]
            e["launches"] += 1
            e["total_ns"] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    for d, key in ((fetch_dir, "fetch_kib"), (write_dir, "write_kib")):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                    continue
                e = fams[family(row["Kernel_Name"])]
                e[key] = (e[key] or 0.0) + float(row["Counter_Value"])
                if key == "fetch_kib":
                    e["pmc_launches"] += 1
    out = {}
    for k, e in fams.items():
        hb = None
        if e["fetch_kib"] is not None and e["write_kib"] is not None and e["pmc_launches"]:
            hb = (2.0 * e["fetch_kib"] + e["write_kib"]) * 1024.0 / e["pmc_launches"] * max(e["launches"], 1)
        out[k] = dict(launches=e["launches"], total_ms=e["total_ns"] / 1e6,
                      avg_us=e["total_ns"] / 1e3 / max(e["launches"], 1), hbm_bytes=hb,
                      hbm_bytes_per_launch=(hb / e["launches"]) if (hb is not None and e["launches"]) else None)
    doc = {"families": out, "source_id": source_id}
    if mfma_dir:
        doc["mfma_util"] = mfma_counters(mfma_dir)
    json.dump(doc, open(prefix + ".json", "w"), indent=1)
    rows = sorted(out.items(), key=lambda kv: -kv[1]["total_ms"])
    with open(prefix + ".md", "w") as fh:
        fh.write("| kernel family | launches | total ms | avg us | HBM MB / launch (PMC) |\n|---|---|---|---|---|\n")
        mu = doc.get("mfma_util", {}).get("by_family", {})
        for k, e in rows[:40]:
            hb = "" if e["hbm_bytes_per_launch"] is None else f"{e['hbm_bytes_per_launch'] / 1e6:.2f}"
            ut = f" (MFMA busy {100 * mu[k]['util']:.1f} %)" if k in mu else ""
            fh.write(f"| `{k}`{ut} | {e['launches']} | {e['total_ms']:.3f} | {e['avg_us']:.1f} | {hb} |\n")
        if doc.get("mfma_util", {}).get("whole_step") is not None:
            fh.write(f"\nMFMA busy over the whole profiled step: {100 * doc['mfma_util']['whole_step']:.2f} % "
                     f"({doc['mfma_util']['formula']}); source id {source_id}\n")
    print("wrote", prefix + ".json", prefix + ".md")
    return


if __name__ == "__main__":
    main()
