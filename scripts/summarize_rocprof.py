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


if __name__ == "__main__":
    main()
