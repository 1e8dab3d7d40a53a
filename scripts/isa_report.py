#!/usr/bin/env python3
"""Static report over the gfx950 assembly the build leaves behind (csrc/build/*-hip-amdgcn-amd-amdhsa-gfx950.s): per kernel the
register and LDS budget, what that allows per SIMD / CU, spills, and the instruction mix — the numbers one otherwise collects by
hand before touching a kernel (which kernels sit at one wave per SIMD, which instances spill, where wait states pile up).

  python scripts/isa_report.py [--all] [--md out.md] [name-substring ...]

Without substrings: the kernel families of the train step (the ones profiles/*_families_* lists) plus every kernel that spills.
Occupancy model (MI355X_MICROARCH / cdna_hip_programming guides): 512 unified VGPRs (arch + acc) per SIMD lane, allocated in
blocks of 8, at most 8 waves per SIMD; 160 KB LDS per CU.  Kernels with dynamic LDS (extern __shared__) show 0 static bytes:
their launchers' sizes are noted where the source states them.  `s_nop` counts are STATIC (per kernel text, loops not weighted).
"""
import argparse
import collections
import glob
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "..", "image-in-speech-watermarking_amd", "csrc", "build")
STEP_FAMILIES = ["gemm_nt_pp_kernel", "gemm_tn_pp_kernel", "gemm_nt_kernel", "gemm_tn_kernel", "attn_half_fwd_kernel", "attn_half_bwd_kernel",
                 "dwconv_fwd_strip_kernel", "dwconv_bwd_kernel", "ln_bwd_kernel", "ln_fwd_kernel", "attn_fwd_fast_kernel",
                 "attn_bwd_fast_kernel", "reduce_jobs_kernel", "adamw_kernel", "analysis_mfma_kernel", "synth_mfma_kernel",
                 "attack_fwd_kernel", "attack_bwd_kernel", "x_to_grid_kernel", "grid_to_dx_kernel", "conv_fanin_kernel", "conv_fanout_kernel"]
# dynamic LDS the launchers request (bytes), where the kernel declares extern __shared__
DYNAMIC_LDS = {"gemm_nt_pp_kernel": 131072, "gemm_tn_pp_kernel": 131072, "gemm_nt_big_kernel": 131072, "gemm_tn_big_kernel": 131072}


def demangle(names):
    """binutils' c++filt does not know the bf16 mangling (DF16b): it is swapped for `half` (Dh) on the way in — a one-for-one token, so
    the substitution indices stay valid — and renamed on the way out."""
    try:
        out = subprocess.run(["c++filt"], input="\n".join(n.replace("DF16b", "Dh") for n in names), capture_output=True, text=True,
                             timeout=60).stdout
        return {n: re.sub(r"\bhalf\b", "bf16", d) for n, d in zip(names, out.splitlines())}
    except (OSError, subprocess.SubprocessError):
        return {n: n for n in names}


def short(dem):
    dem = dem.replace("(anonymous namespace)::", "")
    dem = re.sub(r"\(.*", "", dem)                             # drop the argument list
    return dem.replace("void ", "").replace("__bf16", "bf16")


def classify(op):
    if "mfma" in op:
        return "mfma"
    if op == "s_nop":
        return "nop"
    if op == "s_waitcnt":
        return "wait"
    if op.startswith("v_pk_"):
        return "v_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_barrier"):
        return "barrier"
    return "salu"


INFO = {"NumVgprs": "arch_vgprs", "NumAgprs": "acc_vgprs", "TotalNumVgprs": "vgprs", "ScratchSize": "scratch", "Occupancy": "occupancy",
        "LDSByteSize": "lds", "TotalNumSgprs": "sgprs"}


def parse(path):
    """{kernel symbol: (info, instruction mix)}; info from the '; Kernel info:' comment block hipcc writes behind every kernel
    (register totals of the unified file, scratch bytes = spills, its own waves-per-SIMD figure) and the metadata's workgroup size."""
    s = open(path).read()
    wg = {m.group(2): int(m.group(1)) for m in re.finditer(r"\.max_flat_workgroup_size:\s+(\d+)\n\s+\.name:\s+(\S+)", s)}
    spill = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n){0,12}?\s+\.vgpr_spill_count:\s+(\d+)", s)}
    kernels = {}
    labels = list(re.finditer(r"^(_Z\w+):[ \t]*(;.*)?\n", s, re.M))
    for i, m in enumerate(labels):
        name = m.group(1)
        if name not in wg:
            continue
        end = s.find(".Lfunc_end", m.end())
        nxt = labels[i + 1].start() if i + 1 < len(labels) else len(s)
        blk = s[end:nxt]
        k = blk.find("; Kernel info:")
        if k < 0:
            continue
        info = {"wg": wg[name], "spilled": spill.get(name, 0)}
        for mm in re.finditer(r"^; (\w+): (\d+)", blk[k:k + 1200], re.M):
            if mm.group(1) in INFO:
                info[INFO[mm.group(1)]] = int(mm.group(2))
        mix = collections.Counter()
        for line in s[m.end():end].splitlines():
            t = line.strip()
            if not line.startswith("\t") or not t or t.startswith((".", ";")):
                continue
            mix[classify(t.split()[0])] += 1
        kernels[name] = (info, mix)
    return kernels


def occupancy(info, name):
    """(waves per SIMD, LDS bytes incl. the launcher's dynamic request, workgroups per CU, what bounds them)."""
    alloc = max(8, (info.get("vgprs", 0) + 7) // 8 * 8)
    waves_simd = min(8, 512 // alloc)
    wg_waves = max(1, info["wg"] // 64)
    lds = info.get("lds", 0)
    for k, v in DYNAMIC_LDS.items():
        if k in name:
            lds = max(lds, v)
    wgs_regs = waves_simd * 4 // wg_waves
    wgs_lds = (160 * 1024) // lds if lds else 99
    return waves_simd, lds, max(0, min(wgs_regs, wgs_lds, 32)), ("LDS" if wgs_lds < wgs_regs else "registers")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true", help="every kernel of the library")
    ap.add_argument("--md", help="write the table as markdown to this file")
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    files = sorted(glob.glob(os.path.join(BUILD, "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if not files:
        sys.exit("no assembly under csrc/build: run `make -C image-in-speech-watermarking_amd/csrc` first (it keeps the .s files)")
    kernels = {}
    for f in files:
        for k, v in parse(f).items():
            kernels[k] = v + (os.path.basename(f).split("-hip-")[0],)
    dem = demangle(list(kernels))
    want = a.names or STEP_FAMILIES
    rows = []
    for name, (info, mix, src) in kernels.items():
        d = short(dem[name])
        spills = info.get("scratch", 0) > 0 or info.get("spilled", 0) > 0
        if not (a.all or any(w in d for w in want) or (spills and not a.names)):
            continue
        wps, lds, wgs, bound = occupancy(info, d)
        rows.append((src, d, info.get("vgprs", 0), info.get("acc_vgprs", 0), wps, lds, wgs, bound, info.get("spilled", 0), info.get("scratch", 0),
                     sum(mix.values()), mix["mfma"], mix["valu"] + mix["v_pk"], mix["v_pk"], mix["lds"], mix["vmem"], mix["nop"], mix["wait"]))
    rows.sort(key=lambda r: (r[0], r[1]))
    head = ("| source | kernel instance | VGPRs (of which acc) | waves / SIMD | LDS B | workgroups / CU (bound by) | spilled VGPRs / scratch B | "
            "instructions | MFMA | VALU (packed) | LDS | VMEM | s_nop | s_waitcnt |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    lines = [head]
    for r in rows:
        lines.append(f"| {r[0]} | `{r[1]}` | {r[2]} ({r[3]}) | {r[4]} | {r[5]} | {r[6]} ({r[7]}) | {r[8]} / {r[9]} | {r[10]} | {r[11]} | "
                     f"{r[12]} ({r[13]}) | {r[14]} | {r[15]} | {r[16]} | {r[17]} |")
    text = "\n".join(lines)
    if a.md:
        with open(a.md, "w") as fh:
            fh.write(f"Static ISA report of libwm_hip.so's kernels ({len(rows)} instances; `python scripts/isa_report.py`; occupancy model in the "
                     "script's header; counts are static, loops not weighted)\n\n" + text + "\n")
        print(f"wrote {a.md}: {len(rows)} kernel instances")
    else:
        print(text)


if __name__ == "__main__":
    main()
