#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc -save-temps *.s) for the two patterns that separate the failing build of
analysis_mfma_kernel from the passing one:

  P  v_pk_{add,mul,fma}_f32 whose source registers were delivered by an LDS read (ds_read*) that is still the most
     recent writer of the register when the packed instruction issues;
  W  an LDS read whose DESTINATION registers overlap the DATA registers of an LDS store issued earlier in the same
     straight-line stretch with no `s_waitcnt lgkmcnt(0)` in between (the store's operands are re-used while the store
     may still be queued).

  python isa_patterns.py file.s [file.s ...]      -> one line per kernel that shows either pattern
"""
import re
import sys


def regs(tok):
    tok = tok.rstrip(",")
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def kernels(text):
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
        yield m.group(1), m.group(2).split("\n")


def scan(body):
    store_data = []          # data register sets of LDS stores not yet known to have completed
    last_writer = {}         # vgpr -> "ds" if an LDS read wrote it last, else "valu"
    P = W = 0
    w_examples = []
    for line in body:
        t = line.split()
        if not t or t[0].startswith((";", ".", "//")):
            continue
        op = t[0]
        if op.endswith(":"):                       # label: new straight-line stretch for the purpose of W
            store_data = []
            continue
        if op == "s_waitcnt" and "lgkmcnt(0)" in line:
            store_data = []
        if op == "s_barrier":
            store_data = []
        if op.startswith("ds_write") or op.startswith("ds_store"):
            data = set()
            for tok in t[2:]:
                data |= regs(tok)
            store_data.append(data)
        elif op.startswith("ds_read") or op.startswith("ds_load"):
            d = regs(t[1])
            if any(d & s for s in store_data):
                W += 1
                if len(w_examples) < 2:
                    w_examples.append(line.strip())
            for r in d:
                last_writer[r] = "ds"
        elif op.startswith("v_") or op.startswith("global_load") or op.startswith("buffer_load"):
            d = regs(t[1]) if len(t) > 1 else set()
            if op.startswith("v_pk_") and op.endswith("_f32"):
                srcs = set()
                for tok in t[2:]:
                    srcs |= regs(tok)
                if any(last_writer.get(r) == "ds" for r in srcs):
                    P += 1
            for r in d:
                last_writer[r] = "valu"
    return P, W, w_examples


def main():
    for path in sys.argv[1:]:
        text = open(path).read()
        for name, body in kernels(text):
            P, W, ex = scan(body)
            if P or W:
                print(f"{path.split('/')[-1]}: {name[:90]}: packed-f32-on-fresh-LDS-data={P} LDS-read-into-queued-store-data={W}"
                      + (f"   e.g. {ex[0]}" if ex else ""))


if __name__ == "__main__":
    main()
