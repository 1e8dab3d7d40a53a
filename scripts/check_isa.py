#!/usr/bin/env python3
"""Build-time guard against an instruction form that computes wrong values on MI355X (gfx950, ROCm 7.2).

Found in round 3 (scripts/experiments/fold_hazard, DESIGN.md "Findings"): a packed fp32 VALU instruction whose operand selector
CROSSES the halves of one source pair — the low result takes the source's high dword AND the high result its low dword, e.g.

    v_pk_add_f32 v[12:13], v[4:5], v[6:7] op_sel:[0,1] op_sel_hi:[1,0]

returns wrong sums in lanes 16-31 and 48-63 of the wave (40 of 40 launches with the instruction written by hand into an
otherwise scalar kernel; the same kernel with the pair swapped by two v_mov and a plain v_pk_add_f32 is right 40 of 40).
hipcc's SLP vectoriser emits the form whenever it pairs x[i], x[i + 1] with y[j + 1], y[j].  Broadcast selectors (both results
from the same half: op_sel_hi:[1,0] on constants, op_sel:[1,0] ...) are everywhere in the library and parity-tested.

  python scripts/check_isa.py <libwm_hip.so | file.s ...>      exit code 1 and a listing if the form occurs
"""
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PK = re.compile(r"\bv_pk_(add|mul|fma|min|max)_f32\b")
SEL = re.compile(r"\bop_sel:\[([01,]+)\]")
SEL_HI = re.compile(r"\bop_sel_hi:\[([01,]+)\]")


def crossed(line):
    if not PK.search(line):
        return False
    nsrc = 3 if "v_pk_fma_f32" in line else 2
    m, mh = SEL.search(line), SEL_HI.search(line)
    sel = [int(x) for x in m.group(1).split(",")] if m else [0] * nsrc
    sel_hi = [int(x) for x in mh.group(1).split(",")] if mh else [1] * nsrc
    return any(s == 1 and h == 0 for s, h in zip(sel, sel_hi))


def lines_of(path):
    if path.endswith(".s"):
        return open(path).read().splitlines()
    out = subprocess.run([OBJDUMP, "-d", "--offloading", path], capture_output=True, text=True)
    if out.returncode != 0 or "v_" not in out.stdout:
        # fat binary: extract the gfx950 code object first
        import tempfile, os
        tmp = tempfile.mkdtemp()
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", f"--input={path}"], capture_output=True)
        ext = subprocess.run([OBJDUMP, "--offloading", path], capture_output=True, text=True, cwd=tmp)
        text = []
        for f in os.listdir(tmp):
            d = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], capture_output=True, text=True)
            text += d.stdout.splitlines()
        return text
    return out.stdout.splitlines()


def main():
    bad = 0
    for path in sys.argv[1:]:
        sym = "?"
        for line in lines_of(path):
            m = re.match(r"^[0-9a-f]* ?<([^>]+)>:|^(_Z\w+):", line)
            if m:
                sym = m.group(1) or m.group(2)
            if crossed(line):
                bad += 1
                if bad <= 20:
                    print(f"{path}: {sym[:80]}: {line.strip()[:140]}")
    if bad:
        print(f"check_isa: {bad} packed-fp32 instruction(s) with a half-crossing operand selector (wrong results on gfx950, see the docstring)")
        return 1
    print("check_isa: no half-crossing packed-fp32 operand selectors")
    return 0


if __name__ == "__main__":
    sys.exit(main())
