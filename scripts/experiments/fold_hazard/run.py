#!/usr/bin/env python3
"""One pass of the fold-hazard experiment on a GPU box (libraries built beforehand by build.sh):

for each (build, source order) the forward n_fft = 255 transform is launched `reps` times at a bench-like batch on
identical input and every launch is compared with the product kernel's result (libwm_hip.so) element by element.
Prints one JSON object; a variant that ever differs is reported with the affected frames (frame t of a 32-frame group is
thread t / t + 32 / ... of the fold loop, i.e. lanes t and t + 32 of a wave) and planes (re = sums u, im = differences v).

  python scripts/experiments/fold_hazard/run.py [--batch 96] [--reps 60]
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=96)
    ap.add_argument("--reps", type=int, default=60)
    a = ap.parse_args()
    wm = importlib.import_module("image-in-speech-watermarking_amd")
    ops = wm.ops
    B = a.batch
    g = torch.Generator().manual_seed(3)
    w = (0.2 * torch.randn(B, 8002, generator=g)).cuda()
    ref = torch.empty(B, 2, 128, 128, device="cuda")
    ops.stft_fwd(w, ref, B)
    tab = ops.stft_tables(w.device)
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    # ISA patterns per variant as printed by build.sh (P = packed fp32 folds on fresh LDS data, W = LDS reads returning
    # into the data registers of queued LDS stores)
    patterns = {("slp", 0): "P+W", ("slp", 1): "P", ("slp", 2): "P+W", ("noslp", 0): "-", ("noslp", 1): "-", ("noslp", 2): "W",
                ("slp", 3): "P, all LDS reads retired + 32 idle slots before the first packed add",
                ("slp", 4): "hand-written v_pk_add_f32 op_sel swap", ("slp", 5): "hand-written v_pk_add_f32, no op_sel",
                ("noslp", 3): "- (idle slots only)", ("noslp", 4): "hand-written v_pk_add_f32 op_sel swap, rest scalar",
                ("noslp", 5): "hand-written v_pk_add_f32, no op_sel, rest scalar"}
    out = {"batch": B, "reps": a.reps, "variants": []}
    for build in ("slp", "noslp"):
        lib = C.CDLL(os.path.join(HERE, f"libfold_{build}.so"))
        lib.fold_run.restype = C.c_int
        lib.fold_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        for var in (0, 1, 2, 3, 4, 5):
            bad, frames, planes, nbad_el = 0, set(), set(), 0
            maxdev = 0.0
            for rep in range(a.reps):
                sp = torch.empty(B, 2, 128, 128, device="cuda")
                rc = lib.fold_run(var, w.data_ptr(), sp.data_ptr(), tab.data_ptr(), B, st)
                assert rc == 0, rc
                ne = sp != ref
                if bool(ne.any()):
                    bad += 1
                    idx = ne.nonzero()
                    nbad_el += idx.shape[0]
                    frames |= set((idx[:, 3] % 32).unique().tolist())
                    planes |= set(idx[:, 1].unique().tolist())
                    maxdev = max(maxdev, float((sp - ref).abs().max()))
            out["variants"].append({"build": build, "source_order": var, "isa": patterns[build, var], "launches": a.reps,
                                    "launches_differing": bad, "elements_differing": nbad_el,
                                    "frames_mod_32": sorted(frames), "planes": sorted(planes), "max_abs_dev": maxdev})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
