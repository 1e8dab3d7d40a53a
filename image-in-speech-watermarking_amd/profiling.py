"""Live per-kernel timing with HIP events + the algorithmic work model used for roofline fractions.

``KernelTimer`` wraps the tensor-level launch wrappers of ``ops`` so that every launch is bracketed by a
start/stop event pair recorded on the stream the kernel runs on (torch's current stream).  It is used by
``bench.py`` on an un-captured (eager) step: a hipGraph replay cannot host events between its nodes.
Algorithmic bytes / flops per launch follow SURVEY.md 8(d) and DESIGN.md ("Kernels and their rooflines").
"""
import collections

import torch

from . import ops

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.3 achievable)
MFMA_PEAK_TFLOPS = {0: 157.3, 1: 2500.0}   # dense fp32-input MFMA / dense bf16 MFMA (no sparsity)


def _gemm_cost(a):
    A, B, C, M, N, K = a[:6]
    kw = a[-1]
    flops = 2.0 * M * N * K
    byts = A.element_size() * M * K + B.element_size() * N * K + C.element_size() * M * N
    if kw.get("a_grid_n") or kw.get("a_jump_k"):              # grid addressing: the rows of a K range overlap, the operand is read once
        byts += A.element_size() * (A.numel() - M * K)
    if kw.get("b_seg_cols"):
        byts += B.element_size() * (B.numel() - N * K)
    if kw.get("res") is not None:
        byts += 4.0 * M * N
    if kw.get("gaux") is not None:
        byts += kw["gaux"].element_size() * M * N
    if kw.get("ln") is not None:                              # fused LayerNorm of the output rows: bf16 rows + mean / rstd
        byts += 2.0 * M * N + 8.0 * M
    return flops, byts


def _attn_cost(B, hw, C, products):
    wh = B * (hw // 8) ** 2 * (C // 32)
    return products * 2.0 * 64 * 64 * 32 * wh


COST = {
    "gemm": lambda *a, **k: _gemm_cost(a + (k,)),
    "winattn_fwd": lambda qkv, t, o, B, hw, C, sh, mode, lse=None: (_attn_cost(B, hw, C, 2), qkv.element_size() * B * hw * hw * 4.0 * C),
    "winattn_bwd": lambda qkv, t, do, dq, dt, B, hw, C, sh, mode, out=None, lse=None, scratch=None, defer=None: (_attn_cost(B, hw, C, 6), qkv.element_size() * B * hw * hw * 8.0 * C),   # q|k|v, dO, O read; dq|dk|dv written
    "dwconv_fwd": lambda h1, w, b, h2, B, hw, Ch, a2=None, g2=None: (18.0 * B * hw * hw * Ch, (1.0 + sum(t is not None for t in (h2, a2, g2))) * h1.element_size() * B * hw * hw * Ch),
    "dwconv_bwd": lambda h1, d, w, dh, dw, db, B, hw, Ch, scratch=None, defer=None: (54.0 * B * hw * hw * Ch, 3.0 * h1.element_size() * B * hw * hw * Ch),
    # fused attention half (C <= 128): q|k|v + proj products (8 C^2 MAC per token) + the window attention; forward reads x, writes x1 +
    # xn2 (+ xn1, q|k|v, o for the backward); backward reads dxn2, dy, x1, q|k|v, o, x and writes dx1h, dq|dk|dv, dx, dxh
    "attn_half_fwd": lambda x, g1, b1, mod, wq, bq, tb, wp, bp, g2, b2, x1, *a, **k:
        (8.0 * x.numel() * x.shape[1] + 256.0 * x.numel(), x.numel() * (10.0 + (10.0 if k.get("qkv") is not None else 0.0))),
    "attn_half_bwd": lambda dxn2, dy, *a, **k: (16.0 * dy.numel() * dy.shape[1] + 640.0 * dy.numel(), dy.numel() * 36.0),
    # fused LeFF (C <= 128): two [tokens x C x 4C] products + the depthwise taps; xn2 / dy read, x1 read, x2 / dxn2 written, the two
    # saved pre-activations written (forward) or read (backward) and dh1 written (halo re-reads are L2 traffic, not algorithmic)
    "leff_fwd": lambda xn2, x1, w1, b1, wd, bd, w2, b2, x2, B, hw, C, rowscale=None, h1=None, h2=None:
        (16.0 * B * hw * hw * C * C + 72.0 * B * hw * hw * C, B * hw * hw * C * (10.0 + (16.0 if h1 is not None else 0.0))),
    "leff_bwd": lambda dy, h1, h2, w1, wd, w2, dh1, dxn2, dwd, dbd, B, hw, C, rowscale=None, scratch=None, defer=None:
        (16.0 * B * hw * hw * C * C + 216.0 * B * hw * hw * C, B * hw * hw * C * (dy.element_size() + 26.0)),
    "layernorm_fwd": lambda x, g, b, m, y, *a: (8.0 * x.numel(), (4.0 + y.element_size()) * x.numel()),
    # dy, x and the residual gradient read, dx written (+ its bf16 copy for the next GEMM when asked for)
    "layernorm_bwd": lambda dy, x, *a, **k: (16.0 * x.numel(), (12.0 + dy.element_size() + (2.0 if k.get("dx_bf16") is not None else 0.0)) * x.numel()),
    "im2col": lambda x, cols, *a: (0.0, 4.0 * (x.numel() + cols.numel())),
    "col2im": lambda dc, dx, *a: (0.0, 4.0 * (dx.numel() + dc.numel())),
    "conv_x_to_grid": lambda x, xg, *a: (0.0, 4.0 * x.numel() + 2.0 * xg.numel()),
    "conv_dy_to_grid": lambda dy, dyg, *a: (0.0, float(dy.element_size()) * dy.numel() + 2.0 * dyg.numel()),
    "conv_grid_to_dx": lambda dxg, dx, *a: (0.0, 2.0 * dxg.numel() + 4.0 * dx.numel()),
    "pixel_shuffle": lambda y4, *a: (0.0, 8.0 * y4.numel()),
    "pixel_unshuffle": lambda ds, dy4, *a: (0.0, 8.0 * dy4.numel()),
    "copy_cols": lambda src, lds, s0, dst, ldd, c0, rows, C, accumulate=False: (0.0, 8.0 * rows * C),
    "axpbyz": lambda out, a, x, b, y, n: (2.0 * n, 12.0 * n),
    "istft_fwd": lambda s, w, B, terms=2: (2.0 * 10.2e6 * B, B * (131072.0 + 32008.0)),
    "stft_fwd": lambda w, s, B, terms=2: (2.0 * 8.4e6 * B, B * (131072.0 + 32008.0)),
    "istft_bwd": lambda w, s, B, terms=2: (2.0 * 8.4e6 * B, B * (131072.0 + 32008.0)),
    "stft_bwd": lambda s, w, B, terms=2: (2.0 * 10.2e6 * B, B * (131072.0 + 32008.0)),
    "attack_fwd": lambda d, w, o, s, B, L: (0.0, 8.0 * B * L),
    "attack_bwd": lambda d, w, do, s, dw, B, L: (0.0, 12.0 * B * L),
    "adamw": lambda p, g, m, v, n, *a: (12.0 * n, 28.0 * n),
}


class KernelTimer:
    def __init__(self, detail=False):
        self.records = collections.defaultdict(list)      # key -> [(start_evt, stop_evt, flops, bytes)]
        self._saved = {}
        self.detail = detail                               # key GEMMs by layout and shape as well

    def _key(self, name, args, kwargs):
        if name == "gemm":
            M, N, K = args[3:6]
            mode = kwargs.get("mode", 0)
            a_kc = args[9] if len(args) > 9 else kwargs.get("a_kc", 1)
            b_kc = args[10] if len(args) > 10 else kwargs.get("b_kc", 1)
            lay = {(1, 1): "fwd", (1, 0): "dX", (0, 0): "dW"}.get((a_kc, b_kc), "?")
            # kernel FAMILY as it appears in a rocprofv3 kernel trace: gemm_nt_kernel<...> runs forward and input-gradient
            # GEMMs, gemm_tn_kernel<...> the weight gradients (their split-K sums are deferred to reduce_jobs_kernel inside
            # LeWin blocks, summed by splitk_reduce_kernel elsewhere), gemm_kernel<...> is the fp32 parity-mode kernel
            fam = ("gemm_kernel", "gemm_nt_kernel", "gemm_tn_kernel", "gemm_nt_pp_kernel", "gemm_tn_pp_kernel", "gemm_nt_big_kernel",
                   "gemm_tn_big_kernel")[int(ops.lib.wm_last_gemm_family())]      # what the dispatcher launched (called after fn)
            if fam == "gemm_kernel" and not mode:
                fam = "gemm_kernel(fp32)"
            if self.detail:
                return f"{fam}_{lay}_M{M}_N{N}_K{K}", mode
            return fam, mode
        if name in ("conv_fwd", "conv_bwd") and self.detail:
            d = args[0]
            return f"{name}_Cin{d.Cin}_Cout{d.Cout}_H{d.H}_k{d.KH}{'_T' if d.transposed else ''}", None
        if self.detail and name in ("dwconv_bwd", "dwconv_fwd", "layernorm_bwd", "winattn_bwd", "winattn_fwd"):
            sz = args[0].numel()
            tag = f"{name}_n{sz}"
            md = args[9 if name == "winattn_bwd" else 7] if name.startswith("winattn") else None
            return (tag + ("_bf16" if md else "")), md
        if name.startswith("winattn"):
            md = args[9 if name == "winattn_bwd" else 7]
            return f"{name}_{'bf16' if md else 'f32'}", md
        return name, None

    def __enter__(self):
        for name in dir(ops):
            fn = getattr(ops, name)
            if not callable(fn) or name.startswith("_") or name in ("check", "ptr", "conv_desc", "linear_fwd", "ln_fusable",
                                                                     "linear_bwd_input", "linear_bwd_weight", "down_direct_ok",
                                                                     "down_direct_fwd", "down_direct_bwd", "conv_weights"):
                continue
            if getattr(fn, "__module__", "") != ops.__name__:
                continue
            self._saved[name] = fn
            setattr(ops, name, self._wrap(name, fn))
        return self

    def _wrap(self, name, fn):
        def timed(*args, **kwargs):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn(*args, **kwargs)
            e.record()
            cost = COST.get(name)
            fl, by = cost(*args, **kwargs) if cost else (0.0, 0.0)
            key, mode = self._key(name, args, kwargs)
            self.records[key].append((s, e, fl, by, mode))
            return out
        return timed

    def __exit__(self, *exc):
        for name, fn in self._saved.items():
            setattr(ops, name, fn)
        return False

    def summary(self):
        torch.cuda.synchronize()
        rows = []
        for key, recs in self.records.items():
            t = sum(s.elapsed_time(e) for s, e, *_ in recs) * 1e-3
            # roofline time of the family with the bound chosen PER LAUNCH (a family mixes shapes on both sides of the ridge)
            mfma = key.startswith("gemm") or key.startswith("winattn")
            peak = MFMA_PEAK_TFLOPS[1 if recs[0][4] else 0] * 1e12
            bound_s = sum(max(r[2] / peak if mfma else 0.0, r[3] / (HBM_PEAK_GBS * 1e9)) for r in recs)
            rows.append(dict(kernel=key, launches=len(recs), total_s=t, avg_us=t / len(recs) * 1e6,
                             flops=sum(r[2] for r in recs), bytes=sum(r[3] for r in recs), mode=recs[0][4], bound_s=bound_s))
        rows.sort(key=lambda r: -r["total_s"])
        return rows


def roofline_of(row):
    """Roofline object (bench.py contract) for one summary row.  The bound is whichever of the two rooflines needs MORE
    time for the launch's algorithmic work (flops / dense MFMA peak vs bytes / HBM peak); both are reported."""
    k, n = row["kernel"], row["launches"]
    peak_tf = MFMA_PEAK_TFLOPS[1 if row["mode"] else 0]
    t_mfma = row["flops"] / (peak_tf * 1e12) if (k.startswith("gemm") or k.startswith("winattn")) else 0.0
    t_hbm = row["bytes"] / (HBM_PEAK_GBS * 1e9)
    tf, gb = row["flops"] / row["total_s"] / 1e12, row["bytes"] / row["total_s"] / 1e9
    common = dict(kernel=k, launches=n, avg_launch_us=row["avg_us"], alg_flops_per_launch=row["flops"] / n,
                  alg_bytes_per_launch=row["bytes"] / n, achieved_TFLOPs=tf, achieved_GBs=gb,
                  frac_of_mfma_peak=tf / peak_tf, frac_of_hbm_peak=gb / HBM_PEAK_GBS, traffic=None,
                  frac_bound_per_launch=row.get("bound_s", 0.0) / row["total_s"])
    if t_mfma > t_hbm:
        return dict(bound="mfma", achieved=tf, peak=peak_tf, unit="TFLOP/s", frac=tf / peak_tf, **common)
    return dict(bound="hbm", achieved=gb, peak=HBM_PEAK_GBS, unit="GB/s", frac=gb / HBM_PEAK_GBS, **common)


STFT_PATH = ("istft_fwd", "stft_fwd", "istft_bwd", "stft_bwd", "attack_fwd", "attack_bwd")

# KernelTimer key -> __global__ functions (rocprofv3 kernel families) that carry it
FAMILY_KERNELS = {"reduce_jobs": ["reduce_jobs_kernel"], "winattn_bwd_bf16": ["attn_bwd_fast_kernel"],
                  "winattn_fwd_bf16": ["attn_fwd_fast_kernel"], "layernorm_fwd": ["ln_fwd_kernel"],
                  "layernorm_bwd": ["ln_bwd_kernel"], "istft_fwd": ["synth_mfma_kernel<false>"], "stft_bwd": ["synth_mfma_kernel<true>"],
                  "stft_fwd": ["analysis_mfma_kernel<false>"], "istft_bwd": ["analysis_mfma_kernel<true>"],
                  "attack_fwd": ["attack_fwd_kernel"], "attack_bwd": ["attack_bwd_kernel"],
                  }


def time_stft_path(wm, batch, attack_cfg, reps=20):
    """HIP-event time per launch of the six short kernels of the STFT / iSTFT / attack path (9-50 us each), each replayed
    `reps` times from a captured graph on the launch stream: one event pair around a single 15 us launch measures mostly
    the events (the eager pass of KernelTimer reads 8-12 us high for them; rocprofv3 agrees with this form)."""
    B = int(batch)
    dev = torch.device("cuda")
    spec, wave = torch.randn(B, 2, 128, 128, device=dev), 0.1 * torch.randn(B, 8002, device=dev)
    spec2, wave2 = torch.empty_like(spec), torch.empty_like(wave)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    fns = {"istft_fwd": lambda: ops.istft_fwd(spec, wave2, B), "stft_fwd": lambda: ops.stft_fwd(wave, spec2, B),
           "istft_bwd": lambda: ops.istft_bwd(wave, spec2, B), "stft_bwd": lambda: ops.stft_bwd(spec, wave2, B)}
    if attack_cfg is not None:
        out, sigma = wm.attacks.attack_forward(attack_cfg, wave, step)
        fns["attack_fwd"] = lambda: wm.attacks.attack_forward(attack_cfg, wave, step, out=out, sigma=sigma)
        fns["attack_bwd"] = lambda: wm.attacks.attack_backward(attack_cfg, wave, wave2, sigma, step, dwave=out)
    res = {}
    for name, fn in fns.items():
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        times = []
        for _ in range(5):                                               # median of five replays: one replay can catch a hiccup
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            g.replay()
            e.record()
            torch.cuda.synchronize()
            times.append(s.elapsed_time(e))
        res[name] = sorted(times)[2] / reps * 1e-3                      # seconds per launch
    return res


def source_id():
    """Identity of the kernel sources a measurement belongs to: sha256 over csrc/*.hip, csrc/*.h and include/*.h."""
    import glob
    import hashlib
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h")) +
                   glob.glob(os.path.join(here, "..", "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def find_pmc_summary(profiles_dir, batch, mode, attack):
    """Newest committed PMC families summary (scripts/gpu_profile.sh) taken at this batch / mode / attack ON THESE kernel
    sources, or None: traffic from another build is not reported as this build's."""
    import glob
    import json
    import os
    sid = source_id()
    for path in sorted(glob.glob(os.path.join(profiles_dir, f"r*_families_b{batch}_{mode}_{attack}.json")), reverse=True):
        try:
            if json.load(open(path)).get("source_id") == sid:
                return path
        except (OSError, ValueError):
            continue
    return None


def pmc_traffic(kernel_family, path):
    """HBM bytes per launch of a kernel family from a PMC summary (scripts/summarize_rocprof.py --families output:
    2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md HBM section), or None when no summary is present."""
    import json
    import os
    if not path or not os.path.exists(path):
        return None
    fams = json.load(open(path)).get("families", {})
    names = FAMILY_KERNELS.get(kernel_family, [kernel_family if kernel_family.endswith("_kernel") else kernel_family + "_kernel"])
    tot, launches = 0.0, 0
    for nm in names:
        f = fams.get(nm)
        if not f:
            continue                                         # a member kernel that did not run at this batch
        if f.get("hbm_bytes") is None:
            return None
        tot += f["hbm_bytes"]
        launches += f["launches"]
    return tot / launches if launches else None


def pmc_step_traffic(path, tiles):
    """Whole-step HBM traffic from a PMC summary: sum over kernel families of (2 x FETCH_SIZE + WRITE_SIZE) per launch x launches
    per step (the trace's step count = its adamw_kernel launches), per step and per tile, and the rate over the summed kernel time.
    None without a summary."""
    import json
    import os
    if not path or not os.path.exists(path):
        return None
    fams = json.load(open(path)).get("families", {})
    steps = fams.get("adamw_kernel", {}).get("launches", 0)
    if not steps:
        return None
    byts = sum((f.get("hbm_bytes_per_launch") or 0.0) * f["launches"] for f in fams.values()) / steps
    ms = sum(f["total_ms"] for f in fams.values()) / steps
    return {"GB_per_step": byts / 1e9, "GB_per_tile": byts / 1e9 / max(int(tiles), 1), "kernel_ms_per_step": ms,
            "TBps_over_kernel_time": byts / (ms * 1e-3) / 1e12 if ms > 0 else None,
            "launches_per_step": sum(f["launches"] for f in fams.values()) / steps,
            "formula": "sum over families of (2 x FETCH_SIZE + WRITE_SIZE) per launch x launches / steps in the trace"}


def pmc_mfma_util(path):
    """MFMA utilisation from counters (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)) of the MFMA-carrying
    kernel families and of the whole step, as scripts/summarize_rocprof.py wrote them; None without a summary."""
    import json
    import os
    if not path or not os.path.exists(path):
        return None
    return json.load(open(path)).get("mfma_util")
