"""Host-side mirror of the reference watermark model (uformerWM/model.py) on top of libwm_hip.so.

Same constructor arguments, method names, return tuple and ``state_dict`` keys as the reference's
``UformerAudio`` (model.py:2225-2512), so ``utils.get_arch(opt)`` call sites, checkpoints and the
trainer loop of ``audio_uformer_stft.py`` work unchanged.  What is different is everything underneath:

* parameters (and their gradients / AdamW moments) live in ONE flat fp32 arena per kind; every
  ``nn.Parameter`` is a view into it.  One memset zeroes all gradients, one kernel applies AdamW, and a
  data-parallel all-reduce works on three contiguous slices laid out in backward-completion order;
* forward and backward are explicit sequences of HIP kernel launches (no autograd graph inside the
  model): LayerNorm(+modulator) -> fused QKV GEMM -> window attention -> proj GEMM(+DropPath+residual)
  -> LayerNorm -> GEMM -> depthwise 3x3 -> GEMM(GELU on load, +DropPath+residual); activations stay
  token-major [B, N, C] end to end, rolls / window partitions / NCHW<->NHWC permutes never touch HBM;
* to the outside it is one ``torch.autograd.Function``: ``loss.backward()`` of a reference-style
  training loop calls the hand-written backward, which accumulates straight into ``param.grad`` views.
There is no eager/CPU fallback: CPU tensors raise.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .attacks import AttackConfig, attack_forward, attack_backward, rank_seed

DEPTHS = (1, 2, 8, 8, 2, 8, 8, 2, 1)
HEADS = (1, 2, 4, 8, 16, 16, 8, 4, 2)
WAVE_LEN = 8002


class _Holder(nn.Module):
    """Name-space node so that parameter paths equal the reference's (e.g. ``mlp.linear1.0.weight``)."""


def _rel_pos_index(ws=8):
    ii = torch.arange(ws)
    gy, gx = torch.meshgrid(ii, ii, indexing="ij")
    p = torch.stack((gy.reshape(-1), gx.reshape(-1)))
    d = p[:, :, None] - p[:, None, :] + (ws - 1)
    return d[0] * (2 * ws - 1) + d[1]


# ------------------------------------------------------------------------------------------------
# architecture description -> ordered parameter specs
# ------------------------------------------------------------------------------------------------
def _block_specs(prefix, C, heads, modulator):
    s = []
    if modulator:
        s.append((prefix + "modulator.weight", (64, C), "normal1"))
    s += [(prefix + "norm1.weight", (C,), "ones"), (prefix + "norm1.bias", (C,), "zeros"),
          (prefix + "attn.relative_position_bias_table", (225, heads), "trunc02"),
          # q and kv projection weights / biases adjacent => zero-copy fused [3C, C] QKV operand
          (prefix + "attn.qkv.to_q.weight", (C, C), "trunc02"), (prefix + "attn.qkv.to_kv.weight", (2 * C, C), "trunc02"),
          (prefix + "attn.qkv.to_q.bias", (C,), "zeros"), (prefix + "attn.qkv.to_kv.bias", (2 * C,), "zeros"),
          (prefix + "attn.proj.weight", (C, C), "trunc02"), (prefix + "attn.proj.bias", (C,), "zeros"),
          (prefix + "norm2.weight", (C,), "ones"), (prefix + "norm2.bias", (C,), "zeros"),
          (prefix + "mlp.linear1.0.weight", (4 * C, C), "trunc02"), (prefix + "mlp.linear1.0.bias", (4 * C,), "zeros"),
          (prefix + "mlp.dwconv.0.weight", (4 * C, 1, 3, 3), "conv"), (prefix + "mlp.dwconv.0.bias", (4 * C,), "convb:9"),
          (prefix + "mlp.linear2.0.weight", (C, 4 * C), "trunc02"), (prefix + "mlp.linear2.0.bias", (C,), "zeros")]
    return s


def _conv_specs(prefix, cout, cin, kh, kw, transposed=False):
    shape = (cin, cout, kh, kw) if transposed else (cout, cin, kh, kw)
    fan_in = (cout if transposed else cin) * kh * kw          # torch's fan_in convention (dim 1 * receptive field)
    return [(prefix + "weight", shape, "conv"), (prefix + "bias", (cout,), f"convb:{fan_in}")]


class _Arch:
    def __init__(self, img, embed, depths, heads, dp_rate, modulator):
        self.img, self.embed, self.depths, self.heads, self.dp_rate, self.modulator = img, embed, depths, heads, dp_rate, modulator
        enc = [x.item() for x in torch.linspace(0, dp_rate, sum(depths[:4]))]     # model.py:1268
        self.enc_dpr = []
        o = 0
        for i in range(4):
            self.enc_dpr.append(enc[o:o + depths[i]])
            o += depths[i]
        self.enc_dpr.append([dp_rate] * depths[4])                                  # model.py:1270
        dec = enc[::-1]                                                            # model.py:1126
        self.dec_dpr = []
        o = 0
        for i in range(4):
            self.dec_dpr.append(dec[o:o + depths[5 + i]])
            o += depths[5 + i]

    def trunk_blocks(self):
        """(stage, block, C, hw, heads, shift, rate)"""
        out = []
        for i in range(5):
            C, hw = self.embed << i, self.img >> i
            for j in range(self.depths[i]):
                shift = 0 if (j % 2 == 0 or hw <= 8) else 4
                out.append((i, j, C, hw, self.heads[i], shift, self.enc_dpr[i][j]))
        return out

    def dec_blocks(self):
        out = []
        for i in range(4):
            C, hw = self.embed << (4 - i), self.img >> (3 - i)
            for j in range(self.depths[5 + i]):
                shift = 0 if (j % 2 == 0 or hw <= 8) else 4
                out.append((i, j, C, hw, self.heads[5 + i], shift, self.dec_dpr[i][j]))
        return out


def _trunk_specs(arch, prefix, with_conv2):
    s = _conv_specs(prefix + "input_proj.proj.0.", arch.embed, 2, 3, 3)
    for (i, j, C, hw, heads, shift, rate) in arch.trunk_blocks():
        name = f"{prefix}encoderlayer_{i}.blocks.{j}." if i < 4 else f"{prefix}conv.blocks.{j}."
        s += _block_specs(name, C, heads, False)          # encoders never receive the modulator (model.py:1276-1379)
    for i in range(4):
        C = arch.embed << i
        s += _conv_specs(f"{prefix}dowsample_{i}.conv.0.", 2 * C, C, 4, 4)
    if with_conv2:
        s += _conv_specs(prefix + "conv2.", 1, 1, 8, 8)
    return s


def _decoder_specs(arch, prefix):
    s = []
    cin = [arch.embed * 32, arch.embed * 16, arch.embed * 8, arch.embed * 4]
    for i in range(4):
        s += _conv_specs(f"{prefix}upsample_{i}.deconv.0.", (arch.embed << (4 - i)) // 2, cin[i], 2, 2, transposed=True)
    for (i, j, C, hw, heads, shift, rate) in arch.dec_blocks():
        s += _block_specs(f"{prefix}decoderlayer_{i}.blocks.{j}.", C, heads, arch.modulator)
    return s


def _init_tensor(t, kind):
    if kind == "ones":
        t.fill_(1.0)
    elif kind == "zeros":
        t.zero_()
    elif kind == "trunc02":
        nn.init.trunc_normal_(t, std=0.02)
    elif kind == "normal1":
        t.normal_()
    elif kind == "conv":                                   # torch Conv default: kaiming_uniform(a=sqrt(5))
        fan_in = t.shape[1] * t.shape[2] * t.shape[3]
        bound = 1.0 / math.sqrt(fan_in)
        t.uniform_(-bound, bound)
    elif kind.startswith("convb:"):
        bound = 1.0 / math.sqrt(int(kind.split(":")[1]))
        t.uniform_(-bound, bound)
    else:
        raise ValueError(kind)


# ------------------------------------------------------------------------------------------------
# the autograd boundary
# ------------------------------------------------------------------------------------------------
class _ModelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, net, x, message, opts):
        outs, saved = net._forward_impl(x, message, **opts)
        ctx.net, ctx.saved = net, saved
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        ctx.net._backward_impl(ctx.saved, *douts)
        ctx.saved = None
        return None, None, None, None, None


def _E(dev, *shape):
    return torch.empty(shape, device=dev, dtype=torch.float32)


def _Z(dev, *shape):
    t = torch.empty(shape, device=dev, dtype=torch.float32)
    return ops.zero_(t) if t.is_cuda else t.zero_()


class UformerAudio(nn.Module):
    """Drop-in for reference ``model.UformerAudio`` (model.py:2225).  ``forward`` returns
    ``(stft_new, noise, wm_pred, wm)``; ``wm_decode`` / ``feature_extract`` as in model.py:2379 / :2345.

    Extra (keyword-only, default = reference behaviour):
      compute_mode   'fp32' (exact-accumulation parity mode), 'bf16' (bf16 storage of the wide activations, bf16 MFMA operands,
                     fp32 accumulate) or 'split' (fp32 storage as in 'fp32', every GEMM as three bf16 MFMA products of
                     hi / lo operand halves: the parity tolerance at 3/16 instead of 1/16 of the matrix rate)
      attack         AttackConfig applied to the waveform between iSTFT and STFT (hook model.py:2461)
      extract_from   'y' (reference, model.py:2508) or 'attacked' (the commented alternative, model.py:2507)
    """

    def __init__(self, img_size=128, in_chans=2, dd_in=2, embed_dim=32, depths=DEPTHS, num_heads=HEADS, win_size=8,
                 mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1,
                 norm_layer=nn.LayerNorm, patch_norm=True, use_checkpoint=False, token_projection="linear",
                 token_mlp="leff", shift_flag=True, modulator=False, cross_modulator=False, audio_scale=0,
                 data_min=0, data_max=1, compute_mode="fp32", attack=None, extract_from="y", seed=1234, **kwargs):
        super().__init__()
        if (img_size, in_chans, dd_in, embed_dim, win_size) != (128, 2, 2, 32, 8) or token_projection != "linear" \
                or token_mlp != "leff" or cross_modulator or not shift_flag or mlp_ratio != 4.0:
            raise Exception("Arch error!")                  # same failure style as utils/model_utils.py:107
        self.data_min, self.data_max, self.audio_scale = data_min, data_max, audio_scale
        self.compute_mode = compute_mode
        self.attack = attack
        self.extract_from = extract_from
        self.seed = int(seed)
        self.arch = _Arch(img_size, embed_dim, tuple(depths), tuple(num_heads), float(drop_path_rate), bool(modulator))
        a = self.arch
        # arena order = reverse of backward completion: [A | B | C | unused]
        seg_a = (_conv_specs("encoder_wm.conv1.", 16, 1, 3, 3) + _conv_specs("encoder_wm.conv2.", 4, 16, 3, 3) +
                 _conv_specs("encoder_wm.t_conv1.", 16, 4, 2, 2, True) + _conv_specs("encoder_wm.t_conv2.", 1, 16, 2, 2, True) +
                 _conv_specs("input_proj.proj.0.", embed_dim, dd_in, 3, 3) + _trunk_specs(a, "encoder.", False))
        unused = [s for s in seg_a if s[0].startswith("encoder.input_proj.")]
        seg_a = [s for s in seg_a if not s[0].startswith("encoder.input_proj.")]
        seg_b = _decoder_specs(a, "decoder.") + _conv_specs("output_proj.proj.0.", in_chans, 2 * embed_dim, 3, 3)
        seg_c = (_conv_specs("stft_layer.0.", 4, in_chans, 3, 3) + _conv_specs("stft_layer.2.", in_chans, 4, 3, 3) +
                 _trunk_specs(a, "decoder_wm.", True))
        self._specs = seg_a + seg_b + seg_c + unused
        # every tensor starts on a 32-byte boundary: 16-B loads stay aligned in the fp32 arena AND in its bf16 copy;
        # q|kv weights and biases stay adjacent because their sizes are multiples of 8 elements
        self._offsets, self._seg_sizes, off = {}, [], 0
        for seg in (seg_a, seg_b, seg_c, unused):
            start = off
            for name, shape, _ in seg:
                self._offsets[name] = off
                off += (math.prod(shape) + 7) // 8 * 8
            self._seg_sizes.append(off - start)
        # gradient-ready granules for the data-parallel exchange: a LeWin block or a convolution owns one contiguous arena run
        # (its parameters are consecutive specs); runs tile the optimised part of the arena without gaps
        self._owner_range, runs = {}, []
        for name, shape, _ in seg_a + seg_b + seg_c:
            owner = self._owner_of(name)
            if not runs or runs[-1][0] != owner:
                runs.append([owner, self._offsets[name]])
        ends = [r[1] for r in runs[1:]] + [sum(self._seg_sizes[:3])]
        for (owner, lo), hi in zip(runs, ends):
            self._owner_range[owner] = (lo, hi)
        self._build_arena(torch.device("cpu"), init=True)
        self._anchor = None
        self._step_t = None
        self._grad_ready = None

    @staticmethod
    def _owner_of(name):
        parts = name.split(".")
        if "blocks" in parts:
            k = parts.index("blocks")
            return ".".join(parts[:k + 2]) + "."
        return ".".join(parts[:-1]) + "."

    def segment_bounds(self):
        """[lo, hi) arena element ranges of the three gradient segments in the order backward finishes them (extractor +
        stft_layer, decoder + output_proj, encoder + payload autoencoder)."""
        a, b, c = self._seg_sizes[:3]
        return [(a + b, a + b + c), (a, a + b), (0, a)]

    # ---- parameter arena -------------------------------------------------------------------
    def _build_arena(self, device, init=False, flat=None):
        total = sum(self._seg_sizes)
        pad = (-total) % 64
        self._flat = torch.zeros(total + pad, device=device, dtype=torch.float32) if flat is None else flat
        self._flat_grad = torch.zeros(total + pad, device=device, dtype=torch.float32)
        self._flat_h = None                               # bf16 copy of the arena (GEMM B operands in bf16 mode)
        self._flat_ht, self._tjobs = None, None           # transposed bf16 copies of the wide Linear weights (dX operands)
        self._views, self._gviews = {}, {}
        for name, shape, kind in self._specs:
            n, off = math.prod(shape), self._offsets[name]
            v = self._flat[off:off + n].view(shape)
            if init:
                _init_tensor(v, kind)
            self._views[name], self._gviews[name] = v, self._flat_grad[off:off + n].view(shape)
        for name, shape, kind in self._specs:
            node = self
            parts = name.split(".")
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Holder())
                node = node._modules[part]
            p = nn.Parameter(self._views[name])
            p.grad = self._gviews[name]
            node._parameters[parts[-1]] = p
            if name.endswith("relative_position_bias_table"):
                node._buffers["relative_position_index"] = _rel_pos_index().to(device)
        # fused QKV aliases
        for name in list(self._views):
            if name.endswith("attn.qkv.to_q.weight"):
                pre = name[:-len("to_q.weight")]
                C = self._views[name].shape[0]
                o = self._offsets[name]
                ob = self._offsets[pre + "to_q.bias"]
                self._views[pre + "w"], self._gviews[pre + "w"] = self._flat[o:o + 3 * C * C].view(3 * C, C), self._flat_grad[o:o + 3 * C * C].view(3 * C, C)
                self._views[pre + "b"], self._gviews[pre + "b"] = self._flat[ob:ob + 3 * C], self._flat_grad[ob:ob + 3 * C]

    def _apply(self, fn, recurse=True):
        """``.to()/.cuda()`` re-materialise the arena on the new device and re-point every Parameter at it."""
        probe = fn(self._flat[:1])
        if probe.device != self._flat.device or probe.dtype != self._flat.dtype:
            if probe.dtype != torch.float32:
                raise TypeError("the watermark model keeps fp32 master parameters; choose compute_mode='bf16' instead")
            self._build_arena(probe.device, flat=fn(self._flat).contiguous())
            self._anchor, self._step_t = None, None
        return self

    @property
    def optimized_numel(self):
        return sum(self._seg_sizes[:3])

    def grad_buckets(self):
        """Contiguous gradient slices in the order backward finishes them (extractor, decoder, encoder)."""
        a, b, c = self._seg_sizes[:3]
        g = self._flat_grad
        return [g[a + b:a + b + c], g[a:a + b], g[:a]]

    # ---- reference API -----------------------------------------------------------------------
    def _mode(self):
        return {"fp32": ops.MODE_FP32, "bf16": ops.MODE_BF16, "split": ops.MODE_SPLIT}[self.compute_mode]

    # bf16 mode stores the WIDE activations (LN outputs, qkv, attention output, LeFF hidden tensors, im2col buffers and
    # their gradients) and a copy of the weights in bf16; the residual stream, parameters, gradients and LN statistics
    # stay fp32.  fp32 ("parity") mode keeps everything fp32.
    def _H(self, dev, *shape):
        return torch.empty(shape, device=dev, dtype=torch.bfloat16 if self.compute_mode == "bf16" else torch.float32)

    def _scratch(self, dev, floats):
        """Workspace for per-workgroup partial sums of small parameter gradients (bias table, depthwise weights): one
        buffer, reused by every layer, grown on demand (before any graph capture: the eager warm-up sizes it)."""
        sc = getattr(self, "_scr", None)
        if sc is None or sc.device != dev or sc.numel() < floats:
            if sc is not None:                                     # a captured step may still point into the old buffer
                self.__dict__.setdefault("_scr_retired", []).append(sc)
            sc = self._scr = torch.empty(int(floats), device=dev)
        return sc

    def _gemm_scratch(self, dev):
        return self._scratch(dev, ops.lib.wm_gemm_scratch_floats())

    def _block_scratch(self, dev, B, hw, C):
        """Eight disjoint scratch slices for the deferred reductions of one LeWin block's backward (4 split-K GEMMs, 2
        LayerNorms, depthwise conv, attention bias table) + the host-side job list; one launch sums them all."""
        sizes = [int(ops.lib.wm_gemm_scratch_floats())] * 4 + [int(ops.lib.wm_layernorm_bwd_scratch_floats(C))] * 2 + \
                [int(ops.lib.wm_dwconv_bwd_scratch_floats(B, hw, 4 * C)), int(ops.lib.wm_winattn_bwd_scratch_floats()),
                 int(ops.lib.wm_attn_half_bwd_scratch_floats(B, hw, C, 1))]          # (0 where the fused attention half does not apply)
        sizes = [(n + 63) // 64 * 64 for n in sizes]
        buf = getattr(self, "_blk_scr", None)
        if buf is None or buf.device != dev or buf.numel() < sum(sizes):
            if buf is not None:                                    # a captured step may still point into the old buffer
                self.__dict__.setdefault("_scr_retired", []).append(buf)
            buf = self._blk_scr = torch.empty(sum(sizes), device=dev)
        out, off = [], 0
        for n in sizes:
            out.append(buf[off:off + n])
            off += n
        jobs = getattr(self, "_jobs", None)
        if jobs is None:
            jobs = self._jobs = ops.ReduceJobs()
        return out, jobs

    def _attn_scratch(self, dev):
        return self._scratch(dev, ops.lib.wm_winattn_bwd_scratch_floats())

    # Linear weights of the blocks with C >= BIG_C also get a TRANSPOSED bf16 copy per step: their dX products then run as
    # K-contiguous x K-contiguous GEMMs on the 256 x 256 tile kernel (csrc/gemm_big.hip) like the forward ones
    BIG_C = 256

    def _transpose_jobs(self):
        if self._tjobs is None or self._tjobs[0].device != self._flat.device:
            rows, tile0 = [], 0
            for name, shape, _ in self._specs:
                if len(shape) != 2:
                    continue
                big = name.endswith(("attn.qkv.to_q.weight", "attn.proj.weight", "mlp.linear1.0.weight",
                                     "mlp.linear2.0.weight")) and min(shape) >= self.BIG_C
                # Wqkv^T of the blocks whose attention half runs as one launch (csrc/attn_fused.hip: C <= 128)
                small = name.endswith("attn.qkv.to_q.weight") and shape[0] <= 128
                if big or small:
                    n, k = shape
                    if name.endswith("to_q.weight"):
                        n = 3 * n                                 # the fused [3C, C] q | kv operand (adjacent in the arena)
                    rows.append((self._offsets[name], n, k, tile0))
                    tile0 += ((n + 63) // 64) * ((k + 63) // 64)
            self._tjobs = (torch.tensor(rows, dtype=torch.int32, device=self._flat.device).contiguous(), len(rows), tile0)
        return self._tjobs

    def _refresh_weight_copy(self, transposed=True):
        """bf16 copy of the arena (every forward), and the TRANSPOSED copies of the wide Linear weights, which only the backward's
        dX products read: skipped for inference (wm_decode, feature_extract, forward under no_grad)."""
        if self.compute_mode != "bf16":
            return
        if self._flat_h is None or self._flat_h.device != self._flat.device:
            self._flat_h = torch.empty(self._flat.numel(), device=self._flat.device, dtype=torch.bfloat16)
            self._flat_ht = None
        ops.cast_bf16(self._flat, self._flat_h, self._flat.numel())
        if not transposed:
            return
        jobs, njobs, tiles = self._transpose_jobs()
        if njobs:
            if self._flat_ht is None or self._flat_ht.device != self._flat.device:
                self._flat_ht = torch.empty(self._flat.numel(), device=self._flat.device, dtype=torch.bfloat16)
            ops.transpose_cast_jobs(self._flat, self._flat_ht, jobs, njobs, tiles)

    def _W(self, name):
        """GEMM weight operand: fp32 parameter view, or its bf16 copy in bf16 mode."""
        v = self._views[name]
        if self.compute_mode != "bf16":
            return v
        off = (v.data_ptr() - self._flat.data_ptr()) // 4
        return self._flat_h[off:off + v.numel()].view(v.shape)

    def _WT(self, name):
        """Transposed bf16 copy [in, out] of a wide Linear weight (bf16 mode, C >= BIG_C), else None."""
        v = self._views[name]
        if self.compute_mode != "bf16" or min(v.shape) < self.BIG_C or self._flat_ht is None:
            return None                                            # (no copy: the product reads the weight as it lies)
        off = (v.data_ptr() - self._flat.data_ptr()) // 4
        return self._flat_ht[off:off + v.numel()].view(v.shape[1], v.shape[0])

    def _WqkvT(self, pre):
        """[C, 3C] bf16 transpose of a block's fused q | k | v weight from the per-step batched copy (operand of wm_attn_half_bwd)."""
        v = self._views[pre + "attn.qkv.to_q.weight"]
        C = v.shape[0]
        if self.compute_mode != "bf16" or self._flat_ht is None:
            return None
        off = (v.data_ptr() - self._flat.data_ptr()) // 4
        return self._flat_ht[off:off + 3 * C * C].view(C, 3 * C)

    def _ensure_runtime(self, device):
        if self._anchor is None or self._anchor.device != device:
            self._anchor = torch.zeros(1, device=device, requires_grad=True)
            self._step_t = torch.zeros(1, dtype=torch.int64, device=device)
            rates = [r for (_, _, _, _, _, _, r) in self.arch.trunk_blocks() for _ in (0, 1)]
            rates += [r for (_, _, _, _, _, _, r) in self.arch.dec_blocks() for _ in (0, 1)]
            rates += [r for (_, _, _, _, _, _, r) in self.arch.trunk_blocks() for _ in (0, 1)]
            self._rates = torch.tensor(rates, device=device, dtype=torch.float32)
            self._rates_host = rates

    def forward(self, x, message, mask=None, keep_scales=None):
        if not x.is_cuda:
            raise RuntimeError("UformerAudio (MI355X build) runs on the GPU only; move the model and inputs to cuda")
        self._ensure_runtime(x.device)
        self._reattach_grads()
        if self.training and keep_scales is None:
            # a reference-style loop (net(x, msg); loss.backward(); optim.step()) advances the DropPath / AWGN Philox
            # streams here, once per training forward; TrainStep advances the same counter in its own launch sequence
            ops.step_inc(self._step_t)
        opts = dict(train=self.training, keep_scales=keep_scales, attack=self.attack, extract_from=self.extract_from)
        return _ModelFn.apply(self._anchor, self, x.contiguous().float(), message.contiguous().float(), opts)

    def wm_decode(self, y):
        self._ensure_runtime(y.device)
        with torch.no_grad():
            B = y.shape[0]
            self._refresh_weight_copy(transposed=False)
            feat, _ = self._trunk_fwd("decoder_wm.", y.contiguous().float(), B, None, 0, extractor=True)
            return self._ae_decode_fwd(feat, B)[0]

    def feature_extract(self, x, message):
        self._ensure_runtime(x.device)
        with torch.no_grad():
            outs, _ = self._forward_impl(x.contiguous().float(), message.contiguous().float(), train=False,
                                         keep_scales=None, attack=None, extract_from="y", embed_only=True)
            return outs

    def _reattach_grads(self):
        """``optimizer.zero_grad(set_to_none=True)`` drops our views; put (zeroed) views back before backward."""
        lost = False
        for name, p in self.named_parameters():
            if p.grad is None or p.grad.data_ptr() != self._gviews[name].data_ptr():
                lost = True
                break
        if lost:
            self._flat_grad.zero_()
            for name, p in self.named_parameters():
                p.grad = self._gviews[name]

    # ---- building blocks ---------------------------------------------------------------------
    def _scales_for(self, B, train, keep_scales, dev):
        if keep_scales is not None:
            return keep_scales.contiguous()
        if not train or self.arch.dp_rate == 0.0:
            return None
        n = len(self._rates_host)
        sc = _E(dev, n, B)
        ops.droppath_scales(sc, self._rates, n, B, rank_seed(self.seed), self._step_t)
        return sc

    def _block_fwd(self, pre, x, B, C, hw, shift, s1, s2, mod, ln1=None, next_ln=None):
        """One LeWin block (model.py:937-1019).  In bf16 mode the two LayerNorms ride in the epilogue of the GEMM that produces
        their input where a tile holds whole rows (C <= 128, C = 256 on the 256 x 256 tiles): norm2 behind the attention
        projection, and the NEXT block's norm1 behind fc2 (``next_ln`` = (prefix, shift, modulator?) of that block; its result is
        returned and handed to that block as ``ln1``).  Returns (x2, tape, next block's (xn1, mean1, rstd1) or None)."""
        P, dev, mode = self._views, x.device, self._mode()
        M, N = x.shape[0], hw * hw
        H, W = self._H, self._W
        half = self.compute_mode == "bf16"
        attn_fused = half and ops.attn_half_fusable(hw, C)
        if attn_fused:
            # C <= 128: norm1 (+ modulator) -> q | k | v -> window attention -> projection (+ DropPath, + residual) -> norm2 as ONE
            # window-local launch (csrc/attn_fused.hip); xn1 / qkv / o are written for the backward launches
            xn1, qkv, o = H(dev, M, C), H(dev, M, 3 * C), H(dev, M, C)
            mean1, rstd1, mean2, rstd2, lse = _E(dev, M), _E(dev, M), _E(dev, M), _E(dev, M), _E(dev, M, C // 32)
            x1, xn2 = _E(dev, M, C), H(dev, M, C)
            ops.attn_half_fwd(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"], P[pre + "modulator.weight"] if mod else None,
                              W(pre + "attn.qkv.w"), P[pre + "attn.qkv.b"], P[pre + "attn.relative_position_bias_table"],
                              W(pre + "attn.proj.weight"), P[pre + "attn.proj.bias"], P[pre + "norm2.weight"], P[pre + "norm2.bias"],
                              x1, xn2, mean1, rstd1, mean2, rstd2, lse, B, hw, C, shift, rowscale=s1, xn1=xn1, qkv=qkv, o=o)
            next_ln = None                                         # the next block of this stage normalises its own input
        elif ln1 is None:
            xn1, mean1, rstd1 = H(dev, M, C), _E(dev, M), _E(dev, M)
            ops.layernorm_fwd(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"], P[pre + "modulator.weight"] if mod else None,
                              xn1, mean1, rstd1, hw, shift)
        else:
            xn1, mean1, rstd1 = ln1
        if not attn_fused:
            qkv = H(dev, M, 3 * C)
            ops.linear_fwd(xn1, W(pre + "attn.qkv.w"), P[pre + "attn.qkv.b"], qkv, mode)
            o = H(dev, M, C)
            lse = _E(dev, M, C // 32) if mode else None            # the MFMA kernels (bf16 and split) keep the row log-sum-exp
            ops.winattn_fwd(qkv, P[pre + "attn.relative_position_bias_table"], o, B, hw, C, shift, mode, lse=lse)
            x1 = _E(dev, M, C)
            xn2, mean2, rstd2 = H(dev, M, C), _E(dev, M), _E(dev, M)
        if attn_fused:
            pass
        elif half and ops.ln_fusable(M, C, C, mode):
            ops.linear_fwd(o, W(pre + "attn.proj.weight"), P[pre + "attn.proj.bias"], x1, mode, res=x, rowscale=s1, rows_per_scale=N,
                           ln=(P[pre + "norm2.weight"], P[pre + "norm2.bias"], None, xn2, mean2, rstd2, hw, 0))
        else:
            ops.linear_fwd(o, W(pre + "attn.proj.weight"), P[pre + "attn.proj.bias"], x1, mode, res=x, rowscale=s1, rows_per_scale=N)
            ops.layernorm_fwd(x1, P[pre + "norm2.weight"], P[pre + "norm2.bias"], None, xn2, mean2, rstd2, hw, 0)
        if half and ops.leff_fusable(hw, C):
            # C <= 128: fc1 -> GELU -> depthwise 3 x 3 -> GELU -> fc2 (+ DropPath, + residual) as ONE launch that keeps an 8 x 8-pixel
            # tile on chip (csrc/leff_fused.hip); h1 / h2 are the two pre-activations, all the backward needs
            h1, h2 = H(dev, M, 4 * C), H(dev, M, 4 * C)
            x2 = _E(dev, M, C)
            ops.leff_fwd(xn2, x1, W(pre + "mlp.linear1.0.weight"), P[pre + "mlp.linear1.0.bias"], P[pre + "mlp.dwconv.0.weight"],
                         P[pre + "mlp.dwconv.0.bias"], W(pre + "mlp.linear2.0.weight"), P[pre + "mlp.linear2.0.bias"], x2, B, hw, C,
                         rowscale=s2, h1=h1, h2=h2)
            return x2, (pre, x, B, C, hw, shift, s1, s2, mod, xn1, mean1, rstd1, qkv, o, x1, xn2, mean2, rstd2, h1, h2, None, lse), None
        h1 = H(dev, M, 4 * C)
        ops.linear_fwd(xn2, W(pre + "mlp.linear1.0.weight"), P[pre + "mlp.linear1.0.bias"], h1, mode)
        a2, h2 = H(dev, M, 4 * C), H(dev, M, 4 * C)       # h2 holds gelu'(dwconv output): all backward needs of it
        ops.dwconv_fwd(h1, P[pre + "mlp.dwconv.0.weight"], P[pre + "mlp.dwconv.0.bias"], None, B, hw, 4 * C, a2=a2, g2=h2)
        x2 = _E(dev, M, C)
        ln, nxt = None, None
        if next_ln is not None and half and ops.ln_fusable(M, C, 4 * C, mode):
            npre, nshift, nmod = next_ln
            nxt = (H(dev, M, C), _E(dev, M), _E(dev, M))
            ln = (P[npre + "norm1.weight"], P[npre + "norm1.bias"], P[npre + "modulator.weight"] if nmod else None,
                  nxt[0], nxt[1], nxt[2], hw, nshift)
        ops.linear_fwd(a2, W(pre + "mlp.linear2.0.weight"), P[pre + "mlp.linear2.0.bias"], x2, mode, res=x1,
                       rowscale=s2, rows_per_scale=N, ln=ln)
        return x2, (pre, x, B, C, hw, shift, s1, s2, mod, xn1, mean1, rstd1, qkv, o, x1, xn2, mean2, rstd2, h1, h2, a2, lse), nxt

    def _block_bwd(self, sv, dy, dyh=None, consumer_scale=None):
        """Returns (dx, dxh): dxh is a bf16 copy of dx in bf16 mode (the next block's GEMM operand), else None.
        dyh, if given, is such a copy of dy that ALREADY carries this block's second DropPath factor (the block that wrote
        it was told so through ``consumer_scale``): the two fc2 products read dy * s2 and take no scale of their own.
        ``consumer_scale``: the s2 of the block that will receive dxh (None: the copy stays unscaled)."""
        (pre, x, B, C, hw, shift, s1, s2, mod, xn1, mean1, rstd1, qkv, o, x1, xn2, mean2, rstd2, h1, h2, a2, lse) = sv
        P, G, dev, mode = self._views, self._gviews, x.device, self._mode()
        M, N = x.shape[0], hw * hw
        # every small-gradient reduction of the block (4 split-K, 2 LN, depthwise, bias table) is deferred into one launch
        (sg0, sg1, sg2, sg3, sl0, sl1, sdw, sat, sab), jobs = self._block_scratch(dev, B, hw, C)
        # LeFF
        H, W = self._H, self._W
        half = self.compute_mode == "bf16"
        dyg = dyh if dyh is not None else dy                       # GEMM operand: the bf16 copy when there is one
        s2g = None if dyh is not None else s2                      # ... which already holds dy * s2
        WT = self._WT
        dh1, dxn2 = H(dev, M, 4 * C), H(dev, M, C)
        if a2 is None:
            # fused LeFF (forward took it): linear2's input gradient, the transposed depthwise convolution and linear1's input
            # gradient in one launch; h2 is the PRE-activation here, so linear2's weight gradient applies GELU while staging it
            ops.leff_bwd(dyg, h1, h2, W(pre + "mlp.linear1.0.weight"), P[pre + "mlp.dwconv.0.weight"], W(pre + "mlp.linear2.0.weight"),
                         dh1, dxn2, G[pre + "mlp.dwconv.0.weight"], G[pre + "mlp.dwconv.0.bias"], B, hw, C, rowscale=s2g,
                         scratch=sdw, defer=jobs.take(2))
            ops.linear_bwd_weight(dyg, h2, G[pre + "mlp.linear2.0.weight"], G[pre + "mlp.linear2.0.bias"], mode, x_op=1,
                                  kscale=s2g, k_per_scale=N, scratch=sg0, defer=jobs.take(1))
        else:
            dh2 = H(dev, M, 4 * C)
            ops.linear_bwd_input(dyg, W(pre + "mlp.linear2.0.weight"), dh2, mode, rowscale=s2g, rows_per_scale=N, gaux=h2, gaux_raw=True,
                                 wT=WT(pre + "mlp.linear2.0.weight"))
            ops.linear_bwd_weight(dyg, a2, G[pre + "mlp.linear2.0.weight"], G[pre + "mlp.linear2.0.bias"], mode,
                                  kscale=s2g, k_per_scale=N, scratch=sg0, defer=jobs.take(1))
            ops.dwconv_bwd(h1, dh2, P[pre + "mlp.dwconv.0.weight"], dh1, G[pre + "mlp.dwconv.0.weight"],
                           G[pre + "mlp.dwconv.0.bias"], B, hw, 4 * C, scratch=sdw, defer=jobs.take(2))
            ops.linear_bwd_input(dh1, W(pre + "mlp.linear1.0.weight"), dxn2, mode, wT=WT(pre + "mlp.linear1.0.weight"))
        ops.linear_bwd_weight(dh1, xn2, G[pre + "mlp.linear1.0.weight"], G[pre + "mlp.linear1.0.bias"], mode, scratch=sg1,
                              defer=jobs.take(1))
        if half and ops.attn_half_fusable(hw, C):
            # C <= 128: norm2' -> projection dX -> window-attention backward -> q | k | v dX -> norm1' as ONE window-local launch
            # (csrc/attn_fused.hip); the two weight-gradient products read the bf16 tensors it writes (dx1h carries s1)
            wqT = self._WqkvT(pre)
            if wqT is None:                                        # (no batched copy this step: make the one operand here)
                wqT = H(dev, C, 3 * C)
                ops.strided3(P[pre + "attn.qkv.w"], wqT, (C, 3 * C, 1), (1, C, 0), (3 * C, 1, 0))
            dx1h, dqkv, dx, dxh = H(dev, M, C), H(dev, M, 3 * C), _E(dev, M, C), H(dev, M, C)
            ops.attn_half_bwd(dxn2, dy, x1, mean2, rstd2, P[pre + "norm2.weight"], x, mean1, rstd1, P[pre + "norm1.weight"], qkv, o, lse,
                              P[pre + "attn.relative_position_bias_table"], W(pre + "attn.proj.weight"), wqT, dx1h, dqkv, dx,
                              G[pre + "norm2.weight"], G[pre + "norm2.bias"], G[pre + "norm1.weight"], G[pre + "norm1.bias"],
                              G[pre + "attn.relative_position_bias_table"], B, hw, C, shift, sab, rowscale=s1,
                              copy_scale=consumer_scale, dxh=dxh, dmod=G[pre + "modulator.weight"] if mod else None,
                              defer=jobs.take(6))
            ops.linear_bwd_weight(dx1h, o, G[pre + "attn.proj.weight"], G[pre + "attn.proj.bias"], mode, scratch=sg2, defer=jobs.take(1))
            ops.linear_bwd_weight(dqkv, xn1, G[pre + "attn.qkv.w"], G[pre + "attn.qkv.b"], mode, scratch=sg3, defer=jobs.take(1))
            ops.reduce_jobs(jobs)
            if self._grad_ready is not None:
                self._grad_ready(*self._owner_range[pre])          # this block's gradients are final
            return dx, dxh
        dx1 = _E(dev, M, C)
        dx1h = H(dev, M, C) if half else None
        # the bf16 copy of dx1 feeds the two proj products only, both of which want dx1 * s1
        ops.layernorm_bwd(dxn2, x1, P[pre + "norm2.weight"], mean2, rstd2, dy, dx1, G[pre + "norm2.weight"],
                          G[pre + "norm2.bias"], None, hw, 0, scratch=sl0, defer=jobs.take(2), dx_bf16=dx1h,
                          copy_scale=s1 if half else None, rows_per_scale=N)
        dx1g = dx1h if half else dx1
        s1g = None if half else s1
        # attention
        do = dxn2                                                  # reuse buffer
        ops.linear_bwd_input(dx1g, W(pre + "attn.proj.weight"), do, mode, rowscale=s1g, rows_per_scale=N, wT=WT(pre + "attn.proj.weight"))
        ops.linear_bwd_weight(dx1g, o, G[pre + "attn.proj.weight"], G[pre + "attn.proj.bias"], mode, kscale=s1g, k_per_scale=N,
                              scratch=sg2, defer=jobs.take(1))
        dqkv = H(dev, M, 3 * C)
        ops.winattn_bwd(qkv, P[pre + "attn.relative_position_bias_table"], do, dqkv,
                        G[pre + "attn.relative_position_bias_table"], B, hw, C, shift, mode, out=o, lse=lse,
                        scratch=sat, defer=jobs.take(1))
        dxn1 = H(dev, M, C)
        ops.linear_bwd_input(dqkv, W(pre + "attn.qkv.w"), dxn1, mode, wT=WT(pre + "attn.qkv.w"))
        ops.linear_bwd_weight(dqkv, xn1, G[pre + "attn.qkv.w"], G[pre + "attn.qkv.b"], mode, scratch=sg3, defer=jobs.take(1))
        dx = _E(dev, M, C)
        dxh = H(dev, M, C) if half else None
        ops.layernorm_bwd(dxn1, x, P[pre + "norm1.weight"], mean1, rstd1, dx1, dx, G[pre + "norm1.weight"],
                          G[pre + "norm1.bias"], G[pre + "modulator.weight"] if mod else None, hw, shift,
                          scratch=sl1, defer=jobs.take(2), dx_bf16=dxh, copy_scale=consumer_scale if half else None,
                          rows_per_scale=N)
        ops.reduce_jobs(jobs)
        if self._grad_ready is not None:
            self._grad_ready(*self._owner_range[pre])              # this block's gradients are final
        return dx, dxh

    @staticmethod
    def _consumer_scale(items, k):
        """s2 of the block that receives the gradient the k-th item of a reversed tape produces (None: no such block)."""
        return items[k + 1][1][7] if k + 1 < len(items) and items[k + 1][0] == "block" else None

    def _down_fwd(self, pre, x, B, hw, C):
        P, dev, mode = self._views, x.device, self._mode()
        Mo = B * (hw // 2) ** 2
        if self.compute_mode == "bf16" and ops.down_direct_ok(hw, C):
            # no im2col image: the input as 2 x 2 pixel blocks on a grid (2 C bytes per input token instead of 8 C), the three
            # products as GEMMs with grid addressing (csrc/conv_grid.hip); the grid image is kept for the weight gradient
            w_fwd, w_dx = ops.conv_weights(P[pre + "weight"], C)
            y = _E(dev, Mo, 2 * C)
            xg = ops.down_direct_fwd(x, w_fwd, P[pre + "bias"], y, B, hw, C)
            return y, (pre, x, B, hw, C, w_dx, xg, True)
        cols = self._H(dev, Mo, 16 * C)
        ops.im2col(x, cols, B, hw, C)
        w2 = self._H(dev, 2 * C, 16 * C)
        ops.strided3(P[pre + "weight"], w2, (2 * C, 16, C), (16 * C, 1, 16), (16 * C, C, 1))
        y = _E(dev, Mo, 2 * C)
        ops.linear_fwd(cols, w2, P[pre + "bias"], y, mode)
        # the im2col image is kept for the weight gradient (4 x the layer input, 8 layers: < 1 % of the 288 GB at B = 96)
        # instead of being rebuilt in the backward pass
        return y, (pre, x, B, hw, C, w2, cols)

    def _down_bwd(self, sv, dy, dyh=None):
        G, dev, mode = self._gviews, dy.device, self._mode()
        dyg = dyh if dyh is not None else dy                       # GEMM operand: the bf16 copy when there is one
        if len(sv) == 8:                                           # direct form (see _down_fwd)
            pre, x, B, hw, C, w_dx, xg, _ = sv
            dx = _E(dev, x.shape[0], C)
            dwf = _Z(dev, 2 * C, 16 * C)
            ops.down_direct_bwd(dyg, xg, w_dx, dx, dwf, G[pre + "bias"], B, hw, C, scratch=self._gemm_scratch(dev))
            ops.conv_dweight(dwf, G[pre + "weight"], C)
            if self._grad_ready is not None:
                self._grad_ready(*self._owner_range[pre])
            return dx
        pre, x, B, hw, C, w2, cols = sv
        Mo = dy.shape[0]
        dcols = self._H(dev, Mo, 16 * C)
        ops.linear_bwd_input(dyg, w2, dcols, mode)
        dx = _E(dev, x.shape[0], C)
        ops.col2im(dcols, dx, B, hw, C)
        dw2 = _Z(dev, 2 * C, 16 * C)
        ops.linear_bwd_weight(dyg, cols, dw2, G[pre + "bias"], mode, scratch=self._gemm_scratch(dev))
        ops.strided3(dw2, G[pre + "weight"], (2 * C, 16, C), (16 * C, C, 1), (16 * C, 1, 16), atomic=True)
        if self._grad_ready is not None:
            self._grad_ready(*self._owner_range[pre])
        return dx

    def _up_fwd(self, pre, x, B, hw, Cin, Co, skip):
        """x [B*hw*hw, Cin] -> concat([convT(x), skip]) [B*4hw*hw, 2Co]"""
        P, dev, mode = self._views, x.device, self._mode()
        M = x.shape[0]
        wr = self._H(dev, 4 * Co, Cin)
        ops.strided3(P[pre + "weight"], wr, (4, Co, Cin), (1, 4, 4 * Co), (Co * Cin, Cin, 1))
        y4 = self._H(dev, M, 4 * Co)
        ops.linear_fwd(x, wr, None, y4, mode)
        cat = _E(dev, 4 * M, 2 * Co)
        ops.pixel_shuffle(y4, P[pre + "bias"], cat, B, hw, Co, 2 * Co, 0)
        ops.copy_cols(skip, Co, 0, cat, 2 * Co, Co, 4 * M, Co)
        return cat, (pre, x, B, hw, Cin, Co, wr)

    def _up_bwd(self, sv, dcat):
        pre, x, B, hw, Cin, Co, wr = sv
        G, dev, mode = self._gviews, x.device, self._mode()
        M = x.shape[0]
        dskip = _E(dev, 4 * M, Co)
        ops.copy_cols(dcat, 2 * Co, Co, dskip, Co, 0, 4 * M, Co)
        dy4 = self._H(dev, M, 4 * Co)
        ops.pixel_unshuffle(dcat, dy4, B, hw, Co, 2 * Co, 0)
        dx = _E(dev, M, Cin)
        ops.linear_bwd_input(dy4, wr, dx, mode)
        dwr, db4 = _Z(dev, 4 * Co, Cin), _Z(dev, 4 * Co)
        ops.linear_bwd_weight(dy4, x, dwr, db4, mode, scratch=self._gemm_scratch(dev))
        ops.strided3(dwr, G[pre + "weight"], (4, Co, Cin), (Co * Cin, Cin, 1), (1, 4, 4 * Co), atomic=True)
        ops.strided3(db4, G[pre + "bias"], (1, 4, Co), (0, Co, 1), (0, 0, 1), atomic=True)
        if self._grad_ready is not None:
            self._grad_ready(*self._owner_range[pre])
        return dx, dskip

    def _trunk_fwd(self, pre, img, B, scales, slot0, extractor):
        """img [B,2,128,128] (extractor: own InputProj) or tokens [B*16384, 32]; returns feats (+conv2 map) and tape."""
        P, dev = self._views, img.device
        tape = []
        if extractor:
            d = ops.conv_desc(B, 2, 128, 128, 32, (3, 3), (1, 1), (1, 1), False, True, "leaky")
            y = _E(dev, B * 16384, 32)
            ops.conv_fwd(d, img, P[pre + "input_proj.proj.0.weight"], P[pre + "input_proj.proj.0.bias"], y)
            tape.append(("inproj", d, img, y))
        else:
            y = img
        feats, slot = [], slot0
        blocks = self.arch.trunk_blocks()
        bi = 0
        for i in range(5):
            C, hw = self.arch.embed << i, self.arch.img >> i
            bname = (lambda jj: f"{pre}encoderlayer_{i}.blocks.{jj}." if i < 4 else f"{pre}conv.blocks.{jj}.")
            ln1 = None
            for j in range(self.arch.depths[i]):
                (_, _, _, _, _, shift, rate) = blocks[bi]
                s1 = scales[slot] if (scales is not None and rate > 0) else None
                s2 = scales[slot + 1] if (scales is not None and rate > 0) else None
                nxt = (bname(j + 1), blocks[bi + 1][5], False) if j + 1 < self.arch.depths[i] else None
                y, sv, ln1 = self._block_fwd(bname(j), y, B, C, hw, shift, s1, s2, False, ln1=ln1, next_ln=nxt)
                tape.append(("block", sv))
                slot += 2
                bi += 1
            feats.append(y)
            if i < 4:
                y, sv = self._down_fwd(f"{pre}dowsample_{i}.conv.0.", y, B, hw, C)
                tape.append(("down", sv, i))
        if extractor:
            d2 = ops.conv_desc(B, 1, 64, 512, 1, (8, 8), (16, 8), (0, 0))
            z = _E(dev, B, 4, 8, 8)
            ops.conv_fwd(d2, feats[4], P[pre + "conv2.weight"], P[pre + "conv2.bias"], z)
            tape.append(("conv2", d2, feats[4], z))
            return z, tape
        return feats, tape

    def _trunk_bwd(self, pre, tape, dfeats, dz=None):
        """Backward through a trunk.  dfeats: gradients of the 5 stage outputs (encoder; entries may be None)
        or None (extractor, where the only consumer is conv2 and ``dz`` is the gradient of its map).
        Returns the gradient w.r.t. the trunk input (image for the extractor, tokens for the encoder)."""
        G, P = self._gviews, self._views
        dy = dfeats[4] if dfeats is not None else None
        dyh = None                                                   # bf16 copy of dy while it comes straight out of a block
        items = list(reversed(tape))
        for k, item in enumerate(items):
            kind = item[0]
            if kind == "conv2":
                _, d2, x, z = item
                dy = _E(x.device, *x.shape)
                ops.conv_bwd(d2, x, P[pre + "conv2.weight"], z, dz, dy, G[pre + "conv2.weight"], G[pre + "conv2.bias"])
            elif kind == "down":
                _, sv, i = item
                dy = self._down_bwd(sv, dy, dyh)
                dyh = None
                if dfeats is not None and dfeats[i] is not None:
                    ops.axpbyz(dy, 1.0, dy, 1.0, dfeats[i], dy.numel())
            elif kind == "block":
                dy, dyh = self._block_bwd(item[1], dy, dyh, self._consumer_scale(items, k))
            elif kind == "inproj":
                _, d, img, y = item
                dimg = _E(img.device, *img.shape)
                ops.conv_bwd(d, img, P[pre + "input_proj.proj.0.weight"], y, dy, dimg,
                             G[pre + "input_proj.proj.0.weight"], G[pre + "input_proj.proj.0.bias"])
                return dimg
        return dy

    def _ae_encode_fwd(self, msg, B):
        P, dev = self._views, msg.device
        d1 = ops.conv_desc(B, 1, 32, 32, 16, (3, 3), (1, 1), (1, 1), act="relu")
        a1 = _E(dev, B, 16, 32, 32)
        ops.conv_fwd(d1, msg, P["encoder_wm.conv1.weight"], P["encoder_wm.conv1.bias"], a1)
        p1 = _E(dev, B, 16, 16, 16)
        ops.maxpool_fwd(a1, p1, B * 16, 32, 32, 2, 2)
        d2 = ops.conv_desc(B, 16, 16, 16, 4, (3, 3), (1, 1), (1, 1), act="relu")
        a2 = _E(dev, B, 4, 16, 16)
        ops.conv_fwd(d2, p1, P["encoder_wm.conv2.weight"], P["encoder_wm.conv2.bias"], a2)
        f0 = _E(dev, B, 4, 8, 8)
        ops.maxpool_fwd(a2, f0, B * 4, 16, 16, 2, 2)
        return f0, (d1, msg, a1, p1, d2, a2)

    def _ae_encode_bwd(self, sv, df0, B):
        d1, msg, a1, p1, d2, a2 = sv
        P, G, dev = self._views, self._gviews, msg.device
        da2 = _E(dev, B, 4, 16, 16)
        ops.maxpool_bwd(a2, df0, da2, B * 4, 16, 16, 2, 2)
        dp1 = _E(dev, B, 16, 16, 16)
        ops.conv_bwd(d2, p1, P["encoder_wm.conv2.weight"], a2, da2, dp1, G["encoder_wm.conv2.weight"], G["encoder_wm.conv2.bias"])
        da1 = _E(dev, B, 16, 32, 32)
        ops.maxpool_bwd(a1, dp1, da1, B * 16, 32, 32, 2, 2)
        ops.conv_bwd(d1, msg, P["encoder_wm.conv1.weight"], a1, da1, None, G["encoder_wm.conv1.weight"], G["encoder_wm.conv1.bias"])

    def _ae_decode_fwd(self, f, B):
        P, dev = self._views, f.device
        d1 = ops.conv_desc(B, 4, 8, 8, 16, (2, 2), (2, 2), (0, 0), act="relu", transposed=True)
        t1 = _E(dev, B, 16, 16, 16)
        ops.conv_fwd(d1, f, P["encoder_wm.t_conv1.weight"], P["encoder_wm.t_conv1.bias"], t1)
        d2 = ops.conv_desc(B, 16, 16, 16, 1, (2, 2), (2, 2), (0, 0), act="sigmoid", transposed=True)
        out = _E(dev, B, 1, 32, 32)
        ops.conv_fwd(d2, t1, P["encoder_wm.t_conv2.weight"], P["encoder_wm.t_conv2.bias"], out)
        return out, (d1, f, t1, d2, out)

    def _ae_decode_bwd(self, sv, dout, B):
        d1, f, t1, d2, out = sv
        P, G, dev = self._views, self._gviews, f.device
        dt1 = _E(dev, B, 16, 16, 16)
        ops.conv_bwd(d2, t1, P["encoder_wm.t_conv2.weight"], out, dout, dt1, G["encoder_wm.t_conv2.weight"], G["encoder_wm.t_conv2.bias"])
        df = _E(dev, B, 4, 8, 8)
        ops.conv_bwd(d1, f, P["encoder_wm.t_conv1.weight"], t1, dt1, df, G["encoder_wm.t_conv1.weight"], G["encoder_wm.t_conv1.bias"])
        return df

    # ---- whole model ---------------------------------------------------------------------------
    def _forward_impl(self, x, msg, train, keep_scales, attack, extract_from, embed_only=False):
        P, dev = self._views, x.device
        B = x.shape[0]
        if tuple(x.shape[1:]) != (2, 128, 128) or tuple(msg.shape) != (B, 1, 32, 32):
            raise ValueError("expected x [B,2,128,128] and message [B,1,32,32] (reference model.py:2384-2389)")
        scales = self._scales_for(B, train, keep_scales, dev)
        self._refresh_weight_copy(transposed=train or torch.is_grad_enabled())
        n_trunk = 2 * len(self.arch.trunk_blocks())
        n_dec = 2 * len(self.arch.dec_blocks())
        S = {}
        f0, S["ae_enc"] = self._ae_encode_fwd(msg, B)
        d_in = ops.conv_desc(B, 2, 128, 128, 32, (3, 3), (1, 1), (1, 1), False, True, "leaky")
        tok = _E(dev, B * 16384, 32)
        ops.conv_fwd(d_in, x, P["input_proj.proj.0.weight"], P["input_proj.proj.0.bias"], tok)
        S["inproj"] = (d_in, x, tok)
        feats, S["enc"] = self._trunk_fwd("encoder.", tok, B, scales, 0, extractor=False)
        wm_pred = None
        if not embed_only:
            pooled = _E(dev, B, 4, 8, 8)
            ops.maxpool_fwd(feats[4], pooled, B, 64, 512, 16, 8)
            fsum = _E(dev, B, 4, 8, 8)
            ops.axpbyz(fsum, 1.0, f0, 1.0, pooled, fsum.numel())
            wm_pred, S["dec_pred"] = self._ae_decode_fwd(fsum, B)
        else:
            wm_pred, _ = self._ae_decode_fwd(f0, B)                 # feature_extract: encoder_wm(message) forward
        cat = _E(dev, B * 64, 1024)
        ops.msg_expand_fwd(f0, cat, B)
        ops.copy_cols(feats[4], 512, 0, cat, 1024, 512, B * 64, 512)
        y, dtape, slot = cat, [], n_trunk
        cin = [1024, 512, 256, 128]
        dblocks = self.arch.dec_blocks()
        bi = 0
        for i in range(4):
            C, hw = self.arch.embed << (4 - i), self.arch.img >> (3 - i)
            y, sv = self._up_fwd(f"decoder.upsample_{i}.deconv.0.", y, B, hw // 2, cin[i], C // 2, feats[3 - i])
            dtape.append(("up", sv, i))
            ln1 = None
            for j in range(self.arch.depths[5 + i]):
                (_, _, _, _, _, shift, rate) = dblocks[bi]
                s1 = scales[slot] if (scales is not None and rate > 0) else None
                s2 = scales[slot + 1] if (scales is not None and rate > 0) else None
                nxt = (f"decoder.decoderlayer_{i}.blocks.{j + 1}.", dblocks[bi + 1][5], self.arch.modulator) \
                    if j + 1 < self.arch.depths[5 + i] else None
                y, sv, ln1 = self._block_fwd(f"decoder.decoderlayer_{i}.blocks.{j}.", y, B, C, hw, shift, s1, s2,
                                             self.arch.modulator, ln1=ln1, next_ln=nxt)
                dtape.append(("block", sv))
                slot += 2
                bi += 1
        S["dec"] = dtape
        d_out = ops.conv_desc(B, 64, 128, 128, 2, (3, 3), (1, 1), (1, 1), True, False, "none")
        noise = _E(dev, B, 2, 128, 128)
        ops.conv_fwd(d_out, y, P["output_proj.proj.0.weight"], P["output_proj.proj.0.bias"], noise)
        S["outproj"] = (d_out, y, noise)
        yw = _E(dev, B, 2, 128, 128)
        ops.axpbyz(yw, 1.0, x, 1.0, noise, yw.numel())
        if embed_only:
            return (yw, wm_pred), None
        wave = _E(dev, B, WAVE_LEN)
        # the waveform round trip at fp32 grade outside bf16 mode (three-term transforms, scalar fp32 low-pass passes): the
        # fp32 / split outputs are held to 1e-4 of the CPU oracle BEHIND the extractor
        precise = self.compute_mode != "bf16"
        terms = 3 if precise else 2
        ops.istft_fwd(yw, wave, B, terms)
        att = attack if (attack is not None and attack.flags) else None
        step_used = None
        if att is not None:
            # the backward replays the AWGN / jitter draws from (seed, step): keep THIS pass's step value (a second training
            # forward before this pass's backward advances the live counter)
            step_used = torch.empty_like(self._step_t)
            ops.step_copy(self._step_t, step_used)
            wave_a, sigma = attack_forward(att, wave, step_used, precise=precise)
        else:
            wave_a, sigma = wave, None
        spec = _E(dev, B, 2, 128, 128)
        ops.stft_fwd(wave_a, spec, B, terms)
        S["stft"] = (att, wave, sigma, step_used)
        c1 = ops.conv_desc(B, 2, 128, 128, 4, (3, 3), (1, 1), (1, 1), act="relu")
        mid = _E(dev, B, 4, 128, 128)
        ops.conv_fwd(c1, spec, P["stft_layer.0.weight"], P["stft_layer.0.bias"], mid)
        c2 = ops.conv_desc(B, 4, 128, 128, 2, (3, 3), (1, 1), (1, 1))
        stft_new = _E(dev, B, 2, 128, 128)
        ops.conv_fwd(c2, mid, P["stft_layer.2.weight"], P["stft_layer.2.bias"], stft_new)
        S["stft_layer"] = (c1, spec, mid, c2, stft_new)
        src = yw if extract_from == "y" else spec
        z, S["ext"] = self._trunk_fwd("decoder_wm.", src, B, scales, n_trunk + n_dec, extractor=True)
        wm, S["dec_wm"] = self._ae_decode_fwd(z, B)
        S["B"], S["extract_from"], S["feats"] = B, extract_from, feats
        return (stft_new, noise, wm_pred, wm), S

    def _backward_impl(self, S, d_stft_new, d_noise, d_wm_pred, d_wm, after_segment=None, grad_ready=None):
        """after_segment(i): segment i of segment_bounds() is complete; grad_ready(lo, hi): the arena run [lo, hi) (a LeWin block
        or a down / up-sampling convolution) is final — the data-parallel exchange starts on it while backward goes on."""
        P, G = self._views, self._gviews
        self._grad_ready = grad_ready
        B = S["B"]
        dev = self._flat.device

        def c(t, shape):
            return _Z(dev, *shape) if t is None else t.contiguous().float()

        d_stft_new, d_noise = c(d_stft_new, (B, 2, 128, 128)), c(d_noise, (B, 2, 128, 128))
        d_wm_pred, d_wm = c(d_wm_pred, (B, 1, 32, 32)), c(d_wm, (B, 1, 32, 32))
        # --- extractor (segment C) ---
        dz = self._ae_decode_bwd(S["dec_wm"], d_wm, B)
        d_src = self._trunk_bwd("decoder_wm.", S["ext"], None, dz)
        # --- stft_layer -> stft -> attack -> istft ---
        c1, spec, mid, c2, stft_new = S["stft_layer"]
        dmid = _E(dev, B, 4, 128, 128)
        ops.conv_bwd(c2, mid, P["stft_layer.2.weight"], stft_new, d_stft_new, dmid, G["stft_layer.2.weight"], G["stft_layer.2.bias"])
        dspec = _E(dev, B, 2, 128, 128)
        ops.conv_bwd(c1, spec, P["stft_layer.0.weight"], mid, dmid, dspec, G["stft_layer.0.weight"], G["stft_layer.0.bias"])
        if after_segment is not None:
            after_segment(0)                                        # extractor + stft_layer gradients are final
        if S["extract_from"] != "y":
            ops.axpbyz(dspec, 1.0, dspec, 1.0, d_src, dspec.numel())
        att, wave, sigma, step_used = S["stft"]
        dwave = _E(dev, B, WAVE_LEN)
        precise = self.compute_mode != "bf16"
        terms = 3 if precise else 2
        ops.stft_bwd(dspec, dwave, B, terms)
        if att is not None:
            dwave = attack_backward(att, wave, dwave, sigma, step_used, precise=precise)
        dy = _E(dev, B, 2, 128, 128)
        ops.istft_bwd(dwave, dy, B, terms)
        if S["extract_from"] == "y":
            ops.axpbyz(dy, 1.0, dy, 1.0, d_src, dy.numel())
        # y = x + noise
        ops.axpbyz(dy, 1.0, dy, 1.0, d_noise, dy.numel())
        # --- output_proj + decoder (segment B) ---
        d_out, ytok, noise = S["outproj"]
        dtok = _E(dev, B * 16384, 64)
        ops.conv_bwd(d_out, ytok, P["output_proj.proj.0.weight"], noise, dy, dtok, G["output_proj.proj.0.weight"], G["output_proj.proj.0.bias"])
        dfe = [None] * 5
        g, gh = dtok, None
        ditems = list(reversed(S["dec"]))
        for k, item in enumerate(ditems):
            if item[0] == "block":
                g, gh = self._block_bwd(item[1], g, gh, self._consumer_scale(ditems, k))
            else:
                _, sv, i = item
                g, dskip = self._up_bwd(sv, g)
                gh = None
                dfe[3 - i] = dskip
        if after_segment is not None:
            after_segment(1)                                        # decoder + output_proj gradients are final
        dcat = g                                                    # [B*64, 1024]
        dconv4 = _E(dev, B * 64, 512)
        ops.copy_cols(dcat, 1024, 512, dconv4, 512, 0, B * 64, 512)
        # --- wm_pred branch ---
        dfsum = self._ae_decode_bwd(S["dec_pred"], d_wm_pred, B)
        ops.maxpool_bwd(S["feats"][4], dfsum, dconv4, B, 64, 512, 16, 8, accumulate=True)
        df0 = dfsum                                                 # d(f0 + pooled)/d f0 = identity
        ops.msg_expand_bwd(dcat, df0, B)
        dfe[4] = dconv4
        # --- encoder (segment A) ---
        dtok0 = self._trunk_bwd("encoder.", S["enc"], dfe)
        d_in, x, tok = S["inproj"]
        ops.conv_bwd(d_in, x, P["input_proj.proj.0.weight"], tok, dtok0, None, G["input_proj.proj.0.weight"], G["input_proj.proj.0.bias"])
        self._ae_encode_bwd(S["ae_enc"], df0, B)
        self._grad_ready = None
        if after_segment is not None:
            after_segment(2)


def get_arch(opt):
    """Factory with the reference's signature (utils/model_utils.py:60-109) for the in-scope arch."""
    arch = opt.arch
    if arch == "Uformer_audio":
        return UformerAudio(img_size=opt.train_ps, embed_dim=32, win_size=8, token_projection="linear", token_mlp="leff",
                            depths=[1, 2, 8, 8, 2, 8, 8, 2, 1], modulator=True, dd_in=2, in_chans=2,
                            audio_scale=opt.audio_scale)
    raise Exception("Arch error!")
