"""PyTorch-ROCm custom ops over the C ABI: ``torch.ops.wm.*`` ("each custom op is torch.library-registered
with an autograd formula, launched on torch.cuda.current_stream(), C-ABI entry points in a .so loaded via
torch.ops.load_library").

The library itself carries no torch types: ``torch.ops.load_library`` maps libwm_hip.so into the process (the same
handle ``ctypes`` binds in ``_lib.py``) and the operators are declared here with ``torch.library.custom_op``.  Every op
is registered for the ``cuda`` device type ONLY: a CPU tensor finds no kernel and raises — there is no eager fallback.
Reference sites replaced (file:line in /root/reference/uformerWM):
  wm::istft255 / wm::stft255     torch.istft / torch.stft(n_fft=255)                  model.py:2458, 2463
  wm::linear                     nn.Linear of LinearProjection / proj / LeFF          model.py:460-471, 549, 700, 711
  wm::layer_norm                 nn.LayerNorm (+ window modulator)                    model.py:982, 996, 1017
  wm::window_attention           WindowAttention core + roll / partition / mask       model.py:523-551, 954-1013
  wm::leff_dwconv                GELU -> depthwise 3x3 of LeFF                        model.py:703-709
  wm::waveform_attack            audio_attack.py:21-30, 55-58, 71-83, 99-125 at the hook model.py:2461-2462
  wm::charbonnier_loss / wm::tv_loss_l2 / wm::tv_loss_pow                             losses.py:41-52, 20-37, 8-18
  wm::watermark_loss             the 4-term loss                                      audio_uformer_stft.py:463-482
The whole-model forward keeps its own autograd boundary (model.py here: one ``autograd.Function`` over an explicit
launch sequence with a hand-written backward); these ops are the same kernels exposed one by one for PyTorch code.
"""
import torch
from torch import Tensor
from torch.library import custom_op

from . import ops
from ._lib import LIB_PATH
from .attacks import AttackConfig, attack_backward, attack_forward

torch.ops.load_library(LIB_PATH)          # dlopen through PyTorch's loader: the .so is part of the op library's lifetime

_DEV = "cuda"


def _f32(t):
    return t.contiguous().float()


# ---------------------------------------------------------------------------------------------------- STFT round trip
@custom_op("wm::istft255", mutates_args=(), device_types=_DEV)
def istft255(spec: Tensor) -> Tensor:
    spec = _f32(spec)
    wave = torch.empty(spec.shape[0], 8002, device=spec.device)
    ops.istft_fwd(spec, wave, spec.shape[0])
    return wave


@custom_op("wm::istft255_backward", mutates_args=(), device_types=_DEV)
def istft255_backward(dwave: Tensor) -> Tensor:
    dwave = _f32(dwave)
    dspec = torch.empty(dwave.shape[0], 2, 128, 128, device=dwave.device)
    ops.istft_bwd(dwave, dspec, dwave.shape[0])
    return dspec


@custom_op("wm::stft255", mutates_args=(), device_types=_DEV)
def stft255(wave: Tensor) -> Tensor:
    wave = _f32(wave)
    spec = torch.empty(wave.shape[0], 2, 128, 128, device=wave.device)
    ops.stft_fwd(wave, spec, wave.shape[0])
    return spec


@custom_op("wm::stft255_backward", mutates_args=(), device_types=_DEV)
def stft255_backward(dspec: Tensor) -> Tensor:
    dspec = _f32(dspec)
    dwave = torch.empty(dspec.shape[0], 8002, device=dspec.device)
    ops.stft_bwd(dspec, dwave, dspec.shape[0])
    return dwave


istft255.register_fake(lambda spec: spec.new_empty(spec.shape[0], 8002, dtype=torch.float32))
istft255_backward.register_fake(lambda dwave: dwave.new_empty(dwave.shape[0], 2, 128, 128, dtype=torch.float32))
stft255.register_fake(lambda wave: wave.new_empty(wave.shape[0], 2, 128, 128, dtype=torch.float32))
stft255_backward.register_fake(lambda dspec: dspec.new_empty(dspec.shape[0], 8002, dtype=torch.float32))
# both transforms are linear: the backward of one is the adjoint kernel, whose own backward is the forward again
istft255.register_autograd(lambda ctx, g: torch.ops.wm.istft255_backward(g))
istft255_backward.register_autograd(lambda ctx, g: torch.ops.wm.istft255(g))
stft255.register_autograd(lambda ctx, g: torch.ops.wm.stft255_backward(g))
stft255_backward.register_autograd(lambda ctx, g: torch.ops.wm.stft255(g))


# ---------------------------------------------------------------------------------------------------- linear
@custom_op("wm::linear", mutates_args=(), device_types=_DEV)
def linear(x: Tensor, weight: Tensor, bias: Tensor, mode: int) -> Tensor:
    """y[M, N] = x[M, K] @ weight[N, K]^T + bias  (mode 0: fp32 MFMA, 1: bf16 MFMA operands / fp32 accumulate)."""
    x, weight, bias = _f32(x), _f32(weight), _f32(bias)
    out = torch.empty(x.shape[0], weight.shape[0], device=x.device)
    ops.linear_fwd(x, weight.bfloat16() if mode else weight, bias, out, mode)
    return out


@custom_op("wm::linear_backward", mutates_args=(), device_types=_DEV)
def linear_backward(dy: Tensor, x: Tensor, weight: Tensor, mode: int) -> tuple[Tensor, Tensor, Tensor]:
    dy, x, weight = _f32(dy), _f32(x), _f32(weight)
    dx = torch.empty_like(x)
    ops.linear_bwd_input(dy, weight.bfloat16() if mode else weight, dx, mode)
    dw, db = torch.zeros_like(weight), torch.zeros(weight.shape[0], device=x.device)
    ops.linear_bwd_weight(dy, x, dw, db, mode)
    return dx, dw, db


linear.register_fake(lambda x, weight, bias, mode: x.new_empty(x.shape[0], weight.shape[0], dtype=torch.float32))
linear_backward.register_fake(lambda dy, x, weight, mode: (torch.empty_like(x, dtype=torch.float32),
                                                          torch.empty_like(weight, dtype=torch.float32),
                                                          weight.new_empty(weight.shape[0], dtype=torch.float32)))


def _linear_setup(ctx, inputs, output):
    x, weight, bias, mode = inputs
    ctx.save_for_backward(x, weight)
    ctx.mode = mode


def _linear_bwd(ctx, g):
    x, weight = ctx.saved_tensors
    dx, dw, db = torch.ops.wm.linear_backward(g, x, weight, ctx.mode)
    return dx, dw, db, None


linear.register_autograd(_linear_bwd, setup_context=_linear_setup)


# ---------------------------------------------------------------------------------------------------- LayerNorm
@custom_op("wm::layer_norm", mutates_args=(), device_types=_DEV)
def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor) -> tuple[Tensor, Tensor, Tensor]:
    x = _f32(x)
    rows, C = x.shape
    y, mean, rstd = torch.empty_like(x), torch.empty(rows, device=x.device), torch.empty(rows, device=x.device)
    ops.layernorm_fwd(x, _f32(gamma), _f32(beta), None, y, mean, rstd, 8, 0)
    return y, mean, rstd


@custom_op("wm::layer_norm_backward", mutates_args=(), device_types=_DEV)
def layer_norm_backward(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor) -> tuple[Tensor, Tensor, Tensor]:
    x = _f32(x)
    dx = torch.empty_like(x)
    dg, db = torch.zeros_like(gamma, dtype=torch.float32), torch.zeros_like(gamma, dtype=torch.float32)
    ops.layernorm_bwd(_f32(dy), x, _f32(gamma), mean, rstd, None, dx, dg, db, None, 8, 0)
    return dx, dg, db


layer_norm.register_fake(lambda x, gamma, beta: (torch.empty_like(x, dtype=torch.float32), x.new_empty(x.shape[0], dtype=torch.float32),
                                                 x.new_empty(x.shape[0], dtype=torch.float32)))
layer_norm_backward.register_fake(lambda dy, x, gamma, mean, rstd: (torch.empty_like(x, dtype=torch.float32),
                                                                    torch.empty_like(gamma, dtype=torch.float32),
                                                                    torch.empty_like(gamma, dtype=torch.float32)))


def _ln_setup(ctx, inputs, output):
    x, gamma, beta = inputs
    ctx.save_for_backward(x, gamma, output[1], output[2])


def _ln_bwd(ctx, gy, gmean, grstd):
    x, gamma, mean, rstd = ctx.saved_tensors
    return torch.ops.wm.layer_norm_backward(gy, x, gamma, mean, rstd)


layer_norm.register_autograd(_ln_bwd, setup_context=_ln_setup)


# ---------------------------------------------------------------------------------------------------- window attention
@custom_op("wm::window_attention", mutates_args=(), device_types=_DEV)
def window_attention(qkv: Tensor, bias_table: Tensor, batch: int, hw: int, shift: int, mode: int) -> Tensor:
    """qkv [batch * hw * hw, 3C] (q | k | v, heads of 32) in natural token order -> attention output [batch * hw * hw, C];
    the cyclic shift, the 8x8 window partition and the shift mask are index arithmetic inside the kernel."""
    qkv = _f32(qkv)
    C = qkv.shape[1] // 3
    out = torch.empty(qkv.shape[0], C, device=qkv.device)
    ops.winattn_fwd(qkv, _f32(bias_table), out, batch, hw, C, shift, mode)
    return out


@custom_op("wm::window_attention_backward", mutates_args=(), device_types=_DEV)
def window_attention_backward(dout: Tensor, qkv: Tensor, bias_table: Tensor, batch: int, hw: int, shift: int,
                              mode: int) -> tuple[Tensor, Tensor]:
    qkv, bias_table = _f32(qkv), _f32(bias_table)
    C = qkv.shape[1] // 3
    dqkv, dtab = torch.empty_like(qkv), torch.zeros_like(bias_table)
    ops.winattn_bwd(qkv, bias_table, _f32(dout), dqkv, dtab, batch, hw, C, shift, mode)
    return dqkv, dtab


window_attention.register_fake(lambda qkv, bias_table, batch, hw, shift, mode: qkv.new_empty(qkv.shape[0], qkv.shape[1] // 3,
                                                                                             dtype=torch.float32))
window_attention_backward.register_fake(lambda dout, qkv, bias_table, batch, hw, shift, mode:
                                        (torch.empty_like(qkv, dtype=torch.float32), torch.empty_like(bias_table, dtype=torch.float32)))


def _wa_setup(ctx, inputs, output):
    qkv, table, batch, hw, shift, mode = inputs
    ctx.save_for_backward(qkv, table)
    ctx.args = (batch, hw, shift, mode)


def _wa_bwd(ctx, g):
    qkv, table = ctx.saved_tensors
    dqkv, dtab = torch.ops.wm.window_attention_backward(g, qkv, table, *ctx.args)
    return dqkv, dtab, None, None, None, None


window_attention.register_autograd(_wa_bwd, setup_context=_wa_setup)


# ---------------------------------------------------------------------------------------------------- LeFF depthwise conv
@custom_op("wm::leff_dwconv", mutates_args=(), device_types=_DEV)
def leff_dwconv(h1: Tensor, weight: Tensor, bias: Tensor, batch: int, hw: int) -> Tensor:
    """h2 = dwconv3x3(gelu(h1)) + bias on token-major [batch, hw, hw, Ch] (model.py:703-707; the second GELU is the
    caller's, or the a_op of the next wm::linear)."""
    h1 = _f32(h1)
    Ch = h1.shape[-1]
    h2 = torch.empty_like(h1)
    ops.dwconv_fwd(h1, _f32(weight), _f32(bias), h2, batch, hw, Ch)
    return h2


@custom_op("wm::leff_dwconv_backward", mutates_args=(), device_types=_DEV)
def leff_dwconv_backward(dh2: Tensor, h1: Tensor, weight: Tensor, batch: int, hw: int) -> tuple[Tensor, Tensor, Tensor]:
    h1, weight = _f32(h1), _f32(weight)
    Ch = h1.shape[-1]
    dh1 = torch.empty_like(h1)
    dw, db = torch.zeros(Ch, 9, device=h1.device), torch.zeros(Ch, device=h1.device)
    ops.dwconv_bwd(h1, _f32(dh2), weight, dh1, dw, db, batch, hw, Ch)
    return dh1, dw.view_as(weight).clone(), db


leff_dwconv.register_fake(lambda h1, weight, bias, batch, hw: torch.empty_like(h1, dtype=torch.float32))
leff_dwconv_backward.register_fake(lambda dh2, h1, weight, batch, hw: (torch.empty_like(h1, dtype=torch.float32),
                                                                      torch.empty_like(weight, dtype=torch.float32),
                                                                      weight.new_empty(weight.shape[0], dtype=torch.float32)))


def _dw_setup(ctx, inputs, output):
    h1, weight, bias, batch, hw = inputs
    ctx.save_for_backward(h1, weight)
    ctx.args = (batch, hw)


def _dw_bwd(ctx, g):
    h1, weight = ctx.saved_tensors
    dh1, dw, db = torch.ops.wm.leff_dwconv_backward(g, h1, weight, *ctx.args)
    return dh1, dw, db, None, None


leff_dwconv.register_autograd(_dw_bwd, setup_context=_dw_setup)


# ---------------------------------------------------------------------------------------------------- waveform attacks
def _cfg(flags, snr_db, amp, seed):
    return AttackConfig(resample=bool(flags & 1), snr_db=snr_db if flags & 2 else None, lowpass=bool(flags & 4),
                        amp=amp if flags & 8 else None, seed=seed)


@custom_op("wm::waveform_attack", mutates_args=(), device_types=_DEV)
def waveform_attack(wave: Tensor, step: Tensor, flags: int, snr_db: float, amp: float, seed: int) -> tuple[Tensor, Tensor]:
    """flags: bit0 resample 16k-8k-16k, bit1 AWGN(snr_db), bit2 Butterworth-8 zero-phase low-pass, bit3 x amp, applied in that
    order on [B, L <= 8002] rows; step: int64[1] device counter keying the Philox noise stream.  Returns (attacked, sigma)."""
    out, sigma = attack_forward(_cfg(flags, snr_db, amp, seed), _f32(wave), step)
    return out, sigma


@custom_op("wm::waveform_attack_backward", mutates_args=(), device_types=_DEV)
def waveform_attack_backward(dout: Tensor, wave: Tensor, sigma: Tensor, step: Tensor, flags: int, snr_db: float, amp: float,
                             seed: int) -> Tensor:
    return attack_backward(_cfg(flags, snr_db, amp, seed), _f32(wave), _f32(dout), sigma, step)


waveform_attack.register_fake(lambda wave, step, flags, snr_db, amp, seed: (torch.empty_like(wave, dtype=torch.float32),
                                                                           wave.new_empty(wave.shape[0], dtype=torch.float32)))
waveform_attack_backward.register_fake(lambda dout, wave, sigma, step, flags, snr_db, amp, seed: torch.empty_like(wave, dtype=torch.float32))


def _att_setup(ctx, inputs, output):
    wave, step, flags, snr_db, amp, seed = inputs
    ctx.save_for_backward(wave, output[1], step)
    ctx.args = (flags, snr_db, amp, seed)


def _att_bwd(ctx, gout, gsigma):
    wave, sigma, step = ctx.saved_tensors
    return torch.ops.wm.waveform_attack_backward(gout, wave, sigma, step, *ctx.args), None, None, None, None, None


waveform_attack.register_autograd(_att_bwd, setup_context=_att_setup)


# ---------------------------------------------------------------------------------------------------- losses
@custom_op("wm::charbonnier_loss", mutates_args=(), device_types=_DEV)
def charbonnier_loss(x: Tensor, y: Tensor, eps: float) -> tuple[Tensor, Tensor]:
    """(loss, d loss / d x): mean(sqrt((x - y)^2 + eps^2)) (losses.py:41-52); d loss / d y = -d loss / d x."""
    x, y = _f32(x), _f32(y)
    loss, dx = torch.zeros((), device=x.device), torch.empty_like(x)
    ops.charbonnier_fwd_bwd(x, y, eps, loss, dx, None, 1.0)
    return loss, dx


@custom_op("wm::tv_loss_l2", mutates_args=(), device_types=_DEV)
def tv_loss_l2(x: Tensor, weight: float) -> tuple[Tensor, Tensor]:
    """TVLoss.forward (losses.py:20-37) on NCHW: (loss, d loss / d x)."""
    x = _f32(x)
    loss, dx = torch.zeros((), device=x.device), torch.empty_like(x)
    ops.tvloss_fwd_bwd(x, weight, loss, dx, 1.0)
    return loss, dx


@custom_op("wm::tv_loss_pow", mutates_args=(), device_types=_DEV)
def tv_loss_pow(x: Tensor, beta: float, reg_coeff: float) -> tuple[Tensor, Tensor]:
    """tv_loss (losses.py:8-18) on NCHW: (loss, d loss / d x)."""
    x = _f32(x)
    loss, dx = torch.zeros((), device=x.device), torch.empty_like(x)
    ops.tv_pow_fwd_bwd(x, beta, reg_coeff, loss, dx, 1.0)
    return loss, dx


charbonnier_loss.register_fake(lambda x, y, eps: (x.new_empty((), dtype=torch.float32), torch.empty_like(x, dtype=torch.float32)))
tv_loss_l2.register_fake(lambda x, weight: (x.new_empty((), dtype=torch.float32), torch.empty_like(x, dtype=torch.float32)))
tv_loss_pow.register_fake(lambda x, beta, reg_coeff: (x.new_empty((), dtype=torch.float32), torch.empty_like(x, dtype=torch.float32)))


def _save_dx(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.n_in = sum(isinstance(i, Tensor) for i in inputs)
    ctx.n_all = len(inputs)


def _loss_bwd(ctx, gloss, gdx):
    (dx,) = ctx.saved_tensors
    g = gloss * dx                                   # first-order only: the stored gradient is treated as a constant
    grads = [g] + ([-g] if ctx.n_in == 2 else [])
    return tuple(grads + [None] * (ctx.n_all - len(grads)))


charbonnier_loss.register_autograd(_loss_bwd, setup_context=_save_dx)
tv_loss_l2.register_autograd(_loss_bwd, setup_context=_save_dx)
tv_loss_pow.register_autograd(_loss_bwd, setup_context=_save_dx)


@custom_op("wm::watermark_loss", mutates_args=(), device_types=_DEV)
def watermark_loss(stft_new: Tensor, x: Tensor, noise: Tensor, wm_pred: Tensor, wm: Tensor,
                   message: Tensor) -> tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """The trainer's loss (audio_uformer_stft.py:463-482): MSE(stft_new, x) + MSE(wm_pred, msg) + MSE(wm, msg) +
    (||noise||_2 / B - 1)^2.  Returns (losses[8]: l1..l4, total, ||noise||; d/d stft_new, d/d noise, d/d wm_pred, d/d wm)."""
    a = [_f32(t) for t in (stft_new, x, noise, wm_pred, wm, message)]
    B = a[0].shape[0]
    losses, scratch = torch.zeros(8, device=a[0].device), torch.zeros(8, device=a[0].device)
    d = [torch.empty_like(a[0]), torch.empty_like(a[2]), torch.empty_like(a[3]), torch.empty_like(a[4])]
    ops.loss_fwd_bwd(a[0], a[1], a[2], a[3], a[4], a[5], losses, d[0], d[1], d[2], d[3], B, scratch)
    return losses, d[0], d[1], d[2], d[3]


watermark_loss.register_fake(lambda stft_new, x, noise, wm_pred, wm, message:
                             (stft_new.new_empty(8, dtype=torch.float32), torch.empty_like(stft_new, dtype=torch.float32),
                              torch.empty_like(noise, dtype=torch.float32), torch.empty_like(wm_pred, dtype=torch.float32),
                              torch.empty_like(wm, dtype=torch.float32)))


def _wl_setup(ctx, inputs, output):
    ctx.save_for_backward(*output[1:])


def _wl_bwd(ctx, glosses, *_):
    d0, d1, d2, d3 = ctx.saved_tensors
    g = glosses[4]                                   # gradient flows through the total (losses[4])
    return g * d0, None, g * d1, g * d2, g * d3, None


watermark_loss.register_autograd(_wl_bwd, setup_context=_wl_setup)

OPS = ("istft255", "istft255_backward", "stft255", "stft255_backward", "linear", "linear_backward", "layer_norm",
       "layer_norm_backward", "window_attention", "window_attention_backward", "leff_dwconv", "leff_dwconv_backward",
       "waveform_attack", "waveform_attack_backward", "charbonnier_loss", "tv_loss_l2", "tv_loss_pow", "watermark_loss")
