"""Loader for the reference model (build container only; /root/reference is absent on the GPU box).

Follows SURVEY.md Appendix A: the reference's ``uformerWM/model.py`` needs three symbols of
``timm.models.layers`` (not installed) and the pre-1.8 real-view ``torch.stft/istft`` API.
Both are provided here in memory; the reference source is imported unchanged from where it lies.
Used ONLY by ``make_golden.py`` to produce the committed fixtures.
"""
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference/uformerWM"


class _DropPath(nn.Module):
    """Standard stochastic depth (per-sample Bernoulli keep, scaled by 1/keep)."""

    # When ``queue`` is a list, keep masks are popped from it in call order instead of being
    # drawn from the global RNG, so a train-mode golden can be replayed by the oracle/product.
    queue = None

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        if _DropPath.queue is not None:
            m = _DropPath.queue.pop(0).to(x.dtype).view((x.shape[0],) + (1,) * (x.ndim - 1))
        else:
            m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * m / keep


def _to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def load_reference_model_module():
    sys.dont_write_bytecode = True
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        models = types.ModuleType("timm.models")
        layers = types.ModuleType("timm.models.layers")
        layers.DropPath, layers.to_2tuple = _DropPath, _to_2tuple
        layers.trunc_normal_ = nn.init.trunc_normal_
        timm.models, models.layers = models, layers
        sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})
    sys.modules.setdefault("torchsummary", types.ModuleType("torchsummary"))

    _stft, _istft = torch.stft, torch.istft

    def stft(x, n_fft, hop_length=None, win_length=None, window=None, **kw):
        rc = kw.pop("return_complex", False)
        if window is None:
            window = torch.ones(win_length or n_fft, dtype=x.dtype, device=x.device)
        out = _stft(x, n_fft, hop_length=hop_length, win_length=win_length, window=window,
                    return_complex=True, **kw)
        return out if rc else torch.view_as_real(out)

    def istft(x, n_fft, hop_length=None, win_length=None, window=None, **kw):
        kw.pop("return_complex", None)
        if not torch.is_complex(x):
            x = torch.view_as_complex(x.contiguous())
        if window is None:
            window = torch.ones(win_length or n_fft, dtype=x.real.dtype, device=x.device)
        return _istft(x, n_fft, hop_length=hop_length, win_length=win_length, window=window, **kw)

    torch.stft, torch.istft = stft, istft
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import model  # noqa: the reference's uformerWM/model.py
    model._DropPathStandIn = _DropPath
    return model
