"""Host mirror of the reference's uformerWM/losses.py (imported by audio_uformer_stft.py:26) on device kernels.

Same names, constructor arguments and call signatures as losses.py:8-52: ``tv_loss(x, beta=0.5, reg_coeff=5)``,
``TVLoss(tv_loss_weight=1)(x)``, ``CharbonnierLoss(eps=1e-3)(x, y)``.  Each call is one fused kernel that produces the
loss and its input gradient (``torch.ops.wm.charbonnier_loss`` / ``tv_loss_l2`` / ``tv_loss_pow``, csrc/losses.hip),
differentiable through the registered autograd formula.  GPU tensors only: a CPU tensor raises (no eager fallback).
"""
import torch
import torch.nn as nn

from . import torch_ops  # noqa: F401  (registers torch.ops.wm.*)


def _nchw(x):
    if x.dim() != 4:
        raise ValueError("expected an NCHW tensor (losses.py works on x[:, :, h, w])")
    return x


def tv_loss(x, beta=0.5, reg_coeff=5):
    """losses.py:8-18."""
    return torch.ops.wm.tv_loss_pow(_nchw(x), float(beta), float(reg_coeff))[0]


class TVLoss(nn.Module):
    """losses.py:20-37."""

    def __init__(self, tv_loss_weight=1):
        super().__init__()
        self.tv_loss_weight = tv_loss_weight

    def forward(self, x):
        return torch.ops.wm.tv_loss_l2(_nchw(x), float(self.tv_loss_weight))[0]

    @staticmethod
    def tensor_size(t):
        return t.size()[1] * t.size()[2] * t.size()[3]


class CharbonnierLoss(nn.Module):
    """Charbonnier Loss (L1), losses.py:41-52."""

    def __init__(self, eps=1e-3):
        super().__init__()
        self.eps = eps

    def forward(self, x, y):
        if x.shape != y.shape:
            raise ValueError("CharbonnierLoss: x and y must have the same shape")
        return torch.ops.wm.charbonnier_loss(x, y, float(self.eps))[0]
