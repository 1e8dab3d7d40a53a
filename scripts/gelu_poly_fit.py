#!/usr/bin/env python3
"""What a shorter GELU polynomial would cost in accuracy (CPU only; numbers for DESIGN 7's LeFF row).

The bf16-mode depthwise kernels evaluate  Phi(x) - 1/2 = z Q(z^2)  and  gelu'(x) - 1/2 = z G(z^2)  on z = clamp(x, -4, 4) with degree-7
polynomials in s = z^2 (csrc/gelu_poly.h); each degree is one packed FMA per channel pair in the two hottest element-wise kernels of the
step.  This script fits Q and G at degrees 4 .. 7 (weighted least squares on Chebyshev nodes in s, refined towards equal ripple by a few
Lawson iterations; Q's last coefficient pinned so that z = 4 gives Phi(4), as the product polynomial is) and prints the maximum errors of
Phi, gelu = x Phi and gelu' over [-6, 6] next to what bf16 storage does to the same values (half an ulp: 2^-9 relative).

  python scripts/gelu_poly_fit.py
"""
import math

import numpy as np


def phi_minus_half(z):
    return 0.5 * np.vectorize(math.erf)(z / math.sqrt(2.0))


def gelu_grad_minus_half(z):
    return phi_minus_half(z) + z * np.exp(-0.5 * z * z) / math.sqrt(2.0 * math.pi)


def lawson_fit(f_over_z, deg, pin=None, iters=60, n=4001):
    """coefficients c[0..deg] of P(s) ~ f(z) / z on s = z^2 in [0, 16], near-minimax in the error of z P(z^2) (weight z)."""
    s = 8.0 * (1.0 - np.cos(np.pi * (np.arange(n) + 0.5) / n))          # Chebyshev nodes on [0, 16]
    z = np.sqrt(s)
    target = f_over_z(z)
    w = np.ones(n)
    V = np.vander(s, deg + 1, increasing=True)
    c = None
    for _ in range(iters):
        A, b = V * (w * z)[:, None], target * w * z
        if pin is not None:                                              # P(16) = pin: eliminate the last coefficient
            s0 = 16.0 ** np.arange(deg + 1)
            A2 = A[:, :-1] - np.outer(A[:, -1] / s0[-1], s0[:-1])
            b2 = b - A[:, -1] * pin / s0[-1]
            c = np.linalg.lstsq(A2, b2, rcond=None)[0]
            c = np.append(c, (pin - s0[:-1] @ c) / s0[-1])
        else:
            c = np.linalg.lstsq(A, b, rcond=None)[0]
        err = np.abs((V @ c - target) * z)
        w = w * (0.5 + err / max(err.max(), 1e-300))
        w /= w.max()
    return c


def main():
    x = np.linspace(-6.0, 6.0, 240001)
    z = np.clip(x, -4.0, 4.0)
    s = z * z
    phi_true = 0.5 + phi_minus_half(x)
    gelu_true = x * phi_true
    grad_true = 0.5 + gelu_grad_minus_half(x)
    # what bf16 storage of the results costs by itself (round to nearest even, 8 significant bits)
    def bf16(v):
        u = np.asarray(v, dtype=np.float32).view(np.uint32).astype(np.uint64)
        u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return u.astype(np.uint32).view(np.float32).astype(np.float64)
    print(f"bf16 rounding alone: max |gelu| error {np.abs(bf16(gelu_true) - gelu_true).max():.2e}, max |gelu'| error {np.abs(bf16(grad_true) - grad_true).max():.2e}")
    print("degree  max |Phi err|  max |gelu err|  max |gelu' err|   packed FMAs per pair (Q + G chains)")
    safe = lambda f: (lambda zz: np.where(zz > 1e-9, f(np.maximum(zz, 1e-9)) / np.maximum(zz, 1e-9), f(np.array([1e-9]))[0] / 1e-9))
    for deg in (7, 6, 5, 4):
        pin = float(phi_minus_half(np.array([4.0]))[0] / 4.0)
        q = lawson_fit(safe(phi_minus_half), deg, pin=pin)
        g = lawson_fit(safe(gelu_grad_minus_half), deg)
        P = lambda c: np.polynomial.polynomial.polyval(s, c)
        phi = 0.5 + z * P(q)
        print(f"{deg:4d}    {np.abs(phi - phi_true).max():.2e}      {np.abs(x * phi - gelu_true).max():.2e}       "
              f"{np.abs(0.5 + z * P(g) - grad_true).max():.2e}         {2 * deg}")


if __name__ == "__main__":
    main()
