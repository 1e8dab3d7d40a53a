// Packed GELU / GELU' evaluation shared by the depthwise-convolution kernels (dwconv.hip) and the fused LeFF kernels
// (leff_fused.hip).  Reference: nn.GELU() (erf form) around the depthwise convolution of LeFF, model.py:700-711.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(2))) float f32x2;

// a = x Phi(x), g = Phi(x) + x phi(x) for two values.
// ST = float (parity mode): the Abramowitz-Stegun form of gelu_f in packed arithmetic (|err| <= 1.5e-7).
// ST = __bf16: both results are rounded to bf16 on their way out (2^-9 relative), and the exp / rcp pair is what
// makes these kernels VALU-bound, so Phi(x) - 1/2 = z Q(z^2) and gelu'(x) - 1/2 = z G(z^2) are evaluated as degree-7
// polynomials in z^2 on z = clamp(x, -4, 4) (minimax fits, Q pinned to Phi(4) = 1 so that x >= 4 gives x and x <= -4
// gives 0): |Phi err| <= 3.4e-5, |gelu err| <= 1.4e-4, |gelu' err| <= 2.9e-4 in fp32 Horner form, packed FMAs only.
template <typename ST> struct Gelu;
template <> struct Gelu<float> {
  static __device__ __forceinline__ void pair(f32x2 x, f32x2& a, f32x2& g) {
    f32x2 e, t;
    e.x = __expf(-0.5f * x.x * x.x);
    e.y = __expf(-0.5f * x.y * x.y);
    const f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    const f32x2 den = ax * (0.3275911f * 0.70710678118654752f) + 1.0f;
    t.x = __builtin_amdgcn_rcpf(den.x);                           // v_rcp_f32 (1 ulp), not the IEEE division sequence
    t.y = __builtin_amdgcn_rcpf(den.y);
    f32x2 p = t * 1.061405429f - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const f32x2 q = p * t * e * 0.5f;                           // Phi(-|x|)
    const f32x2 cdf = {x.x >= 0.f ? 1.0f - q.x : q.x, x.y >= 0.f ? 1.0f - q.y : q.y};
    a = x * cdf;
    g = cdf + x * e * 0.39894228040143268f;
  }
};
template <> struct Gelu<__bf16> {
  static __device__ __forceinline__ void pair(f32x2 x, f32x2& a, f32x2& g) {
    const f32x2 z = {__builtin_amdgcn_fmed3f(x.x, -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x.y, -4.0f, 4.0f)};
    const f32x2 s = z * z;
    f32x2 q = s * -1.275896465e-09f + 1.043209509e-07f;
    q = q * s - 3.706793989e-06f;
    q = q * s + 7.615712969e-05f;
    q = q * s - 1.020840068e-03f;
    q = q * s + 9.580739327e-03f;
    q = q * s - 6.606320376e-02f;
    q = q * s + 3.988051945e-01f;
    f32x2 d = s * -1.641966869e-08f + 1.213799991e-06f;
    d = d * s - 3.845945104e-05f;
    d = d * s + 6.876443834e-04f;
    d = d * s - 7.687437958e-03f;
    d = d * s + 5.591482390e-02f;
    d = d * s - 2.620298507e-01f;
    d = d * s + 7.967216468e-01f;
    a = x * (z * q + 0.5f);
    g = z * d + 0.5f;
  }
};
template <typename ST>
__device__ __forceinline__ void gelu_pair(f32x2 x, f32x2& a, f32x2& g) { Gelu<ST>::pair(x, a, g); }

