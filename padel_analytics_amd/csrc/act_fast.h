// Fast SiLU / sigmoid for the convolution epilogues of the h2 and fp16 kernels.
#pragma once
#include "kernels.h"

namespace padel {
namespace {

// SiLU / sigmoid for the epilogues: e^-x through v_exp_f32 on a compensated argument (the product x * log2(e) carried as
// hi + lo), the quotient through v_rcp_f32 + one Newton step on the remainder.  11 VALU instead of the 28 of
// expf() + IEEE division, at the accuracy of the fp32 formula itself (max 2.7 ulp / mean 0.37 ulp against 2.4 / 0.35 for
// correctly rounded exp + division, measured over 2.5 M arguments in [-90, 90]; profiles/h2_silu_accuracy_r3.txt).
__device__ __forceinline__ float fast_exp_neg(float x) {           // e^-x, finite for every finite x (clamped at 2^126)
    const float t = -x * 1.4426950216293335f;
    float tl = fmaf(-x, 1.4426950216293335f, -t);
    tl = fmaf(-x, 1.9259629911783190e-8f, tl);
    const float e0 = __builtin_amdgcn_exp2f(fminf(t, 126.0f));
    return fmaf(e0, tl * 0.6931471805599453f, e0);
}
__device__ __forceinline__ float fast_div(float num, float d) {     // num / d for d in [1, 2^127)
    const float r = __builtin_amdgcn_rcpf(d);
    const float y = num * r;
    return fmaf(fmaf(-y, d, num), r, y);
}
template <int ACT>
__device__ __forceinline__ float fast_act(float x) {
    if (ACT == ACT_SILU) return fast_div(x, 1.0f + fast_exp_neg(x));
    if (ACT == ACT_RELU) return x > 0.0f ? x : 0.0f;
    if (ACT == ACT_SIGMOID) return fast_div(1.0f, 1.0f + fast_exp_neg(x));
    if (ACT == ACT_LEAKY) return x >= 0.0f ? x : 0.01f * x;
    return x;
}

}  // namespace
}  // namespace padel
