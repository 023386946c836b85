// Shared pieces of the "h2" convolution kernels (conv_tap_h2.hip, conv_patch_h2.hip) and of the helper kernels that
// touch h2 activation buffers.
//
// h2 = an fp32 value carried as a PAIR of fp16 numbers, x ~ h + m / 2048 with h = RN16(x), m = RN16((x - h) * 2048):
// 22-23 significant bits (|x - (h + m / 2048)| <= 2^-23 |x| in the normal range, 1.5e-11 absolute below 1.2e-4; fp16
// subnormals are kept, gfx950's MFMA does not flush them).  The bf16x3 scheme (bx3_common.h) carries all 24 bits but
// pays SIX bf16 MFMAs per operand pair; here the product is
//     a * w = ah*wh + (ah*wm + am*wh) / 2048 + O(2^-24 |a w|)            (Ootomo & Yokota's error-corrected scheme)
// THREE v_mfma_f32_16x16x32_f16, with the two correction products in their OWN accumulator: their roundings happen
// 2^-11 below the main sum's, and the main sum sees one third of the additions a six-product chain would give it.
// Range: |x| <= 65504 (fp16); an encoder that meets a larger value (or a NaN) clamps it and raises the model's
// overflow flag — the host side (engine.py) then repeats the call on the full-range bf16x3 kernels.  Weight rows are
// scaled by a per-output-channel power of two on the host (graph.py:pack_conv_weight_h2) so that both planes sit in
// the normal fp16 range whatever the magnitude of the BN-folded weights; the epilogue multiplies by 1 / scale.
//
// Layout in HBM (4 bytes per channel, like fp32): per pixel and 16-channel GROUP 64 bytes = [h of channels 0..15 |
// m of channels 0..15].  The PRODUCER encodes once in its epilogue; consumers fetch ready-made MFMA operands
// (LDS-DMA straight into the operand planes, no VALU in the K loop).
#pragma once
#include "bx3_common.h"
#include "act_fast.h"

namespace padel {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr float kH2Max = 65504.0f;
constexpr float kH2Scale = 2048.0f, kH2InvScale = 1.0f / 2048.0f;
constexpr unsigned kOORh = 0x80000000u;      // out-of-range lane offset that stays out of range under small positive additions

// 3x3 taps are walked COLUMN-major by every h2 kernel and in the packed weights (graph.py:pack_conv_weight_h2): k-step t
// of a channel chunk is tap (ky, kx) = (t % 3, t / 3).  The quad patch kernel (conv_patch_h2q.hip) keeps the input rows of
// one kx in registers across its three ky; one order for all kernels keeps their results bitwise identical.
__host__ __device__ constexpr int h2_tap_ky(int t) { return t % 3; }
__host__ __device__ constexpr int h2_tap_kx(int t) { return t / 3; }

// byte offset, inside a pixel, of the h part of channels [c, c + 4) (c % 4 == 0); the m part sits 32 bytes further
__device__ __forceinline__ long long h2_chan_off(int c) { return (long long)(c >> 4) * 64 + (c & 15) * 2; }

typedef float h2_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2_h16x2 __attribute__((ext_vector_type(2)));

// 4 fp32 -> their h and m parts; `bad` collects "does not fit fp16" (|v| > 65504 or NaN).  Pairs go through the packed
// round-to-nearest conversion (v_cvt_pk_f16_f32): 6 VALU per value
__device__ __forceinline__ void h2_encode4(const f32x4 v, h16x4& h, h16x4& m, bool& bad) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        bad = bad || !(fabsf(v[2 * p]) <= kH2Max) || !(fabsf(v[2 * p + 1]) <= kH2Max);
        const h2_f32x2 x = {__builtin_amdgcn_fmed3f(v[2 * p], -kH2Max, kH2Max), __builtin_amdgcn_fmed3f(v[2 * p + 1], -kH2Max, kH2Max)};
        const h2_h16x2 hh = __builtin_convertvector(x, h2_h16x2);
        const h2_f32x2 rr = {(x[0] - (float)hh[0]) * kH2Scale, (x[1] - (float)hh[1]) * kH2Scale};
        const h2_h16x2 mm = __builtin_convertvector(rr, h2_h16x2);
        h[2 * p] = hh[0]; h[2 * p + 1] = hh[1];
        m[2 * p] = mm[0]; m[2 * p + 1] = mm[1];
    }
}

// SiLU / sigmoid of the epilogues: act_fast.h (shared with the fp16 kernels)
template <int ACT>
__device__ __forceinline__ float h2_act(float x) { return fast_act<ACT>(x); }
__device__ __forceinline__ f32x4 h2_decode4(const h16x4 h, const h16x4 m) {
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaf((float)m[r], kH2InvScale, (float)h[r]);
    return v;
}
__device__ __forceinline__ void h2_raise(unsigned* flag, bool bad) {
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

}  // namespace

// acc = main + cross / 2048 per fragment; then * 1 / row scale + bias, activation, residual, store (h2 pairs, or plain fp32
// for the Detect / Pose head maps and the TrackNet heat map)
template <int MF, int NF, int ACT, bool RES, bool FAST>
__device__ __forceinline__ void h2_epilogue_case(const ConvArgs& a, const f32x4 (&mainacc)[MF][NF], const f32x4 (&cross)[MF][NF],
                                                 const int (&mpix)[MF], int fw, int lq, bool& bad) {
    char* const outb = reinterpret_cast<char*>(a.out);
    const char* const resb = reinterpret_cast<const char*>(a.res);
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int co0 = (fw + j) * 16 + lq * 4;
        f32x4 b, sc;
        if (FAST) {
            b = *reinterpret_cast<const f32x4*>(a.bias + co0);
            sc = *reinterpret_cast<const f32x4*>(a.oscale + co0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { b[r] = a.bias[min(co0 + r, a.n16 * 16 - 1)]; sc[r] = a.oscale[min(co0 + r, a.n16 * 16 - 1)]; }
        }
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int m = mpix[f];
            if (!FAST && m < 0) continue;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = h2_act<ACT>(fmaf(fmaf(cross[f][j][r], kH2InvScale, mainacc[f][j][r]), sc[r], b[r]));
            }
            if (FAST) {
                if (RES) {
                    const char* rp = resb + (long long)m * a.res_cs * 4 + h2_chan_off(a.res_choff + co0);
                    const f32x4 rv = h2_decode4(*reinterpret_cast<const h16x4*>(rp), *reinterpret_cast<const h16x4*>(rp + 32));
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rv[r];
                }
                if (a.out_f32) {
                    *reinterpret_cast<f32x4*>(a.out + (long long)m * a.out_cs + a.out_choff + co0) = v;
                } else {
                    h16x4 hv, mv;
                    h2_encode4(v, hv, mv, bad);
                    char* op = outb + (long long)m * a.out_cs * 4 + h2_chan_off(a.out_choff + co0);
                    *reinterpret_cast<h16x4*>(op) = hv;
                    *reinterpret_cast<h16x4*>(op + 32) = mv;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + r;
                    if (co >= a.cout) continue;
                    float x = v[r];
                    if (RES) {
                        const int rc = a.res_choff + co;
                        const _Float16* rp = reinterpret_cast<const _Float16*>(resb + (long long)m * a.res_cs * 4 + (long long)(rc >> 4) * 64) + (rc & 15);
                        x += fmaf((float)rp[16], kH2InvScale, (float)rp[0]);
                    }
                    if (a.out_f32) {
                        a.out[(long long)m * a.out_cs + a.out_choff + co] = x;
                    } else {
                        bad |= !(fabsf(x) <= kH2Max);
                        const float xc = __builtin_amdgcn_fmed3f(x, -kH2Max, kH2Max);
                        const _Float16 hh = (_Float16)xc;
                        const int oc = a.out_choff + co;
                        _Float16* op = reinterpret_cast<_Float16*>(outb + (long long)m * a.out_cs * 4 + (long long)(oc >> 4) * 64) + (oc & 15);
                        op[0] = hh;
                        op[16] = (_Float16)((xc - (float)hh) * kH2Scale);
                    }
                }
            }
        }
    }
}

// mpix[f] = linear output pixel of this lane's column of pixel fragment f, -1 = outside the tensor
template <int MF, int NF>
__device__ __forceinline__ void h2_epilogue(const ConvArgs& a, const f32x4 (&mainacc)[MF][NF], const f32x4 (&cross)[MF][NF],
                                            const int (&mpix)[MF], int fw, int lq, bool fast) {
    bool bad = false;
#define PADEL_H2_EPI(ACT_)                                                                                        \
    do {                                                                                                          \
        if (a.res) { if (fast) h2_epilogue_case<MF, NF, ACT_, true, true>(a, mainacc, cross, mpix, fw, lq, bad);  \
                     else h2_epilogue_case<MF, NF, ACT_, true, false>(a, mainacc, cross, mpix, fw, lq, bad); }    \
        else       { if (fast) h2_epilogue_case<MF, NF, ACT_, false, true>(a, mainacc, cross, mpix, fw, lq, bad); \
                     else h2_epilogue_case<MF, NF, ACT_, false, false>(a, mainacc, cross, mpix, fw, lq, bad); }   \
    } while (0)
    if (a.act == ACT_SILU) PADEL_H2_EPI(ACT_SILU);
    else if (a.act == ACT_RELU) PADEL_H2_EPI(ACT_RELU);
    else if (a.act == ACT_SIGMOID) PADEL_H2_EPI(ACT_SIGMOID);
    else PADEL_H2_EPI(ACT_NONE);
#undef PADEL_H2_EPI
    if (!a.out_f32) h2_raise(a.ovf_flag, bad);
}

}  // namespace padel
