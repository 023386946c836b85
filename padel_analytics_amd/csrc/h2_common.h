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
// Round 5 — TWO products where the weights allow it: a checkpoint's conv weights are fp16 numbers (Ultralytics stores
// model.half()); when the host keeps BatchNorm's scale out of them (graph.py: Graph.conv(out_scale=), the scale is folded into
// `oscale` instead) their packed m plane is exactly zero, the op is flagged PA_CONV_W_SINGLE and the kernels' WS instantiations
// drop the wm * ah product with its requests and reads:  a * w = ah*wh + am*wh / 2048.  Bitwise the three-product kernels on the
// same blob; DESIGN.md 3.5.
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

typedef unsigned h2_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned h2_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short h2_u16x2 __attribute__((ext_vector_type(2)));

// The fast epilogue of every h2 kernel.  acc = main + cross / 2048 per fragment; then * 1 / row scale + bias, activation,
// residual, store — as h2 pairs, or as plain fp32 for the Detect / Pose head maps and the TrackNet heat map (F32OUT).
//
// Round 4 rewrite.  The round-3 epilogue ran fragment by fragment: residual load -> s_waitcnt vmcnt(0) -> arithmetic ->
// two 8-byte stores, with 64-bit address arithmetic and a branch on out_f32 per fragment — twelve dependent L2 round trips
// per wave, 12.6 k cycles of a 112 k-cycle workgroup life on the 192 -> 192 layers (profiles/r3j_timeline_h2q.txt) and
// a third of it on the 96 -> 96 ones.  Now
//   * every residual piece, bias and scale vector of the wave is requested up front (one round trip);
//   * a lane moves 16 bytes per fragment instead of 8 + 8: in the MFMA result layout lane (lr, lq) owns channels
//     4 lq .. 4 lq + 3 of pixel lr, i.e. 8 bytes of the h half and 8 bytes of the m half of the pixel's 64-byte group
//     [h x 16 | m x 16].  v_permlane16_swap_b32 (rows of 16 lanes: odd rows of the first operand <-> even rows of the
//     second) on (h dword, m dword) leaves lane rows 0 / 1 / 2 / 3 with h[0..8) / m[0..8) / h[8..16) / m[8..16) of their
//     pixel: one 16-byte store per fragment at byte (lq & 1) * 32 + (lq >> 1) * 16 of the group, half the store
//     instructions for the same bytes (the epilogue was store-issue-bound; cdna_hip_programming.md T21).  The swap is an
//     involution: the residual arrives through one 16-byte load per fragment and the same two swaps;
//   * one 64-bit row pointer per pixel fragment, fragments of a row at immediate offsets (64 bytes apart);
//   * the range check of the encoder runs on the packed h halves (one v_and + one v_pk_max_u16 per two values): the flag
//     goes up when an h part IS the largest fp16 number, i.e. for |x| > 65488 (a shade earlier than |x| > 65504; NaN is
//     clamped to -65504 by v_med3 and flagged too).
// Arithmetic per value is unchanged (same operations in the same order): results are bitwise those of round 3.
// Needs whole fragments inside the tensor, choff % 16 == 0 and cs % 16 == 0 (pairs) or % 4 (fp32 out); everything else takes
// h2_epilogue_slow.
template <int MF, int NF, int ACT, bool RES, bool F32OUT>
__device__ __forceinline__ void h2_epilogue_fast(const ConvArgs& a, const f32x4 (&mainacc)[MF][NF], const f32x4 (&cross)[MF][NF],
                                                 const int (&mpix)[MF], int fw, int lq, bool& bad) {
    static_assert(!(RES && F32OUT), "fp32 head maps have no residual");
    f32x4 b[NF], sc[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int co0 = (fw + j) * 16 + lq * 4;
        b[j] = *reinterpret_cast<const f32x4*>(a.bias + co0);
        sc[j] = *reinterpret_cast<const f32x4*>(a.oscale + co0);
    }
    if constexpr (F32OUT) {
        float* op[MF];
#pragma unroll
        for (int f = 0; f < MF; ++f) op[f] = a.out + (long long)mpix[f] * a.out_cs + (a.out_choff + fw * 16 + lq * 4);
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = h2_act<ACT>(fmaf(fmaf(cross[f][j][r], kH2InvScale, mainacc[f][j][r]), sc[j][r], b[j][r]));
                *reinterpret_cast<f32x4*>(op[f] + j * 16) = v;
            }
    } else {
        const int piece = ((lq & 1) << 5) | ((lq >> 1) << 4);
        char* op[MF];
        h2_u32x4 rr[MF][NF];
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            op[f] = reinterpret_cast<char*>(a.out) + (long long)mpix[f] * a.out_cs * 4 + ((((a.out_choff >> 4) + fw) << 6) + piece);
            if constexpr (RES) {
                const char* rp = reinterpret_cast<const char*>(a.res) + (long long)mpix[f] * a.res_cs * 4 + ((((a.res_choff >> 4) + fw) << 6) + piece);
#pragma unroll
                for (int j = 0; j < NF; ++j) rr[f][j] = *reinterpret_cast<const h2_u32x4*>(rp + j * 64);
            }
        }
        h2_u16x2 top = {0, 0};
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = h2_act<ACT>(fmaf(fmaf(cross[f][j][r], kH2InvScale, mainacc[f][j][r]), sc[j][r], b[j][r]));
                if constexpr (RES) {
                    const h2_u32x2 s0 = __builtin_amdgcn_permlane16_swap(rr[f][j][0], rr[f][j][2], false, false);
                    const h2_u32x2 s1 = __builtin_amdgcn_permlane16_swap(rr[f][j][1], rr[f][j][3], false, false);
                    const h2_u32x2 hd = {s0[0], s1[0]}, md = {s0[1], s1[1]};
                    const f32x4 rv = h2_decode4(__builtin_bit_cast(h16x4, hd), __builtin_bit_cast(h16x4, md));
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rv[r];
                }
                h2_u32x2 hd, md;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const h2_f32x2 x = {__builtin_amdgcn_fmed3f(v[2 * p], -kH2Max, kH2Max), __builtin_amdgcn_fmed3f(v[2 * p + 1], -kH2Max, kH2Max)};
                    const h2_h16x2 hh = __builtin_convertvector(x, h2_h16x2);
                    const h2_f32x2 res = {(x[0] - (float)hh[0]) * kH2Scale, (x[1] - (float)hh[1]) * kH2Scale};
                    const h2_h16x2 mm = __builtin_convertvector(res, h2_h16x2);
                    hd[p] = __builtin_bit_cast(unsigned, hh);
                    md[p] = __builtin_bit_cast(unsigned, mm);
                    top = __builtin_elementwise_max(top, __builtin_bit_cast(h2_u16x2, hd[p] & 0x7FFF7FFFu));
                }
                const h2_u32x2 s0 = __builtin_amdgcn_permlane16_swap(hd[0], md[0], false, false);
                const h2_u32x2 s1 = __builtin_amdgcn_permlane16_swap(hd[1], md[1], false, false);
                const h2_u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                *reinterpret_cast<h2_u32x4*>(op[f] + j * 64) = o;
            }
        bad = bad || top[0] >= 0x7BFFu || top[1] >= 0x7BFFu;
    }
}

// Partial tiles, channel counts that are not whole fragments, unaligned slices: element by element
template <int MF, int NF, int ACT, bool RES>
__device__ __forceinline__ void h2_epilogue_slow(const ConvArgs& a, const f32x4 (&mainacc)[MF][NF], const f32x4 (&cross)[MF][NF],
                                                 const int (&mpix)[MF], int fw, int lq, bool& bad) {
    char* const outb = reinterpret_cast<char*>(a.out);
    const char* const resb = reinterpret_cast<const char*>(a.res);
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int co0 = (fw + j) * 16 + lq * 4;
        f32x4 b, sc;
#pragma unroll
        for (int r = 0; r < 4; ++r) { b[r] = a.bias[min(co0 + r, a.n16 * 16 - 1)]; sc[r] = a.oscale[min(co0 + r, a.n16 * 16 - 1)]; }
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int m = mpix[f];
            if (m < 0) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + r;
                if (co >= a.cout) continue;
                float x = h2_act<ACT>(fmaf(fmaf(cross[f][j][r], kH2InvScale, mainacc[f][j][r]), sc[r], b[r]));
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

// mpix[f] = linear output pixel of this lane's column of pixel fragment f, -1 = outside the tensor
template <int MF, int NF>
__device__ __forceinline__ void h2_epilogue(const ConvArgs& a, const f32x4 (&mainacc)[MF][NF], const f32x4 (&cross)[MF][NF],
                                            const int (&mpix)[MF], int fw, int lq, bool fast) {
    // `fast` from the kernel: whole fragments inside the tensor and the channel matrix, slices 4-aligned; the 16-byte moves of
    // the pair path need whole 16-channel groups on top of that, fp32 head maps have no residual path here
    const bool wide = fast && (a.out_f32 ? !a.res : ((((a.out_choff | a.out_cs) & 15) == 0) && (!a.res || (((a.res_choff | a.res_cs) & 15) == 0))));
    bool bad = false;
#define PADEL_H2_EPI(ACT_)                                                                                        \
    do {                                                                                                          \
        if (wide) {                                                                                               \
            if (a.out_f32) h2_epilogue_fast<MF, NF, ACT_, false, true>(a, mainacc, cross, mpix, fw, lq, bad);     \
            else if (a.res) h2_epilogue_fast<MF, NF, ACT_, true, false>(a, mainacc, cross, mpix, fw, lq, bad);    \
            else h2_epilogue_fast<MF, NF, ACT_, false, false>(a, mainacc, cross, mpix, fw, lq, bad);              \
        } else if (a.res) h2_epilogue_slow<MF, NF, ACT_, true>(a, mainacc, cross, mpix, fw, lq, bad);             \
        else h2_epilogue_slow<MF, NF, ACT_, false>(a, mainacc, cross, mpix, fw, lq, bad);                         \
    } while (0)
    if (a.act == ACT_SILU) PADEL_H2_EPI(ACT_SILU);
    else if (a.act == ACT_RELU) PADEL_H2_EPI(ACT_RELU);
    else if (a.act == ACT_SIGMOID) PADEL_H2_EPI(ACT_SIGMOID);
    else if (a.act == ACT_LEAKY) PADEL_H2_EPI(ACT_LEAKY);
    else PADEL_H2_EPI(ACT_NONE);
#undef PADEL_H2_EPI
    if (!a.out_f32) h2_raise(a.ovf_flag, bad);
}

}  // namespace padel
