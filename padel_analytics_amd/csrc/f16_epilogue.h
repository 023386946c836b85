// Epilogue of the fp16 convolution kernels (conv_tap16.hip, conv_patch16.hip): bias, activation, residual, store as fp16
// (saturated: no infinities in HBM) or as fp32 for the Detect / Pose head maps.
//
// Round 4: the same restructuring as the h2 epilogue (h2_common.h).  With ONE MFMA per operand pair these kernels spend as
// long in their epilogues as in their matrix work, and the round-3 epilogues ran fragment by fragment — residual load, wait,
// arithmetic, one 8-byte store, with 64-bit address arithmetic and a branch on out_f32 each time.  Now every residual piece
// and bias vector of the wave is requested up front, a row pointer is computed once per pixel fragment, and lanes move 16
// bytes: lane (lr, lq) owns channels 4 lq .. 4 lq + 3 of pixel lr in every channel fragment, i.e. 8 bytes per fragment; for a
// PAIR of fragments (j, j + 1) v_permlane16_swap_b32 on (fragment j's dword, fragment j + 1's dword) leaves lane rows
// 0 / 1 / 2 / 3 with channels [0, 8) of j / [0, 8) of j + 1 / [8, 16) of j / [8, 16) of j + 1 — one 16-byte store (and one
// 16-byte residual load, un-swapped by the same involution) per two fragments; an odd last fragment keeps 8-byte moves.
// Same arithmetic per value in the same order: results are bitwise those of round 3.
#pragma once
#include "kernels.h"
#include "act_fast.h"

namespace padel {
namespace {

typedef float f16e_f32x4 __attribute__((ext_vector_type(4)));
typedef float f16e_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16e_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16e_h2 __attribute__((ext_vector_type(2)));
typedef unsigned f16e_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned f16e_u32x4 __attribute__((ext_vector_type(4)));

template <int ACT>
__device__ __forceinline__ float f16e_act(float v) { return fast_act<ACT>(v); }

// 4 fp32 -> 4 saturated fp16 (packed round-to-nearest conversion), as two dwords
__device__ __forceinline__ f16e_u32x2 f16e_pack4(const f16e_f32x4 v) {
    f16e_u32x2 o;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f16e_f32x2 x = {__builtin_amdgcn_fmed3f(v[2 * p], -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(v[2 * p + 1], -65504.0f, 65504.0f)};
        o[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16e_h2));
    }
    return o;
}

// whole fragments inside the tensor; fp16 out: choff % 8 == 0 and cs % 8 == 0 (16-byte pieces), fp32 out: % 4, no residual
template <int MF, int NF, int ACT, bool RES, bool F32OUT>
__device__ __forceinline__ void f16_epilogue_fast(const ConvArgs& a, const f16e_f32x4 (&acc)[MF][NF], const int (&mpix)[MF], int fw, int lq) {
    static_assert(!(RES && F32OUT), "fp32 head maps have no residual");
    f16e_f32x4 b[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) b[j] = *reinterpret_cast<const f16e_f32x4*>(a.bias + (fw + j) * 16 + lq * 4);
    if constexpr (F32OUT) {
        float* op[MF];
#pragma unroll
        for (int f = 0; f < MF; ++f) op[f] = a.out + (long long)mpix[f] * a.out_cs + (a.out_choff + fw * 16 + lq * 4);
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                f16e_f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = f16e_act<ACT>(acc[f][j][r] + b[j][r]);
                *reinterpret_cast<f16e_f32x4*>(op[f] + j * 16) = v;
            }
    } else {
        constexpr int NP = NF / 2;                      // fragment pairs; fragment NF - 1 is left over when NF is odd
        // byte offset of the lane's 16-byte piece inside a fragment pair (64 bytes): fragment (lq & 1), channels 8 (lq >> 1) ..
        const int piece = ((lq & 1) << 5) | ((lq >> 1) << 4);
        char* op[MF];
        const char* rp[MF];
        f16e_u32x4 rr[MF][NP > 0 ? NP : 1];
        f16e_u32x2 rl[MF];
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            op[f] = reinterpret_cast<char*>(a.out) + ((long long)mpix[f] * a.out_cs + (a.out_choff + fw * 16)) * 2;
            if constexpr (RES) {
                rp[f] = reinterpret_cast<const char*>(a.res) + ((long long)mpix[f] * a.res_cs + (a.res_choff + fw * 16)) * 2;
#pragma unroll
                for (int p = 0; p < NP; ++p) rr[f][p] = *reinterpret_cast<const f16e_u32x4*>(rp[f] + p * 64 + piece);
                if constexpr (NF & 1) rl[f] = *reinterpret_cast<const f16e_u32x2*>(rp[f] + (NF - 1) * 32 + lq * 8);
            }
        }
        (void)rp; (void)rr; (void)rl;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                f16e_f32x4 v0, v1;
#pragma unroll
                for (int r = 0; r < 4; ++r) { v0[r] = f16e_act<ACT>(acc[f][2 * p][r] + b[2 * p][r]); v1[r] = f16e_act<ACT>(acc[f][2 * p + 1][r] + b[2 * p + 1][r]); }
                if constexpr (RES) {
                    const f16e_u32x2 s0 = __builtin_amdgcn_permlane16_swap(rr[f][p][0], rr[f][p][2], false, false);
                    const f16e_u32x2 s1 = __builtin_amdgcn_permlane16_swap(rr[f][p][1], rr[f][p][3], false, false);
                    const f16e_u32x2 x0 = {s0[0], s1[0]}, x1 = {s0[1], s1[1]};
                    const f16e_h4 r0 = __builtin_bit_cast(f16e_h4, x0), r1 = __builtin_bit_cast(f16e_h4, x1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v0[r] += (float)r0[r]; v1[r] += (float)r1[r]; }
                }
                const f16e_u32x2 o0 = f16e_pack4(v0), o1 = f16e_pack4(v1);
                const f16e_u32x2 s0 = __builtin_amdgcn_permlane16_swap(o0[0], o1[0], false, false);
                const f16e_u32x2 s1 = __builtin_amdgcn_permlane16_swap(o0[1], o1[1], false, false);
                const f16e_u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                *reinterpret_cast<f16e_u32x4*>(op[f] + p * 64 + piece) = o;
            }
        if constexpr (NF & 1) {
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                f16e_f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = f16e_act<ACT>(acc[f][NF - 1][r] + b[NF - 1][r]);
                if constexpr (RES) {
                    const f16e_h4 r0 = __builtin_bit_cast(f16e_h4, rl[f]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)r0[r];
                }
                *reinterpret_cast<f16e_u32x2*>(op[f] + (NF - 1) * 32 + lq * 8) = f16e_pack4(v);
            }
        }
    }
}

// partial tiles, channel counts that are not whole fragments, unaligned slices: element by element
template <int MF, int NF, int ACT, bool RES>
__device__ __forceinline__ void f16_epilogue_slow(const ConvArgs& a, const f16e_f32x4 (&acc)[MF][NF], const int (&mpix)[MF], int fw, int lq) {
    const _Float16* res = reinterpret_cast<const _Float16*>(a.res);
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int co0 = (fw + j) * 16 + lq * 4;
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int m = mpix[f];
            if (m < 0) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + r;
                if (co >= a.cout) continue;
                float x = f16e_act<ACT>(acc[f][j][r] + a.bias[min(co, a.n16 * 16 - 1)]);
                if (RES) x += (float)res[(long long)m * a.res_cs + a.res_choff + co];
                if (a.out_f32) a.out[(long long)m * a.out_cs + a.out_choff + co] = x;
                else reinterpret_cast<_Float16*>(a.out)[(long long)m * a.out_cs + a.out_choff + co] = (_Float16)__builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
            }
        }
    }
}

// mpix[f] = linear output pixel of this lane's column of pixel fragment f, -1 = outside the tensor; `fast` from the kernel:
// every row and channel of the workgroup's tile exists and the slices are 4-channel aligned
template <int MF, int NF>
__device__ __forceinline__ void f16_epilogue(const ConvArgs& a, const f16e_f32x4 (&acc)[MF][NF], const int (&mpix)[MF], int fw, int lq, bool fast) {
    const bool wide = fast && (a.out_f32 ? !a.res : ((((a.out_choff | a.out_cs) & 7) == 0) && (!a.res || (((a.res_choff | a.res_cs) & 7) == 0))));
#define PADEL_F16_EPI(ACT_)                                                                                       \
    do {                                                                                                          \
        if (wide) {                                                                                               \
            if (a.out_f32) f16_epilogue_fast<MF, NF, ACT_, false, true>(a, acc, mpix, fw, lq);                    \
            else if (a.res) f16_epilogue_fast<MF, NF, ACT_, true, false>(a, acc, mpix, fw, lq);                   \
            else f16_epilogue_fast<MF, NF, ACT_, false, false>(a, acc, mpix, fw, lq);                             \
        } else if (a.res) f16_epilogue_slow<MF, NF, ACT_, true>(a, acc, mpix, fw, lq);                            \
        else f16_epilogue_slow<MF, NF, ACT_, false>(a, acc, mpix, fw, lq);                                        \
    } while (0)
    if (a.act == ACT_SILU) PADEL_F16_EPI(ACT_SILU);
    else if (a.act == ACT_RELU) PADEL_F16_EPI(ACT_RELU);
    else if (a.act == ACT_SIGMOID) PADEL_F16_EPI(ACT_SIGMOID);
    else if (a.act == ACT_LEAKY) PADEL_F16_EPI(ACT_LEAKY);
    else PADEL_F16_EPI(ACT_NONE);
#undef PADEL_F16_EPI
}

}  // namespace
}  // namespace padel
