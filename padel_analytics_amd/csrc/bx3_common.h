// Shared pieces of the bf16x3 convolution kernels (conv_tap_bx3.hip: implicit-GEMM tap kernels; conv_patch_bx3.hip: the
// stride-1 3x3 patch kernel): buffer descriptors, LDS-DMA requests, the exact fp32 -> 3 x bf16 split and the fused
// bias / activation / residual epilogue.
#pragma once
#include "kernels.h"
#include <cmath>
#include <cstdint>

namespace padel {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ i32x4 make_rsrc3(const void* base) {
    const unsigned long long b = (unsigned long long)(uintptr_t)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xFFFFu));
    r[2] = (int)0x80000000u;
    r[3] = 0x00020000;
    return r;
}
constexpr unsigned kOOR3 = 0xFFFFFFF0u;

template <int LDS_IMM>
__device__ __forceinline__ void dma3(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds_wave) {
    asm volatile("s_add_u32 m0, %[lb], %[imm]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[vo], %[rs], %[so] offen lds"
                 :
                 : [lb] "s"(lds_wave), [imm] "n"(LDS_IMM), [vo] "v"(voff), [rs] "s"(rsrc), [so] "s"(soff)
                 : "memory", "scc");
}
template <int N>
__device__ __forceinline__ void wait_vm3() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ int fastdiv3(int n, unsigned magic, unsigned shift) {
    return (int)((__umulhi((unsigned)n, magic) + (unsigned)n) >> shift);
}

// 8 fp32 values (x0 = channels 4q..4q+3 of sub-row 0, x1 = of sub-row 1) -> exact bf16 triples, packed 2 per dword
__device__ __forceinline__ void split8(const f32x4 x0, const f32x4 x1, bf8& hi, bf8& mid, bf8& lo) {
    i32x4 h, m, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float xe = p < 2 ? x0[2 * p] : x1[2 * p - 4], xo = p < 2 ? x0[2 * p + 1] : x1[2 * p - 3];
        const unsigned be = __float_as_uint(xe), bo = __float_as_uint(xo);
        const float re = xe - __uint_as_float(be & 0xFFFF0000u), ro = xo - __uint_as_float(bo & 0xFFFF0000u);
        const unsigned bre = __float_as_uint(re), bro = __float_as_uint(ro);
        const float le = re - __uint_as_float(bre & 0xFFFF0000u), lo_ = ro - __uint_as_float(bro & 0xFFFF0000u);
        h[p] = (int)__builtin_amdgcn_perm(bo, be, 0x07060302u);
        m[p] = (int)__builtin_amdgcn_perm(bro, bre, 0x07060302u);
        l[p] = (int)__builtin_amdgcn_perm(__float_as_uint(lo_), __float_as_uint(le), 0x07060302u);
    }
    hi = __builtin_bit_cast(bf8, h);
    mid = __builtin_bit_cast(bf8, m);
    lo = __builtin_bit_cast(bf8, l);
}

constexpr int min_waves3(int frags) { return frags <= 6 ? 2 : 1; }

}  // namespace

template <int MF, int NF, int ACT, bool RES, bool FAST>
__device__ __forceinline__ void bx3_epilogue_case(const ConvArgs& a, const f32x4 (&acc)[MF][NF], const int (&mpix)[MF], int fw, int lq) {
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int co0 = (fw + j) * 16 + lq * 4;
        f32x4 b;
        if (FAST) b = *reinterpret_cast<const f32x4*>(a.bias + co0);
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r] = a.bias[min(co0 + r, a.n16 * 16 - 1)];
        }
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int m = mpix[f];
            if (!FAST && m < 0) continue;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[f][j][r] + b[r];
                if (ACT == ACT_SILU) x = x / (1.0f + expf(-x));
                else if (ACT == ACT_RELU) x = x > 0.0f ? x : 0.0f;
                else if (ACT == ACT_SIGMOID) x = 1.0f / (1.0f + expf(-x));
                else if (ACT == ACT_LEAKY) x = x >= 0.0f ? x : 0.01f * x;
                v[r] = x;
            }
            if (FAST) {
                if (RES) {
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(a.res + (long long)m * a.res_cs + a.res_choff + co0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rv[r];
                }
                *reinterpret_cast<f32x4*>(a.out + (long long)m * a.out_cs + a.out_choff + co0) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + r;
                    if (co >= a.cout) continue;
                    float x = v[r];
                    if (RES) x += a.res[(long long)m * a.res_cs + a.res_choff + co];
                    a.out[(long long)m * a.out_cs + a.out_choff + co] = x;
                }
            }
        }
    }
}

// mpix[f] = linear output pixel (n * Ho + oy) * Wo + ox of this lane's column of pixel fragment f, -1 = outside the tensor
template <int MF, int NF>
__device__ __forceinline__ void bx3_epilogue(const ConvArgs& a, const f32x4 (&acc)[MF][NF], const int (&mpix)[MF], int fw, int lq, bool fast) {
#define PADEL_BX3_EPI(ACT_)                                                                                       \
    do {                                                                                                          \
        if (a.res) { if (fast) bx3_epilogue_case<MF, NF, ACT_, true, true>(a, acc, mpix, fw, lq);               \
                     else bx3_epilogue_case<MF, NF, ACT_, true, false>(a, acc, mpix, fw, lq); }                 \
        else       { if (fast) bx3_epilogue_case<MF, NF, ACT_, false, true>(a, acc, mpix, fw, lq);              \
                     else bx3_epilogue_case<MF, NF, ACT_, false, false>(a, acc, mpix, fw, lq); }                \
    } while (0)
    if (a.act == ACT_SILU) PADEL_BX3_EPI(ACT_SILU);
    else if (a.act == ACT_RELU) PADEL_BX3_EPI(ACT_RELU);
    else if (a.act == ACT_SIGMOID) PADEL_BX3_EPI(ACT_SIGMOID);
    else if (a.act == ACT_LEAKY) PADEL_BX3_EPI(ACT_LEAKY);
    else PADEL_BX3_EPI(ACT_NONE);
#undef PADEL_BX3_EPI
}

}  // namespace padel
