// K3/K4 "bf16x3" — the fp32 convolutions at fp32 accuracy on the BF16 matrix pipe.
//
// CDNA4 has no TF32 and its fp32-input MFMA runs at 1/16 of the bf16 rate (MI355X_MICROARCH.md), which caps the fp32
// kernels of conv_tap.hip at 157 TFLOP/s.  An fp32 number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8
// significand bits: hi = x truncated to bf16, mid = (x - hi) truncated, lo = x - hi - mid), so
//     a * w = (ah + am + al)(wh + wm + wl) = ah*wh + (ah*wm + am*wh) + (ah*wl + am*wm + al*wh) + O(2^-24 |a w|),
// six bf16 products (each exact in fp32) accumulated in fp32: the three dropped terms are below half an fp32 ulp of
// the product, i.e. below what a single fp32 rounding of the fp32 kernels' FMA chain already costs.  Six
// v_mfma_f32_16x16x32_bf16 cover the K = 32 that takes eight v_mfma_f32_16x16x4_f32: 96 instead of 256 matrix-pipe
// cycles per 16x16x32 block.  tools/bf16x3_study.py (round 1) put the error of this scheme at or below the blocked
// fp32 chain's; tests/test_gpu_conv.py holds it to the same 3e-6 bound against fp64 and the parity suite to the same
// noise-floor criteria as the fp32 MFMA kernels before it may become the default.
//
// Nothing changes in HBM for the activations: they stay fp32 NHWC, and travel global -> LDS through the same
// buffer-addressed LDS-DMA ring (zeros for padded taps by the range check, wave-uniform SGPR offsets).  What changes:
//   * a k-step is 32 channels of one tap: two 64-byte sub-rows per pixel (A0 = channels [0,16), A1 = [16,32) of the
//     chunk), each staged exactly like a k-step of conv_tap.hip (same swizzle, same conflict-free fragment reads);
//   * the SPLIT happens in registers right after the fragment read: 2 AND + 2 SUB per value and 3 v_perm per pair —
//     ~90 VALU per k-step for a 2x3-fragment wave, issued next to 36 MFMAs;
//   * weights are pre-split on the host (graph.py:pack_conv_weight_bx3): per output channel and k-step three 64-byte
//     planes (hi | mid | lo, 32 bf16 each) in the lane order the A operand ends up with;
//   * accumulation keeps the two-level scheme (one partial set per 32-channel chunk x 9 taps, flushed into the main
//     accumulators), products are issued smallest first (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi).
// cin % 32 == 16 (yolov8m's 48-channel layers, n-scale's 16): the 3x3 kernel's tail block pairs TAPS instead of channel
// halves (A0 = the 16 channels at tap 2t, A1 = at tap 2t+1: 5 k-steps instead of 9 half-empty ones); the 1x1 kernel runs
// its last k-step with A1 switched off (out-of-range lane offsets -> zeros) against zero-padded weights.
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

// operands swapped like the other tap kernels' successors: A := weights, so D rows = channels, columns = pixels and a
// lane holds 4 consecutive channels of one pixel in the epilogue
#define PADEL_BX3_COMPUTE(ST_)                                                                                    \
    do {                                                                                                          \
        bf8 ah[MF], am[MF], al[MF];                                                                               \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                          \
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(a_rd + (ST_) * STAGE + f * 256);                     \
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(a_rd + (ST_) * STAGE + BM * 16 + f * 256);           \
            split8(x0, x1, ah[f], am[f], al[f]);                                                                  \
        }                                                                                                         \
        bf8 wh[NF], wm[NF], wl[NF];                                                                               \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) {                                                          \
            wh[j] = __builtin_bit_cast(bf8, *reinterpret_cast<const f32x4*>(b_rd + (ST_) * STAGE + j * 256));     \
            wm[j] = __builtin_bit_cast(bf8, *reinterpret_cast<const f32x4*>(b_rd + (ST_) * STAGE + BN * 16 + j * 256)); \
            wl[j] = __builtin_bit_cast(bf8, *reinterpret_cast<const f32x4*>(b_rd + (ST_) * STAGE + 2 * BN * 16 + j * 256)); \
        }                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            part[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], al[f], part[f][j], 0, 0, 0);              \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            part[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], ah[f], part[f][j], 0, 0, 0);              \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            part[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[j], am[f], part[f][j], 0, 0, 0);              \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            part[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], am[f], part[f][j], 0, 0, 0);              \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            part[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[j], ah[f], part[f][j], 0, 0, 0);              \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            part[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], ah[f], part[f][j], 0, 0, 0);              \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)

#define PADEL_BX3_FLUSH()                                                                                         \
    do {                                                                                                          \
        _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                            \
            _Pragma("unroll") for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; } \
    } while (0)

// requests of one k-step into ring stage SR_: A0 / A1 sub-rows (lane offsets VA_ / VB_, SGPR offsets SA0_ / SA1_),
// three weight planes (SGPR offset SB_ + 64 * plane)
#define PADEL_BX3_DMA(SR_, SA0_, SA1_, SB_, VA0_, VA1_, VB0_, VB1_)                                               \
    do {                                                                                                          \
        const unsigned sa0_ = (SA0_), sa1_ = (SA1_), sb_ = (SB_);                                                 \
        dma3<(SR_) * STAGE_B>((VA0_), rsrcA, sa0_, lds_wave);                                                     \
        if constexpr (AP >= 2) dma3<(SR_) * STAGE_B + RP * 64>((VA1_), rsrcA, sa0_, lds_wave);                    \
        dma3<(SR_) * STAGE_B + BM * 64>((VB0_), rsrcA, sa1_, lds_wave);                                           \
        if constexpr (AP >= 2) dma3<(SR_) * STAGE_B + BM * 64 + RP * 64>((VB1_), rsrcA, sa1_, lds_wave);          \
        PADEL_BX3_DMAB(SR_, 0, sb_);                                                                              \
        PADEL_BX3_DMAB(SR_, 1, sb_ + 64u);                                                                        \
        PADEL_BX3_DMAB(SR_, 2, sb_ + 128u);                                                                       \
    } while (0)
#define PADEL_BX3_DMAB(SR_, PL_, SB_)                                                                             \
    do {                                                                                                          \
        if constexpr (BFULL >= 1) dma3<(SR_) * STAGE_B + 2 * BM * 64 + (PL_) * BN * 64>(voffB[0], rsrcB, (SB_), lds_wave); \
        if constexpr (BFULL >= 2) dma3<(SR_) * STAGE_B + 2 * BM * 64 + (PL_) * BN * 64 + RP * 64>(voffB[1], rsrcB, (SB_), lds_wave); \
        if constexpr (BP > BFULL) { if (b_last) dma3<(SR_) * STAGE_B + 2 * BM * 64 + (PL_) * BN * 64 + BFULL * RP * 64>(voffB[BP - 1], rsrcB, (SB_), lds_wave); } \
    } while (0)

template <int MF, int NF, int ACT, bool RES, bool FAST>
__device__ __forceinline__ void bx3_epilogue_case(const ConvArgs& a, const f32x4 (&acc)[MF][NF], int mw, int fw, int lr, int lq) {
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
            const int m = mw + f * 16 + lr;
            if (!FAST && m >= a.M) continue;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[f][j][r] + b[r];
                if (ACT == ACT_SILU) x = x / (1.0f + expf(-x));
                else if (ACT == ACT_RELU) x = x > 0.0f ? x : 0.0f;
                else if (ACT == ACT_SIGMOID) x = 1.0f / (1.0f + expf(-x));
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

template <int MF, int NF>
__device__ __forceinline__ void bx3_epilogue(const ConvArgs& a, const f32x4 (&acc)[MF][NF], int mw, int fw, int lr, int lq, bool fast) {
#define PADEL_BX3_EPI(ACT_)                                                                                       \
    do {                                                                                                          \
        if (a.res) { if (fast) bx3_epilogue_case<MF, NF, ACT_, true, true>(a, acc, mw, fw, lr, lq);               \
                     else bx3_epilogue_case<MF, NF, ACT_, true, false>(a, acc, mw, fw, lr, lq); }                 \
        else       { if (fast) bx3_epilogue_case<MF, NF, ACT_, false, true>(a, acc, mw, fw, lr, lq);              \
                     else bx3_epilogue_case<MF, NF, ACT_, false, false>(a, acc, mw, fw, lr, lq); }                \
    } while (0)
    if (a.act == ACT_SILU) PADEL_BX3_EPI(ACT_SILU);
    else if (a.act == ACT_RELU) PADEL_BX3_EPI(ACT_RELU);
    else if (a.act == ACT_SIGMOID) PADEL_BX3_EPI(ACT_SIGMOID);
    else PADEL_BX3_EPI(ACT_NONE);
#undef PADEL_BX3_EPI
}

#define PADEL_BX3_GEOMETRY()                                                                                      \
    constexpr int NW = WM * WN;                                                                                   \
    constexpr int RP = NW * 16;                                                                                   \
    constexpr int BM = WM * MF * 16, BN = WN * NF * 16;                                                           \
    constexpr int AP = BM / RP, BP = (BN + RP - 1) / RP, BFULL = BN / RP;                                         \
    constexpr int STAGE = (2 * BM + 3 * BN) * 16;      /* 4-byte words per ring stage: A0 | A1 | Whi | Wmid | Wlo */ \
    constexpr int STAGE_B = STAGE * 4;                                                                            \
    constexpr int NREQ = 2 * AP + 3 * BFULL;           /* requests every wave issues per k-step */                \
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");                                              \
    static_assert(BM % RP == 0 && AP <= 2 && BFULL <= 2, "A in 1-2 full passes, B in at most 2 full + 1 partial"); \
    static_assert(3 * STAGE_B <= 160 * 1024, "ring must fit the LDS");                                            \
    __shared__ __attribute__((aligned(16))) float lds[3 * STAGE];                                                 \
    const int tid = threadIdx.x;                                                                                  \
    const int lane = tid & 63;                                                                                    \
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                    \
    const int lr = lane & 15, lq = lane >> 4;                                                                     \
    const int wm = wave / WN, wn = wave % WN;                                                                     \
    const int nmt = a.n_mtiles, nnt = a.n_ntiles;                                                                 \
    const int bid = blockIdx.x;                                                                                   \
    /* XCD-aware 1-D tile map: XCD x (= bid % 8, how the hardware deals out workgroups) owns a contiguous range of  \
       pixel tiles, and inside an XCD consecutive workgroups are the CHANNEL tiles of one pixel tile — they run    \
       concurrently on that XCD, so the input tile is fetched from HBM once and re-read from its L2 */             \
    const int q = nmt >> 3, r = nmt & 7, xcd = bid & 7, idx = bid >> 3;                                           \
    const int mloc = idx / nnt, nt = idx - mloc * nnt;                                                            \
    if (mloc >= q + (xcd < r ? 1 : 0)) return;       /* grid is padded to 8 x max tiles per XCD */                \
    const int mt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + mloc;                                \
    const int m0 = mt * BM;                                                                                       \
    const int f0 = nt * (WN * NF);                                                                                \
    const int HoWo = a.Ho * a.Wo;                                                                                 \
    const int srow = tid >> 2;                                                                                    \
    const int sc = (tid & 3) ^ ((4 - ((srow >> 2) & 3)) & 3);                                                     \
    const int n0 = fastdiv3(m0, a.howo_magic, a.howo_shift), rem0 = m0 - n0 * HoWo;                               \
    const int oy0 = fastdiv3(rem0, a.wo_magic, a.wo_shift), ox0 = rem0 - oy0 * a.Wo;                              \
    const long long lin0 = ((long long)n0 * a.H + oy0 * a.stride) * a.W + ox0 * a.stride;                         \
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + wave * 1024u);            \
    const int ld_off = lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);                                       \
    const float* const a_rd = lds + (wm * MF * 16) * 16 + ld_off;                                                 \
    const float* const b_rd = lds + 2 * BM * 16 + (wn * NF * 16) * 16 + ld_off;                                   \
    const bool b_last = BP > BFULL && (BFULL * RP + wave * 16 < BN);                                              \
    const int nch = (a.cin + 31) >> 5;                 /* 32-channel chunks (the last one half empty if cin & 16) */ \
    const bool half_tail = (a.cin & 16) != 0;                                                                     \
    f32x4 acc[MF][NF], part[MF][NF];                                                                              \
    _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                                \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

// weight rows: nsteps * 192 bytes each (per k-step hi | mid | lo planes of 32 bf16)
#define PADEL_BX3_WEIGHTS(NSTEPS_)                                                                                \
    const unsigned rowb = (unsigned)(NSTEPS_) * 192u;                                                             \
    unsigned voffB[BP];                                                                                           \
    _Pragma("unroll") for (int p = 0; p < BP; ++p) {                                                              \
        const int rr = srow + RP * p;                                                                             \
        const int frag = min(f0 + (rr >> 4), a.n16 - 1);                                                          \
        voffB[p] = (unsigned)(((frag - f0) * 16 + (rr & 15)) * rowb + sc * 16);                                   \
    }                                                                                                             \
    const i32x4 rsrcB = make_rsrc3(reinterpret_cast<const char*>(a.w3) + (long long)f0 * 16 * rowb);

#define PADEL_BX3_FINISH()                                                                                        \
    const bool fast_ = m0 + BM <= a.M && (f0 + WN * NF) * 16 <= a.cout && (((a.out_choff | a.out_cs) & 3) == 0) && \
                       (!a.res || (((a.res_choff | a.res_cs) & 3) == 0));                                         \
    bx3_epilogue<MF, NF>(a, acc, m0 + wm * MF * 16, f0 + wn * NF, lr, lq, fast_);

// =====================================================================================================  3x3
template <int WM, int WN, int MF, int NF>
__global__ void __launch_bounds__(64 * WM * WN, min_waves3(MF * NF)) conv_bx3_kernel(const ConvArgs a) {
    PADEL_BX3_GEOMETRY()
    unsigned voffA[AP][9];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        int m = m0 + srow + RP * p;
        const bool rv = m < a.M;
        if (!rv) m = m0;
        const int n = fastdiv3(m, a.howo_magic, a.howo_shift);
        const int rem = m - n * HoWo;
        const int oy = fastdiv3(rem, a.wo_magic, a.wo_shift);
        const int ox = rem - oy * a.Wo;
        const long long lin = ((long long)n * a.H + oy * a.stride) * a.W + ox * a.stride;
        const unsigned off = (unsigned)(((lin - lin0) * a.in_cs + sc * 4) * 4);
        bool vy[3], vx[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            vy[d] = rv && (unsigned)(oy * a.stride - 1 + d) < (unsigned)a.H;
            vx[d] = (unsigned)(ox * a.stride - 1 + d) < (unsigned)a.W;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) voffA[p][t] = (vy[t / 3] && vx[t % 3]) ? off : kOOR3;
    }
    const i32x4 rsrcA = make_rsrc3(a.in + ((lin0 - (a.W + 1)) * a.in_cs + a.in_choff));
    unsigned tapoff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tapoff[t] = __builtin_amdgcn_readfirstlane((unsigned)((((t / 3) * a.W + (t % 3)) * a.in_cs) * 4));
    // K walk: nfull 32-channel chunks x 9 taps, then — for cin % 32 == 16 (yolov8m's 48-channel layers, n-scale's 16) —
    // a TAIL block that pairs TAPS instead of channel halves: tail step t covers the last 16 channels at taps 2t (as
    // sub-row A0) and 2t+1 (as A1), 5 steps instead of 9 half-empty ones
    const int nfull = a.cin >> 5;
    PADEL_BX3_WEIGHTS(nfull * 9 + (half_tail ? 5 : 0))

    unsigned s_chunk = 0, s_kb = 0;
#define PADEL_BX3_REQ_FULL(SR_, CH_, KB_, T_)                                                                      \
    PADEL_BX3_DMA(SR_, (CH_) + tapoff[T_], (CH_) + tapoff[T_] + 64u, KB_, voffA[0][T_], voffA[AP - 1][T_], voffA[0][T_], voffA[AP - 1][T_])
#define PADEL_BX3_REQ_TAIL(SR_, CH_, KB_, JT_)                                                                     \
    PADEL_BX3_DMA(SR_, (CH_) + tapoff[2 * (JT_)], (CH_) + tapoff[2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8], KB_,    \
                  voffA[0][2 * (JT_)], voffA[AP - 1][2 * (JT_)],                                                  \
                  2 * (JT_) + 1 < 9 ? voffA[0][2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8] : kOOR3,                    \
                  2 * (JT_) + 1 < 9 ? voffA[AP - 1][2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8] : kOOR3)
    bool nxt_tail = false;       // the block after the current full chunk is the tail block

#define PADEL_BX3_STEP(J)                                                                                         \
    do {                                                                                                          \
        wait_vm3<NREQ>();                                                                                         \
        __builtin_amdgcn_s_barrier();                                                                             \
        if constexpr ((J) + 2 < 9) {                                                                              \
            PADEL_BX3_REQ_FULL(((J) + 2) % 3, s_chunk, s_kb + ((J) + 2) * 192u, (J) + 2 < 9 ? (J) + 2 : 0);      \
        } else {                                                                                                  \
            if (nxt_tail) { PADEL_BX3_REQ_TAIL(((J) + 2) % 3, s_chunk + 128u, s_kb + ((J) + 2) * 192u, ((J) + 2) % 9); } \
            else { PADEL_BX3_REQ_FULL(((J) + 2) % 3, s_chunk + 128u, s_kb + ((J) + 2) * 192u, ((J) + 2) % 9); }   \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_BX3_COMPUTE((J) % 3);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
    // tail step JT (0..4): nothing is requested past step 4, so step 4 drains the queue
#define PADEL_BX3_TSTEP(JT)                                                                                       \
    do {                                                                                                          \
        if constexpr ((JT) == 4) wait_vm3<0>(); else wait_vm3<NREQ>();                                            \
        __builtin_amdgcn_s_barrier();                                                                             \
        if constexpr ((JT) + 2 < 5) { PADEL_BX3_REQ_TAIL(((JT) + 2) % 3, s_chunk, s_kb + ((JT) + 2) * 192u, (JT) + 2 < 5 ? (JT) + 2 : 0); } \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_BX3_COMPUTE((JT) % 3);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

    if (nfull > 0) {
        PADEL_BX3_REQ_FULL(0, 0u, 0u, 0);
        PADEL_BX3_REQ_FULL(1, 0u, 192u, 1);
    } else {
        PADEL_BX3_REQ_TAIL(0, 0u, 0u, 0);
        PADEL_BX3_REQ_TAIL(1, 0u, 192u, 1);
    }
    for (int c = 0; c < nfull; ++c) {
        nxt_tail = half_tail && c == nfull - 1;
        PADEL_BX3_STEP(0); PADEL_BX3_STEP(1); PADEL_BX3_STEP(2); PADEL_BX3_STEP(3); PADEL_BX3_STEP(4);
        PADEL_BX3_STEP(5); PADEL_BX3_STEP(6); PADEL_BX3_STEP(7); PADEL_BX3_STEP(8);
        PADEL_BX3_FLUSH();
        s_chunk += 128u;
        s_kb += 9u * 192u;
    }
    if (half_tail) {
        PADEL_BX3_TSTEP(0); PADEL_BX3_TSTEP(1); PADEL_BX3_TSTEP(2); PADEL_BX3_TSTEP(3); PADEL_BX3_TSTEP(4);
        PADEL_BX3_FLUSH();
    } else {
        wait_vm3<0>();      // the two trailing requests (past the last chunk: slack bytes) must land before LDS is released
    }
    PADEL_BX3_FINISH()
#undef PADEL_BX3_STEP
#undef PADEL_BX3_TSTEP
#undef PADEL_BX3_REQ_FULL
#undef PADEL_BX3_REQ_TAIL
}

// =====================================================================================================  1x1
template <int WM, int WN, int MF, int NF>
__global__ void __launch_bounds__(64 * WM * WN, min_waves3(MF * NF)) conv_bx3_1_kernel(const ConvArgs a) {
    PADEL_BX3_GEOMETRY()
    unsigned voffA[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        int m = m0 + srow + RP * p;
        const bool rv = m < a.M;
        if (!rv) m = m0;
        const int n = fastdiv3(m, a.howo_magic, a.howo_shift);
        const int rem = m - n * HoWo;
        const int oy = fastdiv3(rem, a.wo_magic, a.wo_shift);
        const int ox = rem - oy * a.Wo;
        const long long lin = ((long long)n * a.H + oy * a.stride) * a.W + ox * a.stride;
        voffA[p] = rv ? (unsigned)(((lin - lin0) * a.in_cs + sc * 4) * 4) : kOOR3;
    }
    const i32x4 rsrcA = make_rsrc3(a.in + (lin0 * a.in_cs + a.in_choff));
    PADEL_BX3_WEIGHTS(nch)

    unsigned s_k = 0;                         // index of the first k-step of the current 9-step accumulation block
    // step J of a block: chunk s_k + J; its A1 exists unless it is the half-empty last chunk
#define PADEL_BX3_A1(K_, P_) ((half_tail && (int)(K_) >= nch - 1) ? kOOR3 : voffA[P_])
#define PADEL_BX3_1STEP(J)                                                                                        \
    if ((J) < nb) {                                                                                               \
        wait_vm3<NREQ>();                                                                                         \
        __builtin_amdgcn_s_barrier();                                                                             \
        PADEL_BX3_DMA(((J) + 2) % 3, (s_k + (J) + 2) * 128u, (s_k + (J) + 2) * 128u + 64u, (s_k + (J) + 2) * 192u, \
                      voffA[0], voffA[AP - 1], PADEL_BX3_A1(s_k + (J) + 2, 0), PADEL_BX3_A1(s_k + (J) + 2, AP - 1)); \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_BX3_COMPUTE((J) % 3);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    }
    PADEL_BX3_DMA(0, 0u, 64u, 0u, voffA[0], voffA[AP - 1], PADEL_BX3_A1(0, 0), PADEL_BX3_A1(0, AP - 1));
    PADEL_BX3_DMA(1, 128u, 192u, 192u, voffA[0], voffA[AP - 1], PADEL_BX3_A1(1, 0), PADEL_BX3_A1(1, AP - 1));
    for (int k = 0; k < nch; k += 9) {
        const int nb = min(9, nch - k);
        PADEL_BX3_1STEP(0) PADEL_BX3_1STEP(1) PADEL_BX3_1STEP(2) PADEL_BX3_1STEP(3) PADEL_BX3_1STEP(4)
        PADEL_BX3_1STEP(5) PADEL_BX3_1STEP(6) PADEL_BX3_1STEP(7) PADEL_BX3_1STEP(8)
        PADEL_BX3_FLUSH();
        s_k += 9u;
    }
    wait_vm3<0>();
    PADEL_BX3_FINISH()
#undef PADEL_BX3_1STEP
#undef PADEL_BX3_A1
}

template <int WM, int WN, int MF, int NF>
static hipError_t launch_b3(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    constexpr int BM = WM * MF * 16;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.n_ntiles = (a.n16 + WN * NF - 1) / (WN * NF);
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    if (a.ksize == 3) hipLaunchKernelGGL((conv_bx3_kernel<WM, WN, MF, NF>), grid, dim3(64 * WM * WN), 0, s, a);
    else hipLaunchKernelGGL((conv_bx3_1_kernel<WM, WN, MF, NF>), grid, dim3(64 * WM * WN), 0, s, a);
    return hipGetLastError();
}

// tile ids of the fp32 id space (conv_variant_shape); the ring holds 2 BM + 3 BN rows per stage
hipError_t launch_conv_bx3(const ConvArgs& a, int variant, hipStream_t s) {
    if ((a.ksize != 3 && a.ksize != 1) || (a.cin & 15) || a.cin < 16 || !a.w3) return hipErrorNotSupported;
    switch (variant) {
        case 7: return launch_b3<2, 2, 2, 3>(a, s);    //  64 x  96
        case 6: return launch_b3<2, 2, 2, 4>(a, s);    //  64 x 128
        case 9: return launch_b3<4, 1, 2, 4>(a, s);    // 128 x  64
        case 20: return launch_b3<4, 1, 2, 3>(a, s);   // 128 x  48
        case 11: return launch_b3<4, 1, 2, 2>(a, s);   // 128 x  32
        case 12: return launch_b3<4, 1, 2, 1>(a, s);   // 128 x  16
        case 13: return launch_b3<4, 2, 2, 3>(a, s);   // 128 x  96, 8 waves
        case 14: return launch_b3<4, 2, 2, 4>(a, s);   // 128 x 128, 8 waves
        case 25: return launch_b3<4, 1, 1, 5>(a, s);   //  64 x  80, 4 waves of 16 x 80: the 19-fragment (304-channel) fused pose heads
    }
    return hipErrorNotSupported;
}

// Relative tile speeds measured on MI355X (profiles/conv_bx3_sweep_r2d.txt, ..._r2j.txt): 3x3 — 64x96 and 128x48 (4
// waves, 2 workgroups per CU) lead at 173-189 TFLOP/s on the yolov8m bottlenecks, the 8-wave tiles follow at ~0.9;
// 1x1 — since the channel tiles of a pixel tile run side by side on one XCD (the input is fetched from HBM once) the
// same two tiles lead there too (147-169 on the wide C2f cv2 layers).  The rest is padding waste and the fill of the
// last round of workgroups.
int choose_conv_bx3_variant(int M, int n16, int ksize) {
    struct V { int id, bm, nf; float s3, s1; };
    static const V vs[] = {{7, 64, 6, 1.00f, 0.97f},  {20, 128, 3, 1.00f, 1.00f}, {13, 128, 6, 0.92f, 0.94f}, {14, 128, 8, 0.90f, 0.98f},
                           {25, 64, 5, 0.97f, 0.90f},
                           {11, 128, 2, 0.86f, 0.70f}, {9, 128, 4, 0.70f, 0.80f},  {6, 64, 8, 0.55f, 0.56f},  {12, 128, 1, 0.45f, 0.30f}};
    float best = -1.f;
    int bv = 7;
    for (const V& v : vs) {
        const int ntiles = (n16 + v.nf - 1) / v.nf;
        const long long mtiles = (M + v.bm - 1) / v.bm;
        const float fill = (float)n16 / (float)(ntiles * v.nf) * (float)M / (float)(mtiles * v.bm);
        const long long blocks = mtiles * ntiles;
        const long long per_cu = (blocks + 255) / 256;
        const float occ = (float)blocks / (256.f * (float)per_cu);
        const float sc = (ksize == 3 ? v.s3 : v.s1) * fill * occ;
        if (sc > best) { best = sc; bv = v.id; }
    }
    return bv;
}

}  // namespace padel
