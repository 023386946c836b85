// The implicit-GEMM TAP machine of the h2 kernels (conv_tap_h2.hip: 2-stage ring; conv_tap_h2p.hip: 3-stage activation ring):
// geometry, ring requests, the k-step and the epilogue call as macros shared by both files.  Every kernel that uses them has a
// template parameter `bool WS` (the packed weights' m plane is all zero: ConvArgs::w_single — conv_patch_h2q.hip).
#pragma once
#include "h2_common.h"

#define PADEL_H2T_AR(ST_) (((ST_) & 1) ? a_rd1 : a_rd0)
#define PADEL_H2T_BR(ST_) (((ST_) & 1) ? b_rd1 : b_rd0)
#define PADEL_H2T_LW(SR_) (((SR_) & 1) ? lw1 : lw0)
#define PADEL_H2T_COMPUTE(ST_, FIRST_) PADEL_H2T_COMPUTE_AT(PADEL_H2T_AR(ST_), PADEL_H2T_BR(ST_), FIRST_)
// one k-step from the activation sub-rows at AR_ (h plane; m plane BM rows further) and the weight planes at BR_
#define PADEL_H2T_COMPUTE_AT(AR_, BR_, FIRST_)                                                                    \
    do {                                                                                                          \
        h16x8 ah[MF], am[MF], wh[NF], wm[NF];                                                                     \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                          \
            ah[f] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>((AR_) + f * 256));                  \
            am[f] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>((AR_) + BM * 16 + f * 256));        \
        }                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) {                                                          \
            wh[j] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>((BR_) + j * 256));                  \
            if constexpr (!WS) wm[j] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>((BR_) + BN * 16 + j * 256)); \
        }                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            cross[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], am[f], cross[f][j], 0, 0, 0);             \
        if constexpr (!WS) {                /* WS: the weights' m plane is all zero (ConvArgs::w_single): product skipped */ \
            _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)         \
                cross[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[j], ah[f], cross[f][j], 0, 0, 0);         \
        }                                                                                                         \
        if constexpr (FIRST_) {                /* first step of an accumulation block: the main chain starts from the constant 0 */ \
            _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)         \
                part[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], ah[f], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0); \
        } else {                                                                                                  \
            _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)         \
                part[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], ah[f], part[f][j], 0, 0, 0);           \
        }                                                                                                         \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)

#define PADEL_H2T_FLUSH()                                                                                         \
    do {                                                                                                          \
        _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                            \
            _Pragma("unroll") for (int j = 0; j < NF; ++j) acc[f][j] += part[f][j];   /* (part restarts from 0 inside the next block's first MFMAs) */ \
    } while (0)
#define PADEL_H2T_SWAP()                                                                                          \
    do { const float* t_ = a_rd0; a_rd0 = a_rd1; a_rd1 = t_; t_ = b_rd0; b_rd0 = b_rd1; b_rd1 = t_;                \
         const unsigned u_ = lw0; lw0 = lw1; lw1 = u_; } while (0)

// requests of one k-step into ring stage SR_: the h / m sub-rows of the activation tile (lane offsets V0_ / V1_ for
// the 1-2 row passes, SGPR offset SA_ for the h plane, SA_ + 32 for the m plane), the two weight planes (SB_, SB_ + 64)
#define PADEL_H2T_DMA_R(rsrcA, SR_, SA_, SB_, V0_, V1_)                                                           \
    do {                                                                                                          \
        const unsigned sa_ = (SA_), sb_ = (SB_);                                                                  \
        dma3<0>((V0_), rsrcA, sa_, PADEL_H2T_LW(SR_));                                                            \
        if constexpr (AP >= 2) dma3<RP * 64>((V1_), rsrcA, sa_, PADEL_H2T_LW(SR_));                               \
        dma3<BM * 64>((V0_), rsrcA, sa_ + 32u, PADEL_H2T_LW(SR_));                                                \
        if constexpr (AP >= 2) dma3<BM * 64 + RP * 64>((V1_), rsrcA, sa_ + 32u, PADEL_H2T_LW(SR_));               \
        PADEL_H2T_DMAB(SR_, 0, sb_);                                                                              \
        if constexpr (!WS) PADEL_H2T_DMAB(SR_, 1, sb_ + 64u);                                                     \
    } while (0)
#define PADEL_H2T_DMAB(SR_, PL_, SB_)                                                                             \
    do {                                                                                                          \
        if constexpr (BFULL >= 1) dma3<2 * BM * 64 + (PL_) * BN * 64>(voffB[0], rsrcB, (SB_), PADEL_H2T_LW(SR_)); \
        if constexpr (BFULL >= 2) dma3<2 * BM * 64 + (PL_) * BN * 64 + RP * 64>(voffB[1], rsrcB, (SB_), PADEL_H2T_LW(SR_)); \
        if constexpr (BP > BFULL) { if (b_last) dma3<2 * BM * 64 + (PL_) * BN * 64 + BFULL * RP * 64>(voffB[BP - 1], rsrcB, (SB_), PADEL_H2T_LW(SR_)); } \
    } while (0)

#define PADEL_H2T_GEOMETRY() PADEL_H2T_GEOMETRY_(2 * STAGE)
#define PADEL_H2T_GEOMETRY_(LDSW_)                                                                               \
    constexpr int NW = WM * WN;                                                                                   \
    constexpr int RP = NW * 16;                                                                                   \
    constexpr int BM = WM * MF * 16, BN = WN * NF * 16;                                                           \
    constexpr int AP = BM / RP, BP = (BN + RP - 1) / RP, BFULL = BN / RP;                                         \
    constexpr int STAGE = (2 * BM + 2 * BN) * 16;      /* 4-byte words per ring stage: Ah | Am | Wh | Wm */        \
    constexpr int STAGE_B = STAGE * 4;                                                                            \
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");                                              \
    static_assert(BM % RP == 0 && AP <= 2 && BFULL <= 2, "A in 1-2 full passes, B in at most 2 full + 1 partial"); \
    static_assert((LDSW_) * 4 <= 160 * 1024, "ring must fit the LDS");                                            \
    __shared__ __attribute__((aligned(16))) float lds[LDSW_];                                                     \
    const int tid = threadIdx.x;                                                                                  \
    const int lane = tid & 63;                                                                                    \
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                    \
    const int lr = lane & 15, lq = lane >> 4;                                                                     \
    const int wm_ = wave / WN, wn_ = wave % WN;                                                                   \
    const int nmt = a.n_mtiles, nnt = a.n_ntiles;                                                                 \
    const int bid = blockIdx.x;                                                                                   \
    /* XCD-aware 1-D tile map (conv_tap_bx3.hip): the channel tiles of one pixel tile are neighbours on one XCD */ \
    const int q = nmt >> 3, r = nmt & 7, xcd = bid & 7, idx = bid >> 3;                                           \
    const int mloc = idx / nnt, nt = idx - mloc * nnt;                                                            \
    if (mloc >= q + (xcd < r ? 1 : 0)) return;                                                                    \
    const int mt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + mloc;                                \
    const int m0 = mt * BM;                                                                                       \
    const int f0 = nt * (WN * NF);                                                                                \
    const int HoWo = a.Ho * a.Wo;                                                                                 \
    const int srow = tid >> 2;                                                                                    \
    const int sc = (tid & 3) ^ ((4 - ((srow >> 2) & 3)) & 3);      /* logical 16-byte slot this lane fetches */      \
    const bool sc_hi = (sc >> 1) != 0;                                                                            \
    const unsigned slot_b = (unsigned)((sc >> 1) * 64 + (sc & 1) * 16);   /* its place in a 128-byte h2 chunk (h plane) */ \
    const int n0 = fastdiv3(m0, a.howo_magic, a.howo_shift), rem0 = m0 - n0 * HoWo;                               \
    const int oy0 = fastdiv3(rem0, a.wo_magic, a.wo_shift), ox0 = rem0 - oy0 * a.Wo;                              \
    const long long lin0 = ((long long)n0 * a.H + oy0 * a.stride) * a.W + ox0 * a.stride;                         \
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + wave * 1024u);            \
    const int ld_off = lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);                                       \
    const float *a_rd0 = lds + (wm_ * MF * 16) * 16 + ld_off, *a_rd1 = a_rd0 + STAGE;                             \
    const float *b_rd0 = lds + 2 * BM * 16 + (wn_ * NF * 16) * 16 + ld_off, *b_rd1 = b_rd0 + STAGE;               \
    unsigned lw0 = lds_wave, lw1 = __builtin_amdgcn_readfirstlane(lds_wave + (unsigned)STAGE_B);                  \
    const bool b_last = BP > BFULL && (BFULL * RP + wave * 16 < BN);                                              \
    const int nch = (a.cin + 31) >> 5;                 /* 32-channel chunks (the last one half empty if cin & 16) */ \
    const bool half_tail = (a.cin & 16) != 0;                                                                     \
    (void)sc_hi; (void)nch;                                                                                       \
    f32x4 acc[MF][NF], part[MF][NF], cross[MF][NF];                                                               \
    _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                                \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; cross[f][j] = acc[f][j]; }

// weight rows: nsteps * 128 bytes each (per k-step h | m planes of 32 fp16)
#define PADEL_H2T_WEIGHTS(NSTEPS_)                                                                                \
    const unsigned rowb = (unsigned)(NSTEPS_) * 128u;                                                             \
    unsigned voffB[BP];                                                                                           \
    _Pragma("unroll") for (int p = 0; p < BP; ++p) {                                                              \
        const int rr = srow + RP * p;                                                                             \
        const int frag = min(f0 + (rr >> 4), a.n16 - 1);                                                          \
        voffB[p] = (unsigned)(((frag - f0) * 16 + (rr & 15)) * rowb + sc * 16);                                   \
    }                                                                                                             \
    const i32x4 rsrcB = make_rsrc3(reinterpret_cast<const char*>(a.w) + (long long)f0 * 16 * rowb);

#define PADEL_H2T_FINISH()                                                                                        \
    const bool fast_ = m0 + BM <= a.M && (f0 + WN * NF) * 16 <= a.cout && (((a.out_choff | a.out_cs) & 3) == 0) && \
                       (!a.res || (((a.res_choff | a.res_cs) & 3) == 0));                                         \
    int mpix_[MF];                                                                                                \
    _Pragma("unroll") for (int f = 0; f < MF; ++f) { const int m_ = m0 + wm_ * MF * 16 + f * 16 + lr; mpix_[f] = m_ < a.M ? m_ : -1; } \
    h2_epilogue<MF, NF>(a, acc, cross, mpix_, f0 + wn_ * NF, lq, fast_);

