// K3d "h2 quad patch, two taps per barrier" — round-4 variant of the quad patch kernel (conv_patch_h2q.hip: same tile, same
// wave layout, same products in the same order per accumulator: bitwise the same results).
//
// What bounds the quad kernel is its per-tap synchronisation: a tap step is 1650 ticks of which 36 MFMAs are 576 — the wave
// waits ~290 for the weights it requested one step earlier (the request's round trip is ~1000 cycles under load), ~170-270 at the
// barrier, ~230-360 for the operand reads (profiles/r3j_timeline_h2q.txt, r4a_timeline_h2q_192.txt).  A third ring stage would
// hide the round trip, but 2 patch buffers + 3 weight stages are 89 KB and two workgroups per CU have 80.  This kernel
// spends the LDS differently:
//   * ONE patch buffer (24.6 KB) instead of two.  The next chunk's patch is requested right behind the barrier of the
//     chunk's last tap (every wave has finished reading the buffer by then: each waits for its own LDS reads before that
//     barrier) by all four waves, three spans each, and has tap 8's MFMAs to land; so that it comes from L2 / MALL and not
//     from HBM, every wave touches its spans' cache lines of the next chunk early in the chunk (an LDS-DMA of the h piece into a
//     1 KB scratch slot: no registers, in order on the wave's vmcnt);
//   * the 24.6 KB saved hold FOUR one-tap weight slots: taps go in groups (0,1) (2,3) (4,5) (6,7) (8) with ONE barrier per
//     group; the weights of a group are requested at the start of the previous group — two taps (~2000 cycles) ahead — so the
//     wait in front of the barrier is for data that has long landed.  5 barriers and 5 waits per chunk instead of 9;
//   * rows slide through the four register slots exactly as in the quad kernel; the rows of a tap are read right behind the
//     previous tap's MFMAs (no barrier in between inside a chunk: the patch is static); only tap 0's four rows wait for the
//     group's barrier (the patch has just landed).
// LDS: 24 576 (patch) + 4 x 12 288 (weight slots) + 7 168 (prefetch sinks, span offsets) = 80 896 B: 2 workgroups per CU.
#include "h2_common.h"

namespace padel {

namespace {

constexpr int kDPW = 18;                        // patch width in pixels (16 + halo)
constexpr int kDNPix = 180;                     // 10 x 18
constexpr int kDPlaneB = 192 * 64;              // one fp16 plane of a 32-channel chunk, padded to 12 spans of 16 pixels
constexpr int kDPatchB = 2 * kDPlaneB;

__device__ __forceinline__ unsigned hd_off(int p, int q) { return (unsigned)(p * 64 + ((q ^ (((p >> 2) & 1) << 1)) << 4)); }

}  // namespace

#ifndef PADEL_HD_PREFETCH
#define PADEL_HD_PREFETCH 1
#endif

template <int NF>
__global__ void __launch_bounds__(256, 2) conv_h2d_kernel(const ConvArgs a) {
    constexpr int MF = 4;
    constexpr int BN = 2 * NF * 16;              // output channels per workgroup
    constexpr int BPLANE_B = BN * 64;
    constexpr int BSLOT_B = 2 * BPLANE_B;        // one tap's weights (h | m planes)
    static_assert(NF == 3, "weight requests are laid out for 6 spans of 16 rows per plane: 2 per wave 0..2");
    constexpr int SCRATCH_B = 4 * (1024 + 768);  // per wave: 1 KB prefetch sink + its three span offsets (64 lanes x 4 B each)
    static_assert(kDPatchB + 4 * BSLOT_B + SCRATCH_B <= 80 * 1024, "2 workgroups per CU");
    __shared__ __attribute__((aligned(16))) float lds[(kDPatchB + 4 * BSLOT_B + SCRATCH_B) / 4];
    char* const ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 1, wc = wave >> 1;      // pixel half (rows 4 wr ..), channel half (fragments 3 wc ..)
    const int lr = lane & 15, lq = lane >> 4;

    // XCD-aware 1-D tile map: the channel tiles of one pixel patch are neighbours on one XCD
    const int nmt = a.n_mtiles, nnt = a.n_ntiles;
    const int bid = blockIdx.x;
    const int q8 = nmt >> 3, r8 = nmt & 7, xcd = bid & 7, idx = bid >> 3;
    const int mloc = idx / nnt, nt = idx - mloc * nnt;
    if (mloc >= q8 + (xcd < r8 ? 1 : 0)) return;
    const int mt = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + mloc;
    const int txN = (a.Wo + 15) >> 4, tyN = (a.Ho + 7) >> 3;
    const int tpi = tyN * txN;
    const int n = mt / tpi, rt = mt - n * tpi;
    const int ty = rt / txN, tx = rt - ty * txN;
    const int y0 = ty * 8, x0 = tx * 16;
    const int f0 = nt * 2 * NF;

    // ---- the patch: span s of a plane = 16 pixels x 64 bytes, lane i -> pixel 16 s + i / 4, physical 16-byte slot i & 3 =
    // logical chunk q of that pixel (hd_off), which is piece (q & 1) of group (q >> 1) of the pixel's 128 bytes [h0 m0 h1 m1]
    // in HBM; the plane's 32 bytes go in through the scalar offset.  Wave w owns the spans 3 w .. 3 w + 2.
    const float* const in0 = a.in + (((long long)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * a.in_cs + a.in_choff;
    const i32x4 rsrcP = make_rsrc3(in0);
    const unsigned lp0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
    // (the three lane offsets live in LDS between their uses — twice per chunk: in registers they push the kernel over 256)
    unsigned* const ptab = reinterpret_cast<unsigned*>(ldsb + kDPatchB + 4 * BSLOT_B + 4 * 1024) + wave * 192 + lane;
    {
        const int p_lane = lane >> 2;
        const int p_q = (lane & 3) ^ (((lane >> 4) & 1) << 1);
        const unsigned p_piece = (unsigned)((p_q >> 1) * 64 + (p_q & 1) * 16);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int pp = (3 * wave + k) * 16 + p_lane;
            const int py = pp / kDPW, px = pp - py * kDPW;
            const bool ok = pp < kDNPix && (unsigned)(y0 - 1 + py) < (unsigned)a.H && (unsigned)(x0 - 1 + px) < (unsigned)a.W;
            ptab[k * 64] = ok ? (unsigned)((py * a.W + px) * a.in_cs * 4) + p_piece : kOOR3;
        }
    }
    const unsigned lpw = __builtin_amdgcn_readfirstlane(lp0 + (unsigned)(3 * wave) * 1024u);
    // this wave's three spans (both planes) of chunk CH_
#define PADEL_HD_PATCH(CH_)                                                                                       \
    do {                                                                                                          \
        const unsigned so_ = (unsigned)(CH_) * 128u;                                                              \
        const unsigned v0_ = ptab[0], v1_ = ptab[64], v2_ = ptab[128];                                            \
        dma3<0>(v0_, rsrcP, so_, lpw); dma3<kDPlaneB>(v0_, rsrcP, so_ + 32u, lpw);                                \
        dma3<1024>(v1_, rsrcP, so_, lpw); dma3<kDPlaneB + 1024>(v1_, rsrcP, so_ + 32u, lpw);                      \
        dma3<2048>(v2_, rsrcP, so_, lpw); dma3<kDPlaneB + 2048>(v2_, rsrcP, so_ + 32u, lpw);                      \
    } while (0)
#define PADEL_HD_PATCH1(CH_, K_)                                                                                  \
    do {                                                                                                          \
        const unsigned so_ = (unsigned)(CH_) * 128u;                                                              \
        const unsigned v_ = ptab[(K_) * 64];                                                                      \
        dma3<(K_) * 1024>(v_, rsrcP, so_, lpw); dma3<kDPlaneB + (K_) * 1024>(v_, rsrcP, so_ + 32u, lpw);          \
    } while (0)
    // touch the cache lines of this wave's spans of chunk CH_ (the h pieces: one 128-byte line per pixel and chunk)
    const unsigned lscr = __builtin_amdgcn_readfirstlane(lp0 + (unsigned)(kDPatchB + 4 * BSLOT_B) + (unsigned)wave * 1024u);
#define PADEL_HD_TOUCH(CH_)                                                                                       \
    do {                                                                                                          \
        const unsigned so_ = (unsigned)(CH_) * 128u;                                                              \
        const unsigned v0_ = ptab[0], v1_ = ptab[64], v2_ = ptab[128];                                            \
        dma3<0>(v0_, rsrcP, so_, lscr); dma3<0>(v1_, rsrcP, so_, lscr); dma3<0>(v2_, rsrcP, so_, lscr);           \
    } while (0)

    // ---- weights (waves 0..2): rows of (cin / 32) * 9 k-steps x 128 bytes (h | m); wave w requests the spans 2 w, 2 w + 1
    // (16 rows x 64 bytes) of both planes: lane i -> row i / 4 of the span, physical slot i & 3.  Tap g (running index
    // 9 c + t) lives in ring slot g & 3 = (c + t) & 3.
    const int nch = a.cin >> 5;
    const unsigned rowb = (unsigned)(nch * 9) * 128u;
    const int b_row = lane >> 2;
    const int b_sc = (lane & 3) ^ ((4 - ((b_row >> 2) & 3)) & 3);
    unsigned voffB[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int g = min(2 * wave + k, 2 * NF - 1);
        const int frag = min(f0 + g, a.n16 - 1);            // fragments beyond the matrix: any valid rows (never stored)
        voffB[k] = (unsigned)(((frag - f0) * 16 + b_row) * rowb + b_sc * 16);
    }
    const i32x4 rsrcB = make_rsrc3(reinterpret_cast<const char*>(a.w) + (long long)f0 * 16 * rowb);
    const unsigned lwb = __builtin_amdgcn_readfirstlane(lp0 + (unsigned)kDPatchB + (unsigned)wave * 2048u);
    // weights of tap T_ of chunk C_ (k-step 9 C_ + T_) into its ring slot
#define PADEL_HD_DMAW(C_, T_)                                                                                     \
    do {                                                                                                          \
        const unsigned lw_ = lwb + (unsigned)(((C_) + (T_)) & 3) * (unsigned)BSLOT_B;                             \
        const unsigned sb_ = ((unsigned)(C_) * 9u + (unsigned)(T_)) * 128u;                                       \
        dma3<0>(voffB[0], rsrcB, sb_, lw_);                                                                       \
        dma3<1024>(voffB[1], rsrcB, sb_, lw_);                                                                    \
        dma3<BPLANE_B>(voffB[0], rsrcB, sb_ + 64u, lw_);                                                          \
        dma3<BPLANE_B + 1024>(voffB[1], rsrcB, sb_ + 64u, lw_);                                                   \
    } while (0)
    const int ld_off = (3 * wc) * 256 + lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);      // floats
    const float* const b_rd = lds + kDPatchB / 4 + ld_off;
    const int rd_pix = 4 * wr * kDPW + lr;                 // patch pixel of the wave's row 0, kx = 0

    f32x4 acc[MF][NF], part[MF][NF], cross[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = acc[f][j]; cross[f][j] = acc[f][j]; }
    h16x8 ah[4], am[4], wh[NF], wm[NF];       // ah / am: input rows in 4 sliding slots (row r of the current kx in slot r & 3)
    // input row R_ (0..5 of the wave's window) at column shift KX_ into its slot
#define PADEL_HD_READROW(R_, KX_)                                                                                 \
    do {                                                                                                          \
        const char* p_ = ldsb + hd_off(rp_ + (R_) * kDPW + (KX_), lq);                                            \
        ah[(R_) & 3] = *reinterpret_cast<const h16x8*>(p_);                                                       \
        am[(R_) & 3] = *reinterpret_cast<const h16x8*>(p_ + kDPlaneB);                                            \
    } while (0)
    // the rows tap T_ needs that are not in registers yet: ky == 0 -> rows 0..3 of the new column, ky == 1 -> row 4 (slot of
    // row 0), ky == 2 -> row 5 (slot of row 1)
#define PADEL_HD_ROWS(T_)                                                                                         \
    do {                                                                                                          \
        constexpr int kx_ = h2_tap_kx(T_), ky_ = h2_tap_ky(T_);                                                   \
        int rp_ = rd_pix;                                  /* row addresses recomputed per tap (3 VALU each): 18 hoisted ones would spill */ \
        asm volatile("" : "+v"(rp_));                                                                             \
        if constexpr (ky_ == 0) { PADEL_HD_READROW(0, kx_); PADEL_HD_READROW(1, kx_); PADEL_HD_READROW(2, kx_); PADEL_HD_READROW(3, kx_); } \
        else PADEL_HD_READROW(3 + ky_, kx_);                                                                      \
    } while (0)
#define PADEL_HD_READB(T_)                                                                                        \
    do {                                                                                                          \
        const float* const br_ = b_rd + (unsigned)((c + (T_)) & 3) * (unsigned)(BSLOT_B / 4);                     \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) {                                                          \
            wh[j] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(br_ + j * 256));                    \
            wm[j] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(br_ + BPLANE_B / 4 + j * 256));     \
        }                                                                                                         \
    } while (0)
    // the 9 products of output row F_ at tap row KY_ (its input row F_ + KY_ sits in slot (F_ + KY_) & 3)
#define PADEL_HD_MFMA_ROW(F_, KY_)                                                                                \
    do {                                                                                                          \
        constexpr int s_ = ((F_) + (KY_)) & 3;                                                                    \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            cross[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], am[s_], cross[F_][j], 0, 0, 0);          \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            cross[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[j], ah[s_], cross[F_][j], 0, 0, 0);          \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            part[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], ah[s_], part[F_][j], 0, 0, 0);            \
    } while (0)
    // one tap: weights of its slot into registers (the group's barrier published them), 36 MFMAs with this wave's requests
    // REQ0_ / REQ1_ / REQ2_ issued under output rows 0 / 1 / 2 (an LDS-DMA request costs 60-180 cycles of issue: behind the
    // barrier it would sit on the critical path, under nine queued MFMAs it does not), then the rows of the next tap of the
    // chunk (the patch is static inside a chunk)
#define PADEL_HD_TAP(T_, REQ0_, REQ1_, REQ2_)                                                                     \
    do {                                                                                                          \
        constexpr int ky_ = h2_tap_ky(T_);                                                                        \
        PADEL_HD_READB(T_);                                                                                       \
        if constexpr ((T_) == 0) PADEL_HD_ROWS(0);                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        PADEL_HD_MFMA_ROW(0, ky_);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        REQ0_;                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_HD_MFMA_ROW(1, ky_);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        REQ1_;                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_HD_MFMA_ROW(2, ky_);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        REQ2_;                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_HD_MFMA_ROW(3, ky_);                                                                                \
        __builtin_amdgcn_s_setprio(0);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if constexpr ((T_) < 8) PADEL_HD_ROWS((T_) + 1);                                                          \
    } while (0)
    // start of the group whose first tap is T_: this wave's requests have landed (the group's weights, requested under the
    // first tap of the previous group; at T_ == 0 its share of the chunk's patch and the weights requested under tap 8) —
    // barrier: they are published, and the slots / the patch buffer the previous group read are free
#define PADEL_HD_GROUP(T_)                                                                                        \
    do {                                                                                                          \
        if constexpr ((T_) == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* this wave is done with the patch */ \
        wait_vm3<0>();                                                                                            \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
    } while (0)
#define PADEL_HD_W(T_) do { if (wave != 3) PADEL_HD_DMAW(c, T_); } while (0)                         /* weights of tap T_ of this chunk */
#define PADEL_HD_WN(T_) do { if (wave != 3 && c + 1 < nch) PADEL_HD_DMAW(c + 1, T_); } while (0)     /* ... of the next chunk */
#define PADEL_HD_PN(K_) do { if (c + 1 < nch) PADEL_HD_PATCH1(c + 1, K_); } while (0)             /* span K_ of this wave's three of the next patch */
#define PADEL_HD_TN() do { if (PADEL_HD_PREFETCH && c + 1 < nch) PADEL_HD_TOUCH(c + 1); } while (0)
#define PADEL_HD_NONE() do { } while (0)

    PADEL_HD_PATCH(0);
    if (wave != 3) { PADEL_HD_DMAW(0, 0); PADEL_HD_DMAW(0, 1); }
    for (int c = 0; c < nch; ++c) {
        PADEL_HD_GROUP(0); PADEL_HD_TAP(0, PADEL_HD_W(2), PADEL_HD_W(3), PADEL_HD_NONE()); PADEL_HD_TAP(1, PADEL_HD_TN(), PADEL_HD_NONE(), PADEL_HD_NONE());
        PADEL_HD_GROUP(2); PADEL_HD_TAP(2, PADEL_HD_W(4), PADEL_HD_W(5), PADEL_HD_NONE()); PADEL_HD_TAP(3, PADEL_HD_NONE(), PADEL_HD_NONE(), PADEL_HD_NONE());
        PADEL_HD_GROUP(4); PADEL_HD_TAP(4, PADEL_HD_W(6), PADEL_HD_W(7), PADEL_HD_NONE()); PADEL_HD_TAP(5, PADEL_HD_NONE(), PADEL_HD_NONE(), PADEL_HD_NONE());
        PADEL_HD_GROUP(6); PADEL_HD_TAP(6, PADEL_HD_W(8), PADEL_HD_NONE(), PADEL_HD_NONE()); PADEL_HD_TAP(7, PADEL_HD_NONE(), PADEL_HD_NONE(), PADEL_HD_NONE());
        PADEL_HD_GROUP(8); PADEL_HD_TAP(8, do { PADEL_HD_PN(0); PADEL_HD_PN(1); } while (0), do { PADEL_HD_PN(2); PADEL_HD_WN(0); } while (0), PADEL_HD_WN(1));
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
    wait_vm3<0>();
#undef PADEL_HD_GROUP
#undef PADEL_HD_W
#undef PADEL_HD_WN
#undef PADEL_HD_PN
#undef PADEL_HD_TN
#undef PADEL_HD_NONE
#undef PADEL_HD_TAP
#undef PADEL_HD_MFMA_ROW
#undef PADEL_HD_READB
#undef PADEL_HD_ROWS
#undef PADEL_HD_READROW
#undef PADEL_HD_DMAW
#undef PADEL_HD_TOUCH
#undef PADEL_HD_PATCH
#undef PADEL_HD_PATCH1

    int mpix[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int oy = y0 + 4 * wr + f, ox = x0 + lr;
        mpix[f] = (oy < a.Ho && ox < a.Wo) ? (n * a.Ho + oy) * a.Wo + ox : -1;
    }
    const int fw = f0 + NF * wc;
    const bool fast = y0 + 8 <= a.Ho && x0 + 16 <= a.Wo && (fw + NF) * 16 <= a.cout && (((a.out_choff | a.out_cs) & 3) == 0) &&
                      (!a.res || (((a.res_choff | a.res_cs) & 3) == 0));
    if (fw < a.n16) h2_epilogue<MF, NF>(a, acc, cross, mpix, fw, lq, fast);
}

hipError_t launch_conv_h2d(const ConvArgs& a_in, hipStream_t s) {
    if (!conv_h2q_supported(a_in)) return hipErrorNotSupported;
    ConvArgs a = a_in;
    const int batch = a.M / (a.Ho * a.Wo);
    a.n_mtiles = batch * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
    a.n_ntiles = (a.n16 + 5) / 6;
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    hipLaunchKernelGGL((conv_h2d_kernel<3>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace padel
