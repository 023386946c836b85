// K3p "h2 patch" — the stride-1 3x3 convolution of h2 graphs (activations = fp16 pairs, h2_common.h).
//
// Structure of conv_patch_bx3.hip with the split gone: a workgroup owns an 8 x 16 patch of output pixels of one image
// and, per 32-channel chunk,
//   1. fetches the 10 x 18 input patch (halo included; pixels outside the image come back as zeros from the buffer range
//      check), 128 bytes per pixel = [h0 m0 h1 m1] of its two 16-channel groups, into registers one chunk ahead
//      (6 x buffer_load_dwordx4 per lane) and
//   2. stores the pieces unchanged into two fp16 planes in LDS (h, m: 64 bytes per pixel and plane; 16-byte chunk q of
//      pixel p lives at q ^ 2 * ((p >> 2) & 1) — the conflict-free placement of the bf16x3 patch kernel);
//   3. walks the 9 taps as SHIFTED 16-pixel windows of those planes, read as ready-made MFMA operands; only the tap's
//      weights (two planes) travel through a 2-stage LDS-DMA ring.
// Per tap step and wave: 2 (MF + NF) ds_read_b128 + 3 MF NF MFMAs (v_mfma_f32_16x16x32_f16) and no VALU.  Accumulation:
// cross (both correction products, own accumulator), part (main product, flushed into acc once per chunk).  Products
// and their order are those of the tap kernel (conv_tap_h2.hip): results are bitwise identical to it.
//
// LDS: 2 planes x 180 pixels x 64 B = 23 040 B + 2 weight stages of 2 x BN x 64 B: 35 328 B for BN = 48.
// cin % 32 == 16: a 16-channel tail patch (one group: 32 bytes per pixel and plane) walked in 5 steps that pair taps —
// lane group q of an operand holds the 8 channels 8 (q & 1).. of tap 2t + (q >> 1).
// UP: the first a.up_c channels (whole chunks) are read from a.in2, a map of half the spatial size, at [y >> 1][x >> 1]
// (an nn.Upsample(2) + torch.cat in front of this conv — TrackNet's decoder blocks — that is never materialised).
#include "h2_common.h"

namespace padel {

namespace {

typedef unsigned hp_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kHPW = 18;                        // patch width in pixels (16 + halo)
constexpr int kHNPix = 180;                     // 10 x 18
constexpr int kHPlaneB = kHNPix * 64;           // one fp16 plane of a 32-channel chunk
constexpr int kHPatchB = 2 * kHPlaneB;
constexpr int kHItems = kHNPix * 8;             // 16-byte pieces of the 128-byte-per-pixel patch
constexpr int kHPasses = (kHItems + 255) / 256; // 6
constexpr int kHTailPasses = (kHNPix * 4 + 255) / 256;     // 3

// byte offset, inside a plane, of logical 16-byte chunk q (K slots 8q..8q+7) of patch pixel p
__device__ __forceinline__ unsigned hp_off(int p, int q) { return (unsigned)(p * 64 + ((q ^ (((p >> 2) & 1) << 1)) << 4)); }
// tail planes: 32 bytes per pixel, 16-byte slot s (channels 8s..8s+7)
__device__ __forceinline__ unsigned hp_tail_off(int p, int s) { return (unsigned)(p * 32 + ((s ^ ((p >> 3) & 1)) << 4)); }
__device__ __forceinline__ void hp_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

}  // namespace

// PIPE: software-pipelined tap walk.  The plain schedule serialises, per tap step and wave, [wait, barrier, operand
// reads (LDS latency), weight requests] in front of a burst of 3 MF NF MFMAs: ~220 cycles that only OTHER waves can
// cover (measured: matrix pipe busy 0.56 where the six-product bf16x3 kernel, same structure and twice the burst, had
// 0.72).  PIPE splits the burst: after the correction products of step T the wave passes the barrier of step T + 1 and
// issues the operand reads of T + 1 into a SECOND register set, then runs the main products of step T under their
// latency.  Costs 40 more VGPRs (2 waves per SIMD instead of 3).  Same products in the same order per accumulator:
// bitwise equal to the plain schedule.
// PROBE (builds with -DPADEL_H2P_PROBES only; WRONG results, ceilings for tools/conv_bench.py): bit 0 no barrier on tap
// steps 1..8, bit 1 no weight reads there, bit 2 no weight requests there, bit 3 no patch reads there, bit 4 no MFMAs
// WS: the packed weights' m plane is all zero (ConvArgs::w_single): no wm x ah product, no m-plane requests / reads (conv_patch_h2q.hip)
template <int NF, bool TAIL, bool UP, bool PIPE, int PROBE = 0, bool WS = false>
__global__ void __launch_bounds__(256, (NF <= 3 && !UP && !PIPE) ? 3 : 2) conv_h2p_kernel(const ConvArgs a) {
    constexpr int MF = 2;
    constexpr bool TWOL = NF <= 4;               // two-level main accumulation (part -> acc once per chunk) where registers allow
    constexpr int BN = NF * 16;
    constexpr int BSTAGE_B = 2 * BN * 64;
    constexpr int BP = (BN + 63) / 64, BFULL = BN / 64;
    static_assert(BP <= 2, "weights in at most 2 passes of 64 rows");
    static_assert(kHPatchB + 2 * BSTAGE_B <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) float lds[(kHPatchB + 2 * BSTAGE_B) / 4];
    char* const ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int f0 = nt * NF;

    // ---- the patch: piece i * 256 + tid = (pixel, 16-byte piece 0..7 of its 128-byte chunk: group g = piece >> 2,
    // plane (piece >> 1) & 1, half piece & 1); lane offsets are chunk-independent
    unsigned voffP[kHPasses];
#pragma unroll
    for (int i = 0; i < kHPasses; ++i) {
        const int item = i * 256 + tid;
        const int pp = item >> 3, pc = item & 7;
        const int py = pp / kHPW, px = pp - py * kHPW;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = item < kHItems && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        voffP[i] = ok ? (unsigned)((py * a.W + px) * a.in_cs * 4 + pc * 16) : kOOR3;
    }
    // LDS place of piece (pixel pp = i * 32 + tid / 8, piece pc = tid & 7): plane (pc >> 1) & 1, logical chunk (pc >> 2) * 2 + (pc & 1)
    const unsigned wr0 = (unsigned)((((tid & 7) >> 1) & 1) * kHPlaneB);
    const int wr_q = ((tid & 7) >> 2) * 2 + (tid & 1);
    const float* const in0 = a.in + (((long long)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * a.in_cs + a.in_choff;
    const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in0), 0, (int)0x80000000u, 0x00020000);
    // coarse map of an absorbed upsample: descriptor based at the coarse pixel of the patch's top-left halo pixel
    const int H2 = a.H >> 1, W2 = a.W >> 1;
    const int cy0 = (y0 - 1) >> 1, cx0 = (x0 - 1) >> 1;                 // arithmetic shifts: -1 for the halo above / left of the image
    const float* const inU = UP ? a.in2 + (((long long)n * H2 + cy0) * W2 + cx0) * a.in2_cs + a.in2_choff : in0;
    const __amdgpu_buffer_rsrc_t rsrcU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inU), 0, (int)0x80000000u, 0x00020000);
    const int nup = UP ? a.up_c >> 5 : 0;
    (void)H2; (void)W2; (void)cy0; (void)cx0; (void)rsrcU; (void)nup;

    // ---- weights: rows of (cin / 32) * 9 (+ 5) k-steps x 128 bytes (h | m), k-step = chunk * 9 + tap
    const int nch = a.cin >> 5;
    const unsigned rowb = (unsigned)(nch * 9 + (TAIL ? 5 : 0)) * 128u;
    const int srow = tid >> 2;
    const int sc = (tid & 3) ^ ((4 - ((srow >> 2) & 3)) & 3);
    unsigned voffB[BP];
#pragma unroll
    for (int p = 0; p < BP; ++p) {
        const int rr = srow + 64 * p;
        const int frag = min(f0 + (rr >> 4), a.n16 - 1);
        voffB[p] = (unsigned)(((frag - f0) * 16 + (rr & 15)) * rowb + sc * 16);
    }
    const i32x4 rsrcB = make_rsrc3(reinterpret_cast<const char*>(a.w) + (long long)f0 * 16 * rowb);
    const bool b_last = BP > BFULL && (BFULL * 64 + wave * 16 < BN);
    unsigned lw0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + (unsigned)kHPatchB + wave * 1024u);
    unsigned lw1 = __builtin_amdgcn_readfirstlane(lw0 + (unsigned)BSTAGE_B);
    const int ld_off = lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);
    const float* b_rd0 = lds + kHPatchB / 4 + ld_off;
    const float* b_rd1 = b_rd0 + BSTAGE_B / 4;
    const int rd_pix = 2 * wave * kHPW + lr;                // patch pixel of fragment 0, tap (0, 0)

    f32x4 acc[MF][NF], part[MF][NF], cross[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = acc[f][j]; cross[f][j] = acc[f][j]; }
    f32x4 (&pmain)[MF][NF] = TWOL ? part : acc;

#define PADEL_HP_DMAB(SR_, SB_)                                                                                   \
    do {                                                                                                          \
        const unsigned lw_ = ((SR_) & 1) ? lw1 : lw0;                                                             \
        const unsigned sb_ = (SB_);                                                                               \
        PADEL_HP_DMAB1(0, sb_);                                                                                   \
        if constexpr (!WS) PADEL_HP_DMAB1(1, sb_ + 64u);                                                          \
    } while (0)
#define PADEL_HP_DMAB1(PL_, S_)                                                                                   \
    do {                                                                                                          \
        if constexpr (BFULL >= 1) dma3<(PL_) * BN * 64>(voffB[0], rsrcB, (S_), lw_);                              \
        if constexpr (BP > BFULL) { if (b_last) dma3<(PL_) * BN * 64 + BFULL * 4096>(voffB[BP - 1], rsrcB, (S_), lw_); } \
    } while (0)
#define PADEL_HP_LOAD(CH_)                                                                                        \
    do {                                                                                                          \
        const unsigned so_ = (unsigned)(CH_) * 128u;                                                              \
        if (UP && (int)(CH_) < nup) {      /* lane offsets into the coarse map, recomputed (once per chunk) */      \
            _Pragma("unroll") for (int i = 0; i < kHPasses; ++i) {                                                \
                const int item = i * 256 + tid;                                                                   \
                const int pp = item >> 3, pc = item & 7;                                                          \
                const int py = pp / kHPW, px = pp - py * kHPW;                                                    \
                const int iy = y0 - 1 + py, ix = x0 - 1 + px;                                                     \
                const bool ok = item < kHItems && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;  \
                const unsigned vo_ = ok ? (unsigned)((((iy >> 1) - cy0) * W2 + ((ix >> 1) - cx0)) * a.in2_cs * 4 + pc * 16) : kOOR3; \
                pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcU, vo_, so_, 0);                               \
            }                                                                                                     \
        } else {                                                                                                  \
            _Pragma("unroll") for (int i = 0; i < kHPasses; ++i) pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcP, voffP[i], so_, 0); \
        }                                                                                                         \
    } while (0)
#define PADEL_HP_READA(T_)                                                                                        \
    do {                                                                                                          \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                          \
            const char* p_ = ldsb + hp_off(rd_pix + (f + h2_tap_ky(T_)) * kHPW + h2_tap_kx(T_), lq);                        \
            ah[f] = *reinterpret_cast<const h16x8*>(p_);                                                          \
            am[f] = *reinterpret_cast<const h16x8*>(p_ + kHPlaneB);                                               \
        }                                                                                                         \
    } while (0)
    // tail step JT: lane group q reads the 8 channels 8 (q & 1).. of tap 2 JT + (q >> 1) (the 10th "tap" has zero weights:
    // any finite data, tap 8 again)
#define PADEL_HP_TREADA(JT_)                                                                                      \
    do {                                                                                                          \
        constexpr int ta_ = 2 * (JT_), tb_ = 2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8;                               \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                          \
            const int pa_ = rd_pix + (f + h2_tap_ky(ta_)) * kHPW + h2_tap_kx(ta_), pb_ = rd_pix + (f + h2_tap_ky(tb_)) * kHPW + h2_tap_kx(tb_); \
            const char* p_ = ldsb + hp_tail_off((lq >> 1) ? pb_ : pa_, lq & 1);                                   \
            ah[f] = *reinterpret_cast<const h16x8*>(p_);                                                          \
            am[f] = *reinterpret_cast<const h16x8*>(p_ + kHPlaneB);                                               \
        }                                                                                                         \
    } while (0)
#define PADEL_HP_READB(T_)                                                                                        \
    do {                                                                                                          \
        const float* const br_ = ((T_) & 1) ? b_rd1 : b_rd0;                                                      \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) {                                                          \
            wh[j] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(br_ + j * 256));                    \
            if constexpr (!WS) wm[j] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(br_ + BN * 16 + j * 256)); \
        }                                                                                                         \
    } while (0)
#define PADEL_HP_MFMA()                                                                                           \
    do {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            cross[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], am[f], cross[f][j], 0, 0, 0);             \
        if constexpr (!WS) {                                                                                      \
            _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)         \
                cross[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[j], ah[f], cross[f][j], 0, 0, 0);         \
        }                                                                                                         \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            pmain[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], ah[f], pmain[f][j], 0, 0, 0);             \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)
    // ---- PIPE: operand sets indexed by step parity
#define PADEL_HP_READA2(T_, S_)                                                                                   \
    do {                                                                                                          \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                          \
            const char* p_ = ldsb + hp_off(rd_pix + (f + h2_tap_ky(T_)) * kHPW + h2_tap_kx(T_), lq);                        \
            ah2[S_][f] = *reinterpret_cast<const h16x8*>(p_);                                                     \
            am2[S_][f] = *reinterpret_cast<const h16x8*>(p_ + kHPlaneB);                                          \
        }                                                                                                         \
    } while (0)
#define PADEL_HP_TREADA2(JT_, S_)                                                                                 \
    do {                                                                                                          \
        constexpr int ta_ = 2 * (JT_), tb_ = 2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8;                               \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                          \
            const int pa_ = rd_pix + (f + h2_tap_ky(ta_)) * kHPW + h2_tap_kx(ta_), pb_ = rd_pix + (f + h2_tap_ky(tb_)) * kHPW + h2_tap_kx(tb_); \
            const char* p_ = ldsb + hp_tail_off((lq >> 1) ? pb_ : pa_, lq & 1);                                   \
            ah2[S_][f] = *reinterpret_cast<const h16x8*>(p_);                                                     \
            am2[S_][f] = *reinterpret_cast<const h16x8*>(p_ + kHPlaneB);                                          \
        }                                                                                                         \
    } while (0)
#define PADEL_HP_READB2(T_, S_)                                                                                   \
    do {                                                                                                          \
        const float* const br_ = ((T_) & 1) ? b_rd1 : b_rd0;                                                      \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) {                                                          \
            wh2[S_][j] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(br_ + j * 256));               \
            if constexpr (!WS) wm2[S_][j] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(br_ + BN * 16 + j * 256)); \
        }                                                                                                         \
    } while (0)
#define PADEL_HP_MFMA_CROSS(S_)                                                                                   \
    do {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            cross[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh2[S_][j], am2[S_][f], cross[f][j], 0, 0, 0);   \
        if constexpr (!WS) {                                                                                      \
            _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)         \
                cross[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm2[S_][j], ah2[S_][f], cross[f][j], 0, 0, 0); \
        }                                                                                                         \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)
#define PADEL_HP_MFMA_MAIN(S_)                                                                                    \
    do {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            pmain[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh2[S_][j], ah2[S_][f], pmain[f][j], 0, 0, 0);   \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)
    // entry of a block of steps (a chunk's 9 taps or the tail's 5 pairs): this wave's plane writes are fenced and the weights
    // of step 0 — requested during the previous block / the prologue — have landed: barrier, operands of step 0 into set 0,
    // request the weights of step 1 (the stage they go to was last read before the correction products of the previous
    // block's last step, i.e. before every wave reached this barrier)
#define PADEL_HP_PENTRY(READA0_)                                                                                  \
    do {                                                                                                          \
        wait_vm3<0>();                                                                                            \
        hp_lds_fence();                                                                                           \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
        PADEL_HP_READB2(0, 0);                                                                                    \
        READA0_;                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_HP_DMAB(1, s_kb + 128u);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
    // step T_ of a block of NT_ steps (its operands sit in set T_ & 1): correction products; then — for step T_ + 1 — wait
    // for its weights, barrier, operand reads into the other set, request of the weights two steps ahead (the first ones of
    // the next block when MORE_) and, at T_ == 0, whatever EXTRA0_ prefetches; then the main products, under the latency
    // of those reads
#define PADEL_HP_PSTEP(T_, NT_, READNEXT_, MORE_, EXTRA0_)                                                        \
    do {                                                                                                          \
        PADEL_HP_MFMA_CROSS((T_) & 1);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if constexpr ((T_) + 1 < (NT_)) {                                                                         \
            wait_vm3<0>();                                                                                        \
            __builtin_amdgcn_s_barrier();                                                                         \
            asm volatile("" ::: "memory");                                                                        \
            PADEL_HP_READB2((T_) + 1, ((T_) + 1) & 1);                                                            \
            READNEXT_;                                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            if ((T_) + 2 < (NT_) || (MORE_)) PADEL_HP_DMAB((T_) + 2, s_kb + ((T_) + 2) * 128u);                   \
            if constexpr ((T_) == 0) { EXTRA0_; }                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        }                                                                                                         \
        PADEL_HP_MFMA_MAIN((T_) & 1);                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
    // tap step T of the current chunk: its weights (requested one step earlier) have landed for every wave after the
    // barrier, which also releases the other weight stage (read in step T - 1) for the request of step T + 1; step 0
    // additionally publishes the freshly written planes and requests the next chunk's patch
#define PADEL_HP_STEP(T_)                                                                                         \
    do {                                                                                                          \
        constexpr bool first_ = (T_) == 0;                                                                        \
        if constexpr (!first_ && !(PROBE & 8)) PADEL_HP_READA(T_);   /* the planes are static inside a chunk: read under the wait */ \
        if constexpr (first_ || !(PROBE & 4)) wait_vm3<0>();                                                      \
        if constexpr (first_) hp_lds_fence();           /* this wave's plane writes have reached the LDS */         \
        if constexpr (first_ || !(PROBE & 1)) __builtin_amdgcn_s_barrier();                                       \
        asm volatile("" ::: "memory");                                                                            \
        if constexpr (first_ || !(PROBE & 2)) PADEL_HP_READB(T_);                                                 \
        if constexpr (first_) PADEL_HP_READA(T_);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if constexpr (first_ || (T_) == 8 || !(PROBE & 4)) {                                                      \
            if ((T_) < 8 || c + 1 < nch || TAIL) PADEL_HP_DMAB((T_) + 1, s_kb + ((T_) + 1) * 128u);               \
        }                                                                                                         \
        if ((T_) == 0 && c + 1 < nch) PADEL_HP_LOAD(c + 1);                                                       \
        if constexpr (TAIL) { if ((T_) == 0 && c + 1 == nch) PADEL_HP_TLOAD(); }                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if constexpr (!(PROBE & 16)) PADEL_HP_MFMA();                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

    // tail patch: 180 pixels x 4 pieces [h lo | h hi | m lo | m hi] of the one 16-channel group
#define PADEL_HP_TLOAD()                                                                                          \
    do {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < kHTailPasses; ++i) {                                                \
            const int item = i * 256 + tid;                                                                       \
            const int pp = item >> 2, pc = item & 3;                                                              \
            const int py = pp / kHPW, px = pp - py * kHPW;                                                        \
            const bool ok = item < kHNPix * 4 && (unsigned)(y0 - 1 + py) < (unsigned)a.H && (unsigned)(x0 - 1 + px) < (unsigned)a.W; \
            pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcP, ok ? (unsigned)((py * a.W + px) * a.in_cs * 4 + pc * 16) : kOOR3, \
                                                           (unsigned)nch * 128u, 0);                              \
        }                                                                                                         \
    } while (0)
#define PADEL_HP_TSTEP(JT_)                                                                                       \
    do {                                                                                                          \
        wait_vm3<0>();                                                                                            \
        hp_lds_fence();                                                                                           \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
        if constexpr ((JT_) < 4) PADEL_HP_DMAB((JT_) + 1, s_kb + ((JT_) + 1) * 128u);                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_HP_TREADA(JT_);                                                                                     \
        PADEL_HP_READB(JT_);                                                                                      \
        PADEL_HP_MFMA();                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

    hp_u32x4 pre[kHPasses];
    unsigned s_kb = 0;
    if (!TAIL || nch > 0) PADEL_HP_LOAD(0); else PADEL_HP_TLOAD();
    PADEL_HP_DMAB(0, 0u);
    if constexpr (PIPE) {
        h16x8 ah2[2][MF], am2[2][MF], wh2[2][NF], wm2[2][NF];
        for (int c = 0; c < nch; ++c) {
            if (c > 0) {                          // every wave is done with the taps of chunk c - 1: the planes may be overwritten
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int i = 0; i < kHPasses; ++i)
                if (i * 256 + 255 < kHItems || i * 256 + tid < kHItems)
                    *reinterpret_cast<hp_u32x4*>(ldsb + wr0 + hp_off(i * 32 + (tid >> 3), wr_q)) = pre[i];
            const bool more = c + 1 < nch || TAIL;
            PADEL_HP_PENTRY(PADEL_HP_READA2(0, 0));
#define PADEL_HP_PREFETCH() do { if (c + 1 < nch) PADEL_HP_LOAD(c + 1); if constexpr (TAIL) { if (c + 1 == nch) PADEL_HP_TLOAD(); } } while (0)
            PADEL_HP_PSTEP(0, 9, PADEL_HP_READA2(1, 1), more, PADEL_HP_PREFETCH());
            PADEL_HP_PSTEP(1, 9, PADEL_HP_READA2(2, 0), more, (void)0);
            PADEL_HP_PSTEP(2, 9, PADEL_HP_READA2(3, 1), more, (void)0);
            PADEL_HP_PSTEP(3, 9, PADEL_HP_READA2(4, 0), more, (void)0);
            PADEL_HP_PSTEP(4, 9, PADEL_HP_READA2(5, 1), more, (void)0);
            PADEL_HP_PSTEP(5, 9, PADEL_HP_READA2(6, 0), more, (void)0);
            PADEL_HP_PSTEP(6, 9, PADEL_HP_READA2(7, 1), more, (void)0);
            PADEL_HP_PSTEP(7, 9, PADEL_HP_READA2(8, 0), more, (void)0);
            PADEL_HP_PSTEP(8, 9, (void)0, more, (void)0);
#undef PADEL_HP_PREFETCH
            if constexpr (TWOL) {
#pragma unroll
                for (int f = 0; f < MF; ++f)
#pragma unroll
                    for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            }
            { const float* t_ = b_rd0; b_rd0 = b_rd1; b_rd1 = t_; const unsigned u_ = lw0; lw0 = lw1; lw1 = u_; }
            s_kb += 9u * 128u;
        }
        if constexpr (TAIL) {
            if (nch > 0) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int i = 0; i < kHTailPasses; ++i) {
                const int item = i * 256 + tid;
                const int pp = item >> 2, pc = item & 3;
                if (item < kHNPix * 4) *reinterpret_cast<hp_u32x4*>(ldsb + (pc >> 1) * kHPlaneB + hp_tail_off(pp, pc & 1)) = pre[i];
            }
            PADEL_HP_PENTRY(PADEL_HP_TREADA2(0, 0));
            PADEL_HP_PSTEP(0, 5, PADEL_HP_TREADA2(1, 1), false, (void)0);
            PADEL_HP_PSTEP(1, 5, PADEL_HP_TREADA2(2, 0), false, (void)0);
            PADEL_HP_PSTEP(2, 5, PADEL_HP_TREADA2(3, 1), false, (void)0);
            PADEL_HP_PSTEP(3, 5, PADEL_HP_TREADA2(4, 0), false, (void)0);
            PADEL_HP_PSTEP(4, 5, (void)0, false, (void)0);
            if constexpr (TWOL) {
#pragma unroll
                for (int f = 0; f < MF; ++f)
#pragma unroll
                    for (int j = 0; j < NF; ++j) acc[f][j] += part[f][j];
            }
        }
    } else {
    h16x8 ah[MF], am[MF], wh[NF], wm[NF];
    for (int c = 0; c < nch; ++c) {
        if (c > 0) {                          // every wave is done with the taps of chunk c - 1: the planes may be overwritten
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < kHPasses; ++i)
            if (i * 256 + 255 < kHItems || i * 256 + tid < kHItems)
                *reinterpret_cast<hp_u32x4*>(ldsb + wr0 + hp_off(i * 32 + (tid >> 3), wr_q)) = pre[i];
        PADEL_HP_STEP(0); PADEL_HP_STEP(1); PADEL_HP_STEP(2); PADEL_HP_STEP(3); PADEL_HP_STEP(4);
        PADEL_HP_STEP(5); PADEL_HP_STEP(6); PADEL_HP_STEP(7); PADEL_HP_STEP(8);
        if constexpr (TWOL) {
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
        { const float* t_ = b_rd0; b_rd0 = b_rd1; b_rd1 = t_; const unsigned u_ = lw0; lw0 = lw1; lw1 = u_; }
        s_kb += 9u * 128u;
    }
    if constexpr (TAIL) {
        if (nch > 0) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < kHTailPasses; ++i) {
            const int item = i * 256 + tid;
            const int pp = item >> 2, pc = item & 3;
            if (item < kHNPix * 4) *reinterpret_cast<hp_u32x4*>(ldsb + (pc >> 1) * kHPlaneB + hp_tail_off(pp, pc & 1)) = pre[i];
        }
        PADEL_HP_TSTEP(0); PADEL_HP_TSTEP(1); PADEL_HP_TSTEP(2); PADEL_HP_TSTEP(3); PADEL_HP_TSTEP(4);
        if constexpr (TWOL) {
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[f][j] += part[f][j];
        }
    }
    }
    wait_vm3<0>();
#undef PADEL_HP_TSTEP
#undef PADEL_HP_PSTEP
#undef PADEL_HP_PENTRY
#undef PADEL_HP_MFMA_MAIN
#undef PADEL_HP_MFMA_CROSS
#undef PADEL_HP_READB2
#undef PADEL_HP_TREADA2
#undef PADEL_HP_READA2
#undef PADEL_HP_TLOAD
#undef PADEL_HP_TREADA
#undef PADEL_HP_READA
#undef PADEL_HP_READB
#undef PADEL_HP_MFMA
#undef PADEL_HP_STEP
#undef PADEL_HP_LOAD
#undef PADEL_HP_DMAB
#undef PADEL_HP_DMAB1

    int mpix[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int oy = y0 + 2 * wave + f, ox = x0 + lr;
        mpix[f] = (oy < a.Ho && ox < a.Wo) ? (n * a.Ho + oy) * a.Wo + ox : -1;
    }
    const bool fast = y0 + 8 <= a.Ho && x0 + 16 <= a.Wo && (f0 + NF) * 16 <= a.cout && (((a.out_choff | a.out_cs) & 3) == 0) &&
                      (!a.res || (((a.res_choff | a.res_cs) & 3) == 0));
    h2_epilogue<MF, NF>(a, acc, cross, mpix, f0, lq, fast);
}

template <int NF, bool TAIL, bool UP, bool PIPE, int PROBE = 0>
static hipError_t launch_hpt(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    const int batch = a.M / (a.Ho * a.Wo);
    a.n_mtiles = batch * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
    a.n_ntiles = (a.n16 + NF - 1) / NF;
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    if constexpr (PROBE == 0) {
        if (a.w_single) {
            hipLaunchKernelGGL((conv_h2p_kernel<NF, TAIL, UP, PIPE, 0, true>), grid, dim3(256), 0, s, a);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((conv_h2p_kernel<NF, TAIL, UP, PIPE, PROBE>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

template <int NF, bool PIPE>
static hipError_t launch_hp(const ConvArgs& a_in, hipStream_t s) {
    if (a_in.in2) {                    // absorbed upsample: whole 32-channel chunks only, even map size
        if ((a_in.cin & 31) || (a_in.up_c & 31) || a_in.up_c <= 0 || a_in.up_c > a_in.cin || ((a_in.H | a_in.W) & 1)) return hipErrorNotSupported;
        return launch_hpt<NF, false, true, PIPE>(a_in, s);
    }
    if (a_in.cin & 16) return launch_hpt<NF, true, false, PIPE>(a_in, s);
    return launch_hpt<NF, false, false, PIPE>(a_in, s);
}

bool conv_h2p_supported(const ConvArgs& a) {
    return a.ksize == 3 && a.stride == 1 && (a.cin & 15) == 0 && a.cin >= 16 && a.Ho == a.H && a.Wo == a.W && a.w != nullptr;
}

// nf = channel fragments (of 16) per workgroup: 3 (8x16 pixels x 48 channels), 4, 6; + 10: the software-pipelined schedule
hipError_t launch_conv_h2p(const ConvArgs& a, int nf, hipStream_t s) {
    if (!conv_h2p_supported(a)) return hipErrorNotSupported;
    switch (nf) {
        case 3: return launch_hp<3, false>(a, s);
        case 4: return launch_hp<4, false>(a, s);
        case 13: return launch_hp<3, true>(a, s);
        // (round 5 prune: the 6-fragment tile 306 — main product accumulated in ONE level for want of registers, no faster than
        //  303 and 1.3-2 x its RMS error on long K, profiles/conv_h2_sweep_r3a.txt — and the software-pipelined 64-channel tile
        //  314 — 10-15 % slower than 304, profiles/conv_h2_sweep_r3f_pipe.txt — were never chosen automatically: removed)
#ifdef PADEL_H2P_PROBES
#define PADEL_HP_PROBE_CASE(P_) case 32 + (P_): if (a.in2 || (a.cin & 16)) return hipErrorNotSupported; return launch_hpt<3, false, false, false, (P_)>(a, s);
        PADEL_HP_PROBE_CASE(1) PADEL_HP_PROBE_CASE(2) PADEL_HP_PROBE_CASE(3) PADEL_HP_PROBE_CASE(4) PADEL_HP_PROBE_CASE(5)
        PADEL_HP_PROBE_CASE(7) PADEL_HP_PROBE_CASE(8) PADEL_HP_PROBE_CASE(10) PADEL_HP_PROBE_CASE(15) PADEL_HP_PROBE_CASE(16)
        PADEL_HP_PROBE_CASE(17) PADEL_HP_PROBE_CASE(31)
#undef PADEL_HP_PROBE_CASE
#endif
    }
    return hipErrorNotSupported;
}

}  // namespace padel
