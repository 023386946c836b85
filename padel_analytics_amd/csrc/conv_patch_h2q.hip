// K3q "h2 quad patch" — the stride-1 3x3 convolution of h2 graphs for layers with whole 32-channel chunks and a multiple of
// 96 output channels (yolov8m's 96 / 192 / 576-channel bottlenecks: where it beats the 8 x 16 x 48 patch kernel).
//
// What bounds the 48-channel patch kernel (conv_patch_h2.hip) was measured three ways this round:
//   * ablation tiles (profiles/r3h_ablate_h2p.txt): without its MFMAs it still needs 62 % of its time; without barriers,
//     weight requests or operand reads it gains 3-4 % each — no single stage is the bottleneck;
//   * the s_memtime timeline of this kernel (tools/timeline_probe.py --kernel h2q, profiles/r3j_timeline_h2q.txt): per tap
//     step a wave spends ~580 cycles issuing its 36 MFMAs and ~700-900 in barrier skew and LDS latency; the two waves that
//     share a SIMD (one per resident workgroup) overlap those phases only partly, and a workgroup spends a fifth of its life
//     in prologue and epilogue;
//   * a pure-MFMA loop (tools/mfma_f16_ubench.hip, profiles/r3n_mfma_f16_ubench.txt): with random operands the chip
//     sustains 1.70-1.80 PFLOP/s of v_mfma_f32_16x16x32_f16 (2.2 with all-zero operands; nominal 2.5): 580 TFLOP/s in
//     h2's three-product units is the ceiling a perfect kernel would see; the patch kernels reach 375-400.
// This kernel
//   * gives a wave 4 rows x 16 pixels x 48 channels (4 x 3 fragments, 36 MFMAs per tap): a workgroup = 4 waves as
//     2 (pixel halves) x 2 (channel halves) over an 8 x 16 pixel patch and 96 output channels;
//   * walks the taps COLUMN-major (h2_common.h:h2_tap_ky): for one kx the wave needs input rows r .. r + 5 of its column
//     window; rows live in 4 register slots and slide down — 6 row reads per kx instead of 12, so a tap costs 6 (weights)
//     + 4 (rows, average) ds_read_b128 per 36 MFMAs: 0.28 per MFMA against 0.56;
//   * moves the 10 x 18 x 32-channel input patch global -> LDS by LDS-DMA (no registers, no ds_write), double-buffered,
//     requested by wave 3 a sixth per tap under the tap's first MFMAs; waves 0-2 request the weights.  vmcnt is in-order
//     per wave: with the roles split no weight wait ever waits for patch data, which gets a whole chunk (9 taps) to land.
//     One barrier per tap (2-stage weight ring as in the other kernels), none per chunk.
// Measured +2..5 % over the 48-channel tile on 96 -> 96 and 192 -> 192 layers (profiles/r3k_sweep_h2q.txt).  Tried on top
// and dropped (profiles/r3l..r3p_*): persistent workgroups that request their next tile's first chunk during the epilogue
// (+1.6 %: the other resident workgroup already covers most of a prologue), weight requests two steps ahead through a
// second "operands are in registers" barrier per step (+-0: the request latency was not on the critical path), different
// MFMA priorities for the two waves of a SIMD (-4 %).
// Round 4 (gpurun r4a-r4c, tools/ab/ A/B libraries, profiles/r4_quad_variants.txt): the epilogue rewrite (h2_common.h: hoisted
// residual loads, 16-byte stores) gave this kernel +4..8 %.  Five changes to the main loop were then measured on top of it
// and ALL dropped — every combination ran 2-4 % SLOWER than this round-3 loop (96 -> 96: 366 vs 350-352 TFLOP/s, 192 -> 192:
// 415 vs 408-410, pose 96 -> 96: 374 vs 354-356): (1) channel fragments outermost with the third weight operand read under
// the first twelve MFMAs (two weight operand sets, 229 VGPRs): the late read stalls the MFMA stream (36 MFMAs issued in 855
// instead of 673 ticks) by what it saves in front of it; (2) all six weight reads up front in that order: same; (3) the lane
// offsets of the 12 patch spans from an LDS table instead of ~25 VALU per span in wave 3; (4) tap 0's four row reads ahead of
// its barrier (the patch published at the previous chunk's tap-8 barrier); (5) the main chain of a chunk started from the
// constant 0 instead of zeroing 48 registers, its flush either inside tap 8's MFMA stream (tap 8: 1950 ticks) or behind it;
// (6) the first patch requested by all four waves instead of wave 3.  With (3)-(6) on the round-3 row-major order the kernel
// needs 256 VGPRs + 28 bytes of scratch and loses 7 %.  The step is bounded by the weight DMA's round trip (requested one step
// ahead: ~1000 cycles issue -> landed under load), which only a third ring stage would hide — 89 KB of LDS against the 80 KB
// that two workgroups per CU allow.
// Same products in the same order per accumulator as every other h2 kernel (cross: wh am, then wm ah; main: wh ah, flushed
// into acc once per chunk): bitwise identical results.
//
// LDS: 2 patch buffers x 2 planes x 192 pixels x 64 B (180 used; the DMA's last span writes zeros into the pad) = 49 152 B
// + 2 weight stages x 2 planes x 96 rows x 64 B = 24 576 B: 73 728 B, 2 workgroups per CU.
#include "h2_common.h"

namespace padel {

namespace {

constexpr int kQPW = 18;                        // patch width in pixels (16 + halo)
constexpr int kQNPix = 180;                     // 10 x 18
constexpr int kQPlaneB = 192 * 64;              // one fp16 plane of a 32-channel chunk, padded to 12 spans of 16 pixels
constexpr int kQPatchB = 2 * kQPlaneB;

// byte offset, inside a plane, of logical 16-byte chunk q (K slots 8q..8q+7) of patch pixel p (conv_patch_h2.hip:hp_off)
__device__ __forceinline__ unsigned hq_off(int p, int q) { return (unsigned)(p * 64 + ((q ^ (((p >> 2) & 1) << 1)) << 4)); }

// DBG (tuning only, builds with -DPADEL_H2P_PROBES, pa_engine_set_tuning "timeline"): every wave stamps s_memtime at 5
// points of every tap step (step top / own requests landed / barrier passed / operands in registers / last MFMA issued)
// into an LDS ring of 32 steps, dumped to a.dbg at the end (tools/timeline_probe.py --kernel h2q)
constexpr int kQDbgSteps = 32;
constexpr int kQDbgWords = 8 + 4 * kQDbgSteps * 5;

}  // namespace

// WS (round 5): the packed weights' m plane is all zero (ConvArgs::w_single — fp16 checkpoint weights, BatchNorm's scale in the
// output scale): the wm x ah product, the m-plane weight requests and the wm operand reads are compiled out — 24 MFMAs, 2 weight
// requests and 3 + 4 operand reads per tap and wave instead of 36 / 4 / 6 + 4.  Same results as the three-product kernel on such
// weights (the skipped product is exactly zero).  192 -> 192: 408 -> 532 TFLOP/s fp32-equivalent, 96 -> 96: 350 -> 424.
// (Tried on top, gpurun r5r: the freed m-plane stages as a THIRD h-only weight stage, weights requested two steps ahead with a
//  counted vmcnt(2) — bitwise, but 320 / 271 TFLOP/s: slower than the two-stage ring by 40 %; reverted.)
template <int NF, bool DBG = false, bool WS = false>
__global__ void __launch_bounds__(256, 2) conv_h2q_kernel(const ConvArgs a) {
    constexpr int MF = 4;
    constexpr int BN = 2 * NF * 16;              // output channels per workgroup
    constexpr int BPLANE_B = BN * 64;
    constexpr int BSTAGE_B = 2 * BPLANE_B;
    static_assert(NF == 3, "weight requests are laid out for 6 spans of 16 rows per plane: 2 per wave 0..2");
    constexpr int DBG_B = DBG ? (4 * kQDbgSteps * 5 + 4 * 64) * 8 : 0;
    static_assert(2 * kQPatchB + 2 * BSTAGE_B + DBG_B <= 80 * 1024, "2 workgroups per CU, instrumented too");
    __shared__ __attribute__((aligned(16))) float lds[(2 * kQPatchB + 2 * BSTAGE_B + DBG_B) / 4];
    char* const ldsb = reinterpret_cast<char*>(lds);
    unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(ldsb + 2 * kQPatchB + 2 * BSTAGE_B);
    unsigned long long t_begin = 0;
    int dbg_k = 0;
    if constexpr (DBG) t_begin = __builtin_amdgcn_s_memtime();
    (void)stamps; (void)t_begin; (void)dbg_k;
#define PADEL_HQ_STAMP(J, slot)                                                                                   \
    do {                                                                                                          \
        if constexpr (DBG) {                                                                                      \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                           \
            const int real_ = (wave * kQDbgSteps + ((dbg_k + (J)) & (kQDbgSteps - 1))) * 5 + (slot);              \
            const int dummy_ = 4 * kQDbgSteps * 5 + wave * 64 + lane;                                             \
            stamps[lane == 0 ? real_ : dummy_] = t_;                                                              \
        }                                                                                                         \
    } while (0)
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

    // ---- the patch (wave 3): span s of a plane = 16 pixels x 64 bytes, lane i -> pixel 16 s + i / 4, physical 16-byte slot
    // i & 3 = logical chunk q of that pixel (hq_off), which is piece (q & 1) of group (q >> 1) of the pixel's 128 bytes
    // [h0 m0 h1 m1] in HBM; the plane's 32 bytes go in through the scalar offset
    const float* const in0 = a.in + (((long long)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * a.in_cs + a.in_choff;
    const i32x4 rsrcP = make_rsrc3(in0);
    const int p_lane = lane >> 2;
    const int p_q = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    const unsigned p_piece = (unsigned)((p_q >> 1) * 64 + (p_q & 1) * 16);
    const unsigned lp0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
    // spans S0_, S0_ + 1 of both planes (a sixth of a chunk's patch) of chunk CH_ into buffer BUF_
#define PADEL_HQ_PATCH2(CH_, BUF_, S0_)                                                                           \
    do {                                                                                                          \
        const unsigned so_ = (unsigned)(CH_) * 128u;                                                              \
        const unsigned lb_ = lp0 + (unsigned)(BUF_) * (unsigned)kQPatchB;                                         \
        int pl_ = p_lane;                                  /* recomputed per use: 12 hoisted lane offsets would spill */ \
        asm volatile("" : "+v"(pl_));                                                                             \
        PADEL_HQ_PSPAN(S0_); PADEL_HQ_PSPAN((S0_) + 1);                                                           \
    } while (0)
#define PADEL_HQ_PSPAN(S_)                                                                                        \
    do {                                                                                                          \
        const int pp_ = (S_) * 16 + pl_;                                                                          \
        const int py_ = pp_ / kQPW, px_ = pp_ - py_ * kQPW;                                                       \
        const bool ok_ = pp_ < kQNPix && (unsigned)(y0 - 1 + py_) < (unsigned)a.H && (unsigned)(x0 - 1 + px_) < (unsigned)a.W; \
        const unsigned vo_ = ok_ ? (unsigned)((py_ * a.W + px_) * a.in_cs * 4) + p_piece : kOOR3;                 \
        dma3<(S_) * 1024>(vo_, rsrcP, so_, lb_);                                                                  \
        dma3<kQPlaneB + (S_) * 1024>(vo_, rsrcP, so_ + 32u, lb_);                                                 \
    } while (0)

    // ---- weights (waves 0..2): rows of (cin / 32) * 9 k-steps x 128 bytes (h | m); wave w requests the spans 2 w, 2 w + 1
    // (16 rows x 64 bytes) of both planes: lane i -> row i / 4 of the span, physical slot i & 3
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
    unsigned lw0 = __builtin_amdgcn_readfirstlane(lp0 + (unsigned)(2 * kQPatchB) + (unsigned)wave * 2048u);
    unsigned lw1 = __builtin_amdgcn_readfirstlane(lw0 + (unsigned)BSTAGE_B);
#define PADEL_HQ_DMAB(SR_, SB_)                                                                                   \
    do {                                                                                                          \
        const unsigned lw_ = ((SR_) & 1) ? lw1 : lw0;                                                             \
        const unsigned sb_ = (SB_);                                                                               \
        dma3<0>(voffB[0], rsrcB, sb_, lw_);                                                                       \
        dma3<1024>(voffB[1], rsrcB, sb_, lw_);                                                                    \
        if constexpr (!WS) {                                                                                      \
            dma3<BPLANE_B>(voffB[0], rsrcB, sb_ + 64u, lw_);                                                      \
            dma3<BPLANE_B + 1024>(voffB[1], rsrcB, sb_ + 64u, lw_);                                               \
        }                                                                                                         \
    } while (0)
    const int ld_off = (3 * wc) * 256 + lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);      // floats
    const float* b_rd0 = lds + (2 * kQPatchB) / 4 + ld_off;
    const float* b_rd1 = b_rd0 + BSTAGE_B / 4;
    const int rd_pix = 4 * wr * kQPW + lr;                 // patch pixel of the wave's row 0, kx = 0

    f32x4 acc[MF][NF], part[MF][NF], cross[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = acc[f][j]; cross[f][j] = acc[f][j]; }
    h16x8 ah[4], am[4], wh[NF], wm[NF];       // (wm unused with WS) ah / am: input rows in 4 sliding slots (row r of the current kx in slot r & 3)
    // input row R_ (0..5 of the wave's window) at column shift KX_ into its slot
#define PADEL_HQ_READROW(R_, KX_)                                                                                 \
    do {                                                                                                          \
        const char* p_ = pbuf + hq_off(rp_ + (R_) * kQPW + (KX_), lq);                                            \
        ah[(R_) & 3] = *reinterpret_cast<const h16x8*>(p_);                                                       \
        am[(R_) & 3] = *reinterpret_cast<const h16x8*>(p_ + kQPlaneB);                                            \
    } while (0)
#define PADEL_HQ_READB(T_)                                                                                        \
    do {                                                                                                          \
        const float* const br_ = ((T_) & 1) ? b_rd1 : b_rd0;                                                      \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) {                                                          \
            wh[j] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(br_ + j * 256));                    \
            if constexpr (!WS) wm[j] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(br_ + BPLANE_B / 4 + j * 256)); \
        }                                                                                                         \
    } while (0)
    // the 9 products of output row F_ at tap row KY_ (its input row F_ + KY_ sits in slot (F_ + KY_) & 3)
#define PADEL_HQ_MFMA_ROW(F_, KY_)                                                                                \
    do {                                                                                                          \
        constexpr int s_ = ((F_) + (KY_)) & 3;                                                                    \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            cross[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], am[s_], cross[F_][j], 0, 0, 0);          \
        if constexpr (!WS) {                                                                                      \
            _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                        \
                cross[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[j], ah[s_], cross[F_][j], 0, 0, 0);      \
        }                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            part[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], ah[s_], part[F_][j], 0, 0, 0);            \
    } while (0)
    // tap step T_ = 3 kx + ky of the current chunk.  Barrier: the weights of this step (requested one step earlier) have
    // landed for every wave, the other weight stage (read one step earlier) is free for the request of step T_ + 1; at
    // T_ == 0 it also publishes the chunk's patch buffer (wave 3 waited for its DMA) and frees the other patch buffer.
    // Rows: ky == 0 reads rows 0..3 of the new column, ky == 1 row 4 (into the slot of row 0), ky == 2 row 5 (slot of row 1).
    // Requests go out under the first row's MFMAs: waves 0..2 the next step's weights, wave 3 a sixth of the next chunk's
    // patch per tap 0..5.
#define PADEL_HQ_STEP(T_)                                                                                         \
    do {                                                                                                          \
        constexpr int kx_ = h2_tap_kx(T_), ky_ = h2_tap_ky(T_);                                                   \
        int rp_ = rd_pix;                                  /* row addresses recomputed per tap (3 VALU each): 18 hoisted ones would spill */ \
        asm volatile("" : "+v"(rp_));                                                                             \
        PADEL_HQ_STAMP(T_, 0);                                                                                    \
        if constexpr ((T_) > 0) {                          /* the planes are static inside a chunk: read under the wait */ \
            if constexpr (ky_ == 0) { PADEL_HQ_READROW(0, kx_); PADEL_HQ_READROW(1, kx_); PADEL_HQ_READROW(2, kx_); PADEL_HQ_READROW(3, kx_); } \
            else PADEL_HQ_READROW(3 + ky_, kx_);                                                                  \
        }                                                                                                         \
        if ((T_) == 0 || wave != 3) wait_vm3<0>();                                                                \
        PADEL_HQ_STAMP(T_, 1);                                                                                    \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
        PADEL_HQ_STAMP(T_, 2);                                                                                    \
        PADEL_HQ_READB(T_);                                                                                       \
        if constexpr ((T_) == 0) { PADEL_HQ_READROW(0, 0); PADEL_HQ_READROW(1, 0); PADEL_HQ_READROW(2, 0); PADEL_HQ_READROW(3, 0); } \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if constexpr (DBG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PADEL_HQ_STAMP(T_, 3); }          \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        PADEL_HQ_MFMA_ROW(0, ky_);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if (wave != 3) {                                                                                          \
            if ((T_) < 8 || c + 1 < nch) PADEL_HQ_DMAB((T_) + 1, s_kb + ((T_) + 1) * 128u);                       \
        } else if ((T_) < 6 && c + 1 < nch) {                                                                     \
            PADEL_HQ_PATCH2(c + 1, (c + 1) & 1, 2 * ((T_) < 6 ? (T_) : 0));                                       \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_HQ_MFMA_ROW(1, ky_); PADEL_HQ_MFMA_ROW(2, ky_); PADEL_HQ_MFMA_ROW(3, ky_);                          \
        __builtin_amdgcn_s_setprio(0);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_HQ_STAMP(T_, 4);                                                                                    \
    } while (0)

    unsigned s_kb = 0;
    if (wave == 3) {
        PADEL_HQ_PATCH2(0, 0, 0); PADEL_HQ_PATCH2(0, 0, 2); PADEL_HQ_PATCH2(0, 0, 4);
        PADEL_HQ_PATCH2(0, 0, 6); PADEL_HQ_PATCH2(0, 0, 8); PADEL_HQ_PATCH2(0, 0, 10);
    } else {
        PADEL_HQ_DMAB(0, 0u);
    }
    for (int c = 0; c < nch; ++c) {
        const char* const pbuf = ldsb + (c & 1) * kQPatchB;
        PADEL_HQ_STEP(0); PADEL_HQ_STEP(1); PADEL_HQ_STEP(2); PADEL_HQ_STEP(3); PADEL_HQ_STEP(4);
        PADEL_HQ_STEP(5); PADEL_HQ_STEP(6); PADEL_HQ_STEP(7); PADEL_HQ_STEP(8);
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        { const float* t_ = b_rd0; b_rd0 = b_rd1; b_rd1 = t_; const unsigned u_ = lw0; lw0 = lw1; lw1 = u_; }
        s_kb += 9u * 128u;
        dbg_k += 9;
    }
    wait_vm3<0>();
#undef PADEL_HQ_STEP
#undef PADEL_HQ_MFMA_ROW
#undef PADEL_HQ_READB
#undef PADEL_HQ_READROW
#undef PADEL_HQ_DMAB
#undef PADEL_HQ_PSPAN
#undef PADEL_HQ_PATCH2

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
    if constexpr (DBG) {
        if (a.dbg) {
            const unsigned long long t_end = __builtin_amdgcn_s_memtime();
            __syncthreads();
            unsigned long long* d = a.dbg + (long long)blockIdx.x * kQDbgWords;
            for (int i = tid; i < 4 * kQDbgSteps * 5; i += 256) d[8 + i] = stamps[i];
            if (lane == 0) {
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
                const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
                d[wave] = ((unsigned long long)xcc << 32) | hw;
                if (wave == 0) { d[4] = t_begin; d[5] = t_end; d[6] = (unsigned long long)(nch * 9); d[7] = (unsigned long long)bid; }
            }
        }
    }
#undef PADEL_HQ_STAMP
}

bool conv_h2q_supported(const ConvArgs& a) {
    return a.ksize == 3 && a.stride == 1 && (a.cin & 31) == 0 && a.cin >= 32 && a.Ho == a.H && a.Wo == a.W && a.w != nullptr && !a.in2;
}

hipError_t launch_conv_h2q(const ConvArgs& a_in, hipStream_t s) {
    if (!conv_h2q_supported(a_in)) return hipErrorNotSupported;
    ConvArgs a = a_in;
    const int batch = a.M / (a.Ho * a.Wo);
    a.n_mtiles = batch * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
    a.n_ntiles = (a.n16 + 5) / 6;
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
#ifdef PADEL_H2P_PROBES
    if (a.dbg) {
        hipLaunchKernelGGL((conv_h2q_kernel<3, true>), grid, dim3(256), 0, s, a);
        return hipGetLastError();
    }
#endif
    if (a.w_single) hipLaunchKernelGGL((conv_h2q_kernel<3, false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_h2q_kernel<3>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace padel
