// K4s "h2 1x1, register weights" (round 6) — the 1x1 convolutions of h2 graphs, two-product layers (PA_CONV_W_SINGLE) with
// cin % 32 == 0 and no absorbed upsample: the C2f cv1 / cv2, SPPF and head 1x1 layers, 17 ms of the bench's step at 120-375 TFLOP/s
// fp32-equivalent and 2.5-4.9 TB/s — neither at the matrix roof nor at the HBM roof.
//
// A 1x1 layer streams its activation tile once per k-step (no tap reuse): what it needs is requests far enough ahead and nothing
// else between its products.  The tap kernels (conv_tap_h2p.hip: 128 x 96 tile, 4 waves of 2 x 6 fragments) keep two activation
// steps in flight and spend a ring wait, six weight reads and a barrier per 24-MFMA k-step.  Here
//   * a workgroup owns 128 pixels x 96 channels as 2 x 2 waves of 4 x 3 fragments (64 pixels x 48 channels: 3 KB of weights per
//     wave and k-step instead of 6 — the two waves of a channel half share them in the L1);
//   * the WEIGHTS never touch LDS: global -> VGPR from the operand-order copy ([fragment][k-step][h | m][lane][16 B],
//     conv_patch_h2r.hip), two k-steps ahead through three register sets;
//   * the LDS that frees (80 KB per workgroup, two per CU) is a FIVE-stage activation ring of single k-steps (128 pixels x 32
//     channels x 4 B = 16 KB per stage): the request of step J + 4 goes out behind the barrier of step J — four steps (64 KB per
//     workgroup, 128 KB per CU) in flight; the per-step barrier stays (it publishes the stage), but all that sits in front of it is
//     the wave's own counted wait.
// (First version, measured: 32 KB double steps in a two-stage ring, one barrier per 48 MFMAs — 363 vs 363 and 266 vs 276 TFLOP/s on
//  1152 -> 384 / 576 -> 192 against the tap tile: one double step in flight does not cover the memory latency; depth, not barrier
//  count, is what these layers need.)
// The M tail needs no lane masks: the tile's buffer descriptor ends at the tensor's end (out-of-range lanes read zeros); the lane
// offsets are constants of the kernel.
// Accumulation as in the tap kernels — cross: wh am per k-step; main: wh ah in blocks of 9 k-steps, flushed into acc — so results are
// bitwise those of conv_h2_1_kernel / conv_h2_1p_kernel.
#include "h2_common.h"

namespace padel {

namespace {

__device__ __forceinline__ unsigned hs_off(int p, int q) { return (unsigned)(p * 64 + ((q ^ (((p >> 2) & 1) << 1)) << 4)); }

typedef int hs_i32x4 __attribute__((ext_vector_type(4)));

}  // namespace

// WC = waves along the channels: 2 (128 x 96 tile, 4 waves, 2 workgroups per CU, 5-stage ring) or 4 (128 x 192 tile, 8 waves, ONE
// workgroup per CU, 9-stage ring): what bounds a long-K 1x1 layer is the request stream of its activation tile — every 96-channel
// tile of a pixel tile requests the same 16 KB per k-step again (7.4 TB/s of requests on 1152 -> 384) —, and with all MFMAs compiled
// out (PROBE) the kernel is exactly as fast: 265 vs 264 / 309 vs 331 TFLOP/s (profiles/r6B_1x1_probe.txt).  Twice the channels per
// workgroup is half the requests.
// PROBE (tuning word bit 6, WRONG results; the eight-wave tile only beyond 1): 1 = no MFMAs and no operand reads — what the two
// request streams alone sustain; 2 (+ bit 8): the activation requests alone; 3 (+ bit 9): the weight loads alone; 4 (+ bits 8, 9):
// activations alone as whole 128-byte records; 5 (+ bit 10): activations alone in address order; 6 (+ bit 11): activations alone as
// plain register loads; 7 (+ bit 12): both streams, one wave of each pair answered by a zero-record descriptor
// D = how many k-steps ahead a wave loads its weights (D + 1 register sets).  A wave's memory requests complete IN ORDER (one vmcnt
// counter): waiting for W(K), issued D steps ago, also waits for every activation request issued before it — the ring's requests
// have min(ring depth, D + 1) steps to arrive, not the ring depth: D = 2 leaves an eight-step ring a three-step window.  Measured
// (profiles/r6I_1x1_weight_lookahead.txt): D = 4 (five-step window, 246 VGPRs) is 0..1 % SLOWER on every long-K layer, D = 3 on the
// 128 x 96 tile likewise — the window is not what bounds these layers; D = 2 stays, the other is tuning bit 3.
// What does (profiles/r6J_1x1_stream_probes.txt, 1152 -> 384, TFLOP/s-equivalent of the kernel's time): whole kernel 405; both
// request streams without MFMAs 472; the activation requests alone 596 — and the same 590..630 whether they ask for half lines or
// whole 128-byte records, walk the tile in address order, or are plain register loads instead of LDS-DMA; the weight loads alone
// 1 380; half of the weight loads answered by zero-record descriptors: 475.  1 / 596 + 1 / 1380 = 1 / 416: the two streams nearly
// ADD, MFMAs hide under them, and neither the L2 (50 % hits, 6 TB/s of 34) nor HBM (3 + 1 TB/s read + write: activations are
// fetched once, TCC counters) is at its roof — the per-CU vector-memory path serialises them.  The activation-only numbers do not depend
// on how many steps the probe leaves in flight behind its wait either (2 / 4 / 7: 600 / 617 / 592; -DPADEL_HS_AWIN).
// WR = waves along the pixels: 2 (128-pixel tiles) or 1 (WC = 4 only: 64 x 192 tiles of FOUR waves, two workgroups per CU with
// nine 8 KB stages each — the bytes of the 128 x 192 tile per product, but two independent barrier domains per CU)
#ifndef PADEL_HS_AWIN
#define PADEL_HS_AWIN D        // activation-only probes: steps that may stay in flight behind the wait (the real kernel's queue leaves D)
#endif
template <int WC, int PROBE = 0, int D = 2, int WR = 2>
__global__ void __launch_bounds__(64 * WR * WC, 2) conv_h2s_kernel(const ConvArgs a) {
    constexpr int MF = 4, NF = 3, NSET = D + 1;
    constexpr int kSPlaneB = 64 * WR * 64;         // one fp16 plane of a 32-channel chunk: 64 WR pixels x 64 B
    constexpr int kSStageB = 2 * kSPlaneB;         // h | m of one k-step
    constexpr int kSStages = WC == 2 ? 5 : 9;
    constexpr int kSAhead = kSStages - 1;          // request distance in k-steps
    constexpr int NA = 8 / (2 * WC) * 2;           // activation requests per wave and k-step: 8 WR spans x planes over WR x WC waves
    static_assert(WR == 2 || WC == 4, "64-pixel tiles: one span per wave");
    __shared__ __attribute__((aligned(16))) float lds[(kSStages * kSStageB) / 4];
    char* const ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = WR == 2 ? (wave & 1) : 0, wc = WR == 2 ? (wave >> 1) : wave;      // pixel half (64 pixels), channel part (3 fragments)
    const int lr = lane & 15, lq = lane >> 4;

    // XCD-aware 1-D tile map: the channel tiles of one pixel tile are neighbours on one XCD (they share its activations in the L2)
    const int nmt = a.n_mtiles, nnt = a.n_ntiles;
    const int bid = blockIdx.x;
    const int q8 = nmt >> 3, r8 = nmt & 7, xcd = bid & 7, idx = bid >> 3;
    const int mloc = idx / nnt, nt = idx - mloc * nnt;
    if (mloc >= q8 + (xcd < r8 ? 1 : 0)) return;
    const int mt = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + mloc;
    const int m0 = mt * (64 * WR);
    const int f0 = nt * WC * NF;
    const int nch = a.cin >> 5;                   // k-steps

    // ---- activations: span s of a plane = 16 pixels x 64 bytes, lane i -> pixel 16 s + i / 4, physical 16-byte slot i & 3 = logical
    // chunk q of that pixel, which is piece (q & 1) of group (q >> 1) of the pixel's 128 bytes [h0 m0 h1 m1] of the k-step in HBM; the
    // m plane's 32 bytes go in through the scalar offset.  Wave w requests spans w and w + 4 of both planes of a stage.
    // The descriptor starts at the tile's first pixel and ends with the tensor: the M tail reads zeros.
    const long long pix_b = (long long)a.in_cs * 4;
    const char* const in0 = reinterpret_cast<const char*>(a.in + a.in_choff) + (long long)m0 * pix_b;
    i32x4 rsrcA = make_rsrc3(in0);
    {
        const long long left = ((long long)a.M - m0) * pix_b;
        rsrcA[2] = (int)(left > 0x7FFFFFFFll ? 0x7FFFFFFFll : left);
    }
    const int p_q = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    const unsigned p_piece = (unsigned)((p_q >> 1) * 64 + (p_q & 1) * 16);
    unsigned voA[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) voA[k] = (unsigned)(((wave + 4 * k) * 16 + (lane >> 2)) * (unsigned)pix_b) + p_piece;     // (WC = 4: one span per wave)
    unsigned voL[2];                              // PROBE 4: whole 128-byte records, 8 pixels per request (wrong LDS layout)
#pragma unroll
    for (int k = 0; k < 2; ++k) voL[k] = (unsigned)((wave * 16 + 8 * k + (lane >> 3)) * (unsigned)pix_b) + (unsigned)(lane & 7) * 16u;
    hs_i32x4 sink[2] = {};                        // PROBE 6: the activation requests as plain register loads (no LDS-DMA)
    unsigned voQ[2];                              // PROBE 5: the tile's bytes in address order — 16 KB contiguous per k-step (wrong data)
#pragma unroll
    for (int k = 0; k < 2; ++k) voQ[k] = (unsigned)(wave * 2048 + k * 1024 + lane * 16);
    const unsigned lp0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
    const unsigned lpw = __builtin_amdgcn_readfirstlane(lp0 + (unsigned)wave * 1024u);
    // k-step K_ into the stage at byte offset SB_; k-steps beyond the last go through a descriptor of zero records (the request
    // count per step — and with it every counted wait — is static)
#define PADEL_HS_REQA(SB_, K_)                                                                                    \
    if constexpr (PROBE != 3) {                                                                                   \
        const unsigned so_ = (unsigned)(K_) * 128u;                                                               \
        const unsigned lb_ = lpw + (unsigned)(SB_);                                                               \
        i32x4 rs_ = rsrcA;                                                                                        \
        if ((int)(K_) >= nch) rs_[2] = 0;                                                                         \
        if constexpr (PROBE == 4) { dma3<0>(voL[0], rs_, so_, lb_); dma3<kSPlaneB>(voL[1], rs_, so_, lb_); }      \
        else if constexpr (PROBE == 6) {                                                                          \
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(sink[0]) : "v"(voA[0]), "s"(rs_), "s"(so_) : "memory"); \
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(sink[1]) : "v"(voA[0]), "s"(rs_), "s"(so_ + 32u) : "memory"); \
        }                                                                                                         \
        else if constexpr (PROBE == 5) { dma3<0>(voQ[0], rs_, so_ * 128u, lb_); dma3<kSPlaneB>(voQ[1], rs_, so_ * 128u, lb_); } \
        else {                                                                                                    \
        dma3<0>(voA[0], rs_, so_, lb_); if constexpr (WC == 2) dma3<4096>(voA[1], rs_, so_, lb_);                 \
        dma3<kSPlaneB>(voA[0], rs_, so_ + 32u, lb_); if constexpr (WC == 2) dma3<kSPlaneB + 4096>(voA[1], rs_, so_ + 32u, lb_); \
        }                                                                                                         \
    }

    // ---- weights: a.wr = [fragment][k-step][h | m][lane][16 bytes]; NF requests per wave and k-step
    const unsigned fragb = (unsigned)nch * 2048u;
    const unsigned voffW = (unsigned)lane * 16u;
    i32x4 rsrcW[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int frag = min(f0 + NF * wc + j, a.n16 - 1);   // fragments beyond the matrix: any valid rows (never stored)
        rsrcW[j] = make_rsrc3(reinterpret_cast<const char*>(a.wr) + (long long)frag * fragb);
    }
    hs_i32x4 w[NSET][NF];
    // (k-steps beyond the last read the next fragment's first k-steps, or the slack behind the copy — never multiplied)
#define PADEL_HS_LOADW(SET_, K_)                                                                                  \
    if constexpr (PROBE != 2 && PROBE < 4 || PROBE == 7) {                                                        \
        const unsigned so_ = (unsigned)(K_) * 2048u;                                                              \
        if (PROBE == 7 && wr == 1) rsrcW[0][2] = rsrcW[1][2] = rsrcW[2][2] = 0;     /* one wave of a pair loads */ \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(w[SET_][j]) : "v"(voffW), "s"(rsrcW[j]), "s"(so_) : "memory"); \
    }
#define PADEL_HS_WAITW(SET_, N_)                                                                                  \
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(w[SET_][0]), "+v"(w[SET_][1]), "+v"(w[SET_][2]) : "n"(PROBE == 2 || (PROBE >= 4 && PROBE != 7) ? ((N_) == 0 ? 0 : (PADEL_HS_AWIN) * NA) : PROBE == 3 ? (N_) * NF / (NF + NA) : (N_)) : "memory")

    // ---- operand reads: pixel fragment f of the wave = pixels 64 wr + 16 f + lr: hs_off's swizzle depends on lr only, everything
    // else is the stage's offset (scalar, walks the ring) and an immediate
    const unsigned abase = hs_off(64 * wr + lr, lq);
    f32x4 acc[MF][NF], part[MF][NF], cross[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = acc[f][j]; cross[f][j] = acc[f][j]; }
    h16x8 ah[MF], am[MF];
#define PADEL_HS_MFMA(F_, SET_)                                                                                   \
    do {                                                                                                          \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            cross[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, w[SET_][j]), am[F_], cross[F_][j], 0, 0, 0); \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            part[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, w[SET_][j]), ah[F_], part[F_][j], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
    // k-step K_ (weight set SET_ = K_ % NSET, statically: the loop is unrolled by NSET; the stage walks the ring through s_rd / s_wr).
    // Queue of the wave behind W(K_) when it waits (D = 2, WC = 2): A(K_ + 2) x 4, W(K_ + 1) x 3, A(K_ + 3) x 4, W(K_ + 2) x 3 — 14
    // requests = D x (NF + NA); in front of it, in order: A(K_), W(K_ - 1), A(K_ + 1).
#define PADEL_HS_STEP(K_, SET_)                                                                                   \
    do {                                                                                                          \
        PADEL_HS_LOADW(((SET_) + D) % NSET, (K_) + D);                                                            \
        PADEL_HS_WAITW(SET_, D * (NF + NA));                                                                      \
        __builtin_amdgcn_s_barrier();             /* stage s_rd is published; the stage read in step K_ - 1 (= s_wr) is free */ \
        asm volatile("" ::: "memory");                                                                            \
        PADEL_HS_REQA(s_wr, (K_) + kSAhead);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if constexpr (PROBE == 0) {                                                                               \
            const char* p_ = ldsb + abase + s_rd;                                                                 \
            _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                      \
                ah[f] = *reinterpret_cast<const h16x8*>(p_ + f * 1024);                                           \
                am[f] = *reinterpret_cast<const h16x8*>(p_ + f * 1024 + kSPlaneB);                                \
            }                                                                                                     \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if constexpr (PROBE == 0) { PADEL_HS_MFMA(0, SET_); PADEL_HS_MFMA(1, SET_); PADEL_HS_MFMA(2, SET_); PADEL_HS_MFMA(3, SET_); } \
        if (++kblk == 9) {                        /* main sums in blocks of 9 k-steps (the tap kernels' accumulation blocks) */ \
            kblk = 0;                                                                                             \
            _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                        \
                _Pragma("unroll") for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; } \
        }                                                                                                         \
        s_wr = s_rd;                                                                                              \
        s_rd = s_rd + (unsigned)kSStageB == (unsigned)(kSStages * kSStageB) ? 0u : s_rd + (unsigned)kSStageB;     \
    } while (0)

    int kblk = 0;
    unsigned s_rd = 0, s_wr = (unsigned)((kSStages - 1) * kSStageB);      // stage of the current step / the stage the previous step read
    // prologue — the order the steady state leaves: A(0) .. A(kSAhead - D - 1), then W(i), A(kSAhead - D + i) for i < D
    static_assert(kSAhead >= D, "the ring is at least as deep as the weight look-ahead");
#pragma unroll
    for (int k = 0; k < kSAhead - D; ++k) PADEL_HS_REQA(k * kSStageB, k);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        PADEL_HS_LOADW(i, i);
        PADEL_HS_REQA((kSAhead - D + i) * kSStageB, kSAhead - D + i);
    }
#pragma unroll 1
    for (int k = 0; k < nch; k += NSET) {
        PADEL_HS_STEP(k, 0);
#pragma unroll
        for (int i = 1; i < NSET; ++i)
            if (k + i < nch) PADEL_HS_STEP(k + i, i);
    }
    if (kblk != 0) {
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[f][j] += part[f][j];
    }
    // the look-ahead requests behind the last step (zero-record activations, weights nobody uses) before the LDS is released and
    // before the epilogue may reuse the weight registers
#pragma unroll
    for (int i = 0; i < NSET; ++i) PADEL_HS_WAITW(i, 0);
    if constexpr (PROBE == 6) asm volatile("" :: "v"(sink[0]), "v"(sink[1]));
#undef PADEL_HS_STEP
#undef PADEL_HS_MFMA
#undef PADEL_HS_WAITW
#undef PADEL_HS_LOADW
#undef PADEL_HS_REQA

    int mpix[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int m = m0 + 64 * wr + 16 * f + lr;
        mpix[f] = m < a.M ? m : -1;
    }
    const int fw = f0 + NF * wc;
    const bool fast = m0 + 64 * WR <= a.M && (fw + NF) * 16 <= a.cout && (((a.out_choff | a.out_cs) & 3) == 0) &&
                      (!a.res || (((a.res_choff | a.res_cs) & 3) == 0));
    if (fw < a.n16) h2_epilogue<MF, NF>(a, acc, cross, mpix, fw, lq, fast);
}

// ===================================================================================================== stride-2 3x3 (tile 246)
// The stride-2 3x3 layers (L3 / L4 / L5 of the backbone, the PAN's downsampling convs: 6.1 ms of the bench's step on the tap kernel at
// MfmaUtil 0.34) have the 1x1 layers' disease nine times over: the tap walk requests a 16 KB activation tile per TAP and 96-channel
// tile — 7.4 TB/s of LDS-DMA requests on 96 -> 192, the chip's fill rate.  The same machine as conv_h2s_kernel<4>: 128 output pixels x
// 192 channels per workgroup (8 waves, one workgroup per CU, nine 16 KB stages), register weights, k-step = (32-channel chunk, tap) with
// the taps column-major like every h2 kernel and the main sums flushed per chunk (= the tap kernels' blocks of 9 k-steps): bitwise
// conv_h2_kernel.  A lane's request offset is its pixel's top-left input pixel relative to the tile's; the tap enters through the
// scalar offset, taps outside the image through a per-lane validity mask (3 row bits + 3 column bits).
// WR = 1: 64 x 192 tiles of four waves, two workgroups per CU (nine 8 KB stages each), as conv_h2s_kernel<4, .., 1>
template <int PROBE = 0, int WR = 2>
__global__ void __launch_bounds__(256 * WR, 2) conv_h2s3_kernel(const ConvArgs a) {
    constexpr int MF = 4, NF = 3, WC = 4;
    constexpr int kSPlaneB = 64 * WR * 64, kSStageB = 2 * kSPlaneB;
    constexpr int kSStages = 9, kSAhead = 8, NA = 2;
    __shared__ __attribute__((aligned(16))) float lds[(kSStages * kSStageB) / 4];
    char* const ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = WR == 2 ? (wave & 1) : 0, wc = WR == 2 ? (wave >> 1) : wave;
    const int lr = lane & 15, lq = lane >> 4;

    const int nmt = a.n_mtiles, nnt = a.n_ntiles;
    const int bid = blockIdx.x;
    const int q8 = nmt >> 3, r8 = nmt & 7, xcd = bid & 7, idx = bid >> 3;
    const int mloc = idx / nnt, nt = idx - mloc * nnt;
    if (mloc >= q8 + (xcd < r8 ? 1 : 0)) return;
    const int mt = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + mloc;
    const int m0 = mt * (64 * WR);
    const int f0 = nt * WC * NF;
    const int nch = a.cin >> 5;
    const int HoWo = a.Ho * a.Wo;

    // ---- activations.  Output pixel m -> (n, oy, ox); its window starts at input pixel (2 oy - 1, 2 ox - 1); linear input pixel
    // relative to the tile's first one (monotone in m: >= 0).  Wave w requests span w (16 pixels) of both planes of a stage.
    const long long pix_b = (long long)a.in_cs * 4;
    long long lin0;
    {
        const int n = fastdiv3(m0, a.howo_magic, a.howo_shift);
        const int rem = m0 - n * HoWo;
        const int oy = fastdiv3(rem, a.wo_magic, a.wo_shift), ox = rem - oy * a.Wo;
        lin0 = ((long long)n * a.H + (2 * oy - 1)) * a.W + (2 * ox - 1);
    }
    const i32x4 rsrcA = make_rsrc3(reinterpret_cast<const char*>(a.in + a.in_choff) + lin0 * pix_b);
    const int p_q = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    const unsigned p_piece = (unsigned)((p_q >> 1) * 64 + (p_q & 1) * 16);
    unsigned voA, vbits = 0;
    {
        const int m = m0 + wave * 16 + (lane >> 2);
        const bool mv = m < a.M;
        const int mc = mv ? m : m0;
        const int n = fastdiv3(mc, a.howo_magic, a.howo_shift);
        const int rem = mc - n * HoWo;
        const int oy = fastdiv3(rem, a.wo_magic, a.wo_shift), ox = rem - oy * a.Wo;
        const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
        const long long lin = ((long long)n * a.H + iy0) * a.W + ix0;
        voA = (unsigned)((lin - lin0) * pix_b) + p_piece;
        if (mv) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                vbits |= ((unsigned)(iy0 + k) < (unsigned)a.H ? 1u : 0u) << k;
                vbits |= ((unsigned)(ix0 + k) < (unsigned)a.W ? 8u : 0u) << k;
            }
        }
    }
    const unsigned rowb = (unsigned)a.W * (unsigned)pix_b;
    const unsigned lp0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
    const unsigned lpw = __builtin_amdgcn_readfirstlane(lp0 + (unsigned)wave * 1024u);
    // tap TAP_ (column-major: ky = TAP_ % 3, kx = TAP_ / 3) of chunk CH_ into the stage at byte offset SB_; chunks beyond the last go
    // through a descriptor of zero records
#define PADEL_HS3_REQA(SB_, CH_, TAP_)                                                                            \
    do {                                                                                                          \
        constexpr int ky_ = (TAP_) % 3, kx_ = (TAP_) / 3;                                                         \
        constexpr unsigned need_ = (1u << ky_) | (8u << kx_);                                                     \
        const unsigned so_ = (unsigned)(CH_) * 128u + (unsigned)ky_ * rowb + (unsigned)kx_ * (unsigned)pix_b;     \
        const unsigned lb_ = lpw + (unsigned)(SB_);                                                               \
        i32x4 rs_ = rsrcA;                                                                                        \
        if ((int)(CH_) >= nch) rs_[2] = 0;                                                                        \
        const unsigned vo_ = (vbits & need_) == need_ ? voA : kOOR3;                                              \
        dma3<0>(vo_, rs_, so_, lb_); dma3<kSPlaneB>(vo_, rs_, so_ + 32u, lb_);                                    \
    } while (0)

    // ---- weights: a.wr = [fragment][k-step][h | m][lane][16 bytes], k-step = chunk * 9 + tap
    const unsigned fragb = (unsigned)(nch * 9) * 2048u;
    const unsigned voffW = (unsigned)lane * 16u;
    i32x4 rsrcW[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int frag = min(f0 + NF * wc + j, a.n16 - 1);
        rsrcW[j] = make_rsrc3(reinterpret_cast<const char*>(a.wr) + (long long)frag * fragb);
    }
    hs_i32x4 w[3][NF];
    unsigned s_kw = 0;                            // byte offset of the current chunk's first k-step inside a fragment
#define PADEL_HS3_LOADW(SET_, TT_)                                                                                \
    do {                                                                                                          \
        const unsigned so_ = s_kw + (unsigned)((TT_) * 2048);                                                     \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(w[SET_][j]) : "v"(voffW), "s"(rsrcW[j]), "s"(so_) : "memory"); \
    } while (0)
#define PADEL_HS3_WAITW(SET_, N_)                                                                                 \
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(w[SET_][0]), "+v"(w[SET_][1]), "+v"(w[SET_][2]) : "n"(N_) : "memory")

    const unsigned abase = hs_off(64 * wr + lr, lq);
    f32x4 acc[MF][NF], part[MF][NF], cross[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = acc[f][j]; cross[f][j] = acc[f][j]; }
    h16x8 ah[MF], am[MF];
#define PADEL_HS3_MFMA(F_, SET_)                                                                                  \
    do {                                                                                                          \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            cross[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, w[SET_][j]), am[F_], cross[F_][j], 0, 0, 0); \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            part[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, w[SET_][j]), ah[F_], part[F_][j], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
    // tap T_ of chunk c (weight set T_ % 3; the request goes out for the step 8 ahead: tap T_ - 1 of chunk c + 1, or tap 8 of chunk c)
#define PADEL_HS3_STEP(T_)                                                                                        \
    do {                                                                                                          \
        PADEL_HS3_LOADW(((T_) + 2) % 3, (T_) + 2);                                                                \
        PADEL_HS3_WAITW((T_) % 3, 2 * NF + 2 * NA);                                                               \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
        PADEL_HS3_REQA(s_wr, c + ((T_) >= 1 ? 1 : 0), ((T_) + 8) % 9);                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if constexpr (PROBE == 0) {                                                                               \
            const char* p_ = ldsb + abase + s_rd;                                                                 \
            _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                      \
                ah[f] = *reinterpret_cast<const h16x8*>(p_ + f * 1024);                                           \
                am[f] = *reinterpret_cast<const h16x8*>(p_ + f * 1024 + kSPlaneB);                                \
            }                                                                                                     \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if constexpr (PROBE == 0) { PADEL_HS3_MFMA(0, (T_) % 3); PADEL_HS3_MFMA(1, (T_) % 3); PADEL_HS3_MFMA(2, (T_) % 3); PADEL_HS3_MFMA(3, (T_) % 3); } \
        s_wr = s_rd;                                                                                              \
        s_rd = s_rd + (unsigned)kSStageB == (unsigned)(kSStages * kSStageB) ? 0u : s_rd + (unsigned)kSStageB;     \
    } while (0)

    unsigned s_rd = 0, s_wr = (unsigned)((kSStages - 1) * kSStageB);
    // prologue: steps 0..7 = taps 0..7 of chunk 0; behind W(0): A(2), W(1), A(3)
    PADEL_HS3_REQA(0 * kSStageB, 0, 0); PADEL_HS3_REQA(1 * kSStageB, 0, 1);
    PADEL_HS3_REQA(4 * kSStageB, 0, 4); PADEL_HS3_REQA(5 * kSStageB, 0, 5); PADEL_HS3_REQA(6 * kSStageB, 0, 6); PADEL_HS3_REQA(7 * kSStageB, 0, 7);
    PADEL_HS3_LOADW(0, 0);
    PADEL_HS3_REQA(2 * kSStageB, 0, 2);
    PADEL_HS3_LOADW(1, 1);
    PADEL_HS3_REQA(3 * kSStageB, 0, 3);
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        PADEL_HS3_STEP(0); PADEL_HS3_STEP(1); PADEL_HS3_STEP(2); PADEL_HS3_STEP(3); PADEL_HS3_STEP(4);
        PADEL_HS3_STEP(5); PADEL_HS3_STEP(6); PADEL_HS3_STEP(7); PADEL_HS3_STEP(8);
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        s_kw += 9u * 2048u;
    }
    PADEL_HS3_WAITW(0, 0); PADEL_HS3_WAITW(1, 0); PADEL_HS3_WAITW(2, 0);
#undef PADEL_HS3_STEP
#undef PADEL_HS3_MFMA
#undef PADEL_HS3_WAITW
#undef PADEL_HS3_LOADW
#undef PADEL_HS3_REQA

    int mpix[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int m = m0 + 64 * wr + 16 * f + lr;
        mpix[f] = m < a.M ? m : -1;
    }
    const int fw = f0 + NF * wc;
    const bool fast = m0 + 64 * WR <= a.M && (fw + NF) * 16 <= a.cout && (((a.out_choff | a.out_cs) & 3) == 0) &&
                      (!a.res || (((a.res_choff | a.res_cs) & 3) == 0));
    if (fw < a.n16) h2_epilogue<MF, NF>(a, acc, cross, mpix, fw, lq, fast);
}

bool conv_h2s3_supported(const ConvArgs& a) {
    return a.w_single && a.wr && a.ksize == 3 && a.stride == 2 && (a.cin & 31) == 0 && a.cin >= 32 && a.w != nullptr && !a.in2 &&
           a.Ho <= (a.H + 1) / 2 && a.Wo <= (a.W + 1) / 2 && (long long)4 * a.W * a.in_cs * 4 < 0x3FFFFFFFll;
}

hipError_t launch_conv_h2s3(const ConvArgs& a_in, hipStream_t s, bool m64) {
    if (!conv_h2s3_supported(a_in)) return hipErrorNotSupported;
    ConvArgs a = a_in;
    a.n_mtiles = m64 ? (a.M + 63) / 64 : (a.M + 127) / 128;
    a.n_ntiles = (a.n16 + 11) / 12;
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    if (m64) {
        if (a.tune & 64) hipLaunchKernelGGL((conv_h2s3_kernel<1, 1>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_h2s3_kernel<0, 1>), grid, dim3(256), 0, s, a);
        return hipGetLastError();
    }
    if (a.tune & 64) hipLaunchKernelGGL((conv_h2s3_kernel<1>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((conv_h2s3_kernel<0>), grid, dim3(512), 0, s, a);
    return hipGetLastError();
}

bool conv_h2s_supported(const ConvArgs& a) {
    return a.w_single && a.wr && a.ksize == 1 && a.stride == 1 && (a.cin & 31) == 0 && a.cin >= 64 && a.Ho == a.H && a.Wo == a.W && a.w != nullptr &&
           !a.in2 && (long long)128 * a.in_cs * 4 < 0x7FFFFFFFll;
}

hipError_t launch_conv_h2s(const ConvArgs& a_in, bool nf12, hipStream_t s, bool m64) {
    if (!conv_h2s_supported(a_in)) return hipErrorNotSupported;
    ConvArgs a = a_in;
    if (nf12 && m64) {                          // 64 x 192 tiles: four waves, two workgroups per CU
        a.n_mtiles = (a.M + 63) / 64;
        a.n_ntiles = (a.n16 + 11) / 12;
        dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
        if (a.tune & 64) hipLaunchKernelGGL((conv_h2s_kernel<4, 1, 2, 1>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_h2s_kernel<4, 0, 2, 1>), grid, dim3(256), 0, s, a);
        return hipGetLastError();
    }
    a.n_mtiles = (a.M + 127) / 128;
    const bool wide = nf12;                     // 128 x 192 tiles: 8 waves, one workgroup per CU
    a.n_ntiles = wide ? (a.n16 + 11) / 12 : (a.n16 + 5) / 6;
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    if (wide) {
        static bool attr = false;
        if (!attr) {                                // 144 KB of static LDS
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2s_kernel<4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 0);
            attr = true;
        }
        if ((a.tune & 64) && (a.tune & 4096)) hipLaunchKernelGGL((conv_h2s_kernel<4, 7>), grid, dim3(512), 0, s, a);               // both streams, one wave of a pair loads weights
        else if ((a.tune & 64) && (a.tune & 2048)) hipLaunchKernelGGL((conv_h2s_kernel<4, 6>), grid, dim3(512), 0, s, a);               // activations alone, to registers
        else if ((a.tune & 64) && (a.tune & 1024)) hipLaunchKernelGGL((conv_h2s_kernel<4, 5>), grid, dim3(512), 0, s, a);               // activations alone, in address order
        else if ((a.tune & 64) && (a.tune & 256) && (a.tune & 512)) hipLaunchKernelGGL((conv_h2s_kernel<4, 4>), grid, dim3(512), 0, s, a);  // activations alone, whole records
        else if ((a.tune & 64) && (a.tune & 256)) hipLaunchKernelGGL((conv_h2s_kernel<4, 2>), grid, dim3(512), 0, s, a);      // activations only
        else if ((a.tune & 64) && (a.tune & 512)) hipLaunchKernelGGL((conv_h2s_kernel<4, 3>), grid, dim3(512), 0, s, a); // weights only
        else if (a.tune & 64) hipLaunchKernelGGL((conv_h2s_kernel<4, 1>), grid, dim3(512), 0, s, a);
        else if (a.tune & 8) hipLaunchKernelGGL((conv_h2s_kernel<4, 0, 4>), grid, dim3(512), 0, s, a);        // weights four steps ahead: a five-step window (A/B: no faster)
        else hipLaunchKernelGGL((conv_h2s_kernel<4, 0, 2>), grid, dim3(512), 0, s, a);
    } else if (a.tune & 64) hipLaunchKernelGGL((conv_h2s_kernel<2, 1>), grid, dim3(256), 0, s, a);
    else if (a.tune & 8) hipLaunchKernelGGL((conv_h2s_kernel<2, 0, 3>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_h2s_kernel<2, 0, 2>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace padel
