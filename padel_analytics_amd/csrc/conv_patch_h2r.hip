// K3r "h2 register-weights quad patch" (round 6) — the stride-1 3x3 convolution of h2 graphs whose weights are fp16 numbers
// (PA_CONV_W_SINGLE: two products per operand pair), for layers with whole 32-channel chunks and a multiple of 96 output
// channels: the tile of conv_patch_h2q.hip (8 x 16 pixels x 96 channels per workgroup, a wave 4 rows x 16 pixels x 48
// channels) with a main loop built for the two-product regime.
//
// Why: with two products a tap is 24 MFMAs = 384 matrix-pipe cycles per wave; conv_patch_h2q.hip wraps every tap in a serial
// chain of ~775 cycles (vmcnt(0) on the weight stage, one s_barrier, 3 + 4 dependent ds_read_b128) that only the OTHER
// workgroup's wave on the SIMD can cover: MfmaUtil 0.40-0.48 (profiles/r5t_conv_h2_pmc.txt).  The chain exists because the
// weights travel through a workgroup-shared LDS ring.  Here
//   * the WEIGHTS never touch LDS: a wave fetches the three 16 x 32 operand fragments of its 48 channels straight into
//     registers, TWO taps ahead, through three rotating register sets; waits are counted (vmcnt(6) keeps the two younger taps
//     in flight).  They come from a second copy of the h plane in MFMA operand order — [fragment][k-step][lane][16 bytes]:
//     1 KB = 8 whole cache lines per buffer_load_dwordx4 (h2r_repack_kernel, run once per model by the engine).  Reading the
//     row-major blob directly ([row][k-step][h | m]: 16 half-used lines per load) was measured first: 350 / 406 TFLOP/s on
//     96 -> 96 / 192 -> 192 against 441 / 517 for conv_patch_h2q.hip — 48 KB of lines per tap and CU is the whole L2 -> L1
//     fill rate (64 B/clk).  The two waves of a channel half read the same 3 KB per tap;
//   * there is NO per-tap barrier and no per-tap wait on anything another wave did: one s_barrier per 32-channel CHUNK (9 taps,
//     216 MFMAs = 3 456 pipe cycles per wave) hands over the double-buffered input patch;
//   * the input rows are prefetched: the six rows of a kernel column slide through four register slots as before, but every
//     read is issued as soon as its slot's last products have been issued — row 4 under tap ky = 0, row 5 under ky = 1, and
//     the next column's rows 0..3 under ky = 2, whose row order (2, 3, 0, 1) frees the slots in the order the new rows need
//     them — at least 12 MFMAs (192 cycles) before their first use; no read is exposed in the steady state;
//   * the patch of chunk c + 2 is requested by ALL FOUR waves (three 16-pixel spans of both planes each) right behind the
//     chunk barrier in tap 8 of chunk c, into the buffer that barrier just freed: a whole chunk to land.  vmcnt is in order
//     per wave, so the weights issued behind those six requests retire after them: the requests sit where two taps of
//     weights are already in flight in front of them.
// Same products in the same order per accumulator as every other h2 kernel (cross: wh am; main: wh ah, flushed into acc once
// per chunk; the rows of a tap touch different accumulators, so their order is free): bitwise identical results.
//
// LDS: 2 patch buffers x 2 planes x 192 pixels x 64 B, 32 KB apart = 57 344 B + 6 KB of request offsets; registers bound the occupancy (2 workgroups per CU).
#include "h2_common.h"

namespace padel {

namespace {

constexpr int kRPW = 18;                        // patch width in pixels (16 + halo)
constexpr int kRNPix = 180;                     // 10 x 18
constexpr int kRPlaneB = 192 * 64;              // one fp16 plane of a 32-channel chunk, padded to 12 spans of 16 pixels
constexpr int kRPatchB = 2 * kRPlaneB;
constexpr int kRBufStride = 32768;              // the two patch buffers sit 32 KB apart: switching buffers is one XOR per read address

// byte offset, inside a plane, of logical 16-byte chunk q (K slots 8q..8q+7) of patch pixel p (conv_patch_h2.hip:hp_off)
__device__ __forceinline__ unsigned hr_off(int p, int q) { return (unsigned)(p * 64 + ((q ^ (((p >> 2) & 1) << 1)) << 4)); }

typedef int hr_i32x4 __attribute__((ext_vector_type(4)));

// DBG (tuning only, builds with -DPADEL_H2P_PROBES, pa_engine_set_tuning "timeline"): the record format of conv_patch_h2q.hip — every
// wave stamps s_memtime at 5 points of every tap step into an LDS ring of 32 steps (tools/timeline_probe.py --kernel h2r):
// 0 step top / 1 this tap's weights landed (counted wait passed) / 2 chunk barrier passed (tap 8; elsewhere = 1) / 3 the first
// row's six products issued (its operands were in registers) / 4 last product issued
constexpr int kRDbgSteps = 32;
constexpr int kRDbgWords = 8 + 4 * kRDbgSteps * 5;
constexpr int kRDbgTile = 3;                     // the tile of a persistent workgroup whose steps the ring keeps

}  // namespace

// the tile of virtual block id v: XCD-aware 1-D map — the channel tiles of one pixel patch are neighbours on one XCD
// (conv_patch_h2q.hip).  Invalid ids (grid padding) sit at the end of every XCD's range: once a workgroup's id is invalid, all its
// later ones are
struct HrTile { int n, y0, x0, f0; bool valid; };
__device__ __forceinline__ HrTile hr_tile(const ConvArgs& a, int v, int vmax, int frags_per_tile) {
    HrTile t;
    const int nmt = a.n_mtiles, nnt = a.n_ntiles;
    const int q8 = nmt >> 3, r8 = nmt & 7, xcd = v & 7, idx = v >> 3;
    const int mloc = idx / nnt, nt = idx - mloc * nnt;
    t.valid = v < vmax && mloc < q8 + (xcd < r8 ? 1 : 0);
    const int mt = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + mloc;
    const int txN = (a.Wo + 15) >> 4, tyN = (a.Ho + 7) >> 3;
    const int tpi = tyN * txN;
    t.n = mt / tpi;
    const int rt = mt - t.n * tpi;
    const int ty = rt / txN, tx = rt - ty * txN;
    t.y0 = ty * 8; t.x0 = tx * 16;
    t.f0 = nt * frags_per_tile;
    return t;
}

// PERSISTENT workgroups (2 per CU, gridDim.x of them): workgroup b walks the virtual block ids b, b + gridDim.x, ...  What that
// buys is the prologue: a fresh workgroup spends ~9 k cycles between its first instruction and its first MFMA (tile arithmetic,
// the first patch from HBM, a barrier) — with the epilogue (7.7 k) a fixed 17.7 k cycles per tile in which the OTHER workgroup of the
// CU has the matrix pipe to itself but, on a 96 -> 96 layer (27 taps, 10.4 k pipe cycles), not enough work to fill it.  Measured on
// the non-persistent form: with everything but the MFMAs compiled out of the main loop the kernel reaches 533 / 691 TFLOP/s on
// 96 -> 96 / 192 -> 192 — the main loop (489 / 602 then) was within 12 % of free, the fixed cost was the bound
// (profiles/r6h_ablate_h2r.txt).  Here chunk 0 of the NEXT tile's patch is requested in tap 8 of the second-last chunk (where the
// non-persistent form requested zeros to keep its counted waits static) and lands during the last chunk and the epilogue; chunk 1
// follows at the next tile's start behind its first weights — the request order a fresh workgroup's prologue has, so every
// counted wait is the same in every chunk of every tile.
// Measured on top and NOT kept (profiles/r6*): a priority ladder over the rows of a tap (-2 .. 0 %); half a tile of sleep for the
// workgroup in the CU's odd thread-group slot — the tile log of the probe build shows the two workgroups of a CU are not in phase
// anyway (the older one gets the pipe first: 64.5 k vs 83 k ticks per tile, epilogues 0.9 beside the partner's main loop), no
// change; `nt` on the patch requests (-6 .. -8 %), `sc0` on the weight loads (0).
#ifndef PADEL_HR_LATEW
#define PADEL_HR_LATEW 1
#endif
template <int NF, bool WS, bool DBG = false, int ABL = 0, int PRIO = 0>
__global__ void __launch_bounds__(256, 2) conv_h2r_kernel(const ConvArgs a, const int vmax) {
    constexpr int MF = 4;
    // NF channel fragments per wave (a workgroup: 2 NF x 16 channels); WS: two products (the m plane of the weights is all zero and
    // is neither fetched nor multiplied), else three — which fits the registers with NF = 2 only: 64-channel tiles, the shape of
    // every TrackNetV3 layer (fp32 checkpoint: three products) and of the 64- / 256-channel layers of the YOLO graphs
    static_assert((NF == 3 && WS) || NF == 2, "NF = 3 with three products needs 72 weight registers: over the budget of 2 waves per SIMD");
    constexpr int NBUF = 2;                       // double-buffered patch
    constexpr int NPL = WS ? 1 : 2;               // weight planes fetched per fragment
    constexpr int NW = NF * NPL;                  // weight requests per tap and wave
    // the requests of tap T + 2 go out behind tap T's first row of products (their issue time — ~110 cycles for six loads — under the
    // matrix pipe) instead of in front of its counted wait: +1.0..1.4 % on 96 -> 96, +0.2 % on 192 -> 192 with 96-channel tiles, -0.4..1.4 %
    // with 64-channel tiles (four rows of FOUR products hide less): profiles/r7a_h2r_late_weight_requests.txt
    constexpr bool kLateW = PADEL_HR_LATEW != 0 && NF == 3;
    constexpr int kWQ = kLateW ? 1 : 2;           // taps of weight requests that are younger than W(T) when tap T waits for it
    constexpr int DBG_B = DBG ? (4 * kRDbgSteps * 5 + 4 * 64) * 8 : 0;
    constexpr int PATCH_B = (NBUF - 1) * kRBufStride + kRPatchB;
    constexpr int PVO_B = 2 * 3 * 256 * 4;       // the lane offsets of the patch requests, two tiles' worth
    __shared__ __attribute__((aligned(16))) float lds[(PATCH_B + PVO_B + DBG_B) / 4];
    char* const ldsb = reinterpret_cast<char*>(lds);
    unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(ldsb + PATCH_B + PVO_B);
    unsigned long long t_begin = 0, t_end_rec = 0;
    int dbg_k = 0, dbg_tile = 0;
    (void)stamps; (void)t_begin; (void)t_end_rec; (void)dbg_k; (void)dbg_tile;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ablation probes (-DPADEL_H2P_PROBES instantiates ABL != 0, wrong results by construction; pa_engine_set_tuning "tune" selects
    // one): bits switch pieces of the main loop off — 16 the patch requests, 32 the flushes, 64 the chunk barrier, 128 tap 8's row
    // reads, 256 the weight requests, 512 the other row reads
#define PADEL_HR_ON(BIT_) (!(ABL & (BIT_)))
    // (PRIO = 1, tuning only: a priority ladder over the rows of a tap, so that the wave further into its burst keeps the pipe —
    //  measured -2 % .. +0 %, profiles/r6g_prio_ladder.txt; off)
#define PADEL_HR_PRIO(N_) do { if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(N_); } while (0)
#define PADEL_HR_STAMP(J, slot)                                                                                   \
    do {                                                                                                          \
        if constexpr (DBG) {                               /* the ring records ONE tile in the middle of the launch (the fourth) */ \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                           \
            const int real_ = (wave * kRDbgSteps + ((dbg_k + (J)) & (kRDbgSteps - 1))) * 5 + (slot);              \
            const int dummy_ = 4 * kRDbgSteps * 5 + wave * 64 + lane;                                             \
            stamps[(lane == 0 && dbg_tile == kRDbgTile) ? real_ : dummy_] = t_;                                   \
        }                                                                                                         \
    } while (0)
    const int wr = wave & 1, wc = wave >> 1;      // pixel half (rows 4 wr ..), channel half (fragments 3 wc ..)
    const int lr = lane & 15, lq = lane >> 4;
    const int nch = a.cin >> 5;                   // >= 2 (launch_conv_h2r)
    const int G = (int)gridDim.x;
    int v = (int)blockIdx.x;
    HrTile cur = hr_tile(a, v, vmax, 2 * NF);
    if (!cur.valid) return;
    // ---- the patch: span s of a plane = 16 pixels x 64 bytes, lane i -> pixel 16 s + i / 4, physical 16-byte slot i & 3 =
    // logical chunk q of that pixel (hr_off), which is piece (q & 1) of group (q >> 1) of the pixel's 128 bytes [h0 m0 h1 m1]
    // in HBM; the plane's 32 bytes go in through the scalar offset.  Wave w requests spans 3 w .. 3 w + 2 of both planes.
    // The lane offsets of the wave's three spans do not depend on the chunk (it enters through the scalar offset): 3 VGPRs per tile.
    // A tile that does not exist gets a descriptor of zero records: every lane out of range, zeros into a free buffer — the request
    // count per tap, and with it every counted wait, is static.
    const unsigned lp0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
    const unsigned lpw = __builtin_amdgcn_readfirstlane(lp0 + (unsigned)wave * 3072u);
    // Two parameter sets exist during a tile's main loop: this tile's (chunks 2.. and, at the tile's start, chunk 1) and the next
    // tile's (its chunk 0, requested in tap 8 of this tile's second-last chunk).  The lane offsets are needed once per chunk: they
    // live in LDS (2 sets x 3 spans x 256 lanes x 4 B behind the patch buffers, every lane reads what it wrote), not in 6 VGPRs —
    // the main loop has none to spare.  The next tile's are computed at the tile's start, while the accumulators are not live yet.
    i32x4 rsrcP, rsrcPn;
    unsigned* const pvo = reinterpret_cast<unsigned*>(ldsb + PATCH_B) + tid;       // [set][span][256]
#define PADEL_HR_PARAMS(T_, RS_, SET_)                                                                            \
    do {                                                                                                          \
        const float* const in0_ = a.in + (((long long)(T_).n * a.H + ((T_).y0 - 1)) * a.W + ((T_).x0 - 1)) * a.in_cs + a.in_choff; \
        RS_ = make_rsrc3(in0_);                                                                                   \
        RS_[2] = (T_).valid ? (int)0x80000000u : 0;                                                               \
        const int pl_ = lane >> 2;                                                                                \
        const int pq_ = (lane & 3) ^ (((lane >> 4) & 1) << 1);                                                    \
        const unsigned piece_ = (unsigned)((pq_ >> 1) * 64 + (pq_ & 1) * 16);                                     \
        _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                                           \
            const int pp = (3 * wave + k) * 16 + pl_;                                                             \
            const int py = pp / kRPW, px = pp - py * kRPW;                                                        \
            const bool ok = pp < kRNPix && (unsigned)((T_).y0 - 1 + py) < (unsigned)a.H && (unsigned)((T_).x0 - 1 + px) < (unsigned)a.W; \
            pvo[((SET_) * 3 + k) * 256] = ok ? (unsigned)((py * a.W + px) * a.in_cs * 4) + piece_ : kOOR3;        \
        }                                                                                                         \
    } while (0)
    // the wave's requests of span K_ (both planes) of the chunk at scalar offset SO_ of the tile RS_ / parameter set SET_ describe,
    // into buffer BUF_
#define PADEL_HR_PSPAN(K_, SO_, BUF_, RS_, SET_)                                                                  \
    do {                                                                                                          \
        const unsigned lb_ = lpw + (unsigned)(BUF_) * (unsigned)kRBufStride;                                      \
        const unsigned vo_ = pvo[((SET_) * 3 + (K_)) * 256];                                                      \
        dma3<(K_) * 1024>(vo_, RS_, (SO_), lb_); dma3<kRPlaneB + (K_) * 1024>(vo_, RS_, (SO_) + 32u, lb_);        \
    } while (0)
#define PADEL_HR_PATCH(CH_, BUF_, RS_, SET_)                                                                      \
    do {                                                                                                          \
        const unsigned so_ = (unsigned)(CH_) * 128u;                                                              \
        PADEL_HR_PSPAN(0, so_, BUF_, RS_, SET_); PADEL_HR_PSPAN(1, so_, BUF_, RS_, SET_); PADEL_HR_PSPAN(2, so_, BUF_, RS_, SET_); \
    } while (0)
    // tap 8 of chunk c: chunk c + 2 of this tile, or — in the second-last chunk — chunk 0 of the next one; in the last chunk the
    // requests go through a descriptor of zero records (nothing is fetched; they retire in front of the next tile's first weights
    // and leave every counted wait as it is).  Scalar selects only: tap 8 stays one basic block
#define PADEL_HR_PSPAN8(K_)                                                                                       \
    do {                                                                                                          \
        const unsigned lb_ = lpw + (unsigned)gpar * (unsigned)kRBufStride;                                        \
        dma3<(K_) * 1024>(vo8[K_], rs8, so8, lb_); dma3<kRPlaneB + (K_) * 1024>(vo8[K_], rs8, so8 + 32u, lb_);    \
    } while (0)

    // ---- weights: a.wr = [fragment][k-step][h | m][lane][16 bytes]: lane l of fragment j reads bytes [16 l, 16 l + 16) of the
    // k-step's h KB (and, with three products, of its m KB) — its MFMA A operand (row l & 15 at k = 8 (l >> 4)).  One descriptor per
    // fragment, one lane offset
    const unsigned fragb = (unsigned)(nch * 9) * 2048u;
    const unsigned voffW = (unsigned)lane * 16u;
    i32x4 rsrcW[NF];
    hr_i32x4 w[3][NF], wm[3][WS ? 1 : NF];        // (wm: the correction plane's operands; unused with WS)
    unsigned s_kb = 0;                            // byte offset of the current chunk's first k-step inside a fragment
    // tap TT_ (0..10, relative to the current chunk: 9 and 10 are the next chunk's first two) into register set SET_; the reads of
    // the last chunk's 9 / 10 run up to 4 KB past a fragment (into the next fragment, or the slack behind the copy)
#define PADEL_HR_LOADW(SET_, TT_)                                                                                 \
    do {                                                                                                          \
        const unsigned so_ = s_kb + (unsigned)((TT_) * 2048);                                                     \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) {                                                          \
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen"                                               \
                         : "=v"(w[SET_][j]) : "v"(voffW), "s"(rsrcW[j]), "s"(so_) : "memory");                    \
            if constexpr (!WS)                                                                                    \
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024"                               \
                             : "=v"(wm[SET_][j]) : "v"(voffW), "s"(rsrcW[j]), "s"(so_) : "memory");               \
        }                                                                                                         \
    } while (0)
    // the counted wait that publishes set SET_ to the compiler: the registers pass through it.  BASE_: requests of the wave that may
    // stay in flight besides the two younger taps' (0, or 6 patch requests)
#define PADEL_HR_WAITW(SET_, BASE_)                                                                               \
    do {                                                                                                          \
        if constexpr (NF == 3)                                                                                    \
            asm volatile("s_waitcnt vmcnt(%3)" : "+v"(w[SET_][0]), "+v"(w[SET_][1]), "+v"(w[SET_][2]) : "n"((BASE_) + kWQ * NW) : "memory"); \
        else if constexpr (WS)                                                                                    \
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[SET_][0]), "+v"(w[SET_][1]) : "n"((BASE_) + kWQ * NW) : "memory"); \
        else                                                                                                      \
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[SET_][0]), "+v"(w[SET_][1]), "+v"(wm[SET_][0]), "+v"(wm[SET_][WS ? 0 : 1]) \
                         : "n"((BASE_) + kWQ * NW) : "memory");                                                   \
    } while (0)

    // ---- row reads: patch pixel p = p0 + d with p0 = 72 wr + lr (the wave's row 0 at kx = 0) and d = 18 R + KX; hr_off's swizzle
    // term depends on bit 2 of p only, i.e. on d & 7 (adding a multiple of 8 leaves bit 2 alone): eight lane addresses, everything
    // else of a read is an immediate (64 d, + the m plane).  The two patch buffers sit 32 KB apart, so that the chunk hand-over is
    // one XOR per address; a read costs no VALU at all (the first version recomputed ~8 VALU per read: the ky = 2 taps, four
    // reads, took 570 ticks instead of 410 — profiles/r6d_timeline_h2r_192.txt)
    unsigned rbase[8];                            // (set at every tile's start: not live through the epilogue)

    f32x4 acc[MF][NF], part[MF][NF], cross[MF][NF];
    h16x8 ah[4], am[4];                           // input rows in 4 sliding slots (row r of the current kx in slot r & 3)
    // input row R_ (0..5 of the wave's window) at column shift KX_ of the buffer rbase points into, into slot R_ & 3
#define PADEL_HR_READROW(R_, KX_)                                                                                 \
    do {                                                                                                          \
        constexpr int d_ = (R_) * kRPW + (KX_);                                                                   \
        const char* p_ = ldsb + rbase[d_ & 7];                                                                    \
        ah[(R_) & 3] = *reinterpret_cast<const h16x8*>(p_ + d_ * 64);                                             \
        am[(R_) & 3] = *reinterpret_cast<const h16x8*>(p_ + d_ * 64 + kRPlaneB);                                  \
    } while (0)
    // the 6 products of output row F_ at tap row KY_ (its input row F_ + KY_ sits in slot (F_ + KY_) & 3) with weight set SET_;
    // FIRST_: tap 0 starts the chunk's main chain from the MFMA's constant-0 C operand (bitwise 0 + x)
#define PADEL_HR_MFMA_ROW(F_, KY_, SET_, FIRST_)                                                                  \
    do {                                                                                                          \
        constexpr int s_ = ((F_) + (KY_)) & 3;                                                                    \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            cross[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, w[SET_][j]), am[s_], cross[F_][j], 0, 0, 0); \
        if constexpr (!WS) {                                                                                      \
            _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                        \
                cross[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, wm[SET_][WS ? 0 : j]), ah[s_], cross[F_][j], 0, 0, 0); \
        }                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            part[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, w[SET_][j]), ah[s_],   \
                                                                 (FIRST_) ? (f32x4){0.f, 0.f, 0.f, 0.f} : part[F_][j], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
    // the chunk's main sum of output row F_ into acc (two-level summation, as in every h2 kernel)
#define PADEL_HR_FLUSH(F_)                                                                                        \
    do {                                                                                                          \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) acc[F_][j] += part[F_][j];                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
    // tap step T_ = 3 kx + ky of the current chunk (weight set T_ % 3 — 9 % 3 == 0, so a tap uses the same set in every chunk).
    // Queue of the wave behind W(T_) when it waits: [P, issued in tap 8 / at the tile's start] W(T_ + 1) [P] W(T_ + 2): 6 requests, 12
    // in taps 0 / 1.  Tap 8 carries the chunk's bookkeeping between its MFMA rows, where the matrix pipe still holds the previous
    // row's products: the chunk barrier, the buffer switch of the row addresses, the next chunk's first rows, the requests of the
    // patch after next (a span per row; none in a tile's last chunk) and the flush of the main sums (row k's behind row k + 1's
    // products; the last one under the next tap 0).
#define PADEL_HR_STEP(T_)                                                                                         \
    do {                                                                                                          \
        constexpr int kx_ = h2_tap_kx(T_), ky_ = h2_tap_ky(T_), set_ = (T_) % 3;                                  \
        PADEL_HR_STAMP(T_, 0);                                                                                    \
        PADEL_HR_PRIO(0);                                                                                         \
        if constexpr (!kLateW && PADEL_HR_ON(256)) PADEL_HR_LOADW(((T_) + 2) % 3, (T_) + 2);                      \
        PADEL_HR_WAITW(set_, (T_) < 2 ? 6 : 0);                                                                   \
        PADEL_HR_STAMP(T_, 1);                                                                                    \
        if constexpr ((T_) != 8) PADEL_HR_STAMP(T_, 2);                                                           \
        PADEL_HR_PRIO(1);                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if constexpr (ky_ == 0) {                                                                                 \
            PADEL_HR_MFMA_ROW(0, 0, set_, (T_) == 0 && !(ABL & 32));                                              \
            if constexpr (kLateW && PADEL_HR_ON(256)) PADEL_HR_LOADW(((T_) + 2) % 3, (T_) + 2);   /* its issue under this row's products */ \
            PADEL_HR_STAMP(T_, 3);                                                                                \
            if constexpr ((T_) == 0 && !(ABL & 32)) PADEL_HR_FLUSH(1);   /* the previous chunk's last row (chunk 0: + 0) */ \
            if constexpr (PADEL_HR_ON(512)) PADEL_HR_READROW(4, kx_);   /* slot 0, first used by row 3 of ky = 1 */ \
            PADEL_HR_PRIO(2);                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            PADEL_HR_MFMA_ROW(1, 0, set_, (T_) == 0 && !(ABL & 32)); PADEL_HR_PRIO(3);                            \
            PADEL_HR_MFMA_ROW(2, 0, set_, (T_) == 0 && !(ABL & 32)); PADEL_HR_MFMA_ROW(3, 0, set_, (T_) == 0 && !(ABL & 32)); \
        } else if constexpr (ky_ == 1) {                                                                          \
            PADEL_HR_MFMA_ROW(0, 1, set_, false);                                                                 \
            if constexpr (kLateW && PADEL_HR_ON(256)) PADEL_HR_LOADW(((T_) + 2) % 3, (T_) + 2);   /* its issue under this row's products */ \
            PADEL_HR_STAMP(T_, 3);                                                                                \
            if constexpr (PADEL_HR_ON(512)) PADEL_HR_READROW(5, kx_);   /* slot 1, first used by row 3 of ky = 2 */ \
            PADEL_HR_PRIO(2);                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            PADEL_HR_MFMA_ROW(1, 1, set_, false); PADEL_HR_PRIO(3); PADEL_HR_MFMA_ROW(2, 1, set_, false); PADEL_HR_MFMA_ROW(3, 1, set_, false); \
        } else if constexpr (kx_ < 2) {                    /* ky = 2: rows 2, 3, 0, 1 free slots 0, 1, 2, 3 for the next column */ \
            PADEL_HR_MFMA_ROW(2, 2, set_, false);                                                                 \
            if constexpr (kLateW && PADEL_HR_ON(256)) PADEL_HR_LOADW(((T_) + 2) % 3, (T_) + 2);   /* its issue under this row's products */ \
            PADEL_HR_STAMP(T_, 3);                                                                                \
            if constexpr (PADEL_HR_ON(512)) PADEL_HR_READROW(0, kx_ + 1);                                         \
            PADEL_HR_PRIO(2);                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            PADEL_HR_MFMA_ROW(3, 2, set_, false);                                                                 \
            if constexpr (PADEL_HR_ON(512)) PADEL_HR_READROW(1, kx_ + 1);                                         \
            PADEL_HR_PRIO(3);                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            PADEL_HR_MFMA_ROW(0, 2, set_, false);                                                                 \
            if constexpr (PADEL_HR_ON(512)) PADEL_HR_READROW(2, kx_ + 1);                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            PADEL_HR_MFMA_ROW(1, 2, set_, false);                                                                 \
            if constexpr (PADEL_HR_ON(512)) PADEL_HR_READROW(3, kx_ + 1);                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        } else {                                           /* tap 8: the chunk barrier sits behind the first row's products */ \
            PADEL_HR_MFMA_ROW(2, 2, set_, false);                                                                 \
            if constexpr (kLateW && PADEL_HR_ON(256)) PADEL_HR_LOADW(((T_) + 2) % 3, (T_) + 2);                   \
            /* every read of this chunk's patch has returned (row 5 was read under tap 7; lgkmcnt(0) costs nothing here), */ \
            /* and the wave's own requests of the next chunk's patch landed long ago (in order in front of W(8)) */ \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
            if constexpr (PADEL_HR_ON(64)) __builtin_amdgcn_s_barrier();                                          \
            asm volatile("" ::: "memory");                                                                        \
            PADEL_HR_STAMP(T_, 2); PADEL_HR_STAMP(T_, 3);                                                         \
            unsigned vo8[3];                               /* the three lane offsets: read here, used a row of MFMAs later */ \
            _Pragma("unroll") for (int k = 0; k < 3; ++k) vo8[k] = pvo[(set8 * 3 + k) * 256];                     \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) rbase[r] ^= (unsigned)kRBufStride;                      \
            if constexpr (PADEL_HR_ON(128)) PADEL_HR_READROW(0, 0);                                               \
            PADEL_HR_PRIO(2);                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            PADEL_HR_MFMA_ROW(3, 2, set_, false);                                                                 \
            PADEL_HR_PRIO(3);                                                                                     \
            if constexpr (PADEL_HR_ON(32)) PADEL_HR_FLUSH(2);                                                     \
            if constexpr (PADEL_HR_ON(128)) PADEL_HR_READROW(1, 0);                                               \
            if constexpr (PADEL_HR_ON(16)) PADEL_HR_PSPAN8(0);   /* into the buffer this chunk read */ \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            PADEL_HR_MFMA_ROW(0, 2, set_, false);                                                                 \
            if constexpr (PADEL_HR_ON(32)) PADEL_HR_FLUSH(3);                                                     \
            if constexpr (PADEL_HR_ON(128)) PADEL_HR_READROW(2, 0);                                               \
            if constexpr (PADEL_HR_ON(16)) PADEL_HR_PSPAN8(1);                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            PADEL_HR_MFMA_ROW(1, 2, set_, false);                                                                 \
            if constexpr (PADEL_HR_ON(32)) PADEL_HR_FLUSH(0);                                                     \
            if constexpr (PADEL_HR_ON(128)) PADEL_HR_READROW(3, 0);                                               \
            if constexpr (PADEL_HR_ON(16)) PADEL_HR_PSPAN8(2);                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        }                                                                                                         \
        PADEL_HR_STAMP(T_, 4);                                                                                    \
    } while (0)

    int gpar = 0;                                 // parity of the running chunk count = the buffer the current chunk reads
    bool first = true;
    int tpar = 0;                                 // parity of the tile count = the parameter set of the current tile
    PADEL_HR_PARAMS(cur, rsrcPn, 0);
    PADEL_HR_PATCH(0, 0, rsrcPn, 0);
    for (;;) {                                    // ---- tiles
        if constexpr (DBG) {
            const unsigned long long tb_ = __builtin_amdgcn_s_memtime(); dbg_k = 0;
            if (dbg_tile <= kRDbgTile) t_begin = tb_;
            if (dbg_tile == kRDbgTile + 1) t_end_rec = tb_;
            // the tile log (records gridDim.x + blockIdx.x of the dump, behind the per-workgroup records): tile start / epilogue start
            if (a.dbg && tid == 0 && dbg_tile < kRDbgWords / 2) a.dbg[((long long)G + blockIdx.x) * kRDbgWords + 2 * dbg_tile] = tb_;
        }
        // this tile's patch parameters are the ones the previous tile computed as "next"
        rsrcP = rsrcPn;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int frag = min(cur.f0 + NF * wc + j, a.n16 - 1);   // fragments beyond the matrix: any valid rows (never stored)
            rsrcW[j] = make_rsrc3(reinterpret_cast<const char*>(a.wr) + (long long)frag * fragb);
        }
        s_kb = 0;
        // W(0), W(1), P(1): the order the steady state leaves behind tap 8 (P(0): before the loop / tap 8 of the previous tile's
        // second-last chunk)
        PADEL_HR_LOADW(0, 0);
        PADEL_HR_LOADW(1, 1);
        PADEL_HR_PATCH(1, gpar ^ 1, rsrcP, tpar);
        const HrTile nxt = hr_tile(a, v + G, vmax, 2 * NF);        // (the tile arithmetic runs under the latency of the requests above)
        PADEL_HR_PARAMS(nxt, rsrcPn, tpar ^ 1);
        if constexpr ((ABL & 256) != 0) { wait_vm3<0>(); for (int j = 0; j < NF; ++j) { w[2][j] = w[0][j]; if constexpr (!WS) wm[2][j] = wm[0][j]; } }      // (probe: no weight requests inside the loop)
        if (first) {
            wait_vm3<2 * NW + 6>();               // P(0) landed (W(0), W(1), P(1) may be in flight)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        {
            int lr_ = lr;
            asm volatile("" : "+v"(lr_));
            const int p0 = 4 * wr * kRPW + lr_;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                rbase[r] = (unsigned)(p0 * 64 + ((lq ^ ((((p0 + r) >> 2) & 1) << 1)) << 4)) ^ (gpar ? (unsigned)kRBufStride : 0u);
        }
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = acc[f][j]; cross[f][j] = acc[f][j]; }
        PADEL_HR_READROW(0, 0); PADEL_HR_READROW(1, 0); PADEL_HR_READROW(2, 0); PADEL_HR_READROW(3, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            const bool own8 = c + 2 < nch;
            i32x4 rs8 = own8 ? rsrcP : rsrcPn;
            if (c + 1 >= nch || (ABL & 2048)) rs8[2] = 0;      // (probe 2048: every tap-8 request through the zero-record descriptor)
            const unsigned so8 = own8 ? (unsigned)(c + 2) * 128u : 0u;
            const int set8 = own8 ? tpar : tpar ^ 1;
            PADEL_HR_STEP(0); PADEL_HR_STEP(1); PADEL_HR_STEP(2); PADEL_HR_STEP(3); PADEL_HR_STEP(4);
            PADEL_HR_STEP(5); PADEL_HR_STEP(6); PADEL_HR_STEP(7); PADEL_HR_STEP(8);
            s_kb += 9u * 2048u;
            gpar ^= 1;
            dbg_k += 9;
        }
        PADEL_HR_PRIO(0);
        // the look-ahead requests of the last chunk (taps 9 / 10: weights nobody uses, issued to keep the counted waits static) target
        // register sets 0 and 1: they must have landed before the epilogue may reuse those registers
        PADEL_HR_WAITW(0, -kWQ * NW); PADEL_HR_WAITW(1, -kWQ * NW);       // vmcnt(0), both sets' registers through it
        PADEL_HR_FLUSH(1);
        if constexpr ((ABL & 32) != 0) { PADEL_HR_FLUSH(0); PADEL_HR_FLUSH(2); PADEL_HR_FLUSH(3); }      // (probe: one main chain, flushed once)

        if constexpr (DBG) {
            if (a.dbg && tid == 0 && dbg_tile < kRDbgWords / 2) a.dbg[((long long)G + blockIdx.x) * kRDbgWords + 2 * dbg_tile + 1] = __builtin_amdgcn_s_memtime();
            ++dbg_tile;
        }
        int mpix[MF];
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int oy = cur.y0 + 4 * wr + f, ox = cur.x0 + lr;
            mpix[f] = (oy < a.Ho && ox < a.Wo) ? (cur.n * a.Ho + oy) * a.Wo + ox : -1;
        }
        const int fw = cur.f0 + NF * wc;
        const bool fast = cur.y0 + 8 <= a.Ho && cur.x0 + 16 <= a.Wo && (fw + NF) * 16 <= a.cout && (((a.out_choff | a.out_cs) & 3) == 0) &&
                          (!a.res || (((a.res_choff | a.res_cs) & 3) == 0));
        if (fw < a.n16) h2_epilogue<MF, NF>(a, acc, cross, mpix, fw, lq, fast);
        if (!nxt.valid) break;
        cur = nxt;
        v += G;
        tpar ^= 1;
        first = false;
    }
    wait_vm3<0>();                                // the tail's requests (zeros into a free buffer, weights nobody uses) before the LDS is released
#undef PADEL_HR_STEP
#undef PADEL_HR_MFMA_ROW
#undef PADEL_HR_READROW
#undef PADEL_HR_WAITW
#undef PADEL_HR_LOADW
#undef PADEL_HR_PATCH
#undef PADEL_HR_PSPAN
#undef PADEL_HR_PSPAN8
#undef PADEL_HR_PARAMS
#undef PADEL_HR_FLUSH
    if constexpr (DBG) {
        if (a.dbg) {
            const unsigned long long t_end = t_end_rec ? t_end_rec : __builtin_amdgcn_s_memtime();      // the recorded tile's end = the next tile's start
            __syncthreads();
            unsigned long long* d = a.dbg + (long long)blockIdx.x * kRDbgWords;
            for (int i = tid; i < 4 * kRDbgSteps * 5; i += 256) d[8 + i] = stamps[i];
            if (lane == 0) {
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
                const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
                d[wave] = ((unsigned long long)xcc << 32) | hw;
                if (wave == 0) { d[4] = t_begin; d[5] = t_end; d[6] = (unsigned long long)(nch * 9); d[7] = (unsigned long long)blockIdx.x; }      // the LAST tile's life
            }
        }
    }
#undef PADEL_HR_STAMP
#undef PADEL_HR_PRIO
#undef PADEL_HR_ON
}

// one thread per 16 bytes of the copy: [fragment f][k-step t][plane p][lane l] <- bytes [16 (l >> 4), +16) of plane p (h | m) of k-step t
// of row 16 f + (l & 15) of the packed blob ([row][k-step][h x 32 | m x 32] fp16)
__global__ void __launch_bounds__(256) h2r_repack_kernel(const char* __restrict__ w, char* __restrict__ wr, int n16, int ksteps) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n16 * ksteps * 128) return;
    const int l = (int)(i & 63), pl = (int)((i >> 6) & 1);
    const long long ft = i >> 7;
    const int t = (int)(ft % ksteps), f = (int)(ft / ksteps);
    const h2_u32x4 v = *reinterpret_cast<const h2_u32x4*>(w + ((long long)(f * 16 + (l & 15)) * ksteps + t) * 128 + pl * 64 + (l >> 4) * 16);
    *reinterpret_cast<h2_u32x4*>(wr + i * 16) = v;
}

// PA_CONV_W_SINGLE is a promise of the CALLER's (public C-ABI): the m plane of the packed weights is all zero.  Checked once per
// model on the device: any non-zero m bit of the conv's rows x k-steps raises *flag
__global__ void __launch_bounds__(256) h2_mplane_check_kernel(const unsigned* __restrict__ w, long long n_steps, unsigned* flag) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // one thread per 16 bytes of an m half (64 B = 4 threads)
    if (i >= n_steps * 4) return;
    const h2_u32x4 v = *reinterpret_cast<const h2_u32x4*>(w + (i >> 2) * 32 + 16 + (i & 3) * 4);
    if (((v[0] | v[1] | v[2] | v[3]) & 0x7FFF7FFFu) != 0u) atomicOr(flag, 1u);
}

hipError_t launch_h2_mplane_check(const float* w, long long rows_x_ksteps, unsigned* flag, hipStream_t s) {
    hipLaunchKernelGGL(h2_mplane_check_kernel, dim3((unsigned)((rows_x_ksteps * 4 + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const unsigned*>(w), rows_x_ksteps, flag);
    return hipGetLastError();
}

// k-steps of a stride-1 3x3 layer in the packed blob: 9 taps per whole 32-channel chunk + 5 tap pairs for a 16-channel tail
// (1x1 layers: one k-step per 32 channels, a 16-channel tail half-filled)
static int h2_ksteps(int cin, int ksize) { return ksize == 3 ? (cin >> 5) * 9 + ((cin & 16) ? 5 : 0) : (cin + 31) >> 5; }
size_t conv_h2r_copy_bytes(int n16, int cin, int ksize) { return (size_t)n16 * (size_t)h2_ksteps(cin, ksize) * 2048; }

hipError_t launch_h2r_repack(const float* w, void* wr, int n16, int cin, int ksize, hipStream_t s) {
    const int ksteps = h2_ksteps(cin, ksize);
    const long long n = (long long)n16 * ksteps * 128;
    hipLaunchKernelGGL(h2r_repack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const char*>(w),
                       reinterpret_cast<char*>(wr), n16, ksteps);
    return hipGetLastError();
}

bool conv_h2r_supported(const ConvArgs& a) {
    return a.wr && a.ksize == 3 && a.stride == 1 && (a.cin & 31) == 0 && a.cin >= 64 && a.Ho == a.H && a.Wo == a.W && a.w != nullptr && !a.in2;
}

// nf = channel fragments per wave: 3 (96-channel tiles; two-product layers only) or 2 (64-channel tiles; two or three products)
hipError_t launch_conv_h2r(const ConvArgs& a_in, int nf, hipStream_t s) {
    if (!conv_h2r_supported(a_in) || (nf != 2 && nf != 3) || (nf == 3 && !a_in.w_single)) return hipErrorNotSupported;
    ConvArgs a = a_in;
    const int batch = a.M / (a.Ho * a.Wo);
    a.n_mtiles = batch * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
    a.n_ntiles = (a.n16 + 2 * nf - 1) / (2 * nf);
    const int vmax = 8 * ((a.n_mtiles + 7) / 8) * a.n_ntiles;          // virtual block ids
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; hipDeviceProp_t p; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n_cu = p.multiProcessorCount; if (n_cu <= 0) n_cu = 256; }
    const int per = (a.tune & 4) ? vmax : 2 * n_cu;                     // tuning bit 2: one workgroup per tile (the non-persistent form)
    dim3 grid((unsigned)(vmax < per ? vmax : per), 1, 1);               // 2 workgroups per CU (registers); both multiples of 8
#ifdef PADEL_H2P_PROBES
    if (nf == 3) {
        if (a.dbg) {
            hipLaunchKernelGGL((conv_h2r_kernel<3, true, true>), grid, dim3(256), 0, s, a, vmax);
            return hipGetLastError();
        }
#define PADEL_HR_ABL(N_) case N_: hipLaunchKernelGGL((conv_h2r_kernel<3, true, false, N_>), grid, dim3(256), 0, s, a, vmax); return hipGetLastError();
        switch ((a.tune >> 5) << 4) {      // tuning word bits 5.. = ABL >> 4
            PADEL_HR_ABL(16) PADEL_HR_ABL(240) PADEL_HR_ABL(1008) PADEL_HR_ABL(2048)
            default: break;
        }
#undef PADEL_HR_ABL
    }
#endif
    if (nf == 3) hipLaunchKernelGGL((conv_h2r_kernel<3, true>), grid, dim3(256), 0, s, a, vmax);
    else if (a.w_single) hipLaunchKernelGGL((conv_h2r_kernel<2, true>), grid, dim3(256), 0, s, a, vmax);
    else hipLaunchKernelGGL((conv_h2r_kernel<2, false>), grid, dim3(256), 0, s, a, vmax);
    return hipGetLastError();
}

}  // namespace padel
