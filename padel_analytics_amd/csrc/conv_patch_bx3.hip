// K3p "bf16x3 patch" — the stride-1 3x3 convolution with the input tile split ONCE per 32-channel chunk.
//
// The tap kernel of conv_tap_bx3.hip treats the 3x3 conv as an implicit GEMM over (tap, channel): every k-step fetches
// its own 128 x 32 fp32 activation tile through the LDS-DMA ring and every wave splits the values it reads into bf16
// triples — nine times per input value, once per tap.  Probes of that kernel (profiles/conv_bx3_probes_r2m.txt) put
// the split at 18 % and the activation requests at 13 % of its time, 35 % together.  Here a workgroup owns an 8 x 16
// patch of output pixels of one image and, per 32-channel chunk,
//   1. loads the 10 x 18 input patch (halo included; pixels outside the image come back as zeros from the buffer range
//      check) straight into registers, 6 x buffer_load_dwordx4 per lane, one chunk ahead of its use;
//   2. splits it once (exact hi / mid / lo bf16 truncations, bx3_common.h) and writes three bf16 planes to LDS, 64 bytes
//      per pixel and plane in the K order the pre-split weights expect;
//   3. walks the 9 taps: a tap is a SHIFTED 16-pixel window of the same planes, read as ready-made MFMA operands
//      (ds_read_b128, conflict-free for every shift: 16-byte chunk q of pixel p lives at q ^ 2 * ((p >> 2) & 1));
//      only the weights of the tap still travel through a 2-stage LDS-DMA ring.
// Per k-step and wave that is 15 ds_read_b128 + 36 MFMAs + <= 3 requests and no VALU, against 13 reads + 88 split VALU
// + 7 requests before.  Products, their order and the two-level accumulation are those of the tap
// kernel, so results are bitwise identical to it (tests/test_gpu_conv.py).
//
// LDS: 3 planes x 180 pixels x 64 B = 34 560 B + 2 weight stages of 3 x BN x 64 B: 52 992 B for BN = 48 -> 3 workgroups
// per CU.  Scope: ksize 3, stride 1, cin % 16 == 0 (launch_conv_bx3p reports anything else as not supported and the
// dispatcher keeps the tap kernel for it).
#include "bx3_common.h"

namespace padel {

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kPW = 18;                       // patch width in pixels (16 + halo)
constexpr int kNPix = 180;                    // 10 x 18
constexpr int kPlaneB = kNPix * 64;           // one bf16 plane of a 32-channel chunk
constexpr int kPatchB = 3 * kPlaneB;
constexpr int kItems = kNPix * 8;             // 16-byte (4-channel) pieces of the fp32 patch
constexpr int kPasses = (kItems + 255) / 256; // 6

// byte offset, inside a plane, of logical 16-byte chunk q (K slots 8q..8q+7) of patch pixel p
__device__ __forceinline__ unsigned patch_off(int p, int q) { return (unsigned)(p * 64 + ((q ^ (((p >> 2) & 1) << 1)) << 4)); }

// 4 fp32 -> exact bf16 triples, 2 per dword (same arithmetic as split8)
__device__ __forceinline__ void split4(const u32x4 x, u32x2& hi, u32x2& mid, u32x2& lo) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const unsigned be = x[2 * p], bo = x[2 * p + 1];
        const float xe = __uint_as_float(be), xo = __uint_as_float(bo);
        const float re = xe - __uint_as_float(be & 0xFFFF0000u), ro = xo - __uint_as_float(bo & 0xFFFF0000u);
        const unsigned bre = __float_as_uint(re), bro = __float_as_uint(ro);
        const float le = re - __uint_as_float(bre & 0xFFFF0000u), lo_ = ro - __uint_as_float(bro & 0xFFFF0000u);
        hi[p] = __builtin_amdgcn_perm(bo, be, 0x07060302u);
        mid[p] = __builtin_amdgcn_perm(bro, bre, 0x07060302u);
        lo[p] = __builtin_amdgcn_perm(__float_as_uint(lo_), __float_as_uint(le), 0x07060302u);
    }
}

// tail planes: 32 bytes per pixel, 8-byte slot q (4 channels)
__device__ __forceinline__ unsigned tail_off(int p, int q) { return (unsigned)(p * 32 + ((q ^ (((p >> 3) & 1) << 1)) << 3)); }
constexpr int kTailPasses = (kNPix * 4 + 255) / 256;     // 3

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

}  // namespace

// TAIL: cin % 32 == 16 — after the full chunks a 16-channel patch (32 bytes per pixel and plane) is walked in 5 k-steps
// that pair TAPS like the tap kernel's tail block: K slots 0-3 of a lane = its 4 channels at tap 2t, slots 4-7 = at tap
// 2t + 1 (two ds_read_b64 per operand; 8-byte slot q of pixel p lives at q ^ 2 * ((p >> 3) & 1): conflict-free)
// UP: the first a.up_c channels (whole chunks) are read from a.in2, a map of half the spatial size, at [y >> 1][x >> 1]:
// an nn.Upsample(2) + torch.cat in front of this conv (TrackNet's decoder blocks) that is never materialised
template <int NF, bool TAIL, bool UP>
__global__ void __launch_bounds__(256, NF <= 3 ? 3 : 2) conv_bx3p_kernel(const ConvArgs a) {
    constexpr int MF = 2;
    constexpr int BN = NF * 16;
    constexpr int BSTAGE_B = 3 * BN * 64;
    constexpr int BP = (BN + 63) / 64, BFULL = BN / 64;
    static_assert(BP <= 2, "weights in at most 2 passes of 64 rows");
    static_assert(kPatchB + 2 * BSTAGE_B <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) float lds[(kPatchB + 2 * BSTAGE_B) / 4];
    char* const ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;

    // XCD-aware 1-D tile map (see PADEL_BX3_GEOMETRY): the channel tiles of one pixel patch are neighbours on one XCD
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

    // ---- the fp32 patch: piece i * 256 + tid = (pixel, 4-channel group); lane offsets are chunk-independent
    unsigned voffP[kPasses], wrP[kPasses];
#pragma unroll
    for (int i = 0; i < kPasses; ++i) {
        const int item = i * 256 + tid;
        const int pp = item >> 3, c4 = item & 7;
        const int py = pp / kPW, px = pp - py * kPW;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = item < kItems && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        voffP[i] = ok ? (unsigned)(((py * a.W + px) * a.in_cs + c4 * 4) * 4) : kOOR3;
        wrP[i] = patch_off(pp, c4 & 3) + (unsigned)((c4 >> 2) * 8);       // channels [0,16) of the chunk: K slots 0-3 of a lane, [16,32): 4-7
    }
    const float* const in0 = a.in + (((long long)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * a.in_cs + a.in_choff;
    const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in0), 0, (int)0x80000000u, 0x00020000);
    // coarse map of an absorbed upsample: descriptor based at the coarse pixel of the patch's top-left halo pixel
    const int H2 = a.H >> 1, W2 = a.W >> 1;
    const int cy0 = (y0 - 1) >> 1, cx0 = (x0 - 1) >> 1;                 // arithmetic shifts: -1 for the halo above / left of the image
    const float* const inU = UP ? a.in2 + (((long long)n * H2 + cy0) * W2 + cx0) * a.in2_cs + a.in2_choff : in0;
    const __amdgpu_buffer_rsrc_t rsrcU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inU), 0, (int)0x80000000u, 0x00020000);
    const int nup = UP ? a.up_c >> 5 : 0;
    (void)H2; (void)W2; (void)cy0; (void)cx0; (void)rsrcU; (void)nup;

    // ---- weights: rows of (cin / 32) * 9 k-steps x 192 bytes (hi | mid | lo), k-step = chunk * 9 + tap
    const int nch = a.cin >> 5;
    const unsigned rowb = (unsigned)(nch * 9 + (TAIL ? 5 : 0)) * 192u;
    const int srow = tid >> 2;
    const int sc = (tid & 3) ^ ((4 - ((srow >> 2) & 3)) & 3);
    unsigned voffB[BP];
#pragma unroll
    for (int p = 0; p < BP; ++p) {
        const int rr = srow + 64 * p;
        const int frag = min(f0 + (rr >> 4), a.n16 - 1);
        voffB[p] = (unsigned)(((frag - f0) * 16 + (rr & 15)) * rowb + sc * 16);
    }
    const i32x4 rsrcB = make_rsrc3(reinterpret_cast<const char*>(a.w3) + (long long)f0 * 16 * rowb);
    const bool b_last = BP > BFULL && (BFULL * 64 + wave * 16 < BN);
    unsigned lw0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + (unsigned)kPatchB + wave * 1024u);
    unsigned lw1 = __builtin_amdgcn_readfirstlane(lw0 + (unsigned)BSTAGE_B);
    const int ld_off = lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);
    const float* b_rd0 = lds + kPatchB / 4 + ld_off;
    const float* b_rd1 = b_rd0 + BSTAGE_B / 4;
    const int rd_pix = 2 * wave * kPW + lr;                 // patch pixel of fragment 0, tap (0, 0)

    f32x4 acc[MF][NF], part[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

#define PADEL_P_DMAB(SR_, SB_)                                                                                    \
    do {                                                                                                          \
        const unsigned lw_ = ((SR_) & 1) ? lw1 : lw0;                                                             \
        const unsigned sb_ = (SB_);                                                                               \
        PADEL_P_DMAB1(0, sb_);                                                                                    \
        PADEL_P_DMAB1(1, sb_ + 64u);                                                                              \
        PADEL_P_DMAB1(2, sb_ + 128u);                                                                             \
    } while (0)
#define PADEL_P_DMAB1(PL_, S_)                                                                                    \
    do {                                                                                                          \
        if constexpr (BFULL >= 1) dma3<(PL_) * BN * 64>(voffB[0], rsrcB, (S_), lw_);                              \
        if constexpr (BP > BFULL) { if (b_last) dma3<(PL_) * BN * 64 + BFULL * 4096>(voffB[BP - 1], rsrcB, (S_), lw_); } \
    } while (0)
#define PADEL_P_LOAD(CH_)                                                                                         \
    do {                                                                                                          \
        const unsigned so_ = (unsigned)(CH_) * 128u;                                                              \
        if (UP && (int)(CH_) < nup) {      /* lane offsets into the coarse map, recomputed (once per chunk) */      \
            _Pragma("unroll") for (int i = 0; i < kPasses; ++i) {                                                 \
                const int item = i * 256 + tid;                                                                   \
                const int pp = item >> 3, c4 = item & 7;                                                          \
                const int py = pp / kPW, px = pp - py * kPW;                                                      \
                const int iy = y0 - 1 + py, ix = x0 - 1 + px;                                                     \
                const bool ok = item < kItems && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;   \
                const unsigned vo_ = ok ? (unsigned)(((((iy >> 1) - cy0) * W2 + ((ix >> 1) - cx0)) * a.in2_cs + c4 * 4) * 4) : kOOR3; \
                pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcU, vo_, so_, 0);                               \
            }                                                                                                     \
        } else {                                                                                                  \
            _Pragma("unroll") for (int i = 0; i < kPasses; ++i) pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcP, voffP[i], so_, 0); \
        }                                                                                                         \
    } while (0)
#define PADEL_P_READA(T_)                                                                                         \
    do {                                                                                                          \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                          \
            const char* p_ = ldsb + patch_off(rd_pix + (f + (T_) / 3) * kPW + (T_) % 3, lq);                      \
            ah[f] = *reinterpret_cast<const bf8*>(p_);                                                            \
            am[f] = *reinterpret_cast<const bf8*>(p_ + kPlaneB);                                                  \
            al[f] = *reinterpret_cast<const bf8*>(p_ + 2 * kPlaneB);                                              \
        }                                                                                                         \
    } while (0)
    // tail step JT: taps 2 JT and 2 JT + 1 (the 10th "tap" has zero weights: any finite data, tap 8 again)
#define PADEL_P_TREADA(JT_)                                                                                       \
    do {                                                                                                          \
        constexpr int ta_ = 2 * (JT_), tb_ = 2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8;                               \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                          \
            const char* pa_ = ldsb + tail_off(rd_pix + (f + ta_ / 3) * kPW + ta_ % 3, lq);                        \
            const char* pb_ = ldsb + tail_off(rd_pix + (f + tb_ / 3) * kPW + tb_ % 3, lq);                        \
            u32x4 v_;                                                                                             \
            { const u32x2 x_ = *reinterpret_cast<const u32x2*>(pa_), y_ = *reinterpret_cast<const u32x2*>(pb_);    \
              v_ = (u32x4){x_[0], x_[1], y_[0], y_[1]}; ah[f] = __builtin_bit_cast(bf8, v_); }                    \
            { const u32x2 x_ = *reinterpret_cast<const u32x2*>(pa_ + kPlaneB), y_ = *reinterpret_cast<const u32x2*>(pb_ + kPlaneB); \
              v_ = (u32x4){x_[0], x_[1], y_[0], y_[1]}; am[f] = __builtin_bit_cast(bf8, v_); }                    \
            { const u32x2 x_ = *reinterpret_cast<const u32x2*>(pa_ + 2 * kPlaneB), y_ = *reinterpret_cast<const u32x2*>(pb_ + 2 * kPlaneB); \
              v_ = (u32x4){x_[0], x_[1], y_[0], y_[1]}; al[f] = __builtin_bit_cast(bf8, v_); }                    \
        }                                                                                                         \
    } while (0)
#define PADEL_P_READB(T_)                                                                                         \
    do {                                                                                                          \
        const float* const br_ = ((T_) & 1) ? b_rd1 : b_rd0;                                                      \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) {                                                          \
            wh[j] = __builtin_bit_cast(bf8, *reinterpret_cast<const f32x4*>(br_ + j * 256));                      \
            wm[j] = __builtin_bit_cast(bf8, *reinterpret_cast<const f32x4*>(br_ + BN * 16 + j * 256));            \
            wl[j] = __builtin_bit_cast(bf8, *reinterpret_cast<const f32x4*>(br_ + 2 * BN * 16 + j * 256));        \
        }                                                                                                         \
    } while (0)
#define PADEL_P_MFMA()                                                                                            \
    do {                                                                                                          \
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
    // tap step T of the current chunk: its weights (requested one step earlier) have landed for every wave after the
    // barrier, which also releases the other weight stage (read in step T - 1) for the request of step T + 1; step 0
    // additionally publishes the freshly written planes and requests the next chunk's patch
#define PADEL_P_STEP(T_)                                                                                          \
    do {                                                                                                          \
        bf8 ah[MF], am[MF], al[MF], wh[NF], wm[NF], wl[NF];                                                       \
        /* the planes do not change inside a chunk, so from tap 1 on the activation operands are requested BEFORE   \
           the barrier (their LDS latency runs under the wait), and the weight operands before the next step's      \
           LDS-DMA requests are issued: +1..2.5 % (profiles/conv_bx3_sweep_r2r_read_order.txt) */                 \
        if constexpr ((T_) > 0) PADEL_P_READA(T_);                                                                \
        wait_vm3<0>();                                                                                            \
        if constexpr ((T_) == 0) lds_fence();          /* this wave's plane writes have reached the LDS */         \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
        PADEL_P_READB(T_);                                                                                        \
        if constexpr ((T_) == 0) PADEL_P_READA(T_);                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if ((T_) < 8 || c + 1 < nch || TAIL) PADEL_P_DMAB((T_) + 1, s_kb + ((T_) + 1) * 192u);                    \
        if ((T_) == 0 && c + 1 < nch) PADEL_P_LOAD(c + 1);    /* one request per tap step instead: measured -1 % */ \
        if constexpr (TAIL) { if ((T_) == 0 && c + 1 == nch) PADEL_P_TLOAD(); }                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_P_MFMA();                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

    // tail patch: 180 pixels x 4 pieces of 4 channels
#define PADEL_P_TLOAD()                                                                                           \
    do {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < kTailPasses; ++i) {                                                 \
            const int item = i * 256 + tid;                                                                       \
            const int pp = item >> 2, c4 = item & 3;                                                              \
            const int py = pp / kPW, px = pp - py * kPW;                                                          \
            const bool ok = item < kNPix * 4 && (unsigned)(y0 - 1 + py) < (unsigned)a.H && (unsigned)(x0 - 1 + px) < (unsigned)a.W; \
            pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcP, ok ? (unsigned)(((py * a.W + px) * a.in_cs + c4 * 4) * 4) : kOOR3, \
                                                           (unsigned)nch * 128u, 0);                              \
        }                                                                                                         \
    } while (0)
#define PADEL_P_TSTEP(JT_)                                                                                        \
    do {                                                                                                          \
        bf8 ah[MF], am[MF], al[MF], wh[NF], wm[NF], wl[NF];                                                       \
        wait_vm3<0>();                                                                                            \
        lds_fence();                                                                                              \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
        if constexpr ((JT_) < 4) PADEL_P_DMAB((JT_) + 1, s_kb + ((JT_) + 1) * 192u);                              \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_P_TREADA(JT_);                                                                                      \
        PADEL_P_READB(JT_);                                                                                       \
        PADEL_P_MFMA();                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

    u32x4 pre[kPasses];
    unsigned s_kb = 0;
    if (!TAIL || nch > 0) PADEL_P_LOAD(0); else PADEL_P_TLOAD();
    PADEL_P_DMAB(0, 0u);
    for (int c = 0; c < nch; ++c) {
        if (c > 0) {                          // every wave is done with the taps of chunk c - 1: the planes may be overwritten
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < kPasses; ++i) {
            u32x2 h, m, l;
            split4(pre[i], h, m, l);
            if (i * 256 + 255 < kItems || i * 256 + tid < kItems) {
                *reinterpret_cast<u32x2*>(ldsb + wrP[i]) = h;
                *reinterpret_cast<u32x2*>(ldsb + kPlaneB + wrP[i]) = m;
                *reinterpret_cast<u32x2*>(ldsb + 2 * kPlaneB + wrP[i]) = l;
            }
        }
        PADEL_P_STEP(0); PADEL_P_STEP(1); PADEL_P_STEP(2); PADEL_P_STEP(3); PADEL_P_STEP(4);
        PADEL_P_STEP(5); PADEL_P_STEP(6); PADEL_P_STEP(7); PADEL_P_STEP(8);
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        { const float* t_ = b_rd0; b_rd0 = b_rd1; b_rd1 = t_; const unsigned u_ = lw0; lw0 = lw1; lw1 = u_; }
        s_kb += 9u * 192u;
    }
    if constexpr (TAIL) {
        if (nch > 0) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < kTailPasses; ++i) {
            const int item = i * 256 + tid;
            const int pp = item >> 2, c4 = item & 3;
            u32x2 h, m, l;
            split4(pre[i], h, m, l);
            if (item < kNPix * 4) {
                char* const w_ = ldsb + tail_off(pp, c4);
                *reinterpret_cast<u32x2*>(w_) = h;
                *reinterpret_cast<u32x2*>(w_ + kPlaneB) = m;
                *reinterpret_cast<u32x2*>(w_ + 2 * kPlaneB) = l;
            }
        }
        PADEL_P_TSTEP(0); PADEL_P_TSTEP(1); PADEL_P_TSTEP(2); PADEL_P_TSTEP(3); PADEL_P_TSTEP(4);
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[f][j] += part[f][j];
    }
    wait_vm3<0>();
#undef PADEL_P_TSTEP
#undef PADEL_P_TLOAD
#undef PADEL_P_TREADA
#undef PADEL_P_READA
#undef PADEL_P_READB
#undef PADEL_P_MFMA
#undef PADEL_P_STEP
#undef PADEL_P_LOAD
#undef PADEL_P_DMAB
#undef PADEL_P_DMAB1

    int mpix[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int oy = y0 + 2 * wave + f, ox = x0 + lr;
        mpix[f] = (oy < a.Ho && ox < a.Wo) ? (n * a.Ho + oy) * a.Wo + ox : -1;
    }
    const bool fast = y0 + 8 <= a.Ho && x0 + 16 <= a.Wo && (f0 + NF) * 16 <= a.cout && (((a.out_choff | a.out_cs) & 3) == 0) &&
                      (!a.res || (((a.res_choff | a.res_cs) & 3) == 0));
    bx3_epilogue<MF, NF>(a, acc, mpix, f0, lq, fast);
}

template <int NF, bool TAIL, bool UP>
static hipError_t launch_pt(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    const int batch = a.M / (a.Ho * a.Wo);
    a.n_mtiles = batch * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
    a.n_ntiles = (a.n16 + NF - 1) / NF;
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    hipLaunchKernelGGL((conv_bx3p_kernel<NF, TAIL, UP>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

template <int NF>
static hipError_t launch_p(const ConvArgs& a_in, hipStream_t s) {
    if (a_in.in2) {                    // absorbed upsample: whole 32-channel chunks only, even map size
        if ((a_in.cin & 31) || (a_in.up_c & 31) || a_in.up_c <= 0 || a_in.up_c > a_in.cin || ((a_in.H | a_in.W) & 1)) return hipErrorNotSupported;
        return launch_pt<NF, false, true>(a_in, s);
    }
    if (a_in.cin & 16) return launch_pt<NF, true, false>(a_in, s);
    return launch_pt<NF, false, false>(a_in, s);
}

bool conv_bx3p_supported(const ConvArgs& a) {
    return a.ksize == 3 && a.stride == 1 && (a.cin & 15) == 0 && a.cin >= 16 && a.Ho == a.H && a.Wo == a.W && a.w3 != nullptr;
}

// nf = channel fragments (of 16) per workgroup: 3 (8x16 pixels x 48 channels, 3 workgroups per CU), 4, 6
hipError_t launch_conv_bx3p(const ConvArgs& a, int nf, hipStream_t s) {
    if (!conv_bx3p_supported(a)) return hipErrorNotSupported;
    switch (nf) {
        case 3: return launch_p<3>(a, s);
        case 4: return launch_p<4>(a, s);
        case 6: return launch_p<6>(a, s);
    }
    return hipErrorNotSupported;
}

}  // namespace padel
