// K3w "h2 wide patch" — the stride-1 3x3 convolution of h2 graphs for layers with FEW input channels (16 / 32 / 48: the P2
// bottlenecks of yolov8m — 48 -> 48 at 320 x 320 — and the first levels of yolov8n).  Their K loop is 5-14 steps long: on the
// 8 x 16 patch kernel (conv_patch_h2.hip) a workgroup lives 28 k cycles for 4 k cycles of MFMAs per wave — prologue (the
// patch's round trip to HBM), epilogue (SiLU + pair encoding of 24 outputs per lane) and the per-step barriers are the
// kernel: 0.24-0.31 of the matrix pipe, 7 ms of the bench's step (profiles/r3z_ops_c3.csv).
//
// Here a workgroup owns 16 x 16 output pixels (halo 1.27 x instead of 1.41 x) and the WHOLE K extent of its input lives in
// LDS: the 18 x 18 patch of the 32-channel chunk (if any) and of the 16-channel tail (if any) are requested up front by
// LDS-DMA — no patch buffer swap, no per-chunk synchronisation — and only the weights travel through the 2-stage ring.
// A wave owns 4 rows x 16 pixels x all NF channel fragments (12 NF MFMAs per tap: twice the work per barrier); the full
// chunk is walked column-major with its input rows sliding through 4 register slots (conv_patch_h2q.hip), the tail in 5
// steps that pair taps (lane groups 0, 1 of an operand: channels 0-7 / 8-15 at tap 2t; groups 2, 3: at tap 2t + 1).
// Same products in the same order per accumulator as every other h2 kernel: bitwise identical results.
//
// LDS (NF = 3, cin = 48): 43 008 B chunk patch (2 planes x 21 spans of 16 pixels) + 22 528 B tail patch (2 planes x 11 spans
// of 32 pixels) + 2 weight stages x 2 planes x 48 rows x 64 B = 12 288 B: 77 824 B, 2 workgroups per CU.
#include "h2_common.h"

namespace padel {

namespace {

constexpr int kWPW = 18, kWNPix = 18 * 18;            // 16 x 16 output pixels + halo
constexpr int kWSpans = (kWNPix + 15) / 16;            // 21 spans of 16 pixels x 64 B per chunk plane
constexpr int kWPlaneB = kWSpans * 1024;
constexpr int kWTSpans = (kWNPix + 31) / 32;           // 11 spans of 32 pixels x 32 B per tail plane
constexpr int kWTPlaneB = kWTSpans * 1024;

__device__ __forceinline__ unsigned hw_off(int p, int q) { return (unsigned)(p * 64 + ((q ^ (((p >> 2) & 1) << 1)) << 4)); }
__device__ __forceinline__ unsigned hw_tail_off(int p, int s) { return (unsigned)(p * 32 + ((s ^ ((p >> 3) & 1)) << 4)); }

}  // namespace

// WS: the packed weights' m plane is all zero (ConvArgs::w_single): no wm x ah product, no m-plane requests / reads (conv_patch_h2q.hip)
template <int NF, bool CHUNK, bool TAIL, bool WS = false>
__global__ void __launch_bounds__(256, 2) conv_h2w_kernel(const ConvArgs a) {
    static_assert(CHUNK || TAIL, "cin = 32 CHUNK + 16 TAIL");
    constexpr int MF = 4;
    constexpr int BN = NF * 16;
    constexpr int BPLANE_B = BN * 64;
    constexpr int BSTAGE_B = 2 * BPLANE_B;
    constexpr int PATCH_B = (CHUNK ? 2 * kWPlaneB : 0);
    constexpr int TPATCH_B = (TAIL ? 2 * kWTPlaneB : 0);
    constexpr int NSTEPS = (CHUNK ? 9 : 0) + (TAIL ? 5 : 0);
    static_assert(PATCH_B + TPATCH_B + 2 * BSTAGE_B <= 80 * 1024, "2 workgroups per CU");
    __shared__ __attribute__((aligned(16))) float lds[(PATCH_B + TPATCH_B + 2 * BSTAGE_B) / 4];
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
    const int txN = (a.Wo + 15) >> 4, tyN = (a.Ho + 15) >> 4;
    const int tpi = tyN * txN;
    const int n = mt / tpi, rt = mt - n * tpi;
    const int ty = rt / txN, tx = rt - ty * txN;
    const int y0 = ty * 16, x0 = tx * 16;
    const int f0 = nt * NF;

    const float* const in0 = a.in + (((long long)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * a.in_cs + a.in_choff;
    const i32x4 rsrcP = make_rsrc3(in0);
    const unsigned lp0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);

    // ---- the whole input, requested up front (spans round-robin over the 4 waves).
    // chunk plane span s: 16 pixels x 64 bytes, lane i -> pixel 16 s + i / 4, physical 16-byte slot i & 3 = logical chunk q of
    // that pixel (hw_off), piece (q & 1) of group (q >> 1) of the pixel's 128 bytes [h0 m0 h1 m1]; + 32 bytes for the m plane
    if constexpr (CHUNK) {
        const int p_lane = lane >> 2;
        const int p_q = (lane & 3) ^ (((lane >> 4) & 1) << 1);
        const unsigned p_piece = (unsigned)((p_q >> 1) * 64 + (p_q & 1) * 16);
#define PADEL_HW_PSPAN(S_)                                                                                        \
        if (((S_) & 3) == wave) {                                                                                 \
            const int pp_ = (S_) * 16 + p_lane;                                                                   \
            const int py_ = pp_ / kWPW, px_ = pp_ - py_ * kWPW;                                                   \
            const bool ok_ = pp_ < kWNPix && (unsigned)(y0 - 1 + py_) < (unsigned)a.H && (unsigned)(x0 - 1 + px_) < (unsigned)a.W; \
            const unsigned vo_ = ok_ ? (unsigned)((py_ * a.W + px_) * a.in_cs * 4) + p_piece : kOOR3;             \
            dma3<(S_) * 1024>(vo_, rsrcP, 0u, lp0);                                                               \
            dma3<kWPlaneB + (S_) * 1024>(vo_, rsrcP, 32u, lp0);                                                   \
        }
        PADEL_HW_PSPAN(0) PADEL_HW_PSPAN(1) PADEL_HW_PSPAN(2) PADEL_HW_PSPAN(3) PADEL_HW_PSPAN(4) PADEL_HW_PSPAN(5) PADEL_HW_PSPAN(6)
        PADEL_HW_PSPAN(7) PADEL_HW_PSPAN(8) PADEL_HW_PSPAN(9) PADEL_HW_PSPAN(10) PADEL_HW_PSPAN(11) PADEL_HW_PSPAN(12) PADEL_HW_PSPAN(13)
        PADEL_HW_PSPAN(14) PADEL_HW_PSPAN(15) PADEL_HW_PSPAN(16) PADEL_HW_PSPAN(17) PADEL_HW_PSPAN(18) PADEL_HW_PSPAN(19) PADEL_HW_PSPAN(20)
#undef PADEL_HW_PSPAN
    }
    // tail plane span s: 32 pixels x 32 bytes, lane i -> pixel 32 s + i / 2, physical slot i & 1 = logical slot (channels 8 s'..)
    // (i & 1) ^ ((p >> 3) & 1) of the tail group's 64 bytes [h x 16 | m x 16] behind the full chunks
    if constexpr (TAIL) {
        const int t_lane = lane >> 1;
        const unsigned t_piece = (unsigned)(((lane & 1) ^ ((lane >> 4) & 1)) * 16);
        const unsigned t_so = CHUNK ? 128u : 0u;
#define PADEL_HW_TSPAN(S_)                                                                                        \
        if (((S_) & 3) == ((wave + 1) & 3)) {                                                                     \
            const int pp_ = (S_) * 32 + t_lane;                                                                   \
            const int py_ = pp_ / kWPW, px_ = pp_ - py_ * kWPW;                                                   \
            const bool ok_ = pp_ < kWNPix && (unsigned)(y0 - 1 + py_) < (unsigned)a.H && (unsigned)(x0 - 1 + px_) < (unsigned)a.W; \
            const unsigned vo_ = ok_ ? (unsigned)((py_ * a.W + px_) * a.in_cs * 4) + t_piece : kOOR3;             \
            dma3<PATCH_B + (S_) * 1024>(vo_, rsrcP, t_so, lp0);                                                   \
            dma3<PATCH_B + kWTPlaneB + (S_) * 1024>(vo_, rsrcP, t_so + 32u, lp0);                                 \
        }
        PADEL_HW_TSPAN(0) PADEL_HW_TSPAN(1) PADEL_HW_TSPAN(2) PADEL_HW_TSPAN(3) PADEL_HW_TSPAN(4) PADEL_HW_TSPAN(5)
        PADEL_HW_TSPAN(6) PADEL_HW_TSPAN(7) PADEL_HW_TSPAN(8) PADEL_HW_TSPAN(9) PADEL_HW_TSPAN(10)
#undef PADEL_HW_TSPAN
    }

    // ---- weights: rows of NSTEPS k-steps x 128 bytes (h | m).  A stage = 2 planes x NF spans of 16 rows x 64 bytes; wave w
    // requests the spans w and w + 4 (span = plane * NF + row group): lane i -> row i / 4 of the span, physical slot i & 3
    constexpr unsigned rowb = (unsigned)NSTEPS * 128u;
    const int b_row = lane >> 2;
    const int b_sc = (lane & 3) ^ ((4 - ((b_row >> 2) & 3)) & 3);
    unsigned voffB[2];
    bool haveB[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int sp = wave + 4 * k;
        haveB[k] = sp < (WS ? NF : 2 * NF);          // spans NF .. 2 NF - 1 are the m plane
        const int pl = sp / NF, g = sp - pl * NF;
        const int frag = min(f0 + min(g, NF - 1), a.n16 - 1);
        voffB[k] = (unsigned)(((frag - f0) * 16 + b_row) * rowb + pl * 64 + b_sc * 16);
    }
    const i32x4 rsrcB = make_rsrc3(reinterpret_cast<const char*>(a.w) + (long long)f0 * 16 * rowb);
    const unsigned lw0 = __builtin_amdgcn_readfirstlane(lp0 + (unsigned)(PATCH_B + TPATCH_B) + (unsigned)wave * 1024u);
    const unsigned lw1 = __builtin_amdgcn_readfirstlane(lw0 + (unsigned)BSTAGE_B);
#define PADEL_HW_DMAB(ST_)                                                                                        \
    do {                                                                                                          \
        const unsigned lw_ = ((ST_) & 1) ? lw1 : lw0;                                                             \
        if (haveB[0]) dma3<0>(voffB[0], rsrcB, (unsigned)(ST_) * 128u, lw_);                                      \
        if (haveB[1]) dma3<4096>(voffB[1], rsrcB, (unsigned)(ST_) * 128u, lw_);                                   \
    } while (0)
    const int ld_off = lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);      // floats
    const float* const b_rd0 = lds + (PATCH_B + TPATCH_B) / 4 + ld_off;
    const float* const b_rd1 = b_rd0 + BSTAGE_B / 4;
    const int rd_pix = 4 * wave * kWPW + lr;               // patch pixel of the wave's row 0, kx = 0

    f32x4 acc[MF][NF], part[MF][NF], cross[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = acc[f][j]; cross[f][j] = acc[f][j]; }
    h16x8 ah[4], am[4], wh[NF], wm[NF];       // full chunk: input rows in 4 sliding slots (row r of the current kx in slot r & 3)
#define PADEL_HW_READROW(R_, KX_)                                                                                 \
    do {                                                                                                          \
        const char* p_ = ldsb + hw_off(rp_ + (R_) * kWPW + (KX_), lq);                                            \
        ah[(R_) & 3] = *reinterpret_cast<const h16x8*>(p_);                                                       \
        am[(R_) & 3] = *reinterpret_cast<const h16x8*>(p_ + kWPlaneB);                                            \
    } while (0)
#define PADEL_HW_READB(ST_)                                                                                       \
    do {                                                                                                          \
        const float* const br_ = ((ST_) & 1) ? b_rd1 : b_rd0;                                                     \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) {                                                          \
            wh[j] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(br_ + j * 256));                    \
            if constexpr (!WS) wm[j] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(br_ + BPLANE_B / 4 + j * 256)); \
        }                                                                                                         \
    } while (0)
    // the 3 NF products of output row F_ with the operands in slot S_
#define PADEL_HW_MFMA_ROW(F_, S_)                                                                                 \
    do {                                                                                                          \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            cross[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], am[S_], cross[F_][j], 0, 0, 0);          \
        if constexpr (!WS) {                                                                                      \
            _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                        \
                cross[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[j], ah[S_], cross[F_][j], 0, 0, 0);      \
        }                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            part[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], ah[S_], part[F_][j], 0, 0, 0);            \
    } while (0)
    // step ST_ of the K walk (weights of the step in stage ST_ & 1): barrier = this step's weights (requested one step earlier)
    // have landed for every wave and the other stage is free for the request of step ST_ + 1; the first barrier also
    // publishes the patches
#define PADEL_HW_SYNC(ST_)                                                                                        \
    do {                                                                                                          \
        wait_vm3<0>();                                                                                            \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
        PADEL_HW_READB(ST_);                                                                                      \
    } while (0)
    // full-chunk tap T_ = 3 kx + ky: ky == 0 reads rows 0..3 of the new column, ky == 1 row 4 (slot of row 0), ky == 2 row 5
#define PADEL_HW_STEP(T_)                                                                                         \
    do {                                                                                                          \
        constexpr int kx_ = h2_tap_kx(T_), ky_ = h2_tap_ky(T_);                                                   \
        int rp_ = rd_pix;                                  /* row addresses recomputed per tap: hoisted ones would spill */ \
        asm volatile("" : "+v"(rp_));                                                                             \
        if constexpr ((T_) > 0) {                          /* the patch is static: read under the wait */           \
            if constexpr (ky_ == 0) { PADEL_HW_READROW(0, kx_); PADEL_HW_READROW(1, kx_); PADEL_HW_READROW(2, kx_); PADEL_HW_READROW(3, kx_); } \
            else PADEL_HW_READROW(3 + ky_, kx_);                                                                  \
        }                                                                                                         \
        PADEL_HW_SYNC(T_);                                                                                        \
        if constexpr ((T_) == 0) { PADEL_HW_READROW(0, 0); PADEL_HW_READROW(1, 0); PADEL_HW_READROW(2, 0); PADEL_HW_READROW(3, 0); } \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        PADEL_HW_MFMA_ROW(0, (0 + ky_) & 3);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if ((T_) + 1 < NSTEPS) PADEL_HW_DMAB((T_) + 1);                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_HW_MFMA_ROW(1, (1 + ky_) & 3); PADEL_HW_MFMA_ROW(2, (2 + ky_) & 3); PADEL_HW_MFMA_ROW(3, (3 + ky_) & 3); \
        __builtin_amdgcn_s_setprio(0);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
    // tail step JT_ (step ST_ of the walk): lane group q of an operand holds the 8 channels 8 (q & 1).. of tap 2 JT_ + (q >> 1)
    // (the 10th "tap" has zero weights: any finite data, tap 8 again)
#define PADEL_HW_TSTEP(JT_, ST_)                                                                                  \
    do {                                                                                                          \
        constexpr int ta_ = 2 * (JT_), tb_ = 2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8;                               \
        int rp_ = rd_pix;                                                                                         \
        asm volatile("" : "+v"(rp_));                                                                             \
        if constexpr ((ST_) > 0) {                                                                                \
            _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                      \
                const int pa_ = rp_ + (f + h2_tap_ky(ta_)) * kWPW + h2_tap_kx(ta_), pb_ = rp_ + (f + h2_tap_ky(tb_)) * kWPW + h2_tap_kx(tb_); \
                const char* p_ = ldsb + PATCH_B + hw_tail_off((lq >> 1) ? pb_ : pa_, lq & 1);                     \
                ah[f] = *reinterpret_cast<const h16x8*>(p_);                                                      \
                am[f] = *reinterpret_cast<const h16x8*>(p_ + kWTPlaneB);                                          \
            }                                                                                                     \
        }                                                                                                         \
        PADEL_HW_SYNC(ST_);                                                                                       \
        if constexpr ((ST_) == 0) {                                                                               \
            _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                      \
                const int pa_ = rp_ + (f + h2_tap_ky(ta_)) * kWPW + h2_tap_kx(ta_), pb_ = rp_ + (f + h2_tap_ky(tb_)) * kWPW + h2_tap_kx(tb_); \
                const char* p_ = ldsb + PATCH_B + hw_tail_off((lq >> 1) ? pb_ : pa_, lq & 1);                     \
                ah[f] = *reinterpret_cast<const h16x8*>(p_);                                                      \
                am[f] = *reinterpret_cast<const h16x8*>(p_ + kWTPlaneB);                                          \
            }                                                                                                     \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        PADEL_HW_MFMA_ROW(0, 0);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if ((ST_) + 1 < NSTEPS) PADEL_HW_DMAB((ST_) + 1);                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_HW_MFMA_ROW(1, 1); PADEL_HW_MFMA_ROW(2, 2); PADEL_HW_MFMA_ROW(3, 3);                                \
        __builtin_amdgcn_s_setprio(0);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#define PADEL_HW_FLUSH()                                                                                          \
    do {                                                                                                          \
        _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                            \
            _Pragma("unroll") for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; } \
    } while (0)

    PADEL_HW_DMAB(0);
    if constexpr (CHUNK) {
        PADEL_HW_STEP(0); PADEL_HW_STEP(1); PADEL_HW_STEP(2); PADEL_HW_STEP(3); PADEL_HW_STEP(4);
        PADEL_HW_STEP(5); PADEL_HW_STEP(6); PADEL_HW_STEP(7); PADEL_HW_STEP(8);
        PADEL_HW_FLUSH();
    }
    if constexpr (TAIL) {
        constexpr int S0 = CHUNK ? 9 : 0;
        PADEL_HW_TSTEP(0, S0); PADEL_HW_TSTEP(1, S0 + 1); PADEL_HW_TSTEP(2, S0 + 2); PADEL_HW_TSTEP(3, S0 + 3); PADEL_HW_TSTEP(4, S0 + 4);
        PADEL_HW_FLUSH();
    }
    wait_vm3<0>();
#undef PADEL_HW_FLUSH
#undef PADEL_HW_TSTEP
#undef PADEL_HW_STEP
#undef PADEL_HW_SYNC
#undef PADEL_HW_MFMA_ROW
#undef PADEL_HW_READB
#undef PADEL_HW_READROW
#undef PADEL_HW_DMAB

    int mpix[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int oy = y0 + 4 * wave + f, ox = x0 + lr;
        mpix[f] = (oy < a.Ho && ox < a.Wo) ? (n * a.Ho + oy) * a.Wo + ox : -1;
    }
    const bool fast = y0 + 16 <= a.Ho && x0 + 16 <= a.Wo && (f0 + NF) * 16 <= a.cout && (((a.out_choff | a.out_cs) & 3) == 0) &&
                      (!a.res || (((a.res_choff | a.res_cs) & 3) == 0));
    h2_epilogue<MF, NF>(a, acc, cross, mpix, f0, lq, fast);
}

bool conv_h2w_supported(const ConvArgs& a) {
    return a.ksize == 3 && a.stride == 1 && (a.cin == 16 || a.cin == 32 || a.cin == 48) && a.Ho == a.H && a.Wo == a.W && a.w != nullptr && !a.in2;
}

template <int NF>
static hipError_t launch_hw(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    const int batch = a.M / (a.Ho * a.Wo);
    a.n_mtiles = batch * ((a.Ho + 15) / 16) * ((a.Wo + 15) / 16);
    a.n_ntiles = (a.n16 + NF - 1) / NF;
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    if (a.w_single) {
        if (a.cin == 48) hipLaunchKernelGGL((conv_h2w_kernel<NF, true, true, true>), grid, dim3(256), 0, s, a);
        else if (a.cin == 32) hipLaunchKernelGGL((conv_h2w_kernel<NF, true, false, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_h2w_kernel<NF, false, true, true>), grid, dim3(256), 0, s, a);
    } else if (a.cin == 48) hipLaunchKernelGGL((conv_h2w_kernel<NF, true, true>), grid, dim3(256), 0, s, a);
    else if (a.cin == 32) hipLaunchKernelGGL((conv_h2w_kernel<NF, true, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_h2w_kernel<NF, false, true>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// nf = channel fragments (of 16) per workgroup: 1, 2, 3
hipError_t launch_conv_h2w(const ConvArgs& a, int nf, hipStream_t s) {
    if (!conv_h2w_supported(a)) return hipErrorNotSupported;
    switch (nf) {
        case 1: return launch_hw<1>(a, s);
        case 2: return launch_hw<2>(a, s);
        case 3: return launch_hw<3>(a, s);
    }
    return hipErrorNotSupported;
}

}  // namespace padel
