// K3p fp16 — the stride-1 3x3 convolution of fp16 models as a patch kernel (the fp16 twin of conv_patch_bx3.hip).
//
// The fp16 tap kernels (conv_tap16.hip) fetch a 128 x 32-channel activation tile per k-step and run ONE MFMA per
// fragment pair on it: 6-12 MFMAs (96-192 matrix-pipe cycles) between two barriers, nine fetches of every input pixel —
// they live off instruction issue, not off the matrix pipe (0.2 of its peak).  Here a workgroup owns an 8 x 16 patch of
// output pixels of one image and, per 32-channel chunk,
//   * the 10 x 18 input patch (64 bytes per pixel) travels global -> LDS ONCE by LDS-DMA, double-buffered, one chunk
//     ahead (out-of-image pixels are zeros by the buffer range check; the XOR swizzle of conv_patch_bx3.hip is applied on
//     the global side: the lane that fills physical 16-byte slot s of pixel p fetches logical chunk s ^ 2*((p>>2)&1));
//   * the 9 taps are shifted 16-pixel windows of that image, read as ready-made MFMA operands (conflict-free);
//   * the weights travel through a 2-stage LDS-DMA ring whose stage holds the THREE taps of one kernel column: one barrier per 3 taps,
//     6 (2 + NF) ds_read_b128 and 6 NF MFMAs per wave between barriers.
// Weights are the fp16 [Npad][Ktot] rows of the tap kernels (K order: 64-channel chunk, tap, 32-channel half; a
// 32-channel tail block of 9 k-steps when cin % 64 == 32), fp32 accumulation, same epilogue (fp16 or fp32 output).
// LDS: 2 x 12 KB patch + 2 x 3 x BN x 64 B: 48 KB for BN = 64 -> 3 workgroups per CU; 60 KB for BN = 96 -> 2.
#include "kernels.h"
#include "act_fast.h"
#include "f16_epilogue.h"
#include <cmath>
#include <cstdint>

namespace padel {

namespace {

typedef float p16_f32x4 __attribute__((ext_vector_type(4)));
typedef int p16_i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 p16_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 p16_h4 __attribute__((ext_vector_type(4)));

constexpr int kPW16 = 18, kNPix16 = 180;
constexpr int kPieces16 = kNPix16 * 4;          // 16-byte pieces of a 32-channel fp16 patch
constexpr int kPatch16B = 768 * 16;             // 3 passes of 256 lanes (the last 48 pieces are padding)
constexpr unsigned kOOR16 = 0xFFFFFFF0u;

__device__ __forceinline__ p16_i32x4 p16_rsrc(const void* base) {
    const unsigned long long b = (unsigned long long)(uintptr_t)base;
    p16_i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xFFFFu));
    r[2] = (int)0x80000000u;
    r[3] = 0x00020000;
    return r;
}
template <int LDS_IMM>
__device__ __forceinline__ void p16_dma(unsigned voff, p16_i32x4 rsrc, unsigned soff, unsigned lds_wave) {
    asm volatile("s_add_u32 m0, %[lb], %[imm]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[vo], %[rs], %[so] offen lds"
                 :
                 : [lb] "s"(lds_wave), [imm] "n"(LDS_IMM), [vo] "v"(voff), [rs] "s"(rsrc), [so] "s"(soff)
                 : "memory", "scc");
}
// request k of a wave of the quad kernel: LDS span (wave + 3 k) -> immediate 3072 k on top of the wave's base
template <int NF>
__device__ __forceinline__ void q16_dma_k(int k, unsigned voff, p16_i32x4 rsrc, unsigned soff, unsigned lds_wave) {
    switch (k) {
        case 0: p16_dma<0>(voff, rsrc, soff, lds_wave); break;
        case 1: p16_dma<3072>(voff, rsrc, soff, lds_wave); break;
        case 2: p16_dma<6144>(voff, rsrc, soff, lds_wave); break;
        case 3: p16_dma<9216>(voff, rsrc, soff, lds_wave); break;
        case 4: p16_dma<12288>(voff, rsrc, soff, lds_wave); break;
        case 5: p16_dma<15360>(voff, rsrc, soff, lds_wave); break;
    }
}
__device__ __forceinline__ void p16_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned p16_off(int p, int q) { return (unsigned)(p * 64 + ((q ^ (((p >> 2) & 1) << 1)) << 4)); }
// epilogue: f16_epilogue.h (shared with conv_tap16.hip)

}  // namespace

template <int NF>
__global__ void __launch_bounds__(256, NF <= 4 ? 3 : 2) conv_p16_kernel(const ConvArgs a) {
    constexpr int MF = 2;
    constexpr int BN = NF * 16;
    constexpr int BTAP_B = BN * 64;               // one tap of the weight stage
    constexpr int BSTAGE_B = 3 * BTAP_B;
    constexpr int BP = (BN + 63) / 64, BFULL = BN / 64;
    static_assert(BP <= 2, "weights in at most 2 passes of 64 rows");
    __shared__ __attribute__((aligned(16))) float lds[(2 * kPatch16B + 2 * BSTAGE_B) / 4];
    char* const ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;

    // XCD-aware 1-D tile map (conv_patch_bx3.hip)
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

    // ---- patch requests: piece i * 256 + tid fills physical slot (piece & 3) of patch pixel piece >> 2
    unsigned voffP[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int piece = i * 256 + tid;
        const int pp = piece >> 2;
        const int qq = (piece & 3) ^ (((pp >> 2) & 1) << 1);
        const int py = pp / kPW16, px = pp - py * kPW16;
        const bool ok = piece < kPieces16 && (unsigned)(y0 - 1 + py) < (unsigned)a.H && (unsigned)(x0 - 1 + px) < (unsigned)a.W;
        voffP[i] = ok ? (unsigned)(((py * a.W + px) * a.in_cs + qq * 8) * 2) : kOOR16;
    }
    const _Float16* const in16 = reinterpret_cast<const _Float16*>(a.in);
    const p16_i32x4 rsrcP = p16_rsrc(in16 + (((long long)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * a.in_cs + a.in_choff);

    // ---- weights: fp16 rows of Ktot = 9 cin halves; k-step (32 channels of one tap) = 64 bytes
    const int nch = a.cin >> 5;                    // 32-channel chunks
    const int nfull64 = a.cin >> 6;
    const unsigned rowb = (unsigned)a.cin * 18u;
    const int srow = tid >> 2;
    const int sc = (tid & 3) ^ ((4 - ((srow >> 2) & 3)) & 3);
    unsigned voffB[BP];
#pragma unroll
    for (int p = 0; p < BP; ++p) {
        const int rr = srow + 64 * p;
        const int frag = min(f0 + (rr >> 4), a.n16 - 1);
        voffB[p] = (unsigned)(((frag - f0) * 16 + (rr & 15)) * rowb + sc * 16);
    }
    const p16_i32x4 rsrcB = p16_rsrc(reinterpret_cast<const char*>(a.w) + (long long)f0 * 16 * rowb);
    const bool b_last = BP > BFULL && (BFULL * 64 + wave * 16 < BN);

    const unsigned lds0 = (unsigned)(uintptr_t)lds + wave * 1024u;
    unsigned lwP0 = __builtin_amdgcn_readfirstlane(lds0), lwP1 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)kPatch16B);
    unsigned lwB0 = __builtin_amdgcn_readfirstlane(lds0 + 2u * kPatch16B), lwB1 = __builtin_amdgcn_readfirstlane(lds0 + 2u * kPatch16B + (unsigned)BSTAGE_B);
    const char* pA0 = ldsb;                        // patch image being read / being filled
    const char* pA1 = ldsb + kPatch16B;
    const int ld_off = lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);
    const float* b_rd0 = lds + (2 * kPatch16B) / 4 + ld_off;
    const float* b_rd1 = b_rd0 + BSTAGE_B / 4;
    const int rd_pix = 2 * wave * kPW16 + lr;

    p16_f32x4 acc[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[f][j] = (p16_f32x4){0.f, 0.f, 0.f, 0.f};

    // byte offset of k-step (chunk C_, tap T_) inside a weight row
#define PADEL_P16_KS(C_, T_) ((unsigned)(((C_) >> 1) < nfull64 ? ((C_) >> 1) * 18 + (T_) * 2 + ((C_) & 1) : nfull64 * 18 + (T_)) * 64u)
#define PADEL_P16_DMAB_TAP(LW_, TI_, SOFF_)                                                                       \
    do {                                                                                                          \
        const unsigned so_ = (SOFF_);                                                                             \
        if constexpr (BFULL >= 1) p16_dma<(TI_) * BTAP_B>(voffB[0], rsrcB, so_, (LW_));                           \
        if constexpr (BP > BFULL) { if (b_last) p16_dma<(TI_) * BTAP_B + BFULL * 4096>(voffB[BP - 1], rsrcB, so_, (LW_)); } \
    } while (0)
    // weight stage S_ = kernel COLUMN S_ (taps (ky, kx) = (0..2, S_): rows of the weight matrix keep the row-major tap order of
    // the tap kernels, the walk is column-major like the quad kernel's below) of chunk C_ into ring stage LW_
#define PADEL_P16_DMAB(LW_, C_, S_)                                                                               \
    do {                                                                                                          \
        PADEL_P16_DMAB_TAP(LW_, 0, PADEL_P16_KS(C_, (S_)));                                                       \
        PADEL_P16_DMAB_TAP(LW_, 1, PADEL_P16_KS(C_, 3 + (S_)));                                                   \
        PADEL_P16_DMAB_TAP(LW_, 2, PADEL_P16_KS(C_, 6 + (S_)));                                                   \
    } while (0)
#define PADEL_P16_DMAP(LW_, C_)                                                                                   \
    do {                                                                                                          \
        const unsigned so_ = (unsigned)(C_) * 64u;                                                                \
        p16_dma<0>(voffP[0], rsrcP, so_, (LW_));                                                                  \
        p16_dma<4096>(voffP[1], rsrcP, so_, (LW_));                                                               \
        p16_dma<8192>(voffP[2], rsrcP, so_, (LW_));                                                               \
    } while (0)
    // stage S_ of the current chunk: everything requested one stage earlier has landed for every wave after the barrier,
    // which also releases the other weight stage and (at S_ == 0) the other patch image
#define PADEL_P16_STAGE(S_)                                                                                       \
    do {                                                                                                          \
        p16_wait_all();                                                                                           \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
        const unsigned lwn_ = ((S_) & 1) ? lwB0 : lwB1;                                                           \
        if constexpr ((S_) < 2) { PADEL_P16_DMAB(lwn_, c, (S_) + 1); }                                            \
        else { if (c + 1 < nch) PADEL_P16_DMAB(lwn_, c + 1, 0); }                                                 \
        if constexpr ((S_) == 0) { if (c + 1 < nch) PADEL_P16_DMAP(lwP1, c + 1); }                                \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        const float* const br_ = ((S_) & 1) ? b_rd1 : b_rd0;                                                      \
        p16_h8 av[3][MF], bv[3][NF];                                                                              \
        _Pragma("unroll") for (int t = 0; t < 3; ++t) {                                                           \
            _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                        \
                av[t][f] = *reinterpret_cast<const p16_h8*>(pA0 + p16_off(rd_pix + (f + t) * kPW16 + (S_), lq));  \
            _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                        \
                bv[t][j] = __builtin_bit_cast(p16_h8, *reinterpret_cast<const p16_f32x4*>(br_ + t * (BTAP_B / 4) + j * 256)); \
        }                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                             \
            _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                        \
                _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                    \
                    acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv[t][j], av[t][f], acc[f][j], 0, 0, 0);   \
        __builtin_amdgcn_s_setprio(0);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

    PADEL_P16_DMAP(lwP0, 0);
    PADEL_P16_DMAB(lwB0, 0, 0);
    for (int c = 0; c < nch; ++c) {
        PADEL_P16_STAGE(0); PADEL_P16_STAGE(1); PADEL_P16_STAGE(2);
        // 3 stages per chunk: the weight stages swap roles, and so do the patch images
        { const float* t_ = b_rd0; b_rd0 = b_rd1; b_rd1 = t_; unsigned u_ = lwB0; lwB0 = lwB1; lwB1 = u_;
          const char* p_ = pA0; pA0 = pA1; pA1 = p_; u_ = lwP0; lwP0 = lwP1; lwP1 = u_; }
    }
    p16_wait_all();
#undef PADEL_P16_STAGE
#undef PADEL_P16_DMAP
#undef PADEL_P16_DMAB
#undef PADEL_P16_DMAB_TAP
#undef PADEL_P16_KS

    int mpix[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int oy = y0 + 2 * wave + f, ox = x0 + lr;
        mpix[f] = (oy < a.Ho && ox < a.Wo) ? (n * a.Ho + oy) * a.Wo + ox : -1;
    }
    const bool fast = y0 + 8 <= a.Ho && x0 + 16 <= a.Wo && (f0 + NF) * 16 <= a.cout && (((a.out_choff | a.out_cs) & 3) == 0) &&
                      (!a.res || (((a.res_choff | a.res_cs) & 3) == 0));
    f16_epilogue<MF, NF>(a, acc, mpix, f0, lq, fast);
}

// ---- the quad variant: a workgroup owns 16 x 16 output pixels x BN channels, a wave 4 rows x 16 pixels x ALL NF fragments.
// The 8 x 16 kernel above reads 3 (2 + NF) operands per 6 NF MFMAs and wave (0.67 per MFMA at NF = 6): 4 waves x 24 reads x
// 8 clocks = 768 LDS clocks per stage against 576 matrix-pipe cycles — it is LDS-bound.  Here the three taps of a kernel
// column share their input rows (rows r .. r + 5 of the wave's window, read once per column) and every weight fragment
// feeds 4 pixel fragments: 6 + 3 NF reads per 12 NF MFMAs (0.33 per MFMA at NF = 6, 72 MFMAs between barriers).  The
// 18 x 18 patch is requested by wave 3 (a third per stage), the weights by waves 0..2: vmcnt is in-order per wave, so no
// weight wait ever waits for patch data.  Same K walk (32-channel chunk, column, row) as the kernel above: bitwise equal.
// LDS: 2 x 21 KB patch (324 pixels, padded to 21 spans of 16) + 2 stages x 3 taps x BN x 64 B: 78 KB for BN = 96.
constexpr int kQ16PW = 18, kQ16NPix = 18 * 18;
constexpr int kQ16Spans = (kQ16NPix + 15) / 16;          // 21
constexpr int kQ16PatchB = kQ16Spans * 1024;

template <int NF>
__global__ void __launch_bounds__(256, 2) conv_p16q_kernel(const ConvArgs a) {
    constexpr int MF = 4;
    constexpr int BN = NF * 16;
    constexpr int BTAP_B = BN * 64;
    constexpr int BSTAGE_B = 3 * BTAP_B;
    static_assert(2 * kQ16PatchB + 2 * BSTAGE_B <= 80 * 1024, "2 workgroups per CU");
    static_assert(kQ16Spans == 21, "7 patch spans per stage");
    __shared__ __attribute__((aligned(16))) float lds[(2 * kQ16PatchB + 2 * BSTAGE_B) / 4];
    char* const ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;

    // XCD-aware 1-D tile map (conv_patch_bx3.hip)
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

    // ---- patch (wave 3): span s = 16 pixels x 64 bytes, lane i -> pixel 16 s + i / 4, physical slot i & 3 = logical chunk
    // (i & 3) ^ 2 ((p >> 2) & 1) of that pixel's 64 bytes
    const _Float16* const in16 = reinterpret_cast<const _Float16*>(a.in);
    const p16_i32x4 rsrcP = p16_rsrc(in16 + (((long long)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * a.in_cs + a.in_choff);
    const int p_lane = lane >> 2;
    const unsigned p_piece = (unsigned)(((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16);
    const unsigned lp0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
#define PADEL_Q16_PSPAN(S_)                                                                                       \
    do {                                                                                                          \
        const int pp_ = (S_) * 16 + pl_;                                                                          \
        const int py_ = pp_ / kQ16PW, px_ = pp_ - py_ * kQ16PW;                                                   \
        const bool ok_ = pp_ < kQ16NPix && (unsigned)(y0 - 1 + py_) < (unsigned)a.H && (unsigned)(x0 - 1 + px_) < (unsigned)a.W; \
        p16_dma<(S_) * 1024>(ok_ ? (unsigned)((py_ * a.W + px_) * a.in_cs * 2) + p_piece : kOOR16, rsrcP, so_, lb_); \
    } while (0)
    // spans 7 G_ .. 7 G_ + 6 (a third) of the patch of chunk CH_ into buffer BUF_
#define PADEL_Q16_PATCH(CH_, BUF_, G_)                                                                            \
    do {                                                                                                          \
        const unsigned so_ = (unsigned)(CH_) * 64u;                                                               \
        const unsigned lb_ = lp0 + (unsigned)(BUF_) * (unsigned)kQ16PatchB;                                       \
        int pl_ = p_lane;                                  /* recomputed per use: hoisted lane offsets would cost 21 registers */ \
        asm volatile("" : "+v"(pl_));                                                                             \
        PADEL_Q16_PSPAN(7 * (G_)); PADEL_Q16_PSPAN(7 * (G_) + 1); PADEL_Q16_PSPAN(7 * (G_) + 2); PADEL_Q16_PSPAN(7 * (G_) + 3); \
        PADEL_Q16_PSPAN(7 * (G_) + 4); PADEL_Q16_PSPAN(7 * (G_) + 5); PADEL_Q16_PSPAN(7 * (G_) + 6);              \
    } while (0)

    // ---- weights (waves 0..2): a stage = 3 taps x NF spans of 16 rows x 64 bytes; wave w requests the spans w, w + 3, ...
    // (span s = tap s / NF, row group s % NF); fp16 rows of Ktot = 9 cin halves, k-step (32 channels of one tap) = 64 bytes
    const int nch = a.cin >> 5;
    const int nfull64 = a.cin >> 6;
    const unsigned rowb = (unsigned)a.cin * 18u;
    const int b_row = lane >> 2;
    const int b_sc = (lane & 3) ^ ((4 - ((b_row >> 2) & 3)) & 3);
    unsigned voffB[NF];
    int tapB[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        const int sp = min(wave, 2) + 3 * k;
        const int g = sp % NF;
        tapB[k] = __builtin_amdgcn_readfirstlane(sp / NF);
        const int frag = min(f0 + g, a.n16 - 1);
        voffB[k] = (unsigned)(((frag - f0) * 16 + b_row) * rowb + b_sc * 16);
    }
    const p16_i32x4 rsrcB = p16_rsrc(reinterpret_cast<const char*>(a.w) + (long long)f0 * 16 * rowb);
    unsigned lwB0 = __builtin_amdgcn_readfirstlane(lp0 + 2u * kQ16PatchB + (unsigned)min(wave, 2) * 1024u);
    unsigned lwB1 = __builtin_amdgcn_readfirstlane(lwB0 + (unsigned)BSTAGE_B);
    // byte offset of k-step (chunk C_, tap T_) inside a weight row (taps row-major in memory: T_ = 3 ky + kx)
#define PADEL_Q16_KS(C_, T_) ((unsigned)(((C_) >> 1) < nfull64 ? ((C_) >> 1) * 18 + (T_) * 2 + ((C_) & 1) : nfull64 * 18 + (T_)) * 64u)
    // column KX_ of chunk C_ into ring stage LW_: this wave's NF spans
#define PADEL_Q16_DMAB(LW_, C_, KX_)                                                                              \
    do {                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < NF; ++k) q16_dma_k<NF>(k, voffB[k], rsrcB, PADEL_Q16_KS(C_, 3 * tapB[k] + (KX_)), (LW_)); \
    } while (0)
    const int ld_off = lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);
    const float* b_rd0 = lds + (2 * kQ16PatchB) / 4 + ld_off;
    const float* b_rd1 = b_rd0 + BSTAGE_B / 4;
    const int rd_pix = 4 * wave * kQ16PW + lr;              // patch pixel of the wave's row 0, kx = 0

    p16_f32x4 acc[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[f][j] = (p16_f32x4){0.f, 0.f, 0.f, 0.f};

    // stage S_ = kernel column S_ of the current chunk: everything requested one stage earlier has landed for every wave
    // after the barrier, which also releases the other weight stage and (at S_ == 0) the other patch buffer
#define PADEL_Q16_STAGE(S_)                                                                                       \
    do {                                                                                                          \
        int rp_ = rd_pix;                                  /* row addresses recomputed per stage */                \
        asm volatile("" : "+v"(rp_));                                                                             \
        p16_h8 av[6], bv[2][NF];                           /* weights of the tap in flight / of the next tap */       \
        if constexpr ((S_) > 0) {                          /* the patch is static inside a chunk: read under the wait */ \
            _Pragma("unroll") for (int r = 0; r < 6; ++r)                                                         \
                av[r] = *reinterpret_cast<const p16_h8*>(pbuf + p16_off(rp_ + r * kQ16PW + (S_), lq));            \
        }                                                                                                         \
        if ((S_) == 0 || wave != 3) p16_wait_all();                                                               \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
        const float* const br_ = ((S_) & 1) ? b_rd1 : b_rd0;                                                      \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) bv[0][j] = __builtin_bit_cast(p16_h8, *reinterpret_cast<const p16_f32x4*>(br_ + j * 256)); \
        if constexpr ((S_) == 0) {                                                                                \
            _Pragma("unroll") for (int r = 0; r < 6; ++r)                                                         \
                av[r] = *reinterpret_cast<const p16_h8*>(pbuf + p16_off(rp_ + r * kQ16PW, lq));                   \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        const unsigned lwn_ = ((S_) & 1) ? lwB0 : lwB1;                                                           \
        if (wave != 3) {                                                                                          \
            if constexpr ((S_) < 2) { PADEL_Q16_DMAB(lwn_, c, (S_) + 1); }                                        \
            else { if (c + 1 < nch) PADEL_Q16_DMAB(lwn_, c + 1, 0); }                                             \
        } else if (c + 1 < nch) {                                                                                 \
            PADEL_Q16_PATCH(c + 1, (c + 1) & 1, S_);                                                              \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int t = 0; t < 3; ++t) {                                                           \
            if (t < 2) {                                   /* next tap's weights under this tap's MFMAs */           \
                _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                    \
                    bv[(t + 1) & 1][j] = __builtin_bit_cast(p16_h8, *reinterpret_cast<const p16_f32x4*>(br_ + (t + 1) * (BTAP_B / 4) + j * 256)); \
            }                                                                                                     \
            _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                        \
                _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                    \
                    acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv[t & 1][j], av[f + t], acc[f][j], 0, 0, 0); \
        }                                                                                                         \
        __builtin_amdgcn_s_setprio(0);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

    if (wave == 3) { PADEL_Q16_PATCH(0, 0, 0); PADEL_Q16_PATCH(0, 0, 1); PADEL_Q16_PATCH(0, 0, 2); }
    else PADEL_Q16_DMAB(lwB0, 0, 0);
    for (int c = 0; c < nch; ++c) {
        const char* const pbuf = ldsb + (c & 1) * kQ16PatchB;
        PADEL_Q16_STAGE(0); PADEL_Q16_STAGE(1); PADEL_Q16_STAGE(2);
        // 3 stages per chunk: the weight stages swap roles
        { const float* t_ = b_rd0; b_rd0 = b_rd1; b_rd1 = t_; const unsigned u_ = lwB0; lwB0 = lwB1; lwB1 = u_; }
    }
    p16_wait_all();
#undef PADEL_Q16_STAGE
#undef PADEL_Q16_DMAB
#undef PADEL_Q16_KS
#undef PADEL_Q16_PATCH
#undef PADEL_Q16_PSPAN

    int mpix[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int oy = y0 + 4 * wave + f, ox = x0 + lr;
        mpix[f] = (oy < a.Ho && ox < a.Wo) ? (n * a.Ho + oy) * a.Wo + ox : -1;
    }
    const bool fast = y0 + 16 <= a.Ho && x0 + 16 <= a.Wo && (f0 + NF) * 16 <= a.cout && (((a.out_choff | a.out_cs) & 3) == 0) &&
                      (!a.res || (((a.res_choff | a.res_cs) & 3) == 0));
    f16_epilogue<MF, NF>(a, acc, mpix, f0, lq, fast);
}

template <int NF>
static hipError_t launch_p16q(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    const int batch = a.M / (a.Ho * a.Wo);
    a.n_mtiles = batch * ((a.Ho + 15) / 16) * ((a.Wo + 15) / 16);
    a.n_ntiles = (a.n16 + NF - 1) / NF;
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    hipLaunchKernelGGL((conv_p16q_kernel<NF>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

template <int NF>
static hipError_t launch_p16(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    const int batch = a.M / (a.Ho * a.Wo);
    a.n_mtiles = batch * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
    a.n_ntiles = (a.n16 + NF - 1) / NF;
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    hipLaunchKernelGGL((conv_p16_kernel<NF>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

bool conv_p16_supported(const ConvArgs& a) {
    return a.ksize == 3 && a.stride == 1 && (a.cin & 31) == 0 && a.cin >= 32 && a.Ho == a.H && a.Wo == a.W;
}

// nf = channel fragments (of 16) per workgroup: 3, 4 (3 workgroups per CU) or 6; + 20: the quad kernel (16 x 16 pixels)
hipError_t launch_conv_p16(const ConvArgs& a, int nf, hipStream_t s) {
    if (!conv_p16_supported(a)) return hipErrorNotSupported;
    switch (nf) {
        case 3: return launch_p16<3>(a, s);
        case 4: return launch_p16<4>(a, s);
        case 6: return launch_p16<6>(a, s);
        case 23: return launch_p16q<3>(a, s);       // quad: 16 x 16 pixels x 48 channels
        case 24: return launch_p16q<4>(a, s);       //                       x 64
        case 26: return launch_p16q<6>(a, s);       //                       x 96
    }
    return hipErrorNotSupported;
}

}  // namespace padel
