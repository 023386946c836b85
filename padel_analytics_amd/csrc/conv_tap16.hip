// K3/K4 fp16 — the tap-unrolled LDS-DMA ring (conv_tap.hip) on 16-bit operands: fp16 activations and weights,
// fp32 accumulation on v_mfma_f32_16x16x32_f16 (BASELINE configs[4]: "1920x1080 fp16 ... HBM-bound stress";
// the reference itself runs half=False, players_tracker.py:351-359, so this path reports its own L-inf).
//
// What carries over unchanged from the fp32 kernel is everything that moves bytes: a k-step is still one 64-byte
// run per tile row (now 32 channels instead of 16), the XOR-swizzled LDS image, the 3-/4-stage ring filled by
// `buffer_load_dwordx4 ... offen lds` with per-lane tap offsets whose out-of-image / past-M values are out of range
// (zeros by the raw-buffer range check), the wave-uniform SGPR offsets, the counted vmcnt + one raw s_barrier per
// k-step, the XCD-aware tile map.  A 3x3 accumulation block is one 64-channel chunk (9 taps x 2 halves = 18
// k-steps), cin % 64 == 32 adds the 9-step tail block; the 1x1 kernel walks 32-channel k-steps.
//
// What changes is the arithmetic per byte: a lane's 16-byte fragment is 8 halves = ONE 16x16x32 MFMA (8 passes)
// instead of four 16x16x4 fp32 MFMAs (32 passes), so the matrix pipe has ~16x less work per k-step and the kernel
// lives off instruction issue and HBM.  Consequences built in here:
//   * operands are swapped (A := weights, B := pixels), so a lane ends up with 4 CONSECUTIVE CHANNELS of one pixel:
//     the epilogue does one 8-byte bias load, one 8-byte residual load and one 8-byte store per fragment instead of
//     four scattered 2-byte ones;
//   * single-level accumulation (fp32 accumulators over K <= 5184: the two-level scheme bought parity margin at the
//     fp32 noise floor, irrelevant at fp16 input precision) — no partial set, no flush;
//   * convs that feed the Detect/Pose decode write fp32 (ConvArgs::out_f32): the head maps stay fp32, so DFL
//     softmax / box decode / NMS are the same kernels and the same arithmetic as on the fp32 path.
#include "kernels.h"
#include "act_fast.h"
#include "f16_epilogue.h"
#include <cmath>
#include <cstdint>

namespace padel {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ i32x4 make_rsrc16(const void* base) {
    const unsigned long long b = (unsigned long long)(uintptr_t)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xFFFFu));
    r[2] = (int)0x80000000u;
    r[3] = 0x00020000;
    return r;
}
constexpr unsigned kOutOfRange16 = 0xFFFFFFF0u;

template <int LDS_IMM>
__device__ __forceinline__ void dma16h(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds_wave) {
    asm volatile("s_add_u32 m0, %[lb], %[imm]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[vo], %[rs], %[so] offen lds"
                 :
                 : [lb] "s"(lds_wave), [imm] "n"(LDS_IMM), [vo] "v"(voff), [rs] "s"(rsrc), [so] "s"(soff)
                 : "memory", "scc");
}

template <int N>
__device__ __forceinline__ void wait_vm16() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ int fastdiv16(int n, unsigned magic, unsigned shift) {
    return (int)((__umulhi((unsigned)n, magic) + (unsigned)n) >> shift);
}

constexpr int min_waves16(int nw, int frags) { return nw == 4 ? (frags <= 6 ? 4 : (frags <= 8 ? 3 : 2)) : 2; }

}  // namespace

#define PADEL_T16_DMA(SR_, SA_, SB_, VA0_, VA1_)                                                                  \
    do {                                                                                                          \
        const unsigned sa_ = (SA_), sb_ = (SB_);                                                                  \
        dma16h<(SR_) * STAGE_B>((VA0_), rsrcA, sa_, lds_wave);                                                    \
        if constexpr (AP >= 2) dma16h<(SR_) * STAGE_B + RP * 64>((VA1_), rsrcA, sa_, lds_wave);                   \
        if constexpr (BFULL >= 1) dma16h<(SR_) * STAGE_B + BM * 64>(voffB[0], rsrcB, sb_, lds_wave);              \
        if constexpr (BFULL >= 2) dma16h<(SR_) * STAGE_B + BM * 64 + RP * 64>(voffB[1], rsrcB, sb_, lds_wave);    \
        if constexpr (BP > BFULL) { if (b_last) dma16h<(SR_) * STAGE_B + BM * 64 + BFULL * RP * 64>(voffB[BP - 1], rsrcB, sb_, lds_wave); } \
    } while (0)

// fragments of ring stage ST_ -> MF * NF MFMAs (weights as the A operand: D rows = channels, D columns = pixels)
#define PADEL_T16_COMPUTE(ST_)                                                                                    \
    do {                                                                                                          \
        f32x4 A_[MF], B_[NF];                                                                                     \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) A_[f] = *reinterpret_cast<const f32x4*>(a_rd + (ST_) * STAGE + f * 256); \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) B_[j] = *reinterpret_cast<const f32x4*>(b_rd + (ST_) * STAGE + j * 256); \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                            \
            _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                        \
                acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, B_[j]), __builtin_bit_cast(h8, A_[f]), acc[f][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)

// epilogue: f16_epilogue.h (shared with conv_patch16.hip)

#define PADEL_T16_GEOMETRY(NST_)                                                                                  \
    constexpr int NW = WM * WN;                                                                                   \
    constexpr int RP = NW * 16;                                                                                   \
    constexpr int BM = WM * MF * 16, BN = WN * NF * 16;                                                           \
    constexpr int AP = BM / RP, BP = (BN + RP - 1) / RP, BFULL = BN / RP;                                         \
    constexpr int NST = (NST_);                                                                                   \
    constexpr int STAGE = (BM + BN) * 16;    /* 4-byte words per ring stage: A rows then B rows, 64 bytes each */ \
    constexpr int STAGE_B = STAGE * 4;                                                                            \
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");                                              \
    static_assert(BM % RP == 0 && AP <= 2 && BFULL <= 2, "A in 1-2 full passes, B in at most 2 full + 1 partial"); \
    __shared__ __attribute__((aligned(16))) float lds[NST * STAGE];                                               \
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
    const int n0 = fastdiv16(m0, a.howo_magic, a.howo_shift), rem0 = m0 - n0 * HoWo;                              \
    const int oy0 = fastdiv16(rem0, a.wo_magic, a.wo_shift), ox0 = rem0 - oy0 * a.Wo;                             \
    const long long lin0 = ((long long)n0 * a.H + oy0 * a.stride) * a.W + ox0 * a.stride;                         \
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + wave * 1024u);            \
    const int ld_off = lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);                                       \
    const float* const a_rd = lds + (wm * MF * 16) * 16 + ld_off;                                                 \
    const float* const b_rd = lds + BM * 16 + (wn * NF * 16) * 16 + ld_off;                                       \
    const bool b_last = BP > BFULL && (BFULL * RP + wave * 16 < BN);                                              \
    const _Float16* const in16 = reinterpret_cast<const _Float16*>(a.in);                                         \
    const _Float16* const w16 = reinterpret_cast<const _Float16*>(a.w);                                           \
    f32x4 acc[MF][NF];                                                                                            \
    _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                                \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define PADEL_T16_WEIGHTS()                                                                                       \
    unsigned voffB[BP];                                                                                           \
    _Pragma("unroll") for (int p = 0; p < BP; ++p) {                                                              \
        const int rr = srow + RP * p;                                                                             \
        const int frag = min(f0 + (rr >> 4), a.n16 - 1);                                                          \
        voffB[p] = (unsigned)((((frag - f0) * 16 + (rr & 15)) * Ktot) * 2 + sc * 16);                             \
    }                                                                                                             \
    const i32x4 rsrcB = make_rsrc16(w16 + (long long)f0 * 16 * Ktot);

#define PADEL_T16_FINISH()                                                                                        \
    const bool fast_ = m0 + BM <= a.M && (f0 + WN * NF) * 16 <= a.cout && ((a.out_choff & 3) == 0) &&             \
                       ((a.out_cs & 3) == 0) && (!a.res || (((a.res_choff | a.res_cs) & 3) == 0));                \
    int mpix_[MF];                                                                                                \
    _Pragma("unroll") for (int f = 0; f < MF; ++f) { const int m_ = m0 + wm * MF * 16 + f * 16 + lr; mpix_[f] = m_ < a.M ? m_ : -1; } \
    f16_epilogue<MF, NF>(a, acc, mpix_, f0 + wn * NF, lq, fast_);

// =====================================================================================================  3x3
template <int WM, int WN, int MF, int NF>
__global__ void __launch_bounds__(64 * WM * WN, min_waves16(WM * WN, MF * NF)) conv_tap16_kernel(const ConvArgs a) {
    PADEL_T16_GEOMETRY(3)
    const int nfull = a.cin >> 6;            // 64-channel chunks: 18 k-steps each
    const bool has_tail = (a.cin & 32) != 0; // + 9 k-steps of the last 32 channels
    const int Ktot = (nfull * 18 + (has_tail ? 9 : 0)) * 32;      // halves per weight row

    unsigned voffA[AP][9];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        int m = m0 + srow + RP * p;
        const bool rv = m < a.M;
        if (!rv) m = m0;
        const int n = fastdiv16(m, a.howo_magic, a.howo_shift);
        const int rem = m - n * HoWo;
        const int oy = fastdiv16(rem, a.wo_magic, a.wo_shift);
        const int ox = rem - oy * a.Wo;
        const long long lin = ((long long)n * a.H + oy * a.stride) * a.W + ox * a.stride;
        const unsigned off = (unsigned)((lin - lin0) * a.in_cs * 2 + sc * 16);
        bool vy[3], vx[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            vy[d] = rv && (unsigned)(oy * a.stride - 1 + d) < (unsigned)a.H;
            vx[d] = (unsigned)(ox * a.stride - 1 + d) < (unsigned)a.W;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) voffA[p][t] = (vy[t / 3] && vx[t % 3]) ? off : kOutOfRange16;
    }
    const i32x4 rsrcA = make_rsrc16(in16 + ((lin0 - (a.W + 1)) * a.in_cs + a.in_choff));
    unsigned tapoff[18];
#pragma unroll
    for (int j = 0; j < 18; ++j) {
        const int t = j >> 1, ky = t / 3, kx = t % 3;
        tapoff[j] = __builtin_amdgcn_readfirstlane((unsigned)((ky * a.W + kx) * a.in_cs * 2 + (j & 1) * 64));
    }
    PADEL_T16_WEIGHTS()

    unsigned s_chunk = 0, s_kb = 0;
    bool nxt_tail = nfull == 0;
    unsigned wrap_off1 = nxt_tail ? tapoff[2] : tapoff[1];
    unsigned wrapv1[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) wrapv1[p] = nxt_tail ? voffA[p][1] : voffA[p][0];

#define PADEL_T16_STEP(J)                                                                                         \
    do {                                                                                                          \
        wait_vm16<AP + BFULL>();                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                             \
        if constexpr ((J) + 2 < 18)                                                                               \
            PADEL_T16_DMA(((J) + 2) % 3, s_chunk + tapoff[((J) + 2) % 18], s_kb + ((J) + 2) * 64u,                \
                          voffA[0][(((J) + 2) % 18) >> 1], voffA[AP - 1][(((J) + 2) % 18) >> 1]);                 \
        else if constexpr ((J) == 16)                                                                             \
            PADEL_T16_DMA(0, s_chunk + 128u + tapoff[0], s_kb + 18 * 64u, voffA[0][0], voffA[AP - 1][0]);         \
        else                                                                                                      \
            PADEL_T16_DMA(1, s_chunk + 128u + wrap_off1, s_kb + 19 * 64u, wrapv1[0], wrapv1[AP - 1]);             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_T16_COMPUTE((J) % 3);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#define PADEL_T16_TSTEP(J)                                                                                        \
    do {                                                                                                          \
        if constexpr ((J) == 8) wait_vm16<0>(); else wait_vm16<AP + BFULL>();                                     \
        __builtin_amdgcn_s_barrier();                                                                             \
        if constexpr ((J) + 2 < 9)                                                                                \
            PADEL_T16_DMA(((J) + 2) % 3, s_chunk + tapoff[2 * ((J) + 2 < 9 ? (J) + 2 : 0)], s_kb + ((J) + 2) * 64u, \
                          voffA[0][(J) + 2 < 9 ? (J) + 2 : 0], voffA[AP - 1][(J) + 2 < 9 ? (J) + 2 : 0]);         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_T16_COMPUTE((J) % 3);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

    PADEL_T16_DMA(0, tapoff[0], 0u, voffA[0][0], voffA[AP - 1][0]);
    PADEL_T16_DMA(1, wrap_off1, 64u, wrapv1[0], wrapv1[AP - 1]);
    for (int c = 0; c < nfull; ++c) {
        nxt_tail = has_tail && c == nfull - 1;
        wrap_off1 = nxt_tail ? tapoff[2] : tapoff[1];
#pragma unroll
        for (int p = 0; p < AP; ++p) wrapv1[p] = nxt_tail ? voffA[p][1] : voffA[p][0];
        PADEL_T16_STEP(0);  PADEL_T16_STEP(1);  PADEL_T16_STEP(2);  PADEL_T16_STEP(3);  PADEL_T16_STEP(4);  PADEL_T16_STEP(5);
        PADEL_T16_STEP(6);  PADEL_T16_STEP(7);  PADEL_T16_STEP(8);  PADEL_T16_STEP(9);  PADEL_T16_STEP(10); PADEL_T16_STEP(11);
        PADEL_T16_STEP(12); PADEL_T16_STEP(13); PADEL_T16_STEP(14); PADEL_T16_STEP(15); PADEL_T16_STEP(16); PADEL_T16_STEP(17);
        s_chunk += 128u;
        s_kb += 18u * 64u;
    }
    if (has_tail) {
        PADEL_T16_TSTEP(0); PADEL_T16_TSTEP(1); PADEL_T16_TSTEP(2); PADEL_T16_TSTEP(3); PADEL_T16_TSTEP(4);
        PADEL_T16_TSTEP(5); PADEL_T16_TSTEP(6); PADEL_T16_TSTEP(7); PADEL_T16_TSTEP(8);
    } else {
        wait_vm16<0>();
    }
    PADEL_T16_FINISH()
#undef PADEL_T16_STEP
#undef PADEL_T16_TSTEP
}

// =====================================================================================================  1x1
constexpr int tap16_1_min_waves(int nw, int frags, int stage_bytes) {
    const int by_lds = (160 * 1024 / (4 * stage_bytes)) * nw / 4;
    const int want = min_waves16(nw, frags);
    return by_lds < want ? (by_lds < 1 ? 1 : by_lds) : want;
}

template <int WM, int WN, int MF, int NF>
__global__ void __launch_bounds__(64 * WM * WN, tap16_1_min_waves(WM * WN, MF * NF, (WM * MF + WN * NF) * 16 * 64)) conv_tap16_1_kernel(const ConvArgs a) {
    PADEL_T16_GEOMETRY(4)
    const int nks = a.cin >> 5;              // 32 channels per k-step
    const int Ktot = nks * 32;

    unsigned voffA[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        int m = m0 + srow + RP * p;
        const bool rv = m < a.M;
        if (!rv) m = m0;
        const int n = fastdiv16(m, a.howo_magic, a.howo_shift);
        const int rem = m - n * HoWo;
        const int oy = fastdiv16(rem, a.wo_magic, a.wo_shift);
        const int ox = rem - oy * a.Wo;
        const long long lin = ((long long)n * a.H + oy * a.stride) * a.W + ox * a.stride;
        voffA[p] = rv ? (unsigned)((lin - lin0) * a.in_cs * 2 + sc * 16) : kOutOfRange16;
    }
    const i32x4 rsrcA = make_rsrc16(in16 + (lin0 * a.in_cs + a.in_choff));
    PADEL_T16_WEIGHTS()

    unsigned s_k = 0;
    constexpr int PD = 3;                     // whole 4-stage ring in flight: the matrix pipe is no cover here
#define PADEL_T16_1STEP(J)                                                                                        \
    if ((J) < nb) {                                                                                               \
        wait_vm16<(PD - 1) * (AP + BFULL)>();                                                                     \
        __builtin_amdgcn_s_barrier();                                                                             \
        PADEL_T16_DMA(((J) + PD) % 4, s_k + ((J) + PD) * 64u, s_k + ((J) + PD) * 64u, voffA[0], voffA[AP - 1]);   \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_T16_COMPUTE((J) % 4);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    }
    PADEL_T16_DMA(0, 0u, 0u, voffA[0], voffA[AP - 1]);
    PADEL_T16_DMA(1, 64u, 64u, voffA[0], voffA[AP - 1]);
    PADEL_T16_DMA(2, 128u, 128u, voffA[0], voffA[AP - 1]);
    for (int k = 0; k < nks; k += 16) {
        const int nb = min(16, nks - k);
        PADEL_T16_1STEP(0)  PADEL_T16_1STEP(1)  PADEL_T16_1STEP(2)  PADEL_T16_1STEP(3)
        PADEL_T16_1STEP(4)  PADEL_T16_1STEP(5)  PADEL_T16_1STEP(6)  PADEL_T16_1STEP(7)
        PADEL_T16_1STEP(8)  PADEL_T16_1STEP(9)  PADEL_T16_1STEP(10) PADEL_T16_1STEP(11)
        PADEL_T16_1STEP(12) PADEL_T16_1STEP(13) PADEL_T16_1STEP(14) PADEL_T16_1STEP(15)
        s_k += 16u * 64u;
    }
    wait_vm16<0>();
    PADEL_T16_FINISH()
#undef PADEL_T16_1STEP
}

// ===================================================================================================== double steps
// The same kernels with 64-channel k-steps: two 64-byte sub-rows per tile row and step (A0 | A1 | B0 | B1), i.e. two
// MFMAs per fragment pair between consecutive barriers.  The fp16 MFMAs are so short (16 cycles) that the fixed cost
// of a k-step — counted wait, barrier, DMA issue — dominates the single-step kernels; doubling the work per step
// amortises it.  No repacking: the two halves of a (chunk, tap) are adjacent 64-byte runs in pixels and weight rows
// alike.  cin % 64 == 32: the 3x3 kernel's tail block pairs taps (A0 = tap 2t, A1 = tap 2t+1 of the last 32
// channels; its weight rows are already tap-major), the 1x1 kernel switches the last step's second sub-row off —
// for A AND for the weights (the bytes behind a weight row are not weights: 0 x NaN-pattern would poison the sum).
#define PADEL_T16D_DMA(SR_, SA0_, SA1_, SB0_, SB1_, VA00_, VA01_, VA10_, VA11_, VBX_)                             \
    do {                                                                                                          \
        const unsigned sa0_ = (SA0_), sa1_ = (SA1_), sb0_ = (SB0_), sb1_ = (SB1_);                                \
        dma16h<(SR_) * STAGE_B>((VA00_), rsrcA, sa0_, lds_wave);                                                  \
        if constexpr (AP >= 2) dma16h<(SR_) * STAGE_B + RP * 64>((VA01_), rsrcA, sa0_, lds_wave);                 \
        dma16h<(SR_) * STAGE_B + BM * 64>((VA10_), rsrcA, sa1_, lds_wave);                                        \
        if constexpr (AP >= 2) dma16h<(SR_) * STAGE_B + BM * 64 + RP * 64>((VA11_), rsrcA, sa1_, lds_wave);       \
        if constexpr (BFULL >= 1) dma16h<(SR_) * STAGE_B + 2 * BM * 64>(voffB[0], rsrcB, sb0_, lds_wave);         \
        if constexpr (BFULL >= 2) dma16h<(SR_) * STAGE_B + 2 * BM * 64 + RP * 64>(voffB[1], rsrcB, sb0_, lds_wave); \
        if constexpr (BP > BFULL) { if (b_last) dma16h<(SR_) * STAGE_B + 2 * BM * 64 + BFULL * RP * 64>(voffB[BP - 1], rsrcB, sb0_, lds_wave); } \
        if constexpr (BFULL >= 1) dma16h<(SR_) * STAGE_B + (2 * BM + BN) * 64>((VBX_) ? kOutOfRange16 : voffB[0], rsrcB, sb1_, lds_wave); \
        if constexpr (BFULL >= 2) dma16h<(SR_) * STAGE_B + (2 * BM + BN) * 64 + RP * 64>((VBX_) ? kOutOfRange16 : voffB[1], rsrcB, sb1_, lds_wave); \
        if constexpr (BP > BFULL) { if (b_last) dma16h<(SR_) * STAGE_B + (2 * BM + BN) * 64 + BFULL * RP * 64>((VBX_) ? kOutOfRange16 : voffB[BP - 1], rsrcB, sb1_, lds_wave); } \
    } while (0)

#define PADEL_T16D_COMPUTE(ST_)                                                                                   \
    do {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int sub = 0; sub < 2; ++sub) {                                                     \
            f32x4 A_[MF], B_[NF];                                                                                 \
            _Pragma("unroll") for (int f = 0; f < MF; ++f) A_[f] = *reinterpret_cast<const f32x4*>(a_rd + (ST_) * STAGE + sub * BM * 16 + f * 256); \
            _Pragma("unroll") for (int j = 0; j < NF; ++j) B_[j] = *reinterpret_cast<const f32x4*>(b_rd + (ST_) * STAGE + sub * BN * 16 + j * 256); \
            _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                        \
                _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                    \
                    acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, B_[j]), __builtin_bit_cast(h8, A_[f]), acc[f][j], 0, 0, 0); \
        }                                                                                                         \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)

#define PADEL_T16D_GEOMETRY()                                                                                     \
    constexpr int NW = WM * WN;                                                                                   \
    constexpr int RP = NW * 16;                                                                                   \
    constexpr int BM = WM * MF * 16, BN = WN * NF * 16;                                                           \
    constexpr int AP = BM / RP, BP = (BN + RP - 1) / RP, BFULL = BN / RP;                                         \
    constexpr int STAGE = 2 * (BM + BN) * 16;      /* 4-byte words per ring stage: A0 | A1 | B0 | B1 */           \
    constexpr int STAGE_B = STAGE * 4;                                                                            \
    constexpr int NREQ = 2 * AP + 2 * BFULL;                                                                      \
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
    const int n0 = fastdiv16(m0, a.howo_magic, a.howo_shift), rem0 = m0 - n0 * HoWo;                              \
    const int oy0 = fastdiv16(rem0, a.wo_magic, a.wo_shift), ox0 = rem0 - oy0 * a.Wo;                             \
    const long long lin0 = ((long long)n0 * a.H + oy0 * a.stride) * a.W + ox0 * a.stride;                         \
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + wave * 1024u);            \
    const int ld_off = lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);                                       \
    const float* const a_rd = lds + (wm * MF * 16) * 16 + ld_off;                                                 \
    const float* const b_rd = lds + 2 * BM * 16 + (wn * NF * 16) * 16 + ld_off;                                   \
    const bool b_last = BP > BFULL && (BFULL * RP + wave * 16 < BN);                                              \
    const _Float16* const in16 = reinterpret_cast<const _Float16*>(a.in);                                         \
    const _Float16* const w16 = reinterpret_cast<const _Float16*>(a.w);                                           \
    f32x4 acc[MF][NF];                                                                                            \
    _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                                \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

constexpr int min_waves16d(int nw, int frags) { return nw == 4 ? (frags <= 6 ? 3 : 2) : 2; }

template <int WM, int WN, int MF, int NF>
__global__ void __launch_bounds__(64 * WM * WN, min_waves16d(WM * WN, MF * NF)) conv_tap16d_kernel(const ConvArgs a) {
    PADEL_T16D_GEOMETRY()
    const int nfull = a.cin >> 6;
    const bool has_tail = (a.cin & 32) != 0;
    const int Ktot = (nfull * 18 + (has_tail ? 9 : 0)) * 32;      // halves per weight row (same packing as the single-step kernel)

    unsigned voffA[AP][9];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        int m = m0 + srow + RP * p;
        const bool rv = m < a.M;
        if (!rv) m = m0;
        const int n = fastdiv16(m, a.howo_magic, a.howo_shift);
        const int rem = m - n * HoWo;
        const int oy = fastdiv16(rem, a.wo_magic, a.wo_shift);
        const int ox = rem - oy * a.Wo;
        const long long lin = ((long long)n * a.H + oy * a.stride) * a.W + ox * a.stride;
        const unsigned off = (unsigned)((lin - lin0) * a.in_cs * 2 + sc * 16);
        bool vy[3], vx[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            vy[d] = rv && (unsigned)(oy * a.stride - 1 + d) < (unsigned)a.H;
            vx[d] = (unsigned)(ox * a.stride - 1 + d) < (unsigned)a.W;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) voffA[p][t] = (vy[t / 3] && vx[t % 3]) ? off : kOutOfRange16;
    }
    const i32x4 rsrcA = make_rsrc16(in16 + ((lin0 - (a.W + 1)) * a.in_cs + a.in_choff));
    unsigned tapoff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tapoff[t] = __builtin_amdgcn_readfirstlane((unsigned)(((t / 3) * a.W + (t % 3)) * a.in_cs * 2));
    PADEL_T16_WEIGHTS()

    unsigned s_chunk = 0, s_kb = 0;
#define PADEL_T16D_REQ_FULL(SR_, CH_, KB_, T_)                                                                    \
    PADEL_T16D_DMA(SR_, (CH_) + tapoff[T_], (CH_) + tapoff[T_] + 64u, KB_, (KB_) + 64u,                           \
                   voffA[0][T_], voffA[AP - 1][T_], voffA[0][T_], voffA[AP - 1][T_], false)
#define PADEL_T16D_REQ_TAIL(SR_, CH_, KB_, JT_)                                                                   \
    PADEL_T16D_DMA(SR_, (CH_) + tapoff[2 * (JT_)], (CH_) + tapoff[2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8], KB_, (KB_) + 64u, \
                   voffA[0][2 * (JT_)], voffA[AP - 1][2 * (JT_)],                                                 \
                   2 * (JT_) + 1 < 9 ? voffA[0][2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8] : kOutOfRange16,           \
                   2 * (JT_) + 1 < 9 ? voffA[AP - 1][2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8] : kOutOfRange16,      \
                   2 * (JT_) + 1 >= 9)
    bool nxt_tail = false;
#define PADEL_T16D_STEP(J)                                                                                        \
    do {                                                                                                          \
        wait_vm16<NREQ>();                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                             \
        if constexpr ((J) + 2 < 9) {                                                                              \
            PADEL_T16D_REQ_FULL(((J) + 2) % 3, s_chunk, s_kb + ((J) + 2) * 128u, (J) + 2 < 9 ? (J) + 2 : 0);      \
        } else {                                                                                                  \
            if (nxt_tail) { PADEL_T16D_REQ_TAIL(((J) + 2) % 3, s_chunk + 128u, s_kb + ((J) + 2) * 128u, ((J) + 2) % 9); } \
            else { PADEL_T16D_REQ_FULL(((J) + 2) % 3, s_chunk + 128u, s_kb + ((J) + 2) * 128u, ((J) + 2) % 9); }  \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_T16D_COMPUTE((J) % 3);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#define PADEL_T16D_TSTEP(JT)                                                                                      \
    do {                                                                                                          \
        if constexpr ((JT) == 4) wait_vm16<0>(); else wait_vm16<NREQ>();                                          \
        __builtin_amdgcn_s_barrier();                                                                             \
        if constexpr ((JT) + 2 < 5) { PADEL_T16D_REQ_TAIL(((JT) + 2) % 3, s_chunk, s_kb + ((JT) + 2) * 128u, (JT) + 2 < 5 ? (JT) + 2 : 0); } \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_T16D_COMPUTE((JT) % 3);                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

    if (nfull > 0) {
        PADEL_T16D_REQ_FULL(0, 0u, 0u, 0);
        PADEL_T16D_REQ_FULL(1, 0u, 128u, 1);
    } else {
        PADEL_T16D_REQ_TAIL(0, 0u, 0u, 0);
        PADEL_T16D_REQ_TAIL(1, 0u, 128u, 1);
    }
    for (int c = 0; c < nfull; ++c) {
        nxt_tail = has_tail && c == nfull - 1;
        PADEL_T16D_STEP(0); PADEL_T16D_STEP(1); PADEL_T16D_STEP(2); PADEL_T16D_STEP(3); PADEL_T16D_STEP(4);
        PADEL_T16D_STEP(5); PADEL_T16D_STEP(6); PADEL_T16D_STEP(7); PADEL_T16D_STEP(8);
        s_chunk += 128u;
        s_kb += 9u * 128u;
    }
    if (has_tail) {
        PADEL_T16D_TSTEP(0); PADEL_T16D_TSTEP(1); PADEL_T16D_TSTEP(2); PADEL_T16D_TSTEP(3); PADEL_T16D_TSTEP(4);
    } else {
        wait_vm16<0>();
    }
    PADEL_T16_FINISH()
#undef PADEL_T16D_STEP
#undef PADEL_T16D_TSTEP
#undef PADEL_T16D_REQ_FULL
#undef PADEL_T16D_REQ_TAIL
}

template <int WM, int WN, int MF, int NF>
__global__ void __launch_bounds__(64 * WM * WN, min_waves16d(WM * WN, MF * NF)) conv_tap16d_1_kernel(const ConvArgs a) {
    PADEL_T16D_GEOMETRY()
    const int nks = (a.cin + 63) >> 6;            // 64 channels per k-step; the last one half empty if cin & 32
    const bool half_tail = (a.cin & 32) != 0;
    const int Ktot = (a.cin >> 5) * 32;

    unsigned voffA[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        int m = m0 + srow + RP * p;
        const bool rv = m < a.M;
        if (!rv) m = m0;
        const int n = fastdiv16(m, a.howo_magic, a.howo_shift);
        const int rem = m - n * HoWo;
        const int oy = fastdiv16(rem, a.wo_magic, a.wo_shift);
        const int ox = rem - oy * a.Wo;
        const long long lin = ((long long)n * a.H + oy * a.stride) * a.W + ox * a.stride;
        voffA[p] = rv ? (unsigned)((lin - lin0) * a.in_cs * 2 + sc * 16) : kOutOfRange16;
    }
    const i32x4 rsrcA = make_rsrc16(in16 + (lin0 * a.in_cs + a.in_choff));
    PADEL_T16_WEIGHTS()

    unsigned s_k = 0;
#define PADEL_T16D_OFF1(K_) (half_tail && (int)(K_) >= nks - 1)
#define PADEL_T16D_REQ1(SR_, K_)                                                                                  \
    PADEL_T16D_DMA(SR_, (K_) * 128u, (K_) * 128u + 64u, (K_) * 128u, (K_) * 128u + 64u, voffA[0], voffA[AP - 1],  \
                   PADEL_T16D_OFF1(K_) ? kOutOfRange16 : voffA[0], PADEL_T16D_OFF1(K_) ? kOutOfRange16 : voffA[AP - 1], PADEL_T16D_OFF1(K_))
#define PADEL_T16D_1STEP(J)                                                                                       \
    if ((J) < nb) {                                                                                               \
        wait_vm16<NREQ>();                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                             \
        PADEL_T16D_REQ1(((J) + 2) % 3, s_k + (J) + 2);                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_T16D_COMPUTE((J) % 3);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    }
    PADEL_T16D_REQ1(0, 0u);
    PADEL_T16D_REQ1(1, 1u);
    for (int k = 0; k < nks; k += 9) {
        const int nb = min(9, nks - k);
        PADEL_T16D_1STEP(0) PADEL_T16D_1STEP(1) PADEL_T16D_1STEP(2) PADEL_T16D_1STEP(3) PADEL_T16D_1STEP(4)
        PADEL_T16D_1STEP(5) PADEL_T16D_1STEP(6) PADEL_T16D_1STEP(7) PADEL_T16D_1STEP(8)
        s_k += 9u;
    }
    wait_vm16<0>();
    PADEL_T16_FINISH()
#undef PADEL_T16D_1STEP
#undef PADEL_T16D_REQ1
#undef PADEL_T16D_OFF1
}

template <int WM, int WN, int MF, int NF>
static hipError_t launch_t16d(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    constexpr int BM = WM * MF * 16;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.n_ntiles = (a.n16 + WN * NF - 1) / (WN * NF);
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    if (a.ksize == 3) hipLaunchKernelGGL((conv_tap16d_kernel<WM, WN, MF, NF>), grid, dim3(64 * WM * WN), 0, s, a);
    else hipLaunchKernelGGL((conv_tap16d_1_kernel<WM, WN, MF, NF>), grid, dim3(64 * WM * WN), 0, s, a);
    return hipGetLastError();
}

template <int WM, int WN, int MF, int NF>
static hipError_t launch_t16(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    constexpr int BM = WM * MF * 16;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.n_ntiles = (a.n16 + WN * NF - 1) / (WN * NF);
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    if (a.ksize == 3) hipLaunchKernelGGL((conv_tap16_kernel<WM, WN, MF, NF>), grid, dim3(64 * WM * WN), 0, s, a);
    else hipLaunchKernelGGL((conv_tap16_1_kernel<WM, WN, MF, NF>), grid, dim3(64 * WM * WN), 0, s, a);
    return hipGetLastError();
}

// tile ids: the fp32 id space (conv_variant_shape) + 30.. for the larger per-wave tiles only the fp16 path has
hipError_t launch_conv_tap16(const ConvArgs& a, int variant, hipStream_t s) {
    if ((a.ksize != 3 && a.ksize != 1) || (a.cin & 31) || a.cin < 32) return hipErrorNotSupported;
    if (variant >= 300 && variant < 400) {           // fp16 patch kernel (conv_patch16.hip), or a tap tile where it does not apply
        const int nf = variant - 300;
        if (conv_p16_supported(a)) return launch_conv_p16(a, nf, s);
        variant = nf == 3 ? 20 : nf == 4 ? 9 : 31;
    }
    switch (variant) {
        case 6: return launch_t16<2, 2, 2, 4>(a, s);    //  64 x 128
        case 7: return launch_t16<2, 2, 2, 3>(a, s);    //  64 x  96
        case 9: return launch_t16<4, 1, 2, 4>(a, s);    // 128 x  64
        case 11: return launch_t16<4, 1, 2, 2>(a, s);   // 128 x  32
        case 12: return launch_t16<4, 1, 2, 1>(a, s);   // 128 x  16
        case 20: return launch_t16<4, 1, 2, 3>(a, s);   // 128 x  48
        case 30: return launch_t16<2, 2, 4, 4>(a, s);   // 128 x 128, 4 waves of 64 x 64
        case 31: return launch_t16<2, 2, 4, 3>(a, s);   // 128 x  96
        case 32: return launch_t16<2, 2, 4, 2>(a, s);   // 128 x  64
        // + 40: the same tile with 64-channel (double) k-steps
        case 46: return launch_t16d<2, 2, 2, 4>(a, s);  //  64 x 128
        case 47: return launch_t16d<2, 2, 2, 3>(a, s);  //  64 x  96
        case 49: return launch_t16d<4, 1, 2, 4>(a, s);  // 128 x  64
        case 51: return launch_t16d<4, 1, 2, 2>(a, s);  // 128 x  32
        case 60: return launch_t16d<4, 1, 2, 3>(a, s);  // 128 x  48
        case 70: return launch_t16d<2, 2, 4, 4>(a, s);  // 128 x 128
        case 71: return launch_t16d<2, 2, 4, 3>(a, s);  // 128 x  96
        case 72: return launch_t16d<2, 2, 4, 2>(a, s);  // 128 x  64
    }
    return hipErrorNotSupported;
}

bool conv_tap16_variant_shape(int variant, int* bm, int* bn) {
    static const int t[][3] = {{6, 64, 128}, {7, 64, 96}, {9, 128, 64}, {11, 128, 32}, {12, 128, 16}, {20, 128, 48},
                               {30, 128, 128}, {31, 128, 96}, {32, 128, 64},
                               {46, 64, 128}, {47, 64, 96}, {49, 128, 64}, {51, 128, 32}, {60, 128, 48}, {70, 128, 128}, {71, 128, 96}, {72, 128, 64}};
    for (const auto& v : t)
        if (v[0] == variant) { *bm = v[1]; *bn = v[2]; return true; }
    return false;
}

// Tile choice: these kernels are issue- / HBM-bound, not MFMA-bound, so (a) the channel tile should cover all of
// cout when it can (every extra channel tile re-reads the whole input from L2 / HBM), (b) bigger per-wave tiles
// amortise the fixed per-k-step instruction cost, (c) the grid still has to fill 256 CUs.
int choose_conv_tap16_variant(const ConvArgs& a) {
    const int M = a.M, n16 = a.n16, ksize = a.ksize, cin = a.cin;
    struct V { int id, bm, nf; float speed; bool dbl; };
    // double-step tiles (ids + 40) win where K is a whole number of 64-channel steps and long enough to matter:
    // 3x3 with cin % 64 == 0 (yolov8m 192 -> 192 / 304: 684-696 vs 626 TFLOP/s, profiles/conv_tap16_sweep_r2i.txt);
    // with a 32-channel tail or on the short-K 1x1 layers the single-step tiles stay ahead
    static const V vs[] = {{30, 128, 8, 1.30f, false}, {31, 128, 6, 1.25f, false}, {32, 128, 4, 1.10f, false}, {6, 64, 8, 1.05f, false},
                           {7, 64, 6, 1.00f, false},   {9, 128, 4, 1.00f, false},  {20, 128, 3, 0.95f, false}, {11, 128, 2, 0.85f, false},
                           {12, 128, 1, 0.60f, false}, {49, 128, 4, 1.40f, true},  {72, 128, 4, 1.40f, true}};
    const bool dbl_ok = ksize == 3 && (cin & 63) == 0 && cin >= 128;
    float best = -1.f;
    int bv = 7;
    for (const V& v : vs) {
        if (v.dbl && !dbl_ok) continue;
        const int ntiles = (n16 + v.nf - 1) / v.nf;
        const long long mtiles = (M + v.bm - 1) / v.bm;
        const float fill = (float)n16 / (float)(ntiles * v.nf) * (float)M / (float)(mtiles * v.bm);
        const long long blocks = mtiles * ntiles;
        const long long per_cu = (blocks + 255) / 256;
        const float occ = (float)blocks / (256.f * (float)per_cu);
        const float reread = 1.0f / (1.0f + 0.25f * (float)(ntiles - 1));        // input re-read per extra channel tile
        const float sc = v.speed * fill * occ * (v.dbl ? 1.0f : reread);
        if (sc > best) { best = sc; bv = v.id; }
    }
    // stride-1 3x3: the patch kernel (conv_patch16.hip) fetches the input once per chunk instead of once per tap:
    // 806 vs 662 TFLOP/s on 192 -> 192, 875 vs 674 on the 256-channel heads, 558 vs 461 on 48(64) -> 64, about even on
    // the short-K 96 -> 96 layers (profiles/conv_tap16_sweep_r2x_patch.txt) — hence the K-length factor
    if (conv_p16_supported(a)) {
        struct P { int nf; float sp; };
        static const P ps[] = {{4, 1.70f}, {3, 1.55f}, {6, 1.45f}};
        const int nch = cin >> 5;
        const long long patches = (long long)(M / (a.Ho * a.Wo)) * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
        for (const P& v : ps) {
            const int ntiles = (n16 + v.nf - 1) / v.nf;
            const float fill = (float)n16 / (float)(ntiles * v.nf) * (float)M / (float)(patches * 128);
            const long long blocks = patches * ntiles;
            const long long per_cu = (blocks + 255) / 256;
            const float sc = v.sp * (float)nch / (float)(nch + 1) * fill * (float)blocks / (256.f * (float)per_cu);
            if (sc > best) { best = sc; bv = 300 + v.nf; }
        }
        // the quad kernel (16 x 16 pixels x 96 channels per workgroup): 749-770 vs 656-682 TFLOP/s on 96 -> 96, 885 vs 857 on
        // 192 -> 192; behind on partial channel tiles and on maps that do not fill 16-row tiles (profiles/r3s_sweep_p16q.txt)
        {
            const long long qpatches = (long long)(M / (a.Ho * a.Wo)) * ((a.Ho + 15) / 16) * ((a.Wo + 15) / 16);
            const int ntiles = (n16 + 5) / 6;
            const float fill = (float)n16 / (float)(ntiles * 6) * (float)M / (float)(qpatches * 256);
            const long long blocks = qpatches * ntiles;
            const long long per_cu = (blocks + 255) / 256;
            const float sc = 1.78f * (float)nch / (float)(nch + 1) * fill * (float)blocks / (256.f * (float)per_cu);
            if (sc > best) { best = sc; bv = 326; }
        }
    }
    return bv;
}

}  // namespace padel
