// K3/K4 v5 — tap-unrolled LDS-DMA ring for the 3x3 and 1x1 convs: the instruction diet.
//
// Same tiles, operands, K order, swizzled LDS image, ring and two-level accumulation as conv_ring.hip (results are
// bit-identical with v1/v2/v4).  Why another kernel: the s_memtime timeline of v2 (tools/timeline_probe.py,
// profiles/conv_timeline_r1.txt) shows that with 4-5 waves per SIMD a wave spends only ~25 % of a k-step inside its
// 24-MFMA burst; the other ~2900 cycles go to the ~110 non-MFMA instructions of the k-step (tap decode, 64-bit
// address arithmetic, bounds tests, zero-page selects, buffer toggles, branches), each of which costs ~26 cycles to
// issue while the other waves keep the matrix pipe busy.  Here the 18 k-steps of one 32-channel chunk (9 taps x 2
// halves = one accumulation block) are unrolled, so tap, half and ring stage are compile-time constants, and the
// loader is rebuilt around buffer addressing:
//
//   * per lane and tap ONE precomputed 32-bit byte offset (relative to the workgroup's lowest address); taps that
//     fall outside the image, and rows past M, hold an out-of-range offset: the raw-buffer range check returns
//     zeros for them — no compare, no select, no zero page in the loop;
//   * everything that changes from step to step is wave-uniform and lives in the SGPR offset of
//     `buffer_load_dwordx4 ... offen lds` (chunk + tap + half: one s_add per request);
//   * ring stage and fragment offsets are immediates of the ds_read_b128 / of the M0 add.
//
// A k-step is: s_waitcnt vmcnt(n) ; s_barrier ; 2-3 LDS-DMA requests (3 SALU each) ; 5 ds_read_b128 ; 24 MFMA.
// Requests run two steps ahead and read up to 128 bytes past the last chunk of a pixel / weight row (data that is
// never used); the engine allocates its buffers with that slack.
//
// cin % 32 == 16 (the 16-channel K tail: 9 steps of one half per tap) is a second unrolled block; the two requests
// that cross from the last full chunk into it take their tap offsets from registers selected once per chunk.
// The 1x1 kernel (conv_tap1_kernel) is the same machine with one offset per lane, a 4-stage ring (its accumulation
// block has 16 steps; 16 % 4 == 0 keeps the stage a compile-time constant) and a per-step `J < steps` guard for
// the last, shorter block.
#include "kernels.h"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace padel {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply5(float v, int act) {
    if (act == ACT_SILU) return v / (1.0f + expf(-v));
    if (act == ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    if (act == ACT_LEAKY) return v >= 0.0f ? v : 0.01f * v;
    return v;
}

// raw buffer descriptor (gfx9 family): base, stride 0, num_records = 2 GiB, 32-bit data format
__device__ __forceinline__ i32x4 make_rsrc(const float* base) {
    const unsigned long long b = (unsigned long long)(uintptr_t)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xFFFFu));
    r[2] = (int)0x80000000u;
    r[3] = 0x00020000;
    return r;
}
constexpr unsigned kOutOfRange = 0xFFFFFFF0u;      // >= num_records: the load returns zeros

// 64 lanes x 16 bytes, buffer (rsrc base + soff + per-lane voff) -> LDS (lds_wave + LDS_IMM + 16 * lane)
template <int LDS_IMM>
__device__ __forceinline__ void dma16(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds_wave) {
    asm volatile("s_add_u32 m0, %[lb], %[imm]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[vo], %[rs], %[so] offen lds"
                 :
                 : [lb] "s"(lds_wave), [imm] "n"(LDS_IMM), [vo] "v"(voff), [rs] "s"(rsrc), [so] "s"(soff)
                 : "memory", "scc");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// n / d for 0 <= n < 2^31 with the (magic, shift) pair of fill_fastdiv (kernels.h): 3 instructions instead of ~35
__device__ __forceinline__ int fastdiv(int n, unsigned magic, unsigned shift) {
    return (int)((__umulhi((unsigned)n, magic) + (unsigned)n) >> shift);
}

constexpr int tap_min_waves(int nw, int frags) {
    return nw == 4 ? (frags <= 6 ? 5 : 4) : 4;
}

// ---- pieces shared by the 3x3 and the 1x1 kernel (macros: everything must stay in registers of the caller)

// LDS-DMA requests of one k-step into ring stage SR_: A passes with per-pass lane offsets VA0_/VA1_ and SGPR offset
// SA_, B passes with SGPR offset SB_
#define PADEL_TAP_DMA(SR_, SA_, SB_, VA0_, VA1_)                                                                  \
    do {                                                                                                          \
        const unsigned sa_ = (SA_), sb_ = (SB_);                                                                  \
        dma16<(SR_) * STAGE_B>((VA0_), rsrcA, sa_, lds_wave);                                                     \
        if constexpr (AP >= 2) dma16<(SR_) * STAGE_B + RP * 64>((VA1_), rsrcA, sa_, lds_wave);                    \
        if constexpr (BFULL >= 1) dma16<(SR_) * STAGE_B + BM * 64>(voffB[0], rsrcB, sb_, lds_wave);               \
        if constexpr (BFULL >= 2) dma16<(SR_) * STAGE_B + BM * 64 + RP * 64>(voffB[1], rsrcB, sb_, lds_wave);     \
        if constexpr (BP > BFULL) { if (b_last) dma16<(SR_) * STAGE_B + BM * 64 + BFULL * RP * 64>(voffB[BP - 1], rsrcB, sb_, lds_wave); } \
    } while (0)

// fragments of ring stage ST_ -> MF * NF * 4 MFMAs on `part`
#define PADEL_TAP_COMPUTE(ST_)                                                                                    \
    do {                                                                                                          \
        f32x4 A_[MF], B_[NF];                                                                                     \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) A_[f] = *reinterpret_cast<const f32x4*>(a_rd + (ST_) * STAGE + f * 256); \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) B_[j] = *reinterpret_cast<const f32x4*>(b_rd + (ST_) * STAGE + j * 256); \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                          \
            _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                        \
                _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                    \
                    part[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[f][kk], B_[j][kk], part[f][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)

// the same two halves separately (timeline instantiation only)
#define PADEL_TAP_FRAGS_DBG(ST_)                                                                                  \
    f32x4 A_[MF], B_[NF];                                                                                         \
    _Pragma("unroll") for (int f = 0; f < MF; ++f) A_[f] = *reinterpret_cast<const f32x4*>(a_rd + (ST_) * STAGE + f * 256); \
    _Pragma("unroll") for (int j = 0; j < NF; ++j) B_[j] = *reinterpret_cast<const f32x4*>(b_rd + (ST_) * STAGE + j * 256);
#define PADEL_TAP_MFMAS_DBG()                                                                                     \
    do {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                          \
            _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                        \
                _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                    \
                    part[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[f][kk], B_[j][kk], part[f][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)

#define PADEL_TAP_FLUSH()                                                                                         \
    do {                                                                                                          \
        _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                            \
            _Pragma("unroll") for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; } \
    } while (0)

// epilogue: lane holds D[row = lq*4 + r][col = lr] of each 16x16 fragment.  Under MFMA contention every epilogue
// instruction costs a wave ~26 cycles (profiles/conv_timeline_r1.txt), so the activation switch, the residual test
// and the tile-edge predicates are hoisted into wave-uniform template cases instead of being re-decided per element
// (same arithmetic per element in every case: results do not depend on which case runs).
template <int MF, int NF, int ACT, bool RES, bool FULL>
__device__ __forceinline__ void tap_epilogue_case(const ConvArgs& a, const f32x4 (&acc)[MF][NF], int mw, int fw, int lr, int lq) {
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int fr = fw + j;
        const int co = fr * 16 + lr;
        const bool cv = FULL || co < a.cout;
        const float b = a.bias[(FULL ? fr : min(fr, a.n16 - 1)) * 16 + lr];
#pragma unroll
        for (int f = 0; f < MF; ++f) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int m = mw + f * 16 + lq * 4 + rr;
                if (FULL || (cv && m < a.M)) {
                    float v = acc[f][j][rr] + b;
                    if (ACT == ACT_SILU) v = v / (1.0f + expf(-v));
                    else if (ACT == ACT_RELU) v = v > 0.0f ? v : 0.0f;
                    else if (ACT == ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                    else if (ACT == ACT_LEAKY) v = v >= 0.0f ? v : 0.01f * v;
                    if (RES) v += a.res[(long long)m * a.res_cs + a.res_choff + co];
                    a.out[(long long)m * a.out_cs + a.out_choff + co] = v;
                }
            }
        }
    }
}

template <int MF, int NF, int ACT>
__device__ __forceinline__ void tap_epilogue_act(const ConvArgs& a, const f32x4 (&acc)[MF][NF], int mw, int fw, int lr, int lq, bool full) {
    if (a.res) {
        if (full) tap_epilogue_case<MF, NF, ACT, true, true>(a, acc, mw, fw, lr, lq);
        else tap_epilogue_case<MF, NF, ACT, true, false>(a, acc, mw, fw, lr, lq);
    } else {
        if (full) tap_epilogue_case<MF, NF, ACT, false, true>(a, acc, mw, fw, lr, lq);
        else tap_epilogue_case<MF, NF, ACT, false, false>(a, acc, mw, fw, lr, lq);
    }
}

// `full`: every row and channel of the WORKGROUP's tile exists (wave-uniform)
template <int MF, int NF>
__device__ __forceinline__ void tap_epilogue(const ConvArgs& a, const f32x4 (&acc)[MF][NF], int mw, int fw, int lr, int lq, bool full) {
    if (a.act == ACT_SILU) tap_epilogue_act<MF, NF, ACT_SILU>(a, acc, mw, fw, lr, lq, full);
    else if (a.act == ACT_RELU) tap_epilogue_act<MF, NF, ACT_RELU>(a, acc, mw, fw, lr, lq, full);
    else if (a.act == ACT_SIGMOID) tap_epilogue_act<MF, NF, ACT_SIGMOID>(a, acc, mw, fw, lr, lq, full);
    else if (a.act == ACT_LEAKY) tap_epilogue_act<MF, NF, ACT_LEAKY>(a, acc, mw, fw, lr, lq, full);
    else tap_epilogue_act<MF, NF, ACT_NONE>(a, acc, mw, fw, lr, lq, full);
}

// common head of both kernels: tile geometry, wave / lane ids, XCD-aware tile index
#define PADEL_TAP_GEOMETRY(NST_)                                                                                  \
    constexpr int NW = WM * WN;              /* waves per workgroup (4 or 8) */                                   \
    constexpr int RP = NW * 16;              /* tile rows staged per pass: one wave-instruction (16 rows) per wave */ \
    constexpr int BM = WM * MF * 16, BN = WN * NF * 16;                                                           \
    constexpr int AP = BM / RP, BP = (BN + RP - 1) / RP, BFULL = BN / RP;                                         \
    constexpr int NST = (NST_);                                                                                   \
    constexpr int STAGE = (BM + BN) * 16;    /* floats per ring stage: A rows then B rows, 64 bytes each */       \
    constexpr int STAGE_B = STAGE * 4;                                                                            \
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");                                              \
    static_assert(BM % RP == 0 && AP <= 2 && BFULL <= 2, "A in 1-2 full passes, B in at most 2 full + 1 partial"); \
    __shared__ __attribute__((aligned(16))) float lds[NST * STAGE];                                               \
    const int tid = threadIdx.x;                                                                                  \
    const int lane = tid & 63;                                                                                    \
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                    \
    const int lr = lane & 15, lq = lane >> 4;                                                                     \
    const int wm = wave / WN, wn = wave % WN;                                                                     \
    const int nmt = a.n_mtiles;                                                                                   \
    const int bid = blockIdx.x;                                                                                   \
    const int q = nmt >> 3, r = nmt & 7, xcd = bid & 7, idx = bid >> 3;                                           \
    const int mt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;                                 \
    const int m0 = mt * BM;                                                                                       \
    const int f0 = blockIdx.y * (WN * NF);   /* first 16-channel fragment of this workgroup */                    \
    const int HoWo = a.Ho * a.Wo;                                                                                 \
    const int srow = tid >> 2;                                                                                    \
    const int sc = (tid & 3) ^ ((4 - ((srow >> 2) & 3)) & 3);     /* source chunk of LDS slot tid&3 (swizzle) */  \
    const int n0 = fastdiv(m0, a.howo_magic, a.howo_shift), rem0 = m0 - n0 * HoWo;                                \
    const int oy0 = fastdiv(rem0, a.wo_magic, a.wo_shift), ox0 = rem0 - oy0 * a.Wo;                               \
    const long long lin0 = ((long long)n0 * a.H + oy0 * a.stride) * a.W + ox0 * a.stride;     /* uniform */       \
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + wave * 1024u);            \
    const int ld_off = lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);                                       \
    const float* const a_rd = lds + (wm * MF * 16) * 16 + ld_off;                                                 \
    const float* const b_rd = lds + BM * 16 + (wn * NF * 16) * 16 + ld_off;                                       \
    const bool b_last = BP > BFULL && (BFULL * RP + wave * 16 < BN);   /* this wave takes part in the partial B pass */ \
    f32x4 acc[MF][NF], part[MF][NF];                                                                              \
    _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                                \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

// B: weight rows of this workgroup's channel tile (lane offsets relative to the tile's first row)
#define PADEL_TAP_WEIGHTS()                                                                                       \
    unsigned voffB[BP];                                                                                           \
    _Pragma("unroll") for (int p = 0; p < BP; ++p) {                                                              \
        const int rr = srow + RP * p;                       /* row inside the BN tile */                          \
        const int frag = min(f0 + (rr >> 4), a.n16 - 1);    /* clamp: partial last channel tile */                \
        voffB[p] = (unsigned)((((frag - f0) * 16 + (rr & 15)) * Ktot + sc * 4) * 4);                              \
    }                                                                                                             \
    const i32x4 rsrcB = make_rsrc(a.w + (long long)f0 * 16 * Ktot);

// =====================================================================================================  3x3
// DBG (tuning only, PADEL_CONV_DIAG=16 on the 64x96 tile): every wave stamps s_memtime at 5 points of every k-step
// (step top / own requests landed / barrier passed / fragments in registers / last MFMA issued) into an LDS ring that is
// dumped to a.dbg at the end — read by tools/timeline_probe.py --kernel tap.
template <int WM, int WN, int MF, int NF, bool DBG = false>
__global__ void __launch_bounds__(64 * WM * WN, tap_min_waves(WM * WN, MF * NF)) conv_tap_kernel(const ConvArgs a) {
    PADEL_TAP_GEOMETRY(3)
    __shared__ unsigned long long stamps[DBG ? 4 * kConvDbgSteps * 5 + 4 * 64 : 1];
    unsigned long long t_begin = 0;
    int dbg_k = 0;                           // global k-step index of step 0 of the current block
    if constexpr (DBG) t_begin = __builtin_amdgcn_s_memtime();
#define PADEL_TAP_STAMP(J, slot)                                                                                  \
    do {                                                                                                          \
        if constexpr (DBG) {                                                                                      \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                           \
            const int real_ = ((wave & 3) * kConvDbgSteps + ((dbg_k + (J)) & (kConvDbgSteps - 1))) * 5 + (slot);  \
            const int dummy_ = 4 * kConvDbgSteps * 5 + (wave & 3) * 64 + lane;                                    \
            stamps[lane == 0 ? real_ : dummy_] = t_;                                                              \
        }                                                                                                         \
    } while (0)
    const int nfull = a.cin >> 5;            // 32-channel chunks: 18 k-steps each
    const bool has_tail = (a.cin & 16) != 0; // + 9 k-steps of the last 16 channels
    const int Ktot = (nfull * 18 + (has_tail ? 9 : 0)) * 16;

    // ---- A: per lane and tap one byte offset relative to the tap-(0,0) pixel of the workgroup's first row
    unsigned voffA[AP][9];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        int m = m0 + srow + RP * p;
        const bool rv = m < a.M;
        if (!rv) m = m0;
        const int n = fastdiv(m, a.howo_magic, a.howo_shift);
        const int rem = m - n * HoWo;
        const int oy = fastdiv(rem, a.wo_magic, a.wo_shift);
        const int ox = rem - oy * a.Wo;
        const long long lin = ((long long)n * a.H + oy * a.stride) * a.W + ox * a.stride;
        const unsigned off = (unsigned)(((lin - lin0) * a.in_cs + sc * 4) * 4);
        bool vy[3], vx[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            vy[d] = rv && (unsigned)(oy * a.stride - 1 + d) < (unsigned)a.H;
            vx[d] = (unsigned)(ox * a.stride - 1 + d) < (unsigned)a.W;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) voffA[p][t] = (vy[t / 3] && vx[t % 3]) ? off : kOutOfRange;
    }
    // base of the A descriptor: channel slice of the tap-(0,0) pixel of row m0 (may lie below a.in: never dereferenced there)
    const i32x4 rsrcA = make_rsrc(a.in + ((lin0 - (a.W + 1)) * a.in_cs + a.in_choff));
    unsigned tapoff[18];                      // SGPRs: byte offset of (tap, half) inside a chunk
#pragma unroll
    for (int j = 0; j < 18; ++j) {
        const int t = j >> 1, ky = t / 3, kx = t % 3;
        tapoff[j] = __builtin_amdgcn_readfirstlane((unsigned)(((ky * a.W + kx) * a.in_cs + (j & 1) * 16) * 4));
    }
    PADEL_TAP_WEIGHTS()

    unsigned s_chunk = 0;                     // byte offset of the current 32-channel chunk inside a pixel
    unsigned s_kb = 0;                        // byte offset of the current block inside a weight row (64 B per k-step)
    // step 1 of the NEXT block is (tap 0, half 1) in a full chunk but (tap 1, half 0) in the tail block: its SGPR and
    // lane offsets sit in registers that are re-selected once per chunk (step 0 is (tap 0, half 0) either way)
    bool nxt_tail = nfull == 0;
    unsigned wrap_off1 = nxt_tail ? tapoff[2] : tapoff[1];
    unsigned wrapv1[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) wrapv1[p] = nxt_tail ? voffA[p][1] : voffA[p][0];

    // chunk-relative step J of a full chunk: read stage J % 3, request step J + 2 (steps 16, 17 request the next block)
#define PADEL_TAP_STEP(J)                                                                                         \
    do {                                                                                                          \
        PADEL_TAP_STAMP(J, 0);                                                                                    \
        wait_vm<AP + BFULL>();                  /* own requests of step J landed (the younger ones may fly on) */ \
        PADEL_TAP_STAMP(J, 1);                                                                                    \
        __builtin_amdgcn_s_barrier();           /* ... everybody's did; stage (J + 2) % 3 is free again */        \
        PADEL_TAP_STAMP(J, 2);                                                                                    \
        if constexpr ((J) + 2 < 18)                                                                               \
            PADEL_TAP_DMA(((J) + 2) % 3, s_chunk + tapoff[((J) + 2) % 18], s_kb + ((J) + 2) * 64u,                \
                          voffA[0][(((J) + 2) % 18) >> 1], voffA[AP - 1][(((J) + 2) % 18) >> 1]);                 \
        else if constexpr ((J) == 16)                                                                             \
            PADEL_TAP_DMA(0, s_chunk + 128u + tapoff[0], s_kb + 18 * 64u, voffA[0][0], voffA[AP - 1][0]);         \
        else                                                                                                      \
            PADEL_TAP_DMA(1, s_chunk + 128u + wrap_off1, s_kb + 19 * 64u, wrapv1[0], wrapv1[AP - 1]);             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if constexpr (DBG) {                                                                                      \
            PADEL_TAP_FRAGS_DBG((J) % 3)                                                                          \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
            PADEL_TAP_STAMP(J, 3);                                                                                \
            PADEL_TAP_MFMAS_DBG();                                                                                \
            PADEL_TAP_STAMP(J, 4);                                                                                \
        } else {                                                                                                  \
            PADEL_TAP_COMPUTE((J) % 3);                                                                           \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
    // step J of the tail block (tap J, half 0): nothing is requested past its last step, so step 8 drains the queue
#define PADEL_TAP_TSTEP(J)                                                                                        \
    do {                                                                                                          \
        if constexpr ((J) == 8) wait_vm<0>(); else wait_vm<AP + BFULL>();                                         \
        __builtin_amdgcn_s_barrier();                                                                             \
        if constexpr ((J) + 2 < 9)                                                                                \
            PADEL_TAP_DMA(((J) + 2) % 3, s_chunk + tapoff[2 * ((J) + 2 < 9 ? (J) + 2 : 0)], s_kb + ((J) + 2) * 64u, \
                          voffA[0][(J) + 2 < 9 ? (J) + 2 : 0], voffA[AP - 1][(J) + 2 < 9 ? (J) + 2 : 0]);         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_TAP_COMPUTE((J) % 3);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

    PADEL_TAP_DMA(0, tapoff[0], 0u, voffA[0][0], voffA[AP - 1][0]);
    PADEL_TAP_DMA(1, wrap_off1, 64u, wrapv1[0], wrapv1[AP - 1]);
    for (int c = 0; c < nfull; ++c) {
        nxt_tail = has_tail && c == nfull - 1;
        wrap_off1 = nxt_tail ? tapoff[2] : tapoff[1];
#pragma unroll
        for (int p = 0; p < AP; ++p) wrapv1[p] = nxt_tail ? voffA[p][1] : voffA[p][0];
        PADEL_TAP_STEP(0);  PADEL_TAP_STEP(1);  PADEL_TAP_STEP(2);  PADEL_TAP_STEP(3);  PADEL_TAP_STEP(4);  PADEL_TAP_STEP(5);
        PADEL_TAP_STEP(6);  PADEL_TAP_STEP(7);  PADEL_TAP_STEP(8);  PADEL_TAP_STEP(9);  PADEL_TAP_STEP(10); PADEL_TAP_STEP(11);
        PADEL_TAP_STEP(12); PADEL_TAP_STEP(13); PADEL_TAP_STEP(14); PADEL_TAP_STEP(15); PADEL_TAP_STEP(16); PADEL_TAP_STEP(17);
        PADEL_TAP_FLUSH();
        s_chunk += 128u;
        s_kb += 18u * 64u;
        dbg_k += 18;
    }
    if (has_tail) {
        PADEL_TAP_TSTEP(0); PADEL_TAP_TSTEP(1); PADEL_TAP_TSTEP(2); PADEL_TAP_TSTEP(3); PADEL_TAP_TSTEP(4);
        PADEL_TAP_TSTEP(5); PADEL_TAP_TSTEP(6); PADEL_TAP_TSTEP(7); PADEL_TAP_TSTEP(8);
        PADEL_TAP_FLUSH();
    } else {
        wait_vm<0>();       // the two trailing requests must land before this workgroup's LDS is released
    }
    tap_epilogue<MF, NF>(a, acc, m0 + wm * MF * 16, f0 + wn * NF, lr, lq, m0 + BM <= a.M && (f0 + WN * NF) * 16 <= a.cout);
    if constexpr (DBG) {
        if (a.dbg) {
            const unsigned long long t_end = __builtin_amdgcn_s_memtime();
            __syncthreads();
            unsigned long long* d = a.dbg + (long long)(blockIdx.y * gridDim.x + blockIdx.x) * kConvDbgWords;
            for (int i = tid; i < 4 * kConvDbgSteps * 5; i += 64 * NW) d[8 + i] = stamps[i];
            if (lane == 0 && wave < 4) {
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
                const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
                d[wave] = ((unsigned long long)xcc << 32) | hw;
                if (wave == 0) { d[4] = t_begin; d[5] = t_end; d[6] = (unsigned long long)(nfull * 18 + (has_tail ? 9 : 0)); d[7] = (unsigned long long)bid; }
            }
        }
    }
#undef PADEL_TAP_STAMP
#undef PADEL_TAP_STEP
#undef PADEL_TAP_TSTEP
}

// =====================================================================================================  1x1
// the 4-stage ring caps residency at 160 KB / (4 stages): ask the register allocator for no more than that
constexpr int tap1_min_waves(int nw, int frags, int stage_bytes) {
    const int by_lds = (160 * 1024 / (4 * stage_bytes)) * nw / 4;
    const int want = tap_min_waves(nw, frags);
    return by_lds < want ? (by_lds < 1 ? 1 : by_lds) : want;
}

template <int WM, int WN, int MF, int NF, int PD>
__global__ void __launch_bounds__(64 * WM * WN, tap1_min_waves(WM * WN, MF * NF, (WM * MF + WN * NF) * 16 * 64)) conv_tap1_kernel(const ConvArgs a) {
    PADEL_TAP_GEOMETRY(4)
    const int nks = a.cin >> 4;              // 16 channels per k-step
    const int Ktot = nks * 16;

    unsigned voffA[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        int m = m0 + srow + RP * p;
        const bool rv = m < a.M;
        if (!rv) m = m0;
        const int n = fastdiv(m, a.howo_magic, a.howo_shift);
        const int rem = m - n * HoWo;
        const int oy = fastdiv(rem, a.wo_magic, a.wo_shift);
        const int ox = rem - oy * a.Wo;
        const long long lin = ((long long)n * a.H + oy * a.stride) * a.W + ox * a.stride;
        voffA[p] = rv ? (unsigned)(((lin - lin0) * a.in_cs + sc * 4) * 4) : kOutOfRange;
    }
    const i32x4 rsrcA = make_rsrc(a.in + (lin0 * a.in_cs + a.in_choff));
    PADEL_TAP_WEIGHTS()

    unsigned s_k = 0;                         // byte offset of the current block: 64 B per k-step in a pixel AND in a weight row
    // step J of a 16-step accumulation block: read stage J % 4, request step J + PD (past the end: unused slack
    // bytes).  PD = 3 uses the whole ring: the stage being refilled was read one step ago, and every wave finished
    // those reads before it arrived at this step's barrier.
    static_assert(PD == 2 || PD == 3, "prefetch distance");
#define PADEL_TAP1_STEP(J)                                                                                        \
    if ((J) < nb) {                                                                                               \
        wait_vm<(PD - 1) * (AP + BFULL)>();                                                                       \
        __builtin_amdgcn_s_barrier();                                                                             \
        PADEL_TAP_DMA(((J) + PD) % 4, s_k + ((J) + PD) * 64u, s_k + ((J) + PD) * 64u, voffA[0], voffA[AP - 1]);   \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_TAP_COMPUTE((J) % 4);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    }

    PADEL_TAP_DMA(0, 0u, 0u, voffA[0], voffA[AP - 1]);
    PADEL_TAP_DMA(1, 64u, 64u, voffA[0], voffA[AP - 1]);
    if constexpr (PD == 3) PADEL_TAP_DMA(2, 128u, 128u, voffA[0], voffA[AP - 1]);
    for (int k = 0; k < nks; k += 16) {
        const int nb = min(16, nks - k);
        PADEL_TAP1_STEP(0)  PADEL_TAP1_STEP(1)  PADEL_TAP1_STEP(2)  PADEL_TAP1_STEP(3)
        PADEL_TAP1_STEP(4)  PADEL_TAP1_STEP(5)  PADEL_TAP1_STEP(6)  PADEL_TAP1_STEP(7)
        PADEL_TAP1_STEP(8)  PADEL_TAP1_STEP(9)  PADEL_TAP1_STEP(10) PADEL_TAP1_STEP(11)
        PADEL_TAP1_STEP(12) PADEL_TAP1_STEP(13) PADEL_TAP1_STEP(14) PADEL_TAP1_STEP(15)
        PADEL_TAP_FLUSH();
        s_k += 16u * 64u;
    }
    wait_vm<0>();           // the two trailing requests must land before this workgroup's LDS is released
    tap_epilogue<MF, NF>(a, acc, m0 + wm * MF * 16, f0 + wn * NF, lr, lq, m0 + BM <= a.M && (f0 + WN * NF) * 16 <= a.cout);
#undef PADEL_TAP1_STEP
}

template <int WM, int WN, int MF, int NF>
static hipError_t launch_t(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    constexpr int BM = WM * MF * 16;
    a.n_mtiles = (a.M + BM - 1) / BM;
    dim3 grid(a.n_mtiles, (a.n16 + WN * NF - 1) / (WN * NF), 1);
    const int pd = a.tap_pd;
    if constexpr (WM == 2 && WN == 2 && MF == 2 && NF == 3) {
        if (a.ksize == 3 && a.dbg) {        // timeline instantiation (pa_engine_set_tuning "timeline")
            hipLaunchKernelGGL((conv_tap_kernel<2, 2, 2, 3, true>), grid, dim3(256), 0, s, a);
            return hipGetLastError();
        }
    }
    if (a.ksize == 3) hipLaunchKernelGGL((conv_tap_kernel<WM, WN, MF, NF>), grid, dim3(64 * WM * WN), 0, s, a);
    else if (pd == 3) hipLaunchKernelGGL((conv_tap1_kernel<WM, WN, MF, NF, 3>), grid, dim3(64 * WM * WN), 0, s, a);
    else hipLaunchKernelGGL((conv_tap1_kernel<WM, WN, MF, NF, 2>), grid, dim3(64 * WM * WN), 0, s, a);
    return hipGetLastError();
}

// tile ids of the fp32 id space (shared by the tap, bf16x3 (+0), h2 (+200) and fp16 families): waves WM x WN, fragments MF x NF per wave
struct TileShape { int id, wm, wn, mf, nf; };
static const TileShape tile_shapes[] = {
    {0, 2, 2, 4, 4},   // 128 x 128
    {1, 2, 2, 4, 3},   // 128 x  96
    {2, 4, 1, 4, 4},   // 256 x  64
    {3, 4, 1, 4, 3},   // 256 x  48
    {4, 4, 1, 4, 2},   // 256 x  32
    {5, 4, 1, 4, 1},   // 256 x  16
    {6, 2, 2, 2, 4},   //  64 x 128
    {7, 2, 2, 2, 3},   //  64 x  96
    {8, 4, 1, 2, 5},   // 128 x  80
    {9, 4, 1, 2, 4},   // 128 x  64
    {10, 2, 2, 4, 2},  // 128 x  64 (2x2 waves)
    {11, 4, 1, 2, 2},  // 128 x  32
    {12, 4, 1, 2, 1},  // 128 x  16
    // tap kernel only (conv_tap.hip): 8 waves per workgroup / 128 x 48
    {13, 4, 2, 2, 3},  // 128 x  96
    {14, 4, 2, 2, 4},  // 128 x 128
    {15, 4, 2, 2, 2},  // 128 x  64
    {20, 4, 1, 2, 3},  // 128 x  48
    {25, 4, 1, 1, 5},  //  64 x  80 (bf16x3 kernels only)
};

bool conv_variant_shape(int variant, int* bm, int* bn) {
    for (const auto& v : tile_shapes)
        if (v.id == variant) { *bm = v.wm * v.mf * 16; *bn = v.wn * v.nf * 16; return true; }
    return false;
}

// hipErrorNotSupported when the layer or the tile is not covered
hipError_t launch_conv_tap(const ConvArgs& a, int variant, hipStream_t s) {
    if ((a.ksize != 3 && a.ksize != 1) || (a.cin & 15) || a.cin < 16) return hipErrorNotSupported;
    switch (variant) {
        case 6: return launch_t<2, 2, 2, 4>(a, s);    //  64 x 128
        case 7: return launch_t<2, 2, 2, 3>(a, s);    //  64 x  96
        case 9: return launch_t<4, 1, 2, 4>(a, s);    // 128 x  64
        case 10: return launch_t<2, 2, 4, 2>(a, s);   // 128 x  64 (2x2 waves)
        case 11: return launch_t<4, 1, 2, 2>(a, s);   // 128 x  32
        case 12: return launch_t<4, 1, 2, 1>(a, s);   // 128 x  16
        case 13: return launch_t<4, 2, 2, 3>(a, s);   // 128 x  96,  8 waves
        case 14: return launch_t<4, 2, 2, 4>(a, s);   // 128 x 128,  8 waves
        case 15: return launch_t<4, 2, 2, 2>(a, s);   // 128 x  64,  8 waves
        case 20: return launch_t<4, 1, 2, 3>(a, s);   // 128 x  48
    }
    return hipErrorNotSupported;
}

// Tile choice for the tap kernels.  Measured on MI355X (profiles/conv_tap_sweep_r1.txt) every tile runs at
// 113-121 TFLOP/s when its shape fits the layer exactly (128x16: ~0.8 of that), so the choice is about padding waste
// (channel tile and pixel tile fill) and about the last, partly filled round of workgroups.
int choose_conv_tap_variant(int M, int n16) {
    struct V { int id, bm, nf; float speed; };
    static const V vs[] = {{7, 64, 6, 1.00f},  {13, 128, 6, 0.99f}, {9, 128, 4, 0.99f}, {14, 128, 8, 1.00f}, {6, 64, 8, 0.98f},
                           {20, 128, 3, 0.98f}, {11, 128, 2, 0.99f}, {12, 128, 1, 0.80f}};
    float best = -1.f;
    int bv = 7;
    for (const V& v : vs) {
        const int ntiles = (n16 + v.nf - 1) / v.nf;
        const long long mtiles = (M + v.bm - 1) / v.bm;
        const float fill = (float)n16 / (float)(ntiles * v.nf) * (float)M / (float)(mtiles * v.bm);
        const long long blocks = mtiles * ntiles;
        const long long per_cu = (blocks + 255) / 256;
        const float occ = (float)blocks / (256.f * (float)per_cu);
        const float sc = v.speed * fill * occ;
        if (sc > best) { best = sc; bv = v.id; }
    }
    return bv;
}

}  // namespace padel
