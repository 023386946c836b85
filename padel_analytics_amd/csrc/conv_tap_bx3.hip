// K3/K4 "bf16x3" — the fp32 convolutions at fp32 accuracy on the BF16 matrix pipe.
//
// CDNA4 has no TF32 and its fp32-input MFMA runs at 1/16 of the bf16 rate (MI355X_MICROARCH.md), which caps the fp32
// kernels of conv_tap.hip at 157 TFLOP/s.  An fp32 number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8
// significand bits: hi = x truncated to bf16, mid = (x - hi) truncated, lo = x - hi - mid), so
//     a * w = (ah + am + al)(wh + wm + wl) = ah*wh + (ah*wm + am*wh) + (ah*wl + am*wm + al*wh) + O(2^-24 |a w|),
// six bf16 products (each exact in fp32) accumulated in fp32: the three dropped terms are below half an fp32 ulp of
// the product, i.e. below what a single fp32 rounding of the fp32 kernels' FMA chain already costs.  Six
// v_mfma_f32_16x16x32_bf16 cover the K = 32 that takes eight v_mfma_f32_16x16x4_f32: 96 instead of 256 matrix-pipe
// cycles per 16x16x32 block.  tools/bf16x3_study.py (round 1) put the error of this scheme at or below the blocked
// fp32 chain's; tests/test_gpu_conv.py holds it to the same 3e-6 bound against fp64 and the parity suite to the same
// noise-floor criteria as the fp32 MFMA kernels before it may become the default.
//
// Nothing changes in HBM for the activations: they stay fp32 NHWC, and travel global -> LDS through the same
// buffer-addressed LDS-DMA ring (zeros for padded taps by the range check, wave-uniform SGPR offsets).  What changes:
//   * a k-step is 32 channels of one tap: two 64-byte sub-rows per pixel (A0 = channels [0,16), A1 = [16,32) of the
//     chunk), each staged exactly like a k-step of conv_tap.hip (same swizzle, same conflict-free fragment reads);
//   * the SPLIT happens in registers right after the fragment read: 2 AND + 2 SUB per value and 3 v_perm per pair —
//     ~90 VALU per k-step for a 2x3-fragment wave, issued next to 36 MFMAs;
//   * weights are pre-split on the host (graph.py:pack_conv_weight_bx3): per output channel and k-step three 64-byte
//     planes (hi | mid | lo, 32 bf16 each) in the lane order the A operand ends up with;
//   * accumulation keeps the two-level scheme (one partial set per 32-channel chunk x 9 taps, flushed into the main
//     accumulators), products are issued smallest first (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi).
// cin % 32 == 16 (yolov8m's 48-channel layers, n-scale's 16): the 3x3 kernel's tail block pairs TAPS instead of channel
// halves (A0 = the 16 channels at tap 2t, A1 = at tap 2t+1: 5 k-steps instead of 9 half-empty ones); the 1x1 kernel runs
// its last k-step with A1 switched off (out-of-range lane offsets -> zeros) against zero-padded weights.
//
// Since the second half of round 2 this file serves the stride-2 3x3 layers, the 1x1 layers and whatever stride-1 3x3 the
// patch kernel (conv_patch_bx3.hip: input split once per chunk instead of once per tap) does not take.  Both kernels
// exist with a 3-stage ring (prefetch distance 2, ids 6..25) and a 2-stage ring (prefetch distance 1, ids + 200: two
// thirds of the LDS -> 3 workgroups per CU, the default); the 1x1 kernel can absorb a preceding nn.Upsample(2)
// (template flag UP: the first up_c channels come from the coarse map).  -DPADEL_BX3_PROBES adds instantiations of
// tile 220 with parts of the k-step removed (wrong results; ceiling measurements, profiles/conv_bx3_probes_r2m.txt).
#include "bx3_common.h"

namespace padel {

// operands swapped like the other tap kernels' successors: A := weights, so D rows = channels, columns = pixels and a
// lane holds 4 consecutive channels of one pixel in the epilogue
// ring stage ST_ -> LDS read pointers / DMA base.  NSTG == 3: compile-time stage offsets (prefetch distance 2);
// NSTG == 2: the two stages alternate by step parity, the pointers are swapped after every odd-length block
// (prefetch distance 1: 2/3 of the LDS -> one more workgroup per CU)
#define PADEL_BX3_AR(ST_) (NSTG == 3 ? a_rd + (ST_) * STAGE : (((ST_) & 1) ? a_rd1 : a_rd0))
#define PADEL_BX3_BR(ST_) (NSTG == 3 ? b_rd + (ST_) * STAGE : (((ST_) & 1) ? b_rd1 : b_rd0))
#define PADEL_BX3_LW(SR_) (NSTG == 3 ? lds_wave : (((SR_) & 1) ? lw1 : lw0))
#define PADEL_BX3_IMM(SR_) (NSTG == 3 ? (SR_) * STAGE_B : 0)
#define PADEL_BX3_COMPUTE(ST_)                                                                                    \
    do {                                                                                                          \
        bf8 ah[MF], am[MF], al[MF];                                                                               \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                          \
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(PADEL_BX3_AR(ST_) + f * 256);                     \
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(PADEL_BX3_AR(ST_) + BM * 16 + f * 256);           \
            if constexpr (DBG & 1) {       /* ceiling probe: no split arithmetic (results are wrong) */          \
                ah[f] = __builtin_bit_cast(bf8, x0); am[f] = __builtin_bit_cast(bf8, x1); al[f] = ah[f];          \
            } else {                                                                                              \
                split8(x0, x1, ah[f], am[f], al[f]);                                                              \
            }                                                                                                     \
        }                                                                                                         \
        bf8 wh[NF], wm[NF], wl[NF];                                                                               \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) {                                                          \
            wh[j] = __builtin_bit_cast(bf8, *reinterpret_cast<const f32x4*>(PADEL_BX3_BR(ST_) + j * 256));     \
            wm[j] = __builtin_bit_cast(bf8, *reinterpret_cast<const f32x4*>(PADEL_BX3_BR(ST_) + BN * 16 + j * 256)); \
            wl[j] = __builtin_bit_cast(bf8, *reinterpret_cast<const f32x4*>(PADEL_BX3_BR(ST_) + 2 * BN * 16 + j * 256)); \
        }                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        if constexpr (DBG & 2) {           /* ceiling probe: full split, one product group instead of six */      \
            _Pragma("unroll") for (int f = 0; f < MF; ++f) { asm volatile("" :: "v"(am[f]), "v"(al[f])); }        \
            _Pragma("unroll") for (int j = 0; j < NF; ++j) { asm volatile("" :: "v"(wm[j]), "v"(wl[j])); }        \
        } else {                                                                                                  \
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
        }                                                                                                         \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            part[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], ah[f], part[f][j], 0, 0, 0);              \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)

#define PADEL_BX3_FLUSH()                                                                                         \
    do {                                                                                                          \
        _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                            \
            _Pragma("unroll") for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; } \
    } while (0)

// requests of one k-step into ring stage SR_: A0 / A1 sub-rows (lane offsets VA_ / VB_, SGPR offsets SA0_ / SA1_),
// three weight planes (SGPR offset SB_ + 64 * plane)
#define PADEL_BX3_DMA(SR_, SA0_, SA1_, SB_, VA0_, VA1_, VB0_, VB1_) PADEL_BX3_DMA_R(rsrcA, SR_, SA0_, SA1_, SB_, VA0_, VA1_, VB0_, VB1_)
#define PADEL_BX3_DMA_R(rsrcA, SR_, SA0_, SA1_, SB_, VA0_, VA1_, VB0_, VB1_)                                      \
    do {                                                                                                          \
        const unsigned sa0_ = (SA0_), sa1_ = (SA1_), sb_ = (SB_);                                                 \
        if constexpr (!(DBG & 4)) {        /* probe bit 4: no activation requests */                              \
        dma3<PADEL_BX3_IMM(SR_)>((VA0_), rsrcA, sa0_, PADEL_BX3_LW(SR_));                                                     \
        if constexpr (AP >= 2) dma3<PADEL_BX3_IMM(SR_) + RP * 64>((VA1_), rsrcA, sa0_, PADEL_BX3_LW(SR_));                    \
        dma3<PADEL_BX3_IMM(SR_) + BM * 64>((VB0_), rsrcA, sa1_, PADEL_BX3_LW(SR_));                                           \
        if constexpr (AP >= 2) dma3<PADEL_BX3_IMM(SR_) + BM * 64 + RP * 64>((VB1_), rsrcA, sa1_, PADEL_BX3_LW(SR_));          \
        }                                                                                                         \
        if constexpr (!(DBG & 8)) {        /* probe bit 8: no weight requests */                                  \
        PADEL_BX3_DMAB(SR_, 0, sb_);                                                                              \
        PADEL_BX3_DMAB(SR_, 1, sb_ + 64u);                                                                        \
        PADEL_BX3_DMAB(SR_, 2, sb_ + 128u);                                                                       \
        }                                                                                                         \
    } while (0)
#define PADEL_BX3_DMAB(SR_, PL_, SB_)                                                                             \
    do {                                                                                                          \
        if constexpr (BFULL >= 1) dma3<PADEL_BX3_IMM(SR_) + 2 * BM * 64 + (PL_) * BN * 64>(voffB[0], rsrcB, (SB_), PADEL_BX3_LW(SR_)); \
        if constexpr (BFULL >= 2) dma3<PADEL_BX3_IMM(SR_) + 2 * BM * 64 + (PL_) * BN * 64 + RP * 64>(voffB[1], rsrcB, (SB_), PADEL_BX3_LW(SR_)); \
        if constexpr (BP > BFULL) { if (b_last) dma3<PADEL_BX3_IMM(SR_) + 2 * BM * 64 + (PL_) * BN * 64 + BFULL * RP * 64>(voffB[BP - 1], rsrcB, (SB_), PADEL_BX3_LW(SR_)); } \
    } while (0)


#define PADEL_BX3_GEOMETRY()                                                                                      \
    constexpr int NW = WM * WN;                                                                                   \
    constexpr int RP = NW * 16;                                                                                   \
    constexpr int BM = WM * MF * 16, BN = WN * NF * 16;                                                           \
    constexpr int AP = BM / RP, BP = (BN + RP - 1) / RP, BFULL = BN / RP;                                         \
    constexpr int STAGE = (2 * BM + 3 * BN) * 16;      /* 4-byte words per ring stage: A0 | A1 | Whi | Wmid | Wlo */ \
    constexpr int STAGE_B = STAGE * 4;                                                                            \
    constexpr int NREQ = 2 * AP + 3 * BFULL;           /* requests every wave issues per k-step */                \
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");                                              \
    static_assert(BM % RP == 0 && AP <= 2 && BFULL <= 2, "A in 1-2 full passes, B in at most 2 full + 1 partial"); \
    static_assert(NSTG * STAGE_B <= 160 * 1024, "ring must fit the LDS");                                         \
    __shared__ __attribute__((aligned(16))) float lds[NSTG * STAGE];                                              \
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
    const int n0 = fastdiv3(m0, a.howo_magic, a.howo_shift), rem0 = m0 - n0 * HoWo;                               \
    const int oy0 = fastdiv3(rem0, a.wo_magic, a.wo_shift), ox0 = rem0 - oy0 * a.Wo;                              \
    const long long lin0 = ((long long)n0 * a.H + oy0 * a.stride) * a.W + ox0 * a.stride;                         \
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + wave * 1024u);            \
    const int ld_off = lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);                                       \
    const float* const a_rd = lds + (wm * MF * 16) * 16 + ld_off;                                                 \
    const float* const b_rd = lds + 2 * BM * 16 + (wn * NF * 16) * 16 + ld_off;                                   \
    const float *a_rd0 = a_rd, *a_rd1 = a_rd + STAGE, *b_rd0 = b_rd, *b_rd1 = b_rd + STAGE;   /* NSTG == 2 */      \
    unsigned lw0 = lds_wave, lw1 = __builtin_amdgcn_readfirstlane(lds_wave + (unsigned)STAGE_B);                  \
    (void)a_rd0; (void)a_rd1; (void)b_rd0; (void)b_rd1; (void)lw0; (void)lw1;                                     \
    const bool b_last = BP > BFULL && (BFULL * RP + wave * 16 < BN);                                              \
    const int nch = (a.cin + 31) >> 5;                 /* 32-channel chunks (the last one half empty if cin & 16) */ \
    const bool half_tail = (a.cin & 16) != 0;                                                                     \
    f32x4 acc[MF][NF], part[MF][NF];                                                                              \
    _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                                \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

// weight rows: nsteps * 192 bytes each (per k-step hi | mid | lo planes of 32 bf16)
#define PADEL_BX3_WEIGHTS(NSTEPS_)                                                                                \
    const unsigned rowb = (unsigned)(NSTEPS_) * 192u;                                                             \
    unsigned voffB[BP];                                                                                           \
    _Pragma("unroll") for (int p = 0; p < BP; ++p) {                                                              \
        const int rr = srow + RP * p;                                                                             \
        const int frag = min(f0 + (rr >> 4), a.n16 - 1);                                                          \
        voffB[p] = (unsigned)(((frag - f0) * 16 + (rr & 15)) * rowb + sc * 16);                                   \
    }                                                                                                             \
    const i32x4 rsrcB = make_rsrc3(reinterpret_cast<const char*>(a.w3) + (long long)f0 * 16 * rowb);

#define PADEL_BX3_FINISH()                                                                                        \
    const bool fast_ = m0 + BM <= a.M && (f0 + WN * NF) * 16 <= a.cout && (((a.out_choff | a.out_cs) & 3) == 0) && \
                       (!a.res || (((a.res_choff | a.res_cs) & 3) == 0));                                         \
    int mpix_[MF];                                                                                                \
    _Pragma("unroll") for (int f = 0; f < MF; ++f) { const int m_ = m0 + wm * MF * 16 + f * 16 + lr; mpix_[f] = m_ < a.M ? m_ : -1; } \
    bx3_epilogue<MF, NF>(a, acc, mpix_, f0 + wn * NF, lq, fast_);

// =====================================================================================================  3x3
template <int WM, int WN, int MF, int NF, int NSTG, int DBG = 0>
__global__ void __launch_bounds__(64 * WM * WN, min_waves3(MF * NF) + (NSTG == 2 ? 1 : 0)) conv_bx3_kernel(const ConvArgs a) {
    PADEL_BX3_GEOMETRY()
    unsigned voffA[AP][9];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        int m = m0 + srow + RP * p;
        const bool rv = m < a.M;
        if (!rv) m = m0;
        const int n = fastdiv3(m, a.howo_magic, a.howo_shift);
        const int rem = m - n * HoWo;
        const int oy = fastdiv3(rem, a.wo_magic, a.wo_shift);
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
        for (int t = 0; t < 9; ++t) voffA[p][t] = (vy[t / 3] && vx[t % 3]) ? off : kOOR3;
    }
    const i32x4 rsrcA = make_rsrc3(a.in + ((lin0 - (a.W + 1)) * a.in_cs + a.in_choff));
    unsigned tapoff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tapoff[t] = __builtin_amdgcn_readfirstlane((unsigned)((((t / 3) * a.W + (t % 3)) * a.in_cs) * 4));
    // K walk: nfull 32-channel chunks x 9 taps, then — for cin % 32 == 16 (yolov8m's 48-channel layers, n-scale's 16) —
    // a TAIL block that pairs TAPS instead of channel halves: tail step t covers the last 16 channels at taps 2t (as
    // sub-row A0) and 2t+1 (as A1), 5 steps instead of 9 half-empty ones
    const int nfull = a.cin >> 5;
    PADEL_BX3_WEIGHTS(nfull * 9 + (half_tail ? 5 : 0))

    unsigned s_chunk = 0, s_kb = 0;
#define PADEL_BX3_REQ_FULL(SR_, CH_, KB_, T_)                                                                      \
    PADEL_BX3_DMA(SR_, (CH_) + tapoff[T_], (CH_) + tapoff[T_] + 64u, KB_, voffA[0][T_], voffA[AP - 1][T_], voffA[0][T_], voffA[AP - 1][T_])
#define PADEL_BX3_REQ_TAIL(SR_, CH_, KB_, JT_)                                                                     \
    PADEL_BX3_DMA(SR_, (CH_) + tapoff[2 * (JT_)], (CH_) + tapoff[2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8], KB_,    \
                  voffA[0][2 * (JT_)], voffA[AP - 1][2 * (JT_)],                                                  \
                  2 * (JT_) + 1 < 9 ? voffA[0][2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8] : kOOR3,                    \
                  2 * (JT_) + 1 < 9 ? voffA[AP - 1][2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8] : kOOR3)
    bool nxt_tail = false;       // the block after the current full chunk is the tail block

#define PADEL_BX3_STEP(J)                                                                                         \
    do {                                                                                                          \
        wait_vm3<NREQ>();                                                                                         \
        __builtin_amdgcn_s_barrier();                                                                             \
        if constexpr ((J) + 2 < 9) {                                                                              \
            PADEL_BX3_REQ_FULL(((J) + 2) % 3, s_chunk, s_kb + ((J) + 2) * 192u, (J) + 2 < 9 ? (J) + 2 : 0);      \
        } else {                                                                                                  \
            if (nxt_tail) { PADEL_BX3_REQ_TAIL(((J) + 2) % 3, s_chunk + 128u, s_kb + ((J) + 2) * 192u, ((J) + 2) % 9); } \
            else { PADEL_BX3_REQ_FULL(((J) + 2) % 3, s_chunk + 128u, s_kb + ((J) + 2) * 192u, ((J) + 2) % 9); }   \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_BX3_COMPUTE((J) % 3);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
    // tail step JT (0..4): nothing is requested past step 4, so step 4 drains the queue
#define PADEL_BX3_TSTEP(JT)                                                                                       \
    do {                                                                                                          \
        if constexpr ((JT) == 4) wait_vm3<0>(); else wait_vm3<NREQ>();                                            \
        __builtin_amdgcn_s_barrier();                                                                             \
        if constexpr ((JT) + 2 < 5) { PADEL_BX3_REQ_TAIL(((JT) + 2) % 3, s_chunk, s_kb + ((JT) + 2) * 192u, (JT) + 2 < 5 ? (JT) + 2 : 0); } \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_BX3_COMPUTE((JT) % 3);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

    if constexpr (NSTG == 3) {
        if (nfull > 0) {
            PADEL_BX3_REQ_FULL(0, 0u, 0u, 0);
            PADEL_BX3_REQ_FULL(1, 0u, 192u, 1);
        } else {
            PADEL_BX3_REQ_TAIL(0, 0u, 0u, 0);
            PADEL_BX3_REQ_TAIL(1, 0u, 192u, 1);
        }
        for (int c = 0; c < nfull; ++c) {
            nxt_tail = half_tail && c == nfull - 1;
            PADEL_BX3_STEP(0); PADEL_BX3_STEP(1); PADEL_BX3_STEP(2); PADEL_BX3_STEP(3); PADEL_BX3_STEP(4);
            PADEL_BX3_STEP(5); PADEL_BX3_STEP(6); PADEL_BX3_STEP(7); PADEL_BX3_STEP(8);
            PADEL_BX3_FLUSH();
            s_chunk += 128u;
            s_kb += 9u * 192u;
        }
        if (half_tail) {
            PADEL_BX3_TSTEP(0); PADEL_BX3_TSTEP(1); PADEL_BX3_TSTEP(2); PADEL_BX3_TSTEP(3); PADEL_BX3_TSTEP(4);
            PADEL_BX3_FLUSH();
        } else {
            wait_vm3<0>();      // the two trailing requests (past the last chunk: slack bytes) must land before LDS is released
        }
    } else {
        // ---- 2-stage ring: step J waits for ITS requests (issued one step earlier), passes the barrier, requests
        // step J + 1 into the stage everybody just finished reading, computes.  Stage = parity of the step inside
        // the block; blocks have odd length (9 / 5), so the two stages swap roles after every block.
#define PADEL_BX3_SWAP()                                                                                          \
        do { const float* t_ = a_rd0; a_rd0 = a_rd1; a_rd1 = t_; t_ = b_rd0; b_rd0 = b_rd1; b_rd1 = t_;            \
             const unsigned u_ = lw0; lw0 = lw1; lw1 = u_; } while (0)
#define PADEL_BX3_STEP2(J)                                                                                        \
        do {                                                                                                      \
            wait_vm3<0>();                                                                                        \
            __builtin_amdgcn_s_barrier();                                                                         \
            if constexpr ((J) + 1 < 9) {                                                                          \
                PADEL_BX3_REQ_FULL((J) + 1, s_chunk, s_kb + ((J) + 1) * 192u, (J) + 1 < 9 ? (J) + 1 : 0);         \
            } else {                                                                                              \
                if (nxt_tail) { PADEL_BX3_REQ_TAIL((J) + 1, s_chunk + 128u, s_kb + ((J) + 1) * 192u, 0); }        \
                else if (c + 1 < nfull) { PADEL_BX3_REQ_FULL((J) + 1, s_chunk + 128u, s_kb + ((J) + 1) * 192u, 0); } \
            }                                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            PADEL_BX3_COMPUTE(J);                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        } while (0)
#define PADEL_BX3_TSTEP2(JT)                                                                                      \
        do {                                                                                                      \
            wait_vm3<0>();                                                                                        \
            __builtin_amdgcn_s_barrier();                                                                         \
            if constexpr ((JT) + 1 < 5) { PADEL_BX3_REQ_TAIL((JT) + 1, s_chunk, s_kb + ((JT) + 1) * 192u, (JT) + 1 < 5 ? (JT) + 1 : 0); } \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            PADEL_BX3_COMPUTE(JT);                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        } while (0)
        if (nfull > 0) { PADEL_BX3_REQ_FULL(0, 0u, 0u, 0); } else { PADEL_BX3_REQ_TAIL(0, 0u, 0u, 0); }
        for (int c = 0; c < nfull; ++c) {
            nxt_tail = half_tail && c == nfull - 1;
            PADEL_BX3_STEP2(0); PADEL_BX3_STEP2(1); PADEL_BX3_STEP2(2); PADEL_BX3_STEP2(3); PADEL_BX3_STEP2(4);
            PADEL_BX3_STEP2(5); PADEL_BX3_STEP2(6); PADEL_BX3_STEP2(7); PADEL_BX3_STEP2(8);
            PADEL_BX3_FLUSH();
            PADEL_BX3_SWAP();
            s_chunk += 128u;
            s_kb += 9u * 192u;
        }
        if (half_tail) {
            PADEL_BX3_TSTEP2(0); PADEL_BX3_TSTEP2(1); PADEL_BX3_TSTEP2(2); PADEL_BX3_TSTEP2(3); PADEL_BX3_TSTEP2(4);
            PADEL_BX3_FLUSH();
        }
        wait_vm3<0>();
#undef PADEL_BX3_STEP2
#undef PADEL_BX3_TSTEP2
#undef PADEL_BX3_SWAP
    }
    PADEL_BX3_FINISH()
#undef PADEL_BX3_STEP
#undef PADEL_BX3_TSTEP
#undef PADEL_BX3_REQ_FULL
#undef PADEL_BX3_REQ_TAIL
}

// =====================================================================================================  1x1
// UP: the first a.up_c channels (whole 32-channel chunks) are read from a.in2, a map of half the spatial size, at
// [y >> 1][x >> 1] — an nn.Upsample(2) + torch.cat in front of this conv that is never materialised
template <int WM, int WN, int MF, int NF, int NSTG, bool UP>
__global__ void __launch_bounds__(64 * WM * WN, min_waves3(MF * NF) + (NSTG == 2 ? 1 : 0)) conv_bx3_1_kernel(const ConvArgs a) {
    constexpr int DBG = 0;
    PADEL_BX3_GEOMETRY()
    unsigned voffA[AP];
    unsigned voffU[AP];
    const int H2 = a.H >> 1, W2 = a.W >> 1;
    const long long linU0 = ((long long)n0 * H2 + (oy0 >> 1)) * W2;      // first coarse pixel of the coarse row of m0: every pixel of the tile is at or after it
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        int m = m0 + srow + RP * p;
        const bool rv = m < a.M;
        if (!rv) m = m0;
        const int n = fastdiv3(m, a.howo_magic, a.howo_shift);
        const int rem = m - n * HoWo;
        const int oy = fastdiv3(rem, a.wo_magic, a.wo_shift);
        const int ox = rem - oy * a.Wo;
        const long long lin = ((long long)n * a.H + oy * a.stride) * a.W + ox * a.stride;
        voffA[p] = rv ? (unsigned)(((lin - lin0) * a.in_cs + sc * 4) * 4) : kOOR3;
        if constexpr (UP) {
            const long long linU = ((long long)n * H2 + (oy >> 1)) * W2 + (ox >> 1);
            voffU[p] = rv ? (unsigned)(((linU - linU0) * a.in2_cs + sc * 4) * 4) : kOOR3;
        }
    }
    (void)voffU; (void)linU0;
    const i32x4 rsrcA = make_rsrc3(a.in + (lin0 * a.in_cs + a.in_choff));
    const i32x4 rsrcU = make_rsrc3(UP ? a.in2 + (linU0 * a.in2_cs + a.in2_choff) : a.in);
    const unsigned nup = UP ? (unsigned)(a.up_c >> 5) : 0u;
    (void)rsrcU; (void)nup;
    PADEL_BX3_WEIGHTS(nch)
    // requests of chunk K_ into stage SR_: from the coarse map while K_ < nup
#define PADEL_BX3_1REQ(SR_, K_)                                                                                   \
    do {                                                                                                          \
        const unsigned k_ = (K_);                                                                                 \
        if (UP && k_ < nup) {                                                                                     \
            PADEL_BX3_DMA_R(rsrcU, SR_, k_ * 128u, k_ * 128u + 64u, k_ * 192u, voffU[0], voffU[AP - 1], voffU[0], voffU[AP - 1]); \
        } else {                                                                                                  \
            PADEL_BX3_DMA_R(rsrcA, SR_, k_ * 128u, k_ * 128u + 64u, k_ * 192u, voffA[0], voffA[AP - 1], PADEL_BX3_A1(k_, 0), PADEL_BX3_A1(k_, AP - 1)); \
        }                                                                                                         \
    } while (0)

    unsigned s_k = 0;                         // index of the first k-step of the current 9-step accumulation block
    // step J of a block: chunk s_k + J; its A1 exists unless it is the half-empty last chunk
#define PADEL_BX3_A1(K_, P_) ((half_tail && (int)(K_) >= nch - 1) ? kOOR3 : voffA[P_])
#define PADEL_BX3_1STEP(J)                                                                                        \
    if ((J) < nb) {                                                                                               \
        wait_vm3<NREQ>();                                                                                         \
        __builtin_amdgcn_s_barrier();                                                                             \
        PADEL_BX3_1REQ(((J) + 2) % 3, s_k + (J) + 2);                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_BX3_COMPUTE((J) % 3);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    }
    // 2-stage ring (see the 3x3 kernel): stage = parity of the step inside the 9-step block, pointers swapped per block
#define PADEL_BX3_1STEP2(J)                                                                                       \
    if ((J) < nb) {                                                                                               \
        wait_vm3<0>();                                                                                            \
        __builtin_amdgcn_s_barrier();                                                                             \
        PADEL_BX3_1REQ((J) + 1, s_k + (J) + 1);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_BX3_COMPUTE(J);                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    }
    PADEL_BX3_1REQ(0, 0u);
    if constexpr (NSTG == 3) {
        PADEL_BX3_1REQ(1, 1u);
        for (int k = 0; k < nch; k += 9) {
            const int nb = min(9, nch - k);
            PADEL_BX3_1STEP(0) PADEL_BX3_1STEP(1) PADEL_BX3_1STEP(2) PADEL_BX3_1STEP(3) PADEL_BX3_1STEP(4)
            PADEL_BX3_1STEP(5) PADEL_BX3_1STEP(6) PADEL_BX3_1STEP(7) PADEL_BX3_1STEP(8)
            PADEL_BX3_FLUSH();
            s_k += 9u;
        }
    } else {
        for (int k = 0; k < nch; k += 9) {
            const int nb = min(9, nch - k);
            PADEL_BX3_1STEP2(0) PADEL_BX3_1STEP2(1) PADEL_BX3_1STEP2(2) PADEL_BX3_1STEP2(3) PADEL_BX3_1STEP2(4)
            PADEL_BX3_1STEP2(5) PADEL_BX3_1STEP2(6) PADEL_BX3_1STEP2(7) PADEL_BX3_1STEP2(8)
            PADEL_BX3_FLUSH();
            { const float* t_ = a_rd0; a_rd0 = a_rd1; a_rd1 = t_; t_ = b_rd0; b_rd0 = b_rd1; b_rd1 = t_;
              const unsigned u_ = lw0; lw0 = lw1; lw1 = u_; }
            s_k += 9u;
        }
    }
    wait_vm3<0>();
    PADEL_BX3_FINISH()
#undef PADEL_BX3_1STEP
#undef PADEL_BX3_1STEP2
#undef PADEL_BX3_1REQ
#undef PADEL_BX3_A1
}

template <int WM, int WN, int MF, int NF, int NSTG = 3, int DBG = 0>
static hipError_t launch_b3(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    constexpr int BM = WM * MF * 16;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.n_ntiles = (a.n16 + WN * NF - 1) / (WN * NF);
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    if (a.ksize == 3) hipLaunchKernelGGL((conv_bx3_kernel<WM, WN, MF, NF, NSTG, DBG>), grid, dim3(64 * WM * WN), 0, s, a);
    else if (a.in2) return hipErrorNotSupported;                // absorbed upsample: launch_b3u tiles only
    else hipLaunchKernelGGL((conv_bx3_1_kernel<WM, WN, MF, NF, NSTG, false>), grid, dim3(64 * WM * WN), 0, s, a);
    return hipGetLastError();
}
// 1x1 with an absorbed upsample (a.in2): 2-stage ring tiles 213 / 220 / 209
template <int WM, int WN, int MF, int NF>
static hipError_t launch_b3u(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    constexpr int BM = WM * MF * 16;
    if (a.ksize != 1 || a.stride != 1 || (a.up_c & 31) || a.up_c <= 0 || a.up_c > a.cin || ((a.H | a.W) & 1)) return hipErrorNotSupported;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.n_ntiles = (a.n16 + WN * NF - 1) / (WN * NF);
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    hipLaunchKernelGGL((conv_bx3_1_kernel<WM, WN, MF, NF, 2, true>), grid, dim3(64 * WM * WN), 0, s, a);
    return hipGetLastError();
}

// tile ids of the fp32 id space (conv_variant_shape); the ring holds 2 BM + 3 BN rows per stage
hipError_t launch_conv_bx3(const ConvArgs& a, int variant, hipStream_t s) {
    if ((a.ksize != 3 && a.ksize != 1) || (a.cin & 15) || a.cin < 16 || !a.w3) return hipErrorNotSupported;
    if (a.in2 && a.ksize == 3) {                                // absorbed upsample in front of a 3x3: the patch kernel only
        if (variant < 300 || variant >= 400 || !conv_bx3p_supported(a)) return hipErrorNotSupported;
        return launch_conv_bx3p(a, variant - 300, s);
    }
    if (a.in2) {                                                // absorbed upsample in front of a 1x1: the three tiles instantiated for it
        if (a.ksize != 1) return hipErrorNotSupported;
        if (variant == 209 || variant == 9 || variant == 304) return launch_b3u<4, 1, 2, 4>(a, s);
        if (variant == 213 || variant == 13 || variant == 14 || variant == 306 || variant == 206 || variant == 6) return launch_b3u<4, 1, 2, 6>(a, s);
        return launch_b3u<4, 1, 2, 3>(a, s);
    }
    if (variant >= 300 && variant < 400) {                      // patch kernel, or its tap-kernel sibling where it does not apply
        const int nf = variant - 300;
        if (conv_bx3p_supported(a)) return launch_conv_bx3p(a, nf, s);
        variant = nf == 3 ? 220 : nf == 4 ? 209 : 206;
    }
    switch (variant) {
        case 7: return launch_b3<2, 2, 2, 3>(a, s);    //  64 x  96
        case 6: return launch_b3<2, 2, 2, 4>(a, s);    //  64 x 128
        case 9: return launch_b3<4, 1, 2, 4>(a, s);    // 128 x  64
        case 20: return launch_b3<4, 1, 2, 3>(a, s);   // 128 x  48
        case 11: return launch_b3<4, 1, 2, 2>(a, s);   // 128 x  32
        case 12: return launch_b3<4, 1, 2, 1>(a, s);   // 128 x  16
        case 13: return launch_b3<4, 2, 2, 3>(a, s);   // 128 x  96, 8 waves
        case 14: return launch_b3<4, 2, 2, 4>(a, s);   // 128 x 128, 8 waves
        // + 200: 2-stage ring (prefetch distance 1, 3 workgroups per CU)
        case 207: return launch_b3<2, 2, 2, 3, 2>(a, s);
        case 220: return launch_b3<4, 1, 2, 3, 2>(a, s);
        case 206: return launch_b3<2, 2, 2, 4, 2>(a, s);
        case 209: return launch_b3<4, 1, 2, 4, 2>(a, s);
        case 211: return launch_b3<4, 1, 2, 2, 2>(a, s);
        case 225: return launch_b3<4, 1, 1, 5, 2>(a, s);
        case 213: return launch_b3<4, 1, 2, 6, 2>(a, s);   // 128 x 96 with 4 waves (2 x 6 fragments each), 2 workgroups per CU
#ifdef PADEL_BX3_PROBES      // ceiling probes of tile 220 (WRONG results; tools/conv_bench.py only), DBG bits: 1 no split VALU, 2 one of the 6 MFMA groups, 4 / 8 no activation / weight requests
        case 420: return a.ksize == 3 ? launch_b3<4, 1, 2, 3, 2, 1>(a, s) : hipErrorNotSupported;
        case 520: return a.ksize == 3 ? launch_b3<4, 1, 2, 3, 2, 2>(a, s) : hipErrorNotSupported;
        case 620: return a.ksize == 3 ? launch_b3<4, 1, 2, 3, 2, 3>(a, s) : hipErrorNotSupported;    // no split, 1 group: data movement only
        case 720: return a.ksize == 3 ? launch_b3<4, 1, 2, 3, 2, 4>(a, s) : hipErrorNotSupported;    // no activation requests
        case 820: return a.ksize == 3 ? launch_b3<4, 1, 2, 3, 2, 8>(a, s) : hipErrorNotSupported;    // no weight requests
        case 920: return a.ksize == 3 ? launch_b3<4, 1, 2, 3, 2, 12>(a, s) : hipErrorNotSupported;   // no requests at all
        case 1020: return a.ksize == 3 ? launch_b3<4, 1, 2, 3, 2, 13>(a, s) : hipErrorNotSupported;  // LDS reads + 6 MFMA groups + barriers only
        case 1120: return a.ksize == 3 ? launch_b3<4, 1, 2, 3, 2, 5>(a, s) : hipErrorNotSupported;   // no split, no activation requests
#endif
        case 25: return launch_b3<4, 1, 1, 5>(a, s);   //  64 x  80, 4 waves of 16 x 80: the 19-fragment (304-channel) fused pose heads
    }
    return hipErrorNotSupported;
}

// Relative tile speeds measured on MI355X (profiles/conv_bx3_sweep_r2d.txt, ..._r2j.txt): 3x3 — 64x96 and 128x48 (4
// waves, 2 workgroups per CU) lead at 173-189 TFLOP/s on the yolov8m bottlenecks, the 8-wave tiles follow at ~0.9;
// 1x1 — since the channel tiles of a pixel tile run side by side on one XCD (the input is fetched from HBM once) the
// same two tiles lead there too (147-169 on the wide C2f cv2 layers).  The rest is padding waste and the fill of the
// last round of workgroups.
int choose_conv_bx3_variant(const ConvArgs& a) {
    const int M = a.M, n16 = a.n16, ksize = a.ksize;
    struct V { int id, bm, nf; float s3, s1; };
    // ids + 200 = the 2-stage ring: 51-53 KB of LDS instead of 77-80 -> 3 workgroups per CU, measured +5..12 % on every
    // yolov8m 3x3 layer shape and +4..20 % on the 1x1 ones (profiles/conv_bx3_sweep_r2k.txt, ..._r2l.txt)
    static const V vs[] = {{213, 128, 6, 1.07f, 1.08f},     // 128 x 96 with 4 waves of 2 x 6 fragments, 2 workgroups per CU (..._r2p.txt)
                           {220, 128, 3, 1.00f, 1.00f}, {207, 64, 6, 0.98f, 0.95f}, {209, 128, 4, 1.00f, 1.00f}, {206, 64, 8, 0.90f, 0.70f},
                           {211, 128, 2, 0.85f, 0.87f}, {225, 64, 5, 0.95f, 0.90f},
                           {7, 64, 6, 0.93f, 0.84f},    {20, 128, 3, 0.93f, 0.87f}, {13, 128, 6, 0.85f, 0.82f}, {14, 128, 8, 0.83f, 0.85f},
                           {25, 64, 5, 0.90f, 0.78f},
                           {11, 128, 2, 0.80f, 0.61f},  {9, 128, 4, 0.65f, 0.70f},  {6, 64, 8, 0.50f, 0.49f},  {12, 128, 1, 0.45f, 0.26f}};
    float best = -1.f;
    int bv = 7;
    for (const V& v : vs) {
        const int ntiles = (n16 + v.nf - 1) / v.nf;
        const long long mtiles = (M + v.bm - 1) / v.bm;
        const float fill = (float)n16 / (float)(ntiles * v.nf) * (float)M / (float)(mtiles * v.bm);
        const long long blocks = mtiles * ntiles;
        const long long per_cu = (blocks + 255) / 256;
        const float occ = (float)blocks / (256.f * (float)per_cu);
        const float sp = ksize == 3 ? v.s3 : v.s1;
        if (sp <= 0.0f) continue;
        const float sc = sp * fill * occ;
        if (sc > best) { best = sc; bv = v.id; }
    }
    // stride-1 3x3 layers with cin % 32 == 0: the patch kernel (conv_patch_bx3.hip) measured 1.18x (48-channel tiles,
    // 3 workgroups per CU) / 1.15x (64-channel tiles) the best tap tile on full 8 x 16 patches
    // (profiles/conv_bx3_sweep_r2n.txt); its fill counts the pixels of partial patches at the right / bottom edge
    if (conv_bx3p_supported(a)) {
        struct P { int nf; float sp; };
        static const P ps[] = {{3, 1.17f}, {4, 1.14f}};
        const long long patches = (long long)(M / (a.Ho * a.Wo)) * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
        for (const P& v : ps) {
            const int ntiles = (n16 + v.nf - 1) / v.nf;
            const float fill = (float)n16 / (float)(ntiles * v.nf) * (float)M / (float)(patches * 128);
            const long long blocks = patches * ntiles;
            const long long per_cu = (blocks + 255) / 256;
            const float sc = v.sp * fill * (float)blocks / (256.f * (float)per_cu);
            if (sc > best) { best = sc; bv = 300 + v.nf; }
        }
    }
    return bv;
}

}  // namespace padel
