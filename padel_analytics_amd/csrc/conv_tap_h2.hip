// K3/K4 "h2" — the fp32 convolutions on the F16 matrix pipe with THREE products per operand pair (h2_common.h).
//
// Activations arrive in HBM as fp16 pairs (x ~ h + m / 2048, 16-channel groups of 64 bytes [h | m]) written by their
// producer's epilogue, weights as pre-split planes [Npad][k-step][h | m][32].  This file is the implicit-GEMM TAP
// kernel family for them — stride-2 3x3, 1x1 (optionally absorbing a preceding nn.Upsample(2)) and whatever stride-1
// 3x3 the patch kernel (conv_patch_h2.hip) does not take.  It is the machine of conv_tap_bx3.hip (buffer-addressed
// LDS-DMA ring with prefetch distance 1, XOR-swizzled 64-byte rows, one raw barrier per k-step, XCD-aware tile map)
// with the in-register operand split REMOVED: sub-row A0 of a k-step is the h plane of its 32 channels, A1 the m plane,
// both fetched with the same lane offsets (the lane that fills 16-byte slot s of a row fetches K slots 8s..8s+7: group
// s >> 1, half s & 1 of the 128-byte chunk), and a k-step is 3 x MF x NF v_mfma_f32_16x16x32_f16:
//     cross += wh * am;   cross += wm * ah;   part += wh * ah;          part -> acc once per 9-step block (two-level)
// cin % 32 == 16: the 3x3 tail block pairs TAPS (K slots 0-15 = the 16 channels at tap 2t, 16-31 = at tap 2t+1: the
// lanes of slots 2, 3 of a row fetch from the second tap's pixel); the 1x1 runs its last k-step with the lanes of the
// absent group switched off (out-of-range offsets -> zeros).
#include "h2_tap.h"

namespace padel {

// =====================================================================================================  3x3
template <int WM, int WN, int MF, int NF, bool WS = false>
__global__ void __launch_bounds__(64 * WM * WN, MF * NF <= 6 ? 3 : 2) conv_h2_kernel(const ConvArgs a) {
    PADEL_H2T_GEOMETRY()
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
        const unsigned off = (unsigned)((lin - lin0) * a.in_cs * 4) + slot_b;
        bool vy[3], vx[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            vy[d] = rv && (unsigned)(oy * a.stride - 1 + d) < (unsigned)a.H;
            vx[d] = (unsigned)(ox * a.stride - 1 + d) < (unsigned)a.W;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) voffA[p][t] = (vy[h2_tap_ky(t)] && vx[h2_tap_kx(t)]) ? off : kOORh;
    }
    const i32x4 rsrcA = make_rsrc3(a.in + ((lin0 - (a.W + 1)) * a.in_cs + a.in_choff));
    unsigned tapoff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tapoff[t] = __builtin_amdgcn_readfirstlane((unsigned)(((h2_tap_ky(t) * a.W + h2_tap_kx(t)) * a.in_cs) * 4));
    // K walk: nfull 32-channel chunks x 9 taps, then — for cin % 32 == 16 — a TAIL block of 5 steps that pair taps
    const int nfull = a.cin >> 5;
    PADEL_H2T_WEIGHTS(nfull * 9 + (half_tail ? 5 : 0))

    unsigned s_chunk = 0, s_kb = 0;
#define PADEL_H2T_REQ_FULL(SR_, CH_, KB_, T_)                                                                      \
    PADEL_H2T_DMA_R(rsrcA, SR_, (CH_) + tapoff[T_], KB_, voffA[0][T_], voffA[AP - 1][T_])
    // tail step JT_: lanes of slots 0, 1 fetch the group's parts at tap 2 JT_, lanes of slots 2, 3 at tap 2 JT_ + 1 (their
    // offsets carry + 64 for "second group of a chunk": taken back out; the tap distance goes in instead)
    // (column-major taps: the second tap of a pair may lie BEFORE the first in memory — the scalar offset carries the smaller
    //  of the two tap offsets, each lane group adds its own distance: lane offsets stay non-negative)
#define PADEL_H2T_TB(JT_) (2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8)
#define PADEL_H2T_TMIN(JT_) min(tapoff[2 * (JT_)], tapoff[PADEL_H2T_TB(JT_)])
#define PADEL_H2T_TV(P_, JT_)                                                                                      \
    (sc_hi ? (2 * (JT_) + 1 < 9 ? voffA[P_][PADEL_H2T_TB(JT_)] + (tapoff[PADEL_H2T_TB(JT_)] - PADEL_H2T_TMIN(JT_)) - 64u : kOORh) \
           : voffA[P_][2 * (JT_)] + (tapoff[2 * (JT_)] - PADEL_H2T_TMIN(JT_)))
#define PADEL_H2T_REQ_TAIL(SR_, CH_, KB_, JT_)                                                                     \
    do {                                                                                                          \
        const unsigned v0_ = PADEL_H2T_TV(0, JT_), v1_ = PADEL_H2T_TV(AP - 1, JT_);                                \
        PADEL_H2T_DMA_R(rsrcA, SR_, (CH_) + PADEL_H2T_TMIN(JT_), KB_, v0_, v1_);                                   \
    } while (0)
    bool nxt_tail = false;       // the block after the current full chunk is the tail block

    // step J waits for ITS requests (issued one step earlier), passes the barrier, requests step J + 1 into the stage
    // everybody just finished reading, computes.  Stage = parity of the step inside the block; blocks have odd length
    // (9 / 5), so the two stages swap roles after every block.
#define PADEL_H2T_STEP2(J)                                                                                        \
    do {                                                                                                          \
        wait_vm3<0>();                                                                                            \
        __builtin_amdgcn_s_barrier();                                                                             \
        if constexpr ((J) + 1 < 9) {                                                                              \
            PADEL_H2T_REQ_FULL((J) + 1, s_chunk, s_kb + ((J) + 1) * 128u, (J) + 1 < 9 ? (J) + 1 : 0);             \
        } else {                                                                                                  \
            if (nxt_tail) { PADEL_H2T_REQ_TAIL((J) + 1, s_chunk + 128u, s_kb + ((J) + 1) * 128u, 0); }            \
            else if (c + 1 < nfull) { PADEL_H2T_REQ_FULL((J) + 1, s_chunk + 128u, s_kb + ((J) + 1) * 128u, 0); }  \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_H2T_COMPUTE(J, (J) == 0);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#define PADEL_H2T_TSTEP2(JT)                                                                                      \
    do {                                                                                                          \
        wait_vm3<0>();                                                                                            \
        __builtin_amdgcn_s_barrier();                                                                             \
        if constexpr ((JT) + 1 < 5) { PADEL_H2T_REQ_TAIL((JT) + 1, s_chunk, s_kb + ((JT) + 1) * 128u, (JT) + 1 < 5 ? (JT) + 1 : 0); } \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_H2T_COMPUTE(JT, (JT) == 0);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
    if (nfull > 0) { PADEL_H2T_REQ_FULL(0, 0u, 0u, 0); } else { PADEL_H2T_REQ_TAIL(0, 0u, 0u, 0); }
    for (int c = 0; c < nfull; ++c) {
        nxt_tail = half_tail && c == nfull - 1;
        PADEL_H2T_STEP2(0); PADEL_H2T_STEP2(1); PADEL_H2T_STEP2(2); PADEL_H2T_STEP2(3); PADEL_H2T_STEP2(4);
        PADEL_H2T_STEP2(5); PADEL_H2T_STEP2(6); PADEL_H2T_STEP2(7); PADEL_H2T_STEP2(8);
        PADEL_H2T_FLUSH();
        PADEL_H2T_SWAP();
        s_chunk += 128u;
        s_kb += 9u * 128u;
    }
    if (half_tail) {
        PADEL_H2T_TSTEP2(0); PADEL_H2T_TSTEP2(1); PADEL_H2T_TSTEP2(2); PADEL_H2T_TSTEP2(3); PADEL_H2T_TSTEP2(4);
        PADEL_H2T_FLUSH();
    }
    wait_vm3<0>();
    PADEL_H2T_FINISH()
#undef PADEL_H2T_STEP2
#undef PADEL_H2T_TSTEP2
#undef PADEL_H2T_REQ_FULL
#undef PADEL_H2T_REQ_TAIL
#undef PADEL_H2T_TV
#undef PADEL_H2T_TMIN
#undef PADEL_H2T_TB
}

// =====================================================================================================  1x1
// UP: the first a.up_c channels (whole 32-channel chunks) are read from a.in2, a map of half the spatial size, at
// [y >> 1][x >> 1] — an nn.Upsample(2) + torch.cat in front of this conv that is never materialised
template <int WM, int WN, int MF, int NF, bool UP, bool WS = false>
__global__ void __launch_bounds__(64 * WM * WN, MF * NF <= 6 ? 3 : 2) conv_h2_1_kernel(const ConvArgs a) {
    PADEL_H2T_GEOMETRY()
    unsigned voffA[AP], voffT[AP], voffU[AP];
    const int H2 = a.H >> 1, W2 = a.W >> 1;
    const long long linU0 = ((long long)n0 * H2 + (oy0 >> 1)) * W2;      // first coarse pixel of the coarse row of m0
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
        voffA[p] = rv ? (unsigned)((lin - lin0) * a.in_cs * 4) + slot_b : kOORh;
        voffT[p] = sc_hi ? kOORh : voffA[p];               // last chunk of a cin % 32 == 16 layer: its second group does not exist
        if constexpr (UP) {
            const long long linU = ((long long)n * H2 + (oy >> 1)) * W2 + (ox >> 1);
            voffU[p] = rv ? (unsigned)((linU - linU0) * a.in2_cs * 4) + slot_b : kOORh;
        }
    }
    (void)voffU; (void)linU0;
    const i32x4 rsrcA = make_rsrc3(a.in + (lin0 * a.in_cs + a.in_choff));
    const i32x4 rsrcU = make_rsrc3(UP ? a.in2 + (linU0 * a.in2_cs + a.in2_choff) : a.in);
    const unsigned nup = UP ? (unsigned)(a.up_c >> 5) : 0u;
    (void)rsrcU; (void)nup;
    PADEL_H2T_WEIGHTS(nch)
    // requests of chunk K_ into stage SR_: from the coarse map while K_ < nup
#define PADEL_H2T_1REQ(SR_, K_)                                                                                   \
    do {                                                                                                          \
        const unsigned k_ = (K_);                                                                                 \
        if (UP && k_ < nup) {                                                                                     \
            PADEL_H2T_DMA_R(rsrcU, SR_, k_ * 128u, k_ * 128u, voffU[0], voffU[AP - 1]);                            \
        } else if (half_tail && (int)k_ >= nch - 1) {                                                             \
            PADEL_H2T_DMA_R(rsrcA, SR_, k_ * 128u, k_ * 128u, voffT[0], voffT[AP - 1]);                            \
        } else {                                                                                                  \
            PADEL_H2T_DMA_R(rsrcA, SR_, k_ * 128u, k_ * 128u, voffA[0], voffA[AP - 1]);                            \
        }                                                                                                         \
    } while (0)

    unsigned s_k = 0;                         // index of the first k-step of the current 9-step accumulation block
#define PADEL_H2T_1STEP2(J)                                                                                       \
    if ((J) < nb) {                                                                                               \
        wait_vm3<0>();                                                                                            \
        __builtin_amdgcn_s_barrier();                                                                             \
        if ((int)(s_k + (J) + 1) < nch) PADEL_H2T_1REQ((J) + 1, s_k + (J) + 1);                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_H2T_COMPUTE(J, (J) == 0);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    }
    PADEL_H2T_1REQ(0, 0u);
    for (int k = 0; k < nch; k += 9) {
        const int nb = min(9, nch - k);
        PADEL_H2T_1STEP2(0) PADEL_H2T_1STEP2(1) PADEL_H2T_1STEP2(2) PADEL_H2T_1STEP2(3) PADEL_H2T_1STEP2(4)
        PADEL_H2T_1STEP2(5) PADEL_H2T_1STEP2(6) PADEL_H2T_1STEP2(7) PADEL_H2T_1STEP2(8)
        PADEL_H2T_FLUSH();
        PADEL_H2T_SWAP();
        s_k += 9u;
    }
    wait_vm3<0>();
    PADEL_H2T_FINISH()
#undef PADEL_H2T_1STEP2
#undef PADEL_H2T_1REQ
}

template <int WM, int WN, int MF, int NF>
static hipError_t launch_h2t(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    constexpr int BM = WM * MF * 16;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.n_ntiles = (a.n16 + WN * NF - 1) / (WN * NF);
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    if (a.ksize == 3) {
        if (a.in2) return hipErrorNotSupported;
        if (a.w_single) hipLaunchKernelGGL((conv_h2_kernel<WM, WN, MF, NF, true>), grid, dim3(64 * WM * WN), 0, s, a);
        else hipLaunchKernelGGL((conv_h2_kernel<WM, WN, MF, NF>), grid, dim3(64 * WM * WN), 0, s, a);
    } else if (a.in2) {
        if (a.stride != 1 || (a.up_c & 31) || a.up_c <= 0 || a.up_c > a.cin || ((a.H | a.W) & 1)) return hipErrorNotSupported;
        if (a.w_single) hipLaunchKernelGGL((conv_h2_1_kernel<WM, WN, MF, NF, true, true>), grid, dim3(64 * WM * WN), 0, s, a);
        else hipLaunchKernelGGL((conv_h2_1_kernel<WM, WN, MF, NF, true>), grid, dim3(64 * WM * WN), 0, s, a);
    } else {
        if (a.w_single) hipLaunchKernelGGL((conv_h2_1_kernel<WM, WN, MF, NF, false, true>), grid, dim3(64 * WM * WN), 0, s, a);
        else hipLaunchKernelGGL((conv_h2_1_kernel<WM, WN, MF, NF, false>), grid, dim3(64 * WM * WN), 0, s, a);
    }
    return hipGetLastError();
}

// tile ids follow the bf16x3 ids (conv_variant_shape + 200); 30x = the patch kernel (conv_patch_h2.hip) or, where it does
// not apply, its tap sibling
hipError_t launch_conv_h2(const ConvArgs& a, int variant, hipStream_t s) {
    if ((a.ksize != 3 && a.ksize != 1) || (a.cin & 15) || a.cin < 16 || !a.w || !a.oscale || !a.ovf_flag) return hipErrorNotSupported;
    if (variant >= 341 && variant <= 343) {  // the wide patch kernel for 16 / 32 / 48 input channels (conv_patch_h2w.hip), 1..3 channel fragments
        if (!(a.tune & 8) && conv_h2v_supported(a)) {      // its register-weights form (conv_patch_h2v.hip; tuning bit 3: the round-3 kernel)
            const hipError_t e = launch_conv_h2v(a, variant - 340, s);
            if (e != hipErrorNotSupported) return e;
        }
        if (conv_h2w_supported(a)) return launch_conv_h2w(a, variant - 340, s);
        variant = 303;
    }
    if (variant == 324 || variant == 325) {   // the register-weights quad kernels (conv_patch_h2r.hip): 324 = 96-channel tiles (two-product layers),
        if (conv_h2r_supported(a)) {           // 325 = 64-channel tiles (two or three products); elsewhere the quad kernel / the 64-channel patch tile
            const hipError_t e = launch_conv_h2r(a, variant == 324 ? 3 : 2, s);
            if (e != hipErrorNotSupported) return e;
        }
        variant = variant == 324 ? 323 : 304;
    }
    if (variant == 323) {                  // the quad patch kernel (conv_patch_h2q.hip); where it does not apply, the 48-channel patch tile
        if (conv_h2q_supported(a)) return launch_conv_h2q(a, s);
        variant = 303;
    }
    if (variant >= 300 && variant < 400) {
        const int nf = variant - 300;
        if (conv_h2p_supported(a)) return launch_conv_h2p(a, nf, s);
        if (a.in2 && a.ksize == 3) return hipErrorNotSupported;      // an absorbed upsample in front of a 3x3: the patch kernel only
        variant = (nf == 3 || nf == 13) ? 220 : (nf == 4 || nf == 14) ? 209 : 213;      // where the patch kernel does not apply: its tap sibling
    }
    if (a.in2 && a.ksize == 3) return hipErrorNotSupported;
    if (variant == 248) {                  // ... as 64 x 192 tiles of four waves
        if (conv_h2s3_supported(a)) return launch_conv_h2s3(a, s, true);
        variant = 213;
    }
    if (variant == 246) {                  // stride-2 3x3 on the register-weights ring machine (conv_1x1_h2s.hip: 128 x 192); elsewhere the 128 x 96 tap tile
        if (conv_h2s3_supported(a)) return launch_conv_h2s3(a, s);
        variant = 213;
    }
    if (variant == 247) {                  // 64 x 192 tiles of the 1x1 register-weights kernel (four waves, two workgroups per CU)
        if (conv_h2s_supported(a)) return launch_conv_h2s(a, true, s, true);
        variant = 243;
    }
    if (variant == 244 || variant == 245) {   // 1x1 with register weights and a deep activation ring (conv_1x1_h2s.hip): 128 x 96 / 128 x 192; elsewhere the deep-ring tile
        if (conv_h2s_supported(a)) return launch_conv_h2s(a, variant == 245, s);
        variant = 243;
    }
    if (variant == 243 || variant == 239) {   // 1x1 with the three-stage activation ring (conv_h2_1p_kernel); other kernel sizes: the plain tile
        const hipError_t e = launch_conv_h2_deep(a, variant, s);
        if (e != hipErrorNotSupported) return e;
        variant -= 30;
    }
    switch (variant) {
        case 207: return launch_h2t<2, 2, 2, 3>(a, s);    //  64 x  96
        case 220: return launch_h2t<4, 1, 2, 3>(a, s);    // 128 x  48
        case 209: return launch_h2t<4, 1, 2, 4>(a, s);    // 128 x  64
        case 211: return launch_h2t<4, 1, 2, 2>(a, s);    // 128 x  32
        case 213: return launch_h2t<4, 1, 2, 6>(a, s);    // 128 x  96, 4 waves of 2 x 6 fragments
        case 225: return launch_h2t<4, 1, 1, 5>(a, s);    //  64 x  80: the 19-fragment (304-channel) fused pose heads
        // (the eight-wave tiles 230 = 128 x 192 and 231 = 256 x 96 of round 4 were timed at the start of round 5 and removed:
        //  8-25 % slower than 128 x 96 on the K >= 576 1x1 and the stride-2 3x3 layers, 20-70 % on the P2 layers —
        //  profiles/r5a_tiles_230_231.txt; eight waves in lock step behind one barrier leave one workgroup per CU)
    }
    return hipErrorNotSupported;
}

// Per-layer tile choice.  Relative speeds start from the bf16x3 measurements (profiles/conv_bx3_sweep_r2*.txt) and are
// re-measured for h2 in profiles/conv_h2_sweep_r3*.txt; the rest is padding waste and the fill of the last round.
static const float kH2WSpeed[3] = {1.28f, 1.42f, 1.46f};  // relative speed of the wide patch kernel with 1 / 2 / 3 fragments: 16 -> 16: 121 vs 42 TFLOP/s,
                                                           // 32 -> 32: 185 vs 148, 48 -> 48: 244 vs 211 for the best 8 x 16 / tap tile (profiles/r3_sweep_h2w.txt)

int choose_conv_h2_variant(const ConvArgs& a) {
    const int M = a.M, n16 = a.n16, ksize = a.ksize;
    struct V { int id, bm, nf; float s3, s1; };
    static const V vs[] = {{213, 128, 6, 1.07f, 1.08f}, {220, 128, 3, 1.00f, 1.00f}, {207, 64, 6, 0.98f, 0.95f}, {209, 128, 4, 1.00f, 1.00f},
                           {211, 128, 2, 0.85f, 0.87f}, {225, 64, 5, 0.95f, 0.90f}};
    float best = -1.f;
    int bv = 220;
    for (const V& v : vs) {
        const int ntiles = (n16 + v.nf - 1) / v.nf;
        const long long mtiles = (M + v.bm - 1) / v.bm;
        const float fill = (float)n16 / (float)(ntiles * v.nf) * (float)M / (float)(mtiles * v.bm);
        const long long blocks = mtiles * ntiles;
        const long long per_cu = (blocks + 255) / 256;
        const float occ = (float)blocks / (256.f * (float)per_cu);
        const float sc = (ksize == 3 ? v.s3 : v.s1) * fill * occ;
        if (sc > best) { best = sc; bv = v.id; }
    }
    // 1x1 layers with K >= 192 stream their activations from HBM: the three-stage activation ring (conv_tap_h2p.hip) measured
    // +2..3 % at K = 192, +7..8 % at K = 576 / 1152, -2.5 % at K = 96 (profiles/r5b_tiles_1x1_deep_ring.txt); same results
    if (ksize == 1 && a.cin >= 192 && (bv == 213 || bv == 209)) bv += 30;
    // Round 6: what bounds a long-K 1x1 layer is the LDS-DMA stream of its activation tile, requested again by every 96-channel tile
    // of a pixel tile (conv_1x1_h2s.hip: the kernel is as fast with its MFMAs compiled out).  128 x 192 tiles (8 waves, one workgroup
    // per CU, register weights) halve the requests: +10..14 % on 768 / 960 / 1152 -> 384 / 576, +4..7 % on 384 / 576 -> 384, level or
    // behind on 192-channel outputs (profiles/r6D_1x1_tile_245.txt); bitwise the same results
    if (ksize == 1 && a.w_single && n16 >= 24 && a.cin >= 384 && conv_h2s_supported(a)) bv = 245;
    // ... and the same bytes as 64 x 192 tiles of FOUR waves, two workgroups per CU (two barrier domains instead of eight waves in
    // lock step): +8..10 % over the better of the two on 192 / 384 / 576 -> 192 and 384 / 576 -> 384, level at 768 -> 384, 1152 -> 576,
    // -2 % at K = 1152 -> 384 (profiles/r6L_1x1_tile_247.txt); taken for K < 960 where 192-channel tiles fit
    if (ksize == 1 && a.w_single && n16 >= 12 && (float)(((n16 + 11) / 12) * 12) <= 1.1f * (float)n16 && a.cin >= 192 && a.cin < 960 &&
        conv_h2s_supported(a)) bv = 247;
    // The stride-2 3x3 layers request a 16 KB tile per TAP and 96-channel tile (7.4 TB/s of requests on 96 -> 192): the same tile
    // gives +10..12 % on 96 -> 192, +23..31 % on 192 -> 192 / 384 / 576 (profiles/r6F_s2_tile_246.txt); taken where 192-channel tiles
    // waste at most a fifth of their columns
    if (ksize == 3 && a.stride == 2 && n16 >= 10 && (float)(((n16 + 11) / 12) * 12) <= 1.2f * (float)n16 && conv_h2s3_supported(a)) bv = 246;
    // (as 64 x 192 four-wave tiles, two workgroups per CU: +3..5 % at 96 and 384 input channels, -1..3 % at 192 — profiles/r6N_s2_tile_248.txt)
    if (bv == 246 && a.cin != 192) bv = 248;
    if (conv_h2p_supported(a)) {
        struct P { int nf; float sp; };
        // (the 6-fragment patch tile accumulates its main product in ONE level — registers — and measured no faster than the
        //  3-fragment one, profiles/conv_h2_sweep_r3a.txt: never chosen automatically, so results do not depend on the tile)
        static const P ps[] = {{3, 1.17f}, {4, 1.10f}};
        const long long patches = (long long)(M / (a.Ho * a.Wo)) * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
        for (const P& v : ps) {
            const int ntiles = (n16 + v.nf - 1) / v.nf;
            const float fill = (float)n16 / (float)(ntiles * v.nf) * (float)M / (float)(patches * 128);
            const long long blocks = patches * ntiles;
            const long long per_cu = (blocks + 255) / 256;
            const float sc = v.sp * fill * (float)blocks / (256.f * (float)per_cu);
            if (sc > best) { best = sc; bv = 300 + v.nf; }
        }
        // the quad patch kernel (conv_patch_h2q.hip: 8 x 16 pixels x 96 channels per workgroup, 2 workgroups per CU) measured
        // 1.02-1.05 x the 48-channel tile where the channels fill its tiles (profiles/r3k_sweep_h2q.txt); bitwise the same results
        if (conv_h2q_supported(a) && n16 % 6 == 0) {
            const int ntiles = n16 / 6;
            const float fill = (float)M / (float)(patches * 128);
            const long long blocks = patches * ntiles;
            const long long per_cu = (blocks + 255) / 256;
            const float sc = 1.21f * fill * (float)blocks / (256.f * (float)per_cu);
            if (sc > best) { best = sc; bv = 323; }
        }
        // two-product layers (PA_CONV_W_SINGLE) with at least two 32-channel chunks: the register-weights form of the quad tile
        // (conv_patch_h2r.hip, round 6: weights global -> VGPR, one barrier per chunk, 2 persistent workgroups per CU) measured
        // 1.15-1.18 x the quad kernel (96 -> 96: 440 -> 505, 192 -> 192: 520 -> 615 TFLOP/s; profiles/r6_sweep_h2r.txt) — fast enough to
        // take channel counts that leave its last 96-channel tile part empty (192 -> 256: 553 vs 473 on the 64-channel tile)
        if (conv_h2r_supported(a) && a.w_single) {
            const int ntiles = (n16 + 5) / 6;
            const float fill = (float)n16 / (float)(ntiles * 6) * (float)M / (float)(patches * 128);
            // (no round-quantisation term: the persistent workgroups start their next tile's loads under the current tile, and
            //  a 2.25-tiles-per-workgroup launch — the players graph's 192 -> 192 at 24 x 40 — still measured 461 vs 377 TFLOP/s)
            const float sc = 1.40f * fill;
            if (sc > best) { best = sc; bv = 324; }
        }
        // the same kernel on 64-channel tiles (tile 325: a wave 4 rows x 2 fragments): two-product layers whose channels are not a
        // multiple of 96 (192 -> 256: 600 vs 544 on tile 324 vs 472 on the 64-channel patch tile; 192 -> 64: 569 vs 451; 64 -> 64:
        // 401 vs 325) and every THREE-product layer with whole chunks — TrackNetV3's fp32 checkpoints (64 -> 64 .. 512 -> 512: 314-438
        // vs 294-393 TFLOP/s) — profiles/r6u_sweep_h2r_nf2.txt
        if (conv_h2r_supported(a)) {
            const int ntiles = (n16 + 3) / 4;
            const float fill = (float)n16 / (float)(ntiles * 4) * (float)M / (float)(patches * 128);
            const float sc = (a.w_single ? 1.33f : 1.22f) * fill;
            if (sc > best) { best = sc; bv = 325; }
        }
        // few input channels (16 / 32 / 48: 5-14 k-steps): the wide patch kernel keeps the whole K extent of a 16 x 16 pixel
        // tile in LDS (conv_patch_h2w.hip; measured against the 8 x 16 tiles in profiles/r3_sweep_h2w.txt)
        if (conv_h2w_supported(a)) {
            const long long wpatches = (long long)(M / (a.Ho * a.Wo)) * ((a.Ho + 15) / 16) * ((a.Wo + 15) / 16);
            for (int nf = 1; nf <= 3; ++nf) {
                const int ntiles = (n16 + nf - 1) / nf;
                const float fill = (float)n16 / (float)(ntiles * nf) * (float)M / (float)(wpatches * 256);
                const long long blocks = wpatches * ntiles;
                const long long per_cu = (blocks + 255) / 256;
                const float sc = kH2WSpeed[nf - 1] * fill * (float)blocks / (256.f * (float)per_cu);
                if (sc > best) { best = sc; bv = 340 + nf; }
            }
        }
        // cin % 32 == 16 (yolov8m's 48-channel P2 layers): 14 short steps per tile — there the software-pipelined schedule
        // (313: operand reads of the next step under this step's main products) measured +6..9 % although it runs 2 waves
        // per SIMD instead of 3; on whole-chunk layers it loses 10-15 % (profiles/conv_h2_sweep_r3f_pipe.txt).  Same
        // products in the same order: results do not depend on the choice.
        if (bv == 303 && (a.cin & 16)) bv = 313;
    }
    return bv;
}

}  // namespace padel
