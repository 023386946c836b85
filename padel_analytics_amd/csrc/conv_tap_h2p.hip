// K3/K4 "h2", tap kernels with a THREE-stage activation ring (round 5).  Macros and conventions: h2_tap.h / conv_tap_h2.hip.
#include "h2_tap.h"

namespace padel {

// Ring requests of the deep-ring kernels.  LDS: activation stages 0..2 (h sub-rows | m sub-rows: ASTG_B bytes each), then the
// two weight stages (h plane | m plane).  lds_wave = LDS base + 1024 * wave; lwb0 / lwb1 = the weight stages of this wave.
#define PADEL_H2P_A(rsrc_, AS_, SA_, V0_, V1_)                                                                    \
    do {                                                                                                          \
        const unsigned sa_ = (SA_);                                                                               \
        dma3<(AS_) * ASTG_B>((V0_), rsrc_, sa_, lds_wave);                                                        \
        if constexpr (AP >= 2) dma3<(AS_) * ASTG_B + RP * 64>((V1_), rsrc_, sa_, lds_wave);                       \
        dma3<(AS_) * ASTG_B + BM * 64>((V0_), rsrc_, sa_ + 32u, lds_wave);                                        \
        if constexpr (AP >= 2) dma3<(AS_) * ASTG_B + BM * 64 + RP * 64>((V1_), rsrc_, sa_ + 32u, lds_wave);       \
    } while (0)
#define PADEL_H2P_WPL(LW_, PL_, SB_)                                                                              \
    do {                                                                                                          \
        if constexpr (BFULL >= 1) dma3<(PL_) * BN * 64>(voffB[0], rsrcB, (SB_), LW_);                             \
        if constexpr (BFULL >= 2) dma3<(PL_) * BN * 64 + RP * 64>(voffB[1], rsrcB, (SB_), LW_);                   \
        if constexpr (BP > BFULL) { if (b_last) dma3<(PL_) * BN * 64 + BFULL * RP * 64>(voffB[BP - 1], rsrcB, (SB_), LW_); } \
    } while (0)
// the two weight planes of the k-step at byte offset KB_ of a weight row into the weight stage of parity WS_
#define PADEL_H2P_W(WS_, KB_)                                                                                     \
    do {                                                                                                          \
        const unsigned kb_ = (KB_);                                                                               \
        const unsigned lw_ = ((WS_) & 1) ? lwb1 : lwb0;                                                           \
        PADEL_H2P_WPL(lw_, 0, kb_);                                                                               \
        if constexpr (!WS) PADEL_H2P_WPL(lw_, 1, kb_ + 64u);                                                      \
    } while (0)

// =====================================================================================================  1x1, deep A ring
// Round 5.  The 1x1 layers stream their activation tile from HBM / L2 once per k-step (no tap reuse), and the ring above
// requests step J + 1 at the top of step J: a step of a 128 x 96 tile is 36 MFMAs per wave (~600-1700 cycles with the
// sibling workgroup), an activation request under load takes longer than that to land, so every step starts with a wait
// (PMC: waves parked 0.33, matrix pipe 0.33-0.41 busy on the K >= 576 layers).  Here the ACTIVATION sub-rows get a
// THREE-stage ring (requested two steps ahead) and the weights — L2-resident, short round trip — keep two stages:
// 3 x 16 + 2 x 12 = 72 KB for 128 x 96, still two workgroups per CU.  Order of a wave's requests inside step J:
// W(J + 1) first, then A(J + 2); vmcnt is in-order per wave, so "all but the 2 AP newest" at the top of step J + 1 means
// W(J + 1) and A(J + 1) have landed while A(J + 2) stays in flight.  Same products in the same order: bitwise the results
// of conv_h2_1_kernel.
template <int WM, int WN, int MF, int NF, bool UP, bool WS = false>
__global__ void __launch_bounds__(64 * WM * WN, MF * NF <= 6 ? 3 : 2) conv_h2_1p_kernel(const ConvArgs a) {
    constexpr int ASTG_ = 2 * (WM * MF * 16) * 16, WSTG_ = 2 * (WN * NF * 16) * 16;      // words per activation / weight stage
    PADEL_H2T_GEOMETRY_(3 * ASTG_ + 2 * WSTG_)
    static_assert(ASTG_ == 2 * BM * 16 && WSTG_ == 2 * BN * 16, "stage sizes");
    (void)a_rd0; (void)a_rd1; (void)b_rd0; (void)b_rd1; (void)lw0; (void)lw1;
    unsigned voffA[AP], voffT[AP], voffU[AP];
    const int H2 = a.H >> 1, W2 = a.W >> 1;
    const long long linU0 = ((long long)n0 * H2 + (oy0 >> 1)) * W2;
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
        voffT[p] = sc_hi ? kOORh : voffA[p];
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
    // LDS: activation stages 0..2 (h sub-rows | m sub-rows), then weight stages 0..1 (h plane | m plane)
    constexpr int ASTG_B = ASTG_ * 4, WSTG_B = WSTG_ * 4;
    const float* const a_rs = lds + (wm_ * MF * 16) * 16 + ld_off;
    const float* b_rs0 = lds + 3 * ASTG_ + (wn_ * NF * 16) * 16 + ld_off;
    const float* b_rs1 = b_rs0 + WSTG_;
    unsigned lwb0 = __builtin_amdgcn_readfirstlane(lds_wave + 3u * (unsigned)ASTG_B);
    unsigned lwb1 = __builtin_amdgcn_readfirstlane(lwb0 + (unsigned)WSTG_B);
#define PADEL_H2P_REQA(AS_, K_)                                                                                   \
    do {                                                                                                          \
        const unsigned k_ = (K_);                                                                                 \
        if (UP && k_ < nup) { PADEL_H2P_A(rsrcU, AS_, k_ * 128u, voffU[0], voffU[AP - 1]); }                      \
        else if (half_tail && (int)k_ >= nch - 1) { PADEL_H2P_A(rsrcA, AS_, k_ * 128u, voffT[0], voffT[AP - 1]); } \
        else { PADEL_H2P_A(rsrcA, AS_, k_ * 128u, voffA[0], voffA[AP - 1]); }                                     \
    } while (0)
#define PADEL_H2P_REQW(WS_, K_) PADEL_H2P_W(WS_, (K_) * 128u)

    unsigned s_k = 0;                         // index of the first k-step of the current 9-step accumulation block
    // step J of a block (activation stage J % 3: blocks are 9 steps long; weight stage parity J & 1, swapped per block)
#define PADEL_H2P_STEP(J)                                                                                         \
    if ((J) < nb) {                                                                                               \
        if ((int)(s_k + (J) + 1) < nch) wait_vm3<2 * AP>(); else wait_vm3<0>();                                   \
        __builtin_amdgcn_s_barrier();                                                                             \
        if ((int)(s_k + (J) + 1) < nch) PADEL_H2P_REQW((J) + 1, s_k + (J) + 1);                                   \
        if ((int)(s_k + (J) + 2) < nch) PADEL_H2P_REQA(((J) + 2) % 3, s_k + (J) + 2);                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_H2T_COMPUTE_AT(a_rs + ((J) % 3) * ASTG_, (((J) & 1) ? b_rs1 : b_rs0), (J) == 0);                    \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    }
    PADEL_H2P_REQA(0, 0u);
    PADEL_H2P_REQW(0, 0u);
    if (1 < nch) PADEL_H2P_REQA(1, 1u);
    for (int k = 0; k < nch; k += 9) {
        const int nb = min(9, nch - k);
        PADEL_H2P_STEP(0) PADEL_H2P_STEP(1) PADEL_H2P_STEP(2) PADEL_H2P_STEP(3) PADEL_H2P_STEP(4)
        PADEL_H2P_STEP(5) PADEL_H2P_STEP(6) PADEL_H2P_STEP(7) PADEL_H2P_STEP(8)
        PADEL_H2T_FLUSH();
        { const float* t_ = b_rs0; b_rs0 = b_rs1; b_rs1 = t_; const unsigned u_ = lwb0; lwb0 = lwb1; lwb1 = u_; }
        s_k += 9u;
    }
    wait_vm3<0>();
    PADEL_H2T_FINISH()
#undef PADEL_H2P_STEP
#undef PADEL_H2P_REQW
#undef PADEL_H2P_REQA
}

template <int WM, int WN, int MF, int NF>
static hipError_t launch_h2t1p(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    constexpr int BM = WM * MF * 16;
    if (a.ksize != 1) return hipErrorNotSupported;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.n_ntiles = (a.n16 + WN * NF - 1) / (WN * NF);
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    if (a.in2) {
        if (a.stride != 1 || (a.up_c & 31) || a.up_c <= 0 || a.up_c > a.cin || ((a.H | a.W) & 1)) return hipErrorNotSupported;
        if (a.w_single) hipLaunchKernelGGL((conv_h2_1p_kernel<WM, WN, MF, NF, true, true>), grid, dim3(64 * WM * WN), 0, s, a);
        else hipLaunchKernelGGL((conv_h2_1p_kernel<WM, WN, MF, NF, true>), grid, dim3(64 * WM * WN), 0, s, a);
    } else {
        if (a.w_single) hipLaunchKernelGGL((conv_h2_1p_kernel<WM, WN, MF, NF, false, true>), grid, dim3(64 * WM * WN), 0, s, a);
        else hipLaunchKernelGGL((conv_h2_1p_kernel<WM, WN, MF, NF, false>), grid, dim3(64 * WM * WN), 0, s, a);
    }
    return hipGetLastError();
}


// (The same ring under the 3x3 tap walk — stride-2 layers — was built, bitwise-checked and timed in round 5: 320.5 vs 319.6 and
//  205.9 vs 202.7 TFLOP/s on 96 -> 192 / 48 -> 96 stride 2, profiles/r5c_tiles_deep_ring.txt.  A 3x3 reads every input pixel nine
//  times, its activation requests hit the L2 and land inside a step already; removed again.)

// tile ids = the 2-stage tile's id + 30 (243: 128 x 96, 239: 128 x 64)
hipError_t launch_conv_h2_deep(const ConvArgs& a, int variant, hipStream_t s) {
    if (a.ksize == 1) {
        if (variant == 243) return launch_h2t1p<4, 1, 2, 6>(a, s);
        if (variant == 239) return launch_h2t1p<4, 1, 2, 4>(a, s);
        // (64 x 96 with the deep ring — 48 KB, three workgroups per CU — measured like its two-stage sibling 207: 10-18 % behind
        //  128 x 96 on every 1x1 shape, profiles/r5o_tiles_237.txt; not kept)
        // (128 x 96 as 2 x 2 waves of 4 x 3 fragments — 14 instead of 16 operand reads per 36 MFMAs — measured 1-2 % slower than
        //  243's 4 x 1 waves of 2 x 6: profiles/r5h_tiles_1x1_wave_grid.txt; not kept)
    }
    return hipErrorNotSupported;
}

}  // namespace padel
