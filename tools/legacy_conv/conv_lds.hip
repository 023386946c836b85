// K3/K4 v2 — LDS-staged implicit-GEMM conv on v_mfma_f32_16x16x4_f32 (gfx950).
//
// Same math, operands, K order and two-level accumulation as conv_igemm.hip (so results are
// bit-identical between the two kernels); what changes is how operands reach the matrix pipe.
// The register-direct kernel issues one 16-byte global load per fragment per wave and tops out at
// ~80-90 TFLOP/s because every wave re-fetches the weight tile and its pixels from L1/L2.  Here a
// workgroup of 4 waves (WM x WN) shares a BM x BN output tile:
//
//   * per k-step (16 channels of one 3x3 tap) the 256 threads stage A[BM][16] (im2col rows, zero-padded
//     taps read a zero page) and Wt[BN][16] into LDS with ONE 16-byte global load per 64 bytes of tile
//     row -> 2-4 loads per thread feed 32-64 MFMAs per wave (the direct kernel needs 6-8);
//   * LDS rows are 64 bytes; the 16-byte chunk index is XOR-swizzled with f(row>>2) = (4-(row>>2))&3 so
//     that both the 8-lane ds_write_b128 groups and the four 16-lane ds_read_b128 groups of a wave touch
//     16 distinct 16-byte bank slots (conflict-free, MI355X_MICROARCH.md §LDS);
//   * double-buffered: global loads for step k+1 are in flight (in registers) while step k's fragments
//     are read from LDS and its MF*NF*4 MFMAs issue; registers are written to the other LDS buffer
//     after the MFMAs; one __syncthreads per k-step.
//   * fragments: lane l reads the 16 bytes [4*(l>>4), +4) of row (l&15) -> exactly the A[i][k=l>>4]
//     / B[k=l>>4][j] operand layout of 16x16x4, four MFMAs per ds_read_b128 pair.
#include "kernels.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace padel {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply2(float v, int act) {
    if (act == ACT_SILU) return v / (1.0f + expf(-v));
    if (act == ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    if (act == ACT_LEAKY) return v >= 0.0f ? v : 0.01f * v;
    return v;
}

// second launch-bound = minimum waves per SIMD the register allocation must allow: small per-wave tiles are asked
// to fit 5 resident workgroups per CU (<= 96 registers); measured residency curve of the 64x96 tile: 1 -> 60,
// 2 -> 81, 3 -> 90, 4 -> 95 TFLOP/s (tools/occ_probe.sh)
// DIAG (tuning probes only, results are WRONG when != 0; PADEL_CONV_DIAG, variant 7 / 3x3 only): bit0 = no global
// loads in the main loop, bit1 = no LDS stores, bit2 = no barrier, bit3 = no LDS fragment reads.
template <int WM, int WN, int MF, int NF, int KS, int KB, int DIAG = 0>
__global__ void __launch_bounds__(256, (MF * NF <= 6 && KB == 1) ? 5 : 1) conv_lds_kernel(const ConvArgs a) {
    constexpr int TAPS = KS * KS;
    constexpr int pad = KS >> 1;
    constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
    constexpr int AP = BM / 64;              // A rows staged per thread
    constexpr int BP = (BN + 63) / 64;       // B rows staged per thread (last pass may be partial)
    static_assert(WM * WN == 4, "4 waves per workgroup");
    // KB k-steps (16 channels each) are staged per barrier interval
    __shared__ __attribute__((aligned(16))) float lds[2 * KB * (BM + BN) * 16];
    float* const As = lds;
    float* const Bs = lds + 2 * KB * BM * 16;
    // DIAG 16: every wave stamps s_memtime at 5 points of every k-step into an LDS ring (lane 0 writes the real slot,
    // the other lanes write a per-lane dummy slot: no branch, no bank conflict), dumped to a.dbg at the end
    __shared__ unsigned long long stamps[(DIAG & 16) ? 4 * kConvDbgSteps * 5 + 4 * 64 : 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    unsigned long long t_begin = 0;
    if constexpr ((DIAG & 16) != 0) t_begin = __builtin_amdgcn_s_memtime();
#define PADEL_STAMP(step, slot)                                                                        \
    do {                                                                                               \
        if constexpr ((DIAG & 16) != 0) {                                                              \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                \
            const int real_ = (wave * kConvDbgSteps + ((step) & (kConvDbgSteps - 1))) * 5 + (slot);   \
            const int dummy_ = 4 * kConvDbgSteps * 5 + wave * 64 + lane;                               \
            stamps[lane == 0 ? real_ : dummy_] = t_;                                                   \
        }                                                                                              \
    } while (0)

    // XCD-aware (bijective) remap of the pixel-tile index
    const int nmt = a.n_mtiles;
    const int bid = blockIdx.x;
    const int q = nmt >> 3, r = nmt & 7, xcd = bid & 7, idx = bid >> 3;
    const int mt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int nt = blockIdx.y;
    const int m0 = mt * BM;
    const int f0 = nt * (WN * NF);           // first 16-channel fragment of this workgroup

    bool prio = (a.tune & 1) != 0;
    if (a.tune & 12) {
        // static per-workgroup priority instead of the toggle: co-resident workgroups that arbitrate MFMA by MFMA
        // finish their bursts together and then sit in their load/barrier phases together (pipe idle); distinct
        // priorities let one burst run through while the others wait, which staggers the phases
        const int pl = (a.tune & 4) ? (idx >> 5) & 3 : idx & 3;
        if (pl == 1) __builtin_amdgcn_s_setprio(1);
        else if (pl == 2) __builtin_amdgcn_s_setprio(2);
        else if (pl == 3) __builtin_amdgcn_s_setprio(3);
        prio = false;
    }
    if (a.tune & 2) {
        // co-resident workgroups start together and run at the same rate: without an offset their
        // non-MFMA phases (barrier, LDS refill) coincide and the matrix pipe idles
        const int ph = (bid >> 3) & 3;
        if (ph == 1) __builtin_amdgcn_s_sleep(4);
        else if (ph == 2) __builtin_amdgcn_s_sleep(8);
        else if (ph == 3) __builtin_amdgcn_s_sleep(12);
    }
    const int HoWo = a.Ho * a.Wo;
    const int nfull = a.cin >> 5;
    const int steps_full = nfull * TAPS * 2;
    const int nks = steps_full + ((a.cin & 16) ? TAPS : 0);
    const int Ktot = nks * 16;

    // ---- staging assignment: thread -> (row = tid>>2 (+64 per pass), 16-byte chunk = tid&3)
    const int srow = tid >> 2, sc = tid & 3;
    long long aoff[AP];
    int iy0[AP], ix0[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        int m = m0 + srow + 64 * p;
        const bool rv = m < a.M;
        if (!rv) m = 0;
        const int n = m / HoWo;
        const int rem = m - n * HoWo;
        const int oy = rem / a.Wo;
        const int ox = rem - oy * a.Wo;
        iy0[p] = rv ? oy * a.stride - pad : -(1 << 20);
        ix0[p] = ox * a.stride - pad;
        aoff[p] = (((long long)n * a.H + (oy * a.stride - pad)) * a.W + ix0[p]) * a.in_cs + a.in_choff + sc * 4;
    }
    const float* wrow[BP];
#pragma unroll
    for (int p = 0; p < BP; ++p) {
        const int rr = srow + 64 * p;                       // row inside the BN tile
        const int frag = min(f0 + (rr >> 4), a.n16 - 1);    // clamp: partial last channel tile
        wrow[p] = a.w + ((long long)(frag * 16 + (rr & 15))) * Ktot + sc * 4;
    }
    // swizzled LDS offsets (floats)
    const int swz_w = ((sc ^ ((4 - ((srow >> 2) & 3)) & 3)) << 2);       // same for every pass: +64 rows keeps (row>>2)&3
    const int st_off = srow * 16 + swz_w;
    const int ld_off = lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);
    const int a_rd = (wm * MF * 16) * 16 + ld_off;
    const int b_rd = (wn * NF * 16) * 16 + ld_off;

    f32x4 acc[MF][NF], part[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    constexpr int FLUSH = (KS == 3) ? TAPS * 2 : 16;       // k-steps per accumulation block
    f32x4 ga[KB][AP], gb[KB][BP];
    // k-position of the step being PREFETCHED, kept incrementally in scalar registers: (pf_c0 = first
    // channel of the 16-wide slice, pf_tap); `gload` is called for steps 0,1,2,... in order.
    int pf_tap = 0, pf_half = 0, pf_c32 = 0;
    auto gload = [&](const int u, const int ks_, const int tap, const int c0) {
        const bool live = ks_ < nks;                              // odd tail of a KB=2 pair: all-zero operands
        const int ky = (KS == 3) ? (tap * 11) >> 5 : 0;          // tap / 3 for tap in [0, 9)
        const int kx = (KS == 3) ? tap - ky * 3 : 0;
        const long long toff = ((long long)ky * a.W + kx) * a.in_cs + c0;
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const bool v = live && (unsigned)(iy0[p] + ky) < (unsigned)a.H && (unsigned)(ix0[p] + kx) < (unsigned)a.W;
            const float* ptr = v ? a.in + (aoff[p] + toff) : a.zeros;
            ga[u][p] = *reinterpret_cast<const f32x4*>(ptr);
        }
#pragma unroll
        for (int p = 0; p < BP; ++p)
            if (BN % 64 == 0 || srow + 64 * p < BN)
                gb[u][p] = *reinterpret_cast<const f32x4*>(live ? wrow[p] + ks_ * 16 : a.zeros);
    };
    (void)steps_full;
    auto lstore = [&](const int u, const int buf) {
        float* ad = As + (buf * KB + u) * (BM * 16) + st_off;
        float* bd = Bs + (buf * KB + u) * (BN * 16) + st_off;
#pragma unroll
        for (int p = 0; p < AP; ++p) *reinterpret_cast<f32x4*>(ad + p * 64 * 16) = ga[u][p];
#pragma unroll
        for (int p = 0; p < BP; ++p)
            if (BN % 64 == 0 || srow + 64 * p < BN) *reinterpret_cast<f32x4*>(bd + p * 64 * 16) = gb[u][p];
    };
    int stamp_step = 0;
    auto compute = [&](const int u, const int buf) {
        const float* ab = As + (buf * KB + u) * (BM * 16) + a_rd;
        const float* bb = Bs + (buf * KB + u) * (BN * 16) + b_rd;
        f32x4 A[MF], B[NF];
        if (DIAG & 8) {
#pragma unroll
            for (int f = 0; f < MF; ++f) A[f] = ga[0][f % AP];
#pragma unroll
            for (int j = 0; j < NF; ++j) B[j] = gb[0][j % BP];
        } else {
#pragma unroll
            for (int f = 0; f < MF; ++f) A[f] = *reinterpret_cast<const f32x4*>(ab + f * 256);
#pragma unroll
            for (int j = 0; j < NF; ++j) B[j] = *reinterpret_cast<const f32x4*>(bb + j * 256);
        }
        if constexpr ((DIAG & 16) != 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PADEL_STAMP(stamp_step, 1);
        }
        if (prio) __builtin_amdgcn_s_setprio(1);          // keep the matrix pipe fed ahead of waves in their load phase
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    part[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[f][kk], B[j][kk], part[f][j], 0, 0, 0);
        if (prio) __builtin_amdgcn_s_setprio(0);
        PADEL_STAMP(stamp_step, 2);
    };

    // Nested loops: the inner loop (one accumulation block of FLUSH k-steps) contains nothing but
    // loads / LDS traffic / MFMAs on `part`; the part -> acc flush sits between inner loops.  With the flush
    // as a branch INSIDE the k-loop hipcc moved every accumulator AGPR<->VGPR on every k-step
    // (48 v_accvgpr moves per 24 MFMAs in the 64x96 variant).
#define PADEL_PF_ADVANCE()                                                              \
    do {                                                                               \
        if (pf_c32 >= nfull) { ++pf_tap; }                                             \
        else { pf_half ^= 1; if (!pf_half) { if (++pf_tap == TAPS) { pf_tap = 0; ++pf_c32; } } } \
    } while (0)
    // super-step = KB consecutive k-steps between two barriers
    const int nss = (nks + KB - 1) / KB;
    constexpr int FLUSH_SS = FLUSH / KB;                    // FLUSH is even
#pragma unroll
    for (int u = 0; u < KB; ++u) { gload(u, u, pf_tap, pf_c32 * 32 + pf_half * 16); PADEL_PF_ADVANCE(); }
#pragma unroll
    for (int u = 0; u < KB; ++u) lstore(u, 0);
    __syncthreads();
    int ss = 0;
    while (ss < nss - 1) {
        const int nb = min(FLUSH_SS, nss - 1 - ss);
        for (int i = 0; i < nb; ++i, ++ss) {
            PADEL_STAMP(ss, 0);
#pragma unroll
            for (int u = 0; u < KB; ++u) {                 // in flight under the MFMAs below
                if (!(DIAG & 1)) gload(u, (ss + 1) * KB + u, pf_tap, pf_c32 * 32 + pf_half * 16);
                PADEL_PF_ADVANCE();
            }
            __builtin_amdgcn_sched_barrier(0);
            stamp_step = ss;
#pragma unroll
            for (int u = 0; u < KB; ++u) compute(u, ss & 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((DIAG & 16) != 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PADEL_STAMP(ss, 3);
            }
#pragma unroll
            for (int u = 0; u < KB; ++u) if (!(DIAG & 2)) lstore(u, (ss + 1) & 1);
            if (!(DIAG & 4)) __syncthreads();
            PADEL_STAMP(ss, 4);
        }
        if (nb == FLUSH_SS) {
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
    }
    stamp_step = nss - 1;
#pragma unroll
    for (int u = 0; u < KB; ++u) compute(u, (nss - 1) & 1);
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[f][j] += part[f][j];

    // epilogue: lane holds D[row = lq*4 + r][col = lr] of each 16x16 fragment
    const int act = a.act;
    const int mw = m0 + wm * MF * 16;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int fr = f0 + wn * NF + j;
        const int co = fr * 16 + lr;
        const bool cv = co < a.cout;
        const float b = a.bias[min(fr, a.n16 - 1) * 16 + lr];
#pragma unroll
        for (int f = 0; f < MF; ++f) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int m = mw + f * 16 + lq * 4 + rr;
                if (cv && m < a.M) {
                    float v = act_apply2(acc[f][j][rr] + b, act);
                    if (a.res) v += a.res[(long long)m * a.res_cs + a.res_choff + co];
                    a.out[(long long)m * a.out_cs + a.out_choff + co] = v;
                }
            }
        }
    }
    if constexpr ((DIAG & 16) != 0) {
        if (a.dbg) {
            const unsigned long long t_end = __builtin_amdgcn_s_memtime();
            __syncthreads();
            unsigned long long* d = a.dbg + (long long)(blockIdx.y * gridDim.x + blockIdx.x) * kConvDbgWords;
            for (int i = tid; i < 4 * kConvDbgSteps * 5; i += 256) d[8 + i] = stamps[i];
            if (lane == 0) {
                // HW_REG_HW_ID (4) and HW_REG_XCC_ID (20): which CU / SIMD / XCD this wave ran on
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
                const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
                d[wave] = ((unsigned long long)xcc << 32) | hw;
                if (wave == 0) { d[4] = t_begin; d[5] = t_end; d[6] = (unsigned long long)nks; d[7] = (unsigned long long)bid; }
            }
        }
    }
}

template <int WM, int WN, int MF, int NF>
static hipError_t launch_l(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    constexpr int BM = WM * MF * 16;
    a.n_mtiles = (a.M + BM - 1) / BM;
    dim3 grid(a.n_mtiles, (a.n16 + WN * NF - 1) / (WN * NF), 1);
    if (a.ksize == 3) hipLaunchKernelGGL((conv_lds_kernel<WM, WN, MF, NF, 3, 1>), grid, dim3(256), 0, s, a);
    else if (a.ksize == 1) hipLaunchKernelGGL((conv_lds_kernel<WM, WN, MF, NF, 1, 1>), grid, dim3(256), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// variant ids: see lds_variants[] below
struct LdsVariant { int id, wm, wn, mf, nf; };
static const LdsVariant lds_variants[] = {
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
constexpr int kNumLdsVariants = 13;      // ids 0..12 are instantiated for the LDS kernel


hipError_t launch_conv_lds(const ConvArgs& a, int variant, hipStream_t s) {
    switch (variant) {
        case 0: return launch_l<2, 2, 4, 4>(a, s);
        case 1: return launch_l<2, 2, 4, 3>(a, s);
        case 2: return launch_l<4, 1, 4, 4>(a, s);
        case 3: return launch_l<4, 1, 4, 3>(a, s);
        case 4: return launch_l<4, 1, 4, 2>(a, s);
        case 5: return launch_l<4, 1, 4, 1>(a, s);
        case 6: return launch_l<2, 2, 2, 4>(a, s);
        case 7: return launch_l<2, 2, 2, 3>(a, s);
        case 8: return launch_l<4, 1, 2, 5>(a, s);
        case 9: return launch_l<4, 1, 2, 4>(a, s);
        case 10: return launch_l<2, 2, 4, 2>(a, s);
        case 11: return launch_l<4, 1, 2, 2>(a, s);
        case 12: return launch_l<4, 1, 2, 1>(a, s);
    }
    return hipErrorNotSupported;
}

int choose_conv_lds_variant(int M, int n16) {
    // relative speeds measured on MI355X (profiles/conv_lds_sweep_r1.txt): mid-size tiles at 3-4 waves/SIMD
    // beat the 128x128 tile (1 wave/SIMD cannot hide its own barriers)
    static const float speed[] = {0.60f, 0.92f, 0.70f, 0.85f, 0.88f, 0.62f, 0.99f, 1.00f, 0.75f, 1.05f, 1.03f, 0.92f, 0.50f};
    float best = -1.f;
    int bv = 0;
    for (const auto& v : lds_variants) {
        if (v.id >= kNumLdsVariants) break;
        const int bm = v.wm * v.mf * 16, nfw = v.wn * v.nf;
        const int ntiles = (n16 + nfw - 1) / nfw;
        const long long mtiles = (M + bm - 1) / bm;
        const float fill = (float)n16 / (float)(ntiles * nfw) * (float)M / (float)(mtiles * bm);
        const long long blocks = mtiles * ntiles;
        // MFMA-bound: a CU's throughput is shared by its resident workgroups, so time ~ max workgroups per CU
        const long long per_cu = (blocks + 255) / 256;
        const float occ = (float)blocks / (256.f * (float)per_cu);
        const float sc = speed[v.id] * fill * occ;
        if (sc > best) { best = sc; bv = v.id; }
    }
    return bv;
}

int conv_lds_num_variants() { return kNumLdsVariants; }

bool conv_variant_shape(int variant, int* bm, int* bn) {
    for (const auto& v : lds_variants)
        if (v.id == variant) { *bm = v.wm * v.mf * 16; *bn = v.wn * v.nf * 16; return true; }
    return false;
}

}  // namespace padel
