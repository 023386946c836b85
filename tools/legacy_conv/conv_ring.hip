// K3/K4 v4 — LDS-DMA ring: implicit-GEMM conv on v_mfma_f32_16x16x4_f32 whose operand tiles travel
// global -> LDS with `global_load_lds_dwordx4` (no staging registers, no ds_write) into a 3-stage ring.
//
// Same tiles, operands, K order, swizzled LDS image and two-level accumulation as conv_lds.hip, so the
// results are bit-identical with v1/v2.  What changes is the dependency structure of one k-step.  The
// k-step decomposition of v2 (tools/diag_probe.sh, profiles/conv_diag_r1.txt; 192->192 3x3, 64x96 tile, full
// residency) is: MFMA only 1.28 ms, + fragment ds_reads 1.36, + ds_writes and barrier 1.46, + the global loads
// that feed those writes 1.74 — the `global_load -> s_waitcnt vmcnt -> ds_write -> barrier` chain of every wave
// is what the other resident waves fail to cover.  Here
//
//   * step k+2's tile is requested right after the barrier of step k, as LDS-DMA: lane l of a wave-instruction
//     moves 16 bytes to (wave-uniform LDS base + 16 l).  A wave-instruction therefore fills 16 consecutive
//     64-byte LDS rows linearly; the XOR swizzle of conv_lds.hip (slot s of row r holds chunk s ^ f(r>>2)) is
//     applied on the SOURCE side: lane (row, s) fetches chunk s ^ f of its row;
//   * a wave waits only with a counted `s_waitcnt vmcnt(n)` for the step it is about to read (the younger
//     request stays in flight across the barrier: two k-steps, >1500 cycles, of prefetch distance);
//   * one raw `s_barrier` per k-step orders both hazards: RAW (every wave's vmcnt wait for step k precedes the
//     barrier, the ds_reads of step k follow it) and WAR (stage (k+2)%3 was last read in iteration k-1 and every
//     wave consumed those fragments before arriving at the barrier of iteration k).
//
// The LDS-DMA requests are inline asm: hipcc neither counts them nor inserts `vmcnt(0)` for them.
#include "kernels.h"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace padel {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply4(float v, int act) {
    if (act == ACT_SILU) return v / (1.0f + expf(-v));
    if (act == ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

// 64 lanes x 16 bytes: global (per-lane address) -> LDS (lds_dst + 16 * lane); M0 carries the LDS base
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// minimum waves per SIMD asked of the register allocator: 4-wave workgroups of 6 fragments -> 5 workgroups per CU;
// 8-wave workgroups -> 3 (6 fragments) or 2 (8 fragments) per CU; 16-wave workgroups -> 1 per CU
constexpr int ring_min_waves(int nw, int frags) {
    return nw == 4 ? (frags <= 6 ? 5 : 1) : nw == 8 ? (frags <= 6 ? 6 : (frags <= 8 ? 4 : 2)) : 4;
}

template <int WM, int WN, int MF, int NF, int KS>
__global__ void __launch_bounds__(64 * WM * WN, ring_min_waves(WM * WN, MF * NF)) conv_ring_kernel(const ConvArgs a) {
    constexpr int TAPS = KS * KS;
    constexpr int pad = KS >> 1;
    constexpr int NW = WM * WN;              // waves per workgroup (4, 8 or 16)
    constexpr int RP = NW * 16;              // tile rows staged per pass: one wave-instruction (16 rows) per wave
    constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
    constexpr int AP = (BM + RP - 1) / RP;   // A passes; in the last one only waves with 16*wave < BM % RP take part
    constexpr int BP = (BN + RP - 1) / RP;   // B passes, likewise
    constexpr int NST = 3;
    constexpr int STAGE = (BM + BN) * 16;    // floats per ring stage: A rows then B rows, 64 bytes each
    static_assert(NW == 4 || NW == 8 || NW == 16, "4, 8 or 16 waves per workgroup");
    __shared__ __attribute__((aligned(16))) float lds[NST * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;

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
    const int HoWo = a.Ho * a.Wo;
    const int nfull = a.cin >> 5;
    const int nks = nfull * TAPS * 2 + ((a.cin & 16) ? TAPS : 0);
    const int Ktot = nks * 16;

    // ---- staging assignment: lane -> (row = tid>>2 (+64 per pass), LDS slot = tid&3, source chunk = slot ^ f(row))
    const int srow = tid >> 2;
    const int sc = (tid & 3) ^ ((4 - ((srow >> 2) & 3)) & 3);     // +RP rows keeps (row>>2)&3: same for every pass
    long long aoff[AP];
    int iy0[AP], ix0[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        int m = m0 + srow + RP * p;
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
        const int rr = srow + RP * p;                       // row inside the BN tile
        const int frag = min(f0 + (rr >> 4), a.n16 - 1);    // clamp: partial last channel tile
        wrow[p] = a.w + ((long long)(frag * 16 + (rr & 15))) * Ktot + sc * 4;
    }
    // LDS byte addresses (the low 32 bits of a flat LDS address are the LDS offset)
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    const unsigned dma_a = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024u);                 // + pass * 4096 + stage
    const unsigned dma_b = __builtin_amdgcn_readfirstlane(lds0 + BM * 64u + wave * 1024u);
    const int ld_off = lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);
    const int a_rd = (wm * MF * 16) * 16 + ld_off;
    const int b_rd = BM * 16 + (wn * NF * 16) * 16 + ld_off;
    // which passes this wave takes part in (wave-uniform), and how many requests that makes per k-step
    bool aon[AP], bon[BP];
    int nreq = 0;
#pragma unroll
    for (int p = 0; p < AP; ++p) { aon[p] = p * RP + wave * 16 < BM; nreq += aon[p] ? 1 : 0; }
#pragma unroll
    for (int p = 0; p < BP; ++p) { bon[p] = p * RP + wave * 16 < BN; nreq += bon[p] ? 1 : 0; }

    f32x4 acc[MF][NF], part[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    constexpr int FLUSH = (KS == 3) ? TAPS * 2 : 16;       // k-steps per accumulation block
    // request the tile of k-step ks_ (tap, first channel c0) into ring stage `st`; steps past the end request
    // the zero page (keeps the per-iteration request count, and with it the vmcnt arithmetic, constant)
    auto request = [&](const int st, const int ks_, const int tap, const int c0) {
        const bool live = ks_ < nks;
        const int ky = (KS == 3) ? (tap * 11) >> 5 : 0;          // tap / 3 for tap in [0, 9)
        const int kx = (KS == 3) ? tap - ky * 3 : 0;
        const long long toff = ((long long)ky * a.W + kx) * a.in_cs + c0;
        const unsigned sb = (unsigned)st * (STAGE * 4u);
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const bool v = live && (unsigned)(iy0[p] + ky) < (unsigned)a.H && (unsigned)(ix0[p] + kx) < (unsigned)a.W;
            const float* ptr = v ? a.in + (aoff[p] + toff) : a.zeros;
            if (aon[p]) glds16(ptr, __builtin_amdgcn_readfirstlane(dma_a + sb + p * (RP * 64u)));
        }
#pragma unroll
        for (int p = 0; p < BP; ++p)
            if (bon[p])
                glds16(live ? wrow[p] + ks_ * 16 : a.zeros, __builtin_amdgcn_readfirstlane(dma_b + sb + p * (RP * 64u)));
    };
    auto compute = [&](const int st) {
        const float* ab = lds + st * STAGE + a_rd;
        const float* bb = lds + st * STAGE + b_rd;
        f32x4 A[MF], B[NF];
#pragma unroll
        for (int f = 0; f < MF; ++f) A[f] = *reinterpret_cast<const f32x4*>(ab + f * 256);
#pragma unroll
        for (int j = 0; j < NF; ++j) B[j] = *reinterpret_cast<const f32x4*>(bb + j * 256);
        if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    part[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[f][kk], B[j][kk], part[f][j], 0, 0, 0);
        if (prio) __builtin_amdgcn_s_setprio(0);
    };

    // k-position of the next step to REQUEST, kept incrementally in scalar registers
    int pf_tap = 0, pf_half = 0, pf_c32 = 0;
#define PADEL_RQ_ADVANCE()                                                              \
    do {                                                                               \
        if (pf_c32 >= nfull) { ++pf_tap; }                                             \
        else { pf_half ^= 1; if (!pf_half) { if (++pf_tap == TAPS) { pf_tap = 0; ++pf_c32; } } } \
    } while (0)

    request(0, 0, pf_tap, pf_c32 * 32 + pf_half * 16); PADEL_RQ_ADVANCE();
    request(1, 1, pf_tap, pf_c32 * 32 + pf_half * 16); PADEL_RQ_ADVANCE();
    int rd = 0, wr = 2;                                     // ring stage of step k / of step k+2
    int k = 0;
    while (k < nks) {
        const int nb = min(FLUSH, nks - k);
        for (int i = 0; i < nb; ++i, ++k) {
            // own requests of step k have landed (those of step k+1 may still be in flight) ...
            if (nreq == AP + BP) wait_vmcnt<AP + BP>();
            else if (nreq == AP + BP - 1) wait_vmcnt<AP + BP - 1>();
            else wait_vmcnt<(AP + BP >= 2 ? AP + BP - 2 : 0)>();
            __builtin_amdgcn_s_barrier();                   // ... and so have everybody else's; stage `wr` is free
            request(wr, k + 2, pf_tap, pf_c32 * 32 + pf_half * 16);
            PADEL_RQ_ADVANCE();
            __builtin_amdgcn_sched_barrier(0);
            compute(rd);
            __builtin_amdgcn_sched_barrier(0);
            rd = (rd == NST - 1) ? 0 : rd + 1;
            wr = (wr == NST - 1) ? 0 : wr + 1;
        }
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
    wait_vmcnt<0>();        // the two trailing zero-page requests must land before this workgroup's LDS is released

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
                    float v = act_apply4(acc[f][j][rr] + b, act);
                    if (a.res) v += a.res[(long long)m * a.res_cs + a.res_choff + co];
                    a.out[(long long)m * a.out_cs + a.out_choff + co] = v;
                }
            }
        }
    }
}

template <int WM, int WN, int MF, int NF>
static hipError_t launch_r(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    constexpr int BM = WM * MF * 16;
    a.n_mtiles = (a.M + BM - 1) / BM;
    dim3 grid(a.n_mtiles, (a.n16 + WN * NF - 1) / (WN * NF), 1);
    const size_t dyn = getenv("PADEL_CONV_DYNLDS") ? (size_t)atoi(getenv("PADEL_CONV_DYNLDS")) : 0;
    if (getenv("PADEL_CONV_OCC")) {
        int nb = -1;
        hipFuncAttributes fa{};
        if (a.ksize == 3) {
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_ring_kernel<WM, WN, MF, NF, 3>, 64 * WM * WN, dyn);
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(conv_ring_kernel<WM, WN, MF, NF, 3>));
        }
        fprintf(stderr, "[occ ring] tile %dx%d ks%d: %d workgroups/CU (dyn LDS %zu), regs %d, static LDS %zu\n", BM,
                WN * NF * 16, a.ksize, nb, dyn, fa.numRegs, fa.sharedSizeBytes);
    }
    if (a.ksize == 3) hipLaunchKernelGGL((conv_ring_kernel<WM, WN, MF, NF, 3>), grid, dim3(64 * WM * WN), dyn, s, a);
    else if (a.ksize == 1) hipLaunchKernelGGL((conv_ring_kernel<WM, WN, MF, NF, 1>), grid, dim3(64 * WM * WN), dyn, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// same variant ids as conv_lds.hip; hipErrorNotSupported for tiles this kernel is not instantiated for
hipError_t launch_conv_ring(const ConvArgs& a, int variant, hipStream_t s) {
    switch (variant) {
        case 1: return launch_r<2, 2, 4, 3>(a, s);    // 128 x  96
        case 4: return launch_r<4, 1, 4, 2>(a, s);    // 256 x  32
        case 6: return launch_r<2, 2, 2, 4>(a, s);    //  64 x 128
        case 7: return launch_r<2, 2, 2, 3>(a, s);    //  64 x  96
        case 9: return launch_r<4, 1, 2, 4>(a, s);    // 128 x  64
        case 10: return launch_r<2, 2, 4, 2>(a, s);   // 128 x  64 (2x2 waves)
        case 11: return launch_r<4, 1, 2, 2>(a, s);   // 128 x  32
        case 12: return launch_r<4, 1, 2, 1>(a, s);   // 128 x  16
        // ring-only tiles: 8 / 16 waves share a larger tile (fewer L2 bytes per flop, same registers per wave)
        case 13: return launch_r<4, 2, 2, 3>(a, s);   // 128 x  96,  8 waves
        case 14: return launch_r<4, 2, 2, 4>(a, s);   // 128 x 128,  8 waves
        case 15: return launch_r<4, 2, 2, 2>(a, s);   // 128 x  64,  8 waves
        case 16: return launch_r<8, 2, 2, 3>(a, s);   // 256 x  96, 16 waves
        case 17: return launch_r<4, 4, 2, 3>(a, s);   // 128 x 192, 16 waves
        case 18: return launch_r<2, 4, 2, 3>(a, s);   //  64 x 192,  8 waves
        case 19: return launch_r<8, 1, 2, 3>(a, s);   // 256 x  48,  8 waves
        case 20: return launch_r<4, 1, 2, 3>(a, s);   // 128 x  48
    }
    return hipErrorNotSupported;
}

}  // namespace padel
