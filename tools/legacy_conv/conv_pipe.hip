// K3/K4 v3 (opt-in, PADEL_CONV_PIPE=1) — deeper software pipeline of the LDS-staged implicit-GEMM conv
// (see conv_lds.hip for layout, swizzle and numerics; results are bit-identical).  Motivation:
// with the 2-stage kernel a lone wave keeps the matrix pipe only ~40 % busy because ds_write, barrier and
// ds_read sit between two MFMA bursts.  Here the LDS ring has 3 stages and fragments / staging registers are
// double-buffered, so everything a burst needs is in registers before the barrier that precedes it.
#include "kernels.h"
#include <cstdio>
#include <cstdlib>

namespace padel {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply3(float v, int act) {
    if (act == ACT_SILU) return v / (1.0f + expf(-v));
    if (act == ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

template <int WM, int WN, int MF, int NF, int KS>
__global__ void __launch_bounds__(256) conv_pipe_kernel(const ConvArgs a) {
    constexpr int TAPS = KS * KS;
    constexpr int pad = KS >> 1;
    constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
    constexpr int AP = BM / 64;              // A rows staged per thread
    constexpr int BP = (BN + 63) / 64;       // B rows staged per thread (last pass may be partial)
    static_assert(WM * WN == 4, "4 waves per workgroup");
    __shared__ __attribute__((aligned(16))) float lds[3 * (BM + BN) * 16];       // 3-stage ring
    float* const As = lds;
    float* const Bs = lds + 3 * BM * 16;

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

    const bool prio = (a.tune & 1) != 0;
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

    constexpr int FLUSH = (KS == 3) ? TAPS * 2 : 16;       // k-steps per accumulation block (even)
    f32x4 ga0[AP], gb0[BP], ga1[AP], gb1[BP];              // global staging sets
    f32x4 FA0[MF], FB0[NF], FA1[MF], FB1[NF];              // MFMA fragment sets
    int pf_tap = 0, pf_half = 0, pf_c32 = 0, pf_ks = 0;    // k-position of the step being prefetched (SALU)
    auto gload = [&](f32x4(&ga)[AP], f32x4(&gb)[BP], const int ks_, const int tap, const int c0) {
        const bool live = ks_ < nks;                       // steps past the end read the zero page
        const int ky = (KS == 3) ? (tap * 11) >> 5 : 0;
        const int kx = (KS == 3) ? tap - ky * 3 : 0;
        const long long toff = ((long long)ky * a.W + kx) * a.in_cs + c0;
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const bool v = live && (unsigned)(iy0[p] + ky) < (unsigned)a.H && (unsigned)(ix0[p] + kx) < (unsigned)a.W;
            const float* ptr = v ? a.in + (aoff[p] + toff) : a.zeros;
            ga[p] = *reinterpret_cast<const f32x4*>(ptr);
        }
#pragma unroll
        for (int p = 0; p < BP; ++p)
            if (BN % 64 == 0 || srow + 64 * p < BN)
                gb[p] = *reinterpret_cast<const f32x4*>(live ? wrow[p] + ks_ * 16 : a.zeros);
    };
#define PADEL_GLOAD(GA, GB)                                                            \
    do {                                                                               \
        gload(GA, GB, pf_ks, pf_tap, pf_c32 * 32 + pf_half * 16);                      \
        ++pf_ks;                                                                       \
        if (pf_c32 >= nfull) { ++pf_tap; }                                             \
        else { pf_half ^= 1; if (!pf_half) { if (++pf_tap == TAPS) { pf_tap = 0; ++pf_c32; } } } \
    } while (0)
    auto lstore = [&](const f32x4(&ga)[AP], const f32x4(&gb)[BP], const int buf) {
        float* ad = As + buf * (BM * 16) + st_off;
        float* bd = Bs + buf * (BN * 16) + st_off;
#pragma unroll
        for (int p = 0; p < AP; ++p) *reinterpret_cast<f32x4*>(ad + p * 64 * 16) = ga[p];
#pragma unroll
        for (int p = 0; p < BP; ++p)
            if (BN % 64 == 0 || srow + 64 * p < BN) *reinterpret_cast<f32x4*>(bd + p * 64 * 16) = gb[p];
    };
    auto fread = [&](f32x4(&FA)[MF], f32x4(&FB)[NF], const int buf) {
        const float* ab = As + buf * (BM * 16) + a_rd;
        const float* bb = Bs + buf * (BN * 16) + b_rd;
#pragma unroll
        for (int f = 0; f < MF; ++f) FA[f] = *reinterpret_cast<const f32x4*>(ab + f * 256);
#pragma unroll
        for (int j = 0; j < NF; ++j) FB[j] = *reinterpret_cast<const f32x4*>(bb + j * 256);
    };
    auto mma = [&](const f32x4(&FA)[MF], const f32x4(&FB)[NF]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    part[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(FA[f][kk], FB[j][kk], part[f][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // Invariants at the start of iteration k:  F[k&1] = fragments of step k (registers);  LDS stage (k+1)%3 =
    // step k+1 (visible);  G[k&1] = global data of step k+2 (loads possibly still in flight);  G[(k+1)&1] free.
    // Iteration k:  issue global loads of step k+3 -> G[(k+1)&1];  ds_write step k+2 from G[k&1] -> stage (k+2)%3;
    //               ds_read fragments of step k+1 from stage (k+1)%3 -> F[(k+1)&1];  MFMAs on F[k&1];  barrier.
    // Stage (k+2)%3 was last read (as step k-1) in iteration k-2; two barriers lie in between.  Nothing issued
    // after the barrier is needed before the NEXT barrier except registers that are already loaded, so a wave's
    // MFMA burst starts immediately after the barrier and LDS / global latencies drain underneath it.
    PADEL_GLOAD(ga0, gb0);                 // step 0
    PADEL_GLOAD(ga1, gb1);                 // step 1
    lstore(ga0, gb0, 0);
    lstore(ga1, gb1, 1);
    PADEL_GLOAD(ga0, gb0);                 // step 2
    __syncthreads();
    fread(FA0, FB0, 0);
    int ks = 0, st = 0;                    // st = ks % 3
    while (ks < nks - 1) {
        const int nb = min(FLUSH, nks - 1 - ks);          // FLUSH is even: blocks start on even ks
        for (int i = 0; i < nb; i += 2) {
            {   // even k-step
                const int s1 = st == 2 ? 0 : st + 1, s2 = s1 == 2 ? 0 : s1 + 1;
                PADEL_GLOAD(ga1, gb1);                     // step ks+3
                lstore(ga0, gb0, s2);                      // step ks+2
                fread(FA1, FB1, s1);                       // step ks+1
                mma(FA0, FB0);
                __syncthreads();
                st = s1; ++ks;
            }
            if (i + 1 < nb) {   // odd k-step
                const int s1 = st == 2 ? 0 : st + 1, s2 = s1 == 2 ? 0 : s1 + 1;
                PADEL_GLOAD(ga0, gb0);
                lstore(ga1, gb1, s2);
                fread(FA0, FB0, s1);
                mma(FA1, FB1);
                __syncthreads();
                st = s1; ++ks;
            }
        }
        if (nb == FLUSH) {
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
    }
    if ((nks - 1) & 1) mma(FA1, FB1); else mma(FA0, FB0);
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
                    float v = act_apply3(acc[f][j][rr] + b, act);
                    if (a.res) v += a.res[(long long)m * a.res_cs + a.res_choff + co];
                    a.out[(long long)m * a.out_cs + a.out_choff + co] = v;
                }
            }
        }
    }
}

template <int WM, int WN, int MF, int NF>
static hipError_t launch_p(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    constexpr int BM = WM * MF * 16;
    a.n_mtiles = (a.M + BM - 1) / BM;
    dim3 grid(a.n_mtiles, (a.n16 + WN * NF - 1) / (WN * NF), 1);
    if (a.ksize == 3) hipLaunchKernelGGL((conv_pipe_kernel<WM, WN, MF, NF, 3>), grid, dim3(256), 0, s, a);
    else if (a.ksize == 1) hipLaunchKernelGGL((conv_pipe_kernel<WM, WN, MF, NF, 1>), grid, dim3(256), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// same variant ids as conv_lds.hip; returns hipErrorNotSupported for tiles this kernel is not instantiated for
hipError_t launch_conv_pipe(const ConvArgs& a, int variant, hipStream_t s) {
    switch (variant) {
        case 0: return launch_p<2, 2, 4, 4>(a, s);
        case 1: return launch_p<2, 2, 4, 3>(a, s);
        case 6: return launch_p<2, 2, 2, 4>(a, s);
        case 7: return launch_p<2, 2, 2, 3>(a, s);
        case 9: return launch_p<4, 1, 2, 4>(a, s);
        case 10: return launch_p<2, 2, 4, 2>(a, s);
        case 11: return launch_p<4, 1, 2, 2>(a, s);
    }
    return hipErrorNotSupported;
}

}  // namespace padel
