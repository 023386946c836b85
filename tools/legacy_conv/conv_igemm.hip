// K3/K4 — fused Conv(+folded BN bias)+activation(+residual) as an implicit GEMM on the CDNA4
// fp32 matrix pipe (v_mfma_f32_16x16x4_f32), written for gfx950 only.
//
// Replaces what the reference reaches through ultralytics' fused Conv2d+SiLU
// (players_tracker.py:351-359, players_keypoints_tracker.py:285-292) and through
// Conv2DBlock of TrackNet (trackers/ball_tracker/models.py:5-17).
//
// GEMM view:  D[m][n] = sum_k A[m][k] * Wt[n][k]
//   m = output pixel (batch folded in, NHWC order), n = output channel, k = (c32-chunk, tap, c16-half).
//
// Design (fp32 parity path; the fp32 MFMA issues once per 32 cycles per SIMD, so the kernel's job
// is to keep that pipe fed, not to minimise bytes):
//  * one wave owns MF x NF fragments of 16x16 outputs: MF*16 consecutive pixels x NF*16 channels;
//    the 4 waves of a workgroup take 4 consecutive pixel groups -> no inter-wave sharing, no LDS,
//    no barriers: every wave is its own software pipeline.
//  * operand fragments are loaded from HBM/L2 *directly in MFMA register layout*: for
//    16x16x4, lane l supplies A[row l&15][k = l>>4]; a lane loads the 16 bytes
//    [c0 + 4*(l>>4), +4) of its pixel, so 16 lanes x 4 quarter-groups read one contiguous
//    64-byte channel run per pixel, and the four floats feed four consecutive MFMAs
//    (k-permutation inside a 16-wide k-step is harmless: A and B use the same one).
//  * K order (c32 chunk outer, 3x3 tap, 16-channel half inner) revisits each 128-byte line of
//    the input 6x back-to-back (3 kx taps x 2 halves), so tap re-reads are L1/L2 hits, not HBM.
//  * next k-step's fragments are prefetched into a second register set while the current
//    step's MF*NF*4 MFMAs issue (>= 512 cycles of cover per step).
//  * epilogue fused: + bias, SiLU / ReLU / sigmoid, + residual, store into a channel slice of the
//    consumer's concat buffer.
//  * blockIdx -> pixel-tile map is XCD-aware (each XCD walks a contiguous range of tiles so the
//    halo rows shared by neighbouring tiles stay in that XCD's L2).
#include "kernels.h"
#include <cstdlib>

namespace padel {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == ACT_SILU) return v / (1.0f + expf(-v));
    if (act == ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

template <int MF, int NF, int KS>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const ConvArgs a) {
    constexpr int TAPS = KS * KS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lr = lane & 15, lq = lane >> 4;

    // XCD-aware (bijective) remap of the pixel-tile index
    const int nmt = a.n_mtiles;
    const int bid = blockIdx.x;
    const int q = nmt >> 3, r = nmt & 7, xcd = bid & 7, idx = bid >> 3;
    const int mt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int nt = blockIdx.y;

    const int m_wave = (mt * 4 + wave) * (MF * 16);
    if (m_wave >= a.M) return;

    constexpr int taps = TAPS;
    constexpr int pad = KS >> 1;
    const int HoWo = a.Ho * a.Wo;

    long long aoff[MF];
    int iy0[MF], ix0[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        int m = m_wave + f * 16 + lr;
        const bool rv = m < a.M;
        if (!rv) m = 0;
        const int n = m / HoWo;
        const int rem = m - n * HoWo;
        const int oy = rem / a.Wo;
        const int ox = rem - oy * a.Wo;
        // invalid rows get an iy0 that fails every bounds test
        iy0[f] = rv ? oy * a.stride - pad : -(1 << 20);
        ix0[f] = ox * a.stride - pad;
        aoff[f] = (((long long)n * a.H + (oy * a.stride - pad)) * a.W + ix0[f]) * a.in_cs + a.in_choff + lq * 4;
    }

    const int nfull = a.cin >> 5;
    const int steps_full = nfull * taps * 2;
    const int nks = steps_full + ((a.cin & 16) ? taps : 0);
    const int Ktot = nks * 16;
    // the last channel tile may be partial: fragment rows past n16 are clamped onto the last real
    // fragment (recomputed, never stored) so weight / bias reads stay inside the packed blob
    const float* wlane = a.w + (long long)lr * Ktot + lq * 4;
    long long wrow[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) wrow[j] = (long long)min(nt * NF + j, a.n16 - 1) * 16 * Ktot;

    // two-level (blocked) accumulation: `part` collects one block of FLUSH k-steps (a 32-channel
    // chunk x all taps for 3x3), then is added into `acc`.  Rounding error of a length-K fp32 sum grows
    // like sqrt(K); blocking makes it sqrt(K/b) + sqrt(b), which keeps the long reductions of the
    // m-scale graphs (K up to 5184) closer to the exact result than a plain sequential chain.
    f32x4 acc[MF][NF], part[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // position of k-step `ks` in the (c32 chunk, tap, c16 half) order: pure function of the
    // wave-uniform loop counter (KS is a template constant, so the divisions are SALU mul-shifts)
    auto load = [&](const int ks_, f32x4(&A)[MF], f32x4(&B)[NF]) {
        int tap, c0;
        if (ks_ < steps_full) {
            const int c32 = ks_ / (TAPS * 2);
            const int rr = ks_ - c32 * (TAPS * 2);
            tap = rr >> 1;
            c0 = c32 * 32 + (rr & 1) * 16;
        } else {                                        // trailing 16-channel chunk (cin % 32 == 16)
            tap = ks_ - steps_full;
            c0 = nfull * 32;
        }
        const int ky = (KS == 3) ? tap / 3 : 0;
        const int kx = (KS == 3) ? tap - ky * 3 : 0;
        const long long toff = ((long long)ky * a.W + kx) * a.in_cs + c0;
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            // every load is unconditional (keeps hipcc's vmcnt counting exact): a tap that falls in
            // the zero padding, or a row past M, reads a 64-byte page of zeros instead
            const bool v = (unsigned)(iy0[f] + ky) < (unsigned)a.H && (unsigned)(ix0[f] + kx) < (unsigned)a.W;
            const float* p = v ? a.in + (aoff[f] + toff) : a.zeros;
            A[f] = *reinterpret_cast<const f32x4*>(p);
        }
#pragma unroll
        for (int j = 0; j < NF; ++j)
            B[j] = *reinterpret_cast<const f32x4*>(wlane + wrow[j] + ks_ * 16);
    };

    auto compute = [&](const f32x4(&A)[MF], const f32x4(&B)[NF]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    part[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[f][kk], B[j][kk], part[f][j], 0, 0, 0);
    };

    // software pipeline: set 0 holds step ks, set 1 is being fetched.  The main loop has no
    // conditional loads so hipcc's vmcnt bookkeeping stays exact (waits only for the older set).
    f32x4 A0[MF], B0[NF], A1[MF], B1[NF];
    // sched_barrier(0) pins "issue the next step's loads, THEN run this step's MFMAs": left alone,
    // hipcc sinks the prefetch to the tail of the MFMA chain (one register set, ~250 cycles of
    // cover); pinned, every load has a full k-step (>= MF*NF*128 cycles) to land.
    load(0, A0, B0);
    int ks = 0;
    constexpr int FLUSH = (KS == 3) ? TAPS * 2 : 16;      // k-steps per accumulation block (even)
    // nested loops: the part -> acc flush sits BETWEEN inner loops (as a branch inside the k-loop it made
    // hipcc shuffle every accumulator AGPR<->VGPR each iteration)
    while (ks + 2 < nks) {
        int pairs = 0;
        for (; pairs < FLUSH / 2 && ks + 2 < nks; ++pairs, ks += 2) {
            load(ks + 1, A1, B1);
            __builtin_amdgcn_sched_barrier(0);
            compute(A0, B0);
            __builtin_amdgcn_sched_barrier(0);
            load(ks + 2, A0, B0);
            __builtin_amdgcn_sched_barrier(0);
            compute(A1, B1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (pairs == FLUSH / 2) {
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
    }
    if (ks + 1 < nks) {
        load(ks + 1, A1, B1);
        __builtin_amdgcn_sched_barrier(0);
        compute(A0, B0);
        compute(A1, B1);
    } else {
        compute(A0, B0);
    }
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[f][j] += part[f][j];

    // epilogue: lane holds D[row = lq*4 + r][col = lr] of each 16x16 fragment
    const int act = a.act;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int co = (nt * NF + j) * 16 + lr;
        const bool cv = co < a.cout;
        const float b = a.bias[min(nt * NF + j, a.n16 - 1) * 16 + lr];
#pragma unroll
        for (int f = 0; f < MF; ++f) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int m = m_wave + f * 16 + lq * 4 + rr;
                if (cv && m < a.M) {
                    float v = act_apply(acc[f][j][rr] + b, act);
                    if (a.res) v += a.res[(long long)m * a.res_cs + a.res_choff + co];
                    a.out[(long long)m * a.out_cs + a.out_choff + co] = v;
                }
            }
        }
    }
}

template <int MF, int NF>
static hipError_t launch_t(const ConvArgs& a, hipStream_t s) {
    dim3 grid(a.n_mtiles, (a.n16 + NF - 1) / NF, 1);
    if (a.ksize == 3) hipLaunchKernelGGL((conv_igemm_kernel<MF, NF, 3>), grid, dim3(256), 0, s, a);
    else if (a.ksize == 1) hipLaunchKernelGGL((conv_igemm_kernel<MF, NF, 1>), grid, dim3(256), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_conv_igemm(const ConvArgs& a_in, int mf, int nf, hipStream_t s) {
    ConvArgs a = a_in;
    a.n_mtiles = (a.M + 4 * mf * 16 - 1) / (4 * mf * 16);
#define PADEL_CASE(MFv, NFv) if (mf == MFv && nf == NFv) return launch_t<MFv, NFv>(a, s);
    PADEL_CASE(1, 1) PADEL_CASE(1, 2) PADEL_CASE(1, 3) PADEL_CASE(1, 4) PADEL_CASE(1, 5) PADEL_CASE(1, 6)
    PADEL_CASE(2, 1) PADEL_CASE(2, 2) PADEL_CASE(2, 3) PADEL_CASE(2, 4) PADEL_CASE(2, 5) PADEL_CASE(2, 6)
    PADEL_CASE(4, 1) PADEL_CASE(4, 2) PADEL_CASE(4, 3) PADEL_CASE(4, 4) PADEL_CASE(4, 5) PADEL_CASE(4, 6)
#undef PADEL_CASE
    return hipErrorInvalidValue;
}

void choose_conv_tile(int M, int n16, int* mf_out, int* nf_out) {
    // tuning override (tools/conv_bench.py): PADEL_CONV_MF / PADEL_CONV_NF
    const int env_mf = getenv("PADEL_CONV_MF") ? atoi(getenv("PADEL_CONV_MF")) : 0;
    const int env_nf = getenv("PADEL_CONV_NF") ? atoi(getenv("PADEL_CONV_NF")) : 0;
    if (env_mf > 0 && env_nf > 0) { *mf_out = env_mf; *nf_out = env_nf; return; }
    // measured on MI355X (tools/conv_bench.py, profiles/conv_tile_sweep_r1.txt): throughput rises with
    // fragments per wave (fewer operand loads per MFMA) even at 1-2 waves/SIMD, so score each legal
    // tile by (relative speed of the shape) x (fraction of the channel tiles that is not padding)
    static const struct { int mf, nf; float speed; } cand[] = {
        {4, 4, 1.00f}, {4, 3, 0.97f}, {2, 6, 0.95f}, {4, 2, 0.88f}, {2, 4, 0.86f}, {2, 5, 0.86f}, {2, 3, 0.82f},
        {4, 1, 0.72f}, {2, 2, 0.74f}, {1, 6, 0.70f}, {1, 4, 0.62f}, {2, 1, 0.50f}, {1, 3, 0.50f}, {1, 2, 0.40f}, {1, 1, 0.30f}};
    float best = -1.f;
    int bm = 1, bn = 1;
    for (const auto& t : cand) {
        const int ntiles = (n16 + t.nf - 1) / t.nf;
        const float fill = (float)n16 / (float)(ntiles * t.nf);
        // keep >= 2 workgroups per CU in flight when the problem allows it (256 CUs)
        const long long blocks = (long long)((M + 64 * t.mf - 1) / (64 * t.mf)) * ntiles;
        const float occ = blocks >= 512 ? 1.f : (blocks >= 256 ? 0.9f : (float)blocks / 256.f * 0.8f);
        const float sc = t.speed * fill * occ;
        if (sc > best) { best = sc; bm = t.mf; bn = t.nf; }
    }
    *mf_out = bm;
    *nf_out = bn;
}

}  // namespace padel
