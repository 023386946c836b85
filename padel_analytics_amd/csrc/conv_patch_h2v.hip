// K3v "h2 wide patch, register weights" (round 6) — conv_patch_h2w.hip's tile (16 x 16 output pixels x NF channel fragments, the
// WHOLE K extent of a 16 / 32 / 48-channel input resident in LDS: yolov8m's 48 -> 48 P2 bottlenecks, yolov8n's first levels,
// TrackNetV3's 32 -> 64) with the main loop of conv_patch_h2r.hip: the weights of a k-step come global -> VGPR from the
// operand-order copy ([fragment][k-step][h | m][lane][16 B]), two steps ahead through three register sets with counted waits; the
// only barrier of the kernel is the one that publishes the patch.  The wide kernel wrapped each of its 5-14 steps (24 MFMAs per wave
// with two products) in vmcnt(0) + s_barrier + three weight reads: on 48 -> 48 at 320 x 320 it ran at 257 TFLOP/s fp32-equivalent,
// MfmaUtil 0.27 (profiles/r5t_conv_h2_pmc.txt).  All four waves of a workgroup own the same channel fragments (a wave = 4 rows x 16
// pixels x NF fragments): their weight requests are the same addresses — one L1 fill serves four waves.
// Full-chunk taps prefetch their input rows like conv_patch_h2r.hip (row 4 under ky = 0, row 5 under ky = 1, the next column under
// ky = 2, eight precomputed lane addresses); the five tail steps (16-channel tail, taps paired) read their four operands at the
// step's top.  Same products in the same order per accumulator as every other h2 kernel: bitwise identical results.
//
// LDS: the wide kernel's patches without its weight stages — 43 008 B chunk patch + 22 528 B tail patch (cin = 48): 65 536 B,
// 2 workgroups per CU.
#include "h2_common.h"

namespace padel {

namespace {

constexpr int kVPW = 18, kVNPix = 18 * 18;            // 16 x 16 output pixels + halo
constexpr int kVSpans = (kVNPix + 15) / 16;            // 21 spans of 16 pixels x 64 B per chunk plane
constexpr int kVPlaneB = kVSpans * 1024;
constexpr int kVTSpans = (kVNPix + 31) / 32;           // 11 spans of 32 pixels x 32 B per tail plane
constexpr int kVTPlaneB = kVTSpans * 1024;

__device__ __forceinline__ unsigned hv_tail_off(int p, int s) { return (unsigned)(p * 32 + ((s ^ ((p >> 3) & 1)) << 4)); }

typedef int hv_i32x4 __attribute__((ext_vector_type(4)));

}  // namespace

template <int NF, bool CHUNK, bool TAIL, bool WS>
__global__ void __launch_bounds__(256, 2) conv_h2v_kernel(const ConvArgs a) {
    static_assert(CHUNK || TAIL, "cin = 32 CHUNK + 16 TAIL");
    static_assert(WS || NF <= 2, "three products: 2 fragments per wave (6 register sets of 4 VGPRs per step)");
    constexpr int MF = 4;
    constexpr int PATCH_B = (CHUNK ? 2 * kVPlaneB : 0);
    constexpr int TPATCH_B = (TAIL ? 2 * kVTPlaneB : 0);
    constexpr int NSTEPS = (CHUNK ? 9 : 0) + (TAIL ? 5 : 0);
    constexpr int NPL = WS ? 1 : 2;
    constexpr int NW = NF * NPL;                  // weight requests per step and wave
    __shared__ __attribute__((aligned(16))) float lds[(PATCH_B + TPATCH_B) / 4];
    char* const ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;

    // XCD-aware 1-D tile map: the channel tiles of one pixel patch are neighbours on one XCD
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

    const float* const in0 = a.in + (((long long)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * a.in_cs + a.in_choff;
    const i32x4 rsrcP = make_rsrc3(in0);
    const unsigned lp0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);

    // ---- the whole input, requested up front (spans round-robin over the 4 waves; conv_patch_h2w.hip)
    if constexpr (CHUNK) {
        const int p_lane = lane >> 2;
        const int p_q = (lane & 3) ^ (((lane >> 4) & 1) << 1);
        const unsigned p_piece = (unsigned)((p_q >> 1) * 64 + (p_q & 1) * 16);
#define PADEL_HV_PSPAN(S_)                                                                                        \
        if (((S_) & 3) == wave) {                                                                                 \
            const int pp_ = (S_) * 16 + p_lane;                                                                   \
            const int py_ = pp_ / kVPW, px_ = pp_ - py_ * kVPW;                                                   \
            const bool ok_ = pp_ < kVNPix && (unsigned)(y0 - 1 + py_) < (unsigned)a.H && (unsigned)(x0 - 1 + px_) < (unsigned)a.W; \
            const unsigned vo_ = ok_ ? (unsigned)((py_ * a.W + px_) * a.in_cs * 4) + p_piece : kOOR3;             \
            dma3<(S_) * 1024>(vo_, rsrcP, 0u, lp0);                                                               \
            dma3<kVPlaneB + (S_) * 1024>(vo_, rsrcP, 32u, lp0);                                                   \
        }
        PADEL_HV_PSPAN(0) PADEL_HV_PSPAN(1) PADEL_HV_PSPAN(2) PADEL_HV_PSPAN(3) PADEL_HV_PSPAN(4) PADEL_HV_PSPAN(5) PADEL_HV_PSPAN(6)
        PADEL_HV_PSPAN(7) PADEL_HV_PSPAN(8) PADEL_HV_PSPAN(9) PADEL_HV_PSPAN(10) PADEL_HV_PSPAN(11) PADEL_HV_PSPAN(12) PADEL_HV_PSPAN(13)
        PADEL_HV_PSPAN(14) PADEL_HV_PSPAN(15) PADEL_HV_PSPAN(16) PADEL_HV_PSPAN(17) PADEL_HV_PSPAN(18) PADEL_HV_PSPAN(19) PADEL_HV_PSPAN(20)
#undef PADEL_HV_PSPAN
    }
    if constexpr (TAIL) {
        const int t_lane = lane >> 1;
        const unsigned t_piece = (unsigned)(((lane & 1) ^ ((lane >> 4) & 1)) * 16);
        const unsigned t_so = CHUNK ? 128u : 0u;
#define PADEL_HV_TSPAN(S_)                                                                                        \
        if (((S_) & 3) == ((wave + 1) & 3)) {                                                                     \
            const int pp_ = (S_) * 32 + t_lane;                                                                   \
            const int py_ = pp_ / kVPW, px_ = pp_ - py_ * kVPW;                                                   \
            const bool ok_ = pp_ < kVNPix && (unsigned)(y0 - 1 + py_) < (unsigned)a.H && (unsigned)(x0 - 1 + px_) < (unsigned)a.W; \
            const unsigned vo_ = ok_ ? (unsigned)((py_ * a.W + px_) * a.in_cs * 4) + t_piece : kOOR3;             \
            dma3<PATCH_B + (S_) * 1024>(vo_, rsrcP, t_so, lp0);                                                   \
            dma3<PATCH_B + kVTPlaneB + (S_) * 1024>(vo_, rsrcP, t_so + 32u, lp0);                                 \
        }
        PADEL_HV_TSPAN(0) PADEL_HV_TSPAN(1) PADEL_HV_TSPAN(2) PADEL_HV_TSPAN(3) PADEL_HV_TSPAN(4) PADEL_HV_TSPAN(5)
        PADEL_HV_TSPAN(6) PADEL_HV_TSPAN(7) PADEL_HV_TSPAN(8) PADEL_HV_TSPAN(9) PADEL_HV_TSPAN(10)
#undef PADEL_HV_TSPAN
    }

    // ---- weights: a.wr = [fragment][k-step][h | m][lane][16 bytes] (conv_patch_h2r.hip), NSTEPS k-steps per fragment
    constexpr unsigned fragb = (unsigned)NSTEPS * 2048u;
    const unsigned voffW = (unsigned)lane * 16u;
    i32x4 rsrcW[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int frag = min(f0 + j, a.n16 - 1);     // fragments beyond the matrix: any valid rows (never stored)
        rsrcW[j] = make_rsrc3(reinterpret_cast<const char*>(a.wr) + (long long)frag * fragb);
    }
    hv_i32x4 w[3][NF], wm[3][WS ? 1 : NF];
#define PADEL_HV_LOADW(ST_)                                                                                       \
    do {                                                                                                          \
        const unsigned so_ = (unsigned)((ST_) * 2048);                                                            \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) {                                                          \
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen"                                               \
                         : "=v"(w[(ST_) % 3][j]) : "v"(voffW), "s"(rsrcW[j]), "s"(so_) : "memory");               \
            if constexpr (!WS)                                                                                    \
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024"                               \
                             : "=v"(wm[(ST_) % 3][j]) : "v"(voffW), "s"(rsrcW[j]), "s"(so_) : "memory");          \
        }                                                                                                         \
    } while (0)
    // the counted wait that publishes step ST_'s set to the compiler; N_: younger requests of the wave that may stay in flight
#define PADEL_HV_WAITW(ST_, N_)                                                                                   \
    do {                                                                                                          \
        if constexpr (NF == 3)                                                                                    \
            asm volatile("s_waitcnt vmcnt(%3)" : "+v"(w[(ST_) % 3][0]), "+v"(w[(ST_) % 3][1]), "+v"(w[(ST_) % 3][2]) : "n"(N_) : "memory"); \
        else if constexpr (NF == 2 && WS)                                                                         \
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[(ST_) % 3][0]), "+v"(w[(ST_) % 3][1]) : "n"(N_) : "memory"); \
        else if constexpr (NF == 2)                                                                               \
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[(ST_) % 3][0]), "+v"(w[(ST_) % 3][1]), "+v"(wm[(ST_) % 3][0]), "+v"(wm[(ST_) % 3][WS ? 0 : NF - 1]) \
                         : "n"(N_) : "memory");                                                                   \
        else if constexpr (WS)                                                                                    \
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[(ST_) % 3][0]) : "n"(N_) : "memory");                     \
        else                                                                                                      \
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[(ST_) % 3][0]), "+v"(wm[(ST_) % 3][0]) : "n"(N_) : "memory"); \
    } while (0)

    // ---- row reads of the full chunk (conv_patch_h2r.hip): p = p0 + d, p0 = 72 wave + lr, d = 18 R + KX; 8 lane addresses by d & 7
    unsigned rbase[8];
    if constexpr (CHUNK) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int p0 = 4 * wave * kVPW + lr;
            rbase[r] = (unsigned)(p0 * 64 + ((lq ^ ((((p0 + r) >> 2) & 1) << 1)) << 4));
        }
    }
    const int rd_pix = 4 * wave * kVPW + lr;

    f32x4 acc[MF][NF], part[MF][NF], cross[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = acc[f][j]; cross[f][j] = acc[f][j]; }
    h16x8 ah[4], am[4];
#define PADEL_HV_READROW(R_, KX_)                                                                                 \
    do {                                                                                                          \
        constexpr int d_ = (R_) * kVPW + (KX_);                                                                   \
        const char* p_ = ldsb + rbase[d_ & 7];                                                                    \
        ah[(R_) & 3] = *reinterpret_cast<const h16x8*>(p_ + d_ * 64);                                             \
        am[(R_) & 3] = *reinterpret_cast<const h16x8*>(p_ + d_ * 64 + kVPlaneB);                                  \
    } while (0)
    // the products of output row F_ with the operands in slot S_ and the weights of step ST_
#define PADEL_HV_MFMA_ROW(F_, S_, ST_)                                                                            \
    do {                                                                                                          \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            cross[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, w[(ST_) % 3][j]), am[S_], cross[F_][j], 0, 0, 0); \
        if constexpr (!WS) {                                                                                      \
            _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                        \
                cross[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, wm[(ST_) % 3][WS ? 0 : j]), ah[S_], cross[F_][j], 0, 0, 0); \
        }                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            part[F_][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, w[(ST_) % 3][j]), ah[S_], part[F_][j], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
    // top of step ST_: the request of step ST_ + 2, then the counted wait for this step's weights (younger: steps ST_ + 1, ST_ + 2
    // where they exist)
#define PADEL_HV_TOP(ST_)                                                                                         \
    do {                                                                                                          \
        if constexpr ((ST_) + 2 < NSTEPS) PADEL_HV_LOADW((ST_) + 2);                                              \
        PADEL_HV_WAITW(ST_, NW * (((ST_) + 1 < NSTEPS ? 1 : 0) + ((ST_) + 2 < NSTEPS ? 1 : 0)));                  \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
    // full-chunk tap T_ = 3 kx + ky (step T_): rows prefetched as in conv_patch_h2r.hip; after tap 8 nothing is prefetched
#define PADEL_HV_STEP(T_)                                                                                         \
    do {                                                                                                          \
        constexpr int kx_ = h2_tap_kx(T_), ky_ = h2_tap_ky(T_);                                                   \
        PADEL_HV_TOP(T_);                                                                                         \
        if constexpr (ky_ == 0) {                                                                                 \
            PADEL_HV_MFMA_ROW(0, 0, T_);                                                                          \
            PADEL_HV_READROW(4, kx_);                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            PADEL_HV_MFMA_ROW(1, 1, T_); PADEL_HV_MFMA_ROW(2, 2, T_); PADEL_HV_MFMA_ROW(3, 3, T_);                \
        } else if constexpr (ky_ == 1) {                                                                          \
            PADEL_HV_MFMA_ROW(0, 1, T_);                                                                          \
            PADEL_HV_READROW(5, kx_);                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            PADEL_HV_MFMA_ROW(1, 2, T_); PADEL_HV_MFMA_ROW(2, 3, T_); PADEL_HV_MFMA_ROW(3, 0, T_);                \
        } else {                                           /* ky = 2: rows 2, 3, 0, 1 free slots 0, 1, 2, 3 for the next column */ \
            PADEL_HV_MFMA_ROW(2, 0, T_);                                                                          \
            if constexpr (kx_ < 2) PADEL_HV_READROW(0, kx_ + 1);                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            PADEL_HV_MFMA_ROW(3, 1, T_);                                                                          \
            if constexpr (kx_ < 2) PADEL_HV_READROW(1, kx_ + 1);                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            PADEL_HV_MFMA_ROW(0, 2, T_);                                                                          \
            if constexpr (kx_ < 2) PADEL_HV_READROW(2, kx_ + 1);                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            PADEL_HV_MFMA_ROW(1, 3, T_);                                                                          \
            if constexpr (kx_ < 2) PADEL_HV_READROW(3, kx_ + 1);                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        }                                                                                                         \
    } while (0)
    // tail step JT_ (step ST_ of the walk): lane group q of an operand holds the 8 channels 8 (q & 1).. of tap 2 JT_ + (q >> 1)
    // (the 10th "tap" has zero weights: any finite data, tap 8 again).  Its four operands are read at the step's top, under the wait
#define PADEL_HV_TSTEP(JT_, ST_)                                                                                  \
    do {                                                                                                          \
        constexpr int ta_ = 2 * (JT_), tb_ = 2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8;                               \
        int rp_ = rd_pix;                                                                                         \
        asm volatile("" : "+v"(rp_));                                                                             \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                          \
            const int pa_ = rp_ + (f + h2_tap_ky(ta_)) * kVPW + h2_tap_kx(ta_), pb_ = rp_ + (f + h2_tap_ky(tb_)) * kVPW + h2_tap_kx(tb_); \
            const char* p_ = ldsb + PATCH_B + hv_tail_off((lq >> 1) ? pb_ : pa_, lq & 1);                         \
            ah[f] = *reinterpret_cast<const h16x8*>(p_);                                                          \
            am[f] = *reinterpret_cast<const h16x8*>(p_ + kVTPlaneB);                                              \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_HV_TOP(ST_);                                                                                        \
        PADEL_HV_MFMA_ROW(0, 0, ST_); PADEL_HV_MFMA_ROW(1, 1, ST_); PADEL_HV_MFMA_ROW(2, 2, ST_); PADEL_HV_MFMA_ROW(3, 3, ST_); \
    } while (0)
#define PADEL_HV_FLUSH()                                                                                          \
    do {                                                                                                          \
        _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                            \
            _Pragma("unroll") for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; } \
    } while (0)

    // W(0), W(1) behind the patch requests; the patch is published by the kernel's only barrier (every wave waits for its own
    // requests: with W(0), W(1) allowed in flight)
    PADEL_HV_LOADW(0);
    if constexpr (NSTEPS > 1) PADEL_HV_LOADW(1);
    wait_vm3<NW * (NSTEPS > 1 ? 2 : 1)>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (CHUNK) {
        PADEL_HV_READROW(0, 0); PADEL_HV_READROW(1, 0); PADEL_HV_READROW(2, 0); PADEL_HV_READROW(3, 0);
        __builtin_amdgcn_sched_barrier(0);
        PADEL_HV_STEP(0); PADEL_HV_STEP(1); PADEL_HV_STEP(2); PADEL_HV_STEP(3); PADEL_HV_STEP(4);
        PADEL_HV_STEP(5); PADEL_HV_STEP(6); PADEL_HV_STEP(7); PADEL_HV_STEP(8);
        PADEL_HV_FLUSH();
    }
    if constexpr (TAIL) {
        constexpr int S0 = CHUNK ? 9 : 0;
        PADEL_HV_TSTEP(0, S0); PADEL_HV_TSTEP(1, S0 + 1); PADEL_HV_TSTEP(2, S0 + 2); PADEL_HV_TSTEP(3, S0 + 3); PADEL_HV_TSTEP(4, S0 + 4);
        PADEL_HV_FLUSH();
    }
#undef PADEL_HV_FLUSH
#undef PADEL_HV_TSTEP
#undef PADEL_HV_STEP
#undef PADEL_HV_TOP
#undef PADEL_HV_MFMA_ROW
#undef PADEL_HV_READROW
#undef PADEL_HV_WAITW
#undef PADEL_HV_LOADW

    int mpix[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int oy = y0 + 4 * wave + f, ox = x0 + lr;
        mpix[f] = (oy < a.Ho && ox < a.Wo) ? (n * a.Ho + oy) * a.Wo + ox : -1;
    }
    const bool fast = y0 + 16 <= a.Ho && x0 + 16 <= a.Wo && (f0 + NF) * 16 <= a.cout && (((a.out_choff | a.out_cs) & 3) == 0) &&
                      (!a.res || (((a.res_choff | a.res_cs) & 3) == 0));
    h2_epilogue<MF, NF>(a, acc, cross, mpix, f0, lq, fast);
}

bool conv_h2v_supported(const ConvArgs& a) {
    return a.wr && a.ksize == 3 && a.stride == 1 && (a.cin == 16 || a.cin == 32 || a.cin == 48) && a.Ho == a.H && a.Wo == a.W && a.w != nullptr && !a.in2;
}

template <int NF, bool WS>
static hipError_t launch_hv(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    const int batch = a.M / (a.Ho * a.Wo);
    a.n_mtiles = batch * ((a.Ho + 15) / 16) * ((a.Wo + 15) / 16);
    a.n_ntiles = (a.n16 + NF - 1) / NF;
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8) * (unsigned)a.n_ntiles, 1, 1);
    if (a.cin == 48) hipLaunchKernelGGL((conv_h2v_kernel<NF, true, true, WS>), grid, dim3(256), 0, s, a);
    else if (a.cin == 32) hipLaunchKernelGGL((conv_h2v_kernel<NF, true, false, WS>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_h2v_kernel<NF, false, true, WS>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// nf = channel fragments (of 16) per workgroup: 1, 2, 3 (3 with two products only)
hipError_t launch_conv_h2v(const ConvArgs& a, int nf, hipStream_t s) {
    if (!conv_h2v_supported(a)) return hipErrorNotSupported;
    if (a.w_single) {
        switch (nf) {
            case 1: return launch_hv<1, true>(a, s);
            case 2: return launch_hv<2, true>(a, s);
            case 3: return launch_hv<3, true>(a, s);
        }
    } else {
        switch (nf) {
            case 1: return launch_hv<1, false>(a, s);
            case 2: return launch_hv<2, false>(a, s);
        }
    }
    return hipErrorNotSupported;
}

}  // namespace padel
