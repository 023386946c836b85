// K5+K3 fused: model.0 (stem Conv(3, c, 3, 2) + BN + SiLU from the u8 network input) and model.1 (Conv(c, 2c, 3, 2) + BN +
// SiLU) of an h2 YOLOv8 graph in ONE kernel — the stem's output map (5 GB for the 64 x 1280^2 pose batch: written once, read
// back once, 2.9 + 3.2 ms of the bench's step for 0.15 TFLOP) never reaches HBM.  OPT-IN this round (pa_engine_set_tuning
// "fuse_stem" = 1; tests/test_gpu_h2.py::test_fused_stem_layer1_matches_unfused): same values as the two kernels it replaces —
// the stem phase is stem_mfma_kernel's arithmetic (kernels_misc.hip), the conv phase walks K like conv_h2_kernel.
//
// A workgroup owns 4 x 16 output pixels of layer 1 and all of its 2c channels.
//   phase 1: the 9 x 33 stem pixels under that tile (19 fragments of 16 positions over the 4 waves; ROUND 5: on the F16 matrix
//            pipe — the u8 input bytes ARE fp16 numbers (0..255, exact, no correction part), the stem weights divided by 255
//            are split into row-scaled fp16 pairs in the kernel's prologue, so a 16 x 16 fragment is TWO
//            v_mfma_f32_16x16x32_f16 (main wh x v, correction wm x v; K = 27 -> 32) instead of the EIGHT
//            v_mfma_f32_16x16x4_f32 (32 cycles each) of the stand-alone stem kernel: 32 instead of 256 matrix-pipe cycles
//            per fragment, 3 600 of a wave's ~15 700 busy cycles per tile.  sum w (v / 255) becomes sum (w / 255) v: the same
//            value to fp32 rounding (the stand-alone stem kernel keeps the literal order; tests/test_gpu_h2.py compares the
//            two to 2e-5 of the head maps).  Then bias, fast SiLU, pair encoding) go to LDS — zeros where layer 1 sees its padding — in COLUMN-PARITY planes:
//            entry ((row * 2 + (col & 1)) * 17 + col / 2), so that the stride-2 window of a tap (cols 2 ox + kx) is 16
//            CONSECUTIVE entries, read as MFMA operands exactly like the stride-1 patch kernels read theirs;
//   phase 2: layer 1 from those planes: waves as 2 row pairs x 2 channel halves (2 x NF fragments each), weights through the
//            2-stage LDS-DMA ring, 9 tap steps per 32-channel chunk (taps column-major) + 5 tap-pair steps for a 16-channel tail.
// c = 48 (yolov8m): the chunk planes (40 KB) and the tail planes would not fit next to the weight ring, so the tail channels
// of the stem are computed AFTER the chunk steps into the same LDS region (their u8 operands are kept in registers).
#include "h2_common.h"

namespace padel {

namespace {

constexpr int kFRows = 9, kFCols = 33, kFPos = kFRows * kFCols;     // stem positions under a 4 x 16 tile of layer 1
constexpr int kFFrags = (kFPos + 15) / 16;                          // 19
constexpr int kFPerWave = (kFFrags + 3) / 4;                        // 5
constexpr int kFEntries = 320;                                      // 9 rows x 2 parities x 17 = 306 entries, padded
constexpr int kFPlaneB = kFEntries * 64, kFTPlaneB = kFEntries * 32;

__device__ __forceinline__ unsigned fs_off(int p, int q) { return (unsigned)(p * 64 + ((q ^ (((p >> 2) & 1) << 1)) << 4)); }
__device__ __forceinline__ unsigned fs_tail_off(int p, int s) { return (unsigned)(p * 32 + ((s ^ ((p >> 3) & 1)) << 4)); }
__device__ __forceinline__ int fs_entry(int srow, int scol) { return (srow * 2 + (scol & 1)) * 17 + (scol >> 1); }

}  // namespace

// NF = c / 16 = stem channel fragments = layer-1 fragments per wave (layer 1 has 2c channels: 2 NF fragments per channel half pair)
// WS: layer 1's packed weights have an all-zero m plane (ConvArgs::w_single): no wm x ah product and no m-plane operand reads.  (The
// weight ring's requests still carry the all-zero plane — PADEL_FS_DMAB covers both planes of a k-step with one set of spans; the
// zeros land in LDS and are never read.  ADVICE r5.)
// WR (round 6, with WS): layer 1's weights come global -> VGPR from the operand-order copy (ConvArgs::wr, conv_patch_h2r.hip), two
// steps ahead through three register sets with counted waits, and its operands are read one step ahead into a second register set:
// phase 2 has NO per-step barrier and no weight stages in LDS (a step is 2 x NF x 2 = 12 MFMAs per wave at c = 48 — 192 pipe
// cycles that used to sit behind a barrier, a ring wait and NF weight reads each).  The barriers that remain publish the stem's
// planes (one, two more around the c = 48 tail planes).
template <int NF, bool WS = false, bool WR = false>
__global__ void __launch_bounds__(256, 2) stem_l1_h2_kernel(const StemArgs st, const ConvArgs a) {
    static_assert(!WR || WS, "register weights: two-product layers");
    constexpr bool CHUNK = NF >= 2, TAIL = (NF & 1) != 0;
    constexpr int NCF = CHUNK ? 2 : 0;               // stem fragments that form the 32-channel chunk
    constexpr int MF = 2;
    constexpr int BN = 2 * NF * 16;
    constexpr int BPLANE_B = BN * 64, BSTAGE_B = 2 * BPLANE_B;
    constexpr int S_B = CHUNK ? 2 * kFPlaneB : 2 * kFTPlaneB;      // the tail planes of c = 48 reuse the chunk's region
    constexpr int NSTEPS = (CHUNK ? 9 : 0) + (TAIL ? 5 : 0);
    constexpr int NSTG = 3;                          // weight ring: step T + 2 is requested while step T computes (a layer-1 step
                                                     // is 18 MFMAs per wave: with a 2-stage ring every step waited out a DMA round trip)
    static_assert(S_B + NSTG * BSTAGE_B + 1024 <= 80 * 1024, "2 workgroups per CU");
    constexpr int WSTG_B = WR ? 0 : NSTG * BSTAGE_B;               // (register weights: no weight stages)
    __shared__ __attribute__((aligned(16))) float lds[(S_B + WSTG_B + 1024) / 4];
    char* const ldsb = reinterpret_cast<char*>(lds);
    float* const s_osc = lds + (S_B + WSTG_B) / 4;                  // 1 / row scale of the stem's weight rows (NF x 16 floats)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 1, wc = wave >> 1;
    const int lr = lane & 15, lq = lane >> 4;

    // XCD-aware 1-D tile map (one channel tile)
    const int nmt = a.n_mtiles;
    const int bid = blockIdx.x;
    const int q8 = nmt >> 3, r8 = nmt & 7, xcd = bid & 7, mloc = bid >> 3;
    if (mloc >= q8 + (xcd < r8 ? 1 : 0)) return;
    const int mt = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + mloc;
    const int txN = (a.Wo + 15) >> 4, tyN = (a.Ho + 3) >> 2;
    const int tpi = tyN * txN;
    const int n = mt / tpi, rt = mt - n * tpi;
    const int ty = rt / txN, tx = rt - ty * txN;
    const int oy0 = ty * 4, ox0 = tx * 16;

    // ---- layer-1 weights: rows of NSTEPS k-steps x 128 bytes (h | m).  A stage = 2 planes x 2 NF spans of 16 rows x 64 bytes;
    // wave w requests the spans w, w + 4, w + 8 (span = plane * 2 NF + row group): lane i -> row i / 4, physical slot i & 3
    constexpr unsigned rowb = (unsigned)NSTEPS * 128u;
    const unsigned lp0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
    const int b_row = lane >> 2;
    const int b_sc = (lane & 3) ^ ((4 - ((b_row >> 2) & 3)) & 3);
    unsigned voffB[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        const int sp = wave + 4 * k;                 // < 4 NF
        const int pl = sp / (2 * NF), g = sp - pl * (2 * NF);
        const int frag = min(g, a.n16 - 1);
        voffB[k] = (unsigned)((frag * 16 + b_row) * rowb + pl * 64 + b_sc * 16);
    }
    const i32x4 rsrcB = make_rsrc3(a.w);
    const unsigned lw0 = __builtin_amdgcn_readfirstlane(lp0 + (unsigned)S_B + (unsigned)wave * 1024u);
    const unsigned lw1 = __builtin_amdgcn_readfirstlane(lw0 + (unsigned)BSTAGE_B);
    const unsigned lw2 = __builtin_amdgcn_readfirstlane(lw1 + (unsigned)BSTAGE_B);
#define PADEL_FS_DMAB(ST_)                                                                                        \
    do {                                                                                                          \
        const unsigned lw_ = ((ST_) % 3) == 0 ? lw0 : ((ST_) % 3) == 1 ? lw1 : lw2;                               \
        dma3<0>(voffB[0], rsrcB, (unsigned)(ST_) * 128u, lw_);                                                    \
        if constexpr (NF >= 2) dma3<4096>(voffB[NF >= 2 ? 1 : 0], rsrcB, (unsigned)(ST_) * 128u, lw_);            \
        if constexpr (NF >= 3) dma3<8192>(voffB[NF >= 3 ? 2 : 0], rsrcB, (unsigned)(ST_) * 128u, lw_);            \
    } while (0)
    // register weights: fragment wc NF + j of layer 1 (its 2 NF fragments, NSTEPS k-steps of 2 KB each: [h | m][lane][16 B])
    typedef int fs_i32x4 __attribute__((ext_vector_type(4)));
    const unsigned voffW = (unsigned)lane * 16u;
    i32x4 rsrcW[NF];
    fs_i32x4 wreg[3][NF];
    (void)voffW; (void)rsrcW; (void)wreg;
    if constexpr (WR) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
            rsrcW[j] = make_rsrc3(reinterpret_cast<const char*>(a.wr) + (long long)(wc * NF + j) * (NSTEPS * 2048));
    }
#define PADEL_FS_LOADW(ST_)                                                                                       \
    do {                                                                                                          \
        const unsigned so_ = (unsigned)((ST_) * 2048);                                                            \
        _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                            \
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen"                                               \
                         : "=v"(wreg[(ST_) % 3][j]) : "v"(voffW), "s"(rsrcW[j]), "s"(so_) : "memory");            \
    } while (0)
#define PADEL_FS_WAITW(ST_, N_)                                                                                   \
    do {                                                                                                          \
        if constexpr (NF == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(wreg[(ST_) % 3][0]), "+v"(wreg[(ST_) % 3][1]), "+v"(wreg[(ST_) % 3][NF - 1]) : "n"(N_) : "memory"); \
        else if constexpr (NF == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wreg[(ST_) % 3][0]), "+v"(wreg[(ST_) % 3][NF - 1]) : "n"(N_) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wreg[(ST_) % 3][0]) : "n"(N_) : "memory");                \
    } while (0)
    if constexpr (WR) {
        PADEL_FS_LOADW(0);
        if constexpr (NSTEPS > 1) PADEL_FS_LOADW(1);
    } else {
        PADEL_FS_DMAB(0);
        PADEL_FS_DMAB(1);
    }

    // ---- phase 1 operands.  K slot kk of lane group lq is k = 8 lq + kk -> tap (dy, dx) = (k / 9, k / 3 % 3), colour byte k % 3
    // (k >= 27: zero weights AND zero data).  Weights of row lr = channel 16 j + lr: w / 255, scaled by the power of two that
    // puts the row's largest magnitude into [2^12, 2^13) (graph.py:h2_row_scale), split into fp16 pairs like the packed weights
    // of every other h2 layer; every wave computes all NF rows sets (24 values per lane), wave 0 publishes the inverse scales.
    h16x8 wfh[NF], wfm[NF];
    f32x4 bias4[NF], osc4[NF];
    int kdy[8], kdx[8], ksh[8];
    bool kval[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int k = 8 * lq + kk;
        kval[kk] = k < 27;
        const int t = k / 3;
        ksh[kk] = 8 * (k - 3 * t);
        kdy[kk] = t / 3;
        kdx[kk] = t - 3 * kdy[kk];
    }
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        float wv[8], mx = 0.0f;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            wv[kk] = kval[kk] ? st.w[(j * 16 + lr) * 27 + 8 * lq + kk] / 255.0f : 0.0f;
            mx = fmaxf(mx, fabsf(wv[kk]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));              // the row's largest |w / 255| (its 27 weights sit in the 4 lanes lr + 16 q)
        const int ex = (int)((__float_as_uint(mx) >> 23) & 255u);          // biased exponent: floor(log2 mx) + 127
        const int e = ex == 0 ? 0 : min(max(139 - ex, -100), 100);         // 12 - floor(log2 mx); denormal / all-zero rows: scale 1
        const float sc = __uint_as_float((unsigned)(e + 127) << 23), isc = __uint_as_float((unsigned)(127 - e) << 23);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float x = wv[kk] * sc;
            const _Float16 hh = (_Float16)x;
            wfh[j][kk] = hh;
            wfm[j][kk] = (_Float16)((x - (float)hh) * kH2Scale);
        }
        if (wave == 0 && lq == 0) s_osc[j * 16 + lr] = isc;
        bias4[j] = *reinterpret_cast<const f32x4*>(st.bias + j * 16 + lq * 4);
    }
    __syncthreads();                                 // the inverse scales
#pragma unroll
    for (int j = 0; j < NF; ++j) osc4[j] = *reinterpret_cast<const f32x4*>(s_osc + j * 16 + lq * 4);
    const uint32_t* const img = reinterpret_cast<const uint32_t*>(st.in) + (long long)n * st.H * st.W;
    h16x8 apx[kFPerWave];                            // per owned position fragment: the 8 K slots of this lane as fp16 (exact bytes)
    int s_ent[kFPerWave];                            // LDS entry of the fragment's position of this lane, -1: no such position
    bool s_in[kFPerWave];                            // the position lies inside the stem map (else layer 1 sees its zero padding)
#pragma unroll
    for (int i = 0; i < kFPerWave; ++i) {
        const int p = (wave + 4 * i) * 16 + lr;
        const bool pv = (wave + 4 * i) < kFFrags && p < kFPos;
        const int pc = pv ? p : 0;
        const int srow = pc / kFCols, scol = pc - srow * kFCols;
        const int sy = 2 * oy0 - 1 + srow, sx = 2 * ox0 - 1 + scol;
        s_ent[i] = pv ? fs_entry(srow, scol) : -1;
        s_in[i] = pv && (unsigned)sy < (unsigned)st.Ho && (unsigned)sx < (unsigned)st.Wo;
        uint32_t pxw[8];
        bool okk[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {             // UNCONDITIONAL loads from clamped coordinates, masked afterwards
            const int iy = sy * 2 - 1 + kdy[kk], ix = sx * 2 - 1 + kdx[kk];
            okk[kk] = s_in[i] && kval[kk] && (unsigned)iy < (unsigned)st.H && (unsigned)ix < (unsigned)st.W;
            const int iyc = min(max(iy, 0), st.H - 1), ixc = min(max(ix, 0), st.W - 1);
#if defined(PADEL_STEM_PROBE) && (PADEL_STEM_PROBE == 2 || PADEL_STEM_PROBE == 5)
            pxw[kk] = (uint32_t)(iyc * 3 + ixc * 7);      // probe: no input gathers
#else
            pxw[kk] = img[(long long)iyc * st.W + ixc];
#endif
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int v = okk[kk] ? (int)((pxw[kk] >> ksh[kk]) & 255u) : 0;
            apx[i][kk] = (_Float16)v;
        }
    }
    bool bad = false;
#if defined(PADEL_STEM_PROBE) && (PADEL_STEM_PROBE == 1 || PADEL_STEM_PROBE == 5)
#define PADEL_STEM_ACT(x_) (x_)                      // probe: no SiLU in the stem phase (wrong results)
#else
#define PADEL_STEM_ACT(x_) h2_act<ACT_SILU>(x_)
#endif
    // stem fragments [J0_, J0_ + NJ_) of every owned position -> pairs in LDS.  TL_: they are the 16-channel tail group
    // (32 bytes per entry and plane), else halves 0 / 1 of the 32-channel chunk (64 bytes per entry and plane)
#define PADEL_FS_STEM(J0_, NJ_, TL_)                                                                              \
    do {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < kFPerWave; ++i) {                                                   \
            f32x4 smain[NJ_], scross[NJ_];                                                                        \
            _Pragma("unroll") for (int j = 0; j < (NJ_); ++j) {                                                   \
                scross[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfm[(J0_) + j], apx[i], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0); \
                smain[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfh[(J0_) + j], apx[i], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);  \
            }                                                                                                     \
            if (s_ent[i] >= 0) {                                                                                  \
                _Pragma("unroll") for (int j = 0; j < (NJ_); ++j) {                                               \
                    f32x4 v;                                                                                      \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                               \
                        const float x = PADEL_STEM_ACT(fmaf(fmaf(scross[j][r], kH2InvScale, smain[j][r]), osc4[(J0_) + j][r], bias4[(J0_) + j][r])); \
                        v[r] = s_in[i] ? x : 0.0f;                                                                \
                    }                                                                                             \
                    h16x4 hv, mv;                                                                                 \
                    h2_encode4(v, hv, mv, bad);                                                                   \
                    char* op;                                                                                     \
                    if (TL_) op = ldsb + fs_tail_off(s_ent[i], lq >> 1) + (lq & 1) * 8;                           \
                    else op = ldsb + fs_off(s_ent[i], 2 * j + (lq >> 1)) + (lq & 1) * 8;                          \
                    *reinterpret_cast<h16x4*>(op) = hv;                                                           \
                    *reinterpret_cast<h16x4*>(op + ((TL_) ? kFTPlaneB : kFPlaneB)) = mv;                          \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)

    // ---- phase 2 state
    const int ld_off = (wc * NF) * 256 + lr * 16 + ((lq ^ ((4 - ((lr >> 2) & 3)) & 3)) << 2);      // floats
    const float* const b_rd0 = lds + S_B / 4 + ld_off;
    const float* const b_rd1 = b_rd0 + BSTAGE_B / 4;
    const float* const b_rd2 = b_rd1 + BSTAGE_B / 4;
    f32x4 acc[MF][NF], part[MF][NF], cross[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < NF; ++j) { acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; part[f][j] = acc[f][j]; cross[f][j] = acc[f][j]; }
    h16x8 ah[MF], am[MF], wh[NF], wm[NF];
    h16x8 ah2[2][MF], am2[2][MF];                  // (WR: the operands of step T in set T & 1, read one step ahead)
    (void)ah2; (void)am2;
    // entry of output row (2 wr + F_) of this wave at tap T_ (column-major: (ky, kx) = (T_ % 3, T_ / 3)), column lr
#define PADEL_FS_ENT(F_, T_) fs_entry(2 * (2 * wr + (F_)) + h2_tap_ky(T_), 2 * lr + h2_tap_kx(T_))
#define PADEL_FS_READB(ST_)                                                                                       \
    do {                                                                                                          \
        const float* const br_ = ((ST_) % 3) == 0 ? b_rd0 : ((ST_) % 3) == 1 ? b_rd1 : b_rd2;                     \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) {                                                          \
            wh[j] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(br_ + j * 256));                    \
            if constexpr (!WS) wm[j] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(br_ + BPLANE_B / 4 + j * 256)); \
        }                                                                                                         \
    } while (0)
#define PADEL_FS_MFMA()                                                                                           \
    do {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            cross[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], am[f], cross[f][j], 0, 0, 0);             \
        if constexpr (!WS) {                                                                                      \
            _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)         \
                cross[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[j], ah[f], cross[f][j], 0, 0, 0);         \
        }                                                                                                         \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            part[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], ah[f], part[f][j], 0, 0, 0);               \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)
    // step ST_ of the K walk: this wave's requests of the step (issued TWO steps earlier; vmcnt is in-order: the NF requests
    // of step ST_ + 1 may stay in flight) have landed; barrier = they have for every wave, and the stage read in step ST_ - 1
    // is free for the request of step ST_ + 2
#define PADEL_FS_SYNC(ST_)                                                                                        \
    do {                                                                                                          \
        if constexpr ((ST_) + 1 < NSTEPS) wait_vm3<NF>(); else wait_vm3<0>();                                     \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
        PADEL_FS_READB(ST_);                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if constexpr ((ST_) + 2 < NSTEPS) PADEL_FS_DMAB((ST_) + 2);                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#define PADEL_FS_READA(T_)                                                                                        \
    do {                                                                                                          \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                          \
            const char* p_ = ldsb + fs_off(PADEL_FS_ENT(f, T_), lq);                                              \
            ah[f] = *reinterpret_cast<const h16x8*>(p_);                                                          \
            am[f] = *reinterpret_cast<const h16x8*>(p_ + kFPlaneB);                                               \
        }                                                                                                         \
    } while (0)
    // (the planes are static after their first barrier: from the second step on the operands are read under the wait)
#define PADEL_FS_STEP(T_)                                                                                         \
    do {                                                                                                          \
        if constexpr ((T_) > 0) PADEL_FS_READA(T_);                                                               \
        PADEL_FS_SYNC(T_);                                                                                        \
        if constexpr ((T_) == 0) PADEL_FS_READA(T_);       /* the barrier of the first step publishes the planes */   \
        PADEL_FS_MFMA();                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
    // tail step JT_ (step ST_ of the walk): lane group q of an operand holds the 8 channels 8 (q & 1).. of tap 2 JT_ + (q >> 1)
#define PADEL_FS_TREADA(JT_)                                                                                      \
    do {                                                                                                          \
        constexpr int ta_ = 2 * (JT_), tb_ = 2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8;                               \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                          \
            const int ea_ = PADEL_FS_ENT(f, ta_), eb_ = PADEL_FS_ENT(f, tb_);                                     \
            const char* p_ = ldsb + fs_tail_off((lq >> 1) ? eb_ : ea_, lq & 1);                                   \
            ah[f] = *reinterpret_cast<const h16x8*>(p_);                                                          \
            am[f] = *reinterpret_cast<const h16x8*>(p_ + kFTPlaneB);                                              \
        }                                                                                                         \
    } while (0)
#define PADEL_FS_TSTEP(JT_, ST_)                                                                                  \
    do {                                                                                                          \
        if constexpr ((JT_) > 0) PADEL_FS_TREADA(JT_);                                                            \
        PADEL_FS_SYNC(ST_);                                                                                       \
        if constexpr ((JT_) == 0) PADEL_FS_TREADA(JT_);    /* the barrier of the first tail step publishes the tail planes */ \
        PADEL_FS_MFMA();                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#define PADEL_FS_FLUSH()                                                                                          \
    do {                                                                                                          \
        _Pragma("unroll") for (int f = 0; f < MF; ++f)                                                            \
            _Pragma("unroll") for (int j = 0; j < NF; ++j) { acc[f][j] += part[f][j]; part[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; } \
    } while (0)

    // ---- register-weight steps (WR)
#define PADEL_FS_READA2(SET_, T_)                                                                                 \
    do {                                                                                                          \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                          \
            const char* p_ = ldsb + fs_off(PADEL_FS_ENT(f, T_), lq);                                              \
            ah2[SET_][f] = *reinterpret_cast<const h16x8*>(p_);                                                   \
            am2[SET_][f] = *reinterpret_cast<const h16x8*>(p_ + kFPlaneB);                                        \
        }                                                                                                         \
    } while (0)
#define PADEL_FS_TREADA2(SET_, JT_)                                                                               \
    do {                                                                                                          \
        constexpr int ta_ = 2 * (JT_), tb_ = 2 * (JT_) + 1 < 9 ? 2 * (JT_) + 1 : 8;                               \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) {                                                          \
            const int ea_ = PADEL_FS_ENT(f, ta_), eb_ = PADEL_FS_ENT(f, tb_);                                     \
            const char* p_ = ldsb + fs_tail_off((lq >> 1) ? eb_ : ea_, lq & 1);                                   \
            ah2[SET_][f] = *reinterpret_cast<const h16x8*>(p_);                                                   \
            am2[SET_][f] = *reinterpret_cast<const h16x8*>(p_ + kFTPlaneB);                                       \
        }                                                                                                         \
    } while (0)
#define PADEL_FS_MFMA2(SET_, ST_)                                                                                 \
    do {                                                                                                          \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            cross[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, wreg[(ST_) % 3][j]), am2[SET_][f], cross[f][j], 0, 0, 0); \
        _Pragma("unroll") for (int f = 0; f < MF; ++f) _Pragma("unroll") for (int j = 0; j < NF; ++j)             \
            part[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, wreg[(ST_) % 3][j]), ah2[SET_][f], part[f][j], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
    // step ST_ of the walk: request step ST_ + 2's weights, read the NEXT step's operands (NEXT_: the read macro call, or nothing at
    // the end of a plane set), wait for this step's weights (younger: ST_ + 1, ST_ + 2 where they exist), multiply
#define PADEL_FS_STEP2(ST_, NEXT_)                                                                                \
    do {                                                                                                          \
        if constexpr ((ST_) + 2 < NSTEPS) PADEL_FS_LOADW((ST_) + 2);                                              \
        NEXT_;                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_FS_WAITW(ST_, NF * (((ST_) + 1 < NSTEPS ? 1 : 0) + ((ST_) + 2 < NSTEPS ? 1 : 0)));                  \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PADEL_FS_MFMA2((ST_) & 1, ST_);                                                                           \
    } while (0)
    if constexpr (WR) {
        if constexpr (CHUNK) {
            PADEL_FS_STEM(0, 2, false);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's plane writes have reached the LDS
            __builtin_amdgcn_s_barrier();                        // ... and every wave's: the planes are published
            asm volatile("" ::: "memory");
            PADEL_FS_READA2(0, 0);
            PADEL_FS_STEP2(0, PADEL_FS_READA2(1, 1)); PADEL_FS_STEP2(1, PADEL_FS_READA2(0, 2)); PADEL_FS_STEP2(2, PADEL_FS_READA2(1, 3));
            PADEL_FS_STEP2(3, PADEL_FS_READA2(0, 4)); PADEL_FS_STEP2(4, PADEL_FS_READA2(1, 5)); PADEL_FS_STEP2(5, PADEL_FS_READA2(0, 6));
            PADEL_FS_STEP2(6, PADEL_FS_READA2(1, 7)); PADEL_FS_STEP2(7, PADEL_FS_READA2(0, 8)); PADEL_FS_STEP2(8, (void)0);
            PADEL_FS_FLUSH();
        }
        if constexpr (TAIL) {
            constexpr int S0 = CHUNK ? 9 : 0;
            if constexpr (CHUNK) {                   // every wave is done with the chunk planes: the tail planes take their place
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            PADEL_FS_STEM(NCF, 1, true);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            PADEL_FS_TREADA2(S0 & 1, 0);
            PADEL_FS_STEP2(S0, PADEL_FS_TREADA2((S0 + 1) & 1, 1)); PADEL_FS_STEP2(S0 + 1, PADEL_FS_TREADA2((S0 + 2) & 1, 2));
            PADEL_FS_STEP2(S0 + 2, PADEL_FS_TREADA2((S0 + 3) & 1, 3)); PADEL_FS_STEP2(S0 + 3, PADEL_FS_TREADA2((S0 + 4) & 1, 4));
            PADEL_FS_STEP2(S0 + 4, (void)0);
            PADEL_FS_FLUSH();
        }
    } else {
    if constexpr (CHUNK) {
        PADEL_FS_STEM(0, 2, false);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's plane writes have reached the LDS
        // (the barrier of step 0 publishes the planes)
        PADEL_FS_STEP(0); PADEL_FS_STEP(1); PADEL_FS_STEP(2); PADEL_FS_STEP(3); PADEL_FS_STEP(4);
        PADEL_FS_STEP(5); PADEL_FS_STEP(6); PADEL_FS_STEP(7); PADEL_FS_STEP(8);
        PADEL_FS_FLUSH();
    }
    if constexpr (TAIL) {
        if constexpr (CHUNK) {                       // every wave is done with the chunk planes: the tail planes take their place
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        PADEL_FS_STEM(NCF, 1, true);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int S0 = CHUNK ? 9 : 0;
        PADEL_FS_TSTEP(0, S0); PADEL_FS_TSTEP(1, S0 + 1); PADEL_FS_TSTEP(2, S0 + 2); PADEL_FS_TSTEP(3, S0 + 3); PADEL_FS_TSTEP(4, S0 + 4);
        PADEL_FS_FLUSH();
    }
    }
    wait_vm3<0>();
#undef PADEL_FS_STEP2
#undef PADEL_FS_MFMA2
#undef PADEL_FS_TREADA2
#undef PADEL_FS_READA2
#undef PADEL_FS_WAITW
#undef PADEL_FS_LOADW
#undef PADEL_FS_FLUSH
#undef PADEL_FS_TSTEP
#undef PADEL_FS_TREADA
#undef PADEL_FS_STEP
#undef PADEL_FS_READA
#undef PADEL_FS_SYNC
#undef PADEL_FS_MFMA
#undef PADEL_FS_READB
#undef PADEL_FS_ENT
#undef PADEL_FS_STEM
#undef PADEL_FS_DMAB
    h2_raise(st.ovf_flag, bad);

    int mpix[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int oy = oy0 + 2 * wr + f, ox = ox0 + lr;
        mpix[f] = (oy < a.Ho && ox < a.Wo) ? (n * a.Ho + oy) * a.Wo + ox : -1;
    }
    const int fw = NF * wc;
    const bool fast = oy0 + 4 <= a.Ho && ox0 + 16 <= a.Wo && (fw + NF) * 16 <= a.cout && (((a.out_choff | a.out_cs) & 3) == 0) &&
                      (!a.res || (((a.res_choff | a.res_cs) & 3) == 0));
#if defined(PADEL_STEM_PROBE) && (PADEL_STEM_PROBE == 4 || PADEL_STEM_PROBE == 5)
    ConvArgs a_lin = a;                              // probe: layer 1's epilogue without its SiLU (wrong results)
    a_lin.act = ACT_NONE;
    h2_epilogue<MF, NF>(a_lin, acc, cross, mpix, fw, lq, fast);
#else
    h2_epilogue<MF, NF>(a, acc, cross, mpix, fw, lq, fast);
#endif
}

// stem (h2 output, c = 16 / 32 / 48 channels) followed by a 3x3 stride-2 conv over exactly those channels with 2c outputs
bool stem_l1_h2_supported(const StemArgs& st, const ConvArgs& a) {
    const int c = st.cout;
    return st.out_f16 == 2 && (c == 16 || c == 32 || c == 48) && a.ksize == 3 && a.stride == 2 && a.cin == c && a.n16 * 16 == 2 * c &&
           a.cout == 2 * c && a.H == st.Ho && a.W == st.Wo && a.Ho == (st.Ho + 1) / 2 && a.Wo == (st.Wo + 1) / 2 && a.w && a.oscale &&
           a.ovf_flag && !a.in2 && !a.res && (st.Ho & 1) == 0 && (st.Wo & 1) == 0;
}

hipError_t launch_stem_l1_h2(const StemArgs& st, const ConvArgs& a_in, hipStream_t s) {
    if (!stem_l1_h2_supported(st, a_in)) return hipErrorNotSupported;
    ConvArgs a = a_in;
    const int batch = a.M / (a.Ho * a.Wo);
    a.n_mtiles = batch * ((a.Ho + 3) / 4) * ((a.Wo + 15) / 16);
    a.n_ntiles = 1;
    dim3 grid(8u * (unsigned)((a.n_mtiles + 7) / 8), 1, 1);
    if (a.w_single && a.wr && !(a.tune & 8)) {           // register weights (tuning bit 3: the round-5 weight ring)
        switch (st.cout / 16) {
            case 1: hipLaunchKernelGGL((stem_l1_h2_kernel<1, true, true>), grid, dim3(256), 0, s, st, a); return hipGetLastError();
            case 2: hipLaunchKernelGGL((stem_l1_h2_kernel<2, true, true>), grid, dim3(256), 0, s, st, a); return hipGetLastError();
            // (3 workgroups per CU — the LDS would allow them now — need 168 VGPRs: 11 spilled, 3.42 instead of 3.32 ms on the pose graph)
            case 3: hipLaunchKernelGGL((stem_l1_h2_kernel<3, true, true>), grid, dim3(256), 0, s, st, a); return hipGetLastError();
            default: return hipErrorNotSupported;
        }
    }
    switch (st.cout / 16) {
        case 1: if (a.w_single) hipLaunchKernelGGL((stem_l1_h2_kernel<1, true>), grid, dim3(256), 0, s, st, a);
                else hipLaunchKernelGGL((stem_l1_h2_kernel<1>), grid, dim3(256), 0, s, st, a);
                break;
        case 2: if (a.w_single) hipLaunchKernelGGL((stem_l1_h2_kernel<2, true>), grid, dim3(256), 0, s, st, a);
                else hipLaunchKernelGGL((stem_l1_h2_kernel<2>), grid, dim3(256), 0, s, st, a);
                break;
        case 3: if (a.w_single) hipLaunchKernelGGL((stem_l1_h2_kernel<3, true>), grid, dim3(256), 0, s, st, a);
                else hipLaunchKernelGGL((stem_l1_h2_kernel<3>), grid, dim3(256), 0, s, st, a);
                break;
        default: return hipErrorNotSupported;
    }
    return hipGetLastError();
}

}  // namespace padel
