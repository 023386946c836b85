// K1/K2/K5/K6/K7 — HBM-bound helper kernels of the trackers' hot path (gfx950).
//   letterbox_kernel      : ultralytics LetterBox (cv2 INTER_LINEAR u8 + pad 114) -> u8 NHWC4 net input
//                           (players_tracker.py:346-359 via [upstream] preprocess)
//   resample_pass_kernel  : one separable pass of Pillow's 8-bit bicubic resize
//                           (players_keypoints_tracker.py:264-266, ball_tracker/iterable.py:80,188)
//   stem_kernel           : model.0 Conv(3,c,3,2)+BN+SiLU straight from the u8 net input (K=27 is
//                           MFMA-unfriendly and <1.5% of the FLOPs: plain VALU FMA)
//   pool5_kernel          : MaxPool2d(5,1,2) on a channel slice (SPPF, chained 3x like upstream)
//   upsample2x / maxpool2 : nearest x2 into a concat slice; MaxPool2d(2,2) (TrackNet models.py:60-64)
#include "h2_common.h"
#include <algorithm>

namespace padel {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// 16-byte vectors of the two activation types: 4 floats or 8 halves (pools / upsample work on whole vectors)
template <typename V> struct VecT;
template <> struct VecT<f32x4> { static constexpr int N = 4; using E = float; };
template <> struct VecT<f16x8> { static constexpr int N = 8; using E = _Float16; };
template <typename V> __device__ __forceinline__ V vmax(V a, V b) {
    V r;
#pragma unroll
    for (int i = 0; i < VecT<V>::N; ++i) r[i] = a[i] > b[i] ? a[i] : b[i];
    return r;
}

// ------------------------------------------------------------------------------ letterbox
__global__ void __launch_bounds__(256) letterbox_kernel(const LetterboxArgs a) {
    const long long total = (long long)a.B * a.nh * a.nw;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % a.nw);
        const long long t = i / a.nw;
        const int y = (int)(t % a.nh);
        const int b = (int)(t / a.nh);
        int c0 = 114, c1 = 114, c2 = 114;
        const int ry = y - a.top, rx = x - a.left;
        if (ry >= 0 && ry < a.rh && rx >= 0 && rx < a.rw) {
            const uint8_t* img = a.src + (long long)b * a.h0 * a.w0 * 3;
            if (a.mode == 0) {
                const uint8_t* p = img + ((long long)ry * a.w0 + rx) * 3;
                c0 = p[0]; c1 = p[1]; c2 = p[2];
            } else if (a.mode == 1) {           // OpenCV area-fast path for exact 2x decimation
                const uint8_t* p = img + ((long long)(2 * ry) * a.w0 + 2 * rx) * 3;
                const uint8_t* q = p + (long long)a.w0 * 3;
                c0 = (p[0] + p[3] + q[0] + q[3] + 2) >> 2;
                c1 = (p[1] + p[4] + q[1] + q[4] + 2) >> 2;
                c2 = (p[2] + p[5] + q[2] + q[5] + 2) >> 2;
            } else {                            // 11-bit fixed-point bilinear (INTER_RESIZE_COEF_BITS)
                const int sx = a.xtab[rx * 3], xa0 = a.xtab[rx * 3 + 1], xa1 = a.xtab[rx * 3 + 2];
                const int sy = a.ytab[ry * 3], yb0 = a.ytab[ry * 3 + 1], yb1 = a.ytab[ry * 3 + 2];
                const int sx1 = min(sx + 1, a.w0 - 1), sy1 = min(sy + 1, a.h0 - 1);
                const uint8_t* r0 = img + (long long)sy * a.w0 * 3;
                const uint8_t* r1 = img + (long long)sy1 * a.w0 * 3;
                int o[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int h0 = r0[sx * 3 + c] * xa0 + r0[sx1 * 3 + c] * xa1;
                    const int h1 = r1[sx * 3 + c] * xa0 + r1[sx1 * 3 + c] * xa1;
                    int v = (((yb0 * (h0 >> 4)) >> 16) + ((yb1 * (h1 >> 4)) >> 16) + 2) >> 2;
                    o[c] = min(max(v, 0), 255);
                }
                c0 = o[0]; c1 = o[1]; c2 = o[2];
            }
        }
        uchar4 o4;
        if (a.reverse) { o4.x = (uint8_t)c2; o4.y = (uint8_t)c1; o4.z = (uint8_t)c0; }
        else           { o4.x = (uint8_t)c0; o4.y = (uint8_t)c1; o4.z = (uint8_t)c2; }
        o4.w = 0;
        reinterpret_cast<uchar4*>(a.dst)[i] = o4;
    }
}

hipError_t launch_letterbox(const LetterboxArgs& a, hipStream_t s) {
    const long long total = (long long)a.B * a.nh * a.nw;
    const int grid = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(letterbox_kernel, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ Pillow resample
__global__ void __launch_bounds__(256) resample_pass_kernel(const ResamplePassArgs a) {
    const long long total = (long long)a.B * a.out_h * a.out_w;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % a.out_w);
        const long long t = i / a.out_w;
        const int y = (int)(t % a.out_h);
        const int b = (int)(t / a.out_h);
        const int o = a.vertical ? y : x;
        const int lo = a.bounds[o * 2], n = a.bounds[o * 2 + 1];
        const int32_t* k = a.coefs + (long long)o * a.ksize;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
        const uint8_t* img = a.in + (long long)b * a.in_h * a.in_w * a.in_c;
        for (int j = 0; j < n; ++j) {
            const uint8_t* p = a.vertical ? img + ((long long)(lo + j) * a.in_w + x) * a.in_c
                                          : img + ((long long)y * a.in_w + lo + j) * a.in_c;
            const int kk = k[j];
            s0 += p[0] * kk; s1 += p[1] * kk; s2 += p[2] * kk;
        }
        int v0 = min(max(s0 >> 22, 0), 255), v1 = min(max(s1 >> 22, 0), 255), v2 = min(max(s2 >> 22, 0), 255);
        uint8_t* q = a.out + i * a.out_c;
        if (a.reverse) { q[0] = (uint8_t)v2; q[1] = (uint8_t)v1; q[2] = (uint8_t)v0; }
        else           { q[0] = (uint8_t)v0; q[1] = (uint8_t)v1; q[2] = (uint8_t)v2; }
        if (a.out_c == 4) q[3] = 0;
    }
}

// The vertical pass from 3-byte to 4-byte pixels (the pose tracker's 720p -> 1280 x 1280 stretch is exactly this pass: the width
// stays 1280) with FOUR output pixels per thread: an input row contributes 12 bytes = 3 aligned dwords per thread instead of
// 12 byte loads, the 4 output pixels leave as one 16-byte store instead of 16 byte stores.  Same integer arithmetic per
// channel (22-bit fixed point, accumulator from 1 << 21, >> 22, clip): byte-exact the generic kernel / Pillow.  Round 5: the
// generic kernel moved this pass at 0.9 TB/s (0.68 ms per 64-frame pose batch).  Needs in_w % 4 == 0 and 4-byte aligned rows.
__global__ void __launch_bounds__(256) resample_vpass4_kernel(const ResamplePassArgs a) {
    const int qw = a.out_w >> 2;                                // groups of 4 pixels per row
    const long long total = (long long)a.B * a.out_h * qw;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xq = (int)(i % qw);
        const long long t = i / qw;
        const int y = (int)(t % a.out_h);
        const int b = (int)(t / a.out_h);
        const int lo = a.bounds[y * 2], n = a.bounds[y * 2 + 1];
        const int32_t* k = a.coefs + (long long)y * a.ksize;
        int acc[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) acc[c] = 1 << 21;
        const uint32_t* col = reinterpret_cast<const uint32_t*>(a.in + ((long long)b * a.in_h * a.in_w + xq * 4) * 3);
        const long long rowd = (long long)a.in_w * 3 / 4;      // dwords per input row
        for (int j = 0; j < n; ++j) {
            const uint32_t* p = col + (long long)(lo + j) * rowd;
            const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
            const int kk = k[j];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[c] += (int)((d0 >> (8 * c)) & 255u) * kk;
                acc[4 + c] += (int)((d1 >> (8 * c)) & 255u) * kk;
                acc[8 + c] += (int)((d2 >> (8 * c)) & 255u) * kk;
            }
        }
        uint32_t o[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const unsigned v0 = (unsigned)min(max(acc[3 * px] >> 22, 0), 255), v1 = (unsigned)min(max(acc[3 * px + 1] >> 22, 0), 255),
                           v2 = (unsigned)min(max(acc[3 * px + 2] >> 22, 0), 255);
            o[px] = a.reverse ? (v2 | (v1 << 8) | (v0 << 16)) : (v0 | (v1 << 8) | (v2 << 16));
        }
        uint4 o4 = {o[0], o[1], o[2], o[3]};
        *reinterpret_cast<uint4*>(a.out + (((long long)b * a.out_h + y) * a.out_w + xq * 4) * 4) = o4;
    }
}

hipError_t launch_resample_pass(const ResamplePassArgs& a, hipStream_t s) {
    if (a.vertical && a.in_c == 3 && a.out_c == 4 && (a.in_w & 3) == 0 && a.out_w == a.in_w &&
        (reinterpret_cast<uintptr_t>(a.in) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 && (((long long)a.in_h * a.in_w * 3) & 3) == 0) {
        const long long total4 = (long long)a.B * a.out_h * (a.out_w >> 2);
        const int grid4 = (int)((total4 + 255) / 256 > 65536 ? 65536 : (total4 + 255) / 256);
        hipLaunchKernelGGL(resample_vpass4_kernel, dim3(grid4), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    const long long total = (long long)a.B * a.out_h * a.out_w;
    const int grid = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(resample_pass_kernel, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ stem conv
// model.0: Conv(3, c, 3, 2) + BN + SiLU straight from the u8 NHWC4 network input, on the matrix pipe: an implicit
// GEMM with K = 27 (ky, kx, c) padded to 32 = 8 x v_mfma_f32_16x16x4_f32 per 16 pixels x 16 channels.  One wave owns
// 16 output pixels x all channels per iteration of a grid-stride loop; its weight fragments (NF x 8 floats per lane)
// and the tap decode of its 8 K-slots stay in registers across iterations.  Operands are swapped (weights = A), so a
// lane ends up with 4 consecutive channels of one pixel: one 16-byte (fp32) / 8-byte (fp16) store per fragment.
// Input values are u8 -> float(u8)/255 through a 256-entry LDS table (bit-identical to `im.float() /= 255`);
// out-of-image taps contribute exact zeros.  The layer is HBM-write-bound (48-64 output floats per 4 input bytes):
// round 1's VALU kernel needed 4.1 ms for the 64 x 1280^2 pose batch, the output alone is 1.0 ms at 5 TB/s.
// OM: output storage — 0 fp32, 1 fp16, 2 h2 pairs (h2_common.h)
template <int NF, int OM>
__global__ void __launch_bounds__(256) stem_mfma_kernel(const StemArgs a) {
    __shared__ float lut[256];
    lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    // K slot kk of this lane: k = 4*kk + lq  ->  tap (dy, dx) and colour byte
    int dy[8], dx[8], sh[8];
    bool kv[8];
    float wreg[NF][8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int k = 4 * kk + lq;
        kv[kk] = k < 27;
        const int t = k / 3;
        sh[kk] = 8 * (k - 3 * t);
        dy[kk] = t / 3;
        dx[kk] = t - 3 * dy[kk];
#pragma unroll
        for (int j = 0; j < NF; ++j) wreg[j][kk] = kv[kk] ? a.w[(j * 16 + lr) * 27 + k] : 0.0f;
    }
    f32x4 bias4[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) bias4[j] = *reinterpret_cast<const f32x4*>(a.bias + j * 16 + lq * 4);
    const int P = a.B * a.Ho * a.Wo;                 // < 2^31 (launch_stem checks)
    const int ntiles = (P + 15) / 16;
    const int HoWo = a.Ho * a.Wo;
    const uint32_t* const img0 = reinterpret_cast<const uint32_t*>(a.in);
    bool bad = false;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int p = tile * 16 + lr;
        const bool pv = p < P;
        const int pc = pv ? p : 0;
        // (n, oy, ox) without integer-division sequences: q = (umulhi(x, magic) + x) >> shift (kernels.h:fill_fastdiv)
        const int n = (int)((__umulhi((unsigned)pc, a.howo_magic) + (unsigned)pc) >> a.howo_shift);
        const int rem = pc - n * HoWo;
        const int oy = (int)((__umulhi((unsigned)rem, a.wo_magic) + (unsigned)rem) >> a.wo_shift), ox = rem - oy * a.Wo;
        const uint32_t* img = img0 + (long long)n * a.H * a.W;
        // the 8 pixel words are loaded UNCONDITIONALLY from clamped (always valid) addresses and masked afterwards: a
        // conditional load per tap compiles to 8 branches whose loads wait for each other — 8 memory round trips per tile
        // (measured: 2.7 ms for the 64 x 1280^2 pose batch at 0.4 VALU / 0.2 MFMA utilisation, profiles/r3_pmc_stem.txt)
        uint32_t pxw[8];
        bool okk[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int iy = oy * 2 - 1 + dy[kk], ix = ox * 2 - 1 + dx[kk];
            okk[kk] = pv && kv[kk] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const int iyc = min(max(iy, 0), a.H - 1), ixc = min(max(ix, 0), a.W - 1);
            pxw[kk] = img[(long long)iyc * a.W + ixc];
        }
        float av[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float v = lut[(pxw[kk] >> sh[kk]) & 255u];
            av[kk] = okk[kk] ? v : 0.0f;
        }
        f32x4 acc[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][kk], av[kk], acc[j], 0, 0, 0);
        if (pv) {
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = acc[j][r] + bias4[j][r];
                    v[r] = OM != 0 ? h2_act<ACT_SILU>(x) : x / (1.0f + expf(-x));     // fast SiLU (act_fast.h) for the h2 and fp16 models
                }
                const long long o = (long long)p * a.out_cs + a.out_choff + j * 16 + lq * 4;
                if (OM == 2) {
                    h16x4 hv, mv;
                    h2_encode4(v, hv, mv, bad);
#ifdef PADEL_STEM_PROBE            /* ceiling probe (WRONG layout): one 16-byte store per lane, 64 contiguous bytes per pixel and instruction */
                    char* op = reinterpret_cast<char*>(a.out) + (long long)p * a.out_cs * 4 + (long long)((a.out_choff >> 4) + j) * 64 + lq * 16;
                    h16x8 both = {hv[0], hv[1], hv[2], hv[3], mv[0], mv[1], mv[2], mv[3]};
                    *reinterpret_cast<h16x8*>(op) = both;
#else
                    char* op = reinterpret_cast<char*>(a.out) + (long long)p * a.out_cs * 4 + h2_chan_off(a.out_choff + j * 16 + lq * 4);
                    *reinterpret_cast<h16x4*>(op) = hv;
                    *reinterpret_cast<h16x4*>(op + 32) = mv;
#endif
                } else if (OM == 1) {
                    _Float16* oh = reinterpret_cast<_Float16*>(a.out) + o;
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    h4 hv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) hv[r] = (_Float16)__builtin_amdgcn_fmed3f(v[r], -65504.0f, 65504.0f);
                    *reinterpret_cast<h4*>(oh) = hv;
                } else {
                    *reinterpret_cast<f32x4*>(a.out + o) = v;
                }
            }
        }
    }
    if (OM == 2) h2_raise(a.ovf_flag, bad);
}

template <int NF>
static void launch_stem_nf(const StemArgs& a, unsigned grid, hipStream_t s) {
    if (a.out_f16 == 2) hipLaunchKernelGGL((stem_mfma_kernel<NF, 2>), dim3(grid), dim3(256), 0, s, a);
    else if (a.out_f16) hipLaunchKernelGGL((stem_mfma_kernel<NF, 1>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((stem_mfma_kernel<NF, 0>), dim3(grid), dim3(256), 0, s, a);
}

hipError_t launch_stem(const StemArgs& a_in, hipStream_t s) {
    StemArgs a = a_in;
    if ((long long)a.B * a.Ho * a.Wo >= (1ll << 31) - 16) return hipErrorInvalidValue;
    fill_fastdiv((unsigned)(a.Ho * a.Wo), &a.howo_magic, &a.howo_shift);
    fill_fastdiv((unsigned)a.Wo, &a.wo_magic, &a.wo_shift);
    const long long ntiles = ((long long)a.B * a.Ho * a.Wo + 15) / 16;
    const unsigned grid = (unsigned)std::min<long long>((ntiles + 3) / 4, 256 * 16);
    if ((a.out_choff | a.out_cs) & 3) return hipErrorInvalidValue;     // 4-channel vector stores
    switch (a.cout / 16) {
        case 1: launch_stem_nf<1>(a, grid, s); break;
        case 2: launch_stem_nf<2>(a, grid, s); break;
        case 3: launch_stem_nf<3>(a, grid, s); break;
        case 4: launch_stem_nf<4>(a, grid, s); break;
        case 5: launch_stem_nf<5>(a, grid, s); break;     // yolov8x: c1 = 80
        default: return hipErrorNotSupported;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ pools / upsample
template <typename V>
__global__ void __launch_bounds__(256) pool5_kernel(typename VecT<V>::E* buf, int cs, int src_off, int dst_off, int c4n,
                                                     int B, int H, int W) {
    constexpr int VN = VecT<V>::N;
    const long long total = (long long)B * H * W * c4n;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % c4n);
    long long t = i / c4n;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    // window coordinates are CLAMPED to the map instead of skipped: a clamped tap re-reads a pixel of the window, which cannot
    // change a maximum, and 25 unconditional loads issue back to back where 25 guarded ones wait for each other
    V m = *reinterpret_cast<const V*>(buf + (((long long)n * H + y) * W + x) * cs + src_off + c4 * VN);
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
        const int yy = min(max(y + dy, 0), H - 1);
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const int xx = min(max(x + dx, 0), W - 1);
            m = vmax(m, *reinterpret_cast<const V*>(buf + (((long long)n * H + yy) * W + xx) * cs + src_off + c4 * VN));
        }
    }
    *reinterpret_cast<V*>(buf + (((long long)n * H + y) * W + x) * cs + dst_off + c4 * VN) = m;
}

// ---- h2 buffers: a "unit" is 8 channels = 16 bytes of h + 16 bytes of m (32 bytes further) inside a 16-channel group.
// Max-pooling compares the VALUES (h + m / 2048) and copies the winning pair: exact, nothing is re-encoded.
//
// One ORDER for every h2 max-pool: the value h + m / 2048 in fp32 first, the bits of the pair (h << 16 | m) between pairs of
// equal value.  A total order makes the maximum of a window independent of how the window is walked, which is what lets
// sppf_h2_kernel (separable passes, three levels in LDS) reproduce three chained pool5_h2_kernel launches bit for bit.  An
// element travels as (value, packed pair): one select per field, the running maximum's value is never recomputed.
template <int N> struct H2Vec;
template <> struct H2Vec<4> { typedef float F __attribute__((ext_vector_type(4))); typedef unsigned U __attribute__((ext_vector_type(4))); };
template <> struct H2Vec<8> { typedef float F __attribute__((ext_vector_type(8))); typedef unsigned U __attribute__((ext_vector_type(8))); };
template <int N>
struct H2Elems { typename H2Vec<N>::F v; typename H2Vec<N>::U p; };
__device__ __forceinline__ const char* h2_unit_ptr(const float* buf, long long pix, int cs, int choff, int u) {
    const int c = choff + u * 8;
    return reinterpret_cast<const char*>(buf) + pix * cs * 4 + (long long)(c >> 4) * 64 + (c & 15) * 2;
}
__device__ __forceinline__ void h2_elem(_Float16 h, _Float16 m, float& v, unsigned& pk) {
    v = fmaf((float)m, kH2InvScale, (float)h);
    pk = ((unsigned)__builtin_bit_cast(unsigned short, h) << 16) | __builtin_bit_cast(unsigned short, m);
}
__device__ __forceinline__ H2Elems<8> h2_load_unit(const char* q) {
    const h16x8 h = *reinterpret_cast<const h16x8*>(q), m = *reinterpret_cast<const h16x8*>(q + 32);
    H2Elems<8> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v; unsigned pk;
        h2_elem(h[i], m[i], v, pk);
        r.v[i] = v; r.p[i] = pk;
    }
    return r;
}
__device__ __forceinline__ void h2_store_unit(char* q, const H2Elems<8>& e) {
    h16x8 h, m;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = __builtin_bit_cast(_Float16, (unsigned short)(e.p[i] >> 16));
        m[i] = __builtin_bit_cast(_Float16, (unsigned short)(e.p[i] & 0xffffu));
    }
    *reinterpret_cast<h16x8*>(q) = h;
    *reinterpret_cast<h16x8*>(q + 32) = m;
}
template <int N>
__device__ __forceinline__ H2Elems<N> h2_max(const H2Elems<N>& a, const H2Elems<N>& b) {
    const auto ta = (a.v > b.v) | ((a.v == b.v) & (a.p > b.p));          // lane-wise, all ones where a wins
    H2Elems<N> r;
    r.v = ta ? a.v : b.v;
    r.p = ta ? a.p : b.p;
    return r;
}
__global__ void __launch_bounds__(256) pool5_h2_kernel(float* buf, int cs, int src_off, int dst_off, int un, int B, int H, int W) {
    const long long total = (long long)B * H * W * un;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= total) return;
    const int u = (int)(i % un);
    long long t = i / un;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    H2Elems<8> m = h2_load_unit(h2_unit_ptr(buf, ((long long)n * H + y) * W + x, cs, src_off, u));
    // clamped window (see pool5_kernel): a re-read pixel is the same element, the maximum is the same
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
        const int yy = min(max(y + dy, 0), H - 1);
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const int xx = min(max(x + dx, 0), W - 1);
            m = h2_max(h2_load_unit(h2_unit_ptr(buf, ((long long)n * H + yy) * W + xx, cs, src_off, u)), m);
        }
    }
    h2_store_unit(const_cast<char*>(h2_unit_ptr(buf, ((long long)n * H + y) * W + x, cs, dst_off, u)), m);
}

// SPPF on h2 buffers as ONE kernel (round 4; three pool5_h2_kernel launches otherwise).  A workgroup owns the whole H x W map
// of four channels of one image: the elements live in LDS, a 5 x 5 maximum is a row pass and a column pass (5 + 5 reads
// instead of 25), the three chained levels (5, 9, 13 pixels wide) are computed back to back and each is written once.  Window
// coordinates are clamped like pool5_h2_kernel's.  LDS: 2 planes x H*W x 32 bytes (40 x 40: 100 KB).  Workgroup ids are
// mapped so that the four 4-channel units of a 16-channel group (one 64-byte line per pixel) run on the same XCD, next to
// each other in time: their 8-byte pieces of a line meet in that XCD's L2.
typedef H2Elems<4> Elem4;
template <int NT>
__global__ void __launch_bounds__(NT) sppf_h2_kernel(float* buf, int cs, int choff, int c, int B, int H, int W) {
    extern __shared__ unsigned long long sppf_lds[];
    const int HW = H * W;
    const float inv_w = 1.0f / (float)W;
    Elem4* X = reinterpret_cast<Elem4*>(sppf_lds);
    Elem4* T = X + HW;
    // id -> (XCD, slot); slot -> (group q of the XCD, unit); group index g = q * 8 + xcd over B x ceil(c / 16) groups
    const int ngrp = (c + 15) / 16, G = B * ngrp;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = (slot >> 2) * 8 + xcd, ul = slot & 3;
    if (g >= G) return;
    const int n = g / ngrp, u = (g - n * ngrp) * 4 + ul;
    if (u * 4 >= c) return;
    const int c0 = choff + u * 4;
    char* base = reinterpret_cast<char*>(buf) + (long long)n * HW * cs * 4 + (long long)(c0 >> 4) * 64 + (c0 & 15) * 2;
    for (int p = threadIdx.x; p < HW; p += NT) {
        const char* q = base + (long long)p * cs * 4;
        const h16x4 hv = *reinterpret_cast<const h16x4*>(q), mv = *reinterpret_cast<const h16x4*>(q + 32);
        Elem4 k;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v; unsigned pk;
            h2_elem(hv[i], mv[i], v, pk);
            k.v[i] = v; k.p[i] = pk;
        }
        X[p] = k;
    }
    __syncthreads();
    for (int level = 1; level <= 3; ++level) {
        for (int p = threadIdx.x; p < HW; p += NT) {
            const int y = (int)(((float)p + 0.5f) * inv_w), x = p - y * W;     // exact: p + 0.5 is at least 0.5 / W from a multiple of W
            Elem4 m = X[p];
#pragma unroll
            for (int d = -2; d <= 2; ++d)
                if (d) m = h2_max(m, X[y * W + min(max(x + d, 0), W - 1)]);
            T[p] = m;
        }
        __syncthreads();
        const int off = level * c;                                   // slice `level` of the concat buffer: c channels further
        char* ob = reinterpret_cast<char*>(buf) + (long long)n * HW * cs * 4 + (long long)((c0 + off) >> 4) * 64 + ((c0 + off) & 15) * 2;
        for (int p = threadIdx.x; p < HW; p += NT) {
            const int y = (int)(((float)p + 0.5f) * inv_w), x = p - y * W;
            Elem4 m = T[p];
#pragma unroll
            for (int d = -2; d <= 2; ++d)
                if (d) m = h2_max(m, T[min(max(y + d, 0), H - 1) * W + x]);
            X[p] = m;                                                 // (only this thread touches X[p] in this pass: T is the source)
            h16x4 hv, mv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                hv[i] = __builtin_bit_cast(_Float16, (unsigned short)(m.p[i] >> 16));
                mv[i] = __builtin_bit_cast(_Float16, (unsigned short)(m.p[i] & 0xffffu));
            }
            char* q = ob + (long long)p * cs * 4;
            *reinterpret_cast<h16x4*>(q) = hv;
            *reinterpret_cast<h16x4*>(q + 32) = mv;
        }
        __syncthreads();
    }
}
// The same for fp16 buffers (8 channels = one 16-byte vector per pixel and workgroup).  Written at the end of round 4, timed at
// the start of round 5 (profiles/r5e_sppf_f16.txt, c4 = 64 x 1080p fp16: players 0.053 -> 0.033, ball 0.029 -> 0.016, pose 0.258 ->
// 0.188 ms) and the default of fp16 graphs since (fuse_sppf = 0: the three pool5_kernel launches); same head maps and
// detections as the three launches (tests/test_gpu_fp16.py).
template <int NT>
__global__ void __launch_bounds__(NT) sppf_f16_kernel(_Float16* buf, int cs, int choff, int c, int B, int H, int W) {
    extern __shared__ unsigned long long sppf_lds[];
    const int HW = H * W;
    const float inv_w = 1.0f / (float)W;
    f16x8* X = reinterpret_cast<f16x8*>(sppf_lds);
    f16x8* T = X + HW;
    const int nu = c / 8;
    const int u = blockIdx.x % nu, n = blockIdx.x / nu;
    if (n >= B) return;
    _Float16* base = buf + (long long)n * HW * cs + choff + u * 8;
    for (int p = threadIdx.x; p < HW; p += NT) X[p] = *reinterpret_cast<const f16x8*>(base + (long long)p * cs);
    __syncthreads();
    for (int level = 1; level <= 3; ++level) {
        for (int p = threadIdx.x; p < HW; p += NT) {
            const int y = (int)(((float)p + 0.5f) * inv_w), x = p - y * W;
            f16x8 m = X[p];
#pragma unroll
            for (int d = -2; d <= 2; ++d)
                if (d) m = vmax(m, X[y * W + min(max(x + d, 0), W - 1)]);
            T[p] = m;
        }
        __syncthreads();
        _Float16* ob = base + level * c;
        for (int p = threadIdx.x; p < HW; p += NT) {
            const int y = (int)(((float)p + 0.5f) * inv_w), x = p - y * W;
            f16x8 m = T[p];
#pragma unroll
            for (int d = -2; d <= 2; ++d)
                if (d) m = vmax(m, T[min(max(y + d, 0), H - 1) * W + x]);
            X[p] = m;
            *reinterpret_cast<f16x8*>(ob + (long long)p * cs) = m;
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) maxpool2_h2_kernel(const float* in, int in_cs, int in_choff, float* out, int out_cs, int out_choff,
                                                           int un, int B, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)B * Ho * Wo * un;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= total) return;
    const int u = (int)(i % un);
    long long t = i / un;
    const int x = (int)(t % Wo); t /= Wo;
    const int y = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const long long p00 = ((long long)n * H + 2 * y) * W + 2 * x;
    const H2Elems<8> a0 = h2_load_unit(h2_unit_ptr(in, p00, in_cs, in_choff, u)), a1 = h2_load_unit(h2_unit_ptr(in, p00 + 1, in_cs, in_choff, u));
    const H2Elems<8> b0 = h2_load_unit(h2_unit_ptr(in, p00 + W, in_cs, in_choff, u)), b1 = h2_load_unit(h2_unit_ptr(in, p00 + W + 1, in_cs, in_choff, u));
    const H2Elems<8> m = h2_max(h2_max(a0, a1), h2_max(b0, b1));
    h2_store_unit(const_cast<char*>(h2_unit_ptr(out, ((long long)n * Ho + y) * Wo + x, out_cs, out_choff, u)), m);
}
// fp32 NHWC -> h2 pairs, whole 16-channel groups (the network input of a generic h2 graph, pa_tracknet_infer)
__global__ void __launch_bounds__(256) h2_encode_kernel(const float* in, float* out, long long n4, unsigned* ovf) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;      // 4-channel piece
    bool bad = false;
    if (i < n4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + i * 4);
        h16x4 hv, mv;
        h2_encode4(v, hv, mv, bad);
        char* op = reinterpret_cast<char*>(out) + (i >> 2) * 64 + (i & 3) * 8;
        *reinterpret_cast<h16x4*>(op) = hv;
        *reinterpret_cast<h16x4*>(op + 32) = mv;
    }
    h2_raise(ovf, bad);
}
hipError_t launch_h2_encode(const float* in, float* out, long long n_floats, unsigned* ovf_flag, hipStream_t s) {
    if (n_floats & 15) return hipErrorInvalidValue;
    const long long n4 = n_floats / 4;
    hipLaunchKernelGGL(h2_encode_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, in, out, n4, ovf_flag);
    return hipGetLastError();
}

constexpr size_t kSppfMaxLds = 150 * 1024;
// once per engine, outside any stream capture: kernels whose dynamic LDS exceeds the 64 KB default
hipError_t init_misc_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_h2_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSppfMaxLds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_h2_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSppfMaxLds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_h2_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSppfMaxLds);
    // (hipFuncSetAttribute applies to the current device: once per engine, not behind a process-wide flag — ADVICE r4)
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_f16_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSppfMaxLds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_f16_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSppfMaxLds);
    return e;
}

// f16: storage type of the buffer — 0 fp32, 1 fp16, 2 h2 pairs
hipError_t launch_sppf_pool(float* buf, int cs, int choff, int c, int B, int H, int W, hipStream_t s, int f16, int fused) {
    if (f16 == 2) {
        if ((c | choff) & 7) return hipErrorInvalidValue;
        const size_t lds = (size_t)2 * H * W * sizeof(Elem4);
        const long long wgs = ((long long)B * ((c + 15) / 16) + 7) / 8 * 8 * 4;          // see the id mapping in sppf_h2_kernel
        // measured on the c3 graphs (profiles/r4n_sppf_ab.txt: three launches 0.69 / 0.135 / 0.072 ms for the 40 x 40 x 288,
        // 12 x 20 x 288 and 12 x 20 x 128 maps of 64 images; fused 0.35 / 0.043 / 0.023): big maps want the 16 waves of a
        // 1024-thread workgroup (their two LDS planes leave room for one workgroup per CU), small ones several 256-thread
        // workgroups per CU.  fused = 2 / 3 / 4: force 256 / 512 / 1024 threads (tuning only)
        if (fused && lds <= kSppfMaxLds && wgs < (1ll << 31)) {          // (init_misc_kernels raised the kernel's dynamic-LDS limit)
            const dim3 grid((unsigned)wgs);
            const int nt = fused == 2 ? 256 : fused == 3 ? 512 : fused == 4 ? 1024 : (H * W > 1024 ? 1024 : 256);
            if (nt == 256) hipLaunchKernelGGL(sppf_h2_kernel<256>, grid, dim3(256), lds, s, buf, cs, choff, c, B, H, W);
            else if (nt == 512) hipLaunchKernelGGL(sppf_h2_kernel<512>, grid, dim3(512), lds, s, buf, cs, choff, c, B, H, W);
            else hipLaunchKernelGGL(sppf_h2_kernel<1024>, grid, dim3(1024), lds, s, buf, cs, choff, c, B, H, W);
            return hipGetLastError();
        }
        const long long total = (long long)B * H * W * (c / 8);
        for (int k = 0; k < 3; ++k) {
            hipLaunchKernelGGL(pool5_h2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, buf, cs, choff + k * c,
                               choff + (k + 1) * c, c / 8, B, H, W);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    if (f16 == 1 && fused && !((c | choff | cs) & 7) && (size_t)2 * H * W * 16 <= kSppfMaxLds && (long long)B * (c / 8) < (1ll << 31)) {
        const size_t lds = (size_t)2 * H * W * 16;
        const dim3 grid((unsigned)(B * (c / 8)));
        _Float16* hb = reinterpret_cast<_Float16*>(buf);
        // (init_misc_kernels raised the kernel's dynamic-LDS limit on this engine's device)
        if (H * W > 1024) hipLaunchKernelGGL(sppf_f16_kernel<1024>, grid, dim3(1024), lds, s, hb, cs, choff, c, B, H, W);
        else hipLaunchKernelGGL(sppf_f16_kernel<256>, grid, dim3(256), lds, s, hb, cs, choff, c, B, H, W);
        return hipGetLastError();
    }
    const int vn = f16 ? 8 : 4;
    const long long total = (long long)B * H * W * (c / vn);
    const unsigned grid = (unsigned)((total + 255) / 256);
    for (int k = 0; k < 3; ++k) {
        if (f16) hipLaunchKernelGGL(pool5_kernel<f16x8>, dim3(grid), dim3(256), 0, s, reinterpret_cast<_Float16*>(buf), cs,
                                    choff + k * c, choff + (k + 1) * c, c / vn, B, H, W);
        else hipLaunchKernelGGL(pool5_kernel<f32x4>, dim3(grid), dim3(256), 0, s, buf, cs, choff + k * c, choff + (k + 1) * c,
                                c / vn, B, H, W);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <typename V>
__global__ void __launch_bounds__(256) upsample2x_kernel(const typename VecT<V>::E* in, int in_cs, int in_choff,
                                                          typename VecT<V>::E* out, int out_cs, int out_choff, int c4n,
                                                          int B, int H, int W) {
    constexpr int VN = VecT<V>::N;
    const int Ho = H * 2, Wo = W * 2;
    const long long total = (long long)B * Ho * Wo * c4n;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % c4n);
    long long t = i / c4n;
    const int x = (int)(t % Wo); t /= Wo;
    const int y = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const V v = *reinterpret_cast<const V*>(in + (((long long)n * H + (y >> 1)) * W + (x >> 1)) * in_cs + in_choff + c4 * VN);
    *reinterpret_cast<V*>(out + (((long long)n * Ho + y) * Wo + x) * out_cs + out_choff + c4 * VN) = v;
}

hipError_t launch_upsample2x(const float* in, int in_cs, int in_choff, float* out, int out_cs, int out_choff,
                             int c, int B, int H, int W, hipStream_t s, int f16) {
    if (f16 == 2) {                    // h2: whole 16-channel groups are moved as bytes, like fp32
        if ((c | in_choff | out_choff) & 15) return hipErrorInvalidValue;
        f16 = 0;
    }
    const int vn = f16 ? 8 : 4;
    const long long total = (long long)B * H * 2 * W * 2 * (c / vn);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (f16) hipLaunchKernelGGL(upsample2x_kernel<f16x8>, grid, dim3(256), 0, s, reinterpret_cast<const _Float16*>(in), in_cs,
                                in_choff, reinterpret_cast<_Float16*>(out), out_cs, out_choff, c / vn, B, H, W);
    else hipLaunchKernelGGL(upsample2x_kernel<f32x4>, grid, dim3(256), 0, s, in, in_cs, in_choff, out, out_cs, out_choff,
                            c / vn, B, H, W);
    return hipGetLastError();
}

template <typename V>
__global__ void __launch_bounds__(256) maxpool2_kernel(const typename VecT<V>::E* in, int in_cs, int in_choff,
                                                        typename VecT<V>::E* out, int out_cs, int out_choff, int c4n,
                                                        int B, int H, int W) {
    constexpr int VN = VecT<V>::N;
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)B * Ho * Wo * c4n;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % c4n);
    long long t = i / c4n;
    const int x = (int)(t % Wo); t /= Wo;
    const int y = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const auto* p = in + (((long long)n * H + 2 * y) * W + 2 * x) * in_cs + in_choff + c4 * VN;
    const V a0 = *reinterpret_cast<const V*>(p);
    const V a1 = *reinterpret_cast<const V*>(p + in_cs);
    const V b0 = *reinterpret_cast<const V*>(p + (long long)W * in_cs);
    const V b1 = *reinterpret_cast<const V*>(p + (long long)W * in_cs + in_cs);
    *reinterpret_cast<V*>(out + (((long long)n * Ho + y) * Wo + x) * out_cs + out_choff + c4 * VN) = vmax(vmax(a0, a1), vmax(b0, b1));
}

hipError_t launch_maxpool2(const float* in, int in_cs, int in_choff, float* out, int out_cs, int out_choff,
                           int c, int B, int H, int W, hipStream_t s, int f16) {
    if (f16 == 2) {
        if ((c | in_choff | out_choff) & 7) return hipErrorInvalidValue;
        const long long tot = (long long)B * (H / 2) * (W / 2) * (c / 8);
        hipLaunchKernelGGL(maxpool2_h2_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, in, in_cs, in_choff, out, out_cs,
                           out_choff, c / 8, B, H, W);
        return hipGetLastError();
    }
    const int vn = f16 ? 8 : 4;
    const long long total = (long long)B * (H / 2) * (W / 2) * (c / vn);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (f16) hipLaunchKernelGGL(maxpool2_kernel<f16x8>, grid, dim3(256), 0, s, reinterpret_cast<const _Float16*>(in), in_cs,
                                in_choff, reinterpret_cast<_Float16*>(out), out_cs, out_choff, c / vn, B, H, W);
    else hipLaunchKernelGGL(maxpool2_kernel<f32x4>, grid, dim3(256), 0, s, in, in_cs, in_choff, out, out_cs, out_choff,
                            c / vn, B, H, W);
    return hipGetLastError();
}

}  // namespace padel
