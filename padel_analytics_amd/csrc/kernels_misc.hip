// K1/K2/K5/K6/K7 — HBM-bound helper kernels of the trackers' hot path (gfx950).
//   letterbox_kernel      : ultralytics LetterBox (cv2 INTER_LINEAR u8 + pad 114) -> u8 NHWC4 net input
//                           (players_tracker.py:346-359 via [upstream] preprocess)
//   resample_pass_kernel  : one separable pass of Pillow's 8-bit bicubic resize
//                           (players_keypoints_tracker.py:264-266, ball_tracker/iterable.py:80,188)
//   stem_kernel           : model.0 Conv(3,c,3,2)+BN+SiLU straight from the u8 net input (K=27 is
//                           MFMA-unfriendly and <1.5% of the FLOPs: plain VALU FMA)
//   pool5_kernel          : MaxPool2d(5,1,2) on a channel slice (SPPF, chained 3x like upstream)
//   upsample2x / maxpool2 : nearest x2 into a concat slice; MaxPool2d(2,2) (TrackNet models.py:60-64)
#include "kernels.h"

namespace padel {

typedef float f32x4 __attribute__((ext_vector_type(4)));
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

hipError_t launch_resample_pass(const ResamplePassArgs& a, hipStream_t s) {
    const long long total = (long long)a.B * a.out_h * a.out_w;
    const int grid = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(resample_pass_kernel, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ stem conv
// one thread = one output pixel x 16 output channels; blockIdx.y = channel group
__global__ void __launch_bounds__(256) stem_kernel(const StemArgs a) {
    __shared__ float lut[256];          // u8 -> u8/255 exactly as `im.float() /= 255`
    __shared__ float ws[16 * 27 + 16];
    const int cg = blockIdx.y;
    lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
    for (int i = threadIdx.x; i < 16 * 27; i += 256) ws[i] = a.w[cg * 16 * 27 + i];
    if (threadIdx.x < 16) ws[16 * 27 + threadIdx.x] = a.bias[cg * 16 + threadIdx.x];
    __syncthreads();
    const long long total = (long long)a.B * a.Ho * a.Wo;
    const long long p = blockIdx.x * 256ll + threadIdx.x;
    if (p >= total) return;
    const int ox = (int)(p % a.Wo);
    const long long t = p / a.Wo;
    const int oy = (int)(t % a.Ho);
    const int n = (int)(t / a.Ho);
    const uint32_t* img = reinterpret_cast<const uint32_t*>(a.in) + (long long)n * a.H * a.W;
    float x[27];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
            const bool v = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const uint32_t px = v ? img[(long long)iy * a.W + ix] : 0u;
            x[(ky * 3 + kx) * 3 + 0] = v ? lut[px & 255u] : 0.0f;
            x[(ky * 3 + kx) * 3 + 1] = v ? lut[(px >> 8) & 255u] : 0.0f;
            x[(ky * 3 + kx) * 3 + 2] = v ? lut[(px >> 16) & 255u] : 0.0f;
        }
    }
    float* o = a.out + p * a.out_cs + a.out_choff + cg * 16;
    _Float16* oh = reinterpret_cast<_Float16*>(a.out) + p * a.out_cs + a.out_choff + cg * 16;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
        f32x4 r;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int c = c4 * 4 + cc;
            float acc = ws[16 * 27 + c];
#pragma unroll
            for (int k = 0; k < 27; ++k) acc = fmaf(x[k], ws[c * 27 + k], acc);
            r[cc] = acc / (1.0f + expf(-acc));
        }
        if (a.out_f16) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) oh[c4 * 4 + cc] = (_Float16)r[cc];
        } else {
            *reinterpret_cast<f32x4*>(o + c4 * 4) = r;
        }
    }
}

hipError_t launch_stem(const StemArgs& a, hipStream_t s) {
    const long long total = (long long)a.B * a.Ho * a.Wo;
    dim3 grid((unsigned)((total + 255) / 256), a.cout / 16, 1);
    hipLaunchKernelGGL(stem_kernel, grid, dim3(256), 0, s, a);
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
    V m = *reinterpret_cast<const V*>(buf + (((long long)n * H + y) * W + x) * cs + src_off + c4 * VN);   // the centre is always inside
    for (int dy = -2; dy <= 2; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)H) continue;
        for (int dx = -2; dx <= 2; ++dx) {
            const int xx = x + dx;
            if ((unsigned)xx >= (unsigned)W) continue;
            m = vmax(m, *reinterpret_cast<const V*>(buf + (((long long)n * H + yy) * W + xx) * cs + src_off + c4 * VN));
        }
    }
    *reinterpret_cast<V*>(buf + (((long long)n * H + y) * W + x) * cs + dst_off + c4 * VN) = m;
}

hipError_t launch_sppf_pool(float* buf, int cs, int choff, int c, int B, int H, int W, hipStream_t s, int f16) {
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
