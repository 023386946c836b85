// K10 — TrackNet pre/post kernels of the ball path (gfx950), HBM-bound.
//   ball_assemble_kernel : builds the (27 -> 32 channel) fp32 NHWC network input of a batch of 8-frame
//                          windows from the resized background + resized frames kept as uint8 in HBM
//                          (reference: ball_tracker/iterable.py:167-199 process_chunck, bg_mode "concat").
//                          u8 -> float goes through a 256-entry table built on the host as
//                          float(double(u)/255.0), i.e. exactly `frames /= 255.` (float64) then `.float()`.
//   ball_ensemble_kernel : temporal ensemble of the 8 overlapping window outputs + threshold
//                          (ball_tracker.py:449-509, predict.py:184-189): out = sum_k c_k * Y[row0+k][slot 7-k]
//                          (products rounded separately, summed in k order like torch's (rows*w).sum(0)),
//                          or sum_k Y / div for the head / tail means; mask = out > 0.5.
#include "h2_common.h"

namespace padel {

__global__ void __launch_bounds__(256) ball_assemble_kernel(const BallAssembleArgs a) {
    __shared__ float lut[256];
    lut[threadIdx.x] = a.lut[threadIdx.x];
    __syncthreads();
    const long long total = (long long)a.B * a.H * a.W;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= total) return;
    const int pix = (int)(i % ((long long)a.H * a.W));
    const int b = (int)(i / ((long long)a.H * a.W));
    float v[32];
    const uint8_t* m = a.median + (long long)pix * 3;
    v[0] = lut[m[0]]; v[1] = lut[m[1]]; v[2] = lut[m[2]];
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        const int slot = (a.first_slot + b + f) % a.ring;
        const uint8_t* p = a.frames + ((long long)slot * a.H * a.W + pix) * 3;
        v[3 + 3 * f] = lut[p[0]]; v[4 + 3 * f] = lut[p[1]]; v[5 + 3 * f] = lut[p[2]];
    }
#pragma unroll
    for (int c = 27; c < 32; ++c) v[c] = 0.0f;
    if (a.out_h2) {                      // h2 graph: two 16-channel groups of fp16 pairs (values in [0, 1]: always in range)
        bool bad = false;
        char* ob = reinterpret_cast<char*>(a.out + i * 32);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            h16x4 hv, mv;
            h2_encode4((f32x4){v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]}, hv, mv, bad);
            char* op = ob + (c >> 2) * 64 + (c & 3) * 8;
            *reinterpret_cast<h16x4*>(op) = hv;
            *reinterpret_cast<h16x4*>(op + 32) = mv;
        }
        return;
    }
    f32x4* o = reinterpret_cast<f32x4*>(a.out + i * 32);
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (f32x4){v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
}

hipError_t launch_ball_assemble(const BallAssembleArgs& a, hipStream_t s) {
    const long long total = (long long)a.B * a.H * a.W;
    hipLaunchKernelGGL(ball_assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) ball_ensemble_kernel(const BallEnsembleArgs a) {
    const int o = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int HW = a.H * a.W;
    if (pix >= HW) return;
    const int row0 = a.row0[o];
    const int mode = a.mode[o];          // 0: weighted sum, 1: plain sum / div
    const float div = a.div[o];
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float y = a.Y[((long long)(row0 + k) * HW + pix) * a.cs + (7 - k)];
        acc = (mode == 0) ? __fadd_rn(acc, __fmul_rn(y, a.w[k])) : __fadd_rn(acc, y);
    }
    if (mode == 1) acc = acc / div;
    if (a.heat) a.heat[(long long)o * HW + pix] = acc;
    a.mask[(long long)o * HW + pix] = acc > a.threshold ? 255 : 0;
}

hipError_t launch_ball_ensemble(const BallEnsembleArgs& a, int nout, hipStream_t s) {
    dim3 grid((a.H * a.W + 255) / 256, nout, 1);
    hipLaunchKernelGGL(ball_ensemble_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace padel

// ------------------------------------------------------------------------------------------------
// ball_locate_kernel — predict_location (reference predict.py:7-39) on the device: 8-connected components of
// the thresholded heat map, bounding rectangle of each, the one of maximal w*h wins; ties go to the component
// discovered LAST in raster order (cv2.findContours returns contours in reverse discovery order and the
// reference keeps the first maximum with a strict '>').
//
// Masks are almost empty (a ball is ~100 pixels), so the kernel works on a sparse list: (a) compact the
// foreground pixel indices into LDS, (b) label[p] = p+1, (c) iterate "min over the 8 neighbours" + one pointer
// jump until nothing changes (labels converge to the smallest raster index of the component = the pixel a raster
// scan discovers first), (d) per-root bounding boxes with atomics, (e) arg-max of (area, root index).
// One 1024-thread workgroup per frame; rect = {x, y, w, h}, all zero when the mask is empty, w = -1 when the
// foreground does not fit the LDS list (caller falls back to the mask).
#define LOC_THREADS 1024
#define LOC_CAP 12288

namespace padel {

__global__ void __launch_bounds__(LOC_THREADS) ball_locate_kernel(const BallLocateArgs a) {
    __shared__ int fg[LOC_CAP];
    __shared__ int n_fg;
    __shared__ int changed;
    __shared__ unsigned long long best;
    const int o = blockIdx.x, tid = threadIdx.x;
    const int H = a.H, W = a.W, HW = H * W;
    const uint8_t* mask = a.mask + (long long)o * HW;
    int* label = a.label + (long long)o * HW;
    int* bx0 = a.bbox + (long long)o * 4 * HW;
    int* bx1 = bx0 + HW; int* by0 = bx1 + HW; int* by1 = by0 + HW;
    if (tid == 0) { n_fg = 0; best = 0ull; }
    __syncthreads();
    for (int p4 = tid; p4 < HW / 4; p4 += LOC_THREADS) {
        const uchar4 m = reinterpret_cast<const uchar4*>(mask)[p4];
        const uint8_t mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (mm[k]) {
                const int slot = atomicAdd(&n_fg, 1);
                if (slot < LOC_CAP) fg[slot] = p4 * 4 + k;
            }
    }
    __syncthreads();
    const int n = n_fg;
    int* rect = a.rect + o * 4;
    if (n == 0) { if (tid == 0) { rect[0] = rect[1] = rect[2] = rect[3] = 0; } return; }
    if (n > LOC_CAP) { if (tid == 0) { rect[0] = rect[1] = rect[3] = 0; rect[2] = -1; } return; }
    for (int i = tid; i < n; i += LOC_THREADS) label[fg[i]] = fg[i] + 1;
    __syncthreads();
    for (int it = 0; it < 4 * (H + W); ++it) {            // upper bound on the geodesic diameter; exits early
        if (tid == 0) changed = 0;
        __syncthreads();
        for (int i = tid; i < n; i += LOC_THREADS) {
            const int p = fg[i];
            const int y = p / W, x = p - y * W;
            const int l = label[p];
            int m = l;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if ((dy | dx) && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W && mask[yy * W + xx])
                        m = min(m, label[yy * W + xx]);
                }
            m = min(m, label[m - 1]);                     // pointer jump towards the root
            if (m < l) { atomicMin(&label[p], m); changed = 1; }
        }
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    for (int i = tid; i < n; i += LOC_THREADS) {
        const int p = fg[i];
        if (label[p] == p + 1) { const int y = p / W, x = p - y * W; bx0[p] = x; bx1[p] = x; by0[p] = y; by1[p] = y; }
    }
    __syncthreads();
    for (int i = tid; i < n; i += LOC_THREADS) {
        const int p = fg[i];
        const int r = label[p] - 1;
        const int y = p / W, x = p - y * W;
        atomicMin(&bx0[r], x); atomicMax(&bx1[r], x); atomicMin(&by0[r], y); atomicMax(&by1[r], y);
    }
    __syncthreads();
    for (int i = tid; i < n; i += LOC_THREADS) {
        const int p = fg[i];
        if (label[p] == p + 1) {
            const unsigned long long area = (unsigned long long)(bx1[p] - bx0[p] + 1) * (unsigned long long)(by1[p] - by0[p] + 1);
            atomicMax(&best, (area << 32) | (unsigned)(p + 1));
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int p = (int)(best & 0xffffffffull) - 1;
        rect[0] = bx0[p]; rect[1] = by0[p]; rect[2] = bx1[p] - bx0[p] + 1; rect[3] = by1[p] - by0[p] + 1;
    }
}

hipError_t launch_ball_locate(const BallLocateArgs& a, int nout, hipStream_t s) {
    hipLaunchKernelGGL(ball_locate_kernel, dim3(nout), dim3(LOC_THREADS), 0, s, a);
    return hipGetLastError();
}

}  // namespace padel

// ------------------------------------------------------------------------------------------------
// median_kernel — K11: per-pixel, per-channel median over the first N frames of the clip
// (reference ball_tracker/iterable.py:59-74: np.median(np.array(frames_rgb), 0) then .astype('uint8')).
// One thread per (pixel, channel) byte; its 256-bin histogram (u16 counts, N <= 65535) lives in LDS laid
// out [bin][thread] so that neighbouring threads touch neighbouring addresses; frames are read once,
// coalesced (consecutive threads = consecutive bytes).  np.median semantics: odd N -> the middle element;
// even N -> mean of the two middle elements computed in float64, and the uint8 cast truncates -> (a+b)>>1.
// Input frames are BGR, the output is RGB (channel 2-c), like the reference's cvtColor before the median.
#define MED_THREADS 128

namespace padel {

__global__ void __launch_bounds__(MED_THREADS) median_kernel(const uint8_t* frames, int N, long long frame_bytes,
                                                              uint8_t* out_rgb) {
    __shared__ unsigned short hist[256 * MED_THREADS];
    const int t = threadIdx.x;
    const long long e = (long long)blockIdx.x * MED_THREADS + t;
    for (int b = 0; b < 256; ++b) hist[b * MED_THREADS + t] = 0;
    if (e >= frame_bytes) return;
    for (int n = 0; n < N; ++n) {
        const int v = frames[(long long)n * frame_bytes + e];
        hist[v * MED_THREADS + t] += 1;
    }
    const int k_hi = N >> 1, k_lo = (N & 1) ? k_hi : k_hi - 1;       // 0-based ranks of the middle element(s)
    int cum = 0, lo = -1, hi = -1;
    for (int b = 0; b < 256; ++b) {
        cum += hist[b * MED_THREADS + t];
        if (lo < 0 && cum > k_lo) lo = b;
        if (cum > k_hi) { hi = b; break; }
    }
    const long long pix = e / 3;
    const int c = (int)(e - pix * 3);
    out_rgb[pix * 3 + (2 - c)] = (uint8_t)((lo + hi) >> 1);
}

hipError_t launch_median(const uint8_t* frames, int N, long long frame_bytes, uint8_t* out_rgb, hipStream_t s) {
    const long long blocks = (frame_bytes + MED_THREADS - 1) / MED_THREADS;
    hipLaunchKernelGGL(median_kernel, dim3((unsigned)blocks), dim3(MED_THREADS), 0, s, frames, N, frame_bytes, out_rgb);
    return hipGetLastError();
}

}  // namespace padel
