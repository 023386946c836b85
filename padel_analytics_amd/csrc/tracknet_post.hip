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
#include "kernels.h"

namespace padel {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
