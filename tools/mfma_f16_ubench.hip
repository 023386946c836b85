// Pure-MFMA microbenchmark (tuning tool): sustained v_mfma_f32_16x16x32_f16 (and, round 4, v_mfma_f32_32x32x16_f16) rate on gfx950 with no memory traffic, as a
// function of resident waves per SIMD, independent accumulators per wave and operand DATA (all-zero operands draw far
// less power than random ones: the clock the chip sustains, and with it the reachable fraction of the 2.5 PFLOP/s nominal
// peak, depends on it).  This is the ceiling the h2 / fp16 convolution kernels are measured against in DESIGN.md.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_ubench.hip -o tools/mfma_f16_ubench && tools/mfma_f16_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(256) kf16(float* out, int iters, int mode) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    h16x8 a[3], b[2];
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { s = s * 1664525u + 1013904223u; a[i][k] = mode ? (_Float16)(((int)(s >> 20) - 2048) * (1.0f / 1024.0f)) : (_Float16)0.f; }
#pragma unroll
        for (int i = 0; i < 2; ++i) { s = s * 1664525u + 1013904223u; b[i][k] = mode ? (_Float16)(((int)(s >> 20) - 2048) * (1.0f / 1024.0f)) : (_Float16)0.f; }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i % 3], b[(i + r) & 1], acc[i], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 12345.678f) out[0] = t;
}

// the 32 x 32 x 16 shape (same FLOPs per operand register, half the instructions; 16 accumulator registers per tile)
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) kf16_32(float* out, int iters, int mode) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h16x8 a[3], b[2];
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { s = s * 1664525u + 1013904223u; a[i][k] = mode ? (_Float16)(((int)(s >> 20) - 2048) * (1.0f / 1024.0f)) : (_Float16)0.f; }
#pragma unroll
        for (int i = 0; i < 2; ++i) { s = s * 1664525u + 1013904223u; b[i][k] = mode ? (_Float16)(((int)(s >> 20) - 2048) * (1.0f / 1024.0f)) : (_Float16)0.f; }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i % 3], b[(i + r) & 1], acc[i], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][r];
    if (t == 12345.678f) out[0] = t;
}

template <class K>
static double run(K kern, int blocks, int iters, int mode, double flop_per_iter_per_wave, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return flop_per_iter_per_wave * iters * blocks * 4.0 / (ms * 1e-3) / 1e12;
}

int main() {
    float* d; hipMalloc(&d, 1024);
    const double fl = 2.0 * 16 * 16 * 32;      // per MFMA
    for (int iters : {20000, 200000}) {          // ~4 ms and ~40 ms of MFMAs: the longer run sees the settled clock
        printf("iters %d (TFLOP/s of v_mfma_f32_16x16x32_f16; nominal peak 2516 at 2.4 GHz)\n", iters);
        printf("%-18s waves/SIMD  NACC=4      6     12     18     36\n", "operands");
        for (int mode : {0, 1})
            for (int wps : {1, 2, 3}) {
                const int blocks = 256 * wps;         // 256 CUs x wps workgroups of 4 waves (one wave per SIMD each)
                printf("%-18s %9d  ", mode ? "random in [-2,2)" : "all zero", wps);
                printf("%7.0f ", run(kf16<4>, blocks, iters, mode, 4.0 * 4 * fl, d));
                printf("%6.0f ", run(kf16<6>, blocks, iters, mode, 4.0 * 6 * fl, d));
                printf("%6.0f ", run(kf16<12>, blocks, iters, mode, 4.0 * 12 * fl, d));
                printf("%6.0f ", run(kf16<18>, blocks, iters, mode, 4.0 * 18 * fl, d));
                printf("%6.0f\n", run(kf16<36>, blocks, iters, mode, 4.0 * 36 * fl, d));
            }
    }
    const double fl32 = 2.0 * 32 * 32 * 16;
    for (int iters : {10000, 100000}) {
        printf("iters %d (TFLOP/s of v_mfma_f32_32x32x16_f16)\n", iters);
        printf("%-18s waves/SIMD  NACC=1      2      3      6      9\n", "operands");
        for (int mode : {0, 1})
            for (int wps : {1, 2, 3}) {
                const int blocks = 256 * wps;
                printf("%-18s %9d  ", mode ? "random in [-2,2)" : "all zero", wps);
                printf("%7.0f ", run(kf16_32<1>, blocks, iters, mode, 4.0 * 1 * fl32, d));
                printf("%6.0f ", run(kf16_32<2>, blocks, iters, mode, 4.0 * 2 * fl32, d));
                printf("%6.0f ", run(kf16_32<3>, blocks, iters, mode, 4.0 * 3 * fl32, d));
                printf("%6.0f ", run(kf16_32<6>, blocks, iters, mode, 4.0 * 6 * fl32, d));
                printf("%6.0f\n", run(kf16_32<9>, blocks, iters, mode, 4.0 * 9 * fl32, d));
            }
    }
    return 0;
}
