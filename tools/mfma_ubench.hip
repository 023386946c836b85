// Pure-MFMA microbenchmark (tuning tool): v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 issue rate on gfx950 as a
// function of resident waves per SIMD and independent accumulators per wave, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_ubench.hip -o tools/mfma_ubench && tools/mfma_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
    if (s == 12345.678f) out[0] = s;
}

template <class K>
static double run(K kern, int blocks, int iters, double flop_per_iter_per_wave, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return flop_per_iter_per_wave * iters * blocks * 4.0 / (ms * 1e-3) / 1e12;
}

int main() {
    float* d; hipMalloc(&d, 1024);
    const int iters = 20000;
    printf("waves/SIMD  k16x4:NACC=2     4       6       8      16   | k32x2:NACC=1     2       4\n");
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;         // 256 CUs x wps workgroups of 4 waves (one wave per SIMD each)
        printf("%9d  ", wps);
        printf("%14.1f ", run(k16<2>, blocks, iters, 4.0 * 2 * 2048, d));
        printf("%7.1f ", run(k16<4>, blocks, iters, 4.0 * 4 * 2048, d));
        printf("%7.1f ", run(k16<6>, blocks, iters, 4.0 * 6 * 2048, d));
        printf("%7.1f ", run(k16<8>, blocks, iters, 4.0 * 8 * 2048, d));
        printf("%7.1f   |", run(k16<16>, blocks, iters, 4.0 * 16 * 2048, d));
        printf("%14.1f ", run(k32<1>, blocks, iters, 4.0 * 1 * 4096, d));
        printf("%7.1f ", run(k32<2>, blocks, iters, 4.0 * 2 * 4096, d));
        printf("%7.1f\n", run(k32<4>, blocks, iters, 4.0 * 4 * 4096, d));
    }
    return 0;
}
