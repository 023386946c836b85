// Semantics probe for `buffer_load_dwordx4 ... offen lds` on gfx950 (tuning tool): where do the 16 bytes of lane l
// land, what do out-of-range lanes write, is the SGPR offset part of the range check?
//   hipcc --offload-arch=gfx950 -O2 tools/dma_probe.hip -o tools/dma_probe && tools/dma_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int IMM>
__device__ __forceinline__ void dma16(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lb) {
    asm volatile("s_add_u32 m0, %[lb], %[imm]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[vo], %[rs], %[so] offen lds"
                 :: [lb] "s"(lb), [imm] "n"(IMM), [vo] "v"(voff), [rs] "s"(rsrc), [so] "s"(soff) : "memory", "scc");
}
__global__ void k(const float* in, float* out, unsigned soff) {
    __shared__ float lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 7.0f;
    __syncthreads();
    const unsigned long long b = (unsigned long long)(uintptr_t)in;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xFFFFu));
    r[2] = (int)0x80000000u;
    r[3] = 0x00020000;
    const unsigned lb = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
    unsigned voff = (63 - threadIdx.x) * 16u;            // reversed source order: lane l reads floats [4*(63-l), +4)
    if ((threadIdx.x & 7) == 3) voff = 0xFFFFFFF0u;      // out of range
    dma16<1024>(voff, r, __builtin_amdgcn_readfirstlane(soff), lb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 1000.f + i;
    float *din, *dout;
    hipMalloc(&din, 4096 * 4); hipMalloc(&dout, 2048 * 4);
    hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    const unsigned soff = 512 * 4;                        // SGPR offset: + 512 floats
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, soff);
    std::vector<float> o(2048);
    if (hipMemcpy(o.data(), dout, 2048 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAILED: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    int ok_place = 1, ok_zero = 1, untouched = 1;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const float got = o[256 + l * 4 + e];         // LDS byte 1024 + 16 l
            const bool oob = (l & 7) == 3;
            const float want = oob ? 0.f : 1000.f + 512 + (63 - l) * 4 + e;
            if (oob) { if (got != 0.f) ok_zero = 0; } else if (got != want) ok_place = 0;
        }
    for (int i = 0; i < 256; ++i) if (o[i] != 7.f) untouched = 0;
    for (int i = 512; i < 2048; ++i) if (o[i] != 7.f) untouched = 0;
    printf("dma_probe: placement(base+imm+16*lane, soffset added) %s ; out-of-range lanes write zeros %s (lane 3 got %g %g) ; rest of LDS untouched %s\n",
           ok_place ? "OK" : "WRONG", ok_zero ? "OK" : "NO", o[256 + 12], o[256 + 13], untouched ? "OK" : "NO");
    return (ok_place && ok_zero && untouched) ? 0 : 2;
}
