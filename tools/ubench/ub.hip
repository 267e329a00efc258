// micro-benchmarks: issue cost of the building blocks of render_kernel on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../cuda-raytracer_amd/csrc/philox.h"

__device__ __forceinline__ rt_u32x4 philox_mad64(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    rt_u32x4 o = {c0, c1, c2, c3};
    return o;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t k0, uint32_t k1, float fa, float fb)
{
    uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0; float f = fa + tid * 1e-7f, g = fb;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { rt_u32x4 r = rt_philox4x32_10(i, 0, tid, acc, k0, k1); acc ^= r.x ^ r.y ^ r.z ^ r.w; }
        if (MODE == 1) { rt_u32x4 r = philox_mad64(i, 0, tid, acc, k0, k1); acc ^= r.x ^ r.y ^ r.z ^ r.w; }
        if (MODE == 2) {  // 16 dependent-free IEEE divisions
#pragma unroll
            for (int j = 0; j < 16; ++j) f += (g + j) / (f + j);
        }
        if (MODE == 3) {  // 160 fma
#pragma unroll
            for (int j = 0; j < 160; ++j) f = __builtin_fmaf(f, g, 1.0f + j);
        }
        if (MODE == 4) {  // 16 fast divisions (rcp * )
#pragma unroll
            for (int j = 0; j < 16; ++j) f += (g + j) * __builtin_amdgcn_rcpf(f + j);
        }
        if (MODE == 5) {  // 40 mul_hi
#pragma unroll
            for (int j = 0; j < 40; ++j) acc = __umulhi(acc + j, 0xD2511F53u) ^ tid;
        }
        if (MODE == 6) {  // 40 mul_lo
#pragma unroll
            for (int j = 0; j < 40; ++j) acc = (acc + j) * 0xD2511F53u ^ tid;
        }
    }
    out[tid] = acc + (uint32_t)f;
}
template <int MODE> void run(const char* name, uint32_t* d, int iters, double units)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int blocks = 256 * 8;
    k<MODE><<<blocks, 256>>>(d, 16, 1, 2, 1.5f, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(d, iters, 1, 2, 1.5f, 0.999f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double lanes = (double)blocks * 256;
    double per = ms * 1e-3 / (lanes * iters * units);           // seconds per unit per lane
    // chip issue: 1024 SIMDs; wave-instr per SIMD-second = 1/(per*64*...)
    double wave_units_per_s = lanes / 64 * iters * units / (ms * 1e-3);
    printf("%-28s %8.3f ms  %8.2f Gunit/s(lane)  %6.2f cycles/unit/SIMD @2.4GHz\n", name, ms, 1e-9 / per,
           2.4e9 * 1024 / wave_units_per_s);
}
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<3>("fma x160", d, 2000, 160);
    run<0>("philox (mulhi+mullo)/block", d, 4000, 1);
    run<1>("philox (64-bit product)/block", d, 4000, 1);
    run<5>("mul_hi_u32 x40", d, 4000, 40);
    run<6>("mul_lo_u32 x40", d, 4000, 40);
    run<2>("IEEE div x16", d, 4000, 16);
    run<4>("rcp*mul x16", d, 4000, 16);
    return 0;
}
