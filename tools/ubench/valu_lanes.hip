// Does a VALU instruction of a wave64 cost the same whether 4, 16 or 64 of its lanes are active?  (tools/ubench, round 3)
// Every wave runs the same loop of independent fp32 FMAs under an EXEC mask chosen by `mask_kind`:
//   0: all 64 lanes; 1: lanes 0-47; 2: lanes 0-31; 3: lanes 0-15; 4: lanes 0, 16, 32, 48 (one per quarter); 5: every other lane
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_lanes.hip -o tools/ubench/valu_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ __launch_bounds__(256) void k(float* out, int iters, int mask_kind)
{
    const int lane = threadIdx.x & 63;
    bool on = true;
    if (mask_kind == 1) on = lane < 48;
    if (mask_kind == 2) on = lane < 32;
    if (mask_kind == 3) on = lane < 16;
    if (mask_kind == 4) on = (lane & 15) == 0;
    if (mask_kind == 5) on = (lane & 1) == 0;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    const float m = 1.0001f, c = 0.5f;
    if (on) {
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
            a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
    const int blocks = 256 * 8, iters = 20000;      // 8 blocks of 4 waves per CU: 8 waves per SIMD
    float* out;
    CHK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const char* names[] = {"64 lanes", "lanes 0-47", "lanes 0-31", "lanes 0-15", "4 lanes, one per quarter", "32 lanes, every other"};
    for (int kind = 0; kind < 6; ++kind) {
        k<<<blocks, 256>>>(out, 100, kind);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        k<<<blocks, 256>>>(out, iters, kind);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double insts = (double)blocks * 4 * iters * 8;      // wave-instructions
        std::printf("%-28s %8.3f ms   %.2f cycles per wave-instruction per SIMD (2.4 GHz, 1024 SIMDs)\n", names[kind], ms, ms * 1e-3 * 2.4e9 * 1024 / insts);
    }
    return 0;
}
