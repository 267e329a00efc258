// What does a packed fp32 instruction cost next to a plain one on gfx950?  (tools/ubench, round 3)
// Eight independent chains per lane, 8 waves per SIMD; cycles per wave-instruction per SIMD at the clock the run reports.
// Plain and packed fma / mul / add, byte conversion, min / max / med3, moves and selects, compares, the pieces of an IEEE
// division, the integer multiplies of a Philox round.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_costs.hip -o tools/ubench/valu_costs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    const int lane = threadIdx.x & 63;
    f2 a[8];
    float s[8];
    uint32_t u[8];
    for (int q = 0; q < 8; ++q) { a[q] = f2{(float)(lane + q), (float)(lane - q)}; s[q] = (float)(lane + q); u[q] = (uint32_t)(lane * 7 + q); }
    f2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    asm volatile("s_mov_b32 s22, 0x55555555\n s_mov_b32 s23, 0x55555555" : : : "s22", "s23");
    for (int i = 0; i < iters; ++i) {
#define PKFMA(q) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[q]) : "v"(m), "v"(c));
#define FMA(q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[q]) : "v"(m.x), "v"(c.x));
#define PKMUL(q) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[q]) : "v"(m));
#define MUL(q) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[q]) : "v"(m.x));
#define PKADD(q) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[q]) : "v"(c));
#define ADD(q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[q]) : "v"(c.x));
#define CVT(q) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(s[q]) : "v"(u[q]));
#define MIN(q) asm volatile("v_min_f32 %0, %0, %1" : "+v"(s[q]) : "v"(c.x));
#define MAX3(q) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(s[q]) : "v"(c.x), "v"(m.x));
#define MOV(q) asm volatile("v_mov_b32 %0, %1" : "=v"(s[q]) : "v"(u[q]));
        if (KIND == 0) { REP8(FMA) }
        if (KIND == 1) { REP8(PKFMA) }
        if (KIND == 2) { REP8(MUL) }
        if (KIND == 3) { REP8(PKMUL) }
        if (KIND == 4) { REP8(ADD) }
        if (KIND == 5) { REP8(PKADD) }
        if (KIND == 6) { REP8(CVT) }
        if (KIND == 7) { REP8(MIN) }
        if (KIND == 8) { REP8(MAX3) }
        if (KIND == 9) { REP8(MOV) }
#define CNDMASK(q) asm volatile("v_cndmask_b32 %0, %0, %1, s[22:23]" : "+v"(s[q]) : "v"(c.x) : );
#define CMP(q) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(s[q]), "v"(c.x) : "vcc");
#define CMPS(q) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(s[q]), "v"(c.x) : "s20", "s21");
#define RCP(q) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[q]));
#define SQRT(q) asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[q]));
#define DSCALE(q) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(s[q]) : "v"(c.x) : "vcc");
#define DFMAS(q) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(s[q]) : "v"(c.x), "v"(m.x) : "vcc");
#define DFIXUP(q) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(s[q]) : "v"(c.x), "v"(m.x));
#define MAD64(q) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, 0" : "+v"(a[q]) : "v"(u[q]), "v"(u[(q + 1) & 7]) : "s20", "s21");
#define XOR(q) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[q]) : "v"(u[(q + 1) & 7]));
#define MED3(q) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(s[q]) : "v"(c.x), "v"(m.x));
#define MULLO(q) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[q]) : "v"(u[(q + 1) & 7]));
#define MULHI(q) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[q]) : "v"(u[(q + 1) & 7]));
#define CVTU(q) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(s[q]) : "v"(u[q]));
#define ADDU(q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[q]) : "v"(u[(q + 1) & 7]));
#define SUB(q) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s[q]) : "v"(c.x));
#define PKMOV(q) asm volatile("v_pk_mov_b32 %0, %1, %1" : "=v"(a[q]) : "v"(m));
        if (KIND == 10) { REP8(CNDMASK) }
        if (KIND == 11) { REP8(CMP) }
        if (KIND == 12) { REP8(CMPS) }
        if (KIND == 13) { REP8(RCP) }
        if (KIND == 14) { REP8(SQRT) }
        if (KIND == 15) { REP8(DSCALE) }
        if (KIND == 16) { REP8(DFMAS) }
        if (KIND == 17) { REP8(DFIXUP) }
        if (KIND == 18) { REP8(MAD64) }
        if (KIND == 19) { REP8(XOR) }
        if (KIND == 20) { REP8(MED3) }
        if (KIND == 21) { REP8(MULLO) }
        if (KIND == 22) { REP8(MULHI) }
        if (KIND == 23) { REP8(CVTU) }
        if (KIND == 24) { REP8(ADDU) }
        if (KIND == 25) { REP8(SUB) }
        if (KIND == 26) { REP8(PKMOV) }
    }
    float r = 0.f;
    for (int q = 0; q < 8; ++q) r += a[q].x + a[q].y + s[q] + (float)u[q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND>
static void run(const char* name, float* out, hipEvent_t e0, hipEvent_t e1)
{
    const int blocks = 256 * 8, iters = 100000;
    k<KIND><<<blocks, 256>>>(out, 100);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    k<KIND><<<blocks, 256>>>(out, iters);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double insts = (double)blocks * 4 * iters * 8;
    std::printf("{\"instruction\": \"%s\", \"ms\": %.3f, \"cycles_per_wave_instruction_per_SIMD_at_2.4GHz\": %.2f}\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / insts);
}

int main()
{
    float* out;
    CHK(hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    run<0>("v_fma_f32", out, e0, e1); run<1>("v_pk_fma_f32", out, e0, e1); run<2>("v_mul_f32", out, e0, e1); run<3>("v_pk_mul_f32", out, e0, e1);
    run<4>("v_add_f32", out, e0, e1); run<5>("v_pk_add_f32", out, e0, e1); run<6>("v_cvt_f32_ubyte0", out, e0, e1); run<7>("v_min_f32", out, e0, e1);
    run<8>("v_max3_f32", out, e0, e1); run<9>("v_mov_b32", out, e0, e1);
    run<10>("v_cndmask_b32 (sgpr pair)", out, e0, e1); run<11>("v_cmp_lt_f32 -> vcc", out, e0, e1); run<12>("v_cmp_lt_f32 -> sgpr pair", out, e0, e1);
    run<13>("v_rcp_f32", out, e0, e1); run<14>("v_sqrt_f32", out, e0, e1); run<15>("v_div_scale_f32", out, e0, e1); run<16>("v_div_fmas_f32", out, e0, e1);
    run<17>("v_div_fixup_f32", out, e0, e1); run<18>("v_mad_u64_u32", out, e0, e1); run<19>("v_xor_b32", out, e0, e1); run<20>("v_med3_f32", out, e0, e1);
    run<21>("v_mul_lo_u32", out, e0, e1); run<22>("v_mul_hi_u32", out, e0, e1); run<23>("v_cvt_f32_u32", out, e0, e1); run<24>("v_add_u32", out, e0, e1);
    run<25>("v_sub_f32", out, e0, e1); run<26>("v_pk_mov_b32", out, e0, e1);
    return 0;
}
