/*
 * philox.h — Philox4x32-10 counter-based RNG (Salmon et al., SC'11), the
 * replacement for the per-pixel XORWOW curandState of reference main.cu:108-117.
 *
 * Stream definition (identical in the oracle and in the kernels).  Every consumer of randomness
 * owns whole blocks addressed by what they are for, so no draw depends on how many draws another
 * consumer made (no serial stream to keep in step across divergent lanes):
 *   key     = (seed_lo, seed_hi)
 *   counter = (c0, purpose<<24 | bounce, pixel, sample)
 *   RT_RNG_CAMERA  c0=0     words 0,1 = pixel jitter u,v, word 2 = shutter time
 *   RT_RNG_DISK    c0=iter  words 0,1 = one lens-disk rejection candidate
 *   RT_RNG_SPHERE  c0=iter  words 0,1,2 = one unit-ball rejection candidate (x first)
 *   RT_RNG_GLASS   c0=0     word 0 = reflect-or-refract draw
 *   RT_RNG_MEDIUM  c0=slot  word 0 = free-flight draw of constant medium `slot`
 * Each block equals rocRAND's device engine at rocrand_init(seed, subsequence = pixel | sample<<32,
 * offset = 4*(c0 | c1<<32)).  Uniforms are in (0,1] like curand_uniform (log(xi), Volumes.cuh:40).
 */
#ifndef RT_PHILOX_H
#define RT_PHILOX_H
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

enum { RT_RNG_CAMERA = 0, RT_RNG_DISK = 1, RT_RNG_SPHERE = 2, RT_RNG_GLASS = 3, RT_RNG_MEDIUM = 4 };

struct rt_u32x4 { uint32_t x, y, z, w; };

RT_HD uint32_t rt_mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

RT_HD rt_u32x4 rt_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        /* one 32x32->64 product per multiplier (v_mad_u64_u32): hi and lo words in one instruction,
         * measured 222 vs 304 cycles per block per SIMD against separate mul_hi/mul_lo */
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += W0; k1 += W1;
    }
    rt_u32x4 o = {c0, c1, c2, c3};
    return o;
}

/* (0,1]: 2^-32 + v*2^-32, v rounded to float first (same as rocrand_uniform) */
RT_HD float rt_u32_to_uniform(uint32_t v)
{
    return 2.3283064e-10f + ((float)v * 2.3283064e-10f);
}

#endif
