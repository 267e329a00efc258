/*
 * rt_oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE; see rt_oracle.c).
 * All pointers are host memory.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>
#include "../include/rt_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rto_hit {
    int32_t hit;
    int32_t prim;        /* prim_id of the leaf object, -1 on miss */
    float   t, u, v;
    float   p[3];
    float   n[3];
    int32_t front_face;
    int32_t material;
} rto_hit;               /* 52 bytes */

typedef struct rto_scatter_out {
    int32_t ok;
    float   attenuation[3];
    float   origin[3];
    float   direction[3];
    float   time;
    float   emitted[3];
    uint32_t draws;      /* uniforms consumed */
} rto_scatter_out;       /* 60 bytes */

typedef struct rto_stats {
    uint64_t rays;          /* closest-hit queries = iterations of colour()'s loop */
    uint64_t samples;       /* camera paths */
    uint64_t prim_tests;    /* primitive hit() evaluations (a Box counts 6) */
    uint64_t nodes_visited; /* BVH nodes popped (use_bvh only) */
    uint64_t philox_blocks;
    uint64_t sphere_iters;  /* randomInUnitSphere rejection iterations */
    uint64_t rect_tests;    /* split of prim_tests by kind (a Box adds 6 rect tests) */
    uint64_t sphere_tests;
    uint64_t tri_tests;
    uint64_t rot_ops;       /* RotateY / Translate wrappers applied to a ray */
    uint64_t trans_ops;
} rto_stats;

void rto_trace_rays(const rt_flat_scene* s, int n, const float* rays7, float tmin, float tmax,
                    int use_bvh, rto_hit* out, rto_stats* st);
void rto_object_hit(const rt_flat_scene* s, int object, int n, const float* rays7, float tmin, float tmax,
                    rto_hit* out);
/* ... with the addressed draws of (seed, pixel, sample, bounce): a ConstantMedium object takes its free-flight draw */
void rto_object_hit_addressed(const rt_flat_scene* s, int object, int n, const float* rays7, float tmin, float tmax,
                              uint64_t seed, const uint32_t* pixel_sample_bounce3, rto_hit* out);
void rto_trace_primary(const rt_flat_scene* s, int W, int H, float s_off, float t_off, int use_bvh,
                       int32_t* prim, float* t_out);
void rto_texture_value(const rt_flat_scene* s, int texture, int n, const float* uvp5, float* rgb);
/* turbulence sum of an RT_TEX_NOISE table (RT_NOISE_TABLE_BYTES) at p */
float rto_noise_turb(const uint8_t* table, const float p[3], int depth);
/* xi: per call, xi_stride scripted uniforms consumed in the order the reference's scatter() draws them */
void rto_scatter(const rt_flat_scene* s, int material, int n, const float* ray_in7, const float* rec_in9,
                 const float* xi, int xi_stride, rto_scatter_out* out);
void rto_scatter_addressed(const rt_flat_scene* s, int n, const float* ray_in7, const float* rec_in9, const int32_t* material,
                           uint64_t seed, const uint32_t* pixel_sample_bounce3, rto_scatter_out* out);
void rto_camera_rays(const rt_flat_scene* s, int n, const float* st, uint64_t seed,
                     const uint32_t* pixel_sample, float* rays_out7);
void rto_uniforms(uint64_t seed, uint32_t pixel, uint32_t sample, int n, float* out);
void rto_philox_block(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void rto_render(const rt_flat_scene* s, int W, int H, int s_begin, int s_end, uint64_t seed, int max_depth,
                int background, int use_bvh, int n_threads, float* accum, rto_stats* st_out);
void rto_resolve(const float* accum, int n_pixels, int spp, int gamma, float* rgb);

#ifdef __cplusplus
}
#endif
#endif
