/*
 * rt_kernels.hip — HIP kernels for gfx950 and the device half of the C ABI (include/rt_api.h).
 *
 * render_kernel replaces the reference's `render` (main.cu:125-198).  The reference runs one
 * thread per pixel and lets a 32-lane warp wait for its slowest lane through numSamples x <=100
 * bounces.  Here every lane of a 64-lane wave is a small state machine: the moment its path ends it
 * starts its pixel's next sample, and when a run of samples of that pixel is done it moves on to its
 * pixel in the block's next tile, so the wave's closest-hit step runs with (almost) all lanes live to
 * the end; a run's sum stays in registers until the run is finished — no queues in HBM, no atomics,
 * a fixed summation order.
 * For scenes whose tables sit in the scalar cache (Cornell: 9 objects) the kernel is bound by
 * VALU issue; DESIGN.md has the numbers and why a queue-based wavefront would instead be bound
 * by its own queue traffic there.
 */
#include <hip/hip_runtime.h>
#include <chrono>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/rt_api.h"
#include "rt_device.h"
#include "rt_host.h"
#include "rt_wavefront.h"
#include "rt_wavefront2.h"
#include "rt_wavefront8.h"
#include "rt_wavefront4.h"
#include "rt_reduce.h"

/* feature sets the render kernel is compiled for (rt_host.h RT_F_*); a scene runs on the first one
 * that covers it.  Leaving out what a scene cannot contain keeps the kernel's register and
 * instruction-cache footprint down; the arithmetic of what remains is unchanged. */
#define RT_V_PLANAR       (RT_F_PLANAR | RT_F_XFORM)                                   /* Cornell box */
#define RT_V_PLANAR_MEDIA (RT_F_PLANAR | RT_F_XFORM | RT_F_MEDIA | RT_F_ISO)           /* Cornell smoke */
#define RT_V_LIST         (RT_F_ALL & ~RT_F_BVH)                                       /* any scene without a BVH */
#define RT_V_BVH_TRI      (RT_F_BVH | RT_F_TRI | RT_F_PLANAR)                          /* triangle mesh + area light */
#define RT_V_NOISE        (RT_F_ALL | RT_F_NOISE)                                      /* anything with a NoiseTexture */
#define RT_V_NOTRI        (RT_F_ALL & ~(RT_F_TRI | RT_F_GROUPLIST))                    /* wavefront form: the final scene */
#ifndef RT_WF_AUTO_ITEMS
#define RT_WF_AUTO_ITEMS 128          /* BVH items from which RT_INTEGRATOR_AUTO takes the wavefront form */
#endif
/* Slots (pixel x sample stream) a wavefront call is cut into; a slot's samples run one after the other.  In the one-kernel
 * form (rt_wavefront4.h) a slot costs 16 B of HBM (2 GB at this default) and its samples run in ONE lane, so when the
 * work list is empty the call waits for the longest slot: measured at 1080p, 1M triangles x 256 spp / final scene x 64 spp,
 * Grays/s: 8 M slots 3.24 / 2.60, 33 M 3.50 / 3.73, 128 M 3.54 / 4.24, 260 M 3.53 / 4.25.  In the two-kernel forms a slot is a
 * path in flight (88 B each: 11.3 GB) and every launch ends with a drain in which the waves wait for their last rays, the
 * more rays a launch carries the less that costs (round 1, same two scenes at 128 spp: 8 M 1.38 / 1.52, 32 M 1.62 / 1.75,
 * 128 M 1.81 / 1.89, 256 M 1.84 / 1.94).  The environment variable RT_WF_SLOTS overrides it; an allocation that fails is
 * retried at half the size. */
#ifndef RT_WF_SLOTS
#define RT_WF_SLOTS (1u << 27)
#endif
#define RT_N_COUNTERS 32        /* rays, samples, then diagnostic slots */
#define WF_MAX_POOLS 4
#define WF_POOL_COUNT_WORDS (9 * 32)   /* per pool: one 128-B line of queue lengths + legacy head, then a line per band head */
#ifndef WF_KERNEL_DEFAULT
#define WF_KERNEL_DEFAULT 4     /* what wavefront renders run unless RT_WF_KERNEL says otherwise (wf_pick_kernel): 4 = the whole path in one
                                 * persistent kernel (rt_wavefront4.h); 1 / 2 / 3 = intersect kernel + logic kernel, with the intersect
                                 * kernel of rt_wavefront.h / rt_wavefront2.h / rt_wavefront8.h */
#endif
#ifndef WF_POOLS_DEFAULT
#define WF_POOLS_DEFAULT 3
#endif
/* waves per SIMD each render-kernel variant is compiled for (its VGPR budget).  The kernels are VALU-issue bound and
 * more waves fill issue slots, until the spills cost more.  Measured on the Cornell variant (current build): 5 waves
 * (96 VGPRs, no spill) 25.4 Grays/s, 6 (80 VGPRs, 9 spilled, 40 B of scratch per lane) 26.6, 7 (72 VGPRs, 28
 * spilled) 26.0; an early build read 4 -> 19.7, 5 -> 20.6, 6 -> 21.1, 8 -> 18.1.  Cornell smoke: 4 -> 22.6, 5 -> 24.2,
 * 6 -> 24.8.  The spills show up as scratch traffic in the counters (profiles/), far from any bandwidth limit.
 * Since the device code is built without SLP vectorisation (Makefile KFLAGS) the Cornell variant needs 72 VGPRs and
 * spills nothing at 6 AND at 7 waves: 6 -> 28.0, 7 -> 28.6, 8 -> 28.1 Grays/s; Cornell smoke 25.5-25.7 at 6 either way. */
constexpr int rt_render_waves(uint32_t F)
{
#ifndef RT_PLANAR_WAVES
#define RT_PLANAR_WAVES 7
#endif
#ifndef RT_MEDIA_WAVES
#define RT_MEDIA_WAVES 6
#endif
    return (F & RT_F_BVH) ? 3 : ((F & ~(RT_F_PLANAR | RT_F_XFORM)) == 0 ? RT_PLANAR_WAVES : ((F & ~(RT_F_PLANAR | RT_F_XFORM | RT_F_MEDIA | RT_F_ISO)) == 0 ? RT_MEDIA_WAVES : 4));
}

/* Diagnostic build only (-DRT_STAMPS, tools/diag): shader-clock time of each section of the render
 * loop, summed per wave into counters[4 + section].  No stamp executes in the shipped library. */
#ifdef RT_STAMPS
#define RT_STAMP_DECL unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}; unsigned long long st_last = __builtin_amdgcn_s_memtime();
#define RT_STAMP(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long now_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); st_acc[i] += now_ - st_last; st_last = now_; }
#define RT_STAMP_FLUSH if ((threadIdx.x & 63) == 0) { for (int q_ = 0; q_ < 6; ++q_) atomicAdd(&counters[4 + q_], st_acc[q_]); }
#else
#define RT_STAMP_DECL
#define RT_STAMP(i)
#define RT_STAMP_FLUSH
#endif

/* Scratch of rt_render_accumulate_noalloc (include/rt_api.h): the per-run partial frames of the megakernel, for callers
 * that cannot allocate.  One array per GPU the library runs on, held until the process ends.  RT_NOALLOC_FLOATS =
 * 16 runs x 512 x 512 pixels x 3 floats = 48 MiB.  In host code the array's name is only a stand-in: the device address
 * is looked up once per device (noalloc_scratch_for) and only that pointer is handed to kernels. */
#define RT_NOALLOC_FLOATS (16u * 512u * 512u * 3u)
#define RT_MAX_DEVICES 64
static __device__ __align__(256) float rt_noalloc_scratch[RT_NOALLOC_FLOATS];

/* ------------------------------------------------------------------ kernels -- */
template <uint32_t F>
__global__ __launch_bounds__(RT_BLOCK) void trace_rays_kernel(DScene S, int n, const float* __restrict__ rays,
                                                               float tmin, float tmax, rt_hit_record* __restrict__ out)
{
    extern __shared__ int stack_mem[];                 /* RT_F_BVH: stack_depth x RT_BLOCK ints */
    int* stack = stack_mem + ((F & RT_F_BVH) ? threadIdx.x : 0);
    int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float* q = rays + 7 * (size_t)i;
    DRay r;
    r.o = mk(q[0], q[1], q[2]);
    r.d = mk(q[3], q[4], q[5]);
    r.time = q[6];
    HitLite hl = closest_hit<F>(S, r, tmin, tmax, nullptr, 0u, stack);
    rt_hit_record o;
    memset(&o, 0, sizeof o);
    o.prim = -1;
    if (hl.obj >= 0) {
        HitRec h;
        rebuild<F, true>(S, r, hl, tmin, h);
        o.hit = 1; o.prim = h.prim_id; o.t = h.t; o.u = h.u; o.v = h.v;
        o.p[0] = h.p.x; o.p[1] = h.p.y; o.p[2] = h.p.z;
        o.n[0] = h.n.x; o.n[1] = h.n.y; o.n[2] = h.n.z;
        o.front_face = h.front; o.material = h.material;
    }
    out[i] = o;
}

/* Material::scatter + emitted for caller-supplied (ray, record, material, RNG address): the shading parity hook.  Same
 * device code as the integrators: the wave-wide unit-ball draw, then scatter<F>. */
template <uint32_t F>
__global__ __launch_bounds__(RT_BLOCK) void shade_rays_kernel(DScene S, int n, const float* __restrict__ rays, const float* __restrict__ recs,
                                                               const int32_t* __restrict__ materials, int n_materials, const uint32_t* __restrict__ psb,
                                                               uint32_t k0, uint32_t k1, rt_scatter_record* __restrict__ out)
{
    __shared__ int slot_mem[RT_BLOCK];
    int* slots = slot_mem + (threadIdx.x & ~63u);
    const int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    const bool active = i < n;
    DRay r = {mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 1.f), 0.f};
    HitRec rec;
    rec.t = 0.f; rec.u = 0.f; rec.v = 0.f; rec.p = mk(0.f, 0.f, 0.f); rec.n = mk(0.f, 0.f, 0.f);
    rec.material = 0; rec.prim_id = -1; rec.front = 0;
    PathRng g;
    rng_start(g, k0, k1, 0u, 0u);
    uint32_t bounce = 0;
    int mtype = RT_MAT_LIGHT;
    bool in_range = false;
    if (active) {
        const float* q = rays + 7 * (size_t)i;
        const float* h = recs + 9 * (size_t)i;
        r.o = mk(q[0], q[1], q[2]); r.d = mk(q[3], q[4], q[5]); r.time = q[6];
        rec.t = h[0]; rec.u = h[1]; rec.v = h[2];
        rec.p = mk(h[3], h[4], h[5]); rec.n = mk(h[6], h[7], h[8]);
        rec.material = materials[i];
        rng_start(g, k0, k1, psb[3 * (size_t)i], psb[3 * (size_t)i + 1]);
        bounce = psb[3 * (size_t)i + 2];
        in_range = rec.material >= 0 && rec.material < n_materials;      /* the indices are the caller's: never followed unchecked */
        if (in_range) mtype = S.mats[rec.material].type;
    }
    /* a lane with a bad index stays in the wave-wide draw (as a lane that needs nothing) and reports ok = -1 */
    const bool need_ball = active && in_range && mtype != RT_MAT_LIGHT && mtype != RT_MAT_GLASS;
    const V3 ball = random_in_unit_sphere_wave(g, bounce, need_ball, slots);
    if (!active) return;
    if (!in_range) {
        rt_scatter_record bad;
        memset(&bad, 0, sizeof bad);
        bad.ok = -1;
        out[i] = bad;
        return;
    }
    V3 attenuation, emitted;
    DRay scattered;
    const bool ok = scatter<F>(S, r, rec, g, bounce, ball, attenuation, scattered, emitted);
    rt_scatter_record o;
    o.ok = ok ? 1 : 0;
    o.attenuation[0] = attenuation.x; o.attenuation[1] = attenuation.y; o.attenuation[2] = attenuation.z;
    o.origin[0] = scattered.o.x; o.origin[1] = scattered.o.y; o.origin[2] = scattered.o.z;
    o.direction[0] = scattered.d.x; o.direction[1] = scattered.d.y; o.direction[2] = scattered.d.z;
    o.time = scattered.time;
    o.emitted[0] = emitted.x; o.emitted[1] = emitted.y; o.emitted[2] = emitted.z;
    o.pad = 0u;
    out[i] = o;
}

template <uint32_t F>
__global__ __launch_bounds__(RT_BLOCK) void texture_values_kernel(DScene S, int texture, int n, const float* __restrict__ uvp, float* __restrict__ rgb)
{
    const int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float* q = uvp + 5 * (size_t)i;
    const V3 c = texture_value<F>(S, texture, q[0], q[1], mk(q[2], q[3], q[4]));
    rgb[3 * (size_t)i] = c.x; rgb[3 * (size_t)i + 1] = c.y; rgb[3 * (size_t)i + 2] = c.z;
}

/* lane -> pixel inside a 16x16 block tile; each wave covers an 8x8 sub-tile */
__device__ __forceinline__ void tile_pixel(int W, int tile, int tid, int& i, int& j)
{
    int tiles_x = (W + 15) >> 4;
    int tx = tile % tiles_x, ty = tile / tiles_x;
    int lx = (tid & 7) | (((tid >> 6) & 1) << 3);
    int ly = ((tid >> 3) & 7) | (((tid >> 7) & 1) << 3);
    i = tx * 16 + lx;
    j = ty * 16 + ly;
}

/* blocks b and b+8 share an XCD (and its L2): hand each XCD a contiguous run of tiles */
__device__ __forceinline__ int xcd_tile(int b, int nb)
{
    int q = nb >> 3, r = nb & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

template <uint32_t F>
__global__ __launch_bounds__(RT_BLOCK) void trace_primary_kernel(DScene S, int W, int H, float s_off, float t_off,
                                                                  int32_t* __restrict__ prim, float* __restrict__ t_out)
{
    extern __shared__ int stack_mem[];
    int* stack = stack_mem + ((F & RT_F_BVH) ? threadIdx.x : 0);
    int i, j;
    tile_pixel(W, xcd_tile(blockIdx.x, gridDim.x), threadIdx.x, i, j);
    if (i >= W || j >= H) return;
    float u = float(i + s_off) / float(W);
    float v = float(j + t_off) / float(H);
    DRay r = primary_ray(S.cam, u, v);
    HitLite hl = closest_hit<F>(S, r, .001f, FLT_MAX, nullptr, 0u, stack);
    size_t px = (size_t)j * W + i;
    prim[px] = hl.obj >= 0 ? S.objs[hl.obj].prim_id : -1;
    t_out[px] = hl.obj >= 0 ? hl.t : 0.f;
}

template <uint32_t F, int BG>
__global__ __launch_bounds__(RT_BLOCK, rt_render_waves(F)) void render_kernel(DScene S, int W, int H, int4 win, uint32_t s_first, uint32_t s_last,
                                                           uint32_t k0, uint32_t k1, int max_depth,
                                                           float* __restrict__ accum,
                                                           unsigned long long* __restrict__ counters,
                                                           uint32_t per_group, float* __restrict__ partial,
                                                           uint32_t n_groups, uint32_t tile_step)
{
    /* The call's sample range is cut into n_groups runs of per_group samples (rt_render_accumulate).  Block B
     * renders run k of tile (B + k * tile_step) mod n_tiles for k = 0 .. n_groups-1, i.e. one run each of n_groups
     * tiles spread over the frame, and every LANE walks through them on its own: pixel `lane` of the first tile for
     * run 0, then pixel `lane` of the next tile for run 1, without waiting for the other lanes.  A wave therefore
     * ends when its slowest lane has done several pixels' worth of work, not when the slowest pixel of one tile is
     * done: tiles that straddle the light or a silhouette hold pixels whose paths differ severalfold in length.
     * For each k the map B -> tile is a bijection, so every (tile, run) is rendered exactly once.  With more than one
     * run the per-pixel sums go to `partial`, one frame per run, and combine_kernel adds them to accum in run order. */
    extern __shared__ int stack_mem[];                 /* RT_BLOCK ints, then RT_F_BVH: stack_depth x RT_BLOCK */
    int* slots = stack_mem + (threadIdx.x & ~63u);     /* 64 ints per wave for the wave-wide ball draw */
    int* stack = stack_mem + RT_BLOCK + ((F & RT_F_BVH) ? threadIdx.x : 0);
    /* the grid tiles the window win = (x0, y0, width, height) of the W x H frame; pixel coordinates and the
     * pixel index (RNG address, accum row) are the frame's, so a window renders what the full frame would */
    uint32_t k = 0;                                    /* the run this lane is on */
    uint32_t s;
    PathRng g;
    rng_start(g, k0, k1, 0u, 0u);                      /* g.pixel = the pixel's index in the frame, ~0u outside the window */
    /* depth 0: colour()'s loop body never runs and every sample is black (main.cu:53,71) */
#define RT_RUN_END(k_) (s_last - s_first - (k_) * per_group < per_group ? s_last : s_first + ((k_) + 1u) * per_group)
#define RT_ENTER_RUN                                                                                      \
    {                                                                                                     \
        int i_, j_;                                                                                       \
        tile_pixel(win.z, (int)((blockIdx.x + k * tile_step) % gridDim.x), threadIdx.x, i_, j_);          \
        const bool valid_ = i_ < win.z && j_ < win.w;                                                     \
        g.pixel = valid_ ? (uint32_t)((j_ + win.y) * W + i_ + win.x) : 0xffffffffu;                       \
        s = (valid_ && max_depth > 0) ? s_first + k * per_group : RT_RUN_END(k);                          \
    }
#define RT_FLUSH_RUN                                                                                      \
    if (g.pixel != 0xffffffffu) {                                                                         \
        if (partial) {                                                                                    \
            float* a_ = partial + 3 * ((size_t)k * (size_t)W * (size_t)H + (size_t)g.pixel);              \
            a_[0] = sum.x; a_[1] = sum.y; a_[2] = sum.z;                                                  \
        } else {                                                                                          \
            float* a_ = accum + 3 * (size_t)g.pixel;                                                      \
            a_[0] += sum.x; a_[1] += sum.y; a_[2] += sum.z;                                               \
        }                                                                                                 \
    }
    RT_ENTER_RUN

    V3 sum = mk(0.f, 0.f, 0.f);
    bool alive = false;
    DRay ray = {mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 1.f), 0.f};
    V3 att = mk(1.f, 1.f, 1.f), emi = mk(0.f, 0.f, 0.f);
    int depth = 0;
    uint32_t n_rays = 0;

    RT_STAMP_DECL
    while (true) {
        RT_STAMP(5)
        if (!alive && s >= RT_RUN_END(k) && k + 1 < n_groups) {
            /* this lane's run is done: hand in its sum and go on to the lane's pixel in the block's next tile */
            RT_FLUSH_RUN
            ++k;
            RT_ENTER_RUN
            sum = mk(0.f, 0.f, 0.f);
        }
        if (!alive && s < RT_RUN_END(k)) {
            /* main.cu:185-188: jittered pixel position, camera ray, fresh path */
            g.sample = s;
            rt_u32x4 cw = rng_block(g, RT_RNG_CAMERA, 0u, 0u);
            const uint32_t j = g.pixel / (uint32_t)W, i = g.pixel - j * (uint32_t)W;
            float u = float((int)i + rt_u32_to_uniform(cw.x)) / float(W);
            float v = float((int)j + rt_u32_to_uniform(cw.y)) / float(H);
            ray = generate_ray<F>(cam_fetch(S), u, v, rt_u32_to_uniform(cw.z), g);
            att = mk(1.f, 1.f, 1.f);
            emi = mk(0.f, 0.f, 0.f);
            depth = 0;
            alive = true;
            ++s;
        }
        RT_STAMP(0)
        if (__ballot(alive || k + 1 < n_groups) == 0ull) break;
        /* one iteration of colour() / colourUnlit(), main.cu:53-70 / :77-98, for every live lane.  The
         * body is kept flat (flags, not nested branches) because the unit-ball draw in the middle is
         * made by all 64 lanes together. */
        HitLite hl = {FLT_MAX, -1, -1, 0};
        if (alive) {
            hl = closest_hit<F>(S, ray, .001f, FLT_MAX, &g, (uint32_t)depth, stack);
            ++n_rays;
        }
        RT_STAMP(1)
        const bool hit = alive && hl.obj >= 0;
        HitRec rec;
        rec.t = 0.f; rec.u = 0.f; rec.v = 0.f; rec.p = mk(0.f, 0.f, 0.f); rec.n = mk(0.f, 0.f, 0.f);
        rec.material = 0; rec.prim_id = -1; rec.front = 0;
        int mtype = RT_MAT_LIGHT;
        if (hit) {
            rebuild<F, false>(S, ray, hl, .001f, rec);
            mtype = S.mats[rec.material].type;
        }
        RT_STAMP(2)
        const bool need_ball = hit && mtype != RT_MAT_LIGHT && mtype != RT_MAT_GLASS;
        const V3 ball = random_in_unit_sphere_wave(g, (uint32_t)depth, need_ball, slots);
        RT_STAMP(3)
        if (alive) {
            V3 result;
            bool done;
            if (hit) {
                V3 attenuation, emitted;
                DRay scattered;
                if (scatter<F>(S, ray, rec, g, (uint32_t)depth, ball, attenuation, scattered, emitted)) {
                    att = att * attenuation;
                    if (BG == RT_BG_BLACK) emi = emi + emitted * att;
                    ray = scattered;
                    done = false;
                    result = mk(0.f, 0.f, 0.f);
                } else {
                    result = BG == RT_BG_BLACK ? emi + emitted * att : mk(0.f, 0.f, 0.f);
                    done = true;
                }
            } else {
                if (BG == RT_BG_BLACK) result = emi;
                else {
                    V3 ud = unit(ray.d);
                    float t = .5f * (ud.y + 1.f);
                    V3 c = (1.f - t) * mk(1.f, 1.f, 1.f) + t * mk(.5f, .7f, 1.f);
                    result = att * c;
                }
                done = true;
            }
            ++depth;
            if (!done && depth >= max_depth) {
                result = BG == RT_BG_BLACK ? emi : mk(0.f, 0.f, 0.f);
                done = true;
            }
            if (done) {
                sum = sum + result;
                alive = false;
            }
        }
    }
    RT_STAMP(4)
    RT_STAMP_FLUSH
    RT_FLUSH_RUN
#undef RT_FLUSH_RUN
#undef RT_ENTER_RUN
#undef RT_RUN_END
    /* ray / sample counters: one atomic per wave; every sample of the window is rendered (black when max_depth is 0) */
    unsigned long long r = n_rays, p = (blockIdx.x == 0 && threadIdx.x == 0) ? (unsigned long long)win.z * win.w * (s_last - s_first) : 0ull;
    for (int off = 32; off > 0; off >>= 1) {
        r += __shfl_down(r, off);
        p += __shfl_down(p, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counters[0], r);
        atomicAdd(&counters[1], p);
    }
}

/* accum[window] += partial[0] + partial[1] + ... : the sample groups of render_kernel, in group order */
__global__ __launch_bounds__(RT_BLOCK) void combine_kernel(float* __restrict__ accum, const float* __restrict__ partial, int W, int H, int4 win, int groups)
{
    const size_t i = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x;         /* float index inside the window */
    const size_t row = (size_t)win.z * 3;
    if (i >= row * (size_t)win.w) return;
    const size_t at = ((size_t)(win.y + (int)(i / row)) * (size_t)W + (size_t)win.x) * 3 + i % row;
    float v = accum[at];
    for (int g = 0; g < groups; ++g) v += partial[(size_t)g * (size_t)W * (size_t)H * 3 + at];
    accum[at] = v;
}

__global__ __launch_bounds__(RT_BLOCK) void resolve_kernel(const float* __restrict__ accum, size_t n, float inv_spp,
                                                           int gamma, float* __restrict__ rgb)
{
    size_t i = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    float c = accum[i] * inv_spp;          /* outCol /= float(numSamples): multiply by reciprocal */
    rgb[i] = gamma ? sqrtf(c) : c;
}

/* --------------------------------------------------------------------- host -- */
#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            rt_set_error(std::string("HIP error ") + hipGetErrorString(e_) + " at " #expr);    \
            return RT_ERR_HIP;                                                                 \
        }                                                                                      \
    } while (0)

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool have_scene = false;
    bool has_bvh = false;
    int n_bvh_items = 0;
    int n_textures = 0, n_materials = 0, n_objects = 0;
    uint32_t features = 0;    /* RT_F_* of the uploaded scene */
    int integrator = RT_INTEGRATOR_AUTO;
    int last_integrator = 0;          /* form the latest render call ran in (rt_counters) */
    int last_wf_pools = 0;            /* pools the latest wavefront call ran as (rt_counters) */
    int last_wf_kernel = 0;           /* and its form (wf_pick_kernel) */
    /* wavefront path state (rt_wavefront.h), allocated on first use and kept */
    WfState wf = {};
    void* wf_mem = nullptr;
    size_t wf_slots_cap = 0;
    bool wf_cap_full = false;       /* wf_mem holds the full path state (two-kernel forms, rt_trace_rays_wavefront), not only the slot sums */
    size_t wf_slots_target = RT_WF_SLOTS;   /* paths to keep in flight */
    uint32_t* wf_host_count = nullptr;   /* pinned: 4 words of readback per pool, then the pools' first queue lengths */
    hipStream_t wf_streams[WF_MAX_POOLS] = {};        /* the pools' streams: found to run side by side (wf_pick_streams) */
    int wf_n_streams = -1;                            /* how many of them; -1 = not looked for yet */
    hipEvent_t wf_ev[WF_MAX_POOLS + 1] = {};
    int stack_depth = 0;      /* LDS stack entries per lane the uploaded BVHs need */
    int stack4 = 1;           /* entries a traversal of the 4-wide trees can need (3 per level) */
    int depth8 = 0;           /* deepest chain of 8-wide nodes; 0 = the scene has no 8-wide trees (a leaf of more than two items) */
    size_t n_nodes8 = 0;
    float box_abs = 0.f;
    DScene ds;
    void* d_objs = nullptr;  void* d_ops = nullptr;   void* d_mats = nullptr; void* d_texs = nullptr;
    void* d_image = nullptr; void* d_nodes = nullptr; void* d_leafs = nullptr; void* d_prog = nullptr; void* d_nodes4 = nullptr; void* d_cam = nullptr; void* d_nodes8 = nullptr; void* d_leafs8 = nullptr;
    float* d_partial = nullptr; size_t partial_floats = 0;    /* per-group frames of the megakernel's sample groups */
    float* noalloc_scratch = nullptr;  /* this device's copy of rt_noalloc_scratch, looked up at rt_ctx_create */
    void* d_prog_wf = nullptr; void* d_wf_groups = nullptr;   /* the world list in the wavefront integrator's form */
    int n_wf_groups = 0, n_wf_plain = 0;
    unsigned long long* d_counters = nullptr;
    /* render launches whose event pair has not been read back yet are timed lazily */
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double render_ms = 0.0, last_ms = 0.0;
    uint32_t launches = 0;
};

static void free_scene(rt_ctx* c)
{
    void** ptrs[] = {&c->d_objs, &c->d_ops, &c->d_mats, &c->d_texs, &c->d_image, &c->d_nodes, &c->d_leafs, &c->d_prog, &c->d_nodes4, &c->d_cam, &c->d_nodes8, &c->d_leafs8,
                     &c->d_prog_wf, &c->d_wf_groups};
    for (void** p : ptrs) { if (*p) (void)hipFree(*p); *p = nullptr; }
    c->have_scene = false;
}

/* device address of rt_noalloc_scratch on `device`, looked up once per device with that device current
 * (every GPU holds its own copy of the array at its own address) */
static float* g_noalloc_ptr[RT_MAX_DEVICES];
static std::mutex g_noalloc_mutex;
static int noalloc_scratch_for(int device, float** out)
{
    if (device < 0 || device >= RT_MAX_DEVICES) { rt_set_error("no-allocation scratch: device index beyond the table"); return RT_ERR_ARG; }
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (!g_noalloc_ptr[device]) {
        HIPCHK(hipSetDevice(device));
        float* p = nullptr;
        HIPCHK(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(rt_noalloc_scratch)));
        if (!p) { rt_set_error("no-allocation scratch: hipGetSymbolAddress returned null"); return RT_ERR_HIP; }
        g_noalloc_ptr[device] = p;
    }
    if (out) *out = g_noalloc_ptr[device];
    return RT_OK;
}

extern "C" int rt_ctx_create(int device, void* stream, rt_ctx** out)
{
    if (!out) { rt_set_error("rt_ctx_create: null out"); return RT_ERR_ARG; }
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) { rt_set_error("rt_ctx_create: no such device"); return RT_ERR_ARG; }
    HIPCHK(hipSetDevice(device));
    float* scratch = nullptr;
    { int src = noalloc_scratch_for(device, &scratch); if (src) return src; }
    rt_ctx* c = new rt_ctx();
    c->device = device;
    c->noalloc_scratch = scratch;
    if (const char* e = std::getenv("RT_WF_SLOTS")) { const long long v = std::atoll(e); if (v > 0) c->wf_slots_target = (size_t)v; }
    if (stream) c->stream = (hipStream_t)stream;
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; rt_set_error("hipStreamCreate failed"); return RT_ERR_HIP; }
        c->own_stream = true;
    }
    if (hipMalloc((void**)&c->d_counters, RT_N_COUNTERS * sizeof(unsigned long long)) != hipSuccess ||
        hipMemsetAsync(c->d_counters, 0, RT_N_COUNTERS * sizeof(unsigned long long), c->stream) != hipSuccess) {
        rt_ctx_destroy(c);
        rt_set_error("counter allocation failed");
        return RT_ERR_HIP;
    }
    *out = c;
    return RT_OK;
}

extern "C" void rt_ctx_destroy(rt_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    free_scene(c);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->wf_mem) (void)hipFree(c->wf_mem);
    if (c->d_partial) (void)hipFree(c->d_partial);
    if (c->wf_host_count) (void)hipHostFree(c->wf_host_count);
    for (hipStream_t st : c->wf_streams) if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t e : c->wf_ev) if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

template <class T>
static int upload(rt_ctx* c, void** dst, const T* src, size_t n)
{
    size_t bytes = (n ? n : 1) * sizeof(T);
    HIPCHK(hipMalloc(dst, bytes));
    if (n) HIPCHK(hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return RT_OK;
}

extern "C" int rt_scene_upload(rt_ctx* c, const rt_flat_scene* s)
{
    if (!c || !s) { rt_set_error("rt_scene_upload: null argument"); return RT_ERR_ARG; }
    std::string why;
    int bvh_depth = 0;
    if (!rt_validate_scene(s, why, &bvh_depth)) { rt_set_error("rt_scene_upload: invalid scene: " + why); return RT_ERR_ARG; }
    std::vector<DObj> objs((size_t)s->n_objects);
    rt_pack_objects(s, objs.data());
    std::vector<DNode> nodes;
    std::vector<DObj> leafs;
    int stack_depth = 0;
    if (!rt_pack_bvh(s, objs, nodes, leafs, &stack_depth, why)) { rt_set_error("rt_scene_upload: " + why); return RT_ERR_UNSUPPORTED; }
    std::vector<DNode4> nodes4;
    int depth4 = 0;
    rt_pack_bvh4(s, objs, nodes4, &depth4);
    std::vector<DNode8> nodes8;
    std::vector<DLeaf> leafs8;
    int depth8 = 0;
    float box_abs = 0.f;
    if (!rt_pack_bvh8(s, objs, nodes8, leafs8, &depth8, &box_abs)) { nodes8.clear(); leafs8.clear(); depth8 = 0; }
    std::vector<DProg> prog, prog_wf;
    std::vector<int> wf_groups;
    rt_pack_program(s, prog, &prog_wf, &wf_groups);
    if (stack_depth > RT_MAX_STACK) { rt_set_error("rt_scene_upload: BVH deeper than the traversal stack (64)"); return RT_ERR_UNSUPPORTED; }
    /* the path kernel addresses a node row by a 32-bit byte offset from the table's base (rt_wavefront4.h) */
    if (nodes4.size() >= ((size_t)1 << 24)) { rt_set_error("rt_scene_upload: more than 2^24 four-wide BVH nodes"); return RT_ERR_UNSUPPORTED; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    free_scene(c);

    int rc;
    if ((rc = upload(c, &c->d_objs, objs.data(), objs.size()))) return rc;
    if ((rc = upload(c, &c->d_ops, s->ops, (size_t)s->n_ops))) return rc;
    /* device copy of the material table: pad = 1 when scatter()/emitted() read the record's u,v,
     * i.e. an ImageTexture is reachable from the material's texture (Metal passes 0,0: Material.cuh:76) */
    std::vector<rt_material> mats(s->materials, s->materials + s->n_materials);
    for (rt_material& m : mats) {
        m.pad = 0;
        if (m.type == RT_MAT_GLASS || m.type == RT_MAT_METAL || m.texture < 0) continue;
        std::vector<int> todo(1, m.texture);
        for (size_t guard = 0; !todo.empty() && guard < 64; ++guard) {
            const rt_texture& tx = s->textures[todo.back()];
            todo.pop_back();
            if (tx.type == RT_TEX_IMAGE) m.pad = 1;
            else if (tx.type == RT_TEX_CHECKER) { todo.push_back(tx.even); todo.push_back(tx.odd); }
        }
        if (!todo.empty()) m.pad = 1;   /* very deep checker nest: be safe */
    }
    if ((rc = upload(c, &c->d_mats, mats.data(), mats.size()))) return rc;
    if ((rc = upload(c, &c->d_texs, s->textures, (size_t)s->n_textures))) return rc;
    if ((rc = upload(c, &c->d_image, s->image_bytes, (size_t)s->n_image_bytes))) return rc;
    if ((rc = upload(c, &c->d_nodes, nodes.data(), nodes.size()))) return rc;
    if ((rc = upload(c, &c->d_leafs, leafs.data(), leafs.size()))) return rc;
    if ((rc = upload(c, &c->d_prog, prog.data(), prog.size()))) return rc;
    if ((rc = upload(c, &c->d_nodes4, nodes4.data(), nodes4.size()))) return rc;
    if ((rc = upload(c, &c->d_cam, &s->camera, (size_t)1))) return rc;
    if ((rc = upload(c, &c->d_nodes8, nodes8.data(), nodes8.size()))) return rc;
    if ((rc = upload(c, &c->d_leafs8, leafs8.data(), leafs8.size()))) return rc;
    if ((rc = upload(c, &c->d_prog_wf, prog_wf.data(), prog_wf.size()))) return rc;
    if ((rc = upload(c, &c->d_wf_groups, wf_groups.data(), wf_groups.size()))) return rc;
    c->n_wf_groups = (int)wf_groups.size();
    c->n_wf_plain = (int)prog_wf.size() - 2;      /* without the two terminators */
    HIPCHK(hipStreamSynchronize(c->stream)); /* objs, nodes, nodes4, leafs, the programs and mats are locals */
    c->ds.objs = (const DObj*)c->d_objs;
    c->ds.ops = (const rt_xform_op*)c->d_ops;
    c->ds.mats = (const rt_material*)c->d_mats;
    c->ds.texs = (const rt_texture*)c->d_texs;
    c->ds.image = (const uint8_t*)c->d_image;
    c->ds.nodes = (const DNode*)c->d_nodes;
    c->ds.leafs = (const DObj*)c->d_leafs;
    c->ds.prog = (const DProg*)c->d_prog;
    c->ds.nodes4 = (const DNode4*)c->d_nodes4;
    c->ds.cam_ptr = (const rt_camera*)c->d_cam;
    c->ds.nodes8 = (const DNode8*)c->d_nodes8;
    c->ds.leafs8 = (const DLeaf*)c->d_leafs8;
    c->depth8 = depth8; c->n_nodes8 = nodes8.size(); c->box_abs = box_abs;
    c->stack4 = 3 * depth4 + 1;
    c->features = rt_scene_features(s);
    c->ds.world = s->world;
    c->ds.n_media = s->n_media;
    c->ds.cam = s->camera;
    c->has_bvh = s->n_bvh_nodes > 0;
    c->n_bvh_items = s->n_bvh_items;
    c->n_textures = s->n_textures;
    c->n_materials = s->n_materials;
    c->n_objects = s->n_objects;
    c->stack_depth = stack_depth < 1 ? 1 : stack_depth;
    c->have_scene = true;
    return RT_OK;
}

static size_t stack_bytes(const rt_ctx* c) { return c->has_bvh ? (size_t)c->stack_depth * RT_BLOCK * sizeof(int) : 0; }

static int need_scene(rt_ctx* c, const char* who)
{
    if (!c) { rt_set_error(std::string(who) + ": null context"); return RT_ERR_ARG; }
    if (!c->have_scene) { rt_set_error(std::string(who) + ": no scene uploaded"); return RT_ERR_NO_SCENE; }
    return RT_OK;
}

extern "C" int rt_trace_rays(rt_ctx* c, int n, const float* rays, float tmin, float tmax, rt_hit_record* out)
{
    int rc = need_scene(c, "rt_trace_rays");
    if (rc) return rc;
    if (n < 0 || (n && (!rays || !out))) { rt_set_error("rt_trace_rays: bad arguments"); return RT_ERR_ARG; }
    if (n == 0) return RT_OK;
    HIPCHK(hipSetDevice(c->device));
    dim3 grid((unsigned)((n + RT_BLOCK - 1) / RT_BLOCK));
    if (c->has_bvh) trace_rays_kernel<RT_F_ALL><<<grid, RT_BLOCK, stack_bytes(c), c->stream>>>(c->ds, n, rays, tmin, tmax, out);
    else trace_rays_kernel<RT_V_LIST><<<grid, RT_BLOCK, 0, c->stream>>>(c->ds, n, rays, tmin, tmax, out);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

extern "C" int rt_shade_rays(rt_ctx* c, int n, const float* rays, const float* recs, const int32_t* materials, const uint32_t* psb,
                             uint64_t seed, rt_scatter_record* out)
{
    int rc = need_scene(c, "rt_shade_rays");
    if (rc) return rc;
    if (n < 0 || (n && (!rays || !recs || !materials || !psb || !out))) { rt_set_error("rt_shade_rays: bad arguments"); return RT_ERR_ARG; }
    if (n == 0) return RT_OK;
    HIPCHK(hipSetDevice(c->device));
    /* material indices come from the caller's device memory: the kernel checks them against the scene's material count
     * and reports ok = -1 for one out of range */
    shade_rays_kernel<RT_V_NOISE><<<dim3((unsigned)((n + RT_BLOCK - 1) / RT_BLOCK)), RT_BLOCK, 0, c->stream>>>(c->ds, n, rays, recs, materials, c->n_materials, psb,
                                                                                                          (uint32_t)seed, (uint32_t)(seed >> 32), out);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

extern "C" int rt_texture_values(rt_ctx* c, int texture, int n, const float* uvp, float* rgb)
{
    int rc = need_scene(c, "rt_texture_values");
    if (rc) return rc;
    if (n < 0 || texture < 0 || texture >= c->n_textures || (n && (!uvp || !rgb))) { rt_set_error("rt_texture_values: bad arguments"); return RT_ERR_ARG; }
    if (n == 0) return RT_OK;
    HIPCHK(hipSetDevice(c->device));
    texture_values_kernel<RT_V_NOISE><<<dim3((unsigned)((n + RT_BLOCK - 1) / RT_BLOCK)), RT_BLOCK, 0, c->stream>>>(c->ds, texture, n, uvp, rgb);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

static bool frame_ok(int w, int h) { return w > 0 && h > 0 && (long long)w * h <= (1ll << 30); }
static unsigned n_tiles(int w, int h) { return (unsigned)(((w + 15) / 16) * ((h + 15) / 16)); }

extern "C" int rt_trace_primary(rt_ctx* c, int w, int h, float s_off, float t_off, int32_t* prim, float* t)
{
    int rc = need_scene(c, "rt_trace_primary");
    if (rc) return rc;
    if (!frame_ok(w, h) || !prim || !t) { rt_set_error("rt_trace_primary: bad arguments"); return RT_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    dim3 grid(n_tiles(w, h));
    if (c->has_bvh) trace_primary_kernel<RT_F_ALL><<<grid, RT_BLOCK, stack_bytes(c), c->stream>>>(c->ds, w, h, s_off, t_off, prim, t);
    else trace_primary_kernel<RT_V_LIST><<<grid, RT_BLOCK, 0, c->stream>>>(c->ds, w, h, s_off, t_off, prim, t);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

static int drain_events(rt_ctx* c, size_t keep_newest = 0);
/* a caller that renders forever without reading the counters must not pile up events */
static int note_render(rt_ctx* c, hipEvent_t e0, hipEvent_t e1)
{
    c->pending.push_back({e0, e1});
    return c->pending.size() > 64 ? drain_events(c, 32) : RT_OK;
}

/* ------------------------------------------------------------ wavefront host -- */
/* Streams for the pools.  A process has few hardware queues (GPU_MAX_HW_QUEUES, 4 here) and the runtime spreads its
 * streams over them; two streams on one queue run their kernels one after the other, and pools that do not overlap
 * are worse than one pool.  Which streams share a queue cannot be asked, so it is measured once per context: a kernel
 * that waits ~0.3 ms of wall clock is launched on a set of streams at once, and a candidate joins the set if the whole
 * set still finishes in about one wait. */
__global__ void wf_spin_kernel(unsigned long long ticks)
{
    const long long t0 = wall_clock64();                /* constant-rate counter, 100 MHz */
    for (int i = 0; i < 100000; ++i) {                   /* bounded whatever the counter does */
        if ((unsigned long long)(wall_clock64() - t0) >= ticks) break;
        __builtin_amdgcn_s_sleep(16);
    }
}
static int wf_pick_streams(rt_ctx* c)
{
    c->wf_n_streams = 0;
    const unsigned long long ticks = 30000ull;          /* 0.3 ms */
    auto timed = [&](hipStream_t extra, double& ms) -> int {
        for (int i = 0; i < c->wf_n_streams; ++i) HIPCHK(hipStreamSynchronize(c->wf_streams[i]));
        HIPCHK(hipStreamSynchronize(extra));
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < c->wf_n_streams; ++i) wf_spin_kernel<<<1, 64, 0, c->wf_streams[i]>>>(ticks);
        wf_spin_kernel<<<1, 64, 0, extra>>>(ticks);
        HIPCHK(hipGetLastError());
        for (int i = 0; i < c->wf_n_streams; ++i) HIPCHK(hipStreamSynchronize(c->wf_streams[i]));
        HIPCHK(hipStreamSynchronize(extra));
        ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return RT_OK;
    };
    double one = 0.0;
    std::vector<hipStream_t> rejected;      /* kept alive until the end: a stream made right after one was destroyed takes its queue again */
    int rc = RT_OK;
    for (int tries = 0; tries < 12 && c->wf_n_streams < WF_MAX_POOLS && rc == RT_OK; ++tries) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        double ms = 0.0;
        rc = timed(st, ms);
        if (rc == RT_OK && tries == 0) rc = timed(st, ms);      /* the first launch also loads the kernel */
        if (rc != RT_OK) { rejected.push_back(st); break; }
        if (c->wf_n_streams == 0) one = ms;
        if (std::getenv("RT_WF_DEBUG")) std::fprintf(stderr, "wf_pick_streams: candidate %d with %d chosen: %.3f ms (one: %.3f)\n", tries, c->wf_n_streams, ms, one);
        if (c->wf_n_streams == 0 || ms < 1.6 * one) c->wf_streams[c->wf_n_streams++] = st;
        else rejected.push_back(st);
    }
    for (hipStream_t st : rejected) (void)hipStreamDestroy(st);
    if (rc != RT_OK) return rc;
    return RT_OK;
}

/* sums_only: the one-kernel form (rt_wavefront4.h) keeps no path state and no queues in HBM, only a slot's sum (16 B a slot
 * instead of 88) */
static int wf_reserve(rt_ctx* c, size_t n_slots, bool sums_only = false)
{
    if (!c->wf_host_count) HIPCHK(hipHostMalloc((void**)&c->wf_host_count, 5 * WF_MAX_POOLS * sizeof(uint32_t), hipHostMallocDefault));
    for (hipEvent_t& e : c->wf_ev) if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (n_slots > c->wf_slots_cap || (!sums_only && !c->wf_cap_full)) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->wf_mem) { (void)hipFree(c->wf_mem); c->wf_mem = nullptr; c->wf_slots_cap = 0; }
        const size_t per_slot = sums_only ? sizeof(float4) : 5 * sizeof(float4) + 2 * sizeof(uint32_t);
        const size_t bytes = n_slots * per_slot + WF_MAX_POOLS * WF_POOL_COUNT_WORDS * sizeof(uint32_t) + 128
                             + (n_slots / WF_SPAN + 2 * WF_MAX_POOLS + 1) * sizeof(uint32_t);
        if (hipMalloc(&c->wf_mem, bytes) != hipSuccess) {
            (void)hipGetLastError();
            c->wf_mem = nullptr;
            rt_set_error("wavefront integrator: no memory for the path state");
            return RT_ERR_HIP;
        }
        c->wf_slots_cap = n_slots;
        c->wf_cap_full = !sums_only;
    }
    /* carve: five float4 arrays and two queues (or the sums alone), then the counters */
    char* p = (char*)c->wf_mem;
    const size_t cap = c->wf_slots_cap;
    if (c->wf_cap_full) {
        c->wf.ray_o = (float4*)p; p += cap * sizeof(float4);
        c->wf.ray_d = (float4*)p; p += cap * sizeof(float4);
        c->wf.att   = (float4*)p; p += cap * sizeof(float4);
        c->wf.sum   = (float4*)p; p += cap * sizeof(float4);
        c->wf.hit   = (float4*)p; p += cap * sizeof(float4);
        c->wf.queue[0] = (uint32_t*)p; p += cap * sizeof(uint32_t);
        c->wf.queue[1] = (uint32_t*)p; p += cap * sizeof(uint32_t);
    } else {
        c->wf.ray_o = c->wf.ray_d = c->wf.att = c->wf.hit = nullptr;
        c->wf.queue[0] = c->wf.queue[1] = nullptr;
        c->wf.sum = (float4*)p; p += cap * sizeof(float4);
    }
    c->wf.count = (uint32_t*)(((uintptr_t)p + 127u) & ~(uintptr_t)127u);      /* the counters' lines are 128-B aligned */
    c->wf.scan = c->wf.count + WF_MAX_POOLS * WF_POOL_COUNT_WORDS;             /* cap / WF_SPAN + 2 * WF_MAX_POOLS + 1 words */
    return RT_OK;
}

template <uint32_t V>
static int wf_run_part(rt_ctx* c, int w, int h, int4 win, uint32_t s0, uint32_t s1, uint32_t k0, uint32_t k1, int max_depth,
                       int background, float* accum);

__global__ void wf_count_samples_kernel(unsigned long long* counters, unsigned long long n) { atomicAdd(&counters[1], n); }

/* which intersect kernel: 2 = staged rays + queued leaves (rt_wavefront2.h), 1 = the first design (rt_wavefront.h); the
 * second packs a leaf item and a lane into 32 bits, an object and a group ordinal into the mailbox word, and its
 * all-feature variants have a shorter stack.  RT_WF_KERNEL in the environment overrides (A/B measurements). */
template <uint32_t V>
static int wf_pick_kernel(const rt_ctx* c, bool for_render = false)
{
    const bool fits2 = c->n_bvh_items < (1 << WF2_ITEM_BITS) && c->n_objects < (1 << WF2_OBJ_BITS) - 1 && c->n_wf_groups <= WF2_MAX_GROUPS;
    int kernel = WF_KERNEL_DEFAULT == 2 && fits2 ? 2 : 1;
    /* 3 = the eight-wide quantised trees (rt_wavefront8.h): needs them built (leaves of at most two items), their rows
     * addressable in a queue entry and their depth inside its stack */
    const bool fits8 = fits2 && c->depth8 > 0 && c->depth8 <= WF8_LDS_STACK + WF8_OVF_STACK && 16 * c->n_nodes8 < ((size_t)1 << WF2_ITEM_BITS);
    if (WF_KERNEL_DEFAULT == 3 && fits8) kernel = 3;
    if (const char* e = std::getenv("RT_WF_KERNEL")) { const int v = std::atoi(e); if (v == 1 || (v == 2 && fits2) || (v == 3 && fits8)) kernel = v; }
    if (kernel == 2 && c->stack4 > wf2_lds_stack(V & ~RT_F_NOISE) + wf2_ovf(V & ~RT_F_NOISE)) kernel = 1;      /* a tree deeper than this variant's stack */
    if (!wf2_built(V)) kernel = 1;
    /* 4 = the whole path in one persistent kernel (rt_wavefront4.h): renders only, any variant */
    const char* e4 = std::getenv("RT_WF_KERNEL");
    const int v4 = e4 ? std::atoi(e4) : 0;
    if (for_render && (v4 == 4 || (WF_KERNEL_DEFAULT == 4 && !(v4 >= 1 && v4 <= 3)))) kernel = 4;
    return kernel;
}

/* A call is normally ONE part.  While the queue was appended to in the order in which the logic kernel's blocks happened
 * to finish, it drifted away from slot order over hundreds of iterations and long calls were cut into parts of four
 * samples per stream, each starting from a fresh queue (final scene 3840x2160 x 4096 spp as one call: 3.14 Grays/s whole,
 * 3.68 in parts).  Since the queue keeps slot order by construction (rt_wavefront.h, wf_logic_kernel) the parts only add
 * drains: 512 spp of that scene 3.88 whole against 3.80 in parts.  RT_WF_PART = samples per part remains as a
 * measurement switch (0 or unset = whole). */
template <uint32_t V>
static int wf_run(rt_ctx* c, int w, int h, int4 win, uint32_t s0, uint32_t s1, uint32_t k0, uint32_t k1, int max_depth,
                  int background, float* accum)
{
    const size_t n_pixels = (size_t)win.z * (size_t)win.w;
    size_t streams = (c->wf_slots_target + n_pixels - 1) / n_pixels;       /* what wf_run_part will settle on, or a little more */
    if (streams > s1 - s0) streams = s1 - s0;
    unsigned long long part = ~0ull;
    if (const char* e = std::getenv("RT_WF_PART")) { const long long v = std::atoll(e); if (v > 0) part = (unsigned long long)v; }
    if (streams <= 1 || part == ~0ull || (unsigned long long)(s1 - s0) <= part + part / 2) return wf_run_part<V>(c, w, h, win, s0, s1, k0, k1, max_depth, background, accum);
    for (unsigned long long a = s0; a < s1; a += part) {
        const uint32_t b = a + part < s1 ? (uint32_t)(a + part) : s1;
        int rc = wf_run_part<V>(c, w, h, win, (uint32_t)a, b, k0, k1, max_depth, background, accum);
        if (rc) return rc;
    }
    return RT_OK;
}

template <uint32_t V>
static int wf_run_part(rt_ctx* c, int w, int h, int4 win, uint32_t s0, uint32_t s1, uint32_t k0, uint32_t k1, int max_depth,
                       int background, float* accum)
{
    const uint32_t n_pixels = (uint32_t)win.z * (uint32_t)win.w;
    /* about RT_WF_SLOTS paths in flight (each persistent lane should see many rays per launch, so that the
     * drain at the end of a launch, when the queue is empty and waves wait for their last rays, stays small): K sample streams per pixel, never more than there are samples */
    uint32_t streams = (uint32_t)((c->wf_slots_target + n_pixels - 1) / n_pixels);
    if (streams > s1 - s0) streams = s1 - s0;
    if (streams < 1) streams = 1;
    while ((size_t)n_pixels * streams > 0xffffffffull) streams = (streams + 1) / 2;    /* slot indices are 32-bit */
    size_t n_slots = (size_t)n_pixels * streams;
    const int kernel = wf_pick_kernel<V>(c, true);
    c->last_wf_kernel = kernel;
    int rc = wf_reserve(c, n_slots, kernel == 4);
    while (rc && streams > 1) {            /* no room for that many paths: halve, and remember */
        streams = (streams + 1) / 2;
        n_slots = (size_t)n_pixels * streams;
        c->wf_slots_target = n_slots;
        rc = wf_reserve(c, n_slots, kernel == 4);
    }
    if (rc) return rc;
    WfState& W = c->wf;
    W.n_slots = (uint32_t)n_slots; W.n_pixels = n_pixels; W.streams = streams;
    W.win_x0 = (uint32_t)win.x; W.win_y0 = (uint32_t)win.y; W.win_w = (uint32_t)win.z; W.frame_w = (uint32_t)w;
    W.prog = (const DProg*)c->d_prog_wf; W.groups = (const int*)c->d_wf_groups; W.n_groups = c->n_wf_groups;
    /* a refill runs the plain records of the world list for the lanes it fills, so it waits for a batch of idle
     * lanes.  Measured at 1080p (tools/refill_sweep.py), Grays/s at 16 / 24 / 32 / 40 / 48 / 56 idle lanes: final scene
     * (11 plain records) 2.78 / 2.93 / 3.01 / 3.04 / 3.01 / 2.90; 1M triangles (1 record) 2.08 / 2.10 / 2.12 / 2.12 / 2.12 / 2.09 */
    W.refill_min = 40;
    if (kernel == 2 || kernel == 3) W.refill_min = 12;        /* handing over a staged ray is a few LDS reads */
    W.bare = 0;
    W.box_abs = c->box_abs;
    /* XCD-affine claims of the queue (rt_wavefront.h): with the queue in slot order they raise the intersect kernel's L2
     * hit rate on the 1M-triangle scene from 30 % to 56 % (L2 misses x 0.61) for +1 % in rate there and +-0 on the final
     * scene: the misses were one of three limits that sit close together.  RT_WF_AFFINE=0 turns them off. */
    W.affine = 1;
    if (const char* e = std::getenv("RT_WF_AFFINE")) W.affine = std::atoi(e) ? 1 : 0;
    /* RT_WF_REFILL in the environment overrides it (how those figures were measured) */
    if (const char* e = std::getenv("RT_WF_REFILL")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) W.refill_min = v; }
    if (c->stack4 > WF_LDS_STACK + WF_OVF_STACK) { rt_set_error("wavefront integrator: BVH deeper than its traversal stack"); return RT_ERR_UNSUPPORTED; }
    const int lds_stack = kernel == 2 ? wf2_lds_stack(V & ~RT_F_NOISE) : WF_LDS_STACK;
    const size_t lds = kernel == 3 ? (size_t)(c->depth8 < WF8_LDS_STACK ? c->depth8 : WF8_LDS_STACK) * RT_BLOCK * sizeof(uint2)
                                   : (size_t)(c->stack4 < lds_stack ? c->stack4 : lds_stack) * RT_BLOCK * sizeof(int);
    if (kernel == 4) {
        /* one launch: as many blocks as the chip holds at once (later ones would only find the work list empty) */
        W.stream0 = 0;
        HIPCHK(hipMemsetAsync(W.count, 0, WF_MAX_POOLS * WF_POOL_COUNT_WORDS * sizeof(uint32_t), c->stream));
        c->last_wf_pools = 1;
        if (max_depth <= 0) {
            HIPCHK(hipMemsetAsync(W.sum, 0, n_slots * sizeof(float4), c->stream));
            wf_count_samples_kernel<<<1, 1, 0, c->stream>>>(c->d_counters, (unsigned long long)n_pixels * (unsigned long long)(s1 - s0));
        } else {
            const size_t lds4 = lds + WF4_LDS_EXTRA_WORDS * sizeof(int);
            /* four instantiations: black / sky background x one / several BVH groups in the world list (rt_wavefront4.h) */
            using PathKernel = void (*)(DScene, WfState, int, int, uint32_t, uint32_t, uint32_t, uint32_t, int, unsigned long long*);
            const bool mg = c->n_wf_groups != 1;
            const PathKernel kern = background == RT_BG_BLACK ? (mg ? wf_path_kernel<V, RT_BG_BLACK, true> : wf_path_kernel<V, RT_BG_BLACK, false>)
                                                              : (mg ? wf_path_kernel<V, RT_BG_SKY, true> : wf_path_kernel<V, RT_BG_SKY, false>);
            int per_cu = 0, cus = 0;
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, RT_BLOCK, lds4));
            HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
            unsigned pblocks = (unsigned)(per_cu > 0 ? per_cu : 1) * (unsigned)(cus > 0 ? cus : 1);
            if (const char* e = std::getenv("RT_WF_IBLOCKS")) { const int v = std::atoi(e); if (v > 0) pblocks = (unsigned)v; }
            const unsigned all = (unsigned)((n_slots + RT_BLOCK - 1) / RT_BLOCK);
            if (pblocks > all) pblocks = all;
            if (std::getenv("RT_WF_DEBUG")) std::fprintf(stderr, "wf_run_part: path kernel, %u blocks (%d per CU), %zu B of LDS\n", pblocks, per_cu, lds4);
            kern<<<dim3(pblocks), RT_BLOCK, lds4, c->stream>>>(c->ds, W, w, h, s0, s1, k0, k1, max_depth, c->d_counters);
        }
        HIPCHK(hipGetLastError());
        wf_finalize_kernel<<<dim3((n_pixels + RT_BLOCK - 1) / RT_BLOCK), RT_BLOCK, 0, c->stream>>>(W, accum);
        HIPCHK(hipGetLastError());
        return RT_OK;
    }
    /* Pools.  The slots are cut into three pools by sample stream and each pool runs its own intersect / logic loop
     * on a stream of its own, with a persistent intersect grid of a third of the size: while one pool's intersect
     * launch drains (its waves waiting for their last rays) or its logic kernel waits on memory, the other pools'
     * intersect blocks have the wave slots.  The pools run on streams of the context that were found to overlap
     * (wf_pick_streams), fenced against the context's own stream by events at both ends.
     * Measured (tools/iblocks_sweep.sh; final scene 1080p x 64 spp / 1M triangles x 32 spp / x 256 spp, Grays/s):
     * 1 pool x 2048 blocks 3.24 / 2.14 / 2.51; 2 x 2048 3.46 / 2.19 / 2.56; 2 x 1024 3.43 / 2.39 / 2.68;
     * 3 x 682 3.59 / 2.45 / 2.66; 4 x 512 3.62 / 2.51 / 2.71.  Three are used: the process has 4 hardware queues
     * (GPU_MAX_HW_QUEUES) and one stays free for the caller's own streams.
     * RT_WF_POOLS and RT_WF_IBLOCKS in the environment override the two numbers. */
    int n_pools = WF_POOLS_DEFAULT;
    if (const char* e = std::getenv("RT_WF_POOLS")) { const int v = std::atoi(e); if (v >= 1 && v <= WF_MAX_POOLS) n_pools = v; }
    if ((uint32_t)n_pools > streams) n_pools = (int)streams;
    if (n_pools > 1) {
        if (c->wf_n_streams < 0) { int prc = wf_pick_streams(c); if (prc) return prc; }
        if (n_pools > c->wf_n_streams) n_pools = c->wf_n_streams;
        if (n_pools < 2) n_pools = 1;          /* no two streams that overlap: one pool on the context's stream */
    }
    c->last_wf_pools = n_pools;
    if (std::getenv("RT_WF_DEBUG")) std::fprintf(stderr, "wf_run_part: %d pool(s), %d overlapping stream(s) found\n", n_pools, c->wf_n_streams);
    struct Pool { WfState W; hipStream_t st; uint32_t known; int cur; bool pending; unsigned long long iters, max_iters; uint32_t* host; };
    Pool pools[WF_MAX_POOLS];
    W.stream0 = 0;
    HIPCHK(hipMemsetAsync(W.count, 0, WF_MAX_POOLS * WF_POOL_COUNT_WORDS * sizeof(uint32_t), c->stream));
    for (int p = 0; p < n_pools; ++p) {
        Pool& P = pools[p];
        const uint32_t k_first = (uint32_t)((unsigned long long)streams * p / n_pools);
        const uint32_t k_count = (uint32_t)((unsigned long long)streams * (p + 1) / n_pools) - k_first;
        const size_t off = (size_t)k_first * n_pixels;
        std::memcpy((void*)&P.W, (const void*)&W, sizeof(WfState));
        P.W.ray_o += off; P.W.ray_d += off; P.W.att += off; P.W.sum += off; P.W.hit += off;
        P.W.queue[0] += off; P.W.queue[1] += off;
        P.W.count = W.count + WF_POOL_COUNT_WORDS * p;
        P.W.scan = W.scan + off / WF_SPAN + 2 * (size_t)p;
        P.W.n_slots = k_count * n_pixels;
        P.W.stream0 = k_first;
        P.st = n_pools == 1 ? c->stream : c->wf_streams[p];
        P.known = P.W.n_slots;             /* upper bound of the current queue's length (it never grows) */
        P.cur = 0; P.pending = false; P.iters = 0;
        P.host = c->wf_host_count + 4 * p;
        /* every iteration advances every live path by one bounce, and a slot runs at most ceil(samples / streams)
         * paths of at most max_depth bounces: the loop cannot legitimately run longer than this */
        P.max_iters = ((unsigned long long)(s1 - s0 + streams - 1) / streams) * (unsigned long long)(max_depth > 0 ? max_depth : 1) + 4ull;
        if (max_depth > 0) {               /* the first queue is every slot, in order (wf_init_kernel) */
            c->wf_host_count[4 * WF_MAX_POOLS + p] = P.W.n_slots;
            HIPCHK(hipMemcpyAsync(P.W.count, &c->wf_host_count[4 * WF_MAX_POOLS + p], sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        } else P.known = 0;
        const unsigned all_blocks = (unsigned)((P.W.n_slots + RT_BLOCK - 1) / RT_BLOCK);
        wf_init_kernel<V><<<dim3(all_blocks), RT_BLOCK, 0, c->stream>>>(c->ds, P.W, w, h, s0, s1, k0, k1, max_depth, c->d_counters);
        HIPCHK(hipGetLastError());
    }
    if (n_pools > 1) {
        HIPCHK(hipEventRecord(c->wf_ev[WF_MAX_POOLS], c->stream));
        for (int p = 0; p < n_pools; ++p) HIPCHK(hipStreamWaitEvent(pools[p].st, c->wf_ev[WF_MAX_POOLS], 0));
    }
    /* the persistent intersect grids of all pools together: about 2048 blocks (1.6 - 2 x what the chip holds at once) */
    unsigned iblocks_max = 2048u / (unsigned)n_pools;
    if (const char* e = std::getenv("RT_WF_IBLOCKS")) { const int v = std::atoi(e); if (v > 0) iblocks_max = (unsigned)v; }
    const int batch = 4;
    /* test hook: RT_WF_DEBUG_MAX_ITERS caps the iterations a pool may run, so that the "did not drain" exit can be forced */
    if (const char* e = std::getenv("RT_WF_DEBUG_MAX_ITERS")) { const long long v = std::atoll(e); for (int p = 0; p < n_pools; ++p) pools[p].max_iters = (unsigned long long)(v < 0 ? 0 : v); }
    /* The loop proper.  Every failure inside it leaves through ONE exit below, which waits for all the pools' streams:
     * an early return would leave intersect / logic kernels and readback copies of the other pools in flight on
     * wf_mem and wf_host_count, which the caller's next render reuses (or rt_ctx_destroy frees). */
    auto run_pools = [&]() -> int {
        while (true) {
            bool any = false;
            for (int p = 0; p < n_pools; ++p) {
                Pool& P = pools[p];
                if (P.pending) {               /* the length its queue had after the previous batch */
                    HIPCHK(hipEventSynchronize(c->wf_ev[p]));
                    P.known = P.host[P.cur];
                    P.pending = false;
                }
                if (P.known == 0) continue;
                any = true;
                if (P.iters > P.max_iters) { rt_set_error("wavefront integrator: the path queue did not drain"); return RT_ERR_HIP; }
                P.iters += (unsigned)batch;
                for (int b = 0; b < batch; ++b) {
                    const unsigned blocks = (P.known + RT_BLOCK - 1) / RT_BLOCK;
                    const unsigned iblocks = blocks < iblocks_max ? blocks : iblocks_max;
                    const unsigned lblocks = (P.known + WF_SPAN - 1) / WF_SPAN;
                    if constexpr (wf2_built(V)) {
                        if (kernel == 2) wf_intersect2_kernel<(V & ~RT_F_NOISE)><<<dim3(iblocks), RT_BLOCK, lds, P.st>>>(c->ds, P.W, P.cur, k0, k1, c->d_counters);
                        if (kernel == 3) wf_intersect8_kernel<(V & ~RT_F_NOISE)><<<dim3(iblocks), RT_BLOCK, lds, P.st>>>(c->ds, P.W, P.cur, k0, k1, c->d_counters);
                    }
                    if (kernel == 1) wf_intersect_kernel<(V & ~RT_F_NOISE)><<<dim3(iblocks), RT_BLOCK, lds, P.st>>>(c->ds, P.W, P.cur, k0, k1, c->d_counters);   /* reads no texture */
                    if (background == RT_BG_BLACK)
                        wf_logic_kernel<V, RT_BG_BLACK><<<dim3(lblocks), RT_BLOCK, 0, P.st>>>(c->ds, P.W, P.cur, w, h, s1, k0, k1, max_depth);
                    else
                        wf_logic_kernel<V, RT_BG_SKY><<<dim3(lblocks), RT_BLOCK, 0, P.st>>>(c->ds, P.W, P.cur, w, h, s1, k0, k1, max_depth);
                    wf_scan_kernel<<<1, 1024, 0, P.st>>>(P.W, P.cur);            /* survivors per span -> offsets; rewinds the fetch heads */
                    wf_move_kernel<<<dim3(lblocks), RT_BLOCK, 0, P.st>>>(P.W, P.cur);
                    P.cur ^= 1;
                }
                HIPCHK(hipGetLastError());
                HIPCHK(hipMemcpyAsync(P.host, P.W.count, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, P.st));
                HIPCHK(hipEventRecord(c->wf_ev[p], P.st));
                P.pending = true;
            }
            if (!any) break;
        }
        for (int p = 0; p < n_pools && n_pools > 1; ++p) {     /* the pools' last batches have been waited for on the host; order the streams too */
            HIPCHK(hipEventRecord(c->wf_ev[p], pools[p].st));
            HIPCHK(hipStreamWaitEvent(c->stream, c->wf_ev[p], 0));
        }
        return RT_OK;
    };
    const int loop_rc = run_pools();
    if (loop_rc != RT_OK) {
        const std::string why = rt_last_error();
        for (int p = 0; p < n_pools; ++p) (void)hipStreamSynchronize(pools[p].st);
        (void)hipStreamSynchronize(c->stream);
        (void)hipGetLastError();
        rt_set_error(why);
        return loop_rc;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    wf_finalize_kernel<<<dim3((n_pixels + RT_BLOCK - 1) / RT_BLOCK), RT_BLOCK, 0, c->stream>>>(W, accum);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

/* ---- rt_trace_rays_wavefront: caller-supplied rays through the wavefront form's own traversal ---------------------- */
__global__ __launch_bounds__(RT_BLOCK) void wf_trace_fill_kernel(WfState W, int n, const float* __restrict__ rays)
{
    const int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float* q = rays + 7 * (size_t)i;
    W.ray_o[i] = make_float4(q[0], q[1], q[2], q[6]);
    W.ray_d[i] = make_float4(q[3], q[4], q[5], __uint_as_float(0u));
    W.att[i] = make_float4(1.f, 1.f, 1.f, __uint_as_float(0u));
    W.hit[i] = make_float4(FLT_MAX, __int_as_float(-1), __int_as_float(-1), __int_as_float(0));
    W.queue[0][i] = (uint32_t)i;
}
template <uint32_t F>
__global__ __launch_bounds__(RT_BLOCK) void wf_trace_out_kernel(DScene S, WfState W, int n, const float* __restrict__ rays, rt_hit_record* __restrict__ out)
{
    const int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float* q = rays + 7 * (size_t)i;
    DRay r;
    r.o = mk(q[0], q[1], q[2]); r.d = mk(q[3], q[4], q[5]); r.time = q[6];
    const float4 h4 = W.hit[i];
    HitLite hl = {h4.x, __float_as_int(h4.y), __float_as_int(h4.z), __float_as_int(h4.w)};
    rt_hit_record o;
    memset(&o, 0, sizeof o);
    o.prim = -1;
    if (hl.obj >= 0) {
        HitRec h;
        rebuild<F, true>(S, r, hl, .001f, h);
        o.hit = 1; o.prim = h.prim_id; o.t = h.t; o.u = h.u; o.v = h.v;
        o.p[0] = h.p.x; o.p[1] = h.p.y; o.p[2] = h.p.z;
        o.n[0] = h.n.x; o.n[1] = h.n.y; o.n[2] = h.n.z;
        o.front_face = h.front; o.material = h.material;
    }
    out[i] = o;
}
template <uint32_t V>
static int wf_trace(rt_ctx* c, int n, const float* rays, rt_hit_record* out)
{
    int rc = wf_reserve(c, (size_t)n);
    if (rc) return rc;
    WfState W = c->wf;
    W.n_slots = (uint32_t)n; W.n_pixels = (uint32_t)n; W.streams = 1; W.stream0 = 0;
    W.win_x0 = W.win_y0 = 0; W.win_w = W.frame_w = (uint32_t)n;
    W.prog = (const DProg*)c->d_prog_wf; W.groups = (const int*)c->d_wf_groups; W.n_groups = c->n_wf_groups;
    const int kernel = wf_pick_kernel<V>(c);
    W.refill_min = kernel >= 2 ? 12 : 40;
    W.affine = 0;
    W.bare = 1;
    W.box_abs = c->box_abs;
    if (c->stack4 > WF_LDS_STACK + WF_OVF_STACK) { rt_set_error("rt_trace_rays_wavefront: BVH deeper than the traversal stack"); return RT_ERR_UNSUPPORTED; }
    const int lds_stack = kernel == 2 ? wf2_lds_stack(V & ~RT_F_NOISE) : WF_LDS_STACK;
    const size_t lds = kernel == 3 ? (size_t)(c->depth8 < WF8_LDS_STACK ? c->depth8 : WF8_LDS_STACK) * RT_BLOCK * sizeof(uint2)
                                   : (size_t)(c->stack4 < lds_stack ? c->stack4 : lds_stack) * RT_BLOCK * sizeof(int);
    const unsigned blocks = (unsigned)((n + RT_BLOCK - 1) / RT_BLOCK);
    HIPCHK(hipMemsetAsync(W.count, 0, WF_POOL_COUNT_WORDS * sizeof(uint32_t), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));                  /* the pinned word below may still be read by an earlier call's copy */
    c->wf_host_count[4 * WF_MAX_POOLS] = (uint32_t)n;
    HIPCHK(hipMemcpyAsync(W.count, &c->wf_host_count[4 * WF_MAX_POOLS], sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    wf_trace_fill_kernel<<<dim3(blocks), RT_BLOCK, 0, c->stream>>>(W, n, rays);
    const unsigned iblocks = blocks < 2048u ? blocks : 2048u;
    if constexpr (wf2_built(V)) {
        if (kernel == 2) wf_intersect2_kernel<(V & ~RT_F_NOISE)><<<dim3(iblocks), RT_BLOCK, lds, c->stream>>>(c->ds, W, 0, 0u, 0u, c->d_counters + 3);
        if (kernel == 3) wf_intersect8_kernel<(V & ~RT_F_NOISE)><<<dim3(iblocks), RT_BLOCK, lds, c->stream>>>(c->ds, W, 0, 0u, 0u, c->d_counters + 3);
    }
    if (kernel == 1) wf_intersect_kernel<(V & ~RT_F_NOISE)><<<dim3(iblocks), RT_BLOCK, lds, c->stream>>>(c->ds, W, 0, 0u, 0u, c->d_counters + 3);
    wf_trace_out_kernel<V><<<dim3(blocks), RT_BLOCK, 0, c->stream>>>(c->ds, W, n, rays, out);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

/* No render: a parity hook like rt_trace_rays, through the intersect kernel the wavefront form renders with. */
extern "C" int rt_trace_rays_wavefront(rt_ctx* c, int n, const float* rays, rt_hit_record* out)
{
    int rc = need_scene(c, "rt_trace_rays_wavefront");
    if (rc) return rc;
    if (n < 0 || (n && (!rays || !out))) { rt_set_error("rt_trace_rays_wavefront: bad arguments"); return RT_ERR_ARG; }
    if (n == 0) return RT_OK;
    HIPCHK(hipSetDevice(c->device));
    const char* force = std::getenv("RT_FORCE_VARIANT");          /* test hook, as in the render calls */
    const bool force_all = force && (!std::strcmp(force, "all") || !std::strcmp(force, "list")) && (c->features & ~RT_F_ALL) == 0;
    return force_all ? wf_trace<RT_F_ALL>(c, n, rays, out)
           : (c->features & ~RT_V_BVH_TRI) == 0 ? wf_trace<RT_V_BVH_TRI>(c, n, rays, out)
           : (c->features & ~RT_V_NOTRI) == 0 ? wf_trace<RT_V_NOTRI>(c, n, rays, out)
           : (c->features & ~RT_F_ALL) == 0   ? wf_trace<RT_F_ALL>(c, n, rays, out)
                                              : wf_trace<RT_V_NOISE>(c, n, rays, out);
}

/* Runs of the megakernel.  The sample range is cut into up to 16 runs of >= 64 samples, and a block renders one run each of
 * as many tiles spread over the frame, its lanes walking through them independently (render_kernel).  Measured on the
 * Cornell box, 1024 x 1024 x 1024 spp: 1 run (a wave stays on one tile until its slowest pixel is done) 20.9 Grays/s
 * before this and 19.2 with the walking code in the loop; 4 runs 20.5, 8 runs 23.6, 16 runs 24.5, 32-128 runs 24.5.
 * The number of runs depends on the sample count only, so a window and the full frame group a pixel's samples
 * alike; calls of fewer than 128 samples have one run and sum a pixel's samples in order. */
static void mega_runs(uint32_t n_samples, uint32_t& groups, uint32_t& per_group)
{
    groups = n_samples / 64u;
    groups = groups < 1u ? 1u : (groups > 16u ? 16u : groups);
    /* RT_MEGA_GROUPS in the environment overrides the number of runs (how the figures above were measured) */
    if (const char* e = std::getenv("RT_MEGA_GROUPS")) { groups = (uint32_t)std::atoi(e); if (groups < 1 || groups > n_samples) groups = 1; }
    per_group = (n_samples + groups - 1) / groups;
    groups = (n_samples + per_group - 1) / per_group;
}

/* render_kernel (+ combine_kernel when the runs' frames go through `partial`) on the context's stream */
static hipError_t mega_launch(rt_ctx* c, int w, int h, int4 win, uint32_t s0, uint32_t s1, uint32_t k0, uint32_t k1, int max_depth,
                              int background, float* accum, uint32_t groups, uint32_t per_group, float* partial)
{
    const size_t lds = RT_BLOCK * sizeof(int) + stack_bytes(c);
    dim3 grid(n_tiles(win.z, win.w));
    if (!partial) groups = 1;
    const uint32_t per = groups > 1 ? per_group : s1 - s0;
    /* block B takes run k of tile (B + k * tile_step) mod n_tiles: steps of about 0.38 of the frame */
    uint32_t tile_step = (uint32_t)(grid.x * 0.3819660113);
    if (tile_step == 0) tile_step = 1;
    /* the smallest compiled feature set that covers the scene */
    static const uint32_t variants[] = {RT_V_PLANAR, RT_V_PLANAR_MEDIA, RT_V_LIST, RT_V_BVH_TRI, RT_F_ALL, RT_V_NOISE};
    uint32_t pick = RT_V_NOISE;
    for (uint32_t v : variants) if ((c->features & ~v) == 0) { pick = v; break; }
    /* test hook: RT_FORCE_VARIANT=list|all|noise runs a LARGER compiled variant than the scene needs (same results, by
     * construction; tests/test_gpu_parity.py checks it: DESIGN.md 3.4).  Ignored when it does not cover the scene. */
    if (const char* e = std::getenv("RT_FORCE_VARIANT")) {
        const uint32_t want = !std::strcmp(e, "list") ? RT_V_LIST : !std::strcmp(e, "all") ? RT_F_ALL : !std::strcmp(e, "noise") ? RT_V_NOISE : 0u;
        if (want && (c->features & ~want) == 0) pick = want;
    }
#define RT_LAUNCH(V, G) render_kernel<V, G><<<grid, RT_BLOCK, lds, c->stream>>>(c->ds, w, h, win, s0, s1, k0, k1, max_depth, accum, c->d_counters, per, partial, groups, tile_step)
#define RT_LAUNCH_BG(V) { if (background == RT_BG_BLACK) RT_LAUNCH(V, RT_BG_BLACK); else RT_LAUNCH(V, RT_BG_SKY); }
    switch (pick) {
    case RT_V_PLANAR:       RT_LAUNCH_BG(RT_V_PLANAR) break;
    case RT_V_PLANAR_MEDIA: RT_LAUNCH_BG(RT_V_PLANAR_MEDIA) break;
    case RT_V_LIST:         RT_LAUNCH_BG(RT_V_LIST) break;
    case RT_V_BVH_TRI:      RT_LAUNCH_BG(RT_V_BVH_TRI) break;
    case RT_F_ALL:          RT_LAUNCH_BG(RT_F_ALL) break;
    default:                RT_LAUNCH_BG(RT_V_NOISE) break;
    }
#undef RT_LAUNCH_BG
#undef RT_LAUNCH
    hipError_t le = hipGetLastError();
    if (le == hipSuccess && partial) {
        const size_t n_win = (size_t)win.z * win.w * 3;
        combine_kernel<<<dim3((unsigned)((n_win + RT_BLOCK - 1) / RT_BLOCK)), RT_BLOCK, 0, c->stream>>>(accum, partial, w, h, win, (int)groups);
        le = hipGetLastError();
    }
    return le;
}

extern "C" int rt_ctx_set_integrator(rt_ctx* c, int mode)
{
    if (!c || mode < RT_INTEGRATOR_AUTO || mode > RT_INTEGRATOR_WAVEFRONT) { rt_set_error("rt_ctx_set_integrator: bad arguments"); return RT_ERR_ARG; }
    c->integrator = mode;
    return RT_OK;
}

extern "C" int rt_render_accumulate(rt_ctx* c, int w, int h, uint32_t s0, uint32_t s1, uint64_t seed, int max_depth,
                                    int background, float* accum)
{
    if (!frame_ok(w, h)) { rt_set_error("rt_render_accumulate: bad arguments"); return RT_ERR_ARG; }
    return rt_render_accumulate_window(c, w, h, 0, 0, w, h, s0, s1, seed, max_depth, background, accum);
}

extern "C" int rt_render_accumulate_window(rt_ctx* c, int w, int h, int x0, int y0, int x1, int y1, uint32_t s0, uint32_t s1,
                                           uint64_t seed, int max_depth, int background, float* accum)
{
    int rc = need_scene(c, "rt_render_accumulate");
    if (rc) return rc;
    if (!frame_ok(w, h) || !accum || s1 < s0 || max_depth < 0 || (background != RT_BG_BLACK && background != RT_BG_SKY) ||
        x0 < 0 || y0 < 0 || x1 > w || y1 > h || x0 >= x1 || y0 >= y1) {
        rt_set_error("rt_render_accumulate: bad arguments");
        return RT_ERR_ARG;
    }
    const int4 win = make_int4(x0, y0, x1 - x0, y1 - y0);
    if (s1 == s0) return RT_OK;
    HIPCHK(hipSetDevice(c->device));
    /* the event pair that times the call: from here to note_render (which takes it over) every failure leaves through
     * `fail`, which destroys both */
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto fail = [&](hipError_t e, const char* what) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        rt_set_error(std::string("HIP error ") + hipGetErrorString(e) + " at " + what);
        return RT_ERR_HIP;
    };
#define EVCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(e_, #expr); } while (0)
    EVCHK(hipEventCreate(&e0));
    EVCHK(hipEventCreate(&e1));
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    EVCHK(hipEventRecord(e0, c->stream));
    /* auto: the wavefront form pays off as soon as a BVH is traversed.  Measured at 1920x1080 x 32 spp (tools/compare_integrators.py),
     * wavefront (one kernel, round 3) / megakernel: 1M triangles 3.18, 200 k 2.58, 20 k 1.82; final scene with 11 k / 1.4 k items
     * 3.27 / 3.55; Cornell + 1000-sphere cluster 2.43; 488 spheres in a world BVH 1.95; 16 boxes 1.02; no BVH at all (Cornell,
     * Cornell smoke) 0.92 / 0.88.  (Two-kernel form, round 2: 1.97, 1.61, 1.21; 2.29-2.41; 1.51; 1.17; 0.48; 0.31.) */
    const bool big = c->n_bvh_items >= RT_WF_AUTO_ITEMS;
    const bool wavefront = c->integrator == RT_INTEGRATOR_WAVEFRONT || (c->integrator == RT_INTEGRATOR_AUTO && big && c->stack4 <= WF_LDS_STACK + WF_OVF_STACK);
    c->last_integrator = wavefront ? RT_INTEGRATOR_WAVEFRONT : RT_INTEGRATOR_MEGAKERNEL;
    if (wavefront) {
        /* the first variant that covers the scene's features */
        const char* force = std::getenv("RT_FORCE_VARIANT");          /* test hook, as in mega_launch */
        const bool force_all = force && (!std::strcmp(force, "all") || !std::strcmp(force, "list")) && (c->features & ~RT_F_ALL) == 0;
        int wrc = force_all ? wf_run<RT_F_ALL>(c, w, h, win, s0, s1, k0, k1, max_depth, background, accum)
                  : (c->features & ~RT_V_BVH_TRI) == 0 ? wf_run<RT_V_BVH_TRI>(c, w, h, win, s0, s1, k0, k1, max_depth, background, accum)
                  : (c->features & ~RT_V_NOTRI) == 0 ? wf_run<RT_V_NOTRI>(c, w, h, win, s0, s1, k0, k1, max_depth, background, accum)
                  : (c->features & ~RT_F_ALL) == 0   ? wf_run<RT_F_ALL>(c, w, h, win, s0, s1, k0, k1, max_depth, background, accum)
                                                     : wf_run<RT_V_NOISE>(c, w, h, win, s0, s1, k0, k1, max_depth, background, accum);
        EVCHK(hipEventRecord(e1, c->stream));
        int nrc = note_render(c, e0, e1);
        return wrc ? wrc : nrc;
    }
    uint32_t groups, per_group;
    mega_runs(s1 - s0, groups, per_group);
    float* partial = nullptr;
    if (groups > 1) {
        const size_t need = (size_t)groups * (size_t)w * (size_t)h * 3;
        if (need > c->partial_floats) {
            EVCHK(hipStreamSynchronize(c->stream));
            if (c->d_partial) { (void)hipFree(c->d_partial); c->d_partial = nullptr; c->partial_floats = 0; }
            if (hipMalloc((void**)&c->d_partial, need * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); groups = 1; }
            else c->partial_floats = need;
        }
        partial = groups > 1 ? c->d_partial : nullptr;
    }
    hipError_t le = mega_launch(c, w, h, win, s0, s1, k0, k1, max_depth, background, accum, groups, per_group, partial);
    EVCHK(hipEventRecord(e1, c->stream));
#undef EVCHK
    int nrc = note_render(c, e0, e1);
    if (le != hipSuccess) { rt_set_error(std::string("render launch failed: ") + hipGetErrorString(le)); return RT_ERR_HIP; }
    return nrc;
}

/* No counterpart in the reference (include/rt_api.h): rt_render_accumulate in megakernel form for callers that cannot
 * allocate.  The per-run partial frames live in rt_noalloc_scratch (this GPU's copy, address looked up at rt_ctx_create);
 * no hipMalloc / hipFree / event creation happens here, and the call is not timed (rt_counters' times skip it). */
extern "C" int rt_render_accumulate_noalloc(rt_ctx* c, int w, int h, uint32_t s0, uint32_t s1, uint64_t seed, int max_depth,
                                            int background, float* accum)
{
    int rc = need_scene(c, "rt_render_accumulate_noalloc");
    if (rc) return rc;
    if (!frame_ok(w, h) || !accum || s1 < s0 || max_depth < 0 || (background != RT_BG_BLACK && background != RT_BG_SKY)) {
        rt_set_error("rt_render_accumulate_noalloc: bad arguments");
        return RT_ERR_ARG;
    }
    if (s1 == s0) return RT_OK;
    uint32_t groups, per_group;
    mega_runs(s1 - s0, groups, per_group);
    if (groups > 16u) {              /* RT_MEGA_GROUPS (a measurement switch) may ask for more runs: the documented limit is 16 here */
        groups = 16u;
        per_group = (s1 - s0 + groups - 1) / groups;
        groups = (s1 - s0 + per_group - 1) / per_group;
    }
    if (groups > 1 && (unsigned long long)groups * (unsigned long long)w * (unsigned long long)h * 3ull > (unsigned long long)RT_NOALLOC_FLOATS) {
        rt_set_error("rt_render_accumulate_noalloc: " + std::to_string(groups) + " runs of a " + std::to_string(w) + "x" + std::to_string(h) +
                     " frame need more than the " + std::to_string((unsigned long long)RT_NOALLOC_FLOATS * 4ull >> 20) +
                     " MiB of built-in scratch (16 runs of 512x512 pixels); render fewer than 128 samples per call, a smaller frame, or use rt_render_accumulate");
        return RT_ERR_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    float* scratch = c->noalloc_scratch;      /* resolved at rt_ctx_create: no lock, no lookup here */
    if (!scratch) { rt_set_error("rt_render_accumulate_noalloc: the context has no scratch address"); return RT_ERR_HIP; }
    c->last_integrator = RT_INTEGRATOR_MEGAKERNEL;
    /* every (run, pixel) cell the combine step reads is written by the render kernel of the same call first (a pixel hands
     * in each of its runs, black ones included), so what an earlier call left in the array is never read: nothing to clear */
    hipError_t le = mega_launch(c, w, h, make_int4(0, 0, w, h), s0, s1, (uint32_t)seed, (uint32_t)(seed >> 32), max_depth, background,
                                accum, groups, per_group, groups > 1 ? scratch : nullptr);
    if (le != hipSuccess) { rt_set_error(std::string("render launch failed: ") + hipGetErrorString(le)); return RT_ERR_HIP; }
    return RT_OK;
}

extern "C" int rt_resolve(rt_ctx* c, const float* accum, int w, int h, uint32_t spp, int gamma, float* rgb)
{
    if (!c || !frame_ok(w, h) || !accum || !rgb || spp == 0) { rt_set_error("rt_resolve: bad arguments"); return RT_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    size_t n = (size_t)w * h * 3;
    float inv = 1.0f / float(spp);
    resolve_kernel<<<dim3((unsigned)((n + RT_BLOCK - 1) / RT_BLOCK)), RT_BLOCK, 0, c->stream>>>(accum, n, inv, gamma, rgb);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

/* ---- RCCL reduce of per-GPU accumulation buffers: csrc/rt_reduce.cpp holds the exchange logic (plain device / stream /
 * buffer lists, no rt_ctx); here the contexts are turned into those lists. */
static int ctx_lists(const char* who, rt_ctx* const* ctxs, float* const* accum, int n, std::vector<int>& devices, std::vector<hipStream_t>& streams)
{
    if (!ctxs || !accum || n < 1) { rt_set_error(std::string(who) + ": bad arguments"); return RT_ERR_ARG; }
    devices.resize(n);
    streams.resize(n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) { rt_set_error(std::string(who) + ": null context or buffer"); return RT_ERR_ARG; }
        devices[i] = ctxs[i]->device;
        streams[i] = ctxs[i]->stream;
    }
    return RT_OK;
}

extern "C" int rt_reduce_accum(rt_ctx* const* ctxs, float* const* accum, int n, uint64_t count, int root)
{
    if (n < 1 || root < 0 || root >= n || count == 0) { rt_set_error("rt_reduce_accum: bad arguments"); return RT_ERR_ARG; }
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
    int rc = ctx_lists("rt_reduce_accum", ctxs, accum, n, devices, streams);
    if (rc) return rc;
    return rt_reduce_accum_lists(devices.data(), streams.data(), accum, n, count, root);
}

extern "C" int rt_reduce_scatter_rows(rt_ctx* const* ctxs, float* const* accum, int n, int w, int h, int* rows)
{
    if (n < 1 || !frame_ok(w, h) || !rows) { rt_set_error("rt_reduce_scatter_rows: bad arguments"); return RT_ERR_ARG; }
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
    int rc = ctx_lists("rt_reduce_scatter_rows", ctxs, accum, n, devices, streams);
    if (rc) return rc;
    return rt_reduce_scatter_lists(devices.data(), streams.data(), accum, n, w, h, rows);
}

extern "C" int rt_synchronize(rt_ctx* c)
{
    if (!c) { rt_set_error("rt_synchronize: null context"); return RT_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return RT_OK;
}

/* folds the event pairs of finished renders into the timing totals; keep_newest > 0 leaves that many pairs pending */
static int drain_events(rt_ctx* c, size_t keep_newest)
{
    if (keep_newest) {
        while (c->pending.size() > keep_newest) {
            auto p = c->pending.front();
            float ms = 0.f;
            HIPCHK(hipEventSynchronize(p.second));
            HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
            c->render_ms += ms;
            c->last_ms = ms;
            c->launches++;
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
            c->pending.erase(c->pending.begin());
        }
        return RT_OK;
    }
    for (auto& p : c->pending) {
        float ms = 0.f;
        HIPCHK(hipEventSynchronize(p.second));
        HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
        c->render_ms += ms;
        c->last_ms = ms;
        c->launches++;
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    c->pending.clear();
    return RT_OK;
}

extern "C" int rt_get_counters(rt_ctx* c, rt_counters* out)
{
    if (!c || !out) { rt_set_error("rt_get_counters: null argument"); return RT_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    int rc = drain_events(c);
    if (rc) return rc;
    unsigned long long h[2] = {0, 0};
    HIPCHK(hipMemcpy(h, c->d_counters, sizeof h, hipMemcpyDeviceToHost));
    out->rays = h[0];
    out->samples = h[1];
    out->render_ms = c->render_ms;
    out->last_render_ms = c->last_ms;
    out->render_launches = c->launches;
    out->last_integrator = (uint32_t)c->last_integrator;
    out->wf_pools = (uint32_t)c->last_wf_pools;
    out->wf_kernel = (uint32_t)c->last_wf_kernel;
    out->reserved = 0u;
    out->wf_streams_found = c->wf_n_streams < 0 ? 0u : (uint32_t)c->wf_n_streams;
    return RT_OK;
}

#ifdef RT_STAMPS
/* words 16..31 of the counters: the section cycles of wf_intersect2_kernel */
extern "C" int rt_debug_sections(rt_ctx* c, unsigned long long* out16)
{
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out16, c->d_counters + 16, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RT_OK;
}
extern "C" int rt_debug_stamps(rt_ctx* c, unsigned long long* out12)
{
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out12, c->d_counters + 4, 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpyFromSymbol(out12 + 6, HIP_SYMBOL(rt_dbg_counts), 4 * sizeof(unsigned long long)));
    return RT_OK;
}
#endif

extern "C" int rt_reset_counters(rt_ctx* c)
{
    if (!c) { rt_set_error("rt_reset_counters: null context"); return RT_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    int rc = drain_events(c);
    if (rc) return rc;
    /* on the context's own stream: it may be non-blocking, and a null-stream memset would not be ordered with the next render */
    HIPCHK(hipMemsetAsync(c->d_counters, 0, RT_N_COUNTERS * sizeof(unsigned long long), c->stream));
    c->render_ms = c->last_ms = 0.0;
    c->launches = 0;
    return RT_OK;
}
