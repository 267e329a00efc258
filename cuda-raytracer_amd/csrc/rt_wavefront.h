/*
 * rt_wavefront.h — the wavefront form of the path tracer, used for scenes with a BVH.
 *
 * The reference's `render` (main.cu:125-198) and the megakernel in rt_kernels.hip keep a path in one
 * lane from camera to termination.  On a BVH scene the closest-hit query dominates and its cost per
 * ray is heavy-tailed (1M-triangle scene: 43 traversal steps on average, several hundred for a grazing
 * ray), so a wave that only refills its lanes between bounces runs its traversal loop with a few lanes
 * live (measured 8 % in the node loop).  Here the path state lives in HBM and two kernels alternate:
 *
 *   wf_intersect  persistent waves; a batch of idle lanes takes rays from the queue, runs the plain records of
 *                 the world list for them at once, and from then on every lane is a small interpreter that
 *                 advances ITS ray through the list's BVH groups by one step per trip (a 4-wide node, or one
 *                 leaf primitive); the moment its ray is finished the lane is free for the next batch — no
 *                 lane waits for another lane's ray.
 *   wf_logic      one lane per queued path: rebuild the hit record, scatter, accumulate, start the next
 *                 sample of the slot when the path ends, and queue the slot again while it is alive.
 *
 * A slot owns a pixel and the sample indices s_begin + k, s_begin + k + K, ... (k = its stream), runs them
 * one after another and keeps their sum; wf_finalize adds the K partial sums of a pixel in stream order,
 * so the image is reproducible bit for bit from run to run.  The RNG is addressed by (pixel, sample,
 * bounce), so every sample's path is the same one the megakernel and the oracle trace.
 *
 * State traffic is 5 x 16 B per slot read and written about once per ray: ~170 B/ray, against the
 * ~2.8 KB/ray of BVH nodes and leaf rows the intersect kernel fetches on the triangle scene.
 */
#ifndef RT_WAVEFRONT_H
#define RT_WAVEFRONT_H
#include "rt_device.h"

struct WfState {
    float4* __restrict__ ray_o;   /* origin xyz, time */
    float4* __restrict__ ray_d;   /* direction xyz, bounce index (bits) */
    float4* __restrict__ att;     /* attenuation xyz, sample index in flight (bits) */
    float4* __restrict__ sum;     /* radiance sum of the slot's finished samples xyz, unused */
    float4* __restrict__ hit;     /* t, object, group, aux (bits) */
    uint32_t* __restrict__ queue[2];
    uint32_t* __restrict__ count; /* [0],[1] = entries of queue[0],queue[1]; [2] = intersect fetch head */
    uint32_t* __restrict__ scan;  /* n_slots / 1024 + 2 words: survivors per logic block, then their prefix sums */
    uint32_t n_slots, n_pixels, streams;    /* n_pixels = pixels of the rendered window; streams = sample streams per pixel of the whole call */
    uint32_t stream0;                       /* first stream of this pool (the slots of a call are run as one or two pools, rt_kernels.hip wf_run) */
    uint32_t win_x0, win_y0, win_w, frame_w;   /* the window inside the frame (rt_render_accumulate_window) */
    /* the world list for this integrator (rt_host.h rt_pack_program): its plain records, run in one go when a ray
     * is handed to a lane, and the BVH groups, walked one after the other by the step loop */
    const DProg* __restrict__ prog;
    const int* __restrict__ groups;
    int n_groups;
    int refill_min;    /* idle lanes that make a refill (and its run over the plain records) worth while */
    int affine;        /* 1: a wave takes its chunks of the queue from the image band of its XCD first (below) */
    int bare;          /* 1: a bare closest-hit query (rt_trace_rays_wavefront): no RNG, media are ignored as in rt_trace_rays */
    float box_abs;     /* the largest absolute coordinate of any BVH box (rt_wavefront8.h: scales the rounding allowance) */
};

/* slot -> pixel: stream k of window pixel p is slot k * n_pixels + p; returns the pixel's index in the FRAME,
 * which is what addresses the RNG and the accumulation buffer */
__device__ __forceinline__ uint32_t wf_frame_pixel(const WfState& W, uint32_t slot, uint32_t& i, uint32_t& j)
{
    const uint32_t p = slot % W.n_pixels;
    i = W.win_x0 + p % W.win_w;
    j = W.win_y0 + p / W.win_w;
    return j * W.frame_w + i;
}
__device__ __forceinline__ uint32_t wf_frame_pixel(const WfState& W, uint32_t slot)
{
    if (W.win_w == W.frame_w) return slot % W.n_pixels + W.win_y0 * W.frame_w;     /* full-width band: no division by the width */
    uint32_t i, j;
    return wf_frame_pixel(W, slot, i, j);
}

enum { WF_NEED = 0, WF_BVH = 2 };
/* waves per SIMD the intersect kernel is compiled for: the triangle-mesh variant fits 96 VGPRs without spilling
 * (5 waves: +1.5 % on the 1M-triangle scene; 6 waves = 80 VGPRs with 14 spilled: -16 %); the general ones spill at 5 (final scene: all-feature variant -27 %,
 * the one without triangles, 19 spilled registers, -9 %) */
#ifndef WF_GENERAL_WAVES
#define WF_GENERAL_WAVES 4
#endif
#ifndef WF_TRI_WAVES
#define WF_TRI_WAVES 5
#endif
constexpr int wf_intersect_waves(uint32_t F) { return (F & ~(RT_F_BVH | RT_F_TRI | RT_F_PLANAR)) == 0 ? WF_TRI_WAVES : WF_GENERAL_WAVES; }
#define WF_HEAD_WORDS 32     /* a band's fetch head sits on a 128-B line of its own: count[32 (1 + band)] */
#define WF_LDS_STACK 20      /* traversal-stack entries per lane kept in LDS; deeper ones go to scratch */
#define WF_OVF_STACK 44      /* entries of the scratch overflow: 3 per level of a 4-wide tree, rarely reached */
#ifndef WF_PRIM_BATCH
#define WF_CHUNK 1024        /* queue entries a wave claims per atomic, at most */
#define WF_PRIM_BATCH 20     /* lanes at a leaf that make a primitive step worth running */
#define WF_NODE_LOW 16       /* fewer interior lanes than this: run whatever else is waiting */
/* a flat optimum: PRIM_BATCH 16 / 24 and NODE_LOW 8 / 24 are all within 0.5 % of these on the 1M-triangle and final
 * scenes (tools/lib_sweep.py) */
#endif

/* The node step picks a box's near and far plane along an axis by the SIGN of the ray's direction instead of computing both
 * distances and taking v_min / v_max of them: a min or max costs 1.7 multiplications on this chip (profiles/r03_valu_costs.jsonl)
 * and a four-wide node step held 24 of them for this alone.  DNode4 is axis-major (rt_host.h): the four children's min x in
 * one 16-byte group, their max x in the next, and so on, so "near" and "far" are two LOADS whose offsets differ in one bit
 * that depends on the ray only.  Byte k of the result = offset of axis k's near group in the row: 32 k, + 16 when the ray
 * runs towards negative k (sign bit of 1 / d, so -0 counts as negative like its reciprocal -inf); far = near ^ 16.
 * The distances themselves are slab()'s: (plane - o) * (1 / d), near <= far by the monotonicity of rounding, NaNs (0 * inf)
 * ignored by the v_max / v_min that remain, an unused slot's inverted box still a miss. */
__device__ __forceinline__ uint32_t wf4_sign_offsets(V3 id)
{
    const uint32_t sx = __float_as_uint(id.x) >> 31, sy = __float_as_uint(id.y) >> 31, sz = __float_as_uint(id.z) >> 31;
    return (sx << 4) | ((32u + (sy << 4)) << 8) | ((64u + (sz << 4)) << 16);
}
/* one child of the node: column k of the near / far plane groups */
__device__ __forceinline__ bool wf4_child(float nx, float ny, float nz, float fx, float fy, float fz, V3 o, V3 id, float tMin, float tMax, float& tnear)
{
    const float tn = fmaxf(fmaxf((nx - o.x) * id.x, (ny - o.y) * id.y), fmaxf((nz - o.z) * id.z, tMin));
    const float tf = fminf(fminf((fx - o.x) * id.x, (fy - o.y) * id.y), fminf((fz - o.z) * id.z, tMax));
    tnear = tn;
    return tn <= tf * 1.0000008f;
}

/* the node step's test of the four children of node `ref` (>= 0): loads, distances, and a child that is missed (or an unused
 * slot) left with t = +inf and no reference.  Declares r0..r3 and t0..t3.  128-byte rows addressed by a 32-bit offset from
 * the table's base (rt_scene_upload refuses tables beyond 2^24 nodes). */
#define WF_NODE4_TEST(o_, id_, soff_, tMin_, tMax_)                                                                                   \
            const char* nrow_ = reinterpret_cast<const char*>(S.nodes4);                                                              \
            const uint32_t nbase_ = (uint32_t)ref << 7;                                                                               \
            const uint32_t onx_ = nbase_ + ((soff_) & 0xffu), ony_ = nbase_ + (((soff_) >> 8) & 0xffu), onz_ = nbase_ + (((soff_) >> 16) & 0xffu); \
            const float4 nx_ = *reinterpret_cast<const float4*>(nrow_ + onx_), fx_ = *reinterpret_cast<const float4*>(nrow_ + (onx_ ^ 16u)); \
            const float4 ny_ = *reinterpret_cast<const float4*>(nrow_ + ony_), fy_ = *reinterpret_cast<const float4*>(nrow_ + (ony_ ^ 16u)); \
            const float4 nz_ = *reinterpret_cast<const float4*>(nrow_ + onz_), fz_ = *reinterpret_cast<const float4*>(nrow_ + (onz_ ^ 16u)); \
            const float4 n6_ = *reinterpret_cast<const float4*>(nrow_ + nbase_ + 96u);                                                \
            int r0 = __float_as_int(n6_.x), r1 = __float_as_int(n6_.y), r2 = __float_as_int(n6_.z), r3 = __float_as_int(n6_.w);       \
            float t0, t1, t2, t3;                                                                                                     \
            if (!(wf4_child(nx_.x, ny_.x, nz_.x, fx_.x, fy_.x, fz_.x, o_, id_, tMin_, tMax_, t0) && r0 != RT_REF_NONE)) { t0 = FLT_MAX; r0 = RT_REF_NONE; } \
            if (!(wf4_child(nx_.y, ny_.y, nz_.y, fx_.y, fy_.y, fz_.y, o_, id_, tMin_, tMax_, t1) && r1 != RT_REF_NONE)) { t1 = FLT_MAX; r1 = RT_REF_NONE; } \
            if (!(wf4_child(nx_.z, ny_.z, nz_.z, fx_.z, fy_.z, fz_.z, o_, id_, tMin_, tMax_, t2) && r2 != RT_REF_NONE)) { t2 = FLT_MAX; r2 = RT_REF_NONE; } \
            if (!(wf4_child(nx_.w, ny_.w, nz_.w, fx_.w, fy_.w, fz_.w, o_, id_, tMin_, tMax_, t3) && r3 != RT_REF_NONE)) { t3 = FLT_MAX; r3 = RT_REF_NONE; }

/* ------------------------------------------------------------------ intersect -- */
template <uint32_t F>
__global__ __launch_bounds__(RT_BLOCK, wf_intersect_waves(F)) void wf_intersect_kernel(DScene S, WfState W, int qi, uint32_t k0, uint32_t k1,
                                                                unsigned long long* __restrict__ counters)
{
    extern __shared__ int stack_mem[];
    int* stack = stack_mem + threadIdx.x;
    const uint32_t n_queue = W.count[qi];
    const uint32_t* __restrict__ queue = W.queue[qi];
    const int lane = (int)(threadIdx.x & 63u);
    const float tMin = .001f;

    int mode = WF_NEED;
    uint32_t slot = 0;
    DRay rw = {mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 1.f), 0.f}, rl_own = rw;
    DRay& rl = (F & RT_F_XFORM) ? rl_own : rw;      /* the ray in the group's frame: the world ray itself where no scene object is wrapped */
    V3 id = mk(0.f, 0.f, 0.f);
    uint32_t soff = 0u;                         /* wf4_sign_offsets(id): where in a node row this ray's NEAR planes are */
    HitLite best = {FLT_MAX, -1, -1, 0};
    int gidx = 0, gi = -1, ref = RT_REF_NONE, pleaf = RT_REF_NONE, sp = 0, li = 0;
    int ovf[WF_OVF_STACK];                      /* stack entries beyond WF_LDS_STACK (scratch) */
#define WF_PUSH(v) { const int v_ = (v); if (sp < WF_LDS_STACK) stack[sp * RT_BLOCK] = v_; else ovf[sp - WF_LDS_STACK] = v_; ++sp; }
    /* two loads in two branches: written as one conditional expression the compiler selects between the ADDRESSES and pops
     * with a flat_load, which goes through the vector memory path as well as the LDS */
#define WF_POP(dst) { dst = RT_REF_NONE; if (sp > 0) { --sp; if (sp < WF_LDS_STACK) dst = stack[sp * RT_BLOCK]; else dst = ovf[sp - WF_LDS_STACK]; } }
    /* the ray enters the gidx-th BVH group of the world list (GENERIC record of closest_hit: the group's own
     * wrapper chain on the world ray), or, past the last group, is finished: its hit goes to HBM */
#define WF_NEXT_GROUP                                                                                            \
    {                                                                                                            \
        if (gidx < W.n_groups) {                                                                                 \
            const int obj_ = W.groups[gidx];                                                                     \
            const DObj& o_ = S.objs[obj_];                                                                       \
            if (F & RT_F_XFORM) { rl = rw; apply_chain(S, o_, rl); }                                             \
            id = mk(1.f / rl.d.x, 1.f / rl.d.y, 1.f / rl.d.z);                                                   \
            soff = wf4_sign_offsets(id);                                                                         \
            mode = WF_BVH; gi = obj_ == S.world ? -1 : obj_; ref = o_.pad[0]; pleaf = RT_REF_NONE; sp = 0; li = 0; \
        } else {                                                                                                 \
            W.hit[slot] = make_float4(best.t, __int_as_float(best.obj), __int_as_float(best.grp), __int_as_float(best.aux)); \
            mode = WF_NEED;                                                                                      \
        }                                                                                                        \
    }
    PathRng g;
    rng_start(g, k0, k1, 0u, 0u);
    uint32_t bounce = 0;
    uint32_t n_rays = 0;
    bool drained = false;                       /* wave-uniform: the queue has nothing left */
    uint32_t chunk_next = 0, chunk_end = 0;     /* wave-uniform: this wave's claimed part of the queue */
    /* chunk size: about half of a wave's fair share of the queue, so that short queues still spread over
     * the whole grid, within [16, WF_CHUNK] */
    uint32_t chunk = n_queue / (gridDim.x * (RT_BLOCK / 64) * 2u);
    chunk = chunk < 16u ? 16u : (chunk > (uint32_t)WF_CHUNK ? (uint32_t)WF_CHUNK : chunk);
    /* XCD-affine claims (see the refill): chunk ids, chunk ids per stream (at least 8, so that every band has one) */
    const uint32_t af_chunks = (n_queue + chunk - 1u) / chunk;
    const uint32_t af_streams = W.n_slots / (W.n_pixels ? W.n_pixels : 1u);
    uint32_t af_cps = (af_chunks + (af_streams ? af_streams : 1u) - 1u) / (af_streams ? af_streams : 1u);
    af_cps = af_cps < 8u ? 8u : af_cps;
    int band = (int)(blockIdx.x & 7u), band_tried = 0;
#ifdef RT_STAMPS
    unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};   /* trips, busy lanes, node, prim, stalled lanes, trips with a node step, trips with a prim step, cycles */
    const unsigned long long dg_t0 = __builtin_amdgcn_s_memtime();
#endif

    while (true) {
        /* ---- refill: lanes without a ray take the next queue entries, a wave-wide batch at a time ---- */
        const unsigned long long need = __ballot(mode == WF_NEED);
        const int n_need = __popcll(need);
#ifdef RT_STAMPS_REFILL
        const bool rf_on = !drained && (n_need >= W.refill_min || n_need == 64);
        const unsigned long long rf_t0 = __builtin_amdgcn_s_memtime();
#endif
        if (!drained && (n_need >= W.refill_min || n_need == 64)) {
            /* The wave owns a chunk [chunk_next, chunk_end) of the queue and hands its entries to idle
             * lanes; one returning atomic on the shared head per WF_CHUNK rays (a single word sustains
             * ~90 M dequeues/s on this chip, far below the ray rate if every refill paid one). */
            if (chunk_next >= chunk_end && !W.affine) {
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(&W.count[2], chunk);
                chunk_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                chunk_end = chunk_next + chunk < n_queue ? chunk_next + chunk : n_queue;
                if (chunk_next >= n_queue) { drained = true; chunk_next = chunk_end = n_queue; }
            }
            /* XCD-affine claims.  The queue is (nearly) in slot order: stream after stream, inside a stream the pixels of
             * the frame row by row.  Chunk c therefore lies in image band (c mod cps) * 8 / cps of its stream, cps =
             * chunks per stream.  Band x has a head of its own (each on a cache line of its own) that enumerates the
             * chunks of band x over all streams, and the waves of XCD x (blockIdx mod 8) take from head x first: the
             * XCD's L2 then serves rays from one eighth of the image.  A band that is used up is followed by the band
             * with the most chunks left.  Every chunk id is handed out exactly once whatever the bands really contain
             * (the mapping band, j -> c is a bijection onto the chunk ids), so affinity is only ever a hint. */
            while (chunk_next >= chunk_end && W.affine && !drained) {            /* wave-uniform */
                const uint32_t lo = (uint32_t)(((unsigned long long)band * af_cps + 7u) >> 3);
                const uint32_t n_x = (uint32_t)((((unsigned long long)band + 1u) * af_cps + 7u) >> 3) - lo;
                uint32_t j = 0;
                if (lane == 0) j = atomicAdd(&W.count[WF_HEAD_WORDS * (1 + band)], 1u);
                j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
                const uint32_t seg = n_x ? j / n_x : af_chunks;
                const unsigned long long c = (unsigned long long)seg * af_cps + lo + (n_x ? j - seg * n_x : 0u);
                if (n_x && c < af_chunks) {
                    chunk_next = (uint32_t)c * chunk;
                    chunk_end = chunk_next + chunk < n_queue ? chunk_next + chunk : n_queue;
                } else {
                    /* this band is used up: go on with the one that has the most chunks left */
                    uint32_t most = 0u;
                    int where = -1;
                    for (int x = 0; x < 8; ++x) {
                        const uint32_t xlo = (uint32_t)(((unsigned long long)x * af_cps + 7u) >> 3);
                        const uint32_t xn = (uint32_t)((((unsigned long long)x + 1u) * af_cps + 7u) >> 3) - xlo;
                        const uint32_t full = af_chunks / af_cps, rest = af_chunks - full * af_cps;
                        const uint32_t total = full * xn + (rest > xlo ? (rest - xlo < xn ? rest - xlo : xn) : 0u);
                        const uint32_t hd = __hip_atomic_load(&W.count[WF_HEAD_WORDS * (1 + x)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t left = total > hd ? total - hd : 0u;
                        if (left > most) { most = left; where = x; }
                    }
                    where = __builtin_amdgcn_readfirstlane(where);
                    if (where < 0 || ++band_tried > 32) { drained = true; chunk_next = chunk_end = n_queue; }
                    else band = where;
                }
            }
            const uint32_t base = chunk_next;
            const uint32_t avail = chunk_end - chunk_next;
            chunk_next += (uint32_t)n_need < avail ? (uint32_t)n_need : avail;
            const uint32_t n_queue_w = chunk_end;     /* entries at or past the chunk's end wait for the next refill */
            if (mode == WF_NEED) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                const uint32_t e = base + rank;
                if (e < n_queue_w) {
                    slot = queue[e];
                    const float4 o = W.ray_o[slot], d = W.ray_d[slot];
                    rw.o = mk(o.x, o.y, o.z); rw.time = o.w;
                    rw.d = mk(d.x, d.y, d.z);
                    bounce = __float_as_uint(d.w);
                    g.pixel = wf_frame_pixel(W, slot);
                    if (F & RT_F_MEDIA) g.sample = __float_as_uint(W.att[slot].w);   /* the RNG address is only read by media */
                    best.t = FLT_MAX; best.obj = -1; best.grp = -1; best.aux = 0;
                    ++n_rays;
                    /* the plain records of the world list, for all the lanes of this batch together (one scalar
                     * record stream, as in the megakernel's closest_hit), then on to the first BVH group */
                    scan_program<(F & ~RT_F_BVH)>(S, W.prog, rw, tMin, best, W.bare ? nullptr : &g, bounce, stack);
                    gidx = 0;
                    WF_NEXT_GROUP
                }
            }
        }
#ifdef RT_STAMPS_REFILL
        if (rf_on) { dg[4] += __builtin_amdgcn_s_memtime() - rf_t0; dg[5] += 1; }      /* cycles inside refills, refills (instead of stalled lanes, node trips) */
#endif
        if (__ballot(mode != WF_NEED) == 0ull) {
            if (drained) break;
            continue;                            /* a batch that finished at once (no BVH group), or all 64 lanes idle */
        }

        /* ---- one trip ----
         * The kinds of step cost very differently (a 4-wide node ~130 VALU instructions, a triangle ~130) and
         * a trip pays for every kind any lane runs, so each kind should run with as many lanes as possible:
         *  - a lane that reaches a leaf does not stop to test it: the leaf is POSTPONED (`pleaf`) and the
         *    lane goes on with the next node from its stack; it can take a node step and a primitive step in
         *    the same trip, and only stalls when a second leaf turns up before the first is finished;
         *  - primitive steps run once WF_PRIM_BATCH lanes hold a postponed leaf, or the node steps run low.
         * The closest hit does not depend on the order in which a ray's leaves are tested (exact ties aside). */
        const bool in_bvh = mode == WF_BVH;
        const bool interior = in_bvh && ref >= 0;
        const bool has_leaf = in_bvh && pleaf != RT_REF_NONE;
        const int n_node = __popcll(__ballot(interior)), n_prim = __popcll(__ballot(has_leaf));
        const bool do_prim = n_prim >= WF_PRIM_BATCH || (n_prim > 0 && n_node < WF_NODE_LOW);
        const bool prim_step = has_leaf && do_prim;
#ifdef RT_STAMPS
        {
            const int nb = __popcll(__ballot(mode != WF_NEED));
            dg[0] += 1; dg[1] += nb; dg[2] += n_node; dg[3] += do_prim ? n_prim : 0; dg[6] += do_prim ? 1 : 0;
#ifndef RT_STAMPS_REFILL
            dg[4] += __popcll(__ballot(in_bvh && !interior && !prim_step));      /* lanes with a ray and nothing to do this trip */
            dg[5] += n_node > 0 ? 1 : 0;
#endif
        }
#endif
        /* the loads of both steps of the trip go out together */
        const int lfirst = (~pleaf) >> 4, lcount = ((~pleaf) & 15) + 1;
        const float4* src = reinterpret_cast<const float4*>(S.leafs + (prim_step ? lfirst + li : 0));
        /* the leaf row first (zero-filled where no primitive step runs: leaving the registers undefined there made the
         * allocator spill, final scene 3.72 -> 2.65 Grays/s), then the node row inside the node step's own block, which
         * needs no fill at all */
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0, q3 = q0;
        if (prim_step) { q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3]; }

        if (interior) {
            /* near / far planes picked by the ray's signs, as in the one-kernel form (WF_NODE4_TEST above) */
            WF_NODE4_TEST(rl.o, id, soff, tMin, best.t)
            /* 5-exchange sorting network on (t, ref), nearest first */
#define WF_CSWAP(ta, ra, tb, rb) { const bool sw_ = tb < ta; const float tt_ = sw_ ? tb : ta; const int rr_ = sw_ ? rb : ra; tb = sw_ ? ta : tb; rb = sw_ ? ra : rb; ta = tt_; ra = rr_; }
            WF_CSWAP(t0, r0, t1, r1) WF_CSWAP(t2, r2, t3, r3) WF_CSWAP(t0, r0, t2, r2) WF_CSWAP(t1, r1, t3, r3) WF_CSWAP(t1, r1, t2, r2)
#undef WF_CSWAP
            /* farthest first onto the stack, nearest entered next */
            if (r3 != RT_REF_NONE) WF_PUSH(r3)
            if (r2 != RT_REF_NONE) WF_PUSH(r2)
            if (r1 != RT_REF_NONE) WF_PUSH(r1)
            ref = r0;
            if (ref == RT_REF_NONE) WF_POP(ref)
        }
        if (prim_step) {
            const uint32_t head = __float_as_uint(q0.x);
            const int oi = __float_as_int(q0.z);
            const float f[13] = {q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            float t; int aux;
            bool h;
            if (!(F & RT_F_XFORM) || obj_op_count(head) == 0) h = prim_t<F>(obj_type(head), f, rl, tMin, best.t, t, aux);
            else h = leaf_t<F>(S, S.objs[oi], rl, tMin, best.t, t, aux);
            /* word 1 of the row (the material index, never a NaN pattern) is not needed here, but with a hole at word 1 the
             * compiler fetches the row as dword + 3 x dwordx4 at offsets 8, 24, 40 + dwordx2: five lane loads, three of them
             * misaligned, in a kernel whose trips last as long as their lane loads.  Comparing the word with itself as a
             * float cannot be folded away, so the row arrives as four aligned dwordx4. */
#ifndef WF_LEAF_NARROW
            h = h && q0.y == q0.y;
#endif
            if (h) { best.t = t; best.obj = oi; best.grp = gi; best.aux = aux; }
            if (++li == lcount) { li = 0; pleaf = RT_REF_NONE; }
        }
        if (in_bvh) {
            /* a leaf reference in `ref` moves to the postponed slot as soon as that is free */
            if (pleaf == RT_REF_NONE && ref < 0 && ref != RT_REF_NONE) { pleaf = ref; li = 0; WF_POP(ref) }
            if (ref == RT_REF_NONE && pleaf == RT_REF_NONE) { ++gidx; WF_NEXT_GROUP }   /* the group is done */
        }
    }
#undef WF_NEXT_GROUP
    unsigned long long r = n_rays;
    for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off);
    if (lane == 0) atomicAdd(&counters[0], r);
#ifdef RT_STAMPS
    dg[7] = __builtin_amdgcn_s_memtime() - dg_t0;
    if (lane == 0) for (int q = 0; q < 8; ++q) atomicAdd(&counters[4 + (q < 6 ? q : q + 4)], dg[q]);     /* rt_debug_stamps overlays words 6..9 */
#endif
}

/* ---------------------------------------------------------------------- logic -- */
template <uint32_t F>
__device__ __forceinline__ void wf_start_sample(const DScene& S, const WfState& W, uint32_t slot, uint32_t sample, int Wd, int Hd,
                                                uint32_t k0, uint32_t k1, float4& o4, float4& d4, float4& a4)
{
    uint32_t ui, uj;
    const uint32_t pixel = wf_frame_pixel(W, slot, ui, uj);
    const int i = (int)ui, j = (int)uj;
    PathRng g;
    rng_start(g, k0, k1, pixel, sample);
    rt_u32x4 cw = rng_block(g, RT_RNG_CAMERA, 0u, 0u);
    float u = float(i + rt_u32_to_uniform(cw.x)) / float(Wd);
    float v = float(j + rt_u32_to_uniform(cw.y)) / float(Hd);
    DRay ray = generate_ray<F>(cam_fetch(S), u, v, rt_u32_to_uniform(cw.z), g);
    o4 = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.time);
    d4 = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(0u));
    a4 = make_float4(1.f, 1.f, 1.f, __uint_as_float(sample));
}

/* wave-aggregated append of `slot` to a queue */
__device__ __forceinline__ void wf_push(uint32_t* __restrict__ queue, uint32_t* __restrict__ count, bool want, uint32_t slot)
{
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return;
    const int lane = (int)(threadIdx.x & 63u);
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader);
    if (want) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        queue[base + rank] = slot;
    }
}

/* first launch of a render call: every slot starts its first sample.  There are never more streams than samples,
 * so every slot is alive (unless max_depth is 0) and the first queue is simply the slots in order: no atomics
 * (one per wave on a single word took 11 ms for 66 M slots); the host sets its length. */
template <uint32_t F>
__global__ __launch_bounds__(RT_BLOCK) void wf_init_kernel(DScene S, WfState W, int Wd, int Hd, uint32_t s_begin, uint32_t s_end,
                                                           uint32_t k0, uint32_t k1, int max_depth, unsigned long long* __restrict__ counters)
{
    const uint32_t slot = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (slot < W.n_slots) {
        const uint32_t k = W.stream0 + slot / W.n_pixels;
        const uint32_t s = s_begin + k;
        W.sum[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (max_depth > 0) {
            float4 o4, d4, a4;
            wf_start_sample<F>(S, W, slot, s, Wd, Hd, k0, k1, o4, d4, a4);
            W.ray_o[slot] = o4; W.ray_d[slot] = d4; W.att[slot] = a4;
            W.queue[0][slot] = slot;
        }
    }
    /* samples counter: every sample of the range is rendered (black when max_depth is 0) */
    if (slot == 0 && W.stream0 == 0) atomicAdd(&counters[1], (unsigned long long)W.n_pixels * (unsigned long long)(s_end - s_begin));
}

/* one queued path through one bounce; returns whether its slot goes on (with the same path or its next sample).
 * Called by all 64 lanes of a wave together (the ball draw is wave-wide). */
template <uint32_t F, int BG>
__device__ __forceinline__ bool wf_logic_one(const DScene& S, const WfState& W, bool active, uint32_t slot, int Wd, int Hd, uint32_t s_end,
                                             uint32_t k0, uint32_t k1, int max_depth, int* slots)
{

    DRay ray = {mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 1.f), 0.f};
    /* colour()'s running emission L (main.cu:64) is not part of the stored state: only DiffuseLight emits
     * (Material.cuh:121-131; Material::emitted returns black) and it never scatters, so L is still zero
     * whenever a path goes on to its next bounce (an attenuation that has overflowed to inf, which takes albedos
     * above 1, would make it NaN in the megakernel: 0 * inf) */
    V3 att = mk(1.f, 1.f, 1.f), emi = mk(0.f, 0.f, 0.f);
    HitLite hl = {FLT_MAX, -1, -1, 0};
    uint32_t sample = 0;
    int depth = 0;
    PathRng g;
    rng_start(g, k0, k1, 0u, 0u);
    if (active) {
        const float4 o = W.ray_o[slot], d = W.ray_d[slot], a = W.att[slot], h = W.hit[slot];
        ray.o = mk(o.x, o.y, o.z); ray.time = o.w; ray.d = mk(d.x, d.y, d.z);
        depth = (int)__float_as_uint(d.w);
        att = mk(a.x, a.y, a.z); sample = __float_as_uint(a.w);
        hl.t = h.x; hl.obj = __float_as_int(h.y); hl.grp = __float_as_int(h.z); hl.aux = __float_as_int(h.w);
        rng_start(g, k0, k1, wf_frame_pixel(W, slot), sample);
    }
    /* one iteration of colour() / colourUnlit(), main.cu:53-70 / :77-98 — same body as render_kernel */
    const bool hit = active && hl.obj >= 0;
    HitRec rec;
    rec.t = 0.f; rec.u = 0.f; rec.v = 0.f; rec.p = mk(0.f, 0.f, 0.f); rec.n = mk(0.f, 0.f, 0.f);
    rec.material = 0; rec.prim_id = -1; rec.front = 0;
    int mtype = RT_MAT_LIGHT;
    if (hit) {
        rebuild<F, false>(S, ray, hl, .001f, rec);
        mtype = S.mats[rec.material].type;
    }
    const bool need_ball = hit && mtype != RT_MAT_LIGHT && mtype != RT_MAT_GLASS;
    const V3 ball = random_in_unit_sphere_wave(g, (uint32_t)depth, need_ball, slots);
    bool alive = false;
    if (active) {
        V3 result = mk(0.f, 0.f, 0.f);
        bool done;
        if (hit) {
            V3 attenuation, emitted;
            DRay scattered;
            if (scatter<F>(S, ray, rec, g, (uint32_t)depth, ball, attenuation, scattered, emitted)) {
                att = att * attenuation;
                if (BG == RT_BG_BLACK) emi = emi + emitted * att;
                ray = scattered;
                done = false;
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
        if (!done) {
            W.ray_o[slot] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.time);
            W.ray_d[slot] = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float((uint32_t)depth));
            W.att[slot] = make_float4(att.x, att.y, att.z, __uint_as_float(sample));
            alive = true;
        } else {
            float4 sm = W.sum[slot];
            sm.x += result.x; sm.y += result.y; sm.z += result.z;
            W.sum[slot] = sm;
            const uint32_t next = sample + W.streams;
            if (next < s_end && next > sample) {
                float4 o4, d4, a4;
                wf_start_sample<F>(S, W, slot, next, Wd, Hd, k0, k1, o4, d4, a4);
                W.ray_o[slot] = o4; W.ray_d[slot] = d4; W.att[slot] = a4;
                alive = true;
            }
        }
    }
    return alive;
}

/* The queue stays in SLOT ORDER from iteration to iteration.  A logic block runs WF_LOGIC_TILES consecutive tiles of
 * 256 queue entries, moves its survivors to the front of its own span of the queue, in entry order, and leaves their
 * number in scan[block]; wf_scan_kernel (one block) turns the numbers into offsets and wf_move_kernel copies every
 * span to its place in the other queue.  No atomics, and a position in the queue keeps saying where in the frame its
 * slot is.  (The first version appended each block's survivors to the other queue with one atomic per block, in the
 * order in which the blocks happened to finish: over hundreds of iterations the queue drifted away from slot order,
 * the state accesses lost their coalescing, and calls had to be cut into parts that rebuilt the queue.) */
#ifndef WF_LOGIC_TILES
#define WF_LOGIC_TILES 4
#endif
#define WF_SPAN (RT_BLOCK * WF_LOGIC_TILES)
/* waves per SIMD the logic kernel is compiled for.  The all-feature variant needs 117 VGPRs; at 5 waves it fits 96 with
 * 8 bytes spilled and at 6 waves 80 with 48 bytes, and both are SLOWER (final scene 3.89 -> 3.75 / 3.68 Grays/s, Cornell +
 * sphere cluster 4.76 -> 4.61; the triangle variant has 70 and does not care): the kernel waits for its state rows,
 * but more of its waves only take wave slots and memory requests away from the intersect kernels of the other pools */
#ifndef WF_LOGIC_WAVES
#define WF_LOGIC_WAVES 4
#endif
template <uint32_t F, int BG>
__global__ __launch_bounds__(RT_BLOCK, WF_LOGIC_WAVES) void wf_logic_kernel(DScene S, WfState W, int qi, int Wd, int Hd, uint32_t s_end,
                                                            uint32_t k0, uint32_t k1, int max_depth)
{
    __shared__ int slot_mem[RT_BLOCK];
    __shared__ uint32_t keep_slot[WF_LOGIC_TILES][RT_BLOCK];
    __shared__ unsigned long long keep_mask[WF_LOGIC_TILES][RT_BLOCK / 64];
    int* slots = slot_mem + (threadIdx.x & ~63u);
    const uint32_t n_queue = W.count[qi];
    const uint32_t first = blockIdx.x * WF_SPAN;
    if (first >= n_queue) return;                             /* whole block beyond the queue */
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    uint32_t* __restrict__ queue = W.queue[qi];
    /* the queue entries of all the tiles first: one round trip instead of one per tile in front of the state loads */
    uint32_t slots_of[WF_LOGIC_TILES];
#pragma unroll
    for (int tile = 0; tile < WF_LOGIC_TILES; ++tile) {
        const uint32_t e = first + tile * RT_BLOCK + threadIdx.x;
        slots_of[tile] = e < n_queue ? queue[e] : 0u;
    }
    static_assert(WF_LOGIC_TILES == 4, "the select below is written for four tiles");
    for (int tile = 0; tile < WF_LOGIC_TILES; ++tile) {       /* stays a loop: the body is the whole bounce */
        const uint32_t e = first + tile * RT_BLOCK + threadIdx.x;
        const bool active = e < n_queue;
        const uint32_t slot = tile == 0 ? slots_of[0] : (tile == 1 ? slots_of[1] : (tile == 2 ? slots_of[2] : slots_of[3]));
        bool alive = false;
        if (first + tile * RT_BLOCK < n_queue)                /* block-uniform: skip tiles beyond the queue */
            alive = wf_logic_one<F, BG>(S, W, active, slot, Wd, Hd, s_end, k0, k1, max_depth, slots);
        keep_slot[tile][threadIdx.x] = slot;
        const unsigned long long m = __ballot(alive);
        if (lane == 0) keep_mask[tile][wave] = m;
    }
    __syncthreads();                                          /* every entry of the span has been read */
    uint32_t before = 0;                                      /* survivors of earlier (tile, wave) pairs */
    for (int t = 0; t < WF_LOGIC_TILES; ++t)
        for (int w = 0; w < RT_BLOCK / 64; ++w) {
            const unsigned long long m = keep_mask[t][w];
            if (w == wave && ((m >> lane) & 1ull)) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                queue[first + before + rank] = keep_slot[t][threadIdx.x];
            }
            before += (uint32_t)__popcll(m);
        }
    if (threadIdx.x == 0) W.scan[blockIdx.x] = before;
}

/* One block.  scan[0 .. n) holds the survivors of the n spans of queue[qi]; afterwards scan[b] is where span b's
 * survivors start in the other queue, scan[n] and count[qi ^ 1] their total.  Also rewinds the intersect kernel's fetch
 * heads for the next iteration. */
__global__ __launch_bounds__(1024) void wf_scan_kernel(WfState W, int qi)
{
    __shared__ uint32_t sums[1024];
    const uint32_t n_queue = W.count[qi];
    const uint32_t n = (n_queue + WF_SPAN - 1) / WF_SPAN;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t b0 = threadIdx.x * per < n ? threadIdx.x * per : n, b1 = b0 + per < n ? b0 + per : n;
    uint32_t mine = 0;
    for (uint32_t b = b0; b < b1; ++b) mine += W.scan[b];
    sums[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {          /* Hillis-Steele inclusive scan */
        const uint32_t v = threadIdx.x >= off ? sums[threadIdx.x - off] : 0u;
        __syncthreads();
        sums[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = sums[threadIdx.x] - mine;                  /* exclusive prefix of this thread's range */
    for (uint32_t b = b0; b < b1; ++b) { const uint32_t v = W.scan[b]; W.scan[b] = run; run += v; }
    if (threadIdx.x == 1023u) { W.scan[n] = sums[1023]; W.count[qi ^ 1] = sums[1023]; W.count[2] = 0u; }
    if (threadIdx.x < 8) W.count[WF_HEAD_WORDS * (1 + threadIdx.x)] = 0u;     /* the bands' fetch heads */
}
__global__ __launch_bounds__(RT_BLOCK) void wf_move_kernel(WfState W, int qi)
{
    const uint32_t n_queue = W.count[qi];
    const uint32_t first = blockIdx.x * WF_SPAN;
    if (first >= n_queue) return;
    const uint32_t to = W.scan[blockIdx.x], n = W.scan[blockIdx.x + 1] - to;
    const uint32_t* __restrict__ in = W.queue[qi] + first;
    uint32_t* __restrict__ out = W.queue[qi ^ 1] + to;
    for (uint32_t i = threadIdx.x; i < n; i += RT_BLOCK) out[i] = in[i];
}

/* accum[pixel] += the K stream sums of the pixel, in stream order */
__global__ __launch_bounds__(RT_BLOCK) void wf_finalize_kernel(WfState W, float* __restrict__ accum)
{
    const uint32_t p = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (p >= W.n_pixels) return;
    float x = 0.f, y = 0.f, z = 0.f;
    for (uint32_t k = 0; k < W.streams; ++k) {
        const float4 s = W.sum[(size_t)k * W.n_pixels + p];
        x += s.x; y += s.y; z += s.z;
    }
    float* a = accum + 3 * (size_t)wf_frame_pixel(W, p);
    a[0] += x; a[1] += y; a[2] += z;
}

#endif
