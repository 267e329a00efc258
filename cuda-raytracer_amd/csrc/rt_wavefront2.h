/*
 * rt_wavefront2.h — the intersect kernel of the wavefront form, second design (round 3).
 *
 * wf_intersect_kernel (rt_wavefront.h) ran with half its lanes idle: per trip, of 64 lanes, 36 took the node step, 13 only
 * held a postponed leaf and 15 waited for the next refill (a refill ran the world list's plain records for the lanes it
 * filled, so it waited for a batch of 40 idle lanes), and the primitive step ran for 29 lanes every second trip.  Three
 * changes, all of them about WHO does a piece of work, none about the arithmetic that decides a hit:
 *
 *  R  Rays are staged.  A wave prefetches its next queue entries and their rays into LDS (global_load_lds, one pipeline
 *     stage per trip: slots -> rays -> scan) and runs the plain records of the world list for a whole batch of staged
 *     rays at once, at full width, BEFORE any lane needs them.  Handing a staged ray to a lane is then a few LDS reads, so
 *     it is done for a handful of idle lanes instead of forty.
 *  Q  Leaves are queued.  A lane that reaches a leaf does not test it and does not keep it: it appends (lane, item) to a
 *     queue of the WAVE in LDS and goes on walking.  When 64 entries are waiting, all 64 lanes test one entry each: the
 *     ray comes from its owner's registers (ds_bpermute), the result goes to the owner's mailbox in LDS with a 64-bit
 *     atomic min on (t, object).  A primitive step therefore runs with 64 lanes whatever the rays are doing, and no lane
 *     ever stalls on a second leaf.  A ray whose traversal is finished waits until the queue has passed its last entry.
 *     Culling with a bound that lags by the entries still queued is conservative (a stale bound only visits more).
 *  F  The slab test is three fused multiply-adds per plane pair instead of subtract + multiply, on near / far planes
 *     picked by the ray's octant, with the rounding of the precomputed -o/d terms covered by a per-ray widening.  Box
 *     culling never decides a result: every primitive behind a box is still tested with the reference's own arithmetic.
 *
 * The mailbox of a ray holds its closest hit so far over everything, as one 64-bit word: t bits << 32 | group ordinal << 29 |
 * (object + 1) << 3 | face, group ordinal 0 = a plain record of the world list.  Of two hits at exactly the same t the lower
 * word wins: a plain record before any BVH primitive, an earlier group before a later one (both as in the first kernel),
 * and inside a group the lower object index (the first kernel kept the one it met first).  north_star leaves ties free.
 */
#ifndef RT_WAVEFRONT2_H
#define RT_WAVEFRONT2_H
#include "rt_wavefront.h"

/* Every variant runs 4 waves per SIMD (128 VGPRs: at 96 the triangle variant spilled 14 registers), i.e. 40 KB of LDS per
 * block of four waves: 64 staged rays of 44 B per wave, mailboxes, a leaf queue of 256 entries per wave, the table of the
 * BVH groups, the world ray of every lane where a scene wraps anything (a ray that leaves a wrapped group enters the
 * next one from the WORLD ray: kept in LDS, not fetched again), and the traversal stack: 20 entries per lane, 16 in the
 * variants that keep the world ray. */
constexpr uint32_t wf2_stage(uint32_t) { return 64u; }      /* staged rays per wave (a ring) */
constexpr uint32_t wf2_qcap(uint32_t) { return 256u; }      /* leaf-queue entries per wave (a ring) */
constexpr int wf2_waves(uint32_t) { return 4; }
constexpr int wf2_lds_stack(uint32_t F) { return (F & RT_F_XFORM) ? 16 : WF_LDS_STACK; }     /* stack entries per lane kept in LDS; deeper ones go to scratch */
/* entries of the stack's overflow array in scratch (the host sends scenes whose trees need more to the first kernel) */
constexpr int wf2_ovf(uint32_t) { return WF_OVF_STACK + 4; }
/* The kernel is built for the two feature sets the target scenes use (triangle mesh + area light; everything but triangles
 * and list-scanned groups).  The all-feature sets, triangles and spheres and media in one kernel, need 130 more registers
 * than a wave has at this occupancy; scenes that need them keep the first kernel. */
constexpr bool wf2_built(uint32_t F) { return !((F & RT_F_TRI) && (F & RT_F_SPHERE)); }
#define WF2_ITEM_BITS 26        /* a queue entry is lane << 26 | item: scenes with 2^26 BVH items or more use the first kernel */
#ifndef WF2_NODE_LOW
#define WF2_NODE_LOW 16         /* fewer walking lanes than this: test what is queued */
#define WF2_WAIT_HI 8           /* this many lanes waiting for their last leaves: test what is queued */
#endif
#define WF2_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define WF2_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

/* per-axis constants of the slab test for one ray in one frame: t_near = b_near * i + lo, t_far = b_far * i + hi */
struct Slab3 { V3 i, lo, hi; };
__device__ __forceinline__ void slab_axis(float o, float d, float dmax, float& i, float& lo, float& hi)
{
    /* A direction component of (nearly) zero is given the size 2^-40 of the largest one, keeping its sign: for the box
     * test the ray is tilted by 1e-12 of its length, far inside the padding of the host's boxes (lower.cpp pad_box: 1e-5
     * relative), and the axis keeps culling.  Dropping such an axis from the test instead (the first version) made a ray
     * with d.x == 0 walk a whole strip of the terrain: one bounce ray in a million, 50 000 trips each, every launch
     * waiting for its last few. */
    const float floor_ = dmax * 9.094947e-13f;
    const float dd = fabsf(d) < floor_ ? __builtin_copysignf(floor_, d) : d;
    i = __builtin_amdgcn_rcpf(dd);
    const float oi = o * i;
    /* |oi| 2^-22 covers the roundings of o * i and of the sum below */
    const float e = fabsf(oi) * 2.3841858e-7f;
    const bool ok = fabsf(oi) < 1e37f && fabsf(i) < 1e37f;
    /* What is left: an origin so far out that o / d overflows (the axis is left out of the test: i = 0, t in (-inf, +inf)),
     * and a ray that is not a number at all (a path that left a rect through the reference's t = NaN hit,
     * Rectangle.cuh:27-33) or has no direction: it meets no box, as in slab(), where every comparison with a NaN is false. */
    const bool dead = !(fabsf(o) <= FLT_MAX) || !(fabsf(d) <= FLT_MAX) || !(dmax > 0.f);
    lo = ok ? -oi - e : (dead ? INFINITY : -INFINITY);
    hi = ok ? -oi + e : (dead ? -INFINITY : INFINITY);
    i = ok ? i : 0.f;
}
__device__ __forceinline__ void slab_setup(const DRay& r, Slab3& s)
{
    const float dmax = fmaxf(fmaxf(fabsf(r.d.x), fabsf(r.d.y)), fabsf(r.d.z));
    slab_axis(r.o.x, r.d.x, dmax, s.i.x, s.lo.x, s.hi.x);
    slab_axis(r.o.y, r.d.y, dmax, s.i.y, s.lo.y, s.hi.y);
    slab_axis(r.o.z, r.d.z, dmax, s.i.z, s.lo.z, s.hi.z);
}
/* one child box b = (min xyz, max xyz); sx/sy/sz = the ray runs towards negative x/y/z.  Conservative like slab()
 * (rt_device.h): the far bound is widened by a few ulps for the 1-ulp reciprocal and the roundings of the two fmas. */
__device__ __forceinline__ bool slab2(const float* b, const Slab3& s, bool sx, bool sy, bool sz, float tMin, float tMax, float& tnear)
{
    const float nx = sx ? b[3] : b[0], fx = sx ? b[0] : b[3];
    const float ny = sy ? b[4] : b[1], fy = sy ? b[1] : b[4];
    const float nz = sz ? b[5] : b[2], fz = sz ? b[2] : b[5];
    const float tn = fmaxf(fmaxf(__builtin_fmaf(nx, s.i.x, s.lo.x), __builtin_fmaf(ny, s.i.y, s.lo.y)), fmaxf(__builtin_fmaf(nz, s.i.z, s.lo.z), tMin));
    const float tf = fminf(fminf(__builtin_fmaf(fx, s.i.x, s.hi.x), __builtin_fmaf(fy, s.i.y, s.hi.y)), fminf(__builtin_fmaf(fz, s.i.z, s.hi.z), tMax));
    tnear = tn;
    return tn <= tf * 1.0000012f;
}

/* one claim of the ray queue for a wave: sets [chunk_next, chunk_end) or `drained` (see wf_intersect_kernel for the
 * XCD-affine form: band x's head enumerates the chunks of image band x over all streams, every chunk id is handed
 * out exactly once) */
struct WfClaim {
    uint32_t n_queue, chunk, af_chunks, af_cps;
    uint32_t chunk_next, chunk_end;
    int band, band_tried;
    bool drained;
};
__device__ __forceinline__ void wf_claim_init(WfClaim& c, const WfState& W, uint32_t n_queue)
{
    c.n_queue = n_queue;
    uint32_t chunk = n_queue / (gridDim.x * (RT_BLOCK / 64) * 2u);
    c.chunk = chunk < 16u ? 16u : (chunk > (uint32_t)WF_CHUNK ? (uint32_t)WF_CHUNK : chunk);
    c.af_chunks = (n_queue + c.chunk - 1u) / c.chunk;
    const uint32_t streams = W.n_slots / (W.n_pixels ? W.n_pixels : 1u);
    uint32_t cps = (c.af_chunks + (streams ? streams : 1u) - 1u) / (streams ? streams : 1u);
    c.af_cps = cps < 8u ? 8u : cps;
    c.chunk_next = c.chunk_end = 0u;
    c.band = (int)(blockIdx.x & 7u);
    c.band_tried = 0;
    c.drained = n_queue == 0u;
}
__device__ __forceinline__ void wf_claim_next(WfClaim& c, const WfState& W, int lane)
{
    if (!W.affine) {
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(&W.count[2], c.chunk);
        c.chunk_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
        c.chunk_end = c.chunk_next + c.chunk < c.n_queue ? c.chunk_next + c.chunk : c.n_queue;
        if (c.chunk_next >= c.n_queue) { c.drained = true; c.chunk_next = c.chunk_end = c.n_queue; }
        return;
    }
    while (c.chunk_next >= c.chunk_end && !c.drained) {            /* wave-uniform */
        const uint32_t lo = (uint32_t)(((unsigned long long)c.band * c.af_cps + 7u) >> 3);
        const uint32_t n_x = (uint32_t)((((unsigned long long)c.band + 1u) * c.af_cps + 7u) >> 3) - lo;
        uint32_t j = 0;
        if (lane == 0) j = atomicAdd(&W.count[WF_HEAD_WORDS * (1 + c.band)], 1u);
        j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
        const uint32_t seg = n_x ? j / n_x : c.af_chunks;
        const unsigned long long id = (unsigned long long)seg * c.af_cps + lo + (n_x ? j - seg * n_x : 0u);
        if (n_x && id < c.af_chunks) {
            c.chunk_next = (uint32_t)id * c.chunk;
            c.chunk_end = c.chunk_next + c.chunk < c.n_queue ? c.chunk_next + c.chunk : c.n_queue;
        } else {
            uint32_t most = 0u;
            int where = -1;
            for (int x = 0; x < 8; ++x) {
                const uint32_t xlo = (uint32_t)(((unsigned long long)x * c.af_cps + 7u) >> 3);
                const uint32_t xn = (uint32_t)((((unsigned long long)x + 1u) * c.af_cps + 7u) >> 3) - xlo;
                const uint32_t full = c.af_chunks / c.af_cps, rest = c.af_chunks - full * c.af_cps;
                const uint32_t total = full * xn + (rest > xlo ? (rest - xlo < xn ? rest - xlo : xn) : 0u);
                const uint32_t hd = __hip_atomic_load(&W.count[WF_HEAD_WORDS * (1 + x)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t left = total > hd ? total - hd : 0u;
                if (left > most) { most = left; where = x; }
            }
            where = __builtin_amdgcn_readfirstlane(where);
            if (where < 0 || ++c.band_tried > 32) { c.drained = true; c.chunk_next = c.chunk_end = c.n_queue; }
            else c.band = where;
        }
    }
}

enum { WF2_NEED = 0, WF2_BVH = 1 };
#define WF2_MAX_GROUPS 7        /* group ordinals 1..7 fit the three bits of the mailbox word */
#define WF2_OBJ_BITS 25

/* The mailbox word of a hit: t bits << 32 | group ordinal << 29 | class << 28 | key << 3 | face.  An atomic min over such
 * words keeps, of two hits at exactly the same t, what the reference's list scan (HitableList.cuh:17-32) would keep where
 * that can be said without knowing the order of the tests: a plain record of the world list before any BVH primitive and an
 * earlier group before a later one (ordinal, 0 = plain); inside a group a rect or box before a sphere or triangle, the
 * LATER one of two rects / boxes (their range test is inclusive, Rectangle.cuh:27: the later one replaces the earlier)
 * and the EARLIER one of two spheres / triangles (strict, Sphere.cuh:35): class 0 = inclusive with key = 2^25 - 1 - (object +
 * 1), class 1 = strict with key = object + 1. */
__device__ __forceinline__ unsigned long long wf2_word(float t, int ordinal, int obj, int aux, bool inclusive)
{
    const uint32_t k = (uint32_t)obj + 1u, key = inclusive ? ((1u << WF2_OBJ_BITS) - 1u) - k : k;
    return ((unsigned long long)__float_as_uint(t) << 32) |
           (unsigned long long)(((uint32_t)ordinal << 29) | (inclusive ? 0u : 1u << 28) | (key << 3) | (uint32_t)aux);
}
__device__ __forceinline__ int wf2_word_obj(uint32_t pay)
{
    const uint32_t key = (pay >> 3) & ((1u << WF2_OBJ_BITS) - 1u);
    return (int)(((pay >> 28) & 1u) ? key : ((1u << WF2_OBJ_BITS) - 1u) - key) - 1;
}

/* what NEXT_GROUP needs of a BVH group's row, kept in LDS for the block (the rows sit behind two dependent loads) */
struct Wf2Group { int32_t obj, root; uint32_t head; float op0[3], op1[3]; int32_t op_begin; };

template <uint32_t F>
__global__ __launch_bounds__(RT_BLOCK, wf2_waves(F)) void wf_intersect2_kernel(DScene S, WfState W, int qi, uint32_t k0, uint32_t k1,
                                                                 unsigned long long* __restrict__ counters)
{
    extern __shared__ int stack_mem[];                                  /* n_stack x RT_BLOCK ints */
    constexpr uint32_t STAGE = wf2_stage(F), BATCH = STAGE / 2u, QCAP = wf2_qcap(F), NW = RT_BLOCK / 64;
    constexpr int LDS_STACK = wf2_lds_stack(F);
    __shared__ float4 st_o[NW * STAGE], st_d[NW * STAGE];               /* staged rays: origin | time, direction | bounce */
    __shared__ unsigned long long st_b[NW * STAGE];                     /* ... and their closest hit over the plain records, as a mailbox word */
    __shared__ uint32_t st_slot[NW * STAGE];
    __shared__ uint32_t st_smp[(F & RT_F_MEDIA) ? NW * STAGE : 1];
    __shared__ unsigned long long mail[RT_BLOCK];                       /* per lane: the closest hit so far of the lane's ray */
    __shared__ uint32_t lq_mem[NW * QCAP];
    __shared__ Wf2Group groups[WF2_MAX_GROUPS];
    __shared__ float world_ray[(F & RT_F_XFORM) ? 7 * RT_BLOCK : 1];    /* o.xyz, time, d.xyz of every lane's ray, component-major */
    int* stack = stack_mem + threadIdx.x;
    const uint32_t wave_base = threadIdx.x & ~63u;
    const uint32_t stage_base = (threadIdx.x >> 6) * STAGE;
    uint32_t* lq = lq_mem + (threadIdx.x >> 6) * QCAP;
    const uint32_t* __restrict__ queue = W.queue[qi];
    const int lane = (int)(threadIdx.x & 63u);
    const float tMin = .001f;
    uint32_t* my_mail = reinterpret_cast<uint32_t*>(&mail[threadIdx.x]);

    if ((int)threadIdx.x < W.n_groups && threadIdx.x < WF2_MAX_GROUPS) {
        const int obj = W.groups[threadIdx.x];
        const DObj& o = S.objs[obj];
        Wf2Group g;
        g.obj = obj; g.root = o.pad[0]; g.head = o.head; g.op_begin = o.op_begin;
        for (int k = 0; k < 3; ++k) { g.op0[k] = o.op0[k]; g.op1[k] = o.op1[k]; }
        groups[threadIdx.x] = g;
    }
    __syncthreads();

    WfClaim cl;
    wf_claim_init(cl, W, W.count[qi]);
    /* wave-uniform pipeline state */
    uint32_t st_head = 0u, st_tail = 0u;           /* staged rays handed out / ready */
    int pf_state = 0;                              /* 0 idle, 1 slots in flight, 2 rays in flight */
    uint32_t pf_n = 0u, pf_pos = 0u;
    uint32_t q_head = 0u, q_tail = 0u;             /* leaf queue */

    int mode = WF2_NEED;
    uint32_t slot = 0u;
    DRay rl = {mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 1.f), 0.f};     /* the ray in the frame of the group it is walking */
    Slab3 sl = {mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 0.f)};
    int gidx = 0, ref = RT_REF_NONE, sp = 0, li = 0;
    uint32_t mark = 0u;                            /* the queue position behind this ray's last entry */
    int ovf[wf2_ovf(F)];
#define WF2_PUSH(v) { const int v_ = (v); if (sp < LDS_STACK) stack[sp * RT_BLOCK] = v_; else ovf[sp - LDS_STACK] = v_; ++sp; }
#define WF2_POP(dst) { dst = RT_REF_NONE; if (sp > 0) { --sp; if (sp < LDS_STACK) dst = stack[sp * RT_BLOCK]; else dst = ovf[sp - LDS_STACK]; } }
    /* enter the gidx-th BVH group of the world list (the group's wrapper chain applied to the world ray), or, past the
     * last group, hand in the hit.  Everything it reads is in LDS. */
#define WF2_NEXT_GROUP(FIRST)                                                                                    \
    {                                                                                                            \
        if (gidx < W.n_groups) {                                                                                 \
            const Wf2Group& g_ = groups[gidx];                                                                   \
            if (F & RT_F_XFORM) {                                                                                \
                if (!(FIRST)) { const float* w_ = world_ray + threadIdx.x;                                       \
                                rl.o = mk(w_[0], w_[RT_BLOCK], w_[2 * RT_BLOCK]); rl.time = w_[3 * RT_BLOCK];    \
                                rl.d = mk(w_[4 * RT_BLOCK], w_[5 * RT_BLOCK], w_[6 * RT_BLOCK]); }               \
                DObj o_;                                                                                         \
                o_.head = g_.head; o_.op_begin = g_.op_begin;                                                    \
                o_.op0[0] = g_.op0[0]; o_.op0[1] = g_.op0[1]; o_.op0[2] = g_.op0[2];                             \
                o_.op1[0] = g_.op1[0]; o_.op1[1] = g_.op1[1]; o_.op1[2] = g_.op1[2];                             \
                apply_chain(S, o_, rl);                                                                          \
            }                                                                                                    \
            slab_setup(rl, sl);                                                                                  \
            mode = WF2_BVH; ref = g_.root; sp = 0; li = 0; mark = q_head;                                        \
        } else {                                                                                                 \
            const unsigned long long mm_ = mail[threadIdx.x];                                                    \
            const uint32_t pay_ = (uint32_t)mm_;                                                                 \
            const int ord_ = (int)(pay_ >> 29);                                                                  \
            int grp_ = -1;                                                                                       \
            if (ord_ > 0) { const int go_ = groups[ord_ - 1].obj; grp_ = go_ == S.world ? -1 : go_; }            \
            W.hit[slot] = make_float4(__uint_as_float((uint32_t)(mm_ >> 32)), __int_as_float(wf2_word_obj(pay_)), \
                                      __int_as_float(grp_), __int_as_float((int)(pay_ & 7u)));                   \
            mode = WF2_NEED;                                                                                     \
        }                                                                                                        \
    }
    uint32_t n_rays = 0;
#ifdef RT_STAMPS
    unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};   /* trips, busy lanes, node lanes, prim lanes, waiting lanes, trips with a node step, prim steps, cycles */
    const unsigned long long dg_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long dbg_spins = 0, dbg_claims = 0;
    unsigned long long sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};   /* cycles: refill, trip top + loads, pipeline, node step, queueing + stack, prim step, tail, spins */
    unsigned long long sec_last = dg_t0;
#define WF2_SEC(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); sec[i] += now_ - sec_last; sec_last = now_; }
#else
#define WF2_SEC(i)
#endif

    /* The staging pipeline, one stage per call: queue entries -> their rays -> the run over the plain records.  Its wait
     * is for EVERYTHING the wave has in flight (loads and stores share one counter here), so it is called after the
     * trip's two steps, when the trip's own rows have been waited for and used and nothing else is outstanding; the loads
     * it issues itself fly through the rest of this trip and the next one's steps.  (Called at the top of the trip it
     * waited, alone, for the hit record the previous trip had just stored: 17 % of the kernel's cycles on the triangle
     * scene, 31 % on the final one; called between a trip's loads and its node step it kept 44 row registers alive
     * across the run over the plain records: 23-140 spilled registers.) */
    auto pipeline = [&]() {
        if (pf_state == 2) {
            /* the rays have landed: run the plain records of the world list for the whole batch at once (one scalar
             * record stream, as in the megakernel's closest_hit) and put the result beside the ray */
            __builtin_amdgcn_s_waitcnt(0x0f70);                 /* vmcnt(0) */
            if ((uint32_t)lane < pf_n) {
                const uint32_t at = stage_base + pf_pos + (uint32_t)lane;
                const float4 o = st_o[at], d = st_d[at];
                DRay r;
                r.o = mk(o.x, o.y, o.z); r.time = o.w; r.d = mk(d.x, d.y, d.z);
                PathRng g;
                rng_start(g, k0, k1, wf_frame_pixel(W, st_slot[at]), (F & RT_F_MEDIA) ? st_smp[at] : 0u);
                HitLite b = {FLT_MAX, -1, -1, 0};
                scan_program<(F & ~RT_F_BVH)>(S, W.prog, r, tMin, b, W.bare ? nullptr : &g, __float_as_uint(d.w), stack);
                st_b[at] = wf2_word(b.t, 0, b.obj, b.aux, false);      /* a plain record's hit has no group: ordinal 0 (a miss: object -1) */
            }
            st_tail += pf_n;
            pf_state = 0;
        } else if (pf_state == 1) {
            __builtin_amdgcn_s_waitcnt(0x0f70);
            if ((uint32_t)lane < pf_n) {
                const uint32_t s_ = st_slot[stage_base + pf_pos + (uint32_t)lane];
                __builtin_amdgcn_global_load_lds(WF2_GPTR(W.ray_o + s_), WF2_LPTR(st_o + stage_base + pf_pos), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(WF2_GPTR(W.ray_d + s_), WF2_LPTR(st_d + stage_base + pf_pos), 16, 0, 0);
                if (F & RT_F_MEDIA) __builtin_amdgcn_global_load_lds(WF2_GPTR(&W.att[s_].w), WF2_LPTR(st_smp + stage_base + pf_pos), 4, 0, 0);
            }
            pf_state = 2;
        }
        if (pf_state == 0 && !cl.drained) {
#ifdef RT_STAMPS
            if (cl.chunk_next >= cl.chunk_end) ++dbg_claims;
#endif
            if (cl.chunk_next >= cl.chunk_end) wf_claim_next(cl, W, lane);
            if (!cl.drained) {
                const uint32_t pos = st_tail & (STAGE - 1u);
                uint32_t want = cl.chunk_end - cl.chunk_next;
                want = want < BATCH ? want : BATCH;
                want = want < STAGE - pos ? want : STAGE - pos;
                if (STAGE - (st_tail - st_head) >= want) {
                    pf_n = want; pf_pos = pos;
                    if ((uint32_t)lane < pf_n)
                        __builtin_amdgcn_global_load_lds(WF2_GPTR(queue + cl.chunk_next + (uint32_t)lane), WF2_LPTR(st_slot + stage_base + pf_pos), 4, 0, 0);
                    cl.chunk_next += pf_n;
                    pf_state = 1;
                }
            }
        }
    };

    while (true) {
        /* ---- idle lanes take staged rays (LDS only) ---- */
        const unsigned long long need = __ballot(mode == WF2_NEED);
        const int n_need = __popcll(need);
        const uint32_t avail = st_tail - st_head;
        if (avail > 0u && (n_need >= W.refill_min || n_need == 64 || (n_need > 0 && cl.drained && pf_state == 0))) {
            const uint32_t take = (uint32_t)n_need < avail ? (uint32_t)n_need : avail;
            if (mode == WF2_NEED) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                if (rank < take) {
                    const uint32_t at = stage_base + ((st_head + rank) & (STAGE - 1u));
                    const float4 o = st_o[at], d = st_d[at];
                    slot = st_slot[at];
                    mail[threadIdx.x] = st_b[at];
                    rl.o = mk(o.x, o.y, o.z); rl.time = o.w; rl.d = mk(d.x, d.y, d.z);
                    if (F & RT_F_XFORM) {
                        float* w_ = world_ray + threadIdx.x;
                        w_[0] = o.x; w_[RT_BLOCK] = o.y; w_[2 * RT_BLOCK] = o.z; w_[3 * RT_BLOCK] = o.w;
                        w_[4 * RT_BLOCK] = d.x; w_[5 * RT_BLOCK] = d.y; w_[6 * RT_BLOCK] = d.z;
                    }
                    ++n_rays;
                    gidx = 0;
                    WF2_NEXT_GROUP(true)
                }
            }
            st_head += take;
        }
        WF2_SEC(0)
        if (__ballot(mode != WF2_NEED) == 0ull) {
            if (cl.drained && pf_state == 0 && st_head == st_tail) break;
            pipeline();                          /* nothing to walk yet: the pipeline is still filling, or a batch finished at once */
#ifdef RT_STAMPS
            ++dbg_spins;
#endif
            WF2_SEC(7)
            continue;
        }

        /* ---- one trip ---- */
        const bool in_bvh = mode == WF2_BVH;
        const bool waiting = in_bvh && ref == RT_REF_NONE;             /* traversal done, last leaves still queued */
        const int n_node = __popcll(__ballot(in_bvh && ref >= 0)), n_wait = __popcll(__ballot(waiting));
        const uint32_t q_count = q_tail - q_head;
        const bool do_prim = q_count >= 64u || (q_count > 0u && (n_node < WF2_NODE_LOW || n_wait >= WF2_WAIT_HI));
#ifdef WF2_SPLIT_STEPS
        const bool interior = in_bvh && ref >= 0 && !do_prim;           /* a trip is a node step OR a primitive step */
#else
        const bool interior = in_bvh && ref >= 0;
#endif
        const uint32_t n_p = !do_prim ? 0u : (q_count < 64u ? q_count : 64u);
        const bool prim_lane = (uint32_t)lane < n_p;
#ifdef RT_STAMPS
        {
            dg[0] += 1; dg[1] += __popcll(__ballot(mode != WF2_NEED)); dg[2] += __popcll(__ballot(interior)); dg[3] += n_p; dg[4] += n_wait;
            dg[5] += n_node > 0 ? 1 : 0; dg[6] += do_prim ? 1 : 0;
        }
#endif
        /* the loads of both steps of the trip go out together */
        uint32_t entry = 0u;
        if (prim_lane) entry = lq[(q_head + (uint32_t)lane) & (QCAP - 1u)];
        const uint32_t item = entry & ((1u << WF2_ITEM_BITS) - 1u);
        const int owner = (int)(entry >> WF2_ITEM_BITS);
        const float4* nsrc = reinterpret_cast<const float4*>(S.nodes4 + (interior ? ref : 0));
        const float4* src = reinterpret_cast<const float4*>(S.leafs + (prim_lane ? item : 0u));
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0, q3 = q0;
        if (prim_lane) { q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3]; }
        WF2_SEC(1)

        int r0 = RT_REF_NONE, r1 = RT_REF_NONE, r2 = RT_REF_NONE, r3 = RT_REF_NONE;
        float t0 = FLT_MAX, t1 = FLT_MAX, t2 = FLT_MAX, t3 = FLT_MAX;
        if (interior) {
            const float4 n0 = nsrc[0], n1 = nsrc[1], n2 = nsrc[2], n3 = nsrc[3], n4 = nsrc[4], n5 = nsrc[5], n6 = nsrc[6];
            const float cur_t = __uint_as_float(__hip_atomic_load(my_mail + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
            /* DNode4 is axis-major (n0 = the four children's min x, n1 = max x, n2 / n3 = y, n4 / n5 = z): a child's box is a column */
            const float b0[6] = {n0.x, n2.x, n4.x, n1.x, n3.x, n5.x}, b1[6] = {n0.y, n2.y, n4.y, n1.y, n3.y, n5.y};
            const float b2[6] = {n0.z, n2.z, n4.z, n1.z, n3.z, n5.z}, b3[6] = {n0.w, n2.w, n4.w, n1.w, n3.w, n5.w};
            r0 = __float_as_int(n6.x); r1 = __float_as_int(n6.y); r2 = __float_as_int(n6.z); r3 = __float_as_int(n6.w);
            const bool sx = sl.i.x < 0.f, sy = sl.i.y < 0.f, sz = sl.i.z < 0.f;
            /* a child that is missed (or an unused slot: inverted box, reference NONE) sorts last with t = +inf */
            if (!(slab2(b0, sl, sx, sy, sz, tMin, cur_t, t0) && r0 != RT_REF_NONE)) { t0 = FLT_MAX; r0 = RT_REF_NONE; }
            if (!(slab2(b1, sl, sx, sy, sz, tMin, cur_t, t1) && r1 != RT_REF_NONE)) { t1 = FLT_MAX; r1 = RT_REF_NONE; }
            if (!(slab2(b2, sl, sx, sy, sz, tMin, cur_t, t2) && r2 != RT_REF_NONE)) { t2 = FLT_MAX; r2 = RT_REF_NONE; }
            if (!(slab2(b3, sl, sx, sy, sz, tMin, cur_t, t3) && r3 != RT_REF_NONE)) { t3 = FLT_MAX; r3 = RT_REF_NONE; }
        }
        WF2_SEC(3)
        /* The leaf children that were met go to the wave's queue straight from the node step, all of them at once: a lane's
         * number of items (0 .. 8: up to four leaves of one or two items, which is what the host's builder makes) is summed
         * over the lower lanes bit by bit (ballot + mbcnt), which gives every lane its place behind the queue's tail.  In
         * the tree they then count as missed, and only interior children are sorted and stacked.  Leaves of more than two
         * items (a caller's own tree), and all leaves when the ring has no room for a burst, are stacked like interior
         * children instead; the round further down queues them one item per trip when they come off the stack. */
#define WF2_SMALL_LEAF(r_) ((r_) < 0 && (r_) != RT_REF_NONE && ((uint32_t)(~(r_)) & 14u) == 0u)
#define WF2_ITEMS(r_) (WF2_SMALL_LEAF(r_) ? ((uint32_t)(~(r_)) & 1u) + 1u : 0u)
        const uint32_t cnt = WF2_ITEMS(r0) + WF2_ITEMS(r1) + WF2_ITEMS(r2) + WF2_ITEMS(r3);
        uint32_t before = 0u, total = 0u;
#pragma unroll
        for (int bit = 0; bit < 4; ++bit) {
            const unsigned long long m = __ballot((cnt >> bit) & 1u);
            before += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) << bit;
            total += (uint32_t)__popcll(m) << bit;
        }
        const bool room = q_tail - q_head + total <= QCAP;           /* wave-uniform */
        if (interior) {
            if (room && cnt > 0u) {
                uint32_t pos = q_tail + before;
                const uint32_t tag = (uint32_t)lane << WF2_ITEM_BITS;
#define WF2_QUEUE_LEAF(r_, t_)                                                                               \
                if (WF2_SMALL_LEAF(r_)) {                                                                        \
                    const uint32_t first_ = (uint32_t)(~r_) >> 4, two_ = (uint32_t)(~r_) & 1u;                   \
                    lq[pos & (QCAP - 1u)] = tag | first_;                                                        \
                    if (two_) lq[(pos + 1u) & (QCAP - 1u)] = tag | (first_ + 1u);                                \
                    pos += 1u + two_;                                                                            \
                    r_ = RT_REF_NONE; t_ = FLT_MAX;                                                              \
                }
                WF2_QUEUE_LEAF(r0, t0) WF2_QUEUE_LEAF(r1, t1) WF2_QUEUE_LEAF(r2, t2) WF2_QUEUE_LEAF(r3, t3)
#undef WF2_QUEUE_LEAF
                mark = q_tail + total;
            }
#define WF_CSWAP(ta, ra, tb, rb) { const bool sw_ = tb < ta; const float tt_ = sw_ ? tb : ta; const int rr_ = sw_ ? rb : ra; tb = sw_ ? ta : tb; rb = sw_ ? ra : rb; ta = tt_; ra = rr_; }
            WF_CSWAP(t0, r0, t1, r1) WF_CSWAP(t2, r2, t3, r3) WF_CSWAP(t0, r0, t2, r2) WF_CSWAP(t1, r1, t3, r3) WF_CSWAP(t1, r1, t2, r2)
#undef WF_CSWAP
            /* farthest first onto the stack, nearest entered next.  While the three entries fit the LDS part of the stack
             * they are written without a branch: each at the place the valid ones before it leave, in this order, so
             * that a missing one is overwritten by the next (the LDS takes a wave's writes in order) */
            if (sp + 3 <= LDS_STACK) {
                const int p2 = sp + (r3 != RT_REF_NONE ? 1 : 0), p1 = p2 + (r2 != RT_REF_NONE ? 1 : 0);
                stack[sp * RT_BLOCK] = r3;
                stack[p2 * RT_BLOCK] = r2;
                stack[p1 * RT_BLOCK] = r1;
                sp = p1 + (r1 != RT_REF_NONE ? 1 : 0);
            } else {
                if (r3 != RT_REF_NONE) WF2_PUSH(r3)
                if (r2 != RT_REF_NONE) WF2_PUSH(r2)
                if (r1 != RT_REF_NONE) WF2_PUSH(r1)
            }
            ref = r0;
            if (ref == RT_REF_NONE) WF2_POP(ref)
        }
        if (room) q_tail += total;
#undef WF2_ITEMS
#undef WF2_SMALL_LEAF
        WF2_SEC(4)
        if (do_prim) {          /* wave-uniform: every lane takes part in the exchange of the rays, the first n_p test */
            const int oa = owner << 2;
            DRay r;
            r.o.x = __int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(rl.o.x)));
            r.o.y = __int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(rl.o.y)));
            r.o.z = __int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(rl.o.z)));
            r.d.x = __int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(rl.d.x)));
            r.d.y = __int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(rl.d.y)));
            r.d.z = __int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(rl.d.z)));
            r.time = (F & RT_F_MSPHERE) ? __int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(rl.time))) : 0.f;
            const int ordinal = W.n_groups > 1 ? __builtin_amdgcn_ds_bpermute(oa, gidx) + 1 : 1;
            if (prim_lane) {
                unsigned long long* box = &mail[wave_base + (uint32_t)owner];
                /* the bound is the mailbox's t plus one ulp: a primitive AT the recorded t still passes its own test, and the
                 * atomic min below then decides between the two by the rest of the word, whichever was tested first (tested
                 * against the bare bound, the later one of a tie would lose, and which one is later depends on timing) */
                const float tmax = __uint_as_float(__hip_atomic_load(reinterpret_cast<uint32_t*>(box) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) + 1u);
                const uint32_t head = __float_as_uint(q0.x);
                const int oi = __float_as_int(q0.z);
                const float f[13] = {q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
                float t; int aux;
                bool h;
                if (!(F & RT_F_XFORM) || obj_op_count(head) == 0) h = prim_t<F>(obj_type(head), f, r, tMin, tmax, t, aux);
                else h = leaf_t<F>(S, S.objs[oi], r, tMin, tmax, t, aux);
                h = h && q0.y == q0.y;           /* keeps the row's four loads whole and aligned (see wf_intersect_kernel) */
                const int ty = obj_type(head);
                if (h) (void)__hip_atomic_fetch_min(box, wf2_word(t, ordinal, oi, aux, ty >= RT_OBJ_RECT_XY && ty <= RT_OBJ_BOX), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            q_head += n_p;
        }
        WF2_SEC(5)
        pipeline();
        WF2_SEC(2)
        /* ---- a lane that still holds a leaf in `ref` (the root of a tiny tree, a leaf of more than two items, or a burst
         * the ring had no room for): one item per trip ---- */
        {
            const bool at_leaf = in_bvh && ref < 0 && ref != RT_REF_NONE;
            const unsigned long long m = __ballot(at_leaf);
            const uint32_t n_m = (uint32_t)__popcll(m);
            if (m != 0ull && q_tail - q_head + n_m <= QCAP) {
                if (at_leaf) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    const uint32_t first = (uint32_t)(~ref) >> 4, count = ((uint32_t)(~ref) & 15u) + 1u;
                    lq[(q_tail + rank) & (QCAP - 1u)] = ((uint32_t)lane << WF2_ITEM_BITS) | (first + (uint32_t)li);
                    mark = q_tail + n_m;
                    if ((uint32_t)++li == count) { li = 0; WF2_POP(ref) }
                }
                q_tail += n_m;
            }
        }
        /* ---- a ray that has walked its group and whose last leaves have been tested moves on ---- */
        if (in_bvh && ref == RT_REF_NONE && (int)(mark - q_head) <= 0) {
            ++gidx;
            WF2_NEXT_GROUP(false)
        }
        WF2_SEC(6)
    }
#undef WF2_NEXT_GROUP
#undef WF2_PUSH
#undef WF2_POP
    unsigned long long r = n_rays;
    for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off);
    if (lane == 0) atomicAdd(&counters[0], r);
#ifdef RT_STAMPS
    dg[7] = __builtin_amdgcn_s_memtime() - dg_t0;
    if (lane == 0) for (int q = 0; q < 8; ++q) atomicAdd(&counters[4 + (q < 6 ? q : q + 4)], dg[q]);
    if (lane == 0) { for (int q = 0; q < 8; ++q) atomicAdd(&counters[16 + q], sec[q]); atomicAdd(&counters[24], dbg_spins); atomicAdd(&counters[25], dbg_claims); }
#endif
#undef WF2_SEC
}

#endif
