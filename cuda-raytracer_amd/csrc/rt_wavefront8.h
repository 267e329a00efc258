/*
 * rt_wavefront8.h — the intersect kernel of the wavefront form over 8-wide nodes with quantised child boxes (round 3).
 *
 * Everything around the tree is rt_wavefront2.h's: rays are staged in LDS and run over the plain records of the world list a
 * batch at a time, leaves are not tested by the lane that meets them but queued per wave and tested 64 at a time, the
 * closest hit of a ray is a 64-bit mailbox word in LDS.  What changes is the node (rt_host.h DNode8, built by
 * rt_pack_bvh8): eight children behind 8-bit boxes in 80 bytes.  Counted on the 1M-triangle scene, the four-wide kernel
 * spends per ray 19 node rows of 7 sixteen-byte lane loads each (a CU's address unit takes about one such load per clock)
 * and 19 rounds of everything a trip costs beside the box tests; the idea was fewer, smaller rows:
 *  - one row = five lane loads, three of them the 48 bytes of boxes;
 *  - a child's planes are q * (2^e / d) + (p - o) / d: two constants per axis per node, then one byte-to-float conversion and
 *    one fused multiply-add per plane;
 *  - no sorting: the host put the child that lies towards octant s of the node into slot s, and a ray whose direction signs
 *    form the octant o takes the slots it hit by ascending s ^ o (Ylitie, Karras, Laine 2017).  The stack holds one entry per
 *    NODE with children still to visit: child base | interior mask | hit mask in that order, 64 bits;
 *  - the items of the leaf children that were hit are a 16-bit mask over the node's own rows of the leaf table (row 16 node +
 *    slot, + 8 for a second item), queued in one go.
 * MEASURED (profiles/r03_kernel_ab.txt, tools/stamps.py): it does not pay on that scene.  Per ray 19.2 eight-wide rows (the
 * four-wide tree: 18.9) and 17.5 leaf rows (7.9): octant order visits children the sorted four-wide walk culls, and the
 * mesh's boxes overlap whatever the width (tools/sim/bvh_sim.cpp).  1M triangles 2.40 Grays/s against 2.87 for the
 * four-wide kernel in the same two-kernel form and 3.66 for the one-kernel form.  Opt-in: RT_WF_KERNEL=3.
 * Box culling never decides a result: the boxes are rounded outwards on the host (rt_flat_bvh8_check walks every item up
 * through all its ancestors' dequantised boxes), the rounding of the two constants is covered by the per-ray widening of
 * slab_axis8, and every primitive behind a box is tested with the reference's own arithmetic.
 */
#ifndef RT_WAVEFRONT8_H
#define RT_WAVEFRONT8_H
#include "rt_wavefront2.h"

#define WF8_LDS_STACK 8          /* 64-bit stack entries per lane kept in LDS (one per tree level with siblings left) */
#define WF8_OVF_STACK 8          /* ... and in scratch; the host keeps deeper trees on the four-wide kernels */
constexpr uint32_t wf8_qcap(uint32_t F) { return (F & RT_F_XFORM) ? 128u : 256u; }      /* leaf-queue entries per wave */

/* slab_axis (rt_wavefront2.h) for planes given as p + q 2^e: beside o / d the product p / d is rounded as well, so the
 * widening also counts the largest coordinate any box has (box_abs, from the host) */
__device__ __forceinline__ void slab_axis8(float o, float d, float dmax, float box_abs, float& i, float& lo, float& hi)
{
    const float floor_ = dmax * 9.094947e-13f;
    const float dd = fabsf(d) < floor_ ? __builtin_copysignf(floor_, d) : d;
    i = __builtin_amdgcn_rcpf(dd);
    const float oi = o * i;
    const float e = (fabsf(oi) + box_abs * fabsf(i)) * 4.7683716e-7f;
    const bool ok = fabsf(oi) < 1e37f && fabsf(i) < 1e30f;
    const bool dead = !(fabsf(o) <= FLT_MAX) || !(fabsf(d) <= FLT_MAX) || !(dmax > 0.f);
    lo = ok ? -oi - e : (dead ? INFINITY : -INFINITY);
    hi = ok ? -oi + e : (dead ? -INFINITY : INFINITY);
    i = ok ? i : 0.f;
}
__device__ __forceinline__ void slab_setup8(const DRay& r, float box_abs, Slab3& s)
{
    const float dmax = fmaxf(fmaxf(fabsf(r.d.x), fabsf(r.d.y)), fabsf(r.d.z));
    slab_axis8(r.o.x, r.d.x, dmax, box_abs, s.i.x, s.lo.x, s.hi.x);
    slab_axis8(r.o.y, r.d.y, dmax, box_abs, s.i.y, s.lo.y, s.hi.y);
    slab_axis8(r.o.z, r.d.z, dmax, box_abs, s.i.z, s.lo.z, s.hi.z);
}
__device__ __forceinline__ float wf8_byte(uint32_t w, int b) { return (float)((w >> (8 * b)) & 0xffu); }

template <uint32_t F>
__global__ __launch_bounds__(RT_BLOCK, wf2_waves(F)) void wf_intersect8_kernel(DScene S, WfState W, int qi, uint32_t k0, uint32_t k1,
                                                                 unsigned long long* __restrict__ counters)
{
    extern __shared__ uint2 stack8_mem[];                               /* n_stack x RT_BLOCK entries */
    constexpr uint32_t STAGE = wf2_stage(F), BATCH = STAGE / 2u, QCAP = wf8_qcap(F), NW = RT_BLOCK / 64;
    __shared__ float4 st_o[NW * STAGE], st_d[NW * STAGE];               /* staged rays: origin | time, direction | bounce */
    __shared__ unsigned long long st_b[NW * STAGE];                     /* ... and their closest hit over the plain records, as a mailbox word */
    __shared__ uint32_t st_slot[NW * STAGE];
    __shared__ uint32_t st_smp[(F & RT_F_MEDIA) ? NW * STAGE : 1];
    __shared__ unsigned long long mail[RT_BLOCK];                       /* per lane: the closest hit so far of the lane's ray */
    __shared__ uint32_t lq_mem[NW * QCAP];
    __shared__ Wf2Group groups[WF2_MAX_GROUPS];
    __shared__ float world_ray[(F & RT_F_XFORM) ? 7 * RT_BLOCK : 1];    /* o.xyz, time, d.xyz of every lane's ray, component-major */
    uint2* stack = stack8_mem + threadIdx.x;
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
        g.obj = obj; g.root = o.pad[1]; g.head = o.head; g.op_begin = o.op_begin;       /* pad[1]: the group's 8-wide root */
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
    uint32_t oct = 0u;                             /* bit k set: the ray runs towards negative k */
    int gidx = 0, node = -1, sp = 0;
    uint32_t cur_base = 0u, cur_bits = 0u;         /* the node whose children are being visited: child base; interior mask | hit mask by priority << 8 */
    uint32_t pend_bits = 0u, pend_row = 0u;        /* leaf items of the node just visited that are not queued yet */
    uint32_t mark = 0u;                            /* the queue position behind this ray's last entry */
    uint2 ovf[WF8_OVF_STACK];
    /* enter the gidx-th BVH group of the world list, or, past the last one, hand in the hit (all from LDS) */
#define WF8_NEXT_GROUP(FIRST)                                                                                    \
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
            slab_setup8(rl, W.box_abs, sl);                                                                      \
            oct = (sl.i.x < 0.f ? 1u : 0u) | (sl.i.y < 0.f ? 2u : 0u) | (sl.i.z < 0.f ? 4u : 0u);                \
            mode = WF2_BVH; node = g_.root; sp = 0; cur_bits = 0u; pend_bits = 0u; mark = q_head;                \
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

    /* the staging pipeline of rt_wavefront2.h, one stage per call, after the trip's steps */
    auto pipeline = [&]() {
        if (pf_state == 2) {
            __builtin_amdgcn_s_waitcnt(0x0f70);                 /* vmcnt(0) */
            if ((uint32_t)lane < pf_n) {
                const uint32_t at = stage_base + pf_pos + (uint32_t)lane;
                const float4 o = st_o[at], d = st_d[at];
                DRay r;
                r.o = mk(o.x, o.y, o.z); r.time = o.w; r.d = mk(d.x, d.y, d.z);
                PathRng g;
                rng_start(g, k0, k1, wf_frame_pixel(W, st_slot[at]), (F & RT_F_MEDIA) ? st_smp[at] : 0u);
                HitLite b = {FLT_MAX, -1, -1, 0};
                scan_program<(F & ~RT_F_BVH)>(S, W.prog, r, tMin, b, W.bare ? nullptr : &g, __float_as_uint(d.w), nullptr);
                st_b[at] = wf2_word(b.t, 0, b.obj, b.aux, false);
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
                    WF8_NEXT_GROUP(true)
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
        const bool walking = in_bvh && node >= 0 && pend_bits == 0u;
        const bool waiting = in_bvh && node < 0;                        /* traversal done, last leaves still queued */
        const int n_node = __popcll(__ballot(walking)), n_wait = __popcll(__ballot(waiting));
        const uint32_t q_count = q_tail - q_head;
        const bool do_prim = q_count >= 64u || (q_count > 0u && (n_node < WF2_NODE_LOW || n_wait >= WF2_WAIT_HI));
        const uint32_t n_p = !do_prim ? 0u : (q_count < 64u ? q_count : 64u);
        const bool prim_lane = (uint32_t)lane < n_p;
#ifdef RT_STAMPS
        {
            dg[0] += 1; dg[1] += __popcll(__ballot(mode != WF2_NEED)); dg[2] += n_node; dg[3] += n_p; dg[4] += n_wait;
            dg[5] += n_node > 0 ? 1 : 0; dg[6] += do_prim ? 1 : 0;
        }
#endif
        /* the loads of both steps of the trip go out together */
        uint32_t entry = 0u;
        if (prim_lane) entry = lq[(q_head + (uint32_t)lane) & (QCAP - 1u)];
        const uint32_t item = entry & ((1u << WF2_ITEM_BITS) - 1u);
        const int owner = (int)(entry >> WF2_ITEM_BITS);
        const uint4* nsrc = reinterpret_cast<const uint4*>(S.nodes8 + (walking ? node : 0));
        const float4* src = reinterpret_cast<const float4*>(S.leafs8 + (prim_lane ? item : 0u));
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0, q3 = q0;
        if (prim_lane) { q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3]; }
        WF2_SEC(1)

        if (walking) {
            const uint4 h0 = nsrc[0], h1 = nsrc[1], a0 = nsrc[2], a1 = nsrc[3], a2 = nsrc[4];
            const float cur_t = __uint_as_float(__hip_atomic_load(my_mail + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
            /* per axis: A = 2^e / d, C = (p - o) / d widened down (near planes) and up (far planes) */
            const float ax = __uint_as_float((h0.w & 0xffu) << 23) * sl.i.x, ay = __uint_as_float(((h0.w >> 8) & 0xffu) << 23) * sl.i.y,
                        az = __uint_as_float(((h0.w >> 16) & 0xffu) << 23) * sl.i.z;
            const float px = __uint_as_float(h0.x), py = __uint_as_float(h0.y), pz = __uint_as_float(h0.z);
            const float cnx = __builtin_fmaf(px, sl.i.x, sl.lo.x), cny = __builtin_fmaf(py, sl.i.y, sl.lo.y), cnz = __builtin_fmaf(pz, sl.i.z, sl.lo.z);
            const float cfx = __builtin_fmaf(px, sl.i.x, sl.hi.x), cfy = __builtin_fmaf(py, sl.i.y, sl.hi.y), cfz = __builtin_fmaf(pz, sl.i.z, sl.hi.z);
            const bool sx = (oct & 1u) != 0u, sy = (oct & 2u) != 0u, sz = (oct & 4u) != 0u;
            /* the near and far planes' bytes of the eight slots, two words per axis (qlo x: a0.xy, y: a0.zw, z: a1.xy; qhi x: a1.zw, y: a2.xy, z: a2.zw) */
            const uint32_t nx[2] = {sx ? a1.z : a0.x, sx ? a1.w : a0.y}, fx[2] = {sx ? a0.x : a1.z, sx ? a0.y : a1.w};
            const uint32_t ny[2] = {sy ? a2.x : a0.z, sy ? a2.y : a0.w}, fy[2] = {sy ? a0.z : a2.x, sy ? a0.w : a2.y};
            const uint32_t nz[2] = {sz ? a2.z : a1.x, sz ? a2.w : a1.y}, fz[2] = {sz ? a1.x : a2.z, sz ? a1.y : a2.w};
            uint32_t hits = 0u;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int w = s >> 2, b = s & 3;
                const float tn = fmaxf(fmaxf(__builtin_fmaf(wf8_byte(nx[w], b), ax, cnx), __builtin_fmaf(wf8_byte(ny[w], b), ay, cny)),
                                       fmaxf(__builtin_fmaf(wf8_byte(nz[w], b), az, cnz), tMin));
                const float tf = fminf(fminf(__builtin_fmaf(wf8_byte(fx[w], b), ax, cfx), __builtin_fmaf(wf8_byte(fy[w], b), ay, cfy)),
                                       fminf(__builtin_fmaf(wf8_byte(fz[w], b), az, cfz), cur_t));
                hits |= tn <= tf * 1.0000012f ? 1u << s : 0u;
            }
            const uint32_t imask = h0.w >> 24, lmask = h1.y & 0xffu, l2mask = (h1.y >> 8) & 0xffu;
            const uint32_t hit_leaf = hits & lmask;
            uint32_t m = hits & imask;
            /* the interior hits by priority: bit s moves to bit s ^ oct (three conditional swaps) */
            if (oct & 1u) m = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u);
            if (oct & 2u) m = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u);
            if (oct & 4u) m = ((m & 0x0fu) << 4) | ((m >> 4) & 0x0fu);
            pend_bits = hit_leaf | ((hit_leaf & l2mask) << 8);
            pend_row = 16u * (uint32_t)node;
            /* the node's own entry replaces the current one; what the current one still holds goes onto the stack */
            if (m != 0u) {
                if ((cur_bits >> 8) != 0u) {
                    if (sp < WF8_LDS_STACK) stack[sp * RT_BLOCK] = make_uint2(cur_base, cur_bits); else ovf[sp - WF8_LDS_STACK] = make_uint2(cur_base, cur_bits);
                    ++sp;
                }
                cur_base = h1.x; cur_bits = imask | (m << 8);
            }
            if ((cur_bits >> 8) == 0u && sp > 0) {
                --sp;
                const uint2 e = sp < WF8_LDS_STACK ? stack[sp * RT_BLOCK] : ovf[sp - WF8_LDS_STACK];
                cur_base = e.x; cur_bits = e.y;
            }
            /* the next node: the waiting child of the current entry with the lowest s ^ oct */
            const uint32_t left = cur_bits >> 8;
            if (left != 0u) {
                const uint32_t prio = (uint32_t)__builtin_ctz(left), s = prio ^ oct;
                node = (int)(cur_base + (uint32_t)__popc(cur_bits & 0xffu & ((1u << s) - 1u)));
                cur_bits = (cur_bits & 0xffu) | ((left & (left - 1u)) << 8);
            } else node = -1;
        }
        WF2_SEC(3)
        /* ---- the leaf items that were met go to the wave's queue: every lane's place behind the tail is the sum of the item
         * counts of the lanes below it (0 .. 16, summed bit by bit with ballot + mbcnt).  Lanes whose items do not fit the
         * ring keep them and do not walk until they are queued (a prefix of the lanes always fits: bursts only). ---- */
        {
            const uint32_t cnt = in_bvh ? (uint32_t)__popc(pend_bits) : 0u;
            uint32_t before = 0u, total = 0u;
#pragma unroll
            for (int bit = 0; bit < 5; ++bit) {
                const unsigned long long bm = __ballot((cnt >> bit) & 1u);
                before += __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u)) << bit;
                total += (uint32_t)__popcll(bm) << bit;
            }
            if (total != 0u) {                                   /* wave-uniform */
                const uint32_t room = QCAP - (q_tail - q_head);
                const bool fits = cnt != 0u && before + cnt <= room;
                uint32_t added = total;
                if (total > room) {
                    const unsigned long long fm = __ballot(fits);
                    added = fm ? (uint32_t)__builtin_amdgcn_readlane((int)(before + cnt), 63 - __builtin_clzll(fm)) : 0u;
                }
                if (fits) {
                    uint32_t pos = q_tail + before, bits = pend_bits;
                    const uint32_t tag = (uint32_t)lane << WF2_ITEM_BITS;
                    while (bits != 0u) {
                        lq[pos & (QCAP - 1u)] = tag | (pend_row + (uint32_t)__builtin_ctz(bits));
                        bits &= bits - 1u;
                        ++pos;
                    }
                    pend_bits = 0u;
                    mark = q_tail + added;
                }
                q_tail += added;
            }
        }
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
                /* the bound is the mailbox's t plus one ulp (see wf_intersect2_kernel) */
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
        /* ---- a ray that has walked its group and whose last leaves have been tested moves on ---- */
        if (in_bvh && node < 0 && pend_bits == 0u && (int)(mark - q_head) <= 0) {
            ++gidx;
            WF8_NEXT_GROUP(false)
        }
        WF2_SEC(6)
    }
#undef WF8_NEXT_GROUP
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
