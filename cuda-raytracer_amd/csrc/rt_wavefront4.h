/*
 * rt_wavefront4.h — the whole path in ONE persistent kernel (RT_WF_KERNEL=4): the step interpreter of rt_wavefront.h with
 * the bounce logic of wf_logic_kernel run inside its refill.
 *
 * What the two-kernel form pays for beyond its traversal (profiles/r03*_final_pmc.txt, r02d_tri_pmc.txt): the logic
 * kernel is a quarter of the call (26 % final scene, 29 % triangles) and spends three quarters of its wave cycles waiting
 * for the state rows it reads and writes back; every iteration of every pool ends in a drain, in which the persistent
 * intersect grid waits for its last rays with most lanes idle; and the pools overlap the two kernels only where wave
 * slots happen to be free (the serialised kernel times add up to the overlapped call's).  Here a lane keeps its path in
 * registers and LDS from the camera to the end: when its ray's closest hit is known the lane is idle, and once enough
 * lanes of the wave are idle they run one bounce TOGETHER (rebuild, cooperative ball draw, scatter: wf_logic_one's body on
 * register state), paths that ended start the slot's next sample or take a fresh slot from the work list, and the batch
 * goes through the plain records of the world list as a refill of wf_intersect_kernel does and on into the first BVH group.
 * No path state in HBM, no queues, no logic / scan / move launches, ONE drain per call.
 *
 * The work list is the slot index space itself (slot = stream * n_pixels + window pixel, claimed in chunks with the same
 * XCD-affine heads); a slot runs its samples s_begin + k, s_begin + k + K, ... one after the other and leaves their sum in
 * W.sum[slot], and wf_finalize_kernel adds a pixel's K sums in stream order: the same sums in the same order as the
 * two-kernel form, so the image has the same bits (ties between equal-t primitives aside, as between any two traversals).
 */
#ifndef RT_WAVEFRONT4_H
#define RT_WAVEFRONT4_H
#include "rt_wavefront.h"

/* waves per SIMD: measured when the kernel was first built (1080p; 1M triangles x 256 spp / final scene x 64 spp, Grays/s)
 * 3 waves 3.20 / 3.69, 4 waves 3.19 / 4.20, 5 waves 3.51 / 3.92 -- the triangle variant lives with 96 registers (13 spilled),
 * the general ones do not.  Again at the end of round 3 (tools/lib_ab.sh): general variants at 3 / 4 / 5 waves, final scene
 * 4.08 / 4.57 / 4.42, Cornell + sphere cluster 5.75 / 6.52 / 4.51; triangle variant at 4 waves with the inner trip loop 3.58
 * against 3.70 as it is */
#ifndef WF4_GENERAL_WAVES
#define WF4_GENERAL_WAVES 4
#endif
#ifndef WF4_TRI_WAVES
#define WF4_TRI_WAVES 5
#endif
constexpr int wf4_waves(uint32_t F) { return (F & ~(RT_F_BVH | RT_F_TRI | RT_F_PLANAR)) == 0 ? WF4_TRI_WAVES : WF4_GENERAL_WAVES; }
#ifndef WF4_INNER_LOOP_TRI
#define WF4_INNER_LOOP_TRI false
#endif
constexpr bool wf4_inner_loop(uint32_t F) { return (F & ~(RT_F_BVH | RT_F_TRI | RT_F_PLANAR)) == 0 ? WF4_INNER_LOOP_TRI : true; }
#ifndef WF4_ENTER_BATCH
#define WF4_ENTER_BATCH 16      /* lanes waiting to enter their next BVH group that make the entry worth running */
#endif
#define WF_ENTER 1
#ifndef WF4_CHUNKS_PER_WAVE
#define WF4_CHUNKS_PER_WAVE 256
#endif
#define WF4_LDS_EXTRA_WORDS (7 * RT_BLOCK)      /* per lane: attenuation xyz, sum xyz; per wave: 64 ints for the ball draw */

/* MG: the world list holds more than one BVH group (the final scene: the box floor and the sphere cluster).  With a single
 * group (the triangle mesh) a ray enters it in the refill and is finished when it leaves it: the trip loop then carries
 * no entry block at all. */
template <uint32_t F, int BG, bool MG>
__global__ __launch_bounds__(RT_BLOCK, wf4_waves(F)) void wf_path_kernel(DScene S, WfState W, int Wd, int Hd, uint32_t s_begin, uint32_t s_end,
                                                                      uint32_t k0, uint32_t k1, int max_depth,
                                                                      unsigned long long* __restrict__ counters)
{
    extern __shared__ int stack_mem[];
    float* pst = reinterpret_cast<float*>(stack_mem) + threadIdx.x;                    /* pst[k * RT_BLOCK]: att 0..2, sum 3..5 */
    int* ball_slots = stack_mem + 6 * RT_BLOCK + (threadIdx.x & ~63u);
    int* stack = stack_mem + WF4_LDS_EXTRA_WORDS + threadIdx.x;
    const uint32_t n_queue = W.n_slots;
    const int lane = (int)(threadIdx.x & 63u);
    const float tMin = .001f;

    int mode = WF_NEED;
    bool live = false;                          /* the lane owns a slot with a path in flight */
    uint32_t slot = 0;
    DRay rw = {mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 1.f), 0.f}, rl_own = rw;
    DRay& rl = (F & RT_F_XFORM) ? rl_own : rw;
    V3 id = mk(0.f, 0.f, 0.f);
    uint32_t soff = 0u;                         /* wf4_sign_offsets(id): where in a node row this ray's NEAR planes are */
    HitLite best = {FLT_MAX, -1, -1, 0};
    int gidx = 0, gi = -1, ref = RT_REF_NONE, pleaf = RT_REF_NONE, sp = 0, li = 0;
    int ovf[WF_OVF_STACK];
#define WF_PUSH(v) { const int v_ = (v); if (sp < WF_LDS_STACK) stack[sp * RT_BLOCK] = v_; else ovf[sp - WF_LDS_STACK] = v_; ++sp; }
#define WF_POP(dst) { dst = RT_REF_NONE; if (sp > 0) { --sp; if (sp < WF_LDS_STACK) dst = stack[sp * RT_BLOCK]; else dst = ovf[sp - WF_LDS_STACK]; } }
    /* as in wf_intersect_kernel, except that a finished ray stays in the lane: mode WF_NEED with `live` set */
#define WF_NEXT_GROUP                                                                                            \
    {                                                                                                            \
        if (gidx < W.n_groups) {                                                                                 \
            const int obj_ = W.groups[gidx];                                                                     \
            const DObj& o_ = S.objs[obj_];                                                                       \
            if (F & RT_F_XFORM) { rl = rw; apply_chain(S, o_, rl); }                                             \
            id = mk(1.f / rl.d.x, 1.f / rl.d.y, 1.f / rl.d.z);                                                   \
            soff = wf4_sign_offsets(id);                                                                         \
            mode = WF_BVH; gi = obj_ == S.world ? -1 : obj_; ref = o_.pad[0]; pleaf = RT_REF_NONE; sp = 0; li = 0; \
        } else mode = WF_NEED;                                                                                   \
    }
    PathRng g;
    rng_start(g, k0, k1, 0u, 0u);
    uint32_t bounce = 0;
    uint32_t n_rays = 0;
    bool drained = false;
    uint32_t chunk_next = 0, chunk_end = 0;
    /* a slot is many rays here (all its samples), and a wave owns its chunk to the end: chunks are kept small enough that
     * every wave takes a few hundred of them, or the waves with the heaviest last chunks decide when the call ends
     * (1M triangles x 256 spp with 32 M slots: 2.93 Grays/s in chunks of 1024 against 3.19 with 128 M slots) */
    uint32_t chunk = n_queue / (gridDim.x * (RT_BLOCK / 64) * (uint32_t)WF4_CHUNKS_PER_WAVE);
    chunk = chunk < 64u ? 64u : (chunk > (uint32_t)WF_CHUNK ? (uint32_t)WF_CHUNK : chunk);
    const uint32_t af_chunks = (n_queue + chunk - 1u) / chunk;
    const uint32_t af_streams = W.n_slots / (W.n_pixels ? W.n_pixels : 1u);
    uint32_t af_cps = (af_chunks + (af_streams ? af_streams : 1u) - 1u) / (af_streams ? af_streams : 1u);
    af_cps = af_cps < 8u ? 8u : af_cps;
    int band = (int)(blockIdx.x & 7u);
#ifdef RT_STAMPS
    unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long dg_t0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef RT_STAMPS_RF3
    unsigned long long rf3[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      /* cycles in a refill's bounce / claims / plain records, lanes shaded, rays started; lanes and batches per material kind */
#endif

    while (true) {
        /* ---- refill: idle lanes run their bounce, ended paths take the next sample or a new slot, the batch enters the scene ---- */
        const bool idle = mode == WF_NEED;
        const unsigned long long shade_m = __ballot(idle && live);
        const int n_work = __popcll(__ballot(idle && (live || !drained)));
        const int n_busy = 64 - __popcll(__ballot(idle));
        /* enough lanes to make the batch worth its instructions; once the work list is empty the batches shrink with the
         * number of paths the wave still holds */
        const int n_live = __popcll(__ballot(live));
        const int thr = drained ? (n_live + 3) >> 2 : W.refill_min;
#ifdef RT_STAMPS_REFILL
        const bool rf_on = n_work > 0 && (n_work >= thr || n_busy == 0);
        const unsigned long long rf_t0 = __builtin_amdgcn_s_memtime();
#endif
        if (n_work > 0 && (n_work >= thr || n_busy == 0)) {
            bool want_ray = false;
            /* a lane that starts a sample: the next one of its slot (a path ended in the bounce) or the first one of a slot
             * it claims.  ONE camera-ray block serves both: it is a Philox block, the lens draw and the ray, a few hundred
             * instructions that a handful of the batch's lanes run, and two copies of it ran in most refills */
            bool starting = false;
            uint32_t start_s = 0u;
            if (shade_m) {
                /* one iteration of colour() / colourUnlit(), main.cu:53-70 / :77-98 — wf_logic_one on register state */
                const bool active = idle && live;
                const bool hit = active && best.obj >= 0;
                HitRec rec;
                rec.t = 0.f; rec.u = 0.f; rec.v = 0.f; rec.p = mk(0.f, 0.f, 0.f); rec.n = mk(0.f, 0.f, 0.f);
                rec.material = 0; rec.prim_id = -1; rec.front = 0;
                int mtype = RT_MAT_LIGHT;
                if (hit) {
                    rebuild<F, false>(S, rw, best, .001f, rec);
                    mtype = S.mats[rec.material].type;
                }
#ifdef RT_STAMPS_RF3
                for (int mt_ = 0; mt_ < 5; ++mt_) {      /* lanes of each material kind in this batch, batches that hold the kind */
                    const int nm_ = __popcll(__ballot(hit && mtype == mt_));
                    rf3[5 + mt_] += (unsigned long long)nm_; rf3[10 + mt_] += nm_ ? 1ull : 0ull;
                }
#endif
                const bool need_ball = hit && mtype != RT_MAT_LIGHT && mtype != RT_MAT_GLASS;
                const V3 ball = random_in_unit_sphere_wave(g, bounce, need_ball, ball_slots);
                if (active) {
                    V3 att = mk(pst[0], pst[RT_BLOCK], pst[2 * RT_BLOCK]);
                    V3 result = mk(0.f, 0.f, 0.f);
                    bool done;
                    if (hit) {
                        V3 attenuation, emitted;
                        DRay scattered;
                        if (scatter<F>(S, rw, rec, g, bounce, ball, attenuation, scattered, emitted)) {
                            att = att * attenuation;
                            rw = scattered;
                            done = false;
                        } else {
                            result = BG == RT_BG_BLACK ? emitted * att : mk(0.f, 0.f, 0.f);
                            done = true;
                        }
                    } else {
                        if (BG != RT_BG_BLACK) {
                            V3 ud = unit(rw.d);
                            float t = .5f * (ud.y + 1.f);
                            V3 c = (1.f - t) * mk(1.f, 1.f, 1.f) + t * mk(.5f, .7f, 1.f);
                            result = att * c;
                        }
                        done = true;
                    }
                    ++bounce;
                    if (!done && (int)bounce >= max_depth) { result = mk(0.f, 0.f, 0.f); done = true; }
                    if (!done) {
                        pst[0] = att.x; pst[RT_BLOCK] = att.y; pst[2 * RT_BLOCK] = att.z;
                        want_ray = true;
                    } else {
                        const float sx = pst[3 * RT_BLOCK] + result.x, sy = pst[4 * RT_BLOCK] + result.y, sz = pst[5 * RT_BLOCK] + result.z;
                        const uint32_t next = g.sample + W.streams;
                        if (next < s_end && next > g.sample) {
                            pst[3 * RT_BLOCK] = sx; pst[4 * RT_BLOCK] = sy; pst[5 * RT_BLOCK] = sz;
                            start_s = next; starting = true;      /* the slot's next sample: its camera ray is made below */
                        } else {
                            W.sum[slot] = make_float4(sx, sy, sz, 0.f);
                            live = false;
                        }
                    }
                }
            }
#ifdef RT_STAMPS_RF3
            const unsigned long long rf_t1 = __builtin_amdgcn_s_memtime();
#endif
            /* lanes without a path take the next slots of the wave's chunk of the work list (claims as in wf_intersect_kernel) */
            const unsigned long long want = __ballot(idle && !live);
            const int n_want = __popcll(want);
            if (n_want > 0 && !drained) {
                if (chunk_next >= chunk_end && !W.affine) {
                    uint32_t b = 0;
                    if (lane == 0) b = atomicAdd(&W.count[2], chunk);
                    chunk_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                    chunk_end = chunk_next + chunk < n_queue ? chunk_next + chunk : n_queue;
                    if (chunk_next >= n_queue) { drained = true; chunk_next = chunk_end = n_queue; }
                }
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
                        /* every pass of this loop advances a head, and a head only moves up: the loop ends when all eight are
                         * past their bands.  (No cap on the passes as in wf_intersect_kernel: a wave that gave up early here
                         * would leave slots that no later launch picks up.) */
                        if (where < 0) { drained = true; chunk_next = chunk_end = n_queue; }
                        else band = where;
                    }
                }
                const uint32_t base = chunk_next;
                const uint32_t avail = chunk_end - chunk_next;
                chunk_next += (uint32_t)n_want < avail ? (uint32_t)n_want : avail;
                if (idle && !live) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                    const uint32_t e = base + rank;
                    if (e < chunk_end) {
                        slot = e;
                        start_s = s_begin + W.stream0 + slot / W.n_pixels;      /* the slot's first sample (never more streams than samples) */
                        g.pixel = wf_frame_pixel(W, slot);
                        pst[3 * RT_BLOCK] = 0.f; pst[4 * RT_BLOCK] = 0.f; pst[5 * RT_BLOCK] = 0.f;
                        live = true;
                        starting = true;
                    }
                }
            }
            if (starting) {
                float4 o4, d4, a4;
                wf_start_sample<F>(S, W, slot, start_s, Wd, Hd, k0, k1, o4, d4, a4);
                rw.o = mk(o4.x, o4.y, o4.z); rw.time = o4.w; rw.d = mk(d4.x, d4.y, d4.z);
                g.sample = start_s; bounce = 0;
                pst[0] = 1.f; pst[RT_BLOCK] = 1.f; pst[2 * RT_BLOCK] = 1.f;
                want_ray = true;
            }
#ifdef RT_STAMPS_RF3
            const unsigned long long rf_t2 = __builtin_amdgcn_s_memtime();
#endif
            if (want_ray) {
                best.t = FLT_MAX; best.obj = -1; best.grp = -1; best.aux = 0;
                ++n_rays;
                scan_program<(F & ~RT_F_BVH)>(S, W.prog, rw, tMin, best, &g, bounce, stack);
                gidx = 0;
                if (MG) mode = WF_ENTER;      /* the batch enters its first group in the trip below, with whoever else waits */
                else WF_NEXT_GROUP
            }
#ifdef RT_STAMPS_RF3
            { const unsigned long long rf_t3 = __builtin_amdgcn_s_memtime(); rf3[0] += rf_t1 - rf_t0; rf3[1] += rf_t2 - rf_t1; rf3[2] += rf_t3 - rf_t2; rf3[3] += (unsigned long long)__popcll(shade_m); rf3[4] += (unsigned long long)__popcll(__ballot(want_ray)); }
#endif
        }
#ifdef RT_STAMPS_REFILL
        if (rf_on) { dg[4] += __builtin_amdgcn_s_memtime() - rf_t0; dg[5] += 1; }
#endif
        if (__ballot(mode != WF_NEED) == 0ull) {
            if (drained && __ballot(live) == 0ull) break;
            continue;
        }

        /* In the general variants the trips run as an inner loop of their own until the next batch is due: what only the
         * refill reads (the frame, the sample range, the claim state, the camera) then lives OUTSIDE the loop that runs ten
         * to twenty times as often, and the register allocator keeps its spills and reloads of those values out of the
         * trips (final scene 4.44 -> 4.60 Grays/s at 1080p).  The triangle variant, squeezed into 96 registers for its fifth
         * wave, does better as one flat loop (3.66 against 3.52). */
        bool again;
        do {
        /* ---- one trip: wf_intersect_kernel's, except that a ray which has finished a group does not enter the next one on
         * the spot.  Entering costs a wrapper chain and three divisions, about 70 instructions, and in a scene with two
         * groups two or three lanes finish their first group in every trip: run where it happens, the entry was a tenth of
         * the trip's instructions at 4 % of its lanes.  The lane waits (WF_ENTER) until enough lanes do, or the node steps
         * run low, like a postponed leaf. ---- */
        if (MG) {
            const int n_enter = __popcll(__ballot(mode == WF_ENTER));
            if (n_enter > 0) {
                const int n_walk = __popcll(__ballot(mode == WF_BVH && ref >= 0));
                if (n_enter >= WF4_ENTER_BATCH || n_walk < WF_NODE_LOW) {
                    if (mode == WF_ENTER) WF_NEXT_GROUP
                }
            }
        }
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
            dg[4] += __popcll(__ballot(in_bvh && !interior && !prim_step));
            dg[5] += n_node > 0 ? 1 : 0;
#endif
        }
#endif
        const int lfirst = (~pleaf) >> 4, lcount = ((~pleaf) & 15) + 1;
        const float4* src = reinterpret_cast<const float4*>(S.leafs + (prim_step ? lfirst + li : 0));
        /* the leaf row, loaded only by the lanes that take a primitive step.  The other lanes' copies must still be DEFINED
         * (left undefined, the allocator spilled: final scene 3.72 -> 2.65 Grays/s in wf_intersect_kernel), which cost 16 zero
         * moves per trip there; an empty asm that "writes" them defines them without an instruction (1M triangles +2.6 %,
         * final scene and sphere cluster +0.4 / +1.9 %) */
        float4 q0, q1, q2, q3;
#ifdef WF4_ZERO_FILL
        q0 = make_float4(0.f, 0.f, 0.f, 0.f); q1 = q0; q2 = q0; q3 = q0;
#else
        asm volatile("" : "=v"(q0.x), "=v"(q0.y), "=v"(q0.z), "=v"(q0.w), "=v"(q1.x), "=v"(q1.y), "=v"(q1.z), "=v"(q1.w));
        asm volatile("" : "=v"(q2.x), "=v"(q2.y), "=v"(q2.z), "=v"(q2.w), "=v"(q3.x), "=v"(q3.y), "=v"(q3.z), "=v"(q3.w));
#endif
        if (prim_step) { q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3]; }

        if (interior) {
            WF_NODE4_TEST(rl.o, id, soff, tMin, best.t)
#define WF_CSWAP(ta, ra, tb, rb) { const bool sw_ = tb < ta; const float tt_ = sw_ ? tb : ta; const int rr_ = sw_ ? rb : ra; tb = sw_ ? ta : tb; rb = sw_ ? ra : rb; ta = tt_; ra = rr_; }
            WF_CSWAP(t0, r0, t1, r1) WF_CSWAP(t2, r2, t3, r3) WF_CSWAP(t0, r0, t2, r2) WF_CSWAP(t1, r1, t3, r3) WF_CSWAP(t1, r1, t2, r2)
#undef WF_CSWAP
            if (r3 != RT_REF_NONE) WF_PUSH(r3)
            if (r2 != RT_REF_NONE) WF_PUSH(r2)
            if (r1 != RT_REF_NONE) WF_PUSH(r1)
            ref = r0;
            if (ref == RT_REF_NONE) WF_POP(ref)
        }
        if (prim_step) {
            /* (the compiler hoists the first uses of the row into the block of its loads, so the wave waits for the leaf row
             * before it asks for its node row; keeping them here with an asm barrier measured +-0: the pipes are busy anyway) */
            const uint32_t head = __float_as_uint(q0.x);
            const int oi = __float_as_int(q0.z);
            const float f[13] = {q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            float t; int aux;
            bool h;
            if (!(F & RT_F_XFORM) || obj_op_count(head) == 0) h = prim_t<F>(obj_type(head), f, rl, tMin, best.t, t, aux);
            else h = leaf_t<F>(S, S.objs[oi], rl, tMin, best.t, t, aux);
            h = h && q0.y == q0.y;            /* keeps the row's four loads aligned dwordx4 (rt_wavefront.h) */
            if (h) { best.t = t; best.obj = oi; best.grp = gi; best.aux = aux; }
            if (++li == lcount) { li = 0; pleaf = RT_REF_NONE; }
        }
        if (in_bvh) {
            if (pleaf == RT_REF_NONE && ref < 0 && ref != RT_REF_NONE) { pleaf = ref; li = 0; WF_POP(ref) }
            if (ref == RT_REF_NONE && pleaf == RT_REF_NONE) { ++gidx; mode = MG && gidx < W.n_groups ? WF_ENTER : WF_NEED; }   /* the group is done */
        }
            again = false;
            if (wf4_inner_loop(F)) {
                const bool idle2 = mode == WF_NEED;
                const int n_idle2 = __popcll(__ballot(idle2));
                const int n_work2 = __popcll(__ballot(idle2 && (live || !drained)));
                const int thr2 = drained ? (__popcll(__ballot(live)) + 3) >> 2 : W.refill_min;
                again = n_idle2 < 64 && !(n_work2 > 0 && n_work2 >= thr2);
            }
        } while (again);
    }
#undef WF_NEXT_GROUP
#undef WF_PUSH
#undef WF_POP
    unsigned long long r = n_rays;
    for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off);
    if (lane == 0) atomicAdd(&counters[0], r);
    if (blockIdx.x == 0 && threadIdx.x == 0 && W.stream0 == 0)
        atomicAdd(&counters[1], (unsigned long long)W.n_pixels * (unsigned long long)(s_end - s_begin));
#ifdef RT_STAMPS_RF3
    if (lane == 0) for (int q = 0; q < 15; ++q) atomicAdd(&counters[16 + q], rf3[q]);
#endif
#ifdef RT_STAMPS
    dg[7] = __builtin_amdgcn_s_memtime() - dg_t0;
    if (lane == 0) for (int q = 0; q < 8; ++q) atomicAdd(&counters[4 + (q < 6 ? q : q + 4)], dg[q]);
#endif
}

#endif
