/*
 * rt_device.h — device-side scene tables and the per-lane intersection / shading code.
 *
 * One lane = one path.  Nothing here is virtual: the reference's Hitable / Material /
 * Texture hierarchies (Hitable.cuh:23-28, Material.cuh:44-49, Texture.cuh:7-11) are
 * type-tagged table rows, and every hit()/scatter()/value() body is a switch arm that
 * keeps the reference's operand order so results are reproducible against the oracle
 * (the library is compiled with -ffp-contract=off; division and sqrt are IEEE).
 *
 * Closest hit works in two steps to keep the live state small: the search only carries
 * (t, object, group, face) and the full record (p, normal, u, v, material) is rebuilt
 * once for the winner by re-running that one primitive — the arithmetic is the same, so
 * the record is bit-identical to what the reference builds on the fly.
 */
#ifndef RT_DEVICE_H
#define RT_DEVICE_H
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>
#include "../../include/rt_scene.h"
#include "philox.h"
#include "rt_host.h"

#define RT_PI_D 3.14159265358979323846

/* diagnostic build only (-DRT_STAMPS): per-lane and per-wave trip counts of the BVH loops */
#ifdef RT_STAMPS
__device__ unsigned long long rt_dbg_counts[4];
#define RT_COUNT(i) { atomicAdd(&rt_dbg_counts[i], 1ull); if (__builtin_amdgcn_read_exec() != 0 && (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(__builtin_amdgcn_read_exec() >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)__builtin_amdgcn_read_exec(), 0u)) == 0) atomicAdd(&rt_dbg_counts[2 + i], 64ull); }
#else
#define RT_COUNT(i)
#endif
#define RT_BLOCK 256

struct DScene {
    const DObj*        __restrict__ objs;
    const rt_xform_op* __restrict__ ops;
    const rt_material* __restrict__ mats;
    const rt_texture*  __restrict__ texs;
    const uint8_t*     __restrict__ image;
    const DNode*       __restrict__ nodes;   /* 64-byte two-box nodes (rt_host.h) */
    const DNode4*      __restrict__ nodes4;  /* 128-byte four-box nodes of the same trees (wavefront) */
    const DObj*        __restrict__ leafs;   /* BVH leaf table: first 64 bytes of each row are read */
    const DNode8*      __restrict__ nodes8;  /* 80-byte eight-wide quantised nodes of the same trees (rt_wavefront8.h) */
    const DLeaf*       __restrict__ leafs8;  /* their leaf table: 16 rows of 64 bytes per node */
    const DProg*       __restrict__ prog;    /* the world list as 32-byte scan records (rt_host.h) */
    int32_t world;
    int32_t n_media;
    rt_camera cam;
    const rt_camera*   __restrict__ cam_ptr; /* the same 96 bytes in device memory, for cam_fetch() */
};

__device__ __forceinline__ int      obj_type(uint32_t head)     { return (int)(head & 0xffu); }
__device__ __forceinline__ uint32_t obj_flags(uint32_t head)    { return (head >> 8) & 0xffu; }
__device__ __forceinline__ int      obj_op_count(uint32_t head) { return (int)((head >> 16) & 0xffu); }

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float a, float b, float c) { V3 r = {a, b, c}; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(float t, V3 v) { return mk(t * v.x, t * v.y, t * v.z); }
__device__ __forceinline__ V3 operator/(V3 v, float t) { return mk(v.x / t, v.y / t, v.z / t); }
__device__ __forceinline__ V3 operator-(V3 v) { return mk(-v.x, -v.y, -v.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b)
{
    return mk((a.y * b.z - a.z * b.y), (-(a.x * b.z - a.z * b.x)), (a.x * b.y - a.y * b.x));
}
__device__ __forceinline__ float length(V3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
__device__ __forceinline__ V3 unit(V3 v) { return v / length(v); }
__device__ __forceinline__ V3 ld3(const float* f) { return mk(f[0], f[1], f[2]); }

/* The divisions of the primitive tests.  Shipped build: IEEE division, like the reference's host build and the oracle
 * (bit-exact t).  -DRT_RELAXED=1|2 (make relaxed: a MEASUREMENT build under tools/diag, never loaded by tests or bench
 * by default) replaces them with v_rcp_f32 (1 ulp) and, at 2, one Newton step on the reciprocal: what north_star's
 * tolerance (t within 1e-5 relative) would allow.  DESIGN.md section 6 has what it buys. */
#ifdef RT_RELAXED
__device__ __forceinline__ float rt_div(float a, float b)
{
    float r = __builtin_amdgcn_rcpf(b);
#if RT_RELAXED >= 2
    r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.f), r, r);
#endif
    return a * r;
}
#else
__device__ __forceinline__ float rt_div(float a, float b) { return a / b; }
#endif

struct DRay { V3 o, d; float time; };
__device__ __forceinline__ V3 at(const DRay& r, float t) { return r.o + t * r.d; }

/* what the closest-hit search carries */
struct HitLite {
    float   t;
    int32_t obj;   /* winning leaf object, -1 = miss */
    int32_t grp;   /* group object it was reached through, -1 = directly in the world list */
    int32_t aux;   /* Box: which of the six faces produced t */
};

/* the rebuilt record (Intersect, Hitable.cuh:10-22) */
struct HitRec {
    float t, u, v;
    V3 p, n;
    int32_t material, prim_id, front;
};

/* ------------------------------------------------------------------ RNG ---- */
struct PathRng {
    uint32_t k0, k1, pixel, sample;
};
__device__ __forceinline__ void rng_start(PathRng& g, uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample)
{
    g.k0 = k0; g.k1 = k1; g.pixel = pixel; g.sample = sample;
}
/* the block owned by (purpose, c0) at `bounce` of this path — see philox.h */
__device__ __forceinline__ rt_u32x4 rng_block(const PathRng& g, uint32_t purpose, uint32_t bounce, uint32_t c0)
{
    return rt_philox4x32_10(c0, (purpose << 24) | bounce, g.pixel, g.sample, g.k0, g.k1);
}
__device__ __forceinline__ float rng_medium(const PathRng& g, uint32_t bounce, int slot)
{
    return rt_u32_to_uniform(rng_block(g, RT_RNG_MEDIUM, bounce, (uint32_t)slot).x);
}

/* ----------------------------------------------------------- primitives ---- */
/* Each *_t function answers "does it hit in (tMin,tMax), and at what t" with the
 * reference's arithmetic; the *_rec twin rebuilds the record for that t. */

/* Sphere::hit Sphere.cuh:25-56 / MovingSphere::hit :91-120 (centre passed in) */
__device__ __forceinline__ bool sphere_t(V3 centre, float radius, const DRay& r, float tMin, float tMax, float& t)
{
    V3 oc = r.o - centre;
    float a = dot(r.d, r.d);
    float b = 2.f * dot(oc, r.d);
    float c = dot(oc, oc) - radius * radius;
    float discrim = b * b - 4.f * a * c;
    if (discrim > 0) {
        float sq = sqrtf(discrim);
        float temp = rt_div(-b - sq, 2 * a);
        if (temp < tMax && temp > tMin) { t = temp; return true; }
        temp = rt_div(-b + sq, 2 * a);
        if (temp < tMax && temp > tMin) { t = temp; return true; }
    }
    return false;
}
/* The two boundary queries of ConstantMedium::hit (Volumes.cuh:29-30) against a Sphere boundary: the roots do not
 * depend on the query's range, so hit(r, -FLT_MAX, FLT_MAX) and hit(r, t1 + .0001f, FLT_MAX) share one
 * discriminant, one square root and the two divisions.  Returns the number of queries that hit (0, 1, 2). */
__device__ __forceinline__ int sphere_two_t(V3 centre, float radius, const DRay& r, float& t1, float& t2)
{
    V3 oc = r.o - centre;
    float a = dot(r.d, r.d);
    float b = 2.f * dot(oc, r.d);
    float c = dot(oc, oc) - radius * radius;
    float discrim = b * b - 4.f * a * c;
    if (!(discrim > 0)) return 0;
    const float sq = sqrtf(discrim);
    const float lo = rt_div(-b - sq, 2 * a), hi = rt_div(-b + sq, 2 * a);
    if (lo < FLT_MAX && lo > -FLT_MAX) t1 = lo;
    else if (hi < FLT_MAX && hi > -FLT_MAX) t1 = hi;
    else return 0;
    const float t0 = t1 + .0001f;
    if (lo < FLT_MAX && lo > t0) t2 = lo;
    else if (hi < FLT_MAX && hi > t0) t2 = hi;
    else return 1;
    return 2;
}
__device__ __forceinline__ V3 moving_centre(const float* f, float time)
{
    V3 c0 = ld3(f), c1 = ld3(f + 3);
    return c0 + ((time - f[6]) / (f[7] - f[6])) * (c1 - c0);     /* Sphere.cuh:86-89 */
}

/* XYRect/XZRect/YZRect::hit Rectangle.cuh:24-38,55-69,86-100; KA = fixed axis */
template <int KA>
__device__ __forceinline__ bool rect_t(float a0, float a1, float b0, float b1, float k, const DRay& r,
                                       float t0, float t1, float& t_out)
{
    float ok = KA == 2 ? r.o.z : (KA == 1 ? r.o.y : r.o.x);
    float dk = KA == 2 ? r.d.z : (KA == 1 ? r.d.y : r.d.x);
    float oa = KA == 0 ? r.o.y : r.o.x;
    float da = KA == 0 ? r.d.y : r.d.x;
    float ob = KA == 2 ? r.o.y : r.o.z;
    float db = KA == 2 ? r.d.y : r.d.z;
    float t = rt_div(k - ok, dk);
    float a = oa + t * da;
    float b = ob + t * db;
    /* the reference's two early returns, evaluated without branching: same comparisons, same result
     * (NaNs fail none of the reject tests, exactly as in `if (t<t0 || t>t1) return false`) */
    bool reject = (t < t0) | (t > t1) | (a < a0) | (a > a1) | (b < b0) | (b > b1);
    t_out = t;
    return !reject;
}
__device__ __forceinline__ bool rect_any_t(int type, const float* f, const DRay& r, float t0, float t1, float& t)
{
    if (type == RT_OBJ_RECT_XY) return rect_t<2>(f[0], f[1], f[2], f[3], f[4], r, t0, t1, t);
    if (type == RT_OBJ_RECT_XZ) return rect_t<1>(f[0], f[1], f[2], f[3], f[4], r, t0, t1, t);
    return rect_t<0>(f[0], f[1], f[2], f[3], f[4], r, t0, t1, t);
}
/* record of a rect hit at t: u,v,p and the ray-facing axis normal (setFaceNormal, Hitable.cuh:17-21).
 * The fixed axis KA is a run-time value picked with selects, not a template argument: the winners of the
 * lanes of a wave are rects of all three kinds and box faces, and one branch-free body serves them all
 * (a switch over the kinds ran every arm for a sixth of the lanes each). */
template <bool WANT_UV>
__device__ __forceinline__ void rect_rec(int KA, float a0, float a1, float b0, float b1, const DRay& r, float t, bool want_uv, HitRec& h)
{
    float oa = KA == 0 ? r.o.y : r.o.x;
    float da = KA == 0 ? r.d.y : r.d.x;
    float ob = KA == 2 ? r.o.y : r.o.z;
    float db = KA == 2 ? r.d.y : r.d.z;
    float dk = KA == 2 ? r.d.z : (KA == 1 ? r.d.y : r.d.x);
    if (WANT_UV || want_uv) {
        float a = oa + t * da;
        float b = ob + t * db;
        h.u = (a - a0) / (a1 - a0);
        h.v = (b - b0) / (b1 - b0);
    }
    h.t = t;
    h.p = at(r, t);
    bool front = dk < 0;           /* dot(dir, +axis) < 0 */
    float s = front ? 1.f : -1.f;
    float z = front ? 0.f : -0.f;
    h.n = mk(KA == 0 ? s : z, KA == 1 ? s : z, KA == 2 ? s : z);
    h.front = front;
}

/* Box = six rects in list order (Box.cuh:24-36); returns the winning face */
__device__ __forceinline__ bool box_t(const float* f, const DRay& r, float t0, float t1, float& t_out, int& face)
{
    bool any = false;
    float closest = t1, t;
    int fc = 0;
#define RT_FACE(KA, A0, A1, B0, B1, K, ID)                                              \
    {                                                                                   \
        bool h_ = rect_t<KA>(A0, A1, B0, B1, K, r, t0, closest, t);                     \
        any |= h_;                                                                      \
        closest = h_ ? t : closest;                                                     \
        fc = h_ ? ID : fc;                                                              \
    }
    RT_FACE(2, f[0], f[3], f[1], f[4], f[5], 0)
    RT_FACE(2, f[0], f[3], f[1], f[4], f[2], 1)
    RT_FACE(1, f[0], f[3], f[2], f[5], f[4], 2)
    RT_FACE(1, f[0], f[3], f[2], f[5], f[1], 3)
    RT_FACE(0, f[1], f[4], f[2], f[5], f[3], 4)
    RT_FACE(0, f[1], f[4], f[2], f[5], f[0], 5)
#undef RT_FACE
    t_out = closest;
    face = fc;
    return any;
}

/* The two boundary queries of ConstantMedium::hit (Volumes.cuh:29-30) against a Box boundary, in one pass over
 * the faces: boundary->hit(r, -FLT_MAX, FLT_MAX) then boundary->hit(r, t1 + .0001f, FLT_MAX).  A face's t and its
 * bounds test do not depend on the query's range, so both queries run the reference's accept sequence (list order,
 * shrinking closest-so-far, the same comparisons) over one set of six divisions.  Returns the number of queries
 * that hit (0, 1, 2). */
__device__ __forceinline__ int box_two_t(const float* f, const DRay& r, float& t1_out, float& t2_out)
{
    float t[6];
    bool oob[6];
#define RT_FACE_T(N, KA, A0, A1, B0, B1, K)                                             \
    {                                                                                   \
        const float ok_ = KA == 2 ? r.o.z : (KA == 1 ? r.o.y : r.o.x);                  \
        const float dk_ = KA == 2 ? r.d.z : (KA == 1 ? r.d.y : r.d.x);                  \
        const float oa_ = KA == 0 ? r.o.y : r.o.x, da_ = KA == 0 ? r.d.y : r.d.x;       \
        const float ob_ = KA == 2 ? r.o.y : r.o.z, db_ = KA == 2 ? r.d.y : r.d.z;       \
        t[N] = rt_div(K - ok_, dk_);                                                       \
        const float a_ = oa_ + t[N] * da_, b_ = ob_ + t[N] * db_;                       \
        oob[N] = (a_ < A0) | (a_ > A1) | (b_ < B0) | (b_ > B1);                         \
    }
    RT_FACE_T(0, 2, f[0], f[3], f[1], f[4], f[5])
    RT_FACE_T(1, 2, f[0], f[3], f[1], f[4], f[2])
    RT_FACE_T(2, 1, f[0], f[3], f[2], f[5], f[4])
    RT_FACE_T(3, 1, f[0], f[3], f[2], f[5], f[1])
    RT_FACE_T(4, 0, f[1], f[4], f[2], f[5], f[3])
    RT_FACE_T(5, 0, f[1], f[4], f[2], f[5], f[0])
#undef RT_FACE_T
    bool any = false;
    float closest = FLT_MAX;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const bool h = !((t[k] < -FLT_MAX) | (t[k] > closest) | oob[k]);
        any |= h;
        closest = h ? t[k] : closest;
    }
    if (!any) return 0;
    t1_out = closest;
    const float t0 = closest + .0001f;
    any = false;
    closest = FLT_MAX;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const bool h = !((t[k] < t0) | (t[k] > closest) | oob[k]);
        any |= h;
        closest = h ? t[k] : closest;
    }
    t2_out = closest;
    return any ? 2 : 1;
}

/* Triangle::hit Triangle.cuh:21-46 with the unit normal and plane offset precomputed on the
 * host by the same expressions (f[9..12]); plus the tMin<t<tMax filter */
__device__ __forceinline__ bool triangle_t(const float* f, const DRay& r, float tMin, float tMax, float& t_out)
{
    V3 v0 = ld3(f), v1 = ld3(f + 3), v2 = ld3(f + 6), normal = ld3(f + 9);
    float d = f[12];
    float distance = rt_div(-dot(r.o, normal) + d, dot(r.d, normal));
    if (distance > FLT_EPSILON) {
        V3 inter = at(r, distance);
        float a = dot(cross(v1 - v0, inter - v0), normal);
        float b = dot(cross(v2 - v1, inter - v1), normal);
        float c = dot(cross(v0 - v2, inter - v2), normal);
        if (a > 0 && b > 0 && c > 0) {
            if (!(distance < tMax && distance > tMin)) return false;
            t_out = distance;
            return true;
        }
    }
    return false;
}

/* ------------------------------------------------------------ transforms ---- */
/* Translate::hit Transform.cuh:19-28 and RotateY::hit :90-112, ray side */
__device__ __forceinline__ void apply_op(int type, float a, float b, float c3, DRay& r)
{
    if (type == RT_OP_TRANSLATE) {
        r.o = r.o - mk(a, b, c3);
    } else {
        float s = a, c = b;
        float ox = c * r.o.x - s * r.o.z, oz = s * r.o.x + c * r.o.z;
        float dx = c * r.d.x - s * r.d.z, dz = s * r.d.x + c * r.d.z;
        r.o.x = ox; r.o.z = oz; r.d.x = dx; r.d.z = dz;
    }
}
__device__ __forceinline__ void apply_chain(const DScene& S, const DObj& o, DRay& r)
{
    const int n = obj_op_count(o.head);
    if (n > 0) apply_op((int)((o.head >> 24) & 0xfu), o.op0[0], o.op0[1], o.op0[2], r);
    if (n > 1) apply_op((int)((o.head >> 28) & 0xfu), o.op1[0], o.op1[1], o.op1[2], r);
    for (int k = 2; k < n; ++k) { const rt_xform_op p = S.ops[o.op_begin + k]; apply_op(p.type, p.a, p.b, p.c, r); }
}

/* bare primitive (ray already in the object's frame): t only.  F = the RT_F_* features compiled in. */
template <uint32_t F>
__device__ __forceinline__ bool prim_t(int type, const float* f, const DRay& r, float tMin, float tMax, float& t, int& aux)
{
    aux = 0;
    if ((F & RT_F_SPHERE) && type == RT_OBJ_SPHERE) return sphere_t(ld3(f), f[3], r, tMin, tMax, t);
    if ((F & RT_F_MSPHERE) && type == RT_OBJ_MOVING_SPHERE) return sphere_t(moving_centre(f, r.time), f[8], r, tMin, tMax, t);
    if ((F & RT_F_PLANAR) && type >= RT_OBJ_RECT_XY && type <= RT_OBJ_RECT_YZ) return rect_any_t(type, f, r, tMin, tMax, t);
    if ((F & RT_F_PLANAR) && type == RT_OBJ_BOX) return box_t(f, r, tMin, tMax, t, aux);
    if ((F & RT_F_TRI) && type == RT_OBJ_TRIANGLE) return triangle_t(f, r, tMin, tMax, t);
    return false;
}

/* leaf with its own wrapper chain */
template <uint32_t F>
__device__ __forceinline__ bool leaf_t(const DScene& S, const DObj& o, DRay r, float tMin, float tMax, float& t, int& aux)
{
    if (F & RT_F_XFORM) apply_chain(S, o, r);
    return prim_t<F>(obj_type(o.head), o.f, r, tMin, tMax, t, aux);
}

/* A row of the object table read through the constant address space (see prog_load): where the index is the
 * same for the whole wave — the objects of the world list, a group's children, a medium's boundary — the row
 * arrives by scalar loads and lives in SGPRs; a lane-dependent index still works (vector loads). */
__device__ __forceinline__ DObj obj_const(const DObj* p)
{
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    typedef const __attribute__((address_space(4))) u32x16* const_half_ptr;
    static_assert(sizeof(DObj) == 2 * sizeof(u32x16), "DObj is two 64-byte halves");
    const u32x16 w[2] = {((const_half_ptr)(uintptr_t)p)[0], ((const_half_ptr)(uintptr_t)p)[1]};
    DObj o;
    __builtin_memcpy(&o, w, sizeof o);
    return o;
}

/* ConstantMedium::hit Volumes.cuh:26-49 (ray already in the medium's frame) */
template <uint32_t F>
__device__ __forceinline__ bool medium_t(const DScene& S, const DObj& o, const DRay& r, float tMin, float tMax,
                                         const PathRng& g, uint32_t bounce, float& t_out)
{
    const DObj b = obj_const(S.objs + o.child_begin);
    float t1, t2;
    int unused;
    if ((F & RT_F_PLANAR) && obj_type(b.head) == RT_OBJ_BOX) {
        DRay rb = r;
        if (F & RT_F_XFORM) apply_chain(S, b, rb);
        if (box_two_t(b.f, rb, t1, t2) < 2) return false;
    } else if ((F & RT_F_SPHERE) && obj_type(b.head) == RT_OBJ_SPHERE) {
        DRay rb = r;
        if (F & RT_F_XFORM) apply_chain(S, b, rb);
        if (sphere_two_t(ld3(b.f), b.f[3], rb, t1, t2) < 2) return false;
    } else {
        if (!leaf_t<F>(S, b, r, -FLT_MAX, FLT_MAX, t1, unused)) return false;
        if (!leaf_t<F>(S, b, r, t1 + .0001f, FLT_MAX, t2, unused)) return false;
    }
    if (t1 < tMin) t1 = tMin;
    if (t2 > tMax) t2 = tMax;
    if (t1 >= t2) return false;
    if (t1 < 0) t1 = 0;
    const float rayLength = length(r.d);
    const float distanceInsideBoundary = (t2 - t1) * rayLength;
    const float hitDistance = o.f[0] * logf(rng_medium(g, bounce, o.medium_slot));
    if (hitDistance > distanceInsideBoundary) return false;
    t_out = t1 + hitDistance / rayLength;
    return true;
}

/* ------------------------------------------------------------ BVH traversal ---- */
/* Slab test of one child box with the reciprocal direction.  Culling never decides a result (the
 * primitives do, with the reference's arithmetic), it only has to be conservative: the far bound is
 * widened by a few ulps so that rounding in (b - o) * (1/d) cannot drop a box the exact test keeps
 * (host-side boxes are padded as well, lower.cpp pad_box). */
__device__ __forceinline__ bool slab(const float* bmin, const float* bmax, V3 o, V3 id, float tMin, float tMax, float& tnear)
{
    float x0 = (bmin[0] - o.x) * id.x, x1 = (bmax[0] - o.x) * id.x;
    float y0 = (bmin[1] - o.y) * id.y, y1 = (bmax[1] - o.y) * id.y;
    float z0 = (bmin[2] - o.z) * id.z, z1 = (bmax[2] - o.z) * id.z;
    float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tMin));
    float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tMax));
    tnear = tn;
    return tn <= tf * 1.0000008f;
}

/* Closest hit over one BVH.  One loop, one step per lane per trip: a lane holding an interior
 * reference tests that node's two child boxes; a lane that has reached a leaf POSTPONES it (`pleaf`) and goes
 * on with the next node from its stack, and the leaf's primitives are tested one per trip in trips where
 * enough lanes hold a postponed leaf (or the node steps run low).  A trip pays for every kind of step any
 * lane runs, and a primitive test (a Box is six rects) costs several node steps: run whenever a single lane
 * had reached a leaf, it executed with 7 % of its lanes on the final scene.  The closest hit does not depend
 * on the order in which a ray's leaves are tested (exact ties aside).
 * A while-while traversal (walk to a leaf, then test the leaf) measured 9 % lane efficiency on the
 * 1M-triangle scene at 64 lanes: every trip of either inner loop waits for the wave's slowest lane.
 * The per-lane stack lives in LDS, entry i at stack[i * RT_BLOCK]; near child first, far child pushed,
 * best.t shrinking.  Called by the live lanes of a wave together; a lane that is done idles until the
 * last one is (it would wait at the reconvergence point anyway). */
#ifndef RT_BVH_PRIM_BATCH
#define RT_BVH_PRIM_BATCH 16   /* lanes holding a leaf that make a primitive step worth running */
#define RT_BVH_NODE_LOW 8      /* fewer interior lanes than this: run the primitive step anyway */
#endif
template <uint32_t F>
__device__ __forceinline__ void bvh_closest(const DScene& S, int gi, int root, const DRay& r, float tMin,
                                            HitLite& best, int* stack)
{
    const V3 id = mk(1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z);
    int sp = 0;
    int ref = root;
    int pleaf = RT_REF_NONE;                      /* the postponed leaf */
    int li = 0;                                   /* next primitive of the postponed leaf */
    while (true) {
        if (pleaf == RT_REF_NONE && ref < 0 && ref != RT_REF_NONE) {
            pleaf = ref; li = 0;
            ref = RT_REF_NONE;
            if (sp > 0) { --sp; ref = stack[sp * RT_BLOCK]; }
        }
        const bool interior = ref >= 0;
        const bool has_leaf = pleaf != RT_REF_NONE;
        const int n_node = __popcll(__ballot(interior)), n_prim = __popcll(__ballot(has_leaf));
        if (n_node + n_prim == 0) break;
        const bool prim_step = has_leaf && (n_prim >= RT_BVH_PRIM_BATCH || n_node < RT_BVH_NODE_LOW);
        /* the loads of both steps of the trip go out together */
        const int lfirst = (~pleaf) >> 4, lcount = ((~pleaf) & 15) + 1;
        const float4* nsrc = reinterpret_cast<const float4*>(S.nodes + (interior ? ref : 0));
        const float4* lsrc = reinterpret_cast<const float4*>(S.leafs + (prim_step ? lfirst + li : 0));
        float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0, n3 = n0, q0 = n0, q1 = n0, q2 = n0, q3 = n0;
        if (interior) { n0 = nsrc[0]; n1 = nsrc[1]; n2 = nsrc[2]; n3 = nsrc[3]; }
        if (prim_step) { q0 = lsrc[0]; q1 = lsrc[1]; q2 = lsrc[2]; q3 = lsrc[3]; }
        if (interior) {
            RT_COUNT(0)
            const float lmin[3] = {n0.x, n0.y, n0.z}, lmax[3] = {n0.w, n1.x, n1.y};
            const float rmin[3] = {n1.z, n1.w, n2.x}, rmax[3] = {n2.y, n2.z, n2.w};
            const int left = __float_as_int(n3.x), right = __float_as_int(n3.y);
            float tl, tr;
            const bool hl = slab(lmin, lmax, r.o, id, tMin, best.t, tl);
            const bool hr = slab(rmin, rmax, r.o, id, tMin, best.t, tr);
            if (hl & hr) {
                const bool lnear = tl <= tr;
                stack[sp * RT_BLOCK] = lnear ? right : left;
                ++sp;
                ref = lnear ? left : right;
            } else if (hl | hr) {
                ref = hl ? left : right;
            } else {
                ref = RT_REF_NONE;
                if (sp > 0) { --sp; ref = stack[sp * RT_BLOCK]; }
            }
        }
        if (prim_step) {
            RT_COUNT(1)
            const uint32_t head = __float_as_uint(q0.x);
            const int oi = __float_as_int(q0.z);                 /* leaf rows carry the object index here */
            const float f[13] = {q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            float t; int aux;
            bool h;
            if (!(F & RT_F_XFORM) || obj_op_count(head) == 0) h = prim_t<F>(obj_type(head), f, r, tMin, best.t, t, aux);
            else h = leaf_t<F>(S, S.objs[oi], r, tMin, best.t, t, aux);     /* wrapped leaf: full row */
            if (h) { best.t = t; best.obj = oi; best.grp = gi; best.aux = aux; }
            if (++li == lcount) { li = 0; pleaf = RT_REF_NONE; }
        }
    }
}

/* ------------------------------------------------------------ closest hit ---- */
/* one object through its row: a group (BVH or list), a medium, or a wrapped / bare leaf */
template <uint32_t F>
__device__ __forceinline__ void group_closest(const DScene& S, int gi, const DObj& g, DRay r, float tMin,
                                              HitLite& best, const PathRng* rng, uint32_t bounce, int* stack)
{
    if (F & RT_F_XFORM) apply_chain(S, g, r);
    if ((F & RT_F_BVH) && (obj_flags(g.head) & RT_DOBJ_HAS_BVH)) {
        bvh_closest<F>(S, gi, g.bvh_root, r, tMin, best, stack);
        return;
    }
    if (!(F & RT_F_GROUPLIST) && gi >= 0) return;
    for (int i = 0; i < g.child_count; ++i) {
        int oi = g.child_begin + i;
        const DObj o = obj_const(S.objs + oi);
        float t; int aux;
        if ((F & RT_F_MEDIA) && obj_type(o.head) == RT_OBJ_MEDIUM) {
            if (rng) {
                DRay rm = r;
                if (F & RT_F_XFORM) apply_chain(S, o, rm);
                if (medium_t<F>(S, o, rm, tMin, best.t, *rng, bounce, t)) { best.t = t; best.obj = oi; best.grp = gi; best.aux = 0; }
            }
        } else if (leaf_t<F>(S, o, r, tMin, best.t, t, aux)) { best.t = t; best.obj = oi; best.grp = gi; best.aux = aux; }
    }
}

template <uint32_t F>
__device__ __forceinline__ void generic_closest(const DScene& S, int oi, const DRay& r, float tMin, HitLite& best,
                                                const PathRng* rng, uint32_t bounce, int* stack)
{
    const DObj o = obj_const(S.objs + oi);
    const int ty = obj_type(o.head);
    float t; int aux;
    if (ty == RT_OBJ_GROUP) {
        if (F & (RT_F_BVH | RT_F_GROUPLIST)) group_closest<F>(S, oi, o, r, tMin, best, rng, bounce, stack);
    } else if (ty == RT_OBJ_MEDIUM) {
        if ((F & RT_F_MEDIA) && rng) {
            DRay rm = r;
            if (F & RT_F_XFORM) apply_chain(S, o, rm);
            if (medium_t<F>(S, o, rm, tMin, best.t, *rng, bounce, t)) { best.t = t; best.obj = oi; best.grp = -1; best.aux = 0; }
        }
    } else if (leaf_t<F>(S, o, r, tMin, best.t, t, aux)) { best.t = t; best.obj = oi; best.grp = -1; best.aux = aux; }
}

/* one 32-byte scan record, fetched through the scalar cache */
struct ProgRec { uint32_t head; int32_t obj; float v0, v1, v2, v3, v4, v5; };
/* The record is read through the constant address space: the program is never written while a kernel runs, and
 * a load from a wave-uniform constant address is a scalar load (s_load_dwordx8) whatever else the kernel stores.
 * Read as ordinary global memory it stops being one as soon as the loop around it holds a store the compiler
 * cannot tell apart from the program (the per-run flush of render_kernel, the hit write of wf_intersect): the
 * records then come through 64 identical vector loads and every test on them runs on the VALU. */
typedef uint32_t rt_u32x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ ProgRec prog_load(const DProg* p)
{
    typedef const __attribute__((address_space(4))) rt_u32x8_t* const_rec_ptr;
    const rt_u32x8_t w = *(const_rec_ptr)(uintptr_t)p;
    ProgRec r;
    r.head = w[0]; r.obj = (int32_t)w[1];
    r.v0 = __uint_as_float(w[2]); r.v1 = __uint_as_float(w[3]);
    r.v2 = __uint_as_float(w[4]); r.v3 = __uint_as_float(w[5]); r.v4 = __uint_as_float(w[6]); r.v5 = __uint_as_float(w[7]);
    return r;
}

/* HitableList::hit over the world (HitableList.cuh:17-32): list order, tMax shrinks.  The list is run
 * as the scan program of rt_host.h: every lane of the wave executes the same record, so records come
 * through the scalar cache and the next one is requested before the current one is evaluated. */
template <uint32_t F>
__device__ __forceinline__ void scan_program(const DScene& S, const DProg* pc, const DRay& r, float tMin, HitLite& best,
                                             const PathRng* rng, uint32_t bounce, int* stack)
{
    DRay rl = r;                                   /* the ray in the current record's frame */
    ProgRec cur = prog_load(pc);
    while (true) {
        const int kind = (int)(cur.head & 0xffu);
        if (kind == RT_P_END) break;
        ++pc;
        const ProgRec nxt = prog_load(pc);
        if (kind == RT_P_RECT) {
            if (F & RT_F_PLANAR) {
                /* The record's axis is wave-uniform: the ray components are picked with selects on a scalar
                 * condition and ONE body (Rectangle.cuh:24-38 with a = x, b = y, k = z) serves the three
                 * kinds; three bodies behind a scalar branch were merged by the compiler into one behind
                 * ~15 register moves.  The winner is kept with selects as well. */
                const int ka = (int)((cur.head >> 8) & 0xffu);
                DRay ra;
                ra.o = mk(ka == 0 ? rl.o.y : rl.o.x, ka == 2 ? rl.o.y : rl.o.z, ka == 2 ? rl.o.z : (ka == 1 ? rl.o.y : rl.o.x));
                ra.d = mk(ka == 0 ? rl.d.y : rl.d.x, ka == 2 ? rl.d.y : rl.d.z, ka == 2 ? rl.d.z : (ka == 1 ? rl.d.y : rl.d.x));
                ra.time = rl.time;
                float t;
                const bool h = rect_t<2>(cur.v1, cur.v2, cur.v3, cur.v4, cur.v0, ra, tMin, best.t, t);
                best.t = h ? t : best.t;
                best.obj = h ? cur.obj : best.obj;
                best.grp = h ? -1 : best.grp;
                best.aux = h ? (int)((cur.head >> 16) & 0xffu) : best.aux;
            }
        } else if (kind == RT_P_BOX) {
            if (F & RT_F_PLANAR) {
                const float f[6] = {cur.v0, cur.v1, cur.v2, cur.v3, cur.v4, cur.v5};
                float t; int face;
                const bool h = box_t(f, rl, tMin, best.t, t, face);
                best.t = h ? t : best.t;
                best.obj = h ? cur.obj : best.obj;
                best.grp = h ? -1 : best.grp;
                best.aux = h ? face : best.aux;
            }
        } else if (kind == RT_P_XFORM) {
            if (F & RT_F_XFORM) {
                rl = r;
                const int n = (int)((cur.head >> 8) & 0xffu);
                if (n > 0) apply_op((int)((cur.head >> 16) & 0xfu), cur.v0, cur.v1, cur.v2, rl);
                if (n > 1) apply_op((int)((cur.head >> 20) & 0xfu), cur.v3, cur.v4, cur.v5, rl);
            }
        } else if (kind == RT_P_SPHERE) {
            if (F & RT_F_SPHERE) {
                float t;
                if (sphere_t(mk(cur.v0, cur.v1, cur.v2), cur.v3, rl, tMin, best.t, t)) { best.t = t; best.obj = cur.obj; best.grp = -1; best.aux = 0; }
            }
        } else {
            generic_closest<F>(S, cur.obj, r, tMin, best, rng, bounce, stack);
        }
        cur = nxt;
    }
}
template <uint32_t F>
__device__ __forceinline__ HitLite closest_hit(const DScene& S, const DRay& r, float tMin, float tMax,
                                               const PathRng* rng, uint32_t bounce, int* stack)
{
    HitLite best = {tMax, -1, -1, 0};
    if (F & RT_F_BVH) {
        const DObj& w = S.objs[S.world];
        if (obj_flags(w.head) & RT_DOBJ_HAS_BVH) {
            group_closest<F>(S, -1, w, r, tMin, best, rng, bounce, stack);
            return best;
        }
    }
    scan_program<F>(S, S.prog, r, tMin, best, rng, bounce, stack);
    return best;
}

/* --------------------------------------------------- rebuild the record ---- */
/* What rebuild needs of the winner's row and of its group's row, fetched by the lane in ONE round
 * trip (every address is known from the HitLite) instead of field by field along the way. */
struct RowRegs {
    uint32_t head; int32_t material, prim_id;
    float f[13];
    float op0[3], op1[3];
    int32_t op_begin;
};
template <uint32_t F, bool FULL>
__device__ __forceinline__ void row_load(const DObj* p, RowRegs& w)
{
    const float4* q = reinterpret_cast<const float4*>(p);
    const float4 a = q[0], b = q[1], c = q[2];
    w.head = __float_as_uint(a.x); w.material = __float_as_int(a.y); w.prim_id = __float_as_int(a.z);
    w.f[0] = a.w; w.f[1] = b.x; w.f[2] = b.y; w.f[3] = b.z; w.f[4] = b.w;
    w.f[5] = c.x; w.f[6] = c.y; w.f[7] = c.z; w.f[8] = c.w;
    if (FULL && (F & (RT_F_TRI | RT_F_MSPHERE))) { const float4 d = q[3]; w.f[9] = d.x; w.f[10] = d.y; w.f[11] = d.z; w.f[12] = d.w; }
    else { w.f[9] = w.f[10] = w.f[11] = w.f[12] = 0.f; }
    if (F & RT_F_XFORM) {
        const float4 e = q[4], g = q[5];
        w.op0[0] = e.x; w.op0[1] = e.y; w.op0[2] = e.z; w.op1[0] = e.w; w.op1[1] = g.x; w.op1[2] = g.y;
        w.op_begin = __float_as_int(g.z);
    } else {
        w.op0[0] = w.op0[1] = w.op0[2] = w.op1[0] = w.op1[1] = w.op1[2] = 0.f;
        w.op_begin = 0;
    }
}
struct OpV { int type; float a, b, c; };
template <uint32_t F>
__device__ __forceinline__ OpV row_op(const DScene& S, const RowRegs& o, int k)
{
    OpV v;
    if (k == 0) { v.type = (int)((o.head >> 24) & 0xfu); v.a = o.op0[0]; v.b = o.op0[1]; v.c = o.op0[2]; }
    else if (k == 1 || !(F & RT_F_LONGCHAIN)) { v.type = (int)((o.head >> 28) & 0xfu); v.a = o.op1[0]; v.b = o.op1[1]; v.c = o.op1[2]; }
    else { const rt_xform_op p = S.ops[o.op_begin + k]; v.type = p.type; v.a = p.a; v.b = p.b; v.c = p.c; }
    return v;
}
/* combined wrapper chain of the winner: the group's ops then the leaf's, outermost first */
template <uint32_t F>
__device__ __forceinline__ OpV chain_op(const DScene& S, const RowRegs& grp, int gc, const RowRegs& o, int k)
{
    return k < gc ? row_op<F>(S, grp, k) : row_op<F>(S, o, k - gc);
}

template <uint32_t F, bool WANT_UV>
__device__ __forceinline__ void rebuild(const DScene& S, const DRay& world_ray, const HitLite& hl, float tMin, HitRec& h)
{
    RowRegs o, grp;
    row_load<F, true>(S.objs + hl.obj, o);
    const bool in_group = (F & (RT_F_BVH | RT_F_GROUPLIST)) && hl.grp >= 0;
    if (F & (RT_F_BVH | RT_F_GROUPLIST)) row_load<F, false>(S.objs + (in_group ? hl.grp : hl.obj), grp);
    else grp = o;
    const int gc = in_group ? obj_op_count(grp.head) : 0;
    const int n = (F & RT_F_XFORM) ? gc + obj_op_count(o.head) : 0;
    DRay r = world_ray;
    for (int k = 0; k < n; ++k) { const OpV op = chain_op<F>(S, grp, gc, o, k); apply_op(op.type, op.a, op.b, op.c, r); }

    const float t = hl.t;
    h.t = t; h.u = 0.f; h.v = 0.f; h.front = 0;
    h.material = o.material;
    h.prim_id = o.prim_id;
    /* u,v feed image textures only (Texture.cuh:49-61); the render kernel skips them otherwise */
    const bool want_uv = WANT_UV || ((F & RT_F_IMAGE) && S.mats[o.material].pad != 0);
    const int ty = obj_type(o.head);
    if ((F & RT_F_PLANAR) && ty >= RT_OBJ_RECT_XY && ty <= RT_OBJ_BOX) {
        /* the three rect kinds and the six box faces (Box.cuh:24-36) through one body */
        const float* f = o.f;
        const bool box = ty == RT_OBJ_BOX;
        const int face = hl.aux;      /* the face whose test produced t in the search */
        const int KA = box ? 2 - (face >> 1) : (int)RT_OBJ_RECT_YZ - ty;
        const float a0 = box ? (KA == 0 ? f[1] : f[0]) : f[0];
        const float a1 = box ? (KA == 0 ? f[4] : f[3]) : f[1];
        const float b0 = box ? (KA == 2 ? f[1] : f[2]) : f[2];
        const float b1 = box ? (KA == 2 ? f[4] : f[5]) : f[3];
        rect_rec<WANT_UV>(KA, a0, a1, b0, b1, r, t, want_uv, h);
        if (box && (face & 1)) h.front = !h.front;          /* FlipNormals, Surface.cuh:24 */
    } else if ((F & RT_F_SPHERE) && ty == RT_OBJ_SPHERE) {
        V3 centre = ld3(o.f);
        h.p = at(r, t);
        h.n = (h.p - centre) / o.f[3];
        if (want_uv) {
            V3 q = (h.p - centre) / o.f[3];
            float phi = atan2f(q.z, q.x);                       /* getSphereUV Sphere.cuh:64-70 */
            float theta = asinf(q.y);
            h.u = (float)(1 - (phi + RT_PI_D) / (2 * RT_PI_D));
            h.v = (float)((theta + RT_PI_D / 2) / RT_PI_D);
        }
    } else if ((F & RT_F_MSPHERE) && ty == RT_OBJ_MOVING_SPHERE) {
        V3 centre = moving_centre(o.f, r.time);
        h.p = at(r, t);
        h.n = (h.p - centre) / o.f[8];
    } else if ((F & RT_F_TRI) && ty == RT_OBJ_TRIANGLE) {
        V3 normal = ld3(o.f + 9);
        h.p = at(r, t);
        h.n = dot(normal, r.d) > 0 ? -normal : normal;      /* Triangle.cuh:38-41 */
    } else { /* RT_OBJ_MEDIUM, Volumes.cuh:44-46 */
        h.p = at(r, t);
        h.n = mk(1.f, 0.f, 0.f);
        h.front = 1;
    }
    if (obj_flags(o.head) & RT_OBJ_FLAG_FLIP) h.front = !h.front;

    /* unwind the wrappers, innermost first (Transform.cuh:24, :101-108) */
    for (int k = n - 1; k >= 0; --k) {
        const OpV op = chain_op<F>(S, grp, gc, o, k);
        if (op.type == RT_OP_TRANSLATE) {
            h.p = h.p + mk(op.a, op.b, op.c);
        } else {
            /* direction of the ray as this RotateY saw it after rotating: ops 0..k applied */
            V3 d = world_ray.d;
            for (int j = 0; j <= k; ++j) {
                const OpV oj = chain_op<F>(S, grp, gc, o, j);
                if (oj.type == RT_OP_ROTATE_Y) {
                    float dx = oj.b * d.x - oj.a * d.z, dz = oj.a * d.x + oj.b * d.z;
                    d.x = dx; d.z = dz;
                }
            }
            float s = op.a, cs = op.b;
            V3 p = h.p, nn = h.n;
            p.x = cs * h.p.x + s * h.p.z;
            p.z = -s * h.p.x + cs * h.p.z;
            nn.x = cs * h.n.x + s * h.n.z;
            nn.z = -s * h.n.x + cs * h.n.z;
            h.p = p;
            h.front = dot(d, nn) < 0;                        /* setFaceNormal(rotatedR, normal) */
            h.n = h.front ? nn : -nn;
        }
    }
}

/* --------------------------------------------------------------- textures ---- */
/* Perlin gradient noise with Hermite-smoothed trilinear weights and its turbulence sum ("The Next Week"
 * ch. 5: perlin::noise, perlin_interp, perlin::turb); the reference has no noise class, the oracle's
 * noise_turb is the same arithmetic.  `tab` = float gradient[256][3] then three 256-byte permutations. */
__device__ inline float noise_at(const uint8_t* tab, V3 p)
{
    const float* grad = reinterpret_cast<const float*>(tab);
    const uint8_t* perm = tab + 256 * 3 * 4;
    const float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    const float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const float lim = 2147483648.f;
    const int i = (fx >= -lim && fx < lim) ? (int)fx : 0;
    const int j = (fy >= -lim && fy < lim) ? (int)fy : 0;
    const int k = (fz >= -lim && fz < lim) ? (int)fz : 0;
    const float uu = u * u * (3 - 2 * u), vv = v * v * (3 - 2 * v), ww = w * w * (3 - 2 * w);
    float accum = 0;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                const int h = perm[(i + di) & 255] ^ perm[256 + ((j + dj) & 255)] ^ perm[512 + ((k + dk) & 255)];
                const V3 c = mk(grad[3 * h], grad[3 * h + 1], grad[3 * h + 2]);
                const V3 wv = mk(u - di, v - dj, w - dk);
                accum += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) * (dk * ww + (1 - dk) * (1 - ww)) * dot(c, wv);
            }
    return accum;
}
__device__ inline float noise_turb(const uint8_t* tab, V3 p, int depth)
{
    float accum = 0, weight = 1.f;
    for (int i = 0; i < depth; ++i) {
        accum += weight * noise_at(tab, p);
        weight *= 0.5f;
        p = 2.f * p;
    }
    return fabsf(accum);
}

template <uint32_t F>
__device__ __forceinline__ V3 texture_value(const DScene& S, int ti, float u, float v, V3 p)
{
    rt_texture t = S.texs[ti];
    /* CheckerTexture::value Texture.cuh:31-36 (checkers may nest) */
    if (F & RT_F_CHECKER) {
        for (int guard = 0; guard < 8 && t.type == RT_TEX_CHECKER; ++guard) {
            float sines = sinf(10 * p.x) * sinf(10 * p.y) * sinf(10 * p.z);
            t = S.texs[sines < 0 ? t.odd : t.even];
        }
    }
    if ((F & RT_F_IMAGE) && t.type == RT_TEX_IMAGE) {                            /* ImageTexture::value Texture.cuh:49-61 */
        const uint8_t* data = S.image + t.image_offset;
        int i = (int)(u * t.width);
        int j = (int)((1 - v) * t.height - .001f);
        if (i < 0) i = 0;
        if (j < 0) j = 0;
        if (i > t.width - 1) i = t.width - 1;
        if (j > t.height - 1) j = t.height - 1;
        const uint8_t* px = data + 3 * i + 3 * t.width * j;
        return mk((int)px[0] / 255.f, (int)px[1] / 255.f, (int)px[2] / 255.f);
    }
    if ((F & RT_F_NOISE) && t.type == RT_TEX_NOISE) {
        const float m = 0.5f * (1 + sinf(t.colour[0] * p.z + 10 * noise_turb(S.image + t.image_offset, p, t.width)));
        return mk(m, m, m);
    }
    return mk(t.colour[0], t.colour[1], t.colour[2]);
}

/* --------------------------------------------------------------- materials ---- */
__device__ __forceinline__ V3 random_in_unit_sphere(const PathRng& g, uint32_t bounce)   /* Material.cuh:34-40 */
{
    V3 p;
    uint32_t k = 0;
    do {
        rt_u32x4 w = rng_block(g, RT_RNG_SPHERE, bounce, k++);
        p = 2.f * mk(rt_u32_to_uniform(w.x), rt_u32_to_uniform(w.y), rt_u32_to_uniform(w.z)) - mk(1.f, 1.f, 1.f);
    } while (dot(p, p) >= 1.f);
    return p;
}
/* The same sample, drawn by the whole wave.  Each lane that needs a unit-ball point (`need`) owns the
 * candidate sequence k = 0,1,2,... of its own (pixel, sample, bounce); a rejection loop run per lane keeps
 * the wave going for its unluckiest lane (about 7 rounds for a mean of 1.9 candidates) with most lanes idle.
 * Here, once at most half the lanes still need a point, the idle lanes evaluate further candidates of the
 * needy ones: with n lanes in need, lane l evaluates candidate k_owner + l / n of the (l mod n)-th needy lane,
 * and every owner takes the first accepted one, so the result is bit for bit what its own loop would give.
 * Must be called by all 64 lanes of the wave together.  `slots` = 64 ints of LDS private to this wave. */
__device__ __forceinline__ V3 random_in_unit_sphere_wave(const PathRng& g, uint32_t bounce, bool need, int* slots)
{
    const int lane = (int)(threadIdx.x & 63u);
    V3 res = mk(0.f, 0.f, 0.f);
    uint32_t k = 0;
    unsigned long long mask = __ballot(need);
    while (mask) {
        const int n = __popcll(mask);
        if (n > 32) {
            if (need) {
                rt_u32x4 w = rng_block(g, RT_RNG_SPHERE, bounce, k++);
                V3 p = 2.f * mk(rt_u32_to_uniform(w.x), rt_u32_to_uniform(w.y), rt_u32_to_uniform(w.z)) - mk(1.f, 1.f, 1.f);
                if (!(dot(p, p) >= 1.f)) { res = p; need = false; }
            }
        } else {
            int c = 64 / n;
            c = c > 8 ? 8 : c;
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (need) slots[rank] = lane;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int j = lane / n, r = lane - j * n;
            const bool helper = j < c;
            const int owner = helper ? slots[r] : lane;
            const uint32_t o_pixel = (uint32_t)__builtin_amdgcn_ds_bpermute(owner << 2, (int)g.pixel);
            const uint32_t o_sample = (uint32_t)__builtin_amdgcn_ds_bpermute(owner << 2, (int)g.sample);
            const uint32_t o_ctr = (uint32_t)__builtin_amdgcn_ds_bpermute(owner << 2, (int)((RT_RNG_SPHERE << 24) | bounce));
            const uint32_t o_k = (uint32_t)__builtin_amdgcn_ds_bpermute(owner << 2, (int)k);
            rt_u32x4 w = rt_philox4x32_10(o_k + (uint32_t)j, o_ctr, o_pixel, o_sample, g.k0, g.k1);
            V3 p = 2.f * mk(rt_u32_to_uniform(w.x), rt_u32_to_uniform(w.y), rt_u32_to_uniform(w.z)) - mk(1.f, 1.f, 1.f);
            const unsigned long long acc = __ballot(helper && !(dot(p, p) >= 1.f));
            int src = lane;
            bool found = false;
            for (int jj = c - 1; jj >= 0; --jj) {          /* c is wave-uniform; lowest jj wins */
                const int h = rank + jj * n;
                if (need && ((acc >> h) & 1ull)) { src = h; found = true; }
            }
            const float px = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(p.x)));
            const float py = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(p.y)));
            const float pz = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(p.z)));
            if (need) {
                if (found) { res = mk(px, py, pz); need = false; }
                else k += (uint32_t)c;
            }
            __builtin_amdgcn_wave_barrier();
        }
        mask = __ballot(need);
    }
    return res;
}
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - (2.f * dot(v, n)) * n; }   /* Material.cuh:42 */
__device__ __forceinline__ float schlick(float cosine, float refIdx)              /* Material.cuh:12-17 */
{
    float r0 = (1.f - refIdx) / (1.f + refIdx);
    r0 = r0 * r0;
    return r0 + (1.f - r0) * powf((1.f - cosine), 5.f);
}
__device__ __forceinline__ bool refract(V3 v, V3 n, float niOverNt, V3& refracted)   /* Material.cuh:19-30 */
{
    V3 uv = unit(v);
    float dt = dot(uv, n);
    float discrim = 1.f - niOverNt * niOverNt * (1 - dt * dt);
    if (discrim > 0) {
        refracted = niOverNt * (uv - dt * n) - sqrtf(discrim) * n;
        return true;
    }
    return false;
}

/* Material::scatter + emitted (Material.cuh:51-144). Returns false when the path ends. */
/* `ball` = randomInUnitSphere() for this bounce, drawn by the caller (random_in_unit_sphere_wave) when the
 * material is Lambert, Metal or Isotropic */
template <uint32_t F>
__device__ __forceinline__ bool scatter(const DScene& S, const DRay& in, const HitRec& rec, const PathRng& g,
                                        uint32_t bounce, V3 ball, V3& attenuation, DRay& out, V3& emitted)
{
    const rt_material m = S.mats[rec.material];
    emitted = mk(0.f, 0.f, 0.f);
    attenuation = mk(0.f, 0.f, 0.f);
    out.o = rec.p;
    out.d = mk(0.f, 0.f, 1.f);
    out.time = in.time;
    if (m.type == RT_MAT_LAMBERT) {
        V3 target = rec.p + rec.n + ball;
        out.d = target - rec.p;
        attenuation = texture_value<F>(S, m.texture, rec.u, rec.v, rec.p);
        return true;
    }
    if (m.type == RT_MAT_LIGHT) {
        emitted = texture_value<F>(S, m.texture, rec.u, rec.v, rec.p);
        return false;
    }
    if ((F & RT_F_METAL) && m.type == RT_MAT_METAL) {
        V3 reflected = reflect(unit(in.d), rec.n);
        out.d = reflected + m.param * ball;
        attenuation = texture_value<F>(S, m.texture, 0, 0, rec.p);
        return dot(out.d, rec.n) > 0.f;
    }
    if ((F & RT_F_GLASS) && m.type == RT_MAT_GLASS) {
        float refIdx = m.param;
        V3 outwardNorm;
        V3 reflected = reflect(in.d, rec.n);
        float niOverNt, cosine, reflectProb;
        attenuation = mk(1.f, 1.f, 1.f);
        V3 refracted = mk(0.f, 0.f, 0.f);
        if (dot(in.d, rec.n) > 0.f) {
            outwardNorm = -rec.n;
            niOverNt = refIdx;
            cosine = dot(in.d, rec.n) / length(in.d);
            cosine = sqrtf(1.f - refIdx * refIdx * (1 - cosine * cosine));
        } else {
            outwardNorm = rec.n;
            niOverNt = 1.f / refIdx;
            cosine = -dot(in.d, rec.n) / length(in.d);
        }
        if (refract(in.d, outwardNorm, niOverNt, refracted)) reflectProb = schlick(cosine, refIdx);
        else reflectProb = 1.f;
        out.d = rt_u32_to_uniform(rng_block(g, RT_RNG_GLASS, bounce, 0u).x) < reflectProb ? reflected : refracted;
        return true;
    }
    if (F & RT_F_ISO) { /* Isotropic: time resets to 0 (Material.cuh:140) */
        out.d = ball;
        out.time = 0.f;
        attenuation = texture_value<F>(S, m.texture, rec.u, rec.v, rec.p);
        return true;
    }
    return false;
}

/* ------------------------------------------------------------------ camera ---- */
/* The camera block fetched by scalar loads AT THE POINT OF USE.  A render kernel starts a camera ray in most iterations of its
 * loop, for a handful of lanes; read from the kernel arguments, the block's 24 words are loaded once and then held in SGPRs
 * across the whole loop, where the scan program and the nested lane masks want them too: the allocator spilled them to
 * VGPR lanes and fetched them back with two dozen v_readlane in every iteration (each a VALU instruction).  The empty asm
 * makes the address new in every iteration, so the three s_load_dwordx8 stay where the ray is made; they hit the scalar
 * cache and are waited for behind the Philox block that precedes the ray. */
__device__ __forceinline__ rt_camera cam_fetch(const DScene& S)
{
    uintptr_t a = (uintptr_t)S.cam_ptr;
    asm volatile("" : "+s"(a));
    typedef const __attribute__((address_space(4))) rt_u32x8_t* const_cam_ptr;
    static_assert(sizeof(rt_camera) == 3 * sizeof(rt_u32x8_t), "rt_camera is three 32-byte groups");
    const rt_u32x8_t w[3] = {((const_cam_ptr)a)[0], ((const_cam_ptr)a)[1], ((const_cam_ptr)a)[2]};
    rt_camera c;
    __builtin_memcpy(&c, w, sizeof c);
    return c;
}

template <uint32_t F>
__device__ __forceinline__ DRay generate_ray(const rt_camera& c, float s, float t, float xi_time, const PathRng& g)   /* Camera.cuh:40-46 */
{
    /* randomInUnitDisk (Camera.cuh:12-18).  Its result is multiplied by lensRadius, and its draws are
     * addressed blocks of their own, so a pinhole camera skips the loop without changing any other draw. */
    V3 p = mk(0.f, 0.f, 0.f);
    if ((F & RT_F_LENS) && c.lens_radius != 0.f) {
        uint32_t k = 0;
        do {
            rt_u32x4 w = rng_block(g, RT_RNG_DISK, 0u, k++);
            p = 2.0f * mk(rt_u32_to_uniform(w.x), rt_u32_to_uniform(w.y), 0.f) - mk(1.f, 1.f, 0.f);
        } while (dot(p, p) > 1.f);
    }
    V3 rd = c.lens_radius * p;
    V3 offset = rd.x * ld3(c.u) + rd.y * ld3(c.v);
    float time = c.time0 + xi_time * (c.time1 - c.time0);
    DRay r;
    r.o = ld3(c.origin) + offset;
    r.d = ld3(c.lower_left) + s * ld3(c.horizontal) + t * ld3(c.vertical) - ld3(c.origin) - offset;
    r.time = time;
    return r;
}
__device__ __forceinline__ DRay primary_ray(const rt_camera& c, float s, float t)
{
    DRay r;
    V3 offset = mk(0.f, 0.f, 0.f);
    r.o = ld3(c.origin) + offset;
    r.d = ld3(c.lower_left) + s * ld3(c.horizontal) + t * ld3(c.vertical) - ld3(c.origin) - offset;
    r.time = c.time0;
    return r;
}

#endif
