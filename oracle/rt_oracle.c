/*
 * rt_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A scalar CPU restatement, in plain C, of the reference path tracer's hot path
 * (goopey7/CUDA-RayTracer, raytracer/<name>.cuh + main.cu:48-100,136-149) over the flat
 * scene tables of include/rt_scene.h.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (cuda-raytracer_amd/) never does.
 *
 * Pinning ("Oracle and parity"):
 *   - PINNED by the reference compiled here with no stand-in of any kind
 *     (oracle/ref_geom.cpp -> oracle/_ref/libref_geom.so, built where /root/reference
 *     exists; the Makefile refuses the library if it names an RNG symbol): geometry
 *     (Sphere, MovingSphere, XY/XZ/YZRect, FlipNormals, Box, Triangle, Translate,
 *     RotateY, HitableList) and the textures, through the fixtures in tests/golden/
 *     made by oracle/gen_golden.py.
 *   - CROSS-CHECKED THROUGH A LABELLED STAND-IN, which pins nothing: the materials
 *     (Material.cuh names curandState / curand_uniform from a toolkit the image
 *     lacks; oracle/ref_shade.cpp -> oracle/_ref/libref_shade.so binds the two names
 *     to rocRAND's Philox, and every fixture made from it says so in `produced_by`).
 *   - PARITY UNPINNED: Camera::generateRay (Camera.cuh:40-46), ConstantMedium::hit
 *     (Volumes.cuh:26-49) and the integrator loops colour()/colourUnlit()
 *     (main.cu:48-100): Camera.cuh and Volumes.cuh include <curand_kernel.h> and
 *     main.cu needs the CUDA runtime, neither of which this image has, and the
 *     reference ships no tests or golden outputs.  They are restated line by line
 *     below with the reference line cited.  generateRay and ConstantMedium::hit are
 *     restated a second time in numpy (tests/test_cross_restatement.py) and the two
 *     readings must agree bit for bit: a guard against a typo, not a pin.
 *
 * Floating point: every expression keeps the reference's operand order and its
 * float/double promotions (noted where they matter).  Build with
 * -ffp-contract=off so no FMA is formed.
 *
 * RNG: the reference's curand XORWOW is replaced (north_star) by Philox4x32-10;
 * the stream layout is documented in cuda-raytracer_amd/csrc/philox.h and
 * restated here independently.  Vector3(xi,xi,xi) argument order is unspecified
 * in C++; this restatement draws left to right (x first), as clang does.
 */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/rt_scene.h"
#include "rt_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------- Vector3 ---- */
typedef struct { float e[3]; } V3;

static V3 v3(float a, float b, float c) { V3 r; r.e[0] = a; r.e[1] = b; r.e[2] = c; return r; }
static V3 vadd(V3 a, V3 b) { return v3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }      /* Vector3.cuh:70 */
static V3 vsub(V3 a, V3 b) { return v3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }      /* :75 */
static V3 vmul(V3 a, V3 b) { return v3(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }      /* :80 */
static V3 vscale(float t, V3 v) { return v3(t * v.e[0], t * v.e[1], t * v.e[2]); }                 /* :90,:100 */
static V3 vdivf(V3 v, float t) { return v3(v.e[0] / t, v.e[1] / t, v.e[2] / t); }                  /* :95 true division */
static V3 vneg(V3 v) { return v3(-v.e[0], -v.e[1], -v.e[2]); }
static float vdot(V3 a, V3 b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }      /* :105 */
static V3 vcross(V3 a, V3 b)                                                                       /* :110 */
{
    return v3((a.e[1] * b.e[2] - a.e[2] * b.e[1]),
              (-(a.e[0] * b.e[2] - a.e[2] * b.e[0])),
              (a.e[0] * b.e[1] - a.e[1] * b.e[0]));
}
static float vlen(V3 v) { return sqrtf(v.e[0] * v.e[0] + v.e[1] * v.e[1] + v.e[2] * v.e[2]); }     /* :47 */
static float vsqlen(V3 v) { return v.e[0] * v.e[0] + v.e[1] * v.e[1] + v.e[2] * v.e[2]; }
static V3 vunit(V3 v) { return vdivf(v, vlen(v)); }                                                /* :166 */

typedef struct { V3 A, B; float time; } Ray;                                                       /* Ray.cuh:12-28 */
static V3 ray_at(const Ray* r, float t) { return vadd(r->A, vscale(t, r->B)); }

typedef struct {
    float t, u, v;
    V3 p, normal;
    int material;   /* index, stands for Intersect::matPtr */
    int front_face;
    int prim_id;    /* extra: which leaf object produced the record */
} Hit;

static void set_face_normal(Hit* h, const Ray* r, V3 outward)                                       /* Hitable.cuh:17-21 */
{
    h->front_face = vdot(r->B, outward) < 0;
    h->normal = h->front_face ? outward : vneg(outward);
}

/* --------------------------------------------------------------- Philox ---- */
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static float u32_to_uniform(uint32_t v) { return 2.3283064e-10f + ((float)v * 2.3283064e-10f); } /* (0,1] */

/* Stream layout (same as cuda-raytracer_amd/csrc/philox.h): every consumer of randomness owns whole
 * Philox blocks addressed by what they are for, so no draw depends on how many draws another
 * consumer made:  counter = (c0, purpose<<24 | bounce, pixel, sample), key = seed.
 *   RNG_CAMERA  c0=0        words 0,1 = pixel jitter u,v (main.cu:185-186), word 2 = shutter time (Camera.cuh:44)
 *   RNG_DISK    c0=iter     words 0,1 = one randomInUnitDisk candidate (Camera.cuh:15)
 *   RNG_SPHERE  c0=iter     words 0,1,2 = one randomInUnitSphere candidate, x first (Material.cuh:37)
 *   RNG_GLASS   c0=0        word 0 = the reflect/refract draw (Material.cuh:114)
 *   RNG_MEDIUM  c0=slot     word 0 = the free-flight draw (Volumes.cuh:40)
 * For known-answer tests against the reference's own scatter() the draws can instead be scripted:
 * then every uniform is taken, in the order the reference code asks for them, from a list. */
enum { RNG_CAMERA = 0, RNG_DISK = 1, RNG_SPHERE = 2, RNG_GLASS = 3, RNG_MEDIUM = 4 };
typedef struct {
    uint32_t key[2];
    uint32_t pixel, sample;
    uint32_t bounce;
    const float* script;   /* non-NULL: scripted uniforms */
    uint32_t script_pos;
    rto_stats* st;
} Rng;

static void rng_init(Rng* g, uint64_t seed, uint32_t pixel, uint32_t sample, rto_stats* st)
{
    g->key[0] = (uint32_t)seed; g->key[1] = (uint32_t)(seed >> 32);
    g->pixel = pixel; g->sample = sample; g->bounce = 0; g->script = NULL; g->script_pos = 0; g->st = st;
}
/* n (<=4) uniforms for (purpose, c0) at the current bounce */
static void rng_draw(Rng* g, int purpose, uint32_t c0, int n, float* out)
{
    if (g->script) {
        for (int i = 0; i < n; ++i) out[i] = g->script[g->script_pos++];
        return;
    }
    uint32_t c[4] = {c0, ((uint32_t)purpose << 24) | g->bounce, g->pixel, g->sample}, w[4];
    philox4x32_10(c, g->key, w);
    if (g->st) g->st->philox_blocks++;
    for (int i = 0; i < n; ++i) out[i] = u32_to_uniform(w[i]);
}
static float rng_medium(Rng* g, int slot)           /* the draw of Volumes.cuh:40 */
{
    float x;
    rng_draw(g, RNG_MEDIUM, (uint32_t)slot, 1, &x);
    return x;
}

/* ------------------------------------------------------------- geometry ---- */
typedef struct {
    const rt_flat_scene* s;
    Rng* rng;           /* NULL: media are skipped (pure surface queries) */
    int use_bvh;
    rto_stats* st;
} Ctx;

static int hit_object(const Ctx* c, int oi, const Ray* r, float tMin, float tMax, Hit* rec);

/* Sphere::hit, Sphere.cuh:25-56; getSphereUV :64-70 (phi/theta float, the rest double) */
static int hit_sphere(const rt_object* o, const Ray* r, float tMin, float tMax, Hit* rec)
{
    V3 centre = v3(o->f[0], o->f[1], o->f[2]);
    float radius = o->f[3];
    V3 oc = vsub(r->A, centre);
    float a = vdot(r->B, r->B);
    float b = 2.f * vdot(oc, r->B);
    float cc = vdot(oc, oc) - radius * radius;
    float discrim = b * b - 4.f * a * cc;
    if (discrim > 0) {
        for (int root = 0; root < 2; ++root) {
            float temp = root == 0 ? (-b - sqrtf(discrim)) / (2 * a) : (-b + sqrtf(discrim)) / (2 * a);
            if (temp < tMax && temp > tMin) {
                rec->t = temp;
                rec->p = ray_at(r, rec->t);
                rec->normal = vdivf(vsub(rec->p, centre), radius);
                V3 q = vdivf(vsub(rec->p, centre), radius);
                float phi = atan2f(q.e[2], q.e[0]);
                float theta = asinf(q.e[1]);
                rec->u = (float)(1 - (phi + M_PI) / (2 * M_PI));
                rec->v = (float)((theta + M_PI / 2) / M_PI);
                return 1;
            }
        }
    }
    return 0;
}

/* MovingSphere::hit, Sphere.cuh:86-120 (u,v are left unset by the reference; 0 here) */
static int hit_moving_sphere(const rt_object* o, const Ray* r, float tMin, float tMax, Hit* rec)
{
    V3 c0 = v3(o->f[0], o->f[1], o->f[2]), c1 = v3(o->f[3], o->f[4], o->f[5]);
    float time0 = o->f[6], time1 = o->f[7], radius = o->f[8];
    V3 centre = vadd(c0, vscale((r->time - time0) / (time1 - time0), vsub(c1, c0)));
    V3 oc = vsub(r->A, centre);
    float a = vdot(r->B, r->B);
    float b = 2.f * vdot(oc, r->B);
    float cc = vdot(oc, oc) - radius * radius;
    float discrim = b * b - 4.f * a * cc;
    if (discrim > 0) {
        for (int root = 0; root < 2; ++root) {
            float temp = root == 0 ? (-b - sqrtf(discrim)) / (2 * a) : (-b + sqrtf(discrim)) / (2 * a);
            if (temp < tMax && temp > tMin) {
                rec->t = temp;
                rec->p = ray_at(r, rec->t);
                rec->normal = vdivf(vsub(rec->p, centre), radius);
                rec->u = 0.f; rec->v = 0.f;
                return 1;
            }
        }
    }
    return 0;
}

/* XYRect/XZRect/YZRect::hit, Rectangle.cuh:24-38, :55-69, :86-100.
 * axis k = fixed axis, (a,b) = the two in-plane axes in the reference's order. */
static int hit_rect_axes(const float* f, int ka, int aa, int ba, const Ray* r, float t0, float t1, Hit* rec)
{
    float k = f[4];
    float t = (k - r->A.e[ka]) / r->B.e[ka];
    if (t < t0 || t > t1) return 0;
    float a = r->A.e[aa] + t * r->B.e[aa];
    float b = r->A.e[ba] + t * r->B.e[ba];
    if (a < f[0] || a > f[1] || b < f[2] || b > f[3]) return 0;
    rec->u = (a - f[0]) / (f[1] - f[0]);
    rec->v = (b - f[2]) / (f[3] - f[2]);
    rec->t = t;
    rec->p = ray_at(r, t);
    V3 n = v3(0.f, 0.f, 0.f);
    n.e[ka] = 1.f;
    set_face_normal(rec, r, n);
    return 1;
}
static int hit_rect(int type, const float* f, const Ray* r, float t0, float t1, Hit* rec)
{
    switch (type) {
    case RT_OBJ_RECT_XY: return hit_rect_axes(f, 2, 0, 1, r, t0, t1, rec);
    case RT_OBJ_RECT_XZ: return hit_rect_axes(f, 1, 0, 2, r, t0, t1, rec);
    default:             return hit_rect_axes(f, 0, 1, 2, r, t0, t1, rec);
    }
}

/* Box::Box + Box::hit, Box.cuh:24-41: six rects in a HitableList, in this order;
 * FlipNormals (Surface.cuh:20-28) on faces 1,3,5 only toggles bFrontFace. */
static int hit_box(const rt_object* o, const Ray* r, float t0, float t1, Hit* rec, rto_stats* st)
{
    const float* p0 = o->f;
    const float* p1 = o->f + 3;
    float faces[6][5] = {
        {p0[0], p1[0], p0[1], p1[1], p1[2]},   /* XYRect  z = p1.z        */
        {p0[0], p1[0], p0[1], p1[1], p0[2]},   /* Flip(XYRect z = p0.z)   */
        {p0[0], p1[0], p0[2], p1[2], p1[1]},   /* XZRect  y = p1.y        */
        {p0[0], p1[0], p0[2], p1[2], p0[1]},   /* Flip(XZRect y = p0.y)   */
        {p0[1], p1[1], p0[2], p1[2], p1[0]},   /* YZRect  x = p1.x        */
        {p0[1], p1[1], p0[2], p1[2], p0[0]}};  /* Flip(YZRect x = p0.x)   */
    static const int types[6] = {RT_OBJ_RECT_XY, RT_OBJ_RECT_XY, RT_OBJ_RECT_XZ, RT_OBJ_RECT_XZ, RT_OBJ_RECT_YZ, RT_OBJ_RECT_YZ};
    Hit temp;
    int any = 0;
    float closest = t1;
    for (int i = 0; i < 6; ++i) {            /* HitableList::hit, HitableList.cuh:17-32 */
        if (st) { st->prim_tests++; st->rect_tests++; }
        if (hit_rect(types[i], faces[i], r, t0, closest, &temp)) {
            if (i & 1) temp.front_face = !temp.front_face;
            any = 1;
            closest = temp.t;
            *rec = temp;
        }
    }
    return any;
}

/* Triangle::hit, Triangle.cuh:21-46.  The reference ignores tMin/tMax (:27); as
 * SURVEY 8(a) A7 prescribes, the unmodified test runs first and the range filter
 * tMin < t < tMax is applied to its result. `distance`, a, b, c are doubles that
 * hold float-computed values. */
static int hit_triangle(const rt_object* o, const Ray* r, float tMin, float tMax, Hit* rec)
{
    V3 v0 = v3(o->f[0], o->f[1], o->f[2]), v1 = v3(o->f[3], o->f[4], o->f[5]), v2 = v3(o->f[6], o->f[7], o->f[8]);
    V3 normal = vunit(vcross(vsub(v1, v0), vsub(v2, v0)));
    float d = vdot(vsub(v3(0.f, 0.f, 0.f), v0), normal);
    V3 origin = r->A;
    double distance = (-vdot(origin, normal) + d) / (vdot(r->B, normal));
    if (distance > FLT_EPSILON) {
        V3 inter = ray_at(r, (float)distance);
        double a = vdot(vcross(vsub(v1, v0), vsub(inter, v0)), normal);
        double b = vdot(vcross(vsub(v2, v1), vsub(inter, v1)), normal);
        double c = vdot(vcross(vsub(v0, v2), vsub(inter, v2)), normal);
        if (a > 0 && b > 0 && c > 0) {
            float t = (float)distance;
            if (!(t < tMax && t > tMin)) return 0;       /* added range filter */
            rec->t = t;
            rec->p = ray_at(r, rec->t);
            if (vdot(normal, r->B) > 0) rec->normal = vneg(normal);
            else rec->normal = normal;
            rec->u = 0.f; rec->v = 0.f;
            return 1;
        }
    }
    return 0;
}

/* ConstantMedium::hit, Volumes.cuh:26-49 */
static int hit_medium(const Ctx* c, const rt_object* o, const Ray* r, float tMin, float tMax, Hit* rec)
{
    if (!c->rng) return 0;
    Hit rec1, rec2;
    Ctx inner = *c;
    inner.rng = NULL; /* boundaries are surfaces */
    if (!hit_object(&inner, o->child_begin, r, -FLT_MAX, FLT_MAX, &rec1)) return 0;
    if (!hit_object(&inner, o->child_begin, r, rec1.t + .0001f, FLT_MAX, &rec2)) return 0;
    if (rec1.t < tMin) rec1.t = tMin;
    if (rec2.t > tMax) rec2.t = tMax;
    if (rec1.t >= rec2.t) return 0;
    if (rec1.t < 0) rec1.t = 0;
    const float rayLength = vlen(r->B);
    const float distanceInsideBoundary = (rec2.t - rec1.t) * rayLength;
    const float hitDistance = o->f[0] * logf(rng_medium(c->rng, o->medium_slot));
    if (hitDistance > distanceInsideBoundary) return 0;
    rec->t = rec1.t + hitDistance / rayLength;
    rec->p = ray_at(r, rec->t);
    rec->normal = v3(1, 0, 0);
    rec->front_face = 1;
    rec->u = 0.f; rec->v = 0.f;
    return 1;
}

/* Aabb::hit, Aabb.cuh:31-44 (used only when traversing the host-built BVH) */
static int aabb_hit(const rt_bvh_node* n, const Ray* r, float tMin, float tMax)
{
    for (int a = 0; a < 3; a++) {
        float ta = (n->bmin[a] - r->A.e[a]) / r->B.e[a];
        float tb = (n->bmax[a] - r->A.e[a]) / r->B.e[a];
        float t0 = ta < tb ? ta : tb;
        float t1 = ta > tb ? ta : tb;
        tMin = t0 > tMin ? t0 : tMin;
        tMax = t1 < tMax ? t1 : tMax;
        if (tMax <= tMin) return 0;
    }
    return 1;
}

/* closest hit over a group's children: HitableList::hit, HitableList.cuh:17-32
 * (list order, tMax shrinks, first wins ties).  With use_bvh the same children are
 * reached through the host-built BVH instead (closest hit is order-independent
 * except for exact ties). */
static int hit_group(const Ctx* c, const rt_object* g, const Ray* r, float tMin, float tMax, Hit* rec)
{
    Hit temp;
    int any = 0;
    float closest = tMax;
    if (c->use_bvh && g->bvh_root >= 0) {
        int stack[64], sp = 0;
        stack[sp++] = g->bvh_root;
        while (sp) {
            const rt_bvh_node* n = &c->s->bvh_nodes[stack[--sp]];
            if (c->st) c->st->nodes_visited++;
            if (!aabb_hit(n, r, tMin, closest)) continue;
            if (n->count) {
                for (uint32_t i = 0; i < n->count; ++i)
                    if (hit_object(c, c->s->bvh_items[n->left_first + i], r, tMin, closest, &temp)) {
                        any = 1; closest = temp.t; *rec = temp;
                    }
            } else {
                /* visit the child whose box centre is nearer along the ray first */
                const rt_bvh_node* l = &c->s->bvh_nodes[n->left_first];
                const rt_bvh_node* rr = l + 1;
                float dl = 0.f, dr = 0.f;
                for (int a = 0; a < 3; ++a) {
                    dl += (0.5f * (l->bmin[a] + l->bmax[a]) - r->A.e[a]) * r->B.e[a];
                    dr += (0.5f * (rr->bmin[a] + rr->bmax[a]) - r->A.e[a]) * r->B.e[a];
                }
                if (sp + 2 > 64) return any; /* cannot happen for SAH trees of < 2^31 items */
                if (dl <= dr) { stack[sp++] = (int)n->left_first + 1; stack[sp++] = (int)n->left_first; }
                else          { stack[sp++] = (int)n->left_first;     stack[sp++] = (int)n->left_first + 1; }
            }
        }
        return any;
    }
    for (int i = 0; i < g->child_count; i++) {
        if (hit_object(c, g->child_begin + i, r, tMin, closest, &temp)) {
            any = 1;
            closest = temp.t;
            *rec = temp;
        }
    }
    return any;
}

static int hit_bare(const Ctx* c, int oi, const Ray* r, float tMin, float tMax, Hit* rec)
{
    const rt_object* o = &c->s->objects[oi];
    int h = 0;
    if (c->st && o->type != RT_OBJ_GROUP && o->type != RT_OBJ_BOX) {
        c->st->prim_tests++;
        if (o->type == RT_OBJ_SPHERE || o->type == RT_OBJ_MOVING_SPHERE) c->st->sphere_tests++;
        else if (o->type == RT_OBJ_TRIANGLE) c->st->tri_tests++;
        else if (o->type != RT_OBJ_MEDIUM) c->st->rect_tests++;
    }
    switch (o->type) {
    case RT_OBJ_SPHERE:        h = hit_sphere(o, r, tMin, tMax, rec); break;
    case RT_OBJ_MOVING_SPHERE: h = hit_moving_sphere(o, r, tMin, tMax, rec); break;
    case RT_OBJ_RECT_XY:
    case RT_OBJ_RECT_XZ:
    case RT_OBJ_RECT_YZ:       h = hit_rect(o->type, o->f, r, tMin, tMax, rec); break;
    case RT_OBJ_BOX:           h = hit_box(o, r, tMin, tMax, rec, c->st); break;
    case RT_OBJ_TRIANGLE:      h = hit_triangle(o, r, tMin, tMax, rec); break;
    case RT_OBJ_MEDIUM:        h = hit_medium(c, o, r, tMin, tMax, rec); break;
    case RT_OBJ_GROUP:         return hit_group(c, o, r, tMin, tMax, rec);
    default: return 0;
    }
    if (h) {
        rec->material = o->material;
        rec->prim_id = o->prim_id;
        if ((o->flags & RT_OBJ_FLAG_FLIP) && o->type != RT_OBJ_GROUP) rec->front_face = !rec->front_face;
    }
    return h;
}

/* wrappers, outermost first: Translate::hit Transform.cuh:19-28, RotateY::hit :90-112 */
static int hit_chain(const Ctx* c, int oi, int k, const Ray* r, float tMin, float tMax, Hit* rec)
{
    const rt_object* o = &c->s->objects[oi];
    if (k == o->op_count) return hit_bare(c, oi, r, tMin, tMax, rec);
    const rt_xform_op* op = &c->s->ops[o->op_begin + k];
    if (c->st) { if (op->type == RT_OP_TRANSLATE) c->st->trans_ops++; else c->st->rot_ops++; }
    if (op->type == RT_OP_TRANSLATE) {
        V3 offset = v3(op->a, op->b, op->c);
        Ray moved = {vsub(r->A, offset), r->B, r->time};
        if (hit_chain(c, oi, k + 1, &moved, tMin, tMax, rec)) {
            rec->p = vadd(rec->p, offset);
            return 1;
        }
        return 0;
    } else {
        float sinTheta = op->a, cosTheta = op->b;
        V3 origin = r->A, direction = r->B;
        origin.e[0] = cosTheta * r->A.e[0] - sinTheta * r->A.e[2];
        origin.e[2] = sinTheta * r->A.e[0] + cosTheta * r->A.e[2];
        direction.e[0] = cosTheta * r->B.e[0] - sinTheta * r->B.e[2];
        direction.e[2] = sinTheta * r->B.e[0] + cosTheta * r->B.e[2];
        Ray rotated = {origin, direction, r->time};
        if (hit_chain(c, oi, k + 1, &rotated, tMin, tMax, rec)) {
            V3 p = rec->p, normal = rec->normal;
            p.e[0] = cosTheta * rec->p.e[0] + sinTheta * rec->p.e[2];
            p.e[2] = -sinTheta * rec->p.e[0] + cosTheta * rec->p.e[2];
            normal.e[0] = cosTheta * rec->normal.e[0] + sinTheta * rec->normal.e[2];
            normal.e[2] = -sinTheta * rec->normal.e[0] + cosTheta * rec->normal.e[2];
            rec->p = p;
            set_face_normal(rec, &rotated, normal);   /* rotated ray vs world normal, as the reference */
            return 1;
        }
        return 0;
    }
}

static int hit_object(const Ctx* c, int oi, const Ray* r, float tMin, float tMax, Hit* rec)
{
    return hit_chain(c, oi, 0, r, tMin, tMax, rec);
}

/* ------------------------------------------------------------- textures ---- */
/* RT_TEX_NOISE has no counterpart in the reference (Texture.cuh ends at ImageTexture): this restates
 * perlin::noise / perlin_interp / perlin::turb of "Ray Tracing: The Next Week" ch. 5.  PARITY UNPINNED:
 * no reference code or fixture covers it; tests check the HIP path against this and its own properties. */
static float noise_at(const uint8_t* tab, V3 p)
{
    const float* grad = (const float*)tab;
    const uint8_t* perm = tab + 256 * 3 * 4;
    const float fx = floorf(p.e[0]), fy = floorf(p.e[1]), fz = floorf(p.e[2]);
    const float u = p.e[0] - fx, v = p.e[1] - fy, w = p.e[2] - fz;
    const float lim = 2147483648.f;
    const int i = (fx >= -lim && fx < lim) ? (int)fx : 0;
    const int j = (fy >= -lim && fy < lim) ? (int)fy : 0;
    const int k = (fz >= -lim && fz < lim) ? (int)fz : 0;
    const float uu = u * u * (3 - 2 * u), vv = v * v * (3 - 2 * v), ww = w * w * (3 - 2 * w);
    float accum = 0;
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                const int h = perm[(i + di) & 255] ^ perm[256 + ((j + dj) & 255)] ^ perm[512 + ((k + dk) & 255)];
                const V3 c = v3(grad[3 * h], grad[3 * h + 1], grad[3 * h + 2]);
                const V3 wv = v3(u - di, v - dj, w - dk);
                accum += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) * (dk * ww + (1 - dk) * (1 - ww)) * vdot(c, wv);
            }
    return accum;
}
static float noise_turb(const uint8_t* tab, V3 p, int depth)
{
    float accum = 0, weight = 1.f;
    for (int i = 0; i < depth; ++i) {
        accum += weight * noise_at(tab, p);
        weight *= 0.5f;
        p = vscale(2.f, p);
    }
    return fabsf(accum);
}
float rto_noise_turb(const uint8_t* table, const float p[3], int depth)   /* test hook */
{
    return noise_turb(table, v3(p[0], p[1], p[2]), depth);
}

static V3 texture_value(const rt_flat_scene* s, int ti, float u, float v, V3 p)      /* Texture.cuh */
{
    const rt_texture* t = &s->textures[ti];
    switch (t->type) {
    case RT_TEX_CONSTANT: return v3(t->colour[0], t->colour[1], t->colour[2]);       /* :19-22 */
    case RT_TEX_CHECKER: {                                                            /* :31-36 */
        float sines = sinf(10 * p.e[0]) * sinf(10 * p.e[1]) * sinf(10 * p.e[2]);
        if (sines < 0) return texture_value(s, t->odd, u, v, p);
        else return texture_value(s, t->even, u, v, p);
    }
    case RT_TEX_NOISE: {
        float m = 0.5f * (1 + sinf(t->colour[0] * p.e[2] + 10 * noise_turb(s->image_bytes + t->image_offset, p, t->width)));
        return v3(m, m, m);
    }
    default: {                                                                        /* :49-61 */
        const uint8_t* data = s->image_bytes + t->image_offset;
        int width = t->width, height = t->height;
        int i = (int)(u * width);
        int j = (int)((1 - v) * height - .001f);
        if (i < 0) i = 0;
        if (j < 0) j = 0;
        if (i > width - 1) i = width - 1;
        if (j > height - 1) j = height - 1;
        float r = (int)(data[3 * i + 3 * width * j]) / 255.f;
        float g = (int)(data[3 * i + 3 * width * j + 1]) / 255.f;
        float b = (int)(data[3 * i + 3 * width * j + 2]) / 255.f;
        return v3(r, g, b);
    }
    }
}

/* ------------------------------------------------------------ materials ---- */
static float schlick(float cosine, float refIdx)                                      /* Material.cuh:12-17 */
{
    float r0 = (1.f - refIdx) / (1.f + refIdx);
    r0 = r0 * r0;
    return r0 + (1.f - r0) * powf((1.f - cosine), 5.f);
}
static int refract(V3 v, V3 n, float niOverNt, V3* refracted)                          /* :19-30 */
{
    V3 uv = vunit(v);
    float dt = vdot(uv, n);
    float discrim = 1.f - niOverNt * niOverNt * (1 - dt * dt);
    if (discrim > 0) {
        *refracted = vsub(vscale(niOverNt, vsub(uv, vscale(dt, n))), vscale(sqrtf(discrim), n));
        return 1;
    }
    return 0;
}
static V3 random_in_unit_sphere(Rng* g)                                                /* :34-40 */
{
    V3 p;
    uint32_t k = 0;
    do {
        float x[3];
        rng_draw(g, RNG_SPHERE, k++, 3, x);
        p = vsub(vscale(2.f, v3(x[0], x[1], x[2])), v3(1.f, 1.f, 1.f));
        if (g->st) g->st->sphere_iters++;
    } while (vsqlen(p) >= 1.f);
    return p;
}
static V3 reflect(V3 v, V3 n) { return vsub(v, vscale(2.f * vdot(v, n), n)); }          /* :42 */

static V3 emitted(const rt_flat_scene* s, int mi, float u, float v, V3 p)               /* :48, :127-130 */
{
    const rt_material* m = &s->materials[mi];
    if (m->type == RT_MAT_LIGHT) return texture_value(s, m->texture, u, v, p);
    return v3(0, 0, 0);
}

static int scatter(const rt_flat_scene* s, int mi, const Ray* rIn, const Hit* rec, V3* attenuation,
                   Ray* scattered, Rng* g)
{
    const rt_material* m = &s->materials[mi];
    switch (m->type) {
    case RT_MAT_LAMBERT: {                                                              /* :56-62 */
        V3 target = vadd(vadd(rec->p, rec->normal), random_in_unit_sphere(g));
        scattered->A = rec->p; scattered->B = vsub(target, rec->p); scattered->time = rIn->time;
        *attenuation = texture_value(s, m->texture, rec->u, rec->v, rec->p);
        return 1;
    }
    case RT_MAT_METAL: {                                                                /* :73-79 */
        V3 reflected = reflect(vunit(rIn->B), rec->normal);
        scattered->A = rec->p;
        scattered->B = vadd(reflected, vscale(m->param, random_in_unit_sphere(g)));
        scattered->time = rIn->time;
        *attenuation = texture_value(s, m->texture, 0, 0, rec->p);
        return (vdot(scattered->B, rec->normal) > 0.f);
    }
    case RT_MAT_GLASS: {                                                                /* :89-118 */
        float refIdx = m->param;
        V3 outwardNorm;
        V3 reflected = reflect(rIn->B, rec->normal);
        float niOverNt;
        *attenuation = v3(1.f, 1.f, 1.f);
        V3 refracted = v3(0, 0, 0);
        float reflectProb;
        float cosine;
        if (vdot(rIn->B, rec->normal) > 0.f) {
            outwardNorm = vneg(rec->normal);
            niOverNt = refIdx;
            cosine = vdot(rIn->B, rec->normal) / vlen(rIn->B);
            cosine = sqrtf(1.f - refIdx * refIdx * (1 - cosine * cosine));
        } else {
            outwardNorm = rec->normal;
            niOverNt = 1.f / refIdx;
            cosine = -vdot(rIn->B, rec->normal) / vlen(rIn->B);
        }
        if (refract(rIn->B, outwardNorm, niOverNt, &refracted)) reflectProb = schlick(cosine, refIdx);
        else reflectProb = 1.f;
        scattered->A = rec->p; scattered->time = rIn->time;
        float xi;
        rng_draw(g, RNG_GLASS, 0, 1, &xi);
        if (xi < reflectProb) scattered->B = reflected;
        else scattered->B = refracted;
        return 1;
    }
    case RT_MAT_LIGHT:                                                                  /* :126 */
        return 0;
    default: {                                                                          /* Isotropic :138-143 */
        scattered->A = rec->p; scattered->B = random_in_unit_sphere(g); scattered->time = 0.f;
        *attenuation = texture_value(s, m->texture, rec->u, rec->v, rec->p);
        return 1;
    }
    }
}

/* --------------------------------------------------------------- camera ---- */
static V3 ld3(const float* f) { return v3(f[0], f[1], f[2]); }

static V3 random_in_unit_disk(Rng* g)                                                  /* Camera.cuh:12-18 */
{
    V3 p;
    uint32_t k = 0;
    do {
        float x[2];
        rng_draw(g, RNG_DISK, k++, 2, x);
        p = vsub(vscale(2.0f, v3(x[0], x[1], 0)), v3(1.f, 1.f, 0.f));
    } while (vdot(p, p) > 1.f);
    return p;
}
static Ray generate_ray(const rt_camera* c, float s, float t, float xi_time, Rng* g)     /* Camera.cuh:40-46 */
{
    V3 rd = vscale(c->lens_radius, random_in_unit_disk(g));
    V3 offset = vadd(vscale(rd.e[0], ld3(c->u)), vscale(rd.e[1], ld3(c->v)));
    float time = c->time0 + xi_time * (c->time1 - c->time0);
    Ray r;
    r.A = vadd(ld3(c->origin), offset);
    r.B = vsub(vsub(vadd(vadd(ld3(c->lower_left), vscale(s, ld3(c->horizontal))), vscale(t, ld3(c->vertical))), ld3(c->origin)), offset);
    r.time = time;
    return r;
}

/* ----------------------------------------------------------- integrator ---- */
/* colour() main.cu:48-72 (background 0) and colourUnlit() main.cu:73-100 (background 1) */
static V3 colour(const Ctx* c, Ray r, int depth, int background, Rng* g, uint64_t* n_rays)
{
    Ray currentRay = r;
    V3 currentAttenuation = v3(1.f, 1.f, 1.f);
    V3 currentEmitted = v3(0.f, 0.f, 0.f);
    for (int i = 0; i < depth; i++) {
        Hit rec;
        g->bounce = (uint32_t)i;
        (*n_rays)++;
        if (hit_object(c, c->s->world, &currentRay, .001f, FLT_MAX, &rec)) {
            Ray scattered;
            V3 attenuation = v3(0, 0, 0);
            if (background == RT_BG_BLACK) {
                V3 em = emitted(c->s, rec.material, rec.u, rec.v, rec.p);
                if (scatter(c->s, rec.material, &currentRay, &rec, &attenuation, &scattered, g)) {
                    currentAttenuation = vmul(currentAttenuation, attenuation);
                    currentEmitted = vadd(currentEmitted, vmul(em, currentAttenuation));
                    currentRay = scattered;
                } else return vadd(currentEmitted, vmul(em, currentAttenuation));
            } else {
                if (scatter(c->s, rec.material, &currentRay, &rec, &attenuation, &scattered, g)) {
                    currentAttenuation = vmul(currentAttenuation, attenuation);
                    currentRay = scattered;
                } else return v3(0.f, 0.f, 0.f);
            }
        } else {
            if (background == RT_BG_BLACK) return currentEmitted;
            V3 unitDir = vunit(currentRay.B);
            float t = .5f * (unitDir.e[1] + 1.f);
            V3 col = vadd(vscale((1.f - t), v3(1.f, 1.f, 1.f)), vscale(t, v3(.5f, .7f, 1.f)));
            return vmul(currentAttenuation, col);
        }
    }
    return background == RT_BG_BLACK ? currentEmitted : v3(0.f, 0.f, 0.f);
}

/* ------------------------------------------------------------ public API ---- */
static void fill_out(rto_hit* o, int h, const Hit* rec)
{
    memset(o, 0, sizeof *o);
    o->hit = h;
    o->prim = -1;
    if (!h) return;
    o->prim = rec->prim_id;
    o->t = rec->t; o->u = rec->u; o->v = rec->v;
    for (int k = 0; k < 3; ++k) { o->p[k] = rec->p.e[k]; o->n[k] = rec->normal.e[k]; }
    o->front_face = rec->front_face;
    o->material = rec->material;
}

void rto_trace_rays(const rt_flat_scene* s, int n, const float* rays, float tmin, float tmax,
                    int use_bvh, rto_hit* out, rto_stats* st)
{
    Ctx c = {s, NULL, use_bvh, st};
    for (int i = 0; i < n; ++i) {
        const float* q = rays + 7 * (size_t)i;
        Ray r = {v3(q[0], q[1], q[2]), v3(q[3], q[4], q[5]), q[6]};
        Hit rec;
        memset(&rec, 0, sizeof rec);
        int h = hit_object(&c, s->world, &r, tmin, tmax, &rec);
        if (st) st->rays++;
        fill_out(&out[i], h, &rec);
    }
}

void rto_object_hit(const rt_flat_scene* s, int object, int n, const float* rays, float tmin, float tmax,
                    rto_hit* out)
{
    Ctx c = {s, NULL, 0, NULL};
    for (int i = 0; i < n; ++i) {
        const float* q = rays + 7 * (size_t)i;
        Ray r = {v3(q[0], q[1], q[2]), v3(q[3], q[4], q[5]), q[6]};
        Hit rec;
        memset(&rec, 0, sizeof rec);
        int h = hit_object(&c, object, &r, tmin, tmax, &rec);
        fill_out(&out[i], h, &rec);
    }
}

/* one object's hit() with the path tracer's ADDRESSED draws for (seed, pixel, sample, bounce): unlike rto_object_hit
 * a ConstantMedium takes its free-flight draw (Volumes.cuh:40).  Test hook of tests/test_cross_restatement.py. */
void rto_object_hit_addressed(const rt_flat_scene* s, int object, int n, const float* rays, float tmin, float tmax,
                              uint64_t seed, const uint32_t* pixel_sample_bounce, rto_hit* out)
{
    for (int i = 0; i < n; ++i) {
        const float* q = rays + 7 * (size_t)i;
        Ray r = {v3(q[0], q[1], q[2]), v3(q[3], q[4], q[5]), q[6]};
        Rng g;
        rng_init(&g, seed, pixel_sample_bounce[3 * i], pixel_sample_bounce[3 * i + 1], NULL);
        g.bounce = pixel_sample_bounce[3 * i + 2];
        Ctx c = {s, &g, 0, NULL};
        Hit rec;
        memset(&rec, 0, sizeof rec);
        int h = hit_object(&c, object, &r, tmin, tmax, &rec);
        fill_out(&out[i], h, &rec);
    }
}

/* the deterministic pinhole-centre primary ray of pixel (i,j): generateRay with the
 * lens offset forced to 0 and time = time0 (no draws) */
static Ray primary_ray(const rt_camera* c, float s, float t)
{
    Ray r;
    V3 offset = v3(0, 0, 0);
    r.A = vadd(ld3(c->origin), offset);
    r.B = vsub(vsub(vadd(vadd(ld3(c->lower_left), vscale(s, ld3(c->horizontal))), vscale(t, ld3(c->vertical))), ld3(c->origin)), offset);
    r.time = c->time0;
    return r;
}

void rto_trace_primary(const rt_flat_scene* s, int W, int H, float s_off, float t_off, int use_bvh,
                       int32_t* prim, float* t_out)
{
    Ctx c = {s, NULL, use_bvh, NULL};
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            float u = (float)(i + s_off) / (float)W;
            float v = (float)(j + t_off) / (float)H;
            Ray r = primary_ray(&s->camera, u, v);
            Hit rec;
            int h = hit_object(&c, s->world, &r, .001f, FLT_MAX, &rec);
            prim[(size_t)j * W + i] = h ? rec.prim_id : -1;
            t_out[(size_t)j * W + i] = h ? rec.t : 0.f;
        }
}

void rto_texture_value(const rt_flat_scene* s, int texture, int n, const float* uvp, float* rgb)
{
    for (int i = 0; i < n; ++i) {
        const float* q = uvp + 5 * (size_t)i;
        V3 c = texture_value(s, texture, q[0], q[1], v3(q[2], q[3], q[4]));
        for (int k = 0; k < 3; ++k) rgb[3 * (size_t)i + k] = c.e[k];
    }
}

void rto_scatter(const rt_flat_scene* s, int material, int n, const float* ray_in, const float* rec_in,
                 const float* xi, int xi_stride, rto_scatter_out* out)
{
    for (int i = 0; i < n; ++i) {
        const float* q = ray_in + 7 * (size_t)i;
        const float* h = rec_in + 9 * (size_t)i; /* t u v p[3] n[3] */
        Ray r = {v3(q[0], q[1], q[2]), v3(q[3], q[4], q[5]), q[6]};
        Hit rec;
        memset(&rec, 0, sizeof rec);
        rec.t = h[0]; rec.u = h[1]; rec.v = h[2];
        rec.p = v3(h[3], h[4], h[5]);
        rec.normal = v3(h[6], h[7], h[8]);
        rec.material = material;
        Rng g;
        rng_init(&g, 0, 0, 0, NULL);
        g.script = xi + (size_t)i * xi_stride;   /* uniforms in the order the reference consumes them */
        V3 att = v3(0, 0, 0);
        Ray sc = {v3(0, 0, 0), v3(0, 0, 0), 0.f};
        V3 em = emitted(s, material, rec.u, rec.v, rec.p);
        rto_scatter_out* o = &out[i];
        o->ok = scatter(s, material, &r, &rec, &att, &sc, &g);
        for (int k = 0; k < 3; ++k) {
            o->attenuation[k] = att.e[k];
            o->origin[k] = sc.A.e[k];
            o->direction[k] = sc.B.e[k];
            o->emitted[k] = em.e[k];
        }
        o->time = sc.time;
        o->draws = g.script_pos;
    }
}

/* scatter() + emitted() with the path tracer's own ADDRESSED draws (the blocks the render loop would use for this
 * (seed, pixel, sample, bounce)): what the GPU parity hook rt_shade_rays is compared with */
void rto_scatter_addressed(const rt_flat_scene* s, int n, const float* ray_in, const float* rec_in, const int32_t* material,
                           uint64_t seed, const uint32_t* pixel_sample_bounce, rto_scatter_out* out)
{
    for (int i = 0; i < n; ++i) {
        const float* q = ray_in + 7 * (size_t)i;
        const float* h = rec_in + 9 * (size_t)i; /* t u v p[3] n[3] */
        Ray r = {v3(q[0], q[1], q[2]), v3(q[3], q[4], q[5]), q[6]};
        Hit rec;
        memset(&rec, 0, sizeof rec);
        rec.t = h[0]; rec.u = h[1]; rec.v = h[2];
        rec.p = v3(h[3], h[4], h[5]);
        rec.normal = v3(h[6], h[7], h[8]);
        rec.material = material[i];
        Rng g;
        rng_init(&g, seed, pixel_sample_bounce[3 * i], pixel_sample_bounce[3 * i + 1], NULL);
        g.bounce = pixel_sample_bounce[3 * i + 2];
        V3 att = v3(0, 0, 0);
        Ray sc = {v3(0, 0, 0), v3(0, 0, 0), 0.f};
        V3 em = emitted(s, material[i], rec.u, rec.v, rec.p);
        rto_scatter_out* o = &out[i];
        o->ok = scatter(s, material[i], &r, &rec, &att, &sc, &g);
        for (int k = 0; k < 3; ++k) {
            o->attenuation[k] = att.e[k];
            o->origin[k] = sc.A.e[k];
            o->direction[k] = sc.B.e[k];
            o->emitted[k] = em.e[k];
        }
        o->time = sc.time;
        o->draws = 0;
    }
}

void rto_camera_rays(const rt_flat_scene* s, int n, const float* st, uint64_t seed,
                     const uint32_t* pixel_sample, float* rays_out)
{
    for (int i = 0; i < n; ++i) {
        Rng g;
        float cam[3];
        rng_init(&g, seed, pixel_sample[2 * i], pixel_sample[2 * i + 1], NULL);
        rng_draw(&g, RNG_CAMERA, 0, 3, cam);
        Ray r = generate_ray(&s->camera, st[2 * i], st[2 * i + 1], cam[2], &g);
        float* q = rays_out + 7 * (size_t)i;
        for (int k = 0; k < 3; ++k) { q[k] = r.A.e[k]; q[3 + k] = r.B.e[k]; }
        q[6] = r.time;
    }
}

/* n uniforms of the rocRAND-compatible sequential stream (seed, subsequence = pixel | sample<<32,
 * offset 0): word i&3 of block (i>>2, 0, pixel, sample) */
void rto_uniforms(uint64_t seed, uint32_t pixel, uint32_t sample, int n, float* out)
{
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, w[4];
    for (int i = 0; i < n; ++i) {
        if ((i & 3) == 0) {
            uint32_t c[4] = {(uint32_t)i >> 2, 0u, pixel, sample};
            philox4x32_10(c, key, w);
        }
        out[i] = u32_to_uniform(w[i & 3]);
    }
}

void rto_philox_block(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    philox4x32_10(ctr, key, out);
}

/* render main.cu:136-141 for every pixel of rows [row0,row1) stepping by row_step, samples
 * [s_begin,s_end): accum[3*(j*W+i)+c] += sum over samples of colour() (linear, no sqrt). */
typedef struct {
    const rt_flat_scene* s;
    int W, H, s_begin, s_end, max_depth, background, use_bvh;
    uint64_t seed;
    float* accum;
    int row0, row_step;
    rto_stats st;
} Job;

static void* render_rows(void* arg)
{
    Job* jb = (Job*)arg;
    Ctx c = {jb->s, NULL, jb->use_bvh, &jb->st};
    for (int j = jb->row0; j < jb->H; j += jb->row_step)
        for (int i = 0; i < jb->W; ++i) {
            uint32_t pixel = (uint32_t)(j * jb->W + i);
            V3 sum = v3(0, 0, 0);
            for (int sidx = jb->s_begin; sidx < jb->s_end; ++sidx) {
                Rng g;
                float cam[3];
                rng_init(&g, jb->seed, pixel, (uint32_t)sidx, &jb->st);
                c.rng = &g;
                rng_draw(&g, RNG_CAMERA, 0, 3, cam);
                float u = (float)(i + cam[0]) / (float)jb->W;           /* main.cu:185 */
                float v = (float)(j + cam[1]) / (float)jb->H;           /* main.cu:186 */
                Ray r = generate_ray(&jb->s->camera, u, v, cam[2], &g);
                V3 col = colour(&c, r, jb->max_depth, jb->background, &g, &jb->st.rays);
                sum = vadd(sum, col);                                   /* outCol += ... main.cu:188 */
                jb->st.samples++;
            }
            float* a = jb->accum + 3 * (size_t)pixel;
            for (int k = 0; k < 3; ++k) a[k] += sum.e[k];
        }
    return NULL;
}

void rto_render(const rt_flat_scene* s, int W, int H, int s_begin, int s_end, uint64_t seed, int max_depth,
                int background, int use_bvh, int n_threads, float* accum, rto_stats* st_out)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    Job* jobs = (Job*)calloc((size_t)n_threads, sizeof(Job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int k = 0; k < n_threads; ++k) {
        Job* jb = &jobs[k];
        jb->s = s; jb->W = W; jb->H = H; jb->s_begin = s_begin; jb->s_end = s_end;
        jb->max_depth = max_depth; jb->background = background; jb->use_bvh = use_bvh;
        jb->seed = seed; jb->accum = accum; jb->row0 = k; jb->row_step = n_threads;
        if (n_threads == 1) render_rows(jb);
        else pthread_create(&th[k], NULL, render_rows, jb);
    }
    rto_stats tot;
    memset(&tot, 0, sizeof tot);
    for (int k = 0; k < n_threads; ++k) {
        if (n_threads > 1) pthread_join(th[k], NULL);
        tot.rays += jobs[k].st.rays; tot.samples += jobs[k].st.samples;
        tot.prim_tests += jobs[k].st.prim_tests; tot.nodes_visited += jobs[k].st.nodes_visited;
        tot.philox_blocks += jobs[k].st.philox_blocks; tot.sphere_iters += jobs[k].st.sphere_iters;
        tot.rect_tests += jobs[k].st.rect_tests; tot.sphere_tests += jobs[k].st.sphere_tests;
        tot.tri_tests += jobs[k].st.tri_tests; tot.rot_ops += jobs[k].st.rot_ops; tot.trans_ops += jobs[k].st.trans_ops;
    }
    if (st_out) *st_out = tot;
    free(jobs);
    free(th);
}

/* resolve main.cu:192-196: mean over samples, then sqrtf per channel */
void rto_resolve(const float* accum, int n_pixels, int spp, int gamma, float* rgb)
{
    for (size_t i = 0; i < (size_t)n_pixels * 3; ++i) {
        float k = 1.0f / (float)spp;        /* outCol /= float(numSamples): Vector3.cuh:157-164 */
        float c = accum[i] * k;
        rgb[i] = gamma ? sqrtf(c) : c;
    }
}
