/*
 * ref_geom.cpp — TEST INFRASTRUCTURE.  Host build of the REFERENCE's own geometry and texture headers,
 * with NO stand-in of any kind.
 *
 * The only translation unit of oracle/_ref/libref_geom.so.  It #includes, where they lie under
 * /root/reference/raytracer:
 *     Vector3 Ray Aabb Hitable HitableList Surface Sphere Triangle Rectangle Box Transform Texture
 * None of these names curandState, curand_uniform or any CUDA library, so nothing is bound, shimmed or
 * typedef'ed here: `__host__`/`__device__` come from the image's own <hip/hip_runtime.h>, which defines
 * them empty for a plain C++ compile, and that is all the build needs.  Everything this library returns
 * — hit flags, primitive ids, t, p, normal, u, v, front-face, texture colours, RotateY's sin/cos — is the
 * reference's own arithmetic; it pins SURVEY 8(a) rows A3-A9 and A15 and north_star criterion 1.
 * Material.cuh (A11-A14) needs an RNG type and lives in ref_shade.cpp; Camera.cuh / Volumes.cuh / main.cu
 * cannot be built here at all (see oracle/Makefile).
 *
 * Primitive ids: every leaf is given the dummy Material* (prim_id + 1); the reference carries matPtr
 * through hit() untouched (Hitable.cuh:10-22), so rec.matPtr is the primitive id with no edit to its code.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <thread>
#include <vector>
#include <hip/hip_runtime.h>

#include "Vector3.cuh"
#include "Ray.cuh"
#include "HitableList.cuh"
#include "Sphere.cuh"
#include "Triangle.cuh"
#include "Surface.cuh"
#include "Rectangle.cuh"
#include "Box.cuh"
#include "Transform.cuh"
#include "Texture.cuh"

#include "ref_common.h"
#include "rt_oracle.h" /* rto_hit record layout is shared with the restatement */

namespace {

/* Triangle has no boundingBox() in the reference and is therefore abstract; this adds the
 * missing override and nothing else, so Triangle::hit below is the reference's own code. */
struct TriangleWithBox : public Triangle {
    TriangleWithBox(Vector3 a, Vector3 b, Vector3 c, Material* m) : Triangle(a, b, c, m) {}
    bool boundingBox(float, float, Aabb& box) const override
    {
        box = Aabb(v0, v0);
        return true;
    }
};
/* the range filter SURVEY 8(a) A7 adds on top of the unmodified Triangle::hit (Triangle.cuh:21-46) */
struct RangedTriangle : public Surface {
    TriangleWithBox tri;
    RangedTriangle(Vector3 a, Vector3 b, Vector3 c, Material* m) : tri(a, b, c, m) {}
    bool hit(const Ray& r, float tMin, float tMax, Intersect& rec) const override
    {
        Intersect tmp;
        if (!tri.hit(r, tMin, tMax, tmp)) return false;
        if (!(tmp.t < tMax && tmp.t > tMin)) return false;
        tmp.u = 0.f; tmp.v = 0.f; tmp.bFrontFace = false;
        rec = tmp;
        return true;
    }
    bool boundingBox(float a, float b, Aabb& box) const override { return tri.boundingBox(a, b, box); }
};

struct RefScene {
    const rt_flat_scene* s;
    std::vector<Hitable*> objects; /* built lazily per object index, with wrappers */
};

Hitable* build_object(RefScene& R, int oi)
{
    if (R.objects[oi]) return R.objects[oi];
    const rt_object& o = R.s->objects[oi];
    const float* f = o.f;
    Material* mat = (Material*)(uintptr_t)(o.prim_id + 1);
    Hitable* h = nullptr;
    Surface* surf = nullptr;
    switch (o.type) {
    case RT_OBJ_SPHERE:        h = surf = new Sphere(Vector3(f[0], f[1], f[2]), f[3], mat); break;
    case RT_OBJ_MOVING_SPHERE: h = surf = new MovingSphere(Vector3(f[0], f[1], f[2]), Vector3(f[3], f[4], f[5]), f[6], f[7], f[8], mat); break;
    case RT_OBJ_RECT_XY:       h = surf = new XYRect(f[0], f[1], f[2], f[3], f[4], mat); break;
    case RT_OBJ_RECT_XZ:       h = surf = new XZRect(f[0], f[1], f[2], f[3], f[4], mat); break;
    case RT_OBJ_RECT_YZ:       h = surf = new YZRect(f[0], f[1], f[2], f[3], f[4], mat); break;
    case RT_OBJ_BOX:           h = surf = new Box(Vector3(f[0], f[1], f[2]), Vector3(f[3], f[4], f[5]), mat); break;
    case RT_OBJ_TRIANGLE:      h = surf = new RangedTriangle(Vector3(f[0], f[1], f[2]), Vector3(f[3], f[4], f[5]), Vector3(f[6], f[7], f[8]), mat); break;
    case RT_OBJ_GROUP: {
        /* flat-list semantics are the parity target for groups (SURVEY 7 "Hard parts") */
        std::vector<Hitable*> kids;
        for (int i = 0; i < o.child_count; ++i) {
            Hitable* k = build_object(R, o.child_begin + i);
            if (k) kids.push_back(k);
        }
        Hitable** arr = new Hitable*[kids.size() + 1];
        for (size_t i = 0; i < kids.size(); ++i) arr[i] = kids[i];
        h = new HitableList(arr, (int)kids.size());
    } break;
    default: /* RT_OBJ_MEDIUM: Volumes.cuh is not buildable here */
        return nullptr;
    }
    if ((o.flags & RT_OBJ_FLAG_FLIP) && surf && o.type >= RT_OBJ_RECT_XY && o.type <= RT_OBJ_RECT_YZ) {
        /* FlipNormals reads p->matPtr of the Surface base, which the rect ctors leave unset
         * (they shadow it); it only matters for the pointer it copies, never used in hit() */
        h = new FlipNormals(surf);
    }
    for (int k = o.op_count - 1; k >= 0; --k) {
        const rt_xform_op& op = R.s->ops[o.op_begin + k];
        if (op.type == RT_OP_TRANSLATE) h = new Translate(h, Vector3(op.a, op.b, op.c));
        else {
            RotateY* r = new RotateY(h, 0.f);
            r->sinTheta = op.a; /* the flat scene stores sin/cos; ref_rotate_sincos() checks how */
            r->cosTheta = op.b; /* they are derived from the angle                              */
            h = r;
        }
    }
    R.objects[oi] = h;
    return h;
}

void record(rto_hit* o, bool h, const Intersect& rec)
{
    memset(o, 0, sizeof *o);
    o->hit = h ? 1 : 0;
    o->prim = -1;
    if (!h) return;
    o->prim = (int32_t)((uintptr_t)rec.matPtr) - 1;
    o->t = rec.t; o->u = rec.u; o->v = rec.v;
    for (int k = 0; k < 3; ++k) { o->p[k] = rec.p[k]; o->n[k] = rec.normal[k]; }
    o->front_face = rec.bFrontFace ? 1 : 0;
    o->material = -1;
}

void hit_range(const Hitable* h, const float* rays, float tmin, float tmax, rto_hit* out, int i0, int i1)
{
    for (int i = i0; i < i1; ++i) {
        const float* q = rays + 7 * (size_t)i;
        Ray r(Vector3(q[0], q[1], q[2]), Vector3(q[3], q[4], q[5]), q[6]);
        Intersect rec;
        memset(&rec, 0, sizeof rec);
        bool hit = h && h->hit(r, tmin, tmax, rec);
        record(&out[i], hit, rec);
    }
}

} // namespace

extern "C" {

/* says which build this is; gen_golden.py stores it in every fixture it makes from this library */
const char* ref_library_id(void) { return "libref_geom.so: reference geometry+texture headers, no stand-in"; }

/* closest hit of `object` (with its wrappers; object = scene->world for the whole list).  threads > 1 only
 * splits the rays over host threads (hit() is const); the per-ray arithmetic is the same */
void ref_object_hit_mt(const rt_flat_scene* s, int object, int n, const float* rays, float tmin, float tmax,
                       rto_hit* out, int threads)
{
    RefScene R;
    R.s = s;
    R.objects.assign((size_t)s->n_objects, nullptr);
    Hitable* h = build_object(R, object);
    if (threads < 2 || n < 2 * threads) { hit_range(h, rays, tmin, tmax, out, 0, n); return; }
    std::vector<std::thread> pool;
    for (int k = 0; k < threads; ++k)
        pool.emplace_back(hit_range, h, rays, tmin, tmax, out, (int)((long long)n * k / threads), (int)((long long)n * (k + 1) / threads));
    for (std::thread& t : pool) t.join();
}

void ref_object_hit(const rt_flat_scene* s, int object, int n, const float* rays, float tmin, float tmax,
                    rto_hit* out)
{
    ref_object_hit_mt(s, object, n, rays, tmin, tmax, out, 1);
}

void ref_texture_value(const rt_flat_scene* s, int texture, int n, const float* uvp, float* rgb)
{
    std::vector<Texture*> made((size_t)s->n_textures, nullptr);
    Texture* t = ref_build_texture(s, made, texture);
    for (int i = 0; i < n; ++i) {
        const float* q = uvp + 5 * (size_t)i;
        Vector3 c = t->value(q[0], q[1], Vector3(q[2], q[3], q[4]));
        for (int k = 0; k < 3; ++k) rgb[3 * (size_t)i + k] = c[k];
    }
}

/* how RotateY's constructor turns degrees into sin/cos (Transform.cuh:59-61) */
void ref_rotate_sincos(float degrees, float* sin_out, float* cos_out)
{
    Sphere dummy(Vector3(0, 0, 0), 1.f, nullptr);
    RotateY r(&dummy, degrees);
    *sin_out = r.sinTheta;
    *cos_out = r.cosTheta;
}

} // extern "C"
