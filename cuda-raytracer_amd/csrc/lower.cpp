/*
 * lower.cpp — walk the recorded scene graph once and emit the flat tables.
 *
 * Replaces the reference's createWorld<<<1,1>>> (main.cu:527-539): instead of one
 * GPU thread running `new` for every node and recursively building BvhNode with an
 * in-kernel sort (BvhNode.cuh:60-90), the host flattens wrappers into per-object
 * transform chains, de-duplicates materials/textures into tables and builds a
 * binned-SAH BVH with 32-byte nodes.
 */
#include "lower.h"
#include "philox.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>

/* ------------------------------------------------------------------ arena -- */
/* perlin_generate / perlin_generate_perm of "The Next Week" ch. 5, drawing from the scene's stream.
 * curand_uniform is (0,1], the book's drand48 [0,1): the shuffle target is clamped to i. */
NoiseTexture::NoiseTexture(float sc, curandState* randState) : scale(sc)
{
    curandState own;
    if (!randState) { curand_init(419, 0, 0, &own); randState = &own; }
    for (int i = 0; i < 256; ++i) {
        float x = -1.f + 2.f * curand_uniform(randState);
        float y = -1.f + 2.f * curand_uniform(randState);
        float z = -1.f + 2.f * curand_uniform(randState);
        Vector3 g = unitVector(Vector3(x, y, z));
        for (int a = 0; a < 3; ++a) ranvec[i][a] = g.e[a];
    }
    unsigned char* perms[3] = {permX, permY, permZ};
    for (unsigned char* p : perms) {
        for (int i = 0; i < 256; ++i) p[i] = (unsigned char)i;
        for (int i = 255; i > 0; --i) {
            int target = int(curand_uniform(randState) * (i + 1));
            if (target > i) target = i;
            unsigned char tmp = p[i];
            p[i] = p[target];
            p[target] = tmp;
        }
    }
}

namespace rt {
static thread_local Arena* g_arena = nullptr;
Arena::Arena() : prev_(g_arena) { g_arena = this; }
Arena::~Arena()
{
    for (void* p : nodes_) { static_cast<Node*>(p)->~Node(); std::free(p); }
    for (void* p : blocks_) std::free(p);
    g_arena = prev_;
}
void* Arena::allocate_node(std::size_t n)
{
    void* p = std::malloc(n ? n : 1);
    nodes_.push_back(p);
    return p;
}
void* Arena::allocate(std::size_t n)
{
    void* p = std::malloc(n ? n : 1);
    blocks_.push_back(p);
    return p;
}
Arena* Arena::current() { return g_arena; }
void* Node::operator new(std::size_t n)
{
    /* outside any arena the node simply lives until process exit */
    return g_arena ? g_arena->allocate_node(n) : std::malloc(n ? n : 1);
}
} // namespace rt

/* ------------------------------------------------- scene-construction RNG -- */
void curand_init(unsigned long long seed, unsigned long long subsequence, unsigned long long offset,
                 curandState* state)
{
    state->seed = seed;
    state->subsequence = subsequence;
    state->offset = offset;
}
float curand_uniform(curandState* s)
{
    unsigned long long blk = s->offset >> 2;
    rt_u32x4 r = rt_philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)s->subsequence,
                                  (uint32_t)(s->subsequence >> 32), (uint32_t)s->seed,
                                  (uint32_t)(s->seed >> 32));
    uint32_t w[4] = {r.x, r.y, r.z, r.w};
    float u = rt_u32_to_uniform(w[s->offset & 3]);
    s->offset++;
    return u;
}

namespace rt {

void FlatScene::refresh_view()
{
    view.objects = objects.data();         view.n_objects = (int32_t)objects.size();
    view.ops = ops.data();                 view.n_ops = (int32_t)ops.size();
    view.materials = materials.data();     view.n_materials = (int32_t)materials.size();
    view.textures = textures.data();       view.n_textures = (int32_t)textures.size();
    view.image_bytes = image_bytes.data(); view.n_image_bytes = image_bytes.size();
    view.bvh_nodes = bvh_nodes.data();     view.n_bvh_nodes = (int32_t)bvh_nodes.size();
    view.bvh_items = bvh_items.data();     view.n_bvh_items = (int32_t)bvh_items.size();
}

/* -------------------------------------------------------------- bounds ---- */
struct Box3 {
    float lo[3], hi[3];
    Box3() { for (int i = 0; i < 3; ++i) { lo[i] = FLT_MAX; hi[i] = -FLT_MAX; } }
    void grow(const float p[3]) { for (int i = 0; i < 3; ++i) { lo[i] = std::min(lo[i], p[i]); hi[i] = std::max(hi[i], p[i]); } }
    void grow(const Box3& b) { grow(b.lo); grow(b.hi); }
    float area() const
    {
        float e0 = hi[0] - lo[0], e1 = hi[1] - lo[1], e2 = hi[2] - lo[2];
        return e0 * e1 + e1 * e2 + e2 * e0;
    }
};

static Box3 local_bounds(const rt_object& o, float shutter0, float shutter1)
{
    Box3 b;
    const float* f = o.f;
    switch (o.type) {
    case RT_OBJ_SPHERE: {
        float r = std::fabs(f[3]);
        float lo[3] = {f[0] - r, f[1] - r, f[2] - r}, hi[3] = {f[0] + r, f[1] + r, f[2] + r};
        b.grow(lo); b.grow(hi);
    } break;
    case RT_OBJ_MOVING_SPHERE: {
        /* union of the boxes at both ends of the shutter (the reference inherits
         * Sphere::boundingBox over uninitialised members, Sphere.cuh:73-84) */
        float r = std::fabs(f[8]);
        const float times[4] = {f[6], f[7], shutter0, shutter1};
        for (int k = 0; k < 4; ++k) {
            float s = (times[k] - f[6]) / (f[7] - f[6]);
            float c[3] = {f[0] + s * (f[3] - f[0]), f[1] + s * (f[4] - f[1]), f[2] + s * (f[5] - f[2])};
            float lo[3] = {c[0] - r, c[1] - r, c[2] - r}, hi[3] = {c[0] + r, c[1] + r, c[2] + r};
            b.grow(lo); b.grow(hi);
        }
    } break;
    case RT_OBJ_RECT_XY: {
        float lo[3] = {f[0], f[2], f[4] - 1e-4f}, hi[3] = {f[1], f[3], f[4] + 1e-4f};
        b.grow(lo); b.grow(hi);
    } break;
    case RT_OBJ_RECT_XZ: { /* correct axes; Rectangle.cuh:51 puts z on y */
        float lo[3] = {f[0], f[4] - 1e-4f, f[2]}, hi[3] = {f[1], f[4] + 1e-4f, f[3]};
        b.grow(lo); b.grow(hi);
    } break;
    case RT_OBJ_RECT_YZ: {
        float lo[3] = {f[4] - 1e-4f, f[0], f[2]}, hi[3] = {f[4] + 1e-4f, f[1], f[3]};
        b.grow(lo); b.grow(hi);
    } break;
    case RT_OBJ_BOX:
        b.grow(f); b.grow(f + 3);
        break;
    case RT_OBJ_TRIANGLE: {
        /* Triangle.cuh:24,26 intersect the plane dot(P,n) = -dot(v0,n), i.e. the triangle's image
         * shifted by -2 dot(v0,n) n along its own normal (a sign slip in the reference that parity
         * requires us to keep): bound the region hit() can actually report, not the vertices. */
        float e1[3] = {f[3] - f[0], f[4] - f[1], f[5] - f[2]}, e2[3] = {f[6] - f[0], f[7] - f[1], f[8] - f[2]};
        float c[3] = {e1[1] * e2[2] - e1[2] * e2[1], -(e1[0] * e2[2] - e1[2] * e2[0]), e1[0] * e2[1] - e1[1] * e2[0]};
        float len = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        if (len > 0.f && std::isfinite(len)) {
            float n[3] = {c[0] / len, c[1] / len, c[2] / len};
            float s2 = 2.f * (f[0] * n[0] + f[1] * n[1] + f[2] * n[2]);
            for (int k = 0; k < 3; ++k) {
                float q[3] = {f[3 * k] - s2 * n[0], f[3 * k + 1] - s2 * n[1], f[3 * k + 2] - s2 * n[2]};
                b.grow(q);
            }
            /* the plane offset is rounded in fp32 at the magnitude of the vertices */
            float m = 0.f;
            for (int k = 0; k < 9; ++k) m = std::max(m, std::fabs(f[k]));
            for (int k = 0; k < 3; ++k) { b.lo[k] -= 4e-6f * m; b.hi[k] += 4e-6f * m; }
        } else {
            b.grow(f); /* degenerate: hit() never succeeds (NaN normal) */
        }
    } break;
    default: break;
    }
    return b;
}

/* object-space box -> the space outside wrapper `op` */
static Box3 apply_op(const Box3& in, const rt_xform_op& op)
{
    Box3 out;
    if (op.type == RT_OP_TRANSLATE) {
        float lo[3] = {in.lo[0] + op.a, in.lo[1] + op.b, in.lo[2] + op.c};
        float hi[3] = {in.hi[0] + op.a, in.hi[1] + op.b, in.hi[2] + op.c};
        out.grow(lo); out.grow(hi);
    } else {
        float s = op.a, c = op.b;
        for (int i = 0; i < 8; ++i) {
            float x = (i & 1) ? in.hi[0] : in.lo[0];
            float y = (i & 2) ? in.hi[1] : in.lo[1];
            float z = (i & 4) ? in.hi[2] : in.lo[2];
            float p[3] = {c * x + s * z, y, -s * x + c * z};
            out.grow(p);
        }
    }
    return out;
}

/* keep boxes conservative against the rounding of transformed rays */
static void pad_box(Box3& b)
{
    for (int i = 0; i < 3; ++i) {
        float m = std::max(std::fabs(b.lo[i]), std::fabs(b.hi[i]));
        float pad = 1e-5f * m + 1e-6f;
        if (b.hi[i] - b.lo[i] < 2e-4f) pad += 1e-4f;
        b.lo[i] -= pad;
        b.hi[i] += pad;
    }
}

/* -------------------------------------------------------------- lowering -- */
struct Lowerer {
    FlatScene& S;
    const LowerOptions& opt;
    std::string& err;
    std::map<const Material*, int> mat_ix;
    std::map<const Texture*, int>  tex_ix;
    std::map<const unsigned char*, uint64_t> img_ix;
    int next_prim = 0, n_media = 0;
    float shutter0 = 0.f, shutter1 = 0.f;
    bool ok = true;

    Lowerer(FlatScene& s, const LowerOptions& o, std::string& e) : S(s), opt(o), err(e) {}

    bool fail(const char* msg) { if (ok) err = msg; ok = false; return false; }

    int texture(const Texture* t)
    {
        if (!t) return -1;
        auto it = tex_ix.find(t);
        if (it != tex_ix.end()) return it->second;
        int ix = (int)S.textures.size();
        tex_ix[t] = ix;
        S.textures.push_back(rt_texture());
        rt_texture r;
        std::memset(&r, 0, sizeof r);
        r.even = r.odd = -1;
        switch (t->kind()) {
        case Texture::CONSTANT: {
            const ConstantTexture* c = static_cast<const ConstantTexture*>(t);
            r.type = RT_TEX_CONSTANT;
            for (int i = 0; i < 3; ++i) r.colour[i] = c->colour.e[i];
        } break;
        case Texture::CHECKER: {
            const CheckerTexture* c = static_cast<const CheckerTexture*>(t);
            r.type = RT_TEX_CHECKER;
            r.even = texture(c->even);
            r.odd = texture(c->odd);
        } break;
        case Texture::IMAGE: {
            const ImageTexture* c = static_cast<const ImageTexture*>(t);
            r.type = RT_TEX_IMAGE;
            r.width = c->width;
            r.height = c->height;
            auto im = img_ix.find(c->data);
            if (im == img_ix.end()) {
                uint64_t off = S.image_bytes.size();
                size_t n = (size_t)c->width * (size_t)c->height * 3u;
                if (!c->data || c->width <= 0 || c->height <= 0) { fail("ImageTexture without pixel data"); n = 0; }
                S.image_bytes.insert(S.image_bytes.end(), c->data, c->data + n);
                img_ix[c->data] = off;
                r.image_offset = off;
            } else r.image_offset = im->second;
        } break;
        case Texture::NOISE: {
            /* RT_NOISE_TABLE_BYTES at a 4-byte-aligned offset of the byte blob: gradients, then the three permutations */
            const NoiseTexture* c = static_cast<const NoiseTexture*>(t);
            r.type = RT_TEX_NOISE;
            r.colour[0] = c->scale;
            r.width = RT_NOISE_OCTAVES;
            while (S.image_bytes.size() & 3u) S.image_bytes.push_back(0);
            r.image_offset = S.image_bytes.size();
            const unsigned char* g = reinterpret_cast<const unsigned char*>(&c->ranvec[0][0]);
            S.image_bytes.insert(S.image_bytes.end(), g, g + sizeof c->ranvec);
            S.image_bytes.insert(S.image_bytes.end(), c->permX, c->permX + 256);
            S.image_bytes.insert(S.image_bytes.end(), c->permY, c->permY + 256);
            S.image_bytes.insert(S.image_bytes.end(), c->permZ, c->permZ + 256);
        } break;
        }
        S.textures[ix] = r;
        return ix;
    }

    int material(const Material* m)
    {
        if (!m) return -1;
        auto it = mat_ix.find(m);
        if (it != mat_ix.end()) return it->second;
        rt_material r;
        r.pad = 0;
        r.param = 0.f;
        r.texture = -1;
        switch (m->kind()) {
        case Material::LAMBERT:   r.type = RT_MAT_LAMBERT;   r.texture = texture(static_cast<const Lambert*>(m)->diffuse); break;
        case Material::METAL:     r.type = RT_MAT_METAL;     r.texture = texture(static_cast<const Metal*>(m)->diffuse);
                                  r.param = static_cast<const Metal*>(m)->fuzz; break;
        case Material::GLASS:     r.type = RT_MAT_GLASS;     r.param = static_cast<const Glass*>(m)->refIdx; break;
        case Material::LIGHT:     r.type = RT_MAT_LIGHT;     r.texture = texture(static_cast<const DiffuseLight*>(m)->emit); break;
        case Material::ISOTROPIC: r.type = RT_MAT_ISOTROPIC; r.texture = texture(static_cast<const Isotropic*>(m)->diffuse); break;
        }
        int ix = (int)S.materials.size();
        S.materials.push_back(r);
        mat_ix[m] = ix;
        return ix;
    }

    /* members of a list-like node, with op-free nested lists spliced in */
    void gather(Hitable* h, std::vector<Hitable*>& out)
    {
        if (!h) return;
        if (h->kind() == Hitable::LIST) {
            HitableList* l = static_cast<HitableList*>(h);
            for (int i = 0; i < l->listSize; ++i) gather(l->list[i], out);
        } else out.push_back(h);
    }

    static bool is_surface(Hitable* h)
    {
        while (h && (h->kind() == Hitable::TRANSLATE || h->kind() == Hitable::ROTATE_Y || h->kind() == Hitable::FLIP)) {
            if (h->kind() == Hitable::TRANSLATE) h = static_cast<Translate*>(h)->hitablePtr;
            else if (h->kind() == Hitable::ROTATE_Y) h = static_cast<RotateY*>(h)->hitablePtr;
            else h = static_cast<FlipNormals*>(h)->surfacePtr;
        }
        return h && h->kind() != Hitable::LIST && h->kind() != Hitable::BVH && h->kind() != Hitable::MEDIUM;
    }

    void set_bounds(rt_object& o, Box3 b)
    {
        for (int i = o.op_count - 1; i >= 0; --i) b = apply_op(b, S.ops[o.op_begin + i]);
        for (int i = 0; i < 3; ++i) { o.bmin[i] = b.lo[i]; o.bmax[i] = b.hi[i]; }
    }

    /* fill objects[slot] from node h (h may be a stack of wrappers) */
    void fill(int slot, Hitable* h, int depth)
    {
        rt_object o;
        std::memset(&o, 0, sizeof o);
        o.material = o.prim_id = o.bvh_root = o.medium_slot = -1;
        o.op_begin = (int)S.ops.size();
        while (h && (h->kind() == Hitable::TRANSLATE || h->kind() == Hitable::ROTATE_Y || h->kind() == Hitable::FLIP)) {
            if (h->kind() == Hitable::TRANSLATE) {
                Translate* t = static_cast<Translate*>(h);
                rt_xform_op op = {RT_OP_TRANSLATE, t->offset.e[0], t->offset.e[1], t->offset.e[2]};
                S.ops.push_back(op);
                h = t->hitablePtr;
            } else if (h->kind() == Hitable::ROTATE_Y) {
                RotateY* r = static_cast<RotateY*>(h);
                rt_xform_op op = {RT_OP_ROTATE_Y, r->sinTheta, r->cosTheta, 0.f};
                S.ops.push_back(op);
                h = r->hitablePtr;
            } else {
                o.flags ^= RT_OBJ_FLAG_FLIP;
                h = static_cast<FlipNormals*>(h)->surfacePtr;
            }
        }
        o.op_count = (int)S.ops.size() - o.op_begin;
        if (!h) { fail("null Hitable in scene graph"); return; }
        float* f = o.f;
        switch (h->kind()) {
        case Hitable::SPHERE: {
            Sphere* s = static_cast<Sphere*>(h);
            o.type = RT_OBJ_SPHERE;
            f[0] = s->centre.e[0]; f[1] = s->centre.e[1]; f[2] = s->centre.e[2]; f[3] = s->radius;
            o.material = material(s->matPtr);
        } break;
        case Hitable::MOVING_SPHERE: {
            MovingSphere* s = static_cast<MovingSphere*>(h);
            o.type = RT_OBJ_MOVING_SPHERE;
            for (int i = 0; i < 3; ++i) { f[i] = s->centre0.e[i]; f[3 + i] = s->centre1.e[i]; }
            f[6] = s->time0; f[7] = s->time1; f[8] = s->radius;
            o.material = material(s->matPtr);
        } break;
        case Hitable::TRIANGLE: {
            Triangle* t = static_cast<Triangle*>(h);
            o.type = RT_OBJ_TRIANGLE;
            for (int i = 0; i < 3; ++i) { f[i] = t->v0.e[i]; f[3 + i] = t->v1.e[i]; f[6 + i] = t->v2.e[i]; }
            o.material = material(t->matPtr);
        } break;
        case Hitable::XY_RECT: {
            XYRect* r = static_cast<XYRect*>(h);
            o.type = RT_OBJ_RECT_XY;
            f[0] = r->x0; f[1] = r->x1; f[2] = r->y0; f[3] = r->y1; f[4] = r->k;
            o.material = material(r->matPtr);
        } break;
        case Hitable::XZ_RECT: {
            XZRect* r = static_cast<XZRect*>(h);
            o.type = RT_OBJ_RECT_XZ;
            f[0] = r->x0; f[1] = r->x1; f[2] = r->z0; f[3] = r->z1; f[4] = r->k;
            o.material = material(r->matPtr);
        } break;
        case Hitable::YZ_RECT: {
            YZRect* r = static_cast<YZRect*>(h);
            o.type = RT_OBJ_RECT_YZ;
            f[0] = r->y0; f[1] = r->y1; f[2] = r->z0; f[3] = r->z1; f[4] = r->k;
            o.material = material(r->matPtr);
        } break;
        case Hitable::BOX: {
            Box* b = static_cast<Box*>(h);
            o.type = RT_OBJ_BOX;
            for (int i = 0; i < 3; ++i) { f[i] = b->pMin.e[i]; f[3 + i] = b->pMax.e[i]; }
            o.material = material(b->matPtr);
        } break;
        case Hitable::MEDIUM: {
            ConstantMedium* m = static_cast<ConstantMedium*>(h);
            o.type = RT_OBJ_MEDIUM;
            f[0] = m->negInvdensity;
            o.material = material(m->matPtr);
        } break;
        case Hitable::LIST:
        case Hitable::BVH:
            o.type = RT_OBJ_GROUP;
            break;
        default:
            fail("unsupported node kind");
            return;
        }

        if (o.type == RT_OBJ_GROUP) {
            if (depth >= 2) { fail("groups nested deeper than world -> group are not supported"); return; }
            std::vector<Hitable*> members;
            if (h->kind() == Hitable::LIST) gather(h, members);
            else for (Hitable* m : static_cast<BvhNode*>(h)->members) gather(m, members);
            /* A long world list that also holds groups or media would be scanned linearly.  Closest hit
             * does not depend on list order (except exact ties), so its plain surfaces are moved into one
             * synthetic BVH group placed first; groups and media keep their relative order after it. */
            BvhNode surfaces;
            if (depth == 0) {
                std::vector<Hitable*> rest;
                for (Hitable* m : members) (is_surface(m) ? surfaces.members : rest).push_back(m);
                if (!rest.empty() && (int)surfaces.members.size() >= opt.bvh_min_children) {
                    members.clear();
                    members.push_back(&surfaces);
                    members.insert(members.end(), rest.begin(), rest.end());
                }
            }
            o.child_begin = (int)S.objects.size();
            o.child_count = (int)members.size();
            S.objects.resize(S.objects.size() + members.size());
            S.objects[slot] = o;
            for (size_t i = 0; i < members.size(); ++i) {
                fill(o.child_begin + (int)i, members[i], depth + 1);
                if (!ok) return;
            }
            Box3 b;
            bool all_leaves = true;
            for (int i = 0; i < o.child_count; ++i) {
                const rt_object& c = S.objects[o.child_begin + i];
                Box3 cb; cb.grow(c.bmin); cb.grow(c.bmax);
                b.grow(cb);
                if (c.type == RT_OBJ_GROUP || c.type == RT_OBJ_MEDIUM) all_leaves = false;
            }
            if (o.child_count == 0) { float z[3] = {0, 0, 0}; b.grow(z); }
            if (all_leaves && o.child_count >= opt.bvh_min_children)
                o.bvh_root = build_bvh(S.objects, o.child_begin, o.child_count, opt.bvh_leaf_size, S.bvh_nodes, S.bvh_items);
            /* child bounds are in the group's frame; lift through the group's own chain */
            set_bounds(o, b);
            S.objects[slot] = o;
            return;
        }

        o.prim_id = next_prim++;
        if (o.type == RT_OBJ_MEDIUM) {
            ConstantMedium* m = static_cast<ConstantMedium*>(h);
            o.medium_slot = n_media++;
            o.child_begin = (int)S.objects.size();
            o.child_count = 1;
            S.objects.resize(S.objects.size() + 1);
            S.objects[slot] = o;
            int saved_prim = next_prim;
            fill(o.child_begin, m->boundary, 2);
            if (!ok) return;
            next_prim = saved_prim; /* the boundary copy is never reported */
            rt_object& bo = S.objects[o.child_begin];
            if (bo.type == RT_OBJ_GROUP || bo.type == RT_OBJ_MEDIUM) { fail("ConstantMedium boundary must be a single surface"); return; }
            bo.prim_id = -1;
            Box3 b; b.grow(bo.bmin); b.grow(bo.bmax);
            set_bounds(o, b);
            S.objects[slot] = o;
            return;
        }
        Box3 b = local_bounds(o, shutter0, shutter1);
        set_bounds(o, b);
        S.objects[slot] = o;
    }
};

/* The reference-named members of a Camera from its recorded constructor arguments (called by the constructor,
 * api/rt_scene_api.h).  The values must carry the roundings of the reference's constructor (Camera.cuh:23-39), so the
 * order of the fp32 operations is the reference's: theta = (vfov * pi) / 180 with pi narrowed to float first; half height =
 * tan(theta / 2) in float; basis w = unit(from - at), u = unit(up x w), v = w x u; corner = ((from - (hw * u) * fd) -
 * (hh * v) * fd) - w * fd; spans = ((2 * hw) * u) * fd and ((2 * hh) * v) * fd; lens radius = aperture / 2.  Built with
 * -ffp-contract=off like everything else here. */
void derive_camera(Camera& k)
{
    const float theta = k.vfovDegrees * ((float)M_PI) / 180.f;
    const float hh = std::tan(theta / 2);
    const float hw = k.aspectRatio * hh;
    const float fd = k.focusDistance;
    const Vector3 back = unitVector(k.eye - k.target);
    const Vector3 right = unitVector(cross(k.upHint, back));
    const Vector3 above = cross(back, right);
    const float two_hw = 2.f * hw, two_hh = 2.f * hh;
    for (int a = 0; a < 3; ++a) {
        const float along_u = hw * right.e[a] * fd;
        const float along_v = hh * above.e[a] * fd;
        k.origin.e[a] = k.eye.e[a];
        k.lowerLeftCorner.e[a] = k.eye.e[a] - along_u - along_v - back.e[a] * fd;
        k.horizontal.e[a] = two_hw * right.e[a] * fd;
        k.vertical.e[a] = two_hh * above.e[a] * fd;
    }
    k.u = right; k.v = above; k.w = back;
    k.lensRadius = k.apertureDiameter / 2.f;
    k.time0 = k.shutterOpen;
    k.time1 = k.shutterClose;
}

/* the 96-byte camera block of include/rt_scene.h: the members as derive_camera left them (a scene function may have
 * edited them afterwards, as it could the reference's) */
static void camera_block(const Camera& k, rt_camera& c)
{
    for (int a = 0; a < 3; ++a) {
        c.origin[a] = k.origin.e[a];
        c.lower_left[a] = k.lowerLeftCorner.e[a];
        c.horizontal[a] = k.horizontal.e[a];
        c.vertical[a] = k.vertical.e[a];
        c.u[a] = k.u.e[a];
        c.v[a] = k.v.e[a];
        c.w[a] = k.w.e[a];
    }
    c.lens_radius = k.lensRadius;
    c.time0 = k.time0;
    c.time1 = k.time1;
}

bool lower(Hitable* world, const Camera* cam, const LowerOptions& opt, FlatScene& out, std::string& err)
{
    out = FlatScene();
    Lowerer L(out, opt, err);
    if (!world || !cam) { err = "null world or camera"; return false; }
    L.shutter0 = cam->time0;
    L.shutter1 = cam->time1;
    /* the root is always a GROUP; a bare object becomes a one-element world */
    out.objects.resize(1);
    if (world->kind() == Hitable::LIST || world->kind() == Hitable::BVH) {
        L.fill(0, world, 0);
    } else {
        Hitable* one[1] = {world};
        HitableList tmp(one, 1);
        L.fill(0, &tmp, 0);
    }
    if (!L.ok) return false;

    camera_block(*cam, out.view.camera);
    out.view.world = 0;
    out.view.n_prims = L.next_prim;
    out.view.n_media = L.n_media;
    out.refresh_view();
    return true;
}

/* ------------------------------------------------------------ BVH builder -- */
#ifndef RT_BVH_BINS
#define RT_BVH_BINS 16     /* SAH bins per axis */
#endif
namespace {
struct Item { Box3 box; float cen[3]; int obj; };

struct Builder {
    std::vector<Item>& items;
    std::vector<rt_bvh_node>& nodes;
    std::vector<int32_t>& out_items;
    int leaf_size;
    int item_base; /* offset of this BVH's items in out_items */

    void set_node(int n, const Box3& b, uint32_t left_first, uint32_t count)
    {
        rt_bvh_node& nd = nodes[n];
        for (int i = 0; i < 3; ++i) { nd.bmin[i] = b.lo[i]; nd.bmax[i] = b.hi[i]; }
        nd.left_first = left_first;
        nd.count = count;
    }

    void build(int node, int first, int count)
    {
        Box3 bb, cb;
        for (int i = first; i < first + count; ++i) { bb.grow(items[i].box); cb.grow(items[i].cen); }
        if (count <= leaf_size) { set_node(node, bb, (uint32_t)(item_base + first), (uint32_t)count); return; }

        const int NB = RT_BVH_BINS;
        int best_axis = -1, best_split = -1;
        float best_cost = FLT_MAX;
        for (int ax = 0; ax < 3; ++ax) {
            float lo = cb.lo[ax], ext = cb.hi[ax] - cb.lo[ax];
            if (!(ext > 0.f)) continue;
            Box3 bins[NB];
            int cnt[NB] = {0};
            float scale = NB / ext;
            for (int i = first; i < first + count; ++i) {
                int b = (int)((items[i].cen[ax] - lo) * scale);
                b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                bins[b].grow(items[i].box);
                cnt[b]++;
            }
            float right_area[NB];
            int right_cnt[NB];
            Box3 acc;
            int c = 0;
            for (int b = NB - 1; b > 0; --b) {
                if (cnt[b]) acc.grow(bins[b]);
                c += cnt[b];
                right_area[b] = c ? acc.area() : 0.f;
                right_cnt[b] = c;
            }
            Box3 lacc;
            int lc = 0;
            for (int b = 0; b < NB - 1; ++b) {
                if (cnt[b]) lacc.grow(bins[b]);
                lc += cnt[b];
                if (lc == 0 || right_cnt[b + 1] == 0) continue;
                float cost = lacc.area() * lc + right_area[b + 1] * right_cnt[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = b; }
            }
        }
        int mid;
        if (best_axis < 0) {
            /* all centroids coincide: split the range in half */
            mid = first + count / 2;
        } else {
            float lo = cb.lo[best_axis], scale = NB / (cb.hi[best_axis] - cb.lo[best_axis]);
            int ax = best_axis, sp = best_split;
            auto it = std::partition(items.begin() + first, items.begin() + first + count, [&](const Item& it_) {
                int b = (int)((it_.cen[ax] - lo) * scale);
                b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                return b <= sp;
            });
            mid = (int)(it - items.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        int left = (int)nodes.size();
        nodes.resize(nodes.size() + 2);
        set_node(node, bb, (uint32_t)left, 0);
        build(left, first, mid - first);
        build(left + 1, mid, first + count - mid);
    }
};
} // namespace

int build_bvh(const std::vector<rt_object>& objects, int first, int count, int leaf_size,
              std::vector<rt_bvh_node>& nodes, std::vector<int32_t>& out_items)
{
    std::vector<Item> items((size_t)count);
    for (int i = 0; i < count; ++i) {
        const rt_object& o = objects[first + i];
        Item& it = items[i];
        it.box.grow(o.bmin);
        it.box.grow(o.bmax);
        pad_box(it.box);
        for (int k = 0; k < 3; ++k) it.cen[k] = 0.5f * (it.box.lo[k] + it.box.hi[k]);
        it.obj = first + i;
    }
    int root = (int)nodes.size();
    nodes.resize(nodes.size() + 1);
    Builder B{items, nodes, out_items, leaf_size, (int)out_items.size()};
    B.build(root, 0, count);
    for (int i = 0; i < count; ++i) out_items.push_back(items[i].obj);
    return root;
}

} // namespace rt
