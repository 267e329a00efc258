/*
 * rt_scene.h — the flattened scene: plain-old-data tables that cross the C ABI.
 *
 * The reference keeps its world as a heap of virtually-dispatched nodes built
 * inside a kernel (reference raytracer/main.cu:203-524, Hitable.cuh:23-28).
 * Here the same world is a set of type-tagged tables built on the host by the
 * scene-description API (cuda-raytracer_amd/api/) and lowered by
 * cuda-raytracer_amd/csrc/lower.cpp.  Everything is host memory owned by the
 * caller; rt_scene_upload() copies it.
 *
 * Semantics every consumer of these tables must implement are those of the
 * reference's own hit()/scatter()/value() code; each enum value cites it.
 */
#ifndef RT_SCENE_H
#define RT_SCENE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- geometry object kinds ------------------------------------------------ */
enum rt_obj_type {
    RT_OBJ_SPHERE        = 0, /* Sphere.cuh:25-56      f = cx cy cz r                      */
    RT_OBJ_MOVING_SPHERE = 1, /* Sphere.cuh:86-120     f = c0xyz c1xyz t0 t1 r             */
    RT_OBJ_RECT_XY       = 2, /* Rectangle.cuh:24-38   f = a0 a1 b0 b1 k  (a=x b=y, k=z)   */
    RT_OBJ_RECT_XZ       = 3, /* Rectangle.cuh:55-69   f = a0 a1 b0 b1 k  (a=x b=z, k=y)   */
    RT_OBJ_RECT_YZ       = 4, /* Rectangle.cuh:86-100  f = a0 a1 b0 b1 k  (a=y b=z, k=x)   */
    RT_OBJ_BOX           = 5, /* Box.cuh:24-41         f = p0xyz p1xyz (six rects)         */
    RT_OBJ_TRIANGLE      = 6, /* Triangle.cuh:21-46    f = v0xyz v1xyz v2xyz (+range test) */
    RT_OBJ_MEDIUM        = 7, /* Volumes.cuh:26-49     f = -1/density; child = boundary    */
    RT_OBJ_GROUP         = 8  /* HitableList.cuh:17-32 / BvhNode.cuh:100-128: closest hit
                                 over children [child_begin, child_begin+child_count)     */
};

/* FlipNormals (Surface.cuh:14-33) only toggles Intersect::bFrontFace, which no
 * material reads; it lowers to this flag and has no effect on shading. */
#define RT_OBJ_FLAG_FLIP 1u

typedef struct rt_object {
    int32_t  type;        /* enum rt_obj_type */
    uint32_t flags;
    int32_t  material;    /* index into materials; -1 for groups */
    int32_t  prim_id;     /* id reported by closest-hit queries; -1 for groups.
                             Leaves are numbered in scene construction order. */
    int32_t  op_begin;    /* transform chain, OUTERMOST wrapper first */
    int32_t  op_count;
    int32_t  child_begin; /* GROUP: first child object; MEDIUM: boundary object */
    int32_t  child_count;
    int32_t  bvh_root;    /* GROUP: index of its root in bvh_nodes, or -1 = scan children in order */
    int32_t  medium_slot; /* MEDIUM: ordinal among media (keys its RNG stream) */
    float    f[10];
    float    bmin[3];     /* world-space bounds of the object INCLUDING its own transform chain */
    float    bmax[3];     /* (correct unions; Aabb.cuh:46-55 z-bug is not reproduced)           */
} rt_object;              /* 104 bytes */

/* ---- transform ops (Transform.cuh) ---------------------------------------- */
enum rt_op_type {
    RT_OP_TRANSLATE = 0,  /* Transform.cuh:19-28   a,b,c = offset xyz       */
    RT_OP_ROTATE_Y  = 1   /* Transform.cuh:90-112  a = sinTheta, b = cosTheta */
};
typedef struct rt_xform_op {
    int32_t type;
    float   a, b, c;
} rt_xform_op;

/* ---- materials (Material.cuh) ---------------------------------------------- */
enum rt_mat_type {
    RT_MAT_LAMBERT   = 0, /* Material.cuh:51-63    */
    RT_MAT_METAL     = 1, /* Material.cuh:65-82   param = fuzz (already clamped to <=1) */
    RT_MAT_GLASS     = 2, /* Material.cuh:84-119  param = refIdx */
    RT_MAT_LIGHT     = 3, /* Material.cuh:121-131 */
    RT_MAT_ISOTROPIC = 4  /* Material.cuh:133-144 */
};
typedef struct rt_material {
    int32_t type;
    int32_t texture;      /* index into textures, -1 for glass */
    float   param;
    int32_t pad;
} rt_material;

/* ---- textures (Texture.cuh) -------------------------------------------------- */
enum rt_tex_type {
    RT_TEX_CONSTANT = 0,  /* Texture.cuh:13-23 */
    RT_TEX_CHECKER  = 1,  /* Texture.cuh:25-37  even/odd are texture indices */
    RT_TEX_IMAGE    = 2,  /* Texture.cuh:39-61  RGB8, 3 bytes per texel     */
    RT_TEX_NOISE    = 3   /* no counterpart in Texture.cuh: the Perlin marble of "Ray Tracing: The Next Week"
                             ch. 5, (1,1,1) * 0.5 * (1 + sin(scale*p.z + 10*turb(p, octaves)));
                             colour[0] = scale, width = octaves, image_offset = 4-byte-aligned offset of an
                             RT_NOISE_TABLE_BYTES table: float gradient[256][3], then uint8 permX[256],
                             permY[256], permZ[256] */
};
#define RT_NOISE_OCTAVES     7
#define RT_NOISE_TABLE_BYTES (256 * 3 * 4 + 3 * 256)
typedef struct rt_texture {
    int32_t  type;
    float    colour[3];
    int32_t  even, odd;
    int32_t  width, height;
    uint64_t image_offset; /* byte offset into rt_flat_scene::image_bytes */
} rt_texture;

/* ---- camera (Camera.cuh:20-50) ------------------------------------------------ */
typedef struct rt_camera {
    float lower_left[3];
    float horizontal[3];
    float vertical[3];
    float origin[3];
    float u[3], v[3], w[3];
    float lens_radius, time0, time1;
} rt_camera;               /* 96 bytes, same members as the reference class */

/* ---- BVH (host-built; replaces BvhNode.cuh) ---------------------------------- */
/* 32-byte node.  count == 0: interior, children at left_first and left_first+1.
 * count  > 0: leaf over bvh_items[left_first .. left_first+count).            */
typedef struct rt_bvh_node {
    float    bmin[3];
    uint32_t left_first;
    float    bmax[3];
    uint32_t count;
} rt_bvh_node;

typedef struct rt_flat_scene {
    const rt_object*   objects;     int32_t n_objects;
    const rt_xform_op* ops;         int32_t n_ops;
    const rt_material* materials;   int32_t n_materials;
    const rt_texture*  textures;    int32_t n_textures;
    const uint8_t*     image_bytes; uint64_t n_image_bytes;
    const rt_bvh_node* bvh_nodes;   int32_t n_bvh_nodes;
    const int32_t*     bvh_items;   int32_t n_bvh_items;   /* object indices */
    int32_t            world;       /* index of the root GROUP object (the world list) */
    int32_t            n_prims;     /* number of prim_ids handed out */
    int32_t            n_media;
    rt_camera          camera;
} rt_flat_scene;

enum rt_background {
    RT_BG_BLACK = 0,  /* colour()      main.cu:48-72  (emission, black miss)   */
    RT_BG_SKY   = 1   /* colourUnlit() main.cu:73-100 (sky gradient, no emit)  */
};

#ifdef __cplusplus
}
#endif
#endif
