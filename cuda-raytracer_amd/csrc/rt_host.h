/*
 * rt_host.h — host-side pieces shared by the C ABI translation units.
 */
#ifndef RT_HOST_H
#define RT_HOST_H
#include <stdint.h>
#include <string>
#include "../../include/rt_scene.h"

/* Device row of the object table: 128 bytes = two 64-byte halves.
 * The first half is a self-contained "bare" primitive (type, material, id, parameters): one
 * s_load_dwordx16 when the row is wave-uniform (list scan), one cache line per lane in a BVH leaf.
 * The second half holds what only wrapped objects, groups and media need: the first two wrapper ops
 * inline (Translate(RotateY(x)) is the reference's idiom, main.cu:294-295), the rest of the chain by
 * index, and the child / BVH / medium links. */
struct alignas(16) DObj {
    uint32_t head;        /* type | flags<<8 | op_count<<16 | op0 type<<24 | op1 type<<28 */
    int32_t  material;
    int32_t  prim_id;     /* in the BVH leaf table (DScene::leafs): the object's index instead */
    float    f[13];       /* rt_object::f; triangles add the unit normal f[9..11] and plane offset f[12] */
    float    op0[3];      /* outermost wrapper: Translate offset, or RotateY (sin, cos, -) */
    float    op1[3];
    int32_t  op_begin;    /* chains longer than two continue in the ops table at op_begin + 2 */
    int32_t  child_begin, child_count;
    int32_t  bvh_root;    /* child reference of the group's BVH root (valid when RT_DOBJ_HAS_BVH) */
    int32_t  medium_slot;
    int32_t  pad[5];
};
#define RT_DOBJ_HAS_BVH 0x80u     /* in the flags byte of head */

/* Device BVH node, 64 bytes = one cache line: the boxes of BOTH children, so that a visit decides
 * which children to enter, and in which order, from one line (the 32-byte ABI node needs its own
 * line plus both children's to do the same).  A child reference >= 0 is another DNode; < 0 is a leaf:
 * ~ref = first_item << 4 | (count - 1), items being rows of the leaf table in BVH order. */
struct alignas(16) DNode {
    float   lmin[3], lmax[3], rmin[3], rmax[3];
    int32_t left, right;
    int32_t pad[2];
};
/* The world list as a linear program of 32-byte records, read wave-uniformly (one s_load_dwordx8
 * each, the next one prefetched) by the closest-hit scan.  Consecutive objects under the same wrapper chain share one XFORM record; everything
 * else (groups, media, moving spheres, triangles, chains longer than two) is a GENERIC record that
 * sends the scan through the object's row. */
enum { RT_P_END = 0, RT_P_XFORM = 1, RT_P_RECT = 2, RT_P_SPHERE = 3, RT_P_GENERIC = 4, RT_P_BOX = 5 };
struct alignas(32) DProg {
    uint32_t head;   /* kind | XFORM: nops<<8 | op0 type<<16 | op1 type<<20 ; RECT: axis<<8 | face<<16 */
    int32_t  obj;    /* object index (RECT/SPHERE/GENERIC) */
    float    v[6];   /* XFORM: op0 abc, op1 abc; RECT: k a0 a1 b0 b1; SPHERE: cx cy cz r; BOX: p0 xyz, p1 xyz */
};

/* scene feature bits: a render kernel is compiled for a set of them and leaves the rest out */
enum {
    RT_F_BVH = 1u << 0,      /* some group has a BVH */
    RT_F_GROUPLIST = 1u << 1,/* some group is scanned as a list */
    RT_F_MEDIA = 1u << 2,
    RT_F_SPHERE = 1u << 3,
    RT_F_MSPHERE = 1u << 4,
    RT_F_TRI = 1u << 5,
    RT_F_PLANAR = 1u << 6,   /* rects and boxes */
    RT_F_LONGCHAIN = 1u << 7,/* a wrapper chain longer than two ops */
    RT_F_LENS = 1u << 8,     /* camera aperture > 0 */
    RT_F_GLASS = 1u << 9,
    RT_F_METAL = 1u << 10,
    RT_F_ISO = 1u << 11,
    RT_F_CHECKER = 1u << 12,
    RT_F_IMAGE = 1u << 13,
    RT_F_XFORM = 1u << 14,   /* any wrapper op at all */
    RT_F_ALL = (1u << 15) - 1,   /* everything the reference's classes can express */
    RT_F_NOISE = 1u << 15    /* a NoiseTexture (not in the reference; kept out of RT_F_ALL so the kernels
                                for reference scenes do not carry it) */
};

/* 4-wide node for the wavefront intersect kernel, 128 bytes: the boxes of up to four children (binary
 * nodes collapsed by surface area on the host) and their references (same encoding as DNode; an unused
 * slot is RT_REF_NONE with an inverted box).  Traversal there is bound by dependent fetches, and a 4-wide
 * step does the work of two or three binary ones per round trip. */
struct alignas(16) DNode4 {
    float   plane[3][2][4];   /* [axis][0 = min, 1 = max][child]: AXIS-major, so that the four children's near planes along an
                                 axis are ONE 16-byte load whose offset the ray's direction sign picks (rt_wavefront4.h) */
    int32_t ref[4];
    int32_t pad[4];
};

/* 8-wide node with quantised child boxes for the third intersect kernel (rt_wavefront8.h), 80 bytes = five 16-byte lane
 * loads (the four-wide node above needs seven, and a ray visits about 1.6 times as many of those).  A child box is
 * p + q * 2^e per axis with 8-bit q, rounded outwards on the host.  Slot s holds the child that lies towards the octant s
 * of the node (bit k of s set = the upper side along axis k), so that a ray whose direction signs form the octant o
 * (bit k set = negative) meets the slots in roughly front-to-back order when it takes them by ascending s ^ o: no sort.
 * Interior children are the nodes child_base + (number of interior slots below s); the items of leaf children are rows
 * of a leaf table laid out by NODE: row 16 * node + s is the first item of slot s, row 16 * node + 8 + s its second
 * (leaves of one or two items, which is what the host's builder makes; rows of absent items are never read). */
struct alignas(16) DNode8 {
    float    p[3];
    uint32_t e_imask;      /* exponent bytes e0 | e1 << 8 | e2 << 16 (biased by 127, i.e. the float's own exponent field) | interior mask << 24 */
    uint32_t child_base;
    uint32_t leaf_masks;   /* slots that are leaves | slots whose leaf has a second item << 8 */
    uint32_t pad[2];
    uint32_t qlo[3][2];    /* per axis: q of the lower planes of slots 0..3, 4..7, one byte each */
    uint32_t qhi[3][2];
};
/* one row of that leaf table: the first 64 bytes of the object's DObj (head, material, object index, 13 parameters) */
struct alignas(16) DLeaf { uint32_t w[16]; };

#define RT_REF_NONE ((int32_t)0x80000000)
#define RT_MAX_LEAF 16
#define RT_MAX_STACK 64

void rt_set_error(const std::string& msg);

/* Checks every index the kernels will follow (children, ops, materials, textures, image
 * extents, BVH nodes/items, nesting) so that a malformed table is refused on the host
 * instead of faulting the GPU.  bvh_depth receives the deepest BVH found. */
bool rt_validate_scene(const rt_flat_scene* s, std::string& why, int* bvh_depth);

/* rt_object -> DObj; precomputes each triangle's unit normal and plane offset with the
 * expressions of Triangle.cuh:23-24.  bvh_root is left for rt_pack_bvh to fill. */
void rt_pack_objects(const rt_flat_scene* s, DObj* out);

/* 32-byte ABI nodes -> DNodes + the leaf table (first halves of the item objects, in item order);
 * sets bvh_root / RT_DOBJ_HAS_BVH of every group that has a BVH.  Returns false (why) for a leaf
 * larger than RT_MAX_LEAF.  depth receives the deepest DNode chain (= traversal stack entries). */
#include <vector>
bool rt_pack_bvh(const rt_flat_scene* s, std::vector<DObj>& objs, std::vector<DNode>& nodes,
                 std::vector<DObj>& leafs_full, int* depth, std::string& why);

/* the same trees collapsed to 4-wide nodes; each BVH group's root reference goes to DObj::pad[0].
 * depth receives the deepest chain of DNode4s (a traversal needs at most 3 stack entries per level). */
void rt_pack_bvh4(const rt_flat_scene* s, std::vector<DObj>& objs, std::vector<DNode4>& nodes4, int* depth);

/* the same trees collapsed to 8-wide quantised nodes + their leaf table (16 rows per node); each BVH group's root node
 * goes to DObj::pad[1] (always a node, even for a tree that is one leaf).  Returns false when a leaf holds more than two
 * items (a caller's own tree): such scenes keep the four-wide kernels.  depth = the deepest chain of nodes, box_abs = the
 * largest absolute coordinate of any box (the traversal's rounding allowance is scaled by it). */
bool rt_pack_bvh8(const rt_flat_scene* s, std::vector<DObj>& objs, std::vector<DNode8>& nodes8, std::vector<DLeaf>& leafs8, int* depth, float* box_abs);

/* the world list -> program records (terminated by one RT_P_END record) */
/* prog: the world list as scan records.  For the wavefront integrator, which runs the list's plain records when
 * it hands a ray to a lane and then walks the BVH groups one after the other: prog_nobvh = the same program
 * without the records of BVH groups, bvh_groups = those groups' object indices in list order (the world
 * itself when it has a BVH of its own). */
void rt_pack_program(const rt_flat_scene* s, std::vector<DProg>& prog, std::vector<DProg>* prog_nobvh = nullptr,
                     std::vector<int>* bvh_groups = nullptr);

/* RT_F_* bits the scene uses */
uint32_t rt_scene_features(const rt_flat_scene* s);

#endif
