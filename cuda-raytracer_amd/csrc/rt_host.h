/*
 * rt_host.h — host-side pieces shared by the C ABI translation units.
 */
#ifndef RT_HOST_H
#define RT_HOST_H
#include <stdint.h>
#include <string>
#include "../../include/rt_scene.h"

/* device row of the object table: 128 bytes, two cache-line halves */
struct alignas(16) DObj {
    int32_t  type, material, op_begin, op_count;
    int32_t  child_begin, child_count, bvh_root, medium_slot;
    int32_t  prim_id;
    uint32_t flags;
    int32_t  pad0, pad1;
    float    f[16];   /* rt_object::f; triangles add the unit normal f[9..11] and plane offset f[12] */
    float    pad2[4];
};

void rt_set_error(const std::string& msg);

/* Checks every index the kernels will follow (children, ops, materials, textures, image
 * extents, BVH nodes/items, nesting) so that a malformed table is refused on the host
 * instead of faulting the GPU.  bvh_depth receives the deepest BVH found. */
bool rt_validate_scene(const rt_flat_scene* s, std::string& why, int* bvh_depth);

/* rt_object -> DObj; precomputes each triangle's unit normal and plane offset with the
 * expressions of Triangle.cuh:23-24 */
void rt_pack_objects(const rt_flat_scene* s, DObj* out);

#endif
