/*
 * ref_common.h — TEST INFRASTRUCTURE.  Shared by ref_geom.cpp and ref_shade.cpp, the two translation units
 * that #include the REFERENCE's own .cuh sources where they lie (-I/root/reference/raytracer; nothing of the
 * reference is copied into this repository).  Only the texture builder is shared: both libraries need the
 * reference's Texture objects (Texture.cuh names no RNG type).  Include after the reference headers.
 */
#ifndef REF_COMMON_H
#define REF_COMMON_H
#include <vector>
#include "../include/rt_scene.h"

/* the reference's Texture object for row `ti` of the flat texture table (Texture.cuh:18,30,45) */
static Texture* ref_build_texture(const rt_flat_scene* s, std::vector<Texture*>& made, int ti)
{
    if (ti < 0) return nullptr;
    if (made[ti]) return made[ti];
    const rt_texture& t = s->textures[ti];
    Texture* out = nullptr;
    switch (t.type) {
    case RT_TEX_CONSTANT: out = new ConstantTexture(Vector3(t.colour[0], t.colour[1], t.colour[2])); break;
    case RT_TEX_CHECKER:  out = new CheckerTexture(ref_build_texture(s, made, t.even), ref_build_texture(s, made, t.odd)); break;
    case RT_TEX_NOISE:    /* the reference has no noise class; nothing pinned reads it */
                          out = new ConstantTexture(Vector3(.5f, .5f, .5f)); break;
    default:              out = new ImageTexture(const_cast<unsigned char*>(s->image_bytes + t.image_offset), t.width, t.height); break;
    }
    made[ti] = out;
    return out;
}
#endif
