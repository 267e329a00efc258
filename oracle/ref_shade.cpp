/*
 * ref_shade.cpp — TEST INFRASTRUCTURE.  Host build of the reference's Material.cuh — WITH A STAND-IN, and
 * labelled as one.
 *
 * The only translation unit of oracle/_ref/libref_shade.so.  Material.cuh (Material.cuh:32-40,47,114) names
 * `curandState` and `curand_uniform` without including anything; the CUDA toolkit that defines them is not
 * in this image.  To compile the reference's scatter()/emitted() code at all, the two names are bound below
 * to the image's rocRAND Philox4x32-10 device engine (host-callable).  That binding is a stand-in for a
 * library the image lacks: vectors made by THIS library therefore do NOT count as "the reference compiled
 * here" for the parity pin.  They are kept as a cross-check of the restatement's scatter arithmetic
 * (recorded in DESIGN.md as "pinned through a stand-in"), and every fixture made
 * from this library says so in its `produced_by` field.  Geometry, textures and RotateY come from
 * ref_geom.cpp, which has no stand-in.
 *
 * Compile with clang++ (left-to-right evaluation of Vector3(xi,xi,xi) arguments, the order the
 * restatement documents).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_philox4x32_10.h>
#include <rocrand/rocrand_uniform.h>

/* ---- the stand-in: two names Material.cuh expects from <curand_kernel.h> ---- */
typedef rocrand_state_philox4x32_10 curandState;
static inline float curand_uniform(curandState* s) { return rocrand_uniform(s); }

#include "Vector3.cuh"
#include "Ray.cuh"
#include "Hitable.cuh"
#include "Texture.cuh"
#include "Material.cuh"

#include "ref_common.h"
#include "rt_oracle.h" /* rto_scatter_out record layout is shared with the restatement */

namespace {

Material* build_material(const rt_flat_scene* s, std::vector<Texture*>& texs, int mi)
{
    if (mi < 0) return nullptr;
    const rt_material& m = s->materials[mi];
    switch (m.type) {
    case RT_MAT_LAMBERT:   return new Lambert(ref_build_texture(s, texs, m.texture));
    case RT_MAT_METAL:     return new Metal(ref_build_texture(s, texs, m.texture), m.param);
    case RT_MAT_GLASS:     return new Glass(m.param);
    case RT_MAT_LIGHT:     return new DiffuseLight(ref_build_texture(s, texs, m.texture));
    default:               return new Isotropic(ref_build_texture(s, texs, m.texture));
    }
}

} // namespace

extern "C" {

const char* ref_library_id(void) { return "libref_shade.so: reference Material.cuh with curandState/curand_uniform bound to rocRAND Philox (stand-in)"; }

/* the rocRAND Philox4x32-10 stream (seed, subsequence pixel | sample << 32): NOT reference code, it pins the
 * generator the north_star prescribes against the image's own rocRAND */
void ref_uniforms(uint64_t seed, uint32_t pixel, uint32_t sample, int n, float* out)
{
    curandState st;
    rocrand_init(seed, (unsigned long long)pixel | ((unsigned long long)sample << 32), 0ULL, &st);
    for (int i = 0; i < n; ++i) out[i] = curand_uniform(&st);
}

void ref_scatter(const rt_flat_scene* s, int material, int n, const float* ray_in, const float* rec_in,
                 uint64_t seed, const uint32_t* pixel_sample, rto_scatter_out* out)
{
    std::vector<Texture*> texs((size_t)s->n_textures, nullptr);
    Material* m = build_material(s, texs, material);
    std::vector<float> stream(4096);
    for (int i = 0; i < n; ++i) {
        const float* q = ray_in + 7 * (size_t)i;
        const float* h = rec_in + 9 * (size_t)i;
        Ray r(Vector3(q[0], q[1], q[2]), Vector3(q[3], q[4], q[5]), q[6]);
        Intersect rec;
        memset(&rec, 0, sizeof rec);
        rec.t = h[0]; rec.u = h[1]; rec.v = h[2];
        rec.p = Vector3(h[3], h[4], h[5]);
        rec.normal = Vector3(h[6], h[7], h[8]);
        rec.matPtr = m;
        curandState st;
        unsigned long long subseq = (unsigned long long)pixel_sample[2 * i] | ((unsigned long long)pixel_sample[2 * i + 1] << 32);
        rocrand_init(seed, subseq, 0ULL, &st);
        Vector3 att(0, 0, 0);
        Ray sc(Vector3(0, 0, 0), Vector3(0, 0, 0), 0.f);
        Vector3 em = m->emitted(rec.u, rec.v, rec.p);
        rto_scatter_out* o = &out[i];
        o->ok = m->scatter(r, rec, att, sc, &st) ? 1 : 0;
        for (int k = 0; k < 3; ++k) {
            o->attenuation[k] = att[k];
            o->origin[k] = sc.origin()[k];
            o->direction[k] = sc.direction()[k];
            o->emitted[k] = em[k];
        }
        o->time = sc.time();
        /* how many uniforms did scatter() consume?  locate the next two draws in a fresh stream */
        ref_uniforms(seed, pixel_sample[2 * i], pixel_sample[2 * i + 1], (int)stream.size(), stream.data());
        float n0 = curand_uniform(&st), n1 = curand_uniform(&st);
        o->draws = 0xFFFFFFFFu;
        for (size_t k = 0; k + 1 < stream.size(); ++k)
            if (stream[k] == n0 && stream[k + 1] == n1) { o->draws = (uint32_t)k; break; }
    }
}

/* Metal's fuzz clamp (Material.cuh:71) as the constructor applies it */
float ref_metal_fuzz(float f)
{
    Metal m(nullptr, f);
    return m.fuzz;
}

} // extern "C"
