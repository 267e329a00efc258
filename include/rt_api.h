/*
 * rt_api.h — C ABI of librt_amd.so, the MI355X path tracer.
 *
 * The reference has no library boundary: its host side is one function,
 *     Vector3* gpuRender(int width,int height,int numSamples,int tx,int ty,int gpuId)
 * (reference raytracer/main.cu:553-620), which allocates, builds the world in a
 * <<<1,1>>> kernel, renders and frees.  The entry points below are what a binding for that
 * path calls instead; each one names the piece of gpuRender / main.cu it replaces.
 * INTEGRATION.md shows the drop-in gpuRender() written over them.
 *
 * Conventions: every function returns 0 on success or a negative rt_status; nothing
 * throws or exits (the reference prints and exit(99)s, main.cu:33-44).  rt_last_error()
 * gives the message of the calling thread's last failure.  Pointers named *_dev are
 * device (HBM) pointers on the context's device; everything else is host memory.
 * A context is bound to one device and one HIP stream and is not thread-safe; use one
 * context per GPU (the reference uses one host thread per GPU, main.cu:655-656).
 */
#ifndef RT_API_H
#define RT_API_H
#include <stdint.h>
#include "rt_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

enum rt_status {
    RT_OK            =  0,
    RT_ERR_ARG       = -1,   /* bad argument / scene fails validation */
    RT_ERR_HIP       = -2,   /* a HIP runtime call failed */
    RT_ERR_UNSUPPORTED = -3, /* scene graph the flat tables cannot express */
    RT_ERR_NO_SCENE  = -4,
    RT_ERR_RCCL      = -5    /* librccl could not be loaded, or an RCCL call failed */
};

typedef struct rt_ctx   rt_ctx;
typedef struct rt_scene rt_scene;

/* closest-hit record, the Intersect of Hitable.cuh:10-22 with the matPtr replaced by ids */
typedef struct rt_hit_record {
    int32_t hit;
    int32_t prim;          /* rt_object::prim_id, -1 on miss */
    float   t, u, v;
    float   p[3];
    float   n[3];
    int32_t front_face;
    int32_t material;
} rt_hit_record;           /* 52 bytes */

typedef struct rt_counters {
    uint64_t rays;            /* closest-hit queries issued by render calls since the last reset */
    uint64_t samples;         /* camera paths started */
    double   render_ms;       /* HIP-event time of those render kernels, summed */
    double   last_render_ms;  /* HIP-event time of the most recent render kernel */
    uint32_t render_launches;
    uint32_t last_integrator; /* form the most recent render call ran in: RT_INTEGRATOR_MEGAKERNEL or _WAVEFRONT; 0 = none yet */
    uint32_t wf_pools;        /* pools the most recent wavefront call ran as: 1 for the one-kernel form; 3 for the two-kernel forms
                                 (fewer when the context found fewer streams that really run side by side); 0 = no wavefront call yet */
    uint32_t wf_streams_found;/* streams the context found to overlap when it looked (once, at its first pooled wavefront call) */
    uint32_t wf_kernel;       /* form of the most recent wavefront call: 4 = the whole path in one persistent kernel (default; it
                                 runs as one "pool"), 1 / 2 / 3 = intersect + logic kernels (RT_WF_KERNEL); 0 = none yet */
    uint32_t reserved;
} rt_counters;                /* 56 bytes */

const char* rt_last_error(void);

/* ---- scenes (replaces createWorld<<<1,1>>> + the sceneN functions, main.cu:203-539) ---- */
/* Built-in scenes written against the scene-description API (cuda-raytracer_amd/api/):
 *   "cornell"         Cornell box                      (the reference's active scene, main.cu:278-306)
 *   "cornell_smoke"   same, boxes as ConstantMedium    (main.cu:296-297)
 *   "random_spheres"  ~488 spheres, sky integrator     (main.cu:203-248; procedural image texture)
 *   "lit_sphere"      RotateY'd spheres + emissive     (main.cu:251-275)
 *   "bvh_boxes"       BvhNode of boxes + light sphere  (main.cu:309-352)
 *   "cornell_cluster" glass box + instanced sphere BVH (main.cu:417-451)
 *   "final"           Next-Week final scene; param = boxes per side (0 -> 20)   (main.cu:355-414)
 *   "final_custom"    the reference's own final-scene variant: random box widths, light sphere + green
 *                     light rect below; param = boxes per side (0 -> 20)                 (main.cu:454-524)
 *   "tri_mesh"        synthetic triangle mesh + light; param = triangle count (0 -> 1,000,000)
 *   "kat"             one of every primitive / wrapper, for parity tests
 * C++ callers with their own scene functions use rt::lower() (csrc/lower.h) and
 * rt_scene_from_flat() instead. */
int  rt_scene_create_builtin(const char* name, int width, int height, int param, rt_scene** out);
/* deep-copies caller-owned tables */
int  rt_scene_from_flat(const rt_flat_scene* flat, rt_scene** out);
const rt_flat_scene* rt_scene_flat(const rt_scene* s);
int  rt_scene_default_background(const rt_scene* s);   /* RT_BG_* the scene is meant to be lit with */
void rt_scene_destroy(rt_scene* s);

/* ---- context (replaces cudaSetDevice + the allocations of gpuRender, main.cu:555-595) ---- */
/* stream: a hipStream_t to run on, or NULL for a private, non-blocking one.  NULL never means the device's default
 * stream: work the caller enqueues elsewhere (fills of accum_dev, copies, collectives) is then NOT ordered against
 * the library's kernels, and the caller orders it with rt_synchronize / its own synchronisation.  To share a stream
 * with torch, make a torch.cuda.Stream current and pass its handle (bench.py). */
int  rt_ctx_create(int device, void* stream, rt_ctx** out);
void rt_ctx_destroy(rt_ctx* c);
/* validates every index in the tables, then copies them to the device */
int  rt_scene_upload(rt_ctx* c, const rt_flat_scene* flat);

/* ---- closest hit (the body of colour()'s loop, main.cu:56; HitableList::hit) ---- */
/* n rays of 7 floats (origin, direction, time); media are ignored (no RNG in a bare query) */
int  rt_trace_rays(rt_ctx* c, int n, const float* rays7_dev, float tmin, float tmax, rt_hit_record* out_dev);
/* The same query through the traversal the WAVEFRONT form of the integrator renders with (the step interpreter over the
 * four-wide trees, run by the persistent intersect kernel over a queue of the n rays; RT_WF_KERNEL=2 / 3 select the
 * leaf-queue and the eight-wide intersect kernels) instead of the megakernel's, so that its hits can be held against the
 * reference's vectors directly and not only through whole renders.  tmin = .001f and
 * tmax = FLT_MAX are the integrator's own (main.cu:56) and fixed; media are ignored here too.  Uses the context's
 * wavefront state: not to be overlapped with a render on the same context. */
int  rt_trace_rays_wavefront(rt_ctx* c, int n, const float* rays7_dev, rt_hit_record* out_dev);
/* pinhole-centre primary ray of every pixel: u=(i+s_off)/W, v=(j+t_off)/H, lens offset 0,
 * time=time0; tmin=.001f, tmax=FLT_MAX.  prim_dev/t_dev are W*H, row 0 = bottom. */
int  rt_trace_primary(rt_ctx* c, int width, int height, float s_off, float t_off,
                      int32_t* prim_dev, float* t_dev);

/* ---- shading (Material::scatter + emitted, Material.cuh:44-144; Texture::value, Texture.cuh:7-61) ---- */
/* Parity hooks for the shading half of colour()'s loop body (main.cu:58-66), the counterpart of rt_trace_rays for the
 * geometry half.  Element i: incoming ray rays7[i] (origin, direction, time), hit record recs9[i] (t, u, v, p[3],
 * normal[3]: the fields of Intersect, Hitable.cuh:10-22, that materials read), material index materials[i], and the
 * RNG address (pixel, sample, bounce) = psb3[i] of the path it belongs to: the draws are the ones rt_render_accumulate
 * makes for that (seed, pixel, sample, bounce).  out[i] = what scatter() and emitted() return. */
typedef struct rt_scatter_record {
    int32_t ok;              /* scatter()'s return value */
    float   attenuation[3];
    float   origin[3], direction[3], time;   /* the scattered ray */
    float   emitted[3];
    uint32_t pad;
} rt_scatter_record;       /* 60 bytes */
/* materials_dev[i] outside the scene's material table: out[i].ok = -1, everything else 0 (never followed). */
int  rt_shade_rays(rt_ctx* c, int n, const float* rays7_dev, const float* recs9_dev, const int32_t* materials_dev,
                   const uint32_t* psb3_dev, uint64_t seed, rt_scatter_record* out_dev);
/* rgb_dev[3 i ..] = value(u, v, p) of texture `texture` for uvp5_dev[5 i ..] = (u, v, p[3]) */
int  rt_texture_values(rt_ctx* c, int texture, int n, const float* uvp5_dev, float* rgb_dev);

/* ---- render (replaces renderInit + render, main.cu:108-198, minus the mean and sqrt) ---- */
/* accum_dev[3*(j*W+i)+c] += sum over samples s in [sample_begin, sample_end) of colour() (background
 * RT_BG_BLACK) or colourUnlit() (RT_BG_SKY) for the camera path (pixel, s), max_depth bounces
 * (the reference passes 100).  Linear radiance sums; the image does not depend on how the sample
 * range is split over calls or GPUs except for fp32 summation order.  Asynchronous on the
 * context's stream. */
int  rt_render_accumulate(rt_ctx* c, int width, int height, uint32_t sample_begin, uint32_t sample_end,
                          uint64_t seed, int max_depth, int background, float* accum_dev);
/* The same for the pixels of the window [x0,x1) x [y0,y1) only; accum_dev is still the full W*H*3 buffer and
 * pixels outside the window are not touched.  Image tiles are the other way to split a frame over GPUs: the
 * reference's own two-GPU mode renders the left and right column halves (main.cu:128-174, gpuId 0/1) and
 * stitches them on the host (main.cu:711-730).  A pixel's paths are addressed by (seed, frame pixel, sample),
 * so windows that tile the frame add up to the full-frame image. */
int  rt_render_accumulate_window(rt_ctx* c, int width, int height, int x0, int y0, int x1, int y1,
                                 uint32_t sample_begin, uint32_t sample_end, uint64_t seed, int max_depth,
                                 int background, float* accum_dev);
/* No counterpart in the reference.  rt_render_accumulate for callers that cannot allocate (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc): always the megakernel form, full frame, and the
 * per-run partial frames that rt_render_accumulate keeps in a hipMalloc'ed per-context buffer live in a fixed
 * 48 MiB array built into the library's device image (one copy per GPU the library runs on, held until the process
 * ends).  The call makes no hipMalloc / hipFree / event or stream creation and is not timed (rt_counters' rays and
 * samples count it, its times do not).  Limit: calls of >= 128 samples are cut into runs (up to 16), and
 * runs x width x height x 3 floats must fit the array, i.e. up to 512 x 512 pixels at 16 runs; larger calls are refused
 * with RT_ERR_ARG (calls of < 128 samples use no scratch and have no frame limit).  Same bits as
 * rt_render_accumulate under RT_INTEGRATOR_MEGAKERNEL.  All calls on one GPU share that GPU's array: calls that
 * overlap on one GPU (two contexts or streams) race; serialise them. */
int  rt_render_accumulate_noalloc(rt_ctx* c, int width, int height, uint32_t sample_begin, uint32_t sample_end,
                                  uint64_t seed, int max_depth, int background, float* accum_dev);
/* Which form of the integrator rt_render_accumulate runs (no counterpart in the reference, whose render
 * kernel is one monolithic per-pixel loop, main.cu:125-198):
 *   RT_INTEGRATOR_MEGAKERNEL  one kernel, a path stays in its lane from camera to termination and the lane
 *                             starts its pixel's next sample the moment the path ends; asynchronous.
 *   RT_INTEGRATOR_WAVEFRONT   one persistent kernel whose every lane advances ITS ray through the BVHs by one step per
 *                             trip (a four-wide node, or one leaf primitive) and is handed its path's next ray, or the
 *                             next path, as soon as enough lanes of its wave are free (csrc/rt_wavefront4.h);
 *                             asynchronous.  RT_WF_KERNEL=1..3 in the environment select the older forms instead (path
 *                             state in HBM, an intersect kernel and a shading kernel alternating over a ray queue),
 *                             which block the host while their loop runs.
 *   RT_INTEGRATOR_AUTO        (default) wavefront for scenes whose BVHs hold at least 128 items (measured: 1.95x the
 *                             megakernel on 488 spheres, 1.02x on 16 boxes, 0.92x on the Cornell box); megakernel otherwise.
 * Both trace the same path for the same (seed, pixel, sample). */
enum rt_integrator { RT_INTEGRATOR_AUTO = 0, RT_INTEGRATOR_MEGAKERNEL = 1, RT_INTEGRATOR_WAVEFRONT = 2 };
int  rt_ctx_set_integrator(rt_ctx* c, int mode);
/* rgb_dev = gamma ? sqrtf(accum/spp) : accum/spp   (main.cu:192-196).  Asynchronous. */
int  rt_resolve(rt_ctx* c, const float* accum_dev, int width, int height, uint32_t spp, int gamma,
                float* rgb_dev);
/* Multi-GPU exchange step: adds the n accumulation buffers (accum_dev[i] on ctxs[i]'s device, `count` floats
 * each, normally 3*W*H) into accum_dev[root] with one RCCL ncclReduce per device over xGMI, enqueued on each
 * context's stream; the other buffers are left as they were.  Contexts must sit on n different devices of
 * this process.  Replaces the host-side stitch of the two half images (main.cu:711-730).  librccl is loaded
 * on the first call, not when this library is loaded; communicators are kept for the next call over the same
 * devices.  One process per GPU (bench.py) does the same through torch.distributed instead. */
int  rt_reduce_accum(rt_ctx* const* ctxs, float* const* accum_dev, int n, uint64_t count, int root);
/* The same sum, left distributed (the reference gathers its two halves on the host,
 * main.cu:711-730): the frame's rows are cut into n bands, rows_out[i]..rows_out[i+1] for context i (rows_out has
 * n + 1 entries; row 0 is the bottom row, as in accum), and context i's buffer receives the total of ITS band, in
 * place; the rest of its buffer keeps its own partial sums.  Every GPU then resolves and reads back 1/n of the
 * frame instead of GPU 0 all of it.  One ncclReduce per band, rooted at the band's owner, all in one group. */
int  rt_reduce_scatter_rows(rt_ctx* const* ctxs, float* const* accum_dev, int n, int width, int height, int* rows_out);
int  rt_synchronize(rt_ctx* c);
int  rt_get_counters(rt_ctx* c, rt_counters* out);   /* synchronizes first */
int  rt_reset_counters(rt_ctx* c);

/* ---- introspection (host only, no GPU): what rt_scene_upload derives from the tables ---- */
/* RT_F_* feature bits of the scene (cuda-raytracer_amd/csrc/rt_host.h): they select the kernel variant */
uint32_t rt_flat_features(const rt_flat_scene* flat);
/* the world list as the 32-byte scan records the closest-hit scan executes (replaces walking
 * HitableList::list, HitableList.cuh:17-32): writes up to cap records of 8 words and returns how many the
 * scene has (terminators included), or a negative rt_status */
int  rt_flat_program(const rt_flat_scene* flat, uint32_t* out_words, int cap_records);
/* the same list as the wavefront integrator takes it: the plain records without the BVH groups (return value =
 * their count, terminators included), and the BVH groups' object indices in list order (*n_groups; the world
 * itself when it has a BVH of its own) */
int  rt_flat_program_split(const rt_flat_scene* flat, uint32_t* out_words, int cap_records, int32_t* out_groups, int cap_groups,
                           int* n_groups);
/* sizes of the device BVHs: two-box nodes, four-box nodes, traversal stack entries each needs */
int  rt_flat_bvh_stats(const rt_flat_scene* flat, int* nodes2, int* depth2, int* nodes4, int* depth4);
/* the 8-wide quantised trees the third intersect kernel walks, built and checked on the host: every BVH item in exactly
 * one leaf-table row, every dequantised child box containing what it was made from, every node reached once.  Returns
 * the node count, or a negative rt_status (RT_ERR_UNSUPPORTED: a leaf of more than two items; such scenes render with
 * the four-wide kernels).  The out-parameters are optional. */
int  rt_flat_bvh8_check(const rt_flat_scene* flat, int* depth8, int* rows_used, float* max_q_slack);
/* how many child slots of the four-box nodes hold ONE item behind that item's own rt_object::bmin/bmax (a leaf is
 * opened that way only when its items' bounds are finite, ordered and inside the leaf's box; stale bounds in a
 * caller's table leave the leaf whole), or a negative rt_status */
int  rt_flat_bvh4_single_refs(const rt_flat_scene* flat);

/* ---- image output (replaces the quantise loop of main(), main.cu:668-697) ---- */
/* rgb (host, W*H*3, row 0 = bottom) -> 8-bit, rows flipped to top-down, int(255.99*c) clamped to
 * 0..255 (the reference does not clamp and wraps above 1).  out is W*H*3 bytes. */
int  rt_quantise_rgb8(const float* rgb, int width, int height, uint8_t* out);
int  rt_write_ppm(const char* path, const uint8_t* rgb8, int width, int height);
/* 8-bit RGB PNG, the format main() writes through stbi_write_png (main.cu:697, writeImage.h); rows are
 * stored uncompressed, so the file is a pure function of the pixels and needs no third-party code */
int  rt_write_png(const char* path, const uint8_t* rgb8, int width, int height);

/* ---- texture input (stands where main.cu:566-570 calls stbi_load on a JPEG; stb is out of scope) ---- */
/* binary PPM (P6, maxval 255) -> malloc'ed W*H*3 RGB8, top row first: what ImageTexture(pixels, w, h)
 * (Texture.cuh:45) takes.  Free with rt_free_pixels. */
int  rt_read_ppm(const char* path, uint8_t** rgb8_out, int* width, int* height);
void rt_free_pixels(uint8_t* rgb8);

#ifdef __cplusplus
}
#endif
#endif
