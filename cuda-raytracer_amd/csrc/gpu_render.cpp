/*
 * gpu_render.cpp — gpuRender() of the reference (main.cu:553-620) over the C ABI; contract in api/gpuRender.h.
 * What the reference does in six kernels (textureInit, randInit, createWorld, renderInit, render, freeWorld) is here:
 * run the scene function on the host, rt::lower(), upload, one rt_render_accumulate_window, one rt_resolve.
 */
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../include/rt_api.h"
#include "../api/gpuRender.h"
#include "lower.h"

namespace {

std::atomic<rtSceneFunction> g_scene_fn{nullptr};
std::atomic<int> g_background{RT_BG_BLACK};

/* main.cu:33-44: message, device reset, exit(99) */
void fail(const char* what, const std::string& why)
{
    std::fprintf(stderr, "gpuRender error at '%s': %s\n", what, why.c_str());
    (void)hipDeviceReset();
    std::exit(99);
}
void ck(int rc, const char* what) { if (rc != RT_OK) fail(what, rt_last_error()); }
void ckh(hipError_t e, const char* what) { if (e != hipSuccess) fail(what, hipGetErrorString(e)); }

} // namespace

void gpuRenderUseScene(rtSceneFunction scene, int background)
{
    g_scene_fn = scene;
    g_background = background == RT_BG_SKY ? RT_BG_SKY : RT_BG_BLACK;
}

Vector3* gpuRender(int width, int height, int numSamples, int /*tx*/, int /*ty*/, int gpuId)
{
    if (width <= 0 || height <= 0 || numSamples <= 0) fail("gpuRender", "width, height and numSamples must be positive");
    int n_dev = 0;
    ckh(hipGetDeviceCount(&n_dev), "hipGetDeviceCount");
    int device = (gpuId == 0 || gpuId == 1) ? gpuId : 0;                    /* main.cu:555-556 */
    if (device >= n_dev) device = 0;                                        /* one-GPU machine: both halves on device 0 */

    /* ---- createWorld (main.cu:527-539) on the host ---- */
    rt_scene* builtin = nullptr;
    rt::FlatScene flat;
    const rt_flat_scene* view = nullptr;
    int background = g_background;
    const rtSceneFunction fn = g_scene_fn;
    if (fn) {
        rt::Arena arena;                                                    /* owns every node the scene function news */
        std::vector<Hitable*> dList(488 + 2048);                            /* main.cu:590 allocates 488; the scenes of main.cu write up to 11 */
        Hitable* world = nullptr;
        Camera* cam = nullptr;
        curandState rs;
        curand_init(419, 0, 0, &rs);                                        /* randInit, main.cu:102-106 */
        uint8_t* pixels = nullptr;
        int tw = 0, th = 0;
        std::vector<uint8_t> procedural;
        if (rt_read_ppm("assets/earthmap.ppm", &pixels, &tw, &th) != RT_OK) {
            /* no asset: a deterministic 64 x 32 map, so that scenes with an image-textured sphere still lower */
            tw = 64; th = 32;
            procedural.resize((size_t)tw * th * 3);
            for (int j = 0; j < th; ++j)
                for (int i = 0; i < tw; ++i) {
                    uint8_t* p = &procedural[3 * ((size_t)j * tw + i)];
                    const bool land = ((i / 8 + j / 8) & 1) != 0;
                    p[0] = land ? 70 : 20; p[1] = land ? 140 : 60; p[2] = land ? 50 : (uint8_t)(160 + (i & 31));
                }
        }
        ImageTexture* tex = new ImageTexture(pixels ? pixels : procedural.data(), tw, th);     /* textureInit, main.cu:119-123 */
        fn(dList.data(), &world, &cam, width, height, &tex, &rs);
        std::string err;
        const bool ok = rt::lower(world, cam, rt::LowerOptions(), flat, err);
        rt_free_pixels(pixels);                                             /* the flat scene holds its own copy */
        if (!ok) fail("createWorld", "lowering failed: " + err);
        view = &flat.view;
    } else {
        ck(rt_scene_create_builtin("cornell", width, height, 0, &builtin), "rt_scene_create_builtin");     /* main.cu:533 */
        view = rt_scene_flat(builtin);
        background = rt_scene_default_background(builtin);
    }

    /* ---- renderInit + render (main.cu:108-198) ---- */
    rt_ctx* ctx = nullptr;
    ck(rt_ctx_create(device, nullptr, &ctx), "rt_ctx_create");
    ck(rt_scene_upload(ctx, view), "rt_scene_upload");
    const size_t n = (size_t)width * height * 3;
    float *accum = nullptr, *rgb = nullptr;
    ckh(hipSetDevice(device), "hipSetDevice");
    ckh(hipMalloc((void**)&accum, n * sizeof(float)), "hipMalloc");
    ckh(hipMalloc((void**)&rgb, n * sizeof(float)), "hipMalloc");
    ckh(hipMemset(accum, 0, n * sizeof(float)), "hipMemset");
    ckh(hipDeviceSynchronize(), "hipDeviceSynchronize");                    /* the context renders on a stream of its own */
    /* main.cu:132,150-153: GPU 0 takes the columns i < .5 * width, GPU 1 the same count shifted by int(.5 * width) */
    const int half = (width + 1) / 2;
    int x0 = 0, x1 = width;
    if (gpuId == 0) x1 = half;
    else if (gpuId == 1) x0 = width / 2;
    ck(rt_render_accumulate_window(ctx, width, height, x0, 0, x1, height, 0u, (uint32_t)numSamples, /*seed*/419, /*depth*/100, background, accum),
       "rt_render_accumulate_window");
    ck(rt_resolve(ctx, accum, width, height, (uint32_t)numSamples, /*gamma*/1, rgb), "rt_resolve");     /* main.cu:191-196 */
    ck(rt_synchronize(ctx), "rt_synchronize");

    static_assert(sizeof(Vector3) == 3 * sizeof(float), "Vector3 is three floats (Vector3.cuh)");
    Vector3* fb = nullptr;
    ckh(hipHostMalloc((void**)&fb, n * sizeof(float), hipHostMallocDefault), "hipHostMalloc");
    ckh(hipMemcpy(fb, rgb, n * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy");
    (void)hipFree(accum);
    (void)hipFree(rgb);
    rt_ctx_destroy(ctx);                                                    /* freeWorld + the cudaFree calls, main.cu:610-617 */
    if (builtin) rt_scene_destroy(builtin);
    return fb;
}

void gpuRenderFree(Vector3* fb)
{
    if (fb) (void)hipHostFree(fb);
}
