/*
 * rt_main.cpp — the reference's host program over librt_amd.so.
 *
 * Keeps the contract of the reference's main() and gpuRender() (main.cu:553-733) for whoever scripted
 * it: the four prompts on stdout ("Width: ", "Height: ", "Number of Samples: ", "# of GPUs: "), four
 * ints on stdin, the banner and "Rendered in X seconds to <file>" on stderr, the output names
 * output/singleGPUTesting / output/multiGPUTesting, row 0 of the frame buffer = bottom of the image,
 * gamma 2, int(255.99 c) quantisation, exit code 99 on a device error (main.cu:33-44).
 * What changes underneath:
 *   - the scene function below is written against the scene-description API exactly as a user of the
 *     reference writes sceneN() (same signature, same `new` calls); it runs on the host, rt::lower()
 *     flattens what it built, and nothing is built inside a kernel (main.cu:527-539);
 *   - gpuRender() is a few calls on the C ABI (include/rt_api.h);
 *   - "# of GPUs" = N shards the SAMPLES over N devices (any N, not only 2) instead of splitting the
 *     image into halves that each rebuild the world (main.cu:128-174, 652-699); the N sums are added onto
 *     GPU 0 by rt_reduce_accum (RCCL over xGMI; bench.py shows the one-process-per-GPU form of the same);
 *   - the image is written as <name>.png like the reference's (main.cu:663,697,707; rt_write_png instead of
 *     the vendored stb writer) and as <name>.ppm, with values clamped to 0..255 (the reference lets values
 *     above 1 wrap, main.cu:688-694);
 *   - the texture handed to the scene function is assets/earthmap.ppm when that file exists (the reference
 *     decodes assets/earthmap.jpg through stb, main.cu:566-570; neither the asset nor a JPEG decoder is here).
 * Optional argv: a built-in scene name (default "cornell" = the reference's active scene, main.cu:533;
 * "user" = the scene function in this file) and a scene parameter.
 */
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <future>
#include <iostream>
#include <string>
#include <sys/stat.h>
#include <vector>
#include "../../include/rt_api.h"
#include "../csrc/lower.h"

static void check_rt(int rc, const char* what, const char* file, int line)
{
    if (rc == RT_OK) return;
    /* the reference's convention: message, reset, exit(99) */
    std::cerr << "RT error = " << rc << " (" << rt_last_error() << ") at " << file << ":" << line << " '" << what << "'\n";
    (void)hipDeviceReset();
    std::exit(99);
}
#define checkRt(val) check_rt((val), #val, __FILE__, __LINE__)
#define checkHip(val) check_rt(((val) == hipSuccess ? RT_OK : RT_ERR_HIP), #val, __FILE__, __LINE__)

/* ---- a user's scene, in the reference's form (compare main.cu:278-306): Cornell box with a glass sphere ---- */
static void userScene(Hitable** dList, Hitable** dWorld, Camera** dCamera, int width, int height,
                      ImageTexture** texture, curandState* /*randState*/)
{
    int i = 0;
    Material* red = new Lambert(new ConstantTexture(Vector3(.65f, .05f, .05f)));
    Material* white = new Lambert(new ConstantTexture(Vector3(.73f, .73f, .73f)));
    Material* green = new Lambert(new ConstantTexture(Vector3(.12f, .45f, .15f)));
    Material* light = new DiffuseLight(new ConstantTexture(Vector3(7.f, 7.f, 7.f)));
    dList[i++] = new YZRect(0, 555, 0, 555, 555, green);
    dList[i++] = new YZRect(0, 555, 0, 555, 0, red);
    dList[i++] = new XZRect(213, 343, 227, 332, 554, light);
    dList[i++] = new XZRect(0, 555, 0, 555, 555, white);
    dList[i++] = new XZRect(0, 555, 0, 555, 0, white);
    dList[i++] = new FlipNormals(new XYRect(0, 555, 0, 555, 555, white));
    dList[i++] = new Translate(new RotateY(new Box(Vector3(0, 0, 0), Vector3(165, 330, 165), white), 15), Vector3(265, 0, 295));
    dList[i++] = new Sphere(Vector3(190, 90, 190), 90, new Glass(1.5f));
    if (*texture) dList[i++] = new Sphere(Vector3(420, 60, 120), 60, new Lambert(*texture));   /* main.cu:241 style */
    *dWorld = new HitableList(dList, i);
    *dCamera = new Camera(Vector3(278, 278, -800), Vector3(278, 278, 0), Vector3(0, 1, 0), 40.f,
                          float(width) / float(height), 0.f, 10.f, 0.f, 1.f);
}

static std::string g_scene = "cornell";
static int g_param = 0;
static int g_shards = 1;

/* one device's share of the render: its context, its W*H*3 buffer of linear radiance SUMS (not yet averaged) */
struct Shard {
    rt_ctx* ctx = nullptr;
    rt_scene* scene = nullptr;
    float* accum = nullptr;     /* device memory */
};

/* replaces gpuRender (main.cu:553-620): renders this device's share of the samples into its own buffer and
 * returns with the work enqueued and finished; gpuId = shard index = device index. */
static Shard renderShard(int width, int height, int numSamples, int gpuId)
{
    Shard sh;
    int background = RT_BG_BLACK;
    rt::FlatScene flat;                 /* storage of a user scene; unused for built-ins */
    const rt_flat_scene* view = nullptr;
    if (g_scene == "user") {
        rt::Arena arena;                /* owns every node the scene function news */
        Hitable* world = nullptr;
        Camera* cam = nullptr;
        std::vector<Hitable*> dList(488);                                  /* main.cu:590 */
        curandState rs;
        curand_init(419, 0, 0, &rs);                                       /* main.cu:105 */
        ImageTexture* tex = nullptr;
        uint8_t* earth = nullptr;
        int tw = 0, th = 0;
        if (rt_read_ppm("assets/earthmap.ppm", &earth, &tw, &th) == RT_OK) tex = new ImageTexture(earth, tw, th);
        userScene(dList.data(), &world, &cam, width, height, &tex, &rs);
        std::string err;
        if (!rt::lower(world, cam, rt::LowerOptions(), flat, err)) { std::cerr << "lowering failed: " << err << "\n"; std::exit(99); }
        rt_free_pixels(earth);          /* the flat scene holds its own copy */
        view = &flat.view;
    } else {
        checkRt(rt_scene_create_builtin(g_scene.c_str(), width, height, g_param, &sh.scene));
        view = rt_scene_flat(sh.scene);
        background = rt_scene_default_background(sh.scene);
    }
    checkRt(rt_ctx_create(gpuId, nullptr, &sh.ctx));
    checkRt(rt_scene_upload(sh.ctx, view));
    const size_t n = (size_t)width * height * 3;
    checkHip(hipSetDevice(gpuId));
    checkHip(hipMalloc((void**)&sh.accum, n * sizeof(float)));
    checkHip(hipMemset(sh.accum, 0, n * sizeof(float)));
    checkHip(hipDeviceSynchronize());      /* the context renders on a stream of its own */
    /* this shard's samples: [s0, s1) of the global range, so that N shards together render numSamples */
    const uint32_t s0 = (uint32_t)((long long)numSamples * gpuId / g_shards);
    const uint32_t s1 = (uint32_t)((long long)numSamples * (gpuId + 1) / g_shards);
    checkRt(rt_render_accumulate(sh.ctx, width, height, s0, s1, /*seed*/419, /*depth*/100, background, sh.accum));
    checkRt(rt_synchronize(sh.ctx));
    rt_counters c;
    checkRt(rt_get_counters(sh.ctx, &c));
    std::cerr << "GPU " << gpuId << ": samples [" << s0 << "," << s1 << "), " << c.rays << " rays in " << c.render_ms << " ms = "
              << (c.render_ms > 0 ? c.rays / c.render_ms / 1e3 : 0.0) << " Mrays/s\n";
    return sh;
}

int main(int argc, char** argv)
{
    if (argc > 1) g_scene = argv[1];
    if (argc > 2) g_param = std::atoi(argv[2]);
    std::cout << "Width: ";
    int width = 0;
    std::cin >> width;
    std::cout << "Height: ";
    int height = 0;
    std::cin >> height;
    std::cout << "Number of Samples: ";
    int numSamples = 0;
    std::cin >> numSamples;
    std::cout << "# of GPUs: ";
    int numGPUs = 0;
    std::cin >> numGPUs;
    if (!std::cin || width <= 0 || height <= 0 || numSamples <= 0) { std::cerr << "bad input\n"; return 1; }
    int visible = 0;
    checkHip(hipGetDeviceCount(&visible));
    if (numGPUs < 1) numGPUs = 1;
    if (numGPUs > visible) { std::cerr << "only " << visible << " GPU(s) visible\n"; numGPUs = visible > 0 ? visible : 1; }
    if (numGPUs > numSamples) numGPUs = numSamples;
    g_shards = numGPUs;
    std::cerr << "Rendering a " << width << "x" << height << " image with " << numGPUs << (numGPUs > 1 ? " GPUs" : " GPU");
    time_t now = time(0);
    std::cerr << "\nRender initiated on " << ctime(&now) << std::endl;

    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::future<Shard>> jobs;                                      /* one host thread per GPU, main.cu:655-656 */
    for (int g = 0; g < numGPUs; ++g) jobs.push_back(std::async(std::launch::async, &renderShard, width, height, numSamples, g));
    std::vector<Shard> shards;
    for (auto& j : jobs) shards.push_back(j.get());
    const size_t n = (size_t)width * height * 3;
    if (numGPUs > 1) {                                                         /* RCCL reduce over xGMI onto GPU 0 */
        std::vector<rt_ctx*> ctxs;
        std::vector<float*> bufs;
        for (Shard& sh : shards) { ctxs.push_back(sh.ctx); bufs.push_back(sh.accum); }
        checkRt(rt_reduce_accum(ctxs.data(), bufs.data(), numGPUs, n, 0));
        for (Shard& sh : shards) checkRt(rt_synchronize(sh.ctx));
    }
    std::vector<float> sum(n);
    checkHip(hipSetDevice(0));
    checkHip(hipMemcpy(sum.data(), shards[0].accum, n * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t g = 0; g < shards.size(); ++g) {
        checkHip(hipSetDevice((int)g));
        checkHip(hipFree(shards[g].accum));
        rt_ctx_destroy(shards[g].ctx);
        if (shards[g].scene) rt_scene_destroy(shards[g].scene);
    }
    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const std::string stem = numGPUs > 1 ? "output/multiGPUTesting" : "output/singleGPUTesting";
    const std::string fileName = stem + ".png";
    std::cerr << "\nRendered in " << elapsed << " seconds to " << fileName << std::endl;

    /* mean, gamma 2 (main.cu:192-196), row flip + quantise (main.cu:668-697) */
    const float inv = 1.0f / float(numSamples);
    for (float& v : sum) { v *= inv; v = std::sqrt(v); }
    std::vector<uint8_t> pixels((size_t)width * height * 3);
    checkRt(rt_quantise_rgb8(sum.data(), width, height, pixels.data()));
    (void)mkdir("output", 0777);
    checkRt(rt_write_png(fileName.c_str(), pixels.data(), width, height));
    checkRt(rt_write_ppm((stem + ".ppm").c_str(), pixels.data(), width, height));
    return 0;
}
