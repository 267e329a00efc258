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
 * "user" = the scene function in this file) and a scene parameter; and, with no counterpart in the reference
 * (progressive rendering):
 *   --scatter          leave the sum distributed: every GPU receives and hands back one row band (rt_reduce_scatter_rows)
 *   --tiles            with N GPUs, split the IMAGE into N column bands (the reference's own split, main.cu:128-174,
 *                      generalised from two halves) instead of splitting the samples; the reduce then acts as the stitch
 *   --chunk N          render N samples per call instead of all at once
 *   --checkpoint FILE  after every chunk write the sums so far (one GPU only); the file is replaced atomically
 *   --resume FILE      start from such a file; width, height, sample count and scene must be the same
 *   --stop-at N        leave after sample N with the checkpoint written and no image (an interrupted run)
 * A resumed run ends with bit for bit the image of the uninterrupted run with the same --chunk.
 */
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
static int g_chunk = 0, g_stop_at = 0;
static bool g_tiles = false;
static bool g_scatter = false;
static std::string g_checkpoint, g_resume;

/* checkpoint file: this header, then width*height*3 float sums (row 0 = bottom) */
struct CheckpointHeader {
    char magic[8];              /* "RTACCUM1" */
    int32_t width, height;
    uint32_t samples_done, samples_total;
    uint64_t seed;
    char scene[32];
    int32_t param, pad;
};

static bool writeCheckpoint(const std::string& path, const CheckpointHeader& h, const std::vector<float>& sums)
{
    const std::string tmp = path + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && std::fwrite(sums.data(), sizeof(float), sums.size(), f) == sums.size();
    ok = std::fclose(f) == 0 && ok;
    return ok && std::rename(tmp.c_str(), path.c_str()) == 0;
}
static bool readCheckpoint(const std::string& path, CheckpointHeader& h, std::vector<float>& sums)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, "RTACCUM1", 8) == 0 && h.width > 0 && h.height > 0 &&
              (long long)h.width * h.height <= (1ll << 30);
    if (ok) {
        sums.resize((size_t)h.width * h.height * 3);
        ok = std::fread(sums.data(), sizeof(float), sums.size(), f) == sums.size() && std::fgetc(f) == EOF;
    }
    std::fclose(f);
    return ok;
}

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
    /* this shard's samples: [s0, s1) of the global range, so that N shards together render numSamples */
    uint32_t s0 = (uint32_t)((long long)numSamples * gpuId / g_shards);
    uint32_t s1 = (uint32_t)((long long)numSamples * (gpuId + 1) / g_shards);
    int x0 = 0, x1 = width;
    if (g_tiles) {                      /* every sample, this GPU's column band */
        s0 = 0; s1 = (uint32_t)numSamples;
        x0 = (int)((long long)width * gpuId / g_shards);
        x1 = (int)((long long)width * (gpuId + 1) / g_shards);
    }
    CheckpointHeader ck;
    std::memset(&ck, 0, sizeof ck);
    std::memcpy(ck.magic, "RTACCUM1", 8);
    ck.width = width; ck.height = height; ck.samples_total = (uint32_t)numSamples; ck.seed = 419; ck.param = g_param;
    std::strncpy(ck.scene, g_scene.c_str(), sizeof ck.scene - 1);
    std::vector<float> host;
    if (!g_resume.empty()) {
        CheckpointHeader old;
        if (!readCheckpoint(g_resume, old, host)) { std::cerr << "cannot read checkpoint " << g_resume << "\n"; std::exit(1); }
        if (old.width != width || old.height != height || old.samples_total != ck.samples_total || old.seed != ck.seed ||
            old.param != ck.param || std::strncmp(old.scene, ck.scene, sizeof ck.scene) != 0 || old.samples_done > ck.samples_total) {
            std::cerr << "checkpoint " << g_resume << " belongs to another render\n";
            std::exit(1);
        }
        checkHip(hipMemcpy(sh.accum, host.data(), n * sizeof(float), hipMemcpyHostToDevice));
        s0 = old.samples_done;
        std::cerr << "resuming at sample " << s0 << "\n";
    }
    checkHip(hipDeviceSynchronize());      /* the context renders on a stream of its own */
    if (g_stop_at > 0 && (uint32_t)g_stop_at < s1) s1 = (uint32_t)g_stop_at;
    const uint32_t chunk = g_chunk > 0 ? (uint32_t)g_chunk : (s1 > s0 ? s1 - s0 : 1u);
    for (uint32_t s = s0; s < s1; s += chunk) {
        const uint32_t e = s1 - s < chunk ? s1 : s + chunk;
        checkRt(rt_render_accumulate_window(sh.ctx, width, height, x0, 0, x1, height, s, e, /*seed*/419, /*depth*/100, background, sh.accum));
        checkRt(rt_synchronize(sh.ctx));
        if (!g_checkpoint.empty()) {
            host.resize(n);
            checkHip(hipMemcpy(host.data(), sh.accum, n * sizeof(float), hipMemcpyDeviceToHost));
            ck.samples_done = e;
            if (!writeCheckpoint(g_checkpoint, ck, host)) { std::cerr << "cannot write checkpoint " << g_checkpoint << "\n"; std::exit(1); }
        }
        if (g_chunk > 0) std::cerr << "samples " << e << " / " << numSamples << "\n";
    }
    rt_counters c;
    checkRt(rt_get_counters(sh.ctx, &c));
    std::cerr << "GPU " << gpuId << ": columns [" << x0 << "," << x1 << "), samples [" << s0 << "," << s1 << "), " << c.rays << " rays in " << c.render_ms << " ms = "
              << (c.render_ms > 0 ? c.rays / c.render_ms / 1e3 : 0.0) << " Mrays/s\n";
    return sh;
}

int main(int argc, char** argv)
{
    int positional = 0;
    for (int a = 1; a < argc; ++a) {
        const std::string arg = argv[a];
        const bool has_value = a + 1 < argc;
        if (arg == "--tiles") g_tiles = true;
        else if (arg == "--scatter") g_scatter = true;
        else if (arg == "--chunk" && has_value) g_chunk = std::atoi(argv[++a]);
        else if (arg == "--stop-at" && has_value) g_stop_at = std::atoi(argv[++a]);
        else if (arg == "--checkpoint" && has_value) g_checkpoint = argv[++a];
        else if (arg == "--resume" && has_value) g_resume = argv[++a];
        else if (arg.rfind("--", 0) == 0) { std::cerr << "unknown or incomplete option " << arg << "\n"; return 1; }
        else if (positional == 0) { g_scene = arg; ++positional; }
        else if (positional == 1) { g_param = std::atoi(arg.c_str()); ++positional; }
        else { std::cerr << "too many arguments\n"; return 1; }
    }
    if (g_chunk < 0 || g_stop_at < 0) { std::cerr << "bad option value\n"; return 1; }
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
    if (numGPUs > (g_tiles ? width : numSamples)) numGPUs = g_tiles ? width : numSamples;
    if (numGPUs > 1 && (!g_checkpoint.empty() || !g_resume.empty() || g_stop_at > 0)) {
        std::cerr << "--checkpoint / --resume / --stop-at need # of GPUs = 1\n";
        return 1;
    }
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
    std::vector<rt_ctx*> ctxs;
    std::vector<float*> bufs;
    for (Shard& sh : shards) { ctxs.push_back(sh.ctx); bufs.push_back(sh.accum); }
    std::vector<float> sum(n);
    if (g_scatter) {                                                           /* every GPU ends with the total of its row band */
        std::vector<int> rows((size_t)numGPUs + 1);
        checkRt(rt_reduce_scatter_rows(ctxs.data(), bufs.data(), numGPUs, width, height, rows.data()));
        for (int g = 0; g < numGPUs; ++g) {
            checkRt(rt_synchronize(shards[g].ctx));
            const size_t off = (size_t)rows[g] * width * 3, cnt = (size_t)(rows[g + 1] - rows[g]) * width * 3;
            checkHip(hipSetDevice(g));
            if (cnt) checkHip(hipMemcpy(sum.data() + off, shards[g].accum + off, cnt * sizeof(float), hipMemcpyDeviceToHost));
        }
    } else {
        if (numGPUs > 1) {                                                     /* RCCL reduce over xGMI onto GPU 0 */
            checkRt(rt_reduce_accum(ctxs.data(), bufs.data(), numGPUs, n, 0));
            for (Shard& sh : shards) checkRt(rt_synchronize(sh.ctx));
        }
        checkHip(hipSetDevice(0));
        checkHip(hipMemcpy(sum.data(), shards[0].accum, n * sizeof(float), hipMemcpyDeviceToHost));
    }
    for (size_t g = 0; g < shards.size(); ++g) {
        checkHip(hipSetDevice((int)g));
        checkHip(hipFree(shards[g].accum));
        rt_ctx_destroy(shards[g].ctx);
        if (shards[g].scene) rt_scene_destroy(shards[g].scene);
    }
    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (g_stop_at > 0 && g_stop_at < numSamples) {
        std::cerr << "\nStopped after sample " << g_stop_at << " of " << numSamples << " in " << elapsed << " seconds" << std::endl;
        return 0;
    }
    const std::string stem = numGPUs > 1 ? "output/multiGPUTesting" : "output/singleGPUTesting";
    const std::string fileName = stem + ".png";
    std::cerr << "\nRendered in " << elapsed << " seconds to " << fileName << std::endl;

    /* mean, gamma 2 (main.cu:192-196), row flip + quantise (main.cu:668-697) */
    /* outCol /= float(numSamples) in the reference IS a multiplication by the reciprocal (Vector3.cuh:157-164: k = 1.0f / t) */
    const float inv = 1.0f / float(numSamples);
    for (float& v : sum) { v *= inv; v = std::sqrt(v); }
    std::vector<uint8_t> pixels((size_t)width * height * 3);
    checkRt(rt_quantise_rgb8(sum.data(), width, height, pixels.data()));
    (void)mkdir("output", 0777);
    checkRt(rt_write_png(fileName.c_str(), pixels.data(), width, height));
    checkRt(rt_write_ppm((stem + ".ppm").c_str(), pixels.data(), width, height));
    return 0;
}
