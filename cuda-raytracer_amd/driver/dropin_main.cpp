/*
 * dropin_main.cpp — a caller of gpuRender() written the way the reference's main() calls it (main.cu:652-704):
 * one std::async per GPU with gpuId 0 / 1, wait, stitch the two halves by column, flip the rows, quantise, write
 * the image; or one call with gpuId 2 for the full frame.  It includes the drop-in headers by the reference's
 * names and links librt_amd.so; tests/test_driver.py runs it and compares the frame buffers with the C ABI's.
 *   rt_dropin WIDTH HEIGHT SAMPLES N_GPUS [user]      writes output/dropin.png and output/dropin.f32 (the stitched
 *                                                     Vector3 frame buffer, row 0 = bottom)
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <iostream>
#include <string>
#include <sys/stat.h>
#include <vector>
#include "Vector3.cuh"
#include "Hitable.cuh"
#include "HitableList.cuh"
#include "Sphere.cuh"
#include "Camera.cuh"
#include "Material.cuh"
#include "Rectangle.cuh"
#include "Box.cuh"
#include "Transform.cuh"
#include "Texture.cuh"
#include "gpuRender.h"
#include "../../include/rt_api.h"

/* a scene function in the reference's form (compare main.cu:278-306) */
static void myScene(Hitable** dList, Hitable** dWorld, Camera** dCamera, int width, int height, ImageTexture** texture, curandState* randState)
{
    int i = 0;
    Material* white = new Lambert(new ConstantTexture(Vector3(.73f, .73f, .73f)));
    dList[i++] = new FlipNormals(new YZRect(0, 555, 0, 555, 555, new Lambert(new ConstantTexture(Vector3(.12f, .45f, .15f)))));
    dList[i++] = new YZRect(0, 555, 0, 555, 0, new Lambert(new ConstantTexture(Vector3(.65f, .05f, .05f))));
    dList[i++] = new XZRect(163, 393, 177, 382, 554, new DiffuseLight(new ConstantTexture(Vector3(4, 4, 4))));
    dList[i++] = new XZRect(0, 555, 0, 555, 0, white);
    dList[i++] = new FlipNormals(new XYRect(0, 555, 0, 555, 555, white));
    dList[i++] = new Translate(new RotateY(new Box(Vector3(0, 0, 0), Vector3(165, 165, 165), white), -18), Vector3(130, 0, 65));
    dList[i++] = new Sphere(Vector3(380, 100, 300), 100, new Lambert(*texture));
    dList[i++] = new Sphere(Vector3(200, 260 + 40 * curand_uniform(randState), 250), 60, new Metal(new ConstantTexture(Vector3(.8f, .8f, .9f)), .2f));
    *dWorld = new HitableList(dList, i);
    *dCamera = new Camera(Vector3(278, 278, -800), Vector3(278, 278, 0), Vector3(0, 1, 0), 40.f, float(width) / float(height), 0.f, 10.f, 0.f, 1.f);
}

int main(int argc, char** argv)
{
    if (argc < 5) { std::cerr << "usage: rt_dropin WIDTH HEIGHT SAMPLES N_GPUS [user]\n"; return 1; }
    const int width = std::atoi(argv[1]), height = std::atoi(argv[2]), numSamples = std::atoi(argv[3]), numGPUs = std::atoi(argv[4]);
    if (argc > 5 && !std::strcmp(argv[5], "user")) gpuRenderUseScene(&myScene, 0);
    const int tx = 24, ty = 24;
    std::vector<Vector3> frame((size_t)width * height);
    if (numGPUs == 2) {
        std::future<Vector3*> t0 = std::async(std::launch::async, &gpuRender, width, height, numSamples, tx, ty, 0);
        std::future<Vector3*> t1 = std::async(std::launch::async, &gpuRender, width, height, numSamples, tx, ty, 1);
        t0.wait();
        t1.wait();
        Vector3* fb0 = t0.get();
        Vector3* fb1 = t1.get();
        for (int j = 0; j < height; j++)
            for (int i = 0; i < width; i++) {
                const size_t pixelIndex = (size_t)j * width + i;
                frame[pixelIndex] = i < width / 2 ? fb0[pixelIndex] : fb1[pixelIndex];
            }
        gpuRenderFree(fb0);
        gpuRenderFree(fb1);
    } else {
        Vector3* fb = gpuRender(width, height, numSamples, tx, ty, 2);
        for (size_t p = 0; p < frame.size(); p++) frame[p] = fb[p];
        gpuRenderFree(fb);
    }
    /* main.cu:668-697: rows top-down, int(255.99 c) */
    std::vector<uint8_t> pixels((size_t)width * height * 3);
    size_t index = 0;
    for (int j = height - 1; j >= 0; j--)
        for (int i = 0; i < width; i++) {
            const Vector3& c = frame[(size_t)j * width + i];
            for (int k = 0; k < 3; k++) {
                int v = int(255.99 * c[k]);
                pixels[index++] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    (void)mkdir("output", 0777);
    if (rt_write_png("output/dropin.png", pixels.data(), width, height) != RT_OK) return 99;
    FILE* f = std::fopen("output/dropin.f32", "wb");
    if (!f || std::fwrite(frame.data(), sizeof(Vector3), frame.size(), f) != frame.size()) return 1;
    std::fclose(f);
    return 0;
}
