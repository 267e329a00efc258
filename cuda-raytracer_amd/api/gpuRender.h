/*
 * gpuRender.h — the reference's render entry with the reference's own signature, over the C ABI.
 *
 *     Vector3* gpuRender(int width, int height, int numSamples, int tx, int ty, int gpuId)      (main.cu:553)
 *
 * A program written like the reference's main() (main.cu:627-733) keeps calling it as before:
 *   - the result is a width*height array of Vector3 holding gamma-2 values (sqrt of the mean, main.cu:191-196),
 *     row 0 = the BOTTOM row of the image, pixel (i, j) at fb[j * width + i];
 *   - gpuId 0 renders the left half of the columns and gpuId 1 the right half into a full-size buffer, each on the
 *     device of that number (main.cu:128-174, 555-556); any other gpuId renders the full frame on device 0.  A
 *     two-GPU caller stitches `i < width / 2 ? fb0 : fb1` exactly as main.cu:672-687 does;
 *   - tx, ty (the reference's 24 x 24 CUDA block, main.cu:644-645) are accepted and ignored: the launch shape is
 *     the library's business;
 *   - a device or scene error prints a message and exits with code 99 (checkCudaErrors, main.cu:33-44).
 * Differences a caller sees: the buffer is pinned host memory instead of CUDA managed memory and is released
 * with gpuRenderFree() where the reference calls cudaFree (main.cu:698-699); when the machine has no device
 * `gpuId` (a one-GPU box) the half is rendered on device 0; every path's random numbers are Philox4x32-10
 * addressed by (seed 419, pixel, sample) instead of one XORWOW state per pixel (main.cu:116).
 *
 * The reference chooses its scene by editing createWorld (main.cu:531-537) and its integrator by editing the
 * render kernel (main.cu:141-142).  Here: gpuRenderUseScene(fn, background) with a scene function of the
 * reference's signature (main.cu:203) before the first gpuRender call; without it the Cornell box of main.cu:533.
 * The scene function is handed assets/earthmap.ppm as its ImageTexture when that file exists (the reference
 * decodes assets/earthmap.jpg through stb, main.cu:566-574), else a built-in procedural map.
 */
#ifndef RT_GPU_RENDER_H
#define RT_GPU_RENDER_H
#include "rt_scene_api.h"

typedef void (*rtSceneFunction)(Hitable** dList, Hitable** dWorld, Camera** dCamera, int width, int height,
                                ImageTexture** texture, curandState* randState);

/* background: 0 = colour() (black, emitters light the scene), 1 = colourUnlit() (sky gradient)  (main.cu:48-100) */
void gpuRenderUseScene(rtSceneFunction scene, int background);
Vector3* gpuRender(int width, int height, int numSamples, int tx, int ty, int gpuId);
void gpuRenderFree(Vector3* fb);

#endif
