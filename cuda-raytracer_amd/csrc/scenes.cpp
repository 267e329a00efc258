/*
 * scenes.cpp — built-in scenes, written against the scene-description API exactly the way a
 * user of the reference writes a sceneN() function (reference main.cu:203-524 are the models:
 * same objects, same constants, same list order), plus the rt_scene_* half of the C ABI.
 *
 * Scene-construction randomness comes from the API's host curand_uniform (Philox, seed 419);
 * the reference's XORWOW stream is not reproducible off CUDA, so random layouts match the
 * reference in distribution, not draw for draw.  `earthmap.jpg` is not in the reference
 * repository (.gitignore:7-8), so image-textured spheres get a procedural RGB8 map.
 */
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "../../include/rt_api.h"
#include "lower.h"
#include "rt_host.h"

struct rt_scene {
    rt::FlatScene flat;
    int background = RT_BG_BLACK;
};

namespace {

struct Built {
    Hitable* world = nullptr;
    Camera* cam = nullptr;
    int background = RT_BG_BLACK;
};

#define RND (curand_uniform(&rng))

/* pointer arrays live in the scene's arena like the nodes do */
Hitable** ptr_array(size_t n) { return (Hitable**)rt::Arena::current()->allocate((n ? n : 1) * sizeof(Hitable*)); }

/* a 256x128 RGB8 "planet": bands + a continent-like blob pattern, deterministic.  Built once, by the
 * initialiser of a function-local static (thread-safe in C++11): rt_render builds the scene of every GPU in a
 * thread of its own (driver/rt_main.cpp), so several threads can arrive here at once. */
const std::vector<unsigned char>& procedural_map(int& w, int& h)
{
    static const std::vector<unsigned char> img = [] {
        const int mw = 256, mh = 128;
        std::vector<unsigned char> m((size_t)mw * mh * 3);
        for (int j = 0; j < mh; ++j)
            for (int i = 0; i < mw; ++i) {
                unsigned v = (unsigned)(i * 7 + j * 13) ^ (unsigned)((i >> 3) * (j >> 2));
                bool land = ((i / 16 + j / 16) % 3 == 0) ^ ((v & 31u) < 6u);
                unsigned char* p = &m[3 * ((size_t)j * mw + i)];
                p[0] = land ? (unsigned char)(60 + (v & 63u)) : (unsigned char)(10 + (j & 15));
                p[1] = land ? (unsigned char)(120 + (v & 31u)) : (unsigned char)(40 + (i & 31));
                p[2] = land ? (unsigned char)(40 + (v & 15u)) : (unsigned char)(150 + (v & 63u));
            }
        return m;
    }();
    w = 256; h = 128;
    return img;
}

Camera* look(Vector3 from, Vector3 at, float vfov, int width, int height, float aperture, float focus)
{
    return new Camera(from, at, Vector3(0, 1, 0), vfov, float(width) / float(height), aperture, focus, 0.f, 1.f);
}

void cornell_walls(Hitable** list, int& i, Material* white)
{
    Material* red = new Lambert(new ConstantTexture(Vector3(.65, .05, .05)));
    Material* green = new Lambert(new ConstantTexture(Vector3(.12, .45, .15)));
    Material* light = new DiffuseLight(new ConstantTexture(Vector3(1, 1, 1)));
    list[i++] = new YZRect(0, 555, 0, 555, 555, green);
    list[i++] = new YZRect(0, 555, 0, 555, 0, red);
    list[i++] = new XZRect(113, 443, 127, 432, 554, light);
    list[i++] = new XZRect(0, 555, 0, 555, 555, white);
    list[i++] = new XZRect(0, 555, 0, 555, 0, white);
    list[i++] = new FlipNormals(new XYRect(0, 555, 0, 555, 555, white));
}

Built cornell(int width, int height, bool smoke)
{
    Hitable** list = ptr_array(16);
    int i = 0;
    Material* white = new Lambert(new ConstantTexture(Vector3(.73, .73, .73)));
    cornell_walls(list, i, white);
    Hitable* b1 = new Translate(new RotateY(new Box(Vector3(0, 0, 0), Vector3(165, 165, 165), white), -18), Vector3(130, 0, 65));
    Hitable* b2 = new Translate(new RotateY(new Box(Vector3(0, 0, 0), Vector3(165, 330, 165), white), 15), Vector3(265, 0, 295));
    if (smoke) {
        curandState unused;
        curand_init(419, 0, 0, &unused);
        list[i++] = new ConstantMedium(b1, .01, new ConstantTexture(Vector3(1.f, 1.f, 1.f)), &unused);
        list[i++] = new ConstantMedium(b2, .01, new ConstantTexture(Vector3(0.f, 0.f, 0.f)), &unused);
    } else {
        list[i++] = b1;
        list[i++] = b2;
    }
    Built b;
    b.world = new HitableList(list, i);
    b.cam = look(Vector3(278, 278, -800), Vector3(278, 278, 0), 40.f, width, height, 0.f, 10.f);
    return b;
}

Built random_spheres(int width, int height)
{
    curandState rng;
    curand_init(419, 0, 0, &rng);
    Hitable** list = ptr_array(22 * 22 + 8);
    Texture* checker = new CheckerTexture(new ConstantTexture(Vector3(.2f, .3f, .1f)), new ConstantTexture(Vector3(.9f, .9f, .9f)));
    int i = 0;
    list[i++] = new Sphere(Vector3(0, -1000.0, -1), 1000, new Lambert(checker));
    for (int a = -11; a < 11; a++)
        for (int b = -11; b < 11; b++) {
            float choose = RND;
            float cx = a + RND, cz = b + RND;
            Vector3 centre(cx, .2f, cz);
            if (choose < .8f) {
                float lift = .5f * RND;
                float r = RND * RND, g = RND * RND, bl = RND * RND;
                list[i++] = new MovingSphere(centre, centre + Vector3(0, lift, 0), 0.f, 1.f, .2f,
                                             new Lambert(new ConstantTexture(Vector3(r, g, bl))));
            } else if (choose < .95f) {
                float r = .5f * (1.f + RND), g = .5f * (1.f + RND), bl = .5f * (1.f + RND);
                float fuzz = 0.5f * RND;
                list[i++] = new Sphere(centre, .2f, new Metal(new ConstantTexture(Vector3(r, g, bl)), fuzz));
            } else list[i++] = new Sphere(centre, .2f, new Glass(1.5f));
        }
    int tw, th;
    const std::vector<unsigned char>& img = procedural_map(tw, th);
    list[i++] = new Sphere(Vector3(-4, 1, 0), 1.0, new Glass(1.5f));
    list[i++] = new Sphere(Vector3(0, 1, 0), 1.f, new Lambert(new ImageTexture(const_cast<unsigned char*>(img.data()), tw, th)));
    list[i++] = new Sphere(Vector3(4, 1, 0), 1.0, new Metal(new ConstantTexture(Vector3(0.7, 0.6, 0.5)), .1));
    Built b;
    b.world = new HitableList(list, i);
    b.cam = look(Vector3(13, 2, 3), Vector3(0, 0, 0), 20.f, width, height, 0.f, 10.f);
    b.background = RT_BG_SKY; /* the scene has no emitter */
    return b;
}

Built lit_sphere(int width, int height)
{
    Hitable** list = ptr_array(4);
    Texture* checker = new CheckerTexture(new ConstantTexture(Vector3(.2f, .3f, .1f)), new ConstantTexture(Vector3(.9f, .9f, .9f)));
    int tw, th;
    const std::vector<unsigned char>& img = procedural_map(tw, th);
    int i = 0;
    list[i++] = new Sphere(Vector3(0.f, -1000.f, -1.f), 1000, new Lambert(checker));
    list[i++] = new RotateY(new Sphere(Vector3(0, 1, 0), 1, new Lambert(new ImageTexture(const_cast<unsigned char*>(img.data()), tw, th))), 0);
    list[i++] = new RotateY(new Sphere(Vector3(0, 7, 0), 2, new DiffuseLight(new ConstantTexture(Vector3(4, 4, 4)))), 120);
    Built b;
    b.world = new HitableList(list, i);
    b.cam = look(Vector3(13, 2, 3), Vector3(0, 0, 0), 30.f, width, height, 0.1f, 10.f);
    return b;
}

Built bvh_boxes(int width, int height)
{
    curandState rng;
    curand_init(419, 0, 0, &rng);
    const int nb = 4;
    Hitable** boxes = ptr_array(nb * nb);
    Material* ground = new Lambert(new ConstantTexture(Vector3(0.48, 0.83, 0.53)));
    int n = 0;
    for (int i = 0; i < nb; i++)
        for (int j = 0; j < nb; j++) {
            float w = 100, x0 = -1000 + w * i, z0 = -1000 + w * j;
            float y1 = (RND + 0.01) * 100;
            boxes[n++] = new Box(Vector3(x0, 0, z0), Vector3(x0 + w, y1, z0 + w), ground);
        }
    Hitable** list = ptr_array(2);
    list[0] = new BvhNode(boxes, n, 0, 1, &rng);
    list[1] = new Sphere(Vector3(-800, 400, -800), 120, new DiffuseLight(new ConstantTexture(Vector3(8, 8, 8))));
    Built b;
    b.world = new HitableList(list, 2);
    /* main.cu:339-340 looks from (13,2,3) at the origin and sees none of the boxes; the camera is moved
     * so that they are in view */
    b.cam = look(Vector3(-800, 320, -1350), Vector3(-800, 40, -800), 40.f, width, height, 0.f, 10.f);
    return b;
}

Built cornell_cluster(int width, int height)
{
    curandState rng;
    curand_init(419, 0, 0, &rng);
    Hitable** list = ptr_array(16);
    int i = 0;
    Material* white = new Lambert(new ConstantTexture(Vector3(.73, .73, .73)));
    cornell_walls(list, i, white);
    list[i++] = new Translate(new RotateY(new Box(Vector3(0, 0, 0), Vector3(165, 165, 165), new Glass(1.5)), -18), Vector3(130, 0, 65));
    const int ns = 1000;
    Hitable** cluster = ptr_array(ns);
    for (int j = 0; j < ns; j++) {
        float x = 165 * RND, y = 330 * RND, z = 165 * RND;
        cluster[j] = new Sphere(Vector3(x, y, z), 10, white);
    }
    list[i++] = new Translate(new RotateY(new BvhNode(cluster, ns, 0.f, 1.f, &rng), 15), Vector3(265, 0, 295));
    Built b;
    b.world = new HitableList(list, i);
    b.cam = look(Vector3(278, 278, -800), Vector3(278, 278, 0), 40.f, width, height, 0.f, 10.f);
    return b;
}

Built final_scene(int width, int height, int per_side)
{
    curandState rng;
    curand_init(419, 0, 0, &rng);
    const int nb = per_side > 0 ? per_side : 20;
    /* the ground grid always spans 20*150 units; more boxes per side = finer boxes */
    const float w = 3000.f / float(nb);
    Hitable** boxes = ptr_array((size_t)nb * nb);
    Material* white = new Lambert(new ConstantTexture(Vector3(.73, .73, .73)));
    Material* ground = new Lambert(new ConstantTexture(Vector3(.48, .83, .53)));
    int n = 0;
    for (int i = 0; i < nb; i++)
        for (int j = 0; j < nb; j++) {
            float x0 = -1000 + i * w, z0 = -1000 + j * w;
            float y1 = 100 * (RND + 0.01);
            boxes[n++] = new Box(Vector3(x0, 0, z0), Vector3(x0 + w, y1, z0 + w), ground);
        }
    Hitable** list = ptr_array(16);
    int l = 0;
    list[l++] = new BvhNode(boxes, n, 0, 200, &rng);
    list[l++] = new XZRect(123, 423, 147, 412, 554, new DiffuseLight(new ConstantTexture(Vector3(3, 3, 3))));
    Vector3 centre(400, 400, 200);
    list[l++] = new MovingSphere(centre, centre + Vector3(30, 0, 0), 0, 1, 50, new Lambert(new ConstantTexture(Vector3(.7, .3, .1))));
    list[l++] = new Sphere(Vector3(260, 150, 45), 50, new Glass(1.5));
    list[l++] = new Sphere(Vector3(0, 150, 145), 50, new Metal(new ConstantTexture(Vector3(.8, .8, .9)), 10.f));
    Hitable* boundary = new Sphere(Vector3(360, 150, 145), 70, new Glass(1.5));
    list[l++] = boundary;
    list[l++] = new ConstantMedium(boundary, .2, new ConstantTexture(Vector3(.2, .4, .9)), &rng);
    boundary = new Sphere(Vector3(0, 0, 0), 5000, new Glass(1.5f));
    list[l++] = new ConstantMedium(boundary, .0001, new ConstantTexture(Vector3(1, 1, 1)), &rng);
    int tw, th;
    const std::vector<unsigned char>& img = procedural_map(tw, th);
    list[l++] = new Sphere(Vector3(400, 200, 400), 100, new Lambert(new ImageTexture(const_cast<unsigned char*>(img.data()), tw, th)));
    const int ns = 1000;
    Hitable** cluster = ptr_array(ns);
    for (int j = 0; j < ns; j++) {
        float x = 165 * RND, y = 165 * RND, z = 165 * RND;
        cluster[j] = new Sphere(Vector3(x, y, z), 10, white);
    }
    list[l++] = new Translate(new RotateY(new BvhNode(cluster, ns, 0.f, 1.f, &rng), 15), Vector3(-100, 270, 395));
    Built b;
    b.world = new HitableList(list, l);
    b.cam = look(Vector3(478, 278, -600), Vector3(278, 278, 0), 40.f, width, height, 0.f, 10.f);
    return b;
}

/* The reference's seventh scene (main.cu:454-524), its own variation of the final scene: ground boxes whose WIDTH is
 * random as well (:473, so the grid overlaps and gaps), no ceiling light but a green light rect far below the world at
 * y = -3000 (:491) and a white light sphere of radius 500 at y = 1500 (:509).  The box sizes come from a COPY of the
 * scene's RNG state (`localRandState`, :456) and the cluster's sphere centres from the state itself (:503), so the two
 * sequences start at the same uniforms: kept.  param = boxes per side (0 -> 20). */
Built final_custom(int width, int height, int per_side)
{
    curandState rng;                      /* RND draws from this one: the reference's localRandState copy */
    curand_init(419, 0, 0, &rng);
    curandState shared = rng;             /* *randState, untouched by the box loop */
    const int nb = per_side > 0 ? per_side : 20;
    Hitable** boxes = ptr_array((size_t)nb * nb);
    Material* white = new Lambert(new ConstantTexture(Vector3(.73, .73, .73)));
    Material* ground = new Lambert(new ConstantTexture(Vector3(.48, .83, .53)));
    int n = 0;
    for (float i = 0; i < nb; i++)
        for (float j = 0; j < nb; j++) {
            const float w = 100 * (RND + .01f);
            const float x0 = -1000 + i * w, z0 = -1000 + j * w;
            const float y1 = 100 * (RND + 0.01f);
            boxes[n++] = new Box(Vector3(x0, 0, z0), Vector3(x0 + w, y1, z0 + w), ground);
        }
    Hitable** list = ptr_array(16);
    int l = 0;
    list[l++] = new BvhNode(boxes, n, 0, 200, &shared);
    list[l++] = new XZRect(-1000, 1000, -1000, 1000, -3000,
                           new DiffuseLight(new ConstantTexture(Vector3(2 * (22 / 255.f), 2 * (217 / 255.f), 2 * (25 / 255.f)))));
    const Vector3 centre(400, 400, 200);
    list[l++] = new MovingSphere(centre, centre + Vector3(30, 0, 0), 0, 1, 50, new Lambert(new ConstantTexture(Vector3(.7, .3, .1))));
    list[l++] = new Sphere(Vector3(260, 150, 45), 50, new Glass(1.5));
    list[l++] = new Sphere(Vector3(0, 150, 145), 50, new Metal(new ConstantTexture(Vector3(.8, .8, .9)), 10.f));
    Hitable* boundary = new Sphere(Vector3(360, 150, 145), 70, new Glass(1.5));
    list[l++] = boundary;
    list[l++] = new ConstantMedium(boundary, .2, new ConstantTexture(Vector3(.2, .4, .9)), &shared);
    boundary = new Sphere(Vector3(0, 0, 0), 5000, new Glass(1.5f));
    list[l++] = new ConstantMedium(boundary, .0001, new ConstantTexture(Vector3(1, 1, 1)), &shared);
    int tw, th;
    const std::vector<unsigned char>& img = procedural_map(tw, th);
    list[l++] = new Sphere(Vector3(400, 200, 400), 100, new Lambert(new ImageTexture(const_cast<unsigned char*>(img.data()), tw, th)));
    const int ns = 1000;
    Hitable** cluster = ptr_array(ns);
    for (int j = 0; j < ns; j++) {
        const float x = 165 * curand_uniform(&shared), y = 165 * curand_uniform(&shared), z = 165 * curand_uniform(&shared);
        cluster[j] = new Sphere(Vector3(x, y, z), 10, white);
    }
    list[l++] = new Translate(new RotateY(new BvhNode(cluster, ns, 0.f, 1.f, &shared), 15), Vector3(-100, 270, 395));
    list[l++] = new Sphere(Vector3(0, 1500, 0), 500, new DiffuseLight(new ConstantTexture(Vector3(1, 1, 1))));
    Built b;
    b.world = new HitableList(list, l);
    b.cam = look(Vector3(478, 278, -600), Vector3(278, 278, 0), 40.f, width, height, 0.f, 10.f);
    return b;
}

/* Synthetic mesh: a displaced nx x nz grid (two Lambert triangles per cell) over [-500,500]^2,
 * heights from an LCG (seed 1), under a large emissive rect. */
Built tri_mesh(int width, int height, int n_tris)
{
    if (n_tris <= 0) n_tris = 1000000;
    int cells = n_tris / 2;
    if (cells < 1) cells = 1;
    int nx = 1;
    while ((long long)(nx + 1) * (nx + 1) <= cells) ++nx;
    int nz = cells / nx;
    uint32_t lcg = 1u;
    std::vector<float> hgt((size_t)(nx + 1) * (nz + 1));
    for (float& hv : hgt) {
        lcg = lcg * 1664525u + 1013904223u;
        hv = 40.f * (float)(lcg >> 8) * (1.0f / 16777216.0f);
    }
    Material* mats[4] = {new Lambert(new ConstantTexture(Vector3(.73, .73, .73))), new Lambert(new ConstantTexture(Vector3(.65, .25, .2))),
                         new Lambert(new ConstantTexture(Vector3(.2, .45, .65))), new Lambert(new ConstantTexture(Vector3(.3, .6, .25)))};
    Hitable** tris = ptr_array((size_t)nx * nz * 2);
    size_t n = 0;
    const float sx = 1000.f / nx, sz = 1000.f / nz;
    /* Triangle::hit (Triangle.cuh:24,26) reports the triangle on the plane mirrored through the
     * origin along its normal; each triangle is therefore stored pre-mirrored (shifted by
     * -2 dot(r0,n) n), so that what the reference semantics render is the connected terrain. */
    auto emit = [&](Vector3 r0, Vector3 r1, Vector3 r2, Material* m) {
        Vector3 nrm = unitVector(cross(r1 - r0, r2 - r0));
        Vector3 shift = (2.f * dot(r0, nrm)) * nrm;
        tris[n++] = new Triangle(r0 - shift, r1 - shift, r2 - shift, m);
    };
    for (int iz = 0; iz < nz; ++iz)
        for (int ix = 0; ix < nx; ++ix) {
            auto P = [&](int a, int b) { return Vector3(-500.f + a * sx, hgt[(size_t)b * (nx + 1) + a], -500.f + b * sz); };
            Material* m = mats[((ix >> 4) + (iz >> 4)) & 3];
            emit(P(ix, iz), P(ix + 1, iz), P(ix + 1, iz + 1), m);
            emit(P(ix, iz), P(ix + 1, iz + 1), P(ix, iz + 1), m);
        }
    curandState rng;
    curand_init(419, 0, 0, &rng);
    Hitable** list = ptr_array(2);
    list[0] = new BvhNode(tris, (int)n, 0.f, 1.f, &rng);
    list[1] = new XZRect(-300, 300, -300, 300, 400, new DiffuseLight(new ConstantTexture(Vector3(6, 6, 6))));
    Built b;
    b.world = new HitableList(list, 2);
    b.cam = look(Vector3(0, 260, -760), Vector3(0, 20, 0), 40.f, width, height, 0.f, 10.f);
    return b;
}

/* one of everything, for known-answer tests */
Built kat(int width, int height)
{
    int tw, th;
    const std::vector<unsigned char>& img = procedural_map(tw, th);
    Texture* checker = new CheckerTexture(new ConstantTexture(Vector3(.2f, .3f, .1f)), new ConstantTexture(Vector3(.9f, .9f, .9f)));
    Material* lam = new Lambert(new ConstantTexture(Vector3(.5, .6, .7)));
    Material* lamc = new Lambert(checker);
    Material* lami = new Lambert(new ImageTexture(const_cast<unsigned char*>(img.data()), tw, th));
    Material* met = new Metal(new ConstantTexture(Vector3(.8, .7, .6)), .3f);
    Material* met2 = new Metal(new ConstantTexture(Vector3(.8, .8, .9)), 10.f);
    Material* gls = new Glass(1.5f);
    Material* lgt = new DiffuseLight(new ConstantTexture(Vector3(4, 3, 2)));
    Hitable** list = ptr_array(24);
    int i = 0;
    list[i++] = new Sphere(Vector3(0, 0, 0), 1.f, lami);
    list[i++] = new Sphere(Vector3(3, 0.5f, -1), -0.8f, gls);          /* negative radius flips the normal */
    list[i++] = new MovingSphere(Vector3(-3, 0, 1), Vector3(-3, 1, 1.5f), 0.f, 1.f, .7f, lam);
    list[i++] = new XYRect(-2, 2, -1, 3, -4, lamc);
    list[i++] = new XZRect(-5, 5, -5, 5, -1.5f, lamc);
    list[i++] = new YZRect(-1, 2, -3, 3, 6, met);
    list[i++] = new FlipNormals(new XYRect(-6, 6, -2, 5, 8, lgt));
    list[i++] = new Box(Vector3(-6, -1, -3), Vector3(-4.5f, 1, -1), met2);
    list[i++] = new Translate(new RotateY(new Box(Vector3(0, 0, 0), Vector3(1.5f, 2, 1), lam), 33), Vector3(1.5f, -1.4f, 2.5f));
    list[i++] = new RotateY(new Translate(new Sphere(Vector3(0, 0, 0), .6f, met), Vector3(4, 1.5f, 0)), -50);
    list[i++] = new Triangle(Vector3(-1, 2, -2), Vector3(1, 2.2f, -2.5f), Vector3(0, 3.5f, -1.5f), lam);
    list[i++] = new Translate(new Triangle(Vector3(0, 0, 0), Vector3(2, 0, .5f), Vector3(.5f, 1.5f, 0), met), Vector3(-5, 2, 3));
    curandState unused;
    curand_init(419, 0, 0, &unused);
    list[i++] = new ConstantMedium(new Sphere(Vector3(-1.5f, 1.f, 4.f), 1.2f, gls), .8f, new ConstantTexture(Vector3(.9, .5, .3)), &unused);
    Built b;
    b.world = new HitableList(list, i);
    b.cam = look(Vector3(2, 3, -12), Vector3(0, .5f, 0), 45.f, width, height, 0.05f, 12.f);
    return b;
}

/* "The Next Week" ch. 5-6: two marble spheres under a light sphere and a light rect.  Not one of the
 * reference's scenes (it has no noise class); exercises NoiseTexture under Lambert, Metal and a checker. */
Built perlin_spheres(int width, int height)
{
    curandState rng;
    curand_init(419, 0, 0, &rng);
    Texture* marble = new NoiseTexture(4.f, &rng);
    Texture* fine = new NoiseTexture(11.f, &rng);
    Texture* mixed = new CheckerTexture(fine, new ConstantTexture(Vector3(.8f, .3f, .2f)));
    Material* light = new DiffuseLight(new ConstantTexture(Vector3(4, 4, 4)));
    Hitable** list = ptr_array(8);
    int i = 0;
    list[i++] = new Sphere(Vector3(0, -1000, 0), 1000, new Lambert(marble));
    list[i++] = new Sphere(Vector3(0, 2, 0), 2, new Lambert(marble));
    list[i++] = new Sphere(Vector3(0, 7, 0), 2, light);
    list[i++] = new XYRect(3, 5, 1, 3, -2, light);
    list[i++] = new Sphere(Vector3(2.5f, 1, 3.5f), 1, new Metal(fine, .05f));
    list[i++] = new Translate(new RotateY(new Box(Vector3(0, 0, 0), Vector3(1.5f, 1.5f, 1.5f), new Lambert(mixed)), 25), Vector3(3, 0, -4.5f));
    Built b;
    b.world = new HitableList(list, i);
    b.cam = look(Vector3(26, 3, 6), Vector3(0, 2, 0), 20.f, width, height, 0.f, 10.f);
    return b;
}

Built build_named(const std::string& name, int width, int height, int param, bool& known)
{
    known = true;
    if (name == "perlin") return perlin_spheres(width, height);
    if (name == "cornell") return cornell(width, height, false);
    if (name == "cornell_smoke") return cornell(width, height, true);
    if (name == "random_spheres") return random_spheres(width, height);
    if (name == "lit_sphere") return lit_sphere(width, height);
    if (name == "bvh_boxes") return bvh_boxes(width, height);
    if (name == "cornell_cluster") return cornell_cluster(width, height);
    if (name == "final") return final_scene(width, height, param);
    if (name == "final_custom") return final_custom(width, height, param);
    if (name == "tri_mesh") return tri_mesh(width, height, param);
    if (name == "kat") return kat(width, height);
    known = false;
    return Built();
}

} // namespace

extern "C" int rt_scene_create_builtin(const char* name, int width, int height, int param, rt_scene** out)
{
    if (!name || !out || width <= 0 || height <= 0) { rt_set_error("rt_scene_create_builtin: bad arguments"); return RT_ERR_ARG; }
    rt::Arena arena; /* owns every node the scene function news; freed on return */
    bool known = false;
    Built b = build_named(name, width, height, param, known);
    if (!known) { rt_set_error(std::string("unknown built-in scene: ") + name); return RT_ERR_ARG; }
    std::unique_ptr<rt_scene> s(new rt_scene());
    std::string err;
    if (!rt::lower(b.world, b.cam, rt::LowerOptions(), s->flat, err)) { rt_set_error("lowering failed: " + err); return RT_ERR_UNSUPPORTED; }
    s->background = b.background;
    *out = s.release();
    return RT_OK;
}

extern "C" int rt_scene_from_flat(const rt_flat_scene* f, rt_scene** out)
{
    if (!f || !out) { rt_set_error("rt_scene_from_flat: null argument"); return RT_ERR_ARG; }
    std::string why;
    if (!rt_validate_scene(f, why, nullptr)) { rt_set_error("rt_scene_from_flat: invalid scene: " + why); return RT_ERR_ARG; }
    rt_scene* s = new rt_scene();
    s->flat.objects.assign(f->objects, f->objects + f->n_objects);
    s->flat.ops.assign(f->ops, f->ops + f->n_ops);
    s->flat.materials.assign(f->materials, f->materials + f->n_materials);
    s->flat.textures.assign(f->textures, f->textures + f->n_textures);
    s->flat.image_bytes.assign(f->image_bytes, f->image_bytes + f->n_image_bytes);
    s->flat.bvh_nodes.assign(f->bvh_nodes, f->bvh_nodes + f->n_bvh_nodes);
    s->flat.bvh_items.assign(f->bvh_items, f->bvh_items + f->n_bvh_items);
    s->flat.view = *f;
    s->flat.refresh_view();
    *out = s;
    return RT_OK;
}

extern "C" const rt_flat_scene* rt_scene_flat(const rt_scene* s) { return s ? &s->flat.view : nullptr; }
extern "C" int rt_scene_default_background(const rt_scene* s) { return s ? s->background : RT_BG_BLACK; }
extern "C" void rt_scene_destroy(rt_scene* s) { delete s; }
