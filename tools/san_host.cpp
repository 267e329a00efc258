/*
 * san_host.cpp — the host half of the library (scene API recorders, lowering, BVH builder, validation, packing, image IO)
 * under AddressSanitizer + UndefinedBehaviorSanitizer (the reference has no sanitizer runs at all).
 * Built from csrc/lower.cpp, csrc/scenes.cpp and csrc/rt_host.cpp only: no HIP, no GPU.  `make -C cuda-raytracer_amd san`
 * builds and runs it; tests/test_sanitize.py does the same.  Exit code 0 = every check held and no sanitizer report.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../include/rt_api.h"

#define REQUIRE(x) do { if (!(x)) { std::fprintf(stderr, "san_host: %s failed at line %d: %s\n", #x, __LINE__, rt_last_error()); return 1; } } while (0)

int main()
{
    const char* names[] = {"cornell", "cornell_smoke", "random_spheres", "lit_sphere", "bvh_boxes", "cornell_cluster", "final",
                           "final_custom", "tri_mesh", "kat", "perlin"};
    for (const char* n : names) {
        rt_scene* s = nullptr;
        REQUIRE(rt_scene_create_builtin(n, 64, 48, !std::strcmp(n, "tri_mesh") ? 20000 : 0, &s) == RT_OK);
        const rt_flat_scene* f = rt_scene_flat(s);
        REQUIRE(f && f->n_objects > 0);
        (void)rt_flat_features(f);
        std::vector<uint32_t> words(8 * 4096);
        std::vector<int32_t> groups(64);
        int n_groups = 0;
        REQUIRE(rt_flat_program(f, words.data(), 4096) > 0);
        REQUIRE(rt_flat_program_split(f, words.data(), 4096, groups.data(), 64, &n_groups) >= 0);
        int a = 0, b = 0, c = 0, d = 0;
        REQUIRE(rt_flat_bvh_stats(f, &a, &b, &c, &d) == RT_OK);
        rt_scene* copy = nullptr;
        REQUIRE(rt_scene_from_flat(f, &copy) == RT_OK);
        rt_scene_destroy(copy);
        /* corrupted tables must be refused, not read out of bounds */
        rt_flat_scene bad = *f;
        std::vector<rt_object> objs(f->objects, f->objects + f->n_objects);
        bad.objects = objs.data();
        for (rt_object& o : objs) if (o.type != RT_OBJ_GROUP) { o.material = 1 << 20; break; }
        REQUIRE(rt_scene_from_flat(&bad, &copy) < 0);
        objs.assign(f->objects, f->objects + f->n_objects);
        objs[(size_t)f->world].child_count = 1 << 28;
        REQUIRE(rt_scene_from_flat(&bad, &copy) < 0);
        if (f->n_bvh_items > 0) {
            objs.assign(f->objects, f->objects + f->n_objects);
            std::vector<int32_t> items(f->bvh_items, f->bvh_items + f->n_bvh_items);
            items[0] = -5;
            bad.bvh_items = items.data();
            REQUIRE(rt_scene_from_flat(&bad, &copy) < 0);
            std::vector<rt_bvh_node> nodes(f->bvh_nodes, f->bvh_nodes + f->n_bvh_nodes);
            bad.bvh_items = f->bvh_items;
            bad.bvh_nodes = nodes.data();
            nodes[0].left_first = 0x7fffffffu;
            REQUIRE(rt_scene_from_flat(&bad, &copy) < 0);
        }
        rt_scene_destroy(s);
    }
    /* scenes built from several threads at once (rt_render builds one per GPU thread) */
    std::vector<std::thread> pool;
    std::vector<int> rc(6, -1);
    for (int t = 0; t < 6; ++t)
        pool.emplace_back([&rc, t] { rt_scene* s = nullptr; rc[(size_t)t] = rt_scene_create_builtin(t & 1 ? "final" : "random_spheres", 32, 32, 4, &s); rt_scene_destroy(s); });
    for (std::thread& t : pool) t.join();
    for (int r : rc) REQUIRE(r == RT_OK);
    /* image output and input */
    const int W = 5, H = 3;
    std::vector<float> rgb((size_t)W * H * 3);
    for (size_t i = 0; i < rgb.size(); ++i) rgb[i] = -0.5f + 0.11f * (float)i;      /* below 0 and above 1: clamped */
    std::vector<uint8_t> px((size_t)W * H * 3);
    REQUIRE(rt_quantise_rgb8(rgb.data(), W, H, px.data()) == RT_OK);
    REQUIRE(rt_write_ppm("/tmp/san_host.ppm", px.data(), W, H) == RT_OK);
    REQUIRE(rt_write_png("/tmp/san_host.png", px.data(), W, H) == RT_OK);
    uint8_t* back = nullptr;
    int bw = 0, bh = 0;
    REQUIRE(rt_read_ppm("/tmp/san_host.ppm", &back, &bw, &bh) == RT_OK && bw == W && bh == H && !std::memcmp(back, px.data(), px.size()));
    rt_free_pixels(back);
    FILE* f = std::fopen("/tmp/san_host_bad.ppm", "wb");
    std::fputs("P6\n70000 70000\n255\nxx", f);
    std::fclose(f);
    REQUIRE(rt_read_ppm("/tmp/san_host_bad.ppm", &back, &bw, &bh) < 0);
    REQUIRE(rt_read_ppm("/nonexistent/x.ppm", &back, &bw, &bh) < 0);
    std::puts("san_host ok");
    return 0;
}
