/*
 * bvh_sim.cpp — node visits and primitive tests per ray for the device trees of a built-in scene, walked on the host
 * (a measurement aid for rt_wavefront8.h; nothing links it).  Compares, on the same rays (primary rays + one random
 * bounce from every primary hit):
 *   A  four-wide nodes, children sorted by entry distance, leaves tested at once          (wf_intersect_kernel's order)
 *   B  eight-wide quantised nodes, octant order, leaves tested at once
 *   C  eight-wide quantised nodes, children sorted by entry distance, leaves tested at once (what the octant order loses)
 *   D  as B, but a leaf's items are only tested LAG node steps after they were met           (queued leaves: a stale bound)
 * build: g++ -O2 -std=c++17 -I cuda-raytracer_amd/api tools/sim/bvh_sim.cpp cuda-raytracer_amd/csrc/{lower,scenes,rt_host}.cpp -o /tmp/bvh_sim -lpthread
 */
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../include/rt_api.h"
#include "../../cuda-raytracer_amd/csrc/rt_host.h"

struct Ray { float o[3], d[3]; };
struct Tri { float f[13]; };
static const rt_flat_scene* F;
static std::vector<DObj> objs;
static std::vector<DNode4> n4;
static std::vector<DNode8> n8;
static std::vector<DLeaf> l8;
static std::vector<DObj> leafs;          /* item order */
static std::vector<int> level_of;        /* depth of each four-wide node, root = 0 */

static bool tri_hit(const float* f, const Ray& r, float tmin, float tmax, float& t)
{
    const float* v0 = f; const float* v1 = f + 3; const float* v2 = f + 6; const float* n = f + 9;
    const float dn = r.d[0] * n[0] + r.d[1] * n[1] + r.d[2] * n[2];
    const float dist = (-(r.o[0] * n[0] + r.o[1] * n[1] + r.o[2] * n[2]) + f[12]) / dn;
    if (!(dist > FLT_EPSILON)) return false;
    float p[3] = {r.o[0] + dist * r.d[0], r.o[1] + dist * r.d[1], r.o[2] + dist * r.d[2]};
    auto side = [&](const float* a, const float* b) {
        float e[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, w[3] = {p[0] - a[0], p[1] - a[1], p[2] - a[2]};
        float c[3] = {e[1] * w[2] - e[2] * w[1], -(e[0] * w[2] - e[2] * w[0]), e[0] * w[1] - e[1] * w[0]};
        return c[0] * n[0] + c[1] * n[1] + c[2] * n[2];
    };
    if (!(side(v0, v1) > 0 && side(v1, v2) > 0 && side(v2, v0) > 0)) return false;
    if (!(dist < tmax && dist > tmin)) return false;
    t = dist;
    return true;
}
static bool item_hit(const DObj& row, const Ray& r, float tmax, float& t)
{
    if ((row.head & 0xffu) != RT_OBJ_TRIANGLE) return false;      /* the light rect: ignored by the count */
    return tri_hit(row.f, r, .001f, tmax, t);
}
static bool slab(const DNode4& n, int child, const Ray& r, const float* id, float tmax, float& tn)      /* DNode4: planes axis-major */
{
    float a = .001f, b = tmax;
    for (int k = 0; k < 3; ++k) {
        float t0 = (n.plane[k][0][child] - r.o[k]) * id[k], t1 = (n.plane[k][1][child] - r.o[k]) * id[k];
        if (t0 > t1) std::swap(t0, t1);
        a = std::max(a, t0); b = std::min(b, t1);
    }
    tn = a;
    return a <= b * 1.0000012f;
}
struct Count { double nodes = 0, prims = 0, rays = 0, hits = 0; };

static double visits_below[4] = {0, 0, 0, 0};      /* four-wide node visits in the top 1, 2, 3, 4 levels of the tree (1, 5, 21, 85 nodes) */
static float walk4(int root, const Ray& r, Count& c)
{
    float id[3] = {1.f / r.d[0], 1.f / r.d[1], 1.f / r.d[2]}, best = FLT_MAX;
    std::vector<int> st{root};
    while (!st.empty()) {
        int ref = st.back(); st.pop_back();
        if (ref >= 0 && ref != RT_REF_NONE) for (int q = 0; q < 4; ++q) if (level_of[(size_t)ref] <= q) visits_below[q] += 1;
        if (ref == RT_REF_NONE) continue;
        if (ref < 0) {
            uint32_t first = (uint32_t)(~ref) >> 4, cnt = ((uint32_t)(~ref) & 15u) + 1u;
            for (uint32_t j = 0; j < cnt; ++j) { float t; c.prims += 1; if (item_hit(leafs[first + j], r, best, t)) best = t; }
            continue;
        }
        c.nodes += 1;
        const DNode4& n = n4[(size_t)ref];
        std::pair<float, int> ch[4]; int m = 0;
        for (int k = 0; k < 4; ++k) { float tn; if (n.ref[k] != RT_REF_NONE && slab(n, k, r, id, best, tn)) ch[m++] = {tn, n.ref[k]}; }
        std::sort(ch, ch + m);
        for (int k = m - 1; k >= 0; --k) st.push_back(ch[k].second);
    }
    return best;
}
/* mode 0: octant order; 1: sorted by entry distance.  lag: node steps between meeting an item and testing it */
static float walk8(int root, const Ray& r, Count& c, int mode, int lag)
{
    float id[3] = {1.f / r.d[0], 1.f / r.d[1], 1.f / r.d[2]}, best = FLT_MAX;
    const uint32_t oct = (r.d[0] < 0 ? 1u : 0u) | (r.d[1] < 0 ? 2u : 0u) | (r.d[2] < 0 ? 4u : 0u);
    int rank[8];      /* mode 2: slots in the order of their octant corner's projection on the ray's direction */
    {
        float dn[3]; const float len = std::sqrt(r.d[0] * r.d[0] + r.d[1] * r.d[1] + r.d[2] * r.d[2]);
        for (int a = 0; a < 3; ++a) dn[a] = r.d[a] / len;
        float dot[8];
        for (int s = 0; s < 8; ++s) dot[s] = ((s & 1) ? dn[0] : -dn[0]) + ((s & 2) ? dn[1] : -dn[1]) + ((s & 4) ? dn[2] : -dn[2]);
        for (int s = 0; s < 8; ++s) { rank[s] = 0; for (int q = 0; q < 8; ++q) if (dot[q] < dot[s] || (dot[q] == dot[s] && q < s)) ++rank[s]; }
    }
    std::vector<int> st{root};
    std::vector<std::pair<long, uint32_t>> pend;     /* (due step, row) */
    long step = 0;
    auto flush = [&](long now) {
        size_t k = 0;
        for (; k < pend.size() && pend[k].first <= now; ++k) {
            float t; c.prims += 1;
            DObj row; std::memcpy(&row, &l8[pend[k].second], sizeof(DLeaf));
            if (item_hit(row, r, best, t)) best = t;
        }
        pend.erase(pend.begin(), pend.begin() + (long)k);
    };
    while (!st.empty()) {
        const int ni = st.back(); st.pop_back();
        ++step;
        flush(step);
        c.nodes += 1;
        const DNode8& n = n8[(size_t)ni];
        const uint32_t imask = n.e_imask >> 24, lmask = n.leaf_masks & 0xffu, l2 = (n.leaf_masks >> 8) & 0xffu;
        struct Ch { float tn; int prio; int node; }; Ch ch[8]; int m = 0, below = 0;
        for (int s = 0; s < 8; ++s) {
            const bool inner = (imask >> s) & 1u, leaf = (lmask >> s) & 1u;
            const int child = inner ? (int)n.child_base + below : -1;
            if (inner) ++below;
            if (!inner && !leaf) continue;
            float lo[3], hi[3];
            for (int a = 0; a < 3; ++a) {
                const float sc = std::ldexp(1.f, (int)((n.e_imask >> (8 * a)) & 0xffu) - 127);
                lo[a] = n.p[a] + sc * (float)((n.qlo[a][s >> 2] >> (8 * (s & 3))) & 0xffu);
                hi[a] = n.p[a] + sc * (float)((n.qhi[a][s >> 2] >> (8 * (s & 3))) & 0xffu);
            }
            float tn;
            if (!slab(lo, hi, r, id, best, tn)) continue;
            if (leaf) {
                pend.push_back({step + lag, (uint32_t)(16 * ni + s)});
                if ((l2 >> s) & 1u) pend.push_back({step + lag, (uint32_t)(16 * ni + 8 + s)});
            } else ch[m++] = {tn, (int)((uint32_t)s ^ oct), child};
        }
        if (lag == 0) flush(step);
        if (mode == 2) for (int k = 0; k < m; ++k) ch[k].prio = rank[ch[k].prio ^ (int)oct];      /* prio held slot ^ oct: back to the slot */
        if (mode == 0 || mode == 2) std::sort(ch, ch + m, [](const Ch& a, const Ch& b) { return a.prio < b.prio; });
        else std::sort(ch, ch + m, [](const Ch& a, const Ch& b) { return a.tn < b.tn; });
        for (int k = m - 1; k >= 0; --k) st.push_back(ch[k].node);
    }
    flush(LONG_MAX);
    return best;
}

int main(int argc, char** argv)
{
    const int n_tri = argc > 1 ? std::atoi(argv[1]) : 200000, W = 160, H = 90;
    rt_scene* sc = nullptr;
    if (rt_scene_create_builtin("tri_mesh", W, H, n_tri, &sc) != RT_OK) { std::fprintf(stderr, "%s\n", rt_last_error()); return 1; }
    F = rt_scene_flat(sc);
    objs.resize((size_t)F->n_objects);
    rt_pack_objects(F, objs.data());
    std::vector<DNode> n2; int d2 = 0, d4 = 0, d8 = 0; std::string why; float amax = 0;
    rt_pack_bvh(F, objs, n2, leafs, &d2, why);
    rt_pack_bvh4(F, objs, n4, &d4);
    if (!rt_pack_bvh8(F, objs, n8, l8, &d8, &amax)) { std::fprintf(stderr, "no 8-wide tree\n"); return 1; }
    int group = -1;
    for (int i = 0; i < F->n_objects; ++i) if (F->objects[i].type == RT_OBJ_GROUP && F->objects[i].bvh_root >= 0) group = i;
    const int root4 = objs[(size_t)group].pad[0], root8 = objs[(size_t)group].pad[1];
    std::printf("tri_mesh %d: nodes4 %zu (depth %d), nodes8 %zu (depth %d)\n", n_tri, n4.size(), d4, n8.size(), d8);
    level_of.assign(n4.size(), 99);
    {
        std::vector<std::pair<int, int>> st{{root4, 0}};
        while (!st.empty()) {
            auto [ni, lv] = st.back(); st.pop_back();
            if (ni < 0 || ni == RT_REF_NONE) continue;
            level_of[(size_t)ni] = lv;
            for (int k = 0; k < 4; ++k) st.push_back({n4[(size_t)ni].ref[k], lv + 1});
        }
    }
    /* rays */
    std::vector<Ray> rays;
    const rt_camera& cam = F->camera;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    Count dummy;
    for (int j = 0; j < H; ++j) for (int i = 0; i < W; ++i) {
        Ray r; const float u = (i + .5f) / W, v = (j + .5f) / H;
        for (int a = 0; a < 3; ++a) { r.o[a] = cam.origin[a]; r.d[a] = cam.lower_left[a] + u * cam.horizontal[a] + v * cam.vertical[a] - cam.origin[a]; }
        rays.push_back(r);
        const float t = walk4(root4, r, dummy);
        if (t < FLT_MAX) for (int b = 0; b < 3; ++b) {        /* three random bounce rays from the hit point */
            Ray q; float len2;
            do { len2 = 0; for (int a = 0; a < 3; ++a) { q.d[a] = U(rng); len2 += q.d[a] * q.d[a]; } } while (len2 > 1.f || len2 < 1e-4f);
            for (int a = 0; a < 3; ++a) q.o[a] = r.o[a] + t * r.d[a];
            rays.push_back(q);
        }
    }
    auto report = [&](const char* name, Count c) { std::printf("%-44s nodes/ray %6.2f  prims/ray %6.2f\n", name, c.nodes / rays.size(), c.prims / rays.size()); };
    Count a, b, c, d1, d3, d6, e0, e3; double diff = 0;
    for (double& v : visits_below) v = 0;      /* the pass that made the bounce rays does not count */
    for (const Ray& r : rays) {
        const float ta = walk4(root4, r, a), tb = walk8(root8, r, b, 0, 0);
        walk8(root8, r, c, 1, 0); walk8(root8, r, e0, 2, 0); walk8(root8, r, e3, 2, 3); walk8(root8, r, d1, 0, 1); walk8(root8, r, d3, 0, 3); walk8(root8, r, d6, 0, 6);
        if (ta != tb) diff += 1;
    }
    std::printf("%zu rays, %g with another t between the two trees\n", rays.size(), diff);
    report("A four-wide, sorted, leaves at once", a);
    std::printf("   of those, in the top 1 / 2 / 3 / 4 levels (1 / 5 / 21 / 85 nodes): %.2f / %.2f / %.2f / %.2f per ray\n",
                visits_below[0] / rays.size(), visits_below[1] / rays.size(), visits_below[2] / rays.size(), visits_below[3] / rays.size());
    report("B eight-wide, octant order, leaves at once", b);
    report("C eight-wide, sorted, leaves at once", c);
    report("E eight-wide, corner-projection order, at once", e0);
    report("E eight-wide, corner-projection order, 3 late", e3);
    report("D eight-wide, octant order, leaves 1 step late", d1);
    report("D eight-wide, octant order, leaves 3 steps late", d3);
    report("D eight-wide, octant order, leaves 6 steps late", d6);
    return 0;
}
