/*
 * rt_host.cpp — error string, scene validation, object packing, image output.
 */
#include "rt_host.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <vector>
#include "../../include/rt_api.h"

/* what a node step and a primitive test cost against each other when rt_pack_bvh8 chooses its children */
#ifndef RT_BVH8_C_NODE
#define RT_BVH8_C_NODE 1.0f
#endif
#ifndef RT_BVH8_C_PRIM
#define RT_BVH8_C_PRIM 0.3f
#endif

static thread_local std::string g_err;
void rt_set_error(const std::string& msg) { g_err = msg; }
extern "C" const char* rt_last_error(void) { return g_err.c_str(); }

static int bvh_depth_of(const rt_flat_scene* s, int root, int limit, bool& ok, std::vector<uint8_t>& seen)
{
    /* iterative DFS; every node may be reached once (a DAG or cycle is refused) */
    struct E { int node, depth; };
    std::vector<E> st;
    st.push_back({root, 1});
    int deepest = 0;
    while (!st.empty()) {
        E e = st.back();
        st.pop_back();
        if (e.node < 0 || e.node >= s->n_bvh_nodes || seen[e.node] || e.depth > limit) { ok = false; return 0; }
        seen[e.node] = 1;
        if (e.depth > deepest) deepest = e.depth;
        const rt_bvh_node& n = s->bvh_nodes[e.node];
        if (n.count) {
            if ((uint64_t)n.left_first + n.count > (uint64_t)s->n_bvh_items) { ok = false; return 0; }
        } else {
            if ((uint64_t)n.left_first + 1 >= (uint64_t)s->n_bvh_nodes) { ok = false; return 0; }
            st.push_back({(int)n.left_first, e.depth + 1});
            st.push_back({(int)n.left_first + 1, e.depth + 1});
        }
    }
    return deepest;
}

bool rt_validate_scene(const rt_flat_scene* s, std::string& why, int* bvh_depth)
{
    auto bad = [&](const char* m) { why = m; return false; };
    if (bvh_depth) *bvh_depth = 0;
    if (s->n_objects < 1 || !s->objects) return bad("no objects");
    if (s->world < 0 || s->world >= s->n_objects || s->objects[s->world].type != RT_OBJ_GROUP) return bad("world is not a group");
    if ((s->n_ops && !s->ops) || (s->n_materials && !s->materials) || (s->n_textures && !s->textures) ||
        (s->n_image_bytes && !s->image_bytes) || (s->n_bvh_nodes && !s->bvh_nodes) || (s->n_bvh_items && !s->bvh_items))
        return bad("null table");
    for (int i = 0; i < s->n_textures; ++i) {
        const rt_texture& t = s->textures[i];
        if (t.type == RT_TEX_CHECKER) {
            if (t.even < 0 || t.even >= s->n_textures || t.odd < 0 || t.odd >= s->n_textures) return bad("checker child out of range");
        } else if (t.type == RT_TEX_IMAGE) {
            if (t.width <= 0 || t.height <= 0) return bad("image texture without extent");
            if (t.image_offset + (uint64_t)t.width * (uint64_t)t.height * 3u > s->n_image_bytes) return bad("image texture out of range");
        } else if (t.type == RT_TEX_NOISE) {
            if (t.width < 1 || t.width > 16) return bad("noise texture octaves out of range");
            if ((t.image_offset & 3u) || t.image_offset + RT_NOISE_TABLE_BYTES > s->n_image_bytes) return bad("noise texture table out of range");
        } else if (t.type != RT_TEX_CONSTANT) return bad("unknown texture type");
    }
    for (int i = 0; i < s->n_materials; ++i) {
        const rt_material& m = s->materials[i];
        if (m.type < RT_MAT_LAMBERT || m.type > RT_MAT_ISOTROPIC) return bad("unknown material type");
        if (m.type != RT_MAT_GLASS && (m.texture < 0 || m.texture >= s->n_textures)) return bad("material texture out of range");
    }
    for (int i = 0; i < s->n_ops; ++i)
        if (s->ops[i].type != RT_OP_TRANSLATE && s->ops[i].type != RT_OP_ROTATE_Y) return bad("unknown transform op");
    /* device leaf references are ~((first item << 4) | (count - 1)) in an int32 (rt_pack_bvh, rt_pack_bvh4): the first
     * item needs to stay below 2^27 or the reference turns non-negative and reads as an interior node */
    if (s->n_bvh_items >= (1 << 27)) return bad("more than 2^27 - 1 BVH items");
    /* the scan program and rebuild() treat the world list's frame as the world frame: a wrapper chain on the world
     * object itself would be traversed in one frame and shaded in another (rt::lower never produces one) */
    if (s->objects[s->world].op_count != 0) return bad("the world object carries a transform chain");
    for (int i = 0; i < s->n_bvh_items; ++i) {
        int oi = s->bvh_items[i];
        if (oi < 0 || oi >= s->n_objects) return bad("BVH item out of range");
        int ty = s->objects[oi].type;
        if (ty == RT_OBJ_GROUP || ty == RT_OBJ_MEDIUM) return bad("BVH item is not a surface");
    }
    std::vector<uint8_t> seen((size_t)s->n_bvh_nodes, 0);
    int deepest = 0;
    for (int i = 0; i < s->n_objects; ++i) {
        const rt_object& o = s->objects[i];
        if (o.type < RT_OBJ_SPHERE || o.type > RT_OBJ_GROUP) return bad("unknown object type");
        if (o.op_count < 0 || o.op_begin < 0 || (int64_t)o.op_begin + o.op_count > s->n_ops) return bad("op chain out of range");
        if (o.op_count > 8) return bad("op chain longer than 8");
        if (o.type == RT_OBJ_GROUP) {
            if (o.child_count < 0 || o.child_begin < 0 || (int64_t)o.child_begin + o.child_count > s->n_objects) return bad("group children out of range");
            for (int k = 0; k < o.child_count; ++k) {
                const rt_object& ch = s->objects[o.child_begin + k];
                if (ch.type == RT_OBJ_GROUP && i != s->world) return bad("group nested below a group");
            }
            if (o.bvh_root >= 0) {
                bool ok = true;
                int d = bvh_depth_of(s, o.bvh_root, 4096, ok, seen);
                if (!ok) return bad("malformed BVH");
                if (d > deepest) deepest = d;
            }
        } else {
            if (o.material < 0 || o.material >= s->n_materials) return bad("object material out of range");
            if (o.type == RT_OBJ_MEDIUM) {
                if (o.child_begin < 0 || o.child_begin >= s->n_objects) return bad("medium boundary out of range");
                int bt = s->objects[o.child_begin].type;
                if (bt == RT_OBJ_GROUP || bt == RT_OBJ_MEDIUM) return bad("medium boundary is not a surface");
                if (o.medium_slot < 0) return bad("medium without a slot");
            }
        }
    }
    if (bvh_depth) *bvh_depth = deepest;
    return true;
}

void rt_pack_objects(const rt_flat_scene* s, DObj* out)
{
    for (int i = 0; i < s->n_objects; ++i) {
        const rt_object& o = s->objects[i];
        DObj d;
        std::memset(&d, 0, sizeof d);
        uint32_t t0 = o.op_count > 0 ? (uint32_t)s->ops[o.op_begin].type : 0u;
        uint32_t t1 = o.op_count > 1 ? (uint32_t)s->ops[o.op_begin + 1].type : 0u;
        d.head = (uint32_t)o.type | ((o.flags & 0x7fu) << 8) | ((uint32_t)o.op_count << 16) | (t0 << 24) | (t1 << 28);
        d.material = o.material; d.prim_id = o.prim_id;
        d.op_begin = o.op_begin; d.child_begin = o.child_begin; d.child_count = o.child_count;
        d.bvh_root = RT_REF_NONE; d.medium_slot = o.medium_slot;
        if (o.op_count > 0) { const rt_xform_op& p = s->ops[o.op_begin];     d.op0[0] = p.a; d.op0[1] = p.b; d.op0[2] = p.c; }
        if (o.op_count > 1) { const rt_xform_op& p = s->ops[o.op_begin + 1]; d.op1[0] = p.a; d.op1[1] = p.b; d.op1[2] = p.c; }
        for (int k = 0; k < 10; ++k) d.f[k] = o.f[k];
        if (o.type == RT_OBJ_TRIANGLE) {
            const float* f = o.f;
            float e1[3] = {f[3] - f[0], f[4] - f[1], f[5] - f[2]};
            float e2[3] = {f[6] - f[0], f[7] - f[1], f[8] - f[2]};
            float c[3] = {(e1[1] * e2[2] - e1[2] * e2[1]), (-(e1[0] * e2[2] - e1[2] * e2[0])), (e1[0] * e2[1] - e1[1] * e2[0])};
            float len = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
            float n[3] = {c[0] / len, c[1] / len, c[2] / len};
            float z[3] = {0.f - f[0], 0.f - f[1], 0.f - f[2]};
            d.f[9] = n[0]; d.f[10] = n[1]; d.f[11] = n[2];
            d.f[12] = z[0] * n[0] + z[1] * n[1] + z[2] * n[2];
        }
        out[i] = d;
    }
}

bool rt_pack_bvh(const rt_flat_scene* s, std::vector<DObj>& objs, std::vector<DNode>& nodes,
                 std::vector<DObj>& leafs, int* depth, std::string& why)
{
    nodes.clear();
    leafs.assign((size_t)s->n_bvh_items, DObj());
    for (int j = 0; j < s->n_bvh_items; ++j) {
        leafs[j] = objs[(size_t)s->bvh_items[j]];
        leafs[j].prim_id = s->bvh_items[j];           /* object index: rebuild and reporting go through objs */
    }
    int deepest = 0;
    bool ok = true;
    auto child_ref = [&](uint32_t ni, std::vector<std::pair<uint32_t, int>>& todo) -> int32_t {
        const rt_bvh_node& c = s->bvh_nodes[ni];
        if (c.count) {
            if (c.count > RT_MAX_LEAF) { ok = false; return RT_REF_NONE; }
            return ~(int32_t)((c.left_first << 4) | (c.count - 1));
        }
        int32_t id = (int32_t)nodes.size();
        nodes.push_back(DNode());
        todo.push_back({ni, id});
        return id;
    };
    for (int gi = 0; gi < s->n_objects; ++gi) {
        const rt_object& g = s->objects[gi];
        if (g.type != RT_OBJ_GROUP || g.bvh_root < 0) continue;
        std::vector<std::pair<uint32_t, int>> todo;   /* (ABI interior node, DNode index) */
        std::vector<int> level;                        /* depth of each DNode of this BVH, by (index - base) */
        const int base = (int)nodes.size();
        int32_t root = child_ref((uint32_t)g.bvh_root, todo);
        if (!ok) { why = "BVH leaf larger than 16 items"; return false; }
        level.assign(nodes.size() - base, 1);
        while (!todo.empty()) {
            std::pair<uint32_t, int> e = todo.back();
            todo.pop_back();
            const rt_bvh_node& n = s->bvh_nodes[e.first];
            const rt_bvh_node& l = s->bvh_nodes[n.left_first];
            const rt_bvh_node& r = s->bvh_nodes[n.left_first + 1];
            size_t before = nodes.size();
            int32_t lr = child_ref(n.left_first, todo);
            int32_t rr = child_ref(n.left_first + 1, todo);
            if (!ok) { why = "BVH leaf larger than 16 items"; return false; }
            DNode& d = nodes[(size_t)e.second];
            for (int k = 0; k < 3; ++k) { d.lmin[k] = l.bmin[k]; d.lmax[k] = l.bmax[k]; d.rmin[k] = r.bmin[k]; d.rmax[k] = r.bmax[k]; }
            d.left = lr; d.right = rr; d.pad[0] = d.pad[1] = 0;
            int lv = level[(size_t)(e.second - base)];
            if (lv > deepest) deepest = lv;
            for (size_t q = before; q < nodes.size(); ++q) level.push_back(lv + 1);
        }
        objs[(size_t)gi].bvh_root = root;
        objs[(size_t)gi].head |= RT_DOBJ_HAS_BVH << 8;
    }
    if (depth) *depth = deepest;
    return true;
}

static bool obj_is_bvh_group(const DObj& o) { return (o.head & 0xffu) == RT_OBJ_GROUP && ((o.head >> 8) & RT_DOBJ_HAS_BVH); }

void rt_pack_bvh4(const rt_flat_scene* s, std::vector<DObj>& objs, std::vector<DNode4>& nodes4, int* depth)
{
    nodes4.clear();
    int deepest = 0;
    auto area = [&](uint32_t ni) {
        const rt_bvh_node& n = s->bvh_nodes[ni];
        float e0 = n.bmax[0] - n.bmin[0], e1 = n.bmax[1] - n.bmin[1], e2 = n.bmax[2] - n.bmin[2];
        return e0 * e1 + e1 * e2 + e2 * e0;
    };
    struct Todo { uint32_t abi; int node4; int level; };
    for (int gi = 0; gi < s->n_objects; ++gi) {
        const rt_object& g = s->objects[gi];
        if (g.type != RT_OBJ_GROUP || g.bvh_root < 0) continue;
        std::vector<Todo> todo;
        auto child_ref = [&](uint32_t ni, int level) -> int32_t {
            const rt_bvh_node& c = s->bvh_nodes[ni];
            if (c.count) return ~(int32_t)((c.left_first << 4) | (c.count - 1));
            int32_t id = (int32_t)nodes4.size();
            nodes4.push_back(DNode4());
            todo.push_back({ni, id, level});
            return id;
        };
        objs[(size_t)gi].pad[0] = child_ref((uint32_t)g.bvh_root, 1);
        while (!todo.empty()) {
            Todo e = todo.back();
            todo.pop_back();
            if (e.level > deepest) deepest = e.level;
            const rt_bvh_node& n = s->bvh_nodes[e.abi];
            /* a child slot holds a node of the binary tree or, once a leaf has been opened up, ONE of its items */
            struct Cand { uint32_t node; int32_t item; };            /* item >= 0: index into bvh_items */
            Cand cands[4] = {{n.left_first, -1}, {n.left_first + 1, -1}, {0, -1}, {0, -1}};
            int nc = 2;
            while (nc < 4) {
                int pick = -1;
                float best = -1.f;
                for (int k = 0; k < nc; ++k)
                    if (cands[k].item < 0 && s->bvh_nodes[cands[k].node].count == 0 && area(cands[k].node) > best) { best = area(cands[k].node); pick = k; }
                if (pick < 0) break;
                const uint32_t lf = s->bvh_nodes[cands[pick].node].left_first;
                cands[pick].node = lf;
                cands[nc++].node = lf + 1;
            }
#ifndef RT_BVH4_KEEP_LEAVES
            /* Slots still free (the bottom of the tree: two leaf children and nothing left to open): a leaf of several
             * items is opened into one child per item, each behind its OWN box, so that a primitive is only fetched and
             * tested when the ray meets that primitive's box, not its neighbour's.  The node count does not change. */
            /* An opened item sits behind rt_object::bmin/bmax, which nothing else on the device reads: a caller's table
             * (rt_scene_from_flat) may have the node boxes right and the object bounds unset or stale.  A leaf is only
             * opened when every item's padded box is finite, ordered and inside the leaf's own box (what a builder that
             * made the node from the items guarantees, up to rounding); otherwise it stays one child behind the node's box. */
            auto openable = [&](const rt_bvh_node& leaf) {
                for (uint32_t q = 0; q < leaf.count; ++q) {
                    const rt_object& o = s->objects[s->bvh_items[leaf.left_first + q]];
                    for (int a = 0; a < 3; ++a) {
                        if (!(std::isfinite(o.bmin[a]) && std::isfinite(o.bmax[a]) && o.bmin[a] <= o.bmax[a])) return false;
                        const float m = std::max(std::fabs(o.bmin[a]), std::fabs(o.bmax[a])), slack = 4e-5f * m + 4e-4f;
                        if (o.bmin[a] < leaf.bmin[a] - slack || o.bmax[a] > leaf.bmax[a] + slack) return false;
                    }
                }
                return true;
            };
            while (nc < 4) {
                int pick = -1;
                uint32_t most = 1;
                for (int k = 0; k < nc; ++k)
                    if (cands[k].item < 0 && s->bvh_nodes[cands[k].node].count > most && (int)s->bvh_nodes[cands[k].node].count <= 5 - nc &&
                        openable(s->bvh_nodes[cands[k].node])) {
                        most = s->bvh_nodes[cands[k].node].count; pick = k;
                    }
                if (pick < 0) break;
                const rt_bvh_node& leaf = s->bvh_nodes[cands[pick].node];
                cands[pick].item = (int32_t)leaf.left_first;
                for (uint32_t q = 1; q < leaf.count; ++q) { cands[nc].node = cands[pick].node; cands[nc++].item = (int32_t)(leaf.left_first + q); }
            }
#endif
            int32_t refs[4];
            for (int k = 0; k < nc; ++k)
                refs[k] = cands[k].item >= 0 ? ~(int32_t)(((uint32_t)cands[k].item << 4) | 0u) : child_ref(cands[k].node, e.level + 1);
            DNode4& d = nodes4[(size_t)e.node4];      /* taken after child_ref: it may grow the vector */
            for (int k = 0; k < 4; ++k) {
                if (k < nc && cands[k].item >= 0) {
                    /* the item's own box, padded as the builder pads it (lower.cpp pad_box) */
                    const rt_object& o = s->objects[s->bvh_items[cands[k].item]];
                    for (int a = 0; a < 3; ++a) {
                        const float m = std::max(std::fabs(o.bmin[a]), std::fabs(o.bmax[a]));
                        float pad = 1e-5f * m + 1e-6f;
                        if (o.bmax[a] - o.bmin[a] < 2e-4f) pad += 1e-4f;
                        d.plane[a][0][k] = o.bmin[a] - pad; d.plane[a][1][k] = o.bmax[a] + pad;
                    }
                    d.ref[k] = refs[k];
                } else if (k < nc) {
                    const rt_bvh_node& c = s->bvh_nodes[cands[k].node];
                    for (int a = 0; a < 3; ++a) { d.plane[a][0][k] = c.bmin[a]; d.plane[a][1][k] = c.bmax[a]; }
                    d.ref[k] = refs[k];
                } else {
                    for (int a = 0; a < 3; ++a) { d.plane[a][0][k] = 3.0e38f; d.plane[a][1][k] = -3.0e38f; }
                    d.ref[k] = RT_REF_NONE;
                }
                d.pad[k] = 0;
            }
        }
    }
    if (depth) *depth = deepest;
}

bool rt_pack_bvh8(const rt_flat_scene* s, std::vector<DObj>& objs, std::vector<DNode8>& nodes8, std::vector<DLeaf>& leafs8, int* depth, float* box_abs)
{
    nodes8.clear();
    leafs8.clear();
    int deepest = 0;
    float amax = 0.f;
    struct Box { float lo[3], hi[3]; };
    auto node_box = [&](uint32_t ni) { Box b; const rt_bvh_node& n = s->bvh_nodes[ni]; for (int a = 0; a < 3; ++a) { b.lo[a] = n.bmin[a]; b.hi[a] = n.bmax[a]; } return b; };
    auto area = [&](uint32_t ni) {
        const rt_bvh_node& n = s->bvh_nodes[ni];
        float e0 = n.bmax[0] - n.bmin[0], e1 = n.bmax[1] - n.bmin[1], e2 = n.bmax[2] - n.bmin[2];
        return e0 * e1 + e1 * e2 + e2 * e0;
    };
    /* an item behind its own padded box (as rt_pack_bvh4 opens leaves), if its bounds can be trusted */
    auto item_box = [&](int32_t item, const rt_bvh_node& leaf, Box& b) {
        const rt_object& o = s->objects[s->bvh_items[item]];
        for (int a = 0; a < 3; ++a) {
            if (!(std::isfinite(o.bmin[a]) && std::isfinite(o.bmax[a]) && o.bmin[a] <= o.bmax[a])) return false;
            const float m = std::max(std::fabs(o.bmin[a]), std::fabs(o.bmax[a])), slack = 4e-5f * m + 4e-4f;
            if (o.bmin[a] < leaf.bmin[a] - slack || o.bmax[a] > leaf.bmax[a] + slack) return false;
            float pad = 1e-5f * m + 1e-6f;
            if (o.bmax[a] - o.bmin[a] < 2e-4f) pad += 1e-4f;
            b.lo[a] = o.bmin[a] - pad; b.hi[a] = o.bmax[a] + pad;
        }
        return true;
    };
    /* Which binary nodes become the children of an eight-wide node is chosen to minimise the surface-area cost of the
     * whole wide tree (Ylitie, Karras, Laine 2017, section 3.1): cost[n][i-1] = the cheapest way to hand the subtree of
     * binary node n to its ancestor as at most i children, where a child is a leaf slot (one item behind its own box, or
     * a binary leaf of two behind the leaf's box) or an eight-wide node (its box's area * c_node + the cost of its own
     * eight).  pick[n][i-1] = how many of the i go to the left child (0: use the answer for i - 1; for a leaf of two,
     * 1: open it into two slots).  A greedy top-down collapse leaves the bottom nodes of a balanced tree a third full. */
    const float c_node = RT_BVH8_C_NODE, c_prim = RT_BVH8_C_PRIM;
    struct Dp { float cost[8]; uint8_t pick[8]; };
    std::vector<Dp> dp((size_t)s->n_bvh_nodes);
    auto box_area = [](const Box& b) { const float e0 = b.hi[0] - b.lo[0], e1 = b.hi[1] - b.lo[1], e2 = b.hi[2] - b.lo[2]; return e0 * e1 + e1 * e2 + e2 * e0; };
    auto solve = [&](uint32_t root) -> bool {
        std::vector<std::pair<uint32_t, int>> st{{root, 0}};
        while (!st.empty()) {
            const uint32_t ni = st.back().first;
            const rt_bvh_node& n = s->bvh_nodes[ni];
            Dp& d = dp[ni];
            if (n.count) {
                st.pop_back();
                if (n.count > 2) return false;
                Box b0, b1;
                for (int i = 0; i < 8; ++i) { d.cost[i] = area(ni) * (float)n.count * c_prim; d.pick[i] = 0; }
                if (n.count == 1 && item_box((int32_t)n.left_first, n, b0)) for (int i = 0; i < 8; ++i) d.cost[i] = box_area(b0) * c_prim;
                if (n.count == 2 && item_box((int32_t)n.left_first, n, b0) && item_box((int32_t)n.left_first + 1, n, b1)) {
                    const float open = (box_area(b0) + box_area(b1)) * c_prim;
                    if (open < d.cost[0]) for (int i = 1; i < 8; ++i) { d.cost[i] = open; d.pick[i] = 1; }
                }
                continue;
            }
            if (st.back().second == 0) { st.back().second = 1; st.push_back({n.left_first, 0}); st.push_back({n.left_first + 1, 0}); continue; }
            st.pop_back();
            const Dp& l = dp[n.left_first];
            const Dp& r = dp[n.left_first + 1];
            auto distribute = [&](int j, uint8_t& k_best) {            /* j children in all, k of them from the left */
                float best = 3.0e38f;
                for (int k = 1; k < j; ++k) {
                    if (k > 7 || j - k > 7) continue;
                    const float c = l.cost[k - 1] + r.cost[j - k - 1];
                    if (c < best) { best = c; k_best = (uint8_t)k; }
                }
                return best;
            };
            d.cost[0] = distribute(8, d.pick[0]) + area(ni) * c_node;
            for (int i = 2; i <= 8; ++i) {
                uint8_t k = 0;
                const float c = distribute(i, k);
                if (c < d.cost[i - 2]) { d.cost[i - 1] = c; d.pick[i - 1] = k; } else { d.cost[i - 1] = d.cost[i - 2]; d.pick[i - 1] = 0; }
            }
        }
        return true;
    };
    struct Cand { uint32_t node; int32_t item; Box box; };      /* item >= 0: ONE item of the leaf `node`, behind its own box */
    struct Todo { uint32_t abi; uint32_t node8; int level; };
    /* the children the table chose for binary node ni given i slots; own = ni is the eight-wide node being filled */
    auto gather = [&](auto& self, uint32_t ni, int i, bool own, std::vector<Cand>& out) -> void {
        const rt_bvh_node& n = s->bvh_nodes[ni];
        const Dp& d = dp[ni];
        if (!own) {
            while (i > 1 && d.pick[i - 1] == 0) --i;
            if (i == 1) { out.push_back({ni, -1, node_box(ni)}); return; }
            if (n.count) {            /* a leaf of two, opened */
                Box b0, b1;
                item_box((int32_t)n.left_first, n, b0); item_box((int32_t)n.left_first + 1, n, b1);
                out.push_back({ni, (int32_t)n.left_first, b0});
                out.push_back({ni, (int32_t)n.left_first + 1, b1});
                return;
            }
        }
        const int k = own ? d.pick[0] : d.pick[i - 1];
        self(self, n.left_first, k, false, out);
        self(self, n.left_first + 1, (own ? 8 : i) - k, false, out);
    };
    for (int gi = 0; gi < s->n_objects; ++gi) {
        const rt_object& g = s->objects[gi];
        if (g.type != RT_OBJ_GROUP || g.bvh_root < 0) continue;
        if (!solve((uint32_t)g.bvh_root)) return false;
        std::vector<Todo> todo;
        objs[(size_t)gi].pad[1] = (int32_t)nodes8.size();
        nodes8.push_back(DNode8());
        todo.push_back({(uint32_t)g.bvh_root, (uint32_t)nodes8.size() - 1u, 1});
        while (!todo.empty()) {
            const Todo e = todo.back();
            todo.pop_back();
            if (e.level > deepest) deepest = e.level;
            const rt_bvh_node& n = s->bvh_nodes[e.abi];
            std::vector<Cand> cands;
            if (n.count) {            /* a tree that is one leaf */
                Box b0, b1;
                if (n.count == 2 && item_box((int32_t)n.left_first, n, b0) && item_box((int32_t)n.left_first + 1, n, b1)) {
                    cands.push_back({e.abi, (int32_t)n.left_first, b0});
                    cands.push_back({e.abi, (int32_t)n.left_first + 1, b1});
                } else cands.push_back({e.abi, -1, node_box(e.abi)});
            } else gather(gather, e.abi, 8, true, cands);
            /* the node's own box = the union of what it holds; slots by octant: the child whose centre lies furthest
             * towards an octant's corner takes that slot, greedily */
            Box nb = cands[0].box;
            for (const Cand& c : cands) for (int a = 0; a < 3; ++a) { nb.lo[a] = std::min(nb.lo[a], c.box.lo[a]); nb.hi[a] = std::max(nb.hi[a], c.box.hi[a]); }
            int slot_of[8], cand_at[8];
            for (int k = 0; k < 8; ++k) { slot_of[k] = -1; cand_at[k] = -1; }
            for (size_t round = 0; round < cands.size(); ++round) {
                double best = -1e300;
                int bc = -1, bs = -1;
                for (size_t k = 0; k < cands.size(); ++k) {
                    if (slot_of[k] >= 0) continue;
                    for (int sl = 0; sl < 8; ++sl) {
                        if (cand_at[sl] >= 0) continue;
                        double score = 0.0;
                        for (int a = 0; a < 3; ++a) {
                            const double rel = 0.5 * ((double)cands[k].box.lo[a] + cands[k].box.hi[a]) - 0.5 * ((double)nb.lo[a] + nb.hi[a]);
                            score += ((sl >> a) & 1) ? rel : -rel;
                        }
                        if (score > best) { best = score; bc = (int)k; bs = sl; }
                    }
                }
                slot_of[bc] = bs; cand_at[bs] = bc;
            }
            /* interior children get consecutive nodes in slot order */
            uint32_t imask = 0, lmask = 0, l2mask = 0;
            const uint32_t child_base = (uint32_t)nodes8.size();
            for (int sl = 0; sl < 8; ++sl) {
                if (cand_at[sl] < 0) continue;
                const Cand& c = cands[(size_t)cand_at[sl]];
                if (c.item < 0 && s->bvh_nodes[c.node].count == 0) {
                    imask |= 1u << sl;
                    nodes8.push_back(DNode8());
                    todo.push_back({c.node, (uint32_t)nodes8.size() - 1u, e.level + 1});
                } else {
                    lmask |= 1u << sl;
                    if (c.item < 0 && s->bvh_nodes[c.node].count == 2) l2mask |= 1u << sl;
                }
            }
            DNode8& d = nodes8[(size_t)e.node8];          /* taken after the push_backs: they may move the vector */
            std::memset(&d, 0, sizeof d);
            uint32_t ebytes = 0;
            double scale[3];
            for (int a = 0; a < 3; ++a) {
                d.p[a] = nb.lo[a];
                amax = std::max(amax, std::max(std::fabs(nb.lo[a]), std::fabs(nb.hi[a])));
                const double ext = (double)nb.hi[a] - (double)nb.lo[a];
                int ex = ext > 0.0 ? (int)std::ceil(std::log2(ext / 255.0)) : -100;
                ex = std::max(-100, std::min(100, ex));
                while (std::ldexp(255.0, ex) < ext) ++ex;
                scale[a] = std::ldexp(1.0, ex);
                ebytes |= (uint32_t)(ex + 127) << (8 * a);
            }
            d.e_imask = ebytes | (imask << 24);
            d.child_base = child_base;
            d.leaf_masks = lmask | (l2mask << 8);
            for (int sl = 0; sl < 8; ++sl) {
                for (int a = 0; a < 3; ++a) {
                    uint32_t ql = 255u, qh = 0u;            /* an empty slot: inverted */
                    if (cand_at[sl] >= 0) {
                        const Box& b = cands[(size_t)cand_at[sl]].box;
                        const double lo = std::floor(((double)b.lo[a] - (double)d.p[a]) / scale[a]), hi = std::ceil(((double)b.hi[a] - (double)d.p[a]) / scale[a]);
                        ql = (uint32_t)std::max(0.0, std::min(255.0, lo));
                        qh = (uint32_t)std::max(0.0, std::min(255.0, hi));
                    }
                    d.qlo[a][sl >> 2] |= ql << (8 * (sl & 3));
                    d.qhi[a][sl >> 2] |= qh << (8 * (sl & 3));
                }
            }
            /* the items of the leaf children: rows 16 * node + slot and 16 * node + 8 + slot */
            if (lmask) {
                const size_t base = 16 * (size_t)e.node8;
                if (leafs8.size() < base + 16) leafs8.resize(base + 16);
                for (int sl = 0; sl < 8; ++sl) {
                    if (!((lmask >> sl) & 1u)) continue;
                    const Cand& c = cands[(size_t)cand_at[sl]];
                    const int32_t first = c.item >= 0 ? c.item : (int32_t)s->bvh_nodes[c.node].left_first;
                    const int cnt = c.item >= 0 ? 1 : (int)s->bvh_nodes[c.node].count;
                    for (int j = 0; j < cnt; ++j) {
                        DObj row = objs[(size_t)s->bvh_items[first + j]];
                        row.prim_id = s->bvh_items[first + j];          /* the object's index, as in the other leaf table */
                        std::memcpy(&leafs8[base + 8 * (size_t)j + (size_t)sl], &row, sizeof(DLeaf));
                    }
                }
            }
        }
    }
    if (leafs8.size() < 16 * nodes8.size()) leafs8.resize(16 * nodes8.size());
    if (depth) *depth = deepest;
    if (box_abs) *box_abs = amax;
    return true;
}

void rt_pack_program(const rt_flat_scene* s, std::vector<DProg>& prog, std::vector<DProg>* prog_nobvh, std::vector<int>* bvh_groups)
{
    prog.clear();
    if (prog_nobvh) prog_nobvh->clear();
    if (bvh_groups) bvh_groups->clear();
    /* both programs get every record except that BVH groups go to the group list instead of prog_nobvh */
    auto emit = [&](const DProg& r, bool bvh_group) {
        prog.push_back(r);
        if (bvh_group) { if (bvh_groups) bvh_groups->push_back(r.obj); }
        else if (prog_nobvh) prog_nobvh->push_back(r);
    };
    const rt_object& w = s->objects[s->world];
    /* the wrapper chain the scan's local ray currently has: starts as the world frame */
    int cur_n = 0;
    rt_xform_op cur[2];
    std::memset(cur, 0, sizeof cur);
    auto same_chain = [&](const rt_object& o) {
        if (o.op_count != cur_n) return false;
        for (int k = 0; k < cur_n; ++k)
            if (std::memcmp(&s->ops[o.op_begin + k], &cur[k], sizeof(rt_xform_op)) != 0) return false;
        return true;
    };
    /* a world that has a BVH of its own is traversed directly: no scan records */
    if (w.bvh_root >= 0 && bvh_groups) bvh_groups->push_back(s->world);
    for (int i = 0; i < (w.bvh_root >= 0 ? 0 : w.child_count); ++i) {
        const int oi = w.child_begin + i;
        const rt_object& o = s->objects[oi];
        DProg r;
        std::memset(&r, 0, sizeof r);
        r.obj = oi;
        const bool simple = (o.type == RT_OBJ_SPHERE || (o.type >= RT_OBJ_RECT_XY && o.type <= RT_OBJ_BOX)) && o.op_count <= 2;
        /* a GENERIC record may leave the scan's local ray in its own frame (a BVH group is traversed in
         * place by the wavefront intersect kernel): the next simple object re-establishes its frame */
        if (!simple) { r.head = RT_P_GENERIC; emit(r, o.type == RT_OBJ_GROUP && o.bvh_root >= 0); cur_n = -1; continue; }
        if (!same_chain(o)) {
            DProg x;
            std::memset(&x, 0, sizeof x);
            x.obj = oi;
            cur_n = o.op_count;
            uint32_t ty[2] = {0, 0};
            for (int k = 0; k < cur_n; ++k) {
                cur[k] = s->ops[o.op_begin + k];
                ty[k] = (uint32_t)cur[k].type;
                x.v[3 * k] = cur[k].a; x.v[3 * k + 1] = cur[k].b; x.v[3 * k + 2] = cur[k].c;
            }
            x.head = RT_P_XFORM | ((uint32_t)cur_n << 8) | (ty[0] << 16) | (ty[1] << 20);
            emit(x, false);
        }
        const float* f = o.f;
        if (o.type == RT_OBJ_SPHERE) {
            r.head = RT_P_SPHERE;
            for (int k = 0; k < 4; ++k) r.v[k] = f[k];
            emit(r, false);
        } else if (o.type == RT_OBJ_BOX) {
            /* one record, six faces tested in the order of Box.cuh:24-36 by straight-line code */
            r.head = RT_P_BOX;
            for (int k = 0; k < 6; ++k) r.v[k] = f[k];
            emit(r, false);
        } else {
            const int ka = RT_OBJ_RECT_YZ - o.type;   /* XY -> 2, XZ -> 1, YZ -> 0 */
            r.head = RT_P_RECT | ((uint32_t)ka << 8);
            r.v[0] = f[4]; r.v[1] = f[0]; r.v[2] = f[1]; r.v[3] = f[2]; r.v[4] = f[3];
            emit(r, false);
        }
    }
    DProg e;
    std::memset(&e, 0, sizeof e);
    e.head = RT_P_END;
    emit(e, false);
    emit(e, false);      /* the scan prefetches one record past the one it runs */
}

uint32_t rt_scene_features(const rt_flat_scene* s)
{
    uint32_t f = 0;
    for (int i = 0; i < s->n_objects; ++i) {
        const rt_object& o = s->objects[i];
        if (o.op_count > 0) f |= RT_F_XFORM;
        if (o.op_count > 2) f |= RT_F_LONGCHAIN;
        switch (o.type) {
        case RT_OBJ_SPHERE: f |= RT_F_SPHERE; break;
        case RT_OBJ_MOVING_SPHERE: f |= RT_F_MSPHERE; break;
        case RT_OBJ_RECT_XY: case RT_OBJ_RECT_XZ: case RT_OBJ_RECT_YZ: case RT_OBJ_BOX: f |= RT_F_PLANAR; break;
        case RT_OBJ_TRIANGLE: f |= RT_F_TRI; break;
        case RT_OBJ_MEDIUM: f |= RT_F_MEDIA; break;
        case RT_OBJ_GROUP:
            if (o.bvh_root >= 0) f |= RT_F_BVH;
            else if (i != s->world) f |= RT_F_GROUPLIST;
            break;
        default: break;
        }
    }
    for (int i = 0; i < s->n_materials; ++i) {
        switch (s->materials[i].type) {
        case RT_MAT_GLASS: f |= RT_F_GLASS; break;
        case RT_MAT_METAL: f |= RT_F_METAL; break;
        case RT_MAT_ISOTROPIC: f |= RT_F_ISO; break;
        default: break;
        }
    }
    for (int i = 0; i < s->n_textures; ++i) {
        if (s->textures[i].type == RT_TEX_CHECKER) f |= RT_F_CHECKER;
        if (s->textures[i].type == RT_TEX_IMAGE) f |= RT_F_IMAGE;
        if (s->textures[i].type == RT_TEX_NOISE) f |= RT_F_NOISE;
    }
    if (s->camera.lens_radius != 0.f) f |= RT_F_LENS;
    return f;
}

extern "C" uint32_t rt_flat_features(const rt_flat_scene* flat)
{
    std::string why;
    if (!flat || !rt_validate_scene(flat, why, nullptr)) return 0u;
    return rt_scene_features(flat);
}

extern "C" int rt_flat_program(const rt_flat_scene* flat, uint32_t* out_words, int cap_records)
{
    std::string why;
    if (!flat || cap_records < 0 || (cap_records && !out_words)) { rt_set_error("rt_flat_program: bad arguments"); return RT_ERR_ARG; }
    if (!rt_validate_scene(flat, why, nullptr)) { rt_set_error("rt_flat_program: invalid scene: " + why); return RT_ERR_ARG; }
    std::vector<DProg> prog;
    rt_pack_program(flat, prog);
    const size_t n = prog.size() < (size_t)cap_records ? prog.size() : (size_t)cap_records;
    if (n) std::memcpy(out_words, prog.data(), n * sizeof(DProg));
    return (int)prog.size();
}

extern "C" int rt_flat_program_split(const rt_flat_scene* flat, uint32_t* out_words, int cap_records, int32_t* out_groups, int cap_groups,
                                     int* n_groups)
{
    std::string why;
    if (!flat || cap_records < 0 || cap_groups < 0 || (cap_records && !out_words) || (cap_groups && !out_groups) || !n_groups) {
        rt_set_error("rt_flat_program_split: bad arguments");
        return RT_ERR_ARG;
    }
    if (!rt_validate_scene(flat, why, nullptr)) { rt_set_error("rt_flat_program_split: invalid scene: " + why); return RT_ERR_ARG; }
    std::vector<DProg> prog, plain;
    std::vector<int> groups;
    rt_pack_program(flat, prog, &plain, &groups);
    const size_t n = plain.size() < (size_t)cap_records ? plain.size() : (size_t)cap_records;
    if (n) std::memcpy(out_words, plain.data(), n * sizeof(DProg));
    const size_t g = groups.size() < (size_t)cap_groups ? groups.size() : (size_t)cap_groups;
    if (g) std::memcpy(out_groups, groups.data(), g * sizeof(int32_t));
    *n_groups = (int)groups.size();
    return (int)plain.size();
}

/* single-item leaf references of the 4-wide trees (opened leaves and leaves the builder made of one item) */
extern "C" int rt_flat_bvh4_single_refs(const rt_flat_scene* flat)
{
    std::string why;
    if (!flat || !rt_validate_scene(flat, why, nullptr)) { rt_set_error("rt_flat_bvh4_single_refs: invalid scene: " + why); return RT_ERR_ARG; }
    std::vector<DObj> objs((size_t)flat->n_objects);
    rt_pack_objects(flat, objs.data());
    std::vector<DNode4> n4;
    rt_pack_bvh4(flat, objs, n4, nullptr);
    int n = 0;
    for (const DNode4& d : n4) for (int k = 0; k < 4; ++k) if (d.ref[k] < 0 && d.ref[k] != RT_REF_NONE && ((uint32_t)(~d.ref[k]) & 15u) == 0u) ++n;
    return n;
}

/* The 8-wide quantised trees of a scene, checked on the host: every BVH item sits in exactly one row of the leaf table,
 * every dequantised child box contains the box it was made from (interior child: the union of ITS children's dequantised
 * boxes; leaf child: its items' own bounds), every node is reached exactly once.  Returns the number of nodes, or a
 * negative rt_status (RT_ERR_UNSUPPORTED: a leaf of more than two items).  depth8 / rows_used / max_q_slack are optional. */
extern "C" int rt_flat_bvh8_check(const rt_flat_scene* flat, int* depth8, int* rows_used, float* max_q_slack)
{
    std::string why;
    if (!flat || !rt_validate_scene(flat, why, nullptr)) { rt_set_error("rt_flat_bvh8_check: invalid scene: " + why); return RT_ERR_ARG; }
    std::vector<DObj> objs((size_t)flat->n_objects);
    rt_pack_objects(flat, objs.data());
    std::vector<DNode8> n8;
    std::vector<DLeaf> l8;
    int d8 = 0;
    float amax = 0.f;
    if (!rt_pack_bvh8(flat, objs, n8, l8, &d8, &amax)) { rt_set_error("rt_flat_bvh8_check: a leaf holds more than two items"); return RT_ERR_UNSUPPORTED; }
    std::vector<int> item_seen((size_t)flat->n_objects, 0), node_seen(n8.size(), 0);
    int rows = 0;
    double slack = 0.0;
    struct B { double lo[3], hi[3]; };
    auto child_box = [&](const DNode8& d, int sl) {
        B b;
        for (int a = 0; a < 3; ++a) {
            const double sc = std::ldexp(1.0, (int)((d.e_imask >> (8 * a)) & 0xffu) - 127);
            b.lo[a] = (double)d.p[a] + sc * (double)((d.qlo[a][sl >> 2] >> (8 * (sl & 3))) & 0xffu);
            b.hi[a] = (double)d.p[a] + sc * (double)((d.qhi[a][sl >> 2] >> (8 * (sl & 3))) & 0xffu);
        }
        return b;
    };
    /* every item must lie inside its own slot's box AND inside the slot box of every ancestor it is reached through */
    bool ok = true;
    std::vector<B> above;
    std::function<void(uint32_t)> walk = [&](uint32_t ni) {
        if (ni >= n8.size() || node_seen[ni]++) { ok = false; return; }
        const DNode8& d = n8[ni];
        const uint32_t imask = d.e_imask >> 24, lmask = d.leaf_masks & 0xffu, l2 = (d.leaf_masks >> 8) & 0xffu;
        if (imask & lmask) ok = false;
        int below = 0;
        for (int sl = 0; sl < 8; ++sl) {
            const B b = child_box(d, sl);
            if ((imask >> sl) & 1u) {
                above.push_back(b);
                walk(d.child_base + (uint32_t)below++);
                above.pop_back();
            } else if ((lmask >> sl) & 1u) {
                for (int j = 0; j < (((l2 >> sl) & 1u) ? 2 : 1); ++j) {
                    const DLeaf& row = l8[16 * (size_t)ni + 8 * (size_t)j + (size_t)sl];
                    const int oi = (int)row.w[2];
                    if (oi < 0 || oi >= flat->n_objects || item_seen[(size_t)oi]++) { ok = false; continue; }
                    ++rows;
                    const rt_object& o = flat->objects[oi];
                    for (int a = 0; a < 3; ++a) {
                        if ((double)o.bmin[a] < b.lo[a] || (double)o.bmax[a] > b.hi[a]) ok = false;
                        for (const B& up : above) if ((double)o.bmin[a] < up.lo[a] || (double)o.bmax[a] > up.hi[a]) ok = false;
                        slack = std::max(slack, std::max((double)o.bmin[a] - b.lo[a], b.hi[a] - (double)o.bmax[a]));
                    }
                }
            }
        }
    };
    for (int gi = 0; gi < flat->n_objects; ++gi)
        if (flat->objects[gi].type == RT_OBJ_GROUP && flat->objects[gi].bvh_root >= 0) (void)walk((uint32_t)objs[(size_t)gi].pad[1]);
    for (size_t i = 0; i < n8.size(); ++i) if (node_seen[i] != 1) ok = false;
    for (int i = 0; i < flat->n_bvh_items; ++i) if (item_seen[(size_t)flat->bvh_items[i]] != 1) ok = false;
    if (!ok || rows != flat->n_bvh_items) { rt_set_error("rt_flat_bvh8_check: the 8-wide trees do not hold every item once behind boxes that contain it"); return RT_ERR_UNSUPPORTED; }
    if (depth8) *depth8 = d8;
    if (rows_used) *rows_used = rows;
    if (max_q_slack) *max_q_slack = (float)slack;
    return (int)n8.size();
}

extern "C" int rt_flat_bvh_stats(const rt_flat_scene* flat, int* nodes2, int* depth2, int* nodes4, int* depth4)
{
    std::string why;
    if (!flat || !rt_validate_scene(flat, why, nullptr)) { rt_set_error("rt_flat_bvh_stats: invalid scene: " + why); return RT_ERR_ARG; }
    std::vector<DObj> objs((size_t)flat->n_objects);
    rt_pack_objects(flat, objs.data());
    std::vector<DNode> n2;
    std::vector<DObj> leafs;
    std::vector<DNode4> n4;
    int d2 = 0, d4 = 0;
    if (!rt_pack_bvh(flat, objs, n2, leafs, &d2, why)) { rt_set_error("rt_flat_bvh_stats: " + why); return RT_ERR_UNSUPPORTED; }
    rt_pack_bvh4(flat, objs, n4, &d4);
    /* every leaf reference of both trees must land inside the leaf table, and together cover it once */
    std::vector<uint8_t> seen(leafs.size(), 0);
    size_t covered = 0;
    auto leaf = [&](int32_t ref) {
        if (ref >= 0 || ref == RT_REF_NONE) return true;
        const uint32_t first = (uint32_t)(~ref) >> 4, count = ((uint32_t)(~ref) & 15u) + 1u;
        if ((size_t)first + count > leafs.size()) return false;
        for (uint32_t i = 0; i < count; ++i) { if (seen[first + i]) return false; seen[first + i] = 1; ++covered; }
        return true;
    };
    for (const DNode4& n : n4) for (int k = 0; k < 4; ++k) if (!leaf(n.ref[k])) { rt_set_error("rt_flat_bvh_stats: bad leaf reference in a 4-wide node"); return RT_ERR_UNSUPPORTED; }
    for (const DObj& o : objs) if ((obj_is_bvh_group(o)) && !leaf(o.pad[0])) { rt_set_error("rt_flat_bvh_stats: bad root reference"); return RT_ERR_UNSUPPORTED; }
    if (covered != leafs.size()) { rt_set_error("rt_flat_bvh_stats: the 4-wide trees do not cover the leaf table exactly once"); return RT_ERR_UNSUPPORTED; }
    if (nodes2) *nodes2 = (int)n2.size();
    if (depth2) *depth2 = d2;
    if (nodes4) *nodes4 = (int)n4.size();
    if (depth4) *depth4 = d4;
    return RT_OK;
}

extern "C" int rt_quantise_rgb8(const float* rgb, int width, int height, uint8_t* out)
{
    if (!rgb || !out || width <= 0 || height <= 0) { rt_set_error("rt_quantise_rgb8: bad arguments"); return RT_ERR_ARG; }
    size_t index = 0;
    for (int j = height - 1; j >= 0; j--)
        for (int i = 0; i < width; i++) {
            const float* px = rgb + 3 * ((size_t)j * width + i);
            for (int k = 0; k < 3; ++k) {
                double v = 255.99 * px[k];
                int q = v != v ? 0 : (v < 0 ? 0 : (v > 255 ? 255 : (int)v));
                out[index++] = (uint8_t)q;
            }
        }
    return RT_OK;
}

extern "C" int rt_write_ppm(const char* path, const uint8_t* rgb8, int width, int height)
{
    if (!path || !rgb8 || width <= 0 || height <= 0) { rt_set_error("rt_write_ppm: bad arguments"); return RT_ERR_ARG; }
    FILE* f = std::fopen(path, "wb");
    if (!f) { rt_set_error(std::string("rt_write_ppm: cannot open ") + path); return RT_ERR_ARG; }
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    size_t n = (size_t)width * height * 3;
    bool ok = std::fwrite(rgb8, 1, n, f) == n;
    ok = std::fclose(f) == 0 && ok;
    if (!ok) { rt_set_error("rt_write_ppm: short write"); return RT_ERR_ARG; }
    return RT_OK;
}

/* ---- PNG out, PPM in: the image files either side of the path (main.cu:566-570, 688-697) ------------
 * The reference writes 8-bit RGB PNG through the vendored stb writer.  This writes the same container with
 * the pixel rows in stored (uncompressed) deflate blocks: any PNG reader opens it, nothing third-party is
 * needed, and the bytes are a pure function of the pixels. */
static uint32_t png_crc(const uint8_t* p, size_t n, uint32_t crc)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return crc;
}
static void png_be32(std::vector<uint8_t>& v, uint32_t x)
{
    for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s));
}
static bool png_chunk(FILE* f, const char type[4], const std::vector<uint8_t>& body)
{
    std::vector<uint8_t> head;
    png_be32(head, (uint32_t)body.size());
    head.insert(head.end(), type, type + 4);
    uint32_t crc = png_crc(head.data() + 4, 4, 0xffffffffu);
    if (!body.empty()) crc = png_crc(body.data(), body.size(), crc);
    crc ^= 0xffffffffu;
    std::vector<uint8_t> tail;
    png_be32(tail, crc);
    /* IEND has an empty body: fwrite must not be handed a null pointer, not even for zero bytes */
    return std::fwrite(head.data(), 1, 8, f) == 8 && (body.empty() || std::fwrite(body.data(), 1, body.size(), f) == body.size()) &&
           std::fwrite(tail.data(), 1, 4, f) == 4;
}

extern "C" int rt_write_png(const char* path, const uint8_t* rgb8, int width, int height)
{
    if (!path || !rgb8 || width <= 0 || height <= 0) { rt_set_error("rt_write_png: bad arguments"); return RT_ERR_ARG; }
    const size_t row = (size_t)width * 3, raw_n = (row + 1) * (size_t)height;
    if (raw_n > 0xfff00000u) { rt_set_error("rt_write_png: image too large for one IDAT chunk"); return RT_ERR_UNSUPPORTED; }
    std::vector<uint8_t> ihdr;
    png_be32(ihdr, (uint32_t)width);
    png_be32(ihdr, (uint32_t)height);
    const uint8_t fmt[5] = {8, 2, 0, 0, 0};           /* 8 bits, colour type 2 = RGB, deflate, filter 0, no interlace */
    ihdr.insert(ihdr.end(), fmt, fmt + 5);
    /* zlib stream: header, stored blocks of <= 65535 bytes over (filter byte 0 + row) per row, Adler-32 */
    std::vector<uint8_t> z;
    z.reserve(raw_n + raw_n / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    std::vector<uint8_t> raw(raw_n);
    for (int y = 0; y < height; ++y) {
        raw[(row + 1) * y] = 0;
        std::memcpy(&raw[(row + 1) * y + 1], rgb8 + row * y, row);
    }
    for (size_t off = 0; off < raw_n;) {
        const size_t n = raw_n - off < 65535 ? raw_n - off : 65535;
        z.push_back(off + n == raw_n ? 1 : 0);
        z.push_back((uint8_t)(n & 0xff)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xff)); z.push_back((uint8_t)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        off += n;
    }
    for (size_t i = 0; i < raw_n;) {                  /* Adler-32, reduced every 5552 bytes */
        const size_t end = raw_n - i < 5552 ? raw_n : i + 5552;
        for (; i < end; ++i) { a += raw[i]; b += a; }
        a %= 65521u; b %= 65521u;
    }
    png_be32(z, (b << 16) | a);
    FILE* f = std::fopen(path, "wb");
    if (!f) { rt_set_error(std::string("rt_write_png: cannot open ") + path); return RT_ERR_ARG; }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    bool ok = std::fwrite(sig, 1, 8, f) == 8;
    ok = ok && png_chunk(f, "IHDR", ihdr) && png_chunk(f, "IDAT", z) && png_chunk(f, "IEND", std::vector<uint8_t>());
    ok = std::fclose(f) == 0 && ok;
    if (!ok) { rt_set_error("rt_write_png: short write"); return RT_ERR_ARG; }
    return RT_OK;
}

/* binary PPM (P6, maxval 255, '#' comments in the header): how an RGB8 texture such as the reference's
 * earth map (main.cu:566-570 loads a JPEG through stb, out of scope) reaches ImageTexture */
extern "C" int rt_read_ppm(const char* path, uint8_t** rgb8_out, int* width, int* height)
{
    if (!path || !rgb8_out || !width || !height) { rt_set_error("rt_read_ppm: bad arguments"); return RT_ERR_ARG; }
    *rgb8_out = nullptr; *width = *height = 0;
    FILE* f = std::fopen(path, "rb");
    if (!f) { rt_set_error(std::string("rt_read_ppm: cannot open ") + path); return RT_ERR_ARG; }
    auto token = [&](long& out) {
        int c = std::fgetc(f);
        for (;;) {
            while (c == ' ' || c == '\t' || c == '\n' || c == '\r') c = std::fgetc(f);
            if (c != '#') break;
            while (c != '\n' && c != EOF) c = std::fgetc(f);
        }
        if (c < '0' || c > '9') return false;
        out = 0;
        while (c >= '0' && c <= '9') { out = out * 10 + (c - '0'); if (out > 1000000000L) return false; c = std::fgetc(f); }
        return c == ' ' || c == '\t' || c == '\n' || c == '\r';     /* exactly one whitespace byte ends a token */
    };
    long w = 0, h = 0, maxv = 0;
    const bool magic = std::fgetc(f) == 'P' && std::fgetc(f) == '6';
    if (!magic || !token(w) || !token(h) || !token(maxv) || w <= 0 || h <= 0 || maxv != 255 || w > 65536 || h > 65536) {
        std::fclose(f);
        rt_set_error("rt_read_ppm: not a binary PPM with maxval 255");
        return RT_ERR_UNSUPPORTED;
    }
    const size_t n = (size_t)w * (size_t)h * 3;
    uint8_t* px = (uint8_t*)std::malloc(n);
    if (!px) { std::fclose(f); rt_set_error("rt_read_ppm: out of memory"); return RT_ERR_ARG; }
    const bool ok = std::fread(px, 1, n, f) == n;
    std::fclose(f);
    if (!ok) { std::free(px); rt_set_error("rt_read_ppm: truncated pixel data"); return RT_ERR_ARG; }
    *rgb8_out = px; *width = (int)w; *height = (int)h;
    return RT_OK;
}
extern "C" void rt_free_pixels(uint8_t* rgb8) { std::free(rgb8); }
