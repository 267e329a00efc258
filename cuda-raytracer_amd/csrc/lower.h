/*
 * lower.h — scene graph (api/rt_scene_api.h recorders) -> flat tables (include/rt_scene.h).
 */
#ifndef RT_LOWER_H
#define RT_LOWER_H
#include <string>
#include <vector>
#include "../../include/rt_scene.h"
#include "../api/rt_scene_api.h"

#ifndef RT_DEFAULT_LEAF
#define RT_DEFAULT_LEAF 2
#endif

namespace rt {

/* Owns the storage behind an rt_flat_scene. */
struct FlatScene {
    std::vector<rt_object>   objects;
    std::vector<rt_xform_op> ops;
    std::vector<rt_material> materials;
    std::vector<rt_texture>  textures;
    std::vector<uint8_t>     image_bytes;
    std::vector<rt_bvh_node> bvh_nodes;
    std::vector<int32_t>     bvh_items;
    rt_flat_scene            view;   /* pointers into the vectors above */
    void refresh_view();
};

struct LowerOptions {
    int bvh_min_children = 9;   /* groups with at least this many leaf children get a BVH */
    int bvh_leaf_size    = RT_DEFAULT_LEAF;   /* max objects per BVH leaf */
};

/* Returns false and fills `err` when the graph uses nesting the tables cannot express. */
bool lower(Hitable* world, const Camera* cam, const LowerOptions& opt, FlatScene& out, std::string& err);

/* Binned-SAH BVH over objects [first, first+count) of `objects`; appends to nodes/items
 * and returns the root node index. */
int build_bvh(const std::vector<rt_object>& objects, int first, int count, int leaf_size,
              std::vector<rt_bvh_node>& nodes, std::vector<int32_t>& items);

} // namespace rt
#endif
