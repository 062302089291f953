// Device builder of the pruning traversal's tree (see lbvh_build.hip).
#pragma once
#include <string>

#include "../../include/rt_amd.h"
#include "rt_types.h"

namespace rt {

// Arrays in device memory, laid out like FlatScene's: nodes[0] is the entry, both reference pairs of a node are equal,
// tris/info/nrm in leaf order with run == run_fine.  Owned by the caller (free_device_tree).
struct DeviceTree {
    Node64* nodes = nullptr; Tri48* tris = nullptr; TriInfo* info = nullptr; float* nrm = nullptr;
    int64_t n_tris = 0, n_nodes = 0;
    int max_depth = 0;
};

// meshes: host arrays as given to rt_scene_create; rank_of_id[n]: tie rank of each triangle by global id (from the
// reference's trees); n: total triangles (> RT_LBVH_LEAF).  Builds on the current device and waits for the result.
bool build_lbvh_device(const rt_mesh_desc* meshes, int n_meshes, const int32_t* rank_of_id, int64_t n, DeviceTree* out, std::string* err);
void free_device_tree(DeviceTree* t);

}  // namespace rt
