#pragma once
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_types.h"

namespace rt {

struct FlatScene {
    std::vector<Node64> nodes;   // nodes[0] is the entry node
    std::vector<Tri48> tris;     // traversal order (the reference's leaves stay contiguous)
    std::vector<TriInfo> info;   // same order
    std::vector<float> nrm;      // [n_tris][3][3] vertex normals, same order as tris
    int n_leaves = 0;
    int max_depth = 0;           // longest chain of inner nodes from the entry to a leaf
};

// Builds the per-mesh trees and the top tree; returns false and fills *err on bad input.
bool build_flat_scene(const rt_mesh_desc* meshes, int n_meshes, FlatScene* out, std::string* err);

// Second tree over the same triangles for the pruning traversal: one binned surface-area-heuristic tree over ALL
// meshes (the pruning traversal's answer does not depend on the tree: ties are decided by the ranks taken from
// `exact`), leaves of at most four triangles.  Fills nodes (both reference pairs the same), tris (run == run_fine),
// info, nrm in its own leaf order.
bool build_sah_scene(const rt_mesh_desc* meshes, int n_meshes, const FlatScene& exact, FlatScene* out, std::string* err);

}  // namespace rt
