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

}  // namespace rt
