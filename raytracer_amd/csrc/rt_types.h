// Device-side data layout of a flattened scene (shared by the host builder and the HIP kernels).
#pragma once
#include <stdint.h>

namespace rt {

// Child reference of a node:  >= 0  inner node index;  RT_NONE  no child;
// otherwise a leaf: ~ref is the index of the leaf's first triangle in the traversal-order triangle array.
static const int32_t RT_NONE = INT32_MIN;

// 64-byte inner node: the boxes of BOTH children live in the parent, so one fetch decides both.
// The reference stores a node's own box in the node (src/geometry.h:41-45); a child's box test there
// (src/geometry.cpp:157) is this parent's test of the matching half here.
// Two pairs of child references share the boxes: lref/rref describe the reference's tree exactly (its bottom
// nodes are leaves, however many triangles they hold); lref_fine/rref_fine are the same except that a leaf with
// more than a few triangles is replaced by a small subtree over that leaf's triangles (appended nodes).
// In the tree of the pruning traversal, as a scene holds it on the device, the low mantissa byte of lbox[0..3] and of
// rbox[0..3] spells lref_fine and rref_fine once more (byte k in plane k, planes moved outwards to make room): a lane
// fetching its own node reads the first 48 bytes only.
struct alignas(16) Node64 {
    float lbox[6];   // left child: lo.xyz, hi.xyz
    float rbox[6];   // right child
    int32_t lref;
    int32_t rref;
    int32_t lref_fine;
    int32_t rref_fine;
};

// 48-byte triangle, traversal order.  Edges are stored pre-subtracted exactly as Triangle::hit forms them
// (src/geometry.cpp:268-273: a-b and a-c component-wise), so the device arithmetic is the same rounding sequence.
struct alignas(16) Tri48 {
    float ax, ay, az;  int32_t run;        // vertex a; run = triangles from this one to the end of the reference's leaf (in the
                                           // pruning traversal's own copies on the device: the number of the reference's leaf
                                           // that holds the triangle, see DScene::leaf_box)
    float abx, aby, abz; int32_t run_fine; // a - b; the same for the refined leaves
    float acx, acy, acz; int32_t rank;     // a - c; tie rank: among equal t the reference returns the lowest rank
};

// Per triangle (same order as Tri48), read only for the final hit.
struct TriInfo { int32_t id; int32_t mesh; };   // caller's global triangle id (mesh order, then face order); mesh index

struct Material {
    int32_t kind;
    float base[3], ambient[3], spec[3], transp[3];
    float shininess, reflectivity, refractivity;
};

struct Light { int32_t kind; float vec[3]; float color[3]; float pad; };

struct Camera { float pos[3], up[3], w[3], side[3]; float near_clip, far_clip; };

}  // namespace rt
