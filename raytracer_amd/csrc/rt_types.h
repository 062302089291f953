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
struct alignas(16) Node64 {
    float lbox[6];   // left child: lo.xyz, hi.xyz
    float rbox[6];   // right child
    int32_t lref;
    int32_t rref;
    int32_t pad[2];
};

// 48-byte triangle, traversal order.  Edges are stored pre-subtracted exactly as Triangle::hit forms them
// (src/geometry.cpp:268-273: a-b and a-c component-wise), so the device arithmetic is the same rounding sequence.
struct alignas(16) Tri48 {
    float ax, ay, az;  int32_t run;      // vertex a; run = triangles from this one to the end of its leaf
    float abx, aby, abz; int32_t id;     // a - b; id = caller's global triangle id (mesh order, then face order)
    float acx, acy, acz; int32_t mesh;   // a - c; mesh index
};

struct Material {
    int32_t kind;
    float base[3], ambient[3], spec[3], transp[3];
    float shininess, reflectivity, refractivity;
};

struct Light { int32_t kind; float vec[3]; float color[3]; float pad; };

struct Camera { float pos[3], up[3], w[3], side[3]; float near_clip, far_clip; };

}  // namespace rt
