// Device builder of the tree the pruning traversal walks (a linear bounding-volume hierarchy):
//   1. one key per triangle: the mesh's place in a Morton order of the meshes' centres (high bits), then the Morton
//      code of the centre of the triangle's box on a 16384-cube grid over ITS MESH (42 bits).  Keeping the meshes apart
//      makes the top of the hierarchy a tree over meshes and gives every mesh a subtree of its own - the arrangement of
//      the reference (a tree per mesh, src/AABBTree.cpp:10) - so that a mesh of a few huge triangles (a ground plane
//      under a field of small objects) does not widen the boxes of the small triangles its centres fall among;
//   2. a stable radix sort of (key, triangle) pairs (rocPRIM); equal keys are told apart by their sorted position;
//   3. the hierarchy over the sorted keys: inner node i splits the range of keys that share its longest common prefix
//      (Karras, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees", 2012) - every inner node
//      is found independently;
//   4. boxes bottom-up: a thread per triangle climbs towards the root, the second thread to arrive at a node carries
//      on with the union of both children; a subtree of at most RT_LBVH_LEAF triangles becomes one leaf (a run of
//      consecutive triangles in key order);
//   5. the triangle records (Tri48, TriInfo, vertex normals) in key order, with the tie ranks taken from the
//      reference's trees.
// The reference builds its trees on one host thread (src/geometry.cpp:61-153); its own trees are still built on the
// host (they define the tie ranks and serve the exact-box mode) - this replaces the second, surface-area-heuristic
// tree for scenes where build time matters more than the last tenth of traversal speed.  The pruning traversal's
// answer does not depend on the tree.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "lbvh_build.h"

#ifndef RT_LBVH_LEAF
#define RT_LBVH_LEAF 4
#endif

namespace rt {
namespace {

#define LB_TRY(expr)                                                                  \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) { *err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } \
    } while (0)

constexpr int B = 256;
inline unsigned blocks(int64_t n) { return (unsigned)((n + B - 1) / B); }

// order-preserving map float -> int (for atomicMin / atomicMax on floats)
__device__ __forceinline__ int f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__host__ __device__ inline float ord2f(int i) {
    const int j = i >= 0 ? i : i ^ 0x7fffffff;
    float f; memcpy(&f, &j, 4);
    return f;
}

struct TriBox { float lo[3], hi[3]; };
__device__ __forceinline__ TriBox tri_box(const float* p) {
    TriBox b;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = fminf(p[a], fminf(p[3 + a], p[6 + a]));
        b.hi[a] = fmaxf(p[a], fmaxf(p[3 + a], p[6 + a]));
    }
    return b;
}

// bounds[m][0..2] = min of the box centres of mesh m's triangles (as ordered ints), [3..5] = max
__global__ __launch_bounds__(B) void k_bounds(const float* pos, const int32_t* mesh_of, int64_t n, int* bounds) {
    const int64_t i = (int64_t)blockIdx.x * B + threadIdx.x;
    float c[3] = {INFINITY, INFINITY, INFINITY}, d[3] = {-INFINITY, -INFINITY, -INFINITY};
    int m = -1;
    if (i < n) {
        const TriBox b = tri_box(pos + i * 9);
        for (int a = 0; a < 3; ++a) c[a] = d[a] = 0.5f * b.lo[a] + 0.5f * b.hi[a];
        m = mesh_of[i];
    }
    // triangles come mesh by mesh: nearly every wave holds one mesh only and needs six atomics
    const int m0 = __builtin_amdgcn_readfirstlane(m);
    if (__builtin_amdgcn_ballot_w64(m != m0) == 0) {
        if (m0 < 0) return;
        for (int a = 0; a < 3; ++a)
            for (int off = 32; off > 0; off >>= 1) {
                c[a] = fminf(c[a], __shfl_down(c[a], off, 64));
                d[a] = fmaxf(d[a], __shfl_down(d[a], off, 64));
            }
        if ((threadIdx.x & 63) == 0)
            for (int a = 0; a < 3; ++a) { atomicMin(&bounds[m0 * 6 + a], f2ord(c[a])); atomicMax(&bounds[m0 * 6 + 3 + a], f2ord(d[a])); }
    } else if (m >= 0) {
        for (int a = 0; a < 3; ++a) { atomicMin(&bounds[m * 6 + a], f2ord(c[a])); atomicMax(&bounds[m * 6 + 3 + a], f2ord(d[a])); }
    }
}

__device__ __forceinline__ unsigned long long spread21(unsigned long long v) {       // 21 bits -> every third bit
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

__global__ __launch_bounds__(B) void k_keys(const float* pos, const int32_t* mesh_of, const uint32_t* mesh_place, int64_t n,
                                            const int* bounds, unsigned long long* keys, uint32_t* ids) {
    const int64_t i = (int64_t)blockIdx.x * B + threadIdx.x;
    if (i >= n) return;
    const TriBox b = tri_box(pos + i * 9);
    const int m = mesh_of[i];
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float lo = ord2f(bounds[m * 6 + a]), hi = ord2f(bounds[m * 6 + 3 + a]);
        const float ext = hi - lo;
        float u = ext > 0.f ? ((0.5f * b.lo[a] + 0.5f * b.hi[a]) - lo) / ext : 0.f;
        u = fminf(fmaxf(u * 16384.f, 0.f), 16383.f);
        q[a] = (uint32_t)u;
    }
    keys[i] = ((unsigned long long)mesh_place[m] << 42) | (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    ids[i] = (uint32_t)i;
}

// length of the common prefix of the keys at sorted positions i and j (-1 outside the array); equal keys continue with
// the common prefix of the positions themselves, which makes every (key, position) distinct
__device__ __forceinline__ int prefix(const unsigned long long* keys, int64_t n, int64_t i, int64_t j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long x = keys[i] ^ keys[j];
    return x ? __clzll((long long)x) : 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

// Inner node i of n - 1: its key range [lo, hi], its two children (an inner node j as j, a single key k as ~k) and
// the children's parent links.
__global__ __launch_bounds__(B) void k_hierarchy(const unsigned long long* keys, int64_t n, int32_t* left, int32_t* right,
                                                 int32_t* range_lo, int32_t* range_hi, int32_t* parent_inner, int32_t* parent_leaf) {
    const int64_t i = (int64_t)blockIdx.x * B + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = prefix(keys, n, i, i - d);
    int64_t lmax = 2;
    while (prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int64_t l = 0;
    for (int64_t t = lmax / 2; t >= 1; t /= 2)
        if (prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int64_t j = i + l * d;
    const int dnode = prefix(keys, n, i, j);
    int64_t s = 0;
    for (int64_t t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int64_t gamma = i + s * d + (d < 0 ? -1 : 0);
    const int64_t lo = i < j ? i : j, hi = i < j ? j : i;
    range_lo[i] = (int32_t)lo; range_hi[i] = (int32_t)hi;
    if (lo == gamma) { left[i] = ~(int32_t)gamma; parent_leaf[gamma] = (int32_t)i; }
    else { left[i] = (int32_t)gamma; parent_inner[gamma] = (int32_t)i; }
    if (hi == gamma + 1) { right[i] = ~(int32_t)(gamma + 1); parent_leaf[gamma + 1] = (int32_t)i; }
    else { right[i] = (int32_t)(gamma + 1); parent_inner[gamma + 1] = (int32_t)i; }
}

// One thread per sorted triangle climbs from its key towards the root.  At each node it stores the box of the subtree it
// comes from (and that subtree's reference: the inner node, or a leaf when the subtree has at most RT_LBVH_LEAF
// triangles) into the node's left or right half; the first thread to arrive stops, the second one - which then sees
// both halves - carries on with the union.  height: levels of inner nodes below (a leaf has height 0).
__global__ __launch_bounds__(B) void k_boxes(const float* pos, const uint32_t* ids, int64_t n, const int32_t* left,
                                             const int32_t* range_lo, const int32_t* range_hi, const int32_t* parent_inner,
                                             const int32_t* parent_leaf, Node64* nodes, uint32_t* arrived, int32_t* height, int32_t* root_height) {
    const int64_t k = (int64_t)blockIdx.x * B + threadIdx.x;
    if (k >= n) return;
    const uint32_t id = ids[k];
    const TriBox tb = tri_box(pos + (int64_t)id * 9);
    float box[6] = {tb.lo[0], tb.lo[1], tb.lo[2], tb.hi[0], tb.hi[1], tb.hi[2]};
    int32_t ref = ~(int32_t)k;          // reference of the subtree this thread stands for
    int32_t lo = (int32_t)k, hi = (int32_t)k, h = 0;
    int32_t from = ~(int32_t)k;         // Karras child reference, to tell left from right
    int32_t node = parent_leaf[k];
    for (;;) {
        const bool is_left = left[node] == from;
        Node64* N = nodes + node;
        float* dst = is_left ? N->lbox : N->rbox;
        for (int a = 0; a < 6; ++a) dst[a] = box[a];
        if (is_left) { N->lref = ref; N->lref_fine = ref; } else { N->rref = ref; N->rref_fine = ref; }
        height[node * 2 + (is_left ? 0 : 1)] = h;
        __threadfence();
        if (atomicAdd(&arrived[node], 1u) == 0u) return;       // the sibling subtree is not finished yet
        __threadfence();
        const volatile float* oth = is_left ? N->rbox : N->lbox;
        for (int a = 0; a < 3; ++a) { box[a] = fminf(box[a], oth[a]); box[3 + a] = fmaxf(box[3 + a], oth[3 + a]); }
        const int32_t ho = ((const volatile int32_t*)height)[node * 2 + (is_left ? 1 : 0)];
        lo = range_lo[node]; hi = range_hi[node];
        const bool leaf = hi - lo + 1 <= RT_LBVH_LEAF;
        h = leaf ? 0 : 1 + (h > ho ? h : ho);
        ref = leaf ? ~lo : node;
        if (node == 0) { *root_height = h; return; }
        from = node;
        node = parent_inner[node];
    }
}

// Triangle records in key order.  run = triangles from this one to the end of its leaf: the leaf is the highest
// ancestor whose key range holds at most RT_LBVH_LEAF triangles.
__global__ __launch_bounds__(B) void k_emit(const float* pos, const float* nrm, const int32_t* mesh_of, const int32_t* rank_of,
                                            const uint32_t* ids, int64_t n, const int32_t* range_lo, const int32_t* range_hi,
                                            const int32_t* parent_inner, const int32_t* parent_leaf, Tri48* tris, TriInfo* info, float* nrm_out) {
    const int64_t k = (int64_t)blockIdx.x * B + threadIdx.x;
    if (k >= n) return;
    const uint32_t id = ids[k];
    int32_t hi = (int32_t)k;
    int32_t node = parent_leaf[k];
    while (range_hi[node] - range_lo[node] + 1 <= RT_LBVH_LEAF) {
        hi = range_hi[node];
        if (node == 0) break;
        node = parent_inner[node];
    }
    const float* a = pos + (int64_t)id * 9;
    Tri48 q;
    q.ax = a[0]; q.ay = a[1]; q.az = a[2]; q.run = hi - (int32_t)k + 1;
    q.abx = a[0] - a[3]; q.aby = a[1] - a[4]; q.abz = a[2] - a[5]; q.run_fine = q.run;
    q.acx = a[0] - a[6]; q.acy = a[1] - a[7]; q.acz = a[2] - a[8]; q.rank = rank_of[id];
    tris[k] = q;
    TriInfo fi; fi.id = (int32_t)id; fi.mesh = mesh_of[id];
    info[k] = fi;
    const float* s = nrm + (int64_t)id * 9;
    float* d = nrm_out + k * 9;
    for (int c = 0; c < 9; ++c) d[c] = s[c];
}

}  // namespace

void free_device_tree(DeviceTree* t) {
    if (t->nodes) (void)hipFree(t->nodes);
    if (t->tris) (void)hipFree(t->tris);
    if (t->info) (void)hipFree(t->info);
    if (t->nrm) (void)hipFree(t->nrm);
    *t = DeviceTree{};
}

bool build_lbvh_device(const rt_mesh_desc* meshes, int n_meshes, const int32_t* rank_of_id, int64_t n, DeviceTree* out, std::string* err) {
    *out = DeviceTree{};
    if (n <= RT_LBVH_LEAF) { *err = "too few triangles for the device builder"; return false; }
    if (n >= ((int64_t)1 << 31) - 1) { *err = "too many triangles for the device builder"; return false; }
    struct Temp {
        std::vector<void*> p;
        ~Temp() { for (void* q : p) (void)hipFree(q); }
        bool get(void** dst, size_t bytes, std::string* err) {
            hipError_t e = hipMalloc(dst, bytes ? bytes : 1);
            if (e != hipSuccess) { *err = std::string("hipMalloc: ") + hipGetErrorString(e); return false; }
            p.push_back(*dst);
            return true;
        }
    } tmp;
    float *d_pos = nullptr, *d_nrm = nullptr;
    int32_t *d_mesh = nullptr, *d_rank = nullptr, *d_left = nullptr, *d_right = nullptr, *d_lo = nullptr, *d_hi = nullptr, *d_pi = nullptr, *d_pl = nullptr, *d_height = nullptr;
    int* d_bounds = nullptr; uint32_t *d_arrived = nullptr, *d_place = nullptr;
    unsigned long long *d_keys = nullptr, *d_sorted = nullptr;
    uint32_t *d_ids = nullptr, *d_ids_sorted = nullptr;
    if (!tmp.get((void**)&d_pos, (size_t)n * 36, err) || !tmp.get((void**)&d_nrm, (size_t)n * 36, err) ||
        !tmp.get((void**)&d_mesh, (size_t)n * 4, err) || !tmp.get((void**)&d_rank, (size_t)n * 4, err) ||
        !tmp.get((void**)&d_keys, (size_t)n * 8, err) || !tmp.get((void**)&d_sorted, (size_t)n * 8, err) ||
        !tmp.get((void**)&d_ids, (size_t)n * 4, err) || !tmp.get((void**)&d_ids_sorted, (size_t)n * 4, err) ||
        !tmp.get((void**)&d_left, (size_t)n * 4, err) || !tmp.get((void**)&d_right, (size_t)n * 4, err) ||
        !tmp.get((void**)&d_lo, (size_t)n * 4, err) || !tmp.get((void**)&d_hi, (size_t)n * 4, err) ||
        !tmp.get((void**)&d_pi, (size_t)n * 4, err) || !tmp.get((void**)&d_pl, (size_t)n * 4, err) ||
        !tmp.get((void**)&d_height, (size_t)n * 8 + 4, err) || !tmp.get((void**)&d_arrived, (size_t)n * 4, err) ||
        !tmp.get((void**)&d_bounds, (size_t)n_meshes * 6 * sizeof(int), err) || !tmp.get((void**)&d_place, (size_t)n_meshes * 4, err))
        return false;
    if (n_meshes > (1 << 21)) { *err = "too many meshes for the device builder"; return false; }
    // inputs: the meshes' triangles in id order, the mesh of each triangle, the tie rank of each triangle
    {
        std::vector<int32_t> mesh_of((size_t)n);
        int64_t at = 0;
        for (int mi = 0; mi < n_meshes; ++mi) {
            const int64_t c = meshes[mi].n_tris;
            if (c <= 0) continue;
            LB_TRY(hipMemcpy(d_pos + at * 9, meshes[mi].pos, (size_t)c * 36, hipMemcpyHostToDevice));
            LB_TRY(hipMemcpy(d_nrm + at * 9, meshes[mi].nrm, (size_t)c * 36, hipMemcpyHostToDevice));
            for (int64_t k = 0; k < c; ++k) mesh_of[(size_t)(at + k)] = mi;
            at += c;
        }
        if (at != n) { *err = "internal: triangle count mismatch"; return false; }
        LB_TRY(hipMemcpy(d_mesh, mesh_of.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        LB_TRY(hipMemcpy(d_rank, rank_of_id, (size_t)n * 4, hipMemcpyHostToDevice));
    }
    {
        std::vector<int> hb((size_t)n_meshes * 6);
        for (int m = 0; m < n_meshes; ++m)
            for (int a = 0; a < 3; ++a) { hb[(size_t)m * 6 + a] = INT32_MAX; hb[(size_t)m * 6 + 3 + a] = INT32_MIN; }
        LB_TRY(hipMemcpy(d_bounds, hb.data(), hb.size() * sizeof(int), hipMemcpyHostToDevice));
        k_bounds<<<blocks(n), B>>>(d_pos, d_mesh, n, d_bounds);
        LB_TRY(hipGetLastError());
        LB_TRY(hipMemcpy(hb.data(), d_bounds, hb.size() * sizeof(int), hipMemcpyDeviceToHost));
        // the meshes in Morton order of their centres (1024-cube grid over all centres); empty meshes last
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        std::vector<float> ctr((size_t)n_meshes * 3, 0.f);
        for (int m = 0; m < n_meshes; ++m) {
            if (meshes[m].n_tris <= 0) continue;
            for (int a = 0; a < 3; ++a) {
                const float c = 0.5f * ord2f(hb[(size_t)m * 6 + a]) + 0.5f * ord2f(hb[(size_t)m * 6 + 3 + a]);
                ctr[(size_t)m * 3 + a] = c; lo[a] = std::min(lo[a], c); hi[a] = std::max(hi[a], c);
            }
        }
        auto spread10 = [](uint32_t v) {
            v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu;
            v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u;
            return v;
        };
        std::vector<std::pair<uint32_t, int>> order;
        for (int m = 0; m < n_meshes; ++m) {
            uint32_t code = 0xFFFFFFFFu;
            if (meshes[m].n_tris > 0) {
                uint32_t q[3];
                for (int a = 0; a < 3; ++a) {
                    const float ext = hi[a] - lo[a];
                    const float u = ext > 0.f ? (ctr[(size_t)m * 3 + a] - lo[a]) / ext * 1024.f : 0.f;
                    q[a] = (uint32_t)std::min(std::max(u, 0.f), 1023.f);
                }
                code = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
            }
            order.push_back({code, m});
        }
        std::sort(order.begin(), order.end());
        std::vector<uint32_t> place((size_t)n_meshes);
        for (int k = 0; k < n_meshes; ++k) place[(size_t)order[(size_t)k].second] = (uint32_t)k;
        LB_TRY(hipMemcpy(d_place, place.data(), place.size() * 4, hipMemcpyHostToDevice));
        k_keys<<<blocks(n), B>>>(d_pos, d_mesh, d_place, n, d_bounds, d_keys, d_ids);
        LB_TRY(hipGetLastError());
    }
    {
        size_t bytes = 0;
        LB_TRY(rocprim::radix_sort_pairs(nullptr, bytes, d_keys, d_sorted, d_ids, d_ids_sorted, (size_t)n, 0, 63));
        void* d_tmp = nullptr;
        if (!tmp.get(&d_tmp, bytes, err)) return false;
        LB_TRY(rocprim::radix_sort_pairs(d_tmp, bytes, d_keys, d_sorted, d_ids, d_ids_sorted, (size_t)n, 0, 63));
    }
    DeviceTree t{};
    t.n_tris = n; t.n_nodes = n - 1;
    hipError_t e = hipMalloc((void**)&t.nodes, (size_t)(n - 1) * sizeof(Node64));
    if (e == hipSuccess) e = hipMalloc((void**)&t.tris, (size_t)n * sizeof(Tri48));
    if (e == hipSuccess) e = hipMalloc((void**)&t.info, (size_t)n * sizeof(TriInfo));
    if (e == hipSuccess) e = hipMalloc((void**)&t.nrm, (size_t)n * 36);
    if (e == hipSuccess) e = hipMemset(t.nodes, 0, (size_t)(n - 1) * sizeof(Node64));     // slots inside leaves stay unused
    if (e == hipSuccess) e = hipMemset(d_arrived, 0, (size_t)n * 4);
    if (e != hipSuccess) { free_device_tree(&t); *err = std::string("hipMalloc: ") + hipGetErrorString(e); return false; }
    int32_t* d_root_h = d_height + 2 * n;
    k_hierarchy<<<blocks(n - 1), B>>>(d_sorted, n, d_left, d_right, d_lo, d_hi, d_pi, d_pl);
    k_boxes<<<blocks(n), B>>>(d_pos, d_ids_sorted, n, d_left, d_lo, d_hi, d_pi, d_pl, t.nodes, d_arrived, d_height, d_root_h);
    k_emit<<<blocks(n), B>>>(d_pos, d_nrm, d_mesh, d_rank, d_ids_sorted, n, d_lo, d_hi, d_pi, d_pl, t.tris, t.info, t.nrm);
    int32_t root_h = 0;
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(&root_h, d_root_h, 4, hipMemcpyDeviceToHost);      // also waits for the kernels
    if (e != hipSuccess) { free_device_tree(&t); *err = std::string("device tree build: ") + hipGetErrorString(e); return false; }
    t.max_depth = root_h;
    *out = t;
    return true;
}

}  // namespace rt
