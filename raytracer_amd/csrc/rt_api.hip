// C-ABI of the library (declared and documented in include/rt_amd.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_amd.h"
#include "bvh_build.h"
#include "lbvh_build.h"
#include "rt_launch.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

bool is_device_ptr(const void* p) {
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// Makes `dev` the current device for the scope and puts the caller's device back afterwards.
struct DeviceGuard {
    int prev = -1; bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// No exception may cross the C boundary: every extern "C" body with host allocations runs inside these.
#define RT_API_BEGIN try {
#define RT_API_END(fn_)                                                                            \
    } catch (const std::bad_alloc&) { return fail(RT_ERR_INVALID, fn_ ": out of host memory");     \
    } catch (const std::exception& e_) { return fail(RT_ERR_INVALID, std::string(fn_ ": ") + e_.what()); \
    } catch (...) { return fail(RT_ERR_INVALID, fn_ ": unknown exception"); }

enum { RT_MAX_DEVICES = 64 };

// Staging buffers of rt_resolve_u8 for host images: one per device, grown on demand, kept until the process ends.
struct Staging { void* p = nullptr; size_t cap = 0; std::mutex mu; };
Staging g_staging[RT_MAX_DEVICES];

struct LevelBuf {
    rt::Level L{};
    int64_t cap = 0;
    int lights = 0;
    bool draws = false;    // carries the arrays of the two-pass replay
};

}  // namespace

#ifndef RT_TWO_PASS_MIN_SAMPLES
#define RT_TWO_PASS_MIN_SAMPLES 16
// most level-0 samples per band (log2) for scenes whose materials spawn rays: the levels below hold several times as
// many rays, and every band costs a host round trip per level
#ifndef RT_BAND_BUDGET_LOG2
#define RT_BAND_BUDGET_LOG2 27
#endif
#endif

enum { RT_N_COUNTERS = 16 + rt::RT_TRACED_SLOTS * rt::RT_TRACED_STRIDE * 2 };   // u32 words of the counter block
enum { RT_REDO_CAP = 1 << 20, RT_REDO_CAP_FIXED = 1 << 16 };   // entries per list of a scene's redo queues / of the fixed workspace's
// words of the counter block the redo queues use: count, ticket, running total per kernel kind (rt::RedoQueue)
enum { RT_W_REDO_EXT = 2, RT_W_REDO_SHA = 3, RT_W_TICKET_EXT = 12, RT_W_TICKET_SHA = 13, RT_W_TOTAL_EXT = 14, RT_W_TOTAL_SHA = 15 };

struct rt_scene {
    int device = 0;
    rt::DScene d{};                          // the reference's trees (exact_boxes)
    rt::DScene d_fast{};                     // the surface-area-heuristic tree of the pruning traversal
    int64_t n_tris = 0, n_inner = 0, n_leaves = 0, max_depth = 0;
    int64_t fast_nodes = 0; int fast_depth = 0; bool built_on_device = false;    // the pruning traversal's tree
    int n_lights = 0;
    bool any_spawning_material = false;
    std::vector<void*> owned;
    // render workspace, grown on demand; one render at a time per scene
    std::mutex mu;
    LevelBuf lv[rt::RT_MAX_LEVELS];
    uint32_t* d_counters = nullptr;          // [0] children of the level, [1] spawn cursor, [16...] light-ray slots (u64),
                                             // [4..7] extend nodes/tris (2 x u64), [8..11] shadow nodes/tris
    uint32_t* h_counters = nullptr;          // pinned mirror
    float* d_stage = nullptr; int64_t stage_cap = 0;
    float* d_share = nullptr; int64_t share_cap = 0;     // this GPU's rows of a frame rendered by several GPUs
    hipStream_t own_stream = nullptr;                    // rt_render_frame_multi: one stream per scene
    void* d_io = nullptr; int64_t io_cap = 0;  // staging for rt_trace / rt_resolve_u8 host buffers
    uint32_t redo_cap = 1u << 30;            // RT_AMD_DEBUG_REDO_CAP lowers the entries a redo list takes (tests of the overflow path)
    void* d_redo = nullptr;                  // lists of the rays the pruning traversal walks twice (rt::RedoQueue): RT_REDO_CAP
                                             // node numbers (u32), then as many (item, ray) entries (u64)
    uint32_t* d_fault = nullptr;             // set by a kernel whose traversal stack overflowed (cannot happen for a tree
                                             // that passed the depth check; read back by every synchronising call)
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [0..3] per-kernel timing, [4..5] whole call
    hipEvent_t ev_done = nullptr;            // end of the last call's work: the next call's stream waits for it before it
    int64_t level_limit = (int64_t)1 << 31;  // light-ray items one level may hold (rt_scene_set_level_limit)
    bool pending = false;                    // touches the workspace (calls may come on different streams)
};

// Fixed workspace of rt_render_rect_noalloc: one array per GPU, held from the library's load to the process's end.
// In host code the NAME below is only a handle for hipGetSymbolAddress; the device address it returns is what kernels
// and memsets get (scratch_address()).
enum : size_t { RT_SCRATCH_BYTES = (size_t)64 << 20 };
static __device__ __align__(256) unsigned char scratch[RT_SCRATCH_BYTES];

namespace {

template <class T>
int upload(rt_scene* sc, const T* src, size_t n, const T** dst) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
    sc->owned.push_back(p);
    if (n) HIP_TRY(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T*>(p);
    return RT_OK;
}

void free_level(LevelBuf& b) {
    if (b.L.dir) (void)hipFree(b.L.dir);
    if (b.L.meta) (void)hipFree(b.L.meta);
    if (b.L.rec) (void)hipFree(b.L.rec);
    if (b.L.geo) (void)hipFree(b.L.geo);
    if (b.L.hit) (void)hipFree(b.L.hit);
    if (b.L.pos) (void)hipFree(b.L.pos);
    if (b.L.tri) (void)hipFree(b.L.tri);
    if (b.L.kids) (void)hipFree(b.L.kids);
    if (b.L.aux) (void)hipFree(b.L.aux);
    if (b.L.unocc) (void)hipFree(b.L.unocc);
    if (b.L.draw) (void)hipFree(b.L.draw);
    if (b.L.first) (void)hipFree(b.L.first);
    b = LevelBuf();
}

// implicit: level 0, whose rays are recomputed instead of stored (no org/dir/meta arrays)
int ensure_level(LevelBuf& b, int64_t n, int lights, bool implicit, bool draws) {
    lights = std::max(lights, 1);
    if (b.cap >= n && b.lights >= lights && (b.draws || !draws)) return RT_OK;
    const int64_t cap = std::max<int64_t>(n + n / 4, 1024);
    free_level(b);
    b.L.implicit = implicit ? 1 : 0;
    if (!implicit) {
        HIP_TRY(hipMalloc((void**)&b.L.dir, cap * sizeof(float4)));
        HIP_TRY(hipMalloc((void**)&b.L.meta, cap * sizeof(int4)));
    }
    if (implicit) {
        HIP_TRY(hipMalloc((void**)&b.L.hit, cap * sizeof(float4)));
        HIP_TRY(hipMalloc((void**)&b.L.pos, cap * sizeof(float4)));
        HIP_TRY(hipMalloc((void**)&b.L.tri, cap * sizeof(int32_t)));
        HIP_TRY(hipMalloc((void**)&b.L.kids, cap * sizeof(int2)));
        HIP_TRY(hipMalloc((void**)&b.L.aux, cap * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&b.L.unocc, cap * lights * sizeof(uint8_t)));
    } else {
        HIP_TRY(hipMalloc((void**)&b.L.rec, cap * sizeof(rt::NodeRec)));
        HIP_TRY(hipMalloc((void**)&b.L.geo, cap * sizeof(rt::NodeGeo)));
    }
    if (draws) {
        HIP_TRY(hipMalloc((void**)&b.L.draw, cap * sizeof(rt::DrawRec)));
        if (implicit) HIP_TRY(hipMalloc((void**)&b.L.first, cap * sizeof(int2)));
    }
    b.cap = cap; b.lights = lights; b.draws = draws;
    return RT_OK;
}

int ensure_bytes(void** p, int64_t* cap, int64_t need) {
    if (*cap >= need) return RT_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    HIP_TRY(hipMalloc(p, (size_t)need));
    *cap = need;
    return RT_OK;
}

// Device address of `scratch` on the CURRENT device (every GPU holds its own copy at its own address), looked up once
// per device and kept.
unsigned char* g_scratch_ptr[RT_MAX_DEVICES];
std::mutex g_scratch_mu;
// Every call that uses a GPU's fixed workspace holds `use` of that GPU while it enqueues its work and makes its stream
// wait for `done`, the end of the previous such call on that GPU - whatever scene handle, stream or thread that came
// from.  The event is created with the address (set-up, not the calls, may allocate).
struct ScratchOrder { std::mutex use; hipEvent_t done = nullptr; bool pending = false; };
ScratchOrder g_scratch_order[RT_MAX_DEVICES];
int scratch_address(int dev, unsigned char** out) {
    if (dev < 0 || dev >= RT_MAX_DEVICES) return fail(RT_ERR_UNSUPPORTED, "device index beyond RT_MAX_DEVICES");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(RT_ERR_HIP, "hipGetSymbolAddress returned no address for the fixed workspace");
        HIP_TRY(hipEventCreateWithFlags(&g_scratch_order[dev].done, hipEventDisableTiming));
        g_scratch_ptr[dev] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[dev];
    return RT_OK;
}

// true while `s` records into a HIP graph instead of running: events of the library must then be neither waited for
// nor recorded (a recorded one would become a node of the caller's graph, and a later plain call waiting for it would
// be drawn into the capture)
bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st == hipStreamCaptureStatusActive;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }

int rt_init(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device visible");
    if (device >= 0) {
        if (device >= n) return fail(RT_ERR_INVALID, "device index out of range");
        HIP_TRY(hipSetDevice(device));
    }
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cur));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(RT_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    unsigned char* sp = nullptr;                 // the fixed workspace of rt_render_rect_noalloc on this device
    return scratch_address(cur, &sp);
}

int rt_scene_create(const rt_mesh_desc* meshes, int n_meshes, const rt_material_desc* materials, int n_materials,
                    const rt_light_desc* lights, int n_lights, const rt_camera_desc* camera, rt_scene** out) {
    // RT_AMD_BUILDER=device (environment): the pruning traversal's tree is built on the GPU
    const char* b = getenv("RT_AMD_BUILDER");
    return rt_scene_create_ex(meshes, n_meshes, materials, n_materials, lights, n_lights, camera,
                              (b && strcmp(b, "device") == 0) ? RT_BUILD_TREE_ON_DEVICE : 0u, out);
}

int rt_scene_create_ex(const rt_mesh_desc* meshes, int n_meshes, const rt_material_desc* materials, int n_materials,
                       const rt_light_desc* lights, int n_lights, const rt_camera_desc* camera, uint32_t flags, rt_scene** out) {
    if (!meshes || n_meshes <= 0 || !materials || n_materials <= 0 || !camera || !out || n_lights < 0 || (n_lights > 0 && !lights))
        return fail(RT_ERR_INVALID, "rt_scene_create: null or empty argument");
    if (n_materials >= (1 << 22) - 1) return fail(RT_ERR_UNSUPPORTED, "more than 4194302 materials (a stored ray keeps the material it travels in in 22 bits)");
    RT_API_BEGIN
    if (n_lights > RT_MAX_LIGHTS) return fail(RT_ERR_UNSUPPORTED, "more lights than RT_MAX_LIGHTS");
    for (int i = 0; i < n_meshes; ++i) {
        if (meshes[i].material < 0 || meshes[i].material >= n_materials) return fail(RT_ERR_INVALID, "mesh material index out of range");
        for (int j = 0; j < i; ++j)
            if (strncmp(meshes[i].name, meshes[j].name, 32) == 0) return fail(RT_ERR_INVALID, "mesh names must be unique");
    }
    rt::FlatScene flat;
    std::string err;
    if (!rt::build_flat_scene(meshes, n_meshes, &flat, &err)) return fail(RT_ERR_INVALID, err);
    if (flat.max_depth + 2 > RT_STACK) return fail(RT_ERR_UNSUPPORTED, "tree deeper than the traversal stack");
    // the tree the pruning traversal walks: surface-area heuristic on host threads, or (RT_BUILD_TREE_ON_DEVICE, scenes
    // of more than a handful of triangles) a linear hierarchy built on the GPU
    bool on_device = (flags & RT_BUILD_TREE_ON_DEVICE) != 0 && flat.tris.size() > 16;
    rt::FlatScene fast;
    auto host_tree = [&]() -> int {
        if (!rt::build_sah_scene(meshes, n_meshes, flat, &fast, &err)) return fail(RT_ERR_INVALID, err);
        if (fast.max_depth + 2 > RT_STACK) return fail(RT_ERR_UNSUPPORTED, "tree deeper than the traversal stack");
        return RT_OK;
    };
    int rc;
    if (!on_device && (rc = host_tree()) != RT_OK) return rc;

    rt_scene* sc = new rt_scene();
    (void)hipGetDevice(&sc->device);
    std::vector<rt::Material> mats(n_materials);
    for (int i = 0; i < n_materials; ++i) {
        static_assert(sizeof(rt::Material) == sizeof(rt_material_desc), "material layouts must match");
        memcpy(&mats[i], &materials[i], sizeof(rt::Material));
        if (materials[i].kind == RT_MAT_PHONG || materials[i].kind == RT_MAT_REFRAC) sc->any_spawning_material = true;
    }
    std::vector<int32_t> mesh_mat(n_meshes);
    for (int i = 0; i < n_meshes; ++i) mesh_mat[i] = meshes[i].material;
    if ((rc = upload(sc, flat.nodes.data(), flat.nodes.size(), &sc->d.nodes)) != RT_OK ||
        (rc = upload(sc, flat.tris.data(), flat.tris.size(), &sc->d.tris)) != RT_OK ||
        (rc = upload(sc, flat.info.data(), flat.info.size(), &sc->d.info)) != RT_OK ||
        (rc = upload(sc, flat.nrm.data(), flat.nrm.size(), &sc->d.nrm)) != RT_OK ||
        (rc = upload(sc, mats.data(), mats.size(), &sc->d.mats)) != RT_OK ||
        (rc = upload(sc, mesh_mat.data(), mesh_mat.size(), &sc->d.mesh_mat)) != RT_OK) {
        rt_scene_destroy(sc);
        return rc;
    }
    sc->d_fast = sc->d;
    if (on_device) {
        std::vector<int32_t> rank_of(flat.tris.size());
        for (size_t k = 0; k < flat.tris.size(); ++k) rank_of[(size_t)flat.info[k].id] = flat.tris[k].rank;
        rt::DeviceTree dt;
        const auto tb0 = std::chrono::steady_clock::now();
        bool built = rt::build_lbvh_device(meshes, n_meshes, rank_of.data(), (int64_t)flat.tris.size(), &dt, &err);
        if (getenv("RT_AMD_DEBUG_TREE"))
            fprintf(stderr, "rt_amd: device tree build incl. upload of the triangles %.3f s%s%s\n",
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count(), built ? "" : " - FAILED: ", built ? "" : err.c_str());
        // A linear hierarchy is as deep as its keys are alike (many triangles at one spot: up to the key's 63 bits plus
        // log2 of the duplicates), and the build needs memory of its own.  Neither makes the scene unusable: the host
        // builder takes over - a slower build, not an error.
        // (RT_AMD_DEBUG_FAIL_DEVICE_BUILD in the environment throws a finished device tree away: the only way to
        // exercise the take-over on a GPU with memory to spare)
        if (built && (dt.max_depth + 2 > RT_STACK || getenv("RT_AMD_DEBUG_FAIL_DEVICE_BUILD"))) { rt::free_device_tree(&dt); built = false; }
        if (!built) {
            (void)hipGetLastError();
            on_device = false;
            if ((rc = host_tree()) != RT_OK) { rt_scene_destroy(sc); return rc; }
        } else {
            sc->owned.push_back(dt.nodes); sc->owned.push_back(dt.tris); sc->owned.push_back(dt.info); sc->owned.push_back(dt.nrm);
            sc->d_fast.nodes = dt.nodes; sc->d_fast.tris = dt.tris; sc->d_fast.info = dt.info; sc->d_fast.nrm = dt.nrm;
            sc->fast_nodes = dt.n_nodes; sc->fast_depth = dt.max_depth;
        }
    }
    if (!on_device) {
        if ((rc = upload(sc, fast.nodes.data(), fast.nodes.size(), &sc->d_fast.nodes)) != RT_OK ||
            (rc = upload(sc, fast.tris.data(), fast.tris.size(), &sc->d_fast.tris)) != RT_OK ||
            (rc = upload(sc, fast.info.data(), fast.info.size(), &sc->d_fast.info)) != RT_OK ||
            (rc = upload(sc, fast.nrm.data(), fast.nrm.size(), &sc->d_fast.nrm)) != RT_OK) {
            rt_scene_destroy(sc);
            return rc;
        }
        sc->fast_nodes = (int64_t)fast.nodes.size(); sc->fast_depth = fast.max_depth;
    }
    sc->built_on_device = on_device;
    rt::launch_pack_refs(const_cast<rt::Node64*>(sc->d_fast.nodes), sc->fast_nodes, nullptr);
    {
        // The reference's leaves and their boxes, out of its flattened trees: a leaf hangs in exactly one child slot of
        // one node, with the box the reference tests before it looks at the leaf's triangles.  The pruning traversal vets
        // its candidates against these (rt_device.h ref_leaf_box_hit): box table on the device, and every triangle record
        // of the pruning traversal's tree told the number of its leaf.
        std::vector<float> boxes;
        std::vector<int32_t> leaf_of_id(flat.tris.size(), 0);
        int32_t n_leaf = 0;
        // (walked from the entry node along the reference's own child references: the refined leaves' subtrees, appended to
        // the same node array, are not the reference's)
        std::vector<int32_t> todo{0};
        while (!todo.empty()) {
            const rt::Node64& n = flat.nodes[(size_t)todo.back()];
            todo.pop_back();
            for (int side = 0; side < 2; ++side) {
                const int32_t ref = side ? n.rref : n.lref;
                if (ref == rt::RT_NONE) continue;
                if (ref >= 0) {
                    if ((size_t)ref >= flat.nodes.size() || todo.size() > flat.nodes.size()) { rt_scene_destroy(sc); return fail(RT_ERR_INVALID, "internal: node reference out of range"); }
                    todo.push_back(ref);
                    continue;
                }
                const float* bx = side ? n.rbox : n.lbox;
                boxes.insert(boxes.end(), bx, bx + 6); boxes.push_back(0.f); boxes.push_back(0.f);
                const size_t first = (size_t)~ref;
                if (first >= flat.tris.size()) { rt_scene_destroy(sc); return fail(RT_ERR_INVALID, "internal: leaf reference out of range"); }
                for (size_t k = first, e = first + (size_t)flat.tris[first].run; k < e && k < flat.tris.size(); ++k) leaf_of_id[(size_t)flat.info[k].id] = n_leaf;
                ++n_leaf;
            }
        }
        const float* d_boxes = nullptr; const int32_t* d_leaf = nullptr;
        if ((rc = upload(sc, boxes.data(), boxes.size(), &d_boxes)) != RT_OK || (rc = upload(sc, leaf_of_id.data(), leaf_of_id.size(), &d_leaf)) != RT_OK) {
            rt_scene_destroy(sc);
            return rc;
        }
        sc->d_fast.leaf_box = reinterpret_cast<const float4*>(d_boxes);
        rt::launch_set_ref_leaf(const_cast<rt::Tri48*>(sc->d_fast.tris), sc->d_fast.info, d_leaf, (int64_t)flat.tris.size(), nullptr);
    }
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { rt_scene_destroy(sc); return fail(RT_ERR_HIP, "packing the tree's references failed"); }
    const rt::DScene fast_ptrs = sc->d_fast;
    sc->d.n_lights = n_lights;
    for (int i = 0; i < n_lights; ++i) {
        static_assert(sizeof(rt::Light) == sizeof(rt_light_desc), "light layouts must match");
        memcpy(&sc->d.lights[i], &lights[i], sizeof(rt::Light));
    }
    memcpy(sc->d.cam.pos, camera->pos, 12); memcpy(sc->d.cam.up, camera->up, 12);
    memcpy(sc->d.cam.w, camera->w, 12); memcpy(sc->d.cam.side, camera->side, 12);
    sc->d.cam.near_clip = camera->near_clip; sc->d.cam.far_clip = camera->far_clip;
    sc->d_fast = sc->d;
    sc->d_fast.nodes = fast_ptrs.nodes; sc->d_fast.tris = fast_ptrs.tris; sc->d_fast.info = fast_ptrs.info; sc->d_fast.nrm = fast_ptrs.nrm;
    sc->d_fast.leaf_box = fast_ptrs.leaf_box;
    sc->n_lights = n_lights;
    sc->n_tris = (int64_t)flat.tris.size();
    sc->n_inner = (int64_t)flat.nodes.size();
    sc->n_leaves = flat.n_leaves;
    sc->max_depth = flat.max_depth;
    hipError_t e = hipMalloc((void**)&sc->d_counters, RT_N_COUNTERS * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(sc->d_counters, 0, RT_N_COUNTERS * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&sc->d_redo, (size_t)RT_REDO_CAP * 12);
    if (e == hipSuccess) e = hipMalloc((void**)&sc->d_fault, 256);
    if (e == hipSuccess) e = hipMemset(sc->d_fault, 0, 256);
    if (e == hipSuccess) e = hipHostMalloc((void**)&sc->h_counters, (RT_N_COUNTERS + 16) * sizeof(uint32_t));
    for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipEventCreate(&sc->ev[i]);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sc->ev_done, hipEventDisableTiming);
    if (e != hipSuccess) { rt_scene_destroy(sc); return fail(RT_ERR_HIP, hipGetErrorString(e)); }
    sc->d.fault = sc->d_fault; sc->d_fast.fault = sc->d_fault;
    // a stack never holds more entries than its tree has levels: such a tree takes the kernels whose stack is LDS only
    if (const char* cap = getenv("RT_AMD_DEBUG_REDO_CAP")) sc->redo_cap = (uint32_t)std::max(1, atoi(cap));
    sc->d.shallow = 0; sc->d_fast.shallow = (sc->fast_depth <= RT_LDS_SLOTS && !getenv("RT_AMD_DEBUG_DEEP_STACK")) ? 1 : 0;
    *out = sc;
    return RT_OK;
    RT_API_END("rt_scene_create")
}

// Invariants of a flattened tree (which != 0: the refined reference pair and run lengths): every triangle in exactly one
// leaf, every child box holds its subtree's vertices and its children's boxes, child references in range, tie ranks a
// permutation, depth within the traversal stack.  out as documented for rt_tree_check.
// packed: the tree as a scene holds it on the device - its planes carry reference bytes (rt_device.h packed_ref) and have
// each moved outwards by up to 511 units in the last place on their own, so a child's box may stick out of its parent's by
// that much; the packed references must equal the stored ones.
static uint32_t float_key(float f) { uint32_t b; memcpy(&b, &f, 4); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
static int verify_tree(const rt::FlatScene& f, int which, const rt_mesh_desc* meshes, int n_meshes, int64_t out[8], bool packed = false) {
    const int64_t nt = (int64_t)f.tris.size(), nn = (int64_t)f.nodes.size();
    if ((int64_t)f.info.size() != nt || (int64_t)f.nrm.size() != nt * 9) return fail(RT_ERR_INVALID, "tree check: array sizes differ");
    // original vertices by global triangle id
    std::vector<const float*> vert((size_t)nt);
    { int64_t id = 0; for (int m = 0; m < n_meshes; ++m) for (int t = 0; t < meshes[m].n_tris; ++t) vert[(size_t)id++] = meshes[m].pos + (size_t)t * 9; }
    std::vector<uint8_t> seen_tri((size_t)nt, 0), seen_rank((size_t)nt, 0), seen_node((size_t)nn, 0);
    int64_t reached_tris = 0, reached_nodes = 0, leaves = 0, max_leaf = 0, max_depth = 0;
    struct Item { int32_t ref; float box[6]; int depth; };
    std::vector<Item> todo;
    auto push_children = [&](int32_t node, int depth) -> bool {
        if (node < 0 || node >= nn || seen_node[(size_t)node]) return false;
        seen_node[(size_t)node] = 1; ++reached_nodes;
        const rt::Node64& n = f.nodes[(size_t)node];
        const int32_t refs[2] = {which ? n.lref_fine : n.lref, which ? n.rref_fine : n.rref};
        for (int s = 0; s < 2; ++s) {
            if (refs[s] == rt::RT_NONE) continue;
            Item it; it.ref = refs[s]; it.depth = depth + 1;
            memcpy(it.box, s ? n.rbox : n.lbox, 24);
            todo.push_back(it);
        }
        return true;
    };
    if (!push_children(0, 0)) return fail(RT_ERR_INVALID, "tree check: no entry node");
    while (!todo.empty()) {
        const Item it = todo.back(); todo.pop_back();
        max_depth = std::max<int64_t>(max_depth, it.depth);
        if (it.ref >= 0) {
            // a child box must contain the boxes of the child's own children
            if (it.ref >= nn) return fail(RT_ERR_INVALID, "tree check: node reference out of range");
            const rt::Node64& c = f.nodes[(size_t)it.ref];
            const int32_t r2[2] = {which ? c.lref_fine : c.lref, which ? c.rref_fine : c.rref};
            for (int s = 0; s < 2; ++s) {
                if (r2[s] == rt::RT_NONE) continue;
                const float* b = s ? c.rbox : c.lbox;
                const int64_t slack = packed ? 1024 : 0;          // in units in the last place
                for (int i = 0; i < 3; ++i)
                    if ((int64_t)float_key(b[i]) + slack < (int64_t)float_key(it.box[i]) || (int64_t)float_key(b[3 + i]) > (int64_t)float_key(it.box[3 + i]) + slack)
                        return fail(RT_ERR_INVALID, "tree check: a child box sticks out of its parent's");
                if (packed) {
                    uint32_t ref = 0;
                    for (int k = 0; k < 4; ++k) { uint32_t w; memcpy(&w, &b[k], 4); ref |= (w & 0xffu) << (8 * k); }
                    if ((int32_t)ref != r2[s]) return fail(RT_ERR_INVALID, "tree check: a reference packed into the box planes differs from the stored one");
                }
            }
            if (!push_children(it.ref, it.depth)) return fail(RT_ERR_INVALID, "tree check: node reached twice or out of range");
            continue;
        }
        const int64_t first = ~it.ref;
        if (first < 0 || first >= nt) return fail(RT_ERR_INVALID, "tree check: leaf reference out of range");
        const int64_t run = which ? f.tris[(size_t)first].run_fine : f.tris[(size_t)first].run;
        if (run <= 0 || first + run > nt) return fail(RT_ERR_INVALID, "tree check: leaf run out of range");
        ++leaves; max_leaf = std::max(max_leaf, run);
        for (int64_t k = first; k < first + run; ++k) {
            if (seen_tri[(size_t)k]) return fail(RT_ERR_INVALID, "tree check: triangle in two leaves");
            seen_tri[(size_t)k] = 1; ++reached_tris;
            const int64_t left = which ? f.tris[(size_t)k].run_fine : f.tris[(size_t)k].run;
            if (left != first + run - k) return fail(RT_ERR_INVALID, "tree check: run lengths inside a leaf do not count down");
            const int32_t id = f.info[(size_t)k].id, rank = f.tris[(size_t)k].rank;
            if (id < 0 || id >= nt || rank < 0 || rank >= nt || seen_rank[(size_t)rank]) return fail(RT_ERR_INVALID, "tree check: ids or tie ranks are not a permutation");
            seen_rank[(size_t)rank] = 1;
            const float* v = vert[(size_t)id];
            if (f.tris[(size_t)k].ax != v[0] || f.tris[(size_t)k].ay != v[1] || f.tris[(size_t)k].az != v[2]) return fail(RT_ERR_INVALID, "tree check: triangle does not match its id");
            for (int c3 = 0; c3 < 3; ++c3)
                for (int i = 0; i < 3; ++i)
                    if (v[c3 * 3 + i] < it.box[i] || v[c3 * 3 + i] > it.box[3 + i]) return fail(RT_ERR_INVALID, "tree check: a vertex lies outside its leaf box");
        }
    }
    if (reached_tris != nt) return fail(RT_ERR_INVALID, "tree check: not every triangle is reachable");
    if (max_depth + 2 > RT_STACK) return fail(RT_ERR_UNSUPPORTED, "tree deeper than the traversal stack");
    out[0] = nt; out[1] = nn; out[2] = leaves; out[3] = max_depth; out[4] = max_leaf; out[5] = reached_tris; out[6] = reached_nodes; out[7] = 0;
    return RT_OK;
}

int rt_tree_check(const rt_mesh_desc* meshes, int n_meshes, int which, int64_t out[8]) {
    if (!meshes || n_meshes <= 0 || !out || (which != 0 && which != 1)) return fail(RT_ERR_INVALID, "rt_tree_check: bad argument");
    RT_API_BEGIN
    rt::FlatScene flat, fast;
    std::string err;
    const bool dbg = getenv("RT_AMD_DEBUG_TREE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    if (!rt::build_flat_scene(meshes, n_meshes, &flat, &err)) return fail(RT_ERR_INVALID, err);
    const double t1 = now();
    if (which == 1 && !rt::build_sah_scene(meshes, n_meshes, flat, &fast, &err)) return fail(RT_ERR_INVALID, err);
    if (dbg) fprintf(stderr, "rt_amd: reference trees %.2f s, surface-area trees %.2f s\n", t1 - t0, now() - t1);
    return verify_tree(which ? fast : flat, which, meshes, n_meshes, out);
    RT_API_END("rt_tree_check")
}

int rt_scene_destroy(rt_scene* sc) {
    if (!sc) return RT_OK;
    DeviceGuard dg(sc->device);
    if (sc->pending && sc->ev_done) (void)hipEventSynchronize(sc->ev_done);    // work of the last call may still be running
    for (void* p : sc->owned) (void)hipFree(p);
    for (auto& b : sc->lv) free_level(b);
    if (sc->d_counters) (void)hipFree(sc->d_counters);
    if (sc->d_fault) (void)hipFree(sc->d_fault);
    if (sc->d_redo) (void)hipFree(sc->d_redo);
    if (sc->ev_done) (void)hipEventDestroy(sc->ev_done);
    if (sc->h_counters) (void)hipHostFree(sc->h_counters);
    if (sc->d_stage) (void)hipFree(sc->d_stage);
    if (sc->d_share) (void)hipFree(sc->d_share);
    if (sc->own_stream) (void)hipStreamDestroy(sc->own_stream);
    if (sc->d_io) (void)hipFree(sc->d_io);
    for (auto& e : sc->ev) if (e) (void)hipEventDestroy(e);
    delete sc;
    return RT_OK;
}

int rt_scene_check_tree(rt_scene* sc, const rt_mesh_desc* meshes, int n_meshes, int64_t out[8]) {
    if (!sc || !meshes || n_meshes <= 0 || !out) return fail(RT_ERR_INVALID, "rt_scene_check_tree: null argument");
    DeviceGuard dg(sc->device);
    if (!dg.ok) return fail(RT_ERR_HIP, "rt_scene_check_tree: cannot make the scene's device current");
    RT_API_BEGIN
    rt::FlatScene f;
    f.nodes.resize((size_t)sc->fast_nodes); f.tris.resize((size_t)sc->n_tris); f.info.resize((size_t)sc->n_tris); f.nrm.resize((size_t)sc->n_tris * 9);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(f.nodes.data(), sc->d_fast.nodes, f.nodes.size() * sizeof(rt::Node64), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(f.tris.data(), sc->d_fast.tris, f.tris.size() * sizeof(rt::Tri48), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(f.info.data(), sc->d_fast.info, f.info.size() * sizeof(rt::TriInfo), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(f.nrm.data(), sc->d_fast.nrm, f.nrm.size() * sizeof(float), hipMemcpyDeviceToHost));
    const int rc = verify_tree(f, 1, meshes, n_meshes, out, /*packed*/ true);
    if (rc != RT_OK) return rc;
    if (out[3] != sc->fast_depth) return fail(RT_ERR_INVALID, "tree check: the recorded depth differs from the tree's");
    out[7] = sc->built_on_device ? 1 : 0;
    return RT_OK;
    RT_API_END("rt_scene_check_tree")
}


int rt_scene_set_level_limit(rt_scene* sc, int64_t max_items) {
    if (!sc) return fail(RT_ERR_INVALID, "rt_scene_set_level_limit: null scene");
    if (max_items < 0 || max_items > ((int64_t)1 << 31)) return fail(RT_ERR_INVALID, "rt_scene_set_level_limit: 0 (default) .. 2^31");
    std::lock_guard<std::mutex> lock(sc->mu);
    sc->level_limit = max_items ? max_items : (int64_t)1 << 31;
    return RT_OK;
}

int rt_scene_info(const rt_scene* sc, int64_t counts[4]) {
    if (!sc || !counts) return fail(RT_ERR_INVALID, "rt_scene_info: null argument");
    counts[0] = sc->n_tris; counts[1] = sc->n_inner; counts[2] = sc->n_leaves; counts[3] = sc->max_depth;
    return RT_OK;
}

// Start and end of a call's use of the scene's workspace.  Calls on one handle may come on different streams: each
// call's stream first waits for the end of the previous call's work, recorded in ev_done.
static int workspace_begin(rt_scene* sc, hipStream_t s) {
    if (sc->pending && !stream_is_capturing(s)) HIP_TRY(hipStreamWaitEvent(s, sc->ev_done, 0));
    return RT_OK;
}
// Created once workspace_begin has succeeded: whichever way the call then leaves - the last line or an error return
// after some work was enqueued - the end of what it enqueued on `s` is recorded, so that the next call on the handle
// (possibly on another stream) is ordered behind it.  order != NULL: the call also used the GPU's fixed workspace.
struct WorkspaceScope {
    rt_scene* sc; hipStream_t s; ScratchOrder* order;
    WorkspaceScope(rt_scene* sc_, hipStream_t s_, ScratchOrder* order_ = nullptr) : sc(sc_), s(s_), order(order_) {}
    WorkspaceScope(const WorkspaceScope&) = delete;
    ~WorkspaceScope() {
        if (stream_is_capturing(s)) return;
        if (hipEventRecord(sc->ev_done, s) == hipSuccess) sc->pending = true; else (void)hipGetLastError();
        if (order && order->done) { if (hipEventRecord(order->done, s) == hipSuccess) order->pending = true; else (void)hipGetLastError(); }
    }
};
// Set when a call ends because its redo queue overflowed: the entry points that render then run the call again in the
// reference-arithmetic mode, whose frame the vetted pruning mode reproduces anyway.
static thread_local bool g_redo_overflow = false;
// The scene's fault word, fetched together with other results a call waits for anyway.
static int check_fault(rt_scene* sc, hipStream_t s) {
    uint32_t* h = sc->h_counters + RT_N_COUNTERS;
    HIP_TRY(hipMemcpyAsync(h, sc->d_fault, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (*h) {
        const uint32_t what = *h;
        (void)hipMemsetAsync(sc->d_fault, 0, sizeof(uint32_t), s);      // reported once; the next call starts clean
        if (what == 2u) {
            g_redo_overflow = true;
            return fail(RT_ERR_UNSUPPORTED, "more rays needed a second, vetted walk than the redo queue holds (a scene of needles?); "
                                            "render with exact_boxes = 1");
        }
        return fail(RT_ERR_UNSUPPORTED, "a traversal stack overflowed: tree deeper than the traversal stack");
    }
    return RT_OK;
}

static int trace_impl(rt_scene* sc, const float* org, const float* dir, float t0, float t1, int64_t n, int any_hit, int exact,
                      float* t_out, int32_t* tri_out, float* nrm_out, uint32_t* nodes_out, uint32_t* tris_out, void* stream) {
    if (!sc || n < 0 || (n > 0 && (!org || !dir || !t_out))) return fail(RT_ERR_INVALID, "rt_trace: null argument or negative count");
    if (exact < 0 || exact > 2) return fail(RT_ERR_INVALID, "rt_trace: exact_boxes is 0, 1 or 2");
    if (n == 0) return RT_OK;
    hipStream_t s = (hipStream_t)stream;
    DeviceGuard dg(sc->device);
    if (!dg.ok) return fail(RT_ERR_HIP, "rt_trace: cannot make the scene's device current");
    std::lock_guard<std::mutex> lock(sc->mu);
    const bool dev = is_device_ptr(org);
    if (dev != is_device_ptr(dir) || dev != is_device_ptr(t_out)) return fail(RT_ERR_INVALID, "rt_trace: mix of host and device buffers");
    const float *d_org = org, *d_dir = dir;
    float *d_t = t_out, *d_n = nrm_out; int32_t* d_tri = tri_out; uint32_t *d_cn = nodes_out, *d_ct = tris_out;
    int rc = workspace_begin(sc, s);
    if (rc != RT_OK) return rc;
    WorkspaceScope scope(sc, s);
    if (!dev) {
        // layout of the staging block: org[3n] dir[3n] t[n] nrm[3n] tri[n] cn[n] ct[n]
        rc = ensure_bytes(&sc->d_io, &sc->io_cap, n * 13 * 4);
        if (rc != RT_OK) return rc;
        float* base = (float*)sc->d_io;
        HIP_TRY(hipMemcpyAsync(base, org, n * 12, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(base + 3 * n, dir, n * 12, hipMemcpyHostToDevice, s));
        d_org = base; d_dir = base + 3 * n; d_t = base + 6 * n;
        d_n = nrm_out ? base + 7 * n : nullptr;
        d_tri = tri_out ? (int32_t*)(base + 10 * n) : nullptr;
        d_cn = nodes_out ? (uint32_t*)(base + 11 * n) : nullptr;
        d_ct = tris_out ? (uint32_t*)(base + 12 * n) : nullptr;
    }
    rt::launch_trace(any_hit != 0, exact == 1, exact == 0, exact == 1 ? sc->d : sc->d_fast, d_org, d_dir, t0, t1, n, d_t, d_tri, d_n, d_cn, d_ct, s);
    HIP_TRY(hipGetLastError());
    if (!dev) {
        HIP_TRY(hipMemcpyAsync(t_out, d_t, n * 4, hipMemcpyDeviceToHost, s));
        if (nrm_out) HIP_TRY(hipMemcpyAsync(nrm_out, d_n, n * 12, hipMemcpyDeviceToHost, s));
        if (tri_out) HIP_TRY(hipMemcpyAsync(tri_out, d_tri, n * 4, hipMemcpyDeviceToHost, s));
        if (nodes_out) HIP_TRY(hipMemcpyAsync(nodes_out, d_cn, n * 4, hipMemcpyDeviceToHost, s));
        if (tris_out) HIP_TRY(hipMemcpyAsync(tris_out, d_ct, n * 4, hipMemcpyDeviceToHost, s));
        if ((rc = check_fault(sc, s)) != RT_OK) return rc;       // synchronises the stream
    }
    return RT_OK;
}

int rt_trace(rt_scene* sc, const float* org, const float* dir, float t0, float t1, int64_t n, int any_hit, int exact_boxes,
             float* t_out, int32_t* tri_out, float* nrm_out, void* stream) {
    RT_API_BEGIN
    return trace_impl(sc, org, dir, t0, t1, n, any_hit, exact_boxes, t_out, tri_out, nrm_out, nullptr, nullptr, stream);
    RT_API_END("rt_trace")
}

int rt_trace_counted(rt_scene* sc, const float* org, const float* dir, float t0, float t1, int64_t n, int any_hit, int exact_boxes,
                     float* t_out, int32_t* tri_out, uint32_t* nodes_out, uint32_t* tris_out, void* stream) {
    RT_API_BEGIN
    return trace_impl(sc, org, dir, t0, t1, n, any_hit, exact_boxes, t_out, tri_out, nullptr, nodes_out, tris_out, stream);
    RT_API_END("rt_trace_counted")
}

// Where the first level's arrays and the counter block live when the caller may not allocate: carved out of one
// fixed block (rt_render_rect_noalloc) instead of the scene's grow-on-demand buffers.
struct FixedWorkspace { unsigned char* base; size_t bytes; };

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// rt_render_rect and rt_render_rect_noalloc (fixed != NULL).
static int render_impl(rt_scene* sc, const rt_options* o, int x0, int y0, int x1, int y1, float* rgb_out, rt_stats* stats,
                       void* stream, const FixedWorkspace* fixed, ScratchOrder* order = nullptr) {
    if (!sc || !o || !rgb_out) return fail(RT_ERR_INVALID, "rt_render_rect: null argument");
    if (o->width <= 0 || o->height <= 0 || x0 < 0 || y0 < 0 || x1 > o->width || x0 > x1 || y0 > y1)
        return fail(RT_ERR_INVALID, "rt_render_rect: rectangle outside the image");
    if (o->row_stride > 1) {
        if (o->row_band < 1 || o->row_phase < 0 || o->row_phase >= o->row_stride)
            return fail(RT_ERR_INVALID, "rt_render_rect: bad row interleave (row_band >= 1, 0 <= row_phase < row_stride)");
        if (y1 > y0) {      // the last local row must map inside the image
            const int l = y1 - 1;
            if (((l / o->row_band) * o->row_stride + o->row_phase) * o->row_band + l % o->row_band >= o->height)
                return fail(RT_ERR_INVALID, "rt_render_rect: local rows reach past the image");
        }
    } else if (y1 > o->height) return fail(RT_ERR_INVALID, "rt_render_rect: rectangle outside the image");
    if (o->exact_boxes < 0 || o->exact_boxes > 2) return fail(RT_ERR_INVALID, "rt_render_rect: exact_boxes is 0, 1 or 2");
    if (o->spp < 1 || o->spp > 16 || o->spr < 0 || o->spr > 15 || o->max_bounce < 0 || o->max_bounce > 8 ||
        o->max_refrac_bounce < 0 || o->max_refrac_bounce > 8)
        return fail(RT_ERR_UNSUPPORTED, "rt_render_rect: spp 1..16, spr 0..15, bounce and rbounce 0..8");
    if (o->max_bounce + std::max(o->max_bounce, o->max_refrac_bounce) + 1 > rt::RT_MAX_LEVELS)
        return fail(RT_ERR_UNSUPPORTED, "rt_render_rect: bounce depths exceed the level limit");
    if (stats) memset(stats, 0, sizeof(*stats));
    const int rw = x1 - x0, rh = y1 - y0;
    if (rw == 0 || rh == 0) return RT_OK;
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(sc->mu);

    rt::RenderParams P{};
    P.W = o->width; P.H = o->height; P.x0 = x0; P.y0 = y0; P.rw = rw; P.rh = rh;
    P.spp = o->spp; P.spr = o->spr; P.D = o->max_bounce; P.Dr = o->max_refrac_bounce;
    P.eps = o->eps; P.ray_eps = o->ray_eps; P.jitter = o->jitter_mode; P.seed = o->seed;
    P.row_stride = o->row_stride; P.row_band = o->row_band; P.row_phase = o->row_phase;
    P.can_spawn = sc->any_spawning_material ? 1 : 0;
    P.skip_unlit = (o->skip_unlit != 0 && !sc->any_spawning_material) ? 1 : 0;
    rt::set_divisors(P, sc->n_lights);
    {   // src/render_thread.cpp:34-40 (PI is the float 3.1415926, :3)
        const float aspect_ratio = o->height / (float)o->width;
        const float d = sc->d.cam.near_clip;
        const float a = d * tanf(3.1415926f * o->hfov / 360);
        P.fl = -a; P.fr = a; P.ft = a * aspect_ratio; P.fb = -P.ft; P.fd = d;
    }
    const bool exact = o->exact_boxes == 1;
    const bool vetted = o->exact_boxes == 0;           // 2: the pruning traversal without the second, vetted walk of suspect rays
    const bool counting = o->count_traversal != 0;
    uint64_t work[4] = {0, 0, 0, 0};
    const int spp2 = o->spp * o->spp;
    const int nl = sc->n_lights;
    const bool out_dev = is_device_ptr(rgb_out);

    // band height: a level-0 budget of samples; scenes that can spawn get a smaller one because levels multiply
    int64_t budget = (int64_t)1 << 24;
    uint32_t* d_counters = sc->d_counters;
    const size_t counter_bytes = align_up(RT_N_COUNTERS * sizeof(uint32_t), 256);
    const size_t slot_bytes = 16 + 16 + 4 + (size_t)std::max(nl, 1);           // hit, pos, tri, unocc per level-0 slot
    const size_t fixed_redo_bytes = (size_t)RT_REDO_CAP_FIXED * 12;
    if (fixed) {
        if (sc->any_spawning_material)
            return fail(RT_ERR_UNSUPPORTED, "rt_render_rect_noalloc: the scene has reflective or refractive materials, whose ray "
                                            "queues grow with the frame; use rt_render_rect");
        if (!out_dev) return fail(RT_ERR_INVALID, "rt_render_rect_noalloc: rgb_out must be device memory");
        d_counters = (uint32_t*)fixed->base;
        // four arrays, each rounded up to 256 bytes
        budget = (int64_t)((fixed->bytes - counter_bytes - fixed_redo_bytes - 4 * 256) / slot_bytes);
        const int64_t min_band = (int64_t)((rw + 3) / 4) * 16 * spp2;           // one row of 4x4-pixel tiles
        if (min_band > budget)
            return fail(RT_ERR_UNSUPPORTED, "rt_render_rect_noalloc: rectangle too wide for the fixed workspace (" +
                        std::to_string((long long)min_band) + " samples per 4-row band, limit " + std::to_string((long long)budget) + ")");
    } else if (sc->any_spawning_material) {
        // levels below the first can hold many times the band's samples (a mirror hit spawns spr+1 rays per bounce);
        // a band must still be large enough to fill the GPU when there are many samples per pixel
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        // the queues of all levels of such a band came to about 600 bytes per level-0 sample on the largest scene measured
        // (10 M triangles, mixed materials, 3 bounces): a quarter of the free memory at most, within 2^21 .. 2^LOG2
        budget = std::min<int64_t>((int64_t)1 << RT_BAND_BUDGET_LOG2, std::max<int64_t>((int64_t)1 << 21, (int64_t)(free_b / 2400)));
    }
    int band_rows = (int)std::max<int64_t>(1, std::min<int64_t>(rh, budget / ((int64_t)rw * spp2)));
    if (band_rows < rh) band_rows = std::max(4, band_rows & ~3);       // whole 4-row tiles per band
    if (fixed)      // tile padding counts against the fixed block too
        while (band_rows > 4 && (int64_t)((rw + 3) / 4) * ((band_rows + 3) / 4) * 16 * spp2 > budget) band_rows -= 4;

    // replay in two passes (a lane per sample, then a lane per pixel) where a pixel's samples are many and its ray
    // trees can be deep; one fused pass otherwise (it moves fewer bytes)
    const rt::DScene& D = exact ? sc->d : sc->d_fast;
    const bool two_pass = sc->any_spawning_material && spp2 >= RT_TWO_PASS_MIN_SAMPLES;
    int rc = workspace_begin(sc, s);
    if (rc != RT_OK) return rc;
    // the GPU's fixed workspace is shared by every handle on it: behind the previous call that used it
    if (order && order->pending && !stream_is_capturing(s)) HIP_TRY(hipStreamWaitEvent(s, order->done, 0));
    WorkspaceScope scope(sc, s, order);
    float* d_out = rgb_out;
    if (!out_dev) {
        rc = ensure_bytes((void**)&sc->d_stage, &sc->stage_cap, (int64_t)rw * rh * 12);
        if (rc != RT_OK) return rc;
        d_out = sc->d_stage;
    }
    const bool timing = stats != nullptr;
    float ms_extend = 0, ms_shadow = 0, ms_shade = 0, ms_total = 0;
    if (timing) HIP_TRY(hipEventRecord(sc->ev[4], s));
    uint64_t rays_closest = 0, rays_any = 0, rays_idle = 0;   // idle: light rays the reference traces and this path need not
                                                              // (rt_options.skip_unlit; companions whose hit has no record)
    int max_levels = 0;
    // light rays traced: only counted when the caller asked for statistics (the count costs an atomic per wave)
    unsigned long long* d_traced = timing ? (unsigned long long*)(d_counters + 16) : nullptr;
    // a scene none of whose materials spawns rays has one level and its kernels never touch the counters unless
    // statistics are asked for: such a frame is three launches and nothing else
    const bool single_level_async = !sc->any_spawning_material && !timing;
    // rays of the pruning traversal whose answer is a triangle the reference never looks at are walked a second time
    rt::RedoQueue redo{};
    {
        unsigned char* lists = fixed ? fixed->base + counter_bytes : (unsigned char*)sc->d_redo;
        redo.cap = std::min(sc->redo_cap, fixed ? (uint32_t)RT_REDO_CAP_FIXED : (uint32_t)RT_REDO_CAP);
        redo.count_extend = d_counters + RT_W_REDO_EXT; redo.ticket_extend = d_counters + RT_W_TICKET_EXT; redo.total_extend = d_counters + RT_W_TOTAL_EXT;
        redo.count_shadow = d_counters + RT_W_REDO_SHA; redo.ticket_shadow = d_counters + RT_W_TICKET_SHA; redo.total_shadow = d_counters + RT_W_TOTAL_SHA;
        redo.list_extend = (uint32_t*)lists; redo.list_shadow = (unsigned long long*)(lists + (size_t)redo.cap * 4);
    }
    uint64_t redone_extend = 0, redone_shadow = 0;

    // A band whose deeper levels outgrow the item limit or the memory is started again at half the height: nothing of a
    // band reaches the frame before its replay, and the band's statistics are added only once it is through.
    for (int by = 0; by < rh;) {
        const int rows = std::min(band_rows, rh - by);
        bool too_big = false;
        uint64_t b_closest = 0, b_any = 0, b_idle = 0, b_work[4] = {0, 0, 0, 0};
        float b_extend = 0, b_shadow = 0;
        const int64_t n_pix = (int64_t)rows * rw;
        const int64_t n_real = n_pix * spp2;
        int64_t n = (int64_t)((rw + 3) / 4) * ((rows + 3) / 4) * 16 * spp2;   // level-0 slots incl. tile padding
        rt::Level fixed_level{};
        if (fixed) {
            unsigned char* q = fixed->base + counter_bytes + fixed_redo_bytes;
            fixed_level.implicit = 1;
            fixed_level.hit = (float4*)q; q += align_up((size_t)n * 16, 256);
            fixed_level.pos = (float4*)q; q += align_up((size_t)n * 16, 256);
            fixed_level.tri = (int32_t*)q; q += align_up((size_t)n * 4, 256);
            fixed_level.unocc = (uint8_t*)q; q += align_up((size_t)n * std::max(nl, 1), 256);
            if ((size_t)(q - fixed->base) > fixed->bytes) return fail(RT_ERR_UNSUPPORTED, "rt_render_rect_noalloc: band does not fit the fixed workspace");
        } else {
            rc = ensure_level(sc->lv[0], n, nl, true, two_pass);
            if (rc != RT_OK) return rc;
        }
        P.band_y0 = y0 + by; P.band_rows = rows;
        int lvl = 0;
        rt::Level level0 = fixed ? fixed_level : sc->lv[0].L;
        for (;;) {
            rt::Level& L = (lvl == 0) ? level0 : sc->lv[lvl].L;
            if (!single_level_async)
                HIP_TRY(hipMemsetAsync(d_counters, 0, RT_N_COUNTERS * sizeof(uint32_t), s));
            if (timing) HIP_TRY(hipEventRecord(sc->ev[0], s));
            rt::launch_extend(exact, D, P, L, n, d_counters, counting ? (unsigned long long*)(d_counters + 4) : nullptr, vetted ? &redo : nullptr, s);
            if (timing) HIP_TRY(hipEventRecord(sc->ev[1], s));
            rt::launch_shadow(exact, D, P, L, n * nl, d_traced, counting ? (unsigned long long*)(d_counters + 8) : nullptr, vetted ? &redo : nullptr, s);
            if (timing) HIP_TRY(hipEventRecord(sc->ev[2], s));
            if (single_level_async) {
                // the level count is known (one) and the call stays asynchronous
                ++lvl;
                break;
            }
            HIP_TRY(hipMemcpyAsync(sc->h_counters, d_counters, RT_N_COUNTERS * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            if ((rc = check_fault(sc, s)) != RT_OK) return rc;      // synchronises the stream
            HIP_TRY(hipGetLastError());
            if (timing) {
                float a = 0, b = 0;
                HIP_TRY(hipEventElapsedTime(&a, sc->ev[0], sc->ev[1]));
                HIP_TRY(hipEventElapsedTime(&b, sc->ev[1], sc->ev[2]));
                b_extend += a; b_shadow += b;
            }
            b_closest += (uint64_t)(lvl == 0 ? n_real : n);
            if (timing)
                for (int q = 0; q < rt::RT_TRACED_SLOTS; ++q) {
                    unsigned long long v; memcpy(&v, sc->h_counters + 16 + q * rt::RT_TRACED_STRIDE * 2, 8);
                    b_any += v;
                    memcpy(&v, sc->h_counters + 16 + q * rt::RT_TRACED_STRIDE * 2 + 4, 8);
                    b_idle += v;
                }
            for (int q = 0; q < 4; ++q) { unsigned long long v; memcpy(&v, sc->h_counters + 4 + 2 * q, 8); b_work[q] += v; }
            int64_t kids = 0;
            // (signed: the second walk of a suspect node corrects slot 0 by the difference in children, which may be negative)
            for (int q = 0; q < rt::RT_TRACED_SLOTS; ++q) kids += (int32_t)sc->h_counters[rt::RT_KIDS_WORD0 + q * rt::RT_TRACED_STRIDE * 2];
            redone_extend += sc->h_counters[RT_W_TOTAL_EXT]; redone_shadow += sc->h_counters[RT_W_TOTAL_SHA];
            ++lvl;
            if (kids == 0) break;
            if (lvl >= rt::RT_MAX_LEVELS) return fail(RT_ERR_UNSUPPORTED, "ray tree deeper than the level limit");
            if (kids * (int64_t)std::max(nl, 1) >= sc->level_limit) {
                if (rows > 4) { too_big = true; break; }
                return fail(RT_ERR_UNSUPPORTED, "a level of the ray tree exceeds the limit of light-ray items (2^31 unless "
                                                "rt_scene_set_level_limit lowered it) on a band of four rows; render a narrower rectangle");
            }
            rc = ensure_level(sc->lv[lvl], kids, nl, false, two_pass);
            if (rc != RT_OK) {
                if (rows > 4) { (void)hipGetLastError(); too_big = true; break; }    // out of memory: a lower band needs less
                return rc;
            }
            // the new level's rays start at the hit points of this one
            sc->lv[lvl].L.ppos = L.implicit ? L.pos : reinterpret_cast<const float4*>(L.geo);
            sc->lv[lvl].L.pstride = L.implicit ? 1 : 2;
            rt::launch_spawn(D, P, L, sc->lv[lvl].L, n, d_counters + 1, s);
            n = kids;
        }
        if (too_big) {
            band_rows = std::max(4, (rows / 2) & ~3);
            continue;
        }
        rays_closest += b_closest; rays_any += b_any; rays_idle += b_idle;
        ms_extend += b_extend; ms_shadow += b_shadow;
        for (int q = 0; q < 4; ++q) work[q] += b_work[q];
        max_levels = std::max(max_levels, lvl);
        rt::LevelSet ls{};
        ls.lv[0] = level0;
        for (int q = 1; q < lvl; ++q) ls.lv[q] = sc->lv[q].L;
        if (timing) HIP_TRY(hipEventRecord(sc->ev[0], s));
        const int64_t n0 = (int64_t)((rw + 3) / 4) * ((rows + 3) / 4) * 16 * spp2;
        if (two_pass) rt::launch_shade_sum(D, P, ls, n0, n_pix, d_out + (int64_t)by * rw * 3, s);
        else if (!sc->any_spawning_material && spp2 <= 64) rt::launch_replay_flat(D, P, ls, n0 / spp2, d_out + (int64_t)by * rw * 3, s);
        else rt::launch_replay(D, P, ls, n_pix, d_out + (int64_t)by * rw * 3, s);
        if (timing) {
            HIP_TRY(hipEventRecord(sc->ev[1], s));
            HIP_TRY(hipEventSynchronize(sc->ev[1]));
            float a = 0; HIP_TRY(hipEventElapsedTime(&a, sc->ev[0], sc->ev[1]));
            ms_shade += a;
        }
        by += rows;
    }
    HIP_TRY(hipGetLastError());
    if (timing) HIP_TRY(hipEventRecord(sc->ev[5], s));
    if (!out_dev) {
        HIP_TRY(hipMemcpyAsync(rgb_out, d_out, (size_t)rw * rh * 12, hipMemcpyDeviceToHost, s));
        if ((rc = check_fault(sc, s)) != RT_OK) return rc;          // synchronises the stream
    }
    if (timing) {
        HIP_TRY(hipEventSynchronize(sc->ev[5]));
        HIP_TRY(hipEventElapsedTime(&ms_total, sc->ev[4], sc->ev[5]));
        stats->rays_closest = rays_closest; stats->rays_any = rays_any;
        stats->ref_get_hit_calls = 2 * rays_closest + rays_any + rays_idle;
        stats->ms_total = ms_total; stats->ms_extend = ms_extend; stats->ms_shadow = ms_shadow; stats->ms_shade = ms_shade;
        stats->levels = max_levels;
        stats->reserved_[0] = 0;
        stats->reserved_[1] = (int32_t)std::min<uint64_t>(redone_extend, INT32_MAX); stats->reserved_[2] = (int32_t)std::min<uint64_t>(redone_shadow, INT32_MAX);
        stats->nodes_extend = work[0]; stats->tris_extend = work[1]; stats->nodes_shadow = work[2]; stats->tris_shadow = work[3];
    }
    return RT_OK;
}

// A call in the vetted pruning mode that ends because more rays needed the second walk than its queue holds is run
// again in the reference-arithmetic mode: the same frame and ray counts, more slowly.  (A call that returns without
// waiting - device output, no statistics - cannot see the overflow; the next call on the handle reports it.)
static int render_or_fall_back(rt_scene* sc, const rt_options* o, int x0, int y0, int x1, int y1, float* rgb_out, rt_stats* stats, void* stream,
                               const FixedWorkspace* fixed, ScratchOrder* order) {
    g_redo_overflow = false;
    int rc = render_impl(sc, o, x0, y0, x1, y1, rgb_out, stats, stream, fixed, order);
    if (rc != RT_OK && g_redo_overflow && o && o->exact_boxes == 0) {
        rt_options again = *o;
        again.exact_boxes = 1;
        rc = render_impl(sc, &again, x0, y0, x1, y1, rgb_out, stats, stream, fixed, order);
        if (rc == RT_OK && stats) stats->reserved_[0] = 1;       // the call fell back
    }
    return rc;
}

int rt_render_rect(rt_scene* sc, const rt_options* o, int x0, int y0, int x1, int y1, float* rgb_out, rt_stats* stats, void* stream) {
    if (!sc) return fail(RT_ERR_INVALID, "rt_render_rect: null argument");
    DeviceGuard dg(sc->device);
    if (!dg.ok) return fail(RT_ERR_HIP, "rt_render_rect: cannot make the scene's device current");
    RT_API_BEGIN
    return render_or_fall_back(sc, o, x0, y0, x1, y1, rgb_out, stats, stream, nullptr, nullptr);
    RT_API_END("rt_render_rect")
}

int rt_render_rect_noalloc(rt_scene* sc, const rt_options* o, int x0, int y0, int x1, int y1, float* rgb_out, rt_stats* stats, void* stream) {
    if (!sc) return fail(RT_ERR_INVALID, "rt_render_rect_noalloc: null argument");
    DeviceGuard dg(sc->device);
    if (!dg.ok) return fail(RT_ERR_HIP, "rt_render_rect_noalloc: cannot make the scene's device current");
    RT_API_BEGIN
    FixedWorkspace fw{nullptr, RT_SCRATCH_BYTES};
    int rc = scratch_address(sc->device, &fw.base);     // looked up by rt_init; here only on a device rt_init never saw
    if (rc != RT_OK) return rc;
    ScratchOrder* order = &g_scratch_order[sc->device];
    std::lock_guard<std::mutex> use(order->use);        // one call at a time enqueues work on this GPU's workspace
    return render_or_fall_back(sc, o, x0, y0, x1, y1, rgb_out, stats, stream, &fw, order);
    RT_API_END("rt_render_rect_noalloc")
}

int rt_render_frame_multi(rt_scene* const* scenes, int n_scenes, const rt_options* opt, int band_rows, float* rgb_out, rt_stats* stats) {
    if (!scenes || n_scenes <= 0 || n_scenes > RT_MAX_DEVICES || !opt || !rgb_out || band_rows < 1)
        return fail(RT_ERR_INVALID, "rt_render_frame_multi: null argument, no scenes or band_rows < 1");
    for (int i = 0; i < n_scenes; ++i) {
        if (!scenes[i]) return fail(RT_ERR_INVALID, "rt_render_frame_multi: null scene handle");
        for (int j = 0; j < i; ++j)
            if (scenes[i] == scenes[j]) return fail(RT_ERR_INVALID, "rt_render_frame_multi: the same scene handle twice (create one per GPU)");
    }
    if (opt->width <= 0 || opt->height <= 0) return fail(RT_ERR_INVALID, "rt_render_frame_multi: empty image");
    RT_API_BEGIN
    const int W = opt->width, H = opt->height, N = n_scenes;
    std::vector<int> rc((size_t)N, RT_OK);
    std::vector<std::string> msg((size_t)N);
    std::vector<rt_stats> st((size_t)N);
    // one host thread per GPU, like the reference's one thread per tile (src/main.cpp:489-507); joined below
    auto work_body = [&](int i) {
        rt_scene* sc = scenes[i];
        auto bad = [&](int code, const std::string& m) { rc[(size_t)i] = code; msg[(size_t)i] = m; };
        DeviceGuard dg(sc->device);
        if (!dg.ok) return bad(RT_ERR_HIP, "cannot make the scene's device current");
        // local rows of share i: every N-th band of band_rows rows
        int n_local = 0;
        for (int y0 = i * band_rows; y0 < H; y0 += N * band_rows) n_local += std::min(band_rows, H - y0);
        if (n_local == 0) return;
        hipError_t e = hipSuccess;
        {
            std::lock_guard<std::mutex> lock(sc->mu);
            if (!sc->own_stream) e = hipStreamCreateWithFlags(&sc->own_stream, hipStreamNonBlocking);
            if (e == hipSuccess && sc->share_cap < (int64_t)n_local * W * 12) {
                if (sc->d_share) (void)hipFree(sc->d_share);
                sc->d_share = nullptr; sc->share_cap = 0;
                e = hipMalloc((void**)&sc->d_share, (size_t)n_local * W * 12);
                if (e == hipSuccess) sc->share_cap = (int64_t)n_local * W * 12;
            }
        }
        if (e != hipSuccess) return bad(RT_ERR_HIP, hipGetErrorString(e));
        rt_options o = *opt;
        o.row_stride = N; o.row_band = band_rows; o.row_phase = i;
        const int r = render_or_fall_back(sc, &o, 0, 0, W, n_local, sc->d_share, stats ? &st[(size_t)i] : nullptr, sc->own_stream, nullptr, nullptr);
        if (r != RT_OK) return bad(r, g_err);
        // the bands go straight to their rows of the frame (host memory, or any GPU's: peer copies over the fabric)
        int l = 0;
        for (int y0 = i * band_rows; y0 < H && e == hipSuccess; y0 += N * band_rows) {
            const int rows = std::min(band_rows, H - y0);
            e = hipMemcpyAsync(rgb_out + (size_t)y0 * W * 3, sc->d_share + (size_t)l * W * 3, (size_t)rows * W * 12, hipMemcpyDefault, sc->own_stream);
            l += rows;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(sc->own_stream);
        if (e != hipSuccess) return bad(RT_ERR_HIP, hipGetErrorString(e));
    };
    // nothing may leave a worker thread as an exception (that would end the process), and a thread that cannot be
    // started leaves its share to the calling thread
    auto work = [&](int i) {
        try { work_body(i); }
        catch (const std::exception& e_) { rc[(size_t)i] = RT_ERR_INVALID; try { msg[(size_t)i] = e_.what(); } catch (...) {} }
        catch (...) { rc[(size_t)i] = RT_ERR_INVALID; }
    };
    std::vector<std::thread> pool;
    pool.reserve((size_t)N);
    int started = 1;
    try { for (; started < N; ++started) pool.emplace_back(work, started); } catch (...) {}
    work(0);
    for (int i = started; i < N; ++i) work(i);
    for (auto& t : pool) t.join();
    for (int i = 0; i < N; ++i)
        if (rc[(size_t)i] != RT_OK) return fail(rc[(size_t)i], "rt_render_frame_multi, share " + std::to_string(i) + ": " + msg[(size_t)i]);
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        for (const rt_stats& q : st) {
            stats->rays_closest += q.rays_closest; stats->rays_any += q.rays_any; stats->ref_get_hit_calls += q.ref_get_hit_calls;
            stats->nodes_extend += q.nodes_extend; stats->tris_extend += q.tris_extend; stats->nodes_shadow += q.nodes_shadow; stats->tris_shadow += q.tris_shadow;
            stats->ms_total = std::max(stats->ms_total, q.ms_total); stats->ms_extend = std::max(stats->ms_extend, q.ms_extend);
            stats->ms_shadow = std::max(stats->ms_shadow, q.ms_shadow); stats->ms_shade = std::max(stats->ms_shade, q.ms_shade);
            stats->levels = std::max(stats->levels, q.levels);
        }
    }
    return RT_OK;
    RT_API_END("rt_render_frame_multi")
}

int rt_resolve_u8(const float* rgb, int width, int height, uint8_t* out, void* stream) {
    if (!rgb || !out || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "rt_resolve_u8: null argument or empty image");
    RT_API_BEGIN
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)width * height * 3;
    const bool in_dev = is_device_ptr(rgb), out_dev = is_device_ptr(out);
    const float* d_in = rgb; uint8_t* d_out = out;
    Staging* stg = nullptr;
    std::unique_lock<std::mutex> lock;
    if (!in_dev || !out_dev) {
        // host images go through this device's staging buffer, kept between calls; the call holds it until its copies
        // are done
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        if (dev < 0 || dev >= RT_MAX_DEVICES) return fail(RT_ERR_UNSUPPORTED, "device index beyond RT_MAX_DEVICES");
        stg = &g_staging[dev];
        lock = std::unique_lock<std::mutex>(stg->mu);
        if (stg->cap < (size_t)n * 5) {
            if (stg->p) (void)hipFree(stg->p);
            stg->p = nullptr; stg->cap = 0;
            HIP_TRY(hipMalloc(&stg->p, (size_t)n * 5));
            stg->cap = (size_t)n * 5;
        }
        if (!in_dev) { HIP_TRY(hipMemcpyAsync(stg->p, rgb, (size_t)n * 4, hipMemcpyHostToDevice, s)); d_in = (const float*)stg->p; }
        if (!out_dev) d_out = (uint8_t*)stg->p + n * 4;
    }
    rt::launch_resolve_u8(d_in, width, height, d_out, s);
    HIP_TRY(hipGetLastError());
    if (!out_dev) HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)n, hipMemcpyDeviceToHost, s));
    if (stg) HIP_TRY(hipStreamSynchronize(s));
    return RT_OK;
    RT_API_END("rt_resolve_u8")
}

}  // extern "C"
