/*
 * rt_amd.h — C-ABI of the MI355X ray-tracing hot path.
 *
 * The reference (Wei-Parker-Guo/RayTracer) has no FFI layer; its renderer is
 * reached through three C++ seams.  Each entry point below replaces one of
 * them and cites it.  Everything crosses the boundary as plain pointers and
 * sizes; handles are opaque; every call returns an int status (RT_OK == 0)
 * and never throws.  The caller owns all input arrays (they are copied
 * during the call) and every output buffer; the library owns device memory
 * until rt_scene_destroy.
 *
 * Pointer arguments documented as "host or device" are classified with
 * hipPointerGetAttributes: device pointers are used in place, host pointers
 * are staged through a device buffer and copied on the given stream.
 *
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).
 * All work of a call is enqueued on that stream; calls that write to host
 * memory synchronise the stream before returning.
 *
 * Devices: a scene lives on the device that was current when it was created
 * (one handle per GPU; rt_init(device) selects it).  Every call on a scene
 * makes that device current for its duration and restores the caller's
 * device, so one host thread can drive scenes on several GPUs in turn; the
 * stream passed with a scene must belong to the scene's device.
 *
 * Ordering: calls on one scene handle are serialised on the host by a mutex
 * and on the device by an event - a call's stream first waits for the work
 * of the previous call on that handle, whichever stream that ran on.  Calls
 * on different handles are independent (those of rt_render_rect_noalloc share
 * one array per GPU and are ordered by the library, see there).
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    RT_OK = 0,
    RT_ERR_INVALID = 1,     /* bad argument (null, negative size, rect outside the image ...) */
    RT_ERR_HIP = 2,         /* a HIP runtime call failed; see rt_last_error */
    RT_ERR_NO_DEVICE = 3,   /* no gfx950 device visible */
    RT_ERR_UNSUPPORTED = 4  /* e.g. more lights than RT_MAX_LIGHTS */
};

enum { RT_MAT_DEFAULT = 0, RT_MAT_LAMBERT = 1, RT_MAT_PHONG = 2, RT_MAT_REFRAC = 3 };
enum { RT_LIGHT_DIRECTIONAL = 0, RT_LIGHT_POINT = 1 };
enum { RT_JITTER_OFF = 0, RT_JITTER_HASH = 1 };
enum { RT_MAX_LIGHTS = 8 };

/* One mesh = what Mesh::construct_unit_surfaces produces (reference
 * src/geometry.cpp:360-376): triangles in face order, world space, with the
 * three vertex normals.  pos/nrm are [n_tris][3][3] floats (host memory).
 * name must be unique in the scene and at most 31 bytes (src/geometry.h:108). */
typedef struct rt_mesh_desc {
    char name[32];
    int32_t material;       /* index into the materials array */
    int32_t n_tris;
    const float* pos;
    const float* nrm;
} rt_mesh_desc;

/* Material parameters as the reference's Lambert/Phong/Refrac constructors
 * read them (src/materials.cpp:10-21,45-66,90-118).  64 bytes. */
typedef struct rt_material_desc {
    int32_t kind;           /* RT_MAT_* */
    float base[3];
    float ambient[3];
    float spec[3];
    float transp[3];
    float shininess;
    float reflectivity;
    float refractivity;
} rt_material_desc;

/* src/lights.cpp:7-25.  Directional: vec = unit vector toward the light
 * (already negated as src/main.cpp:369 does).  Point: vec = position.  32 bytes. */
typedef struct rt_light_desc {
    int32_t kind;           /* RT_LIGHT_* */
    float vec[3];
    float color[3];
    float pad_;
} rt_light_desc;

/* src/cameras.cpp:3-18: w is Camera::lookat, i.e. the BACKWARD unit vector.  64 bytes. */
typedef struct rt_camera_desc {
    float pos[3];
    float up[3];
    float w[3];
    float side[3];
    float near_clip;
    float far_clip;
    float pad_[2];
} rt_camera_desc;

/* The argument list of RenderThread::operator() (src/render_thread.h:22-25)
 * plus the image size the Rasterizer carries, and the knobs this build adds. */
typedef struct rt_options {
    int32_t width, height;      /* -dispw / -disph */
    float hfov;                 /* -hfov, degrees */
    int32_t spp;                /* -spp n: n x n samples per pixel */
    int32_t spr;                /* -spr n: n jittered + 1 exact ray per split */
    int32_t max_bounce;         /* -bounce */
    int32_t max_refrac_bounce;  /* -rbounce */
    float eps;                  /* -eps */
    float ray_eps;              /* -reps */
    int32_t jitter_mode;        /* RT_JITTER_OFF: every rand() is 100 (no jitter);
                                   RT_JITTER_HASH: counter hash keyed by (seed, pixel, sample, ray path, call site) */
    uint32_t seed;
    int32_t exact_boxes;        /* 0: the pruning traversal (reciprocal multiplies, nearest child first, subtrees beyond the best
                                   hit skipped, a tree of its own); the winner of every ray is checked against the box of the
                                   REFERENCE's leaf that holds it, with the reference's arithmetic, and a ray whose winner the
                                   reference would never have looked at is walked a second time - so the answer is the
                                   reference's own, not merely one that is as near.
                                   1: the reference's slab arithmetic (IEEE divisions) on the reference's tree without pruning -
                                   visits exactly the triangles the reference visits; several times slower.
                                   2: the pruning traversal without that check (1 - 6 % faster; a few pixels per million can
                                   then show a hit the reference loses between two of its boxes) */
    int32_t count_traversal;    /* 1: also count inner nodes fetched and triangles tested per kernel kind into
                                   rt_stats (slower kernels; for traffic accounting, not for timing) */
    int32_t row_stride;         /* 0 or 1: rows are image rows.  N > 1: the rows given to rt_render_rect are LOCAL rows of
                                   one of N interleaved shares of the image: local row l is image row
                                   ((l / row_band) * N + row_phase) * row_band + l % row_band  (multi-GPU row bands) */
    int32_t row_band;           /* band height of the interleave (>= 1 when row_stride > 1) */
    int32_t row_phase;          /* which share, 0 .. row_stride-1 */
    int32_t skip_unlit;         /* 1: in a scene none of whose materials spawns rays, the light rays of a hit are not traced
                                   toward a light that cannot change its colour whatever they find: a default-material hit
                                   (src/materials.cpp:4-8 ignores the shadow record) or a Lambert hit whose shaded term
                                   base * (light * max(0, n.l) + ambient) is zero (src/shaders/basic_shaders.cpp:6-13),
                                   e.g. a surface facing away from the light with no ambient colour.  The reference traces
                                   them (src/render_thread.cpp:146-161) and multiplies the zero by what they return; the
                                   frame is the same bit for bit, rt_stats.rays_any counts only what was traced.
                                   0: trace every light ray the reference traces.  Ignored in scenes that spawn rays
                                   (there a later ray of the pixel can read the record these rays leave behind). */
} rt_options;

typedef struct rt_stats {
    uint64_t rays_closest;      /* nearest-hit queries traced (primary + secondary render rays) */
    uint64_t rays_any;          /* any-hit queries traced (light-visibility rays) */
    uint64_t ref_get_hit_calls; /* get_hit calls the reference would make for the same work: it traces every render
                                   ray twice (src/render_thread.cpp:82-104,269-280), and it sends light rays this
                                   library counts without tracing - those rt_options.skip_unlit leaves out, and
                                   those from a hit whose record the reference never filled (every t it was given
                                   was +inf or NaN, src/render_thread.cpp:13-19), which start at a non-finite point */
    uint64_t nodes_extend;      /* with count_traversal: 64-byte inner nodes fetched by the nearest-hit kernels */
    uint64_t tris_extend;       /*                       48-byte triangles tested by them */
    uint64_t nodes_shadow;      /* the same for the any-hit kernels */
    uint64_t tris_shadow;
    float ms_total;             /* device time of the whole call (HIP events on the call's stream) */
    float ms_extend;            /* nearest-hit kernels */
    float ms_shadow;            /* any-hit kernels */
    float ms_shade;             /* shade / spawn / resolve kernels */
    int32_t levels;             /* bounce levels processed */
    int32_t reserved_[3];       /* [1], [2]: nearest-hit / light rays that were walked a second time because their first answer
                                   was a triangle the reference never looks at (exact_boxes = 0).  [0]: 1 when more rays needed
                                   that second walk than its queue holds (2^20 per launch) and the call was therefore run again
                                   with exact_boxes = 1 - the same frame, more slowly */
} rt_stats;

typedef struct rt_scene rt_scene;

/* Library/device bring-up.  device < 0 keeps the current HIP device. */
int rt_init(int device);

/* Text of the last error on the calling thread. */
const char* rt_last_error(void);

/* Replaces AABBTree::AABBTree(std::vector<Mesh*>&) (src/AABBTree.cpp:3-16) together with
 * BVHNode::BVHNode(std::vector<Surface*>, int, int) (src/geometry.cpp:61-153): builds the same
 * midpoint-partition tree per mesh on the host, flattens it and uploads it to the current device. */
int rt_scene_create(const rt_mesh_desc* meshes, int n_meshes,
                    const rt_material_desc* materials, int n_materials,
                    const rt_light_desc* lights, int n_lights,
                    const rt_camera_desc* camera, rt_scene** out);
/* The same with build options.  RT_BUILD_TREE_ON_DEVICE: the second tree of a scene - the one the pruning traversal
 * (exact_boxes = 0) walks, on the host a surface-area-heuristic build on a few threads - is built on the GPU instead
 * (Morton keys, radix sort, all inner nodes in parallel, boxes bottom-up): a fraction of a second for ten million
 * triangles, at some cost in traversal speed.  The reference's own trees (tie ranks, exact_boxes = 1) are built on the
 * host either way.  Results do not depend on the choice.  rt_scene_create passes 0, or this flag when the environment
 * has RT_AMD_BUILDER=device. */
enum { RT_BUILD_TREE_ON_DEVICE = 1 };
int rt_scene_create_ex(const rt_mesh_desc* meshes, int n_meshes,
                       const rt_material_desc* materials, int n_materials,
                       const rt_light_desc* lights, int n_lights,
                       const rt_camera_desc* camera, uint32_t flags, rt_scene** out);
int rt_scene_destroy(rt_scene* scene);

/* Downloads the tree the pruning traversal of `scene` walks and verifies it on the host like rt_tree_check (meshes: the
 * arrays the scene was created from).  out as for rt_tree_check; out[7] = 1 when the tree was built on the device.
 * The device copy carries each node's child references a second time in the low mantissa bytes of its box planes (the
 * planes moved outwards by less than 512 units in the last place to make room): the check also verifies those bytes and
 * lets a child's box stick out of its parent's by that much. */
int rt_scene_check_tree(rt_scene* scene, const rt_mesh_desc* meshes, int n_meshes, int64_t out[8]);

/* Shape of the flattened tree: counts[0]=triangles, [1]=inner nodes, [2]=leaves, [3]=max depth. */
int rt_scene_info(const rt_scene* scene, int64_t counts[4]);

/* Bounds the memory of rt_render_rect on scenes with reflective or refractive materials.  No counterpart in the
 * reference, whose RayPool grows by 4000 rays at a time without a bound (src/raypool.cpp:3-24).  The renderer works
 * through a rectangle in bands of rows and, per band, through the levels of the ray tree; a band whose level would hold
 * max_items or more light-ray items (rays of the level x lights), or whose queues the GPU cannot allocate, is started
 * again at half its height (down to four rows; then RT_ERR_UNSUPPORTED).  Frames do not depend on the band height.
 * Default and maximum 2^31 (the width of an item index); 0 restores the default. */
int rt_scene_set_level_limit(rt_scene* scene, int64_t max_items);

/* Host-only check of the two tree builders (needs no device): builds, from the same input as rt_scene_create, the
 * flattened form of the reference's trees (BVHNode::BVHNode, src/geometry.cpp:61-153; which = 0) or the
 * surface-area-heuristic tree the pruning traversal walks (which = 1), verifies it (every triangle in exactly one
 * leaf, every child box contains its subtree's triangles, leaf sizes, child references in range, tie ranks a
 * permutation, depth within the traversal stack) and reports
 * out[0]=triangles, [1]=node records, [2]=leaves, [3]=max depth, [4]=largest leaf, [5]=triangles reached from the
 * entry node, [6]=nodes reached, [7]=0.  Returns RT_ERR_INVALID with a message when a check fails. */
int rt_tree_check(const rt_mesh_desc* meshes, int n_meshes, int which, int64_t out[8]);

/* Replaces bool get_hit(const AABBTree&, const Ray*, float t0, float t1, hitrec&)
 * (src/render_thread.h:17, src/render_thread.cpp:5-23) in batch form.
 * org/dir: [n][3] floats, host or device.  t_out[n] = hit distance or +inf; tri_out[n] = global
 * triangle id (mesh order, then face order) or -1; nrm_out[n][3] = interpolated unit normal
 * (may be NULL).  any_hit != 0 answers only "is there a hit" (what src/render_thread.cpp:158 uses):
 * t_out is then 0 (occluded) or +inf and tri_out is 0/-1.  exact_boxes as in rt_options (0, 1 or 2). */
int rt_trace(rt_scene* scene, const float* org, const float* dir, float t0, float t1, int64_t n,
             int any_hit, int exact_boxes, float* t_out, int32_t* tri_out, float* nrm_out, void* stream);

/* rt_trace plus, per ray, the number of inner nodes fetched and triangles tested (for traffic accounting).
 * No counterpart in the reference. */
int rt_trace_counted(rt_scene* scene, const float* org, const float* dir, float t0, float t1, int64_t n,
                     int any_hit, int exact_boxes, float* t_out, int32_t* tri_out,
                     uint32_t* nodes_out, uint32_t* tris_out, void* stream);

/* Replaces RenderThread::operator()(Rasterizer*, AABBTree&, Camera*, std::vector<Light*>, int startX,
 * int startY, int endX, int endY, ...) (src/render_thread.h:22-25, src/render_thread.cpp:25-488).
 * Renders pixels x in [x0,x1), y in [y0,y1) (y = 0 is the bottom row, as in the reference; with
 * rt_options.row_stride > 1 y counts the local rows of one interleaved share, see there) and writes
 * float RGB = sum(colours)/sum(weights) per pixel (src/rasterizer.cpp:88-103; no draw -> NaN) into
 * rgb_out[(y - y0) * (x1 - x0) + (x - x0)][3], host or device.  stats may be NULL. */
int rt_render_rect(rt_scene* scene, const rt_options* opt, int x0, int y0, int x1, int y1,
                   float* rgb_out, rt_stats* stats, void* stream);

/* rt_render_rect for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same results bit for bit, with
 * these limits: the scene has no reflective or refractive material (their ray queues grow with the frame), rgb_out is
 * device memory, and one 4-row band of the rectangle fits the fixed workspace: ceil(width/4) * 16 * spp^2 samples at
 * (36 + lights) bytes each within 64 MiB; larger ones are refused with RT_ERR_UNSUPPORTED.  Taller rectangles are
 * rendered band by band.  The call makes no hipMalloc/hipFree: its queues live in one 64 MiB array that every GPU the
 * library runs on holds until the process ends.  All calls on one GPU share that array, whatever their scene handle;
 * the library orders them itself: a call holds a per-GPU lock while it enqueues its work, and its stream first waits for
 * the end of the previous such call on that GPU (an event), so handles, streams and host threads may be mixed freely.
 * On a stream that is being captured into a HIP graph (stats must be NULL: statistics need a host synchronisation) the
 * call records its kernels and touches none of the library's events; replays of such a graph are then ordered against
 * other users of the scene handle and of the GPU's array by the caller, like any other use of the caller's buffers. */
int rt_render_rect_noalloc(rt_scene* scene, const rt_options* opt, int x0, int y0, int x1, int y1,
                           float* rgb_out, rt_stats* stats, void* stream);

/* Replaces renderFrame's fan-out of one frame over worker threads and its join (src/main.cpp:474-507) by a fan-out over
 * GPUs driven from ONE host process: scenes[i] is the same scene created on GPU i (rt_init(i); rt_scene_create(...)).
 * The frame is cut into bands of band_rows rows; band k is rendered by scenes[k % n_scenes] (sky and geometry spread
 * evenly), each share by one rt_render_rect-equivalent call on a host thread and stream of its own, and every band is
 * copied straight to its rows of rgb_out ([height][width][3], row 0 = bottom; host memory or memory of any of the GPUs -
 * peer copies).  Returns when the whole frame is in place.  The result is bit-identical to one rt_render_rect of the
 * whole frame.  stats (may be NULL): ray counts summed, times the maximum over the GPUs.  (Processes that hold one
 * GPU each use rt_options.row_stride with rt_render_rect instead and gather the rows themselves.) */
int rt_render_frame_multi(rt_scene* const* scenes, int n_scenes, const rt_options* opt, int band_rows,
                          float* rgb_out, rt_stats* stats);

/* Replaces the 8-bit conversion of save_jpg (src/main.cpp:157-167):
 * out[(H-1-y)*W + x][c] = max(0, min(255, (int)floor(rgb * 256.0))), NaN -> 0.  rgb is [H][W][3] with row 0
 * at the bottom; out is top-down.  Both host or device. */
int rt_resolve_u8(const float* rgb, int width, int height, uint8_t* out, void* stream);

/* Replaces the file output of save_jpg (src/main.cpp:169-176, which hands the bytes to the third-party TooJpeg::writeJpeg,
 * src/toojpeg.h:35, with isRGB = true, quality 90, downsample = false, comment "Render Result"): a baseline JFIF file,
 * YCbCr 4:4:4, standard quantisation tables scaled by `quality` (1..100), standard Huffman tables, `comment` as a COM
 * segment (may be NULL).  rgb is the [height][width][3] 8-bit image rt_resolve_u8 produces (HOST memory, row 0 = top).
 * An independent encoder: decoders give the same picture within quantisation error, the byte stream is not TooJpeg's. */
int rt_save_jpeg(const char* path, const uint8_t* rgb, int width, int height, int quality, const char* comment);
/* The same into memory: *out_size = bytes needed / written; out may be NULL to ask for the size only. */
int rt_encode_jpeg(const uint8_t* rgb, int width, int height, int quality, const char* comment, uint8_t* out,
                   int64_t out_cap, int64_t* out_size);

/* ---- host side: scene files and option files (no device work, usable without a GPU) ----------------------------------
 *
 * Replaces load_scene (src/main.cpp:256-395; node transforms retrieve_node_gtrans :240-253; material by name
 * :331-349 with the constructors' defaults src/materials.cpp:10-118; lights src/lights.cpp:7-25 with the direction
 * negated as :369 does; camera basis src/cameras.cpp:3-18), which imports through Assimp.  This library reads the
 * reference's scene format - binary FBX - itself (node records, deflated arrays, connections, the FBX node transform,
 * polygons fan-triangulated in face order, smooth normals where a mesh brings none) and hands back exactly the arrays
 * rt_scene_create takes: world-space triangles with corner normals per mesh, materials, lights, one camera.  It also
 * reads the flat ".rtscene" form (8 bytes "RTSCN01\0", uint32 counts {meshes, materials, lights}, then the C-ABI structs
 * themselves: camera, materials, lights, and per mesh a 32-byte name, int32 material, int32 n_tris, positions, normals).
 * The arrays belong to the handle and stay valid until rt_host_scene_free.  Errors: rt_host_last_error(). */
typedef struct rt_host_scene rt_host_scene;
int rt_host_scene_load(const char* path, rt_host_scene** out);
int rt_host_scene_desc(const rt_host_scene* scene, const rt_mesh_desc** meshes, int* n_meshes,
                       const rt_material_desc** materials, int* n_materials,
                       const rt_light_desc** lights, int* n_lights, const rt_camera_desc** camera);
int rt_host_scene_free(rt_host_scene* scene);
const char* rt_host_last_error(void);

/* The reference's option defaults (src/main.cpp:74-75,103-115: 960x540, -hfov 54.43, -spp 2, -spr 4, -bounce 3,
 * -rbounce 3, -eps 0.0001, -reps 0.1) with hash jitter, seed 1, and the pruning traversal. */
int rt_options_default(rt_options* opt);
/* Replaces read_cmd_tokens (src/main.cpp:646-683) for one line "-key value" of an option file; keys -dispw -disph -hfov
 * -spp -bounce -spr -eps -reps -rbounce, and -mpbs (the display block size, a host matter: *display_block, may be NULL).
 * Returns 1 if the line set something, 0 if it was ignored (unknown key, no value, blank - the reference crashes on a
 * blank line, src/main.cpp:654). */
int rt_options_parse_line(rt_options* opt, const char* line, int* display_block);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
