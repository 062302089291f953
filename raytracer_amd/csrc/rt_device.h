// Device-side building blocks: float3 helpers that keep the reference's operation order, the slab and
// triangle tests, the stack traversal, ray generation and the counter-hash random numbers.
// Compiled with -ffp-contract=off: every * and + below is one IEEE float32 operation unless fmaf is
// written out, because results are compared bit-for-bit with a CPU that has no fused multiply-add.
#pragma once
#include "rt_types.h"

#include <hip/hip_runtime.h>
#define RT_HD __device__ __forceinline__
#define RT_HDM __device__ __forceinline__   // the same for member functions
#define RT_HOST_HD __host__ __device__ inline   // small helpers the host side calls too
#define RT_CLZ(x) __clz(x)
#define RT_F2I(x) __float_as_int(x)
#define RT_I2F(x) __int_as_float(x)
#define RT_RCP(x) __builtin_amdgcn_rcpf(x)
#define RT_LDS_PTR __attribute__((address_space(3)))
// Threads per workgroup of the kernels (and stride of the LDS stack columns): ONE wave.  A workgroup's LDS is held until
// its slowest wave is done, and waves of light rays end at very different times; with four waves per workgroup the
// finished ones' slots stayed empty behind a straggler (64 against 256: bunny frame 1.073 -> 1.047 ms, 10 M-triangle
// field 1.36 -> 1.27 s; 512 was 12 % slower than 256).
#define RT_BLOCK 64
// keeps all four lanes of a loaded float4 "used" so that the load stays one 16-byte instruction
#define RT_KEEP4(q) asm volatile("" ::"v"((q).x), "v"((q).y), "v"((q).z), "v"((q).w))

// Fetch of one 64-byte node + the node step.  When every active lane of the wave is at the same node (rays that are
// still together near the top of the tree) the node is read ONCE through the scalar cache into SGPRs instead of by
// four 16-byte vector loads per lane: the vector memory pipeline is the busiest unit of the traversal kernels and a
// wave-wide vector load costs it the same whether the lanes ask for one node or for 64.
typedef float rt_f16 __attribute__((ext_vector_type(16)));
#define RT_NODE_STEP(EXACT, OCT, ORDER, nodes, cur, st, o, d, rd, lo, lim)                                          \
    do {                                                                                                     \
        const int ucur_ = __builtin_amdgcn_readfirstlane(cur);                                               \
        if (__builtin_amdgcn_ballot_w64((cur) != ucur_) == 0) {                                              \
            rt_f16 s_;                                                                                       \
            const void* up_ = (const void*)((nodes) + ucur_);                                                \
            asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s_) : "s"(up_) : "memory"); \
            cur = node_step<EXACT, OCT, ORDER>(s_[0], s_[1], s_[2], s_[3], s_[4], s_[5], s_[6], s_[7], s_[8], s_[9], s_[10], \
                                   s_[11], RT_F2I(s_[12]), RT_F2I(s_[13]), RT_F2I(s_[14]), RT_F2I(s_[15]), st, o, d, \
                                   rd, lo, lim);                                                             \
        } else {                                                                                             \
            const float4* np_ = reinterpret_cast<const float4*>((nodes) + (cur));                            \
            const float4 q0 = np_[0], q1 = np_[1], q2 = np_[2];                                              \
            RT_KEEP4(q0); RT_KEEP4(q1); RT_KEEP4(q2);                                                        \
            if (EXACT) {                                                                                     \
                const float4 q3 = np_[3];                                                                    \
                RT_KEEP4(q3);                                                                                \
                cur = node_step<EXACT, OCT, ORDER>(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, \
                                       RT_F2I(q3.x), RT_F2I(q3.y), RT_F2I(q3.z), RT_F2I(q3.w), st, o, d, rd, lo, lim); \
            } else {                                                                                         \
                /* the pruning traversal's tree carries its two references in the low bytes of eight box planes */ \
                const int32_t rl_ = packed_ref(q0.x, q0.y, q0.z, q0.w), rr_ = packed_ref(q1.z, q1.w, q2.x, q2.y); \
                cur = node_step<EXACT, OCT, ORDER>(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, \
                                       rl_, rr_, rl_, rr_, st, o, d, rd, lo, lim);                            \
            }                                                                                                \
        }                                                                                                    \
    } while (0)

// The pruning traversal's nodes carry their two child references a second time, byte by byte in the lowest mantissa
// byte of eight of the twelve box planes (rt_api.hip packs them when a scene is created, moving each plane outwards by
// less than 512 units in the last place so that the box only grows): a lane that fetches its own node then needs 48
// bytes, three loads, instead of 64 - the vector-memory path is the busiest unit of these kernels.  Byte k of the
// reference is the low byte of plane k of its box (lo.x, lo.y, lo.z, hi.x).
__device__ __forceinline__ int32_t packed_ref(float a, float b, float c, float d) {
    const uint32_t ab = __builtin_amdgcn_perm((uint32_t)__float_as_int(b), (uint32_t)__float_as_int(a), 0x0c0c0400u);
    const uint32_t cd = __builtin_amdgcn_perm((uint32_t)__float_as_int(d), (uint32_t)__float_as_int(c), 0x0c0c0400u);
    return (int32_t)__builtin_amdgcn_perm(cd, ab, 0x05040100u);
}

#ifndef RT_ANY_HIT_ORDER
// Order in which an any-hit ray descends where it hits both children.  Only a ray that IS occluded can gain from an order
// (a free ray walks every box it meets, whatever the order), and it gains when it reaches a triangle in its way sooner.
// Light rays start on a surface, inside the boxes of that surface's own neighbourhood, whose triangles lie beside or
// behind them; what blocks them stands farther along the ray.  4, the child whose box the ray LEAVES later, is what
// sends them there: frames 7 % (bunny), 17 % (Phong bunny), 15 % (mirror room), 13 % (glass balls) faster than 0, left
// child first, with the same answers.  1 (the child entered first, the order of nearest-hit rays) is slower than no order at
// all, 2 (entered later) and 3 (longest stay) lie between; a leaf before an inner node, ties of 4 broken by entry, and the
// mean of entry and exit were tried on top of 4 and change nothing.
#define RT_ANY_HIT_ORDER 4
#endif

namespace rt {

struct DScene {
    const Node64* nodes;
    const Tri48* tris;
    const TriInfo* info;       // per triangle, same order: caller's id and mesh index
    const float* nrm;          // [n_tris][9], traversal order
    const Material* mats;      // per material
    const int32_t* mesh_mat;   // mesh index -> material index
    uint32_t* fault;           // one word, set to 1 by a lane whose traversal stack overflowed
    int32_t shallow;           // the tree is no deeper than RT_LDS_SLOTS: its traversal stack never leaves LDS (see StackT)
    const float4* leaf_box;    // pruning traversal's tree only: the boxes of the REFERENCE's leaves, two float4 per leaf
                               // (lo.xyz hi.x | hi.yz - -); a triangle's leaf number sits in the first spare word of its
                               // record there (see ref_leaf_box_hit)
    Light lights[8];
    int32_t n_lights;
    Camera cam;
};

// Division of a number below 2^31 by a divisor fixed for the whole call: q = (mulhi(n, m) + n) >> s with
// s = ceil(log2 d), m = floor(2^32 (2^s - d) / d) + 1 (round-up method; the sum cannot overflow because n < 2^31).
// A runtime integer division costs the GPU some 25 instructions (a 64-bit one over 100); every lane of every kernel
// needs a few of them to find its pixel, sample, light and ray from its index.
struct FastDiv { uint32_t m, s; };
RT_HOST_HD FastDiv make_fastdiv(uint32_t d) {
    FastDiv f; uint32_t l = 0;
    while (l < 31 && (1u << l) < d) ++l;
    f.s = l; f.m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
    return f;
}
RT_HOST_HD uint32_t fdiv(uint32_t n, FastDiv f) { return ((uint32_t)(((uint64_t)n * f.m) >> 32) + n) >> f.s; }

struct RenderParams {
    int32_t W, H;              // full image
    int32_t x0, y0, rw, rh;    // rect origin and size
    int32_t spp, spr, D, Dr;
    float eps, ray_eps;
    int32_t jitter; uint32_t seed;
    float fl, fr, fb, ft, fd;  // frustum l, r, b, t and near distance d (src/render_thread.cpp:34-40)
    int32_t row_stride, row_band, row_phase;   // interleaved row shares (row_stride <= 1: rows are image rows)
    int32_t band_y0, band_rows;                // the band of rows being rendered: first (local) row and height
    int32_t can_spawn;                         // some material of the scene spawns rays (Phong / refractive)
    int32_t skip_unlit;                        // rt_options.skip_unlit, and the scene spawns no rays
    FastDiv div_spp, div_spp2, div_tiles_x, div_lights, div_rays;   // by spp, spp^2, 4x4 tiles per row, lights, spr + 1
};

// fills the divisors once the other fields are set
RT_HOST_HD void set_divisors(RenderParams& P, int n_lights) {
    P.div_spp = make_fastdiv((uint32_t)P.spp); P.div_spp2 = make_fastdiv((uint32_t)(P.spp * P.spp));
    P.div_tiles_x = make_fastdiv((uint32_t)((P.rw + 3) >> 2));
    P.div_lights = make_fastdiv((uint32_t)(n_lights > 0 ? n_lights : 1)); P.div_rays = make_fastdiv((uint32_t)(P.spr + 1));
}

// image row of local row l (see rt_options.row_stride)
RT_HD int image_row(const RenderParams& P, int l) {
    if (P.row_stride <= 1) return l;
    return ((l / P.row_band) * P.row_stride + P.row_phase) * P.row_band + l % P.row_band;
}

struct V3 { float x, y, z; };

RT_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD V3 v3p(const float* p) { return v3(p[0], p[1], p[2]); }
RT_HD V3 vadd(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD V3 vsub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD V3 vscale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RT_HD V3 vmul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
// src/linmath.h:30-37: p = 0; p += b[i]*a[i] for i = 0,1,2
RT_HD float vdot(V3 a, V3 b) { float p = 0.f; p += b.x * a.x; p += b.y * a.y; p += b.z * a.z; return p; }
// src/linmath.h:42-46: k = 1/sqrt(v.v); r = v*k
RT_HD V3 vnorm(V3 a) { float k = 1.f / sqrtf(vdot(a, a)); return vscale(a, k); }

// std::max / std::min as the reference calls them (src/shaders/basic_shaders.cpp:8,26, src/fast_math.cpp:25-27)
RT_HD float max_std(float a, float b) { return (a < b) ? b : a; }
RT_HD float min_std(float a, float b) { return (b < a) ? b : a; }

// src/fast_math.cpp:14-22, unrolled from the top bit down: fp(y) = fp(y/2)^2 or x*fp(y/2)*fp(y/2)
RT_HD float fast_pow_i(float x, int y) {
    if (y <= 0) return 1.f;
    float r = 1.f;
    for (int b = 31 - RT_CLZ(y); b >= 0; --b) r = ((y >> b) & 1) ? x * r * r : r * r;
    return r;
}

// ---- counter-hash random numbers ---------------------------------------------------------------------
RT_HD uint32_t h_mix(uint32_t h, uint32_t v) {
    h ^= v * 0x9E3779B1u;
    h = ((h << 13) | (h >> 19)) * 5u + 0xE6546B64u;
    return h;
}
RT_HD uint32_t h_fin(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
enum { SITE_PRIMARY = 1, SITE_SH_LIGHT = 2, SITE_SH_SPLIT = 3, SITE_SHADE_LIGHT = 4, SITE_REFLECT_SPLIT = 5 };
// What rand() returns at call sites (a, b0), (a, b0 + 1), (a, b0 + 2) of one node - the three components of a jitter
// vector.  jitter == 0: every call returns 100, which makes every jitter term of the reference zero; otherwise
// h_fin(h_mix(h_mix(h_mix(h_mix(seed, node), site), a), b)) >> 1 per call, the part the three share computed once.
RT_HD void rnd3(int jitter, uint32_t seed, uint32_t node, uint32_t site, uint32_t a, uint32_t b0, int& r0, int& r1, int& r2) {
    if (jitter == 0) { r0 = r1 = r2 = 100; return; }
    const uint32_t pre = h_mix(h_mix(h_mix(seed, node), site), a);
    r0 = (int)(h_fin(h_mix(pre, b0)) >> 1);
    r1 = (int)(h_fin(h_mix(pre, b0 + 1u)) >> 1);
    r2 = (int)(h_fin(h_mix(pre, b0 + 2u)) >> 1);
}
RT_HD uint32_t node_root(uint32_t seed, uint32_t pixel, uint32_t sample) {
    return h_fin(h_mix(h_mix(seed ^ 0x5bd1e995u, pixel), sample));
}
RT_HD uint32_t node_child(uint32_t parent, uint32_t rank) { return h_fin(h_mix(h_mix(parent, 0xC0FFEEu), rank)); }

// x / 100.0f for an integer-valued 0 <= x < 100000, bit-equal to the IEEE division the reference performs after
// rand() % 100 or % 200: the product with the correctly rounded reciprocal is within one unit in the last place, the
// fused residual is exact, and one correction step then rounds correctly (checked exhaustively over that range).
// Three instructions instead of the eleven of a general float division.
RT_HD float div100(float x) {
    const float r = 0.01f;
    const float q = x * r;
    return fmaf(fmaf(-100.0f, q, x), r, q);
}

// ---- primary rays: src/render_thread.cpp:57-79 ----------------------------------------------------------
// i = column, j = row from the bottom, s = x*spp + y
RT_HD V3 primary_dir(const DScene& sc, const RenderParams& P, int i, int j, int s, uint32_t node) {
    const int x = (int)fdiv((uint32_t)s, P.div_spp), y = s - x * P.spp;
    const float unit_len = 1.0f / P.spp;
    int r0, r1, r2;
    rnd3(P.jitter, P.seed, node, SITE_PRIMARY, 0, 0, r0, r1, r2);    // the third value is not used (and not computed)
    const float jx = div100((float)(r0 % 100));
    const float jy = div100((float)(r1 % 100));
    const float u = P.fl + (P.fr - P.fl) * (i + unit_len * x + jx * unit_len) / P.W;
    const float v = P.fb + (P.ft - P.fb) * (j + unit_len * y + jy * unit_len) / P.H;
    V3 du = vscale(v3p(sc.cam.side), u), dv = vscale(v3p(sc.cam.up), v), dw = vscale(v3p(sc.cam.w), -P.fd);
    return vnorm(vadd(vadd(du, dv), dw));
}

// ---- lights: src/lights.cpp:13-54 -------------------------------------------------------------------------
RT_HD V3 light_dir(const Light& L, V3 p, const RenderParams& P, uint32_t node, uint32_t site, uint32_t li) {
    if (L.kind == 0) return v3p(L.vec);
    V3 r = vnorm(vsub(v3p(L.vec), p));
    V3 add;
    int r0, r1, r2;
    rnd3(P.jitter, P.seed, node, site, li, 0, r0, r1, r2);
    add.x = 0.01f * (div100((float)(r0 % 200)) - 1.0f);
    add.y = 0.01f * (div100((float)(r1 % 200)) - 1.0f);
    add.z = 0.01f * (div100((float)(r2 % 200)) - 1.0f);
    return vnorm(vadd(r, add));
}
// src/ray.cpp:47-61: one jittered copy of an exact direction
RT_HD V3 split_dir(V3 exact, float eps, const RenderParams& P, uint32_t node, uint32_t site, uint32_t a, uint32_t k) {
    V3 add;
    int r0, r1, r2;
    rnd3(P.jitter, P.seed, node, site, a, k * 3, r0, r1, r2);
    add.x = eps * (div100((float)(r0 % 200)) - 1.0f);
    add.y = eps * (div100((float)(r1 % 200)) - 1.0f);
    add.z = eps * (div100((float)(r2 % 200)) - 1.0f);
    return vnorm(vadd(exact, add));
}

// ---- traversal ---------------------------------------------------------------------------------------------
struct HitRec { float t; int32_t tri; int32_t rank; float beta, gamma;      // tri = position in the triangle array, -1 = none
                int32_t leaf; };                                             // pruning traversal: the reference's leaf of the hit triangle

// Slab test of one box with the reference's arithmetic (src/geometry.cpp:25-37: six true divisions, fmin/fmax that
// drop NaNs).
RT_HD bool slab_exact(const float* b, V3 o, V3 d, float& tmin) {
    const float t1 = (b[0] - o.x) / d.x, t2 = (b[3] - o.x) / d.x;
    const float t3 = (b[1] - o.y) / d.y, t4 = (b[4] - o.y) / d.y;
    const float t5 = (b[2] - o.z) / d.z, t6 = (b[5] - o.z) / d.z;
    tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    const float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    return !(tmax < 0.f || tmin > tmax);
}

// Slab test of the pruning traversal: is some t of [lo, lim] inside the box?  A multiplication by the precomputed
// reciprocal replaces each division; the subtraction is the reference's, (bound - e) * (1/d), so d == 0 gives the
// same +-inf / NaN pattern as its division (a fused b*rd - e*rd would turn inf - inf into NaN and lose the slab).
// OCT < 8: bit a of OCT is the sign bit of direction component a for every lane of the wave, which tells at compile
// time which of the two planes of an axis is entered first: half of the min/max operations disappear (they run at
// half the rate of a multiply or add on this hardware).  OCT == 8: lanes differ, min/max per axis.
// The exit distance is widened by a few units in the last place (the reciprocal form is that far off), which keeps
// a ray that passes exactly through a box corner or edge (a shared vertex of a mesh) from slipping between boxes.
// tmin returns the entry distance, not below lo.
// What the pruning traversal keeps per ray for its box tests: 1/d and, for the octant forms, e/d widened downwards and
// upwards (see slab_fast).
struct SlabRay { V3 rd, en, ef; };

template <int OCT>
RT_HD bool slab_fast(const float* b, V3 o, const SlabRay& r, float lo, float lim, float& tmin, float& tfar) {
    float tn, tf;
    if (OCT < 8) {
        // One fused multiply-add per plane, bound / d - e / d, instead of a subtraction and a multiplication.  The
        // rounded e / d is off by up to 2^-24 of itself, which need not be small against the distance: en and ef are
        // e / d moved by 2^-22 of itself towards a later entry plane value being smaller and an exit value being larger,
        // so the interval only ever grows.  (No zero or tiny direction component takes this form: traverse sends those
        // rays to OCT == 8.)
        const float nx = fmaf(b[(OCT & 1) ? 3 : 0], r.rd.x, -r.en.x), fx = fmaf(b[(OCT & 1) ? 0 : 3], r.rd.x, -r.ef.x);
        const float ny = fmaf(b[(OCT & 2) ? 4 : 1], r.rd.y, -r.en.y), fy = fmaf(b[(OCT & 2) ? 1 : 4], r.rd.y, -r.ef.y);
        const float nz = fmaf(b[(OCT & 4) ? 5 : 2], r.rd.z, -r.en.z), fz = fmaf(b[(OCT & 4) ? 2 : 5], r.rd.z, -r.ef.z);
        tn = fmaxf(fmaxf(nx, ny), nz);
        tf = fminf(fminf(fx, fy), fz);
    } else {
        const float t1 = (b[0] - o.x) * r.rd.x, t2 = (b[3] - o.x) * r.rd.x;
        const float t3 = (b[1] - o.y) * r.rd.y, t4 = (b[4] - o.y) * r.rd.y;
        const float t5 = (b[2] - o.z) * r.rd.z, t6 = (b[5] - o.z) * r.rd.z;
        tn = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
        tf = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    }
    tmin = fmaxf(tn, lo);
    tfar = fminf(tf * 1.000001f, lim);
    return !(tmin > tfar);
}

// 1/d (hardware reciprocal: within one unit in the last place, well inside what slab_fast allows for) and the widened
// e/d.  usable_oct is false when the fused form must not be used: a component of d that is zero or so small that e/d
// is not a moderate finite number.
RT_HD SlabRay make_slab_ray(V3 o, V3 d, bool& usable_oct) {
    SlabRay r;
    r.rd = v3(RT_RCP(d.x), RT_RCP(d.y), RT_RCP(d.z));
    const V3 ed = vmul(o, r.rd);
    const float w = 2.384185791015625e-07f;   // 2^-22
    const V3 m = v3(fabsf(ed.x) * w, fabsf(ed.y) * w, fabsf(ed.z) * w);
    r.en = vadd(ed, m); r.ef = vsub(ed, m);
    const float big = fmaxf(fmaxf(fabsf(ed.x), fabsf(ed.y)), fabsf(ed.z));
    const float small = fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
    usable_oct = (big < 1e30f) && (small > 1e-18f);
    return r;
}

// Bounds the pruning traversal hands to slab_fast.
// lim, "this subtree starts beyond the best hit so far": the triangle distances it is compared with are rounded
// results of a different formula than the box distances, so equal or almost equal hits in different leaves (a ray
// through a shared vertex or edge) must all be looked at: the reference picks among them by (t, tie rank), which
// needs every candidate.  The Cramer distance of a sliver met at a shallow angle is good to a few 1e-5 of itself, not to
// the 1e-7 of a box distance (a ray aimed at a vertex of the dome's rim from 16 units away: 15.95199 and 15.95211 from
// the two slivers that share the vertex, 15.95229 to the vertex itself), so the slack is a thousandth: with 1e-5 the
// nearer of those two was pruned whenever the farther was met first.  Costs nothing measurable (subtrees that begin
// within 0.1 % beyond the best hit are few).  lo, "this box ends before the ray's interval starts": a ray that leaves a
// flat surface skips that surface's own boxes at once instead of descending to the leaves around its origin.
#ifndef RT_PRUNE_SLACK
#define RT_PRUNE_SLACK 1.001f
#endif
RT_HD float prune_limit(float best_t) { return best_t * RT_PRUNE_SLACK; }
RT_HD float entry_limit(float t0) { return fmaxf(t0 * 0.999f, 0.f); }

// Work counters of the counting kernels: steps of this lane, or (profiling build, -DRT_COUNT_WAVE_STEPS) steps of the
// wave - only the first active lane counts, so the sum over lanes * 64 against the per-lane sum is the lane
// utilisation of the loop.
#if defined(RT_COUNT_WAVE_STEPS)
#define RT_COUNT_STEP(c) do { if ((int)threadIdx.x == __builtin_amdgcn_readfirstlane((int)threadIdx.x)) ++(c); } while (0)
#else
#define RT_COUNT_STEP(c) ++(c)
#endif

#ifndef RT_STACK
#define RT_STACK 96       // deepest tree the traversal accepts (checked when the scene is created)
#endif
#define RT_LDS_SLOTS 16   // stack entries per lane kept in LDS; deeper entries go to a per-lane array

// Per-lane stack of postponed child references.  The first RT_LDS_SLOTS entries live in LDS, laid out
// [slot][lane of the workgroup] so that a wave's access to one slot is conflict-free; the rest spill to a private
// array.  Near-first descent rarely holds more than a handful of entries.  The struct holds only scalars (the spill
// array is a separate object) so that it stays in registers.
// DEEP == false: for trees no deeper than RT_LDS_SLOTS (a stack never holds more entries than its tree has levels; the
// host checks the depth when it picks the kernel).  Push and pop are then one LDS access each - without the tiers, their
// nested branches around every push and pop, and the private array: 1 - 3 % of every frame whose scene has such a tree
// (the surface-area-heuristic trees of all the smaller scenes: bunny 15 levels, mirror room 16, glass balls 14).
template <bool DEEP>
struct StackT {
    RT_LDS_PTR int32_t* lds; // this lane's column: slot s is lds[s * RT_BLOCK]
    int32_t* spill;          // RT_STACK - RT_LDS_SLOTS entries
    uint32_t* fault;         // where an overflow is reported (the scene's fault word; the same for every lane)
    int sp;
    RT_HDM void push(int32_t v) {
        if (!DEEP) { lds[sp * RT_BLOCK] = v; ++sp; return; }
        if (sp < RT_LDS_SLOTS) lds[sp * RT_BLOCK] = v;
        else if (sp < RT_STACK) spill[sp - RT_LDS_SLOTS] = v;
        else { *fault = 1u; return; }   // the subtree is lost and sp stays put, so every pop stays inside the arrays; the
                                        // host checks tree depth against RT_STACK before any launch, so this only fires
                                        // if that check and the tree disagree, and the host then reports it
        ++sp;
    }
    RT_HDM int32_t pop() {
        --sp;
        if (!DEEP) return lds[sp * RT_BLOCK];
        return (sp < RT_LDS_SLOTS) ? lds[sp * RT_BLOCK] : spill[sp - RT_LDS_SLOTS];
    }
};
typedef StackT<true> StackLds;

// Would the reference have looked at this triangle at all?  It tests a triangle iff the box of the leaf that holds it in
// ITS tree passes ITS slab test (all boxes above contain that box, and the test is monotone in the box, so they pass
// whenever it does).  The pruning traversal walks another tree with widened boxes and so meets triangles the reference
// never tests; a few of those pass the Cramer test by rounding (a ray grazing a corner within units in the last place).
// Checking a candidate against its reference leaf's box with the reference's arithmetic removes them: what the pruning
// traversal reports is then what the reference reports, not merely something nearer or equal.
RT_HD bool ref_leaf_box_hit(const DScene& sc, int32_t leaf, V3 o, V3 d) {
    const float4 b0 = sc.leaf_box[2 * (int64_t)leaf], b1 = sc.leaf_box[2 * (int64_t)leaf + 1];
    const float b[6] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y};
    float tmin;
    return slab_exact(b, o, d, tmin);
}

// The same question for the winner of a walk, asked cheaply first: with 1/d at hand the six plane distances are six
// multiplications, each within 4e-7 of itself of the reference's rounded quotient.  The reference's test asks whether the
// latest entry is at most the earliest exit; an entry and an exit of ONE axis are ordered by construction, so what decides
// are the six pairs of different axes - if each holds with 1e-6 of the magnitudes to spare (and the earliest exit is
// clearly positive), the exact arithmetic agrees and its six divisions are not needed.  Anything else - a ray along a
// box face, a zero direction component (infinities and NaNs fail every comparison here) - takes the exact test.
RT_HD bool ref_leaf_box_hit_quick(const DScene& sc, int32_t leaf, V3 o, V3 d, V3 rd) {
    const float4 b0 = sc.leaf_box[2 * (int64_t)leaf], b1 = sc.leaf_box[2 * (int64_t)leaf + 1];
    const float x1 = (b0.x - o.x) * rd.x, x2 = (b0.w - o.x) * rd.x;
    const float y1 = (b0.y - o.y) * rd.y, y2 = (b1.x - o.y) * rd.y;
    const float z1 = (b0.z - o.z) * rd.z, z2 = (b1.y - o.z) * rd.z;
    const float nx = fminf(x1, x2), fx = fmaxf(x1, x2), ny = fminf(y1, y2), fy = fmaxf(y1, y2), nz = fminf(z1, z2), fz = fmaxf(z1, z2);
    const float k = 1e-6f;
    const bool clear = (int)(fmaf(k, fabsf(nx) + fabsf(fy), nx) <= fy) & (int)(fmaf(k, fabsf(nx) + fabsf(fz), nx) <= fz) &
                       (int)(fmaf(k, fabsf(ny) + fabsf(fx), ny) <= fx) & (int)(fmaf(k, fabsf(ny) + fabsf(fz), ny) <= fz) &
                       (int)(fmaf(k, fabsf(nz) + fabsf(fx), nz) <= fx) & (int)(fmaf(k, fabsf(nz) + fabsf(fy), nz) <= fy) &
                       (int)(fminf(fminf(fx, fy), fz) > 1e-30f);
    if (clear) return true;
    const float b[6] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y};
    float tmin;
    return slab_exact(b, o, d, tmin);
}

// One triangle against the ray: the reference's Cramer's-rule test (src/geometry.cpp:268-295, same names, same
// operation order).  Its three true divisions are the expensive part, so each quotient is first estimated with
// the hardware reciprocal (relative error of a few units in the last place): a triangle whose estimate is outside
// the accepted range by far more than that error is rejected at once, everything else takes the exact division.
// Returns true when the reference's test passes; t, beta, gamma are then the reference's values bit for bit.
RT_HD bool tri_test(const float4 p0, const float4 p1, const float4 p2, V3 o, V3 d, float t0, float tlim,
                    float& t_out, float& beta_out, float& gamma_out) {
    const float a_ = p1.x, b_ = p1.y, c_ = p1.z;
    const float d_ = p2.x, e_ = p2.y, f_ = p2.z;
    const float g_ = d.x, h_ = d.y, i_ = d.z;
    const float j_ = p0.x - o.x, k_ = p0.y - o.y, l_ = p0.z - o.z;
    const float ei_hf = e_ * i_ - h_ * f_;
    const float gf_di = g_ * f_ - d_ * i_;
    const float dh_eg = d_ * h_ - e_ * g_;
    const float ak_jb = a_ * k_ - j_ * b_;
    const float jc_al = j_ * c_ - a_ * l_;
    const float bl_kc = b_ * l_ - k_ * c_;
    const float m = a_ * ei_hf + b_ * gf_di + c_ * dh_eg;
    const float rm = RT_RCP(m);
    const float tn = f_ * ak_jb + e_ * jc_al + d_ * bl_kc;
    const float gn = i_ * ak_jb + h_ * jc_al + g_ * bl_kc;
    const float bn = j_ * ei_hf + k_ * gf_di + l_ * dh_eg;
    // estimates; a NaN estimate fails every comparison below and so falls through to the exact path.  Only a reciprocal
    // that is a normal number estimates anything: the hardware reciprocal of a subnormal determinant is +-inf and that
    // of one beyond 8.5e37 is flushed to zero, while the exact quotients below can still be finite and inside the ranges.
    // (A select, not a branch around the comparisons: 0.3 % against 2.5 % of the light-ray kernel.)
    const float rme = __builtin_amdgcn_classf(rm, 0x108) ? rm : __builtin_nanf("");      // -normal | +normal
    const float te = -(tn * rme), ge = gn * rme, be = bn * rme;
    const float slack = 1e-4f;
    if (te < t0 - slack * fabsf(t0) - 1e-30f || te > tlim + slack * fabsf(tlim)) return false;
    if (ge < -slack || ge > 1.f + slack) return false;
    if (be < -slack || be + ge > 1.f + 2.f * slack) return false;
    const float t = tn / -m;
    if (t < t0 || t > tlim) return false;
    const float gamma = gn / m;
    if (gamma < 0.f || gamma > 1.f) return false;
    const float beta = bn / m;
    if (beta < 0.f || beta > (1.f - gamma)) return false;
    t_out = t; beta_out = beta; gamma_out = gamma;
    return true;
}

// One inner node: tests both children and returns the next reference to visit (the far child is pushed when both are
// hit; RT_NONE when the stack is empty).  The sixteen node words are passed as scalars so that the same code serves
// words held in vector registers and words held in scalar registers (a node the whole wave fetched once).
template <bool EXACT, int OCT, int ORDER, class ST>
RT_HD int32_t node_step(float b0, float b1, float b2, float b3, float b4, float b5, float b6, float b7, float b8,
                        float b9, float b10, float b11, int32_t r0, int32_t r1, int32_t r2, int32_t r3, ST& st,
                        V3 o, V3 d, const SlabRay& rd, float lo, float lim) {
    const float lb[6] = {b0, b1, b2, b3, b4, b5};
    const float rb[6] = {b6, b7, b8, b9, b10, b11};
    // the reference's own leaves when its exact box tests are asked for, the refined ones otherwise
    const int32_t lref = EXACT ? r0 : r2, rref = EXACT ? r1 : r3;
    float tl, tr, fl = 0.f, fr = 0.f;
    bool hl, hr;
    if (EXACT) { hl = slab_exact(lb, o, d, tl); hr = slab_exact(rb, o, d, tr); }
    else { hl = slab_fast<OCT>(lb, o, rd, lo, lim, tl, fl); hr = slab_fast<OCT>(rb, o, rd, lo, lim, tr, fr); }
    hr = hr && (rref != RT_NONE);
    if (hl && hr) {
        // ORDER 1: the child entered first; 2: the child entered last; 3: the child the ray stays in longest; 4: the child left last
        const bool swap = !EXACT && ((ORDER == 1 && tr < tl) || (ORDER == 2 && tr > tl) || (ORDER == 3 && (fr - tr) > (fl - tl)) || (ORDER == 4 && fr > fl));
        st.push(swap ? lref : rref);
        return swap ? rref : lref;
    }
    if (hl) return lref;
    if (hr) return rref;
    return st.sp ? st.pop() : RT_NONE;
}

// Nearest hit (ANY == false) or any hit (ANY == true) with t in [t0, t1].
// Returns true on a hit.  Nearest: best = the candidate minimal in (t, tie rank) order, which is the reference's
// choice among equal t (ranks: see bvh_build.cpp); when false, best.tri is -1, or -2 if some triangle passed the test
// without a finite t (the reference then reports a hit whose record nobody filled).  EXACT visits every leaf whose box passes the reference's test, like the
// reference's exhaustive recursion (src/geometry.cpp:155-187); otherwise subtrees whose entry distance exceeds the
// best t so far are skipped and the nearer child is visited first.
// Shape: an inner loop that only descends through inner nodes and an inner loop that only tests a leaf's triangles,
// so that the lanes of a wave run the same kind of work together.
template <bool ANY, bool EXACT, int OCT, bool VET, class ST>
RT_HD bool traverse_oct(const DScene& sc, ST& st, V3 o, V3 d, const SlabRay& rd, float t0, float t1, HitRec& best,
                        uint32_t* n_nodes, uint32_t* n_tris) {
    // the empty record loses no tie: a triangle that passes the test with t = +inf (a ray parallel to it through one
    // of its edges: 0/0 barycentrics pass every comparison) is not a hit - the reference's get_hit keeps a record only
    // if its t is BELOW the initial INFINITY (src/render_thread.cpp:8,15)
    const float lo = entry_limit(t0);
    best.t = INFINITY; best.tri = -1; best.rank = INT32_MIN; best.beta = 0.f; best.gamma = 0.f; best.leaf = 0;
    st.sp = 0;
    int32_t cur = 0;
    bool found = false;
    for (;;) {
        // ---- descend until a leaf is reached (cur < 0) or nothing is left (cur == RT_NONE)
        while (cur >= 0) {
            if (n_nodes) RT_COUNT_STEP(*n_nodes);
            const float lim = EXACT ? INFINITY : (ANY ? t1 : prune_limit(best.t));
            RT_NODE_STEP(EXACT, OCT, (ANY ? RT_ANY_HIT_ORDER : 1), sc.nodes, cur, st, o, d, rd, lo, lim);
        }
        if (cur == RT_NONE) break;
        // ---- one leaf: a run of consecutive triangles
        {
            int32_t k = ~cur;
            const float4* tp = reinterpret_cast<const float4*>(sc.tris + k);
            int32_t run = 1;
            for (int32_t r = 0; r < run; ++r, ++k, tp += 3) {
                const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
                RT_KEEP4(p0); RT_KEEP4(p1); RT_KEEP4(p2);
                if (r == 0) run = RT_F2I(EXACT ? p0.w : p1.w);
                if (n_tris) RT_COUNT_STEP(*n_tris);
                float t, beta, gamma;
                if (!tri_test(p0, p1, p2, o, d, t0, ANY ? t1 : fminf(t1, best.t), t, beta, gamma)) continue;
                // VET (the rare second walk of a ray, kernels of their own): a candidate of the pruning traversal counts only
                // if the reference would have looked at it
                if (VET && !ref_leaf_box_hit(sc, RT_F2I(p0.w), o, d)) continue;
                // any-hit with the reference's arithmetic on the reference's tree: whatever passes the test occludes, a
                // NaN t included (0/0: a zero-area triangle, a needle whose determinant cancels to zero on this ray) -
                // the reference's boolean (src/render_thread.cpp:158).  On the pruning traversal's own tree a NaN t is
                // no occluder: which rays reach such a triangle depends on the boxes around it.
                if (ANY) { if (EXACT || t == t) { best.tri = k; best.leaf = RT_F2I(p0.w); return true; } }
                else if (t < best.t || (t == best.t && RT_F2I(p2.w) < best.rank)) {
                    best.t = t; best.tri = k; best.rank = RT_F2I(p2.w); best.beta = beta; best.gamma = gamma;
                    if (!EXACT) best.leaf = RT_F2I(p0.w);
                    found = true;
                }
                // passed the test with a t that is +inf or NaN and nothing recorded so far: the reference's get_hit
                // would answer "hit" and leave its record unset (src/render_thread.cpp:13-19); noted as tri = -2
                else if (!(t < INFINITY) && best.tri == -1) best.tri = -2;
            }
        }
        if (st.sp == 0) break;
        cur = st.pop();
    }
    return found;
}

// Picks the traversal specialised for the wave's direction octant (see slab_fast).
// VET == false (the kernels every ray goes through): *suspect is set when the winner (nearest hit) or the occluder (any
// hit) turns out to be a triangle the reference never looks at - met through the widened boxes of the pruning tree and
// accepted by rounding; the caller then queues the ray for the kernels with VET == true, which walk it again vetting
// every candidate on the way and so find the reference's own answer.  (About one ray in a hundred thousand; vetting
// inside the loop of every ray cost 17 % and a function call from it as much, in registers.)
template <bool ANY, bool EXACT, class ST, bool VET = false>
RT_HD bool traverse(const DScene& sc, ST& st, V3 o, V3 d, float t0, float t1, HitRec& best,
                    uint32_t* n_nodes = nullptr, uint32_t* n_tris = nullptr, bool* suspect = nullptr) {
    SlabRay rd; rd.rd = rd.en = rd.ef = v3(0, 0, 0);
    if (EXACT) return traverse_oct<ANY, true, 8, false>(sc, st, o, d, rd, t0, t1, best, n_nodes, n_tris);
    bool usable;
    rd = make_slab_ray(o, d, usable);
    int oct = (int)(((uint32_t)RT_F2I(d.x) >> 31) | (((uint32_t)RT_F2I(d.y) >> 31) << 1) | (((uint32_t)RT_F2I(d.z) >> 31) << 2));
    // a zero component makes (bound - e) * (1/d) a NaN where the ray's origin lies exactly in a box plane; the
    // reference's fmin/fmax then answer by their NaN-dropping order, which only the min/max form reproduces
    if (!usable) oct = 8;
    const int uoct = __builtin_amdgcn_readfirstlane(oct);
    oct = (__builtin_amdgcn_ballot_w64(oct != uoct) == 0) ? uoct : 8;
    if (VET) return traverse_oct<ANY, false, 8, true>(sc, st, o, d, rd, t0, t1, best, n_nodes, n_tris);
    bool found;
    switch (oct) {
        case 0: found = traverse_oct<ANY, false, 0, false>(sc, st, o, d, rd, t0, t1, best, n_nodes, n_tris); break;
        case 1: found = traverse_oct<ANY, false, 1, false>(sc, st, o, d, rd, t0, t1, best, n_nodes, n_tris); break;
        case 2: found = traverse_oct<ANY, false, 2, false>(sc, st, o, d, rd, t0, t1, best, n_nodes, n_tris); break;
        case 3: found = traverse_oct<ANY, false, 3, false>(sc, st, o, d, rd, t0, t1, best, n_nodes, n_tris); break;
        case 4: found = traverse_oct<ANY, false, 4, false>(sc, st, o, d, rd, t0, t1, best, n_nodes, n_tris); break;
        case 5: found = traverse_oct<ANY, false, 5, false>(sc, st, o, d, rd, t0, t1, best, n_nodes, n_tris); break;
        case 6: found = traverse_oct<ANY, false, 6, false>(sc, st, o, d, rd, t0, t1, best, n_nodes, n_tris); break;
        case 7: found = traverse_oct<ANY, false, 7, false>(sc, st, o, d, rd, t0, t1, best, n_nodes, n_tris); break;
        default: found = traverse_oct<ANY, false, 8, false>(sc, st, o, d, rd, t0, t1, best, n_nodes, n_tris); break;
    }
    if (suspect && found && !ref_leaf_box_hit_quick(sc, best.leaf, o, d, rd.rd)) *suspect = true;
    return found;
}

// src/geometry.cpp:297-306: interpolated unit normal of the hit
RT_HD V3 hit_normal(const DScene& sc, const HitRec& h) {
    const float* n = sc.nrm + (size_t)h.tri * 9;
    const float alpha = 1.f - h.gamma - h.beta;
    V3 pn = vscale(v3p(n), alpha), bf = vscale(v3p(n + 3), h.beta), gf = vscale(v3p(n + 6), h.gamma);
    return vnorm(vadd(vadd(pn, bf), gf));
}

}  // namespace rt
