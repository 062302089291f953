// rt_oracle.cpp — TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path.
//
// CPU restatement of the reference ray-tracing hot path (Wei-Parker-Guo/RayTracer, src/), written
// from the algorithm, float32 throughout, same operation order, no FMA contraction (build with
// -ffp-contract=off).  Every function cites the reference lines it follows.
//
// WHAT PINS IT: the reference's sources cannot be built or run here, and it ships no tests or golden vectors for
// this path.  What it does ship are two of its own 1920x1080 result images and figures recorded from its runs; this
// restatement is checked against those (tests/test_reference_images.py, tests/test_oracle.py: silhouette, horizon,
// shadow side and hit counts) and against analytic known answers.  That pins geometry, camera and light conventions
// and the ray bookkeeping; the last bits of a colour are NOT pinned by anything the reference holds ("parity
// unpinned" at the bit level).
//
// Defined where the reference is undefined (uninitialised reads):
//   * reflect children inherit the parent's weight and refraci (src/render_thread.cpp:240-262 never set
//     them; inheriting refraci also keeps the ray tree finite: only a ray with refraci == 1 resets depth, :322-336);
//   * refract/reflect children of a refractive hit get total_previous_contrib = 0
//     (src/render_thread.cpp:319-337);
//   * a TriangleSet's box is the union of its triangles' boxes (src/geometry.cpp:413-417 starts the
//     union from an uninitialised box); a depth-capped node gets that box too (src/geometry.cpp:65-69
//     leaves it unset);
//   * the per-pixel shadow record starts as 1.0 per light (src/render_thread.cpp:115 starts empty);
//   * a get_hit that returns true without having filled the record - every t a mesh reported was +inf (ray parallel
//     to a triangle through one of its edges) or NaN (zero-area triangle, ray with a NaN direction) - leaves the caller
//     with a hitrec nobody wrote (src/render_thread.cpp:121): here its mesh matches none, so a render ray draws
//     nothing, and its shadow twin's light rays start at a non-finite point.  PixelStats::unset counts these.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../include/rt_amd.h"

namespace {

typedef float vec3[3];
struct Box { vec3 lo, hi; };

// ---- src/linmath.h:10-65, src/fast_math.cpp:14-84 ------------------------------------------------
inline void v_copy(vec3 r, const vec3 a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
inline void v_add(vec3 r, const vec3 a, const vec3 b) { for (int i = 0; i < 3; ++i) r[i] = a[i] + b[i]; }
inline void v_sub(vec3 r, const vec3 a, const vec3 b) { for (int i = 0; i < 3; ++i) r[i] = a[i] - b[i]; }
inline void v_scale(vec3 r, const vec3 a, float s) { for (int i = 0; i < 3; ++i) r[i] = a[i] * s; }
inline float v_dot(const vec3 a, const vec3 b) {
    float p = 0.f;
    for (int i = 0; i < 3; ++i) p += b[i] * a[i];
    return p;
}
inline void v_norm(vec3 r, const vec3 a) {            // linmath.h:42-46: k = 1/sqrt(a.a), then scale
    float k = 1.f / sqrtf(v_dot(a, a));
    v_scale(r, a, k);
}
inline void v_mul(vec3 r, const vec3 a, const vec3 b) { for (int i = 0; i < 3; ++i) r[i] = a[i] * b[i]; }
inline float sqr(float x) { return x * x; }
float fast_pow(float x, int y) {                       // fast_math.cpp:14-22
    if (y == 0) return 1;
    float temp = fast_pow(x, y / 2);
    if (y % 2 == 0) return temp * temp;
    return x * temp * temp;
}

// ---- counter-hash random numbers (this build's replacement for the unseeded global rand()) -------
inline uint32_t h_mix(uint32_t h, uint32_t v) {
    h ^= v * 0x9E3779B1u;
    h = ((h << 13) | (h >> 19)) * 5u + 0xE6546B64u;
    return h;
}
inline uint32_t h_fin(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
enum { SITE_PRIMARY = 1, SITE_SH_LIGHT = 2, SITE_SH_SPLIT = 3, SITE_SHADE_LIGHT = 4, SITE_REFLECT_SPLIT = 5 };
struct Rng {
    int mode; uint32_t seed;
    // value of one rand() call: 100 with jitter off (100 % 100 == 0, 100 % 200 / 100.f - 1 == 0)
    int draw(uint32_t node, uint32_t site, uint32_t a, uint32_t b) const {
        if (mode == RT_JITTER_OFF) return 100;
        return (int)(h_fin(h_mix(h_mix(h_mix(h_mix(seed, node), site), a), b)) >> 1);
    }
};
inline uint32_t node_root(uint32_t seed, uint32_t pixel, uint32_t sample) {
    return h_fin(h_mix(h_mix(seed ^ 0x5bd1e995u, pixel), sample));
}
inline uint32_t node_child(uint32_t parent, uint32_t rank) { return h_fin(h_mix(h_mix(parent, 0xC0FFEEu), rank)); }

// ---- geometry -------------------------------------------------------------------------------------
struct RayG { vec3 e, d; };
struct Hit { float t; vec3 norm; int mesh; int tri; };

struct Counters { uint64_t box_tests = 0, tri_tests = 0; };
thread_local Counters* g_cnt = nullptr;

inline void box_union(const Box& a, const Box& b, Box& out) {          // geometry.cpp:3-12
    Box r = a;
    for (int i = 0; i < 3; ++i) { r.lo[i] = fminf(r.lo[i], b.lo[i]); r.hi[i] = fmaxf(r.hi[i], b.hi[i]); }
    out = r;
}

bool box_hit(const Box& bb, const RayG& r) {                             // geometry.cpp:25-37
    if (g_cnt) g_cnt->box_tests++;
    float t1 = (bb.lo[0] - r.e[0]) / r.d[0];
    float t2 = (bb.hi[0] - r.e[0]) / r.d[0];
    float t3 = (bb.lo[1] - r.e[1]) / r.d[1];
    float t4 = (bb.hi[1] - r.e[1]) / r.d[1];
    float t5 = (bb.lo[2] - r.e[2]) / r.d[2];
    float t6 = (bb.hi[2] - r.e[2]) / r.d[2];
    float tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    return !(tmax < 0 || tmin > tmax);
}

struct Surf {
    virtual ~Surf() {}
    virtual bool hit(const RayG& r, float t0, float t1, Hit& rec) const = 0;
    virtual void bbox(Box& b) const = 0;
};

struct Tri : Surf {
    vec3 a, b, c, an, bn, cn;
    int mesh, id;
    bool hit(const RayG& r, float t0, float t1, Hit& rec) const override {   // geometry.cpp:266-312
        if (g_cnt) g_cnt->tri_tests++;
        float a_ = a[0] - b[0], b_ = a[1] - b[1], c_ = a[2] - b[2];
        float d_ = a[0] - c[0], e_ = a[1] - c[1], f_ = a[2] - c[2];
        float g_ = r.d[0], h_ = r.d[1], i_ = r.d[2];
        float j_ = a[0] - r.e[0], k_ = a[1] - r.e[1], l_ = a[2] - r.e[2];
        float ei_hf = e_ * i_ - h_ * f_;
        float gf_di = g_ * f_ - d_ * i_;
        float dh_eg = d_ * h_ - e_ * g_;
        float ak_jb = a_ * k_ - j_ * b_;
        float jc_al = j_ * c_ - a_ * l_;
        float bl_kc = b_ * l_ - k_ * c_;
        float m = a_ * ei_hf + b_ * gf_di + c_ * dh_eg;
        float t = (f_ * ak_jb + e_ * jc_al + d_ * bl_kc) / -m;
        if (t < t0 || t > t1) return false;
        float gamma = (i_ * ak_jb + h_ * jc_al + g_ * bl_kc) / m;
        if (gamma < 0 || gamma > 1) return false;
        float beta = (j_ * ei_hf + k_ * gf_di + l_ * dh_eg) / m;
        if (beta < 0 || beta > (1 - gamma)) return false;
        float alpha = 1 - gamma - beta;
        vec3 pn, bf, gf;
        v_scale(pn, an, alpha); v_scale(bf, bn, beta); v_scale(gf, cn, gamma);
        v_add(pn, pn, bf); v_add(pn, pn, gf);
        v_norm(pn, pn);
        rec.t = t; v_copy(rec.norm, pn); rec.mesh = mesh; rec.tri = id;
        return true;
    }
    void bbox(Box& o) const override {                                         // geometry.cpp:314-321
        for (int i = 0; i < 3; ++i) {
            o.lo[i] = fminf(a[i], fminf(b[i], c[i]));
            o.hi[i] = fmaxf(a[i], fmaxf(b[i], c[i]));
        }
    }
};

struct TriSet : Surf {                                                         // geometry.cpp:411-440
    std::vector<Surf*> tris; Box box;
    explicit TriSet(const std::vector<Surf*>& l) : tris(l) {
        tris[0]->bbox(box);
        for (Surf* s : tris) { Box b; s->bbox(b); box_union(box, b, box); }
    }
    bool hit(const RayG& r, float t0, float t1, Hit& rec) const override {
        float oldt = INFINITY; bool result = false;
        for (Surf* s : tris) {
            Hit n;
            if (s->hit(r, t0, t1, n) && n.t < oldt) { result = true; rec = n; oldt = rec.t; }
        }
        return result;
    }
    void bbox(Box& b) const override { b = box; }
};

struct Node : Surf {                                                           // geometry.cpp:61-191
    Surf* left = nullptr; Surf* right = nullptr; Box box; bool owns_l = false, owns_r = false;
    Node(std::vector<Surf*> s, int axis, int depth) {
        const int N = (int)s.size();
        if (depth == 0) { left = new TriSet(s); owns_l = true; left->bbox(box); return; }
        if (N == 1) { left = s[0]; s[0]->bbox(box); return; }
        if (N == 2) {
            left = s[0]; right = s[1];
            Box lb, rb; s[0]->bbox(lb); s[1]->bbox(rb); box_union(lb, rb, box);
            return;
        }
        Box sb; s[0]->bbox(sb);
        for (Surf* x : s) { Box b; x->bbox(b); box_union(sb, b, sb); }
        const float m = sb.lo[axis] + (sb.hi[axis] - sb.lo[axis]) / 2;          // midpoint on the INCOMING axis
        int div_dif = std::numeric_limits<int>::max(), best_axis = 0;
        for (int a = 0; a < 3; ++a) {                                            // same m reused for every tried axis
            int nl = 0, nr = 0;
            for (Surf* x : s) {
                Box b; x->bbox(b);
                float bm = b.lo[axis] + (b.hi[axis] - b.lo[axis]) / 2;
                if (bm < m) ++nl; else ++nr;
            }
            int d = std::abs(nl - nr);
            if (d < div_dif) { div_dif = d; best_axis = axis; }
            axis = (axis + 1) % 3;
        }
        axis = best_axis;
        std::vector<Surf*> ll, rl;
        for (Surf* x : s) {
            Box b; x->bbox(b);
            float bm = b.lo[axis] + (b.hi[axis] - b.lo[axis]) / 2;
            if (bm < m) ll.push_back(x); else rl.push_back(x);
        }
        if (ll.empty()) { right = new TriSet(rl); owns_r = true; right->bbox(box); return; }
        if (rl.empty()) { left = new TriSet(ll); owns_l = true; left->bbox(box); return; }
        left = new Node(ll, (axis + 1) % 3, depth - 1); owns_l = true;
        right = new Node(rl, (axis + 1) % 3, depth - 1); owns_r = true;
        box_union(static_cast<Node*>(left)->box, static_cast<Node*>(right)->box, box);
    }
    ~Node() override { if (owns_l) delete left; if (owns_r) delete right; }
    bool hit(const RayG& r, float t0, float t1, Hit& rec) const override {       // geometry.cpp:155-187
        if (!box_hit(box, r)) return false;
        Hit lr, rr;
        bool lh = left && left->hit(r, t0, t1, lr);
        bool rh = right && right->hit(r, t0, t1, rr);
        if (lh && rh) { rec = (lr.t < rr.t) ? lr : rr; return true; }            // tie -> right
        if (lh) { rec = lr; return true; }
        if (rh) { rec = rr; return true; }
        return false;
    }
    void bbox(Box& b) const override { b = box; }
    void shape(int depth, int64_t c[4]) const {
        const Surf* ch[2] = {left, right}; const bool own[2] = {owns_l, owns_r};
        bool inner = false;
        for (int k = 0; k < 2; ++k)
            if (ch[k] && own[k]) if (const Node* n = dynamic_cast<const Node*>(ch[k])) { inner = true; n->shape(depth + 1, c); }
        if (inner) c[1]++; else c[2]++;
        if (depth > c[3]) c[3] = depth;
    }
};

struct MeshO {                                                                  // geometry.cpp:323-382
    char name[32]; int material; std::vector<Tri> tris; Box aabb; Node* root = nullptr;
    bool hit(const RayG& r, float t0, float t1, Hit& rec) const {
        if (!box_hit(aabb, r)) return false;
        return root->hit(r, t0, t1, rec);
    }
};

}  // namespace

struct orc_scene {
    std::vector<MeshO*> meshes;
    std::vector<rt_material_desc> mats;
    std::vector<rt_light_desc> lights;
    rt_camera_desc cam;
    ~orc_scene() { for (MeshO* m : meshes) { delete m->root; delete m; } }
};

namespace {

// src/render_thread.cpp:5-23: linear over all meshes, first mesh wins ties, true if ANY mesh reported a hit
bool get_hit(const orc_scene& sc, const RayG& ray, float t0, float t1, Hit& rec) {
    bool hit = false;
    rec.t = INFINITY; rec.tri = -1; rec.mesh = -1;
    for (MeshO* m : sc.meshes) {
        Hit n;
        if (m->hit(ray, t0, t1, n)) {
            hit = true;
            if (n.t < rec.t) rec = n;
        }
    }
    return hit;
}

// ---- lights: src/lights.cpp:13-54 --------------------------------------------------------------------
void light_dir(const rt_light_desc& L, const vec3 p, vec3 r, const Rng& rng, uint32_t node, uint32_t site, uint32_t li) {
    if (L.kind == RT_LIGHT_DIRECTIONAL) { v_copy(r, L.vec); return; }
    v_sub(r, L.vec, p);
    v_norm(r, r);
    vec3 add;
    for (int c = 0; c < 3; ++c) add[c] = 0.01f * (rng.draw(node, site, li, c) % 200 / 100.0f - 1.0f);
    v_add(r, r, add);
    v_norm(r, r);
}

// ---- shaders: src/shaders/basic_shaders.cpp:6-30 ------------------------------------------------------
void lambert_shade(const vec3 ca, const vec3 cr, const vec3 cl, const vec3 n, const vec3 l, vec3 c) {
    float r = std::max(0.0f, v_dot(n, l));
    v_scale(c, cl, r);
    v_add(c, c, ca);
    v_mul(c, cr, c);
}
void phong_shade(const vec3 cl, const vec3 cp, const vec3 l, const vec3 e, const vec3 n, int p, vec3 c) {
    vec3 r;
    v_scale(r, n, 2.0f * v_dot(l, n));
    v_sub(r, r, l);
    v_norm(r, r);
    vec3 cph;
    v_scale(cph, cp, fast_pow(std::max(0.0f, v_dot(e, r)), p));
    v_mul(cph, cl, cph);
    v_add(c, c, cph);
    for (int i = 0; i < 3; ++i) c[i] = std::min(1.0f, c[i]);
}

// src/materials.cpp:4-8,28-43,68-88,120-140
void apply_shade(const orc_scene& sc, const rt_material_desc& M, vec3 r, const vec3 p, const vec3 eye, const vec3 n,
                 const std::vector<float>& sh, const Rng& rng, uint32_t node) {
    if (M.kind == RT_MAT_DEFAULT) { r[0] = 1.f; r[1] = 1.f; r[2] = 0.f; return; }
    r[0] = r[1] = r[2] = 0.f;
    vec3 view = {0, 0, 0};
    if (M.kind != RT_MAT_LAMBERT) { v_sub(view, eye, p); v_norm(view, view); }
    for (size_t i = 0; i < sc.lights.size(); ++i) {
        vec3 l, c_add = {0, 0, 0};
        light_dir(sc.lights[i], p, l, rng, node, SITE_SHADE_LIGHT, (uint32_t)i);
        lambert_shade(M.ambient, M.base, sc.lights[i].color, n, l, c_add);
        if (M.kind != RT_MAT_LAMBERT) phong_shade(sc.lights[i].color, M.spec, l, view, n, (int)M.shininess, c_add);
        v_scale(c_add, c_add, sh[i]);
        v_add(r, r, c_add);
    }
}

// ---- rays: src/ray.h:16-56, src/ray.cpp:32-96 ------------------------------------------------------------
enum { RAY_SHADOW = 0, RAY_REFLECT = 1, RAY_REFRACTIVE = 2 };
struct Ray {
    int type; RayG g; int depth;
    vec3 c_cache; float contrib, tpc, refraci, weight;
    uint32_t node;        // path key of the render ray this ray is, or accompanies
};
inline void get_point(const RayG& r, float t, vec3 p) { v_scale(p, r.d, t); v_add(p, r.e, p); }
inline void reflect(const RayG& in, const vec3 n, float t, RayG& out) {          // ray.cpp:37-45, linmath.h:59-65
    float p = 2.f * v_dot(in.d, n);
    for (int i = 0; i < 3; ++i) out.d[i] = in.d[i] - p * n[i];
    get_point(in, t, out.e);
}
inline void split_dir(const RayG& exact, float eps, const Rng& rng, uint32_t node, uint32_t site, uint32_t a, uint32_t k, RayG& out) {
    vec3 add;                                                                      // ray.cpp:47-61
    for (int c = 0; c < 3; ++c) add[c] = eps * (rng.draw(node, site, a, k * 3 + c) % 200 / 100.0f - 1.0f);
    v_add(out.d, exact.d, add);
    v_norm(out.d, out.d);
    v_copy(out.e, exact.e);
}
float refract(const RayG& in, const vec3 norm, float n, float nt, float t, RayG& out) {   // ray.cpp:63-96
    get_point(in, t, out.e);
    vec3 term1, term2, tvec;
    v_scale(term1, norm, -1 * v_dot(in.d, norm));
    v_add(term1, in.d, term1);
    v_scale(term1, term1, n / nt);
    v_scale(term2, norm, sqrtf(1 - sqr(n / nt) * (1 - sqr(v_dot(in.d, norm)))));
    v_sub(tvec, term1, term2);
    float c;
    if (v_dot(in.d, norm) < 0) c = -(v_dot(in.d, norm));
    else c = v_dot(tvec, norm);
    float r0 = sqr((nt - 1) / (nt + 1));
    float r_theta = r0 + (1 - r0) * fast_pow(1 - c, 5);
    if (r_theta < 0) return r_theta;
    v_norm(out.d, tvec);
    return r_theta;
}

// unset: get_hit calls of the bounce loop that returned true without ever filling the record (every reported t was
// +inf or NaN): the reference then reads a hitrec it never wrote (src/render_thread.cpp:121,141,180), so its result
// for such a pixel is not defined and nothing here can pin it
struct PixelStats { uint64_t closest = 0, any = 0, unset = 0; };

// src/render_thread.cpp:43-485 for ONE pixel (i = column, j = row counted from the bottom)
void render_pixel(const orc_scene& sc, const rt_options& o, int i, int j, float out_rgb[3], PixelStats& st) {
    const Rng rng{o.jitter_mode, o.seed};
    const int W = o.width, H = o.height;
    const float aspect_ratio = H / (float)W;                                     // :34-40
    const float d = sc.cam.near_clip;
    const float a = d * tanf(3.1415926f * o.hfov / 360);
    const float l = -a, r = a, t = a * aspect_ratio, b = -t;
    const int spp = o.spp, spr = o.spr, D = o.max_bounce;
    const float Dr = (float)o.max_refrac_bounce;                                 // the reference passes it as float
    const uint32_t pixel = (uint32_t)(j * W + i);
    const size_t nl = sc.lights.size();

    std::vector<Ray> pool;                                                        // LIFO, src/raypool.cpp:30-70
    std::vector<float> cs;                                                        // drawn colours, in draw order
    for (int x = 0; x < spp; ++x)
        for (int y = 0; y < spp; ++y) {                                           // :57-109
            const uint32_t s = (uint32_t)(x * spp + y);
            const uint32_t node = node_root(o.seed, pixel, s);
            float unit_len = 1.0f / spp;
            float jx = rng.draw(node, SITE_PRIMARY, 0, 0) % 100 / 100.0f;
            float jy = rng.draw(node, SITE_PRIMARY, 0, 1) % 100 / 100.0f;
            float u = l + (r - l) * (i + unit_len * x + jx * unit_len) / W;
            float v = b + (t - b) * (j + unit_len * y + jy * unit_len) / H;
            vec3 du, dv, dw, dir;
            v_scale(du, sc.cam.side, u); v_scale(dv, sc.cam.up, v); v_scale(dw, sc.cam.w, -d);
            v_add(dir, du, dv); v_add(dir, dir, dw); v_norm(dir, dir);
            Ray rr{};
            rr.type = RAY_REFLECT; v_copy(rr.g.e, sc.cam.pos); v_copy(rr.g.d, dir);
            rr.depth = D; rr.contrib = 1; rr.tpc = 0; rr.refraci = 1; rr.weight = 1;
            rr.c_cache[0] = rr.c_cache[1] = rr.c_cache[2] = 0; rr.node = node;
            pool.push_back(rr);
            Ray sr = rr; sr.type = RAY_SHADOW;
            pool.push_back(sr);
        }

    std::vector<float> shadow_rec(nl, 1.0f);                                     // :115 (see header: defined start)
    float weight_sum[3] = {0, 0, 0};
    while (!pool.empty()) {                                                       // :119-482
        Ray ray = pool.back(); pool.pop_back();
        Hit rec;
        bool hit = get_hit(sc, ray.g, o.eps, INFINITY, rec);
        st.closest++;
        if (hit && rec.mesh < 0) st.unset++;
        vec3 c = {0, 0, 0};
        bool draw = false;
        if (hit) {
            if (ray.type == RAY_SHADOW) {                                         // :134-171
                for (size_t li = 0; li < nl; ++li) {
                    float sh_frac = 0;
                    vec3 ppos; get_point(ray.g, rec.t, ppos);
                    vec3 ldir; light_dir(sc.lights[li], ppos, ldir, rng, ray.node, SITE_SH_LIGHT, (uint32_t)li);
                    RayG exact; v_copy(exact.e, ppos); v_norm(exact.d, ldir);   // Ray ctor normalises, ray.cpp:11-30
                    const int n_rays = spr + 1;
                    for (int k = 0; k < n_rays; ++k) {
                        RayG lr;
                        if (k < spr) split_dir(exact, o.ray_eps, rng, ray.node, SITE_SH_SPLIT, (uint32_t)li, (uint32_t)k, lr);
                        else lr = exact;
                        Hit sh;
                        st.any++;
                        if (get_hit(sc, lr, o.eps, sc.cam.far_clip, sh)) sh_frac += 0; else sh_frac += 1;
                    }
                    sh_frac /= n_rays;
                    shadow_rec[li] = sh_frac;
                }
            } else if (rec.mesh >= 0) {                                             // :177-410
                // rec.mesh < 0: a mesh reported a hit whose t is NaN or +inf, which get_hit
                // never copies (:15); the reference then compares an unset mesh name with every mesh (:178-180),
                // matches none and draws nothing
                const MeshO& mesh = *sc.meshes[rec.mesh];
                const rt_material_desc& M = sc.mats[mesh.material];
                const bool use_refl = (M.kind == RT_MAT_PHONG || M.kind == RT_MAT_REFRAC);
                const bool use_refr = (M.kind == RT_MAT_REFRAC);
                const float m_refl = use_refl ? M.reflectivity : 0.f;
                vec3 p; get_point(ray.g, rec.t, p);
                if (!use_refl) {                                                    // :187-200
                    vec3 contrib;
                    apply_shade(sc, M, contrib, p, sc.cam.pos, rec.norm, shadow_rec, rng, ray.node);
                    if (ray.contrib != 1) v_scale(contrib, contrib, 1 - ray.tpc);
                    v_add(ray.c_cache, ray.c_cache, contrib);
                    v_copy(c, ray.c_cache);
                    for (int q = 0; q < 3; ++q) weight_sum[q] += ray.weight;
                    draw = true;
                } else if (ray.depth > 0 && !use_refr) {                            // :207-283
                    vec3 self = {0, 0, 0};
                    float prev = m_refl;
                    ray.contrib = 1 - prev;
                    for (int n = ray.depth; n < D; ++n) { ray.contrib = prev * (1 - m_refl); prev = prev - ray.contrib; }
                    apply_shade(sc, M, self, p, sc.cam.pos, rec.norm, shadow_rec, rng, ray.node);
                    v_scale(self, self, ray.contrib);
                    if (ray.depth == D) v_copy(ray.c_cache, self); else v_add(ray.c_cache, ray.c_cache, self);
                    ray.tpc += ray.contrib;
                    Ray rr{};
                    rr.depth = ray.depth - 1; rr.type = RAY_REFLECT;
                    reflect(ray.g, rec.norm, rec.t, rr.g);
                    v_copy(rr.c_cache, ray.c_cache);
                    rr.contrib = ray.contrib; rr.tpc = ray.tpc;
                    rr.weight = ray.weight; rr.refraci = ray.refraci;               // defined (header)
                    std::vector<Ray> subs;
                    for (int k = 0; k < spr; ++k) {
                        Ray s = rr;
                        split_dir(rr.g, o.ray_eps, rng, ray.node, SITE_REFLECT_SPLIT, 0, (uint32_t)k, s.g);
                        s.node = node_child(ray.node, (uint32_t)k);
                        subs.push_back(s);
                    }
                    rr.node = node_child(ray.node, (uint32_t)spr);
                    subs.push_back(rr);
                    for (const Ray& s : subs) {                                     // :269-280
                        pool.push_back(s);
                        Ray sh = s; sh.type = RAY_SHADOW; sh.depth = -1; sh.g = rr.g;   // exact direction
                        pool.push_back(sh);
                    }
                } else if (ray.depth <= 0 && !use_refr) {                           // :286-293
                    v_copy(c, ray.c_cache);
                    for (int q = 0; q < 3; ++q) weight_sum[q] += ray.weight;
                    draw = true;
                } else if (ray.depth > 0 && use_refr) {                             // :300-397
                    vec3 c_add;
                    if (ray.refraci == M.refractivity) {
                        for (int q = 0; q < 3; ++q) c_add[q] = M.transp[q] * expf(-M.transp[q] * rec.t);
                    } else c_add[0] = c_add[1] = c_add[2] = 1;
                    Ray rf{}, rt{};
                    rf.contrib = 1; rf.type = RAY_REFRACTIVE; rf.tpc = 0;
                    rf.depth = (ray.refraci == 1) ? D : ray.depth - 1;
                    rf.refraci = M.refractivity;
                    rt.contrib = 1; rt.type = RAY_REFRACTIVE; rt.tpc = 0;
                    rt.depth = (ray.refraci == 1) ? (int)Dr : ray.depth - 1;
                    rt.refraci = M.refractivity;
                    rf.node = node_child(ray.node, 0); rt.node = node_child(ray.node, 1);
                    reflect(ray.g, rec.norm, rec.t, rf.g);
                    float r_theta = refract(ray.g, rec.norm, ray.refraci, M.refractivity, rec.t, rt.g);
                    if (r_theta <= 0) {                                             // :345-360
                        rf.weight = ray.weight;
                        v_mul(rf.c_cache, ray.c_cache, c_add);
                        pool.push_back(rf);
                        Ray sh = rf; sh.type = RAY_SHADOW; sh.depth = -1; pool.push_back(sh);
                    } else {                                                        // :362-394
                        rf.weight = r_theta * ray.weight;
                        rt.weight = (1 - r_theta) * ray.weight;
                        v_mul(rf.c_cache, ray.c_cache, c_add);
                        v_scale(rf.c_cache, rf.c_cache, r_theta);
                        pool.push_back(rf);
                        Ray sh = rf; sh.type = RAY_SHADOW; sh.depth = -1; pool.push_back(sh);
                        v_mul(rt.c_cache, ray.c_cache, c_add);
                        v_scale(rt.c_cache, rt.c_cache, 1 - r_theta);
                        pool.push_back(rt);
                        Ray sh2 = rt; sh2.type = RAY_SHADOW; sh2.depth = -1; pool.push_back(sh2);
                    }
                } else if (ray.depth <= 0 && use_refr) {                            // :400-406
                    v_copy(c, ray.c_cache);
                    for (int q = 0; q < 3; ++q) weight_sum[q] += ray.weight;
                    draw = true;
                }
            }
        } else {                                                                    // :414-430
            if ((ray.type == RAY_REFLECT && ray.depth < D) || (ray.type == RAY_REFRACTIVE && ray.depth < Dr)) {
                v_copy(c, ray.c_cache);
                for (int q = 0; q < 3; ++q) weight_sum[q] += ray.weight;
                draw = true;
            }
        }
        if (draw) { cs.push_back(c[0]); cs.push_back(c[1]); cs.push_back(c[2]); }
    }
    vec3 sum = {0, 0, 0};                                                           // rasterizer.cpp:88-103
    for (size_t q = 0; q < cs.size(); q += 3) { sum[0] = sum[0] + cs[q]; sum[1] = sum[1] + cs[q + 1]; sum[2] = sum[2] + cs[q + 2]; }
    for (int q = 0; q < 3; ++q) out_rgb[q] = sum[q] / weight_sum[q];
}

inline float fast_sqrt_bits(float x) {                                              // fast_math.cpp:4-12
    uint32_t i; memcpy(&i, &x, 4);
    i += 127u << 23; i >>= 1;
    float r; memcpy(&r, &i, 4);
    return r;
}

}  // namespace

extern "C" {

int orc_scene_create(const rt_mesh_desc* meshes, int n_meshes, const rt_material_desc* materials, int n_materials,
                     const rt_light_desc* lights, int n_lights, const rt_camera_desc* camera, orc_scene** out) {
    if (!meshes || n_meshes <= 0 || !materials || !camera || !out) return RT_ERR_INVALID;
    orc_scene* sc = new orc_scene();
    sc->mats.assign(materials, materials + n_materials);
    if (n_lights > 0) sc->lights.assign(lights, lights + n_lights);
    sc->cam = *camera;
    int gid = 0;
    for (int mi = 0; mi < n_meshes; ++mi) {                                         // AABBTree.cpp:3-16
        const rt_mesh_desc& md = meshes[mi];
        MeshO* m = new MeshO();
        memcpy(m->name, md.name, 32); m->material = md.material;
        m->tris.resize(md.n_tris);
        for (int k = 0; k < md.n_tris; ++k) {
            Tri& t = m->tris[k];
            const float* P = md.pos + (size_t)k * 9; const float* Nn = md.nrm + (size_t)k * 9;
            v_copy(t.a, P); v_copy(t.b, P + 3); v_copy(t.c, P + 6);
            v_copy(t.an, Nn); v_copy(t.bn, Nn + 3); v_copy(t.cn, Nn + 6);
            t.mesh = mi; t.id = gid++;
        }
        // mesh box over all corner positions, geometry.cpp:349-356
        v_copy(m->aabb.lo, m->tris[0].a); v_copy(m->aabb.hi, m->tris[0].a);
        for (const Tri& t : m->tris) {
            const float* vs[3] = {t.a, t.b, t.c};
            for (int q = 0; q < 3; ++q) { Box b; v_copy(b.lo, vs[q]); v_copy(b.hi, vs[q]); box_union(m->aabb, b, m->aabb); }
        }
        std::vector<Surf*> surf;
        for (Tri& t : m->tris) surf.push_back(&t);
        m->root = new Node(surf, 0, 64);
        sc->meshes.push_back(m);
    }
    *out = sc;
    return RT_OK;
}

void orc_scene_destroy(orc_scene* sc) { delete sc; }

int orc_scene_info(const orc_scene* sc, int64_t counts[4]) {
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    for (MeshO* m : sc->meshes) { counts[0] += (int64_t)m->tris.size(); m->root->shape(1, counts); }
    return RT_OK;
}

// get_hit in batch form; counters[0] += box tests, counters[1] += triangle tests (may be NULL)
int orc_trace(const orc_scene* sc, const float* org, const float* dir, float t0, float t1, int64_t n, int any_hit,
              float* t_out, int32_t* tri_out, float* nrm_out, uint64_t* counters) {
    Counters cnt; g_cnt = counters ? &cnt : nullptr;
    for (int64_t k = 0; k < n; ++k) {
        RayG r; v_copy(r.e, org + 3 * k); v_copy(r.d, dir + 3 * k);
        Hit h;
        bool hit = get_hit(*sc, r, t0, t1, h);
        if (any_hit) {
            t_out[k] = hit ? 0.f : INFINITY; if (tri_out) tri_out[k] = hit ? 0 : -1;
        } else {
            // a mesh can report a hit whose t is NaN (0/0 on a degenerate triangle) which get_hit then never
            // keeps (src/render_thread.cpp:15): hit is true but the record stays empty
            bool kept = hit && h.tri >= 0;
            t_out[k] = kept ? h.t : INFINITY;
            if (tri_out) tri_out[k] = kept ? h.tri : -1;
            if (nrm_out) for (int q = 0; q < 3; ++q) nrm_out[3 * k + q] = kept ? h.norm[q] : 0.f;
        }
    }
    if (counters) { counters[0] += cnt.box_tests; counters[1] += cnt.tri_tests; }
    g_cnt = nullptr;
    return RT_OK;
}

// RenderThread::operator() for the rect [x0,x1) x [y0,y1); rgb_out[(y-y0)*(x1-x0) + (x-x0)][3]
int orc_render_rect(const orc_scene* sc, const rt_options* o, int x0, int y0, int x1, int y1, float* rgb_out, rt_stats* stats) {
    if (!sc || !o || !rgb_out || x0 < 0 || y0 < 0 || x1 > o->width || y1 > o->height || x0 > x1 || y0 > y1) return RT_ERR_INVALID;
    PixelStats st;
    const int w = x1 - x0;
    for (int i = x0; i < x1; ++i)
        for (int j = y0; j < y1; ++j)
            render_pixel(*sc, *o, i, j, rgb_out + ((size_t)(j - y0) * w + (i - x0)) * 3, st);
    if (stats) { memset(stats, 0, sizeof(*stats)); stats->rays_closest = st.closest; stats->rays_any = st.any; stats->ref_get_hit_calls = st.closest + st.any; stats->reserved_[0] = (int32_t)std::min<uint64_t>(st.unset, 0x7fffffff); }
    return RT_OK;
}

// renderFrame's fork/join partitioning (src/main.cpp:474-507) over the whole image with hw = n_threads:
// prog_disp_span blocks, each split into sub-tiles of block_len = min(w,h)/thread_n, one std::thread per sub-tile,
// joined per block.  Returns wall seconds; rgb_out is [H][W][3], row 0 at the bottom.
double orc_render_frame_mt(const orc_scene* sc, const rt_options* o, int hw_threads, int prog_disp_span, float* rgb_out, rt_stats* stats) {
    const int thread_n = (int)ceilf(fast_sqrt_bits((float)hw_threads));             // main.cpp:109
    const int W = o->width, H = o->height;
    std::atomic<uint64_t> n_closest(0), n_any(0), n_unset(0);
    auto t_start = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < W; i += prog_disp_span)
        for (int j = 0; j < H; j += prog_disp_span) {
            int cbx = i + prog_disp_span, cby = j + prog_disp_span;
            const int img_w = prog_disp_span - cbx / W * (cbx % W);
            const int img_h = prog_disp_span - cby / H * (cby % H);
            int block_len = (img_w > img_h) ? img_h / thread_n : img_w / thread_n;
            if (block_len < 1) block_len = 1;
            std::vector<std::thread> ths;
            for (int sx = i; sx < i + img_w; sx += block_len)
                for (int sy = j; sy < j + img_h; sy += block_len) {
                    int cbi = sx + block_len, ci = i + img_w, cbj = sy + block_len, cj = j + img_h;
                    const int ex = sx + block_len - cbi / ci * (cbi % ci);
                    const int ey = sy + block_len - cbj / cj * (cbj % cj);
                    ths.emplace_back([=, &n_closest, &n_any, &n_unset]() {
                        PixelStats st;
                        for (int x = sx; x < ex; ++x)
                            for (int y = sy; y < ey; ++y)
                                render_pixel(*sc, *o, x, y, rgb_out + ((size_t)y * W + x) * 3, st);
                        n_closest += st.closest; n_any += st.any; n_unset += st.unset;
                    });
                }
            for (std::thread& t : ths) t.join();
        }
    auto t_stop = std::chrono::high_resolution_clock::now();
    if (stats) { memset(stats, 0, sizeof(*stats)); stats->rays_closest = n_closest; stats->rays_any = n_any; stats->ref_get_hit_calls = n_closest + n_any; stats->reserved_[0] = (int32_t)std::min<uint64_t>(n_unset, 0x7fffffff); }
    return std::chrono::duration<double>(t_stop - t_start).count();
}

// src/main.cpp:157-167
void orc_resolve_u8(const float* rgb, int W, int H, uint8_t* out) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c) {
                float v = rgb[((size_t)(H - y - 1) * W + x) * 3 + c];
                double s = floor(v * 256.0);
                int q = (s != s) ? std::numeric_limits<int>::min() : (s >= 2147483648.0 || s < -2147483648.0 ? std::numeric_limits<int>::min() : (int)s);
                out[((size_t)y * W + x) * 3 + c] = (uint8_t)std::max(0, std::min(255, q));
            }
}

}  // extern "C"
