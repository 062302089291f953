// Kernel bodies of the wavefront pipeline.  One call = the work of one lane; the __global__ wrappers in
// rt_kernels.hip only compute the lane's index and reduce counters.
//
// Pipeline for one band of pixels (all stages cite the reference's src/render_thread.cpp):
//   generate : level-0 nodes = primary render rays (:57-94); implicit - every kernel recomputes the primary ray of its
//              node from the node's number instead of reading a stored copy
//   per level, while the level is not empty:
//     extend : nearest hit of every node's ray (:122 for render rays; the shadow-type twin that the reference
//              traces along the same ray, :97-104 / :269-280, has the same answer and is not traced again)
//     shadow : the light-visibility rays of every node's companion shadow ray (:134-171), any-hit
//     spawn  : children of reflective / refractive hits (:207-283, :300-397) into the next level
//   replay   : per pixel, walks the stored tree in the reference's LIFO pop order and does all colour
//              arithmetic (:128-438) and the final sum/weight division (src/rasterizer.cpp:88-103) in that order.
// Splitting geometry (which rays exist and what they hit) from colour is possible because no ray's direction
// depends on a colour; it makes the float sums happen in exactly the reference's order.
#pragma once
#include "rt_device.h"

namespace rt {

enum { RAY_SHADOW = 0, RAY_REFLECT = 1, RAY_REFRACTIVE = 2,      // src/ray.h:10-14
       RAY_NONE = 3 };                                          // padding slot of a partly covered pixel tile
enum { RT_MAX_LEVELS = 20 };

// Everything the kernels after the nearest-hit search need to know about a node, in TWO 32-byte records: the replay
// walks a pixel's ray tree node by node, and neighbouring lanes are at unrelated nodes below the first level, so every
// separately stored field costs a memory sector of its own.  NodeRec is what every node of the walk needs (which
// children, what kind of ray, its weight, what it hit, the light counts); NodeGeo - where it hit - only the nodes that are
// shaded or spawn: a ray that hit nothing, or a mirror at the last bounce, draws its parent's colour and never reads it
// (most nodes of a room of mirrors: the memory system fetches 32-byte sectors, so the replay moves a third fewer bytes).
struct alignas(32) NodeGeo {
    float px, py, pz; uint32_t key;   // hit point e + t d (src/ray.cpp:32-35); path key of the node's ray
    float nx, ny, nz, t;              // unit normal at the hit; t
};
struct alignas(32) NodeRec {
    int32_t kid_base;                 // first child in the next level (written by spawn)
    int32_t flags;                    // type | depth << 8 | rank << 16 | REC_SAME_MEDIUM; byte 3 = number of children
    float weight;                     // src/ray.h:36
    int32_t tri;                      // -1: the ray hit nothing; otherwise the MATERIAL index of the hit mesh (all that the
                                      // later stages want from the triangle, and three dependent loads nearer)
    float aux;                        // Schlick term r_theta of a refractive spawn
    int32_t comp;                     // node (this level) whose ray the companion shadow ray repeats
    uint8_t unocc[8];                 // per light: light rays that reached it, out of spr+1; RT_NO_COMPANION_HIT where
                                      // the companion shadow ray hit nothing and none was traced (written by shadow)
};
enum { REC_SAME_MEDIUM = 1 << 21 };   // the ray travels in the hit material's own medium (refraci == refractivity)

// What one popped render ray adds to its pixel (src/render_thread.cpp:434-438) and which node draws next.
struct alignas(32) DrawRec { float cx, cy, cz, w; int32_t next_lvl, next_idx, pad0, pad1; };

struct Level {
    // A stored node's ray in 32 bytes.  Its origin is not stored: it is the hit point of its parent, which the parent's
    // own results hold (ppos), and a node's children sit next to each other, so the lanes of a wave fetch few of them.
    float4* dir;       // xyz ray direction, w = weight (src/ray.h:36)
    int4* meta;        // x parent index in the previous level, y = type | depth << 2 | rank << 6 | medium << 10 (medium: 0 =
                       // the ray travels where the refractive index is 1, m + 1 = inside material m: refraci of
                       // src/render_thread.cpp:328,337 is always one of these), z = path key of the ray,
                       // w = index (this level) of the node whose ray the companion shadow ray repeats
    const float4* ppos;   // results of the previous level: xyz hit point of node k at ppos[k * pstride]
    int32_t pstride;      // 1: the previous level is the implicit one (its pos array), 2: a stored one (its NodeGeo array)
    NodeRec* rec;      // results of a stored level
    NodeGeo* geo;
    // results of the implicit level 0, one array per field: there neighbouring lanes ARE neighbouring nodes, so
    // separate arrays are read and written fully coalesced and fields a scene never uses cost nothing
    float4* hit;       // xyz unit normal at the hit, w = t
    float4* pos;       // xyz hit point, w = bits(path key)
    int32_t* tri;      // triangle position or -1
    int2* kids;        // first child in the next level, number of children (scenes that can spawn only)
    float* aux;        // Schlick term of a refractive spawn
    uint8_t* unocc;    // [n][n_lights]
    // two-pass replay only (see replay_sample): what each drawing node adds to its pixel, chained in pop order
    struct DrawRec* draw;   // per node of this level
    int2* first;            // level 0: (level, node) of the first drawing node of the sample's tree, (-1, -1) if none
    int32_t implicit;  // 1: level 0 - org/dir/meta are not stored; node g IS the primary ray of slot g / spp^2,
                       // sample g % spp^2, and is recomputed wherever it is needed (cheaper than 48 B per sample)
};

// ---- access to a node's results in either layout
// >= 0 where node i's ray hit something (a triangle position at level 0, a material index below: see NodeRec)
RT_HD int32_t res_tri(const Level& L, int64_t i) { return L.implicit ? L.tri[i] : L.rec[i].tri; }
// material index of the hit; v = res_tri(L, i) >= 0
RT_HD int32_t res_material(const DScene& sc, const Level& L, int32_t v) { return L.implicit ? sc.mesh_mat[sc.info[v].mesh] : v; }
RT_HD float4 res_pos(const Level& L, int64_t i) { return L.implicit ? L.pos[i] : *reinterpret_cast<const float4*>(&L.geo[i].px); }
RT_HD float4 res_hit(const Level& L, int64_t i) { return L.implicit ? L.hit[i] : *reinterpret_cast<const float4*>(&L.geo[i].nx); }
// children the nearest-hit search announced for node i
RT_HD int res_kid_count(const Level& L, int64_t i) { return L.implicit ? L.kids[i].y : (int)((uint32_t)L.rec[i].flags >> 24); }
RT_HD void set_kids(const Level& L, int64_t i, int32_t base, int cnt) {
    if (L.implicit) { int2 kd; kd.x = base; kd.y = cnt; L.kids[i] = kd; }
    else { L.rec[i].kid_base = base; reinterpret_cast<uint8_t*>(&L.rec[i].flags)[3] = (uint8_t)cnt; }
}

// What the kernels need to know about a node's ray.
struct NodeIn {
    V3 o, d; uint32_t key; float weight;
    int32_t packed;      // type | depth << 8 | rank << 16
    int32_t medium;      // what Level::meta keeps of refraci: 0, or material + 1
    int32_t refraci;     // float bits
    int32_t comp;        // node whose ray the companion shadow ray repeats
};

RT_HD int meta_type(int y) { return y & 0xff; }
RT_HD int meta_depth(int y) { return (y >> 8) & 0xff; }
RT_HD int meta_pack(int type, int depth, int rank) { return type | (depth << 8) | (rank << 16); }
RT_HD int meta_store(int type, int depth, int rank, int medium) { return type | (depth << 2) | (rank << 6) | (int)((uint32_t)medium << 10); }   // Level::meta.y

RT_HD V3 f4v(float4 a) { return v3(a.x, a.y, a.z); }
RT_HD float4 mk4(V3 a, float w) { float4 r; r.x = a.x; r.y = a.y; r.z = a.z; r.w = w; return r; }
RT_HD V3 point_at(V3 e, V3 d, float t) { return vadd(e, vscale(d, t)); }    // src/ray.cpp:32-35

// ---------------------------------------------------------------------------------------------- generate
// Level-0 nodes are stored tile by tile: 4x4 pixels x spp^2 samples are consecutive, and inside a tile 2x2 pixels are,
// so that the 64 lanes of a wave work on a compact patch of the image (rays that stay together longer) instead of a
// one-pixel-high strip - whether a wave takes 64 nodes (nearest hit) or 64 / (spr + 1) nodes (light rays).
// Slot of pixel (px, py) of the band (px = column in the rect, py = row in the band), times spp^2, + sample.
RT_HD int64_t pixel_slot(const RenderParams& P, int px, int py) {
    const int tiles_x = (P.rw + 3) >> 2;
    const int q = (px & 1) | ((py & 1) << 1) | ((px & 2) << 1) | ((py & 2) << 2);   // 2x2 blocks of 2x2 pixels
    return ((int64_t)((py >> 2) * tiles_x + (px >> 2)) << 4) + q;
}

// Node g of the implicit level 0: g = slot * spp^2 + sample (src/render_thread.cpp:57-94).
RT_HD NodeIn primary_node(const DScene& sc, const RenderParams& P, int64_t g) {
    const uint32_t slot = fdiv((uint32_t)g, P.div_spp2);
    const int s = (int)((uint32_t)g - slot * (uint32_t)(P.spp * P.spp));
    const uint32_t tile = slot >> 4, tile_y = fdiv(tile, P.div_tiles_x), tile_x = tile - tile_y * (uint32_t)((P.rw + 3) >> 2);
    const int q = (int)(slot & 15);
    const int px = (int)tile_x * 4 + ((q & 1) | ((q >> 1) & 2)), py = (int)tile_y * 4 + (((q >> 1) & 1) | ((q >> 2) & 2));
    NodeIn n;
    n.o = v3p(sc.cam.pos); n.d = v3(0, 0, 0); n.key = 0; n.weight = 1.f;
    n.refraci = RT_F2I(1.f); n.medium = 0; n.comp = (int32_t)g;
    if (px >= P.rw || py >= P.band_rows) {                         // padding of a partly covered tile: never traced
        n.packed = meta_pack(RAY_NONE, 0, 0);
        return n;
    }
    const int i = P.x0 + px, j = image_row(P, P.band_y0 + py);
    n.key = node_root(P.seed, (uint32_t)(j * P.W + i), (uint32_t)s);
    n.d = primary_dir(sc, P, i, j, s, n.key);
    n.packed = meta_pack(RAY_REFLECT, P.D, 0);
    return n;
}

RT_HD NodeIn node_in(const DScene& sc, const RenderParams& P, const Level& L, int64_t i) {
    if (L.implicit) return primary_node(sc, P, i);
    const float4 d = L.dir[i];
    const int4 m = L.meta[i];
    const float4 o = L.ppos[(int64_t)m.x * L.pstride];
    NodeIn n;
    n.o = f4v(o); n.d = f4v(d); n.key = (uint32_t)m.z; n.weight = d.w;
    n.packed = meta_pack(m.y & 3, (m.y >> 2) & 15, (m.y >> 6) & 15);
    n.medium = (int32_t)((uint32_t)m.y >> 10);
    n.refraci = n.medium ? RT_F2I(sc.mats[n.medium - 1].refractivity) : RT_F2I(1.f);
    n.comp = m.w;
    return n;
}

// ------------------------------------------------------------------------------------------------ extend
// src/ray.cpp:63-96.  Returns the Schlick term; out = refracted direction (valid when the term is >= 0).
RT_HD float refract_dir(V3 d, V3 norm, float n, float nt, V3& out) {
    const float dn = vdot(d, norm);
    V3 term1 = vscale(norm, -1 * dn);
    term1 = vadd(d, term1);
    term1 = vscale(term1, n / nt);
    const float q = n / nt;
    const V3 term2 = vscale(norm, sqrtf(1 - (q * q) * (1 - dn * dn)));
    const V3 tvec = vsub(term1, term2);
    float c;
    if (dn < 0) c = -dn; else c = vdot(tvec, norm);
    const float rr = (nt - 1) / (nt + 1);
    const float r0 = rr * rr;
    const float r_theta = r0 + (1 - r0) * fast_pow_i(1 - c, 5);
    if (r_theta < 0) return r_theta;
    out = vnorm(tvec);
    return r_theta;
}

RT_HD int4 mk_i4(int x, int y, int z, int w) { int4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

// What a node's triangle field holds when the reference's get_hit returns true with a record it never filled (every t
// reported to it was +inf or NaN): not a hit for anything here, but not the reference's "hit nothing" branch either.
enum { RT_UNSET_HIT = -2 };

// Count byte of a (hit, light) pair whose light rays are not traced because nothing they find can change the colour
// (rt_options.skip_unlit): the replay multiplies a zero by count / (spr + 1), any finite count gives the same bits.
enum { RT_UNLIT = 254 };
RT_HD V3 lambert_shade(V3 ca, V3 cr, V3 cl, V3 n, V3 l);

// true when the light cannot change what apply_shade returns for a hit at p with normal n: the default material ignores
// the shadow record (src/materials.cpp:4-8), a Lambert hit adds lambert_shade(...) * record - zero times anything
// finite when the term itself is zero.  The term is computed by the function the replay calls, on the same inputs.
RT_HD bool light_is_idle(const DScene& sc, const RenderParams& P, const Material& M, V3 p, V3 n, uint32_t key, int li) {
    if (M.kind == 0) return true;
    if (M.kind != 1) return false;
    const V3 l = light_dir(sc.lights[li], p, P, key, SITE_SHADE_LIGHT, (uint32_t)li);
    const V3 c = lambert_shade(v3p(M.ambient), v3p(M.base), v3p(sc.lights[li].color), n, l);
    return c.x == 0.f && c.y == 0.f && c.z == 0.f;
}

// returns the number of children this node will spawn
// suspect (may be NULL; pruning traversal only): set when the hit is a triangle the reference never looks at - the caller
// queues the node for the same function with VET == true, which redoes all of the node's results (see traverse)
template <bool EXACT, class ST, bool VET = false>
RT_HD int body_extend(const DScene& sc, const RenderParams& P, const Level& L, ST& st, int64_t i,
                      uint32_t* n_nodes = nullptr, uint32_t* n_tris = nullptr, bool* suspect = nullptr) {
    const NodeIn in = node_in(sc, P, L, i);
    HitRec h;
    bool hit = false;
    const V3 d = in.d;
    if (meta_type(in.packed) != RAY_NONE)
        hit = traverse<false, EXACT, ST, VET>(sc, st, in.o, d, P.eps, INFINITY, h, n_nodes, n_tris, suspect);
    float r_theta = 0.f;
    int kids = 0, flags = in.packed;
    float4 pk = mk4(v3(0, 0, 0), 0.f), hn = mk4(v3(0, 0, 0), INFINITY);
    // no hit: -1, or RT_UNSET_HIT where the reference's get_hit says "hit" without a record (see traverse_oct): such a
    // node draws nothing and has no children, and the light rays the reference sends from its non-finite hit point
    // are counted, not traced
    // On the pruning traversal's own tree only for a ray that is itself not finite (every triangle of any tree then
    // passes): whether a FINITE ray meets an edge-on triangle whose test divides 0 by 0 depends on the boxes around it,
    // and widened boxes let through what the reference's never test.
    const bool any_tree = EXACT || !(fabsf(in.o.x) + fabsf(in.o.y) + fabsf(in.o.z) + fabsf(d.x) + fabsf(d.y) + fabsf(d.z) < INFINITY);
    const int none = (!hit && meta_type(in.packed) != RAY_NONE && h.tri == RT_UNSET_HIT && any_tree) ? RT_UNSET_HIT : -1;
    int kind = 0, mat = none;
    if (hit) {
        const V3 n = hit_normal(sc, h);
        pk = mk4(point_at(in.o, d, h.t), RT_I2F((int32_t)in.key));
        hn = mk4(n, h.t);
        mat = sc.mesh_mat[sc.info[h.tri].mesh];
        const Material& M = sc.mats[mat];
        kind = M.kind;
        if (RT_I2F(in.refraci) == M.refractivity) flags |= REC_SAME_MEDIUM;
        const int depth = meta_depth(in.packed);
        if (P.can_spawn && depth > 0 && M.kind == 2) kids = P.spr + 1;
        if (P.can_spawn && depth > 0 && M.kind == 3) {
            V3 td;
            r_theta = refract_dir(d, n, RT_I2F(in.refraci), M.refractivity, td);
            kids = (r_theta <= 0) ? 1 : 2;                              // src/render_thread.cpp:345
        }
    }
    if (L.implicit) {
        L.tri[i] = hit ? h.tri : none;
        if (P.can_spawn) { int2 kd; kd.x = -1; kd.y = kids; L.kids[i] = kd; }
        if (hit) {
            L.hit[i] = hn; L.pos[i] = pk;
            if (kind == 3) L.aux[i] = r_theta;
            if (P.skip_unlit) {
                // every pair gets its byte: the light-ray kernel overwrites the zeros with counts and leaves the others
                const Material& M = sc.mats[mat];
                for (int li = 0; li < sc.n_lights; ++li)
                    L.unocc[i * sc.n_lights + li] = light_is_idle(sc, P, M, f4v(pk), f4v(hn), in.key, li) ? (uint8_t)RT_UNLIT : (uint8_t)0;
            }
        }
    } else {
        NodeRec* R = L.rec + i;
        if (hit) {
            NodeGeo* G = L.geo + i;
            *reinterpret_cast<float4*>(&G->px) = pk;
            *reinterpret_cast<float4*>(&G->nx) = hn;
        }
        *reinterpret_cast<int4*>(&R->kid_base) = mk_i4(-1, flags | (kids << 24), RT_F2I(in.weight), mat);
        *reinterpret_cast<int4*>(&R->aux) = mk_i4(RT_F2I(r_theta), in.comp, 0, 0);
    }
    return kids;
}

// ------------------------------------------------------------------------------------------------ shadow
// Item w = node * n_lights + light: the spr jittered rays and the exact ray toward one light from the hit point of
// the node's companion shadow ray (src/render_thread.cpp:146-161); unocc[w] = how many of them reached the light.
struct ShadowItem { V3 p, exact; uint32_t key; int li; bool idle; };   // idle: counted but not traced, see RT_UNLIT, RT_UNSET_HIT

// false: the shadow-type ray missed, nothing is traced and the record stays stale
RT_HD bool shadow_item(const DScene& sc, const RenderParams& P, const Level& L, int64_t w, ShadowItem& it) {
    const uint32_t i = fdiv((uint32_t)w, P.div_lights);
    it.li = (int)((uint32_t)w - i * (uint32_t)sc.n_lights);
    it.idle = false;
    const int comp = L.implicit ? (int)i : L.meta[i].w;
    const int comp_tri = res_tri(L, comp);
    if (comp_tri < 0) { it.idle = (comp_tri == RT_UNSET_HIT); return false; }
    if (P.skip_unlit && L.unocc[w] == (uint8_t)RT_UNLIT) { it.idle = true; return false; }   // level 0 only: such scenes have no other
    const float4 pk = res_pos(L, comp);
    it.p = f4v(pk);
    it.key = L.implicit ? (uint32_t)RT_F2I(pk.w) : (uint32_t)L.meta[i].z;
    it.exact = vnorm(light_dir(sc.lights[it.li], it.p, P, it.key, SITE_SH_LIGHT, (uint32_t)it.li));   // Ray ctor normalises
    return true;
}

// Count byte of an item whose companion ray hit nothing (no light ray is traced, the reference keeps the stale shadow
// record, src/render_thread.cpp:134): larger than any count (at most spr + 1 <= 16).  The replay reads it instead of
// fetching the companion's record - one dependent load less per node of a walk that is all dependent loads.
enum { RT_NO_COMPANION_HIT = 255 };
RT_HD void store_unocc(const DScene& sc, const RenderParams& P, const Level& L, int64_t w, uint32_t unocc) {
    if (L.implicit) { L.unocc[w] = (uint8_t)unocc; return; }
    const uint32_t i = fdiv((uint32_t)w, P.div_lights);
    L.rec[i].unocc[(uint32_t)w - i * (uint32_t)sc.n_lights] = (uint8_t)unocc;
}

// one more light ray of item w reached its light after all (the second, vetted walk of a suspect ray): the item's count
// byte + 1, as an atomic on the word that holds it (other lanes may be at its neighbours)
RT_HD void bump_unocc(const DScene& sc, const RenderParams& P, const Level& L, int64_t w) {
    uint8_t* p;
    if (L.implicit) p = L.unocc + w;
    else { const uint32_t i = fdiv((uint32_t)w, P.div_lights); p = L.rec[i].unocc + ((uint32_t)w - i * (uint32_t)sc.n_lights); }
    const uintptr_t a = (uintptr_t)p;
    atomicAdd(reinterpret_cast<uint32_t*>(a & ~(uintptr_t)3), 1u << (8 * (a & 3)));
}

// ray k of the item (k < spr: jittered copy, k == spr: the exact direction); true when it reaches the light
template <bool EXACT, class ST, bool VET = false>
RT_HD bool shadow_ray(const DScene& sc, const RenderParams& P, const ShadowItem& it, ST& st, int k,
                      uint32_t* n_nodes = nullptr, uint32_t* n_tris = nullptr, bool* suspect = nullptr) {
    const V3 d = (k < P.spr) ? split_dir(it.exact, P.ray_eps, P, it.key, SITE_SH_SPLIT, (uint32_t)it.li, (uint32_t)k) : it.exact;
    HitRec h;
    return !traverse<true, EXACT, ST, VET>(sc, st, it.p, d, P.eps, sc.cam.far_clip, h, n_nodes, n_tris, suspect);
}

// ------------------------------------------------------------------------------------------------- spawn
// Child k of node i, written to N[at]: the kernel gives every CHILD a lane, so that consecutive lanes write consecutive
// records.  first = place of the node's first child (the caller reserved res_kid_count(L, i) places from there).
// Reflective hit: spr jittered copies of the mirror direction and the mirror direction itself, whose node (first + spr)
// is the companion of all of them; refractive hit: the reflected ray and, unless reflection is total, the refracted one.
RT_HD void spawn_child(const DScene& sc, const RenderParams& P, const Level& L, const Level& N, uint32_t first, uint32_t at, int64_t i, int k) {
    const int tri = res_tri(L, i);
    const NodeIn in = node_in(sc, P, L, i);
    const int depth = meta_depth(in.packed);
    const Material& M = sc.mats[res_material(sc, L, tri)];
    const V3 d = in.d;
    const V3 n = f4v(res_hit(L, i));
    const float pp = 2.f * vdot(d, n);                                  // mirror direction, src/linmath.h:59-65
    const V3 rd = v3(d.x - pp * n.x, d.y - pp * n.y, d.z - pp * n.z);
    const uint32_t ck = node_child(in.key, (uint32_t)k);
    if (M.kind == 2) {                                                  // src/render_thread.cpp:239-280
        const V3 dk = (k < P.spr) ? split_dir(rd, P.ray_eps, P, in.key, SITE_REFLECT_SPLIT, 0, (uint32_t)k) : rd;
        N.dir[at] = mk4(dk, in.weight);
        int4 cm; cm.x = (int)i; cm.y = meta_store(RAY_REFLECT, depth - 1, k, in.medium); cm.z = (int)ck; cm.w = (int)(first + P.spr);
        N.meta[at] = cm;
    } else {                                                            // src/render_thread.cpp:318-394
        const float refraci = RT_I2F(in.refraci);
        V3 td = v3(0, 0, 0);
        const float r_theta = refract_dir(d, n, refraci, M.refractivity, td);
        const bool both = !(r_theta <= 0);
        const int medium = res_material(sc, L, tri) + 1;                  // the children travel in the hit material (:328, :337)
        int4 cm; cm.x = (int)i; cm.z = (int)ck; cm.w = (int)at;
        if (k == 0) {
            N.dir[at] = mk4(rd, both ? r_theta * in.weight : in.weight);
            cm.y = meta_store(RAY_REFRACTIVE, (refraci == 1) ? P.D : depth - 1, 0, medium);
        } else {
            N.dir[at] = mk4(td, (1 - r_theta) * in.weight);
            cm.y = meta_store(RAY_REFRACTIVE, (refraci == 1) ? P.Dr : depth - 1, 1, medium);
        }
        N.meta[at] = cm;
    }
}

// ------------------------------------------------------------------------------------------------ replay
// src/shaders/basic_shaders.cpp:6-13
RT_HD V3 lambert_shade(V3 ca, V3 cr, V3 cl, V3 n, V3 l) {
    const float r = max_std(0.0f, vdot(n, l));
    V3 c = vscale(cl, r);
    c = vadd(c, ca);
    return vmul(cr, c);
}
// src/shaders/basic_shaders.cpp:17-30
RT_HD V3 phong_shade(V3 cl, V3 cp, V3 l, V3 e, V3 n, int p, V3 clambert) {
    V3 r = vscale(n, 2.0f * vdot(l, n));
    r = vsub(r, l);
    r = vnorm(r);
    V3 cph = vscale(cp, fast_pow_i(max_std(0.0f, vdot(e, r)), p));
    cph = vmul(cl, cph);
    V3 c = vadd(clambert, cph);
    c.x = min_std(1.0f, c.x); c.y = min_std(1.0f, c.y); c.z = min_std(1.0f, c.z);
    return c;
}
// src/materials.cpp:4-8,28-43,68-88,120-140; sh(li) = the shadow record's entry for light li
template <class Sh>
RT_HD V3 apply_shade_with(const DScene& sc, const RenderParams& P, const Material& M, V3 p, V3 n, Sh sh, uint32_t key) {
    if (M.kind == 0) return v3(1.f, 1.f, 0.f);
    V3 r = v3(0, 0, 0);
    V3 view = v3(0, 0, 0);
    if (M.kind != 1) view = vnorm(vsub(v3p(sc.cam.pos), p));
    for (int li = 0; li < sc.n_lights; ++li) {
        const V3 l = light_dir(sc.lights[li], p, P, key, SITE_SHADE_LIGHT, (uint32_t)li);
        const V3 lc = v3p(sc.lights[li].color);
        V3 c_add = lambert_shade(v3p(M.ambient), v3p(M.base), lc, n, l);
        if (M.kind != 1) c_add = phong_shade(lc, v3p(M.spec), l, view, n, (int)M.shininess, c_add);
        c_add = vscale(c_add, sh(li));
        r = vadd(r, c_add);
    }
    return r;
}
struct ShadowRecArray { const float* v; RT_HDM float operator()(int li) const { return v[li]; } };
RT_HD V3 apply_shade(const DScene& sc, const RenderParams& P, const Material& M, V3 p, V3 n, const float* sh, uint32_t key) {
    ShadowRecArray a; a.v = sh;
    return apply_shade_with(sc, P, M, p, n, a, key);
}

// What a node leaves behind for its children while the replay is below it.
struct RFrame {
    int32_t base, next;       // first child, next child to pop (counts down: the last pushed pops first)
    int32_t refr, count;      // children of a refractive hit; how many
    float r_theta;            // Schlick term of a refractive hit
    V3 c; float contrib, tpc; // colour state handed to the children (src/render_thread.cpp:245-248, :348, :367-382)
};

// Where the replay keeps the frames of the levels it is below.  The first RT_FRAME_LDS_LEVELS levels live in LDS, two
// 16-byte words per level and lane laid out [level][word][lane of the wave] (a wave's access to one word is conflict
// free); deeper levels go to a per-lane array in scratch.  Scratch loads and stores are vector-memory instructions like
// the record loads this walk consists of - with every frame in scratch they were two fifths of its memory instructions.
#ifndef RT_FRAME_LDS_LEVELS
#define RT_FRAME_LDS_LEVELS 5     // 10 KB per wave; 3 / 5 / 8 measured: mirror room 108.1 / 106.7 / 107.0 ms, glass balls 64.5 / 63.6 / 63.9
#endif
typedef float rt_f4 __attribute__((ext_vector_type(4)));
struct FrameStore {
    RT_LDS_PTR rt_f4* lds;        // this lane's column: word w of level l is lds[(l * 2 + w) * 64]
    RFrame* spill;                // levels from RT_FRAME_LDS_LEVELS on (indexed by level)
    RT_HDM void put(int lvl, const RFrame& F) {
        if (lvl < RT_FRAME_LDS_LEVELS) {
            rt_f4 a, b;
            a.x = RT_I2F(F.base); a.y = RT_I2F((F.next & 0xff) | (F.count << 8) | (F.refr << 16)); a.z = F.r_theta; a.w = F.contrib;
            b.x = F.c.x; b.y = F.c.y; b.z = F.c.z; b.w = F.tpc;
            lds[(lvl * 2) * 64] = a; lds[(lvl * 2 + 1) * 64] = b;
        } else spill[lvl] = F;
    }
    // next child to pop at this level (-1: none left)
    RT_HDM int next(int lvl) const {
        if (lvl < RT_FRAME_LDS_LEVELS) return (int)(int8_t)(RT_F2I(lds[(lvl * 2) * 64].y) & 0xff);
        return spill[lvl].next;
    }
    // the frame, with its next-child counter already moved on; the returned next is the child to pop now
    RT_HDM RFrame take(int lvl) {
        RFrame F;
        if (lvl < RT_FRAME_LDS_LEVELS) {
            const rt_f4 a = lds[(lvl * 2) * 64], b = lds[(lvl * 2 + 1) * 64];
            const int pk = RT_F2I(a.y);
            F.base = RT_F2I(a.x); F.next = (int)(int8_t)(pk & 0xff); F.count = (pk >> 8) & 0xff; F.refr = (pk >> 16) & 1;
            F.r_theta = a.z; F.contrib = a.w; F.c = v3(b.x, b.y, b.z); F.tpc = b.w;
            reinterpret_cast<RT_LDS_PTR int32_t*>(lds + (lvl * 2) * 64)[1] = (pk & ~0xff) | ((F.next - 1) & 0xff);
        } else {
            F = spill[lvl];
            spill[lvl].next = F.next - 1;
        }
        return F;
    }
};

// A material in four 16-byte loads (the array is 64-byte aligned and a material is 64 bytes).
RT_HD Material load_material(const Material* p) {
    static_assert(sizeof(Material) == 64, "material layout");
    const float4* q = reinterpret_cast<const float4*>(p);
    float4 v[4] = {q[0], q[1], q[2], q[3]};
    Material m;
    memcpy(&m, v, sizeof(m));
    return m;
}

// One sample: walks the ray tree under level-0 node g in the reference's pop order (the newest pushed ray pops
// first) and hands every drawn colour to emit(level, node, colour, weight).
// The per-pixel shadow record (src/render_thread.cpp:134-171) is refreshed by the level-0 node before anything below
// it is shaded (a primary ray that hit nothing has no children), so its value on entry never matters and samples can
// be replayed independently of each other; the ORDER of the additions to the pixel is what must be kept.
template <class Emit>
RT_HD void replay_sample(const DScene& sc, const RenderParams& P, const Level* lv, int64_t g, Emit& emit, FrameStore& fr) {
    const float Dr = (float)P.Dr;
    float shadow_rec[8];
    for (int q = 0; q < 8; ++q) shadow_rec[q] = 1.f;
    int lvl = 0;
    int64_t i = g;
    V3 cc = v3(0, 0, 0);
    float contrib = 1.f, tpc = 0.f;
    for (;;) {
        // ---- pop one render ray: node i of level lvl with colour state (cc, contrib, tpc)
        const Level& L = lv[lvl];
        // kid_base, flags (type, depth, children), weight, triangle; then the rest of what this node needs
        int4 head; float aux = 0.f; int comp_tri;
        uint32_t un_lo = 0, un_hi = 0;               // per-light counts, one byte each
        if (L.implicit) {
            head.w = L.tri[i];
            head.y = meta_pack(RAY_REFLECT, P.D, 0); head.z = RT_F2I(1.f); head.x = -1;
            if (P.can_spawn && head.w >= 0) { const int2 kd = L.kids[i]; head.x = kd.x; head.y |= kd.y << 24; }
            comp_tri = head.w;                       // the companion repeats the node's own ray
            if (comp_tri >= 0)
                for (int li = 0; li < sc.n_lights; ++li) {
                    const uint32_t u = L.unocc[i * sc.n_lights + li];
                    if (li < 4) un_lo |= u << (8 * li); else un_hi |= u << (8 * (li - 4));
                }
        } else {
            // the whole record in a few 16-byte loads: lanes are at unrelated records, and the memory pipeline pays
            // per instruction and lane, not per byte
            const NodeRec* R = L.rec + i;
            head = *reinterpret_cast<const int4*>(&R->kid_base);
            const int4 tail = *reinterpret_cast<const int4*>(&R->aux);
            aux = RT_I2F(tail.x);
            un_lo = (uint32_t)tail.z; un_hi = (uint32_t)tail.w;
            comp_tri = ((un_lo & 255u) == (uint32_t)RT_NO_COMPANION_HIT) ? -1 : 0;   // what the light-ray kernel noted
        }
        const int type = meta_type(head.y), depth = meta_depth(head.y), n_kids = (int)((uint32_t)head.y >> 24);
        const float weight = RT_I2F(head.z);
        const int tri = head.w;
        // its companion shadow-type ray pops just before it (:134-171)
        if (comp_tri >= 0)
            for (int li = 0; li < sc.n_lights; ++li) {
                float sh_frac = (float)(((li < 4 ? un_lo >> (8 * li) : un_hi >> (8 * (li - 4)))) & 255u);   // +1 per light ray that hit nothing
                sh_frac /= (float)(P.spr + 1);
                shadow_rec[li] = sh_frac;
            }
        bool draw = false, has_kids = false;
        V3 c = v3(0, 0, 0);
        if (tri >= 0) {
            const Material M = load_material(sc.mats + res_material(sc, L, tri));
            // where the ray hit: fetched only by the branches that shade (a mirror at its last bounce does not)
            float4 hn; V3 n, p; uint32_t key;
            auto fetch_geometry = [&]() {
                hn = res_hit(L, i);
                n = f4v(hn);
                if (L.implicit) {
                    // this kernel waits for memory: recomputing the primary ray is cheaper than reading the hit point
                    const NodeIn in = primary_node(sc, P, i);
                    p = point_at(in.o, in.d, hn.w); key = in.key;
                } else {
                    const float4 pk = res_pos(L, i);
                    p = f4v(pk); key = (uint32_t)RT_F2I(pk.w);
                }
            };
            const bool use_refl = (M.kind == 2 || M.kind == 3), use_refr = (M.kind == 3);
            if (!use_refl) {                                                 // :187-200
                fetch_geometry();
                V3 add = apply_shade(sc, P, M, p, n, shadow_rec, key);
                if (contrib != 1) add = vscale(add, 1 - tpc);
                cc = vadd(cc, add);
                c = cc; draw = true;
            } else if (depth > 0 && !use_refr) {                             // :207-283
                float prev = M.reflectivity;
                contrib = 1 - prev;
                for (int q = depth; q < P.D; ++q) { contrib = prev * (1 - M.reflectivity); prev = prev - contrib; }
                fetch_geometry();
                V3 self = apply_shade(sc, P, M, p, n, shadow_rec, key);
                self = vscale(self, contrib);
                if (depth == P.D) cc = self; else cc = vadd(cc, self);
                tpc += contrib;
                RFrame F;
                F.base = head.x; F.count = n_kids; F.next = n_kids - 1; F.refr = 0; F.r_theta = 0.f;
                F.c = cc; F.contrib = contrib; F.tpc = tpc;
                fr.put(lvl, F);
                has_kids = n_kids > 0;
            } else if (depth <= 0) {                                         // :286-293, :400-406
                c = cc; draw = true;
            } else {                                                         // :300-397
                V3 c_add;
                if (L.implicit ? (1.f == M.refractivity) : ((head.y & REC_SAME_MEDIUM) != 0)) {
                    const float t_hit = res_hit(L, i).w;
                    c_add.x = M.transp[0] * expf(-M.transp[0] * t_hit);
                    c_add.y = M.transp[1] * expf(-M.transp[1] * t_hit);
                    c_add.z = M.transp[2] * expf(-M.transp[2] * t_hit);
                } else c_add = v3(1, 1, 1);
                RFrame F;
                F.base = head.x; F.count = n_kids; F.next = n_kids - 1; F.refr = 1; F.r_theta = L.implicit ? L.aux[i] : aux;
                F.c = vmul(cc, c_add); F.contrib = 1.f; F.tpc = 0.f;
                fr.put(lvl, F);
                has_kids = n_kids > 0;
            }
        } else if (tri == -1) {                                              // :414-430; RT_UNSET_HIT draws nothing (:178-180 match no mesh)
            if ((type == RAY_REFLECT && depth < P.D) || (type == RAY_REFRACTIVE && depth < Dr)) { c = cc; draw = true; }
        }
        if (draw) emit(lvl, i, c, weight);                                   // :434-438
        // ---- next ray to pop: the newest pushed one
        if (!has_kids) {
            --lvl;
            while (lvl >= 0 && fr.next(lvl) < 0) --lvl;
            if (lvl < 0) break;
        }
        const RFrame F = fr.take(lvl);
        const int k = F.next;
        i = F.base + k;
        cc = F.c; contrib = F.contrib; tpc = F.tpc;
        if (F.refr && F.count == 2) cc = vscale(cc, k == 0 ? F.r_theta : 1 - F.r_theta);   // :368, :382
        ++lvl;
    }
}

// replay_sample for a frame none of whose materials spawns rays: a sample's walk is its level-0 node alone - no frames
// of parents, no colour state handed down - and what is left of the generic walk is one shading call.  Same operations
// in the same order on the same inputs (the sum starts from the zero the generic walk starts from); c is the colour
// drawn, with weight 1.  false: the sample draws nothing (a primary ray that hit nothing, src/render_thread.cpp:414-430
// with depth == D, or a hit without a record).
struct ShadowRecBytes {
    const uint8_t* u; float rays;
    RT_HDM float operator()(int li) const { float f = (float)u[li]; f /= rays; return f; }     // :163
};
RT_HD bool replay_flat_sample(const DScene& sc, const RenderParams& P, const Level& L, int64_t i, V3& c) {
    const int tri = L.tri[i];
    if (tri < 0) return false;
    const Material M = load_material(sc.mats + res_material(sc, L, tri));
    const float4 hn = L.hit[i];
    const NodeIn in = primary_node(sc, P, i);
    ShadowRecBytes sh; sh.u = L.unocc + i * sc.n_lights; sh.rays = (float)(P.spr + 1);
    const V3 add = apply_shade_with(sc, P, M, point_at(in.o, in.d, hn.w), f4v(hn), sh, in.key);
    c = vadd(v3(0, 0, 0), add);
    return true;
}

// adds to the pixel at once (src/rasterizer.cpp:92-96)
struct EmitSum {
    V3 sum, wsum;
    RT_HDM void operator()(int, int64_t, V3 c, float weight) { sum = vadd(sum, c); wsum = vadd(wsum, v3(weight, weight, weight)); }
};

// One pixel: pl = row_in_band * rw + column_in_rect.  Writes rgb_out[pl] = sum(colours) / sum(weights).
RT_HD void body_replay(const DScene& sc, const RenderParams& P, const Level* lv, int64_t pl, float* rgb_out, FrameStore& fr) {
    const int spp2 = P.spp * P.spp;
    const int64_t slot = pixel_slot(P, (int)(pl % P.rw), (int)(pl / P.rw));
    EmitSum e; e.sum = v3(0, 0, 0); e.wsum = v3(0, 0, 0);
    // push order of src/render_thread.cpp:57-109 is sample 0 first, so the last sample pops first
    for (int s = spp2 - 1; s >= 0; --s) replay_sample(sc, P, lv, slot * spp2 + s, e, fr);
    rgb_out[pl * 3 + 0] = e.sum.x / e.wsum.x;
    rgb_out[pl * 3 + 1] = e.sum.y / e.wsum.y;
    rgb_out[pl * 3 + 2] = e.sum.z / e.wsum.z;
}

// ---- the same in two passes, for scenes with many samples per pixel and deep ray trees: one lane per SAMPLE shades
// (64 or more times the lanes, each with a short walk), storing what every drawing node adds and chaining those nodes
// in pop order; then one lane per pixel only adds the stored colours in that order.
struct EmitChain {
    const Level* lv; int64_t g;
    int plvl; int64_t pidx; V3 pc; float pw;      // the previous drawing node, written once its successor is known
    RT_HDM void flush(int nl, int64_t ni) {
        if (plvl < 0) { int2 f; f.x = nl; f.y = (int)ni; lv[0].first[g] = f; return; }
        DrawRec d; d.cx = pc.x; d.cy = pc.y; d.cz = pc.z; d.w = pw; d.next_lvl = nl; d.next_idx = (int32_t)ni; d.pad0 = 0; d.pad1 = 0;
        lv[plvl].draw[pidx] = d;
    }
    RT_HDM void operator()(int l, int64_t i, V3 c, float weight) { flush(l, i); plvl = l; pidx = i; pc = c; pw = weight; }
};
RT_HD void body_shade(const DScene& sc, const RenderParams& P, const Level* lv, int64_t g, FrameStore& fr) {
    EmitChain e; e.lv = lv; e.g = g; e.plvl = -1; e.pidx = 0; e.pc = v3(0, 0, 0); e.pw = 0.f;
    if (lv[0].tri[g] >= 0) replay_sample(sc, P, lv, g, e, fr);    // a primary ray that hit nothing draws nothing
    e.flush(-1, -1);    // closes the chain, or marks the sample as drawing nothing
}
RT_HD void body_sum(const RenderParams& P, const Level* lv, int64_t pl, float* rgb_out) {
    const int spp2 = P.spp * P.spp;
    const int64_t slot = pixel_slot(P, (int)(pl % P.rw), (int)(pl / P.rw));
    V3 sum = v3(0, 0, 0), wsum = v3(0, 0, 0);
    for (int s = spp2 - 1; s >= 0; --s) {
        int2 at = lv[0].first[slot * spp2 + s];
        while (at.x >= 0) {
            const DrawRec d = lv[at.x].draw[at.y];
            sum = vadd(sum, v3(d.cx, d.cy, d.cz));
            wsum = vadd(wsum, v3(d.w, d.w, d.w));
            at.x = d.next_lvl; at.y = d.next_idx;
        }
    }
    rgb_out[pl * 3 + 0] = sum.x / wsum.x;
    rgb_out[pl * 3 + 1] = sum.y / wsum.y;
    rgb_out[pl * 3 + 2] = sum.z / wsum.z;
}

// ------------------------------------------------------------------------------------------- batch trace
template <bool ANY, bool EXACT>
RT_HD void body_trace(const DScene& sc, const float* org, const float* dir, float t0, float t1, int64_t i,
                      float* t_out, int32_t* tri_out, float* nrm_out, uint32_t* cnt_nodes, uint32_t* cnt_tris, StackLds& st, bool vet) {
    const V3 o = v3p(org + 3 * i), d = v3p(dir + 3 * i);
    HitRec h;
    uint32_t nn = 0, nt = 0;
    bool suspect = false;
    bool hit = traverse<ANY, EXACT>(sc, st, o, d, t0, t1, h, cnt_nodes ? &nn : nullptr, cnt_tris ? &nt : nullptr, (EXACT || !vet) ? nullptr : &suspect);
    // a batch is no hot path: a suspect answer of the pruning traversal is redone at once, every candidate vetted
    if (!EXACT && suspect) hit = traverse<ANY, false, StackLds, true>(sc, st, o, d, t0, t1, h, cnt_nodes ? &nn : nullptr, cnt_tris ? &nt : nullptr);
    if (ANY) {
        t_out[i] = hit ? 0.f : INFINITY;
        if (tri_out) tri_out[i] = hit ? 0 : -1;
    } else {
        t_out[i] = hit ? h.t : INFINITY;
        if (tri_out) tri_out[i] = hit ? sc.info[h.tri].id : -1;
        if (nrm_out) {
            const V3 n = hit ? hit_normal(sc, h) : v3(0, 0, 0);
            nrm_out[3 * i] = n.x; nrm_out[3 * i + 1] = n.y; nrm_out[3 * i + 2] = n.z;
        }
    }
    if (cnt_nodes) cnt_nodes[i] = nn;
    if (cnt_tris) cnt_tris[i] = nt;
}

// src/main.cpp:157-167: out row 0 is the top; (int) of a NaN or out-of-range double is INT_MIN on x86-64
RT_HD void body_resolve_u8(const float* rgb, int W, int H, int64_t i, uint8_t* out) {
    const int64_t px = i / 3; const int c = (int)(i - px * 3);
    const int y = (int)(px / W), x = (int)(px - (int64_t)y * W);
    const float v = rgb[((int64_t)(H - y - 1) * W + x) * 3 + c];
    const double s = floor((double)v * 256.0);
    int q;
    if (!(s == s) || s >= 2147483648.0 || s < -2147483648.0) q = INT32_MIN; else q = (int)s;
    q = q > 255 ? 255 : q; q = q < 0 ? 0 : q;
    out[i] = (uint8_t)q;
}

}  // namespace rt
