// Host-side tree build + flattening.
//
// Builds, per mesh, the same midpoint-partition tree as the reference's
// BVHNode::BVHNode(std::vector<Surface*>, int AXIS, int depth) (src/geometry.cpp:61-153, called from
// src/AABBTree.cpp:10 with AXIS 0 and depth 64) and writes it straight into the flat device layout
// (rt_types.h).  Differences in representation, not in the set of boxes and triangle groups:
//   * a node's box is stored in its parent (both children's boxes in one 64-byte record);
//   * the bottom nodes of the reference (a node holding one triangle, two triangles, or a TriangleSet)
//     become leaves: a run of consecutive triangles plus the box the reference tests before them;
//   * every triangle gets a tie rank: wherever the reference prefers one candidate on equal t (right child
//     over left, src/geometry.cpp:171-175; lower index inside a set, :428; earlier mesh,
//     src/render_thread.cpp:15) that candidate gets the lower rank, so "lowest rank wins ties" reproduces its
//     choice under any traversal order;
//   * a second pair of child references per node ("fine") replaces leaves of more than RT_FINE_LEAF triangles
//     by a median-split subtree over the leaf's triangles; it changes which boxes are tested, never the answer
//     (ties are decided by rank), and is what the pruning traversal uses;
//   * the meshes' roots hang under a small top tree over mesh boxes instead of being looped over
//     (src/render_thread.cpp:10); a parent box is the exact union of its children, so it never fails
//     where a descendant passes.
// Where the reference reads uninitialised memory (TriangleSet box seed, src/geometry.cpp:413; box of a
// depth-capped node, :65-69) the union of the triangles' boxes is used.
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace rt {
namespace {

struct Box { float lo[3], hi[3]; };

inline void box_union(Box& a, const Box& b) {                 // src/geometry.cpp:3-12
    for (int i = 0; i < 3; ++i) { a.lo[i] = fminf(a.lo[i], b.lo[i]); a.hi[i] = fmaxf(a.hi[i], b.hi[i]); }
}

struct Built { int32_t ref, ref_fine; Box box; int depth; };

constexpr int RT_FINE_LEAF = 4;
#ifndef RT_TOP_BIG_ITEM
#define RT_TOP_BIG_ITEM 0.25f  // share of a group's box area from which a mesh is set apart in the top tree (0.5 / 0.25 / 0.1 measured)
#endif
#ifndef RT_SAH_LEAF
#define RT_SAH_LEAF 3       // most triangles in a leaf of the surface-area-heuristic tree (2, 3, 4, 6, 8 measured: 3 is 1 - 4 % ahead of 4)
#endif

inline void set_node(Node64& n, const Built& L, const Built& R) {
    memcpy(n.lbox, L.box.lo, 12); memcpy(n.lbox + 3, L.box.hi, 12);
    memcpy(n.rbox, R.box.lo, 12); memcpy(n.rbox + 3, R.box.hi, 12);
    n.lref = L.ref; n.rref = R.ref; n.lref_fine = L.ref_fine; n.rref_fine = R.ref_fine;
}

struct Builder {
    const float* pos;      // current mesh: [n][3][3]
    const float* nrm;
    int mesh_index;
    int id_base;           // global id of the mesh's first triangle
    int rank_base;         // rank of the mesh's most preferred triangle (earlier meshes rank lower)
    std::vector<Box> tbox; // per triangle of the current mesh
    FlatScene* out;

    Box tri_box(int k) const {                                  // src/geometry.cpp:314-321
        const float* a = pos + (size_t)k * 9; const float* b = a + 3; const float* c = a + 6;
        Box r;
        for (int i = 0; i < 3; ++i) {
            r.lo[i] = fminf(a[i], fminf(b[i], c[i]));
            r.hi[i] = fmaxf(a[i], fmaxf(b[i], c[i]));
        }
        return r;
    }

    int next_rank = 0;     // ranks follow the reference's preference order (this builder's emission order)

    void emit(int t, int run, int run_fine, int rank) {
        const float* a = pos + (size_t)t * 9; const float* b = a + 3; const float* c = a + 6;
        Tri48 q;
        q.ax = a[0]; q.ay = a[1]; q.az = a[2]; q.run = run;
        q.abx = a[0] - b[0]; q.aby = a[1] - b[1]; q.abz = a[2] - b[2]; q.run_fine = run_fine;
        q.acx = a[0] - c[0]; q.acy = a[1] - c[1]; q.acz = a[2] - c[2]; q.rank = rank;
        out->tris.push_back(q);
        TriInfo fi; fi.id = id_base + t; fi.mesh = mesh_index;
        out->info.push_back(fi);
        const float* nn = nrm + (size_t)t * 9;
        out->nrm.insert(out->nrm.end(), nn, nn + 9);
    }

    struct Ranked { int t, rank; };

    // Subtree over the triangles of one big leaf: median split of box centres along the widest axis.
    // `left` = triangles of the whole leaf still to be emitted after this group (for the coarse run lengths).
    Built refine(Ranked* v, int n, int left) {
        Built r; r.depth = 0;
        r.box = tbox[v[0].t];
        for (int k = 1; k < n; ++k) box_union(r.box, tbox[v[k].t]);
        if (n <= RT_FINE_LEAF) {
            r.ref = r.ref_fine = ~(int32_t)out->tris.size();
            for (int k = 0; k < n; ++k) emit(v[k].t, left + n - k, n - k, v[k].rank);
            return r;
        }
        float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < n; ++k)
            for (int i = 0; i < 3; ++i) {
                const float c = tbox[v[k].t].lo[i] + tbox[v[k].t].hi[i];
                cmin[i] = fminf(cmin[i], c); cmax[i] = fmaxf(cmax[i], c);
            }
        int ax = 0;
        for (int i = 1; i < 3; ++i) if (cmax[i] - cmin[i] > cmax[ax] - cmin[ax]) ax = i;
        const int mid = n / 2;
        std::nth_element(v, v + mid, v + n, [&](const Ranked& x, const Ranked& y) {
            const float cx = tbox[x.t].lo[ax] + tbox[x.t].hi[ax], cy = tbox[y.t].lo[ax] + tbox[y.t].hi[ax];
            return cx < cy || (cx == cy && x.rank < y.rank);
        });
        const int32_t idx = (int32_t)out->nodes.size();
        out->nodes.emplace_back();
        const Built L = refine(v, mid, left + (n - mid));
        const Built R = refine(v + mid, n - mid, left);
        Built Lf = L, Rf = R;                      // inside the subtree both reference pairs are the fine ones
        Lf.ref = L.ref_fine; Rf.ref = R.ref_fine;
        set_node(out->nodes[idx], Lf, Rf);
        r.ref = r.ref_fine = idx;
        r.depth = 1 + std::max(L.depth, R.depth);
        return r;
    }

    Built leaf(const int* ids, int n) {
        Built r; r.depth = 0;
        r.box = tbox[ids[0]];
        for (int k = 0; k < n; ++k) box_union(r.box, tbox[ids[k]]);
        r.ref = ~(int32_t)out->tris.size();
        std::vector<Ranked> v(n);
        for (int k = 0; k < n; ++k) { v[k].t = ids[k]; v[k].rank = rank_base + next_rank++; }
        if (n <= RT_FINE_LEAF) {
            r.ref_fine = r.ref;
            for (int k = 0; k < n; ++k) emit(v[k].t, n - k, n - k, v[k].rank);
        } else {
            const Built f = refine(v.data(), n, 0);
            r.ref_fine = f.ref_fine;
            r.depth = f.depth;
        }
        out->n_leaves++;
        return r;
    }

    Built build(const std::vector<int>& s, int axis, int depth) {
        const int N = (int)s.size();
        if (depth == 0 || N == 1) return leaf(s.data(), N);
        if (N == 2) { const int sw[2] = {s[1], s[0]}; return leaf(sw, 2); }   // the right one wins ties
        Box sb = tbox[s[0]];
        for (int t : s) box_union(sb, tbox[t]);
        const float m = sb.lo[axis] + (sb.hi[axis] - sb.lo[axis]) / 2;        // :97, midpoint on the incoming axis
        int div_dif = std::numeric_limits<int>::max(), best_axis = 0;
        for (int a = 0; a < 3; ++a) {                                          // :104-121, the same m for every axis
            int nl = 0;
            for (int t : s) {
                const Box& b = tbox[t];
                const float bm = b.lo[axis] + (b.hi[axis] - b.lo[axis]) / 2;
                if (bm < m) ++nl;
            }
            const int d = std::abs(nl - (N - nl));
            if (d < div_dif) { div_dif = d; best_axis = axis; }
            axis = (axis + 1) % 3;
        }
        axis = best_axis;
        std::vector<int> ll, rl;
        for (int t : s) {
            const Box& b = tbox[t];
            const float bm = b.lo[axis] + (b.hi[axis] - b.lo[axis]) / 2;
            if (bm < m) ll.push_back(t); else rl.push_back(t);
        }
        if (ll.empty()) return leaf(rl.data(), N);                             // :135-139
        if (rl.empty()) return leaf(ll.data(), N);                             // :142-146
        const int32_t idx = (int32_t)out->nodes.size();
        out->nodes.emplace_back();
        const Built R = build(rl, (axis + 1) % 3, depth - 1);                  // right first: lower tie positions
        const Built L = build(ll, (axis + 1) % 3, depth - 1);
        set_node(out->nodes[idx], L, R);
        Built r; r.ref = r.ref_fine = idx; r.box = L.box; box_union(r.box, R.box);
        r.depth = 1 + std::max(L.depth, R.depth);
        return r;
    }
};

// Top tree over mesh roots: median split of box centres along the widest axis.
Built build_top(FlatScene* out, std::vector<Built>& roots, int lo, int hi) {
    if (hi - lo == 1) return roots[lo];
    float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = lo; k < hi; ++k)
        for (int i = 0; i < 3; ++i) {
            const float c = 0.5f * (roots[k].box.lo[i] + roots[k].box.hi[i]);
            cmin[i] = fminf(cmin[i], c); cmax[i] = fmaxf(cmax[i], c);
        }
    int ax = 0;
    for (int i = 1; i < 3; ++i) if (cmax[i] - cmin[i] > cmax[ax] - cmin[ax]) ax = i;
    const int mid = (lo + hi) / 2;
    std::nth_element(roots.begin() + lo, roots.begin() + mid, roots.begin() + hi, [ax](const Built& a, const Built& b) {
        return a.box.lo[ax] + a.box.hi[ax] < b.box.lo[ax] + b.box.hi[ax];
    });
    const int32_t idx = (int32_t)out->nodes.size();
    out->nodes.emplace_back();
    const Built L = build_top(out, roots, lo, mid);
    const Built R = build_top(out, roots, mid, hi);
    set_node(out->nodes[idx], L, R);
    Built r; r.ref = r.ref_fine = idx; r.box = L.box; box_union(r.box, R.box);
    r.depth = 1 + std::max(L.depth, R.depth);
    return r;
}

}  // namespace

bool build_flat_scene(const rt_mesh_desc* meshes, int n_meshes, FlatScene* out, std::string* err) {
    out->nodes.clear(); out->tris.clear(); out->info.clear(); out->nrm.clear();
    out->n_leaves = 0; out->max_depth = 0;
    out->nodes.emplace_back();                       // node 0 = entry, filled in last
    std::vector<int> id_base((size_t)n_meshes + 1, 0);
    for (int mi = 0; mi < n_meshes; ++mi) {
        const rt_mesh_desc& md = meshes[mi];
        if (md.n_tris < 0 || (md.n_tris > 0 && (!md.pos || !md.nrm))) { *err = "mesh with null arrays or negative size"; return false; }
        id_base[(size_t)mi + 1] = id_base[(size_t)mi] + md.n_tris;
    }
    // The reference builds one tree per mesh (src/AABBTree.cpp:10); the trees are independent, so they are built on a
    // few host threads, each into a fragment of its own, and spliced in mesh order with shifted references.
    struct Frag { FlatScene f; Built root; };
    std::vector<Frag> frags((size_t)n_meshes);
    {
        std::atomic<int> next(0);
        std::atomic<bool> failed(false);
        auto work = [&]() {
            try {
                for (;;) {
                    const int mi = next.fetch_add(1);
                    if (mi >= n_meshes || failed.load()) break;
                    const rt_mesh_desc& md = meshes[mi];
                    if (md.n_tris == 0) continue;
                    Builder b;
                    b.pos = md.pos; b.nrm = md.nrm; b.mesh_index = mi; b.id_base = id_base[(size_t)mi]; b.rank_base = id_base[(size_t)mi];
                    b.out = &frags[(size_t)mi].f;
                    b.tbox.resize(md.n_tris);
                    std::vector<int> ids(md.n_tris);
                    for (int k = 0; k < md.n_tris; ++k) { b.tbox[k] = b.tri_box(k); ids[k] = k; }
                    frags[(size_t)mi].root = b.build(ids, 0, 64);         // src/AABBTree.cpp:10
                }
            } catch (...) { failed.store(true); }
        };
        const int n_threads = (int)std::max(1u, std::min(16u, std::min((unsigned)n_meshes, std::thread::hardware_concurrency())));
        std::vector<std::thread> pool;
        try {
            for (int t = 1; t < n_threads; ++t) pool.emplace_back(work);
        } catch (...) { /* fewer threads than hoped for */ }
        work();
        for (auto& th : pool) th.join();
        if (failed.load()) { *err = "out of host memory while building the trees"; return false; }
    }
    std::vector<Built> roots;
    for (int mi = 0; mi < n_meshes; ++mi) {
        if (meshes[mi].n_tris == 0) continue;
        Frag& fr = frags[(size_t)mi];
        const int32_t node_off = (int32_t)out->nodes.size(), tri_off = (int32_t)out->tris.size();
        auto shift = [&](int32_t ref) { return ref == RT_NONE ? ref : (ref >= 0 ? ref + node_off : ~(~ref + tri_off)); };
        for (Node64 n : fr.f.nodes) {
            n.lref = shift(n.lref); n.rref = shift(n.rref); n.lref_fine = shift(n.lref_fine); n.rref_fine = shift(n.rref_fine);
            out->nodes.push_back(n);
        }
        out->tris.insert(out->tris.end(), fr.f.tris.begin(), fr.f.tris.end());
        out->info.insert(out->info.end(), fr.f.info.begin(), fr.f.info.end());
        out->nrm.insert(out->nrm.end(), fr.f.nrm.begin(), fr.f.nrm.end());
        out->n_leaves += fr.f.n_leaves;
        Built r = fr.root;
        r.ref = shift(r.ref); r.ref_fine = shift(r.ref_fine);
        roots.push_back(r);
        fr.f = FlatScene();
    }
    if (roots.empty()) { *err = "scene has no triangles"; return false; }
    Node64 entry;
    memset(&entry, 0, sizeof(entry));
    int depth;
    if (roots.size() == 1) {
        memcpy(entry.lbox, roots[0].box.lo, 12); memcpy(entry.lbox + 3, roots[0].box.hi, 12);
        entry.lref = roots[0].ref; entry.rref = RT_NONE;
        entry.lref_fine = roots[0].ref_fine; entry.rref_fine = RT_NONE;
        depth = 1 + roots[0].depth;
    } else {
        const Built top = build_top(out, roots, 0, (int)roots.size());
        entry = out->nodes[top.ref];                  // copy the top node into slot 0 (its own slot stays unused)
        depth = top.depth;
    }
    out->nodes[0] = entry;
    out->max_depth = depth;
    return true;
}

namespace {

// ---- surface-area-heuristic build -----------------------------------------------------------------------------
struct SahRef { Box b; float c[3]; int32_t id; };   // triangle box, box centre, global triangle id

inline float half_area(const Box& b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

struct SahBuilder {
    enum { BINS = 16, LEAF = RT_SAH_LEAF, MAX_SAH_DEPTH = 64 };
    std::vector<SahRef> refs;
    const rt_mesh_desc* meshes;
    std::vector<int32_t> mesh_of, first_of;   // per mesh: first global id; per global id lookups go through these
    std::vector<int32_t> rank_of;             // by global triangle id
    const int32_t* rank_ptr = nullptr;        // the same table when it is owned elsewhere
    FlatScene* out;

    void emit_tri(int32_t id, int run) {
        const int mi = (int)(std::upper_bound(first_of.begin(), first_of.end(), id) - first_of.begin()) - 1;
        const int t = id - first_of[mi];
        const float* a = meshes[mi].pos + (size_t)t * 9; const float* b = a + 3; const float* c = a + 6;
        Tri48 q;
        q.ax = a[0]; q.ay = a[1]; q.az = a[2]; q.run = run;
        q.abx = a[0] - b[0]; q.aby = a[1] - b[1]; q.abz = a[2] - b[2]; q.run_fine = run;
        q.acx = a[0] - c[0]; q.acy = a[1] - c[1]; q.acz = a[2] - c[2]; q.rank = rank_ptr ? rank_ptr[id] : rank_of[id];
        out->tris.push_back(q);
        TriInfo fi; fi.id = id; fi.mesh = mi;
        out->info.push_back(fi);
        const float* nn = meshes[mi].nrm + (size_t)t * 9;
        out->nrm.insert(out->nrm.end(), nn, nn + 9);
    }

    Built build(int lo, int hi, int depth) {
        const int n = hi - lo;
        Built r; r.depth = 0;
        r.box = refs[lo].b;
        for (int k = lo + 1; k < hi; ++k) box_union(r.box, refs[k].b);
        if (n <= LEAF) {
            r.ref = r.ref_fine = ~(int32_t)out->tris.size();
            for (int k = 0; k < n; ++k) emit_tri(refs[lo + k].id, n - k);
            out->n_leaves++;
            return r;
        }
        float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = lo; k < hi; ++k)
            for (int i = 0; i < 3; ++i) { cmin[i] = fminf(cmin[i], refs[k].c[i]); cmax[i] = fmaxf(cmax[i], refs[k].c[i]); }
        int best_axis = -1, best_bin = 0; float best_cost = INFINITY;
        if (depth < MAX_SAH_DEPTH)
            for (int ax = 0; ax < 3; ++ax) {
                const float ext = cmax[ax] - cmin[ax];
                if (!(ext > 0.f) || !std::isfinite(ext)) continue;
                const float scale = BINS / ext;
                Box bb[BINS]; int cnt[BINS];
                for (int q = 0; q < BINS; ++q) { cnt[q] = 0; for (int i = 0; i < 3; ++i) { bb[q].lo[i] = INFINITY; bb[q].hi[i] = -INFINITY; } }
                for (int k = lo; k < hi; ++k) {
                    int q = (int)((refs[k].c[ax] - cmin[ax]) * scale);
                    q = q < 0 ? 0 : (q >= BINS ? BINS - 1 : q);
                    ++cnt[q]; box_union(bb[q], refs[k].b);
                }
                float right_area[BINS]; int right_cnt[BINS];
                Box acc; for (int i = 0; i < 3; ++i) { acc.lo[i] = INFINITY; acc.hi[i] = -INFINITY; }
                int c = 0;
                for (int q = BINS - 1; q > 0; --q) {
                    if (cnt[q]) box_union(acc, bb[q]);
                    c += cnt[q]; right_cnt[q] = c; right_area[q] = c ? half_area(acc) : 0.f;
                }
                for (int i = 0; i < 3; ++i) { acc.lo[i] = INFINITY; acc.hi[i] = -INFINITY; }
                c = 0;
                for (int q = 0; q < BINS - 1; ++q) {                 // split after bin q
                    if (cnt[q]) box_union(acc, bb[q]);
                    c += cnt[q];
                    if (c == 0 || right_cnt[q + 1] == 0) continue;
                    const float cost = half_area(acc) * c + right_area[q + 1] * right_cnt[q + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = q; }
                }
            }
        int mid;
        if (best_axis >= 0) {
            const float scale = BINS / (cmax[best_axis] - cmin[best_axis]);
            const float c0 = cmin[best_axis];
            const int ax = best_axis, bbin = best_bin;
            SahRef* m = std::partition(refs.data() + lo, refs.data() + hi, [=](const SahRef& x) {
                int q = (int)((x.c[ax] - c0) * scale);
                q = q < 0 ? 0 : (q >= BINS ? BINS - 1 : q);
                return q <= bbin;
            });
            mid = (int)(m - refs.data());
        } else {
            // all centres coincide (or the tree got too deep): halve by count along the widest axis
            int ax = 0;
            for (int i = 1; i < 3; ++i) if (cmax[i] - cmin[i] > cmax[ax] - cmin[ax]) ax = i;
            mid = lo + n / 2;
            std::nth_element(refs.begin() + lo, refs.begin() + mid, refs.begin() + hi, [ax](const SahRef& x, const SahRef& y) {
                return x.c[ax] < y.c[ax] || (x.c[ax] == y.c[ax] && x.id < y.id);
            });
        }
        if (mid <= lo || mid >= hi) mid = lo + n / 2;
        const int32_t idx = (int32_t)out->nodes.size();
        out->nodes.emplace_back();
        const Built L = build(lo, mid, depth + 1);
        const Built R = build(mid, hi, depth + 1);
        set_node(out->nodes[idx], L, R);
        r.ref = r.ref_fine = idx;
        r.depth = 1 + std::max(L.depth, R.depth);
        return r;
    }
};

// Top tree over already built subtrees (one per mesh), also by binned surface-area heuristic; an item weighs as many
// triangles as its subtree holds.
struct TopItem { Built b; float c[3]; int n; };

Built build_top_sah(FlatScene* out, std::vector<TopItem>& it, int lo, int hi) {
    if (hi - lo == 1) return it[lo].b;
    enum { BINS = 16 };
    float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = lo; k < hi; ++k)
        for (int i = 0; i < 3; ++i) { cmin[i] = fminf(cmin[i], it[k].c[i]); cmax[i] = fmaxf(cmax[i], it[k].c[i]); }
    int best_axis = -1, best_bin = 0; float best_cost = INFINITY;
    for (int ax = 0; ax < 3; ++ax) {
        const float ext = cmax[ax] - cmin[ax];
        if (!(ext > 0.f) || !std::isfinite(ext)) continue;
        const float scale = BINS / ext;
        Box bb[BINS]; int64_t cnt[BINS]; int items[BINS];
        for (int q = 0; q < BINS; ++q) { cnt[q] = 0; items[q] = 0; for (int i = 0; i < 3; ++i) { bb[q].lo[i] = INFINITY; bb[q].hi[i] = -INFINITY; } }
        for (int k = lo; k < hi; ++k) {
            int q = (int)((it[k].c[ax] - cmin[ax]) * scale);
            q = q < 0 ? 0 : (q >= BINS ? BINS - 1 : q);
            cnt[q] += it[k].n; ++items[q]; box_union(bb[q], it[k].b.box);
        }
        for (int q = 0; q < BINS - 1; ++q) {
            Box l, r; int64_t nl = 0, nr = 0; int il = 0, ir = 0;
            for (int i = 0; i < 3; ++i) { l.lo[i] = r.lo[i] = INFINITY; l.hi[i] = r.hi[i] = -INFINITY; }
            for (int z = 0; z <= q; ++z) if (items[z]) { box_union(l, bb[z]); nl += cnt[z]; il += items[z]; }
            for (int z = q + 1; z < BINS; ++z) if (items[z]) { box_union(r, bb[z]); nr += cnt[z]; ir += items[z]; }
            if (!il || !ir) continue;
            const float cost = half_area(l) * (float)nl + half_area(r) * (float)nr;
            if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = q; }
        }
    }
    int mid;
    // An item whose box is a large part of the whole group's (the ground under a field of meshes, a wall of a room) sits
    // somewhere in the middle by its centre, so no split of the centres can set it apart: it would ride down the tree,
    // level after level, in a child whose box it blows up to the size of the group.  It becomes a child of its own.
    int big = lo; Box all = it[lo].b.box;
    for (int k = lo + 1; k < hi; ++k) { box_union(all, it[k].b.box); if (half_area(it[k].b.box) > half_area(it[big].b.box)) big = k; }
    if (hi - lo >= 3 && half_area(it[big].b.box) >= RT_TOP_BIG_ITEM * half_area(all)) {
        std::swap(it[lo], it[big]);
        mid = lo + 1;
    } else if (best_axis >= 0) {
        const float scale = BINS / (cmax[best_axis] - cmin[best_axis]), c0 = cmin[best_axis];
        const int ax = best_axis, bbin = best_bin;
        TopItem* m = std::partition(it.data() + lo, it.data() + hi, [=](const TopItem& x) {
            int q = (int)((x.c[ax] - c0) * scale);
            q = q < 0 ? 0 : (q >= BINS ? BINS - 1 : q);
            return q <= bbin;
        });
        mid = (int)(m - it.data());
    } else mid = (lo + hi) / 2;
    if (mid <= lo || mid >= hi) mid = (lo + hi) / 2;
    const int32_t idx = (int32_t)out->nodes.size();
    out->nodes.emplace_back();
    const Built L = build_top_sah(out, it, lo, mid);
    const Built R = build_top_sah(out, it, mid, hi);
    set_node(out->nodes[idx], L, R);
    Built r; r.ref = r.ref_fine = idx; r.box = L.box; box_union(r.box, R.box);
    r.depth = 1 + std::max(L.depth, R.depth);
    return r;
}

// Expected work of a random ray through the tree (surface-area heuristic): every inner node weighs its box area,
// every leaf its box area times its triangle count.  Used to choose between candidate trees.
double tree_cost(const FlatScene& f) {
    double cost = 0;
    for (const Node64& n : f.nodes) {
        const int32_t ref[2] = {n.lref_fine, n.rref_fine};
        const float* bx[2] = {n.lbox, n.rbox};
        for (int s = 0; s < 2; ++s) {
            if (ref[s] == RT_NONE) continue;
            Box b; memcpy(b.lo, bx[s], 12); memcpy(b.hi, bx[s] + 3, 12);
            const double a = half_area(b);
            if (ref[s] >= 0) cost += a * 1.0;
            else cost += a * 1.2 * f.tris[~ref[s]].run_fine;
        }
    }
    return cost;
}

}  // namespace

bool build_sah_scene(const rt_mesh_desc* meshes, int n_meshes, const FlatScene& exact, FlatScene* out, std::string* err) {
    out->nodes.clear(); out->tris.clear(); out->info.clear(); out->nrm.clear();
    out->n_leaves = 0; out->max_depth = 0;
    SahBuilder b;
    b.meshes = meshes; b.out = out;
    int64_t total = 0;
    for (int mi = 0; mi < n_meshes; ++mi) { b.first_of.push_back((int32_t)total); total += meshes[mi].n_tris; }
    if (total <= 0 || total != (int64_t)exact.tris.size()) { *err = "triangle count mismatch between the two trees"; return false; }
    b.rank_of.assign((size_t)total, 0);
    for (size_t k = 0; k < exact.tris.size(); ++k) b.rank_of[exact.info[k].id] = exact.tris[k].rank;
    b.refs.resize((size_t)total);
    for (int mi = 0; mi < n_meshes; ++mi)
        for (int t = 0; t < meshes[mi].n_tris; ++t) {
            const float* a = meshes[mi].pos + (size_t)t * 9; const float* bb = a + 3; const float* c = a + 6;
            SahRef& r = b.refs[(size_t)b.first_of[mi] + t];
            for (int i = 0; i < 3; ++i) {
                r.b.lo[i] = fminf(a[i], fminf(bb[i], c[i]));
                r.b.hi[i] = fmaxf(a[i], fmaxf(bb[i], c[i]));
                r.c[i] = 0.5f * r.b.lo[i] + 0.5f * r.b.hi[i];
            }
            r.id = b.first_of[mi] + t;
        }
    if (total <= SahBuilder::LEAF) {
        out->nodes.emplace_back();
        const Built leaf = b.build(0, (int)total, 0);
        Node64 entry; memset(&entry, 0, sizeof(entry));
        memcpy(entry.lbox, leaf.box.lo, 12); memcpy(entry.lbox + 3, leaf.box.hi, 12);
        entry.lref = entry.lref_fine = leaf.ref; entry.rref = entry.rref_fine = RT_NONE;
        out->nodes[0] = entry;
        out->max_depth = 1;
        return true;
    }
    int n_nonempty = 0;
    for (int mi = 0; mi < n_meshes; ++mi) n_nonempty += meshes[mi].n_tris > 0;
    const std::vector<SahRef> all = b.refs;
    // candidate 1: one tree over all triangles (not tried for scenes of very many meshes: instanced fields, where the
    // per-mesh trees are the natural clusters and the build time matters)
    const bool try_one = n_nonempty < 2 || n_nonempty <= 64;
    if (try_one) {
        out->nodes.reserve((size_t)total); out->tris.reserve((size_t)total); out->info.reserve((size_t)total); out->nrm.reserve((size_t)total * 9);
        const Built root = b.build(0, (int)total, 0);      // the root is the first node emitted: index 0
        out->max_depth = root.depth;
        if (root.ref != 0) { *err = "internal: tree root is not node 0"; return false; }
        if (n_nonempty < 2) return true;
    }
    // candidate 2: a tree per mesh under a top tree over the meshes (keeps a few huge triangles, such as the walls of
    // a room, from widening the boxes of everything they are mixed with).  The per-mesh builds are independent and run
    // on a few host threads, each into a fragment of its own; the fragments are then spliced with shifted references.
    struct Frag { FlatScene f; Built root; };
    std::vector<Frag> frags((size_t)n_meshes);
    {
        std::atomic<int> next(0);
        std::atomic<bool> failed(false);     // an exception must not leave a worker thread (it would end the process)
        auto work = [&]() {
            try {
                for (;;) {
                    const int mi = next.fetch_add(1);
                    if (mi >= n_meshes || failed.load()) break;
                    const int n = meshes[mi].n_tris;
                    if (n <= 0) continue;
                    SahBuilder w;
                    w.meshes = meshes; w.first_of = b.first_of; w.out = &frags[(size_t)mi].f;
                    w.rank_of.clear();
                    w.rank_ptr = b.rank_of.data();
                    w.refs.assign(all.begin() + b.first_of[mi], all.begin() + b.first_of[mi] + n);
                    frags[(size_t)mi].root = w.build(0, n, 0);
                }
            } catch (...) { failed.store(true); }
        };
        int n_threads = (int)std::max(1u, std::min(16u, std::min((unsigned)n_nonempty, std::thread::hardware_concurrency())));
        std::vector<std::thread> pool;
        try {
            for (int t = 1; t < n_threads; ++t) pool.emplace_back(work);
        } catch (...) { /* fewer threads than hoped for: the ones that started and this one share the work */ }
        work();
        for (auto& th : pool) th.join();
        if (failed.load()) { *err = "out of host memory while building the trees"; return false; }
    }
    FlatScene two;
    two.nodes.reserve((size_t)total); two.tris.reserve((size_t)total); two.info.reserve((size_t)total); two.nrm.reserve((size_t)total * 9);
    two.nodes.emplace_back();                           // node 0 = entry, filled in last
    std::vector<TopItem> items;
    for (int mi = 0; mi < n_meshes; ++mi) {
        if (meshes[mi].n_tris <= 0) continue;
        Frag& fr = frags[(size_t)mi];
        const int32_t node_off = (int32_t)two.nodes.size(), tri_off = (int32_t)two.tris.size();
        auto shift = [&](int32_t ref) { return ref == RT_NONE ? ref : (ref >= 0 ? ref + node_off : ~(~ref + tri_off)); };
        for (Node64 n : fr.f.nodes) {
            n.lref = shift(n.lref); n.rref = shift(n.rref); n.lref_fine = shift(n.lref_fine); n.rref_fine = shift(n.rref_fine);
            two.nodes.push_back(n);
        }
        two.tris.insert(two.tris.end(), fr.f.tris.begin(), fr.f.tris.end());
        two.info.insert(two.info.end(), fr.f.info.begin(), fr.f.info.end());
        two.nrm.insert(two.nrm.end(), fr.f.nrm.begin(), fr.f.nrm.end());
        two.n_leaves += fr.f.n_leaves;
        TopItem ti;
        ti.b = fr.root; ti.b.ref = ti.b.ref_fine = shift(fr.root.ref);
        ti.n = meshes[mi].n_tris;
        for (int i = 0; i < 3; ++i) ti.c[i] = 0.5f * ti.b.box.lo[i] + 0.5f * ti.b.box.hi[i];
        items.push_back(ti);
        fr.f = FlatScene();                                  // release the fragment
    }
    const Built top = build_top_sah(&two, items, 0, (int)items.size());
    two.nodes[0] = two.nodes[top.ref];                  // copy the top node into slot 0 (its own slot stays unused)
    memset(&two.nodes[top.ref], 0, sizeof(Node64));
    two.nodes[top.ref].lref = two.nodes[top.ref].rref = two.nodes[top.ref].lref_fine = two.nodes[top.ref].rref_fine = RT_NONE;
    two.max_depth = top.depth;
    const double c_one = try_one ? tree_cost(*out) : 0.0, c_two = tree_cost(two);
    if (getenv("RT_AMD_DEBUG_TREE")) fprintf(stderr, "rt_amd: expected cost, one tree %.4g, tree per mesh %.4g\n", c_one, c_two);
    if (!try_one || c_two < c_one) *out = std::move(two);
    return true;
}

}  // namespace rt
