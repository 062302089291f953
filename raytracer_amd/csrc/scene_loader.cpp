// Host-side scene loading and option parsing behind the C-ABI (no device work): what the reference's main() does
// before it builds its tree - load_scene (src/main.cpp:256-395 with retrieve_node_gtrans :240-253 and the material,
// light and camera constructors src/materials.cpp:10-118, src/lights.cpp:7-25, src/cameras.cpp:3-18) and the option
// file (src/main.cpp:646-683, defaults :74-75,103-115).
//
// The reference imports through Assimp 5.0.1, which this build neither ships nor links.  Its scene files are binary FBX
// (Maya exports: one Model node per mesh / light / camera, Geometry with Vertices + PolygonVertexIndex +
// LayerElementNormal, Material, NodeAttribute), and this file reads that format directly: the node records, the
// zlib-deflated arrays, the object connections, the FBX node transform, fan triangulation of polygons in face order,
// smooth normals for meshes that bring none (the import's aiProcess_GenSmoothNormals).  Vertices go to world space by the
// product of the node chain's transforms, normals by its inverse transpose, re-normalised (src/main.cpp:316-329); the
// 4x4 arithmetic is float32 in Assimp's operation order (aiMatrix4x4 operator*, Inverse by cofactors).  Whether every
// float equals what Assimp's importer hands the reference is not verified anywhere (no Assimp here): the triangle counts,
// names, materials, lights and camera of the shipped scenes match what was recorded from the reference's loader, and the
// rendered silhouettes match the reference's own result images.
//
// Also reads the flat ".rtscene" form (raytracer_amd/scene.py), whose records are the C-ABI structs themselves.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rt_amd.h"

namespace {

thread_local std::string g_load_err;
int lfail(int code, const std::string& msg) { g_load_err = msg; return code; }

typedef float F;

// ------------------------------------------------------------------------------------------- binary FBX records
struct Prop {
    char type = 0;                 // Y C I F D L (scalars), f d l i b (arrays), S R (bytes)
    int64_t i = 0; double d = 0;   // scalar value (integers also as d)
    std::string s;                 // S / R
    std::vector<double> ad;        // f, d arrays (floats widened exactly)
    std::vector<int64_t> ai;       // l, i, b arrays
    double num() const { return (type == 'F' || type == 'D') ? d : (double)i; }
};
struct Node {
    std::string name;
    std::vector<Prop> props;
    std::vector<Node> children;
    const Node* find(const char* n) const { for (const Node& c : children) if (c.name == n) return &c; return nullptr; }
};

struct Reader {
    const uint8_t* b; size_t n; bool wide; std::string err;
    bool need(size_t p, size_t k) { if (p + k > n || p + k < p) { err = "truncated FBX file"; return false; } return true; }
    template <class T> bool rd(size_t& p, T* v) { if (!need(p, sizeof(T))) return false; memcpy(v, b + p, sizeof(T)); p += sizeof(T); return true; }

    bool prop(size_t& p, Prop* o) {
        if (!need(p, 1)) return false;
        const char t = (char)b[p++];
        o->type = t;
        switch (t) {
            case 'Y': { int16_t v; if (!rd(p, &v)) return false; o->i = v; o->d = v; return true; }
            case 'C': { uint8_t v; if (!rd(p, &v)) return false; o->i = v != 0; o->d = (double)o->i; return true; }
            case 'I': { int32_t v; if (!rd(p, &v)) return false; o->i = v; o->d = v; return true; }
            case 'F': { float v; if (!rd(p, &v)) return false; o->d = v; o->i = (int64_t)v; return true; }
            case 'D': { double v; if (!rd(p, &v)) return false; o->d = v; o->i = (int64_t)v; return true; }
            case 'L': { int64_t v; if (!rd(p, &v)) return false; o->i = v; o->d = (double)v; return true; }
            case 'S': case 'R': {
                uint32_t len; if (!rd(p, &len) || !need(p, len)) return false;
                o->s.assign((const char*)b + p, len); p += len; return true;
            }
            case 'f': case 'd': case 'l': case 'i': case 'b': {
                uint32_t cnt, enc, clen;
                if (!rd(p, &cnt) || !rd(p, &enc) || !rd(p, &clen) || !need(p, clen)) return false;
                const size_t width = (t == 'f' || t == 'i') ? 4 : (t == 'b' ? 1 : 8);
                std::vector<uint8_t> raw;
                const uint8_t* src = b + p;
                if (enc == 1) {
                    if ((uint64_t)cnt * width > (uint64_t)clen * 1040 + 64) { err = "FBX array claims more than its bytes can inflate to"; return false; }
                    raw.resize((size_t)cnt * width);
                    uLongf out_len = (uLongf)raw.size();
                    if (uncompress(raw.data(), &out_len, b + p, clen) != Z_OK || out_len != raw.size()) { err = "FBX array does not inflate"; return false; }
                    src = raw.data();
                } else if ((size_t)cnt * width > clen) { err = "FBX array shorter than its count"; return false; }
                p += clen;
                if (t == 'f' || t == 'd') {
                    o->ad.resize(cnt);
                    for (uint32_t k = 0; k < cnt; ++k) {
                        if (t == 'f') { float v; memcpy(&v, src + 4 * (size_t)k, 4); o->ad[k] = v; }
                        else { double v; memcpy(&v, src + 8 * (size_t)k, 8); o->ad[k] = v; }
                    }
                } else {
                    o->ai.resize(cnt);
                    for (uint32_t k = 0; k < cnt; ++k) {
                        if (t == 'i') { int32_t v; memcpy(&v, src + 4 * (size_t)k, 4); o->ai[k] = v; }
                        else if (t == 'l') { int64_t v; memcpy(&v, src + 8 * (size_t)k, 8); o->ai[k] = v; }
                        else o->ai[k] = src[k];
                    }
                }
                return true;
            }
            default: err = std::string("unknown FBX property type '") + t + "'"; return false;
        }
    }

    // returns false on error; *sentinel = the record was the all-zero terminator of a child list
    bool node(size_t& p, Node* o, bool* sentinel, int depth) {
        if (depth > 64) { err = "FBX nesting too deep"; return false; }
        uint64_t end, nprops, plen;
        if (wide) { if (!rd(p, &end) || !rd(p, &nprops) || !rd(p, &plen)) return false; }
        else { uint32_t a, c, d; if (!rd(p, &a) || !rd(p, &c) || !rd(p, &d)) return false; end = a; nprops = c; plen = d; }
        uint8_t nlen; if (!rd(p, &nlen)) return false;
        *sentinel = end == 0;
        if (end == 0) return true;
        if (end > n || !need(p, nlen)) { err = "FBX record runs past the file"; return false; }
        o->name.assign((const char*)b + p, nlen); p += nlen;
        const size_t pend = p + plen;
        if (pend > end) { err = "FBX property list runs past its record"; return false; }
        if (nprops > plen) { err = "FBX record claims more properties than bytes"; return false; }     // a property is at least one byte
        o->props.resize(nprops);
        for (uint64_t k = 0; k < nprops; ++k) if (!prop(p, &o->props[k])) return false;
        p = pend;
        while (p < end) {
            Node c; bool s = false;
            if (!node(p, &c, &s, depth + 1)) return false;
            if (s) break;
            o->children.push_back(std::move(c));
        }
        p = end;
        return true;
    }
};

bool read_fbx(const std::vector<uint8_t>& file, Node* root, std::string* err) {
    static const char magic[] = "Kaydara FBX Binary  ";
    if (file.size() < 27 || memcmp(file.data(), magic, 20) != 0 || file[20] != 0) { *err = "not a binary FBX file"; return false; }
    uint32_t version; memcpy(&version, file.data() + 23, 4);
    Reader r{file.data(), file.size(), version >= 7500, ""};
    size_t p = 27;
    while (p < file.size()) {
        Node c; bool s = false;
        if (!r.node(p, &c, &s, 0)) { *err = r.err; return false; }
        if (s) break;
        root->children.push_back(std::move(c));
    }
    return true;
}

// ------------------------------------------------------------------------------------------- float32 4x4 as Assimp's
struct M4 { F m[4][4]; };
M4 ident() { M4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = i == j ? 1.f : 0.f; return r; }
// aiMatrix4x4t::operator*=: every element a sum of four products, left to right
M4 mul(const M4& a, const M4& b) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
M4 translate(const double v[3]) { M4 r = ident(); for (int i = 0; i < 3; ++i) r.m[i][3] = (F)v[i]; return r; }
M4 scale(const double v[3]) { M4 r = ident(); for (int i = 0; i < 3; ++i) r.m[i][i] = (F)v[i]; return r; }
// FBX eEulerXYZ: R = Rz * Ry * Rx, angles in degrees
M4 rot_xyz(const double deg[3]) {
    F c[3], s[3];
    for (int i = 0; i < 3; ++i) { const F a = (F)(deg[i] * (3.14159265358979323846 / 180.0)); c[i] = (F)std::cos((double)a); s[i] = (F)std::sin((double)a); }
    M4 rx = ident(), ry = ident(), rz = ident();
    rx.m[1][1] = c[0]; rx.m[1][2] = -s[0]; rx.m[2][1] = s[0]; rx.m[2][2] = c[0];
    ry.m[0][0] = c[1]; ry.m[0][2] = s[1]; ry.m[2][0] = -s[1]; ry.m[2][2] = c[1];
    rz.m[0][0] = c[2]; rz.m[0][1] = -s[2]; rz.m[1][0] = s[2]; rz.m[1][1] = c[2];
    return mul(mul(rz, ry), rx);
}
// inverse by cofactors over the determinant, float32 (the form of aiMatrix4x4t::Inverse); a singular matrix gives NaNs
M4 inverse(const M4& a) {
    const F (*m)[4] = a.m;
    auto det3 = [&](int r0, int r1, int r2, int c0, int c1, int c2) {
        return m[r0][c0] * (m[r1][c1] * m[r2][c2] - m[r1][c2] * m[r2][c1]) + m[r0][c1] * (m[r1][c2] * m[r2][c0] - m[r1][c0] * m[r2][c2]) +
               m[r0][c2] * (m[r1][c0] * m[r2][c1] - m[r1][c1] * m[r2][c0]);
    };
    F cof[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            int r[3], c[3], ri = 0, ci = 0;
            for (int k = 0; k < 4; ++k) { if (k != i) r[ri++] = k; if (k != j) c[ci++] = k; }
            const F d = det3(r[0], r[1], r[2], c[0], c[1], c[2]);
            cof[i][j] = ((i + j) & 1) ? -d : d;
        }
    const F det = m[0][0] * cof[0][0] + m[0][1] * cof[0][1] + m[0][2] * cof[0][2] + m[0][3] * cof[0][3];
    const F inv = 1.f / det;
    M4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = inv * cof[j][i];
    return r;
}
// aiMatrix4x4 * aiVector3D: a1*x + a2*y + a3*z + a4 per row
void xform(const M4& a, const F v[3], F out[3]) { for (int r = 0; r < 3; ++r) out[r] = a.m[r][0] * v[0] + a.m[r][1] * v[1] + a.m[r][2] * v[2] + a.m[r][3]; }
void normalize3(F v[3]) { const F l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; }

// ------------------------------------------------------------------------------------------- FBX objects
typedef std::map<std::string, std::vector<Prop>> P70;       // property name -> its values (the record's props from the 5th on)

P70 props70(const Node* n) {
    P70 out;
    const Node* p = n ? n->find("Properties70") : nullptr;
    if (!p) return out;
    for (const Node& c : p->children)
        if (c.name == "P" && c.props.size() >= 4) out[c.props[0].s] = std::vector<Prop>(c.props.begin() + 4, c.props.end());
    return out;
}
void get3(const P70& p, const char* key, const double dflt[3], double out[3]) {
    auto it = p.find(key);
    for (int i = 0; i < 3; ++i) out[i] = (it != p.end() && it->second.size() > (size_t)i) ? it->second[i].num() : dflt[i];
}
double get1(const P70& p, const char* key, double dflt) { auto it = p.find(key); return (it != p.end() && !it->second.empty()) ? it->second[0].num() : dflt; }
bool has(const P70& p, const char* key) { return p.find(key) != p.end(); }

// T * Roff * Rp * Rpre * R * Rpost^-1 * Rp^-1 * Soff * Sp * S * Sp^-1 (the FBX node model)
M4 node_transform(const P70& p) {
    static const double z3[3] = {0, 0, 0}, o3[3] = {1, 1, 1};
    double v[3];
    get3(p, "Lcl Translation", z3, v); M4 m = translate(v);
    get3(p, "RotationOffset", z3, v); m = mul(m, translate(v));
    get3(p, "RotationPivot", z3, v); const M4 rp = translate(v); m = mul(m, rp);
    get3(p, "PreRotation", z3, v); m = mul(m, rot_xyz(v));
    get3(p, "Lcl Rotation", z3, v); m = mul(m, rot_xyz(v));
    get3(p, "PostRotation", z3, v); m = mul(m, inverse(rot_xyz(v)));
    m = mul(m, inverse(rp));
    get3(p, "ScalingOffset", z3, v); m = mul(m, translate(v));
    get3(p, "ScalingPivot", z3, v); const M4 sp = translate(v); m = mul(m, sp);
    get3(p, "Lcl Scaling", o3, v); m = mul(m, scale(v));
    return mul(m, inverse(sp));
}

std::string object_name(const std::string& s) { const size_t k = s.find(std::string("\x00\x01", 2)); return k == std::string::npos ? s : s.substr(0, k); }

struct HostMesh { std::string name; int material; std::vector<F> pos, nrm; };

}  // namespace

struct rt_host_scene {
    std::vector<HostMesh> meshes;
    std::vector<rt_mesh_desc> mesh_desc;
    std::vector<rt_material_desc> materials;
    std::vector<rt_light_desc> lights;
    rt_camera_desc camera;
};

namespace {

void set3(float* d, F a, F b, F c) { d[0] = a; d[1] = b; d[2] = c; }

// src/cameras.cpp:3-18: up normalised, lookat negated and normalised, side = up x lookat normalised
void camera_look(rt_camera_desc* cam, const F pos[3], const F up_in[3], const F look_in[3], F near_clip, F far_clip) {
    F up[3] = {up_in[0], up_in[1], up_in[2]}, w[3] = {look_in[0] * -1.0f, look_in[1] * -1.0f, look_in[2] * -1.0f};
    auto norm = [](F v[3]) {                              // src/linmath.h:42-46: k = 1/sqrt(v.v); v *= k
        F p = 0.f; for (int i = 0; i < 3; ++i) p += v[i] * v[i];
        const F k = 1.f / (F)std::sqrt(p);
        for (int i = 0; i < 3; ++i) v[i] = v[i] * k;
    };
    norm(up); norm(w);
    F side[3] = {up[1] * w[2] - up[2] * w[1], up[2] * w[0] - up[0] * w[2], up[0] * w[1] - up[1] * w[0]};
    norm(side);
    memset(cam, 0, sizeof(*cam));
    memcpy(cam->pos, pos, 12); memcpy(cam->up, up, 12); memcpy(cam->w, w, 12); memcpy(cam->side, side, 12);
    cam->near_clip = near_clip; cam->far_clip = far_clip;
}

rt_material_desc default_material(int kind) {            // the constructors' defaults, src/materials.cpp:10-118
    rt_material_desc m; memset(&m, 0, sizeof(m));
    m.kind = kind;
    set3(m.base, 0.5f, 0.5f, 0.5f); set3(m.ambient, 0.05f, 0.05f, 0.05f); set3(m.spec, 0.5f, 0.5f, 0.5f); set3(m.transp, 1.f, 1.f, 1.f);
    m.shininess = 32.f; m.reflectivity = 0.5f; m.refractivity = 1.333f;
    return m;
}

// material kind by NAME substring (src/main.cpp:331-349); colours = <X>Color * <X>Factor as the FBX importer combines them
rt_material_desc convert_material(const Node& mnode, const P70& templ) {
    const std::string name = mnode.props.size() > 1 ? object_name(mnode.props[1].s) : "";
    P70 p = templ;
    for (auto& kv : props70(&mnode)) p[kv.first] = kv.second;
    int kind = RT_MAT_DEFAULT;
    if (name.find("lambert") != std::string::npos) kind = RT_MAT_LAMBERT;
    else if (name.find("phong") != std::string::npos) kind = RT_MAT_PHONG;
    else if (name.find("refrac") != std::string::npos) kind = RT_MAT_REFRAC;
    rt_material_desc m = default_material(kind);
    auto colour = [&](const char* base, float* out) {
        const std::string ck = std::string(base) + "Color", fk = std::string(base) + "Factor";
        if (!has(p, ck.c_str())) return;
        static const double z3[3] = {0, 0, 0};
        double c[3]; get3(p, ck.c_str(), z3, c);
        const bool f = has(p, fk.c_str());
        const F k = (F)get1(p, fk.c_str(), 1.0);
        for (int i = 0; i < 3; ++i) out[i] = f ? (F)c[i] * k : (F)c[i];
    };
    colour("Diffuse", m.base); colour("Ambient", m.ambient);
    if (has(p, "SpecularColor")) { static const double z3[3] = {0, 0, 0}; double c[3]; get3(p, "SpecularColor", z3, c); for (int i = 0; i < 3; ++i) m.spec[i] = (F)c[i]; }
    if (has(p, "ShininessExponent")) m.shininess = (F)get1(p, "ShininessExponent", 32.0);
    if (has(p, "ReflectionFactor")) m.reflectivity = (F)get1(p, "ReflectionFactor", 0.5);
    if (has(p, "TransparencyFactor") && has(p, "TransparentColor")) {
        static const double z3[3] = {0, 0, 0};
        double c[3]; get3(p, "TransparentColor", z3, c);
        const F k = (F)get1(p, "TransparencyFactor", 1.0);
        for (int i = 0; i < 3; ++i) m.transp[i] = (F)c[i] * k;
    } else colour("Transparent", m.transp);
    return m;
}

// Values of a layer element (normals) per polygon corner.
bool layer_values(const Node& geom, const std::vector<int64_t>& poly_idx, const std::vector<int64_t>& vert_idx, std::vector<F>* out, std::string* err) {
    const Node* le = geom.find("LayerElementNormal");
    if (!le) return false;
    const Node *data = le->find("Normals"), *mi = le->find("MappingInformationType"), *ri = le->find("ReferenceInformationType");
    if (!data || !mi || !ri || data->props.empty() || mi->props.empty() || ri->props.empty()) { *err = "incomplete LayerElementNormal"; return false; }
    const std::vector<double>& v = data->props[0].ad;
    const std::string mapping = mi->props[0].s, ref = ri->props[0].s;
    const Node* idxn = le->find("NormalsIndex");
    const size_t nc = vert_idx.size();
    out->resize(nc * 3);
    for (size_t c = 0; c < nc; ++c) {
        int64_t k;
        if (mapping == "ByPolygonVertex") k = (int64_t)c;
        else if (mapping == "ByVertice" || mapping == "ByVertex") k = vert_idx[c];
        else if (mapping == "ByPolygon") k = poly_idx[c];
        else if (mapping == "AllSame") k = 0;
        else { *err = "unsupported normal mapping " + mapping; return false; }
        if (ref == "IndexToDirect") {
            if (!idxn || idxn->props.empty() || k < 0 || (size_t)k >= idxn->props[0].ai.size()) { *err = "normal index out of range"; return false; }
            k = idxn->props[0].ai[(size_t)k];
        }
        if (k < 0 || (size_t)k * 3 + 2 >= v.size()) { *err = "normal index out of range"; return false; }
        for (int i = 0; i < 3; ++i) (*out)[c * 3 + i] = (F)v[(size_t)k * 3 + i];
    }
    return true;
}

// Per-corner normals for a mesh that brings none, the way the import step GenSmoothNormals builds them: the unit normal of
// each corner's face ((v1 - v0) x (v2 - v0) of the polygon's first three corners), then for all corners at one position
// (within 1e-5 of the bounding-box diagonal) the normalised sum of their faces' normals, leaving out faces at more than
// 175 degrees to the corner's own.  Zero-area faces contribute nothing.
void smooth_normals(const std::vector<F>& verts, const std::vector<int64_t>& vert_idx, const std::vector<char>& ends, std::vector<F>* out) {
    const size_t nc = vert_idx.size();
    std::vector<double> face(nc * 3, 0.0), pos(nc * 3);
    for (size_t c = 0; c < nc; ++c) for (int i = 0; i < 3; ++i) pos[c * 3 + i] = verts[(size_t)vert_idx[c] * 3 + i];
    size_t start = 0;
    for (size_t e = 0; e < nc; ++e) {
        if (!ends[e]) continue;
        if (e - start + 1 >= 3) {
            const double *p0 = &pos[start * 3], *p1 = &pos[(start + 1) * 3], *p2 = &pos[(start + 2) * 3];
            const double a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
            double n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
            const double ln = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (ln > 0) for (size_t c = start; c <= e; ++c) for (int i = 0; i < 3; ++i) face[c * 3 + i] = n[i] / ln;
        }
        start = e + 1;
    }
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t c = 0; c < nc; ++c) for (int i = 0; i < 3; ++i) { lo[i] = std::min(lo[i], pos[c * 3 + i]); hi[i] = std::max(hi[i], pos[c * 3 + i]); }
    const double diag = nc ? std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2])) : 0.0;
    const double eps = std::max(diag * 1e-5, 1e-30);
    // corners at one position: cells of a grid of the tolerance (exact duplicates - an indexed mesh - always share a cell)
    struct Key { int64_t k[3]; bool operator<(const Key& o) const { return std::lexicographical_compare(k, k + 3, o.k, o.k + 3); } };
    std::map<Key, std::vector<size_t>> groups;
    for (size_t c = 0; c < nc; ++c) { Key k; for (int i = 0; i < 3; ++i) k.k[i] = (int64_t)std::nearbyint(pos[c * 3 + i] / eps); groups[k].push_back(c); }
    const double cos_limit = std::cos(175.0 * 3.14159265358979323846 / 180.0);
    out->assign(nc * 3, 0.f);
    for (auto& g : groups)
        for (size_t a : g.second) {
            double s[3] = {0, 0, 0};
            for (size_t b : g.second) {
                const double dot = face[a * 3] * face[b * 3] + face[a * 3 + 1] * face[b * 3 + 1] + face[a * 3 + 2] * face[b * 3 + 2];
                if (g.second.size() == 1 || dot >= cos_limit) for (int i = 0; i < 3; ++i) s[i] += face[b * 3 + i];
            }
            const double ln = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
            for (int i = 0; i < 3; ++i) (*out)[a * 3 + i] = (F)(ln > 0 ? s[i] / ln : face[a * 3 + i]);
        }
}

int load_fbx(const std::vector<uint8_t>& file, rt_host_scene* sc) {
    Node root;
    std::string err;
    if (!read_fbx(file, &root, &err)) return lfail(RT_ERR_INVALID, err);
    const Node *objects = root.find("Objects"), *conn = root.find("Connections"), *defs = root.find("Definitions");
    if (!objects || !conn) return lfail(RT_ERR_INVALID, "FBX file without Objects / Connections");
    std::unordered_map<int64_t, const Node*> by_id;
    for (const Node& o : objects->children) if (!o.props.empty()) by_id[o.props[0].i] = &o;
    struct C { int64_t child, parent; };
    std::vector<C> conns;
    for (const Node& c : conn->children)
        if (c.name == "C" && c.props.size() >= 3 && c.props[0].s == "OO") conns.push_back({c.props[1].i, c.props[2].i});
    std::map<std::string, P70> templates;
    if (defs)
        for (const Node& ot : defs->children)
            if (ot.name == "ObjectType" && !ot.props.empty())
                if (const Node* pt = ot.find("PropertyTemplate")) templates[ot.props[0].s] = props70(pt);
    auto kind_of = [&](int64_t id) -> const Node* { auto it = by_id.find(id); return it == by_id.end() ? nullptr : it->second; };
    auto children_of = [&](int64_t pid, const char* kind) {
        std::vector<const Node*> out;
        for (const C& c : conns) if (c.parent == pid) { const Node* n = kind_of(c.child); if (n && n->name == kind) out.push_back(n); }
        return out;
    };
    // product of the node transforms from the root down to the node (retrieve_node_gtrans, src/main.cpp:240-253)
    auto world = [&](int64_t mid) {
        std::vector<int64_t> chain{mid};
        for (int guard = 0; guard < 256; ++guard) {
            int64_t up = 0; bool found = false;
            for (const C& c : conns) if (c.child == chain.back()) { const Node* n = kind_of(c.parent); if (n && n->name == "Model") { up = c.parent; found = true; break; } }
            if (!found) break;
            chain.push_back(up);
        }
        M4 m = ident();
        for (size_t k = chain.size(); k-- > 0;) m = mul(m, node_transform(props70(by_id[chain[k]])));
        return m;
    };
    // depth-first over the node hierarchy in connection order, like the importer's node walk
    std::vector<const Node*> order;
    std::vector<const Node*> stack = children_of(0, "Model");
    std::reverse(stack.begin(), stack.end());
    while (!stack.empty()) {
        const Node* m = stack.back(); stack.pop_back();
        order.push_back(m);
        std::vector<const Node*> ch = children_of(m->props[0].i, "Model");
        for (size_t k = ch.size(); k-- > 0;) stack.push_back(ch[k]);
        if (order.size() > 1000000) return lfail(RT_ERR_INVALID, "FBX node hierarchy does not end");
    }
    std::map<int64_t, int> mat_index;
    bool have_camera = false;
    for (const Node* model : order) {
        const int64_t mid = model->props[0].i;
        const std::string name = model->props.size() > 1 ? object_name(model->props[1].s) : "";
        const M4 W = world(mid);
        for (const Node* geom : children_of(mid, "Geometry")) {
            const Node *vn = geom->find("Vertices"), *pn = geom->find("PolygonVertexIndex");
            if (!vn || !pn || vn->props.empty() || pn->props.empty()) continue;
            std::vector<F> verts(vn->props[0].ad.size());
            for (size_t k = 0; k < verts.size(); ++k) verts[k] = (F)vn->props[0].ad[k];
            const std::vector<int64_t>& pvi = pn->props[0].ai;
            const size_t nc = pvi.size();
            std::vector<int64_t> vert_idx(nc), poly_idx(nc);
            std::vector<char> ends(nc);
            int64_t poly = 0;
            for (size_t c = 0; c < nc; ++c) {
                ends[c] = pvi[c] < 0; vert_idx[c] = pvi[c] < 0 ? ~pvi[c] : pvi[c]; poly_idx[c] = poly; poly += ends[c];
                if (vert_idx[c] < 0 || (size_t)vert_idx[c] * 3 + 2 >= verts.size()) return lfail(RT_ERR_INVALID, "FBX vertex index out of range in " + name);
            }
            std::vector<F> normals;
            if (!layer_values(*geom, poly_idx, vert_idx, &normals, &err)) {
                if (!err.empty()) return lfail(RT_ERR_INVALID, err + " in " + name);
                smooth_normals(verts, vert_idx, ends, &normals);
            }
            M4 Wn = inverse(W);
            { M4 t; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t.m[i][j] = Wn.m[j][i]; Wn = t; }
            for (int i = 0; i < 3; ++i) Wn.m[i][3] = 0.f;
            std::vector<F> cpos(nc * 3), cnrm(nc * 3);
            for (size_t c = 0; c < nc; ++c) {
                xform(W, &verts[(size_t)vert_idx[c] * 3], &cpos[c * 3]);
                xform(Wn, &normals[c * 3], &cnrm[c * 3]);
                normalize3(&cnrm[c * 3]);
            }
            HostMesh hm; hm.name = name.substr(0, 31);
            size_t start = 0;
            for (size_t e = 0; e < nc; ++e) {
                if (!ends[e]) continue;
                const size_t n = e - start + 1;
                for (size_t k = 1; k + 1 < n; ++k) {                  // fan (0, k, k+1) in face order
                    const size_t ids[3] = {start, start + k, start + k + 1};
                    for (size_t q : ids) { hm.pos.insert(hm.pos.end(), &cpos[q * 3], &cpos[q * 3] + 3); hm.nrm.insert(hm.nrm.end(), &cnrm[q * 3], &cnrm[q * 3] + 3); }
                }
                start = e + 1;
            }
            const std::vector<const Node*> mats = children_of(mid, "Material");
            if (!mats.empty()) {
                const int64_t id = mats[0]->props[0].i;
                if (!mat_index.count(id)) { mat_index[id] = (int)sc->materials.size(); sc->materials.push_back(convert_material(*mats[0], templates["Material"])); }
                hm.material = mat_index[id];
            } else { hm.material = (int)sc->materials.size(); sc->materials.push_back(default_material(RT_MAT_DEFAULT)); }
            sc->meshes.push_back(std::move(hm));
        }
        for (const Node* attr : children_of(mid, "NodeAttribute")) {
            const std::string kind = attr->props.size() > 2 ? attr->props[2].s : "";
            P70 p = templates["NodeAttribute"];
            for (auto& kv : props70(attr)) p[kv.first] = kv.second;
            auto rot = [&](F x, F y, F z, F out[3]) { for (int r = 0; r < 3; ++r) out[r] = W.m[r][0] * x + W.m[r][1] * y + W.m[r][2] * z; };
            const F wpos[3] = {W.m[0][3], W.m[1][3], W.m[2][3]};
            if (kind == "Light") {
                static const double one3[3] = {1, 1, 1};
                double col[3]; get3(p, "Color", one3, col);
                const F k = (F)(get1(p, "Intensity", 100.0) / 100.0);
                rt_light_desc L; memset(&L, 0, sizeof(L));
                for (int i = 0; i < 3; ++i) L.color[i] = (F)col[i] * k;
                if ((int)get1(p, "LightType", 0.0) == 1) {
                    F d[3]; rot(0.f, -1.f, 0.f, d);
                    for (int i = 0; i < 3; ++i) d[i] = d[i] * -1.0f;         // toward the light, src/main.cpp:369
                    normalize3(d);                                            // src/lights.cpp:9
                    L.kind = RT_LIGHT_DIRECTIONAL; memcpy(L.vec, d, 12);
                } else { L.kind = RT_LIGHT_POINT; memcpy(L.vec, wpos, 12); }
                sc->lights.push_back(L);
            } else if (kind == "Camera" && !have_camera) {
                F look[3], up[3]; rot(1.f, 0.f, 0.f, look); rot(0.f, 1.f, 0.f, up);
                camera_look(&sc->camera, wpos, up, look, (F)get1(p, "NearPlane", 0.1), (F)get1(p, "FarPlane", 10000.0));
                have_camera = true;
            }
        }
    }
    if (!have_camera) { const F pos[3] = {0, 0, 5}, up[3] = {0, 1, 0}, look[3] = {0, 0, -1}; camera_look(&sc->camera, pos, up, look, 0.1f, 10000.f); }
    if (sc->meshes.empty()) return lfail(RT_ERR_INVALID, "FBX file without a mesh");
    return RT_OK;
}

int load_rtscene(const std::vector<uint8_t>& b, rt_host_scene* sc) {
    if (b.size() < 20 + 64 || memcmp(b.data(), "RTSCN01", 8) != 0) return lfail(RT_ERR_INVALID, "not an .rtscene file");
    uint32_t cnt[3]; memcpy(cnt, b.data() + 8, 12);
    size_t p = 20;
    auto need = [&](size_t k) { return p + k <= b.size() && p + k >= p; };
    memcpy(&sc->camera, b.data() + p, 64); p += 64;
    if (!need((size_t)cnt[1] * 64 + (size_t)cnt[2] * 32)) return lfail(RT_ERR_INVALID, "short .rtscene file");
    sc->materials.resize(cnt[1]); memcpy(sc->materials.data(), b.data() + p, (size_t)cnt[1] * 64); p += (size_t)cnt[1] * 64;
    sc->lights.resize(cnt[2]); if (cnt[2]) memcpy(sc->lights.data(), b.data() + p, (size_t)cnt[2] * 32); p += (size_t)cnt[2] * 32;
    for (uint32_t i = 0; i < cnt[0]; ++i) {
        if (!need(40)) return lfail(RT_ERR_INVALID, "short .rtscene file");
        HostMesh hm; hm.name.assign((const char*)b.data() + p, strnlen((const char*)b.data() + p, 31));
        int32_t mat, nt; memcpy(&mat, b.data() + p + 32, 4); memcpy(&nt, b.data() + p + 36, 4); p += 40;
        if (nt < 0 || !need((size_t)nt * 72)) return lfail(RT_ERR_INVALID, "short .rtscene file");
        hm.material = mat;
        hm.pos.resize((size_t)nt * 9); hm.nrm.resize((size_t)nt * 9);
        memcpy(hm.pos.data(), b.data() + p, (size_t)nt * 36); p += (size_t)nt * 36;
        memcpy(hm.nrm.data(), b.data() + p, (size_t)nt * 36); p += (size_t)nt * 36;
        sc->meshes.push_back(std::move(hm));
    }
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_host_last_error(void) { return g_load_err.c_str(); }

int rt_host_scene_load(const char* path, rt_host_scene** out) {
    if (!path || !out) return lfail(RT_ERR_INVALID, "rt_host_scene_load: null argument");
    try {
        FILE* f = fopen(path, "rb");
        if (!f) return lfail(RT_ERR_INVALID, std::string("cannot open ") + path);
        std::vector<uint8_t> b;
        uint8_t chunk[1 << 16];
        size_t n;
        while ((n = fread(chunk, 1, sizeof(chunk), f)) > 0) b.insert(b.end(), chunk, chunk + n);
        fclose(f);
        rt_host_scene* sc = new rt_host_scene();
        memset(&sc->camera, 0, sizeof(sc->camera));
        const int rc = (b.size() >= 8 && memcmp(b.data(), "RTSCN01", 8) == 0) ? load_rtscene(b, sc) : load_fbx(b, sc);
        if (rc != RT_OK) { delete sc; return rc; }
        sc->mesh_desc.resize(sc->meshes.size());
        for (size_t i = 0; i < sc->meshes.size(); ++i) {
            rt_mesh_desc& d = sc->mesh_desc[i];
            memset(&d, 0, sizeof(d));
            strncpy(d.name, sc->meshes[i].name.c_str(), 31);
            d.material = sc->meshes[i].material; d.n_tris = (int32_t)(sc->meshes[i].pos.size() / 9);
            d.pos = sc->meshes[i].pos.data(); d.nrm = sc->meshes[i].nrm.data();
        }
        *out = sc;
        return RT_OK;
    } catch (const std::exception& e) { return lfail(RT_ERR_INVALID, std::string("rt_host_scene_load: ") + e.what());
    } catch (...) { return lfail(RT_ERR_INVALID, "rt_host_scene_load: unknown exception"); }
}

int rt_host_scene_desc(const rt_host_scene* sc, const rt_mesh_desc** meshes, int* n_meshes, const rt_material_desc** materials, int* n_materials,
                       const rt_light_desc** lights, int* n_lights, const rt_camera_desc** camera) {
    if (!sc) return lfail(RT_ERR_INVALID, "rt_host_scene_desc: null scene");
    if (meshes) *meshes = sc->mesh_desc.data();
    if (n_meshes) *n_meshes = (int)sc->mesh_desc.size();
    if (materials) *materials = sc->materials.data();
    if (n_materials) *n_materials = (int)sc->materials.size();
    if (lights) *lights = sc->lights.empty() ? nullptr : sc->lights.data();
    if (n_lights) *n_lights = (int)sc->lights.size();
    if (camera) *camera = &sc->camera;
    return RT_OK;
}

int rt_host_scene_free(rt_host_scene* sc) { delete sc; return RT_OK; }

// src/main.cpp:74-75,103-115: 960x540, hfov 54.43, eps 1e-4, ray eps 0.1, -spp 2, -spr 4, 3 bounces, 3 refractive bounces
int rt_options_default(rt_options* o) {
    if (!o) return lfail(RT_ERR_INVALID, "rt_options_default: null argument");
    memset(o, 0, sizeof(*o));
    o->width = 960; o->height = 540; o->hfov = 54.43f; o->spp = 2; o->spr = 4; o->max_bounce = 3; o->max_refrac_bounce = 3;
    o->eps = 0.0001f; o->ray_eps = 0.1f; o->jitter_mode = RT_JITTER_HASH; o->seed = 1;
    return RT_OK;
}

// One line of the option file, "-key value" (src/main.cpp:618-683); unknown keys and lines without a value are ignored
// (a blank line crashes the reference, src/main.cpp:654; here it is skipped).  Returns 1 if the line set something.
int rt_options_parse_line(rt_options* o, const char* line, int* display_block) {
    if (!o || !line) return 0;
    char key[64] = {0}, val[64] = {0};
    if (sscanf(line, " %63s %63s", key, val) != 2) return 0;
    char* end = nullptr;
    const double v = strtod(val, &end);
    if (end == val) return 0;
    if (!strcmp(key, "-dispw")) o->width = (int)v;
    else if (!strcmp(key, "-disph")) o->height = (int)v;
    else if (!strcmp(key, "-hfov")) o->hfov = (float)v;
    else if (!strcmp(key, "-spp")) o->spp = (int)v;
    else if (!strcmp(key, "-bounce")) o->max_bounce = (int)v;
    else if (!strcmp(key, "-spr")) o->spr = (int)v;
    else if (!strcmp(key, "-eps")) o->eps = (float)v;
    else if (!strcmp(key, "-reps")) o->ray_eps = (float)v;
    else if (!strcmp(key, "-rbounce")) o->max_refrac_bounce = (int)v;
    else if (!strcmp(key, "-mpbs")) { if (display_block) *display_block = (int)v; }
    else return 0;
    return 1;
}

}  // extern "C"
