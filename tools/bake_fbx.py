"""Bake a binary FBX scene into a .rtscene fixture.

Restates, for the simple Maya-exported scenes the reference ships
(input_files/**/*.fbx), what its loader produces (reference
src/main.cpp:256-395 on top of Assimp 5.0.1's FBX importer):

* one mesh per Model node with a Geometry, polygons fan-triangulated
  (0,1,2),(0,2,3) in face order, corners un-indexed;
* vertices moved to world space by the node transform, normals by its
  inverse transpose and re-normalised (src/main.cpp:316-329);
* material kind from the material NAME substring lambert/phong/refrac
  (src/main.cpp:331-349), colours = <X>Color * <X>Factor;
* directional / point lights with the direction negated so that it points
  toward the light (src/main.cpp:369), camera basis as src/cameras.cpp:3-18.

The reference's own loader cannot be built in this image, so the result is
"what our reader understands the file to say"; it is an input fixture, not a
checked reproduction of Assimp's floats.

usage: python tools/bake_fbx.py in.fbx out.rtscene
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools.fbx_reader import read_fbx  # noqa: E402
from raytracer_amd.scene import (Camera, Light, Material, Mesh, Scene, MAT_DEFAULT, MAT_LAMBERT,  # noqa: E402
                                 MAT_PHONG, MAT_REFRAC, LIGHT_DIRECTIONAL, LIGHT_POINT)

F = np.float32


def props70(node):
    out = {}
    p70 = node.find("Properties70") if node is not None else None
    if p70 is None:
        return out
    for p in p70.find_all("P"):
        out[p.props[0]] = p.props[4:]
    return out


def obj_name(s):
    return s.split("\x00\x01")[0]


def mat4(rows):
    return np.array(rows, dtype=F)


def translate(v):
    m = np.eye(4, dtype=F)
    m[0, 3], m[1, 3], m[2, 3] = F(v[0]), F(v[1]), F(v[2])
    return m


def scale(v):
    m = np.eye(4, dtype=F)
    m[0, 0], m[1, 1], m[2, 2] = F(v[0]), F(v[1]), F(v[2])
    return m


def rot_xyz(deg):
    """FBX eEulerXYZ: R = Rz * Ry * Rx, angles in degrees, float32 like aiMatrix4x4."""
    ax, ay, az = [F(math.radians(float(d))) for d in deg]
    cx, sx = F(math.cos(ax)), F(math.sin(ax))
    cy, sy = F(math.cos(ay)), F(math.sin(ay))
    cz, sz = F(math.cos(az)), F(math.sin(az))
    rx = mat4([[1, 0, 0, 0], [0, cx, -sx, 0], [0, sx, cx, 0], [0, 0, 0, 1]])
    ry = mat4([[cy, 0, sy, 0], [0, 1, 0, 0], [-sy, 0, cy, 0], [0, 0, 0, 1]])
    rz = mat4([[cz, -sz, 0, 0], [sz, cz, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    return (rz @ ry @ rx).astype(F)


def node_transform(p):
    """T * Roff * Rp * Rpre * R * Rpost^-1 * Rp^-1 * Soff * Sp * S * Sp^-1 (FBX SDK node model)."""
    def g(k, d):
        return [float(x) for x in p.get(k, d)]
    z3, o3 = [0, 0, 0], [1, 1, 1]
    T = translate(g("Lcl Translation", z3))
    Roff = translate(g("RotationOffset", z3))
    Rp = translate(g("RotationPivot", z3))
    Rpre = rot_xyz(g("PreRotation", z3))
    R = rot_xyz(g("Lcl Rotation", z3))
    Rpost = rot_xyz(g("PostRotation", z3))
    Soff = translate(g("ScalingOffset", z3))
    Sp = translate(g("ScalingPivot", z3))
    S = scale(g("Lcl Scaling", o3))
    inv = lambda m: np.linalg.inv(m.astype(np.float64)).astype(F)
    m = T
    for x in (Roff, Rp, Rpre, R, inv(Rpost), inv(Rp), Soff, Sp, S, inv(Sp)):
        m = (m @ x).astype(F)
    return m


def xform_point(m, v):
    # aiMatrix4x4 * aiVector3D: a1*x + a2*y + a3*z + a4, float32, left to right
    out = np.empty_like(v)
    for r in range(3):
        out[:, r] = ((m[r, 0] * v[:, 0] + m[r, 1] * v[:, 1]).astype(F) + m[r, 2] * v[:, 2]).astype(F) + m[r, 3]
    return out.astype(F)


def normalize_rows(v):
    l = np.sqrt(((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]).astype(F) + v[:, 2] * v[:, 2]).astype(F)).astype(F)
    return (v / l[:, None]).astype(F)


def layer_values(geom, layer_name, data_name, index_name, poly_idx, vert_idx, width):
    le = geom.find(layer_name)
    if le is None:
        return None
    data = le.find(data_name).props[0].astype(np.float64).reshape(-1, width)
    mapping = le.find("MappingInformationType").props[0]
    ref = le.find("ReferenceInformationType").props[0]
    ncorner = len(vert_idx)
    if mapping == "ByPolygonVertex":
        idx = np.arange(ncorner)
    elif mapping in ("ByVertice", "ByVertex"):
        idx = vert_idx
    elif mapping == "ByPolygon":
        idx = poly_idx
    elif mapping == "AllSame":
        idx = np.zeros(ncorner, dtype=np.int64)
    else:
        raise ValueError("unsupported mapping " + mapping)
    if ref == "IndexToDirect":
        idx = le.find(index_name).props[0][idx]
    return data[idx]


def bake(path):
    root = read_fbx(path)
    objects = root.find("Objects")
    conns = [c.props for c in root.find("Connections").find_all("C") if c.props[0] == "OO"]
    by_id = {o.props[0]: o for o in objects.children}
    templates = {}
    for ot in root.find("Definitions").find_all("ObjectType"):
        pt = ot.find("PropertyTemplate")
        if pt is not None:
            templates[ot.props[0]] = props70(pt)

    def children_of(pid, kind):
        return [by_id[c[1]] for c in conns if c[2] == pid and c[1] in by_id and by_id[c[1]].name == kind]

    def parent_chain(mid):
        chain = [mid]
        while True:
            ps = [c[2] for c in conns if c[1] == chain[-1] and c[2] in by_id and by_id[c[2]].name == "Model"]
            if not ps:
                return chain
            chain.append(ps[0])

    def world(mid):
        m = np.eye(4, dtype=F)
        for nid in reversed(parent_chain(mid)):
            m = (m @ node_transform(props70(by_id[nid]))).astype(F)
        return m

    meshes, materials, lights, camera = [], [], [], None
    mat_index = {}
    models = [by_id[c[1]] for c in conns if c[2] == 0 and c[1] in by_id and by_id[c[1]].name == "Model"]
    # depth-first over the node hierarchy in connection order, like the importer's node walk
    order = []

    def walk(m):
        order.append(m)
        for ch in children_of(m.props[0], "Model"):
            walk(ch)
    for m in models:
        walk(m)

    for model in order:
        mid = model.props[0]
        name = obj_name(model.props[1])
        W = world(mid)
        for geom in children_of(mid, "Geometry"):
            verts = geom.find("Vertices").props[0].astype(F).reshape(-1, 3)
            pvi = geom.find("PolygonVertexIndex").props[0].astype(np.int64)
            ends = pvi < 0
            vert_idx = np.where(ends, ~pvi, pvi)
            poly_idx = np.concatenate([[0], np.cumsum(ends)[:-1]])
            normals = layer_values(geom, "LayerElementNormal", "Normals", "NormalsIndex", poly_idx, vert_idx, 3)
            if normals is None:
                # the reference imports with aiProcess_GenSmoothNormals (src/main.cpp:295-301): a mesh without normals
                # gets one normal per vertex position, averaged over the faces around it
                normals = smooth_normals(verts, vert_idx, ends)
            cpos = xform_point(W, verts[vert_idx])
            Wn = np.linalg.inv(W.astype(np.float64)).T.astype(F)
            Wn[:3, 3] = 0
            cnrm = normalize_rows(xform_point(Wn, normals.astype(F)))
            tris_p, tris_n = [], []
            start = 0
            for e in np.nonzero(ends)[0]:
                n = e - start + 1
                for k in range(1, n - 1):   # fan (0,k,k+1); non-triangles beyond quads are approximated the same way
                    ids = [start, start + k, start + k + 1]
                    tris_p.append(cpos[ids])
                    tris_n.append(cnrm[ids])
                start = e + 1
            mats = children_of(mid, "Material")
            if mats:
                mnode = mats[0]
                if mnode.props[0] not in mat_index:
                    mat_index[mnode.props[0]] = len(materials)
                    materials.append(convert_material(mnode, templates.get("Material", {})))
                mi = mat_index[mnode.props[0]]
            else:
                mi = len(materials)
                materials.append(Material(kind=MAT_DEFAULT))
            meshes.append(Mesh(name, mi, np.array(tris_p, dtype=F), np.array(tris_n, dtype=F)))
        for attr in children_of(mid, "NodeAttribute"):
            kind = attr.props[2]
            p = dict(templates.get("NodeAttribute", {}))
            p.update(props70(attr))
            R = W[:3, :3]
            if kind == "Light":
                lt = int(p.get("LightType", [0])[0])
                col = np.array([float(x) for x in p.get("Color", [1, 1, 1])], dtype=F) * F(float(p.get("Intensity", [100.0])[0]) / 100.0)
                d = (R @ np.array([0, -1, 0], dtype=F)).astype(F)
                d = (d * F(-1.0)).astype(F)                       # src/main.cpp:369
                if lt == 1:
                    lights.append(Light(LIGHT_DIRECTIONAL, normalize_rows(d[None])[0], col))
                else:
                    lights.append(Light(LIGHT_POINT, W[:3, 3].copy(), col))
            elif kind == "Camera" and camera is None:
                look = (R @ np.array([1, 0, 0], dtype=F)).astype(F)
                up = (R @ np.array([0, 1, 0], dtype=F)).astype(F)
                camera = Camera.look(W[:3, 3].copy(), up, look,
                                     float(p.get("NearPlane", [0.1])[0]), float(p.get("FarPlane", [10000.0])[0]))
    if camera is None:
        camera = Camera.look([0, 0, 5], [0, 1, 0], [0, 0, -1])
    return Scene(meshes, materials, lights, camera)


def smooth_normals(verts, vert_idx, ends):
    """Per polygon-vertex normals for a mesh that brings none, as Assimp's GenSmoothNormals step builds them: every
    corner first gets its face's unit normal ((v1 - v0) x (v2 - v0) of the polygon's first three corners), then all
    corners that sit at the same position (within 1e-5 of the mesh's bounding-box diagonal) get the normalised sum of
    their face normals; there is no angle limit (the step's default of 175 degrees only exempts near-opposite faces, which
    are left out of each other's sums here as well).  Zero-area faces contribute nothing."""
    n_corner = len(vert_idx)
    face_n = np.zeros((n_corner, 3), dtype=np.float64)
    start = 0
    for e in np.nonzero(ends)[0]:
        ids = vert_idx[start:e + 1]
        if len(ids) >= 3:
            p0, p1, p2 = verts[ids[0]].astype(np.float64), verts[ids[1]].astype(np.float64), verts[ids[2]].astype(np.float64)
            n = np.cross(p1 - p0, p2 - p0)
            ln = np.linalg.norm(n)
            if ln > 0:
                face_n[start:e + 1] = n / ln
        start = e + 1
    pos = verts[vert_idx].astype(np.float64)
    diag = np.linalg.norm(pos.max(0) - pos.min(0)) if n_corner else 0.0
    eps = max(diag * 1e-5, 1e-30)
    # corners at the same position: snap to a grid of the tolerance and group (neighbouring cells differ by more than
    # the tolerance for all practical meshes; exact duplicates - the common case of an indexed mesh - always group)
    key = np.round(pos / eps).astype(np.int64)
    _, inv = np.unique(key, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    out = np.zeros((n_corner, 3), dtype=np.float64)
    cos_limit = np.cos(np.radians(175.0))
    order = np.argsort(inv, kind="stable")
    bounds = np.nonzero(np.diff(inv[order]))[0] + 1
    for grp in np.split(order, bounds):
        fn = face_n[grp]
        if len(grp) == 1:
            out[grp] = fn
            continue
        ok = (fn @ fn.T) >= cos_limit                    # which neighbours each corner accepts
        out[grp] = ok.astype(np.float64) @ fn
    ln = np.linalg.norm(out, axis=1, keepdims=True)
    out = np.where(ln > 0, out / np.maximum(ln, 1e-300), face_n)
    return out.astype(F)


def convert_material(mnode, template):
    name = obj_name(mnode.props[1])
    p = dict(template)
    p.update(props70(mnode))

    def colour(base, default):
        if base + "Color" not in p:
            return np.array(default, dtype=F)
        c = np.array([float(x) for x in p[base + "Color"]], dtype=F)
        if base + "Factor" in p:
            c = (c * F(float(p[base + "Factor"][0]))).astype(F)
        return c
    m = Material()
    m.base = colour("Diffuse", [0.5, 0.5, 0.5])
    m.ambient = colour("Ambient", [0.05, 0.05, 0.05])
    if "SpecularColor" in p:
        m.spec = np.array([float(x) for x in p["SpecularColor"]], dtype=F)
    if "ShininessExponent" in p:
        m.shininess = float(p["ShininessExponent"][0])
    if "ReflectionFactor" in p:
        m.reflectivity = float(p["ReflectionFactor"][0])
    if "TransparentColor" in p:
        m.transp = colour("Transparent", [1, 1, 1]) if "TransparencyFactor" not in p else \
            (np.array([float(x) for x in p["TransparentColor"]], dtype=F) * F(float(p["TransparencyFactor"][0]))).astype(F)
    if "lambert" in name:
        m.kind = MAT_LAMBERT
    elif "phong" in name:
        m.kind = MAT_PHONG
    elif "refrac" in name:
        m.kind = MAT_REFRAC
    else:
        m.kind = MAT_DEFAULT
    return m


if __name__ == "__main__":
    sc = bake(sys.argv[1])
    sc.save(sys.argv[2])
    print("%s: %d meshes %s, %d tris, %d materials, %d lights" % (
        sys.argv[2], len(sc.meshes), [(m.name, m.n_tris) for m in sc.meshes], sc.n_tris,
        len(sc.materials), len(sc.lights)))
    print(" camera pos", sc.camera.pos, "up", sc.camera.up, "w", sc.camera.w, "side", sc.camera.side,
          "near/far", sc.camera.near, sc.camera.far)
    for l in sc.lights:
        print(" light", l.kind, l.vec, l.color)
    for m in sc.materials:
        print(" material", m.kind, m.base, m.ambient)
