"""Procedural scenes for the benchmark configurations whose assets the reference does not ship.

The reference's README shows results for a mirrored room of spheres, glass balls under a point light and
a field of bunnies (reference README.md:58-122) but only the Lambert bunny scene is in its input_files/.
These builders author scenes of the same character; they are inputs, not reproductions.
"""
from __future__ import annotations

import os

import numpy as np

from .scene import (Camera, Light, Material, Mesh, Scene, LIGHT_DIRECTIONAL, LIGHT_POINT, MAT_LAMBERT, MAT_PHONG,
                    MAT_REFRAC)

F = np.float32
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def uv_sphere(center, radius, nu=20, nv=20):
    """Triangles and smooth vertex normals of a UV sphere (nu slices, nv stacks): 2*nu*(nv-1) triangles."""
    c = np.asarray(center, dtype=np.float64)
    tris, nrms = [], []

    def pt(i, j):
        th = np.pi * j / nv
        ph = 2 * np.pi * (i % nu) / nu
        n = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
        return c + radius * n, n
    for j in range(nv):
        for i in range(nu):
            p00, n00 = pt(i, j)
            p10, n10 = pt(i + 1, j)
            p01, n01 = pt(i, j + 1)
            p11, n11 = pt(i + 1, j + 1)
            if j > 0:
                tris.append([p00, p10, p11]); nrms.append([n00, n10, n11])
            if j < nv - 1:
                tris.append([p00, p11, p01]); nrms.append([n00, n11, n01])
    return np.array(tris, dtype=F), np.array(nrms, dtype=F)


def quad(p0, p1, p2, p3, normal, n=1):
    """A planar quad p0..p3 (counter-clockwise) split into n x n cells of two triangles."""
    p0, p1, p2, p3 = [np.asarray(p, dtype=np.float64) for p in (p0, p1, p2, p3)]
    nn = np.asarray(normal, dtype=np.float64)
    tris = []
    for a in range(n):
        for b in range(n):
            def at(u, v):
                return (p0 * (1 - u) * (1 - v) + p1 * u * (1 - v) + p2 * u * v + p3 * (1 - u) * v)
            u0, u1, v0, v1 = a / n, (a + 1) / n, b / n, (b + 1) / n
            q0, q1, q2, q3 = at(u0, v0), at(u1, v0), at(u1, v1), at(u0, v1)
            tris.append([q0, q1, q2]); tris.append([q0, q2, q3])
    tris = np.array(tris, dtype=F)
    nrms = np.broadcast_to(nn.astype(F), tris.shape).copy()
    return tris, nrms


def bunny(phong_surroundings=False, second_light=False) -> Scene:
    """The reference's input_files/defaultScene/stanford_bunny.fbx as baked by tools/bake_fbx.py.

    phong_surroundings: ground and dome become Phong with the reference's default parameters
    (src/materials.cpp:45-66), the rabbit stays Lambert (the README's second result, README.md:42-56)."""
    sc = Scene.load(os.path.join(GOLDEN, "stanford_bunny.rtscene"))
    if phong_surroundings:
        base = sc.materials[0]
        sc.materials.append(Material(kind=MAT_PHONG, base=base.base.copy(), ambient=base.ambient.copy()))
        for m in sc.meshes:
            if m.name != "rabbit":
                m.material = 1
    if second_light:
        d = np.array([0.6, 0.7, -0.2], dtype=F)
        d = (d / np.sqrt((d * d).sum())).astype(F)
        sc.lights.append(Light(LIGHT_DIRECTIONAL, d, np.array([0.6, 0.6, 0.6], dtype=F)))
    return sc


def mirror_room(n_spheres=12, seed=3, sphere_res=20) -> Scene:
    """Six Phong walls around Lambert and Phong spheres, one point light (README.md:58-72)."""
    rng = np.random.RandomState(seed)
    S = 4.0
    meshes, mats = [], []
    mats.append(Material(kind=MAT_PHONG, base=np.array([0.3, 0.3, 0.35], F), ambient=np.array([0.02, 0.02, 0.02], F),
                         reflectivity=0.6))
    walls = [
        ("floor", [-S, 0, S], [S, 0, S], [S, 0, -S], [-S, 0, -S], [0, 1, 0]),
        ("ceil", [-S, 2 * S, -S], [S, 2 * S, -S], [S, 2 * S, S], [-S, 2 * S, S], [0, -1, 0]),
        ("back", [-S, 0, -S], [S, 0, -S], [S, 2 * S, -S], [-S, 2 * S, -S], [0, 0, 1]),
        ("front", [S, 0, S], [-S, 0, S], [-S, 2 * S, S], [S, 2 * S, S], [0, 0, -1]),
        ("left", [-S, 0, S], [-S, 0, -S], [-S, 2 * S, -S], [-S, 2 * S, S], [1, 0, 0]),
        ("right", [S, 0, -S], [S, 0, S], [S, 2 * S, S], [S, 2 * S, -S], [-1, 0, 0]),
    ]
    for name, a, b, c, d, n in walls:
        p, nn = quad(a, b, c, d, n, n=4)
        meshes.append(Mesh(name, 0, p, nn))
    for k in range(n_spheres):
        r = 0.35 + 0.35 * rng.rand()
        c = [(-S + 1) + (2 * S - 2) * rng.rand(), r + 0.01 + 2.5 * rng.rand(), (-S + 1) + (S - 0.5) * rng.rand()]
        col = (0.2 + 0.7 * rng.rand(3)).astype(F)
        kind = MAT_PHONG if k % 2 == 0 else MAT_LAMBERT
        mats.append(Material(kind=kind, base=col, ambient=np.array([0.03, 0.03, 0.03], F), reflectivity=0.5))
        p, nn = uv_sphere(c, r, sphere_res, sphere_res)
        meshes.append(Mesh("sphere%03d" % k, len(mats) - 1, p, nn))
    lights = [Light(LIGHT_POINT, np.array([0.5, 2 * S - 1.0, 1.0], F), np.array([1, 1, 1], F))]
    cam = Camera.look([0, 2.2, S - 0.3], [0, 1, 0], [0, -0.15, -1])
    return Scene(meshes, mats, lights, cam)


def glass_balls(sphere_res=24) -> Scene:
    """Three refractive spheres on a Lambert plane under a point light (README.md:90-105)."""
    mats = [Material(kind=MAT_LAMBERT, base=np.array([0.45, 0.45, 0.45], F), ambient=np.array([0.05, 0.05, 0.05], F)),
            Material(kind=MAT_REFRAC, base=np.array([0.1, 0.1, 0.1], F), ambient=np.array([0, 0, 0], F),
                     transp=np.array([0.9, 0.95, 1.0], F), refractivity=1.333)]
    p, n = quad([-12, 0, 12], [12, 0, 12], [12, 0, -12], [-12, 0, -12], [0, 1, 0], n=10)
    meshes = [Mesh("plane", 0, p, n)]
    p, n = quad([-12, 0, -6], [12, 0, -6], [12, 10, -6], [-12, 10, -6], [0, 0, 1], n=6)
    meshes.append(Mesh("backdrop", 0, p, n))
    for k, (cx, r) in enumerate([(-2.2, 1.0), (0.0, 1.2), (2.4, 0.9)]):
        p, n = uv_sphere([cx, r + 0.02, 0.0], r, sphere_res, sphere_res)
        meshes.append(Mesh("ball%d" % k, 1, p, n))
    lights = [Light(LIGHT_POINT, np.array([3.0, 7.0, 5.0], F), np.array([1, 1, 1], F))]
    cam = Camera.look([0, 2.5, 8.0], [0, 1, 0], [0, -0.18, -1])
    return Scene(meshes, mats, lights, cam)


def bunny_field(n_copies=2013, seed=7) -> Scene:
    """The rabbit mesh replicated on a jittered grid with mixed materials (README.md:107-122); about 10 M triangles
    at the default count."""
    src = Scene.load(os.path.join(GOLDEN, "stanford_bunny.rtscene"))
    rabbit = [m for m in src.meshes if m.name == "rabbit"][0]
    rng = np.random.RandomState(seed)
    side = int(np.ceil(np.sqrt(n_copies)))
    mats = [Material(kind=MAT_LAMBERT, base=np.array([0.4, 0.4, 0.4], F), ambient=np.array([0.03, 0.03, 0.03], F))]
    meshes = []
    ext = side * 1.2
    p, n = quad([-ext, 0, ext], [ext, 0, ext], [ext, 0, -ext], [-ext, 0, -ext], [0, 1, 0], n=32)
    meshes.append(Mesh("ground", 0, p, n))
    for k in range(n_copies):
        gx, gz = k % side, k // side
        off = np.array([(gx - side / 2) * 2.2 + 0.6 * rng.rand(), 0.0, (gz - side / 2) * 2.2 + 0.6 * rng.rand()], F)
        kind = [MAT_LAMBERT, MAT_PHONG, MAT_REFRAC][rng.randint(3)]
        mats.append(Material(kind=kind, base=(0.2 + 0.6 * rng.rand(3)).astype(F),
                             ambient=np.array([0.03, 0.03, 0.03], F)))
        meshes.append(Mesh("bunny%05d" % k, len(mats) - 1, (rabbit.pos + off).astype(F), rabbit.nrm))
    lights = [Light(LIGHT_DIRECTIONAL, src.lights[0].vec.copy(), np.array([1, 1, 1], F))]
    cam = Camera.look([0, side * 0.9, side * 1.6], [0, 1, 0], [0, -0.55, -1])
    return Scene(meshes, mats, lights, cam)
