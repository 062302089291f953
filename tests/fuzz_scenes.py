"""Seeded random scenes and options for differential runs of the HIP path against the CPU oracle.

The hand-made scenes of the parity tests are all well-behaved meshes; these are not: triangle soups, height-field grids
whose corners sit on a coarse lattice (rays through shared edges and corners, boxes whose faces coincide), needles,
coincident copies of triangles (ties), axis-aligned sheets (hits exactly in a box face), several meshes with every
material kind, directional and point lights anywhere (also below the geometry and inside it), a camera anywhere.
A scene is a pure function of its seed.
"""
import numpy as np

from raytracer_amd._abi import Options, RT_JITTER_HASH, RT_JITTER_OFF
from raytracer_amd.scene import (Camera, Light, Material, Mesh, Scene, MAT_DEFAULT, MAT_LAMBERT, MAT_PHONG, MAT_REFRAC)

F = np.float32


def _unit(v):
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    return (v / np.where(n > 0, n, 1)).astype(F)


def _face_normals(pos):
    n = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0])
    bad = np.linalg.norm(n, axis=1) < 1e-20
    n[bad] = np.array([0, 1, 0])
    return np.repeat(_unit(n)[:, None, :], 3, axis=1)


def _soup(rng, n, size):
    c = (rng.rand(n, 1, 3) - 0.5) * 4.0
    return (c + (rng.rand(n, 3, 3) - 0.5) * size).astype(F)


def _grid(rng, nx, nz, snap):
    """A height field over [-2, 2]^2 as 2 nx nz triangles; with snap the corners sit on a lattice of 1/4."""
    xs, zs = np.linspace(-2, 2, nx + 1), np.linspace(-2, 2, nz + 1)
    h = rng.rand(nx + 1, nz + 1) * 0.8 - 1.0
    if snap:
        h = np.round(h * 4) / 4
    v = np.stack([np.repeat(xs[:, None], nz + 1, 1), h, np.repeat(zs[None, :], nx + 1, 0)], axis=-1)
    a, b, c, d = v[:-1, :-1], v[1:, :-1], v[:-1, 1:], v[1:, 1:]
    t = np.concatenate([np.stack([a, c, b], axis=2).reshape(-1, 3, 3), np.stack([b, c, d], axis=2).reshape(-1, 3, 3)])
    return t.astype(F)


def _sheet(rng):
    """Two triangles in an axis-aligned plane at a lattice coordinate: every box around them has zero thickness."""
    axis, at = rng.randint(3), float(rng.randint(-4, 5)) / 2
    u, w = [k for k in range(3) if k != axis]
    q = np.zeros((4, 3))
    q[:, axis] = at
    ext = 1.0 + 2.0 * rng.rand()
    q[:, u] = [-ext, ext, ext, -ext]
    q[:, w] = [-ext, -ext, ext, ext]
    t = np.array([[q[0], q[1], q[2]], [q[0], q[2], q[3]]])
    if rng.rand() < 0.5:
        t = t[:, ::-1]
    return t.astype(F)


def _needles(rng, n):
    a = (rng.rand(n, 3) - 0.5) * 4
    d = _unit(rng.normal(size=(n, 3))) * (1 + 2 * rng.rand(n, 1))
    w = _unit(rng.normal(size=(n, 3))) * 1e-3
    return np.stack([a, a + d, a + d + w], axis=1).astype(F)


def random_scene(seed):
    rng = np.random.RandomState(seed)
    mats = []
    for _ in range(rng.randint(1, 5)):
        kind = [MAT_DEFAULT, MAT_LAMBERT, MAT_LAMBERT, MAT_PHONG, MAT_PHONG, MAT_REFRAC][rng.randint(6)]
        mats.append(Material(kind=kind, base=rng.rand(3).astype(F), ambient=(0.2 * rng.rand(3)).astype(F), spec=rng.rand(3).astype(F),
                             transp=(0.5 + 0.5 * rng.rand(3)).astype(F), shininess=float(2 ** rng.randint(0, 8)),
                             reflectivity=float(rng.rand()), refractivity=float(1.05 + rng.rand())))
    meshes = []
    for k in range(rng.randint(1, 6)):
        form = rng.randint(6)
        if form == 0:
            pos = _soup(rng, rng.randint(1, 400), 0.2 + 1.5 * rng.rand())
        elif form == 1:
            pos = _grid(rng, rng.randint(1, 14), rng.randint(1, 14), snap=rng.rand() < 0.6)
        elif form == 2:
            pos = _sheet(rng)
        elif form == 3:
            pos = _needles(rng, rng.randint(1, 60))
        elif form == 4:                                   # every triangle two or three times: ties at equal distance
            base = _soup(rng, rng.randint(1, 60), 1.0)
            pos = np.concatenate([base] * rng.randint(2, 4))
            pos = pos[rng.permutation(len(pos))]
        else:                                             # corners on a lattice of 1/2: shared planes everywhere
            pos = np.round(_soup(rng, rng.randint(1, 200), 2.0) * 2) / 2
            keep = np.linalg.norm(np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]), axis=1) > 0
            pos = pos[keep] if keep.any() else _sheet(rng)
        pos = pos.astype(F)
        nrm = _face_normals(pos.astype(np.float64)) if rng.rand() < 0.5 else _unit(rng.normal(size=pos.shape))
        meshes.append(Mesh("mesh%d" % k, int(rng.randint(len(mats))), pos, nrm))
    lights = []
    for _ in range(rng.randint(0, 4)):
        if rng.rand() < 0.5:
            lights.append(Light(0, _unit(rng.normal(size=3)), rng.rand(3).astype(F)))
        else:
            lights.append(Light(1, (rng.normal(size=3) * 3).astype(F), rng.rand(3).astype(F)))
    eye = _unit(rng.normal(size=3)) * (1.0 + 6.0 * rng.rand())
    look = (rng.rand(3) - 0.5) - eye
    up = [0, 1, 0] if abs(_unit(look)[1]) < 0.95 else [1, 0, 0]
    return Scene(meshes, mats, lights, Camera.look(eye.astype(F), up, look.astype(F)))


def random_big_scene(seed):
    """Tens of thousands of triangles in clusters of very different density (deep, lopsided trees: the traversal stack
    beyond its LDS part, leaves at every depth), a fine height field and a sheet, with the random materials, lights
    and camera of random_scene."""
    small = random_scene(seed)
    rng = np.random.RandomState(seed + 424242)
    parts = []
    for _ in range(rng.randint(3, 9)):
        n = int(10 ** (2 + 2.3 * rng.rand()))
        centre, spread, size = (rng.rand(1, 1, 3) - 0.5) * 4, 10 ** (-2 + 2 * rng.rand()), 10 ** (-3 + 2 * rng.rand())
        parts.append(centre + rng.normal(size=(n, 1, 3)) * spread + (rng.rand(n, 3, 3) - 0.5) * size)
    soup = np.concatenate(parts).astype(F)
    soup = soup[rng.permutation(len(soup))]
    cut = len(soup) // 2
    grid = _grid(rng, rng.randint(40, 120), rng.randint(40, 120), snap=rng.rand() < 0.5)
    meshes = []
    for k, pos in enumerate([soup[:cut], soup[cut:], grid, _sheet(rng)]):
        nrm = _face_normals(pos.astype(np.float64))
        meshes.append(Mesh("big%d" % k, int(rng.randint(len(small.materials))), pos, nrm))
    return Scene(meshes, small.materials, small.lights, small.camera)


def random_options(seed, exact):
    rng = np.random.RandomState(seed + 77777)
    return Options(width=int(rng.randint(8, 49)), height=int(rng.randint(8, 33)), hfov=float(20 + 80 * rng.rand()),
                   spp=int(rng.randint(1, 4)), spr=int(rng.randint(0, 5)), max_bounce=int(rng.randint(0, 4)),
                   max_refrac_bounce=int(rng.randint(1, 5)), eps=float([1e-4, 1e-3, 1e-2][rng.randint(3)]),
                   ray_eps=float([1e-3, 1e-2, 0.1][rng.randint(3)]),
                   jitter_mode=[RT_JITTER_OFF, RT_JITTER_HASH][rng.randint(2)], seed=int(rng.randint(1 << 20)), exact_boxes=int(exact))


def random_options_wide(seed, exact):
    """The corners of the option space random_options leaves out: up to 81 samples per pixel (both replay forms:
    fused, and in two passes from 16 samples on scenes that spawn rays), up to 16 light rays per hit and light, up to
    five bounces - on small frames, with the number of children bounded where the ray tree is deep."""
    rng = np.random.RandomState(seed + 31337)
    spp = int([1, 2, 3, 4, 4, 5, 6, 9][rng.randint(8)])
    bounce = int(rng.randint(0, 6))
    spr = int([0, 1, 2, 3, 5, 8, 15][rng.randint(7 if bounce <= 1 else (5 if bounce == 2 else 3))])
    side = 20 if spp <= 3 else 10
    return Options(width=int(rng.randint(4, side + 1)), height=int(rng.randint(4, side + 1)), hfov=float(20 + 80 * rng.rand()),
                   spp=spp, spr=spr, max_bounce=bounce, max_refrac_bounce=int(rng.randint(1, 7)),
                   eps=float([1e-4, 1e-3, 1e-2][rng.randint(3)]), ray_eps=float([1e-3, 1e-2, 0.1][rng.randint(3)]),
                   jitter_mode=[RT_JITTER_OFF, RT_JITTER_HASH][rng.randint(2)], seed=int(rng.randint(1 << 20)), exact_boxes=int(exact))


def more_lights(scene, seed):
    """Up to eight lights (the most a scene may have)."""
    rng = np.random.RandomState(seed + 999)
    while len(scene.lights) < rng.randint(1, 9):
        if rng.rand() < 0.5:
            scene.lights.append(Light(0, _unit(rng.normal(size=3)), (0.3 * rng.rand(3)).astype(F)))
        else:
            scene.lights.append(Light(1, (rng.normal(size=3) * 3).astype(F), (0.3 * rng.rand(3)).astype(F)))
    return scene


def has_refraction(scene):
    used = {m.material for m in scene.meshes}
    return any(scene.materials[k].kind == MAT_REFRAC for k in used)
