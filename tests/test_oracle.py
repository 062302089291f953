"""The CPU oracle against known answers.

The reference ships no tests or golden vectors for this path and cannot be built here, so the oracle is
"parity unpinned"; what can be checked is that it answers analytic cases correctly, that its tree never
changes an answer (brute force over all triangles gives the same hit) and that its tie rules are the
reference's (src/geometry.cpp:171-175, :428; src/render_thread.cpp:15)."""
import numpy as np
import pytest

from raytracer_amd import scenes
from raytracer_amd._abi import Options
from raytracer_amd.scene import (Camera, Light, Material, Mesh, Scene, LIGHT_DIRECTIONAL, MAT_DEFAULT, MAT_LAMBERT)
from tests.conftest import GOLDEN, bits_equal, random_rays

F = np.float32


def one_mesh_scene(pos, nrm=None, material=None, lights=None):
    pos = np.asarray(pos, F).reshape(-1, 3, 3)
    if nrm is None:
        e1, e2 = pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]
        n = np.cross(e1, e2)
        n = n / np.linalg.norm(n, axis=1, keepdims=True)
        nrm = np.repeat(n[:, None, :], 3, axis=1)
    mats = [material or Material(kind=MAT_LAMBERT, base=np.array([0.5, 0.25, 1.0], F), ambient=np.array([0.1, 0.1, 0.1], F))]
    return Scene([Mesh("m", 0, pos, nrm)], mats, lights or [], Camera.look([0, 0, 5], [0, 1, 0], [0, 0, -1]))


def brute_force(scene, org, d, t0, t1):
    """float64 Cramer over every triangle; returns nearest t and its global id."""
    P = np.concatenate([m.pos for m in scene.meshes]).astype(np.float64)
    A, B, Cc = P[:, 0], P[:, 1], P[:, 2]
    best_t = np.full(len(org), np.inf)
    best_i = np.full(len(org), -1)
    for k in range(len(org)):
        o, dd = org[k].astype(np.float64), d[k].astype(np.float64)
        ab, ac, ao = A - B, A - Cc, A - o
        M = np.stack([ab, ac, np.broadcast_to(dd, ab.shape)], axis=2)
        det = np.linalg.det(M)
        ok = np.abs(det) > 1e-14
        sol = np.zeros((len(A), 3))
        sol[ok] = np.linalg.solve(M[ok], ao[ok][..., None])[..., 0]
        beta, gamma, t = sol[:, 0], sol[:, 1], sol[:, 2]
        hit = ok & (t >= t0) & (t <= t1) & (gamma >= 0) & (gamma <= 1) & (beta >= 0) & (beta <= 1 - gamma)
        if hit.any():
            tt = np.where(hit, t, np.inf)
            best_i[k] = int(np.argmin(tt))
            best_t[k] = tt[best_i[k]]
    return best_t, best_i


def test_single_triangle_known_answer(oracle_mod):
    sc = one_mesh_scene([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]],
                        nrm=[[[0, 0, 1], [0, 1, 0], [1, 0, 0]]])
    o = oracle_mod.OracleScene(sc)
    org = np.array([[0.25, 0.25, 2.0], [0.9, 0.9, 2.0], [0.25, 0.25, -2.0], [0.25, 0.25, 2.0]], F)
    d = np.array([[0, 0, -1], [0, 0, -1], [0, 0, 1], [0, 0, 1]], F)
    t, tri, nrm = o.trace(org, d, 1e-4, np.inf)
    assert t[0] == F(2.0) and tri[0] == 0
    assert tri[1] == -1 and np.isinf(t[1])          # outside: beta + gamma > 1
    assert t[2] == F(2.0) and tri[2] == 0           # no back-face culling (src/geometry.cpp:266-312)
    assert tri[3] == -1                              # behind the origin
    # barycentric normal: alpha = .5 (vertex a), beta = .25 (b), gamma = .25 (c) -> normalise(.25, .25, .5)
    want = np.array([0.25, 0.25, 0.5]) / np.linalg.norm([0.25, 0.25, 0.5])
    assert np.allclose(nrm[0], want, atol=1e-6)
    # t range is inclusive at both ends (src/geometry.cpp:291)
    t, tri, _ = o.trace(org[:1], d[:1], 2.0, 2.0)
    assert tri[0] == 0
    t, tri, _ = o.trace(org[:1], d[:1], 2.0001, 10.0)
    assert tri[0] == -1


def test_tie_rules(oracle_mod):
    # two identical triangles in one mesh: a two-surface node prefers the RIGHT one (index 1)
    tri = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    o = oracle_mod.OracleScene(one_mesh_scene([tri, tri]))
    _, idx, _ = o.trace(np.array([[0.2, 0.2, 1]], F), np.array([[0, 0, -1]], F), 1e-4, np.inf)
    assert idx[0] == 1
    # three identical triangles cannot be split: they become one set, which keeps the LOWEST index
    o = oracle_mod.OracleScene(one_mesh_scene([tri, tri, tri]))
    _, idx, _ = o.trace(np.array([[0.2, 0.2, 1]], F), np.array([[0, 0, -1]], F), 1e-4, np.inf)
    assert idx[0] == 0
    # the same triangle in two meshes: the FIRST mesh wins
    sc = one_mesh_scene([tri])
    sc.meshes.append(Mesh("m2", 0, sc.meshes[0].pos.copy(), sc.meshes[0].nrm.copy()))
    o = oracle_mod.OracleScene(sc)
    _, idx, _ = o.trace(np.array([[0.2, 0.2, 1]], F), np.array([[0, 0, -1]], F), 1e-4, np.inf)
    assert idx[0] == 0


@pytest.mark.parametrize("name", ["unit_cube", "unit_sphere", "unit_cylinder", "xyz"])
def test_tree_matches_brute_force(oracle_mod, name):
    sc = Scene.load("%s/%s.rtscene" % (GOLDEN, name))
    o = oracle_mod.OracleScene(sc)
    org, d = random_rays(300, 11, center=(0, 0, 0), spread=5.0)
    t, tri, _ = o.trace(org, d, 1e-4, np.inf)
    bt, bi = brute_force(sc, org, d, 1e-4, np.inf)
    hit = bi >= 0
    assert ((tri >= 0) == hit).mean() > 0.99         # grazing rays may differ between float32 and float64
    both = hit & (tri >= 0)
    assert np.allclose(t[both], bt[both], rtol=1e-4, atol=1e-5)
    assert (tri[both] == bi[both]).mean() > 0.97      # shared edges: equal t on two triangles


def test_unit_sphere_distance(oracle_mod):
    sc = Scene.load(GOLDEN + "/unit_sphere.rtscene")
    P = np.concatenate([m.pos for m in sc.meshes]).reshape(-1, 3)
    r_out = np.linalg.norm(P - P.mean(0), axis=1).max()
    o = oracle_mod.OracleScene(sc)
    rng = np.random.RandomState(5)
    dirs = rng.randn(200, 3)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    c = P.mean(0)
    org = (c + 4.0 * dirs).astype(F)
    t, tri, nrm = o.trace(org, (-dirs).astype(F), 1e-4, np.inf)
    assert (tri >= 0).all()
    # a faceted sphere lies between the inscribed and the circumscribed sphere
    assert (t >= 4.0 - r_out - 1e-4).all() and (t <= 4.0 - 0.9 * r_out).all()
    # smooth normals point back along the ray
    assert ((nrm * dirs).sum(1) > 0.9).all()


def test_lambert_pixel_known_answer(oracle_mod):
    # a big wall facing the camera, one directional light: every sample hits, nothing occludes
    L = Light(LIGHT_DIRECTIONAL, np.array([0, 0.6, 0.8], F), np.array([1.0, 0.5, 0.25], F))
    wall = [[[-50, -50, 0], [50, -50, 0], [50, 50, 0]], [[-50, -50, 0], [50, 50, 0], [-50, 50, 0]]]
    sc = one_mesh_scene(wall, lights=[L])
    o = oracle_mod.OracleScene(sc)
    opt = Options(width=8, height=6, hfov=60, spp=2, spr=3, max_bounce=3)
    img, st = o.render_rect(opt, 0, 0, 8, 6)
    m = sc.materials[0]
    want = m.base * (m.ambient + L.color * F(0.8))     # cr * (ca + cl * max(0, n.l)), shadow fraction 1
    assert np.allclose(img, want[None, None, :], rtol=1e-6)
    assert st.rays_closest == 8 * 6 * 4 * 2 and st.rays_any == 8 * 6 * 4 * 4
    # an occluder between wall and light: fully shadowed pixels are exactly 0 (ambient is scaled too, src/materials.cpp:40)
    occl = [[[-50, -70, 40], [50, -70, 40], [50, 130, 40]], [[-50, -70, 40], [50, 130, 40], [-50, 130, 40]]]
    sc2 = one_mesh_scene(wall + occl, lights=[L])
    img2, _ = oracle_mod.OracleScene(sc2).render_rect(opt, 3, 2, 5, 4)
    assert (img2 == 0).all()


def test_default_material_and_background(oracle_mod):
    sc = one_mesh_scene([[[-1, -1, 0], [1, -1, 0], [0, 1, 0]]], material=Material(kind=MAT_DEFAULT))
    o = oracle_mod.OracleScene(sc)
    opt = Options(width=16, height=16, hfov=40, spp=1, spr=0, max_bounce=1)
    img, _ = o.render_rect(opt, 0, 0, 16, 16)
    assert np.isnan(img[0, 0]).all()                    # no draw: 0/0 (src/rasterizer.cpp:97)
    centre = img[8, 8]
    assert (centre == np.array([1, 1, 0], F)).all()     # src/materials.cpp:4-8
    u8 = oracle_mod.resolve_u8(img)
    assert (u8[15, 0] == 0).all()                       # NaN -> black (src/main.cpp:165-167)
    assert (u8[16 - 8 - 1, 8] == np.array([255, 255, 0])).all()


def test_tree_shape_of_bunny(oracle_mod, bunny_scene):
    info = oracle_mod.OracleScene(bunny_scene).info()
    # total BVHNode count of the reference's trees on this scene: 6085 + 3055 + 15 (inner nodes + bottom nodes)
    assert info["tris"] == 7864 and info["inner"] + info["leaves"] == 9155 and info["depth"] == 20


def test_golden_vectors(oracle_mod):
    """Regression vectors written by tests/golden/make_golden.py from this same oracle (they pin the oracle against
    accidental edits, not against the reference)."""
    g = np.load(GOLDEN + "/oracle_vectors.npz")
    o = oracle_mod.OracleScene(scenes.bunny())
    t, tri, nrm = o.trace(g["org"], g["dir"], 1e-4, np.inf)
    assert (tri == g["tri"]).all() and bits_equal(t, g["t"]).all() and bits_equal(nrm, g["nrm"]).all()
    opt = Options(width=96, height=54, hfov=60, spp=2, spr=3, max_bounce=3)
    img, _ = o.render_rect(opt, 0, 0, 96, 54)
    assert bits_equal(img, g["bunny_96x54"]).all()
    opt = Options(width=64, height=36, hfov=60, spp=2, spr=3, max_bounce=2, eps=1e-3)
    img, _ = oracle_mod.OracleScene(scenes.bunny(phong_surroundings=True)).render_rect(opt, 0, 0, 64, 36)
    assert bits_equal(img, g["bunny_phong_64x36"]).all()


def test_random_scenes_are_reproducible_and_the_oracle_renders_them(oracle_mod):
    """The seeded scene generator of the GPU parity runs: the same seed gives the same bytes, and the oracle renders
    whatever comes out (degenerate sheets, needles, coincident triangles) to the same frame twice."""
    from tests.fuzz_scenes import random_scene, random_options
    for seed in (0, 7, 30, 49):
        a, b = random_scene(seed), random_scene(seed)
        assert len(a.meshes) == len(b.meshes) and all((x.pos == y.pos).all() and (x.nrm == y.nrm).all() for x, y in zip(a.meshes, b.meshes))
        a.validate()
        o = oracle_mod.OracleScene(a)
        opt = random_options(seed, 1)
        f1, s1 = o.render_rect(opt, 0, 0, opt.width, opt.height)
        f2, s2 = o.render_rect(opt, 0, 0, opt.width, opt.height)
        assert f1.tobytes() == f2.tobytes() and (s1.rays_closest, s1.rays_any) == (s2.rays_closest, s2.rays_any)
        # seed 49 refracts through random normals: some rays come out NaN and "hit" without a record
        assert (s1.reserved_[0] > 0) == (seed in (30, 49))
