"""Parity of the HIP path (through the C-ABI of librt_amd.so) with the CPU oracle, on a real MI355X.

Bars (stated by the project's target): primary-hit triangle ids equal on >= 99.99 % of rays, hit t within 1e-5
relative, 8-bit pixels within 1.  What is asserted here is stricter wherever the design makes it exact:
  * exact_boxes=1 (the reference's slab arithmetic, no pruning): ids, t and normals BIT-EQUAL to the oracle;
  * exact_boxes=0 (reciprocal slab test + pruning, suspect winners walked again against the reference's leaf boxes):
    the frame and the rays of exact_boxes=1; against the oracle the looser bar is kept where scenes hold needles,
    whose far-outside "hits" of the reference's Cramer test no tree of tighter boxes reaches;
  * exact_boxes=2 (the same without the second walk): ids >= 99.99 % equal, and where the id is equal t and the
    normal are bit-equal (the triangle test is the same arithmetic);
  * float framebuffers: bit-equal for Lambert/Phong scenes (colour arithmetic is replayed in the reference's
    order); within 1e-5 relative for refractive scenes (device expf differs from glibc expf in the last place).
"""
import numpy as np
import pytest

from raytracer_amd import api, scenes
from raytracer_amd._abi import Options, RT_JITTER_HASH, RT_JITTER_OFF
from raytracer_amd.scene import Scene
from tests.conftest import GOLDEN, bits_equal, random_rays

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    api.init()


def _pair(scene, oracle_mod):
    return api.TracerScene(scene), oracle_mod.OracleScene(scene)


@pytest.mark.parametrize("name", ["stanford_bunny", "simple_knife", "unit_sphere_cam", "unit_cube"])
@pytest.mark.parametrize("exact", [1, 0])
def test_trace_parity_random_rays(oracle_mod, name, exact):
    sc = Scene.load("%s/%s.rtscene" % (GOLDEN, name))
    g, o = _pair(sc, oracle_mod)
    P = np.concatenate([m.pos for m in sc.meshes]).reshape(-1, 3)
    c = P.mean(0)
    ext = float(np.abs(P - c).max())
    org, d = random_rays(40000, 7, center=c, spread=3.0 * ext)
    # aim half of the rays at points near the surface so that thin meshes are hit often too
    rng = np.random.RandomState(3)
    k = len(org) // 2
    tgt = P[rng.randint(len(P), size=k)] + (rng.rand(k, 3) - 0.5) * 0.02 * ext
    dd = tgt - org[-k:]
    d[-k:] = (dd / np.linalg.norm(dd, axis=1, keepdims=True)).astype(np.float32)
    t0 = 1e-4
    to, trio, nro = o.trace(org, d, t0, np.inf)
    tg, trig, nrg = g.trace(org, d, t0, np.inf, exact_boxes=bool(exact))
    same = trio == trig
    if exact:
        assert same.all()
    else:
        assert same.mean() >= 0.9999
    assert bits_equal(to[same], tg[same]).all()
    assert bits_equal(nro[same], nrg[same]).all()
    assert (trio >= 0).mean() > 0.05
    # any-hit answers the reference's boolean (src/render_thread.cpp:158)
    ao, _, _ = o.trace(org, d, t0, 10000.0, any_hit=True)
    ag, _, _ = g.trace(org, d, t0, 10000.0, any_hit=True, exact_boxes=bool(exact))
    assert (ao == ag).mean() >= (1.0 if exact else 0.9999)


def test_trace_edge_cases(oracle_mod, bunny_scene):
    g, o = _pair(bunny_scene, oracle_mod)
    # empty batch
    t, tri, n = g.trace(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), 1e-4, np.inf)
    assert t.shape == (0,)
    # a single ray, rays starting inside the dome, rays along a shared edge of the ground grid, t range cut-offs
    org = np.array([[0, 1.5, 4], [0, 0.3, 0], [0.0, 5.0, 0.0], [2.0, 5.0, 2.0], [0, 1.5, 4]], np.float32)
    d = np.array([[0, -0.2, -1], [0, 1, 0], [0, -1, 0], [0, -1, 0], [0, 1, 0]], np.float32)
    # The first ray runs exactly inside the plane x = 0 with direction x = 0: where that plane is a face of one of the
    # reference's boxes, its NaN-dropping fmin/fmax discard the box on one side and it misses the nearer triangle.
    # The reference's slab arithmetic on the reference's tree (exact_boxes) reproduces that; the pruning traversal
    # walks another tree and may only differ by finding a NEARER hit.
    for exact in (True, False):
        for t1 in (np.inf, 2.0):
            to, trio, _ = o.trace(org, d, 1e-4, t1)
            tg, trig, _ = g.trace(org, d, 1e-4, t1, exact_boxes=exact)
            same = trio == trig
            assert bits_equal(to[same], tg[same]).all()
            if exact:
                assert same.all()
            else:
                assert same[1:].all() and (tg[~same] < to[~same]).all()


def test_zero_area_triangle_in_a_mesh_of_its_own(oracle_mod):
    """A triangle whose three corners coincide makes Triangle::hit divide 0 by 0: t is NaN, which passes every
    comparison of the test, and get_hit never keeps such a record (reference src/render_thread.cpp:15).  With the
    degenerate triangle in a mesh of its own the reference's answer is well defined - the ray goes on to whatever lies
    behind - and the GPU must give it in both box modes: for batches of rays and inside a frame."""
    from tests.test_oracle import one_mesh_scene
    from raytracer_amd.scene import Mesh, Light
    real = [[-1, -1, 0], [1, -1, 0], [0, 1, 0]]
    sc = one_mesh_scene([real], lights=[Light(0, np.array([0, 0, 1], np.float32), np.array([1, 1, 1], np.float32))])
    pt = np.array([[[0.25, 0.25, 0.5]] * 3], np.float32)
    sc.meshes.insert(0, Mesh("speck", 0, pt, np.tile(np.array([0, 0, 1], np.float32), (1, 3, 1))))
    g, o = _pair(sc, oracle_mod)
    org = np.array([[0.25, 0.25, 1.0], [0.25, 0.25, 1.0], [0.0, 0.0, 2.0], [0.25, 0.25, 0.75]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, 1], [0.125, 0.125, -0.75], [0, 0, -1]], np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    to, trio, _ = o.trace(org, d, 1e-4, np.inf)
    assert trio[0] == 1 and trio[1] == -1 and trio[2] == 1        # the speck is never the answer
    for exact in (True, False):
        tg, trig, _ = g.trace(org, d, 1e-4, np.inf, exact_boxes=exact)
        assert (trig == trio).all() and bits_equal(tg, to).all()
        hit_any, _, _ = g.trace(org, d, 1e-4, np.inf, any_hit=True, exact_boxes=exact)
        assert (np.isfinite(hit_any) == (trio >= 0)).all()
        opt = Options(width=48, height=32, hfov=40, spp=2, spr=2, max_bounce=2, jitter_mode=RT_JITTER_OFF, exact_boxes=int(exact))
        img_o = o.render_rect(opt, 0, 0, 48, 32)[0]
        img_g = g.render_rect(opt, 0, 0, 48, 32)[0]
        assert bits_equal(img_o, img_g).all()


def test_triangles_with_subnormal_determinants(oracle_mod):
    """Triangles with edges around 1e-20 at the origin: the determinant of the reference's Cramer test (edge x edge . d)
    is then a SUBNORMAL float on nearly every ray that meets one, while the quotients t, beta and gamma are ordinary
    numbers.  The hardware reciprocal of a subnormal is infinite, so the reciprocal estimate that normally rejects far-off
    triangles early must stand aside and the exact divisions decide: ids, distances and normals bit-equal to the oracle
    in the reference-arithmetic mode, the any-hit answer equal, the pruning mode never farther."""
    from tests.test_oracle import one_mesh_scene
    from raytracer_amd.scene import Mesh
    F = np.float32
    rng = np.random.RandomState(11)
    n, m = 300, 4000
    a = (rng.rand(n, 1, 3) - 0.5) * 2e-19
    size = 10 ** rng.uniform(-21, -18.5, size=(n, 1, 1))
    pos = (a + (rng.rand(n, 3, 3) - 0.5) * size).astype(F)
    sc = one_mesh_scene([[[-3, -3, -1], [3, -3, -1], [0, 3, -1]]])            # something ordinary behind the specks
    # every speck a mesh of its own: where products underflow to a determinant of exactly zero the test divides 0 by 0,
    # and inside ONE mesh the reference's pairwise merge lets such a NaN record hide its siblings' hits
    # (src/geometry.cpp:171-175, see test_zero_area_triangle_in_a_mesh_of_its_own); across meshes get_hit drops it
    for k in range(n):
        sc.meshes.append(Mesh("speck%d" % k, 0, pos[k:k + 1], np.tile(np.array([0, 0, 1], F), (1, 3, 1))))
    org = rng.normal(size=(m, 3))
    org = (org / np.linalg.norm(org, axis=1, keepdims=True) * rng.uniform(0.3, 2, size=(m, 1))).astype(F)
    corner = pos.reshape(-1, 3)[rng.randint(3 * n, size=m)].astype(np.float64)
    inside = (pos[rng.randint(n, size=m)].astype(np.float64) * rng.dirichlet([1, 1, 1], size=m)[:, :, None]).sum(1)
    tgt = np.where((np.arange(m) % 2 == 0)[:, None], corner, inside)
    d = tgt - org.astype(np.float64)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F)
    g, o = _pair(sc, oracle_mod)
    to, trio, nro = o.trace(org, d, 1e-4, np.inf)
    hit = trio >= 1
    assert hit.sum() > 1000                                                   # the specks are hit, and then ...
    T = np.concatenate([msh.pos for msh in sc.meshes])[trio[hit]]
    det = ((T[:, 0] - T[:, 1]).astype(F) * np.cross((T[:, 0] - T[:, 2]).astype(F), d[hit])).sum(1)
    assert (np.abs(det) < 1.17549435e-38).mean() > 0.99                       # ... through a subnormal determinant
    ao, _, _ = o.trace(org, d, 1e-4, 10000.0, any_hit=True)
    tg, trig, nrg = g.trace(org, d, 1e-4, np.inf, exact_boxes=True)
    assert (trig == trio).all() and bits_equal(tg, to).all() and bits_equal(nrg, nro).all()
    ag, _, _ = g.trace(org, d, 1e-4, 10000.0, any_hit=True, exact_boxes=True)
    assert (ag == ao).all()
    # The pruning traversal has no "never farther" bar here: three hundred specks sit at one point, and a distance that is
    # a quotient of two numbers of nine significant bits is off by tens of per cent (the same point comes out at t = 1.0
    # from one triangle and 1.29 from the next), so which of them is "nearest" is noise that pruning by box distance
    # cannot and need not follow.  What holds: the same triangle gives the same bits, and a ray that hits something for
    # the oracle hits something.
    tp, trip, nrp = g.trace(org, d, 1e-4, np.inf, exact_boxes=False)
    same = trip == trio
    assert bits_equal(tp[same], to[same]).all() and bits_equal(nrp[same], nro[same]).all()
    assert (trip >= 0)[trio >= 0].all()
    far = np.where(trio >= 1)[0]
    true_dist = np.linalg.norm(org, axis=1)                                    # the specks are at the origin
    assert (np.abs(tp[far] - true_dist[far]) <= 0.6 * true_dist[far]).all()


def test_triangles_with_determinants_beyond_the_reciprocal(oracle_mod):
    """The other end: triangles with edges around 1.2e19 whose determinant lands between 8.5e37 and the largest float.
    The hardware reciprocal of such a number is a subnormal, which it flushes to zero - an estimate of zero for every
    quotient - while the exact quotients are ordinary (rays that start within a unit of the triangle's first corner and
    hit the wedge beside it).  As above: the estimate stands aside, the exact divisions decide, bit-equal to the oracle."""
    from tests.test_oracle import one_mesh_scene
    from raytracer_amd.scene import Mesh
    F = np.float32
    rng = np.random.RandomState(5)
    n, per = 60, 40
    sc = one_mesh_scene([[[-3, -3, -40], [3, -3, -40], [0, 3, -40]]])
    org, d = [], []
    for k in range(n):
        a = rng.rand(3) - 0.5
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        w = rng.normal(size=3); w -= u * (w @ u); w /= np.linalg.norm(w)
        v = 0.3 * u + 0.95 * w; v /= np.linalg.norm(v)                       # a wide corner: sine about 0.95
        L1, L2 = rng.uniform(1.0e19, 1.4e19, size=2)
        tri = np.stack([a, a + L1 * u, a + L2 * v]).astype(F)
        sc.meshes.append(Mesh("huge%d" % k, 0, tri[None], np.tile(np.array([0, 0, 1], F), (1, 3, 1))))
        nrm = np.cross(u, v); nrm /= np.linalg.norm(nrm)
        for _ in range(per):
            q = a + rng.uniform(0.05, 1.0) * (rng.uniform(0.1, 0.9) * u + rng.uniform(0.1, 0.9) * v)     # inside the wedge at a
            side = 1.0 if rng.rand() < 0.5 else -1.0
            e = q + side * nrm * rng.uniform(0.3, 1.5) + 0.2 * rng.normal(size=3)
            dd = q - e
            org.append(e); d.append(dd / np.linalg.norm(dd))
    org, d = np.array(org, F), np.array(d, F)
    g, o = _pair(sc, oracle_mod)
    to, trio, nro = o.trace(org, d, 1e-4, np.inf)
    hit = trio >= 1
    assert hit.mean() > 0.5
    T = np.concatenate([msh.pos for msh in sc.meshes])[trio[hit]]
    det = np.abs(((T[:, 0] - T[:, 1]).astype(F) * np.cross((T[:, 0] - T[:, 2]).astype(F), d[hit])).sum(1))
    assert ((det > 8.6e37) & np.isfinite(det)).sum() > 800                    # where the hardware reciprocal is zero
    for exact in (True, False):
        tg, trig, nrg = g.trace(org, d, 1e-4, np.inf, exact_boxes=exact)
        assert (trig == trio).all() and bits_equal(tg, to).all() and bits_equal(nrg, nro).all()
        ag, _, _ = g.trace(org, d, 1e-4, 1e30, any_hit=True, exact_boxes=exact)
        ao, _, _ = o.trace(org, d, 1e-4, 1e30, any_hit=True)
        assert (ag == ao).all()


def test_tie_rules_on_gpu(oracle_mod):
    from tests.test_oracle import one_mesh_scene
    from raytracer_amd.scene import Mesh
    tri = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    org, d = np.array([[0.2, 0.2, 1]], np.float32), np.array([[0, 0, -1]], np.float32)
    for tris, want in (([tri, tri], 1), ([tri, tri, tri], 0)):
        for exact in (True, False):
            _, idx, _ = api.TracerScene(one_mesh_scene(tris)).trace(org, d, 1e-4, np.inf, exact_boxes=exact)
            assert idx[0] == want
    sc = one_mesh_scene([tri])
    sc.meshes.append(Mesh("m2", 0, sc.meshes[0].pos.copy(), sc.meshes[0].nrm.copy()))
    _, idx, _ = api.TracerScene(sc).trace(org, d, 1e-4, np.inf)
    assert idx[0] == 0


CASES = {
    "bunny_lambert": (lambda: scenes.bunny(), dict(width=192, height=108, hfov=60, spp=2, spr=3, max_bounce=3), True),
    "bunny_2lights": (lambda: scenes.bunny(second_light=True), dict(width=160, height=90, hfov=60, spp=2, spr=3, max_bounce=3), True),
    "bunny_phong": (lambda: scenes.bunny(phong_surroundings=True), dict(width=128, height=72, hfov=60, spp=2, spr=3, max_bounce=2, eps=1e-3), True),
    "mirror_room": (lambda: scenes.mirror_room(6, sphere_res=8), dict(width=80, height=48, hfov=60, spp=2, spr=3, max_bounce=3, eps=1e-3, ray_eps=0.01), True),
    "glass_balls": (lambda: scenes.glass_balls(10), dict(width=80, height=48, hfov=54.5, spp=2, spr=4, max_bounce=3, max_refrac_bounce=5, eps=1e-3), False),
    "spp1_spr0": (lambda: scenes.bunny(phong_surroundings=True), dict(width=64, height=36, hfov=60, spp=1, spr=0, max_bounce=1), True),
    "spp3": (lambda: scenes.bunny(), dict(width=48, height=27, hfov=54.43, spp=3, spr=4, max_bounce=3), True),
    # 16 samples per pixel on scenes whose materials spawn rays: the replay runs in two passes (a lane per sample
    # shades, a lane per pixel adds in pop order)
    "mirror_room_spp4": (lambda: scenes.mirror_room(5, sphere_res=6), dict(width=40, height=24, hfov=60, spp=4, spr=2, max_bounce=3, eps=1e-3, ray_eps=0.01), True),
    "glass_balls_spp4": (lambda: scenes.glass_balls(8), dict(width=40, height=24, hfov=54.5, spp=4, spr=2, max_bounce=2, max_refrac_bounce=4, eps=1e-3), False),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("jitter", [RT_JITTER_OFF, RT_JITTER_HASH])
@pytest.mark.parametrize("exact", [1, 0])
def test_render_parity(oracle_mod, case, jitter, exact):
    make, kw, bit_exact = CASES[case]
    sc = make()
    g, o = _pair(sc, oracle_mod)
    opt = Options(jitter_mode=jitter, seed=5, exact_boxes=exact, **kw)
    W, H = opt.width, opt.height
    a, sa = o.render_rect(opt, 0, 0, W, H)
    b, sb = g.render_rect(opt, 0, 0, W, H)
    assert (np.isnan(a) == np.isnan(b)).mean() >= 0.9999
    fin = np.isfinite(a) & np.isfinite(b)
    if bit_exact and exact:
        assert bits_equal(a, b).all()
    elif bit_exact:
        # the pruning traversal may differ where a ray passes exactly through an edge shared by two triangles (equal t):
        # the reference takes whichever of the two its own boxes let it test, the pruning traversal tests both and takes
        # the lower tie rank.  At most one pixel in ten thousand, or a single pixel of these small frames.
        assert (~bits_equal(a, b).all(axis=-1)).sum() <= max(1, 1e-4 * W * H)
    else:
        assert np.allclose(a[fin], b[fin], rtol=1e-5, atol=1e-6)
    ua, ub = oracle_mod.resolve_u8(a), api.resolve_u8(b)
    diff = np.abs(ua.astype(int) - ub.astype(int))
    assert diff.max() <= 1 or (diff > 1).mean() < 1e-4
    # the GPU traces each render ray once, the reference twice (its shadow-type twin repeats the ray)
    if exact:
        assert sb.rays_any == sa.rays_any and 2 * sb.rays_closest == sa.rays_closest
        assert sb.ref_get_hit_calls == sa.rays_closest + sa.rays_any


def _default_and_point_light_scene():
    """The bunny scene with the rabbit on the default material (its colour ignores the shadow record) under a point
    light (the shaded light direction is jittered per hit, src/lights.cpp:34-36)."""
    from raytracer_amd.scene import Light, Material, MAT_DEFAULT
    sc = scenes.bunny()
    sc.materials.append(Material(kind=MAT_DEFAULT))
    for m in sc.meshes:
        if m.name == "rabbit":
            m.material = len(sc.materials) - 1
    sc.lights = [Light(1, np.array([3.0, 6.0, 4.0], np.float32), np.array([1, 0.9, 0.8], np.float32))]
    return sc


@pytest.mark.parametrize("case", ["bunny_lambert", "bunny_2lights", "spp3", "default_and_point_light"])
@pytest.mark.parametrize("jitter", [RT_JITTER_OFF, RT_JITTER_HASH])
@pytest.mark.parametrize("exact", [1, 0])
def test_skipping_idle_light_rays_changes_no_pixel(oracle_mod, case, jitter, exact):
    """rt_options.skip_unlit leaves out the light rays of hits whose colour the light cannot change (default material,
    Lambert term zero): the frame must stay the frame the oracle renders WITH those rays, bit for bit; fewer rays are
    traced, and the count of get_hit calls the reference would make is still reported."""
    if case == "default_and_point_light":
        sc, kw = _default_and_point_light_scene(), dict(width=160, height=90, hfov=60, spp=2, spr=3, max_bounce=3)
    else:
        make, kw, _ = CASES[case]
        sc = make()
    g, o = _pair(sc, oracle_mod)
    opt = Options(jitter_mode=jitter, seed=5, exact_boxes=exact, **kw)
    W, H = opt.width, opt.height
    a, sa = o.render_rect(opt, 0, 0, W, H)
    full, sf = g.render_rect(opt, 0, 0, W, H)
    opt.skip_unlit = 1
    b, sb = g.render_rect(opt, 0, 0, W, H)
    assert bits_equal(full, b).all()                       # the same frame as with every light ray traced
    if exact:
        assert bits_equal(a, b).all()                      # which is the oracle's
        assert sb.ref_get_hit_calls == sa.rays_closest + sa.rays_any
    assert sb.rays_closest == sf.rays_closest and sb.rays_any < sf.rays_any
    assert sb.ref_get_hit_calls == sf.ref_get_hit_calls
    # the option is ignored where a later ray of the pixel may read the record these rays leave behind
    g2 = api.TracerScene(scenes.bunny(phong_surroundings=True))
    o2 = Options(width=64, height=36, hfov=60, spp=2, spr=3, max_bounce=2, eps=1e-3, jitter_mode=jitter, seed=5, exact_boxes=exact)
    f0, s0 = g2.render_rect(o2, 0, 0, 64, 36)
    o2.skip_unlit = 1
    f1, s1 = g2.render_rect(o2, 0, 0, 64, 36)
    assert bits_equal(f0, f1).all() and s0.rays_any == s1.rays_any


def test_option_and_scene_limits(oracle_mod):
    """The largest values the boundary accepts - 8 lights (directional and point), -spr 15 (16 light rays per hit and
    light, 16 children per reflective hit), -spp 16 (256 samples per pixel) - against the oracle, and the status codes
    one step beyond them."""
    from raytracer_amd.scene import Light
    rng = np.random.RandomState(3)
    sc = scenes.bunny(phong_surroundings=True)
    sc.lights = []
    for k in range(8):
        v = rng.normal(size=3).astype(np.float32)
        v[1] = abs(v[1]) + 0.3
        if k % 2 == 0:
            sc.lights.append(Light(0, (v / np.linalg.norm(v)).astype(np.float32), np.full(3, 0.2, np.float32)))
        else:
            sc.lights.append(Light(1, (4 * v).astype(np.float32), np.full(3, 0.2, np.float32)))
    g, o = _pair(sc, oracle_mod)
    for kw in (dict(width=24, height=16, spp=2, spr=15, max_bounce=1), dict(width=6, height=4, spp=16, spr=1, max_bounce=1)):
        opt = Options(hfov=60, eps=1e-3, jitter_mode=RT_JITTER_HASH, seed=2, exact_boxes=1, **kw)
        a, sa = o.render_rect(opt, 0, 0, opt.width, opt.height)
        b, sb = g.render_rect(opt, 0, 0, opt.width, opt.height)
        assert bits_equal(a, b).all()
        assert sb.ref_get_hit_calls == sa.rays_closest + sa.rays_any
    opt = Options(width=8, height=8, spp=2, spr=16)
    with pytest.raises(api.RtError):
        g.render_rect(opt, 0, 0, 8, 8)
    opt = Options(width=8, height=8, spp=17, spr=1)
    with pytest.raises(api.RtError):
        g.render_rect(opt, 0, 0, 8, 8)
    sc.lights.append(sc.lights[0])
    with pytest.raises(api.RtError):
        api.TracerScene(sc)


def test_rect_and_band_invariance(bunny_scene):
    """A frame rendered as one rectangle, as tiles, and as bands is the same frame; two runs are bit-identical."""
    g = api.TracerScene(scenes.bunny(phong_surroundings=True))
    opt = Options(width=160, height=90, hfov=60, spp=2, spr=3, max_bounce=2, eps=1e-3, jitter_mode=RT_JITTER_HASH, seed=9)
    full, _ = g.render_rect(opt, 0, 0, 160, 90)
    again, _ = g.render_rect(opt, 0, 0, 160, 90)
    assert bits_equal(full, again).all()
    tiles = np.empty_like(full)
    for (x0, y0, x1, y1) in [(0, 0, 77, 40), (77, 0, 160, 40), (0, 40, 160, 41), (0, 41, 5, 90), (5, 41, 160, 90)]:
        part, _ = g.render_rect(opt, x0, y0, x1, y1)
        tiles[y0:y1, x0:x1] = part
    assert bits_equal(full, tiles).all()
    empty, _ = g.render_rect(opt, 10, 10, 10, 30)
    assert empty.shape == (20, 0, 3)


def test_bands_that_outgrow_the_level_limit_are_split():
    """A scene with mirrors whose deeper levels exceed the item limit of a level: the band is rendered again at half its
    height until its levels fit; the frame, the ray counts and the reference-equivalent call count are those of the
    unsplit frame.  A limit that even a four-row band exceeds is reported."""
    g = api.TracerScene(scenes.mirror_room())
    opt = Options(width=192, height=108, hfov=60, spp=2, spr=3, max_bounce=3, eps=1e-3, jitter_mode=RT_JITTER_HASH, seed=4)
    full, st = g.render_rect(opt, 0, 0, 192, 108, stats=True)
    widest = st.rays_closest                       # far more than any single level of a split band may now hold
    for limit in (widest // 3, widest // 8, widest // 16):
        g.set_level_limit(limit)
        part, st2 = g.render_rect(opt, 0, 0, 192, 108, stats=True)
        assert bits_equal(full, part).all(), limit
        assert (st2.rays_closest, st2.rays_any, st2.ref_get_hit_calls) == (st.rays_closest, st.rays_any, st.ref_get_hit_calls)
    g.set_level_limit(8)
    with pytest.raises(RuntimeError, match="limit of light-ray items"):
        g.render_rect(opt, 0, 0, 192, 108)
    g.set_level_limit(0)
    again, _ = g.render_rect(opt, 0, 0, 192, 108)
    assert bits_equal(full, again).all()
    with pytest.raises(RuntimeError):
        g.set_level_limit(-1)


def test_many_meshes_mixed_materials(oracle_mod):
    """An instanced field (61 meshes, 200 k triangles, Lambert / Phong / refractive at random): the top tree over
    mesh boxes answers like the reference's loop over meshes, and a small frame matches within the refractive bar.
    Half of the rays pass exactly through mesh vertices: there the reference's own box tests decide by the last bit
    whether a leaf is looked at, so only the reference's slab arithmetic (exact_boxes) can agree on every ray; the
    pruning traversal widens its box tests instead and must never return a farther hit than the reference."""
    sc = scenes.bunny_field(40, seed=11)
    g, o = _pair(sc, oracle_mod)
    assert g.info()["tris"] == sc.n_tris
    P = np.concatenate([m.pos.reshape(-1, 3)[::50] for m in sc.meshes])
    c = P.mean(0)
    org, d = random_rays(20000, 21, center=c, spread=30.0)
    rng = np.random.RandomState(4)
    k = len(org) // 2
    tgt = P[rng.randint(len(P), size=k)]
    dd = tgt - org[-k:]
    d[-k:] = (dd / np.linalg.norm(dd, axis=1, keepdims=True)).astype(np.float32)
    to, trio, nro = o.trace(org, d, 1e-3, np.inf)
    for exact in (True, False):
        tg, trig, nrg = g.trace(org, d, 1e-3, np.inf, exact_boxes=exact)
        same = trio == trig
        assert bits_equal(to[same], tg[same]).all() and bits_equal(nro[same], nrg[same]).all()
        if exact:
            assert same.all()
            continue
        assert same[:-k].mean() >= 0.9999 and same[-k:].mean() >= 0.99
        t_ref = np.where(trio >= 0, to, np.inf)
        t_gpu = np.where(trig >= 0, tg, np.inf)
        assert (t_gpu <= t_ref * (1 + 1e-5)).all()
    assert (trio >= 0).mean() > 0.3
    opt = Options(width=64, height=36, hfov=54.5, spp=2, spr=2, max_bounce=2, max_refrac_bounce=3, eps=1e-3,
                  jitter_mode=RT_JITTER_HASH, seed=3, exact_boxes=1)
    a, _ = o.render_rect(opt, 8, 4, 56, 30)
    b, _ = g.render_rect(opt, 8, 4, 56, 30)
    assert (np.isnan(a) == np.isnan(b)).all()
    fin = np.isfinite(a)
    assert np.allclose(a[fin], b[fin], rtol=1e-5, atol=1e-6)


def test_interleaved_row_shares_reassemble_the_frame():
    """What N ranks would render (each its interleaved bands, one call per rank) is the single-GPU frame."""
    from raytracer_amd.distributed import local_to_image_rows
    g = api.TracerScene(scenes.bunny())
    W, H, band = 96, 54, 4
    base = dict(width=W, height=H, hfov=60, spp=2, spr=3, max_bounce=3, jitter_mode=RT_JITTER_HASH, seed=2)
    full, _ = g.render_rect(Options(**base), 0, 0, W, H)
    for world in (2, 3, 8):
        frame = np.full_like(full, -1.0)
        for rank in range(world):
            rows = local_to_image_rows(H, world, rank, band)
            part, _ = g.render_rect(Options(row_stride=world, row_band=band, row_phase=rank, **base), 0, 0, W, len(rows),
                                    stats=False)
            frame[rows] = part
        assert bits_equal(full, frame).all()
    with pytest.raises(api.RtError):       # one local row too many
        n = len(local_to_image_rows(H, 2, 1, band))
        g.render_rect(Options(row_stride=2, row_band=band, row_phase=1, **base), 0, 0, W, n + 1)


def test_bad_arguments_are_reported():
    g = api.TracerScene(scenes.bunny())
    opt = Options(width=64, height=36)
    with pytest.raises(api.RtError):
        g.render_rect(opt, 0, 0, 65, 36)
    with pytest.raises(api.RtError):
        g.render_rect(Options(width=64, height=36, spp=0), 0, 0, 64, 36)
    sc = scenes.bunny()
    sc.meshes[0].material = 7
    with pytest.raises(Exception):
        api.TracerScene(sc)


def test_full_size_frame_properties(oracle_mod, bunny_scene):
    """1920x1080, 4 spp, 3 bounces (the headline configuration): checked through properties that need no full CPU
    frame - a strided subset of pixels against the oracle, the device-resident output path against the host path,
    and the 8-bit conversion against the oracle's."""
    import torch
    g, o = _pair(bunny_scene, oracle_mod)
    opt = Options(width=1920, height=1080, hfov=60, spp=2, spr=3, max_bounce=3, jitter_mode=RT_JITTER_OFF)
    out = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda")
    _, st = g.render_rect(opt, 0, 0, 1920, 1080, out=out)
    torch.cuda.synchronize()
    frame = out.cpu().numpy()
    host, _ = g.render_rect(opt, 0, 0, 1920, 1080)
    assert bits_equal(frame, host).all()
    assert st.rays_closest == 1920 * 1080 * 4 and st.levels == 1
    hit = np.isfinite(frame[..., 0]).mean()
    assert 0.6 < hit < 0.7                       # about two thirds of the pixels see geometry
    bad = 0
    n = 0
    for y in range(7, 1080, 97):
        row, _ = o.render_rect(opt, 0, y, 1920, y + 1)
        bad += (~bits_equal(row[0], frame[y])).any(axis=-1).sum()
        n += 1920
    assert bad / n <= 1e-4
    u8 = api.resolve_u8(frame)
    assert (u8 == oracle_mod.resolve_u8(frame)).all()
    u8d = torch.empty((1080, 1920, 3), dtype=torch.uint8, device="cuda")
    api.resolve_u8(out, out=u8d)
    torch.cuda.synchronize()
    assert (u8d.cpu().numpy() == u8).all()


def test_headless_frame_driver(oracle_mod, tmp_path):
    """`python -m raytracer_amd.render` end to end: option file in, image out, the bytes are the oracle's bytes; block by
    block rendering gives the same image."""
    import zlib
    from raytracer_amd import render
    scene_path = "%s/stanford_bunny.rtscene" % GOLDEN
    optfile = tmp_path / "options.txt"
    optfile.write_text("-dispw 96\n-disph 54\n-hfov 60\n-spp 2\n-spr 3\n-bounce 3\n-eps 0.0001\n-reps 0.1\n-mpbs 40\n")
    out1, out2 = str(tmp_path / "a.png"), str(tmp_path / "b.ppm")
    assert render.main([scene_path, str(optfile), "-o", out1, "--seed", "7", "--exact", "--log", str(tmp_path / "logs.txt")]) == 0
    assert render.main([scene_path, str(optfile), "-o", out2, "--seed", "7", "--exact", "--blocks"]) == 0
    sc = Scene.load(scene_path)
    opt = Options(width=96, height=54, hfov=60, spp=2, spr=3, max_bounce=3, eps=1e-4, ray_eps=0.1,
                  jitter_mode=RT_JITTER_HASH, seed=7, exact_boxes=1)
    a, _ = oracle_mod.OracleScene(sc).render_rect(opt, 0, 0, 96, 54)
    want = oracle_mod.resolve_u8(a)
    raw = open(out1, "rb").read()
    rows = zlib.decompress(raw[raw.index(b"IDAT") + 4: raw.index(b"IEND") - 8])
    png = np.frombuffer(rows, np.uint8).reshape(54, 1 + 96 * 3)[:, 1:].reshape(54, 96, 3)
    assert (png == want).all()
    ppm = np.frombuffer(open(out2, "rb").read()[-54 * 96 * 3:], np.uint8).reshape(54, 96, 3)
    assert (ppm == want).all()
    assert "Samples per pixel: 4" in open(str(tmp_path / "logs.txt")).read()
    # the reference's own output format (src/main.cpp:148-180): JPEG, quality 90, comment "Render Result"
    from PIL import Image
    out3 = str(tmp_path / "c.jpg")
    assert render.main([scene_path, str(optfile), "-o", out3, "--seed", "7", "--exact"]) == 0
    im = Image.open(out3)
    assert im.size == (96, 54) and im.info.get("comment") == b"Render Result"
    assert np.abs(np.asarray(im).astype(int) - want.astype(int)).mean() < 2.0


@pytest.mark.parametrize("name", ["bunny_lambert_1080p", "bunny_phong_1080p", "mirror_room", "glass_balls"])
def test_bench_workloads_both_box_modes_agree(name):
    """Every bench workload at a quarter of its size, rendered with the reference's slab arithmetic on the reference's
    trees (the mode the oracle pins bit for bit on the small frames above), with the pruning traversal on the
    surface-area-heuristic trees whose winners are checked against the reference's leaf boxes (the mode the bench times):
    the same frame bit for bit and the same rays; and with the unchecked pruning traversal: all but a few pixels per
    ten thousand, the 8-bit images within one step."""
    import bench
    make, kw = bench.WORKLOADS[name]
    kw = dict(kw, width=kw["width"] // 2, height=kw["height"] // 2)
    g = api.TracerScene(make())
    frames, stats = {}, {}
    for mode in (1, 0, 2):
        opt = Options(jitter_mode=RT_JITTER_HASH, seed=1, exact_boxes=mode, **kw)
        frames[mode], stats[mode] = g.render_rect(opt, 0, 0, opt.width, opt.height)
    assert bits_equal(frames[0], frames[1]).all()
    assert (stats[0].rays_closest, stats[0].rays_any, stats[0].ref_get_hit_calls) == (stats[1].rays_closest, stats[1].rays_any, stats[1].ref_get_hit_calls)
    a, b = frames[1], frames[2]
    assert (np.isnan(a) == np.isnan(b)).mean() >= 0.9999
    same = bits_equal(a, b).all(axis=-1)
    assert same.mean() >= 0.999
    u = np.abs(api.resolve_u8(a).astype(int) - api.resolve_u8(b).astype(int))
    assert (u > 1).mean() < 1e-3
    # which rays exist does not depend on the box mode except through those few pixels
    assert abs(stats[1].rays_closest - stats[2].rays_closest) <= 1e-3 * stats[1].rays_closest
    assert abs(stats[1].rays_any - stats[2].rays_any) <= 1e-3 * stats[1].rays_any


def test_ten_million_triangle_field_sample(oracle_mod):
    """The largest configuration (2013 instanced bunnies + ground, 10.0 M triangles, mixed materials): a random sample
    of rays against the oracle in both box modes, and a sparse strip of the 64-samples-per-pixel 4K frame (the two-pass
    replay on a deep scene) against the oracle's pixels."""
    sc = scenes.bunny_field()
    assert sc.n_tris > 10_000_000
    g, o = _pair(sc, oracle_mod)
    P = np.concatenate([m.pos.reshape(-1, 3)[::997] for m in sc.meshes])
    c = P.mean(0)
    ext = float(np.abs(P - c).max())
    org, d = random_rays(60000, 33, center=c + np.array([0, 0.3 * ext, 0], np.float32), spread=1.2 * ext)
    to, trio, nro = o.trace(org, d, 1e-3, np.inf)
    assert (trio >= 0).mean() > 0.2
    for exact in (True, False):
        tg, trig, nrg = g.trace(org, d, 1e-3, np.inf, exact_boxes=exact)
        same = trio == trig
        assert same.mean() >= (1.0 if exact else 0.9999)
        assert bits_equal(to[same], tg[same]).all() and bits_equal(nro[same], nrg[same]).all()
    opt = Options(width=3840, height=2160, hfov=54.5, spp=8, spr=3, max_bounce=3, max_refrac_bounce=3, eps=1e-3, ray_eps=0.1,
                  jitter_mode=RT_JITTER_HASH, seed=1, exact_boxes=1)
    a, _ = o.render_rect(opt, 1900, 700, 1912, 702)
    b, _ = g.render_rect(opt, 1900, 700, 1912, 702)
    assert (np.isnan(a) == np.isnan(b)).all()
    fin = np.isfinite(a)
    assert np.allclose(a[fin], b[fin], rtol=1e-5, atol=1e-6)
    # band invariance at the full width and sample count (the bench mode): the lower half of the 4K frame in the default
    # bands and in bands that a low level limit forces to be split several times over
    opt.exact_boxes = 0
    whole, s0 = g.render_rect(opt, 0, 0, 3840, 1080)
    g.set_level_limit(1 << 24)
    split, s1 = g.render_rect(opt, 0, 0, 3840, 1080)
    g.set_level_limit(0)
    assert bits_equal(whole, split).all()
    assert (s0.rays_closest, s0.rays_any, s0.ref_get_hit_calls) == (s1.rays_closest, s1.rays_any, s1.ref_get_hit_calls)
    assert s0.rays_closest > 3840 * 1080 * 64


@pytest.mark.parametrize("name,bit_exact", [("bunny_phong_1080p", True), ("mirror_room", True), ("glass_balls", False)])
def test_full_size_strips_of_the_other_workloads(oracle_mod, name, bit_exact):
    """Configurations two to four at their full size: two strips of each frame (where the oracle is affordable) against
    the oracle, with the reference's box arithmetic; bit for bit except through expf on the refractive scene."""
    import bench
    make, kw = bench.WORKLOADS[name]
    sc = make()
    g, o = _pair(sc, oracle_mod)
    opt = Options(jitter_mode=RT_JITTER_HASH, seed=1, exact_boxes=1, **kw)
    W, H = opt.width, opt.height
    for (x0, y0) in ((W // 2 - 20, H // 2), (W // 3, H // 4)):
        a, sa = o.render_rect(opt, x0, y0, x0 + 40, y0 + 2)
        b, sb = g.render_rect(opt, x0, y0, x0 + 40, y0 + 2)
        assert (np.isnan(a) == np.isnan(b)).all()
        if bit_exact:
            assert bits_equal(a, b).all()
        else:
            fin = np.isfinite(a)
            assert np.allclose(a[fin], b[fin], rtol=1e-5, atol=1e-6)
        assert sb.rays_any == sa.rays_any and 2 * sb.rays_closest == sa.rays_closest


def test_primary_hits_of_the_1080p_frame(oracle_mod, bunny_scene):
    """The bar of the target as it is worded: primary-hit triangle ids agree on at least 99.99 % of the rays and t within
    1e-5 relative - every second pixel centre of every second row of the 1920x1080 bunny frame, pruning traversal on the
    surface-area-heuristic tree against the oracle's exhaustive one."""
    sc = bunny_scene
    g, o = _pair(sc, oracle_mod)
    cam = sc.camera
    W, H = 1920, 1080
    a = cam.near * np.tan(3.1415926 * 60 / 360)
    xs, ys = np.arange(0, W, 2), np.arange(0, H, 2)
    u = (-a + 2 * a * (xs + 0.5) / W)[None, :].repeat(len(ys), 0)
    v = (-a * H / W + 2 * a * H / W * (ys + 0.5) / H)[:, None].repeat(len(xs), 1)
    d = u[..., None] * cam.side + v[..., None] * cam.up - cam.near * cam.w
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32).reshape(-1, 3)
    org = np.tile(cam.pos.astype(np.float32), (d.shape[0], 1))
    to, trio, _ = o.trace(org, d, 1e-4, np.inf)
    tg, trig, _ = g.trace(org, d, 1e-4, np.inf, exact_boxes=False)
    assert 0.6 < (trio >= 0).mean() < 0.7
    assert (trio == trig).mean() >= 0.9999
    both = (trio >= 0) & (trig >= 0)
    assert (np.abs(tg[both] - to[both]) <= 1e-5 * to[both]).all()


def test_one_process_drives_several_scene_handles():
    """rt_render_frame_multi: one host process fans a frame out over scene handles (one per GPU; here three handles on
    the one GPU of the test box, each on its own host thread and stream) and the bands land in the caller's frame, host
    or device memory: bit-identical to one whole-frame call."""
    import torch
    sc = scenes.bunny(phong_surroundings=True)
    gs = [api.TracerScene(sc) for _ in range(3)]
    opt = Options(width=200, height=117, hfov=60, spp=2, spr=3, max_bounce=2, eps=1e-3, jitter_mode=RT_JITTER_HASH, seed=11)
    whole, sw = gs[0].render_rect(opt, 0, 0, 200, 117)
    for band in (8, 5):
        got, sg = api.render_frame_multi(gs, opt, band_rows=band)
        assert bits_equal(whole, got).all()
        assert (sg.rays_closest, sg.rays_any) == (sw.rays_closest, sw.rays_any)
    dev = torch.full((117, 200, 3), -1.0, dtype=torch.float32, device="cuda")
    api.render_frame_multi(gs[:2], opt, band_rows=16, out=dev, stats=False)
    torch.cuda.synchronize()
    assert bits_equal(whole, dev.cpu().numpy()).all()
    with pytest.raises(api.RtError, match="twice"):
        api.render_frame_multi([gs[0], gs[0]], opt)


@pytest.mark.parametrize("name", ["bunny", "mirror_room", "field"])
def test_device_built_tree(oracle_mod, name):
    """The pruning traversal's tree built on the GPU (Morton keys, radix sort, parallel hierarchy, bottom-up boxes): its
    invariants hold after download (every triangle in one leaf of at most four, boxes nested, ranks a permutation), and
    traces and frames equal those of the host-built tree and the oracle - answers do not depend on the tree."""
    sc = {"bunny": lambda: scenes.bunny(), "mirror_room": lambda: scenes.mirror_room(6, sphere_res=8),
          "field": lambda: scenes.bunny_field(40, seed=11)}[name]()
    dev, host, o = api.TracerScene(sc, device_build=True), api.TracerScene(sc, device_build=False), oracle_mod.OracleScene(sc)
    info = dev.check_tree()
    assert info["built_on_device"] and info["tris"] == sc.n_tris == info["reached_tris"] and info["max_leaf"] <= 4
    assert not host.check_tree()["built_on_device"]
    P = np.concatenate([m.pos.reshape(-1, 3)[::7] for m in sc.meshes])
    c = P.mean(0)
    ext = float(np.abs(P - c).max())
    org, d = random_rays(30000, 5, center=c, spread=2.5 * ext)
    to, trio, nro = o.trace(org, d, 1e-3, np.inf)
    for g in (dev, host):
        tg, trig, nrg = g.trace(org, d, 1e-3, np.inf, exact_boxes=False)
        same = trio == trig
        assert same.mean() >= 0.9999
        assert bits_equal(to[same], tg[same]).all() and bits_equal(nro[same], nrg[same]).all()
        ag, _, _ = g.trace(org, d, 1e-3, 10000.0, any_hit=True, exact_boxes=False)
        ao, _, _ = o.trace(org, d, 1e-3, 10000.0, any_hit=True)
        assert (ao == ag).mean() >= 0.9999
    opt = Options(width=96, height=54, hfov=60, spp=2, spr=2, max_bounce=2, max_refrac_bounce=3, eps=1e-3,
                  jitter_mode=RT_JITTER_HASH, seed=3, exact_boxes=0)
    a, sa = host.render_rect(opt, 0, 0, 96, 54)
    b, sb = dev.render_rect(opt, 0, 0, 96, 54)
    assert (~bits_equal(a, b).all(axis=-1)).sum() <= 1
    assert abs(sa.rays_any - sb.rays_any) <= 1e-3 * sa.rays_any


def test_failed_device_build_falls_back_to_the_host_builder(oracle_mod, monkeypatch):
    """A device build that fails (no memory for it, a hierarchy deeper than the traversal stack) is not an error: the
    host builder takes over and the scene reports that its tree was not built on the device.  Many coincident triangles
    (equal sort keys, the deepest hierarchies) must build either way."""
    sc = scenes.mirror_room(4, sphere_res=6)
    tri = np.array([[[0.1, 0.2, 0.3], [0.4, 0.2, 0.3], [0.1, 0.6, 0.35]]], np.float32)
    from raytracer_amd.scene import Mesh
    sc.meshes.append(Mesh("pile", 0, np.repeat(tri, 3000, 0), np.tile(np.array([0, 0, 1], np.float32), (3000, 3, 1))))
    dev = api.TracerScene(sc, device_build=True)
    assert dev.check_tree()["reached_tris"] == sc.n_tris
    monkeypatch.setenv("RT_AMD_DEBUG_FAIL_DEVICE_BUILD", "1")
    fell = api.TracerScene(sc, device_build=True)
    monkeypatch.delenv("RT_AMD_DEBUG_FAIL_DEVICE_BUILD")
    info = fell.check_tree()
    assert not info["built_on_device"] and info["reached_tris"] == sc.n_tris
    o = oracle_mod.OracleScene(sc)
    P = np.concatenate([m.pos.reshape(-1, 3) for m in sc.meshes])
    org, d = random_rays(8000, 9, center=P.mean(0), spread=2.0 * float(np.abs(P - P.mean(0)).max()))
    to, trio, _ = o.trace(org, d, 1e-3, np.inf)
    for g in (dev, fell):
        tg, trig, _ = g.trace(org, d, 1e-3, np.inf, exact_boxes=False)
        same = trig == trio
        assert same.mean() >= 0.9999 and bits_equal(tg[same], to[same]).all()


def test_shallow_trees_take_the_lds_only_stack_and_change_nothing(monkeypatch):
    """Scenes whose pruning-traversal tree is no deeper than the LDS part of the traversal stack run kernels without
    the private-memory tier (every benchmark scene but the 10 M-triangle field).  The same scene created with
    RT_AMD_DEBUG_DEEP_STACK=1 takes the general kernels: frames and ray counts must be identical, and the depth the
    choice rests on is the depth of the tree as downloaded and re-walked."""
    for make, kw in ((lambda: scenes.bunny(phong_surroundings=True), dict(width=160, height=90, hfov=60, spp=2, spr=3, max_bounce=2, eps=1e-3)),
                     (lambda: scenes.glass_balls(10), dict(width=96, height=54, hfov=54.5, spp=2, spr=4, max_bounce=3, max_refrac_bounce=5, eps=1e-3))):
        sc = make()
        shallow = api.TracerScene(sc)
        assert shallow.check_tree()["depth"] <= 16
        monkeypatch.setenv("RT_AMD_DEBUG_DEEP_STACK", "1")
        deep = api.TracerScene(sc)
        monkeypatch.delenv("RT_AMD_DEBUG_DEEP_STACK")
        opt = Options(jitter_mode=RT_JITTER_HASH, seed=3, exact_boxes=0, **kw)
        a, sa = shallow.render_rect(opt, 0, 0, opt.width, opt.height)
        b, sb = deep.render_rect(opt, 0, 0, opt.width, opt.height)
        assert bits_equal(a, b).all()
        assert (sa.rays_closest, sa.rays_any, sa.ref_get_hit_calls) == (sb.rays_closest, sb.rays_any, sb.ref_get_hit_calls)


@pytest.mark.parametrize("name", ["bunny_phong", "mirror_room", "glass_balls", "field"])
def test_pruning_mode_returns_the_frame_of_the_reference_arithmetic(name, monkeypatch):
    """exact_boxes = 0, the mode that is timed, must give the frame of exact_boxes = 1 (which the other tests pin to the
    oracle bit for bit): a ray whose pruning-traversal winner lies in a leaf of the REFERENCE's tree whose box the ray
    misses under the reference's own slab arithmetic (src/AABBTree.cpp:133-160) is walked again with that test in the
    loop.  Frames, traced rays and reference-equivalent get_hit calls are equal; exact_boxes = 2 (no second walk) is
    allowed a few pixels; and a queue too small for the suspects makes the call fall back to the reference-arithmetic
    mode - still the same frame."""
    sc, kw = {"bunny_phong": lambda: (scenes.bunny(phong_surroundings=True), dict(width=480, height=270, hfov=60, spp=2, spr=3, max_bounce=2, eps=1e-3)),
              "mirror_room": lambda: (scenes.mirror_room(12, sphere_res=12), dict(width=320, height=180, hfov=60, spp=2, spr=2, max_bounce=3, eps=1e-3)),
              "glass_balls": lambda: (scenes.glass_balls(12), dict(width=256, height=144, hfov=54.5, spp=2, spr=3, max_bounce=3, max_refrac_bounce=5, eps=1e-3)),
              "field": lambda: (scenes.bunny_field(60, seed=5), dict(width=320, height=180, hfov=60, spp=2, spr=2, max_bounce=2, eps=1e-3))}[name]()
    g = api.TracerScene(sc)
    frames, stats = {}, {}
    for mode in (1, 0, 2):
        opt = Options(jitter_mode=RT_JITTER_HASH, seed=9, exact_boxes=mode, **kw)
        frames[mode], stats[mode] = g.render_rect(opt, 0, 0, opt.width, opt.height)
    assert bits_equal(frames[0], frames[1]).all()
    for field in ("rays_closest", "rays_any", "ref_get_hit_calls", "levels"):
        assert getattr(stats[0], field) == getattr(stats[1], field), field
    assert (~bits_equal(frames[2], frames[1]).all(axis=-1)).mean() <= 1e-4
    assert list(stats[1].reserved_) == [0, 0, 0] and list(stats[2].reserved_) == [0, 0, 0] and stats[0].reserved_[0] == 0
    redone = stats[0].reserved_[1] + stats[0].reserved_[2]
    print(name, "rays walked twice:", stats[0].reserved_[1], "+", stats[0].reserved_[2], "of", stats[0].rays_closest, "+", stats[0].rays_any)
    assert redone <= 1e-3 * (stats[0].rays_closest + stats[0].rays_any)
    # a batch of rays through the same three modes
    P = np.concatenate([m.pos.reshape(-1, 3)[::5] for m in sc.meshes])
    c = P.mean(0)
    org, d = random_rays(60000, 21, center=c, spread=2.5 * float(np.abs(P - c).max()))
    rng = np.random.RandomState(2)
    k = len(org) // 2                                           # half of them exactly at corners of triangles
    dd = P[rng.randint(len(P), size=k)] - org[-k:]
    d[-k:] = (dd / np.maximum(np.linalg.norm(dd, axis=1, keepdims=True), 1e-20)).astype(np.float32)
    t1, i1, n1 = g.trace(org, d, 1e-3, np.inf, exact_boxes=1)
    t0, i0, n0 = g.trace(org, d, 1e-3, np.inf, exact_boxes=0)
    assert (i0 == i1).all() and bits_equal(t0, t1).all() and bits_equal(n0, n1).all()
    a1, _, _ = g.trace(org, d, 1e-3, 1e4, any_hit=True, exact_boxes=1)
    a0, _, _ = g.trace(org, d, 1e-3, 1e4, any_hit=True, exact_boxes=0)
    assert (np.isfinite(a0) == np.isfinite(a1)).all()
    _, i2, _ = g.trace(org, d, 1e-3, np.inf, exact_boxes=2)
    assert (i2 == i1)[:-k].mean() >= 0.9999                     # (rays through a corner: any of its triangles may win there)
    with pytest.raises(api.RtError, match="exact_boxes"):
        g.trace(org[:4], d[:4], 1e-3, np.inf, exact_boxes=3)
    with pytest.raises(api.RtError, match="exact_boxes"):
        g.render_rect(Options(jitter_mode=RT_JITTER_HASH, seed=9, exact_boxes=-1, **kw), 0, 0, 8, 8)
    if redone:
        # a scene handle whose redo lists take one entry: the second suspect of a launch overflows them
        monkeypatch.setenv("RT_AMD_DEBUG_REDO_CAP", "1")
        small = api.TracerScene(sc)
        monkeypatch.delenv("RT_AMD_DEBUG_REDO_CAP")
        opt = Options(jitter_mode=RT_JITTER_HASH, seed=9, exact_boxes=0, **kw)
        f, st = small.render_rect(opt, 0, 0, opt.width, opt.height)
        assert bits_equal(f, frames[1]).all()
        assert st.reserved_[0] == 1 or st.reserved_[1] + st.reserved_[2] == redone
        if name == "glass_balls":                # two suspects in one launch of this frame
            assert st.reserved_[0] == 1
        assert (st.rays_closest, st.rays_any) == (stats[1].rays_closest, stats[1].rays_any)
        f2, st2 = small.render_rect(opt, 0, 0, opt.width, opt.height)          # and the handle is good for the next call
        assert bits_equal(f2, frames[1]).all()


def test_calls_on_one_handle_from_two_streams_do_not_race():
    """Asynchronous calls on ONE scene handle from different streams share the handle's queues: each call's stream first
    waits for the previous call's work (an event recorded at its end), so back-to-back frames with different options
    come out as if rendered one after the other."""
    import torch
    g = api.TracerScene(scenes.bunny())
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    optA = Options(width=640, height=360, hfov=60, spp=2, spr=3, max_bounce=3, jitter_mode=RT_JITTER_HASH, seed=1)
    optB = Options(width=512, height=300, hfov=50, spp=3, spr=2, max_bounce=3, jitter_mode=RT_JITTER_HASH, seed=2)
    wantA, _ = g.render_rect(optA, 0, 0, 640, 360)
    wantB, _ = g.render_rect(optB, 0, 0, 512, 300)
    for _ in range(3):
        a = torch.empty((360, 640, 3), dtype=torch.float32, device="cuda")
        b = torch.empty((300, 512, 3), dtype=torch.float32, device="cuda")
        g.render_rect(optA, 0, 0, 640, 360, out=a, stats=False, stream=s1.cuda_stream)     # returns while the GPU works
        g.render_rect(optB, 0, 0, 512, 300, out=b, stats=False, stream=s2.cuda_stream)
        g.render_rect(optA, 0, 0, 640, 360, out=a, stats=False, stream=s2.cuda_stream)
        torch.cuda.synchronize()
        assert bits_equal(wantA, a.cpu().numpy()).all() and bits_equal(wantB, b.cpu().numpy()).all()


@pytest.mark.parametrize("first", [0, 20, 40, 60])
def test_random_scenes(oracle_mod, first):
    """Twenty seeded random scenes per case (tests/fuzz_scenes.py: soups, lattice grids, zero-thickness sheets, needles,
    coincident triangles, every material and light kind, random options): frame, ray counts and a batch of rays - half
    of them aimed exactly at triangle corners - against the oracle in the reference-arithmetic mode, bit for bit; the
    pruning traversal on the host-built and on the device-built tree never returns a farther hit and never lets an
    occluded ray pass (tools/fuzz_parity.py states the two exceptions it sets aside)."""
    from tools.fuzz_parity import check_seed
    failed = {}
    for seed in range(first, first + 20):
        problems, _ = check_seed(seed, api, oracle_mod.OracleScene, np)
        if problems:
            failed[seed] = problems
    assert not failed


def test_random_scenes_wide_options(oracle_mod):
    """Random scenes again with the options pushed outwards (tests/fuzz_scenes.py::random_options_wide): up to 81
    samples per pixel - the fused replay, the flat one and the two-pass one all come up -, up to 16 light rays per hit
    and light, five bounces, eight lights."""
    from tools.fuzz_parity import check_seed
    failed = {}
    for seed in range(0, 30):
        problems, _ = check_seed(seed, api, oracle_mod.OracleScene, np, wide=True)
        if problems:
            failed[seed] = problems
    assert not failed


def test_big_random_scenes(oracle_mod):
    """The same checks on scenes of tens of thousands of triangles in clusters of very different density (deep,
    lopsided trees; tests/fuzz_scenes.py::random_big_scene)."""
    from tools.fuzz_parity import check_seed
    failed = {}
    for seed in (0, 2, 5):
        problems, _ = check_seed(seed, api, oracle_mod.OracleScene, np, big=True)
        if problems:
            failed[seed] = problems
    assert not failed
