"""Differential run of the HIP path against the CPU oracle over seeded random scenes (tests/fuzz_scenes.py): frames in the
reference-arithmetic mode must be bit-equal (within 1e-5 where a refractive material is in use: device expf), the
ray counts equal; the pruning traversal - on the host-built and on the device-built tree - must give the oracle's frame
but for rays through shared edges and coincident triangles (reported, not judged here), and must never report a farther
hit than the oracle on a batch of rays (set aside and only noted: a hit of the oracle that lies outside its own needle
triangle, and rays that graze a triangle at less than 1e-3 - both explained where they are tested).  One line per failing seed, a summary at the end.
usage: python tools/fuzz_parity.py <first seed> <count> [big | wide]
(big: tens of thousands of triangles per scene; wide: up to 81 samples per pixel, 16 light rays, 5 bounces, 8 lights)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def record_less_candidates(T, o, d, t0):
    """How many triangles of T ([n][3][3] float32) pass the comparisons of the reference's Triangle::hit
    (src/geometry.cpp:268-295, same operations in float32) for the ray (o, d) with a distance that is not finite."""
    import numpy as np
    f = np.float32
    with np.errstate(all="ignore"):
        a_, b_, c_ = (T[:, 0, k] - T[:, 1, k] for k in range(3))
        d_, e_, f_ = (T[:, 0, k] - T[:, 2, k] for k in range(3))
        g_, h_, i_ = f(d[0]), f(d[1]), f(d[2])
        j_, k_, l_ = (T[:, 0, k] - f(o[k]) for k in range(3))
        ei_hf = e_ * i_ - h_ * f_; gf_di = g_ * f_ - d_ * i_; dh_eg = d_ * h_ - e_ * g_
        ak_jb = a_ * k_ - j_ * b_; jc_al = j_ * c_ - a_ * l_; bl_kc = b_ * l_ - k_ * c_
        m = a_ * ei_hf + b_ * gf_di + c_ * dh_eg
        t = (f_ * ak_jb + e_ * jc_al + d_ * bl_kc) / -m
        gamma = (i_ * ak_jb + h_ * jc_al + g_ * bl_kc) / m
        beta = (j_ * ei_hf + k_ * gf_di + l_ * dh_eg) / m
        passes = ~((t < t0) | (t > np.inf)) & ~((gamma < 0) | (gamma > 1)) & ~((beta < 0) | (beta > (f(1) - gamma)))
    return int((passes & ~np.isfinite(t)).sum())


def check_seed(seed, api, OracleScene, np, big=False, wide=False):
    from tests.fuzz_scenes import random_scene, random_big_scene, random_options, random_options_wide, more_lights, has_refraction
    from tests.conftest import bits_equal, random_rays
    sc = random_big_scene(seed) if big else random_scene(seed)
    if wide:
        sc = more_lights(sc, seed)
    problems, notes = [], []
    o = OracleScene(sc)
    g = api.TracerScene(sc, device_build=False)
    gd = api.TracerScene(sc, device_build=True)
    opt = random_options_wide(seed, 1) if wide else random_options(seed, 1)
    W, H = opt.width, opt.height
    a, sa = o.render_rect(opt, 0, 0, W, H)
    b, sb = g.render_rect(opt, 0, 0, W, H)
    refr = has_refraction(sc)
    frame = []
    if not (np.isnan(a) == np.isnan(b)).all():
        frame.append("exact: NaN pattern differs in %d values" % int((np.isnan(a) != np.isnan(b)).sum()))
    if refr:
        # The device's expf and glibc's differ in the last place.  Random materials and normals give Fresnel weights
        # outside [0, 1], a pixel is then a difference of large terms, and that last place grows: moving every expf of
        # the ORACLE by one unit in the last place moves such pixels by 1e-5 absolute (seeds 955 and 2257).  Beyond the
        # 1e-5 bar of the well-conditioned test scenes a difference is therefore noted up to 1e-4 and fails above.
        fin = np.isfinite(a) & np.isfinite(b)
        if not np.allclose(a[fin], b[fin], rtol=1e-4, atol=1e-4):
            frame.append("exact: refractive frame beyond 1e-4 (max abs %.3g)" % float(np.abs(a[fin] - b[fin]).max()))
        elif not np.allclose(a[fin], b[fin], rtol=1e-5, atol=1e-6):
            notes.append("exact: refractive frame within 1e-4 but beyond 1e-5 (max abs %.3g)" % float(np.abs(a[fin] - b[fin]).max()))
    elif not bits_equal(a, b).all():
        frame.append("exact: %d of %d pixels differ" % (int((~bits_equal(a, b).all(axis=-1)).sum()), W * H))
    # The reference-equivalent call count includes the light rays the reference sends from hits without a record, which
    # the GPU counts and does not trace; with no such hit the traced rays themselves are equal.
    if sb.ref_get_hit_calls != sa.rays_closest + sa.rays_any or (not sa.reserved_[0] and sb.rays_any != sa.rays_any):
        frame.append("exact: get_hit calls %d (traced %d + %d), oracle %d + %d" % (sb.ref_get_hit_calls, sb.rays_closest, sb.rays_any, sa.rays_closest, sa.rays_any))
    # reserved_[0] of the ORACLE's statistics: get_hit calls that returned true with a record nobody filled (a ray with
    # a NaN direction, a hit at t = +inf).  The reference's behaviour is undefined there; the oracle draws nothing for
    # such a ray and the HIP path does the same, so frames are compared all the same.
    problems += ["%s (%d hits without a record in the oracle's frame)" % (f, sa.reserved_[0]) if sa.reserved_[0] else f for f in frame]
    # the same frame whatever way it is asked for: idle light rays skipped, a part of the frame alone, interleaved rows
    opt.skip_unlit = 1
    b2, _ = g.render_rect(opt, 0, 0, W, H)
    opt.skip_unlit = 0
    if not bits_equal(b, b2).all():
        problems.append("skip_unlit changes %d pixels" % int((~bits_equal(b, b2).all(axis=-1)).sum()))
    x0, y0 = W // 3, H // 4
    part, _ = g.render_rect(opt, x0, y0, W - 1, H - 2)
    if not bits_equal(b[y0:H - 2, x0:W - 1], part).all():
        problems.append("a rectangle of the frame differs from the frame")
    from raytracer_amd.distributed import local_to_image_rows
    rows = local_to_image_rows(H, 3, 1, 2)
    if len(rows):
        opt.row_stride, opt.row_band, opt.row_phase = 3, 2, 1
        band, _ = g.render_rect(opt, 0, 0, W, len(rows), stats=False)
        opt.row_stride = opt.row_band = opt.row_phase = 0
        if not bits_equal(b[rows], band).all():
            problems.append("interleaved rows differ from the frame")
    if all(m.kind in (0, 1) for m in sc.materials):                          # nothing spawns rays: the entry point that
        import torch                                                          # allocates nothing takes the scene
        dev = torch.full((H, W, 3), -5.0, dtype=torch.float32, device="cuda")
        _, sn = g.render_rect(opt, 0, 0, W, H, out=dev, noalloc=True)
        torch.cuda.synchronize()
        if not bits_equal(b, dev.cpu().numpy()).all() or (sn.rays_closest, sn.rays_any) != (sb.rays_closest, sb.rays_any):
            problems.append("rt_render_rect_noalloc differs from rt_render_rect")
    opt.exact_boxes = 0
    for name, h in (("host tree", g), ("device tree", gd)):
        c, _ = h.render_rect(opt, 0, 0, W, H)
        fin = np.isfinite(a) & np.isfinite(c)
        px = ~(np.isclose(a, c, rtol=1e-5, atol=1e-6) | (np.isnan(a) & np.isnan(c))).all(axis=-1)
        if px.any():
            notes.append("pruning, %s: %d of %d pixels differ" % (name, int(px.sum()), W * H))
    # a batch of rays: ids, distances, the any-hit answer
    P = np.concatenate([m.pos.reshape(-1, 3) for m in sc.meshes])
    c0 = P.mean(0)
    ext = float(np.abs(P - c0).max()) + 0.1
    org, d = random_rays(4000, seed, center=c0, spread=3.0 * ext)
    rng = np.random.RandomState(seed)
    k = len(org) // 2                                           # half of them exactly at corners of triangles
    dd = P[rng.randint(len(P), size=k)] - org[-k:]
    d[-k:] = (dd / np.maximum(np.linalg.norm(dd, axis=1, keepdims=True), 1e-20)).astype(np.float32)
    to, trio, nro = o.trace(org, d, 1e-4, np.inf)
    ao, _, _ = o.trace(org, d, 1e-4, 10000.0, any_hit=True)
    tg, trig, nrg = g.trace(org, d, 1e-4, np.inf, exact_boxes=True)
    differ = (trig != trio) | ~bits_equal(tg, to) | ~bits_equal(nrg, nro).all(axis=-1)
    if differ.any():
        # One deviation is documented and set aside here: a triangle that passes the reference's test WITHOUT a finite
        # distance (0/0: the ray runs exactly inside the triangle's plane, in practice an axis-parallel ray along an
        # axis-aligned sheet).  The reference's pairwise merge (src/geometry.cpp:171-175, `left.t < right.t ? left : right`)
        # hands such a record upwards whenever it comes from the right, and get_hit then drops the whole mesh's answer
        # (src/render_thread.cpp:15); the flattened traversal ranks candidates by (t, rank) and keeps the finite hit.
        T = np.concatenate([m.pos for m in sc.meshes]).astype(np.float32)
        for i in np.nonzero(differ)[0]:
            if record_less_candidates(T, org[i], d[i], np.float32(1e-4)) > 0:
                differ[i] = False
                notes.append("exact trace: ray %d meets a triangle that passes the test without a finite distance (merge order of the reference)" % i)
    if differ.any():
        problems.append("exact trace: %d of %d rays differ" % (int(differ.sum()), len(org)))
    ag, _, _ = g.trace(org, d, 1e-4, 10000.0, any_hit=True, exact_boxes=True)
    if not (ag == ao).all():
        problems.append("exact any-hit: %d rays differ" % int((ag != ao).sum()))
    t_ref = np.where(trio >= 0, to, np.inf)
    for name, h in (("host tree", g), ("device tree", gd)):
        tp, trip, nrp = h.trace(org, d, 1e-4, np.inf, exact_boxes=False)
        t_gpu = np.where(trip >= 0, tp, np.inf)
        far = ~(t_gpu <= t_ref * (1 + 1e-5))
        if far.any():
            # The reference's Cramer test on a needle (determinant near zero) accepts points far outside the triangle;
            # which of those a tree lets through depends on its boxes.  Such a "hit" of the oracle is no farther-hit
            # error of the pruning traversal: tell it by the hit point lying outside the triangle's own box.
            T = np.concatenate([m.pos for m in sc.meshes]).astype(np.float64)
            for i in np.nonzero(far)[0]:
                o64, d64 = org[i].astype(np.float64), d[i].astype(np.float64)
                p = o64 + float(to[i]) * d64
                tri = T[trio[i]]
                slack = 1e-4 * (1.0 + np.abs(p).max())
                n = np.cross(tri[1] - tri[0], tri[2] - tri[0])
                graze = abs(np.dot(n, d64)) / max(np.linalg.norm(n) * np.linalg.norm(d64), 1e-300)
                if ((p < tri.min(0) - slack) | (p > tri.max(0) + slack)).any():
                    far[i] = False
                    notes.append("pruning trace, %s: ray %d - the oracle's hit lies outside its needle triangle" % (name, i))
                elif graze < 1e-3 and t_gpu[i] <= t_ref[i] * (1 + 1e-3):
                    # a ray almost inside the triangle's plane: the distance itself is only good to 1e-7 / graze, and
                    # two triangles that share the corner it passes through report distances that far apart
                    far[i] = False
                    notes.append("pruning trace, %s: ray %d grazes its triangle (sine %.1e), distances %.8g / %.8g" % (name, i, graze, t_ref[i], t_gpu[i]))
        if far.any():
            problems.append("pruning trace, %s: %d rays report a FARTHER hit than the oracle" % (name, int(far.sum())))
        same = trip == trio
        if not (bits_equal(tp[same], to[same]).all() and bits_equal(nrp[same], nro[same]).all()):
            problems.append("pruning trace, %s: same triangle, other distance or normal" % name)
        if same[:-k].mean() < 0.999:
            notes.append("pruning trace, %s: %d free rays differ in id" % (name, int((~same[:-k]).sum())))
        ap, _, _ = h.trace(org, d, 1e-4, 10000.0, any_hit=True, exact_boxes=False)
        miss = np.isfinite(ao) & ~np.isfinite(ap)
        # occluded for the oracle by something without a finite distance (the nearest-hit query finds nothing): a NaN t,
        # which only the reference's arithmetic on the reference's tree takes for an occluder
        ghost = miss & (trio < 0)
        if ghost.any():
            notes.append("pruning any-hit, %s: %d rays occluded for the oracle by a triangle without a finite distance" % (name, int(ghost.sum())))
        miss &= ~ghost
        if miss.any():
            problems.append("pruning any-hit, %s: %d rays the oracle finds occluded pass" % (name, int(miss.sum())))
    info = gd.check_tree()
    if info["reached_tris"] != sc.n_tris:
        problems.append("device tree reaches %d of %d triangles" % (info["reached_tris"], sc.n_tris))
    g.check_tree()
    for h in (g, gd):
        h.close()
    return problems, notes


def main():
    import numpy as np
    import torch  # noqa: F401  (loads the HIP runtime the library shares)
    from raytracer_amd import api
    from oracle.oracle import OracleScene
    api.init(0)
    first, count = int(sys.argv[1]), int(sys.argv[2])
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
    bad = noted = 0
    t0 = time.perf_counter()
    for seed in range(first, first + count):
        problems, notes = check_seed(seed, api, OracleScene, np, big, wide)
        for p in problems:
            print("seed %d: FAIL %s" % (seed, p), flush=True)
        for n in notes:
            print("seed %d: note %s" % (seed, n), flush=True)
        bad += bool(problems)
        noted += bool(notes)
        if (seed - first) % (2 if big else 20) == (1 if big else 19):
            print("... %d seeds, %.0f s" % (seed - first + 1, time.perf_counter() - t0), flush=True)
    print("%d seeds from %d: %d failed, %d with notes, %.0f s" % (count, first, bad, noted, time.perf_counter() - t0), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
