"""Prints the rays on which a seed of tools/fuzz_parity.py disagrees.  usage: python tools/fuzz_debug.py <seed> [...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch  # noqa: F401
    from raytracer_amd import api
    from oracle.oracle import OracleScene
    from tests.fuzz_scenes import random_scene, random_options
    from tests.conftest import bits_equal, random_rays
    api.init(0)
    np.set_printoptions(precision=9, floatmode="maxprec", linewidth=200)
    for seed in [int(a) for a in sys.argv[1:]]:
        sc = random_scene(seed)
        o, g = OracleScene(sc), api.TracerScene(sc, device_build=False)
        P = np.concatenate([m.pos.reshape(-1, 3) for m in sc.meshes])
        T = np.concatenate([m.pos for m in sc.meshes])
        c0 = P.mean(0)
        ext = float(np.abs(P - c0).max()) + 0.1
        org, d = random_rays(4000, seed, center=c0, spread=3.0 * ext)
        rng = np.random.RandomState(seed)
        k = len(org) // 2
        dd = P[rng.randint(len(P), size=k)] - org[-k:]
        d[-k:] = (dd / np.maximum(np.linalg.norm(dd, axis=1, keepdims=True), 1e-20)).astype(np.float32)
        to, trio, nro = o.trace(org, d, 1e-4, np.inf)
        for exact in (True, False):
            tg, trig, nrg = g.trace(org, d, 1e-4, np.inf, exact_boxes=exact)
            t_ref = np.where(trio >= 0, to, np.inf)
            t_gpu = np.where(trig >= 0, tg, np.inf)
            bad = (trig != trio) | ~bits_equal(tg, to) if exact else ~(t_gpu <= t_ref * (1 + 1e-5))
            for i in np.nonzero(bad)[0]:
                print("seed %d %s ray %d (%s): org %s dir %s" % (seed, "exact" if exact else "pruning", i, "corner" if i >= len(org) - k else "free", org[i], d[i]))
                print("   oracle tri %d t %r   gpu tri %d t %r" % (trio[i], float(to[i]), trig[i], float(tg[i])))
                for tri in {int(trio[i]), int(trig[i])} - {-1}:
                    print("   tri %d of %d: %s" % (tri, len(T), T[tri].tolist()))
        ao, _, _ = o.trace(org, d, 1e-4, 10000.0, any_hit=True)
        ag, _, _ = g.trace(org, d, 1e-4, 10000.0, any_hit=True, exact_boxes=True)
        for i in np.nonzero(ao != ag)[0]:
            print("seed %d any-hit ray %d (%s): org %s dir %s: oracle %r gpu %r; nearest: oracle tri %d t %r" % (seed, i, "corner" if i >= len(org) - k else "free", org[i], d[i], float(ao[i]), float(ag[i]), trio[i], float(to[i])))
            # which triangles does the reference's test let through on this ray (float32 Cramer, as the oracle does it)
            f = np.float32
            for ti, tri in enumerate(T):
                a_, b_, c_ = tri[0].astype(f), tri[1].astype(f), tri[2].astype(f)
                ab, ac, ao_ = a_ - b_, a_ - c_, a_ - org[i]
                dd = d[i]
                ei_hf = f(ac[1] * dd[2]) - f(dd[1] * ac[2]); gf_di = f(dd[0] * ac[2]) - f(ac[0] * dd[2]); dh_eg = f(ac[0] * dd[1]) - f(ac[1] * dd[0])
                m = f(f(ab[0] * ei_hf) + f(ab[1] * gf_di)) + f(ab[2] * dh_eg)
                if abs(m) < 1e-12:
                    print("   tri %d has determinant %r on this ray: %s" % (ti, float(m), tri.tolist()))
        opt = random_options(seed, 1)
        W, H = opt.width, opt.height
        a, sa = o.render_rect(opt, 0, 0, W, H)
        b, sb = g.render_rect(opt, 0, 0, W, H)
        print("seed %d frame %dx%d spp %d spr %d bounce %d refr bounce %d eps %g ray_eps %g jitter %d: oracle rays %d/%d gpu %d/%d"
              % (seed, W, H, opt.spp, opt.spr, opt.max_bounce, opt.max_refrac_bounce, opt.eps, opt.ray_eps, opt.jitter_mode,
                 sa.rays_closest // 2, sa.rays_any, sb.rays_closest, sb.rays_any))
        px = ~(np.isclose(a, b, rtol=1e-5, atol=1e-6) | (np.isnan(a) & np.isnan(b))).all(axis=-1)
        for y, x in zip(*np.nonzero(px)):
            one, s1 = o.render_rect(opt, x, y, x + 1, y + 1)
            two, s2 = g.render_rect(opt, x, y, x + 1, y + 1)
            print("   pixel (%d, %d): oracle %s gpu %s; alone: oracle %s rays %d/%d, gpu %s rays %d/%d"
                  % (x, y, a[y, x], b[y, x], one[0, 0], s1.rays_closest // 2, s1.rays_any, two[0, 0], s2.rays_closest, s2.rays_any))
        g.close()


if __name__ == "__main__":
    main()
