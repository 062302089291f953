"""One-off check at the BASELINE configurations' FULL sizes: the whole frame rendered by the CPU oracle (all host cores,
the reference's tile partitioning) against the HIP path in both box modes, bit by bit.  The test suite covers these
sizes through strips and strided rows only (a mirror-room frame takes the oracle minutes); this tool prints one line per
workload and mode for the record.   usage: python tools/full_frame_parity.py <workload> [...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch  # noqa: F401  (loads the HIP runtime the library shares)
    import bench
    from raytracer_amd import api
    from raytracer_amd._abi import Options, RT_JITTER_HASH
    from oracle.oracle import OracleScene, resolve_u8
    api.init(0)
    import threading

    def heartbeat():                      # the oracle can take minutes per frame; a silent run is taken for a hung one
        t0 = time.perf_counter()
        while True:
            time.sleep(30)
            print("  ... %d s" % (time.perf_counter() - t0), flush=True)
    threading.Thread(target=heartbeat, daemon=True).start()
    for wl in sys.argv[1:]:
        make, kw = bench.WORKLOADS[wl]
        scene = make()
        g, o = api.TracerScene(scene), OracleScene(scene)
        opt = Options(jitter_mode=RT_JITTER_HASH, seed=1, exact_boxes=1, **kw)
        W, H = opt.width, opt.height
        print("%s: oracle renders %dx%d ..." % (wl, W, H), flush=True)
        t0 = time.perf_counter()
        ref, so, _ = o.render_frame_mt(opt, 256)
        print("%s: oracle %.1f s, %d get_hit calls" % (wl, time.perf_counter() - t0, so.rays_closest + so.rays_any), flush=True)
        ref8 = resolve_u8(ref)
        for exact in (1, 0):
            opt.exact_boxes = exact
            img, st = g.render_rect(opt, 0, 0, W, H)
            same = (img.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(img) & np.isnan(ref))
            px = same.all(axis=2)
            fin = np.isfinite(img) & np.isfinite(ref)
            rel = np.abs(img[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-6)
            d8 = np.abs(api.resolve_u8(img).astype(int) - ref8.astype(int))
            print("%s: %s: %d of %d pixels bit-equal (%d differ), NaN pattern equal %s, max rel diff %.2e, 8-bit: max diff %d, "
                  "%d values differ by more than 1; reference-equivalent calls %d (oracle %d), %.1f ms"
                  % (wl, "reference arithmetic" if exact else "pruning traversal  ", int(px.sum()), px.size, int((~px).sum()),
                     bool((np.isnan(img) == np.isnan(ref)).all()), float(rel.max()) if rel.size else 0.0, int(d8.max()),
                     int((d8 > 1).sum()), st.ref_get_hit_calls, so.rays_closest + so.rays_any, st.ms_total), flush=True)
        g.close()


if __name__ == "__main__":
    main()
