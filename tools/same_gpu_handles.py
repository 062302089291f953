"""Does a frame of a scene that spawns rays get faster when k scene handles on the SAME GPU render interleaved bands of it
concurrently (rt_render_frame_multi, one host thread and stream per handle)?  A band of such a scene is a chain of
levels with a counter read-back after each; a second chain running beside it fills the GPU while the first waits for the
host or runs the tail of a kernel.   usage: python tools/same_gpu_handles.py <workload> [max handles] [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from raytracer_amd import api
    from raytracer_amd._abi import Options, RT_JITTER_HASH
    wl = sys.argv[1]
    kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    make, kw = bench.WORKLOADS[wl]
    scene = make()
    api.init(0)
    dev = torch.device("cuda", 0)
    opt = Options(jitter_mode=RT_JITTER_HASH, seed=1, exact_boxes=0, **kw)
    out = torch.empty((opt.height, opt.width, 3), dtype=torch.float32, device=dev)
    tracers = []
    ref = None
    for k in range(1, kmax + 1):
        tracers.append(api.TracerScene(scene))
        for bands in (8, 32):
            api.render_frame_multi(tracers, opt, bands, out=out, stats=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(frames):
                api.render_frame_multi(tracers, opt, bands, out=out, stats=False)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / frames * 1e3
            if ref is None:
                ref = out.clone()
            same = bool(((out.view(torch.int32) == ref.view(torch.int32)) | (torch.isnan(out) & torch.isnan(ref))).all())
            print("%s: %d handle(s), bands of %2d rows: %.2f ms per frame, frame %s" % (wl, k, bands, ms, "identical" if same else "DIFFERS"), flush=True)


if __name__ == "__main__":
    main()
