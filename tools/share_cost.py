"""What one rank's share of a frame costs on ONE GPU, without the exchange: for N = 1, 2, 4, 8 renders rank 0's share of
an N-way split (every N-th band of 8 rows, one rt_render_rect call + rt_resolve_u8 per frame) and prints ms per frame and
the ratio to an ideal 1/N - the ceiling of the strong-scaling efficiency that the per-frame fixed costs (launches, the
tails of small grids) allow before any gather.   usage: python tools/share_cost.py [workload] [frames]"""
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
    from raytracer_amd.distributed import local_to_image_rows
    wl = sys.argv[1] if len(sys.argv) > 1 else "bench_default"
    wl = "bunny_lambert_1080p" if wl == "bench_default" else wl
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    make, kw = bench.WORKLOADS[wl]
    scene = make()
    api.init(0)
    n_streams = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    tracers = [api.TracerScene(scene) for _ in range(n_streams)]
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.current_stream()] if n_streams == 1 else [torch.cuda.Stream() for _ in range(n_streams)]
    t1 = None
    for n in (1, 2, 4, 8):
        opt = Options(jitter_mode=RT_JITTER_HASH, seed=1, exact_boxes=0, **kw)
        W, H = opt.width, opt.height
        rows = len(local_to_image_rows(H, n, 0, bench.BAND_ROWS)) if n > 1 else H
        if n > 1:
            opt.row_stride, opt.row_band, opt.row_phase = n, bench.BAND_ROWS, 0
        rgb = [torch.empty((rows, W, 3), dtype=torch.float32, device=dev) for _ in range(n_streams)]
        u8 = [torch.empty((rows, W, 3), dtype=torch.uint8, device=dev) for _ in range(n_streams)]

        def frame(k):
            b = k % n_streams
            tracers[b].render_rect(opt, 0, 0, W, rows, out=rgb[b], stats=False, stream=streams[b].cuda_stream)
            api.resolve_u8(rgb[b], out=u8[b], stream=streams[b].cuda_stream)
        for k in range(6):
            frame(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(frames):
            frame(k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / frames * 1e3
        t1 = ms if n == 1 else t1
        print("%d stream(s)  %s  1/%d of the frame (%4d rows): %.4f ms per frame, ideal %.4f, ceiling of efficiency %.2f" % (n_streams, wl, n, rows, ms, t1 / n, t1 / n / ms), flush=True)


if __name__ == "__main__":
    main()
