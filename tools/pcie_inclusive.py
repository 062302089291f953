"""The bench frame with the framebuffer handed back in HOST memory: rt_render_rect into a numpy array (the library copies
the 24.9 MB of floats over PCIe), and the variant a host that only wants the image would use - render into HBM, convert to
8 bits on the device, copy 6.2 MB.   usage: python tools/pcie_inclusive.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    from raytracer_amd import api
    from raytracer_amd._abi import Options, RT_JITTER_HASH
    make, kw = bench.WORKLOADS["bunny_lambert_1080p"]
    api.init(0)
    tr = api.TracerScene(make())
    opt = Options(jitter_mode=RT_JITTER_HASH, seed=1, exact_boxes=0, **kw)
    W, H = opt.width, opt.height
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    d_u8 = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    h_rgb = np.empty((H, W, 3), np.float32)
    h_u8 = torch.empty((H, W, 3), dtype=torch.uint8).pin_memory()
    h_pin = torch.empty((H, W, 3), dtype=torch.float32).pin_memory().numpy()

    def in_hbm():
        tr.render_rect(opt, 0, 0, W, H, out=d_rgb, stats=False, stream=stream)
        api.resolve_u8(d_rgb, out=d_u8, stream=stream)

    def floats_to_host():
        tr.render_rect(opt, 0, 0, W, H, out=h_rgb, stats=False, stream=stream)

    def floats_to_pinned():
        tr.render_rect(opt, 0, 0, W, H, out=h_pin, stats=False, stream=stream)

    def bytes_to_host():
        in_hbm()
        h_u8.copy_(d_u8, non_blocking=True)

    for name, fn in (("framebuffer and 8-bit image left in HBM (what the bench times)", in_hbm),
                     ("float framebuffer returned in pageable host memory", floats_to_host),
                     ("float framebuffer returned in pinned host memory", floats_to_pinned),
                     ("8-bit image copied to pinned host memory", bytes_to_host)):
        for _ in range(40):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 100 * 1e3
        print("%-66s %.3f ms per frame, %.0f Mrays/s" % (name + ":", ms, 30050272 / ms / 1e3), flush=True)


if __name__ == "__main__":
    main()
