"""Host cost of one rt_render_rect call on a frame so small that the GPU is never the limit (what bounds the frame rate
of a rank that renders one eighth of a 1080p frame).  usage: python tools/host_overhead.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes  # noqa: E402
from raytracer_amd._abi import Options, RT_JITTER_HASH  # noqa: E402

api.init(0)
g = api.TracerScene(scenes.bunny())
for (w, h) in ((64, 36), (1920, 136)):
    opt = Options(width=1920, height=1080, hfov=60, spp=2, spr=3, max_bounce=3, jitter_mode=RT_JITTER_HASH, seed=1)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(20):
        g.render_rect(opt, 0, 0, w, h, out=out, stats=False, stream=s)
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        g.render_rect(opt, 0, 0, w, h, out=out, stats=False, stream=s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%dx%d: %.1f us per call to enqueue, %.1f us per call including the GPU" % (w, h, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
