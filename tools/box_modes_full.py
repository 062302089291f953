"""The bench mode (pruning traversal) against the reference-arithmetic mode of the same library at FULL size, every
workload the oracle pins in that mode: pixels whose float value differs at all, 8-bit values that differ by more than one,
and the difference in reference-equivalent get_hit calls.  usage: python tools/box_modes_full.py [workload ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import bench  # noqa: E402
from raytracer_amd import api  # noqa: E402
from raytracer_amd._abi import Options, RT_JITTER_HASH  # noqa: E402

api.init(0)
names = sys.argv[1:] or ["bunny_lambert_1080p", "bunny_lambert_2lights_1080p", "bunny_phong_1080p", "mirror_room", "glass_balls"]
for name in names:
    make, kw = bench.WORKLOADS[name]
    g = api.TracerScene(make())
    frames, stats = [], []
    for exact in (1, 0):
        opt = Options(jitter_mode=RT_JITTER_HASH, seed=1, exact_boxes=exact, **kw)
        f, st = g.render_rect(opt, 0, 0, opt.width, opt.height)
        frames.append(f); stats.append(st)
    a, b = frames
    same = ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all(axis=-1)
    u = np.abs(api.resolve_u8(a).astype(int) - api.resolve_u8(b).astype(int))
    print("%-28s pixels that differ %d of %d; 8-bit values off by more than one %d; get_hit calls %d vs %d (diff %d)" % (
        name, int((~same).sum()), same.size, int((u > 1).sum()), stats[0].ref_get_hit_calls, stats[1].ref_get_hit_calls,
        stats[1].ref_get_hit_calls - stats[0].ref_get_hit_calls), flush=True)
    g.close()
