"""One-off check at full size: the 4K frame of the 10 M-triangle field rendered in the default bands and in bands forced
small by a low level limit is the same frame bit for bit, with the same ray counts.  usage: python tools/band_invariance_4k.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from raytracer_amd import api  # noqa: E402
from raytracer_amd._abi import Options, RT_JITTER_HASH  # noqa: E402

make, kw = bench.WORKLOADS["bunny_field_4k"]
t = api.TracerScene(make())
opt = Options(jitter_mode=RT_JITTER_HASH, seed=1, exact_boxes=0, **kw)
s = torch.cuda.current_stream().cuda_stream
frames, stats = [], []
for limit in (0, 1 << 24):
    t.set_level_limit(limit)
    out = torch.empty((opt.height, opt.width, 3), dtype=torch.float32, device="cuda")
    _, st = t.render_rect(opt, 0, 0, opt.width, opt.height, out=out, stats=True, stream=s)
    torch.cuda.synchronize()
    frames.append(out); stats.append(st)
    print("level limit %s: %.0f ms, %d levels, rays %d + %d, reference-equivalent get_hit calls %d"
          % (limit or "default", st.ms_total, st.levels, st.rays_closest, st.rays_any, st.ref_get_hit_calls), flush=True)
a, b = frames
same = (a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))
print("values that differ: %d of %d" % (int((~same).sum()), same.numel()))
print("counts equal:", all(getattr(stats[0], k) == getattr(stats[1], k) for k in ("rays_closest", "rays_any", "ref_get_hit_calls")))
