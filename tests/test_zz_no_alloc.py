"""rt_render_rect_noalloc (the entry point that allocates nothing: its ray queues live in a fixed per-GPU array) against
rt_render_rect, through the C-ABI, bit for bit; two calls in a row, the second with a smaller input, so that what the
first call left in the shared array cannot leak into the second."""
import numpy as np
import pytest

from raytracer_amd import api, scenes
from raytracer_amd._abi import Options, RT_JITTER_HASH

pytestmark = pytest.mark.gpu


def _same(a, b):
    a, b = a.cpu().numpy(), b.cpu().numpy()
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


def test_noalloc_render_matches_render_rect():
    import torch
    api.init(0)
    g = api.TracerScene(scenes.bunny(second_light=True))
    for exact in (0, 1):
        # first call: the headline frame (rendered in several bands of the fixed array); second: a small off-centre
        # rectangle of a smaller image with other sampling settings
        for (W, H, rect, kw) in ((1920, 1080, (0, 0, 1920, 1080), dict(spp=2, spr=3)),
                                 (333, 190, (21, 10, 301, 177), dict(spp=3, spr=2))):
            opt = Options(width=W, height=H, hfov=60, max_bounce=3, jitter_mode=RT_JITTER_HASH, seed=4, exact_boxes=exact, **kw)
            x0, y0, x1, y1 = rect
            want = torch.full((y1 - y0, x1 - x0, 3), -7.0, dtype=torch.float32, device="cuda")
            got = torch.full((y1 - y0, x1 - x0, 3), -9.0, dtype=torch.float32, device="cuda")
            _, sw = g.render_rect(opt, x0, y0, x1, y1, out=want)
            _, sg = g.render_rect(opt, x0, y0, x1, y1, out=got, noalloc=True)
            torch.cuda.synchronize()
            assert _same(want, got)
            assert (sw.rays_closest, sw.rays_any) == (sg.rays_closest, sg.rays_any)
            # and without statistics (the fully asynchronous form)
            got2 = torch.full_like(got, -3.0)
            g.render_rect(opt, x0, y0, x1, y1, out=got2, stats=False, noalloc=True)
            torch.cuda.synchronize()
            assert _same(want, got2)


def test_noalloc_limits_are_reported():
    import torch
    api.init(0)
    out = torch.empty((8, 64, 3), dtype=torch.float32, device="cuda")
    opt = Options(width=64, height=8, spp=2, spr=1)
    with pytest.raises(api.RtError, match="reflective or refractive"):
        api.TracerScene(scenes.bunny(phong_surroundings=True)).render_rect(opt, 0, 0, 64, 8, out=out, noalloc=True)
    g = api.TracerScene(scenes.bunny())
    with pytest.raises(api.RtError, match="device memory"):
        g.render_rect(opt, 0, 0, 64, 8, out=np.empty((8, 64, 3), np.float32), noalloc=True)
    # 16 x 16 samples per pixel on a 20000-pixel-wide row of tiles: 20.5 M samples per band, beyond the 64 MiB array
    wide = Options(width=20000, height=4, spp=16, spr=0)
    big = torch.empty((4, 20000, 3), dtype=torch.float32, device="cuda")
    with pytest.raises(api.RtError, match="too wide"):
        g.render_rect(wide, 0, 0, 20000, 4, out=big, noalloc=True)
