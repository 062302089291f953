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


def test_noalloc_allocates_nothing():
    """The property in the name: free device memory is the same before and after a hundred calls (rectangles of
    changing size, so a grow-on-demand buffer would show), while the same calls through rt_render_rect on a fresh handle
    do take memory."""
    import torch
    api.init(0)
    sc = scenes.bunny()
    g = api.TracerScene(sc)
    out = torch.empty((540, 960, 3), dtype=torch.float32, device="cuda")
    opt = Options(width=960, height=540, hfov=60, spp=2, spr=3, max_bounce=3, jitter_mode=RT_JITTER_HASH, seed=1)
    g.render_rect(opt, 0, 0, 64, 32, out=out[:32, :64].contiguous(), stats=False, noalloc=True)     # set-up happens here at the latest
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for k in range(100):
        h = 40 + 5 * k
        g.render_rect(opt, 0, 0, 960, h, out=out[:h], stats=False, noalloc=True)
    g.render_rect(opt, 0, 0, 960, 540, out=out, stats=True, noalloc=True)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    g2 = api.TracerScene(sc)
    free1 = torch.cuda.mem_get_info()[0]
    g2.render_rect(opt, 0, 0, 960, 540, out=out, stats=False)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] < free1       # the check can see an allocation when there is one


def test_noalloc_two_handles_two_streams_share_the_workspace_in_order():
    """All handles on a GPU share its one fixed array.  Two scene handles (different scenes) called back to back on two
    streams, without any synchronisation by the caller, must each get their own frame: the library orders the calls that
    use the array behind each other on the device."""
    import torch
    api.init(0)
    ga, gb = api.TracerScene(scenes.bunny()), api.TracerScene(scenes.bunny(second_light=True))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    optA = Options(width=1280, height=720, hfov=60, spp=2, spr=3, max_bounce=3, jitter_mode=RT_JITTER_HASH, seed=1)
    optB = Options(width=800, height=450, hfov=50, spp=3, spr=2, max_bounce=3, jitter_mode=RT_JITTER_HASH, seed=2)
    wantA = torch.empty((720, 1280, 3), dtype=torch.float32, device="cuda")
    wantB = torch.empty((450, 800, 3), dtype=torch.float32, device="cuda")
    ga.render_rect(optA, 0, 0, 1280, 720, out=wantA, stats=False)
    gb.render_rect(optB, 0, 0, 800, 450, out=wantB, stats=False)
    torch.cuda.synchronize()
    for _ in range(3):
        a = torch.full_like(wantA, -1.0)
        b = torch.full_like(wantB, -2.0)
        torch.cuda.synchronize()
        ga.render_rect(optA, 0, 0, 1280, 720, out=a, stats=False, stream=s1.cuda_stream, noalloc=True)
        gb.render_rect(optB, 0, 0, 800, 450, out=b, stats=False, stream=s2.cuda_stream, noalloc=True)
        ga.render_rect(optA, 0, 0, 1280, 720, out=a, stats=False, stream=s2.cuda_stream, noalloc=True)
        torch.cuda.synchronize()
        assert _same(wantA, a) and _same(wantB, b)


def test_noalloc_under_graph_capture():
    """The caller the entry point names first: a stream that is being captured into a HIP graph.  The call (without
    statistics: those need a host synchronisation) is recorded, not run; replaying the graph renders the frame, twice, bit
    for bit; and a plain call on the same handle afterwards still works (the library's ordering events were kept out
    of the capture)."""
    import torch
    api.init(0)
    g = api.TracerScene(scenes.bunny())
    opt = Options(width=640, height=360, hfov=60, spp=2, spr=3, max_bounce=3, jitter_mode=RT_JITTER_HASH, seed=3)
    want = torch.empty((360, 640, 3), dtype=torch.float32, device="cuda")
    g.render_rect(opt, 0, 0, 640, 360, out=want, stats=False)
    got = torch.full_like(want, -4.0)
    g.render_rect(opt, 0, 0, 640, 360, out=got, stats=False, noalloc=True)       # set-up outside the capture
    torch.cuda.synchronize()
    got.fill_(-4.0)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g.render_rect(opt, 0, 0, 640, 360, out=got, stats=False, stream=torch.cuda.current_stream().cuda_stream, noalloc=True)
    torch.cuda.synchronize()
    assert float(got[0, 0, 0]) == -4.0                   # captured, not executed
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        assert _same(want, got)
        got.fill_(-4.0)
    g.render_rect(opt, 0, 0, 640, 360, out=got, stats=False, noalloc=True)
    torch.cuda.synchronize()
    assert _same(want, got)
