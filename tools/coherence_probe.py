"""How much would sorting the rays of a deep bounce level buy?  Builds level-2-like rays of a workload's scene with rt_trace
(primary rays through pixel centres -> mirror direction + jitter -> again), then times rt_trace on those rays and on
light rays from their hit points in three orders: as spawned (siblings adjacent, parents in pixel order), shuffled, and
sorted by (direction octant, Morton code of the origin).   usage: python tools/coherence_probe.py [workload] [jitter]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    from raytracer_amd import api
    wl = sys.argv[1] if len(sys.argv) > 1 else "bunny_phong_1080p"
    jit = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
    make, kw = bench.WORKLOADS[wl]
    scene = make()
    api.init(0)
    tr = api.TracerScene(scene)
    dev = torch.device("cuda", 0)
    lib = api.lib()
    stream = torch.cuda.current_stream().cuda_stream

    def trace(o, d, any_hit, reps=3):
        n = o.shape[0]
        t = torch.empty(n, dtype=torch.float32, device=dev); tri = torch.empty(n, dtype=torch.int32, device=dev)
        nrm = torch.empty((n, 3), dtype=torch.float32, device=dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(reps):
            a.record()
            rc = lib.rt_trace(tr._h, o.data_ptr(), d.data_ptr(), C.c_float(1e-3), C.c_float(float("inf")), n, int(any_hit), 0,
                              t.data_ptr(), tri.data_ptr(), nrm.data_ptr(), stream)
            api._check(rc, "rt_trace")
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        return t, tri, nrm, best

    def unit(v):
        return v / v.norm(dim=1, keepdim=True)

    cam = scene.camera
    W, H = kw["width"], kw["height"]
    a = cam.near * np.tan(3.1415926 * kw.get("hfov", 60) / 360)
    u = torch.linspace(-a, a, W, device=dev)[None, :].expand(H, W)
    v = torch.linspace(-a * H / W, a * H / W, H, device=dev)[:, None].expand(H, W)
    side, up, w = (torch.tensor(np.asarray(x, np.float32), device=dev) for x in (cam.side, cam.up, cam.w))
    d = unit((u[..., None] * side + v[..., None] * up - cam.near * w).reshape(-1, 3)).contiguous()
    o = torch.tensor(np.asarray(cam.pos, np.float32), device=dev).expand_as(d).contiguous()
    torch.manual_seed(1)
    for level in (1, 2):
        t, tri, n, _ = trace(o, d, False, 1)
        hit = tri >= 0
        p = (o + d * t[:, None])[hit]; n = n[hit]; d = d[hit]
        r = d - 2 * (d * n).sum(1, keepdim=True) * n
        o = p.repeat_interleave(4, 0).contiguous()
        d = unit(r.repeat_interleave(4, 0) + jit * (2 * torch.rand((o.shape[0], 3), device=dev) - 1)).contiguous()
        print("level %d: %d rays" % (level, o.shape[0]), flush=True)
    lo, hi = o.min(0).values, o.max(0).values

    def key(o, d):
        q = ((o - lo) / (hi - lo + 1e-9) * 31.999).to(torch.int64).clamp(0, 31)
        m = torch.zeros(o.shape[0], dtype=torch.int64, device=dev)
        for bit in range(5):
            for ax in range(3):
                m |= ((q[:, ax] >> bit) & 1) << (3 * bit + ax)
        octant = (d[:, 0] < 0).to(torch.int64) | ((d[:, 1] < 0).to(torch.int64) << 1) | ((d[:, 2] < 0).to(torch.int64) << 2)
        return (octant << 15) | m

    def orders(o, d):
        n = o.shape[0]
        yield "as spawned", torch.arange(n, device=dev)
        yield "shuffled", torch.randperm(n, device=dev)
        yield "sorted by (octant, origin cell)", torch.sort(key(o, d)).indices
        yield "sorted by origin cell only", torch.sort(key(o, d) & 0x7FFF).indices

    for name, idx in orders(o, d):
        _, _, _, ms = trace(o[idx].contiguous(), d[idx].contiguous(), False)
        print("nearest hit, %-34s %8.3f ms  %6.2f Grays/s" % (name + ":", ms, o.shape[0] / ms / 1e6), flush=True)
    # light rays from the hit points of those rays
    t, tri, n, _ = trace(o, d, False, 1)
    hit = tri >= 0
    p = (o + d * t[:, None])[hit]
    L = scene.lights[0]
    lv = torch.tensor(np.asarray(L.vec, np.float32), device=dev)
    base = unit(lv[None, :].expand_as(p)) if L.kind == 0 else unit(lv[None, :] - p)
    so = p.repeat_interleave(4, 0).contiguous()
    sd = unit(base.repeat_interleave(4, 0) + jit * (2 * torch.rand((so.shape[0], 3), device=dev) - 1)).contiguous()
    print("light rays: %d" % so.shape[0], flush=True)
    for name, idx in orders(so, sd):
        _, _, _, ms = trace(so[idx].contiguous(), sd[idx].contiguous(), True)
        print("any hit,     %-34s %8.3f ms  %6.2f Grays/s" % (name + ":", ms, so.shape[0] / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
