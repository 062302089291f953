"""Scene-creation time and frame time of a workload with the pruning traversal's tree built on host threads and on the GPU.
usage: python tools/build_time.py [workload]   (RT_AMD_DEBUG_TREE=1 adds the library's own phase times)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from raytracer_amd import api  # noqa: E402
from raytracer_amd._abi import Options, RT_JITTER_HASH  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "bunny_field_4k"
make, kw = bench.WORKLOADS[wl]
api.init(0)
scene = make()
opt = Options(jitter_mode=RT_JITTER_HASH, seed=1, exact_boxes=0, **kw)
out = torch.empty((opt.height, opt.width, 3), dtype=torch.float32, device="cuda")
for dev in (True, False, True):
    t0 = time.perf_counter()
    tr = api.TracerScene(scene, device_build=dev)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    info = tr.check_tree() if "--check" in sys.argv else {}
    tr.render_rect(opt, 0, 0, opt.width, opt.height, out=out, stats=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, st = tr.render_rect(opt, 0, 0, opt.width, opt.height, out=out, stats=True)
    torch.cuda.synchronize()
    t_frame = time.perf_counter() - t0
    print("%s: %d triangles, tree on %s: scene creation %.2f s, frame %.1f ms (extend %.1f shadow %.1f replay %.1f) %s" % (
        wl, scene.n_tris, "device" if dev else "host  ", t_build, t_frame * 1e3, st.ms_extend, st.ms_shadow, st.ms_shade,
        {k: info[k] for k in ("depth", "leaves", "reached_nodes")} if info else ""), flush=True)
    tr.close()
