"""Lane utilisation of the traversal loops: runs one counted frame with the normal library (steps per lane) and one with
a build made with -DRT_COUNT_WAVE_STEPS (steps per wave), and prints  lane steps / (64 * wave steps)  per loop.
build the second library first:
  cd raytracer_amd/csrc && hipcc <Makefile flags> -DRT_COUNT_WAVE_STEPS -shared -pthread -o ../../tools/librt_wave.so <the Makefile's SRC>
usage: python tools/wave_util.py [workload]   (runs itself once per library, in child processes)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(libpath, workload):
    import torch
    from raytracer_amd import api
    api.LIB_PATH = libpath
    import bench
    from raytracer_amd._abi import Options, RT_JITTER_HASH
    make, kw = bench.WORKLOADS[workload]
    scene = make()
    opt = Options(jitter_mode=RT_JITTER_HASH, seed=1, exact_boxes=0, **kw)
    api.init(0)
    tracer = api.TracerScene(scene)
    opt.count_traversal = 1
    out = torch.empty((opt.height, opt.width, 3), dtype=torch.float32, device="cuda:0")
    _, st = tracer.render_rect(opt, 0, 0, opt.width, opt.height, out=out, stats=True)
    print(json.dumps({k: getattr(st, k) for k in ("rays_closest", "rays_any", "nodes_extend", "tris_extend", "nodes_shadow", "tris_shadow")}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3])
        sys.exit(0)
    wl = sys.argv[1] if len(sys.argv) > 1 else "bunny_lambert_1080p"
    res = []
    for lib in (os.path.join(ROOT, "raytracer_amd", "librt_amd.so"), os.path.join(ROOT, "tools", "librt_wave.so")):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, wl], capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            print(out.stderr[-2000:]); sys.exit(1)
        res.append(json.loads(out.stdout.strip().splitlines()[-1]))
    lane, wave = res
    print(wl, lane)
    for k in ("nodes_extend", "tris_extend", "nodes_shadow", "tris_shadow"):
        print("%-13s lane steps %12d  wave steps %11d  utilisation %.3f" % (k, lane[k], wave[k], lane[k] / (64.0 * max(wave[k], 1))))
