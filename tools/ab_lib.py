"""Runs the quick bench summary with alternative builds of the library (A/B of build-time constants).
usage: python tools/ab_lib.py <lib.so> [workload ...]   (each run is a child process: the library is loaded once per process)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.abspath(sys.argv[1])
for wl in sys.argv[2:] or ["bunny_lambert_1080p"]:
    code = ("import sys; sys.path.insert(0, %r); sys.argv=['bench.py','--no-cpu-baseline','--no-workloads','--workload',%r]; "
            "from raytracer_amd import api; api.LIB_PATH=%r; import bench; bench.main()" % (ROOT, wl, lib))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    if out.returncode != 0:
        print(out.stderr[-1500:]); sys.exit(1)
    d = json.loads(out.stdout.strip().splitlines()[-1]); r = d["roofline"]
    print(os.path.basename(lib), wl, "ms/frame", d["ms_per_step"], "| extend", r["ms_extend"], "shadow", r["ms_shadow"], "replay", r["ms_replay"],
          "| nodes/ray", r["nodes_per_ray"], "tris/ray", r["tris_per_ray"], flush=True)
