import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from raytracer_amd import api
from tests.fuzz_scenes import random_scene
from tests.conftest import random_rays, bits_equal
from oracle.oracle import OracleScene
if len(sys.argv) > 2: api.LIB_PATH = os.path.abspath(sys.argv[2])
api.init(0)
seed = int(sys.argv[1])
sc = random_scene(seed)
o = OracleScene(sc); g = api.TracerScene(sc, device_build=False)
P = np.concatenate([m.pos.reshape(-1, 3) for m in sc.meshes])
c0 = P.mean(0); ext = float(np.abs(P - c0).max()) + 0.1
org, d = random_rays(4000, seed, center=c0, spread=3.0 * ext)
rng = np.random.RandomState(seed)
k = len(org) // 2
dd = P[rng.randint(len(P), size=k)] - org[-k:]
d[-k:] = (dd / np.maximum(np.linalg.norm(dd, axis=1, keepdims=True), 1e-20)).astype(np.float32)
to, trio, nro = o.trace(org, d, 1e-4, np.inf)
tg, trig, nrg = g.trace(org, d, 1e-4, np.inf, exact_boxes=True)
bad = np.nonzero((trig != trio) | ~bits_equal(tg, to) | ~bits_equal(nrg, nro).all(axis=-1))[0]
print("meshes", [(m.name, m.pos.shape[0]) for m in sc.meshes], "bad", bad)
T = np.concatenate([m.pos for m in sc.meshes])
for b in bad:
    print("ray", b, "o", org[b].tolist(), "d", d[b].tolist())
    print("  oracle tri", trio[b], "t", repr(float(to[b])), "n", nro[b].tolist())
    print("  gpu    tri", trig[b], "t", repr(float(tg[b])), "n", nrg[b].tolist())
    for tri in {int(trio[b]), int(trig[b])}:
        if tri >= 0: print("   tri", tri, T[tri].tolist())
    for mode in (0, 2):
        tm, im, _ = g.trace(org[b:b+1], d[b:b+1], 1e-4, np.inf, exact_boxes=mode)
        print("  mode", mode, im[0], repr(float(tm[0])))
