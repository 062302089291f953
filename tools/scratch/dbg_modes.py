import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from raytracer_amd import api, scenes
from tests.conftest import random_rays, bits_equal
from oracle.oracle import OracleScene
api.init(0)
sc = scenes.bunny(phong_surroundings=True)
g = api.TracerScene(sc)
P = np.concatenate([m.pos.reshape(-1, 3)[::5] for m in sc.meshes])
c = P.mean(0)
org, d = random_rays(60000, 21, center=c, spread=2.5 * float(np.abs(P - c).max()))
rng = np.random.RandomState(2)
k = len(org) // 2
dd = P[rng.randint(len(P), size=k)] - org[-k:]
d[-k:] = (dd / np.maximum(np.linalg.norm(dd, axis=1, keepdims=True), 1e-20)).astype(np.float32)
t1, i1, n1 = g.trace(org, d, 1e-3, np.inf, exact_boxes=1)
t0, i0, n0 = g.trace(org, d, 1e-3, np.inf, exact_boxes=0)
t2, i2, n2 = g.trace(org, d, 1e-3, np.inf, exact_boxes=2)
bad = np.nonzero((i0 != i1) | ~bits_equal(t0, t1))[0]
print("differ", len(bad), bad[:10])
o = OracleScene(sc)
to, io, no = o.trace(org[bad], d[bad], 1e-3, np.inf)
T = np.concatenate([m.pos for m in sc.meshes])
for j, b in enumerate(bad[:10]):
    print("ray", b, "o", org[b], "d", d[b])
    print("  mode1", i1[b], repr(float(t1[b])), " mode0", i0[b], repr(float(t0[b])), " mode2", i2[b], repr(float(t2[b])), " oracle", io[j], repr(float(to[j])))
    for nm, tri in (("mode1", i1[b]), ("mode0", i0[b])):
        if tri >= 0: print("  ", nm, "tri", T[tri].tolist())
    # distances reported for each of the two triangles alone
    for tri in {int(i1[b]), int(i0[b])}:
        if tri >= 0:
            from raytracer_amd.scene import Scene, Mesh
