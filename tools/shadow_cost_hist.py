"""Where do the light rays of the bunny frame spend their node fetches?  Primary rays of a 480x270 frame, then one
any-hit ray per hit point toward the light (exact direction), counted; grouped by the mesh the ray starts on and by
the answer.  usage: python tools/shadow_cost_hist.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes  # noqa: E402

sc = scenes.bunny()
api.init(0)
g = api.TracerScene(sc)
W, H = 480, 270
cam = sc.camera
a = cam.near * np.tan(3.1415926 * 60 / 360)
u = (-a + 2 * a * (np.arange(W) + 0.5) / W)[None, :].repeat(H, 0)
v = (-a * H / W + 2 * a * H / W * (np.arange(H) + 0.5) / H)[:, None].repeat(W, 1)
d = u[..., None] * cam.side + v[..., None] * cam.up - cam.near * cam.w
d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32).reshape(-1, 3)
o = np.tile(cam.pos.astype(np.float32), (d.shape[0], 1))
t, tri, _ = g.trace(o, d, 1e-4, np.inf)
hit = tri >= 0
p = (o + d * t[:, None])[hit].astype(np.float32)
bounds = np.cumsum([0] + [m.n_tris for m in sc.meshes])
mesh = np.searchsorted(bounds, tri[hit], side="right") - 1
L = sc.lights[0].vec.astype(np.float32)
L = L / np.linalg.norm(L)
ld = np.tile(L, (p.shape[0], 1)).astype(np.float32)
ts, _, cn, ct = g.trace(p, ld, 1e-4, cam.far, any_hit=True, counted=True)
occl = np.isfinite(ts)
print("hit fraction %.3f" % hit.mean())
for m in range(len(sc.meshes)):
    for oc in (False, True):
        sel = (mesh == m) & (occl == oc)
        if sel.sum():
            print("%-8s %-9s rays %7d  nodes mean %6.1f p50 %4.0f p90 %4.0f max %4d   tris mean %5.1f" % (
                sc.meshes[m].name, "occluded" if oc else "lit", sel.sum(), cn[sel].mean(), np.percentile(cn[sel], 50),
                np.percentile(cn[sel], 90), cn[sel].max(), ct[sel].mean()))
print("all: nodes mean %.1f, share of node fetches by rays above 2x the mean: %.2f" % (cn.mean(), cn[cn > 2 * cn.mean()].sum() / cn.sum()))

# ---- what would sorting the cheap rays out buy?  Waves of 64 rays from 8x8 pixel blocks (all pixels, misses idle).
full = np.zeros(W * H, np.int64)
full[np.nonzero(hit)[0]] = cn
img = full.reshape(H, W)[: H // 8 * 8, : W // 8 * 8]
blocks = img.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
now = blocks.max(axis=1).sum()
print("wave steps now (max per wave): %d, lane steps %d, utilisation %.3f" % (now, blocks.sum(), blocks.sum() / (64.0 * now)))
for K in (3, 4, 6, 8):
    first = np.minimum(blocks, K).max(axis=1).sum()                       # everybody does at most K steps
    rest = blocks[blocks > K]                                             # the others start again, packed densely
    pad = (-len(rest)) % 64
    packed = np.concatenate([rest, np.zeros(pad, np.int64)]).reshape(-1, 64)
    second = packed.max(axis=1).sum()
    print("two passes, cut at %d steps: %d + %d = %d wave steps (%.2f of now); %.0f %% of the rays go to the second pass" % (
        K, first, second, first + second, (first + second) / now, 100.0 * len(rest) / max((blocks > 0).sum(), 1)))
srt = np.sort(blocks[blocks > 0])
pad = (-len(srt)) % 64
print("lower bound, rays sorted by cost: %.2f of now" % (np.concatenate([srt, np.zeros(pad, np.int64)]).reshape(-1, 64).max(axis=1).sum() / now))
