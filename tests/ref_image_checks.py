"""Checks of a tracer (CPU oracle or the HIP library) against the two bunny-scene result images the reference ships
(fixtures: tests/golden/ref_images.npz, written by tests/golden/make_ref_image_fixtures.py).

What the images pin, and what each would catch:
  * silhouette / horizon / dome outline: camera basis and frustum convention (reference src/render_thread.cpp:34-40,57-79,
    src/cameras.cpp:3-18), scene baking (node transforms), the nearest-hit search as a whole;
  * the rabbit's teal pixels: a left-right or up-down mirrored basis (the rest of the scene is nearly symmetric);
  * where image 1 is black on geometry: the sign of the light direction (src/main.cpp:369) and the light-visibility rays;
  * teal outside the rabbit in image 2: the mirror direction of reflect rays (src/ray.cpp:37-45).
A `tracer` is anything with trace(org, dir, t0, t1, any_hit=False) -> (t, tri, normal) over numpy arrays.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
W, H = 1920, 1080
HW, HH = W // 2, H // 2


def load_fixture():
    z = np.load(os.path.join(GOLDEN, "ref_images.npz"))
    fx = {k: np.unpackbits(z[k], axis=1)[:, :HW].astype(bool) for k in ("bright1", "bright2", "teal1", "teal2")}
    fx["luma1"] = z["luma1"]
    return fx


def centre_rays(cam, hfov=60.0, step=2, first=1):
    """Primary rays through the centres of pixels (first + step*x, first + step*y) of the 1920x1080 frame
    (src/render_thread.cpp:34-40,57-79 with the jitter terms at one half)."""
    a = cam.near * np.tan(3.1415926 * hfov / 360)
    xs, ys = np.arange(first, W, step), np.arange(first, H, step)
    u = (-a + 2 * a * (xs + 0.5) / W)[None, :].repeat(len(ys), 0)
    v = (-a * H / W + 2 * a * H / W * (ys + 0.5) / H)[:, None].repeat(len(xs), 1)
    d = u[..., None] * cam.side + v[..., None] * cam.up - cam.near * cam.w
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32).reshape(-1, 3)
    org = np.tile(cam.pos.astype(np.float32), (d.shape[0], 1))
    return org, d, (len(ys), len(xs))


def mesh_range(scene, name):
    off = np.cumsum([0] + [m.n_tris for m in scene.meshes])
    k = [m.name for m in scene.meshes].index(name)
    return off[k], off[k + 1]


def check_silhouette(hit, bright, within1=0.99):
    """hit: [540, 960] bool, nearest hit exists at the pixel centre.  bright: the image is not black there."""
    miss = ~hit
    assert (bright & miss).sum() <= 2e-3 * miss.sum()           # background is black (JPEG ringing at edges aside)
    assert (bright & miss).sum() <= 1e-3 * bright.sum()          # whatever is bright is geometry
    assert abs(hit.mean() - 0.655) < 0.01                        # about two thirds of the pixels see geometry
    # outline from above, per column: horizon of the ground, dome outline
    top_h = np.where(hit.any(0), HH - 1 - np.argmax(hit[::-1], axis=0), -1)
    top_b = np.where(bright.any(0), HH - 1 - np.argmax(bright[::-1], axis=0), -1)
    d = np.abs(top_h - top_b)
    assert (d <= 1).mean() >= within1 and (d <= 3).mean() >= 0.99 and d.max() <= 6
    # the flat part of the horizon (columns left of the dome) sits in the same row
    assert abs(np.median(top_h[:150]) - np.median(top_b[:150])) <= 1


def iou(a, b):
    return (a & b).sum() / max((a | b).sum(), 1)


def check_rabbit(rabbit, teal1, teal2):
    """rabbit: [540, 960] bool, the nearest hit is on the rabbit mesh."""
    assert (teal1 & ~rabbit).sum() <= 5e-3 * teal1.sum()         # image 1: teal only on the rabbit
    assert (teal1 & rabbit).sum() >= 0.85 * rabbit.sum()         # (the rest of the rabbit is in shadow)
    assert (rabbit & ~teal2).sum() <= 1e-2 * rabbit.sum()        # image 2: the whole rabbit is teal
    assert iou(teal1, rabbit) >= 0.85
    assert iou(teal1[:, ::-1], rabbit) <= 0.7 and iou(teal1[::-1], rabbit) <= 0.7   # a mirrored basis would show


def check_shadow_side(hit, lit, bright1):
    """lit: [540, 960] bool, the centre's hit point faces light 0 and its exact light ray reaches it.  With a second
    light in the image a pixel lit by light 0 is never black; black pixels on geometry are unlit by light 0."""
    dark = ~bright1 & hit
    assert dark.sum() > 0.03 * hit.sum()
    assert (dark & lit).sum() <= 0.015 * dark.sum()
    assert (lit & ~bright1).sum() <= 2e-3 * lit.sum()
    assert 0.7 < lit.sum() / hit.sum() < 0.82


def check_reflection(refl_rabbit, rabbit, teal2):
    """refl_rabbit: [540, 960] bool, the centre hits ground or dome and its exact mirror ray hits the rabbit."""
    tr = teal2 & ~rabbit
    assert tr.sum() > 10000
    assert (tr & ~refl_rabbit).sum() <= 5e-3 * tr.sum()
    assert iou(tr, refl_rabbit) >= 0.8


def check_ground_luma(frame, ground, luma, min_corr=0.90, max_resid=0.003):
    """frame: [270, 480, 3] float render of the shipped scene at the image's settings (-spp 2 -spr 3 -reps 0.1, jitter on);
    ground: [270, 480] bool, the pixel centre's nearest hit is on the ground mesh; luma: image 1's max(R, G, B) at those
    pixel centres.  The image was made with a second light and other colours than the committed scene file, so values do
    not compare - but the ground is one flat Lambert surface, so both pictures are, on it, an affine function of ONE
    thing: the fraction of light rays that reach light 0 (src/render_thread.cpp:146-165).  One scale and one offset fitted
    over all ground pixels must therefore leave nothing but JPEG noise: this pins the soft-shadow stage - where the
    penumbra lies and how wide the ray jitter of src/ray.cpp:47-61 makes it - against a picture the reference rendered.
    Returns (correlation, mean absolute residual of the fit)."""
    g = ground.copy()                                  # away from the ground's outline (JPEG ringing, mixed pixels)
    g[1:] &= ground[:-1]; g[:-1] &= ground[1:]; g[:, 1:] &= ground[:, :-1]; g[:, :-1] &= ground[:, 1:]
    mine = np.nan_to_num(frame).max(axis=2)[g].astype(np.float64)
    ref = luma[g].astype(np.float64) / 255.0
    A = np.stack([mine, np.ones_like(mine)], 1)
    co = np.linalg.lstsq(A, ref, rcond=None)[0]
    resid = float(np.abs(A @ co - ref).mean())
    corr = float(np.corrcoef(mine, ref)[0, 1])
    assert g.sum() > 50000 and co[0] > 0.2
    assert corr >= min_corr and resid <= max_resid, (corr, resid)
    return corr, resid


def ground_mask_480(tracer, scene):
    """[270, 480] bool: nearest hit of the centre of pixel (2 + 4x, 2 + 4y) of the 1920x1080 frame is on the ground."""
    org, d, shape = centre_rays(scene.camera, step=4, first=2)
    _, tri, _ = tracer.trace(org, d, 1e-4, np.inf)
    lo, hi = mesh_range(scene, "ground")
    return ((tri >= lo) & (tri < hi)).reshape(shape)


def run_all(tracer, scene, fx=None):
    """All of the above from primary, light and mirror rays traced by `tracer`."""
    fx = fx or load_fixture()
    cam = scene.camera
    org, d, shape = centre_rays(cam)
    t, tri, nrm = tracer.trace(org, d, 1e-4, np.inf)
    hit = tri >= 0
    lo, hi = mesh_range(scene, "rabbit")
    rabbit = (tri >= lo) & (tri < hi)
    check_silhouette(hit.reshape(shape), fx["bright1"])
    # image 2: the bright rim of the reflective dome rings further into the black sky under JPEG compression
    check_silhouette(hit.reshape(shape), fx["bright2"], within1=0.93)
    check_rabbit(rabbit.reshape(shape), fx["teal1"], fx["teal2"])
    p = (org + d * np.where(hit, t, 0)[:, None]).astype(np.float32)
    L = scene.lights[0].vec.astype(np.float32)
    ta, _, _ = tracer.trace(p, np.tile(L, (len(p), 1)), 1e-4, float(cam.far), any_hit=True)
    lit = hit & ~(ta == 0) & ((nrm * L).sum(1) > 0.05)
    check_shadow_side(hit.reshape(shape), lit.reshape(shape), fx["bright1"])
    dn = (d * nrm).sum(1, keepdims=True)
    r = np.where(hit[:, None], d - 2 * dn * nrm, np.array([0, 1, 0])).astype(np.float32)
    _, tri2, _ = tracer.trace(p, r, 1e-3, np.inf)
    refl = hit & ~rabbit & (tri2 >= lo) & (tri2 < hi)
    check_reflection(refl.reshape(shape), rabbit.reshape(shape), fx["teal2"])


def frame_mask_check(frame, fx=None):
    """A rendered 1920x1080 float frame (row 0 = bottom, NaN where nothing was drawn): its drawn/undrawn mask against
    image 1's silhouette."""
    fx = fx or load_fixture()
    drawn = np.isfinite(frame[..., 0])[1::2, 1::2]
    check_silhouette(drawn, fx["bright1"])
