"""Host-side logic that needs no GPU: scene files, option files, the C-ABI's exported symbols, band sharding."""
import ctypes
import os
import re

import numpy as np
import pytest

from raytracer_amd import api, scenes
from raytracer_amd._abi import MaterialDesc, LightDesc, CameraDesc, Options, Stats
from raytracer_amd.options import parse_options
from raytracer_amd.scene import Scene
from tests.conftest import GOLDEN, ROOT


def test_scene_roundtrip(tmp_path):
    sc = scenes.glass_balls(6)
    p = str(tmp_path / "s.rtscene")
    sc.save(p)
    sc2 = Scene.load(p)
    assert [m.name for m in sc2.meshes] == [m.name for m in sc.meshes]
    for a, b in zip(sc.meshes, sc2.meshes):
        assert a.material == b.material and (a.pos == b.pos).all() and (a.nrm == b.nrm).all()
    assert all((x.packed() == y.packed()).all() for x, y in zip(sc.materials, sc2.materials))
    assert all((x.packed() == y.packed()).all() for x, y in zip(sc.lights, sc2.lights))
    assert (sc.camera.packed() == sc2.camera.packed()).all()


def test_baked_reference_scenes_have_the_reference_triangle_counts():
    want = {"stanford_bunny": [200, 4968, 2696], "simple_knife": [27212], "unit_cube": [108], "unit_sphere": [760],
            "unit_cylinder": [240], "2020_plane_light": [200]}
    for name, counts in want.items():
        sc = Scene.load(os.path.join(GOLDEN, name + ".rtscene"))
        assert [m.n_tris for m in sc.meshes] == counts


def test_duplicate_mesh_names_rejected():
    sc = scenes.bunny()
    sc.meshes[1].name = sc.meshes[0].name
    with pytest.raises(ValueError):
        sc.validate()


def test_option_file_keys():
    # the keys of the reference's option file (reference README.md:212-250, src/options.txt)
    text = "-dispw 1920\n-disph 1080\n-mpbs 200\n-hfov 60\n-spp 2\n-bounce 3\n-spr 3\n-eps 0.0001\n-reps 0.1\n-rbounce 5\n\n-unknown 1\n"
    opt, extra = parse_options(text)
    assert (opt.width, opt.height, opt.spp, opt.spr, opt.max_bounce, opt.max_refrac_bounce) == (1920, 1080, 2, 3, 3, 5)
    assert abs(opt.hfov - 60) < 1e-6 and abs(opt.eps - 1e-4) < 1e-10 and abs(opt.ray_eps - 0.1) < 1e-8
    assert extra["mpbs"] == 200
    d = Options()     # the reference's defaults, src/main.cpp:74-75,103-111
    assert (d.width, d.height, d.spp, d.spr, d.max_bounce, d.max_refrac_bounce) == (960, 540, 2, 4, 3, 3)


def test_abi_struct_sizes():
    assert ctypes.sizeof(MaterialDesc) == 64 and ctypes.sizeof(LightDesc) == 32 and ctypes.sizeof(CameraDesc) == 64
    assert ctypes.sizeof(Options) == 68 and ctypes.sizeof(Stats) == 88


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|const char\*)\s+(rt_\w+)\s*\(", header, flags=re.M)))
    assert declared == sorted(api.EXPORTS)
    L = api.lib()       # raises if librt_amd.so was not built
    for name in declared:
        assert hasattr(L, name), name


def test_compute_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.RtError):
        api.init()


def test_band_schedule_covers_every_row_once():
    from raytracer_amd.distributed import band_schedule
    for h, n, b in [(1080, 8, 8), (1077, 3, 16), (5, 8, 8), (2160, 8, 12)]:
        sched = band_schedule(h, n, b)
        rows = sorted(y for bands in sched for y0, y1 in bands for y in range(y0, y1))
        assert rows == list(range(h))
        sizes = [sum(y1 - y0 for y0, y1 in bands) for bands in sched]
        assert max(sizes) - min(sizes) <= b
        # the closed form the library uses for local row -> image row (rt_options.row_stride)
        from raytracer_amd.distributed import local_to_image_rows
        for r in range(n):
            rows = local_to_image_rows(h, n, r, b)
            assert rows == [((l // b) * n + r) * b + l % b for l in range(len(rows))]


def _soup(tris, name="m"):
    from raytracer_amd.scene import Mesh
    pos = np.asarray(tris, np.float32).reshape(-1, 3, 3)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (pos.shape[0], 3, 1))
    return Mesh(name, 0, pos, nrm)


@pytest.mark.parametrize("which", [0, 1])
def test_both_tree_builders_keep_their_invariants(which):
    """Host-side check of the flattened reference trees (0) and of the surface-area-heuristic tree (1): every
    triangle in exactly one leaf, boxes nested and containing their triangles, tie ranks a permutation."""
    for sc in (scenes.bunny(), scenes.mirror_room(6, sphere_res=8), scenes.glass_balls(10), scenes.bunny_field(12, seed=3)):
        info = api.tree_check(sc, which)
        assert info["tris"] == sc.n_tris == info["reached_tris"]
        if which == 1:
            assert info["max_leaf"] <= 4
    # the reference's tree of the shipped bunny scene: 4579 bottom nodes + 4576 inner nodes = the 9155 BVH nodes and the
    # depth of 20 counted on the real reference (two more node records here join its three meshes)
    if which == 0:
        info = api.tree_check(scenes.bunny(), 0)
        assert info["leaves"] == 4579 and info["reached_nodes"] - 2 + info["leaves"] == 9155 and info["depth"] == 20


@pytest.mark.parametrize("which", [0, 1])
def test_tree_builders_on_degenerate_input(which):
    base = scenes.bunny()
    one = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    cases = [
        [one],                                              # a single triangle
        [one] * 9,                                          # nine coincident triangles (no split separates them)
        [[[0, 0, 0], [0, 0, 0], [0, 0, 0]]] * 6,            # zero-area triangles in one point
        [[[k, 0, 0], [k + 1, 0, 0], [k, 1, 0]] for k in range(7)] + [[[-1e6, 0, 0], [1e6, 0, 0], [0, 1e6, 0]]],   # a huge one among small
    ]
    for tris in cases:
        sc = Scene(meshes=[_soup(tris)], materials=base.materials[:1], lights=base.lights, camera=base.camera)
        info = api.tree_check(sc, which)
        assert info["reached_tris"] == len(tris)
    two = Scene(meshes=[_soup(cases[0], "a"), _soup(cases[1], "b")], materials=base.materials[:1], lights=base.lights, camera=base.camera)
    assert api.tree_check(two, which)["reached_tris"] == 10


def test_frame_driver_host_parts(tmp_path):
    """Display-block order, report lines and the PNG writer of the headless frame driver (no GPU involved)."""
    import struct
    import zlib
    from raytracer_amd import render
    blocks = list(render.display_blocks(250, 130, 100))
    assert blocks[0] == (0, 0, 100, 100) and blocks[2] == (200, 0, 250, 100) and blocks[-1] == (200, 100, 250, 130)
    cover = np.zeros((130, 250), int)
    for x0, y0, x1, y1 in blocks:
        cover[y0:y1, x0:x1] += 1
    assert (cover == 1).all()
    opt, extras = parse_options("-spp 2\n-spr 3\n-bounce 3\n-dispw 64\n-disph 36\n-mpbs 50\n")
    lines = render.report_lines(opt)
    assert "Samples per pixel: 4" in lines and "Samples per ray: 4" in lines and "Resolution: 64 X 36" in lines
    assert extras["mpbs"] == 50
    img = (np.arange(5 * 7 * 3) % 251).astype(np.uint8).reshape(5, 7, 3)
    p = str(tmp_path / "a.png")
    render.write_png(p, img)
    raw = open(p, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">II", raw[16:24]) == (7, 5)
    idat = raw[raw.index(b"IDAT") + 4: raw.index(b"IEND") - 8]
    rows = zlib.decompress(idat)
    back = np.frombuffer(rows, np.uint8).reshape(5, 1 + 7 * 3)[:, 1:].reshape(5, 7, 3)
    assert (back == img).all()
    render.write_ppm(str(tmp_path / "a.ppm"), img)
    assert open(str(tmp_path / "a.ppm"), "rb").read().endswith(img.tobytes())
    assert render.find_scene(GOLDEN).endswith(".rtscene")


def test_jpeg_writer_matches_the_reference_save_path(tmp_path):
    """The result file as the reference's save_jpg writes it (src/main.cpp:148-180): baseline JFIF, 4:4:4, quality 90,
    comment "Render Result".  Checked with an independent decoder (PIL): markers, the comment, the sampling factors, and
    the decoded picture against the input (the encoder is this build's own, so the bytes are not TooJpeg's; a decoder
    sees the same picture within quantisation error)."""
    from PIL import Image
    rng = np.random.RandomState(3)
    h, w = 75, 133                                    # not multiples of 8: edge blocks
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 // w), (yy * 255 // h), ((xx + yy) % 64) * 4], -1).astype(np.uint8)
    img[20:40, 30:90] = rng.randint(0, 256, (20, 60, 3))          # busy region: long runs, escapes of 0xFF bytes
    img[50:, :20] = 0
    p = str(tmp_path / "a.jpg")
    api.save_jpeg(p, img)
    raw = open(p, "rb").read()
    assert raw[:4] == b"\xff\xd8\xff\xe0" and raw[6:11] == b"JFIF\x00" and raw[-2:] == b"\xff\xd9"
    assert b"\xff\xfe\x00\x0fRender Result" in raw
    im = Image.open(p)
    assert im.size == (w, h) and im.mode == "RGB" and im.info.get("comment") == b"Render Result"
    assert [c[1:3] for c in im.layer] == [(1, 1)] * 3 if hasattr(im, "layer") else True      # no chroma subsampling
    back = np.asarray(im.convert("RGB")).astype(np.float64)
    mse = ((back - img) ** 2).mean()
    assert 10 * np.log10(255 ** 2 / mse) > 30
    # PIL's own encoder (libjpeg) with the same settings lands at the same size and error
    q = str(tmp_path / "pil.jpg")
    Image.fromarray(img).save(q, quality=90, subsampling=0)
    assert 0.8 < len(raw) / os.path.getsize(q) < 1.25
    back_pil = np.asarray(Image.open(q)).astype(np.float64)
    assert mse < 1.15 * ((back_pil - img) ** 2).mean() and np.abs(back - img).max() <= 1.3 * np.abs(back_pil - img).max()
    # a flat black frame (the reference's background) decodes to exact black
    api.save_jpeg(p, np.zeros((16, 24, 3), np.uint8))
    assert (np.asarray(Image.open(p)) == 0).all()


def test_jpeg_header_segments_equal_the_reference_files_byte_for_byte():
    """Everything of a result file that does not depend on the pixels, against files the reference's own save_jpg wrote
    (six of its docs&images/render_result_*.jpg; extracted by tests/golden/make_jpeg_segment_fixtures.py): SOI, APP0
    (JFIF 1.1, no density), COM "Render Result", DQT (Annex K tables at quality 90, src/toojpeg.cpp:413-417, both in one
    segment, zig-zag order), SOF0 (8 bit, size, three components 1x1 = 4:4:4), DHT (the four standard tables in one
    segment) and the SOS header - 624 bytes, in this order.  The entropy-coded bytes after them depend on the encoder's
    DCT and rounding and are checked through a decoder in the test above."""
    import ctypes as C
    z = np.load(os.path.join(GOLDEN, "ref_jpeg_headers.npz"))
    names = [k for k in z.files if not k.endswith("_size")]
    assert len(names) == 6
    L = api.lib()
    for k in names:
        w, h = (int(v) for v in z[k + "_size"])
        ref = z[k]
        assert len(ref) == 624
        img = np.full((h, w, 3), 37, np.uint8)
        buf = np.empty(w * h // 4 + 4096, np.uint8)
        n = C.c_int64()
        assert L.rt_encode_jpeg(img.ctypes.data, w, h, 90, b"Render Result", buf.ctypes.data, buf.size, C.byref(n)) == 0
        assert n.value > 624 and (buf[:624] == ref).all(), k
    # and they do depend on what they should: another quality or comment changes DQT / COM
    n = C.c_int64()
    buf = np.empty(1 << 16, np.uint8)
    img = np.zeros((16, 16, 3), np.uint8)
    assert L.rt_encode_jpeg(img.ctypes.data, 16, 16, 75, b"Render Result", buf.ctypes.data, buf.size, C.byref(n)) == 0
    assert (buf[:20 + 17] == z[names[0]][:37]).all() and not (buf[37:37 + 134] == z[names[0]][37:37 + 134]).all()


def test_baker_generates_smooth_normals_for_meshes_without_normals():
    """The reference imports with GenSmoothNormals (src/main.cpp:295-301): corners at one position share the normalised
    sum of the unit normals of the faces around it."""
    from tools.bake_fbx import smooth_normals
    # a pyramid: apex (0,0,1) over a square, four side faces as one indexed mesh
    verts = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0], [0, 0, 1]], np.float32)
    vert_idx = np.array([0, 1, 4, 1, 2, 4, 2, 3, 4, 3, 0, 4])
    ends = np.tile([False, False, True], 4)
    n = smooth_normals(verts, vert_idx, ends)
    assert np.allclose(np.linalg.norm(n, axis=1), 1, atol=1e-6)
    apex = n[[2, 5, 8, 11]]
    assert np.allclose(apex, [0, 0, 1], atol=1e-6)                       # the four side normals sum to straight up
    # base corner 1 belongs to faces 0 (front, normal (0,-1,1)/sqrt2) and 1 (right, (1,0,1)/sqrt2)
    want = np.array([1, -1, 2], np.float64); want /= np.linalg.norm(want)
    assert np.allclose(n[1], want, atol=1e-6) and np.allclose(n[3], want, atol=1e-6)
    # a flat quad as two triangles keeps the plane normal; a zero-area face contributes nothing
    quad = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    n = smooth_normals(quad, np.array([0, 1, 2, 0, 2, 3, 0, 0, 0]), np.tile([False, False, True], 3))
    assert np.allclose(n[:6], [0, 0, 1], atol=1e-6)
    # on the shipped sphere the generated normals are the shipped smooth normals to within the faceting
    sc = Scene.load(os.path.join(GOLDEN, "unit_sphere.rtscene"))
    m = sc.meshes[0]
    pos = m.pos.reshape(-1, 3)
    g = smooth_normals(pos, np.arange(len(pos)), np.tile([False, False, True], m.n_tris))
    cosang = (g * m.nrm.reshape(-1, 3)).sum(1)
    assert np.median(cosang) > 0.995 and cosang.min() > 0.9
