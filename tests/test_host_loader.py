"""The host side in the host language: the library's own scene loader (rt_host_scene_load: binary FBX the way the
reference's load_scene consumes it, reference src/main.cpp:256-395), option parser (src/main.cpp:646-683) and the C++
host program examples/raytracer_host.cpp (the reference's main() without its window).

tests/golden/fbx/*.fbx are the reference's own scene files (input_files/defaultScene and input_files/testScene; data, not
code).  What the loader must produce for them is pinned twice: by the .rtscene fixtures (baked earlier by the independent
Python reader tools/bake_fbx.py + tools/fbx_reader.py) - every position, normal, material, light and the camera bit for
bit - and by the mesh names and triangle counts recorded from the reference's own loader on these files (ground 200, rabbit
4 968, dome 2 696; unit cube 108*, sphere 760, cylinder 240, plane 200; * with its quads split).  Whether each float
equals Assimp's is not verified anywhere (no Assimp in this image)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import api
from raytracer_amd._abi import Options
from raytracer_amd.scene import Scene
from tests.conftest import GOLDEN, ROOT, bits_equal

FBX = os.path.join(GOLDEN, "fbx")
HOST = os.path.join(ROOT, "examples", "raytracer_host")
NAMES = ["2020_plane_light", "stanford_bunny", "unit_cube", "unit_cylinder", "unit_sphere", "unit_sphere_cam", "xyz"]


@pytest.mark.parametrize("name", NAMES)
def test_cpp_loader_reproduces_the_baked_scene_bit_for_bit(name):
    a = api.load_scene_file(os.path.join(FBX, name + ".fbx"))
    b = Scene.load(os.path.join(GOLDEN, name + ".rtscene"))
    assert [(m.name, m.material, m.n_tris) for m in a.meshes] == [(m.name, m.material, m.n_tris) for m in b.meshes]
    for x, y in zip(a.meshes, b.meshes):
        assert bits_equal(x.pos, y.pos).all() and bits_equal(x.nrm, y.nrm).all()
    assert len(a.materials) == len(b.materials) and all(bits_equal(x.packed(), y.packed()).all() for x, y in zip(a.materials, b.materials))
    assert len(a.lights) == len(b.lights) and all(bits_equal(x.packed(), y.packed()).all() for x, y in zip(a.lights, b.lights))
    assert bits_equal(a.camera.packed(), b.camera.packed()).all()
    # and the flat form read back by the same entry point is itself
    c = api.load_scene_file(os.path.join(GOLDEN, name + ".rtscene"))
    assert all(bits_equal(x.pos, y.pos).all() and bits_equal(x.nrm, y.nrm).all() and x.name == y.name for x, y in zip(b.meshes, c.meshes))
    assert bits_equal(b.camera.packed(), c.camera.packed()).all()


def test_cpp_loader_matches_what_was_recorded_from_the_reference_loader():
    sc = api.load_scene_file(os.path.join(FBX, "stanford_bunny.fbx"))
    assert [(m.name, m.n_tris) for m in sc.meshes] == [("ground", 200), ("rabbit", 4968), ("dome", 2696)]
    assert [m.kind for m in sc.materials] == [1] and len(sc.lights) == 1 and sc.lights[0].kind == 0
    assert np.allclose(sc.camera.pos, [0, 1.5, 4], atol=1e-5) and sc.camera.near == pytest.approx(0.1) and sc.camera.far == pytest.approx(10000.0)
    assert np.allclose(sc.materials[0].base, 0.4, atol=1e-6) and np.allclose(sc.materials[0].ambient, 0.0)
    counts = {n: sum(m.n_tris for m in api.load_scene_file(os.path.join(FBX, n + ".fbx")).meshes)
              for n in ("unit_cube", "unit_sphere", "unit_cylinder", "2020_plane_light")}
    assert counts == {"unit_cube": 108, "unit_sphere": 760, "unit_cylinder": 240, "2020_plane_light": 200}


def test_cpp_loader_rejects_what_is_not_a_scene(tmp_path):
    L = api.lib()
    h = C.c_void_p()
    junk = tmp_path / "junk.fbx"
    junk.write_bytes(b"Kaydara FBX Binary  \x00\x1a\x00" + (7400).to_bytes(4, "little") + b"\xff" * 40)
    for path in (str(tmp_path / "absent.fbx"), str(junk), __file__):
        assert L.rt_host_scene_load(path.encode(), C.byref(h)) != 0 and L.rt_host_last_error()
    whole = open(os.path.join(FBX, "unit_cube.fbx"), "rb").read()
    cut = tmp_path / "cut.fbx"
    for n in (27, 200, len(whole) // 2, len(whole) - 200):            # truncated files end in an error, not in a crash
        cut.write_bytes(whole[:n])
        rc = L.rt_host_scene_load(str(cut).encode(), C.byref(h))
        if rc == 0:
            L.rt_host_scene_free(h)
    assert L.rt_host_scene_load(None, C.byref(h)) != 0


def test_option_lines_as_the_reference_reads_them():
    from raytracer_amd.options import parse_options
    L = api.lib()
    o = Options()
    assert L.rt_options_default(C.byref(o)) == 0
    assert (o.width, o.height, o.spp, o.spr, o.max_bounce, o.max_refrac_bounce) == (960, 540, 2, 4, 3, 3)
    assert o.hfov == pytest.approx(54.43) and o.eps == pytest.approx(1e-4) and o.ray_eps == pytest.approx(0.1)
    text = "-dispw 1920\n-disph 1080\n\n-hfov 60\n-spp 2\n-spr 3\n-bounce 2\n-rbounce 5\n-eps 0.001\n-reps 0.01\n-mpbs 40\n-unknown 7\n-spp\n"
    blk = C.c_int(100)
    used = sum(L.rt_options_parse_line(C.byref(o), line.encode(), C.byref(blk)) for line in text.splitlines())
    assert used == 10 and blk.value == 40
    p, extras = parse_options(text)
    for f in ("width", "height", "hfov", "spp", "spr", "max_bounce", "max_refrac_bounce", "eps", "ray_eps"):
        assert getattr(o, f) == getattr(p, f), f
    assert extras["mpbs"] == 40


def test_cpp_host_builds_against_the_header_alone():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    undefined = subprocess.check_output(["nm", "-u", HOST]).decode()
    used = sorted(l.split()[-1] for l in undefined.splitlines() if " rt_" in l)
    assert used == ["rt_host_last_error", "rt_host_scene_desc", "rt_host_scene_free", "rt_host_scene_load", "rt_init", "rt_last_error",
                    "rt_options_default", "rt_options_parse_line", "rt_render_frame_multi", "rt_render_rect", "rt_resolve_u8", "rt_save_jpeg",
                    "rt_scene_create", "rt_scene_destroy"]
    assert not [l for l in undefined.splitlines() if " hip" in l]              # no HIP runtime symbol: the boundary is the C-ABI
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([HOST, FBX, "--index", "1"], capture_output=True, text=True)
        assert r.returncode == 1 and "rt_init" in r.stderr and "Detected Meshes" in r.stdout and "rabbit" in r.stdout


@pytest.mark.gpu
def test_cpp_host_renders_the_reference_scene_from_its_fbx_file(oracle_mod, bunny_scene, tmp_path):
    """Scene FOLDER and option file in, JPEG out, as the reference's executable is used: the 8-bit picture (written as PPM
    for an exact comparison) is the oracle's byte for byte in the reference-arithmetic mode, rendered whole, block by
    block and over two scene handles; the JPEG carries the reference's comment and decodes to the same picture."""
    from PIL import Image
    from raytracer_amd._abi import RT_JITTER_HASH
    assert os.path.exists(HOST), "examples/raytracer_host is built by __graft_entry__.build() and travels with the tree"
    optfile = tmp_path / "options.txt"
    optfile.write_text("-dispw 160\n-disph 90\n-hfov 60\n-spp 2\n-spr 3\n-bounce 3\n-eps 0.0001\n-reps 0.1\n-mpbs 64\n")
    opt = Options(width=160, height=90, hfov=60, spp=2, spr=3, max_bounce=3, eps=1e-4, ray_eps=0.1, jitter_mode=RT_JITTER_HASH, seed=7, exact_boxes=1)
    a, st = oracle_mod.OracleScene(bunny_scene).render_rect(opt, 0, 0, 160, 90)
    want = oracle_mod.resolve_u8(a)
    for extra in ([], ["--blocks"], ["--gpus", "1"]):
        out = str(tmp_path / "o.ppm")
        r = subprocess.run([HOST, FBX, str(optfile), "--index", "1", "-o", out, "--seed", "7", "--exact", "--log", str(tmp_path / "logs.txt")] + extra,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        got = np.frombuffer(open(out, "rb").read()[-90 * 160 * 3:], np.uint8).reshape(90, 160, 3)
        assert (got == want).all(), extra
        assert "Samples per pixel: 4" in r.stdout and "Samples per ray: 4" in r.stdout and "Resolution: 160 X 90" in r.stdout
        assert "get_hit %d times" % (st.rays_closest + st.rays_any) in r.stdout
    assert "Rendering finished with total duration" in open(str(tmp_path / "logs.txt")).read()
    jpg = str(tmp_path / "render_result.jpg")
    r = subprocess.run([HOST, os.path.join(FBX, "stanford_bunny.fbx"), str(optfile), "-o", jpg, "--seed", "7", "--exact"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    im = Image.open(jpg)
    assert im.size == (160, 90) and im.info.get("comment") == b"Render Result"
    assert np.abs(np.asarray(im).astype(int) - want.astype(int)).mean() < 2.0
