"""The boundary used from C: include/rt_amd.h compiles as C99, examples/host.c (plain gcc, no hipcc) links against
librt_amd.so, and on a GPU its image of the bunny scene is the oracle's image byte for byte."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import GOLDEN, ROOT

HOST = os.path.join(ROOT, "examples", "host")


def test_header_is_c99_and_the_example_links(tmp_path):
    src = tmp_path / "hdr.c"
    src.write_text('#include "%s"\nint main(void) { return sizeof(rt_options) == 68 && sizeof(rt_stats) == 88 ? 0 : 1; }\n'
                   % os.path.join(ROOT, "include", "rt_amd.h"))
    exe = str(tmp_path / "hdr")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-o", exe, str(src)])
    assert subprocess.call([exe]) == 0
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    # every library symbol the example needs resolves against librt_amd.so (the dynamic loader checks at start-up;
    # without a GPU rt_init then fails with a message, which is the contract)
    undefined = subprocess.check_output(["nm", "-u", HOST]).decode()
    used = sorted(l.split()[-1] for l in undefined.splitlines() if " rt_" in l)
    assert used == ["rt_init", "rt_last_error", "rt_render_rect", "rt_resolve_u8", "rt_scene_create", "rt_scene_destroy", "rt_scene_info"]
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([HOST, os.path.join(GOLDEN, "stanford_bunny.rtscene"), "32", "18"], capture_output=True, text=True)
        assert r.returncode == 1 and "rt_init" in r.stderr


@pytest.mark.gpu
def test_c_host_renders_the_oracles_image(oracle_mod, bunny_scene, tmp_path):
    from raytracer_amd._abi import Options
    assert os.path.exists(HOST), "examples/host is built by __graft_entry__.build() and travels with the tree"
    out = str(tmp_path / "c.ppm")
    r = subprocess.run([HOST, os.path.join(GOLDEN, "stanford_bunny.rtscene"), "128", "72", out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = {l.split()[0]: l.split()[1:] for l in r.stdout.splitlines()}
    opt = Options(width=128, height=72, exact_boxes=1)          # host.c uses the reference's defaults, jitter off
    a, st = oracle_mod.OracleScene(bunny_scene).render_rect(opt, 0, 0, 128, 72)
    want = oracle_mod.resolve_u8(a)
    ppm = np.frombuffer(open(out, "rb").read()[-72 * 128 * 3:], np.uint8).reshape(72, 128, 3)
    assert (ppm == want).all()
    h = 1469598103934665603
    for b in want.tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert lines["fnv64"][0] == "%016x" % h
    # the reference traces every render ray twice; the library once
    assert int(lines["rays_closest"][0]) * 2 == st.rays_closest and int(lines["rays_closest"][2]) == st.rays_any
    assert lines["triangles"][0] == "7864"
