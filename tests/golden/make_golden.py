"""Writes tests/golden/oracle_vectors.npz from the CPU oracle (the reference itself cannot be built in this image,
so these are regression vectors of the restatement, not outputs of the reference).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import OracleScene  # noqa: E402
from raytracer_amd import scenes  # noqa: E402
from raytracer_amd._abi import Options  # noqa: E402
from tests.conftest import random_rays  # noqa: E402

o = OracleScene(scenes.bunny())
org, d = random_rays(4000, 1234)
t, tri, nrm = o.trace(org, d, 1e-4, np.inf)
img, _ = o.render_rect(Options(width=96, height=54, hfov=60, spp=2, spr=3, max_bounce=3), 0, 0, 96, 54)
img2, _ = OracleScene(scenes.bunny(phong_surroundings=True)).render_rect(
    Options(width=64, height=36, hfov=60, spp=2, spr=3, max_bounce=2, eps=1e-3), 0, 0, 64, 36)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz"), org=org, dir=d, t=t, tri=tri, nrm=nrm,
                    bunny_96x54=img, bunny_phong_64x36=img2)
print("hits", (tri >= 0).mean())
