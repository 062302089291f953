"""Turns the two result images the reference ships for its bunny scene into small data fixtures.

    python tests/golden/make_ref_image_fixtures.py        (build container only: reads /root/reference, needs PIL)

The reference's README shows its own 1920x1080 renders of input_files/defaultScene with the settings of the first two
benchmark configurations (reference README.md:26-56):
    docs&images/render_result_standford_rabbit_lambert_softshadow.jpg                  Lambert, shadow rays, hfov 60
    docs&images/render_result_standford_rabbit_lambert_softshadow_softreflection.jpg   ground and dome Phong-reflective
They are JPEGs made with jitter, two lights and a teal rabbit material that the committed scene file does not have, so
colours are not comparable; what is exact in them is geometry: the background is exactly black (no draw -> NaN -> 0,
reference src/main.cpp:165-167), the rabbit is the only teal object, and in the second image teal outside the rabbit is
the rabbit's mirror image in the ground and the dome.  Written to tests/golden/ref_images.npz, sampled at the centre of
every 2x2 pixel block (960x540, row 0 = BOTTOM row of the image, as the renderer counts rows):
    bright1, bright2   max(R, G, B) > 6        (bit-packed along x)
    teal1, teal2       B - R > 10              (bit-packed along x)
    luma1              max(R, G, B) of image 1 at every 4th pixel (480x270), uint8
"""
import os

import numpy as np
from PIL import Image

REF = "/root/reference/docs&images/"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_images.npz")
NAMES = ["render_result_standford_rabbit_lambert_softshadow.jpg",
         "render_result_standford_rabbit_lambert_softshadow_softreflection.jpg"]

out = {}
for k, name in enumerate(NAMES, 1):
    im = np.asarray(Image.open(REF + name).convert("RGB")).astype(np.int32)
    assert im.shape == (1080, 1920, 3)
    im = im[::-1]                                   # row 0 = bottom
    half = im[1::2, 1::2]
    out["bright%d" % k] = np.packbits(half.max(axis=2) > 6, axis=1)
    out["teal%d" % k] = np.packbits(half[..., 2] - half[..., 0] > 10, axis=1)
    if k == 1:
        out["luma1"] = im[2::4, 2::4].max(axis=2).astype(np.uint8)
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
