"""Extracts the header segments of the result images the reference wrote with its own save_jpg.

    python tests/golden/make_jpeg_segment_fixtures.py        (build container only: reads /root/reference; data only)

Six of the JPEGs under docs&images/ are direct output of save_jpg (reference src/main.cpp:148-180: TooJpeg::writeJpeg with
isRGB = true, quality 90, downsample = false, comment "Render Result"); their byte stream up to the first entropy-coded
byte is SOI, APP0 (JFIF 1.1), COM, DQT (luma and chroma table in one segment), SOF0 (8 bit, three components, all 1x1),
DHT (four tables in one segment) and the SOS header.  Those bytes depend on nothing but quality, comment and image size,
so the library's encoder can be held to them byte for byte.  (The other two files were re-saved by an image editor -
EXIF/Photoshop segments - and are skipped.)  Written to tests/golden/ref_jpeg_headers.npz:
    <name>          uint8 array: the file's bytes from SOI up to the end of the SOS header
    <name>_size     (width, height)
"""
import glob
import os
import struct

import numpy as np

REF = "/root/reference/docs&images/"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_jpeg_headers.npz")

out = {}
for path in sorted(glob.glob(REF + "render_result_*.jpg")):
    b = open(path, "rb").read()
    assert b[:2] == b"\xff\xd8"
    i, kinds, size = 2, [], None
    while True:
        assert b[i] == 0xFF
        m, n = b[i + 1], struct.unpack(">H", b[i + 2:i + 4])[0]
        kinds.append(m)
        if m == 0xC0:
            size = struct.unpack(">HH", b[i + 5:i + 9])[::-1]
        i += 2 + n
        if m == 0xDA:
            break
    if kinds != [0xE0, 0xFE, 0xDB, 0xC0, 0xC4, 0xDA]:
        print("skipped (not save_jpg's own output):", os.path.basename(path), [hex(k) for k in kinds])
        continue
    name = os.path.basename(path)[len("render_result_"):-4]
    out[name] = np.frombuffer(b[:i], np.uint8)
    out[name + "_size"] = np.array(size, np.int32)
    print(name, size, i, "header bytes")
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
