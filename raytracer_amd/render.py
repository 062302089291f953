"""Headless frame driver: what the reference's main() does between "scene loaded" and "render_result saved"
(reference src/main.cpp:440-526 renderFrame, :646-683 option file, :700-776 main), without the window.

    python -m raytracer_amd.render <scene.rtscene | folder holding one> [option file] [-o render_result.jpg]
                                   [--seed N] [--jitter off|hash] [--exact] [--blocks] [--log logs.txt]

The scene is a baked .rtscene (tools/bake_fbx.py makes one from the reference's FBX input); the option file has the
reference's keys (-dispw -disph -hfov -spp -spr -bounce -rbounce -eps -reps -mpbs).  The report lines are the
reference's.  The image is written by extension: .jpg (the default, like the reference's save_jpg, src/main.cpp:148-180:
baseline JPEG, quality 90, 4:4:4, comment "Render Result", through the library's own encoder rt_save_jpeg), .png or binary
.ppm.  --blocks renders display block by display block (-mpbs pixels square) like the reference's progressive
loop; the default is one call for the whole frame, which is the same image."""
import argparse
import glob
import os
import struct
import sys
import time
import zlib

import numpy as np

from . import api
from ._abi import Options, RT_JITTER_HASH, RT_JITTER_OFF
from .options import parse_options
from .scene import Scene


def find_scene(path):
    """A .rtscene file, or the only one inside a folder (the reference takes a scene folder, src/main.cpp:704-712)."""
    if os.path.isdir(path):
        found = sorted(glob.glob(os.path.join(path, "*.rtscene")))
        if not found:
            raise FileNotFoundError("no .rtscene in %s" % path)
        return found[0]
    return path


def write_png(path, rgb8):
    """rgb8: [H][W][3] uint8, row 0 at the top."""
    h, w, _ = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_ppm(path, rgb8):
    h, w, _ = rgb8.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(rgb8.tobytes())


def report_lines(opt, n_threads_note="GPU"):
    """The parameter block of the reference's rendering report (src/main.cpp:447-461)."""
    return ["[Rendering Report]",
            "Time: %sRendering starts." % time.ctime() + "\n",
            "Compute: %s" % n_threads_note,
            "Samples per pixel: %d" % (opt.spp * opt.spp),
            "Samples per ray: %d" % (opt.spr + 1),
            "Epsilon: %f" % opt.eps,
            "Ray Jitter Epsilon: %f" % opt.ray_eps,
            "Max bounces: %d" % opt.max_bounce,
            "Max refractive bounces: %d" % opt.max_refrac_bounce,
            "Field of view: %.2f" % opt.hfov,
            "Resolution: %d X %d" % (opt.width, opt.height)]


def display_blocks(width, height, span):
    """The reference's progressive order: blocks of `span` pixels, rows of blocks from the bottom (src/main.cpp:474-507)."""
    span = max(1, int(span))
    for j in range(0, height, span):
        for i in range(0, width, span):
            yield i, j, min(i + span, width), min(j + span, height)


def render_frame(tracer, opt, blocks=None, log=print):
    """float32 [H][W][3] frame (row 0 = bottom row, like the reference's buffer) and the statistics of the call(s)."""
    W, H = opt.width, opt.height
    log("Drawing... ")
    t0 = time.perf_counter()
    if not blocks:
        frame, st = tracer.render_rect(opt, 0, 0, W, H)
        stats = [st]
    else:
        frame = np.empty((H, W, 3), np.float32)
        stats = []
        for (x0, y0, x1, y1) in display_blocks(W, H, blocks):
            part, st = tracer.render_rect(opt, x0, y0, x1, y1)
            frame[y0:y1, x0:x1] = part
            stats.append(st)
    dt = time.perf_counter() - t0
    log("done")
    log("Rendering finished with total duration of %f seconds." % dt)
    return frame, stats, dt


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m raytracer_amd.render", description=__doc__.split("\n\n")[0])
    ap.add_argument("scene")
    ap.add_argument("options", nargs="?")
    ap.add_argument("-o", "--output", default="render_result.jpg",
                    help="result image; .jpg (the reference's format: quality 90, 4:4:4, comment 'Render Result'), .png or .ppm")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--jitter", choices=["off", "hash"], default="hash")
    ap.add_argument("--exact", action="store_true", help="the reference's box arithmetic and leaves (bit-equal to it, slower)")
    ap.add_argument("--blocks", action="store_true", help="render display block by display block (-mpbs)")
    ap.add_argument("--log", default=None, help="also append the report to this file (the reference writes logs.txt)")
    args = ap.parse_args(argv)

    logf = open(args.log, "a") if args.log else None

    def log(msg):
        print(msg, flush=True)
        if logf:
            logf.write(msg + "\n")

    path = find_scene(args.scene)
    scene = Scene.load(path)
    log("Loaded Scene %s" % os.path.basename(path))
    text = ""
    if args.options:
        try:
            text = open(args.options).read()
            log("Initialized using option file [%s]" % args.options)
        except OSError:
            log("Can't open the option file, using default.")
    opt, extras = parse_options(text, Options())
    opt.jitter_mode = RT_JITTER_HASH if args.jitter == "hash" else RT_JITTER_OFF
    opt.seed = args.seed
    opt.exact_boxes = 1 if args.exact else 0

    api.init(-1)
    log("\nConstructing AABB Tree for hit detection...")
    tracer = api.TracerScene(scene)
    log("done")
    for line in report_lines(opt, "one gfx950 device"):
        log(line)
    frame, stats, dt = render_frame(tracer, opt, extras["mpbs"] if args.blocks else None, log)
    rays = sum(s.rays_closest + s.rays_any for s in stats)
    log("Rays traced: %d (%.1f Mrays/s)" % (rays, rays / dt / 1e6))
    rgb8 = api.resolve_u8(frame)
    ext = os.path.splitext(args.output)[1].lower()
    if ext in (".jpg", ".jpeg"):
        api.save_jpeg(args.output, rgb8)            # src/main.cpp:169-176
    else:
        (write_ppm if ext == ".ppm" else write_png)(args.output, rgb8)
    log("Render result saved to %s" % args.output)
    if logf:
        logf.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
