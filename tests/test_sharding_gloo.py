"""The N > 1 path on CPU: two gloo ranks render interleaved row bands with a stand-in tile function and gather them."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd.distributed import band_schedule, gather_frame, render_local_bands

H, W, BAND = 37, 16, 4


def _pattern(x0, y0, x1, y1, out):
    ys = torch.arange(y0, y1, dtype=torch.float32).view(-1, 1, 1)
    xs = torch.arange(x0, x1, dtype=torch.float32).view(1, -1, 1)
    cs = torch.arange(3, dtype=torch.float32).view(1, 1, 3)
    out.copy_(ys * 1000 + xs * 10 + cs)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bands = band_schedule(H, world, BAND)[rank]
    local = render_local_bands(_pattern, bands, W, "cpu")
    frame = gather_frame(local, H, W, world, BAND)
    dist.barrier()
    q.put((rank, frame.numpy()))
    dist.destroy_process_group()


def test_two_ranks_reassemble_the_frame():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = torch.empty(H, W, 3)
    _pattern(0, 0, W, H, want)
    for r in (0, 1):
        assert (got[r] == want.numpy()).all()
