"""The N > 1 path on CPU: two gloo ranks render interleaved row bands with a stand-in tile function and gather them."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd.distributed import FrameGather, gather_frame, local_to_image_rows, run_pipelined

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
    rows = local_to_image_rows(H, world, rank, BAND)
    local = torch.empty((len(rows), W, 3))
    for l, y in enumerate(rows):                 # stand-in for the renderer: fills local row l with image row y
        _pattern(0, y, W, y + 1, local[l:l + 1])
    frame = gather_frame(local, H, W, world, BAND)
    # the benchmark's loop: five frames, gather of frame k overlapped with "rendering" frame k+1; frame k adds k
    fg = FrameGather(H, W, world, BAND, "cpu")
    bufs, stacks = [fg.new_local() for _ in range(2)], [fg.new_stack() for _ in range(2)]

    def render(k, buf):
        buf[:len(rows)] = local + float(k)
    last = run_pipelined(5, render, fg, bufs, stacks)
    # the same with three frames in flight (what the benchmark does on several ranks: a buffer set per stream)
    bufs3, stacks3 = [fg.new_local() for _ in range(3)], [fg.new_stack() for _ in range(3)]
    last3 = run_pipelined(7, render, fg, bufs3, stacks3)
    assert torch.equal(last3, last + 2.0)
    # the benchmark's default exchange: 8-bit rows as rt_resolve_u8 leaves them (top row of the SHARE first), assembled
    # into the top-down image; the padding rows of the shorter share hold 255 and must never reach the frame
    fg8 = FrameGather(H, W, world, BAND, "cpu", dtype=torch.uint8, flipped=True)
    b8, s8 = fg8.new_local(), fg8.new_stack()
    b8.fill_(255)
    b8[:len(rows)] = (local % 251).to(torch.uint8).flip(0)
    top_down = fg8.finish(fg8.start(b8, s8), s8)
    dist.barrier()
    q.put((rank, frame.numpy(), last.numpy(), top_down.numpy()))
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
    got = {}
    for _ in procs:
        r, a, b, c = q.get(timeout=120)
        got[r] = a
        got[(r, "last")] = b
        got[(r, "u8")] = c
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = torch.empty(H, W, 3)
    _pattern(0, 0, W, H, want)
    for r in (0, 1):
        assert (got[r] == want.numpy()).all()
        assert (got[(r, "last")] == want.numpy() + 4.0).all()
        assert (got[(r, "u8")] == (want % 251).to(torch.uint8).flip(0).numpy()).all()


def test_bench_starts_the_ranks_itself(monkeypatch, capsys):
    """`python bench.py --gpus N` outside a torchrun job starts N ranks as child processes (before anything touches a GPU)
    and relays rank 0's line; inside a job (RANK set) it must not start another."""
    import subprocess
    import bench
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        return subprocess.CompletedProcess(cmd, 0, stdout='noise\n{"metric": "m", "n_gpus": 4}\n')
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("RANK", raising=False)
    assert bench.main(["--gpus", "4", "--steps", "2", "--warmup", "1"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    assert capsys.readouterr().out.strip() == '{"metric": "m", "n_gpus": 4}'
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: subprocess.CompletedProcess(cmd, 3, stdout=""))
    assert bench.main(["--gpus", "2"]) == 3


@pytest.mark.gpu
def test_two_ranks_render_and_gather_the_real_frame_on_one_gpu():
    """The sharded benchmark path with the real renderer: two ranks share the one GPU (gloo carries the exchange, since
    RCCL refuses ranks on one device), each renders its interleaved bands through the C-ABI, the 8-bit rows are gathered
    and every rank checks the assembled frame against the frame it renders whole (--verify)."""
    import json
    import subprocess
    import sys
    from tests.conftest import ROOT
    env = dict(os.environ, RT_BENCH_BACKEND="gloo", RT_BENCH_WATCHDOG="100")
    for gather in ("u8", "f32"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--verify", "--steps", "5", "--warmup", "3",
                            "--gather", gather, "--no-cpu-baseline", "--no-workloads"], env=env, capture_output=True, text=True, timeout=200)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and line["config"]["verified"] is True and line["config"]["rays_per_frame"] == 30050272


@pytest.mark.gpu
@pytest.mark.parametrize("workload,streams", [("mirror_room", 0), ("mirror_room", 3), ("glass_balls", 0)])
def test_two_ranks_share_a_frame_of_a_scene_that_spawns_rays(workload, streams):
    """The same rehearsal on the workloads whose frames are NOT one asynchronous level: mirrors and glass make every rank
    go through its interleaved share level by level (a host round trip per level, children spawned into the next level's
    queues, the colour walk over the stored tree), at a third of the frame's size.  With the default one handle per rank
    and with three handles on three streams per rank; every rank compares the gathered frame with the whole frame."""
    import json
    import subprocess
    import sys
    from tests.conftest import ROOT
    env = dict(os.environ, RT_BENCH_BACKEND="gloo", RT_BENCH_WATCHDOG="150")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--verify", "--steps", "4", "--warmup", "1", "--workload", workload,
           "--scale", "0.34", "--no-cpu-baseline", "--no-workloads", "--no-roofline"]
    if streams:
        cmd += ["--streams", str(streams)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["verified"] is True and line["config"]["scale"] == 0.34
    assert line["config"]["streams"] == (streams or 1)
    assert line["config"]["rays_per_frame"] > 50_000_000


_RCCL_REHEARSAL = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
from raytracer_amd import api, scenes
from raytracer_amd._abi import Options, RT_JITTER_HASH
from raytracer_amd.distributed import FrameGather, run_pipelined

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = sys.argv[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
api.init(-1)
W, H, BAND = 640, 356, 8
opt = Options(width=W, height=H, hfov=60, spp=2, spr=3, max_bounce=3, jitter_mode=RT_JITTER_HASH, seed=1)
tracers = [api.TracerScene(scenes.bunny()) for _ in range(3)]
want = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
tracers[0].render_rect(opt, 0, 0, W, H, out=want, stats=False)
want8 = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
api.resolve_u8(want, out=want8)
torch.cuda.synchronize()
for n_streams in (1, 3):
    fg = FrameGather(H, W, 1, BAND, dev, dtype=torch.uint8, flipped=True, collective_on_one_rank=True)
    n_sets = max(2, n_streams)
    rgb = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(n_sets)]
    locals_ = [fg.new_local() for _ in range(n_sets)]
    stacks = [fg.new_stack() for _ in range(n_sets)]
    streams = [torch.cuda.Stream() for _ in range(n_streams)] if n_streams > 1 else None

    def render(k, buf):
        b = k % n_sets
        cur = torch.cuda.current_stream().cuda_stream
        o = Options.from_buffer_copy(opt)
        o.seed = 1 if k == 8 else 100 + k                 # only the last frame is the one compared
        tracers[b % n_streams].render_rect(o, 0, 0, W, H, out=rgb[b], stats=False, stream=cur)
        api.resolve_u8(rgb[b], out=buf, stream=cur)
    frame = run_pipelined(9, render, fg, locals_, stacks, streams)
    torch.cuda.synchronize()
    assert frame.shape == (H, W, 3) and bool((frame == want8).all()), "gathered frame differs with %d streams" % n_streams
dist.barrier()
dist.destroy_process_group()
print("rccl rehearsal ok")
"""


@pytest.mark.gpu
def test_one_rank_rehearses_the_rccl_gather():
    """The exchange of the sharded path on the backend the benchmark uses with several GPUs (RCCL: collectives on a stream
    of the process group's own, asynchronous work handles), as far as one GPU allows: a single rank whose FrameGather goes
    through all_gather_into_tensor instead of the one-rank copy, nine pipelined frames with different seeds on one and on
    three streams, the last one compared with the frame rendered directly."""
    import subprocess
    import sys
    from tests.conftest import ROOT
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-c", _RCCL_REHEARSAL, str(port)], cwd=ROOT, capture_output=True, text=True, timeout=200,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0 and "rccl rehearsal ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


@pytest.mark.gpu
def test_bench_rehearses_its_gather_on_rccl_with_one_rank():
    """bench.py with RT_BENCH_REHEARSE_GATHER=1: one rank initialises the RCCL process group and runs the code the
    multi-GPU job runs - three handles on three streams, the pipelined loop, one all_gather_into_tensor per frame, the
    row reordering - and checks the gathered frame against the frame rendered whole."""
    import json
    import subprocess
    import sys
    from tests.conftest import ROOT
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RT_BENCH_REHEARSE_GATHER="1", MASTER_PORT=str(port), RT_BENCH_WATCHDOG="150")
    for gather in ("u8", "f32"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--verify", "--steps", "6", "--warmup", "2", "--gather", gather,
                            "--no-cpu-baseline", "--no-workloads", "--no-roofline"], env=env, capture_output=True, text=True, timeout=200)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["config"]["gather_rehearsal"] is True and line["config"]["verified"] is True and line["config"]["streams"] == 3
