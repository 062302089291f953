"""Row-band sharding of a frame over the GPUs of one node.

The reference renders disjoint pixel rectangles on threads that share one framebuffer (reference
src/main.cpp:474-507).  Here rank r of N renders every N-th band of `band` rows (band k belongs to rank k % N, so sky
and geometry rows are spread evenly) in ONE call, with the scene replicated on its GPU; nothing is exchanged while
tracing, and the finished rows are all-gathered once per frame (RCCL over xGMI for GPU tensors, gloo on CPU in tests).
"""
import torch
import torch.distributed as dist


def band_schedule(height, world_size, band_rows=8):
    """Bands [(y0, y1), ...] per rank: band k = rows [k*band_rows, (k+1)*band_rows) goes to rank k % world_size."""
    bands = [(y, min(y + band_rows, height)) for y in range(0, height, band_rows)]
    return [[b for k, b in enumerate(bands) if k % world_size == r] for r in range(world_size)]


def local_to_image_rows(height, world_size, rank, band_rows=8):
    """Image row of each local row of `rank`'s share, in local order (the mapping rt_options.row_stride documents)."""
    rows = []
    for y0, y1 in band_schedule(height, world_size, band_rows)[rank]:
        rows.extend(range(y0, y1))
    return rows


class FrameGather:
    """All-gather of the ranks' row shares into whole frames, reusable across frames (buffers and index are cached)."""

    def __init__(self, height, width, world_size, band_rows, device, group=None, dtype=torch.float32, flipped=False,
                 collective_on_one_rank=False):
        """collective_on_one_rank: with a single rank, still go through the process group's all-gather instead of a plain
        copy (a rehearsal of the RCCL path - its own stream, asynchronous work handles - on a box with one GPU).
        flipped: the ranks exchange their rows as rt_resolve_u8 writes them - top row first WITHIN the share (local
        row l of a share of n rows sits at n-1-l) - and the assembled frame is top-down as well (the 8-bit image the
        reference's save_jpg builds, reference src/main.cpp:157-167)."""
        self.h, self.w, self.world, self.group, self.dtype = height, width, world_size, group, dtype
        shares = [local_to_image_rows(height, world_size, r, band_rows) for r in range(world_size)]
        self.max_rows = max(len(s) for s in shares)
        self.n_rows = [len(s) for s in shares]
        # output row k is row src[k] of the gathered [world * max_rows] stack; rows beyond a rank's share (padding of the
        # shorter shares) are never referenced
        src = torch.empty(height, dtype=torch.long)
        for r, s in enumerate(shares):
            for l, y in enumerate(s):
                if flipped:
                    src[height - 1 - y] = r * self.max_rows + (len(s) - 1 - l)
                else:
                    src[y] = r * self.max_rows + l
        self.src = src.to(device)
        self.device = device
        self.copy_only = world_size == 1 and not collective_on_one_rank

    def new_local(self):
        """A buffer for this rank's rows, padded to the largest share so that all ranks gather equal sizes."""
        return torch.zeros((self.max_rows, self.w, 3), dtype=self.dtype, device=self.device)

    def new_stack(self):
        return torch.empty((self.world * self.max_rows, self.w, 3), dtype=self.dtype, device=self.device)

    def start(self, local, stack):
        """Starts the all-gather of `local` into `stack`; returns a handle to pass to finish() (None on one rank)."""
        if self.copy_only:
            stack.copy_(local)
            return None
        if local.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal of the sharded path by several ranks on ONE GPU (RCCL refuses ranks that share a device): the
            # exchange goes through host memory, synchronously (gloo's own asynchronous staging of GPU tensors stalled
            # with two gathers in flight)
            h = local.cpu()
            hs = torch.empty(stack.shape, dtype=stack.dtype)
            dist.all_gather_into_tensor(hs.view(-1), h.view(-1), group=self.group)
            stack.copy_(hs)
            return None
        return dist.all_gather_into_tensor(stack.view(-1), local.view(-1), group=self.group, async_op=True)

    def finish(self, work, stack):
        """Waits for the gather and returns the frame [height, width, 3] in image row order."""
        if work is not None:
            work.wait()
        return stack.index_select(0, self.src)


def run_pipelined(steps, render, gather, locals_, stacks, streams=None):
    """Renders `steps` frames with render(k, local_buffer) and gathers each one.  Frame k uses buffer set k % n
    (n = len(locals_) >= 2), so the gather of frame k runs while the next frames render.  With `streams` (one
    torch.cuda.Stream per set) frame k is rendered, gathered and reordered on stream k % n: consecutive frames then
    overlap on the GPU as well, which is what keeps it busy when a rank's share of a frame is small (a kernel over an
    eighth of a frame is mostly ramp-up and tail).  Every rank issues its gathers in frame order.
    Returns the last whole frame (None when steps == 0).

    render(k, buf) must enqueue its work on the stream that is current when it is called; gather is a FrameGather."""
    n = len(locals_)
    pending = [None] * n
    frame = None

    def on(b):
        return torch.cuda.stream(streams[b]) if streams else _NullContext()

    if streams:
        # the buffers were created (and zero-filled) on the caller's stream: the side streams start behind it
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())

    for k in range(steps):
        b = k % n
        with on(b):
            if pending[b] is not None:                  # the buffers of frame k-n are free once its gather has finished
                frame = gather.finish(pending[b][0], stacks[b])
                pending[b] = None
            render(k, locals_[b])
            pending[b] = (gather.start(locals_[b], stacks[b]),)
    for k in range(steps, steps + n):                   # drain in frame order
        b = k % n
        if pending[b] is not None:
            with on(b):
                frame = gather.finish(pending[b][0], stacks[b])
            pending[b] = None
    if streams:
        for st in streams:                              # the caller's stream continues after every frame is complete
            torch.cuda.current_stream().wait_stream(st)
        if frame is not None and frame.is_cuda:
            # the frame was allocated on a side stream; the caching allocator must not hand its memory to later
            # side-stream work while the caller's stream still reads it
            frame.record_stream(torch.cuda.current_stream())
    return frame


class _NullContext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def gather_frame(local_rows, height, width, world_size, band_rows=8, group=None):
    """One-shot form: pads `local_rows` ([n_local, width, 3]), gathers and reorders; returns the whole frame."""
    fg = FrameGather(height, width, world_size, band_rows, local_rows.device, group)
    local = fg.new_local()
    local[:local_rows.shape[0]] = local_rows
    stack = fg.new_stack()
    return fg.finish(fg.start(local, stack), stack)
