"""Row-band sharding of a frame over the GPUs of one node.

The reference renders disjoint pixel rectangles on threads that share one framebuffer (reference
src/main.cpp:474-507); here each rank renders interleaved bands of rows with the scene replicated on its GPU,
nothing is exchanged while tracing, and the finished rows are gathered once (RCCL all_gather over xGMI when the
tensors are on GPUs, gloo on CPU for tests)."""
import torch
import torch.distributed as dist


def band_schedule(height, world_size, band_rows=8):
    """Bands [(y0, y1), ...] of at most band_rows rows, band k owned by rank k % world_size (interleaved so that sky
    and geometry rows are spread evenly)."""
    bands = [(y, min(y + band_rows, height)) for y in range(0, height, band_rows)]
    return [[b for k, b in enumerate(bands) if k % world_size == r] for r in range(world_size)]


def render_local_bands(render_rect, bands, width, device):
    """Calls render_rect(x0, y0, x1, y1, out) for each band; returns this rank's rows stacked [n_rows, width, 3]."""
    n_rows = sum(y1 - y0 for y0, y1 in bands)
    local = torch.empty((max(n_rows, 1), width, 3), dtype=torch.float32, device=device)
    r = 0
    for y0, y1 in bands:
        render_rect(0, y0, width, y1, local[r:r + (y1 - y0)])
        r += y1 - y0
    return local[:n_rows]


def gather_frame(local_rows, height, width, world_size, band_rows=8, group=None):
    """All-gathers every rank's rows and puts them back in image order; returns [height, width, 3] on every rank."""
    sched = band_schedule(height, world_size, band_rows)
    max_rows = max(sum(y1 - y0 for y0, y1 in b) for b in sched)
    pad = torch.zeros((max_rows, width, 3), dtype=torch.float32, device=local_rows.device)
    pad[:local_rows.shape[0]] = local_rows
    if world_size > 1:
        parts = [torch.empty_like(pad) for _ in range(world_size)]
        dist.all_gather(parts, pad, group=group)
    else:
        parts = [pad]
    frame = torch.empty((height, width, 3), dtype=torch.float32, device=local_rows.device)
    for r in range(world_size):
        k = 0
        for y0, y1 in sched[r]:
            frame[y0:y1] = parts[r][k:k + (y1 - y0)]
            k += y1 - y0
    return frame
