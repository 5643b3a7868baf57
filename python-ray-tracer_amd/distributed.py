"""Multi-GPU frame assembly: the (3,w,h) frame is cut along the width axis into one contiguous
column slab per rank (every pixel is independent — kernels.py:10-26 — and the scene is replicated),
each rank renders its slab with its own x offset, and the frame is assembled on rank 0 by a gather
(RCCL over xGMI when the process group is `nccl`; `gloo` on CPU for tests).

With equal slabs (w divisible by 8*world, e.g. 1920 on 1/2/4/8 GPUs) the exchange is ONE gather per
frame into a (world, 3, ws, h) staging buffer followed by one strided device copy into the (3,w,h)
frame; it can be issued asynchronously so that frame i is gathered while frame i+1 is rendered
(FrameGatherer, double-buffered).  Ragged slabs fall back to per-plane point-to-point transfers
straight into the frame's contiguous column runs.
"""


def slab_bounds(w, world_size, rank, align=8):
    """Columns [x0,x1) of rank `rank`: tile-aligned (8 columns = one wavefront tile) contiguous slabs
    whose sizes differ by at most one tile; the last non-empty slab absorbs the unaligned remainder."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    tiles = (w + align - 1) // align
    base, extra = divmod(tiles, world_size)
    t0 = rank * base + min(rank, extra)
    t1 = t0 + base + (1 if rank < extra else 0)
    return min(t0 * align, w), min(t1 * align, w)


class FrameGatherer:
    """Assembles per-rank slabs of shape (3, x1-x0, h) into (3,w,h) frames on rank `dst`.

    submit(slab, slot) starts the exchange for one frame (asynchronously where the backend allows) and
    finish(slot) completes it and returns the frame on `dst` (None elsewhere).  `slots` exchanges may be in
    flight; a slot's slab must not be overwritten between its submit() and finish().

    batch=F > 1 moves F frames per exchange: slabs are (F, 3, x1-x0, h), finish() returns (F, 3, w, h).  A
    collective costs tens of microseconds however small it is, a slab of a 1080p frame renders in less, so a
    sequence of frames is assembled F at a time (fewer, larger collectives)."""

    def __init__(self, w, h, dtype, device, dist, dst=0, slots=2, batch=1):
        import torch
        self.torch, self.dist, self.dst = torch, dist, dst
        self.w, self.h, self.batch = w, h, int(batch)
        assert self.batch >= 1
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.bounds = [slab_bounds(w, self.world, r) for r in range(self.world)]
        widths = {b - a for a, b in self.bounds}
        self.equal = len(widths) == 1
        self.ws = self.bounds[self.rank][1] - self.bounds[self.rank][0]
        self.pending = [None] * slots
        self.frames = self.stage = None
        if self.rank == dst:
            self.frames = [torch.empty((self.batch, 3, w, h), dtype=dtype, device=device) for _ in range(slots)]
            if self.equal:
                self.stage = [torch.empty((self.world, self.batch, 3, self.ws, h), dtype=dtype, device=device) for _ in range(slots)]

    def submit(self, slab, slot):
        assert self.pending[slot] is None, "slot still in flight: call finish(slot) first"
        if self.batch == 1 and slab.dim() == 3:
            slab = slab.unsqueeze(0)
        assert tuple(slab.shape) == (self.batch, 3, self.ws, self.h) and slab.is_contiguous()
        dist, root = self.dist, self.rank == self.dst
        if self.equal:
            recv = [self.stage[slot][r] for r in range(self.world)] if root else None
            self.pending[slot] = ("gather", dist.gather(slab, recv, dst=self.dst, async_op=True))
            return
        reqs = []                                       # ragged: plane-wise straight into the frame
        for j in range(self.batch):
            for c in range(3):
                if root:
                    for r, (a, b) in enumerate(self.bounds):
                        if r == self.dst:
                            self.frames[slot][j, c, a:b].copy_(slab[j, c])
                        elif b > a:
                            reqs.append(dist.irecv(self.frames[slot][j, c, a:b], src=r))
                elif self.ws:
                    reqs.append(dist.isend(slab[j, c], dst=self.dst))
        self.pending[slot] = ("p2p", reqs)

    def finish(self, slot):
        kind, work = self.pending[slot]
        self.pending[slot] = None
        if kind == "gather":
            work.wait()
            if self.rank != self.dst:
                return None
            f = self.frames[slot]
            f.view(self.batch, 3, self.world, self.ws, self.h).copy_(self.stage[slot].permute(1, 2, 0, 3, 4))
        else:
            for q in work:
                q.wait()
            if self.rank != self.dst:
                return None
            f = self.frames[slot]
        return f[0] if self.batch == 1 else f


def gather_frame(slab, w, h, dist, dst=0):
    """Synchronous convenience wrapper: gather one frame; returns it on `dst`, None elsewhere."""
    if dist.get_world_size() == 1:
        return slab
    g = FrameGatherer(w, h, slab.dtype, slab.device, dist, dst=dst, slots=1)
    g.submit(slab.contiguous(), 0)
    return g.finish(0)
