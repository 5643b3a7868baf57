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


class SequencePipeline:
    """Renders a SEQUENCE of frames slab-parallel and assembles them on rank `dst` (what bench.py times).

    Two things decide the rate of a frame that takes a fraction of a millisecond:
      * consecutive frames must overlap on the device, so frames are queued round-robin on `streams` streams, each
        frame in flight with its own output buffers (with one in-order stream every frame pays its own ramp-up and
        tail, and a slab of 1/8 frame is less than one round of workgroups);
      * collectives must be few, so the uint8 slabs of `frames_per_gather` consecutive frames travel in ONE gather,
        issued on a separate stream behind events from the render streams, two exchanges in flight.

    submit(launch) queues one frame: launch(u8, f32, stream) must enqueue the rendering of this rank's slab into the
    (3, ws, h) tensors `u8` / `f32` on `stream` (a raw stream handle, or None on CPU, where it runs synchronously).
    drain() completes everything queued.  on_frames(first_index, frames, count), if given, is called on `dst` for
    every assembled batch (`frames` is (F, 3, w, h); only the first `count` are new).  Without a process group
    (dist=None) nothing is exchanged and on_frames is not called; last_slab() returns the newest slab.

    On a CUDA/HIP device this uses torch streams and events; on CPU (the gloo tests) everything is synchronous."""

    def __init__(self, w, h, ws, device, dist=None, dst=0, streams=3, frames_per_gather=8, want_f32=True, on_frames=None):
        import torch
        self.torch, self.dist, self.dst, self.on_frames = torch, dist, dst, on_frames
        self.gpu = torch.device(device).type == "cuda"
        self.NS = max(1, int(streams)) if self.gpu else 1
        self.F = max(1, int(frames_per_gather)) if dist is not None else 1
        self.SLOTS = 2 if dist is not None else self.NS
        self.streams = [torch.cuda.Stream(device=device) for _ in range(self.NS)] if self.gpu else [None]
        self.comm = torch.cuda.Stream(device=device) if (self.gpu and dist is not None) else None
        self.u8 = [torch.zeros((self.F, 3, ws, h), dtype=torch.uint8, device=device) for _ in range(self.SLOTS)]
        self.f32 = [torch.zeros((self.F, 3, ws, h), dtype=torch.float32, device=device) for _ in range(self.SLOTS)] if want_f32 else None
        self.gatherer = FrameGatherer(w, h, torch.uint8, device, dist, dst=dst, slots=self.SLOTS, batch=self.F) if dist is not None else None
        # per-frame views, made once: indexing a tensor costs microseconds, and a 1080p slab renders in ~100
        self.u8v = [[t[j] for j in range(self.F)] for t in self.u8]
        self.f32v = [[t[j] for j in range(self.F)] for t in self.f32] if want_f32 else None
        self.handles = [s.cuda_stream if s is not None else None for s in self.streams]
        self.first = [None] * self.SLOTS        # index of the first frame of the batch in flight in each slot
        self.count = [0] * self.SLOTS
        self.n = 0                              # position in the slot/batch cycle (padded to a batch boundary by drain())
        self.index = 0                          # frames submitted so far
        self._last = None

    def stream_handle(self, i):
        return self.handles[i % self.NS]

    def _collect(self, slot):
        ctx = self.torch.cuda.stream(self.comm) if self.comm is not None else _nullcontext()
        with ctx:                               # the gather's stream dependencies follow torch's current stream
            f = self.gatherer.finish(slot)
            if f is not None and self.on_frames is not None:
                self.on_frames(self.first[slot], f if self.F > 1 else f.unsqueeze(0), self.count[slot])
        self.first[slot] = None

    def _close_batch(self, slot):
        if self.comm is not None:
            for t in self.streams:
                self.comm.wait_event(t.record_event())
        ctx = self.torch.cuda.stream(self.comm) if self.comm is not None else _nullcontext()
        with ctx:
            self.gatherer.submit(self.u8[slot], slot)

    def submit(self, launch):
        i = self.n
        self.n += 1
        idx = self.index
        self.index += 1
        handle = self.handles[i % self.NS]
        if self.gatherer is None:
            b = i % self.SLOTS
            self._last = self.u8v[b][0]
            launch(self._last, self.f32v[b][0] if self.f32v is not None else None, handle)
            return
        slot, j = (i // self.F) % self.SLOTS, i % self.F
        if j == 0:
            if self.first[slot] is not None:    # the slot's slabs are reused: its exchange must have completed
                self._collect(slot)
                if self.comm is not None:
                    ev = self.comm.record_event()
                    for t in self.streams:
                        t.wait_event(ev)
            self.first[slot], self.count[slot] = idx, 0
        self._last = self.u8v[slot][j]
        launch(self._last, self.f32v[slot][j] if self.f32v is not None else None, handle)
        self.count[slot] += 1
        if j == self.F - 1:                     # the batch is complete: one gather
            self._close_batch(slot)

    def drain(self):
        """Complete every queued frame (a partly filled last batch is exchanged as it is)."""
        if self.gatherer is not None and self.n:
            cur = ((self.n - 1) // self.F) % self.SLOTS
            if self.first[cur] is not None and self.gatherer.pending[cur] is None:
                self._close_batch(cur)          # the sequence ended inside a batch
            for slot in [(cur + 1 + k) % self.SLOTS for k in range(self.SLOTS)]:     # oldest first
                if self.first[slot] is not None:
                    self._collect(slot)
            self.n = -(-self.n // self.F) * self.F      # the next frame starts a new batch
        if self.gpu:
            self.torch.cuda.synchronize()

    def last_slab(self):
        return self._last


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False
