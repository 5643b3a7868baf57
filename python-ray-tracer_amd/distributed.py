"""Multi-GPU frame assembly: the (3,w,h) frame is cut along the width axis into one contiguous
column slab per rank (every pixel is independent — kernels.py:10-26 — and the scene is replicated),
each rank renders its slab with its own x offset, and the frame is assembled on rank 0 by a gather
(RCCL over xGMI when the process group is `nccl`; `gloo` on CPU for tests).

The exchange is ONE gather per batch of frames into a (world, F, 3, ws, h) staging buffer followed by a
strided device copy into the (F,3,w,h) frames; it is issued asynchronously so that batch i is gathered while
batch i+1 is rendered (FrameGatherer, double-buffered).  Slabs may be unequal — cost-weighted boundaries
(weighted_slab_bounds, SlabBalancer) or a width that does not divide: every rank then renders into a buffer
padded to the widest slab, so that it is still one gather.
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


def weighted_slab_bounds(tile_col_cost, w, world_size, align=8):
    """Tile-aligned contiguous column slabs of (nearly) equal COST instead of equal width.

    tile_col_cost[i] > 0 is the cost of tile column i (`align` pixel columns; e.g. column sums of the per-tile
    cycles rt_set_tile_stats records, or the density SlabBalancer refines from measured slab times).  Every pixel
    is independent (kernels.py:10-26), so any partition yields the same frame; only the ranks' finishing times
    depend on it.  Boundary r is placed where the running cost passes r/world of the total; every rank keeps at
    least one tile column while there are enough of them.  Returns [(x0, x1)] * world_size."""
    cost = [max(float(c), 0.0) for c in tile_col_cost]
    tiles = (w + align - 1) // align
    if len(cost) != tiles:
        raise ValueError(f"expected {tiles} tile-column costs, got {len(cost)}")
    total = sum(cost)
    if not total > 0.0:
        return [slab_bounds(w, world_size, r, align) for r in range(world_size)]
    cuts, run, i = [0], 0.0, 0
    for r in range(1, world_size):
        target = total * r / world_size
        while i < tiles and run + cost[i] * 0.5 < target:      # cut at the tile boundary nearest to the target
            run += cost[i]
            i += 1
        lo = cuts[-1] + (1 if tiles - cuts[-1] >= world_size - r + 1 else 0)    # leave this rank a tile ...
        hi = max(lo, tiles - (world_size - r))                                   # ... and one for each rank behind it
        cuts.append(min(max(i, lo), hi))
        while i < cuts[-1]:
            run += cost[i]
            i += 1
    cuts.append(tiles)
    return [(min(a * align, w), min(b * align, w)) for a, b in zip(cuts[:-1], cuts[1:])]


class SlabBalancer:
    """Refines a per-tile-column cost density from MEASURED slab times: after every round, the density inside each
    rank's slab is rescaled so that the slab's total equals the time that rank measured, and the boundaries are
    re-cut for equal totals.  A few rounds bring the slowest rank within a few percent of the mean (contiguous equal
    slabs of the headline frame differ by 23 %).  All ranks must call update() with the same list of times."""

    def __init__(self, w, world_size, tile_col_cost=None, align=8):
        self.w, self.world, self.align = w, world_size, align
        tiles = (w + align - 1) // align
        self.density = [1.0] * tiles if tile_col_cost is None else [max(float(c), 1e-12) for c in tile_col_cost]
        self.bounds = weighted_slab_bounds(self.density, w, world_size, align)

    def update(self, slab_times):
        for (a, b), t in zip(self.bounds, slab_times):
            ta, tb = a // self.align, (b + self.align - 1) // self.align
            tot = sum(self.density[ta:tb])
            if tb > ta and tot > 0 and t > 0:
                k = float(t) / tot
                for i in range(ta, tb):
                    self.density[i] *= k
        self.bounds = weighted_slab_bounds(self.density, self.w, self.world, self.align)
        return self.bounds


class FrameGatherer:
    """Assembles per-rank slabs of shape (3, x1-x0, h) into (3,w,h) frames on rank `dst`.

    submit(slab, slot) starts the exchange for one frame (asynchronously where the backend allows) and
    finish(slot) completes it and returns the frame on `dst` (None elsewhere).  `slots` exchanges may be in
    flight; a slot's slab must not be overwritten between its submit() and finish().

    batch=F > 1 moves F frames per exchange: slabs are (F, 3, x1-x0, h), finish() returns (F, 3, w, h).  A
    collective costs tens of microseconds however small it is, a slab of a 1080p frame renders in less, so a
    sequence of frames is assembled F at a time (fewer, larger collectives)."""

    def __init__(self, w, h, dtype, device, dist, dst=0, slots=2, batch=1, bounds=None, any_root=False):
        import torch
        self.torch, self.dist, self.dst = torch, dist, dst
        self.any_root = bool(any_root)     # submit(..., dst=r) may name any rank: every rank holds frame and staging buffers
        self.w, self.h, self.batch = w, h, int(batch)
        assert self.batch >= 1
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.bounds = [tuple(b) for b in bounds] if bounds is not None else [slab_bounds(w, self.world, r) for r in range(self.world)]
        assert len(self.bounds) == self.world and self.bounds[0][0] == 0 and self.bounds[-1][1] == w
        assert all(self.bounds[i][1] == self.bounds[i + 1][0] for i in range(self.world - 1)), "slabs must tile the width"
        widths = [b - a for a, b in self.bounds]
        self.equal = len(set(widths)) == 1
        self.ws = widths[self.rank]
        # ragged slabs (weighted boundaries, or a width that does not divide): every rank's slab lives in a buffer
        # padded to the widest slab, so that the exchange is still ONE gather; the padding columns travel unused
        self.ws_pad = max(widths)
        self.pending = [None] * slots
        self.root = [dst] * slots          # the rank each slot's exchange in flight goes to
        self.frames = self.stage = None
        if self.rank == dst or self.any_root:
            self.frames = [torch.empty((self.batch, 3, w, h), dtype=dtype, device=device) for _ in range(slots)]
            self.stage = [torch.empty((self.world, self.batch, 3, self.ws_pad, h), dtype=dtype, device=device) for _ in range(slots)]

    def submit(self, slab, slot, dst=None):
        """slab: (batch, 3, ws_pad, h) contiguous; columns [0, ws) of it are this rank's pixels (ws_pad == ws for
        equal slabs).  Render with plane_stride = ws_pad * h to fill it in place.  dst (any_root only): the rank
        that assembles THIS exchange — all ranks must pass the same one."""
        assert self.pending[slot] is None, "slot still in flight: call finish(slot) first"
        if self.batch == 1 and slab.dim() == 3:
            slab = slab.unsqueeze(0)
        assert tuple(slab.shape) == (self.batch, 3, self.ws_pad, self.h) and slab.is_contiguous()
        dst = self.dst if dst is None else int(dst)
        assert dst == self.dst or self.any_root
        self.root[slot] = dst
        recv = [self.stage[slot][r] for r in range(self.world)] if self.rank == dst else None
        self.pending[slot] = self.dist.gather(slab, recv, dst=dst, async_op=True)

    def finish(self, slot):
        work = self.pending[slot]
        self.pending[slot] = None
        work.wait()
        if self.rank != self.root[slot]:
            return None
        f = self.frames[slot]
        if self.equal:
            f.view(self.batch, 3, self.world, self.ws, self.h).copy_(self.stage[slot].permute(1, 2, 0, 3, 4))
        else:
            for r, (a, b) in enumerate(self.bounds):
                if b > a:
                    f[:, :, a:b].copy_(self.stage[slot][r][:, :, : b - a])
        return f[0] if self.batch == 1 else f


def gather_frame(slab, w, h, dist, dst=0, bounds=None):
    """Synchronous convenience wrapper: gather one frame from compact (3, ws, h) slabs; returns it on `dst`, None
    elsewhere."""
    if dist.get_world_size() == 1:
        return slab
    g = FrameGatherer(w, h, slab.dtype, slab.device, dist, dst=dst, slots=1, bounds=bounds)
    if g.ws_pad != g.ws:
        padded = slab.new_zeros((3, g.ws_pad, h))
        padded[:, : g.ws] = slab
        slab = padded
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

    submit(launch) queues one frame: launch(u8, f32, stream) must enqueue the rendering of this rank's slab into
    columns [0, ws) of the (3, ws_pad, h) tensors `u8` / `f32` (plane stride = self.plane_stride elements; ws_pad = ws
    unless the ranks' slabs are unequal) on `stream` (a raw stream handle, or None on CPU, where it runs synchronously).
    drain() completes everything queued.  on_frames(first_index, frames, count[, frames_f32 with gather_f32]), if given, is called on `dst` (with
    rotate_root: on the batch's own root) for every assembled batch (`frames` is (F, 3, w, h); only the first `count` are new).  Without a process group
    (dist=None) nothing is exchanged and on_frames is not called; last_slab() returns the newest slab.

    On a CUDA/HIP device this uses torch streams and events; on CPU (the gloo tests) everything is synchronous."""

    def __init__(self, w, h, ws, device, dist=None, dst=0, streams=3, frames_per_gather=8, want_f32=True, on_frames=None,
                 bounds=None, rotate_root=False, frames_per_launch=1, gather_f32=False):
        import torch
        self.torch, self.dist, self.dst, self.on_frames = torch, dist, dst, on_frames
        # rotate_root: batch b is assembled on rank (dst + b) % world instead of always on `dst`, and on_frames is called
        # THERE.  With every frame converging on one rank, that rank's inbound links carry (world-1)/world of every frame;
        # at 8 GPUs and a 15 us slab that is more than they deliver, while rotating spreads it over all ranks' links.
        self.rotate_root = bool(rotate_root) and dist is not None
        self.batches = 0                        # batches closed so far (all ranks count alike)
        self.gpu = torch.device(device).type == "cuda"
        self.NS = max(1, int(streams)) if self.gpu else 1
        # F = frames per batch: the frames one gather moves (with a process group) / one launch of submit_frames() renders
        self.F = max(1, int(frames_per_gather)) if dist is not None else max(1, int(frames_per_launch))
        self.SLOTS = 2 if dist is not None else self.NS
        self.streams = [torch.cuda.Stream(device=device) for _ in range(self.NS)] if self.gpu else [None]
        self.comm = torch.cuda.Stream(device=device) if (self.gpu and dist is not None) else None
        self.gatherer = FrameGatherer(w, h, torch.uint8, device, dist, dst=dst, slots=self.SLOTS, batch=self.F, bounds=bounds,
                                      any_root=self.rotate_root) if dist is not None else None
        # gather_f32: the float32 pre-clip planes (the artefact north_star's 1e-5 tolerance is stated on) are assembled too,
        # by a second padded gather per batch (4x the bytes of the uint8 one; off by default — bench.py --gather f32)
        self.gatherer32 = FrameGatherer(w, h, torch.float32, device, dist, dst=dst, slots=self.SLOTS, batch=self.F, bounds=bounds,
                                        any_root=self.rotate_root) if (dist is not None and gather_f32 and want_f32) else None
        if self.gatherer is not None:
            assert self.gatherer.ws == ws, "ws must be this rank's slab width under `bounds`"
        # slabs are stored padded to the widest rank's width (one gather even when the slabs are unequal): render
        # with plane_stride = self.plane_stride
        self.ws, self.ws_pad = ws, (self.gatherer.ws_pad if self.gatherer is not None else ws)
        self.plane_stride = self.ws_pad * h
        self.u8 = [torch.zeros((self.F, 3, self.ws_pad, h), dtype=torch.uint8, device=device) for _ in range(self.SLOTS)]
        self.f32 = [torch.zeros((self.F, 3, self.ws_pad, h), dtype=torch.float32, device=device) for _ in range(self.SLOTS)] if want_f32 else None
        # per-frame views, made once: indexing a tensor costs microseconds, and a 1080p slab renders in ~100
        self.u8v = [[t[j] for j in range(self.F)] for t in self.u8]
        self.f32v = [[t[j] for j in range(self.F)] for t in self.f32] if want_f32 else None
        self.handles = [s.cuda_stream if s is not None else None for s in self.streams]
        self.first = [None] * self.SLOTS        # index of the first frame of the batch in flight in each slot
        self.count = [0] * self.SLOTS
        self.n = 0                              # position in the slot/batch cycle (padded to a batch boundary by drain())
        self.index = 0                          # frames submitted so far
        self._last = self._last32 = None

    def stream_handle(self, i):
        return self.handles[i % self.NS]

    def _collect(self, slot):
        ctx = self.torch.cuda.stream(self.comm) if self.comm is not None else _nullcontext()
        with ctx:                               # the gather's stream dependencies follow torch's current stream
            f = self.gatherer.finish(slot)
            f32 = self.gatherer32.finish(slot) if self.gatherer32 is not None else None
            if f is not None and self.on_frames is not None:
                if self.gatherer32 is not None:
                    self.on_frames(self.first[slot], f if self.F > 1 else f.unsqueeze(0), self.count[slot],
                                   f32 if self.F > 1 else f32.unsqueeze(0))
                else:
                    self.on_frames(self.first[slot], f if self.F > 1 else f.unsqueeze(0), self.count[slot])
        self.first[slot] = None

    def _close_batch(self, slot):
        if self.comm is not None:
            for t in self.streams:
                self.comm.wait_event(t.record_event())
        ctx = self.torch.cuda.stream(self.comm) if self.comm is not None else _nullcontext()
        root = (self.dst + self.batches) % self.dist.get_world_size() if self.rotate_root else self.dst
        self.batches += 1
        with ctx:
            self.gatherer.submit(self.u8[slot], slot, dst=root)
            if self.gatherer32 is not None:
                self.gatherer32.submit(self.f32[slot], slot, dst=root)

    def _open_batch(self, slot, idx):
        if self.first[slot] is not None:        # the slot's slabs are reused: its exchange must have completed
            self._collect(slot)
            if self.comm is not None:
                ev = self.comm.record_event()
                for t in self.streams:
                    t.wait_event(ev)
        self.first[slot], self.count[slot] = idx, 0

    def submit(self, launch):
        i = self.n
        self.n += 1
        idx = self.index
        self.index += 1
        handle = self.handles[i % self.NS]
        slot, j = (i // self.F) % self.SLOTS, i % self.F
        if self.gatherer is None:
            self._last = self.u8v[slot][j]
            self._last32 = self.f32v[slot][j] if self.f32v is not None else None
            launch(self._last, self._last32, handle)
            return
        if j == 0:
            self._open_batch(slot, idx)
        self._last = self.u8v[slot][j]
        self._last32 = self.f32v[slot][j] if self.f32v is not None else None
        launch(self._last, self._last32, handle)
        self.count[slot] += 1
        if j == self.F - 1:                     # the batch is complete: one gather
            self._close_batch(slot)

    def submit_frames(self, launch_seq, count):
        """Queue `count` frames, a whole batch (F frames) per call of launch_seq(u8, f32, nframes, stream): `u8` / `f32` are
        (nframes, 3, ws_pad, h) tensors (consecutive frames of one batch buffer: frame stride = 3 * plane_stride) and the
        callback renders all of them on `stream` — with Renderer.render_sequence that is ONE kernel launch per batch, so a
        settled frame costs the host a fraction of a launch (every frame a Python call and a launch of its own: ~10 us, more
        than a 1/8 slab of the headline frame takes to render).  Batches go round-robin over the streams.  Returns the
        (stream index, frames) of every launch made, in order."""
        made = []
        while count > 0:
            i = self.n
            b = i // self.F
            slot, j = b % self.SLOTS, i % self.F
            nf = min(count, self.F - j)
            si = b % self.NS
            if self.gatherer is not None and j == 0:
                self._open_batch(slot, self.index)
            whole = (j == 0 and nf == self.F)
            u8 = self.u8[slot] if whole else self.u8[slot][j:j + nf]
            f32 = None if self.f32 is None else (self.f32[slot] if whole else self.f32[slot][j:j + nf])
            launch_seq(u8, f32, nf, self.handles[si])
            self._last = self.u8v[slot][j + nf - 1]
            self._last32 = self.f32v[slot][j + nf - 1] if self.f32v is not None else None
            self.n += nf
            self.index += nf
            count -= nf
            made.append((si, nf))
            if self.gatherer is not None:
                self.count[slot] += nf
                if j + nf == self.F:
                    self._close_batch(slot)
        return made

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

    def last_slab_f32(self):
        return self._last32

    def close(self, renderer=None):
        """Before the torch streams go away: let the renderer's context forget their handles (it fences streams that
        launched on it when it rebuilds what they may still read; include/mi355rt.h: rt_stream_forget)."""
        if self.gpu:
            self.torch.cuda.synchronize()
            if renderer is not None:
                for hnd in self.handles + ([self.comm.cuda_stream] if self.comm is not None else []):
                    renderer.stream_forget(hnd)


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False
