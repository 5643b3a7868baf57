"""Multi-GPU frame assembly: the (3,w,h) frame is cut along the width axis into one contiguous
column slab per rank (every pixel is independent — kernels.py:10-26 — and the scene is replicated),
each rank renders its slab with its own x offset, and the frame is assembled on rank 0 by a gather
(RCCL over xGMI when the process group is `nccl`; `gloo` on CPU for tests).

Because the frame is C-ordered with x as the slower spatial axis, rank r's part of colour plane c is
one contiguous run of (x1-x0)*h elements at offset c*w*h + x0*h, so each plane is gathered straight
into place with no permutation pass afterwards.
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


def gather_frame(slab, w, h, dist, dst=0):
    """Gather per-rank slabs (torch tensors of shape (3, x1-x0, h)) into a (3,w,h) frame on `dst`.
    One gather per colour plane, each receiving directly into the frame's contiguous column run."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    if world == 1:
        return slab
    frame = None
    if rank == dst:
        frame = torch.empty((3, w, h), dtype=slab.dtype, device=slab.device)
    for c in range(3):
        recv = None
        if rank == dst:
            recv = [frame[c, a:b] for a, b in (slab_bounds(w, world, r) for r in range(world))]
        if all(b - a == slab.shape[1] for a, b in (slab_bounds(w, world, r) for r in range(world))):
            dist.gather(slab[c].contiguous(), recv, dst=dst)
        else:  # ragged slabs: gather requires equal sizes, so fall back to point-to-point
            if rank == dst:
                reqs = []
                for r in range(world):
                    if r == dst:
                        recv[r].copy_(slab[c])
                    elif recv[r].numel():
                        reqs.append(dist.irecv(recv[r], src=r))
                for q in reqs:
                    q.wait()
            elif slab[c].numel():
                dist.send(slab[c].contiguous(), dst=dst)
    return frame
