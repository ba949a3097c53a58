"""Scanline-band partition of a frame over ranks and the gather of the
finished bands to rank 0 through torch.distributed (backend "nccl" is RCCL on ROCm;
"gloo" in the CPU tests).  Pixels are independent (Renderer.cpp:393-467 touches only
accumulated_data[index] / target[index]) and the RNG counter uses the global pixel index,
so an N-band render is bit-identical to a one-device render.
"""


def band_rows(height, world_size, rank):
    """Rows [begin, end) of band `rank`: contiguous, in order, sizes differ by at most one."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} / world_size {world_size}")
    if height < world_size:
        raise ValueError(f"height {height} < world_size {world_size}: a band would be empty")
    return rank * height // world_size, (rank + 1) * height // world_size


# Rehearsal switch: run the collectives even when world_size == 1 (a one-rank RCCL gather), so
# the exact torch.distributed calls can be exercised on a one-GPU box.  Off in normal use.
FORCE_COLLECTIVE = False

STRIPE_ROWS = 15   # 1080, 2160 and 4320 are multiples of 15 * {1, 2, 4, 8}


def stripe_rows_for(height, world_size, stripe=STRIPE_ROWS):
    """Stripe height to interleave with, or 0 when the frame does not divide evenly (then the
    caller uses contiguous bands)."""
    return stripe if world_size > 1 and height % (stripe * world_size) == 0 else 0


def striped_rows(height, world_size, rank, stripe):
    """Frame rows of `rank` under interleaved stripes, in the order the context stores them."""
    if height % (stripe * world_size):
        raise ValueError("height must be a multiple of stripe * world_size")
    return [y for q in range(height // (stripe * world_size))
            for y in range((q * world_size + rank) * stripe, (q * world_size + rank + 1) * stripe)]


def gather_striped(band, height, width, rank, world_size, stripe, dst=0):
    """Gather interleaved-stripe bands on `dst` and put the stripes back in frame order.
    Every band has height/world_size rows, so this is one equal-size gather plus a permute."""
    import torch
    import torch.distributed as dist

    if world_size == 1 and not FORCE_COLLECTIVE:
        return band
    gathered = torch.empty((world_size,) + tuple(band.shape), dtype=band.dtype, device=band.device) if rank == dst else None
    dist.gather(band, gather_list=list(gathered.unbind(0)) if rank == dst else None, dst=dst)
    if rank != dst:
        return None
    q = height // (stripe * world_size)
    return gathered.view(world_size, q, stripe, width).permute(1, 0, 2, 3).reshape(height, width)


def gather_frame(band, height, width, rank, world_size, dst=0):
    """Gather every rank's band tensor ([rows, width], int32 view of 0x00RRGGBB) on `dst`.
    Returns the [height, width] frame on dst and None elsewhere.  One exchange step, no
    reduction: each peer sends its band over its own link into the root."""
    import torch
    import torch.distributed as dist

    if world_size == 1 and not FORCE_COLLECTIVE:
        return band
    sizes = [band_rows(height, world_size, r) for r in range(world_size)]
    if len({e - b for b, e in sizes}) == 1:
        frame = torch.empty((height, width), dtype=band.dtype, device=band.device) if rank == dst else None
        chunks = list(frame.view(world_size, -1, width).unbind(0)) if rank == dst else None
        dist.gather(band, gather_list=chunks, dst=dst)
        return frame
    # ragged bands: point-to-point into the root's frame
    if rank == dst:
        frame = torch.empty((height, width), dtype=band.dtype, device=band.device)
        reqs = []
        for r, (b, e) in enumerate(sizes):
            if r == dst:
                frame[b:e].copy_(band)
            else:
                reqs.append(dist.irecv(frame[b:e], src=r))
        for q in reqs:
            q.wait()
        return frame
    dist.send(band, dst=dst)
    return None


class PipelinedGather:
    """Two band buffers per rank so that the gather of frame i (RCCL, its own stream) overlaps
    the render of frame i+1.  Usage per frame:  b = pg.acquire(i); <render into b>; pg.submit(i).
    acquire() makes the current stream wait for the gather that last used that buffer, so the
    renderer never overwrites a band that is still being sent.  finish() drains and, on dst,
    returns the last frame in row order."""

    def __init__(self, height, width, rank, world_size, stripe, device, rows=None, dst=0):
        import torch
        self.h, self.w, self.rank, self.world, self.stripe, self.dst = height, width, rank, world_size, stripe, dst
        if world_size > 1 and height % world_size:
            raise ValueError("pipelined gather needs equal bands")
        rows = height // world_size if rows is None else rows
        self.bufs = [torch.empty((rows, width), dtype=torch.int32, device=device) for _ in range(2)]
        self.collective = world_size > 1 or FORCE_COLLECTIVE
        self.gathered = ([torch.empty((world_size, rows, width), dtype=torch.int32, device=device) for _ in range(2)]
                         if rank == dst and self.collective else [None, None])
        self.works = [None, None]
        self.last = None

    def acquire(self, i):
        b = i & 1
        if self.works[b] is not None:
            self.works[b].wait()
            self.works[b] = None
        return self.bufs[b]

    def submit(self, i):
        import torch.distributed as dist
        b = i & 1
        self.last = b
        if not self.collective:
            return
        g = self.gathered[b]
        self.works[b] = dist.gather(self.bufs[b], gather_list=list(g.unbind(0)) if g is not None else None,
                                    dst=self.dst, async_op=True)

    def finish(self):
        for b in (0, 1):
            if self.works[b] is not None:
                self.works[b].wait()
                self.works[b] = None
        if self.last is None:
            return None
        if not self.collective:
            return self.bufs[self.last]
        if self.rank != self.dst:
            return None
        g = self.gathered[self.last]
        if g.numel() != self.h * self.w:       # bench.py --emulate-world: one band of a larger frame
            return g
        if self.stripe:
            q = self.h // (self.stripe * self.world)
            return g.view(self.world, q, self.stripe, self.w).permute(1, 0, 2, 3).reshape(self.h, self.w)
        return g.reshape(self.h, self.w)
