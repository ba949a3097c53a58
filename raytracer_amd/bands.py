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

    if world_size == 1:
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

    if world_size == 1:
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
