"""Column-strip decomposition of the viewport across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
The splat buffer is replicated, every rank runs the bit-exact sort and the projection for all splats, bins and
blends only the tiles of its own strip, and the strips are gathered to rank 0 -- the only exchange step of the
path.  Strips are tile-aligned (16 px) so that no tile is shared between ranks.
"""
TILE = 16


def strip_bounds(width, world, rank):
    """[x0, x1) of rank's strip: whole 16-px tile columns, as even as possible; the last strip takes the ragged edge."""
    tiles = (width + TILE - 1) // TILE
    t0, t1 = tiles * rank // world, tiles * (rank + 1) // world
    return t0 * TILE, min(t1 * TILE, width)


def strip_widths(width, world):
    return [b - a for a, b in (strip_bounds(width, world, r) for r in range(world))]


def gather_strips(strip_flat, width, height, dist, gathered=None, dst=0):
    """Gather every rank's strip (a flat uint8 tensor holding tight H x w_r x 4 rows at its front, padded to the
    widest strip so all messages have one size) to `dst` and assemble the row-major H x W x 4 frame there.
    Returns the frame on dst, None elsewhere.  Both steps are queued on the CURRENT torch stream: bench.py makes that
    the stream of the pipeline lane the frame was rendered on (Context.frame_stream), so the gather follows the frame's
    blend and precedes the next frame on that lane with no cross-stream event."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if rank == dst and gathered is None:
        gathered = [torch.empty_like(strip_flat) for _ in range(world)]
    dist.gather(strip_flat, gathered if rank == dst else None, dst=dst)
    if rank != dst:
        return None
    widths = strip_widths(width, world)
    return torch.cat([g[: height * w * 4].view(height, w, 4) for g, w in zip(gathered, widths) if w > 0], dim=1)


def strip_buffer_bytes(width, height, world):
    return height * max(strip_widths(width, world)) * 4
