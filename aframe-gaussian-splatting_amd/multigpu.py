"""Frames over the GPUs of one node: who renders what, and the torch.distributed stand-in of the gather.

One process per GPU.  The splat buffer is replicated, every rank runs the bit-exact sort for the same view and renders
only its own PIECES of the frame: tile-aligned column strips of the one viewport, or -- XR -- the two eyes divided
between the ranks (eye k -> rank k at world 2, index.js:13-15 and :441: one shared sort from the head camera).  The
pieces are gathered on rank 0, the only exchange step of the path.

The product path is in the C library: `gs_partition` decides the pieces, `gs_render_gathered` renders and gathers them
over RCCL on the frame's own HIP stream (capi.Context.render_gathered; bench.py uses it for N > 1).  This module mirrors
the gather with torch.distributed collectives for the CPU test tier (backend gloo, where there is no GPU and no RCCL) and
for bench.py's GS_BENCH_TORCH_GATHER=1 fallback; the partition always comes from the C library.
"""
TILE = 16


def partition(widths, world):
    """[(view, x0, x1, owner)] in gather order: gs_partition of the C library (one source of truth)."""
    from . import capi
    return capi.partition(widths, world)


def strip_bounds(width, world, rank):
    """[x0, x1) of rank's strip of a single viewport (x0 == x1: more ranks than tile columns, nothing to render)."""
    for _, x0, x1, owner in partition([width], world):
        if owner == rank:
            return x0, x1
    return width, width


def strip_widths(width, world):
    return [b - a for a, b in (strip_bounds(width, world, r) for r in range(world))]


def gather_strips(strip_flat, width, height, dist, gathered=None, dst=0):
    """Gather every rank's strip (a flat uint8 tensor holding tight H x w_r x 4 rows at its front, padded to the
    widest strip so all messages have one size) to `dst` and assemble the row-major H x W x 4 frame there.
    Returns the frame on dst, None elsewhere.  Both steps are queued on the CURRENT torch stream."""
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


def piece_buffer_bytes(widths, heights, world):
    """Bytes of the per-rank message when pieces are gathered with one fixed-size collective: the largest total any rank owns."""
    per_rank = {}
    for v, x0, x1, owner in partition(widths, world):
        per_rank[owner] = per_rank.get(owner, 0) + (x1 - x0) * heights[v] * 4
    return max(per_rank.values())


def gather_views(local_flat, widths, heights, dist, dst=0):
    """The XR / multi-view form of gather_strips: every rank has its own pieces back to back (partition order) at the front
    of `local_flat`; rank `dst` gets the list of assembled H_v x W_v x 4 images, the others None."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    gathered = [torch.empty_like(local_flat) for _ in range(world)] if rank == dst else None
    dist.gather(local_flat, gathered, dst=dst)
    if rank != dst:
        return None
    frames = [torch.zeros(heights[v], widths[v], 4, dtype=torch.uint8, device=local_flat.device) for v in range(len(widths))]
    cursor = [0] * world
    for v, x0, x1, owner in partition(widths, world):
        nb = (x1 - x0) * heights[v] * 4
        frames[v][:, x0:x1] = gathered[owner][cursor[owner]: cursor[owner] + nb].view(heights[v], x1 - x0, 4)
        cursor[owner] += nb
    return frames
