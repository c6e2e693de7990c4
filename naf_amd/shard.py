"""Multi-GPU sharding of one archive: every rank produces a contiguous byte range of
the output text with naf_gpu_unnaf_range; the ranges are gathered to one rank with a single collective
(RCCL all_gather over xGMI on GPUs; the same code runs on gloo for the CPU tests).  No other exchange
is needed: record / line / mask context is recomputed per rank from the (small) side streams."""
import torch
import torch.distributed as dist


def byte_range(total: int, rank: int, world: int, align: int = 4096):
    """Contiguous [begin, end) of rank's share; boundaries aligned so each rank writes whole 4 KiB tiles."""
    per = (total + world - 1) // world
    per = (per + align - 1) // align * align
    b = min(total, rank * per)
    e = min(total, b + per)
    return b, e


def gather_ranges(local: torch.Tensor, total: int, dst: int = 0, group=None):
    """Concatenate every rank's range on `dst` (None elsewhere).  One all_gather on equal-size padded
    segments: xGMI is point-to-point, so one large collective beats many small sends."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = max(byte_range(total, r, world)[1] - byte_range(total, r, world)[0] for r in range(world))
    seg = torch.zeros(per, dtype=torch.uint8, device=local.device)
    seg[: local.numel()] = local
    out = [torch.empty(per, dtype=torch.uint8, device=local.device) for _ in range(world)]
    dist.all_gather(out, seg, group=group)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        b, e = byte_range(total, r, world)
        parts.append(out[r][: e - b])
    return torch.cat(parts)


def unnaf_sharded(ctx, d_naf, out_type=0, use_mask=True, line_length=-1, dst=0, group=None):
    """Each rank holds the archive (it is ~25 % of the text); returns the whole text on `dst`."""
    total = ctx.unnaf_size(d_naf, out_type, use_mask, line_length)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    b, e = byte_range(total, rank, world)
    local = ctx.unnaf_range(d_naf, b, e, out_type, use_mask, line_length)
    if world == 1:
        return local
    return gather_ranges(local, total, dst, group)
