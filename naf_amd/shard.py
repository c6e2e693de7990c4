"""One archive on several GPUs (BASELINE configs[3] and [4]).

Decode: every rank produces a contiguous byte range of the output text with naf_gpu_unnaf_range (record / line / mask context is
recomputed per rank from the small side streams) and the ranges are gathered to one rank: the root posts one receive per peer
straight into that peer's place in the output buffer, the peers send -- a gather-to-root of unequal segments as one group of
point-to-point transfers (ncclGroupStart / ncclSend / ncclRecv on RCCL: xGMI is point-to-point, each peer's segment travels over
its own link into the root).  No padding, no copy of the whole text on every rank.

Encode: the text is cut into one slice per rank, every rank runs naf_gpu_ennaf_shard_begin / _finish on its slice, the fixed-size
shard records are all-gathered in between, and the compressed parts are gathered into ONE archive with one zstd frame per stream
(include/naf_gpu.h, "ennaf of ONE input on several GPUs").

The same functions run on gloo with CPU tensors (tests/test_host_cpu.py drives them with a stand-in context)."""
import ctypes as C

import torch
import torch.distributed as dist

from . import capi

EOL = (0x0A, 0x0B, 0x0C, 0x0D)


# ---------------------------------------------------------------------------------------------------------------- decode
def byte_range(total: int, rank: int, world: int, align: int = 4096):
    """Contiguous [begin, end) of rank's share; boundaries aligned so each rank writes whole 4 KiB tiles."""
    per = (total + world - 1) // world
    per = (per + align - 1) // align * align
    b = min(total, rank * per)
    e = min(total, b + per)
    return b, e


def _staged(t, group):
    """Device tensors over a backend that only moves host memory (gloo: the two-process tests on a box with one GPU, a job without
    RCCL): the transfer goes through a host copy.  RCCL moves device memory itself."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


P2P_MAX_BYTES = 1 << 29      # a transfer of more than this goes as pieces of this size, in order, on both sides


def _pieces(ops):
    """Transfers of more than P2P_MAX_BYTES as several, in order (sender and receiver cut a range of one length the same way).  RCCL
    2.26's send / recv of a rank to ITSELF drops the second half of a message above 1 GiB (profiles/r06_rccl_self_exchange.txt: what
    `--force-sharded` on one GPU runs into); a rank's range of a 100 GB text is 12.5 GB, and nothing is lost by not finding out on an
    8-GPU box whether the path between two ranks has a limit of its own."""
    out = []
    for op in ops:
        t = op.tensor
        nb = t.numel() * t.element_size()
        if nb <= P2P_MAX_BYTES or t.dim() != 1 or not t.is_contiguous():
            out.append(op)
            continue
        step = P2P_MAX_BYTES // t.element_size()
        for i in range(0, t.numel(), step):
            out.append(dist.P2POp(op.op, t[i:i + step], op.peer, op.group))
    return out


def _p2p_post(ops):
    """Post a group of point-to-point transfers; returns what _p2p_wait needs.  Over RCCL the group is enqueued on the communicator's
    stream behind what the current stream holds NOW: kernels launched on the current stream after this call run beside the transfers."""
    if not ops:
        return None
    ops = _pieces(ops)
    back = []
    real = []
    for op in ops:
        if _staged(op.tensor, op.group):
            h = op.tensor.cpu() if op.op is dist.isend else torch.empty(op.tensor.shape, dtype=op.tensor.dtype)
            if op.op is dist.irecv:
                back.append((op.tensor, h))
            real.append(dist.P2POp(op.op, h, op.peer, op.group))
        else:
            real.append(op)
    return dist.batch_isend_irecv(real), back


def _p2p_wait(posted):
    if posted is None:
        return
    works, back = posted
    for w in works:
        w.wait()
    for d, h in back:
        d.copy_(h)


def _p2p(ops):
    _p2p_wait(_p2p_post(ops))


def _all_gather(outs, t, group):
    if _staged(t, group):
        hs = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        dist.all_gather(hs, t.cpu(), group=group)
        for o, h in zip(outs, hs):
            o.copy_(h)
    else:
        dist.all_gather(outs, t, group=group)


def _broadcast(t, src, group):
    if _staged(t, group):
        h = t.cpu()
        dist.broadcast(h, src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src, group=group)


def _self_exchange(local: torch.Tensor, dst_view: torch.Tensor, group=None):
    """A world of ONE rank has nobody to exchange with: under `self_exchange` the rank posts the SAME group of point-to-point
    transfers a root and a peer would post between them -- a receive into the range's place and a send of the decoded range, to
    itself -- so that the communicator's send / recv path (RCCL on a GPU box) moves the bytes once on the hardware there is.  What a
    test or `bench.py --force-sharded` on a one-GPU box can exercise of the N > 1 path; not a product mode."""
    _p2p([dist.P2POp(dist.irecv, dst_view, dist.get_rank(group), group), dist.P2POp(dist.isend, local, dist.get_rank(group), group)])


def gather_ranges(local: torch.Tensor, total: int, dst: int = 0, group=None, out: torch.Tensor = None, self_exchange: bool = False):
    """Gather-to-root of every rank's byte range: returns the whole text on `dst` (None elsewhere).  `out`: where the root
    wants it (at least `total` bytes); the root's own range is copied in place, the others are received in place."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1 and self_exchange and local.numel():
        if out is None:
            out = torch.empty(max(total, 1), dtype=torch.uint8, device=local.device)
        _self_exchange(local[:total], out[:total], group)
        return out[:total]
    if rank != dst:
        if local.numel():
            _p2p([dist.P2POp(dist.isend, local, dst, group)])
        return None
    if out is None:
        out = torch.empty(max(total, 1), dtype=torch.uint8, device=local.device)
    ops = []
    for r in range(world):
        b, e = byte_range(total, r, world)
        if e == b:
            continue
        if r == rank:
            out[b:e].copy_(local[: e - b])
        else:
            ops.append(dist.P2POp(dist.irecv, out[b:e], r, group))
    _p2p(ops)
    return out[:total]


def unnaf_sharded(ctx, d_naf, out_type=0, use_mask=True, line_length=-1, dst=0, group=None, out=None, total=None, scratch=None, self_exchange=False):
    """Each rank holds the archive (it is ~25 % of the text) and decodes its byte range of the text; the root decodes its own
    range straight into its place of `out` and receives the others in place.  Returns the whole text on `dst`, None elsewhere.
    `scratch`: buffer for a non-root rank's range (allocated when missing)."""
    if total is None:
        total = ctx.unnaf_size(d_naf, out_type, use_mask, line_length)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    b, e = byte_range(total, rank, world)
    if world == 1 and self_exchange and dist.is_initialized() and e > b:
        # (see _self_exchange: the range is decoded into scratch and travels through the communicator to its place)
        local = ctx.unnaf_range(d_naf, b, e, out_type, use_mask, line_length, out=scratch)
        if out is None:
            out = torch.empty(max(total, 1), dtype=torch.uint8, device=d_naf.device)
        _self_exchange(local, out[b:e], group)
        return out[:total]
    if world == 1:
        return ctx.unnaf_range(d_naf, b, e, out_type, use_mask, line_length, out=out)
    if rank != dst:
        if e > b:
            local = ctx.unnaf_range(d_naf, b, e, out_type, use_mask, line_length, out=scratch)
            _p2p([dist.P2POp(dist.isend, local, dst, group)])
        return None
    if out is None:
        out = torch.empty(max(total, 1), dtype=torch.uint8, device=d_naf.device)
    # the receives first, the root's own range beside them: the peers' ranges arrive over their own xGMI links while the root decodes
    # (the transfers take ~20 times what a range decode takes -- DESIGN.md section 6 -- so the root's decode is hidden entirely)
    ops = []
    for r in range(world):
        rb, re_ = byte_range(total, r, world)
        if r != rank and re_ > rb:
            ops.append(dist.P2POp(dist.irecv, out[rb:re_], r, group))
    posted = _p2p_post(ops)
    if e > b:
        ctx.unnaf_range(d_naf, b, e, out_type, use_mask, line_length, out=out[b:e])
    _p2p_wait(posted)
    return out[:total]


# ---------------------------------------------------------------------------------------------------------------- encode
def make_opts(fmt=capi.FMT_AUTO, seq_type=capi.SEQ_DNA, no_mask=False, strict=False, level=1, line_length=-1, title=None, long_log=0):
    return capi.EnnafOpts(fmt, seq_type, int(no_mask), int(strict), level, line_length, title, long_log)


def _is_eol(b):
    return int(b) in EOL


def cuts_local(ctx, d_text, fmt, p0, n_shards):
    """Start offset of every shard (n_shards + 1 entries, the last one = len) of a text one context can address: nominal equal
    slices moved forward to the next place a shard may begin (FASTA: behind an EOL; FASTQ: a line start whose ordinal is 0 mod 4)."""
    n = d_text.numel()
    cuts = [p0]
    for k in range(1, n_shards):
        a = p0 + (n - p0) * k // n_shards
        a = max(a, cuts[-1])
        if a >= n or a == 0:
            cuts.append(min(a, n) if a else cuts[-1])
            continue
        # the slice handed to the library starts one byte early with prev_is_eol = 0: that byte decides about position a itself
        if fmt == capi.FMT_FASTA:
            off = ctx.ennaf_find_cut(d_text[a - 1:], fmt, False)
        else:
            lines = ctx.ennaf_count_lines(d_text[p0:a], True) if a > p0 else 0
            off = ctx.ennaf_find_cut(d_text[a - 1:], fmt, False, (-lines) % 4)
        cuts.append(min(n, a - 1 + off))
    cuts.append(n)
    return cuts


def ennaf_sharded_local(ctxs, d_text, opts=None, out=None):
    """N contexts of one process (one per device, or several on one device): returns (archive, report).  Each context sees a
    view of the same text; with one device per context the caller places the slices (see naf_amd/host/ennaf.c)."""
    opts = opts or make_opts()
    c0 = ctxs[0]
    fmt, p0 = c0.ennaf_sniff(d_text, opts.format)
    if fmt == 0 or len(ctxs) == 1:
        return c0.ennaf(d_text, seq_type=opts.seq_type, fmt=opts.format, no_mask=bool(opts.no_mask), level=opts.level,
                        line_length=opts.line_length, title=opts.title, out=out, strict=bool(opts.strict), long_log=opts.long_log)
    n = len(ctxs)
    cuts = cuts_local(c0, d_text, fmt, p0, n)
    slices = [d_text[cuts[k]:cuts[k + 1]] for k in range(n)]
    infos = [ctxs[k].ennaf_shard_begin(slices[k], opts, fmt, k, n) for k in range(n)]
    bufs, pieces = [], []
    for k in range(n):
        b, pc = ctxs[k].ennaf_shard_finish(opts, infos, slices[k].numel())
        bufs.append(b)
        pieces.append(pc)
    segs, lit, naf_len, rep = capi.stitch_plan(opts, infos, pieces)
    if out is None:
        out = torch.empty(max(naf_len, 1), dtype=torch.uint8, device=d_text.device)
    c0.ennaf_stitch(segs, lit, bufs, out)
    return out[:naf_len], rep


class _Lap:
    """NAF_SHARD_TIMING=1: wall-clock of the steps of ennaf_sharded on rank 0 (each step ends with a device synchronisation)."""
    def __init__(self, dev):
        import os, time
        self.on = bool(os.environ.get("NAF_SHARD_TIMING")) and dist.get_rank() == 0
        self.dev, self.time, self.t = dev, time, None
        if self.on:
            self.t = time.perf_counter()

    def __call__(self, label):
        if self.on:
            if self.dev.type == "cuda":
                torch.cuda.synchronize(self.dev)
            now = self.time.perf_counter()
            print("  shard-timing %-18s %8.3f ms" % (label, (now - self.t) * 1e3), flush=True)
            self.t = now


HEAD_WINDOW = 8192


def _all_gather_bytes(raw: bytes, device, group):
    """all_gather of one fixed-size record per rank (a single rank: its own record, no device round trip)."""
    world = dist.get_world_size(group)
    if world == 1:
        return [bytes(raw)]
    t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    _all_gather(outs, t, group)
    return [bytes(o.cpu().numpy().tobytes()) for o in outs]


def _all_gather_ints(vals, device, group):
    world = dist.get_world_size(group)
    if world == 1:
        return [[int(x) for x in vals]]
    t = torch.tensor(list(vals), dtype=torch.int64, device=device)
    outs = [torch.empty_like(t) for _ in range(world)]
    _all_gather(outs, t, group)
    return [[int(x) for x in o.cpu().tolist()] for o in outs]


def ennaf_sharded(ctx, d_buf, n, opts=None, dst=0, group=None, everywhere=False):
    """Ranks of a process group; rank r holds bytes [a_r, a_{r+1}) of the text in d_buf[:n] (consecutive slices in rank order;
    d_buf may be longer than n: spare room behind the slice takes the few bytes a shard borrows from the next slice).
    Returns (archive, report, info) on `dst` -- on every rank with everywhere=True -- and (None, report, info) elsewhere;
    info = {"cut": bytes of this slice that went to the previous shard, "halo": bytes borrowed from the following slices}."""
    opts = opts or make_opts()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = d_buf.device
    lap = _Lap(dev)
    # ONE exchange in front of the cut: the format and first record (rank 0 looks at its slice), every slice's size and its last byte
    # (an EOL in front of a slice makes its position 0 a place where a line starts).  The last byte never visits the host on its own:
    # it rides in the gathered record.
    meta = [0, 0]
    if rank == 0:
        meta = list(ctx.ennaf_sniff(d_buf[:n], opts.format))
    if world == 1:
        first = [[meta[0], meta[1], n, -1]]
    else:
        rec = torch.tensor([meta[0], meta[1], n, -1], dtype=torch.int64, device=dev)
        if n > (meta[1] if rank == 0 else 0):
            rec[3] = d_buf[n - 1]
        outs = [torch.empty_like(rec) for _ in range(world)]
        _all_gather(outs, rec, group)
        first = torch.stack(outs).cpu().tolist()
    fmt, p0 = int(first[0][0]), int(first[0][1])
    if fmt == 0:
        # nothing but white space in the first slice.  The C host hands such an input to one device (naf_amd/host/ennaf.c: sh_fallback);
        # here the slices live on different ranks, so that only works when the other slices are empty too: rank 0 makes the archive
        # of "no records" with the one-call encoder, as the C host would
        sizes = [[int(f[2])] for f in first]
        if any(sz[0] for sz in sizes[1:]):
            raise ValueError("the first rank's slice holds no record start: give rank 0 the beginning of the text")
        arc, rep = ctx.ennaf(d_buf[:n], seq_type=opts.seq_type, fmt=opts.format, no_mask=bool(opts.no_mask), level=opts.level,
                             line_length=opts.line_length, title=opts.title, strict=bool(opts.strict), long_log=opts.long_log)
        info = {"cut": 0, "halo": 0}
        if everywhere or dst != 0:                               # a few dozen bytes: everybody gets them
            ln = _all_gather_ints([arc.numel() if rank == 0 else 0], dev, group)[0][0]
            if rank != 0:
                arc = torch.empty(ln, dtype=torch.uint8, device=dev)
            _broadcast(arc, dist.get_global_rank(group, 0) if group is not None else 0, group)
        return (arc if (everywhere or rank == dst) else None), rep, info
    lo = p0 if rank == 0 else 0
    tails = [[int(f[3]) if int(f[2]) > (p0 if r == 0 else 0) else -1] for r, f in enumerate(first)]
    prev = 0x0A                                   # the byte "in front of" p0 counts as a line end
    for r in range(rank):
        if tails[r][0] >= 0:
            prev = tails[r][0]
    prev_is_eol = rank == 0 or _is_eol(prev)
    mine = d_buf[lo:n]
    lap("sniff+tails")
    # where my shard begins inside my slice
    skip = 0
    if fmt == capi.FMT_FASTQ:
        cnt = ctx.ennaf_count_lines(mine, prev_is_eol) if mine.numel() else 0
        cnts = _all_gather_ints([cnt], dev, group)
        skip = (-sum(c[0] for c in cnts[:rank])) % 4
    if rank == 0:
        cut = 0
    else:
        cut = ctx.ennaf_find_cut(mine, fmt, prev_is_eol, skip) if mine.numel() else 0
    lap("lines+cut")
    # heads: the bytes in front of each rank's cut belong to the shard before it; a slice without a cut is all head
    # one exchange: (cut, slice size) and the first HEAD_WINDOW bytes of the slice (a cut lies a line or a record into the slice: nearly
    # always inside the window); only a longer head costs a second exchange
    heads = None
    if world == 1:
        lens = [[cut, int(mine.numel())]]
    else:
        rec = torch.zeros(16 + HEAD_WINDOW, dtype=torch.uint8, device=dev)
        rec[:16] = torch.tensor([cut, int(mine.numel())], dtype=torch.int64).view(torch.uint8).to(dev)
        k = min(cut, HEAD_WINDOW)
        if k:
            rec[16:16 + k] = mine[:k]
        outs = [torch.empty_like(rec) for _ in range(world)]
        _all_gather(outs, rec, group)
        lens = [[int(x) for x in row] for row in torch.stack([o[:16] for o in outs]).cpu().view(torch.int64).tolist()]
        maxc = max(c for c, _ in lens)
        if maxc <= HEAD_WINDOW:
            heads = [o[16:] for o in outs]
        else:
            h = torch.zeros(maxc, dtype=torch.uint8, device=dev)
            h[:cut] = mine[:cut]
            heads = [torch.empty_like(h) for _ in range(world)]
            _all_gather(heads, h, group)
    borrow = []
    for r in range(rank + 1, world):
        c, ln = lens[r]
        if c:
            borrow.append(heads[r][:c])
        if c < ln:
            break
    halo = sum(int(b.numel()) for b in borrow)
    own = mine.numel() - cut
    if own == 0 and lens[rank][1] > 0 and rank > 0:
        borrow, halo = [], 0                                      # my whole slice went to an earlier shard: nothing starts here
    if halo and d_buf.numel() >= n + halo:
        pos = n
        for b in borrow:
            d_buf[pos:pos + b.numel()] = b
            pos += b.numel()
        text = d_buf[lo + cut:n + halo]
    elif halo:
        text = torch.cat([mine[cut:]] + borrow)
    else:
        text = mine[cut:]
    lap("heads+halo")
    info = ctx.ennaf_shard_begin(text, opts, fmt, rank, world)
    lap("shard_begin")
    infos = [capi.ShardInfo.from_buffer_copy(b) for b in _all_gather_bytes(bytes(info), dev, group)]
    lap("gather infos")
    buf, pc = ctx.ennaf_shard_finish(opts, infos, text.numel())
    lap("shard_finish")
    pieces = [capi.ShardPieces.from_buffer_copy(b) for b in _all_gather_bytes(bytes(pc), dev, group)]
    segs, lit, naf_len, rep = capi.stitch_plan(opts, infos, pieces)
    extra = {"cut": cut, "halo": halo, "format": fmt, "text_len": int(text.numel())}
    out = None
    ops = []
    if rank == dst:
        out = torch.empty(max(naf_len, 1), dtype=torch.uint8, device=dev)
        lit_t = torch.frombuffer(bytearray(lit), dtype=torch.uint8).to(dev)
        for g in segs:
            if g.len == 0:
                continue
            if g.shard < 0:
                out[g.dst_off:g.dst_off + g.len] = lit_t[g.src_off:g.src_off + g.len]
            elif g.shard == rank:
                out[g.dst_off:g.dst_off + g.len] = buf[g.src_off:g.src_off + g.len]
            else:
                ops.append(dist.P2POp(dist.irecv, out[g.dst_off:g.dst_off + g.len], g.shard, group))
    else:
        for g in segs:
            if g.shard == rank and g.len:
                ops.append(dist.P2POp(dist.isend, buf[g.src_off:g.src_off + g.len], dst, group))
    _p2p(ops)
    lap("stitch+gather")
    if everywhere:
        if rank != dst:
            out = torch.empty(max(naf_len, 1), dtype=torch.uint8, device=dev)
        _broadcast(out, dist.get_global_rank(group, dst) if group is not None else dst, group)
    return (out[:naf_len] if out is not None else None), rep, extra
