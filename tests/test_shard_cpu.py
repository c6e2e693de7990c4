"""CPU tests of the multi-GPU encode protocol (include/naf_gpu.h, "ennaf of ONE input on several GPUs"): the library's host-only
carry and stitch-plan functions and naf_amd/shard.py are driven with a stand-in for the per-shard device calls
(tests/shard_standin.py, built on the oracle); the joined archive must hold exactly the six streams the oracle makes of the whole
text.  The device calls themselves are checked by tests/test_gpu_shard.py on the GPU."""
import os
import sys

import numpy as np
import pytest

SEED = int(os.environ.get("NAF_TEST_SEED", "0"))          # other texts of the same kinds (tests/test_gpu_encode.py)

from conftest import ROOT

from naf_amd import capi, shard
from shard_standin import StandInCtx


def join_local(O, text, n_shards, opts=None):
    import torch
    opts = opts or shard.make_opts()
    ctxs = [StandInCtx(O) for _ in range(n_shards)]
    t = torch.frombuffer(bytearray(text), dtype=torch.uint8)
    naf, rep = shard.ennaf_sharded_local(ctxs, t, opts)
    return naf.numpy().tobytes(), rep, ctxs


def check_against_whole(O, text, naf, rep, seq_type=0, no_mask=False):
    sp = O.split_text(text, seq_type, no_mask)
    h = O.parse_naf(naf)
    want = [sp.ids, sp.comments, sp.lengths, sp.mask, sp.seq, sp.qual]
    store_mask = not (no_mask or seq_type >= 2)
    for i, name in enumerate(("ids", "comments", "lengths", "mask", "seq", "qual")):
        if (i == 3 and not store_mask) or (i == 5 and sp.format != O.FMT_FASTQ):
            assert h.payload_off[i] is None, name
            continue
        assert O.zstd_decompress(h.frame(naf, i), len(want[i]) + 16) == want[i], name
    assert h.n_sequences == sp.n_sequences and h.orig[O.SEQ] == sp.n_bases and h.line_length == sp.longest_line
    assert rep.n_sequences == sp.n_sequences and rep.n_bases == sp.n_bases and rep.longest_line == sp.longest_line
    assert list(rep.unexpected_seq) == sp.unexpected["seq"] and list(rep.unexpected_id) == sp.unexpected["id"]
    ref = O.ennaf(text, seq_type, no_mask)
    assert naf[: h.header_bytes] == ref[: O.parse_naf(ref).header_bytes]
    if sp.n_sequences:
        assert O.unnaf(naf, -1) == O.unnaf(ref, -1)


def fasta_fuzz(rng, n_records, max_len, width=None):
    bases = np.frombuffer(b"ACGTACGTNacgtnRYKM-", dtype=np.uint8)
    out = bytearray()
    for r in range(n_records):
        out += b">r%d some comment %d\n" % (r, r * 7) if rng.random() < 0.7 else b">r%d\n" % r
        total = int(rng.integers(0, max_len))
        # runs of one case so that mask runs cross the cuts
        seq = bytearray()
        while len(seq) < total:
            run = bases[rng.integers(0, len(bases), int(rng.integers(1, 400)))].tobytes()
            seq += run.lower() if rng.random() < 0.5 else run.upper()
        seq = bytes(seq[:total])
        w = width or int(rng.integers(1, 90))
        for a in range(0, total, w):
            out += seq[a:a + w] + (b"\r\n" if rng.random() < 0.05 else b"\n")
    return bytes(out)


def test_shard_carry_join_fasta_fuzz(oracle):
    rng = np.random.default_rng(41 + SEED)
    for i in range(60):
        text = fasta_fuzz(rng, int(rng.integers(1, 8)), int(rng.integers(1, 3000)))
        for n in (2, 3, 8):
            naf, rep, _ = join_local(oracle, text, n)
            check_against_whole(oracle, text, naf, rep)


def test_shard_one_record_mask_run_across_three_shards(oracle):
    # one record: every cut falls inside it; a lower-case run covers the second shard entirely; odd base counts at the cuts
    seq = b"ACG" + b"acgtn" * 161 + b"TTGCA" * 50 + b"a"
    text = b">chr1 one record\n" + b"".join(seq[a:a + 7] + b"\n" for a in range(0, len(seq), 7))
    for n in (2, 3, 5, 8):
        naf, rep, ctxs = join_local(oracle, text, n)
        check_against_whole(oracle, text, naf, rep)
    # all lower case, all upper case, alternating by base
    for body in (b"acgt" * 300, b"ACGT" * 300, b"aCgT" * 300, b"a", b"A"):
        text = b">x\n" + b"".join(body[a:a + 11] + b"\n" for a in range(0, len(body), 11))
        for n in (2, 3, 8):
            naf, rep, _ = join_local(oracle, text, n)
            check_against_whole(oracle, text, naf, rep)


def test_shard_more_shards_than_lines(oracle):
    for text in (b">a\nACGT\n", b">a\n", b">a", b">a b\nAC\n>c\n\n>d\nacgtn", b">x\n" + b"A" * 300):
        for n in (2, 3, 8):
            naf, rep, _ = join_local(oracle, text, n)
            check_against_whole(oracle, text, naf, rep)


def test_shard_join_protein_text_nomask(oracle):
    rng = np.random.default_rng(3 + SEED)
    text = fasta_fuzz(rng, 5, 900)
    for st, nm in ((oracle.PROTEIN, False), (oracle.TEXT, False), (oracle.DNA, True), (oracle.RNA, False)):
        for n in (2, 3):
            naf, rep, _ = join_local(oracle, text, n, shard.make_opts(seq_type=st, no_mask=nm))
            check_against_whole(oracle, text, naf, rep, st, nm)


def test_shard_join_fastq(oracle):
    from naf_amd import synth
    rng = np.random.default_rng(11 + SEED)
    for i in range(12):
        text = synth.fastq_reads(int(rng.integers(1, 80)), int(rng.integers(1, 120)), seed=200 + i, var_len=bool(i % 2))
        for n in (2, 3, 8):
            naf, rep, _ = join_local(oracle, text, n)
            check_against_whole(oracle, text, naf, rep)
    weird = b"\n\n@r1 c\nAC GT\n+\n!!!!\n\n@r2\tcomment\nACNNxz\n\n+r2 again\n\nIIIIII\n@r3\nA\n+\n~"
    for n in (2, 3):
        naf, rep, _ = join_local(oracle, weird, n)
        check_against_whole(oracle, weird, naf, rep)


def test_shard_carry_fields(oracle):
    """The carry of the middle shard of three, spelled out (include/naf_gpu.h: naf_gpu_shard_carry)."""
    def info(k, n_seq, T, lead, first, last, changes=0, fchg=2 ** 64 - 1):
        x = capi.ShardInfo()
        x.shard, x.n_shards, x.format, x.seq_type = k, 3, capi.FMT_FASTA, 0
        x.n_sequences, x.n_bases, x.lead_bases = n_seq, T, lead
        x.first_base, x.last_base, x.mask_changes, x.mask_first_change, x.store_mask = first, last, changes, fchg, 1
        x.n_ids = n_seq
        return x
    infos = [info(0, 2, 7, 0, ord("A"), ord("a")), info(1, 0, 10, 10, ord("c"), ord("c")), info(2, 1, 9, 4, ord("g"), ord("T"), 1, 6)]
    k1 = capi.shard_carry(infos, 1)
    assert (k1.first_record, k1.skip_first, k1.tail_hi, k1.prev_masked, k1.skip_run0, k1.run_ext, k1.tail_extra) == (2, 1, 2, 1, 1, 6, 0)
    k0 = capi.shard_carry(infos, 0)
    assert (k0.skip_first, k0.tail_hi, k0.tail_extra, k0.run_ext, k0.skip_run0) == (0, 4, 14, 16, 0)       # 7 bases: 'c' completes the byte; record +10 +4
    k2 = capi.shard_carry(infos, 2)
    assert (k2.first_record, k2.skip_first, k2.tail_hi, k2.skip_run0) == (2, 1, 0, 1)
    assert list(k0.first) == [1] * 6 and list(k2.first) == [0] * 6
    assert list(k2.last) == [1, 1, 1, 1, 1, 1]                                   # quality: nobody has any, the last shard closes the frame
    assert list(k0.last) == [0] * 6 and list(k1.last) == [0] * 6


# ---- the same protocol across two ranks (gloo) -------------------------------------------------------------------------------------
def _worker(rank, world, port, q, kind):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    O.lib()
    from naf_amd import shard as sh, synth
    from shard_standin import StandInCtx
    import test_shard_cpu as me
    rng = np.random.default_rng(99 + SEED)
    if kind == "fasta":
        text = b"\n \n" + me.fasta_fuzz(rng, 3, 5000, width=60)
    else:
        text = synth.fastq_reads(57, 90, seed=5, var_len=True)
        sh.P2P_MAX_BYTES = 700                                      # (every part of the archive travels as several pieces, cut the same way on both ranks: shard._pieces)
    a = [0, len(text) * 2 // 5, len(text)]                          # uneven nominal slices, cut anywhere
    mine = text[a[rank]:a[rank + 1]]
    buf = torch.zeros(len(mine) + 4096, dtype=torch.uint8)
    buf[: len(mine)] = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
    naf, rep, extra = sh.ennaf_sharded(StandInCtx(O), buf, len(mine), dst=0)
    ok = True
    if rank == 0:
        try:
            me.check_against_whole(O, text, naf.numpy().tobytes(), rep)
        except AssertionError as e:
            ok = repr(e)
    q.put((rank, ok, extra))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["fasta", "fastq"])
def test_sharded_ennaf_two_ranks_gloo(kind):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + (7 if kind == "fastq" else 0)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, kind)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(60)
    assert all(r[1] is True for r in res), res
    assert res[1][2]["cut"] > 0 and res[0][2]["halo"] == res[1][2]["cut"]      # rank 0 borrowed exactly what rank 1 gave up
