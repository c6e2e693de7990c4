"""TEST INFRASTRUCTURE: a CPU stand-in for the per-shard calls of include/naf_gpu.h ("ennaf of ONE input on several GPUs"),
built on the oracle's split of a slice.  It lets the CPU suite drive the real protocol code -- naf_amd/shard.py, and the
library's host-only naf_gpu_ennaf_shard_carry / naf_gpu_ennaf_stitch_plan -- without a GPU: the stand-in produces the shard
records from the oracle's view of a slice and applies the carries the LIBRARY computes; the joined archive must then equal what
the oracle makes of the whole text.  Frames are stored as Raw blocks (any conformant decoder reads them)."""
import numpy as np
import torch

from naf_amd import capi

EOL = (0x0A, 0x0B, 0x0C, 0x0D)
SPACE = (0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x20)
NUC = b"-TGKCYSBAWRDMHVN"


def _b(t):
    return t.cpu().numpy().tobytes() if isinstance(t, torch.Tensor) else bytes(t)


def line_starts(text: bytes, prev_is_eol: bool):
    prev = prev_is_eol
    out = []
    for i, c in enumerate(text):
        e = c in EOL
        if not e and prev:
            out.append(i)
        prev = e
    return out


def raw_part(data: bytes, first: bool, last: bool) -> bytes:
    out = bytearray(b"\x00\x48" if first else b"")
    blocks = [data[i:i + 100000] for i in range(0, len(data), 100000)]
    if not blocks and last:
        blocks = [b""]
    for i, blk in enumerate(blocks):
        hdr = (len(blk) << 3) | (1 if (last and i == len(blocks) - 1) else 0)
        out += hdr.to_bytes(3, "little") + blk
    return bytes(out)


def mask_units(case: np.ndarray, prev_masked: bool, skip_run0: bool, run_ext: int) -> bytes:
    """encoders.c:98-146 restated for a shard: runs that START in this shard, the last one `run_ext` bases longer."""
    T = len(case)
    if T == 0:
        return b""
    prev = np.concatenate([[prev_masked], case[:-1]])
    bnd = np.nonzero(case != prev)[0].tolist()
    starts = ([] if skip_run0 else [0]) + bnd
    ends = bnd + [T + run_ext]
    if skip_run0:
        ends = ends[1:] if bnd else []
    out = bytearray()
    for s, e in zip(starts, ends):
        ln = e - s
        out += b"\xff" * (ln // 255) + bytes([ln % 255])
    return bytes(out)


class StandInCtx:
    def __init__(self, oracle):
        self.O = oracle
        self.device = torch.device("cpu")
        self.st = None

    # ---- cuts
    def ennaf_sniff(self, t, fmt=0):
        text = _b(t)
        p0 = next((i for i, c in enumerate(text) if c not in SPACE), len(text))
        if p0 == len(text):
            return 0, p0
        return (capi.FMT_FASTA if text[p0] == ord(">") else capi.FMT_FASTQ), p0

    def ennaf_count_lines(self, t, prev_is_eol):
        return len(line_starts(_b(t), bool(prev_is_eol)))

    def ennaf_find_cut(self, t, fmt, prev_is_eol, skip_lines=0):
        text = _b(t)
        if fmt == capi.FMT_FASTA:
            if prev_is_eol:
                return 0
            return next((i + 1 for i in range(len(text) - 1) if text[i] in EOL), len(text))
        ls = line_starts(text, bool(prev_is_eol))
        return ls[skip_lines] if skip_lines < len(ls) else len(text)

    # ---- shard
    def ennaf_shard_begin(self, t, opts, fmt, shard, n_shards):
        O = self.O
        text = _b(t)
        info = capi.ShardInfo()
        info.shard, info.n_shards, info.format, info.seq_type, info.text_len = shard, n_shards, fmt, opts.seq_type, len(text)
        fourbit = opts.seq_type <= 1
        store_mask = fourbit and not opts.no_mask
        info.store_mask, info.store_quality = int(store_mask), int(fmt == capi.FMT_FASTQ)
        dummy = fmt == capi.FMT_FASTA and not text.startswith(b">")
        src = (b">\n" + text) if dummy else text
        st = {"fourbit": fourbit, "store_mask": store_mask, "ids": b"", "cmt": b"", "qual": b"", "lens": [], "codes": np.zeros(0, np.uint8),
              "case": np.zeros(0, bool), "bytes": b""}
        self.st = st
        self.rank = shard
        if not src.strip(bytes(SPACE)):
            info.lead_bases = 0
            return info
        sp = O.split_text(src, opts.seq_type, bool(opts.no_mask), forced_format=fmt)
        units = np.frombuffer(sp.lengths, dtype="<u4").astype(np.uint64).tolist()
        lens, acc = [], 0
        for u in units:
            acc += u
            if u != 0xFFFFFFFF:
                lens.append(acc)
                acc = 0
        ids, cmt = sp.ids, sp.comments
        T = sp.n_bases
        lead = 0
        if dummy:
            lead = lens.pop(0)
            ids, cmt = ids[1:], cmt[1:]
        if not lens:
            lead = T
        if fourbit:
            pk = np.frombuffer(sp.seq, dtype=np.uint8)
            codes = np.empty(2 * len(pk), np.uint8)
            codes[0::2] = pk & 15
            codes[1::2] = pk >> 4
            st["codes"] = codes[:T]
        else:
            st["bytes"] = sp.seq
        case = np.zeros(T, bool)
        if store_mask and T:
            pos, on, acc = 0, False, 0
            for u in sp.mask:
                acc += u
                if u != 255:
                    case[pos:pos + acc] = on
                    pos += acc
                    acc = 0
                    on = not on
        st.update(ids=ids, cmt=cmt, qual=sp.qual, lens=lens, case=case)
        info.n_sequences, info.n_bases, info.longest_line, info.lead_bases = len(lens), T, sp.longest_line, lead
        info.n_ids, info.n_comments, info.n_quality = len(ids), len(cmt), len(sp.qual)
        if T:
            ch = np.nonzero(case[1:] != case[:-1])[0] + 1
            info.mask_changes = len(ch)
            info.mask_first_change = int(ch[0]) if len(ch) else 2 ** 64 - 1
            info.mask_last_change = int(ch[-1]) if len(ch) else 0
            if fourbit:
                letter = lambda code, low: NUC[code] | (0x20 if low and NUC[code] != ord("-") else 0)
                info.first_base = letter(st["codes"][0], case[0])
                info.last_base = letter(st["codes"][-1], case[-1])
            else:
                info.first_base, info.last_base = st["bytes"][0], st["bytes"][-1]
        for k, key in enumerate(("id", "comment", "seq", "qual")):
            for i in range(257):
                info.unexpected[k][i] = sp.unexpected[key][i]
        return info

    def ennaf_shard_finish(self, opts, infos, text_len):
        st = self.st
        k = self.rank
        K = capi.shard_carry(infos, k)
        lens = list(st["lens"])
        if lens:
            lens[-1] += K.tail_extra
        lu = bytearray()
        for ln in lens:
            while ln >= 0xFFFFFFFF:
                lu += b"\xff\xff\xff\xff"
                ln -= 0xFFFFFFFF
            lu += int(ln).to_bytes(4, "little")
        if st["fourbit"]:
            codes = st["codes"][K.skip_first:].astype(np.uint8)
            if len(codes) & 1:
                codes = np.concatenate([codes, [K.tail_hi]]).astype(np.uint8)
            seq = (codes[0::2] | (codes[1::2] << 4)).astype(np.uint8).tobytes()
        else:
            seq = st["bytes"]
        mask = mask_units(st["case"], bool(K.prev_masked), bool(K.skip_run0), K.run_ext) if st["store_mask"] else b""
        streams = [st["ids"], st["cmt"], bytes(lu), mask, seq, st["qual"]]
        present = [True, True, True, st["store_mask"], True, infos[0].store_quality != 0]
        raws = [len(st["ids"]), len(st["cmt"]), len(lu), len(mask), len(st["codes"]) if st["fourbit"] else len(seq), len(st["qual"])]
        pc = capi.ShardPieces()
        buf = bytearray()
        for s in range(6):
            if not present[s]:
                continue
            part = raw_part(streams[s], bool(K.first[s]), bool(K.last[s]))
            pc.off[s], pc.len[s], pc.raw[s] = len(buf), len(part), raws[s]
            buf += part
        pc.total = len(buf)
        self.raw_streams = streams
        return torch.frombuffer(bytearray(buf) + bytearray(1), dtype=torch.uint8), pc

    def ennaf_stitch(self, segs, lit, bufs, out):
        for g in segs:
            src = lit if g.shard < 0 else _b(bufs[g.shard])
            out[g.dst_off:g.dst_off + g.len] = torch.frombuffer(bytearray(src[g.src_off:g.src_off + g.len]), dtype=torch.uint8)
