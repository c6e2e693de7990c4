"""ctypes binding of oracle/libnaf_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
package (naf_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libnaf_oracle.so")
REF_ENNAF = os.path.join(_HERE, "_ref", "ennaf")
REF_UNNAF = os.path.join(_HERE, "_ref", "unnaf")
REF_TIMEOUT = 120      # the reference can spin forever on malformed archives

DNA, RNA, PROTEIN, TEXT = 0, 1, 2, 3
FMT_UNKNOWN, FMT_FASTA, FMT_FASTQ = 0, 1, 2
MODE_FASTA, MODE_FASTQ, MODE_SEQ, MODE_SEQUENCES, MODE_4BIT = 0, 1, 2, 3, 4
IDS, COMMENTS, LENGTHS, MASK, SEQ, QUAL = range(6)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


class _Buf(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("len", C.c_size_t), ("cap", C.c_size_t)]

    def bytes(self):
        return C.string_at(self.data, self.len) if self.len else b""


class _Split(C.Structure):
    _fields_ = [("format", C.c_int), ("n_sequences", C.c_uint64), ("longest_line", C.c_uint64),
                ("n_bases", C.c_uint64),
                ("ids", _Buf), ("comments", _Buf), ("lengths", _Buf), ("mask", _Buf), ("seq", _Buf), ("qual", _Buf),
                ("unexpected_id", C.c_uint64 * 257), ("unexpected_comment", C.c_uint64 * 257),
                ("unexpected_seq", C.c_uint64 * 257), ("unexpected_qual", C.c_uint64 * 257),
                ("error", C.c_char * 256)]


class _Naf(C.Structure):
    _fields_ = [("version", C.c_int), ("seq_type", C.c_int), ("flags", C.c_int), ("separator", C.c_uint8),
                ("line_length", C.c_uint64), ("n_sequences", C.c_uint64),
                ("title", C.c_void_p), ("title_len", C.c_uint64),
                ("orig", C.c_uint64 * 6), ("comp", C.c_uint64 * 6), ("payload", C.c_void_p * 6),
                ("header_bytes", C.c_size_t), ("error", C.c_char * 128)]


class FrameInfo(C.Structure):
    _fields_ = [("n_blocks", C.c_uint32), ("n_raw", C.c_uint32), ("n_rle", C.c_uint32), ("n_compressed", C.c_uint32),
                ("lit_raw", C.c_uint32), ("lit_rle", C.c_uint32), ("lit_huf", C.c_uint32), ("lit_treeless", C.c_uint32),
                ("seq_blocks", C.c_uint32), ("n_sequences", C.c_uint64),
                ("mode_count", (C.c_uint32 * 4) * 3),
                ("window_log", C.c_uint32), ("single_segment", C.c_uint32), ("has_checksum", C.c_uint32),
                ("has_fcs", C.c_uint32), ("max_offset", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.nafo_zstd_decompress.restype = C.c_longlong
        L.nafo_zstd_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.nafo_zstd_decompressed_size.restype = C.c_longlong
        L.nafo_zstd_decompressed_size.argtypes = [C.c_char_p, C.c_size_t]
        L.nafo_zstd_store_raw.restype = C.c_longlong
        L.nafo_zstd_store_raw.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.nafo_zstd_frame_info_get.restype = C.c_longlong
        L.nafo_zstd_frame_info_get.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(FrameInfo)]
        L.nafo_split_text.restype = C.c_int
        L.nafo_split_text.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_Split)]
        L.nafo_split_free.argtypes = [C.POINTER(_Split)]
        L.nafo_write_naf.restype = C.c_longlong
        L.nafo_write_naf.argtypes = [C.POINTER(_Split), C.c_int, C.c_int, C.c_longlong, C.c_char_p, C.c_void_p, C.c_size_t]
        L.nafo_vle_write.restype = C.c_size_t
        L.nafo_vle_write.argtypes = [C.c_uint64, C.c_void_p]
        L.nafo_vle_read.restype = C.c_int
        L.nafo_vle_read.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)]
        L.nafo_parse_naf.restype = C.c_int
        L.nafo_parse_naf.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_Naf)]
        L.nafo_unnaf.restype = C.c_longlong
        L.nafo_unnaf.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_size_t, C.c_char_p]
        L.nafo_unnaf_size.restype = C.c_longlong
        L.nafo_unnaf_size.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_longlong]
        L.nafo_pack_4bit.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        L.nafo_unpack_4bit.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p]
        L.nafo_mask_rle.restype = C.c_size_t
        L.nafo_mask_rle.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.nafo_mask_apply.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
        _lib = L
    return _lib


# ---- zstd ----------------------------------------------------------------------------------------
def zstd_decompress(frame: bytes, size_hint: int = None) -> bytes:
    L = lib()
    if size_hint is None:
        size_hint = L.nafo_zstd_decompressed_size(frame, len(frame))
        if size_hint < 0:
            raise ValueError("oracle zstd: error %d" % size_hint)
    out = C.create_string_buffer(max(size_hint, 1))
    n = L.nafo_zstd_decompress(frame, len(frame), out, size_hint)
    if n < 0:
        raise ValueError("oracle zstd: error %d" % n)
    return out.raw[:n]


def zstd_store_raw(data: bytes) -> bytes:
    cap = len(data) + 32 + 3 * (len(data) // 131072 + 1)
    out = C.create_string_buffer(cap)
    n = lib().nafo_zstd_store_raw(data, len(data), out, cap)
    assert n > 0
    return out.raw[:n]


def zstd_frame_info(frame: bytes) -> FrameInfo:
    fi = FrameInfo()
    n = lib().nafo_zstd_frame_info_get(frame, len(frame), C.byref(fi))
    if n < 0:
        raise ValueError("oracle zstd: error %d" % n)
    return fi


# ---- ennaf side ------------------------------------------------------------------------------------
class Split:
    """The six uncompressed streams + header fields the reference's ennaf would produce."""

    def __init__(self, s: _Split):
        self.format = s.format
        self.n_sequences = s.n_sequences
        self.longest_line = s.longest_line
        self.n_bases = s.n_bases
        self.ids, self.comments, self.lengths = s.ids.bytes(), s.comments.bytes(), s.lengths.bytes()
        self.mask, self.seq, self.qual = s.mask.bytes(), s.seq.bytes(), s.qual.bytes()
        self.unexpected = {k: list(getattr(s, "unexpected_" + k)) for k in ("id", "comment", "seq", "qual")}


def split_text(text: bytes, seq_type=DNA, no_mask=False, well_formed=False, forced_format=FMT_UNKNOWN) -> Split:
    s = _Split()
    rc = lib().nafo_split_text(text, len(text), seq_type, int(no_mask), int(well_formed), forced_format, C.byref(s))
    if rc < 0:
        msg = s.error.decode("latin1")
        lib().nafo_split_free(C.byref(s))
        raise ValueError(msg)
    out = Split(s)
    lib().nafo_split_free(C.byref(s))
    return out


def ennaf(text: bytes, seq_type=DNA, no_mask=False, line_length=-1, title=None, well_formed=False) -> bytes:
    """Oracle .naf (streams stored as Raw-block zstd frames -- compressed bytes are never a parity target)."""
    s = _Split()
    rc = lib().nafo_split_text(text, len(text), seq_type, int(no_mask), int(well_formed), FMT_UNKNOWN, C.byref(s))
    if rc < 0:
        msg = s.error.decode("latin1")
        lib().nafo_split_free(C.byref(s))
        raise ValueError(msg)
    cap = len(text) + (1 << 16) + 64 * int(s.n_sequences)
    out = C.create_string_buffer(cap)
    n = lib().nafo_write_naf(C.byref(s), seq_type, int(no_mask), line_length, title, out, cap)
    lib().nafo_split_free(C.byref(s))
    assert n > 0
    return out.raw[:n]


# ---- unnaf side ------------------------------------------------------------------------------------
class NafHeader:
    def __init__(self, h: _Naf, naf: bytes):
        self.version, self.seq_type, self.flags, self.separator = h.version, h.seq_type, h.flags, h.separator
        self.line_length, self.n_sequences = h.line_length, h.n_sequences
        self.orig, self.comp = list(h.orig), list(h.comp)
        base = C.cast(C.c_char_p(naf), C.c_void_p).value
        self.payload_off = [(h.payload[i] - base) if h.payload[i] else None for i in range(6)]
        self.header_bytes = h.header_bytes

    def frame(self, naf: bytes, i: int) -> bytes:
        """Section i as a complete zstd frame (magic re-prefixed, unnaf utils.c:144-150)."""
        o = self.payload_off[i]
        return b"\x28\xb5\x2f\xfd" + naf[o:o + self.comp[i]]


def parse_naf(naf: bytes) -> NafHeader:
    h = _Naf()
    if lib().nafo_parse_naf(naf, len(naf), C.byref(h)) < 0:
        raise ValueError(h.error.decode("latin1"))
    return NafHeader(h, naf)


def unnaf(naf: bytes, mode=-1, use_mask=True, line_length=-1) -> bytes:
    L = lib()
    err = C.create_string_buffer(128)
    h = parse_naf(naf)
    cap = int(h.orig[SEQ] * 2 + h.orig[IDS] + h.orig[COMMENTS] + h.orig[QUAL] + 8 * h.n_sequences + 64)
    if line_length > 0:
        cap += int(h.orig[SEQ] // line_length) + 1
    out = C.create_string_buffer(max(cap, 1))
    n = L.nafo_unnaf(naf, len(naf), mode, int(use_mask), line_length, out, cap, err)
    if n < 0:
        raise ValueError(err.value.decode("latin1") or "oracle unnaf error %d" % n)
    return out.raw[:n]


def pack_4bit(bases: bytes) -> bytes:
    out = C.create_string_buffer(max((len(bases) + 1) // 2, 1))
    lib().nafo_pack_4bit(bases, len(bases), out)
    return out.raw[:(len(bases) + 1) // 2]


def unpack_4bit(packed: bytes, n_bases: int, rna=False) -> bytes:
    out = C.create_string_buffer(max(n_bases, 1))
    lib().nafo_unpack_4bit(packed, n_bases, int(rna), out)
    return out.raw[:n_bases]


def mask_rle(bases: bytes) -> bytes:
    cap = len(bases) + 2
    out = C.create_string_buffer(cap)
    n = lib().nafo_mask_rle(bases, len(bases), out, cap)
    return out.raw[:n]


def mask_apply(bases: bytes, units: bytes) -> bytes:
    buf = C.create_string_buffer(bases, max(len(bases), 1))
    lib().nafo_mask_apply(buf, len(bases), units, len(units))
    return buf.raw[:len(bases)]


# ---- the real reference (oracle/_ref) --------------------------------------------------------------
def have_ref() -> bool:
    return os.access(REF_ENNAF, os.X_OK) and os.access(REF_UNNAF, os.X_OK)


def ref_ennaf(text: bytes, args=(), tmpdir="/tmp") -> bytes:
    env = dict(os.environ, TMPDIR=tmpdir)
    p = subprocess.run([REF_ENNAF, *args, "-c"], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=REF_TIMEOUT)
    if p.returncode != 0:
        raise ValueError(p.stderr.decode("latin1"))
    return p.stdout


def ref_ennaf_full(text: bytes, args=(), tmpdir="/tmp"):
    env = dict(os.environ, TMPDIR=tmpdir)
    p = subprocess.run([REF_ENNAF, *args, "-c"], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=REF_TIMEOUT)
    return p.returncode, p.stdout, p.stderr


def ref_unnaf(naf: bytes, args=()) -> bytes:
    p = subprocess.run([REF_UNNAF, *args, "-c"], input=naf, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=REF_TIMEOUT)
    if p.returncode != 0:
        raise ValueError(p.stderr.decode("latin1"))
    return p.stdout
