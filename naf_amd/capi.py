"""ctypes binding of the libnaf_gpu.so C-ABI (include/naf_gpu.h).

Used by the tests and bench.py; the shipped hosts are the C programs in naf_amd/host/.  Device
buffers are torch uint8 CUDA(HIP) tensors -- torch is plumbing for HBM allocations and streams only.
There is no CPU fallback: loading fails loudly if the shared library is missing, and `Context()`
raises if no gfx950 device is usable.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NAF_GPU_LIB") or os.path.join(_HERE, "libnaf_gpu.so")       # NAF_GPU_LIB: another build of the same C-ABI (A / B measurements)

OUT_DEFAULT, OUT_FASTA, OUT_FASTQ, OUT_SEQ, OUT_SEQUENCES, OUT_4BIT = -1, 0, 1, 2, 3, 4
SEQ_DNA, SEQ_RNA, SEQ_PROTEIN, SEQ_TEXT = 0, 1, 2, 3
FMT_AUTO, FMT_FASTA, FMT_FASTQ = 0, 1, 2
E_CAP = -6

EXPORTS = [
    "naf_gpu_init", "naf_gpu_shutdown", "naf_gpu_strerror", "naf_gpu_last_error", "naf_gpu_set_stream",
    "naf_gpu_synchronize", "naf_gpu_reserve", "naf_gpu_malloc", "naf_gpu_free", "naf_gpu_host_alloc",
    "naf_gpu_host_free", "naf_gpu_upload", "naf_gpu_download", "naf_gpu_download_async", "naf_gpu_histogram", "naf_gpu_zstd_decompress",
    "naf_gpu_mem_info", "naf_gpu_release_scratch", "naf_gpu_zstd_compress", "naf_gpu_zstd_compress_bound", "naf_gpu_parse_header", "naf_gpu_parse_header_host",
    "naf_gpu_unnaf_size", "naf_gpu_unnaf", "naf_gpu_unnaf_range", "naf_gpu_ennaf_bound", "naf_gpu_ennaf",
    "naf_gpu_set_timing", "naf_gpu_get_timing",
    "naf_gpu_ennaf_sniff", "naf_gpu_ennaf_count_lines", "naf_gpu_ennaf_find_cut", "naf_gpu_ennaf_shard_begin", "naf_gpu_ennaf_shard_bound",
    "naf_gpu_ennaf_shard_finish", "naf_gpu_ennaf_shard_carry", "naf_gpu_ennaf_stitch_plan", "naf_gpu_ennaf_stitch",
    "naf_gpu_read_file", "naf_gpu_write_file", "naf_gpu_copy", "naf_gpu_gather_ranges", "naf_gpu_get_timing_streams",
    "naf_gpu_set_option", "naf_gpu_get_trace", "naf_gpu_clear_trace", "naf_gpu_write_fd",
]
MAX_SHARDS = 64


class UnnafOpts(C.Structure):
    _fields_ = [("out_type", C.c_int), ("use_mask", C.c_int), ("line_length", C.c_int64)]


class Header(C.Structure):
    _fields_ = [("version", C.c_int), ("seq_type", C.c_int), ("flags", C.c_int), ("separator", C.c_uint8),
                ("line_length", C.c_uint64), ("n_sequences", C.c_uint64), ("title_off", C.c_uint64), ("title_len", C.c_uint64),
                ("orig_size", C.c_uint64 * 6), ("comp_size", C.c_uint64 * 6), ("payload_off", C.c_uint64 * 6)]


class EnnafOpts(C.Structure):
    _fields_ = [("format", C.c_int), ("seq_type", C.c_int), ("no_mask", C.c_int), ("strict", C.c_int), ("level", C.c_int),
                ("line_length", C.c_int64), ("title", C.c_char_p), ("long_log", C.c_int)]


class EnnafReport(C.Structure):
    _fields_ = [("format", C.c_int), ("n_sequences", C.c_uint64), ("n_bases", C.c_uint64), ("longest_line", C.c_uint64),
                ("unexpected_id", C.c_uint64 * 257), ("unexpected_comment", C.c_uint64 * 257),
                ("unexpected_seq", C.c_uint64 * 257), ("unexpected_qual", C.c_uint64 * 257),
                ("section_orig", C.c_uint64 * 6), ("section_comp", C.c_uint64 * 6)]


class ShardInfo(C.Structure):
    """naf_gpu_shard_info: what one shard of a multi-GPU ennaf tells the others (exchanged verbatim as bytes)."""
    _fields_ = [("shard", C.c_uint32), ("n_shards", C.c_uint32), ("format", C.c_int32), ("seq_type", C.c_int32), ("text_len", C.c_uint64),
                ("n_sequences", C.c_uint64), ("n_bases", C.c_uint64), ("longest_line", C.c_uint64), ("lead_bases", C.c_uint64),
                ("n_ids", C.c_uint64), ("n_comments", C.c_uint64), ("n_quality", C.c_uint64),
                ("mask_changes", C.c_uint64), ("mask_first_change", C.c_uint64), ("mask_last_change", C.c_uint64),
                ("first_base", C.c_uint8), ("last_base", C.c_uint8), ("store_mask", C.c_uint8), ("store_quality", C.c_uint8), ("pad_", C.c_uint8 * 4),
                ("err_kind", C.c_int32), ("err_char", C.c_uint32), ("err_record", C.c_uint64), ("err_a", C.c_uint64), ("err_b", C.c_uint64),
                ("unexpected", (C.c_uint64 * 257) * 4)]


class ShardPieces(C.Structure):
    _fields_ = [("off", C.c_uint64 * 6), ("len", C.c_uint64 * 6), ("raw", C.c_uint64 * 6), ("total", C.c_uint64)]


class ShardCarry(C.Structure):
    _fields_ = [("first_record", C.c_uint64), ("tail_extra", C.c_uint64), ("run_ext", C.c_uint64), ("skip_first", C.c_uint32), ("tail_hi", C.c_uint32),
                ("prev_masked", C.c_int32), ("skip_run0", C.c_int32), ("first", C.c_uint8 * 6), ("last", C.c_uint8 * 6), ("pad_", C.c_uint8 * 4)]


class StitchSeg(C.Structure):
    _fields_ = [("dst_off", C.c_uint64), ("len", C.c_uint64), ("src_off", C.c_uint64), ("shard", C.c_int32), ("stream", C.c_int32)]


class NafGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("naf_gpu error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


_lib = None


def load():
    """dlopen libnaf_gpu.so (no HIP call is made here, so this works on a machine without a GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libnaf_gpu.so is not built (run `make` or __graft_entry__.build()); there is no CPU fallback")
        # torch first, when it is there: it brings its own libamdhip64, and the library must bind to the HIP runtime the process
        # already has -- loaded the other way round (library, then torch) the process holds two runtimes and naf_gpu_init finds no
        # device.  The C hosts link /opt/rocm's runtime and never see torch.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
        L.naf_gpu_init.argtypes = [i, C.POINTER(vp)]
        L.naf_gpu_shutdown.argtypes = [vp]
        L.naf_gpu_shutdown.restype = None
        L.naf_gpu_strerror.restype = C.c_char_p
        L.naf_gpu_last_error.restype = C.c_char_p
        L.naf_gpu_last_error.argtypes = [vp]
        L.naf_gpu_set_stream.argtypes = [vp, vp]
        L.naf_gpu_synchronize.argtypes = [vp]
        L.naf_gpu_reserve.argtypes = [vp, sz]
        L.naf_gpu_histogram.argtypes = [vp, vp, sz, C.POINTER(C.c_uint64)]
        L.naf_gpu_zstd_decompress.argtypes = [vp, vp, sz, i, vp, sz, C.POINTER(sz)]
        L.naf_gpu_parse_header.argtypes = [vp, vp, sz, C.POINTER(Header)]
        L.naf_gpu_parse_header_host.argtypes = [C.c_char_p, sz, C.POINTER(Header), C.c_char_p]
        L.naf_gpu_unnaf_size.argtypes = [vp, vp, sz, C.POINTER(UnnafOpts), C.POINTER(sz)]
        L.naf_gpu_unnaf.argtypes = [vp, vp, sz, C.POINTER(UnnafOpts), vp, sz, C.POINTER(sz)]
        L.naf_gpu_unnaf_range.argtypes = [vp, vp, sz, C.POINTER(UnnafOpts), C.c_uint64, C.c_uint64, vp, sz, C.POINTER(sz)]
        L.naf_gpu_set_timing.argtypes = [vp, i]
        L.naf_gpu_get_timing.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(i), i]
        for opt in ("naf_gpu_zstd_compress", "naf_gpu_ennaf"):
            if hasattr(L, opt):
                pass
        if hasattr(L, "naf_gpu_zstd_compress"):
            L.naf_gpu_zstd_compress.argtypes = [vp, vp, sz, i, vp, sz, C.POINTER(sz)]
            L.naf_gpu_zstd_compress_bound.argtypes = [sz]
            L.naf_gpu_zstd_compress_bound.restype = sz
        if hasattr(L, "naf_gpu_ennaf"):
            L.naf_gpu_ennaf.argtypes = [vp, vp, sz, C.POINTER(EnnafOpts), vp, sz, C.POINTER(sz), C.POINTER(EnnafReport)]
            L.naf_gpu_ennaf_bound.argtypes = [sz]
            L.naf_gpu_ennaf_bound.restype = sz
        u64p = C.POINTER(C.c_uint64)
        L.naf_gpu_ennaf_sniff.argtypes = [vp, vp, sz, i, C.POINTER(i), u64p]
        L.naf_gpu_ennaf_count_lines.argtypes = [vp, vp, sz, i, u64p]
        L.naf_gpu_ennaf_find_cut.argtypes = [vp, vp, sz, i, i, C.c_uint64, u64p]
        L.naf_gpu_ennaf_shard_begin.argtypes = [vp, vp, sz, C.POINTER(EnnafOpts), i, C.c_uint32, C.c_uint32, C.POINTER(ShardInfo)]
        L.naf_gpu_ennaf_shard_bound.argtypes = [sz]
        L.naf_gpu_ennaf_shard_bound.restype = sz
        L.naf_gpu_ennaf_shard_finish.argtypes = [vp, C.POINTER(EnnafOpts), C.POINTER(ShardInfo), vp, sz, C.POINTER(ShardPieces)]
        L.naf_gpu_ennaf_shard_carry.argtypes = [C.POINTER(ShardInfo), C.c_uint32, C.c_uint32, C.POINTER(ShardCarry)]
        L.naf_gpu_ennaf_stitch_plan.argtypes = [C.POINTER(EnnafOpts), C.POINTER(ShardInfo), C.POINTER(ShardPieces), C.c_uint32, C.POINTER(StitchSeg), sz,
                                                C.POINTER(sz), C.c_char_p, sz, C.POINTER(sz), u64p, C.POINTER(EnnafReport)]
        L.naf_gpu_ennaf_stitch.argtypes = [vp, C.POINTER(StitchSeg), sz, C.c_char_p, C.POINTER(vp), vp, sz]
        L.naf_gpu_copy.argtypes = [vp, vp, vp, sz]
        L.naf_gpu_write_file.argtypes = [vp, i, C.c_uint64, vp, sz]
        L.naf_gpu_read_file.argtypes = [vp, i, C.c_uint64, sz, vp]
        L.naf_gpu_write_fd.argtypes = [vp, i, vp, sz]
        L.naf_gpu_release_scratch.argtypes = [vp]
        L.naf_gpu_get_timing_streams.argtypes = [vp, C.POINTER(C.c_float)]
        L.naf_gpu_gather_ranges.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(vp), u64p, C.POINTER(sz), i]
        L.naf_gpu_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.naf_gpu_get_trace.argtypes = [vp]
        L.naf_gpu_get_trace.restype = C.c_char_p
        L.naf_gpu_clear_trace.argtypes = [vp]
        L.naf_gpu_clear_trace.restype = None
        _lib = L
    return _lib


def shard_carry(infos, k):
    """Host-only: what shard k's finish applies on behalf of its neighbours (naf_gpu_ennaf_shard_carry)."""
    L = load()
    arr = (ShardInfo * len(infos))(*infos)
    out = ShardCarry()
    rc = L.naf_gpu_ennaf_shard_carry(arr, len(infos), k, C.byref(out))
    if rc:
        raise NafGpuError(rc, L.naf_gpu_strerror(rc).decode())
    return out


def stitch_plan(opts, infos, pieces):
    """Host-only: (segments, literal bytes, archive length, report) of the joined archive (naf_gpu_ennaf_stitch_plan)."""
    L = load()
    n = len(infos)
    ia = (ShardInfo * n)(*infos)
    pa = (ShardPieces * n)(*pieces)
    segs = (StitchSeg * (7 + 6 * n))()
    title = opts.title or b""
    lit = C.create_string_buffer(256 + len(title))
    ns, ll, nl = C.c_size_t(), C.c_size_t(), C.c_uint64()
    rep = EnnafReport()
    rc = L.naf_gpu_ennaf_stitch_plan(C.byref(opts), ia, pa, n, segs, len(segs), C.byref(ns), lit, len(lit), C.byref(ll), C.byref(nl), C.byref(rep))
    if rc:
        raise NafGpuError(rc, L.naf_gpu_strerror(rc).decode())
    return [segs[k] for k in range(ns.value)], lit.raw[:ll.value], nl.value, rep


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class _SyncedLib:
    """The library as a Context sees it.  The library reads its NAF_GPU_* switches from the environment once, in naf_gpu_init, and
    afterwards only through naf_gpu_set_option; tests and tools flip switches between calls by changing os.environ, so every entry point
    taken through this proxy first hands the library what changed since the last one (a NAF_GPU_DEBUG_* variable turns TRACE on), and a
    trace a call left is written to stderr when the call returns -- where the tests read which path ran."""

    def __init__(self, lib, ctx):
        object.__setattr__(self, "_lib", lib)
        object.__setattr__(self, "_ctx", ctx)

    def __getattr__(self, name):
        f = getattr(self._lib, name)
        ctx = self._ctx
        if not ctx.h or name in ("naf_gpu_set_option", "naf_gpu_get_trace", "naf_gpu_clear_trace", "naf_gpu_last_error", "naf_gpu_strerror", "naf_gpu_shutdown"):
            return f
        ctx._sync_options()
        if not ctx._tracing:
            return f

        def traced(*a):
            try:
                return f(*a)
            finally:
                t = self._lib.naf_gpu_get_trace(ctx.h)
                if t:
                    os.write(2, t)
                    self._lib.naf_gpu_clear_trace(ctx.h)
        return traced


def _env_options():
    o = {k[8:]: v for k, v in os.environ.items() if k.startswith("NAF_GPU_")}
    if any(k.startswith("DEBUG_") for k in o) and "TRACE" not in o:
        o["TRACE"] = "1"
    return o


class Context:
    """One per device / per process rank.  Work is enqueued on torch's current stream for the device."""

    def __init__(self, device=0, use_torch_stream=True):
        import torch
        self.h = C.c_void_p()
        self._opts = {}
        self._tracing = False
        self.L = _SyncedLib(load(), self)
        rc = load().naf_gpu_init(device, C.byref(self.h))
        if rc:
            raise NafGpuError(rc, load().naf_gpu_strerror(rc).decode())
        self._opts = {k[8:]: v for k, v in os.environ.items() if k.startswith("NAF_GPU_")}      # what naf_gpu_init has read
        self.device = torch.device("cuda", device)
        if use_torch_stream:
            s = torch.cuda.current_stream(self.device)
            self._check(self.L.naf_gpu_set_stream(self.h, C.c_void_p(s.cuda_stream)))

    def close(self):
        if self.h:
            self.L.naf_gpu_shutdown(self.h)
            self.h = C.c_void_p()

    def set_option(self, name, value):
        """naf_gpu_set_option: one switch of this context (value None: not set)."""
        self._check(load().naf_gpu_set_option(self.h, name.encode(), None if value is None else str(value).encode()))

    def _sync_options(self):
        # (os.environ is looked through per call: cheap beside a call, but not inside a timed loop of short ones -- the raw items are
        # compared first, the dictionary work only when one of them changed)
        key = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("NAF_GPU_")))
        if key == getattr(self, "_opts_key", None):
            return
        self._opts_key = key
        want = _env_options()
        if want != self._opts:
            for k in set(self._opts) - set(want):
                self.set_option(k, None)
            for k, v in want.items():
                if self._opts.get(k) != v:
                    self.set_option(k, v)
            self._opts = want
        self._tracing = want.get("TRACE", "")[:1] == "1"

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise NafGpuError(rc, self.L.naf_gpu_last_error(self.h).decode("latin1"))

    def to_device(self, data: bytes):
        import torch
        import numpy as np
        if len(data) == 0:
            return torch.empty(0, dtype=torch.uint8, device=self.device)
        return torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(self.device)

    def reserve(self, nbytes):
        self._check(self.L.naf_gpu_reserve(self.h, nbytes))

    def write_file(self, fd, file_off, t):
        """naf_gpu_write_file: the bytes of device tensor t at file_off of descriptor fd (a regular file)."""
        self._check(self.L.naf_gpu_write_file(self.h, int(fd), C.c_uint64(int(file_off)), _ptr(t), int(t.numel())))

    def release_scratch(self):
        """Give the context's scratch arena back to the device (the next call grows it again)."""
        self._check(self.L.naf_gpu_release_scratch(self.h))

    def gather_ranges(self, out, parts):
        """naf_gpu_gather_ranges: parts = [(ctx, tensor, dst_offset)] -- every part is pushed into `out` (a tensor on this context's
        device) by its own context's stream; this context's stream waits for all of them."""
        n = len(parts)
        srcs = (C.c_void_p * n)(*[p[0].h.value for p in parts])
        ptrs = (C.c_void_p * n)(*[p[1].data_ptr() for p in parts])
        offs = (C.c_uint64 * n)(*[int(p[2]) for p in parts])
        lens = (C.c_size_t * n)(*[int(p[1].numel()) for p in parts])
        self._check(self.L.naf_gpu_gather_ranges(self.h, _ptr(out), srcs, ptrs, offs, lens, n))

    # ---- zstd ----
    def zstd_decompress(self, d_frame, out_cap, has_magic=True):
        import torch
        out = torch.empty(max(out_cap, 1), dtype=torch.uint8, device=self.device)
        n = C.c_size_t()
        self._check(self.L.naf_gpu_zstd_decompress(self.h, _ptr(d_frame), d_frame.numel(), int(has_magic), _ptr(out), out_cap, C.byref(n)))
        return out[:n.value]

    def zstd_compress(self, d_src, level=1):
        import torch
        cap = self.L.naf_gpu_zstd_compress_bound(d_src.numel())
        out = torch.empty(cap, dtype=torch.uint8, device=self.device)
        n = C.c_size_t()
        self._check(self.L.naf_gpu_zstd_compress(self.h, _ptr(d_src), d_src.numel(), level, _ptr(out), cap, C.byref(n)))
        return out[:n.value]

    # ---- unnaf ----
    def parse_header(self, d_naf):
        h = Header()
        self._check(self.L.naf_gpu_parse_header(self.h, _ptr(d_naf), d_naf.numel(), C.byref(h)))
        return h

    def unnaf_size(self, d_naf, out_type=OUT_DEFAULT, use_mask=True, line_length=-1):
        o = UnnafOpts(out_type, int(use_mask), line_length)
        n = C.c_size_t()
        self._check(self.L.naf_gpu_unnaf_size(self.h, _ptr(d_naf), d_naf.numel(), C.byref(o), C.byref(n)))
        return n.value

    def unnaf(self, d_naf, out_type=OUT_DEFAULT, use_mask=True, line_length=-1, out=None):
        import torch
        o = UnnafOpts(out_type, int(use_mask), line_length)
        if out is None:
            size = self.unnaf_size(d_naf, out_type, use_mask, line_length)
            out = torch.empty(max(size, 1), dtype=torch.uint8, device=self.device)
        n = C.c_size_t()
        self._check(self.L.naf_gpu_unnaf(self.h, _ptr(d_naf), d_naf.numel(), C.byref(o), _ptr(out), out.numel(), C.byref(n)))
        return out[:n.value]

    def unnaf_range(self, d_naf, begin, end, out_type=OUT_DEFAULT, use_mask=True, line_length=-1, out=None):
        import torch
        o = UnnafOpts(out_type, int(use_mask), line_length)
        if out is None:
            out = torch.empty(max(end - begin, 1), dtype=torch.uint8, device=self.device)
        n = C.c_size_t()
        self._check(self.L.naf_gpu_unnaf_range(self.h, _ptr(d_naf), d_naf.numel(), C.byref(o), begin, end, _ptr(out), out.numel(), C.byref(n)))
        return out[:n.value]

    def histogram(self, d_buf):
        """Byte counts of a device buffer (unnaf --charcount)."""
        cnt = (C.c_uint64 * 256)()
        self._check(self.L.naf_gpu_histogram(self.h, _ptr(d_buf), d_buf.numel(), cnt))
        return list(cnt)

    # ---- ennaf ----
    def ennaf(self, d_text, seq_type=SEQ_DNA, fmt=FMT_AUTO, no_mask=False, level=1, line_length=-1, title=None, out=None, strict=False, long_log=0):
        import torch
        o = EnnafOpts(fmt, seq_type, int(no_mask), int(strict), level, line_length, title, long_log)
        if out is None:
            cap = self.L.naf_gpu_ennaf_bound(d_text.numel())
            out = torch.empty(cap, dtype=torch.uint8, device=self.device)
        n = C.c_size_t()
        rep = EnnafReport()
        self._check(self.L.naf_gpu_ennaf(self.h, _ptr(d_text), d_text.numel(), C.byref(o), _ptr(out), out.numel(), C.byref(n), C.byref(rep)))
        return out[:n.value], rep

    # ---- ennaf of one input on several GPUs: the per-shard calls (orchestration in naf_amd/shard.py) ----
    def ennaf_sniff(self, d_text, fmt=FMT_AUTO):
        f, p0 = C.c_int(), C.c_uint64()
        self._check(self.L.naf_gpu_ennaf_sniff(self.h, _ptr(d_text), d_text.numel(), fmt, C.byref(f), C.byref(p0)))
        return f.value, p0.value

    def ennaf_count_lines(self, d_slice, prev_is_eol):
        n = C.c_uint64()
        self._check(self.L.naf_gpu_ennaf_count_lines(self.h, _ptr(d_slice), d_slice.numel(), int(prev_is_eol), C.byref(n)))
        return n.value

    def ennaf_find_cut(self, d_slice, fmt, prev_is_eol, skip_lines=0):
        off = C.c_uint64()
        self._check(self.L.naf_gpu_ennaf_find_cut(self.h, _ptr(d_slice), d_slice.numel(), fmt, int(prev_is_eol), skip_lines, C.byref(off)))
        return off.value

    def ennaf_shard_begin(self, d_slice, opts, fmt, shard, n_shards):
        info = ShardInfo()
        self._check(self.L.naf_gpu_ennaf_shard_begin(self.h, _ptr(d_slice), d_slice.numel(), C.byref(opts), fmt, shard, n_shards, C.byref(info)))
        return info

    def ennaf_shard_finish(self, opts, infos, text_len):
        import torch
        cap = self.L.naf_gpu_ennaf_shard_bound(text_len)
        buf = torch.empty(cap, dtype=torch.uint8, device=self.device)
        arr = (ShardInfo * len(infos))(*infos)
        pc = ShardPieces()
        self._check(self.L.naf_gpu_ennaf_shard_finish(self.h, C.byref(opts), arr, _ptr(buf), cap, C.byref(pc)))
        return buf, pc

    def ennaf_stitch(self, segs, lit, bufs, out):
        sa = (StitchSeg * len(segs))(*segs)
        pa = (C.c_void_p * len(bufs))(*[b.data_ptr() for b in bufs])
        self._check(self.L.naf_gpu_ennaf_stitch(self.h, sa, len(segs), lit, pa, _ptr(out), out.numel()))

    # ---- timing ----
    def set_timing(self, on):
        self._check(self.L.naf_gpu_set_timing(self.h, int(on)))

    def get_timing(self):
        cap = 160
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        cnt = (C.c_int * cap)()
        n = self.L.naf_gpu_get_timing(self.h, names, ms, cnt, cap)
        return [(names[i].decode(), ms[i], cnt[i]) for i in range(n)]

    def get_timing_streams(self):
        """Kernel time of the last get_timing() per stream: [caller's stream, side chain 1..4]."""
        ms = (C.c_float * 5)()
        self._check(self.L.naf_gpu_get_timing_streams(self.h, ms))
        return [float(x) for x in ms]

    def synchronize(self):
        self._check(self.L.naf_gpu_synchronize(self.h))
