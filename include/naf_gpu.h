/*
 * naf_gpu.h -- C-ABI of libnaf_gpu.so: the MI355X (gfx950) implementation of the ennaf/unnaf hot path.
 *
 * The reference (KirillKryukov/naf v1.3.0) has no plugin/FFI API; its hot path is reached through
 * three internal seams, and each entry point below replaces one of them (paths are relative to the
 * reference tree):
 *
 *   naf_gpu_zstd_decompress   <- ZSTD_decompress / ZSTD_decompressStream call sites
 *                                unnaf/src/input.c:155,183,212,230 (one-shot sections) and
 *                                input.c:262-285,368,399,426, output.c:646 (streamed sequence/quality)
 *   naf_gpu_unnaf             <- print_fasta / print_fastq / print_dna / print_sequences / print_4bit
 *                                unnaf/src/output.c:608-674, output-fastq.c:100-149, output.c:457-512,
 *                                output-sequences.c:60-116, output.c:266-292 (incl. write_4bit_as_fasta
 *                                output.c:445, mask_dna_buffer output.c:295, print_dna_split_into_lines :339)
 *   naf_gpu_ennaf             <- process() + the seq/name/comm/qual writers + compress() + section writer
 *                                ennaf/src/process.c:586-615 (parsers :314-544), process.c:12-57,
 *                                encoders.c:30-146, compressor.c:119-147, ennaf.c:538-589
 *   naf_gpu_zstd_compress     <- compress()/compressor_end_stream(), ennaf/src/compressor.c:64-147
 *
 * Conventions: extern "C", plain pointers and sizes, no C++/torch types.  Every function returns 0
 * on success or a negative NAF_GPU_E* code; naf_gpu_last_error(ctx) gives the text (for parity with
 * the reference's die() strings where the reference defines one).  One ctx per device and per host
 * thread; calls on a ctx are serialised on its HIP stream.  "d_" pointers are device (HBM) addresses,
 * "h_" pointers are host addresses.  There is NO CPU fallback: without a usable gfx950 device
 * naf_gpu_init fails with NAF_GPU_ENODEV.
 */
#ifndef NAF_GPU_H
#define NAF_GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct naf_gpu_ctx naf_gpu_ctx;

enum {
    NAF_GPU_OK = 0,
    NAF_GPU_ENODEV = -1,     /* no HIP device / wrong architecture */
    NAF_GPU_EHIP = -2,       /* HIP runtime error (text in last_error) */
    NAF_GPU_ENOMEM = -3,     /* device workspace allocation failed */
    NAF_GPU_EFORMAT = -4,    /* malformed .naf container (reference: die() in input.c:31-77) */
    NAF_GPU_EZSTD = -5,      /* corrupt / unsupported zstd frame */
    NAF_GPU_ECAP = -6,       /* output capacity too small; required size is reported */
    NAF_GPU_EINPUT = -7,     /* input text rejected (reference die() messages of process.c) */
    NAF_GPU_EARG = -8
};

/* sequence types (NAF header byte; ennaf.c:52, unnaf.c:27) and text formats (ennaf.c:47) */
enum { NAF_SEQ_DNA = 0, NAF_SEQ_RNA = 1, NAF_SEQ_PROTEIN = 2, NAF_SEQ_TEXT = 3 };
enum { NAF_FMT_AUTO = 0, NAF_FMT_FASTA = 1, NAF_FMT_FASTQ = 2 };
/* unnaf output types that produce sequence text (unnaf.c:16-24) */
enum { NAF_OUT_DEFAULT = -1, NAF_OUT_FASTA = 0, NAF_OUT_FASTQ = 1, NAF_OUT_SEQ = 2, NAF_OUT_SEQUENCES = 3, NAF_OUT_4BIT = 4 };

/* ---- context ---------------------------------------------------------------------------------- */
int         naf_gpu_init(int device, naf_gpu_ctx **ctx);
void        naf_gpu_shutdown(naf_gpu_ctx *ctx);
const char *naf_gpu_strerror(int code);
const char *naf_gpu_last_error(const naf_gpu_ctx *ctx);
/* Run on a caller-owned hipStream_t (e.g. the framework's current stream).  NULL = HIP's default
 * (null) stream, as in the HIP API.  Until called, the ctx uses a private non-blocking stream. */
int         naf_gpu_set_stream(naf_gpu_ctx *ctx, void *hip_stream);
int         naf_gpu_synchronize(naf_gpu_ctx *ctx);
/* Pre-size the internal scratch arena (otherwise grown on demand; growth synchronises).  The contexts of the side chains
 * (side sections, a FASTQ's quality stream, the chains behind an encode's split) get an eighth of `bytes` each, at most
 * 4 GiB.  A whole call (unnaf, unnaf_range, ennaf) that had to grow an arena leaves it as ONE allocation when it returns:
 * the growing call pays, the call after it is already in steady state. */
int         naf_gpu_reserve(naf_gpu_ctx *ctx, size_t bytes);

/* ---- device memory for hosts that do not link HIP themselves (the C CLIs) ---------------------------- */
int  naf_gpu_malloc(naf_gpu_ctx *ctx, size_t bytes, void **d_ptr);
int  naf_gpu_free(naf_gpu_ctx *ctx, void *d_ptr);
/* free / total device memory as the runtime sees it now (the hosts size their chunks of an input larger than HBM with it);
 * the scratch arena of the context counts as used: naf_gpu_release_scratch gives it back */
int  naf_gpu_mem_info(naf_gpu_ctx *ctx, size_t *free_bytes, size_t *total_bytes);
int  naf_gpu_release_scratch(naf_gpu_ctx *ctx);
int  naf_gpu_host_alloc(naf_gpu_ctx *ctx, size_t bytes, void **h_pinned);
int  naf_gpu_host_free(naf_gpu_ctx *ctx, void *h_pinned);
int  naf_gpu_upload(naf_gpu_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);     /* async on the stream */
int  naf_gpu_download(naf_gpu_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);   /* returns after completion */
int  naf_gpu_download_async(naf_gpu_ctx *ctx, void *h_pinned_dst, const void *d_src, size_t bytes);   /* async on the stream; pair with naf_gpu_synchronize */
int  naf_gpu_copy(naf_gpu_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);         /* device -> device, async on the stream */

/* ---- the collective of the decode path (BASELINE configs[3]: "per-GPU frame ranges, gather") ----------------
 * One process, one context per GPU (what the C hosts do under NAF_GPUS=0,1,...): ctx k decodes its byte range of the text with
 * naf_gpu_unnaf_range on its own device; this call brings the ranges together in d_dst on dst's device -- every source pushes
 * its range over its own xGMI link (peer copy on the source's stream, behind its decode), dst's stream waits for all of them.
 * srcs[k] may be dst itself (its range is copied in place, or left where it is when d_src[k] already lies at its offset).
 * Decision on RCCL (north_star names "an RCCL gather over xGMI"): between the GPUs of ONE process a gather-to-root is N - 1
 * point-to-point pushes whatever library issues them, bound by the root's xGMI ingress (7 links x ~153 GB/s) -- this entry
 * issues exactly those and libnaf_gpu.so stays free of a communicator.  Jobs of one PROCESS per GPU (torch.distributed; bench.py
 * and the tests) do the same exchange as one group of RCCL send/recv (naf_amd/shard.py: gather_ranges).  When the consumer is the
 * HOST (a file, a pipe), no gather between GPUs is wanted at all: every GPU downloads its range over its own PCIe link
 * (naf_gpu_write_file / naf_gpu_download_async per context -- unnaf.c under NAF_GPUS). */
int  naf_gpu_gather_ranges(naf_gpu_ctx *dst, void *d_dst, naf_gpu_ctx *const *srcs, const void *const *d_src,
                           const uint64_t *dst_off, const size_t *len, int n);

/* File <-> HBM through pinned staging on several host threads (io.hip; NAF_GPU_IO_THREADS, default 8): what the reference does with
 * fread / fwrite of 16 KiB (ennaf/src/process.c:143-150, unnaf/src/files.c).  fd must support pread / pwrite (a regular file); the
 * calls return when the transfer is complete.  The copies run on the ctx stream, in order behind whatever was queued there (the
 * kernels that make d_src). */
int  naf_gpu_read_file(naf_gpu_ctx *ctx, int fd, uint64_t file_off, size_t len, void *d_dst);
int  naf_gpu_write_file(naf_gpu_ctx *ctx, int fd, uint64_t file_off, const void *d_src, size_t len);
/* The same for a descriptor that takes its bytes in order (a pipe, a terminal, /dev/null, `>>`): write() from the calling thread, the
 * next chunks already on the link.  The reference's counterpart is fwrite to stdout (unnaf/src/output.c:640-651). */
int  naf_gpu_write_fd(naf_gpu_ctx *ctx, int fd, const void *d_src, size_t len);

/* Byte histogram of a device buffer (unnaf --charcount over the --seq text, output.c:515-605). */
int  naf_gpu_histogram(naf_gpu_ctx *ctx, const void *d_buf, size_t n, uint64_t counts[256]);

/* ---- zstd ----------------------------------------------------------------------------------------- */
/* Decode one or more concatenated zstd frames (RFC 8878, no dictionaries) resident in HBM.
 * has_magic = 0: the first frame lacks its 4-byte magic, exactly as stored inside a .naf section
 * (compressor.c:150-173 strips it, unnaf utils.c:144-150 re-adds it).  *out_len receives the decoded
 * size; with NAF_GPU_ECAP it receives the required size. */
int  naf_gpu_zstd_decompress(naf_gpu_ctx *ctx, const void *d_src, size_t src_len, int has_magic,
                             void *d_dst, size_t dst_cap, size_t *out_len);

/* Compress d_src into ONE zstd frame made of independently coded blocks (single frame: SURVEY.md R1).
 * level <= 1: entropy-only blocks (Huffman literals, RLE, raw), Huffman weights written directly wherever the format allows
 * (up to 128 weights); level >= 2 adds the LZ stage (matches inside a block, coded with the predefined FSE sequence tables)
 * and FSE-codes the Huffman weights when that is smaller.  Blocks never depend on each other at any level. */
int  naf_gpu_zstd_compress(naf_gpu_ctx *ctx, const void *d_src, size_t src_len, int level,
                           void *d_dst, size_t dst_cap, size_t *out_len);
size_t naf_gpu_zstd_compress_bound(size_t src_len);

/* ---- unnaf ------------------------------------------------------------------------------------------ */
typedef struct {
    int      out_type;          /* NAF_OUT_* ; DEFAULT = FASTQ if the archive has quality else FASTA */
    int      use_mask;          /* 0 = --no-mask */
    int64_t  line_length;       /* <0: use the stored value; >=0: --line-length N (0 = no wrapping) */
} naf_gpu_unnaf_opts;

typedef struct {
    int      version, seq_type, flags;
    uint8_t  separator;
    uint64_t line_length, n_sequences;
    uint64_t title_off, title_len;
    uint64_t orig_size[6], comp_size[6], payload_off[6];   /* ids, comments, lengths, mask, sequence, quality */
} naf_gpu_header;

/* Parse the container framing of an archive resident in HBM (header bytes are pulled to the host). */
int  naf_gpu_parse_header(naf_gpu_ctx *ctx, const void *d_naf, size_t naf_len, naf_gpu_header *hdr);
/* Same for an archive in host memory (used by the CLI before upload). */
int  naf_gpu_parse_header_host(const void *h_naf, size_t naf_len, naf_gpu_header *hdr, char errbuf[128]);

/* Exact size of the text naf_gpu_unnaf will produce (runs the small-section decode + offset scans). */
int  naf_gpu_unnaf_size(naf_gpu_ctx *ctx, const void *d_naf, size_t naf_len, const naf_gpu_unnaf_opts *opts,
                        size_t *out_len);
/* Archive in HBM -> FASTA/FASTQ/... text in HBM.  Bit-exact with reference unnaf on the same archive. */
int  naf_gpu_unnaf(naf_gpu_ctx *ctx, const void *d_naf, size_t naf_len, const naf_gpu_unnaf_opts *opts,
                   void *d_out, size_t out_cap, size_t *out_len);
/* Multi-GPU sharding: produce only output bytes [out_begin, out_end) of the full text into d_out
 * (d_out[0] = byte out_begin).  Only the zstd blocks that feed that byte range are decoded. */
int  naf_gpu_unnaf_range(naf_gpu_ctx *ctx, const void *d_naf, size_t naf_len, const naf_gpu_unnaf_opts *opts,
                         uint64_t out_begin, uint64_t out_end, void *d_out, size_t out_cap, size_t *out_len);

/* ---- ennaf ------------------------------------------------------------------------------------------ */
typedef struct {
    int      format;            /* NAF_FMT_* (AUTO = sniff, process.c:547-583) */
    int      seq_type;          /* NAF_SEQ_* */
    int      no_mask;           /* --no-mask */
    int      strict;            /* --strict */
    int      level;             /* --level: 1 = ids / names / lengths matched inside a block, other streams entropy-coded; >= 2 = every stream
                                 * matched across blocks inside libzstd's window for that level */
    int64_t  line_length;       /* <0: store the longest line; >=0: --line-length N */
    const char *title;          /* --title or NULL */
    int      long_log;          /* --long N (ennaf.c:247-273): window 2^N for the sequence stream; 0 = not given */
} naf_gpu_ennaf_opts;

typedef struct {
    int      format;            /* detected NAF_FMT_* (0 = empty input) */
    uint64_t n_sequences, n_bases, longest_line;
    uint64_t unexpected_id[257], unexpected_comment[257], unexpected_seq[257], unexpected_qual[257];
    uint64_t section_orig[6], section_comp[6];
} naf_gpu_ennaf_report;

size_t naf_gpu_ennaf_bound(size_t text_len);
/* FASTA/FASTQ text in HBM -> complete .naf archive bytes in HBM. */
int  naf_gpu_ennaf(naf_gpu_ctx *ctx, const void *d_text, size_t text_len, const naf_gpu_ennaf_opts *opts,
                   void *d_naf, size_t naf_cap, size_t *naf_len, naf_gpu_ennaf_report *report);

/* ---- ennaf of ONE input on several GPUs (BASELINE configs[4]) ------------------------------------------------
 * The text is cut into consecutive slices, one per GPU ("shard"); every shard runs the same kernels as naf_gpu_ennaf on its
 * slice and the parts are joined into ONE archive with ONE zstd frame per stream, as reference unnaf requires.
 * What the reference carries from chunk to chunk in its static state travels between shards in a small fixed-size record:
 *   - the half-filled byte of the 4-bit packer   `parity` + pending nibble, ennaf/src/encoders.c:30-69, flushed ennaf.c:525-529
 *   - the open soft-mask run                      `mask_on` / `mask_len`,    encoders.c:126-146, flushed ennaf.c:511
 *   - the record a cut falls into                 add_length(),              process.c:424 (FASTA slices may start inside a record)
 *   - n_sequences, seq_size_original, longest_line_length, the unexpected-character tallies (process.c:389-393,424-425)
 * Protocol (the same calls whether the shards are threads of one process or ranks of torch.distributed / MPI):
 *   0. naf_gpu_ennaf_sniff on the start of the text -> format and p0; slices start at p0.
 *      Cuts: FASTA behind any EOL-class byte (naf_gpu_ennaf_find_cut); FASTQ at a line start whose ordinal is a multiple of 4
 *      (naf_gpu_ennaf_count_lines of every nominal slice, prefix sum, naf_gpu_ennaf_find_cut with the lines to skip).
 *   1. every shard: naf_gpu_ennaf_shard_begin(slice) -> naf_gpu_shard_info.            [all-gather the infos]
 *   2. every shard: naf_gpu_ennaf_shard_finish(all infos) -> its parts of the six frames in a device buffer + their sizes.
 *                                                                                      [gather the naf_gpu_shard_pieces]
 *   3. anyone: naf_gpu_ennaf_stitch_plan -> where every part and the framing bytes go; move the bytes (naf_gpu_ennaf_stitch for
 *      buffers one process can address, RCCL send/recv or per-GPU D2H + pwrite otherwise).
 * An error of any shard (strict mode, malformed FASTQ) is reported by every shard's finish with the reference's message and
 * the record numbered across shards. */
enum { NAF_GPU_MAX_SHARDS = 64 };
typedef struct {
    uint32_t shard, n_shards;
    int32_t  format, seq_type;
    uint64_t text_len;
    uint64_t n_sequences, n_bases, longest_line;
    uint64_t lead_bases;                    /* FASTA: bases in front of the slice's first header (they end an earlier shard's record) */
    uint64_t n_ids, n_comments, n_quality;  /* bytes of the slice's ids / comments / quality streams */
    uint64_t mask_changes;                  /* case changes at base positions >= 1 of the slice, the first and the last of them */
    uint64_t mask_first_change, mask_last_change;
    uint8_t  first_base, last_base;         /* post-replacement; valid when n_bases > 0 */
    uint8_t  store_mask, store_quality, pad_[4];
    int32_t  err_kind; uint32_t err_char;   /* 0 = none; see naf_gpu_ennaf_shard_finish */
    uint64_t err_record, err_a, err_b;
    uint64_t unexpected[4][257];            /* id, comment, sequence, quality (process.c:75-96) */
} naf_gpu_shard_info;

typedef struct {
    uint64_t off[6], len[6];                /* this shard's part of each stream's frame inside its piece buffer (ids, comments, lengths, mask, sequence, quality) */
    uint64_t raw[6];                        /* uncompressed bytes behind each part (sequence: bases) */
    uint64_t total;                         /* bytes used in the piece buffer */
} naf_gpu_shard_pieces;

typedef struct {
    uint64_t dst_off, len, src_off;         /* archive offset; source offset inside the shard's piece buffer, or inside `lit` */
    int32_t  shard, stream;                 /* shard < 0: framing bytes from `lit` */
} naf_gpu_stitch_seg;

/* process.c:547-583: format of the text and the offset of its first record.  *format = 0 for an empty / all-space text. */
int  naf_gpu_ennaf_sniff(naf_gpu_ctx *ctx, const void *d_text, size_t text_len, int want_format, int *format, uint64_t *p0);
/* FASTQ: line starts (non-EOL byte behind an EOL-class byte) inside a slice; prev_is_eol: the byte in front of the slice is
 * EOL-class, or the slice starts at p0. */
int  naf_gpu_ennaf_count_lines(naf_gpu_ctx *ctx, const void *d_slice, size_t len, int prev_is_eol, uint64_t *n_line_starts);
/* Where a shard may begin inside a slice: FASTA -- the first byte behind an EOL-class byte; FASTQ -- line start number
 * skip_lines (0-based) of the slice.  *offset = len when the slice holds no such place. */
int  naf_gpu_ennaf_find_cut(naf_gpu_ctx *ctx, const void *d_slice, size_t len, int format, int prev_is_eol, uint64_t skip_lines,
                            uint64_t *offset);
/* Step 1.  format: NAF_FMT_FASTA or NAF_FMT_FASTQ (from the sniff).  The slice must stay in place until the finish. */
int  naf_gpu_ennaf_shard_begin(naf_gpu_ctx *ctx, const void *d_slice, size_t len, const naf_gpu_ennaf_opts *opts, int format,
                               uint32_t shard, uint32_t n_shards, naf_gpu_shard_info *info);
size_t naf_gpu_ennaf_shard_bound(size_t slice_len);
/* Step 2.  infos[n_shards] in shard order (this context's own among them).  NAF_GPU_EINPUT + last_error when any shard failed. */
int  naf_gpu_ennaf_shard_finish(naf_gpu_ctx *ctx, const naf_gpu_ennaf_opts *opts, const naf_gpu_shard_info *infos,
                                void *d_pieces, size_t cap, naf_gpu_shard_pieces *pieces);
/* What the finish of shard `shard` applies on behalf of its neighbours, derived from the infos alone (host only, no device needed;
 * exported so that hosts can log it and tests can check it against the reference's chunk-to-chunk state). */
typedef struct {
    uint64_t first_record;      /* records in the shards in front (numbers the reference's messages) */
    uint64_t tail_extra;        /* bases of later shards that belong to this shard's last record */
    uint64_t run_ext;           /* bases of later shards that continue this shard's last soft-mask run */
    uint32_t skip_first;        /* 1: the shard's first base is the high nibble of an earlier shard's last packed byte */
    uint32_t tail_hi;           /* 4-bit code completing the shard's last packed byte when its pack window is odd */
    int32_t  prev_masked;       /* case in front of the shard's first base */
    int32_t  skip_run0;         /* 1: the bases in front of the shard's first case change are emitted by an earlier shard's run */
    uint8_t  first[6], last[6]; /* per stream: this part opens the frame (2-byte header) / closes it (last-block flag) */
    uint8_t  pad_[4];
} naf_gpu_shard_carry;
int  naf_gpu_ennaf_shard_carry(const naf_gpu_shard_info *infos, uint32_t n_shards, uint32_t shard, naf_gpu_shard_carry *out);
/* Step 3 (host only, no device needed).  segs: at least 7 + 6 * n_shards entries; lit: at least 256 + strlen(title) bytes.
 * report (optional) receives the totals of the whole input. */
int  naf_gpu_ennaf_stitch_plan(const naf_gpu_ennaf_opts *opts, const naf_gpu_shard_info *infos, const naf_gpu_shard_pieces *pieces,
                               uint32_t n_shards, naf_gpu_stitch_seg *segs, size_t seg_cap, size_t *n_segs,
                               uint8_t *lit, size_t lit_cap, size_t *lit_len, uint64_t *naf_len, naf_gpu_ennaf_report *report);
/* Execute a plan when this process can address every piece buffer (one device, or peers with access enabled). */
int  naf_gpu_ennaf_stitch(naf_gpu_ctx *ctx, const naf_gpu_stitch_seg *segs, size_t n_segs, const uint8_t *lit,
                          const void *const *d_piece_bufs, void *d_naf, size_t naf_cap);

/* ---- switches ------------------------------------------------------------------------------------------
 * The library's cross-check levers and development aids (INTEGRATION.md section 6 lists them).  naf_gpu_init reads the NAF_GPU_<NAME>
 * variables of the environment ONCE into the context; no later call looks at the environment.  naf_gpu_set_option changes one
 * afterwards (name with or without the NAF_GPU_ prefix; value NULL = not set).  None is needed in production.
 * TRACE=1: the calls keep the verdicts of the paths they took ("[flat mixed] nblk ... decoded ...") as text in the context instead of
 * printing anything; naf_gpu_get_trace returns what has accumulated (valid until the next call on the context), naf_gpu_clear_trace
 * empties it.  The tests ask through this which kernels ran. */
int         naf_gpu_set_option(naf_gpu_ctx *ctx, const char *name, const char *value);
const char *naf_gpu_get_trace(naf_gpu_ctx *ctx);
void        naf_gpu_clear_trace(naf_gpu_ctx *ctx);

/* ---- instrumentation ---------------------------------------------------------------------------------- */
/* Per-kernel device time (hipEvent pairs on the ctx stream) of the last call, for bench.py's roofline
 * object.  names[i] points to static strings.  Returns the number of entries written (<= cap). */
int  naf_gpu_set_timing(naf_gpu_ctx *ctx, int enable);
int  naf_gpu_get_timing(naf_gpu_ctx *ctx, const char **names, float *ms, int *launches, int cap);
/* kernel time of that aggregation per stream: [0] the caller's stream, [1..4] the side chains' (a call is at least the largest) */
int  naf_gpu_get_timing_streams(naf_gpu_ctx *ctx, float ms[5]);

#ifdef __cplusplus
}
#endif
#endif
