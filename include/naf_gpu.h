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
/* Pre-size the internal scratch arena (otherwise grown on demand; growth synchronises). */
int         naf_gpu_reserve(naf_gpu_ctx *ctx, size_t bytes);

/* ---- device memory for hosts that do not link HIP themselves (the C CLIs) ---------------------------- */
int  naf_gpu_malloc(naf_gpu_ctx *ctx, size_t bytes, void **d_ptr);
int  naf_gpu_free(naf_gpu_ctx *ctx, void *d_ptr);
int  naf_gpu_host_alloc(naf_gpu_ctx *ctx, size_t bytes, void **h_pinned);
int  naf_gpu_host_free(naf_gpu_ctx *ctx, void *h_pinned);
int  naf_gpu_upload(naf_gpu_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);     /* async on the stream */
int  naf_gpu_download(naf_gpu_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);   /* returns after completion */
int  naf_gpu_download_async(naf_gpu_ctx *ctx, void *h_pinned_dst, const void *d_src, size_t bytes);   /* async on the stream; pair with naf_gpu_synchronize */

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
    int      level;             /* --level: ids / names / lengths always take the LZ stage; >= 2 extends it to mask, sequence, quality */
    int64_t  line_length;       /* <0: store the longest line; >=0: --line-length N */
    const char *title;          /* --title or NULL */
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

/* ---- instrumentation ---------------------------------------------------------------------------------- */
/* Per-kernel device time (hipEvent pairs on the ctx stream) of the last call, for bench.py's roofline
 * object.  names[i] points to static strings.  Returns the number of entries written (<= cap). */
int  naf_gpu_set_timing(naf_gpu_ctx *ctx, int enable);
int  naf_gpu_get_timing(naf_gpu_ctx *ctx, const char **names, float *ms, int *launches, int cap);

#ifdef __cplusplus
}
#endif
#endif
