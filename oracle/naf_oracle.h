/*
 * naf_oracle.h -- TEST INFRASTRUCTURE ONLY (the "oracle").
 *
 * CPU restatement, in plain C, of the reference ennaf/unnaf hot path (KirillKryukov/naf v1.3.0)
 * and of the zstd frame format it embeds (facebook/zstd, pinned by the reference only in prose at
 * v1.5.0 -- CHANGELOG.md:10; the zstd/ submodule is empty, so the zstd part restates the published
 * format, RFC 8878).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (naf_amd/, libnaf_gpu.so, ennaf/unnaf CLIs) never links or calls it.
 *
 * Parity pinning: tests/test_oracle_*.py check every function here against
 *   (a) the reference's own fixtures (tests/ *.out-ref, NAFv2.pdf VLE table, nucleotide table),
 *   (b) golden vectors produced by the real reference binaries (oracle/_ref, built from
 *       /root/reference by oracle/Makefile) and by the image's libzstd -- tests/golden/.
 */
#ifndef NAF_ORACLE_H
#define NAF_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- zstd (RFC 8878) : zstd_spec.c ------------------------------------------------------- */
/* Decompress ONE zstd frame (with magic).  Returns bytes produced, or <0 on error.
 * *consumed (optional) receives the number of source bytes the frame occupied. */
long long nafo_zstd_decompress_frame(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                                     size_t *consumed);
/* Decompress a concatenation of frames (what one-shot ZSTD_decompress accepts). */
long long nafo_zstd_decompress(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap);
/* Content size by walking (decoding) the frame(s); <0 on error. */
long long nafo_zstd_decompressed_size(const uint8_t *src, size_t src_len);
/* Store `src` as one zstd frame of Raw blocks (valid for any conformant decoder).  Returns size. */
long long nafo_zstd_store_raw(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap);

typedef struct {
    uint32_t n_blocks, n_raw, n_rle, n_compressed;
    uint32_t lit_raw, lit_rle, lit_huf, lit_treeless;
    uint32_t seq_blocks;            /* compressed blocks with nbSeq > 0 */
    uint64_t n_sequences;
    uint32_t mode_count[3][4];      /* [LL,OF,ML][predefined,rle,fse,repeat] */
    uint32_t window_log;            /* 0 when single-segment */
    uint32_t single_segment, has_checksum, has_fcs;
    uint64_t max_offset;
} nafo_zstd_frame_info;
/* Decode the first frame and report its shape (test-side classification of golden frames). */
long long nafo_zstd_frame_info_get(const uint8_t *src, size_t src_len, nafo_zstd_frame_info *info);

/* ---- NAF transforms : naf_oracle.c -------------------------------------------------------- */
enum { NAFO_DNA = 0, NAFO_RNA = 1, NAFO_PROTEIN = 2, NAFO_TEXT = 3 };
enum { NAFO_FMT_UNKNOWN = 0, NAFO_FMT_FASTA = 1, NAFO_FMT_FASTQ = 2 };

typedef struct { uint8_t *data; size_t len, cap; } nafo_buf;

/* Result of the ennaf parse/split stage (reference: ennaf/src/process.c, encoders.c). */
typedef struct {
    int      format;                /* NAFO_FMT_* as sniffed (process.c:547-583) */
    uint64_t n_sequences;
    uint64_t longest_line;
    uint64_t n_bases;               /* seq_size_original */
    nafo_buf ids, comments, lengths, mask, seq, qual;   /* seq = packed 4-bit (DNA/RNA) or text */
    uint64_t unexpected_id[257], unexpected_comment[257], unexpected_seq[257], unexpected_qual[257];
    char     error[256];            /* die() message if the reference would have died */
} nafo_split;

/* Tolerant (default) or --well-formed parser.  Returns 0, or -1 with s->error set (reference die). */
int  nafo_split_text(const uint8_t *text, size_t len, int seq_type, int no_mask, int well_formed,
                     int forced_format, nafo_split *s);
void nafo_split_free(nafo_split *s);

/* Container writer (ennaf.c:538-589).  Each stream is stored as a Raw-block zstd frame.
 * line_length < 0 => use s->longest_line.  Returns bytes written or <0. */
long long nafo_write_naf(const nafo_split *s, int seq_type, int no_mask, long long line_length,
                         const char *title, uint8_t *dst, size_t dst_cap);

/* VLE numbers (encoders.c:175-190, unnaf utils.c:117-141). */
size_t    nafo_vle_write(uint64_t v, uint8_t out[10]);
/* returns bytes consumed, 0 on truncation, -1 on leading 0x80, -2 on overflow */
int       nafo_vle_read(const uint8_t *p, size_t len, uint64_t *v);

typedef struct {
    int      version, seq_type, flags;
    uint8_t  separator;
    uint64_t line_length, n_sequences;
    const uint8_t *title; uint64_t title_len;
    /* per section: original size, compressed size (without magic), pointer to frame[4:] */
    uint64_t orig[6], comp[6]; const uint8_t *payload[6];    /* ids,comments,lengths,mask,seq,qual */
    size_t   header_bytes;
    char     error[128];
} nafo_naf;
enum { NAFO_IDS = 0, NAFO_COMMENTS, NAFO_LENGTHS, NAFO_MASK, NAFO_SEQ, NAFO_QUAL };
int nafo_parse_naf(const uint8_t *naf, size_t len, nafo_naf *h);

/* Whole-file unnaf (unnaf.c:356-456 dispatch; output.c, output-fastq.c).
 * mode: 0 FASTA, 1 FASTQ, 2 --seq, 3 --sequences, 4 --4bit ; -1 = default (FASTQ if quality).
 * line_length_override < 0 => use stored.  Returns bytes written or <0 (h/err in errbuf). */
long long nafo_unnaf(const uint8_t *naf, size_t len, int mode, int use_mask, long long line_length_override,
                     uint8_t *dst, size_t dst_cap, char errbuf[128]);
/* Upper bound of the output size for a mode (exact for FASTA/FASTQ). */
long long nafo_unnaf_size(const uint8_t *naf, size_t len, int mode, long long line_length_override);

/* Building blocks, exported for unit parity tests. */
void   nafo_pack_4bit(const uint8_t *bases, size_t n, uint8_t *out /* (n+1)/2 */);
void   nafo_unpack_4bit(const uint8_t *packed, size_t n_bases, int rna, uint8_t *out);
size_t nafo_mask_rle(const uint8_t *bases, size_t n, uint8_t *units, size_t cap);   /* whole-stream */
void   nafo_mask_apply(uint8_t *bases, size_t n, const uint8_t *units, size_t n_units);

#ifdef __cplusplus
}
#endif
#endif
