/*
 * naf_oracle.c -- TEST INFRASTRUCTURE ONLY (oracle).  See naf_oracle.h.
 *
 * Restatement of the reference's stream transforms.  Every function cites the reference lines
 * it follows (paths relative to /root/reference).  Written as whole-buffer transforms: the
 * reference's 16 KiB / 1 MB / 128 KiB staging buffers are invisible in its output
 *  so they are not mirrored.
 */
#include "naf_oracle.h"
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <ctype.h>

#define INEOF 256

/* ---- character classes (ennaf/src/tables.c:28-145) ------------------------------------------- */
static int is_eol(unsigned c)   { return c >= 0x0A && c <= 0x0D; }                       /* tables.c:28-36 */
static int is_space(unsigned c) { return (c >= 0x09 && c <= 0x0D) || c == 0x20; }        /* tables.c:47-55 */
static int unexp_text(unsigned c)    { return c <= 0x20 || c == 0x7F || c >= 0xFF; }     /* tables.c:115-123 */
static int unexp_comment(unsigned c) { return c < 0x20 || c == 0x7F || c >= 0xFF; }      /* tables.c:126-134 */
static int unexp_qual(unsigned c)    { return c < 0x21 || c > 0x7E; }                    /* tables.c:137-145 */
static int unexp_seq(unsigned c, int seq_type, int fasta_text)
{
    if (c >= 256) return 1;
    switch (seq_type) {
    case NAFO_DNA: case NAFO_RNA: {                                                       /* tables.c:72-90 */
        if (c == '-') return 0;
        if (!((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'))) return 1;
        static const char dna[] = "ABCDGHKMNRSTVWY", rna[] = "ABCDGHKMNRSUVWY";
        return strchr(seq_type == NAFO_DNA ? dna : rna, (int)(c & ~0x20u)) == NULL;
    }
    case NAFO_PROTEIN:                                                                    /* tables.c:104-112 */
        return !((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '*' || c == '-');
    default:                                                                              /* tables.c:115-123, ennaf.c:478 */
        return unexp_text(c) || (fasta_text && c == '>');
    }
}

/* ASCII -> 4-bit code, tables.c:189-197 (inverse of unnaf.c:13 "-TGKCYSBAWRDMHVN"). */
static uint8_t nuc_code(uint8_t c)
{
    static const char tab[] = "-TGKCYSBAWRDMHVN";
    if (c == '-') return 0;
    uint8_t u = (uint8_t)(c & ~0x20);
    if (!((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'))) return 15;
    if (u == 'U') return 1;
    const char *p = strchr(tab + 1, u);
    return p ? (uint8_t)(p - tab) : 15;
}

/* ---- growable buffers ---------------------------------------------------------------------- */
static void buf_put(nafo_buf *b, const void *p, size_t n)
{
    if (b->len + n > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 4096;
        while (nc < b->len + n) nc *= 2;
        b->data = (uint8_t *)realloc(b->data, nc); b->cap = nc;
    }
    memcpy(b->data + b->len, p, n); b->len += n;
}
static void buf_putc(nafo_buf *b, uint8_t c) { buf_put(b, &c, 1); }

void nafo_split_free(nafo_split *s)
{
    free(s->ids.data); free(s->comments.data); free(s->lengths.data);
    free(s->mask.data); free(s->seq.data); free(s->qual.data);
    memset(s, 0, sizeof *s);
}

/* ---- building blocks ------------------------------------------------------------------------- */
/* encoders.c:30-69 + ennaf.c:525-529: first base in the low nibble, odd tail has high nibble 0 */
void nafo_pack_4bit(const uint8_t *bases, size_t n, uint8_t *out)
{
    for (size_t i = 0; i + 1 < n; i += 2) out[i / 2] = (uint8_t)(nuc_code(bases[i]) | (nuc_code(bases[i + 1]) << 4));
    if (n & 1) out[n / 2] = nuc_code(bases[n - 1]);
}

/* unnaf.c:13-14,369; utils.c:74-83; output.c:445-454 */
void nafo_unpack_4bit(const uint8_t *packed, size_t n_bases, int rna, uint8_t *out)
{
    char tab[17] = "-TGKCYSBAWRDMHVN";
    if (rna) tab[1] = 'U';
    for (size_t i = 0; i < n_bases; i++) {
        uint8_t b = packed[i / 2];
        out[i] = (uint8_t)tab[(i & 1) ? (b >> 4) : (b & 15)];
    }
}

/* encoders.c:126-146 extract_mask + :98-123 add_mask + ennaf.c:511 final flush */
static void mask_emit(nafo_buf *m, uint64_t len)
{
    while (len >= 255) { buf_putc(m, 255); len -= 255; }
    buf_putc(m, (uint8_t)len);
}
static void mask_rle_buf(const uint8_t *bases, size_t n, nafo_buf *m)
{
    int on = 0; uint64_t run = 0;
    for (size_t i = 0; i < n; i++) {
        int lower = bases[i] >= 96;
        if (lower != on) { mask_emit(m, run); run = 0; on = lower; }
        run++;
    }
    if (run > 0) mask_emit(m, run);
}
size_t nafo_mask_rle(const uint8_t *bases, size_t n, uint8_t *units, size_t cap)
{
    nafo_buf m = {0}; mask_rle_buf(bases, n, &m);
    size_t len = m.len;
    if (len <= cap && len) memcpy(units, m.data, len);
    free(m.data);
    return len;
}

/* input.c:236-245 (leading zero unit) + output.c:295-322 mask_dna_buffer, as one pass */
void nafo_mask_apply(uint8_t *bases, size_t n, const uint8_t *units, size_t n_units)
{
    int on = 0; size_t pos = 0;
    for (size_t k = 0; k < n_units && pos < n; k++) {
        size_t adv = units[k]; if (adv > n - pos) adv = n - pos;
        if (on) for (size_t i = 0; i < adv; i++) bases[pos + i] = (uint8_t)(bases[pos + i] + 32);
        pos += adv;
        if (units[k] != 255) on = !on;
    }
}

/* encoders.c:72-95 */
static void put_length(nafo_buf *b, uint64_t len)
{
    uint32_t u;
    while (len >= 0xFFFFFFFFull) { u = 0xFFFFFFFFu; buf_put(b, &u, 4); len -= 0xFFFFFFFFull; }
    u = (uint32_t)len; buf_put(b, &u, 4);
}

/* ---- ennaf parse/split ------------------------------------------------------------------------ */
typedef struct {
    const uint8_t *t; size_t len, pos;
    int seq_type, fasta_text;
    nafo_split *s;
    nafo_buf bases;                 /* post-replacement sequence bytes, before mask/pack */
} parser;

static unsigned getc_(parser *p)  { return p->pos < p->len ? p->t[p->pos++] : INEOF; }
static unsigned peekc_(parser *p) { return p->pos < p->len ? p->t[p->pos] : INEOF; }
static int useq(parser *p, unsigned c) { return unexp_seq(c, p->seq_type, p->fasta_text); }
static uint8_t seq_replacement(int seq_type) { return seq_type <= NAFO_RNA ? 'N' : (seq_type == NAFO_PROTEIN ? 'X' : '?'); }  /* ennaf.c:447-470 */

/* process.c:363-377 (FASTA) == :482-496 (FASTQ): ID, then comment.  Returns header terminator. */
static unsigned read_header_tolerant(parser *p)
{
    nafo_split *s = p->s; unsigned c;
    for (;;) {
        /* in text+FASTA mode the reference flips is_unexpected_text_arr['>'] itself (ennaf.c:478),
         * so '>' also terminates ID scanning there */
        while ((c = getc_(p)) != INEOF && !(unexp_text(c) || (p->fasta_text && c == '>'))) buf_putc(&s->ids, (uint8_t)c);
        if (c == INEOF || is_space(c)) break;
        s->unexpected_id[c]++; buf_putc(&p->bases, '?');                     /* R7 quirk, process.c:366 */
    }
    buf_putc(&s->ids, 0);
    if (c != INEOF && !is_eol(c)) {
        for (;;) {
            while ((c = getc_(p)) != INEOF && !unexp_comment(c)) buf_putc(&s->comments, (uint8_t)c);
            if (c == INEOF || is_eol(c)) break;
            s->unexpected_comment[c]++; buf_putc(&s->comments, '?');
        }
    }
    buf_putc(&s->comments, 0);
    return c;
}

static void bad_seq_char(parser *p, unsigned c) { p->s->unexpected_seq[c]++; buf_putc(&p->bases, seq_replacement(p->seq_type)); }

/* process.c:358-427 */
static void fasta_tolerant(parser *p)
{
    nafo_split *s = p->s; unsigned c;
    do {
        c = read_header_tolerant(p);
        uint64_t rec_start = p->bases.len;
        if (c != INEOF) {
            if (peekc_(p) == '>') p->pos++;                                   /* empty sequence, :383 */
            else {
                uint64_t line_start = rec_start;
                for (;;) {
                    while ((c = getc_(p)) != INEOF && !useq(p, c)) buf_putc(&p->bases, (uint8_t)c);
                    if (c == INEOF) break;
                    if (is_eol(c)) {
                        if (p->bases.len - line_start > s->longest_line) s->longest_line = p->bases.len - line_start;
                        line_start = p->bases.len;
                        c = getc_(p);
                        if (!useq(p, c)) { buf_putc(&p->bases, (uint8_t)c); continue; }
                        if (c == '>' || c == INEOF) break;
                        if (is_eol(c)) {
                            while (c != INEOF && is_eol(c)) c = getc_(p);
                            if (c == '>' || c == INEOF) break;
                            if (!useq(p, c)) { buf_putc(&p->bases, (uint8_t)c); continue; }
                            if (!is_space(c)) bad_seq_char(p, c);
                        }
                        else if (!is_space(c)) bad_seq_char(p, c);
                    }
                    else if (is_space(c)) {}
                    else if (c == '>' && p->seq_type == NAFO_TEXT) buf_putc(&p->bases, '>');
                    else bad_seq_char(p, c);
                }
                if (c == INEOF && p->bases.len - line_start > s->longest_line) s->longest_line = p->bases.len - line_start;
            }
        }
        put_length(&s->lengths, p->bases.len - rec_start);
        s->n_sequences++;
    } while (c != INEOF);
}

/* process.c:314-355 */
static void fasta_well_formed(parser *p)
{
    nafo_split *s = p->s; unsigned c;
    do {
        while ((c = getc_(p)) != INEOF && c != '\n' && c != ' ') buf_putc(&s->ids, (uint8_t)c);
        buf_putc(&s->ids, 0);
        if (c == ' ') while ((c = getc_(p)) != INEOF && c != '\n') buf_putc(&s->comments, (uint8_t)c);
        buf_putc(&s->comments, 0);
        uint64_t rec_start = p->bases.len;
        if (c != INEOF) {
            if (peekc_(p) == '>') p->pos++;
            else {
                uint64_t line_start = rec_start;
                for (;;) {
                    while ((c = getc_(p)) != INEOF && c != '\n') buf_putc(&p->bases, (uint8_t)c);
                    if (c == INEOF) break;
                    if (p->bases.len - line_start > s->longest_line) s->longest_line = p->bases.len - line_start;
                    line_start = p->bases.len;
                    c = getc_(p);
                    if (c == '>' || c == INEOF) break;
                    p->pos--;
                }
                if (c == INEOF && p->bases.len - line_start > s->longest_line) s->longest_line = p->bases.len - line_start;
            }
        }
        put_length(&s->lengths, p->bases.len - rec_start);
        s->n_sequences++;
    } while (c != INEOF);
}

#define DIE(...) do { snprintf(s->error, sizeof s->error, __VA_ARGS__); return -1; } while (0)

/* process.c:477-544 */
static int fastq_tolerant(parser *p)
{
    nafo_split *s = p->s; unsigned c;
    for (;;) {
        c = read_header_tolerant(p);
        if (c == INEOF) DIE("truncated FASTQ input: last sequence has no sequence data\n");
        uint64_t start = p->bases.len;
        for (;;) {
            while ((c = getc_(p)) != INEOF && !useq(p, c)) buf_putc(&p->bases, (uint8_t)c);
            if (c == INEOF || is_eol(c)) break;
            if (!is_space(c)) bad_seq_char(p, c);
        }
        uint64_t read_len = p->bases.len - start;
        if (read_len > s->longest_line) s->longest_line = read_len;
        if (c == INEOF) DIE("truncated FASTQ input: last sequence has no quality\n");
        do { c = getc_(p); } while (is_eol(c));
        if (c == INEOF) DIE("truncated FASTQ input: last sequence has no quality\n");
        if (c != '+') DIE("invalid FASTQ input: can't find '+' line of sequence %llu\n", (unsigned long long)s->n_sequences + 1);
        while ((c = getc_(p)) != INEOF && !is_eol(c)) {}
        if (c == INEOF) DIE("truncated FASTQ input: last sequence has no quality\n");
        do { c = getc_(p); } while (is_eol(c));
        if (c == INEOF) DIE("truncated FASTQ input: last sequence has no quality\n");
        uint64_t qstart = s->qual.len;
        buf_putc(&s->qual, (uint8_t)c);                                          /* :522, unconditionally */
        for (;;) {
            while ((c = getc_(p)) != INEOF && !unexp_qual(c)) buf_putc(&s->qual, (uint8_t)c);
            if (c == INEOF || is_eol(c)) break;
            if (!is_space(c)) { s->unexpected_qual[c]++; buf_putc(&s->qual, '!'); }
        }
        uint64_t qlen = s->qual.len - qstart;
        if (qlen != read_len)
            DIE("quality length of sequence %llu (%llu) doesn't match sequence length (%llu)\n",
                (unsigned long long)s->n_sequences + 1, (unsigned long long)qlen, (unsigned long long)read_len);
        put_length(&s->lengths, read_len);
        s->n_sequences++;
        do { c = getc_(p); } while (is_eol(c));
        if (c == INEOF) break;
        if (c != '@') DIE("invalid FASTQ input: Can't find '@' after sequence %llu\n", (unsigned long long)s->n_sequences);
    }
    return 0;
}

/* process.c:430-474 */
static int fastq_well_formed(parser *p)
{
    nafo_split *s = p->s; unsigned c;
    for (;;) {
        while ((c = getc_(p)) != INEOF && c != '\n' && c != ' ') buf_putc(&s->ids, (uint8_t)c);
        buf_putc(&s->ids, 0);
        if (c == ' ') while ((c = getc_(p)) != INEOF && c != '\n') buf_putc(&s->comments, (uint8_t)c);
        buf_putc(&s->comments, 0);
        if (c == INEOF) DIE("truncated FASTQ input: last sequence has no sequence data\n");
        uint64_t start = p->bases.len;
        while ((c = getc_(p)) != INEOF && c != '\n') buf_putc(&p->bases, (uint8_t)c);
        uint64_t read_len = p->bases.len - start;
        if (read_len > s->longest_line) s->longest_line = read_len;
        c = getc_(p);
        if (c != '+') { if (c == INEOF) DIE("truncated FASTQ input: last sequence has no quality\n"); DIE("not well-formed FASTQ input\n"); }
        c = getc_(p);
        if (c != '\n') DIE("not well-formed FASTQ input\n");
        uint64_t qstart = s->qual.len;
        while ((c = getc_(p)) != INEOF && c != '\n') buf_putc(&s->qual, (uint8_t)c);
        if (s->qual.len - qstart != read_len) DIE("quality length of sequence %llu doesn't match sequence length\n", (unsigned long long)s->n_sequences + 1);
        put_length(&s->lengths, read_len);
        s->n_sequences++;
        c = getc_(p);
        if (c != '@') { if (c == INEOF) break; DIE("not well-formed FASTQ input\n"); }
    }
    return 0;
}

int nafo_split_text(const uint8_t *text, size_t len, int seq_type, int no_mask, int well_formed,
                    int forced_format, nafo_split *s)
{
    memset(s, 0, sizeof *s);
    parser p; memset(&p, 0, sizeof p);
    p.t = text; p.len = len; p.seq_type = seq_type; p.s = s;

    /* confirm_input_format, process.c:547-583 */
    unsigned last = '\n', c;
    while ((c = getc_(&p)) != INEOF && is_space(c)) last = c;
    if (c == INEOF) { s->format = NAFO_FMT_UNKNOWN; return 0; }               /* empty input: process.c:589 */
    if (c == '>' && is_eol(last)) s->format = NAFO_FMT_FASTA;
    else if (c == '@' && is_eol(last)) s->format = NAFO_FMT_FASTQ;
    else if (c == '>' || c == '@') DIE("invalid input - first '%c' is not at the beginning of the line\n", (int)c);
    else DIE("input data is in unknown format - first non-space character is neither '>' nor '@'\n");
    if (forced_format != NAFO_FMT_UNKNOWN && forced_format != s->format)
        DIE("input format is different from format specified in the command line\n");
    p.fasta_text = (seq_type == NAFO_TEXT && s->format == NAFO_FMT_FASTA);       /* ennaf.c:478 */

    int rc = 0;
    if (s->format == NAFO_FMT_FASTA) { if (well_formed) fasta_well_formed(&p); else fasta_tolerant(&p); }
    else rc = well_formed ? fastq_well_formed(&p) : fastq_tolerant(&p);
    if (rc == 0) {
        s->n_bases = p.bases.len;
        if (seq_type <= NAFO_RNA) {                                                 /* process.c:595-596 */
            if (!no_mask) mask_rle_buf(p.bases.data, p.bases.len, &s->mask);
            size_t pl = (p.bases.len + 1) / 2;
            uint8_t *pk = (uint8_t *)malloc(pl ? pl : 1);
            nafo_pack_4bit(p.bases.data, p.bases.len, pk);
            buf_put(&s->seq, pk, pl); free(pk);
        } else {
            if (no_mask) for (size_t i = 0; i < p.bases.len; i++) p.bases.data[i] = (uint8_t)toupper(p.bases.data[i]);   /* process.c:46-51 */
            buf_put(&s->seq, p.bases.data, p.bases.len);
        }
    }
    free(p.bases.data);
    return rc;
}

/* ---- container ---------------------------------------------------------------------------------- */
size_t nafo_vle_write(uint64_t v, uint8_t out[10])                               /* encoders.c:175-190 */
{
    uint8_t tmp[10]; int n = 0;
    tmp[n++] = (uint8_t)(v & 127); v >>= 7;
    while (v) { tmp[n++] = (uint8_t)(128 | (v & 127)); v >>= 7; }
    for (int i = 0; i < n; i++) out[i] = tmp[n - 1 - i];
    return (size_t)n;
}

int nafo_vle_read(const uint8_t *p, size_t len, uint64_t *v)                     /* unnaf utils.c:117-141 */
{
    uint64_t a = 0; size_t i = 0;
    if (len == 0) return 0;
    uint8_t c = p[i++];
    if (c == 128) return -1;
    while (c & 128) {
        if (a & (127ull << 57)) return -2;
        a = (a << 7) | (c & 127);
        if (i >= len) return 0;
        c = p[i++];
    }
    if (a & (127ull << 57)) return -2;
    *v = (a << 7) | c;
    return (int)i;
}

static long long put_section(uint8_t *dst, size_t cap, size_t pos, uint64_t orig, const nafo_buf *b)
{
    uint8_t v[10]; size_t n = nafo_vle_write(orig, v);
    size_t fcap = b->len + 16 + 3 * (b->len / (128 * 1024) + 1);
    uint8_t *frame = (uint8_t *)malloc(fcap);
    long long fl = nafo_zstd_store_raw(b->data, b->len, frame, fcap);
    if (fl < 4) { free(frame); return -1; }
    uint8_t v2[10]; size_t n2 = nafo_vle_write((uint64_t)fl - 4, v2);
    if (pos + n + n2 + (size_t)fl - 4 > cap) { free(frame); return -1; }
    memcpy(dst + pos, v, n); pos += n;
    memcpy(dst + pos, v2, n2); pos += n2;
    memcpy(dst + pos, frame + 4, (size_t)fl - 4); pos += (size_t)fl - 4;       /* compressor.c:150-173: magic stripped */
    free(frame);
    return (long long)pos;
}

long long nafo_write_naf(const nafo_split *s, int seq_type, int no_mask, long long line_length,
                         const char *title, uint8_t *dst, size_t cap)
{                                                                                 /* ennaf.c:538-589 */
    int store_mask = !(no_mask || seq_type >= NAFO_PROTEIN);                      /* ennaf.c:445 */
    int store_qual = s->format == NAFO_FMT_FASTQ;                                 /* ennaf.c:477 */
    size_t pos = 0; uint8_t v[10]; size_t n;
    if (cap < 32) return -1;
    dst[pos++] = 0x01; dst[pos++] = 0xF9; dst[pos++] = 0xEC;
    if (seq_type == NAFO_DNA) dst[pos++] = 1; else { dst[pos++] = 2; dst[pos++] = (uint8_t)seq_type; }
    dst[pos++] = (uint8_t)(((title != NULL) << 6) | (1 << 5) | (1 << 4) | (1 << 3) | (store_mask << 2) | (1 << 1) | store_qual);
    dst[pos++] = ' ';
    n = nafo_vle_write(line_length >= 0 ? (uint64_t)line_length : s->longest_line, v); memcpy(dst + pos, v, n); pos += n;
    n = nafo_vle_write(s->n_sequences, v); memcpy(dst + pos, v, n); pos += n;
    if (title) {
        size_t tl = strlen(title);
        n = nafo_vle_write(tl, v); memcpy(dst + pos, v, n); pos += n;
        if (pos + tl > cap) return -1;
        memcpy(dst + pos, title, tl); pos += tl;
    }
    long long r;
    if ((r = put_section(dst, cap, pos, s->ids.len, &s->ids)) < 0) return r; pos = (size_t)r;
    if ((r = put_section(dst, cap, pos, s->comments.len, &s->comments)) < 0) return r; pos = (size_t)r;
    if ((r = put_section(dst, cap, pos, s->lengths.len, &s->lengths)) < 0) return r; pos = (size_t)r;
    if (store_mask) { if ((r = put_section(dst, cap, pos, s->mask.len, &s->mask)) < 0) return r; pos = (size_t)r; }
    if ((r = put_section(dst, cap, pos, s->n_bases, &s->seq)) < 0) return r; pos = (size_t)r;   /* ennaf.c:582: bases, not bytes */
    if (store_qual) { if ((r = put_section(dst, cap, pos, s->qual.len, &s->qual)) < 0) return r; pos = (size_t)r; }
    return (long long)pos;
}

int nafo_parse_naf(const uint8_t *naf, size_t len, nafo_naf *h)
{                                                                                 /* input.c:31-77, unnaf.c:402-404 */
    memset(h, 0, sizeof *h);
#define HFAIL(msg) do { snprintf(h->error, sizeof h->error, "%s", msg); return -1; } while (0)
    if (len == 0) HFAIL("empty input");
    if (len < 3) HFAIL("incomplete or truncated input\n");
    if (naf[0] != 0x01 || naf[1] != 0xF9 || naf[2] != 0xEC) HFAIL("not a NAF format\n");
    size_t pos = 3;
    if (pos >= len) HFAIL("incomplete or truncated input\n");
    h->version = naf[pos++];
    if (h->version < 1 || h->version > 2) { snprintf(h->error, sizeof h->error, "unknown version (%d) of NAF format\n", h->version); return -1; }
    h->seq_type = NAFO_DNA;
    if (h->version > 1) {
        if (pos >= len) HFAIL("incomplete or truncated input\n");
        int t = naf[pos++];
        if (t < 1 || t > 3) { snprintf(h->error, sizeof h->error, "unknown sequence type (%d) found in NAF file\n", t); return -1; }
        h->seq_type = t;
    }
    if (pos + 2 > len) HFAIL("incomplete or truncated input\n");
    h->flags = naf[pos++];
    h->separator = naf[pos++];
    if (h->separator < 0x20 || h->separator > 0x7E) HFAIL("unsupported name separator character\n");
#define RDNUM(dst) do { int k = nafo_vle_read(naf + pos, len - pos, &(dst)); \
        if (k == 0) HFAIL("incomplete or truncated input\n"); \
        if (k == -1) HFAIL("invalid input: error parsing variable length encoded number\n"); \
        if (k == -2) HFAIL("invalid input: overflow reading a variable length encoded number\n"); pos += (size_t)k; } while (0)
    RDNUM(h->line_length); RDNUM(h->n_sequences);
    if (h->flags & 0x40) { RDNUM(h->title_len); if (pos + h->title_len > len) HFAIL("incomplete or truncated input\n"); h->title = naf + pos; pos += h->title_len; }
    h->header_bytes = pos;
    static const int bit[6] = { 0x20, 0x10, 0x08, 0x04, 0x02, 0x01 };
    for (int i = 0; i < 6; i++) {
        if (!(h->flags & bit[i])) continue;
        RDNUM(h->orig[i]); RDNUM(h->comp[i]);
        if (pos + h->comp[i] > len) HFAIL("incomplete or truncated input\n");
        h->payload[i] = naf + pos; pos += h->comp[i];
    }
    return 0;
}

/* ---- unnaf ------------------------------------------------------------------------------------- */
static long long load_section(const nafo_naf *h, int i, uint8_t **out, uint64_t expect_len)
{                                                                                 /* input.c:145-246: magic re-prefixed */
    size_t cl = (size_t)h->comp[i] + 4;
    uint8_t *c = (uint8_t *)malloc(cl);
    c[0] = 0x28; c[1] = 0xB5; c[2] = 0x2F; c[3] = 0xFD; memcpy(c + 4, h->payload[i], (size_t)h->comp[i]);
    uint8_t *o = (uint8_t *)malloc(expect_len ? expect_len : 1);
    long long n = nafo_zstd_decompress(c, cl, o, expect_len);
    free(c);
    if (n < 0 || (uint64_t)n != expect_len) { free(o); return -1; }
    *out = o;
    return n;
}

typedef struct { uint8_t *d; size_t pos, cap; int overflow; } sink;
static void sk_put(sink *k, const void *p, size_t n) { if (k->pos + n > k->cap) { k->overflow = 1; k->pos += n; return; } if (!k->overflow) memcpy(k->d + k->pos, p, n); k->pos += n; }
static void sk_putc(sink *k, uint8_t c) { sk_put(k, &c, 1); }

typedef struct { const char **id, **nm; int has_ids, has_names; uint8_t sep; } namer;
static void put_name(sink *k, const namer *nm, uint64_t i)                        /* output.c:105-124 */
{
    if (nm->has_ids) sk_put(k, nm->id[i], strlen(nm->id[i]));
    if (nm->has_names) {
        if (!nm->has_ids) sk_put(k, nm->nm[i], strlen(nm->nm[i]));
        else if (nm->nm[i][0]) { sk_putc(k, nm->sep); sk_put(k, nm->nm[i], strlen(nm->nm[i])); }
    }
}

static int index_strings(const uint8_t *buf, uint64_t size, uint64_t n, const char ***out)
{                                                                                 /* input.c:157-168 */
    if (size == 0 || buf[size - 1] != 0) return -1;
    const char **v = (const char **)malloc(sizeof(char *) * (n ? n : 1));
    const char *p = (const char *)buf, *end = (const char *)buf + size;
    for (uint64_t i = 0; i < n; i++) {
        if (p >= end) { free(v); return -1; }
        v[i] = p; p += strlen(p) + 1;
    }
    *out = v; return 0;
}

static void put_wrapped(sink *k, const uint8_t *b, uint64_t n, uint64_t L, uint64_t *line_rem)
{                                                                                 /* output.c:339-360 */
    if (L == 0) { sk_put(k, b, n); return; }
    while (n > *line_rem) { sk_put(k, b, *line_rem); sk_putc(k, '\n'); b += *line_rem; n -= *line_rem; *line_rem = L; }
    sk_put(k, b, n); *line_rem -= n;
}

long long nafo_unnaf(const uint8_t *naf, size_t len, int mode, int use_mask, long long line_override,
                     uint8_t *dst, size_t dst_cap, char errbuf[128])
{
    nafo_naf h; errbuf[0] = 0;
#define UFAIL(msg) do { snprintf(errbuf, 128, "%s", msg); rc = -1; goto done; } while (0)
    if (nafo_parse_naf(naf, len, &h) < 0) { snprintf(errbuf, 128, "%s", h.error); return -1; }
    int has_ids = (h.flags >> 5) & 1, has_names = (h.flags >> 4) & 1, has_len = (h.flags >> 3) & 1,
        has_mask = (h.flags >> 2) & 1, has_data = (h.flags >> 1) & 1, has_qual = h.flags & 1;
    if (mode < 0) mode = has_qual ? 1 : 0;                                        /* unnaf.c:372-375 */
    uint64_t L = line_override >= 0 ? (uint64_t)line_override : h.line_length, N = h.n_sequences;
    sink k = { dst, 0, dst_cap, 0 };
    uint8_t *ids = NULL, *names = NULL, *lens = NULL, *mask = NULL, *seq = NULL, *qual = NULL, *bases = NULL;
    const char **idv = NULL, **nmv = NULL; long long rc = 0;
    if (N == 0 || !has_data) return 0;                                            /* unnaf.c:409, output.c:610 */
    if (mode == 1 && !has_qual) { snprintf(errbuf, 128, "FASTQ output requested, but input has no qualities\n"); return -1; }
    if (mode == 4 && h.seq_type >= NAFO_PROTEIN) { snprintf(errbuf, 128, "input has no 4-bit encoded data, but %s sequences\n", h.seq_type == NAFO_PROTEIN ? "protein" : "text"); return -1; }
    int fourbit = h.seq_type <= NAFO_RNA;
    uint64_t T = h.orig[NAFO_SEQ];
    uint64_t seq_bytes = fourbit ? (T + 1) / 2 : T;
    if (load_section(&h, NAFO_SEQ, &seq, seq_bytes) < 0) UFAIL("can't decompress sequence\n");
    if (mode == 4) { sk_put(&k, seq, seq_bytes); goto done; }                     /* output.c:266-292 */
    if (mode == 0 || mode == 1) {
        if (has_ids) { if (load_section(&h, NAFO_IDS, &ids, h.orig[NAFO_IDS]) < 0) UFAIL("can't decompress ids\n"); if (index_strings(ids, h.orig[NAFO_IDS], N, &idv) < 0) UFAIL("corrupted ids - not 0-terminated\n"); }
        if (has_names) { if (load_section(&h, NAFO_COMMENTS, &names, h.orig[NAFO_COMMENTS]) < 0) UFAIL("can't decompress names\n"); if (index_strings(names, h.orig[NAFO_COMMENTS], N, &nmv) < 0) UFAIL("corrupted names - not 0-terminated\n"); }
    }
    uint64_t n_len = 0;
    if (has_len && mode != 2) { if (load_section(&h, NAFO_LENGTHS, &lens, h.orig[NAFO_LENGTHS]) < 0) UFAIL("can't decompress lengths\n"); n_len = h.orig[NAFO_LENGTHS] / 4; }
    const uint32_t *lu = (const uint32_t *)lens;
    int masking = use_mask && has_mask && mode != 1;                              /* unnaf.c:442: FASTQ never masks */
    if (masking) { if (load_section(&h, NAFO_MASK, &mask, h.orig[NAFO_MASK]) < 0) UFAIL("can't decompress mask\n"); if (h.orig[NAFO_MASK] > 0 && mask[0] == 0 && h.orig[NAFO_MASK] < 2) UFAIL("corrupted mask\n"); }
    bases = (uint8_t *)malloc(T ? T : 1);
    if (fourbit) nafo_unpack_4bit(seq, T, h.seq_type == NAFO_RNA, bases);
    else { memcpy(bases, seq, T); if (!use_mask) for (uint64_t i = 0; i < T; i++) bases[i] = (uint8_t)toupper(bases[i]); }   /* output.c:363-366 */
    if (masking) nafo_mask_apply(bases, T, mask, h.orig[NAFO_MASK]);
    namer nm = { idv, nmv, has_ids, has_names, h.separator };

    if (mode == 2) { sk_put(&k, bases, T); }                                      /* output.c:457-512 */
    else if (mode == 3) {                                                         /* output-sequences.c:7-116 */
        if (T > 0 && n_len > 0) {
            uint64_t li = 0, pos = 0, rem = T, cur = lu[0];
            while (rem >= cur) {
                if (cur) { sk_put(&k, bases + pos, cur); pos += cur; rem -= cur; }
                if (lu[li] != 0xFFFFFFFFu) sk_putc(&k, '\n');
                if (++li >= n_len) break;
                cur = lu[li];
            }
            if (rem > 0 && li >= n_len) sk_put(&k, bases + pos, rem); else if (rem > 0) sk_put(&k, bases + pos, rem);
        }
    }
    else if (mode == 0) {                                                         /* output.c:369-430,608-674 */
        uint64_t li = 0, ri = 0, pos = 0, rem = T, line_rem = L;
        while (li < n_len && ri < N && lu[li] == 0) { sk_putc(&k, '>'); put_name(&k, &nm, ri); sk_putc(&k, '\n'); li++; ri++; }
        if (ri < N && li < n_len) {
            sk_putc(&k, '>'); put_name(&k, &nm, ri); sk_putc(&k, '\n');
            uint64_t cur = lu[li];
            if (T > 0) {
                while (rem >= cur) {
                    if (cur) { put_wrapped(&k, bases + pos, cur, L, &line_rem); pos += cur; rem -= cur; }
                    if (lu[li] == 0xFFFFFFFFu) li++;
                    else {
                        sk_putc(&k, '\n'); li++; ri++;
                        while (li < n_len && ri < N && lu[li] == 0) { sk_putc(&k, '>'); put_name(&k, &nm, ri); sk_putc(&k, '\n'); li++; ri++; }
                        if (ri < N) { sk_putc(&k, '>'); put_name(&k, &nm, ri); sk_putc(&k, '\n'); line_rem = L; }
                    }
                    if (li >= n_len) break;
                    cur = lu[li];
                }
                if (rem > 0) put_wrapped(&k, bases + pos, rem, L, &line_rem);
            }
        }
    }
    else if (mode == 1) {                                                         /* output-fastq.c:100-149 */
        if (load_section(&h, NAFO_QUAL, &qual, h.orig[NAFO_QUAL]) < 0) UFAIL("can't decompress quality\n");
        uint64_t li = 0, qi = 0, pos = 0, qpos = 0;
        for (uint64_t ri = 0; ri < N; ri++) {
            sk_putc(&k, '@'); put_name(&k, &nm, ri); sk_putc(&k, '\n');
            while (li < n_len && lu[li] == 0xFFFFFFFFu) { sk_put(&k, bases + pos, lu[li]); pos += lu[li]; li++; }
            if (li < n_len) { sk_put(&k, bases + pos, lu[li]); pos += lu[li]; li++; }
            sk_put(&k, "\n+\n", 3);
            while (qi < n_len && lu[qi] == 0xFFFFFFFFu) { sk_put(&k, qual + qpos, lu[qi]); qpos += lu[qi]; qi++; }
            if (qi < n_len) { sk_put(&k, qual + qpos, lu[qi]); qpos += lu[qi]; qi++; }
            sk_putc(&k, '\n');
        }
    }
done:
    free(ids); free(names); free(lens); free(mask); free(seq); free(qual); free(bases); free((void *)idv); free((void *)nmv);
    if (rc < 0) return rc;
    if (k.overflow) return -2 - (long long)0;       /* destination too small; size is in nafo_unnaf_size */
    return (long long)k.pos;
}

long long nafo_unnaf_size(const uint8_t *naf, size_t len, int mode, long long line_override)
{
    char err[128]; uint8_t dummy;
    nafo_naf h;
    if (nafo_parse_naf(naf, len, &h) < 0) return -1;
    /* run the emitter against a zero-capacity sink: it only counts */
    sink probe = { &dummy, 0, 0, 0 }; (void)probe;
    /* simple approach: allocate generously and measure */
    uint64_t T = h.orig[NAFO_SEQ];
    size_t cap = (size_t)(T * 2 + h.orig[NAFO_IDS] + h.orig[NAFO_COMMENTS] + h.orig[NAFO_QUAL] + 8 * h.n_sequences + 64);
    if (line_override > 0) cap += (size_t)(T / (uint64_t)line_override) + 1; else if (line_override == 0) {}
    uint8_t *tmp = (uint8_t *)malloc(cap);
    if (!tmp) return -1;
    long long n = nafo_unnaf(naf, len, mode, 1, line_override, tmp, cap, err);
    free(tmp);
    return n;
}
