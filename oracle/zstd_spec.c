/*
 * zstd_spec.c -- TEST INFRASTRUCTURE ONLY (oracle).  See naf_oracle.h.
 *
 * A from-the-specification zstd frame decoder (RFC 8878 "Zstandard Compression and the
 * 'application/zstd' Media Type", sections 3.1.1 frames, 3.1.1.2 blocks, 3.1.1.3 literals and
 * sequences, 4.1 FSE, 4.2 Huffman).  The reference reaches this arithmetic through libzstd, which is
 * NOT in /root/reference (zstd/ is an empty submodule; pin "v1.5.0" only in CHANGELOG.md:10).  Call
 * sites this stands in for: unnaf/src/input.c:155,183,212,230 (one-shot ZSTD_decompress) and
 * input.c:262-285,368,399,426 / output.c:646 (ZSTD_decompressStream).  Dictionaries are not
 * supported (the reference never uses one).  Pinned against the image's libzstd 1.4.9 and the
 * golden frames in tests/golden/ by tests/test_oracle_zstd.py.
 */
#include "naf_oracle.h"
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
/* test aid: NAF_ORACLE_SEQ_DUMP=<path> appends "match position, literal length, match length, offset" of every sequence decoded */
static FILE *seq_dump = NULL;
static void seq_dump_open(void) { static int tried = 0; if (!tried) { tried = 1; const char *pth = getenv("NAF_ORACLE_SEQ_DUMP"); if (pth) seq_dump = fopen(pth, "a"); } }


#define ERR_SRC   (-1)   /* truncated / malformed source */
#define ERR_DST   (-2)   /* destination too small */
#define ERR_CORR  (-3)   /* corrupted bit-stream */
#define ERR_UNSUP (-4)   /* dictionary, reserved fields */

#define MAX_BLOCK (128 * 1024)
#define HUF_MAX_LOG 11

typedef struct { uint8_t sym; uint8_t nbits; } huf_entry;
typedef struct { uint8_t sym; uint8_t nbits; uint16_t base; } fse_entry;
typedef struct { fse_entry e[512]; int log; } fse_table;

typedef struct {
    huf_entry huf[1 << HUF_MAX_LOG]; int huf_log; int huf_valid;
    fse_table ll, of, ml; int ll_valid, of_valid, ml_valid;
    uint64_t rep[3];
    nafo_zstd_frame_info *info;
} frame_ctx;

/* ---- forward bit reader (FSE table descriptions) ---------------------------------------- */
typedef struct { const uint8_t *p; size_t len; size_t bitpos; } fwd_bits;
static uint32_t fwd_peek(const fwd_bits *b, int n)
{
    uint64_t v = 0; size_t byte = b->bitpos >> 3;
    for (int i = 0; i < 5; i++) if (byte + (size_t)i < b->len) v |= (uint64_t)b->p[byte + (size_t)i] << (8 * i);
    return (uint32_t)((v >> (b->bitpos & 7)) & ((1ull << n) - 1));
}

/* ---- backward bit reader (Huffman + FSE payload streams, RFC 8878 4.1 / 4.2.2) ----------- */
typedef struct { const uint8_t *p; long long bits; } bwd_bits;   /* bits = number of unread bits */
static int bwd_init(bwd_bits *b, const uint8_t *p, size_t len)
{
    if (len == 0) return ERR_CORR;
    uint8_t last = p[len - 1];
    if (last == 0) return ERR_CORR;
    int hb = 7; while (!((last >> hb) & 1)) hb--;
    b->p = p; b->bits = (long long)(len - 1) * 8 + hb;           /* skip the final-bit flag */
    return 0;
}
/* Read n bits (n <= 57); bits below the start of the stream read as zero and drive bits negative. */
static uint64_t bwd_read(bwd_bits *b, int n)
{
    long long hi = b->bits, lo = hi - n;
    b->bits = lo;
    if (n == 0 || hi <= 0) return 0;
    long long from = lo < 0 ? 0 : lo;
    size_t b0 = (size_t)(from >> 3), b1 = (size_t)((hi + 7) >> 3);
    int sh = (int)(from & 7), need = (int)(hi - from);
    uint64_t acc = 0;
    for (size_t i = b0; i < b1 && i < b0 + 8; i++) acc |= (uint64_t)b->p[i] << (8 * (i - b0));
    uint64_t v = (acc >> sh) & ((1ull << need) - 1);
    if (lo < 0) v <<= (int)(-lo);
    return v;
}
static uint64_t bwd_peek(const bwd_bits *b, int n) { bwd_bits t = *b; return bwd_read(&t, n); }

/* ---- FSE (RFC 8878 4.1) -------------------------------------------------------------------- */
static int highbit(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; }

static int fse_build(fse_table *t, const int16_t *norm, int nsym, int log)
{
    int size = 1 << log, high = size - 1;
    uint16_t next[256];
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == -1) { t->e[high--].sym = (uint8_t)s; next[s] = 1; }
        else next[s] = (uint16_t)norm[s];
    }
    int step = (size >> 1) + (size >> 3) + 3, mask = size - 1, pos = 0;
    for (int s = 0; s < nsym; s++) {
        for (int i = 0; i < norm[s]; i++) {
            t->e[pos].sym = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    }
    if (pos != 0) return ERR_CORR;
    for (int u = 0; u < size; u++) {
        uint16_t ns = next[t->e[u].sym]++;
        int nb = log - highbit(ns);
        t->e[u].nbits = (uint8_t)nb;
        t->e[u].base = (uint16_t)((ns << nb) - size);
    }
    t->log = log;
    return 0;
}

/* Parse an FSE table description; returns bytes consumed or <0. */
static int fse_read_desc(const uint8_t *p, size_t len, int max_log, int max_sym, int16_t *norm, int *nsym, int *logp)
{
    fwd_bits b = { p, len, 0 };
    if (len < 1) return ERR_SRC;
    int log = (int)fwd_peek(&b, 4) + 5; b.bitpos += 4;
    if (log > max_log) return ERR_CORR;
    int remaining = (1 << log) + 1, threshold = 1 << log, nbits = log + 1, s = 0;
    while (remaining > 1 && s <= max_sym) {
        int max = (2 * threshold - 1) - remaining;
        uint32_t v = fwd_peek(&b, nbits);
        int count;
        if ((int)(v & (uint32_t)(threshold - 1)) < max) { count = (int)(v & (uint32_t)(threshold - 1)); b.bitpos += (size_t)(nbits - 1); }
        else { count = (int)(v & (uint32_t)(2 * threshold - 1)); if (count >= threshold) count -= max; b.bitpos += (size_t)nbits; }
        count--;
        remaining -= count < 0 ? -count : count;
        norm[s++] = (int16_t)count;
        if (count == 0) {
            for (;;) {
                uint32_t r = fwd_peek(&b, 2); b.bitpos += 2;
                for (uint32_t i = 0; i < r && s <= max_sym; i++) norm[s++] = 0;
                if (r != 3) break;
            }
        }
        while (remaining < threshold && threshold > 1) { nbits--; threshold >>= 1; }
        if ((b.bitpos + 7) / 8 > len) return ERR_SRC;
    }
    if (remaining != 1) return ERR_CORR;
    *nsym = s; *logp = log;
    return (int)((b.bitpos + 7) / 8);
}

static const int16_t LL_DEFAULT[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
static const int16_t ML_DEFAULT[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
static const int16_t OF_DEFAULT[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };
static const uint32_t LL_BASE[36] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536 };
static const uint8_t  LL_BITS[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
static const uint32_t ML_BASE[53] = { 3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539 };
static const uint8_t  ML_BITS[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };

/* ---- Huffman (RFC 8878 4.2) ---------------------------------------------------------------- */
static int huf_build(frame_ctx *c, const uint8_t *weights, int n /* explicit weights */)
{
    uint32_t sum = 0;
    for (int i = 0; i < n; i++) { if (weights[i] > HUF_MAX_LOG) return ERR_CORR; if (weights[i]) sum += 1u << (weights[i] - 1); }
    if (sum == 0) return ERR_CORR;
    int log = highbit(sum) + 1;
    if (log > HUF_MAX_LOG) return ERR_CORR;
    uint32_t rest = (1u << log) - sum;
    if (rest & (rest - 1)) return ERR_CORR;                 /* last weight must be a power of two */
    uint8_t w[256]; memcpy(w, weights, (size_t)n); w[n] = (uint8_t)(highbit(rest) + 1); n++;
    uint32_t rank_start[HUF_MAX_LOG + 2] = {0}, cnt[HUF_MAX_LOG + 2] = {0};
    for (int i = 0; i < n; i++) cnt[w[i]]++;
    uint32_t pos = 0;
    for (int r = 1; r <= log; r++) { rank_start[r] = pos; pos += cnt[r] << (r - 1); }
    for (int i = 0; i < n; i++) {
        if (!w[i]) continue;
        uint32_t len = 1u << (w[i] - 1), st = rank_start[w[i]];
        for (uint32_t k = 0; k < len; k++) { c->huf[st + k].sym = (uint8_t)i; c->huf[st + k].nbits = (uint8_t)(log + 1 - w[i]); }
        rank_start[w[i]] += len;
    }
    c->huf_log = log; c->huf_valid = 1;
    return 0;
}

/* Returns bytes consumed by the tree description or <0. */
static int huf_read_tree(frame_ctx *c, const uint8_t *p, size_t len)
{
    if (len < 1) return ERR_SRC;
    uint8_t weights[256]; int n; int used;
    uint8_t hb = p[0];
    if (hb >= 128) {
        n = hb - 127; used = 1 + (n + 1) / 2;
        if ((size_t)used > len) return ERR_SRC;
        for (int i = 0; i < n; i++) weights[i] = (i & 1) ? (p[1 + i / 2] & 15) : (p[1 + i / 2] >> 4);
    } else {
        used = 1 + hb;
        if ((size_t)used > len || hb == 0) return ERR_SRC;
        int16_t norm[256]; int nsym, log;
        int d = fse_read_desc(p + 1, hb, 6, 255, norm, &nsym, &log);
        if (d < 0) return d;
        fse_table t; int r = fse_build(&t, norm, nsym, log); if (r < 0) return r;
        bwd_bits b; r = bwd_init(&b, p + 1 + d, (size_t)(hb - d)); if (r < 0) return r;
        uint32_t s1 = (uint32_t)bwd_read(&b, log), s2 = (uint32_t)bwd_read(&b, log);
        n = 0;
        for (;;) {                                   /* two interleaved states, RFC 8878 4.2.1.2 */
            if (n >= 255) return ERR_CORR;
            weights[n++] = t.e[s1].sym;
            s1 = t.e[s1].base + (uint32_t)bwd_read(&b, t.e[s1].nbits);
            if (b.bits < 0) { if (n >= 255) return ERR_CORR; weights[n++] = t.e[s2].sym; break; }
            if (n >= 255) return ERR_CORR;
            weights[n++] = t.e[s2].sym;
            s2 = t.e[s2].base + (uint32_t)bwd_read(&b, t.e[s2].nbits);
            if (b.bits < 0) { if (n >= 255) return ERR_CORR; weights[n++] = t.e[s1].sym; break; }
        }
    }
    int r = huf_build(c, weights, n);
    return r < 0 ? r : used;
}

static int huf_decode_stream(const frame_ctx *c, const uint8_t *p, size_t len, uint8_t *out, size_t n)
{
    bwd_bits b; int r = bwd_init(&b, p, len); if (r < 0) return r;
    int log = c->huf_log;
    for (size_t i = 0; i < n; i++) {
        uint32_t idx = (uint32_t)bwd_peek(&b, log);
        huf_entry e = c->huf[idx];
        out[i] = e.sym; b.bits -= e.nbits;
    }
    return b.bits == 0 ? 0 : ERR_CORR;
}

/* ---- block decoding ------------------------------------------------------------------------ */
static int read_seq_table(frame_ctx *c, int mode, fse_table *t, int *valid, const uint8_t **pp, const uint8_t *end,
                          const int16_t *def, int def_n, int def_log, int max_log, int max_sym)
{
    if (mode == 0) { *valid = 1; return fse_build(t, def, def_n, def_log); }
    if (mode == 1) {
        if (*pp >= end) return ERR_SRC;
        if (**pp > max_sym) return ERR_CORR;
        t->log = 0; t->e[0].sym = **pp; t->e[0].nbits = 0; t->e[0].base = 0; (*pp)++; *valid = 1; return 0;
    }
    if (mode == 2) {
        int16_t norm[64]; int nsym, log;
        int d = fse_read_desc(*pp, (size_t)(end - *pp), max_log, max_sym, norm, &nsym, &log);
        if (d < 0) return d;
        *pp += d; *valid = 1;
        return fse_build(t, norm, nsym, log);
    }
    (void)c;
    return *valid ? 0 : ERR_CORR;                  /* repeat mode needs a previous table */
}

static long long decode_compressed_block(frame_ctx *c, const uint8_t *src, size_t len,
                                         uint8_t *dst_base, size_t dst_pos, size_t dst_cap, size_t frame_start)
{
    static uint8_t litbuf[MAX_BLOCK + 32];
    const uint8_t *p = src, *end = src + len;
    if (len < 1) return ERR_SRC;
    int ltype = p[0] & 3, sf = (p[0] >> 2) & 3;
    size_t regen, comp = 0; int nstreams = 1; size_t hdr;
    const uint8_t *lits;
    if (ltype < 2) {
        if (!(sf & 1)) { regen = p[0] >> 3; hdr = 1; }
        else if (sf == 1) { if (len < 2) return ERR_SRC; regen = (p[0] >> 4) + ((size_t)p[1] << 4); hdr = 2; }
        else { if (len < 3) return ERR_SRC; regen = (p[0] >> 4) + ((size_t)p[1] << 4) + ((size_t)p[2] << 12); hdr = 3; }
        if (regen > MAX_BLOCK) return ERR_CORR;
        if (ltype == 0) {
            if (hdr + regen > len) return ERR_SRC;
            lits = p + hdr; p += hdr + regen; if (c->info) c->info->lit_raw++;
        } else {
            if (hdr + 1 > len) return ERR_SRC;
            memset(litbuf, p[hdr], regen); lits = litbuf; p += hdr + 1; if (c->info) c->info->lit_rle++;
        }
    } else {
        uint64_t h;
        if (sf == 0 || sf == 1) { if (len < 3) return ERR_SRC; h = p[0] | (p[1] << 8) | ((uint64_t)p[2] << 16); regen = (h >> 4) & 0x3FF; comp = (h >> 14) & 0x3FF; hdr = 3; nstreams = sf == 0 ? 1 : 4; }
        else if (sf == 2) { if (len < 4) return ERR_SRC; h = p[0] | (p[1] << 8) | ((uint64_t)p[2] << 16) | ((uint64_t)p[3] << 24); regen = (h >> 4) & 0x3FFF; comp = (h >> 18) & 0x3FFF; hdr = 4; nstreams = 4; }
        else { if (len < 5) return ERR_SRC; h = p[0] | (p[1] << 8) | ((uint64_t)p[2] << 16) | ((uint64_t)p[3] << 24) | ((uint64_t)p[4] << 32); regen = (h >> 4) & 0x3FFFF; comp = (h >> 22) & 0x3FFFF; hdr = 5; nstreams = 4; }
        if (regen > MAX_BLOCK) return ERR_CORR;
        if (hdr + comp > len) return ERR_SRC;
        const uint8_t *q = p + hdr, *qend = q + comp;
        if (ltype == 2) {
            int used = huf_read_tree(c, q, comp); if (used < 0) return used;
            q += used; if (c->info) c->info->lit_huf++;
        } else { if (!c->huf_valid) return ERR_CORR; if (c->info) c->info->lit_treeless++; }
        if (nstreams == 1) {
            int r = huf_decode_stream(c, q, (size_t)(qend - q), litbuf, regen); if (r < 0) return r;
        } else {
            if (qend - q < 6) return ERR_SRC;
            size_t s1 = q[0] | (q[1] << 8), s2 = q[2] | (q[3] << 8), s3 = q[4] | (q[5] << 8);
            q += 6;
            size_t tot = (size_t)(qend - q);
            if (s1 + s2 + s3 > tot) return ERR_CORR;
            size_t s4 = tot - s1 - s2 - s3;
            size_t per = (regen + 3) / 4;
            if (per * 3 > regen) return ERR_CORR;
            int r;
            if ((r = huf_decode_stream(c, q, s1, litbuf, per)) < 0) return r;
            if ((r = huf_decode_stream(c, q + s1, s2, litbuf + per, per)) < 0) return r;
            if ((r = huf_decode_stream(c, q + s1 + s2, s3, litbuf + 2 * per, per)) < 0) return r;
            if ((r = huf_decode_stream(c, q + s1 + s2 + s3, s4, litbuf + 3 * per, regen - 3 * per)) < 0) return r;
        }
        lits = litbuf; p += hdr + comp;
    }

    /* sequences section */
    if (p >= end) return ERR_SRC;
    uint32_t nseq;
    if (p[0] == 0) { nseq = 0; p += 1; }
    else if (p[0] < 128) { nseq = p[0]; p += 1; }
    else if (p[0] < 255) { if (end - p < 2) return ERR_SRC; nseq = ((uint32_t)(p[0] - 128) << 8) + p[1]; p += 2; }
    else { if (end - p < 3) return ERR_SRC; nseq = p[1] + ((uint32_t)p[2] << 8) + 0x7F00; p += 3; }

    uint8_t *out = dst_base + dst_pos; size_t room = dst_cap - dst_pos; size_t op = 0, lp = 0;
    if (nseq == 0) {
        if (p != end) return ERR_CORR;
        if (regen > room) return ERR_DST;
        memcpy(out, lits, regen);
        return (long long)regen;
    }
    if (p >= end) return ERR_SRC;
    uint8_t modes = *p++;
    if (modes & 3) return ERR_CORR;
    int llm = modes >> 6, ofm = (modes >> 4) & 3, mlm = (modes >> 2) & 3, r;
    if (c->info) { c->info->seq_blocks++; c->info->n_sequences += nseq; c->info->mode_count[0][llm]++; c->info->mode_count[1][ofm]++; c->info->mode_count[2][mlm]++; }
    if ((r = read_seq_table(c, llm, &c->ll, &c->ll_valid, &p, end, LL_DEFAULT, 36, 6, 9, 35)) < 0) return r;
    if ((r = read_seq_table(c, ofm, &c->of, &c->of_valid, &p, end, OF_DEFAULT, 29, 5, 8, 31)) < 0) return r;
    if ((r = read_seq_table(c, mlm, &c->ml, &c->ml_valid, &p, end, ML_DEFAULT, 53, 6, 9, 52)) < 0) return r;
    bwd_bits b; if ((r = bwd_init(&b, p, (size_t)(end - p))) < 0) return r;
    uint32_t sl = (uint32_t)bwd_read(&b, c->ll.log), so = (uint32_t)bwd_read(&b, c->of.log), sm = (uint32_t)bwd_read(&b, c->ml.log);
    if (b.bits < 0) return ERR_CORR;
    for (uint32_t i = 0; i < nseq; i++) {
        int ofc = c->of.e[so].sym, mlc = c->ml.e[sm].sym, llc = c->ll.e[sl].sym;
        if (ofc > 31 || mlc > 52 || llc > 35) return ERR_CORR;
        uint64_t ofv = (1ull << ofc) + bwd_read(&b, ofc);
        uint64_t ml = ML_BASE[mlc] + bwd_read(&b, ML_BITS[mlc]);
        uint64_t ll = LL_BASE[llc] + bwd_read(&b, LL_BITS[llc]);
        if (b.bits < 0) return ERR_CORR;
        uint64_t off;
        if (ofv > 3) { off = ofv - 3; c->rep[2] = c->rep[1]; c->rep[1] = c->rep[0]; c->rep[0] = off; }
        else {
            uint64_t idx = ofv - 1 + (ll == 0);
            if (idx == 0) off = c->rep[0];
            else {
                off = idx == 3 ? c->rep[0] - 1 : c->rep[idx];
                if (off == 0) off = 1;              /* libzstd: "0 is not valid; force offset to 1" */
                if (idx > 1) c->rep[2] = c->rep[1];
                c->rep[1] = c->rep[0]; c->rep[0] = off;
            }
        }
        if (i + 1 < nseq) {
            sl = c->ll.e[sl].base + (uint32_t)bwd_read(&b, c->ll.e[sl].nbits);
            sm = c->ml.e[sm].base + (uint32_t)bwd_read(&b, c->ml.e[sm].nbits);
            so = c->of.e[so].base + (uint32_t)bwd_read(&b, c->of.e[so].nbits);
            if (b.bits < 0) return ERR_CORR;
        }
        if (lp + ll > regen) return ERR_CORR;
        if (op + ll + ml > room) return ERR_DST;
        if (op + ll + ml > MAX_BLOCK) return ERR_CORR;
        memcpy(out + op, lits + lp, ll); op += ll; lp += ll;
        if (seq_dump) fprintf(seq_dump, "%llu %llu %llu %llu\n", (unsigned long long)(dst_pos + op), (unsigned long long)ll, (unsigned long long)ml, (unsigned long long)off);
        if (off > dst_pos + op - frame_start) return ERR_CORR;
        if (c->info && off > c->info->max_offset) c->info->max_offset = off;
        for (uint64_t k = 0; k < ml; k++) { out[op] = out[op - off]; op++; }
    }
    if (b.bits != 0) return ERR_CORR;
    size_t rest = regen - lp;
    if (op + rest > room) return ERR_DST;
    if (op + rest > MAX_BLOCK) return ERR_CORR;
    memcpy(out + op, lits + lp, rest); op += rest;
    return (long long)op;
}

static long long decode_frame(const uint8_t *src, size_t len, uint8_t *dst, size_t dst_pos, size_t dst_cap,
                              size_t *consumed, nafo_zstd_frame_info *info)
{
    seq_dump_open();
    if (len < 4) return ERR_SRC;
    uint32_t magic = src[0] | (src[1] << 8) | (src[2] << 16) | ((uint32_t)src[3] << 24);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {                /* skippable frame */
        if (len < 8) return ERR_SRC;
        uint32_t sz = src[4] | (src[5] << 8) | (src[6] << 16) | ((uint32_t)src[7] << 24);
        if ((size_t)sz + 8 > len) return ERR_SRC;
        if (consumed) *consumed = (size_t)sz + 8;
        return 0;
    }
    if (magic != 0xFD2FB528u) return ERR_SRC;
    const uint8_t *p = src + 4, *end = src + len;
    if (p >= end) return ERR_SRC;
    uint8_t fhd = *p++;
    int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did = fhd & 3;
    if (fhd & 8) return ERR_UNSUP;
    uint64_t window = 0; int wlog = 0;
    if (!single) {
        if (p >= end) return ERR_SRC;
        uint8_t wd = *p++; wlog = 10 + (wd >> 3);
        window = (1ull << wlog) + ((1ull << wlog) >> 3) * (wd & 7);
    }
    static const int did_size[4] = { 0, 1, 2, 4 };
    if (did) { if (end - p < did_size[did]) return ERR_SRC; for (int i = 0; i < did_size[did]; i++) if (p[i]) return ERR_UNSUP; p += did_size[did]; }
    int fcs_size = fcs_flag == 0 ? single : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
    uint64_t fcs = 0;
    if (end - p < fcs_size) return ERR_SRC;
    for (int i = 0; i < fcs_size; i++) fcs |= (uint64_t)p[i] << (8 * i);
    if (fcs_size == 2) fcs += 256;
    p += fcs_size;
    if (single) window = fcs;
    (void)window;
    if (info) { memset(info, 0, sizeof *info); info->window_log = (uint32_t)wlog; info->single_segment = (uint32_t)single; info->has_checksum = (uint32_t)checksum; info->has_fcs = fcs_size != 0; }

    frame_ctx *c = (frame_ctx *)calloc(1, sizeof *c);
    if (!c) return ERR_DST;
    c->rep[0] = 1; c->rep[1] = 4; c->rep[2] = 8; c->info = info;
    size_t start = dst_pos; long long rc = 0;
    for (;;) {
        if (end - p < 3) { rc = ERR_SRC; break; }
        uint32_t bh = p[0] | (p[1] << 8) | ((uint32_t)p[2] << 16); p += 3;
        int last = bh & 1, type = (bh >> 1) & 3; size_t bsize = bh >> 3;
        if (info) info->n_blocks++;
        if (type == 0) {
            if (bsize > MAX_BLOCK) { rc = ERR_CORR; break; }
            if ((size_t)(end - p) < bsize) { rc = ERR_SRC; break; }
            if (dst_cap - dst_pos < bsize) { rc = ERR_DST; break; }
            memcpy(dst + dst_pos, p, bsize); dst_pos += bsize; p += bsize; if (info) info->n_raw++;
        } else if (type == 1) {
            if (bsize > MAX_BLOCK) { rc = ERR_CORR; break; }
            if (end - p < 1) { rc = ERR_SRC; break; }
            if (dst_cap - dst_pos < bsize) { rc = ERR_DST; break; }
            memset(dst + dst_pos, *p, bsize); dst_pos += bsize; p += 1; if (info) info->n_rle++;
        } else if (type == 2) {
            if (bsize > MAX_BLOCK) { rc = ERR_CORR; break; }
            if ((size_t)(end - p) < bsize) { rc = ERR_SRC; break; }
            long long n = decode_compressed_block(c, p, bsize, dst, dst_pos, dst_cap, start);
            if (n < 0) { rc = n; break; }
            dst_pos += (size_t)n; p += bsize; if (info) info->n_compressed++;
        } else { rc = ERR_CORR; break; }
        if (last) break;
    }
    free(c);
    if (rc < 0) return rc;
    if (checksum) { if (end - p < 4) return ERR_SRC; p += 4; }   /* XXH64 low 32 bits: not verified */
    if (fcs_size && fcs != dst_pos - start) return ERR_CORR;
    if (consumed) *consumed = (size_t)(p - src);
    return (long long)(dst_pos - start);
}

long long nafo_zstd_decompress_frame(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap, size_t *consumed)
{
    return decode_frame(src, src_len, dst, 0, dst_cap, consumed, NULL);
}

long long nafo_zstd_decompress(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap)
{
    size_t pos = 0, out = 0;
    while (pos < src_len) {
        size_t used = 0;
        long long n = decode_frame(src + pos, src_len - pos, dst + out, 0, dst_cap - out, &used, NULL);
        if (n < 0) return n;
        out += (size_t)n; pos += used;
    }
    return (long long)out;
}

long long nafo_zstd_decompressed_size(const uint8_t *src, size_t src_len)
{
    /* no shortcut through Frame_Content_Size: the reference's ennaf frames never carry it */
    size_t cap = src_len * 4 + (1u << 20);
    for (;;) {
        uint8_t *tmp = (uint8_t *)malloc(cap);
        if (!tmp) return ERR_DST;
        long long n = nafo_zstd_decompress(src, src_len, tmp, cap);
        free(tmp);
        if (n != ERR_DST) return n;
        cap *= 4;
        if (cap > ((size_t)1 << 40)) return ERR_DST;
    }
}

long long nafo_zstd_frame_info_get(const uint8_t *src, size_t src_len, nafo_zstd_frame_info *info)
{
    size_t cap = src_len * 4 + (1u << 20);
    for (;;) {
        uint8_t *tmp = (uint8_t *)malloc(cap);
        if (!tmp) return ERR_DST;
        long long n = decode_frame(src, src_len, tmp, 0, cap, NULL, info);
        free(tmp);
        if (n != ERR_DST) return n;
        cap *= 4;
        if (cap > ((size_t)1 << 40)) return ERR_DST;
    }
}

long long nafo_zstd_store_raw(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap)
{
    /* magic, FHD=0 (no FCS, no checksum, windowed), Window_Descriptor 0x48 = 2^19 like ennaf -1 */
    size_t nblocks = src_len ? (src_len + MAX_BLOCK - 1) / MAX_BLOCK : 1;
    size_t need = 6 + nblocks * 3 + src_len;
    if (need > dst_cap) return ERR_DST;
    uint8_t *p = dst;
    *p++ = 0x28; *p++ = 0xB5; *p++ = 0x2F; *p++ = 0xFD; *p++ = 0x00; *p++ = 0x48;
    size_t pos = 0;
    for (size_t b = 0; b < nblocks; b++) {
        size_t n = src_len - pos < MAX_BLOCK ? src_len - pos : MAX_BLOCK;
        uint32_t bh = (uint32_t)(n << 3) | (b + 1 == nblocks ? 1u : 0u);
        *p++ = (uint8_t)bh; *p++ = (uint8_t)(bh >> 8); *p++ = (uint8_t)(bh >> 16);
        memcpy(p, src + pos, n); p += n; pos += n;
    }
    return (long long)(p - dst);
}
