// zstd_enc_emul.cpp -- DEVELOPMENT/TEST HARNESS (host build of the encoder's per-lane logic).
#include "../../naf_amd/csrc/zstd_enc_core.h"
#include <vector>

extern "C" long long emul_zstd_compress(const u8 *src, size_t n, u32 block, u8 *dst, size_t cap)
{
    u8 *p = dst;
    *p++ = 0x28; *p++ = 0xB5; *p++ = 0x2F; *p++ = 0xFD; *p++ = 0x00; *p++ = 0x58;   // FHD 0, windowLog 21
    size_t nblk = n ? (n + block - 1) / block : 1;
    for (size_t b = 0; b < nblk; b++) {
        const u8 *s = src + b * block; u32 bn = (u32)(n - b * block < block ? n - b * block : block);
        std::vector<u32> hist(1024, 0);
        u32 per = (bn + 3) / 4;
        for (u32 i = 0; i < bn; i++) hist[256 * (i / (per ? per : 1) > 3 ? 3 : i / (per ? per : 1)) + s[i]]++;
        ZEncPlan pl; u8 len[256], tree[192];
        zenc_plan_block(hist.data(), bn, pl, len, tree);
        if ((size_t)(p - dst) + pl.csize > cap) return -1;
        u32 off = zenc_write_block_prefix(p, pl, tree, b + 1 == nblk, bn ? s[0] : 0);
        if (pl.kind == ZK_RAW) memcpy(p + off, s, bn);
        else if (pl.kind == ZK_HUF) {
            u16 code[256]; u32 codes[256];
            huf_assign_codes(len, pl.log, code);
            for (u32 i = 0; i < 256; i++) codes[i] = code[i] | ((u32)len[i] << 16);
            u32 o = off;
            for (u32 k = 0; k < 4; k++) {
                u32 cnt = k < 3 ? per : bn - 3 * per;
                u32 w = huf_encode_stream(p + o, s + k * per, cnt, codes);
                if (w != pl.ssz[k]) return -100 - (long long)k;
                o += w;
            }
            p[o] = 0;                                           // Number_of_Sequences = 0
            if (o + 1 != pl.csize) return -2;
        }
        p += pl.csize;
    }
    return p - dst;
}

// ---- LZ stage, serial reference of the block format the GPU LZ path writes --------------------------------------------------
// Greedy hash matching inside each block (offsets never leave the block), literals Huffman-coded like the literal-only
// path, sequences with predefined FSE tables.  Used to pin the FORMAT of LZ blocks against the from-spec decoder and libzstd.
#include <vector>
#include <string.h>
static u32 lz_hash(u32 v) { return (v * 2654435761u) >> 20; }     // 12 bits
extern "C" long long emul_zstd_compress_lz(const u8 *src, size_t n, u32 block, u8 *dst, size_t cap)
{
    if (block > 32768) return -30;                                // lengths and distances are kept in 16 bits
    u8 *p = dst;
    *p++ = 0x28; *p++ = 0xB5; *p++ = 0x2F; *p++ = 0xFD; *p++ = 0x00; *p++ = 0x58;   // FHD 0, windowLog 21
    SeqCTabs T; zenc_build_predefined(T); SeqCTab ct[3]; zenc_seq_ctabs(T, ct);
    size_t nblk = n ? (n + block - 1) / block : 1;
    std::vector<u8> lits(block + 16), seqb(block * 4 + 64);
    std::vector<u16> ll(block / 4 + 2 + 8), ml(block / 4 + 2 + 8), of(block / 4 + 2 + 8);   // + 8: read in groups of eight
    for (size_t b = 0; b < nblk; b++) {
        const u8 *s = src + b * block; u32 bn = (u32)(n - b * block < block ? n - b * block : block);
        bool last = b + 1 == nblk;
        // greedy parse
        std::vector<i32> tab(4096, -1);
        u32 nseq = 0, nl = 0, anchor = 0, i = 0;
        while (i + 4 <= bn) {
            u32 v; memcpy(&v, s + i, 4);
            u32 h = lz_hash(v); i32 c = tab[h]; tab[h] = (i32)i;
            u32 cv = 0; if (c >= 0) memcpy(&cv, s + c, 4);
            if (c >= 0 && cv == v) {
                u32 m = 4; while (i + m < bn && s[c + m] == s[i + m]) m++;
                ll[nseq] = (u16)(i - anchor); ml[nseq] = (u16)m; of[nseq] = (u16)(i - c); nseq++;
                memcpy(lits.data() + nl, s + anchor, i - anchor); nl += i - anchor;
                i += m; anchor = i;
            } else i++;
        }
        memcpy(lits.data() + nl, s + anchor, bn - anchor); nl += bn - anchor;
        // literals section
        u8 litsec[ZBLOCK_MAX + 512]; u32 lsz = 0;
        u32 hist[1024]; memset(hist, 0, sizeof hist);
        u32 per = (nl + 3) / 4; if (!per) per = 1;
        for (u32 k = 0; k < nl; k++) { u32 q = k / per; if (q > 3) q = 3; hist[q * 256 + lits[k]]++; }
        ZEncPlan pl; u8 len[256], tree[192];
        zenc_plan_block(hist, nl, pl, len, tree);
        if (nl == 0) lsz = zenc_lit_header_raw(litsec, 0, 0);
        else if (pl.kind == ZK_RLE) { lsz = zenc_lit_header_raw(litsec, 1, nl); litsec[lsz++] = lits[0]; }
        else if (pl.kind == ZK_RAW) { lsz = zenc_lit_header_raw(litsec, 0, nl); memcpy(litsec + lsz, lits.data(), nl); lsz += nl; }
        else {
            u16 code[256]; u32 codes[256];
            huf_assign_codes(len, pl.log, code);
            for (u32 k = 0; k < 256; k++) codes[k] = code[k] | ((u32)len[k] << 16);
            u32 o = zenc_write_huf_lit_prefix(litsec, pl, tree);
            for (u32 k = 0; k < 4; k++) {
                u32 cnt = k < 3 ? per : nl - 3 * per;
                u32 w = huf_encode_stream(litsec + o, lits.data() + (size_t)k * per, cnt, codes);
                if (w != pl.ssz[k]) return -31;
                o += w;
            }
            lsz = o;
        }
        u32 ssz = 1;
        if (nseq) { ssz = zenc_write_sequences(seqb.data(), (u32)seqb.size(), ll.data(), ml.data(), of.data(), nseq, ct); if (!ssz) return -32; }
        else seqb[0] = 0;
        if ((size_t)(p - dst) + 3 + bn + 16 > cap || (size_t)(p - dst) + 3 + lsz + ssz + 16 > cap) return -33;
        if (lsz + ssz >= bn) { zenc_write_block_header(p, 0, bn, last); memcpy(p + 3, s, bn); p += 3 + bn; }
        else { zenc_write_block_header(p, 2, lsz + ssz, last); memcpy(p + 3, litsec, lsz); memcpy(p + 3 + lsz, seqb.data(), ssz); p += 3 + lsz + ssz; }
    }
    return p - dst;
}

// symbols -> one Huffman stream (this build's own code construction) -> the decoder kernel's window reader, at a given
// byte alignment of the stream start; 0 when the window reader reproduces the plain reader.
extern "C" int emul_window_reader(const u8 *stream, u32 size, const u8 *weights, u32 nw, u32 log, u32 n, u64 align_off, int big);
extern "C" int emul_window_roundtrip(const u8 *syms, u32 n, u64 align_off, int big)
{
    u32 hist[256]; memset(hist, 0, sizeof hist);
    for (u32 i = 0; i < n; i++) hist[syms[i]]++;
    u8 len[256]; u32 log = huf_build_lengths(hist, len);
    if (!log) return -10;
    u8 w[256]; u32 last = 0;
    for (u32 s = 0; s < 256; s++) { w[s] = len[s] ? (u8)(log + 1 - len[s]) : 0; if (len[s]) last = s; }
    u16 code[256]; u32 codes[256];
    huf_assign_codes(len, log, code);
    for (u32 k = 0; k < 256; k++) codes[k] = code[k] | ((u32)len[k] << 16);
    std::vector<u8> st(n * 2 + 64);
    u32 sz = huf_encode_stream(st.data(), syms, n, codes);
    return emul_window_reader(st.data(), sz, w, last + 1, log, n, align_off, big);
}

// ---- encoder front end: SWAR byte classes (enc_swar.h) ----------------------------------------------------------------------
#include "../../naf_amd/csrc/enc_swar.h"
extern "C" void emul_piece_flags(const uint8_t *piece16, uint32_t qlo, uint32_t qhi, uint32_t out[8])
{
    u32 w[4]; memcpy(w, piece16, 16);
    PieceFlags f = piece_flags(w);
    u32 nq = piece_not_quick(w, qlo, qhi);
    out[0] = f.eol; out[1] = f.sp; out[2] = f.gt; out[3] = (nq & ~f.sp) == 0; out[4] = nq == 0; out[5] = piece_all_quality(w); out[6] = piece_ctl_mask(w); out[7] = piece_not_quality_mask(w);
}
extern "C" int emul_piece_plain(const uint8_t *piece16, uint32_t plo, uint32_t phi, uint32_t *eol)
{
    u32 w[4]; memcpy(w, piece16, 16);
    return piece_plain(w, plo, phi, eol) ? 1 : 0;
}
