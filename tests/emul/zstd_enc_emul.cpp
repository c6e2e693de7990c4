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

// symbols -> one Huffman stream -> the parts algorithm of k_huf_par (zstd_emul.cpp: emul_huf_parts)
extern "C" int emul_huf_parts(const u8 *stream, u32 size, const u8 *weights, u32 nw, u32 log, u32 n, u32 P, u32 margin, u64 align_off, u32 *rounds_out);
extern "C" int emul_parts_roundtrip(const u8 *syms, u32 n, u32 P, u32 margin, u64 align_off, u32 *rounds_out)
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
    return emul_huf_parts(st.data(), sz, w, last + 1, log, n, P, margin, align_off, rounds_out);
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

// ---- cross-block matching with repeat offsets and per-block FSE tables: a serial model of k_ldm_insert + k_lz_parse (level >= 2, --long) ----
// anchors: positions whose 8-byte hash has its top three bits clear; table of FIRST occurrences per epoch of 2^(wlog-1) bytes
static inline u64 ldm_mix(u64 a) { return a * 0x9E3779B185EBCA87ull; }
extern "C" long long emul_zstd_compress_lzx(const u8 *src, size_t n, u32 block, u32 wlog, u8 *dst, size_t cap, u32 *stats /* nseq, nrep, nlit */)
{
    if (block > 65535 || wlog < 10 || wlog > 31) return -30;
    u8 *p = dst;
    *p++ = 0x28; *p++ = 0xB5; *p++ = 0x2F; *p++ = 0xFD; *p++ = 0x00; *p++ = (u8)((wlog - 10) << 3);
    SeqCTabs T; zenc_build_predefined(T);
    std::vector<u8> pad(src, src + n); pad.resize(n + 64, 0);
    const u8 *s0 = pad.data();
    const u32 elog = wlog - 1; const u64 E = 1ull << elog;
    const u64 nep = (n >> elog) + 1;
    u32 tlog = elog > 3 ? elog - 3 + 1 : 4; if (tlog > 22) tlog = 22;
    { u32 need = 4; while ((1ull << need) < (n < E ? n : E) / 4 + 16) need++; if (need < tlog) tlog = need; }
    std::vector<u32> tab((size_t)nep << tlog, 0xFFFFFFFFu);
    auto key_of = [&](u64 pos, u32 &idx) -> bool {
        u64 a, b; memcpy(&a, s0 + pos, 8); memcpy(&b, s0 + pos + 8, 8);
        if ((ldm_mix(a) >> 61) != 0) return false;
        idx = (u32)((ldm_mix(a) ^ (ldm_mix(b ^ 0x5555555555555555ull) >> 7)) >> (64 - tlog));
        return true;
    };
    for (u64 pos = 0; pos + 16 <= n; pos++) { u32 idx; if (key_of(pos, idx)) { u32 &t = tab[((pos >> elog) << tlog) + idx]; u32 rel = (u32)(pos & (E - 1)); if (rel < t) t = rel; } }
    size_t nblk = n ? (n + block - 1) / block : 1;
    std::vector<u8> lits(block + 16), seqb(block * 4 + 512);
    std::vector<u16> ll(block / 4 + 16), ml(block / 4 + 16); std::vector<u32> of(block / 4 + 16);
    SeqWS ws;
    u32 st_seq = 0, st_rep = 0, st_lit = 0;
    for (size_t b = 0; b < nblk; b++) {
        const u64 lo = (u64)b * block; const u8 *s = s0 + lo; u32 bn = (u32)(n - lo < block ? n - lo : block);
        bool last = b + 1 == nblk;
        std::vector<i32> itab(4096, -1);
        RepState R = { { 0, 0, 0 } };
        u32 nseq = 0, nl = 0, anchor = 0, i = 0;
        auto emit = [&](u32 at, u32 m, u32 d) {
            ll[nseq] = (u16)(at - anchor); ml[nseq] = (u16)m; of[nseq] = zenc_offset_value(R, d, at - anchor); if (of[nseq] <= 3) st_rep++;
            nseq++;
            memcpy(lits.data() + nl, s + anchor, at - anchor); nl += at - anchor;
            anchor = at + m;
        };
        auto mlen = [&](u32 at, u64 d) -> u32 { u32 m = 0; while (at + m < bn && s0[lo + at + m - d] == s[at + m]) m++; return m; };
        while (i + 4 <= bn) {
            u32 best_m = 0; u64 best_d = 0;
            u32 v; memcpy(&v, s + i, 4);
            u32 h = lz_hash(v); i32 c = itab[h]; itab[h] = (i32)i;
            if (c >= 0) { u32 m = mlen(i, i - (u32)c); if (m >= 5) { best_m = m; best_d = i - (u32)c; } }
            u32 idx;
            if (i + 16 <= bn && key_of(lo + i, idx)) {
                const u64 pa = lo + i, e = pa >> elog;
                u64 q = ~0ull;
                u32 c1 = tab[(e << tlog) + idx];
                if (c1 != 0xFFFFFFFFu && (e << elog) + c1 < pa) q = (e << elog) + c1;
                else if (e > 0) { u32 c0 = tab[((e - 1) << tlog) + idx]; if (c0 != 0xFFFFFFFFu) q = ((e - 1) << elog) + c0; }
                if (q != ~0ull && pa - q < (1ull << wlog) && !memcmp(s0 + q, s0 + pa, 16)) {
                    u32 m = mlen(i, pa - q);
                    if (m > best_m) { best_m = m; best_d = pa - q; }
                }
            }
            if (best_m >= 5) {
                u32 at = i;
                while (at > anchor && lo + at > best_d && s0[lo + at - 1 - best_d] == s[at - 1]) { at--; best_m++; }   // backwards into the pending literals
                emit(at, best_m, (u32)best_d);
                i = anchor;
                for (;;) {                                       // the same offset again behind one to three literals (a substituted base)
                    bool again = false;
                    for (u32 skip = 1; skip <= 3 && !again; skip++) {
                        if (i + skip + 4 > bn) break;
                        u32 m = mlen(i + skip, best_d);
                        if (m >= 4) { emit(i + skip, m, (u32)best_d); i = anchor; again = true; }
                    }
                    if (!again) break;
                }
            } else i++;
        }
        memcpy(lits.data() + nl, s + anchor, bn - anchor); nl += bn - anchor;
        st_seq += nseq; st_lit += nl;
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
        if (nseq) { ssz = zenc_write_sequences_x(seqb.data(), (u32)seqb.size(), ll.data(), ml.data(), of.data(), nseq, T, ws); if (!ssz) return -32; }
        else seqb[0] = 0;
        if ((size_t)(p - dst) + 3 + bn + 16 > cap || (size_t)(p - dst) + 3 + lsz + ssz + 16 > cap) return -33;
        if (lsz + ssz >= bn) { zenc_write_block_header(p, 0, bn, last); memcpy(p + 3, s, bn); p += 3 + bn; }
        else { zenc_write_block_header(p, 2, lsz + ssz, last); memcpy(p + 3, litsec, lsz); memcpy(p + 3 + lsz, seqb.data(), ssz); p += 3 + lsz + ssz; }
    }
    if (stats) { stats[0] = st_seq; stats[1] = st_rep; stats[2] = st_lit; }
    return p - dst;
}
