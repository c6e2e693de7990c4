// zstd_emul.cpp -- DEVELOPMENT/TEST HARNESS, not product code.
// Compiles the per-lane kernel logic of naf_amd/csrc/zstd_dec_core.h for the host and single-steps it
// in the same phase order as the HIP kernels (scan -> parse -> ownership -> tables -> sequences ->
// rep chain -> offsets -> literals -> execute), so kernel logic can be checked on a machine without a
// GPU.  Nothing in libnaf_gpu.so links this file; the GPU tests exercise the real kernels.
#include "../../naf_amd/csrc/zstd_dec_core.h"
#include <vector>
#include <stdlib.h>

static unsigned g_max_huf_log = 0;
extern "C" unsigned emul_max_huf_log(void) { unsigned v = g_max_huf_log; g_max_huf_log = 0; return v; }
extern "C" long long emul_zstd_decompress_frame(const u8 *src, size_t len, u8 *dst, size_t cap)
{
    if (len < 5 || ld32(src) != 0xFD2FB528u) return -1;
    src += 4; len -= 4;
    ZFrameHdr fh = zstd_parse_frame_header(src, len);
    if (fh.err) return -10 - fh.err;
    // k_scan_blocks
    std::vector<ZBlock> blk;
    u64 pos = fh.hdr_size;
    for (;;) {
        if (pos + 3 > len) return -2;
        u32 h = ld24(src + pos), last = h & 1, type = (h >> 1) & 3, size = h >> 3;
        if (type == 3 || size > ZBLOCK_MAX) return -3;
        u32 csize = type == BT_RLE ? 1 : size;
        if (pos + 3 + csize > len) return -2;
        ZBlock b; memset(&b, 0, sizeof b);
        b.src_off = pos + 3; b.bsize = size; b.btype = (u8)type; b.last = (u8)last;
        blk.push_back(b); pos += 3 + csize;
        if (last) break;
    }
    u32 n = (u32)blk.size();
    // k_parse_blocks + ownership (inclusive max-scan)
    std::vector<i32> own[4];
    for (auto &o : own) o.assign(n, -1);
    for (u32 i = 0; i < n; i++) {
        zstd_parse_block(src + blk[i].src_off, blk[i]);
        if (blk[i].err) return -100 - blk[i].err;
        bool comp = blk[i].btype == BT_COMP, sq = comp && blk[i].nseq > 0;
        own[0][i] = comp && blk[i].lit_type == LIT_HUF ? (i32)i : -1;
        for (int k = 0; k < 3; k++) own[1 + k][i] = sq && blk[i].modes[k] != SM_REPEAT ? (i32)i : -1;
    }
    for (auto &o : own) for (u32 i = 1; i < n; i++) if (o[i - 1] > o[i]) o[i] = o[i - 1];
    // k_build_huf / k_build_fse
    std::vector<std::vector<u16>> huf(n);
    std::vector<std::vector<FseE>> fse[3]; for (auto &f : fse) f.resize(n);
    FseE predef[160]; u16 nx[64];
    static const i16 LL[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
    static const i16 OF[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };
    static const i16 ML[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
    fse_build_table(predef, LL, 36, 6, nx); fse_build_table(predef + 64, OF, 29, 5, nx); fse_build_table(predef + 96, ML, 53, 6, nx);
    const u32 max_log[3] = { 9, 8, 9 }, max_sym[3] = { 35, 31, 52 };
    for (u32 i = 0; i < n; i++) {
        ZBlock &b = blk[i]; const u8 *c = src + b.src_off;
        if (b.btype != BT_COMP) continue;
        if (b.lit_type == LIT_HUF) {
            u8 w[256]; u32 nw = 0, used = 0;
            u32 log = huf_read_weights(c + b.lit_off, b.lit_csize, w, &nw, &used);
            if (!log) return -4;
            huf[i].resize(huf_tab_bytes(log) / 2); huf_build_any(huf[i].data(), w, nw, log); b.huf_log = (u8)log; if (log > g_max_huf_log) g_max_huf_log = log;
            if (used != b.huf_streams_off - b.lit_off) return -5;
        }
        if (b.nseq) {
            u32 p = b.seq_off;
            for (int k = 0; k < 3; k++) {
                if (b.modes[k] == SM_RLE) p++;
                else if (b.modes[k] == SM_FSE) {
                    i16 norm[64]; u32 nsym, log;
                    u32 d = fse_read_ncount(c + p, b.bsize - p, max_log[k], max_sym[k], norm, &nsym, &log);
                    if (!d || log != b.fse_log[k]) return -6;
                    p += d; fse[k][i].resize(1u << log);
                    if (!fse_build_table(fse[k][i].data(), norm, nsym, log, nx)) return -7;
                }
            }
            if (p != b.seq_bits_off) return -8;
        }
    }
    // k_decode_seq
    std::vector<std::vector<u32>> sll(n), sml(n), sof(n);
    for (u32 i = 0; i < n; i++) {
        ZBlock &b = blk[i];
        if (b.btype != BT_COMP || !b.nseq) continue;
        SeqTab tab[3]; const u32 po[3] = { 0, 64, 96 }, pl[3] = { 6, 5, 6 };
        for (int k = 0; k < 3; k++) {
            i32 ob = own[1 + k][i]; if (ob < 0) return -9;
            u32 m = blk[ob].modes[k];
            tab[k].rle = m == SM_RLE; tab[k].rle_sym = blk[ob].fse_tab[k];
            if (m == SM_PREDEF) { tab[k].t = predef + po[k]; tab[k].log = pl[k]; }
            else if (m == SM_FSE) { tab[k].t = fse[k][ob].data(); tab[k].log = blk[ob].fse_log[k]; }
            else { tab[k].t = predef; tab[k].log = 0; }
        }
        sll[i].resize(b.nseq); sml[i].resize(b.nseq); sof[i].resize(b.nseq);
        u64 tl = 0, tm = 0; u32 ro[3];
        u8 e = 0xFF;
        if (blk[own[1][i]].modes[0] == SM_PREDEF && blk[own[2][i]].modes[1] == SM_PREDEF && blk[own[3][i]].modes[2] == SM_PREDEF) {
            u32 llt[36], mlt[53]; zstd_seq_code_tables(llt, mlt, 0, 1);          // the kernel's fast routine for blocks of predefined tables
            e = zstd_decode_sequences_predef<const FseE *, const u32 *, true>(src + b.src_off + b.seq_bits_off, b.seq_bits_size, b.nseq, predef, predef + 64, predef + 96, llt, mlt,
                                                                             sll[i].data(), sml[i].data(), sof[i].data(), ro, &tl, &tm, nullptr);
        }
        if (e == 0xFF) e = zstd_decode_sequences<BitReloadWindow, true>(src + b.src_off + b.seq_bits_off, b.seq_bits_size, b.nseq, tab, sll[i].data(), sml[i].data(), sof[i].data(), ro, &tl, &tm, nullptr, BitReloadWindow());
        if (e) return -200 - e;
        if (tl > b.lit_regen) return -11;
        b.rep_out[0] = ro[0]; b.rep_out[1] = ro[1]; b.rep_out[2] = ro[2];
        b.regen = (u32)(b.lit_regen + tm);
    }
    // k_rep_chain + offsets
    u32 rep[3] = { 1, 4, 8 }; u64 off = 0;
    for (u32 i = 0; i < n; i++) {
        ZBlock &b = blk[i];
        b.out_off = off; off += b.regen;
        if (b.btype == BT_COMP && b.nseq) {
            b.rep_in[0] = rep[0]; b.rep_in[1] = rep[1]; b.rep_in[2] = rep[2];
            u32 r0 = sym_resolve(b.rep_out[0], rep), r1 = sym_resolve(b.rep_out[1], rep), r2 = sym_resolve(b.rep_out[2], rep);
            rep[0] = r0; rep[1] = r1; rep[2] = r2;
        }
    }
    if (off > cap) return -12;
    if (fh.has_fcs && fh.content_size != off) return -13;
    // literals + execution
    std::vector<u8> lit(ZBLOCK_MAX + 64);
    for (u32 i = 0; i < n; i++) {
        ZBlock &b = blk[i]; const u8 *c = src + b.src_off; u8 *out = dst + b.out_off;
        if (b.btype == BT_RAW) { memcpy(out, c, b.bsize); continue; }
        if (b.btype == BT_RLE) { memset(out, c[0], b.bsize); continue; }
        u8 *lp = b.nseq ? lit.data() : out;
        if (b.lit_type == LIT_RAW) memcpy(lp, c + b.lit_off, b.lit_regen);
        else if (b.lit_type == LIT_RLE) memset(lp, c[b.lit_off], b.lit_regen);
        else {
            i32 ob = own[0][i]; if (ob < 0) return -14;
            const u16 *tab = huf[ob].data(); u32 log = blk[ob].huf_log;
            const u8 *s = c + b.huf_streams_off;
            if (b.nstreams == 1) { if (huf_decode_stream(s, b.huf_streams_size, tab, log, lp, b.lit_regen)) return -15; }
            else {
                u32 s1 = ld16(s), s2 = ld16(s + 2), s3 = ld16(s + 4), tot = b.huf_streams_size - 6, per = (b.lit_regen + 3) / 4;
                if (s1 + s2 + s3 >= tot) return -16;
                u32 offs[4] = { 0, s1, s1 + s2, s1 + s2 + s3 }, szs[4] = { s1, s2, s3, tot - s1 - s2 - s3 };
                for (u32 k = 0; k < 4; k++)
                    if (huf_decode_stream(s + 6 + offs[k], szs[k], tab, log, lp + k * per, k < 3 ? per : b.lit_regen - 3 * per)) return -17;
            }
        }
        if (!b.nseq) continue;
        u32 op = 0, l = 0;
        for (u32 q = 0; q < b.nseq; q++) {
            u32 ll = sll[i][q], ml = sml[i][q], of = sym_resolve(sof[i][q], b.rep_in);
            memcpy(out + op, lit.data() + l, ll); op += ll; l += ll;
            if (of > b.out_off + op) return -18;
            const u8 *from = out + op - of;
            for (u32 k = 0; k < ml; k++) out[op + k] = from[of >= ml ? k : k % of];
            op += ml;
        }
        memcpy(out + op, lit.data() + l, b.lit_regen - l); op += b.lit_regen - l;
        if (op != b.regen) return -19;
    }
    return (long long)off;
}

// ---- the sector-window reader of k_huf_literals, single-stepped for one stream --------------------------------------------
// Same statements as the kernel's per-lane loop (ring of two 64-byte sectors in "LDS", register-staged prefetch), so the
// window arithmetic can be checked on the host against the plain reader for any stream shape.  Returns 0 when both agree.
extern "C" int emul_window_reader(const u8 *stream, u32 size, const u8 *weights, u32 nw, u32 log, u32 n, u64 align_off, int big)
{
    std::vector<u8> hay(size + 1024 + 256);
    u8 *base = hay.data() + 256; base += (64 - ((uintptr_t)base & 63)) & 63; base += align_off;
    memset(hay.data(), 0xAA, hay.size());
    memcpy(base, stream, size);
    std::vector<u16> tabv(huf_tab_bytes(log) / 2); huf_build_any(tabv.data(), weights, nw, log); const u16 *tab = tabv.data();
    std::vector<u8> ref(n + 64), out(n + 64);
    if (huf_decode_stream(base, size, tab, log, ref.data(), n)) return -1;
    BitR br; bitr_init(br, base, size); if (br.bad) return -2;
    if (!big && log > 7) return -5;
    const u32 HUF_ROUND = 32, rmask = big ? 255u : 127u, guard = big ? 192u : 160u;
    u8 irow[264];
    u32 rounds = n / HUF_ROUND, R = 0;
    bool live = (u64)(br.ptr - br.start) >= guard + 32;
    u64 gp = (u64)br.ptr, lo = 0; u8 st[64]; bool pending = false;
    if (live) { u64 top = (gp + 7) & ~63ull; lo = top - 64; for (int q = 0; q < 8; q++) memcpy(irow + ((lo + 16 * q) & rmask), (const u8 *)(lo + 16 * q), 16); }
    u32 bits = br.consumed;
    for (; R < rounds; R++) {
        if (!(live && gp - (u64)br.start >= guard)) break;
        if (pending) { lo -= 64; memcpy(irow + (lo & rmask), st, 64); pending = false; }
        if (lo + (big ? 96u : 56u) > gp) { memcpy(st, (const u8 *)(lo - 64), 64); pending = true; }
        for (u32 g = 0; g < HUF_ROUND / 8; g++) {
            for (u32 h = 0; h < (big ? 2u : 1u); h++) {
                gp -= bits >> 3; bits &= 7;
                u32 o = (u32)(gp & rmask), sh = (o & 7) * 8;
                u64 q0, q1; memcpy(&q0, irow + (o & ~7u), 8); memcpy(&q1, irow + (((o & ~7u) + 8) & rmask), 8);
                u64 w = (sh ? (q0 >> sh) | (q1 << (64 - sh)) : q0) << bits;
                for (u32 q = 0; q < (big ? 4u : 8u); q++) {
                    u32 e = huf_look(tab, (u32)(w >> 32) >> (32 - log), log); u32 nb = hufe_nb(e); w <<= nb; bits += nb;
                    out[R * HUF_ROUND + g * 8 + (big ? 4 * h : 0) + q] = (u8)hufe_sym(e);
                }
            }
        }
    }
    if (live) { gp -= bits >> 3; bits &= 7; br.c = ld64((const u8 *)gp); br.consumed = bits; }
    br.ptr = (const u8 *)gp;
    u32 done = R * HUF_ROUND;
    if (huf_decode_n(br, tab, log, out.data() + done, n - done)) return -3;
    bitr_reload(br); if (!bitr_finished(br)) return -4;
    for (u32 i = 0; i < n; i++) if (out[i] != ref[i]) return (int)i + 1;
    return 0;
}

// ---- a stream decoded in parts (k_huf_par), single-stepped: the lanes of one stream one after the other, round after round ----------
// Same per-lane functions as the kernel (zstd_dec_core.h: hufw_*, hufp_*).  Returns 0 when the parts reproduce the serial decode;
// *rounds_out = rounds of re-walking it took (0: every part fell into step inside its margin).  margin = 0: the kernel's own choice.
extern "C" int emul_huf_parts(const u8 *stream, u32 size, const u8 *weights, u32 nw, u32 log, u32 n, u32 P, u32 margin, u64 align_off, u32 *rounds_out)
{
    const bool tight = (align_off >> 32) != 0; align_off &= 0xFFFFFFFFull;   // tight: the stream IS the readable buffer (sector loads at both ends fall back to bytes)
    std::vector<u8> hay(size + 1024 + 256);
    u8 *base = hay.data() + 256 + align_off;
    memset(hay.data(), 0xAA, hay.size());                          // whatever lies around the stream must not matter
    memcpy(base, stream, size);
    const u8 *lo_ok = tight ? base : hay.data(), *hi_ok = tight ? base + size : hay.data() + hay.size();
    std::vector<u16> tabv(huf_tab_bytes(log) / 2); huf_build_any(tabv.data(), weights, nw, log); const u16 *tab = tabv.data();
    std::vector<u8> ref(n + 64), out(n + 64, 0x55);
    if (huf_decode_stream(base, size, tab, log, ref.data(), n)) return -1;
    if (!size || !base[size - 1] || P < 1 || P > 64) return -2;
    const u32 E = 8 * (size - 1) + (u32)hibit32(base[size - 1]);
    const u32 M = margin ? margin : hufp_margin(E, n, log);
    i32 s[64], e[64]; u32 c[64];
    for (u32 k = 0; k < P; k++) {
        const i32 Bk = hufp_cut(E, P, k), Bk1 = hufp_cut(E, P, k + 1);
        i32 p = k == 0 ? (i32)E : ((u64)Bk + M < E ? Bk + (i32)M : (i32)E);
        HufWin r; hufw_init(r, base, size, p, lo_ok, hi_ok);
        if (k) hufw_walk(r, p, Bk, tab, log);
        s[k] = p;
        c[k] = hufw_walk(r, p, Bk1, tab, log);
        e[k] = p;
    }
    u32 rounds = 0;
    for (;;) {
        bool any = false; i32 ns[64]; bool mis[64];
        for (u32 k = 0; k < P; k++) { mis[k] = k > 0 && s[k] != e[k - 1]; ns[k] = k ? e[k - 1] : s[0]; any |= mis[k]; }   // all lanes look before any lane moves
        if (!any) break;
        if (++rounds > 64) return -3;
        for (u32 k = 0; k < P; k++) if (mis[k]) {
            i32 p = ns[k]; s[k] = p;
            HufWin r; hufw_init(r, base, size, p, lo_ok, hi_ok);
            c[k] = hufw_walk(r, p, hufp_cut(E, P, k + 1), tab, log);
            e[k] = p;
        }
    }
    if (rounds_out) *rounds_out = rounds;
    u32 tot = 0; for (u32 k = 0; k < P; k++) tot += c[k];
    if (tot != n || e[P - 1] != 0) return -4;
    u32 off = 0;
    for (u32 k = 0; k < P; k++) {
        i32 p = s[k]; HufWin r; hufw_init(r, base, size, p, lo_ok, hi_ok);
        hufw_decode(r, p, tab, log, out.data() + off, c[k]);
        if (p != e[k]) return -5;
        off += c[k];
    }
    for (u32 i = 0; i < n; i++) if (out[i] != ref[i]) return (int)i + 1;
    if (out[n] != 0x55) return -6;                                 // nothing written behind the last symbol ... except by the 8-byte group stores inside
    return 0;
}

// block statistics of a frame (no magic check on the decoded data): out[0] blocks, [1] compressed blocks, [2] sequences,
// [3] literal bytes regenerated, [4] bytes of literals sections, [5] bytes of sequences sections
extern "C" long long emul_zstd_frame_stats(const u8 *src, size_t len, u64 *out)
{
    if (len < 5 || ld32(src) != 0xFD2FB528u) return -1;
    src += 4; len -= 4;
    ZFrameHdr fh = zstd_parse_frame_header(src, len);
    if (fh.err) return -10 - fh.err;
    u64 pos = fh.hdr_size;
    for (int k = 0; k < 6; k++) out[k] = 0;
    for (;;) {
        if (pos + 3 > len) return -2;
        u32 h = ld24(src + pos), last = h & 1, type = (h >> 1) & 3, size = h >> 3;
        u32 csize = type == BT_RLE ? 1 : size;
        if (pos + 3 + csize > len) return -2;
        ZBlock b; memset(&b, 0, sizeof b);
        b.src_off = pos + 3; b.bsize = size; b.btype = (u8)type; b.last = (u8)last;
        out[0]++;
        if (type == BT_COMP) {
            zstd_parse_block(src + b.src_off, b);
            if (b.err) return -100 - b.err;
            out[1]++; out[2] += b.nseq; out[3] += b.lit_regen; out[4] += b.seq_off ? b.seq_off : size; out[5] += b.seq_off ? size - b.seq_off : 0;
        } else out[3] += size;
        pos += 3 + csize;
        if (last) break;
    }
    return (long long)pos;
}
