// zstd_enc_core.h -- per-lane logic of the zstd block encoder (RFC 8878 compliant output; replaces
// libzstd behind ennaf/src/compressor.c:119-147 compress() / :64-96 compressor_end_stream()).
//
// Blocks are coded independently (no repeat offsets, no treeless literals, no cross-block matches)
// so any block range can be produced by any GPU and still concatenates into ONE valid frame
// (reference unnaf accepts exactly one frame for sequence and quality).
// Level-1 style coding: literals-only Compressed blocks with a 4-stream Huffman literals section and
// zero sequences; RLE blocks for constant data; Raw blocks when entropy coding does not pay.
#pragma once
#include "common.h"
#include "zstd_dec_core.h"                                     // ll_base / ll_bits / ml_base / ml_bits for the sequences encoder

#define ZENC_HUF_MAXBITS 11

// ---- length-limited Huffman code lengths from a histogram -------------------------------------------------
// cnt[256] -> len[256] (0 for absent symbols).  Returns the maximum length (tableLog), or 0 when fewer
// than two distinct symbols are present.  Scratch: order[256], node arrays of 512 entries.
// Code lengths from symbols already sorted by (count, symbol) ascending: order[0..n), n >= 2.  len[] must be
// zero for absent symbols.  w/parent/depth: 512-entry workspaces (LDS on the GPU).
NAF_HD u32 huf_lengths_sorted(const u32 *cnt, const u16 *order, u32 n, u8 *len, u32 *w, u16 *parent, u8 *depth, u32 maxbits = ZENC_HUF_MAXBITS)
{
    // two-queue Huffman construction; parent links give depths
    for (u32 i = 0; i < n; i++) w[i] = cnt[order[i]];
    u32 leaf = 0, inode = n, next = n;              // queue 1: leaves [leaf,n) ; queue 2: internal [inode,next)
    while ((n - leaf) + (next - inode) > 1) {
        u32 a, b;
        if (leaf < n && (inode >= next || w[leaf] <= w[inode])) a = leaf++; else a = inode++;
        if (leaf < n && (inode >= next || w[leaf] <= w[inode])) b = leaf++; else b = inode++;
        w[next] = w[a] + w[b]; parent[a] = (u16)next; parent[b] = (u16)next; next++;
    }
    u32 root = next - 1;
    depth[root] = 0;
    for (u32 i = root; i-- > 0;) depth[i] = (u8)(depth[parent[i]] + 1);
    u32 maxlen = 0;
    for (u32 i = 0; i < n; i++) { u32 d = depth[i]; if (d > maxlen) maxlen = d; len[order[i]] = (u8)d; }
    if (maxlen <= maxbits) return maxlen;
    // length limiting: clamp, then repair the Kraft sum (units of 2^-MAXBITS) to exactly 1
    i32 K = 0;
    for (u32 i = 0; i < n; i++) { u8 &l = len[order[i]]; if (l > maxbits) l = maxbits; K += 1 << (maxbits - l); }
    const i32 full = 1 << maxbits;
    while (K > full) {                              // lengthen the rarest symbol that can still grow
        for (u32 i = 0; i < n && K > full; i++) {
            u8 &l = len[order[i]];
            if (l < maxbits) { K -= 1 << (maxbits - l - 1); l++; }
        }
    }
    while (K < full) {                              // shorten the most frequent symbols that fit
        bool any = false;
        for (u32 i = n; i-- > 0 && K < full;) {
            u8 &l = len[order[i]];
            if (l > 1 && K + (1 << (maxbits - l)) <= full) { K += 1 << (maxbits - l); l--; any = true; }
        }
        if (!any) break;
    }
    if (K != full) return 0;
    maxlen = 0;
    for (u32 i = 0; i < n; i++) if (len[order[i]] > maxlen) maxlen = len[order[i]];
    return maxlen;
}

NAF_HD u32 huf_build_lengths(const u32 *cnt, u8 *len)
{
    u16 order[256]; u32 n = 0;
    for (u32 s = 0; s < 256; s++) { len[s] = 0; if (cnt[s]) order[n++] = (u16)s; }
    if (n < 2) return 0;
    // sort present symbols by count ascending, ties by symbol (stable insertion sort)
    for (u32 i = 1; i < n; i++) {
        u16 v = order[i]; u32 c = cnt[v]; u32 j = i;
        while (j > 0 && cnt[order[j - 1]] > c) { order[j] = order[j - 1]; j--; }
        order[j] = v;
    }
    u32 w[512]; u16 parent[512]; u8 depth[512];
    return huf_lengths_sorted(cnt, order, n, len, w, parent, depth);
}

// Canonical code values in the order the zstd decoder expects (4.2.1): within the decoding table,
// symbols are laid out by increasing weight (= decreasing length), ties by symbol value; the code of a
// symbol is the index of its first table cell >> (log - len).
NAF_HD void huf_assign_codes(const u8 *len, u32 log, u16 *code)
{
    u32 cnt[ZENC_HUF_MAXBITS + 2], start[ZENC_HUF_MAXBITS + 2];
    for (u32 r = 0; r <= ZENC_HUF_MAXBITS + 1; r++) cnt[r] = 0;
    for (u32 s = 0; s < 256; s++) if (len[s]) cnt[log + 1 - len[s]]++;            // index by weight
    u32 pos = 0;
    for (u32 wgt = 1; wgt <= log; wgt++) { start[wgt] = pos; pos += cnt[wgt] << (wgt - 1); }
    for (u32 s = 0; s < 256; s++) {
        if (!len[s]) { code[s] = 0; continue; }
        u32 wgt = log + 1 - len[s];
        code[s] = (u16)(start[wgt] >> (wgt - 1));                                // = cell index >> (log - len)
        start[wgt] += 1u << (wgt - 1);
    }
}

// ---- forward bit writer (all zstd bit-streams are written forward, little-endian, read backward) -------------
struct BitW { u8 *p; u64 acc; u32 n; };
NAF_HD void bitw_init(BitW &b, u8 *p) { b.p = p; b.acc = 0; b.n = 0; }
NAF_HD void bitw_add(BitW &b, u32 v, u32 nbits) { b.acc |= (u64)(v & ((nbits >= 32) ? 0xFFFFFFFFu : ((1u << nbits) - 1))) << b.n; b.n += nbits; }
NAF_HD void bitw_flush(BitW &b) { while (b.n >= 8) { *b.p++ = (u8)b.acc; b.acc >>= 8; b.n -= 8; } }
// four bytes at once when there are that many: leaves fewer than 32 bits, so another 32 can be added before the next flush
NAF_HD void bitw_flush32(BitW &b) { if (b.n >= 32) { st32(b.p, (u32)b.acc); b.p += 4; b.acc >>= 32; b.n -= 32; } }
// final marker bit + padding; returns end pointer
NAF_HD u8 *bitw_close(BitW &b) { bitw_add(b, 1, 1); bitw_flush(b); if (b.n) { *b.p++ = (u8)b.acc; b.n = 0; } return b.p; }

// ---- FSE encoder for Huffman weights (4.2.1.2) --------------------------------------------------------------------
struct FseCSym { i32 deltaNbBits; i32 deltaFindState; };

// Normalize counts of symbols 0..maxsym to sum 2^log, every present symbol >= 1.  Returns false if impossible.
NAF_HD bool fse_normalize(const u32 *cnt, u32 maxsym, u32 total, u32 log, i16 *norm)
{
    u32 size = 1u << log, present = 0;
    for (u32 s = 0; s <= maxsym; s++) present += cnt[s] != 0;
    if (present > size || present < 2) return false;
    u32 sum = 0, big = 0;
    for (u32 s = 0; s <= maxsym; s++) {
        if (!cnt[s]) { norm[s] = 0; continue; }
        u32 v = (u32)(((u64)cnt[s] * size) / total);
        if (v == 0) v = 1;
        norm[s] = (i16)v; sum += v;
        if (cnt[s] > cnt[big] || !cnt[big]) big = s;
    }
    // repair the sum: first on the most frequent symbol, then round-robin
    while (sum != size) {
        if (sum < size) { norm[big] += (i16)(size - sum); sum = size; }
        else {
            u32 over = sum - size;
            if ((u32)norm[big] > over + 0) { norm[big] -= (i16)over; sum = size; if (norm[big] < 1) return false; }
            else {
                bool any = false;
                for (u32 s = 0; s <= maxsym && sum > size; s++) if (norm[s] > 1) { norm[s]--; sum--; any = true; }
                if (!any) return false;
            }
        }
    }
    return true;
}

// Table description (4.1.1), mirror of fse_read_ncount.  Returns bytes written.
NAF_HD u32 fse_write_ncount(u8 *out, const i16 *norm, u32 maxsym, u32 log)
{
    BitW b; bitw_init(b, out);
    bitw_add(b, log - 5, 4);
    i32 remaining = (1 << log) + 1, threshold = 1 << log; u32 nbits = log + 1, s = 0;
    while (remaining > 1 && s <= maxsym) {
        i32 count = norm[s++];
        i32 max = (2 * threshold - 1) - remaining;
        remaining -= count < 0 ? -count : count;
        count++;
        if (count >= threshold) count += max;
        bitw_add(b, (u32)count, nbits - (count < max ? 1 : 0));
        bitw_flush(b);
        if (count == 1) {                                      // probability 0: repeat flags
            u32 start = s;
            while (s <= maxsym && norm[s] == 0) s++;
            u32 run = s - start;
            while (run >= 3) { bitw_add(b, 3, 2); run -= 3; bitw_flush(b); }
            bitw_add(b, run, 2); bitw_flush(b);
        }
        while (remaining < threshold && threshold > 1) { nbits--; threshold >>= 1; }
    }
    bitw_flush(b);
    if (b.n) { *b.p++ = (u8)b.acc; }
    return (u32)(b.p - out);
}

// Workspace of the weight encoder (LDS on the GPU: dynamically indexed private arrays would live in scratch memory)
struct FseWS { u32 cnt[16]; i16 norm[16]; u16 tableU16[64]; FseCSym tt[16]; u8 tsym[64]; u32 cumul[18]; };

// Encoding tables (state table + per-symbol transforms).  tableU16 needs 2^log entries.
NAF_HD void fse_build_ctable(const i16 *norm, u32 maxsym, u32 log, u16 *tableU16, FseCSym *tt, u8 *tsym, u32 *cumul)
{
    u32 size = 1u << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    u32 high = size - 1;
    cumul[0] = 0;
    for (u32 u = 1; u <= maxsym + 1; u++) {
        if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; tsym[high--] = (u8)(u - 1); }
        else cumul[u] = cumul[u - 1] + (u32)norm[u - 1];
    }
    u32 pos = 0;
    for (u32 s = 0; s <= maxsym; s++)
        for (i32 i = 0; i < norm[s]; i++) { tsym[pos] = (u8)s; do { pos = (pos + step) & mask; } while (pos > high); }
    for (u32 u = 0; u < size; u++) { u32 s = tsym[u]; tableU16[cumul[s]++] = (u16)(size + u); }
    i32 total = 0;
    for (u32 s = 0; s <= maxsym; s++) {
        i32 nc = norm[s];
        if (nc == 0) { tt[s].deltaNbBits = (i32)(((log + 1) << 16) - (1u << log)); tt[s].deltaFindState = 0; }
        else if (nc == -1 || nc == 1) { tt[s].deltaNbBits = (i32)((log << 16) - (1u << log)); tt[s].deltaFindState = total - 1; total++; }
        else {
            u32 maxBitsOut = log - (u32)hibit32((u32)nc - 1);
            u32 minStatePlus = (u32)nc << maxBitsOut;
            tt[s].deltaNbBits = (i32)((maxBitsOut << 16) - minStatePlus);
            tt[s].deltaFindState = total - nc;
            total += nc;
        }
    }
}

// FSE encoder state steps (the encoder mirror of the decoder's base/nbits cells)
NAF_HD u32 fse_cinit(const u16 *tableU16, const FseCSym *tt, u32 sym)
{
    u32 nbBitsOut = (u32)((tt[sym].deltaNbBits + (1 << 15)) >> 16);
    i32 v = (i32)(nbBitsOut << 16) - tt[sym].deltaNbBits;
    return tableU16[(v >> nbBitsOut) + tt[sym].deltaFindState];
}
NAF_HD void fse_cencode(BitW &b, u32 &st, const u16 *tableU16, const FseCSym *tt, u32 sym)
{
    u32 nbBitsOut = (u32)(((i32)st + tt[sym].deltaNbBits) >> 16);
    bitw_add(b, st, nbBitsOut); bitw_flush(b);
    st = tableU16[((i32)st >> nbBitsOut) + tt[sym].deltaFindState];
}
// the same without the flush (the caller flushes 32 bits at a time)
NAF_HD void fse_cencode_nf(BitW &b, u32 &st, const u16 *tableU16, const FseCSym *tt, u32 sym)
{
    u32 nbBitsOut = (u32)(((i32)st + tt[sym].deltaNbBits) >> 16);
    bitw_add(b, st, nbBitsOut);
    st = tableU16[((i32)st >> nbBitsOut) + tt[sym].deltaFindState];
}

// FSE-compress `n` weights (values 0..maxsym) with two interleaved states.  Returns bytes written, 0 on failure.
NAF_HD u32 fse_compress_weights(u8 *out, u32 cap, const u8 *w, u32 n, FseWS &ws)
{
    if (n < 2) return 0;
    u32 *cnt = ws.cnt; i16 *norm = ws.norm; u16 *tableU16 = ws.tableU16; FseCSym *tt = ws.tt;
    for (u32 i = 0; i < 16; i++) cnt[i] = 0;
    u32 maxsym = 0, maxcnt = 0;
    for (u32 i = 0; i < n; i++) { cnt[w[i]]++; if (w[i] > maxsym) maxsym = w[i]; }
    for (u32 s = 0; s <= maxsym; s++) if (cnt[s] > maxcnt) maxcnt = cnt[s];
    if (maxcnt == n || maxcnt == 1) return 0;                  // single symbol / nothing to gain
    u32 log = n > 24 ? 6 : 5;
    if (!fse_normalize(cnt, maxsym, n, log, norm)) return 0;
    if (cap < 64) return 0;
    u32 hdr = fse_write_ncount(out, norm, maxsym, log);
    fse_build_ctable(norm, maxsym, log, tableU16, tt, ws.tsym, ws.cumul);
    BitW b; bitw_init(b, out + hdr);
    u32 st1, st2;
    auto init_state = [&](u32 sym) -> u32 {
        u32 nbBitsOut = (u32)((tt[sym].deltaNbBits + (1 << 15)) >> 16);
        i32 v = (i32)(nbBitsOut << 16) - tt[sym].deltaNbBits;
        return tableU16[(v >> nbBitsOut) + tt[sym].deltaFindState];
    };
    auto encode = [&](u32 &st, u32 sym) {
        u32 nbBitsOut = (u32)(((i32)st + tt[sym].deltaNbBits) >> 16);
        bitw_add(b, st, nbBitsOut); bitw_flush(b);
        st = tableU16[((i32)st >> nbBitsOut) + tt[sym].deltaFindState];
    };
    const u8 *ip = w + n;
    if (n & 1) { st1 = init_state(*--ip); st2 = init_state(*--ip); encode(st1, *--ip); }
    else { st2 = init_state(*--ip); st1 = init_state(*--ip); }
    while (ip > w) { encode(st2, *--ip); if (ip > w) encode(st1, *--ip); }
    bitw_add(b, st2, log); bitw_flush(b);
    bitw_add(b, st1, log); bitw_flush(b);
    u8 *end = bitw_close(b);
    u32 total = (u32)(end - out);
    if (total > cap) return 0;
    return total;
}

// Huffman tree description (4.2.1): FSE-compressed weights when smaller (mandatory above 128 weights),
// else direct 4-bit weights.  len[] = code lengths, log = max length.  Returns bytes, 0 = not representable.
// w[0..n) = weights of symbols 0..n-1 (n = index of the last present symbol, whose weight is implied).
// try_fse = false: direct weights whenever they are representable (n <= 128) -- a dozen bytes per block more, none of the
// serial FSE coding.
NAF_HD u32 huf_write_tree_w(u8 *out, const u8 *w, u32 n, u8 *tmp /*160*/, FseWS &ws, bool try_fse = true)
{
    if (n == 0) return 0;
    u32 fs = (try_fse || n > 128) ? fse_compress_weights(tmp, 160, w, n, ws) : 0;
    if (fs > 1 && fs < 128 && fs < (n + 1) / 2 + 0u + 1) { out[0] = (u8)fs; for (u32 i = 0; i < fs; i++) out[1 + i] = tmp[i]; return 1 + fs; }
    if (n > 128) return 0;
    out[0] = (u8)(127 + n);
    for (u32 i = 0; i < n; i += 2) out[1 + i / 2] = (u8)((w[i] << 4) | (i + 1 < n ? w[i + 1] : 0));
    return 1 + (n + 1) / 2;
}
NAF_HD u32 huf_write_tree(u8 *out, const u8 *len, u32 log)
{
    u8 w[256]; u32 last = 0;
    for (u32 s = 0; s < 256; s++) { w[s] = len[s] ? (u8)(log + 1 - len[s]) : 0; if (len[s]) last = s; }
    u8 tmp[160]; FseWS ws;
    return huf_write_tree_w(out, w, last, tmp, ws);            // weights for symbols 0..last-1; the last is implied
}

// Size in bytes of one Huffman stream holding the given per-symbol counts.
NAF_HD u32 huf_stream_bytes(const u32 *cnt, const u8 *len)
{
    u64 bits = 0;
    for (u32 s = 0; s < 256; s++) bits += (u64)cnt[s] * len[s];
    return (u32)((bits + 1 + 7) / 8);                          // + final marker bit
}

// Encode src[0..n) into one Huffman stream (4.2.2: written forward, so that the LAST symbol is read first).
// codes: code | len << 16 per symbol.  Input is consumed 8 bytes per load (walking down), output leaves as 8-byte
// words -- byte-granular stores from 600 k lanes cost 12x the algorithmic HBM write traffic.  Returns bytes written.
template <typename TabPtr>
NAF_HD u32 huf_encode_stream(u8 *out, const u8 *src, u32 n, TabPtr codes)
{
    u64 acc = 0; u32 nb = 0; u8 *p = out;
    u32 i = n;
    while (i >= 8) {                                           // symbols i-1 .. i-8, highest index first
        i -= 8;
        u64 w = ld64(src + i);
#pragma unroll
        for (int k = 7; k >= 0; k--) {
            u32 e = codes[(u32)(w >> (8 * k)) & 0xFF];
            u32 len = e >> 16; u64 v = (u64)(e & 0xFFFF);
            acc |= v << nb;                                    // nb < 64 here; bits that do not fit are re-added after the flush
            if (nb + len >= 64) { st64(p, acc); p += 8; acc = nb ? (v >> (64 - nb)) : 0; nb = nb + len - 64; }
            else nb += len;
        }
    }
    while (i-- > 0) {
        u32 e = codes[src[i]];
        u32 len = e >> 16; u64 v = (u64)(e & 0xFFFF);
        acc |= v << nb;
        if (nb + len >= 64) { st64(p, acc); p += 8; acc = nb ? (v >> (64 - nb)) : 0; nb = nb + len - 64; }
        else nb += len;
    }
    acc |= 1ull << nb; nb++;                                   // final marker bit (nb <= 63 before, so it fits)
    while (nb > 0) { *p++ = (u8)acc; acc >>= 8; nb = nb > 8 ? nb - 8 : 0; }
    return (u32)(p - out);
}

// ---- block planning ------------------------------------------------------------------------------------------------
enum { ZK_RAW = 0, ZK_RLE = 1, ZK_HUF = 2 };
struct ZEncPlan {
    u32 n;              // regenerated size of the block
    u32 csize;          // bytes of the block INCLUDING its 3-byte header
    u32 ssz[4];         // Huffman stream sizes
    u16 tree_bytes;     // Huffman tree description size
    u8  kind, log, lhdr;// ZK_*, table log, literals-section header size (3/4/5)
    u8  pad;
    u8  frame;          // 1: coded with the FRAME's code (zstd_enc.hip: frame tree); tree_bytes == 0 then means a treeless literals section
    u8  pad2;
};

NAF_HD void zenc_plan_finish(ZEncPlan &p, u32 n, u32 log, u32 tb, u32 min_gain = 0);
// hist[4][256] = byte counts of the four stream quarters of the block.  Fills plan, len[256], tree[<=160].
NAF_HD void zenc_plan_block(const u32 *hist, u32 n, ZEncPlan &p, u8 *len, u8 *tree)
{
    p.n = n; p.kind = ZK_RAW; p.csize = 3 + n; p.log = 0; p.tree_bytes = 0; p.lhdr = 0; p.frame = 0; p.pad2 = 0;
    if (n == 0) return;
    u32 tot[256]; u32 distinct = 0, only = 0;
    for (u32 s = 0; s < 256; s++) { tot[s] = hist[s] + hist[256 + s] + hist[512 + s] + hist[768 + s]; if (tot[s]) { distinct++; only = s; } }
    if (distinct == 1) { p.kind = ZK_RLE; p.csize = 4; (void)only; return; }
    if (n < 64) return;
    u32 log = huf_build_lengths(tot, len);
    if (!log) return;
    u32 tb = huf_write_tree(tree, len, log);
    if (!tb) return;
    for (u32 k = 0; k < 4; k++) p.ssz[k] = huf_stream_bytes(hist + 256 * k, len);
    zenc_plan_finish(p, n, log, tb);
}
// last step of the plan: p.ssz[] hold the four stream sizes; decides Huffman vs Raw
// min_gain: 256ths of the block that entropy coding has to save before it is used (0: any gain; the mask stream asks for an eighth --
// its units are close to uniform bytes, and a Raw block is a copy for every decoder where a Huffman stream is one lane's serial walk)
NAF_HD void zenc_plan_finish(ZEncPlan &p, u32 n, u32 log, u32 tb, u32 min_gain)
{
    u32 body = tb + 6;
    for (u32 k = 0; k < 4; k++) { body += p.ssz[k]; if (p.ssz[k] > 0xFFFF) return; }
    u32 lhdr = (n < 1024 && body < 1024) ? 3 : ((n < 16384 && body < 16384) ? 4 : 5);
    if (body >= (1u << 18)) return;
    u32 csize = 3 + lhdr + body + 1;                           // + sequences header (0 sequences)
    if (csize + (u32)(((u64)n * min_gain) >> 8) >= 3 + n) return;   // entropy coding does not pay (enough): Raw block
    p.kind = ZK_HUF; p.csize = csize; p.log = (u8)log; p.tree_bytes = (u16)tb; p.lhdr = (u8)lhdr;
}

// The plan of a block coded with the frame's code once it is known whether the block has to carry the tree (tb bytes) or not (0):
// Huffman whatever it costs -- its neighbours' sections were planned on this block being a Huffman block.
NAF_HD void zenc_plan_force(ZEncPlan &p, u32 tb)
{
    u32 body = tb + 6;
    for (u32 k = 0; k < 4; k++) body += p.ssz[k];
    const u32 lhdr = (p.n < 1024 && body < 1024) ? 3 : ((p.n < 16384 && body < 16384) ? 4 : 5);
    p.kind = ZK_HUF; p.csize = 3 + lhdr + body + 1; p.tree_bytes = (u16)tb; p.lhdr = (u8)lhdr;
}

// Huffman literals section header + tree + jump table at out[0..); returns the offset of the first stream.
// body = tree + 6 + streams (Compressed_Size), p.n = Regenerated_Size.
NAF_HD u32 zenc_write_huf_lit_prefix(u8 *out, const ZEncPlan &p, const u8 *tree)
{
    u32 body = p.csize - 3 - p.lhdr - 1, pos = 0;
    const u32 ty = (p.frame && p.tree_bytes == 0) ? 3u : 2u;     // Treeless_Literals_Block (3.1.1.3.1.1): the tree of the last block that carried one
    if (p.lhdr == 3) { u32 h = ty | (1u << 2) | (p.n << 4) | (body << 14); out[pos] = (u8)h; out[pos + 1] = (u8)(h >> 8); out[pos + 2] = (u8)(h >> 16); }
    else if (p.lhdr == 4) { u32 h = ty | (2u << 2) | (p.n << 4) | (body << 18); st32(out + pos, h); }
    else { u64 h = ty | (3u << 2) | ((u64)p.n << 4) | ((u64)body << 22); st32(out + pos, (u32)h); out[pos + 4] = (u8)(h >> 32); }
    pos += p.lhdr;
    for (u32 i = 0; i < p.tree_bytes; i++) out[pos + i] = tree[i];
    pos += p.tree_bytes;
    out[pos] = (u8)p.ssz[0]; out[pos + 1] = (u8)(p.ssz[0] >> 8);
    out[pos + 2] = (u8)p.ssz[1]; out[pos + 3] = (u8)(p.ssz[1] >> 8);
    out[pos + 4] = (u8)p.ssz[2]; out[pos + 5] = (u8)(p.ssz[2] >> 8);
    return pos + 6;
}
NAF_HD void zenc_write_block_header(u8 *out, u32 type, u32 bsize, bool last)
{
    u32 bh = (last ? 1u : 0u) | (type << 1) | (bsize << 3);
    out[0] = (u8)bh; out[1] = (u8)(bh >> 8); out[2] = (u8)(bh >> 16);
}
// Block header + literals header + tree + jump table.  Returns the offset of the first Huffman stream.
NAF_HD u32 zenc_write_block_prefix(u8 *out, const ZEncPlan &p, const u8 *tree, bool last, u8 rle_byte)
{
    u32 type = p.kind == ZK_HUF ? 2 : p.kind;
    zenc_write_block_header(out, type, p.kind == ZK_HUF ? p.csize - 3 : p.n, last);
    if (p.kind == ZK_RLE) { out[3] = rle_byte; return 4; }
    if (p.kind == ZK_RAW) return 3;
    return 3 + zenc_write_huf_lit_prefix(out + 3, p, tree);
}

// ---- sequences section (3.1.1.3.2), predefined distributions only -------------------------------------------------------------
// The LZ stage of this encoder finds matches inside a block only and codes every offset as a new offset (value = offset + 3),
// so blocks stay independent of each other.  LL / OF / ML codes use the predefined FSE distributions (mode byte 0): no table
// descriptions to build or transmit.
struct SeqCTab { const u16 *tableU16; const FseCSym *tt; u32 log; };
NAF_HD u32 zenc_ll_code(u32 ll) { if (ll < 16) return ll; u32 c = 16; while (c < 35 && ll_base(c + 1) <= ll) c++; return c; }
NAF_HD u32 zenc_ml_code(u32 ml) { if (ml < 35) return ml - 3; u32 c = 32; while (c < 52 && ml_base(c + 1) <= ml) c++; return c; }

// Encoding tables of the three predefined distributions: tableU16[64 + 32 + 64], tt[36 + 29 + 53].
struct SeqCTabs { u16 tableU16[160]; FseCSym tt[118]; };
NAF_HD void zenc_build_predefined(SeqCTabs &T)
{
    const i16 LL[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
    const i16 OF[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };
    const i16 ML[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
    u8 tsym[64]; u32 cumul[56];
    fse_build_ctable(LL, 35, 6, T.tableU16, T.tt, tsym, cumul);
    fse_build_ctable(OF, 28, 5, T.tableU16 + 64, T.tt + 36, tsym, cumul);
    fse_build_ctable(ML, 52, 6, T.tableU16 + 96, T.tt + 65, tsym, cumul);
}
NAF_HD void zenc_seq_ctabs(const SeqCTabs &T, SeqCTab ct[3])
{
    ct[0].tableU16 = T.tableU16; ct[0].tt = T.tt; ct[0].log = 6;
    ct[1].tableU16 = T.tableU16 + 64; ct[1].tt = T.tt + 36; ct[1].log = 5;
    ct[2].tableU16 = T.tableU16 + 96; ct[2].tt = T.tt + 65; ct[2].log = 6;
}

// Sequences_Section for nseq >= 1 sequences (ll = literals before the match, ml = match length >= 3, of = distance >= 1).
// Written forward; the decoder reads it from the end, so the LAST sequence is coded first.  Returns bytes, 0 if cap is short.
NAF_HD u32 zenc_write_sequences(u8 *out, u32 cap, const u16 *ll, const u16 *ml, const u16 *of, u32 nseq, const SeqCTab ct[3])
{
    if (cap < 64) return 0;
    u32 pos = 0;
    if (nseq < 128) out[pos++] = (u8)nseq;
    else if (nseq < 0x7F00) { out[pos++] = (u8)((nseq >> 8) + 128); out[pos++] = (u8)nseq; }
    else { out[pos++] = 255; out[pos++] = (u8)(nseq - 0x7F00); out[pos++] = (u8)((nseq - 0x7F00) >> 8); }
    out[pos++] = 0;                                            // Symbol_Compression_Modes: predefined x3
    BitW b; bitw_init(b, out + pos);
    // The three arrays are read eight entries (16 bytes) at a time, walking down: one lane per block reads them from HBM, and an
    // access per sequence and array was a memory latency each.  The arrays must be readable up to the next multiple of 8 entries.
    u16 cl[8], cm[8], co[8];
    u32 i = nseq - 1;
    memcpy(cl, ll + (i & ~7u), 16); memcpy(cm, ml + (i & ~7u), 16); memcpy(co, of + (i & ~7u), 16);
    u32 vl = cl[i & 7], vm = cm[i & 7];
    u32 llc = zenc_ll_code(vl), mlc = zenc_ml_code(vm), ofv = (u32)co[i & 7] + 3, ofc = (u32)hibit32(ofv);
    u32 sML = fse_cinit(ct[2].tableU16, ct[2].tt, mlc), sOF = fse_cinit(ct[1].tableU16, ct[1].tt, ofc), sLL = fse_cinit(ct[0].tableU16, ct[0].tt, llc);
    bitw_add(b, vl - ll_base(llc), ll_bits(llc)); bitw_add(b, vm - ml_base(mlc), ml_bits(mlc)); bitw_flush32(b);
    bitw_add(b, ofv - (1u << ofc), ofc); bitw_flush32(b);
    while (i-- > 0) {
        if ((u32)(b.p - out) + 32 > cap) return 0;                // a sequence adds at most 74 bits
        if ((i & 7) == 7) { memcpy(cl, ll + i - 7, 16); memcpy(cm, ml + i - 7, 16); memcpy(co, of + i - 7, 16); }
        vl = cl[i & 7]; vm = cm[i & 7];
        llc = zenc_ll_code(vl); mlc = zenc_ml_code(vm); ofv = (u32)co[i & 7] + 3; ofc = (u32)hibit32(ofv);
        // fewer than 32 bits are pending at each step: + 26 (states) / + 32 (ll, ml extra bits) / + 16 (offset extra bits)
        fse_cencode_nf(b, sOF, ct[1].tableU16, ct[1].tt, ofc);
        fse_cencode_nf(b, sML, ct[2].tableU16, ct[2].tt, mlc);
        fse_cencode_nf(b, sLL, ct[0].tableU16, ct[0].tt, llc);
        bitw_flush32(b);
        bitw_add(b, vl - ll_base(llc), ll_bits(llc)); bitw_add(b, vm - ml_base(mlc), ml_bits(mlc)); bitw_flush32(b);
        bitw_add(b, ofv - (1u << ofc), ofc); bitw_flush32(b);
    }
    bitw_flush(b);
    bitw_add(b, sML, ct[2].log); bitw_flush(b);
    bitw_add(b, sOF, ct[1].log); bitw_flush(b);
    bitw_add(b, sLL, ct[0].log); bitw_flush(b);
    u8 *end = bitw_close(b);
    return (u32)(end - out);
}

// ---- sequences with repeat offsets and per-block FSE tables (levels >= 2, --long) -----------------------------------------------
// Repeat offsets (3.1.1.5) as far as ONE block can know them: a block is coded without knowing the three offsets the decoder holds
// when it starts (blocks are produced independently), so a slot is 0 = unknown until a sequence of this block has defined it; an
// unknown slot is never used as a repeat code.  zenc_offset_value returns the Offset_Value of the sequence (1..3: repeat code,
// else distance + 3) and applies the decoder's update.
struct RepState { u32 r[3]; };
NAF_HD u32 zenc_offset_value(RepState &R, u32 d, u32 ll)
{
    u32 v = d + 3;
    if (ll) { if (d == R.r[0]) v = 1; else if (d == R.r[1]) v = 2; else if (d == R.r[2]) v = 3; }
    else { if (d == R.r[1]) v = 1; else if (d == R.r[2]) v = 2; else if (R.r[0] > 1 && d == R.r[0] - 1) v = 3; }
    if (v > 3) { R.r[2] = R.r[1]; R.r[1] = R.r[0]; R.r[0] = d; }
    else {
        const u32 idx = v - 1 + (ll ? 0u : 1u);
        if (idx == 1) { const u32 t = R.r[1]; R.r[1] = R.r[0]; R.r[0] = t; }
        else if (idx >= 2) { const u32 t = idx == 2 ? R.r[2] : R.r[0] - 1; R.r[2] = R.r[1]; R.r[1] = R.r[0]; R.r[0] = t; }
    }
    return v;
}

// Workspace of the sequences encoder of one block (LDS on the GPU).  Table sizes: LL 2^9, OF 2^8, ML 2^9 at most (3.1.1.3.2.1.1).
#define ZSEQ_LL_LOG 9
#define ZSEQ_OF_LOG 8
#define ZSEQ_ML_LOG 9
struct SeqWS {
    u32 cnt[3][64];
    i16 norm[64];
    u16 tableU16[(1 << ZSEQ_LL_LOG) + (1 << ZSEQ_OF_LOG) + (1 << ZSEQ_ML_LOG)];
    FseCSym tt[36 + 32 + 53];
    u8 tsym[1 << ZSEQ_LL_LOG]; u32 cumul[56];
};
// cost in 1/256 bit of coding `cnt` symbols with probabilities norm / 2^log (norm -1 counts as 1); ~0u when a symbol has no code
NAF_HD u32 zenc_log2_256(u32 v)                                 // 256 * log2(v), v >= 1, piecewise linear between powers of two
{
    const u32 h = (u32)hibit32(v);
    const u32 frac = h >= 8 ? (v >> (h - 8)) - 256 : (v << (8 - h)) - 256;       // 0..255
    return h * 256 + frac;
}
NAF_HD u64 zenc_fse_cost(const u32 *cnt, u32 maxsym, const i16 *norm, u32 log)
{
    u64 c = 0;
    for (u32 s = 0; s <= maxsym; s++) {
        if (!cnt[s]) continue;
        const i32 nv = norm[s];
        if (nv == 0) return ~0ull;
        c += (u64)cnt[s] * (log * 256 - zenc_log2_256((u32)(nv < 0 ? 1 : nv)));
    }
    return c;
}
// One of the three code tables of a block: picks predefined / RLE / FSE_Compressed (3.1.1.3.2.1), writes the table description
// for the last one and builds the encoding table.  Returns the mode (0, 1, 2); *hdr_bytes = bytes appended at `out`.
NAF_HD u32 zenc_seq_table(u8 *out, u32 *hdr_bytes, const u32 *cnt, u32 alphabet, u32 nseq, u32 max_log, const i16 *pre_norm, u32 pre_n, u32 pre_log,
                          const SeqCTab &pre, i16 *norm, u16 *tableU16, FseCSym *tt, u8 *tsym, u32 *cumul, SeqCTab &ct)
{
    *hdr_bytes = 0;
    u32 maxsym = 0, present = 0, only = 0;
    for (u32 s = 0; s < alphabet; s++) if (cnt[s]) { maxsym = s; present++; only = s; }
    if (present == 1) { out[0] = (u8)only; *hdr_bytes = 1; ct.tableU16 = nullptr; ct.tt = nullptr; ct.log = 0; return 1; }
    // predefined: usable when every symbol has a code in it
    u64 cost_pre = ~0ull;
    if (maxsym < pre_n) cost_pre = zenc_fse_cost(cnt, maxsym, pre_norm, pre_log);
    // FSE_Compressed: accuracy from the number of sequences (more states than sequences buy nothing)
    u32 log = (u32)hibit32(nseq) + 1; if (log < 5) log = 5; if (log > max_log) log = max_log;
    while ((1u << log) < present) log++;
    u64 cost_fse = ~0ull; u32 nc = 0;
    if (log <= max_log && fse_normalize(cnt, maxsym, nseq, log, norm)) {
        nc = fse_write_ncount(out, norm, maxsym, log);
        cost_fse = zenc_fse_cost(cnt, maxsym, norm, log) + (u64)nc * 8 * 256;
    }
    if (cost_fse < cost_pre) {
        fse_build_ctable(norm, maxsym, log, tableU16, tt, tsym, cumul);
        *hdr_bytes = nc; ct.tableU16 = tableU16; ct.tt = tt; ct.log = log;
        return 2;
    }
    if (cost_pre == ~0ull) return 3;                            // no way to code this block's sequences: the caller keeps it literal-only
    ct = pre;
    return 0;
}

// Sequences_Section (3.1.1.3.2) for nseq >= 1 sequences with offset VALUES (repeat codes included) and the code tables chosen per
// block.  ll / ml as in zenc_write_sequences; ofv[i] = Offset_Value >= 1.  Returns bytes written, 0 when cap is short or a table
// could not be made.
NAF_HD u32 zenc_write_sequences_x(u8 *out, u32 cap, const u16 *ll, const u16 *ml, const u32 *ofv, u32 nseq, const SeqCTabs &P, SeqWS &ws)
{
    if (cap < 256) return 0;
    const i16 LLn[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
    const i16 OFn[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };
    const i16 MLn[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
    u32 pos = 0;
    if (nseq < 128) out[pos++] = (u8)nseq;
    else if (nseq < 0x7F00) { out[pos++] = (u8)((nseq >> 8) + 128); out[pos++] = (u8)nseq; }
    else { out[pos++] = 255; out[pos++] = (u8)(nseq - 0x7F00); out[pos++] = (u8)((nseq - 0x7F00) >> 8); }
    for (u32 t = 0; t < 3; t++) for (u32 s = 0; s < 64; s++) ws.cnt[t][s] = 0;
    for (u32 i = 0; i < nseq; i++) { ws.cnt[0][zenc_ll_code(ll[i])]++; ws.cnt[1][(u32)hibit32(ofv[i])]++; ws.cnt[2][zenc_ml_code(ml[i])]++; }
    SeqCTab pre[3]; zenc_seq_ctabs(P, pre);
    SeqCTab ct[3];
    const u32 modes_at = pos++;
    u32 hb, mode[3];
    mode[0] = zenc_seq_table(out + pos, &hb, ws.cnt[0], 36, nseq, ZSEQ_LL_LOG, LLn, 36, 6, pre[0], ws.norm, ws.tableU16, ws.tt, ws.tsym, ws.cumul, ct[0]); pos += hb;
    mode[1] = zenc_seq_table(out + pos, &hb, ws.cnt[1], 32, nseq, ZSEQ_OF_LOG, OFn, 29, 5, pre[1], ws.norm, ws.tableU16 + (1 << ZSEQ_LL_LOG), ws.tt + 36, ws.tsym, ws.cumul, ct[1]); pos += hb;
    mode[2] = zenc_seq_table(out + pos, &hb, ws.cnt[2], 53, nseq, ZSEQ_ML_LOG, MLn, 53, 6, pre[2], ws.norm, ws.tableU16 + (1 << ZSEQ_LL_LOG) + (1 << ZSEQ_OF_LOG), ws.tt + 68, ws.tsym, ws.cumul, ct[2]); pos += hb;
    if (mode[0] == 3 || mode[1] == 3 || mode[2] == 3) return 0;
    out[modes_at] = (u8)((mode[0] << 6) | (mode[1] << 4) | (mode[2] << 2));
    BitW b; bitw_init(b, out + pos);
    // an RLE table has one state and no bits (Accuracy_Log 0)
    auto init = [&](const SeqCTab &c, u32 sym) -> u32 { return c.tableU16 ? fse_cinit(c.tableU16, c.tt, sym) : 0u; };
    auto step = [&](const SeqCTab &c, u32 &st, u32 sym) { if (c.tableU16) fse_cencode_nf(b, st, c.tableU16, c.tt, sym); };
    u32 i = nseq - 1;
    u32 vl = ll[i], vm = ml[i], vo = ofv[i];
    u32 llc = zenc_ll_code(vl), mlc = zenc_ml_code(vm), ofc = (u32)hibit32(vo);
    u32 sML = init(ct[2], mlc), sOF = init(ct[1], ofc), sLL = init(ct[0], llc);
    bitw_add(b, vl - ll_base(llc), ll_bits(llc)); bitw_add(b, vm - ml_base(mlc), ml_bits(mlc)); bitw_flush32(b);
    bitw_add(b, vo - (1u << ofc), ofc); bitw_flush32(b);
    while (i-- > 0) {
        if ((u32)(b.p - out) + 40 > cap) return 0;                // a sequence adds at most 26 + 32 + 31 bits
        vl = ll[i]; vm = ml[i]; vo = ofv[i];
        llc = zenc_ll_code(vl); mlc = zenc_ml_code(vm); ofc = (u32)hibit32(vo);
        // fewer than 32 bits are pending at each flush point: + 26 (states) / + 32 (ll, ml extra bits) / + 31 (offset extra bits)
        step(ct[1], sOF, ofc); step(ct[2], sML, mlc); step(ct[0], sLL, llc);
        bitw_flush32(b);
        bitw_add(b, vl - ll_base(llc), ll_bits(llc)); bitw_add(b, vm - ml_base(mlc), ml_bits(mlc)); bitw_flush32(b);
        bitw_add(b, vo - (1u << ofc), ofc); bitw_flush32(b);
    }
    bitw_flush(b);
    bitw_add(b, sML, ct[2].log); bitw_flush(b);
    bitw_add(b, sOF, ct[1].log); bitw_flush(b);
    bitw_add(b, sLL, ct[0].log); bitw_flush(b);
    u8 *end = bitw_close(b);
    return (u32)(end - out);
}

// Raw / RLE literals section header (3.1.1.3.1.1): returns header bytes.
NAF_HD u32 zenc_lit_header_raw(u8 *out, u32 type, u32 regen)
{
    if (regen < 32) { out[0] = (u8)(type | (regen << 3)); return 1; }
    if (regen < 4096) { u32 h = type | (1u << 2) | (regen << 4); out[0] = (u8)h; out[1] = (u8)(h >> 8); return 2; }
    u32 h = type | (3u << 2) | (regen << 4); out[0] = (u8)h; out[1] = (u8)(h >> 8); out[2] = (u8)(h >> 16); return 3;
}
