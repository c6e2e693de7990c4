// zstd_dec.hip -- zstd frame decoder for gfx950 (RFC 8878; replaces libzstd behind
// unnaf/src/input.c:155,183,212,230 one-shot sections and input.c:262-434 streamed sequence/quality).
//
// Pipeline per frame (all data stays in HBM):
//   k_scan_blocks    one lane walks the 3-byte block headers -> ZBlock[]           (serial by format)
//   k_parse_blocks   one lane per block: literals / sequences section headers
//   max-scans        which earlier block owns the Huffman / LL / OF / ML table in force (treeless, repeat)
//   k_build_huf/fse  one lane per defining block builds decoding tables into a pool
//   k_decode_seq     one lane per block with sequences: FSE decode -> (ll, ml, offset) arrays,
//                    repeat offsets kept symbolic against the block-entry state
//   k_rep_chain      one wave composes the repeat-offset state across blocks
//   scan             regenerated sizes -> output offsets
//   k_huf_literals   ONE LANE PER HUFFMAN STREAM (4 per block, 16 blocks per wave), tables in LDS
//   k_copy_fill      raw / RLE blocks and raw / RLE literals, one workgroup per block
//   k_lz_prep/_deps/_exec   sequences executed as dataflow: literals and positions per block, per match the earlier matches its
//                    source was written by, then units of 1024 sequences sweep over whatever has its sources done
//                    (k_exec_seq_lds: blocks of at most 16 KiB assembled in LDS, in block order; k_exec_batch / k_exec_seq: cross-checks)
#include "ctx.h"
#include "wgscan.h"
#include "zstd_dec_core.h"
#include "emit_core.h"

struct ZStat {                 // device-side counters read back by the host
    u32 nblk; u32 err; u64 end_off;
    u32 n_huf_def, n_seq_blk, max_huf_log, rep_slow;   // rep_slow bit 0: some block's exit repeat offsets depend on its entry state; bit 1: some sequence uses a repeat code
    u32 huf_pool_used, fse_pool_used;
    u64 total_seq, total_out;
    u32 ticket, n_plain_huf;     // n_plain_huf: compressed blocks with Huffman literals and no sequences
    u32 max_seq_regen, n_flat;    // largest regenerated size among the blocks that have sequences; table-defining blocks whose tree is flat
    u32 n_huf_distinct, n_huf_built;   // tree descriptions that differ from their predecessor's (k_huf_dedup); tables actually built
    u32 max_lit_regen, n_huf_pending;  // largest literals section of a Huffman-coded block (k_huf_par sizes its parts by it); trees left without a table by k_build_huf's first phase
    u32 last_raw, flat_main_inv;       // size of the frame's last block when it is a Raw or (bit 31) RLE one, else 0; 0xFFFFFFFF - index of the FIRST block that defines a flat 4-bit tree (0: none)
    u32 n_exec_done, n_wave;           // blocks the sequence executors have published: a waiting block gives up only when this stands still; blocks k_decode_seq left to k_decode_seq_wave(2)
#ifdef NAF_EXEC_PROF
    unsigned long long prof[8];
#endif
};

static __device__ __forceinline__ void set_err(ZStat *st, u32 e) { if (e) atomicMax(&st->err, e); }
static __device__ __forceinline__ u64 shfl_u64(u64 v, int l) { u32 lo = __shfl((u32)v, l, 64), hi = __shfl((u32)(v >> 32), l, 64); return ((u64)hi << 32) | lo; }

#define SCAN_WIN (48u * 1024u)
// Serial walk of the block headers (frames too small for the parallel index, or not one well-formed chain).  The header chain
// is one dependent load per block, so the frame is pulled through LDS a 48 KiB window at a time (coalesced) and walked there.
// The window (dynamic LDS, `win` + 16 bytes) is sized to the frame: a 48 KiB allocation waits for a CU to drain when a big
// kernel of another stream fills the device, and the frames that come here are mostly a few hundred bytes.
__global__ __launch_bounds__(64) void k_scan_blocks(const u8 *src, u64 len, u64 first_off, ZBlock *blk, u32 cap, ZStat *st, u32 win)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    extern __shared__ __attribute__((aligned(16))) u8 buf[];
    __shared__ u64 s_pos; __shared__ u32 s_n, s_err, s_done;
    if (threadIdx.x == 0) { s_pos = first_off; s_n = 0; s_err = 0; s_done = 0; }
    __syncthreads();
    for (;;) {
        const u64 wlo = s_pos;
        const u32 wn = len - wlo < win ? (u32)(len > wlo ? len - wlo : 0) : win;
        for (u32 i = threadIdx.x * 16; i < wn; i += 64 * 16) {
            if (i + 16 <= wn) { uint4 v; memcpy(&v, src + wlo + i, 16); *(uint4 *)(buf + i) = v; }
            else for (u32 k = i; k < wn; k++) buf[k] = src[wlo + k];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            u64 pos = wlo; u32 n = s_n, err = 0; bool done = false;
            for (;;) {
                if (pos + 3 > len) { err = ZE_TRUNC; done = true; break; }
                if (pos + 3 > wlo + wn) break;                          // header outside the window: slide
                u32 o = (u32)(pos - wlo);
                u32 h = (u32)buf[o] | ((u32)buf[o + 1] << 8) | ((u32)buf[o + 2] << 16);
                u32 last = h & 1, type = (h >> 1) & 3, size = h >> 3;
                if (type == 3 || size > ZBLOCK_MAX) { err = ZE_CORRUPT; done = true; break; }
                u32 csize = type == BT_RLE ? 1 : size;
                if (pos + 3 + csize > len) { err = ZE_TRUNC; done = true; break; }
                if (n < cap) { ZBlock &b = blk[n]; b.src_off = pos + 3; b.bsize = size; b.btype = (u8)type; b.last = (u8)last; }
                n++; pos += 3 + csize;
                if (last) { done = true; break; }
            }
            s_pos = pos; s_n = n; s_err = err; s_done = done ? 1u : 0u;
        }
        __syncthreads();
        if (s_done) break;
    }
    if (threadIdx.x == 0) { st->nblk = s_n; st->end_off = s_pos; st->err = s_err; }
}

// ---- speculative parallel block index -----------------------------------------------------------------------
// The 3-byte block headers form a linked list that has to be walked serially by the format.  To index a
// multi-GB frame without a multi-ms serial walk:
//   1. k_spec_find   cut the frame into 1 MiB chunks; in the first 128 KiB + 3 of each chunk (a block start
//                    must lie there) test every byte as a candidate start and keep the smallest one whose
//                    chain survives SPEC_HOPS header validations (type 3 / size > 128 KiB kill ~95 % of
//                    random positions per hop; survivors are on, or quickly merge into, the true chain)
//   2. k_spec_land   walk each chunk's candidate chain to the first block start in the NEXT chunk
//                    ("landing"), once from the candidate and once from the previous chunk's landing
//   3. k_spec_resolve one wave runs over the chunks in order: the true start of chunk c+1 is F_c(true start
//                    of chunk c); when the speculated start of chunk c equals the true one, the precomputed
//                    landing is reused, otherwise the chunk is re-walked on the spot.  Starts are therefore
//                    EXACT by construction -- speculation only decides how much is reused.
//   4. k_spec_walk   count, then write, the ZBlock records of every chunk in parallel.
#define SPEC_CHUNK   (1024u * 1024u)
#define SPEC_CHUNK_SMALL (16u * 1024u)    // frames of up to 4 MiB: chunks of 256 B .. 16 KiB, every byte is a candidate
#define SPEC_WINDOW  (ZBLOCK_MAX + 4u)
#define SPEC_HOPS    10
#define SPEC_NONE    0xFFFFFFFFFFFFFFFFull
#define SPEC_END     0xFFFFFFFFFFFFFFFEull       // chain reached the last block of the frame

__device__ __forceinline__ bool spec_hdr_ok(u32 h, u64 pos, u64 len, u32 &adv)
{
    u32 type = (h >> 1) & 3, size = h >> 3;
    if (type == 3 || size > ZBLOCK_MAX) return false;
    adv = 3 + (type == BT_RLE ? 1 : size);
    return pos + adv <= len;
}
__device__ __forceinline__ bool spec_hop(const u8 *src, u64 len, u64 &pos, bool &last)
{
    if (pos + 3 > len) return false;
    u32 h = ld24(src + pos), adv;
    if (!spec_hdr_ok(h, pos, len, adv)) return false;
    last = h & 1; pos += adv;
    return true;
}
// Walk from `pos` to the first block start >= stop.  Returns SPEC_END after the last block, SPEC_NONE on an invalid header.
__device__ __forceinline__ u64 spec_land(const u8 *src, u64 len, u64 pos, u64 stop, u64 *end_off)
{
    bool last = false;
    while (pos < stop) {
        if (!spec_hop(src, len, pos, last)) return SPEC_NONE;
        if (last) { if (end_off) *end_off = pos; return SPEC_END; }
    }
    return pos;
}

// Candidates are searched in window bytes [w_lo, w_hi) of every chunk.  Two passes: the first 40 KiB (where the
// first block start of a chunk lies whenever blocks compress to under 40 KiB -- this build's 32 KiB blocks, and
// 128 KiB blocks at ratios above 3.2), then the rest of the 128 KiB window for the chunks still without a candidate.
#define SPEC_WINDOW0 (20u * 1024u)
#define SPEC_WINDOW1 (40u * 1024u)
__global__ __launch_bounds__(256) void k_spec_find(const u8 *src, u64 len, u32 nchunks, u64 *first, u32 w_lo, u32 w_hi, u32 chunk, const u32 *skip)
{
    if (skip && skip[1] == 1) return;                          // the stride index has the frame (k_stride_tail)
    const u32 tiles = (w_hi - w_lo + 4095) / 4096;
    u32 c = blockIdx.x / tiles + 1;                          // chunk 0 starts at the known first block
    if (c >= nchunks) return;
    if (w_lo && first[c] != SPEC_NONE) return;               // second pass: only chunks the first pass left empty
    u64 base = (u64)c * chunk;
    u32 k0 = w_lo + ((blockIdx.x % tiles) * 256 + threadIdx.x) * 16;
    if (k0 >= w_hi || base + k0 + 24 > len) {
        if (k0 >= w_hi || base + k0 + 3 > len) return;
    }
    // 16 candidate positions from 18 bytes held in registers
    u64 w0 = 0, w1 = 0, w2 = 0;
    if (base + k0 + 24 <= len) { w0 = ld64(src + base + k0); w1 = ld64(src + base + k0 + 8); w2 = ld64(src + base + k0 + 16); }
    else { u8 t[24]; for (int i = 0; i < 24; i++) t[i] = base + k0 + i < len ? src[base + k0 + i] : 0xFF; w0 = ld64(t); w1 = ld64(t + 8); w2 = ld64(t + 16); }
    for (u32 k = 0; k < 16; k++) {
        if (k0 + k >= w_hi) break;
        u32 sh = 8 * (k & 7);
        u64 lo = k < 8 ? w0 : w1, hi = k < 8 ? w1 : w2;
        u32 h = (u32)((sh ? (lo >> sh) | (hi << (64 - sh)) : lo) & 0xFFFFFF), adv;
        u64 pos = base + k0 + k;
        if (!spec_hdr_ok(h, pos, len, adv)) continue;
        // An empty Raw block that is not the last one is legal but nobody writes it, while runs of zero bytes (a direct Huffman
        // weight table whose first symbols do not occur) read as chains of exactly that: not taken as candidates.  (If a frame
        // really holds one, the chunks in front of it get no candidate and are re-walked by the resolve pass: still exact.)
        if ((h >> 1) == 0 && !(h & 1)) continue;
        bool last = h & 1, ok = true;
        u64 p = pos + adv;
        if (last) ok = p + 4 >= len;
        for (int hop = 1; ok && !last && hop < SPEC_HOPS; hop++) {
            if (p + 3 <= len && (ld24(src + p) >> 1) == 0 && !(ld24(src + p) & 1)) { ok = false; break; }
            if (!spec_hop(src, len, p, last)) ok = false;
            else if (last) ok = p + 4 >= len;                // a chain may only end at the end of the frame (+checksum)
        }
        if (ok) { atomicMin((unsigned long long *)&first[c], (unsigned long long)pos); break; }
    }
}

// land[c+1] = F_c(first[c]) ; first[0] is the known true start.
__global__ void k_spec_land(const u8 *src, u64 len, const u64 *first, u32 nchunks, u64 *land, u32 chunk, const u32 *skip)
{
    if (skip && skip[1] == 1) return;
    u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    u64 s = first[c];
    land[c + 1] = s == SPEC_NONE ? SPEC_NONE : spec_land(src, len, s, (u64)(c + 1) * chunk, nullptr);
    if (c == 0) land[0] = s;
}
// G[c] = F_c(land[c]) (reuses land[c+1] when the candidate already was the landing)
__global__ void k_spec_land2(const u8 *src, u64 len, const u64 *first, const u64 *land, u32 nchunks, u64 *G, u32 chunk, const u32 *skip)
{
    if (skip && skip[1] == 1) return;
    u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    u64 l = land[c];
    if (l == first[c]) G[c] = land[c + 1];
    else if (l >= SPEC_END) G[c] = SPEC_NONE;
    else G[c] = spec_land(src, len, l, (u64)(c + 1) * chunk, nullptr);
}
// One wave: exact chunk starts.  start[c] for c in [0, nchunks], start[nchunks] = SPEC_END when the frame is well formed.
__global__ void k_spec_resolve(const u8 *src, u64 len, const u64 *land, const u64 *G, u32 nchunks, u64 *start, ZStat *st, u32 chunk, const u32 *skip)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    if (skip && skip[1] == 1) return;
    int lane = threadIdx.x;
    u64 t = land[0];                                           // true start of chunk 0
    if (lane == 0) start[0] = t;
    for (u32 base = 0; base < nchunks; base += 64) {
        u32 c = base + lane;
        u64 lc = c < nchunks ? land[c] : SPEC_NONE, gc = c < nchunks ? G[c] : SPEC_NONE;
        // whole batch at once when the speculation holds: t is the speculated start of the batch's first chunk and every
        // chunk's landing is the speculated start of the next one -- then (induction) all of them are the true starts
        if (base + 64 < nchunks) {
            u64 ln = land[c + 1];
            bool ok = lc < SPEC_END && gc < SPEC_END && gc == ln;
            if (shfl_u64(lc, 0) == t && __all(ok)) {
                start[c + 1] = gc;
                t = shfl_u64(gc, 63);
                continue;
            }
        }
        for (int j = 0; j < 64 && base + j < nchunks; j++) {
            u64 lj = shfl_u64(lc, j), gj = shfl_u64(gc, j);
            u64 nxt;
            if (t >= SPEC_END) nxt = t;                        // past the end of the frame (or broken): propagate
            else if (lj == t) nxt = gj;
            else { u64 e = 0; nxt = spec_land(src, len, t, (u64)(base + j + 1) * chunk, &e); }   // speculation missed: re-walk (uniform)
            t = nxt;
            if (lane == 0) start[base + j + 1] = t;
        }
    }
    if (lane == 0 && t != SPEC_END) atomicMax(&st->err, (u32)ZE_CORRUPT);
}

// WRITE=false: count blocks per chunk; WRITE=true: emit ZBlock records at count[c] (exclusive-scanned).
template <bool WRITE>
__global__ void k_spec_walk(const u8 *src, u64 len, const u64 *start, u32 nchunks, u64 *count, ZBlock *blk, ZStat *st, const u32 *skip)
{
    if (skip && skip[1] == 1) return;
    u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    u64 pos = start[c], lim = start[c + 1];
    u64 n = 0, o = WRITE ? count[c] : 0;
    if (pos < SPEC_END) {
        bool last = false;
        while (pos < lim) {
            u64 p0 = pos; u32 h = ld24(src + pos);
            if (!spec_hop(src, len, pos, last)) { atomicMax(&st->err, (u32)ZE_CORRUPT); break; }
            if (WRITE) { ZBlock &b = blk[o + n]; b.src_off = p0 + 3; b.bsize = h >> 3; b.btype = (u8)((h >> 1) & 3); b.last = (u8)(h & 1); }
            n++;
            if (last) { if (!WRITE) st->end_off = pos; break; }
        }
    }
    if (!WRITE) count[c] = n;
}

// ---- stride index -------------------------------------------------------------------------------------------------------------
// A stream of packed bases coded with fixed-width codes (this build's flat blocks, §4.6 of DESIGN.md) is a string of blocks of ONE
// compressed size: block i starts at off0 + i S when the headers at off0, off0 + S, ... all repeat the first one (induction: a
// header gives the next start).  k_stride_probe tests every such position at once, k_stride_tail walks what lies behind the first
// mismatch (the shorter last block, the Raw block of an odd stream's last byte) -- a frame whose tail is longer than STRIDE_TAIL
// blocks is left to the speculative index, whose kernels return at once when res[1] == 1.  res[0] = blocks of the prefix.
#define STRIDE_TAIL 64u
__global__ void k_stride_probe(const u8 *src, u64 len, u64 off0, u32 S, u32 h0, u32 nmax, u32 *res)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nmax) return;
    const u64 pos = off0 + (u64)i * S;
    const bool ok = pos + S <= len && ld24(src + pos) == h0;
    const u64 bad = __ballot(!ok);
    if (bad && (threadIdx.x & 63) == (u32)(__ffsll((long long)bad) - 1)) atomicMin(&res[0], i);
}
__global__ void k_stride_tail(const u8 *src, u64 len, u64 off0, u32 S, u32 nmax, u32 *res, ZBlock *blk, ZStat *st)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    if (threadIdx.x || blockIdx.x) return;
    const u32 np = res[0] < nmax ? res[0] : nmax;
    res[0] = np;
    u64 pos = off0 + (u64)np * S; u32 n = 0; bool done = false;
    // (more bytes behind the prefix than STRIDE_TAIL blocks of the largest size hold: a frame of mixed blocks, not worth a walk of dependent loads)
    while (n < STRIDE_TAIL && len - pos <= (u64)STRIDE_TAIL * (ZBLOCK_MAX + 3u)) {
        if (pos + 3 > len) break;
        const u32 h = ld24(src + pos), last = h & 1, type = (h >> 1) & 3, size = h >> 3;
        if (type == 3 || size > ZBLOCK_MAX) break;
        const u32 csize = type == BT_RLE ? 1 : size;
        if (pos + 3 + csize > len) break;
        ZBlock &b = blk[np + n]; b.src_off = pos + 3; b.bsize = size; b.btype = (u8)type; b.last = (u8)last;
        n++; pos += 3 + csize;
        if (last) { done = true; break; }
    }
    res[1] = done ? 1u : 0u;
    if (done) { st->nblk = np + n; st->end_off = pos; }
    // (a prefix of equal blocks that ends far in front of the frame's end: the frame may be RUNS of equal blocks -- k_runs_*)
    else if (np >= 16 && n == 0) res[1] = 2u;
}
__global__ void k_stride_write(u64 off0, u32 S, u32 h0, const u32 *res, ZBlock *blk, const u32 *uni)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (res[1] != 1 || i >= res[0]) return;
    if (uni && uni[0] && !uni[1]) return;                        // a uniform flat frame (k_uni_head / k_uni_streams: UniInfo.ok, .bad): nobody reads the block table
    ZBlock &b = blk[i]; b.src_off = off0 + (u64)i * S + 3; b.bsize = h0 >> 3; b.btype = (u8)((h0 >> 1) & 3); b.last = 0;
}

__global__ void k_parse_blocks(const u8 *src, ZBlock *blk, u32 nblk, i32 *own_huf, i32 *own_ll, i32 *own_of, i32 *own_ml,
                               u64 *seq_cnt, u64 *sizes, ZStat *st)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblk) return;
    ZBlock b = blk[i];
    zstd_parse_block(src + b.src_off, b);
    b.huf_flat = 0; b.huf_tab = 0; b.huf_log = 0;                // (set by the table builders for the blocks that define a tree)
    blk[i] = b;
    set_err(st, b.err);
    bool comp = b.btype == BT_COMP && !b.err;
    own_huf[i] = (comp && b.lit_type == LIT_HUF) ? (i32)i : -1;
    bool sq = comp && b.nseq > 0;
    own_ll[i] = (sq && b.modes[0] != SM_REPEAT) ? (i32)i : -1;
    own_of[i] = (sq && b.modes[1] != SM_REPEAT) ? (i32)i : -1;
    own_ml[i] = (sq && b.modes[2] != SM_REPEAT) ? (i32)i : -1;
    seq_cnt[i] = sq ? (b.nseq + 3u) & ~3u : 0;                  // a block's sequences start at a multiple of four: k_decode_seq stores them 16 bytes at a time
    sizes[i] = b.regen;                                          // final unless the block has sequences (k_decode_seq then rewrites it)
    if (comp && b.lit_type == LIT_HUF) atomicAdd(&st->n_huf_def, 1u);
    if (comp && b.lit_type >= LIT_HUF) atomicMax(&st->max_lit_regen, b.lit_regen);
    if (comp && b.lit_type >= LIT_HUF && b.nseq == 0) atomicAdd(&st->n_plain_huf, 1u);
    if (i + 1 == nblk && (b.btype == BT_RAW || b.btype == BT_RLE) && b.bsize) st->last_raw = b.bsize | (b.btype == BT_RLE ? 0x80000000u : 0u);
    if (sq) atomicAdd(&st->n_seq_blk, 1u);
}

// A block whose tree description repeats its predecessor's byte for byte (every block of a frame of random ACGT; long stretches of
// any frame coded with a cached tree) defines nothing new: it is taken out of the ownership scan, so that it -- and the treeless
// blocks behind it -- use the earlier block's table, and no table is built for it.
__global__ void k_huf_dedup(const u8 *src, const ZBlock *blk, u32 nblk, i32 *own_huf, ZStat *st)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    if (i < nblk) {
        const ZBlock &b = blk[i];
        const bool def = b.btype == BT_COMP && b.lit_type == LIT_HUF && !b.err;
        keep = def;
        if (def && i > 0) {
            const ZBlock &a = blk[i - 1];
            const u32 n = b.huf_streams_off - b.lit_off;
            if (a.btype == BT_COMP && a.lit_type == LIT_HUF && !a.err && a.huf_streams_off - a.lit_off == n) {
                const u8 *p = src + a.src_off + a.lit_off, *q = src + b.src_off + b.lit_off;
                bool same = true;
                for (u32 k = 0; k < n && same; k++) same = p[k] == q[k];
                if (same) { keep = false; own_huf[i] = -1; }
            }
        }
    }
    const u64 bal = __ballot(keep);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&st->n_huf_distinct, (u32)__popcll(bal));
}

// A flat tree: 2^log symbols, every one of weight 1, i.e. every code exactly `log` bits long (packed random ACGT: sixteen 4-bit
// codes).  Such a stream is a string of fixed-width fields and needs no serial walk (k_flat_literals).
static __device__ __forceinline__ bool huf_is_flat(const u8 *w, u32 nw, u32 log)
{
    if (log > 8) return false;
    u32 n1 = 0;
    for (u32 i = 0; i < nw; i++) { if (w[i] > 1) return false; n1 += w[i]; }
    return n1 == (1u << log);
}

// Direct (4-bit) weight list of a flat tree, recognised from the description alone: 2^L - 1 listed weights, every one of them 0 or
// 1, 2^L - 1 ones among them (the last symbol's weight is implied: 1).  No table is built for such a block -- k_flat_literals
// takes its symbols from the same description.  Returns L, or 0.
static __device__ __forceinline__ u32 huf_flat_direct(const u8 *d, u32 len)
{
    if (len < 1 || d[0] < 128) return 0;
    const u32 nw = d[0] - 127, bytes = (nw + 1) / 2;
    if (1 + bytes > len) return 0;
    u32 ones = 0;
    for (u32 k = 0; k < bytes; k++) {
        const u32 v = d[1 + k], hi = v >> 4, lo = (2 * k + 1 < nw) ? (v & 15) : 0;
        if (hi > 1 || lo > 1) return 0;
        ones += hi + lo;
    }
    const u32 total = ones + 1;                                 // with the implied last symbol
    if (total < 2 || total > 256 || (total & (total - 1))) return 0;
    return (u32)hibit32(total);
}

// range4 (optional, device): [4] = first block whose table may be in force in the wanted byte range, [1] = one past its last block
// (k_find_range); blocks outside need no table.
#define HUF_FEW 2048u           // up to this many distinct trees in a frame: k_build_huf_few (LDS) builds them, else k_build_huf
// phase 0: every tree.  Phase 2: every tree not yet marked flat -- a long frame whose caller can read flat blocks in place first gets its
// flat tree found and its repetitions marked (k_flat_find_main, k_flat_mark_owner: cheap), and the tables of the other trees are built
// here later, on another stream beside the emit of the flat tiles: this one-lane-per-tree build is 0.45 ms whatever the number of trees.
// One tree by one lane, the weights and the builder's workspace in the lane's private arrays (scratch memory).
struct HufSerialWS { u8 w[256]; HufBuildWS ws; };
static __device__ void build_huf_one_serial_ws(const u8 *src, ZBlock *blk, u32 i, u8 *pool, u32 pool_cap, ZStat *st, u32 always_table, u8 *w, HufBuildWS &ws);
static __device__ void build_huf_one_serial(const u8 *src, ZBlock *blk, u32 i, u8 *pool, u32 pool_cap, ZStat *st, u32 always_table)
{
    u8 w[256]; HufBuildWS ws;
    build_huf_one_serial_ws(src, blk, i, pool, pool_cap, st, always_table, w, ws);
}
static __device__ void build_huf_one_serial_ws(const u8 *src, ZBlock *blk, u32 i, u8 *pool, u32 pool_cap, ZStat *st, u32 always_table, u8 *w, HufBuildWS &ws)
{
    const u8 *c = src + blk[i].src_off;
    {
        const u32 fl = always_table ? 0u : huf_flat_direct(c + blk[i].lit_off, blk[i].lit_csize);
        if (fl) {
            blk[i].huf_tab = 0xFFFFFFFFu; blk[i].huf_log = (u8)fl; blk[i].huf_flat = 1;
            atomicMax(&st->max_huf_log, fl); atomicAdd(&st->n_flat, 1u); atomicAdd(&st->n_huf_built, 1u);
            if (fl == 4) atomicMax(&st->flat_main_inv, 0xFFFFFFFFu - i);
            return;
        }
    }
    u32 nw = 0, used = 0;
    u32 log = huf_read_weights_ws(c + blk[i].lit_off, blk[i].lit_csize, w, &nw, &used, ws);
    if (!log) { set_err(st, ZE_CORRUPT); blk[i].err = ZE_CORRUPT; return; }
    u32 bytes = huf_tab_bytes(log);
    u32 off = atomicAdd(&st->huf_pool_used, bytes);
    if (off + bytes > pool_cap) { set_err(st, ZE_POOL); return; }
    huf_build_any_ws((u16 *)(pool + off), w, nw, log, ws);
    blk[i].huf_tab = off; blk[i].huf_log = (u8)log;
    atomicMax(&st->max_huf_log, log);
    const bool flat = huf_is_flat(w, nw, log);
    blk[i].huf_flat = flat; if (flat) atomicAdd(&st->n_flat, 1u);
    if (flat && log == 4) atomicMax(&st->flat_main_inv, 0xFFFFFFFFu - i);
    atomicAdd(&st->n_huf_built, 1u);
}
static __device__ __forceinline__ bool build_huf_wanted(const ZBlock *blk, u32 i, u32 nblk, const ZStat *st, const u64 *range4, const i32 *own_huf, u32 gate, u32 phase)
{
    if (i >= nblk) return false;
    if (gate && st->n_huf_distinct <= HUF_FEW) return false;       // k_build_huf_few has this frame
    if (own_huf[i] != (i32)i) return false;                        // repeats its predecessor's tree (k_huf_dedup)
    if (range4 && (i < (u32)range4[4] || i >= (u32)range4[1])) return false;
    if (blk[i].btype != BT_COMP || blk[i].lit_type != LIT_HUF || blk[i].err) return false;
    if (phase == 2 && blk[i].huf_flat) return false;               // the frame's flat tree and its repetitions: k_flat_find_main / k_flat_mark_owner had them
    return true;
}
__global__ void k_build_huf(const u8 *src, ZBlock *blk, u32 nblk, u8 *pool, u32 pool_cap, ZStat *st, u32 first, const u64 *range4, u32 always_table, const i32 *own_huf, u32 gate, u32 phase)
{
    u32 i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (!build_huf_wanted(blk, i, nblk, st, range4, own_huf, gate, phase)) return;
    build_huf_one_serial(src, blk, i, pool, pool_cap, st, always_table);
}

// The same job, sixteen lanes per tree (four trees per wavefront, one wavefront per workgroup): a frame of tens of thousands of distinct
// trees -- a quality stream, a tree per block -- kept one lane per tree busy with private arrays in scratch memory for 1.8 ms (61 K
// trees) in front of the first literal.  Directly stored weights (4.2.1.1: up to 128 of four bits) whose longest code fits the
// single-level table are handled by the group: eight weights per lane in registers, weight groups ranked by shuffles, the table made
// in LDS and copied out 16 bytes per lane.  FSE-coded weights and codes longer than HUF_FULL_LOG bits: the group's first lane, the old way.
// HUFG_TREES trees per workgroup: four = ONE wavefront and 7 KB of LDS.  With sixteen (256 threads, 27.5 KB) the launch of a FASTQ's read
// names waited 8.9 ms for its turn beside the two Huffman walks of that call, whose wavefronts hold 150 of a CU's 160 KB of LDS: a
// workgroup that needs four wave slots and 27 KB on ONE CU at once gets them when several walks there have ended together, a single
// wavefront takes any slot as it frees (profiles/r05_timeline_fastq_12GB_before.txt; the same finding as DESIGN.md 4.29).
template <u32 HUFG_TREES>
__global__ __launch_bounds__(16 * HUFG_TREES) void k_build_huf16(const u8 *src, ZBlock *blk, u32 nblk, u8 *pool, u32 pool_cap, ZStat *st, u32 first, const u64 *range4, u32 always_table, const i32 *own_huf, u32 gate, u32 phase)
{
    __shared__ __attribute__((aligned(16))) u16 s_tab[HUFG_TREES][HUF_TAB_MAX / 2];
    const u32 lane = threadIdx.x & 63u, grp = threadIdx.x >> 4, sl = threadIdx.x & 15u;
    const u32 i = first + blockIdx.x * HUFG_TREES + grp;
    const bool want = build_huf_wanted(blk, i, nblk, st, range4, own_huf, gate, phase);
    const u8 *d = src; u32 len = 0, hb = 0;
    if (want) { d = src + blk[i].src_off + blk[i].lit_off; len = blk[i].lit_csize; hb = len ? d[0] : 0u; }
    // 0: nothing to do, 1: the group builds, 2: the first lane builds the old way
    u32 how = !want ? 0u : (hb >= 128 ? 1u : 2u);
    const u32 n = hb >= 128 ? hb - 127 : 0u, nbytes = (n + 1) / 2;                // n explicit weights, symbols 0 .. n-1; symbol n is implied
    if (how == 1 && 1 + nbytes > len) how = 2;                                    // (the old way reports it)
    u32 w8 = 0;                                                                   // weights of symbols 8 sl .. 8 sl + 7, four bits each (symbol 8 sl + j at bits 4j)
    if (how == 1) {
#pragma unroll
        for (u32 k = 0; k < 4; k++) {
            const u32 bi = 4 * sl + k;
            if (bi < nbytes) { const u32 v = d[1 + bi]; w8 |= ((v >> 4) | ((2 * bi + 1 < n) ? (v & 15u) << 4 : 0u)) << (8 * k); }
        }
    }
    u32 sum = 0, ones = 0; bool big = false;
#pragma unroll
    for (u32 j = 0; j < 8; j++) { const u32 w = (w8 >> (4 * j)) & 15u; if (w > HUF_LOG_MAX) big = true; else if (w) sum += 1u << (w - 1); ones += w == 1; }
    u32 over1 = 0;                                                                // some weight above 1
#pragma unroll
    for (u32 j = 0; j < 8; j++) over1 |= ((w8 >> (4 * j)) & 15u) > 1u;
    for (u32 dd = 1; dd < 16; dd <<= 1) { sum += (u32)__shfl_xor((int)sum, (int)dd, 64); ones += (u32)__shfl_xor((int)ones, (int)dd, 64); over1 |= (u32)__shfl_xor((int)over1, (int)dd, 64); }
    const u64 gmask = 0xFFFFull << (lane & 48u);
    if (how == 1 && (__ballot(big) & gmask)) how = 2;
    u32 log = 0, lastw = 0;
    u32 g_log = 0, g_flat = 0, g_built = 0;                                      // what this group adds to the frame's counters (summed per workgroup:
                                                                                  // tens of thousands of trees each doing four atomics on the same words took 1.5 ms)
    if (how == 1) {
        // a flat tree recognised from its description (huf_flat_direct): no table, k_flat_literals reads the description itself
        const u32 total = ones + 1;
        if (!always_table && !over1 && total >= 2 && total <= 256 && !(total & (total - 1))) {
            const u32 fl = (u32)hibit32(total);
            if (sl == 0) {
                blk[i].huf_tab = 0xFFFFFFFFu; blk[i].huf_log = (u8)fl; blk[i].huf_flat = 1;
                if (fl == 4) atomicMax(&st->flat_main_inv, 0xFFFFFFFFu - i);
            }
            g_log = fl; g_flat = 1; g_built = 1;
            how = 0;
        } else if (sum == 0) how = 2;
        else {
            log = (u32)hibit32(sum) + 1;
            const u32 rest = (1u << log) - sum;
            if (log > HUF_FULL_LOG || (rest & (rest - 1))) how = 2;                // compact tables and corrupt descriptions: the old way
            else lastw = (u32)hibit32(rest) + 1;
        }
    }
    // (the serial lane's weights and workspace in LDS: private arrays are scratch memory, and a kernel whose every wave wants a scratch
    // allocation took 2.6 ms for 61 K trees and held up every other stream's launches meanwhile)
    __shared__ HufSerialWS s_ws[HUFG_TREES];
    if (how == 2 && sl == 0) build_huf_one_serial_ws(src, blk, i, pool, pool_cap, st, always_table, s_ws[grp].w, s_ws[grp].ws);
    u16 *tab = s_tab[grp];
    u32 pos = 0, n_w1 = 0;
    if (how == 1) {
        for (u32 r = 1; r <= log; r++) {                                          // weight groups in ascending order, symbols in index order inside a group
            u32 m = 0;
#pragma unroll
            for (u32 j = 0; j < 8; j++) m |= (((w8 >> (4 * j)) & 15u) == r ? 1u : 0u) << j;
            const u32 cnt = (u32)__popc(m);
            u32 incl = cnt;
            for (u32 dd = 1; dd < 16; dd <<= 1) { const u32 o = (u32)__shfl_up((int)incl, dd, 16); if (sl >= dd) incl += o; }
            const u32 tot = (u32)__shfl((int)incl, 15, 16);
            const u32 cells = 1u << (r - 1);
            u32 rank = incl - cnt;
            if (r == 1) n_w1 = tot + (lastw == 1);
            if (cells <= 16) {
                while (m) {
                    const u32 j = (u32)__ffs((int)m) - 1; m &= m - 1;
                    const u16 e = (u16)((log + 1 - r) | ((8 * sl + j) << 8));
                    const u32 at = pos + rank * cells;
                    for (u32 k = 0; k < cells; k++) tab[at + k] = e;
                    rank++;
                }
                if (lastw == r && sl == 0) { const u16 e = (u16)((log + 1 - r) | (n << 8)); const u32 at = pos + tot * cells; for (u32 k = 0; k < cells; k++) tab[at + k] = e; }
            } else {
                // a few symbols of many cells each: the group fills them together, one symbol after the other
                for (u32 owner = 0; owner < 16; owner++) {
                    u32 mo = (u32)__shfl((int)m, (int)owner, 16); u32 rk = (u32)__shfl((int)rank, (int)owner, 16);
                    while (mo) {
                        const u32 j = (u32)__ffs((int)mo) - 1; mo &= mo - 1;
                        const u16 e = (u16)((log + 1 - r) | ((8 * owner + j) << 8));
                        const u32 at = pos + rk * cells;
                        for (u32 k = sl; k < cells; k += 16) tab[at + k] = e;
                        rk++;
                    }
                }
                if (lastw == r) { const u16 e = (u16)((log + 1 - r) | (n << 8)); const u32 at = pos + tot * cells; for (u32 k = sl; k < cells; k += 16) tab[at + k] = e; }
            }
            pos += (tot + (lastw == r ? 1u : 0u)) * cells;
        }
    }
    // pool space and counters: one atomic of each kind per workgroup
    __shared__ u32 s_bytes[HUFG_TREES], s_logs[HUFG_TREES], s_flat[HUFG_TREES], s_built[HUFG_TREES], s_base;
    const u32 bytes = how == 1 ? huf_tab_bytes(log) : 0u;
    const bool flat = how == 1 && !over1 && lastw == 1 && log <= 8 && n_w1 == (1u << log);      // huf_is_flat
    if (how == 1) { g_log = log; g_flat = flat; g_built = 1; }
    if (sl == 0) { s_bytes[grp] = bytes; s_logs[grp] = g_log; s_flat[grp] = g_flat; s_built[grp] = g_built; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 tb = 0, ml = 0, nf = 0, nb = 0;
        for (u32 k = 0; k < HUFG_TREES; k++) { const u32 b = s_bytes[k]; s_bytes[k] = tb; tb += b; ml = s_logs[k] > ml ? s_logs[k] : ml; nf += s_flat[k]; nb += s_built[k]; }
        s_base = tb ? atomicAdd(&st->huf_pool_used, tb) : 0u;
        if (ml) atomicMax(&st->max_huf_log, ml);
        if (nf) atomicAdd(&st->n_flat, nf);
        if (nb) atomicAdd(&st->n_huf_built, nb);
    }
    __syncthreads();
    if (how == 1) {
        const u32 off = s_base + s_bytes[grp];
        if (off + bytes > pool_cap) { if (sl == 0) set_err(st, ZE_POOL); return; }
        uint4 *dst = (uint4 *)(pool + off);
        for (u32 k = sl; k < bytes / 16; k += 16) dst[k] = ((const uint4 *)tab)[k];
        if (sl == 0) {
            blk[i].huf_tab = off; blk[i].huf_log = (u8)log;
            blk[i].huf_flat = flat;
            if (flat && log == 4) atomicMax(&st->flat_main_inv, 0xFFFFFFFFu - i);
        }
    }
}

// The same for streams of a few blocks (ids, names, lengths, the last block of a mask stream): one block per workgroup, the tree
// description, the weights, the builder's workspace and the table in LDS, so that the lone working lane waits for LDS, not for
// scratch memory (0.3 - 1 ms per launch otherwise, on the critical path of every small stream).
struct HufLdsWS { HufBuildWS ws; __attribute__((aligned(16))) u8 in[192], w[256]; __attribute__((aligned(16))) u16 tab[HUF_TAB_MAX / 2]; u32 log, off; };
// One tree by one wavefront (blockDim.x == 64), everything in LDS.  What is serial by nature -- FSE-coded weights (4.2.1.2), the
// compact table of codes longer than HUF_FULL_LOG bits -- is lane 0's; directly stored weights, their check and the single-level
// table are spread over the lanes (one lane doing all of it from LDS took 70 - 200 us per tree, in front of the literals of a
// frame with one tree and of a mask stream with thousands).
// local_dst != nullptr: the table goes there (LDS of the calling kernel) instead of into the pool, and nothing is recorded in the block --
// k_huf_par builds the tables of the blocks it decodes this way when nobody has built them (S.log tells it the table's log, 0 = corrupt)
__device__ void build_huf_one_lds(const u8 *src, ZBlock *blk, u32 i, u8 *pool, u32 pool_cap, ZStat *st, HufLdsWS &S, u8 *local_dst = nullptr)
{
    const u8 *c = src + blk[i].src_off + blk[i].lit_off;
    const u32 len = blk[i].lit_csize, n_in = len < 192 ? len : 192;          // a tree description takes at most 129 bytes
    const u32 lane = threadIdx.x;
    __shared__ u32 s_nw;
    __syncthreads();
    for (u32 k = lane; k < n_in; k += 64) S.in[k] = c[k];
    if (lane == 0) { S.log = 0; S.off = 0; s_nw = 0; }
    __syncthreads();
    const u32 hb = n_in ? S.in[0] : 0u;
    if (n_in && hb >= 128) {
        // direct representation: hb - 127 weights of four bits
        const u32 n = hb - 127, bytes = (n + 1) / 2;
        bool bad = 1 + bytes > n_in;
        u32 sum = 0;
        if (!bad) for (u32 k = lane; k < n; k += 64) {
            const u32 w = (k & 1) ? (S.in[1 + k / 2] & 15u) : (S.in[1 + k / 2] >> 4);
            S.w[k] = (u8)w;
            if (w > HUF_LOG_MAX) bad = true; else if (w) sum += 1u << (w - 1);
        }
        for (int d = 32; d; d >>= 1) sum += (u32)__shfl_xor((int)sum, d, 64);
        bad = __ballot(bad) != 0;
        if (lane == 0 && !bad && sum) {
            const u32 log = (u32)hibit32(sum) + 1, rest = (1u << log) - sum;
            if (log <= HUF_LOG_MAX && !(rest & (rest - 1))) { S.w[n] = (u8)(hibit32(rest) + 1); s_nw = n + 1; S.log = log; }
        }
    } else if (lane == 0 && n_in) {
        u32 nw = 0, used = 0;
        const u32 log = huf_read_weights_ws(S.in, n_in, S.w, &nw, &used, S.ws);
        s_nw = nw; S.log = log;
    }
    __syncthreads();
    u32 log = S.log; const u32 nw = s_nw;
    if (lane == 0) {
        u32 off = 0;
        if (!log) { set_err(st, ZE_CORRUPT); if (!local_dst) blk[i].err = ZE_CORRUPT; }
        else if (local_dst) { if (log > HUF_FULL_LOG) huf_build_compact(S.tab, S.w, nw, log, S.ws); }
        else {
            const u32 bytes = huf_tab_bytes(log);
            off = atomicAdd(&st->huf_pool_used, bytes);
            if (off + bytes > pool_cap) { set_err(st, ZE_POOL); log = 0; }
            else {
                blk[i].huf_tab = off; blk[i].huf_log = (u8)log; atomicMax(&st->max_huf_log, log);
                const bool flat = huf_is_flat(S.w, nw, log);
                blk[i].huf_flat = flat; if (flat) atomicAdd(&st->n_flat, 1u);
                if (flat && log == 4) atomicMax(&st->flat_main_inv, 0xFFFFFFFFu - i);
                atomicAdd(&st->n_huf_built, 1u);
                if (log > HUF_FULL_LOG) huf_build_compact(S.tab, S.w, nw, log, S.ws);
            }
        }
        S.log = log; S.off = off;
        for (u32 r = 0; r <= HUF_LOG_MAX + 1; r++) S.ws.cnt[r] = 0;
    }
    __syncthreads();
    log = S.log;
    if (!log) return;
    if (log <= HUF_FULL_LOG) {
        // single-level table (huf_build_table): weight groups in ascending order, inside a group the symbols in index order, symbol i
        // of weight w fills 2^(w-1) cells with  log + 1 - w | i << 8
        u32 wq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const u32 k = 64u * (u32)q + lane; wq[q] = k < nw ? S.w[k] : 0u; if (wq[q]) atomicAdd(&S.ws.cnt[wq[q]], 1u); }
        __syncthreads();
        if (lane == 0) { u32 pos = 0; for (u32 r = 1; r <= log; r++) { S.ws.start[r] = pos; pos += S.ws.cnt[r] << (r - 1); } }
        __syncthreads();
        u32 before[HUF_FULL_LOG + 1];                                  // symbols of weight r in the chunks done so far
#pragma unroll
        for (u32 r = 0; r <= HUF_FULL_LOG; r++) before[r] = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            u32 rank = 0;
#pragma unroll
            for (u32 r = 1; r <= HUF_FULL_LOG; r++) {
                const u64 bal = __ballot(wq[q] == r);
                if (wq[q] == r) rank = before[r] + (u32)__popcll(bal & ((1ull << lane) - 1));
                before[r] += (u32)__popcll(bal);
            }
            const u32 w = wq[q], cells = w ? 1u << (w - 1) : 0u, at = w ? S.ws.start[w] + rank * cells : 0u;
            const u16 e = (u16)((log + 1 - w) | ((64u * (u32)q + lane) << 8));
            if (cells && cells <= 16) for (u32 k = 0; k < cells; k++) S.tab[at + k] = e;
            u64 big = __ballot(cells > 16);                            // a few symbols with many cells: the whole wave fills them
            while (big) {
                const int l = __ffsll((long long)big) - 1; big &= big - 1;
                const u32 a2 = (u32)__shfl((int)at, l, 64), n2 = (u32)__shfl((int)cells, l, 64), e2 = (u32)__shfl((int)e, l, 64);
                for (u32 k = lane; k < n2; k += 64) S.tab[a2 + k] = (u16)e2;
            }
        }
        __syncthreads();
    }
    const u32 bytes = huf_tab_bytes(log);
    uint4 *dst = local_dst ? (uint4 *)local_dst : (uint4 *)(pool + S.off);
    for (u32 k = lane; k < bytes / 16; k += 64) dst[k] = ((const uint4 *)S.tab)[k];
}
__global__ __launch_bounds__(64) void k_build_huf_lds(const u8 *src, ZBlock *blk, u32 nblk, u8 *pool, u32 pool_cap, ZStat *st, u32 first, const i32 *own_huf)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    __shared__ HufLdsWS S;
    u32 i = first + blockIdx.x;
    if (i >= nblk) return;
    if (blk[i].btype != BT_COMP || blk[i].lit_type != LIT_HUF || blk[i].err || own_huf[i] != (i32)i) return;
    build_huf_one_lds(src, blk, i, pool, pool_cap, st, S);
}
// The frame's flat 4-bit tree, looked for among the first FIND_MAIN_TREES trees the frame defines (a genome's sequence stream carries
// it from the start or not at all): one wavefront builds them one after the other until one is flat with sixteen 4-bit codes.  (The sixteen pair codes of packed bases reach up to symbol 0x88: more than the 128 weights a direct
// description holds, so the description is FSE-coded and has to be decoded to be recognised.)  Nothing happens in a frame of at most
// HUF_FEW distinct trees (k_build_huf_few builds all of those).
#define FIND_MAIN_TREES 8u
__global__ __launch_bounds__(64) void k_flat_find_main(const u8 *src, ZBlock *blk, u32 nblk, u8 *pool, u32 pool_cap, ZStat *st, const i32 *own_huf)
{
    __builtin_amdgcn_s_setprio(3);
    __shared__ HufLdsWS S;
    if (st->n_huf_distinct <= HUF_FEW) return;
    // workgroup w builds the w-th tree the frame defines (one wavefront building them one after the other until it met the flat one was
    // 140 us in front of a realistic genome's tile index: its first block holds the telomere's Ns, the second tree is the flat one);
    // flat_main_inv keeps the FIRST flat 4-bit tree whichever workgroup finishes first
    u32 seen = 0, i = 0;
    for (; i < nblk; i++) {                                                    // (uniform: every lane walks the same blocks)
        if (own_huf[i] != (i32)i || blk[i].btype != BT_COMP || blk[i].lit_type != LIT_HUF || blk[i].err) continue;
        if (seen == blockIdx.x) break;
        seen++;
    }
    if (i >= nblk) return;
    build_huf_one_lds(src, blk, i, pool, pool_cap, st, S);                     // table, huf_flat, flat_main_inv (a wavefront per tree, in LDS)
}
// A frame of many blocks and few distinct trees: every workgroup looks through its share of the blocks, 64 at a time, and builds the
// few owners it finds.  Does nothing when the frame has more than HUF_FEW distinct trees (k_build_huf has it then).
__global__ __launch_bounds__(64) void k_build_huf_few(const u8 *src, ZBlock *blk, u32 nblk, u8 *pool, u32 pool_cap, ZStat *st, const u64 *range4, const i32 *own_huf)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    __shared__ HufLdsWS S;
    if (st->n_huf_distinct > HUF_FEW || st->n_huf_distinct == 0) return;     // (a stream of RLE / raw blocks -- an unmasked genome's mask -- has no tree at all)
    u32 lo = 0, hi = nblk;
    if (range4) { lo = (u32)range4[4]; hi = (u32)range4[1]; if (hi > nblk) hi = nblk; }
    const u32 per = (hi - lo + gridDim.x - 1) / gridDim.x;
    const u32 a = lo + blockIdx.x * per, b = a + per < hi ? a + per : hi;
    for (u32 base = a; base < b; base += 64) {
        const u32 i = base + threadIdx.x;
        const bool own = i < b && own_huf[i] == (i32)i && blk[i].btype == BT_COMP && blk[i].lit_type == LIT_HUF && !blk[i].err;
        u64 m = __ballot(own);
        while (m) {
            const u32 k = (u32)__ffsll((long long)m) - 1; m &= m - 1;
            build_huf_one_lds(src, blk, base + k, pool, pool_cap, st, S);
        }
    }
}

// (a lane per TABLE: the lane of a block's second or third table measures the descriptions in front of its own first -- the walk over
// the symbols' counts is short beside the spreading and numbering of up to 512 cells)
__global__ void k_build_fse(const u8 *src, ZBlock *blk, u32 nblk, FseE *pool, u32 pool_cap, ZStat *st)
{
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 i = tid / 3, mine = tid % 3;
    if (i >= nblk) return;
    if (blk[i].btype != BT_COMP || blk[i].nseq == 0 || blk[i].err) return;
    if (blk[i].modes[mine] != SM_FSE) return;
    const u8 *c = src + blk[i].src_off;
    u32 pos = blk[i].seq_off, len = blk[i].bsize;
    const u32 max_log[3] = { 9, 8, 9 }, max_sym[3] = { 35, 31, 52 };
    for (u32 k = 0; k < mine; k++) {
        const u32 m = blk[i].modes[k];
        if (m == SM_RLE) pos++;
        else if (m == SM_FSE) {
            u32 nsym, log;
            const u32 d = fse_read_ncount(c + pos, len - pos, max_log[k], max_sym[k], NULL, &nsym, &log);
            if (!d) return;                                      // (the lane of that table says so)
            pos += d;
        }
    }
    i16 norm[64]; u16 next[64]; u32 nsym, log;
    const u32 d = fse_read_ncount(c + pos, len - pos, max_log[mine], max_sym[mine], norm, &nsym, &log);
    if (!d) { set_err(st, ZE_CORRUPT); return; }
    const u32 off = atomicAdd(&st->fse_pool_used, 1u << log);
    if (off + (1u << log) > pool_cap) { set_err(st, ZE_POOL); return; }
    if (!fse_build_table(pool + off, norm, nsym, log, next)) { set_err(st, ZE_CORRUPT); return; }
    blk[i].fse_tab[mine] = off;
}

__global__ void k_decode_seq(const u8 *src, ZBlock *blk, u32 nblk, const i32 *own_ll, const i32 *own_of, const i32 *own_ml,
                             const u64 *seq_base, const FseE *pool, const FseE *predef,
                             u32 *o_ll, u32 *o_ml, u32 *o_of, u64 *sizes, ZStat *st, u32 skip_own_tables, u32 *wave_list)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    // the predefined tables (what this build's own LZ blocks use) in LDS: the lane's whole job is a chain of dependent table reads
    __shared__ FseE s_pre[160];
    __shared__ u32 s_llt[36], s_mlt[53];
    for (u32 k = threadIdx.x; k < 160; k += blockDim.x) s_pre[k] = predef[k];
    zstd_seq_code_tables(s_llt, s_mlt, threadIdx.x, blockDim.x);
    __syncthreads();
    predef = s_pre;
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblk) return;
    ZBlock &b = blk[i];
    sizes[i] = b.regen;
    if (b.btype != BT_COMP || b.nseq == 0 || b.err) return;
    const i32 *own[3] = { own_ll, own_of, own_ml };
    // blocks left to k_decode_seq_wave(2) go on its list (a frame of this build's own has none, and a launch of a workgroup per block
    // only to find that out was 0.7 ... 2.2 ms of dispatch for the hundreds of thousands of blocks of a FASTQ's names)
    if (skip_own_tables == 2 || (skip_own_tables && own_ll[i] >= 0 && own_of[i] >= 0 && own_ml[i] >= 0 &&
        !(blk[own_ll[i]].modes[0] == SM_PREDEF && blk[own_of[i]].modes[1] == SM_PREDEF && blk[own_ml[i]].modes[2] == SM_PREDEF))) {
        wave_list[atomicAdd(&st->n_wave, 1u)] = i;
        return;
    }
    const u32 predef_off[3] = { 0, 64, 96 }, predef_log[3] = { 6, 5, 6 };
    SeqTab tab[3];
    for (int k = 0; k < 3; k++) {
        i32 ob = own[k][i];
        if (ob < 0) { set_err(st, ZE_CORRUPT); b.err = ZE_CORRUPT; return; }      // repeat mode without a table
        b.fse_owner[k] = ob;
        u32 m = blk[ob].modes[k];
        tab[k].rle = m == SM_RLE; tab[k].rle_sym = blk[ob].fse_tab[k];
        if (m == SM_PREDEF) { tab[k].t = predef + predef_off[k]; tab[k].log = predef_log[k]; }
        else if (m == SM_FSE) { tab[k].t = pool + blk[ob].fse_tab[k]; tab[k].log = blk[ob].fse_log[k]; }
        else { tab[k].t = predef; tab[k].log = 0; }
    }
    u64 base = seq_base[i], sll = 0, sml = 0;
    b.seq_base = base;
    u32 rep_out[3];
    bool uses_rep = false;
    u8 e;
    if (blk[own[0][i]].modes[0] == SM_PREDEF && blk[own[1][i]].modes[1] == SM_PREDEF && blk[own[2][i]].modes[2] == SM_PREDEF) {
        // the predefined tables (this build's LZ blocks at level 1): cells read from LDS as LDS -- through the table pointers of the general
        // case, which may point into the pool in global memory, every look-up was a flat load
        e = zstd_decode_sequences_predef<const FseE *, const u32 *, true>(src + b.src_off + b.seq_bits_off, b.seq_bits_size, b.nseq, s_pre, s_pre + 64, s_pre + 96, s_llt, s_mlt,
                                 o_ll + base, o_ml + base, o_of + base, rep_out, &sll, &sml, &uses_rep);
        if (e == 0xFF) {                                           // a stream of fewer than 8 bytes: the general routine
            SeqTab tp[3];
            tp[0].t = s_pre; tp[0].log = 6; tp[1].t = s_pre + 64; tp[1].log = 5; tp[2].t = s_pre + 96; tp[2].log = 6;
            tp[0].rle = tp[1].rle = tp[2].rle = false; tp[0].rle_sym = tp[1].rle_sym = tp[2].rle_sym = 0;
            e = zstd_decode_sequences<BitReloadWindow, true>(src + b.src_off + b.seq_bits_off, b.seq_bits_size, b.nseq, tp,
                                 o_ll + base, o_ml + base, o_of + base, rep_out, &sll, &sml, &uses_rep, BitReloadWindow());
        }
    } else
    e = zstd_decode_sequences<BitReloadWindow, true>(src + b.src_off + b.seq_bits_off, b.seq_bits_size, b.nseq, tab,
                                 o_ll + base, o_ml + base, o_of + base, rep_out, &sll, &sml, &uses_rep, BitReloadWindow());
    if (e) { set_err(st, e); b.err = e; return; }
    if (uses_rep) atomicOr(&st->rep_slow, 2u);
    if (sll > b.lit_regen || b.lit_regen + sml > ZBLOCK_MAX) { set_err(st, ZE_CORRUPT); b.err = ZE_CORRUPT; return; }
    b.rep_out[0] = rep_out[0]; b.rep_out[1] = rep_out[1]; b.rep_out[2] = rep_out[2];
    b.regen = (u32)(b.lit_regen + sml);
    sizes[i] = b.regen;
    atomicMax(&st->max_seq_regen, b.regen);
}

// The sequences of a block that is NOT under the three predefined tables (what libzstd writes: FSE-coded tables of the block's own or of
// an earlier block, RLE symbols) -- a WAVEFRONT per block: its 64 lanes copy the three tables in force into LDS (at most 512 + 256 + 512
// cells, an RLE symbol as one cell of zero bits under log 0), then one lane walks the sequences with the routine the predefined tables
// have (zstd_decode_sequences_predef with the tables' own logs): every cell a 4-byte LDS read.  A lane per block on tables in the pool
// (k_decode_seq's general routine: three dependent flat loads from global memory per sequence) took 2.4 us per sequence -- 12 ms for a
// block of five thousand, whatever the size of the frame (profiles/r05_levels_before.txt).
#define SEQW_CELLS (512 + 256 + 512)
__device__ __forceinline__ void seq_wave_block(const u8 *src, ZBlock *blk, u32 i, const i32 *own_ll, const i32 *own_of, const i32 *own_ml,
                                               const u64 *seq_base, const FseE *pool, const FseE *predef,
                                               u32 *o_ll, u32 *o_ml, u32 *o_of, u64 *sizes, ZStat *st)
{
    __shared__ FseE s_tab[512 + 256 + 512];
    __shared__ u32 s_llt[36], s_mlt[53];
    const u32 lane = threadIdx.x;
    ZBlock &b = blk[i];
    if (b.err) return;
    const i32 ob[3] = { own_ll[i], own_of[i], own_ml[i] };
    if (ob[0] < 0 || ob[1] < 0 || ob[2] < 0) { if (lane == 0) { set_err(st, ZE_CORRUPT); b.err = ZE_CORRUPT; } return; }      // repeat mode without a table
    const u32 m0 = blk[ob[0]].modes[0], m1 = blk[ob[1]].modes[1], m2 = blk[ob[2]].modes[2];
    if (m0 == SM_PREDEF && m1 == SM_PREDEF && m2 == SM_PREDEF) return;                                                  // k_decode_seq's
    const u32 tab_off[3] = { 0, 512, 768 }, predef_off[3] = { 0, 64, 96 }, predef_log[3] = { 6, 5, 6 }, max_sym[3] = { 35, 31, 52 };
    const u32 mode[3] = { m0, m1, m2 };
    u32 log[3]; bool bad = false;
    for (int k = 0; k < 3; k++) {
        const ZBlock &q = blk[ob[k]];
        if (mode[k] == SM_RLE) {
            log[k] = 0;
            if (q.fse_tab[k] > max_sym[k]) bad = true;
            if (lane == 0) { FseE r; r.sym = (u8)q.fse_tab[k]; r.nbits = 0; r.base = 0; s_tab[tab_off[k]] = r; }
        } else {
            const FseE *from = mode[k] == SM_PREDEF ? predef + predef_off[k] : pool + q.fse_tab[k];
            log[k] = mode[k] == SM_PREDEF ? predef_log[k] : q.fse_log[k];
            for (u32 x = lane; x < (1u << log[k]); x += 64) s_tab[tab_off[k] + x] = from[x];
        }
    }
    zstd_seq_code_tables(s_llt, s_mlt, lane, 64);
    __syncthreads();
    if (lane) return;
    if (bad) { set_err(st, ZE_CORRUPT); b.err = ZE_CORRUPT; return; }
    b.fse_owner[0] = ob[0]; b.fse_owner[1] = ob[1]; b.fse_owner[2] = ob[2];
    const u64 base = seq_base[i]; u64 sll = 0, sml = 0;
    b.seq_base = base;
    u32 rep_out[3]; bool uses_rep = false;
    u8 e = zstd_decode_sequences_predef<const FseE *, const u32 *, true>(src + b.src_off + b.seq_bits_off, b.seq_bits_size, b.nseq, s_tab, s_tab + 512, s_tab + 768, s_llt, s_mlt,
                                                                        o_ll + base, o_ml + base, o_of + base, rep_out, &sll, &sml, &uses_rep, log[0], log[1], log[2]);
    if (e == 0xFF) {                                               // a stream of fewer than 8 bytes: the general routine
        SeqTab tp[3];
        for (int k = 0; k < 3; k++) { tp[k].t = s_tab + tab_off[k]; tp[k].log = log[k]; tp[k].rle = mode[k] == SM_RLE; tp[k].rle_sym = blk[ob[k]].fse_tab[k]; }
        e = zstd_decode_sequences<BitReloadWindow, true>(src + b.src_off + b.seq_bits_off, b.seq_bits_size, b.nseq, tp,
                                                          o_ll + base, o_ml + base, o_of + base, rep_out, &sll, &sml, &uses_rep, BitReloadWindow());
    }
    if (e) { set_err(st, e); b.err = e; return; }
    if (uses_rep) atomicOr(&st->rep_slow, 2u);
    if (sll > b.lit_regen || b.lit_regen + sml > ZBLOCK_MAX) { set_err(st, ZE_CORRUPT); b.err = ZE_CORRUPT; return; }
    b.rep_out[0] = rep_out[0]; b.rep_out[1] = rep_out[1]; b.rep_out[2] = rep_out[2];
    b.regen = (u32)(b.lit_regen + sml);
    sizes[i] = b.regen;
    atomicMax(&st->max_seq_regen, b.regen);
}

// (a bounded grid over the list k_decode_seq left: see there)
__global__ __launch_bounds__(64) void k_decode_seq_wave(const u8 *src, ZBlock *blk, const u32 *wave_list, const i32 *own_ll, const i32 *own_of, const i32 *own_ml,
                                                         const u64 *seq_base, const FseE *pool, const FseE *predef,
                                                         u32 *o_ll, u32 *o_ml, u32 *o_of, u64 *sizes, ZStat *st)
{
    __builtin_amdgcn_s_setprio(3);
    const u32 n = st->n_wave;
    for (u32 t = blockIdx.x; t < n; t += gridDim.x) {
        seq_wave_block(src, blk, wave_list[t], own_ll, own_of, own_ml, seq_base, pool, predef, o_ll, o_ml, o_of, sizes, st);
        __syncthreads();
    }
}
// ---- the same, the walk split in two ------------------------------------------------------------------------------------------------
// k_decode_seq_wave's lane spends 1.1 us on a sequence (2600 cycles for some 150 instructions and twenty branches: refills, the repeat
// codes, three dependent rounds of LDS reads), and nothing of it overlaps: the next cell depends on the last bit taken.  But of a
// sequence's bits only the three STATE fields are on that chain -- the extra bits of the offset, match length and literal length are
// skipped by their count, which a cell can carry (SQC_TB: state bits + the extra bits that belong to its symbol).  So:
//   pass 1, the chain, branch-free: three cells (one LDS read each), the end of the unread bits e -= extra bits + state bits, one
//     32-bit window of the stream at e (two dwords of the segment staged in LDS, v_alignbit), three bit fields, three shift-adds.  Every
//     lane runs it on the same values; what sequence j of the batch started from (e and the three cells) goes to row j of s_rec;
//   pass 2, a lane per sequence: the extra bits out of two windows, the values, coalesced stores;
//   the repeat offsets: a batch without repeat codes (ballot) shifts its last three offsets in; one whose codes are all "the last offset
//     again" takes the last new offset in front of each lane; any other batch composes its sequences' turns by a prefix scan (rep_compose).
// The bit stream is staged 2 KB at a time (a batch of 64 sequences reads at most 64 x 89 bits of it), as aligned dwords counted from
// the aligned address below the stream's start; positions are bit indices from there, the stream read downwards (RFC 8878 4.1).
// a cell: the next state's base as a BYTE offset into its table (states are kept that way: no shift in front of a cell read), the
// state bits to read, the symbol, and state bits + the symbol's extra bits -- what the walk moves down by
#define SQC_PACK(base, nb, sym, xb) ((u32)(base) << 2 | (u32)(nb) << 12 | (u32)(sym) << 16 | ((u32)(xb) + (u32)(nb)) << 22)
#define SQC_BASE4(c) ((c) & 0xFFFu)
#define SQC_NB(c)   (((c) >> 12) & 15u)
#define SQC_SYM(c)  (((c) >> 16) & 63u)
#define SQC_TB(c)   (((c) >> 22) & 63u)
#define SEQW_SEG_DW 512u
#define SEQW_BATCH_BITS (64 * 89 + 64)
__device__ __forceinline__ u32 seqw_window(const u32 *s_seg, i32 lo, u32 segD)         // the 32 bits of the stream from bit lo up
{
    const u32 l = (u32)(lo < (i32)(segD << 5) ? (i32)(segD << 5) : lo);                  // (a corrupt stream runs off its start: clamped, caught by the final position)
    const u32 d = (l >> 5) - segD;
    return __builtin_amdgcn_alignbit(s_seg[d + 1], s_seg[d], l & 31);
}
// ---- repeat offsets of a batch of 64 sequences as a scan (3.1.1.5) ---------------------------------------------------------------------
// A sequence turns the three offsets (r0, r1, r2) into three new ones, each of which is one of the old ones, k times "minus one", or a
// constant (its own new offset); such a turn after another is a turn of the same kind, so the lanes' turns are composed by a prefix
// scan (six steps) instead of walked one after the other.  A term: tag = source slot 0 .. 2 (3: the constant `val`) | k << 2.
struct RepT { u32 tag[3], val[3]; };
__device__ __forceinline__ u32 rep_minus(u32 x, u32 k) { return x >= SYM_BASE ? x + k : (x > k ? x - k : 1u); }     // sym_minus1, k times
__device__ __forceinline__ u32 rep_eval(u32 tag, u32 val, u32 r0, u32 r1, u32 r2)
{
    const u32 src = tag & 3u, k = tag >> 2;
    u32 x = r2; x = src == 1 ? r1 : x; x = src == 0 ? r0 : x;
    const u32 y = rep_minus(x, k);
    return src == 3 ? val : y;
}
// B after A: what B's terms read are A's terms
__device__ __forceinline__ RepT rep_compose(const RepT &A, const RepT &B)
{
    RepT R;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const u32 src = B.tag[i] & 3u, k = B.tag[i] >> 2;
        u32 at = A.tag[2], av = A.val[2];
        at = src == 1 ? A.tag[1] : at; av = src == 1 ? A.val[1] : av;
        at = src == 0 ? A.tag[0] : at; av = src == 0 ? A.val[0] : av;
        const bool a_const = (at & 3u) == 3u;
        const u32 cv = av > k ? av - k : 1u;                    // (constants are concrete offsets)
        const u32 nt = a_const ? at : at + (k << 2), nv = a_const ? cv : av;
        R.tag[i] = src == 3 ? B.tag[i] : nt; R.val[i] = src == 3 ? B.val[i] : nv;
    }
    return R;
}
__device__ __forceinline__ void seq_wave2_block(const u8 *src, ZBlock *blk, u32 i, const i32 *own_ll, const i32 *own_of, const i32 *own_ml,
                                                const u64 *seq_base, const FseE *pool, const FseE *predef,
                                                u32 *o_ll, u32 *o_ml, u32 *o_of, u64 *sizes, ZStat *st, u32 how)
{
    const u32 with_predef = how & 1u; const bool rep_walk = (how & 2u) != 0;
    __shared__ u32 s_cell[SEQW_CELLS];
    __shared__ u32 s_seg[SEQW_SEG_DW + 2];
    __shared__ u32 s_llt[36], s_mlt[53];
    const u32 lane = threadIdx.x;
    ZBlock &b = blk[i];
    if (b.err) return;
    const i32 ob[3] = { own_ll[i], own_of[i], own_ml[i] };
    if (ob[0] < 0 || ob[1] < 0 || ob[2] < 0) { if (lane == 0) { set_err(st, ZE_CORRUPT); b.err = ZE_CORRUPT; } return; }      // repeat mode without a table
    const u32 mode[3] = { blk[ob[0]].modes[0], blk[ob[1]].modes[1], blk[ob[2]].modes[2] };
    if (!with_predef && mode[0] == SM_PREDEF && mode[1] == SM_PREDEF && mode[2] == SM_PREDEF) return;                   // k_decode_seq's
    const u32 tab_off[3] = { 0, 512, 768 }, predef_off[3] = { 0, 64, 96 }, predef_log[3] = { 6, 5, 6 }, max_sym[3] = { 35, 31, 52 };
    u32 log[3]; bool bad = false;
    for (int k = 0; k < 3; k++) {
        const ZBlock &q = blk[ob[k]];
        if (mode[k] == SM_RLE) {
            log[k] = 0;
            const u32 sym = q.fse_tab[k];
            if (sym > max_sym[k]) bad = true;
            else if (lane == 0) s_cell[tab_off[k]] = SQC_PACK(0, 0, sym, k == 0 ? ll_bits(sym) : k == 1 ? sym : ml_bits(sym));
        } else {
            const FseE *from = mode[k] == SM_PREDEF ? predef + predef_off[k] : pool + q.fse_tab[k];
            log[k] = mode[k] == SM_PREDEF ? predef_log[k] : q.fse_log[k];
            for (u32 x = lane; x < (1u << log[k]); x += 64) {
                const FseE e = from[x];
                s_cell[tab_off[k] + x] = SQC_PACK(e.base, e.nbits, e.sym, k == 0 ? ll_bits(e.sym) : k == 1 ? e.sym : ml_bits(e.sym));
            }
        }
    }
    zstd_seq_code_tables(s_llt, s_mlt, lane, 64);
    const u32 size = b.seq_bits_size, nseq = b.nseq;
    const u8 *const bits = src + b.src_off + b.seq_bits_off, *const end = bits + size;
    if (bad || size == 0 || end[-1] == 0) { if (lane == 0) { set_err(st, ZE_CORRUPT); b.err = ZE_CORRUPT; } return; }
    const u32 pre = (u32)((u64)bits & 3);
    const u8 *const A = bits - pre;                                // (at least 8 readable bytes lie in front of the stream)
    i32 e = (i32)(8 * (pre + size) - (8 - (u32)hibit32(end[-1])));  // the end of the unread bits: below the padding and its marker
    u32 segD = 0;
    auto stage = [&](i32 e_now) {                                  // the 2 KB of the stream below e_now (all of it, if it is shorter), dword segD first
        const u32 hiD = ((u32)(e_now > 0 ? e_now : 0) + 31) >> 5;
        segD = hiD > SEQW_SEG_DW ? hiD - SEQW_SEG_DW : 0u;
        __syncthreads();
        for (u32 x = lane; x < SEQW_SEG_DW + 2; x += 64) {
            const u8 *q = A + 4 * (u64)(segD + x);
            u32 v = 0;
            if (q + 4 <= end) v = *(const u32 *)q;
            else for (u32 k = 0; k < 4; k++) if (q + k < end) v |= (u32)q[k] << (8 * k);
            s_seg[x] = v;
        }
        __syncthreads();
    };
    stage(e);
    // the three initial states: LL, OF, ML from the top (as byte offsets into their tables)
    u32 s0, s1, s2;
    {
        e -= (i32)(log[0] + log[1] + log[2]);
        const u32 x = seqw_window(s_seg, e, segD);
        s2 = __builtin_amdgcn_ubfe(x, 0, log[2]) << 2; s1 = __builtin_amdgcn_ubfe(x, log[2], log[1]) << 2; s0 = __builtin_amdgcn_ubfe(x, log[2] + log[1], log[0]) << 2;
    }
    const u64 base = seq_base[i];
    u32 r0 = sym_make(0, 0), r1 = sym_make(1, 0), r2 = sym_make(2, 0);
    bool any_rep = false, corrupt = e < (i32)(8 * pre);
    u32 tll = 0, tml = 0;
    __shared__ uint4 s_rec[64];                                  // what sequence j of the batch started from: e and its three cells
    const u8 *const cells = (const u8 *)s_cell;
    for (u32 i0 = 0; i0 < nseq; i0 += 64) {
        const u32 nb = nseq - i0 < 64 ? nseq - i0 : 64u;
        if (segD && e - (i32)SEQW_BATCH_BITS < (i32)(segD << 5)) stage(e);
        // pass 1.  (In vector registers: every lane holds the same values, and left to itself the compiler moves them to scalar registers
        // and back -- v_readfirstlane in front of every use, v_mov in front of every LDS address: a fifth slower on a genome's frame.
        // Thirty-odd instructions a sequence, and their number is what the walk costs.)
        for (u32 j = 0; j < nb; j++) {
            asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(e));
            const u32 c0 = *(const u32 *)(cells + s0), c1 = *(const u32 *)(cells + 2048 + s1), c2 = *(const u32 *)(cells + 3072 + s2);
            s_rec[j] = make_uint4((u32)e, c0, c1, c2);           // (every lane the same value to the same place)
            const u32 n0 = SQC_NB(c0), n1 = SQC_NB(c1), n2 = SQC_NB(c2);
            e -= (i32)(SQC_TB(c0) + SQC_TB(c1) + SQC_TB(c2));
            if (i0 + j + 1 < nseq) {
                const u32 x = seqw_window(s_seg, e, segD);     // from the top: LL's bits, ML's, OF's
                s1 = SQC_BASE4(c1) + (__builtin_amdgcn_ubfe(x, 0, n1) << 2);
                s2 = SQC_BASE4(c2) + (__builtin_amdgcn_ubfe(x, n1, n2) << 2);
                s0 = SQC_BASE4(c0) + (__builtin_amdgcn_ubfe(x, n1 + n2, n0) << 2);
            } else e += (i32)(n0 + n1 + n2);                   // (the last sequence leaves the states alone)
        }
        const uint4 rec = s_rec[lane];
        const i32 my_e = (i32)rec.x; const u32 my_c0 = rec.y, my_c1 = rec.z, my_c2 = rec.w;
        // pass 2: from the top of a sequence's bits the offset's extra bits, the match length's, the literal length's
        u32 ofv = 4, ll = 1, ml = 0;
        if (lane < nb) {
            const u32 ofc = SQC_SYM(my_c1), lt = s_llt[SQC_SYM(my_c0)], mt = s_mlt[SQC_SYM(my_c2)];
            const u32 lb = lt >> 24, mb = mt >> 24;
            ofv = (1u << ofc) + __builtin_amdgcn_ubfe(seqw_window(s_seg, my_e - (i32)ofc, segD), 0, ofc);
            const u32 x = seqw_window(s_seg, my_e - (i32)(ofc + mb + lb), segD);
            ll = (lt & 0xFFFFFFu) + __builtin_amdgcn_ubfe(x, 0, lb);
            ml = (mt & 0xFFFFFFu) + __builtin_amdgcn_ubfe(x, lb, mb);
            if (ofv - 3 >= SYM_BASE && ofv > 3) corrupt = true;    // beyond any window the format allows
            tll += ll; tml += ml;
        }
        // the repeat offsets (3.1.1.5)
        u32 off = ofv - 3;
        const u64 reps = __ballot(lane < nb && ofv <= 3);
        if (!reps) {
            const u32 a = __builtin_amdgcn_readlane(off, nb - 1), bb = nb >= 2 ? __builtin_amdgcn_readlane(off, nb - 2) : r0,
                      cc = nb >= 3 ? __builtin_amdgcn_readlane(off, nb - 3) : (nb == 2 ? r0 : r1);
            r0 = a; r1 = bb; r2 = cc;
        } else if (!__ballot(lane < nb && ofv <= 3 && ofv - 1 + (ll == 0) != 0)) {
            // repeat codes, but only "the last offset again" (code 1 behind literals: what a run of records with one distance is made
            // of): such a lane takes the last NEW offset in front of it in the batch, or the one the batch came in with; the state behind
            // the batch is its last three new offsets
            any_rep = true;
            const u64 m_new = __ballot(lane < nb && ofv > 3), below = m_new & ((1ull << lane) - 1);
            const u32 got = (u32)__shfl((int)off, below ? 63 - __clzll((long long)below) : 0, 64);
            if (lane < nb && ofv <= 3) off = below ? got : r0;
            if (m_new) {
                u64 m = m_new;
                const int a = 63 - __clzll((long long)m); m &= ~(1ull << a);
                const u32 va = __builtin_amdgcn_readlane(off, a);
                if (!m) { r2 = r1; r1 = r0; }
                else {
                    const int bl = 63 - __clzll((long long)m); m &= ~(1ull << bl);
                    const u32 vb = __builtin_amdgcn_readlane(off, bl);
                    r2 = m ? __builtin_amdgcn_readlane(off, 63 - __clzll((long long)m)) : r0;
                    r1 = vb;
                }
                r0 = va;
            }
        } else if (!rep_walk) {
            any_rep = true;
            // every lane's turn, composed with the turns in front of it
            const bool act = lane < nb, isnew = act && ofv > 3;
            const u32 idx = act && !isnew ? ofv - 1 + (ll == 0) : 0u;           // (idle lanes and new offsets aside: 0 = nothing moves)
            RepT P;
            P.tag[0] = isnew ? 3u : idx == 1 ? 1u : idx == 2 ? 2u : idx == 3 ? (0u | 1u << 2) : 0u; P.val[0] = off;
            P.tag[1] = (isnew || idx != 0) ? 0u : 1u; P.val[1] = 0;
            P.tag[2] = (isnew || idx >= 2) ? 1u : 2u; P.val[2] = 0;
#pragma unroll
            for (u32 d = 1; d < 64; d <<= 1) {
                RepT A;
#pragma unroll
                for (int i = 0; i < 3; i++) { A.tag[i] = (u32)__shfl_up((int)P.tag[i], d, 64); A.val[i] = (u32)__shfl_up((int)P.val[i], d, 64); }
                const RepT C = rep_compose(A, P);
                if (lane >= d) P = C;
            }
            // the offsets in force in front of this lane's sequence: the turns of the lanes below, applied to what the batch came in with
            RepT E;
#pragma unroll
            for (int i = 0; i < 3; i++) { E.tag[i] = (u32)__shfl_up((int)P.tag[i], 1, 64); E.val[i] = (u32)__shfl_up((int)P.val[i], 1, 64); if (lane == 0) { E.tag[i] = (u32)i; E.val[i] = 0; } }
            const u32 b0 = rep_eval(E.tag[0], E.val[0], r0, r1, r2), b1 = rep_eval(E.tag[1], E.val[1], r0, r1, r2), b2 = rep_eval(E.tag[2], E.val[2], r0, r1, r2);
            u32 o = rep_minus(b0, 1);
            o = idx == 2 ? b2 : o; o = idx == 1 ? b1 : o; o = idx == 0 ? b0 : o;
            if (act && !isnew) off = o;
            const u32 f0 = rep_eval(P.tag[0], P.val[0], r0, r1, r2), f1 = rep_eval(P.tag[1], P.val[1], r0, r1, r2), f2 = rep_eval(P.tag[2], P.val[2], r0, r1, r2);
            r0 = __builtin_amdgcn_readlane(f0, 63); r1 = __builtin_amdgcn_readlane(f1, 63); r2 = __builtin_amdgcn_readlane(f2, 63);
        } else {
            any_rep = true;
            const u32 llz = ll == 0;
            // (SEQ_REP=walk, a cross-check of the scan above: one sequence after the other; selects in vector registers, every lane the same values: as scalar code the compiler makes a ladder of branches of it, six
            // taken ones a sequence)
            for (u32 j = 0; j < nb; j++) {
                u32 v = __builtin_amdgcn_readlane(ofv, j), idx = v - 1 + __builtin_amdgcn_readlane(llz, j);
                asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(v), "+v"(idx));
                const bool isnew = v > 3, i0_ = idx == 0, i1_ = idx == 1, i2_ = idx == 2, sh2 = isnew | (idx >= 2), change = isnew | !i0_;
                const u32 dec = r0 - 1, m1c = dec ? dec : 1u, m1 = r0 >= SYM_BASE ? r0 + 1 : m1c;      // sym_minus1
                u32 o = m1;
                o = i2_ ? r2 : o; o = i1_ ? r1 : o; o = i0_ ? r0 : o; o = isnew ? v - 3 : o;
                r2 = sh2 ? r1 : r2;
                r1 = change ? r0 : r1;
                r0 = o;
                if (lane == j) off = o;
            }
        }
        if (lane < nb) { o_ll[base + i0 + lane] = ll; o_ml[base + i0 + lane] = ml; o_of[base + i0 + lane] = off; }
    }
    corrupt = __ballot(corrupt) != 0 || e != (i32)(8 * pre);    // every bit of the stream, and no more
    u64 sll = tll, sml = tml;
    for (u32 d = 32; d; d >>= 1) { sll += __shfl_xor(sll, d); sml += __shfl_xor(sml, d); }
    if (lane) return;
    if (corrupt) { set_err(st, ZE_CORRUPT); b.err = ZE_CORRUPT; return; }
    b.fse_owner[0] = ob[0]; b.fse_owner[1] = ob[1]; b.fse_owner[2] = ob[2];
    b.seq_base = base;
    if (any_rep) atomicOr(&st->rep_slow, 2u);
    if (sll > b.lit_regen || b.lit_regen + sml > ZBLOCK_MAX) { set_err(st, ZE_CORRUPT); b.err = ZE_CORRUPT; return; }
    b.rep_out[0] = r0; b.rep_out[1] = r1; b.rep_out[2] = r2;
    b.regen = (u32)(b.lit_regen + sml);
    sizes[i] = b.regen;
    atomicMax(&st->max_seq_regen, b.regen);
}

__global__ __launch_bounds__(64) void k_decode_seq_wave2(const u8 *src, ZBlock *blk, const u32 *wave_list, const i32 *own_ll, const i32 *own_of, const i32 *own_ml,
                                                          const u64 *seq_base, const FseE *pool, const FseE *predef,
                                                          u32 *o_ll, u32 *o_ml, u32 *o_of, u64 *sizes, ZStat *st, u32 with_predef)
{
    __builtin_amdgcn_s_setprio(3);
    const u32 n = st->n_wave;
    for (u32 t = blockIdx.x; t < n; t += gridDim.x) {
        seq_wave2_block(src, blk, wave_list[t], own_ll, own_of, own_ml, seq_base, pool, predef, o_ll, o_ml, o_of, sizes, st, with_predef);
        __syncthreads();
    }
}

// Entry repeat offsets of every block with sequences, in parallel: a block that introduces three new offsets leaves a state that
// does not depend on what it entered with, so its successor just takes it; a symbolic exit state (few sequences in a block) is
// resolved by substituting the predecessors' exit states one after the other.  The serial composition below remains as the
// fallback for frames with errors in them.
__global__ void k_rep_fast(ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, ZStat *st)
{
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seq_blk) return;
    ZBlock &b = blk[seq_list[t]];
    // walk back over the predecessors, substituting their exit states into what is still symbolic: stops at the first block
    // that leaves a fully concrete state (almost always the immediate predecessor) or at the start of the frame (1, 4, 8)
    u32 cur[3] = { sym_make(0, 0), sym_make(1, 0), sym_make(2, 0) };
    for (u32 p = t; sym_is(cur[0]) || sym_is(cur[1]) || sym_is(cur[2]);) {
        u32 prev[3];
        if (p == 0) { prev[0] = 1; prev[1] = 4; prev[2] = 8; }                     // RFC 8878 3.1.1.5: everything resolves here
        else { const ZBlock &q = blk[seq_list[--p]]; if (q.err) { atomicOr(&st->rep_slow, 1u); return; } prev[0] = q.rep_out[0]; prev[1] = q.rep_out[1]; prev[2] = q.rep_out[2]; }
        for (int k = 0; k < 3; k++) {
            if (!sym_is(cur[k])) continue;
            u32 slot = sym_slot(cur[k]), delta = sym_delta(cur[k]), v = prev[slot];
            if (sym_is(v)) cur[k] = v + delta;                                      // still symbolic: deltas add up
            else { u32 r = v - delta; cur[k] = r ? r : 1; }
        }
    }
    b.rep_in[0] = cur[0]; b.rep_in[1] = cur[1]; b.rep_in[2] = cur[2];
}

// One wave: 64 blocks per step are loaded coalesced, then composed lane by lane through shuffles.
__global__ void k_rep_chain(ZBlock *blk, u32 nblk)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    int lane = threadIdx.x;
    u32 r0 = 1, r1 = 4, r2 = 8;                              // RFC 8878 3.1.1.5 initial repeat offsets
    for (u32 base = 0; base < nblk; base += 64) {
        u32 i = base + lane;
        u32 has = 0, o0 = 0, o1 = 0, o2 = 0;
        if (i < nblk && blk[i].btype == BT_COMP && blk[i].nseq > 0 && !blk[i].err) { has = 1; o0 = blk[i].rep_out[0]; o1 = blk[i].rep_out[1]; o2 = blk[i].rep_out[2]; }
        u32 in0 = 0, in1 = 0, in2 = 0;
        for (int j = 0; j < 64; j++) {
            u32 hj = __shfl(has, j, 64);
            if (!hj) continue;                               // wave-uniform
            u32 a = __shfl(o0, j, 64), b = __shfl(o1, j, 64), c = __shfl(o2, j, 64);
            if (lane == j) { in0 = r0; in1 = r1; in2 = r2; }
            u32 rin[3] = { r0, r1, r2 };
            u32 n0 = sym_resolve(a, rin), n1 = sym_resolve(b, rin), n2 = sym_resolve(c, rin);
            r0 = n0; r1 = n1; r2 = n2;
        }
        if (has) { blk[i].rep_in[0] = in0; blk[i].rep_in[1] = in1; blk[i].rep_in[2] = in2; }
    }
}

__global__ void k_set_offsets(ZBlock *blk, u32 nblk, const u64 *offs, u32 *done, u32 *seq_list, u32 *seq_list_n)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblk) return;
    blk[i].out_off = offs[i];
    bool sq = blk[i].btype == BT_COMP && blk[i].nseq > 0;
    if (done) done[i] = sq ? 0u : 1u;
    (void)seq_list; (void)seq_list_n;
}

// Compact list of blocks with sequences, in block order (rank = exclusive count of seq blocks before i).
__global__ void k_seq_list(const ZBlock *blk, u32 nblk, const u64 *rank, u32 *list)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblk) return;
    if (blk[i].btype == BT_COMP && blk[i].nseq > 0) list[rank[i]] = i;
}
__global__ void k_seq_flag(const ZBlock *blk, u32 nblk, u64 *flag)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nblk) flag[i] = (blk[i].btype == BT_COMP && blk[i].nseq > 0) ? 1 : 0;
}


// ---- fused decode + FASTA emit -----------------------------------------------------------------------------------
// For frames made only of literal blocks (this build's own archives, reference archives of random DNA) the decoded
// 4-bit stream never has to exist in HBM: the lane that decodes a Huffman stream knows the base index of every
// symbol it produces, so it can write the FASTA text itself -- 16 bases per 8 decoded bytes, with the line-end and
// record-end newlines that follow its bases.  Headers are written by k_emit_headers.  This removes the 2 x 4.94 GB
// (10 GB config) round trip of the packed stream and the separate emit pass.
struct LaneEmit { u64 g, rec_end_g, tp, r, tk; u32 col, mstate; };

__device__ __forceinline__ void le_next_record(const EmitP &P, LaneEmit &e)
{
    u64 r = e.r + 1;
    while (r < P.N && P.rec_base[r + 1] == P.rec_base[r]) r++;            // records without bases have no body
    e.r = r;
    if (r < P.N) { e.tp = P.rec_out[r] + P.hdr_len[r]; e.rec_end_g = P.rec_base[r + 1]; }
    else e.rec_end_g = ~0ull;
    e.col = 0;
}
__device__ __forceinline__ void le_init(const EmitP &P, LaneEmit &e, u64 g)
{
    e.g = g; e.col = 0; e.tk = 0; e.mstate = 0; e.r = 0; e.tp = 0; e.rec_end_g = ~0ull;
    if (g >= P.T) return;
    u64 r = upper_bound_u64(P.rec_base, 0, P.N + 1, g) - 1;
    u64 j = g - P.rec_base[r];
    e.r = r; e.rec_end_g = P.rec_base[r + 1];
    e.tp = P.rec_out[r] + P.hdr_len[r] + j + (P.L ? j / P.L : 0);
    e.col = P.L ? (u32)(j % P.L) : 0;
    if (P.masking) {                                                      // tk = number of toggles < g
        u64 lo = 0, hi = P.n_toggles;
        while (lo < hi) { u64 mid = (lo + hi) >> 1; if (P.toggles[mid] < g) lo = mid + 1; else hi = mid; }
        e.tk = lo; e.mstate = (u32)(lo & 1);
    }
}
__device__ __forceinline__ void le_one_base(const EmitP &P, LaneEmit &e, u8 *text, u32 code)
{
    if (P.masking) while (e.tk < P.n_toggles && P.toggles[e.tk] <= e.g) { e.mstate ^= 1; e.tk++; }
    u32 ch = (P.lut[code >> 2] >> (8 * (code & 3))) & 0xFF;
    if (e.mstate) ch += 32;
    text[e.tp++] = (u8)ch;
    e.g++; e.col++;
    bool rec_done = e.g == e.rec_end_g;
    if ((P.L && e.col == P.L) || rec_done) { text[e.tp++] = '\n'; e.col = 0; }
    if (rec_done) le_next_record(P, e);
}
// nb bases (<= 16) held as nibbles of `nib`.  lut2 (LDS): packed byte -> two ASCII bytes.  own_ahead: this lane
// will itself write at least the next 32 bases after this group, so a 16-byte store may run past the group's end.
__device__ __forceinline__ void le_group(const EmitP &P, LaneEmit &e, u8 *text, u64 nib, u32 nb, const u16 *lut2, bool own_ahead)
{
    if (e.g >= P.T) return;
    if (P.T - e.g < nb) nb = (u32)(P.T - e.g);                              // odd total: the final high nibble is padding
    if (nb == 16 && e.g + 16 <= e.rec_end_g && (P.L == 0 || P.L >= 16)) {
        u32 lo32 = (u32)nib, hi32 = (u32)(nib >> 32);
        u32 d0 = (u32)lut2[lo32 & 0xFF] | ((u32)lut2[(lo32 >> 8) & 0xFF] << 16);
        u32 d1 = (u32)lut2[(lo32 >> 16) & 0xFF] | ((u32)lut2[lo32 >> 24] << 16);
        u32 d2 = (u32)lut2[hi32 & 0xFF] | ((u32)lut2[(hi32 >> 8) & 0xFF] << 16);
        u32 d3 = (u32)lut2[(hi32 >> 16) & 0xFF] | ((u32)lut2[hi32 >> 24] << 16);
        if (P.masking && (e.mstate || (e.tk < P.n_toggles && P.toggles[e.tk] < e.g + 16))) {
            u32 m16 = 0; u64 pos = e.g;
            for (;;) {
                u64 nxt = e.tk < P.n_toggles ? P.toggles[e.tk] : ~0ull;
                u64 end = nxt < e.g + 16 ? nxt : e.g + 16;
                if (e.mstate && end > pos) m16 |= (u32)(((1u << (end - pos)) - 1) << (pos - e.g));
                if (nxt >= e.g + 16) break;
                pos = nxt > pos ? nxt : pos; e.mstate ^= 1; e.tk++;
            }
            auto sp = [](u32 x) { return ((x & 1) | ((x & 2) << 7) | ((x & 4) << 14) | ((x & 8) << 21)) * 0x20u; };
            d0 += sp(m16 & 15); d1 += sp((m16 >> 4) & 15); d2 += sp((m16 >> 8) & 15); d3 += sp(m16 >> 12);
        }
        u8 *o = text + e.tp;
        uint4 v; v.x = d0; v.y = d1; v.z = d2; v.w = d3;
        if (P.L && e.col + 16 >= P.L) {
            u32 nl = (u32)P.L - e.col;
            if (nl == 16) { memcpy(o, &v, 16); o[16] = '\n'; }
            else if (own_ahead) {
                // bytes [0,nl) | '\n' | bytes [nl,16): store the 16 bytes, then the tail again one byte later, then the
                // newline.  The second store runs past byte 16; this lane overwrites that region with its next groups.
                u32 w = nl >> 2, bsh = nl & 3;
                u32 c0 = w == 0 ? d0 : (w == 1 ? d1 : (w == 2 ? d2 : d3));
                u32 c1 = w == 0 ? d1 : (w == 1 ? d2 : (w == 2 ? d3 : 0));
                u32 c2 = w == 0 ? d2 : (w == 1 ? d3 : 0);
                u32 c3 = w == 0 ? d3 : 0;
                uint4 t;
                t.x = __builtin_amdgcn_alignbyte(c1, c0, bsh); t.y = __builtin_amdgcn_alignbyte(c2, c1, bsh);
                t.z = __builtin_amdgcn_alignbyte(c3, c2, bsh); t.w = __builtin_amdgcn_alignbyte(0u, c3, bsh);
                memcpy(o, &v, 16);
                memcpy(o + nl + 1, &t, 16);
                o[nl] = '\n';
            } else {
                u64 lo = (u64)d0 | ((u64)d1 << 32), hi = (u64)d2 | ((u64)d3 << 32);
                u32 extra = splice_newline(lo, hi, (int)nl);
                v.x = (u32)lo; v.y = (u32)(lo >> 32); v.z = (u32)hi; v.w = (u32)(hi >> 32);
                memcpy(o, &v, 16); o[16] = (u8)extra;
            }
            e.tp += 17; e.col = e.col + 16 - (u32)P.L;
        } else {
            memcpy(o, &v, 16);
            e.tp += 16; e.col += 16;
        }
        e.g += 16;
        if (e.g == e.rec_end_g) {
            if (!(P.L && e.col == 0)) text[e.tp++] = '\n';                   // unless the line-end newline was the record end
            le_next_record(P, e);
        }
        return;
    }
    for (u32 i = 0; i < nb; i++) le_one_base(P, e, text, (u32)(nib >> (4 * i)) & 15);
}

__global__ void k_emit_headers(EmitP P, u8 *text)
{
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.N) return;
    u32 hl = P.hdr_len[r]; u64 at = P.rec_out[r];
    u64 ids0 = 0, idl = 0, nm0 = 0;
    if (P.has_ids) { ids0 = r ? P.idz[r - 1] + 1 : 0; idl = P.idz[r] - ids0; }
    if (P.has_names) nm0 = r ? P.nmz[r - 1] + 1 : 0;
    for (u32 k = 0; k < hl; k++) {
        u32 ch;
        if (k == 0) ch = P.hdr_char;
        else if (k == hl - 1) ch = '\n';
        else if (P.has_ids) { u64 q = k - 1; ch = q < idl ? P.ids[ids0 + q] : (q == idl ? P.sep : P.names[nm0 + (q - idl - 1)]); }
        else ch = P.names[nm0 + k - 1];
        text[at + k] = (u8)ch;
    }
}

// ---- Huffman literals: one lane per stream, 16 blocks per 64-lane workgroup, tables staged in LDS ----------
// Both sides of every stream go through LDS so that each lane's ~4.5 B/symbol-group trickle becomes whole
// 64-byte sectors on the memory side (64 lanes walk 64 streams that are KiB apart: per-lane 8-byte global
// accesses re-fetch every sector ~8 times once 2048 streams per CU overflow the 32 KiB L1):
//   input : per-lane circular window of two 64-byte sectors in LDS; the next lower sector is loaded into
//           registers one round (32 symbols) before it is committed to LDS, so its latency is hidden
//   output: 16 symbols are gathered in registers and leave as one 16-byte store per lane
// Tables with codes longer than 7 bits (a round could cross more than one sector) use the register-prefetch
// reader instead.
#define HUF_BLOCKS_PER_WG 16
#define HUF_ROUND 32                       // symbols per lane per round
#define HUF_OROW 72                        // output row pitch (64 + 8)
#define HUF_IROW 136                       // input window pitch (128 + 8)
#define HUF_IROW_BIG 264                   // four-sector window for tables of more than 7 bits
// SHARED: launched for a frame of few trees with room for ONE table per workgroup (13.6 instead of 17.4 KB of LDS per wavefront for
// 7-bit codes: 11 instead of 9 wavefronts per CU of a kernel that is bound by its chains' latency at the occupancy LDS allows).  A
// workgroup whose blocks are under different trees marks them in `redo` (as a `sel` array) and leaves them to a launch of the plain kernel.
template <bool FUSE, bool SHARED = false>
__global__ __launch_bounds__(64) void k_huf_literals(const u8 *src, const ZBlock *blk, u32 nblk, const i32 *own_huf,
                                                      const u8 *pool, u32 slot_bytes, u8 *dst, u8 *lit_scratch, ZStat *st, u32 b_first,
                                                      EmitP EP, u8 *text, u32 ipitch, u64 src_len, u32 flat_on, const u8 *sel, u8 *redo = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    u8 *irows = lds + (SHARED ? 1u : HUF_BLOCKS_PER_WG) * slot_bytes;   // 64 input rings of ipitch bytes (136: 2 sectors, 264: 4 sectors)
    u16 *lut2 = (u16 *)(irows + 64 * ipitch);                         // FUSE: packed byte -> two ASCII bytes
    u8 *orows = (u8 *)lut2;                                           // !FUSE: 64 output rows of 72 B + 64 row pointers
    u64 *row_out = (u64 *)(orows + 64 * HUF_OROW);
    if (FUSE) for (u32 v = threadIdx.x; v < 256; v += 64) {
        u32 a = v & 15, b = v >> 4;
        lut2[v] = (u16)(((EP.lut[a >> 2] >> (8 * (a & 3))) & 0xFF) | (((EP.lut[b >> 2] >> (8 * (b & 3))) & 0xFF) << 8));
    }
    int lane = threadIdx.x;
    u32 b0 = b_first + blockIdx.x * HUF_BLOCKS_PER_WG;
    // which of the sixteen blocks have streams for this kernel at all (sel: only the blocks it names; blocks of a flat tree belong to
    // k_flat_literals): a workgroup without any returns before it stages a table
    u64 wanted;
    {
        const u32 bi = b0 + ((u32)lane >> 2);
        bool want = false;
        if (bi < nblk && (!sel || (sel[bi] & 2))) {
            const ZBlock &b = blk[bi];
            if (b.btype == BT_COMP && b.lit_type >= LIT_HUF && !b.err) { const i32 ob = own_huf[bi]; want = ob < 0 || FUSE || !(flat_on && blk[ob].huf_flat); }
        }
        wanted = __ballot(want);
        if (!wanted) return;
    }
    if (SHARED) {
        // one tree for all the blocks decoded here?  (the first wanted block's owner against every wanted block's)
        const u32 j0 = (u32)(__ffsll((long long)wanted) - 1) >> 2;
        const i32 ob0 = own_huf[b0 + j0];
        const u32 bi = b0 + ((u32)lane >> 2);
        const bool mine = (wanted >> (4 * ((u32)lane >> 2))) & 1;
        const bool same = ob0 >= 0 && __all(!mine || own_huf[bi] == ob0);
        if (!same) { if (mine && (lane & 3) == 0) redo[bi] = 2; return; }
        const u32 bytes = huf_tab_bytes(blk[ob0].huf_log);
        const uint4 *g = (const uint4 *)(pool + blk[ob0].huf_tab);
        uint4 *l = (uint4 *)lds;
        for (u32 k = lane; k < bytes / 16; k += 64) l[k] = g[k];
    } else
    for (u32 j = 0; j < HUF_BLOCKS_PER_WG; j++) {                     // stage the table in force for each block
        u32 bi = b0 + j;
        if (bi >= nblk) break;
        if (!((wanted >> (4 * j)) & 1)) continue;
        const ZBlock &b = blk[bi];
        if (b.btype != BT_COMP || b.lit_type < LIT_HUF || b.err) continue;
        i32 ob = own_huf[bi];
        if (ob < 0) continue;
        u32 bytes = huf_tab_bytes(blk[ob].huf_log);
        const uint4 *g = (const uint4 *)(pool + blk[ob].huf_tab);
        uint4 *l = (uint4 *)(lds + j * slot_bytes);
        for (u32 k = lane; k < bytes / 16; k += 64) l[k] = g[k];
    }
    // per-lane stream set-up
    u32 j = lane >> 2, s = lane & 3, bi = b0 + j;
    bool valid = false; u32 log = 1, n = 0; u8 *out = nullptr; const u16 *tab = (const u16 *)lds;
    BitR br; br.consumed = 64; br.c = 0; br.ptr = br.start = src; br.bad = false;
    u8 err = 0;
    if (bi < nblk && ((wanted >> (4 * j)) & 1)) {
        const ZBlock &b = blk[bi];
        if (b.btype == BT_COMP && b.lit_type >= LIT_HUF && !b.err) {
            i32 ob = own_huf[bi];
            if (ob < 0) { if (s == 0) err = ZE_CORRUPT; }                // treeless without a previous table
            else if (!FUSE && flat_on && blk[ob].huf_flat) {}            // fixed-width codes: k_flat_literals has them
            else {
                log = blk[ob].huf_log; tab = (const u16 *)(lds + (SHARED ? 0u : j * slot_bytes));
                const u8 *c = src + b.src_off + b.huf_streams_off;
                u8 *o = (b.nseq == 0 ? dst : lit_scratch) + b.out_off;
                u32 regen = b.lit_regen;
                if (b.nstreams == 1) {
                    if (s == 0) { valid = true; n = regen; out = o; bitr_init(br, c, b.huf_streams_size); }
                } else {
                    u32 s1 = ld16(c), s2 = ld16(c + 2), s3 = ld16(c + 4), tot = b.huf_streams_size - 6, per = (regen + 3) / 4;
                    if (s1 + s2 + s3 >= tot || !s1 || !s2 || !s3 || per * 3 > regen) { if (s == 0) err = ZE_CORRUPT; }
                    else {
                        u32 off = s == 0 ? 0 : (s == 1 ? s1 : (s == 2 ? s1 + s2 : s1 + s2 + s3));
                        u32 sz = s == 0 ? s1 : (s == 1 ? s2 : (s == 2 ? s3 : tot - s1 - s2 - s3));
                        valid = true; n = s < 3 ? per : regen - 3 * per; out = o + s * per;
                        bitr_init(br, c + 6 + off, sz);
                    }
                }
                if (valid && br.bad) { valid = false; err = ZE_CORRUPT; }
            }
        }
    }
    if (!FUSE) row_out[lane] = valid ? (u64)out : 0;
    LaneEmit le; le.g = 0; le.rec_end_g = 0; le.tp = 0; le.r = 0; le.tk = 0; le.col = 0; le.mstate = 0;
    if (FUSE && valid) le_init(EP, le, 2 * (u64)out);                       // dst is null: `out` is the byte offset in the packed stream
    u32 my_rounds = valid ? n / HUF_ROUND : 0xFFFFFFFFu;                 // wave-uniform round count
    for (int d = 32; d; d >>= 1) { u32 o = (u32)__shfl_xor((int)my_rounds, d, 64); my_rounds = o < my_rounds ? o : my_rounds; }
    u32 rounds = my_rounds == 0xFFFFFFFFu ? 0 : my_rounds;
    if (ipitch & 0x8000u) { rounds = 0; ipitch &= 0x7FFFu; }     // NAF_GPU_HUF_GENERIC=1: every symbol through the generic reader (cross-check)
    // tables of more than 7 bits: 32 symbols can take 44 bytes, so the input ring is 4 sectors and a refill feeds 4 symbols
    const bool big = ipitch > HUF_IROW;
    const u32 rmask = big ? 255u : 127u, guard = big ? 192u : 160u;
    __syncthreads();
    u32 R = 0;
    {
        // ---- sector-window reader -------------------------------------------------------------------------
        u8 *irow = irows + lane * ipitch;
        // the window's first fill reads whole sectors around the stream's end: never past the end of the source buffer (the
        // last stream of the last block of a buffer then simply takes the plain reader)
        bool live = valid && (u64)(br.ptr - br.start) >= guard + 32 && (((u64)br.ptr + 7) & ~63ull) + 64 <= (u64)src + src_len;
        u64 gp = (u64)br.ptr, lo = 0;
        uint4 st0, st1, st2, st3; bool pending = false;
        st0 = st1 = st2 = st3 = make_uint4(0, 0, 0, 0);
        if (live) {
            u64 top = (gp + 7) & ~63ull;                                  // sector holding the last container byte
            lo = top - 64;
#pragma unroll
            for (int q = 0; q < 8; q++) { uint4 v = ldg_at<uint4>(lo + 16 * q); u32 o = (u32)((lo + 16 * q) & rmask); *(u64 *)(irow + o) = (u64)v.x | ((u64)v.y << 32); *(u64 *)(irow + o + 8) = (u64)v.z | ((u64)v.w << 32); }
        }
        u32 bits = br.consumed;                                            // bits consumed since the container at gp
        for (; R < rounds; R++) {
            if (!__all(!valid || (live && gp - (u64)br.start >= guard))) break;   // near a stream start: generic reader finishes
            if (valid) {
                if (pending) {                                            // commit the sector fetched during the previous round
                    lo -= 64; u32 o = (u32)(lo & rmask);
                    *(u64 *)(irow + o) = (u64)st0.x | ((u64)st0.y << 32); *(u64 *)(irow + o + 8) = (u64)st0.z | ((u64)st0.w << 32);
                    *(u64 *)(irow + o + 16) = (u64)st1.x | ((u64)st1.y << 32); *(u64 *)(irow + o + 24) = (u64)st1.z | ((u64)st1.w << 32);
                    *(u64 *)(irow + o + 32) = (u64)st2.x | ((u64)st2.y << 32); *(u64 *)(irow + o + 40) = (u64)st2.z | ((u64)st2.w << 32);
                    *(u64 *)(irow + o + 48) = (u64)st3.x | ((u64)st3.y << 32); *(u64 *)(irow + o + 56) = (u64)st3.z | ((u64)st3.w << 32);
                    pending = false;
                }
                // request the sector below when the round after this one may read below `lo`.  The sector is committed at the next
                // round start over the ring's top sector, which must be dead by then: gp < lo + 56 in the two-sector ring
                // (reads reach (gp & ~7) + 15), whatever the stream's rate -- a 1-bit code moves gp by only 4 bytes a round
                if (lo + (big ? 96u : 56u) > gp) {
                    st0 = ldg_at<uint4>(lo - 64); st1 = ldg_at<uint4>(lo - 48); st2 = ldg_at<uint4>(lo - 32); st3 = ldg_at<uint4>(lo - 16); pending = true;
                }
                u64 accs[HUF_ROUND / 8];
                if (!big) {
#pragma unroll
                    for (u32 g = 0; g < HUF_ROUND / 8; g++) {
                        // refill: move the container down by the whole bytes consumed, keep the window top-aligned
                        gp -= bits >> 3; bits &= 7;
                        u32 o = (u32)(gp & 127), sh = (o & 7) * 8;
                        u64 q0 = *(const u64 *)(irow + (o & ~7u)), q1 = *(const u64 *)(irow + (((o & ~7u) + 8) & 127));
                        const u64 w = (sh ? (q0 >> sh) | (q1 << (64 - sh)) : q0) << bits;
                        // the container as two dwords: a code of 1 .. 11 bits leaves through v_alignbit_b32 + a 32-bit shift (a 64-bit
                        // shift is a quarter-rate instruction, and this loop is bound by its vector instructions)
                        u32 whi = (u32)(w >> 32), wlo = (u32)w, a0 = 0, a1 = 0;
#pragma unroll
                        for (u32 q = 0; q < 8; q++) {
                            const u32 e = tab[whi >> (32 - log)], nb = hufe_nb(e);
                            whi = __builtin_amdgcn_alignbit(whi, wlo, 32u - nb); wlo <<= nb; bits += nb;
                            if (q < 4) a0 |= hufe_sym(e) << (8 * q); else a1 |= hufe_sym(e) << (8 * (q - 4));
                        }
                        accs[g] = (u64)a0 | ((u64)a1 << 32);
                    }
                } else {
#pragma unroll
                    for (u32 g = 0; g < HUF_ROUND / 8; g++) {
                        u64 acc = 0;
#pragma unroll
                        for (u32 h = 0; h < 2; h++) {
                            gp -= bits >> 3; bits &= 7;
                            u32 o = (u32)(gp & 255), sh = (o & 7) * 8;
                            u64 q0 = *(const u64 *)(irow + (o & ~7u)), q1 = *(const u64 *)(irow + (((o & ~7u) + 8) & 255));
                            const u64 w = (sh ? (q0 >> sh) | (q1 << (64 - sh)) : q0) << bits;
                            u32 whi = (u32)(w >> 32), wlo = (u32)w, a4 = 0;
#pragma unroll
                            for (u32 q = 0; q < 4; q++) {
                                const u32 e = huf_look(tab, whi >> (32 - log), log), nb = hufe_nb(e);
                                whi = __builtin_amdgcn_alignbit(whi, wlo, 32u - nb); wlo <<= nb; bits += nb;
                                a4 |= hufe_sym(e) << (8 * q);
                            }
                            acc |= (u64)a4 << (32 * h);
                        }
                        accs[g] = acc;
                    }
                }
                if (FUSE) {
#pragma unroll
                    for (u32 g = 0; g < HUF_ROUND / 8; g++) le_group(EP, le, text, accs[g], 16, lut2, true);   // >= 160 more input bytes follow: this lane owns far more than the next 32 bases
                } else {
                    u8 *orow = orows + lane * HUF_OROW + (R & 1) * 32;
#pragma unroll
                    for (u32 g = 0; g < HUF_ROUND / 8; g++) *(u64 *)(orow + 8 * g) = accs[g];
                }
            }
            if (!FUSE && (R & 1)) {                                      // two rounds = 64 bytes per lane: write rows out, 4 lanes per row
                __syncthreads();
#pragma unroll
                for (u32 jj = 0; jj < 4; jj++) {
                    u32 row = jj * 16 + (lane >> 2), piece = lane & 3;
                    u64 o = row_out[row];
                    if (o) {
                        const u8 *r = orows + row * HUF_OROW + piece * 16;
                        uint4 v; u64 a = *(const u64 *)r, bb = *(const u64 *)(r + 8);
                        v.x = (u32)a; v.y = (u32)(a >> 32); v.z = (u32)bb; v.w = (u32)(bb >> 32);
                        memcpy((u8 *)o + (u64)(R - 1) * HUF_ROUND + piece * 16, &v, 16);
                    }
                }
                __syncthreads();
            }
        }
        if (!FUSE && (R & 1) && valid) {                                  // an odd number of rounds ran: flush the pending half row
            const u8 *r = orows + lane * HUF_OROW;
            for (u32 q = 0; q < 32; q += 8) st64(out + (u64)(R - 1) * HUF_ROUND + q, *(const u64 *)(r + q));
        }
        if (valid && live) { gp -= bits >> 3; bits &= 7; br.c = ld64((const u8 *)gp); br.consumed = bits; }
        br.ptr = (const u8 *)gp;
    }
    if (valid) {
        u32 done = R * HUF_ROUND;
        u8 e = 0;
        if (FUSE) {
            u32 rem = n - done;
            while (rem && !e) {                                               // 8 symbols (16 bases) at a time through the generic reader
                u32 k = rem < 8 ? rem : 8; u8 tmp[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
                e = huf_decode_n(br, tab, log, tmp, k);
                u64 nib; memcpy(&nib, tmp, 8);
                le_group(EP, le, text, nib, 2 * k, lut2, false);
                rem -= k;
            }
        } else e = huf_decode_n(br, tab, log, out + done, n - done);
        if (!e) { bitr_reload(br); if (!bitr_finished(br)) e = ZE_CORRUPT; }
        if (e) err = e;
    }
    if (err) set_err(st, err);
}

__device__ __forceinline__ u32 wave_excl_sum(u32 v, u32 *total);
// ---- Huffman literals in PARTS: P lanes per stream (zstd_dec_core.h "a Huffman stream decoded in parts") ---------------------------------
// Lane g of the launch: part g % P of stream (g / P) % 4 of block b_first + g / (4 P); P a power of two up to 64, so a wavefront holds
// whole streams and a lane's predecessor part is the lane below it.  Tables of the workgroup's blocks (one block from P = 16 up) in
// LDS, input straight from global memory through the register window of the reader (a part is a few hundred consecutive bytes, read
// downwards a word at a time, two words ahead), output by the lane itself, 8 symbols per store, into its place of the block's
// literals.  sel / flat_on as in k_huf_literals.
__global__ __launch_bounds__(64) void k_huf_par(const u8 *src, const ZBlock *blk, u32 nblk, const i32 *own_huf, const u8 *pool, u32 slot_bytes,
                                                 u8 *dst, u8 *lit_scratch, ZStat *st, u32 b_first, u32 plog, const u8 *sel, u32 flat_on, u32 margin_env, u32 build, u64 src_len)
{
    // build: the trees of the blocks decoded here may have no table yet (huf_log == 0: the caller of a mostly-flat frame left them out,
    // k_build_huf phase 2) -- the workgroup builds them itself, in LDS, from their descriptions (slot_bytes is HUF_TAB_MAX then)
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    __shared__ HufLdsWS S;
    __shared__ u32 s_log[16];
    const u32 lane = threadIdx.x, P = 1u << plog;
    const u64 g = (u64)blockIdx.x * 64 + lane;
    const u32 k = (u32)g & (P - 1), s = (u32)(g >> plog) & 3;
    const u64 bi64 = (u64)b_first + (g >> (plog + 2));
    const u32 bi = bi64 < nblk ? (u32)bi64 : nblk;
    const u32 wg_first = b_first + (u32)(((u64)blockIdx.x * 64) >> (plog + 2));
    const u32 nb_wg = plog >= 4 ? 1u : 16u >> plog, slot_i = plog >= 4 ? 0u : lane >> (plog + 2);
    bool want = false; i32 ob = -1;
    if (bi < nblk && (!sel || (sel[bi] & 2))) {
        const ZBlock &b = blk[bi];
        if (b.btype == BT_COMP && b.lit_type >= LIT_HUF && !b.err) { ob = own_huf[bi]; want = ob < 0 || !(flat_on && blk[ob].huf_flat); }
    }
    const u64 wanted = __ballot(want);
    if (!wanted) return;
    for (u32 j = 0; j < nb_wg; j++) {                                  // the tables in force, for the blocks that have streams here
        const u32 bj = wg_first + j;
        if (bj >= nblk) break;
        if (!((wanted >> (j << (plog + 2))) & 1)) continue;            // (first lane of block j)
        const i32 oj = own_huf[bj];
        if (oj < 0) continue;
        if (build && blk[oj].huf_log == 0) {                              // (uniform: the whole workgroup builds)
            build_huf_one_lds(src, (ZBlock *)blk, (u32)oj, nullptr, 0u, st, S, lds + j * slot_bytes);
            __syncthreads();
            if (lane == 0) s_log[j] = S.log;
            continue;
        }
        if (lane == 0) s_log[j] = blk[oj].huf_log;
        const u32 bytes = huf_tab_bytes(blk[oj].huf_log);
        const uint4 *gt = (const uint4 *)(pool + blk[oj].huf_tab);
        uint4 *lt = (uint4 *)(lds + j * slot_bytes);
        for (u32 q = lane; q < bytes / 16; q += 64) lt[q] = gt[q];
    }
    __syncthreads();
    // the lane's stream
    bool valid = false; u8 err = 0;
    const u8 *sp = src; u32 sz = 0, n = 0, log = 1; u8 *out = nullptr;
    const u16 *tab = (const u16 *)(lds + slot_i * slot_bytes);
    if (want) {
        const ZBlock &b = blk[bi];
        if (ob < 0) { if (s == 0 && k == 0) err = ZE_CORRUPT; }        // treeless without a previous table
        else if ((log = s_log[slot_i]) == 0) { if (s == 0 && k == 0) err = ZE_CORRUPT; }   // its tree description is corrupt
        else {
            const u8 *c = src + b.src_off + b.huf_streams_off;
            u8 *o = (b.nseq == 0 ? dst : lit_scratch) + b.out_off;
            const u32 regen = b.lit_regen;
            if (b.nstreams == 1) { if (s == 0) { valid = true; sp = c; sz = b.huf_streams_size; n = regen; out = o; } }
            else {
                const u32 s1 = ld16(c), s2 = ld16(c + 2), s3 = ld16(c + 4), tot = b.huf_streams_size - 6, per = (regen + 3) / 4;
                if (s1 + s2 + s3 >= tot || !s1 || !s2 || !s3 || per * 3 > regen) { if (s == 0 && k == 0) err = ZE_CORRUPT; }
                else {
                    const u32 off = s == 0 ? 0 : (s == 1 ? s1 : (s == 2 ? s1 + s2 : s1 + s2 + s3));
                    sz = s == 0 ? s1 : (s == 1 ? s2 : (s == 2 ? s3 : tot - s1 - s2 - s3));
                    valid = true; sp = c + 6 + off; n = s < 3 ? per : regen - 3 * per; out = o + s * per;
                }
            }
            if (valid && (sz == 0 || sp[sz - 1] == 0)) { valid = false; if (k == 0) err = ZE_CORRUPT; }   // no end marker
        }
    }
    u32 E = 0; i32 Bk = 0, Bk1 = 0;
    i32 sk = 0, ek = 0; u32 ck = 0;
    if (valid) {
        E = 8u * (sz - 1) + (u32)hibit32(sp[sz - 1]);
        Bk = hufp_cut(E, P, k); Bk1 = hufp_cut(E, P, k + 1);
        const u32 M = margin_env ? margin_env : hufp_margin(E, n, log);
        i32 p = (k == 0 || (u64)Bk + M >= E) ? (i32)E : Bk + (i32)M;
        HufWin r; hufw_init(r, sp, sz, p, src, src + src_len);
        if (k) hufw_walk(r, p, Bk, tab, log);
        sk = p;
        ck = hufw_walk(r, p, Bk1, tab, log);
        ek = p;
    }
    // every start must be its predecessor's end; the lanes that find otherwise walk again from there, until none does
    for (u32 round = 0; round <= P; round++) {
        const i32 prev = __shfl_up(ek, 1, 64);
        const bool mis = valid && k > 0 && sk != prev;
        if (!__ballot(mis)) break;
        if (mis) {
            i32 p = prev; sk = p;
            HufWin r; hufw_init(r, sp, sz, p, src, src + src_len);
            ck = hufw_walk(r, p, Bk1, tab, log);
            ek = p;
        }
    }
    // the parts' places: prefix sum of the counts inside the stream; all of it must add up before anything is written
    u32 tot_wave;
    const u32 ex = wave_excl_sum(valid ? ck : 0u, &tot_wave);
    const u32 gbase = (u32)__shfl((int)ex, (int)(lane & ~(P - 1)), 64);
    const u32 off = ex - gbase;
    bool bad = valid && k == P - 1 && (off + ck != n || ek != 0);
    bad = __shfl((int)bad, (int)(lane | (P - 1)), 64) != 0;
    if (valid && bad) { if (k == 0) err = ZE_CORRUPT; valid = false; }
    if (valid && ck) {
        i32 p = sk;
        HufWin r; hufw_init(r, sp, sz, p, src, src + src_len);
        hufw_decode(r, p, tab, log, out + off, ck);
        if (p != ek) err = ZE_CORRUPT;
    }
    if (err) set_err(st, err);
}

// ---- Huffman literals of a flat tree: fixed-width fields, nothing serial -------------------------------------------------------------
// When every code of a block's table is L bits long, symbol k of a stream sits at bits [E - L(k+1), E - Lk) of the stream (E = the
// data bits below the end marker; a stream is read from its last byte backwards, RFC 8878 4.2.2), so the 8192 symbols one lane of
// k_huf_literals walks one after the other can be taken by all lanes at once: one workgroup per block, one wavefront per stream,
// a lane per group of 16 symbols -- 8 bytes of input at consecutive (descending) addresses across the lanes, 16 bytes of output
// at consecutive addresses.  L = 4 (sixteen symbols: packed random ACGT) goes through a 256-entry table byte -> two symbols;
// other widths take the fields out one by one.  Reads stay inside the stream; a stream whose size does not match n x L bits is
// corrupt (the serial reader's "all bits consumed" test).
// WGT = 256: a workgroup per block, a wavefront per stream; WGT = 64: a workgroup per STREAM (four per block, each with the tables of its own) --
// beside kernels of single-wavefront workgroups a workgroup of 256 waits for four wave slots of one CU to be free at once.
template <u32 WGT>
__global__ __launch_bounds__(WGT) void k_flat_literals(const u8 *src, const ZBlock *blk, u32 nblk, const i32 *own_huf, const u8 *pool,
                                                        u8 *dst, u8 *lit_scratch, ZStat *st, u32 b_first, const u8 *sel)
{
    __shared__ u16 pair[256];                                    // L == 4: byte -> symbol of its high nibble | symbol of its low nibble << 8
    __shared__ u8 sym[256];                                      // code -> symbol
    constexpr u32 PER = 256u / WGT;                              // workgroups per block
    const u32 bi = b_first + blockIdx.x / PER;
    if (bi >= nblk) return;
    if (sel && !(sel[bi] & 2)) return;
    const ZBlock &b = blk[bi];
    if (b.btype != BT_COMP || b.lit_type < LIT_HUF || b.err) return;
    const i32 ob = own_huf[bi];
    if (ob < 0 || !blk[ob].huf_flat) return;
    const u32 L = blk[ob].huf_log;
    if (blk[ob].huf_tab != 0xFFFFFFFFu) {
        const u16 *tab = (const u16 *)(pool + blk[ob].huf_tab);
        for (u32 t = threadIdx.x; t < (1u << L); t += WGT) sym[t] = (u8)(tab[t] >> 8);
    } else if (threadIdx.x < 64) {
        // no table was built (huf_flat_direct): code k is the k-th symbol of weight 1, in symbol order; the last one is implied
        // (the first wavefront, 64 symbols at a time)
        const u8 *d = src + blk[ob].src_off + blk[ob].lit_off;
        const u32 nw = (u32)d[0] - 127; u32 run = 0;
        for (u32 t0 = 0; t0 <= nw; t0 += 64) {
            const u32 t = t0 + threadIdx.x;
            const bool one = t < nw ? (((t & 1) ? d[1 + (t >> 1)] & 15 : d[1 + (t >> 1)] >> 4) == 1) : t == nw;
            const u64 bal = __ballot(one);
            const u32 rank = run + (u32)__popcll(bal & ((1ull << threadIdx.x) - 1));
            if (one && rank < 256) sym[rank] = (u8)t;
            run += (u32)__popcll(bal);
        }
    }
    __syncthreads();
    if (L == 4) for (u32 t = threadIdx.x; t < 256; t += WGT) pair[t] = (u16)(sym[t >> 4] | ((u32)sym[t & 15] << 8));
    __syncthreads();
    const u32 wave = WGT == 256 ? threadIdx.x >> 6 : blockIdx.x % PER, lane = threadIdx.x & 63;
    const u8 *c = src + b.src_off + b.huf_streams_off;
    u8 *o = (b.nseq == 0 ? dst : lit_scratch) + b.out_off;
    const u32 regen = b.lit_regen;
    const u8 *sp; u32 sz, n;
    if (b.nstreams == 1) { if (wave) return; sp = c; sz = b.huf_streams_size; n = regen; }
    else {
        const u32 s1 = ld16(c), s2 = ld16(c + 2), s3 = ld16(c + 4), tot = b.huf_streams_size - 6, per = (regen + 3) / 4;
        if (s1 + s2 + s3 >= tot || !s1 || !s2 || !s3 || per * 3 > regen) { if (threadIdx.x == 0) set_err(st, ZE_CORRUPT); return; }
        const u32 off = wave == 0 ? 0 : (wave == 1 ? s1 : (wave == 2 ? s1 + s2 : s1 + s2 + s3));
        sz = wave == 0 ? s1 : (wave == 1 ? s2 : (wave == 2 ? s3 : tot - s1 - s2 - s3));
        sp = c + 6 + off; n = wave < 3 ? per : regen - 3 * per; o += wave * per;
    }
    const u32 last = sz ? sp[sz - 1] : 0;
    if (!last) { if (lane == 0) set_err(st, ZE_CORRUPT); return; }
    const u64 E = 8ull * (sz - 1) + (u32)hibit32(last);        // data bits of the stream
    if (E != (u64)n * L) { if (lane == 0) set_err(st, ZE_CORRUPT); return; }
    if (L == 4) {
        const u32 groups = n >> 4;
        for (u32 g = lane; g < groups; g += 64) {
            const u64 B = E - 64ull * (g + 1);                   // first bit of the group's 16 fields (the last of them is lowest)
            const u64 a = B >> 3; const u32 s = (u32)B & 7;
            u64 v = ld64(sp + a);
            if (s) v = (v >> s) | ((u64)sp[a + 8] << (64 - s));
            const u32 hi = (u32)(v >> 32), lo = (u32)v;
            uint4 r;
            r.x = (u32)pair[hi >> 24] | ((u32)pair[(hi >> 16) & 0xFF] << 16);
            r.y = (u32)pair[(hi >> 8) & 0xFF] | ((u32)pair[hi & 0xFF] << 16);
            r.z = (u32)pair[lo >> 24] | ((u32)pair[(lo >> 16) & 0xFF] << 16);
            r.w = (u32)pair[(lo >> 8) & 0xFF] | ((u32)pair[lo & 0xFF] << 16);
            memcpy(o + 16ull * g, &r, 16);
        }
        for (u32 k = (groups << 4) + lane; k < n; k += 64) {     // the last, partial group
            const u64 B = E - 4ull * (k + 1); const u64 a = B >> 3; const u32 s = (u32)B & 7;
            u32 v = sp[a]; if (s > 4) v |= (u32)sp[a + 1] << 8;
            o[k] = sym[(v >> s) & 15];
        }
    } else {
        for (u32 k = lane; k < n; k += 64) {
            const u64 B = E - (u64)L * (k + 1); const u64 a = B >> 3; const u32 s = (u32)B & 7;
            u32 v = sp[a]; if (s + L > 8) v |= (u32)sp[a + 1] << 8;
            o[k] = sym[(v >> s) & ((1u << L) - 1)];
        }
    }
}

// Stream table of a flat frame for the fused emit (ctx.h: ZFlat): four slots per block, the unused ones of a single-stream block
// empty (they start where the block ends).  Checks what k_flat_literals checks: jump table, end marker, size = n x 4 bits.
// Slot s of Huffman block b: false = the block's jump table or the stream's end marker is not what a flat stream has.
static __device__ __forceinline__ bool flat_slot(const u8 *src, const ZBlock &b, u32 s, FlatStream &f)
{
    const u8 *c = src + b.src_off + b.huf_streams_off;
    const u32 regen = b.lit_regen;
    f.q0 = b.out_off + regen; f.A = 0;
    u32 sz = 0, n = 0; const u8 *sp = c;
    if (b.nstreams == 1) { if (s == 0) { sz = b.huf_streams_size; n = regen; f.q0 = b.out_off; } }
    else {
        const u32 s1 = ld16(c), s2 = ld16(c + 2), s3 = ld16(c + 4), tot = b.huf_streams_size - 6, per = (regen + 3) / 4;
        if (s1 + s2 + s3 >= tot || !s1 || !s2 || !s3 || per * 3 > regen) return false;
        const u32 off = s == 0 ? 0 : (s == 1 ? s1 : (s == 2 ? s1 + s2 : s1 + s2 + s3));
        sz = s == 0 ? s1 : (s == 1 ? s2 : (s == 2 ? s3 : tot - s1 - s2 - s3));
        sp = c + 6 + off; n = s < 3 ? per : regen - 3 * per; f.q0 = b.out_off + (u64)s * per;
    }
    bool ok = true;
    if (n) {
        const u32 last = sz ? sp[sz - 1] : 0;
        const u64 E = last ? 8ull * (sz - 1) + (u32)hibit32(last) : 0;
        if (!last || E != 4ull * n) ok = false;
        f.A = 8ull * (u64)(sp - src) + E;
    }
    return ok;
}
// code -> symbol of a flat 4-bit tree from its description (block content + lit_off): code k is the k-th symbol of weight 1 in symbol
// order, the last one implied.  One wavefront; false = not sixteen codes of four bits.  Directly stored weights or FSE-coded ones (the
// sixteen pair codes of packed bases reach up to symbol 0x88: more than the 128 weights a direct description holds).
struct FlatSymWS { HufBuildWS ws; __attribute__((aligned(16))) u8 in[192], w[256]; u32 nw, log; };
static __device__ bool flat_sym_of_tree(const u8 *desc, u32 len, u8 *sym, FlatSymWS &S)
{
    // (one wavefront of a larger workgroup: no workgroup barrier in here)
    const u32 lane = threadIdx.x & 63, n_in = len < 192 ? len : 192;
    for (u32 k = lane; k < n_in; k += 64) S.in[k] = desc[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) { u32 nw = 0, used = 0; S.log = n_in ? huf_read_weights_ws(S.in, n_in, S.w, &nw, &used, S.ws) : 0u; S.nw = nw; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const u32 nw = S.nw;
    if (S.log != 4 || nw > 256) return false;
    u32 run = 0; bool bad = false;
    for (u32 b0 = 0; b0 < nw; b0 += 64) {
        const u32 k = b0 + lane, w = k < nw ? S.w[k] : 0u;
        if (w > 1) bad = true;
        const u64 bal = __ballot(w == 1);
        const u32 rank = run + (u32)__popcll(bal & ((1ull << lane) - 1));
        if (w == 1 && rank < 16) sym[rank] = (u8)k;
        run += (u32)__popcll(bal);
    }
    return !__ballot(bad) && run == 16;
}
__global__ void k_flat_streams(const u8 *src, const ZBlock *blk, u32 nblk, const i32 *own_huf, const u8 *pool, FlatStream *si, u8 *sym, ZStat *st, const u64 *total_out, u64 tail_bytes)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) si[4ull * nblk].q0 = *total_out - tail_bytes, si[4ull * nblk].A = 0;      // (nblk: the Huffman blocks; a final Raw block's bytes follow them)
    if (blockIdx.x == 0) {                                        // code -> symbol, from the one tree of the frame (uniform per workgroup)
        u32 b0 = 0; while (b0 < nblk && own_huf[b0] < 0) b0++;
        if (b0 < nblk) {
            const ZBlock &m = blk[own_huf[b0]];
            if (m.huf_tab != 0xFFFFFFFFu) { if (t < 16) sym[t] = (u8)(((const u16 *)(pool + m.huf_tab))[t] >> 8); }
            else {                                                // a flat tree recognised from its directly stored weights: no table was built
                __shared__ FlatSymWS S;
                if (threadIdx.x < 64 && !flat_sym_of_tree(src + m.src_off + m.lit_off, m.lit_csize, sym, S) && threadIdx.x == 0) set_err(st, ZE_CORRUPT);
            }
        }
    }
    if (t >= 4ull * nblk) return;
    FlatStream f;
    if (!flat_slot(src, blk[(u32)(t >> 2)], (u32)t & 3, f)) set_err(st, ZE_CORRUPT);
    si[t] = f;
}

// ---- a UNIFORM flat frame ----------------------------------------------------------------------------------------------------------
// The frames this build's encoder makes of packed random bases are a string of blocks that repeat the first one in everything but
// their stream bytes: same size (which is what the stride index rests on), same literals header, same tree description, same jump
// table, no sequences.  For such a frame the stream table the in-place emit needs is arithmetic -- block i's slots are block 0's,
// moved by i strides -- and nothing of the general front (parse of every block, ownership scans, tables, offsets: 0.26 ms in front of
// the emit of a 10 GB text, profiles/r04_timeline_uniform_10GB.txt) has to run.  k_uni_head parses block 0 and the few blocks behind the
// stride prefix (k_stride_tail) the ordinary way and decides whether the shape is the one; k_uni_streams has four lanes per block
// compare the block's front bytes with block 0's, check the sequences byte and the end marker of every stream -- everything
// k_parse_blocks, k_huf_dedup and k_flat_streams would have checked -- and write the table; one more workgroup of it decodes the
// tree description for the code -> symbol table.  Any doubt leaves U->ok = 0 or sets U->bad and the frame takes the general way.
struct UniInfo { u32 ok, bad, np, nblk, nhb, regen0, nstreams0, hso0, front, pos0, per, last_raw; u32 off[4], sz[4], n[4]; u64 total_out, end_off; };
__global__ __launch_bounds__(64) void k_uni_head(const u8 *src, u64 off0, u32 S, u32 h0, u32 nmax, const u32 *res, const ZBlock *sblk, const ZStat *st, UniInfo *U, ZBlock *ublk, u32 spec_min)
{
    const u32 lane = threadIdx.x;
    if (lane == 0) { U->ok = 0; U->bad = 0; }
    if (res[1] != 1) return;
    const u32 np = res[0], nblk = st->nblk;
    if (np < 1 || np > nmax || nblk <= spec_min || nblk < np || nblk - np > STRIDE_TAIL) return;
    const u32 ntail = nblk - np;
    if (lane < ntail) { ZBlock b = sblk[np + lane]; zstd_parse_block(src + b.src_off, b); ublk[1 + lane] = b; }
    if (lane == 63) { ZBlock b; memset(&b, 0, sizeof b); b.src_off = off0 + 3; b.bsize = h0 >> 3; b.btype = (u8)((h0 >> 1) & 3); b.last = 0; zstd_parse_block(src + b.src_off, b); b.out_off = 0; ublk[0] = b; }
    __threadfence_block();
    __syncthreads();
    if (lane) return;
    const ZBlock b0 = ublk[0];
    if (b0.err || b0.btype != BT_COMP || b0.lit_type != LIT_HUF || b0.nseq != 0 || b0.lit_regen == 0) return;
    UniInfo u; memset(&u, 0, sizeof u);
    u.np = np; u.nblk = nblk; u.regen0 = b0.lit_regen; u.nstreams0 = b0.nstreams; u.hso0 = b0.huf_streams_off;
    u.front = b0.huf_streams_off + (b0.nstreams == 4 ? 6u : 0u); u.pos0 = b0.lit_off + b0.lit_csize;
    const u8 *c0 = src + b0.src_off;
    if (b0.nstreams == 4) {
        const u8 *c = c0 + b0.huf_streams_off;
        const u32 s1 = ld16(c), s2 = ld16(c + 2), s3 = ld16(c + 4), tot = b0.huf_streams_size - 6, per = (b0.lit_regen + 3) / 4;
        if (s1 + s2 + s3 >= tot || !s1 || !s2 || !s3 || per * 3 > b0.lit_regen) return;
        u.per = per;
        u.off[0] = 6; u.off[1] = 6 + s1; u.off[2] = 6 + s1 + s2; u.off[3] = 6 + s1 + s2 + s3;
        u.sz[0] = s1; u.sz[1] = s2; u.sz[2] = s3; u.sz[3] = tot - s1 - s2 - s3;
        u.n[0] = u.n[1] = u.n[2] = per; u.n[3] = b0.lit_regen - 3 * per;
    } else { u.per = 0; u.off[0] = 0; u.sz[0] = b0.huf_streams_size; u.n[0] = b0.lit_regen; }
    // the blocks behind the prefix: plain Huffman blocks of the same tree (its description repeated byte for byte, or none), then at
    // most one final Raw / RLE block that is not empty
    u64 out = (u64)np * b0.lit_regen; u32 nhb = np;
    const u32 dlen = b0.huf_streams_off - b0.lit_off;
    for (u32 j = 0; j < ntail; j++) {
        ZBlock &t = ublk[1 + j];
        if (t.err) return;
        if (t.btype == BT_COMP) {
            if (nhb != np + j || t.lit_type < LIT_HUF || t.nseq != 0) return;
            if (t.lit_type == LIT_HUF) {
                if (t.huf_streams_off - t.lit_off != dlen) return;
                const u8 *p = c0 + b0.lit_off, *q = src + t.src_off + t.lit_off;
                for (u32 k = 0; k < dlen; k++) if (p[k] != q[k]) return;
            }
            t.out_off = out; out += t.regen; nhb++;
        } else {
            if (j + 1 != ntail || !t.bsize || nblk < 2) return;
            u.last_raw = t.bsize | (t.btype == BT_RLE ? 0x80000000u : 0u);
            out += t.bsize;
        }
    }
    u.nhb = nhb; u.total_out = out; u.end_off = st->end_off;
    u.ok = 1;
    *U = u;
}
__global__ __launch_bounds__(256) void k_uni_streams(const u8 *src, u64 off0, u32 S, const ZBlock *ublk, UniInfo *U, FlatStream *si, u8 *sym)
{
    if (!U->ok) return;
    if (blockIdx.x + 1 == gridDim.x) {                            // the tree: sixteen 4-bit codes, and which symbols they are
        __shared__ FlatSymWS W;
        if (threadIdx.x < 64) {
            const ZBlock &b0 = ublk[0];
            if (!flat_sym_of_tree(src + b0.src_off + b0.lit_off, b0.huf_streams_off - b0.lit_off, sym, W) && threadIdx.x == 0) atomicOr(&U->bad, 1u);
        }
        return;
    }
    const u32 np = U->np, nhb = U->nhb;
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 4ull * nhb) { si[t].q0 = U->total_out - (U->last_raw & 0x7FFFFFFFu); si[t].A = 0; }
    if (t >= 4ull * nhb) return;
    const u32 bi = (u32)(t >> 2), s = (u32)t & 3;
    FlatStream f; bool ok = true;
    if (bi < np) {
        const u8 *c = src + off0 + (u64)bi * S + 3, *c0 = src + off0 + 3;
        if (bi) {                                                 // the front bytes, a quarter per lane; the sequences byte
            const u32 F = U->front, lo = F * s / 4, hi = F * (s + 1) / 4;
            for (u32 k = lo; k < hi; k++) ok = ok && c[k] == c0[k];
            if (s == 3) ok = ok && c[U->pos0] == 0;
        }
        const u32 regen = U->regen0, n = U->n[s];
        f.q0 = (u64)bi * regen + (U->nstreams0 == 4 ? (u64)s * U->per : (s ? regen : 0u)); f.A = 0;
        if (n) {
            const u8 *sp = c + U->hso0 + U->off[s]; const u32 sz = U->sz[s];
            const u32 last = sz ? sp[sz - 1] : 0;
            const u64 E = last ? 8ull * (sz - 1) + (u32)hibit32(last) : 0;
            if (!last || E != 4ull * n) ok = false;
            f.A = 8ull * (u64)(sp - src) + E;
        }
    } else ok = flat_slot(src, ublk[1 + (bi - np)], s, f);
    if (!ok) atomicOr(&U->bad, 1u);
    si[t] = f;
}

// ---- RUNS of equal blocks (k_runs_next, k_runs_probe, k_uni_head_runs, k_uni_streams_runs) --------------------------------------------
// The frame the SHARDED encoder makes of packed bases (enc.hip: naf_gpu_ennaf_shard_finish) is one run of equal blocks per shard
// with a short block where two shards' parts meet: the stride index's prefix ends at the first seam, and every range call of an N-GPU
// decode paid the universal decoder's front (2.8 ms per 50 GB of frame).  Behind a prefix that ends far from the frame's end the walk goes
// on: one thread steps over the few odd blocks (k_runs_next) until a header repeats the first block's -- a new run, whose length the
// next launch probes at all its places at once (k_runs_probe) -- up to URUN_MAX times, then the uniform frame's verdict and stream
// table are made of the segments (runs by arithmetic, odd blocks parsed one by one).  Nothing of it runs for a frame of one run.
#define URUN_MAX 16u
#define USEG_MAX (2u * URUN_MAX + STRIDE_TAIL)
struct UniRuns {
    u32 n_seg, done, fail, nblk, pending, n_odd, pad0, pad1;
    u64 cur, run_off, end_off;
    u32 seg_first[USEG_MAX], seg_n[USEG_MAX], seg_odd[USEG_MAX];  // first block, blocks; seg_odd: index + 1 into the odd blocks (0: a run)
    u64 seg_off[USEG_MAX], seg_out[USEG_MAX];                     // a run's first header; where the segment's bytes land
};
__global__ void k_runs_next(const u8 *src, u64 len, u64 off0, u32 S, u32 h0, const u32 *res, UniRuns *R, ZBlock *odd, u32 *probe, int first)
{
    if (threadIdx.x || blockIdx.x) return;
    if (first) {
        memset(R, 0, sizeof(UniRuns));
        if (res[1] != 2u) { R->fail = 1; return; }
        R->seg_first[0] = 0; R->seg_n[0] = res[0]; R->seg_off[0] = off0; R->seg_odd[0] = 0; R->n_seg = 1;
        R->nblk = res[0]; R->cur = off0 + (u64)res[0] * S;
    }
    if (R->done || R->fail) return;
    if (R->pending) {                                              // the run k_runs_probe measured
        const u64 room = (len - R->run_off) / S;
        const u32 n = *probe < room ? *probe : (u32)room;
        if (!n || R->n_seg >= USEG_MAX) { R->fail = 1; return; }
        const u32 j = R->n_seg++;
        R->seg_first[j] = R->nblk; R->seg_n[j] = n; R->seg_off[j] = R->run_off; R->seg_odd[j] = 0;
        R->nblk += n; R->cur = R->run_off + (u64)n * S; R->pending = 0;
    }
    u64 pos = R->cur;
    for (u32 k = 0; k < STRIDE_TAIL; k++) {
        if (pos + 3 > len) { R->fail = 1; return; }
        const u32 h = ld24(src + pos), last = h & 1, type = (h >> 1) & 3, size = h >> 3;
        if (h == h0 && pos + S <= len && k > 0) {                      // a new run (k == 0: the probe said this block differs -- it cannot be one)
            if (R->n_seg >= 2u * URUN_MAX) { R->fail = 1; return; }
            R->run_off = pos; R->pending = 1; R->cur = pos; *probe = 0xFFFFFFFFu;
            return;
        }
        if (type == 3 || size > ZBLOCK_MAX) { R->fail = 1; return; }
        const u32 csize = type == BT_RLE ? 1 : size;
        if (pos + 3 + csize > len || R->n_odd >= STRIDE_TAIL || R->n_seg >= USEG_MAX) { R->fail = 1; return; }
        ZBlock &b = odd[R->n_odd]; memset(&b, 0, sizeof b); b.src_off = pos + 3; b.bsize = size; b.btype = (u8)type; b.last = (u8)last;
        const u32 j = R->n_seg++;
        R->seg_first[j] = R->nblk; R->seg_n[j] = 1; R->seg_off[j] = pos; R->seg_odd[j] = ++R->n_odd;
        R->nblk++; pos += 3 + csize; R->cur = pos;
        if (last) { R->done = 1; R->end_off = pos; return; }
        // (more than a few odd blocks in a row somewhere else than at the frame's end: not this shape)
        if (k >= 4 && len - pos > (u64)STRIDE_TAIL * (ZBLOCK_MAX + 3u)) { R->fail = 1; return; }
    }
    R->fail = 1;
}
__global__ void k_runs_probe(const u8 *src, u64 len, u32 S, u32 h0, const UniRuns *R, u32 *probe)
{
    if (!R->pending || R->done || R->fail) return;
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 pos = R->run_off + i * S;
    if (pos >= len) return;                                        // (what lies behind the frame is nobody's place: the first place that does not fit is `room` in k_runs_next)
    const bool ok = pos + S <= len && ld24(src + pos) == h0;
    const u64 bad = __ballot(!ok);
    if (bad && (threadIdx.x & 63) == (u32)(__ffsll((long long)bad) - 1)) atomicMin(probe, (u32)i);
}
// the verdict on a frame of runs: k_uni_head's, with the odd blocks wherever they stand (plain Huffman blocks of the first block's
// tree, or treeless; a Raw / RLE block only as the frame's last); fills the segments' landing places
__global__ __launch_bounds__(64) void k_uni_head_runs(const u8 *src, u64 off0, u32 h0, UniRuns *R, ZBlock *odd, UniInfo *U, ZBlock *ublk0, u32 spec_min)
{
    const u32 lane = threadIdx.x;
    if (lane == 0) { U->ok = 0; U->bad = 0; }
    if (!R->done || R->fail || R->nblk <= spec_min) return;
    const u32 n_odd = R->n_odd;
    if (lane < n_odd) { ZBlock b = odd[lane]; zstd_parse_block(src + b.src_off, b); odd[lane] = b; }
    if (lane == 63) { ZBlock b; memset(&b, 0, sizeof b); b.src_off = off0 + 3; b.bsize = h0 >> 3; b.btype = (u8)((h0 >> 1) & 3); b.last = 0; zstd_parse_block(src + b.src_off, b); b.out_off = 0; ublk0[0] = b; }
    __threadfence_block();
    __syncthreads();
    if (lane) return;
    const ZBlock b0 = ublk0[0];
    if (b0.err || b0.btype != BT_COMP || b0.lit_type != LIT_HUF || b0.nseq != 0 || b0.lit_regen == 0) return;
    UniInfo u; memset(&u, 0, sizeof u);
    u.np = R->seg_n[0]; u.nblk = R->nblk; u.regen0 = b0.lit_regen; u.nstreams0 = b0.nstreams; u.hso0 = b0.huf_streams_off;
    u.front = b0.huf_streams_off + (b0.nstreams == 4 ? 6u : 0u); u.pos0 = b0.lit_off + b0.lit_csize;
    const u8 *c0 = src + b0.src_off;
    if (b0.nstreams == 4) {
        const u8 *c = c0 + b0.huf_streams_off;
        const u32 s1 = ld16(c), s2 = ld16(c + 2), s3 = ld16(c + 4), tot = b0.huf_streams_size - 6, per = (b0.lit_regen + 3) / 4;
        if (s1 + s2 + s3 >= tot || !s1 || !s2 || !s3 || per * 3 > b0.lit_regen) return;
        u.per = per;
        u.off[0] = 6; u.off[1] = 6 + s1; u.off[2] = 6 + s1 + s2; u.off[3] = 6 + s1 + s2 + s3;
        u.sz[0] = s1; u.sz[1] = s2; u.sz[2] = s3; u.sz[3] = tot - s1 - s2 - s3;
        u.n[0] = u.n[1] = u.n[2] = per; u.n[3] = b0.lit_regen - 3 * per;
    } else { u.per = 0; u.off[0] = 0; u.sz[0] = b0.huf_streams_size; u.n[0] = b0.lit_regen; }
    u64 out = 0; u32 nhb = 0;
    const u32 dlen = b0.huf_streams_off - b0.lit_off;
    for (u32 j = 0; j < R->n_seg; j++) {
        R->seg_out[j] = out;
        if (!R->seg_odd[j]) { out += (u64)R->seg_n[j] * b0.lit_regen; nhb += R->seg_n[j]; continue; }
        ZBlock &t = odd[R->seg_odd[j] - 1];
        if (t.err) return;
        if (t.btype == BT_COMP) {
            if (t.lit_type < LIT_HUF || t.nseq != 0) return;
            if (t.lit_type == LIT_HUF) {
                if (t.huf_streams_off - t.lit_off != dlen) return;
                const u8 *p = c0 + b0.lit_off, *q = src + t.src_off + t.lit_off;
                for (u32 k = 0; k < dlen; k++) if (p[k] != q[k]) return;
            }
            t.out_off = out; out += t.regen; nhb++;
        } else {
            if (j + 1 != R->n_seg || !t.bsize || R->nblk < 2) return;
            u.last_raw = t.bsize | (t.btype == BT_RLE ? 0x80000000u : 0u);
            out += t.bsize;
        }
    }
    u.nhb = nhb; u.total_out = out; u.end_off = R->end_off;
    u.ok = 1;
    *U = u;
}
__global__ __launch_bounds__(256) void k_uni_streams_runs(const u8 *src, u32 S, const UniRuns *R, const ZBlock *odd, const ZBlock *ublk0, UniInfo *U, FlatStream *si, u8 *sym)
{
    if (!U->ok) return;
    if (blockIdx.x + 1 == gridDim.x) {                            // the tree: sixteen 4-bit codes, and which symbols they are
        __shared__ FlatSymWS W;
        if (threadIdx.x < 64) {
            const ZBlock &b0 = ublk0[0];
            if (!flat_sym_of_tree(src + b0.src_off + b0.lit_off, b0.huf_streams_off - b0.lit_off, sym, W) && threadIdx.x == 0) atomicOr(&U->bad, 1u);
        }
        return;
    }
    __shared__ u32 s_first[USEG_MAX]; __shared__ u32 s_nseg;
    if (threadIdx.x < USEG_MAX) s_first[threadIdx.x] = threadIdx.x < R->n_seg ? R->seg_first[threadIdx.x] : 0xFFFFFFFFu;
    if (threadIdx.x == 0) s_nseg = R->n_seg;
    __syncthreads();
    const u32 nhb = U->nhb;
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 4ull * nhb) { si[t].q0 = U->total_out - (U->last_raw & 0x7FFFFFFFu); si[t].A = 0; }
    if (t >= 4ull * nhb) return;
    const u32 bi = (u32)(t >> 2), s = (u32)t & 3;
    u32 lo = 0, hi = s_nseg;                                       // the segment that holds block bi: the last one with seg_first <= bi
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (s_first[mid] <= bi) lo = mid; else hi = mid; }
    FlatStream f; bool ok = true;
    if (!R->seg_odd[lo]) {
        const u32 k = bi - s_first[lo];
        const u8 *c0 = src + ublk0[0].src_off, *cb = src + R->seg_off[lo] + (u64)k * S + 3;
        if (bi) {
            const u32 F = U->front, l0 = F * s / 4, h1 = F * (s + 1) / 4;
            for (u32 q = l0; q < h1; q++) ok = ok && cb[q] == c0[q];
            if (s == 3) ok = ok && cb[U->pos0] == 0;
        }
        const u32 regen = U->regen0, n = U->n[s];
        f.q0 = R->seg_out[lo] + (u64)k * regen + (U->nstreams0 == 4 ? (u64)s * U->per : (s ? regen : 0u)); f.A = 0;
        if (n) {
            const u8 *sp = cb + U->hso0 + U->off[s]; const u32 sz = U->sz[s];
            const u32 last = sz ? sp[sz - 1] : 0;
            const u64 E = last ? 8ull * (sz - 1) + (u32)hibit32(last) : 0;
            if (!last || E != 4ull * n) ok = false;
            f.A = 8ull * (u64)(sp - src) + E;
        }
    } else ok = flat_slot(src, odd[R->seg_odd[lo] - 1], s, f);
    if (!ok) atomicOr(&U->bad, 1u);
    si[t] = f;
}

// ---- a frame that is MOSTLY flat (ctx.h: ZFlat, `cls`) --------------------------------------------------------------------------------
// The flat tree of the frame is the one of its first block that defines a flat 4-bit tree (`main`).  A defining block carries the same
// tree when its description repeats main's byte for byte (huf_flat = 2).
// by_bytes: the other trees have no tables yet (k_build_huf phase 2 comes later): a repetition of main's description takes main's table
__global__ void k_flat_mark_owner(const u8 *src, ZBlock *blk, u32 nblk, const i32 *own_huf, u32 main, u32 by_bytes)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblk || own_huf[i] != (i32)i) return;
    ZBlock &b = blk[i];
    if (b.btype != BT_COMP || b.lit_type != LIT_HUF || b.err) return;
    if (!by_bytes && (!b.huf_flat || b.huf_log != 4)) return;
    const ZBlock &m = blk[main];
    const u32 n = b.huf_streams_off - b.lit_off;
    if (m.huf_streams_off - m.lit_off != n) return;
    const u8 *p = src + m.src_off + m.lit_off, *q = src + b.src_off + b.lit_off;
    for (u32 k = 0; k < n; k++) if (p[k] != q[k]) return;
    b.huf_flat = 2;
    if (by_bytes) { b.huf_log = 4; b.huf_tab = m.huf_tab; }
}
// code -> symbol of the main tree: from its table when one was built, else from its directly stored weights (code k is the k-th
// symbol of weight 1 in symbol order, the last one implied -- as in k_flat_literals).  One workgroup of 256.
__global__ __launch_bounds__(256) void k_flat_sym(const u8 *src, const ZBlock *blk, u32 main, const u8 *pool, u8 *sym)
{
    const ZBlock &m = blk[main];
    const u32 t = threadIdx.x;
    if (m.huf_tab != 0xFFFFFFFFu) { if (t < 16) sym[t] = (u8)(((const u16 *)(pool + m.huf_tab))[t] >> 8); return; }
    __shared__ u32 s_wc[4];
    const u8 *d = src + m.src_off + m.lit_off;
    const u32 nw = (u32)d[0] - 127;
    const bool one = t < nw ? (((t & 1) ? d[1 + (t >> 1)] & 15 : d[1 + (t >> 1)] >> 4) == 1) : t == nw;
    const u64 bal = __ballot(one);
    if ((t & 63) == 0) s_wc[t >> 6] = (u32)__popcll(bal);
    __syncthreads();
    u32 rank = (u32)__popcll(bal & ((1ull << (t & 63)) - 1));
    for (u32 q = 0; q < (t >> 6); q++) rank += s_wc[q];
    if (one && rank < 16) sym[rank] = (u8)t;
}
// cls0[i] = 1: block i can be read in place (a plain Huffman block whose tree in force is the main one), else 2
__global__ void k_flat_class(const ZBlock *blk, u32 nblk, const i32 *own_huf, u8 *cls0, u32 *n_walk)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    bool walk = false;
    if (i < nblk) {
        const ZBlock &b = blk[i];
        const bool huf = b.btype == BT_COMP && b.lit_type >= LIT_HUF && !b.err;
        bool f = huf && b.nseq == 0 && b.lit_regen > 0;
        const i32 ob = huf ? own_huf[i] : -1;
        if (f) f = ob >= 0 && blk[ob].huf_flat == 2;
        cls0[i] = f ? 1 : 2;
        walk = huf && !(ob >= 0 && blk[ob].huf_flat);             // its streams need the Huffman walk (k_huf_literals / k_huf_par)
    }
    const u64 bal = __ballot(walk);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(n_walk, (u32)__popcll(bal));
}
// ... and the flat neighbours of a block that is decoded are decoded too (3 = either way): a tile of text that lies across the border
// then finds all of its packed bytes on one side or the other
__global__ void k_flat_class2(const u8 *cls0, u32 nblk, u8 *cls, u32 *n_decoded)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    u32 c = 0;
    if (i < nblk) {
        c = cls0[i];
        if (c == 1 && ((i > 0 && cls0[i - 1] == 2) || (i + 1 < nblk && cls0[i + 1] == 2))) c = 3;
        cls[i] = (u8)c;
    }
    const u64 bal = __ballot(c & 2);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(n_decoded, (u32)__popcll(bal));
}
// stream table as k_flat_streams makes it, for every block: a block that cannot be read in place is one slot of FLAT_DECODED
__global__ void k_flat_streams_mixed(const u8 *src, const ZBlock *blk, u32 nblk, const u8 *cls, FlatStream *si, ZStat *st, const u64 *total_out)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) si[4ull * nblk].q0 = *total_out, si[4ull * nblk].A = 0;
    if (t >= 4ull * nblk) return;
    const u32 bi = (u32)(t >> 2), s = (u32)t & 3;
    const ZBlock &b = blk[bi];
    FlatStream f; f.q0 = b.out_off + b.regen; f.A = 0;
    if (!(cls[bi] & 1)) { f.A = FLAT_DECODED; if (s == 0) f.q0 = b.out_off; si[t] = f; return; }
    const u8 *c = src + b.src_off + b.huf_streams_off;
    const u32 regen = b.lit_regen;
    u32 sz = 0, n = 0; const u8 *sp = c;
    if (b.nstreams == 1) { if (s == 0) { sz = b.huf_streams_size; n = regen; f.q0 = b.out_off; } }
    else {
        const u32 s1 = ld16(c), s2 = ld16(c + 2), s3 = ld16(c + 4), tot = b.huf_streams_size - 6, per = (regen + 3) / 4;
        if (s1 + s2 + s3 >= tot || !s1 || !s2 || !s3 || per * 3 > regen) { set_err(st, ZE_CORRUPT); si[t] = f; return; }
        const u32 off = s == 0 ? 0 : (s == 1 ? s1 : (s == 2 ? s1 + s2 : s1 + s2 + s3));
        sz = s == 0 ? s1 : (s == 1 ? s2 : (s == 2 ? s3 : tot - s1 - s2 - s3));
        sp = c + 6 + off; n = s < 3 ? per : regen - 3 * per; f.q0 = b.out_off + (u64)s * per;
    }
    if (n) {
        const u32 last = sz ? sp[sz - 1] : 0;
        const u64 E = last ? 8ull * (sz - 1) + (u32)hibit32(last) : 0;
        if (!last || E != 4ull * n) set_err(st, ZE_CORRUPT);
        f.A = 8ull * (u64)(sp - src) + E;
    }
    si[t] = f;
}

// ---- raw / RLE blocks and raw / RLE literal sections: one workgroup per block ------------------------------
template <u32 WGT>
__global__ __launch_bounds__(WGT) void k_copy_fill(const u8 *src, const ZBlock *blk, u32 nblk, u8 *dst, u8 *lit_scratch, u32 b_first, const u8 *sel)
{
    u32 i = b_first + blockIdx.x / (256u / WGT);                 // (WGT = 64: four workgroups per block, as in k_flat_literals)
    if (i >= nblk) return;
    if (sel && !(sel[i] & 2)) return;
    const ZBlock &b = blk[i];
    const u8 *from; u8 *to; u32 n; bool fill;
    if (b.btype == BT_RAW) { from = src + b.src_off; to = dst + b.out_off; n = b.bsize; fill = false; }
    else if (b.btype == BT_RLE) { from = src + b.src_off; to = dst + b.out_off; n = b.bsize; fill = true; }
    else if (!b.err && b.lit_type <= LIT_RLE) {
        from = src + b.src_off + b.lit_off; to = (b.nseq == 0 ? dst : lit_scratch) + b.out_off; n = b.lit_regen; fill = b.lit_type == LIT_RLE;
    } else return;
    u32 t = (blockIdx.x % (256u / WGT)) * WGT + threadIdx.x;
    if (fill) {
        u64 v = 0x0101010101010101ull * from[0];
        for (u32 k = t * 8; k + 8 <= n; k += 256 * 8) st64(to + k, v);
        for (u32 k = (n & ~7u) + t; k < n; k += 256) to[k] = (u8)v;
    } else {
        for (u32 k = t * 8; k + 8 <= n; k += 256 * 8) st64(to + k, ld64(from + k));
        for (u32 k = (n & ~7u) + t; k < n; k += 256) to[k] = from[k];
    }
}

// the block that holds output byte x, among blocks [0, hi) (offs[hi] > x is known); shift: log2 of the frame's block size when its first
// blocks regenerate the same power of two (libzstd: 2^17 until the last block), 0xFF otherwise
__device__ __forceinline__ u32 find_block(const u64 *offs, u32 hi, u64 x, u32 shift)
{
    if (shift < 64) { const u64 g = x >> shift; if (g < hi && offs[g] <= x && x < offs[g + 1]) return (u32)g; }
    u32 lo = 0;
    while (lo + 1 < hi) { const u32 mid = (lo + hi) >> 1; if (offs[mid] <= x) lo = mid; else hi = mid; }
    return lo;
}
// ---- sequence execution (3.1.1.4): one wave per block with sequences ----------------------------------------
// A wait gives up when nothing anywhere has run for a count of polls AND for three seconds of wall time (s_memrealtime, 100 MHz): a
// device that is time-sliced or busy with another stream's kernels polls slowly without being stuck -- a valid frame must
// never be called corrupt because its executor was kept waiting.
#define EXEC_GIVE_UP_TICKS 300000000ull
__device__ __forceinline__ bool exec_stalled(u64 &since)
{
    const u64 now = __builtin_amdgcn_s_memrealtime();
    if (!since) { since = now ? now : 1; return false; }
    return now - since > EXEC_GIVE_UP_TICKS;
}

__device__ __forceinline__ void wait_block_done(volatile u32 *done, u32 j, ZStat *st)
{
    // lane 0 polls (relaxed, agent scope); one acquire afterwards drops stale L1 lines (guide G16)
    if (threadIdx.x == 0) {
        // (bounded: a block that never completes -- a bug, or a frame whose sequences lie about their sources -- must end in wrong bytes
        // and an error, not in a device that has to be reset; about ten seconds)
        // (the count starts again whenever some block of the launch has finished meanwhile: a long chain of blocks is not a hang)
        u32 spins = 0, seen = __hip_atomic_load(&st->n_exec_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); u64 since = 0;
        while (__hip_atomic_load(&done[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins >= (1u << 22)) {
                const u32 now = __hip_atomic_load(&st->n_exec_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (now != seen) since = 0;
                else if (exec_stalled(since)) { set_err(st, ZE_CORRUPT); break; }   // gave up: what is read from block j is not its output -- the call reports the frame as corrupt
                seen = now; spins = 0;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// n bytes by the 64 lanes of a wavefront, 16 per lane and step, at any alignment (the regions do not overlap, or the source lies
// at least n bytes below the destination).  A block of libzstd's with one match in it is 128 KiB of literals around that match:
// copied a byte per lane and step they were two thousand dependent round trips -- 2 ms for a kernel with almost nothing to do.
__device__ __forceinline__ void wave_copy(u8 *d, const u8 *s, u32 n, u32 lane)
{
#pragma unroll 4
    for (u32 i = lane * 16; i + 16 <= n; i += 64 * 16) { uint4 v; __builtin_memcpy(&v, s + i, 16); __builtin_memcpy(d + i, &v, 16); }
    const u32 done = n & ~15u;
    if (lane < (n & 15u)) d[done + lane] = s[done + lane];
}
__global__ __launch_bounds__(64) void k_exec_seq(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, const u64 *offs, u32 nblk,
                                                  const u32 *o_ll, const u32 *o_ml, const u32 *o_of,
                                                  const u8 *lit_scratch, u8 *dst, u32 *done, ZStat *st)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    __shared__ u32 sh_ticket;
    int lane = threadIdx.x;
    if (lane == 0) sh_ticket = atomicAdd(&st->ticket, 1u);
    __syncthreads();
    u32 t = sh_ticket;
    if (t >= n_seq_blk) return;
    u32 bi = seq_list[t];
    const ZBlock &b = blk[bi];
    u64 out_off = b.out_off;
    u8 *out = dst + out_off;
    const u8 *lits = lit_scratch + out_off;
    u32 rep_in[3] = { b.rep_in[0], b.rep_in[1], b.rep_in[2] };
    u64 sbase = b.seq_base;
    u32 op = 0, lp = 0, nseq = b.err ? 0 : b.nseq;
    u32 lo_idx = bi;                       // blocks [lo_idx, bi) are known complete
    u32 fenced = 0;                        // bytes of this block's output known visible to the whole wave
    bool bad = false;
    for (u32 s = 0; s < nseq; s++) {
        u32 ll = o_ll[sbase + s], ml = o_ml[sbase + s];
        u32 off = sym_resolve(o_of[sbase + s], rep_in);
        wave_copy(out + op, lits + lp, ll, (u32)lane);
        op += ll; lp += ll;
        u64 pos_abs = out_off + op;
        if (off > pos_abs) { bad = true; break; }                     // reaches before the frame start
        u64 src_abs = pos_abs - off;
        if (src_abs < offs[lo_idx]) {
            // source (partly) in blocks not yet known complete: confirm every block from the one holding src_abs up to lo_idx.  (The
            // test is against the lowest block confirmed so far, not against this block's start: a later match into an already
            // confirmed block found "the block in front of it" here and waited for that one -- an idle wait in a whole decode, for
            // ever in a range decode, where blocks in front of the range's closure never run.)
            u32 lo = 0, hi = lo_idx;                                 // largest j with offs[j] <= src_abs
            while (lo + 1 < hi) { u32 mid = (lo + hi) >> 1; if (offs[mid] <= src_abs) lo = mid; else hi = mid; }
            for (u32 j = lo_idx; j-- > lo;) wait_block_done(done, j, st);
            if (lo < lo_idx) lo_idx = lo;
        }
        u64 src_end = src_abs + (off < ml ? off : ml);
        if (src_end > out_off + fenced && src_end > out_off) {      // depends on bytes this wave wrote since the last fence
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __syncthreads();
            fenced = op;
        }
        const u8 *from = dst + src_abs;
        if (off >= ml) wave_copy(out + op, from, ml, (u32)lane);
        else           { for (u32 k = lane; k < ml; k += 64) out[op + k] = from[k % off]; }
        op += ml;
    }
    if (!bad && !b.err) {
        u32 rest = b.lit_regen - lp;
        wave_copy(out + op, lits + lp, rest, (u32)lane);
        op += rest;
        if (op != b.regen) bad = true;
    }
    if (bad && lane == 0) set_err(st, ZE_CORRUPT);
    // publish: every lane's stores drained, then agent-scope release, then the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&done[bi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&st->n_exec_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Same job for frames whose blocks regenerate at most EXEC_LDS bytes each (this build's LZ-coded streams: 16 KiB blocks): the
// block is assembled in LDS, where a match that reads what the previous sequence wrote costs a barrier instead of a
// round trip through L2, and leaves as one coalesced copy.  Sequences are taken 64 at a time: every lane places the literals
// of one sequence (their positions are a prefix sum), then the matches run in order.
#define EXEC_LDS 16384u
#define EXEC_PJ_MAX 1024u                // a step's output up to this many bytes is resolved byte by byte (k_exec_seq_lds)
__device__ __forceinline__ u32 wave_excl_sum(u32 v, u32 *total)
{
    const u32 x = wave_scan_inclusive<u32, OpAdd>(v);              // (DPP row shifts: six trips through the LDS crossbar as shuffles)
    *total = (u32)__builtin_amdgcn_readlane((int)x, 63);
    return x - v;
}
__global__ __launch_bounds__(64) void k_exec_seq_lds(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, const u64 *offs, u32 nblk,
                                                      const u32 *o_ll, const u32 *o_ml, const u32 *o_of,
                                                      const u8 *lit_scratch, u8 *dst, u32 *done, ZStat *st, u32 obuf_cap)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    extern __shared__ __attribute__((aligned(16))) u8 obuf[];      // the frame's largest block with sequences + 64 (the launch sizes it: smaller blocks, more workgroups per CU)
    __shared__ u32 sh_ticket;
    const u32 lane = threadIdx.x;
    if (lane == 0) sh_ticket = atomicAdd(&st->ticket, 1u);
    __syncthreads();
    u32 t = sh_ticket;
    if (t >= n_seq_blk) return;
    u32 bi = seq_list[t];
    const ZBlock &b = blk[bi];
    const u64 out_off = b.out_off;
    const u8 *lits = lit_scratch + out_off;
    u32 rep_in[3] = { b.rep_in[0], b.rep_in[1], b.rep_in[2] };
    const u64 sbase = b.seq_base;
    u32 op = 0, lp = 0, nseq = b.err ? 0 : b.nseq;
    u32 shift = 0xFF;
    if (nblk > 1) { const u64 b0 = offs[1] - offs[0]; if (b0 && !(b0 & (b0 - 1))) shift = (u32)(63 - __builtin_clzll(b0)); }
    bool bad = b.regen > obuf_cap;
#ifdef NAF_EXEC_PROF
    unsigned long long pt[6] = { 0, 0, 0, 0, 0, 0 }; unsigned long long tA = __builtin_readcyclecounter(), tB;
#define PROF(k) { tB = __builtin_readcyclecounter(); pt[k] += tB - tA; tA = tB; }
#else
#define PROF(k)
#endif
    u16 *par = (u16 *)(obuf + obuf_cap + 64);                       // EXEC_PJ_MAX entries: where each byte of a step's output comes from
    // (the next step's triples are asked for while this one runs: three dependent trips to memory a step were a fifth of the kernel)
    u32 n_ll = 0, n_ml = 0, n_of = 0;
    if (lane < nseq) { n_ll = o_ll[sbase + lane]; n_ml = o_ml[sbase + lane]; n_of = o_of[sbase + lane]; }
    for (u32 s0 = 0; s0 < nseq && !bad; s0 += 64) {
        u32 n = nseq - s0 < 64 ? nseq - s0 : 64;
        u32 ll = 0, ml = 0, of = 0;
        if (lane < n) { ll = n_ll; ml = n_ml; of = sym_resolve(n_of, rep_in); }
        if (s0 + 64 + lane < nseq) { n_ll = o_ll[sbase + s0 + 64 + lane]; n_ml = o_ml[sbase + s0 + 64 + lane]; n_of = o_of[sbase + s0 + 64 + lane]; }
        u32 tot_all, tot_ll;
        u32 my_op = op + wave_excl_sum(ll + ml, &tot_all), my_lp = lp + wave_excl_sum(ll, &tot_ll);
        if (op + tot_all > b.regen || lp + tot_ll > b.lit_regen) { bad = true; break; }
        PROF(0)
        // literals: short runs by their own lane (up to eight bytes: ONE load, not a trip to memory per byte), long ones by the whole wave
        if (ll && ll <= 8 && my_lp + 8 <= b.regen) {
            u64 v; __builtin_memcpy(&v, lits + my_lp, 8);
            for (u32 k = 0; k < ll; k++) obuf[my_op + k] = (u8)(v >> (8 * k));
        } else
        if (ll <= 32) for (u32 k = 0; k < ll; k++) obuf[my_op + k] = lits[my_lp + k];
        u64 big = __ballot(ll > 32);
        while (big) {
            int j = __ffsll((long long)big) - 1; big &= big - 1;
            u32 l = (u32)__shfl((int)ll, j, 64), o = (u32)__shfl((int)my_op, j, 64), p = (u32)__shfl((int)my_lp, j, 64);
            for (u32 k = lane; k < l; k += 64) obuf[o + k] = lits[p + k];
        }
        __syncthreads();
        PROF(1)
        // ---- a step of short sequences whose sources all lie in this block: every BYTE finds where it comes from.  The lane-per-match
        // hops below settle a match only when its source lies wholly inside ONE earlier match; names that differ from their predecessor
        // in a digit or two break that every tenth name (...19 -> ...20 matches a byte less, ...21 a byte more: its source straddles a
        // match and the literal behind it), every later name of the step hops to that straddling source, and 85 % of a FASTQ's names went
        // one by one (46 % of the kernel, profiles/r06_exec_phases.txt).  Per byte there is nothing to straddle: a literal byte is its
        // own source, a match byte's is the byte `of` in front of it, and pointer doubling (par[p] = par[par[p]], in place: whatever a
        // lane reads is an ancestor) ends in a literal of the step or a byte in front of it after log2(chain) rounds; then one gather.
        if (tot_all <= EXEC_PJ_MAX && !__ballot(lane < n && (of == 0 || of > my_op + ll))) {
            constexpr u32 PJ = EXEC_PJ_MAX / 64;                                         // bytes a lane looks after: lane, lane + 64, ...
#pragma unroll
            for (u32 i = 0; i < PJ; i++) { const u32 p = lane + 64 * i; if (p < tot_all) par[p] = (u16)(op + p); }
            __syncthreads();
            {
                const u32 dm = my_op + ll;
                if (lane < n && ml <= 64) for (u32 k = 0; k < ml; k++) par[dm - op + k] = (u16)(dm + k - of);
                for (u64 bigm = __ballot(lane < n && ml > 64); bigm; bigm &= bigm - 1) {
                    const int j = __ffsll((long long)bigm) - 1;
                    const u32 dj = (u32)__builtin_amdgcn_readlane((int)dm, j), mlj = (u32)__builtin_amdgcn_readlane((int)ml, j), ofj = (u32)__builtin_amdgcn_readlane((int)of, j);
                    for (u32 k = lane; k < mlj; k += 64) par[dj - op + k] = (u16)(dj + k - ofj);
                }
            }
            __syncthreads();
            // (a lane's sixteen look-ups of a round are asked for together: one after the other, each behind the store of the one before, they
            // were two LDS round trips a byte and the step no faster than the matches one by one)
            u32 q[PJ];
#pragma unroll
            for (u32 i = 0; i < PJ; i++) { const u32 p = lane + 64 * i; q[i] = p < tot_all ? (u32)par[p] : 0u; }
            for (u32 round = 0; round < 12; round++) {                                    // (a chain is at most the step's 2^10 bytes long)
                u32 nq[PJ]; bool changed = false;
#pragma unroll
                for (u32 i = 0; i < PJ; i++) nq[i] = q[i] >= op ? (u32)par[q[i] - op] : q[i];
#pragma unroll
                for (u32 i = 0; i < PJ; i++) if (nq[i] != q[i]) { par[lane + 64 * i] = (u16)nq[i]; q[i] = nq[i]; changed = true; }
                __syncthreads();
                if (!__ballot(changed)) break;
            }
            {
                u8 v[PJ];
#pragma unroll
                for (u32 i = 0; i < PJ; i++) v[i] = obuf[q[i]];
#pragma unroll
                for (u32 i = 0; i < PJ; i++) { const u32 p = lane + 64 * i; if (p < tot_all && q[i] != op + p) obuf[op + p] = v[i]; }
            }
            __syncthreads();
            PROF(2)
            op += tot_all; lp += tot_ll;
            continue;
        }
        // ---- the matches of the step that need not wait for each other, all at once.  A stream of similar records (a FASTQ's read names:
        // every name copies its predecessor's first bytes) is a chain -- match j reads what match j - 1 wrote, which read what j - 2
        // wrote ... -- and one match per round trip through LDS was the executor's time (2.8 ms for the 490 MB of names of a 12.5 GB
        // FASTQ with the device to itself).  But a source that lies wholly inside an earlier plain match of the step is the same bytes
        // as the place THAT match copied from: the lane moves its source there, and again (the owner's own moved source: the hops double),
        // until the source is in front of the step, or in literal bytes, or straddles a boundary.  Sources of the first two kinds are
        // final: those matches are copied a lane each, side by side; the rest go in order below.
        u64 todo;
        {
            const bool valid = lane < n;
            const u32 dm = valid ? my_op + ll : 0xFFFFFFFFu;                             // where the match lands (ascending with the lane)
            const bool plain = valid && of >= ml && of != 0 && of <= dm;                // copies bytes of this block that do not overlap it
            u32 src = plain ? dm - of : 0;
            // (uniform control flow around every shuffle: a lane that sits out cannot be read)
            for (int round = 0; round < 7; round++) {
                const bool need = plain && src + ml > op;                                // reaches into what this step writes
                if (!__ballot(need)) break;
                // the last lane i whose match starts at or in front of src (match starts ascend): six probes
                u32 cnt = 0;
#pragma unroll
                for (u32 bit = 32; bit; bit >>= 1) { const u32 cand = cnt + bit; const u32 dc = (u32)__shfl((int)dm, (int)(cand - 1) & 63, 64); if (need && dc <= src) cnt = cand; }
                const int i = (int)cnt - 1;
                const u32 di = (u32)__shfl((int)dm, i & 63, 64), mli = (u32)__shfl((int)ml, i & 63, 64), si = (u32)__shfl((int)src, i & 63, 64);
                const bool pi = __shfl((int)plain, i & 63, 64) != 0;
                const bool hop = need && i >= 0 && (u32)i < lane && pi && src + ml <= di + mli;
                if (!__ballot(hop)) break;
                if (hop) src = si + (src - di);
            }
            PROF(2)
            // final sources: in front of the step, or between two matches of it (literal bytes)
            bool fin = plain && src + ml <= op;
            {
                const bool need = plain && !fin;
                u32 cnt = 0;
#pragma unroll
                for (u32 bit = 32; bit; bit >>= 1) { const u32 cand = cnt + bit; const u32 dc = (u32)__shfl((int)dm, (int)(cand - 1) & 63, 64); if (need && dc <= src) cnt = cand; }
                const int i = (int)cnt - 1;
                const u32 di = (u32)__shfl((int)dm, i & 63, 64), mli = (u32)__shfl((int)ml, i & 63, 64), dnext = (u32)__shfl((int)dm, (i + 1) & 63, 64);
                const u32 next_start = i + 1 < 64 ? dnext : 0xFFFFFFFFu;                  // (0xFFFFFFFF behind the step's last sequence)
                if (need) fin = (i < 0 || src >= di + mli) && src + ml <= next_start;
            }
            if (fin && ml <= 64) for (u32 k = 0; k < ml; k++) obuf[dm + k] = obuf[src + k];
            for (u64 bigm = __ballot(fin && ml > 64); bigm; bigm &= bigm - 1) {
                const int j = __ffsll((long long)bigm) - 1;
                const u32 dj = (u32)__builtin_amdgcn_readlane((int)dm, j), mlj = (u32)__builtin_amdgcn_readlane((int)ml, j), sj = (u32)__builtin_amdgcn_readlane((int)src, j);
                for (u32 k = lane; k < mlj; k += 64) obuf[dj + k] = obuf[sj + k];
            }
            todo = __ballot(valid && !fin);
            __syncthreads();
            PROF(3)
#ifdef NAF_EXEC_PROF
            pt[5] += __popcll(todo);
#endif
        }
        for (u32 j = 0; j < n; j++) {
            if (!((todo >> j) & 1)) continue;
            // (j is uniform: v_readlane, not a trip through the LDS crossbar three times per match)
            const int ju = __builtin_amdgcn_readfirstlane((int)j);
            u32 mlj = (u32)__builtin_amdgcn_readlane((int)ml, ju), ofj = (u32)__builtin_amdgcn_readlane((int)of, ju);
            u32 d = (u32)__builtin_amdgcn_readlane((int)(my_op + ll), ju);
            u64 pos_abs = out_off + d;
            if (ofj == 0 || ofj > pos_abs) { bad = true; break; }                     // reaches before the frame start
            if (ofj <= d) {                                                            // source inside this block: LDS to LDS
                const u8 *from = obuf + d - ofj;
                if (ofj >= mlj) { for (u32 k = lane; k < mlj; k += 64) obuf[d + k] = from[k]; }
                else {
                    // (the pattern's phase by adding, not by a division per byte: a block that is ONE overlapping match -- 8 KiB of `len=150`, of the number 150 -- was 6 K instructions of `%`)
                    const u32 stepr = 64u % ofj; u32 r = lane % ofj;
                    for (u32 k = lane; k < mlj; k += 64) { obuf[d + k] = from[r]; r += stepr; if (r >= ofj) r -= ofj; }
                }
            } else {
                // source (partly) in earlier blocks: confirm every block from the one holding it up to the lowest one confirmed so far, then read HBM
                u64 src_abs = pos_abs - ofj;
                u32 outside = (u32)(out_off - src_abs);                                // bytes of the pattern that lie before this block
                {
                    // only the blocks that hold those bytes (every block from there up to this one used to be confirmed: a frame whose
                    // matches reach far back ran block after block)
                    const u32 span = ofj < mlj ? ofj : mlj;
                    const u64 last = src_abs + (span < outside ? span : outside) - 1;
                    const u32 a = find_block(offs, bi, src_abs, shift);
                    const u32 z = last < offs[a + 1] ? a : find_block(offs, bi, last, shift);
                    for (u32 q = a; q <= z; q++) wait_block_done(done, q, st);
                }
                for (u32 k = lane; k < mlj; k += 64) {
                    u32 r = ofj >= mlj ? k : k % ofj;
                    obuf[d + k] = r < outside ? dst[src_abs + r] : obuf[r - outside];
                }
            }
            __syncthreads();
        }
        op += tot_all; lp += tot_ll;
        PROF(4)
    }
#ifdef NAF_EXEC_PROF
    if (lane == 0) { for (int k = 0; k < 6; k++) atomicAdd(&st->prof[k], pt[k]); atomicAdd(&st->prof[6], 1ull); atomicAdd(&st->prof[7], (unsigned long long)nseq); }
#endif
    if (!bad && !b.err) {
        u32 rest = b.lit_regen - lp;
        if (op + rest != b.regen) bad = true;
        else for (u32 k = lane; k < rest; k += 64) obuf[op + k] = lits[lp + k];
    }
    __syncthreads();
    if (!bad && !b.err) {
        u8 *out = dst + out_off; u32 nb = b.regen;
        u32 head = (u32)((16 - ((uintptr_t)out & 15)) & 15); if (head > nb) head = nb;
        if (lane < head) out[lane] = obuf[lane];
        u32 words = (nb - head) >> 4;
        for (u32 w = lane; w < words; w += 64) { uint4 v; memcpy(&v, obuf + head + 16 * w, 16); *(uint4 *)(out + head + 16 * w) = v; }
        for (u32 k = head + 16 * words + lane; k < nb; k += 64) out[k] = obuf[k];
    }
    if (bad && lane == 0) set_err(st, ZE_CORRUPT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&done[bi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&st->n_exec_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- sequence execution, 64 sequences at a time (k_exec_batch) -------------------------------------------------------------------
// k_exec_seq above walks a block's sequences one by one -- a load of the triple, a literal copy, a match copy, each behind the other,
// ~1 us per sequence -- and, worse, confirmed EVERY block between a match's source and itself before it read that source: a frame whose
// matches reach far back (a genome's repeats under libzstd's --long, or under levels that chain blocks) ran as one serial chain, block
// after block: 0.75 s for 25 MB of packed bases, 15 s for 500 MB, at which point the bounded wait called the frame corrupt
// (profiles/r05_levels_before.txt).  Here a wavefront still owns a block, but
//   * takes 64 sequences per step: triples in one coalesced load, output and literal positions by two prefix sums;
//   * copies the 64 literal runs a lane each (16 bytes at a time; long runs by the whole wavefront);
//   * splits the matches: FAR ones -- the source ends in front of what this step writes, and does not overlap the match -- are
//     independent of each other and of the step's own output: a lane each, as soon as the blocks their source lies in are done (only
//     those blocks are looked at: one or two `done` flags found by a shift when the frame's blocks are equal, a binary search otherwise);
//     NEAR ones -- source inside the step's own output, or overlapping -- go in order, the wavefront on one at a time, behind the far ones;
//   * gives up a wait only when no block of the whole launch has finished for ~4 M polls (ZStat.n_exec_done), not after a fixed count.
// Blocks are ticket-ordered as before: whatever a block waits for holds an earlier ticket, so it is running or done.
__device__ __forceinline__ void lane_copy(u8 *d, const u8 *s, u32 n)                 // n bytes by ONE lane, at any alignment
{
    u32 i = 0;
    for (; i + 16 <= n; i += 16) { uint4 v; __builtin_memcpy(&v, s + i, 16); __builtin_memcpy(d + i, &v, 16); }
    if (n & 8) { u64 v; __builtin_memcpy(&v, s + i, 8); __builtin_memcpy(d + i, &v, 8); i += 8; }
    if (n & 4) { u32 v; __builtin_memcpy(&v, s + i, 4); __builtin_memcpy(d + i, &v, 4); i += 4; }
    if (n & 2) { u16 v; __builtin_memcpy(&v, s + i, 2); __builtin_memcpy(d + i, &v, 2); i += 2; }
    if (n & 1) d[i] = s[i];
}
// blocks j0 .. j1 - 1 done, and block j1 done or at least `need` bytes into its output (prog: what a running block has published)
__device__ __forceinline__ bool source_ready(const u32 *done, const u32 *prog, u32 j0, u32 j1, u32 need)
{
    for (u32 j = j0; j < j1; j++) if (__hip_atomic_load(&done[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return false;
    return __hip_atomic_load(&done[j1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || __hip_atomic_load(&prog[j1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
}
#define EXEC_LANE_MAX 256u              // literal runs and far matches up to this many bytes are copied by their own lane
__global__ __launch_bounds__(64) void k_exec_batch(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, const u64 *offs, u32 nblk,
                                                    const u32 *o_ll, const u32 *o_ml, const u32 *o_of,
                                                    const u8 *lit_scratch, u8 *dst, u32 *done, u32 *prog, ZStat *st)
{
    __builtin_amdgcn_s_setprio(3);
    __shared__ u32 sh_ticket;
    const u32 lane = threadIdx.x;
    if (lane == 0) sh_ticket = atomicAdd(&st->ticket, 1u);
    __syncthreads();
    const u32 t = sh_ticket;
    if (t >= n_seq_blk) return;
    const u32 bi = seq_list[t];
    const ZBlock &b = blk[bi];
    const u64 out_off = b.out_off;
    u8 *out = dst + out_off;
    const u8 *lits = lit_scratch + out_off;
    u32 rep_in[3] = { b.rep_in[0], b.rep_in[1], b.rep_in[2] };
    const u64 sbase = b.seq_base;
    const u32 nseq = b.err ? 0 : b.nseq;
    u32 shift = 0xFF;
    if (nblk > 1) { const u64 b0 = offs[1] - offs[0]; if (b0 && !(b0 & (b0 - 1))) shift = (u32)(63 - __builtin_clzll(b0)); }
    u32 op = 0, lp = 0;
    bool bad = false;
    u64 since = 0; u32 spins = 0, seen = 0;                                   // the bounded wait: wave-uniform
    for (u32 s0 = 0; s0 < nseq; s0 += 64) {
        const u32 n = nseq - s0 < 64 ? nseq - s0 : 64;
        u32 ll = 0, ml = 0, of = 0;
        if (lane < n) { ll = o_ll[sbase + s0 + lane]; ml = o_ml[sbase + s0 + lane]; of = sym_resolve(o_of[sbase + s0 + lane], rep_in); }
        u32 tot_all, tot_ll;
        const u32 my_op = op + wave_excl_sum(ll + ml, &tot_all), my_lp = lp + wave_excl_sum(ll, &tot_ll);
        if ((u64)op + tot_all > b.regen || (u64)lp + tot_ll > b.lit_regen) { bad = true; break; }
        // what the earlier steps stored is visible to every lane from here on
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        // literals
        if (ll && ll <= EXEC_LANE_MAX) lane_copy(out + my_op, lits + my_lp, ll);
        for (u64 big = __ballot(ll > EXEC_LANE_MAX); big; big &= big - 1) {
            const int j = __ffsll((long long)big) - 1;
            wave_copy(out + (u32)__builtin_amdgcn_readlane((int)my_op, j), lits + (u32)__builtin_amdgcn_readlane((int)my_lp, j), (u32)__builtin_amdgcn_readlane((int)ll, j), lane);
        }
        // matches
        const bool valid = lane < n;
        const u32 d = my_op + ll;                                  // where the match lands, block-relative
        const u64 pos_abs = out_off + d;
        if (valid && (of == 0 || of > pos_abs)) bad = true;        // no offset, or one that reaches before the frame start
        if (__ballot(bad)) { bad = true; break; }
        const u64 src_abs = pos_abs - of, batch_abs = out_off + op;
        const bool far = valid && of >= ml && src_abs + ml <= batch_abs;
        // a far match's source blocks (in front of this block); inside this block the source was written by earlier steps
        const bool outside = far && src_abs < out_off;
        u32 j0 = 0, j1 = 0, need = 0;
        if (outside) {
            j0 = find_block(offs, bi, src_abs, shift);
            const u64 last = src_abs + ml - 1;
            j1 = last < out_off ? (last < offs[j0 + 1] ? j0 : find_block(offs, bi, last, shift)) : bi - 1;
            need = last < out_off ? (u32)(last + 1 - offs[j1]) : 0xFFFFFFFFu;           // (a far match ends in front of this step: last < out_off unless the step is the block's first)
        }
        bool pend = far;
        while (__ballot(pend)) {
            const bool can = pend && (!outside || source_ready(done, prog, j0, j1, need));
            const u64 m = __ballot(can);
            if (m) {
                if (__ballot(can && outside)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // what the other blocks stored, not what an old cache line says
                if (can && ml <= EXEC_LANE_MAX) lane_copy(out + d, dst + src_abs, ml);
                for (u64 big = __ballot(can && ml > EXEC_LANE_MAX); big; big &= big - 1) {
                    const int j = __ffsll((long long)big) - 1;
                    const u32 dj = (u32)__builtin_amdgcn_readlane((int)d, j), mlj = (u32)__builtin_amdgcn_readlane((int)ml, j), ofj = (u32)__builtin_amdgcn_readlane((int)of, j);
                    wave_copy(out + dj, dst + (out_off + dj - ofj), mlj, lane);
                }
                pend = pend && !can; spins = 0;
            } else {
                __builtin_amdgcn_s_sleep(4);
                if (++spins >= (1u << 22)) {                        // nothing of this step became ready for a long time: has ANY block finished meanwhile?
                    const u32 now = __hip_atomic_load(&st->n_exec_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (now != seen) { seen = now; since = 0; } else if (exec_stalled(since)) { bad = true; break; }
                    spins = 0;
                }
            }
        }
        if (bad) break;
        // near matches, in order, behind everything above
        u64 near = __ballot(valid && !far);
        if (near) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        for (; near; near &= near - 1) {
            const int j = __ffsll((long long)near) - 1;
            const u32 dj = (u32)__builtin_amdgcn_readlane((int)d, j), mlj = (u32)__builtin_amdgcn_readlane((int)ml, j), ofj = (u32)__builtin_amdgcn_readlane((int)of, j);
            const u64 sj = out_off + dj - ofj;
            if (sj < out_off) {
                // reaches into earlier blocks: those that hold [sj, min(sj + mlj, out_off)) -- wave-uniform, lane 0 polls
                const u32 a = find_block(offs, bi, sj, shift);
                const u64 last = sj + (ofj < mlj ? ofj : mlj) - 1;
                const u32 z = last < out_off ? (last < offs[a + 1] ? a : find_block(offs, bi, last, shift)) : bi - 1;
                const u32 needz = last < out_off ? (u32)(last + 1 - offs[z]) : 0xFFFFFFFFu;
                for (;;) {
                    if (source_ready(done, prog, a, z, needz)) break;
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins >= (1u << 22)) {
                        const u32 now = __hip_atomic_load(&st->n_exec_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (now != seen) { seen = now; since = 0; } else if (exec_stalled(since)) { bad = true; break; }
                        spins = 0;
                    }
                }
                if (bad) break;
                spins = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            const u8 *from = dst + sj;
            if (ofj >= mlj) wave_copy(out + dj, from, mlj, lane);
            else { for (u32 k = lane; k < mlj; k += 64) out[dj + k] = from[k % ofj]; }
            if (near & (near - 1)) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");       // the next near match may read this one
        }
        if (bad) break;
        op += tot_all; lp += tot_ll;
        // what is complete so far, for the blocks whose matches read this one: they need not wait for its end
        if (s0 + 64 < nseq) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(&prog[bi], op, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (!bad && !b.err) {
        const u32 rest = b.lit_regen - lp;
        if (op + rest != b.regen) bad = true;
        else wave_copy(out + op, lits + lp, rest, lane);
    }
    if (bad && lane == 0) set_err(st, ZE_CORRUPT);
    // publish: every lane's stores drained, then agent-scope release, then the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&done[bi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&st->n_exec_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- sequence execution as DATAFLOW (k_lz_prep, k_lz_deps, k_lz_exec) --------------------------------------------------------------
// k_exec_batch keeps a block's sequences in order: a step of 64 waits for the slowest source among its matches.  On a frame whose
// matches read all over the blocks in front of them (a genome's repeats under `--long`, or any level that matches across blocks) nearly
// every step has a source in a block that is itself still waiting, and the blocks end up running one behind the other: 0.46 ms per
// 128 KiB block whatever the device (tools/perf_exec.py: 344 ms for 200 MB of text with the waits, 3 ms without them).  What has to be
// respected is less than block order: a match may run as soon as the MATCHES that wrote its source bytes have run -- literals are final
// from the start.  So:
//   k_lz_prep   a wavefront per block: positions of its sequences (two prefix sums per 64), offsets resolved, every literal run copied;
//               per sequence: where its match lands (block-relative), its final offset, and a `done` byte (0 = a match still to run);
//   k_lz_deps   per match: the range [lo, lo + n) of earlier sequences whose match bytes intersect its source -- the first sequence whose
//               match ends behind the source's first byte, the last whose match starts in front of its end (two searches in the arrays
//               of the blocks that hold those bytes);
//   k_lz_exec   a wavefront per block, ticket-ordered, sweeps over its pending matches: whatever has all its `done` bytes set is copied
//               -- a lane per match, long ones by the wavefront -- and marked done; the others stay for the next sweep.
// Everything the executing wavefronts exchange -- match bytes and `done` bytes -- moves with agent-scope (sc1) loads and stores, which
// bypass the caches that are not coherent between XCDs: no release / acquire fence in the loop (an agent-scope acquire invalidates an
// XCD's whole L2 share; tools/perf_exec.py: the fences alone were five times the copies).  The earliest pending match of the frame always
// has its sources done, so a valid frame always moves; a wavefront gives up when nothing anywhere has run for a long time.
// (the bulk copies below use these at ANY byte alignment.  A relaxed agent-scope atomic of 2 / 4 / 8 bytes lowers on gfx950 to one
// global_load / global_store with the sc1 bit, and gfx950 handles unaligned global accesses in hardware (common.h: ld64) -- single-copy
// atomicity is not asked of the match bytes, only that they bypass the per-XCD caches; the `done` flags, which ARE synchronisation, are
// single bytes.  Another target would have to give the bytes a cache-bypassing load of their own: pinned here.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "ld_sc1 / st_sc1: unaligned sc1 accesses as relaxed atomics are a gfx950 lowering; port lane_copy_sc1 / wave_copy_sc1 before building for another target"
#endif
template <typename T> __device__ __forceinline__ T ld_sc1(const void *p) { return __hip_atomic_load((const T *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void st_sc1(void *p, T v) { __hip_atomic_store((T *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lane_copy_sc1(u8 *d, const u8 *s, u32 n)            // n bytes by ONE lane, any alignment, source clear of the destination
{
    // (the two loads of a step in flight together, and a tail of 1 .. 15 bytes as two pieces that may overlap -- [i, i + w) and [n - w, n) --
    // instead of 8 + 4 + 2 + 1 one behind the other: a seven-byte copy was three round trips through memory, now one)
    u32 i = 0;
    for (; i + 16 <= n; i += 16) { const u64 a = ld_sc1<u64>(s + i), b = ld_sc1<u64>(s + i + 8); st_sc1<u64>(d + i, a); st_sc1<u64>(d + i + 8, b); }
    const u32 rem = n - i;
    if (rem >= 8) { const u64 a = ld_sc1<u64>(s + i), b = ld_sc1<u64>(s + n - 8); st_sc1<u64>(d + i, a); st_sc1<u64>(d + n - 8, b); }
    else if (rem >= 4) { const u32 a = ld_sc1<u32>(s + i), b = ld_sc1<u32>(s + n - 4); st_sc1<u32>(d + i, a); st_sc1<u32>(d + n - 4, b); }
    else if (rem >= 2) { const u16 a = ld_sc1<u16>(s + i), b = ld_sc1<u16>(s + n - 2); st_sc1<u16>(d + i, a); st_sc1<u16>(d + n - 2, b); }
    else if (rem) st_sc1<u8>(d + i, ld_sc1<u8>(s + i));
}
__device__ __forceinline__ void wave_copy_sc1(u8 *d, const u8 *s, u32 n, u32 lane)  // the same by 64 lanes
{
    // (four loads in flight per lane before the first store: one at a time a 128 KiB copy -- a block that repeats its predecessor -- was 256
    // round trips through memory in a row, 80 us)
    u32 i = lane * 8;
    for (; i + 3 * 512 + 8 <= n; i += 4 * 512) {
        const u64 a = ld_sc1<u64>(s + i), b = ld_sc1<u64>(s + i + 512), c = ld_sc1<u64>(s + i + 1024), e = ld_sc1<u64>(s + i + 1536);
        st_sc1<u64>(d + i, a); st_sc1<u64>(d + i + 512, b); st_sc1<u64>(d + i + 1024, c); st_sc1<u64>(d + i + 1536, e);
    }
    for (; i + 8 <= n; i += 64 * 8) st_sc1<u64>(d + i, ld_sc1<u64>(s + i));
    const u32 done = n & ~7u;
    if (lane < (n & 7u)) st_sc1<u8>(d + done + lane, ld_sc1<u8>(s + done + lane));
}
struct LzArrays { u32 *x_dst; u32 *ml; u32 *of; u32 *dep_lo; u32 *dep_n; u8 *sdone; u8 *stail; };    // x_dst: the sequence arrays' `ll` slot, rewritten
// A match that overlaps itself (offset < length: a run of a repeated unit -- a FASTQ's "len=150" names and its lengths under libzstd are
// ONE such match per 128 KiB block, each continuing the run of the block before) writes its last LZ_TAIL bytes first and says so in
// `stail`; a match whose source lies in those bytes of one such match (the next block's eight-byte seed) waits for that, not for the
// 128 KiB in front of it: the chain through the blocks of a run costs a round trip and a few stores per link instead of a block's copy
// (names of 2 GB of reads, 376 blocks: 206 ms a byte a lane and a block per link, 31 ms with the doubling copy below, 2.9 ms with the tail).
#define LZ_TAIL 256u
#define LZ_DEP_TAIL 0x80000000u
__global__ __launch_bounds__(64) void k_lz_prep(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, const u64 *seq_cnt, u32 nblk, u64 ns_total,
                                                 LzArrays A, const u8 *lit_scratch, u8 *dst, ZStat *st)
{
    const u32 t = blockIdx.x, lane = threadIdx.x;
    if (t >= n_seq_blk) return;
    const u32 bi = seq_list[t];
    const ZBlock &b = blk[bi];
    const u64 out_off = b.out_off, sbase = b.seq_base;
    u8 *out = dst + out_off;
    const u8 *lits = lit_scratch + out_off;
    u32 rep_in[3] = { b.rep_in[0], b.rep_in[1], b.rep_in[2] };
    const u32 nseq = b.err ? 0 : b.nseq;
    const u32 cnt = (u32)((bi + 1 < nblk ? seq_cnt[bi + 1] : ns_total) - seq_cnt[bi]);     // with the padding up to a multiple of four
    u32 op = 0, lp = 0;
    bool bad = false;
    for (u32 s0 = 0; s0 < cnt; s0 += 64) {
        const bool real = s0 + lane < nseq && !bad;                         // (behind an error every entry gets its no-op form: nothing may wait for it)
        u32 ll = 0, ml = 0, of = 0;
        if (real) { ll = A.x_dst[sbase + s0 + lane]; ml = A.ml[sbase + s0 + lane]; of = sym_resolve(A.of[sbase + s0 + lane], rep_in); }
        u32 tot_all, tot_ll;
        const u32 my_op = op + wave_excl_sum(ll + ml, &tot_all), my_lp = lp + wave_excl_sum(ll, &tot_ll);
        if ((u64)op + tot_all > b.regen || (u64)lp + tot_ll > b.lit_regen) { bad = true; ll = 0; ml = 0; }
        if (ll && ll <= EXEC_LANE_MAX) lane_copy(out + my_op, lits + my_lp, ll);
        for (u64 big = __ballot(ll > EXEC_LANE_MAX); big; big &= big - 1) {
            const int j = __ffsll((long long)big) - 1;
            wave_copy(out + (u32)__builtin_amdgcn_readlane((int)my_op, j), lits + (u32)__builtin_amdgcn_readlane((int)my_lp, j), (u32)__builtin_amdgcn_readlane((int)ll, j), lane);
        }
        const u32 d = my_op + ll;
        if (real && ml && (of == 0 || of > out_off + d)) { bad = true; ml = 0; }       // no offset, or one that reaches before the frame start
        if (s0 + lane < cnt) {
            const u64 i = sbase + s0 + lane;
            A.x_dst[i] = real && ml ? d : 0xFFFFFFFFu; A.ml[i] = ml; A.of[i] = of;
            A.sdone[i] = ml ? 0 : 1; A.stail[i] = (ml && of < ml) ? 0 : 1;
        }
        bad = __ballot(bad) != 0;
        if (!bad) { op += tot_all; lp += tot_ll; }
    }
    if (!__ballot(bad) && !b.err) {
        const u32 rest = b.lit_regen - lp;
        if (op + rest != b.regen) bad = true;
        else wave_copy(out + op, lits + lp, rest, lane);
    }
    if (__ballot(bad) && lane == 0) set_err(st, ZE_CORRUPT);
}
// ---- chains of copies (k_lz_collapse) ----------------------------------------------------------------------------------------------
// Records that resemble their neighbours -- a FASTQ's read names under libzstd: every name copies the first bytes of the one before --
// make chains: match i reads what match i - 1 wrote, which read what i - 2 wrote ...  As dataflow that is one round trip through memory
// per LINK (the reference's archive of 2 GB of reads: 210 ms in k_lz_exec for chains of ten thousand names per block).  But a source
// that lies wholly inside an earlier PLAIN match holds the same bytes as the place that match copied from: the match may as well read
// there, and from where THAT place was copied from.  Per unit of 1024 sequences (the units of k_lz_exec), in LDS: every match whose
// source lies inside a match of its unit moves it to that match's own current source, all matches at once, round after round -- the
// hops double, ten rounds cover a unit -- until the source leaves the unit, falls into literals, or straddles a boundary.  The moved
// source is written back as a larger offset; k_lz_deps and k_lz_exec never know.  Chains that cross units keep one link per unit.
#define LZ_UNIT 1024u
// (the collapse has units of its own: 8192 sequences -- 96 KB of LDS, a workgroup of 256 per CU -- because what it leaves is a link per unit
// edge, and a frame that is one chain, a counter's names through every block, is then as long as its unit edges are many)
// (two shapes: 8192 sequences by 1024 threads, 96 KB of LDS -- a CU to itself -- for the frames that are chains; 1024 by a single wavefront
// and 12 KB for every other frame, because a workgroup that needs sixteen wave slots and most of a CU's LDS AT ONCE does not get in beside a
// long-running kernel: in the decode of the reference's archive of 10 GB the job's collapse waited 1.6 ms for the flat emit to end, §4.32)
#define LZ_CUNIT_BIG 8192u
#define LZ_CWG_BIG 1024u
#define LZ_CUNIT_SMALL 1024u
#define LZ_CWG_SMALL 64u
#define LZ_DEPS_PARTS 8u
#define LZ_FAR_ROUNDS 14u
template <u32 LZ_CUNIT, u32 LZ_CWG>
__global__ __launch_bounds__(LZ_CWG) void k_lz_collapse(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, const u64 *unit_base, const u64 *n_units, LzArrays A, u32 *hops)
{
    extern __shared__ __attribute__((aligned(16))) u32 lzc[];              // s_dst | s_ml | s_src, LZ_CUNIT each
    u32 *s_dst = lzc, *s_ml = lzc + LZ_CUNIT, *s_src = lzc + 2 * LZ_CUNIT;
    const u64 u = blockIdx.x;
    if (u >= *n_units) return;
    u32 t = 0;
    { u32 hi = n_seq_blk; while (t + 1 < hi) { const u32 mid = (t + hi) >> 1; if (unit_base[mid] <= u) t = mid; else hi = mid; } }
    const ZBlock &b = blk[seq_list[t]];
    const u32 s_first = (u32)(u - unit_base[t]) * LZ_CUNIT, nseq = b.err ? 0 : b.nseq;
    if (s_first >= nseq) return;
    const u32 cnt = nseq - s_first < LZ_CUNIT ? nseq - s_first : LZ_CUNIT;
    const u64 sbase = b.seq_base + s_first;
    for (u32 idx = threadIdx.x; idx < cnt; idx += LZ_CWG) {
        const u32 d = A.x_dst[sbase + idx], ml = A.ml[sbase + idx], of = A.of[sbase + idx];
        const bool plain = ml && of >= ml && of <= d;                            // a copy of bytes of this block that do not overlap it
        s_dst[idx] = d; s_ml[idx] = plain ? ml : 0; s_src[idx] = plain ? d - of : 0xFFFFFFFFu;
    }
    __syncthreads();
    const u32 first_dst = s_dst[0];
    bool moved_any = false;
    // (a lane writes only its own sources and may read another's while it moves: either value names the same bytes)
    for (int round = 0; round < 16; round++) {
        bool hop = false;
        for (u32 idx = threadIdx.x; idx < cnt; idx += LZ_CWG) {
            const u32 ml = s_ml[idx], s = s_src[idx];
            if (!ml || s == 0xFFFFFFFFu || s < first_dst) continue;              // not a plain copy, or a source in front of the unit
            u32 lo = 0, hi = idx;                                                // the last sequence j < idx whose match starts at or in front of s
            while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (s_dst[mid] <= s) lo = mid + 1; else hi = mid; }
            if (!lo) continue;
            const u32 j = lo - 1, dj = s_dst[j], mlj = s_ml[j], sj = s_src[j];
            if (mlj && sj != 0xFFFFFFFFu && s + ml <= dj + mlj) { s_src[idx] = sj + (s - dj); hop = true; }
        }
        if (!__syncthreads_or(hop)) break;
        moved_any = true;
    }
    if (!moved_any) return;
    u32 n_moved = 0;                                             // matches of the unit that read somewhere else now: what k_lz_collapse_far asks before it starts
    for (u32 idx = threadIdx.x; idx < cnt; idx += LZ_CWG) if (s_ml[idx]) {
        const u32 nof = s_dst[idx] - s_src[idx];
        if (nof != A.of[sbase + idx]) { A.of[sbase + idx] = nof; n_moved++; }
    }
#pragma unroll
    for (u32 sh = 32; sh; sh >>= 1) n_moved += (u32)__shfl_xor((int)n_moved, (int)sh, 64);
    if ((threadIdx.x & 63) == 0 && n_moved) atomicAdd(hops, n_moved);
}
// ---- ... and across units (k_lz_collapse_far) -----------------------------------------------------------------------------------------
// What the collapse above leaves of a chain is a link per unit edge and block edge: the ids of the reference's archive of 2 GB of reads,
// every name a copy of the name in front of it through all 556 blocks, were 18.7 ms in k_lz_exec -- 10 us a link.  The same jump from unit
// to unit: a unit takes the source of its first match, finds the unit T that holds that place (blocks by their out_off, units by their
// first landing place), stages T's matches in LDS, then T + 1's (a unit's sources span about a unit), and every match of its own whose
// source lies wholly inside a plain match j of those adds j's offset to its own (its bytes are the bytes j copied); then the same from
// the first match whose source lay in neither, LZ_FAR_TARGETS times a round -- names that count have chains at 1000, 10 000, 100 000
// names' distance in the same unit, and the unit itself is one of the places: a match of it that has moved takes those behind it along
// (the ids above: depth 1071 staging one place a round and never the unit itself, 486 with four and itself; 466 is what hops through
// WHOLE matches can reach there -- the rest reads a match and the literal behind it).  One launch, a workgroup per unit, each going round by itself until a round moves nothing (at most LZ_FAR_ROUNDS; a unit that
// could not move never will: its first match, the two units it stages and the edges of what they hold stay the same) -- no barrier
// between units: T's offsets may be moving while they are read, and either value names the same bytes.  Units start in order, so
// most find the units in front of them finished and reach the chain's root in a hop or two; those in flight together double their hops.
// Only for frames that are chains (most MATCHES moved their source within their unit: hops[0] -- a `-3 --long 27` genome, where every
// unit has a few that do, paid 6.9 ms here to save 3 in the executor).  Sources in other units than those two keep
// their links: k_lz_exec works them off as before.
template <u32 LZ_CUNIT>
__device__ __forceinline__ void lz_locate_unit(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, const u64 *unit_base, u64 u, u32 &t, u32 &s_first, u32 &cnt)
{
    t = 0;
    { u32 hi = n_seq_blk; while (t + 1 < hi) { const u32 mid = (t + hi) >> 1; if (unit_base[mid] <= u) t = mid; else hi = mid; } }
    const ZBlock &b = blk[seq_list[t]];
    const u32 nseq = b.err ? 0 : b.nseq;
    s_first = (u32)(u - unit_base[t]) * LZ_CUNIT;
    cnt = s_first >= nseq ? 0u : (nseq - s_first < LZ_CUNIT ? nseq - s_first : LZ_CUNIT);
}
#define LZ_FAR_TARGETS 4u
template <u32 LZ_CUNIT, u32 LZ_CWG>
__global__ __launch_bounds__(LZ_CWG) void k_lz_collapse_far(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, const u64 *unit_base, const u64 *n_units, LzArrays A, u32 *hops, u64 ns_total)
{
    extern __shared__ __attribute__((aligned(16))) u32 lzc[];              // s_dst | s_ml | s_of of the staged unit, LZ_CUNIT each
    u32 *s_dst = lzc, *s_ml = lzc + LZ_CUNIT, *s_of = lzc + 2 * LZ_CUNIT;
    __shared__ u32 s_pick, s_geo[4];
    __shared__ u64 s_pos[3];
    constexpr u32 PER = LZ_CUNIT / LZ_CWG;
    const u64 u = blockIdx.x, nu = *n_units;
    if (u >= nu) return;
    if (4 * (u64)hops[0] < ns_total) return;                     // worth it for frames that are chains: a quarter of the matches and more moved their source within their unit (names that count: 57 %; a genome's repeats: under 1 %)
    u32 t, s_first, cnt; lz_locate_unit<LZ_CUNIT>(blk, seq_list, n_seq_blk, unit_base, u, t, s_first, cnt);
    if (!cnt) return;
    const ZBlock &b = blk[seq_list[t]];
    const u64 sbase = b.seq_base + s_first, B_u = b.out_off;
    u32 d[PER], ml[PER], of[PER]; u32 elig = 0;
#pragma unroll
    for (u32 k = 0; k < PER; k++) {
        const u32 idx = threadIdx.x + k * LZ_CWG;
        d[k] = 0xFFFFFFFFu; ml[k] = 0; of[k] = 0;
        if (idx < cnt) { d[k] = A.x_dst[sbase + idx]; ml[k] = A.ml[sbase + idx]; of[k] = A.of[sbase + idx]; }
        if (ml[k] && of[k] >= ml[k] && d[k] != 0xFFFFFFFFu) elig |= 1u << k;      // a plain copy
    }
    u32 rounds = 0;
    for (; rounds < LZ_FAR_ROUNDS; rounds++) {
        bool moved = false;
        u32 cov = 0;                                             // matches whose source lay in a unit staged this round
        for (u32 tg = 0; tg < LZ_FAR_TARGETS; tg++) {
            // the first match not looked at yet this round: where it reads is the next unit to stage
            if (threadIdx.x == 0) s_pick = 0xFFFFFFFFu;
            __syncthreads();
            u32 mine = 0xFFFFFFFFu;
#pragma unroll
            for (u32 k = 0; k < PER; k++) if (((elig & ~cov) >> k) & 1) { const u32 idx = threadIdx.x + k * LZ_CWG; mine = idx < mine ? idx : mine; }
#pragma unroll
            for (u32 sh = 32; sh; sh >>= 1) { const u32 o = (u32)__shfl_xor((int)mine, (int)sh, 64); mine = o < mine ? o : mine; }
            if ((threadIdx.x & 63) == 0 && mine != 0xFFFFFFFFu) atomicMin(&s_pick, mine);
            __syncthreads();
            const u32 pick = s_pick;
            if (pick == 0xFFFFFFFFu) break;
#pragma unroll
            for (u32 k = 0; k < PER; k++) if (threadIdx.x + k * LZ_CWG == pick) { s_pos[0] = B_u + d[k] - of[k]; cov |= 1u << k; }
            __syncthreads();
            if (threadIdx.x == 0) {
                // the block that holds that place, then the unit of it
                const u64 P = s_pos[0];
                u32 tb = 0; { u32 hi = t + 1; while (tb + 1 < hi) { const u32 mid = (tb + hi) >> 1; if (blk[seq_list[mid]].out_off <= P) tb = mid; else hi = mid; } }
                const ZBlock &q = blk[seq_list[tb]];
                const u32 nsq = q.err ? 0 : q.nseq, nun = (nsq + LZ_CUNIT - 1) / LZ_CUNIT;
                const u64 rel = P - q.out_off;
                u32 k = 0; { u32 hi = nun; while (k + 1 < hi) { const u32 mid = (k + hi) >> 1; if ((u64)A.x_dst[q.seq_base + (u64)mid * LZ_CUNIT] <= rel) k = mid; else hi = mid; } }
                s_pos[1] = unit_base[tb] + k;
            }
            __syncthreads();
            const u64 T0 = s_pos[1];
            for (u32 pass = 0; pass < 2; pass++) {
                const u64 T = T0 + pass;
                if (T > u) break;                                // (the unit itself is a target like any other: a match in front of this one that has moved since)
                if (threadIdx.x == 0) {
                    u32 t2, sf2, c2; lz_locate_unit<LZ_CUNIT>(blk, seq_list, n_seq_blk, unit_base, T, t2, sf2, c2);
                    s_geo[0] = t2; s_geo[1] = sf2; s_geo[2] = c2;
                }
                __syncthreads();
                const u32 t2 = s_geo[0], sf2 = s_geo[1], c2 = s_geo[2];
                const ZBlock &q = blk[seq_list[t2]];
                const u64 sb2 = q.seq_base + sf2, B_T = q.out_off;
                for (u32 idx = threadIdx.x; idx < c2; idx += LZ_CWG) {
                    const u32 dj = A.x_dst[sb2 + idx], mj = A.ml[sb2 + idx], oj = __hip_atomic_load(A.of + sb2 + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_dst[idx] = dj; s_ml[idx] = (mj && oj >= mj) ? mj : 0; s_of[idx] = oj;
                    if (idx + 1 == c2) s_pos[2] = dj == 0xFFFFFFFFu ? 0 : (u64)dj + mj;          // where the unit's matches end
                }
                __syncthreads();
                if (c2 && s_dst[0] != 0xFFFFFFFFu) {
                    const u64 lo_abs = B_T + s_dst[0], hi_abs = B_T + s_pos[2];
#pragma unroll
                    for (u32 k = 0; k < PER; k++) {
                        if (!(((elig & ~cov) >> k) & 1)) continue;
                        const u64 sabs = B_u + d[k] - of[k];
                        if (sabs < lo_abs || sabs >= hi_abs) continue;
                        cov |= 1u << k;
                        const u32 srel = (u32)(sabs - B_T);
                        u32 lo = 0, hi = c2;                                         // the last match of the staged unit that lands at or in front of srel
                        while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (s_dst[mid] <= srel) lo = mid + 1; else hi = mid; }
                        if (!lo) continue;
                        const u32 j = lo - 1, dj = s_dst[j], mj = s_ml[j], oj = s_of[j];
                        if (!mj || (u64)srel + ml[k] > (u64)dj + mj) continue;
                        const u32 nof = of[k] + oj;
                        if (nof < of[k] || nof >= 0x80000000u) continue;           // (offsets stay 31-bit)
                        of[k] = nof; moved = true;
                        __hip_atomic_store(A.of + sbase + threadIdx.x + k * LZ_CWG, nof, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                __syncthreads();
            }
        }
        if (!__syncthreads_or(moved)) break;
    }
    if (rounds && threadIdx.x == 0) atomicAdd(hops + 1 + (rounds < LZ_FAR_ROUNDS ? rounds : LZ_FAR_ROUNDS), 1u);      // (tracing: units by the rounds they took)
}
// first j in [0, n) with x_dst[j] + ml[j] > rel (match ends are increasing; padding entries: 0xFFFFFFFF + 0)
__device__ __forceinline__ u32 lz_first_end_after(const u32 *x, const u32 *m, u32 n, u32 rel)
{
    u32 lo = 0, hi = n;
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if ((u64)x[mid] + m[mid] > rel) hi = mid; else lo = mid + 1; }
    return lo;
}
// number of j in [0, n) with x_dst[j] < rel_end (match starts are increasing)
__device__ __forceinline__ u32 lz_count_start_before(const u32 *x, u32 n, u32 rel_end)
{
    u32 lo = 0, hi = n;
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (x[mid] < rel_end) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(64) void k_lz_deps(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, const u64 *offs, const u64 *seq_cnt, u32 nblk, u64 ns_total, LzArrays A)
{
    const u32 t = blockIdx.x, lane = threadIdx.x;
    if (t >= n_seq_blk) return;
    const u32 bi = seq_list[t];
    const ZBlock &b = blk[bi];
    const u64 out_off = b.out_off, sbase = b.seq_base;
    const u32 nseq = b.err ? 0 : b.nseq;
    u32 shift = 0xFF;
    if (nblk > 1) { const u64 b0 = offs[1] - offs[0]; if (b0 && !(b0 & (b0 - 1))) shift = (u32)(63 - __builtin_clzll(b0)); }
    // (gridDim.y wavefronts share a block: a frame of a few hundred blocks of ten thousand sequences each was 2 ms of one wavefront's
    // dependent searches per block)
    for (u32 s = blockIdx.y * 64 + lane; s < nseq; s += 64 * gridDim.y) {
        const u64 i = sbase + s;
        const u32 ml = A.ml[i];
        if (!ml) { A.dep_lo[i] = 0; A.dep_n[i] = 0; continue; }
        const u32 d = A.x_dst[i], of = A.of[i];
        const u64 src = out_off + d - of, src_end = src + (of < ml ? of : ml);          // the bytes that exist before the match runs
        // the blocks of the first and of the last source byte (this block among them)
        const u32 ja = src >= out_off ? bi : find_block(offs, bi, src, shift);
        const u32 jb = src_end - 1 >= out_off ? bi : (src_end - 1 < offs[ja + 1] ? ja : find_block(offs, bi, src_end - 1, shift));
        const u64 base_a = seq_cnt[ja], base_b = seq_cnt[jb];
        const u32 cnt_a = (u32)((ja + 1 < nblk ? seq_cnt[ja + 1] : ns_total) - base_a), cnt_b = ja == jb ? cnt_a : (u32)((jb + 1 < nblk ? seq_cnt[jb + 1] : ns_total) - base_b);
        // (in this block only the sequences in front of this one count: its own match starts at d >= the source's end)
        const u64 lo = base_a + lz_first_end_after(A.x_dst + base_a, A.ml + base_a, ja == bi ? s : cnt_a, (u32)(src - offs[ja]));
        const u64 hi = base_b + lz_count_start_before(A.x_dst + base_b, jb == bi ? s : cnt_b, (u32)(src_end - offs[jb]));     // one past the last
        u64 lo2 = lo, hi2 = hi;                                                    // (without the padding entries at the blocks' ends: they are done from the start)
        while (hi2 > lo2 && A.ml[hi2 - 1] == 0) hi2--;
        while (hi2 > lo2 && A.ml[lo2] == 0) lo2++;
        u32 n = hi2 > lo2 ? (u32)(hi2 - lo2) : 0;
        if (n == 1) {
            // written by ONE match: when that one overlaps itself and the source starts in its last LZ_TAIL bytes, its tail is all this one waits for
            const bool in_b = lo2 >= base_b && lo2 < base_b + cnt_b, in_a = lo2 >= base_a && lo2 < base_a + cnt_a;
            if (in_a || in_b) {
                const u32 mlj = A.ml[lo2], ofj = A.of[lo2];
                const u64 endj = offs[in_b ? jb : ja] + A.x_dst[lo2] + mlj;
                if (ofj < mlj && src + (mlj < LZ_TAIL ? mlj : LZ_TAIL) >= endj) n |= LZ_DEP_TAIL;      // (what of the source lies behind that match's end is not its bytes)
            }
        }
        A.dep_lo[i] = (u32)lo2; A.dep_n[i] = n;
    }
}
__device__ __forceinline__ bool lz_deps_done(const u8 *sdone, u32 lo, u32 n)
{
    // (the first four without a branch between them: independent loads in flight together; most matches have one to three)
    bool ok = true;
    if (n > 0) ok &= ld_sc1<u8>(sdone + lo) != 0;
    if (n > 1) ok &= ld_sc1<u8>(sdone + lo + 1) != 0;
    if (n > 2) ok &= ld_sc1<u8>(sdone + lo + 2) != 0;
    if (n > 3) ok &= ld_sc1<u8>(sdone + lo + 3) != 0;
    if (n <= 4 || !ok) return ok;
    u32 k = 4;
    for (; k + 8 <= n; k += 8) if (ld_sc1<u64>(sdone + lo + k) != 0x0101010101010101ull) return false;
    for (; k < n; k++) if (!ld_sc1<u8>(sdone + lo + k)) return false;
    return true;
}
// ---- runs that continue from block to block (k_lz_runs_mark, k_lz_runs_check, k_lz_runs_apply) ----------------------------------------
// What libzstd makes of a FASTQ's lengths ("150" as a u32, a few hundred MB of it) and of names that repeat ("len=150\0"): per 128 KiB
// block ONE literal and ONE match that overlaps itself -- offset p, the period -- whose source is the last p - 1 bytes of the block in
// front and that literal, block after block: a chain with a link per block even with the tail trick above (30 - 40 us a link: 48 ms for
// the lengths of 12.5 GB of reads).  But every byte of such a run is a byte of the p-byte SEED in front of the run's first match,
// out[x] = seed[(x - seed) mod p] -- as long as the literals on the way are what the pattern says they are.  So, behind k_lz_deps: a
// block whose first sequence has fewer than p literals and overlaps itself with the offset of the last sequence of the block in
// front, which runs to that block's end, is a LINK (k_lz_runs_mark; a running maximum names the block whose last match the run began
// with); where that match's seed is final from the start, every link compares its literals with the pattern (k_lz_runs_check); a
// link with no mismatch between the run's first block and itself becomes "fill with period p from the seed, in phase" and waits for
// nothing (k_lz_runs_apply): every block of the run goes at once.  Encoded for k_lz_exec (the only reader behind this point) as
// of = distance from the match to the seed, ml |= p << 18 (a match is at most 131074 bytes: 18 bits).
#define LZ_PER_SHIFT 18u
#define LZ_PER_MAX 8191u
#define LZ_ML(x) ((x) & ((1u << LZ_PER_SHIFT) - 1))
// (Two shapes of link in libzstd's frames: the match overlaps itself with the run's period as its offset -- the first blocks -- or, once
// the window holds a whole block, copies the block in front, offset 131072: any multiple of the period whose source lies inside the run
// is the same bytes.)
__global__ void k_lz_runs_mark(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, LzArrays A, i32 *through, u32 *lnk)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seq_blk) return;
    // lnk[t] = 1 when block t's first match MAY continue a run that reaches it: the block in front is the one in front in the frame and
    // ends with a match (k_lz_runs_check looks at the rest once the run's period is known); through[t] = -1 when block t would
    // moreover hand the run on (that match is its only one and runs to its last byte), else t: a run that goes on behind block t starts
    // with block t's last match.
    const u32 bi = seq_list[t]; const ZBlock &b = blk[bi];
    u32 cand = 0; bool thr = false;
    if (t && !b.err && b.nseq && seq_list[t - 1] + 1 == bi) {
        const ZBlock &a = blk[seq_list[t - 1]];
        if (!a.err && a.nseq && a.out_off + a.regen == b.out_off) {
            const u64 i = b.seq_base, j = a.seq_base + a.nseq - 1;
            const u32 ml = A.ml[i], d = A.x_dst[i], mlj = A.ml[j];
            if (ml && d <= LZ_PER_MAX && mlj && A.x_dst[j] + mlj == a.regen) { cand = 1; thr = b.nseq == 1 && d + ml == b.regen; }
        }
    }
    lnk[t] = cand; through[t] = thr ? -1 : (i32)t;
}
// the run that reaches block t, if any: it began with the last match of block h = head[t - 1] -- one that overlaps itself (its offset is the
// period), runs to its block's end and has a seed that is final from the start
__device__ __forceinline__ bool lz_run_of(const ZBlock *blk, const u32 *seq_list, const LzArrays &A, i32 h, u32 &per, u64 &seed)
{
    if (h < 0) return false;
    const ZBlock &hb = blk[seq_list[h]];
    if (hb.err || !hb.nseq) return false;
    const u64 ih = hb.seq_base + hb.nseq - 1;
    const u32 mlh = A.ml[ih], ofh = A.of[ih];                         // (never a sequence k_lz_runs_apply rewrites while this is read: see there)
    if (!mlh || (mlh >> LZ_PER_SHIFT) || ofh >= mlh || ofh > LZ_PER_MAX || A.x_dst[ih] + mlh != hb.regen || A.dep_n[ih] != 0) return false;
    per = ofh; seed = hb.out_off + A.x_dst[ih] - ofh;
    return true;
}
// bad[t] = t when block t is no link of the run that reaches it -- its offset is no multiple of the period, its source begins in front of
// the seed, it has a period's worth of literals or more, or its literals differ from the pattern -- else -1; the running maximum says
// whether a link behind it may be rewritten
__global__ void k_lz_runs_check(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, LzArrays A, const i32 *head, const u32 *lnk, const u8 *dst, i32 *bad)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seq_blk) return;
    i32 v = -1;
    if (t && lnk[t]) {
        v = (i32)t;
        u32 per; u64 seed;
        if (lz_run_of(blk, seq_list, A, head[t - 1], per, seed)) {
            const ZBlock &b = blk[seq_list[t]];
            const u32 ll = A.x_dst[b.seq_base], of = A.of[b.seq_base];
            if (ll < per && of % per == 0 && b.out_off + ll >= seed + of) {
                bool same = true;
                for (u32 q = 0; q < ll; q++) same = same && dst[b.out_off + q] == dst[seed + (b.out_off + q - seed) % per];
                if (same) v = -1;
            }
        }
    }
    bad[t] = v;
}
__global__ void k_lz_runs_apply(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, LzArrays A, const i32 *head /* running maximum of `through` */, const u32 *lnk, const i32 *lastbad /* of `bad` */)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 || t >= n_seq_blk || !lnk[t]) return;
    const i32 h = head[t - 1];
    if (h < 0 || lastbad[t] > h) return;                              // (a block between the run's first and this one, or this one, is no link of it)
    // (the run's first match is the LAST match of a block that is not handed through: if it is that block's first match as well it does
    // not run to the block's end, and lz_run_of turns the run down whatever this kernel has made of it meanwhile)
    u32 per; u64 seed;
    if (!lz_run_of(blk, seq_list, A, h, per, seed)) return;
    const ZBlock &b = blk[seq_list[t]];
    const u64 i = b.seq_base;
    const u64 dist = b.out_off + A.x_dst[i] - seed;
    if (dist >= 0x80000000ull) return;
    A.dep_lo[i] = 0; A.dep_n[i] = 0;                                  // (the seed is final)
    A.of[i] = (u32)dist; A.ml[i] = LZ_ML(A.ml[i]) | (per << LZ_PER_SHIFT);
}
// A wavefront owns a UNIT of U x 64 consecutive sequences of one block (not the whole block: a sweep over a block's hundred words took
// as long as a hundred round trips, and that was the time a level of the dependency graph cost).  Units are numbered in frame order
// (unit_base: exclusive sum of the blocks' unit counts, k_lz_units) and taken by ticket: what a unit waits for lies in units with
// earlier tickets or in itself.  The unit's sequences stay in registers; a sweep is one round of `done` loads for everything pending.
__global__ void k_lz_units(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, u32 per_unit, u64 *units)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seq_blk) return;
    const ZBlock &b = blk[seq_list[t]];
    units[t] = b.err ? 0 : (b.nseq + per_unit - 1) / per_unit;
}
template <u32 U>
__global__ __launch_bounds__(64) void k_lz_exec(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, const u64 *unit_base, const u64 *n_units, LzArrays A, u8 *dst, ZStat *st)
{
    __shared__ u32 sh_ticket;
    const u32 lane = threadIdx.x;
    if (lane == 0) sh_ticket = atomicAdd(&st->ticket, 1u);
    __syncthreads();
    const u64 u = sh_ticket;
    if (u >= *n_units) return;
    if (ld_sc1<u32>(&st->err)) return;                                       // k_lz_prep found the frame corrupt: the call fails, nothing to copy
    u32 t = 0;
    { u32 hi = n_seq_blk; while (t + 1 < hi) { const u32 mid = (t + hi) >> 1; if (unit_base[mid] <= u) t = mid; else hi = mid; } }      // last block with unit_base <= u
    const ZBlock &b = blk[seq_list[t]];
    const u32 s_first = (u32)(u - unit_base[t]) * (U * 64u), nseq = b.nseq;
    const u64 sbase = b.seq_base + s_first;
    u8 *out = dst + b.out_off;
    // (where a match lands, its length and offset wait in LDS -- a wavefront's own columns, no barrier -- until the match runs: in registers
    // beside the dependency ranges they made the kernel 268 VGPRs with the long-copy paths, one wavefront per SIMD instead of two)
    __shared__ u32 s_d[U * 64], s_ml[U * 64], s_of[U * 64];
    u32 dlo[U], dn[U];
    u64 pend[U];
    u32 left = 0;
#pragma unroll
    for (u32 w = 0; w < U; w++) {
        const u32 s = s_first + w * 64 + lane;
        const u64 i = sbase + w * 64 + lane;
        const u32 m = s < nseq ? A.ml[i] : 0;                                 // (with the period of a run's continuation in its high bits: k_lz_runs_apply)
        dlo[w] = 0; dn[w] = 0;
        s_ml[w * 64 + lane] = m;
        if (m) { dlo[w] = A.dep_lo[i]; dn[w] = A.dep_n[i]; s_d[w * 64 + lane] = A.x_dst[i]; s_of[w * 64 + lane] = A.of[i]; }
        pend[w] = __ballot(m != 0);
        left += (u32)__popcll(pend[w]);
    }
    u32 idle = 0, seen = 0, unsaid = 0; u64 since = 0;
    while (left) {
        u64 rdy[U]; u64 any = 0;
        // the first `done` byte of everything pending, all loads in flight together (a word at a time they were U round trips in a row: the
        // time of a sweep, and a sweep is what a link of a chain costs); a match with several sources looks at the others once its first is done
        u8 f0[U];
#pragma unroll
        for (u32 w = 0; w < U; w++) f0[w] = (((pend[w] >> lane) & 1) && (dn[w] & ~LZ_DEP_TAIL)) ? ld_sc1<u8>(((dn[w] & LZ_DEP_TAIL) ? A.stail : A.sdone) + dlo[w]) : (u8)1;
#pragma unroll
        for (u32 w = 0; w < U; w++) {
            bool ready = ((pend[w] >> lane) & 1) && f0[w];
            if (ready && (dn[w] & ~LZ_DEP_TAIL) > 1) ready = lz_deps_done(A.sdone, dlo[w] + 1, dn[w] - 1);
            rdy[w] = pend[w] ? __ballot(ready) : 0;
            any |= rdy[w];
        }
        if (any) {
            asm volatile("" ::: "memory");                                   // (the sources are read after their `done` bytes, not before)
#pragma unroll
            for (u32 w = 0; w < U; w++) {
                const u64 r = rdy[w];
                if (!r) continue;
                const bool ready = (r >> lane) & 1;
                const u32 d = s_d[w * 64 + lane], mlp = s_ml[w * 64 + lane], ml = mlp & ((1u << LZ_PER_SHIFT) - 1), per = mlp >> LZ_PER_SHIFT, of = s_of[w * 64 + lane];
                const bool plain = ready && !per && of >= ml;
                if (plain && ml <= EXEC_LANE_MAX) lane_copy_sc1(out + d, out + d - of, ml);
                for (u64 big = __ballot(plain && ml > EXEC_LANE_MAX); big; big &= big - 1) {
                    const int j = __ffsll((long long)big) - 1;
                    const u32 dj = (u32)__builtin_amdgcn_readlane((int)d, j), mlj = (u32)__builtin_amdgcn_readlane((int)ml, j), ofj = (u32)__builtin_amdgcn_readlane((int)of, j);
                    wave_copy_sc1(out + dj, out + dj - ofj, mlj, lane);
                }
                // a match that overlaps itself repeats its first `of` bytes: only bytes in front of the match are read
                for (u64 ov = __ballot(ready && (per || of < ml)); ov; ov &= ov - 1) {
                    const int j = __ffsll((long long)ov) - 1;
                    u32 dj = (u32)__builtin_amdgcn_readlane((int)d, j), mlj = (u32)__builtin_amdgcn_readlane((int)ml, j), ofj = (u32)__builtin_amdgcn_readlane((int)of, j);
                    const u32 pj = (u32)__builtin_amdgcn_readlane((int)per, j);
                    if (pj) {
                        // a run's continuation: its first period from the seed (ofj bytes in front of the match), in phase; the rest repeats it
                        const u8 *seed = out + dj - ofj;
                        const u32 first = pj < mlj ? pj : mlj, ph = ofj % pj;
                        for (u32 k = lane; k < first; k += 64) st_sc1<u8>(out + dj + k, ld_sc1<u8>(seed + (ph + k) % pj));
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        dj += first; mlj -= first; ofj = pj;
                        if (!mlj) { if (lane == 0) st_sc1<u8>(A.stail + sbase + w * 64 + (u32)j, (u8)1); continue; }
                    }
                    const u8 *from = out + dj - ofj;
                    if (mlj <= 2 * LZ_TAIL) { for (u32 k = lane; k < mlj; k += 64) st_sc1<u8>(out + dj + k, ld_sc1<u8>(from + k % ofj)); continue; }
                    // a long run: its last LZ_TAIL bytes first (the run of the next block starts from them: k_lz_deps, LZ_DEP_TAIL) ...
                    for (u32 k = mlj - LZ_TAIL + lane; k < mlj; k += 64) st_sc1<u8>(out + dj + k, ld_sc1<u8>(from + k % ofj));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) st_sc1<u8>(A.stail + sbase + w * 64 + (u32)j, (u8)1);
                    // ... then the rest by doubling: what is there -- the unit in front of the match and `have` bytes of the match, a whole
                    // number of units -- is copied behind itself (eight bytes a lane; a byte a lane took 0.55 ms per 128 KiB block)
                    const u32 body = mlj - LZ_TAIL;
                    u32 have = 0;
                    if (ofj < 64) { const u32 first = (64 / ofj) * ofj < body ? (64 / ofj) * ofj : body; for (u32 k = lane; k < first; k += 64) st_sc1<u8>(out + dj + k, ld_sc1<u8>(from + k % ofj)); have = first; }
                    while (have < body) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        const u32 n = ofj + have < body - have ? ((ofj + have) / ofj) * ofj : body - have;
                        wave_copy_sc1(out + dj + have, from, n, lane);
                        have += n;
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the bytes are out before anybody is told
            u32 ran = 0;
#pragma unroll
            for (u32 w = 0; w < U; w++) {
                if ((rdy[w] >> lane) & 1) { st_sc1<u8>(A.sdone + sbase + w * 64 + lane, (u8)1); if ((s_ml[w * 64 + lane] >> LZ_PER_SHIFT) || s_of[w * 64 + lane] < s_ml[w * 64 + lane]) st_sc1<u8>(A.stail + sbase + w * 64 + lane, (u8)1); }   // (`stail` of a short overlapping match: set with the match)
                pend[w] &= ~rdy[w]; ran += (u32)__popcll(rdy[w]);
            }
            left -= ran;
            // (the frame's progress counter -- what a unit that cannot move looks at before it gives up -- is told in lumps: one atomic per
            // sweep from every unit of the device was a queue on one address, a third of the launch's time on a stream of short copies)
            unsaid += ran;
            if (unsaid >= 512 || !left) { if (lane == 0) __hip_atomic_fetch_add(&st->n_exec_done, unsaid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); unsaid = 0; }
            idle = 0;
            continue;
        }
        __builtin_amdgcn_s_sleep(2);
        if (unsaid && idle == 1024) { if (lane == 0) __hip_atomic_fetch_add(&st->n_exec_done, unsaid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); unsaid = 0; }
        if (++idle >= (1u << 18)) {                                           // nothing of this unit could run for a long time: has ANYTHING run meanwhile?
            const u32 now = ld_sc1<u32>(&st->n_exec_done);
            if (now != seen) since = 0;
            if (ld_sc1<u32>(&st->err) || (now == seen && exec_stalled(since))) { if (lane == 0) set_err(st, ZE_CORRUPT); break; }
            seen = now; idle = 0;
        }
    }
}

// (TRACE only) what a frame's matches look like: counts by kind and the first sequences of the first executed block
struct LzStats { u64 n_match, n_overlap, n_dep0, n_dep1, n_dep2, n_dep3p, sum_ml, max_dep; u32 first[32][4]; };
__global__ void k_lz_stats(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, LzArrays A, LzStats *S)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seq_blk) return;
    const ZBlock &b = blk[seq_list[t]];
    const u32 nseq = b.err ? 0 : b.nseq;
    u64 nm = 0, no = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0, sm = 0, mx = 0;
    for (u32 s = 0; s < nseq; s++) {
        const u64 i = b.seq_base + s; const u32 ml = A.ml[i] & ((1u << LZ_PER_SHIFT) - 1); if (!ml) continue;
        nm++; sm += ml; if (A.of[i] < ml || (A.ml[i] >> LZ_PER_SHIFT)) no++;
        const u32 n = A.dep_n[i] & ~LZ_DEP_TAIL; if (n == 0) d0++; else if (n == 1) d1++; else if (n == 2) d2++; else d3++; if (n > mx) mx = n;
        if (t < 2 && s < 16) { S->first[16 * t + s][0] = A.x_dst[i]; S->first[16 * t + s][1] = ml; S->first[16 * t + s][2] = A.of[i]; S->first[16 * t + s][3] = A.dep_n[i] ? (u32)((i - A.dep_lo[i]) & 0xFFFFF) | ((n > 255 ? 255 : n) << 20) | (A.dep_n[i] & LZ_DEP_TAIL) : 0xFFFFFFFFu; }
    }
    atomicAdd((unsigned long long *)&S->n_match, nm); atomicAdd((unsigned long long *)&S->n_overlap, no); atomicAdd((unsigned long long *)&S->n_dep0, d0); atomicAdd((unsigned long long *)&S->n_dep1, d1);
    atomicAdd((unsigned long long *)&S->n_dep2, d2); atomicAdd((unsigned long long *)&S->n_dep3p, d3); atomicAdd((unsigned long long *)&S->sum_ml, sm); atomicMax((unsigned long long *)&S->max_dep, mx);
}

// The sequences of blocks seq_list[0 .. nx) executed: as dataflow (above), or in block order (EXEC=batch / =serial: the cross-checks; frames
// of more than 2^32 sequences).  `done` / `prog`: per-block arrays of the block-ordered executors.
static int launch_lz_exec(naf_gpu_ctx *c, const ZBlock *blk, const u32 *seq_list, u32 nx, const u64 *offs, const u64 *seq_cnt, u32 nblk, u64 ns_total,
                          u32 *o_ll, u32 *o_ml, u32 *o_of, const u8 *lits, u8 *d_dst, u32 *done, ZStat *st)
{
    const char *how = ctx_opt(c, "EXEC");
    if (how && how[0] == 's') {
        LAUNCH(c, "zstd_exec_seq", k_exec_seq, nx, 64, 0, blk, seq_list, nx, offs, nblk, (const u32 *)o_ll, (const u32 *)o_ml, (const u32 *)o_of, lits, d_dst, done, st);
        return 0;
    }
    if ((how && how[0] == 'b') || ns_total >= 0xFFFFFFF0ull) {
        u32 *prog = arena_new<u32>(c, nblk); if (!prog) return NAF_GPU_ENOMEM;
        HIP_TRY(c, hipMemsetAsync(prog, 0, 4 * (size_t)nblk, c->stream));
        LAUNCH(c, "zstd_exec_seq", k_exec_batch, nx, 64, 0, blk, seq_list, nx, offs, nblk, (const u32 *)o_ll, (const u32 *)o_ml, (const u32 *)o_of, lits, d_dst, done, prog, st);
        return 0;
    }
    LzArrays A; A.x_dst = o_ll; A.ml = o_ml; A.of = o_of;
    A.dep_lo = arena_new<u32>(c, ns_total + 1); A.dep_n = arena_new<u32>(c, ns_total + 1); A.sdone = (u8 *)arena_alloc(c, 2 * (ns_total + 16));
    if (!A.dep_lo || !A.dep_n || !A.sdone) return NAF_GPU_ENOMEM;
    A.stail = A.sdone + ns_total + 16;
    HIP_TRY(c, hipMemsetAsync(A.sdone, 1, 2 * (ns_total + 16), c->stream));
    LAUNCH(c, "zstd_lz_prep", k_lz_prep, nx, 64, 0, blk, seq_list, nx, seq_cnt, nblk, ns_total, A, lits, d_dst, st);
    // units of 1024 sequences in frame order (k_lz_exec's), of 8192 (k_lz_collapse's)
    u64 *units = arena_new<u64>(c, 2 * ((size_t)nx + 2)); if (!units) return NAF_GPU_ENOMEM;
    u64 *cunits = units + nx + 2;
    LAUNCH(c, "zstd_lz_units", k_lz_units, cdiv(nx, 64), 64, 0, blk, seq_list, nx, LZ_UNIT, units);
    int rc = scan_exclusive_u64(c, units, nx, units + nx + 1); if (rc) return rc;
    const u32 grid = (u32)(ns_total / LZ_UNIT + nx + 1);                     // (an upper bound known without a read-back: wavefronts behind the last unit leave at once)
    // (a frame of a few thousand sequences has no chain worth four launches: the few matches of a random genome's frame)
    if (!ctx_opt_is(c, "EXEC_COLLAPSE", '0') && (ns_total >= 16384 || ctx_opt_is(c, "EXEC_COLLAPSE", 's'))) {
        // the shape by the sequences a block holds on average: a frame that is a chain of copies is matches and little else
        const bool big = ns_total / nx >= 4096 && !ctx_opt_is(c, "EXEC_COLLAPSE", 's');
        const u32 cunit = big ? LZ_CUNIT_BIG : LZ_CUNIT_SMALL, cwg = big ? LZ_CWG_BIG : LZ_CWG_SMALL, clds = 3 * cunit * 4;
        LAUNCH(c, "zstd_lz_units", k_lz_units, cdiv(nx, 64), 64, 0, blk, seq_list, nx, cunit, cunits);
        if ((rc = scan_exclusive_u64(c, cunits, nx, cunits + nx + 1))) return rc;
        const u32 cgrid = (u32)(ns_total / cunit + nx + 1);
        u32 *hops = arena_new<u32>(c, LZ_FAR_ROUNDS + 2); if (!hops) return NAF_GPU_ENOMEM;
        HIP_TRY(c, hipMemsetAsync(hops, 0, (LZ_FAR_ROUNDS + 2) * 4, c->stream));
        if (big) {
            HIP_TRY(c, hipFuncSetAttribute((const void *)k_lz_collapse<LZ_CUNIT_BIG, LZ_CWG_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)clds));
            LAUNCH(c, "zstd_lz_collapse", (k_lz_collapse<LZ_CUNIT_BIG, LZ_CWG_BIG>), cgrid, cwg, clds, blk, seq_list, nx, (const u64 *)cunits, (const u64 *)(cunits + nx + 1), A, hops);
        } else LAUNCH(c, "zstd_lz_collapse", (k_lz_collapse<LZ_CUNIT_SMALL, LZ_CWG_SMALL>), cgrid, cwg, clds, blk, seq_list, nx, (const u64 *)cunits, (const u64 *)(cunits + nx + 1), A, hops);
        if (!ctx_opt_is(c, "EXEC_COLLAPSE", 'n')) {                       // ('n': within units only)
            if (big) {
                HIP_TRY(c, hipFuncSetAttribute((const void *)k_lz_collapse_far<LZ_CUNIT_BIG, LZ_CWG_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)clds));
                LAUNCH(c, "zstd_lz_collapse_far", (k_lz_collapse_far<LZ_CUNIT_BIG, LZ_CWG_BIG>), cgrid, cwg, clds, blk, seq_list, nx, (const u64 *)cunits, (const u64 *)(cunits + nx + 1), A, hops, ns_total);
            } else LAUNCH(c, "zstd_lz_collapse_far", (k_lz_collapse_far<LZ_CUNIT_SMALL, LZ_CWG_SMALL>), cgrid, cwg, clds, blk, seq_list, nx, (const u64 *)cunits, (const u64 *)(cunits + nx + 1), A, hops, ns_total);
            if (ctx_tracing(c)) {
                u32 hh[LZ_FAR_ROUNDS + 2]; if ((rc = ctx_readback(c, hh, hops, sizeof hh))) return rc;
                ctx_trace(c, "[lz] matches that moved their source within their unit: %u of %llu sequences; units by the rounds they took across units (1, 2, ...):", hh[0], (unsigned long long)ns_total);
                for (u32 r = 1; r <= LZ_FAR_ROUNDS; r++) ctx_trace(c, " %u", hh[r]);
                ctx_trace(c, "\n");
            }
        }
    }
    LAUNCH(c, "zstd_lz_deps", k_lz_deps, (dim3(nx, nx >= 4096 ? 1u : (4096u / nx < LZ_DEPS_PARTS ? 4096u / nx : LZ_DEPS_PARTS))), 64, 0, blk, seq_list, nx, offs, seq_cnt, nblk, ns_total, A);
    // runs that continue from block to block read their seed (NAF_GPU_EXEC_RUNS=0: every block from the block in front, the cross-check)
    if (nx >= 2 && !ctx_opt_is(c, "EXEC_RUNS", '0')) {
        i32 *through = arena_new<i32>(c, (size_t)nx + 1), *bad = arena_new<i32>(c, (size_t)nx + 1); u32 *lnk = arena_new<u32>(c, (size_t)nx + 1);
        if (!through || !lnk || !bad) return NAF_GPU_ENOMEM;
        LAUNCH(c, "zstd_lz_runs", k_lz_runs_mark, cdiv(nx, 256), 256, 0, blk, seq_list, nx, A, through, lnk);
        if ((rc = scan_inclusive_max_i32(c, through, nx))) return rc;
        LAUNCH(c, "zstd_lz_runs", k_lz_runs_check, cdiv(nx, 256), 256, 0, blk, seq_list, nx, A, (const i32 *)through, (const u32 *)lnk, (const u8 *)d_dst, bad);
        if ((rc = scan_inclusive_max_i32(c, bad, nx))) return rc;
        LAUNCH(c, "zstd_lz_runs", k_lz_runs_apply, cdiv(nx, 256), 256, 0, blk, seq_list, nx, A, (const i32 *)through, (const u32 *)lnk, (const i32 *)bad);
    }
    if (ctx_tracing(c)) {
        LzStats *S = arena_new<LzStats>(c, 1), hs; if (!S) return NAF_GPU_ENOMEM;
        HIP_TRY(c, hipMemsetAsync(S, 0, sizeof(LzStats), c->stream));
        LAUNCH(c, "zstd_lz_stats", k_lz_stats, cdiv(nx, 64), 64, 0, blk, seq_list, nx, A, S);
        if ((rc = ctx_readback(c, &hs, S, sizeof hs))) return rc;
        ctx_trace(c, "[lz] blocks %u sequences %llu matches %llu (overlapping %llu) bytes %llu; sources written by 0 / 1 / 2 / more matches: %llu / %llu / %llu / %llu (most: %llu)\n", nx, (unsigned long long)ns_total,
                  (unsigned long long)hs.n_match, (unsigned long long)hs.n_overlap, (unsigned long long)hs.sum_ml, (unsigned long long)hs.n_dep0, (unsigned long long)hs.n_dep1, (unsigned long long)hs.n_dep2, (unsigned long long)hs.n_dep3p, (unsigned long long)hs.max_dep);
        for (int k = 0; k < 32; k++) if (hs.first[k][1]) {
            const u32 v = hs.first[k][3];
            if (v == 0xFFFFFFFFu) ctx_trace(c, "[lz]   block %d seq %2d: lands at %6u length %6u offset %7u, source final\n", k / 16, k % 16, hs.first[k][0], hs.first[k][1], hs.first[k][2]);
            else ctx_trace(c, "[lz]   block %d seq %2d: lands at %6u length %6u offset %7u, source written by %u match(es) from %u sequences back%s\n", k / 16, k % 16, hs.first[k][0], hs.first[k][1], hs.first[k][2], (v >> 20) & 255, v & 0xFFFFF, (v & LZ_DEP_TAIL) ? " (their tail)" : "");
        }
    }
    LAUNCH(c, "zstd_exec_seq", k_lz_exec<LZ_UNIT / 64>, grid, 64, 0, blk, seq_list, nx, (const u64 *)units, (const u64 *)(units + nx + 1), A, d_dst, st);
    return 0;
}

// Blocks whose regenerated bytes intersect [want_lo, want_hi): first block index, one-past-last, and their byte span.
__global__ void k_find_range(const u64 *offs, u32 nblk, const u64 *total_p, u64 want_lo, u64 want_hi, const i32 *own_huf, u64 *out4)
{
    if (threadIdx.x || blockIdx.x) return;
    u32 lo = 0, hi = nblk;                                   // last block with offs <= want_lo
    while (lo + 1 < hi) { u32 mid = (lo + hi) >> 1; if (offs[mid] <= want_lo) lo = mid; else hi = mid; }
    u32 b_lo = lo;
    lo = b_lo; hi = nblk;                                    // first block with offs >= want_hi
    while (lo < hi) { u32 mid = (lo + hi) >> 1; if (offs[mid] < want_hi) lo = mid + 1; else hi = mid; }
    u32 b_hi = lo;
    out4[0] = b_lo; out4[1] = b_hi; out4[2] = offs[b_lo]; out4[3] = b_hi < nblk ? offs[b_hi] : *total_p;
    i32 ob = own_huf[b_lo]; out4[4] = ob < 0 ? b_lo : (u64)ob;      // first block whose Huffman table can be in force in the range
}

// ---- byte-range decode of a frame WITH sequences: the dependency closure of the range (unnaf/src/input.c:271 lets a match reach back
// 2^31 bytes; what a block range needs in front of it is not a window but everything its matches read, and what THOSE blocks' matches
// read, and so on) -------------------------------------------------------------------------------------------------------------------
// f[b] = the block that holds the earliest byte a match of block b reads (b itself when nothing reaches in front of it)
__global__ void k_seq_reach(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, const u64 *offs, const u32 *o_ll, const u32 *o_ml, const u32 *o_of, u32 *f)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seq_blk) return;
    const u32 bi = seq_list[t];
    const ZBlock &b = blk[bi];
    if (b.err) return;
    const u64 start = offs[bi];
    u64 pos = start, m = start;
    const u32 rep_in[3] = { b.rep_in[0], b.rep_in[1], b.rep_in[2] };
    for (u32 q = 0; q < b.nseq; q++) {
        const u32 ll = o_ll[b.seq_base + q], ml = o_ml[b.seq_base + q], off = sym_resolve(o_of[b.seq_base + q], rep_in);
        pos += ll;
        const u64 src = off > pos ? 0 : pos - off;
        if (src < m) m = src;
        pos += ml;
    }
    if (m >= start) return;                                    // f[bi] stays bi
    u32 lo = 0, hi = bi;                                       // largest j with offs[j] <= m
    while (lo + 1 < hi) { const u32 mid = (lo + hi) >> 1; if (offs[mid] <= m) lo = mid; else hi = mid; }
    f[bi] = lo;
}
__global__ void k_iota_u32(u32 *f, u32 n) { const u32 i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) f[i] = i; }
// The blocks a block's matches copy from, marked "to be decoded" (2) in the class table of a mostly-flat frame (ZFlat.cls): a lane per
// block with sequences walks them; a source block that has sequences itself is in the table already and marks its own sources.
__global__ void k_seq_sources(const ZBlock *blk, const u32 *seq_list, u32 n_seq_blk, const u64 *offs, const u32 *o_ll, const u32 *o_ml, const u32 *o_of, u8 *cls0)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seq_blk) return;
    const u32 bi = seq_list[t];
    const ZBlock &b = blk[bi];
    if (b.err) return;
    const u64 start = offs[bi];
    u64 pos = start;
    const u32 rep_in[3] = { b.rep_in[0], b.rep_in[1], b.rep_in[2] };
    for (u32 q = 0; q < b.nseq; q++) {
        const u32 ll = o_ll[b.seq_base + q], ml = o_ml[b.seq_base + q], off = sym_resolve(o_of[b.seq_base + q], rep_in);
        pos += ll;
        if (off && off <= pos && ml) {                          // (an offset beyond the start of the output is the executor's error to report)
            const u64 src = pos - off, end = src + ml < start ? src + ml : start;
            if (src < start) {
                u32 lo = 0, hi = bi;                                // largest j with offs[j] <= src
                while (lo + 1 < hi) { const u32 mid = (lo + hi) >> 1; if (offs[mid] <= src) lo = mid; else hi = mid; }
                for (u32 j = lo; j < bi && offs[j] < end; j++) cls0[j] = 2;
            }
        }
        pos += ml;
    }
}
// One wavefront: blocks [b_lo, b_hi) hold the wanted bytes (k_find_range); the closure's first block c is the least fixed point of
// c = min(c, f[b] for b in [c, b_hi)).  out: [0] c, [1] b_hi, [2] offs[c], [3] offs[b_hi] (or the total), [4] block whose Huffman
// table is in force at c, [5] / [6] ranks of c / b_hi among the blocks with sequences.
__global__ __launch_bounds__(64) void k_range_closure(const u32 *f, const u64 *offs, u32 nblk, const u64 *total_p, const u64 *r4, const i32 *own_huf, const u64 *seq_rank, u32 n_seq_blk, u64 *out)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    const u32 lane = threadIdx.x;
    const u32 b_lo = (u32)r4[0], b_hi = (u32)r4[1];
    u32 c = b_lo, lo = b_lo, hi = b_hi;
    for (;;) {
        u32 m = c;
        for (u32 b = lo + lane; b < hi; b += 64) { const u32 v = f[b]; m = v < m ? v : m; }
        for (int d = 32; d; d >>= 1) { const u32 o = (u32)__shfl_xor((int)m, d, 64); m = o < m ? o : m; }
        if (m >= c) break;
        hi = c; lo = m; c = m;                                    // the blocks newly taken in may reach further back
    }
    if (lane == 0) {
        out[0] = c; out[1] = b_hi; out[2] = offs[c]; out[3] = b_hi < nblk ? offs[b_hi] : *total_p;
        const i32 ob = own_huf[c]; out[4] = ob < 0 ? c : (u64)ob;
        out[5] = seq_rank[c]; out[6] = b_hi < nblk ? seq_rank[b_hi] : n_seq_blk;
    }
}

// ---- a small frame in one launch ------------------------------------------------------------------------------------------------
// The pipeline above costs a frame about 25 launches and half a dozen host read-backs whatever its size -- half a millisecond for
// the ids, the names and the lengths of an archive of a hundred chromosomes, a few hundred bytes each, and the emit kernels wait
// for all three.  A frame of up to SMALL_SRC bytes is decoded here by ONE lane, block after block, the way a serial decoder does
// it (same per-block functions as the kernels above: zstd_dec_core.h), with the Huffman and FSE tables in LDS; one launch, one
// read-back.  res: [0] error (ZE_*; SMALL_TOO_BIG when the output does not fit `cap` -- the caller then takes the long way, which
// reports the size needed), [1] bytes produced, [2] bytes of the frame.
#define SMALL_SRC 16384u
#define SMALL_OUT 32768u
#define SMALL_SEQ 2048u
#define SMALL_TOO_BIG 100u
struct SmallRes { u32 err, out, pos; };
// The frame's bytes, its output, a block's literals and sequences all sit in LDS (a dependent access costs an LDS round trip, not an
// HBM one).  EVERY lane of the one wavefront runs this routine: what is serial by nature (headers, FSE-coded weights, the sequences'
// state machines) all 64 lanes compute alike -- same reads, same values written to the same LDS words, the time of one lane -- and
// what is not is spread over them: copies and fills, the Huffman table (build_huf_one_lds), the four streams of a literals section
// on four lanes, the execution of the sequences (literal runs beside each other, then the matches one after the other, each by the
// whole wavefront).  One lane doing all of it took 120 - 180 us for the few hundred bytes of ids, names and lengths of an archive of
// a hundred records -- which the emit of 10 GB waits for (profiles/r04_timeline_uniform_10GB.txt).
struct SmallWS { HufLdsWS H; ZBlock b; ZStat st; };
__device__ void small_frame_decode(const u8 *src, u32 len, u8 *dst, u32 cap, u8 *lit, u32 *sll, u32 *sml, u32 *sof, const FseE *predef,
                                   SmallWS &W, u16 *huf, FseE *fse, i16 *norm, u16 *nx, SmallRes *res, const u32 *llt, const u32 *mlt)
{
    const u32 lane = threadIdx.x;
    u32 err = 0, out = 0, pos = 0;
    ZFrameHdr fh = zstd_parse_frame_header(src, len);
    if (fh.err) { res->err = (u32)fh.err; res->out = 0; res->pos = 0; return; }
    pos = fh.hdr_size;
    u32 huf_log = 0; bool have_huf = false;
    SeqTab tab[3]; bool have_tab[3] = { false, false, false };
    const u32 fo[3] = { 0, 512, 768 }, po[3] = { 0, 64, 96 }, pl[3] = { 6, 5, 6 }, max_log[3] = { 9, 8, 9 }, max_sym[3] = { 35, 31, 52 };
    u32 rep[3] = { 1, 4, 8 };
    for (;;) {
        if (pos + 3 > len) { err = ZE_TRUNC; break; }
        const u32 h = ld24(src + pos), last = h & 1, type = (h >> 1) & 3, size = h >> 3;
        if (type == 3 || size > ZBLOCK_MAX) { err = ZE_CORRUPT; break; }
        const u32 csize = type == BT_RLE ? 1 : size;
        if (pos + 3 + csize > len) { err = ZE_TRUNC; break; }
        const u8 *c = src + pos + 3;
        pos += 3 + csize;
        if (type != BT_COMP) {
            if ((u64)out + size > cap) { err = SMALL_TOO_BIG; break; }
            if (type == BT_RAW) for (u32 k = lane; k < size; k += 64) dst[out + k] = c[k];
            else { const u8 v = c[0]; for (u32 k = lane; k < size; k += 64) dst[out + k] = v; }
            __syncthreads();
            out += size;
            if (last) break;
            continue;
        }
        ZBlock b; b.src_off = 0; b.bsize = size; b.btype = BT_COMP; b.last = (u8)last;
        zstd_parse_block(c, b);
        if (b.err) { err = b.err; break; }
        if ((u64)out + b.lit_regen > cap) { err = SMALL_TOO_BIG; break; }             // (lit is as large as the output buffer)
        // literals: straight into the output when the block has no sequences
        u8 *lp = b.nseq ? lit : dst + out;
        if (b.lit_type == LIT_RAW) for (u32 k = lane; k < b.lit_regen; k += 64) lp[k] = c[b.lit_off + k];
        else if (b.lit_type == LIT_RLE) { const u8 v = c[b.lit_off]; for (u32 k = lane; k < b.lit_regen; k += 64) lp[k] = v; }
        else {
            if (b.lit_type == LIT_HUF) {
                __syncthreads();
                if (lane == 0) { W.b = b; W.b.src_off = 0; }
                __syncthreads();
                build_huf_one_lds(c, &W.b, 0, nullptr, 0, &W.st, W.H, (u8 *)huf);
                __syncthreads();
                const u32 lg = W.H.log;
                if (!lg) { err = ZE_CORRUPT; break; }
                huf_log = lg; have_huf = true;
            } else if (!have_huf) { err = ZE_CORRUPT; break; }              // treeless without a previous table
            const u8 *sp = c + b.huf_streams_off;
            bool bad = false;
            if (b.nstreams == 1) { if (lane == 0) bad = huf_decode_stream(sp, b.huf_streams_size, (const u16 *)huf, huf_log, lp, b.lit_regen) != 0; }
            else {
                const u32 s1 = ld16(sp), s2 = ld16(sp + 2), s3 = ld16(sp + 4), tot = b.huf_streams_size - 6, per = (b.lit_regen + 3) / 4;
                if (s1 + s2 + s3 >= tot || !s1 || !s2 || !s3 || per * 3 > b.lit_regen) { err = ZE_CORRUPT; break; }
                if (lane < 4) {                                             // a lane per stream
                    const u32 k = lane;
                    const u32 o = k == 0 ? 0 : (k == 1 ? s1 : (k == 2 ? s1 + s2 : s1 + s2 + s3)), z = k == 0 ? s1 : (k == 1 ? s2 : (k == 2 ? s3 : tot - s1 - s2 - s3));
                    bad = huf_decode_stream(sp + 6 + o, z, (const u16 *)huf, huf_log, lp + k * per, k < 3 ? per : b.lit_regen - 3 * per) != 0;
                }
            }
            if (__ballot(bad)) { err = ZE_CORRUPT; break; }
        }
        __syncthreads();
        if (!b.nseq) { out += b.lit_regen; if (last) break; continue; }
        if (b.nseq > SMALL_SEQ) { err = SMALL_TOO_BIG; break; }
        // sequence tables (3.1.1.3.2.1): predefined, RLE, FSE description, or the table of the previous block
        u32 p = b.seq_off; bool tbad = false;
        for (int k = 0; k < 3 && !tbad; k++) {
            const u32 m = b.modes[k];
            if (m == SM_PREDEF) { tab[k].t = predef + po[k]; tab[k].log = pl[k]; tab[k].rle = false; tab[k].rle_sym = 0; have_tab[k] = true; }
            else if (m == SM_RLE) { tab[k].t = predef; tab[k].log = 0; tab[k].rle = true; tab[k].rle_sym = b.fse_tab[k]; have_tab[k] = true; p++; }
            else if (m == SM_FSE) {
                u32 nsym, lg;
                const u32 d = fse_read_ncount(c + p, b.bsize - p, max_log[k], max_sym[k], norm, &nsym, &lg);
                if (!d || lg != b.fse_log[k] || !fse_build_table(fse + fo[k], norm, nsym, lg, nx)) { tbad = true; break; }
                p += d; tab[k].t = fse + fo[k]; tab[k].log = lg; tab[k].rle = false; tab[k].rle_sym = 0; have_tab[k] = true;
            } else if (!have_tab[k]) tbad = true;
        }
        if (tbad || p != b.seq_bits_off) { err = ZE_CORRUPT; break; }
        u64 tl = 0, tm = 0; u32 ro[3];
        // (blocks of predefined tables: the routine k_decode_seq uses for them -- a lane alone on the general one took 2 - 3 us per sequence)
        u8 e = 0xFF;
        if (b.modes[0] == SM_PREDEF && b.modes[1] == SM_PREDEF && b.modes[2] == SM_PREDEF)
            e = zstd_decode_sequences_predef<const FseE *, const u32 *, false>(c + b.seq_bits_off, b.seq_bits_size, b.nseq, predef, predef + 64, predef + 96, llt, mlt, sll, sml, sof, ro, &tl, &tm, nullptr);
        if (e == 0xFF) e = zstd_decode_sequences(c + b.seq_bits_off, b.seq_bits_size, b.nseq, tab, sll, sml, sof, ro, &tl, &tm);
        if (e) { err = e; break; }
        if (tl > b.lit_regen) { err = ZE_CORRUPT; break; }
        const u64 regen = b.lit_regen + tm;
        if (regen > ZBLOCK_MAX) { err = ZE_CORRUPT; break; }
        if ((u64)out + regen > cap) { err = SMALL_TOO_BIG; break; }
        __syncthreads();
        // execution, 64 sequences at a time: where each one's literals come from and go to by a scan of the lengths; the literal runs
        // of all of them at once (a lane each); then the matches in order, every one by the whole wavefront (a match that overlaps
        // itself repeats its first `of` bytes)
        u32 op = out, l = 0; bool xbad = false;
        for (u32 q0 = 0; q0 < b.nseq && !xbad; q0 += 64) {
            const u32 q = q0 + lane; const bool on = q < b.nseq;
            const u32 ll = on ? sll[q] : 0, ml = on ? sml[q] : 0;
            const u32 ill = wave_scan_inclusive<u32, OpAdd>(ll), iml = wave_scan_inclusive<u32, OpAdd>(ml);
            const u32 my_l = l + ill - ll, my_op = op + (ill - ll) + (iml - ml);
            { u32 k = 0; for (; k + 8 <= ll; k += 8) st64(dst + my_op + k, ld64(lit + my_l + k)); for (; k < ll; k++) dst[my_op + k] = lit[my_l + k]; }
            __syncthreads();
            const u32 cnt = b.nseq - q0 < 64 ? b.nseq - q0 : 64;
            for (u32 j = 0; j < cnt; j++) {
                const u32 mop = (u32)__shfl((int)(my_op + ll), (int)j, 64), mlj = (u32)__shfl((int)ml, (int)j, 64), of = sym_resolve(sof[q0 + j], rep);
                if (of == 0 || of > mop) { xbad = true; break; }
                if (of >= mlj || of >= 64) for (u32 k0 = 0; k0 < mlj; k0 += 64) { const u32 k = k0 + lane; if (k < mlj) { const u8 v = dst[mop + k - of]; dst[mop + k] = v; } __syncthreads(); }
                else { for (u32 k = lane; k < mlj; k += 64) dst[mop + k] = dst[mop - of + k % of]; __syncthreads(); }
            }
            op += (u32)__shfl((int)(ill + iml), 63, 64); l += (u32)__shfl((int)ill, 63, 64);
        }
        if (xbad) { err = ZE_CORRUPT; break; }
        { const u32 rest = b.lit_regen - l; for (u32 k = lane; k < rest; k += 64) dst[op + k] = lit[l + k]; op += rest; }
        __syncthreads();
        { const u32 r0 = sym_resolve(ro[0], rep), r1 = sym_resolve(ro[1], rep), r2 = sym_resolve(ro[2], rep); rep[0] = r0; rep[1] = r1; rep[2] = r2; }
        out = op;
        if (last) break;
    }
    if (!err && fh.checksum) { if (pos + 4 > len) err = ZE_TRUNC; else pos += 4; }
    if (!err && fh.has_fcs && fh.content_size != out) err = ZE_CORRUPT;
    res->err = err; res->out = out; res->pos = pos;
}
__global__ __launch_bounds__(64) void k_small_frame(const u8 *src, u32 len, u8 *dst, u32 cap, const FseE *predef, u32 *res)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    __shared__ __attribute__((aligned(16))) u8 s_src[SMALL_SRC + 16], s_out[SMALL_OUT + 16], s_lit[SMALL_OUT + 16];
    __shared__ u32 s_ll[SMALL_SEQ], s_ml[SMALL_SEQ], s_of[SMALL_SEQ];
    __shared__ SmallWS W;
    __shared__ __attribute__((aligned(16))) u16 huf[HUF_TAB_MAX / 2];
    __shared__ FseE fse[512 + 256 + 512];
    __shared__ FseE s_predef[160];
    __shared__ i16 norm[64];
    __shared__ u16 nx[64];
    __shared__ SmallRes r;
    __shared__ u32 s_llt[36], s_mlt[53];
    zstd_seq_code_tables(s_llt, s_mlt, threadIdx.x, 64);
    for (u32 k = threadIdx.x; k < len; k += 64) s_src[k] = src[k];
    for (u32 k = threadIdx.x; k < 16; k += 64) s_src[len + k] = 0;
    for (u32 k = threadIdx.x; k < 160; k += 64) s_predef[k] = predef[k];
    __syncthreads();
    small_frame_decode(s_src, len, s_out, cap < SMALL_OUT ? cap : SMALL_OUT, s_lit, s_ll, s_ml, s_of, s_predef, W, huf, fse, norm, nx, &r, s_llt, s_mlt);
    __syncthreads();
    if (r.err == 0) for (u32 k = threadIdx.x; k < r.out; k += 64) dst[k] = s_out[k];
    if (threadIdx.x == 0) { res[0] = r.err; res[1] = r.out; res[2] = r.pos; }
}

// Several small frames (the ids, names and lengths of an archive with few records) in one launch, one workgroup each.
struct SmallJobs { const u8 *src[4]; u8 *dst[4]; u32 len[4], cap[4]; };
__global__ __launch_bounds__(64) void k_small_frames(SmallJobs J, const FseE *predef, u32 *res)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    __shared__ __attribute__((aligned(16))) u8 s_src[SMALL_SRC + 16], s_out[SMALL_OUT + 16], s_lit[SMALL_OUT + 16];
    __shared__ u32 s_ll[SMALL_SEQ], s_ml[SMALL_SEQ], s_of[SMALL_SEQ];
    __shared__ SmallWS W;
    __shared__ __attribute__((aligned(16))) u16 huf[HUF_TAB_MAX / 2];
    __shared__ FseE fse[512 + 256 + 512];
    __shared__ FseE s_predef[160];
    __shared__ i16 norm[64];
    __shared__ u16 nx[64];
    __shared__ SmallRes r;
    __shared__ u32 s_llt[36], s_mlt[53];
    zstd_seq_code_tables(s_llt, s_mlt, threadIdx.x, 64);
    const u32 j = blockIdx.x;
    const u8 *src = J.src[j]; const u32 len = J.len[j], cap = J.cap[j]; u8 *dst = J.dst[j];
    for (u32 k = threadIdx.x; k < len; k += 64) s_src[k] = src[k];
    for (u32 k = threadIdx.x; k < 16; k += 64) s_src[len + k] = 0;
    for (u32 k = threadIdx.x; k < 160; k += 64) s_predef[k] = predef[k];
    __syncthreads();
    small_frame_decode(s_src, len, s_out, cap < SMALL_OUT ? cap : SMALL_OUT, s_lit, s_ll, s_ml, s_of, s_predef, W, huf, fse, norm, nx, &r, s_llt, s_mlt);
    __syncthreads();
    if (r.err == 0) for (u32 k = threadIdx.x; k < r.out; k += 64) dst[k] = s_out[k];
    if (threadIdx.x == 0) { res[4 * j] = r.err; res[4 * j + 1] = r.out; res[4 * j + 2] = r.pos; }
}
// n <= 4 frames without their magic number (as stored in .naf sections); ok[k] = frame k decoded, consumed all of its source and
// produced exactly cap[k] bytes.  Frames that are not small, or fail in any way, are left to the caller's ordinary path (which
// also words the error).
int zstd_small_batch(naf_gpu_ctx *c, int n, const u8 *const *src, const size_t *len, u8 *const *dst, const size_t *cap, bool *ok)
{
    SmallJobs J; memset(&J, 0, sizeof J); int m = 0, map[4];
    for (int k = 0; k < n && k < 4; k++) {
        ok[k] = false;
        if (len[k] == 0 || len[k] > SMALL_SRC || cap[k] > SMALL_OUT) continue;
        J.src[m] = src[k]; J.len[m] = (u32)len[k]; J.dst[m] = dst[k]; J.cap[m] = (u32)cap[k]; map[m++] = k;
    }
    if (!m) return 0;
    u32 *d_res = arena_new<u32>(c, 16); if (!d_res) return NAF_GPU_ENOMEM;
    LAUNCH(c, "zstd_small_frame", k_small_frames, m, 64, 0, J, (const FseE *)c->d_predef, d_res);
    u32 res[16]; int rc = ctx_readback(c, res, d_res, 16 * (size_t)m); if (rc) return rc;
    for (int q = 0; q < m; q++) ok[map[q]] = res[4 * q] == 0 && res[4 * q + 1] == J.cap[q] && res[4 * q + 2] == J.len[q];
    return 0;
}
// The same without the read-back, for a caller whose next kernel checks the results itself: frame k is good when
// res[4k] == 0, res[4k + 1] == cap[k] and res[4k + 2] == len[k].  Every frame must be small (zstd_small_fits).
bool zstd_small_fits(size_t len, size_t cap) { return len != 0 && len <= SMALL_SRC && cap <= SMALL_OUT; }
int zstd_small_launch(naf_gpu_ctx *c, int n, const u8 *const *src, const size_t *len, u8 *const *dst, const size_t *cap, u32 **d_res_out)
{
    SmallJobs J; memset(&J, 0, sizeof J);
    if (n < 1 || n > 4) return NAF_GPU_EARG;
    for (int k = 0; k < n; k++) { if (!zstd_small_fits(len[k], cap[k])) return NAF_GPU_EARG; J.src[k] = src[k]; J.len[k] = (u32)len[k]; J.dst[k] = dst[k]; J.cap[k] = (u32)cap[k]; }
    u32 *d_res = arena_new<u32>(c, 16); if (!d_res) return NAF_GPU_ENOMEM;
    LAUNCH(c, "zstd_small_frame", k_small_frames, n, 64, 0, J, (const FseE *)c->d_predef, d_res);
    *d_res_out = d_res;
    return 0;
}

// ---- host orchestration ------------------------------------------------------------------------------------------
int zstd_init_tables(naf_gpu_ctx *c)
{
    static const i16 LL[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
    static const i16 OF[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };
    static const i16 ML[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
    FseE tabs[64 + 32 + 64]; u16 next[64];
    if (!fse_build_table(tabs, LL, 36, 6, next) || !fse_build_table(tabs + 64, OF, 29, 5, next) || !fse_build_table(tabs + 96, ML, 53, 6, next))
        return ctx_fail(c, NAF_GPU_EZSTD, "predefined FSE tables");
    HIP_TRY(c, hipMalloc(&c->d_predef, sizeof tabs));
    HIP_TRY(c, hipMemcpy(c->d_predef, tabs, sizeof tabs, hipMemcpyHostToDevice));
    return 0;
}


// Parts per stream for k_huf_par, a power of two up to 64 (returned as its logarithm; 0 = the one-lane-per-stream kernel).
// Measured (tools/perf_huf.py, profiles/r03_huf_parts.txt): one lane per stream takes 0.95 ms for the 8 K symbols of a stream of this
// build's 32 KiB blocks and 3.6 - 5 ms for the 32 K of libzstd's 128 KiB blocks however few streams there are, and moves 1 GB of
// symbols per millisecond once the device is full; k_huf_par divides the chain by P and moves half of that (every part is walked once to
// be counted, then decoded).  So parts are taken where the chain is what bounds the serial kernel -- less than 84 KiB of literals
// per symbol of the longest stream: 0.7 GB of this build's blocks, 2.8 GB of libzstd's -- with as many parts as leave a lane
// NAF_GPU_HUF_PART symbols (default 128) and the device no more than a quarter of a million lanes.
// NAF_GPU_HUF_PAR=0: never, =N: 2^N parts wherever the streams are long enough.
static u32 huf_par_plog(const naf_gpu_ctx *c, u32 max_lit_regen, u64 n_blocks)
{
    const char *e = ctx_opt(c, "HUF_PAR");
    if (e && e[0] == '0') return 0;
    const char *t = ctx_opt(c, "HUF_PART");
    u32 target = t ? (u32)atoi(t) : 128u; if (target < 64) target = 64;
    const u32 nmax = (max_lit_regen + 3) / 4;                     // symbols of the longest stream (blocks of more than 1 KiB have four)
    u32 plog = 0;
    while (plog < 6 && (nmax >> (plog + 1)) >= target) plog++;
    if (e && e[0] >= '1' && e[0] <= '6') { const u32 f = (u32)(e[0] - '0'); return f < plog ? f : plog; }
    if (n_blocks * (u64)max_lit_regen >= (u64)nmax * 86016) return 0;
    // (the device holds 200 k of this kernel's lanes at a time, but a lane's chain is what a frame of few long streams waits for: the
    // reference's archive of a 4 GB genome -- 61 K streams of 32 K symbols -- 9.2 ms with 4 parts a stream, 6.9 with 16, 6.7 with 32)
    const char *ll = ctx_opt(c, "HUF_LANES_LOG");                 // (the cap as a lever: 18 = round 5's)
    // (... for streams of libzstd's length only: this build's own realistic archive, 8 K symbols a stream beside the flat emit, 3.04 -> 3.35 ms with the larger cap)
    const u32 lanes_log = ll && atoi(ll) >= 10 && atoi(ll) <= 24 ? (u32)atoi(ll) : (nmax >= 16384 ? 20u : 18u);
    while (plog && ((n_blocks * 4) << plog) > (1ull << lanes_log)) plog--;
    return plog;
}
static u32 huf_par_margin_env(const naf_gpu_ctx *c) { const char *m = ctx_opt(c, "HUF_MARGIN"); return m ? (u32)atoi(m) : 0u; }

#define ZSTD_NEED_TWO_PASS (-100)
// tables of many distinct trees: sixteen lanes per tree (NAF_GPU_HUF_BUILD16=0: one lane per tree, the cross-check)
static int launch_build_huf(naf_gpu_ctx *c, u32 count, const u8 *src, ZBlock *blk, u32 nblk, u8 *pool, u32 pool_cap, ZStat *st, u32 first, const u64 *range4, u32 always_table, const i32 *own_huf, u32 gate, u32 phase)
{
    const char *e = ctx_opt(c, "HUF_BUILD16");
    if (e && e[0] == '0') LAUNCH(c, "zstd_build_huf", k_build_huf, cdiv(count, 64), 64, 0, src, blk, nblk, pool, pool_cap, st, first, range4, always_table, own_huf, gate, phase);
    else if (ctx_opt_is(c, "HUF_BUILD_WG", '1')) LAUNCH(c, "zstd_build_huf", k_build_huf16<16>, cdiv(count, 16), 256, 0, src, blk, nblk, pool, pool_cap, st, first, range4, always_table, own_huf, gate, phase);
    else LAUNCH(c, "zstd_build_huf", k_build_huf16<4>, cdiv(count, 4), 64, 0, src, blk, nblk, pool, pool_cap, st, first, range4, always_table, own_huf, gate, phase);
    return 0;
}
static int zerr(naf_gpu_ctx *c, u32 e, const char *where)
{
    const char *m = e == ZE_TRUNC ? "truncated" : e == ZE_CORRUPT ? "corrupt" : e == ZE_UNSUP ? "unsupported feature" : "table pool";
    return ctx_fail(c, NAF_GPU_EZSTD, "zstd frame %s (%s)", m, where);
}

// Decode ONE frame whose header (after the magic) starts at d_src[0].  *consumed = bytes of the frame.
static int zstd_decode_one(naf_gpu_ctx *c, const u8 *d_src, size_t src_len, u8 *d_dst, size_t dst_cap,
                           size_t *out_len, size_t *consumed, ZRange *rg, const EmitP *fuse, u8 *text, const u8 *head = nullptr)
{
    int rc;
    // small frames (side streams of archives with few records): one launch, one read-back
    if (!rg && !fuse && src_len && src_len <= SMALL_SRC && dst_cap <= SMALL_OUT) {
        u32 *d_res = arena_new<u32>(c, 4);
        if (!d_res) return NAF_GPU_ENOMEM;
        LAUNCH(c, "zstd_small_frame", k_small_frame, 1, 64, 0, d_src, (u32)src_len, d_dst, (u32)dst_cap, (const FseE *)c->d_predef, d_res);
        u32 res[3]; if ((rc = ctx_readback(c, res, d_res, 12))) return rc;
        if (res[0] == 0) { *out_len = res[1]; *consumed = res[2]; return 0; }
        if (res[0] != SMALL_TOO_BIG) return zerr(c, res[0], "small frame");
    }
    u8 hb[18]; size_t hl = src_len < 18 ? src_len : 18;
    if (head) memcpy(hb, head, hl);
    else { rc = ctx_readback(c, hb, d_src, hl); if (rc) return rc; }
    ZFrameHdr fh = zstd_parse_frame_header(hb, hl);
    if (fh.err) return zerr(c, (u32)fh.err, "frame header");

    ZStat *st = arena_new<ZStat>(c, 1);
    if (!st) return NAF_GPU_ENOMEM;
    HIP_TRY(c, hipMemsetAsync(st, 0, sizeof(ZStat), c->stream));
    // ---- block index
    ZBlock *blk = nullptr; ZStat hs; bool indexed = false;
    const char *nospec = ctx_opt(c, "SERIAL_INDEX");
    const char *fl_env = ctx_opt(c, "FLAT");                                 // "0": every block through the serial kernel (cross-check)
    const u32 always_table = (fl_env && fl_env[0] == '0') ? 1u : 0u;
    const char *smin = ctx_opt(c, "SPEC_MIN");                      // tests: frames of a few dozen blocks through the paths of the big ones
    const u32 spec_min = smin ? (u32)atoi(smin) : 512u;
    const char *un_env = ctx_opt(c, "UNIFORM");                              // "0": a uniform flat frame takes the general front too
    const bool uni_wanted = c->zflat && !fuse && !always_table && !(un_env && un_env[0] == '0');
    // Frames of more than 4 MiB: 1 MiB chunks, candidates in the first 40 KiB / 128 KiB of each.  Smaller frames can still hold
    // thousands of tiny blocks (ids / names / lengths that compress 100:1 in 16 KiB blocks, a mask stream that is 1200 RLE blocks
    // of 4 bytes -- a serial walk of those costs milliseconds): small chunks with every byte tested as a candidate.  The chunk size changes nothing but how much of
    // the speculation is reused: the resolve pass re-walks from the true start wherever a chunk's candidate was wrong or missing.
    u32 chunk = SPEC_CHUNK, win1 = SPEC_WINDOW1, win2 = SPEC_WINDOW;
    if (src_len <= 4ull * SPEC_CHUNK) {                          // about 128 chunks of 256 B .. 16 KiB, every byte a candidate
        chunk = 256; while (chunk < SPEC_CHUNK_SMALL && (u64)chunk * 128 < src_len) chunk *= 2;
        win1 = win2 = chunk;
    } else if (src_len <= 256ull * SPEC_CHUNK && fh.hdr_size + 3 <= hl) {
        // In between (a soft-masked genome's mask: 11 MB of 5 KB blocks): a chunk is walked by one lane, block after block, and 1 MiB
        // of small blocks is a long walk for eleven lanes.  The frame's first block says what to expect: chunks of about sixteen
        // such blocks (never wrong, only more or less of the speculation reused).
        const u32 h0 = (u32)hb[fh.hdr_size] | ((u32)hb[fh.hdr_size + 1] << 8) | ((u32)hb[fh.hdr_size + 2] << 16);
        const u32 b0 = ((h0 >> 1) & 3) == 1 ? 4u : (h0 >> 3) + 3;     // bytes of the first block (an RLE block stores one byte)
        u32 want = SPEC_CHUNK_SMALL; while (want < SPEC_CHUNK && want < 16 * b0) want *= 2;
        if (want < SPEC_CHUNK) { chunk = want; if (win1 > chunk) win1 = chunk; if (win2 > chunk) win2 = chunk; }
    }
    if (src_len >= 2048 && !(nospec && nospec[0] == '1')) {
        u32 nchunks = (u32)((src_len + chunk - 1) / chunk);
        u64 *first = arena_new<u64>(c, nchunks + 1), *land = arena_new<u64>(c, nchunks + 2), *G = arena_new<u64>(c, nchunks + 1);
        u64 *start = arena_new<u64>(c, nchunks + 2), *cnt = arena_new<u64>(c, (size_t)nchunks + 4);      // (+ the scan's total and the stride index's verdict behind it)
        if (!first || !land || !G || !start || !cnt) return NAF_GPU_ENOMEM;
        HIP_TRY(c, hipMemsetAsync(first, 0xFF, (size_t)nchunks * 8, c->stream));
        u64 *h0 = (u64 *)c->h_stage; *h0 = fh.hdr_size;
        HIP_TRY(c, hipMemcpyAsync(first, h0, 8, hipMemcpyHostToDevice, c->stream));
        u32 gl = cdiv(nchunks, 64);
        // the stride index first (frames of more than 4 MiB whose first block is a compressed one that is not the last): three launches and
        // a read-back; a frame it cannot take costs that read-back before the speculative index starts
        {
            const char *se = ctx_opt(c, "STRIDE_INDEX");
            if (ctx_tracing(c)) ctx_trace(c, "[stride?] len %zu hdr %u hl %zu first %02x %02x %02x\n", src_len, fh.hdr_size, hl, hb[fh.hdr_size], hb[fh.hdr_size + 1], hb[fh.hdr_size + 2]);
            if (src_len > 4ull * SPEC_CHUNK && fh.hdr_size + 3 <= hl && !(se && se[0] == '0')) {
                const u32 h0 = (u32)hb[fh.hdr_size] | ((u32)hb[fh.hdr_size + 1] << 8) | ((u32)hb[fh.hdr_size + 2] << 16);
                const u32 S = 3 + (h0 >> 3);
                const u64 nmax64 = (src_len - fh.hdr_size) / S;
                if (((h0 >> 1) & 3) == BT_COMP && !(h0 & 1) && (h0 >> 3) <= ZBLOCK_MAX && S >= 256 && nmax64 >= 64 && nmax64 < 0x7FFFFF00ull) {
                    const u32 nmax = (u32)nmax64;
                    u32 *sres = (u32 *)(cnt + nchunks + 2); ZBlock *sblk = arena_new<ZBlock>(c, (size_t)nmax + STRIDE_TAIL);
                    if (!sblk) return NAF_GPU_ENOMEM;
                    HIP_TRY(c, hipMemsetAsync(sres, 0xFF, 8, c->stream));
                    LAUNCH(c, "zstd_index_stride", k_stride_probe, cdiv(nmax, 256), 256, 0, d_src, (u64)src_len, (u64)fh.hdr_size, S, h0, nmax, sres);
                    LAUNCH(c, "zstd_index_stride", k_stride_tail, 1, 64, 0, d_src, (u64)src_len, (u64)fh.hdr_size, S, nmax, sres, sblk, st);
                    // a uniform flat frame (k_uni_head) needs nothing of what follows: its stream table is made here, beside the verdict
                    UniInfo *U = nullptr; FlatStream *usi = nullptr; u8 *usym = nullptr;
                    if (uni_wanted) {
                        U = arena_new<UniInfo>(c, 1); ZBlock *ublk = arena_new<ZBlock>(c, STRIDE_TAIL + 1);
                        usi = arena_new<FlatStream>(c, 4 * ((size_t)nmax + STRIDE_TAIL) + 1); usym = (u8 *)arena_alloc(c, 16);
                        if (!U || !ublk || !usi || !usym) return NAF_GPU_ENOMEM;
                        LAUNCH(c, "zstd_flat_uniform", k_uni_head, 1, 64, 0, d_src, (u64)fh.hdr_size, S, h0, nmax, (const u32 *)sres, (const ZBlock *)sblk, (const ZStat *)st, U, ublk, spec_min);
                        LAUNCH(c, "zstd_flat_uniform", k_uni_streams, cdiv(4ull * ((u64)nmax + STRIDE_TAIL) + 1, 256) + 1, 256, 0, d_src, (u64)fh.hdr_size, S, (const ZBlock *)ublk, U, usi, usym);
                    }
                    LAUNCH(c, "zstd_index_stride", k_stride_write, cdiv(nmax, 256), 256, 0, (u64)fh.hdr_size, S, h0, (const u32 *)sres, sblk, (const u32 *)U);
                    u32 res2[2] = { 0, 0 }; UniInfo hu; memset(&hu, 0, sizeof hu);
                    if (U) { void *hp[3] = { &hs, res2, &hu }; const void *dp[3] = { st, sres, U }; const size_t nb[3] = { sizeof hs, 8, sizeof hu }; rc = ctx_readbackv(c, 3, hp, dp, nb); }
                    else rc = ctx_readback2(c, &hs, st, sizeof hs, res2, sres, 8);
                    if (rc) return rc;
                    if (ctx_tracing(c)) ctx_trace(c, "[stride] len %zu S %u nmax %u prefix %u verdict %u err %u nblk %u\n", src_len, S, nmax, res2[0], res2[1], hs.err, hs.nblk);
                    if (ctx_tracing(c) && U) ctx_trace(c, "[uniform?] ok %u bad %u prefix %u blocks %u huffman %u regen %u total %llu\n", hu.ok, hu.bad, hu.np, hu.nblk, hu.nhb, hu.regen0, (unsigned long long)hu.total_out);
                    if (res2[1] == 1u && !hs.err && hs.nblk && hu.ok && !hu.bad) {
                        // every block repeats the first: the caller's emit kernel reads the streams in place (as below, without the block table)
                        ZFlat *zf = c->zflat;
                        const size_t frame_end = hs.end_off + (fh.checksum ? 4 : 0);
                        if (frame_end > src_len) return zerr(c, ZE_TRUNC, "checksum");
                        *consumed = frame_end;
                        const bool flat_tail = hu.last_raw != 0;
                        zf->src = d_src; zf->si = usi; zf->nslots = 4ull * hu.nhb; zf->sym = usym; zf->status = st; zf->ready = true;
                        const u32 tn = hu.last_raw & 0x7FFFFFFFu; const bool rle = (hu.last_raw >> 31) != 0;       // an RLE block stores one byte
                        zf->tail = flat_tail ? d_src + (hs.end_off - (rle ? 1u : tn)) : nullptr; zf->tail_q = hu.total_out - (flat_tail ? tn : 0u);
                        zf->tail_n = flat_tail ? (rle ? tn | 0x80000000u : tn) : 0u;
                        if (rg) { rg->got_lo = 0; rg->got_hi = hu.total_out; rg->ranged = false; }
                        *out_len = hu.total_out;
                        if (fh.has_fcs && fh.content_size != hu.total_out) return zerr(c, ZE_CORRUPT, "content size mismatch");
                        return 0;
                    }
                    if (res2[1] == 1u && !hs.err && hs.nblk) { blk = sblk; indexed = true; }
                    // runs of equal blocks (the sharded encoder's frames): NAF_GPU_STRIDE_RUNS=0: the universal front, as before
                    if (res2[1] == 2u && !hs.err && uni_wanted && !ctx_opt_is(c, "STRIDE_RUNS", '0')) {
                        UniRuns *R = arena_new<UniRuns>(c, 1); ZBlock *odd = arena_new<ZBlock>(c, STRIDE_TAIL + 1), *ublk0 = arena_new<ZBlock>(c, 1);
                        u32 *probe = arena_new<u32>(c, 2); UniInfo *U2 = arena_new<UniInfo>(c, 1);
                        FlatStream *usi2 = arena_new<FlatStream>(c, 4 * ((size_t)nmax + USEG_MAX) + 1); u8 *usym2 = (u8 *)arena_alloc(c, 16);
                        if (!R || !odd || !ublk0 || !probe || !U2 || !usi2 || !usym2) return NAF_GPU_ENOMEM;
                        for (u32 r = 0; r <= URUN_MAX; r++) {
                            LAUNCH(c, "zstd_index_stride", k_runs_next, 1, 64, 0, d_src, (u64)src_len, (u64)fh.hdr_size, S, h0, (const u32 *)sres, R, odd, probe, r == 0 ? 1 : 0);
                            if (r < URUN_MAX) LAUNCH(c, "zstd_index_stride", k_runs_probe, cdiv(nmax, 256), 256, 0, d_src, (u64)src_len, S, h0, (const UniRuns *)R, probe);
                        }
                        LAUNCH(c, "zstd_flat_uniform", k_uni_head_runs, 1, 64, 0, d_src, (u64)fh.hdr_size, h0, R, odd, U2, ublk0, spec_min);
                        LAUNCH(c, "zstd_flat_uniform", k_uni_streams_runs, cdiv(4ull * ((u64)nmax + USEG_MAX) + 1, 256) + 1, 256, 0, d_src, S, (const UniRuns *)R, (const ZBlock *)odd, (const ZBlock *)ublk0, U2, usi2, usym2);
                        UniInfo hu2; memset(&hu2, 0, sizeof hu2);
                        if ((rc = ctx_readback(c, &hu2, U2, sizeof hu2))) return rc;
                        if (ctx_tracing(c)) ctx_trace(c, "[runs?] ok %u bad %u first run %u blocks %u huffman %u total %llu\n", hu2.ok, hu2.bad, hu2.np, hu2.nblk, hu2.nhb, (unsigned long long)hu2.total_out);
                        if (hu2.ok && !hu2.bad) {
                            ZFlat *zf = c->zflat;
                            const size_t frame_end = hu2.end_off + (fh.checksum ? 4 : 0);
                            if (frame_end > src_len) return zerr(c, ZE_TRUNC, "checksum");
                            *consumed = frame_end;
                            const bool flat_tail = hu2.last_raw != 0;
                            zf->src = d_src; zf->si = usi2; zf->nslots = 4ull * hu2.nhb; zf->sym = usym2; zf->status = st; zf->ready = true;
                            const u32 tn = hu2.last_raw & 0x7FFFFFFFu; const bool rle = (hu2.last_raw >> 31) != 0;
                            zf->tail = flat_tail ? d_src + (hu2.end_off - (rle ? 1u : tn)) : nullptr; zf->tail_q = hu2.total_out - (flat_tail ? tn : 0u);
                            zf->tail_n = flat_tail ? (rle ? tn | 0x80000000u : tn) : 0u;
                            if (rg) { rg->got_lo = 0; rg->got_hi = hu2.total_out; rg->ranged = false; }
                            *out_len = hu2.total_out;
                            if (fh.has_fcs && fh.content_size != hu2.total_out) return zerr(c, ZE_CORRUPT, "content size mismatch");
                            return 0;
                        }
                    }
                }
            }
        }
        const u32 *skip = nullptr;
        if (!indexed) {
        // the first block start of a chunk lies in its first 20 KiB whenever blocks compress to less than that (this build's 32 KiB
        // blocks of packed bases: 16 KiB); the passes behind it only run for the chunks still without a candidate
        const u32 win0 = chunk == SPEC_CHUNK ? SPEC_WINDOW0 : win1;
        LAUNCH(c, "zstd_index_find", k_spec_find, cdiv(win0, 256 * 16) * (nchunks - 1), 256, 0, d_src, (u64)src_len, nchunks, first, 0u, win0, chunk, skip);
        if (win1 > win0) LAUNCH(c, "zstd_index_find2", k_spec_find, cdiv(win1 - win0, 256 * 16) * (nchunks - 1), 256, 0, d_src, (u64)src_len, nchunks, first, win0, win1, chunk, skip);
        if (win2 > win1) LAUNCH(c, "zstd_index_find2", k_spec_find, cdiv(win2 - win1, 256 * 16) * (nchunks - 1), 256, 0, d_src, (u64)src_len, nchunks, first, win1, win2, chunk, skip);
        LAUNCH(c, "zstd_index_land", k_spec_land, gl, 64, 0, d_src, (u64)src_len, (const u64 *)first, nchunks, land, chunk, skip);
        LAUNCH(c, "zstd_index_land2", k_spec_land2, gl, 64, 0, d_src, (u64)src_len, (const u64 *)first, (const u64 *)land, nchunks, G, chunk, skip);
        LAUNCH(c, "zstd_index_resolve", k_spec_resolve, 1, 64, 0, d_src, (u64)src_len, (const u64 *)land, (const u64 *)G, nchunks, start, st, chunk, skip);
        LAUNCH(c, "zstd_index_count", (k_spec_walk<false>), gl, 64, 0, d_src, (u64)src_len, (const u64 *)start, nchunks, cnt, (ZBlock *)nullptr, st, skip);
        u64 *d_tot = cnt + nchunks + 1;
        if ((rc = scan_exclusive_u64(c, cnt, nchunks, d_tot))) return rc;
        u64 tot = 0;
        rc = ctx_readback2(c, &hs, st, sizeof hs, &tot, d_tot, 8); if (rc) return rc;
        if (!hs.err && tot > 0 && tot < 0x7FFFFFFFull) {
            blk = arena_new<ZBlock>(c, tot);
            if (!blk) return NAF_GPU_ENOMEM;
            LAUNCH(c, "zstd_index_write", (k_spec_walk<true>), gl, 64, 0, d_src, (u64)src_len, (const u64 *)start, nchunks, cnt, blk, st, (const u32 *)nullptr);
            hs.nblk = (u32)tot; indexed = true;
        } else {
            HIP_TRY(c, hipMemsetAsync(st, 0, sizeof(ZStat), c->stream));     // not one well-formed frame for the parallel walk: serial walk decides
        }
        }
    }
    if (!indexed) {
        u32 cap = (u32)(src_len / 16 + 1024);                         // one pass for anything but pathological runs of empty blocks
        for (int attempt = 0; attempt < 2; attempt++) {
            blk = arena_new<ZBlock>(c, cap);
            if (!blk) return NAF_GPU_ENOMEM;
            const u32 win = src_len < SCAN_WIN ? (u32)((src_len + 15) & ~15ull) + 16 : SCAN_WIN;
            LAUNCH(c, "zstd_scan_blocks", k_scan_blocks, 1, 64, win + 16, d_src, (u64)src_len, (u64)fh.hdr_size, blk, cap, st, win);
            rc = ctx_readback(c, &hs, st, sizeof hs); if (rc) return rc;
            if (hs.err) return zerr(c, hs.err, "block headers");
            if (hs.nblk <= cap) break;
            cap = hs.nblk;
        }
    }
    u32 nblk = hs.nblk;
    size_t frame_end = hs.end_off + (fh.checksum ? 4 : 0);
    if (frame_end > src_len) return zerr(c, ZE_TRUNC, "checksum");
    *consumed = frame_end;

    // ---- parse + ownership
    i32 *own = arena_new<i32>(c, (size_t)nblk * 4);
    u64 *seq_cnt = arena_new<u64>(c, (size_t)nblk + 1), *sizes = arena_new<u64>(c, (size_t)nblk + 1);
    if (!own || !seq_cnt || !sizes) return NAF_GPU_ENOMEM;
    i32 *own_huf = own, *own_ll = own + nblk, *own_of = own + 2 * (size_t)nblk, *own_ml = own + 3 * (size_t)nblk;
    u32 g = cdiv(nblk, 64);
    LAUNCH(c, "zstd_parse_blocks", k_parse_blocks, g, 64, 0, d_src, blk, nblk, own_huf, own_ll, own_of, own_ml, seq_cnt, sizes, st);
    LAUNCH(c, "zstd_huf_dedup", k_huf_dedup, g, 64, 0, d_src, (const ZBlock *)blk, nblk, own_huf, st);
    if ((rc = scan_inclusive_max_i32(c, own_huf, nblk))) return rc;
    u64 *d_total_out = (u64 *)((u8 *)st + offsetof(ZStat, total_out));
    // Speculative continuation.  Most frames that are long enough to matter are literal-only (this build's own sequence, mask and
    // quality streams); for those nothing below needs the host: block sizes are final after the parse, so offsets, the block range
    // of a byte-range request, the Huffman tables and the part boundaries of a split decode are queued right away and the counters
    // come back in ONE read-back.  A frame that does have sequences then takes the long way from here (its tables are kept).
    const bool spec = nblk > spec_min && !fuse;
    u64 *r4 = nullptr, *ends = nullptr; u8 *huf_pool = nullptr; u32 pool_cap = 0; bool tables_built = false; bool ranged_build = false; bool two_phase = false;
    u64 h4[5] = { 0, 0, 0, 0, 0 }, hends[ZSPLIT_MAX] = { 0 };
    bool late_build = false;
    if (spec) {
        if ((rc = scan_exclusive_u64(c, sizes, nblk, d_total_out))) return rc;
        u64 *extra = arena_new<u64>(c, 8 + ZSPLIT_MAX); if (!extra) return NAF_GPU_ENOMEM;
        if (rg && rg->want_hi > rg->want_lo) {
            r4 = extra;
            LAUNCH(c, "zstd_find_range", k_find_range, 1, 64, 0, (const u64 *)sizes, nblk, (const u64 *)d_total_out, rg->want_lo, rg->want_hi, (const i32 *)own_huf, r4);
            ranged_build = true;
        }
        const u64 want_pool = (u64)nblk * HUF_TAB_MAX + 4096;
        pool_cap = want_pool > 0xFFFFF000ull ? 0xFFFFF000u : (u32)want_pool;
        huf_pool = (u8 *)arena_alloc(c, pool_cap);
        if (!huf_pool) return NAF_GPU_ENOMEM;
        if (nblk <= 16384) {
            // a stream of a few thousand blocks (a soft-masked genome's mask: a tree per block): a wavefront per block, all at once --
            // one lane per tree (k_build_huf) is 0.35 ms of serial table building in front of the literals there
            LAUNCH(c, "zstd_build_huf", k_build_huf_lds, nblk, 64, 0, d_src, blk, nblk, huf_pool, pool_cap, st, 0u, (const i32 *)own_huf);
        } else {
            LAUNCH(c, "zstd_build_huf", k_build_huf_few, 1024, 64, 0, d_src, blk, nblk, huf_pool, pool_cap, st, (const u64 *)r4, (const i32 *)own_huf);
            // (a caller that can read flat blocks in place gets the flat trees recognised now and the other tables later: see phase 2 below)
            two_phase = c->zflat && !rg && !always_table;
            if (two_phase) LAUNCH(c, "zstd_build_huf", k_flat_find_main, FIND_MAIN_TREES, 64, 0, d_src, blk, nblk, huf_pool, pool_cap, st, (const i32 *)own_huf);
            else late_build = true;
            // (the group builder for frames of MANY distinct trees is queued once the counters say there are that many: its workgroups hold
            // 64 KB of LDS each, and beside a Huffman walk of another stream -- which fills every CU's LDS -- even workgroups that find
            // nothing to do waited a millisecond to start: a FASTQ's sequence frame, one tree, behind its quality frame's walk)
        }
        ZSplit *sp = c->zsplit;
        if (sp && !rg && sp->parts >= 2) {
            ends = extra + 8;
            for (int k = 0; k + 1 < sp->parts; k++) {
                u32 hi_b = (u32)((u64)nblk * (k + 1) / sp->parts) & ~(HUF_BLOCKS_PER_WG - 1u);
                HIP_TRY(c, hipMemcpyAsync(ends + k, sizes + hi_b, 8, hipMemcpyDeviceToDevice, c->stream));
            }
        }
        u64 hx[8 + ZSPLIT_MAX];
        rc = ctx_readback2(c, &hs, st, sizeof hs, hx, extra, sizeof hx); if (rc) return rc;
        memcpy(h4, hx, sizeof h4); memcpy(hends, hx + 8, sizeof hends);
        if (late_build && !hs.err && hs.n_huf_distinct > HUF_FEW) {
            if ((rc = launch_build_huf(c, nblk, d_src, blk, nblk, huf_pool, pool_cap, st, 0u, (const u64 *)r4, always_table, (const i32 *)own_huf, 1u, 0u))) return rc;
            rc = ctx_readback(c, &hs, st, sizeof hs); if (rc) return rc;
        }
        tables_built = true;
    } else {
        rc = ctx_readback(c, &hs, st, sizeof hs); if (rc) return rc;
    }
    if (hs.err) return zerr(c, hs.err, "block parse");
    u32 n_seq_blk = hs.n_seq_blk, n_huf_def = hs.n_huf_def;
    // fused decode+emit needs every block to be a literal-only Huffman block
    if (fuse && !(n_seq_blk == 0 && hs.n_plain_huf == nblk && nblk > 0)) return ZSTD_NEED_TWO_PASS;
    const bool lit_only_spec = spec && n_seq_blk == 0;
    if (spec && n_seq_blk && ranged_build) {
        // the block range was worked out from sizes that sequences will change: forget those tables
        tables_built = false;
        HIP_TRY(c, hipMemsetAsync((u8 *)st + offsetof(ZStat, huf_pool_used), 0, 4, c->stream));
        HIP_TRY(c, hipMemsetAsync((u8 *)st + offsetof(ZStat, max_huf_log), 0, 4, c->stream));
        HIP_TRY(c, hipMemsetAsync((u8 *)st + offsetof(ZStat, n_flat), 0, 4, c->stream));
        HIP_TRY(c, hipMemsetAsync((u8 *)st + offsetof(ZStat, n_huf_built), 0, 4, c->stream));
    }
    // (a final Raw block is allowed: this build's encoder puts the byte with the padding nibble of an odd stream there, so that it does
    // not bring a seventeenth symbol into the last Huffman block)
    const bool flat_tail = hs.last_raw != 0 && nblk >= 2 && hs.n_plain_huf == nblk - 1;
    if (ctx_tracing(c) && c->zflat) ctx_trace(c, "[flat?] spec %d nblk %u seq_blk %u distinct %u built %u n_flat %u log %u plain %u last_raw %u always %u\n", (int)spec, nblk, n_seq_blk, hs.n_huf_distinct, hs.n_huf_built, hs.n_flat, hs.max_huf_log, hs.n_plain_huf, hs.last_raw, always_table);
    if (c->zflat && lit_only_spec && nblk > 0 && hs.n_huf_distinct == 1 && hs.n_huf_built == 1 && hs.n_flat == 1 && hs.max_huf_log == 4 && (hs.n_plain_huf == nblk || flat_tail) && !always_table) {
        // every block a plain Huffman block of the same flat 4-bit tree: the caller's emit kernel reads the streams in place
        ZFlat *zf = c->zflat;
        const u32 nhb = flat_tail ? nblk - 1 : nblk;                  // the Huffman blocks
        FlatStream *si = arena_new<FlatStream>(c, 4 * (size_t)nhb + 1); u8 *d_sym = (u8 *)arena_alloc(c, 16);
        if (!si || !d_sym) return NAF_GPU_ENOMEM;
        LAUNCH(c, "zstd_set_offsets", k_set_offsets, g, 64, 0, blk, nblk, (const u64 *)sizes, (u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr);
        LAUNCH(c, "zstd_flat_streams", k_flat_streams, cdiv(4ull * nhb, 256), 256, 0, d_src, (const ZBlock *)blk, nhb, (const i32 *)own_huf, (const u8 *)huf_pool, si, d_sym, st, (const u64 *)d_total_out, flat_tail ? (u64)(hs.last_raw & 0x7FFFFFFFu) : 0ull);
        zf->src = d_src; zf->si = si; zf->nslots = 4ull * nhb; zf->sym = d_sym; zf->status = st; zf->ready = true;
        {
            const u32 tn = hs.last_raw & 0x7FFFFFFFu; const bool rle = (hs.last_raw >> 31) != 0;       // an RLE block stores one byte
            zf->tail = flat_tail ? d_src + (hs.end_off - (rle ? 1u : tn)) : nullptr; zf->tail_q = hs.total_out - (flat_tail ? tn : 0u);
            zf->tail_n = flat_tail ? (rle ? tn | 0x80000000u : tn) : 0u;
        }
        if (rg) { rg->got_lo = 0; rg->got_hi = hs.total_out; rg->ranged = false; }      // nothing was decoded: the emit kernel finds any byte of the stream itself
        *out_len = hs.total_out;
        if (fh.has_fcs && fh.content_size != hs.total_out) return zerr(c, ZE_CORRUPT, "content size mismatch");
        return 0;
    }
    // Most blocks flat, some not (ctx.h: ZFlat, `cls`): the blocks that are not, and their neighbours, are decoded into d_dst at their
    // natural offsets; the caller's emit reads the rest in place.  Whole-stream calls only; a frame whose blocks mostly need
    // decoding takes the ordinary path below (NAF_GPU_FLAT_MIXED=0: always).
    {
        const char *fm = ctx_opt(c, "FLAT_MIXED");
        if (c->zflat && lit_only_spec && nblk > 0 && !always_table && !rg && hs.flat_main_inv && d_dst && hs.total_out <= dst_cap && !(fm && fm[0] == '0')) {
            const u32 main = 0xFFFFFFFFu - hs.flat_main_inv;
            ZFlat *zf = c->zflat;
            FlatStream *si = arena_new<FlatStream>(c, 4 * (size_t)nblk + 1); u8 *d_sym = (u8 *)arena_alloc(c, 16);
            u8 *cls0 = (u8 *)arena_alloc(c, nblk), *cls = (u8 *)arena_alloc(c, nblk); u32 *d_nx = arena_new<u32>(c, 2);
            if (!si || !d_sym || !cls0 || !cls || !d_nx) return NAF_GPU_ENOMEM;
            HIP_TRY(c, hipMemsetAsync(d_nx, 0, 8, c->stream));
            LAUNCH(c, "zstd_set_offsets", k_set_offsets, g, 64, 0, blk, nblk, (const u64 *)sizes, (u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr);
            LAUNCH(c, "zstd_flat_class", k_flat_mark_owner, g, 64, 0, d_src, blk, nblk, (const i32 *)own_huf, main, two_phase ? 1u : 0u);
            LAUNCH(c, "zstd_flat_class", k_flat_sym, 1, 256, 0, d_src, (const ZBlock *)blk, main, (const u8 *)huf_pool, d_sym);
            LAUNCH(c, "zstd_flat_class", k_flat_class, g, 64, 0, (const ZBlock *)blk, nblk, (const i32 *)own_huf, cls0, d_nx + 1);
            LAUNCH(c, "zstd_flat_class", k_flat_class2, g, 64, 0, (const u8 *)cls0, nblk, cls, d_nx);
            u32 nx2[2] = { 0, 0 };
            if ((rc = ctx_readback(c, nx2, d_nx, 8))) return rc;
            const u32 n_dec = nx2[0], n_walk = nx2[1];
            if (ctx_tracing(c)) ctx_trace(c, "[flat mixed] nblk %u decoded %u main %u\n", nblk, n_dec, main);
            if ((u64)n_dec * 2 <= nblk) {
                LAUNCH(c, "zstd_flat_streams", k_flat_streams_mixed, cdiv(4ull * nblk, 256), 256, 0, d_src, (const ZBlock *)blk, nblk, (const u8 *)cls, si, st, (const u64 *)d_total_out);
                zf->decoded_ev = nullptr; zf->later = nullptr;
                if (n_dec) {
                    // The decode of those blocks -- with the tables still to be built for them -- is handed back to the caller as a job: it
                    // runs once the caller has queued its tile index, on the caller's spare stream when there is one, beside the emit of
                    // the flat tiles, which needs none of it (zstd_flat_later).
                    const bool pending = two_phase && hs.n_huf_distinct > HUF_FEW;
                    const ZStat hs0 = hs; naf_gpu_ctx *mc = c; naf_gpu_ctx *aux = zf->aux;
                    const u64 src_len64 = (u64)src_len;
                    zf->later = new std::function<int()>([=]() -> int {
                        naf_gpu_ctx *c = aux ? aux : mc;
                        if (c != mc) HIP_TRY(mc, hipStreamWaitEvent(c->stream, mc->split_ev[0], 0));      // recorded by the caller behind its tile index
                        const u32 plog = n_walk ? huf_par_plog(c, hs0.max_lit_regen, n_walk) : 0u;
                        u32 max_log = hs0.max_huf_log;
                        if (pending && n_walk && !plog) {
                            // the one-lane-per-stream kernel takes its tables from the pool: the trees left out so far, now
                            { int r3 = launch_build_huf(c, nblk, d_src, blk, nblk, huf_pool, pool_cap, st, 0u, (const u64 *)nullptr, 0u, (const i32 *)own_huf, 1u, 2u); if (r3) { if (c != mc) memcpy(mc->err, c->err, sizeof mc->err); return r3; } }
                            ZStat h2; int r2 = ctx_readback(c, &h2, st, sizeof h2);
                            if (r2) { if (c != mc) memcpy(mc->err, c->err, sizeof mc->err); return r2; }
                            if (h2.err) return zerr(mc, h2.err, "Huffman tables");
                            max_log = h2.max_huf_log;
                        }
                        if (hs0.n_plain_huf != nblk) LAUNCH(c, "zstd_copy_fill", k_copy_fill<64>, 4 * nblk, 64, 0, d_src, (const ZBlock *)blk, nblk, d_dst, (u8 *)nullptr, 0u, (const u8 *)cls);
                        LAUNCH(c, "zstd_flat_literals", k_flat_literals<64>, 4 * nblk, 64, 0, d_src, (const ZBlock *)blk, nblk, (const i32 *)own_huf, (const u8 *)huf_pool, d_dst, (u8 *)nullptr, st, 0u, (const u8 *)cls);
                        if (n_walk && plog) {
                            // (k_huf_par builds the tables it lacks itself, a workgroup at a time, in LDS)
                            const u32 slot = pending ? (u32)HUF_TAB_MAX : huf_slot_bytes(max_log);
                            LAUNCH(c, "zstd_huf_literals", k_huf_par, cdiv((u64)nblk << (plog + 2), 64), 64, (plog >= 4 ? 1u : 16u >> plog) * slot,
                                   d_src, (const ZBlock *)blk, nblk, (const i32 *)own_huf, (const u8 *)huf_pool, slot, d_dst, (u8 *)nullptr, st, 0u, plog, (const u8 *)cls, 1u, huf_par_margin_env(c), pending ? 1u : 0u, src_len64);
                        } else if (n_walk) {
                            const u32 slot = huf_slot_bytes(max_log), ipitch = max_log > 7 ? HUF_IROW_BIG : HUF_IROW;
                            EmitP ep; memset(&ep, 0, sizeof ep);
                            LAUNCH(c, "zstd_huf_literals", (k_huf_literals<false>), cdiv(nblk, HUF_BLOCKS_PER_WG), 64, slot * HUF_BLOCKS_PER_WG + 64 * ipitch + 64 * HUF_OROW + 512,
                                   d_src, (const ZBlock *)blk, nblk, (const i32 *)own_huf, (const u8 *)huf_pool, slot, d_dst, (u8 *)nullptr, st, 0u, ep, (u8 *)nullptr, ipitch, src_len64, 1u, (const u8 *)cls);
                        }
                        return 0;
                    });
                }
                zf->src = d_src; zf->si = si; zf->nslots = 4ull * nblk; zf->sym = d_sym; zf->status = st; zf->ready = true;
                zf->tail = nullptr; zf->tail_q = hs.total_out; zf->tail_n = 0; zf->cls = cls; zf->n_decoded = n_dec; zf->n_walk = n_walk;
                *out_len = hs.total_out;
                if (fh.has_fcs && fh.content_size != hs.total_out) return zerr(c, ZE_CORRUPT, "content size mismatch");
                return 0;
            }
        }
    }
    if (two_phase && hs.n_huf_distinct > HUF_FEW) {
        // not a frame for the in-place emit after all: the tables phase 1 left out, now
        if ((rc = launch_build_huf(c, nblk, d_src, blk, nblk, huf_pool, pool_cap, st, 0u, (const u64 *)nullptr, 0u, (const i32 *)own_huf, 1u, 2u))) return rc;
        rc = ctx_readback(c, &hs, st, sizeof hs); if (rc) return rc;
        if (hs.err) return zerr(c, hs.err, "Huffman tables");
    }
    u64 *d_total_seq = (u64 *)((u8 *)st + offsetof(ZStat, total_seq));
    FseE *fse_pool = nullptr; u32 *o_ll = nullptr, *o_ml = nullptr, *o_of = nullptr;
    if (n_seq_blk) {
        if ((rc = scan_inclusive_max_i32(c, own_ll, nblk))) return rc;
        if ((rc = scan_inclusive_max_i32(c, own_of, nblk))) return rc;
        if ((rc = scan_inclusive_max_i32(c, own_ml, nblk))) return rc;
        if ((rc = scan_exclusive_u64(c, seq_cnt, nblk, d_total_seq))) return rc;
        u32 fse_cap = n_seq_blk * (512 + 256 + 512);
        fse_pool = arena_new<FseE>(c, fse_cap);
        if (!fse_pool) return NAF_GPU_ENOMEM;
        LAUNCH(c, "zstd_build_fse", k_build_fse, cdiv(3 * (u64)nblk, 64), 64, 0, d_src, blk, nblk, fse_pool, fse_cap, st);
        rc = ctx_readback(c, &hs, st, sizeof hs); if (rc) return rc;
        if (hs.err) return zerr(c, hs.err, "table build");
        size_t ns = hs.total_seq ? hs.total_seq : 1;
        o_ll = arena_new<u32>(c, ns); o_ml = arena_new<u32>(c, ns); o_of = arena_new<u32>(c, ns);
        if (!o_ll || !o_ml || !o_of) return NAF_GPU_ENOMEM;
    }
    u32 *seq_list = nullptr;                                 // indices of the blocks that have sequences, in order
    u64 *seq_rank = nullptr;                                 // blocks with sequences in front of block i
    if (n_seq_blk) {
        seq_list = arena_new<u32>(c, n_seq_blk);
        u64 *flag = arena_new<u64>(c, (size_t)nblk + 1); seq_rank = flag;
        if (!seq_list || !flag) return NAF_GPU_ENOMEM;
        LAUNCH(c, "zstd_seq_flag", k_seq_flag, g, 64, 0, (const ZBlock *)blk, nblk, flag);
        if ((rc = scan_exclusive_u64(c, flag, nblk, (u64 *)nullptr))) return rc;
        LAUNCH(c, "zstd_seq_list", k_seq_list, g, 64, 0, (const ZBlock *)blk, nblk, (const u64 *)flag, seq_list);
    }
    if (!lit_only_spec) {
        // (a lane per block, every lane on a chain of its own: a frame of a few thousand blocks spreads over more wavefronts, 16 lanes each)
        const u32 dsl = nblk < 32768 ? 16u : 64u;
        // blocks under tables of their own (libzstd's) by a wavefront each with the tables in LDS, the others a lane per block
        const u32 all_wave = ctx_opt_is(c, "SEQ_WAVE", 'a') ? 1u : 0u;
        const u32 own_tabs = (n_seq_blk && !ctx_opt_is(c, "SEQ_WAVE", '0')) ? 1u + all_wave : 0u;
        u32 *wave_list = own_tabs ? arena_new<u32>(c, n_seq_blk) : nullptr;
        if (own_tabs && !wave_list) return NAF_GPU_ENOMEM;
        LAUNCH(c, "zstd_decode_seq", k_decode_seq, cdiv(nblk, dsl), dsl, 0, d_src, blk, nblk, (const i32 *)own_ll, (const i32 *)own_of, (const i32 *)own_ml,
               (const u64 *)seq_cnt, (const FseE *)fse_pool, (const FseE *)c->d_predef, o_ll, o_ml, o_of, sizes, st, own_tabs, wave_list);
        const u32 wgrid = n_seq_blk < 8192u ? n_seq_blk : 8192u;
        if (own_tabs && ctx_opt_is(c, "SEQ_WAVE", 'l'))       // (kept as a cross-check: one lane walking the block with the general routine's shape)
            LAUNCH(c, "zstd_decode_seq", k_decode_seq_wave, wgrid, 64, 0, d_src, blk, (const u32 *)wave_list, (const i32 *)own_ll, (const i32 *)own_of, (const i32 *)own_ml,
                   (const u64 *)seq_cnt, (const FseE *)fse_pool, (const FseE *)c->d_predef, o_ll, o_ml, o_of, sizes, st);
        else if (own_tabs)
            LAUNCH(c, "zstd_decode_seq", k_decode_seq_wave2, wgrid, 64, 0, d_src, blk, (const u32 *)wave_list, (const i32 *)own_ll, (const i32 *)own_of, (const i32 *)own_ml,
                   (const u64 *)seq_cnt, (const FseE *)fse_pool, (const FseE *)c->d_predef, o_ll, o_ml, o_of, sizes, st, all_wave | (ctx_opt_is(c, "SEQ_REP", 'w') ? 2u : 0u));
        if (n_seq_blk) LAUNCH(c, "zstd_rep_fast", k_rep_fast, cdiv(n_seq_blk, 256), 256, 0, blk, (const u32 *)seq_list, n_seq_blk, st);
        if ((rc = scan_exclusive_u64(c, sizes, nblk, d_total_out))) return rc;
        rc = ctx_readback(c, &hs, st, sizeof hs); if (rc) return rc;
        if (hs.err) return zerr(c, hs.err, "sequences");
    }
    const u32 max_seq_regen = hs.max_seq_regen;
    if (ctx_tracing(c) && n_seq_blk) ctx_trace(c, "[seq] blocks %u with sequences %u sequences %llu out %llu\n", nblk, n_seq_blk, (unsigned long long)hs.total_seq, (unsigned long long)hs.total_out);
    // entry states matter only when some sequence of the frame uses a repeat code (this build's own LZ blocks never do)
    if (n_seq_blk && (hs.rep_slow & 1)) LAUNCH(c, "zstd_rep_chain", k_rep_chain, 1, 64, 0, blk, nblk);
    *out_len = hs.total_out;
    if (fh.has_fcs && fh.content_size != hs.total_out) return zerr(c, ZE_CORRUPT, "content size mismatch");
    // A frame that is mostly flat AND has a few blocks with matches -- what libzstd makes of packed random bases: one 4-bit tree,
    // treeless blocks behind it, a chance match every few dozen blocks -- takes the mostly-flat way too: the blocks with sequences, the
    // blocks their matches copy from (k_seq_sources) and the neighbours of both are decoded into d_dst (literals, then the executor,
    // as below), everything else is read in place by the caller's emit.  Whole-stream calls (NAF_GPU_FLAT_SEQ=0: never).
    {
        const char *fm = ctx_opt(c, "FLAT_MIXED"), *fsq = ctx_opt(c, "FLAT_SEQ");
        if (c->zflat && spec && n_seq_blk && tables_built && !fuse && !always_table && !rg && hs.flat_main_inv && d_dst && hs.total_out <= dst_cap &&
            (u64)n_seq_blk * 8 <= nblk && !(fm && fm[0] == '0') && !(fsq && fsq[0] == '0')) {
            const u32 main = 0xFFFFFFFFu - hs.flat_main_inv;
            ZFlat *zf = c->zflat;
            FlatStream *si = arena_new<FlatStream>(c, 4 * (size_t)nblk + 1); u8 *d_sym = (u8 *)arena_alloc(c, 16);
            u8 *cls0 = (u8 *)arena_alloc(c, nblk), *cls = (u8 *)arena_alloc(c, nblk); u32 *d_nx = arena_new<u32>(c, 2);
            u32 *done2 = arena_new<u32>(c, nblk); u8 *lits = (u8 *)arena_alloc(c, hs.total_out + 16);
            if (!si || !d_sym || !cls0 || !cls || !d_nx || !done2 || !lits) return NAF_GPU_ENOMEM;
            HIP_TRY(c, hipMemsetAsync(d_nx, 0, 8, c->stream));
            LAUNCH(c, "zstd_set_offsets", k_set_offsets, g, 64, 0, blk, nblk, (const u64 *)sizes, done2, (u32 *)nullptr, (u32 *)nullptr);
            LAUNCH(c, "zstd_flat_class", k_flat_mark_owner, g, 64, 0, d_src, blk, nblk, (const i32 *)own_huf, main, 0u);
            LAUNCH(c, "zstd_flat_class", k_flat_sym, 1, 256, 0, d_src, (const ZBlock *)blk, main, (const u8 *)huf_pool, d_sym);
            LAUNCH(c, "zstd_flat_class", k_flat_class, g, 64, 0, (const ZBlock *)blk, nblk, (const i32 *)own_huf, cls0, d_nx + 1);
            LAUNCH(c, "zstd_flat_class", k_seq_sources, cdiv(n_seq_blk, 64), 64, 0, (const ZBlock *)blk, (const u32 *)seq_list, n_seq_blk, (const u64 *)sizes, (const u32 *)o_ll, (const u32 *)o_ml, (const u32 *)o_of, cls0);
            LAUNCH(c, "zstd_flat_class", k_flat_class2, g, 64, 0, (const u8 *)cls0, nblk, cls, d_nx);
            u32 nx2[2] = { 0, 0 };
            if ((rc = ctx_readback(c, nx2, d_nx, 8))) return rc;
            const u32 n_dec = nx2[0], n_walk = nx2[1];
            if (ctx_tracing(c)) ctx_trace(c, "[flat mixed] nblk %u decoded %u main %u (blocks with sequences %u)\n", nblk, n_dec, main, n_seq_blk);
            if ((u64)n_dec * 2 <= nblk) {
                LAUNCH(c, "zstd_flat_streams", k_flat_streams_mixed, cdiv(4ull * nblk, 256), 256, 0, d_src, (const ZBlock *)blk, nblk, (const u8 *)cls, si, st, (const u64 *)d_total_out);
                zf->decoded_ev = nullptr;
                const ZStat hs0 = hs; naf_gpu_ctx *mc = c; naf_gpu_ctx *aux = zf->aux; const u64 ns_all = hs.total_seq;
                const u64 src_len64 = (u64)src_len;
                zf->later = new std::function<int()>([=]() -> int {
                    naf_gpu_ctx *c = aux ? aux : mc;
                    if (c != mc) HIP_TRY(mc, hipStreamWaitEvent(c->stream, mc->split_ev[0], 0));      // recorded by the caller behind its tile index
                    const u32 plog = n_walk ? huf_par_plog(c, hs0.max_lit_regen, n_walk) : 0u;
                    LAUNCH(c, "zstd_copy_fill", k_copy_fill<64>, 4 * nblk, 64, 0, d_src, (const ZBlock *)blk, nblk, d_dst, lits, 0u, (const u8 *)cls);
                    LAUNCH(c, "zstd_flat_literals", k_flat_literals<64>, 4 * nblk, 64, 0, d_src, (const ZBlock *)blk, nblk, (const i32 *)own_huf, (const u8 *)huf_pool, d_dst, lits, st, 0u, (const u8 *)cls);
                    if (n_walk && plog) {
                        const u32 slot = huf_slot_bytes(hs0.max_huf_log);
                        LAUNCH(c, "zstd_huf_literals", k_huf_par, cdiv((u64)nblk << (plog + 2), 64), 64, (plog >= 4 ? 1u : 16u >> plog) * slot,
                               d_src, (const ZBlock *)blk, nblk, (const i32 *)own_huf, (const u8 *)huf_pool, slot, d_dst, lits, st, 0u, plog, (const u8 *)cls, 1u, huf_par_margin_env(c), 0u, src_len64);
                    } else if (n_walk) {
                        const u32 slot = huf_slot_bytes(hs0.max_huf_log), ipitch = hs0.max_huf_log > 7 ? HUF_IROW_BIG : HUF_IROW;
                        EmitP ep; memset(&ep, 0, sizeof ep);
                        LAUNCH(c, "zstd_huf_literals", (k_huf_literals<false>), cdiv(nblk, HUF_BLOCKS_PER_WG), 64, slot * HUF_BLOCKS_PER_WG + 64 * ipitch + 64 * HUF_OROW + 512,
                               d_src, (const ZBlock *)blk, nblk, (const i32 *)own_huf, (const u8 *)huf_pool, slot, d_dst, lits, st, 0u, ep, (u8 *)nullptr, ipitch, src_len64, 1u, (const u8 *)cls);
                    }
                    const char *el = ctx_opt(c, "EXEC_LDS");
                    if (hs0.max_seq_regen <= EXEC_LDS && !(el && el[0] == '0'))
                         LAUNCH(c, "zstd_exec_seq", k_exec_seq_lds, n_seq_blk, 64, ((hs0.max_seq_regen + 1023u) & ~1023u) + 64u + 2u * EXEC_PJ_MAX, (const ZBlock *)blk, (const u32 *)seq_list, n_seq_blk, (const u64 *)sizes, nblk,
                               (const u32 *)o_ll, (const u32 *)o_ml, (const u32 *)o_of, (const u8 *)lits, d_dst, done2, st, (hs0.max_seq_regen + 1023u) & ~1023u);
                    else { const int rcx = launch_lz_exec(c, (const ZBlock *)blk, (const u32 *)seq_list, n_seq_blk, (const u64 *)sizes, (const u64 *)seq_cnt, nblk, ns_all, o_ll, o_ml, o_of, (const u8 *)lits, d_dst, done2, st); if (rcx) return rcx; }
                    return 0;
                });
                zf->src = d_src; zf->si = si; zf->nslots = 4ull * nblk; zf->sym = d_sym; zf->status = st; zf->ready = true;
                zf->tail = nullptr; zf->tail_q = hs.total_out; zf->tail_n = 0; zf->cls = cls; zf->n_decoded = n_dec; zf->n_walk = n_walk;
                return 0;
            }
        }
    }
    // Range request (multi-GPU sharding): decode only the blocks that feed [want_lo, want_hi).  Needs blocks that
    // do not reference earlier output, i.e. a frame without sequences (this build's own frames; reference-made
    // random-ACGT frames); otherwise the whole frame is decoded.
    u32 b_first = 0, b_count = nblk, huf_first = 0; u64 bias = 0;
    u32 seq_t0 = 0, seq_t1 = n_seq_blk;                          // the blocks with sequences among the decoded ones: seq_list[seq_t0 .. seq_t1)
    if (fuse) { d_dst = nullptr; dst_cap = ~(size_t)0; }
    if (rg) { rg->got_lo = 0; rg->got_hi = hs.total_out; rg->ranged = false; rg->own_buf = nullptr; }
    if (rg && n_seq_blk == 0 && nblk > 0 && rg->want_hi > rg->want_lo) {
        if (!lit_only_spec) {
            u64 *r4b = arena_new<u64>(c, 5); if (!r4b) return NAF_GPU_ENOMEM;
            LAUNCH(c, "zstd_find_range", k_find_range, 1, 64, 0, (const u64 *)sizes, nblk, (const u64 *)d_total_out, rg->want_lo, rg->want_hi, (const i32 *)own_huf, r4b);
            rc = ctx_readback(c, h4, r4b, 40); if (rc) return rc;
        }
        huf_first = (u32)h4[4];
        b_first = (u32)h4[0]; b_count = (u32)(h4[1] - h4[0]); bias = h4[2];
        rg->got_lo = h4[2]; rg->got_hi = h4[3]; rg->ranged = true;
        if (h4[3] - h4[2] > dst_cap) return ctx_fail(c, NAF_GPU_ECAP, "zstd range output needs %llu bytes, capacity %zu", (unsigned long long)(h4[3] - h4[2]), dst_cap);
        d_dst -= bias;                                       // block b lands at d_dst_orig + (out_off[b] - got_lo)
    } else if (rg && n_seq_blk && nblk > 0 && rg->want_hi > rg->want_lo && !(ctx_opt(c, "RANGE_CLOSURE") && ctx_opt(c, "RANGE_CLOSURE")[0] == '0')) {
        // blocks with matches: the range's dependency closure (kernels above).  On archives that are mostly literals -- what the
        // reference makes of a genome at its default level -- that is the range's own blocks and a few in front of them.
        u32 *f = arena_new<u32>(c, nblk); u64 *r4b = arena_new<u64>(c, 5 + 8); if (!f || !r4b) return NAF_GPU_ENOMEM;
        LAUNCH(c, "zstd_find_range", k_find_range, 1, 64, 0, (const u64 *)sizes, nblk, (const u64 *)d_total_out, rg->want_lo, rg->want_hi, (const i32 *)own_huf, r4b);
        LAUNCH(c, "zstd_range_closure", k_iota_u32, g, 64, 0, f, nblk);
        LAUNCH(c, "zstd_range_closure", k_seq_reach, cdiv(n_seq_blk, 64), 64, 0, (const ZBlock *)blk, (const u32 *)seq_list, n_seq_blk, (const u64 *)sizes, (const u32 *)o_ll, (const u32 *)o_ml, (const u32 *)o_of, f);
        LAUNCH(c, "zstd_range_closure", k_range_closure, 1, 64, 0, (const u32 *)f, (const u64 *)sizes, nblk, (const u64 *)d_total_out, (const u64 *)r4b, (const i32 *)own_huf, (const u64 *)seq_rank, n_seq_blk, r4b + 5);
        u64 h7[7]; rc = ctx_readback(c, h7, r4b + 5, sizeof h7); if (rc) return rc;
        const u64 need = h7[3] - h7[2];
        if (ctx_tracing(c)) ctx_trace(c, "[range] want %llu..%llu -> blocks %llu..%llu (bytes %llu..%llu of %llu), tables from %llu, seq blocks %llu..%llu of %u\n", (unsigned long long)rg->want_lo, (unsigned long long)rg->want_hi,
                    (unsigned long long)h7[0], (unsigned long long)h7[1], (unsigned long long)h7[2], (unsigned long long)h7[3], (unsigned long long)hs.total_out, (unsigned long long)h7[4], (unsigned long long)h7[5], (unsigned long long)h7[6], n_seq_blk);
        if (need < hs.total_out) {
            if (need > dst_cap) {
                // the caller sized its buffer for the range alone: take the closure's from the arena and say so (ZRange.own_buf)
                d_dst = (u8 *)arena_alloc(c, need + 64); if (!d_dst) return NAF_GPU_ENOMEM;
                dst_cap = need; rg->own_buf = d_dst;
            }
            huf_first = (u32)h7[4];
            b_first = (u32)h7[0]; b_count = (u32)(h7[1] - h7[0]); bias = h7[2];
            seq_t0 = (u32)h7[5]; seq_t1 = (u32)h7[6];
            rg->got_lo = h7[2]; rg->got_hi = h7[3]; rg->ranged = true;
            d_dst -= bias;
        } else if (hs.total_out > dst_cap) return ctx_fail(c, NAF_GPU_ECAP, "zstd output needs %llu bytes, capacity %zu", (unsigned long long)hs.total_out, dst_cap);
    } else if (hs.total_out > dst_cap) return ctx_fail(c, NAF_GPU_ECAP, "zstd output needs %llu bytes, capacity %zu", (unsigned long long)hs.total_out, dst_cap);

    u32 *done = nullptr; u8 *lit_scratch = nullptr;
    if (n_seq_blk) {
        done = arena_new<u32>(c, nblk);
        const u64 span = rg && rg->ranged ? rg->got_hi - rg->got_lo : hs.total_out;
        lit_scratch = (u8 *)arena_alloc(c, span + 16);
        if (!done || !lit_scratch) return NAF_GPU_ENOMEM;
        lit_scratch -= bias;                                     // indexed by a block's place in the whole output, like d_dst
    }
    LAUNCH(c, "zstd_set_offsets", k_set_offsets, g, 64, 0, blk, nblk, (const u64 *)sizes, done, (u32 *)nullptr, (u32 *)nullptr);
    bool copy_fill_done = false;
    if (n_huf_def) {
        // tables of the blocks that will be decoded (and of the earlier blocks that own a table in force there)
        u32 hb_end = b_first + b_count, hb_n = hb_end - huf_first;
        if (!tables_built) {
            pool_cap = (hb_n < n_huf_def ? hb_n : n_huf_def) * (u32)HUF_TAB_MAX + 4096u;   // largest table of either form
            huf_pool = (u8 *)arena_alloc(c, pool_cap);
            if (!huf_pool) return NAF_GPU_ENOMEM;
            if (hb_n && hb_n <= 512) LAUNCH(c, "zstd_build_huf", k_build_huf_lds, hb_n, 64, 0, d_src, blk, hb_end, huf_pool, pool_cap, st, huf_first, (const i32 *)own_huf);
            else if (hb_n) { if ((rc = launch_build_huf(c, hb_n, d_src, blk, hb_end, huf_pool, pool_cap, st, huf_first, (const u64 *)nullptr, (always_table || fuse) ? 1u : 0u, (const i32 *)own_huf, 0u, 0u))) return rc; }
            rc = ctx_readback(c, &hs, st, sizeof hs); if (rc) return rc;
            if (hs.err) return zerr(c, hs.err, "Huffman tables");
        }
        u32 slot = huf_slot_bytes(hs.max_huf_log);
        u32 b_end = b_first + b_count;
        if (fuse && hs.max_huf_log > 7) return ZSTD_NEED_TWO_PASS;
        EmitP ep; memset(&ep, 0, sizeof ep); if (fuse) ep = *fuse;
        u32 ipitch = hs.max_huf_log > 7 ? HUF_IROW_BIG : HUF_IROW;
        u32 ipitch_arg = ipitch | ((ctx_opt(c, "HUF_GENERIC") && ctx_opt(c, "HUF_GENERIC")[0] == '1') ? 0x8000u : 0u);
        if (b_count && fuse) LAUNCH(c, "zstd_huf_fused_emit", (k_huf_literals<true>), cdiv(b_count, HUF_BLOCKS_PER_WG), 64, slot * HUF_BLOCKS_PER_WG + 64 * ipitch + 512,
               d_src, (const ZBlock *)blk, b_end, (const i32 *)own_huf, (const u8 *)huf_pool, slot, d_dst, lit_scratch, st, b_first, ep, text, ipitch_arg, (u64)src_len, 0u, (const u8 *)nullptr);
        else if (b_count) {
            const u32 huf_lds = slot * HUF_BLOCKS_PER_WG + 64 * ipitch + 64 * HUF_OROW + 512;
            // blocks whose tree is flat go to k_flat_literals; the serial kernel is not launched when that is all of them
            const u32 flat_on = (hs.n_flat && !always_table) ? 1u : 0u;
            const bool serial_needed = !flat_on || hs.n_flat < hs.n_huf_built;
            const u32 plog = huf_par_plog(c, hs.max_lit_regen, b_count), par_lds = (plog >= 4 ? 1u : 16u >> plog) * slot;
            // a frame of few trees (this build's frame tree, libzstd's runs of treeless blocks): workgroups with ONE table in LDS, the
            // workgroups whose blocks are under several trees through a second launch of the plain kernel (NAF_GPU_HUF_SHARED=0: never)
            u8 *redo = nullptr;
            { const char *hsx = ctx_opt(c, "HUF_SHARED");
              if (serial_needed && !plog && (u64)hs.n_huf_distinct * 64 <= b_count && !(hsx && hsx[0] == '0')) {
                redo = (u8 *)arena_alloc(c, (size_t)nblk + 16); if (!redo) return NAF_GPU_ENOMEM;
                HIP_TRY(c, hipMemsetAsync(redo, 0, nblk, c->stream));
              } }
            const u32 huf_lds_shared = huf_lds - slot * (HUF_BLOCKS_PER_WG - 1u);
            ZSplit *sp = c->zsplit;
            const char *smin = ctx_opt(c, "SPLIT_MIN");                      // blocks per part below which a split is not worth its launches (tests lower it)
            const u32 split_min = smin ? (u32)atoi(smin) : 4096u;
            if (sp && !rg && n_seq_blk == 0 && b_first == 0 && b_count == nblk && b_count >= split_min * (u32)sp->parts && b_count >= 16u * HUF_BLOCKS_PER_WG * (u32)sp->parts) {
                // literal-only frame of a whole-text call: block ranges in order, an event behind each (see ZSplit); the raw / RLE
                // blocks first, so that a finished part is complete
                if (hs.n_plain_huf != nblk) LAUNCH(c, "zstd_copy_fill", k_copy_fill<256>, b_count, 256, 0, d_src, (const ZBlock *)blk, b_first + b_count, d_dst, lit_scratch, b_first, (const u8 *)nullptr);
                copy_fill_done = true;
                // output offsets of the part ends: they came with the counters when the frame took the speculative route
                if (ends) for (int k = 0; k + 1 < sp->parts; k++) sp->out_end[k] = hends[k];
                else {
                    u64 *e2 = arena_new<u64>(c, ZSPLIT_MAX); if (!e2) return NAF_GPU_ENOMEM;
                    for (int k = 0; k + 1 < sp->parts; k++) {
                        u32 hi_b = (u32)((u64)b_count * (k + 1) / sp->parts) & ~(HUF_BLOCKS_PER_WG - 1u);
                        HIP_TRY(c, hipMemcpyAsync(e2 + k, sizes + hi_b, 8, hipMemcpyDeviceToDevice, c->stream));
                    }
                    rc = ctx_readback(c, sp->out_end, e2, 8 * (size_t)(sp->parts - 1)); if (rc) return rc;
                }
                sp->out_end[sp->parts - 1] = hs.total_out;
                u32 lo_b = 0;
                for (int k = 0; k < sp->parts; k++) {
                    u32 hi_b = k + 1 == sp->parts ? b_count : (u32)((u64)b_count * (k + 1) / sp->parts) & ~(HUF_BLOCKS_PER_WG - 1u);
                    if (hi_b > lo_b && flat_on) LAUNCH(c, "zstd_flat_literals", k_flat_literals<256>, hi_b - lo_b, 256, 0, d_src, (const ZBlock *)blk, hi_b, (const i32 *)own_huf, (const u8 *)huf_pool, d_dst, lit_scratch, st, lo_b, (const u8 *)nullptr);
                    if (hi_b > lo_b && serial_needed && plog) LAUNCH(c, "zstd_huf_literals", k_huf_par, cdiv((u64)(hi_b - lo_b) << (plog + 2), 64), 64, par_lds,
                           d_src, (const ZBlock *)blk, hi_b, (const i32 *)own_huf, (const u8 *)huf_pool, slot, d_dst, lit_scratch, st, lo_b, plog, (const u8 *)nullptr, flat_on, huf_par_margin_env(c), 0u, (u64)src_len);
                    else if (hi_b > lo_b && serial_needed && redo) {
                        LAUNCH(c, "zstd_huf_literals", (k_huf_literals<false, true>), cdiv(hi_b - lo_b, HUF_BLOCKS_PER_WG), 64, huf_lds_shared,
                               d_src, (const ZBlock *)blk, hi_b, (const i32 *)own_huf, (const u8 *)huf_pool, slot, d_dst, lit_scratch, st, lo_b, ep, text, ipitch_arg, (u64)src_len, flat_on, (const u8 *)nullptr, redo);
                        LAUNCH(c, "zstd_huf_literals", (k_huf_literals<false>), cdiv(hi_b - lo_b, HUF_BLOCKS_PER_WG), 64, huf_lds,
                               d_src, (const ZBlock *)blk, hi_b, (const i32 *)own_huf, (const u8 *)huf_pool, slot, d_dst, lit_scratch, st, lo_b, ep, text, ipitch_arg, (u64)src_len, flat_on, (const u8 *)redo);
                    }
                    else if (hi_b > lo_b && serial_needed) LAUNCH(c, "zstd_huf_literals", (k_huf_literals<false>), cdiv(hi_b - lo_b, HUF_BLOCKS_PER_WG), 64, huf_lds,
                           d_src, (const ZBlock *)blk, hi_b, (const i32 *)own_huf, (const u8 *)huf_pool, slot, d_dst, lit_scratch, st, lo_b, ep, text, ipitch_arg, (u64)src_len, flat_on, (const u8 *)nullptr);
                    HIP_TRY(c, hipEventRecord(sp->ev[k], c->stream));
                    lo_b = hi_b;
                }
                sp->done = 1;
            } else {
                if (flat_on) LAUNCH(c, "zstd_flat_literals", k_flat_literals<256>, b_count, 256, 0, d_src, (const ZBlock *)blk, b_end, (const i32 *)own_huf, (const u8 *)huf_pool, d_dst, lit_scratch, st, b_first, (const u8 *)nullptr);
                if (serial_needed && plog) LAUNCH(c, "zstd_huf_literals", k_huf_par, cdiv((u64)b_count << (plog + 2), 64), 64, par_lds,
                   d_src, (const ZBlock *)blk, b_end, (const i32 *)own_huf, (const u8 *)huf_pool, slot, d_dst, lit_scratch, st, b_first, plog, (const u8 *)nullptr, flat_on, huf_par_margin_env(c), 0u, (u64)src_len);
                else if (serial_needed && redo) {
                    LAUNCH(c, "zstd_huf_literals", (k_huf_literals<false, true>), cdiv(b_count, HUF_BLOCKS_PER_WG), 64, huf_lds_shared,
                       d_src, (const ZBlock *)blk, b_end, (const i32 *)own_huf, (const u8 *)huf_pool, slot, d_dst, lit_scratch, st, b_first, ep, text, ipitch_arg, (u64)src_len, flat_on, (const u8 *)nullptr, redo);
                    LAUNCH(c, "zstd_huf_literals", (k_huf_literals<false>), cdiv(b_count, HUF_BLOCKS_PER_WG), 64, huf_lds,
                       d_src, (const ZBlock *)blk, b_end, (const i32 *)own_huf, (const u8 *)huf_pool, slot, d_dst, lit_scratch, st, b_first, ep, text, ipitch_arg, (u64)src_len, flat_on, (const u8 *)redo);
                }
                else if (serial_needed) LAUNCH(c, "zstd_huf_literals", (k_huf_literals<false>), cdiv(b_count, HUF_BLOCKS_PER_WG), 64, huf_lds,
                   d_src, (const ZBlock *)blk, b_end, (const i32 *)own_huf, (const u8 *)huf_pool, slot, d_dst, lit_scratch, st, b_first, ep, text, ipitch_arg, (u64)src_len, flat_on, (const u8 *)nullptr);
            }
        }
    }
    if (b_count && !fuse && !copy_fill_done && hs.n_plain_huf != nblk) LAUNCH(c, "zstd_copy_fill", k_copy_fill<256>, b_count, 256, 0, d_src, (const ZBlock *)blk, b_first + b_count, d_dst, lit_scratch, b_first, (const u8 *)nullptr);
    if (seq_t1 > seq_t0) {
        const char *el = ctx_opt(c, "EXEC_LDS");                      // "0": always the HBM executor (cross-check)
        const u32 nx = seq_t1 - seq_t0;
        if (max_seq_regen <= EXEC_LDS && !(el && el[0] == '0'))
             LAUNCH(c, "zstd_exec_seq", k_exec_seq_lds, nx, 64, ((max_seq_regen + 1023u) & ~1023u) + 64u + 2u * EXEC_PJ_MAX, (const ZBlock *)blk, (const u32 *)(seq_list + seq_t0), nx, (const u64 *)sizes, nblk,
                   (const u32 *)o_ll, (const u32 *)o_ml, (const u32 *)o_of, (const u8 *)lit_scratch, d_dst, done, st, (max_seq_regen + 1023u) & ~1023u);
        else if ((rc = launch_lz_exec(c, (const ZBlock *)blk, (const u32 *)(seq_list + seq_t0), nx, (const u64 *)sizes, (const u64 *)seq_cnt, nblk, hs.total_seq, o_ll, o_ml, o_of, (const u8 *)lit_scratch, d_dst, done, st))) return rc;
    }
    if (c->zsplit && c->zsplit->done) { c->zsplit->status = st; return 0; }      // the caller checks the status once the emit is queued (zstd_split_status)
    rc = ctx_readback(c, &hs, st, sizeof hs); if (rc) return rc;
#ifdef NAF_EXEC_PROF
    if (hs.prof[6]) fprintf(stderr, "[exec prof] blocks %llu seqs %llu | cycles per block: load+scan %llu literals %llu hops %llu final copies %llu serial %llu | serial matches per block %.1f\n", hs.prof[6], hs.prof[7],
                            hs.prof[0] / hs.prof[6], hs.prof[1] / hs.prof[6], hs.prof[2] / hs.prof[6], hs.prof[3] / hs.prof[6], hs.prof[4] / hs.prof[6], (double)hs.prof[5] / (double)hs.prof[6]);
#endif
    if (hs.err) return zerr(c, hs.err, "block decode");
    return 0;
}

// the job a mostly-flat frame's decoder left for after the caller's tile index (ctx.h: ZFlat.later); aux_used: the job went to the spare stream
int zstd_flat_later(naf_gpu_ctx *c, ZFlat *zf)
{
    if (!zf || !zf->later) return 0;
    std::function<int()> *f = (std::function<int()> *)zf->later; zf->later = nullptr;
    const int rc = (*f)();
    delete f;
    return rc;
}
void zstd_flat_drop(ZFlat *zf) { if (zf && zf->later) { delete (std::function<int()> *)zf->later; zf->later = nullptr; } }

// status of a split decode whose final read-back was left to the caller
int zstd_split_status(naf_gpu_ctx *c, const ZSplit *sp)
{
    if (!sp || !sp->done || !sp->status) return 0;
    ZStat hs; int rc = ctx_readback(c, &hs, sp->status, sizeof hs); if (rc) return rc;
    if (hs.err) return zerr(c, hs.err, "block decode");
    return 0;
}

int zstd_decode(naf_gpu_ctx *c, const u8 *d_src, size_t src_len, int has_magic, u8 *d_dst, size_t dst_cap, size_t *out_len, const u8 *head)
{
    return zstd_decode_range(c, d_src, src_len, has_magic, d_dst, dst_cap, out_len, nullptr, head);
}

int zstd_decode_fused_fasta(naf_gpu_ctx *c, const u8 *d_src, size_t src_len, int has_magic, size_t *out_len, const EmitP *P, u8 *text)
{
    // single frame only; returns ZSTD_NEED_TWO_PASS when the frame has blocks the fused kernel does not cover
    size_t used = 0; u8 m[4];
    size_t pos = 0;
    if (has_magic) { if (src_len < 4) return zerr(c, ZE_TRUNC, "magic"); int rc = ctx_readback(c, m, d_src, 4); if (rc) return rc; if (ld32(m) != 0xFD2FB528u) return ZSTD_NEED_TWO_PASS; pos = 4; }
    int rc = zstd_decode_one(c, d_src + pos, src_len - pos, nullptr, 0, out_len, &used, nullptr, P, text);
    if (rc) return rc;
    if (pos + used != src_len) return ZSTD_NEED_TWO_PASS;
    LAUNCH(c, "unnaf_emit_headers", k_emit_headers, cdiv(P->N, 256), 256, 0, *P, text);
    return 0;
}

// rg != nullptr: first frame only is range-decoded (sections written by ennaf are exactly one frame)
int zstd_decode_range(naf_gpu_ctx *c, const u8 *d_src, size_t src_len, int has_magic, u8 *d_dst, size_t dst_cap, size_t *out_len, ZRange *rg, const u8 *head)
{
    size_t pos = 0, out = 0; bool first = true;
    *out_len = 0;
    while (pos < src_len || first) {
        if (!(first && !has_magic)) {
            u8 m[8]; size_t ml = src_len - pos < 8 ? src_len - pos : 8;
            if (ml < 4) return zerr(c, ZE_TRUNC, "magic");
            int rc = ctx_readback(c, m, d_src + pos, ml); if (rc) return rc;
            u32 magic = ld32(m);
            if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {              // skippable frame (3.1.2)
                if (ml < 8) return zerr(c, ZE_TRUNC, "skippable frame");
                size_t sz = ld32(m + 4);
                if (pos + 8 + sz > src_len) return zerr(c, ZE_TRUNC, "skippable frame");
                pos += 8 + sz; first = false; continue;
            }
            if (magic != 0xFD2FB528u) return zerr(c, ZE_CORRUPT, "bad magic");
            pos += 4;
        }
        first = false;
        size_t n = 0, used = 0;
        int rc = zstd_decode_one(c, d_src + pos, src_len - pos, d_dst + out, dst_cap > out ? dst_cap - out : 0, &n, &used, out == 0 ? rg : nullptr, nullptr, nullptr,
                                 (head && pos == 0 && !has_magic) ? head : nullptr);
        if (rg && rg->ranged && pos + used < src_len) return ctx_fail(c, NAF_GPU_EZSTD, "range decode needs a single-frame stream");
        if (rc == NAF_GPU_ECAP) { *out_len = out + n; return rc; }
        if (rc) return rc;
        out += n; pos += used;
    }
    *out_len = out;
    return 0;
}

extern "C" int naf_gpu_zstd_decompress(naf_gpu_ctx *c, const void *d_src, size_t src_len, int has_magic,
                                       void *d_dst, size_t dst_cap, size_t *out_len)
{
    if (!c || !d_src || !out_len) return NAF_GPU_EARG;
    arena_reset(c);
    return zstd_decode(c, (const u8 *)d_src, src_len, has_magic, (u8 *)d_dst, dst_cap, out_len);
}
