// zstd_enc.hip -- block-parallel zstd frame encoder for gfx950 (replaces libzstd behind
// ennaf/src/compressor.c:7-21 create_zstd_cstream, :119-147 compress, :64-96 compressor_end_stream).
//
// One frame per stream, built from independently coded blocks:
//   k_zenc_plan   one workgroup per block: 4 quarter histograms in LDS (atomics), then one lane builds the
//                 length-limited Huffman code, the tree description and the exact compressed size
//   scan          compressed block sizes -> byte offsets inside the frame
//   k_zenc_write  16 blocks per 64-lane workgroup: code tables in LDS, ONE LANE PER HUFFMAN STREAM
//                 writes its stream at its final offset; raw/RLE blocks are copied by the whole wave
#include "ctx.h"
#include "zstd_enc_core.h"

#define ZENC_TREE_SLOT 192
#include "wgscan.h"
struct OpMaxU64 { template <typename T> __device__ static T id() { return (T)0; } template <typename T> __device__ static T f(T a, T b) { return a > b ? a : b; } };
#define ZENC_HCOPIES 4
struct ZPlanWS {
    u32 tot[256]; u32 w[512]; u16 order[256]; u16 parent[512]; u8 depth[512];
    u8 len[256], wt[256], tree[ZENC_TREE_SLOT], tmp[160];
    FseWS fse; u32 log, tb;
};

// even split of n bytes into nblk blocks: block b starts at b*(n/nblk) + min(b, n%nblk)
__host__ __device__ static inline u64 zenc_block_lo(u64 n, u32 nblk, u32 b) { u64 q = n / nblk, r = n % nblk; return (u64)b * q + (b < r ? b : r); }

__global__ __launch_bounds__(256) void k_zenc_plan(const u8 *src, u64 n, u32 nblk, ZEncPlan *plan, u32 *codes, u8 *trees, u64 *csize)
{
    // ZENC_HCOPIES copies of the 4 quarter histograms (copy = lane % copies): few distinct symbols (packed ACGT has 16) would
    // otherwise serialise every LDS atomic of a wave on the same handful of addresses
    __shared__ u32 hist[ZENC_HCOPIES * 1024];
    u32 b = blockIdx.x;
    u64 lo = zenc_block_lo(n, nblk, b), hi = zenc_block_lo(n, nblk, b + 1);
    u32 bn = (u32)(hi - lo);
    for (u32 i = threadIdx.x; i < ZENC_HCOPIES * 1024; i += 256) hist[i] = 0;
    __syncthreads();
    u32 per = (bn + 3) / 4; if (!per) per = 1;
    const u8 *s = src + lo;
    // copy c keeps symbol s in slot (s + 8c) & 255: the four copies of a symbol sit in four different LDS banks
    const u32 cpy = threadIdx.x & (ZENC_HCOPIES - 1), rot = 8 * cpy;
    u32 *my = hist + cpy * 1024;
    for (u32 i = threadIdx.x * 8; i < bn; i += 2048) {
        if (i + 8 <= bn) {
            u64 w = ld64(s + i);
#pragma unroll
            for (u32 k = 0; k < 8; k++) { u32 pos = i + k, q = (pos >= per) + (pos >= 2 * per) + (pos >= 3 * per); atomicAdd(&my[q * 256 + (((u32)(w >> (8 * k)) + rot) & 0xFF)], 1u); }
        } else {
            for (u32 k = 0; i + k < bn; k++) { u32 pos = i + k, q = (pos >= per) + (pos >= 2 * per) + (pos >= 3 * per); atomicAdd(&my[q * 256 + ((s[pos] + rot) & 0xFF)], 1u); }
        }
    }
    __syncthreads();
    u32 red4[4];
    for (u32 q = 0; q < 4; q++) { u32 v = 0; for (u32 k = 0; k < ZENC_HCOPIES; k++) v += hist[k * 1024 + q * 256 + ((threadIdx.x + 8 * k) & 0xFF)]; red4[q] = v; }
    __syncthreads();
    for (u32 q = 0; q < 4; q++) hist[q * 256 + threadIdx.x] = red4[q];
    __syncthreads();
    // ---- plan: same decisions as zenc_plan_block, with the per-symbol loops spread over the 256 threads and every
    // table of the serial steps in LDS (one lane walking private arrays in scratch memory cost 0.3 ms per block)
    // the workspace lives where histogram copies 1..3 were (dead after the reduction): 16 KiB of LDS per block instead of 23,
    // and the number of blocks resident per CU is what bounds this kernel (one lane per block runs the serial steps)
    static_assert(sizeof(ZPlanWS) <= (ZENC_HCOPIES - 1) * 4096, "plan workspace must fit the dead histogram copies");
    ZPlanWS &ws = *(ZPlanWS *)(hist + 1024);
    __shared__ u64 red[4];
    const u32 sym = threadIdx.x;
    u32 mine = hist[sym] + hist[256 + sym] + hist[512 + sym] + hist[768 + sym];
    ws.tot[sym] = mine; ws.len[sym] = 0;
    u32 distinct = (u32)__syncthreads_count(mine != 0);
    // the symbols that occur, in symbol order (packed ACGT: 16 of 256): every per-symbol loop below runs over this list
    __shared__ u16 plist[256]; __shared__ u32 wave_n[4];
    u64 bal = __ballot(mine != 0);
    u32 myidx = (u32)__popcll(bal & ((1ull << (threadIdx.x & 63)) - 1));
    if ((threadIdx.x & 63) == 0) wave_n[threadIdx.x >> 6] = (u32)__popcll(bal);
    __syncthreads();
    for (u32 w = 0; w < (threadIdx.x >> 6); w++) myidx += wave_n[w];
    if (mine) plist[myidx] = (u16)sym;
    __syncthreads();
    ZEncPlan p; p.n = bn; p.kind = ZK_RAW; p.csize = 3 + bn; p.log = 0; p.tree_bytes = 0; p.lhdr = 0; p.pad = 0;
    p.ssz[0] = p.ssz[1] = p.ssz[2] = p.ssz[3] = 0;
    bool huf = false;
    if (bn && distinct == 1) { p.kind = ZK_RLE; p.csize = 4; }
    else if (bn >= 64 && distinct >= 2) {
        // rank sort by (count, symbol): the order a stable insertion sort over ascending symbols gives
        if (mine) {
            u32 r = 0;
            for (u32 j = 0; j < distinct; j++) { u32 t = plist[j], c = ws.tot[t]; r += (c < mine || (c == mine && t < sym)); }
            ws.order[r] = (u16)sym;
        }
        __syncthreads();
        if (threadIdx.x == 0) ws.log = huf_lengths_sorted(ws.tot, ws.order, distinct, ws.len, ws.w, ws.parent, ws.depth);
        __syncthreads();
        u32 log = ws.log;
        if (log) {
            u32 l = ws.len[sym];
            ws.wt[sym] = l ? (u8)(log + 1 - l) : 0;
            u64 lastw; wg_scan_inclusive<u64, OpMaxU64>(l ? (u64)sym : 0, &lastw, red);
            if (threadIdx.x == 0) ws.tb = huf_write_tree_w(ws.tree, ws.wt, (u32)lastw, ws.tmp, ws.fse);
            u64 bits;
            for (u32 k = 0; k < 4; k++) { wg_scan_inclusive<u64, OpAdd>((u64)hist[256 * k + sym] * l, &bits, red); p.ssz[k] = (u32)((bits + 1 + 7) / 8); }
            __syncthreads();
            u32 tb = ws.tb;
            if (tb) { zenc_plan_finish(p, bn, log, tb); huf = p.kind == ZK_HUF; }
        }
    }
    if (threadIdx.x == 0) { plan[b] = p; csize[b] = p.csize; }
    if (huf) {
        // canonical codes (huf_assign_codes): symbols of one weight take consecutive cells in symbol order, so the code of a
        // symbol is its weight group's first cell >> (weight - 1) plus its rank inside the group
        __shared__ u32 wcnt[16], wstart[16];
        if (sym < 16) wcnt[sym] = 0;
        __syncthreads();
        u32 l = ws.len[sym], wgt = l ? p.log + 1 - l : 0, rank = 0;
        if (l) { atomicAdd(&wcnt[wgt], 1u); for (u32 j = 0; j < myidx; j++) rank += ws.len[plist[j]] == l; }
        __syncthreads();
        if (sym == 0) { u32 pos = 0; for (u32 w = 1; w <= p.log; w++) { wstart[w] = pos; pos += wcnt[w] << (w - 1); } }
        __syncthreads();
        codes[(u64)b * 256 + sym] = l ? (((wstart[wgt] >> (wgt - 1)) + rank) | (l << 16)) : 0;
        if (sym < p.tree_bytes) trees[(u64)b * ZENC_TREE_SLOT + sym] = ws.tree[sym];
    }
}

#define ZENC_BLOCKS_PER_WG 16
#define ZENC_OROW 80                      // LDS output row per lane: one 64-byte segment + the word that spills over it + pad
// One Huffman stream (4.2.2: written forward so that the LAST symbol is read first) by one lane.  Input is pulled 64 bytes
// at a time into registers (walking down), output is collected in the lane's LDS row and leaves as aligned 64-byte
// segments: per-lane 8-byte loads re-fetch every line 4x and per-lane 8-byte stores cost 6x the bytes at the HBM.
__device__ __forceinline__ void huf_encode_stream_staged(u8 *out, const u8 *src, u32 n, const u32 *codes, u8 *orow)
{
    u64 acc = 0; u32 nb = 0; u8 *p = out;
    u32 fill = 0; bool staged = false;                            // staged: p is 64-byte aligned and row[0..fill) holds the bytes at p
    auto emit_word = [&](u64 w) {
        if (!staged) {
            st64(p, w); p += 8;
            u32 mis = (u32)((uintptr_t)p & 63);
            if (mis < 8) {                                        // crossed into a new segment: its first `mis` bytes go to the row
                p -= mis; staged = true; fill = mis;
                if (mis) { u64 t = w >> (8 * (8 - mis)); __builtin_memcpy(orow, &t, 8); }
            }
            return;
        }
        __builtin_memcpy(orow + fill, &w, 8); fill += 8;
        if (fill >= 64) {
            const uint4 *r = (const uint4 *)orow; uint4 *g = (uint4 *)p;
            g[0] = r[0]; g[1] = r[1]; g[2] = r[2]; g[3] = r[3];
            u64 t; __builtin_memcpy(&t, orow + 64, 8); __builtin_memcpy(orow, &t, 8);
            p += 64; fill -= 64;
        }
    };
    auto put = [&](u32 sym) {
        u32 e = codes[sym]; u32 len = e >> 16; u64 v = (u64)(e & 0xFFFF);
        acc |= v << nb;                                            // nb < 64 here; bits that do not fit are re-added after the word leaves
        if (nb + len >= 64) { emit_word(acc); acc = nb ? (v >> (64 - nb)) : 0; nb = nb + len - 64; }
        else nb += len;
    };
    u32 i = n;
    while (i >= 64) {                                             // symbols i-1 .. i-64, highest index first
        i -= 64;
        uint4 q[4]; __builtin_memcpy(q, src + i, 64);
#pragma unroll
        for (int wdx = 7; wdx >= 0; wdx--) {
            const uint4 &qq = q[wdx >> 1];
            u64 w = (wdx & 1) ? ((u64)qq.z | ((u64)qq.w << 32)) : ((u64)qq.x | ((u64)qq.y << 32));
#pragma unroll
            for (int k = 7; k >= 0; k--) put((u32)(w >> (8 * k)) & 0xFF);
        }
    }
    while (i-- > 0) put(src[i]);
    acc |= 1ull << nb; nb++;                                      // final marker bit (nb <= 63 before, so it fits)
    // tail: what is in the row, then the last bits, byte-wise
    if (staged) { for (u32 k = 0; k < fill; k++) p[k] = orow[k]; p += fill; }
    while (nb > 0) { *p++ = (u8)acc; acc >>= 8; nb = nb > 8 ? nb - 8 : 0; }
}

__global__ __launch_bounds__(64) void k_zenc_write(const u8 *src, u64 n, u32 nblk, const ZEncPlan *plan, const u32 *codes_g, const u8 *trees,
                                                    const u64 *offs, u8 *dst, u64 frame_hdr)
{
    __shared__ __attribute__((aligned(16))) u32 codes[ZENC_BLOCKS_PER_WG][256];
    __shared__ __attribute__((aligned(16))) u8 orows[64 * ZENC_OROW];
    int lane = threadIdx.x;
    u32 b0 = blockIdx.x * ZENC_BLOCKS_PER_WG;
    for (u32 jj = 0; jj < ZENC_BLOCKS_PER_WG; jj++) {               // code tables made by k_zenc_plan: 1 KiB per block, coalesced
        u32 bb = b0 + jj;
        if (bb >= nblk) break;
        if (plan[bb].kind != ZK_HUF) continue;
        const uint4 *g = (const uint4 *)(codes_g + (u64)bb * 256);
        ((uint4 *)codes[jj])[lane] = g[lane];
    }
    __syncthreads();
    u32 j = lane >> 2, k = lane & 3, b = b0 + j;
    if (b < nblk) {
        const ZEncPlan p = plan[b];
        u64 lo = zenc_block_lo(n, nblk, b);
        u8 *out = dst + frame_hdr + offs[b];
        if (k == 0) zenc_write_block_prefix(out, p, trees + (u64)b * ZENC_TREE_SLOT, b + 1 == nblk, p.n ? src[lo] : 0);
        if (p.kind == ZK_HUF) {
            u32 per = (p.n + 3) / 4;
            u32 cnt = k < 3 ? per : p.n - 3 * per;
            u32 o = 3 + p.lhdr + p.tree_bytes + 6;
            for (u32 q = 0; q < k; q++) o += p.ssz[q];
            huf_encode_stream_staged(out + o, src + lo + (u64)k * per, cnt, codes[j], orows + lane * ZENC_OROW);
            if (k == 3) out[p.csize - 1] = 0;                       // Number_of_Sequences = 0
        }
    }
    // raw blocks: whole-wave copy
    for (u32 jj = 0; jj < ZENC_BLOCKS_PER_WG; jj++) {
        u32 bb = b0 + jj;
        if (bb >= nblk) break;
        if (plan[bb].kind != ZK_RAW) continue;
        u64 lo = zenc_block_lo(n, nblk, bb);
        u32 bn = plan[bb].n;
        const u8 *s = src + lo; u8 *o = dst + frame_hdr + offs[bb] + 3;
        for (u32 i = lane * 8; i + 8 <= bn; i += 64 * 8) st64(o + i, ld64(s + i));
        for (u32 i = (bn & ~7u) + lane; i < bn; i += 64) o[i] = s[i];
    }
}

__global__ void k_zenc_frame_header(u8 *dst, int with_magic)
{
    if (threadIdx.x || blockIdx.x) return;
    u32 p = 0;
    if (with_magic) { dst[p++] = 0x28; dst[p++] = 0xB5; dst[p++] = 0x2F; dst[p++] = 0xFD; }
    dst[p++] = 0x00;            // Frame_Header_Descriptor: no FCS, no checksum, no dictionary, windowed (like ennaf -1)
    dst[p++] = 0x48;            // Window_Descriptor: 2^19 (blocks here never reference earlier data)
}

extern "C" size_t naf_gpu_zstd_compress_bound(size_t n)
{
    return n + 3 * (n / 4096 + 2) + 64;
}

// block_log: log2 of the target block size (clamped to 17 = the format maximum of 128 KiB)
int zstd_encode(naf_gpu_ctx *c, const u8 *d_src, size_t n, int level, u8 *d_dst, size_t cap, size_t *out_len, int with_magic)
{
    u32 block_log = 15;                                          // 32 KiB: 4 streams of 8 KiB; more streams = more decode parallelism
    const char *e = getenv("NAF_GPU_BLOCK_LOG");
    if (e) { int v = atoi(e); if (v >= 10 && v <= 17) block_log = (u32)v; }
    (void)level;
    u64 bs = 1ull << block_log;
    u64 nblk64 = n ? (n + bs - 1) / bs : 1;
    if (nblk64 > 0x7FFFFFFFull) return ctx_fail(c, NAF_GPU_EARG, "stream too large");
    u32 nblk = (u32)nblk64;
    u64 hdr = with_magic ? 6 : 2;
    if (cap < hdr + n + 3ull * nblk) return ctx_fail(c, NAF_GPU_ECAP, "zstd_compress capacity %zu too small (bound %llu)", cap, (unsigned long long)(hdr + n + 3ull * nblk));
    ZEncPlan *plan = arena_new<ZEncPlan>(c, nblk);
    u32 *codes = arena_new<u32>(c, (size_t)nblk * 256); u8 *trees = (u8 *)arena_alloc(c, (size_t)nblk * ZENC_TREE_SLOT);
    u64 *offs = arena_new<u64>(c, (size_t)nblk + 2);
    if (!plan || !codes || !trees || !offs) return NAF_GPU_ENOMEM;
    LAUNCH(c, "zenc_plan", k_zenc_plan, nblk, 256, 0, d_src, (u64)n, nblk, plan, codes, trees, offs);
    int rc = scan_exclusive_u64(c, offs, nblk, offs + nblk + 1); if (rc) return rc;
    LAUNCH(c, "zenc_frame_header", k_zenc_frame_header, 1, 64, 0, d_dst, with_magic);
    LAUNCH(c, "zenc_write", k_zenc_write, cdiv(nblk, ZENC_BLOCKS_PER_WG), 64, 0, d_src, (u64)n, nblk, (const ZEncPlan *)plan, (const u32 *)codes, (const u8 *)trees,
           (const u64 *)offs, d_dst, hdr);
    u64 total = 0;
    rc = ctx_readback(c, &total, offs + nblk + 1, 8); if (rc) return rc;
    *out_len = hdr + total;
    return 0;
}

extern "C" int naf_gpu_zstd_compress(naf_gpu_ctx *c, const void *d_src, size_t n, int level, void *d_dst, size_t cap, size_t *out_len)
{
    if (!c || !d_dst || !out_len || (!d_src && n)) return NAF_GPU_EARG;
    arena_reset(c);
    return zstd_encode(c, (const u8 *)d_src, n, level, (u8 *)d_dst, cap, out_len, 1);
}
