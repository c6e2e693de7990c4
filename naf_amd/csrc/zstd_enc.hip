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
#include <vector>
#include "zstd_enc_core.h"

#define ZENC_TREE_SLOT 192
#include "wgscan.h"
#include "enc_swar.h"
struct OpMaxU64 { template <typename T> __device__ static T id() { return (T)0; } template <typename T> __device__ static T f(T a, T b) { return a > b ? a : b; } };
#define ZENC_HCOPIES 4
// Huffman tree descriptions of a sample of the blocks, looked up by weight table (one per zstd_encode call, zeroed by the host).
#define ZENC_CACHE_ENTRIES 256
struct ZTreeEntry { u64 key; u32 tb, pad; u8 wt[256]; u8 tree[ZENC_TREE_SLOT]; };   // key 0: empty
struct ZTreeCache { ZTreeEntry e[ZENC_CACHE_ENTRIES]; };
static_assert(ZENC_CACHE_ENTRIES == 256, "one entry per thread of the plan kernel");
struct ZPlanWS {
    u32 tot[256]; u32 w[512]; u16 order[256]; u16 parent[512]; u8 depth[512];
    u8 len[256], wt[256], tree[ZENC_TREE_SLOT], tmp[160];
    FseWS fse; u32 log, tb;
};

// Tree description of the flat tree of the sixteen pair codes (weight 1 for each byte whose two nibbles are single bits): the same for
// every block k_zenc_flat_scan settles, made once on the host with the kernels' own tree writer.
struct ZFlat16 { u32 tb; u8 tree[ZENC_TREE_SLOT]; };
static const ZFlat16 &zenc_flat16()
{
    static const ZFlat16 T = [] {
        ZFlat16 t; memset(&t, 0, sizeof t);
        u8 wt[256]; memset(wt, 0, sizeof wt);
        for (u32 s = 0; s < 256; s++) if (__builtin_popcount(s & 15u) == 1 && __builtin_popcount(s >> 4) == 1) wt[s] = 1;
        static FseWS ws; u8 tmp[160];
        t.tb = huf_write_tree_w(t.tree, wt, 0x88u, tmp, ws, true);       // weights of symbols 0 .. 0x87; that of 0x88, the last, is implied
        return t;
    }();
    return T;
}
// even split of n bytes into nblk blocks: block b starts at b*(n/nblk) + min(b, n%nblk)
__host__ __device__ static inline u64 zenc_block_lo(u64 n, u32 nblk, u32 b) { u64 q = n / nblk, r = n % nblk; return (u64)b * q + (b < r ? b : r); }

// Flat preference, the quick way (ZENC_PREFER_FLAT: the packed 4-bit stream), one WAVE per block and no histogram: when EVERY byte of
// the block is one of the sixteen pair codes of A C G T (both nibbles a single bit: a nibble-wise population count over eight bytes at
// a time) and the entropy of an eighth of the block (the first half of every fourth 16-byte piece: 4 K symbols of a 32 KiB block) is
// above 1 - 1/prefer_flat of four bits per symbol -- Huffman coding cannot save the threshold then -- everything about the block is known
// without k_zenc_plan: sixteen 4-bit codes in symbol order, four streams of n x 4 bits + the end mark, the tree description f16.  (Codes
// that happen not to occur still get their place in the tree: every such block of a genome then carries the SAME description, which is
// what the decoder's in-place reader wants.)  done[b] = 1 for the blocks settled here; k_zenc_plan leaves them alone.
// A lane's loads are independent and all in flight together; with the test inside k_zenc_plan (a 256-thread block per 32 KiB, its LDS
// workspace cleared first, a dozen barriers) 152 K blocks took 1.6 - 2.0 ms.
#define ZENC_FLATSCAN_WAVES 4
// direct[b] != 0: the block is known to be such a block and its four streams are already coded (enc.hip: direct_word; plan.pad = 2) -- nothing is read.
__global__ __launch_bounds__(64 * ZENC_FLATSCAN_WAVES) void k_zenc_flat_scan(const u8 *src, u64 n, u32 nblk, ZEncPlan *plan, u16 *codes, u8 *trees, u64 *csize, u8 *done, u32 min_gain, u32 prefer_flat, ZFlat16 f16, const u8 *planned, const u8 *direct)
{
    __shared__ u32 bins[ZENC_FLATSCAN_WAVES][64];                 // 4 copies x 16 bins per wave: copy = (lane >> 2) & 3
    const u32 wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // (a bounded grid walks the blocks four at a time: a workgroup per four blocks that only finds them settled was 2 ms of dispatch per 100 GB)
    for (u32 bq = blockIdx.x; bq * ZENC_FLATSCAN_WAVES < nblk; bq += gridDim.x) {
    const u32 b = bq * ZENC_FLATSCAN_WAVES + wv;
    // four direct blocks (nearly every four of a genome's frame): settled by k_zenc_direct_plans, a thread each
    if (planned && bq * ZENC_FLATSCAN_WAVES + ZENC_FLATSCAN_WAVES <= nblk && *(const u32 *)(planned + bq * ZENC_FLATSCAN_WAVES) == 0x01010101u) continue;
    bins[wv][lane] = 0;
    __syncthreads();
    bool ok = b < nblk;
    u32 bn = 0; const u8 *s = src;
    if (ok) {
        const u64 lo = zenc_block_lo(n, nblk, b), hi = zenc_block_lo(n, nblk, b + 1);
        bn = (u32)(hi - lo); s = src + lo;
        ok = bn >= 2048;
    }
    const bool dir = ok && direct && direct[b];
    if (ok && !dir) {
        u64 bad = 0;
        const bool sampler = (lane & 3) == 0;
        u32 *mybins = &bins[wv][((lane >> 2) & 3) * 16];
        const u64 M5 = 0x5555555555555555ull, M3 = 0x3333333333333333ull, ONE = 0x1111111111111111ull;
#pragma unroll 8
        for (u32 i = lane * 16; i + 16 <= bn; i += 1024) {
            const u64 w0 = ld64(s + i), w1 = ld64(s + i + 8);
            u64 c0 = w0 - ((w0 >> 1) & M5), c1 = w1 - ((w1 >> 1) & M5);
            c0 = (c0 & M3) + ((c0 >> 2) & M3); c1 = (c1 & M3) + ((c1 >> 2) & M3);
            bad |= (c0 ^ ONE) | (c1 ^ ONE);
            if (sampler) {
                // rank of a pair code: 4 log2(high nibble) + log2(low nibble); log2 of a one-hot nibble x is (x >> 1) - (x >> 3)
                const u64 lg = ((w0 >> 1) & 0x7777777777777777ull) - ((w0 >> 3) & 0x1111111111111111ull);   // per nibble, no borrows for one-hot nibbles
                const u64 rk = ((lg >> 2) & 0x0C0C0C0C0C0C0C0Cull) | (lg & 0x0303030303030303ull);
#pragma unroll
                for (u32 k = 0; k < 8; k++) atomicAdd(&mybins[(u32)(rk >> (8 * k)) & 15u], 1u);
            }
        }
        if (lane == 0) for (u32 k = bn & ~15u; k < bn; k++) { const u32 v = s[k]; if (__popc(v & 15u) != 1 || __popc(v >> 4) != 1) bad = 1; }
        ok = __ballot(bad != 0) == 0;
    }
    __syncthreads();                                             // (every wave gets here: the sample counts are complete)
    bool fast = dir;
    if (ok && !dir) {
        const u32 c = lane < 16 ? bins[wv][lane] + bins[wv][16 + lane] + bins[wv][32 + lane] + bins[wv][48 + lane] : 0u;
        u32 ns = c;
#pragma unroll
        for (u32 d = 1; d < 16; d <<= 1) ns += __shfl_xor((int)ns, d, 64);
        float h = c ? (float)c * __log2f((float)ns / (float)c) : 0.0f;        // entropy of the sample, bits
#pragma unroll
        for (u32 d = 1; d < 16; d <<= 1) h += __shfl_xor(h, d, 64);
        ns = (u32)__shfl((int)ns, 0, 64); h = __shfl(h, 0, 64);
        fast = ns > 0 && h * (float)prefer_flat > 4.0f * (float)ns * (float)(prefer_flat - 1);
    }
    if (fast) {
        const u32 perq = (bn + 3) / 4;
        ZEncPlan p; p.n = bn; p.kind = ZK_RAW; p.csize = 3 + bn; p.log = 0; p.tree_bytes = 0; p.lhdr = 0; p.pad = 0; p.frame = 0; p.pad2 = 0;
        for (u32 k = 0; k < 4; k++) p.ssz[k] = ((k < 3 ? perq : bn - 3 * perq) * 4 + 8) >> 3;
        zenc_plan_finish(p, bn, 4, f16.tb, min_gain);
        fast = p.kind == ZK_HUF;
        if (fast) {
            p.pad = dir ? 2 : 1;                                  // (k_zenc_write packs such a block with all lanes; 2: copies its ready streams)
            if (lane == 0) { plan[b] = p; if (csize) csize[b] = p.csize; }
            if (!dir)                                              // (nobody encodes a direct block's bytes)
            for (u32 sym = lane; sym < 256; sym += 64) {
                const bool is16 = __popc(sym & 15u) == 1 && __popc(sym >> 4) == 1;
                const u32 hn = sym >> 4, ln = sym & 15u, rank = 4 * ((hn >> 1) - (hn >> 3)) + ((ln >> 1) - (ln >> 3));
                codes[(u64)b * 256 + sym] = is16 ? (u16)(rank | (4u << 12)) : (u16)0;
            }
            for (u32 k = lane; k < p.tree_bytes; k += 64) trees[(u64)b * ZENC_TREE_SLOT + k] = f16.tree[k];
        }
    }
    if (b < nblk && lane == 0) done[b] = fast ? 1 : 0;
    }
}

// the plan of a direct block is the same for all of them (32 KiB, the tree f16, four streams of 4096 + 1 bytes): a thread per block
__global__ void k_zenc_direct_plans(const u8 *direct, u32 nblk, ZEncPlan pd, ZFlat16 f16, ZEncPlan *plan, u8 *trees, u64 *csize, u8 *done)
{
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk || !direct[b]) return;
    plan[b] = pd; if (csize) csize[b] = pd.csize; done[b] = 1;
    for (u32 k = 0; k < pd.tree_bytes; k++) trees[(u64)b * ZENC_TREE_SLOT + k] = f16.tree[k];
}

// blk_len == nullptr: block b is the b-th piece of the even split of src[0..n).  Otherwise block b is src[b*slot .. b*slot + blk_len[b])
// (the literals the LZ stage left of block b).
__global__ __launch_bounds__(256) void k_zenc_plan(const u8 *src, u64 n, u32 nblk, ZEncPlan *plan, u16 *codes, u8 *trees, u64 *csize, const u32 *blk_len, u64 slot, ZTreeCache *cache, u32 sample_stride, u32 try_fse, u32 min_gain, u32 maxbits, u32 prefer_flat, const u8 *done, u8 *wt_defer,
                                                    u32 *fhist = nullptr, const ZEncPlan *fplan = nullptr, const u16 *fcodes = nullptr, const u32 *fratio = nullptr)
{
    // ZENC_HCOPIES copies of the 4 quarter histograms (copy = lane % copies): few distinct symbols (packed ACGT has 16) would
    // otherwise serialise every LDS atomic of a wave on the same handful of addresses
    __shared__ u32 hist[ZENC_HCOPIES * 1024];
    u32 b = sample_stride ? blockIdx.x * sample_stride : blockIdx.x;
    if (done && done[b]) return;                                 // settled by k_zenc_flat_scan
    u64 lo = zenc_block_lo(n, nblk, b), hi = zenc_block_lo(n, nblk, b + 1);
    if (blk_len) { lo = (u64)b * slot; hi = lo + blk_len[b]; }
    u32 bn = (u32)(hi - lo);
    for (u32 i = threadIdx.x; i < ZENC_HCOPIES * 1024; i += 256) hist[i] = 0;
    __syncthreads();
    u32 per = (bn + 3) / 4; if (!per) per = 1;
    const u8 *s = src + lo;
    // copy c keeps symbol s in slot (s + 8c) & 255: the four copies of a symbol sit in four different LDS banks
    const u32 cpy = threadIdx.x & (ZENC_HCOPIES - 1), rot = 8 * cpy;
    u32 *my = hist + cpy * 1024;
    const u64 rot8 = 0x0101010101010101ull * rot;                 // rot < 128: added to every byte at once, carries cut at the byte tops
#ifdef ZPLAN_NOHIST
    if (threadIdx.x < 16) for (u32 q = 0; q < 4; q++) hist[q * 256 + threadIdx.x * 17] = bn / 64;
    if (0)
#endif
    // eight words of a thread in flight together (with one load per trip, a block of 32 KiB was sixteen round trips to memory in a row:
    // 70 of the planner's 420 us per block of a quality stream)
    for (u32 i0 = threadIdx.x * 8; i0 < bn; i0 += 8 * 2048) {
        u64 wv[8];
#pragma unroll
        for (u32 j = 0; j < 8; j++) { const u32 i = i0 + j * 2048; wv[j] = i + 8 <= bn ? ld64(s + i) : 0; }
#pragma unroll
        for (u32 j = 0; j < 8; j++) {
            const u32 i = i0 + j * 2048;
            if (i >= bn) break;
            if (i + 8 <= bn) {
                const u64 w = wv[j];
                const u32 q0 = (i >= per) + (i >= 2 * per) + (i >= 3 * per), q7 = (i + 7 >= per) + (i + 7 >= 2 * per) + (i + 7 >= 3 * per);
                if (q0 == q7) {                                       // the word lies in one quarter (all but three words of a block)
                    const u64 H = 0x8080808080808080ull;
                    const u64 wr = ((w & ~H) + rot8) ^ (w & H);      // (byte + rot) & 0xFF for all eight bytes
                    u32 *hq = my + q0 * 256;
                    const u32 lo = (u32)wr, hi = (u32)(wr >> 32);
#pragma unroll
                    for (u32 k = 0; k < 4; k++) { atomicAdd(&hq[(lo >> (8 * k)) & 0xFF], 1u); atomicAdd(&hq[(hi >> (8 * k)) & 0xFF], 1u); }
                } else {
#pragma unroll
                    for (u32 k = 0; k < 8; k++) { u32 pos = i + k, q = (pos >= per) + (pos >= 2 * per) + (pos >= 3 * per); atomicAdd(&my[q * 256 + (((u32)(w >> (8 * k)) + rot) & 0xFF)], 1u); }
                }
            } else {
                for (u32 k = 0; i + k < bn; k++) { u32 pos = i + k, q = (pos >= per) + (pos >= 2 * per) + (pos >= 3 * per); atomicAdd(&my[q * 256 + ((s[pos] + rot) & 0xFF)], 1u); }
            }
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
    ZEncPlan p; p.n = bn; p.kind = ZK_RAW; p.csize = 3 + bn; p.log = 0; p.tree_bytes = 0; p.lhdr = 0; p.pad = 0; p.frame = 0; p.pad2 = 0;
    p.ssz[0] = p.ssz[1] = p.ssz[2] = p.ssz[3] = 0;
    // ---- the FRAME's tree (zstd_encode_begin: ZENC_FRAME_TREE).  The sample launch leaves the sum of its blocks' histograms; one code is
    // made of it, and a block whose symbols that code covers, at a cost within 3 % (+ 8 bytes) of what such a code makes of the block's own entropy (k_zenc_frame_ratio), is coded with
    // it: no code construction, no tree description (k_zenc_frame_fix decides which of these blocks carry the tree: the first one, and
    // those behind a block with a tree of its own) -- the serial steps of a plan are what the planner's time is, and the decoder meets
    // one tree instead of one per block.
    if (fhist && sample_stride && mine) atomicAdd(&fhist[sym], mine);
    if (fcodes && !sample_stride && fplan->kind == ZK_HUF && bn >= 64 && distinct >= 2) {
        __shared__ u64 fred[4];
        const u32 fl = (u32)fcodes[sym] >> 12;
        if (__syncthreads_and(mine == 0 || fl != 0)) {
            const float ent = mine ? (float)mine * __log2f((float)bn / (float)mine) : 0.f;
            u64 both; wg_scan_inclusive<u64, OpAdd>(((u64)(mine * fl) << 32) | (u64)(u32)(ent + 0.5f), &both, fred);
            const u64 fbits = both >> 32, ebits = both & 0xFFFFFFFFull;
            if (fbits * 1024 * 100 <= ebits * fratio[0] * 103 + 6400ull * 1024) {
                for (u32 k = 0; k < 4; k++) { u64 bits; wg_scan_inclusive<u64, OpAdd>((u64)hist[256 * k + sym] * fl, &bits, fred); p.ssz[k] = (u32)((bits + 1 + 7) / 8); }
                zenc_plan_finish(p, bn, fplan->log, 0, min_gain);
                // (only when the block would stay below its Raw size even if k_zenc_frame_fix makes it carry the tree: a forced block is
                // Huffman "whatever it costs", and naf_gpu_zstd_compress_bound -- and Block_Maximum_Size at 128 KiB blocks -- count on
                // no block being larger than Raw.  A block that close to incompressible goes the general way, Raw fallback included.)
                if (p.kind == ZK_HUF && p.csize + fplan->tree_bytes <= 3 + bn) {
                    p.frame = 1; codes[(u64)b * 256 + sym] = fcodes[sym];
                    if (threadIdx.x == 0) { plan[b] = p; if (csize) csize[b] = p.csize; }
                    return;
                }
                if (p.kind != ZK_HUF) {
                    if (threadIdx.x == 0) { plan[b] = p; if (csize) csize[b] = p.csize; }
                    return;
                }
                p.kind = ZK_RAW; p.csize = 3 + bn; p.log = 0; p.tree_bytes = 0; p.lhdr = 0; p.pad = 0; p.frame = 0; p.pad2 = 0;
                p.ssz[0] = p.ssz[1] = p.ssz[2] = p.ssz[3] = 0;
            }
        }
    }
    bool huf = false, defer = false;
    if (bn && distinct == 1) { p.kind = ZK_RLE; p.csize = 4; }
    // A mask block that is one unit repeated but for a few bytes (the end of an upper-case genome's mask: 0xFF units, then the rest of the
    // run) stays Raw: the frame is then RLE and Raw blocks only, which this build's unnaf turns into toggles in one launch
    // (emit.hip: k_mask_rle_frame) instead of taking it through the whole decoder -- for at most a block's bytes per stream.
    else if (min_gain && bn >= 64 && __syncthreads_or(mine + 16 >= bn)) { }
    else if (bn >= 64 && distinct >= 2) {
        // 2^k symbols whose two rarest together outweigh the commonest one (packed random bases: sixteen near-equal counts): every
        // merge of the two-queue construction pairs leaves before it touches a node, level after level -- a balanced tree, every code
        // k bits, whatever the order of equal counts.  No sort, no construction.
        __shared__ u32 s_flat_log, s_mm[4][3];
        {
            // the two smallest counts (equal ones count twice) and the largest: per wavefront by shuffles, across the four through LDS
            const u32 v = mine ? mine : 0xFFFFFFFFu;
            u32 m1 = v, mx = mine;
            for (int d = 32; d; d >>= 1) { const u32 o = (u32)__shfl_xor((int)m1, d, 64); m1 = o < m1 ? o : m1; const u32 q = (u32)__shfl_xor((int)mx, d, 64); mx = q > mx ? q : mx; }
            const bool twice = __popcll(__ballot(v == m1)) >= 2;
            u32 m2 = (v == m1) ? 0xFFFFFFFFu : v;
            for (int d = 32; d; d >>= 1) { const u32 o = (u32)__shfl_xor((int)m2, d, 64); m2 = o < m2 ? o : m2; }
            if (twice) m2 = m1;
            if ((threadIdx.x & 63) == 0) { s_mm[threadIdx.x >> 6][0] = m1; s_mm[threadIdx.x >> 6][1] = m2; s_mm[threadIdx.x >> 6][2] = mx; }
            __syncthreads();
            if (threadIdx.x == 0) {
                u32 a = 0xFFFFFFFFu, b = 0xFFFFFFFFu, top = 0;                           // a <= b: the two smallest of the eight candidates
                for (int w = 0; w < 4; w++) {
                    for (int k = 0; k < 2; k++) { const u32 x = s_mm[w][k]; if (x < a) { b = a; a = x; } else if (x < b) b = x; }
                    if (s_mm[w][2] > top) top = s_mm[w][2];
                }
                const bool pow2 = (distinct & (distinct - 1)) == 0;
                s_flat_log = (pow2 && distinct >= 2 && b != 0xFFFFFFFFu && (u64)a + b > top) ? (u32)(31 - __clz((int)distinct)) : 0u;
            }
            __syncthreads();
        }
        u32 flat_log = s_flat_log;
        if (flat_log) {
            if (mine) ws.len[sym] = (u8)flat_log;
            if (threadIdx.x == 0) ws.log = flat_log;
        }
        // rank sort by (count, symbol): the order a stable insertion sort over ascending symbols gives
        else if (mine) {
            u32 r = 0;
            for (u32 j = 0; j < distinct; j++) { u32 t = plist[j], c = ws.tot[t]; r += (c < mine || (c == mine && t < sym)); }
            ws.order[r] = (u16)sym;
        }
        __syncthreads();
        // Code lengths.  Up to 64 symbols (packed bases: 16, qualities: ~40): the two-queue construction of huf_lengths_sorted by
        // the first wavefront with the node weights, parents and depths in registers (lane = node index, read and written with
        // v_readlane / compare-and-select under uniform control flow) -- the same picks in the same order, without an LDS round
        // trip per step.  More symbols, or a depth above the limit: the serial routine (its length limiting is rarely needed).
        if (flat_log) { }
        else if (distinct <= 64) {
            if (threadIdx.x < 64) {
                const u32 lane = threadIdx.x, n = distinct;
                u32 wl = lane < n ? ws.tot[ws.order[lane]] : 0xFFFFFFFFu;      // leaf weights, ascending
                u32 wi = 0xFFFFFFFFu, pl = 0, pi = 0;                          // internal node weights / parents (index of an internal node)
                u32 leaf = 0, inode = 0, next = 0;                              // uniform: heads of the two queues, internal nodes made so far
                for (u32 step = 0; step + 1 < n; step++) {
                    u32 sum = 0;
#pragma unroll
                    for (int pick = 0; pick < 2; pick++) {
                        const u32 lw = leaf < n ? (u32)__builtin_amdgcn_readlane((int)wl, (int)__builtin_amdgcn_readfirstlane((int)(leaf < 64 ? leaf : 63))) : 0xFFFFFFFFu;
                        const u32 iw = inode < next ? (u32)__builtin_amdgcn_readlane((int)wi, (int)__builtin_amdgcn_readfirstlane((int)inode)) : 0xFFFFFFFFu;
                        if (leaf < n && (inode >= next || lw <= iw)) { sum += lw; if (lane == leaf) pl = next; leaf++; }
                        else { sum += iw; if (lane == inode) pi = next; inode++; }
                    }
                    if (lane == next) wi = sum;
                    next++;
                }
                // depths: the root is internal node n-2; walk the internal nodes down, then every leaf takes its parent's + 1
                u32 di = 0;
                for (u32 k = n - 2; k-- > 0;) {
                    const u32 par = (u32)__builtin_amdgcn_readlane((int)pi, (int)__builtin_amdgcn_readfirstlane((int)k));
                    const u32 dpar = (u32)__builtin_amdgcn_readlane((int)di, (int)__builtin_amdgcn_readfirstlane((int)par));
                    if (lane == k) di = dpar + 1;
                }
                const u32 dl = (u32)__shfl((int)di, (int)(pl & 63), 64) + 1;
                u32 mx = lane < n ? dl : 0;
                for (int d = 32; d; d >>= 1) { u32 o = (u32)__shfl_xor((int)mx, d, 64); mx = o > mx ? o : mx; }
                if (mx <= maxbits) { if (lane < n) ws.len[ws.order[lane]] = (u8)dl; if (lane == 0) ws.log = mx; }
                else {
                    // Length limiting, huf_lengths_sorted's steps on the lengths in the lanes (lane = rank, rarest first): clamp, then lengthen
                    // the rarest symbols that can still grow until the Kraft sum (units of 2^-maxbits) is no more than 1, then shorten the
                    // most frequent ones that fit until it is 1.  The walks are serial in the sum, so they run as uniform loops over the
                    // ranks with the candidate's length read from its lane (a quality block's 40 symbols: a few hundred scalar steps
                    // instead of one lane redoing the construction in LDS -- 60 of the planner's 420 us per block).
                    u32 l = lane < n ? (dl > maxbits ? maxbits : dl) : 0;
                    i32 K = 0;
                    { u32 k = lane < n ? 1u << (maxbits - l) : 0u; for (int d = 32; d; d >>= 1) k += (u32)__shfl_xor((int)k, d, 64); K = (i32)k; }
                    const i32 full = 1 << maxbits;
                    while (K > full) {
                        for (u32 i = 0; i < n && K > full; i++) {
                            const u32 li = (u32)__builtin_amdgcn_readlane((int)l, (int)__builtin_amdgcn_readfirstlane((int)i));
                            if (li < maxbits) { K -= 1 << (maxbits - li - 1); if (lane == i) l++; }
                        }
                    }
                    while (K < full) {
                        bool any = false;
                        for (u32 i = n; i-- > 0 && K < full;) {
                            const u32 li = (u32)__builtin_amdgcn_readlane((int)l, (int)__builtin_amdgcn_readfirstlane((int)i));
                            if (li > 1 && K + (1 << (maxbits - li)) <= full) { K += 1 << (maxbits - li); if (lane == i) l--; any = true; }
                        }
                        if (!any) break;
                    }
                    u32 ml = l;
                    for (int d = 32; d; d >>= 1) { u32 o = (u32)__shfl_xor((int)ml, d, 64); ml = o > ml ? o : ml; }
                    if (lane < n) ws.len[ws.order[lane]] = (u8)l;
                    if (lane == 0) ws.log = K == full ? ml : 0u;
                }
            }
        } else if (threadIdx.x == 0) ws.log = huf_lengths_sorted(ws.tot, ws.order, distinct, ws.len, ws.w, ws.parent, ws.depth, maxbits < 9 ? 9u : maxbits);   // (more than 64 symbols: 9 bits hold any alphabet)
        __syncthreads();
        u32 log = ws.log;
        // Flat preference (prefer_flat = d: on when Huffman coding would save less than 1/d of the block).  A block of 2^k distinct
        // symbols -- the sixteen pairs of A C G T of a real genome's packed stream -- coded with k bits per symbol can be read in place
        // by this build's decoder (emit.hip: k_emit_tile_flat) and written by all lanes here (zenc_flat4_stream); the few per cent
        // that its skewed pair histogram would save cost the decoder a pass over the packed stream.
        if (prefer_flat && log && !flat_log && (distinct & (distinct - 1)) == 0) {
            const u32 k = (u32)(31 - __clz((int)distinct));
            u64 hb; wg_scan_inclusive<u64, OpAdd>((u64)mine * ws.len[sym], &hb, red);
            const u64 fb = (u64)bn * k;
            if (fb >= hb && (fb - hb) * prefer_flat < fb) {
                ws.len[sym] = mine ? (u8)k : (u8)0;
                if (threadIdx.x == 0) ws.log = k;
                flat_log = k;
            }
            __syncthreads();
            log = ws.log;
        }
        if (log) {
            u32 l = ws.len[sym];
            ws.wt[sym] = l ? (u8)(log + 1 - l) : 0;
            u64 lastw; wg_scan_inclusive<u64, OpMaxU64>(l ? (u64)sym : 0, &lastw, red);
            // The tree description depends on the weights only, and blocks of one stream keep producing the same few weight
            // tables (packed random ACGT: one).  A first launch (sample_stride != 0) plans a sample of the blocks and leaves
            // their descriptions in `cache`; the launch over all blocks looks its weights up there and codes them itself
            // only on a miss -- the serial FSE coding of the weights by one lane was half of this kernel's time.
            u64 hk; wg_scan_inclusive<u64, OpAdd>(l ? (u64)(sym * 16 + (log + 1 - l) + 1) * 0x9E3779B97F4A7C15ull ^ ((u64)sym << 40) : 0, &hk, red);
            hk |= 1;                                                            // 0 marks an empty entry
            bool hit = false;
            if (cache && !sample_stride) {
                __shared__ u32 s_slot;
                if (threadIdx.x == 0) s_slot = ~0u;
                __syncthreads();
                if (cache->e[sym].key == hk) atomicMin(&s_slot, sym);
                __syncthreads();
                u32 slot_i = s_slot;
                if (slot_i != ~0u) {
                    const ZTreeEntry *ce = &cache->e[slot_i];
                    hit = __syncthreads_and(ce->wt[sym] == ws.wt[sym]) != 0;   // exact comparison: the hash only picks the entry
                    if (hit) { if (sym < ZENC_TREE_SLOT) ws.tree[sym] = ce->tree[sym]; if (sym == 0) ws.tb = ce->tb; }
                }
            }
            // direct weights (level 1, up to 128 of them): a byte per thread (huf_write_tree_w's layout)
            const bool direct_w = !try_fse && (u32)lastw >= 1 && (u32)lastw <= 128;
            defer = wt_defer && !sample_stride && !hit && !direct_w && (u32)lastw >= 1;
            if (!hit && direct_w) {
                const u32 nw = (u32)lastw;
                if (sym == 0) { ws.tree[0] = (u8)(127 + nw); ws.tb = 1 + (nw + 1) / 2; }
                if (2 * sym < nw) ws.tree[1 + sym] = (u8)((ws.wt[2 * sym] << 4) | (2 * sym + 1 < nw ? ws.wt[2 * sym + 1] : 0));
            }
            // FSE-coded weights (mandatory above 128 of them: packed bases with an N among them reach symbol 0xFF) are a serial job of a
            // few hundred dependent steps: one lane of this workgroup doing it kept the other 255 waiting -- 6.9 ms for the 30 K blocks of
            // a FASTQ's sequence stream.  The weights go to k_zenc_tree, which codes the trees of 64 blocks per wavefront, a lane each.
            else if (!hit && defer) wt_defer[(u64)b * 256 + sym] = ws.wt[sym];
            else if (!hit && threadIdx.x == 0) ws.tb = huf_write_tree_w(ws.tree, ws.wt, (u32)lastw, ws.tmp, ws.fse, try_fse != 0);
            if (cache && sample_stride) {
                __syncthreads();
                ZTreeEntry *ce = &cache->e[blockIdx.x];
                ce->wt[sym] = ws.wt[sym]; if (sym < ZENC_TREE_SLOT) ce->tree[sym] = ws.tree[sym];
                if (sym == 0) { ce->tb = ws.tb; ce->key = ws.tb ? hk : 0; }
            }
            u64 bits;
            if (flat_log) {                                                      // every code flat_log bits: a quarter's bits are its length times that
                const u32 perq = (bn + 3) / 4;
                for (u32 k = 0; k < 4; k++) p.ssz[k] = ((k < 3 ? perq : bn - 3 * perq) * flat_log + 8) >> 3;
            } else
            for (u32 k = 0; k < 4; k++) { wg_scan_inclusive<u64, OpAdd>((u64)hist[256 * k + sym] * l, &bits, red); p.ssz[k] = (u32)((bits + 1 + 7) / 8); }
            __syncthreads();
            u32 tb = ws.tb;
            if (defer) {
                // pending: k_zenc_tree makes the description and finishes the plan (log and the number of weights parked in the record)
                huf = true; p.log = (u8)log; p.tree_bytes = (u16)lastw; p.lhdr = (u8)(try_fse != 0);
                p.pad = (u8)(0x80u | ((log == 4 && distinct == 16) ? 1u : 0u));
            } else
            if (tb) { zenc_plan_finish(p, bn, log, tb, min_gain); huf = p.kind == ZK_HUF; if (huf && log == 4 && distinct == 16) p.pad = 1; }   // pad = 1: sixteen 4-bit codes (k_zenc_write packs such a block with all lanes)
        }
    }
    if (threadIdx.x == 0) { plan[b] = p; if (csize) csize[b] = p.csize; }
    if (huf && p.log && ws.len[plist[0]] == p.log && distinct == (1u << p.log)) {
        // one weight group (2^log symbols of log bits): the code of a symbol is its rank among the symbols that occur
        const u32 l = ws.len[sym];
        codes[(u64)b * 256 + sym] = l ? (u16)(myidx | (l << 12)) : (u16)0;
        if (!defer && sym < p.tree_bytes) trees[(u64)b * ZENC_TREE_SLOT + sym] = ws.tree[sym];
    } else
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
        codes[(u64)b * 256 + sym] = l ? (u16)(((wstart[wgt] >> (wgt - 1)) + rank) | (l << 12)) : (u16)0;   // code (<= 11 bits) | length << 12
        if (!defer && sym < p.tree_bytes) trees[(u64)b * ZENC_TREE_SLOT + sym] = ws.tree[sym];
    }
}

// A block of ZENC_FRAME_BLOCK bytes whose histogram is the sampled blocks' (every symbol that occurred at least once): the planner's own
// routines then make the frame's code lengths, codes and tree description of it (one more k_zenc_plan launch of one workgroup).
#define ZENC_FRAME_BLOCK 32768u
__global__ __launch_bounds__(256) void k_zenc_frame_block(const u32 *fhist, u8 *blockbuf, u32 *blen)
{
    __shared__ u64 red[4];
    const u32 sym = threadIdx.x, h = fhist[sym];
    u64 tot; wg_scan_inclusive<u64, OpAdd>((u64)h, &tot, red);
    u32 cnt = 0;
    if (h && tot) { cnt = (u32)(((u64)h * (ZENC_FRAME_BLOCK - 256)) / tot); if (!cnt) cnt = 1; }
    u64 all; const u64 inc = wg_scan_inclusive<u64, OpAdd>((u64)cnt, &all, red);
    u8 *o = blockbuf + (inc - cnt);
    for (u32 k = 0; k < cnt; k++) o[k] = (u8)sym;
    if (sym == 0) *blen = (u32)all;
}
// How far the frame's code is from the entropy of the histogram it was made of, in 1024ths (a Huffman code of a skewed alphabet -- a
// mask's units, half of them one value -- loses a few per cent whatever block it is made for): the bound a block's cost is held to
// is its own entropy times this.
__global__ __launch_bounds__(256) void k_zenc_frame_ratio(u32 *fhist, const u16 *fcodes)
{
    __shared__ u64 red[4];
    const u32 h = fhist[threadIdx.x], l = (u32)fcodes[threadIdx.x] >> 12;
    u64 tot = wg_reduce1<u64, OpAdd>((u64)h, red);
    __syncthreads();
    const float ent = (h && tot) ? (float)h * __log2f((float)tot / (float)h) : 0.f;
    const u64 both = wg_reduce1<u64, OpAdd>(((u64)h * l) << 0 | 0ull, red);
    __syncthreads();
    const u64 e = wg_reduce1<u64, OpAdd>((u64)(ent + 0.5f), red);
    if (threadIdx.x == 0) { u64 r = e ? both * 1024 / e : 1024; fhist[257] = (u32)(r < 1024 ? 1024 : r > 1229 ? 1229 : r); }
}
// ---- blocks settled under the frame's code WITHOUT a histogram (k_zenc_frame_quick) ---------------------------------------------------
// What the planner spends on a block is an LDS atomic per byte, half of them conflicts (profiles/r04_fastq_huf_counters.txt) -- 0.6 GB/ms on
// a device that reads 4 GB/ms -- and a frame whose blocks all look like its sample (a FASTQ's qualities, its packed reads with their Ns)
// learns nothing from 280 K histograms that one did not tell it.  What the block's plan needs from its bytes is the four streams' bit
// counts under the frame's code: sums of code lengths, a table look-up per byte.  The table is a row of 64 copies per symbol (lane l reads
// byte 64 s + l: lanes of a wavefront never meet in a word unless they read the same row), a wavefront per quarter of the block.
// Whether the block IS like the sample is judged by three moments of its bytes instead of its entropy: the sums of the code length, of its
// square and of a nibble hash of the byte, each within six standard deviations (of a block of independent bytes drawn from the sample's
// histogram) + 0.5 % of what the sample predicts.  A block that passes is planned as k_zenc_plan plans a block its 3 % rule accepts
// (same record, same sizes); one that fails, or holds a byte the code has no length for, is left to k_zenc_plan and its histogram.
// NAF_GPU_FRAME_QUICK=0: every block by its histogram.
#define ZQ_MISSING 0x80u
__device__ __forceinline__ u32 zq_hash4(u32 x) { return (x ^ (x >> 4)) & 0x0F0F0F0Fu; }     // per byte: low nibble ^ high nibble
__global__ __launch_bounds__(256) void k_zenc_frame_stats(const u32 *fhist, const u16 *fcodes, float *fstat)
{
    __shared__ u64 red[4];
    const u32 sym = threadIdx.x, h = fhist[sym], l = (u32)fcodes[sym] >> 12, x[3] = { l, l * l, (sym ^ (sym >> 4)) & 15u };
    const u64 N = wg_reduce1<u64, OpAdd>((u64)h, red);
    __syncthreads();
    for (u32 k = 0; k < 3; k++) {
        const u64 a = wg_reduce1<u64, OpAdd>((u64)h * x[k], red);
        __syncthreads();
        const u64 b2 = wg_reduce1<u64, OpAdd>((u64)h * x[k] * x[k], red);
        __syncthreads();
        if (sym == 0) {
            const double mu = N ? (double)a / (double)N : 0.0, var = N ? (double)b2 / (double)N - mu * mu : 0.0;
            fstat[2 * k] = (float)mu; fstat[2 * k + 1] = var > 0.0 ? (float)sqrt(var) : 0.f;
        }
    }
}
// tally / stride: a DRY run over every stride-th block that only counts how many it would take (tally[0]) of how many it looked at
// (tally[1]); gate: the tally of such a run -- the launch over all blocks leaves at once when the sample took fewer than half (reads whose
// qualities wander do not look like independent draws from the sample: nearly every block would fail the moments and the pass over the
// stream would be paid for nothing).
__global__ __launch_bounds__(256) void k_zenc_frame_quick(const u8 *src, u64 n, u32 nblk, ZEncPlan *plan, u16 *codes, u64 *csize, u8 *done,
                                                           const ZEncPlan *fplan, const u16 *fcodes, const float *fstat, u32 min_gain, u32 stride, u32 *tally, const u32 *gate)
{
    __shared__ __attribute__((aligned(16))) u8 tab[256 * 64];
    __shared__ u32 s_q[4][4], s_accept;
    const u32 b = stride ? blockIdx.x * stride : blockIdx.x, tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    if (!stride && gate && gate[0] * 2 < gate[1]) return;         // (uniform)
    if (b >= nblk || done[b] || fplan->kind != ZK_HUF) return;    // (uniform)
    const u64 lo = zenc_block_lo(n, nblk, b), hi = zenc_block_lo(n, nblk, b + 1);
    const u32 bn = (u32)(hi - lo);
    if (bn < 64) return;
    const u16 fc = fcodes[tid];
    {
        const u32 l = (u32)fc >> 12, v = (l ? l : ZQ_MISSING) * 0x01010101u;
        uint4 r; r.x = r.y = r.z = r.w = v;
        uint4 *row = (uint4 *)(tab + 64 * tid);
        row[0] = r; row[1] = r; row[2] = r; row[3] = r;
    }
    __syncthreads();
    const u32 per = (bn + 3) / 4;                                  // k_zenc_plan's quarters: [q per, (q + 1) per), the last one to the block's end
    const u32 qlo = q * per < bn ? q * per : bn, qhi = q == 3 ? bn : (qlo + per < bn ? qlo + per : bn);
    const u8 *s = src + lo, *my = tab + lane;
    u32 s1 = 0, s2 = 0, orr = 0, sh = 0;
    for (u32 i0 = qlo + lane * 8; i0 < qhi; i0 += 8 * 512) {
        u64 wv[8];
#pragma unroll
        for (u32 j = 0; j < 8; j++) { const u32 i = i0 + j * 512; wv[j] = i + 8 <= qhi ? ld64(s + i) : 0; }
#pragma unroll
        for (u32 j = 0; j < 8; j++) {
            const u32 i = i0 + j * 512;
            if (i >= qhi) break;
            if (i + 8 <= qhi) {
                const u32 x0 = (u32)wv[j], x1 = (u32)(wv[j] >> 32);
#pragma unroll
                for (u32 k = 0; k < 4; k++) {
                    const u32 t0 = my[((x0 >> (8 * k)) & 0xFF) << 6], t1 = my[((x1 >> (8 * k)) & 0xFF) << 6];
                    s1 += t0 + t1; s2 += t0 * t0; s2 += t1 * t1; orr |= t0 | t1;
                }
                sh = swar_dot4(zq_hash4(x0), 0x01010101u, sh); sh = swar_dot4(zq_hash4(x1), 0x01010101u, sh);
            } else for (u32 k = 0; i + k < qhi; k++) {
                const u32 c = s[i + k], t = my[c << 6];
                s1 += t; s2 += t * t; orr |= t; sh += (c ^ (c >> 4)) & 15u;
            }
        }
    }
#pragma unroll
    for (int d = 32; d; d >>= 1) { s1 += (u32)__shfl_xor((int)s1, d, 64); s2 += (u32)__shfl_xor((int)s2, d, 64); sh += (u32)__shfl_xor((int)sh, d, 64); orr |= (u32)__shfl_xor((int)orr, d, 64); }
    if (lane == 0) { s_q[q][0] = s1; s_q[q][1] = s2; s_q[q][2] = sh; s_q[q][3] = orr; }
    __syncthreads();
    ZEncPlan p;
    if (tid == 0) {
        u32 accept = 0;
        const u32 S[3] = { s_q[0][0] + s_q[1][0] + s_q[2][0] + s_q[3][0], s_q[0][1] + s_q[1][1] + s_q[2][1] + s_q[3][1], s_q[0][2] + s_q[1][2] + s_q[2][2] + s_q[3][2] };
        bool like = !((s_q[0][3] | s_q[1][3] | s_q[2][3] | s_q[3][3]) & ZQ_MISSING);
        const float fb = (float)bn, rt = sqrtf(fb);
        for (u32 k = 0; k < 3 && like; k++) {
            const float want = fb * fstat[2 * k], d = fabsf((float)S[k] - want);
            if (d > 6.f * rt * fstat[2 * k + 1] + 0.005f * want) like = false;
        }
        if (like) {
            p.n = bn; p.kind = ZK_RAW; p.csize = 3 + bn; p.log = 0; p.tree_bytes = 0; p.lhdr = 0; p.pad = 0; p.frame = 0; p.pad2 = 0;
            for (u32 k = 0; k < 4; k++) p.ssz[k] = (s_q[k][0] + 1 + 7) / 8;
            zenc_plan_finish(p, bn, fplan->log, 0, min_gain);
            if (p.kind == ZK_HUF && p.csize + fplan->tree_bytes <= 3 + bn) { p.frame = 1; accept = 1; }
            else if (p.kind != ZK_HUF) accept = 2;
        }
        if (stride) { atomicAdd(&tally[1], 1u); if (accept) atomicAdd(&tally[0], 1u); accept = 0; }
        if (accept) { plan[b] = p; if (csize) csize[b] = p.csize; done[b] = 1; }
        s_accept = accept;
    }
    __syncthreads();
    if (s_accept == 1) codes[(u64)b * 256 + tid] = fc;
}

// Which blocks of the frame's code carry the tree.  v[b] = 2 b + 1 for a Huffman block with a tree of its own, 2 b for one of the frame's
// code, -1 for the others, running maximum taken: a block of the frame's code is treeless when the Huffman block in front of it is one as well.
__global__ void k_zenc_frame_class(const ZEncPlan *plan, u32 nblk, i32 *v)
{
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblk) v[b] = plan[b].kind == ZK_HUF ? (i32)(2 * b + (plan[b].frame ? 0u : 1u)) : -1;
}
__global__ void k_zenc_frame_fix(ZEncPlan *plan, u32 nblk, const i32 *v, const ZEncPlan *fplan, const u8 *ftree, u8 *trees, u64 *csize)
{
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk || plan[b].kind != ZK_HUF || !plan[b].frame) return;
    const i32 prev = b ? v[b - 1] : -1;
    if (prev >= 0 && !(prev & 1)) return;                          // the tree in force is the frame's: treeless, as planned
    ZEncPlan p = plan[b];
    const u32 tb = fplan->tree_bytes;
    zenc_plan_force(p, tb);
    for (u32 k = 0; k < tb; k++) trees[(u64)b * ZENC_TREE_SLOT + k] = ftree[k];
    plan[b] = p; if (csize) csize[b] = p.csize;
}

// Tree descriptions k_zenc_plan left pending (plan.pad bit 7): one LANE per block, 64 blocks per wavefront, the weight coder's workspace
// of every lane in LDS at an odd word stride.  Finishes the plan (Huffman or Raw, sizes) exactly as k_zenc_plan would have.
struct ZTreeLane { FseWS fse; u8 tmp[160]; u8 w[256]; u32 odd; };
static_assert((sizeof(ZTreeLane) / 4) % 2 == 1, "odd word stride: the lanes' workspaces start in different LDS banks");
__global__ __launch_bounds__(64) void k_zenc_tree(u32 nblk, ZEncPlan *plan, u8 *trees, u64 *csize, const u8 *wt_defer, u32 min_gain)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    extern __shared__ __attribute__((aligned(16))) u8 tree_lds[];
    const u32 b = blockIdx.x * 64 + threadIdx.x;
    const bool pending = b < nblk && (plan[b].pad & 0x80u);
    if (!__ballot(pending)) return;
    if (!pending) return;
    ZTreeLane &L = ((ZTreeLane *)tree_lds)[threadIdx.x];
    ZEncPlan p = plan[b];
    const u32 lastw = p.tree_bytes, log = p.log, try_fse = p.lhdr, flat16 = p.pad & 1u;
    for (u32 k = 0; k < 256; k += 4) *(u32 *)(L.w + k) = *(const u32 *)(wt_defer + (u64)b * 256 + k);
    u8 *tree = trees + (u64)b * ZENC_TREE_SLOT;
    const u32 tb = huf_write_tree_w(tree, L.w, lastw, L.tmp, L.fse, try_fse != 0);
    p.kind = ZK_RAW; p.csize = 3 + p.n; p.log = 0; p.tree_bytes = 0; p.lhdr = 0; p.pad = 0; p.frame = 0; p.pad2 = 0;
    if (tb) { zenc_plan_finish(p, p.n, log, tb, min_gain); if (p.kind == ZK_HUF && flat16) p.pad = 1; }
    plan[b] = p;
    if (csize) csize[b] = p.csize;
}

// ======================= LZ stage (level >= 2; always for the id / name streams of an archive) ==============================
// Matches are searched inside a block only and every offset is coded as a new offset, so blocks stay independent (any
// block range can still be produced by any GPU).  One wavefront per block: 64 consecutive positions are tested per round
// against a hash table of earlier positions (block and table in LDS), the first lane with a match of >= LZ_MINMATCH wins,
// the literals before it and the (ll, ml, distance) triple are emitted, and the round restarts behind the match.
#define LZ_MINMATCH 5
// a value of lane l, l the same in every lane (found through a ballot): v_readlane_b32, not a trip through the LDS crossbar
__device__ __forceinline__ u32 rdlane(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
#define LZ_HASH_LOG 12
#define LZ_BLOCK_MAX 32768
#define LZ_LANE_EXT 32                    // bytes a lane extends its own match by before the wave takes over
struct LzBufs {
    u8 *lits; u64 slot;                 // literals of block b at lits + b*slot
    u16 *ll, *ml, *of; u64 seq_slot;    // sequences of block b at [b*seq_slot ..)
    u32 *ofv;                           // cross-block stage: Offset_Values (repeat codes 1..3, else distance + 3) instead of `of`
    u32 *nseq, *nlit;
    u8 *seqbuf; u32 *seq_bytes;         // encoded Sequences_Section of block b at seqbuf + b*slot
};
// Dynamic LDS: the block (block size + 320 bytes of zero padding) followed by the hash table; sized by the host for the
// block size in use, since LDS per wavefront is what bounds the blocks in flight (16 KiB blocks: 6 per CU, 32 KiB: 3).
__global__ __launch_bounds__(64) void k_lz_parse(const u8 *src, u64 n, u32 nblk, LzBufs B, u32 buf_bytes, u32 hash_log, const u8 *fallback = nullptr)
{
    if (fallback && !fallback[blockIdx.x]) return;              // k_lz_parse_lines has the block

    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    extern __shared__ __attribute__((aligned(16))) u8 lz_lds[];
    u8 *buf = lz_lds;
    u16 *tab = (u16 *)(lz_lds + buf_bytes);
    const u32 b = blockIdx.x, lane = threadIdx.x;
    const u64 lo = zenc_block_lo(n, nblk, b);
    const u32 bn = (u32)(zenc_block_lo(n, nblk, b + 1) - lo);
    for (u32 i = lane * 16; i < bn; i += 64 * 16) {
        if (i + 16 <= bn) { uint4 v; __builtin_memcpy(&v, src + lo + i, 16); *(uint4 *)(buf + i) = v; }
        else for (u32 k = i; k < bn; k++) buf[k] = src[lo + k];
    }
    for (u32 i = lane; i < (1u << hash_log); i += 64) tab[i] = 0;     // 0 = empty, else position + 1
    for (u32 i = bn + lane; i < bn + 320 && i < buf_bytes; i += 64) buf[i] = 0;
    __syncthreads();
    u8 *lits = B.lits + (u64)b * B.slot;
    u16 *sll = B.ll + (u64)b * B.seq_slot, *sml = B.ml + (u64)b * B.seq_slot, *sof = B.of + (u64)b * B.seq_slot;
    u32 cur = 0, anchor = 0, ns = 0, nl = 0;
    while (cur + LZ_MINMATCH <= bn) {
        u32 p = cur + lane; bool valid = p + LZ_MINMATCH <= bn;
        u32 v = 0, h = 0, c = 0, m = 0;
        if (valid) {
            __builtin_memcpy(&v, buf + p, 4);
            h = (v * 2654435761u) >> (32 - hash_log);
            c = tab[h];                                              // positions of earlier rounds only (< cur)
            if (c) {
                u32 q = c - 1, cv; __builtin_memcpy(&cv, buf + q, 4);
                if (cv == v) {
                    m = 4;
                    for (;;) {                                       // 8 bytes per step; buf is zero-padded behind bn
                        u64 x, y; __builtin_memcpy(&x, buf + q + m, 8); __builtin_memcpy(&y, buf + p + m, 8);
                        const u64 d = x ^ y;
                        if (d) { m += (u32)(__ffsll((long long)d) - 1) >> 3; break; }
                        m += 8;
                        if (p + m >= bn || m >= LZ_LANE_EXT) break;   // long matches are finished by the whole wave (below)
                    }
                    if (p + m > bn) m = bn - p;
                }
            }
        }
        u64 win = __ballot(m >= LZ_MINMATCH);
        if (valid) tab[h] = (u16)(p + 1);                            // later lanes win ties; any earlier position is a fine candidate
        // greedy left to right over this round's 64 positions: take the first match at or after `anchor`, jump behind it, repeat
        u32 next = cur < anchor ? anchor : cur;
        while (next < cur + 64) {
            u64 w2 = win & ~((1ull << (next - cur)) - 1);
            if (!w2) break;
            u32 f = (u32)__ffsll((long long)w2) - 1;
            u32 pf = cur + f, mf = rdlane(m, f), cf = rdlane(c, f) - 1;
            while (mf >= LZ_LANE_EXT && pf + mf < bn) {              // 256 bytes per step: lane k compares bytes [4k, 4k+4) behind the match so far
                u32 x, y; __builtin_memcpy(&x, buf + cf + mf + 4 * lane, 4); __builtin_memcpy(&y, buf + pf + mf + 4 * lane, 4);
                u32 d = x ^ y;                                        // reads stay inside the zero padding behind bn (bn + 256 + 4)
                u64 ne = __ballot(d != 0);
                if (!ne) { mf += 256; continue; }
                u32 l0 = (u32)__ffsll((long long)ne) - 1;
                u32 d0 = rdlane(d, l0);
                mf += 4 * l0 + (((u32)__ffs((int)d0) - 1) >> 3);
                break;
            }
            if (pf + mf > bn) mf = bn - pf;
            u32 ll = pf - anchor;
            for (u32 k = lane; k < ll; k += 64) lits[nl + k] = buf[anchor + k];
            if (lane == 0) { sll[ns] = (u16)ll; sml[ns] = (u16)mf; sof[ns] = (u16)(pf - cf); }
            nl += ll; ns++;
            anchor = next = pf + mf;
        }
        cur += 64;
        if (cur < anchor) cur = anchor;                             // a match ran past the round: resume behind it
    }
    for (u32 k = lane; k < bn - anchor; k += 64) lits[nl + k] = buf[anchor + k];
    nl += bn - anchor;
    if (lane == 0) { B.nseq[b] = ns; B.nlit[b] = nl; }
}

// ---- streams of zero-terminated names: a LANE per line against the line in front of it (k_lz_parse_lines) --------------------------------
// A FASTQ's read names are 38 M strings that differ from their predecessor in a digit or two.  k_lz_parse finds that -- one match per
// name, offset = the length of the name in front -- but finds it SERIALLY: its greedy walk over a round's 64 positions takes the matches
// one after the other, five names a round, 7.5 ms for the 0.5 GB of names of a 12.5 GB FASTQ (a sixth of the whole encode, on a
// stream that is a twenty-fifth of the text).  Here a lane takes a LINE (the bytes up to and including a zero byte) and compares it with
// the line in front of it: the common prefix (segment A, columns aligned: offset = the previous line's length) and the common suffix
// (segment B, ends aligned: offset = the line's own length -- Illumina names share their comment wherever the coordinates end).  Nothing
// is serial between lines: where a line's literals go and which sequence numbers it writes are prefix sums over the lines.  A prefix
// that continues the match the line in front ended on (a suffix match reaches the terminator at the very offset the next prefix has; a
// line that is wholly its prefix does when the lines are equally long) JOINS it instead of opening a sequence -- what the
// greedy walk gets from running across the line end: `SRR1.12 length=150\0SRR1.1` is one match, and comments that are all alike one per
// block.  The head a joining prefix adds its length to is the sequence open at the end of the line in front, handed on through lines that
// are wholly one match by a scan.  Blocks the picture does not fit -- fewer than half of the bytes matched, more lines than an eighth of
// the bytes, fewer than two, lines of more than 128 bytes on average -- are left to k_lz_parse (fallback[b] = 1), which skips the others.  NAF_GPU_LZ_LINES=0: every block by k_lz_parse.
__device__ __forceinline__ u64 zero_bytes64(u64 w) { const u64 L = 0x7F7F7F7F7F7F7F7Full; return ~(((w & L) + L) | w) & ~L; }   // 0x80 where the byte is zero
// line_div: a block may hold bn / line_div lines (8; 16 for blocks above 16 KiB, whose tables would otherwise leave two workgroups a CU --
// comments of eight bytes in such blocks are the hash table's walk's, which takes a block of them as one match)
__global__ __launch_bounds__(64) void k_lz_parse_lines(const u8 *src, u64 n, u32 nblk, LzBufs B, u32 buf_bytes, u8 *fallback, ZEncPlan *plan, u64 *csize, u8 *done, u32 line_div, u32 ml_slots)
{
    extern __shared__ __attribute__((aligned(16))) u8 lz_lds[];
    u8 *buf = lz_lds;
    u32 *s_ml = (u32 *)(lz_lds + buf_bytes);                              // match length of sequence q: its head's own bytes + what joined it
    u16 *ends = (u16 *)(lz_lds + buf_bytes + 4 * (size_t)ml_slots);
    const u32 b = blockIdx.x, lane = threadIdx.x;
    const u64 lo = zenc_block_lo(n, nblk, b);
    const u32 bn = (u32)(zenc_block_lo(n, nblk, b + 1) - lo);
    const u32 max_lines = bn / line_div;
    for (u32 i = lane * 16; i < bn; i += 64 * 16) {
        if (i + 16 <= bn) { uint4 v; __builtin_memcpy(&v, src + lo + i, 16); *(uint4 *)(buf + i) = v; }
        else for (u32 k = i; k < bn; k++) buf[k] = src[lo + k];
    }
    for (u32 i = bn + lane; i < bn + 64 && i < buf_bytes; i += 64) buf[i] = 0xFF;
    __syncthreads();
    // the terminators: a lane's share of the block, eight bytes at a time
    const u32 per = (((bn + 63) / 64) + 7) & ~7u, c0 = lane * per < bn ? lane * per : bn, c1 = c0 + per < bn ? c0 + per : bn;
    u32 cnt = 0;
    for (u32 i = c0; i < c1; i += 8) { u64 z = zero_bytes64(*(const u64 *)(buf + i)); if (i + 8 > c1) z &= (1ull << (8 * (c1 - i))) - 1; cnt += (u32)__popcll(z); }
    const u32 incl = wave_scan_inclusive<u32, OpAdd>(cnt), L = (u32)__shfl((int)incl, 63, 64);
    // (lines of more than 128 bytes on average are not names: what repeats INSIDE such a line -- a run of one letter -- is the hash table's
    // walk's to find, and it has few matches a block to take one after the other)
    if (L < 2 || L > max_lines || L * 128 < bn) { if (lane == 0) fallback[b] = 1; return; }       // (uniform)
    {
        u32 idx = incl - cnt;
        for (u32 i = c0; i < c1; i += 8) {
            u64 z = zero_bytes64(*(const u64 *)(buf + i)); if (i + 8 > c1) z &= (1ull << (8 * (c1 - i))) - 1;
            while (z) { ends[idx++] = (u16)(i + (((u32)__ffsll((long long)z) - 1) >> 3)); z &= z - 1; }
        }
    }
    __syncthreads();
    const u32 last_end = (u32)ends[L - 1] + 1, NL = L + (last_end < bn ? 1u : 0u);   // (the bytes behind the last terminator: a line without one)
    u16 *sll = B.ll + (u64)b * B.seq_slot, *sml = B.ml + (u64)b * B.seq_slot, *sof = B.of + (u64)b * B.seq_slot;
    u8 *lits = B.lits + (u64)b * B.slot;
    u32 seq_base = 0, matched_base = 0, anchor = 0;
    u32 c_open = 0xFFFFFFFFu, c_of = 0; bool c_tail = false;             // the line in front of the round's first: the sequence open at its end, its offset, whether its last match reached its end
    for (u32 r = 0; r < NL; r += 64) {
        const u32 i = r + lane;
        u32 s = 0, e = 0, of = 0, m1 = 0, k2 = 0, m2 = 0, len = 0;
        if (i < NL) {
            s = i ? (u32)ends[i - 1] + 1 : 0u; e = i < L ? (u32)ends[i] + 1 : bn; len = e - s;
            if (i >= 1) {
                of = s - (i >= 2 ? (u32)ends[i - 2] + 1 : 0u);
                // the common prefix (up to 256 bytes of it), then the common suffix
                const u32 lim = len < 256 ? len : 256u;
                u32 k = 0;
                for (; k < lim; k += 8) { u64 x, y; __builtin_memcpy(&x, buf + s + k, 8); __builtin_memcpy(&y, buf + s + k - of, 8); const u64 d = x ^ y; if (d) { k += ((u32)__ffsll((long long)d) - 1) >> 3; break; } }
                m1 = k < lim ? k : lim;
                if (m1 < len) {
                    // the common SUFFIX with the line in front, ends aligned (offset = this line's own length: the lines need not be equally long):
                    // `...:1101:2345:6789 1:N:0:ACGT` shares its comment with its predecessor wherever the coordinates end
                    const u32 plen = of;                                               // (the line in front is [s - of, s))
                    u32 maxs = len - m1; if (maxs > plen) maxs = plen; if (maxs > 256) maxs = 256;
                    u32 ks = 0; bool stop = false;
                    for (; ks + 8 <= maxs; ks += 8) {
                        u64 x, y; __builtin_memcpy(&x, buf + e - 8 - ks, 8); __builtin_memcpy(&y, buf + s - 8 - ks, 8);
                        const u64 d = x ^ y;
                        if (d) { ks += (u32)__clzll((long long)d) >> 3; stop = true; break; }
                    }
                    if (!stop) while (ks < maxs && buf[e - 1 - ks] == buf[s - 1 - ks]) ks++;
                    m2 = ks; k2 = len - m2;
                }
                if (m2 < LZ_MINMATCH) m2 = 0;
            }
        }
        const bool full = len && m1 == len;                                            // the whole line is its prefix
        const bool tail = (m2 && k2 + m2 == len) || (full && len >= LZ_MINMATCH);      // its last match reaches its end
        const u32 tail_of = m2 ? len : of;                                             // ... at this offset (a suffix match's is the line's own length)
        // does the prefix continue the match the line in front ended on?
        const u32 p_of = (u32)__shfl_up((int)tail_of, 1, 64); const bool p_tail = __shfl_up((int)tail, 1, 64) != 0;
        const bool join = i < NL && i >= 1 && m1 >= 1 && (lane ? (p_tail && p_of == of) : (c_tail && c_of == of));
        const bool a_on = join || m1 >= LZ_MINMATCH, a_head = a_on && !join, b_on = m2 != 0;
        const u32 la = a_on ? m1 : 0u;
        const u32 ns_i = (a_head ? 1u : 0u) + (b_on ? 1u : 0u), mt_i = la + m2;
        const u32 end_i = b_on ? s + k2 + m2 : (a_on ? s + la : 0u);
        const u32 ns_in = wave_scan_inclusive<u32, OpAdd>(ns_i), mt_in = wave_scan_inclusive<u32, OpAdd>(mt_i);
        const u32 an_in = (u32)wave_scan_inclusive<i32, OpMax>((i32)end_i);
        const u32 an_ex = (u32)__shfl_up((int)an_in, 1, 64);
        const u32 my_anchor = lane ? (an_ex > anchor ? an_ex : anchor) : anchor;
        const u32 my_seq = seq_base + ns_in - ns_i, my_matched = matched_base + mt_in - mt_i;
        // the sequence open at the end of each line: B's when B reaches the end; a full line's own head, or -- when it joined -- the one it
        // joined (handed on from the nearest line in front that is not such a line)
        const bool pass = tail && !(m2 && k2 + m2 == len) && join;
        const u32 own_open = !tail ? 0xFFFFFFFFu : ((m2 && k2 + m2 == len) ? my_seq + (a_head ? 1u : 0u) : my_seq);
        const i32 srcl = wave_scan_inclusive<i32, OpMax>(pass ? -1 : (i32)lane);
        const u32 got = (u32)__shfl((int)own_open, srcl < 0 ? 0 : srcl, 64);
        const u32 open_i = srcl < 0 ? c_open : got;
        const u32 p_open = (u32)__shfl_up((int)open_i, 1, 64);
        if (i < NL) {
            u32 q = my_seq, a = my_anchor;
            if (a_head) { sll[q] = (u16)(s - a); sof[q] = (u16)of; s_ml[q] = m1; q++; }
            if (a_on) a = s + m1;
            if (b_on) { sll[q] = (u16)(s + k2 - a); sof[q] = (u16)len; s_ml[q] = m2; }
        }
        if (join) atomicAdd(&s_ml[lane ? p_open : c_open], m1);                        // (behind the heads' stores: LDS operations of a wavefront keep their order)
        if (i < NL) {
            // literals: the line's bytes outside its matches, to their place among the block's literals
            const u32 g0 = s + la, g1 = b_on ? s + k2 : e;
            u32 dstp = g0 - (my_matched + la);
            for (u32 k = g0; k < g1; k++) lits[dstp++] = buf[k];
            if (b_on) for (u32 k = s + k2 + m2; k < e; k++) lits[dstp++] = buf[k];
        }
        seq_base += (u32)__shfl((int)ns_in, 63, 64); matched_base += (u32)__shfl((int)mt_in, 63, 64);
        const u32 an_last = (u32)__shfl((int)an_in, 63, 64);
        if (an_last > anchor) anchor = an_last;
        c_open = (u32)__shfl((int)open_i, 63, 64); c_of = (u32)__shfl((int)tail_of, 63, 64); c_tail = __shfl((int)tail, 63, 64) != 0;
    }
    if (matched_base * 2 < bn) { if (lane == 0) fallback[b] = 1; return; }
    __syncthreads();
    for (u32 k = lane; k < seq_base; k += 64) sml[k] = (u16)s_ml[k];
    if (lane == 0) {
        B.nseq[b] = seq_base; B.nlit[b] = bn - matched_base; fallback[b] = 0;
        // the block without its matches: Raw, as far as k_lz_choose is told (k_zenc_plan leaves it alone)
        ZEncPlan p; p.n = bn; p.kind = ZK_RAW; p.csize = 3 + bn; p.log = 0; p.tree_bytes = 0; p.lhdr = 0; p.pad = 0; p.frame = 0; p.pad2 = 0;
        p.ssz[0] = p.ssz[1] = p.ssz[2] = p.ssz[3] = 0;
        plan[b] = p; if (csize) csize[b] = p.csize; done[b] = 1;
    }
}

// Sequences_Section of every block (one lane per block; predefined FSE encoding tables staged in LDS)
__global__ __launch_bounds__(64) void k_lz_seqenc(u32 nblk, LzBufs B, const SeqCTabs *tabs)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    __shared__ SeqCTabs T;
    for (u32 i = threadIdx.x; i < sizeof(SeqCTabs) / 4; i += 64) ((u32 *)&T)[i] = ((const u32 *)tabs)[i];
    __syncthreads();
    u32 b = blockIdx.x * 64 + threadIdx.x;
    if (b >= nblk) return;
    SeqCTab ct[3]; zenc_seq_ctabs(T, ct);
    u32 ns = B.nseq[b], bytes = 1;
    u8 *out = B.seqbuf + (u64)b * B.slot;
    if (ns == 0) out[0] = 0;
    else bytes = zenc_write_sequences(out, (u32)B.slot, B.ll + (u64)b * B.seq_slot, B.ml + (u64)b * B.seq_slot, B.of + (u64)b * B.seq_slot, ns, ct);
    B.seq_bytes[b] = bytes;                                              // 0: does not fit its slot, the block stays literal-only
}

// ======================= cross-block matching (level >= 2, --long N) =========================================================================
// What the reference gets from libzstd's match finders and its long-distance matcher (compressor.c:7-21): matches against anything
// inside the window, repeat-offset codes, per-block FSE tables.  Built for a GPU as
//   k_ldm_insert   every ANCHOR position of the stream (content-defined: the hash of its 8 bytes has three zero top bits, so one in
//                  eight, and the same ones in every copy of a repeat) enters a table keyed by its 16 bytes; the table keeps the
//                  FIRST occurrence per EPOCH of 2^(wlog-1) bytes (atomicMin), so a position finds a source in its own epoch or
//                  the one before it -- never further back than the window -- and all positions are inserted at once;
//   k_lzx_parse    one wavefront per block as in k_lz_parse, with two more sources of matches: the table (source anywhere in the
//                  window, read from HBM) and, behind every match, the SAME offset again after one to three literals (a
//                  substituted base costs a repeat code, not a new offset); matches grow backwards into the pending literals;
//   k_lzx_seqenc   Offset_Values with the block's own repeat-offset history (zstd_enc_core.h: RepState), code tables chosen per
//                  block between predefined, RLE and FSE_Compressed.
// Blocks still do not depend on each other's OUTPUT, so a shard of a sharded archive codes its part of the frame alone, matching
// inside its own part of the stream.
struct LdmTab { u32 *tab; u32 elog, tlog, wlog, pad; };
__device__ __forceinline__ u64 ldm_mix(u64 a) { return a * 0x9E3779B185EBCA87ull; }
__device__ __forceinline__ bool ldm_key(u64 a, u64 b, u32 tlog, u32 &idx)
{
    const u64 ha = ldm_mix(a);
    if (ha >> 61) return false;
    idx = (u32)((ha ^ (ldm_mix(b ^ 0x5555555555555555ull) >> 7)) >> (64 - tlog));
    return true;
}
__global__ __launch_bounds__(256) void k_ldm_insert(const u8 *src, u64 n, LdmTab L)
{
    const u64 p0 = ((u64)blockIdx.x * 256 + threadIdx.x) * 8;
    if (p0 + 16 > n) return;
    const u64 w0 = ld64(src + p0), w1 = ld64(src + p0 + 8);
    u64 w2 = 0;
    if (p0 + 24 <= n) w2 = ld64(src + p0 + 16); else for (u64 k = p0 + 16; k < n; k++) w2 |= (u64)src[k] << (8 * (k - p0 - 16));
    const u64 E1 = (1ull << L.elog) - 1;
#pragma unroll
    for (u32 k = 0; k < 8; k++) {
        const u64 pos = p0 + k;
        if (pos + 16 > n) break;
        const u64 a = k ? (w0 >> (8 * k)) | (w1 << (64 - 8 * k)) : w0, b = k ? (w1 >> (8 * k)) | (w2 << (64 - 8 * k)) : w1;
        u32 idx;
        if (ldm_key(a, b, L.tlog, idx)) atomicMin(&L.tab[((pos >> L.elog) << L.tlog) + idx], (u32)(pos & E1));
    }
}

// bytes equal from target position t (block-relative, in LDS) and source position s (block-relative, may be negative: HBM), by the
// whole wave, 256 bytes per step; stops at the block end.  The LDS copy is zero-padded behind bn, HBM is not read past n.
__device__ __forceinline__ u32 lzx_wave_match(const u8 *buf, const u8 *gblk /* src + lo */, u64 n_left /* n - lo */, u32 bn, u32 t, i64 s, u32 m0)
{
    const u32 lane = threadIdx.x;
    u32 m = m0;
    while (t + m < bn) {
        const i64 sp = s + (i64)m + 4 * (i64)lane;
        u32 x, y; __builtin_memcpy(&y, buf + t + m + 4 * lane, 4);
        bool oob = false;
        if (sp >= 0) __builtin_memcpy(&x, buf + sp, 4);               // (sp < t + m + 4 lane: inside the padded copy)
        else if (sp + 4 <= (i64)n_left) __builtin_memcpy(&x, gblk + sp, 4);
        else { x = 0; oob = true; }
        const u32 d = x ^ y;
        const u64 ne = __ballot(d != 0 || oob);
        if (!ne) { m += 256; continue; }
        const u32 l0 = (u32)__ffsll((long long)ne) - 1;
        const u32 d0 = rdlane(d, l0);
        m += 4 * l0 + (d0 ? ((u32)__ffs((int)d0) - 1) >> 3 : 0u);
        break;
    }
    if (t + m > bn) m = bn - t;
    return m;
}
// equal bytes in front of (t, s), at most t - floor of them and never in front of the stream's first byte
__device__ __forceinline__ u32 lzx_wave_back(const u8 *buf, const u8 *gblk, u64 lo, u32 t, i64 s, u32 floor_)
{
    const u32 lane = threadIdx.x;
    u32 back = 0;
    for (;;) {
        const u32 k = back + lane + 1;                               // this lane looks at t - k and s - k
        bool ok = t >= floor_ + k && (i64)lo + s - (i64)k >= 0;
        if (ok) { const i64 sp = s - (i64)k; const u32 x = sp >= 0 ? buf[sp] : gblk[sp]; ok = x == buf[t - k]; }
        const u64 bad = __ballot(!ok);
        if (!bad) { back += 64; continue; }
        back += (u32)__ffsll((long long)bad) - 1;
        break;
    }
    return back;
}

__global__ __launch_bounds__(64) void k_lzx_parse(const u8 *src, u64 n, u32 nblk, LzBufs B, u32 buf_bytes, LdmTab L)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    extern __shared__ __attribute__((aligned(16))) u8 lz_lds[];
    u8 *buf = lz_lds;
    u16 *tab = (u16 *)(lz_lds + buf_bytes);
    const u32 b = blockIdx.x, lane = threadIdx.x;
    const u64 lo = zenc_block_lo(n, nblk, b);
    const u32 bn = (u32)(zenc_block_lo(n, nblk, b + 1) - lo);
    const u8 *gblk = src + lo;
    for (u32 i = lane * 16; i < bn; i += 64 * 16) {
        if (i + 16 <= bn) { uint4 v; __builtin_memcpy(&v, gblk + i, 16); *(uint4 *)(buf + i) = v; }
        else for (u32 k = i; k < bn; k++) buf[k] = gblk[k];
    }
    for (u32 i = lane; i < (1u << LZ_HASH_LOG); i += 64) tab[i] = 0;     // 0 = empty, else position + 1
    for (u32 i = bn + lane; i < bn + 320 && i < buf_bytes; i += 64) buf[i] = 0;
    __syncthreads();
    u8 *lits = B.lits + (u64)b * B.slot;
    u16 *sll = B.ll + (u64)b * B.seq_slot, *sml = B.ml + (u64)b * B.seq_slot; u32 *sof = B.ofv + (u64)b * B.seq_slot;
    const u32 max_seq = (u32)B.seq_slot - 2;
    RepState R; R.r[0] = R.r[1] = R.r[2] = 0;
    u32 cur = 0, anchor = 0, ns = 0, nl = 0;
    // literals [anchor, at) and the match (at, ml, distance d) leave the parser; `anchor` moves behind the match
    auto emit = [&](u32 at, u32 ml, u32 d) {
        if (ml > 65535u) ml = 65535u;                                // lengths are kept in 16 bits; the rest of the match is found again
        const u32 ll = at - anchor;
        for (u32 k = lane; k < ll; k += 64) lits[nl + k] = buf[anchor + k];
        const u32 v = zenc_offset_value(R, d, ll);
        if (lane == 0) { sll[ns] = (u16)ll; sml[ns] = (u16)ml; sof[ns] = v; }
        nl += ll; ns++;
        anchor = at + ml;
    };
    while (cur + LZ_MINMATCH <= bn && ns + 8 < max_seq) {
        const u32 p = cur + lane; const bool valid = p + LZ_MINMATCH <= bn;
        u32 v = 0, h = 0, m = 0; i64 c = 0;
        if (valid) {
            __builtin_memcpy(&v, buf + p, 4);
            h = (v * 2654435761u) >> (32 - LZ_HASH_LOG);
            const u32 ci = tab[h];                                   // positions of earlier rounds only (< cur)
            if (ci) {
                const u32 q = ci - 1; u32 cv; __builtin_memcpy(&cv, buf + q, 4);
                if (cv == v) {
                    m = 4; c = (i64)q;
                    for (;;) {
                        u32 x, y; __builtin_memcpy(&x, buf + q + m, 4); __builtin_memcpy(&y, buf + p + m, 4);
                        const u32 d = x ^ y;
                        if (d) { m += (u32)(__ffs((int)d) - 1) >> 3; break; }
                        m += 4;
                        if (p + m >= bn || m >= LZ_LANE_EXT) break;
                    }
                }
            }
            if (p + 16 <= bn) {                                      // an anchor: a source anywhere in the window
                u64 a8, b8; __builtin_memcpy(&a8, buf + p, 8); __builtin_memcpy(&b8, buf + p + 8, 8);
                u32 idx;
                if (ldm_key(a8, b8, L.tlog, idx)) {
                    const u64 pa = lo + p, e = pa >> L.elog;
                    u64 q = ~0ull;
                    const u32 c1 = L.tab[(e << L.tlog) + idx];
                    if (c1 != 0xFFFFFFFFu && (e << L.elog) + c1 < pa) q = (e << L.elog) + c1;
                    else if (e > 0) { const u32 c0 = L.tab[((e - 1) << L.tlog) + idx]; if (c0 != 0xFFFFFFFFu) q = ((e - 1) << L.elog) + c0; }
                    if (q != ~0ull && pa - q < (1ull << L.wlog) && pa - q < 0x7FFFFFF0ull) {
                        const u8 *g = src + q;
                        if (ld64(g) == a8 && ld64(g + 8) == b8) {
                            u32 m2 = 16;
                            while (p + m2 < bn && m2 < LZ_LANE_EXT && q + m2 + 4 <= n) {
                                u32 x, y; __builtin_memcpy(&x, g + m2, 4); __builtin_memcpy(&y, buf + p + m2, 4);
                                const u32 d = x ^ y;
                                if (d) { m2 += (u32)(__ffs((int)d) - 1) >> 3; break; }
                                m2 += 4;
                            }
                            if (m2 > m) { m = m2; c = (i64)q - (i64)lo; }
                        }
                    }
                }
            }
            if (p + m > bn) m = bn - p;
        }
        const u64 win = __ballot(m >= LZ_MINMATCH);
        if (valid) tab[h] = (u16)(p + 1);
        u32 next = cur < anchor ? anchor : cur;
        while (next < cur + 64 && ns + 8 < max_seq) {
            const u64 w2 = win & ~((1ull << (next - cur)) - 1);
            if (!w2) break;
            const u32 f = (u32)__ffsll((long long)w2) - 1;
            u32 pf = cur + f, mf = rdlane(m, f);
            const u32 clo = rdlane((u32)(u64)c, f), chi = rdlane((u32)((u64)c >> 32), f);
            i64 cf = (i64)(((u64)chi << 32) | clo);
            if (mf >= LZ_LANE_EXT) mf = lzx_wave_match(buf, gblk, n - lo, bn, pf, cf, mf);
            const u32 back = lzx_wave_back(buf, gblk, lo, pf, cf, anchor);
            pf -= back; cf -= back; mf += back;
            const u32 d = (u32)((i64)pf - cf);
            emit(pf, mf, d);
            // the same offset again behind one to three literals: a substitution inside a repeat
            for (;;) {
                bool again = false;
                for (u32 skip = 1; skip <= 3 && !again; skip++) {
                    const u32 t = anchor + skip;
                    if (t + 4 > bn) break;
                    const u32 mr = lzx_wave_match(buf, gblk, n - lo, bn, t, (i64)t - (i64)d, 0);
                    if (mr >= 4 && ns + 8 < max_seq) { emit(t, mr, d); again = true; }
                }
                if (!again) break;
            }
            next = anchor;
        }
        cur += 64;
        if (cur < anchor) cur = anchor;
    }
    for (u32 k = lane; k < bn - anchor; k += 64) lits[nl + k] = buf[anchor + k];
    nl += bn - anchor;
    if (lane == 0) { B.nseq[b] = ns; B.nlit[b] = nl; }
}

// Sequences_Section of one block per wavefront: the tables live in LDS, lane 0 writes (zenc_write_sequences_x)
__global__ __launch_bounds__(64) void k_lzx_seqenc(u32 nblk, LzBufs B, const SeqCTabs *tabs)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    __shared__ SeqCTabs T; __shared__ SeqWS ws;
    for (u32 i = threadIdx.x; i < sizeof(SeqCTabs) / 4; i += 64) ((u32 *)&T)[i] = ((const u32 *)tabs)[i];
    __syncthreads();
    const u32 b = blockIdx.x;
    if (threadIdx.x) return;
    const u32 ns = B.nseq[b]; u32 bytes = 1;
    u8 *out = B.seqbuf + (u64)b * B.slot;
    if (ns == 0) out[0] = 0;
    else bytes = zenc_write_sequences_x(out, (u32)B.slot, B.ll + (u64)b * B.seq_slot, B.ml + (u64)b * B.seq_slot, B.ofv + (u64)b * B.seq_slot, ns, T, ws);
    B.seq_bytes[b] = bytes;                                              // 0: does not fit its slot, the block stays literal-only
}

// literals section size of an LZ block from the plan of its literals
__host__ __device__ static inline u32 lz_lit_section_bytes(const ZEncPlan &p)
{
    if (p.kind == ZK_HUF) return p.csize - 3 - 1;
    u32 hdr = p.n < 32 ? 1 : (p.n < 4096 ? 2 : 3);
    return p.kind == ZK_RLE ? hdr + 1 : hdr + p.n;
}
// per block: the smaller of the literal-only coding (plan0) and the LZ coding (plan1 + sequences)
__global__ void k_lz_choose(u32 nblk, const ZEncPlan *plan0, const ZEncPlan *plan1, const u32 *nseq, const u32 *seq_bytes, u8 *mode, u64 *csize)
{
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    u32 s0 = plan0[b].csize, s1 = 3 + lz_lit_section_bytes(plan1[b]) + seq_bytes[b];
    bool lz = nseq[b] > 0 && seq_bytes[b] > 0 && s1 < s0;
    mode[b] = lz ? 1 : 0; csize[b] = lz ? s1 : s0;
}

#define ZENC_BLOCKS_PER_WG 16
#define ZENC_OROW 80                      // LDS output row per lane: one 64-byte segment + the 8-byte store that may start at its byte 63, 16-byte aligned
// One Huffman stream (4.2.2: written forward so that the LAST symbol is read first) by one lane.  Input is pulled 64 bytes
// at a time into registers (walking down), output is collected in the lane's LDS row and leaves as aligned 64-byte
// segments: per-lane 8-byte loads re-fetch every line 4x and per-lane 8-byte stores cost 6x the bytes at the HBM.
// Codes are appended to a 64-bit accumulator in groups that cannot overflow it (8 symbols when codes are at most 7 bits long,
// else 4: fewer than 8 bits are pending between groups), and whole bytes move to the row once per group -- no test per symbol.
template <int GROUP>
__device__ __forceinline__ void huf_encode_stream_staged(u8 *out, const u8 *src, u32 n, const u16 *codes, u8 *orow)
{
    u64 acc = 0; u32 nb = 0;
    u32 lo = (u32)((uintptr_t)out & 63), fill = lo;               // row[lo..fill) = bytes of the 64-byte segment at seg not yet written
    u8 *seg = out - lo;
    auto flush_group = [&]() {
        __builtin_memcpy(orow + fill, &acc, 8);                    // unaligned LDS store; only the whole bytes are kept
        fill += nb >> 3; acc >>= (nb & ~7u); nb &= 7;
        if (fill >= 64) {
            if (lo == 0) { const uint4 *r = (const uint4 *)orow; uint4 *g = (uint4 *)seg; g[0] = r[0]; g[1] = r[1]; g[2] = r[2]; g[3] = r[3]; }
            else { for (u32 k = lo; k < 64; k++) seg[k] = orow[k]; lo = 0; }       // the stream's first, partial segment
            u64 t0, t1; __builtin_memcpy(&t0, orow + 64, 8); __builtin_memcpy(&t1, orow + 72, 8);
            __builtin_memcpy(orow, &t0, 8); __builtin_memcpy(orow + 8, &t1, 8);
            seg += 64; fill -= 64;
        }
    };
    auto put_e = [&](u32 e) { acc |= (u64)(e & 0xFFF) << nb; nb += e >> 12; };
    u32 i = n;
    while (i >= 64) {                                             // symbols i-1 .. i-64, highest index first
        i -= 64;
        uint4 q[4]; __builtin_memcpy(q, src + i, 64);
#pragma unroll
        for (int wdx = 7; wdx >= 0; wdx--) {
            const uint4 &qq = q[wdx >> 1];
            u64 w = (wdx & 1) ? ((u64)qq.z | ((u64)qq.w << 32)) : ((u64)qq.x | ((u64)qq.y << 32));
            // the eight table entries first, then the appends: inside the append chain every symbol would wait for its own LDS
            // round trip (the row stores in between keep the compiler from hoisting the lookups)
            u32 e[8];
#pragma unroll
            for (int k = 0; k < 8; k++) e[k] = codes[(u32)(w >> (8 * k)) & 0xFF];
            if (GROUP == 8) {
#pragma unroll
                for (int k = 7; k >= 0; k--) put_e(e[k]);
                flush_group();
            } else {
                put_e(e[7]); put_e(e[6]); put_e(e[5]); put_e(e[4]); flush_group();
                put_e(e[3]); put_e(e[2]); put_e(e[1]); put_e(e[0]); flush_group();
            }
        }
    }
    while (i > 0) {                                               // the ragged start of the stream, four symbols per group
        u32 g = i < 4 ? i : 4;
        for (u32 k = 0; k < g; k++) put_e(codes[src[--i]]);
        flush_group();
    }
    acc |= 1ull << nb; nb++;                                      // final marker bit (nb < 8 before)
    __builtin_memcpy(orow + fill, &acc, 8); fill += (nb + 7) >> 3;
    for (u32 k = lo; k < fill; k++) seg[k] = orow[k];
}

// A stream of a block whose sixteen symbols all have 4-bit codes is a string of nibbles: nibble q of the stream is the code of
// symbol n-1-q (the last symbol is read first, 4.2.2), nibble n the end marker.  Nothing is serial: a lane packs 32 symbols into 16
// bytes, the wave covers 1024 bytes of the stream per round.  (Packed random bases: every block -- the one-lane-per-stream writer
// below then has nothing to do.)
__device__ __forceinline__ void zenc_flat4_stream(u8 *out, const u8 *s, u32 n, const u16 *codes, u32 lane)
{
    const u32 B = (4 * n + 8) >> 3;                                   // bytes of the stream
    for (u32 t0 = lane * 16; t0 < B; t0 += 64 * 16) {
        const u32 q0 = 2 * t0;
        if (q0 + 32 <= n) {
            uint4 v0, v1; __builtin_memcpy(&v0, s + (n - 32 - q0), 16); __builtin_memcpy(&v1, s + (n - 16 - q0), 16);
            // bytes n-32-q0 .. n-1-q0; nibble q0 + i is the code of byte 31 - i
            const u32 wd[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
            u64 lo = 0, hi = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const u32 a = (wd[(31 - i) >> 2] >> (8 * ((31 - i) & 3))) & 0xFFu, b2 = (wd[(15 - i) >> 2] >> (8 * ((15 - i) & 3))) & 0xFFu;
                lo |= (u64)(codes[a] & 0xFu) << (4 * i);
                hi |= (u64)(codes[b2] & 0xFu) << (4 * i);
            }
            const uint4 o4 = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
            __builtin_memcpy(out + t0, &o4, 16);
        } else {
            const u32 t1 = t0 + 16 < B ? t0 + 16 : B;
            for (u32 t = t0; t < t1; t++) {
                const u32 qa = 2 * t, qb = qa + 1;
                const u32 na = qa < n ? (codes[s[n - 1 - qa]] & 0xFu) : (qa == n ? 1u : 0u), nbv = qb < n ? (codes[s[n - 1 - qb]] & 0xFu) : (qb == n ? 1u : 0u);
                out[t] = (u8)(na | (nbv << 4));
            }
        }
    }
}

// LZ-coded blocks (mode[b] != 0) take their literals from L.lits with the plan / codes / tree of those literals (plan1 ...)
// and append the Sequences_Section made by k_lz_seqenc; all other blocks are coded from src as literal-only blocks.
struct ZWriteLz { const u8 *mode; const ZEncPlan *plan1; const u16 *codes1; const u8 *trees1; LzBufs B; u32 not_last; u32 wave_general; u32 frame_split; };   // frame_split: the blocks coded with the frame's code are k_zenc_write<true>'s   // wave_general: blocks of general Huffman codes are k_zenc_write_wave's   // not_last: the frame continues behind these blocks (a shard's part of a frame)
struct ZencJob { const u8 *src; size_t n; u32 nblk, frame_wlog; ZEncPlan *plan; u16 *codes; u8 *trees; u64 *offs; u64 hdr; int with_magic, empty; ZWriteLz L; bool direct; u32 block_bytes; ZencLoc dloc; u64 known_total; int have_total; };   // have_total: the blocks' bytes were read back already (zstd_encode_size)   // dloc.loc != nullptr: the direct blocks' codes are tile-local (k_zenc_write_direct_loc)
// FRAME: the blocks coded with the FRAME's code only (ZENC_FRAME_TREE: nearly every block of a FASTQ's quality and sequence frames) -- one
// table for the workgroup instead of sixteen, 5.6 KiB of LDS instead of 13: twice the wavefronts per CU for a kernel whose lanes each walk
// a stream of 8 K symbols.  The plain instantiation then leaves those blocks alone (L.frame_split).
template <bool FRAME>
__global__ __launch_bounds__(64) void k_zenc_write(const u8 *src, u64 n, u32 nblk, const ZEncPlan *plan, const u16 *codes_g, const u8 *trees,
                                                    const u64 *offs, u8 *dst, u64 frame_hdr, ZWriteLz L)
{
    // 16-bit entries: 8 KiB of tables per workgroup instead of 16 -- LDS is what bounds the waves resident per CU here
    __shared__ __attribute__((aligned(16))) u16 codes[FRAME ? 1 : ZENC_BLOCKS_PER_WG][256];
    __shared__ __attribute__((aligned(16))) u8 orows[64 * ZENC_OROW];
    int lane = threadIdx.x;
    u32 b0 = blockIdx.x * ZENC_BLOCKS_PER_WG;
    auto framed = [&](u32 bb) { return plan[bb].kind == ZK_HUF && plan[bb].frame != 0 && !plan[bb].pad; };
    if (FRAME) {
        const u32 bb = b0 + (u32)lane;
        const u64 fr = __ballot(lane < ZENC_BLOCKS_PER_WG && bb < nblk && framed(bb));
        if (!fr) return;
        const uint4 *g = (const uint4 *)(codes_g + (u64)(b0 + (u32)__ffsll((long long)fr) - 1) * 256);   // (every such block holds the frame's table)
        if (lane < 32) ((uint4 *)codes[0])[lane] = g[lane];
    } else {
    {   // nothing but direct blocks (k_zenc_write_direct's): one look at the sixteen plans instead of three walks over them
        const u32 bb = b0 + (u32)lane;
        const bool other = lane < ZENC_BLOCKS_PER_WG && bb < nblk && !(plan[bb].kind == ZK_HUF && (plan[bb].pad == 2 || L.wave_general) && !(L.mode && L.mode[bb])) && !(L.frame_split && framed(bb));
        if (__ballot(other) == 0) return;
    }
    for (u32 jj = 0; jj < ZENC_BLOCKS_PER_WG; jj++) {               // code tables made by k_zenc_plan: 512 B per block, coalesced
        u32 bb = b0 + jj;
        if (bb >= nblk) break;
        const bool lzb = L.mode && L.mode[bb];
        if ((lzb ? L.plan1[bb].kind : plan[bb].kind) != ZK_HUF || (!lzb && plan[bb].pad == 2)) continue;
        if (L.frame_split && framed(bb)) continue;
        const uint4 *g = (const uint4 *)((lzb ? L.codes1 : codes_g) + (u64)bb * 256);
        if (lane < 32) ((uint4 *)codes[FRAME ? 0 : jj])[lane] = g[lane];
    }
    }
    __syncthreads();
    // blocks of sixteen 4-bit codes: the whole wave, a stream after the other
    if (!FRAME) for (u32 jj = 0; jj < ZENC_BLOCKS_PER_WG; jj++) {
        const u32 bb = b0 + jj;
        if (bb >= nblk) break;
        if (L.mode && L.mode[bb]) continue;
        const ZEncPlan p = plan[bb];
        if (p.kind != ZK_HUF || !p.pad || L.wave_general) continue;   // (wave_general: every Huffman block that is not direct is k_zenc_write_wave's)
        if (p.pad == 2) continue;                                     // a direct block: prefix and streams are k_zenc_write_direct's
        u8 *out = dst + frame_hdr + offs[bb];
        const u64 lo = zenc_block_lo(n, nblk, bb);
        if (lane == 0) { zenc_write_block_prefix(out, p, trees + (u64)bb * ZENC_TREE_SLOT, bb + 1 == nblk && !L.not_last, src[lo]); out[p.csize - 1] = 0; }
        const u32 per = (p.n + 3) / 4;
        u32 o = 3 + p.lhdr + p.tree_bytes + 6;
        for (u32 q = 0; q < 4; q++) { zenc_flat4_stream(out + o, src + lo + (u64)q * per, q < 3 ? per : p.n - 3 * per, codes[FRAME ? 0 : jj], (u32)lane); o += p.ssz[q]; }
    }
    u32 j = lane >> 2, k = lane & 3, b = b0 + j;
    if (b < nblk && (FRAME ? framed(b) : !(L.frame_split && framed(b)))) {
        const bool lzb = L.mode && L.mode[b];
        const u16 *ct = codes[FRAME ? 0 : j];
        u8 *out = dst + frame_hdr + offs[b];
        if (!lzb && plan[b].kind == ZK_HUF && (plan[b].pad || L.wave_general)) { }          // written above, or k_zenc_write_wave's
        else if (!lzb) {
            const ZEncPlan p = plan[b];
            u64 lo = zenc_block_lo(n, nblk, b);
            if (k == 0) zenc_write_block_prefix(out, p, trees + (u64)b * ZENC_TREE_SLOT, b + 1 == nblk && !L.not_last, p.n ? src[lo] : 0);
            if (p.kind == ZK_HUF) {
                u32 per = (p.n + 3) / 4;
                u32 cnt = k < 3 ? per : p.n - 3 * per;
                const u32 o = 3 + p.lhdr + p.tree_bytes + 6 + (k > 0 ? p.ssz[0] : 0u) + (k > 1 ? p.ssz[1] : 0u) + (k > 2 ? p.ssz[2] : 0u);
                if (p.log <= 7) huf_encode_stream_staged<8>(out + o, src + lo + (u64)k * per, cnt, ct, orows + lane * ZENC_OROW);
                else huf_encode_stream_staged<4>(out + o, src + lo + (u64)k * per, cnt, ct, orows + lane * ZENC_OROW);
                if (k == 3) out[p.csize - 1] = 0;                   // Number_of_Sequences = 0
            }
        } else {
            const ZEncPlan p = L.plan1[b];                          // plan of the block's literals (p.n = their number)
            const u8 *lits = L.B.lits + (u64)b * L.B.slot;
            const u32 lsec = lz_lit_section_bytes(p), sbytes = L.B.seq_bytes[b];
            if (k == 0) zenc_write_block_header(out, 2, lsec + sbytes, b + 1 == nblk && !L.not_last);
            if (p.kind == ZK_HUF) {
                if (k == 0) zenc_write_huf_lit_prefix(out + 3, p, L.trees1 + (u64)b * ZENC_TREE_SLOT);
                u32 per = (p.n + 3) / 4;
                u32 cnt = k < 3 ? per : p.n - 3 * per;
                const u32 o = 3 + p.lhdr + p.tree_bytes + 6 + (k > 0 ? p.ssz[0] : 0u) + (k > 1 ? p.ssz[1] : 0u) + (k > 2 ? p.ssz[2] : 0u);
                if (p.log <= 7) huf_encode_stream_staged<8>(out + o, lits + (u64)k * per, cnt, ct, orows + lane * ZENC_OROW);
                else huf_encode_stream_staged<4>(out + o, lits + (u64)k * per, cnt, ct, orows + lane * ZENC_OROW);
            } else if (k == 0) {
                u32 h = zenc_lit_header_raw(out + 3, p.kind == ZK_RLE ? 1u : 0u, p.n);
                if (p.kind == ZK_RLE) out[3 + h] = lits[0];
                else for (u32 i = 0; i < p.n; i++) out[3 + h + i] = lits[i];      // fewer than 64 literals, or incompressible ones
            }
            const u8 *sq = L.B.seqbuf + (u64)b * L.B.slot;
            for (u32 i = k; i < sbytes; i += 4) out[3 + lsec + i] = sq[i];
        }
    }
    // raw blocks: whole-wave copy
    if (!FRAME) for (u32 jj = 0; jj < ZENC_BLOCKS_PER_WG; jj++) {
        u32 bb = b0 + jj;
        if (bb >= nblk) break;
        if ((L.mode && L.mode[bb]) || plan[bb].kind != ZK_RAW) continue;
        u64 lo = zenc_block_lo(n, nblk, bb);
        u32 bn = plan[bb].n;
        const u8 *s = src + lo; u8 *o = dst + frame_hdr + offs[bb] + 3;
        for (u32 i = lane * 8; i + 8 <= bn; i += 64 * 8) st64(o + i, ld64(s + i));
        for (u32 i = (bn & ~7u) + lane; i < bn; i += 64) o[i] = s[i];
    }
}

// Blocks of general Huffman codes a WAVEFRONT per stream (frames where they are few: the blocks of a genome that hold an N or an IUPAC
// code among direct blocks, the ragged last block of any stream).  One lane per stream walks 8 K symbols one after the other -- 0.4 to
// 1 ms however few the blocks, at the end of an ennaf call.  Code lengths add up: a lane sums the lengths of its 1/64 of the symbols,
// a wave scan gives the bit its piece starts at (the LAST symbol is written first, 4.2.2), the pieces are ORed into the stream's
// image in LDS and the image is copied to its place.  Blocks of up to 32 KiB (11 bits x 8 K symbols per stream).
#define ZWW_OBUF 12288u
#define ZWW_BLOCKS 64u
__global__ __launch_bounds__(256) void k_zenc_write_wave(const u8 *src, u64 n, u32 nblk, const ZEncPlan *plan, const u16 *codes_g, const u8 *trees,
                                                         const u64 *offs, u8 *dst, u64 frame_hdr, u32 not_last)
{
    extern __shared__ __attribute__((aligned(16))) u8 zww[];      // the block's code table, then the four streams' images
    // (a workgroup looks at ZWW_BLOCKS plans: a launch of one workgroup per block that mostly returns at once cost 0.1 ms per 150 K blocks)
    __shared__ u64 s_general;
    if (threadIdx.x < 64) {                                       // which of its blocks are this kernel's: one look by one wavefront
        // (block blockIdx.x + t gridDim.x: neighbouring blocks -- a probed MiB of a genome is 32 blocks of sixteen 4-bit codes in a row, and one
        // wavefront writing sixteen of them one after the other was 0.4 ms at the end of every ennaf call -- go to different workgroups)
        const u64 bb = (u64)blockIdx.x + (u64)threadIdx.x * gridDim.x;
        const u64 m = __ballot(bb < nblk && plan[bb].kind == ZK_HUF && plan[bb].pad != 2);   // (direct blocks, Raw and RLE blocks: k_zenc_write(_direct))
        if (threadIdx.x == 0) s_general = m;
    }
    __syncthreads();
    for (u64 todo = s_general; todo; todo &= todo - 1) {
    const u32 b = blockIdx.x + (u32)(__ffsll((long long)todo) - 1) * gridDim.x;
    const ZEncPlan p = plan[b];
    __syncthreads();                                              // (the images of the block before this one have been copied out)
    u16 *codes = (u16 *)zww;
    if (threadIdx.x < 32) ((uint4 *)codes)[threadIdx.x] = ((const uint4 *)(codes_g + (u64)b * 256))[threadIdx.x];
    const u32 k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32 *ob = (u32 *)(zww + 512 + k * ZWW_OBUF);
    const u32 per = (p.n + 3) / 4, cnt = k < 3 ? per : p.n - 3 * per, sbytes = k == 0 ? p.ssz[0] : k == 1 ? p.ssz[1] : k == 2 ? p.ssz[2] : p.ssz[3];
    for (u32 i = lane; i < (sbytes + 11) / 4; i += 64) ob[i] = 0;
    __syncthreads();
    const u64 lo_b = zenc_block_lo(n, nblk, b);
    const u8 *s = src + lo_b + (u64)k * per;
    u8 *out = dst + frame_hdr + offs[b];
    const u32 o = 3 + p.lhdr + p.tree_bytes + 6 + (k > 0 ? p.ssz[0] : 0u) + (k > 1 ? p.ssz[1] : 0u) + (k > 2 ? p.ssz[2] : 0u);
    if (p.pad == 1) {                                             // sixteen 4-bit codes: the stream straight from the symbols, all lanes (zenc_flat4_stream)
        zenc_flat4_stream(out + o, s, cnt, codes, lane);
        if (threadIdx.x == 0) zenc_write_block_prefix(out, p, trees + (u64)b * ZENC_TREE_SLOT, b + 1 == nblk && !not_last, 0);
        if (threadIdx.x == 255) out[p.csize - 1] = 0;
        continue;
    }
    const u32 chunk = (cnt + 63) / 64, lo = lane * chunk < cnt ? lane * chunk : cnt, hi = lo + chunk < cnt ? lo + chunk : cnt;
    // (the piece eight bytes per load: a byte per load was 256 dependent trips to memory per lane)
    u32 bits = 0;
    {
        u32 i = lo;
        for (; i + 8 <= hi; i += 8) { const u64 v = ld64(s + i);
#pragma unroll
            for (u32 q = 0; q < 8; q++) bits += codes[(u32)(v >> (8 * q)) & 0xFFu] >> 12; }
        for (; i < hi; i++) bits += codes[s[i]] >> 12;
    }
    const u32 incl = wave_scan_inclusive<u32, OpAdd>(bits);
    const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
    u32 pos = total - incl;                                       // where the piece's last symbol goes
    u64 acc = 0; u32 nb = pos & 31, w = pos >> 5;
    {
        u32 i = hi;
        for (; i >= lo + 8; i -= 8) { const u64 v = ld64(s + i - 8);
#pragma unroll
            for (int q = 7; q >= 0; q--) {
                const u32 e = codes[(u32)(v >> (8 * q)) & 0xFFu];
                acc |= (u64)(e & 0xFFFu) << nb; nb += e >> 12;
                if (nb >= 32) { atomicOr(&ob[w], (u32)acc); w++; acc >>= 32; nb -= 32; }
            } }
        for (; i-- > lo;) {
            const u32 e = codes[s[i]];
            acc |= (u64)(e & 0xFFFu) << nb; nb += e >> 12;
            if (nb >= 32) { atomicOr(&ob[w], (u32)acc); w++; acc >>= 32; nb -= 32; }
        }
    }
    if (nb) atomicOr(&ob[w], (u32)acc);
    if (lane == 0) atomicOr(&ob[total >> 5], 1u << (total & 31)); // the end mark
    __syncthreads();
    const u8 *img = (const u8 *)ob;
    for (u32 i = lane * 8; i + 8 <= sbytes; i += 64 * 8) st64(out + o + i, ld64(img + i));
    for (u32 i = (sbytes & ~7u) + lane; i < sbytes; i += 64) out[o + i] = img[i];
    if (threadIdx.x == 0) zenc_write_block_prefix(out, p, trees + (u64)b * ZENC_TREE_SLOT, b + 1 == nblk && !not_last, 0);
    if (threadIdx.x == 255) out[p.csize - 1] = 0;                 // Number_of_Sequences = 0
    }
}

// Direct blocks (plan.pad == 2; enc.hip: direct_word): stream q of block b is ready in bytes [4096 q, 4096 q + 4096) of the block's 32 KiB
// of src; it moves to its place behind the block's prefix (written here too) and the end mark follows it.  A workgroup per block, a
// wavefront per stream, a lane's four 16-byte loads in flight together.
__global__ __launch_bounds__(256) void k_zenc_write_direct(const u8 *src, u32 nblk, const ZEncPlan *plan, const u8 *trees, const u64 *offs, u8 *dst, u64 frame_hdr, u32 not_last)
{
    const u32 b = blockIdx.x, q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const ZEncPlan p = plan[b];
    if (p.kind != ZK_HUF || p.pad != 2) return;
    const u32 o = 3 + p.lhdr + p.tree_bytes + 6 + (q > 0 ? p.ssz[0] : 0u) + (q > 1 ? p.ssz[1] : 0u) + (q > 2 ? p.ssz[2] : 0u);   // (no run-time index into the plan: that puts it in scratch memory)
    const uint4 *in = (const uint4 *)(src + ((u64)b << 15) + 4096u * q);
    u8 *out = dst + frame_hdr + offs[b], *so = out + o;
    if (threadIdx.x == 255) { zenc_write_block_prefix(out, p, trees + (u64)b * ZENC_TREE_SLOT, b + 1 == nblk && !not_last, 0); out[p.csize - 1] = 0; }
    uint4 v[4];
#pragma unroll
    for (u32 r = 0; r < 4; r++) v[r] = in[r * 64 + lane];
#pragma unroll
    for (u32 r = 0; r < 4; r++) __builtin_memcpy(so + 16u * (r * 64 + lane), &v[r], 16);   // (stores at aligned addresses with the LOADS taking the misalignment: 1.04 -> 1.26 ms)
    if (lane == 0) so[4096] = 1;
}

// The same blocks when the split pass read its text once (enc.hip: k_enc_fused): the codes wait tile by tile -- base i of tile t in bits
// 2i, 2i + 1 of the KiB at loc + 1024 t -- and are gathered here: the sixteen bases of group G of the stream (bases 16 G ... of the base
// stream) are one 32-bit word wherever they lie, two bits a base, the first lowest; in the stream that word stands with its nibbles
// reversed at place 1023 - (G & 1023) (enc.hip: direct_word).  A lane takes four groups = 64 bases = 128 bits at a time: five words
// of the tile that holds the first of them, shifted down by the two bits per base it starts behind a word's first; where the
// tile ends inside the 64 bases the rest comes from the first sixteen bytes of the next tile, shifted up (a direct block's tiles are
// regular: at least 3972 bases each, so 64 bases touch two tiles at most and a block's 65536 at most eighteen).
__device__ __forceinline__ u32 nibble_rev32(u32 x) { const u32 r = __builtin_bswap32(x); return ((r & 0x0F0F0F0Fu) << 4) | ((r >> 4) & 0x0F0F0F0Fu); }
__global__ __launch_bounds__(256) void k_zenc_write_direct_loc(ZencLoc D, u32 nblk, const ZEncPlan *plan, const u8 *trees, const u64 *offs, u8 *dst, u64 frame_hdr, u32 not_last)
{
    __shared__ i32 s_bnd[4][ZENC_LOC_BND];                        // base counts in front of tiles t0 .. t0 + 19, less the block's first base (k_direct_blocks)
    const u32 b = blockIdx.x, q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // (the three loads are asked for together: the tables are in bounds for every block, meaningful for the direct ones)
    const u64 t0 = D.blk_t0[b];
    const i32 mine = lane < ZENC_LOC_BND ? D.blk_bnd[(u64)b * ZENC_LOC_BND + lane] : 0;
    const ZEncPlan p = plan[b];
    if (p.kind != ZK_HUF || p.pad != 2) return;
    // every wavefront keeps its own copy: no barrier between the table and the loads that need it
    if (lane < ZENC_LOC_BND) s_bnd[q][lane] = mine;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const i32 *bnd = s_bnd[q];
    const i32 b0rel = bnd[0];                                     // <= 0
    uint4 A[4]; u32 E[4], sh[4], avail[4]; u64 N0[4], N1[4];
#pragma unroll
    for (u32 r = 0; r < 4; r++) {
        const u32 cidx = r * 64 + lane, Q = 255u - cidx;              // sixteen bytes of the stream <- groups 4 Q .. 4 Q + 3 of it
        const i32 r0 = (i32)(16384u * q + 64u * Q);                   // the quad's first base, counted from the block's first
        u32 k = (u32)(r0 - b0rel) >> 12;                              // tiles hold at most 4096 bases: not behind this one ...
        if (bnd[k + 1] <= r0) k++;                                    // ... and, as they hold at least 3972, at most two further
        if (bnd[k + 1] <= r0) k++;
        const u32 i = (u32)(r0 - bnd[k]);
        // the 128 bits from bit 2 i of the tile's string on: five words from word i / 16 (one 16-byte load at a 4-byte boundary and a word)
        const u32 *pw = (const u32 *)(D.loc + (t0 + k) * 1024) + (i >> 4);
        __builtin_memcpy(&A[r], pw, 16); E[r] = pw[4];
        sh[r] = 2u * (i & 15u);
        const u32 av = (u32)(bnd[k + 1] - r0);                      // bases of the quad this tile still has (>= 1)
        avail[r] = av < 64u ? av : 64u;
        N0[r] = N1[r] = 0;
        if (av < 64u) { const u8 *nx = D.loc + (t0 + k + 1) * 1024; N0[r] = ld64(nx); N1[r] = ld64(nx + 8); }   // the tile ends inside the quad: the next tile's first bases
    }
    const u32 o = 3 + p.lhdr + p.tree_bytes + 6 + (q > 0 ? p.ssz[0] : 0u) + (q > 1 ? p.ssz[1] : 0u) + (q > 2 ? p.ssz[2] : 0u);
    u8 *out = dst + frame_hdr + offs[b], *so = out + o;
    if (threadIdx.x == 255) { zenc_write_block_prefix(out, p, trees + (u64)b * ZENC_TREE_SLOT, b + 1 == nblk && !not_last, 0); out[p.csize - 1] = 0; }
    uint4 v[4];
#pragma unroll
    for (u32 r = 0; r < 4; r++) {
        const u32 w0 = __builtin_amdgcn_alignbit(A[r].y, A[r].x, sh[r]), w1 = __builtin_amdgcn_alignbit(A[r].z, A[r].y, sh[r]),
                  w2 = __builtin_amdgcn_alignbit(A[r].w, A[r].z, sh[r]), w3 = __builtin_amdgcn_alignbit(E[r], A[r].w, sh[r]);
        u64 a = (u64)w0 | ((u64)w1 << 32), c = (u64)w2 | ((u64)w3 << 32);
        if (avail[r] < 64u) {                                         // the rest from the next tile
            const u64 n0 = N0[r], n1 = N1[r];
            const u32 s = 2u * avail[r];                              // 2 .. 126
            if (s < 64) { const u64 m = (1ull << s) - 1; a = (a & m) | (n0 << s); c = (n1 << s) | (n0 >> (64 - s)); }
            else if (s == 64) { c = n0; }
            else { const u64 m = (1ull << (s - 64)) - 1; c = (c & m) | (n0 << (s - 64)); }
        }
        v[r] = make_uint4(nibble_rev32((u32)(c >> 32)), nibble_rev32((u32)c), nibble_rev32((u32)(a >> 32)), nibble_rev32((u32)a));
    }
#pragma unroll
    for (u32 r = 0; r < 4; r++) __builtin_memcpy(so + 16u * (r * 64 + lane), &v[r], 16);
    if (lane == 0) so[4096] = 1;
}

__global__ void k_zenc_frame_header(u8 *dst, int with_magic, u32 wlog)
{
    if (threadIdx.x || blockIdx.x) return;
    u32 p = 0;
    if (with_magic) { dst[p++] = 0x28; dst[p++] = 0xB5; dst[p++] = 0x2F; dst[p++] = 0xFD; }
    dst[p++] = 0x00;            // Frame_Header_Descriptor: no FCS, no checksum, no dictionary, windowed (like ennaf -1)
    dst[p++] = (u8)((wlog - 10) << 3);   // Window_Descriptor: 2^19 for blocks that never reference earlier data, else the window the matches were found in
}

// ---- is there anything to match?  (level 1) -----------------------------------------------------------------------------------------
// The reference's level 1 finds the repeats of a repeat-rich genome (its archive is then a fraction of the entropy-coded size); matching
// a stream costs about ten times the rest of a level-1 encode, and random bases have nothing to match.  So level 1 LOOKS first: one
// region of 1 MiB in every 64 MiB enters a table of its own (one anchor position in 64: the atomics are what this costs), then every
// anchor of the same regions asks the table whether an earlier position of its region holds the same 16 bytes.  The share of anchors
// that do is read back.
#define PROBE_REGION_LOG 20
#define PROBE_TLOG 16
__global__ __launch_bounds__(256) void k_ldm_probe(const u8 *src, u64 n, u32 *tab, u32 *counts /* per workgroup: anchors | hits << 16 */, int pass, u32 every)
{
    // thread: 8 consecutive positions of a sampled region (grid: regions x 512 workgroups)
    const u64 region = (u64)(blockIdx.x >> 9) * every, rbase = region << PROBE_REGION_LOG;
    const u64 p0 = rbase + ((u64)(blockIdx.x & 511) * 256 + threadIdx.x) * 8;
    u32 anchors = 0, hits = 0;
    if (p0 + 24 <= n) {
        const u64 w0 = ld64(src + p0), w1 = ld64(src + p0 + 8), w2 = ld64(src + p0 + 16);
        u32 *t = tab + ((u64)(blockIdx.x >> 9) << PROBE_TLOG);
#pragma unroll
        for (u32 k = 0; k < 8; k++) {
            const u64 a = k ? (w0 >> (8 * k)) | (w1 << (64 - 8 * k)) : w0, b = k ? (w1 >> (8 * k)) | (w2 << (64 - 8 * k)) : w1;
            u32 idx;
            if (!ldm_key(a, b, PROBE_TLOG, idx) || (ldm_mix(a) >> 58) != 0) continue;     // three more zero bits than k_ldm_insert asks for
            const u32 rel = (u32)(p0 + k - rbase);
            if (pass == 0) atomicMin(&t[idx], rel);
            else {
                anchors++;
                const u32 q = t[idx];
                if (q < rel && rel - q < (1u << 19) && ld64(src + rbase + q) == a && ld64(src + rbase + q + 8) == b) hits++;   // inside level 1's window
            }
        }
    }
    if (pass) {
        __shared__ u32 s_c[4];
        const u32 tot = wg_reduce1<u32, OpAdd>(anchors | (hits << 16), s_c);        // at most 2048 of either per workgroup
        if (threadIdx.x == 0) counts[blockIdx.x] = tot;                // (one word per workgroup, summed by the host: 40 k atomics on one address took a millisecond)
    }
}
// share of the probed anchors with an earlier copy, in 1/1024
int zenc_repeat_probe(naf_gpu_ctx *c, const u8 *d_src, size_t n, u32 *share_1024)
{
    *share_1024 = 0;
    if (n < 65536) return 0;                                      // nothing a second block could refer to
    if (n < (4u << PROBE_REGION_LOG)) { *share_1024 = 1024; return 0; }   // a few megabytes: matching them costs a millisecond or two, just do it
    const u32 every = zenc_probe_every(n);
    const u64 regions = ((n >> PROBE_REGION_LOG) + every - 1) / every;
    const u32 nwg = (u32)(regions * 512);
    u32 *tab = arena_new<u32>(c, regions << PROBE_TLOG), *cnt = arena_new<u32>(c, nwg);
    if (!tab || !cnt) return NAF_GPU_ENOMEM;
    HIP_TRY(c, hipMemsetAsync(tab, 0xFF, (regions << PROBE_TLOG) * 4, c->stream));
    LAUNCH(c, "zenc_probe_insert", k_ldm_probe, nwg, 256, 0, d_src, (u64)n, tab, cnt, 0, every);
    LAUNCH(c, "zenc_probe_count", k_ldm_probe, nwg, 256, 0, d_src, (u64)n, tab, cnt, 1, every);
    std::vector<u32> hc(nwg);
    int rc = ctx_readback(c, hc.data(), cnt, (size_t)nwg * 4); if (rc) return rc;
    u64 h[2] = { 0, 0 };
    for (u32 v : hc) { h[0] += v & 0xFFFF; h[1] += v >> 16; }
    if (h[0]) *share_1024 = (u32)(h[1] * 1024 / h[0]);
    return 0;
}

// Window of the match finder at a compression level: none below 2 (matches inside a block only), else windowLog of libzstd's
// parameter table for inputs above 256 KiB (clevels.h) -- what the reference's streams get from ZSTD_initCStream(level).
int zenc_level_window(int level)
{
    static const unsigned char w[23] = { 0, 0, 20, 21, 21, 21, 21, 21, 21, 21, 22, 22, 22, 22, 22, 22, 22, 23, 23, 23, 25, 26, 27 };
    if (level < 2) return 0;
    return w[level > 22 ? 22 : level];
}

extern "C" size_t naf_gpu_zstd_compress_bound(size_t n)
{
    return n + 3 * (n / 1024 + 2) + 64;                            // blocks are never smaller than 1 KiB (a 2^10 window)
}

// block_log: log2 of the target block size (clamped to 17 = the format maximum of 128 KiB)
// with_magic: 1 = whole frame with its magic number, 0 = whole frame without it (as stored in a .naf section),
//   ZENC_PART | ZENC_PART_FIRST | ZENC_PART_LAST = a shard's part of a frame: blocks only, behind the 2-byte frame header when
//   FIRST, ending the frame when LAST (an empty part that is not LAST is zero bytes; an empty LAST part is one empty Raw block).
int zstd_encode_begin(naf_gpu_ctx *c, const u8 *d_src, size_t n, int level, int with_magic, int lz, int block_log_hint, int window_log, ZencJob **job, const u8 *direct, u32 nd, const ZencLoc *dloc)
{
    ZencJob *J = new ZencJob; *job = J;                          // released by zstd_encode_finish
    memset(J, 0, sizeof *J);
    J->src = d_src; J->n = n;
    const bool part = (with_magic & ZENC_PART) != 0, part_first = (with_magic & ZENC_PART_FIRST) != 0, part_last = (with_magic & ZENC_PART_LAST) != 0;
    const u32 min_gain = (with_magic & ZENC_PREFER_RAW) ? 32u : 0u;
    const bool frame_tree_ok = level <= 1 && (with_magic & ZENC_FRAME_TREE) != 0;
    // (the same streams -- the mask -- keep their codes to 9 bits: this build's decoder then walks them with its single-level table)
    u32 maxbits = (with_magic & ZENC_PREFER_RAW) ? 9u : (with_magic & ZENC_SHORT_CODES) ? 7u : (u32)ZENC_HUF_MAXBITS;
    // ZENC_PREFER_FLAT: blocks of 2^k distinct symbols take k-bit codes unless Huffman coding saves a sixteenth of the block
    // (NAF_GPU_PREFER_FLAT=0: never; =d: the threshold 1/d)
    u32 prefer_flat = (with_magic & ZENC_PREFER_FLAT) ? 16u : 0u;
    if (prefer_flat) { const char *pf = ctx_opt(c, "PREFER_FLAT"); if (pf && pf[0]) prefer_flat = (u32)atoi(pf); }
    with_magic &= ~(ZENC_PREFER_RAW | ZENC_PREFER_FLAT | ZENC_SHORT_CODES | ZENC_FRAME_TREE);
    if (part) with_magic = 0;
    if (part && n == 0 && !part_last) {
        J->empty = 1; J->hdr = part_first ? 2 : 0; J->frame_wlog = (u32)(window_log >= 10 ? window_log : 19);
        return 0;
    }
    u32 block_log = 15;                                          // 32 KiB: 4 streams of 8 KiB; more streams = more decode parallelism
    if (block_log_hint >= 10 && block_log_hint <= 17) block_log = (u32)block_log_hint;
    const char *e = ctx_opt(c, "BLOCK_LOG");
    if (e) { int v = atoi(e); if (v >= 10 && v <= 17) block_log = (u32)v; }
    const char *el = ctx_opt(c, "LZ");                       // "0": never, "all": every stream (tests)
    bool use_lz = lz || level >= 2;
    if (el && !strcmp(el, "0")) use_lz = false;
    if (el && !strcmp(el, "all")) use_lz = true;
    // LZ-coded streams: 8 KiB blocks.  The decoder decodes and executes the sequences of a block serially -- a lane, then a wavefront per
    // block whose LDS buffer is the block's size -- so smaller blocks are more lanes, more workgroups per CU and shorter chains; matches in
    // ids / names / lengths are a few dozen bytes back anyway (a FASTQ's read names: the archive is no larger than with 16 KiB blocks).
    if (use_lz && !e) block_log = (lz && block_log_hint >= 10 && block_log_hint <= 15) ? (u32)block_log_hint : 13u;   // (the hint of a caller that asked for the match finder itself: ennaf's comments and lengths of many records)
    if (use_lz && block_log > 15) block_log = 15;                // LZ lengths and distances are kept in 16 bits
    // cross-block matching (window_log >= 10; the host maps level / --long to it): 64 KiB blocks -- fewer block and table headers,
    // and a repeat is cut less often; lengths still fit 16 bits (k_lzx_parse clamps a match at 65535)
    const bool lzx = use_lz && window_log >= 10 && n >= 64;
    if (window_log > 31) window_log = 31;
    if (lzx) {
        block_log = 16;
        if (block_log > (u32)window_log) block_log = (u32)window_log;          // Block_Maximum_Size is the smaller of Window_Size and 128 KiB (3.1.1.2.4)
    }
    // The window announced in the frame header follows from the OPTIONS alone, not from this part's size: the header is written by the
    // first part of a sharded / chunked frame, and a first part of a few bytes (the ids of a slice that is one chromosome) says
    // nothing about the offsets the later parts use -- the reference's streaming decoder sizes its history from this field
    const u32 frame_wlog = (use_lz && window_log >= 10) ? (u32)window_log : 19u;
    u64 bs = 1ull << block_log;
    u64 nblk64 = n ? (n + bs - 1) / bs : 1;
    if (nblk64 > 0x7FFFFFFFull) return ctx_fail(c, NAF_GPU_EARG, "stream too large");
    u32 nblk = (u32)nblk64;
    u64 hdr = with_magic ? 6 : (part && !part_first) ? 0 : 2;
    ZEncPlan *plan = arena_new<ZEncPlan>(c, nblk);
    u16 *codes = arena_new<u16>(c, (size_t)nblk * 256); u8 *trees = (u8 *)arena_alloc(c, (size_t)nblk * ZENC_TREE_SLOT);
    u64 *offs = arena_new<u64>(c, (size_t)nblk + 2);
    // level 1 writes Huffman weights directly where the format allows it (up to 128 of them): FSE-coding them saves about a
    // dozen bytes per block and costs one lane a serial pass per block
    const u32 try_fse = level >= 2;
    // tree descriptions of ZENC_CACHE_ENTRIES evenly spaced blocks first (worth a launch from a few thousand blocks up)
    const u32 sample_stride = nblk >= 16 * ZENC_CACHE_ENTRIES ? nblk / ZENC_CACHE_ENTRIES : 0;
    ZTreeCache *cache = sample_stride ? arena_new<ZTreeCache>(c, 1) : nullptr;
    if (!plan || !codes || !trees || !offs || (sample_stride && !cache)) return NAF_GPU_ENOMEM;
    if (cache) HIP_TRY(c, hipMemsetAsync(cache, 0, sizeof(ZTreeCache), c->stream));
    // the packed 4-bit stream: blocks of pure A C G T near four bits of entropy are settled by a wave each, without the planner
    u8 *done = nullptr;
    if (direct && (use_lz || block_log != 15 || nblk != nd || n != (size_t)nd << 15 || prefer_flat < 2 || !zenc_flat16().tb))
        return ctx_fail(c, NAF_GPU_EARG, "direct blocks: the stream must be %u blocks of 32 KiB coded without the match finder", nd);
    if (prefer_flat >= 2 && n >= 2048 && zenc_flat16().tb) {
        done = (u8 *)arena_alloc(c, nblk); if (!done) return NAF_GPU_ENOMEM;
        bool dplans = false;
        if (direct) {
            // what k_zenc_flat_scan makes of a direct block, once, on the host
            ZEncPlan pd; memset(&pd, 0, sizeof pd);
            const u32 bn = 32768, perq = bn / 4;
            pd.n = bn; pd.kind = ZK_RAW; pd.csize = 3 + bn;
            for (u32 k = 0; k < 4; k++) pd.ssz[k] = (perq * 4 + 8) >> 3;
            zenc_plan_finish(pd, bn, 4, zenc_flat16().tb, min_gain);
            if (pd.kind == ZK_HUF) {
                pd.pad = 2; dplans = true;
                LAUNCH(c, "zenc_direct_plans", k_zenc_direct_plans, cdiv(nblk, 256), 256, 0, direct, nblk, pd, zenc_flat16(), plan, trees, offs, done);
            }
        }
        LAUNCH(c, "zenc_flat_scan", k_zenc_flat_scan, (dplans && cdiv(nblk, ZENC_FLATSCAN_WAVES) > 16384u) ? 16384u : cdiv(nblk, ZENC_FLATSCAN_WAVES), 64 * ZENC_FLATSCAN_WAVES, 0, d_src, (u64)n, nblk, plan, codes, trees, offs, done, min_gain, prefer_flat, zenc_flat16(), dplans ? direct : (const u8 *)nullptr, direct);
    }
    // the frame's tree (k_zenc_plan): level 1 without the match finder, frames of enough blocks to sample; NAF_GPU_FRAME_TREE=0: a tree per block
    const char *eft = ctx_opt(c, "FRAME_TREE");
    const bool frame_tree = frame_tree_ok && cache && !use_lz && !direct && !(eft && eft[0] == '0');
    u32 *fhist = nullptr; ZEncPlan *fplan = nullptr; u16 *fcodes = nullptr; u8 *ftree = nullptr;
    if (frame_tree) {
        fhist = arena_new<u32>(c, 256 + 2); fplan = arena_new<ZEncPlan>(c, 1); fcodes = arena_new<u16>(c, 256); ftree = (u8 *)arena_alloc(c, ZENC_TREE_SLOT);
        if (!fhist || !fplan || !fcodes || !ftree) return NAF_GPU_ENOMEM;
        HIP_TRY(c, hipMemsetAsync(fhist, 0, 258 * 4, c->stream));
        HIP_TRY(c, hipMemsetAsync(fplan, 0, sizeof(ZEncPlan), c->stream));
        HIP_TRY(c, hipMemsetAsync(fcodes, 0, 512, c->stream));
    }
    if (cache) LAUNCH(c, "zenc_plan_sample", k_zenc_plan, ZENC_CACHE_ENTRIES, 256, 0, d_src, (u64)n, nblk, plan, codes, trees, offs, (const u32 *)nullptr, (u64)0, cache, sample_stride, try_fse, min_gain, maxbits, prefer_flat, (const u8 *)done, (u8 *)nullptr, fhist, (const ZEncPlan *)nullptr, (const u16 *)nullptr);
    if (frame_tree) {
        u8 *fblock = (u8 *)arena_alloc(c, ZENC_FRAME_BLOCK + 64); if (!fblock) return NAF_GPU_ENOMEM;
        LAUNCH(c, "zenc_frame_block", k_zenc_frame_block, 1, 256, 0, (const u32 *)fhist, fblock, fhist + 256);
        // (no flat preference: the code of the summed histogram as it is, with its tree description -- FSE-coded by k_zenc_tree where
        // the weights reach past symbol 127, a mask stream's units, like any other block's)
        u8 *fwt = (u8 *)arena_alloc(c, 256); if (!fwt) return NAF_GPU_ENOMEM;
        LAUNCH(c, "zenc_plan_frame", k_zenc_plan, 1, 256, 0, (const u8 *)fblock, (u64)0, 1u, fplan, fcodes, ftree, (u64 *)nullptr, (const u32 *)(fhist + 256), (u64)0, (ZTreeCache *)nullptr, 0u, try_fse, 0u, maxbits, 0u, (const u8 *)nullptr, fwt, (u32 *)nullptr, (const ZEncPlan *)nullptr, (const u16 *)nullptr);
        LAUNCH(c, "zenc_tree", k_zenc_tree, 1, 64, 64 * sizeof(ZTreeLane), 1u, fplan, ftree, (u64 *)nullptr, (const u8 *)fwt, 0u);
        LAUNCH(c, "zenc_frame_ratio", k_zenc_frame_ratio, 1, 256, 0, fhist, (const u16 *)fcodes);
        // most blocks of such a frame without a histogram (k_zenc_frame_quick); the planner below takes what that leaves
        const char *fq = ctx_opt(c, "FRAME_QUICK");
        if (!(fq && fq[0] == '0')) {
            float *fstat = arena_new<float>(c, 8); if (!fstat) return NAF_GPU_ENOMEM;
            if (!done) { done = (u8 *)arena_alloc(c, nblk); if (!done) return NAF_GPU_ENOMEM; HIP_TRY(c, hipMemsetAsync(done, 0, nblk, c->stream)); }
            LAUNCH(c, "zenc_frame_stats", k_zenc_frame_stats, 1, 256, 0, (const u32 *)fhist, (const u16 *)fcodes, fstat);
            u32 *tally = arena_new<u32>(c, 2); if (!tally) return NAF_GPU_ENOMEM;
            HIP_TRY(c, hipMemsetAsync(tally, 0, 8, c->stream));
            const u32 qs = nblk / 512 ? nblk / 512 : 1u;                      // (a dry run over ~512 blocks decides whether the pass over all of them is worth its read)
            LAUNCH(c, "zenc_frame_quick", k_zenc_frame_quick, cdiv(nblk, qs), 256, 0, d_src, (u64)n, nblk, plan, codes, offs, done, (const ZEncPlan *)fplan, (const u16 *)fcodes, (const float *)fstat, min_gain, qs, tally, (const u32 *)nullptr);
            LAUNCH(c, "zenc_frame_quick", k_zenc_frame_quick, nblk, 256, 0, d_src, (u64)n, nblk, plan, codes, offs, done, (const ZEncPlan *)fplan, (const u16 *)fcodes, (const float *)fstat, min_gain, 0u, (u32 *)nullptr, (const u32 *)tally);
        }
    }
    // the match finder's buffers; and the blocks of zero-terminated names a lane per line settles (k_lz_parse_lines) in FRONT of the planner:
    // a block that kernel takes is coded with its matches or, should that come out larger, Raw -- the planner's histogram of its bytes, code
    // and tree (the literal-only coding k_lz_choose would weigh against the matches) are not made at all
    LzBufs B; memset(&B, 0, sizeof B);
    u8 *lz_fallback = nullptr;
    const u32 lz_buf = (u32)((bs + 1 + 320 + 15) & ~15ull);               // a block of the even split holds at most bs bytes
    if (use_lz && n >= 64) {
        B.slot = bs; B.seq_slot = bs / 4 + 2; B.of = nullptr; B.ofv = nullptr;
        B.lits = (u8 *)arena_alloc(c, (size_t)nblk * B.slot + 64); B.seqbuf = (u8 *)arena_alloc(c, (size_t)nblk * B.slot + 64);
        B.ll = arena_new<u16>(c, (size_t)nblk * B.seq_slot + 8); B.ml = arena_new<u16>(c, (size_t)nblk * B.seq_slot + 8);   // + 8: read in groups of eight
        if (lzx) B.ofv = arena_new<u32>(c, (size_t)nblk * B.seq_slot + 8); else B.of = arena_new<u16>(c, (size_t)nblk * B.seq_slot + 8);
        B.nseq = arena_new<u32>(c, nblk); B.nlit = arena_new<u32>(c, nblk); B.seq_bytes = arena_new<u32>(c, nblk);
        if (!B.lits || !B.seqbuf || !B.ll || !B.ml || (!B.of && !B.ofv) || !B.nseq || !B.nlit || !B.seq_bytes) return NAF_GPU_ENOMEM;
        const char *ll_ = ctx_opt(c, "LZ_LINES");
        if (!lzx && !(ll_ && ll_[0] == '0') && bs <= 32768 && nblk >= 64) {
            lz_fallback = (u8 *)arena_alloc(c, nblk); if (!lz_fallback) return NAF_GPU_ENOMEM;
            if (!done) { done = (u8 *)arena_alloc(c, nblk); if (!done) return NAF_GPU_ENOMEM; HIP_TRY(c, hipMemsetAsync(done, 0, nblk, c->stream)); }
            const u32 line_div = bs > 16384 ? 16u : 8u, ml_slots = 2 * (u32)(bs / line_div) + 4;   // (at most two sequences a line, a line more than the terminators)
            const u32 lines_lds = lz_buf + 4 * ml_slots + 2 * (u32)(bs / line_div + 2);              // 19 KiB for blocks of 8 KiB, 53 for 32
            LAUNCH(c, "zenc_lz_lines", k_lz_parse_lines, nblk, 64, lines_lds, d_src, (u64)n, nblk, B, lz_buf, lz_fallback, plan, offs, done, line_div, ml_slots);
        }
    }
    // (frames of a few blocks keep the tree with the planner: nothing to gain from a second launch)
    const char *td = ctx_opt(c, "TREE_DEFER");
    u8 *wt_defer = (nblk >= 256 && !(td && td[0] == '0')) ? (u8 *)arena_alloc(c, (size_t)nblk * 256) : nullptr;
    LAUNCH(c, "zenc_plan", k_zenc_plan, nblk, 256, 0, d_src, (u64)n, nblk, plan, codes, trees, offs, (const u32 *)nullptr, (u64)0, cache, 0u, try_fse, min_gain, maxbits, prefer_flat, (const u8 *)done, wt_defer, (u32 *)nullptr, (const ZEncPlan *)fplan, (const u16 *)fcodes, (const u32 *)(fhist ? fhist + 257 : nullptr));
    if (wt_defer) LAUNCH(c, "zenc_tree", k_zenc_tree, cdiv(nblk, 64), 64, 64 * sizeof(ZTreeLane), nblk, plan, trees, offs, (const u8 *)wt_defer, min_gain);
    if (frame_tree) {
        i32 *fv = arena_new<i32>(c, (size_t)nblk + 1); if (!fv) return NAF_GPU_ENOMEM;
        LAUNCH(c, "zenc_frame_class", k_zenc_frame_class, cdiv(nblk, 256), 256, 0, (const ZEncPlan *)plan, nblk, fv);
        int rcs = scan_inclusive_max_i32(c, fv, nblk); if (rcs) return rcs;
        LAUNCH(c, "zenc_frame_fix", k_zenc_frame_fix, cdiv(nblk, 256), 256, 0, plan, nblk, (const i32 *)fv, (const ZEncPlan *)fplan, (const u8 *)ftree, trees, offs);
    }
    ZWriteLz &L = J->L;
    L.not_last = part && !part_last;
    { const char *fs = ctx_opt(c, "FRAME_WRITE"); L.frame_split = (frame_tree && !(fs && fs[0] == '0')) ? 1u : 0u; }   // (NAF_GPU_FRAME_WRITE=0: every block by the sixteen-table writer)
    if (use_lz && n >= 64) {
        if (!c->d_seqctab) {
            SeqCTabs T; zenc_build_predefined(T);
            HIP_TRY(c, hipMalloc(&c->d_seqctab, sizeof T));
            HIP_TRY(c, hipMemcpy(c->d_seqctab, &T, sizeof T, hipMemcpyHostToDevice));
        }
        ZEncPlan *plan1 = arena_new<ZEncPlan>(c, nblk);
        u16 *codes1 = arena_new<u16>(c, (size_t)nblk * 256); u8 *trees1 = (u8 *)arena_alloc(c, (size_t)nblk * ZENC_TREE_SLOT);
        u8 *mode = (u8 *)arena_alloc(c, nblk);
        if (!plan1 || !codes1 || !trees1 || !mode) return NAF_GPU_ENOMEM;
        if (lzx) {
            // table of first occurrences: one in eight positions is an anchor, epochs of half the window, load factor <= 1/2
            LdmTab T; T.wlog = (u32)window_log; T.elog = T.wlog - 1; T.pad = 0;
            const u64 E = 1ull << T.elog, span = n < E ? n : E;
            u32 tl = 4; while ((1ull << tl) < span / 4 + 16) tl++;
            T.tlog = tl;
            const u64 n_ep = (n >> T.elog) + 1, entries = n_ep << T.tlog;
            T.tab = arena_new<u32>(c, entries); if (!T.tab) return NAF_GPU_ENOMEM;
            HIP_TRY(c, hipMemsetAsync(T.tab, 0xFF, entries * 4, c->stream));
            LAUNCH(c, "zenc_ldm_insert", k_ldm_insert, cdiv(cdiv(n, 8), 256), 256, 0, d_src, (u64)n, T);
            HIP_TRY(c, hipFuncSetAttribute((const void *)k_lzx_parse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lz_buf + (2u << LZ_HASH_LOG))));   // above 64 KiB
            LAUNCH(c, "zenc_lz_parse", k_lzx_parse, nblk, 64, lz_buf + (2u << LZ_HASH_LOG), d_src, (u64)n, nblk, B, lz_buf, T);
            LAUNCH(c, "zenc_lz_seqenc", k_lzx_seqenc, nblk, 64, 0, nblk, B, (const SeqCTabs *)c->d_seqctab);
        } else {
            // the table's size is LDS a wavefront holds for the whole block: what bounds the blocks in flight per CU -- and what is left
            // of a CU for the kernels of the other streams (2^11 and 2^10 entries measured: DESIGN.md section 8)
            const u32 hash_log = LZ_HASH_LOG;
            // (names against the name in front, a lane each: k_lz_parse_lines above; what that left to the hash table's walk is marked in `lz_fallback`)
            LAUNCH(c, "zenc_lz_parse", k_lz_parse, nblk, 64, lz_buf + (2u << hash_log), d_src, (u64)n, nblk, B, lz_buf, hash_log, (const u8 *)lz_fallback);
            LAUNCH(c, "zenc_lz_seqenc", k_lz_seqenc, cdiv(nblk, 64), 64, 0, nblk, B, (const SeqCTabs *)c->d_seqctab);
        }
        LAUNCH(c, "zenc_plan", k_zenc_plan, nblk, 256, 0, (const u8 *)B.lits, (u64)n, nblk, plan1, codes1, trees1, (u64 *)nullptr, (const u32 *)B.nlit, B.slot, (ZTreeCache *)nullptr, 0u, try_fse, min_gain, maxbits, 0u, (const u8 *)nullptr, wt_defer);
        if (wt_defer) LAUNCH(c, "zenc_tree", k_zenc_tree, cdiv(nblk, 64), 64, 64 * sizeof(ZTreeLane), nblk, plan1, trees1, (u64 *)nullptr, (const u8 *)wt_defer, min_gain);
        LAUNCH(c, "zenc_lz_choose", k_lz_choose, cdiv(nblk, 256), 256, 0, nblk, (const ZEncPlan *)plan, (const ZEncPlan *)plan1, (const u32 *)B.nseq, (const u32 *)B.seq_bytes, mode, offs);
        L.mode = mode; L.plan1 = plan1; L.codes1 = codes1; L.trees1 = trees1; L.B = B;
    }
    int rc = scan_exclusive_u64(c, offs, nblk, offs + nblk + 1); if (rc) return rc;
    J->direct = direct != nullptr;
    if (direct && dloc) J->dloc = *dloc;
    J->block_bytes = (u32)bs; J->nblk = nblk; J->plan = plan; J->codes = codes; J->trees = trees; J->offs = offs; J->hdr = hdr; J->with_magic = with_magic; J->frame_wlog = frame_wlog;
    return 0;
}

// The second half: the frame's size is read back (before the write when a hook places the frame, after it otherwise) and the blocks
// are written.
static int zenc_finish(naf_gpu_ctx *c, ZencJob *J, u8 *d_dst, size_t cap, size_t *out_len, const ZencPlace *place);
void zstd_encode_drop(ZencJob *J) { delete J; }
int zstd_encode_finish(naf_gpu_ctx *c, ZencJob *J, u8 *d_dst, size_t cap, size_t *out_len, const ZencPlace *place)
{
    if (!J) return NAF_GPU_EARG;
    int rc = zenc_finish(c, J, d_dst, cap, out_len, place);
    delete J;
    return rc;
}
static int zenc_finish(naf_gpu_ctx *c, ZencJob *J, u8 *d_dst, size_t cap, size_t *out_len, const ZencPlace *place)
{
    const u64 hdr = J->hdr; const u32 nblk = J->nblk;
    if (J->empty) {
        if (place && !(d_dst = place->fn(place->ud, hdr))) return NAF_GPU_ECAP;
        if (!place && cap < hdr) return ctx_fail(c, NAF_GPU_ECAP, "zstd_compress capacity %zu too small", cap);
        if (hdr) LAUNCH(c, "zenc_frame_header", k_zenc_frame_header, 1, 64, 0, d_dst, 0, J->frame_wlog);
        *out_len = hdr;
        return 0;
    }
    if (!place && cap < hdr + J->n + 3ull * nblk) return ctx_fail(c, NAF_GPU_ECAP, "zstd_compress capacity %zu too small (bound %llu)", cap, (unsigned long long)(hdr + J->n + 3ull * nblk));
    int rc;
    u64 total = 0;
    if (place) {
        if (J->have_total) total = J->known_total;
        else if ((rc = ctx_readback(c, &total, J->offs + nblk + 1, 8))) return rc;
        if (!(d_dst = place->fn(place->ud, hdr + total))) return NAF_GPU_ECAP;
    }
    if (hdr) LAUNCH(c, "zenc_frame_header", k_zenc_frame_header, 1, 64, 0, d_dst, J->with_magic, J->frame_wlog);
    {
        // few blocks of general codes (a frame of direct blocks, a short frame): a wavefront per stream for those
        J->L.wave_general = (!J->L.mode && J->block_bytes <= 32768u && (J->direct || nblk <= 2048u)) ? 1u : 0u;
    }
    if (J->L.frame_split) LAUNCH(c, "zenc_write", k_zenc_write<true>, cdiv(nblk, ZENC_BLOCKS_PER_WG), 64, 0, J->src, (u64)J->n, nblk, (const ZEncPlan *)J->plan, (const u16 *)J->codes, (const u8 *)J->trees,
           (const u64 *)J->offs, d_dst, hdr, J->L);
    LAUNCH(c, "zenc_write", k_zenc_write<false>, cdiv(nblk, ZENC_BLOCKS_PER_WG), 64, 0, J->src, (u64)J->n, nblk, (const ZEncPlan *)J->plan, (const u16 *)J->codes, (const u8 *)J->trees,
           (const u64 *)J->offs, d_dst, hdr, J->L);
    // (a frame of direct blocks whose size is known already -- nothing of this stream is in flight: the few blocks of general codes beside the gather,
    // on the look's idle stream, instead of 60 us in front of it)
    naf_gpu_ctx *aux = (J->L.wave_general && J->direct && J->have_total && c->side3 && c->split_ev[ZSPLIT_MAX - 1]) ? c->side3 : nullptr;
    if (aux && hipStreamWaitEvent(aux->stream, c->fork_ev, 0) != hipSuccess) aux = nullptr;
    if (aux) {
        hipLaunchKernelGGL(k_zenc_write_wave, dim3(cdiv(nblk, ZWW_BLOCKS)), dim3(256), 512u + 4u * ZWW_OBUF, aux->stream, J->src, (u64)J->n, nblk, (const ZEncPlan *)J->plan, (const u16 *)J->codes, (const u8 *)J->trees,
                           (const u64 *)J->offs, d_dst, hdr, J->L.not_last);
        if (hipGetLastError() != hipSuccess || hipEventRecord(c->split_ev[ZSPLIT_MAX - 1], aux->stream) != hipSuccess) return ctx_fail(c, NAF_GPU_EHIP, "zenc_write_wave beside the gather: launch failed");
    } else
    if (J->L.wave_general) LAUNCH(c, "zenc_write_wave", k_zenc_write_wave, cdiv(nblk, ZWW_BLOCKS), 256, 512u + 4u * ZWW_OBUF, J->src, (u64)J->n, nblk, (const ZEncPlan *)J->plan, (const u16 *)J->codes, (const u8 *)J->trees,
           (const u64 *)J->offs, d_dst, hdr, J->L.not_last);
    if (J->direct && J->dloc.loc) LAUNCH(c, "zenc_write_direct", k_zenc_write_direct_loc, nblk, 256, 0, J->dloc, nblk, (const ZEncPlan *)J->plan, (const u8 *)J->trees, (const u64 *)J->offs, d_dst, hdr, J->L.not_last);
    else if (J->direct) LAUNCH(c, "zenc_write_direct", k_zenc_write_direct, nblk, 256, 0, J->src, nblk, (const ZEncPlan *)J->plan, (const u8 *)J->trees, (const u64 *)J->offs, d_dst, hdr, J->L.not_last);
    if (aux && hipStreamWaitEvent(c->stream, c->split_ev[ZSPLIT_MAX - 1], 0) != hipSuccess) return ctx_fail(c, NAF_GPU_EHIP, "zenc_write_wave beside the gather: join failed");
    if (!place) { if (J->have_total) total = J->known_total; else if ((rc = ctx_readback(c, &total, J->offs + nblk + 1, 8))) return rc; }
    *out_len = hdr + total;
    return 0;
}
// The size of the frame a begun job will write (its header and its blocks), read back ONCE and kept for zstd_encode_finish: a caller that
// places several frames one behind the other learns all their sizes first and then writes them side by side (enc.hip: ennaf_whole).
// (for a caller that reads several jobs' totals back with one copy: where a job's total lies -- null when there is nothing to read -- and the way to hand it in)
const u64 *zstd_encode_total_ptr(const ZencJob *J) { return (!J || J->empty || J->have_total) ? nullptr : J->offs + J->nblk + 1; }
void zstd_encode_set_total(ZencJob *J, u64 total) { if (J && !J->empty) { J->known_total = total; J->have_total = 1; } }
int zstd_encode_size(naf_gpu_ctx *c, ZencJob *J, size_t *frame_len)
{
    if (!J || !frame_len) return NAF_GPU_EARG;
    if (J->empty) { *frame_len = J->hdr; return 0; }
    if (!J->have_total) {
        u64 t = 0; int rc = ctx_readback(c, &t, J->offs + J->nblk + 1, 8); if (rc) return rc;
        J->known_total = t; J->have_total = 1;
    }
    *frame_len = J->hdr + J->known_total;
    return 0;
}

int zstd_encode(naf_gpu_ctx *c, const u8 *d_src, size_t n, int level, u8 *d_dst, size_t cap, size_t *out_len, int with_magic, int lz, int block_log_hint, int window_log, const ZencPlace *place)
{
    ZencJob *J = nullptr;
    int rc = zstd_encode_begin(c, d_src, n, level, with_magic, lz, block_log_hint, window_log, &J);
    if (rc) { delete J; return rc; }
    return zstd_encode_finish(c, J, d_dst, cap, out_len, place);
}

extern "C" int naf_gpu_zstd_compress(naf_gpu_ctx *c, const void *d_src, size_t n, int level, void *d_dst, size_t cap, size_t *out_len)
{
    if (!c || !d_dst || !out_len || (!d_src && n)) return NAF_GPU_EARG;
    arena_reset(c);
    return zstd_encode(c, (const u8 *)d_src, n, level, (u8 *)d_dst, cap, out_len, 1, 0, 0, zenc_level_window(level));
}
