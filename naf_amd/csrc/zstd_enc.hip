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

// even split of n bytes into nblk blocks: block b starts at b*(n/nblk) + min(b, n%nblk)
__host__ __device__ static inline u64 zenc_block_lo(u64 n, u32 nblk, u32 b) { u64 q = n / nblk, r = n % nblk; return (u64)b * q + (b < r ? b : r); }

__global__ __launch_bounds__(256) void k_zenc_plan(const u8 *src, u64 n, u32 nblk, ZEncPlan *plan, u8 *lens, u8 *trees, u64 *csize)
{
    __shared__ u32 hist[1024];
    u32 b = blockIdx.x;
    u64 lo = zenc_block_lo(n, nblk, b), hi = zenc_block_lo(n, nblk, b + 1);
    u32 bn = (u32)(hi - lo);
    for (u32 i = threadIdx.x; i < 1024; i += 256) hist[i] = 0;
    __syncthreads();
    u32 per = (bn + 3) / 4; if (!per) per = 1;
    const u8 *s = src + lo;
    for (u32 i = threadIdx.x * 4; i < bn; i += 1024) {
        // 4 consecutive bytes per thread; quarter index per byte
#pragma unroll
        for (u32 k = 0; k < 4; k++) if (i + k < bn) { u32 q = (i + k) / per; if (q > 3) q = 3; atomicAdd(&hist[q * 256 + s[i + k]], 1u); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ZEncPlan p; u8 len[256]; u8 tree[ZENC_TREE_SLOT];
        zenc_plan_block(hist, bn, p, len, tree);
        plan[b] = p; csize[b] = p.csize;
        if (p.kind == ZK_HUF) {
            for (u32 i = 0; i < 256; i++) lens[(u64)b * 256 + i] = len[i];
            for (u32 i = 0; i < p.tree_bytes; i++) trees[(u64)b * ZENC_TREE_SLOT + i] = tree[i];
        }
    }
}

#define ZENC_BLOCKS_PER_WG 16
__global__ __launch_bounds__(64) void k_zenc_write(const u8 *src, u64 n, u32 nblk, const ZEncPlan *plan, const u8 *lens, const u8 *trees,
                                                    const u64 *offs, u8 *dst, u64 frame_hdr)
{
    __shared__ u32 codes[ZENC_BLOCKS_PER_WG][256];
    int lane = threadIdx.x;
    u32 b0 = blockIdx.x * ZENC_BLOCKS_PER_WG;
    // code tables: one lane per block assigns canonical codes (serial over 256 symbols)
    if (lane < ZENC_BLOCKS_PER_WG && b0 + lane < nblk && plan[b0 + lane].kind == ZK_HUF) {
        u32 b = b0 + lane; u8 len[256]; u16 code[256];
        for (u32 i = 0; i < 256; i++) len[i] = lens[(u64)b * 256 + i];
        huf_assign_codes(len, plan[b].log, code);
        for (u32 i = 0; i < 256; i++) codes[lane][i] = code[i] | ((u32)len[i] << 16);
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
            huf_encode_stream(out + o, src + lo + (u64)k * per, cnt, codes[j]);
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
    u8 *lens = (u8 *)arena_alloc(c, (size_t)nblk * 256), *trees = (u8 *)arena_alloc(c, (size_t)nblk * ZENC_TREE_SLOT);
    u64 *offs = arena_new<u64>(c, (size_t)nblk + 2);
    if (!plan || !lens || !trees || !offs) return NAF_GPU_ENOMEM;
    LAUNCH(c, "zenc_plan", k_zenc_plan, nblk, 256, 0, d_src, (u64)n, nblk, plan, lens, trees, offs);
    int rc = scan_exclusive_u64(c, offs, nblk, offs + nblk + 1); if (rc) return rc;
    LAUNCH(c, "zenc_frame_header", k_zenc_frame_header, 1, 64, 0, d_dst, with_magic);
    LAUNCH(c, "zenc_write", k_zenc_write, cdiv(nblk, ZENC_BLOCKS_PER_WG), 64, 0, d_src, (u64)n, nblk, (const ZEncPlan *)plan, (const u8 *)lens, (const u8 *)trees,
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
