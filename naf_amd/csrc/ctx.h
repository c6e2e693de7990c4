// ctx.h -- internal context of libnaf_gpu: stream, scratch arena, error text, per-kernel timing.
#pragma once
#include "common.h"
#include "../../include/naf_gpu.h"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdarg.h>
#include <vector>
#include <functional>
#include <string>

struct ArenaChunk { u8 *base; size_t cap, used; };

struct KTime { const char *name; hipEvent_t a, b; };

// Decode -> emit pipeline of a whole-text call: the Huffman literals of a literal-only frame are decoded in `parts` launches over
// consecutive block ranges with an event after each, so that the emit of the text behind a finished part can run on a second
// stream beside the decode of the next part.  out_end[k] = bytes of the decoded stream complete once ev[k] has fired.
#define ZSPLIT_MAX 8
struct ZSplit { int parts; hipEvent_t ev[ZSPLIT_MAX]; u64 out_end[ZSPLIT_MAX]; int done; void *status; };   // status: the decoder's device-side ZStat, read by the caller AFTER it has queued the emit (no host wait between decode and emit)
// A literal-only frame whose blocks all use ONE flat 4-bit tree (packed random ACGT) is not decoded at all: symbol k of a stream
// sits at a known bit position, so the emit kernel takes its bases straight from the compressed stream (k_emit_tile_flat) and the
// packed intermediate -- 5 GB written and read back per 10 GB of text -- never exists.  The decoder fills this when asked to and
// the frame qualifies: per (block, stream) slot the packed offset of its first symbol and the bit address of the end of its data.
struct FlatStream { u64 q0, A; };
// A frame where most, not all, blocks carry the flat tree (a real genome coded with the encoder's flat preference: runs of N, IUPAC
// codes, a block with a skewed pair histogram here and there): `cls` holds one byte per block -- bit 0: its streams can be read in
// place, bit 1: its bytes were decoded into the packed stream -- and the blocks that are not flat, with their two neighbours, ARE
// decoded (at their natural offsets of the packed stream), so that every 4 KiB tile of text lies either over blocks that can be
// read in place or over blocks that were decoded.  The slots of a block that cannot be read in place carry A = FLAT_DECODED.
#define FLAT_DECODED (~0ull)
struct ZFlat { const u8 *src; const FlatStream *si; u64 nslots; const u8 *sym; void *status; bool ready;
               const u8 *tail; u64 tail_q; u32 tail_n;
               const u8 *cls; u32 n_decoded; u32 n_walk;   // n_walk: decoded blocks whose literals need the Huffman walk (a latency-bound job beside the emit)
                     // cls == nullptr: every block is flat (nothing was decoded)
               struct naf_gpu_ctx *aux; hipEvent_t decoded_ev; void *later; };   // later: the decode of the blocks that are not flat, as a job to be run by the caller once its tile index is queued (zstd_flat_later)   // aux: a context whose stream is free for the decode of the blocks that are not flat (set by the caller; the emit of the flat tiles runs beside it); decoded_ev: set by the decoder when it used it -- to be waited for before the decoded bytes are read   // tail: a final Raw block (the byte that holds the padding nibble of an odd stream, zstd_enc) -- its bytes lie in the frame as they are; packed index of its first byte; its length
// The NAF_GPU_* switches (cross-check levers of the tests, development aids).  naf_gpu_init reads them from
// the environment ONCE; afterwards only naf_gpu_set_option changes them -- no call looks at the environment.  Names are kept without
// the NAF_GPU_ prefix.  Side contexts look at the options of the context they belong to (`root`).  TRACE=1: the verdicts of the paths a
// call took are kept as text for naf_gpu_get_trace (the tests ask which path ran; nothing is ever printed by the library).
struct CtxOpts { std::vector<std::pair<std::string, std::string>> kv; std::string trace; bool tracing = false; };
struct naf_gpu_ctx {
    int device = 0;
    CtxOpts *opts = nullptr;              // owned by the context naf_gpu_init returned
    naf_gpu_ctx *root = nullptr;          // side contexts: that context
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::vector<ArenaChunk> chunks;       // scratch arena; consolidated into one chunk at reset
    size_t high_water = 0;
    u8 *h_stage = nullptr;                // pinned host staging for small readbacks
    size_t h_stage_cap = 0;
    void *d_predef = nullptr;             // predefined LL/OF/ML FSE tables (RFC 8878 3.1.1.3.2.2)
    void *d_seqctab = nullptr;            // their encoding counterparts (SeqCTabs), made on first use
    char err[512] = {0};
    bool timing = false;
    std::vector<KTime> ktimes;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    // result of the last get_timing aggregation
    std::vector<std::string> agg_names; std::vector<float> agg_ms; std::vector<int> agg_n;
    float stream_ms[5] = { 0, 0, 0, 0, 0 };    // kernel time per stream of that aggregation (naf_gpu_get_timing_streams)
    // second context (own stream, arena, staging) for the host thread that prepares the side streams of an archive while
    // this one decodes the sequence stream; nullptr inside the side context itself
    naf_gpu_ctx *side = nullptr;
    naf_gpu_ctx *side2 = nullptr;         // third one: the quality stream of a FASTQ archive decodes beside the sequence stream
    naf_gpu_ctx *side3 = nullptr;         // fourth: ids and names beside lengths and mask
    naf_gpu_ctx *side4 = nullptr;         // fifth: the names of an archive of many records beside its ids (emit.hip: unnaf_sections_main)
    void *sides_thread = nullptr; int sides_rc = 0;     // naf_gpu_init's thread that makes the side contexts (naf_gpu.hip: ctx_sides_ready)
    hipEvent_t fork_ev = nullptr;
    struct ZSplit *zsplit = nullptr;        // set by unnaf for the sequence stream of a whole-text call: Huffman literals in parts (below)
    hipEvent_t split_ev[ZSPLIT_MAX + 2] = {};
    struct ZFlat *zflat = nullptr;          // set by unnaf when its emit can read a flat frame in place (above)
    void *io_pool = nullptr;                // io.hip: pinned staging lanes of naf_gpu_read_file / naf_gpu_write_file
    void *shard_state = nullptr;            // enc.hip: what naf_gpu_ennaf_shard_begin leaves for naf_gpu_ennaf_shard_finish
    struct HostWorker *worker = nullptr;    // naf_gpu.hip: the host thread that drives this (side) context, kept between calls
};

// A side context's host thread: started on first use and parked between jobs (creating a thread and its HIP state per call cost
// more than the chains it ran).  One job at a time; ctx_worker_join returns once the job has.
void ctx_worker_start(naf_gpu_ctx *x, std::function<void()> job);
void ctx_worker_join(naf_gpu_ctx *x);
int  ctx_sides_ready(naf_gpu_ctx *c);       // before c->side .. c->side4 are looked at: waits for the thread of naf_gpu_init that makes them

int  ctx_fail(naf_gpu_ctx *c, int code, const char *fmt, ...);
const char *ctx_opt(const naf_gpu_ctx *c, const char *name);           // value of switch NAF_GPU_<name>, nullptr when not set
static inline bool ctx_opt_is(const naf_gpu_ctx *c, const char *name, char v) { const char *e = ctx_opt(c, name); return e && e[0] == v; }
bool ctx_tracing(const naf_gpu_ctx *c);
void ctx_trace(naf_gpu_ctx *c, const char *fmt, ...);                  // a path's verdict, kept when TRACE is on
#define HIP_TRY(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return ctx_fail((c), NAF_GPU_EHIP, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

// Scratch arena: pointers stay valid until arena_reset.  Returns nullptr on allocation failure.
void  arena_reset(naf_gpu_ctx *c);
void *arena_alloc(naf_gpu_ctx *c, size_t bytes);
void  arena_settle(naf_gpu_ctx *c);            // end of a whole call: arenas that grew become one allocation each (naf_gpu.hip)
template <typename T> T *arena_new(naf_gpu_ctx *c, size_t n) { return (T *)arena_alloc(c, n * sizeof(T)); }

// Small device->host readback through pinned staging (synchronises the stream).
int ctx_readback(naf_gpu_ctx *c, void *h_dst, const void *d_src, size_t bytes);
int ctx_readback2(naf_gpu_ctx *c, void *h1, const void *d1, size_t n1, void *h2, const void *d2, size_t n2);
int ctx_readbackv(naf_gpu_ctx *c, int n, void *const *h, const void *const *d, const size_t *bytes);

void ktime_begin(naf_gpu_ctx *c, const char *name);
void ktime_end(naf_gpu_ctx *c);

// Launch wrapper: records event pairs when timing is on.
#define LAUNCH(c, name, kern, grid, block, shmem, ...) do { \
    ktime_begin((c), name); \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), (shmem), (c)->stream, __VA_ARGS__); \
    ktime_end((c)); } while (0)

static inline u32 cdiv(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

// ---- scan utilities (scan.hip) ---------------------------------------------------------------------
// Exclusive prefix sum of n u64 values in place -> also returns the total through d_total (device u64).
int scan_exclusive_u64(naf_gpu_ctx *c, u64 *d_vals, size_t n, u64 *d_total);
// Inclusive running maximum of i32 values in place (table-ownership propagation).
int scan_inclusive_max_i32(naf_gpu_ctx *c, i32 *d_vals, size_t n);
int scan_inclusive_max_i64(naf_gpu_ctx *c, i64 *d_vals, size_t n);
// k arrays of one length in one set of launches (scan.hip: scan_multi)
int scan_exclusive_u64_multi(naf_gpu_ctx *c, u64 *const *arrs, int k, size_t n, u64 *const *totals);
int scan_inclusive_max_i64_multi(naf_gpu_ctx *c, i64 *const *arrs, int k, size_t n);

// ---- zstd (zstd_dec.hip) -----------------------------------------------------------------------------
// Decode frames at d_src (device).  If only_size, stops after sizes are known.
// head: optional host copy of the first 24 bytes at d_src (fewer when the source is shorter) -- saves the read-back of the frame header
int zstd_decode(naf_gpu_ctx *c, const u8 *d_src, size_t src_len, int has_magic, u8 *d_dst, size_t dst_cap, size_t *out_len, const u8 *head = nullptr);
int zstd_init_tables(naf_gpu_ctx *c);
void ennaf_shard_state_free(naf_gpu_ctx *c);    // enc.hip
void io_pool_free(naf_gpu_ctx *c);               // io.hip
// Range decode: only the blocks whose output intersects [want_lo, want_hi) are decoded; d_dst[0] then holds
// regenerated byte got_lo.  ranged=false means the whole stream was decoded (d_dst[0] = byte 0).
struct EmitP;
// Fused path: decode a literal-only frame straight into FASTA text (no packed stream in HBM).  Returns -100 when
// the frame needs the two-pass path.
int zstd_decode_fused_fasta(naf_gpu_ctx *c, const u8 *d_src, size_t src_len, int has_magic, size_t *out_len, const EmitP *P, u8 *text);
struct ZRange { u64 want_lo, want_hi, got_lo, got_hi; bool ranged; u8 *own_buf; };   // own_buf: set when the decoder had to take a larger buffer than the caller's (the dependency closure of a range in a frame with matches): byte got_lo is own_buf[0]
int zstd_decode_range(naf_gpu_ctx *c, const u8 *d_src, size_t src_len, int has_magic, u8 *d_dst, size_t dst_cap, size_t *out_len, ZRange *rg, const u8 *head = nullptr);
int zstd_split_status(naf_gpu_ctx *c, const ZSplit *sp);
int zstd_flat_later(naf_gpu_ctx *c, ZFlat *zf);
void zstd_flat_drop(ZFlat *zf);
// up to 4 small frames (no magic) in one launch; ok[k] false = take the ordinary path for frame k
int zstd_small_batch(naf_gpu_ctx *c, int n, const u8 *const *src, const size_t *len, u8 *const *dst, const size_t *cap, bool *ok);
bool zstd_small_fits(size_t len, size_t cap);
int zstd_small_launch(naf_gpu_ctx *c, int n, const u8 *const *src, const size_t *len, u8 *const *dst, const size_t *cap, u32 **d_res_out);
// One frame of independently coded blocks; with_magic=0 omits the 4 magic bytes (as stored in a .naf section).
// level >= 2 (or lz != 0) adds the LZ stage (matches inside a block).
enum { ZENC_PART = 16, ZENC_PART_FIRST = 32, ZENC_PART_LAST = 64, ZENC_PREFER_RAW = 128, ZENC_PREFER_FLAT = 256, ZENC_SHORT_CODES = 512, ZENC_FRAME_TREE = 1024 };   // FRAME_TREE: one Huffman code for the blocks of the frame that it fits, the other blocks of it treeless (zstd_enc.hip)   // SHORT_CODES: Huffman codes of at most 7 bits where the block has at most 64 symbols (9 otherwise): the decoder's one-level table and two-sector window   // PREFER_FLAT: k-bit codes for blocks of 2^k symbols unless Huffman coding saves a sixteenth (zstd_enc.hip)   // PREFER_RAW: Huffman only where it saves an eighth of the block     // with_magic flags: a shard's part of a frame (zstd_enc.hip)
int zenc_level_window(int level);
int zenc_repeat_probe(naf_gpu_ctx *c, const u8 *d_src, size_t n, u32 *share_1024);   // level 1: share of sampled anchors that repeat inside their 1 MiB region
// place != nullptr: the frame's size is read back once it is planned and place->fn(place->ud, size) names where it goes (nullptr = give up,
// the hook has set the context's error); d_dst / cap are not looked at.  Saves the copy of a frame whose position depends on its size.
// The level-1 look at a stream (zenc_repeat_probe) reads one MiB in every `zenc_probe_every(n)`: one in 64, fewer for streams above
// 8 GiB so that at most 128 regions are read whatever the size (the share of repeats in 128 MiB is known well enough, and what the look
// reads are blocks the split pass cannot make direct: enc.hip k_direct_blocks asks the same function)
static inline u32 zenc_probe_every(u64 n) { u32 e = 64; while (((n >> 20) / e) > 128) e <<= 1; return e; }
struct ZencPlace { u8 *(*fn)(void *ud, size_t frame_len); void *ud; };
// zstd_encode in two halves: begin queues the planning of the blocks and returns without waiting; finish reads the size back, writes the
// blocks and releases the job (also to be called after a failed begin that left a job).  What a caller queues in between runs beside the planning.
struct ZencJob;
// direct / nd: blocks b < nd of exactly 32 KiB whose four streams of 4-bit codes are already in d_src (enc.hip: direct_word); n must be nd x 32 KiB
// where the codes of a frame's DIRECT blocks wait when the split pass read its text once (enc.hip: k_enc_fused): tile t's bases, two bits
// each, in the KiB at loc + 1024 t; blk_t0[b] = the tile that holds block b's first base
// blk_bnd[b][i] = the base count in front of tile blk_t0[b] + i less the block's first base 65536 b, for the tiles a block can reach
#define ZENC_LOC_BND 20
struct ZencLoc { const u8 *loc; const i32 *blk_bnd; const u32 *blk_t0; u64 tiles; };
int zstd_encode_begin(naf_gpu_ctx *c, const u8 *d_src, size_t n, int level, int with_magic, int lz, int block_log_hint, int window_log, ZencJob **job, const u8 *direct = nullptr, u32 nd = 0, const ZencLoc *dloc = nullptr);
int zstd_encode_finish(naf_gpu_ctx *c, ZencJob *job, u8 *d_dst, size_t cap, size_t *out_len, const ZencPlace *place);
void zstd_encode_drop(ZencJob *job);
int zstd_encode_size(naf_gpu_ctx *c, ZencJob *job, size_t *frame_len);
const u64 *zstd_encode_total_ptr(const ZencJob *job);
void zstd_encode_set_total(ZencJob *job, u64 total);   // the frame's size, read back once and kept for zstd_encode_finish
int zstd_encode(naf_gpu_ctx *c, const u8 *d_src, size_t n, int level, u8 *d_dst, size_t cap, size_t *out_len, int with_magic, int lz = 0, int block_log_hint = 0, int window_log = 0, const ZencPlace *place = nullptr);   // window_log >= 10: cross-block matching inside that window (zstd_enc.hip)
