// enc.hip -- ennaf on gfx950: FASTA stream split, soft-mask run-length scan, 4-bit packing and the
// container writer.  Replaces (reference, ennaf/src): confirm_input_format process.c:547-583,
// process_non_well_formed_fasta process.c:358-427 (+ in_get_until :258, str_append_char :301),
// name/comm/seq writers process.c:12-57, add_length encoders.c:72-95, extract_mask/add_mask
// encoders.c:98-146, encode_dna encoders.c:30-69 (+ nuc_code tables.c:189-197), header/section writer
// ennaf.c:538-589, write_variable_length_encoded_number encoders.c:175-190.
//
// The reference is a byte-at-a-time state machine.  Here the state of any byte is recovered from two
// running maxima -- position of the last EOL-class byte and of the last space-class byte before it --
// because a FASTA line is a header iff it starts with '>' and the ID/comment split is the first
// space-class byte of the header line.  Tiles of 4 KiB (256 lanes x 16 B) are classified independently
// once those maxima are scanned across tiles; stream offsets come from prefix sums of per-tile counts.
#include "ctx.h"
#include <time.h>
#include "wgscan.h"
#include "enc_swar.h"

#define ET_BYTES 16
#define ET_TILE (256 * ET_BYTES)

struct OpMaxI64 { template <typename T> __device__ static T id() { return (T)(-1); } template <typename T> __device__ static T f(T a, T b) { return (i64)a > (i64)b ? a : b; } };
struct OpMaxU32 { template <typename T> __device__ static T id() { return (T)0; } template <typename T> __device__ static T f(T a, T b) { return a > b ? a : b; } };
struct OpMaxU64 { template <typename T> __device__ static T id() { return (T)0; } template <typename T> __device__ static T f(T a, T b) { return a > b ? a : b; } };

#define REG_ACGT 0x80000000u              // t_reg: the regular tile's letters are A C G T / U only
#define REG_E(reg) (((reg) >> 24) & 0x7Fu)   // t_reg: line ends of the tile
struct EncP {
    const u8 *text; u64 n, p0;
    u32 expected[8];             // bitmap of bytes accepted in sequence lines (tables.c:72-123)
    u8 replacement;              // 'N' / 'X' / '?'  (ennaf.c:447-470)
    u8 id_gt_unexpected;         // text+FASTA: '>' also ends ID scanning (ennaf.c:478 flips the shared table)
    u8 strict, pad;
    u32 qlo, qhi;                // quick table of accepted letters (enc_swar.h), built in set_expected
    u32 plo, phi;                // the same with '\n' and '\r' in slots 5 and 6 (piece_plain)
    u32 slo, shi;                // upper-case A C G T / U and '\n' only, a byte that never matches in the other slots (k_enc_fused's first look)
    u32 nuc32[8];                // 4-bit code of the letter whose low five bits are the index (tables.c:189-197), 15 in the other slots
    u32 *any_case;               // count pass: set to 1 when some byte of a sequence line carries the case bit (see case_bytes16); may be null
};
// one bit per byte that the packing pass would give a case bit (bits 5 and 6 both set, or bit 7): a superset of the masked bases
__device__ __forceinline__ u32 case_bytes16(const u32 w[4])
{
    const u32 H = 0x80808080u;
    return swar_movemask16((w[0] | ((w[0] << 1) & (w[0] << 2))) & H, (w[1] | ((w[1] << 1) & (w[1] << 2))) & H,
                           (w[2] | ((w[2] << 1) & (w[2] << 2))) & H, (w[3] | ((w[3] << 1) & (w[3] << 2))) & H);
}
// a wavefront's verdict: one atomic per wave at most, none once the flag is up
__device__ __forceinline__ void note_case(const EncP &P, bool mine)
{
    if (!P.any_case) return;
    if (__ballot(mine) != 0 && (threadIdx.x & 63) == 0 && __atomic_load_n(P.any_case, __ATOMIC_RELAXED) == 0) atomicOr(P.any_case, 1u);
}

__device__ __forceinline__ bool c_eol(u32 c) { return c >= 0x0A && c <= 0x0D; }
__device__ __forceinline__ bool c_space(u32 c) { return (c >= 0x09 && c <= 0x0D) || c == 0x20; }
__device__ __forceinline__ bool c_unexp_text(u32 c) { return c <= 0x20 || c == 0x7F || c == 0xFF; }
__device__ __forceinline__ bool c_unexp_comment(u32 c) { return c < 0x20 || c == 0x7F || c == 0xFF; }
__device__ __forceinline__ bool c_expected(const EncP &P, u32 c) { return (P.expected[c >> 5] >> (c & 31)) & 1; }

// ---- K1: per-tile last EOL / last space position -------------------------------------------------------------
// A thread's 16 input bytes, held in registers (two 8-byte loads instead of 16 byte loads).
struct Piece { u64 w0, w1; u32 cnt; };
__device__ __forceinline__ Piece load_piece(const EncP &P, u64 base)
{
    Piece pc; pc.w0 = pc.w1 = 0; pc.cnt = 0;
    if (base >= P.n) return pc;
    if (base + ET_BYTES <= P.n) { pc.w0 = ld64(P.text + base); pc.w1 = ld64(P.text + base + 8); pc.cnt = ET_BYTES; return pc; }
    pc.cnt = (u32)(P.n - base);
    for (u32 i = 0; i < pc.cnt; i++) { u64 c = P.text[base + i]; if (i < 8) pc.w0 |= c << (8 * i); else pc.w1 |= c << (8 * (i - 8)); }
    return pc;
}
__device__ __forceinline__ u32 piece_byte(const Piece &pc, u32 k) { return (u32)((k < 8 ? pc.w0 >> (8 * k) : pc.w1 >> (8 * (k - 8))) & 0xFF); }

// Byte classes in LDS: one lookup per byte instead of range compares and a bitmap fetched from kernel arguments.
enum { CL_EOL = 1, CL_SPACE = 2, CL_EXPECTED = 4, CL_GT = 8, CL_UNEXP_TEXT = 16, CL_UNEXP_COMMENT = 32, CL_QUAL = 64 };
__device__ __forceinline__ void fill_classes(const EncP &P, u8 *cls)      // blockDim.x == 256
{
    u32 c = threadIdx.x;
    cls[c] = (u8)((c_eol(c) ? CL_EOL : 0) | (c_space(c) ? CL_SPACE : 0) | (c_expected(P, c) ? CL_EXPECTED : 0) | (c == '>' ? CL_GT : 0) |
                  (c_unexp_text(c) ? CL_UNEXP_TEXT : 0) | (c_unexp_comment(c) ? CL_UNEXP_COMMENT : 0) | ((c >= 0x21 && c <= 0x7E) ? CL_QUAL : 0));
    __syncthreads();
}

// One bit per byte of the piece for each class (bit k = byte k).  With these a piece that lies inside sequence lines needs no
// per-byte state machine: counts are popcounts, the last EOL / space a count-leading-zeros.  EOL / space / '>' come from four-
// bytes-at-a-time arithmetic (enc_swar.h); a zero-padded partial piece sets none of them.
struct PMask { u32 eol, sp, gt; };
__device__ __forceinline__ PMask piece_masks(const Piece &pc)
{
    u32 w[4] = { (u32)pc.w0, (u32)(pc.w0 >> 32), (u32)pc.w1, (u32)(pc.w1 >> 32) };
    PieceFlags f = piece_flags(w);
    PMask m; m.eol = f.eol; m.sp = f.sp; m.gt = f.gt;
    return m;
}
// Every byte of a full piece is space-class (when `allow_space`) or an accepted sequence letter: the quick table answers for
// A C G T N in either case, the bytes it does not know are looked up in the class table one by one.
__device__ __forceinline__ bool piece_all_expected(const EncP &P, const Piece &pc, const PMask &pm, const u8 *cls, bool allow_space)
{
    u32 w[4] = { (u32)pc.w0, (u32)(pc.w0 >> 32), (u32)pc.w1, (u32)(pc.w1 >> 32) };
    u32 cand = piece_not_quick(w, P.qlo, P.qhi);
    if (allow_space) cand &= ~pm.sp;
    while (cand) {
        u32 k = (u32)__ffs((int)cand) - 1; cand &= cand - 1;
        if (!(cls[piece_byte(pc, k)] & CL_EXPECTED)) return false;
    }
    return true;
}
__device__ __forceinline__ bool piece_all_qual(const Piece &pc)
{
    u32 w[4] = { (u32)pc.w0, (u32)(pc.w0 >> 32), (u32)pc.w1, (u32)(pc.w1 >> 32) };
    return piece_all_quality(w);
}
__device__ __forceinline__ void last_eol_space_m(const PMask &m, u64 base, i64 &le, i64 &ls)
{
    le = m.eol ? (i64)(base + (31 - __clz((int)m.eol))) : -1;
    ls = m.sp ? (i64)(base + (31 - __clz((int)m.sp))) : -1;
}
// (position in the tile + 1) of the piece's last EOL-class byte in the low half and of its last space-class byte in the high
// half, 0 = none: both running maxima of a tile travel through one scan of packed 16-bit values.
__device__ __forceinline__ u32 tile_pos_pair(const PMask &m)
{
    u32 t16 = threadIdx.x * ET_BYTES + 1;
    return (m.eol ? t16 + (31 - __clz((int)m.eol)) : 0u) | ((m.sp ? t16 + (31 - __clz((int)m.sp)) : 0u) << 16);
}
// Line starts: a non-EOL byte at i >= p0 whose predecessor is an EOL byte (or i == p0).
__device__ __forceinline__ u32 count_line_starts(const EncP &P, u64 base, const Piece &pc)
{
    u32 n = 0; bool prev_eol = base == 0 ? true : c_eol(P.text[base - 1]);
#pragma unroll
    for (u32 i = 0; i < ET_BYTES; i++) if (i < pc.cnt) { u32 c = piece_byte(pc, i); bool e = c_eol(c); if (!e && base + i >= P.p0 && (prev_eol || base + i == P.p0)) n++; prev_eol = e; }
    return n;
}
__device__ __forceinline__ u32 count_line_starts_m(const EncP &P, u64 base, const Piece &pc, const PMask &m)
{
    if (base + ET_BYTES <= P.p0) return 0;
    if (base < P.p0) return count_line_starts(P, base, pc);               // the piece holding the first marker: byte-wise
    u32 valid = pc.cnt >= 32 ? ~0u : ((1u << pc.cnt) - 1);
    u32 prev = base == 0 ? 1u : (c_eol(P.text[base - 1]) ? 1u : 0u);
    if (base == P.p0) prev = 1;
    return __popc(~m.eol & ((m.eol << 1) | prev) & valid);
}

__device__ __forceinline__ void last_eol_space(const Piece &pc, u64 base, i64 &le, i64 &ls)
{
    le = -1; ls = -1;
#pragma unroll
    for (u32 i = 0; i < ET_BYTES; i++) if (i < pc.cnt) { u32 c = piece_byte(pc, i); if (c_space(c)) { ls = (i64)(base + i); if (c_eol(c)) le = ls; } }
}

__global__ __launch_bounds__(256) void k_enc_last(EncP P, i64 *tile_eol, i64 *tile_sp, u64 *tile_ls)
{
    __shared__ u32 s_pos[4], s_nls[4];
    const u32 bid = blockIdx.x;
    u64 base = (u64)bid * ET_TILE + (u64)threadIdx.x * ET_BYTES;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Piece pc = load_piece(P, base);
    PMask pm = piece_masks(pc);
    u32 v = wave_scan_inclusive<u32, OpPkMaxU16>(tile_pos_pair(pm));
    u32 nls = (tile_ls && pc.cnt) ? count_line_starts_m(P, base, pc, pm) : 0;
    if (tile_ls) nls = wave_scan_inclusive<u32, OpAdd>(nls);
    if (lane == 63) { s_pos[wave] = v; s_nls[wave] = nls; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 m = OpPkMaxU16::f<u32>(OpPkMaxU16::f<u32>(s_pos[0], s_pos[1]), OpPkMaxU16::f<u32>(s_pos[2], s_pos[3]));
        u64 tb = (u64)bid * ET_TILE;
        tile_eol[bid] = (m & 0xFFFF) ? (i64)(tb + (m & 0xFFFF) - 1) : -1;
        tile_sp[bid] = (m >> 16) ? (i64)(tb + (m >> 16) - 1) : -1;
        if (tile_ls) tile_ls[bid] = (u64)s_nls[0] + s_nls[1] + s_nls[2] + s_nls[3];
    }
}

// FASTA wants only the LAST EOL / space of each tile.  In line-wrapped text both sit in the tile's last kilobyte: one wavefront
// per tile looks there first and walks back through the other three quarters only while something is still missing.
// Sixteen lanes look at the last 256 bytes of a tile (four tiles per wave): a FASTA line of up to 255 columns ends in there, and a
// line end is a space-class byte too.  Only a tile whose window holds neither walks its quarters from the back with the whole wave
// (a long header line, an unwrapped sequence) -- the first look used to be the whole last KiB of every tile, a quarter of the text.
#define LAST_TPW 4
__global__ __launch_bounds__(256) void k_enc_last_fa(EncP P, i64 *tile_eol, i64 *tile_sp, u64 tiles)
{
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane >> 4, l = lane & 15;
    const u64 t0 = ((u64)blockIdx.x * 4 + wave) * LAST_TPW;
    if (t0 >= tiles) return;
    {
        const u64 t = t0 + row, tb = t * ET_TILE;
        Piece pc; pc.w0 = pc.w1 = 0; pc.cnt = 0;
        if (t < tiles) pc = load_piece(P, tb + (ET_TILE - 256) + l * ET_BYTES);
        PMask pm = piece_masks(pc);
        const u32 t16 = (ET_TILE - 256) + l * ET_BYTES + 1;
        u32 v = (pm.eol ? t16 + (31 - __clz((int)pm.eol)) : 0u) | ((pm.sp ? t16 + (31 - __clz((int)pm.sp)) : 0u) << 16);
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) v = OpPkMaxU16::f<u32>(v, (u32)__shfl_xor((int)v, d, 64));
        const bool ok = (v & 0xFFFF) && (v >> 16);
        if (t < tiles && ok && l == 0) { tile_eol[t] = (i64)(tb + (v & 0xFFFF) - 1); tile_sp[t] = (i64)(tb + (v >> 16) - 1); }
        u64 need = __ballot(t < tiles && !ok && l == 0);
        while (need) {
            const u32 jj = (u32)(__ffsll((long long)need) - 1) >> 4; need &= need - 1;
            const u64 tt = t0 + jj, tbb = tt * ET_TILE; u32 found = 0;
            for (int q = 3; q >= 0; q--) {
                Piece p2 = load_piece(P, tbb + (u32)q * 1024 + lane * ET_BYTES);
                PMask m2 = piece_masks(p2);
                u32 u16 = (u32)q * 1024 + lane * ET_BYTES + 1;
                u32 w = (m2.eol ? u16 + (31 - __clz((int)m2.eol)) : 0u) | ((m2.sp ? u16 + (31 - __clz((int)m2.sp)) : 0u) << 16);
                w = (u32)__builtin_amdgcn_readlane((int)wave_scan_inclusive<u32, OpPkMaxU16>(w), 63);
                if (!(found & 0xFFFF)) found |= w & 0xFFFF;
                if (!(found >> 16)) found |= w & 0xFFFF0000u;
                if ((found & 0xFFFF) && (found >> 16)) break;
            }
            if (lane == 0) {
                tile_eol[tt] = (found & 0xFFFF) ? (i64)(tbb + (found & 0xFFFF) - 1) : -1;
                tile_sp[tt] = (found >> 16) ? (i64)(tbb + (found >> 16) - 1) : -1;
            }
        }
    }
}

// ---- classification of 16 bytes given the running maxima at the first byte ---------------------------------------
enum { EV_SEQ = 0, EV_IDS = 1, EV_CMT = 2 };
struct TileCtx { i64 last_eol, last_sp; bool hdr; i64 ord; };

// Sink interface: emit(stream, ch); header_start(pos); header_end(pos); line_end(pos) for sequence lines;
// unexpected(kind, ch) with kind 0 id, 1 comment, 2 sequence.
template <typename Sink>
__device__ __forceinline__ void classify_range(const EncP &P, u64 pos, const Piece &pc, bool with_eof, TileCtx ctx, Sink &S, const u8 *cls)
{
    i64 le = ctx.last_eol, ls = ctx.last_sp; bool hdr = ctx.hdr;
    const u32 cnt = pc.cnt;
    for (u32 k = 0; k < cnt + (with_eof ? 1u : 0u); k++) {
        u64 i = pos + k;
        bool eof = k >= cnt;
        u32 c = eof ? 0x0A : piece_byte(pc, k);               // end of input acts as one final line end
        u32 cl = cls[c];
        if (i >= P.p0) {                                       // leading space-class bytes are skipped (process.c:551-553)
            i64 line_start = le + 1;
            if (hdr) {
                if ((i64)i == line_start) { S.header_start(i); }
                else if (ls < line_start) {                    // still inside the ID (process.c:363-368)
                    if (cl & CL_SPACE) { S.emit(EV_IDS, 0); if (cl & CL_EOL) { S.emit(EV_CMT, 0); S.header_end(i); } }
                    else if ((cl & CL_UNEXP_TEXT) || (P.id_gt_unexpected && c == '>')) { S.unexpected(0, c, i); S.emit(EV_SEQ, '?'); }
                    else S.emit(EV_IDS, c);
                } else {                                       // comment (process.c:370-377)
                    if (cl & CL_EOL) { S.emit(EV_CMT, 0); S.header_end(i); }
                    else if (cl & CL_UNEXP_COMMENT) { S.unexpected(1, c, i); S.emit(EV_CMT, '?'); }
                    else S.emit(EV_CMT, c);
                }
            } else {                                           // sequence line (process.c:387-412)
                if (cl & CL_EOL) S.line_end(i);
                else if (cl & CL_SPACE) {}
                else if (cl & CL_EXPECTED) S.emit(EV_SEQ, c);
                else { S.unexpected(2, c, i); S.emit(EV_SEQ, P.replacement); }
            }
        }
        if (eof) break;
        if (cl & CL_SPACE) {
            ls = (i64)i;
            if (cl & CL_EOL) { le = (i64)i; u32 nx = k + 1 < cnt ? piece_byte(pc, k + 1) : (i + 1 < P.n ? P.text[i + 1] : 0u); hdr = nx == '>'; }
        }
    }
}

// Running maxima and header flag at the first byte of this thread's 16-byte piece.
__device__ __forceinline__ TileCtx thread_ctx(const EncP &P, const i64 *tile_eol, const i64 *tile_sp, u64 base, const PMask &pm)
{
    __shared__ u32 wl[4];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 inc = wave_scan_inclusive<u32, OpPkMaxU16>(tile_pos_pair(pm));
    if (lane == 63) wl[wave] = inc;
    u32 ex = wave_shift_up1<u32, OpPkMaxU16>(inc);              // exclusive: what the earlier lanes saw
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 3; w++) if (w < wave) ex = OpPkMaxU16::f<u32>(ex, wl[w]);
    const u64 tile = base / ET_TILE, tb = tile * ET_TILE;           // (the workgroup's tile: not always blockIdx.x, see EncOut::irr_list)
    TileCtx c;
    // a hit inside the tile is later than anything carried in from the tiles before it
    c.last_eol = (ex & 0xFFFF) ? (i64)(tb + (ex & 0xFFFF) - 1) : (tile ? tile_eol[tile - 1] : -1);
    c.last_sp = (ex >> 16) ? (i64)(tb + (ex >> 16) - 1) : (tile ? tile_sp[tile - 1] : -1);
    i64 ls0 = c.last_eol + 1;
    c.hdr = (u64)ls0 < P.n && P.text[ls0] == '>';
    return c;
}

// ---- segment-wise classification of a piece that holds header bytes ---------------------------------------------------------
// The same state machine as classify_range, advanced one LINE SEGMENT at a time: the EOL bits of the piece cut it into at most a few
// runs, each of which is header text (ID up to the first space-class byte, comment after it) or sequence, and moves to its stream
// in bulk.  Valid only for a full piece behind p0 whose bytes need no replacement in the roles they fall into; the caller finds
// that out with a dry run into a RoleSink (one bit per byte and role) before anything is emitted, and falls back to the per-byte
// walk otherwise.  Sink interface: ids_range / cmt_range / seq_range(piece, first, end[, space bits]), term(stream) for the
// \0 terminators, header_start / header_end / line_end as in classify_range.
__device__ __forceinline__ u32 range_mask(u32 a, u32 b) { return ((1u << b) - 1) & ~((1u << a) - 1); }       // bits [a, b), b <= 16
template <typename Sink>
__device__ __forceinline__ void classify_segments(const EncP &P, u64 base, const Piece &pc, const PMask &pm, const TileCtx &ctx, Sink &S)
{
    i64 le = ctx.last_eol, ls = ctx.last_sp; bool hdr = ctx.hdr;
    u32 k = 0;
    while (k < ET_BYTES) {
        const u32 em = pm.eol >> k;
        const u32 e = em ? k + (u32)__ffs((int)em) - 1 : ET_BYTES;         // the EOL that ends this run, or the end of the piece
        if (hdr) {
            const i64 line_start = le + 1;
            u32 a = k;
            if ((i64)(base + a) == line_start) { S.header_start(base + a); a++; }     // the '>' itself
            if (a < e) {
                const u32 spm = pm.sp & range_mask(a, e);
                if (ls < line_start) {                                      // no space-class byte in this line so far: the ID
                    const u32 f = spm ? (u32)__ffs((int)spm) - 1 : e;
                    if (f > a) S.ids_range(pc, a, f);
                    if (f < e) { S.term(EV_IDS); if (f + 1 < e) S.cmt_range(pc, f + 1, e); }
                } else S.cmt_range(pc, a, e);
                if (spm) ls = (i64)(base + (31 - __clz((int)spm)));
            }
            if (e < ET_BYTES) { if (ls < line_start) S.term(EV_IDS); S.term(EV_CMT); S.header_end(base + e); }
        } else {
            if (k < e) { S.seq_range(pc, k, e, pm.sp); const u32 spm = pm.sp & range_mask(k, e); if (spm) ls = (i64)(base + (31 - __clz((int)spm))); }
            if (e < ET_BYTES) S.line_end(base + e);
        }
        if (e >= ET_BYTES) break;
        le = ls = (i64)(base + e);
        const u32 nx = e + 1 < ET_BYTES ? piece_byte(pc, e + 1) : (base + ET_BYTES < P.n ? (u32)P.text[base + ET_BYTES] : 0u);
        hdr = nx == '>';
        k = e + 1;
    }
}
struct RoleSink {
    u32 idm = 0, cmm = 0, sqm = 0;
    __device__ void ids_range(const Piece &, u32 a, u32 b) { idm |= range_mask(a, b); }
    __device__ void cmt_range(const Piece &, u32 a, u32 b) { cmm |= range_mask(a, b); }
    __device__ void seq_range(const Piece &, u32 a, u32 b, u32 sp) { sqm |= range_mask(a, b) & ~sp; }
    __device__ void term(int) {} __device__ void header_start(u64) {} __device__ void header_end(u64) {} __device__ void line_end(u64) {}
};
// Can this piece take the segment-wise path?  (Full piece behind p0, and no byte that its role would replace.)
__device__ __forceinline__ bool segments_ok(const EncP &P, u64 base, const Piece &pc, const PMask &pm, const TileCtx &ctx, const u8 *cls)
{
    if (pc.cnt != ET_BYTES || base <= P.p0) return false;
    RoleSink R; classify_segments(P, base, pc, pm, ctx, R);
    u32 w[4] = { (u32)pc.w0, (u32)(pc.w0 >> 32), (u32)pc.w1, (u32)(pc.w1 >> 32) };
    if ((R.idm | R.cmm) && ((R.idm | R.cmm) & piece_ctl_mask(w))) return false;
    if (P.id_gt_unexpected && (R.idm & pm.gt)) return false;
    u32 cand = R.sqm ? (piece_not_quick(w, P.qlo, P.qhi) & R.sqm) : 0u;
    while (cand) { u32 k = (u32)__ffs((int)cand) - 1; cand &= cand - 1; if (!(cls[piece_byte(pc, k)] & CL_EXPECTED)) return false; }
    return true;
}
// bytes [a, 16) of the piece moved down to byte 0
__device__ __forceinline__ void piece_from(const Piece &pc, u32 a, u64 &lo, u64 &hi)
{
    if (a == 0) { lo = pc.w0; hi = pc.w1; }
    else if (a < 8) { lo = (pc.w0 >> (8 * a)) | (pc.w1 << (64 - 8 * a)); hi = pc.w1 >> (8 * a); }
    else { lo = pc.w1 >> (8 * (a - 8)); hi = 0; }
}

struct CountSink {
    u32 nseq = 0, nids = 0, ncmt = 0, nrec = 0, tail = 0;      // tail = sequence bytes since the last EOL seen
    bool saw_eol = false, cased = false;                         // cased: some sequence byte carries the case bit (what is emitted for it, or the byte itself)
    __device__ void emit(int s, u32 ch) { if (s == EV_SEQ) { nseq++; tail++; cased = cased || (ch & 0x80u) || ((ch & 0x60u) == 0x60u); } else if (s == EV_IDS) nids++; else ncmt++; }
    __device__ void header_start(u64) { nrec++; }
    __device__ void header_end(u64) { tail = 0; saw_eol = true; }
    __device__ void line_end(u64) { tail = 0; saw_eol = true; }
    __device__ void unexpected(int, u32, u64) {}
    __device__ void ids_range(const Piece &, u32 a, u32 b) { nids += b - a; }
    __device__ void cmt_range(const Piece &, u32 a, u32 b) { ncmt += b - a; }
    __device__ void seq_range(const Piece &pc, u32 a, u32 b, u32 sp)
    {
        u32 c = (u32)__popc(range_mask(a, b) & ~sp); nseq += c; tail += c;
        const u32 w[4] = { (u32)pc.w0, (u32)(pc.w0 >> 32), (u32)pc.w1, (u32)(pc.w1 >> 32) };
        cased = cased || (case_bytes16(w) & range_mask(a, b) & ~sp) != 0;
    }
    __device__ void term(int st) { if (st == EV_IDS) nids++; else ncmt++; }
};

// A full 16-byte piece that lies inside sequence lines: not in a header at its first byte, no header starting inside it.
__device__ __forceinline__ bool seq_piece(const EncP &P, u64 base, const Piece &pc, const PMask &pm, const TileCtx &ctx)
{
    return pc.cnt == ET_BYTES && base >= P.p0 && !ctx.hdr && ((pm.eol << 1) & pm.gt & 0xFFFFu) == 0;
}

// ---- K2: per-tile stream byte counts -----------------------------------------------------------------------------------
// A PURE tile -- wholly inside [p0, n), beginning outside a header line and made of plain pieces (piece_plain: A C G T/U N in either
// case, LF, CR; so no '>') -- is sequence lines and nothing else:
// every piece takes the popcount path, no lane needs its own running maxima (thread_ctx) and nothing is added to ids, comments or
// the record table.  Long-record FASTA is pure tiles almost everywhere; the test is workgroup-uniform.
__device__ __forceinline__ bool tile_may_be_pure(const EncP &P, const i64 *tile_eol, u64 tile)
{
    const u64 tb = tile * ET_TILE;
    if (tb < P.p0 || tb + ET_TILE > P.n) return false;
    const i64 le0 = tile ? tile_eol[tile - 1] : -1;                // thread_ctx's c.hdr for a lane with no EOL in front of it in the tile
    const u64 ls0 = (u64)(le0 + 1);
    return !(ls0 < P.n && P.text[ls0] == '>');
}
// bases of the piece behind its last EOL byte (all of them when it has none)
__device__ __forceinline__ u32 piece_tail_bases(const PMask &pm)
{
    return pm.eol ? (u32)__popc(~pm.sp & 0xFFFFu & ~((2u << (31 - __clz((int)pm.eol))) - 1)) : 16u - (u32)__popc(pm.sp);
}

__global__ __launch_bounds__(256) void k_enc_count(EncP P, const i64 *tile_eol, const i64 *tile_sp,
                                                    u64 *t_seq, u64 *t_ids, u64 *t_cmt, u64 *t_rec, u32 *t_tail, u32 *t_reg, u64 *t_irr, const u32 *list)
{
    const u64 tile = list ? list[blockIdx.x] : blockIdx.x;      // list: the tiles k_enc_count_pure left (4-bit sequence types)

    __shared__ u8 cls[256];
    __shared__ u32 s_a[4], s_b[4], s_last[4];
    // REGULAR tiles (t_reg, read by k_enc_scatter<true>): plain pieces whose line ends sit on a lattice -- the first at p1, then one
    // every `period` bytes, period >= 33: lines of one width with one-byte ends, nearly every tile of a line-wrapped genome.  The
    // test needs text positions only: every line end but the tile's first lies `period` behind the one before it.
    // Every wavefront checks its own line ends against its own period (the distance of its first two) and leaves count, first and
    // last position, period and verdict; the tile's first lane puts the four together -- no barrier beyond the one the counts need.
    __shared__ u32 r_cnt[4], r_first[4], r_last[4], r_per[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool maybe = tile_may_be_pure(P, tile_eol, tile);
    u64 base = (u64)tile * ET_TILE + (u64)threadIdx.x * ET_BYTES;
    Piece pc = load_piece(P, base);
    {
        // Plain pieces only (quick letters, LF, CR): their line ends are their only space-class bytes, so a pure tile's base counts
        // are arithmetic on positions -- 4096 less its line-end bytes, and what follows the last of them -- no prefix sum.  Per wave:
        // the number of line-end bytes and the position of the last one.
        const u32 w[4] = { (u32)pc.w0, (u32)(pc.w0 >> 32), (u32)pc.w1, (u32)(pc.w1 >> 32) };
        PMask pl; pl.gt = 0;
        const bool plain = piece_plain(w, P.plo, P.phi, &pl.eol); pl.sp = pl.eol;
        const u64 bal = __ballot(pl.eol != 0);
        const int lastl = bal ? 63 - __clzll((long long)bal) : -1;
        const bool two = __ballot((pl.eol & (pl.eol - 1)) != 0) != 0;                                // some piece holds two line-end bytes (CR LF, short lines)
        u32 nb = (u32)__popcll(bal);
        if (two) nb = (u32)__builtin_amdgcn_readlane((int)wave_scan_inclusive<u32, OpAdd>((u32)__popc(pl.eol)), 63);
        const u32 hib = threadIdx.x * ET_BYTES + (pl.eol ? 31u - (u32)__clz((int)pl.eol) : 0u);
        const u32 lastpos = (u32)__builtin_amdgcn_readlane((int)hib, lastl < 0 ? 0 : lastl);
        const bool wave_bad = __ballot(!plain) != 0;
        if (maybe && !wave_bad) note_case(P, case_bytes16(w) != 0);        // (a pure tile: letters and line ends only; a tile that is not comes back below)
        if (lane == 0) { s_a[wave] = nb | (lastpos << 16); s_last[wave] = (bal ? 1u : 0u) | (wave_bad ? 2u : 0u); }
        if (maybe) {
            const bool has = pl.eol != 0;
            const u32 q = threadIdx.x * ET_BYTES + (has ? (u32)__ffs((int)pl.eol) - 1 : 0u);         // position of the lane's line end in the tile
            const u64 mlow = bal & ((1ull << lane) - 1);
            const u32 qprev = (u32)__shfl((int)q, mlow ? 63 - __clzll((long long)mlow) : lane, 64);
            const u64 bal2 = bal & (bal - 1);
            const int f1 = bal ? __ffsll((long long)bal) - 1 : 0, f2 = bal2 ? __ffsll((long long)bal2) - 1 : 0;
            const u32 q1 = (u32)__builtin_amdgcn_readlane((int)q, f1), q2 = (u32)__builtin_amdgcn_readlane((int)q, f2);
            const u32 ql = (u32)__builtin_amdgcn_readlane((int)q, lastl < 0 ? 0 : lastl);
            const u32 per = bal2 ? q2 - q1 : 0u;                                                      // 0: fewer than two line ends in this wave
            const bool wbad = two || __ballot(has && mlow && q - qprev != per) != 0;                 // two in one piece, or off the wave's lattice
            if (lane == 0) { r_cnt[wave] = (u32)__popcll(bal) | (wbad ? 0x10000u : 0u); r_first[wave] = q1; r_last[wave] = ql; r_per[wave] = per; }
        }
    }
    __syncthreads();
    if (maybe && !((s_last[0] | s_last[1] | s_last[2] | s_last[3]) & 2u)) {
        if (threadIdx.x == 0) {
            // the four waves' line ends as one lattice: equal periods, and every wave's first line end one period behind the last one before it
            u32 E = 0, p1 = 0, period = 0, prev_last = 0; bool ok = true, any = false;
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const u32 cv = r_cnt[v], cnt = cv & 0xFFFFu;
                if (cv >> 16) ok = false;
                if (!cnt) continue;
                if (any) { const u32 gap = r_first[v] - prev_last; if (!period) period = gap; else if (gap != period) ok = false; }
                else p1 = r_first[v];
                if (r_per[v]) { if (!period) period = r_per[v]; else if (r_per[v] != period) ok = false; }
                any = true; prev_last = r_last[v]; E += cnt;
            }
            // (the gather of the scatter pass reads up to 17 bytes from a base's position: not in the text's last tiles)
            // ... and the lattice's NEXT point lies behind the tile: a tile whose line ends stop short of its end (wrapped lines, then a long
            // one) is not regular -- the scatter pass places a regular tile's bases by the lattice alone
            const bool tile_ok = ok && E >= 2 && period >= 33 && p1 + E * period >= ET_TILE && ((u64)tile + 1) * ET_TILE + 32 <= P.n;
            t_reg[tile] = tile_ok ? (p1 | (period << 12) | (E << 24)) : 0u; t_irr[tile] = tile_ok ? 0 : 1;
            u32 tot = ET_TILE, tail = 0; bool found = false;
#pragma unroll
            for (int w = 3; w >= 0; w--) { tot -= s_a[w] & 0xFFFF; if (!found && (s_last[w] & 1u)) { tail = ET_TILE - 1 - (s_a[w] >> 16); found = true; } }
            t_seq[tile] = tot; t_ids[tile] = 0; t_cmt[tile] = 0; t_rec[tile] = 0;
            t_tail[tile] = found ? (tail | 0x80000000u) : tot;
        }
        return;
    }
    PMask pm = piece_masks(pc);
    fill_classes(P, cls);                                         // (its barrier also separates the reads of s_a / s_last above from the writes below)
    TileCtx ctx = thread_ctx(P, tile_eol, tile_sp, base, pm);
    CountSink S;
    if (seq_piece(P, base, pc, pm, ctx)) {
        // a full piece inside sequence lines: every non-space byte is a base (process.c:387-412)
        S.nseq = 16 - __popc(pm.sp); S.saw_eol = pm.eol != 0;
        S.tail = pm.eol ? __popc(~pm.sp & 0xFFFFu & ~((2u << (31 - __clz((int)pm.eol))) - 1)) : S.nseq;
        { const u32 w4[4] = { (u32)pc.w0, (u32)(pc.w0 >> 32), (u32)pc.w1, (u32)(pc.w1 >> 32) }; S.cased = (case_bytes16(w4) & ~pm.sp & 0xFFFFu) != 0; }
    } else if (base <= P.n) {
        // the virtual end-of-input byte belongs to the thread whose piece contains position n
        bool eof_here = (base + pc.cnt == P.n) && pc.cnt < ET_BYTES;
        if (segments_ok(P, base, pc, pm, ctx, cls)) classify_segments(P, base, pc, pm, ctx, S);
        else classify_range(P, base, pc, eof_here, ctx, S, cls);
    }
    // four counts in two scans of packed 16-bit fields (a tile holds 4096 bytes, a byte adds at most 2 to a field); the tail of
    // the tile -- sequence bytes after its last EOL, the whole tile if it has none -- falls out of the same scan: the last
    // thread that saw an EOL knows how many bases follow it.
    note_case(P, S.cased);
    u32 wa = wave_scan_inclusive<u32, OpAdd>(S.nseq | (S.nids << 16)), wb = wave_scan_inclusive<u32, OpAdd>(S.ncmt | (S.nrec << 16));
    u64 bal = __ballot(S.saw_eol);
    if (lane == 63) { s_a[wave] = wa; s_b[wave] = wb; }
    if (lane == 0) s_last[wave] = bal ? (u32)(wave * 64 + (63 - __clzll((long long)bal)) + 1) : 0u;
    __syncthreads();
    u32 pre = 0, tota = 0, totb = 0, last = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { if (w < wave) pre += s_a[w]; tota += s_a[w]; totb += s_b[w]; last = s_last[w] > last ? s_last[w] : last; }
    if (threadIdx.x == 0) {
        t_seq[tile] = tota & 0xFFFF; t_ids[tile] = tota >> 16; t_cmt[tile] = totb & 0xFFFF; t_rec[tile] = totb >> 16;
        if (!last) t_tail[tile] = tota & 0xFFFF;
        t_reg[tile] = 0; t_irr[tile] = 1;
    }
    if (threadIdx.x + 1 == last) t_tail[tile] = ((tota & 0xFFFF) - ((pre + wa) & 0xFFFF) + S.tail) | 0x80000000u;
}

// ---- the same verdicts for PURE tiles, a wavefront per tile (4-bit sequence types).  A lane holds four pieces of its tile (loads of
// neighbouring lanes touch neighbouring bytes, all four in flight together); the whole tile is in one wavefront, so counts, the
// position of the last line end and the lattice test need no LDS and no barrier.  A tile that is not pure (it starts in a header, is
// not wholly inside the text, or holds anything but plain letters and line ends) is left to k_enc_count (t_needf / t_need).
// one tile of k_enc_count_pure: v = the tile's bytes, sixteen per lane and quarter (loaded by the caller when `inside`)
struct PureTile { bool pure, ok, acgt, any; u32 p1, period, E, lastpos; };      // pure: the tables are written; ok: regular (t_reg != 0); any: it holds a line end
// the plain test of a tile's 4 x 16 bytes per lane: line ends per quarter, N and lower case seen (wave-uniform verdict)
__device__ __forceinline__ bool plain_tile(const EncP &P, const uint4 (&v)[4], u32 (&eol)[4], bool &acgt, bool &lower_any)
{
    bool plain = true; u32 has_n = 0, lower = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32 w[4] = { v[k].x, v[k].y, v[k].z, v[k].w }; plain = piece_plain(w, P.plo, P.phi, &eol[k]) && plain;
        // among the plain bytes (A C G T U N in either case, LF, CR) only N has both bit 3 and bit 6
        has_n |= ((w[0] >> 3) & (w[0] >> 6)) | ((w[1] >> 3) & (w[1] >> 6)) | ((w[2] >> 3) & (w[2] >> 6)) | ((w[3] >> 3) & (w[3] >> 6));
        // ... and only lower-case letters have bits 5 and 6 (line ends have neither 6 nor 7)
        lower |= ((w[0] >> 5) & (w[0] >> 6)) | ((w[1] >> 5) & (w[1] >> 6)) | ((w[2] >> 5) & (w[2] >> 6)) | ((w[3] >> 5) & (w[3] >> 6));
    }
    if (__ballot(!plain) != 0) return false;
    lower_any = (lower & 0x01010101u) != 0;
    acgt = __ballot((has_n & 0x01010101u) != 0) == 0;
    return true;
}
// counts, last line end and lattice of a plain tile from its line-end bits; writes the tile's table entries
__device__ __forceinline__ PureTile count_plain_tile(const EncP &P, u64 *t_seq, u64 *t_ids, u64 *t_cmt, u64 *t_rec, u32 *t_tail, u32 *t_reg, u64 *t_irr,
                                                     u32 *t_needf, u64 *t_need, u64 t, const u32 (&eol)[4], bool acgt)
{
    const u32 lane = threadIdx.x & 63;
    u32 E = 0, nbytes = 0, p1 = 0, period = 0, prev_last = 0, lastpos = 0; bool any = false, two = false, lat_ok = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const bool has = eol[k] != 0;
        const u64 bal = __ballot(has);
        if (!bal) continue;                                           // (uniform)
        const bool multi = __ballot((eol[k] & (eol[k] - 1)) != 0) != 0;
        two = two || multi;
        const u32 at = (64u * (u32)k + lane) * ET_BYTES;
        const u32 q = at + (has ? (u32)__ffs((int)eol[k]) - 1 : 0u), hib = at + (has ? 31u - (u32)__clz((int)eol[k]) : 0u);
        const u32 cnt = (u32)__popcll(bal);
        nbytes += multi ? (u32)__builtin_amdgcn_readlane((int)wave_scan_inclusive<u32, OpAdd>((u32)__popc(eol[k])), 63) : cnt;
        const int f1 = __ffsll((long long)bal) - 1, l1 = 63 - __clzll((long long)bal);
        const u32 qf = (u32)__builtin_amdgcn_readlane((int)q, f1), ql = (u32)__builtin_amdgcn_readlane((int)q, l1);
        lastpos = (u32)__builtin_amdgcn_readlane((int)hib, l1);
        const u64 mlow = bal & ((1ull << lane) - 1);
        const u32 qsh = (u32)__shfl((int)q, mlow ? 63 - __clzll((long long)mlow) : (int)lane, 64);   // (every lane takes part: a lane that sits out cannot be read)
        const u32 qprev = mlow ? qsh : prev_last;
        if (!any) { p1 = qf; const u64 bal2 = bal & (bal - 1); if (bal2) period = (u32)__builtin_amdgcn_readlane((int)q, __ffsll((long long)bal2) - 1) - qf; }
        else if (!period) period = qf - prev_last;
        // every line end but the tile's first lies one period behind the one before it
        if (__ballot(has && (any || mlow) && q - qprev != period) != 0) lat_ok = false;
        any = true; prev_last = ql; E += cnt;
    }
    // (the gather of the scatter pass reads up to 17 bytes from a base's position: not in the text's last tiles)
    // (the lattice's next point lies behind the tile: line ends that stop short of the tile's end -- wrapped lines, then a long one -- are no lattice)
    const bool tile_ok = lat_ok && !two && E >= 2 && period >= 33 && p1 + E * period >= ET_TILE && (t + 1) * ET_TILE + 32 <= P.n;
    if (lane == 0) {
        t_reg[t] = tile_ok ? (p1 | (period << 12) | (E << 24) | (acgt ? REG_ACGT : 0u)) : 0u; t_irr[t] = tile_ok ? 0 : 1;
        const u32 tot = ET_TILE - nbytes;
        t_seq[t] = tot; t_ids[t] = 0; t_cmt[t] = 0; t_rec[t] = 0;
        t_tail[t] = any ? ((ET_TILE - 1 - lastpos) | 0x80000000u) : tot;
        t_needf[t] = 0; t_need[t] = 0;
    }
    PureTile r; r.pure = true; r.ok = tile_ok; r.acgt = acgt; r.any = any; r.p1 = p1; r.period = period; r.E = E; r.lastpos = lastpos;
    return r;
}
// The same for k_enc_fused's tiles, in fewer vector instructions (the pass is bound by them): the first, second and last line end from
// the four ballots and a few lane reads; then a tile is regular when no sixteen bytes hold two line ends, every line end lies a whole
// number of periods behind the first, and there are as many as the lattice has points from the first one to the tile's end.
__device__ __forceinline__ PureTile count_plain_tile2(const EncP &P, u64 *t_seq, u64 *t_ids, u64 *t_cmt, u64 *t_rec, u32 *t_tail, u32 *t_reg, u64 *t_irr,
                                                      u32 *t_needf, u64 *t_need, u64 t, const u32 (&eol)[4], bool acgt)
{
    const u32 lane = threadIdx.x & 63;
    const u32 nbytes = (u32)__builtin_amdgcn_readlane((int)wave_scan_inclusive<u32, OpAdd>((u32)(__popc(eol[0]) + __popc(eol[1]) + __popc(eol[2]) + __popc(eol[3]))), 63);
    const u64 B[4] = { __ballot(eol[0] != 0), __ballot(eol[1] != 0), __ballot(eol[2] != 0), __ballot(eol[3] != 0) };
    u32 p1 = 0, q2 = 0, lastpos = 0; int found = 0; bool gotlast = false;      // (all uniform)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u64 bal = B[k];
        if (found < 2 && bal) {
            const int L = __ffsll((long long)bal) - 1;
            const u32 e = (u32)__builtin_amdgcn_readlane((int)eol[k], L), at = (64u * (u32)k + (u32)L) * ET_BYTES;
            if (found == 0) {
                p1 = at + (u32)__ffs((int)e) - 1; found = 1;
                const u32 e2 = e & (e - 1); const u64 b2 = bal & (bal - 1);
                if (e2) { q2 = at + (u32)__ffs((int)e2) - 1; found = 2; }
                else if (b2) { const int L2 = __ffsll((long long)b2) - 1; q2 = (64u * (u32)k + (u32)L2) * ET_BYTES + (u32)__ffs(__builtin_amdgcn_readlane((int)eol[k], L2)) - 1; found = 2; }
            } else { q2 = at + (u32)__ffs((int)e) - 1; found = 2; }
        }
    }
#pragma unroll
    for (int k = 3; k >= 0; k--) {
        const u64 bal = B[k];
        if (!gotlast && bal) {
            const int L = 63 - __clzll((long long)bal);
            lastpos = (64u * (u32)k + (u32)L) * ET_BYTES + 31u - (u32)__clz(__builtin_amdgcn_readlane((int)eol[k], L));
            gotlast = true;
        }
    }
    const bool any = found != 0;
    const u32 period = found == 2 ? q2 - p1 : 0u;
    bool tile_ok = false;
    if (period >= 33 && (t + 1) * ET_TILE + 32 <= P.n) {                 // (uniform)
        const float rP = 1.0f / (float)period;
        u32 nl = (u32)((float)(ET_TILE - 1 - p1) * rP);                  // lattice points from p1 to the tile's end: (4095 - p1) / period + 1
        if (nl * period > ET_TILE - 1 - p1) nl--; else if ((nl + 1) * period <= ET_TILE - 1 - p1) nl++;
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 e = eol[k];
            const u32 dq = (64u * (u32)k + lane) * ET_BYTES + (u32)__ffs((int)e) - 1 - p1;       // (e == 0: not looked at)
            const u32 j = (u32)((float)dq * rP + 0.5f);
            bad = bad || (e != 0 && (j * period != dq || (e & (e - 1)) != 0));
        }
        tile_ok = __ballot(bad) == 0 && nbytes == nl + 1 && nbytes >= 2;
    }
    const u32 E = nbytes;
    if (lane == 0) {
        t_reg[t] = tile_ok ? (p1 | (period << 12) | (E << 24) | (acgt ? REG_ACGT : 0u)) : 0u; t_irr[t] = tile_ok ? 0 : 1;
        const u32 tot = ET_TILE - nbytes;
        t_seq[t] = tot; t_ids[t] = 0; t_cmt[t] = 0; t_rec[t] = 0;
        t_tail[t] = any ? ((ET_TILE - 1 - lastpos) | 0x80000000u) : tot;
        t_needf[t] = 0; t_need[t] = 0;
    }
    PureTile r; r.pure = true; r.ok = tile_ok; r.acgt = acgt; r.any = any; r.p1 = p1; r.period = period; r.E = E; r.lastpos = lastpos;
    return r;
}
__device__ __forceinline__ void count_pure_tile(const EncP &P, const i64 *tile_eol, u64 *t_seq, u64 *t_ids, u64 *t_cmt, u64 *t_rec, u32 *t_tail, u32 *t_reg, u64 *t_irr,
                                                u32 *t_needf, u64 *t_need, u64 t, bool inside, const uint4 (&v)[4])
{
    const u32 lane = threadIdx.x & 63;
    if (!inside || !tile_may_be_pure(P, tile_eol, t)) { if (lane == 0) { t_needf[t] = 1; t_need[t] = 1; } return; }
    u32 eol[4]; bool acgt = false, lower_any = false;
    if (!plain_tile(P, v, eol, acgt, lower_any)) { if (lane == 0) { t_needf[t] = 1; t_need[t] = 1; } return; }
    note_case(P, lower_any);
    count_plain_tile(P, t_seq, t_ids, t_cmt, t_rec, t_tail, t_reg, t_irr, t_needf, t_need, t, eol, acgt);
}
// TW tiles per wavefront, the loads of all of them in flight before the first is looked at (two measured slower than one: 2.23 -> 2.55 ms
// per 10 GB, the call 8.0 -> 8.2 ms)
template <u32 WPW, u32 TW>
__global__ __launch_bounds__(64 * WPW) void k_enc_count_pure(EncP P, const i64 *tile_eol, u64 *t_seq, u64 *t_ids, u64 *t_cmt, u64 *t_rec, u32 *t_tail, u32 *t_reg, u64 *t_irr,
                                                         u32 *t_needf, u64 *t_need, u64 tiles)
{
    const u32 lane = threadIdx.x & 63;
    const u64 t0 = ((u64)xcd_block() * WPW + (threadIdx.x >> 6)) * TW;
    // the tile's sixteen bytes per lane and quarter are asked for BEFORE the look at the line in front of the tile (two dependent loads
    // of its own): three memory latencies in a row were two too many for a kernel that does nothing else
    uint4 v[TW][4]; bool inside[TW];
#pragma unroll
    for (u32 j = 0; j < TW; j++) {
        const u64 t = t0 + j;
        inside[j] = t < tiles && t * ET_TILE >= P.p0 && (t + 1) * ET_TILE <= P.n;
        if (inside[j]) {
            const u8 *tp = P.text + t * ET_TILE;
#pragma unroll
            for (int k = 0; k < 4; k++) __builtin_memcpy(&v[j][k], tp + (64u * (u32)k + lane) * ET_BYTES, 16);
        }
    }
#pragma unroll
    for (u32 j = 0; j < TW; j++) if (t0 + j < tiles) count_pure_tile(P, tile_eol, t_seq, t_ids, t_cmt, t_rec, t_tail, t_reg, t_irr, t_needf, t_need, t0 + j, inside[j], v[j]);
}
// ---- ONE pass over the text (process.c:358-427 with encode_dna, encoders.c:30-69, touches an input byte once; so does this).
// A whole 4-bit FASTA input at level 1 (the inputs that may have DIRECT blocks, see direct_word): the count pass of a tile that turns out
// plain, regular and made of A C G T / U only has everything it needs to CODE the tile's bases -- two bits a base, the final Huffman code
// of a direct block -- except where they go: that is the scan of the counts.  So it writes them at TILE-LOCAL positions (base i of the
// tile in bits 2i, 2i + 1 of the tile's KiB of `loc`), and the kernel that moves a direct block's streams to their place in the frame
// anyway (zstd_enc.hip: k_zenc_write_direct_loc) gathers them by the scanned counts with a funnel shift at the tile seams.  The text is
// read once; tiles of blocks that are not direct (an N, a header, a probed MiB, few line ends) are packed by the scatter kernels as
// before (EncOut::loc_mode: only those).  The kernel also leaves every tile's last line end / blank (k_enc_last_fa's tables): the
// look at the line in front of a tile (tile_may_be_pure) needs their scan, so it comes behind the pass (k_pure_check) and hands the
// few tiles that begin in a header line to k_enc_count.
#define LOC_TILE 1024
// two bits per base: log2 of its one-hot 4-bit code (T 0, G 1, C 2, A 3; tables.c:189-197) from bits 1..2 of the letter (A 0, C 1, T / U 2, G 3),
// four letters -> eight bits, the first letter lowest
__device__ __forceinline__ u32 code2x4(u32 x) { const u32 y = (x >> 1) & 0x03030303u; return swar_dot4(y ^ 0x03030303u ^ ((y >> 1) & 0x01010101u), 0x40100401u, 0); }
__device__ __forceinline__ u32 last_pos_pair(u32 eol, u32 sp, u32 at)
{
    return (eol ? at + 1 + (31 - __clz((int)eol)) : 0u) | ((sp ? at + 1 + (31 - __clz((int)sp)) : 0u) << 16);
}
template <bool LOC>
__device__ __forceinline__ void fused_tile(const EncP &P, i64 *tile_eol, i64 *tile_sp, u64 *t_seq, u64 *t_ids, u64 *t_cmt, u64 *t_rec, u32 *t_tail, u32 *t_reg, u64 *t_irr,
                                           u32 *t_needf, u64 *t_need, u8 *loc, u8 *t_hist, u8 *locc, u8 *t_lower, u64 t, bool inside, const uint4 (&v)[4], u32 *s_code)
{
    const u32 lane = threadIdx.x;
    const u64 tb = t * ET_TILE;
    u32 eol[4], cw[4]; bool acgt = false, lower_any = false, fast = false, tile_lower = false;
    if (inside) {
        // first look: nothing but A C G T / U in either case and '\n' (every tile of the texts this pass is for): a table look-up and a
        // compare per four bytes, the case bit cleared in the bytes that have bit 6 (so '*' is not taken for '\n': piece_plain's way); line
        // ends are the bytes without bit 6.  The same bits 1..3 select a byte's two-bit code -- log2 of its one-hot 4-bit code: T 0, G 1,
        // C 2, A 3 (tables.c:189-197) from the slots A 0, C 1, T / U 2, G 3, zero for a line end -- and four codes are one byte
        // (code2x4's, by table): a lane's sixteen bytes -> one word
        u32 bad = 0, lacc = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 w[4] = { v[k].x, v[k].y, v[k].z, v[k].w }; u32 f[4], c[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const u32 h = w[i] >> 1, sel = h & 0x07070707u, low = w[i] & h & 0x20202020u;       // low: bit 5 of the bytes that have bits 5 and 6 -- lower-case letters
                bad |= swar_perm(P.shi, P.slo, sel) ^ (w[i] ^ low); f[i] = ~w[i] & 0x40404040u; lacc |= low;
                c[i] = swar_dot4(swar_perm(0u, 0x01000203u, sel), 0x40100401u, 0);
            }
            const u32 lo = swar_dot4(f[1], 0x80402010u, swar_dot4(f[0], 0x08040201u, 0)), hi = swar_dot4(f[3], 0x80402010u, swar_dot4(f[2], 0x08040201u, 0));
            eol[k] = (lo >> 6) | ((hi >> 6) << 8);
            cw[k] = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
        }
        fast = __ballot(bad != 0) == 0;
        acgt = fast;
        if (fast) { lower_any = lacc != 0; tile_lower = __ballot(lower_any) != 0; }
    }
    if (!fast && (!inside || !plain_tile(P, v, eol, acgt, lower_any))) {
        // not k_enc_count_pure's: its last line end and blank (all byte classes), the rest is k_enc_count's
        u32 pp = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 at = (64u * (u32)k + lane) * ET_BYTES;
            PMask pm;
            if (inside) { const u32 w[4] = { v[k].x, v[k].y, v[k].z, v[k].w }; const PieceFlags f = piece_flags(w); pm.eol = f.eol; pm.sp = f.sp; }
            else { const Piece pc = load_piece(P, tb + at); pm = piece_masks(pc); }
            pp = OpPkMaxU16::f<u32>(pp, last_pos_pair(pm.eol, pm.sp, at));
        }
        pp = (u32)__builtin_amdgcn_readlane((int)wave_scan_inclusive<u32, OpPkMaxU16>(pp), 63);
        if (lane == 0) {
            tile_eol[t] = (pp & 0xFFFF) ? (i64)(tb + (pp & 0xFFFF) - 1) : -1; tile_sp[t] = (pp >> 16) ? (i64)(tb + (pp >> 16) - 1) : -1;
            t_needf[t] = 1; t_need[t] = 1;
        }
        return;
    }
    note_case(P, lower_any);
    // (a tile that needed the second look -- '\r' as its line end -- leaves no codes: it must not count as a direct block's tile)
    const PureTile r = fast ? count_plain_tile2(P, t_seq, t_ids, t_cmt, t_rec, t_tail, t_reg, t_irr, t_needf, t_need, t, eol, acgt)
                            : count_plain_tile(P, t_seq, t_ids, t_cmt, t_rec, t_tail, t_reg, t_irr, t_needf, t_need, t, eol, false);
    if (lane == 0) { const i64 le = r.any ? (i64)(tb + r.lastpos) : -1; tile_eol[t] = le; tile_sp[t] = le; }   // (a plain tile's blanks are its line ends)
    if (!LOC || !fast || !r.ok) return;
    // the tile's bytes as codes: a lane's four words of the LDS string; and, in a tile with lower case, their case bits (a bit per byte,
    // sixteen per piece: encoders.c:98-124's test, `>= 96`, is bit 5 of a letter here)
    const u32 cw0 = cw[0];
#pragma unroll
    for (int k = 0; k < 4; k++) s_code[64u * (u32)k + lane] = cw[k];
    u16 *s_case = (u16 *)(s_code + 280);
    if (tile_lower) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 w[4] = { v[k].x, v[k].y, v[k].z, v[k].w }; u32 f[4];
#pragma unroll
            for (int i = 0; i < 4; i++) f[i] = w[i] & (w[i] >> 1) & 0x20202020u;
            const u32 lo = swar_dot4(f[1], 0x80402010u, swar_dot4(f[0], 0x08040201u, 0)), hi = swar_dot4(f[3], 0x80402010u, swar_dot4(f[2], 0x08040201u, 0));
            s_case[64u * (u32)k + lane] = (u16)((lo >> 5) | ((hi >> 5) << 8));
        }
        if (lane == 0) { s_case[256] = 0; s_case[257] = 0; t_lower[t] = 1; }
    }
    // what k_direct_verdict weighs a block by: the pair codes of bytes 0, 1 and 2, 3 of every lane's first piece (128 pairs of the tile's
    // 2 K, none that holds a line end), counted in sixteen bins
    u32 *s_hist = s_code + 264;
    if (lane < 16) s_hist[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!(eol[0] & 3u)) atomicAdd(&s_hist[cw0 & 15u], 1u);
    if (!(eol[0] & 12u)) atomicAdd(&s_hist[(cw0 >> 4) & 15u], 1u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 16) t_hist[t * 16 + lane] = (u8)s_hist[lane];
    // the tile's bases in order: lane l takes bases 64 l .. 64 l + 63, four groups of 16; a group is the 32 bits at twice its first base's
    // text position, less the two bits of the line end when one lies among its 17 bytes (lines hold at least 32 bases: one at most)
    const u32 W = r.period - 1, nb = ET_TILE - r.E;
    u32 wd[4] = { 0, 0, 0, 0 }, cs[4] = { 0, 0, 0, 0 };
    const u32 b0 = 64u * lane;
    if (b0 < nb) {
        u32 x, e;                                                     // text position of the group's first base; bases from it to the next line end
        if (b0 < r.p1) { x = b0; e = r.p1 - b0; }
        else {
            const u32 d = b0 - r.p1;
            u32 k = (u32)((float)d * (1.0f / (float)W));
            if (k * W > d) k--; else if ((k + 1) * W <= d) k++;
            x = b0 + 1 + k; e = W - (d - k * W);
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (b0 + 16u * (u32)g >= nb) break;
            const u32 d0 = s_code[x >> 4], d1 = s_code[(x >> 4) + 1], sh = 2u * (x & 15u);
            const u32 lo = __builtin_amdgcn_alignbit(d1, d0, sh);
            u32 cl = 0;
            if (tile_lower) { const u32 *sc = (const u32 *)s_case; cl = __builtin_amdgcn_alignbit(sc[(x >> 5) + 1], sc[x >> 5], x & 31u); }   // the 17 bytes' case bits
            if (e < 16) {
                const u32 lowm = (1u << (2u * e)) - 1u, up = __builtin_amdgcn_alignbit(d1 >> sh, lo, 2);
                wd[g] = (lo & lowm) | (up & ~lowm);
                const u32 lm1 = (1u << e) - 1u;
                cs[g] = ((cl & lm1) | ((cl >> 1) & ~lm1)) & 0xFFFFu;
                x += 17; e += W - 16;
            } else { wd[g] = lo; cs[g] = cl & 0xFFFFu; x += 16; e -= 16; }
        }
    }
    *(uint4 *)(loc + t * LOC_TILE + lane * 16u) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
    if (tile_lower) *(uint2 *)(locc + t * (LOC_TILE / 2) + lane * 8u) = make_uint2(cs[0] | (cs[1] << 16), cs[2] | (cs[3] << 16));
}
// TW tiles per wavefront, the loads of all of them in flight before the first is looked at
template <bool LOC, u32 TW>
__global__ __launch_bounds__(64) void k_enc_fused(EncP P, i64 *tile_eol, i64 *tile_sp, u64 *t_seq, u64 *t_ids, u64 *t_cmt, u64 *t_rec, u32 *t_tail, u32 *t_reg, u64 *t_irr,
                                                  u32 *t_needf, u64 *t_need, u64 tiles, u8 *loc, u8 *t_hist, u8 *locc, u8 *t_lower)
{
    __shared__ u32 s_code[LOC ? 264 + 16 + 132 : 4];               // the tile's bytes as two-bit codes, in text order (line ends among them); sixteen bins; the bytes' case bits
    const u32 lane = threadIdx.x;
    // (xcd_block: an XCD walks its own eighth of the text, and the table entries of neighbouring tiles -- eight bytes each in eleven
    // arrays -- meet in ONE L2 instead of leaving eight of them as partial lines: 24.5 -> 21.1 ms per 100 GB)
    const u64 t0 = (u64)xcd_block() * TW;
    uint4 v[TW][4]; bool inside[TW];
#pragma unroll
    for (u32 j = 0; j < TW; j++) {
        const u64 tb = (t0 + j) * ET_TILE;
        inside[j] = t0 + j < tiles && tb >= P.p0 && tb + ET_TILE <= P.n;
        if (inside[j]) {
#pragma unroll
            for (int k = 0; k < 4; k++) __builtin_memcpy(&v[j][k], P.text + tb + (64u * (u32)k + lane) * ET_BYTES, 16);
        }
    }
#pragma unroll
    for (u32 j = 0; j < TW; j++) {
        if (t0 + j >= tiles) break;
        fused_tile<LOC>(P, tile_eol, tile_sp, t_seq, t_ids, t_cmt, t_rec, t_tail, t_reg, t_irr, t_needf, t_need, loc, t_hist, locc, t_lower, t0 + j, inside[j], v[j], s_code);
        if (TW > 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    }
}
// behind the scan of the last line ends: a tile k_enc_fused took for pure that begins in a header line is k_enc_count's after all
__global__ void k_pure_check(EncP P, const i64 *tile_eol, u64 tiles, u32 *t_needf, u64 *t_need, u32 *t_reg, u64 *t_irr, const u32 *t_tail)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles || t_needf[t]) return;
    // the tile in front is plain too and holds a line end: the line at this tile's first byte begins among plain bytes -- not with '>'
    if (t && !t_needf[t - 1] && (t_tail[t - 1] & 0x80000000u)) return;
    if (!tile_may_be_pure(P, tile_eol, t)) { t_needf[t] = 1; t_need[t] = 1; t_reg[t] = 0; t_irr[t] = 1; }
}
__global__ void k_need_list(const u32 *t_needf, const u64 *pre, u64 tiles, u32 *list, const u32 *t_reg = nullptr)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < tiles && (t_needf ? t_needf[t] != 0 : t_reg[t] == 0)) list[pre[t]] = (u32)t;       // (t_reg: the FASTQ count's verdict, 0 = needs the general kernel)
}

// --strict (process.c:98-140): the reference dies at the FIRST unexpected byte of the input.  Every unexpected byte reports
// (position, stream kind, byte) as one 64-bit key and the smallest key wins; the record number of the message is then counted
// from the text in front of that position (k_count_starts), on the error path only.
__device__ __forceinline__ unsigned long long strict_key(u64 pos, int kind, u32 ch) { return ((unsigned long long)pos << 11) | ((unsigned long long)kind << 9) | (ch & 0x1FFu); }
// FASTA (fastq == 0): headers that start at or before `pos` ('>' first on its line); FASTQ: line starts at or before `pos`.
__global__ __launch_bounds__(256) void k_count_starts(const u8 *text, u64 p0, u64 pos, int fastq, unsigned long long *out)
{
    __shared__ u64 lds[4];
    u64 n = 0;
    for (u64 q = p0 + (u64)blockIdx.x * 256 + threadIdx.x; q <= pos; q += (u64)gridDim.x * 256) {
        const u32 c = text[q];
        const bool first = q == p0 || c_eol(text[q - 1]);
        if (first && (fastq ? !c_eol(c) : c == '>')) n++;
    }
    n = wg_reduce1<u64, OpAdd>(n, lds);
    if (threadIdx.x == 0 && n) atomicAdd(out, (unsigned long long)n);
}

// ---- K3: scatter --------------------------------------------------------------------------------------------------------
struct EncOut {
    u8 *seq, *ids, *cmt;          // seq = one byte per base (post-replacement; protein / text), ids/comments final streams
    u8 *packed; u32 *casebits;    // 4-bit sequence: packed codes and case bits of the shard's base stream (flush_pack); casebits may be null
    u64 *rec_begin, *rec_end;     // base index where record r's bases start / end
    u64 *unexpected;              // [3][257]
    u64 *longest;                 // max line length
    u64 *lead;                    // bases in front of the first header (a shard that starts inside a record; 0 for a whole input)
    u64 *strict_first;            // --strict: min over strict_key of the unexpected bytes (nullptr otherwise)
    const u64 *t_seq, *t_ids, *t_cmt, *t_rec; const u32 *t_tail; const i64 *tile_eol;
    const u32 *irr_list;          // k_enc_scatter<true>: the tiles that are not regular, in order (workgroup b takes irr_list[b]); nullptr: every tile
    const u32 *t_reg;             // k_enc_count's verdict on a tile: p1 | period << 12 | line ends << 24 when it is regular, else 0;
                                  // bit 31 (k_enc_count_pure): and its letters are A C G T / U only
    const u8 *direct; u32 nd;     // DIRECT blocks (k_direct_blocks; nullptr: none): block b < nd of 32 KiB of the packed stream takes its
                                  // four Huffman streams of 4-bit codes straight from the scatter pass (direct_word), never its packed bytes
    u32 loc_mode;                 // 1: the direct blocks' codes were left by k_enc_fused (tile-local, `loc`): the scatter kernels pack only what lies
                                  // in the other blocks
    const u32 *sparse_list; const u32 *n_sparse;   // loc_mode: the regular tiles that reach into a block that is not direct (k_sparse_list)
};


// The tile's sequence bytes are staged in LDS and leave as aligned 8-byte stores: one-byte scattered stores
// cost a partial-line HBM write each.
__device__ __forceinline__ void flush_tile(u8 *dst, const u8 *stage, u32 n)
{
    u32 head = (u32)((8 - ((uintptr_t)dst & 7)) & 7); if (head > n) head = n;
    if (threadIdx.x < head) dst[threadIdx.x] = stage[threadIdx.x];
    u32 words = (n - head) >> 3;
    const u64 *s64 = (const u64 *)stage;
    for (u32 w = threadIdx.x; w < words; w += blockDim.x) {
        u32 off = head + 8 * w, sh = (off & 7) * 8;
        u64 a = s64[off >> 3], b = s64[(off >> 3) + 1];
        *(u64 *)(dst + off) = sh ? (a >> sh) | (b << (64 - sh)) : a;
    }
    u32 done = head + 8 * words;
    if (threadIdx.x < n - done) dst[done + threadIdx.x] = stage[done + threadIdx.x];
}

// ---- 4-bit pack and case bits straight from the staged bases (encoders.c:30-69, tables.c:189-197) ------------------------------------
// The base stream is cut into GROUPS of 16 bases: 8 packed bytes (base 2i in the low nibble, encoders.c:44-57) and 16 case bits
// (bit i = base i is >= 96, the test of extract_mask, encoders.c:98-124).  A tile writes the groups that lie wholly inside it with
// plain stores; the two groups it may share with its neighbours were zeroed by k_pack_edges_zero and take their part with an atomic OR.
// Bytes in a 4-bit sequence stream are accepted letters, '-', the replacement 'N' and the '?' of a broken ID (process.c:366): the
// table look-up on the low five bits is exact for those ('-' and '?' have bit 6 clear and differ in bit 4).
__device__ __forceinline__ u32 nuc4x4(u32 x, const u32 *t)
{
    const u32 sel = x & 0x07070707u;
    const u32 r0 = swar_perm(t[1], t[0], sel), r1 = swar_perm(t[3], t[2], sel), r2 = swar_perm(t[5], t[4], sel), r3 = swar_perm(t[7], t[6], sel);
    const u32 m3 = ((x >> 3) & 0x01010101u) * 0xFFu, m4 = ((x >> 4) & 0x01010101u) * 0xFFu, m6 = ((x >> 6) & 0x01010101u) * 0xFFu;
    const u32 lo = (r0 & ~m3) | (r1 & m3), hi = (r2 & ~m3) | (r3 & m3);
    const u32 tab = (lo & ~m4) | (hi & m4);
    return (tab & m6) | (m4 & 0x0F0F0F0Fu & ~m6);
}
// A C G T/U N in either case: the code sits in the slot (c >> 1) & 7 (A 0, C 1, T/U 2, G 3, N 7)
__device__ __forceinline__ u32 nuc4x4_quick(u32 x) { return swar_perm(0x0F0F0F0Fu, 0x02010408u, (x >> 1) & 0x07070707u); }
__device__ __forceinline__ u32 pack_codes8(u32 c0, u32 c1)               // eight codes, one per byte -> four packed bytes
{
    const u32 t0 = c0 | (c0 >> 4), t1 = c1 | (c1 >> 4);
    return swar_perm(t1, t0, 0x06040200u);
}
__device__ __forceinline__ bool all_quick16(const u32 w[4], u32 qlo, u32 qhi)
{
    const u32 L = 0x7F7F7F7Fu; u32 bad = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { u32 x = w[i], r = swar_perm(qhi, qlo, (x >> 1) & 0x07070707u), d = r ^ (x & 0xDFDFDFDFu); bad |= ((d & L) + L) | d; }
    return (bad & 0x80808080u) == 0;
}
template <bool KNOWN_QUICK>
__device__ __forceinline__ void flush_pack(const EncP &P, u8 *packed, u32 *casebits, u64 tbase, u32 n, const u8 *stage /* 16-aligned, base tbase at stage[tbase & 15] */)
{
    if (!n) return;
    const u32 o = (u32)(tbase & 15), span = o + n, ng = (span + 15) >> 4;
    const u64 G0 = tbase >> 4;
    for (u32 j = threadIdx.x; j < ng; j += blockDim.x) {
        const uint4 v = *(const uint4 *)(stage + 16 * j);
        const u32 w[4] = { v.x, v.y, v.z, v.w };
        u32 cd[4];
        if (KNOWN_QUICK || all_quick16(w, P.qlo, P.qhi)) {
#pragma unroll
            for (int i = 0; i < 4; i++) cd[i] = nuc4x4_quick(w[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) cd[i] = nuc4x4(w[i], P.nuc32);
        }
        u64 pk = (u64)pack_codes8(cd[0], cd[1]) | ((u64)pack_codes8(cd[2], cd[3]) << 32);
        const u32 H = 0x80808080u;
        u32 cb = swar_movemask16((v.x | ((v.x << 1) & (v.x << 2))) & H, (v.y | ((v.y << 1) & (v.y << 2))) & H,
                                 (v.z | ((v.z << 1) & (v.z << 2))) & H, (v.w | ((v.w << 1) & (v.w << 2))) & H);
        const u32 a = j == 0 ? o : 0u, b = span - 16 * j < 16 ? span - 16 * j : 16u;
        if (a == 0 && b == 16) {
            *(u64 *)(packed + 8 * (G0 + j)) = pk;
            if (casebits) ((u16 *)casebits)[G0 + j] = (u16)cb;
        } else {
            const u64 nm = (b == 16 ? ~0ull : ((1ull << (4 * b)) - 1)) & ~((1ull << (4 * a)) - 1);
            cb &= ((1u << b) - 1) & ~((1u << a) - 1);
            atomicOr((unsigned long long *)(packed + 8 * (G0 + j)), (unsigned long long)(pk & nm));
            if (casebits) atomicOr(casebits + ((G0 + j) >> 1), cb << (16 * (u32)((G0 + j) & 1)));
        }
    }
}
// Which blocks of 32 KiB of the packed stream are DIRECT (see direct_word), one wavefront per block, before any base is packed:
//   - every tile that holds one of the block's 65536 bases is regular and made of A C G T / U only (REG_ACGT): the block is pairs of
//     those four and line ends sit where the scatter pass computes them;
//   - the entropy of its pair codes, estimated from 64 bytes of every kilobyte of its text (2 K pairs), is above 1 - 1/prefer_flat of
//     four bits: Huffman coding could not save the encoder's threshold (the same test k_zenc_flat_scan makes on the packed bytes);
//   - it is not in a region the level-1 look at the stream reads (zenc_repeat_probe: 1 MiB in every 64, as PACKED bytes).
// t_seq is the exclusive scan of the tiles' base counts with the total behind it.
#define DIRECT_PROBE_BLOCKS_LOG 5                                 // 32 blocks of 32 KiB per probed MiB, one MiB in zenc_probe_every()
__global__ __launch_bounds__(256) void k_direct_blocks(const u8 *text, const u64 *t_seq, const u32 *t_reg, u64 tiles, u32 nd, u32 prefer_flat, int probed, u8 *direct, const u32 *blk_t0 = nullptr, i32 *blk_bnd = nullptr)
{
    __shared__ u32 bins[4][64];                                   // 4 copies x 16 bins per wave
    const u32 wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 b = blockIdx.x * 4 + wv;
    bins[wv][lane] = 0;
    __syncthreads();
    bool ok = b < nd;
    if (ok && probed && (((b >> DIRECT_PROBE_BLOCKS_LOG) & (u32)(probed - 1)) == 0)) ok = false;       // probed: zenc_probe_every of the stream, 0: no look
    u64 t0 = 0, t1 = 0;
    if (ok) {
        const u64 B0 = (u64)b << 16, B1 = B0 + 65536;
        // last tile with t_seq[t] <= B0: 64-ary search (blk_t0: looked up, k_irregular_list)
        u64 lo = 0, hi = tiles;                                   // answer in [lo, hi)
        if (blk_t0) { lo = blk_t0[b]; hi = lo + 1; }
        while (hi - lo > 1) {
            const u64 step = (hi - lo + 63) / 64;
            const u64 p = lo + (u64)lane * step;
            const bool le = p < hi && t_seq[p] <= B0;
            const u32 cnt = (u32)__popcll(__ballot(le));          // (t_seq[lo] <= B0 always: cnt >= 1)
            const u64 nlo = lo + (u64)(cnt - 1) * step, nhi = nlo + step < hi ? nlo + step : hi;
            lo = nlo; hi = nhi;
        }
        t0 = lo;
        // the tiles behind it that hold bases below B1 (a regular tile holds ~4000: 17 of them at most)
        const u64 t = t0 + lane;
        const u64 ts = t <= tiles ? t_seq[t] : ~0ull;
        const bool in = t < tiles && ts < B1;
        // (k_zenc_write_direct_loc: the base counts in front of the block's tiles, less its first base)
        if (blk_bnd && lane < ZENC_LOC_BND) { const i64 rel = (i64)ts - (i64)B0; blk_bnd[(u64)b * ZENC_LOC_BND + lane] = t <= tiles && rel < (1 << 30) ? (i32)rel : (i32)(1 << 30); }
        const u64 bal = __ballot(in);
        const bool good = !in || ((t_reg[t] & REG_ACGT) != 0);
        ok = __ballot(!good) == 0 && bal != ~0ull;                // (64 tiles and still not at B1: some hold few bases, leave it)
        t1 = t0 + (u64)__popcll(bal) - 1;
    }
    if (ok) {
        const u64 x0 = t0 * ET_TILE, span = (t1 + 1 - t0) * ET_TILE;    // (the tiles are regular: wholly inside the text)
        const u8 *at = text + x0 + (((u64)lane * span) >> 6 & ~63ull);
        uint4 v[4]; __builtin_memcpy(v, at, 64);
        const u32 w[16] = { v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w, v[2].x, v[2].y, v[2].z, v[2].w, v[3].x, v[3].y, v[3].z, v[3].w };
        u32 *mybins = &bins[wv][(lane & 3) * 16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const u32 pr = (w[i] >> (16 * h)) & 0xFFFFu, c0 = pr & 0xFF, c1 = pr >> 8;
                if ((c0 & 0x40) && (c1 & 0x40)) atomicAdd(&mybins[((c0 >> 1) & 3) | (((c1 >> 1) & 3) << 2)], 1u);   // two letters (a line end has no bit 6); bits 1..2 tell A C G T apart
            }
        }
    }
    __syncthreads();
    if (ok) {
        const u32 c = lane < 16 ? bins[wv][lane] + bins[wv][16 + lane] + bins[wv][32 + lane] + bins[wv][48 + lane] : 0u;
        u32 ns = c;
#pragma unroll
        for (u32 d = 1; d < 16; d <<= 1) ns += __shfl_xor((int)ns, d, 64);
        float h = c ? (float)c * __log2f((float)ns / (float)c) : 0.0f;
#pragma unroll
        for (u32 d = 1; d < 16; d <<= 1) h += __shfl_xor(h, d, 64);
        ns = (u32)__shfl((int)ns, 0, 64); h = __shfl(h, 0, 64);
        ok = ns >= 1024 && h * (float)prefer_flat > 4.0f * (float)ns * (float)(prefer_flat - 1);
    }
    if (b < nd && lane == 0) direct[b] = ok ? 1 : 0;
}
// The same verdict a THREAD per block, from what k_enc_fused left per tile (t_hist: sixteen counts of sampled pair codes): no look at
// the text, and everything a block needs lies at addresses known once its first tile is (blk_t0) -- a wavefront per block that samples
// the text was 2.1 ms per 100 GB of dependent loads.  A direct block's tiles are regular (at least 3971 bases each): nineteen at most.
// Also writes blk_bnd (k_zenc_write_direct_loc).
__global__ __launch_bounds__(256) void k_direct_verdict(const u64 *t_seq, const u32 *t_reg, const u8 *t_hist, u64 tiles, u32 nd, u32 prefer_flat, int probed, const u32 *blk_t0, u8 *direct, i32 *blk_bnd)
{
    const u32 b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nd) return;
    const u64 B0 = (u64)b << 16, B1 = B0 + 65536, t0 = blk_t0[b];
    bool ok = !(probed && (((b >> DIRECT_PROBE_BLOCKS_LOG) & (u32)(probed - 1)) == 0));
    u64 ts[ZENC_LOC_BND];
#pragma unroll
    for (u32 i = 0; i < ZENC_LOC_BND; i++) ts[i] = t0 + i <= tiles ? t_seq[t0 + i] : ~0ull;
#pragma unroll
    for (u32 i = 0; i < ZENC_LOC_BND; i++) { const i64 rel = (i64)ts[i] - (i64)B0; blk_bnd[(u64)b * ZENC_LOC_BND + i] = ts[i] != ~0ull && rel < (1 << 30) ? (i32)rel : (i32)(1 << 30); }
    u32 acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };                    // sixteen 16-bit sums
    bool end = false;
#pragma unroll
    for (u32 i = 0; i < ZENC_LOC_BND - 1; i++) {
        if (!end && ok) {
            if (t0 + i >= tiles || ts[i] >= B1) end = true;
            else if (!(t_reg[t0 + i] & REG_ACGT)) ok = false;
            else {
                const uint4 h = *(const uint4 *)(t_hist + (t0 + i) * 16);
                acc[0] += h.x & 0x00FF00FFu; acc[1] += (h.x >> 8) & 0x00FF00FFu; acc[2] += h.y & 0x00FF00FFu; acc[3] += (h.y >> 8) & 0x00FF00FFu;
                acc[4] += h.z & 0x00FF00FFu; acc[5] += (h.z >> 8) & 0x00FF00FFu; acc[6] += h.w & 0x00FF00FFu; acc[7] += (h.w >> 8) & 0x00FF00FFu;
            }
        }
    }
    ok = ok && end;
    if (ok) {
        u32 ns = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) ns += (acc[k] & 0xFFFFu) + (acc[k] >> 16);
        float h = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u32 c0 = acc[k] & 0xFFFFu, c1 = acc[k] >> 16;
            if (c0) h += (float)c0 * __log2f((float)ns / (float)c0);
            if (c1) h += (float)c1 * __log2f((float)ns / (float)c1);
        }
        ok = ns >= 1024 && h * (float)prefer_flat > 4.0f * (float)ns * (float)(prefer_flat - 1);
    }
    direct[b] = ok ? 1 : 0;
}
// The case bits of the direct blocks' bases when the text has lower case (k_enc_fused left them tile by tile like the codes: base i of
// tile t in bit i of the 512 bytes at locc + 512 t, tiles without lower case not at all -- t_lower): a thread per 64 bases, the word of
// the global bit string (extract_mask's input, encoders.c:126-146) funnelled out of the tile that holds its first base and, across a
// tile's end, the next one.  The blocks that are not direct get theirs from the scatter kernels as before.
__global__ __launch_bounds__(256) void k_case_gather(const u8 *direct, u32 nd, const u32 *blk_t0, const i32 *blk_bnd, const u8 *locc, const u8 *t_lower, u64 *casebits)
{
    __shared__ i32 s_bnd[ZENC_LOC_BND]; __shared__ u8 s_low[ZENC_LOC_BND];
    const u32 b = blockIdx.x;                                         // a workgroup a block: 1024 words of 64 bases, four a thread
    if (b >= nd || !direct[b]) return;
    const u64 t0 = blk_t0[b];
    if (threadIdx.x < ZENC_LOC_BND) { s_bnd[threadIdx.x] = blk_bnd[(u64)b * ZENC_LOC_BND + threadIdx.x]; s_low[threadIdx.x] = t_lower[t0 + threadIdx.x]; }   // (t_lower has a tile's byte for every tile a block can reach: ZENC_LOC_BND behind the last)
    __syncthreads();
    u32 a[4], c[4], e[4], sh[4], av[4], kk[4]; u64 nx[4];
#pragma unroll
    for (u32 r = 0; r < 4; r++) {
        const u32 W = r * 256u + threadIdx.x;
        const i32 r0 = (i32)(64u * W);
        u32 k = (u32)(r0 - s_bnd[0]) >> 12;
        if (s_bnd[k + 1] <= r0) k++;
        if (s_bnd[k + 1] <= r0) k++;
        const u32 i = (u32)(r0 - s_bnd[k]);
        av[r] = (u32)(s_bnd[k + 1] - r0); kk[r] = k; sh[r] = i & 31u;
        // (asked for whether the tile has lower case or not -- its 512 bytes exist either way --: the loads do not wait for the flag)
        const u32 *p = (const u32 *)(locc + (t0 + k) * (LOC_TILE / 2)) + (i >> 5);
        a[r] = p[0]; c[r] = p[1]; e[r] = p[2];
        nx[r] = av[r] < 64u ? ld64(locc + (t0 + k + 1) * (LOC_TILE / 2)) : 0ull;
    }
#pragma unroll
    for (u32 r = 0; r < 4; r++) {
        u64 w = s_low[kk[r]] ? (u64)__builtin_amdgcn_alignbit(c[r], a[r], sh[r]) | ((u64)__builtin_amdgcn_alignbit(e[r], c[r], sh[r]) << 32) : 0ull;
        if (av[r] < 64u) {                                            // the tile ends inside the word: the rest from the next tile's first bases
            w &= (1ull << av[r]) - 1;
            if (s_low[kk[r] + 1]) w |= nx[r] << av[r];
        }
        casebits[((u64)b << 10) + r * 256u + threadIdx.x] = w;
    }
}
// the tiles k_enc_count did not find regular, in order (pre = exclusive scan of its 0 / 1 verdicts)
// blk_t0 (may be null): the tile that holds base 65536 b, the first of block b of the packed stream, for every such base there is
// (t_seq: the exclusive scan of the tiles' base counts with the total behind it)
__global__ void k_irregular_list(const u32 *t_reg, const u64 *pre, u64 tiles, u32 *list, const u64 *t_seq = nullptr, u32 *blk_t0 = nullptr)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles) return;
    if (!t_reg[t]) list[pre[t]] = (u32)t;
    if (blk_t0) {
        const u64 lo = t_seq[t], hi = t_seq[t + 1];
        for (u64 b = (lo + 65535) >> 16; (b << 16) < hi; b++) blk_t0[b] = (u32)t;
    }
}
// the groups a tile shares with its neighbours (and the last group of the stream, whose padding must read as zero)
__global__ void k_pack_edges_zero(const u64 *t_seq, u64 tiles, u64 T, u8 *packed, u32 *casebits, const u8 *direct, u32 nd, u32 loc_mode = 0)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles) return;
    const u64 b0 = t_seq[t], b1 = t + 1 < tiles ? t_seq[t + 1] : T;
    if (b1 <= b0) return;
    const u64 g0 = b0 >> 4, g1 = (b1 - 1) >> 4;
    // (a group of a direct block is one word of its stream, see direct_word)
    const bool d0 = direct && (g0 >> 12) < nd && direct[g0 >> 12], d1 = direct && (g1 >> 12) < nd && direct[g1 >> 12];
    if (loc_mode) {                                               // (a direct block's codes are not in `packed` at all, its case bits are k_case_gather's)
        if (!d0) { *(u64 *)(packed + 8 * g0) = 0; if (casebits) ((u16 *)casebits)[g0] = 0; }
        if (!d1) { *(u64 *)(packed + 8 * g1) = 0; if (casebits) ((u16 *)casebits)[g1] = 0; }
        return;
    }
    if (d0) *((u32 *)(packed + ((g0 >> 12) << 15) + (((g0 >> 10) & 3) << 12)) + (1023u - (u32)(g0 & 1023u))) = 0; else *(u64 *)(packed + 8 * g0) = 0;
    if (d1) *((u32 *)(packed + ((g1 >> 12) << 15) + (((g1 >> 10) & 3) << 12)) + (1023u - (u32)(g1 & 1023u))) = 0; else *(u64 *)(packed + 8 * g1) = 0;
    if (casebits) { ((u16 *)casebits)[g0] = 0; ((u16 *)casebits)[g1] = 0; }
}

__device__ __forceinline__ u64 low_bytes(u32 n) { return n >= 8 ? ~0ull : ((1ull << (8 * n)) - 1); }   // lowest n bytes set
// first n (<= 16) bytes of {lo, hi} to LDS at any alignment (gfx950 handles unaligned ds accesses)
__device__ __forceinline__ void lds_store_n(u8 *p, u64 lo, u64 hi, u32 n)
{
    if (n >= 16) { __builtin_memcpy(p, &lo, 8); __builtin_memcpy(p + 8, &hi, 8); return; }
    if (n & 8) { __builtin_memcpy(p, &lo, 8); p += 8; lo = hi; }
    if (n & 4) { u32 v = (u32)lo; __builtin_memcpy(p, &v, 4); p += 4; lo >>= 32; }
    if (n & 2) { u16 v = (u16)lo; __builtin_memcpy(p, &v, 2); p += 2; lo >>= 16; }
    if (n & 1) *p = (u8)lo;
}

// bytes [a, b) of the piece without their space-class bytes, moved down to byte 0; returns how many are left
__device__ __forceinline__ u32 piece_run_compact(const Piece &pc, u32 a, u32 b, u32 sp, u64 &lo, u64 &hi)
{
    piece_from(pc, a, lo, hi);
    u32 d = (sp >> a) & ((1u << (b - a)) - 1);
    const u32 n = (b - a) - (u32)__popc(d);
    while (d) {                                                   // highest first, so the lower positions stay put
        u32 k = 31 - __clz((int)d); d &= ~(1u << k);
        u64 slo = (lo >> 8) | (hi << 56), shi = hi >> 8;
        if (k < 8) { u64 m = low_bytes(k); lo = (lo & m) | (slo & ~m); hi = shi; }
        else { u64 m = low_bytes(k - 8); hi = (hi & m) | (shi & ~m); }
    }
    return n;
}
// first n (<= 16) bytes of {lo, hi} to global memory at any alignment
__device__ __forceinline__ void global_store_n(u8 *p, u64 lo, u64 hi, u32 n)
{
    if (n >= 16) { st64(p, lo); st64(p + 8, hi); return; }
    if (n & 8) { st64(p, lo); p += 8; lo = hi; }
    if (n & 4) { st32(p, (u32)lo); p += 4; lo >>= 32; }
    if (n & 2) { p[0] = (u8)lo; p[1] = (u8)(lo >> 8); p += 2; lo >>= 16; }
    if (n & 1) p[0] = (u8)lo;
}

struct WriteSink {
    const EncOut &O; u64 bseq, bids, bcmt, rec; u64 line_b; u64 best; bool line_valid; u8 *stage; u64 tbase;
    __device__ WriteSink(const EncOut &o) : O(o) {}
    __device__ void emit(int s, u32 ch) { if (s == EV_SEQ) stage[bseq++ - tbase] = (u8)ch; else if (s == EV_IDS) O.ids[bids++] = (u8)ch; else O.cmt[bcmt++] = (u8)ch; }
    __device__ void header_start(u64) { if (rec > 0) O.rec_end[rec - 1] = bseq; else *O.lead = bseq; rec++; }
    __device__ void header_end(u64) { O.rec_begin[rec - 1] = bseq; line_b = bseq; }
    __device__ void line_end(u64) { u64 len = bseq - line_b; if (len > best) best = len; line_b = bseq; }
    __device__ void unexpected(int kind, u32 ch, u64 pos) { atomicAdd((unsigned long long *)&O.unexpected[kind * 257 + ch], 1ull); if (O.strict_first) atomicMin((unsigned long long *)O.strict_first, strict_key(pos, kind, ch)); }
    __device__ void ids_range(const Piece &pc, u32 a, u32 b) { u64 lo, hi; piece_from(pc, a, lo, hi); global_store_n(O.ids + bids, lo, hi, b - a); bids += b - a; }
    __device__ void cmt_range(const Piece &pc, u32 a, u32 b) { u64 lo, hi; piece_from(pc, a, lo, hi); global_store_n(O.cmt + bcmt, lo, hi, b - a); bcmt += b - a; }
    __device__ void seq_range(const Piece &pc, u32 a, u32 b, u32 sp) { u64 lo, hi; u32 n = piece_run_compact(pc, a, b, sp, lo, hi); lds_store_n(stage + (bseq - tbase), lo, hi, n); bseq += n; }
    __device__ void term(int st) { emit(st, 0); }
};

// base count at the start of the line in progress where the tile begins (the line began in an earlier tile t', the one holding the
// last EOL in front of this tile: B = t_seq[t'+1] - tail(t'))
__device__ __forceinline__ u64 tile_line_base(const EncP &P, const EncOut &O, const i64 *tile_eol, u64 tile)
{
    const i64 le = tile ? tile_eol[tile - 1] : -1;
    if (le < 0 || (u64)(le + 1) <= P.p0) return 0;
    const u64 tp = (u64)le / ET_TILE;
    return O.t_seq[tp + 1] - (O.t_tail[tp] & 0x7FFFFFFFu);
}

// ---- REGULAR tiles (k_enc_count's verdict, t_reg): base b of the tile is the text byte b + (b >= p1 ? 1 + (b - p1) / (period - 1) : 0).
// The 16 bases of an output group are fetched straight from the text (16 bytes, 17 with the line end taken out) and packed: no piece,
// no classes, no prefix sum over the lanes, no compaction, no LDS, no barrier.  One wavefront per tile, a lane takes FOUR consecutive
// groups aligned to four (64 bases): 32 bytes of codes and 8 bytes of case bits leave as two 16-byte stores and one 8-byte store --
// with a group per lane the 8-byte and 2-byte stores were two thirds of the pass (6.0 ms; 2.1 ms without any store, 4.3 without the
// 2-byte ones).  Quads that reach over the tile's ends go group by group (atomic OR into the groups shared with the neighbours).
// Tiles that are not regular are skipped here and done by k_enc_scatter, which skips the regular ones.
#define REG_TPW 4
// workgroups of one wavefront for the kernels that work a wavefront per tile anyway (NAF_GPU_ENC_WAVE=0: four wavefronts per workgroup)
static bool enc_wave_wg(const naf_gpu_ctx *c) { const char *e = ctx_opt(c, "ENC_WAVE"); return !(e && e[0] == '0'); }
struct RegGroup { u32 ga, gb, e, nb; };
__device__ __forceinline__ void reg_group_geometry(u32 j, u32 o, u32 span, u32 p1, u32 W, float rW, RegGroup &g, u32 &x)
{
    g.ga = j == 0 ? o : 0u; g.gb = span - 16 * j < 16 ? span - 16 * j : 16u;      // bytes [ga, gb) of the group are this tile's
    const u32 b_lo = 16 * j + g.ga - o; g.nb = g.gb - g.ga;
    if (b_lo < p1) { x = b_lo; g.e = p1 - b_lo; }                                   // text position of base b_lo; bases from it to the next line end
    else {
        const u32 d = b_lo - p1;
        u32 k = (u32)((float)d * rW);
        if (k * W > d) k--; else if ((k + 1) * W <= d) k++;
        x = b_lo + 1 + k; g.e = W - (d - k * W);
    }
}
// the group's 16 bases in place (line end taken out, moved up behind the bytes of the tile in front) -> packed codes and case bits
__device__ __forceinline__ void reg_group_pack(const RegGroup &g, u64 lo, u64 hi, u32 c16, u64 &pk, u32 &cb)
{
    if (g.e < g.nb) {                                                 // the line end at byte e: everything behind it one down, byte 16 comes in
        const u64 slo = (lo >> 8) | (hi << 56), shi = (hi >> 8) | ((u64)c16 << 56);
        if (g.e < 8) { const u64 m = low_bytes(g.e); lo = (lo & m) | (slo & ~m); hi = shi; }
        else { const u64 m = low_bytes(g.e - 8); hi = (hi & m) | (shi & ~m); }
    }
    if (g.ga) {                                                       // the group's first bytes belong to the tile in front: up by ga bytes
        const u32 sh = 8 * g.ga;
        if (sh < 64) { hi = (hi << sh) | (lo >> (64 - sh)); lo <<= sh; } else { hi = lo << (sh - 64); lo = 0; }
    }
    const u32 gw[4] = { (u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32) };
    u32 cd[4];
#pragma unroll
    for (int i = 0; i < 4; i++) cd[i] = nuc4x4_quick(gw[i]);
    pk = (u64)pack_codes8(cd[0], cd[1]) | ((u64)pack_codes8(cd[2], cd[3]) << 32);
    const u32 H = 0x80808080u;
    cb = swar_movemask16((gw[0] | ((gw[0] << 1) & (gw[0] << 2))) & H, (gw[1] | ((gw[1] << 1) & (gw[1] << 2))) & H,
                         (gw[2] | ((gw[2] << 1) & (gw[2] << 2))) & H, (gw[3] | ((gw[3] << 1) & (gw[3] << 2))) & H);
}
// ---- DIRECT blocks.  A block of 32 KiB of the packed stream that holds nothing but pairs of A C G T and is coded with the flat tree of
// the sixteen pair codes (zstd_enc.hip: k_zenc_flat_scan) is four Huffman streams of 8192 4-bit codes each: the code of a pair is
// 4 log2(second base) + log2(first base), i.e. TWO BITS PER BASE, and a stream holds its codes last symbol first (4.2.2).  A group of
// 16 bases (8 symbols) is then one 32-bit word of its stream -- word 1023 - (group & 1023), symbol s of the group in nibble 7 - s --
// and the scatter pass writes that word instead of the group's eight packed bytes: the block's streams are ready where its packed bytes
// would have been (stream k in bytes [4096 k, 4096 k + 4096) of the block's 32 KiB; k_zenc_write copies them out and adds the end marks),
// 5 GB of 10 GB of bases less to write here and to read twice there.
__device__ __forceinline__ u32 direct_code32(u64 pk)
{
    const u64 lg = ((pk >> 1) & 0x7777777777777777ull) - ((pk >> 3) & 0x1111111111111111ull);      // log2 of a one-hot nibble x: (x >> 1) - (x >> 3)
    u64 x = lg & 0x3333333333333333ull;                                                            // two bits of every nibble, moved together
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0xFFFFFFFFull;
    u32 r = __builtin_bswap32((u32)x);                                                             // nibble s -> nibble 7 - s
    return ((r & 0x0F0F0F0Fu) << 4) | ((r >> 4) & 0x0F0F0F0Fu);
}
__device__ __forceinline__ u32 *direct_word(u8 *packed, u64 G)
{
    return (u32 *)(packed + ((G >> 12) << 15) + (((G >> 10) & 3) << 12)) + (1023u - (u32)(G & 1023u));
}
__device__ __forceinline__ bool direct_group(const EncOut &O, u64 G) { const u64 b = G >> 12; return O.direct && b < O.nd && O.direct[b]; }
__device__ __forceinline__ void reg_group_store(const EncOut &O, u64 G, const RegGroup &g, u64 pk, u32 cb)
{
    if (direct_group(O, G)) {
        if (O.loc_mode) return;
        if (g.ga == 0 && g.gb == 16) { *direct_word(O.packed, G) = direct_code32(pk); if (O.casebits) ((u16 *)O.casebits)[G] = (u16)cb; }
        else {
            const u64 nm = (g.gb == 16 ? ~0ull : ((1ull << (4 * g.gb)) - 1)) & ~((1ull << (4 * g.ga)) - 1);
            cb &= ((1u << g.gb) - 1) & ~((1u << g.ga) - 1);
            atomicOr(direct_word(O.packed, G), direct_code32(pk & nm));          // (an absent base adds no bit: log2 of an empty nibble reads 0)
            if (O.casebits) atomicOr(O.casebits + (G >> 1), cb << (16 * (u32)(G & 1)));
        }
        return;
    }
    if (g.ga == 0 && g.gb == 16) {
        *(u64 *)(O.packed + 8 * G) = pk;
        if (O.casebits) ((u16 *)O.casebits)[G] = (u16)cb;
    } else {
        const u64 nm = (g.gb == 16 ? ~0ull : ((1ull << (4 * g.gb)) - 1)) & ~((1ull << (4 * g.ga)) - 1);
        cb &= ((1u << g.gb) - 1) & ~((1u << g.ga) - 1);
        atomicOr((unsigned long long *)(O.packed + 8 * G), (unsigned long long)(pk & nm));
        if (O.casebits) atomicOr(O.casebits + (G >> 1), cb << (16 * (u32)(G & 1)));
    }
}
__device__ __forceinline__ void scatter_regular_tile(const EncP &P, const i64 *tile_eol, const EncOut &O, u64 t, u64 *spk, u16 *scb)
{
    // per wavefront: the tile's packed groups and case bits on their way from "a group per lane" (loads of neighbouring lanes touch
    // neighbouring bytes) to "four groups per lane" (wide stores); slot = group index inside the tile + lead
    const u32 lane = threadIdx.x & 63;
    const u32 reg = O.t_reg[t];
    const u64 tb = O.t_seq[t];
    if (!reg) return;
    const u32 p1 = reg & 0xFFFu, period = (reg >> 12) & 0xFFFu, E = REG_E(reg), W = period - 1, n = ET_TILE - E;
    const u32 o = (u32)(tb & 15), span = o + n, ng = (span + 15) >> 4;
    const u64 G0 = tb >> 4;
    const u32 lead = (u32)(G0 & 3);                                    // groups of the tile's first quad that belong to tiles in front
    const u32 nq = (lead + ng + 3) >> 2;
    const float rW = 1.0f / (float)W;
    const u8 *tt = P.text + t * ET_TILE;
    // phase 1: geometry and loads of groups lane, lane + 64, lane + 128, lane + 192 (all in flight together)
    RegGroup g[4]; u64 lo[4], hi[4]; u32 c16[4]; bool in[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u32 j = 64 * (u32)i + lane;
        in[i] = j < ng;
        u32 x = 0;
        if (in[i]) reg_group_geometry(j, o, span, p1, W, rW, g[i], x); else { g[i].ga = g[i].gb = g[i].e = g[i].nb = 0; }
        const u8 *at = tt + x;
        lo[i] = ld64(at); hi[i] = ld64(at + 8); c16[i] = at[16];
    }
    // phase 2: pack; whole groups wait in LDS, the two the tile shares with its neighbours go out at once
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (!in[i]) continue;
        const u32 j = 64 * (u32)i + lane;
        u64 pk; u32 cb; reg_group_pack(g[i], lo[i], hi[i], c16[i], pk, cb);
        if (g[i].ga == 0 && g[i].gb == 16) { spk[lead + j] = pk; scb[lead + j] = (u16)cb; }
        else reg_group_store(O, G0 + j, g[i], pk, cb);
    }
    if (ng > 256 && lane == 0) {                                       // a 257th group (few line ends and a tile that starts late in its group)
        RegGroup g2; u32 x2; reg_group_geometry(256, o, span, p1, W, rW, g2, x2);
        const u8 *at = tt + x2;
        u64 pk; u32 cb; reg_group_pack(g2, ld64(at), ld64(at + 8), at[16], pk, cb);
        if (g2.ga == 0 && g2.gb == 16) { spk[lead + 256] = pk; scb[lead + 256] = (u16)cb; } else reg_group_store(O, G0 + 256, g2, pk, cb);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // phase 3: four groups per lane, aligned to four: 32 bytes of codes and 8 bytes of case bits per lane
    const u32 j_first_whole = o ? 1u : 0u, j_end_whole = span >> 4;      // groups [j_first_whole, j_end_whole) of the tile are whole
    const u64 Gq0 = G0 & ~3ull;
    for (u32 q = lane; q < nq; q += 64) {
        const u32 s0 = 4 * q;                                           // slot of the quad's first group
        const bool whole = s0 >= lead + j_first_whole && s0 + 4 <= lead + j_end_whole;
        const bool dq = direct_group(O, Gq0 + s0);                      // (a quad lies in one stream of one block)
        if (whole) {
            const uint4 a = *(const uint4 *)(spk + s0), b2 = *(const uint4 *)(spk + s0 + 2);
            if (dq && O.loc_mode) { }
            else if (dq) {
                // the quad's four words, the last group first
                const uint4 cw = make_uint4(direct_code32((u64)b2.z | ((u64)b2.w << 32)), direct_code32((u64)b2.x | ((u64)b2.y << 32)),
                                            direct_code32((u64)a.z | ((u64)a.w << 32)), direct_code32((u64)a.x | ((u64)a.y << 32)));
                *(uint4 *)direct_word(O.packed, Gq0 + s0 + 3) = cw;
            } else {
                uint4 *pp = (uint4 *)(O.packed + 8 * (Gq0 + s0));
                pp[0] = a; pp[1] = b2;
            }
            if (O.casebits && !(dq && O.loc_mode)) *(uint2 *)((u16 *)O.casebits + Gq0 + s0) = *(const uint2 *)(scb + s0);
        } else {
#pragma unroll
            for (u32 i = 0; i < 4; i++) {
                const u32 sl = s0 + i;
                if (sl >= lead + j_first_whole && sl < lead + j_end_whole && !(dq && O.loc_mode)) {
                    if (dq) *direct_word(O.packed, Gq0 + sl) = direct_code32(spk[sl]); else *(u64 *)(O.packed + 8 * (Gq0 + sl)) = spk[sl];
                    if (O.casebits) ((u16 *)O.casebits)[Gq0 + sl] = scb[sl];
                }
            }
        }
    }
    if (lane == 0) {
        // line lengths: the lines inside the tile hold W bases.  The one that ends at p1 began in front of the tile: when that tile is
        // regular with the same period and its last line end lies W bases in front of p1, it holds W too; only otherwise is its
        // start looked up (three dependent loads)
        const u32 rp = t ? O.t_reg[t - 1] : 0u;
        const u32 pl_prev = (rp & 0xFFFu) + (REG_E(rp) - 1) * ((rp >> 12) & 0xFFFu);
        u64 len = W;
        if (!(rp && ((rp >> 12) & 0xFFFu) == period && ET_TILE - 1 - pl_prev + p1 == W)) {
            const u64 first = tb + p1 - tile_line_base(P, O, tile_eol, t);
            if (first > len) len = first;
        }
        if (len > __atomic_load_n(O.longest, __ATOMIC_RELAXED)) atomicMax((unsigned long long *)O.longest, (unsigned long long)len);
    }
}
template <u32 TPW>
__global__ __launch_bounds__(64 * TPW) void k_enc_scatter_regular(EncP P, const i64 *tile_eol, EncOut O, u64 tiles)
{
    __shared__ __attribute__((aligned(16))) u64 s_pk[TPW][264];
    __shared__ __attribute__((aligned(16))) u16 s_cb[TPW][264];
    const u32 wv = threadIdx.x >> 6;
    const u64 t = (u64)blockIdx.x * TPW + wv;                          // the wavefront's tile
    if (t >= tiles) return;
    scatter_regular_tile(P, tile_eol, O, t, s_pk[wv], s_cb[wv]);
}
// loc_mode: the regular tiles that reach into a block that is not direct, from k_sparse_list's list (a bounded grid walks it)
__global__ __launch_bounds__(64) void k_enc_scatter_regular_list(EncP P, const i64 *tile_eol, EncOut O)
{
    __shared__ __attribute__((aligned(16))) u64 s_pk[264];
    __shared__ __attribute__((aligned(16))) u16 s_cb[264];
    const u32 n = *O.n_sparse;
    for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
        scatter_regular_tile(P, tile_eol, O, O.sparse_list[i], s_pk, s_cb);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
// loc_mode: which regular tiles the scatter pass still has to pack (order does not matter), and the longest line of ALL regular tiles
// (scatter_regular_tile's last step, which most of them no longer reach)
__global__ __launch_bounds__(256) void k_sparse_list(EncP P, const i64 *tile_eol, EncOut O, u64 tiles, u32 *list, u32 *n_list)
{
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    const u32 lane = threadIdx.x & 63;
    bool want = false; u64 len = 0;
    if (t < tiles) {
        const u32 reg = O.t_reg[t];
        if (reg) {
            const u64 tb = O.t_seq[t];
            const u32 p1 = reg & 0xFFFu, period = (reg >> 12) & 0xFFFu, E = REG_E(reg), W = period - 1, n = ET_TILE - E;
            want = !(direct_group(O, tb >> 4) && direct_group(O, (tb + n - 1) >> 4));
            const u32 rp = t ? O.t_reg[t - 1] : 0u;
            const u32 pl_prev = (rp & 0xFFFu) + (REG_E(rp) - 1) * ((rp >> 12) & 0xFFFu);
            len = W;
            if (!(rp && ((rp >> 12) & 0xFFFu) == period && ET_TILE - 1 - pl_prev + p1 == W)) {
                const u64 first = tb + p1 - tile_line_base(P, O, tile_eol, t);
                if (first > len) len = first;
            }
        }
    }
    // (one atomic per workgroup: with one per wavefront 38 K of them queued on the one address for a tenth of a millisecond per 10 GB)
    __shared__ u32 s_cnt[4], s_base;
    const u64 bal = __ballot(want);
    const u32 wv = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wv] = (u32)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) { const u32 tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]; s_base = tot ? atomicAdd(n_list, tot) : 0u; }
    __syncthreads();
    if (want) { u32 pre = s_base; for (u32 w = 0; w < wv; w++) pre += s_cnt[w]; list[pre + (u32)__popcll(bal & ((1ull << lane) - 1))] = (u32)t; }
    len = wave_scan_inclusive<u64, OpMaxU64>(len);
    if (lane == 63 && len > __atomic_load_n(O.longest, __ATOMIC_RELAXED)) atomicMax((unsigned long long *)O.longest, (unsigned long long)len);
}

template <bool PACK>
__global__ __launch_bounds__(256, 8) void k_enc_scatter(EncP P, const i64 *tile_eol, const i64 *tile_sp, EncOut O)
{
    const u64 tile = (PACK && O.irr_list) ? O.irr_list[blockIdx.x] : blockIdx.x;   // regular tiles are k_enc_scatter_regular's
    __shared__ __attribute__((aligned(16))) u8 stage[ET_TILE + 48];
    __shared__ u32 s_a[4], s_b[4], s_l[4];
    __shared__ u64 s_best[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 base = (u64)tile * ET_TILE + (u64)threadIdx.x * ET_BYTES;
    const bool maybe = tile_may_be_pure(P, tile_eol, tile);
    const u64 tbase = O.t_seq[tile];
    const u64 line_b0 = tile_line_base(P, O, tile_eol, tile);
    Piece pc = load_piece(P, base);
    // ---- a pure tile (see k_enc_count): prefix of the base counts, the line lengths from the lanes that hold an EOL, bases to LDS,
    // packed codes and case bits out.  No per-lane context, no class table, two barriers.
    {
        const u32 w[4] = { (u32)pc.w0, (u32)(pc.w0 >> 32), (u32)pc.w1, (u32)(pc.w1 >> 32) };
        PMask pm; pm.gt = 0;
        const bool lane_bad = !piece_plain(w, P.plo, P.phi, &pm.eol); pm.sp = pm.eol;
        const u32 nseq = 16u - (u32)__popc(pm.sp);
        const u32 incl = wave_scan_inclusive<u32, OpAdd>(nseq);
        const bool has = pm.eol != 0;
        const u64 bal = __ballot(has);
        const int lastl = bal ? 63 - __clzll((long long)bal) : -1;
        const u32 tail = piece_tail_bases(pm);
        const u32 ew = incl - tail;                               // wave-relative base count at this lane's last EOL (when it has one)
        const bool wave_bad = __ballot(lane_bad) != 0;
        if (lane == 63) { s_a[wave] = incl; s_l[wave] = (bal ? 1u : 0u) | (wave_bad ? 2u : 0u); }
        if (lane == lastl) s_b[wave] = ew;
        __syncthreads();
        if (PACK && maybe && !((s_l[0] | s_l[1] | s_l[2] | s_l[3]) & 2u)) {
            u32 pre = 0, tile_seq = 0;
#pragma unroll
            for (int v = 0; v < 4; v++) { if (v < wave) pre += s_a[v]; tile_seq += s_a[v]; }
            const u32 ex = pre + incl - nseq;                     // tile-relative index of this lane's first base
            u32 best = 0;
            if (has) {
                // base count at the EOL in front of this lane's first one: an earlier lane of the wave, an earlier wave, or the carry
                const u64 mlow = bal & ((1ull << lane) - 1);
                const u32 eprev = (u32)__shfl((int)ew, mlow ? 63 - __clzll((long long)mlow) : lane, 64);
                bool in_tile = mlow != 0; u32 lb = pre + eprev;
                if (!in_tile) {
                    u32 pv = 0;
#pragma unroll
                    for (int v = 0; v < 3; v++) { if (v < wave && (s_l[v] & 1u)) { in_tile = true; lb = pv + s_b[v]; } pv += s_a[v]; }
                }
                u32 e = pm.eol; bool first = true;
                while (e) {                                       // line ends inside the piece (usually one)
                    const u32 k = (u32)__ffs((int)e) - 1; e &= e - 1;
                    const u32 b_here = ex + (u32)__popc(~pm.sp & ((1u << k) - 1));
                    if (first && !in_tile) {                      // the line began in front of the tile: 64-bit, once per tile
                        const u64 len = tbase + b_here - line_b0;
                        if (len > __atomic_load_n(O.longest, __ATOMIC_RELAXED)) atomicMax((unsigned long long *)O.longest, (unsigned long long)len);
                    } else { const u32 len = b_here - lb; best = len > best ? len : best; }
                    lb = b_here; first = false;
                }
            }
            best = wave_scan_inclusive<u32, OpMaxU32>(best);
            if (lane == 63) s_best[wave] = best;
            // drop the space-class bytes (highest first, so lower positions stay put) and park the bases in LDS
            u64 lo = pc.w0, hi = pc.w1; u32 d = pm.sp;
            while (d) {
                const u32 k = 31 - __clz((int)d); d &= ~(1u << k);
                const u64 slo = (lo >> 8) | (hi << 56), shi = hi >> 8;
                if (k < 8) { const u64 m = low_bytes(k); lo = (lo & m) | (slo & ~m); hi = shi; }
                else { const u64 m = low_bytes(k - 8); hi = (hi & m) | (shi & ~m); }
            }
            lds_store_n(stage + (u32)(tbase & 15) + ex, lo, hi, nseq);
            __syncthreads();
            flush_pack<true>(P, O.packed, O.casebits, tbase, tile_seq, stage);
            if (threadIdx.x == 0) {
                u64 b4 = s_best[0] > s_best[1] ? s_best[0] : s_best[1], b5 = s_best[2] > s_best[3] ? s_best[2] : s_best[3];
                b4 = b4 > b5 ? b4 : b5;
                if (b4 > __atomic_load_n(O.longest, __ATOMIC_RELAXED)) atomicMax((unsigned long long *)O.longest, (unsigned long long)b4);
            }
            return;
        }
    }
    __shared__ u8 cls[256];
    fill_classes(P, cls);                                         // (its barrier also separates the reads of s_a / s_l above from the writes below)
    PMask pm = piece_masks(pc);
    TileCtx ctx = thread_ctx(P, tile_eol, tile_sp, base, pm);
    bool active = base <= P.n;
    bool eof_here = active && (base + pc.cnt == P.n) && pc.cnt < ET_BYTES;
    // fast: inside sequence lines and nothing to replace (every non-space byte is an expected one)
    const bool fast = seq_piece(P, base, pc, pm, ctx) && piece_all_expected(P, pc, pm, cls, true);
    CountSink C; bool seg = false;
    if (fast) {
        C.nseq = 16 - __popc(pm.sp); C.saw_eol = pm.eol != 0;
        C.tail = pm.eol ? __popc(~pm.sp & 0xFFFFu & ~((2u << (31 - __clz((int)pm.eol))) - 1)) : C.nseq;
    } else if (active) { seg = segments_ok(P, base, pc, pm, ctx, cls); if (seg) classify_segments(P, base, pc, pm, ctx, C); else classify_range(P, base, pc, eof_here, ctx, C, cls); }
    u32 prea, preb, tota;
    u32 ia = wg_scan1<u32, OpAdd>(C.nseq | (C.nids << 16), &prea, &tota, s_a);
    u32 ib = wave_scan_inclusive<u32, OpAdd>(C.ncmt | (C.nrec << 16));
    preb = 0;                                                     // the second pair shares the barrier of the line-start scan below
    ia += prea;
    u32 iseq = ia & 0xFFFF, iids = ia >> 16, tile_seq = tota & 0xFFFF;
    WriteSink W(O); W.tbase = tbase; W.stage = stage + (PACK ? (u32)(W.tbase & 15) : 0u);   // PACK: groups of 16 bases aligned in LDS
    W.bseq = W.tbase + iseq - C.nseq; W.bids = O.t_ids[tile] + iids - C.nids;
    // base count at the start of the line this thread begins in: B at the most recent EOL before `base`.
    // Within the tile: B at an EOL is non-decreasing with position, so a running max over earlier threads works
    // (tile-relative and +1, so that 0 means "none").
    u32 rel = C.saw_eol ? (u32)(iseq - C.tail) + 1 : 0;
    u32 inc = wave_scan_inclusive<u32, OpMaxU32>(rel);
    if (lane == 63) { s_b[wave] = ib; s_l[wave] = inc; }
    u32 prev = wave_shift_up1<u32, OpMaxU32>(inc);
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 3; w++) if (w < wave) { preb += s_b[w]; prev = s_l[w] > prev ? s_l[w] : prev; }
    ib += preb;
    W.bcmt = O.t_cmt[tile] + (ib & 0xFFFF) - C.ncmt; W.rec = O.t_rec[tile] + (ib >> 16) - C.nrec;
    if (prev) W.line_b = W.tbase + prev - 1;
    else W.line_b = line_b0;                                      // the line began in an earlier tile
    W.best = 0;
    if (fast) {
        u32 e = pm.eol;
        while (e) {                                               // line ends inside the piece (usually one or none)
            u32 k = (u32)__ffs((int)e) - 1; e &= e - 1;
            u64 b_here = W.bseq + __popc(~pm.sp & ((1u << k) - 1));
            u64 len = b_here - W.line_b; if (len > W.best) W.best = len; W.line_b = b_here;
        }
        // drop the space-class bytes (highest first, so lower positions stay put), then store the bases with a few wide,
        // possibly unaligned LDS stores: 16 one-byte stores at a 16-byte lane stride are an 8-way bank conflict each
        u64 lo = pc.w0, hi = pc.w1; u32 d = pm.sp;
        while (d) {
            u32 k = 31 - __clz((int)d); d &= ~(1u << k);
            u64 slo = (lo >> 8) | (hi << 56), shi = hi >> 8;
            if (k < 8) { u64 m = low_bytes(k); lo = (lo & m) | (slo & ~m); hi = shi; }
            else { u64 m = low_bytes(k - 8); hi = (hi & m) | (shi & ~m); }
        }
        lds_store_n(W.stage + (W.bseq - W.tbase), lo, hi, C.nseq);
    } else if (active) { if (seg) classify_segments(P, base, pc, pm, ctx, W); else classify_range(P, base, pc, eof_here, ctx, W, cls); }
    __syncthreads();
    if (PACK) flush_pack<false>(P, O.packed, O.casebits, W.tbase, tile_seq, stage);
    else flush_tile(O.seq + W.tbase, stage, tile_seq);
    u64 best = wg_reduce1<u64, OpMaxU64>(W.best, s_best);
    // millions of workgroups, one address: look before touching it atomically
    if (threadIdx.x == 0 && best > __atomic_load_n(O.longest, __ATOMIC_RELAXED)) atomicMax((unsigned long long *)O.longest, (unsigned long long)best);
}


// ======================= FASTQ (process.c:477-544) =================================================================
// Non-empty lines cycle header / sequence / plus / quality; blank EOL runs between them are skipped, so the
// type of a byte is (ordinal of its line) mod 4 and the ordinal is a prefix count of line starts.
enum { EV_QUAL = 3 };
enum { FQ_E_AT = 0, FQ_E_PLUS = 1, FQ_E_QLEN = 2 };
struct FqOut {
    u8 *seq, *ids, *cmt, *qual;
    u8 *packed; u32 *casebits;    // as in EncOut
    u64 *rec_begin, *rec_end, *q_begin, *q_end;
    u64 *unexpected;              // [4][257]: id, comment, sequence, quality
    u64 *first_error;             // min over (record * 4 + kind)
    u64 *strict_first;            // --strict: min over strict_key of the unexpected bytes (nullptr otherwise)
    const u64 *t_seq, *t_ids, *t_cmt, *t_qual, *t_ls;
    const u32 *piece_cnt;         // per 16-byte piece: the four stream counts found by k_encq_count, 8 bits each
    const u32 *list;              // k_encq_scatter: the tiles that are not regular, in order (nullptr: every tile)
    const u32 *t_reg;             // k_encq_count_reg's / k_fq_pick's verdict on a tile (0: not regular)
    u32 reg_kind;                 // k_encq_scatter_reg: the verdict it takes (1; 2 beside k_fq_scatter_wave)
    u32 *redo_list, *n_redo;      // k_encq_scatter_reg: regular tiles whose letters want the general kernel after all
};

template <typename Sink>
__device__ __forceinline__ void classify_range_fastq(const EncP &P, u64 pos, const Piece &pc, bool with_eof, TileCtx ctx, Sink &S, const u8 *cls)
{
    i64 le = ctx.last_eol, ls = ctx.last_sp, ord = ctx.ord;
    const u32 cnt = pc.cnt;
    for (u32 k = 0; k < cnt + (with_eof ? 1u : 0u); k++) {
        u64 i = pos + k;
        bool eof = k >= cnt;
        u32 c = eof ? 0x0A : piece_byte(pc, k);
        u32 cl = cls[c];
        if (i >= P.p0) {
            bool prev_eol = i == P.p0 || le == (i64)i - 1;
            if (!(cl & CL_EOL)) {
                if (prev_eol) ord++;                                   // a new line starts here
                i64 line_start = le + 1; if ((u64)line_start < P.p0) line_start = (i64)P.p0;
                u64 rec = (u64)ord >> 2; u32 type = (u32)ord & 3; bool first = (i64)i == line_start;
                if (type == 0) {
                    if (first) { if (c != '@') S.error(rec, FQ_E_AT); S.header_start(rec); }
                    else if (ls < line_start) {
                        if (cl & CL_SPACE) S.emit(EV_IDS, 0);
                        else if (cl & CL_UNEXP_TEXT) { S.unexpected(0, c, i); S.emit(EV_SEQ, '?'); }
                        else S.emit(EV_IDS, c);
                    } else {
                        if (cl & CL_UNEXP_COMMENT) { S.unexpected(1, c, i); S.emit(EV_CMT, '?'); }
                        else S.emit(EV_CMT, c);
                    }
                } else if (type == 1) {
                    if (cl & CL_SPACE) {}
                    else if (cl & CL_EXPECTED) S.emit(EV_SEQ, c);
                    else { S.unexpected(2, c, i); S.emit(EV_SEQ, P.replacement); }
                } else if (type == 2) {
                    if (first && c != '+') S.error(rec, FQ_E_PLUS);
                } else {
                    if (first) { S.qual_begin(rec); S.emit(EV_QUAL, c); }   // process.c:522: appended unconditionally
                    else if (c >= 0x21 && c <= 0x7E) S.emit(EV_QUAL, c);
                    else if (cl & CL_SPACE) {}
                    else { S.unexpected(3, c, i); S.emit(EV_QUAL, '!'); }
                }
            } else if (!prev_eol) {                                    // this EOL closes a line
                i64 line_start = le + 1; if ((u64)line_start < P.p0) line_start = (i64)P.p0;
                u64 rec = (u64)ord >> 2; u32 type = (u32)ord & 3;
                if (type == 0) { if (ls < line_start) S.emit(EV_IDS, 0); S.emit(EV_CMT, 0); S.header_end(rec); }
                else if (type == 1) S.seq_end(rec);
                else if (type == 3) S.qual_end(rec);
            } else if (ord >= 0 && (ord & 3) == 0 && !eof) {
                // blank line right after a header: the reference reads an empty read and then needs '+'
                S.error((u64)ord >> 2, FQ_E_PLUS);
            }
        }
        if (eof) break;
        if (cl & CL_SPACE) { ls = (i64)i; if (cl & CL_EOL) le = (i64)i; }
    }
}

struct FqCount {
    u32 nseq = 0, nids = 0, ncmt = 0, nqual = 0;
    __device__ void emit(int s, u32) { if (s == EV_SEQ) nseq++; else if (s == EV_IDS) nids++; else if (s == EV_CMT) ncmt++; else nqual++; }
    __device__ void header_start(u64) {} __device__ void header_end(u64) {} __device__ void seq_end(u64) {}
    __device__ void qual_begin(u64) {} __device__ void qual_end(u64) {}
    __device__ void unexpected(int, u32, u64) {} __device__ void error(u64, int) {}
    __device__ void ids_range(const Piece &, u32 a, u32 b) { nids += b - a; }
    __device__ void cmt_range(const Piece &, u32 a, u32 b) { ncmt += b - a; }
    __device__ void seq_range(const Piece &, u32 a, u32 b, u32 sp) { nseq += (u32)__popc(range_mask(a, b) & ~sp); }
    __device__ void qual_range(const Piece &, u32 a, u32 b, u32 sp) { nqual += (u32)__popc(range_mask(a, b) & ~sp); }
    __device__ void qual_first(u32) { nqual++; }
    __device__ void term(int st) { if (st == EV_IDS) nids++; else ncmt++; }
};
struct FqWrite {
    const FqOut &O; u64 bseq, bids, bcmt, bqual; u8 *sstage, *qstage; u64 sbase, qbase;
    __device__ FqWrite(const FqOut &o) : O(o) {}
    __device__ void emit(int s, u32 ch) { if (s == EV_SEQ) sstage[bseq++ - sbase] = (u8)ch; else if (s == EV_IDS) O.ids[bids++] = (u8)ch; else if (s == EV_CMT) O.cmt[bcmt++] = (u8)ch; else qstage[bqual++ - qbase] = (u8)ch; }
    __device__ void header_start(u64) {}
    __device__ void header_end(u64 r) { O.rec_begin[r] = bseq; }
    __device__ void seq_end(u64 r) { O.rec_end[r] = bseq; }
    __device__ void qual_begin(u64 r) { O.q_begin[r] = bqual; }
    __device__ void qual_end(u64 r) { O.q_end[r] = bqual; }
    __device__ void unexpected(int kind, u32 ch, u64 pos) { atomicAdd((unsigned long long *)&O.unexpected[kind * 257 + ch], 1ull); if (O.strict_first) atomicMin((unsigned long long *)O.strict_first, strict_key(pos, kind, ch)); }
    __device__ void error(u64 r, int kind) { atomicMin((unsigned long long *)O.first_error, (unsigned long long)(r * 4 + kind)); }
    __device__ void ids_range(const Piece &pc, u32 a, u32 b) { u64 lo, hi; piece_from(pc, a, lo, hi); global_store_n(O.ids + bids, lo, hi, b - a); bids += b - a; }
    __device__ void cmt_range(const Piece &pc, u32 a, u32 b) { u64 lo, hi; piece_from(pc, a, lo, hi); global_store_n(O.cmt + bcmt, lo, hi, b - a); bcmt += b - a; }
    __device__ void seq_range(const Piece &pc, u32 a, u32 b, u32 sp) { u64 lo, hi; u32 n = piece_run_compact(pc, a, b, sp, lo, hi); lds_store_n(sstage + (bseq - sbase), lo, hi, n); bseq += n; }
    __device__ void qual_range(const Piece &pc, u32 a, u32 b, u32 sp) { u64 lo, hi; u32 n = piece_run_compact(pc, a, b, sp, lo, hi); lds_store_n(qstage + (bqual - qbase), lo, hi, n); bqual += n; }
    __device__ void qual_first(u32 ch) { emit(EV_QUAL, ch); }
    __device__ void term(int st) { emit(st, 0); }
};

// ---- FASTQ pieces segment-wise (see classify_segments): the runs between EOL bytes are whole lines or parts of lines whose type is
// the line ordinal mod 4.  Same conditions: full piece behind p0, no byte that its role would replace (dry run into FqRoleSink).
template <typename Sink>
__device__ __forceinline__ void classify_segments_fastq(const EncP &P, u64 base, const Piece &pc, const PMask &pm, const TileCtx &ctx, Sink &S)
{
    i64 le = ctx.last_eol, ls = ctx.last_sp, ord = ctx.ord;
    u32 k = 0;
    while (k < ET_BYTES) {
        const u64 i = base + k;
        const bool prev_eol = le == (i64)i - 1;
        i64 line_start = le + 1; if ((u64)line_start < P.p0) line_start = (i64)P.p0;
        if ((pm.eol >> k) & 1) {                                          // one EOL byte
            if (!prev_eol) {                                              // it closes a line
                const u64 rec = (u64)ord >> 2; const u32 type = (u32)ord & 3;
                if (type == 0) { if (ls < line_start) S.term(EV_IDS); S.term(EV_CMT); S.header_end(rec); }
                else if (type == 1) S.seq_end(rec);
                else if (type == 3) S.qual_end(rec);
            } else if (ord >= 0 && (ord & 3) == 0) S.error((u64)ord >> 2, FQ_E_PLUS);   // blank line right after a header
            le = ls = (i64)i; k++;
            continue;
        }
        const u32 em = pm.eol >> k;
        const u32 e = em ? k + (u32)__ffs((int)em) - 1 : ET_BYTES;       // run of non-EOL bytes [k, e)
        if (prev_eol) { ord++; line_start = (i64)i; }                     // a new line starts here
        const u64 rec = (u64)ord >> 2; const u32 type = (u32)ord & 3; const bool first = (i64)i == line_start;
        const u32 spm = pm.sp & range_mask(k, e);
        u32 a = k;
        if (type == 0) {
            if (first) { if (piece_byte(pc, k) != '@') S.error(rec, FQ_E_AT); S.header_start(rec); a++; }
            if (a < e) {
                const u32 sp2 = pm.sp & range_mask(a, e);
                if (ls < line_start) {
                    const u32 f = sp2 ? (u32)__ffs((int)sp2) - 1 : e;
                    if (f > a) S.ids_range(pc, a, f);
                    if (f < e) { S.term(EV_IDS); if (f + 1 < e) S.cmt_range(pc, f + 1, e); }
                } else S.cmt_range(pc, a, e);
            }
        } else if (type == 1) S.seq_range(pc, k, e, pm.sp);
        else if (type == 2) { if (first && piece_byte(pc, k) != '+') S.error(rec, FQ_E_PLUS); }
        else {
            if (first) { S.qual_begin(rec); S.qual_first(piece_byte(pc, k)); a++; }     // process.c:522: the first byte goes in whatever it is
            if (a < e) S.qual_range(pc, a, e, pm.sp);
        }
        if (spm) ls = (i64)(base + (31 - __clz((int)spm)));
        k = e;
    }
}
struct FqRoleSink {
    u32 idm = 0, cmm = 0, sqm = 0, qlm = 0;
    __device__ void ids_range(const Piece &, u32 a, u32 b) { idm |= range_mask(a, b); }
    __device__ void cmt_range(const Piece &, u32 a, u32 b) { cmm |= range_mask(a, b); }
    __device__ void seq_range(const Piece &, u32 a, u32 b, u32 sp) { sqm |= range_mask(a, b) & ~sp; }
    __device__ void qual_range(const Piece &, u32 a, u32 b, u32 sp) { qlm |= range_mask(a, b) & ~sp; }
    __device__ void qual_first(u32) {}
    __device__ void term(int) {} __device__ void header_start(u64) {} __device__ void header_end(u64) {} __device__ void seq_end(u64) {}
    __device__ void qual_begin(u64) {} __device__ void qual_end(u64) {} __device__ void error(u64, int) {}
};
// counts and roles in ONE walk (k_encq_count): the counts are those of the segment-wise classification, valid when the roles pass
struct FqCountRole {
    FqCount C; FqRoleSink R;
    __device__ void ids_range(const Piece &pc, u32 a, u32 b) { C.ids_range(pc, a, b); R.ids_range(pc, a, b); }
    __device__ void cmt_range(const Piece &pc, u32 a, u32 b) { C.cmt_range(pc, a, b); R.cmt_range(pc, a, b); }
    __device__ void seq_range(const Piece &pc, u32 a, u32 b, u32 sp) { C.seq_range(pc, a, b, sp); R.seq_range(pc, a, b, sp); }
    __device__ void qual_range(const Piece &pc, u32 a, u32 b, u32 sp) { C.qual_range(pc, a, b, sp); R.qual_range(pc, a, b, sp); }
    __device__ void qual_first(u32 ch) { C.qual_first(ch); }
    __device__ void term(int st) { C.term(st); }
    __device__ void header_start(u64) {} __device__ void header_end(u64) {} __device__ void seq_end(u64) {}
    __device__ void qual_begin(u64) {} __device__ void qual_end(u64) {} __device__ void error(u64, int) {}
};
// no byte that its role would replace
__device__ __forceinline__ bool roles_ok_fastq(const EncP &P, const Piece &pc, const FqRoleSink &R, const u8 *cls)
{
    u32 w[4] = { (u32)pc.w0, (u32)(pc.w0 >> 32), (u32)pc.w1, (u32)(pc.w1 >> 32) };
    if ((R.idm | R.cmm) && ((R.idm | R.cmm) & piece_ctl_mask(w))) return false;
    if (R.qlm && (R.qlm & piece_not_quality_mask(w))) return false;
    u32 cand = R.sqm ? (piece_not_quick(w, P.qlo, P.qhi) & R.sqm) : 0u;
    while (cand) { u32 k = (u32)__ffs((int)cand) - 1; cand &= cand - 1; if (!(cls[piece_byte(pc, k)] & CL_EXPECTED)) return false; }
    return true;
}

// ordinal of the line in progress at `base`: (#line starts before base) - 1
__device__ __forceinline__ i64 thread_ord(const EncP &P, const u64 *t_ls, u64 base, u64 *lds, const Piece &pc, const PMask &pm)
{
    u32 nls = pc.cnt ? count_line_starts_m(P, base, pc, pm) : 0;
    u64 t; u64 incl = wg_scan_inclusive<u64, OpAdd>((u64)nls, &t, lds);
    return (i64)(t_ls[base / ET_TILE] + incl - nls) - 1;          // (the workgroup's tile: not always blockIdx.x, see k_encq_count's list)
}

struct SlowCtx { i64 le, ls, ord; };
// slow lanes -> dense list (ballot compaction per wave, wave offsets through LDS); also parks each slow lane's context in LDS
__device__ __forceinline__ void slow_gather(bool slow, const TileCtx &ctx, SlowCtx *s_ctx, u16 *s_list, u32 *s_n)
{
    __shared__ u32 wn[4];
    u64 bal = __ballot(slow);
    u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 idx = (u32)__popcll(bal & ((1ull << lane) - 1));
    if (lane == 0) wn[wave] = (u32)__popcll(bal);
    __syncthreads();
    for (u32 w = 0; w < wave; w++) idx += wn[w];
    if (slow) { s_list[idx] = (u16)threadIdx.x; s_ctx[threadIdx.x].le = ctx.last_eol; s_ctx[threadIdx.x].ls = ctx.last_sp; s_ctx[threadIdx.x].ord = ctx.ord; }
    if (threadIdx.x == 0) *s_n = wn[0] + wn[1] + wn[2] + wn[3];
    __syncthreads();
}

// A full piece in the middle of a read's sequence line (returns 1) or quality line (3) with nothing to drop or replace; else 0.
__device__ __forceinline__ int fq_piece(const EncP &P, u64 base, const Piece &pc, const PMask &pm, const TileCtx &ctx, const u8 *cls)
{
    if (pc.cnt != ET_BYTES || pm.eol || ctx.ord < 0) return 0;
    i64 line_start = ctx.last_eol + 1; if ((u64)line_start < P.p0) line_start = (i64)P.p0;
    if (line_start >= (i64)base) return 0;                     // the line's first byte has its own rules (process.c:522)
    u32 type = (u32)ctx.ord & 3;
    if (type == 1 && pm.sp == 0 && piece_all_expected(P, pc, pm, cls, false)) return 1;
    if (type == 3 && piece_all_qual(pc)) return 3;
    return 0;
}

__global__ __launch_bounds__(256) void k_encq_count(EncP P, const i64 *tile_eol, const i64 *tile_sp, const u64 *t_ls,
                                                     u64 *t_seq, u64 *t_ids, u64 *t_cmt, u64 *t_qual, u32 *piece_cnt, const u32 *list, int piece_only)
{
    const u64 tile = list ? list[blockIdx.x] : blockIdx.x;      // list: the tiles k_encq_count_reg left (those that are not regular)
                                                                // piece_only: a tile k_encq_scatter_reg handed back -- its counts stand (and are scanned), its pieces' are wanted
    __shared__ u64 lds[4];
    __shared__ u8 cls[256];
    fill_classes(P, cls);
    u64 base = tile * ET_TILE + (u64)threadIdx.x * ET_BYTES;
    Piece pc = load_piece(P, base);
    PMask pm = piece_masks(pc);
    TileCtx ctx = thread_ctx(P, tile_eol, tile_sp, base, pm);
    ctx.ord = thread_ord(P, t_ls, base, lds, pc, pm);
    FqCount S;
    int fast = fq_piece(P, base, pc, pm, ctx, cls);
    if (fast == 1) S.nseq = 16; else if (fast == 3) S.nqual = 16;
    // The other pieces (headers, line ends: about a quarter of them in 150-bp reads) need the per-byte state machine.  They
    // are gathered and handled by the first lanes of the workgroup, so that one wavefront walks the slow path instead of all four.
    __shared__ SlowCtx s_ctx[256]; __shared__ u16 s_list[256]; __shared__ u32 s_nslow; __shared__ u64 s_cnt[256];
    const bool slow = !fast && base <= P.n;
    slow_gather(slow, ctx, s_ctx, s_list, &s_nslow);
    if (threadIdx.x < s_nslow) {
        u32 who = s_list[threadIdx.x];
        u64 b2 = tile * ET_TILE + (u64)who * ET_BYTES;
        Piece p2 = load_piece(P, b2);
        TileCtx c2; c2.last_eol = s_ctx[who].le; c2.last_sp = s_ctx[who].ls; c2.hdr = false; c2.ord = s_ctx[who].ord;
        FqCount S2; bool segok = false;
        if (p2.cnt == ET_BYTES && b2 > P.p0 && c2.ord >= 0) {            // a full piece behind p0 in a numbered line
            FqCountRole CR; classify_segments_fastq(P, b2, p2, piece_masks(p2), c2, CR);
            if (roles_ok_fastq(P, p2, CR.R, cls)) { S2 = CR.C; segok = true; }
        }
        if (!segok) classify_range_fastq(P, b2, p2, (b2 + p2.cnt == P.n) && p2.cnt < ET_BYTES, c2, S2, cls);
        // bit 63: the piece passed the segment-wise test -- the scatter pass does not walk it a second time to find out
        s_cnt[who] = (u64)S2.nseq | ((u64)S2.nids << 16) | ((u64)S2.ncmt << 32) | ((u64)S2.nqual << 48) | ((u64)(segok ? 1 : 0) << 63);
    }
    __syncthreads();
    u32 segok_bit = 0;
    if (slow) { u64 v = s_cnt[threadIdx.x]; S.nseq = v & 0xFFFF; S.nids = (v >> 16) & 0xFFFF; S.ncmt = (v >> 32) & 0xFFFF; S.nqual = (u32)(v >> 48) & 0x7FFFu; segok_bit = (u32)(v >> 63); }
    // the four counts of every piece (each at most 17) are kept for the scatter pass, which then walks the slow pieces once, not twice;
    // bit 31: the verdict of the segment-wise test
    piece_cnt[tile * 256 + threadIdx.x] = S.nseq | (S.nids << 8) | (S.ncmt << 16) | (S.nqual << 24) | (segok_bit << 31);
    u64 tot;
    wg_scan_inclusive<u64, OpAdd>((u64)S.nseq | ((u64)S.nids << 16) | ((u64)S.ncmt << 32) | ((u64)S.nqual << 48), &tot, lds);
    if (threadIdx.x == 0 && !piece_only) { t_seq[tile] = tot & 0xFFFF; t_ids[tile] = (tot >> 16) & 0xFFFF; t_cmt[tile] = (tot >> 32) & 0xFFFF; t_qual[tile] = tot >> 48; }
}

template <bool PACK>
__global__ __launch_bounds__(256) void k_encq_scatter(EncP P, const i64 *tile_eol, const i64 *tile_sp, FqOut O)
{
    const u64 tile = O.list ? O.list[blockIdx.x] : blockIdx.x;   // list: the tiles that are not regular (k_encq_scatter_reg does the others)
    __shared__ u64 lds[4];
    __shared__ u8 cls[256];
    fill_classes(P, cls);
    u64 base = tile * ET_TILE + (u64)threadIdx.x * ET_BYTES;
    Piece pc = load_piece(P, base);
    PMask pm = piece_masks(pc);
    TileCtx ctx = thread_ctx(P, tile_eol, tile_sp, base, pm);
    ctx.ord = thread_ord(P, O.t_ls, base, lds, pc, pm);
    bool active = base <= P.n;
    FqCount C;
    const int fast = fq_piece(P, base, pc, pm, ctx, cls);
    { u32 v = O.piece_cnt[tile * 256 + threadIdx.x]; C.nseq = v & 0xFF; C.nids = (v >> 8) & 0xFF; C.ncmt = (v >> 16) & 0xFF; C.nqual = (v >> 24) & 0x7F; }
    __shared__ SlowCtx s_ctx[256]; __shared__ u16 s_list[256]; __shared__ u32 s_nslow;
    __shared__ u64 s_w[256][4];                                   // stream positions of the slow pieces for the write pass
    const bool slow = !fast && active;
    slow_gather(slow, ctx, s_ctx, s_list, &s_nslow);
    u64 totp;
    u64 ip = wg_scan_inclusive<u64, OpAdd>((u64)C.nseq | ((u64)C.nids << 16) | ((u64)C.ncmt << 32) | ((u64)C.nqual << 48), &totp, lds);
    u64 iseq = ip & 0xFFFF, iids = (ip >> 16) & 0xFFFF, icmt = (ip >> 32) & 0xFFFF, iq = ip >> 48, tots = totp & 0xFFFF, totq = totp >> 48;
    __shared__ __attribute__((aligned(16))) u8 sstage0[ET_TILE + 48], qstage[ET_TILE + 16];
    FqWrite W(O); W.qstage = qstage; W.sbase = O.t_seq[tile]; W.qbase = O.t_qual[tile];
    u8 *const sstage = sstage0 + (PACK ? (u32)(W.sbase & 15) : 0u); W.sstage = sstage;
    W.bseq = O.t_seq[tile] + iseq - C.nseq; W.bids = O.t_ids[tile] + iids - C.nids;
    W.bcmt = O.t_cmt[tile] + icmt - C.ncmt; W.bqual = O.t_qual[tile] + iq - C.nqual;
    if (fast == 1) lds_store_n(sstage + (W.bseq - W.sbase), pc.w0, pc.w1, 16);
    else if (fast == 3) lds_store_n(qstage + (W.bqual - W.qbase), pc.w0, pc.w1, 16);
    else if (slow) { s_w[threadIdx.x][0] = W.bseq; s_w[threadIdx.x][1] = W.bids; s_w[threadIdx.x][2] = W.bcmt; s_w[threadIdx.x][3] = W.bqual; }
    __syncthreads();
    if (threadIdx.x < s_nslow) {                                  // write pass of the slow pieces, again by the first lanes
        u32 who = s_list[threadIdx.x];
        u64 b2 = tile * ET_TILE + (u64)who * ET_BYTES;
        Piece p2 = load_piece(P, b2);
        TileCtx c2; c2.last_eol = s_ctx[who].le; c2.last_sp = s_ctx[who].ls; c2.hdr = false; c2.ord = s_ctx[who].ord;
        FqWrite W2(O); W2.sstage = sstage; W2.qstage = qstage; W2.sbase = W.sbase; W2.qbase = W.qbase;
        W2.bseq = s_w[who][0]; W2.bids = s_w[who][1]; W2.bcmt = s_w[who][2]; W2.bqual = s_w[who][3];
        if (O.piece_cnt[tile * 256 + who] >> 31) classify_segments_fastq(P, b2, p2, piece_masks(p2), c2, W2);   // k_encq_count's verdict
        else classify_range_fastq(P, b2, p2, (b2 + p2.cnt == P.n) && p2.cnt < ET_BYTES, c2, W2, cls);
    }
    __syncthreads();
    if (PACK) flush_pack<false>(P, O.packed, O.casebits, W.sbase, (u32)tots, sstage0);
    else flush_tile(O.seq + W.sbase, sstage, (u32)tots);
    flush_tile(O.qual + W.qbase, qstage, (u32)totq);
}


// ---- REGULAR tiles of a FASTQ text, by lines ------------------------------------------------------------------------------------
// The reference has a fast parser for FASTQ that is what sequencers write (process_well_formed_fastq, process.c:430-474) beside the
// tolerant one (:477-544); the general kernels above are the tolerant one -- a context per 16-byte piece (two workgroup scans), and a
// third of the pieces of 150-base reads (those with a line end or header bytes) walked by a lane on its own.  A tile is REGULAR when
// the tolerant parser has nothing to tolerate in its STRUCTURE: it lies wholly behind p0 and inside the text; its only EOL-class byte
// is '\n' and no line of it is empty; it has at most 255 line ends; no byte of it is 0x7F or above; header lines begin with '@' and
// hold no byte below 0x21 but the first blank or tab (which ends the ID) and blanks behind it; sequence and quality lines hold no
// byte below 0x21 at all; '+' lines begin with '+'.
// Such a tile is a string of at most 256 LINE SEGMENTS cut by its '\n' bytes whose types are the line ordinal mod 4, counted on from
// the line in progress at its first byte (t_ls, tile_eol: the same tables the general kernels take their contexts from).  One
// wavefront takes a segment per lane: lengths, the ID / comment split of header segments (from a bit map of the bytes below 0x21),
// one scan -- and every byte's place in its stream follows from its segment's.  Counts, streams and record tables are those of the
// general kernels bit for bit (a byte is counted in the tile it lies in); a tile that is not regular is left to them (t_reg = 0, the
// list).  Whether the LETTERS of the sequence lines are accepted ones is not part of the structure (a replaced letter counts like an
// accepted one): the scatter pass looks at them anyway and hands a tile with a letter that is not A C G T/U N back (the redo list).
#define FQR_MAXSEG 256
#define FQR_SKIP ((i32)0x40000000)
struct FqrLds {
    __attribute__((aligned(16))) u8 txt[ET_TILE + 16];
    u64 segoff[FQR_MAXSEG];         // exclusive prefix over the segments of nseq | nids << 16 | ncmt << 32 | nqual << 48
    u64 total;                      // ... and the tile's totals
    u32 segse[FQR_MAXSEG];          // first byte | end << 16 of the segment (positions in the tile)
    u32 seghdr[FQR_MAXSEG];         // header segments: first ID byte | position of the blank that ends the ID (0xFFFF: the ID began and ended in front of the tile; == end: none so far) << 16
    i32 segdst[FQR_MAXSEG];         // scatter: where in the staging buffer the segment's first byte goes, minus its position in the tile (sequence and quality segments; FQR_SKIP: the others)
    u32 so[4];                      // scatter: where the four streams' regions begin in the staging buffer
    u16 nlpos[FQR_MAXSEG];          // the tile's '\n' bytes
    u16 oth[256];                   // per piece: its bytes below 0x21 that are not '\n'
    u8  segtype[FQR_MAXSEG];        // line ordinal & 3 | 4: the segment begins its line | 8: its line ends in the tile
    u32 wcnt[4];
    u32 bad;
};
// '\n' bytes and the other bytes below 0x21 of a piece, one bit per byte; *high: some byte is 0x7F or above
__device__ __forceinline__ void fqr_masks(const u32 w[4], u32 &nlm, u32 &oth, bool &high)
{
    const u32 H = 0x80808080u, L = 0x7F7F7F7Fu; u32 f[4], g[4], hi = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u32 x = w[i], l = x & L, y = x ^ 0x0A0A0A0Au;
        f[i] = ~(((y & L) + L) | y) & H;                          // == 0x0A
        g[i] = ~(x | (l + 0x5F5F5F5Fu)) & H & ~f[i];              // < 0x21 and not 0x0A
        hi |= x | (l + 0x01010101u);                              // >= 0x80, or 0x7F
    }
    nlm = swar_movemask16(f[0], f[1], f[2], f[3]); oth = swar_movemask16(g[0], g[1], g[2], g[3]); high = (hi & H) != 0;
}
// Segments of the tile into LDS; returns false when the tile has too many of them (uniform).  CHECK: also sets S.bad for what makes
// the tile irregular in its segments' framing.  k0 = the segment of the lane's first byte.  Ends on a barrier.
template <bool CHECK>
__device__ __forceinline__ bool fqr_segments(const EncP &P, u64 tile, const i64 *tile_eol, const i64 *tile_sp, const u64 *t_ls, FqrLds &S,
                                             const u32 w[4], u32 nlm, u32 oth, u32 &k0, u32 &nseg, const u64 *bases = nullptr /* scatter: the tile's offsets in the four streams */)
{
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    { uint4 v; v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3]; *(uint4 *)(S.txt + 16 * tid) = v; }
    S.oth[tid] = (u16)oth;
    if (tid == 0) S.bad = 0;
    const u32 cnt = (u32)__popc(nlm);
    const u32 incl = wave_scan_inclusive<u32, OpAdd>(cnt);
    if (lane == 63) S.wcnt[wave] = incl;
    __syncthreads();
    u32 pre = 0, tot = 0;
#pragma unroll
    for (u32 q = 0; q < 4; q++) { const u32 x = S.wcnt[q]; if (q < wave) pre += x; tot += x; }
    k0 = pre + incl - cnt; nseg = tot + 1;
    if (tot >= FQR_MAXSEG) return false;                                             // (uniform)
    { u32 m = nlm, idx = k0; while (m) { S.nlpos[idx++] = (u16)(16 * tid + (u32)__ffs((int)m) - 1); m &= m - 1; } }
    __syncthreads();
    if (wave == 0) {
        const u64 tb = tile * ET_TILE;
        const i64 le = tile_eol[tile - 1], lsp = tile_sp[tile - 1];
        const bool prev_eol = le == (i64)tb - 1;
        i64 ls0 = le + 1; if ((u64)ls0 < P.p0) ls0 = (i64)P.p0;
        const u64 ord0 = t_ls[tile] - (prev_eol ? 0u : 1u);                          // ordinal of segment 0's line
        u64 carry = 0; bool bad = false;
        for (u32 kb = 0; kb <= tot; kb += 64) {                                      // (uniform: 64 segments a round)
            const u32 k = kb + lane;
            u64 c = 0;
            if (k <= tot) {
                const u32 s_ = k ? (u32)S.nlpos[k - 1] + 1 : 0u, e_ = k < tot ? (u32)S.nlpos[k] : (u32)ET_TILE;
                const u32 len = e_ - s_;
                const bool starts = k > 0 || prev_eol, closed = k < tot;
                if (CHECK && len == 0 && starts && closed) bad = true;               // an empty line
                const u32 ty = (u32)((ord0 + k) & 3);
                u32 hdr = 0;
                if (ty == 1) c = (u64)len;
                else if (ty == 3) c = (u64)len << 48;
                else if (ty == 2) { if (CHECK && starts && len && S.txt[s_] != '+') bad = true; }
                else {
                    u32 a = s_;
                    if (starts && len) { if (CHECK && S.txt[s_] != '@') bad = true; a++; }
                    const bool seen = !starts && lsp >= ls0;                         // the blank that ends the ID came in front of the tile
                    u32 f = e_;                                                      // ... or is here (e_: not in this segment)
                    if (a < e_) for (u32 q = a >> 4; q <= (e_ - 1) >> 4; q++) {
                        u32 m = S.oth[q];
                        if (!m) continue;
                        const u32 lo = 16 * q;
                        if (a > lo) m &= ~((1u << (a - lo)) - 1);
                        if (e_ < lo + 16) m &= (1u << (e_ - lo)) - 1;
                        if (m && !seen && f == e_) {
                            f = lo + (u32)__ffs((int)m) - 1; m &= m - 1;
                            if (CHECK) { const u32 ch = S.txt[f]; if (ch != 0x20 && ch != 0x09) bad = true; }
                        }
                        if (!CHECK && (seen || f != e_)) break;
                        while (CHECK && m) { if (S.txt[lo + (u32)__ffs((int)m) - 1] != 0x20) bad = true; m &= m - 1; }   // a comment's bytes below 0x21 are blanks
                    }
                    u32 nids, ncmt;
                    if (seen) { nids = 0; ncmt = e_ - a; f = 0xFFFFu; }
                    else { nids = f - a; ncmt = 0; if (f < e_) { nids++; ncmt = e_ - (f + 1); } }
                    if (closed) { if (!seen && f == e_) nids++; ncmt++; }
                    c = ((u64)nids << 16) | ((u64)ncmt << 32);
                    hdr = a | (f << 16);
                }
                S.segse[k] = s_ | (e_ << 16); S.seghdr[k] = hdr;
                S.segtype[k] = (u8)(ty | (starts ? 4u : 0u) | (closed ? 8u : 0u));
            }
            const u64 inc = wave_scan_inclusive<u64, OpAdd>(c) + carry;
            if (k <= tot) S.segoff[k] = inc - c;
            if (k == tot) S.total = inc;
            carry = shfl_idx_t<u64>(inc, 63);
        }
        if (CHECK && bad) S.bad = 1;
        if (bases) {
            // The staging buffer holds the tile's share of the four streams one behind the other, each region placed so that its address
            // is congruent to the stream's own mod 16: whole 16-byte words of LDS then go out as aligned 16-byte stores.
            const u32 tots = (u32)(carry & 0xFFFF), toti = (u32)((carry >> 16) & 0xFFFF), totc = (u32)((carry >> 32) & 0xFFFF), totq = (u32)(carry >> 48);
            const u32 sO = (u32)(bases[0] & 15);
            const u32 qO = ((sO + tots + 15) & ~15u) + (u32)(bases[3] & 15);
            const u32 iO = ((qO + totq + 15) & ~15u) + (u32)(bases[1] & 15);
            const u32 cO = ((iO + toti + 15) & ~15u) + (u32)(bases[2] & 15);
            (void)totc;
            if (lane == 0) { S.so[0] = sO; S.so[1] = iO; S.so[2] = cO; S.so[3] = qO; }
            for (u32 kb = 0; kb <= tot; kb += 64) {
                const u32 k = kb + lane;
                if (k <= tot) {
                    const u32 ty = S.segtype[k] & 3u, s_ = S.segse[k] & 0xFFFF; const u64 off = S.segoff[k];
                    S.segdst[k] = ty == 1 ? (i32)(sO + (u32)(off & 0xFFFF)) - (i32)s_ : ty == 3 ? (i32)(qO + (u32)(off >> 48)) - (i32)s_ : FQR_SKIP;
                }
            }
        }
    }
    __syncthreads();
    return true;
}
// n bytes from LDS to global memory where stage and dst are congruent mod 16, by the lanes t = 0 .. nt-1 of a group
__device__ __forceinline__ void flush_congruent(u8 *dst, const u8 *stage, u32 n, u32 t, u32 nt)
{
    u32 head = (u32)((16 - ((uintptr_t)dst & 15)) & 15); if (head > n) head = n;
    if (t < head) dst[t] = stage[t];
    const u32 chunks = (n - head) >> 4;
    for (u32 q = t; q < chunks; q += nt) *(uint4 *)(dst + head + 16 * q) = *(const uint4 *)(stage + head + 16 * q);
    const u32 done = head + 16 * chunks;
    if (t < n - done) dst[done + t] = stage[done + t];
}

__global__ __launch_bounds__(256) void k_encq_count_reg(EncP P, const i64 *tile_eol, const i64 *tile_sp, const u64 *t_ls,
                                                         u64 *t_seq, u64 *t_ids, u64 *t_cmt, u64 *t_qual, u32 *t_reg, u64 *t_need, u64 tiles)
{
    const u64 tile = blockIdx.x, tb = tile * ET_TILE;
    __shared__ FqrLds S;
    bool regular = tile > 0 && tb > P.p0 && tb + ET_TILE <= P.n;                      // (uniform)
    if (regular) {
        const u8 *src = P.text + tb + 16 * threadIdx.x;
        const u64 w0 = ld64(src), w1 = ld64(src + 8);
        const u32 w[4] = { (u32)w0, (u32)(w0 >> 32), (u32)w1, (u32)(w1 >> 32) };
        u32 nlm, oth, k0, nseg; bool high;
        fqr_masks(w, nlm, oth, high);
        regular = fqr_segments<true>(P, tile, tile_eol, tile_sp, t_ls, S, w, nlm, oth, k0, nseg);
        if (regular) {
            // a piece's bytes below 0x21 against the types of its segments: none in sequence and quality lines; no EOL-class byte in a
            // '+' line (whatever else it holds is skipped); header segments were looked at by their lanes
            bool bad = high;
            if (oth) {
                u32 a = 0, m = nlm, k = k0;
                for (;;) {
                    const u32 b = m ? (u32)__ffs((int)m) - 1 : 16u;
                    u32 o = b > a ? oth & range_mask(a, b) : 0u;
                    if (o) {
                        const u32 ty = S.segtype[k] & 3u;
                        if (ty & 1) bad = true;
                        else if (ty == 2) while (o) { const u32 q = (u32)__ffs((int)o) - 1; o &= o - 1; const u32 ch = (w[q >> 2] >> (8 * (q & 3))) & 0xFF; if (ch >= 0x0B && ch <= 0x0D) bad = true; }
                    }
                    if (!m) break;
                    m &= m - 1; a = b + 1; k++;
                }
            }
            if (bad) S.bad = 1;
            __syncthreads();
            regular = S.bad == 0;
        }
    }
    if (threadIdx.x == 0) {
        t_reg[tile] = regular ? 1u : 0u; t_need[tile] = regular ? 0u : 1u;
        if (regular) { const u64 t = S.total; t_seq[tile] = t & 0xFFFF; t_ids[tile] = (t >> 16) & 0xFFFF; t_cmt[tile] = (t >> 32) & 0xFFFF; t_qual[tile] = t >> 48; }
    }
}

// ---- the first pass of a FASTQ text read ONCE: k_enc_last's tables and k_encq_count_reg's counts from one look (k_fq_first, k_fq_pick) ----
// What a tile's lines ARE -- header, bases, '+', qualities -- follows from the ordinal of its first line, and that is the scan of the line
// starts of every tile in front of it: k_encq_count_reg reads the text a second time only to apply that one number.  But the ordinal only
// says which of FOUR ways the tile's segments take their roles (segment k is line ordinal + k), so the first look counts under all four:
// per class k & 3 the bytes of its segments, what they would give the ID and the comment stream if they were headers, and whether they
// could be headers, '+' lines, letters at all (the checks of fqr_segments<true> and k_encq_count_reg, role by role).  The segment in
// progress at the tile's first byte keeps its header reading apart, in both forms (the blank that ends its ID seen in front of the tile or
// not: the previous tile's table says which).  k_fq_pick, a lane per tile behind the scans, takes the class each role falls to: counts,
// verdict and tables are those of the two kernels this replaces, value for value (NAF_GPU_FQ_FIRST=0: those two kernels; =check: both,
// compared).  One wavefront per tile, 64 bytes a lane, no barrier that waits for another wavefront.
struct FqCand { u16 L[4], I[4], C[4]; u16 I0, C0, C0s, nseg; u32 flags; };
enum { FQC_OK = 1u, FQC_STARTS = 2u, FQC_BADS = 1u << 4, FQC_BADP = 1u << 8, FQC_BADH = 1u << 12, FQC_BADH0 = 1u << 16, FQC_BADH0S = 1u << 17 };
__device__ __forceinline__ u64 bits_from_to(u32 a, u32 b) { return (b >= 64 ? ~0ull : (1ull << b) - 1) & ~((1ull << a) - 1); }   // bits [a, b), a < 64, b <= 64
__device__ __forceinline__ u32 swar_eq_mask16(const u32 w[4], u32 c4)                  // one bit per byte: == the byte c4 repeats
{
    const u32 H = 0x80808080u, L = 0x7F7F7F7Fu; u32 f[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { const u32 y = w[i] ^ c4; f[i] = ~(((y & L) + L) | y) & H; }
    return swar_movemask16(f[0], f[1], f[2], f[3]);
}
__global__ __launch_bounds__(64) void k_fq_first(EncP P, i64 *tile_eol, i64 *tile_sp, u64 *tile_ls, FqCand *cand)
{
    __shared__ __attribute__((aligned(16))) u8 txt[ET_TILE + 16];
    __shared__ u64 s_oth[64], s_othx[64];
    __shared__ u16 s_nl[FQR_MAXSEG];
    const u64 tile = xcd_block(), tb = tile * ET_TILE;
    const u32 lane = threadIdx.x;
    if (!(tile > 0 && tb > P.p0 && tb + ET_TILE <= P.n)) {
        // the text's first tiles and its last one: the tables as k_enc_last makes them, a piece at a time; nothing to pick from
        u32 pos = 0, nls = 0;
#pragma unroll
        for (u32 k = 0; k < 4; k++) {
            const u32 idx = k * 64 + lane, t16 = idx * ET_BYTES + 1;
            const u64 base = tb + (u64)idx * ET_BYTES;
            const Piece pc = load_piece(P, base);
            const PMask pm = piece_masks(pc);
            pos = OpPkMaxU16::f<u32>(pos, (pm.eol ? t16 + (31 - __clz((int)pm.eol)) : 0u) | ((pm.sp ? t16 + (31 - __clz((int)pm.sp)) : 0u) << 16));
            nls += pc.cnt ? count_line_starts_m(P, base, pc, pm) : 0u;
        }
        pos = wave_scan_inclusive<u32, OpPkMaxU16>(pos); nls = wave_scan_inclusive<u32, OpAdd>(nls);
        if (lane == 63) {
            tile_eol[tile] = (pos & 0xFFFF) ? (i64)(tb + (pos & 0xFFFF) - 1) : -1;
            tile_sp[tile] = (pos >> 16) ? (i64)(tb + (pos >> 16) - 1) : -1;
            tile_ls[tile] = nls;
            cand[tile].flags = 0;
        }
        return;
    }
    const u8 *src = P.text + tb + 64 * lane;
    u64 wv[8];
#pragma unroll
    for (u32 k = 0; k < 4; k++) { uint4 q; __builtin_memcpy(&q, src + 16 * k, 16); wv[2 * k] = (u64)q.x | ((u64)q.y << 32); wv[2 * k + 1] = (u64)q.z | ((u64)q.w << 32); }
    u64 nl = 0, oth = 0, othx = 0, eol = 0, sp = 0; bool high = false;
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
        const u32 w[4] = { (u32)wv[2 * k], (u32)(wv[2 * k] >> 32), (u32)wv[2 * k + 1], (u32)(wv[2 * k + 1] >> 32) };
        u32 nlm, o; bool hi;
        fqr_masks(w, nlm, o, hi);
        const PieceFlags f = piece_flags(w);
        const u32 m20 = swar_eq_mask16(w, 0x20202020u);
        nl |= (u64)nlm << (16 * k); oth |= (u64)o << (16 * k); othx |= (u64)(o & ~m20) << (16 * k);
        eol |= (u64)f.eol << (16 * k); sp |= (u64)f.sp << (16 * k); high |= hi;
        uint4 v; v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
        *(uint4 *)(txt + 64 * lane + 16 * k) = v;
    }
    s_oth[lane] = oth; s_othx[lane] = othx;
    const bool prev_eol = c_eol(P.text[tb - 1]);
    // k_enc_last's three values
    {
        const u32 last_e = (u32)(eol >> 63), up = (u32)__shfl_up((int)last_e, 1, 64);
        const u32 prevbit = lane ? up : (prev_eol ? 1u : 0u);
        const u32 nls = wave_scan_inclusive<u32, OpAdd>((u32)__popcll(~eol & ((eol << 1) | prevbit)));
        const u64 be = __ballot(eol != 0), bs = __ballot(sp != 0);
        const u32 pe = eol ? 64 * lane + 64 - (u32)__clzll((long long)eol) : 0u, ps = sp ? 64 * lane + 64 - (u32)__clzll((long long)sp) : 0u;
        const u32 pe_t = be ? (u32)__shfl((int)pe, 63 - __clzll((long long)be), 64) : 0u, ps_t = bs ? (u32)__shfl((int)ps, 63 - __clzll((long long)bs), 64) : 0u;
        if (lane == 63) { tile_eol[tile] = pe_t ? (i64)(tb + pe_t - 1) : -1; tile_sp[tile] = ps_t ? (i64)(tb + ps_t - 1) : -1; tile_ls[tile] = nls; }
    }
    // the segments
    const u32 cnt = (u32)__popcll(nl), incl = wave_scan_inclusive<u32, OpAdd>(cnt), k0 = incl - cnt;
    const u32 tot = (u32)__shfl((int)incl, 63, 64);
    if (__ballot(high) || tot >= FQR_MAXSEG) { if (lane == 0) cand[tile].flags = 0; return; }   // (uniform)
    { u64 m = nl; u32 idx = k0; while (m) { s_nl[idx++] = (u16)(64 * lane + (u32)__ffsll((long long)m) - 1); m &= m - 1; } }
    // a lane's own bytes: what its segments' classes could not be (letters or qualities with a byte below 0x21; a '+' line with 0x0B..0x0D)
    u32 own = 0;
    if (oth) {
        const u64 crm = eol & ~nl;
        u32 a = 0, k = k0; u64 m = nl;
        for (;;) {
            const u32 b = m ? (u32)__ffsll((long long)m) - 1 : 64u;
            if (b > a) { const u64 r = bits_from_to(a, b); if (oth & r) own |= 1u << (k & 3); if (crm & r) own |= 16u << (k & 3); }
            if (!m) break;
            m &= m - 1; a = b + 1; k++;
        }
    }
#pragma unroll
    for (int d = 32; d; d >>= 1) own |= (u32)__shfl_xor((int)own, d, 64);
    __syncthreads();
    // a lane per segment, 64 a round; the lane's class is lane & 3 in every round
    u64 sums = 0; u32 badp = 0, badh = 0, empty = 0;
    u32 I0 = 0, C0 = 0, C0s = 0, bad0 = 0;
    for (u32 kb = 0; kb <= tot; kb += 64) {
        const u32 k = kb + lane;
        if (k <= tot) {
            const u32 s_ = k ? (u32)s_nl[k - 1] + 1 : 0u, e_ = k < tot ? (u32)s_nl[k] : (u32)ET_TILE, len = e_ - s_;
            const bool starts = k > 0 || prev_eol, closed = k < tot;
            if (len == 0 && starts && closed) empty = 1;
            const u32 first = len ? (u32)txt[s_] : 0u;
            if (starts && len && first != '+') badp = 1;
            u32 a = s_; bool bh = false;
            if (starts && len) { if (first != '@') bh = true; a++; }
            u32 f = e_; bool x_any = false, x_after = false;
            if (a < e_) for (u32 q = a >> 6; q <= (e_ - 1) >> 6; q++) {
                const u32 lo = 64 * q;
                const u64 r = bits_from_to(a > lo ? a - lo : 0u, e_ - lo < 64 ? e_ - lo : 64u);
                const u64 m = s_oth[q] & r; u64 mx = s_othx[q] & r;
                x_any |= mx != 0;
                if (f != e_) x_after |= mx != 0;
                else if (m) { const u32 bit = (u32)__ffsll((long long)m) - 1; f = lo + bit; mx &= ~((2ull << bit) - 1); x_after |= mx != 0; }
            }
            u32 nids = f - a, ncmt = 0;
            if (f < e_) { nids++; ncmt = e_ - (f + 1); const u32 ch = txt[f]; if (ch != 0x20 && ch != 0x09) bh = true; if (x_after) bh = true; }
            if (closed) { if (f == e_) nids++; ncmt++; }
            if (k == 0 && !starts) {                                  // the line in progress: kept apart, both readings
                I0 = nids; C0 = ncmt; C0s = e_ - a + (closed ? 1u : 0u);
                bad0 = (bh ? (u32)FQC_BADH0 : 0u) | (x_any ? (u32)FQC_BADH0S : 0u);
                sums += (u64)len;
            } else { sums += (u64)len | ((u64)nids << 16) | ((u64)ncmt << 32); if (bh) badh = 1; }
        }
    }
    // per class: lanes of one residue mod 4 add up
#pragma unroll
    for (int d = 32; d >= 4; d >>= 1) {
        sums += shfl_idx_t<u64>(sums, (int)(lane ^ (u32)d));
        badp |= (u32)__shfl_xor((int)badp, d, 64); badh |= (u32)__shfl_xor((int)badh, d, 64);
    }
    const u32 fp = (u32)(__ballot(badp != 0) & 15ull), fh = (u32)(__ballot(badh != 0) & 15ull);
    const bool any_empty = __ballot(empty != 0) != 0;
    FqCand *o = cand + tile;
    if (lane < 4) { o->L[lane] = (u16)(sums & 0xFFFF); o->I[lane] = (u16)((sums >> 16) & 0xFFFF); o->C[lane] = (u16)((sums >> 32) & 0xFFFF); }
    if (lane == 0) {
        o->I0 = (u16)I0; o->C0 = (u16)C0; o->C0s = (u16)C0s; o->nseg = (u16)(tot + 1);
        o->flags = (any_empty ? 0u : (u32)FQC_OK) | (prev_eol ? (u32)FQC_STARTS : 0u) | ((own & 15u) * FQC_BADS) | ((fp | ((own >> 4) & 15u)) * FQC_BADP) | (fh * FQC_BADH) | bad0;
    }
}
// The class each role falls to, a lane per tile: k_encq_count_reg's outputs.
__global__ void k_fq_pick(EncP P, const FqCand *cand, const i64 *tile_eol, const i64 *tile_sp, const u64 *t_ls,
                          u64 *t_seq, u64 *t_ids, u64 *t_cmt, u64 *t_qual, u32 *t_reg, u64 *t_need, u64 tiles, u32 *n_wide, u32 wave_max)
{
    // t_reg: 1 = regular, up to 64 segments (k_fq_scatter_wave); 2 = regular with more (k_encq_scatter_reg); 0 = the general kernels
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles) return;
    const FqCand c = cand[t];
    // (the four classes' counts as three words: a class is picked with a shift, not by indexing a private array)
    const u64 L4 = (u64)c.L[0] | ((u64)c.L[1] << 16) | ((u64)c.L[2] << 32) | ((u64)c.L[3] << 48), I4 = (u64)c.I[0] | ((u64)c.I[1] << 16) | ((u64)c.I[2] << 32) | ((u64)c.I[3] << 48),
              C4 = (u64)c.C[0] | ((u64)c.C[1] << 16) | ((u64)c.C[2] << 32) | ((u64)c.C[3] << 48);
    bool regular = (c.flags & FQC_OK) != 0;
    if (regular) {
        const bool starts = (c.flags & FQC_STARTS) != 0;
        const u32 ord0 = (u32)(t_ls[t] - (starts ? 0u : 1u)) & 3u;
        const u32 jh = (0u - ord0) & 3u, js = (1u - ord0) & 3u, jp = (2u - ord0) & 3u, jq = (3u - ord0) & 3u;
        u32 bad = (c.flags & (FQC_BADS << js)) | (c.flags & (FQC_BADS << jq)) | (c.flags & (FQC_BADP << jp)) | (c.flags & (FQC_BADH << jh));
        u32 ids = (u32)(I4 >> (16 * jh)) & 0xFFFFu, cmt = (u32)(C4 >> (16 * jh)) & 0xFFFFu;
        if (jh == 0 && !starts) {
            const i64 le = tile_eol[t - 1], lsp = tile_sp[t - 1];
            i64 ls0 = le + 1; if ((u64)ls0 < P.p0) ls0 = (i64)P.p0;
            if (lsp >= ls0) { cmt += c.C0s; bad |= c.flags & FQC_BADH0S; }
            else { ids += c.I0; cmt += c.C0; bad |= c.flags & FQC_BADH0; }
        }
        regular = bad == 0;
        if (regular) { t_seq[t] = (L4 >> (16 * js)) & 0xFFFFu; t_ids[t] = ids; t_cmt[t] = cmt; t_qual[t] = (L4 >> (16 * jq)) & 0xFFFFu; }
    }
    const bool wide = regular && c.nseg > wave_max;                 // (wave_max = 0xFFFF.. : none, every regular tile is 1)
    t_reg[t] = regular ? (wide ? 2u : 1u) : 0u; t_need[t] = regular ? 0u : 1u;
    if (wide) atomicAdd(n_wide, 1u);
}

__global__ void k_fq_compare(const u32 *r1, const u32 *r2, const u64 *s1, const u64 *i1, const u64 *c1, const u64 *q1, const u64 *x2, u64 tiles, u64 *first)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles) return;
    bool d = (r1[t] != 0) != (r2[t] != 0);
    if (!d && r1[t]) d = s1[t] != x2[t] || i1[t] != x2[(tiles + 2) + t] || c1[t] != x2[2 * (tiles + 2) + t] || q1[t] != x2[3 * (tiles + 2) + t];
    if (d) atomicMin((unsigned long long *)first, (unsigned long long)t);
}

template <bool PACK>
__global__ __launch_bounds__(256) void k_encq_scatter_reg(EncP P, const i64 *tile_eol, const i64 *tile_sp, FqOut O)
{
    const u64 tile = blockIdx.x, tb = tile * ET_TILE;
    if (O.t_reg[tile] != O.reg_kind) return;                                          // (uniform) 0: the general kernel takes it from the list; 1 where k_fq_scatter_wave runs: that kernel's
    __shared__ FqrLds S;
    __shared__ __attribute__((aligned(16))) u8 stage[ET_TILE + 512 + 5 * 16];
    const u32 tid = threadIdx.x;
    const u8 *src = P.text + tb + 16 * tid;
    const u64 w0 = ld64(src), w1 = ld64(src + 8);
    const u32 w[4] = { (u32)w0, (u32)(w0 >> 32), (u32)w1, (u32)(w1 >> 32) };
    u32 nlm, oth, k0, nseg; bool high;
    fqr_masks(w, nlm, oth, high);
    const u64 bases[4] = { O.t_seq[tile], O.t_ids[tile], O.t_cmt[tile], O.t_qual[tile] };
    fqr_segments<false>(P, tile, tile_eol, tile_sp, O.t_ls, S, w, nlm, oth, k0, nseg, bases);
    const u64 sbase = bases[0], qbase = bases[3];
    // every piece moves its sequence and quality bytes to their places in the staging buffer
    {
        const Piece pc = { w0, w1, 16 };
        u32 a = 0, m = nlm, k = k0;
        for (;;) {
            const u32 b = m ? (u32)__ffs((int)m) - 1 : 16u;
            if (b > a) {
                const i32 d = S.segdst[k];
                if (d != FQR_SKIP) { u64 lo, hi; piece_from(pc, a, lo, hi); lds_store_n(stage + (d + (i32)(16 * tid + a)), lo, hi, b - a); }
            }
            if (!m) break;
            m &= m - 1; a = b + 1; k++;
        }
    }
    // a lane per segment: the header's bytes to the ID and comment regions with their terminators, the record tables
    if (tid < nseg) {
        const u32 k = tid, tyf = S.segtype[k], ty = tyf & 3;
        const bool starts = tyf & 4, closed = tyf & 8;
        const u32 s_ = S.segse[k] & 0xFFFF, e_ = S.segse[k] >> 16;
        const u64 off = S.segoff[k];
        const u64 ord = O.t_ls[tile] - ((S.segtype[0] & 4) ? 0u : 1u) + k, rec = ord >> 2;
        if (ty == 0) {
            const u32 a = S.seghdr[k] & 0xFFFF, f = S.seghdr[k] >> 16;
            u8 *ip = stage + S.so[1] + (u32)((off >> 16) & 0xFFFF), *cp = stage + S.so[2] + (u32)((off >> 32) & 0xFFFF);
            u32 ca = a;                                                               // first comment byte
            if (f != 0xFFFFu) {
                for (u32 q = a; q < f; q += 16) { const u32 nb = f - q < 16 ? f - q : 16u; lds_store_n(ip, ld64(S.txt + q), ld64(S.txt + q + 8), nb); ip += nb; }
                if (f < e_ || closed) *ip = 0;
                ca = f < e_ ? f + 1 : e_;
            }
            for (u32 q = ca; q < e_; q += 16) { const u32 nb = e_ - q < 16 ? e_ - q : 16u; lds_store_n(cp, ld64(S.txt + q), ld64(S.txt + q + 8), nb); cp += nb; }
            if (closed) { *cp = 0; O.rec_begin[rec] = sbase + (off & 0xFFFF); }
        } else if (ty == 1) { if (closed) O.rec_end[rec] = sbase + (off & 0xFFFF) + (e_ - s_); }
        else if (ty == 3) {
            if (starts && e_ > s_) O.q_begin[rec] = qbase + (off >> 48);
            if (closed) O.q_end[rec] = qbase + (off >> 48) + (e_ - s_);
        }
    }
    __syncthreads();
    const u64 t = S.total;
    const u32 tots = (u32)(t & 0xFFFF), toti = (u32)((t >> 16) & 0xFFFF), totc = (u32)((t >> 32) & 0xFFFF), totq = (u32)(t >> 48);
    const u32 so = S.so[0];
    // the letters: a group of sixteen staged bases with a byte that is not A C G T/U N (either case) sends the tile to the general
    // kernel (an IUPAC code, a letter to be replaced and counted) -- before anything of its streams is written
    {
        bool odd = false;
        const u32 span = so + tots, ng = (span + 15) >> 4;
        for (u32 j = tid; j < ng; j += 256) {
            uint4 v = *(const uint4 *)(stage + 16 * j);
            u32 x[4] = { v.x, v.y, v.z, v.w };
            const u32 a = j == 0 ? so : 0u, b = span - 16 * j < 16 ? span - 16 * j : 16u;
            if (a || b < 16) {                                                        // what lies outside the tile's own bases reads as 'A'
                const u64 klo = low_bytes(b < 8 ? b : 8u) & ~low_bytes(a < 8 ? a : 8u), khi = low_bytes(b > 8 ? b - 8 : 0u) & ~low_bytes(a > 8 ? a - 8 : 0u);
                const u64 A = 0x4141414141414141ull;
                const u64 lo = ((((u64)x[1] << 32) | x[0]) & klo) | (A & ~klo), hi = ((((u64)x[3] << 32) | x[2]) & khi) | (A & ~khi);
                x[0] = (u32)lo; x[1] = (u32)(lo >> 32); x[2] = (u32)hi; x[3] = (u32)(hi >> 32);
            }
            if (!all_quick16(x, P.qlo, P.qhi)) odd = true;
        }
        if (__syncthreads_or(odd)) {
            if (tid == 0) O.redo_list[atomicAdd(O.n_redo, 1u)] = (u32)tile;
            return;
        }
    }
    if (PACK) flush_pack<true>(P, O.packed, O.casebits, sbase, tots, stage);
    else flush_congruent(O.seq + sbase, stage + so, tots, tid, 256);
    flush_congruent(O.qual + qbase, stage + S.so[3], totq, tid, 256);
    // the two small streams by a wavefront each
    if (tid >= 192) flush_congruent(O.ids + bases[1], stage + S.so[1], toti, tid - 192, 64);
    else if (tid >= 128) flush_congruent(O.cmt + bases[2], stage + S.so[2], totc, tid - 128, 64);
}

// ---- regular tiles of up to 64 segments, a WAVEFRONT per tile -------------------------------------------------------------------------
// k_encq_scatter_reg's work without its barriers: a tile's chain there is a load, four workgroup barriers around one wavefront's walk
// over the segments, the moves, two more barriers, the flush -- 6 us for a tile with eight workgroups to a CU, 1 TB/s.  Here a lane
// holds 64 bytes of the tile in registers, the segment walk is the same lane-per-segment arithmetic (no checks: the count pass made
// them), header bytes move from the registers like the letters and qualities (the text is not staged at all), and a CU holds twenty
// tiles in flight.  Same streams, same record tables, same hand-back of tiles whose letters want the general kernel.
#define FQW_MAXSEG 64
template <bool PACK>
__global__ __launch_bounds__(64) void k_fq_scatter_wave(EncP P, const i64 *tile_eol, const i64 *tile_sp, FqOut O)
{
    const u64 tile = xcd_block(), tb = tile * ET_TILE;
    if (O.t_reg[tile] != 1u) return;
    __shared__ __attribute__((aligned(16))) u8 stage[ET_TILE + 512 + 5 * 16];
    __shared__ u64 s_oth[64];
    __shared__ u16 s_nl[FQW_MAXSEG];
    __shared__ i32 s_dst[FQW_MAXSEG];            // letters and qualities: where the segment's first byte goes in the staging buffer, minus its position; FQR_SKIP: a '+' line; header: the ID's place minus the position of its first byte
    __shared__ u32 s_hdr[FQW_MAXSEG];            // header segments: first ID byte | the position behind the last one << 16
    __shared__ i32 s_cdst[FQW_MAXSEG];           // header segments: the comment's place minus the position of its first byte ...
    __shared__ u16 s_ca[FQW_MAXSEG];             // ... and that position
    __shared__ u8 s_ty[FQW_MAXSEG];
    const u32 lane = threadIdx.x;
    const u8 *src = P.text + tb + 64 * lane;
    u64 wv[8];
#pragma unroll
    for (u32 k = 0; k < 4; k++) { uint4 q; __builtin_memcpy(&q, src + 16 * k, 16); wv[2 * k] = (u64)q.x | ((u64)q.y << 32); wv[2 * k + 1] = (u64)q.z | ((u64)q.w << 32); }
    const u64 bases[4] = { O.t_seq[tile], O.t_ids[tile], O.t_cmt[tile], O.t_qual[tile] };
    const i64 le = tile_eol[tile - 1], lsp = tile_sp[tile - 1];
    const u64 ls_tile = O.t_ls[tile];
    u64 nl = 0, oth = 0;
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
        const u32 w[4] = { (u32)wv[2 * k], (u32)(wv[2 * k] >> 32), (u32)wv[2 * k + 1], (u32)(wv[2 * k + 1] >> 32) };
        u32 nlm, o; bool hi;
        fqr_masks(w, nlm, o, hi);
        nl |= (u64)nlm << (16 * k); oth |= (u64)o << (16 * k);
    }
    s_oth[lane] = oth;
    const u32 cnt = (u32)__popcll(nl), incl = wave_scan_inclusive<u32, OpAdd>(cnt), k0 = incl - cnt;
    const u32 tot = (u32)__shfl((int)incl, 63, 64);                                   // (< FQW_MAXSEG: k_fq_pick)
    { u64 m = nl; u32 idx = k0; while (m) { s_nl[idx++] = (u16)(64 * lane + (u32)__ffsll((long long)m) - 1); m &= m - 1; } }
    __syncthreads();
    const bool prev_eol = le == (i64)tb - 1;
    i64 ls0 = le + 1; if ((u64)ls0 < P.p0) ls0 = (i64)P.p0;
    const u64 ord0 = ls_tile - (prev_eol ? 0u : 1u);
    // a lane per segment
    const u32 k = lane;
    u64 c = 0; u32 s_ = 0, e_ = 0, ty = 0, a = 0, f = 0; bool starts = false, closed = false;
    if (k <= tot) {
        s_ = k ? (u32)s_nl[k - 1] + 1 : 0u; e_ = k < tot ? (u32)s_nl[k] : (u32)ET_TILE;
        const u32 len = e_ - s_;
        starts = k > 0 || prev_eol; closed = k < tot;
        ty = (u32)((ord0 + k) & 3);
        if (ty == 1) c = (u64)len;
        else if (ty == 3) c = (u64)len << 48;
        else if (ty == 0) {
            a = s_; if (starts && len) a++;
            const bool seen = !starts && lsp >= ls0;
            f = e_;
            if (!seen && a < e_) for (u32 q = a >> 6; q <= (e_ - 1) >> 6 && f == e_; q++) {
                const u32 lo = 64 * q;
                const u64 m = s_oth[q] & bits_from_to(a > lo ? a - lo : 0u, e_ - lo < 64 ? e_ - lo : 64u);
                if (m) f = lo + (u32)__ffsll((long long)m) - 1;
            }
            u32 nids, ncmt;
            if (seen) { nids = 0; ncmt = e_ - a; f = 0xFFFFu; }
            else { nids = f - a; ncmt = 0; if (f < e_) { nids++; ncmt = e_ - (f + 1); } }
            if (closed) { if (!seen && f == e_) nids++; ncmt++; }
            c = ((u64)nids << 16) | ((u64)ncmt << 32);
        }
    }
    const u64 inc = wave_scan_inclusive<u64, OpAdd>(c), off = inc - c;
    const u64 total = shfl_idx_t<u64>(inc, 63);
    const u32 tots = (u32)(total & 0xFFFF), toti = (u32)((total >> 16) & 0xFFFF), totc = (u32)((total >> 32) & 0xFFFF), totq = (u32)(total >> 48);
    // the staging buffer: the four streams' shares one behind the other, each congruent to its stream's address mod 16 (fqr_segments)
    const u32 sO = (u32)(bases[0] & 15);
    const u32 qO = ((sO + tots + 15) & ~15u) + (u32)(bases[3] & 15);
    const u32 iO = ((qO + totq + 15) & ~15u) + (u32)(bases[1] & 15);
    const u32 cO = ((iO + toti + 15) & ~15u) + (u32)(bases[2] & 15);
    const u64 sbase = bases[0], qbase = bases[3];
    if (k <= tot) {
        const u64 rec = (ord0 + k) >> 2;
        i32 d = FQR_SKIP; u32 hdr = 0; i32 cd = 0;
        if (ty == 1) { d = (i32)(sO + (u32)(off & 0xFFFF)) - (i32)s_; if (closed) O.rec_end[rec] = sbase + (off & 0xFFFF) + (e_ - s_); }
        else if (ty == 3) {
            d = (i32)(qO + (u32)(off >> 48)) - (i32)s_;
            if (starts && e_ > s_) O.q_begin[rec] = qbase + (off >> 48);
            if (closed) O.q_end[rec] = qbase + (off >> 48) + (e_ - s_);
        } else if (ty == 0) {
            const u32 ip = iO + (u32)((off >> 16) & 0xFFFF), cp = cO + (u32)((off >> 32) & 0xFFFF);
            u32 ca = a;                                                               // first comment byte
            if (f != 0xFFFFu) { if (f < e_ || closed) stage[ip + (f - a)] = 0; ca = f < e_ ? f + 1 : e_; }
            if (closed) { stage[cp + (e_ - ca)] = 0; O.rec_begin[rec] = sbase + (off & 0xFFFF); }
            d = (i32)ip - (i32)a; cd = (i32)cp - (i32)ca;
            hdr = a | ((f == 0xFFFFu ? a : f) << 16);
            s_ca[k] = (u16)ca;
        }
        s_dst[k] = d; s_hdr[k] = hdr; s_cdst[k] = cd; s_ty[k] = (u8)ty;
    }
    __syncthreads();
    // every piece moves its bytes: letters and qualities to their segment's place, a header's to the ID and the comment
    {
        u32 seg = k0;
#pragma unroll
        for (u32 i = 0; i < 4; i++) {
            const Piece pc = { wv[2 * i], wv[2 * i + 1], 16 };
            const u32 pb = 64 * lane + 16 * i;
            u32 x = 0, m = (u32)(nl >> (16 * i)) & 0xFFFFu;
            for (;;) {
                const u32 b = m ? (u32)__ffs((int)m) - 1 : 16u;
                if (b > x) {
                    const u32 t = s_ty[seg]; const i32 d = s_dst[seg];
                    if (t & 1) { u64 lo, hi; piece_from(pc, x, lo, hi); lds_store_n(stage + (d + (i32)(pb + x)), lo, hi, b - x); }
                    else if (t == 0) {
                        const u32 h = s_hdr[seg], ha = h & 0xFFFF, ie = h >> 16, ca = s_ca[seg];
                        const u32 p0 = pb + x, p1 = pb + b;                            // the portion, in tile positions
                        const u32 i0 = p0 > ha ? p0 : ha, i1 = p1 < ie ? p1 : ie;
                        if (i1 > i0) { u64 lo, hi; piece_from(pc, i0 - pb, lo, hi); lds_store_n(stage + (d + (i32)i0), lo, hi, i1 - i0); }
                        const u32 c0 = p0 > ca ? p0 : ca;
                        if (p1 > c0) { u64 lo, hi; piece_from(pc, c0 - pb, lo, hi); lds_store_n(stage + (s_cdst[seg] + (i32)c0), lo, hi, p1 - c0); }
                    }
                }
                if (!m) break;
                m &= m - 1; x = b + 1; seg++;
            }
        }
    }
    __syncthreads();
    // the letters: a group of sixteen staged bases with a byte that is not A C G T/U N (either case) sends the tile to the general kernel
    {
        bool odd = false;
        const u32 span = sO + tots, ng = (span + 15) >> 4;
        for (u32 j = lane; j < ng; j += 64) {
            uint4 v = *(const uint4 *)(stage + 16 * j);
            u32 xw[4] = { v.x, v.y, v.z, v.w };
            const u32 a0 = j == 0 ? sO : 0u, b0 = span - 16 * j < 16 ? span - 16 * j : 16u;
            if (a0 || b0 < 16) {                                                      // what lies outside the tile's own bases reads as 'A'
                const u64 klo = low_bytes(b0 < 8 ? b0 : 8u) & ~low_bytes(a0 < 8 ? a0 : 8u), khi = low_bytes(b0 > 8 ? b0 - 8 : 0u) & ~low_bytes(a0 > 8 ? a0 - 8 : 0u);
                const u64 A = 0x4141414141414141ull;
                const u64 lo = ((((u64)xw[1] << 32) | xw[0]) & klo) | (A & ~klo), hi = ((((u64)xw[3] << 32) | xw[2]) & khi) | (A & ~khi);
                xw[0] = (u32)lo; xw[1] = (u32)(lo >> 32); xw[2] = (u32)hi; xw[3] = (u32)(hi >> 32);
            }
            if (!all_quick16(xw, P.qlo, P.qhi)) odd = true;
        }
        if (__ballot(odd)) {
            if (lane == 0) O.redo_list[atomicAdd(O.n_redo, 1u)] = (u32)tile;
            return;
        }
    }
    if (PACK) flush_pack<true>(P, O.packed, O.casebits, sbase, tots, stage);
    else flush_congruent(O.seq + sbase, stage + sO, tots, lane, 64);
    flush_congruent(O.qual + qbase, stage + qO, totq, lane, 64);
    flush_congruent(O.ids + bases[1], stage + iO, toti, lane, 64);
    flush_congruent(O.cmt + bases[2], stage + cO, totc, lane, 64);
}

struct SmallBytes { u8 b[60]; u32 n; };
__global__ void k_put_bytes(u8 *dst, SmallBytes s) { if (threadIdx.x < s.n) dst[threadIdx.x] = s.b[threadIdx.x]; }
// read lengths, quality-length check (process.c:531-535) and the longest read
__global__ void k_fq_check(const u64 *rec_begin, const u64 *rec_end, const u64 *q_begin, const u64 *q_end, u64 N, u64 *first_error, u64 *longest)
{
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    u64 len = rec_end[r] - rec_begin[r], ql = q_end[r] - q_begin[r];
    if (len != ql) atomicMin((unsigned long long *)first_error, (unsigned long long)(r * 4 + FQ_E_QLEN));
    if (len > __atomic_load_n(longest, __ATOMIC_RELAXED)) atomicMax((unsigned long long *)longest, (unsigned long long)len);
}

// ---- lengths: u32 units with 0xFFFFFFFF continuation (encoders.c:72-95) ----------------------------------------------------
__global__ void k_len_unit_count(const u64 *rec_begin, const u64 *rec_end, u64 N, u64 total_bases, u64 *units, int all_ends)
{
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    u64 end = (all_ends || r + 1 < N) ? rec_end[r] : total_bases;
    u64 len = end - rec_begin[r];
    units[r] = len / 0xFFFFFFFFull + 1;
}
__global__ void k_len_unit_write(const u64 *rec_begin, const u64 *rec_end, u64 N, u64 total_bases, const u64 *unit_off, u32 *out, int all_ends)
{
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    u64 end = (all_ends || r + 1 < N) ? rec_end[r] : total_bases;
    u64 len = end - rec_begin[r], o = unit_off[r];
    while (len >= 0xFFFFFFFFull) { out[o++] = 0xFFFFFFFFu; len -= 0xFFFFFFFFull; }
    out[o] = (u32)len;
}

// ---- soft mask: boundaries of (byte >= 96) runs -> u8 units with 255 continuation (encoders.c:98-146) -----------------------
__global__ void k_add_u64(u64 *p, u64 v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += v; }
// run r (0-based) spans [start_r, start_{r+1}) with start_0 = 0, start_{r} = bnd[r-1]; the last one ends at T + ext, ext = the
// bases of the following shards that continue it (0 for a whole input).  skip0: run 0 -- the bases in front of this shard's first
// case change -- continues a run of an earlier shard, which emits its units.
__global__ void k_mask_run_units(const u64 *bnd, u64 nb, u64 T, u64 *units, u64 ext, int skip0)
{
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nb) return;
    u64 s = r ? bnd[r - 1] : 0, e = r < nb ? bnd[r] : T + ext;
    units[r] = (skip0 && r == 0) ? 0 : (e - s) / 255 + 1;
}
// The units of a mask whose runs are ALL shorter than 255 bases -- reads whose case changes every few bases: 170 M runs per 4 GB of
// such a FASTQ -- are one per run, in run order: written here on that assumption, run r's unit at r (r - 1 behind a skipped run 0).
// *any_long is raised by a run that needs more than one unit; the caller then takes the general way (unit counts, their scan, the
// read-back of their sum, k_mask_units_write) and this kernel's output is dropped.
__global__ void k_mask_units_short(const u64 *bnd, u64 nb, u64 T, u8 *out, u64 ext, int skip0, u64 *any_long)
{
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nb || (skip0 && r == 0)) return;
    u64 s = r ? bnd[r - 1] : 0, e = r < nb ? bnd[r] : T + ext, len = e - s;
    if (len >= 255) { *any_long = 1; return; }
    out[r - (skip0 ? 1 : 0)] = (u8)len;
}
// every unit that is not the last of its run is 255: the array is pre-filled with 255 and one lane per run writes the remainder
__global__ void k_mask_units_write(const u64 *bnd, u64 nb, u64 T, const u64 *unit_off, u8 *out, u64 ext, int skip0)
{
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nb || (skip0 && r == 0)) return;
    u64 s = r ? bnd[r - 1] : 0, e = r < nb ? bnd[r] : T + ext, len = e - s;
    out[unit_off[r] + len / 255] = (u8)(len % 255);
}

// ---- the same scan over the case BITS that flush_pack leaves for a 4-bit stream: 64 bases per lane, 16384 per tile -------------------
#define MBB_TILE (256 * 64)
__device__ __forceinline__ u64 maskb_bits(const u64 *cb, u64 i, u64 T, bool prev0)    // boundaries among bases 64 i .. 64 i + 63
{
    const u64 w = cb[i];
    const u64 prev = i ? cb[i - 1] >> 63 : (prev0 ? 1ull : 0ull);
    u64 m = w ^ ((w << 1) | prev);
    if (T - 64 * i < 64) m &= (1ull << (T - 64 * i)) - 1;
    return m;
}
// tile_last (may be null): position + 1 of the tile's last case change, 0 when it has none (k_maskb_units_direct)
__global__ __launch_bounds__(256) void k_maskb_count(const u64 *cb, u64 T, u64 *tile_cnt, int prev0, i64 *tile_last = nullptr)
{
    __shared__ u32 s_c[4]; __shared__ u64 s_l[4];
    const u32 bid = blockIdx.x;
    const u64 i = (u64)bid * 256 + threadIdx.x;
    const u64 m = 64 * i < T ? maskb_bits(cb, i, T, prev0 != 0) : 0;
    u32 tot = wg_reduce1<u32, OpAdd>((u32)__popcll(m), s_c);
    if (threadIdx.x == 0) tile_cnt[bid] = tot;
    if (tile_last) {
        const u64 last = wg_reduce1<u64, OpMaxU64>(m ? 64 * i + 64 - (u32)__clzll((long long)m) : 0ull, s_l);
        if (threadIdx.x == 0) tile_last[bid] = (i64)last;
    }
}
// The units of a mask whose runs are all shorter than 255 bases straight from the case bits: run r's unit is the distance of case change
// r from the one before it, written at r -- the case change in front of a lane's first one comes from a running maximum over the lanes
// and, across tiles, from the scan of tile_last.  (The way over the list of positions writes and reads eight bytes per case change:
// 16 GB for the billion case changes of 12.5 GB of mixed-case reads.)  *any_long is raised by a run of 255 bases or more; the caller
// then takes the general way and this kernel's output is dropped.  The last run, from the last case change to T + ext, is lane 0's.
__global__ __launch_bounds__(256) void k_maskb_units_direct(const u64 *cb, u64 T, const u64 *tile_pre, const i64 *last_scan, u64 nb, u8 *out, u64 ext, int skip0, int prev0, u64 *any_long)
{
    __shared__ u64 lds[4], ldm[4];
    const u64 t = blockIdx.x, i = t * 256 + threadIdx.x;
    if (t == 0 && threadIdx.x == 0) {
        const u64 lastb = (u64)last_scan[gridDim.x - 1];             // position + 1 of the last case change, 0: none
        const u64 len = T + ext - (lastb ? lastb - 1 : 0);
        if (!(skip0 && nb == 0)) { if (len >= 255) *any_long = 1; else out[nb - (skip0 ? 1 : 0)] = (u8)len; }
    }
    if ((t + 1 < gridDim.x ? tile_pre[t + 1] : nb) == tile_pre[t]) return;
    u64 m = 64 * i < T ? maskb_bits(cb, i, T, prev0 != 0) : 0;
    const u64 c = (u64)__popcll(m);
    u64 tot, mx;
    const u64 incl = wg_scan_inclusive<u64, OpAdd>(c, &tot, lds);
    const u64 mine = m ? 64 * i + 64 - (u32)__clzll((long long)m) : 0ull;
    const u64 minc = wg_scan_inclusive<u64, OpMaxU64>(mine, &mx, ldm);
    u64 prevp = shfl_up_t<u64>(minc, 1);                          // position + 1 of the last case change in front of this lane
    if ((threadIdx.x & 63) == 0) prevp = 0;
    // (wg_scan_inclusive's value is inclusive over the workgroup: the lane in front across a wavefront's edge)
    __shared__ u64 s_edge[4];
    if ((threadIdx.x & 63) == 63) s_edge[threadIdx.x >> 6] = minc;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && threadIdx.x) prevp = s_edge[(threadIdx.x >> 6) - 1];
    const u64 carry = t ? (u64)last_scan[t - 1] : 0ull;
    if (carry > prevp) prevp = carry;
    u64 prev = prevp ? prevp - 1 : 0;                               // (no case change in front: the run began at base 0)
    // The tile's units are staged in LDS and leave as aligned 16-byte stores: a lane's dozen single-byte stores, lane after lane, were
    // 1 G partial writes for the mask of 12.5 GB of mixed-case reads (2.1 ms for 1 GB of units).
    __shared__ __attribute__((aligned(16))) u8 s_out[MBB_TILE + 32];
    const u64 base_k = tile_pre[t];
    const u64 drop = (skip0 && t == 0) ? 1u : 0u;                  // run 0 belongs to the shard in front: its unit is not written
    u8 *dst = out + (base_k + drop - (skip0 ? 1 : 0));             // where the tile's first written unit goes
    const u32 pad = (u32)((uintptr_t)dst & 15);
    u32 j = (u32)(incl - c);                                        // the unit's number in the tile
    bool lng = false;
    while (m) {
        const int b = __ffsll((unsigned long long)m) - 1; m &= m - 1;
        const u64 pos = 64 * i + (u32)b, len = pos - prev;
        prev = pos;
        if (j >= drop) { if (len >= 255) lng = true; s_out[pad + j - (u32)drop] = (u8)len; }
        j++;
    }
    if (lng) *any_long = 1;
    __syncthreads();
    if (tot > drop) flush_congruent(dst, s_out + pad, (u32)(tot - drop), threadIdx.x, 256);
}
__global__ __launch_bounds__(256) void k_maskb_scatter(const u64 *cb, u64 T, const u64 *tile_pre, u64 nb, u64 *bnd, int prev0)
{
    __shared__ u64 lds[4];
    if ((blockIdx.x + 1 < gridDim.x ? tile_pre[blockIdx.x + 1] : nb) == tile_pre[blockIdx.x]) return;
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    u64 m = 64 * i < T ? maskb_bits(cb, i, T, prev0 != 0) : 0;
    u64 c = (u64)__popcll(m), tot;
    u64 incl = wg_scan_inclusive<u64, OpAdd>(c, &tot, lds);
    u64 k = tile_pre[blockIdx.x] + incl - c;
    while (m) { int b = __ffsll((unsigned long long)m) - 1; m &= m - 1; bnd[k++] = 64 * i + (u32)b; }
}
// The census of a shard: case changes INSIDE the shard (positions >= 1) per tile, and their first and last position -- what the
// neighbours need to continue a run across the cut.  out: [0] first internal boundary (~0: none), [1] last internal boundary (0: none).
// One atomic per workgroup at most, and only when it can still improve the value (mixed-case reads change case every few bases).
// tile_last: position + 1 of the tile's last change, 0 when it has none -- k_maskb_count's table (every later tile's last change is larger than
// every earlier one's: an atomicMax per workgroup on ONE address was 176 K atomics in a row, 5 ms of a 12.5 GB shard; its running maximum
// is also what k_maskb_units_direct wants, which a shard could not use without it)
__global__ __launch_bounds__(256) void k_maskb_census(const u64 *cb, u64 T, u64 *tile_cnt, unsigned long long *out, i64 *tile_last)
{
    __shared__ u32 s_c[4]; __shared__ u64 s_lo[4], s_hi[4];
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 m = 64 * i < T ? maskb_bits(cb, i, T, (cb[0] & 1) != 0) : 0;      // prev0 = the first base's own case: position 0 never counts
    const u64 first = m ? 64 * i + (u32)__ffsll((unsigned long long)m) - 1 : ~0ull, last = m ? 64 * i + 63 - (u32)__clzll((long long)m) : 0ull;
    // positions grow with the lane: the first lane with a change holds the tile's first, the last one its last
    const u64 bal = __ballot(m != 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (bal && lane == __ffsll((unsigned long long)bal) - 1) s_lo[wave] = first;
    if (bal && lane == 63 - __clzll((long long)bal)) s_hi[wave] = last;
    if (!bal && lane == 0) { s_lo[wave] = ~0ull; s_hi[wave] = 0; }
    u32 tot = wg_reduce1<u32, OpAdd>((u32)__popcll(m), s_c);
    if (threadIdx.x == 0) {
        tile_cnt[blockIdx.x] = tot;
        u64 lo = ~0ull, hi = 0;
        for (int w = 0; w < 4; w++) { lo = s_lo[w] < lo ? s_lo[w] : lo; hi = s_hi[w] > hi ? s_hi[w] : hi; }
        if (lo < __atomic_load_n(&out[0], __ATOMIC_RELAXED)) atomicMin(&out[0], (unsigned long long)lo);
        tile_last[blockIdx.x] = hi ? (i64)(hi + 1) : 0;
    }
}
// first and last base of a packed stream as letters (upper case from the table of unnaf.c:13, lower when the case bit is set)
__global__ void k_packed_ends(const u8 *packed, const u64 *cb, u64 T, unsigned long long *out)
{
    if (threadIdx.x || blockIdx.x || !T) return;
    const char *tab = "-TGKCYSBAWRDMHVN";
    const u32 c0 = packed[0] & 15, c1 = (packed[(T - 1) >> 1] >> (4 * ((T - 1) & 1))) & 15;
    u32 a = (u32)tab[c0], b = (u32)tab[c1];
    if (cb) { if (c0 && (cb[0] & 1)) a |= 0x20; if (c1 && ((cb[(T - 1) >> 6] >> ((T - 1) & 63)) & 1)) b |= 0x20; }
    *out = (unsigned long long)a | ((unsigned long long)b << 8);
}
// The pack window of a shard whose first base completes the previous shard's last byte (encoders.c:30-69 keeps that half byte in
// `parity` between chunks): every code moves down one nibble.  n_in = bytes of the shard's own packed stream.
__global__ void k_nibble_shift(const u8 *in, u64 n_in, u8 *out, u64 n_out)
{
    const u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= n_out) return;
    if (i + 9 <= n_in && i + 8 <= n_out) { const u64 a = ld64(in + i); st64(out + i, (a >> 4) | ((u64)in[i + 8] << 60)); return; }
    for (u64 k = i; k < n_out && k < i + 8; k++) { u32 lo = in[k] >> 4, hi = k + 1 < n_in ? in[k + 1] & 15u : 0u; out[k] = (u8)(lo | (hi << 4)); }
}
__global__ void k_set_high_nibble(u8 *p, u32 code) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = (u8)((*p & 15u) | (code << 4)); }

__global__ void k_toupper(u8 *p, u64 n)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { u32 c = p[i]; if (c >= 'a' && c <= 'z') p[i] = (u8)(c - 32); }
}

// ---- format sniffing (process.c:547-583) -----------------------------------------------------------------------------------------
__global__ void k_sniff(const u8 *text, u64 n, u64 *out /* p0, first char, prev char */)
{
    int lane = threadIdx.x;
    for (u64 base = 0; base < n; base += 64) {
        u64 i = base + lane;
        u32 c = i < n ? text[i] : 0x100;
        bool nonspace = i < n && !c_space(c);
        u64 m = __ballot(nonspace);
        if (m) {
            int first = __ffsll((long long)m) - 1;
            if (lane == first) { out[0] = i; out[1] = c; out[2] = i ? text[i - 1] : '\n'; }
            return;
        }
    }
    if (lane == 0) { out[0] = n; out[1] = 0x100; out[2] = '\n'; }
}

// ---- shard cuts ------------------------------------------------------------------------------------------------------------------------
// FASTQ records are four non-empty lines (process.c:477-544), so a slice of the text can only be cut where the ordinal of a line
// start is a multiple of four; the ordinals come from a census of every slice.  A line start is a non-EOL byte behind an EOL-class
// byte -- the rule of count_line_starts above, for a slice whose first byte may sit in the middle of a line.
#define LC_TILE 4096
__device__ __forceinline__ u32 slice_line_starts16(const u8 *t, u64 base, u64 n, int prev_is_eol)
{
    u32 m = 0; bool prev = base ? c_eol(t[base - 1]) : prev_is_eol != 0;
    if (base + 16 <= n) {                                         // four bytes per instruction (enc_swar.h); only the EOL bits are kept
        const u64 a = ld64(t + base), b = ld64(t + base + 8);
        const u32 w[4] = { (u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32) };
        const u32 eol = piece_flags(w).eol;
        return ~eol & ((eol << 1) | (prev ? 1u : 0u)) & 0xFFFFu;
    }
    for (u32 i = 0; i < 16 && base + i < n; i++) { const bool e = c_eol(t[base + i]); if (!e && prev) m |= 1u << i; prev = e; }
    return m;
}
__global__ __launch_bounds__(256) void k_line_count(const u8 *t, u64 n, int prev_is_eol, u64 *tile_cnt)
{
    __shared__ u32 s_c[4];
    const u64 base = (u64)blockIdx.x * LC_TILE + (u64)threadIdx.x * 16;
    const u32 tot = wg_reduce1<u32, OpAdd>(base < n ? (u32)__popc(slice_line_starts16(t, base, n, prev_is_eol)) : 0u, s_c);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot;
}
// position of line start number `want` (0-based) of the slice; n when it has fewer.  One workgroup: the tile is found by bisection
// over the scanned tile counts, the byte inside it by a scan of the 256 pieces.
__global__ __launch_bounds__(256) void k_line_find(const u8 *t, u64 n, int prev_is_eol, const u64 *tile_pre, u64 tiles, u64 total, u64 want, u64 *out)
{
    __shared__ u64 lds[4];
    if (want >= total) { if (threadIdx.x == 0) *out = n; return; }
    u64 lo = 0, hi = tiles;                                       // last tile with tile_pre <= want
    while (hi - lo > 1) { const u64 mid = (lo + hi) / 2; if (tile_pre[mid] <= want) lo = mid; else hi = mid; }
    const u64 base = lo * LC_TILE + (u64)threadIdx.x * 16;
    const u32 m = base < n ? slice_line_starts16(t, base, n, prev_is_eol) : 0u;
    u64 tot; const u64 incl = wg_scan_inclusive<u64, OpAdd>((u64)__popc(m), &tot, lds);
    const u64 first = tile_pre[lo] + incl - __popc(m);            // ordinal of this piece's first line start
    if (m && want >= first && want < first + __popc(m)) {
        u32 mm = m; for (u64 k = first; k < want; k++) mm &= mm - 1;
        *out = base + (u32)__ffs((int)mm) - 1;
    }
}
// FASTA can be cut behind any EOL-class byte (the split kernels recover a byte's state from the last EOL in front of it, and the
// start of a slice acts as one): first such position, n when the slice has none.
__global__ __launch_bounds__(256) void k_eol_find(const u8 *t, u64 n, int prev_is_eol, unsigned long long *out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && prev_is_eol) atomicMin(out, 0ull);
    for (u64 q = (u64)blockIdx.x * 256 + threadIdx.x; q + 1 < n; q += (u64)gridDim.x * 256) {
        if (q + 1 >= __atomic_load_n(out, __ATOMIC_RELAXED)) return;       // something earlier is known already
        if (c_eol(t[q])) { atomicMin(out, (unsigned long long)(q + 1)); return; }
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------------------
static size_t vle(u64 v, u8 *out)                                 // encoders.c:175-190
{
    u8 tmp[10]; int n = 0;
    tmp[n++] = (u8)(v & 127); v >>= 7;
    while (v) { tmp[n++] = (u8)(128 | (v & 127)); v >>= 7; }
    for (int i = 0; i < n; i++) out[i] = tmp[n - 1 - i];
    return (size_t)n;
}

extern "C" size_t naf_gpu_ennaf_bound(size_t n) { return n + n / 256 + (1 << 16); }

// host-side copy of nuc4 (the high nibble a shard borrows from its neighbour's first base)
static u32 nuc4_host(u32 c)
{
    static const char tab[] = "-TGKCYSBAWRDMHVN";
    if (c == '-') return 0;
    u32 u = c & ~0x20u;
    if (u == 'U') return 1;
    if (u >= 'A' && u <= 'Z') for (u32 k = 1; k < 16; k++) if ((u32)tab[k] == u) return k;
    return 15;
}

static void set_expected(EncP &P, int seq_type, bool fasta)
{
    memset(P.expected, 0, sizeof P.expected);
    auto set = [&](u32 c) { P.expected[c >> 5] |= 1u << (c & 31); };
    if (seq_type == NAF_SEQ_DNA || seq_type == NAF_SEQ_RNA) {
        const char *a = seq_type == NAF_SEQ_DNA ? "ABCDGHKMNRSTVWY" : "ABCDGHKMNRSUVWY";       // tables.c:72-90
        for (const char *p = a; *p; p++) { set((u32)*p); set((u32)*p | 0x20); }
        set('-'); P.replacement = 'N';
    } else if (seq_type == NAF_SEQ_PROTEIN) {                                                     // tables.c:104-112
        for (u32 c = 'A'; c <= 'Z'; c++) { set(c); set(c | 0x20); }
        set('*'); set('-'); P.replacement = 'X';
    } else {                                                                                      // tables.c:115-123
        for (u32 c = 0x21; c <= 0xFE; c++) if (c != 0x7F) set(c);
        P.replacement = '?';
        P.id_gt_unexpected = fasta;
        // mid-line '>' is kept as data in text mode (process.c:410); a '>' right after an EOL starts a record
    }
    { u8 t[32]; for (u32 i = 0; i < 32; i++) t[i] = (i >= 1 && i <= 26) ? (u8)nuc4_host(0x40u + i) : 15; memcpy(P.nuc32, t, 32); }
    // quick table (enc_swar.h): slot (c >> 1) & 7 holds the accepted upper-case letter out of A C G T U N, 0xFF elsewhere
    u8 q[8]; memset(q, 0xFF, 8);
    auto has = [&](u32 c) { return (P.expected[c >> 5] >> (c & 31)) & 1u; };
    for (const char *p = "ACGTUN"; *p; p++) { u32 ch = (u32)*p; if (has(ch) && has(ch | 0x20) && q[(ch >> 1) & 7] == 0xFF) q[(ch >> 1) & 7] = (u8)ch; }
    memcpy(&P.qlo, q, 4); memcpy(&P.qhi, q + 4, 4);
    q[5] = 0x0A; q[6] = 0x0D; memcpy(&P.plo, q, 4); memcpy(&P.phi, q + 4, 4);   // (slots 5 and 6 belong to no letter of A C G T U N)
    // (0xFF sits in slot 7 and 0x00 in slot 0: neither equals the entry of another slot)
    { u8 t[8] = { q[0], q[1], q[2], q[3], 0x00, 0x0A, 0x00, 0x00 }; memcpy(&P.slo, t, 4); memcpy(&P.shi, t + 4, 4); }
}

// confirm_input_format (process.c:547-583): first non-space byte, the byte in front of it
static int ennaf_sniff(naf_gpu_ctx *c, const u8 *d_text, u64 n, int want_format, int *format, u64 *p0)
{
    u64 *d_sn = arena_new<u64>(c, 4); if (!d_sn) return NAF_GPU_ENOMEM;
    u64 sn[3] = { 0, 0x100, '\n' };
    if (n) { LAUNCH(c, "ennaf_sniff", k_sniff, 1, 64, 0, d_text, n, d_sn); int rc = ctx_readback(c, sn, d_sn, 24); if (rc) return rc; }
    *format = 0; *p0 = sn[0];
    if (sn[1] != 0x100) {
        bool at_line_start = sn[2] >= 0x0A && sn[2] <= 0x0D;
        if (sn[1] == '>' && at_line_start) *format = NAF_FMT_FASTA;
        else if (sn[1] == '@' && at_line_start) *format = NAF_FMT_FASTQ;
        else if (sn[1] == '>' || sn[1] == '@') return ctx_fail(c, NAF_GPU_EINPUT, "invalid input - first '%c' is not at the beginning of the line\n", (int)sn[1]);
        else return ctx_fail(c, NAF_GPU_EINPUT, "input data is in unknown format - first non-space character is neither '>' nor '@'\n");
        if (want_format != NAF_FMT_AUTO && want_format != *format) return ctx_fail(c, NAF_GPU_EINPUT, "input format is different from format specified in the command line\n");
    }
    return 0;
}

// What the text-split pass of one input -- or of one shard of it -- leaves in the arena, and the first reason the reference would
// have died for (record numbers local to the shard; the caller adds the records of the shards in front).
enum { SE_NONE = 0, SE_AT, SE_PLUS, SE_QLEN, SE_NOSEQ, SE_NOQUAL, SE_STRICT };
struct EnnafSplit {
    int format, seq_type; bool fourbit, store_mask, store_qual, no_mask;
    bool no_case = false;                                       // the count pass met no byte with the case bit in a sequence line: the mask is one run (FASTA, one call)
    u8 *bases, *s_ids, *s_cmt, *s_qual;                        // bases: one byte per base (protein / text only)
    u8 *packed; u64 *casebits;                                 // 4-bit: codes of the shard's own base stream (base 0 in the low nibble of byte 0), case bits (store_mask)
    u64 n_ids, n_cmt, n_qual, T, N, longest, lead;
    u64 *rec_begin, *rec_end; int all_ends;
    u64 unexpected[4][257];
    int err_kind; u32 err_char; u64 err_rec, err_a, err_b;
    // soft-mask census of a shard (positions >= 1); tc = per-tile counts kept for the finish
    bool census; u64 *tc; i64 *census_last; u64 mask_changes, mask_first, mask_last; u8 first_base, last_base;   // census_last: running maximum of the tiles' last change + 1 (k_maskb_census)
    // direct blocks (k_direct_blocks): flags of the first nd blocks of 32 KiB of `packed`; rescatter() packs the bases again without them
    // (the stream turned out to be worth matching: the match finder wants packed bytes)
    u8 *direct; u32 nd;
    ZencLoc dloc;                                              // dloc.loc != nullptr: the direct blocks' codes are tile-local (k_enc_fused read the text once)
    struct Scatter4 { EncP P; EncOut O; const i64 *t_eol, *t_sp; const u64 *t_seq; u64 tiles, T, n, n_irregular; } sc4;
};
// the scatter pass of a 4-bit FASTA input (O.direct says which blocks are direct)
static int ennaf_scatter4(naf_gpu_ctx *c, const EnnafSplit::Scatter4 &R, u8 *packed, u64 *casebits)
{
    if (R.T) LAUNCH(c, "ennaf_pack_edges", k_pack_edges_zero, cdiv(R.tiles, 256), 256, 0, R.t_seq, R.tiles, R.T, packed, (u32 *)casebits, R.O.direct, R.O.nd, R.O.loc_mode);
    if (R.O.loc_mode) LAUNCH(c, "ennaf_scatter_regular", k_enc_scatter_regular_list, 8192, 64, 0, R.P, R.t_eol, R.O);
    else if (R.n >= 2 * ET_TILE && enc_wave_wg(c)) LAUNCH(c, "ennaf_scatter_regular", k_enc_scatter_regular<1>, (u32)R.tiles, 64, 0, R.P, R.t_eol, R.O, R.tiles);
    else if (R.n >= 2 * ET_TILE) LAUNCH(c, "ennaf_scatter_regular", k_enc_scatter_regular<REG_TPW>, cdiv(R.tiles, REG_TPW), 256, 0, R.P, R.t_eol, R.O, R.tiles);   // (shorter texts have no regular tile)
    if (R.n_irregular) LAUNCH(c, "ennaf_scatter", k_enc_scatter<true>, R.n_irregular, 256, 0, R.P, R.t_eol, R.t_sp, R.O);
    return 0;
}

static int split_error_text(naf_gpu_ctx *c, int seq_type, int kind, u32 ch, u64 rec, u64 a, u64 b, u64 rec0)
{
    static const char *tn[4] = { "DNA", "RNA", "protein", "text" };
    const unsigned long long r = (unsigned long long)(rec0 + rec);
    switch (kind) {
    case SE_AT: return ctx_fail(c, NAF_GPU_EINPUT, "invalid FASTQ input: Can't find '@' after sequence %llu\n", r);
    case SE_PLUS: return ctx_fail(c, NAF_GPU_EINPUT, "invalid FASTQ input: can't find '+' line of sequence %llu\n", r + 1);
    case SE_QLEN: return ctx_fail(c, NAF_GPU_EINPUT, "quality length of sequence %llu (%llu) doesn't match sequence length (%llu)\n", r + 1, (unsigned long long)a, (unsigned long long)b);
    case SE_NOSEQ: return ctx_fail(c, NAF_GPU_EINPUT, "truncated FASTQ input: last sequence has no sequence data\n");
    case SE_NOQUAL: return ctx_fail(c, NAF_GPU_EINPUT, "truncated FASTQ input: last sequence has no quality\n");
    case SE_STRICT:                                                                                // process.c:98-140; rec is 1-based here
        if (a == 0) return ctx_fail(c, NAF_GPU_EINPUT, "unexpected character '%c' in ID of sequence %llu\n", (int)(unsigned char)ch, r);
        if (a == 1) return ctx_fail(c, NAF_GPU_EINPUT, "unexpected character '%c' in comment of sequence %llu\n", (int)(unsigned char)ch, r);
        if (a == 2) return ctx_fail(c, NAF_GPU_EINPUT, "unexpected %s code '%c' in sequence %llu\n", tn[seq_type & 3], (int)(unsigned char)ch, r);
        return ctx_fail(c, NAF_GPU_EINPUT, "unexpected quality code '%c' in sequence %llu\n", (int)(unsigned char)ch, r);
    default: return 0;
    }
}

// where the bases go: packed codes (+ case bits when the mask is stored) for a 4-bit stream, one byte per base otherwise
static int alloc_bases(naf_gpu_ctx *c, EnnafSplit &S, bool case_bits = true)
{
    if (S.fourbit) {
        S.packed = (u8 *)arena_alloc(c, (S.T + 1) / 2 + 64);
        if (!S.packed) return NAF_GPU_ENOMEM;
        if (S.store_mask && case_bits) { S.casebits = (u64 *)arena_alloc(c, (S.T + 63) / 64 * 8 + 64); if (!S.casebits) return NAF_GPU_ENOMEM; }
    } else { S.bases = (u8 *)arena_alloc(c, S.T + 64); if (!S.bases) return NAF_GPU_ENOMEM; }
    return 0;
}

// The split pass (E1-E4): text -> ids, comments, bases (1 B/base, post-replacement), quality, record table.  p0 = first byte of
// the first record; last_part: the text ends where the input ends (the truncation rules of process.c:499-520 apply).
static int ennaf_split(naf_gpu_ctx *c, const u8 *d_text, u64 n, const naf_gpu_ennaf_opts *o, int format, u64 p0, bool last_part, EnnafSplit &S, bool allow_direct = false)
{
    memset(&S, 0, sizeof S);
    int seq_type = o->seq_type, rc;
    S.format = format; S.seq_type = seq_type;
    S.fourbit = seq_type <= NAF_SEQ_RNA;
    S.store_mask = !(o->no_mask || !S.fourbit);                                                   // ennaf.c:445
    S.store_qual = format == NAF_FMT_FASTQ;                                                       // ennaf.c:477
    S.no_mask = o->no_mask != 0; (void)last_part;
    u8 *&bases = S.bases; u8 *&s_ids = S.s_ids, *&s_cmt = S.s_cmt, *&s_qual = S.s_qual;
    u64 &n_ids = S.n_ids, &n_cmt = S.n_cmt, &n_qual = S.n_qual, &T = S.T, &N = S.N, &longest = S.longest;
    u64 *&rec_begin = S.rec_begin, *&rec_end = S.rec_end;
    if (format == NAF_FMT_FASTQ) {
        EncP P; memset(&P, 0, sizeof P);
        P.text = d_text; P.n = n; P.p0 = p0;
        set_expected(P, seq_type, false);
        u64 tiles = n / ET_TILE + 1;
        i64 *t_eol = arena_new<i64>(c, tiles + 1), *t_sp = arena_new<i64>(c, tiles + 1);
        u64 *t_ls = arena_new<u64>(c, tiles + 2), *t_seq = arena_new<u64>(c, tiles + 2), *t_ids = arena_new<u64>(c, tiles + 2),
            *t_cmt = arena_new<u64>(c, tiles + 2), *t_qual = arena_new<u64>(c, tiles + 2);
        u64 *tot = arena_new<u64>(c, 8);
        u32 *piece_cnt = arena_new<u32>(c, tiles * 256);
        if (!t_eol || !t_sp || !t_ls || !t_seq || !t_ids || !t_cmt || !t_qual || !tot || !piece_cnt) return NAF_GPU_ENOMEM;
        // regular tiles by lines (k_encq_count_reg), the others -- the first and the last one always, whatever the tolerant parser has
        // something to tolerate in -- from a list by the general kernel; NAF_GPU_FQ_REG=0: every tile by the general kernel
        const bool fq_reg = !(ctx_opt(c, "FQ_REG") && ctx_opt(c, "FQ_REG")[0] == '0') && S.fourbit && n >= 16 * ET_TILE;
        // ... and their counts from the look that makes the tables (k_fq_first); NAF_GPU_FQ_FIRST=0: a second look (k_encq_count_reg), =check: both
        const char *ff = ctx_opt(c, "FQ_FIRST");
        const bool fq_first = fq_reg && !(ff && ff[0] == '0'), fq_check = fq_first && ff && !strcmp(ff, "check");
        // ... and split by a wavefront each where they have no more than 64 segments (k_fq_scatter_wave); NAF_GPU_FQ_WAVE=0: by k_encq_scatter_reg's workgroups
        const bool fq_wave = fq_first && !(ctx_opt(c, "FQ_WAVE") && ctx_opt(c, "FQ_WAVE")[0] == '0');
        u32 n_wide = 0;
        FqCand *cand = nullptr;
        if (fq_first) {
            cand = arena_new<FqCand>(c, tiles + 1); if (!cand) return NAF_GPU_ENOMEM;
            LAUNCH(c, "ennaf_fq_first", k_fq_first, tiles, 64, 0, P, t_eol, t_sp, t_ls, cand);
        } else
        LAUNCH(c, "ennaf_last", k_enc_last, tiles, 256, 0, P, t_eol, t_sp, t_ls);
        { i64 *const a2[2] = { t_eol, t_sp }; if ((rc = scan_inclusive_max_i64_multi(c, a2, 2, tiles))) return rc; }
        if ((rc = scan_exclusive_u64(c, t_ls, tiles, tot + 4))) return rc;
        u32 *t_reg = nullptr, *need_list = nullptr; u64 n_need = tiles;
        u32 *redo_list = nullptr, *n_redo = nullptr;
        if (fq_reg) {
            t_reg = arena_new<u32>(c, tiles + 1); need_list = arena_new<u32>(c, tiles + 1); redo_list = arena_new<u32>(c, tiles + 1); n_redo = arena_new<u32>(c, 2);
            u64 *t_need = arena_new<u64>(c, tiles + 2);
            if (!t_reg || !need_list || !t_need || !redo_list || !n_redo) return NAF_GPU_ENOMEM;
            HIP_TRY(c, hipMemsetAsync(n_redo, 0, 8, c->stream));
            if (fq_first) LAUNCH(c, "ennaf_fq_pick", k_fq_pick, cdiv(tiles, 256), 256, 0, P, (const FqCand *)cand, (const i64 *)t_eol, (const i64 *)t_sp, (const u64 *)t_ls, t_seq, t_ids, t_cmt, t_qual, t_reg, t_need, tiles, n_redo + 1, fq_wave ? (u32)FQW_MAXSEG : 0xFFFFFFFFu);
            else LAUNCH(c, "ennaf_fq_count_reg", k_encq_count_reg, tiles, 256, 0, P, (const i64 *)t_eol, (const i64 *)t_sp, (const u64 *)t_ls, t_seq, t_ids, t_cmt, t_qual, t_reg, t_need, tiles);
            if (fq_check) {
                // the second look beside the first: every tile's verdict and, for regular tiles, its four counts must agree
                u32 *r2 = arena_new<u32>(c, tiles + 1); u64 *n2 = arena_new<u64>(c, tiles + 2), *q2 = arena_new<u64>(c, 4 * (tiles + 2)), *dif = arena_new<u64>(c, 2);
                if (!r2 || !n2 || !q2 || !dif) return NAF_GPU_ENOMEM;
                HIP_TRY(c, hipMemsetAsync(dif, 0xFF, 8, c->stream));
                LAUNCH(c, "ennaf_fq_count_reg", k_encq_count_reg, tiles, 256, 0, P, (const i64 *)t_eol, (const i64 *)t_sp, (const u64 *)t_ls, q2, q2 + (tiles + 2), q2 + 2 * (tiles + 2), q2 + 3 * (tiles + 2), r2, n2, tiles);
                LAUNCH(c, "ennaf_fq_compare", k_fq_compare, cdiv(tiles, 256), 256, 0, (const u32 *)t_reg, (const u32 *)r2, (const u64 *)t_seq, (const u64 *)t_ids, (const u64 *)t_cmt, (const u64 *)t_qual, (const u64 *)q2, tiles, dif);
                u64 hd = 0;
                if ((rc = ctx_readback(c, &hd, dif, 8))) return rc;
                if (hd != ~0ull) return ctx_fail(c, NAF_GPU_EHIP, "FASTQ: the first look and the second disagree on tile %llu", (unsigned long long)hd);
            }
            if ((rc = scan_exclusive_u64(c, t_need, tiles, tot + 5))) return rc;
            LAUNCH(c, "ennaf_need_list", k_need_list, cdiv(tiles, 256), 256, 0, (const u32 *)nullptr, (const u64 *)t_need, tiles, need_list, (const u32 *)t_reg);
            if ((rc = ctx_readback2(c, &n_need, tot + 5, 8, &n_wide, n_redo + 1, 4))) return rc;
            if (ctx_tracing(c)) ctx_trace(c, "[fq reg] tiles %llu, not regular %llu\n", (unsigned long long)tiles, (unsigned long long)n_need);
        }
        if (n_need) LAUNCH(c, "ennaf_fq_count", k_encq_count, n_need, 256, 0, P, (const i64 *)t_eol, (const i64 *)t_sp, (const u64 *)t_ls, t_seq, t_ids, t_cmt, t_qual, piece_cnt, (const u32 *)need_list, 0);
        { u64 *const a4[4] = { t_seq, t_ids, t_cmt, t_qual }, *const t4[4] = { tot + 0, tot + 1, tot + 2, tot + 3 }; if ((rc = scan_exclusive_u64_multi(c, a4, 4, tiles, t4))) return rc; }
        u64 h[5]; u8 lastb = 0x0A;
        if (n) { if ((rc = ctx_readback2(c, h, tot, 40, &lastb, d_text + n - 1, 1))) return rc; }
        else if ((rc = ctx_readback(c, h, tot, 40))) return rc;
        T = h[0]; n_ids = h[1]; n_cmt = h[2]; n_qual = h[3];
        u64 nlines = h[4]; N = (nlines + 3) / 4;
        // truncated input (process.c:499,510,513,517,520)
        // (a header line at the very end without its line end: "no sequence data" -- but only if no record in front of it stops the reference first, so
        // the verdict waits for the records' own checks below: a quality line of the wrong length in record 1 was reported as this truncation)
        const bool noseq = nlines % 4 == 1 && !(lastb >= 0x0A && lastb <= 0x0D);
        if ((rc = alloc_bases(c, S))) return rc;
        s_ids = (u8 *)arena_alloc(c, n_ids + 16); s_cmt = (u8 *)arena_alloc(c, n_cmt + 16); s_qual = (u8 *)arena_alloc(c, n_qual + 16);
        rec_begin = arena_new<u64>(c, N + 1); rec_end = arena_new<u64>(c, N + 1);
        u64 *q_begin = arena_new<u64>(c, N + 1), *q_end = arena_new<u64>(c, N + 1);
        const size_t NU = 4 * 257 + 3;                                                             // histograms, first_error, longest, strict key
        u64 *d_unexp = arena_new<u64>(c, NU);
        if (!s_ids || !s_cmt || !s_qual || !rec_begin || !rec_end || !q_begin || !q_end || !d_unexp) return NAF_GPU_ENOMEM;
        HIP_TRY(c, hipMemsetAsync(d_unexp, 0, NU * 8, c->stream));
        HIP_TRY(c, hipMemsetAsync(d_unexp + 4 * 257, 0xFF, 8, c->stream));                        // first_error = none
        HIP_TRY(c, hipMemsetAsync(d_unexp + 4 * 257 + 2, 0xFF, 8, c->stream));                    // strict key = none
        HIP_TRY(c, hipMemsetAsync(rec_begin, 0, (N + 1) * 8, c->stream)); HIP_TRY(c, hipMemsetAsync(rec_end, 0, (N + 1) * 8, c->stream));
        HIP_TRY(c, hipMemsetAsync(q_begin, 0, (N + 1) * 8, c->stream)); HIP_TRY(c, hipMemsetAsync(q_end, 0, (N + 1) * 8, c->stream));
        FqOut O; O.seq = bases; O.packed = S.packed; O.casebits = (u32 *)S.casebits; O.ids = s_ids; O.cmt = s_cmt; O.qual = s_qual; O.rec_begin = rec_begin; O.rec_end = rec_end; O.q_begin = q_begin; O.q_end = q_end;
        O.unexpected = d_unexp; O.first_error = d_unexp + 4 * 257; O.strict_first = o->strict ? d_unexp + 4 * 257 + 2 : nullptr;
        O.t_seq = t_seq; O.t_ids = t_ids; O.t_cmt = t_cmt; O.t_qual = t_qual; O.t_ls = t_ls; O.piece_cnt = piece_cnt; O.list = need_list; O.t_reg = t_reg; O.redo_list = redo_list; O.n_redo = n_redo;
        if (S.fourbit) {
            if (T) LAUNCH(c, "ennaf_pack_edges", k_pack_edges_zero, cdiv(tiles, 256), 256, 0, (const u64 *)t_seq, tiles, T, S.packed, (u32 *)S.casebits, (const u8 *)nullptr, 0u);
            O.reg_kind = fq_wave ? 2u : 1u;
            if (t_reg && fq_wave) LAUNCH(c, "ennaf_fq_scatter_wave", k_fq_scatter_wave<true>, tiles, 64, 0, P, (const i64 *)t_eol, (const i64 *)t_sp, O);
            if (t_reg && (!fq_wave || n_wide)) LAUNCH(c, "ennaf_fq_scatter_reg", k_encq_scatter_reg<true>, tiles, 256, 0, P, (const i64 *)t_eol, (const i64 *)t_sp, O);
            if (n_need) LAUNCH(c, "ennaf_fq_scatter", k_encq_scatter<true>, n_need, 256, 0, P, (const i64 *)t_eol, (const i64 *)t_sp, O);
            if (t_reg) {
                // regular tiles handed back for their letters (an IUPAC code, a letter to be replaced): their pieces' counts, then the general scatter
                u32 nr = 0;
                if ((rc = ctx_readback(c, &nr, n_redo, 4))) return rc;
                if (ctx_tracing(c)) ctx_trace(c, "[fq reg] handed back %u\n", nr);
                if (nr) {
                    LAUNCH(c, "ennaf_fq_count", k_encq_count, nr, 256, 0, P, (const i64 *)t_eol, (const i64 *)t_sp, (const u64 *)t_ls, t_seq, t_ids, t_cmt, t_qual, piece_cnt, (const u32 *)redo_list, 1);
                    O.list = redo_list;
                    LAUNCH(c, "ennaf_fq_scatter", k_encq_scatter<true>, nr, 256, 0, P, (const i64 *)t_eol, (const i64 *)t_sp, O);
                }
            }
        } else LAUNCH(c, "ennaf_fq_scatter", k_encq_scatter<false>, tiles, 256, 0, P, (const i64 *)t_eol, (const i64 *)t_sp, O);
        if (nlines / 4) LAUNCH(c, "ennaf_fq_check", k_fq_check, cdiv(nlines / 4, 256), 256, 0, (const u64 *)rec_begin, (const u64 *)rec_end, (const u64 *)q_begin, (const u64 *)q_end, nlines / 4, O.first_error, d_unexp + 4 * 257 + 1);
        std::vector<u64> hu(NU);
        if ((rc = ctx_readback(c, hu.data(), d_unexp, NU * 8))) return rc;
        // The reference stops at the first problem in input order.  Structural ones come as (record, kind); an unexpected byte under
        // --strict as a text position, turned into (record, line of the record) by counting the line starts in front of it.  Inside
        // a record the order of detection is: '@' (first byte of line 0), id / comment bytes, bases, '+' (line 2), quality bytes,
        // quality length (end of line 3).
        u64 fe = hu[4 * 257];
        u64 trunc_rec = nlines % 4 ? nlines / 4 : ~0ull;                                           // the incomplete record
        u64 best_rec = ~0ull; int best_stage = 99, best_kind = SE_NONE;
        if (fe != ~0ull && (fe >> 2) <= trunc_rec) {
            int kind = (int)(fe & 3);
            best_rec = fe >> 2; best_stage = kind == FQ_E_AT ? 0 : kind == FQ_E_PLUS ? 3 : 5;
            best_kind = kind == FQ_E_AT ? SE_AT : kind == FQ_E_PLUS ? SE_PLUS : SE_QLEN;
        }
        const u64 sk = hu[4 * 257 + 2];
        if (o->strict && sk != ~0ull) {
            const u64 pos = sk >> 11; const int skind = (int)((sk >> 9) & 3);
            unsigned long long *d_cnt = arena_new<unsigned long long>(c, 1); if (!d_cnt) return NAF_GPU_ENOMEM;
            HIP_TRY(c, hipMemsetAsync(d_cnt, 0, 8, c->stream));
            LAUNCH(c, "ennaf_count_starts", k_count_starts, (u32)((pos - p0) / 65536 + 1 < 1024 ? (pos - p0) / 65536 + 1 : 1024), 256, 0, d_text, p0, pos, 1, d_cnt);
            u64 cnt = 0; if ((rc = ctx_readback(c, &cnt, d_cnt, 8))) return rc;
            const u64 ord = cnt ? cnt - 1 : 0, srec = ord >> 2; const int sstage = skind <= 1 ? 1 : skind == 2 ? 2 : 4;
            if (srec <= trunc_rec && (srec < best_rec || (srec == best_rec && sstage < best_stage))) {
                S.err_kind = SE_STRICT; S.err_char = (u32)(sk & 0x1FF); S.err_rec = srec + 1; S.err_a = (u64)skind;
                return 0;
            }
        }
        if (best_kind != SE_NONE) {
            S.err_kind = best_kind; S.err_rec = best_rec;
            if (best_kind == SE_QLEN) {
                u64 v[4]; const u64 r = best_rec;
                if ((rc = ctx_readback2(c, &v[0], rec_begin + r, 8, &v[1], rec_end + r, 8)) || (rc = ctx_readback2(c, &v[2], q_begin + r, 8, &v[3], q_end + r, 8))) return rc;
                S.err_a = v[3] - v[2]; S.err_b = v[1] - v[0];
            }
            return 0;
        }
        if (noseq) { S.err_kind = SE_NOSEQ; return 0; }
        if (nlines % 4) { S.err_kind = SE_NOQUAL; return 0; }
        for (int k = 0; k < 4; k++) for (int i = 0; i < 257; i++) S.unexpected[k][i] = hu[k * 257 + i];
        longest = hu[4 * 257 + 1];
        S.all_ends = 1;
    }
    if (format == NAF_FMT_FASTA) {
        EncP P; memset(&P, 0, sizeof P);
        P.text = d_text; P.n = n; P.p0 = p0;
        set_expected(P, seq_type, true);
        u64 tiles = n / ET_TILE + 1;                                                              // +1: the virtual end-of-input byte
        i64 *t_eol = arena_new<i64>(c, tiles + 1), *t_sp = arena_new<i64>(c, tiles + 1);
        u64 *t_seq = arena_new<u64>(c, tiles + 2), *t_ids = arena_new<u64>(c, tiles + 2), *t_cmt = arena_new<u64>(c, tiles + 2), *t_rec = arena_new<u64>(c, tiles + 2);
        u32 *t_tail = arena_new<u32>(c, tiles + 1), *t_reg = arena_new<u32>(c, tiles + 1), *irr_list = arena_new<u32>(c, tiles + 1);
        u64 *t_irr = arena_new<u64>(c, tiles + 2);
        u64 *tot = arena_new<u64>(c, 8);
        if (!t_eol || !t_sp || !t_seq || !t_ids || !t_cmt || !t_rec || !t_tail || !t_reg || !irr_list || !t_irr || !tot) return NAF_GPU_ENOMEM;
        // the case census rides on the count pass (tot[6]): an upper-case text needs no pass over its case bits to learn that its mask is one run
        HIP_TRY(c, hipMemsetAsync(tot + 6, 0, 8, c->stream));
        P.any_case = (u32 *)(tot + 6);
        // ONE pass over the text (k_enc_fused) where direct blocks are possible at all: the options that allow them are known here, the
        // stream's size and the text's case only behind the counts -- what the pass left in `loc` is then simply not used
        // (NAF_GPU_ONEPASS=0: the two passes, the cross-check; =2: direct blocks from two of them, like NAF_GPU_DIRECT=2)
        const char *e_ed = ctx_opt(c, "DIRECT"), *e_epf = ctx_opt(c, "PREFER_FLAT"), *e_ebl = ctx_opt(c, "BLOCK_LOG"), *e_epr = ctx_opt(c, "PROBE"), *e_elz = ctx_opt(c, "LZ"), *e_op = ctx_opt(c, "ONEPASS");
        const u32 e_prefer_flat = e_epf && e_epf[0] ? (u32)atoi(e_epf) : 16u;
        const bool direct_opts = allow_direct && S.fourbit && o->level <= 1 && !o->long_log && !(e_ed && e_ed[0] == '0') && e_prefer_flat >= 2 && !(e_ebl && atoi(e_ebl) != 15) && !(e_epr && e_epr[0] == '1')
                                 && !(e_elz && !strcmp(e_elz, "all")) && n >= 16 * ET_TILE;
        const bool fused = direct_opts && enc_wave_wg(c) && !(e_op && e_op[0] == '0') && (n >> 17) >= ((e_ed && e_ed[0] == '2') || (e_op && e_op[0] == '2') ? 2u : 256u);
        u8 *loc = nullptr, *t_hist = nullptr, *locc = nullptr, *t_lower = nullptr; u32 *t_needf0 = nullptr; u64 *t_need0 = nullptr;
        if (fused) {
            loc = (u8 *)arena_alloc(c, (tiles + 1) * LOC_TILE + 64); t_hist = (u8 *)arena_alloc(c, (tiles + 1) * 16);
            locc = (u8 *)arena_alloc(c, (tiles + 1) * (LOC_TILE / 2) + 64); t_lower = (u8 *)arena_alloc(c, tiles + ZENC_LOC_BND + 2);
            if (!locc || !t_lower) return NAF_GPU_ENOMEM;
            HIP_TRY(c, hipMemsetAsync(t_lower, 0, tiles + ZENC_LOC_BND + 2, c->stream));
            t_needf0 = arena_new<u32>(c, tiles + 1); t_need0 = arena_new<u64>(c, tiles + 2);
            if (!loc || !t_hist || !t_needf0 || !t_need0) return NAF_GPU_ENOMEM;
            // (two and four tiles per wavefront, their loads in flight together: 23.8 -> 28.3 / 25.9 ms per 100 GB)
            // (a bounded grid of wavefronts that walk the tiles with the next tile's bytes asked for in advance: 117 VGPRs, 2.40 -> 2.87 ms per 10 GB)
            LAUNCH(c, "ennaf_split_once", (k_enc_fused<true, 1>), (u32)tiles, 64, 0, P, t_eol, t_sp, t_seq, t_ids, t_cmt, t_rec, t_tail, t_reg, t_irr, t_needf0, t_need0, tiles, loc, t_hist, locc, t_lower);
        } else
        LAUNCH(c, "ennaf_last", k_enc_last_fa, cdiv(tiles, 4 * LAST_TPW), 256, 0, P, t_eol, t_sp, tiles);
        // running maxima across tiles (positions are non-negative i64; reuse the u64-add scan machinery via max on i64)
        { i64 *const a2[2] = { t_eol, t_sp }; if ((rc = scan_inclusive_max_i64_multi(c, a2, 2, tiles))) return rc; }
        if (S.fourbit && n >= 16 * ET_TILE) {
            // pure tiles (nearly all of a genome) a wavefront per tile; the others, from a list, by the general kernel
            u32 *t_needf = fused ? t_needf0 : arena_new<u32>(c, tiles + 1), *need_list = arena_new<u32>(c, tiles + 1); u64 *t_need = fused ? t_need0 : arena_new<u64>(c, tiles + 2);
            if (!t_needf || !need_list || !t_need) return NAF_GPU_ENOMEM;
            if (fused) LAUNCH(c, "ennaf_pure_check", k_pure_check, cdiv(tiles, 256), 256, 0, P, (const i64 *)t_eol, tiles, t_needf, t_need, t_reg, t_irr, (const u32 *)t_tail);
            else if (enc_wave_wg(c)) LAUNCH(c, "ennaf_count_pure", (k_enc_count_pure<1, 1>), (u32)tiles, 64, 0, P, (const i64 *)t_eol, t_seq, t_ids, t_cmt, t_rec, t_tail, t_reg, t_irr, t_needf, t_need, tiles);
            else LAUNCH(c, "ennaf_count_pure", (k_enc_count_pure<4, 1>), cdiv(tiles, 4), 256, 0, P, (const i64 *)t_eol, t_seq, t_ids, t_cmt, t_rec, t_tail, t_reg, t_irr, t_needf, t_need, tiles);
            if ((rc = scan_exclusive_u64(c, t_need, tiles, tot + 5))) return rc;
            LAUNCH(c, "ennaf_need_list", k_need_list, cdiv(tiles, 256), 256, 0, (const u32 *)t_needf, (const u64 *)t_need, tiles, need_list, (const u32 *)nullptr);
            u64 n_need = 0;
            if ((rc = ctx_readback(c, &n_need, tot + 5, 8))) return rc;
            if (ctx_tracing(c)) {
                std::vector<u32> hr(tiles); hipMemcpy(hr.data(), t_reg, tiles * 4, hipMemcpyDeviceToHost);
                u64 nz = 0; for (u64 i = 0; i < tiles; i++) nz += hr[i] != 0;
                ctx_trace(c, "[reg] tiles %llu need %llu regular %llu; reg[1..4] = %08x %08x %08x %08x\n", (unsigned long long)tiles, (unsigned long long)n_need, (unsigned long long)nz, hr[1], hr[2], hr[3], hr[4]);
            }
            if (n_need) LAUNCH(c, "ennaf_count", k_enc_count, n_need, 256, 0, P, (const i64 *)t_eol, (const i64 *)t_sp, t_seq, t_ids, t_cmt, t_rec, t_tail, t_reg, t_irr, (const u32 *)need_list);
        } else
        LAUNCH(c, "ennaf_count", k_enc_count, tiles, 256, 0, P, (const i64 *)t_eol, (const i64 *)t_sp, t_seq, t_ids, t_cmt, t_rec, t_tail, t_reg, t_irr, (const u32 *)nullptr);
        { u64 *const a5[5] = { t_seq, t_ids, t_cmt, t_rec, t_irr }, *const t5[5] = { tot + 0, tot + 1, tot + 2, tot + 3, tot + 4 }; if ((rc = scan_exclusive_u64_multi(c, a5, 5, tiles, t5))) return rc; }
        // t_seq[tiles] must hold the grand total for the "line began in an earlier tile" lookup
        HIP_TRY(c, hipMemcpyAsync(t_seq + tiles, tot + 0, 8, hipMemcpyDeviceToDevice, c->stream));
        u32 *blk_t0 = fused ? arena_new<u32>(c, (size_t)(n >> 16) + 2) : nullptr;            // (a block's 65536 bases are at least as many bytes of text)
        if (fused && !blk_t0) return NAF_GPU_ENOMEM;
        LAUNCH(c, "ennaf_irregular_list", k_irregular_list, cdiv(tiles, 256), 256, 0, (const u32 *)t_reg, (const u64 *)t_irr, tiles, irr_list, (const u64 *)t_seq, blk_t0);
        u64 h[7];
        if ((rc = ctx_readback(c, h, tot, 56))) return rc;
        T = h[0]; n_ids = h[1]; n_cmt = h[2]; N = h[3];
        S.no_case = (u32)h[6] == 0 && !(ctx_opt(c, "CASE_CENSUS") && ctx_opt(c, "CASE_CENSUS")[0] == '0');
        P.any_case = nullptr;
        const u64 n_irregular = h[4];
        // a whole input in which the count pass met no case bit: its mask is one run whatever the scatter pass would write -- no case bits
        // are made at all (an eighth of a byte per base less to write; a shard's finish reads them for the census of its cut)
        if ((rc = alloc_bases(c, S, !(allow_direct && S.no_case)))) return rc;
        s_ids = (u8 *)arena_alloc(c, n_ids + 16); s_cmt = (u8 *)arena_alloc(c, n_cmt + 16);
        rec_begin = arena_new<u64>(c, N + 1); rec_end = arena_new<u64>(c, N + 1);
        const size_t NU = 3 * 257 + 3;                                                             // histograms, longest, strict key, lead
        u64 *d_unexp = arena_new<u64>(c, NU);
        if (!s_ids || !s_cmt || !rec_begin || !rec_end || !d_unexp) return NAF_GPU_ENOMEM;
        HIP_TRY(c, hipMemsetAsync(d_unexp, 0, NU * 8, c->stream));
        HIP_TRY(c, hipMemsetAsync(d_unexp + 3 * 257 + 1, 0xFF, 8, c->stream));                    // strict key = none
        HIP_TRY(c, hipMemsetAsync(rec_begin, 0, (N + 1) * 8, c->stream));
        EncOut O; O.seq = bases; O.packed = S.packed; O.casebits = (u32 *)S.casebits; O.ids = s_ids; O.cmt = s_cmt; O.rec_begin = rec_begin; O.rec_end = rec_end;
        O.unexpected = d_unexp; O.longest = d_unexp + 3 * 257; O.strict_first = o->strict ? d_unexp + 3 * 257 + 1 : nullptr; O.lead = d_unexp + 3 * 257 + 2;
        O.t_seq = t_seq; O.t_ids = t_ids; O.t_cmt = t_cmt; O.t_rec = t_rec; O.t_tail = t_tail; O.tile_eol = t_eol; O.t_reg = t_reg; O.irr_list = S.fourbit ? irr_list : nullptr;
        O.direct = nullptr; O.nd = 0; O.loc_mode = 0; O.sparse_list = nullptr; O.n_sparse = nullptr;
        if (S.fourbit) {
            // Direct blocks: a whole input at level 1 (no match finder unless the look at the stream says so), blocks of exactly 32 KiB
            // (the stream's ragged end is coded as a part of its own, ennaf_streams), from 8 MiB of packed bases up
            const u64 n_seqb = (T + 1) / 2;
            const char *ed = e_ed, *epr = e_epr;
            const u32 prefer_flat = e_prefer_flat;
            const u64 nd64 = n_seqb >> 15;
            O.loc_mode = 0; O.sparse_list = nullptr; O.n_sparse = nullptr;
            if (direct_opts && nd64 >= ((ed && ed[0] == '2') || (fused && e_op && e_op[0] == '2') ? 2u : 256u) && nd64 < 0x7FFFFFFFull && !((T & 1) && (n_seqb & 32767) == 0)) {
                S.nd = (u32)nd64; S.direct = (u8 *)arena_alloc(c, S.nd); if (!S.direct) return NAF_GPU_ENOMEM;
                // the codes k_enc_fused left are the direct blocks'; a text with lower case has its tiles' case bits beside them (k_case_gather)
                const bool use_loc = fused && !(e_op && e_op[0] == '3' && !S.no_case);       // (NAF_GPU_ONEPASS=3: only texts without a case bit, round 6's first form)
                i32 *blk_bnd = use_loc ? arena_new<i32>(c, (size_t)S.nd * ZENC_LOC_BND + 1) : nullptr;
                if (use_loc && !blk_bnd) return NAF_GPU_ENOMEM;
                if (use_loc) LAUNCH(c, "ennaf_direct_blocks", k_direct_verdict, cdiv(S.nd, 256), 256, 0, (const u64 *)t_seq, (const u32 *)t_reg, (const u8 *)t_hist, tiles, S.nd, prefer_flat, (epr && epr[0] == '0') ? 0 : (int)zenc_probe_every(n_seqb), (const u32 *)blk_t0, S.direct, blk_bnd);
                else LAUNCH(c, "ennaf_direct_blocks", k_direct_blocks, cdiv(S.nd, 4), 256, 0, d_text, (const u64 *)t_seq, (const u32 *)t_reg, tiles, S.nd, prefer_flat, (epr && epr[0] == '0') ? 0 : (int)zenc_probe_every(n_seqb), S.direct, (const u32 *)blk_t0, blk_bnd);
                O.direct = S.direct; O.nd = S.nd;
                if (use_loc) {
                    u32 *sparse_list = arena_new<u32>(c, tiles + 1), *n_sparse = arena_new<u32>(c, 2);
                    if (!sparse_list || !n_sparse) return NAF_GPU_ENOMEM;
                    HIP_TRY(c, hipMemsetAsync(n_sparse, 0, 8, c->stream));
                    O.loc_mode = 1; O.sparse_list = sparse_list; O.n_sparse = n_sparse;
                    LAUNCH(c, "ennaf_sparse_list", k_sparse_list, cdiv(tiles, 256), 256, 0, P, (const i64 *)t_eol, O, tiles, sparse_list, n_sparse);
                    S.dloc.loc = loc; S.dloc.blk_bnd = blk_bnd; S.dloc.blk_t0 = blk_t0; S.dloc.tiles = tiles;
                    if (S.casebits) LAUNCH(c, "ennaf_case_gather", k_case_gather, S.nd, 256, 0, (const u8 *)S.direct, S.nd, (const u32 *)blk_t0, (const i32 *)blk_bnd, (const u8 *)locc, (const u8 *)t_lower, S.casebits);
                }
                if (ctx_tracing(c)) {
                    std::vector<u8> hd(S.nd); hipStreamSynchronize(c->stream); hipMemcpy(hd.data(), S.direct, S.nd, hipMemcpyDeviceToHost);
                    u64 k = 0; for (u8 v : hd) k += v;
                    ctx_trace(c, "[direct] %llu of %u blocks\n", (unsigned long long)k, S.nd);
                }
            }
            // (S.sc4 stays valid until the call ends -- the tables live in the arena: the same pass can run again without direct blocks
            // when the stream turns out to be worth matching; what it then adds to the counts of unexpected bytes has been read by then)
            S.sc4.P = P; S.sc4.O = O; S.sc4.t_eol = t_eol; S.sc4.t_sp = t_sp; S.sc4.t_seq = t_seq; S.sc4.tiles = tiles; S.sc4.T = T; S.sc4.n = n; S.sc4.n_irregular = n_irregular;
            if ((rc = ennaf_scatter4(c, S.sc4, S.packed, S.casebits))) return rc;
        } else LAUNCH(c, "ennaf_scatter", k_enc_scatter<false>, tiles, 256, 0, P, (const i64 *)t_eol, (const i64 *)t_sp, O);
        std::vector<u64> hu(NU);
        if ((rc = ctx_readback(c, hu.data(), d_unexp, NU * 8))) return rc;
        const u64 sk = hu[3 * 257 + 1];
        if (o->strict && sk != ~0ull) {                                                            // process.c:98-140: n_sequences + 1 = headers read so far
            const u64 pos = sk >> 11;
            unsigned long long *d_cnt = arena_new<unsigned long long>(c, 1); if (!d_cnt) return NAF_GPU_ENOMEM;
            HIP_TRY(c, hipMemsetAsync(d_cnt, 0, 8, c->stream));
            LAUNCH(c, "ennaf_count_starts", k_count_starts, (u32)((pos - p0) / 65536 + 1 < 1024 ? (pos - p0) / 65536 + 1 : 1024), 256, 0, d_text, p0, pos, 0, d_cnt);
            u64 cnt = 0; if ((rc = ctx_readback(c, &cnt, d_cnt, 8))) return rc;
            S.err_kind = SE_STRICT; S.err_char = (u32)(sk & 0x1FF); S.err_rec = cnt; S.err_a = (sk >> 9) & 3;
            return 0;
        }
        for (int k = 0; k < 3; k++) for (int i = 0; i < 257; i++) S.unexpected[k][i] = hu[k * 257 + i];
        longest = hu[3 * 257];
        S.lead = N ? hu[3 * 257 + 2] : T;
    }
    return 0;
}

// What the neighbours of a shard contribute to its streams (all zero for a whole input).
struct EnnafCarry {
    u64 tail_extra;        // bases of the following shards that still belong to this shard's last record (FASTA cut inside a record)
    u32 skip_first;        // 1: this shard's first base is the high nibble of the previous shard's last packed byte
    u32 tail_hi;           // code that completes this shard's last packed byte when its pack window is odd (0 at the end of the data)
    int prev_masked;       // case of the last base in front of the shard ("unmasked" in front of base 0, encoders.c:132)
    int skip_run0;         // 1: the bases in front of the shard's first case change continue a run that an earlier shard emits
    u64 run_ext;           // bases of the following shards that continue this shard's last mask run
};
struct EnnafStreams { const u8 *ptr[6]; u64 len[6], orig[6]; int lz[6], block_log[6], window_log[6]; bool present[6];
                      u32 tail[6]; int flags[6];
                      const u8 *direct; u32 nd, tail_packed; const ZencLoc *dloc; };   // direct blocks of the sequence stream (EnnafSplit::direct); tail[4] without them   // flags: ZENC_PREFER_RAW for the mask stream   // tail: bytes at the end of the stream that go into a Raw block of their own (encode_stream)

// E5-E7: lengths, 4-bit pack, mask units -> the six uncompressed streams.
// part: 1 = ids, names, sequence and quality (nothing to wait for), 2 = lengths and mask (a few read-backs), 3 = all of them.  The two
// parts may run on different contexts (naf_gpu_ennaf: part 2 on a side stream beside the planning of the sequence frame).
static int ennaf_streams(naf_gpu_ctx *c, EnnafSplit &S, const EnnafCarry &K, EnnafStreams &X, int part = 3)
{
    if (part & 1) memset(&X, 0, sizeof X);
    int rc; const u64 T = S.T, N = S.N;
    u32 *s_len = nullptr; u8 *s_seq = nullptr, *s_mask = nullptr;
    u64 n_lenb = 0, n_seqb = 0, n_mask = 0; int mask_block_log = 15; u32 seq_tail = 0;
    if (S.format != 0) {
        if (N && (part & 2)) {
            u64 *lu = arena_new<u64>(c, N + 2); if (!lu) return NAF_GPU_ENOMEM;
            const u64 total = T + K.tail_extra;                                                    // where the last record of a FASTA part ends
            LAUNCH(c, "ennaf_len_count", k_len_unit_count, cdiv(N, 256), 256, 0, (const u64 *)S.rec_begin, (const u64 *)S.rec_end, N, total, lu, S.all_ends);
            if ((rc = scan_exclusive_u64(c, lu, N, lu + N + 1))) return rc;
            u64 nu = 0; if ((rc = ctx_readback(c, &nu, lu + N + 1, 8))) return rc;
            s_len = arena_new<u32>(c, nu + 1); if (!s_len) return NAF_GPU_ENOMEM;
            LAUNCH(c, "ennaf_len_write", k_len_unit_write, cdiv(N, 256), 256, 0, (const u64 *)S.rec_begin, (const u64 *)S.rec_end, N, total, (const u64 *)lu, s_len, S.all_ends);
            n_lenb = nu * 4;
        }
        // sequence stream: flush_pack has left the codes of the shard's own base stream; a shard whose first base belongs to its
        // neighbour's last byte moves them down a nibble, and an odd window borrows its last high nibble (ennaf.c:525-529: 0 at the end)
        const u64 mt = (T + MBB_TILE - 1) / MBB_TILE;
        u64 *tc = S.tc;
        if (!(part & 1)) { }
        else if (S.fourbit) {
            const u64 Tp = T > K.skip_first ? T - K.skip_first : 0;                                // bases of the pack window
            n_seqb = (Tp + 1) / 2;
            if (K.skip_first && n_seqb) {
                s_seq = (u8 *)arena_alloc(c, n_seqb + 16); if (!s_seq) return NAF_GPU_ENOMEM;
                LAUNCH(c, "ennaf_nibble_shift", k_nibble_shift, cdiv(cdiv(n_seqb, 8), 256), 256, 0, (const u8 *)S.packed, (T + 1) / 2, s_seq, n_seqb);
            } else s_seq = S.packed;
            if ((Tp & 1) && K.tail_hi) LAUNCH(c, "ennaf_tail_nibble", k_set_high_nibble, 1, 64, 0, s_seq + n_seqb - 1, K.tail_hi);
            // The last byte of an odd stream holds the padding nibble: a byte value the rest of a block of A C G T pairs does not have.
            // Inside the last Huffman block it would be a seventeenth symbol, that block's tree would differ from every other one, and
            // the decoder could not read the frame in place (k_emit_tile_flat, zstd_dec.hip: flat_tail): it goes into a Raw block.
            // (Not for archives with qualities: the reference's FASTQ path holds the sequence frame in memory behind a 4-byte magic number
            // and stops feeding its decoder 4 bytes early (input.c:254-256, :356-357) -- a last block of 3 + 1 bytes would never arrive.)
            if ((Tp & 1) && !K.tail_hi && n_seqb >= 2 && !S.store_qual) seq_tail = 1;
        } else {
            if (S.no_mask && T) LAUNCH(c, "ennaf_toupper", k_toupper, cdiv(T, 256), 256, 0, S.bases, T);   // process.c:46-51
            s_seq = S.bases; n_seqb = T;
        }
        // mask (only ever stored next to a 4-bit sequence stream, ennaf.c:445); a shard has counted its boundaries in the census already
        if (S.store_mask && T && (part & 2)) {
            u64 nb = 0; i64 *tile_last = S.census ? S.census_last : nullptr;
            const bool one_run = !S.census && S.no_case && !K.skip_run0 && !K.prev_masked;         // (the count pass saw no case bit: no pass over the case bits, no scan, no read-back)
            if (!one_run) {
            if (!S.census) {
                tc = arena_new<u64>(c, mt + 2); tile_last = arena_new<i64>(c, mt + 1); if (!tc || !tile_last) return NAF_GPU_ENOMEM;
                LAUNCH(c, "ennaf_mask_count", k_maskb_count, mt, 256, 0, (const u64 *)S.casebits, T, tc, 0, tile_last);
                if ((rc = scan_inclusive_max_i64(c, tile_last, mt))) return rc;
            }
            const int b0 = S.census ? ((S.first_base >= 96) != (K.prev_masked != 0)) : 0;          // a shard's case change at its first base
            if (b0) LAUNCH(c, "ennaf_mask_b0", k_add_u64, 1, 64, 0, tc, (u64)1);
            if ((rc = scan_exclusive_u64(c, tc, mt, tc + mt + 1))) return rc;
            if ((rc = ctx_readback(c, &nb, tc + mt + 1, 8))) return rc;
            }
            u64 nu = 0;
            if (nb == 0 && !K.skip_run0) {
                // no case change at all (a text in one case): one run of T bases, its units known here -- 0xFF but for the last one
                const u64 len = T + K.run_ext;
                nu = len / 255 + 1;
                s_mask = (u8 *)arena_alloc(c, nu + 16); if (!s_mask) return NAF_GPU_ENOMEM;
                HIP_TRY(c, hipMemsetAsync(s_mask, 0xFF, nu, c->stream));
                SmallBytes lb; memset(&lb, 0, sizeof lb); lb.b[0] = (u8)(len % 255); lb.n = 1;
                LAUNCH(c, "ennaf_mask_units", k_put_bytes, 1, 64, 0, s_mask + (nu - 1), lb);
            } else {
            u64 *any_long = arena_new<u64>(c, 1), *bnd = nullptr;
            if (!any_long) return NAF_GPU_ENOMEM;
            u64 longs = 1;
            const char *ms = ctx_opt(c, "MASK_SHORT");
            // many case changes close to each other, all of them (it is assumed) less than 255 bases apart: the units straight from the case bits
            // (NAF_GPU_MASK_SHORT=1: by way of the list of positions, the cross-check; =0: never assumed)
            if (tile_last && nb >= (1u << 16) && nb * 64 >= T && !(ms && (ms[0] == '0' || ms[0] == '1'))) {   // (runs of 64 bases on average: longer ones are likely to hold one of 255)
                nu = nb + 1 - (K.skip_run0 ? 1 : 0);
                s_mask = (u8 *)arena_alloc(c, nu + 16); if (!s_mask) return NAF_GPU_ENOMEM;
                HIP_TRY(c, hipMemsetAsync(any_long, 0, 8, c->stream));
                LAUNCH(c, "ennaf_mask_units", k_maskb_units_direct, mt, 256, 0, (const u64 *)S.casebits, T, (const u64 *)tc, (const i64 *)tile_last, nb, s_mask, K.run_ext, K.skip_run0, K.prev_masked, any_long);
                if ((rc = ctx_readback(c, &longs, any_long, 8))) return rc;
            }
            if (longs) {
            bnd = arena_new<u64>(c, nb + 1); if (!bnd) return NAF_GPU_ENOMEM;
            if (nb) LAUNCH(c, "ennaf_mask_bscatter", k_maskb_scatter, mt, 256, 0, (const u64 *)S.casebits, T, (const u64 *)tc, nb, bnd, K.prev_masked);
            // runs of fewer than 255 bases only: a unit per run, in run order, written on that assumption (NAF_GPU_MASK_SHORT=0: never assumed)
            { 
              if (!(ms && ms[0] == '0') && (nb + 1) * 255 > T + K.run_ext) {          // (fewer runs than that: one of them holds 255 bases)
                  nu = nb + 1 - (K.skip_run0 ? 1 : 0);
                  s_mask = (u8 *)arena_alloc(c, nu + 16); if (!s_mask) return NAF_GPU_ENOMEM;
                  HIP_TRY(c, hipMemsetAsync(any_long, 0, 8, c->stream));
                  LAUNCH(c, "ennaf_mask_units", k_mask_units_short, cdiv(nb + 1, 256), 256, 0, (const u64 *)bnd, nb, T, s_mask, K.run_ext, K.skip_run0, any_long);
                  if ((rc = ctx_readback(c, &longs, any_long, 8))) return rc;
              } }
            }
            if (longs) {
                u64 *ru = arena_new<u64>(c, nb + 3); if (!ru) return NAF_GPU_ENOMEM;
                LAUNCH(c, "ennaf_mask_runs", k_mask_run_units, cdiv(nb + 1, 256), 256, 0, (const u64 *)bnd, nb, T, ru, K.run_ext, K.skip_run0);
                if ((rc = scan_exclusive_u64(c, ru, nb + 1, ru + nb + 2))) return rc;
                if ((rc = ctx_readback(c, &nu, ru + nb + 2, 8))) return rc;
                s_mask = (u8 *)arena_alloc(c, nu + 16); if (!s_mask) return NAF_GPU_ENOMEM;
                if (nu) HIP_TRY(c, hipMemsetAsync(s_mask, 0xFF, nu, c->stream));
                LAUNCH(c, "ennaf_mask_units", k_mask_units_write, cdiv(nb + 1, 256), 256, 0, (const u64 *)bnd, nb, T, (const u64 *)ru, s_mask, K.run_ext, K.skip_run0);
            }
            }
            // Block size of the mask stream.  Real soft-masking (runs of a few hundred bases) gives a few MB of high-entropy units, i.e. a
            // few hundred blocks whose Huffman streams the decoder walks serially: 8 KiB blocks (2 KiB streams) cut that latency to a
            // quarter.  Very long runs give strings of 255s (constant blocks) and one last block with the remainder in it -- whose four
            // streams are one lane's work each on either side: small blocks there too.
            mask_block_log = 13;
            // (a mask of tens of MB -- reads whose case changes every few bases -- has blocks enough for every lane of the decoder at the
            // sequence stream's 32 KiB, and a quarter of the blocks to plan here: NAF_GPU_MASK_BLOCK_LOG sets it)
            { const char *mb = ctx_opt(c, "MASK_BLOCK_LOG"); if (mb && atoi(mb) >= 10 && atoi(mb) <= 17) mask_block_log = atoi(mb); else if (nu >= (64ull << 20)) mask_block_log = 15; }
            n_mask = nu;
        }
    }
    // ids, names and lengths are text-like / repetitive and small: always through the LZ stage (as reference level 1 does);
    // mask, sequence and quality get it from level 2 up
    if (part & 1) {
        X.ptr[0] = S.s_ids; X.len[0] = X.orig[0] = S.n_ids; X.lz[0] = 1; X.present[0] = true;
        X.ptr[1] = S.s_cmt; X.len[1] = X.orig[1] = S.n_cmt; X.lz[1] = 1; X.present[1] = true;
        // Comments and lengths of many records -- `len=150` and the number 150, 38 M times -- in blocks of 32 KiB instead of the match finder's 8:
        // coded 8 KiB at a time they were 56 K blocks of one match each, and this build's own decoder spent 3 of a 12.5 GB FASTQ's 22.9 ms on
        // their per-block chores (index, tables, a workgroup of the LDS executor per block); as 14 K blocks they go through the dataflow
        // executor, whose runs that continue from block to block cost next to nothing: unnaf 22.9 -> 19.9 ms.
        // NAF_GPU_SIDE_BLOCK_LOG=10..15: these two streams' block size whatever their length (13: as before).
        { const char *sb_ = ctx_opt(c, "NAMES_BLOCK_LOG"); const int v = sb_ ? atoi(sb_) : 0; if (v >= 10 && v <= 15) X.block_log[1] = v; else if (S.n_cmt >= (16u << 20)) X.block_log[1] = 15; }
        // direct blocks are blocks of exactly 32 KiB: the stream's ragged end (with the padding nibble, if any) is a part of its own
        X.tail_packed = seq_tail; X.direct = nullptr; X.nd = 0;
        X.dloc = nullptr;
        if (S.direct && (part & 1)) { X.direct = S.direct; X.nd = S.nd; seq_tail = (u32)(n_seqb & 32767); X.dloc = S.dloc.loc ? &S.dloc : nullptr; }
        X.ptr[4] = s_seq; X.len[4] = n_seqb; X.orig[4] = T; X.present[4] = true; X.tail[4] = seq_tail;   // ennaf.c:582: number of bases
        X.ptr[5] = S.s_qual; X.len[5] = X.orig[5] = S.n_qual; X.present[5] = S.store_qual;
    }
    if (part & 2) {
        X.ptr[2] = (const u8 *)s_len; X.len[2] = X.orig[2] = n_lenb; X.lz[2] = 1; X.present[2] = true;
        { const char *sb_ = ctx_opt(c, "SIDE_BLOCK_LOG"); const int v = sb_ ? atoi(sb_) : 0; if (v >= 10 && v <= 15) X.block_log[2] = v; else if (n_lenb >= (16u << 20)) X.block_log[2] = 15; }
        X.ptr[3] = s_mask; X.len[3] = X.orig[3] = n_mask; X.block_log[3] = mask_block_log; X.present[3] = S.store_mask; X.flags[3] = ZENC_PREFER_RAW;   // (no frame tree for the mask: measured -- its 8 KiB blocks' own trees are 4 % smaller and mostly of 7 bits, the frame's code of 9, which the decoder walks with its two-level look-up)
    }
    return 0;
}

// Match windows of the six streams.  Level 1 (the default): ids / names / lengths find matches inside a block, the other streams are
// entropy-coded only.  From level 2 every stream finds matches across blocks inside the window libzstd uses at that level for large
// inputs (clevels.h); --long N (ennaf.c:247-273, :505) gives the SEQUENCE stream a window of 2^N at any level -- the reference turns
// on libzstd's long-distance matcher for that stream only (compressor.c:12-16).
// Level 1 without --long looks at the sequence stream first (zenc_repeat_probe): when a thirty-second of the probed anchors has an
// earlier copy nearby, the stream is matched inside the level's own window of 2^19 -- as the reference's level 1 would.
// defer != nullptr: a needed look at the sequence stream is left to the caller (*defer = true), who runs ennaf_probe_verdict later
static void ennaf_probe_verdict(EnnafStreams &X, u32 share) { if (share >= 32) { X.window_log[4] = 19; X.lz[4] = 1; } }
static int ennaf_windows(naf_gpu_ctx *c, EnnafStreams &X, const naf_gpu_ennaf_opts *o, bool *defer = nullptr)
{
    const int wl = zenc_level_window(o->level);
    for (int i = 0; i < 6; i++) X.window_log[i] = wl;
    // level 1 (the default, "fast"): the sequence stream's blocks of sixteen pair codes stay at four bits per code unless Huffman
    // coding saves a sixteenth (zstd_enc.hip: ZENC_PREFER_FLAT) -- the decoder then reads them in place
    if (o->level <= 1) X.flags[4] |= ZENC_PREFER_FLAT;
    // ... and the sequence and quality streams keep their codes to 7 bits (zstd_enc.hip: ZENC_SHORT_CODES): a packed base pair next to an N
    // is rare enough for an 11-bit code, which costs this build's decoder its one-level table and makes every symbol of the block a
    // two-level look-up -- the decode of a FASTQ's sequence stream took as long as that of its quality stream, twice the size
    if (o->level <= 1) { X.flags[4] |= ZENC_SHORT_CODES; X.flags[5] |= ZENC_SHORT_CODES; }
    // ... and one tree for the frame where it fits (zstd_enc.hip: ZENC_FRAME_TREE): the quality stream, a sequence stream's blocks that are not flat
    if (o->level <= 1) { X.flags[4] |= ZENC_FRAME_TREE; X.flags[5] |= ZENC_FRAME_TREE; }
    if (o->long_log) { X.window_log[4] = o->long_log < 10 ? 10 : o->long_log > 31 ? 31 : o->long_log; X.lz[4] = 1; }
    else if (!wl && X.present[4] && X.len[4]) {
        const char *pe = ctx_opt(c, "PROBE");                 // "0": never look, "1": always match
        u32 share = 0;
        if (pe && pe[0] == '1') share = 1024;
        else if (!(pe && pe[0] == '0') && defer) { *defer = true; return 0; }
        else if (!(pe && pe[0] == '0')) {
            int rc = zenc_repeat_probe(c, X.ptr[4], X.len[4], &share); if (rc) return rc;
        }
        ennaf_probe_verdict(X, share);
    }
    return 0;
}

// container header (ennaf.c:538-556): magic, version, flags, separator, line length, N, title
static size_t naf_header_bytes(const naf_gpu_ennaf_opts *o, bool store_mask, bool store_qual, u64 longest, u64 N, u8 *hd /* >= 40 */)
{
    size_t hl = 0;
    hd[hl++] = 0x01; hd[hl++] = 0xF9; hd[hl++] = 0xEC;
    if (o->seq_type == NAF_SEQ_DNA) hd[hl++] = 1; else { hd[hl++] = 2; hd[hl++] = (u8)o->seq_type; }
    hd[hl++] = (u8)(((o->title ? 1 : 0) << 6) | (1 << 5) | (1 << 4) | (1 << 3) | ((store_mask ? 1 : 0) << 2) | (1 << 1) | (store_qual ? 1 : 0));
    hd[hl++] = ' ';
    hl += vle(o->line_length >= 0 ? (u64)o->line_length : longest, hd + hl);
    hl += vle(N, hd + hl);
    if (o->title) hl += vle(strlen(o->title), hd + hl);
    return hl;
}

// One stream -> its frame, or its part of a frame (flags: ZENC_PART*; 0 = a whole frame without the magic number).  `tail` bytes at
// the end are coded as a part of their own, i.e. end up in a Raw block behind the others.  With `place` the frame goes where the hook
// says once its size is known (zstd_encode): the tail part is coded first, into scratch, so that the hook hears the size of the whole.
struct PlaceTail { const ZencPlace *outer; size_t tail_len; u8 *at; };
static u8 *place_before_tail(void *ud, size_t len) { PlaceTail *t = (PlaceTail *)ud; return t->at = t->outer->fn(t->outer->ud, len + t->tail_len); }
// the two halves of a placed stream (zstd_encode_begin / _finish): what the caller queues between them runs beside the planning
struct StreamJob { ZencJob *main; const u8 *d_stream; u64 len; int level, flags; u32 tail; bool sized; u8 *tail_tmp; size_t tail_len; int tail_flags; };   // tail_flags: what the tail part is coded with beside its PART flags (a shard's ragged block: like the blocks in front of it)   // sized: encode_stream_size ran (the tail part is coded, the main part's size is known)
static int encode_stream_begin(naf_gpu_ctx *c, const u8 *d_stream, u64 len, int level, int flags, int lz, int block_log, int window_log, u32 tail, StreamJob *J, const u8 *direct = nullptr, u32 nd = 0, const ZencLoc *dloc = nullptr)
{
    J->main = nullptr; J->d_stream = d_stream; J->len = len; J->level = level; J->flags = flags; J->tail = (tail && len > tail) ? tail : 0;
    J->sized = false; J->tail_tmp = nullptr; J->tail_len = 0; J->tail_flags = 0;
    int f1 = flags;
    if (J->tail) f1 = ZENC_PART | ((!(flags & ZENC_PART) || (flags & ZENC_PART_FIRST)) ? ZENC_PART_FIRST : 0) | (flags & (ZENC_PREFER_RAW | ZENC_PREFER_FLAT | ZENC_SHORT_CODES | ZENC_FRAME_TREE));
    if (direct && (lz || len - J->tail != (u64)nd << 15)) return ctx_fail(c, NAF_GPU_EARG, "direct blocks need a stream of whole blocks and no match finder");
    int rc = zstd_encode_begin(c, d_stream, len - J->tail, level, f1, lz, block_log, window_log, &J->main, direct, nd, dloc);
    if (rc) { zstd_encode_drop(J->main); J->main = nullptr; }
    return rc;
}
// The size of the stream's frame before anything of it is placed: the tail part is coded (into scratch, as encode_stream_finish would),
// the main part's size read back.  encode_stream_finish then writes without a read-back of its own.
static int encode_stream_size(naf_gpu_ctx *c, StreamJob *J, size_t *clen)
{
    if (J->tail && !J->sized) {
        const int f2 = ZENC_PART | ((!(J->flags & ZENC_PART) || (J->flags & ZENC_PART_LAST)) ? ZENC_PART_LAST : 0) | J->tail_flags;
        const size_t tb = naf_gpu_zstd_compress_bound(J->tail);
        u8 *tmp = (u8 *)arena_alloc(c, tb); if (!tmp) return NAF_GPU_ENOMEM;
        size_t b = 0;
        int rc = zstd_encode(c, J->d_stream + (J->len - J->tail), J->tail, J->level, tmp, tb, &b, f2, 0, 0, 0); if (rc) return rc;
        J->tail_tmp = tmp; J->tail_len = b;
    }
    size_t a = 0;
    int rc = zstd_encode_size(c, J->main, &a); if (rc) return rc;
    J->sized = true;
    *clen = a + J->tail_len;
    return 0;
}
static int encode_stream_finish(naf_gpu_ctx *c, StreamJob *J, size_t *clen, const ZencPlace *place)
{
    ZencJob *mj = J->main; J->main = nullptr;
    if (!J->tail) return zstd_encode_finish(c, mj, nullptr, 0, clen, place);
    const int f2 = ZENC_PART | ((!(J->flags & ZENC_PART) || (J->flags & ZENC_PART_LAST)) ? ZENC_PART_LAST : 0) | J->tail_flags;
    size_t a = 0, b = J->tail_len;
    u8 *tmp = J->tail_tmp;
    int rc = 0;
    if (!J->sized) {
        const size_t tb = naf_gpu_zstd_compress_bound(J->tail);
        tmp = (u8 *)arena_alloc(c, tb); if (!tmp) { zstd_encode_drop(mj); return NAF_GPU_ENOMEM; }
        rc = zstd_encode(c, J->d_stream + (J->len - J->tail), J->tail, J->level, tmp, tb, &b, f2, 0, 0, 0);
        if (rc) { zstd_encode_drop(mj); return rc; }
    }
    PlaceTail T = { place, b, nullptr }; ZencPlace P = { place_before_tail, &T };
    if ((rc = zstd_encode_finish(c, mj, nullptr, 0, &a, &P))) return rc;
    HIP_TRY(c, hipMemcpyAsync(T.at + a, tmp, b, hipMemcpyDeviceToDevice, c->stream));
    *clen = a + b;
    return 0;
}
static int encode_stream(naf_gpu_ctx *c, const u8 *d_stream, u64 len, int level, u8 *dst, size_t cap, size_t *clen, int flags, int lz, int block_log, int window_log, u32 tail, const ZencPlace *place = nullptr)
{
    if (place) {
        StreamJob J; int rc = encode_stream_begin(c, d_stream, len, level, flags, lz, block_log, window_log, tail, &J); if (rc) return rc;
        return encode_stream_finish(c, &J, clen, place);
    }
    if (!tail || len <= tail) return zstd_encode(c, d_stream, len, level, dst, cap, clen, flags, lz, block_log, window_log);
    const bool part = (flags & ZENC_PART) != 0;
    const int f1 = ZENC_PART | ((!part || (flags & ZENC_PART_FIRST)) ? ZENC_PART_FIRST : 0) | (flags & (ZENC_PREFER_RAW | ZENC_PREFER_FLAT | ZENC_SHORT_CODES | ZENC_FRAME_TREE));
    const int f2 = ZENC_PART | ((!part || (flags & ZENC_PART_LAST)) ? ZENC_PART_LAST : 0) | (tail >= 2048 ? (flags & (ZENC_PREFER_FLAT | ZENC_SHORT_CODES)) : 0);   // (a ragged block of some size is coded like the blocks in front of it)
    size_t a = 0, b = 0;
    int rc = zstd_encode(c, d_stream, len - tail, level, dst, cap, &a, f1, lz, block_log, window_log); if (rc) return rc;
    rc = zstd_encode(c, d_stream + (len - tail), tail, level, dst + a, cap - a, &b, f2, 0, 0, 0); if (rc) return rc;
    *clen = a + b;
    return 0;
}

__global__ void k_collect4(const u64 *a, const u64 *b, const u64 *c4, const u64 *d, u64 *out) { if (threadIdx.x < 4) { const u64 *p = threadIdx.x == 0 ? a : threadIdx.x == 1 ? b : threadIdx.x == 2 ? c4 : d; out[threadIdx.x] = p ? *p : 0; } }
struct SecOut { u64 orig, comp; };

// a section = VLE(original size) VLE(compressed size) frame (ennaf.c:538-589): the frame is written behind its header at once
struct SecPlace { naf_gpu_ctx *c; u8 *d_naf; size_t cap, pos; u64 orig; size_t hl; int rc; };
static u8 *place_section(void *ud, size_t clen)
{
    SecPlace *p = (SecPlace *)ud; naf_gpu_ctx *c = p->c;
    SmallBytes h; memset(&h, 0, sizeof h);
    size_t hl = vle(p->orig, h.b); hl += vle(clen, h.b + hl); h.n = (u32)hl;
    if (p->pos + hl + clen > p->cap) { p->rc = ctx_fail(c, NAF_GPU_ECAP, "ennaf output capacity %zu too small", p->cap); return nullptr; }
    hipLaunchKernelGGL(k_put_bytes, dim3(1), dim3(64), 0, c->stream, p->d_naf + p->pos, h);
    if (hipGetLastError() != hipSuccess) { p->rc = ctx_fail(c, NAF_GPU_EHIP, "section header launch failed"); return nullptr; }
    p->hl = hl;
    return p->d_naf + p->pos + hl;
}

// `early`: the stream's planning was queued before (encode_stream_begin); otherwise both halves run here.  The launches go to c's
// stream (c may be a side context), a failure's text lands in `report`.
static int put_section(naf_gpu_ctx *c, naf_gpu_ctx *report, const u8 *d_stream, u64 stream_len, u64 orig, int level, u8 *d_naf, size_t cap, size_t &pos, SecOut &so, int lz, int block_log, int window_log, u32 tail, StreamJob *early, int flags = 0, const u8 *direct = nullptr, u32 nd = 0, const ZencLoc *dloc = nullptr)
{
    SecPlace sp = { c, d_naf, cap, pos, orig, 0, 0 }; ZencPlace P = { place_section, &sp };
    size_t clen = 0;
    StreamJob J;
    int rc = 0;
    if (!early) { rc = encode_stream_begin(c, d_stream, stream_len, level, flags, lz, block_log, window_log, tail, &J, direct, nd, dloc); early = &J; }
    if (!rc) rc = encode_stream_finish(c, early, &clen, &P);
    if (rc) { if (report != c) ctx_fail(report, sp.rc ? sp.rc : rc, "%s", c->err); return sp.rc ? sp.rc : rc; }
    pos += sp.hl + clen;
    so.orig = orig; so.comp = clen;
    return 0;
}

static int ennaf_whole(naf_gpu_ctx *c, const void *d_text_, size_t n, const naf_gpu_ennaf_opts *o,
                       void *d_naf_, size_t cap, size_t *naf_len, naf_gpu_ennaf_report *rep);
extern "C" int naf_gpu_ennaf(naf_gpu_ctx *c, const void *d_text_, size_t n, const naf_gpu_ennaf_opts *o,
                             void *d_naf_, size_t cap, size_t *naf_len, naf_gpu_ennaf_report *rep)
{
    if (!c || !o || !d_naf_ || !naf_len || (!d_text_ && n)) return NAF_GPU_EARG;
    int rc = ennaf_whole(c, d_text_, n, o, d_naf_, cap, naf_len, rep);
    arena_settle(c);                                            // a call that grew an arena leaves it as one allocation (naf_gpu.hip)
    return rc;
}
static int ennaf_whole(naf_gpu_ctx *c, const void *d_text_, size_t n, const naf_gpu_ennaf_opts *o,
                       void *d_naf_, size_t cap, size_t *naf_len, naf_gpu_ennaf_report *rep)
{
    arena_reset(c);
    { int rs = ctx_sides_ready(c); if (rs) return rs; }
    const u8 *d_text = (const u8 *)d_text_; u8 *d_naf = (u8 *)d_naf_;
    naf_gpu_ennaf_report R; memset(&R, 0, sizeof R);
    if (o->seq_type < 0 || o->seq_type > 3) return ctx_fail(c, NAF_GPU_EARG, "bad seq_type");
    int format = 0, rc; u64 p0 = 0;
    if ((rc = ennaf_sniff(c, d_text, n, o->format, &format, &p0))) return rc;
    R.format = format;
    EnnafSplit S;
    if ((rc = ennaf_split(c, d_text, n, o, format, p0, true, S, true))) return rc;
    if (S.err_kind) return split_error_text(c, o->seq_type, S.err_kind, S.err_char, S.err_rec, S.err_a, S.err_b, 0);
    EnnafCarry K; memset(&K, 0, sizeof K);
    // The sequence and quality streams are planned on this context's stream while a side context makes the length and mask units and
    // codes ids, names, lengths and mask (dozens of small launches and a dozen read-backs) on its own; the sections still land in
    // file order, each behind the one before.
    naf_gpu_ctx *sc = c->side;
    const char *eo = ctx_opt(c, "ENC_OVERLAP");
    const bool force_overlap = eo && !strcmp(eo, "2");             // tests: the concurrent path on inputs of any size
    const bool overlap = sc && !(eo && !strcmp(eo, "0")) && (force_overlap || S.T >= (32u << 20) || S.n_qual >= (16u << 20));
    EnnafStreams X;
    if ((rc = ennaf_streams(c, S, K, X, overlap ? 1 : 3))) return rc;
    bool probe_later = false;                                     // level 1: the look at the sequence stream runs beside its planning
    if ((rc = ennaf_windows(c, X, o, overlap ? &probe_later : nullptr))) return rc;
    // the match finder wants packed bytes in every block: the bases are packed again, without direct blocks
    auto undirect = [&]() -> int {
        if (!X.direct) return 0;
        X.direct = nullptr; X.nd = 0; X.tail[4] = X.tail_packed; S.direct = nullptr; S.nd = 0; X.dloc = nullptr; S.dloc.loc = nullptr;
        S.sc4.O.direct = nullptr; S.sc4.O.nd = 0; S.sc4.O.loc_mode = 0;
        return ennaf_scatter4(c, S.sc4, S.packed, S.casebits);
    };
    if (X.lz[4] && (rc = undirect())) return rc;
    // container (ennaf.c:538-589)
    u8 hd[64]; size_t hl = naf_header_bytes(o, S.store_mask, S.store_qual, S.longest, S.N, hd);
    size_t tl = o->title ? strlen(o->title) : 0;
    size_t pos = 0;
    if (hl + tl > cap) return ctx_fail(c, NAF_GPU_ECAP, "ennaf output capacity %zu too small", cap);
    { SmallBytes hb; memset(&hb, 0, sizeof hb); memcpy(hb.b, hd, hl); hb.n = (u32)hl; LAUNCH(c, "ennaf_header", k_put_bytes, 1, 64, 0, d_naf, hb); pos += hl; }
    if (tl) { HIP_TRY(c, hipMemcpyAsync(d_naf + pos, o->title, tl, hipMemcpyHostToDevice, c->stream)); pos += tl; HIP_TRY(c, hipStreamSynchronize(c->stream)); }
    StreamJob big[6]; bool early[6] = { false, false, false, false, false, false };
    naf_gpu_ctx *sb = nullptr; int rcB = 0;                       // second side context (lengths, mask) and what its thread returns
    naf_gpu_ctx *pc = nullptr; int rcP = 0; u32 probe_share = 0; bool probe_out = false;   // third side context: the look at the sequence stream, on a thread of its own
    // (an error below leaves through here: the side contexts' threads work on this frame's variables until they are joined)
    auto bail = [&](int r, naf_gpu_ctx *from) -> int {
        if (sb) ctx_worker_join(sb);
        if (probe_out) { ctx_worker_join(pc); probe_out = false; hipStreamSynchronize(pc->stream); }
        for (int k = 0; k < 6; k++) if (early[k]) { zstd_encode_drop(big[k].main); early[k] = false; }
        hipStreamSynchronize(sc->stream); if (sb) hipStreamSynchronize(sb->stream);
        return from && from != c ? ctx_fail(c, r, "%s", from->err) : r;
    };
    if (overlap) {
        arena_reset(sc);
        HIP_TRY(c, hipEventRecord(c->fork_ev, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(sc->stream, c->fork_ev, 0));
        // The side chains first: with millions of reads they are what the call ends up waiting for (ids through the match finder, the
        // mask units), and queued behind the two big streams' planning they started a millisecond later than they could.
        // Lengths and mask: their units and the first half of their frames on a second side context with a host thread of its own,
        // beside ids and names on the first (many reads make each of the two chains several milliseconds long)
        // (a few records and no mask to speak of: one chain is short enough, and a thread hand-over is not free; the mask of a long text is
        // a pass over its case bits and a stream of MBs to code, a third of a millisecond and more that ids and names need not wait behind)
        sb = (S.N >= 65536 || force_overlap || (S.store_mask && S.T >= (256u << 20))) ? c->side2 : nullptr;
        if (sb) {
            arena_reset(sb);
            HIP_TRY(c, hipStreamWaitEvent(sb->stream, c->fork_ev, 0));
            ctx_worker_start(sb, [&] {
                rcB = ennaf_streams(sb, S, K, X, 2);
                for (int i = 2; i < 4 && !rcB; i++)
                    if (X.present[i]) { rcB = encode_stream_begin(sb, X.ptr[i], X.len[i], o->level, X.flags[i], X.lz[i], X.block_log[i], X.window_log[i], X.tail[i], &big[i]); early[i] = !rcB; }
            });
        } else if ((rc = ennaf_streams(sc, S, K, X, 2))) return bail(rc, sc);
        // The look at the sequence stream has a stream AND a thread of its own (the third side context): its two launches and its read-back
        // are then in flight while this thread queues the four streams' forty launches -- on the first side stream it stood 0.2 ms in front of
        // ids and names, and queued behind them it answered 0.3 ms after they were done (a 10 GB text's call ends on these chains)
        if (probe_later && c->side3) {
            pc = c->side3;
            arena_reset(pc);
            if (hipStreamWaitEvent(pc->stream, c->fork_ev, 0) != hipSuccess) return bail(ctx_fail(c, NAF_GPU_EHIP, "ennaf: the look's stream"), c);
            probe_out = true;
            ctx_worker_start(pc, [&] { rcP = zenc_repeat_probe(pc, X.ptr[4], X.len[4], &probe_share); });
        }
        for (int i = 4; i < 6; i++)
            if (X.present[i]) {
                if ((rc = encode_stream_begin(c, X.ptr[i], X.len[i], o->level, X.flags[i], X.lz[i], X.block_log[i], X.window_log[i], X.tail[i], &big[i], i == 4 ? X.direct : nullptr, i == 4 ? X.nd : 0u, i == 4 ? X.dloc : nullptr))) return bail(rc, c);
                early[i] = true;
            }
        for (int i = 0; i < 2; i++)
            if (X.present[i]) {
                if ((rc = encode_stream_begin(sc, X.ptr[i], X.len[i], o->level, X.flags[i], X.lz[i], X.block_log[i], X.window_log[i], X.tail[i], &big[i]))) return bail(rc, sc);
                early[i] = true;
            }
        if (probe_later) {
            // the frame is planned as if there were nothing to match (what the look says of nearly every input); a repeat-rich
            // stream drops that plan and starts over with the match finder
            if (probe_out) { ctx_worker_join(pc); probe_out = false; if (rcP) return bail(rcP, pc); }
            else if ((rc = zenc_repeat_probe(sc, X.ptr[4], X.len[4], &probe_share))) return bail(rc, sc);
            ennaf_probe_verdict(X, probe_share);
            if (X.lz[4]) {
                zstd_encode_drop(big[4].main); early[4] = false;
                if ((rc = undirect())) return bail(rc, c);
                if ((rc = encode_stream_begin(c, X.ptr[4], X.len[4], o->level, X.flags[4], X.lz[4], X.block_log[4], X.window_log[4], X.tail[4], &big[4]))) return bail(rc, c);
                early[4] = true;
            }
        }
    }
    for (int i = 0; i < 257; i++) { R.unexpected_id[i] = S.unexpected[0][i]; R.unexpected_comment[i] = S.unexpected[1][i]; R.unexpected_seq[i] = S.unexpected[2][i]; R.unexpected_qual[i] = S.unexpected[3][i]; }
    R.n_sequences = S.N; R.n_bases = S.T; R.longest_line = S.longest;

    SecOut so[6]; memset(so, 0, sizeof so);
    bool joined = !overlap, joinedB = sb == nullptr;
    auto join = [&]() -> int {                                     // everything behind this point is on c's stream again
        if (joined) return 0;
        joined = true;
        if (hipEventRecord(c->split_ev[0], sc->stream) != hipSuccess || hipStreamWaitEvent(c->stream, c->split_ev[0], 0) != hipSuccess) return ctx_fail(c, NAF_GPU_EHIP, "ennaf: joining the side stream failed");
        if (sb && (hipEventRecord(c->split_ev[1], sb->stream) != hipSuccess || hipStreamWaitEvent(c->stream, c->split_ev[1], 0) != hipSuccess)) return ctx_fail(c, NAF_GPU_EHIP, "ennaf: joining the side stream failed");
        return 0;
    };
    auto joinB = [&]() -> int {                                    // the thread of the second side context has queued its part
        if (joinedB) return 0;
        joinedB = true;
        ctx_worker_join(sb);
        return rcB ? ctx_fail(c, rcB, "%s", sb->err) : 0;
    };
    const char *sf = ctx_opt(c, "SIZES_FIRST");
    if (overlap && !(sf && sf[0] == '0')) {
        // The sections' SIZES first, then their bytes, side by side: a section's place is the sum of the sizes in front of it, and learning
        // those by writing the sections one after the other made the sequence stream's gather (0.9 ms of a 10 GB text's 4.5) wait for four small
        // frames' read-backs and writes, and a FASTQ's two big writes for its mask's and its names'.  Every begun job's size is one read-back
        // on its own stream (the tail parts are coded on the way); then the main stream's sections are queued first, the others beside them.
        size_t csz[6] = { 0, 0, 0, 0, 0, 0 }, at[6] = { 0, 0, 0, 0, 0, 0 };
        static const int order[6] = { 4, 5, 2, 3, 0, 1 };
        auto ctx_of = [&](int i) -> naf_gpu_ctx * { return i >= 4 ? c : (sb && i >= 2) ? sb : sc; };
        for (int k = 0; k < 6 && !rc; k++) {
            const int i = order[k];
            if (i == 2) {
                if ((rc = joinB())) break;
                // the four side sections' totals with ONE read-back (on the first side stream, behind the second's work) where they are begun jobs
                // without tail parts -- four read-backs one after the other were 0.1 ms in front of the main stream's writes
                bool batch = true; const u64 *tp[4] = { nullptr, nullptr, nullptr, nullptr };
                for (int q = 0; q < 4; q++) if (X.present[q]) { if (!early[q] || big[q].tail) batch = false; else tp[q] = zstd_encode_total_ptr(big[q].main); }
                if (batch && (tp[0] || tp[1] || tp[2] || tp[3])) {
                    u64 *d4 = arena_new<u64>(sc, 4); u64 h4[4] = { 0, 0, 0, 0 };
                    if (!d4) { rc = NAF_GPU_ENOMEM; break; }
                    if (sb && (hipEventRecord(c->split_ev[1], sb->stream) != hipSuccess || hipStreamWaitEvent(sc->stream, c->split_ev[1], 0) != hipSuccess)) { rc = ctx_fail(c, NAF_GPU_EHIP, "ennaf: the side streams' sizes"); break; }
                    hipLaunchKernelGGL(k_collect4, dim3(1), dim3(64), 0, sc->stream, tp[0], tp[1], tp[2], tp[3], d4);
                    if ((rc = ctx_readback(sc, h4, d4, 32))) { ctx_fail(c, rc, "%s", sc->err); break; }
                    for (int q = 0; q < 4; q++) if (tp[q]) zstd_encode_set_total(big[q].main, h4[q]);
                }
            }
            if (!X.present[i]) continue;
            naf_gpu_ctx *w = ctx_of(i);
            if (!early[i]) {
                if ((rc = encode_stream_begin(w, X.ptr[i], X.len[i], o->level, X.flags[i], X.lz[i], X.block_log[i], X.window_log[i], X.tail[i], &big[i], i == 4 ? X.direct : nullptr, i == 4 ? X.nd : 0u, i == 4 ? X.dloc : nullptr))) { if (w != c) ctx_fail(c, rc, "%s", w->err); break; }
                early[i] = true;
            }
            if ((rc = encode_stream_size(w, &big[i], &csz[i])) && w != c) ctx_fail(c, rc, "%s", w->err);
        }
        if (!rc) {
            size_t p = pos;
            for (int i = 0; i < 6; i++) if (X.present[i]) { u8 tmp[24]; at[i] = p; p += vle(X.orig[i], tmp) + vle(csz[i], tmp) + csz[i]; }
            for (int k = 0; k < 6 && !rc; k++) {
                const int i = order[k];
                if (!X.present[i]) continue;
                size_t pp = at[i];
                rc = put_section(ctx_of(i), c, X.ptr[i], X.len[i], X.orig[i], o->level, d_naf, cap, pp, so[i], X.lz[i], X.block_log[i], X.window_log[i], X.tail[i], &big[i], X.flags[i]);
                early[i] = false;
            }
            pos = p;
        }
    } else
    for (int i = 0; i < 6 && !rc; i++) {
        if (i >= 2 && (rc = joinB())) break;
        if (!X.present[i]) continue;
        if (i >= 4 && (rc = join())) break;
        naf_gpu_ctx *w = !overlap || i >= 4 ? c : (sb && i >= 2) ? sb : sc;
        rc = put_section(w, c, X.ptr[i], X.len[i], X.orig[i], o->level, d_naf, cap, pos, so[i], X.lz[i], X.block_log[i], X.window_log[i], X.tail[i], early[i] ? &big[i] : nullptr, X.flags[i], i == 4 ? X.direct : nullptr, i == 4 ? X.nd : 0u, i == 4 ? X.dloc : nullptr);
        early[i] = false;
    }
    if (!joinedB) { ctx_worker_join(sb); joinedB = true; }
    for (int k = 0; k < 6; k++) if (early[k]) { zstd_encode_drop(big[k].main); early[k] = false; }
    if (!rc) rc = join(); else if (overlap) { hipStreamSynchronize(sc->stream); if (sb) hipStreamSynchronize(sb->stream); }
    if (rc) return rc;
    for (int i = 0; i < 6; i++) { R.section_orig[i] = so[i].orig; R.section_comp[i] = so[i].comp; }
    *naf_len = pos;
    if (rep) *rep = R;
    return 0;
}

// ======================= one input on several GPUs (include/naf_gpu.h: "ennaf of ONE input on several GPUs") =========================
struct EnnafShardState { EnnafSplit S; const u8 *text; u64 n; u32 shard, n_shards; bool begun; };

extern "C" int naf_gpu_ennaf_sniff(naf_gpu_ctx *c, const void *d_text, size_t n, int want_format, int *format, uint64_t *p0)
{
    if (!c || !format || !p0 || (!d_text && n)) return NAF_GPU_EARG;
    arena_reset(c);
    u64 p = 0; int rc = ennaf_sniff(c, (const u8 *)d_text, n, want_format, format, &p);
    *p0 = p;
    return rc;
}

static int line_census(naf_gpu_ctx *c, const u8 *t, u64 n, int prev_is_eol, u64 **tile_pre, u64 *tiles_out, u64 *total)
{
    const u64 tiles = (n + LC_TILE - 1) / LC_TILE;
    u64 *cnt = arena_new<u64>(c, tiles + 2); if (!cnt) return NAF_GPU_ENOMEM;
    if (tiles) LAUNCH(c, "ennaf_line_count", k_line_count, tiles, 256, 0, t, n, prev_is_eol, cnt);
    int rc = scan_exclusive_u64(c, cnt, tiles, cnt + tiles + 1); if (rc) return rc;
    if ((rc = ctx_readback(c, total, cnt + tiles + 1, 8))) return rc;
    *tile_pre = cnt; *tiles_out = tiles;
    return 0;
}

extern "C" int naf_gpu_ennaf_count_lines(naf_gpu_ctx *c, const void *d_slice, size_t len, int prev_is_eol, uint64_t *n_line_starts)
{
    if (!c || !n_line_starts || (!d_slice && len)) return NAF_GPU_EARG;
    arena_reset(c);
    u64 *pre, tiles, total = 0;
    int rc = line_census(c, (const u8 *)d_slice, len, prev_is_eol, &pre, &tiles, &total);
    *n_line_starts = total;
    return rc;
}

extern "C" int naf_gpu_ennaf_find_cut(naf_gpu_ctx *c, const void *d_slice, size_t len, int format, int prev_is_eol, uint64_t skip_lines, uint64_t *offset)
{
    if (!c || !offset || (!d_slice && len) || (format != NAF_FMT_FASTA && format != NAF_FMT_FASTQ)) return NAF_GPU_EARG;
    arena_reset(c);
    const u8 *t = (const u8 *)d_slice; int rc;
    u64 *d_out = arena_new<u64>(c, 1); if (!d_out) return NAF_GPU_ENOMEM;
    u64 v = len;
    HIP_TRY(c, hipMemcpyAsync(d_out, &v, 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));                                                   // v is a stack variable
    if (format == NAF_FMT_FASTA) {
        // the answer is normally within the first line: look at the first 64 KiB, then at everything
        const u64 first = len < 65536 ? len : 65536;
        LAUNCH(c, "ennaf_eol_find", k_eol_find, 1, 256, 0, t, first, prev_is_eol, (unsigned long long *)d_out);
        if ((rc = ctx_readback(c, &v, d_out, 8))) return rc;
        if (v >= first && first < len) {
            v = len;
            HIP_TRY(c, hipMemcpyAsync(d_out, &v, 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            LAUNCH(c, "ennaf_eol_find", k_eol_find, 1024, 256, 0, t, (u64)len, prev_is_eol, (unsigned long long *)d_out);
            if ((rc = ctx_readback(c, &v, d_out, 8))) return rc;
        }
        *offset = v > len ? len : v;
        return 0;
    }
    u64 *pre, tiles, total = 0;
    if ((rc = line_census(c, t, len, prev_is_eol, &pre, &tiles, &total))) return rc;
    if (skip_lines < total) LAUNCH(c, "ennaf_line_find", k_line_find, 1, 256, 0, t, (u64)len, prev_is_eol, (const u64 *)pre, tiles, total, (u64)skip_lines, d_out);
    if ((rc = ctx_readback(c, &v, d_out, 8))) return rc;
    *offset = v;
    return 0;
}

extern "C" size_t naf_gpu_ennaf_shard_bound(size_t n) { return n + n / 128 + (1 << 16); }

extern "C" int naf_gpu_ennaf_shard_begin(naf_gpu_ctx *c, const void *d_slice, size_t n, const naf_gpu_ennaf_opts *o, int format,
                                         uint32_t shard, uint32_t n_shards, naf_gpu_shard_info *info)
{
    if (!c || !o || !info || (!d_slice && n) || n_shards == 0 || n_shards > NAF_GPU_MAX_SHARDS || shard >= n_shards) return NAF_GPU_EARG;
    if (format != NAF_FMT_FASTA && format != NAF_FMT_FASTQ) return ctx_fail(c, NAF_GPU_EARG, "shard_begin needs the sniffed format");
    if (o->seq_type < 0 || o->seq_type > 3) return ctx_fail(c, NAF_GPU_EARG, "bad seq_type");
    arena_reset(c);
    if (!c->shard_state) c->shard_state = new EnnafShardState();
    EnnafShardState *st = (EnnafShardState *)c->shard_state;
    st->begun = false; st->text = (const u8 *)d_slice; st->n = n; st->shard = shard; st->n_shards = n_shards;
    EnnafSplit &S = st->S;
    int rc = ennaf_split(c, st->text, n, o, format, 0, shard + 1 == n_shards, S); if (rc) return rc;
    memset(info, 0, sizeof *info);
    info->shard = shard; info->n_shards = n_shards; info->format = format; info->seq_type = o->seq_type; info->text_len = n;
    info->store_mask = S.store_mask; info->store_quality = S.store_qual;
    info->err_kind = S.err_kind; info->err_char = S.err_char; info->err_record = S.err_rec; info->err_a = S.err_a; info->err_b = S.err_b;
    st->begun = true;
    if (S.err_kind) return 0;                                                                      // reported by every shard's finish
    info->n_sequences = S.N; info->n_bases = S.T; info->longest_line = S.longest; info->lead_bases = S.lead;
    info->n_ids = S.n_ids; info->n_comments = S.n_cmt; info->n_quality = S.n_qual;
    memcpy(info->unexpected, S.unexpected, sizeof S.unexpected);
    S.census = true; S.mask_first = ~0ull;
    if (S.T) {
        if (S.store_mask) {
            const u64 mt = (S.T + MBB_TILE - 1) / MBB_TILE;
            S.tc = arena_new<u64>(c, mt + 2); unsigned long long *d_o = arena_new<unsigned long long>(c, 3);
            S.census_last = arena_new<i64>(c, mt + 1);
            if (!S.tc || !d_o || !S.census_last) return NAF_GPU_ENOMEM;
            u64 init[3] = { ~0ull, 0, 0 };
            HIP_TRY(c, hipMemcpyAsync(d_o, init, 24, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            LAUNCH(c, "ennaf_mask_census", k_maskb_census, mt, 256, 0, (const u64 *)S.casebits, S.T, S.tc, d_o, S.census_last);
            if ((rc = scan_inclusive_max_i64(c, S.census_last, mt))) return rc;
            LAUNCH(c, "ennaf_ends", k_packed_ends, 1, 64, 0, (const u8 *)S.packed, (const u64 *)S.casebits, S.T, d_o + 2);
            u64 *d_tot = arena_new<u64>(c, 1); if (!d_tot) return NAF_GPU_ENOMEM;
            u64 got[3]; i64 lastp1 = 0;
            if ((rc = ctx_readback2(c, got, d_o, 24, &lastp1, S.census_last + (mt - 1), 8))) return rc;
            S.mask_first = got[0]; S.mask_last = lastp1 > 0 ? (u64)(lastp1 - 1) : 0; S.first_base = (u8)got[2]; S.last_base = (u8)(got[2] >> 8);
            // number of changes: the per-tile counts are summed by the scan of the finish; here a reduction of the same array
            u64 *tmp = arena_new<u64>(c, mt + 2); if (!tmp) return NAF_GPU_ENOMEM;
            HIP_TRY(c, hipMemcpyAsync(tmp, S.tc, mt * 8, hipMemcpyDeviceToDevice, c->stream));
            if ((rc = scan_exclusive_u64(c, tmp, mt, d_tot))) return rc;
            if ((rc = ctx_readback(c, &S.mask_changes, d_tot, 8))) return rc;
        } else if (S.fourbit) {
            unsigned long long *d_o = arena_new<unsigned long long>(c, 1); if (!d_o) return NAF_GPU_ENOMEM;
            LAUNCH(c, "ennaf_ends", k_packed_ends, 1, 64, 0, (const u8 *)S.packed, (const u64 *)nullptr, S.T, d_o);
            u64 got = 0; if ((rc = ctx_readback(c, &got, d_o, 8))) return rc;
            S.first_base = (u8)got; S.last_base = (u8)(got >> 8);
        } else {
            if ((rc = ctx_readback2(c, &S.first_base, S.bases, 1, &S.last_base, S.bases + S.T - 1, 1))) return rc;
        }
    }
    info->mask_changes = S.mask_changes; info->mask_first_change = S.mask_first; info->mask_last_change = S.mask_last;
    info->first_base = S.first_base; info->last_base = S.last_base;
    return 0;
}

// What shard k inherits from and owes to its neighbours, from everybody's info.
struct ShardView { EnnafCarry K; u64 rec0; bool first[6], last[6]; };
static bool shard_mask_b0(const naf_gpu_shard_info *in, uint32_t j)                               // case change at the first base of shard j
{
    bool prev = false;                                                                             // "unmasked" in front of base 0
    for (uint32_t i = 0; i < j; i++) if (in[i].n_bases) prev = in[i].last_base >= 96;
    return in[j].n_bases && ((in[j].first_base >= 96) != prev);
}
static void shard_view(const naf_gpu_shard_info *in, uint32_t n, uint32_t k, ShardView &V)
{
    memset(&V, 0, sizeof V);
    const bool fourbit = in[k].seq_type <= NAF_SEQ_RNA;
    u64 S_k = 0; bool any_before = false, prev_masked = false;
    for (uint32_t j = 0; j < k; j++) { S_k += in[j].n_bases; V.rec0 += in[j].n_sequences; if (in[j].n_bases) { any_before = true; prev_masked = in[j].last_base >= 96; } }
    EnnafCarry &K = V.K;
    // 4-bit parity (encoders.c:30-69)
    auto skip_of = [&](uint32_t j, u64 Sj) -> u32 { return (fourbit && (Sj & 1) && in[j].n_bases) ? 1u : 0u; };
    K.skip_first = skip_of(k, S_k);
    if (fourbit && in[k].n_bases && ((in[k].n_bases - K.skip_first) & 1))
        for (uint32_t j = k + 1; j < n; j++) if (in[j].n_bases) { K.tail_hi = nuc4_host(in[j].first_base); break; }
    // the record a FASTA cut falls into (process.c:424)
    if (in[k].n_sequences)
        for (uint32_t j = k + 1; j < n; j++) { if (in[j].n_sequences) { K.tail_extra += in[j].lead_bases; break; } K.tail_extra += in[j].n_bases; }
    // soft-mask runs (encoders.c:126-146)
    K.prev_masked = prev_masked; K.skip_run0 = any_before;
    if (in[k].n_bases)
        for (uint32_t j = k + 1; j < n; j++) {
            if (!in[j].n_bases) continue;
            if (shard_mask_b0(in, j)) break;
            if (in[j].mask_changes) { K.run_ext += in[j].mask_first_change; break; }
            K.run_ext += in[j].n_bases;
        }
    // which parts exist: the frame header goes in front of shard 0's part, the last block flag on the last part that has bytes
    auto has = [&](uint32_t j, int s) -> bool {
        switch (s) {
        case 0: return in[j].n_ids != 0;
        case 1: return in[j].n_comments != 0;
        case 2: return in[j].n_sequences != 0;
        case 3: { if (!in[j].store_mask || !in[j].n_bases) return false;
                  bool before = false; for (uint32_t i = 0; i < j; i++) if (in[i].n_bases) before = true;
                  return !before || shard_mask_b0(in, j) || in[j].mask_changes != 0; }
        case 4: { u64 Sj = 0; for (uint32_t i = 0; i < j; i++) Sj += in[i].n_bases; return in[j].n_bases > skip_of(j, Sj); }
        default: return in[j].n_quality != 0;
        }
    };
    for (int s = 0; s < 6; s++) {
        int lastj = -1;
        for (uint32_t j = 0; j < n; j++) if (has(j, s)) lastj = (int)j;
        V.first[s] = k == 0;
        V.last[s] = lastj < 0 ? (k + 1 == n) : ((int)k == lastj);
    }
}

static int shards_error(naf_gpu_ctx *c, const naf_gpu_shard_info *in, uint32_t n)
{
    u64 rec0 = 0;
    for (uint32_t j = 0; j < n; j++) {
        if (in[j].err_kind) return split_error_text(c, in[j].seq_type, in[j].err_kind, in[j].err_char, in[j].err_record, in[j].err_a, in[j].err_b, rec0);
        rec0 += in[j].n_sequences;
    }
    return 0;
}

extern "C" int naf_gpu_ennaf_shard_finish(naf_gpu_ctx *c, const naf_gpu_ennaf_opts *o, const naf_gpu_shard_info *infos, void *d_pieces_, size_t cap, naf_gpu_shard_pieces *pieces)
{
    if (!c || !o || !infos || !pieces || !d_pieces_) return NAF_GPU_EARG;
    EnnafShardState *st = (EnnafShardState *)c->shard_state;
    if (!st || !st->begun) return ctx_fail(c, NAF_GPU_EARG, "shard_finish without shard_begin");
    st->begun = false;
    const uint32_t n = st->n_shards, k = st->shard;
    for (uint32_t j = 0; j < n; j++) if (infos[j].shard != j || infos[j].n_shards != n) return ctx_fail(c, NAF_GPU_EARG, "shard infos out of order");
    int rc = shards_error(c, infos, n); if (rc) return rc;
    ShardView V; shard_view(infos, n, k, V);
    EnnafStreams X;
    if ((rc = ennaf_streams(c, st->S, V.K, X))) return rc;
    if ((rc = ennaf_windows(c, X, o))) return rc;
    memset(pieces, 0, sizeof *pieces);
    u8 *dst = (u8 *)d_pieces_; size_t pos = 0;
    // Every section has a place of its own in the piece buffer -- its bound, not its size, is what it is given -- so no section waits for
    // the one in front: ids and comments are queued on the first side context, lengths and mask on the second, sequence and quality here,
    // and the sizes are read back when all of them are queued (the sections one after the other, each with a read-back behind its write,
    // were 52 ms of a 12.5 GB FASTQ shard whose kernels are 33).  NAF_GPU_SHARD_OVERLAP=0: one after the other on this context.
    auto part_tail = [&](int i) -> u32 {
        // a part of the packed sequence stream is whole blocks of 32 KiB and ONE ragged block behind them (not an even split of two sizes): the
        // stitched frame is then runs of equal blocks with a short block at every seam, which the decoder's stride index takes in place
        // (zstd_dec.hip: k_runs_*); a single part is a uniform frame outright
        u32 tail = V.last[i] ? X.tail[i] : 0u;
        if (i == 4 && st->S.fourbit && !X.lz[4] && !X.block_log[4] && X.len[4] >= 65536 && !(ctx_opt(c, "BLOCK_LOG") && atoi(ctx_opt(c, "BLOCK_LOG")) != 15)) { const u32 rag = (u32)(X.len[4] & 32767); if (rag > tail) tail = rag; }
        return tail;
    };
    size_t at[6] = { 0, 0, 0, 0, 0, 0 }, bound_sum = 0;
    for (int i = 0; i < 6; i++) if (X.present[i]) { at[i] = bound_sum; bound_sum += (naf_gpu_zstd_compress_bound(X.len[i]) + 64 + 15) & ~(size_t)15; }
    const char *so_ = ctx_opt(c, "SHARD_OVERLAP");
    u64 big_bytes = 0; for (int i = 0; i < 6; i++) if (X.present[i]) big_bytes += X.len[i];
    const bool overlap = !(so_ && so_[0] == '0') && bound_sum <= cap && (big_bytes >= (64u << 20) || (so_ && so_[0] == '1')) && ctx_sides_ready(c) == 0 && c->side && c->side2;   // (=1: whatever the size)
    if (overlap) {
        naf_gpu_ctx *sc = c->side, *sb = c->side2;
        arena_reset(sc); arena_reset(sb);
        HIP_TRY(c, hipEventRecord(c->fork_ev, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(sc->stream, c->fork_ev, 0));
        HIP_TRY(c, hipStreamWaitEvent(sb->stream, c->fork_ev, 0));
        StreamJob job[6]; bool begun[6] = { false, false, false, false, false, false };
        auto ctx_of = [&](int i) -> naf_gpu_ctx * { return i >= 4 ? c : i >= 2 ? sb : sc; };
        static const int order[6] = { 0, 1, 2, 3, 4, 5 };
        for (int q = 0; q < 6 && !rc; q++) {
            const int i = order[q];
            if (!X.present[i]) continue;
            naf_gpu_ctx *w = ctx_of(i);
            const int flags = ZENC_PART | (V.first[i] ? ZENC_PART_FIRST : 0) | (V.last[i] ? ZENC_PART_LAST : 0) | X.flags[i];
            const u32 tail = part_tail(i);
            if ((rc = encode_stream_begin(w, X.ptr[i], X.len[i], o->level, flags, X.lz[i], X.block_log[i], X.window_log[i], tail, &job[i]))) { if (w != c) ctx_fail(c, rc, "%s", w->err); break; }
            job[i].tail_flags = tail >= 2048 ? (flags & (ZENC_PREFER_FLAT | ZENC_SHORT_CODES)) : 0;    // (a ragged block of some size is coded like the blocks in front of it: encode_stream)
            begun[i] = true;
        }
        struct Fixed { u8 *at; size_t len; };
        auto fixed_place = [](void *ud, size_t len) -> u8 * { Fixed *f = (Fixed *)ud; f->len = len; return f->at; };
        static const int fin[6] = { 4, 5, 2, 3, 0, 1 };
        for (int q = 0; q < 6 && !rc; q++) {
            const int i = fin[q];
            if (!X.present[i] || !begun[i]) continue;
            naf_gpu_ctx *w = ctx_of(i);
            Fixed F = { dst + at[i], 0 }; ZencPlace P = { fixed_place, &F };
            size_t clen = 0;
            begun[i] = false;
            if ((rc = encode_stream_finish(w, &job[i], &clen, &P))) { if (w != c) ctx_fail(c, rc, "%s", w->err); break; }
            pieces->off[i] = at[i]; pieces->len[i] = clen; pieces->raw[i] = X.orig[i];
        }
        for (int i = 0; i < 6; i++) if (begun[i]) zstd_encode_drop(job[i].main);
        hipStreamSynchronize(sc->stream); hipStreamSynchronize(sb->stream);
        if (rc) { hipStreamSynchronize(c->stream); return rc; }
        pos = bound_sum;
    } else
    for (int i = 0; i < 6; i++) {
        if (!X.present[i]) continue;
        const size_t need = naf_gpu_zstd_compress_bound(X.len[i]);
        if (pos + need > cap) return ctx_fail(c, NAF_GPU_ECAP, "shard piece buffer of %zu bytes is too small", cap);
        size_t clen = 0;
        const int flags = ZENC_PART | (V.first[i] ? ZENC_PART_FIRST : 0) | (V.last[i] ? ZENC_PART_LAST : 0) | X.flags[i];
        const u32 tail = part_tail(i);
        if ((rc = encode_stream(c, X.ptr[i], X.len[i], o->level, dst + pos, cap - pos, &clen, flags, X.lz[i], X.block_log[i], X.window_log[i], tail))) return rc;
        pieces->off[i] = pos; pieces->len[i] = clen; pieces->raw[i] = X.orig[i];
        pos += (clen + 15) & ~(size_t)15;
    }
    pieces->total = pos;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    arena_settle(c);                                            // the shard's split state is no longer needed: a grown arena becomes one allocation
    return 0;
}

extern "C" int naf_gpu_ennaf_stitch_plan(const naf_gpu_ennaf_opts *o, const naf_gpu_shard_info *in, const naf_gpu_shard_pieces *pc, uint32_t n,
                                         naf_gpu_stitch_seg *segs, size_t seg_cap, size_t *n_segs, uint8_t *lit, size_t lit_cap, size_t *lit_len,
                                         uint64_t *naf_len, naf_gpu_ennaf_report *rep)
{
    if (!o || !in || !pc || !segs || !n_segs || !lit || !lit_len || !naf_len || n == 0 || n > NAF_GPU_MAX_SHARDS) return NAF_GPU_EARG;
    const size_t tl = o->title ? strlen(o->title) : 0;
    if (seg_cap < 7 + 6 * (size_t)n || lit_cap < 256 + tl) return NAF_GPU_ECAP;
    naf_gpu_ennaf_report R; memset(&R, 0, sizeof R);
    R.format = in[0].format;
    for (uint32_t j = 0; j < n; j++) {
        if (in[j].err_kind) return NAF_GPU_EINPUT;
        R.n_sequences += in[j].n_sequences; R.n_bases += in[j].n_bases;
        if (in[j].longest_line > R.longest_line) R.longest_line = in[j].longest_line;
        for (int i = 0; i < 257; i++) { R.unexpected_id[i] += in[j].unexpected[0][i]; R.unexpected_comment[i] += in[j].unexpected[1][i];
                                        R.unexpected_seq[i] += in[j].unexpected[2][i]; R.unexpected_qual[i] += in[j].unexpected[3][i]; }
    }
    const bool present[6] = { true, true, true, in[0].store_mask != 0, true, in[0].store_quality != 0 };
    size_t ns = 0, ll = 0; u64 pos = 0;
    size_t hl = naf_header_bytes(o, present[3], present[5], R.longest_line, R.n_sequences, lit);
    if (tl) memcpy(lit + hl, o->title, tl);
    segs[ns++] = { 0, hl + tl, 0, -1, -1 }; ll = hl + tl; pos = hl + tl;
    for (int s = 0; s < 6; s++) {
        if (!present[s]) continue;
        u64 orig = 0, comp = 0;
        for (uint32_t j = 0; j < n; j++) { orig += pc[j].raw[s]; comp += pc[j].len[s]; }
        if (s == 4) orig = R.n_bases;                                                              // ennaf.c:582: number of bases
        size_t h0 = ll; ll += vle(orig, lit + ll); ll += vle(comp, lit + ll);
        segs[ns++] = { pos, ll - h0, h0, -1, s }; pos += ll - h0;
        for (uint32_t j = 0; j < n; j++) if (pc[j].len[s]) { segs[ns++] = { pos, pc[j].len[s], pc[j].off[s], (int32_t)j, s }; pos += pc[j].len[s]; }
        R.section_orig[s] = orig; R.section_comp[s] = comp;
    }
    *n_segs = ns; *lit_len = ll; *naf_len = pos;
    if (rep) *rep = R;
    return 0;
}

extern "C" int naf_gpu_ennaf_stitch(naf_gpu_ctx *c, const naf_gpu_stitch_seg *segs, size_t n_segs, const uint8_t *lit, const void *const *bufs, void *d_naf, size_t cap)
{
    if (!c || !segs || !lit || !bufs || !d_naf) return NAF_GPU_EARG;
    for (size_t i = 0; i < n_segs; i++) {
        const naf_gpu_stitch_seg &g = segs[i];
        if (g.dst_off + g.len > cap) return ctx_fail(c, NAF_GPU_ECAP, "ennaf output capacity %zu too small", cap);
        if (!g.len) continue;
        if (g.shard < 0) HIP_TRY(c, hipMemcpyAsync((u8 *)d_naf + g.dst_off, lit + g.src_off, g.len, hipMemcpyHostToDevice, c->stream));
        else HIP_TRY(c, hipMemcpyAsync((u8 *)d_naf + g.dst_off, (const u8 *)bufs[g.shard] + g.src_off, g.len, hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));                                                   // lit is the caller's
    return 0;
}

void ennaf_shard_state_free(naf_gpu_ctx *c) { delete (EnnafShardState *)c->shard_state; c->shard_state = nullptr; }

extern "C" int naf_gpu_ennaf_shard_carry(const naf_gpu_shard_info *infos, uint32_t n, uint32_t k, naf_gpu_shard_carry *out)
{
    if (!infos || !out || n == 0 || n > NAF_GPU_MAX_SHARDS || k >= n) return NAF_GPU_EARG;
    ShardView V; shard_view(infos, n, k, V);
    memset(out, 0, sizeof *out);
    out->first_record = V.rec0; out->tail_extra = V.K.tail_extra; out->run_ext = V.K.run_ext;
    out->skip_first = V.K.skip_first; out->tail_hi = V.K.tail_hi; out->prev_masked = V.K.prev_masked; out->skip_run0 = V.K.skip_run0;
    for (int s = 0; s < 6; s++) { out->first[s] = V.first[s]; out->last[s] = V.last[s]; }
    return 0;
}
