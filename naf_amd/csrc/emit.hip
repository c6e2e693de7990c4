// emit.hip -- unnaf on gfx950: small-section decode, offset scans and the text re-emit kernels.
//
// Replaces (reference, unnaf/src): load_ids/load_names/load_lengths/load_mask input.c:145-246,
// write_4bit_as_fasta output.c:445-454 (+ codes_to_nucs utils.c:74-83), mask_dna_buffer output.c:295-322,
// print_dna_split_into_lines output.c:339-360, print_dna_buffer_as_fasta output.c:369-430,
// print_fasta output.c:608-674, print_fastq output-fastq.c:100-149, print_dna output.c:457-512,
// print_sequences output-sequences.c:60-116, print_4bit output.c:266-292.
//
// The reference streams 256 KiB buffers through a state machine; here every output byte is a pure
// function of prefix sums (record text offsets, record base offsets, mask toggle positions), so any
// 16-byte output chunk is produced independently by one lane:
//   rec_out[r]  = text offset of record r            (scan of header + body sizes)
//   rec_base[r] = index of record r's first base      (scan of lengths)
//   toggles[k]  = base index where the soft-mask state flips (scan of mask units, units != 255 toggle)
#include "ctx.h"
#include "wgscan.h"
#include "emit_core.h"
#include <thread>

// 16 bytes of output text that nothing on the device reads again: the nontemporal hint keeps them from displacing the compressed stream
// and the tables in L2 (k_emit_tile_flat: 3.08 -> 2.93 ms per 10 GB)
__device__ __forceinline__ void st_text16(u8 *p, const uint4 &v, int nt)
{
    if (nt) {
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t nv; nv.x = v.x; nv.y = v.y; nv.z = v.z; nv.w = v.w;
        __builtin_nontemporal_store(nv, (u32x4_t *)p);
    } else *(uint4 *)p = v;
}

// ---- prep kernels --------------------------------------------------------------------------------------------
__global__ void k_len_flags(const u32 *units, u64 n, u64 *flag)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = units[i] != 0xFFFFFFFFu;
}
__global__ void k_len_acc(const u32 *units, u64 n, const u64 *ridx, u64 *rec_len, u64 N)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ridx[i] < N) atomicAdd((unsigned long long *)&rec_len[ridx[i]], (unsigned long long)units[i]);
}

// positions of zero bytes: tile = 256 threads x 16 bytes
#define ZT_BYTES 16
#define ZT_TILE (256 * ZT_BYTES)
__device__ __forceinline__ u32 zero_byte_mask16(const u8 *p, u64 base, u64 n)
{
    u32 m = 0;
    if (base + ZT_BYTES <= n) {
        u64 a = ld64(p + base), b = ld64(p + base + 8);
#pragma unroll
        for (int i = 0; i < 8; i++) { if (((a >> (8 * i)) & 0xFF) == 0) m |= 1u << i; if (((b >> (8 * i)) & 0xFF) == 0) m |= 1u << (8 + i); }
    } else {
        for (int i = 0; i < ZT_BYTES; i++) if (base + i < n && p[base + i] == 0) m |= 1u << i;
    }
    return m;
}
__global__ __launch_bounds__(256) void k_zero_count(const u8 *p, u64 n, u64 *tile_cnt)
{
    __shared__ u64 lds[4];
    u64 base = (u64)blockIdx.x * ZT_TILE + (u64)threadIdx.x * ZT_BYTES;
    u64 c = base < n ? __popc(zero_byte_mask16(p, base, n)) : 0, tot;
    wg_scan_inclusive<u64, OpAdd>(c, &tot, lds);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void k_zero_scatter(const u8 *p, u64 n, const u64 *tile_pre, u64 *pos_out, u64 cap)
{
    __shared__ u64 lds[4];
    u64 base = (u64)blockIdx.x * ZT_TILE + (u64)threadIdx.x * ZT_BYTES;
    u32 m = base < n ? zero_byte_mask16(p, base, n) : 0;
    u64 c = __popc(m), tot;
    u64 incl = wg_scan_inclusive<u64, OpAdd>(c, &tot, lds);
    u64 k = tile_pre[blockIdx.x] + incl - c;
    while (m) { int b = __ffs(m) - 1; m &= m - 1; if (k < cap) pos_out[k] = base + b; k++; }
}

// mask units: running base position (u64 sum of units) + compaction of toggle positions (units != 255)
#define MT_TILE (256 * 16)
// 16 units of a lane as four words: their sum (v_sad_u8 against 0) and how many of them are not 255.  Units past n read as 255
// with sum 0 (the section buffer is padded by 32 bytes, so the loads themselves are safe).
__device__ __forceinline__ void mask_units16(const u8 *units, u64 base, u64 n, u32 w[4], u32 &sum, u32 &cnt)
{
    sum = 0; cnt = 0; w[0] = w[1] = w[2] = w[3] = 0xFFFFFFFFu;
    if (base >= n) return;
    const u64 a = ld64(units + base), b = ld64(units + base + 8);
    w[0] = (u32)a; w[1] = (u32)(a >> 32); w[2] = (u32)b; w[3] = (u32)(b >> 32);
    const u32 valid = n - base >= 16 ? 16u : (u32)(n - base);
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
        if (valid < 4 * k + 4) { const u32 keep = valid > 4 * k ? valid - 4 * k : 0; w[k] |= keep ? 0xFFFFFFFFu << (8 * keep) : 0xFFFFFFFFu; }
        const u32 y = ~w[k];                                                     // a byte of y is 0 where the unit is 255
        const u32 z = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);    // 0x80 in exactly those bytes
        cnt += 4 - (u32)__popc(z);
        sum += __builtin_amdgcn_sad_u8(w[k], 0u, 0u) - 255u * (u32)__popc(z);
    }
    // (the 255s are added back by the caller: sum + 255 * (16 - cnt) would count the padding units too)
}
__global__ __launch_bounds__(256) void k_mask_count(const u8 *units, u64 n, u64 *tile_sum, u64 *tile_cnt)
{
    __shared__ u64 lds[4];
    u64 base = (u64)blockIdx.x * MT_TILE + (u64)threadIdx.x * 16;
    u32 w[4], s32, c32; mask_units16(units, base, n, w, s32, c32);
    const u32 valid = base >= n ? 0u : (n - base >= 16 ? 16u : (u32)(n - base));
    u64 s = (u64)s32 + 255ull * (valid - c32), c = c32;
    u64 t1, t2;
    wg_scan_inclusive<u64, OpAdd>(s, &t1, lds);
    wg_scan_inclusive<u64, OpAdd>(c, &t2, lds);
    if (threadIdx.x == 0) { tile_sum[blockIdx.x] = t1; tile_cnt[blockIdx.x] = t2; }
}
__global__ __launch_bounds__(256) void k_mask_scatter(const u8 *units, u64 n, const u64 *tile_sum_pre, const u64 *tile_cnt_pre, u64 *toggles, u64 n_toggles)
{
    __shared__ u64 lds[4];
    // a tile of nothing but 255s (the long runs of an unmasked genome) has no toggle to write
    if ((blockIdx.x + 1 < gridDim.x ? tile_cnt_pre[blockIdx.x + 1] : n_toggles) == tile_cnt_pre[blockIdx.x]) return;
    u64 base = (u64)blockIdx.x * MT_TILE + (u64)threadIdx.x * 16;
    u32 w[4], s32, c32; mask_units16(units, base, n, w, s32, c32);
    const u32 valid = base >= n ? 0u : (n - base >= 16 ? 16u : (u32)(n - base));
    u64 s = (u64)s32 + 255ull * (valid - c32), c = c32;
    u64 t;
    u64 si = wg_scan_inclusive<u64, OpAdd>(s, &t, lds);
    u64 ci = wg_scan_inclusive<u64, OpAdd>(c, &t, lds);
    if (!c32) return;
    u64 pos = tile_sum_pre[blockIdx.x] + si - s, k = tile_cnt_pre[blockIdx.x] + ci - c;
    for (u32 i = 0; i < valid; i++) { u32 u = (w[i >> 2] >> (8 * (i & 3))) & 0xFF; pos += u; if (u != 255) toggles[k++] = pos; }
}

// The mask of a text without (or almost without) lower case is a few KB of RLE blocks that regenerate tens of MB of 0xFF units, and the
// general zstd pipeline -- index, parse, tables, copies, then k_mask_count / scans / k_mask_scatter: some 45 dependent launches and four
// read-backs -- was what the tile index of a 10 GB decode waited for.  One workgroup takes the whole frame through LDS instead: a run
// of n units of 255 moves the base position by 255 n, every other unit is a toggle.  Frames with a compressed block, more than `cap`
// toggles or anything unexpected are left to the general path (res[0] = 0).  res: ok, toggles, units.
#define MASK_RLE_SRC (48u * 1024u)
#define MASK_RLE_TOG 8192u
__global__ __launch_bounds__(256) void k_mask_rle_frame(const u8 *src, u32 len, u32 hdr, u64 expect_units, u64 *toggles, u32 cap, u64 *res)
{
    __builtin_amdgcn_s_setprio(3);                           // a serial chain: first in line for the SIMD's issue slots beside the bulk kernels of the other streams
    extern __shared__ __attribute__((aligned(16))) u8 mbuf[];
    __shared__ u64 lds[4]; __shared__ u32 s_first;
    const u32 t = threadIdx.x;
    for (u32 i = t; i < len; i += 256) mbuf[i] = src[i];
    if (t == 0) s_first = 0xFFFFFFFFu;
    __syncthreads();
    // the leading RLE blocks of 0xFF units, four bytes each: all of them at once
    const u32 nA = len > hdr ? (len - hdr) / 4 : 0;
    for (u32 i = t; i < nA; i += 256) {
        const u32 p = hdr + 4 * i, h = (u32)mbuf[p] | ((u32)mbuf[p + 1] << 8) | ((u32)mbuf[p + 2] << 16);
        if (!(((h >> 1) & 3) == 1 && !(h & 1) && mbuf[p + 3] == 255)) { atomicMin(&s_first, i); break; }   // (a thread's later blocks lie behind this one)
    }
    __syncthreads();
    const u32 F = s_first < nA ? s_first : nA;
    u64 mine = 0;
    for (u32 i = t; i < F; i += 256) { const u32 p = hdr + 4 * i; mine += ((u32)mbuf[p] | ((u32)mbuf[p + 1] << 8) | ((u32)mbuf[p + 2] << 16)) >> 3; }
    u64 units = 0;
    wg_scan_inclusive<u64, OpAdd>(mine, &units, lds);
    u64 base = 255ull * units;
    // what follows (any RLE block, Raw blocks), block after block, every thread on the same block
    u32 pos = hdr + 4 * F, nt = 0; bool ok = true, done = false;
    while (ok && !done) {
        if (pos + 3 > len) { ok = false; break; }
        const u32 h = (u32)mbuf[pos] | ((u32)mbuf[pos + 1] << 8) | ((u32)mbuf[pos + 2] << 16);
        const u32 last = h & 1, type = (h >> 1) & 3, size = h >> 3;
        if (type == 1) {
            if (pos + 4 > len) { ok = false; break; }
            const u32 v = mbuf[pos + 3];
            if (v != 255) {
                if ((u64)nt + size > cap) { ok = false; break; }
                for (u32 k = t; k < size; k += 256) toggles[nt + k] = base + (u64)v * (k + 1);
                nt += size;
            }
            base += (u64)v * size; units += size; pos += 4;
        } else if (type == 0) {
            if ((u64)pos + 3 + size > len) { ok = false; break; }
            const u8 *rb = mbuf + pos + 3;
            const u32 per = (size + 255) / 256, lo = t * per < size ? t * per : size, hi = lo + per < size ? lo + per : size;
            u64 sm = 0, cn = 0;
            for (u32 k = lo; k < hi; k++) { const u32 u = rb[k]; sm += u; cn += u != 255; }
            u64 tot_s, tot_c;
            const u64 is = wg_scan_inclusive<u64, OpAdd>(sm, &tot_s, lds);
            const u64 ic = wg_scan_inclusive<u64, OpAdd>(cn, &tot_c, lds);
            if (nt + tot_c > cap) { ok = false; break; }
            u64 at = base + is - sm; u32 idx = nt + (u32)(ic - cn);
            for (u32 k = lo; k < hi; k++) { const u32 u = rb[k]; at += u; if (u != 255) toggles[idx++] = at; }
            base += tot_s; nt += (u32)tot_c; units += size; pos += 3 + size;
        } else ok = false;
        if (last) done = true;
    }
    ok = ok && done && pos == len && units == expect_units;
    if (t == 0) { res[0] = ok ? 1 : 0; res[1] = nt; res[2] = units; }
}

// per-record header length and text size
__device__ __forceinline__ void rec_size_of(u64 r, u64 len, const u64 *idz, const u64 *nmz, int has_ids, int has_names, int mode, u64 L, u32 *hdr, u64 *out)
{
    u64 h = 0;
    if (mode == EM_FASTA || mode == EM_FASTQ) {
        u64 idl = 0, nml = 0;
        if (has_ids) idl = idz[r] - (r ? idz[r - 1] + 1 : 0);
        if (has_names) nml = nmz[r] - (r ? nmz[r - 1] + 1 : 0);
        u64 name = has_ids ? idl + ((has_names && nml) ? 1 + nml : 0) : nml;     // output.c:105-124
        h = 1 + name + 1;
    }
    u64 body;
    if (mode == EM_FASTQ) body = 2 * len + 4;                                       // SEQ \n + \n QUAL \n
    else if (mode == EM_SEQUENCES) body = len + 1;
    else body = len ? len + (L ? (len + L - 1) / L : 1) : 0;                        // ceil(len/L) newlines, none if empty
    *hdr = (u32)h; *out = h + body;
}
__global__ void k_rec_sizes(u64 N, const u64 *rec_len, const u64 *idz, const u64 *nmz, int has_ids, int has_names,
                            int mode, u64 L, u32 *hdr_len, u64 *out_size, u64 *base_size)
{
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    u64 len = rec_len[r];
    rec_size_of(r, len, idz, nmz, has_ids, has_names, mode, L, &hdr_len[r], &out_size[r]);
    base_size[r] = len;
}

// The record tables of an archive of few records in ONE launch behind the one that decodes its ids, names and lengths
// (k_small_frames): the chain above -- zero positions of two streams, lengths, sizes, two scans -- is sixteen launches and four
// read-backs of a few microseconds of work each, and the emit of a 10 GB archive of a hundred chromosomes waits for the last of
// them (profiles/r04_timeline_uniform_10GB.txt: 0.55 ms, a fifth of a millisecond longer than the sequence stream's own front).
// One workgroup; status: [0] 1 = the frames were good and the tables are made (0: the caller takes the long way, which also words the
// errors), [1] zero bytes among the ids, [2] among the names, [3] records the lengths describe, [4] bytes of text, [5] bases.
#define SIDE_FUSED_N 4096u
struct SideJob { const u32 *res; u32 n_jobs; u32 cap[3], len[3];
                 const u8 *ids, *names; const u32 *lens; u64 ids_n, names_n, n_len;
                 u64 N, L; int has_ids, has_names, mode;
                 u64 *idz, *nmz, *rec_len, *rec_out, *rec_base, *status; u32 *hdr_len; };
__device__ u64 wg_zero_positions(const u8 *p, u64 n, u64 *pos, u64 cap, u64 *lds)
{
    u64 run = 0;
    for (u64 tb = 0; tb < n; tb += ZT_TILE) {
        const u64 base = tb + (u64)threadIdx.x * ZT_BYTES;
        u32 m = base < n ? zero_byte_mask16(p, base, n) : 0;
        u64 c = __popc(m), tot;
        const u64 incl = wg_scan_inclusive<u64, OpAdd>(c, &tot, lds);
        u64 k = run + incl - c;
        while (m) { int b = __ffs(m) - 1; m &= m - 1; if (k < cap) pos[k] = base + b; k++; }
        run += tot;
    }
    return run;
}
__global__ __launch_bounds__(256) void k_side_tables(SideJob J)
{
    __shared__ u64 lds[4];
    __shared__ unsigned long long s_len[SIDE_FUSED_N];
    const u32 t = threadIdx.x;
    bool ok = true;
    for (u32 q = 0; q < J.n_jobs; q++) ok = ok && J.res[4 * q] == 0 && J.res[4 * q + 1] == J.cap[q] && J.res[4 * q + 2] == J.len[q];
    if (!ok) { if (t == 0) J.status[0] = 0; return; }
    for (u32 r = t; r < SIDE_FUSED_N; r += 256) s_len[r] = 0;
    const u64 n_ids = J.has_ids ? wg_zero_positions(J.ids, J.ids_n, J.idz, J.N, lds) : 0;
    const u64 n_names = J.has_names ? wg_zero_positions(J.names, J.names_n, J.nmz, J.N, lds) : 0;
    // lengths: a record's units add up, the one that is not 0xFFFFFFFF is its last (k_len_flags / k_len_acc)
    u64 nrec = 0;
    for (u64 b = 0; b < J.n_len; b += 256) {
        const u64 i = b + t; const u32 u = i < J.n_len ? J.lens[i] : 0xFFFFFFFFu;
        u64 f = i < J.n_len && u != 0xFFFFFFFFu, tot;
        const u64 incl = wg_scan_inclusive<u64, OpAdd>(f, &tot, lds);
        const u64 ridx = nrec + incl - f;
        if (i < J.n_len && ridx < J.N) atomicAdd(&s_len[ridx], (unsigned long long)u);
        nrec += tot;
    }
    __syncthreads();
    u64 run_out = 0, run_base = 0;
    for (u64 b = 0; b < J.N; b += 256) {
        const u64 r = b + t; u64 len = 0, o = 0, tot_o, tot_b; u32 hd = 0;
        if (r < J.N) { len = s_len[r]; rec_size_of(r, len, J.idz, J.nmz, J.has_ids, J.has_names, J.mode, J.L, &hd, &o); }
        const u64 io = wg_scan_inclusive<u64, OpAdd>(o, &tot_o, lds), ib = wg_scan_inclusive<u64, OpAdd>(len, &tot_b, lds);
        if (r < J.N) { J.rec_len[r] = len; J.hdr_len[r] = hd; J.rec_out[r] = run_out + io - o; J.rec_base[r] = run_base + ib - len; }
        run_out += tot_o; run_base += tot_b;
    }
    if (t == 0) {
        J.rec_len[J.N] = 0; J.rec_out[J.N] = run_out; J.rec_base[J.N] = run_base;
        J.status[1] = n_ids; J.status[2] = n_names; J.status[3] = nrec; J.status[4] = run_out; J.status[5] = run_base; J.status[0] = 1;
    }
}

// ---- emit ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 lut_char(const EmitP &P, u32 code)
{
    return (P.lut[code >> 2] >> (8 * (code & 3))) & 0xFF;
}

// Packed byte q of a flat frame that is read in place (ctx.h: ZFlat): its stream by bisection, then the 4-bit code at its known
// bit position.  The slow way -- for the few chunks the tile kernel does not take (headers, record ends, stream boundaries).
__device__ u32 flat_packed_byte(const EmitP &P, u64 q)
{
    if (P.ftail && q >= P.ftail_q) {                             // the frame's final Raw or (bit 31 of the length) RLE block
        const u64 k = q - P.ftail_q;
        return k < (P.ftail_n & 0x7FFFFFFFu) ? P.ftail[(P.ftail_n >> 31) ? 0 : k] : 0u;
    }
    const FlatStream *si = (const FlatStream *)P.fsi;
    u64 lo = 0, hi = P.fslots;                                   // last slot with q0 <= q
    while (hi - lo > 1) { const u64 mid = (lo + hi) >> 1; if (si[mid].q0 <= q) lo = mid; else hi = mid; }
    if (q >= si[lo + 1].q0) return 0;                            // past the end of the data
    if (si[lo].A == FLAT_DECODED) return P.seq[q];               // a block that was decoded (ctx.h: ZFlat, `cls`)
    const u64 B = si[lo].A - 4 * (q - si[lo].q0 + 1), a = B >> 3; const u32 sh = (u32)B & 7;
    u32 v = P.fsrc[a]; if (sh > 4) v |= (u32)P.fsrc[a + 1] << 8;
    return P.fsym[(v >> sh) & 15];
}

template <bool FOURBIT>
__device__ __forceinline__ u32 base_char(const EmitP &P, u64 g)
{
    if (FOURBIT && P.fsrc) { u32 b = flat_packed_byte(P, g >> 1); return lut_char(P, (g & 1) ? (b >> 4) : (b & 15)); }
    if (FOURBIT) { u32 b = P.seq[g >> 1]; return lut_char(P, (g & 1) ? (b >> 4) : (b & 15)); }
    u32 ch = P.seq[g];
    if (P.upper && ch >= 'a' && ch <= 'z') ch -= 32;                               // output.c:363-366 toupper
    return ch;
}

__device__ __forceinline__ bool base_masked(const EmitP &P, u64 g, u64 klo, u64 khi)
{
    return (upper_bound_u64(P.toggles, klo, khi, g) & 1) != 0;                      // output.c:295-322 as parity of toggles <= g
}

__device__ __forceinline__ u32 header_char(const EmitP &P, u64 r, u64 k, u32 hl)
{
    if (k == 0) return P.hdr_char;
    if (k == hl - 1) return '\n';
    k -= 1;
    u64 ids0 = 0, idl = 0, nm0 = 0;
    if (P.has_ids) { ids0 = r ? P.idz[r - 1] + 1 : 0; idl = P.idz[r] - ids0; }
    if (P.has_names) nm0 = r ? P.nmz[r - 1] + 1 : 0;
    if (P.has_ids) {
        if (k < idl) return P.ids[ids0 + k];
        if (k == idl) return P.sep;
        return P.names[nm0 + (k - idl - 1)];
    }
    return P.names[nm0 + k];
}

// One output byte (slow path; every edge case goes through here).
template <bool FOURBIT>
__device__ u32 emit_byte(const EmitP &P, u64 p, u64 rlo, u64 rhi, u64 klo, u64 khi)
{
    if (P.mode == EM_SEQ) {
        u32 ch = base_char<FOURBIT>(P, p);
        if (P.masking && base_masked(P, p, klo, khi)) ch += 32;
        return ch;
    }
    u64 r = upper_bound_u64(P.rec_out, rlo, rhi + 1, p) - 1;
    u64 off = p - P.rec_out[r];
    u32 hl = P.hdr_len[r];
    if (off < hl) return header_char(P, r, off, hl);
    u64 q = off - hl, len = P.rec_len[r], j;
    if (P.mode == EM_FASTQ) {
        if (q < len) return base_char<FOURBIT>(P, P.rec_base[r] + q);              // unnaf.c:442: never masked
        q -= len;
        if (q == 0) return '\n';
        if (q == 1) return '+';
        if (q == 2) return '\n';
        q -= 3;
        if (q < len) return P.qual[P.rec_base[r] + q];
        return '\n';
    }
    if (P.mode == EM_SEQUENCES || P.L == 0) { if (q >= len) return '\n'; j = q; }
    else {
        u64 line = q / (P.L + 1), col = q - line * (P.L + 1);
        if (col == P.L) return '\n';
        j = line * P.L + col;
        if (j >= len) return '\n';
    }
    u64 g = P.rec_base[r] + j;
    u32 ch = base_char<FOURBIT>(P, g);
    if (P.masking && base_masked(P, g, klo, khi)) ch += 32;
    return ch;
}

// 16 consecutive bases starting at base index g as ASCII in two u64 (little endian: base 0 in byte 0).
template <bool FOURBIT>
__device__ __forceinline__ void bases16(const EmitP &P, u64 g, u64 &lo, u64 &hi)
{
    if (!FOURBIT) {
        lo = ld64(P.seq + g); hi = ld64(P.seq + g + 8);
        if (P.upper) {
            // toupper on ASCII letters only, 8 bytes at a time
            auto up = [](u64 v) { u64 r = 0; for (int i = 0; i < 8; i++) { u32 c = (v >> (8 * i)) & 0xFF; if (c >= 'a' && c <= 'z') c -= 32; r |= (u64)c << (8 * i); } return r; };
            lo = up(lo); hi = up(hi);
        }
        return;
    }
    if (P.fsrc) {
        u64 nib = 0;
        for (u32 k = 0; k < 8; k++) nib |= (u64)flat_packed_byte(P, (g >> 1) + k) << (8 * k);
        if (g & 1) nib = (nib >> 4) | ((u64)flat_packed_byte(P, (g >> 1) + 8) << 60);
        expand16(P.lut, nib, lo, hi);
        return;
    }
    const u8 *a = P.seq + (g >> 1);
    u64 nib = ld64(a);
    if (g & 1) nib = (nib >> 4) | ((u64)a[8] << 60);
    expand16(P.lut, nib, lo, hi);
}

// One workgroup produces EMIT_SPAN consecutive output bytes as EMIT_SPAN/4096 tiles of 256 lanes x 16 B.
// The record window [rlo, rhi] and the toggle window [klo, khi] are found once per span (they need dependent
// binary-search loads); inside the span a lane keeps its current record's geometry in registers and only
// searches again when a chunk leaves that record.
#define EMIT_SPAN (64u * 1024u)
template <bool FOURBIT>
__global__ __launch_bounds__(256) void k_emit(EmitP P, u8 *out)
{
    __shared__ u64 sh[4];                                       // rlo, rhi, klo, khi for this span
    u64 span_first = P.out_begin + (u64)blockIdx.x * EMIT_SPAN, span_last = span_first + EMIT_SPAN - 1;
    if (span_last >= P.out_end) span_last = P.out_end - 1;
    if (threadIdx.x < 2 && P.mode != EM_SEQ) {
        u64 p = threadIdx.x == 0 ? span_first : span_last;
        sh[threadIdx.x] = upper_bound_u64(P.rec_out, 0, P.N + 1, p) - 1;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        u64 k = 0;
        if (P.masking) {
            // toggle window: every toggle that can affect a base this span touches
            u64 g;
            if (P.mode == EM_SEQ) g = threadIdx.x == 0 ? span_first : span_last + 1;
            else g = threadIdx.x == 0 ? P.rec_base[sh[0]] : P.rec_base[sh[1] + 1];
            if (P.mode != EM_SEQ && threadIdx.x == 0) {
                // tighten the lower bound: first base the span can touch inside record rlo
                u64 r = sh[0], off = span_first - P.rec_out[r], hl = P.hdr_len[r];
                if (off > hl) { u64 q = off - hl; u64 j = (P.mode == EM_FASTA && P.L) ? (q / (P.L + 1)) * P.L : (P.mode == EM_FASTQ ? 0 : q); if (j > P.rec_len[r]) j = P.rec_len[r]; g += j; }
            }
            k = upper_bound_u64(P.toggles, 0, P.n_toggles, g);
        }
        sh[2 + threadIdx.x] = k;
    }
    __syncthreads();
    const u64 rlo = sh[0], rhi = sh[1], klo = sh[2], khi = sh[3];
    const bool any_toggle = P.masking && klo < khi;
    // cached geometry of the lane's current record, and the lane's (line, col) inside it: consecutive tiles of a lane
    // are 4096 bytes apart, so (line, col) advances by a constant instead of a division per chunk
    u64 c_ro = 1, c_rn = 0, c_body = 0, c_len = 0, c_base = 0;
    u64 w_line = 0, w_q0 = 0; u32 w_col = 0; bool w_ok = false;
    const u32 Lp1 = (P.L >> 31) ? 0 : (u32)P.L + 1;
    const u32 dq = Lp1 ? 4096u / Lp1 : 0, dr = Lp1 ? 4096u % Lp1 : 0;

    struct Chunk { u64 p0, g0, len, j0, qoff; int nl_b; u32 nbytes; bool fast, qual_copy, live; };
    // geometry of one 16-byte chunk (no payload access)
    auto prep = [&](u64 tile) -> Chunk {
        Chunk k; k.p0 = tile + (u64)threadIdx.x * 16; k.g0 = 0; k.len = 0; k.j0 = 0; k.qoff = 0; k.nl_b = 64; k.fast = false; k.qual_copy = false;
        k.live = tile <= span_last && k.p0 < P.out_end; k.nbytes = 0;
        if (!k.live) return k;
        u64 p0 = k.p0;
        k.nbytes = P.out_end - p0 < 16 ? (u32)(P.out_end - p0) : 16;
        if (k.nbytes == 16 && !P.force_slow) {
            if (P.mode == EM_SEQ) { k.fast = true; k.g0 = p0; k.len = ~0ull; }
            else {
                if (!(p0 >= c_ro && p0 < c_rn)) {
                    u64 r = upper_bound_u64(P.rec_out, rlo, rhi + 1, p0) - 1;
                    c_ro = P.rec_out[r]; c_rn = P.rec_out[r + 1]; c_body = c_ro + P.hdr_len[r]; c_len = P.rec_len[r]; c_base = P.rec_base[r];
                    w_ok = false;
                }
                if (p0 >= c_body && p0 + 16 <= c_rn) {
                    u64 q0 = p0 - c_body; k.len = c_len;
                    if (P.mode == EM_FASTQ) {
                        if (q0 + 16 <= c_len) { k.fast = true; k.j0 = q0; k.g0 = c_base + q0; k.len = ~0ull; }
                        else if (q0 >= c_len + 3 && q0 + 16 <= 2 * c_len + 3) { k.qual_copy = true; k.qoff = c_base + (q0 - c_len - 3); }
                    } else if (P.mode == EM_SEQUENCES || P.L == 0) { k.fast = true; k.j0 = q0; k.g0 = c_base + q0; }
                    else if (P.L >= 16) {
                        u64 line, col;
                        if (Lp1 && w_ok && q0 == w_q0 + 4096) {      // same record as this lane's previous tile
                            u32 c2 = w_col + dr; line = w_line + dq;
                            if (c2 >= Lp1) { c2 -= Lp1; line++; }
                            col = c2;
                        }
                        else if ((q0 >> 32) == 0 && Lp1) { u32 l32 = (u32)q0 / Lp1; line = l32; col = (u32)q0 - l32 * Lp1; }
                        else { line = q0 / (P.L + 1); col = q0 - line * (P.L + 1); }
                        w_q0 = q0; w_line = line; w_col = (u32)col; w_ok = true;
                        k.j0 = line * P.L + col; k.g0 = c_base + k.j0;
                        u64 d = P.L - col;                      // byte index of the line-end newline
                        k.nl_b = d < 16 ? (int)d : 64;
                        k.fast = true;
                    }
                }
            }
        }
        return k;
    };
    // payload fetch: 16 nibbles (or 16 text bytes) starting at base g0
    auto fetch = [&](const Chunk &k, u64 &a, u64 &b) {
        a = b = 0;
        if (!k.live) return;
        if (k.qual_copy) { a = ld64(P.qual + k.qoff); b = ld64(P.qual + k.qoff + 8); }
        else if (k.fast) {
            if (FOURBIT) { const u8 *s = P.seq + (k.g0 >> 1); a = ld64(s); b = s[8]; }
            else { a = ld64(P.seq + k.g0); b = ld64(P.seq + k.g0 + 8); }
        }
    };
    auto finish = [&](const Chunk &k, u64 a, u64 b) {
        if (!k.live) return;
        u8 *o = out + (k.p0 - P.out_begin);
        if (k.qual_copy) { uint4 v; v.x = (u32)a; v.y = (u32)(a >> 32); v.z = (u32)b; v.w = (u32)(b >> 32); memcpy(o, &v, 16); return; }
        if (k.fast) {
            u64 lo, hi, g0 = k.g0;
            if (FOURBIT) { u64 nib = a; if (g0 & 1) nib = (nib >> 4) | (b << 60); expand16(P.lut, nib, lo, hi); }
            else {
                lo = a; hi = b;
                if (P.upper) {
                    auto up = [](u64 v) { u64 r = 0; for (int i = 0; i < 8; i++) { u32 c = (v >> (8 * i)) & 0xFF; if (c >= 'a' && c <= 'z') c -= 32; r |= (u64)c << (8 * i); } return r; };
                    lo = up(lo); hi = up(hi);
                }
            }
            if (any_toggle) {
                u64 kk = upper_bound_u64(P.toggles, klo, khi, g0);  // toggles <= g0
                u32 state = (u32)(kk & 1), m16 = 0; u64 pos = g0;
                for (;;) {                                      // walk the toggles that fall inside (g0, g0+16)
                    u64 nxt = kk < khi ? P.toggles[kk] : ~0ull;
                    u64 end = nxt < g0 + 16 ? nxt : g0 + 16;
                    if (state && end > pos) m16 |= (u32)(((1u << (end - pos)) - 1) << (pos - g0));
                    if (nxt >= g0 + 16) break;
                    pos = nxt; state ^= 1; kk++;
                }
                lo += spread_bits8(m16 & 0xFF); hi += spread_bits8(m16 >> 8);
            } else if (P.masking && (klo & 1)) { lo += 0x2020202020202020ull; hi += 0x2020202020202020ull; }   // whole span inside one masked run
            if (k.nl_b < 16) splice_newline(lo, hi, k.nl_b);
            u64 j15 = k.j0 + 15 - (15 > k.nl_b ? 1 : 0);
            if (j15 >= k.len) hi = (hi & 0x00FFFFFFFFFFFFFFull) | ((u64)'\n' << 56);     // record-end newline
            uint4 v; v.x = (u32)lo; v.y = (u32)(lo >> 32); v.z = (u32)hi; v.w = (u32)(hi >> 32);
            memcpy(o, &v, 16);
            return;
        }
        for (u32 bb = 0; bb < k.nbytes; bb++) o[bb] = (u8)emit_byte<FOURBIT>(P, k.p0 + bb, rlo, rhi, klo, khi);
    };
    // two tiles per iteration: both payload loads are in flight before either chunk is expanded and stored
    for (u64 tile = span_first; tile <= span_last; tile += 8192) {
        Chunk ka = prep(tile), kb = prep(tile + 4096);
        u64 a0, a1, b0, b1;
        fetch(ka, a0, a1); fetch(kb, b0, b1);
        finish(ka, a0, a1); finish(kb, b0, b1);
    }
}

// ---- short-record emit ----------------------------------------------------------------------------------------
// FASTQ reads and short FASTA records put a record boundary into (almost) every 16-byte chunk, so the long-record
// kernel above would take its byte-at-a-time path everywhere.  Here a workgroup stages the geometry of every record
// its span touches in LDS (binary search and geometry loads stop being dependent global loads), and a lane
// composes its 16 output bytes from SEGMENTS -- header bytes, up to 16 bases, the "\n+\n" joint, up to 16 quality
// bytes, a newline -- each placed with one 128-bit shift.
#define ES_SPAN (32u * 1024u)
#define ES_CAP 1024

// keep the first n bytes of {slo,shi}, shift them to byte offset pos of {lo,hi}
__device__ __forceinline__ void place16(u64 &lo, u64 &hi, u64 slo, u64 shi, u32 pos, u32 n)
{
    if (n <= 8) { shi = 0; slo &= low_bytes_mask((int)n); } else if (n < 16) shi &= low_bytes_mask((int)n - 8);
    if (pos == 0) { lo |= slo; hi |= shi; }
    else if (pos < 8) { lo |= slo << (8 * pos); hi |= (shi << (8 * pos)) | (slo >> (64 - 8 * pos)); }
    else if (pos == 8) hi |= slo;
    else hi |= slo << (8 * (pos - 8));
}

// +32 on the bases of {lo,hi} (16 bases from base index g0) that lie in masked runs (output.c:295-322)
// tog[0..n) are toggles number klo .. klo+n-1 (global memory, or the tile's window staged in LDS)
__device__ __forceinline__ void mask16_from(const u64 *tog, u32 n, u64 klo, u64 g0, u64 &lo, u64 &hi)
{
    u32 a = 0, b = n;                                           // toggles <= g0
    while (a < b) { u32 mid = (a + b) >> 1; if (tog[mid] <= g0) a = mid + 1; else b = mid; }
    u32 kk = a, state = (u32)((klo + kk) & 1), m16 = 0; u64 pos = g0;
    for (;;) {
        u64 nxt = kk < n ? tog[kk] : ~0ull;
        u64 end = nxt < g0 + 16 ? nxt : g0 + 16;
        if (state && end > pos) m16 |= (u32)(((1u << (end - pos)) - 1) << (pos - g0));
        if (nxt >= g0 + 16) break;
        pos = nxt; state ^= 1; kk++;
    }
    lo += spread_bits8(m16 & 0xFF); hi += spread_bits8(m16 >> 8);
}
__device__ __forceinline__ void mask16(const EmitP &P, u64 klo, u64 khi, bool any_toggle, u64 g0, u64 &lo, u64 &hi)
{
    if (any_toggle) {
        if (khi - klo <= 0x7FFFFFFFull) mask16_from(P.toggles + klo, (u32)(khi - klo), klo, g0, lo, hi);
        else {
            u64 kk = upper_bound_u64(P.toggles, klo, khi, g0);      // toggles <= g0
            u32 state = (u32)(kk & 1), m16 = 0; u64 pos = g0;
            for (;;) {
                u64 nxt = kk < khi ? P.toggles[kk] : ~0ull;
                u64 end = nxt < g0 + 16 ? nxt : g0 + 16;
                if (state && end > pos) m16 |= (u32)(((1u << (end - pos)) - 1) << (pos - g0));
                if (nxt >= g0 + 16) break;
                pos = nxt; state ^= 1; kk++;
            }
            lo += spread_bits8(m16 & 0xFF); hi += spread_bits8(m16 >> 8);
        }
    } else if (P.masking && (klo & 1)) { lo += 0x2020202020202020ull; hi += 0x2020202020202020ull; }
}

// record window [rlo, rhi] and toggle window [klo, khi] of one span -> sh[0..3]; needs blockDim.x >= 2
__device__ __forceinline__ void span_windows(const EmitP &P, u64 span_first, u64 span_last, u64 *sh)
{
    if (threadIdx.x < 2 && P.mode != EM_SEQ) {
        u64 p = threadIdx.x == 0 ? span_first : span_last;
        sh[threadIdx.x] = upper_bound_u64(P.rec_out, 0, P.N + 1, p) - 1;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        u64 k = 0;
        if (P.masking) {
            u64 g;
            if (P.mode == EM_SEQ) g = threadIdx.x == 0 ? span_first : span_last + 1;
            else g = threadIdx.x == 0 ? P.rec_base[sh[0]] : P.rec_base[sh[1] + 1];
            if (P.mode != EM_SEQ && threadIdx.x == 0) {
                u64 r = sh[0], off = span_first - P.rec_out[r], hl = P.hdr_len[r];
                if (off > hl) { u64 q = off - hl; u64 j = (P.mode == EM_FASTA && P.L) ? (q / (P.L + 1)) * P.L : (P.mode == EM_FASTQ ? 0 : q); if (j > P.rec_len[r]) j = P.rec_len[r]; g += j; }
            }
            k = upper_bound_u64(P.toggles, 0, P.n_toggles, g);
        }
        sh[2 + threadIdx.x] = k;
    }
    __syncthreads();
}

// Header lines of records [r0, r0+n) as one contiguous byte stream ('>'|'@' name '\n', output.c:105-132), so that
// the emit kernel copies header bytes like any other byte source.  One lane per record.
__global__ void k_hdr_len64(const u32 *hdr_len, u64 r0, u64 n, u64 *out)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = hdr_len[r0 + i];
}
__device__ __forceinline__ void store_upto16(u8 *p, u64 lo, u64 hi, u32 n)
{
    if (n >= 16) { uint4 v; v.x = (u32)lo; v.y = (u32)(lo >> 32); v.z = (u32)hi; v.w = (u32)(hi >> 32); memcpy(p, &v, 16); return; }
    if (n & 8) { st64(p, lo); p += 8; lo = hi; }
    if (n & 4) { st32(p, (u32)lo); p += 4; lo >>= 32; }
    if (n & 2) { p[0] = (u8)lo; p[1] = (u8)(lo >> 8); p += 2; lo >>= 16; }
    if (n & 1) p[0] = (u8)lo;
}
template <u32 LANES = 16>
__device__ __forceinline__ void group_copy(u8 *dst, const u8 *src, u64 n, u32 g)      // LANES lanes, sources padded by >= 16 bytes
{
    for (u64 i = (u64)g * 16; i < n; i += LANES * 16) {
        u64 lo = ld64(src + i), hi = ld64(src + i + 8);
        store_upto16(dst + i, lo, hi, n - i < 16 ? (u32)(n - i) : 16u);
    }
}

// eight lanes per header, 16 bytes per lane and step (one lane per header walking it byte by byte ran at 90 GB/s)
__global__ __launch_bounds__(256) void k_hdr_build(EmitP P, u64 r0, u64 n, const u64 *hdr_off, u8 *text)
{
    u64 i = (u64)blockIdx.x * 32 + (threadIdx.x >> 3);
    u32 g = threadIdx.x & 7;
    if (i >= n) return;
    u64 r = r0 + i; u32 hl = P.hdr_len[r];
    u8 *o = text + hdr_off[i];
    u64 ids0 = 0, idl = 0, nm0 = 0, nml = 0;
    if (P.has_ids) { ids0 = r ? P.idz[r - 1] + 1 : 0; idl = P.idz[r] - ids0; }
    if (P.has_names) { nm0 = r ? P.nmz[r - 1] + 1 : 0; nml = P.nmz[r] - nm0; }
    if (g == 0) { o[0] = P.hdr_char; o[hl - 1] = '\n'; }
    if (P.has_ids) {
        group_copy<8>(o + 1, P.ids + ids0, idl, g);
        if (P.has_names && nml) { if (g == 0) o[1 + idl] = P.sep; group_copy<8>(o + 2 + idl, P.names + nm0, nml, g); }
    } else group_copy<8>(o + 1, P.names + nm0, nml, g);
}

// One 16-byte chunk composed from segments.  One iteration = one SEGMENT: n_main bytes copied from a byte source or
// expanded from bases, followed by up to 3 constant bytes (the newline that ends a line / read / quality string, or the
// "\n+\n" joint).  `geo(r, ...)` supplies a record's geometry (LDS or global).
template <bool FOURBIT, typename Geo>
__device__ __forceinline__ void compose_chunk(const EmitP &P, Geo &geo, u64 p0, u32 nbytes, u64 r, u64 klo, u64 khi, bool any_toggle, u32 Lp1_32, u8 *o)
{
    const bool hdrs = P.hdr_text != nullptr;
    u64 ro, rn, len, base, ho; u32 hl;
    geo(r, ro, rn, len, base, ho, hl);
    u64 lo = 0, hi = 0; u32 pos = 0;
    while (pos < nbytes) {
        u64 p = p0 + pos;
        while (p >= rn) { r++; geo(r, ro, rn, len, base, ho, hl); }   // every record prints at least one byte
        u64 off = p - ro;
        const u8 *src = nullptr; u64 g = 0; bool is_bases = false;
        u64 n_main = 0; u32 tc = 0, n_tc = 0;
        if (off < hl) { src = P.hdr_text + ho + off; n_main = hl - off; }
        else {
            u64 q = off - hl;
            if (P.mode == EM_FASTQ) {                            // SEQ \n + \n QUAL \n (output-fastq.c:100-149); never masked
                if (q < len + 3) {
                    u64 skip = 0;
                    if (q < len) { is_bases = true; g = base + q; n_main = len - q; } else skip = q - len;
                    tc = 0x0A2B0Au >> (8 * (u32)skip); n_tc = 3 - (u32)skip;
                } else { src = P.qual + base + (q - len - 3); n_main = 2 * len + 3 - q; tc = '\n'; n_tc = 1; }
            } else {
                tc = '\n'; n_tc = 1;
                if (P.mode == EM_SEQUENCES || P.L == 0) { if (q < len) { is_bases = true; g = base + q; n_main = len - q; } }
                else {
                    u64 line, col;
                    if (Lp1_32 && (q >> 32) == 0) { u32 l32 = (u32)q / Lp1_32; line = l32; col = (u32)q - l32 * Lp1_32; }
                    else { line = q / (P.L + 1); col = q - line * (P.L + 1); }
                    u64 j = line * P.L + col;
                    if (col != P.L && j < len) { is_bases = true; g = base + j; n_main = len - j < P.L - col ? len - j : P.L - col; }
                }
            }
        }
        u32 rem = nbytes - pos;
        u32 n1 = n_main < rem ? (u32)n_main : rem;
        if (n1) {
            u64 slo, shi;
            if (is_bases) { bases16<FOURBIT>(P, g, slo, shi); if (P.mode != EM_FASTQ) mask16(P, klo, khi, any_toggle, g, slo, shi); }
            else if (!hdrs && off < hl) {                          // no header stream: byte-wise
                slo = shi = 0;
                for (u32 b = 0; b < n1; b++) { u64 ch = header_char(P, r, off + b, hl); if (b < 8) slo |= ch << (8 * b); else shi |= ch << (8 * (b - 8)); }
            }
            else { slo = ld64(src); shi = n1 > 8 ? ld64(src + 8) : 0; }
            place16(lo, hi, slo, shi, pos, n1);
            pos += n1; rem -= n1;
        }
        if (n1 == n_main && rem && n_tc) { u32 n2 = n_tc < rem ? n_tc : rem; place16(lo, hi, tc, 0, pos, n2); pos += n2; }
    }
    if (nbytes == 16) { uint4 v; v.x = (u32)lo; v.y = (u32)(lo >> 32); v.z = (u32)hi; v.w = (u32)(hi >> 32); memcpy(o, &v, 16); }
    else for (u32 b = 0; b < nbytes; b++) o[b] = (u8)((b < 8 ? lo >> (8 * b) : hi >> (8 * (b - 8))) & 0xFF);
}

struct GeoGlobal {
    const EmitP &P;
    __device__ GeoGlobal(const EmitP &p) : P(p) {}
    __device__ __forceinline__ void operator()(u64 r, u64 &ro, u64 &rn, u64 &len, u64 &base, u64 &ho, u32 &hl) const
    { ro = P.rec_out[r]; rn = P.rec_out[r + 1]; len = P.rec_len[r]; base = P.rec_base[r]; hl = P.hdr_len[r]; ho = P.hdr_text ? P.hdr_off[r - P.hdr_r0] : 0; }
};

template <bool FOURBIT>
__global__ __launch_bounds__(256) void k_emit_short(EmitP P, u8 *out)
{
    __shared__ u64 sh[4];
    __shared__ u64 s_ro[ES_CAP + 1], s_len[ES_CAP], s_base[ES_CAP], s_ho[ES_CAP];
    __shared__ u32 s_hl[ES_CAP];
    u64 span_first = P.out_begin + (u64)blockIdx.x * ES_SPAN, span_last = span_first + ES_SPAN - 1;
    if (span_last >= P.out_end) span_last = P.out_end - 1;
    span_windows(P, span_first, span_last, sh);
    const u64 rlo = sh[0], rhi = sh[1], klo = sh[2], khi = sh[3];
    const bool any_toggle = P.masking && klo < khi;
    const u64 nrec = rhi - rlo + 1;
    const bool in_lds = nrec <= ES_CAP;
    const bool hdrs = P.hdr_text != nullptr;
    if (in_lds) {
        for (u32 i = threadIdx.x; i < (u32)nrec; i += 256) {
            u64 r = rlo + i; s_ro[i] = P.rec_out[r]; s_len[i] = P.rec_len[r]; s_base[i] = P.rec_base[r]; s_hl[i] = P.hdr_len[r];
            s_ho[i] = hdrs ? P.hdr_off[r - P.hdr_r0] : 0;
        }
        if (threadIdx.x == 0) s_ro[nrec] = P.rec_out[rhi + 1];
        __syncthreads();
    }
    const u32 Lp1_32 = (P.L + 1) >> 32 ? 0 : (u32)(P.L + 1);
    auto geo_lds = [&](u64 r, u64 &ro, u64 &rn, u64 &len, u64 &base, u64 &ho, u32 &hl) {
        u32 i = (u32)(r - rlo); ro = s_ro[i]; rn = s_ro[i + 1]; len = s_len[i]; base = s_base[i]; hl = s_hl[i]; ho = s_ho[i];
    };
    GeoGlobal geo_g(P);
    for (u64 tile = span_first; tile <= span_last; tile += 4096) {
        u64 p0 = tile + (u64)threadIdx.x * 16;
        if (p0 > span_last) continue;
        u32 nbytes = P.out_end - p0 < 16 ? (u32)(P.out_end - p0) : 16;
        u8 *o = out + (p0 - P.out_begin);
        if (in_lds) {
            u32 lo_ = 0, hi_ = (u32)nrec; while (lo_ < hi_) { u32 mid = (lo_ + hi_) >> 1; if (s_ro[mid] <= p0) lo_ = mid + 1; else hi_ = mid; }
            compose_chunk<FOURBIT>(P, geo_lds, p0, nbytes, rlo + lo_ - 1, klo, khi, any_toggle, Lp1_32, o);
        } else {
            u64 r = upper_bound_u64(P.rec_out, rlo, rhi + 1, p0) - 1;
            compose_chunk<FOURBIT>(P, geo_g, p0, nbytes, r, klo, khi, any_toggle, Lp1_32, o);
        }
    }
}

// ---- whole FASTQ text, record-parallel --------------------------------------------------------------------------------------
// A read is five pieces at known offsets ('@' name '\n' | bases | "\n+\n" | qualities | '\n', output-fastq.c:100-149), so
// when the whole text is wanted there is nothing to search: ER_LANES lanes take one read and copy / expand its pieces 16 bytes per
// lane per step.  About 0.1 instructions per output byte against 37 for the chunk-composing kernel above, which stays for
// byte-range calls and FASTA.
#define ER_STAGE 24576u
// ER_LANES lanes per read, 256 / ER_LANES reads per workgroup: a workgroup's time is two rounds of dependent loads whatever it moves, so
// short reads go many to a workgroup (150-base reads, 4 GB of text: 3.8 ms with 16 lanes per read, 2.9 with 8, 2.4 with 4); the host picks
// the widest grouping whose reads still fit the LDS stage on average.
// WGT: threads per workgroup -- 64 (a wavefront on its own: its slot and its LDS come and go with its own reads; 4 GB of 150-base reads:
// 2.33 -> 2.19 ms; NAF_GPU_EMIT_WAVE=0: 256)
template <bool FOURBIT, u32 ER_LANES, u32 WGT>
__global__ __launch_bounds__(WGT) void k_emit_fastq_records(EmitP P, u8 *out)
{
    constexpr u32 ER_READS = WGT / ER_LANES, STAGE = ER_STAGE / (256u / WGT);
    // the reads of a workgroup are one contiguous piece of text: assembled in LDS (their pieces start at arbitrary byte
    // offsets) and written out as aligned 16-byte stores; only when it does not fit (long reads) the pieces go straight to HBM
    __shared__ __attribute__((aligned(16))) u8 stage[STAGE + 32];
    const u32 g = threadIdx.x & (ER_LANES - 1);
    const u64 r0 = (u64)blockIdx.x * ER_READS, r = r0 + threadIdx.x / ER_LANES;
    const u64 rend = r0 + ER_READS < P.N ? r0 + ER_READS : P.N;
    const u64 wbase = P.rec_out[r0], wspan = P.rec_out[rend] - wbase;
    const bool in_lds = wspan <= STAGE;
    if (r < P.N) {
        const u64 len = P.rec_len[r], base = P.rec_base[r];
        const u32 hl = P.hdr_len[r];
        u8 *o = in_lds ? stage + (P.rec_out[r] - wbase) : out + P.rec_out[r];
        // '@' name '\n' with name = id [sep comment] (output.c:105-124)
        u64 ids0 = 0, idl = 0, nm0 = 0, nml = 0;
        if (P.has_ids) { ids0 = r ? P.idz[r - 1] + 1 : 0; idl = P.idz[r] - ids0; }
        if (P.has_names) { nm0 = r ? P.nmz[r - 1] + 1 : 0; nml = P.nmz[r] - nm0; }
        if (g == 0) { o[0] = P.hdr_char; o[hl - 1] = '\n'; }
        if (P.has_ids) {
            group_copy<ER_LANES>(o + 1, P.ids + ids0, idl, g);
            if (P.has_names && nml) { if (g == 0) o[1 + idl] = P.sep; group_copy<ER_LANES>(o + 2 + idl, P.names + nm0, nml, g); }
        } else group_copy<ER_LANES>(o + 1, P.names + nm0, nml, g);
        // bases, upper case always (unnaf.c:442: FASTQ output ignores the mask)
        u8 *os = o + hl;
        for (u64 i = (u64)g * 16; i < len; i += 16 * ER_LANES) {
            u64 lo, hi; bases16<FOURBIT>(P, base + i, lo, hi);
            store_upto16(os + i, lo, hi, len - i < 16 ? (u32)(len - i) : 16u);
        }
        if (g == 0) { os[len] = '\n'; os[len + 1] = '+'; os[len + 2] = '\n'; os[2 * len + 3] = '\n'; }
        group_copy<ER_LANES>(os + len + 3, P.qual + base, len, g);
    }
    if (!in_lds) return;
    __syncthreads();
    u8 *dst = out + wbase; const u32 n = (u32)wspan;
    u32 head = (u32)((16 - ((uintptr_t)dst & 15)) & 15); if (head > n) head = n;
    if (threadIdx.x < head) dst[threadIdx.x] = stage[threadIdx.x];
    u32 words = (n - head) >> 4;
    for (u32 w = threadIdx.x; w < words; w += WGT) {
        u64 a, b2; __builtin_memcpy(&a, stage + head + 16 * w, 8); __builtin_memcpy(&b2, stage + head + 16 * w + 8, 8);
        uint4 v; v.x = (u32)a; v.y = (u32)(a >> 32); v.z = (u32)b2; v.w = (u32)(b2 >> 32);
        st_text16(dst + head + 16 * w, v, P.nt_store);
    }
    for (u32 k = head + 16 * words + threadIdx.x; k < n; k += WGT) dst[k] = stage[k];
}

// ---- long-record emit: one 4 KiB tile per workgroup, one 16-byte chunk per lane ---------------------------------------
// A plain "read 8 B, write 16 B" kernel of this shape moves 15 GB in 2.4 ms on MI355X, and leaves about 110 vector
// instructions per wavefront before the ALUs become the limit; so everything that is the same for the whole tile
// (record, its geometry, the line/column of the tile's first byte, the mask toggles that can fall inside) is looked
// up once per tile by k_tile_index and read here through scalar loads.
#define EMIT_TOG_LDS 256u
struct TileIdx { u64 gline, k, khi; u32 col, fast; };   // gline: base index of the first base of the tile's first line (wrap) / first byte
#define TI_HDR 0xFFFFFFFFu
// flat frames: the stream that holds the tile's first packed byte qf (first symbol q0, end-of-data bit address A) and the next
// stream that has symbols (q1, A1; it ends at q2) -- a tile that would reach a third stream goes to the slow list
struct TileFlat { u64 q0, A, q1, A1, q2, qf; };
// What k_emit_tile_flat needs of a tile, worked out once by k_tile_index and stored as the tile's flat record (48 bytes): the
// emit kernel ran this 64-bit arithmetic on the scalar unit of every one of its 2.4 M x 4 waves, ran out of scalar registers on it
// (spills through v_readlane / v_writelane) and paid VALU compares for the 64-bit "less than" the scalar unit lacks.
//   base     source address of bit `ub`, a multiple of eight bits that lies 4 * QM bits below the lower of the tile's two streams
//   K0, K1   bit offset from ub of (top - 40) for a chunk whose first packed byte is the tile's first one (qrel = 0), in the first /
//            second stream; a lane's offset is K - 4 * qrel >= 0
//   d1, d2   symbols from the tile's first packed byte to the end of the first / second stream
//   qoff     (a.gline >> 1) - qf, par0 = a.gline & 1
//   a1_addr, a1_sh   the eight bytes holding the top 36 bits of the second stream, and the shift that brings them down
struct TileFlatE { u64 base, a1_addr; u32 a1_sh, K0, K1, d1, d2, qoff, par0, pad; };
static_assert(sizeof(TileFlatE) == sizeof(TileFlat), "the derived record replaces the raw one in place");
#define FLAT_QM 4096
// One pass makes a tile's record, classifies it and derives what the flat kernel needs (until round 3: k_tile_index wrote raw records,
// k_tile_classify read them and their neighbours back -- 2.4 M tiles x 80 bytes written, read and written again were most of the two
// kernels' 164 us in front of a 10 GB emit).  A tile's class needs two values of the tile BEHIND it (its record number and its first
// toggle): a workgroup does 255 tiles and its last thread the next workgroup's first one once more, handing it over through LDS.
//   ti[t], tr[t] for t <= ntiles; ti[t].fast = 0 and a harmless flat record for the `spare` entries behind (the flat kernel's workgroups
//   read whole groups of records).
// fast = 1: the whole tile lies in the body of one record (and the next tile starts in the same record, so that the record's final
// newline is not in it) -- tile kernel of the launch (k_emit_tile, or k_emit_tile_flat reading the frame in place); 2: such a tile of a
// mostly-flat frame over blocks that were decoded -- k_emit_tile_list takes those from `list2`; 0: k_emit_rest (`list`).
#define TILE_WG 255
__global__ __launch_bounds__(256) void k_tile_index(EmitP P, u64 ntiles, TileIdx *ti, u64 *tr, u32 *list, u32 *count, TileFlat *tsig, u32 spare, u32 *list2)
{
    __shared__ u64 s_k[256], s_r[256];
    const u64 t = (u64)blockIdx.x * TILE_WG + threadIdx.x;
    const u64 p = P.out_begin + t * 4096;
    TileIdx x; x.fast = 0; x.gline = 0; x.khi = 0; x.k = P.n_toggles; x.col = TI_HDR;
    TileFlat f; f.q0 = f.A = f.q1 = f.A1 = f.q2 = f.qf = 0;
    u64 r = ~0ull; u32 cbits = 1;
    const bool inside = t < ntiles && p < P.out_end;              // (t == ntiles and the spare entries: the record of "behind the text")
    if (inside) {
        u64 g;
        if (P.mode == EM_SEQ) { r = 0; x.col = 0; g = p; x.gline = p; }
        else {
            r = upper_bound_u64(P.rec_out, 0, P.N + 1, p) - 1;
            u64 off = p - P.rec_out[r], hl = P.hdr_len[r], len = P.rec_len[r], j = 0;
            if (off >= hl) {
                u64 q = off - hl;
                if (P.mode == EM_FASTA && P.L) {
                    u64 line = q / (P.L + 1), col = q - line * (P.L + 1);
                    j = line * P.L + (col < P.L ? col : P.L);
                    if (P.L < 0xFFFFFFF0ull) { x.gline = P.rec_base[r] + line * P.L; x.col = (u32)col; }
                } else if (P.mode != EM_FASTQ) { j = q; x.col = 0; x.gline = P.rec_base[r] + q; }
                if (j > len) j = len;
            }
            g = P.rec_base[r] + j;
        }
        x.k = P.masking ? upper_bound_u64(P.toggles, 0, P.n_toggles, g) : 0;
        if (tsig) {                                              // slot of the stream that holds the tile's first packed byte
            const FlatStream *si = (const FlatStream *)P.fsi;
            const u64 q = g >> 1;
            // streams of this build's frames hold 8192 symbols: slot q >> 13 is the answer or next to it; any other frame costs the
            // gallop a few steps more than the plain binary search took (20 dependent loads per tile either way before)
            u64 lo, hi, gs = q >> 13; if (gs >= P.fslots) gs = P.fslots - 1;
            if (si[gs].q0 <= q) { lo = gs; hi = gs + 1; u64 st = 1; while (hi < P.fslots && si[hi].q0 <= q) { lo = hi; st <<= 1; hi = lo + st; } if (hi > P.fslots) hi = P.fslots; }
            else { hi = gs; u64 st = 1; lo = gs - 1; while (lo > 0 && si[lo].q0 > q) { hi = lo; st <<= 1; lo = lo > st ? lo - st : 0; } }   // (gs >= 1: si[0].q0 = 0 <= q)
            while (hi - lo > 1) { const u64 mid = (lo + hi) >> 1; if (si[mid].q0 <= q) lo = mid; else hi = mid; }
            f.q0 = si[lo].q0; f.A = si[lo].A; f.qf = q;
            if (P.fcls) {
                // what the blocks under the tile's packed bytes [q, q + 2048 + 16] have in common (bit 0: readable in place, bit 1: decoded);
                // four slots per block: block b starts at si[4 b].q0
                u32 c = 3; u64 b = lo >> 2; const u64 nb = P.fslots >> 2, qe = q + 2048 + 16;
                for (u32 k = 0; k < 6 && b < nb && si[4 * b].q0 <= qe; k++, b++) c &= P.fcls[b];
                if (b < nb && si[4 * b].q0 <= qe) c = 0;              // more blocks than that (tiny ones): the slow way
                cbits = c;
            }
            u64 cs = lo + 1;                                          // next slot that has symbols (single-stream blocks leave three empty)
            while (cs < P.fslots && si[cs + 1].q0 == si[cs].q0) cs++;
            f.q1 = si[cs].q0; f.A1 = cs < P.fslots ? si[cs].A : 0;
            u64 ce = cs < P.fslots ? cs + 1 : cs;
            while (ce < P.fslots && si[ce + 1].q0 == si[ce].q0) ce++;
            f.q2 = si[ce].q0;
        }
    }
    s_k[threadIdx.x] = x.k; s_r[threadIdx.x] = r;
    __syncthreads();
    if (threadIdx.x == TILE_WG) return;                          // (the next workgroup's first tile: done for its k and r only)
    TileFlatE e; e.base = (u64)P.fsrc; e.a1_addr = (u64)P.fsrc; e.a1_sh = 4; e.K0 = e.K1 = 0; e.d1 = e.d2 = 0x7FFFFFFFu; e.qoff = 0; e.par0 = 0; e.pad = 0;
    if (t >= ntiles) {
        if (t == ntiles) { tr[t] = ~0ull; ti[t] = x; if (tsig && spare) *(TileFlatE *)&tsig[t] = e; }
        else if (tsig && t < ntiles + spare) { ti[t].fast = 0; *(TileFlatE *)&tsig[t] = e; }
        return;
    }
    tr[t] = r;
    if (!inside) { ti[t] = x; if (tsig) *(TileFlatE *)&tsig[t] = e; list[atomicAdd(count, 1u)] = (u32)t; return; }   // (cannot happen: ntiles covers the text)
    const u64 r_next = s_r[threadIdx.x + 1];
    x.khi = s_k[threadIdx.x + 1];
    const bool wrap = P.mode == EM_FASTA && P.L != 0;
    bool fast = x.col != TI_HDR && r == r_next && P.out_begin + (t + 1) * 4096 <= P.out_end && !P.force_slow && (!wrap || P.L >= 16);
    bool packed = false;
    if (tsig) {
        packed = fast && (cbits & 2);
        if (fast && !(cbits & 1)) fast = false;
        if (fast) {
            fast = f.qf + 2048 + 16 < f.q2 || f.q2 == f.q1;         // at most two streams under the tile (q2 == q1: the data end there)
            if (P.ftail && f.qf + 2048 + 16 >= P.ftail_q) fast = false;   // the bytes of a final Raw block are not in any stream
            if (f.q2 != f.q1 && (f.A1 > f.A ? f.A1 - f.A : f.A - f.A1) >= (1ull << 29)) fast = false;   // both streams are addressed from one 32-bit base
        }
        if (fast) {
            const u64 gline = x.gline;
            const u64 dd1 = f.q1 - f.qf, dd2 = f.q2 - f.qf;
            e.d1 = dd1 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (u32)dd1; e.d2 = dd2 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (u32)dd2;
            const u64 T0 = f.A - 4 * (f.qf - f.q0), T1 = f.A1 + 4 * (u64)e.d1;      // bit above the symbol at qrel = 0, in the first / second stream
            const bool two = f.q2 != f.q1;
            const u64 tmin = two && T1 < T0 ? T1 : T0;
            const u64 ub = (tmin - 40 - 4 * (u64)FLAT_QM) & ~7ull;   // may lie below the buffer: only ub + a lane's offset is ever used as an address
            e.base = (u64)P.fsrc + (u64)((i64)ub >> 3);
            e.K0 = (u32)(T0 - 40 - ub); e.K1 = two ? (u32)(T1 - 40 - ub) : e.K0;
            e.qoff = (u32)((gline >> 1) - f.qf); e.par0 = (u32)gline & 1u;
            const u64 t1 = (f.A1 < 40 ? 40 : f.A1) - 40;
            e.a1_addr = (u64)P.fsrc + (t1 >> 3); e.a1_sh = (u32)(t1 & 7) + 4;
        }
        *(TileFlatE *)&tsig[t] = e;
    }
    x.fast = fast ? 1u : (packed ? 2u : 0u);
    ti[t] = x;
    if (!fast) { if (packed) list2[atomicAdd(count + 1, 1u)] = (u32)t; else list[atomicAdd(count, 1u)] = (u32)t; }
}

template <bool FOURBIT>
__device__ __forceinline__ void emit_tile_body(const EmitP &P, const TileIdx &a, u8 *out_tile)
{
    // soft-masked genomes put a dozen toggles into every tile: the tile's window goes to LDS once instead of every lane
    // walking it in global memory
    __shared__ u64 s_tog[EMIT_TOG_LDS];
    const u32 ntog = (u32)(a.khi - a.k < EMIT_TOG_LDS ? a.khi - a.k : EMIT_TOG_LDS);
    const bool use_tog = P.masking && a.k < a.khi && a.khi - a.k <= EMIT_TOG_LDS;
    if (use_tog) { for (u32 i = threadIdx.x; i < ntog; i += 256) s_tog[i] = P.toggles[a.k + i]; __syncthreads(); }
    const u32 lane16 = threadIdx.x * 16;
    u64 g0; u32 nl_b = 64;
    if (P.mode == EM_FASTA && P.L != 0) {
        const u32 Lp1 = (u32)P.L + 1;
        u32 c = a.col + lane16, dl;
        if (Lp1 < 32768) dl = __umulhi(c, P.Ldiv_magic); else dl = c >= Lp1 ? 1u : 0u;   // c < Lp1 + 4096: exact (see unnaf_run)
        u32 col = c - dl * Lp1;
        g0 = a.gline + (u64)dl * (u32)P.L + col;
        u32 d = (u32)P.L - col;                                  // byte index of the line-end newline
        nl_b = d < 16 ? d : 64;
    } else g0 = a.gline + lane16;
    u64 lo, hi;
    if (FOURBIT) {
        // 16 nibbles from base g0: one 16-byte load from the 8-aligned address below, then a funnel shift
        u64 addr = (u64)P.seq + (g0 >> 1);
        const uint4 q = ldg_at<uint4>(addr & ~7ull);
        u64 q0 = (u64)q.x | ((u64)q.y << 32), q1 = (u64)q.z | ((u64)q.w << 32);
        u32 sh = (u32)(addr & 7) * 8 + (u32)(g0 & 1) * 4;
        expand16(P.lut, sh ? (q0 >> sh) | (q1 << (64 - sh)) : q0, lo, hi);
    } else bases16<false>(P, g0, lo, hi);
    if (use_tog) mask16_from(s_tog, ntog, a.k, g0, lo, hi);
    else mask16(P, a.k, a.khi, P.masking && a.k < a.khi, g0, lo, hi);
    if (nl_b < 16) splice_newline(lo, hi, (int)nl_b);
    uint4 v; v.x = (u32)lo; v.y = (u32)(lo >> 32); v.z = (u32)hi; v.w = (u32)(hi >> 32);
    st_text16(out_tile + lane16, v, P.nt_store);
}
template <bool FOURBIT>
__global__ __launch_bounds__(256) void k_emit_tile(EmitP P, const TileIdx *ti, u8 *out)
{
    const TileIdx a = ti[blockIdx.x];
    if (a.fast != 1) return;
    emit_tile_body<FOURBIT>(P, a, out + (u64)blockIdx.x * 4096);
}
// The same with a wavefront per TW tiles (workgroups of 64; a tile's four KiB as four rows of 64 chunks): every lane has 4 TW loads
// in flight behind one look at the tiles' records, waits for no other wavefront, and leaves its slot when its own stores are out
// (k_emit_tile_flat_wave has the measurement).  The last tile of the text is never fast (k_tile_index), the records behind it read
// fast = 0.
// tn[j]: the wavefront's tiles (~0: none); want: the class of tile taken (TileIdx.fast: 1 = k_emit_tile_wave, 2 = the list kernel's)
template <bool FOURBIT, u32 TW>
__device__ __forceinline__ void emit_tiles_wave(const EmitP &P, const TileIdx *ti, u8 *out, const u64 (&tn)[TW], u32 want, u64 (*s_tog)[EMIT_TOG_LDS])
{
    const u32 lane = threadIdx.x;
    TileIdx A[TW]; bool live[TW];
#pragma unroll
    for (u32 j = 0; j < TW; j++) { A[j] = ti[tn[j] == ~0ull ? 0 : tn[j]]; live[j] = tn[j] != ~0ull && A[j].fast == want; }
    u64 g0s[TW][4]; u32 nls[TW][4]; uint4 Q[TW][4];
#pragma unroll
    for (u32 j = 0; j < TW; j++) {
        const TileIdx &a = A[j];
#pragma unroll
        for (u32 r = 0; r < 4; r++) {
            const u32 lane16 = (r * 64 + lane) * 16;
            u64 g0; u32 nl_b = 64;
            if (P.mode == EM_FASTA && P.L != 0) {
                const u32 Lp1 = (u32)P.L + 1;
                u32 c = a.col + lane16, dl;
                if (Lp1 < 32768) dl = __umulhi(c, P.Ldiv_magic); else dl = c >= Lp1 ? 1u : 0u;
                u32 col = c - dl * Lp1;
                g0 = a.gline + (u64)dl * (u32)P.L + col;
                u32 d = (u32)P.L - col;
                nl_b = d < 16 ? d : 64;
            } else g0 = a.gline + lane16;
            g0s[j][r] = g0; nls[j][r] = nl_b;
            // (a tile that is not fast has no geometry to speak of, and the packed stream of a range call starts at its first base: no load)
            if (FOURBIT) Q[j][r] = live[j] ? ldg_at<uint4>(((u64)P.seq + (g0 >> 1)) & ~7ull) : make_uint4(0, 0, 0, 0);
        }
    }
    if (P.masking) {
#pragma unroll
        for (u32 j = 0; j < TW; j++) {
            const u64 nt = A[j].khi - A[j].k;
            for (u32 i = lane; i < EMIT_TOG_LDS; i += 64) s_tog[j][i] = (live[j] && i < nt && nt <= EMIT_TOG_LDS) ? P.toggles[A[j].k + i] : 0ull;
        }
        __syncthreads();
    }
#pragma unroll
    for (u32 j = 0; j < TW; j++) {
        if (!live[j]) continue;
        const TileIdx &a = A[j];
        const u32 ntog = (u32)(a.khi - a.k < EMIT_TOG_LDS ? a.khi - a.k : EMIT_TOG_LDS);
        const bool use_tog = P.masking && a.k < a.khi && a.khi - a.k <= EMIT_TOG_LDS;
#pragma unroll
        for (u32 r = 0; r < 4; r++) {
            const u64 g0 = g0s[j][r];
            u64 lo, hi;
            if (FOURBIT) {
                const u64 addr = (u64)P.seq + (g0 >> 1);
                const uint4 q = Q[j][r];
                const u64 q0 = (u64)q.x | ((u64)q.y << 32), q1 = (u64)q.z | ((u64)q.w << 32);
                const u32 sh = (u32)(addr & 7) * 8 + (u32)(g0 & 1) * 4;
                expand16(P.lut, sh ? (q0 >> sh) | (q1 << (64 - sh)) : q0, lo, hi);
            } else bases16<false>(P, g0, lo, hi);
            if (use_tog) mask16_from(s_tog[j], ntog, a.k, g0, lo, hi);
            else mask16(P, a.k, a.khi, P.masking && a.k < a.khi, g0, lo, hi);
            if (nls[j][r] < 16) splice_newline(lo, hi, (int)nls[j][r]);
            uint4 v; v.x = (u32)lo; v.y = (u32)(lo >> 32); v.z = (u32)hi; v.w = (u32)(hi >> 32);
            st_text16(out + tn[j] * 4096 + (r * 64 + lane) * 16, v, P.nt_store);
        }
    }
}
template <bool FOURBIT, u32 TW>
__global__ __launch_bounds__(64) void k_emit_tile_wave(EmitP P, const TileIdx *ti, u8 *out, u64 ntiles)
{
    __shared__ u64 s_tog[TW][EMIT_TOG_LDS];
    u64 tn[TW];
#pragma unroll
    for (u32 j = 0; j < TW; j++) { const u64 t = (u64)blockIdx.x * TW + j; tn[j] = t < ntiles ? t : ~0ull; }
    emit_tiles_wave<FOURBIT, TW>(P, ti, out, tn, 1u, s_tog);
}
// ... and the tiles of a list (k_emit_tile_list's), a wavefront per tile
template <bool FOURBIT>
__global__ __launch_bounds__(64) void k_emit_tile_list_wave(EmitP P, const TileIdx *ti, const u32 *list, const u32 *count, u8 *out)
{
    __shared__ u64 s_tog[1][EMIT_TOG_LDS];
    const u32 n = *count;
    for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
        const u64 tn[1] = { list[i] };
        emit_tiles_wave<FOURBIT, 1>(P, ti, out, tn, 2u, s_tog);
        __syncthreads();                                          // s_tog is written again
    }
}
// the tiles of a list (its length stays on the device): the decoded stretches of a mostly-flat frame
template <bool FOURBIT>
__global__ __launch_bounds__(256) void k_emit_tile_list(EmitP P, const TileIdx *ti, const u32 *list, const u32 *count, u8 *out)
{
    const u32 n = *count;
    for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
        const u32 t = list[i];
        const TileIdx a = ti[t];
        emit_tile_body<FOURBIT>(P, a, out + (u64)t * 4096);
        __syncthreads();                                          // s_tog is reused by the next tile
    }
}

// The same tile kernel for a flat frame read in place: a lane's 16 bases are 8 (9 when its first base is an odd one) consecutive
// 4-bit codes of one Huffman stream, i.e. 32 (36) bits at a bit position that follows from the stream table -- one unaligned 8-byte
// load from the COMPRESSED stream, a funnel shift, four look-ups in a 256-entry table "two codes -> four characters" (built per
// workgroup from the frame's code -> packed byte list and the nucleotide table).  Consecutive lanes read consecutive (descending)
// words of the stream; a tile touches one stream, or two where one ends.  Chunks that straddle a stream's end, or sit in its
// last 16 symbols, take the symbols one by one.
__global__ void k_flat_pair(EmitP P, u32 *pair)               // sixteen-entry tables, each as four dwords (for v_perm_b32)
{
    // [0..3] code -> packed byte; [4..7] code -> character of the byte's first base (low nibble, encoders.c:44-57); [8..11] -> of its second
    const u32 t = threadIdx.x;
    // [16 .. 272): two codes -> their four characters; [272 .. 400): the line-end splice table, 32 x uint4 (k_emit_tile_flat_wave copies
    // both into its LDS: a table made by 64 lanes per tile would cost four times what the 256 of k_emit_tile_flat spend on it)
    {
        const u32 pa = P.fsym[t >> 4], pb = P.fsym[t & 15];
        pair[16 + t] = expand_codes4(P.lut, (pa & 15u) | ((pa >> 4) << 8) | ((pb & 15u) << 16) | ((pb >> 4) << 24));
        if (t < 16) {
            const u32 d = t; u32 sel[4], orv[4];
            for (u32 i = 0; i < 4; i++) {
                sel[i] = 0; orv[i] = 0;
                for (u32 j = 0; j < 4; j++) { const u32 q = 4 * i + j; sel[i] |= (q < d ? 4 + j : (q == d ? 0x0Cu : 3 + j)) << (8 * j); if (q == d) orv[i] |= 0x0Au << (8 * j); }
            }
            for (u32 i = 0; i < 4; i++) { pair[272 + 8 * d + i] = sel[i]; pair[272 + 8 * d + 4 + i] = orv[i]; }
        }
    }
    if (t >= 12) return;
    const u8 *lut = (const u8 *)P.lut;
    u32 v = 0;
    for (u32 k = 0; k < 4; k++) {
        const u32 pk = P.fsym[4 * (t & 3) + k];
        const u32 b = t < 4 ? pk : (t < 8 ? lut[pk & 15] : lut[pk >> 4]);
        v |= b << (8 * k);
    }
    pair[t] = v;
}
#define FLAT_TPW 8                                               // tiles per workgroup: a tile alone is three dependent loads and a store, i.e. pure latency
template <u32 TPW>
__global__ __launch_bounds__(256) void k_emit_tile_flat(EmitP P, const TileIdx *ti, const TileFlat *tsig, u8 *out, u64 ntiles, u32 xcd_chunk)
{
    // Workgroup b runs on XCD b % 8 (observed, MI355X_MICROARCH.md): with xcd_chunk = ceil(grid / 8) every XCD walks one contiguous
    // eighth of the text instead of every eighth 16 KiB piece of all of it -- no data is shared between workgroups, but an XCD's
    // address translation then covers an eighth of the pages in flight.  xcd_chunk = 0: workgroups in launch order.
    u32 wg = blockIdx.x;
    if (xcd_chunk) { wg = (blockIdx.x & 7u) * xcd_chunk + (blockIdx.x >> 3); if ((u64)wg * TPW >= ntiles) return; }
    __shared__ u64 s_tog[TPW][EMIT_TOG_LDS];                  // every tile's window of mask toggles (a dozen per tile of a soft-masked genome)
    // a line end inside a chunk: bytes in front of it stay, the byte at it becomes '\n', bytes behind it take the byte in front of
    // them -- per position d of the line end, sixteen v_perm_b32 selector bytes (source = this dword and the one below it; 0x0C
    // selects a zero byte) and the sixteen bytes to OR in
    __shared__ uint4 s_spl[32];
    if (threadIdx.x < 16) {
        const u32 d = threadIdx.x; u32 sel[4], orv[4];
        for (u32 i = 0; i < 4; i++) {
            sel[i] = 0; orv[i] = 0;
            for (u32 j = 0; j < 4; j++) { const u32 p = 4 * i + j; sel[i] |= (p < d ? 4 + j : (p == d ? 0x0Cu : 3 + j)) << (8 * j); if (p == d) orv[i] |= 0x0Au << (8 * j); }
        }
        s_spl[2 * d] = make_uint4(sel[0], sel[1], sel[2], sel[3]); s_spl[2 * d + 1] = make_uint4(orv[0], orv[1], orv[2], orv[3]);
    }
    // two codes -> their four characters: entry (a << 4 | b) holds first and second base of code a's byte, then of code b's.  The
    // look-up costs one LDS read per two codes; sixteen-entry tables through v_perm_b32 cost a dozen instructions for the same, and
    // this kernel is bound by its vector instructions.
    __shared__ u32 s_pair[256];
    {
        const u32 pa = P.fsym[threadIdx.x >> 4], pb = P.fsym[threadIdx.x & 15];
        s_pair[threadIdx.x] = expand_codes4(P.lut, (pa & 15u) | ((pa >> 4) << 8) | ((pb & 15u) << 16) | ((pb >> 4) << 24));
    }
    const u32 lane16 = threadIdx.x * 16;
    // ---- phase 0: the records of the workgroup's tiles (uniform addresses: scalar loads, all of them in flight together -- one
    // tile after the other, each waiting for its own records and then for its own codes, was four memory latencies in a row)
    TileIdx A[TPW]; TileFlatE F[TPW]; bool live[TPW];
    {
        // (both arrays have TPW spare entries behind ntiles, made harmless by k_tile_index; the workgroup's four records are
        // read as whole 16-byte words so that no field waits for a test on another one)
        static_assert(sizeof(TileIdx) == 32 && sizeof(TileFlatE) == 48, "records are read as 16-byte words");
        const u64 t0 = (u64)wg * TPW;
        const uint4 *pa = (const uint4 *)(ti + t0), *pf = (const uint4 *)(tsig + t0);
        uint4 ra[2 * TPW], rf[3 * TPW];
#pragma unroll
        for (u32 i = 0; i < 2 * TPW; i++) ra[i] = pa[i];
#pragma unroll
        for (u32 i = 0; i < 3 * TPW; i++) rf[i] = pf[i];
#pragma unroll
        for (u32 j = 0; j < TPW; j++) {
            __builtin_memcpy(&A[j], &ra[2 * j], 32); __builtin_memcpy(&F[j], &rf[3 * j], 48);
            live[j] = t0 + j < ntiles && A[j].fast == 1;
        }
    }
    // ---- phase 1: where every chunk's codes are, and the loads.  A chunk needs the 36 bits under `top` (nine 4-bit codes; a bit
    // address inside the source buffer): the eight bytes at (top - 40) >> 3 hold them.  Everything here is 32-bit: bases count from
    // the tile's a.gline, packed bytes from its first one, bit addresses from a tile-wide base (TileFlatE, worked out by
    // k_tile_index); the load is scalar base + 32-bit lane offset.
    // No branches here: a value that is only loaded on one path needs a copy where the paths join, and that copy waits for the load.
    // A tile that is not live (it goes to k_emit_rest, or lies behind the last one) has a record that points at the source's first bytes.
    u64 X[TPW]; u32 offs[TPW], grels[TPW], nls[TPW], haves[TPW];
#pragma unroll
    for (u32 j = 0; j < TPW; j++) {
        const TileIdx &a = A[j]; const TileFlatE &e = F[j];
        u32 grel, nl_b = 64;                                          // grel: the chunk's first base, counted from a.gline
        if (P.mode == EM_FASTA && P.L != 0) {
            const u32 Lp1 = (u32)P.L + 1;
            u32 c = a.col + lane16, dl;
            if (Lp1 < 32768) dl = __umulhi(c, P.Ldiv_magic); else dl = c >= Lp1 ? 1u : 0u;
            u32 col = c - dl * Lp1;
            grel = dl * (u32)P.L + col;
            u32 d = (u32)P.L - col;
            nl_b = d < 16 ? d : 64;
        } else grel = lane16;
        grels[j] = grel; nls[j] = nl_b;
        const u32 par = e.par0 + grel;                                // parity of the chunk's first base in bit 0
        const u32 need = 8 + (par & 1u);
        const u32 qrel = (par >> 1) + e.qoff;                         // its first packed byte, counted from the tile's first one
        const bool second = qrel >= e.d1;                             // the chunk starts in the tile's second stream
        const u32 qm = live[j] ? ~0u : 0u;                            // (a dead tile's geometry may be anything: its lanes all read offset K = 0)
        const u32 off = (second ? e.K1 : e.K0) - ((4 * qrel) & qm);   // bit offset of (top - 40) from the tile's base
        const u32 rem = (second ? e.d2 : e.d1) - qrel;                // symbols from the chunk's first one to the end of its stream
        X[j] = ldg_at_unaligned<u64>(e.base + (off >> 3));
        offs[j] = off;
        haves[j] = need > rem ? rem : 16u;                            // it runs over the end of that stream: the rest is the top of the next one
    }
    // the toggles of the four tiles' windows: fetched with the codes (after them: a wait in the middle of phase 2, tile after tile, was
    // half a millisecond per 4 GB of a soft-masked genome) and parked in LDS before the barrier that s_spl needs anyway
    if (P.masking) {
        u64 tg[TPW];
#pragma unroll
        for (u32 j = 0; j < TPW; j++) {
            const u64 nt = A[j].khi - A[j].k;
            tg[j] = (live[j] && threadIdx.x < nt && nt <= EMIT_TOG_LDS) ? P.toggles[A[j].k + threadIdx.x] : 0ull;
        }
#pragma unroll
        for (u32 j = 0; j < TPW; j++) s_tog[j][threadIdx.x] = tg[j];
    }
    __syncthreads();                                              // s_spl, s_tog (the loads are in flight meanwhile)
    // ---- phase 2: codes -> characters, mask, line ends, store
#pragma unroll
    for (u32 j = 0; j < TPW; j++) {
        const u64 t = (u64)wg * TPW + j;
        if (!live[j]) continue;
        const TileIdx &a = A[j];
        const u64 g0 = a.gline + grels[j];
        u64 h36 = X[j] >> ((offs[j] & 7) + 4);                        // nine codes: symbol k in bits 35..32, ... symbol k+8 in bits 3..0
        if (haves[j] < 9) {                                           // the stream ends inside the chunk: the rest is the top of the tile's second stream
            const u64 y = ldg_at_unaligned<u64>(F[j].a1_addr);
            const u64 n36 = (y >> F[j].a1_sh) & 0xFFFFFFFFFull;
            const u32 keep = 4 * haves[j];                             // bits of this stream's symbols
            h36 = (h36 & ~(0xFFFFFFFFFull >> keep)) | (n36 >> keep);
        }
        const u32 h = (u32)(h36 >> 4), n9 = (u32)h36 & 15u;
        u64 lo, hi;
        {
            // codes -> characters without the packed byte in between: the top byte of h is symbols k and k+1, and so on down
            u32 w0 = s_pair[h >> 24], w1 = s_pair[(h >> 16) & 0xFFu], w2 = s_pair[(h >> 8) & 0xFFu], w3 = s_pair[h & 0xFFu];
            if (g0 & 1) {                                             // the chunk starts at a byte's second base: one character down, the ninth symbol's first on top
                const u32 c9 = s_pair[n9 << 4];
                w0 = __builtin_amdgcn_alignbyte(w1, w0, 1); w1 = __builtin_amdgcn_alignbyte(w2, w1, 1);
                w2 = __builtin_amdgcn_alignbyte(w3, w2, 1); w3 = __builtin_amdgcn_alignbyte(c9, w3, 1);
            }
            lo = (u64)w0 | ((u64)w1 << 32); hi = (u64)w2 | ((u64)w3 << 32);
        }
        const u32 ntog = (u32)(a.khi - a.k < EMIT_TOG_LDS ? a.khi - a.k : EMIT_TOG_LDS);
        const bool use_tog = P.masking && a.k < a.khi && a.khi - a.k <= EMIT_TOG_LDS;
        if (use_tog) mask16_from(s_tog[j], ntog, a.k, g0, lo, hi);
        else mask16(P, a.k, a.khi, P.masking && a.k < a.khi, g0, lo, hi);
        uint4 v; v.x = (u32)lo; v.y = (u32)(lo >> 32); v.z = (u32)hi; v.w = (u32)(hi >> 32);
        if (nls[j] < 16) {
            const uint4 sel = s_spl[2 * nls[j]], orv = s_spl[2 * nls[j] + 1];
            const u32 x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
            v.x = __builtin_amdgcn_perm(x0, 0u, sel.x) | orv.x; v.y = __builtin_amdgcn_perm(x1, x0, sel.y) | orv.y;
            v.z = __builtin_amdgcn_perm(x2, x1, sel.z) | orv.z; v.w = __builtin_amdgcn_perm(x3, x2, sel.w) | orv.w;
        }
        st_text16(out + t * 4096 + lane16, v, P.nt_store);
    }
}

// The same with a WAVEFRONT per tile (workgroups of 64): TW tiles per wavefront, a tile's four KiB as four rows of 64 chunks.  A wavefront
// reads the records of its own tiles only (a quarter of the scalar loads per wavefront of the kernel above), waits for no other
// wavefront, and its slot is free again as soon as its own stores are out.  Tables from k_flat_pair's global copies.
template <u32 TW>
__global__ __launch_bounds__(64) void k_emit_tile_flat_wave(EmitP P, const TileIdx *ti, const TileFlat *tsig, u8 *out, u64 ntiles, u32 xcd_chunk)
{
    u32 wg = blockIdx.x;
    if (xcd_chunk) { wg = (blockIdx.x & 7u) * xcd_chunk + (blockIdx.x >> 3); if ((u64)wg * TW >= ntiles) return; }
    __shared__ u64 s_tog[TW][EMIT_TOG_LDS];
    __shared__ uint4 s_spl[32];
    __shared__ u32 s_pair[256];
    const u32 lane = threadIdx.x;
    if (lane < 32) s_spl[lane] = ((const uint4 *)(P.fpair + 272))[lane];
    ((uint4 *)s_pair)[lane] = ((const uint4 *)(P.fpair + 16))[lane];
    TileIdx A[TW]; TileFlatE F[TW]; bool live[TW];
    {
        const u64 t0 = (u64)wg * TW;
        const uint4 *pa = (const uint4 *)(ti + t0), *pf = (const uint4 *)(tsig + t0);
        uint4 ra[2 * TW], rf[3 * TW];
#pragma unroll
        for (u32 i = 0; i < 2 * TW; i++) ra[i] = pa[i];
#pragma unroll
        for (u32 i = 0; i < 3 * TW; i++) rf[i] = pf[i];
#pragma unroll
        for (u32 j = 0; j < TW; j++) {
            __builtin_memcpy(&A[j], &ra[2 * j], 32); __builtin_memcpy(&F[j], &rf[3 * j], 48);
            live[j] = t0 + j < ntiles && A[j].fast == 1;
        }
    }
    u64 X[TW][4]; u32 grels[TW][4], pk[TW][4];                  // pk: line-end position | symbols left in the stream << 8 | bit offset of the codes in their byte << 16
#pragma unroll
    for (u32 j = 0; j < TW; j++) {
        const TileIdx &a = A[j]; const TileFlatE &e = F[j];
#pragma unroll
        for (u32 r = 0; r < 4; r++) {
            const u32 lane16 = (r * 64 + lane) * 16;
            u32 grel, nl_b = 64;
            if (P.mode == EM_FASTA && P.L != 0) {
                const u32 Lp1 = (u32)P.L + 1;
                u32 c = a.col + lane16, dl;
                if (Lp1 < 32768) dl = __umulhi(c, P.Ldiv_magic); else dl = c >= Lp1 ? 1u : 0u;
                u32 col = c - dl * Lp1;
                grel = dl * (u32)P.L + col;
                u32 d = (u32)P.L - col;
                nl_b = d < 16 ? d : 64;
            } else grel = lane16;
            grels[j][r] = grel;
            const u32 par = e.par0 + grel;
            const u32 need = 8 + (par & 1u);
            const u32 qrel = (par >> 1) + e.qoff;
            const bool second = qrel >= e.d1;
            const u32 qm = live[j] ? ~0u : 0u;
            const u32 off = (second ? e.K1 : e.K0) - ((4 * qrel) & qm);
            const u32 rem = (second ? e.d2 : e.d1) - qrel;
            X[j][r] = ldg_at_unaligned<u64>(e.base + (off >> 3));
            pk[j][r] = nl_b | ((need > rem ? rem : 16u) << 8) | ((off & 7u) << 16);
        }
    }
    if (P.masking) {
#pragma unroll
        for (u32 j = 0; j < TW; j++) {
            const u64 nt = A[j].khi - A[j].k;
            for (u32 i = lane; i < EMIT_TOG_LDS; i += 64) s_tog[j][i] = (live[j] && i < nt && nt <= EMIT_TOG_LDS) ? P.toggles[A[j].k + i] : 0ull;
        }
    }
    __syncthreads();                                              // (one wavefront: the tables and toggles it has just written)
#pragma unroll
    for (u32 j = 0; j < TW; j++) {
        const u64 t = (u64)wg * TW + j;
        if (!live[j]) continue;
        const TileIdx &a = A[j];
        const u32 ntog = (u32)(a.khi - a.k < EMIT_TOG_LDS ? a.khi - a.k : EMIT_TOG_LDS);
        const bool use_tog = P.masking && a.k < a.khi && a.khi - a.k <= EMIT_TOG_LDS;
#pragma unroll
        for (u32 r = 0; r < 4; r++) {
            const u64 g0 = a.gline + grels[j][r];
            const u32 nl = pk[j][r] & 0xFFu, have = (pk[j][r] >> 8) & 0xFFu;
            u64 h36 = X[j][r] >> ((pk[j][r] >> 16) + 4);
            if (have < 9) {
                const u64 y = ldg_at_unaligned<u64>(F[j].a1_addr);
                const u64 n36 = (y >> F[j].a1_sh) & 0xFFFFFFFFFull;
                const u32 keep = 4 * have;
                h36 = (h36 & ~(0xFFFFFFFFFull >> keep)) | (n36 >> keep);
            }
            const u32 h = (u32)(h36 >> 4), n9 = (u32)h36 & 15u;
            u64 lo, hi;
            {
                u32 w0 = s_pair[h >> 24], w1 = s_pair[(h >> 16) & 0xFFu], w2 = s_pair[(h >> 8) & 0xFFu], w3 = s_pair[h & 0xFFu];
                if (g0 & 1) {
                    const u32 c9 = s_pair[n9 << 4];
                    w0 = __builtin_amdgcn_alignbyte(w1, w0, 1); w1 = __builtin_amdgcn_alignbyte(w2, w1, 1);
                    w2 = __builtin_amdgcn_alignbyte(w3, w2, 1); w3 = __builtin_amdgcn_alignbyte(c9, w3, 1);
                }
                lo = (u64)w0 | ((u64)w1 << 32); hi = (u64)w2 | ((u64)w3 << 32);
            }
            if (use_tog) mask16_from(s_tog[j], ntog, a.k, g0, lo, hi);
            else mask16(P, a.k, a.khi, P.masking && a.k < a.khi, g0, lo, hi);
            uint4 v; v.x = (u32)lo; v.y = (u32)(lo >> 32); v.z = (u32)hi; v.w = (u32)(hi >> 32);
            if (nl < 16) {
                const uint4 sel = s_spl[2 * nl], orv = s_spl[2 * nl + 1];
                const u32 x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
                v.x = __builtin_amdgcn_perm(x0, 0u, sel.x) | orv.x; v.y = __builtin_amdgcn_perm(x1, x0, sel.y) | orv.y;
                v.z = __builtin_amdgcn_perm(x2, x1, sel.z) | orv.z; v.w = __builtin_amdgcn_perm(x3, x2, sel.w) | orv.w;
            }
            st_text16(out + t * 4096 + (r * 64 + lane) * 16, v, P.nt_store);
        }
    }
}

template <bool FOURBIT>
__device__ __forceinline__ void emit_rest_tile(const EmitP &P, const TileIdx *ti, const u64 *tr, u64 t, u8 *out, u32 chunk)
{
    const TileIdx a = ti[t];
    u64 p0 = P.out_begin + t * 4096 + chunk * 16;
    if (p0 >= P.out_end) return;
    u32 nbytes = P.out_end - p0 < 16 ? (u32)(P.out_end - p0) : 16;
    const u32 Lp1_32 = (P.L + 1) >> 32 ? 0 : (u32)(P.L + 1);
    u8 *o = out + (p0 - P.out_begin);
    if (P.mode == EM_SEQ) {                                      // tail of a --seq range: byte-wise
        for (u32 bb = 0; bb < nbytes; bb++) o[bb] = (u8)emit_byte<FOURBIT>(P, p0 + bb, 0, 0, a.k, a.khi);
        return;
    }
    u64 rlo = tr[t], rhi = tr[t + 1] == ~0ull ? P.N - 1 : tr[t + 1];
    u64 r = upper_bound_u64(P.rec_out, rlo, rhi + 1, p0) - 1;
    GeoGlobal geo_g(P);
    compose_chunk<FOURBIT>(P, geo_g, p0, nbytes, r, a.k, a.khi, P.masking && a.k < a.khi, Lp1_32, o);
}
// The list's length stays on the device: the grid is fixed (EMIT_REST_GRID workgroups stride over the list), so the host queues this
// kernel without reading the count back -- a synchronisation in front of the main emit launch otherwise.
#define EMIT_REST_GRID 4096
template <bool FOURBIT>
__global__ __launch_bounds__(64) void k_emit_rest(EmitP P, const TileIdx *ti, const u64 *tr, const u32 *list, const u32 *count, u8 *out)
{
    // workgroups of one wavefront, a quarter of a tile each (beside k_emit_tile_flat_wave a workgroup of 256 waits for four wave slots of
    // one CU to be free at once: 0.28 -> 1.4 ms for the hundred tiles of the headline text)
    const u32 n = *count;
    for (u32 i = blockIdx.x; i < 4 * n; i += gridDim.x) emit_rest_tile<FOURBIT>(P, ti, tr, list[i >> 2], out, (i & 3) * 64 + threadIdx.x);
}

// Base-index range [g_lo, g_hi) that output bytes [out_begin, out_end) can touch (conservative on both sides).
__global__ void k_range_bases(EmitP P, u64 *out2)
{
    if (threadIdx.x || blockIdx.x) return;
    if (P.mode == EM_SEQ) { out2[0] = P.out_begin; out2[1] = P.out_end; out2[2] = out2[3] = 0; return; }
    u64 pb = P.out_begin, pe = P.out_end - 1;
    u64 r0 = upper_bound_u64(P.rec_out, 0, P.N + 1, pb) - 1, r1 = upper_bound_u64(P.rec_out, 0, P.N + 1, pe) - 1;
    out2[2] = r0; out2[3] = r1;
    auto lower = [&](u64 r, u64 p) -> u64 {                      // bases of record r before text byte p
        u64 off = p - P.rec_out[r], hl = P.hdr_len[r], len = P.rec_len[r];
        if (off <= hl) return 0;
        u64 q = off - hl, j;
        if (P.mode == EM_FASTQ) j = 0;
        else if (P.mode == EM_FASTA && P.L) j = (q / (P.L + 1)) * P.L;
        else j = q;
        return j > len ? len : j;
    };
    auto upper = [&](u64 r, u64 p) -> u64 {                      // one past the last base of record r at or before byte p
        u64 off = p - P.rec_out[r], hl = P.hdr_len[r], len = P.rec_len[r];
        if (off < hl) return 0;
        u64 q = off - hl, j;
        if (P.mode == EM_FASTQ) j = len;
        else if (P.mode == EM_FASTA && P.L) j = (q / (P.L + 1) + 1) * P.L;
        else j = q + 1;
        return j > len ? len : j;
    };
    out2[0] = P.rec_base[r0] + lower(r0, pb);
    out2[1] = P.rec_base[r1] + upper(r1, pe);
}

// ---- bases behind the last record ---------------------------------------------------------------------------------------
__global__ void k_last_nonempty(const u64 *rec_len, u64 N, unsigned long long *out)          // out[0] = 1 + index of the last record with bases
{
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < N && rec_len[r]) atomicMax(out, (unsigned long long)(r + 1));
}
// Text bytes [lo, hi) of the surplus (counted from its first byte): `cc` bases finish the line in progress, then lines of L, no
// newline behind the last one (print_dna_split_into_lines, output.c:339-360, carries cur_line_n_bp_remaining over from the last record).
template <bool FOURBIT>
__global__ void k_emit_surplus(EmitP P, u64 base0, u64 S, u64 cc, u64 L, int wrap, u64 lo, u64 hi, u8 *out)
{
    const u64 j = lo + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= hi) return;
    u64 g; bool nl = false;
    if (!wrap || L == 0 || S <= cc) g = base0 + j;
    else if (j < cc) g = base0 + j;
    else if (j == cc) { nl = true; g = 0; }
    else { const u64 j2 = j - cc - 1, line = j2 / (L + 1), col = j2 - line * (L + 1); if (col == L) { nl = true; g = 0; } else g = base0 + cc + line * L + col; }
    u32 ch = '\n';
    if (!nl) { ch = base_char<FOURBIT>(P, g); if (P.masking && base_masked(P, g, 0, P.n_toggles)) ch += 32; }
    out[j - lo] = (u8)ch;
}

// ---- host side ------------------------------------------------------------------------------------------------------
static int zero_positions(naf_gpu_ctx *c, const u8 *d_buf, u64 n, u64 N, u64 **out, bool names)
{
    u64 *pos = arena_new<u64>(c, N + 1);
    if (!pos) return NAF_GPU_ENOMEM;
    u64 tiles = (n + ZT_TILE - 1) / ZT_TILE;
    u64 *cnt = arena_new<u64>(c, tiles + 2);
    if (!cnt) return NAF_GPU_ENOMEM;
    u64 *d_total = cnt + tiles + 1;
    if (tiles) LAUNCH(c, "unnaf_zero_count", k_zero_count, tiles, 256, 0, d_buf, n, cnt);
    int rc = scan_exclusive_u64(c, cnt, tiles, d_total); if (rc) return rc;
    u64 total = 0;
    rc = ctx_readback(c, &total, d_total, 8); if (rc) return rc;
    if (total < N) return ctx_fail(c, NAF_GPU_EFORMAT, names ? "corrupted names - can't read name %llu\n" : "currupted ids - can't read id %llu\n", (unsigned long long)total);   // input.c:167,195 ("currupted" is the reference's spelling)
    if (tiles) LAUNCH(c, "unnaf_zero_scatter", k_zero_scatter, tiles, 256, 0, d_buf, n, (const u64 *)cnt, pos, N);
    *out = pos;
    return 0;
}

// ---- container framing (unnaf/src/input.c:31-77 read_header, utils.c:117-141 read_number, unnaf.c:402-404) ---------------------------
// One routine for both sides: the host calls it on an archive in host memory (the CLI, before the upload), and a one-thread kernel
// runs it on an archive in HBM -- the section headers form a chain (each one sits behind the previous section's payload), which
// from the host is one read-back per link; on the device it is one launch and one read-back of the finished structure.
// Returns 0 or one of the H_* codes below; H_VERSION / H_SEQTYPE leave the offending value in h->version / h->seq_type.
//
// Size fields a well-formed archive cannot hold: a zstd block regenerates at most 128 KiB from no less than 3 bytes, so a
// section of c compressed bytes decodes to fewer than (c + 4) * 2^16 bytes (twice that many bases for the 4-bit sequence
// stream); every record costs at least one byte of archive in the streams that describe it.  Fields beyond these bounds are a
// corrupted header, reported the way the reference reports a section it cannot decode (input.c:156,184,213,231) -- before any
// buffer is sized from them.  All sizes are compared in subtraction form: a field near 2^64 must not wrap a sum.
enum { H_OK = 0, H_EMPTY, H_TRUNC, H_MAGIC, H_VERSION, H_SEQTYPE, H_SEPARATOR, H_VLE_LEAD, H_VLE_OVERFLOW, H_SANE_SECTION /* + section */, H_SANE_COUNT = H_SANE_SECTION + 6 };
struct ContainerOut { naf_gpu_header h; int code; u8 frame_head[6][24]; };           // frame_head: the first bytes of every section's zstd frame (its header)
NAF_HD int naf_parse_container(const u8 *p, size_t len, naf_gpu_header *h)
{
    size_t pos = 0;
    memset(h, 0, sizeof *h);
    if (len == 0) return H_EMPTY;
    if (len < 3) return H_TRUNC;
    if (p[0] != 0x01 || p[1] != 0xF9 || p[2] != 0xEC) return H_MAGIC;
    pos = 3;
    if (pos >= len) return H_TRUNC;
    h->version = p[pos++];
    if (h->version < 1 || h->version > 2) return H_VERSION;
    h->seq_type = NAF_SEQ_DNA;
    if (h->version > 1) {
        if (pos >= len) return H_TRUNC;
        int t = p[pos++];
        h->seq_type = t;
        if (t < 1 || t > 3) return H_SEQTYPE;
    }
    if (len - pos < 2) return H_TRUNC;
    h->flags = p[pos++]; h->separator = p[pos++];
    if (h->separator < 0x20 || h->separator > 0x7E) return H_SEPARATOR;
    // the numbers in file order: line length, N, [title size], then (original size, compressed size) per present section
    int e = 0;
    auto rd = [&](u64 *v) -> int {
        u64 a = 0; if (pos >= len) return H_TRUNC;
        u8 ch = p[pos++];
        if (ch == 128) return H_VLE_LEAD;
        while (ch & 128) { if (a & (127ull << 57)) return H_VLE_OVERFLOW; a = (a << 7) | (ch & 127); if (pos >= len) return H_TRUNC; ch = p[pos++]; }
        if (a & (127ull << 57)) return H_VLE_OVERFLOW;
        *v = (a << 7) | ch; return 0;
    };
    if ((e = rd(&h->line_length)) || (e = rd(&h->n_sequences))) return e;
    if (h->flags & 0x40) { if ((e = rd(&h->title_len))) return e; h->title_off = pos; if (h->title_len > len - pos) return H_TRUNC; pos += h->title_len; }
    const int bit[6] = { 0x20, 0x10, 0x08, 0x04, 0x02, 0x01 };
    for (int i = 0; i < 6; i++) {
        if (!(h->flags & bit[i])) continue;
        if ((e = rd(&h->orig_size[i])) || (e = rd(&h->comp_size[i]))) return e;
        if (h->comp_size[i] > len - pos) return H_TRUNC;
        h->payload_off[i] = pos; pos += h->comp_size[i];
    }
    for (int i = 0; i < 6; i++) {
        const u64 cs = h->comp_size[i] + 4;                                   // comp_size <= len < 2^63 here
        const u64 lim = cs > (1ull << 46) ? ~0ull : cs << (i == 4 ? 17 : 16);
        if (h->orig_size[i] > lim) return H_SANE_SECTION + i;
    }
    if (h->n_sequences > ((u64)len << 16)) return H_SANE_COUNT;
    return H_OK;
}
static void container_error_text(int code, const naf_gpu_header *h, char *buf, size_t cap)
{
    static const char *what[6] = { "can't decompress ids\n", "can't decompress names\n", "can't decompress lengths\n", "can't decompress mask\n",
                                   "can't decompress sequence\n", "can't decompress quality\n" };
    switch (code) {
    case H_EMPTY: snprintf(buf, cap, "empty input"); break;
    case H_TRUNC: snprintf(buf, cap, "incomplete or truncated input\n"); break;
    case H_MAGIC: snprintf(buf, cap, "not a NAF format\n"); break;
    case H_VERSION: snprintf(buf, cap, "unknown version (%d) of NAF format\n", h->version); break;
    case H_SEQTYPE: snprintf(buf, cap, "unknown sequence type (%d) found in NAF file\n", h->seq_type); break;
    case H_SEPARATOR: snprintf(buf, cap, "unsupported name separator character\n"); break;
    case H_VLE_LEAD: snprintf(buf, cap, "invalid input: error parsing variable length encoded number\n"); break;
    case H_VLE_OVERFLOW: snprintf(buf, cap, "invalid input: overflow reading a variable length encoded number\n"); break;
    case H_SANE_COUNT: snprintf(buf, cap, "corrupted header: more sequences than the archive can describe\n"); break;
    default: if (code >= H_SANE_SECTION && code < H_SANE_SECTION + 6) snprintf(buf, cap, "%s", what[code - H_SANE_SECTION]); else snprintf(buf, cap, "malformed NAF header\n");
    }
}

extern "C" int naf_gpu_parse_header_host(const void *h_naf, size_t len, naf_gpu_header *h, char errbuf[128])
{
    if (!h) return NAF_GPU_EARG;
    const int code = naf_parse_container((const u8 *)h_naf, len, h);
    if (!code) return 0;
    if (errbuf) container_error_text(code, h, errbuf, 128);
    return NAF_GPU_EFORMAT;
}

__global__ void k_parse_container(const u8 *d, u64 len, ContainerOut *out)
{
    if (threadIdx.x || blockIdx.x) return;
    out->code = naf_parse_container(d, (size_t)len, &out->h);
    for (int i = 0; i < 6; i++) {
        const u64 n = out->code ? 0 : (out->h.comp_size[i] < 24 ? out->h.comp_size[i] : 24);
        for (u64 k = 0; k < 24; k++) out->frame_head[i][k] = k < n ? d[out->h.payload_off[i] + k] : 0;
    }
}

static int parse_container_device(naf_gpu_ctx *c, const void *d_naf, size_t len, ContainerOut *co)
{
    ContainerOut *d_co = arena_new<ContainerOut>(c, 1); if (!d_co) return NAF_GPU_ENOMEM;
    LAUNCH(c, "unnaf_parse_container", k_parse_container, 1, 64, 0, (const u8 *)d_naf, (u64)len, d_co);
    int rc = ctx_readback(c, co, d_co, sizeof *co); if (rc) return rc;
    if (co->code) { char buf[160]; container_error_text(co->code, &co->h, buf, sizeof buf); return ctx_fail(c, NAF_GPU_EFORMAT, "%s", buf); }
    return 0;
}

extern "C" int naf_gpu_parse_header(naf_gpu_ctx *c, const void *d_naf, size_t len, naf_gpu_header *h)
{
    if (!c || !h) return NAF_GPU_EARG;
    arena_reset(c);
    ContainerOut co;
    int rc = parse_container_device(c, d_naf, len, &co);
    *h = co.h;
    return rc;
}

enum { S_IDS = 0, S_NAMES, S_LEN, S_MASK, S_SEQ, S_QUAL };

struct UnnafPlan {
    naf_gpu_header h; u8 frame_head[6][24]; EmitP P; u64 total; bool fourbit; bool empty;
    u64 seq_bytes; bool need_qual;
    // bases behind the last record (an input with control bytes inside an ID): the reference prints them, wrapped on from
    // where the last non-empty record's last line stopped, without a final newline (output.c:369-430)
    u64 main_total, surplus, sur_c, sur_base0;
};

static int load_section(naf_gpu_ctx *c, const u8 *d_naf, const naf_gpu_header &h, int i, u64 expect, const char *what, u8 **out, const u8 *head = nullptr)
{
    u8 *buf = (u8 *)arena_alloc(c, expect + 32);
    if (!buf) return NAF_GPU_ENOMEM;
    size_t n = 0;
    int rc = zstd_decode(c, d_naf + h.payload_off[i], h.comp_size[i], 0, buf, expect, &n, head);
    if (rc == NAF_GPU_ECAP || (rc == 0 && n != expect)) return ctx_fail(c, NAF_GPU_EFORMAT, "can't decompress %s\n", what);   // input.c:156,184,213,231
    if (rc) return rc;
    *out = buf;
    return 0;
}

// Everything except the sequence / quality payload decode and the emit itself.
static int unnaf_prepare(naf_gpu_ctx *c, const u8 *d_naf, size_t naf_len, const naf_gpu_unnaf_opts *o, UnnafPlan &pl)
{
    ContainerOut co;
    int rc = parse_container_device(c, d_naf, naf_len, &co); if (rc) return rc;
    pl.h = co.h; memcpy(pl.frame_head, co.frame_head, sizeof pl.frame_head);
    const naf_gpu_header &h = pl.h;
    EmitP &P = pl.P; memset(&P, 0, sizeof P);
    int has_mask = (h.flags >> 2) & 1, has_data = (h.flags >> 1) & 1, has_qual = h.flags & 1;
    int mode = o->out_type == NAF_OUT_DEFAULT ? (has_qual ? NAF_OUT_FASTQ : NAF_OUT_FASTA) : o->out_type;   // unnaf.c:372-375
    pl.empty = false; pl.total = 0; pl.need_qual = false; pl.main_total = pl.surplus = pl.sur_c = pl.sur_base0 = 0;
    pl.fourbit = h.seq_type <= NAF_SEQ_RNA;
    u64 N = h.n_sequences, T = h.orig_size[S_SEQ];
    pl.seq_bytes = pl.fourbit ? (T + 1) / 2 : T;
    if (N == 0 || !has_data) { pl.empty = true; return 0; }                         // unnaf.c:409, output.c:610
    if (mode == NAF_OUT_FASTQ && !has_qual) return ctx_fail(c, NAF_GPU_EFORMAT, "FASTQ output requested, but input has no qualities\n");
    if (mode == NAF_OUT_4BIT) {
        if (!pl.fourbit) return ctx_fail(c, NAF_GPU_EFORMAT, "input has no 4-bit encoded data, but %s sequences\n", h.seq_type == NAF_SEQ_PROTEIN ? "protein" : "text");
        P.mode = -1; pl.total = pl.main_total = pl.seq_bytes; return 0;
    }
    P.mode = mode == NAF_OUT_FASTA ? EM_FASTA : mode == NAF_OUT_FASTQ ? EM_FASTQ : mode == NAF_OUT_SEQ ? EM_SEQ : EM_SEQUENCES;
    P.N = N; P.T = T;
    P.L = o->line_length >= 0 ? (u64)o->line_length : h.line_length;
    P.sep = h.separator; P.hdr_char = P.mode == EM_FASTQ ? '@' : '>';
    P.masking = o->use_mask && has_mask && P.mode != EM_FASTQ && pl.fourbit;         // unnaf.c:442; mask only exists for DNA/RNA
    P.upper = !pl.fourbit && !o->use_mask;
    const char *tab = h.seq_type == NAF_SEQ_RNA ? "-UGKCYSBAWRDMHVN" : "-TGKCYSBAWRDMHVN";   // unnaf.c:13,369
    memcpy(P.lut, tab, 16);
    const char *fs = ctx_opt(c, "FORCE_SLOW"); P.force_slow = fs && fs[0] == '1';
    P.nt_store = 1;
    pl.need_qual = P.mode == EM_FASTQ;

    return 0;
}

// Lengths, ids, names and the prefix tables built from them (text offset / first base of every record).  `aux`: a context of its
// own for ids + names (second host thread and stream); nullptr = everything on c, in order.
template <typename Hook>
static int unnaf_sections_main(naf_gpu_ctx *c, const u8 *d_naf, UnnafPlan &pl, naf_gpu_ctx *aux, Hook early, bool early_has_work, naf_gpu_ctx *aux_names)
{
    const naf_gpu_header &h = pl.h;
    EmitP &P = pl.P;
    int has_ids = (h.flags >> 5) & 1, has_names = (h.flags >> 4) & 1, has_len = (h.flags >> 3) & 1;
    u64 N = h.n_sequences, T = h.orig_size[S_SEQ];
    int rc;
    if (P.mode == EM_SEQ) {
        pl.total = pl.main_total = T;
    } else {
        if (!has_len) return ctx_fail(c, NAF_GPU_EFORMAT, "archive has no lengths");
        // ids and names: on `aux` (its own host thread and stream) beside the lengths when there is one, else after them
        P.has_ids = 0; P.has_names = 0;
        const bool want_names = P.mode == EM_FASTA || P.mode == EM_FASTQ;
        if (want_names) { P.has_ids = has_ids; P.has_names = has_names; }
        // ids, names and lengths of an archive with few records are a few hundred bytes each: one launch decodes all three (the ones
        // that are not small, or fail, take the ordinary path below, which also words the error)
        u8 *pre[3] = { nullptr, nullptr, nullptr };
        auto small3 = [&](naf_gpu_ctx *x) -> int {
            const int sec[3] = { S_IDS, S_NAMES, S_LEN }; const bool wanted[3] = { want_names && has_ids != 0, want_names && has_names != 0, true };
            const u8 *src[3]; size_t len[3], cap[3]; u8 *dst[3]; bool ok[3]; int idx[3], m = 0;
            for (int k = 0; k < 3; k++) {
                if (!wanted[k] || h.orig_size[sec[k]] == 0) continue;
                u8 *b = (u8 *)arena_alloc(x, h.orig_size[sec[k]] + 32); if (!b) return NAF_GPU_ENOMEM;
                src[m] = d_naf + h.payload_off[sec[k]]; len[m] = h.comp_size[sec[k]]; dst[m] = b; cap[m] = h.orig_size[sec[k]]; idx[m++] = k;
            }
            if (!m) return 0;
            int r = zstd_small_batch(x, m, src, len, dst, cap, ok); if (r) return r;
            for (int q = 0; q < m; q++) if (ok[q]) pre[idx[q]] = dst[q];
            return 0;
        };
        // which = 1: ids, 2: names, 3: both (one after the other)
        auto ids_names = [&](naf_gpu_ctx *x, int which = 3) -> int {
            int r;
            if ((which & 1) && want_names && has_ids) {
                u8 *b = pre[0]; u64 *z = nullptr;
                if (h.orig_size[S_IDS] == 0) return ctx_fail(x, NAF_GPU_EFORMAT, "corrupted ids - not 0-terminated\n");
                if (!b && (r = load_section(x, d_naf, h, S_IDS, h.orig_size[S_IDS], "ids", &b, pl.frame_head[S_IDS]))) return r;
                if ((r = zero_positions(x, b, h.orig_size[S_IDS], N, &z, false))) return r;
                P.ids = b; P.idz = z;
            }
            if ((which & 2) && want_names && has_names) {
                u8 *b = pre[1]; u64 *z = nullptr;
                if (h.orig_size[S_NAMES] == 0) return ctx_fail(x, NAF_GPU_EFORMAT, "corrupted names - not 0-terminated\n");
                if (!b && (r = load_section(x, d_naf, h, S_NAMES, h.orig_size[S_NAMES], "names", &b, pl.frame_head[S_NAMES]))) return r;
                if ((r = zero_positions(x, b, h.orig_size[S_NAMES], N, &z, true))) return r;
                P.names = b; P.nmz = z;
            }
            return 0;
        };
        u64 *rec_len = nullptr;
        auto lengths = [&](naf_gpu_ctx *x) -> int {
            int r;
            u8 *lens = pre[2];
            if (!lens && (r = load_section(x, d_naf, h, S_LEN, h.orig_size[S_LEN], "lengths", &lens, pl.frame_head[S_LEN]))) return r;
            u64 n_len = h.orig_size[S_LEN] / 4;
            u64 *flag = arena_new<u64>(x, n_len + 2); rec_len = arena_new<u64>(x, N + 1);
            if (!flag || !rec_len) return NAF_GPU_ENOMEM;
            HIP_TRY(x, hipMemsetAsync(rec_len, 0, (N + 1) * 8, x->stream));
            if (n_len) LAUNCH(x, "unnaf_len_flags", k_len_flags, cdiv(n_len, 256), 256, 0, (const u32 *)lens, n_len, flag);
            if ((r = scan_exclusive_u64(x, flag, n_len, flag + n_len + 1))) return r;
            if (n_len) LAUNCH(x, "unnaf_len_acc", k_len_acc, cdiv(n_len, 256), 256, 0, (const u32 *)lens, n_len, (const u64 *)flag, rec_len, N);
            u64 nrec = 0;
            if ((r = ctx_readback(x, &nrec, flag + n_len + 1, 8))) return r;
            if (nrec < N) return ctx_fail(x, NAF_GPU_EFORMAT, "corrupted lengths: %llu records described, %llu expected", (unsigned long long)nrec, (unsigned long long)N);
            return 0;
        };
        // Few records, every one of the three streams a small frame: two launches and one read-back make all the tables (k_side_tables),
        // on `aux` beside the mask stream when there is one (NAF_GPU_SIDE_FUSED=0: the long way below, which is also where a frame that
        // turns out not to decode goes, for its message)
        u64 tot[2] = { 0, 0 };
        bool fused_done = false;
        {
            const char *sf = ctx_opt(c, "SIDE_FUSED");
            const int sec[3] = { S_IDS, S_NAMES, S_LEN }; const bool wanted[3] = { want_names && has_ids != 0, want_names && has_names != 0, true };
            bool fits = !(sf && sf[0] == '0') && N <= SIDE_FUSED_N && h.orig_size[S_LEN] % 4 == 0;
            for (int k = 0; k < 3; k++) if (wanted[k] && !zstd_small_fits(h.comp_size[sec[k]], h.orig_size[sec[k]])) fits = false;
            for (int k = 0; k < 3; k++) if (wanted[k] && h.orig_size[sec[k]] == 0) fits = false;
            if (fits) {
                u64 st[6] = { 0, 0, 0, 0, 0, 0 };
                auto fused = [&](naf_gpu_ctx *x) -> int {
                    const u8 *src[3]; size_t len[3], cap[3]; u8 *dst[3]; u8 *buf[3] = { nullptr, nullptr, nullptr }; int m = 0, r;
                    SideJob J; memset(&J, 0, sizeof J);
                    for (int k = 0; k < 3; k++) {
                        if (!wanted[k]) continue;
                        u8 *b = (u8 *)arena_alloc(x, h.orig_size[sec[k]] + 32); if (!b) return NAF_GPU_ENOMEM;
                        src[m] = d_naf + h.payload_off[sec[k]]; len[m] = h.comp_size[sec[k]]; dst[m] = b; cap[m] = h.orig_size[sec[k]];
                        J.cap[m] = (u32)cap[m]; J.len[m] = (u32)len[m]; buf[k] = b; m++;
                    }
                    u32 *d_res = nullptr;
                    if ((r = zstd_small_launch(x, m, src, len, dst, cap, &d_res))) return r;
                    J.res = d_res; J.n_jobs = (u32)m;
                    J.ids = buf[0]; J.ids_n = h.orig_size[S_IDS]; J.names = buf[1]; J.names_n = h.orig_size[S_NAMES];
                    J.lens = (const u32 *)buf[2]; J.n_len = h.orig_size[S_LEN] / 4;
                    J.N = N; J.L = P.L; J.has_ids = P.has_ids; J.has_names = P.has_names; J.mode = P.mode;
                    J.idz = P.has_ids ? arena_new<u64>(x, N + 1) : nullptr; J.nmz = P.has_names ? arena_new<u64>(x, N + 1) : nullptr;
                    J.rec_len = arena_new<u64>(x, N + 1); J.hdr_len = arena_new<u32>(x, N + 1);
                    J.rec_out = arena_new<u64>(x, N + 2); J.rec_base = arena_new<u64>(x, N + 2); J.status = arena_new<u64>(x, 8);
                    if ((P.has_ids && !J.idz) || (P.has_names && !J.nmz) || !J.rec_len || !J.hdr_len || !J.rec_out || !J.rec_base || !J.status) return NAF_GPU_ENOMEM;
                    LAUNCH(x, "unnaf_side_tables", k_side_tables, 1, 256, 0, J);
                    if ((r = ctx_readback(x, st, J.status, sizeof st))) return r;
                    if (st[0] != 1) return 0;
                    P.ids = buf[0]; P.idz = J.idz; P.names = buf[1]; P.nmz = J.nmz;
                    P.rec_len = J.rec_len; P.hdr_len = J.hdr_len; P.rec_out = J.rec_out; P.rec_base = J.rec_base;
                    fused_done = true;
                    return 0;
                };
                int rf = 0;
                if (aux) { ctx_worker_start(aux, [&] { rf = fused(aux); }); early(); ctx_worker_join(aux); if (rf) memcpy(c->err, aux->err, sizeof c->err); }   // no return until it is joined
                else rf = fused(c);
                if (rf) return rf;
                if (fused_done) {
                    // the order (and the words) of the long way: lengths, ids, names
                    if (st[3] < N) return ctx_fail(c, NAF_GPU_EFORMAT, "corrupted lengths: %llu records described, %llu expected", (unsigned long long)st[3], (unsigned long long)N);
                    if (P.has_ids && st[1] < N) return ctx_fail(c, NAF_GPU_EFORMAT, "currupted ids - can't read id %llu\n", (unsigned long long)st[1]);
                    if (P.has_names && st[2] < N) return ctx_fail(c, NAF_GPU_EFORMAT, "corrupted names - can't read name %llu\n", (unsigned long long)st[2]);
                    tot[0] = st[4]; tot[1] = st[5];
                }
            }
        }
        if (fused_done) rec_len = (u64 *)P.rec_len;
        else {
        // With a second context: ids and names go there.  When this context also has the mask stream to decode (`early`), the
        // lengths follow the names over there, so that the two chains (mask | ids, names, lengths) are about as long as each other.
        const bool aux_run = aux && want_names && (has_ids || has_names);
        int rc_aux = 0, rc_len = 0;
        const bool len_on_aux = aux_run && early_has_work;
        int rc_small = 0;
        if (aux_run && !len_on_aux) rc_small = small3(c);                    // this context has nothing else to do meanwhile
        // Ids and names of an archive of many records (a FASTQ's 12 M reads per 4 GB: two streams through the sequence executor, 3.9 and
        // 1.1 ms one after the other, and the emit waits for both) go to a context each (NAF_GPU_NAMES_BESIDE=0: both on `aux`).
        bool names_beside = aux_run && aux_names && !len_on_aux && has_ids && has_names && h.orig_size[S_IDS] >= (1u << 20) && h.orig_size[S_NAMES] >= (1u << 20);
        { const char *nb = ctx_opt(c, "NAMES_BESIDE"); if (nb && nb[0] == '0') names_beside = false; }
        int rc_names = 0;
        if (names_beside) ctx_worker_start(aux_names, [&] { rc_names = ids_names(aux_names, 2); });               // pre[] is set by now (small3 ran on c above), no return until it is joined
        if (aux_run) ctx_worker_start(aux, [&] { if (len_on_aux) rc_small = small3(aux); rc_aux = rc_small ? rc_small : ids_names(aux, names_beside ? 1 : 3); if (len_on_aux && !rc_small) rc_len = lengths(aux); });     // no return until it is joined
        early();                                                                                         // work that needs none of this (the mask stream)
        if (!aux_run) rc_small = small3(c);
        if (!len_on_aux) rc_len = rc_small ? rc_small : lengths(c);
        if (aux_run) { ctx_worker_join(aux); hipStreamSynchronize(aux->stream); }
        if (names_beside) { ctx_worker_join(aux_names); hipStreamSynchronize(aux_names->stream); }
        if (rc_len) { if (len_on_aux) memcpy(c->err, aux->err, sizeof c->err); return rc_len; }           // the order a sequential run reports in: lengths, ids, names
        if (!aux_run) rc_aux = ids_names(c);
        else if (rc_aux) memcpy(c->err, aux->err, sizeof c->err);
        if (rc_aux) return rc_aux;
        if (rc_names) { memcpy(c->err, aux_names->err, sizeof c->err); return rc_names; }                      // (ids, then names: the order a sequential run reports in)
        P.rec_len = rec_len;
        u32 *hdr_len = arena_new<u32>(c, N + 1);
        u64 *rec_out = arena_new<u64>(c, N + 2), *rec_base = arena_new<u64>(c, N + 2);
        if (!hdr_len || !rec_out || !rec_base) return NAF_GPU_ENOMEM;
        LAUNCH(c, "unnaf_rec_sizes", k_rec_sizes, cdiv(N, 256), 256, 0, N, (const u64 *)rec_len, P.idz, P.nmz, P.has_ids, P.has_names, P.mode, P.L, hdr_len, rec_out, rec_base);
        // exclusive scans; element N receives the total (scan over N+1 entries with a zero tail)
        HIP_TRY(c, hipMemsetAsync(rec_out + N, 0, 8, c->stream));
        HIP_TRY(c, hipMemsetAsync(rec_base + N, 0, 8, c->stream));
        if ((rc = scan_exclusive_u64(c, rec_out, N + 1, (u64 *)nullptr))) return rc;
        if ((rc = scan_exclusive_u64(c, rec_base, N + 1, (u64 *)nullptr))) return rc;
        if ((rc = ctx_readback2(c, &tot[0], rec_out + N, 8, &tot[1], rec_base + N, 8))) return rc;
        P.hdr_len = hdr_len; P.rec_out = rec_out; P.rec_base = rec_base;
        }
        if (tot[1] > T) return ctx_fail(c, NAF_GPU_EFORMAT, "sum of lengths (%llu) exceeds the stored sequence length (%llu)", (unsigned long long)tot[1], (unsigned long long)T);
        // every base of a read needs its quality byte: a shorter quality stream would be read past its end by the emit kernels (the
        // reference has no message for this -- print_quality_from_file, output-fastq.c:69-85, never returns on such an archive)
        if (P.mode == EM_FASTQ && h.orig_size[S_QUAL] < tot[1])
            return ctx_fail(c, NAF_GPU_EFORMAT, "corrupted quality: %llu quality codes stored for %llu bases\n", (unsigned long long)h.orig_size[S_QUAL], (unsigned long long)tot[1]);
        if (P.mode == EM_SEQUENCES && T == 0) tot[0] = 0;                            // output-sequences.c:81: nothing printed
        pl.total = pl.main_total = tot[0]; pl.surplus = 0; pl.sur_c = 0;
        if (tot[1] < T && P.mode != EM_FASTQ) {
            // more bases than the lengths account for.  FASTQ never prints them (output-fastq.c:100-149 stops after the N-th
            // read); --sequences appends them raw (output-sequences.c:82-116); FASTA wraps them on from the last non-empty record's line
            const u64 S = T - tot[1];
            u64 tail = S;
            if (P.mode == EM_FASTA) {
                unsigned long long *d_last = arena_new<unsigned long long>(c, 2); if (!d_last) return NAF_GPU_ENOMEM;
                HIP_TRY(c, hipMemsetAsync(d_last, 0, 16, c->stream));
                LAUNCH(c, "unnaf_last_nonempty", k_last_nonempty, cdiv(N, 256), 256, 0, (const u64 *)rec_len, N, d_last);
                u64 last1 = 0; if ((rc = ctx_readback(c, &last1, d_last, 8))) return rc;
                if (last1 == 0) tail = 0;                                           // no record ever started a line of bases: nothing is printed
                else {
                    u64 len_last = 0; if ((rc = ctx_readback(c, &len_last, rec_len + (last1 - 1), 8))) return rc;
                    const u64 L = P.L;
                    const u64 cc = L ? ((len_last % L) ? L - len_last % L : 0) : 0;
                    pl.sur_c = cc;
                    if (L && S > cc) { const u64 R = S - cc; tail = cc + 1 + R + (R - 1) / L; }
                }
            }
            pl.surplus = tail ? S : 0; pl.sur_base0 = tot[1];
            pl.total = pl.main_total + tail;
        }
    }
    return 0;
}

// The side streams (lengths, ids, names, mask) and the tables built from them.  Runs on whichever context it is given: the
// archive's own for byte-range calls (aux = aux_mask = nullptr: sequential), the side context for whole-text calls, which also
// hands over contexts for ids + names and for the mask.
static int unnaf_sections(naf_gpu_ctx *c, const u8 *d_naf, UnnafPlan &pl, naf_gpu_ctx *aux = nullptr, naf_gpu_ctx *aux_mask = nullptr, naf_gpu_ctx *aux_names = nullptr)
{
    const naf_gpu_header &h = pl.h;
    EmitP &P = pl.P;
    // mask stream -> toggle table.  A handful of long Huffman streams: its decode is pure latency, so with `aux_mask` it runs on a
    // context of its own from the start of the call
    auto mask_part = [&](naf_gpu_ctx *x) -> int {
        int r;
        u8 *mu = nullptr; u64 n_mask = h.orig_size[S_MASK];
        {
            // a frame of a few KB for MBs of units: Raw / RLE blocks, one launch (NAF_GPU_MASK_RLE=0: never)
            const char *mr = ctx_opt(c, "MASK_RLE");
            const u64 cs = h.comp_size[S_MASK];
            if (cs >= 4 && cs <= MASK_RLE_SRC && n_mask >= 2 * cs && !(mr && mr[0] == '0')) {
                // Frame_Header (RFC 8878 3.1.1.1) of a frame without dictionary and checksum: descriptor, window byte, content size
                const u8 fhd = pl.frame_head[S_MASK][0];
                const u32 fcs_flag = fhd >> 6, single = (fhd >> 5) & 1;
                struct { u32 hdr_size; bool ok; } fh;
                fh.hdr_size = 1 + (single ? 0 : 1) + (fcs_flag == 0 ? single : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8)));
                fh.ok = !(fhd & 0x0F);                               // no reserved bit, no checksum, no dictionary id
                if (fh.ok && fh.hdr_size < cs) {
                    u64 *tg = arena_new<u64>(x, MASK_RLE_TOG + 1), *d_res = arena_new<u64>(x, 4);
                    if (!tg || !d_res) return NAF_GPU_ENOMEM;
                    LAUNCH(x, "unnaf_mask_rle", k_mask_rle_frame, 1, 256, (u32)((cs + 15) & ~15ull), d_naf + h.payload_off[S_MASK], (u32)cs, fh.hdr_size, n_mask, tg, MASK_RLE_TOG, d_res);
                    u64 res[3] = { 0, 0, 0 };
                    if ((r = ctx_readback(x, res, d_res, 24))) return r;
                    if (res[0] == 1) { P.toggles = tg; P.n_toggles = res[1]; return 0; }
                }
            }
        }
        if ((r = load_section(x, d_naf, h, S_MASK, n_mask, "mask", &mu, pl.frame_head[S_MASK]))) return r;
        u64 tiles = (n_mask + MT_TILE - 1) / MT_TILE;
        u64 *ts = arena_new<u64>(x, tiles + 2), *tc = arena_new<u64>(x, tiles + 2);
        if (!ts || !tc) return NAF_GPU_ENOMEM;
        if (tiles) LAUNCH(x, "unnaf_mask_count", k_mask_count, tiles, 256, 0, (const u8 *)mu, n_mask, ts, tc);
        if ((r = scan_exclusive_u64(x, ts, tiles, (u64 *)nullptr))) return r;
        if ((r = scan_exclusive_u64(x, tc, tiles, tc + tiles + 1))) return r;
        u64 ntog = 0;
        if ((r = ctx_readback(x, &ntog, tc + tiles + 1, 8))) return r;
        u64 *tg = arena_new<u64>(x, ntog + 1);
        if (!tg) return NAF_GPU_ENOMEM;
        if (tiles) LAUNCH(x, "unnaf_mask_scatter", k_mask_scatter, tiles, 256, 0, (const u8 *)mu, n_mask, (const u64 *)ts, (const u64 *)tc, tg, ntog);
        P.toggles = tg; P.n_toggles = ntog;
        return 0;
    };
    int rc_mask = 0;
    const bool mask_started = P.masking && aux_mask;
    if (mask_started) ctx_worker_start(aux_mask, [&] { rc_mask = mask_part(aux_mask); });      // joined below: no return before
    // Without a context of its own the mask goes FIRST on this one, right after the ids / names thread has been started: this
    // context would otherwise idle while it waits for that thread, and the mask decode (a few long Huffman streams) is pure latency.
    // Errors keep the order of a sequential run: lengths, ids, names, then mask.
    bool mask_early = false; char mask_err[sizeof c->err];
    auto early = [&]() {
        if (P.masking && !mask_started && P.mode != EM_SEQ && aux) { mask_early = true; rc_mask = mask_part(c); if (rc_mask) memcpy(mask_err, c->err, sizeof c->err); }
    };
    int rc = unnaf_sections_main(c, d_naf, pl, aux, early, P.masking && !mask_started && P.mode != EM_SEQ && aux != nullptr, aux_names);
    if (mask_started) { ctx_worker_join(aux_mask); hipStreamSynchronize(aux_mask->stream); }
    if (rc) return rc;                                                                                     // the order a sequential run reports in
    if (mask_started && rc_mask) { memcpy(c->err, aux_mask->err, sizeof c->err); return rc_mask; }
    if (mask_early) { if (rc_mask) { memcpy(c->err, mask_err, sizeof c->err); return rc_mask; } }
    else if (P.masking && !mask_started && (rc = mask_part(c))) return rc;
    return 0;
}

static int unnaf_run(naf_gpu_ctx *c, const u8 *d_naf, size_t naf_len, const naf_gpu_unnaf_opts *o,
                     u64 out_begin, u64 out_end, bool whole, u8 *d_out, size_t out_cap, size_t *out_len, bool size_only)
{
    if (!c || !d_naf || !o || !out_len) return NAF_GPU_EARG;
    arena_reset(c);
    if (whole && !size_only) { int rs = ctx_sides_ready(c); if (rs) return rs; }        // (a range or a size call stays on this context)
    UnnafPlan pl;
    int rc = unnaf_prepare(c, d_naf, naf_len, o, pl); if (rc) return rc;
    if (pl.empty) { *out_len = 0; return 0; }
    const naf_gpu_header &h = pl.h;
    ZRange rgs, rgq; ZRange *prs = nullptr, *prq = nullptr;
    memset(&rgs, 0, sizeof rgs); memset(&rgq, 0, sizeof rgq);
    u8 *seq = nullptr;
    // sequence (and quality) payload: the dominant zstd streams
    auto payload_seq = [&]() -> int {
        int r;
        u64 seq_need = prs ? (rgs.want_hi - rgs.want_lo) + 2 * 131072 + 64 : pl.seq_bytes + 64;
        if (seq_need > pl.seq_bytes + 64) seq_need = pl.seq_bytes + 64;
        seq = (u8 *)arena_alloc(c, seq_need);
        if (!seq) return NAF_GPU_ENOMEM;
        size_t n = 0;
        r = zstd_decode_range(c, d_naf + h.payload_off[S_SEQ], h.comp_size[S_SEQ], 0, seq, prs ? seq_need : pl.seq_bytes, &n, prs, pl.frame_head[S_SEQ]);
        if (r == NAF_GPU_ECAP && prs) {                                              // dependent blocks: needs the whole stream
            seq = (u8 *)arena_alloc(c, pl.seq_bytes + 64); if (!seq) return NAF_GPU_ENOMEM;
            r = zstd_decode(c, d_naf + h.payload_off[S_SEQ], h.comp_size[S_SEQ], 0, seq, pl.seq_bytes, &n); prs = nullptr;
        }
        if (r == NAF_GPU_ECAP || (r == 0 && n != pl.seq_bytes)) return ctx_fail(c, NAF_GPU_EFORMAT, "can't decompress sequence\n");
        if (r) return r;
        pl.P.seq = (prs && prs->ranged) ? (prs->own_buf ? prs->own_buf : seq) - prs->got_lo : seq;
        return 0;
    };
    // qc: the context the quality stream is decoded on (the archive's own, or the second side context beside the sequence stream)
    auto payload_qual = [&](naf_gpu_ctx *qc) -> int {
        int r;
        u64 qn = h.orig_size[S_QUAL];
        u64 q_need = prq ? (rgq.want_hi - rgq.want_lo) + 2 * 131072 + 64 : qn + 64;
        if (q_need > qn + 64) q_need = qn + 64;
        u8 *q = (u8 *)arena_alloc(qc, q_need); if (!q) return NAF_GPU_ENOMEM;
        size_t qgot = 0;
        r = zstd_decode_range(qc, d_naf + h.payload_off[S_QUAL], h.comp_size[S_QUAL], 0, q, prq ? q_need : qn, &qgot, prq, pl.frame_head[S_QUAL]);
        if (r == NAF_GPU_ECAP && prq) {
            q = (u8 *)arena_alloc(qc, qn + 64); if (!q) return NAF_GPU_ENOMEM;
            r = zstd_decode(qc, d_naf + h.payload_off[S_QUAL], h.comp_size[S_QUAL], 0, q, qn, &qgot); prq = nullptr;
        }
        if (r == NAF_GPU_ECAP || (r == 0 && qgot != qn)) return ctx_fail(qc, NAF_GPU_EFORMAT, "can't decompress quality\n");
        if (r) return r;
        pl.P.qual = (prq && prq->ranged) ? (prq->own_buf ? prq->own_buf : q) - prq->got_lo : q;
        return 0;
    };
    auto payload = [&]() -> int { int r = payload_seq(); if (r) return r; return pl.need_qual ? payload_qual(c) : 0; };
    const char *fuse = ctx_opt(c, "FUSE");
    const bool fuse_on = fuse && fuse[0] == '1';
    // Whole-text call: the side streams (a chain of small launches and read-backs, mostly latency) are prepared by a second
    // host thread on the side context's stream while this thread decodes the payload; they meet before the emit.
    const bool par = whole && !size_only && pl.P.mode != -1 && c->side && !fuse_on;
    bool payload_done = false;
    ZSplit split; split.parts = 0; split.done = 0; split.status = nullptr;
    // A whole 4-bit text with long records goes through the tile kernels: if its sequence stream turns out to be a flat frame
    // (ctx.h: ZFlat) those read it in place and nothing is decoded.  NAF_GPU_FLAT_FUSE=0: always decode first (cross-check).
    ZFlat zflat; memset(&zflat, 0, sizeof zflat);
    struct FlatGuard { ZFlat *z; ~FlatGuard() { zstd_flat_drop(z); } } flat_guard{ &zflat };      // (a job the decoder left and nobody ran: an error on the way)
    const char *ff = ctx_opt(c, "FLAT_FUSE"), *ek0 = ctx_opt(c, "EMIT");
    const bool try_flat = !size_only && pl.fourbit && !fuse_on && !pl.P.force_slow && !(ff && ff[0] == '0') && !(ek0 && ek0[0]) &&
                          (pl.P.mode == EM_FASTA || pl.P.mode == EM_SEQ || pl.P.mode == EM_SEQUENCES) && pl.P.N && h.orig_size[S_SEQ] / pl.P.N >= 16384 &&
                          (pl.P.mode != EM_FASTA || pl.P.L == 0 || pl.P.L >= 16);
    if (par) {
        arena_reset(c->side);
        HIP_TRY(c, hipEventRecord(c->fork_ev, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->side->stream, c->fork_ev, 0));
        int rc_side = 0, rc_q = 0;
        const bool qpar = pl.need_qual && c->side2;
        if (qpar) { arena_reset(c->side2); HIP_TRY(c, hipStreamWaitEvent(c->side2->stream, c->fork_ev, 0)); }
        if (c->side3) { arena_reset(c->side3); HIP_TRY(c, hipStreamWaitEvent(c->side3->stream, c->fork_ev, 0)); }
        if (c->side4) { arena_reset(c->side4); HIP_TRY(c, hipStreamWaitEvent(c->side4->stream, c->fork_ev, 0)); }
        // no early return between here and the joins
        ctx_worker_start(c->side, [&] { rc_side = unnaf_sections(c->side, d_naf, pl, c->side3, nullptr, c->side4); });   // the mask stays on this context: a thread of its own measured slower, with and without the split decode
        if (qpar) ctx_worker_start(c->side2, [&] { rc_q = payload_qual(c->side2); });
        // decode -> emit pipeline (ZSplit): the quality context is free when there is no quality stream
        const char *nsp = ctx_opt(c, "SPLIT");
        const int nparts = nsp ? atoi(nsp) : 4;
        if (!pl.need_qual && c->side2 && nparts >= 2 && nparts <= ZSPLIT_MAX && pl.P.mode != -1) {
            split.parts = nparts; split.done = 0;
            for (int k = 0; k < nparts; k++) split.ev[k] = c->split_ev[k];
            c->zsplit = &split;
            arena_reset(c->side2);                                // (what the pipeline and the flat frame's job allocate there -- launch_lz_exec's arrays among it -- is this call's: ADVICE r05)
        }
        if (try_flat) { zflat.ready = false; zflat.aux = c->zsplit ? c->side2 : nullptr; c->zflat = &zflat; }
        rc = payload_seq();
        c->zsplit = nullptr; c->zflat = nullptr;
        if (!rc && pl.need_qual && !qpar) rc = payload_qual(c);
        ctx_worker_join(c->side);
        if (qpar) ctx_worker_join(c->side2);
        if (rc_side) { memcpy(c->err, c->side->err, sizeof c->err); return rc_side; }   // the order a sequential run reports errors in
        if (rc) return rc;
        if (rc_q) { memcpy(c->err, c->side2->err, sizeof c->err); return rc_q; }
        HIP_TRY(c, hipStreamSynchronize(c->side->stream));
        if (qpar) HIP_TRY(c, hipStreamSynchronize(c->side2->stream));
        payload_done = true;
    } else if (pl.P.mode != -1) {
        if ((rc = unnaf_sections(c, d_naf, pl))) return rc;
    }
    if (whole) { out_begin = 0; out_end = pl.total; }
    if (out_end > pl.total) out_end = pl.total;
    if (out_begin > out_end) out_begin = out_end;
    *out_len = out_end - out_begin;
    if (size_only) { *out_len = pl.total; return 0; }
    if (*out_len > out_cap) return ctx_fail(c, NAF_GPU_ECAP, "unnaf output needs %llu bytes, capacity %zu", (unsigned long long)*out_len, out_cap);
    if (*out_len == 0) return 0;
    // the records' text is [0, main_total); bases behind the last record follow it and go through k_emit_surplus
    const u64 m_begin = out_begin < pl.main_total ? out_begin : pl.main_total, m_end = out_end < pl.main_total ? out_end : pl.main_total;
    const bool has_main = m_end > m_begin, has_tail = out_end > pl.main_total;
    pl.P.out_begin = m_begin; pl.P.out_end = m_end;
    // Byte-range call (multi-GPU shard): find the bases this range touches and decode only the zstd blocks behind them.
    u64 rec0 = 0, rec1 = pl.P.N ? pl.P.N - 1 : 0;                                        // records the byte range touches
    if (!whole && pl.P.mode != -1) {
        u64 g[4] = { pl.sur_base0, pl.sur_base0, rec1, rec1 };
        if (has_main) {
            u64 *d_g = arena_new<u64>(c, 4); if (!d_g) return NAF_GPU_ENOMEM;
            LAUNCH(c, "unnaf_range_bases", k_range_bases, 1, 64, 0, pl.P, d_g);
            if ((rc = ctx_readback(c, g, d_g, 32))) return rc;
        }
        rec0 = g[2]; rec1 = g[3];
        if (g[1] < g[0]) g[1] = g[0];
        if (has_tail) g[1] = pl.P.T;
        rgs.want_lo = pl.fourbit ? g[0] / 2 : g[0]; rgs.want_hi = pl.fourbit ? (g[1] + 1) / 2 : g[1];
        rgq.want_lo = g[0]; rgq.want_hi = g[1];
        prs = &rgs; prq = &rgq;
    }
    // Whole FASTA text of a 4-bit archive: decode and emit in one kernel when the frame is literal-only.
    // (Measured slower than decode + emit on MI355X for now: its per-lane 16-byte text stores are partial-line
    // writes from 600 k streams; kept behind NAF_GPU_FUSE=1 and under test until the text is staged through LDS.)
    if (whole && pl.P.mode == EM_FASTA && pl.fourbit && (pl.P.L == 0 || pl.P.L >= 16) && !pl.P.force_slow && fuse_on && !pl.surplus) {
        size_t n2 = 0;
        rc = zstd_decode_fused_fasta(c, d_naf + h.payload_off[S_SEQ], h.comp_size[S_SEQ], 0, &n2, &pl.P, d_out);
        if (rc == 0) {
            if (n2 != pl.seq_bytes) return ctx_fail(c, NAF_GPU_EFORMAT, "can't decompress sequence\n");
            return 0;
        }
        if (rc != -100) return rc;
    }
    if (!payload_done) {
        if (try_flat) { zflat.ready = false; c->zflat = &zflat; }
        rc = payload();
        c->zflat = nullptr;
        if (rc) return rc;
    }
    if (zflat.ready) { pl.P.fsrc = zflat.src; pl.P.fsi = zflat.si; pl.P.fslots = zflat.nslots; pl.P.fsym = zflat.sym; pl.P.ftail = zflat.tail; pl.P.ftail_q = zflat.tail_q; pl.P.ftail_n = zflat.tail_n; pl.P.fcls = zflat.cls; }
    if (pl.P.mode == -1) {                                                             // --4bit: the stream itself
        HIP_TRY(c, hipMemcpyAsync(d_out, seq + out_begin, out_end - out_begin, hipMemcpyDeviceToDevice, c->stream));
        return 0;
    }
    if (has_main) {
        // short records (FASTQ reads, contigs, proteins): segment-composing kernel; long records: streaming kernel
        const char *ek = ctx_opt(c, "EMIT");
        bool short_rec = pl.P.mode != EM_SEQ && pl.total / pl.P.N < 16384;
        if (ek && !strcmp(ek, "short")) short_rec = pl.P.mode != EM_SEQ;
        if (ek && !strcmp(ek, "long")) short_rec = false;
        if (pl.P.force_slow) { short_rec = false; ek = "span"; }
        if (whole && pl.P.mode == EM_FASTQ && !pl.P.force_slow && !(ek && ek[0])) {
            {
                const u64 avg = pl.P.N ? (pl.P.out_end - pl.P.out_begin) / pl.P.N + 1 : 1;        // bytes of text per read
                const u32 lanes = avg * 64 <= ER_STAGE * 7 / 8 ? 4u : avg * 32 <= ER_STAGE * 7 / 8 ? 8u : 16u;
                const bool w1 = !(ctx_opt(c, "EMIT_WAVE") && ctx_opt(c, "EMIT_WAVE")[0] == '0');
                const u32 wgt = w1 ? 64u : 256u, grid = (u32)cdiv(pl.P.N, wgt / lanes);
#define ER_LAUNCH(FB, LN) do { if (w1) LAUNCH(c, "unnaf_emit_records", (k_emit_fastq_records<FB, LN, 64>), grid, 64, 0, pl.P, d_out); \
                               else LAUNCH(c, "unnaf_emit_records", (k_emit_fastq_records<FB, LN, 256>), grid, 256, 0, pl.P, d_out); } while (0)
                if (pl.fourbit) { if (lanes == 4) ER_LAUNCH(true, 4); else if (lanes == 8) ER_LAUNCH(true, 8); else ER_LAUNCH(true, 16); }
                else { if (lanes == 4) ER_LAUNCH(false, 4); else if (lanes == 8) ER_LAUNCH(false, 8); else ER_LAUNCH(false, 16); }
#undef ER_LAUNCH
            }
        } else if (short_rec) {
            if (pl.P.mode == EM_FASTA || pl.P.mode == EM_FASTQ) {
                u64 nr = rec1 - rec0 + 1, htot = 0;
                u64 *ho = arena_new<u64>(c, nr + 2); if (!ho) return NAF_GPU_ENOMEM;
                LAUNCH(c, "unnaf_hdr_len64", k_hdr_len64, cdiv(nr, 256), 256, 0, pl.P.hdr_len, rec0, nr, ho);
                if ((rc = scan_exclusive_u64(c, ho, nr, ho + nr + 1))) return rc;
                if ((rc = ctx_readback(c, &htot, ho + nr + 1, 8))) return rc;
                u8 *ht = (u8 *)arena_alloc(c, htot + 32); if (!ht) return NAF_GPU_ENOMEM;
                LAUNCH(c, "unnaf_hdr_build", k_hdr_build, cdiv(nr, 32), 256, 0, pl.P, rec0, nr, (const u64 *)ho, ht);
                pl.P.hdr_off = ho; pl.P.hdr_text = ht; pl.P.hdr_r0 = rec0;
            }
            u32 grid = cdiv(m_end - m_begin, ES_SPAN);
            if (pl.fourbit) LAUNCH(c, "unnaf_emit_short", k_emit_short<true>, grid, 256, 0, pl.P, d_out);
            else LAUNCH(c, "unnaf_emit_short", k_emit_short<false>, grid, 256, 0, pl.P, d_out);
        } else if (ek && !strcmp(ek, "span")) {                                              // previous long-record kernel (kept as a cross-check)
            u32 grid = cdiv(m_end - m_begin, EMIT_SPAN);
            if (pl.fourbit) LAUNCH(c, "unnaf_emit", k_emit<true>, grid, 256, 0, pl.P, d_out);
            else LAUNCH(c, "unnaf_emit", k_emit<false>, grid, 256, 0, pl.P, d_out);
        } else {
            u64 ntiles = cdiv(m_end - m_begin, 4096);
            TileIdx *ti = arena_new<TileIdx>(c, ntiles + 2 + FLAT_TPW); u64 *tr = arena_new<u64>(c, ntiles + 2);
            u32 *list = arena_new<u32>(c, ntiles + 1), *cnt = arena_new<u32>(c, 2);
            u32 *list2 = (zflat.ready && zflat.cls) ? arena_new<u32>(c, ntiles + 1) : list;      // tiles over decoded blocks of a mostly-flat frame
            if (!ti || !tr || !list || !cnt || !list2) return NAF_GPU_ENOMEM;
            if (!split.done) HIP_TRY(c, hipMemsetAsync(cnt, 0, 8, c->stream));
            // exact c / (L+1) for c < L + 1 + 4096 as mulhi(c, M), M = floor(2^32 / (L+1)) + 1, valid while c * (L+1) < 2^32
            u64 Lp1 = pl.P.L + 1;
            pl.P.Ldiv_magic = (Lp1 >= 2 && Lp1 < 32768) ? (u32)((1ull << 32) / Lp1 + 1) : 0;
            // With a split decode (ZSplit) the index and the tiles behind the finished parts run on the second stream beside the
            // decode of the next part; this stream takes the tiles behind the last part, the boundary tiles, and waits for the other.
            naf_gpu_ctx *ic = split.done ? c->side2 : c;                                     // context the tile index is built on
            const bool tile_wave = !(ctx_opt(c, "EMIT_WAVE") && ctx_opt(c, "EMIT_WAVE")[0] == '0');      // k_emit_tile_wave / k_emit_tile_flat_wave ("0": the workgroups of 256)
            if (split.done) HIP_TRY(c, hipMemsetAsync(cnt, 0, 8, ic->stream));
            TileFlat *tsig = nullptr;
            if (zflat.ready) {
                tsig = arena_new<TileFlat>(c, ntiles + 2 + FLAT_TPW); u32 *fpair = arena_new<u32>(c, 400); if (!tsig || !fpair) return NAF_GPU_ENOMEM;
                LAUNCH(ic, "unnaf_flat_pair", k_flat_pair, 1, 256, 0, pl.P, fpair);
                pl.P.fpair = fpair;
            }
            LAUNCH(ic, "unnaf_tile_index", k_tile_index, cdiv(ntiles + 1 + FLAT_TPW, TILE_WG), 256, 0, pl.P, ntiles, ti, tr, list, cnt, tsig, (u32)FLAT_TPW, list2);
            // a mostly-flat frame: the blocks that are not flat are decoded now -- on the spare stream, beside the emit of the flat tiles
            naf_gpu_ctx *xc = c;                                                              // where the tiles over decoded blocks are emitted
            const bool flat_job = zflat.ready && zflat.later;
            // the spare stream starts behind the tile index: it takes the job (if any) and the tiles with headers / record ends, beside the bulk emit
            if (zflat.ready && zflat.aux) {
                HIP_TRY(c, hipEventRecord(c->split_ev[0], c->stream));
                if (!flat_job) { HIP_TRY(c, hipStreamWaitEvent(zflat.aux->stream, c->split_ev[0], 0)); xc = zflat.aux; }
            }
            u64 t_done = 0;
            if (split.done) {
                HIP_TRY(c, hipEventRecord(c->split_ev[ZSPLIT_MAX], ic->stream));
                for (int k = 0; k + 1 < split.parts; k++) {
                    // a tile's bases all lie below its last text position, and a lane reads at most 32 packed bytes past its own
                    u64 bases_ready = split.out_end[k] * (pl.fourbit ? 2u : 1u);
                    u64 t_hi = bases_ready / 4096; t_hi = t_hi > 1 ? t_hi - 1 : 0;
                    if (t_hi > ntiles) t_hi = ntiles;
                    if (t_hi <= t_done) continue;
                    HIP_TRY(c, hipStreamWaitEvent(ic->stream, split.ev[k], 0));
                    if (tile_wave && pl.fourbit) LAUNCH(ic, "unnaf_emit", (k_emit_tile_wave<true, 2>), cdiv(t_hi - t_done, 2u), 64, 0, pl.P, (const TileIdx *)(ti + t_done), d_out + t_done * 4096, (u64)(t_hi - t_done));
                    else if (tile_wave) LAUNCH(ic, "unnaf_emit", (k_emit_tile_wave<false, 2>), cdiv(t_hi - t_done, 2u), 64, 0, pl.P, (const TileIdx *)(ti + t_done), d_out + t_done * 4096, (u64)(t_hi - t_done));
                    else if (pl.fourbit) LAUNCH(ic, "unnaf_emit", k_emit_tile<true>, (u32)(t_hi - t_done), 256, 0, pl.P, (const TileIdx *)(ti + t_done), d_out + t_done * 4096);
                    else LAUNCH(ic, "unnaf_emit", k_emit_tile<false>, (u32)(t_hi - t_done), 256, 0, pl.P, (const TileIdx *)(ti + t_done), d_out + t_done * 4096);
                    t_done = t_hi;
                }
                HIP_TRY(c, hipEventRecord(c->split_ev[ZSPLIT_MAX + 1], ic->stream));
                HIP_TRY(c, hipStreamWaitEvent(c->stream, c->split_ev[ZSPLIT_MAX], 0));         // the index
            }
            if (zflat.ready) {
                // four tiles per workgroup (eight measured slower: DESIGN.md section 8), workgroups dealt to the XCDs in contiguous chunks
                const u32 nwg = cdiv(ntiles, 4u), chunk = (nwg + 7) / 8;
                // A wavefront per two tiles (per tile when there are many mask toggles to stage), in workgroups of 64 -- 3.17 -> 2.24 ms per
                // 10 GB against four tiles per workgroup of 256 (k_emit_tile_flat_wave).  A mostly-flat frame whose decode job walks Huffman
                // streams keeps the workgroups of 256: that walk is bound by latency, and beside the wavefront-per-tile emit it takes
                // 1.9 instead of 1.1 ms (a realistic genome, 4 GB: 3.16 -> 3.41 ms); a job of flat literals and matches only (the reference's
                // archive of random bases) is better off with it (4 GB: 2.45 -> 2.31 ms), its kernels in workgroups of 64 as well -- a
                // workgroup of 256 waits for four wave slots of one CU to be free at once.  NAF_GPU_EMIT_WAVE=0: workgroups of 256 everywhere.
                const char *ew = ctx_opt(c, "EMIT_WAVE");
                const bool mixed = flat_job || (zflat.cls && zflat.n_decoded);
                const bool wave = !(ew && ew[0] == '0') && (!mixed || zflat.n_walk == 0);
                // (many: about a toggle per tile or more -- a soft-masked genome, not the odd lower-case stretch)
                if (wave && pl.P.masking && pl.P.n_toggles >= ntiles) { const u32 ch = ((u32)ntiles + 7) / 8; LAUNCH(c, "unnaf_emit_flat", k_emit_tile_flat_wave<1>, ch * 8, 64, 0, pl.P, (const TileIdx *)ti, (const TileFlat *)tsig, d_out, (u64)ntiles, ch); }
                else if (wave) { const u32 ch = (cdiv(ntiles, 2u) + 7) / 8; LAUNCH(c, "unnaf_emit_flat", k_emit_tile_flat_wave<2>, ch * 8, 64, 0, pl.P, (const TileIdx *)ti, (const TileFlat *)tsig, d_out, (u64)ntiles, ch); }
                else
                LAUNCH(c, "unnaf_emit_flat", k_emit_tile_flat<4>, chunk * 8, 256, 0, pl.P, (const TileIdx *)ti, (const TileFlat *)tsig, d_out, (u64)ntiles, chunk);
                if (flat_job) {                                                               // (queued behind the flat emit: the job waits for its tables on the host)
                    if ((rc = zstd_flat_later(c, &zflat))) return rc;
                    if (zflat.aux) xc = zflat.aux;
                }
                if (zflat.cls && zflat.n_decoded) {
                    // a tile costs this kernel about what it costs k_emit_tile: as many workgroups as there can be tiles over decoded blocks, up to a few waves of the device
                    const u64 est = (u64)zflat.n_decoded * 64 + 64;                           // (blocks of up to 128 KiB: 64 tiles each)
                    const u32 lgrid = (u32)(est < ntiles ? (est < 16384 ? est : 16384) : (ntiles < 16384 ? ntiles : 16384));
                    if (wave) LAUNCH(xc, "unnaf_emit", k_emit_tile_list_wave<true>, lgrid, 64, 0, pl.P, (const TileIdx *)ti, (const u32 *)list2, (const u32 *)(cnt + 1), d_out);
                    else LAUNCH(xc, "unnaf_emit", k_emit_tile_list<true>, lgrid, 256, 0, pl.P, (const TileIdx *)ti, (const u32 *)list2, (const u32 *)(cnt + 1), d_out);
                }
            }
            else if (t_done < ntiles) {
                if (tile_wave && pl.fourbit) LAUNCH(c, "unnaf_emit", (k_emit_tile_wave<true, 2>), cdiv(ntiles - t_done, 2u), 64, 0, pl.P, (const TileIdx *)(ti + t_done), d_out + t_done * 4096, (u64)(ntiles - t_done));
                else if (tile_wave) LAUNCH(c, "unnaf_emit", (k_emit_tile_wave<false, 2>), cdiv(ntiles - t_done, 2u), 64, 0, pl.P, (const TileIdx *)(ti + t_done), d_out + t_done * 4096, (u64)(ntiles - t_done));
                else if (pl.fourbit) LAUNCH(c, "unnaf_emit", k_emit_tile<true>, (u32)(ntiles - t_done), 256, 0, pl.P, (const TileIdx *)(ti + t_done), d_out + t_done * 4096);
                else LAUNCH(c, "unnaf_emit", k_emit_tile<false>, (u32)(ntiles - t_done), 256, 0, pl.P, (const TileIdx *)(ti + t_done), d_out + t_done * 4096);
            }
            // tiles holding a header or a record boundary (their number stays on the device)
            const u32 rest_grid = 4u * (u32)(ntiles < EMIT_REST_GRID ? ntiles : EMIT_REST_GRID);      // (quarter tiles: k_emit_rest)
            if (pl.fourbit) LAUNCH(xc, "unnaf_emit_rest", k_emit_rest<true>, rest_grid, 64, 0, pl.P, (const TileIdx *)ti, (const u64 *)tr, (const u32 *)list, (const u32 *)cnt, d_out);
            else LAUNCH(xc, "unnaf_emit_rest", k_emit_rest<false>, rest_grid, 64, 0, pl.P, (const TileIdx *)ti, (const u64 *)tr, (const u32 *)list, (const u32 *)cnt, d_out);
            if (xc != c) { HIP_TRY(c, hipEventRecord(c->split_ev[1], xc->stream)); HIP_TRY(c, hipStreamWaitEvent(c->stream, c->split_ev[1], 0)); }
            if (split.done) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->split_ev[ZSPLIT_MAX + 1], 0));
            if (zflat.ready && ctx_tracing(c)) {          // tests: how the tiles were dealt
                u32 hc[2] = { 0, 0 };
                if (!ctx_readback(c, hc, cnt, 8)) ctx_trace(c, "[flat tiles] total %llu rest %u decoded %u\n", (unsigned long long)ntiles, hc[0], hc[1]);
            }
        }
    }
    if (has_tail) {
        const u64 lo = (out_begin > pl.main_total ? out_begin : pl.main_total) - pl.main_total, hi = out_end - pl.main_total;
        u8 *dst = d_out + (pl.main_total + lo - out_begin);
        const int wrap = pl.P.mode == EM_FASTA;
        if (pl.fourbit) LAUNCH(c, "unnaf_emit_surplus", k_emit_surplus<true>, cdiv(hi - lo, 256), 256, 0, pl.P, pl.sur_base0, pl.surplus, pl.sur_c, pl.P.L, wrap, lo, hi, dst);
        else LAUNCH(c, "unnaf_emit_surplus", k_emit_surplus<false>, cdiv(hi - lo, 256), 256, 0, pl.P, pl.sur_base0, pl.surplus, pl.sur_c, pl.P.L, wrap, lo, hi, dst);
    }
    HIP_TRY(c, hipGetLastError());
    if ((rc = zstd_split_status(c, &split))) return rc;           // a split decode left its status for after the emit was queued
    if (zflat.ready) { ZSplit fs; fs.parts = 0; fs.done = 1; fs.status = zflat.status; if ((rc = zstd_split_status(c, &fs))) return rc; }
    return 0;
}

extern "C" int naf_gpu_unnaf_size(naf_gpu_ctx *c, const void *d_naf, size_t naf_len, const naf_gpu_unnaf_opts *o, size_t *out_len)
{
    return unnaf_run(c, (const u8 *)d_naf, naf_len, o, 0, 0, true, nullptr, 0, out_len, true);
}
extern "C" int naf_gpu_unnaf(naf_gpu_ctx *c, const void *d_naf, size_t naf_len, const naf_gpu_unnaf_opts *o, void *d_out, size_t out_cap, size_t *out_len)
{
    int rc = unnaf_run(c, (const u8 *)d_naf, naf_len, o, 0, 0, true, (u8 *)d_out, out_cap, out_len, false);
    if (c) arena_settle(c);
    return rc;
}
extern "C" int naf_gpu_unnaf_range(naf_gpu_ctx *c, const void *d_naf, size_t naf_len, const naf_gpu_unnaf_opts *o,
                                   uint64_t out_begin, uint64_t out_end, void *d_out, size_t out_cap, size_t *out_len)
{
    int rc = unnaf_run(c, (const u8 *)d_naf, naf_len, o, out_begin, out_end, false, (u8 *)d_out, out_cap, out_len, false);
    if (c) arena_settle(c);
    return rc;
}

// ---- byte histogram (unnaf --charcount, output.c:515-605) ------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_histogram(const u8 *p, u64 n, unsigned long long *counts)
{
    __shared__ u32 h[4][256];                                     // one copy per wavefront: fewer same-address LDS atomics
    for (u32 i = threadIdx.x; i < 1024; i += 256) ((u32 *)h)[i] = 0;
    __syncthreads();
    u32 *my = h[threadIdx.x >> 6];
    const u64 per = 65536;                                       // bytes per workgroup
    u64 lo = (u64)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (u64 i = lo + (u64)threadIdx.x * 8; i < hi; i += 256 * 8) {
        if (i + 8 <= hi) { u64 w = ld64(p + i); for (int k = 0; k < 8; k++) atomicAdd(&my[(w >> (8 * k)) & 0xFF], 1u); }
        else for (u64 k = i; k < hi; k++) atomicAdd(&my[p[k]], 1u);
    }
    __syncthreads();
    u32 v = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
    if (v) atomicAdd(&counts[threadIdx.x], (unsigned long long)v);
}
extern "C" int naf_gpu_histogram(naf_gpu_ctx *c, const void *d_buf, size_t n, uint64_t counts[256])
{
    if (!c || !counts || (!d_buf && n)) return NAF_GPU_EARG;
    arena_reset(c);
    unsigned long long *d = arena_new<unsigned long long>(c, 256); if (!d) return NAF_GPU_ENOMEM;
    HIP_TRY(c, hipMemsetAsync(d, 0, 256 * 8, c->stream));
    if (n) LAUNCH(c, "histogram", k_histogram, cdiv(n, 65536), 256, 0, (const u8 *)d_buf, (u64)n, d);
    return ctx_readback(c, counts, d, 256 * 8);
}
