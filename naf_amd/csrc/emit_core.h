// emit_core.h -- text-geometry parameters and byte-shuffle helpers shared by the emit kernel (emit.hip) and
// the fused Huffman-decode+emit kernel (zstd_dec.hip).
#pragma once
#include "common.h"

enum { EM_FASTA = 0, EM_FASTQ = 1, EM_SEQ = 2, EM_SEQUENCES = 3 };

struct EmitP {
    // text geometry
    const u64 *rec_out, *rec_base;     // N+1 entries each
    const u64 *rec_len;                // N
    const u32 *hdr_len;                // N   (0 for EM_SEQUENCES / EM_SEQ)
    const u64 *idz, *nmz;              // positions of the '\0' terminators in ids / names (N each)
    const u8 *ids, *names;
    const u8 *seq;                     // packed 4-bit codes, or text bytes
    const u8 *qual;
    const u64 *hdr_off; const u8 *hdr_text; u64 hdr_r0;   // short-record path: header lines of records >= hdr_r0 as one stream
    const u64 *toggles; u64 n_toggles;
    u64 N, T, L;
    u64 out_begin, out_end;            // byte range of the full text to produce; out[0] = byte out_begin
    u32 lut[4];                        // 16-entry code -> ASCII table as four dwords
    u32 Ldiv_magic;                    // unused when L+1 >= 2^32
    int mode, has_ids, has_names, masking, upper;
    u8 sep, hdr_char;
    int force_slow;
    int nt_store;                                              // NAF_GPU_EMIT_NT=1: k_emit_tile_flat stores the text with the nontemporal hint
    // a flat frame read in place (ctx.h: ZFlat): stream table, source, code -> packed byte; fsrc == nullptr otherwise
    const u8 *ftail; u64 ftail_q; u32 ftail_n;   // the frame's final Raw block, if it has one (ZFlat)
    const u8 *fsrc; const void *fsi; u64 fslots; const u8 *fsym; const u32 *fpair;
    const u8 *fcls;                    // per block of a mostly-flat frame: bit 0 = readable in place, bit 1 = decoded into `seq` (ctx.h: ZFlat); nullptr = all flat    // fpair: the 16-entry table code -> packed byte as four dwords (for v_perm_b32)
};


__device__ __forceinline__ u64 upper_bound_u64(const u64 *a, u64 lo, u64 hi, u64 v)   // first index in [lo,hi) with a[i] > v
{
    while (lo < hi) { u64 mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}


// 16 four-bit codes (nibble i = base i) -> 16 ASCII bytes in two u64, through the 16-entry table P.lut
// (unnaf.c:13 "-TGKCYSBAWRDMHVN", 'U' for RNA) held in four SGPR dwords and v_perm_b32.
// Eight codes at a time: the even and the odd nibbles of a 32-bit word are each four bytes already (x & 0x0F0F0F0F,
// (x >> 4) & 0x0F0F0F0F); both go through the table and a last v_perm_b32 interleaves the two results.
__device__ __forceinline__ u32 expand_codes4(const u32 lut[4], u32 v)      // four codes, one per byte -> four ASCII bytes
{
    const u32 sel = v & 0x07070707u;
    const u32 l = __builtin_amdgcn_perm(lut[1], lut[0], sel);              // codes 0..7
    const u32 h = __builtin_amdgcn_perm(lut[3], lut[2], sel);              // codes 8..15
    const u32 m = __builtin_amdgcn_perm(0u, 0x0000FF00u, (v >> 3) & 0x01010101u);   // 0xFF in the bytes whose code is >= 8
    return (l & ~m) | (h & m);
}
__device__ __forceinline__ void expand16(const u32 lut[4], u64 nib, u64 &lo, u64 &hi)
{
    u32 out[4];
#pragma unroll
    for (int w = 0; w < 2; w++) {
        const u32 x = (u32)(nib >> (32 * w));
        const u32 E = expand_codes4(lut, x & 0x0F0F0F0Fu), O = expand_codes4(lut, (x >> 4) & 0x0F0F0F0Fu);   // bases 0,2,4,6 / 1,3,5,7 of the word
        out[2 * w] = __builtin_amdgcn_perm(O, E, 0x05010400u);             // E0 O0 E1 O1
        out[2 * w + 1] = __builtin_amdgcn_perm(O, E, 0x07030602u);         // E2 O2 E3 O3
    }
    lo = (u64)out[0] | ((u64)out[1] << 32); hi = (u64)out[2] | ((u64)out[3] << 32);
}

__device__ __forceinline__ u64 spread_bits8(u32 b)            // bit i of b -> 0x20 in byte i
{
    u64 r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r |= (u64)((b >> i) & 1) << (8 * i + 5);
    return r;
}

__device__ __forceinline__ u64 low_bytes_mask(int n)           // n in [0,8] -> lowest n bytes set
{
    return n >= 8 ? ~0ull : ((1ull << (8 * n)) - 1);
}


// Insert '\n' at byte index nl (0..15) of the 16-byte value {lo,hi}: bytes < nl stay, bytes > nl take the previous
// byte; the displaced last byte is returned (it becomes byte 16 of a 17-byte output).
__device__ __forceinline__ u32 splice_newline(u64 &lo, u64 &hi, int nl)
{
    u32 last = (u32)(hi >> 56);
    u64 slo = lo << 8, shi = (hi << 8) | (lo >> 56);
    u64 mlo = low_bytes_mask(nl), mhi = nl > 8 ? low_bytes_mask(nl - 8) : 0;
    u64 m1lo = low_bytes_mask(nl + 1), m1hi = nl + 1 > 8 ? low_bytes_mask(nl + 1 - 8) : 0;
    lo = (lo & mlo) | (slo & ~m1lo); hi = (hi & mhi) | (shi & ~m1hi);
    if (nl < 8) lo |= (u64)'\n' << (8 * nl); else hi |= (u64)'\n' << (8 * (nl - 8));
    return last;
}
