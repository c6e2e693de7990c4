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
};


__device__ __forceinline__ u64 upper_bound_u64(const u64 *a, u64 lo, u64 hi, u64 v)   // first index in [lo,hi) with a[i] > v
{
    while (lo < hi) { u64 mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}


// 16 four-bit codes (nibble i = base i) -> 16 ASCII bytes in two u64, through the 16-entry table P.lut
// (unnaf.c:13 "-TGKCYSBAWRDMHVN", 'U' for RNA) held in four SGPR dwords and v_perm_b32.
__device__ __forceinline__ void expand16(const u32 lut[4], u64 nib, u64 &lo, u64 &hi)
{
    u32 out[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        u32 t = (u32)(nib >> (16 * w)) & 0xFFFF;                // 4 nibbles n3n2n1n0
        u32 y = (t | (t << 8)) & 0x00FF00FF;
        u32 z = (y | (y << 4)) & 0x0F0F0F0F;                    // one nibble per byte
        u32 sel = z & 0x07070707;
        u32 l = __builtin_amdgcn_perm(lut[1], lut[0], sel);     // codes 0..7
        u32 h = __builtin_amdgcn_perm(lut[3], lut[2], sel);     // codes 8..15
        u32 b3 = (z >> 3) & 0x01010101, m = (b3 << 8) - b3;          // 0xFF in the bytes whose code is >= 8 (a multiply by 0xFF is quarter rate)
        out[w] = (l & ~m) | (h & m);
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
