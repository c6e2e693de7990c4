// enc_swar.h -- byte classes of a 16-byte piece of FASTA / FASTQ text, four bytes per instruction.
// The classes are those of the reference's tables (ennaf/src/tables.c:28-36 is_eol / is_space, '>' as the record marker,
// tables.c:72-123 the accepted sequence letters, process.c:522-528 the quality range).  Compiles for the host as well so that
// tests/emul can check every byte value at every position against the plain predicates.
#pragma once
#include "common.h"

#if defined(__HIP_DEVICE_COMPILE__)
NAF_HD u32 swar_dot4(u32 a, u32 b, u32 c) { return __builtin_amdgcn_udot4(a, b, c, false); }
NAF_HD u32 swar_perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#else
NAF_HD u32 swar_dot4(u32 a, u32 b, u32 c) { for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xFF) * ((b >> (8 * i)) & 0xFF); return c; }
NAF_HD u32 swar_perm(u32 hi, u32 lo, u32 sel)
{
    u64 src = ((u64)hi << 32) | lo; u32 r = 0;
    for (int i = 0; i < 4; i++) r |= (u32)((src >> (8 * ((sel >> (8 * i)) & 7))) & 0xFF) << (8 * i);
    return r;
}
#endif

// bit 7 of each byte of f0..f3 (0x80 or 0) -> one bit per byte, byte 0 of f0 first
NAF_HD u32 swar_movemask16(u32 f0, u32 f1, u32 f2, u32 f3)
{
    u32 lo = swar_dot4(f1, 0x80402010u, swar_dot4(f0, 0x08040201u, 0));
    u32 hi = swar_dot4(f3, 0x80402010u, swar_dot4(f2, 0x08040201u, 0));
    return (lo >> 7) | ((hi >> 7) << 8);
}

struct PieceFlags { u32 eol, sp, gt; };          // one bit per byte: 0x0A..0x0D / 0x09..0x0D and 0x20 / '>'
NAF_HD PieceFlags piece_flags(const u32 w[4])
{
    const u32 H = 0x80808080u, L = 0x7F7F7F7Fu;
    u32 fe[4], fs[4], fg[4];
    for (int i = 0; i < 4; i++) {
        u32 x = w[i], l = x & L;
        u32 low = ~x & ~(l + 0x72727272u) & H;                  // byte < 14
        fe[i] = (l + 0x76767676u) & low;                          // 10..13
        u32 y = x ^ 0x20202020u, z = x ^ 0x3E3E3E3Eu;
        fs[i] = ((l + 0x77777777u) & low) | (~(((y & L) + L) | y) & H);
        fg[i] = ~(((z & L) + L) | z) & H;
    }
    PieceFlags f;
    f.eol = swar_movemask16(fe[0], fe[1], fe[2], fe[3]);
    f.sp = swar_movemask16(fs[0], fs[1], fs[2], fs[3]);
    f.gt = swar_movemask16(fg[0], fg[1], fg[2], fg[3]);
    return f;
}

// The quick table: the accepted letter (upper case) whose bits 1..3 equal the slot, 0xFF in unused slots; built by the host from
// the accepted set out of A C G T U N.  A byte passes when the table entry selected by its own bits 1..3 equals the byte with
// bit 5 (the case bit) cleared: a sufficient test for "accepted" that needs no per-byte lookup.  Returns one bit per byte that
// FAILS it; those bytes (a few IUPAC letters in sequence lines) are then looked up in the full table.
NAF_HD u32 piece_not_quick(const u32 w[4], u32 qlo, u32 qhi)
{
    const u32 H = 0x80808080u, L = 0x7F7F7F7Fu;
    u32 nz[4];
    for (int i = 0; i < 4; i++) {
        u32 x = w[i], r = swar_perm(qhi, qlo, (x >> 1) & 0x07070707u), d = r ^ (x & 0xDFDFDFDFu);
        nz[i] = (((d & L) + L) | d) & H;
    }
    return swar_movemask16(nz[0], nz[1], nz[2], nz[3]);
}

// Plain sequence text: every byte of the piece is a quick letter (either case) or '\n' / '\r'.  The quick table with '\n' (slot 5:
// (0x0A >> 1) & 7) and '\r' (slot 6) in two of its free slots answers in one look-up; the case bit is cleared only in bytes that
// have bit 6, so '*' (0x2A) is not taken for 0x0A.  *eol: one bit per '\n' / '\r' (the passing bytes without bit 6); meaningful
// only when the function returns true.
NAF_HD bool piece_plain(const u32 w[4], u32 plo, u32 phi, u32 *eol)
{
    const u32 H = 0x80808080u, L = 0x7F7F7F7Fu;
    u32 bad = 0, e[4];
    for (int i = 0; i < 4; i++) {
        const u32 x = w[i], r = swar_perm(phi, plo, (x >> 1) & 0x07070707u), d = r ^ (x & ~((x >> 1) & 0x20202020u));
        bad |= ((d & L) + L) | d;
        e[i] = ~(x << 1) & H;
    }
    *eol = swar_movemask16(e[0], e[1], e[2], e[3]);
    return (bad & H) == 0;
}

// every byte in 0x21..0x7E (the quality characters stored as they are, process.c:522-528)
NAF_HD bool piece_all_quality(const u32 w[4])
{
    const u32 H = 0x80808080u, L = 0x7F7F7F7Fu;
    u32 ok = H;
    for (int i = 0; i < 4; i++) { u32 x = w[i], l = x & L; ok &= (l + 0x5F5F5F5Fu) & ~(l + 0x01010101u) & ~x; }
    return (ok & H) == H;
}

// one bit per byte that is a control character for header text: < 0x20, 0x7F or 0xFF (tables.c: the bytes replaced by '?' in
// comments; in IDs the space 0x20 joins them, but a space ends the ID before it could be part of it)
NAF_HD u32 piece_ctl_mask(const u32 w[4])
{
    const u32 H = 0x80808080u, L = 0x7F7F7F7Fu;
    u32 f[4];
    for (int i = 0; i < 4; i++) {
        u32 x = w[i], l = x & L;
        u32 lt20 = ~x & ~(l + 0x60606060u) & H;                  // < 128 and low seven bits < 0x20
        u32 y = x ^ 0x7F7F7F7Fu, z = ~x;                          // zero bytes where x is 0x7F / 0xFF
        f[i] = lt20 | (~(((y & L) + L) | y) & H) | (~(((z & L) + L) | z) & H);
    }
    return swar_movemask16(f[0], f[1], f[2], f[3]);
}

// one bit per byte outside 0x21..0x7E
NAF_HD u32 piece_not_quality_mask(const u32 w[4])
{
    const u32 H = 0x80808080u, L = 0x7F7F7F7Fu;
    u32 f[4];
    for (int i = 0; i < 4; i++) { u32 x = w[i], l = x & L; f[i] = ~((l + 0x5F5F5F5Fu) & ~(l + 0x01010101u) & ~x) & H; }
    return swar_movemask16(f[0], f[1], f[2], f[3]);
}
