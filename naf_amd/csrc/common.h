// common.h -- shared types and bit-stream readers for the gfx950 kernels of libnaf_gpu.
// Functions marked NAF_HD also compile for the host so tests/emul can single-step kernel logic
// without a GPU (development harness only -- the product has no CPU path).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NAF_HD __host__ __device__ __forceinline__
#define NAF_HDM __host__ __device__ __forceinline__      /* member functions */
#define NAF_D __device__ __forceinline__
#else
#define NAF_HD static inline
#define NAF_HDM inline
#endif

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;
typedef int16_t i16;

// gfx950 handles unaligned global accesses in hardware; memcpy of a fixed size lowers to one load.
NAF_HD u64 ld64(const u8 *p) { u64 v; memcpy(&v, p, 8); return v; }
NAF_HD u32 ld32(const u8 *p) { u32 v; memcpy(&v, p, 4); return v; }
NAF_HD u32 ld24(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16); }
NAF_HD u32 ld16(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8); }
NAF_HD void st64(u8 *p, u64 v) { memcpy(p, &v, 8); }
NAF_HD void st32(u8 *p, u32 v) { memcpy(p, &v, 4); }

// A load from GLOBAL memory at an address held as an integer.  A plain pointer made from an integer is a flat pointer, and flat
// loads count against lgkmcnt as well as vmcnt: every wait for LDS or scalar data would wait for them too.
template <typename T> NAF_HD T ldg_at(u64 addr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const __attribute__((address_space(1))) T *)addr;
#else
    return *(const T *)addr;
#endif
}
// The same for an address of any alignment.  (A naturally-aligned type promises the compiler its alignment: a load whose address
// is the same in every lane becomes a scalar load, and scalar loads ignore the low two address bits.)
template <typename T> NAF_HD T ldg_at_unaligned(u64 addr)
{
    typedef T __attribute__((aligned(1))) T1;
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const __attribute__((address_space(1))) T1 *)addr;
#else
    return *(const T1 *)addr;
#endif
}
NAF_HD int hibit32(u32 v) { return 31 - __builtin_clz(v); }

// ---- backward bit reader (RFC 8878 4.1 "bitstreams are read backward") -----------------------------
// c holds the 8 bytes at ptr (little endian); `consumed` counts bits used from the top of c.
struct BitR {
    const u8 *start, *ptr;
    u64 c;
    u32 consumed;
    bool bad;
};

NAF_HD void bitr_init(BitR &b, const u8 *src, u32 len)
{
    b.start = src; b.bad = false;
    if (len == 0) { b.bad = true; b.ptr = src; b.c = 0; b.consumed = 64; return; }
    u32 last = src[len - 1];
    if (last == 0) { b.bad = true; last = 1; }
    if (len >= 8) {
        b.ptr = src + len - 8; b.c = ld64(b.ptr);
        b.consumed = 8 - (u32)hibit32(last);
    } else {
        u64 c = 0;
        for (u32 i = 0; i < len; i++) c |= (u64)src[i] << (8 * i);
        b.ptr = src; b.c = c;
        b.consumed = 8 - (u32)hibit32(last) + (8 - len) * 8;
    }
}
// Top n unread bits (1 <= n <= 32); positions below the start of the stream read as 0.
NAF_HD u32 bitr_peek(const BitR &b, u32 n)
{
    u64 t = b.consumed < 64 ? (b.c << b.consumed) : 0;
    return (u32)(t >> (64 - n));
}
NAF_HD void bitr_skip(BitR &b, u32 n) { b.consumed += n; }
NAF_HD u32 bitr_read(BitR &b, u32 n)
{
    if (n == 0) return 0;
    u32 v = bitr_peek(b, n); b.consumed += n; return v;
}
// Make at least 57 bits available again (fewer once the start of the stream is reached).
NAF_HD void bitr_reload(BitR &b)
{
    u32 bytes = b.consumed >> 3;
    u32 avail = (u32)(b.ptr - b.start);
    if (bytes > avail) bytes = avail;
    if (bytes) { b.ptr -= bytes; b.consumed -= bytes * 8; b.c = ld64(b.ptr); }
}
// Exactly all bits consumed?  (consumed==64 with ptr at start; short streams never move ptr)
NAF_HD bool bitr_finished(const BitR &b) { return b.ptr == b.start && b.consumed == 64; }
NAF_HD bool bitr_overrun(const BitR &b) { return b.consumed > 64; }

// ---- forward bit reader (FSE table descriptions) -----------------------------------------------------
struct BitF { const u8 *p; u32 len; u32 bitpos; };
NAF_HD u32 bitf_peek(const BitF &b, u32 n)
{
    u32 byte = b.bitpos >> 3; u64 v = 0;
    for (u32 i = 0; i < 4; i++) if (byte + i < b.len) v |= (u64)b.p[byte + i] << (8 * i);
    return (u32)((v >> (b.bitpos & 7)) & ((1u << n) - 1));
}

#if defined(__HIPCC__)
// Workgroups are dealt to the eight XCDs in turn (workgroup b runs on XCD b % 8: MI355X_MICROARCH.md), each XCD with an L2 of its own.
// xcd_block() renumbers a launch's workgroups so that every XCD walks ONE contiguous eighth of the grid: neighbouring units of work --
// tiles of a text, blocks of a frame -- whose outputs share cache lines (unaligned stream pieces, eight-byte table entries) then meet in
// one L2 instead of leaving partial lines in eight.  A bijection on [0, gridDim.x) whatever the grid.  (-DNAF_NO_XCD_MAP: launch order.)
__device__ __forceinline__ u32 xcd_block()
{
#if defined(NAF_NO_XCD_MAP)
    return blockIdx.x;
#else
    const u32 G = gridDim.x, b = blockIdx.x, q = G >> 3, rem = G & 7u, x = b & 7u;
    return x * q + (x < rem ? x : rem) + (b >> 3);
#endif
}
#endif
