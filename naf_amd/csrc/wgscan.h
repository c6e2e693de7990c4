// wgscan.h -- workgroup-level scan primitives shared by scan.hip and the emit / encode kernels.
#pragma once
#include "common.h"

#define SCAN_THREADS 256

struct OpAdd { template <typename T> __device__ static T id() { return (T)0; } template <typename T> __device__ static T f(T a, T b) { return a + b; } };
struct OpMax { template <typename T> __device__ static T id() { return (T)(-2147483647 - 1); } template <typename T> __device__ static T f(T a, T b) { return a > b ? a : b; } };

template <typename T> __device__ __forceinline__ T shfl_up_t(T v, int d)
{
    if constexpr (sizeof(T) == 8) {
        u32 lo = (u32)(u64)v, hi = (u32)((u64)v >> 32);
        lo = __shfl_up(lo, d, 64); hi = __shfl_up(hi, d, 64);
        return (T)(((u64)hi << 32) | lo);
    } else return (T)__shfl_up((int)v, d, 64);
}
template <typename T> __device__ __forceinline__ T shfl_idx_t(T v, int l)
{
    if constexpr (sizeof(T) == 8) {
        u32 lo = (u32)(u64)v, hi = (u32)((u64)v >> 32);
        lo = __shfl(lo, l, 64); hi = __shfl(hi, l, 64);
        return (T)(((u64)hi << 32) | lo);
    } else return (T)__shfl((int)v, l, 64);
}

// One DPP hop of the wave scan: lanes without a source keep `idv` (the operator's identity).
template <int CTRL, int ROW_MASK, typename T> __device__ __forceinline__ T dpp_hop(T idv, T v)
{
    if constexpr (sizeof(T) == 8) {
        u32 lo = (u32)__builtin_amdgcn_update_dpp((int)(u32)(u64)idv, (int)(u32)(u64)v, CTRL, ROW_MASK, 0xf, false);
        u32 hi = (u32)__builtin_amdgcn_update_dpp((int)(u32)((u64)idv >> 32), (int)(u32)((u64)v >> 32), CTRL, ROW_MASK, 0xf, false);
        return (T)(((u64)hi << 32) | lo);
    } else return (T)__builtin_amdgcn_update_dpp((int)idv, (int)v, CTRL, ROW_MASK, 0xf, false);
}
// Inclusive scan across the 64 lanes of a wavefront on the DPP path (no LDS crossbar): shifts of 1, 2, 4, 8 inside each
// row of 16 lanes, then lane 15 of rows 0 / 2 broadcast into rows 1 / 3, then lane 31 into the upper half.
template <typename T, typename Op> __device__ __forceinline__ T wave_scan_inclusive(T v)
{
    const T idv = Op::template id<T>();
    v = Op::template f<T>(dpp_hop<0x111, 0xf>(idv, v), v);
    v = Op::template f<T>(dpp_hop<0x112, 0xf>(idv, v), v);
    v = Op::template f<T>(dpp_hop<0x114, 0xf>(idv, v), v);
    v = Op::template f<T>(dpp_hop<0x118, 0xf>(idv, v), v);
    v = Op::template f<T>(dpp_hop<0x142, 0xa>(idv, v), v);
    v = Op::template f<T>(dpp_hop<0x143, 0xc>(idv, v), v);
    return v;
}

// Two 16-bit maxima in one 32-bit value (v_pk_max_u16).
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
struct OpPkMaxU16 {
    template <typename T> __device__ static T id() { return (T)0; }
    template <typename T> __device__ static T f(T a, T b) { return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, (u32)a), __builtin_bit_cast(u16x2_t, (u32)b))); }
};
// lane i <- lane i-1 of the wavefront; lane 0 gets the identity
template <typename T, typename Op> __device__ __forceinline__ T wave_shift_up1(T v) { return dpp_hop<0x138, 0xf>(Op::template id<T>(), v); }

// Workgroup scan with ONE barrier: returns the wave-inclusive value, *pre = aggregate of the earlier waves (inclusive result =
// f(*pre, return)), *total = workgroup aggregate.  `slots` (4 entries) must not be written again before another barrier.
template <typename T, typename Op>
__device__ __forceinline__ T wg_scan1(T v, T *pre_out, T *total, T *slots)
{
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_scan_inclusive<T, Op>(v);
    if (lane == 63) slots[wave] = v;
    __syncthreads();
    T pre = Op::template id<T>(), tot = Op::template id<T>();
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) { T x = slots[w]; if (w < wave) pre = Op::template f<T>(pre, x); tot = Op::template f<T>(tot, x); }
    *pre_out = pre; *total = tot;
    return v;
}
template <typename T, typename Op> __device__ __forceinline__ T wg_reduce1(T v, T *slots) { T pre, tot; wg_scan1<T, Op>(v, &pre, &tot, slots); return tot; }

// Inclusive scan of one value per thread across the 256-thread workgroup; returns inclusive result,
// *total = workgroup aggregate.
template <typename T, typename Op>
__device__ __forceinline__ T wg_scan_inclusive(T v, T *total, T *lds /* 4 entries */)
{
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_scan_inclusive<T, Op>(v);
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    T pre = Op::template id<T>(), tot = Op::template id<T>();
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) { T x = lds[w]; if (w < wave) pre = Op::template f<T>(pre, x); tot = Op::template f<T>(tot, x); }
    __syncthreads();
    *total = tot;
    return Op::template f<T>(pre, v);
}
