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

// Inclusive scan of one value per thread across the 256-thread workgroup; returns inclusive result,
// *total = workgroup aggregate.
template <typename T, typename Op>
__device__ __forceinline__ T wg_scan_inclusive(T v, T *total, T *lds /* 4 entries */)
{
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { T o = shfl_up_t(v, d); if (lane >= d) v = Op::template f<T>(o, v); }
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    T pre = Op::template id<T>(), tot = Op::template id<T>();
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) { T x = lds[w]; if (w < wave) pre = Op::template f<T>(pre, x); tot = Op::template f<T>(tot, x); }
    __syncthreads();
    *total = tot;
    return Op::template f<T>(pre, v);
}

