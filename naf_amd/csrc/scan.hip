// scan.hip -- device-wide prefix scans (reduce-then-scan, 2048-element tiles, recursive on tile sums).
// Used for block output offsets, table ownership, record / header / mask-run offsets.
#include "ctx.h"
#include "wgscan.h"

#define SCAN_ITEMS   8
#define SCAN_TILE    (SCAN_THREADS * SCAN_ITEMS)

template <typename T, typename Op>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_reduce(const T *vals, size_t n, T *sums)
{
    __shared__ T lds[4];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    T acc = Op::template id<T>();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) if (base + i < n) acc = Op::template f<T>(acc, vals[base + i]);
    T tot; wg_scan_inclusive<T, Op>(acc, &tot, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// EXCL: exclusive (sum) or inclusive (max) result written in place; tile_pre = exclusive scan of tile sums.
template <typename T, typename Op, bool EXCL>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_apply(T *vals, size_t n, const T *tile_pre)
{
    __shared__ T lds[4];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS]; T acc = Op::template id<T>();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) { v[i] = base + i < n ? vals[base + i] : Op::template id<T>(); acc = Op::template f<T>(acc, v[i]); }
    T tot; T incl = wg_scan_inclusive<T, Op>(acc, &tot, lds);
    // exclusive prefix of this thread = inclusive of previous thread
    T prev = shfl_up_t(incl, 1);
    __shared__ T wave_last[4];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) wave_last[wave] = incl;
    __syncthreads();
    if (lane == 0) prev = wave == 0 ? Op::template id<T>() : wave_last[wave - 1];
    T run = tile_pre ? Op::template f<T>(tile_pre[blockIdx.x], prev) : prev;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (EXCL) { if (base + i < n) vals[base + i] = run; run = Op::template f<T>(run, v[i]); }
        else { run = Op::template f<T>(run, v[i]); if (base + i < n) vals[base + i] = run; }
    }
}

template <typename T> __global__ void k_store_total(const T *tile_excl, const T *tile_sum_last, size_t ntiles, T *total)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *total = tile_excl[ntiles - 1] + *tile_sum_last;
}

template <typename T, typename Op, bool EXCL>
static int scan_rec(naf_gpu_ctx *c, T *vals, size_t n, T *d_total)
{
    if (n == 0) { if (d_total) HIP_TRY(c, hipMemsetAsync(d_total, 0, sizeof(T), c->stream)); return 0; }
    size_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    T *sums = arena_new<T>(c, ntiles + 1);
    if (!sums) return ctx_fail(c, NAF_GPU_ENOMEM, "scan scratch");
    LAUNCH(c, "scan_reduce", (k_scan_tile_reduce<T, Op>), ntiles, SCAN_THREADS, 0, (const T *)vals, n, sums);
    T *tile_pre = nullptr;
    if (ntiles > 1 || d_total) {
        // exclusive scan of tile aggregates (recursive); keep the last aggregate for the total
        T *last = arena_new<T>(c, 1);
        if (!last) return ctx_fail(c, NAF_GPU_ENOMEM, "scan scratch");
        HIP_TRY(c, hipMemcpyAsync(last, sums + ntiles - 1, sizeof(T), hipMemcpyDeviceToDevice, c->stream));
        int rc = scan_rec<T, Op, true>(c, sums, ntiles, (T *)nullptr);
        if (rc) return rc;
        tile_pre = sums;
        if (d_total) LAUNCH(c, "scan_total", (k_store_total<T>), 1, 64, 0, (const T *)sums, (const T *)last, ntiles, d_total);
    }
    LAUNCH(c, "scan_apply", (k_scan_tile_apply<T, Op, EXCL>), ntiles, SCAN_THREADS, 0, vals, n, (const T *)tile_pre);
    return 0;
}

int scan_exclusive_u64(naf_gpu_ctx *c, u64 *d_vals, size_t n, u64 *d_total)
{
    return scan_rec<u64, OpAdd, true>(c, d_vals, n, d_total);
}

int scan_inclusive_max_i32(naf_gpu_ctx *c, i32 *d_vals, size_t n)
{
    // tile aggregates of a max-scan combine with max as well: reuse the recursion with OpMax/exclusive
    return scan_rec<i32, OpMax, false>(c, d_vals, n, (i32 *)nullptr);
}

int scan_inclusive_max_i64(naf_gpu_ctx *c, i64 *d_vals, size_t n)
{
    return scan_rec<i64, OpMax, false>(c, d_vals, n, (i64 *)nullptr);
}
