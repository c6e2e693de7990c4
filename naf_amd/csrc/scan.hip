// scan.hip -- device-wide prefix scans (reduce-then-scan, 2048-element tiles, recursive on tile sums).
// Used for block output offsets, table ownership, record / header / mask-run offsets.
#include "ctx.h"
#include "wgscan.h"

#define SCAN_ITEMS   8
#define SCAN_TILE    (SCAN_THREADS * SCAN_ITEMS)

// Items are taken striped (item i of a thread is i * 256 + thread): every load and store of a wavefront is one contiguous
// 512-byte run.  (Eight consecutive items per thread made each access touch 64 lines.)
template <typename T, typename Op>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_reduce(const T *vals, size_t n, T *sums)
{
    __shared__ T lds[4];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + threadIdx.x;
    T acc = Op::template id<T>();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) { size_t k = base + (size_t)i * SCAN_THREADS; if (k < n) acc = Op::template f<T>(acc, vals[k]); }
    T tot = wg_reduce1<T, Op>(acc, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// EXCL: exclusive (sum) or inclusive (max) result written in place; tile_pre = exclusive scan of tile sums.
template <typename T, typename Op, bool EXCL>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_apply(T *vals, size_t n, const T *tile_pre)
{
    __shared__ T lds[2][4];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + threadIdx.x;
    T v[SCAN_ITEMS];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) { size_t k = base + (size_t)i * SCAN_THREADS; v[i] = k < n ? vals[k] : Op::template id<T>(); }
    T run = tile_pre ? tile_pre[blockIdx.x] : Op::template id<T>();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {                     // one 256-wide scan per stripe; the slot pairs alternate, so one barrier each
        T pre, tot;
        T wi = wg_scan1<T, Op>(v[i], &pre, &tot, lds[i & 1]);
        T r = Op::template f<T>(run, Op::template f<T>(pre, EXCL ? wave_shift_up1<T, Op>(wi) : wi));
        size_t k = base + (size_t)i * SCAN_THREADS;
        if (k < n) vals[k] = r;
        run = Op::template f<T>(run, tot);
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
