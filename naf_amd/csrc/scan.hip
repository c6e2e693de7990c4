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

// Up to SCAN_SMALL values in ONE launch by one workgroup (a thread takes a contiguous run of at most 16: its loads are dependent ones): the block tables of the small streams,
// the tile aggregates of the big ones.  A scan used to be five to seven launches whatever its size, and the front of a decode call is
// a chain of such launches.
#define SCAN_SMALL 4096
template <typename T, typename Op, bool EXCL>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_small(T *vals, size_t n, T *total)
{
    __shared__ T lds[4];
    const size_t per = (n + SCAN_THREADS - 1) / SCAN_THREADS, lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    T acc = Op::template id<T>();
    for (size_t k = lo; k < hi; k++) acc = Op::template f<T>(acc, vals[k]);
    T tot;
    const T incl = wg_scan_inclusive<T, Op>(acc, &tot, lds);
    // exclusive prefix of this thread's run: everything in front of it
    T run = Op::template id<T>();
    {
        __shared__ T s_incl[SCAN_THREADS];
        s_incl[threadIdx.x] = incl;
        __syncthreads();
        if (threadIdx.x) run = s_incl[threadIdx.x - 1];
    }
    for (size_t k = lo; k < hi; k++) {
        const T v = vals[k];
        if (EXCL) { vals[k] = run; run = Op::template f<T>(run, v); }
        else { run = Op::template f<T>(run, v); vals[k] = run; }
    }
    if (total && threadIdx.x == 0) *total = tot;
}

// total (optional): the aggregate of all values.  The tile aggregates are scanned by the same routine, and THEIR total is the grand total.
template <typename T, typename Op, bool EXCL>
static int scan_rec(naf_gpu_ctx *c, T *vals, size_t n, T *d_total)
{
    if (n == 0) { if (d_total) HIP_TRY(c, hipMemsetAsync(d_total, 0, sizeof(T), c->stream)); return 0; }
    if (n <= SCAN_SMALL) { LAUNCH(c, "scan_small", (k_scan_small<T, Op, EXCL>), 1, SCAN_THREADS, 0, vals, n, d_total); return 0; }
    size_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    T *sums = arena_new<T>(c, ntiles + 1);
    if (!sums) return ctx_fail(c, NAF_GPU_ENOMEM, "scan scratch");
    LAUNCH(c, "scan_reduce", (k_scan_tile_reduce<T, Op>), ntiles, SCAN_THREADS, 0, (const T *)vals, n, sums);
    int rc = scan_rec<T, Op, true>(c, sums, ntiles, d_total);      // exclusive scan of the tile aggregates
    if (rc) return rc;
    LAUNCH(c, "scan_apply", (k_scan_tile_apply<T, Op, EXCL>), ntiles, SCAN_THREADS, 0, vals, n, (const T *)sums);
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

// ---- several arrays of one length in ONE set of launches (blockIdx.y = the array) ------------------------------------------------------
// The encoder's split is followed by four or five scans over the tile table, each three launches of a few microseconds that wait for
// each other: fifteen launches as three.  Arrays of more than SCAN_SMALL tiles' aggregates (a 100 GB text) take the plain routine.
#define SCAN_MULTI_MAX 8
template <typename T> struct ScanSet { T *v[SCAN_MULTI_MAX]; T *sums[SCAN_MULTI_MAX]; T *tot[SCAN_MULTI_MAX]; };
template <typename T, typename Op>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_reduce_multi(ScanSet<T> S, size_t n)
{
    __shared__ T lds[4];
    const T *vals = S.v[blockIdx.y];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + threadIdx.x;
    T acc = Op::template id<T>();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) { size_t k = base + (size_t)i * SCAN_THREADS; if (k < n) acc = Op::template f<T>(acc, vals[k]); }
    T tot = wg_reduce1<T, Op>(acc, lds);
    if (threadIdx.x == 0) S.sums[blockIdx.y][blockIdx.x] = tot;
}
template <typename T, typename Op>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_small_multi(ScanSet<T> S, size_t n)      // exclusive scan of every array's tile aggregates
{
    __shared__ T lds[4];
    __shared__ T s_incl[SCAN_THREADS];
    T *vals = S.sums[blockIdx.x];
    const size_t per = (n + SCAN_THREADS - 1) / SCAN_THREADS, lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    T acc = Op::template id<T>();
    for (size_t k = lo; k < hi; k++) acc = Op::template f<T>(acc, vals[k]);
    T tot;
    const T incl = wg_scan_inclusive<T, Op>(acc, &tot, lds);
    s_incl[threadIdx.x] = incl;
    __syncthreads();
    T run = threadIdx.x ? s_incl[threadIdx.x - 1] : Op::template id<T>();
    for (size_t k = lo; k < hi; k++) { const T v = vals[k]; vals[k] = run; run = Op::template f<T>(run, v); }
    if (S.tot[blockIdx.x] && threadIdx.x == 0) *S.tot[blockIdx.x] = tot;
}
template <typename T, typename Op, bool EXCL>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_apply_multi(ScanSet<T> S, size_t n)
{
    __shared__ T lds[2][4];
    T *vals = S.v[blockIdx.y]; const T *tile_pre = S.sums[blockIdx.y];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + threadIdx.x;
    T v[SCAN_ITEMS];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) { size_t k = base + (size_t)i * SCAN_THREADS; v[i] = k < n ? vals[k] : Op::template id<T>(); }
    T run = tile_pre[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        T pre, tot;
        T wi = wg_scan1<T, Op>(v[i], &pre, &tot, lds[i & 1]);
        T r = Op::template f<T>(run, Op::template f<T>(pre, EXCL ? wave_shift_up1<T, Op>(wi) : wi));
        size_t k = base + (size_t)i * SCAN_THREADS;
        if (k < n) vals[k] = r;
        run = Op::template f<T>(run, tot);
    }
}
template <typename T, typename Op, bool EXCL>
static int scan_multi(naf_gpu_ctx *c, T *const *arrs, int k, size_t n, T *const *totals)
{
    const size_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (k < 2 || k > SCAN_MULTI_MAX || n <= SCAN_SMALL || ntiles > SCAN_SMALL) {
        for (int i = 0; i < k; i++) { int rc = scan_rec<T, Op, EXCL>(c, arrs[i], n, totals ? totals[i] : (T *)nullptr); if (rc) return rc; }
        return 0;
    }
    ScanSet<T> S; memset(&S, 0, sizeof S);
    for (int i = 0; i < k; i++) {
        S.v[i] = arrs[i]; S.tot[i] = totals ? totals[i] : nullptr;
        S.sums[i] = arena_new<T>(c, ntiles + 1); if (!S.sums[i]) return ctx_fail(c, NAF_GPU_ENOMEM, "scan scratch");
    }
    LAUNCH(c, "scan_reduce", (k_scan_tile_reduce_multi<T, Op>), dim3((unsigned)ntiles, (unsigned)k), SCAN_THREADS, 0, S, n);
    LAUNCH(c, "scan_small", (k_scan_small_multi<T, Op>), (unsigned)k, SCAN_THREADS, 0, S, ntiles);
    LAUNCH(c, "scan_apply", (k_scan_tile_apply_multi<T, Op, EXCL>), dim3((unsigned)ntiles, (unsigned)k), SCAN_THREADS, 0, S, n);
    return 0;
}
int scan_exclusive_u64_multi(naf_gpu_ctx *c, u64 *const *arrs, int k, size_t n, u64 *const *totals) { return scan_multi<u64, OpAdd, true>(c, arrs, k, n, totals); }
int scan_inclusive_max_i64_multi(naf_gpu_ctx *c, i64 *const *arrs, int k, size_t n) { return scan_multi<i64, OpMax, false>(c, arrs, k, n, (i64 *const *)nullptr); }
