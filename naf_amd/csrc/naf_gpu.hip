// naf_gpu.hip -- context, scratch arena, error reporting, timing, memory helpers of libnaf_gpu.
#include "ctx.h"
#include <thread>
#include <mutex>
#include <condition_variable>
#include <string.h>
#include <map>

int ctx_fail(naf_gpu_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        va_list ap; va_start(ap, fmt);
        vsnprintf(c->err, sizeof c->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

extern "C" const char *naf_gpu_strerror(int code)
{
    switch (code) {
    case NAF_GPU_OK: return "ok";
    case NAF_GPU_ENODEV: return "no usable gfx950 device";
    case NAF_GPU_EHIP: return "HIP runtime error";
    case NAF_GPU_ENOMEM: return "out of device memory";
    case NAF_GPU_EFORMAT: return "malformed NAF container";
    case NAF_GPU_EZSTD: return "corrupt or unsupported zstd frame";
    case NAF_GPU_ECAP: return "output buffer too small";
    case NAF_GPU_EINPUT: return "invalid input text";
    case NAF_GPU_EARG: return "invalid argument";
    default: return "unknown error";
    }
}

extern "C" const char *naf_gpu_last_error(const naf_gpu_ctx *c) { return c ? c->err : "no context"; }

// ---- options ---------------------------------------------------------------------------------------------------------------------
extern char **environ;
static const naf_gpu_ctx *opt_root(const naf_gpu_ctx *c) { return c && c->root ? c->root : c; }
const char *ctx_opt(const naf_gpu_ctx *c, const char *name)
{
    const naf_gpu_ctx *r = opt_root(c);
    if (!r || !r->opts) return nullptr;
    for (const auto &kv : r->opts->kv) if (kv.first == name) return kv.second.c_str();
    return nullptr;
}
bool ctx_tracing(const naf_gpu_ctx *c) { const naf_gpu_ctx *r = opt_root(c); return r && r->opts && r->opts->tracing; }
static std::mutex trace_mutex;
void ctx_trace(naf_gpu_ctx *c, const char *fmt, ...)
{
    naf_gpu_ctx *r = (naf_gpu_ctx *)opt_root(c);
    if (!r || !r->opts || !r->opts->tracing) return;
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    std::lock_guard<std::mutex> g(trace_mutex);
    if (r->opts->trace.size() < (1u << 20)) r->opts->trace += buf;
}
static void opts_set(CtxOpts *o, const char *name, const char *value)
{
    for (size_t i = 0; i < o->kv.size(); i++) if (o->kv[i].first == name) {
        if (value) o->kv[i].second = value; else o->kv.erase(o->kv.begin() + i);
        goto done;
    }
    if (value) o->kv.emplace_back(name, value);
done:
    if (!strcmp(name, "TRACE")) o->tracing = value && value[0] == '1';
}
static void opts_from_environment(CtxOpts *o)
{
    for (char **e = environ; e && *e; e++) {
        if (strncmp(*e, "NAF_GPU_", 8) != 0) continue;
        const char *eq = strchr(*e, '=');
        if (!eq) continue;
        opts_set(o, std::string(*e + 8, (size_t)(eq - (*e + 8))).c_str(), eq + 1);
    }
}
extern "C" int naf_gpu_set_option(naf_gpu_ctx *c, const char *name, const char *value)
{
    if (!c || !c->opts || !name || !name[0]) return NAF_GPU_EARG;
    if (!strncmp(name, "NAF_GPU_", 8)) name += 8;
    opts_set(c->opts, name, value);
    return NAF_GPU_OK;
}
extern "C" const char *naf_gpu_get_trace(naf_gpu_ctx *c) { return c && c->opts ? c->opts->trace.c_str() : ""; }
extern "C" void naf_gpu_clear_trace(naf_gpu_ctx *c) { if (c && c->opts) c->opts->trace.clear(); }

extern "C" int naf_gpu_init(int device, naf_gpu_ctx **out)
{
    if (!out) return NAF_GPU_EARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return NAF_GPU_ENODEV;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return NAF_GPU_ENODEV;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return NAF_GPU_ENODEV;     // kernels are built for gfx950 only
    if (hipSetDevice(device) != hipSuccess) return NAF_GPU_ENODEV;
    naf_gpu_ctx *c = new naf_gpu_ctx();
    c->device = device;
    c->opts = new CtxOpts(); opts_from_environment(c->opts);          // the one look at the environment
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return NAF_GPU_EHIP; }
    c->own_stream = true;
    c->h_stage_cap = 1 << 16;
    if (hipHostMalloc((void **)&c->h_stage, c->h_stage_cap, hipHostMallocDefault) != hipSuccess) { hipStreamDestroy(c->stream); delete c; return NAF_GPU_ENOMEM; }
    int rc = zstd_init_tables(c);
    if (rc) { naf_gpu_shutdown(c); return rc; }
    // side contexts: share the device and the constant tables.  Their streams are made by a thread of their own while the caller goes on
    // (a stream is 10 - 45 ms of hipStreamCreate on this runtime, five of them were most of what naf_gpu_init took: rocprofv3 --hip-trace of
    // the CLIs); whoever needs a side context first waits for that thread (ctx_sides_ready).
    if (hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming) != hipSuccess) { naf_gpu_shutdown(c); return NAF_GPU_EHIP; }
    for (int k = 0; k < ZSPLIT_MAX + 2; k++) if (hipEventCreateWithFlags(&c->split_ev[k], hipEventDisableTiming) != hipSuccess) { naf_gpu_shutdown(c); return NAF_GPU_EHIP; }
    c->sides_rc = 0;
    auto make_sides = [c, device] {
        hipSetDevice(device);
        int prio_lo = 0, prio_hi = 0; hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        for (int k = 0; k < 4; k++) {
            naf_gpu_ctx *sc = new naf_gpu_ctx();
            sc->device = device; sc->d_predef = c->d_predef; sc->h_stage_cap = 1 << 16; sc->root = c;
            // highest priority: the side chains are many tiny kernels; behind the payload's bulk kernels they would only be scheduled
            // once those drain, and the call would end up waiting for them
            if (hipStreamCreateWithPriority(&sc->stream, hipStreamNonBlocking, prio_hi) != hipSuccess || hipHostMalloc((void **)&sc->h_stage, sc->h_stage_cap, hipHostMallocDefault) != hipSuccess) {
                if (sc->stream) hipStreamDestroy(sc->stream);
                delete sc; c->sides_rc = NAF_GPU_EHIP; return;
            }
            sc->own_stream = true;
            (k == 0 ? c->side : k == 1 ? c->side2 : k == 2 ? c->side3 : c->side4) = sc;
        }
    };
    try { c->sides_thread = new std::thread(make_sides); }
    catch (...) { c->sides_thread = nullptr; make_sides(); }       // no thread to be had: made here, as they used to be
    if (!c->sides_thread && c->sides_rc) { naf_gpu_shutdown(c); return NAF_GPU_EHIP; }
    *out = c;
    return NAF_GPU_OK;
}

// The side contexts are there (naf_gpu_init's thread has made them); an error of that thread becomes the caller's.
int ctx_sides_ready(naf_gpu_ctx *c)
{
    if (c->sides_thread) { std::thread *t = (std::thread *)c->sides_thread; t->join(); delete t; c->sides_thread = nullptr; }
    return c->sides_rc ? ctx_fail(c, c->sides_rc, "can't create the side streams") : 0;
}

struct HostWorker { std::thread th; std::mutex m; std::condition_variable cv; std::function<void()> job; int state = 0; /* 0 idle, 1 posted, 2 running */ bool quit = false; };
static void worker_main(HostWorker *w, int device)
{
    hipSetDevice(device);
    std::unique_lock<std::mutex> lk(w->m);
    for (;;) {
        w->cv.wait(lk, [&] { return w->state == 1 || w->quit; });
        if (w->quit) return;
        std::function<void()> f = std::move(w->job);
        w->state = 2;
        lk.unlock(); f(); lk.lock();
        w->state = 0;
        w->cv.notify_all();
    }
}
void ctx_worker_start(naf_gpu_ctx *x, std::function<void()> job)
{
    if (!x->worker) { x->worker = new HostWorker; x->worker->th = std::thread(worker_main, x->worker, x->device); }
    HostWorker *w = x->worker;
    { std::unique_lock<std::mutex> lk(w->m); w->cv.wait(lk, [&] { return w->state == 0; }); w->job = std::move(job); w->state = 1; }
    w->cv.notify_all();
}
void ctx_worker_join(naf_gpu_ctx *x)
{
    HostWorker *w = x->worker; if (!w) return;
    std::unique_lock<std::mutex> lk(w->m);
    w->cv.wait(lk, [&] { return w->state == 0; });
}
static void ctx_worker_stop(naf_gpu_ctx *x)
{
    HostWorker *w = x->worker; if (!w) return;
    { std::unique_lock<std::mutex> lk(w->m); w->cv.wait(lk, [&] { return w->state == 0; }); w->quit = true; }
    w->cv.notify_all();
    w->th.join();
    delete w; x->worker = nullptr;
}

extern "C" void naf_gpu_shutdown(naf_gpu_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    ctx_sides_ready(c);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (naf_gpu_ctx *sc : { c->side, c->side2, c->side3, c->side4 }) {
        if (!sc) continue;
        ctx_worker_stop(sc);
        hipStreamSynchronize(sc->stream);
        for (auto &ch : sc->chunks) hipFree(ch.base);
        for (auto e : sc->ev_pool) hipEventDestroy(e);
        if (sc->d_seqctab) hipFree(sc->d_seqctab);
        if (sc->h_stage) hipHostFree(sc->h_stage);
        if (sc->own_stream) hipStreamDestroy(sc->stream);
        delete sc;
    }
    ennaf_shard_state_free(c);
    io_pool_free(c);
    if (c->fork_ev) hipEventDestroy(c->fork_ev);
    for (int k = 0; k < ZSPLIT_MAX + 2; k++) if (c->split_ev[k]) hipEventDestroy(c->split_ev[k]);
    for (auto &ch : c->chunks) hipFree(ch.base);
    for (auto e : c->ev_pool) hipEventDestroy(e);
    if (c->d_predef) hipFree(c->d_predef);
    if (c->d_seqctab) hipFree(c->d_seqctab);
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c->opts;
    delete c;
}

extern "C" int naf_gpu_set_stream(naf_gpu_ctx *c, void *s)
{
    // s is a hipStream_t; NULL is HIP's default (null) stream, exactly as in the HIP API.
    if (!c) return NAF_GPU_EARG;
    hipStreamSynchronize(c->stream);
    if (c->own_stream) hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)s; c->own_stream = false;
    return 0;
}

extern "C" int naf_gpu_synchronize(naf_gpu_ctx *c) { if (!c) return NAF_GPU_EARG; HIP_TRY(c, hipStreamSynchronize(c->stream)); return 0; }

// ---- arena -----------------------------------------------------------------------------------------------
static const size_t ARENA_ALIGN = 256;
static const size_t ARENA_MIN_CHUNK = (size_t)64 << 20;

void arena_reset(naf_gpu_ctx *c)
{
    // every entry point starts here: the calling thread's current device is made the context's (a host that drives several
    // contexts from one thread, or has just shut another device's context down, would otherwise allocate and launch on that one)
    hipSetDevice(c->device);
    size_t total = 0;
    for (auto &ch : c->chunks) { total += ch.cap; ch.used = 0; }
    if (c->chunks.size() > 1) {                      // consolidate so steady state is one chunk, no hipMalloc
        hipStreamSynchronize(c->stream);
        for (auto &ch : c->chunks) hipFree(ch.base);
        c->chunks.clear();
        u8 *p = nullptr;
        if (hipMalloc((void **)&p, total) == hipSuccess) c->chunks.push_back({ p, total, 0 });
    }
}

// End of a whole call (unnaf, ennaf, shard_finish): an arena that grew during the call -- the main context's or a side context's -- is
// made one allocation NOW, so that the growing call pays for its growth and the next call starts in steady state (the consolidation
// used to happen at the next call's reset: the first TWO calls on a context were slow, and a caller that timed from the second one
// on measured a hipFree + hipMalloc of the whole high-water mark).  Kernels of the main stream read what the side contexts made, so
// every stream of the context is waited for first.
void arena_settle(naf_gpu_ctx *c)
{
    // (while naf_gpu_init's thread is still making the side contexts nobody has used them: they hold nothing)
    naf_gpu_ctx *all[5] = { c, nullptr, nullptr, nullptr, nullptr };
    if (!c->sides_thread) { all[1] = c->side; all[2] = c->side2; all[3] = c->side3; all[4] = c->side4; }
    bool any = false;
    for (naf_gpu_ctx *x : all) if (x && x->chunks.size() > 1) any = true;
    if (!any) return;
    hipSetDevice(c->device);
    for (naf_gpu_ctx *x : all) if (x) hipStreamSynchronize(x->stream);
    for (naf_gpu_ctx *x : all) {
        if (!x || x->chunks.size() <= 1) continue;
        size_t total = 0;
        for (auto &ch : x->chunks) total += ch.cap;
        // the new block first when the device has room for both (a failure then changes nothing); else the chunks go first, and a
        // failure of the one allocation -- the memory was just there -- is left in the context's error text: the next call grows again
        u8 *p = nullptr;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr > total + ((size_t)1 << 30) && hipMalloc((void **)&p, total) == hipSuccess) {
            for (auto &ch : x->chunks) hipFree(ch.base);
            x->chunks.clear();
            x->chunks.push_back({ p, total, 0 });
            continue;
        }
        (void)hipGetLastError();
        for (auto &ch : x->chunks) hipFree(ch.base);
        x->chunks.clear();
        if (hipMalloc((void **)&p, total) == hipSuccess) x->chunks.push_back({ p, total, 0 });
        else { (void)hipGetLastError(); ctx_fail(c, NAF_GPU_ENOMEM, "arena consolidation: hipMalloc(%zu) failed after the call (the next call allocates again)", total); }
    }
}

void *arena_alloc(naf_gpu_ctx *c, size_t bytes)
{
    bytes = (bytes + ARENA_ALIGN - 1) & ~(ARENA_ALIGN - 1);
    if (bytes == 0) bytes = ARENA_ALIGN;
    for (auto &ch : c->chunks)
        if (ch.cap - ch.used >= bytes) { void *p = ch.base + ch.used; ch.used += bytes; return p; }
    size_t cap = bytes > ARENA_MIN_CHUNK ? bytes : ARENA_MIN_CHUNK;
    u8 *p = nullptr;
    if (hipMalloc((void **)&p, cap) != hipSuccess) { ctx_fail(c, NAF_GPU_ENOMEM, "hipMalloc(%zu) failed", cap); return nullptr; }
    c->chunks.push_back({ p, cap, bytes });
    return p;
}

extern "C" int naf_gpu_mem_info(naf_gpu_ctx *c, size_t *free_bytes, size_t *total_bytes)
{
    if (!c || !free_bytes || !total_bytes) return NAF_GPU_EARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemGetInfo(free_bytes, total_bytes));
    return 0;
}
extern "C" int naf_gpu_release_scratch(naf_gpu_ctx *c)
{
    if (!c) return NAF_GPU_EARG;
    hipStreamSynchronize(c->stream);
    ennaf_shard_state_free(c);                                    // a shard between its begin and its finish lives in the arena
    for (auto &ch : c->chunks) hipFree(ch.base);
    c->chunks.clear();
    return 0;
}
static int reserve_one(naf_gpu_ctx *c, size_t bytes);
extern "C" int naf_gpu_reserve(naf_gpu_ctx *c, size_t bytes)
{
    if (!c) return NAF_GPU_EARG;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ctx_sides_ready(c); if (rc) return rc;
    rc = reserve_one(c, bytes);
    // the side contexts (side sections of an archive, a FASTQ's quality stream, the chains behind an encode's split) take an eighth
    // each, at most 4 GiB: their arenas grow past that like any arena, in the first call that needs more
    size_t each = bytes / 8; if (each > ((size_t)4 << 30)) each = (size_t)4 << 30;
    for (naf_gpu_ctx *sc : { c->side, c->side2, c->side3, c->side4 }) if (sc && !rc && each) { rc = reserve_one(sc, each); if (rc) memcpy(c->err, sc->err, sizeof c->err); }
    return rc;
}
static int reserve_one(naf_gpu_ctx *c, size_t bytes)
{
    size_t total = 0;
    for (auto &ch : c->chunks) total += ch.cap;
    if (total >= bytes && c->chunks.size() <= 1) return 0;
    hipStreamSynchronize(c->stream);
    for (auto &ch : c->chunks) hipFree(ch.base);
    c->chunks.clear();
    u8 *p = nullptr;
    if (bytes < total) bytes = total;
    HIP_TRY(c, hipMalloc((void **)&p, bytes));
    c->chunks.push_back({ p, bytes, 0 });
    return 0;
}

// a few bytes straight into the pinned staging buffer (mapped into the device's address space), by a kernel of the stream itself
__global__ void k_small_to_host(u8 *h, const u8 *d, u32 n) { if (threadIdx.x < n) h[threadIdx.x] = d[threadIdx.x]; __threadfence_system(); }

int ctx_readback(naf_gpu_ctx *c, void *h_dst, const void *d_src, size_t bytes)
{
    if (bytes && bytes <= 256) {
        hipLaunchKernelGGL(k_small_to_host, dim3(1), dim3(256), 0, c->stream, (u8 *)c->h_stage, (const u8 *)d_src, (u32)bytes);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        memcpy(h_dst, c->h_stage, bytes);
        return 0;
    }
    if (bytes > c->h_stage_cap) {
        HIP_TRY(c, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_stage, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memcpy(h_dst, c->h_stage, bytes);
    return 0;
}

// two small read-backs with one wait for the stream
int ctx_readback2(naf_gpu_ctx *c, void *h1, const void *d1, size_t n1, void *h2, const void *d2, size_t n2)
{
    if (n1 + n2 + 16 > c->h_stage_cap) { int rc = ctx_readback(c, h1, d1, n1); return rc ? rc : ctx_readback(c, h2, d2, n2); }
    u8 *st = (u8 *)c->h_stage; size_t o2 = (n1 + 15) & ~(size_t)15;
    HIP_TRY(c, hipMemcpyAsync(st, d1, n1, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(st + o2, d2, n2, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memcpy(h1, st, n1); memcpy(h2, st + o2, n2);
    return 0;
}

// n read-backs (each up to a few hundred bytes) with one wait for the stream
int ctx_readbackv(naf_gpu_ctx *c, int n, void *const *h, const void *const *d, const size_t *bytes)
{
    size_t off[8], tot = 0;
    if (n < 1 || n > 8) return NAF_GPU_EARG;
    for (int k = 0; k < n; k++) { off[k] = tot; tot += (bytes[k] + 15) & ~(size_t)15; }
    if (tot > c->h_stage_cap) { for (int k = 0; k < n; k++) { int rc = ctx_readback(c, h[k], d[k], bytes[k]); if (rc) return rc; } return 0; }
    for (int k = 0; k < n; k++) HIP_TRY(c, hipMemcpyAsync(c->h_stage + off[k], d[k], bytes[k], hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < n; k++) memcpy(h[k], c->h_stage + off[k], bytes[k]);
    return 0;
}

// ---- memory helpers for hosts that do not link HIP ----------------------------------------------------------
extern "C" int naf_gpu_malloc(naf_gpu_ctx *c, size_t bytes, void **p) { if (!c || !p) return NAF_GPU_EARG; HIP_TRY(c, hipSetDevice(c->device)); HIP_TRY(c, hipMalloc(p, bytes ? bytes : 1)); return 0; }
extern "C" int naf_gpu_free(naf_gpu_ctx *c, void *p) { if (!c) return NAF_GPU_EARG; HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(p)); return 0; }
extern "C" int naf_gpu_host_alloc(naf_gpu_ctx *c, size_t bytes, void **p) { if (!c || !p) return NAF_GPU_EARG; HIP_TRY(c, hipSetDevice(c->device)); HIP_TRY(c, hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault)); return 0; }
extern "C" int naf_gpu_host_free(naf_gpu_ctx *c, void *p) { if (!c) return NAF_GPU_EARG; HIP_TRY(c, hipHostFree(p)); return 0; }
extern "C" int naf_gpu_upload(naf_gpu_ctx *c, void *d, const void *h, size_t n) { if (!c) return NAF_GPU_EARG; if (n) HIP_TRY(c, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, c->stream)); return 0; }
extern "C" int naf_gpu_download(naf_gpu_ctx *c, void *h, const void *d, size_t n)
{
    if (!c) return NAF_GPU_EARG;
    if (n) HIP_TRY(c, hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int naf_gpu_copy(naf_gpu_ctx *c, void *d_dst, const void *d_src, size_t n)
{
    if (!c || (n && (!d_dst || !d_src))) return NAF_GPU_EARG;
    if (n) HIP_TRY(c, hipMemcpyAsync(d_dst, d_src, n, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

// Ranges of one text decoded by several contexts -- one per GPU of the node, driven by one process (the C hosts' NAF_GPUS) -- brought
// together in one buffer on dst's device: every source pushes its range over its xGMI link to dst (hipMemcpyPeerAsync on the source's
// own stream, behind the range decode queued there, so a rank's copy starts when ITS decode is done), and dst's stream waits for all
// of them.  xGMI is point to point: N - 1 pushes into one GPU use N - 1 different links and are bound by the target's ingress, which
// is what a ring or a tree of RCCL send/recv into one root would be bound by as well; RCCL adds nothing between the GPUs of one
// process, so the product's collective is this (processes per GPU -- bench.py, the tests -- gather with torch.distributed instead).
extern "C" int naf_gpu_gather_ranges(naf_gpu_ctx *dst, void *d_dst, naf_gpu_ctx *const *srcs, const void *const *d_src,
                                     const uint64_t *dst_off, const size_t *len, int n)
{
    if (!dst || !d_dst || !srcs || !d_src || !dst_off || !len || n < 0) return NAF_GPU_EARG;
    for (int k = 0; k < n; k++) {
        naf_gpu_ctx *s = srcs[k];
        if (!s || (len[k] && !d_src[k])) return ctx_fail(dst, NAF_GPU_EARG, "gather_ranges: source %d missing", k);
        if (!len[k]) continue;
        u8 *to = (u8 *)d_dst + dst_off[k];
        HIP_TRY(dst, hipSetDevice(s->device));
        if (s->device == dst->device) {
            if ((const void *)to != d_src[k]) HIP_TRY(dst, hipMemcpyAsync(to, d_src[k], len[k], hipMemcpyDeviceToDevice, s->stream));
        } else {
            hipError_t e = hipDeviceEnablePeerAccess(dst->device, 0);             // without it the copy is staged through the host
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            else if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            HIP_TRY(dst, hipMemcpyPeerAsync(to, dst->device, d_src[k], s->device, len[k], s->stream));
        }
        if (s != dst) {
            if (!s->fork_ev) HIP_TRY(dst, hipEventCreateWithFlags(&s->fork_ev, hipEventDisableTiming));
            HIP_TRY(dst, hipEventRecord(s->fork_ev, s->stream));
            HIP_TRY(dst, hipStreamWaitEvent(dst->stream, s->fork_ev, 0));
        }
    }
    HIP_TRY(dst, hipSetDevice(dst->device));
    return 0;
}

extern "C" int naf_gpu_download_async(naf_gpu_ctx *c, void *h, const void *d, size_t n)
{
    if (!c) return NAF_GPU_EARG;
    if (n) HIP_TRY(c, hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, c->stream));
    return 0;
}

// ---- timing -------------------------------------------------------------------------------------------------
extern "C" int naf_gpu_set_timing(naf_gpu_ctx *c, int enable)
{
    if (!c) return NAF_GPU_EARG;
    ctx_sides_ready(c);
    c->timing = enable != 0; c->ktimes.clear(); c->ev_used = 0;
    for (naf_gpu_ctx *sc : { c->side, c->side2, c->side3, c->side4 }) if (sc) { sc->timing = c->timing; sc->ktimes.clear(); sc->ev_used = 0; }
    return 0;
}

static hipEvent_t ev_get(naf_gpu_ctx *c)
{
    if (c->ev_used == c->ev_pool.size()) { hipEvent_t e; hipEventCreate(&e); c->ev_pool.push_back(e); }
    return c->ev_pool[c->ev_used++];
}

void ktime_begin(naf_gpu_ctx *c, const char *name)
{
#ifdef NAF_GPU_DEVELOPMENT
    if (ctx_opt_is(c, "SYNC_DEBUG", '1')) fprintf(stderr, "[launch] %s\n", name);
#endif
    if (!c->timing) return;
    KTime k; k.name = name; k.a = ev_get(c); k.b = ev_get(c);
    hipEventRecord(k.a, c->stream);
    c->ktimes.push_back(k);
}

void ktime_end(naf_gpu_ctx *c)
{
    // development builds (make DEV=1: -DNAF_GPU_DEVELOPMENT) with SYNC_DEBUG=1: every launch is waited for and named on stderr -- the last
    // name printed before a hang or a fault is the kernel that did it.  Compiled out of the release build.
#ifdef NAF_GPU_DEVELOPMENT
    if (ctx_opt_is(c, "SYNC_DEBUG", '1')) { fprintf(stderr, "[launch] ...\n"); hipError_t e = hipStreamSynchronize(c->stream); fprintf(stderr, "[launch] done: %s\n", hipGetErrorString(e)); }
#endif
    if (!c->timing) return;
    hipEventRecord(c->ktimes.back().b, c->stream);
}

extern "C" int naf_gpu_get_timing(naf_gpu_ctx *c, const char **names, float *ms, int *launches, int cap)
{
    if (!c) return NAF_GPU_EARG;
    ctx_sides_ready(c);
    hipStreamSynchronize(c->stream);
    for (naf_gpu_ctx *sc : { c->side, c->side2, c->side3, c->side4 }) if (sc) hipStreamSynchronize(sc->stream);
    std::map<std::string, std::pair<float, int>> agg;
    std::vector<std::string> order;
    int xi = 0;
    for (int k = 0; k < 5; k++) c->stream_ms[k] = 0;
    for (naf_gpu_ctx *x : { c, c->side, c->side2, c->side3, c->side4 }) {
        const int xk = xi++;
        if (!x) continue;
        for (auto &k : x->ktimes) {
            float t = 0; hipEventElapsedTime(&t, k.a, k.b);
            c->stream_ms[xk] += t;
            // launches of the side context run concurrently with the payload decode: listed under their own names
            std::string nm = x == c ? std::string(k.name) : std::string("side:") + k.name;
            auto it = agg.find(nm);
            if (it == agg.end()) { agg[nm] = { t, 1 }; order.push_back(nm); }
            else { it->second.first += t; it->second.second++; }
        }
        x->ktimes.clear(); x->ev_used = 0;
    }
    c->agg_names = order; c->agg_ms.clear(); c->agg_n.clear();
    for (auto &n : order) { c->agg_ms.push_back(agg[n].first); c->agg_n.push_back(agg[n].second); }
    int n = (int)order.size() < cap ? (int)order.size() : cap;
    for (int i = 0; i < n; i++) { names[i] = c->agg_names[i].c_str(); ms[i] = c->agg_ms[i]; launches[i] = c->agg_n[i]; }
    c->ktimes.clear(); c->ev_used = 0;
    return n;
}

// Sum of the kernel times of the last naf_gpu_get_timing per stream of the context: [0] the caller's stream, [1..4] the side chains'.
// A call cannot be shorter than the largest of them; what it takes beyond that is launch gaps, host read-backs and waits between streams.
extern "C" int naf_gpu_get_timing_streams(naf_gpu_ctx *c, float ms[5])
{
    if (!c || !ms) return NAF_GPU_EARG;
    for (int k = 0; k < 5; k++) ms[k] = c->stream_ms[k];
    return 0;
}
