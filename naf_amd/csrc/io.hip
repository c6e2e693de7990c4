// io.hip -- file <-> HBM ("host I/O path").  The reference moves every byte through one thread and 16 KiB
// fread / fwrite calls (ennaf/src/process.c:143-150, unnaf/src/output.c:640-651, files.c); a device-resident codec that finishes
// 10 GB in milliseconds is then only as fast as its file I/O.  Here a transfer is cut into 16 MiB chunks dealt round-robin to a
// few host threads, each with two pinned staging buffers: while one chunk of a lane is on the PCIe link (hipMemcpyAsync), the lane's
// thread is in pread / pwrite for its other chunk, and the lanes run beside each other -- page cache copies, PCIe and (for tmpfs /
// NVMe) the file system all see several requests in flight.  The copies of all lanes go through the CONTEXT's stream, each followed by
// an event of its lane: one stream feeds the link at 49 GB/s of its 55 (tools/io_probe), a stream per lane was 45 ms of
// hipStreamCreate each (rocprofv3 --hip-trace of the CLIs) -- more than the transfer of a GB -- and copies behind the kernels that
// make their source need no wait in between.
#include "ctx.h"
#include <thread>
#include <atomic>
#include <unistd.h>
#include <errno.h>

static const size_t IO_CHUNK = (size_t)16 << 20;       // chunks of the large transfers
static const size_t IO_CHUNK_SMALL = (size_t)4 << 20;  // ... of transfers up to 2 GiB: more lanes on less pinned memory (pinning costs by the byte)
struct IoLane { void *pin[2] = { nullptr, nullptr }; hipEvent_t ev[2] = { nullptr, nullptr }; bool ready = false; size_t cap = 0; };
struct IoPool { int lanes = 0; IoLane lane[16]; std::vector<void *> blocks; };

static IoPool *io_pool(naf_gpu_ctx *c)
{
    if (c->io_pool) return (IoPool *)c->io_pool;
    IoPool *p = new IoPool();
    const char *e = ctx_opt(c, "IO_THREADS");
    int n = e ? atoi(e) : 8; if (n < 1) n = 1; if (n > 16) n = 16;
    p->lanes = n;
    c->io_pool = p;
    return p;
}
// The pinned buffers of the lanes a transfer is going to use, in ONE allocation for those that have none yet: a hipHostMalloc is
// 4.5 ms whatever its size up to tens of MB and the calls of several threads do not run beside each other (rocprofv3 --hip-trace of
// the CLIs: 16 calls, 72 ms in front of the first byte of a 1 GB archive); a transfer of one lane (every write) still pins one lane's.
static bool lanes_ready(IoPool *P, int T, size_t chunk = IO_CHUNK)
{
    int need = 0;
    for (int t = 0; t < T; t++) if (!P->lane[t].ready || P->lane[t].cap < chunk) need++;
    if (!need) return true;
    u8 *base = nullptr;
    if (hipHostMalloc((void **)&base, (size_t)need * 2 * chunk, hipHostMallocDefault) != hipSuccess) return false;
    P->blocks.push_back(base);
    for (int t = 0; t < T; t++) {
        IoLane &L = P->lane[t]; if (L.ready && L.cap >= chunk) continue;       // (a lane's smaller buffers stay allocated until the pool goes: they are part of a block)
        bool ok = true;
        for (int k = 0; k < 2 && ok; k++) { L.pin[k] = base; base += chunk; if (!L.ev[k]) ok = hipEventCreateWithFlags(&L.ev[k], hipEventDisableTiming) == hipSuccess; }
        if (!ok) return false;
        L.ready = true; L.cap = chunk;
    }
    return true;
}

void io_pool_free(naf_gpu_ctx *c)
{
    IoPool *p = (IoPool *)c->io_pool; if (!p) return;
    for (int i = 0; i < 16; i++) for (int k = 0; k < 2; k++) if (p->lane[i].ev[k]) hipEventDestroy(p->lane[i].ev[k]);
    for (void *b : p->blocks) hipHostFree(b);
    delete p; c->io_pool = nullptr;
}

static bool pread_full(int fd, void *buf, size_t n, u64 off)
{
    u8 *p = (u8 *)buf;
    while (n) { ssize_t r = pread(fd, p, n, (off_t)off); if (r < 0 && errno == EINTR) continue; if (r <= 0) return false; p += r; n -= (size_t)r; off += (u64)r; }
    return true;
}
static bool pwrite_full(int fd, const void *buf, size_t n, u64 off)
{
    const u8 *p = (const u8 *)buf;
    while (n) { ssize_t r = pwrite(fd, p, n, (off_t)off); if (r < 0 && errno == EINTR) continue; if (r <= 0) return false; p += r; n -= (size_t)r; off += (u64)r; }
    return true;
}

extern "C" int naf_gpu_read_file(naf_gpu_ctx *c, int fd, uint64_t file_off, size_t len, void *d_dst)
{
    if (!c || (!d_dst && len)) return NAF_GPU_EARG;
    if (!len) return 0;
    IoPool *P = io_pool(c);
    // up to 2 GiB (an archive): a lane per 64 MiB with two 4 MiB buffers each -- eight readers on 64 MiB of pinned memory; larger
    // (a text): a lane per 256 MiB with two 16 MiB buffers.  Pinning a lane's buffers costs what reading a few dozen MB through them
    // takes, and a single pread thread copies out of the page cache at 5 - 10 GB/s, a fifth of what the link takes.
    const bool small = len <= ((size_t)2 << 30) && !ctx_opt_is(c, "IO_SMALL", '0');
    const size_t CH = small ? IO_CHUNK_SMALL : IO_CHUNK;
    const u64 nchunks = (len + CH - 1) / CH;
    u64 want = (len >> (small ? 26 : 28)) + 1; if (want > (u64)P->lanes) want = (u64)P->lanes;
    const int T = (int)(nchunks < want ? nchunks : want);
    std::atomic<int> bad(0);
    HIP_TRY(c, hipSetDevice(c->device));
    if (!lanes_ready(P, T, CH)) return ctx_fail(c, NAF_GPU_ENOMEM, "can't allocate pinned staging buffers");
    auto work = [&](int t) {
        hipSetDevice(c->device);
        IoLane &L = P->lane[t]; bool used[2] = { false, false };
        for (u64 i = (u64)t, k = 0; i < nchunks && !bad.load(); i += (u64)T, k++) {
            const int slot = (int)(k & 1); const u64 off = i * CH; const size_t n = len - off < CH ? (size_t)(len - off) : CH;
            if (used[slot] && hipEventSynchronize(L.ev[slot]) != hipSuccess) { bad = 2; break; }          // the upload that last read this buffer
            if (!pread_full(fd, L.pin[slot], n, file_off + off)) { bad = 1; break; }
            if (hipMemcpyAsync((u8 *)d_dst + off, L.pin[slot], n, hipMemcpyHostToDevice, c->stream) != hipSuccess || hipEventRecord(L.ev[slot], c->stream) != hipSuccess) { bad = 2; break; }
            used[slot] = true;
        }
        for (int k = 0; k < 2; k++) if (used[k] && hipEventSynchronize(L.ev[k]) != hipSuccess) bad = 2;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    if (bad == 3) return ctx_fail(c, NAF_GPU_ENOMEM, "can't allocate pinned staging buffers");
    if (bad == 1) return ctx_fail(c, NAF_GPU_EARG, "can't read the input (short read or not a seekable file)");
    if (bad) return ctx_fail(c, NAF_GPU_EHIP, "host -> device copy failed: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

extern "C" int naf_gpu_write_file(naf_gpu_ctx *c, int fd, uint64_t file_off, const void *d_src, size_t len)
{
    if (!c || (!d_src && len)) return NAF_GPU_EARG;
    if (!len) return 0;
    IoPool *P = io_pool(c);
    const u64 nchunks = (len + IO_CHUNK - 1) / IO_CHUNK;
    // ONE writer: pwrite() into a new tmpfs file runs 7.5 GB/s from one thread and
    // 4.7 / 3.8 GB/s from two / eight (the file's pages are added under one lock: tools/io_probe, profiles/r03_io_probe.txt), while the
    // link delivers 50 GB/s to a single stream -- the download of chunk i + 1 runs beside the write of chunk i either way
    const int T = 1;
    std::atomic<int> bad(0);
    HIP_TRY(c, hipSetDevice(c->device));
    if (!lanes_ready(P, T)) return ctx_fail(c, NAF_GPU_ENOMEM, "can't allocate pinned staging buffers");
    auto work = [&](int t) {
        hipSetDevice(c->device);
        IoLane &L = P->lane[t];
        auto issue = [&](u64 i, int slot) -> bool {
            const u64 off = i * IO_CHUNK; const size_t n = len - off < IO_CHUNK ? (size_t)(len - off) : IO_CHUNK;
            return hipMemcpyAsync(L.pin[slot], (const u8 *)d_src + off, n, hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipEventRecord(L.ev[slot], c->stream) == hipSuccess;
        };
        if ((u64)t < nchunks && !issue((u64)t, 0)) { bad = 2; return; }
        for (u64 i = (u64)t, k = 0; i < nchunks && !bad.load(); i += (u64)T, k++) {
            const int slot = (int)(k & 1); const u64 off = i * IO_CHUNK; const size_t n = len - off < IO_CHUNK ? (size_t)(len - off) : IO_CHUNK;
            if (i + (u64)T < nchunks && !issue(i + (u64)T, slot ^ 1)) { bad = 2; break; }                   // next chunk on the link while this one is written
            if (hipEventSynchronize(L.ev[slot]) != hipSuccess) { bad = 2; break; }
            if (!pwrite_full(fd, L.pin[slot], n, file_off + off)) { bad = 1; break; }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    if (bad == 3) return ctx_fail(c, NAF_GPU_ENOMEM, "can't allocate pinned staging buffers");
    if (bad == 1) return ctx_fail(c, NAF_GPU_EARG, "can't write to file - disk full?");
    if (bad) return ctx_fail(c, NAF_GPU_EHIP, "device -> host copy failed: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

// Device memory to a descriptor that is written in order (a pipe, /dev/null, a file opened for appending): the next chunks are on the
// link -- two of them queued behind each other on the context's stream -- while the caller's thread is in write() for the one that has
// arrived.  (The hosts' first form waited for the stream before it queued the next chunk: one copy in flight, 25 GB/s of the link's 50.)
extern "C" int naf_gpu_write_fd(naf_gpu_ctx *c, int fd, const void *d_src, size_t len)
{
    if (!c || (!d_src && len)) return NAF_GPU_EARG;
    if (!len) return 0;
    IoPool *P = io_pool(c);
    HIP_TRY(c, hipSetDevice(c->device));
    if (!lanes_ready(P, 2)) return ctx_fail(c, NAF_GPU_ENOMEM, "can't allocate pinned staging buffers");
    void *pin[4] = { P->lane[0].pin[0], P->lane[0].pin[1], P->lane[1].pin[0], P->lane[1].pin[1] };
    hipEvent_t ev[4] = { P->lane[0].ev[0], P->lane[0].ev[1], P->lane[1].ev[0], P->lane[1].ev[1] };
    const size_t CH = IO_CHUNK / 2;                                        // 8 MiB: four of them make the ring
    const u64 nchunks = (len + CH - 1) / CH;
    auto issue = [&](u64 i) -> bool {
        const u64 off = i * CH; const size_t n = len - off < CH ? (size_t)(len - off) : CH;
        return hipMemcpyAsync(pin[i & 3], (const u8 *)d_src + off, n, hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipEventRecord(ev[i & 3], c->stream) == hipSuccess;
    };
    for (u64 i = 0; i < 3 && i < nchunks; i++) if (!issue(i)) return ctx_fail(c, NAF_GPU_EHIP, "device -> host copy failed: %s", hipGetErrorString(hipGetLastError()));
    for (u64 i = 0; i < nchunks; i++) {
        const u64 off = i * CH; const size_t n = len - off < CH ? (size_t)(len - off) : CH;
        if (hipEventSynchronize(ev[i & 3]) != hipSuccess) return ctx_fail(c, NAF_GPU_EHIP, "device -> host copy failed: %s", hipGetErrorString(hipGetLastError()));
        const u8 *p = (const u8 *)pin[i & 3]; size_t left = n;
        while (left) { ssize_t w = write(fd, p, left); if (w < 0 && errno == EINTR) continue; if (w <= 0) return ctx_fail(c, NAF_GPU_EARG, "can't write to file - disk full?"); p += w; left -= (size_t)w; }
        if (i + 3 < nchunks && !issue(i + 3)) return ctx_fail(c, NAF_GPU_EHIP, "device -> host copy failed: %s", hipGetErrorString(hipGetLastError()));   // (into the buffer that was written last time round)
    }
    return 0;
}
