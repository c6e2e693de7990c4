// io_probe.hip -- where the time of file <-> HBM goes on this box: HIP start-up, the PCIe link, tmpfs page
// allocation by pwrite and by page faults, pinning of file mappings.  Development probe, not product code:
//   hipcc --offload-arch=gfx950 -O2 -o tools/io_probe tools/io_probe.hip && tools/io_probe [GB] [dir]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <thread>
#include <vector>
static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <typename F> static void par(int T, F f) { std::vector<std::thread> th; for (int t = 1; t < T; t++) th.emplace_back(f, t); f(0); for (auto &x : th) x.join(); }
int main(int argc, char **argv)
{
    const size_t N = (size_t)(atof(argc > 1 ? argv[1] : "4") * 1e9) & ~(size_t)((16 << 20) - 1);
    const char *dir = argc > 2 ? argv[2] : "/dev/shm";
    char path[512]; snprintf(path, sizeof path, "%s/io_probe.bin", dir);
    double t0 = now();
    CK(hipInit(0)); CK(hipSetDevice(0)); void *d; CK(hipMalloc(&d, N)); CK(hipMemset(d, 0x41, N)); CK(hipDeviceSynchronize());
    printf("HIP init + %.1f GB hipMalloc + memset: %.3f s\n", N / 1e9, now() - t0);
    const size_t CH = 16 << 20; const size_t nch = N / CH;
    // pinned staging, D2H only
    for (int T : { 1, 2, 4, 8 }) {
        std::vector<void *> pin(2 * T); std::vector<hipStream_t> st(T);
        for (int i = 0; i < 2 * T; i++) CK(hipHostMalloc(&pin[i], CH, hipHostMallocDefault));
        for (int i = 0; i < T; i++) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
        t0 = now();
        par(T, [&](int t) { hipSetDevice(0); for (size_t i = t, k = 0; i < nch; i += T, k++) { hipMemcpyAsync(pin[2 * t + (k & 1)], (char *)d + i * CH, CH, hipMemcpyDeviceToHost, st[t]); if (k & 1) hipStreamSynchronize(st[t]); } hipStreamSynchronize(st[t]); });
        double dt = now() - t0;
        printf("D2H into pinned staging, %d streams: %.3f s  %.1f GB/s\n", T, dt, N / dt / 1e9);
        // pwrite from the pinned buffers into a NEW file
        unlink(path); int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
        t0 = now();
        par(T, [&](int t) { for (size_t i = t; i < nch; i += T) { size_t off = i * CH, left = CH; const char *p = (const char *)pin[2 * t]; while (left) { ssize_t r = pwrite(fd, p, left, off); if (r <= 0) { perror("pwrite"); exit(1); } p += r; left -= r; off += r; } } });
        dt = now() - t0;
        printf("pwrite of a new file from pinned memory, %d threads: %.3f s  %.1f GB/s\n", T, dt, N / dt / 1e9);
        t0 = now();
        par(T, [&](int t) { for (size_t i = t; i < nch; i += T) { size_t off = i * CH, left = CH; const char *p = (const char *)pin[2 * t]; while (left) { ssize_t r = pwrite(fd, p, left, off); p += r; left -= r; off += r; } } });
        dt = now() - t0;
        printf("pwrite over the same file again (pages exist), %d threads: %.3f s  %.1f GB/s\n", T, dt, N / dt / 1e9);
        // pread back
        t0 = now();
        par(T, [&](int t) { for (size_t i = t; i < nch; i += T) { size_t off = i * CH, left = CH; char *p = (char *)pin[2 * t]; while (left) { ssize_t r = pread(fd, p, left, off); p += r; left -= r; off += r; } } });
        dt = now() - t0;
        printf("pread into pinned memory, %d threads: %.3f s  %.1f GB/s\n", T, dt, N / dt / 1e9);
        close(fd);
        for (auto p : pin) hipHostFree(p); for (auto s : st) hipStreamDestroy(s);
    }
    // mapping of a new file, populated by T threads, pinned, DMA target
    for (int T : { 4, 8, 16 }) {
        unlink(path); int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
        t0 = now();
        if (ftruncate(fd, (off_t)N)) { perror("ftruncate"); return 1; }
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { perror("mmap"); return 1; }
        const size_t per = (N / T + 4095) & ~(size_t)4095;
        par(T, [&](int t) { size_t a = (size_t)t * per, b = a + per < N ? a + per : N; if (a < b) madvise(m + a, b - a, MADV_POPULATE_WRITE); });
        double t_pop = now() - t0;
        // touched?  (MADV_POPULATE_WRITE needs Linux 5.14; fall back to touching)
        t0 = now();
        par(T, [&](int t) { size_t a = (size_t)t * per, b = a + per < N ? a + per : N; for (size_t i = a; i < b; i += 4096) m[i] = 0; });
        double t_touch = now() - t0;
        t0 = now();
        par(T, [&](int t) { hipSetDevice(0); size_t a = (size_t)t * per, b = a + per < N ? a + per : N; if (a < b && hipHostRegister(m + a, b - a, hipHostRegisterDefault) != hipSuccess) fprintf(stderr, "register failed\n"); });
        double t_reg = now() - t0;
        std::vector<hipStream_t> st(T); for (int i = 0; i < T; i++) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
        t0 = now();
        par(T, [&](int t) { hipSetDevice(0); size_t a = (size_t)t * per, b = a + per < N ? a + per : N; if (a < b) { hipMemcpyAsync(m + a, (char *)d + a, b - a, hipMemcpyDeviceToHost, st[t]); hipStreamSynchronize(st[t]); } });
        double t_dma = now() - t0;
        t0 = now();
        par(T, [&](int t) { size_t a = (size_t)t * per, b = a + per < N ? a + per : N; if (a < b) hipHostUnregister(m + a); });
        munmap(m, N); close(fd);
        double t_un = now() - t0;
        printf("new file mapped, %d threads: populate %.3f s (+ touch %.3f), hipHostRegister %.3f s, D2H straight into it %.3f s (%.1f GB/s), unregister + unmap %.3f s; sum %.3f s\n",
               T, t_pop, t_touch, t_reg, t_dma, N / t_dma / 1e9, t_un, t_pop + t_touch + t_reg + t_dma + t_un);
        for (auto s : st) hipStreamDestroy(s);
    }
    // pageable destination: what hipMemcpy does with an unregistered mapping
    {
        unlink(path); int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
        if (ftruncate(fd, (off_t)N)) return 1;
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        t0 = now(); CK(hipMemcpy(m, d, N, hipMemcpyDeviceToHost)); double dt = now() - t0;
        printf("hipMemcpy D2H into an unregistered new mapping (one call): %.3f s  %.1f GB/s\n", dt, N / dt / 1e9);
        munmap(m, N); close(fd);
    }
    unlink(path);
    return 0;
}
