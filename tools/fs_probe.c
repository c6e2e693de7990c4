/* fs_probe.c -- how fast a NEW file of the output's size can be filled on this box's tmpfs: fallocate, pwrite behind it
 * from 1..8 threads, memcpy into a shared mapping of the preallocated file from 1..8 threads.  gcc -O2 -pthread -o tools/fs_probe tools/fs_probe.c */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
static size_t N; static int fd; static char *src, *map; static int T;
static void *w_pwrite(void *a) { long t = (long)a; size_t per = (N / T + 4095) & ~(size_t)4095, lo = t * per, hi = lo + per < N ? lo + per : N; const size_t CH = 16 << 20;
    for (size_t o = lo; o < hi; o += CH) { size_t n = hi - o < CH ? hi - o : CH; if (pwrite(fd, src, n, o) != (ssize_t)n) { perror("pwrite"); exit(1); } } return NULL; }
static void *w_memcpy(void *a) { long t = (long)a; size_t per = (N / T + 4095) & ~(size_t)4095, lo = t * per, hi = lo + per < N ? lo + per : N; const size_t CH = 16 << 20;
    for (size_t o = lo; o < hi; o += CH) { size_t n = hi - o < CH ? hi - o : CH; memcpy(map + o, src, n); } return NULL; }
static double run(void *(*f)(void *)) { pthread_t th[16]; double t0 = now(); for (long t = 1; t < T; t++) pthread_create(&th[t], NULL, f, (void *)t); f((void *)0); for (int t = 1; t < T; t++) pthread_join(th[t], NULL); return now() - t0; }
int main(int argc, char **argv)
{
    N = (size_t)(atof(argc > 1 ? argv[1] : "4") * 1e9) & ~(size_t)4095;
    const char *dir = argc > 2 ? argv[2] : "/dev/shm"; char path[512]; snprintf(path, sizeof path, "%s/fs_probe.bin", dir);
    src = malloc(16 << 20); memset(src, 0x41, 16 << 20);
    for (int mode = 0; mode < 2; mode++) {
        unlink(path); fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
        double t0 = now(); int rc = mode ? fallocate(fd, FALLOC_FL_KEEP_SIZE, 0, N) : posix_fallocate(fd, 0, N); double dt = now() - t0;
        printf("%s of a new %.1f GB file: %.3f s (rc %d)\n", mode ? "fallocate(KEEP_SIZE)" : "posix_fallocate", N / 1e9, dt, rc);
        for (T = 1; T <= 8; T *= 2) { double d = run(w_pwrite); printf("  pwrite over the preallocated file, %d threads: %.3f s  %.1f GB/s\n", T, d, N / d / 1e9); }
        close(fd);
    }
    for (T = 1; T <= 8; T *= 2) {
        unlink(path); fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
        double t0 = now(); posix_fallocate(fd, 0, N); double ta = now() - t0;
        t0 = now(); map = mmap(NULL, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); double d = run(w_memcpy); double tm = now() - t0;
        t0 = now(); munmap(map, N); double tu = now() - t0;
        printf("posix_fallocate %.3f s + memcpy into a shared mapping, %d threads: %.3f s  %.1f GB/s (copy alone %.3f) + munmap %.3f\n", ta, T, tm, N / tm / 1e9, d, tu);
        close(fd);
    }
    { unlink(path); fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644); T = 1; double d = run(w_pwrite); printf("pwrite of a new file, 1 thread: %.3f s  %.1f GB/s\n", d, N / d / 1e9);
      double t0 = now(); close(fd); fd = open(path, O_WRONLY | O_TRUNC); printf("truncating it again: %.3f s\n", now() - t0); close(fd); }
    unlink(path);
    return 0;
}
