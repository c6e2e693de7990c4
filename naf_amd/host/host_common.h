/* host_common.h -- shared plumbing of the C hosts (ennaf / unnaf command-line programs).
 * The hosts keep the reference's command line, stdout payload, stderr texts and exit codes
 * (ennaf/src/ennaf.c:334-430, unnaf/src/unnaf.c:251-353, and both utils.c die/err/warn/msg) and hand the
 * hot path to libnaf_gpu.so through include/naf_gpu.h.  There is no CPU fallback: without a usable
 * gfx950 device the programs stop with an error. */
#ifndef NAF_HOST_COMMON_H
#define NAF_HOST_COMMON_H
#include <stdarg.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "../../include/naf_gpu.h"

#define VERSION "1.3.0"
#define DATE "2021-05-17"
#define COPYRIGHT_YEARS "2018-2021"

static const char *prog_name = "naf";

static void msg(const char *format, ...) { va_list a; va_start(a, format); vfprintf(stderr, format, a); va_end(a); }
static void err(const char *format, ...) { fprintf(stderr, "%s error: ", prog_name); va_list a; va_start(a, format); vfprintf(stderr, format, a); va_end(a); }
__attribute__((unused)) static void warn(const char *format, ...) { fprintf(stderr, "%s warning: ", prog_name); va_list a; va_start(a, format); vfprintf(stderr, format, a); va_end(a); }
__attribute__((noreturn)) static void die(const char *format, ...)
{
    fprintf(stderr, "%s error: ", prog_name);
    va_list a; va_start(a, format); vfprintf(stderr, format, a); va_end(a);
    exit(1);
}

/* NAF_GPU_CLI_TIMING=1: wall-clock of the host phases on stderr (development aid; off by default so stderr matches the reference) */
#include <time.h>
static void phase(const char *what)
{
    static int on = -1; static double t_last = 0;
    if (on < 0) { const char *e = getenv("NAF_GPU_CLI_TIMING"); on = e && e[0] == '1'; }
    if (!on) return;
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    double t = ts.tv_sec + ts.tv_nsec * 1e-9;
    if (t_last != 0) fprintf(stderr, "[timing] %-28s %8.1f ms\n", what, (t - t_last) * 1e3);
    t_last = t;
}

/* A decimal command-line argument, judged the way the reference judges one: strtoll must consume the whole argument, and the
 * value printed back with %lld must reproduce the argument byte for byte (unnaf/src/unnaf.c:224-239, ennaf/src/ennaf.c:230-273).
 * Returns 0 when strtoll would stop early or convert nothing (trailing garbage), 1 when it converts but the argument is not
 * the canonical spelling of the value (white space, '+', leading zeros, "-0", out of range), 2 when it is.  *v as strtoll
 * would return it (saturated), for results 1 and 2. */
__attribute__((unused)) static int decimal_arg(const char *s, long long *v)
{
    const char *p = s;
    while (*p == ' ' || (*p >= '\t' && *p <= '\r')) p++;
    bool neg = false;
    if (*p == '+' || *p == '-') { neg = *p == '-'; p++; }
    const char *digits = p;
    const unsigned long long lim = neg ? 9223372036854775808ull : 9223372036854775807ull;
    unsigned long long mag = 0; bool saturated = false;
    for (; *p >= '0' && *p <= '9'; p++) {
        unsigned dg = (unsigned)(*p - '0');
        if (saturated || mag > (lim - dg) / 10) { saturated = true; mag = lim; } else mag = mag * 10 + dg;
    }
    *v = 0;
    if (p == digits) return s[0] ? 0 : 1;              /* nothing converted: only the empty argument gets past the end test */
    if (*p) return 0;
    *v = neg ? (long long)(0ull - mag) : (long long)mag;
    bool canonical = !saturated && digits == s + (neg ? 1 : 0) && !(digits[0] == '0' && digits[1]) && !(neg && mag == 0);
    return canonical ? 2 : 1;
}

static naf_gpu_ctx *gpu = NULL;
static void gpu_open(void)
{
    if (gpu) return;
    phase("before GPU init");
    const char *dev = getenv("NAF_GPU_DEVICE");
    int rc = naf_gpu_init(dev ? atoi(dev) : 0, &gpu);
    if (rc) die("can't initialize the GPU path: %s\n", naf_gpu_strerror(rc));
    phase("GPU init");
}
#define GPU_TRY(call) do { int rc_ = (call); if (rc_) { const char *m_ = naf_gpu_last_error(gpu); size_t l_ = strlen(m_); \
    die("%s%s", m_, (l_ && m_[l_ - 1] == '\n') ? "" : "\n"); } } while (0)

/* ---- file <-> device through two pinned chunks: the PCIe copy of one chunk overlaps the file I/O of the next ---- */
#define IO_CHUNK ((size_t)64 << 20)
static void *io_pin[2] = { NULL, NULL };
static void io_open(void)
{
    gpu_open();
    if (!io_pin[0]) { GPU_TRY(naf_gpu_host_alloc(gpu, IO_CHUNK, &io_pin[0])); GPU_TRY(naf_gpu_host_alloc(gpu, IO_CHUNK, &io_pin[1])); }
}
static void write_from_device(FILE *f, const void *d, size_t n)
{
    if (!n) return;
    io_open();
    size_t off = 0; int cur = 0;
    GPU_TRY(naf_gpu_download_async(gpu, io_pin[0], d, n < IO_CHUNK ? n : IO_CHUNK));
    while (off < n) {
        size_t len = n - off < IO_CHUNK ? n - off : IO_CHUNK, nxt = off + len, nlen = n - nxt < IO_CHUNK ? n - nxt : IO_CHUNK;
        GPU_TRY(naf_gpu_synchronize(gpu));                                   /* chunk `cur` has arrived */
        if (nlen) GPU_TRY(naf_gpu_download_async(gpu, io_pin[cur ^ 1], (const char *)d + nxt, nlen));
        if (fwrite(io_pin[cur], 1, len, f) != len) die("can't write to file - disk full?\n");
        off = nxt; cur ^= 1;
    }
}
/* Regular file of known size straight into device memory; returns NULL when the size is not known up front (pipes). */
__attribute__((unused)) static void *read_to_device(FILE *f, size_t *len)
{
    if (f == stdin || fseek(f, 0, SEEK_END) != 0) return NULL;
    long sz = ftell(f); rewind(f);
    if (sz <= 0) return NULL;
    io_open();
    void *d; GPU_TRY(naf_gpu_malloc(gpu, (size_t)sz + 64, &d));
    size_t n = 0; int cur = 0;
    for (;;) {
        size_t want = (size_t)sz - n < IO_CHUNK ? (size_t)sz - n : IO_CHUNK;
        if (!want) break;
        GPU_TRY(naf_gpu_synchronize(gpu));                                   /* the upload that last used this chunk is done */
        size_t r = fread(io_pin[cur], 1, want, f);
        if (!r) break;
        GPU_TRY(naf_gpu_upload(gpu, (char *)d + n, io_pin[cur], r));
        n += r; cur ^= 1;
    }
    GPU_TRY(naf_gpu_synchronize(gpu));
    *len = n;
    return d;
}

/* Whole input into host memory (file or stdin). */
static unsigned char *read_all(FILE *f, size_t *len)
{
    size_t cap = 1 << 20, n = 0;
    unsigned char *buf = NULL;
    if (f != stdin && fseek(f, 0, SEEK_END) == 0) { long sz = ftell(f); if (sz > 0) cap = (size_t)sz + 1; rewind(f); }
    buf = (unsigned char *)malloc(cap);
    if (!buf) die("can't allocate %zu bytes\n", cap);
    for (;;) {
        if (n == cap) { cap *= 2; buf = (unsigned char *)realloc(buf, cap); if (!buf) die("can't allocate %zu bytes\n", cap); }
        size_t r = fread(buf + n, 1, cap - n, f);
        n += r;
        if (r == 0) break;
    }
    *len = n;
    return buf;
}
#endif
