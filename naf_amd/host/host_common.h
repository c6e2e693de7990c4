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
static void warn(const char *format, ...) { fprintf(stderr, "%s warning: ", prog_name); va_list a; va_start(a, format); vfprintf(stderr, format, a); va_end(a); }
__attribute__((noreturn)) static void die(const char *format, ...)
{
    fprintf(stderr, "%s error: ", prog_name);
    va_list a; va_start(a, format); vfprintf(stderr, format, a); va_end(a);
    exit(1);
}

static naf_gpu_ctx *gpu = NULL;
static void gpu_open(void)
{
    if (gpu) return;
    const char *dev = getenv("NAF_GPU_DEVICE");
    int rc = naf_gpu_init(dev ? atoi(dev) : 0, &gpu);
    if (rc) die("can't initialize the GPU path: %s\n", naf_gpu_strerror(rc));
}
#define GPU_TRY(call) do { int rc_ = (call); if (rc_) { const char *m_ = naf_gpu_last_error(gpu); size_t l_ = strlen(m_); \
    die("%s%s", m_, (l_ && m_[l_ - 1] == '\n') ? "" : "\n"); } } while (0)

/* Whole input into pinned host memory (file or stdin). */
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
