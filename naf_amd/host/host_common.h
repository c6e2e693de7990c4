/* host_common.h -- shared plumbing of the C hosts (ennaf / unnaf command-line programs).
 * The hosts keep the reference's command line, stdout payload, stderr texts and exit codes
 * (ennaf/src/ennaf.c:334-430, unnaf/src/unnaf.c:251-353, and both utils.c die/err/warn/msg) and hand the
 * hot path to libnaf_gpu.so through include/naf_gpu.h.  There is no CPU fallback: without a usable
 * gfx950 device the programs stop with an error. */
#ifndef NAF_HOST_COMMON_H
#define NAF_HOST_COMMON_H
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <stdarg.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <signal.h>
#include <fcntl.h>
#include <sys/stat.h>
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
/* NAF_GPU_TRACE=1 (development): which paths the library took, on stderr when the program ends */
static void trace_out(void) { if (gpu) { const char *tr = naf_gpu_get_trace(gpu); if (tr && *tr) { fputs(tr, stderr); fflush(stderr); } naf_gpu_clear_trace(gpu); } }
static void devices_parse(void);
static int first_device(void);
/* The device is opened by a thread of its own while the main thread parses, opens and maps its files (the HIP runtime's start and
 * the first stream are 0.1 s and more of a run that moves 4 GB in a second): gpu_open_early() as soon as it is known that the GPU
 * will be needed, gpu_open() where it is needed first. */
#include <pthread.h>
static pthread_t gpu_init_thread; static bool gpu_init_started = false; static int gpu_init_rc = 0;
static void *gpu_init_main(void *arg) { (void)arg; gpu_init_rc = naf_gpu_init(first_device(), &gpu); return NULL; }
__attribute__((unused)) static void gpu_open_early(void)
{
    if (gpu || gpu_init_started) return;
    devices_parse();
    if (pthread_create(&gpu_init_thread, NULL, gpu_init_main, NULL) == 0) gpu_init_started = true;
}
static void gpu_open(void)
{
    if (gpu_init_started) {
        phase("before GPU init");
        pthread_join(gpu_init_thread, NULL); gpu_init_started = false;
        if (gpu_init_rc) die("can't initialize the GPU path: %s\n", naf_gpu_strerror(gpu_init_rc));
        phase("GPU init (the wait for it)");
        return;
    }
    if (gpu) return;
    phase("before GPU init");
    int rc = naf_gpu_init(first_device(), &gpu);
    if (rc) die("can't initialize the GPU path: %s\n", naf_gpu_strerror(rc));
    phase("GPU init");
}
/* NAF_GPU_DETACH=1 (opt-in; the default is ONE process, as the reference is -- unnaf.c:355, ennaf.c:433): the process the caller waits
 * for ends when the OUTPUT is complete, not when the driver has taken the device state apart.  The work is done by a forked worker, the
 * foreground process waits for its verdict on a pipe and leaves at once with the worker's exit status, while the worker's teardown
 * (4 GB and more of device memory, pinned buffers, queues: 0.15 - 0.2 s of a run that takes one second -- DESIGN.md section 5) goes on
 * behind it -- the next process on the same device finds less free memory for that long, which is why it is not the default.  Until the
 * verdict is sent the two processes are one job: SIGINT / SIGTERM / SIGHUP / SIGQUIT that reach the foreground process are forwarded,
 * and the worker asks to be sent SIGTERM should the foreground process die (PR_SET_PDEATHSIG), so a `kill` or a `timeout` of the pid the
 * caller knows stops the work and no orphan keeps writing.  Called before anything touches the device. */
#include <sys/types.h>
#include <sys/wait.h>
#include <sys/prctl.h>
#include <errno.h>
static int detach_fd = -1;
static volatile pid_t detach_worker = 0;
static void detach_forward(int sig) { if (detach_worker > 0) kill(detach_worker, sig); }
/* the worker's verdict; its standard input and output are let go with it (a reader behind a pipe would otherwise wait for the teardown
 * too); stderr stays when it is a terminal or a file, so that an error of the teardown is still seen */
static void detach_report(int status)
{
    if (detach_fd < 0) return;
    fflush(NULL);
    prctl(PR_SET_PDEATHSIG, 0);                                          /* the job is done: what is left is this process's own end */
    struct stat st_; bool keep_err = fstat(2, &st_) == 0 && (S_ISCHR(st_.st_mode) || S_ISREG(st_.st_mode));
    int dn = open("/dev/null", O_RDWR); if (dn >= 0) { dup2(dn, 0); dup2(dn, 1); if (!keep_err) dup2(dn, 2); if (dn > 2) close(dn); }
    unsigned char b = (unsigned char)status; if (write(detach_fd, &b, 1) != 1) {}
    close(detach_fd); detach_fd = -1;
}
__attribute__((unused)) static void detach_teardown(void)
{
    const char *e = getenv("NAF_GPU_DETACH"); if (!e || e[0] != '1') return;
    int pf[2]; if (pipe(pf) != 0) return;
    fflush(NULL);
    pid_t parent = getpid();
    pid_t pid = fork();
    if (pid < 0) { close(pf[0]); close(pf[1]); return; }
    if (pid == 0) {                                                      /* the worker: reports when it is done (detach_done) or exits (the hosts' exit handler) */
        close(pf[0]); detach_fd = pf[1];
        prctl(PR_SET_PDEATHSIG, SIGTERM);
        if (getppid() != parent) _exit(1);                               /* the foreground process went away between fork and prctl */
        return;
    }
    close(pf[1]);
    detach_worker = pid;
    struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = detach_forward; sigemptyset(&sa.sa_mask);
    sigaction(SIGINT, &sa, NULL); sigaction(SIGTERM, &sa, NULL); sigaction(SIGHUP, &sa, NULL); sigaction(SIGQUIT, &sa, NULL);
    unsigned char b = 0; ssize_t r;
    do r = read(pf[0], &b, 1); while (r < 0 && errno == EINTR);
    if (r == 1) _exit(b);
    int st = 0; pid_t w;
    do w = waitpid(pid, &st, 0); while (w < 0 && errno == EINTR);        /* it ended without a word: its status is this process's */
    if (w == pid) { if (WIFEXITED(st)) _exit(WEXITSTATUS(st)); if (WIFSIGNALED(st)) { signal(WTERMSIG(st), SIG_DFL); raise(WTERMSIG(st)); } }
    _exit(1);
}
__attribute__((unused)) static void detach_done(int status) { detach_report(status); }
#define GPU_TRY(call) do { int rc_ = (call); if (rc_) { const char *m_ = naf_gpu_last_error(gpu); size_t l_ = strlen(m_); \
    die("%s%s", m_, (l_ && m_[l_ - 1] == '\n') ? "" : "\n"); } } while (0)

/* ---- devices ----------------------------------------------------------------------------------------------------------------
 * NAF_GPUS=0,1,2,...  one context per entry (an entry may repeat a device: several contexts on one GPU); without it the single
 * device NAF_GPU_DEVICE (default 0).  The first entry is also the context `gpu` of the single-device paths. */
#include <fcntl.h>
#include <sys/stat.h>
#define MAX_DEVS 64
static int dev_ids[MAX_DEVS], n_devs = 0;
static void devices_parse(void)
{
    if (n_devs) return;
    const char *e = getenv("NAF_GPUS");
    if (e && *e) {
        const char *p = e;
        while (*p && n_devs < MAX_DEVS) {
            char *end; long v = strtol(p, &end, 10);
            if (end == p || v < 0) die("can't parse NAF_GPUS=\"%s\"\n", e);
            dev_ids[n_devs++] = (int)v;
            p = end; while (*p == ',' || *p == ' ') p++;
        }
    }
    if (!n_devs) { const char *d = getenv("NAF_GPU_DEVICE"); dev_ids[n_devs++] = d ? atoi(d) : 0; }
}
static int first_device(void) { devices_parse(); return dev_ids[0]; }
#define CTX_TRY(ctx, call) do { int rc_ = (call); if (rc_) { const char *m_ = naf_gpu_last_error(ctx); size_t l_ = strlen(m_); \
    die("%s%s", m_, (l_ && m_[l_ - 1] == '\n') ? "" : "\n"); } } while (0)
static naf_gpu_ctx *ctx_open(int device)
{
    naf_gpu_ctx *c = NULL;
    int rc = naf_gpu_init(device, &c);
    if (rc) die("can't initialize the GPU path: %s\n", naf_gpu_strerror(rc));
    return c;
}
static bool fd_is_regular(int fd) { struct stat st; return fstat(fd, &st) == 0 && S_ISREG(st.st_mode); }
/* Position of a descriptor that takes pwrite() at explicit offsets from several threads: a regular file that was not opened for
 * appending -- on Linux pwrite() to an O_APPEND descriptor ignores its offset and appends, so the chunks of `unnaf x.naf >> all.fa`
 * would land in arrival order.  -1: write sequentially (pipes, `>>`), the way the reference's fwrite does. */
static off_t fd_pwrite_pos(int fd)
{
    if (!fd_is_regular(fd)) return (off_t)-1;
    int fl = fcntl(fd, F_GETFL);
    if (fl < 0 || (fl & O_APPEND)) return (off_t)-1;
    return lseek(fd, 0, SEEK_CUR);
}

/* ---- file <-> device --------------------------------------------------------------------------------------------------------
 * Regular files go through naf_gpu_read_file / naf_gpu_write_file (several host threads, pinned staging, pread / pwrite at
 * offsets); pipes through the sequential two-chunk ring below. */
#define IO_CHUNK ((size_t)16 << 20)
static void *io_pin[2] = { NULL, NULL };
__attribute__((unused)) static void io_open(void)
{
    gpu_open();
    if (!io_pin[0]) { GPU_TRY(naf_gpu_host_alloc(gpu, IO_CHUNK, &io_pin[0])); GPU_TRY(naf_gpu_host_alloc(gpu, IO_CHUNK, &io_pin[1])); }
}
static void write_from_device(FILE *f, const void *d, size_t n)
{
    if (!n) return;
    fflush(f);
    off_t at = fd_pwrite_pos(fileno(f));
    if (at >= 0) {
        GPU_TRY(naf_gpu_write_file(gpu, fileno(f), (uint64_t)at, d, n));
        if (lseek(fileno(f), at + (off_t)n, SEEK_SET) < 0) die("can't write to file - disk full?\n");
        return;
    }
    if (fflush(f) != 0) die("can't write to file - disk full?\n");
    GPU_TRY(naf_gpu_write_fd(gpu, fileno(f), d, n));
}
/* Regular file of known size straight into device memory; returns NULL when the size is not known up front (pipes). */
__attribute__((unused)) static void *read_to_device(FILE *f, size_t *len)
{
    if (f == stdin && !fd_is_regular(fileno(f))) return NULL;
    if (!fd_is_regular(fileno(f))) return NULL;
    struct stat st; if (fstat(fileno(f), &st) != 0 || st.st_size <= 0) return NULL;
    off_t at = lseek(fileno(f), 0, SEEK_CUR); if (at < 0) at = 0;
    size_t n = (size_t)st.st_size > (size_t)at ? (size_t)st.st_size - (size_t)at : 0;
    if (!n) return NULL;
    gpu_open();
    void *d; GPU_TRY(naf_gpu_malloc(gpu, n + 64, &d));
    GPU_TRY(naf_gpu_read_file(gpu, fileno(f), (uint64_t)at, n, d));
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
