/* tests/c/test_gather.c -- the decode path's collective from a C caller, no Python in between:
 *     test_gather archive.naf [contexts]
 * N contexts on device NAF_GPU_DEVICE (default 0; the C hosts' NAF_GPUS=0,0,... shape): every context decodes its byte range of the
 * text with naf_gpu_unnaf_range, naf_gpu_gather_ranges brings the ranges together in context 0's buffer, and the result is compared
 * byte for byte with one whole-text naf_gpu_unnaf.  Exit status 0 and "gather ok ..." on stdout when they agree. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/naf_gpu.h"

#define TRY(ctx, call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s: %s\n", #call, naf_gpu_last_error(ctx)); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: test_gather archive.naf [contexts]\n"); return 2; }
    int n = argc > 2 ? atoi(argv[2]) : 3; if (n < 1 || n > 16) n = 3;
    int dev = getenv("NAF_GPU_DEVICE") ? atoi(getenv("NAF_GPU_DEVICE")) : 0;
    FILE *f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END); long len = ftell(f); rewind(f);
    unsigned char *h_naf = (unsigned char *)malloc((size_t)len);
    if (fread(h_naf, 1, (size_t)len, f) != (size_t)len) { fprintf(stderr, "short read\n"); return 2; }
    fclose(f);

    naf_gpu_ctx *ctx[16]; void *d_naf[16], *d_part[16];
    for (int k = 0; k < n; k++) {
        int rc = naf_gpu_init(dev, &ctx[k]);
        if (rc) { fprintf(stderr, "naf_gpu_init: %s\n", naf_gpu_strerror(rc)); return 1; }
        TRY(ctx[k], naf_gpu_malloc(ctx[k], (size_t)len + 64, &d_naf[k]));
        TRY(ctx[k], naf_gpu_upload(ctx[k], d_naf[k], h_naf, (size_t)len));
    }
    naf_gpu_unnaf_opts o; o.out_type = NAF_OUT_DEFAULT; o.use_mask = 1; o.line_length = -1;
    size_t total = 0;
    TRY(ctx[0], naf_gpu_unnaf_size(ctx[0], d_naf[0], (size_t)len, &o, &total));
    void *d_whole, *d_gathered;
    TRY(ctx[0], naf_gpu_malloc(ctx[0], total + 64, &d_whole));
    TRY(ctx[0], naf_gpu_malloc(ctx[0], total + 64, &d_gathered));
    size_t got = 0;
    TRY(ctx[0], naf_gpu_unnaf(ctx[0], d_naf[0], (size_t)len, &o, d_whole, total + 64, &got));
    if (got != total) { fprintf(stderr, "size %zu != %zu\n", got, total); return 1; }

    /* ranges of unequal size, cut at arbitrary bytes */
    uint64_t off[17]; size_t plen[16];
    for (int k = 0; k <= n; k++) off[k] = (uint64_t)((double)total * k / n * (k == n ? 1.0 : (0.9 + 0.03 * k)));
    off[n] = total;
    for (int k = 0; k < n; k++) { if (off[k + 1] < off[k]) off[k + 1] = off[k]; plen[k] = (size_t)(off[k + 1] - off[k]); }
    const void *srcs[16];
    for (int k = 0; k < n; k++) {
        TRY(ctx[k], naf_gpu_malloc(ctx[k], plen[k] + 64, &d_part[k]));
        size_t m = 0;
        if (plen[k]) TRY(ctx[k], naf_gpu_unnaf_range(ctx[k], d_naf[k], (size_t)len, &o, off[k], off[k + 1], d_part[k], plen[k] + 64, &m));
        if (m != plen[k]) { fprintf(stderr, "range %d: %zu != %zu\n", k, m, plen[k]); return 1; }
        srcs[k] = d_part[k];
    }
    TRY(ctx[0], naf_gpu_gather_ranges(ctx[0], d_gathered, (naf_gpu_ctx *const *)ctx, srcs, off, plen, n));
    TRY(ctx[0], naf_gpu_synchronize(ctx[0]));
    unsigned char *a = (unsigned char *)malloc(total + 1), *b = (unsigned char *)malloc(total + 1);
    TRY(ctx[0], naf_gpu_download(ctx[0], a, d_whole, total));
    TRY(ctx[0], naf_gpu_download(ctx[0], b, d_gathered, total));
    int same = memcmp(a, b, total) == 0;
    for (int k = n - 1; k >= 0; k--) naf_gpu_shutdown(ctx[k]);
    if (!same) { fprintf(stderr, "gathered text differs from the whole-text decode\n"); return 1; }
    printf("gather ok: %d contexts, %zu bytes\n", n, total);
    return 0;
}
