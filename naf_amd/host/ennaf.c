/* ennaf -- NAF compressor front end for the MI355X path.
 * Command line, outputs and messages follow ennaf/src/ennaf.c:164-600 and process.c:75-96 of the
 * reference; parsing, 4-bit packing, mask extraction and zstd compression run on the GPU
 * through libnaf_gpu.so.  The reference's temp-file machinery (--temp-dir, --name, --keep-temp-files)
 * is accepted for command-line compatibility but unused: streams are assembled in HBM. */
#include "host_common.h"
#include <strings.h>

static bool verbose = false, no_mask = false, force_stdout = false, strict = false, well_formed = false;
static char *in_file_path = NULL, *out_file_path = NULL, *title = NULL, *temp_dir_arg = NULL;
/* Where a temporary file goes (the spill of a pipe larger than the device buffer, the parts of an input encoded in chunks): --temp-dir,
 * then $TMPDIR, then $TMP like the reference (ennaf.c:309-319) -- which dies without one of them; a regular file needs no temporary
 * file here at all, so this falls back to /tmp instead of refusing every run. */
static FILE *temp_file(const char *stem)
{
    const char *td = temp_dir_arg; char path[4096];
    if (!td || !*td) td = getenv("TMPDIR");
    if (!td || !*td) td = getenv("TMP");
    if (!td || !*td) td = "/tmp";
    snprintf(path, sizeof path, "%s/%s-XXXXXX", td, stem);
    int tfd = mkstemp(path); if (tfd < 0) die("can't create temporary file in \"%s\"\n", td);
    unlink(path);
    FILE *f = fdopen(tfd, "w+b"); if (!f) die("can't create temporary file in \"%s\"\n", td);
    return f;
}
static int level = 1, fmt_cmd = NAF_FMT_AUTO, seq_type = NAF_SEQ_DNA, long_log = 0;
static bool line_length_is_specified = false; static long long requested_line_length = 0;
static bool created_output_file = false, success = false;

static void done(int status, void *arg)
{
    (void)arg;
    if (!success && created_output_file && out_file_path) remove(out_file_path);
    trace_out();
    detach_report(status);                                        /* the foreground process leaves with this status now; what follows is nobody's wait */
    if (gpu_init_started) { pthread_join(gpu_init_thread, NULL); gpu_init_started = false; }       /* (an exit while the device is still being opened) */
    if (gpu) naf_gpu_shutdown(gpu);
}

static int parse_input_format(const char *s)
{
    if (!strcasecmp(s, "fasta") || !strcasecmp(s, "fa") || !strcasecmp(s, "fna")) return NAF_FMT_FASTA;
    if (!strcasecmp(s, "fastq") || !strcasecmp(s, "fq")) return NAF_FMT_FASTQ;
    return NAF_FMT_AUTO;
}
static void set_format(const char *s)
{
    if (fmt_cmd != NAF_FMT_AUTO) die("input format specified more than once\n");
    fmt_cmd = parse_input_format(s);
    if (fmt_cmd == NAF_FMT_AUTO) die("unknown input format specified: \"%s\"\n", s);
}
static void set_level(char *str) { char *end; long a = strtol(str, &end, 10); if (a < -131072 || a > 22 || *end) die("invalid value of --level, should be from %ld to %ld\n", -131072l, 22l); level = (int)a; }
static void show_help(void)
{
    msg("Usage: ennaf [OPTIONS] [infile]\nOptions:\n"
        "  -o FILE            - Write compressed output to FILE\n  -c                 - Write to standard output\n"
        "  -#, --level #      - Use compression level # (from %d to %d, default: 1)\n  --long N           - Use window of size 2^N for sequence stream (from %d to %d)\n"
        "  --temp-dir DIR     - Use DIR as temporary directory\n  --name NAME        - Use NAME as prefix for temporary files\n  --title TITLE      - Store TITLE as dataset title\n"
        "  --fasta            - Input is in FASTA format\n  --fastq            - Input is in FASTQ format\n  --dna              - Input sequence is DNA (default)\n"
        "  --rna              - Input sequence is RNA\n  --protein          - Input sequence is protein\n  --text             - Input sequence is text\n"
        "  --strict           - Fail on unexpected input characters\n  --line-length N    - Override line length to N\n  --verbose          - Verbose mode\n"
        "  --keep-temp-files  - Keep temporary files\n  --no-mask          - Don't store mask\n  -h, --help         - Show help\n  -V, --version      - Show version\n", -131072, 22, 10, 31);
}

/* The command line (ennaf/src/ennaf.c:360-430: the same options, texts and order of complaints), as a table: an option that takes a
 * value is only recognised with an argument behind it -- alone at the end it is "unknown or incomplete", like any unknown one. */
enum { OP_TEMP_DIR, OP_NAME, OP_TITLE, OP_LEVEL, OP_LINE_LENGTH, OP_LONG, OP_OUT, OP_IN, OP_IN_FORMAT, OP_HELP, OP_VERSION, OP_VERBOSE, OP_IGNORED,
       OP_NO_MASK, OP_FASTA, OP_FASTQ, OP_SEQ_TYPE, OP_WELL_FORMED, OP_STRICT, OP_STDOUT };
static const struct { const char *name; int op; bool value; int arg; } option_table[] = {
    { "--temp-dir", OP_TEMP_DIR, true, 0 }, { "--name", OP_NAME, true, 0 }, { "--title", OP_TITLE, true, 0 }, { "--level", OP_LEVEL, true, 0 },
    { "--line-length", OP_LINE_LENGTH, true, 0 }, { "--long", OP_LONG, true, 0 }, { "--out", OP_OUT, true, 0 }, { "--in", OP_IN, true, 0 },
    { "--in-format", OP_IN_FORMAT, true, 0 }, { "-o", OP_OUT, true, 0 },
    { "--help", OP_HELP, false, 0 }, { "-h", OP_HELP, false, 0 }, { "--version", OP_VERSION, false, 0 }, { "-V", OP_VERSION, false, 0 }, { "--verbose", OP_VERBOSE, false, 0 },
    { "--binary-stderr", OP_IGNORED, false, 0 }, { "--keep-temp-files", OP_IGNORED, false, 0 }, { "--no-mask", OP_NO_MASK, false, 0 },
    { "--fasta", OP_FASTA, false, 0 }, { "--fastq", OP_FASTQ, false, 0 }, { "--dna", OP_SEQ_TYPE, false, NAF_SEQ_DNA }, { "--rna", OP_SEQ_TYPE, false, NAF_SEQ_RNA },
    { "--protein", OP_SEQ_TYPE, false, NAF_SEQ_PROTEIN }, { "--text", OP_SEQ_TYPE, false, NAF_SEQ_TEXT }, { "--well-formed", OP_WELL_FORMED, false, 0 },
    { "--strict", OP_STRICT, false, 0 }, { "-c", OP_STDOUT, false, 0 } };
static bool print_version = false;
static void set_input_path(char *v) { if (in_file_path) die("can compress only one file at a time\n"); if (!*v) die("empty input file name\n"); in_file_path = v; }
static void apply_option(int op, int arg, char *v)
{
    long long a; int how;
    switch (op) {
    case OP_TEMP_DIR: if (temp_dir_arg) die("double --temp-dir parameter\n"); if (!*v) die("empty --temp-dir parameter\n"); temp_dir_arg = v; break;
    case OP_NAME: if (!*v) die("empty --name parameter\n"); break;
    case OP_TITLE: if (title) die("double --title parameter\n"); if (!*v) die("empty --title parameter\n"); title = v; break;
    case OP_LEVEL: set_level(v); break;
    case OP_LINE_LENGTH:
        how = decimal_arg(v, &a);
        if (how == 0) die("can't parse the value of --line-length parameter\n");
        if (a < 0) die("negative line length specified\n");
        if (how != 2) die("can't parse the value of --line-length parameter\n");
        requested_line_length = a; line_length_is_specified = true; break;
    case OP_LONG:
        if (decimal_arg(v, &a) != 2) die("can't parse the value of --long argument\n");
        if (a < 10) { warn("--long value of is %lld is smaller than the lowest supported value %d, using %d instead\n", a, 10, 10); a = 10; }
        else if (a > 31) { warn("--long value of is %lld is larger than the largest supported value %d, using %d instead\n", a, 31, 31); a = 31; }
        long_log = (int)a; break;
    case OP_OUT: if (out_file_path) die("double --out parameter\n"); if (!*v) die("empty --out parameter\n"); out_file_path = v; break;
    case OP_IN: set_input_path(v); break;
    case OP_IN_FORMAT: set_format(v); break;
    case OP_HELP: show_help(); exit(0);
    case OP_VERSION: print_version = true; break;
    case OP_VERBOSE: verbose = true; break;
    case OP_IGNORED: break;
    case OP_NO_MASK: no_mask = true; break;
    case OP_FASTA: set_format("fasta"); break;
    case OP_FASTQ: set_format("fastq"); break;
    case OP_SEQ_TYPE: seq_type = arg; break;
    case OP_WELL_FORMED: well_formed = true; break;
    case OP_STRICT: strict = true; break;
    case OP_STDOUT: force_stdout = true; break;
    }
}
static void parse_command_line(int argc, char **argv)
{
    for (int i = 1; i < argc; i++) {
        char *arg = argv[i];
        if (arg[0] != '-') { set_input_path(arg); continue; }
        const size_t n_opts = sizeof option_table / sizeof option_table[0];
        size_t k = 0;
        while (k < n_opts && !(!strcmp(arg, option_table[k].name) && (!option_table[k].value || i < argc - 1))) k++;
        if (k < n_opts) { apply_option(option_table[k].op, option_table[k].arg, option_table[k].value ? argv[++i] : NULL); continue; }
        if (arg[1] >= '0' && arg[1] <= '9') { set_level(arg + 1); continue; }                  /* -#: the level */
        die("unknown or incomplete argument \"%s\"\n", arg);
    }
    if (print_version) {
        msg("ennaf - NAF compressor, version " VERSION ", " DATE "\nCopyright (c) " COPYRIGHT_YEARS " Kirill Kryukov\n");
        if (verbose) msg("MI355X path: libnaf_gpu (HIP, gfx950), zstd frames encoded on the GPU\n");
        exit(0);
    }
    if (force_stdout && out_file_path) die("'-c' and '-o' can't be used together\n");
    if (well_formed && strict) die("'--well-formed' and '--strict' can't be used together\n");
}

static void report(const unsigned long long *n, const char *name)      /* process.c:75-96 */
{
    unsigned long long total = 0;
    for (unsigned i = 0; i < 257; i++) total += n[i];
    if (!total) return;
    msg("input has %llu unexpected %s characters:\n", total, name);
    for (unsigned i = 0; i < 32; i++) if (n[i]) msg("    '\\x%02X': %llu\n", i, n[i]);
    for (unsigned i = 32; i < 127; i++) if (n[i]) msg("    '%c': %llu\n", (unsigned char)i, n[i]);
    for (unsigned i = 127; i < 256; i++) if (n[i]) msg("    '\\x%02X': %llu\n", i, n[i]);
    if (n[256]) msg("    EOF: %llu\n", n[256]);
}

/* ---- one input on several GPUs (NAF_GPUS=a,b,...; include/naf_gpu.h "ennaf of ONE input on several GPUs") ----------------------
 * One host thread per device.  Every thread reads its nominal slice of the file straight into its device, the cuts are moved to
 * where a shard may begin (FASTA: behind a line end; FASTQ: a line whose ordinal is a multiple of four, from a census of every
 * slice), each thread fetches the few bytes its shard borrows from the next slice, and the shards run naf_gpu_ennaf_shard_begin /
 * _finish with the fixed-size shard records exchanged through memory in between.  The parts are written into their places of
 * the output file by the threads that made them (per-GPU D2H + pwrite), or in order by the main thread when the output is a pipe. */
#define SHARD_SPARE ((size_t)16 << 20)
typedef struct {
    int k, n, device, fd, prev_is_eol;
    naf_gpu_ctx *c;
    size_t a, b;                    /* nominal slice [a, b) of the file */
    void *d_buf; uint64_t lines, cut;
    size_t lo, len;                 /* the shard's text inside d_buf */
    void *d_pieces; naf_gpu_shard_pieces pc;
    int rc; char msg[512];
} enc_job;
static enc_job ejobs[MAX_DEVS];
static naf_gpu_shard_info sh_infos[MAX_DEVS];
static pthread_barrier_t sh_bar;
static int sh_fmt = 0, sh_fallback = 0, sh_out_fd = -1; static uint64_t sh_p0 = 0; static off_t sh_out_at = 0;
static naf_gpu_ennaf_opts sh_opts;
static naf_gpu_stitch_seg sh_segs[7 + 6 * MAX_DEVS]; static size_t sh_nsegs = 0;

static void job_fail(enc_job *j, int rc) { j->rc = rc; snprintf(j->msg, sizeof j->msg, "%s", naf_gpu_last_error(j->c)); }
#define JOB_TRY(j, call) do { int rc__ = (call); if (rc__ && !(j)->rc) job_fail((j), rc__); } while (0)
static void jobs_check(int n)            /* after a barrier: the first failed shard's message, as the reference would print it */
{
    for (int k = 0; k < n; k++) if (ejobs[k].rc) { size_t l = strlen(ejobs[k].msg); die("%s%s", ejobs[k].msg, (l && ejobs[k].msg[l - 1] == '\n') ? "" : "\n"); }
}

static void *enc_worker(void *arg)
{
    enc_job *j = (enc_job *)arg; const int k = j->k, n = j->n;
    if (!j->c) j->c = ctx_open(j->device);
    naf_gpu_ctx *c = j->c;
    const size_t nom = j->b - j->a;
    JOB_TRY(j, naf_gpu_malloc(c, nom + SHARD_SPARE + 64, &j->d_buf));
    if (!j->rc) JOB_TRY(j, naf_gpu_read_file(c, j->fd, j->a, nom, j->d_buf));
    if (k == 0 && !j->rc) {
        JOB_TRY(j, naf_gpu_ennaf_sniff(c, j->d_buf, nom, sh_opts.format, &sh_fmt, &sh_p0));
        if (!j->rc && (sh_fmt == 0 || sh_p0 >= nom)) sh_fallback = 1;            /* nothing but white space in the first slice: one device does it */
    }
    pthread_barrier_wait(&sh_bar);                                               /* 1: format known */
    if (k == 0) jobs_check(n);
    if (sh_fallback || ejobs[0].rc) return NULL;
    const size_t lo = k == 0 ? (size_t)sh_p0 : 0;
    if (sh_fmt == NAF_FMT_FASTQ) JOB_TRY(j, naf_gpu_ennaf_count_lines(c, (char *)j->d_buf + lo, nom - lo, j->prev_is_eol, &j->lines));
    pthread_barrier_wait(&sh_bar);                                               /* 2: census */
    uint64_t before = 0; for (int i = 0; i < k; i++) before += ejobs[i].lines;
    j->cut = 0;
    if (k > 0 && !j->rc) JOB_TRY(j, naf_gpu_ennaf_find_cut(c, j->d_buf, nom, sh_fmt, j->prev_is_eol, (4 - before % 4) % 4, &j->cut));
    pthread_barrier_wait(&sh_bar);                                               /* 3: cuts */
    if (k == 0) {
        jobs_check(n);
        for (int i = 1; i < n; i++) if (ejobs[i].cut > SHARD_SPARE || ejobs[i].cut >= ejobs[i].b - ejobs[i].a) sh_fallback = 2;    /* a line longer than the spare room, or a slice without a cut */
    }
    pthread_barrier_wait(&sh_bar);                                               /* 4: verdict on the cuts */
    if (sh_fallback) return NULL;
    const size_t borrow = k + 1 < n ? (size_t)ejobs[k + 1].cut : 0;
    if (borrow) JOB_TRY(j, naf_gpu_read_file(c, j->fd, j->b, borrow, (char *)j->d_buf + nom));
    j->lo = lo + (size_t)j->cut; j->len = nom - j->lo + borrow;
    if (!j->rc) JOB_TRY(j, naf_gpu_ennaf_shard_begin(c, (char *)j->d_buf + j->lo, j->len, &sh_opts, sh_fmt, (uint32_t)k, (uint32_t)n, &sh_infos[k]));
    pthread_barrier_wait(&sh_bar);                                               /* 5: shard records */
    if (k == 0) jobs_check(n);
    const size_t pcap = naf_gpu_ennaf_shard_bound(j->len);
    JOB_TRY(j, naf_gpu_malloc(c, pcap, &j->d_pieces));
    if (!j->rc) JOB_TRY(j, naf_gpu_ennaf_shard_finish(c, &sh_opts, sh_infos, j->d_pieces, pcap, &j->pc));
    pthread_barrier_wait(&sh_bar);                                               /* 6: parts */
    if (k == 0) return NULL;                                                     /* the main thread plans and opens the output, then calls enc_write */
    pthread_barrier_wait(&sh_bar);                                               /* 7: plan + output file */
    if (sh_out_fd >= 0)
        for (size_t i = 0; i < sh_nsegs; i++) if (sh_segs[i].shard == k && sh_segs[i].len)
            JOB_TRY(j, naf_gpu_write_file(c, sh_out_fd, (uint64_t)sh_out_at + sh_segs[i].dst_off, (char *)j->d_pieces + sh_segs[i].src_off, sh_segs[i].len));
    pthread_barrier_wait(&sh_bar);                                               /* 8: written */
    return NULL;
}

/* ---- an input larger than the device memory: the same shard protocol, one shard after the other on ONE device ------------------------
 * (process.c:143-150 reads 16 KiB at a time and never holds the input; here a CHUNK is what fits the device.)  Pass 1 walks the
 * file: every chunk ends behind its last complete line (FASTA) or record (FASTQ, from a census of its line starts), runs
 * naf_gpu_ennaf_shard_begin and keeps only the fixed-size shard record.  Pass 2 reads each chunk again, repeats the begin and -- now
 * that every record is known -- finishes it; the parts go to a temporary file ($TMPDIR, like the reference's per-stream files,
 * ennaf.c:478-505) and are joined into the archive by naf_gpu_ennaf_stitch_plan.  NAF_GPU_CHUNK_BYTES sets the chunk size
 * (default: a fifth of the free device memory); an input that fits one chunk takes the one-call path. */
#define MAX_CHUNKS NAF_GPU_MAX_SHARDS
static size_t ch_start[MAX_CHUNKS + 1]; static int ch_n = 0;
static naf_gpu_shard_info ch_infos[MAX_CHUNKS]; static naf_gpu_shard_pieces ch_pcs[MAX_CHUNKS];
static size_t ch_tmp_off[MAX_CHUNKS]; static FILE *ch_tmp = NULL;
static naf_gpu_stitch_seg ch_segs[7 + 6 * MAX_CHUNKS]; static size_t ch_nsegs = 0; static unsigned char ch_lit[4096 + 256];

static size_t chunk_bytes_wanted(size_t file_size)
{
    const char *e = getenv("NAF_GPU_CHUNK_BYTES");
    size_t c = 0, fr = 0, tot = 0;
    GPU_TRY(naf_gpu_mem_info(gpu, &fr, &tot));
    if (e && *e) { char *end; unsigned long long v = strtoull(e, &end, 10); if (*end || v < 4096) die("can't parse NAF_GPU_CHUNK_BYTES=\"%s\"\n", e); c = (size_t)v; }
    else c = fr / 5;
    if (c < 4096) c = 4096;
    if ((file_size + c - 1) / c > MAX_CHUNKS - 1) {
        c = (file_size + MAX_CHUNKS - 2) / (MAX_CHUNKS - 1);     /* cuts fall short of the nominal ends: leave one spare */
        /* a chunk needs its text, the streams split from it and their frames at once: about four times its size */
        if (c > fr / 4) die("input of %zu bytes needs %d chunks of %zu bytes, more than the device can hold at once (%zu bytes free)\n", file_size, MAX_CHUNKS - 1, c, fr);
    }
    return c;
}
/* offset just behind the last EOL-class byte of file range [a, b); a when there is none */
static size_t last_line_end(int fd, size_t a, size_t b)
{
    static unsigned char win[1 << 20];
    size_t hi = b;
    while (hi > a) {
        size_t lo = hi - a > sizeof win ? hi - sizeof win : a, len = hi - lo;
        if (pread(fd, win, len, (off_t)lo) != (ssize_t)len) die("can't read the input\n");
        for (size_t i = len; i-- > 0;) if (win[i] >= 0x0A && win[i] <= 0x0D) return lo + i + 1;
        hi = lo;
    }
    return a;
}
/* (offsets below count from `base`, the descriptor's position when the program started: `(head -c 100; ennaf) < file` encodes the rest) */
static bool encode_chunked(FILE *IN, size_t base, size_t fn, const naf_gpu_ennaf_opts *o, naf_gpu_ennaf_report *R, size_t *naf_len)
{
    const int fd = fileno(IN);
    const size_t C = chunk_bytes_wanted(fn);
    if (fn <= C) return false;
    if (title && strlen(title) >= 4096) return false;
    void *d_buf = NULL; GPU_TRY(naf_gpu_malloc(gpu, C + 64, &d_buf));
    int fmt = 0; uint64_t p0 = 0;
    /* pass 1: the cuts and the shard records */
    size_t pos = 0; int k = 0;
    while (pos < fn) {
        if (k >= MAX_CHUNKS) die("input needs more than %d chunks of %zu bytes\n", MAX_CHUNKS, C);
        size_t end = pos + C < fn ? pos + C : fn;
        GPU_TRY(naf_gpu_read_file(gpu, fd, base + pos, end - pos, d_buf));
        if (k == 0) {
            GPU_TRY(naf_gpu_ennaf_sniff(gpu, d_buf, end - pos, o->format, &fmt, &p0));
            if (fmt == 0 || p0 >= end - pos) { naf_gpu_free(gpu, d_buf); return false; }      /* a chunk of white space in front: the one-call path sorts it out */
            pos = (size_t)p0;
        }
        const char *text = (const char *)d_buf + (k == 0 ? (size_t)p0 : 0);
        if (end < fn) {                                       /* not the last chunk: stop behind the last complete line / record */
            size_t cut;
            if (fmt == NAF_FMT_FASTQ) {
                uint64_t lines = 0, off = 0;
                GPU_TRY(naf_gpu_ennaf_count_lines(gpu, text, end - pos, 1, &lines));
                if (lines < 5) die("a FASTQ record does not fit a chunk of %zu bytes (NAF_GPU_CHUNK_BYTES)\n", C);
                GPU_TRY(naf_gpu_ennaf_find_cut(gpu, text, end - pos, fmt, 1, (lines - 1) / 4 * 4, &off));
                cut = pos + (size_t)off;
            } else cut = last_line_end(fd, base + pos, base + end) - base;
            if (cut <= pos) die("a line does not fit a chunk of %zu bytes (NAF_GPU_CHUNK_BYTES)\n", C);
            end = cut;
        }
        ch_start[k] = pos;
        /* n_shards is not known yet: k + 2 says "not the last one", the records are completed below */
        GPU_TRY(naf_gpu_ennaf_shard_begin(gpu, text, end - pos, o, fmt, (uint32_t)k, (uint32_t)(end == fn ? k + 1 : (k + 2 > MAX_CHUNKS ? MAX_CHUNKS : k + 2)), &ch_infos[k]));
        pos = end; k++;
    }
    ch_n = k; ch_start[k] = fn;
    for (int i = 0; i < ch_n; i++) ch_infos[i].n_shards = (uint32_t)ch_n;
    /* pass 2: the parts */
    ch_tmp = temp_file("ennaf-gpu");
    size_t tmp_at = 0;
    for (k = 0; k < ch_n; k++) {
        const size_t a = ch_start[k], len = ch_start[k + 1] - a;
        GPU_TRY(naf_gpu_read_file(gpu, fd, base + a, len, d_buf));
        naf_gpu_shard_info again;
        GPU_TRY(naf_gpu_ennaf_shard_begin(gpu, d_buf, len, o, fmt, (uint32_t)k, (uint32_t)ch_n, &again));
        const size_t pcap = naf_gpu_ennaf_shard_bound(len);
        void *d_pieces = NULL; GPU_TRY(naf_gpu_malloc(gpu, pcap, &d_pieces));
        GPU_TRY(naf_gpu_ennaf_shard_finish(gpu, o, ch_infos, d_pieces, pcap, &ch_pcs[k]));
        ch_tmp_off[k] = tmp_at;
        if (ch_pcs[k].total) {
            fflush(ch_tmp);
            GPU_TRY(naf_gpu_write_file(gpu, fileno(ch_tmp), tmp_at, d_pieces, ch_pcs[k].total));
            tmp_at += ch_pcs[k].total;
        }
        GPU_TRY(naf_gpu_free(gpu, d_pieces));
    }
    GPU_TRY(naf_gpu_free(gpu, d_buf));
    size_t ll = 0; uint64_t nl = 0;
    int rc = naf_gpu_ennaf_stitch_plan(o, ch_infos, ch_pcs, (uint32_t)ch_n, ch_segs, sizeof ch_segs / sizeof ch_segs[0], &ch_nsegs, ch_lit, sizeof ch_lit, &ll, &nl, R);
    if (rc) die("can't join the parts of the archive: %s\n", naf_gpu_strerror(rc));
    *naf_len = (size_t)nl;
    return true;
}
/* the archive of a chunked run: header bytes from the plan, the parts from the temporary file, in archive order */
static void write_chunked(FILE *OUT)
{
    static unsigned char buf[1 << 20];
    for (size_t i = 0; i < ch_nsegs; i++) {
        const naf_gpu_stitch_seg *g = &ch_segs[i];
        if (!g->len) continue;
        if (g->shard < 0) { if (fwrite(ch_lit + g->src_off, 1, g->len, OUT) != g->len) die("can't write to file - disk full?\n"); continue; }
        size_t left = g->len; off_t at = (off_t)(ch_tmp_off[g->shard] + g->src_off);
        while (left) {
            size_t n = left < sizeof buf ? left : sizeof buf;
            if (pread(fileno(ch_tmp), buf, n, at) != (ssize_t)n) die("can't read temporary file\n");
            if (fwrite(buf, 1, n, OUT) != n) die("can't write to file - disk full?\n");
            left -= n; at += (off_t)n;
        }
    }
    fclose(ch_tmp); ch_tmp = NULL;
}

int main(int argc, char **argv)
{
    prog_name = "ennaf";
    on_exit(done, NULL);
    parse_command_line(argc, argv);
    if (in_file_path == NULL && isatty(fileno(stdin))) { err("no input specified, use \"ennaf -h\" for help\n"); exit(0); }
    int fmt_ext = NAF_FMT_AUTO;
    if (in_file_path) {
        char *ext = in_file_path + strlen(in_file_path);
        while (ext > in_file_path && *(ext - 1) != '/' && *(ext - 1) != '\\' && *(ext - 1) != '.') ext--;
        if (ext > in_file_path && *(ext - 1) == '.') fmt_ext = parse_input_format(ext);
    }
    FILE *IN = in_file_path ? fopen(in_file_path, "rb") : stdin;
    if (!IN) die("can't open input file\n");
    phase("start");
    detach_teardown();                                             /* from here on a worker process; the foreground one leaves when the archive is written (host_common.h) */
    gpu_open_early();                                              /* the device starts beside the rest of the set-up */
    char *auto_path = NULL;
    if (!force_stdout && !out_file_path && isatty(fileno(stdout))) {
        if (!in_file_path) die("output file is not specified\n");
        size_t len = strlen(in_file_path) + 5; auto_path = (char *)malloc(len); snprintf(auto_path, len, "%s.naf", in_file_path); out_file_path = auto_path;
    }
    naf_gpu_ennaf_opts o = { fmt_cmd, seq_type, no_mask, strict, level, line_length_is_specified ? requested_line_length : -1, title, long_log };
    static naf_gpu_ennaf_report R;
    FILE *OUT = stdout;
    void *d_naf = NULL; size_t naf_len = 0;
    bool sharded = false; static unsigned char sh_lit[4096 + 256]; size_t nthreads = 0; pthread_t th[MAX_DEVS];

    /* ---- several devices: a regular file of known size, cut into one slice per context */
    devices_parse();
    struct stat st;
    /* a regular file is read from the descriptor's current position on, whichever path takes it */
    off_t in_at = fd_is_regular(fileno(IN)) ? lseek(fileno(IN), 0, SEEK_CUR) : (off_t)-1; if (in_at < 0) in_at = 0;
    size_t in_left = 0;
    if (fd_is_regular(fileno(IN)) && fstat(fileno(IN), &st) == 0 && (size_t)st.st_size > (size_t)in_at) in_left = (size_t)st.st_size - (size_t)in_at;
    if (n_devs > 1 && in_left >= (size_t)n_devs * 65536 && (!title || strlen(title) < 4096)) {
        const size_t fn = in_left; const int n = n_devs;
        gpu_open();
        sh_opts = o;
        pthread_barrier_init(&sh_bar, NULL, (unsigned)n);
        for (int k = 0; k < n; k++) {
            enc_job *j = &ejobs[k]; memset(j, 0, sizeof *j);
            j->k = k; j->n = n; j->device = dev_ids[k]; j->fd = fileno(IN); j->c = k == 0 ? gpu : NULL;
            j->a = (size_t)in_at + fn / (size_t)n * (size_t)k; j->b = (size_t)in_at + (k + 1 < n ? fn / (size_t)n * (size_t)(k + 1) : fn);
            j->prev_is_eol = 1;
            if (k > 0) { unsigned char pb = 0; if (pread(j->fd, &pb, 1, (off_t)j->a - 1) != 1) die("can't read the input\n"); j->prev_is_eol = pb >= 0x0A && pb <= 0x0D; }
        }
        for (int k = 1; k < n; k++) { if (pthread_create(&th[k], NULL, enc_worker, &ejobs[k]) != 0) die("can't start a device thread\n"); nthreads++; }
        enc_worker(&ejobs[0]);
        if (sh_fallback) {
            for (int k = 1; k < n; k++) pthread_join(th[k], NULL);
            for (int k = 1; k < n; k++) if (ejobs[k].c) { naf_gpu_free(ejobs[k].c, ejobs[k].d_buf); naf_gpu_shutdown(ejobs[k].c); }
            naf_gpu_free(gpu, ejobs[0].d_buf);
            if (sh_fallback == 2) warn("a shard boundary falls into a line longer than %zu bytes; encoding on one device\n", SHARD_SPARE);
        } else {
            jobs_check(n);
            naf_gpu_shard_pieces pcs[MAX_DEVS]; for (int k = 0; k < n; k++) pcs[k] = ejobs[k].pc;
            size_t ll = 0; uint64_t nl = 0;
            int rc = naf_gpu_ennaf_stitch_plan(&o, sh_infos, pcs, (uint32_t)n, sh_segs, sizeof sh_segs / sizeof sh_segs[0], &sh_nsegs, sh_lit, sizeof sh_lit, &ll, &nl, &R);
            if (rc) die("can't join the parts of the archive: %s\n", naf_gpu_strerror(rc));
            naf_len = (size_t)nl; sharded = true;
            phase("ennaf on the GPUs");
        }
    }
    bool chunked = false;
    if (!sharded && in_left > 0) {
        gpu_open();
        chunked = encode_chunked(IN, (size_t)in_at, in_left, &o, &R, &naf_len);
        if (chunked) phase("ennaf in chunks");
    }
    /* ---- a pipe (process.c:143-150 reads one 16 KiB at a time and never knows how much is coming): chunks of it go through two pinned
     * buffers straight into a device buffer of PIPE_LIMIT bytes while the next chunk is being read; an input that ends inside it is
     * encoded from there.  One that does not is spilled -- what has arrived, then the rest -- into a temporary file ($TMPDIR, like
     * the reference's per-stream temporary files, ennaf.c:478-505), which then takes the path of a regular file of any size. */
    void *d_piped = NULL; size_t n_piped = 0; FILE *spill = NULL;
    if (!sharded && !chunked && in_left == 0 && !fd_is_regular(fileno(IN))) {
        io_open();
        const char *pe = getenv("NAF_GPU_PIPE_BYTES");
        size_t limit = pe ? (size_t)strtoull(pe, NULL, 10) : ((size_t)8 << 30);
        { size_t fr = 0, tot = 0; GPU_TRY(naf_gpu_mem_info(gpu, &fr, &tot)); if (limit > fr / 6) limit = fr / 6; if (limit < 4096) limit = 4096; }
        /* the device buffer starts at 256 MiB and grows fourfold up to the limit (a small pipe used to take the whole 8 GiB first) */
        size_t have = limit < ((size_t)256 << 20) ? limit : ((size_t)256 << 20);
        GPU_TRY(naf_gpu_malloc(gpu, have + 64, &d_piped));
        int cur = 0; bool eof = false, busy[2] = { false, false };
        while (!eof && n_piped < limit) {
            if (n_piped == have) {
                size_t more = have * 4 < limit ? have * 4 : limit; void *d_more = NULL;
                GPU_TRY(naf_gpu_malloc(gpu, more + 64, &d_more));
                GPU_TRY(naf_gpu_copy(gpu, d_more, d_piped, n_piped));
                GPU_TRY(naf_gpu_free(gpu, d_piped));                                   /* waits for the stream: the copy and the uploads are done */
                busy[0] = busy[1] = false; d_piped = d_more; have = more;
            }
            size_t want = have - n_piped < IO_CHUNK ? have - n_piped : IO_CHUNK, got = 0;
            if (busy[cur]) { GPU_TRY(naf_gpu_synchronize(gpu)); busy[0] = busy[1] = false; }      /* the upload that last read this buffer */
            while (got < want) { size_t r = fread((char *)io_pin[cur] + got, 1, want - got, IN); if (!r) { eof = true; break; } got += r; }
            if (got) { GPU_TRY(naf_gpu_upload(gpu, (char *)d_piped + n_piped, io_pin[cur], got)); busy[cur] = true; n_piped += got; cur ^= 1; }
        }
        GPU_TRY(naf_gpu_synchronize(gpu));
        if (!eof) { int c1 = fgetc(IN); if (c1 == EOF) eof = true; else ungetc(c1, IN); }
        if (!eof) {
            spill = temp_file("ennaf-gpu-in");
            write_from_device(spill, d_piped, n_piped);
            GPU_TRY(naf_gpu_free(gpu, d_piped)); d_piped = NULL;
            size_t total = n_piped, r;
            while ((r = fread(io_pin[0], 1, IO_CHUNK, IN)) > 0) { if (fwrite(io_pin[0], 1, r, spill) != r) die("can't write to temporary file - disk full?\n"); total += r; }
            if (fflush(spill) != 0) die("can't write to temporary file - disk full?\n");
            rewind(spill);
            if (IN != stdin) fclose(IN);
            IN = spill; in_at = 0; in_left = total;
            phase("pipe spilled to a temporary file");
            chunked = encode_chunked(IN, 0, in_left, &o, &R, &naf_len);
            if (chunked) phase("ennaf in chunks");
        } else phase("pipe read + upload");
    }
    if (!sharded && !chunked) {
        size_t n = n_piped; unsigned char *text = NULL;
        void *d_text = d_piped ? d_piped : read_to_device(IN, &n);                 /* regular file: straight to HBM through the pinned lanes */
        if (!d_text) text = read_all(IN, &n);
        phase("read + upload");
        gpu_open();
        size_t cap = naf_gpu_ennaf_bound(n);
        if (!d_text) { GPU_TRY(naf_gpu_malloc(gpu, n + 64, &d_text)); GPU_TRY(naf_gpu_upload(gpu, d_text, text, n)); }
        GPU_TRY(naf_gpu_malloc(gpu, cap, &d_naf));
        phase("allocation");
        GPU_TRY(naf_gpu_ennaf(gpu, d_text, n, &o, d_naf, cap, &naf_len, &R));
        phase("ennaf on the GPU");
    }
    if (IN != stdin) fclose(IN);
    if (R.format && fmt_ext != NAF_FMT_AUTO && fmt_ext != R.format) warn("input file extension does not match its actual format\n");
    if (fmt_ext != NAF_FMT_AUTO && fmt_cmd != NAF_FMT_AUTO && fmt_ext != fmt_cmd) warn("input file extension does not match format specified in the command line\n");
    if (out_file_path && !force_stdout) { OUT = fopen(out_file_path, "wb"); if (!OUT) die("can't create output file\n"); created_output_file = true; }
    if (verbose) msg("Output line length: %llu\n", line_length_is_specified ? (unsigned long long)requested_line_length : (unsigned long long)R.longest_line);
    if (chunked) write_chunked(OUT);
    else if (!sharded) write_from_device(OUT, d_naf, naf_len);
    else {
        const int n = n_devs;
        fflush(OUT);
        off_t at = fd_pwrite_pos(fileno(OUT));                        /* -1: a pipe or `>>`: the parts in order through the main thread */
        sh_out_fd = at >= 0 ? fileno(OUT) : -1; sh_out_at = at >= 0 ? at : 0;
        pthread_barrier_wait(&sh_bar);                                           /* 7: the workers write their parts */
        for (size_t i = 0; i < sh_nsegs; i++) {
            const naf_gpu_stitch_seg *g = &sh_segs[i];
            if (!g->len) continue;
            if (at >= 0) {
                if (g->shard < 0) { if (pwrite(fileno(OUT), sh_lit + g->src_off, g->len, at + (off_t)g->dst_off) != (ssize_t)g->len) die("can't write to file - disk full?\n"); }
                else if (g->shard == 0) JOB_TRY(&ejobs[0], naf_gpu_write_file(gpu, fileno(OUT), (uint64_t)at + g->dst_off, (char *)ejobs[0].d_pieces + g->src_off, g->len));
            }
        }
        pthread_barrier_wait(&sh_bar);                                           /* 8 */
        for (int k = 1; k < n; k++) pthread_join(th[k], NULL);
        jobs_check(n);
        if (at >= 0) { if (lseek(fileno(OUT), at + (off_t)naf_len, SEEK_SET) < 0) die("can't write to file - disk full?\n"); }
        else for (size_t i = 0; i < sh_nsegs; i++) {                             /* a pipe: the parts in order, through the main thread */
            const naf_gpu_stitch_seg *g = &sh_segs[i];
            if (!g->len) continue;
            if (g->shard < 0) { if (fwrite(sh_lit + g->src_off, 1, g->len, OUT) != g->len) die("can't write to file - disk full?\n"); continue; }
            unsigned char *h = (unsigned char *)malloc(g->len); if (!h) die("can't allocate %llu bytes\n", (unsigned long long)g->len);
            CTX_TRY(ejobs[g->shard].c, naf_gpu_download(ejobs[g->shard].c, h, (char *)ejobs[g->shard].d_pieces + g->src_off, g->len));
            if (fwrite(h, 1, g->len, OUT) != g->len) die("can't write to file - disk full?\n");
            free(h);
        }
        for (int k = 1; k < n; k++) naf_gpu_shutdown(ejobs[k].c);
    }
    phase("download + write");
    if (OUT != stdout) { if (fclose(OUT) != 0) die("can't close file - disk full?\n"); } else fflush(stdout);
    if (!well_formed) {
        static const char *tn[4] = { "DNA", "RNA", "protein", "text" };
        report((const unsigned long long *)R.unexpected_id, "id"); report((const unsigned long long *)R.unexpected_comment, "comment");
        report((const unsigned long long *)R.unexpected_seq, tn[seq_type]); report((const unsigned long long *)R.unexpected_qual, "quality");
    }
    if (verbose) msg("Processed %llu sequences\n", (unsigned long long)R.n_sequences);
    success = true;
    /* everything is written and closed: the process ends here, without the device-side teardown (freeing gigabytes of device memory,
     * streams, the runtime's own exit handlers: 0.1 - 0.2 s that nobody waits for; NAF_GPU_SLOW_EXIT=1 runs it) */
    trace_out(); fflush(NULL); detach_done(0);
    { const char *se = getenv("NAF_GPU_SLOW_EXIT"); if (!(se && se[0] == '1')) _exit(0); }
    return 0;
}
