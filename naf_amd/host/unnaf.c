/* unnaf -- NAF decompressor front end for the MI355X path.
 * Command line, outputs and messages follow unnaf/src/unnaf.c:197-456 and unnaf/src/output.c of the
 * reference; every decompression and the text re-emit run on the GPU through libnaf_gpu.so. */
#include "host_common.h"
#include <sys/mman.h>

typedef enum { UNDECIDED, FORMAT_NAME, PART_LIST, PART_SIZES, NUMBER_OF_SEQUENCES, TITLE, IDS, NAMES, LENGTHS, TOTAL_LENGTH,
               MASK, TOTAL_MASK_LENGTH, FOUR_BIT, DNA, MASKED_DNA, UNMASKED_DNA, SEQ, SEQUENCES, CHARCOUNT,
               FASTA, MASKED_FASTA, UNMASKED_FASTA, FASTQ } OUTPUT_TYPE;
static OUTPUT_TYPE out_type = UNDECIDED;
static bool use_mask = true, force_stdout = false, verbose = false;
static char *in_file_path = NULL, *out_file_path = NULL;
static bool line_length_is_specified = false; static long long requested_line_length = 0;
static FILE *OUT = NULL; static bool created_output_file = false, success = false;

static void done(int status, void *arg)
{
    (void)arg;
    if (!success && created_output_file && out_file_path) remove(out_file_path);
    trace_out();
    detach_report(status);                                        /* the foreground process leaves with this status now; what follows is nobody's wait */
    if (gpu_init_started) { pthread_join(gpu_init_thread, NULL); gpu_init_started = false; }       /* (an exit while the device is still being opened) */
    if (gpu) naf_gpu_shutdown(gpu);
}
static void set_out_type(OUTPUT_TYPE t) { if (out_type != UNDECIDED) die("only one output type should be specified\n"); out_type = t; }

static void set_line_length(char *str)
{
    long long a; int how = decimal_arg(str, &a);
    if (how == 0) die("can't parse the value of --line-length parameter\n");
    if (a < 0) die("negative line length specified\n");
    if (how != 2) die("can't parse the value of --line-length parameter\n");
    requested_line_length = a; line_length_is_specified = true;
}

static void show_help(void)
{
    msg("Usage: unnaf [OUTPUT-TYPE] [file.naf]\n"
        "Options for selecting output type:\n"
        "  --format        - File format version\n  --part-list     - List of parts\n  --sizes         - Part sizes\n"
        "  --number        - Number of sequences\n  --title         - Dataset title\n  --ids           - Sequence ids (accession numbers)\n"
        "  --names         - Full sequence names (including ids)\n  --lengths       - Sequence lengths\n  --total-length  - Sum of sequence lengths\n"
        "  --mask          - Masked region lengths\n  --4bit          - 4bit-encoded nucleotide sequence (binary data)\n"
        "  --seq           - Continuous concatenated sequence\n  --sequences     - One sequence per line, no names\n"
        "  --fasta         - FASTA-formatted sequences\n  --fastq         - FASTQ-formatted sequences\n"
        "Other options:\n  -o FILE         - Decompress into FILE\n  -c              - Write to standard output\n"
        "  --line-length N - Use lines of width N for FASTA output\n  --no-mask       - Ignore mask\n"
        "  --binary-stdout - Set stdout stream to binary mode.\n  --binary-stderr - Set stderr stream to binary mode.\n"
        "  --binary        - Shortcut for \"--binary-stdout --binary-stderr\"\n  -h, --help      - Show help\n  -V, --version   - Show version\n");
}

static void parse_command_line(int argc, char **argv)
{
    bool print_version = false;
    static const struct { const char *name; OUTPUT_TYPE t; } types[] = {
        {"--format", FORMAT_NAME}, {"--part-list", PART_LIST}, {"--sizes", PART_SIZES}, {"--number", NUMBER_OF_SEQUENCES}, {"--title", TITLE},
        {"--ids", IDS}, {"--names", NAMES}, {"--lengths", LENGTHS}, {"--total-length", TOTAL_LENGTH}, {"--mask", MASK},
        {"--total-mask-length", TOTAL_MASK_LENGTH}, {"--4bit", FOUR_BIT}, {"--seq", SEQ}, {"--sequences", SEQUENCES}, {"--charcount", CHARCOUNT},
        {"--fasta", FASTA}, {"--fastq", FASTQ}, {"--dna", DNA}, {"--masked-dna", MASKED_DNA}, {"--unmasked-dna", UNMASKED_DNA},
        {"--masked-fasta", MASKED_FASTA}, {"--unmasked-fasta", UNMASKED_FASTA} };
    /* the other options (unnaf/src/unnaf.c:282-353), as a table: one that takes a value is only recognised with an argument behind it */
    enum { OP_LINE_LENGTH, OP_OUT, OP_NO_MASK, OP_IGNORED, OP_HELP, OP_VERBOSE, OP_VERSION, OP_STDOUT };
    static const struct { const char *name; int op; bool value; } option_table[] = {
        { "--line-length", OP_LINE_LENGTH, true }, { "-o", OP_OUT, true }, { "--no-mask", OP_NO_MASK, false }, { "--binary-stdout", OP_IGNORED, false },
        { "--binary-stderr", OP_IGNORED, false }, { "--binary", OP_IGNORED, false }, { "--help", OP_HELP, false }, { "-h", OP_HELP, false },
        { "--verbose", OP_VERBOSE, false }, { "--version", OP_VERSION, false }, { "-V", OP_VERSION, false }, { "-c", OP_STDOUT, false } };
    for (int i = 1; i < argc; i++) {
        char *arg = argv[i];
        if (arg[0] != '-') {
            if (in_file_path) die("can process only one file at a time\n");
            if (!*arg) die("empty input path specified\n");
            in_file_path = arg; continue;
        }
        size_t k = 0;
        const size_t n_types = sizeof types / sizeof types[0], n_opts = sizeof option_table / sizeof option_table[0];
        while (k < n_types && strcmp(arg, types[k].name)) k++;
        if (k < n_types) { set_out_type(types[k].t); continue; }
        k = 0;
        while (k < n_opts && !(!strcmp(arg, option_table[k].name) && (!option_table[k].value || i < argc - 1))) k++;
        if (k == n_opts) die("unknown or incomplete argument \"%s\"\n", arg);
        char *v = option_table[k].value ? argv[++i] : NULL;
        switch (option_table[k].op) {
        case OP_LINE_LENGTH: set_line_length(v); break;
        case OP_OUT: if (out_file_path) die("double --out parameter\n"); if (!*v) die("empty --out parameter\n"); out_file_path = v; break;
        case OP_NO_MASK: use_mask = false; break;
        case OP_IGNORED: break;
        case OP_HELP: show_help(); exit(0);
        case OP_VERBOSE: verbose = true; break;
        case OP_VERSION: print_version = true; break;
        case OP_STDOUT: force_stdout = true; break;
        }
    }
    if (print_version) {
        msg("unnaf - NAF decompressor, version " VERSION ", " DATE "\nCopyright (c) " COPYRIGHT_YEARS " Kirill Kryukov\n");
        if (verbose) msg("MI355X path: libnaf_gpu (HIP, gfx950), zstd frames decoded on the GPU\n");
        exit(0);
    }
    if (force_stdout && out_file_path) die("-c and -o arguments can't be used together\n");
}

static const unsigned char *naf; static size_t naf_len; static naf_gpu_header H; static void *d_naf = NULL;

static int naf_fd = -1;            /* a regular file: mapped for the header walk and the title, read into HBM by the I/O lanes */
static void upload_to(naf_gpu_ctx *c, void **d)
{
    CTX_TRY(c, naf_gpu_malloc(c, naf_len + 64, d));
    if (naf_fd >= 0) CTX_TRY(c, naf_gpu_read_file(c, naf_fd, 0, naf_len, *d));
    else { CTX_TRY(c, naf_gpu_upload(c, *d, naf, naf_len)); CTX_TRY(c, naf_gpu_synchronize(c)); }
}
static void upload(void) { gpu_open(); if (d_naf) return; upload_to(gpu, &d_naf); phase("archive upload"); }

static unsigned char *load_section(int i, const char *what)
{
    upload();
    unsigned long long n = H.orig_size[i];
    void *d; GPU_TRY(naf_gpu_malloc(gpu, n + 64, &d));
    size_t got = 0;
    int rc = naf_gpu_zstd_decompress(gpu, (const char *)d_naf + H.payload_off[i], H.comp_size[i], 0, d, n, &got);
    if (rc || got != n) die("can't decompress %s\n", what);
    unsigned char *h = (unsigned char *)malloc(n + 1);
    if (!h) die("can't allocate %llu bytes\n", n + 1);
    GPU_TRY(naf_gpu_download(gpu, h, d, n)); h[n] = 0;
    naf_gpu_free(gpu, d);
    return h;
}

/* ids / names: N strings, each with its terminator inside the section (input.c:145-200: "not 0-terminated", then the walk that
 * stops at "can't read id %llu" when the section holds fewer than N of them -- "currupted" is the reference's spelling for ids) */
static unsigned char *load_strings(int i, const char *what, unsigned long long n_strings)
{
    unsigned long long n = H.orig_size[i];
    if (n == 0) die("corrupted %s - not 0-terminated\n", what);
    unsigned char *b = load_section(i, what);
    if (b[n - 1] != 0) die("corrupted %s - not 0-terminated\n", what);
    unsigned long long have = 0;
    for (const unsigned char *p = b, *e = b + n; p < e && have < n_strings; have++) p = (const unsigned char *)memchr(p, 0, (size_t)(e - p)) + 1;
    if (have < n_strings) { if (i == 0) die("currupted ids - can't read id %llu\n", have); else die("corrupted names - can't read name %llu\n", have); }
    return b;
}

/* ---- text output --------------------------------------------------------------------------------------------------------------
 * The text is produced in byte ranges (naf_gpu_unnaf_range): never more than RANGE bytes of it are resident, so archives whose
 * text exceeds HBM decode too, and with NAF_GPUS=a,b,... every device takes a contiguous share of the text and writes it into
 * its place of the output file on its own (per-GPU D2H + pwrite: when the consumer is the host there is nothing to gather over
 * xGMI first).  A pipe gets the ranges in order from one device. */
static size_t range_bytes(void)
{
    const char *e = getenv("NAF_GPU_RANGE_BYTES");
    unsigned long long v = e ? strtoull(e, NULL, 10) : 0;
    if (v < 4096) v = (unsigned long long)16 << 30;
    return (size_t)(v & ~4095ull);
}
typedef struct { int k, n, device; naf_gpu_unnaf_opts o; size_t total, lo, hi; off_t file_at; naf_gpu_ctx *c; } text_job;
static void *text_worker(void *arg)
{
    text_job *j = (text_job *)arg;
    naf_gpu_ctx *c = j->c;
    void *d_arc = d_naf;
    if (!c) {                                                   /* a device of its own: context and a copy of the archive */
        c = ctx_open(j->device);
        upload_to(c, &d_arc);
    }
    const size_t R = range_bytes(), span = j->hi - j->lo, cap = span < R ? span : R;
    void *d; CTX_TRY(c, naf_gpu_malloc(c, cap + 64, &d));
    for (size_t b = j->lo; b < j->hi; b += cap) {
        size_t e = b + cap < j->hi ? b + cap : j->hi, got = 0;
        if (j->n == 1 && cap == j->total) CTX_TRY(c, naf_gpu_unnaf(c, d_arc, naf_len, &j->o, d, cap, &got));       /* everything at once: the whole-text call overlaps its side streams */
        else CTX_TRY(c, naf_gpu_unnaf_range(c, d_arc, naf_len, &j->o, b, e, d, cap, &got));
        if (got != e - b) die("can't decompress sequence\n");
        if (j->n == 1 && getenv("NAF_GPU_CLI_TIMING")) { CTX_TRY(c, naf_gpu_synchronize(c)); phase("unnaf on the GPU (waited for: timing only)"); }
        CTX_TRY(c, naf_gpu_write_file(c, fileno(OUT), (uint64_t)j->file_at + b, d, got));
    }
    naf_gpu_free(c, d);
    if (c != gpu) { naf_gpu_free(c, d_arc); naf_gpu_shutdown(c); }
    return NULL;
}

static void run_text(int mode, int masking_allowed)
{
    upload();
    naf_gpu_unnaf_opts o = { mode, masking_allowed && use_mask, line_length_is_specified ? requested_line_length : -1 };
    size_t n = 0; GPU_TRY(naf_gpu_unnaf_size(gpu, d_naf, naf_len, &o, &n));
    if (!n) return;
    phase("size");
    fflush(OUT);
    off_t at = fd_pwrite_pos(fileno(OUT));                   /* -1: a pipe, or a file opened for appending (`>>`): ranges in order from one device */
    if (at >= 0) {
        devices_parse();
        int nd = n_devs; if ((size_t)nd > n / 4096 + 1) nd = (int)(n / 4096 + 1);
        size_t per = ((n + (size_t)nd - 1) / (size_t)nd + 4095) & ~(size_t)4095;
        text_job jobs[MAX_DEVS]; pthread_t th[MAX_DEVS];
        for (int k = 0; k < nd; k++) {
            size_t lo = (size_t)k * per < n ? (size_t)k * per : n, hi = lo + per < n ? lo + per : n;
            jobs[k] = (text_job){ k, nd, dev_ids[k], o, n, lo, hi, at, k == 0 ? gpu : NULL };
        }
        for (int k = 1; k < nd; k++) if (pthread_create(&th[k], NULL, text_worker, &jobs[k]) != 0) die("can't start a device thread\n");
        text_worker(&jobs[0]);
        for (int k = 1; k < nd; k++) pthread_join(th[k], NULL);
        if (lseek(fileno(OUT), at + (off_t)n, SEEK_SET) < 0) die("can't write to file - disk full?\n");
        phase("unnaf on the GPU + download + write");
        return;
    }
    const size_t R = range_bytes(), cap = n < R ? n : R;
    void *d; GPU_TRY(naf_gpu_malloc(gpu, cap + 64, &d));
    if (cap == n) {                                             /* the whole text at once: one pass over the side streams */
        size_t got = 0; GPU_TRY(naf_gpu_unnaf(gpu, d_naf, naf_len, &o, d, n, &got));
        GPU_TRY(naf_gpu_synchronize(gpu)); phase("unnaf on the GPU");
        write_from_device(OUT, d, got);
    } else for (size_t b = 0; b < n; b += cap) {
        size_t e = b + cap < n ? b + cap : n, got = 0;
        GPU_TRY(naf_gpu_unnaf_range(gpu, d_naf, naf_len, &o, b, e, d, cap, &got));
        write_from_device(OUT, d, got);
    }
    phase("download + write"); naf_gpu_free(gpu, d);
}

int main(int argc, char **argv)
{
    prog_name = "unnaf";
    on_exit(done, NULL);
    parse_command_line(argc, argv);
    if (in_file_path == NULL && isatty(fileno(stdin))) { err("no input specified, use \"unnaf -h\" for help\n"); exit(0); }
    FILE *IN = in_file_path ? fopen(in_file_path, "rb") : stdin;
    if (!IN) die("can't open input file\n");
    phase("start");
    /* every output but the few that the header alone answers needs the device: its start runs beside the reading of the archive */
    if (!(out_type == FORMAT_NAME || out_type == PART_LIST || out_type == PART_SIZES || out_type == NUMBER_OF_SEQUENCES || out_type == TITLE || out_type == TOTAL_LENGTH)) detach_teardown();
    if (!(out_type == FORMAT_NAME || out_type == PART_LIST || out_type == PART_SIZES || out_type == NUMBER_OF_SEQUENCES || out_type == TITLE || out_type == TOTAL_LENGTH)) gpu_open_early();
    struct stat ist;
    if (fd_is_regular(fileno(IN)) && fstat(fileno(IN), &ist) == 0 && ist.st_size > 0) {
        void *m = mmap(NULL, (size_t)ist.st_size, PROT_READ, MAP_PRIVATE, fileno(IN), 0);
        if (m != MAP_FAILED) { naf = (const unsigned char *)m; naf_len = (size_t)ist.st_size; naf_fd = fileno(IN); }
    }
    if (naf_fd < 0) { naf = read_all(IN, &naf_len); if (IN != stdin) fclose(IN); }
    phase("read archive");
    char eb[128] = "";
    if (naf_gpu_parse_header_host(naf, naf_len, &H, eb)) die("%s", eb);
    int has_title = (H.flags >> 6) & 1, has_ids = (H.flags >> 5) & 1, has_names = (H.flags >> 4) & 1, has_lengths = (H.flags >> 3) & 1,
        has_mask = (H.flags >> 2) & 1, has_data = (H.flags >> 1) & 1, has_quality = H.flags & 1;
    static const char *tn[4] = { "DNA", "RNA", "protein", "text" };
    if (out_type == UNDECIDED) out_type = has_quality ? FASTQ : FASTA;
    if ((out_type == DNA || out_type == MASKED_DNA || out_type == UNMASKED_DNA) && H.seq_type != NAF_SEQ_DNA) die("input has not DNA, but %s data\n", tn[H.seq_type]);
    if (out_type == FOUR_BIT && H.seq_type >= NAF_SEQ_PROTEIN) die("input has no 4-bit encoded data, but %s sequences\n", tn[H.seq_type]);

    bool to_orig = has_quality ? (out_type == FASTA) : (out_type == FASTQ);
    char *auto_path = NULL;
    if (to_orig && !force_stdout && in_file_path && !out_file_path && isatty(fileno(stdout))) {
        size_t len = strlen(in_file_path);
        if (len > 4 && !strcmp(in_file_path + len - 4, ".naf") && in_file_path[len - 5] != '/' && in_file_path[len - 5] != '\\') {
            auto_path = (char *)malloc(len - 3); memcpy(auto_path, in_file_path, len - 4); auto_path[len - 4] = 0; out_file_path = auto_path;
        }
    }
    if (out_file_path && !force_stdout) { OUT = fopen(out_file_path, "wb"); if (!OUT) die("can't create output file\n"); created_output_file = true; }
    else OUT = stdout;
    bool large = out_type == IDS || out_type == NAMES || out_type == LENGTHS || out_type == MASK || out_type == FOUR_BIT || out_type == DNA ||
                 out_type == MASKED_DNA || out_type == UNMASKED_DNA || out_type == SEQ || out_type == FASTA || out_type == MASKED_FASTA ||
                 out_type == UNMASKED_FASTA || out_type == FASTQ;
    if (large && !force_stdout && isatty(fileno(OUT)))
        die("output file not specified - please either specify output file with '-o' or '>', or use '-c' option to force writing to console\n");

    unsigned long long N = H.n_sequences;
    if (out_type == FORMAT_NAME) fprintf(OUT, "%s sequences%s in NAF format version %d\n", tn[H.seq_type], has_quality ? " with qualities" : "", H.version);
    else if (out_type == PART_LIST) {
        int printed = 0; const char *nm[7] = { "Title", "IDs", "Names", "Lengths", "Mask", "Data", "Quality" };
        int has[7] = { has_title, has_ids, has_names, has_lengths, has_mask, has_data, has_quality };
        for (int i = 0; i < 7; i++) if (has[i]) { fprintf(OUT, "%s%s", printed ? ", " : "", nm[i]); printed++; }
        fprintf(OUT, "\n");
    }
    else if (out_type == NUMBER_OF_SEQUENCES) fprintf(OUT, "%llu\n", N);
    else if (out_type == PART_SIZES) {
        if (has_title) fprintf(OUT, "Title: %llu\n", (unsigned long long)H.title_len);
        const char *nm[6] = { "IDs", "Names", "Lengths", "Mask", "Data", "Quality" };
        for (int i = 0; i < 6; i++) if (H.flags & (0x20 >> i))
            fprintf(OUT, "%s: %llu / %llu (%.3f%%)\n", nm[i], (unsigned long long)H.comp_size[i], (unsigned long long)H.orig_size[i], (double)H.comp_size[i] / (double)H.orig_size[i] * 100);
    }
    else if (out_type == TITLE) { if (has_title) fwrite(naf + H.title_off, 1, H.title_len, OUT); fputc('\n', OUT); }
    else if (N != 0) {
        if (out_type == IDS) { if (has_ids) { unsigned char *b = load_strings(0, "ids", N); const char *p = (const char *)b; for (unsigned long long i = 0; i < N; i++) { fprintf(OUT, "%s\n", p); p += strlen(p) + 1; } free(b); } }
        else if (out_type == NAMES) {
            if (has_ids || has_names) {
                unsigned char *a = has_ids ? load_strings(0, "ids", N) : NULL, *b = has_names ? load_strings(1, "names", N) : NULL;
                const char *p = (const char *)a, *q = (const char *)b;
                for (unsigned long long i = 0; i < N; i++) {
                    if (p) { fputs(p, OUT); p += strlen(p) + 1; }
                    if (q) { if (!a) fputs(q, OUT); else if (q[0]) { fputc(H.separator, OUT); fputs(q, OUT); } q += strlen(q) + 1; }
                    fputc('\n', OUT);
                }
                free(a); free(b);
            }
        }
        else if (out_type == LENGTHS) {
            if (has_lengths) { unsigned char *b = load_section(2, "lengths"); const unsigned int *u = (const unsigned int *)b; unsigned long long n = H.orig_size[2] / 4;
                for (unsigned long long i = 0; i < n; i++) { unsigned long long len = 0; while (i < n && u[i] == 4294967295u) { len += 4294967295llu; i++; } if (i < n) len += u[i]; fprintf(OUT, "%llu\n", len); } free(b); }
        }
        else if (out_type == TOTAL_LENGTH) { if (has_lengths) fprintf(OUT, "%llu\n", (unsigned long long)H.orig_size[4]); }
        else if (out_type == MASK) {
            if (has_mask) { unsigned char *b = load_section(3, "mask"); unsigned long long n = H.orig_size[3];
                for (unsigned long long i = 0; i < n; i++) { unsigned long long len = 0; while (i < n && b[i] == 255u) { len += 255llu; i++; } if (i < n) len += b[i]; fprintf(OUT, "%llu\n", len); } free(b); }
        }
        else if (out_type == TOTAL_MASK_LENGTH) {
            if (has_mask) { unsigned char *b = load_section(3, "mask"); unsigned long long t = 0; for (unsigned long long i = 0; i < H.orig_size[3]; i++) t += b[i]; fprintf(OUT, "%llu\n", t); free(b); }
            else fprintf(OUT, "0\n");
        }
        else if (out_type == FOUR_BIT) run_text(NAF_OUT_4BIT, 1);
        else if (out_type == DNA || out_type == SEQ || out_type == MASKED_DNA) run_text(NAF_OUT_SEQ, 1);
        else if (out_type == UNMASKED_DNA) run_text(NAF_OUT_SEQ, 0);
        else if (out_type == CHARCOUNT) { if (has_data) {          /* histogram of the --seq text (output.c:515-605), a byte range at a time */
                upload();
                naf_gpu_unnaf_opts o = { NAF_OUT_SEQ, use_mask, -1 }; size_t n = 0; GPU_TRY(naf_gpu_unnaf_size(gpu, d_naf, naf_len, &o, &n));
                const size_t R = range_bytes(), cap = n < R ? n : R;
                unsigned long long counts[256]; memset(counts, 0, sizeof counts);
                void *d; GPU_TRY(naf_gpu_malloc(gpu, cap + 64, &d));
                for (size_t b = 0; b < n; b += cap) {
                    size_t e = b + cap < n ? b + cap : n, got = 0;
                    if (cap == n) GPU_TRY(naf_gpu_unnaf(gpu, d_naf, naf_len, &o, d, n, &got));
                    else GPU_TRY(naf_gpu_unnaf_range(gpu, d_naf, naf_len, &o, b, e, d, cap, &got));
                    uint64_t cnt64[256]; GPU_TRY(naf_gpu_histogram(gpu, d, got, cnt64));
                    for (unsigned i = 0; i < 256; i++) counts[i] += cnt64[i];
                }
                for (unsigned i = 0; i < 33; i++) if (counts[i]) fprintf(OUT, "\\x%02X\t%llu\n", i, counts[i]);
                for (unsigned i = 33; i < 127; i++) if (counts[i]) fprintf(OUT, "%c\t%llu\n", (unsigned char)i, counts[i]);
                for (unsigned i = 127; i < 256; i++) if (counts[i]) fprintf(OUT, "\\x%02X\t%llu\n", i, counts[i]);
                naf_gpu_free(gpu, d); } }
        else if (out_type == SEQUENCES) run_text(NAF_OUT_SEQUENCES, 1);
        else if (out_type == FASTA || out_type == MASKED_FASTA) run_text(NAF_OUT_FASTA, 1);
        else if (out_type == UNMASKED_FASTA) run_text(NAF_OUT_FASTA, 0);
        else if (out_type == FASTQ) { if (!has_quality) die("FASTQ output requested, but input has no qualities\n"); run_text(NAF_OUT_FASTQ, 0); }
        else die("unknown output requested\n");
    }
    if (OUT != stdout) { if (fclose(OUT) != 0) die("can't close file - disk full?\n"); } else fflush(stdout);
    success = true;
    /* everything is written and closed: the process ends here, without the device-side teardown (freeing gigabytes of device memory,
     * streams, the runtime's own exit handlers: 0.1 - 0.2 s that nobody waits for; NAF_GPU_SLOW_EXIT=1 runs it) */
    trace_out(); fflush(NULL); detach_done(0);
    { const char *se = getenv("NAF_GPU_SLOW_EXIT"); if (!(se && se[0] == '1')) _exit(0); }
    return 0;
}
