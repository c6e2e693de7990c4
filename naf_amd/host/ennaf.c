/* ennaf -- NAF compressor front end for the MI355X path.
 * Command line, outputs and messages follow ennaf/src/ennaf.c:164-600 and process.c:75-96 of the
 * reference; parsing, 4-bit packing, mask extraction and zstd compression run on the GPU
 * through libnaf_gpu.so.  The reference's temp-file machinery (--temp-dir, --name, --keep-temp-files)
 * is accepted for command-line compatibility but unused: streams are assembled in HBM. */
#include "host_common.h"
#include <strings.h>

static bool verbose = false, no_mask = false, force_stdout = false, strict = false, well_formed = false;
static char *in_file_path = NULL, *out_file_path = NULL, *title = NULL;
static int level = 1, fmt_cmd = NAF_FMT_AUTO, seq_type = NAF_SEQ_DNA;
static bool line_length_is_specified = false; static long long requested_line_length = 0;
static bool created_output_file = false, success = false;

static void done(void) { if (!success && created_output_file && out_file_path) remove(out_file_path); if (gpu) naf_gpu_shutdown(gpu); }

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

static void parse_command_line(int argc, char **argv)
{
    bool print_version = false;
    for (int i = 1; i < argc; i++) {
        if (argv[i][0] == '-') {
            if (argv[i][1] == '-') {
                if (i < argc - 1) {
                    if (!strcmp(argv[i], "--temp-dir")) { i++; if (!*argv[i]) die("empty --temp-dir parameter\n"); continue; }
                    if (!strcmp(argv[i], "--name")) { i++; if (!*argv[i]) die("empty --name parameter\n"); continue; }
                    if (!strcmp(argv[i], "--title")) { i++; if (title) die("double --title parameter\n"); if (!*argv[i]) die("empty --title parameter\n"); title = argv[i]; continue; }
                    if (!strcmp(argv[i], "--level")) { i++; set_level(argv[i]); continue; }
                    if (!strcmp(argv[i], "--line-length")) { i++; long long a; int how = decimal_arg(argv[i], &a); if (how == 0) die("can't parse the value of --line-length parameter\n"); if (a < 0) die("negative line length specified\n"); if (how != 2) die("can't parse the value of --line-length parameter\n"); requested_line_length = a; line_length_is_specified = true; continue; }
                    if (!strcmp(argv[i], "--long")) { i++; long long a; if (decimal_arg(argv[i], &a) != 2) die("can't parse the value of --long argument\n");
                        if (a < 10) warn("--long value of is %lld is smaller than the lowest supported value %d, using %d instead\n", a, 10, 10);
                        else if (a > 31) warn("--long value of is %lld is larger than the largest supported value %d, using %d instead\n", a, 31, 31);
                        continue; }
                    if (!strcmp(argv[i], "--out")) { i++; if (out_file_path) die("double --out parameter\n"); if (!*argv[i]) die("empty --out parameter\n"); out_file_path = argv[i]; continue; }
                    if (!strcmp(argv[i], "--in")) { i++; if (in_file_path) die("can compress only one file at a time\n"); if (!*argv[i]) die("empty input file name\n"); in_file_path = argv[i]; continue; }
                    if (!strcmp(argv[i], "--in-format")) { i++; set_format(argv[i]); continue; }
                }
                if (!strcmp(argv[i], "--help")) { show_help(); exit(0); }
                if (!strcmp(argv[i], "--version")) { print_version = true; continue; }
                if (!strcmp(argv[i], "--verbose")) { verbose = true; continue; }
                if (!strcmp(argv[i], "--binary-stderr")) continue;
                if (!strcmp(argv[i], "--keep-temp-files")) continue;
                if (!strcmp(argv[i], "--no-mask")) { no_mask = true; continue; }
                if (!strcmp(argv[i], "--fasta")) { set_format("fasta"); continue; }
                if (!strcmp(argv[i], "--fastq")) { set_format("fastq"); continue; }
                if (!strcmp(argv[i], "--dna")) { seq_type = NAF_SEQ_DNA; continue; }
                if (!strcmp(argv[i], "--rna")) { seq_type = NAF_SEQ_RNA; continue; }
                if (!strcmp(argv[i], "--protein")) { seq_type = NAF_SEQ_PROTEIN; continue; }
                if (!strcmp(argv[i], "--text")) { seq_type = NAF_SEQ_TEXT; continue; }
                if (!strcmp(argv[i], "--well-formed")) { well_formed = true; continue; }
                if (!strcmp(argv[i], "--strict")) { strict = true; continue; }
            }
            if (i < argc - 1 && !strcmp(argv[i], "-o")) { i++; if (out_file_path) die("double --out parameter\n"); if (!*argv[i]) die("empty --out parameter\n"); out_file_path = argv[i]; continue; }
            if (!strcmp(argv[i], "-c")) { force_stdout = true; continue; }
            if (argv[i][1] >= '0' && argv[i][1] <= '9') { set_level(argv[i] + 1); continue; }
            if (!strcmp(argv[i], "-h")) { show_help(); exit(0); }
            if (!strcmp(argv[i], "-V")) { print_version = true; continue; }
            die("unknown or incomplete argument \"%s\"\n", argv[i]);
        }
        if (in_file_path) die("can compress only one file at a time\n");
        if (!*argv[i]) die("empty input file name\n");
        in_file_path = argv[i];
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

int main(int argc, char **argv)
{
    prog_name = "ennaf";
    atexit(done);
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
    size_t n = 0; unsigned char *text = NULL;
    void *d_text = read_to_device(IN, &n);                 /* regular file: straight to HBM through the pinned ring */
    if (!d_text) text = read_all(IN, &n);
    if (IN != stdin) fclose(IN);
    phase("read + upload");

    char *auto_path = NULL;
    if (!force_stdout && !out_file_path && isatty(fileno(stdout))) {
        if (!in_file_path) die("output file is not specified\n");
        size_t len = strlen(in_file_path) + 5; auto_path = (char *)malloc(len); snprintf(auto_path, len, "%s.naf", in_file_path); out_file_path = auto_path;
    }
    gpu_open();
    void *d_naf; size_t cap = naf_gpu_ennaf_bound(n), naf_len = 0;
    if (!d_text) { GPU_TRY(naf_gpu_malloc(gpu, n + 64, &d_text)); GPU_TRY(naf_gpu_upload(gpu, d_text, text, n)); }
    GPU_TRY(naf_gpu_malloc(gpu, cap, &d_naf));
    naf_gpu_ennaf_opts o = { fmt_cmd, seq_type, no_mask, strict, level, line_length_is_specified ? requested_line_length : -1, title };
    static naf_gpu_ennaf_report R;
    phase("allocation");
    GPU_TRY(naf_gpu_ennaf(gpu, d_text, n, &o, d_naf, cap, &naf_len, &R));
    phase("ennaf on the GPU");
    if (R.format && fmt_ext != NAF_FMT_AUTO && fmt_ext != R.format) warn("input file extension does not match its actual format\n");
    if (fmt_ext != NAF_FMT_AUTO && fmt_cmd != NAF_FMT_AUTO && fmt_ext != fmt_cmd) warn("input file extension does not match format specified in the command line\n");
    FILE *OUT = stdout;
    if (out_file_path && !force_stdout) { OUT = fopen(out_file_path, "wb"); if (!OUT) die("can't create output file\n"); created_output_file = true; }
    if (verbose) msg("Output line length: %llu\n", line_length_is_specified ? (unsigned long long)requested_line_length : (unsigned long long)R.longest_line);
    write_from_device(OUT, d_naf, naf_len);
    phase("download + write");
    if (OUT != stdout) { if (fclose(OUT) != 0) die("can't close file - disk full?\n"); } else fflush(stdout);
    if (!well_formed) {
        static const char *tn[4] = { "DNA", "RNA", "protein", "text" };
        report((const unsigned long long *)R.unexpected_id, "id"); report((const unsigned long long *)R.unexpected_comment, "comment");
        report((const unsigned long long *)R.unexpected_seq, tn[seq_type]); report((const unsigned long long *)R.unexpected_qual, "quality");
    }
    if (verbose) msg("Processed %llu sequences\n", (unsigned long long)R.n_sequences);
    success = true;
    return 0;
}
