/*
 * oracle_cli.c -- TEST INFRASTRUCTURE ONLY (see seqkit_oracle.h).
 *
 * `seqkit_oracle seq|fx2tab|stats|subseq|convert [flags] [files...]`: the reference commands
 * on the CPU restatement, with the reference's flag letters
 * (seqkit/cmd/seq.go:291-306, fx2tab.go:157-164, stat.go:292-294, subseq.go:549-564,
 * root.go:65-72).
 * Used by the tests to produce expected bytes and by bench.py as the timed CPU
 * baseline; never by the product.
 */
#include "seqkit_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <strings.h>

static int usage(void)
{
    fprintf(stderr, "usage: seqkit_oracle seq|fx2tab|stats|subseq|convert [flags] [files...]\n");
    return 255;
}

/* accepts "-x", "--long", and a value either as the next argument or after '=' */
static int is_flag(const char *a, const char *s, const char *l)
{
    if (s && strcmp(a, s) == 0) return 1;
    if (l && strcmp(a, l) == 0) return 1;
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) return usage();
    const char *cmd = argv[1];
    const char *files[1024]; int nfiles = 0;
    const char *base[64]; int nbase = 0;
    ora_seq_opts so = {0}; so.line_width = 60; so.min_len = so.max_len = -1; so.alphabet = -1; so.guess_len = 10000; so.validate_len = 10000;
    ora_fx2tab_opts fo = {0}; fo.guess_len = 10000;
    ora_stats_opts to = {0}; to.gap_letters = "- ."; to.guess_len = 10000;
    ora_subseq_opts uo = {0}; uo.gtf_tag = "gene_id"; uo.guess_len = 10000;
    static const char *chrs[256], *feats[256];
    const char *outfile = "-";
    int is_seq = !strcmp(cmd, "seq"), is_fx = !strcmp(cmd, "fx2tab"),
        is_st = !strcmp(cmd, "stats") || !strcmp(cmd, "stat"), is_sub = !strcmp(cmd, "subseq"),
        is_cv = !strcmp(cmd, "convert");
    ora_convert_opts co = {0}; co.to = ORA_Q_SANGER; co.nrecords = 1000; co.n_most_common = 4; co.illumina15_frac = 0.1;
    co.guess_len = 10000;
    if (!is_seq && !is_fx && !is_st && !is_sub && !is_cv) return usage();

    for (int i = 2; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-' || !strcmp(a, "-")) { if (nfiles < 1024) files[nfiles++] = a; continue; }
        /* bundled short bool flags, e.g. -rp or -nlg */
        if (!is_sub && !is_cv && a[1] != '-' && strlen(a) > 2) {
            int all_bool = 1;
            for (const char *p = a + 1; *p; p++) if (!strchr("rpnsqilgGHaT", *p)) all_bool = 0;
            if (all_bool && !(is_st && strchr(a, 'G')) && !(is_seq && strchr(a, 'G'))) {
                char one[3] = {'-', 0, 0};
                int rc = 0;
                for (const char *p = a + 1; *p; p++) {
                    one[1] = *p;
                    char *sub[3] = {argv[0], argv[1], one};
                    (void)sub; rc = 0;
                    if (is_seq) {
                        if (*p == 'r') so.reverse = 1; else if (*p == 'p') so.complement = 1;
                        else if (*p == 'n') so.only_name = 1; else if (*p == 's') so.only_seq = 1;
                        else if (*p == 'q') so.only_qual = 1; else if (*p == 'i') so.only_id = 1;
                        else if (*p == 'v') so.validate = 1;
                        else if (*p == 'g') so.remove_gaps = 1; else if (*p == 'l') so.lower_case = 1;
                        else if (*p == 'u') so.upper_case = 1;
                        else rc = 1;
                    } else if (is_fx) {
                        if (*p == 'l') fo.length = 1; else if (*p == 'g') fo.gc = 1;
                        else if (*p == 'G') fo.gc_skew = 1; else if (*p == 'n') fo.only_name = 1;
                        else if (*p == 'i') fo.only_id = 1; else if (*p == 'H') fo.header_line = 1;
                        else if (*p == 'a') fo.alphabet_col = 1; else rc = 1;
                    } else {
                        if (*p == 'a') to.all = 1; else if (*p == 'T') to.tabular = 1; else rc = 1;
                    }
                    if (rc) { fprintf(stderr, "unknown flag in %s\n", a); return 255; }
                }
                continue;
            }
        }
#define NEXT() (i + 1 < argc ? argv[++i] : (fprintf(stderr, "flag %s needs a value\n", a), exit(255), ""))
        if (is_flag(a, "-o", "--out-file")) { outfile = NEXT(); continue; }
        if (is_flag(a, "-w", "--line-width")) { so.line_width = atoi(NEXT()); continue; }
        if (is_flag(a, "-j", "--threads")) { (void)NEXT(); continue; }
        if (is_flag(a, NULL, "--quiet")) { so.quiet = 1; continue; }
        if (is_flag(a, NULL, "--id-ncbi")) { ora_set_id_regexp("\\|([^\\|]+)\\| "); continue; }          /* helper.go:129-131 */
        if (is_flag(a, NULL, "--id-regexp")) {
            const char *v = NEXT();
            if (!v || ora_set_id_regexp(v) != 0) {
                fprintf(stderr, "[ERRO] regular expression must contain \"(\" and \")\" to capture matched ID. default: `^([^\\s]+)\\s?`\n");
                return 255;
            }
            continue;
        }
        if (is_flag(a, NULL, "--alphabet-guess-seq-length")) {
            size_t v = (size_t)atoll(NEXT()); so.guess_len = fo.guess_len = to.guess_len = uo.guess_len = v; continue;
        }
        if (is_flag(a, "-t", "--seq-type")) {
            const char *v = NEXT();
            if (!strcasecmp(v, "dna")) so.alphabet = ORA_ALPHABET_DNAredundant;
            else if (!strcasecmp(v, "rna")) so.alphabet = ORA_ALPHABET_RNAredundant;
            else if (!strcasecmp(v, "protein")) so.alphabet = ORA_ALPHABET_Protein;
            else if (!strcasecmp(v, "unlimit")) so.alphabet = ORA_ALPHABET_Unlimit;
            else so.alphabet = -1;
            continue;
        }
        if (is_seq) {
            if (is_flag(a, "-r", "--reverse")) { so.reverse = 1; continue; }
            if (is_flag(a, "-p", "--complement")) { so.complement = 1; continue; }
            if (is_flag(a, "-n", "--name")) { so.only_name = 1; continue; }
            if (is_flag(a, "-s", "--seq")) { so.only_seq = 1; continue; }
            if (is_flag(a, "-q", "--qual")) { so.only_qual = 1; continue; }
            if (is_flag(a, "-i", "--only-id")) { so.only_id = 1; continue; }
            if (is_flag(a, "-v", "--validate-seq")) { so.validate = 1; continue; }
            if (is_flag(a, "-V", "--validate-seq-length")) { so.validate_len = (size_t)atoll(NEXT()); continue; }
            if (is_flag(a, "-g", "--remove-gaps")) { so.remove_gaps = 1; continue; }
            if (is_flag(a, "-G", "--gap-letters")) { so.gap_letters = NEXT(); continue; }
            if (is_flag(a, "-l", "--lower-case")) { so.lower_case = 1; continue; }
            if (is_flag(a, "-u", "--upper-case")) { so.upper_case = 1; continue; }
            if (is_flag(a, NULL, "--dna2rna")) { so.dna2rna = 1; continue; }
            if (is_flag(a, NULL, "--rna2dna")) { so.rna2dna = 1; continue; }
            if (is_flag(a, "-m", "--min-len")) { so.min_len = atoi(NEXT()); continue; }
            if (is_flag(a, "-M", "--max-len")) { so.max_len = atoi(NEXT()); continue; }
        } else if (is_cv) {
            if (is_flag(a, NULL, "--from") || is_flag(a, NULL, "--to")) {
                const int is_from = a[2] == 'f';
                const char *v = NEXT();
                int e = !strcasecmp(v, "sanger") ? ORA_Q_SANGER : !strcasecmp(v, "solexa") ? ORA_Q_SOLEXA
                      : !strcasecmp(v, "illumina-1.3+") ? ORA_Q_ILLUMINA13 : !strcasecmp(v, "illumina-1.5+") ? ORA_Q_ILLUMINA15
                      : !strcasecmp(v, "illumina-1.8+") ? ORA_Q_ILLUMINA18 : !strcmp(v, "") ? ORA_Q_UNKNOWN : -1;
                if (e < 0) { fprintf(stderr, "[ERRO] unsupported quality encoding: %s\n", v); return 255; }
                if (is_from) co.from = e; else co.to = e;
                continue;
            }
            if (is_flag(a, "-d", "--dry-run")) { co.dry_run = 1; continue; }
            if (is_flag(a, "-f", "--force")) { co.force = 1; continue; }
            if (is_flag(a, "-n", "--nrecords")) { co.nrecords = atoi(NEXT()); continue; }
            if (is_flag(a, "-N", "--thresh-B-in-n-most-common")) { co.n_most_common = atoi(NEXT()); continue; }
            if (is_flag(a, "-F", "--thresh-illumina1.5-frac")) { co.illumina15_frac = atof(NEXT()); continue; }
        } else if (is_sub) {
            if (is_flag(a, "-r", "--region")) { uo.region = NEXT(); continue; }
            if (is_flag(a, NULL, "--gtf")) { uo.gtf = NEXT(); continue; }
            if (is_flag(a, NULL, "--bed")) { uo.bed = NEXT(); continue; }
            if (is_flag(a, NULL, "--gtf-tag")) { uo.gtf_tag = NEXT(); continue; }
            if (is_flag(a, "-u", "--up-stream")) { uo.up_stream = atoi(NEXT()); continue; }
            if (is_flag(a, "-d", "--down-stream")) { uo.down_stream = atoi(NEXT()); continue; }
            if (is_flag(a, "-f", "--only-flank")) { uo.only_flank = 1; continue; }
            if (is_flag(a, NULL, "--chr") || is_flag(a, NULL, "--feature")) {      /* comma-separated, repeatable */
                const int is_chr = a[2] == 'c';
                char *v = strdup(NEXT());
                for (char *tok = strtok(v, ","); tok; tok = strtok(NULL, ",")) {
                    if (is_chr) { if (uo.nchr < 256) chrs[uo.nchr++] = tok; }
                    else if (uo.nfeature < 256) feats[uo.nfeature++] = tok;
                }
                continue;
            }
        } else if (is_fx) {
            if (is_flag(a, "-l", "--length")) { fo.length = 1; continue; }
            if (is_flag(a, "-g", "--gc")) { fo.gc = 1; continue; }
            if (is_flag(a, "-G", "--gc-skew")) { fo.gc_skew = 1; continue; }
            if (is_flag(a, "-n", "--name")) { fo.only_name = 1; continue; }
            if (is_flag(a, "-i", "--only-id")) { fo.only_id = 1; continue; }
            if (is_flag(a, "-H", "--header-line")) { fo.header_line = 1; continue; }
            if (is_flag(a, "-a", "--alphabet")) { fo.alphabet_col = 1; continue; }
            if (is_flag(a, "-B", "--base-content")) { if (nbase < 64) base[nbase++] = NEXT(); continue; }
        } else {
            if (is_flag(a, "-a", "--all")) { to.all = 1; continue; }
            if (is_flag(a, "-T", "--tabular")) { to.tabular = 1; continue; }
            if (is_flag(a, "-G", "--gap-letters")) { to.gap_letters = NEXT(); continue; }
        }
        fprintf(stderr, "unknown flag %s\n", a);
        return 255;
    }
    if (nfiles == 0) files[nfiles++] = "-";   /* seqkit/cmd/helper.go:55-56 */
    fo.nbase = nbase; fo.base = base;
    uo.chr = chrs; uo.feature = feats; uo.line_width = so.line_width;

    if (is_seq && so.alphabet >= 0 && so.alphabet != ORA_ALPHABET_Unlimit) {        /* seq.go:90-93 */
        fprintf(stderr, "[INFO] when flag -t (--seq-type) given, flag -v (--validate-seq) is automatically switched on\n");
        so.validate = 1;
    }
    FILE *out = stdout;
    if (strcmp(outfile, "-") != 0 && !(out = fopen(outfile, "wb"))) { perror(outfile); return 255; }
    static char obuf[1 << 20];
    setvbuf(out, obuf, _IOFBF, sizeof obuf);
    int rc = is_seq ? ora_cmd_seq(&so, nfiles, files, out, stderr)
           : is_fx  ? ora_cmd_fx2tab(&fo, nfiles, files, out, stderr)
           : is_sub ? ora_cmd_subseq(&uo, nfiles, files, out, stderr)
           : is_cv  ? ora_cmd_convert(&co, nfiles, files, out, stderr)
                    : (to.alphabet_plus1 = so.alphabet + 1, ora_cmd_stats(&to, nfiles, files, out, stderr));
    if (out != stdout) fclose(out); else fflush(out);
    return rc;
}
