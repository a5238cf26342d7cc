/*
 * seqkit_oracle.c -- TEST INFRASTRUCTURE ONLY (see seqkit_oracle.h).
 *
 * CPU restatement of the SeqKit hot path.  Each function cites the reference call
 * site it follows (paths relative to the reference checkout) and, where the
 * arithmetic lives in an un-vendored Go module, names the module function whose
 * published behaviour is restated.  Deliberately simple: one record at a time,
 * byte loops, no threads -- the shape of the Go code, not of the GPU code.
 */
#include "seqkit_oracle.h"

#include <ctype.h>
#include <math.h>
#include <regex.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <zlib.h>

/* ------------------------------------------------------------------ alphabets */
/* bio/seq alphabet.go (module github.com/shenwei356/bio, unpinned):
 * NewAlphabet(name, isUnlimit, letters, pairs, gap, ambiguous). */
typedef struct { const char *name; const char *letters; const char *pairs;
                 const char *gap; const char *ambiguous; } alpha_def;

static const alpha_def ALPHA[6] = {
    {"DNA", "acgtACGT", "tgcaTGCA", " -.", "nN."},
    {"DNAredundant", "acgtryswkmbdhvACGTRYSWKMBDHV", "tgcayrswmkvhdbTGCAYRSWMKVHDB", " -.", "nN."},
    {"RNA", "acguACGU", "ugcaUGCA", " -.", "nN."},
    {"RNAredundant", "acguryswkmbdhvACGURYSWKMBDHV", "ugcayrswmkvhdbUGCAYRSWMKVHDB", " -.", "nN."},
    {"Protein", "abcdefghijklmnpqrstvwyzABCDEFGHIJKLMNPQRSTVWYZ*_.",
                "abcdefghijklmnpqrstvwyzABCDEFGHIJKLMNPQRSTVWYZ*_.", " -", "xX*_."},
    {"Unlimit", "", "", "", ""},
};

const char *ora_alphabet_name(int a) { return (a >= 0 && a < 6) ? ALPHA[a].name : ""; }

void ora_alphabet_members(int a, uint8_t m[256])
{
    memset(m, 0, 256);
    if (a < 0 || a >= 6) return;
    for (const char *p = ALPHA[a].letters; *p; p++) m[(uint8_t)*p] = 1;
    for (const char *p = ALPHA[a].gap; *p; p++) m[(uint8_t)*p] = 1;
    for (const char *p = ALPHA[a].ambiguous; *p; p++) m[(uint8_t)*p] = 1;
}

long ora_validate_seq(const uint8_t *seq, size_t n, int alphabet, size_t limit)
{
    if (alphabet < 0 || alphabet >= 6 || alphabet == ORA_ALPHABET_Unlimit) return -1;
    uint8_t member[256];
    ora_alphabet_members(alphabet, member);
    if (limit != 0 && n > limit) n = limit;
    for (size_t i = 0; i < n; i++) if (!member[seq[i]]) return (long)i;
    return -1;
}

/* Alphabet.PairLetter: letters -> pairs, gap and ambiguous letters -> themselves.
 * A byte outside the table makes PairLetter return an error that
 * ComplementInplace ignores; the byte is left as it was (parity unpinned: no
 * vector in the reference tree complements a letter outside its alphabet). */
void ora_pair_table(int a, uint8_t lut[256])
{
    for (int i = 0; i < 256; i++) lut[i] = (uint8_t)i;
    if (a < 0 || a >= 6 || a == ORA_ALPHABET_Unlimit) return;
    const char *l = ALPHA[a].letters, *p = ALPHA[a].pairs;
    for (size_t i = 0; l[i]; i++) lut[(uint8_t)l[i]] = (uint8_t)p[i];
}

/* seq.GuessAlphabet + GuessAlphabetLessConservatively (bio/seq): the set of
 * distinct bytes of the prefix is tested for inclusion in DNA, RNA, DNAredundant,
 * RNAredundant, Protein, in that order; empty -> Unlimit; DNA/RNA are promoted to
 * their redundant forms.  Pinned by tests/test.sh:34-57 and doc/docs/usage.md:303-317. */
int ora_guess_alphabet(const uint8_t *seq, size_t n, size_t threshold)
{
    if (n == 0) return ORA_ALPHABET_Unlimit;
    if (threshold != 0 && n > threshold) n = threshold;
    uint8_t present[256]; memset(present, 0, sizeof present);
    for (size_t i = 0; i < n; i++) present[seq[i]] = 1;
    static const int order[5] = {ORA_ALPHABET_DNA, ORA_ALPHABET_RNA, ORA_ALPHABET_DNAredundant,
                                 ORA_ALPHABET_RNAredundant, ORA_ALPHABET_Protein};
    for (int k = 0; k < 5; k++) {
        uint8_t m[256]; ora_alphabet_members(order[k], m);
        int ok = 1;
        for (int b = 0; b < 256 && ok; b++) if (present[b] && !m[b]) ok = 0;
        if (ok) {
            if (order[k] == ORA_ALPHABET_DNA) return ORA_ALPHABET_DNAredundant;
            if (order[k] == ORA_ALPHABET_RNA) return ORA_ALPHABET_RNAredundant;
            return order[k];
        }
    }
    return ORA_ALPHABET_Unlimit;
}

/* ------------------------------------------------------------------ transforms */
/* seqkit/cmd/seq.go:187-189 -> (*Seq).ReverseInplace: byte swap from both ends;
 * quality is reversed with the bases (doc/docs/download.md:156, v0.4.5 fix). */
void ora_reverse_inplace(uint8_t *seq, uint8_t *qual, size_t n)
{
    for (size_t i = 0, j = n; i + 1 < j; i++) {
        j--;
        uint8_t t = seq[i]; seq[i] = seq[j]; seq[j] = t;
        if (qual) { t = qual[i]; qual[i] = qual[j]; qual[j] = t; }
    }
}

/* seqkit/cmd/seq.go:190-195 -> (*Seq).ComplementInplace: Seq[i] = PairLetter(Seq[i]). */
void ora_complement_inplace(uint8_t *seq, size_t n, int alphabet)
{
    uint8_t lut[256]; ora_pair_table(alphabet, lut);
    for (size_t i = 0; i < n; i++) seq[i] = lut[seq[i]];
}

/* seqkit/cmd/locate.go:154, subseq.go:399 -> RevCom; equals -r then -p (tests/test.sh:79-84). */
void ora_revcom_inplace(uint8_t *seq, uint8_t *qual, size_t n, int alphabet)
{
    ora_reverse_inplace(seq, qual, n);
    ora_complement_inplace(seq, n, alphabet);
}

/* ------------------------------------------------------------------ counting */
static uint64_t count_byte(const uint8_t *s, size_t n, uint8_t b)
{
    uint64_t c = 0;
    for (size_t i = 0; i < n; i++) c += (s[i] == b);
    return c;
}

/* (*Seq).BaseContent numerator: for every byte of `list` (duplicates count again),
 * occurrences of its upper-case form plus, if different, of its lower-case form. */
uint64_t ora_base_count(const uint8_t *seq, size_t n, const char *list)
{
    uint64_t sum = 0;
    for (const char *p = list; *p; p++) {
        uint8_t b = (uint8_t)*p, up = b, lo = b;
        if (b >= 'a' && b <= 'z') up = (uint8_t)(b - 32);
        if (b >= 'A' && b <= 'Z') lo = (uint8_t)(b + 32);
        sum += count_byte(seq, n, up);
        if (lo != up) sum += count_byte(seq, n, lo);
    }
    return sum;
}

/* seqkit/cmd/fx2tab.go:128,129,141: float64(sum)/float64(len); 0 for an empty record. */
double ora_base_content(const uint8_t *seq, size_t n, const char *list)
{
    if (n == 0) return 0.0;
    return (double)ora_base_count(seq, n, list) / (double)n;
}

/* seqkit/cmd/stat.go:125 -> byteutil.CountBytes: bytes of seq that are in `letters`. */
uint64_t ora_count_bytes(const uint8_t *seq, size_t n, const uint8_t *letters, size_t nletters)
{
    uint8_t in[256]; memset(in, 0, sizeof in);
    for (size_t i = 0; i < nletters; i++) in[letters[i]] = 1;
    uint64_t c = 0;
    for (size_t i = 0; i < n; i++) c += in[seq[i]];
    return c;
}

/* ------------------------------------------------------------------ CSR batches */
/* One record at a time, exactly as the Go loops do (seq.go:119-195, fx2tab.go:98-143,
 * stat.go:96-126); the batch is only the container. */
void ora_batch_transform(uint8_t *seq, uint8_t *qual, const uint64_t *offs, uint64_t nrec,
                         const uint8_t *lut, int do_reverse)
{
    for (uint64_t r = 0; r < nrec; r++) {
        uint8_t *s = seq + offs[r]; size_t n = (size_t)(offs[r + 1] - offs[r]);
        if (do_reverse) ora_reverse_inplace(s, qual ? qual + offs[r] : NULL, n);
        if (lut) for (size_t i = 0; i < n; i++) s[i] = lut[s[i]];
    }
}

void ora_batch_count(const uint8_t *seq, const uint64_t *offs, uint64_t nrec,
                     const uint8_t class_of[256], uint32_t nclass, uint64_t *counts)
{
    for (uint64_t r = 0; r < nrec; r++) {
        uint64_t *row = counts + r * nclass;
        for (uint32_t c = 0; c < nclass; c++) row[c] = 0;
        for (uint64_t i = offs[r]; i < offs[r + 1]; i++) {
            uint8_t c = class_of[seq[i]];
            if (c < nclass) row[c]++;
        }
    }
}

void ora_batch_stats(const uint8_t *seq, const uint64_t *offs, uint64_t nrec,
                     const uint8_t *gap_letters, size_t ngap, int64_t out[5])
{
    int64_t num = 0, len_sum = 0, gap_sum = 0, len_min = INT64_MAX, len_max = 0;
    for (uint64_t r = 0; r < nrec; r++) {
        int64_t l = (int64_t)(offs[r + 1] - offs[r]);
        num++; len_sum += l;
        if (l < len_min) len_min = l;
        if (l > len_max) len_max = l;
        gap_sum += (int64_t)ora_count_bytes(seq + offs[r], (size_t)l, gap_letters, ngap);
    }
    out[0] = num; out[1] = len_sum; out[2] = gap_sum; out[3] = len_min; out[4] = len_max;
}

/* ------------------------------------------------------------------ formatting */
/* seqkit/cmd/seq.go:243-253 -> byteutil.WrapByteSlice: '\n' after every `width`
 * bytes except after the last line (tests/test.sh:22-26 pins len%width==0). */
size_t ora_wrap(const uint8_t *s, size_t n, int width, uint8_t *out)
{
    if (width < 1 || n == 0) { memcpy(out, s, n); return n; }
    size_t w = (size_t)width, o = 0;
    for (size_t i = 0; i < n; i += w) {
        size_t k = n - i < w ? n - i : w;
        if (i) out[o++] = '\n';
        memcpy(out + o, s + i, k); o += k;
    }
    return o;
}

/* util/math.Round: trunc((f + 0.5/10^n) * 10^n) / 10^n  (seqkit/cmd/stat.go:158). */
double ora_round(double f, int n)
{
    double p = pow(10.0, n);
    return trunc((f + 0.5 / p) * p) / p;
}

/* seqkit/cmd/stat.go:297-306 */
int64_t ora_median(const int64_t *sorted, size_t l)
{
    if (l == 0) return 0;
    if (l % 2 == 0) return (sorted[l / 2 - 1] + sorted[l / 2]) / 2;
    return sorted[l / 2];
}

/* seqkit/cmd/stat.go:308-326 */
void ora_quartile(const int64_t *sorted, size_t l, int64_t *q1, int64_t *q2, int64_t *q3)
{
    *q1 = *q2 = *q3 = 0;
    if (l == 0) return;
    size_t c1, c2;
    if (l % 2 == 0) { c1 = l / 2; c2 = l / 2; } else { c1 = (l - 1) / 2; c2 = c1 + 1; }
    *q1 = ora_median(sorted, c1);
    *q2 = ora_median(sorted, l);
    *q3 = ora_median(sorted + c2, l - c2);
}

/* seqkit/cmd/stat.go:136-145 */
void ora_n50(const int64_t *sorted, size_t num, int64_t len_sum, int64_t *n50, int64_t *l50)
{
    int64_t sum = 0; *n50 = 0; *l50 = 0;
    for (size_t k = num; k-- > 0;) {
        sum += sorted[k];
        if ((sum << 1) >= len_sum) { *n50 = sorted[k]; *l50 = (int64_t)(num - k); break; }
    }
}

/* go-humanize Comma(int64) */
void ora_comma(int64_t v, char *buf)
{
    char d[32]; int neg = v < 0; uint64_t u = neg ? (uint64_t)(-(v + 1)) + 1 : (uint64_t)v;
    int n = snprintf(d, sizeof d, "%llu", (unsigned long long)u), o = 0;
    if (neg) buf[o++] = '-';
    for (int i = 0; i < n; i++) { if (i && (n - i) % 3 == 0) buf[o++] = ','; buf[o++] = d[i]; }
    buf[o] = 0;
}

/* Go strconv.FormatFloat(v,'f',-1,64): shortest digits that round-trip. */
static void fmt_shortest_f(double v, char *buf)
{
    for (int prec = 0; prec <= 20; prec++) {
        snprintf(buf, 64, "%.*f", prec, v);
        if (strtod(buf, NULL) == v) return;
    }
}

/* go-humanize Commaf(float64): commas in the integer part of the shortest form. */
void ora_commaf(double v, char *buf)
{
    char t[64]; int o = 0;
    if (v < 0) { buf[o++] = '-'; v = -v; }
    fmt_shortest_f(v, t);
    char *dot = strchr(t, '.'); int n = dot ? (int)(dot - t) : (int)strlen(t);
    for (int i = 0; i < n; i++) { if (i && (n - i) % 3 == 0) buf[o++] = ','; buf[o++] = t[i]; }
    if (dot) { strcpy(buf + o, dot); return; }
    buf[o] = 0;
}

/* Go fmt "%.Nf": correctly rounded like C's, but NaN/Inf are spelled NaN/+Inf/-Inf. */
void ora_fmt_fixed(double v, int prec, char *buf)
{
    if (isnan(v)) { strcpy(buf, "NaN"); return; }
    if (isinf(v)) { strcpy(buf, v > 0 ? "+Inf" : "-Inf"); return; }
    snprintf(buf, 400, "%.*f", prec, v);
}

/* ------------------------------------------------------------------ reader */
/* bio/seqio/fastx Reader restated: the format is decided by the first non-blank
 * byte ('>' FASTA, '@' FASTQ); a FASTA record runs to the next line starting with
 * '>'; '\r' and '\n' are dropped from the sequence, every other byte (interior
 * spaces included, doc/docs/usage.md:307-309) is kept; FASTQ is read as
 * head / sequence lines up to the '+' line / quality lines until as many quality
 * bytes as bases were read.  ID = leading run of non-space bytes of the head
 * (default --id-regexp "^([^\s]+)\s?", seqkit/cmd/root.go:68). */
struct ora_reader {
    gzFile fh;
    uint8_t *line; size_t line_cap; size_t line_len; int have_line; int eof;
    int is_fastq; int format_known;
    int alphabet; int forced_alphabet; size_t guess_threshold;
    uint8_t *name, *seq, *qual; size_t name_cap, seq_cap, qual_cap;
};

static int grow(uint8_t **p, size_t *cap, size_t need)
{
    if (need <= *cap) return 0;
    size_t c = *cap ? *cap : 256;
    while (c < need) c *= 2;
    uint8_t *q = (uint8_t *)realloc(*p, c);
    if (!q) return -1;
    *p = q; *cap = c; return 0;
}

/* reads one line without its "\n" / "\r\n"; returns 0 at EOF */
static int read_line(ora_reader *r)
{
    if (r->have_line) return 1;
    if (r->eof) return 0;
    r->line_len = 0;
    for (;;) {
        if (grow(&r->line, &r->line_cap, r->line_len + 65536) < 0) return 0;
        if (!gzgets(r->fh, (char *)r->line + r->line_len, (int)(r->line_cap - r->line_len))) {
            r->eof = 1;
            if (r->line_len == 0) return 0;
            break;
        }
        r->line_len += strlen((char *)r->line + r->line_len);
        if (r->line_len && r->line[r->line_len - 1] == '\n') break;
    }
    while (r->line_len && (r->line[r->line_len - 1] == '\n' || r->line[r->line_len - 1] == '\r'))
        r->line_len--;
    r->have_line = 1;
    return 1;
}

ora_reader *ora_reader_open(const char *path, int forced_alphabet, size_t guess_threshold)
{
    ora_reader *r = (ora_reader *)calloc(1, sizeof *r);
    if (!r) return NULL;
    r->fh = (strcmp(path, "-") == 0) ? gzdopen(0, "rb") : gzopen(path, "rb");
    if (!r->fh) { free(r); return NULL; }
    gzbuffer(r->fh, 1 << 20);
    r->alphabet = -1; r->forced_alphabet = forced_alphabet; r->guess_threshold = guess_threshold;
    return r;
}

/* \s of the default expression `^([^\s]+)\s?` is Go's (RE2's): [\t\n\f\r ] -- a vertical tab is not white space */
static size_t id_length(const uint8_t *name, size_t n)
{
    size_t i = 0;
    while (i < n && !(name[i] == ' ' || (name[i] >= '\t' && name[i] <= '\r' && name[i] != '\v'))) i++;
    return i == 0 ? n : i;   /* regexp does not match -> whole head */
}

/* a custom ID expression (ora_set_id_regexp) */
static regex_t g_id_re;
static int g_id_re_set = 0;

int ora_set_id_regexp(const char *go)
{
    if (g_id_re_set) { regfree(&g_id_re); g_id_re_set = 0; }
    if (!go || !strcmp(go, "^([^\\s]+)\\s?")) return 0;
    if (!strchr(go, '(') || !strchr(go, ')')) return -1;
    /* Perl classes -> POSIX: outside brackets \s = [[:space:]] ..., inside brackets \s = [:space:] ...,
     * any other escaped byte inside brackets stands for itself */
    size_t n = strlen(go);
    char *ere = (char *)malloc(16 * n + 16), *w = ere;
    int in_br = 0;
    for (size_t i = 0; i < n; i++) {
        char c = go[i];
        if (c == '\\' && i + 1 < n) {
            char d = go[++i];
            const char *cls = d == 's' || d == 'S' ? "[:space:]" : d == 'd' || d == 'D' ? "[:digit:]"
                            : d == 'w' || d == 'W' ? "[:alnum:]_" : NULL;
            if (cls) {
                const int neg = d == 'S' || d == 'D' || d == 'W';
                if (in_br) w += sprintf(w, "%s", cls);            /* (negated classes inside brackets: not needed here) */
                else w += sprintf(w, "[%s%s]", neg ? "^" : "", cls);
            } else if (in_br) *w++ = d;
            else { *w++ = '\\'; *w++ = d; }
            continue;
        }
        if (!in_br && c == '[') { in_br = 1; *w++ = c; if (i + 1 < n && go[i + 1] == '^') *w++ = go[++i]; continue; }
        if (in_br && c == ']') in_br = 0;
        *w++ = c;
    }
    *w = 0;
    const int rc = regcomp(&g_id_re, ere, REG_EXTENDED);
    free(ere);
    if (rc != 0) return -1;
    g_id_re_set = 1;
    return 0;
}

/* ID of a head line: where it starts and how long it is */
static const uint8_t *id_of(uint8_t *name, size_t n, size_t *len)
{
    if (!g_id_re_set) { *len = id_length(name, n); return name; }
    regmatch_t m[2];
    const uint8_t saved = name[n];
    name[n] = 0;                                   /* the reader's name buffer has room for it */
    const int hit = regexec(&g_id_re, (const char *)name, 2, m, 0) == 0 && m[1].rm_so >= 0;
    name[n] = saved;
    if (!hit) { *len = n; return name; }
    *len = (size_t)(m[1].rm_eo - m[1].rm_so);
    return name + m[1].rm_so;
}

int ora_reader_next(ora_reader *r, ora_record *rec)
{
    /* skip blank lines between records */
    for (;;) {
        if (!read_line(r)) return 0;
        if (r->line_len) break;
        r->have_line = 0;
    }
    if (!r->format_known) {
        if (r->line[0] == '>') r->is_fastq = 0;
        else if (r->line[0] == '@') r->is_fastq = 1;
        else return -1;
        r->format_known = 1;
    }
    if (r->line[0] != (r->is_fastq ? '@' : '>')) return -1;
    size_t nl = r->line_len - 1;
    if (grow(&r->name, &r->name_cap, nl + 1) < 0) return -2;
    memcpy(r->name, r->line + 1, nl);
    r->have_line = 0;
    size_t sl = 0, ql = 0;
    if (!r->is_fastq) {
        while (read_line(r)) {
            if (r->line_len && r->line[0] == '>') break;
            if (grow(&r->seq, &r->seq_cap, sl + r->line_len + 1) < 0) return -2;
            for (size_t i = 0; i < r->line_len; i++)
                if (r->line[i] != '\r' && r->line[i] != '\n') r->seq[sl++] = r->line[i];
            r->have_line = 0;
        }
    } else {
        while (read_line(r)) {
            if (r->line_len && r->line[0] == '+') { r->have_line = 0; break; }
            if (grow(&r->seq, &r->seq_cap, sl + r->line_len + 1) < 0) return -2;
            memcpy(r->seq + sl, r->line, r->line_len); sl += r->line_len;
            r->have_line = 0;
        }
        while (ql < sl && read_line(r)) {
            if (grow(&r->qual, &r->qual_cap, ql + r->line_len + 1) < 0) return -2;
            memcpy(r->qual + ql, r->line, r->line_len); ql += r->line_len;
            r->have_line = 0;
        }
        if (ql != sl) return -3;
    }
    if (r->alphabet < 0)
        r->alphabet = r->forced_alphabet >= 0 ? r->forced_alphabet
                                              : ora_guess_alphabet(r->seq, sl, r->guess_threshold);
    rec->name = r->name; rec->name_len = nl; rec->id = id_of(r->name, nl, &rec->id_len);
    rec->seq = r->seq; rec->seq_len = sl;
    rec->qual = r->qual; rec->qual_len = ql;
    return 1;
}

int ora_reader_is_fastq(const ora_reader *r) { return r->is_fastq; }
int ora_reader_alphabet(const ora_reader *r) { return r->alphabet; }

void ora_reader_close(ora_reader *r)
{
    if (!r) return;
    gzclose(r->fh);
    free(r->line); free(r->name); free(r->seq); free(r->qual); free(r);
}

/* ------------------------------------------------------------------ seq */
/* seqkit/cmd/seq.go:114-280, flags -r -p -n -s -q -i -m -M -w. */
size_t ora_remove_gaps(uint8_t *seq, uint8_t *qual, size_t len, const char *letters)
{
    uint8_t gap[256];
    memset(gap, 0, sizeof gap);
    for (const char *p = letters; *p; p++) gap[(uint8_t)*p] = 1;
    size_t j = 0;
    for (size_t i = 0; i < len; i++) {
        if (gap[seq[i]]) continue;
        seq[j] = seq[i];
        if (qual) qual[j] = qual[i];
        j++;
    }
    return j;
}

int ora_cmd_seq(const ora_seq_opts *o, int nfiles, const char **files, FILE *out, FILE *err)
{
    uint8_t *wrapbuf = NULL; size_t wrapcap = 0;
    int is_fastq = 0, check_type = 1;
    int once = 1;                                    /* seq.go:107: one "already RNA/DNA" warning per run */
    const char *gap_letters = o->gap_letters ? o->gap_letters : "- .";
    /* seq.go:71-78, 96-98 */
    if (gap_letters[0] == 0) { fprintf(err, "[ERRO] value of flag -G (--gap-letters) should not be empty\n"); return 255; }
    for (const char *p = gap_letters; *p; p++)
        if ((uint8_t)*p > 127) { fprintf(err, "[ERRO] value of -G (--gap-letters) contains non-ASCII characters\n"); return 255; }
    if (o->lower_case && o->upper_case) {
        fprintf(err, "[ERRO] could not give both flags -l (--lower-case) and -u (--upper-case)\n");
        return 255;
    }
    for (int f = 0; f < nfiles; f++) {
        ora_reader *r = ora_reader_open(files[f], o->alphabet, o->guess_len);
        if (!r) { fprintf(err, "[ERRO] open %s: no such file or directory\n", files[f]); return 255; }
        ora_record rec; int rc;
        while ((rc = ora_reader_next(r, &rec)) == 1) {
            if (o->validate) {
                /* the reader builds every record with seq.NewSeq, which validates it: the error
                 * ends the command before the record is filtered or printed (seq.go:120-127) */
                const int vab = ora_reader_alphabet(r);
                const long bad = ora_validate_seq(rec.seq, rec.seq_len, vab, o->validate_len);
                if (bad >= 0) {
                    fflush(out);
                    fprintf(err, "[ERRO] error when parsing seq: %.*s (invalid %s letter: %c)\n", (int)rec.id_len,
                            (const char *)rec.id, ora_alphabet_name(vab), rec.seq[bad]);
                    ora_reader_close(r); free(wrapbuf); return 255;
                }
            }
            if (o->min_len >= 0 && rec.seq_len < (size_t)o->min_len) continue;
            if (o->max_len >= 0 && rec.seq_len > (size_t)o->max_len) continue;
            int width = ora_reader_is_fastq(r) ? 0 : o->line_width;       /* seq.go:136-138 */
            if (check_type) { is_fastq = rec.qual_len > 0; check_type = 0; } /* seq.go:140-145 */
            int pq = is_fastq, pn = 1, ps = 1;
            if (o->only_name && o->only_seq) { pn = ps = 1; }
            else if (o->only_name) { pn = 1; ps = 0; pq = 0; }
            else if (o->only_seq) { pn = 0; ps = 1; pq = 0; }
            else if (o->only_qual) {
                if (!is_fastq) {
                    fprintf(err, "[ERRO] FASTA format has no quality. So do not just use flag -q (--qual)\n");
                    ora_reader_close(r); free(wrapbuf); return 255;
                }
                pn = 0; ps = 0; pq = 1;
            }
            if (pn) {
                size_t hl = o->only_id ? rec.id_len : rec.name_len;
                const uint8_t *hp = o->only_id ? rec.id : rec.name;
                if (ps) fputc(is_fastq ? '@' : '>', out);
                fwrite(hp, 1, hl, out); fputc('\n', out);
            }
            int ab = ora_reader_alphabet(r);
            if (o->reverse) ora_reverse_inplace(rec.seq, rec.qual_len ? rec.qual : NULL, rec.seq_len);
            if (o->complement) {
                if ((!o->quiet && ab == ORA_ALPHABET_Protein) || ab == ORA_ALPHABET_Unlimit)     /* once per record */
                    fprintf(err, "[WARN] Complement does no take effect on protein/unlimit sequence\n");
                ora_complement_inplace(rec.seq, rec.seq_len, ab);
            }
            if (o->remove_gaps) {                                        /* seq.go:196-198 */
                const size_t kept = ora_remove_gaps(rec.seq, rec.qual_len ? rec.qual : NULL, rec.seq_len, gap_letters);
                if (rec.qual_len) rec.qual_len = kept;
                rec.seq_len = kept;
            }
            if (ps) {
                /* seq.go:200-241: only when the sequence is printed; bytes.ToLower / ToUpper on ASCII
                 * text (bytes beyond 127 are left alone here: parity unpinned) */
                if (o->dna2rna) {
                    if (ab == ORA_ALPHABET_RNA || ab == ORA_ALPHABET_RNAredundant) {
                        if (once) { fprintf(err, "[WARN] it's already RNA, no need to convert\n"); once = 0; }
                    } else {
                        for (size_t i = 0; i < rec.seq_len; i++)
                            if (rec.seq[i] == 't') rec.seq[i] = 'u'; else if (rec.seq[i] == 'T') rec.seq[i] = 'U';
                    }
                }
                if (o->rna2dna) {
                    if (ab == ORA_ALPHABET_DNA || ab == ORA_ALPHABET_DNAredundant) {
                        if (once) { fprintf(err, "[WARN] it's already DNA, no need to convert\n"); once = 0; }
                    } else {
                        for (size_t i = 0; i < rec.seq_len; i++)
                            if (rec.seq[i] == 'u') rec.seq[i] = 't'; else if (rec.seq[i] == 'U') rec.seq[i] = 'T';
                    }
                }
                if (o->lower_case) {
                    for (size_t i = 0; i < rec.seq_len; i++) if (rec.seq[i] >= 'A' && rec.seq[i] <= 'Z') rec.seq[i] += 32;
                } else if (o->upper_case) {
                    for (size_t i = 0; i < rec.seq_len; i++) if (rec.seq[i] >= 'a' && rec.seq[i] <= 'z') rec.seq[i] -= 32;
                }
            }
            size_t need = rec.seq_len + (width > 0 ? rec.seq_len / (size_t)width : 0) + 1;
            if (need > wrapcap) { wrapbuf = (uint8_t *)realloc(wrapbuf, need); wrapcap = need; }
            if (ps) {
                size_t n = ora_wrap(rec.seq, rec.seq_len, width, wrapbuf);
                fwrite(wrapbuf, 1, n, out); fputc('\n', out);
            }
            if (pq) {
                if (!o->only_qual) fputs("+\n", out);
                size_t n = ora_wrap(rec.qual, rec.qual_len, width, wrapbuf);
                fwrite(wrapbuf, 1, n, out); fputc('\n', out);
            }
        }
        ora_reader_close(r);
        if (rc < 0) { fprintf(err, "[ERRO] error when parsing %s\n", files[f]); free(wrapbuf); return 255; }
    }
    free(wrapbuf);
    return 0;
}

/* ------------------------------------------------------------------ fx2tab */
/* The numeric columns of fx2tab rows (fx2tab.go:127-143) for n records given by their base counts, without
 * the records themselves: per record "\t<len>\t<GC>\t<GC-skew>\t<content of g>\n", with
 * g = float64(ng)/float64(len), c = float64(nc)/float64(len) divided separately, GC = (g+c)*100,
 * skew = (g-c)/(g+c)*100, content = g*100, each through ora_fmt_fixed(.., 2); a record of length 0 has g = c = 0
 * (fx2tab.go: BaseContent of an empty sequence).  Returns the bytes written (out holds >= 64 * n). */
size_t ora_tab_number_rows(const uint64_t *ng, const uint64_t *nc, const uint64_t *len, size_t n, char *out)
{
    char num[400];
    size_t o = 0;
    for (size_t i = 0; i < n; i++) {
        const double g = len[i] ? (double)ng[i] / (double)len[i] : 0.0;
        const double c = len[i] ? (double)nc[i] / (double)len[i] : 0.0;
        o += (size_t)sprintf(out + o, "\t%llu", (unsigned long long)len[i]);
        ora_fmt_fixed((g + c) * 100, 2, num); o += (size_t)sprintf(out + o, "\t%s", num);
        ora_fmt_fixed((g - c) / (g + c) * 100, 2, num); o += (size_t)sprintf(out + o, "\t%s", num);
        ora_fmt_fixed(g * 100, 2, num); o += (size_t)sprintf(out + o, "\t%s", num);
        out[o++] = '\n';
    }
    return o;
}

static int cmp_u8(const void *a, const void *b) { return (int)*(const uint8_t *)a - (int)*(const uint8_t *)b; }

/* seqkit/cmd/fx2tab.go:68-149. */
int ora_cmd_fx2tab(const ora_fx2tab_opts *o, int nfiles, const char **files, FILE *out, FILE *err)
{
    char num[400];
    if (o->header_line) {
        fputs("#name\tseq\tqual", out);
        if (o->length) fputs("\tlength", out);
        if (o->gc) fputs("\tGC", out);
        if (o->gc_skew) fputs("\tGC-Skew", out);
        for (int i = 0; i < o->nbase; i++) fprintf(out, "\t%s", o->base[i]);
        if (o->alphabet_col) fputs("\talphabet", out);
        fputc('\n', out);
    }
    for (int f = 0; f < nfiles; f++) {
        ora_reader *r = ora_reader_open(files[f], -1, o->guess_len);
        if (!r) { fprintf(err, "[ERRO] open %s: no such file or directory\n", files[f]); return 255; }
        ora_record rec; int rc;
        while ((rc = ora_reader_next(r, &rec)) == 1) {
            fwrite(o->only_id ? rec.id : rec.name, 1, o->only_id ? rec.id_len : rec.name_len, out);
            fputc('\t', out);
            if (!o->only_name) fwrite(rec.seq, 1, rec.seq_len, out);
            fputc('\t', out);
            if (!o->only_name) fwrite(rec.qual, 1, rec.qual_len, out);
            if (o->length) fprintf(out, "\t%zu", rec.seq_len);
            double g = 0, c = 0;
            if (o->gc || o->gc_skew) {                      /* two separate divisions */
                g = ora_base_content(rec.seq, rec.seq_len, "G");
                c = ora_base_content(rec.seq, rec.seq_len, "C");
            }
            if (o->gc) { ora_fmt_fixed((g + c) * 100, 2, num); fprintf(out, "\t%s", num); }
            if (o->gc_skew) { ora_fmt_fixed((g - c) / (g + c) * 100, 2, num); fprintf(out, "\t%s", num); }
            for (int i = 0; i < o->nbase; i++) {
                ora_fmt_fixed(ora_base_content(rec.seq, rec.seq_len, o->base[i]) * 100, 2, num);
                fprintf(out, "\t%s", num);
            }
            if (o->alphabet_col) {                           /* fx2tab.go:167-180 */
                uint8_t present[256], letters[256]; int n = 0;
                memset(present, 0, sizeof present);
                for (size_t i = 0; i < rec.seq_len; i++) present[rec.seq[i]] = 1;
                for (int b = 0; b < 256; b++) if (present[b]) letters[n++] = (uint8_t)b;
                qsort(letters, (size_t)n, 1, cmp_u8);
                fputc('\t', out); fwrite(letters, 1, (size_t)n, out);
            }
            fputc('\n', out);
        }
        ora_reader_close(r);
        if (rc < 0) { fprintf(err, "[ERRO] error when parsing %s\n", files[f]); return 255; }
    }
    return 0;
}

/* ------------------------------------------------------------------ stats */
typedef struct { const char *file; const char *format; char type[16];
                 int64_t num, len_sum, gap_sum, len_min, len_max, n50, l50, q1, q2, q3;
                 double len_avg; } stat_info;

static int cmp_i64(const void *a, const void *b)
{
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : x > y;
}

/* go-prettytable: every cell padded to its column's widest entry (header included),
 * AlignRight columns padded on the left, columns joined by Separator, '\n' per row.
 * Left-aligned padding of a LAST column is parity unpinned (stats ends in a
 * right-aligned column, so it never occurs here). */
static void pretty_table(FILE *out, int ncol, const int *align_right, int nrow, char ***cells)
{
    int width[16];
    for (int c = 0; c < ncol; c++) {
        width[c] = 0;
        for (int r = 0; r < nrow; r++) { int l = (int)strlen(cells[r][c]); if (l > width[c]) width[c] = l; }
    }
    for (int r = 0; r < nrow; r++) {
        for (int c = 0; c < ncol; c++) {
            if (c) fputs("  ", out);                       /* stat.go:236 */
            if (align_right[c]) fprintf(out, "%*s", width[c], cells[r][c]);
            else fprintf(out, "%-*s", width[c], cells[r][c]);
        }
        fputc('\n', out);
    }
}

/* seqkit/cmd/stat.go:78-268. */
int ora_cmd_stats(const ora_stats_opts *o, int nfiles, const char **files, FILE *out, FILE *err)
{
    const char *gaps = o->gap_letters ? o->gap_letters : "- .";
    if (!*gaps) { fprintf(err, "[ERRO] value of flag -G (--gap-letters) should not be empty\n"); return 255; }
    for (const char *p = gaps; *p; p++)
        if ((uint8_t)*p > 127) { fprintf(err, "[ERRO] value of -G (--gap-letters) contains non-ASCII characters\n"); return 255; }
    stat_info *infos = (stat_info *)calloc((size_t)nfiles, sizeof *infos);
    for (int f = 0; f < nfiles; f++) {
        ora_reader *r = ora_reader_open(files[f], o->alphabet_plus1 - 1, o->guess_len);      /* stat.go:87 */
        if (!r) { fprintf(err, "[ERRO] open %s: no such file or directory\n", files[f]); free(infos); return 255; }
        stat_info *s = &infos[f];
        s->file = files[f]; s->format = ""; s->len_min = INT64_MAX;
        int64_t *lens = NULL; size_t cap = 0;
        ora_record rec; int rc;
        while ((rc = ora_reader_next(r, &rec)) == 1) {
            if (s->num == 0) s->format = rec.qual_len > 0 ? "FASTQ" : "FASTA";
            int64_t l = (int64_t)rec.seq_len;
            if (o->all) {
                if ((size_t)s->num >= cap) { cap = cap ? cap * 2 : 1024; lens = (int64_t *)realloc(lens, cap * sizeof *lens); }
                lens[s->num] = l;
            }
            s->num++; s->len_sum += l;
            if (l < s->len_min) s->len_min = l;
            if (l > s->len_max) s->len_max = l;
            s->gap_sum += (int64_t)ora_count_bytes(rec.seq, rec.seq_len, (const uint8_t *)gaps, strlen(gaps));
        }
        int ab = ora_reader_alphabet(r);
        ora_reader_close(r);
        if (rc < 0) { fprintf(err, "[ERRO] error when parsing %s\n", files[f]); free(lens); free(infos); return 255; }
        if (ab == ORA_ALPHABET_DNAredundant) strcpy(s->type, "DNA");
        else if (ab == ORA_ALPHABET_RNAredundant) strcpy(s->type, "RNA");
        else strcpy(s->type, ora_alphabet_name(ab));
        if (o->all) {
            qsort(lens, (size_t)s->num, sizeof *lens, cmp_i64);
            ora_n50(lens, (size_t)s->num, s->len_sum, &s->n50, &s->l50);
            ora_quartile(lens, (size_t)s->num, &s->q1, &s->q2, &s->q3);
        }
        free(lens);
        if (s->num == 0) { s->len_sum = s->gap_sum = s->len_min = 0; s->len_avg = 0; s->n50 = s->l50 = 0; }
        else s->len_avg = ora_round((double)s->len_sum / (double)s->num, 1);
    }
    if (o->tabular) {
        fputs("file\tformat\ttype\tnum_seqs\tsum_len\tmin_len\tavg_len\tmax_len", out);
        if (o->all) fputs("\tQ1\tQ2\tQ3\tsum_gap\tN50", out);
        fputc('\n', out);
        for (int f = 0; f < nfiles; f++) {
            stat_info *s = &infos[f];
            fprintf(out, "%s\t%s\t%s\t%lld\t%lld\t%lld\t%.1f\t%lld", s->file, s->format, s->type,
                    (long long)s->num, (long long)s->len_sum, (long long)s->len_min, s->len_avg,
                    (long long)s->len_max);
            if (o->all)
                fprintf(out, "\t%lld\t%lld\t%lld\t%lld\t%lld", (long long)s->q1, (long long)s->q2,
                        (long long)s->q3, (long long)s->gap_sum, (long long)s->n50);
            fputc('\n', out);
        }
        free(infos);
        return 0;
    }
    int ncol = o->all ? 13 : 8;
    static const char *hdr[13] = {"file", "format", "type", "num_seqs", "sum_len", "min_len", "avg_len",
                                  "max_len", "Q1", "Q2", "Q3", "sum_gap", "N50"};
    static const int right[13] = {0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    int nrow = nfiles + 1;
    char ***cells = (char ***)calloc((size_t)nrow, sizeof *cells);
    for (int r = 0; r < nrow; r++) {
        cells[r] = (char **)calloc((size_t)ncol, sizeof **cells);
        for (int c = 0; c < ncol; c++) cells[r][c] = (char *)calloc(1, 4352);
    }
    for (int c = 0; c < ncol; c++) strcpy(cells[0][c], hdr[c]);
    for (int f = 0; f < nfiles; f++) {
        stat_info *s = &infos[f]; char **row = cells[f + 1];
        snprintf(row[0], 4352, "%s", s->file); strcpy(row[1], s->format); strcpy(row[2], s->type);
        ora_comma(s->num, row[3]); ora_comma(s->len_sum, row[4]); ora_comma(s->len_min, row[5]);
        ora_commaf(s->len_avg, row[6]); ora_comma(s->len_max, row[7]);
        if (o->all) {
            ora_comma(s->q1, row[8]); ora_comma(s->q2, row[9]); ora_comma(s->q3, row[10]);
            ora_comma(s->gap_sum, row[11]); ora_comma(s->n50, row[12]);
        }
    }
    pretty_table(out, ncol, right, nrow, cells);
    for (int r = 0; r < nrow; r++) { for (int c = 0; c < ncol; c++) free(cells[r][c]); free(cells[r]); }
    free(cells); free(infos);
    return 0;
}

/* ------------------------------------------------------------------ subseq */
int ora_sub_location(int64_t length, int64_t start, int64_t end, int64_t *s, int64_t *e)
{
    *s = 0; *e = 0;
    if (length == 0) return 0;
    if (start < 1) {
        if (start == 0) start = 1;
        else {
            if (end < 0 && start > end) return 0;
            start = -start > length ? 1 : length + start + 1;
        }
    }
    if (start > length) return 0;
    if (end > length) end = length;
    if (end < 1) {
        if (end == 0) end = -1;
        end = length + end + 1;
    }
    if (start - 1 > end) return 0;
    *s = start; *e = end;
    return 1;
}

void ora_batch_subseq(const uint8_t *seq, const uint8_t *qual, const uint64_t *offs, uint64_t nrec,
                      const ora_slice *slices, uint64_t n, const uint8_t *lut,
                      uint8_t *out_seq, uint8_t *out_qual, uint64_t *out_offs)
{
    uint64_t w = 0;
    (void)nrec;
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t a = offs[slices[i].record], len = offs[slices[i].record + 1] - a;
        int64_t s, e;
        out_offs[i] = w;
        if (!ora_sub_location((int64_t)len, slices[i].start, slices[i].end, &s, &e)) continue;
        const uint64_t m = e >= s ? (uint64_t)(e - s + 1) : 0;
        memcpy(out_seq + w, seq + a + (uint64_t)(s - 1), m);
        if (qual) memcpy(out_qual + w, qual + a + (uint64_t)(s - 1), m);
        if (slices[i].minus) {                       /* RevComInplace = ReverseInplace + ComplementInplace */
            ora_reverse_inplace(out_seq + w, qual ? out_qual + w : NULL, m);
            if (lut) for (uint64_t k = 0; k < m; k++) out_seq[w + k] = lut[out_seq[w + k]];
        }
        w += m;
    }
    out_offs[n] = w;
}

typedef struct feature {
    char *chr, *type, *label;     /* label: value of the chosen GTF tag / BED name ("" if absent) */
    int64_t start, end;           /* 1-based, inclusive */
    char strand;                  /* '+', '-', or 0 = not given */
} feature;

static char *dup_lower(const char *s, size_t n)
{
    char *d = (char *)malloc(n + 1);
    for (size_t i = 0; i < n; i++) d[i] = (char)tolower((unsigned char)s[i]);
    d[n] = 0;
    return d;
}

static char *dup_n(const char *s, size_t n)
{
    char *d = (char *)malloc(n + 1);
    memcpy(d, s, n); d[n] = 0;
    return d;
}

static int in_list_ci(const char *v, int n, const char **list)
{
    for (int i = 0; i < n; i++) if (strcasecmp(v, list[i]) == 0) return 1;
    return 0;
}

static int split_tabs(char *line, char **items, int max)
{
    int n = 0;
    char *p = line;
    while (n < max) {
        items[n++] = p;
        char *t = strchr(p, '\t');
        if (!t) break;
        *t = 0; p = t + 1;
    }
    return n;
}

/* value of `tag` in a GTF attribute column: `tag "value"; tag2 "value2";` */
static char *gtf_tag_value(const char *attrs, const char *tag)
{
    const size_t tl = strlen(tag);
    const char *p = attrs;
    while (*p) {
        while (*p == ' ' || *p == ';') p++;
        const char *q = p;
        while (*q && *q != ';') {
            if (*q == '"') { q++; while (*q && *q != '"') q++; if (*q) q++; } else q++;
        }
        if ((size_t)(q - p) > tl && strncmp(p, tag, tl) == 0 && p[tl] == ' ') {
            const char *v = p + tl + 1, *ve = q;
            while (ve > v && ve[-1] == ' ') ve--;
            if (ve - v >= 2 && v[0] == '"' && ve[-1] == '"') { v++; ve--; }
            return dup_n(v, (size_t)(ve - v));
        }
        p = q;
    }
    return dup_n("", 0);
}

/* bio/featio/gtf ReadFilteredFeatures (unpinned) and cmd/bed.go:52-121.  Returns -1 on error. */
static long read_features(const ora_subseq_opts *o, int is_bed, feature **out, FILE *err)
{
    const char *path = is_bed ? o->bed : o->gtf;
    gzFile fh = gzopen(path, "rb");
    if (!fh) { fprintf(err, "[ERRO] open %s: no such file or directory\n", path); return -1; }
    size_t cap = 1 << 16, nf = 0, fcap = 64;
    char *line = (char *)malloc(cap);
    feature *fs = (feature *)malloc(fcap * sizeof *fs);
    for (;;) {
        size_t n = 0;
        int c = 0, any = 0;
        while ((c = gzgetc(fh)) != -1) {
            any = 1;
            if (c == '\n') break;
            if (n + 2 > cap) { cap *= 2; line = (char *)realloc(line, cap); }
            line[n++] = (char)c;
        }
        if (!any) break;
        while (n && (line[n - 1] == '\r' || line[n - 1] == '\n')) n--;
        line[n] = 0;
        if (n == 0 || line[0] == '#') continue;
        char *it[16];
        feature f; memset(&f, 0, sizeof f);
        if (is_bed) {
            if ((n > 7 && strncmp(line, "browser", 7) == 0) || (n > 5 && strncmp(line, "track", 5) == 0)) continue;
            int k = split_tabs(line, it, 16);
            if (k < 3) continue;
            if (o->nchr > 0) {                                 /* bed.go:75-79: exact match */
                int hit = 0;
                for (int i = 0; i < o->nchr; i++) if (strcmp(it[0], o->chr[i]) == 0) hit = 1;
                if (!hit) continue;
            }
            char *e1, *e2;
            long long a = strtoll(it[1], &e1, 10), b = strtoll(it[2], &e2, 10);
            if (*e1 || e1 == it[1]) { fprintf(err, "[ERRO] %s: bad start: %s\n", it[0], it[1]); goto fail; }
            if (*e2 || e2 == it[2]) { fprintf(err, "[ERRO] %s: bad end: %s\n", it[0], it[2]); goto fail; }
            if (a >= b) { fprintf(err, "[ERRO] %s: start (%lld) must be <= end (%lld)\n", it[0], a, b); goto fail; }
            if (k >= 6 && strcmp(it[5], "+") != 0 && strcmp(it[5], "-") != 0) {
                fprintf(err, "[ERRO] bad strand: %s\n", it[5]); goto fail;
            }
            f.chr = dup_n(it[0], strlen(it[0])); f.type = dup_n("", 0);
            f.label = k >= 4 ? dup_n(it[3], strlen(it[3])) : dup_n("", 0);
            f.start = a + 1; f.end = b; f.strand = k >= 6 ? it[5][0] : 0;
        } else {
            int k = split_tabs(line, it, 9);
            if (k < 9) continue;
            if (o->nchr > 0 && !in_list_ci(it[0], o->nchr, o->chr)) continue;
            if (o->nfeature > 0 && !in_list_ci(it[2], o->nfeature, o->feature)) continue;
            f.chr = dup_n(it[0], strlen(it[0])); f.type = dup_lower(it[2], strlen(it[2]));
            f.start = atoll(it[3]); f.end = atoll(it[4]);
            f.strand = (it[6][0] == '+' || it[6][0] == '-') ? it[6][0] : 0;
            f.label = gtf_tag_value(it[8], o->gtf_tag);
        }
        if (nf == fcap) { fcap *= 2; fs = (feature *)realloc(fs, fcap * sizeof *fs); }
        fs[nf++] = f;
    }
    gzclose(fh); free(line);
    *out = fs;
    return (long)nf;
fail:
    gzclose(fh); free(line); free(fs);
    return -1;
}

static void write_fasta(FILE *out, const char *name, size_t name_len, const uint8_t *seq, size_t n, int width,
                        uint8_t **wrapbuf, size_t *wrapcap)
{
    size_t need = n + (width > 0 ? n / (size_t)width : 0) + 1;
    if (need > *wrapcap) { *wrapbuf = (uint8_t *)realloc(*wrapbuf, need); *wrapcap = need; }
    fputc('>', out); fwrite(name, 1, name_len, out); fputc('\n', out);
    size_t m = ora_wrap(seq, n, width, *wrapbuf);
    fwrite(*wrapbuf, 1, m, out); fputc('\n', out);
}

int ora_cmd_subseq(const ora_subseq_opts *o, int nfiles, const char **files, FILE *out, FILE *err)
{
    int64_t start = 0, end = 0;
    const int by_region = o->region && o->region[0];
    const int by_gtf = !by_region && o->gtf && o->gtf[0];
    const int by_bed = !by_region && !by_gtf && o->bed && o->bed[0];
    if (o->only_flank) {                                                         /* subseq.go:94-102 */
        if (o->up_stream > 0 && o->down_stream > 0) {
            fprintf(err, "[ERRO] when flag -f (--only-flank) given, only one of flags -u (--up-stream) and -d (--down-stream) is allowed\n");
            return 255;
        }
        if (o->up_stream == 0 && o->down_stream == 0) {
            fprintf(err, "[ERRO] when flag -f (--only-flank) given, one of flags -u (--up-stream) and -d (--down-stream) should be given\n");
            return 255;
        }
    }
    if (by_region) {                                                             /* subseq.go:103-138 */
        if (o->up_stream > 0 || o->down_stream > 0 || o->only_flank) {
            fprintf(err, "[ERRO] when flag -r (--region) given, any of flags -u (--up-stream), -d (--down-stream) and -f (--only-flank) is not allowed\n");
            return 255;
        }
        char *e1 = NULL, *e2 = NULL;
        long long a = strtoll(o->region, &e1, 10), b = 0;
        int ok = e1 != o->region && *e1 == ':';
        if (ok) { b = strtoll(e1 + 1, &e2, 10); ok = e2 != e1 + 1 && *e2 == 0; }
        if (!ok) {
            fprintf(err, "[ERRO] invalid region: %s. type \"seqkit subseq -h\" for more examples\n", o->region);
            return 255;
        }
        if (a == 0 || b == 0) { fprintf(err, "[ERRO] both start and end should not be 0\n"); return 255; }
        if (a < 0 && b > 0) { fprintf(err, "[ERRO] when start < 0, end should not > 0\n"); return 255; }
        start = a; end = b;
    }
    feature *fs = NULL; long nf = 0;
    char **types = NULL; long ntypes = 0;
    if (by_gtf || by_bed) {
        if (by_bed && o->nfeature > 0) {
            fprintf(err, "[ERRO] when given flag -b (--bed), flag -f (--feature) is not allowed\n");
            return 255;
        }
        fprintf(err, by_gtf ? "[INFO] read GTF file ...\n" : "[INFO] read BED file ...\n");
        nf = read_features(o, by_bed, &fs, err);
        if (nf < 0) return 255;
        fprintf(err, "[INFO] %ld %s features loaded\n", nf, by_gtf ? "GTF" : "BED");
        types = (char **)malloc((size_t)(nf + 1) * sizeof *types);
        for (long i = 0; i < nf; i++) {
            int seen = 0;
            for (long t = 0; t < ntypes; t++) if (strcmp(types[t], fs[i].type) == 0) seen = 1;
            if (!seen) types[ntypes++] = fs[i].type;
        }
    }
    uint8_t *wrapbuf = NULL, *sub = NULL; size_t wrapcap = 0, subcap = 0;
    char *name = NULL; size_t namecap = 0;
    int rcode = 0;
    for (int f = 0; f < nfiles && rcode == 0; f++) {
        ora_reader *r = ora_reader_open(files[f], -1, o->guess_len);
        if (!r) { fprintf(err, "[ERRO] open %s: no such file or directory\n", files[f]); rcode = 255; break; }
        ora_record rec; int rc;
        while ((rc = ora_reader_next(r, &rec)) == 1) {
            const int fastq = ora_reader_is_fastq(r);
            const int width = fastq ? 0 : o->line_width;                         /* subseq.go:316-318 */
            const int64_t len = (int64_t)rec.seq_len;
            if (by_region) {                                                     /* subseq.go:352-355 */
                int64_t s, e; size_t m = 0, a = 0;
                if (ora_sub_location(len, start, end, &s, &e) && e >= s) { a = (size_t)(s - 1); m = (size_t)(e - s + 1); }
                /* Record.FormatToWriter picks the layout by "has qualities": a FASTQ record whose
                 * subsequence is empty has none left and prints as FASTA (parity unpinned) */
                if (fastq && m > 0) {
                    fputc('@', out); fwrite(rec.name, 1, rec.name_len, out); fputc('\n', out);
                    fwrite(rec.seq + a, 1, m, out); fputs("\n+\n", out);
                    fwrite(rec.qual + a, 1, m, out); fputc('\n', out);
                } else {
                    write_fasta(out, (const char *)rec.name, rec.name_len, rec.seq + a, m, width, &wrapbuf, &wrapcap);
                }
                continue;
            }
            if (!by_gtf && !by_bed) continue;
            for (long t = 0; t < ntypes; t++) {
                for (long i = 0; i < nf; i++) {
                    const feature *ft = &fs[i];
                    if (strcmp(ft->type, types[t]) != 0) continue;
                    if (strlen(ft->chr) != rec.id_len || strncasecmp(ft->chr, (const char *)rec.id, rec.id_len) != 0) continue;
                    int64_t s = ft->start, e = ft->end;                          /* subseq.go:379-420, 471-512 */
                    const int minus = ft->strand == '-';
                    if (minus) {
                        if (o->only_flank) {
                            if (o->up_stream > 0) { s = ft->end + 1; e = ft->end + o->up_stream; }
                            else { s = ft->start - o->down_stream; e = ft->start - 1; }
                        } else { s = ft->start - o->down_stream; e = ft->end + o->up_stream; }
                    } else {
                        if (o->only_flank) {
                            if (o->up_stream > 0) { s = ft->start - o->up_stream; e = ft->start - 1; }
                            else { s = e + 1; e = e + o->down_stream; }
                        } else { s = ft->start - o->up_stream; e = ft->end + o->down_stream; }
                    }
                    if (s < 1) s = 1;
                    if (e > len) e = len;
                    int64_t a, b; size_t m = 0, a0 = 0;
                    if (ora_sub_location(len, s, e, &a, &b) && b >= a) { a0 = (size_t)(a - 1); m = (size_t)(b - a + 1); }
                    if (m + 1 > subcap) { subcap = m + 1; sub = (uint8_t *)realloc(sub, subcap); }
                    memcpy(sub, rec.seq + a0, m);
                    if (minus) ora_revcom_inplace(sub, NULL, m, ora_reader_alphabet(r));
                    char flank[96] = "";                                         /* subseq.go:434-452 */
                    if (o->up_stream > 0) {
                        if (o->only_flank) snprintf(flank, sizeof flank, "_usf:%d", o->up_stream);
                        else if (o->down_stream > 0) snprintf(flank, sizeof flank, "_us:%d_ds:%d", o->up_stream, o->down_stream);
                        else snprintf(flank, sizeof flank, "_us:%d", o->up_stream);
                    } else if (o->down_stream > 0) {
                        if (o->only_flank) snprintf(flank, sizeof flank, "_dsf:%d", o->down_stream);
                        else snprintf(flank, sizeof flank, "_ds:%d", o->down_stream);
                    }
                    size_t need = rec.id_len + strlen(ft->label) + 160;
                    if (need > namecap) { namecap = need; name = (char *)realloc(name, namecap); }
                    int nl = snprintf(name, namecap, "%.*s_%lld-%lld:%c%s %s", (int)rec.id_len, (const char *)rec.id,
                                      (long long)ft->start, (long long)ft->end, ft->strand ? ft->strand : '.', flank, ft->label);
                    write_fasta(out, name, (size_t)nl, sub, m, width, &wrapbuf, &wrapcap);
                }
            }
        }
        ora_reader_close(r);
        if (rc < 0) { fprintf(err, "[ERRO] error when parsing %s\n", files[f]); rcode = 255; }
    }
    for (long i = 0; i < nf; i++) { free(fs[i].chr); free(fs[i].type); free(fs[i].label); }
    free(fs); free(types); free(wrapbuf); free(sub); free(name);
    return rcode;
}

/* ------------------------------------------------------------------ convert */
const char *ora_quality_name(int enc)
{
    static const char *names[] = {"Unknown", "Sanger", "Solexa", "Illumina-1.3+", "Illumina-1.5+", "Illumina-1.8+"};
    return enc >= 0 && enc < ORA_NQUAL ? names[enc] : "Unknown";
}
int ora_quality_offset(int enc) { return (enc == ORA_Q_SANGER || enc == ORA_Q_ILLUMINA18) ? 33 : 64; }
int ora_quality_is_solexa(int enc) { return enc == ORA_Q_SOLEXA; }

int ora_quality_table(int from, int to, int force, uint8_t lut[256], uint8_t valid[256])
{
    const int off_from = ora_quality_offset(from), off_to = ora_quality_offset(to);
    const int sol_from = ora_quality_is_solexa(from), sol_to = ora_quality_is_solexa(to);
    const int identity = from == to || (!force && off_from == 33 && off_to == 33);   /* Sanger <-> Illumina-1.8+ */
    for (int c = 0; c < 256; c++) {
        lut[c] = (uint8_t)c; valid[c] = 1;
        if (identity) continue;
        double q = (double)(c - off_from);
        if (sol_from) {
            if (q < -5) { valid[c] = 0; continue; }
            if (!sol_to) q = 10.0 * log10(pow(10.0, q / 10.0) + 1.0);
        } else {
            if (q < 0) { valid[c] = 0; continue; }
            if (from == ORA_Q_ILLUMINA15 && q <= 2) q = 0;       /* 'B': read segment quality control indicator */
            if (sol_to) { q = q <= 0 ? -5.0 : 10.0 * log10(pow(10.0, q / 10.0) - 1.0); if (q < -5) q = -5; }
        }
        long r = (long)floor(q + 0.5);
        if (to == ORA_Q_ILLUMINA15 && r < 2) r = 2;
        if (force && to == ORA_Q_SANGER && r > 40) r = 40;   /* -f: "truncate scores > 40 to 40" */
        lut[c] = (uint8_t)(r + off_to);
    }
    return identity;
}

unsigned ora_guess_quality_encoding(const uint8_t *qual, size_t n, int n_most_common)
{
    if (n == 0) return 0;
    size_t cnt[256] = {0};
    int mn = 255, mx = 0;
    for (size_t i = 0; i < n; i++) { cnt[qual[i]]++; if (qual[i] < mn) mn = qual[i]; if (qual[i] > mx) mx = qual[i]; }
    if (mn >= 66 && mx <= 105 && cnt['B'] > 0) {
        int more = 0;                                   /* bytes strictly more frequent than 'B' */
        for (int c = 0; c < 256; c++) if (cnt[c] > cnt['B']) more++;
        if (more < n_most_common) return 1u << ORA_Q_ILLUMINA15;
    }
    unsigned set = 0;
    if (mn >= 33 && mx <= 73) set |= 1u << ORA_Q_SANGER;
    if (mn >= 59 && mx <= 104) set |= 1u << ORA_Q_SOLEXA;
    if (mn >= 64 && mx <= 104) set |= 1u << ORA_Q_ILLUMINA13;
    if (mn >= 66 && mx <= 105) set |= 1u << ORA_Q_ILLUMINA15;
    if (mn >= 33 && mx <= 74) set |= 1u << ORA_Q_ILLUMINA18;
    return set;
}

/* convert.go:100-152 / 185-238: the source encoding from the marks of the leading records.
 * Returns the encoding, or -1 after printing the reference's error. */
static int decide_encoding(const int marks[ORA_NQUAL], int n, int nrecords, double frac, FILE *err)
{
    int guessed[ORA_NQUAL], ng = 0;
    for (int e = 0; e < ORA_NQUAL; e++) if (marks[e] == n) guessed[ng++] = e;
    if ((double)(marks[ORA_Q_ILLUMINA15] / (nrecords + 1)) > frac) { guessed[0] = ORA_Q_ILLUMINA15; ng = 1; }  /* integer division, as written */
    fprintf(err, "[INFO] possible quality encodings: [");
    for (int i = 0; i < ng; i++) fprintf(err, "%s%s", i ? " " : "", ora_quality_name(guessed[i]));
    fprintf(err, "]\n");
    int from;
    if (ng == 0) { fprintf(err, "[ERRO] quality encoding not consistent\n"); return -1; }
    if (ng == 1) from = guessed[0];
    else {
        for (int i = 1; i < ng; i++)
            if (ora_quality_is_solexa(guessed[i]) != ora_quality_is_solexa(guessed[0]) ||
                ora_quality_offset(guessed[i]) != ora_quality_offset(guessed[0])) {
                fprintf(err, "[ERRO] fail to guess the source quality encoding, please specify it\n");
                return -1;
            }
        from = guessed[ng - 1];
        if (from == ORA_Q_ILLUMINA18) from = ORA_Q_SANGER;
    }
    fprintf(err, "[INFO] guessed quality encoding: %s\n", ora_quality_name(from));
    return from;
}

static int encodings_match(int a, int b, int force)          /* convert.go:313-326 */
{
    if (a == b) return 1;
    if (!force && ((a == ORA_Q_SANGER && b == ORA_Q_ILLUMINA18) || (a == ORA_Q_ILLUMINA18 && b == ORA_Q_SANGER))) return 1;
    return 0;
}

typedef struct kept { uint8_t *name, *seq, *qual; size_t nl, sl; } kept;

static int convert_write(FILE *out, FILE *err, int from, const uint8_t lut[256], const uint8_t valid[256],
                         const uint8_t *name, size_t nl, const uint8_t *seq, uint8_t *qual, size_t sl)
{
    for (size_t i = 0; i < sl; i++) {
        if (!valid[qual[i]]) {
            fflush(out);
            fprintf(err, "[ERRO] seq: invalid %s quality\n", ora_quality_is_solexa(from) ? "Solexa" : "Phred");
            return 255;
        }
    }
    for (size_t i = 0; i < sl; i++) qual[i] = lut[qual[i]];
    if (sl > 0) {                                          /* Record.FormatToWriter, line width 0 */
        fputc('@', out); fwrite(name, 1, nl, out); fputc('\n', out);
        fwrite(seq, 1, sl, out); fputs("\n+\n", out); fwrite(qual, 1, sl, out); fputc('\n', out);
    } else {
        fputc('>', out); fwrite(name, 1, nl, out); fputs("\n\n", out);
    }
    return 0;
}

int ora_cmd_convert(const ora_convert_opts *o, int nfiles, const char **files, FILE *out, FILE *err)
{
    int guessing = o->from <= 0, from = o->from, n = 0, rcode = 0;
    int marks[ORA_NQUAL] = {0};
    kept *recs = NULL;
    uint8_t lut[256], valid[256];
    if (guessing) recs = (kept *)calloc((size_t)o->nrecords + 1, sizeof *recs);
    else {
        fprintf(err, "[INFO] converting %s -> %s\n", ora_quality_name(from), ora_quality_name(o->to));
        ora_quality_table(from, o->to, o->force, lut, valid);
    }
    for (int f = 0; f < nfiles && rcode == 0; f++) {
        ora_reader *r = ora_reader_open(files[f], -1, o->guess_len);
        if (!r) { fprintf(err, "[ERRO] open %s: no such file or directory\n", files[f]); rcode = 255; break; }
        int once = 1, rc, stop = 0;
        ora_record rec;
        while (!stop && rcode == 0) {
            rc = ora_reader_next(r, &rec);
            if (rc != 1) {
                if (rc < 0) { fprintf(err, "[ERRO] error when parsing %s\n", files[f]); rcode = 255; break; }
                if (guessing && n < o->nrecords) {         /* convert.go:99-165: fewer records than -n */
                    from = decide_encoding(marks, n, o->nrecords, o->illumina15_frac, err);
                    if (from < 0) { rcode = 255; break; }
                    fprintf(err, "[INFO] converting %s -> %s\n", ora_quality_name(from), ora_quality_name(o->to));
                    if (once && encodings_match(from, o->to, o->force)) { once = 0; fprintf(err, "[WARN] source and target quality encoding match.\n"); }
                    if (!o->dry_run) {
                        ora_quality_table(from, o->to, o->force, lut, valid);
                        for (int i = 0; i < n && rcode == 0; i++)
                            rcode = convert_write(out, err, from, lut, valid, recs[i].name, recs[i].nl, recs[i].seq, recs[i].qual, recs[i].sl);
                    }
                }
                break;
            }
            if (!ora_reader_is_fastq(r)) { fprintf(err, "[ERRO] this command only works for FASTQ format\n"); rcode = 255; break; }
            if (guessing) {
                if (n <= o->nrecords) {                     /* keep a copy (record.Clone()) */
                    kept *k = &recs[n];
                    k->nl = rec.name_len; k->sl = rec.seq_len;
                    k->name = (uint8_t *)malloc(k->nl + 1); memcpy(k->name, rec.name, k->nl);
                    k->seq = (uint8_t *)malloc(k->sl + 1); memcpy(k->seq, rec.seq, k->sl);
                    k->qual = (uint8_t *)malloc(k->sl + 1); memcpy(k->qual, rec.qual, rec.qual_len < k->sl ? rec.qual_len : k->sl);
                }
                if (n < o->nrecords) {
                    unsigned set = ora_guess_quality_encoding(rec.qual, rec.qual_len, o->n_most_common);
                    for (int e = 0; e < ORA_NQUAL; e++) if (set & (1u << e)) marks[e]++;
                } else {                                    /* n == nrecords: convert.go:184-250 */
                    from = decide_encoding(marks, n, o->nrecords, o->illumina15_frac, err);
                    if (from < 0) { rcode = 255; break; }
                    fprintf(err, "[INFO] converting %s -> %s\n", ora_quality_name(from), ora_quality_name(o->to));
                    if (once && encodings_match(from, o->to, o->force)) { once = 0; fprintf(err, "[WARN] source and target quality encoding match.\n"); }
                    if (o->dry_run) { stop = 1; break; }
                    ora_quality_table(from, o->to, o->force, lut, valid);
                    for (int i = 0; i <= n && rcode == 0; i++)
                        rcode = convert_write(out, err, from, lut, valid, recs[i].name, recs[i].nl, recs[i].seq, recs[i].qual, recs[i].sl);
                    guessing = 0;
                }
                n++;
                continue;
            }
            if (once && encodings_match(from, o->to, o->force)) { once = 0; fprintf(err, "[WARN] source and target quality encoding match.\n"); }
            if (o->dry_run) break;
            rcode = convert_write(out, err, from, lut, valid, rec.name, rec.name_len, rec.seq, rec.qual, rec.seq_len);
        }
        ora_reader_close(r);
    }
    if (recs) {
        for (int i = 0; i <= o->nrecords; i++) { free(recs[i].name); free(recs[i].seq); free(recs[i].qual); }
        free(recs);
    }
    return rcode;
}
