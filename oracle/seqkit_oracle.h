/*
 * seqkit_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of SeqKit's per-record hot path (reverse, complement,
 * base content, gap counting, stats reduction) and of the exact text formatting
 * the three commands `seq`, `fx2tab`, `stats` wrap around it.  It exists so the
 * parity tests have something independent to compare the HIP path against.
 * Nothing in the product (seqkit_amd/, include/) may include, link or execute it;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Where the algorithm comes from: the reference tree (sauloalgolang/seqkit,
 * SeqKit 0.7.2s) only holds the CALL SITES -- seqkit/cmd/seq.go:186-195,
 * seqkit/cmd/fx2tab.go:124-148, seqkit/cmd/stat.go:96-160.  The arithmetic lives in
 * Go modules that are neither vendored nor version-pinned in that tree:
 *   github.com/shenwei356/bio   (seq, seqio/fastx)     -- no go.mod, unpinned
 *   github.com/shenwei356/util  (byteutil, math)       -- no go.mod, unpinned
 *   github.com/dustin/go-humanize, github.com/tatsushid/go-prettytable -- unpinned
 * Their published behaviour is restated here and pinned against every known
 * answer the reference's own docs and tests hold for this path (see
 * tests/test_oracle_golden.py, tests/golden/expected/...).  Behaviour that no
 * in-tree vector covers is marked "parity unpinned" at the function.
 */
#ifndef SEQKIT_ORACLE_H
#define SEQKIT_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Alphabets the fastx reader can hand to a command (bio/seq alphabet.go). */
enum {
    ORA_ALPHABET_DNA = 0,
    ORA_ALPHABET_DNAredundant = 1,
    ORA_ALPHABET_RNA = 2,
    ORA_ALPHABET_RNAredundant = 3,
    ORA_ALPHABET_Protein = 4,
    ORA_ALPHABET_Unlimit = 5
};

const char *ora_alphabet_name(int alphabet);
/* membership table of letters+gap+ambiguous letters of the alphabet */
void ora_alphabet_members(int alphabet, uint8_t is_member[256]);
/* PairLetter table; bytes outside the alphabet map to themselves (parity unpinned) */
void ora_pair_table(int alphabet, uint8_t lut[256]);
/* GuessAlphabetLessConservatively on the first `threshold` bytes (0 = all) */
int ora_guess_alphabet(const uint8_t *seq, size_t n, size_t threshold);

/* seq.go:188  (*Seq).ReverseInplace: reverses Seq and, when present, Qual */
void ora_reverse_inplace(uint8_t *seq, uint8_t *qual, size_t n);
/* seq.go:194  (*Seq).ComplementInplace */
void ora_complement_inplace(uint8_t *seq, size_t n, int alphabet);
/* locate.go:154 / subseq.go:399  RevCom = reverse then complement */
void ora_revcom_inplace(uint8_t *seq, uint8_t *qual, size_t n, int alphabet);

/* fx2tab.go:128,129,141  (*Seq).BaseContent(list) */
double ora_base_content(const uint8_t *seq, size_t n, const char *list);
/* integer numerator of BaseContent (what the GPU path produces) */
uint64_t ora_base_count(const uint8_t *seq, size_t n, const char *list);
/* stat.go:125  byteutil.CountBytes */
uint64_t ora_count_bytes(const uint8_t *seq, size_t n, const uint8_t *letters, size_t nletters);

/* seq.go:244  byteutil.WrapByteSlice; `out` needs n + n/width + 1 bytes; returns length */
size_t ora_wrap(const uint8_t *s, size_t n, int width, uint8_t *out);

/* stat.go:158  util/math.Round(f, n) */
double ora_round(double f, int n);
/* stat.go:297-326 on an ascending-sorted array */
int64_t ora_median(const int64_t *sorted, size_t l);
void ora_quartile(const int64_t *sorted, size_t l, int64_t *q1, int64_t *q2, int64_t *q3);
/* stat.go:136-145 */
void ora_n50(const int64_t *sorted, size_t num, int64_t len_sum, int64_t *n50, int64_t *l50);

/* go-humanize Comma / Commaf; buf >= 64 bytes */
void ora_comma(int64_t v, char *buf);
void ora_commaf(double v, char *buf);
/* Go's fmt "%.<prec>f" incl. NaN/Inf spelling; buf >= 400 bytes */
void ora_fmt_fixed(double v, int prec, char *buf);
/* fx2tab.go:127-143 for records given by their counts: "\t<len>\t<GC>\t<GC-skew>\t<content>\n" per record */
size_t ora_tab_number_rows(const uint64_t *ng, const uint64_t *nc, const uint64_t *len, size_t n, char *out);

/* ---- the same per-record functions looped over a CSR batch (seq bytes back to back,
 * offs[nrec+1]); what the GPU entry points are compared with ---- */
/* lut NULL = no complement; qual NULL = no quality plane; in place */
void ora_batch_transform(uint8_t *seq, uint8_t *qual, const uint64_t *offs, uint64_t nrec,
                         const uint8_t *lut, int do_reverse);
/* counts[r*nclass+c] = bytes of record r whose class_of[] is c (0xFF = none) */
void ora_batch_count(const uint8_t *seq, const uint64_t *offs, uint64_t nrec,
                     const uint8_t class_of[256], uint32_t nclass, uint64_t *counts);
/* out = {num, len_sum, gap_sum, len_min, len_max} with the stat.go:91 initial values */
void ora_batch_stats(const uint8_t *seq, const uint64_t *offs, uint64_t nrec,
                     const uint8_t *gap_letters, size_t ngap, int64_t out[5]);

/* ---- FASTA/Q reader (bio/seqio/fastx restated; plain or gzip input, "-" = stdin) ---- */
typedef struct ora_record {
    uint8_t *name; size_t name_len;   /* head line without '>'/'@' */
    size_t id_len;                    /* ID = id[0..id_len): capture group 1 of the ID expression in the name */
    const uint8_t *id;                /* (the default expression makes it a prefix of the name) */
    uint8_t *seq;  size_t seq_len;
    uint8_t *qual; size_t qual_len;   /* 0 for FASTA */
} ora_record;

typedef struct ora_reader ora_reader;
/* --id-regexp / --id-ncbi (seqkit/cmd/root.go:68-69, helper.go:125-135; handed to fastx.NewReader at
 * seq.go:115, fx2tab.go:96, stat.go:87): the ID of every record read afterwards is capture group 1 of the
 * expression matched against the head line, the whole head line where it does not match.  NULL or the
 * default `^([^\s]+)\s?` restores the built-in rule.  The Go expression is handed to POSIX regcomp
 * after its Perl classes (\s \S \d \D \w \W, also inside brackets) have been spelled out; pinned by
 * doc/docs/usage.md:345-348, otherwise parity unpinned.  Returns 0, or -1 when the expression does
 * not compile or has no group. */
int ora_set_id_regexp(const char *go_regexp);
ora_reader *ora_reader_open(const char *path, int forced_alphabet /* -1 = guess */,
                            size_t guess_threshold);
/* 1 = record returned (valid until next call), 0 = EOF, <0 = parse error */
int ora_reader_next(ora_reader *r, ora_record *rec);
int ora_reader_is_fastq(const ora_reader *r);
int ora_reader_alphabet(const ora_reader *r);   /* -1 until the first record was read */
void ora_reader_close(ora_reader *r);

/* seq.NewSeq's check when seq.ValidateSeq is set (bio/seq, unpinned; reached through the
 * reader at seqkit/cmd/seq.go:115-127 with the settings of seq.go:84-93): the first `limit`
 * bytes of a record (all of them when limit == 0) must be letters, gap or ambiguous letters
 * of the alphabet.  Returns the index of the first byte that is not, or -1.  Unlimit (and an
 * unknown alphabet) accepts everything.  Pinned by doc/docs/usage.md:322-324 ("invalid
 * DNAredundant letter: e") and tests/test.sh:31-32. */
long ora_validate_seq(const uint8_t *seq, size_t n, int alphabet, size_t limit);

/* ---- whole-command restatements; write exactly the bytes the Go command writes ---- */
typedef struct ora_seq_opts {
    int reverse, complement;  /* -r -p */
    int only_name, only_seq, only_qual, only_id; /* -n -s -q -i */
    int line_width;           /* -w (default 60) */
    int min_len, max_len;     /* -m -M (-1 = off) */
    int alphabet;             /* -t: -1 auto */
    size_t guess_len;         /* --alphabet-guess-seq-length (default 10000) */
    int validate;             /* -v, or implied by -t (seq.go:84-93) */
    size_t validate_len;      /* -V: bytes of each record that are checked (default 10000, 0 = all) */
    int remove_gaps;          /* -g  (seq.go:196-198) */
    const char *gap_letters;  /* -G  (default "- .", seq.go:298; NULL = default) */
    int lower_case, upper_case;  /* -l -u (seq.go:236-240) */
    int dna2rna, rna2dna;     /* --dna2rna --rna2dna (seq.go:201-235) */
    int quiet;                /* --quiet: silences the complement warning for Protein only (seq.go:191 reads
                               * `!config.Quiet && alphabet == Protein || alphabet == Unlimit`) */
} ora_seq_opts;

/* (*Seq).RemoveGaps(letters) (bio/seq, unpinned module; call site seqkit/cmd/seq.go:197): the
 * bytes of seq that are in `letters` are dropped, the quality bytes at the same positions with
 * them (qual may be NULL).  In place; returns the new length.  Pinned for FASTA by the worked
 * example of doc/docs/usage.md:370-372; what happens to the qualities is parity unpinned. */
size_t ora_remove_gaps(uint8_t *seq, uint8_t *qual, size_t len, const char *letters);
int ora_cmd_seq(const ora_seq_opts *o, int nfiles, const char **files, FILE *out, FILE *err);

typedef struct ora_fx2tab_opts {
    int only_id, length, gc, gc_skew, only_name, header_line, alphabet_col; /* -i -l -g -G -n -H -a */
    int nbase; const char **base;  /* -B values in order */
    size_t guess_len;
} ora_fx2tab_opts;
int ora_cmd_fx2tab(const ora_fx2tab_opts *o, int nfiles, const char **files, FILE *out, FILE *err);

typedef struct ora_stats_opts {
    int all, tabular;           /* -a -T */
    const char *gap_letters;    /* -G (default "- .") */
    size_t guess_len;
    int alphabet_plus1;         /* -t: alphabet id + 1 handed to the reader (stat.go:87), 0 = guess */
} ora_stats_opts;
int ora_cmd_stats(const ora_stats_opts *o, int nfiles, const char **files, FILE *out, FILE *err);

/* ---- subseq ---- */
/* bio/seq SubLocation (unpinned module; called at seqkit/cmd/subseq.go:231 and, through
 * (*Seq).SubSeq, at :353,399,419,491,511): turns the 1-based, inclusive, possibly negative
 * (start, end) of a region into a 1-based inclusive range of a sequence of `length` bytes.
 * Returns 0 when the region selects nothing (SubSeq then yields an empty sequence).  Pinned by
 * the region table of doc/docs/usage.md:408-419 and tests/test.sh:109-137; the result for
 * end == 0 (only reachable from the feature modes) is parity unpinned. */
int ora_sub_location(int64_t length, int64_t start, int64_t end, int64_t *s, int64_t *e);

/* One output record of subseq over a CSR batch: SubSeq(start, end) of record `record`, then
 * RevComInplace when `minus` (subseq.go:399,491).  `ora_batch_subseq` writes the slices back
 * to back into out_seq (and out_qual when qual != NULL; qualities are reversed, never
 * complemented) and their offsets into out_offs[n+1]; out planes need the sum of the slice
 * lengths, at most sum over slices of the source record's length.  lut = pair table of the
 * alphabet (NULL: reverse only). */
typedef struct ora_slice { uint64_t record; int64_t start, end; uint32_t minus, pad; } ora_slice;
void ora_batch_subseq(const uint8_t *seq, const uint8_t *qual, const uint64_t *offs, uint64_t nrec,
                      const ora_slice *slices, uint64_t n, const uint8_t *lut,
                      uint8_t *out_seq, uint8_t *out_qual, uint64_t *out_offs);

/* seqkit/cmd/subseq.go:53-347 on the streaming path (:301-343; the FASTA-index shortcut of
 * :202-298 gives the same records in map order and is not restated).  Feature types are taken
 * in order of first appearance in the GTF file (the reference iterates a Go map there). */
typedef struct ora_subseq_opts {
    const char *region;        /* -r */
    const char *gtf, *bed;     /* --gtf --bed */
    const char *gtf_tag;       /* --gtf-tag (default gene_id) */
    int nchr; const char **chr;        /* --chr */
    int nfeature; const char **feature; /* --feature */
    int up_stream, down_stream, only_flank;   /* -u -d -f */
    int line_width;
    size_t guess_len;
} ora_subseq_opts;
int ora_cmd_subseq(const ora_subseq_opts *o, int nfiles, const char **files, FILE *out, FILE *err);

/* ---- convert (FASTQ quality re-encoding) ---- */
/* bio/seq quality encodings (unpinned module; used at seqkit/cmd/convert.go:54-63,161,245,270). */
enum { ORA_Q_UNKNOWN = 0, ORA_Q_SANGER = 1, ORA_Q_SOLEXA = 2, ORA_Q_ILLUMINA13 = 3, ORA_Q_ILLUMINA15 = 4,
       ORA_Q_ILLUMINA18 = 5, ORA_NQUAL = 6 };
const char *ora_quality_name(int enc);
int ora_quality_offset(int enc);          /* 33 or 64 */
int ora_quality_is_solexa(int enc);
/* seq.QualityConvert as a byte table: lut[c] = converted byte, valid[c] = 0 where QualityConvert
 * fails (score below the encoding's minimum: "seq: invalid Phred quality" / "seq: invalid Solexa
 * quality").  Returns 1 when the conversion is the identity on every byte (from == to, or
 * Sanger <-> Illumina-1.8+ without force).  Pinned by doc/docs/usage.md:878-956 (offset shifts
 * 33 <-> 64, -f truncating 41 to 40, Illumina-1.5+ 'B' -> 0, Solexa rejecting low bytes); the
 * Solexa <-> Phred score formula (Cock et al. 2010, rounded to nearest) and scores below 2 going
 * to Illumina-1.5+ are parity unpinned. */
int ora_quality_table(int from, int to, int force, uint8_t lut[256], uint8_t valid[256]);
/* seq.GuessQualityEncoding: candidate encodings of one quality string as a bit set (1 << enc),
 * by the byte range [min, max] against Sanger 33-73, Solexa 59-104, Illumina-1.3+ 64-104,
 * Illumina-1.5+ 66-105, Illumina-1.8+ 33-74; Illumina-1.5+ alone when it fits and 'B' is among
 * the `n_most_common` most frequent bytes.  Pinned only by the three documented guesses
 * (usage.md:879, 924, 950); otherwise parity unpinned. */
unsigned ora_guess_quality_encoding(const uint8_t *qual, size_t n, int n_most_common);

typedef struct ora_convert_opts {
    int from, to;               /* ORA_Q_*; from UNKNOWN = guess */
    int dry_run, force;         /* -d -f */
    int nrecords;               /* -n (default 1000) */
    int n_most_common;          /* -N (default 4) */
    double illumina15_frac;     /* -F (default 0.1) */
    size_t guess_len;
} ora_convert_opts;
/* seqkit/cmd/convert.go:42-275 */
int ora_cmd_convert(const ora_convert_opts *o, int nfiles, const char **files, FILE *out, FILE *err);

#ifdef __cplusplus
}
#endif
#endif
