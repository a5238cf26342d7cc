/*
 * seqkit_hip.h -- C ABI of libseqkit_hip.so: SeqKit's per-record hot path
 * (reverse / complement / base counting / stats totals) on AMD MI355X (gfx950).
 *
 * The reference (SeqKit 0.7.2s, Go) has no FFI on this path: its commands call
 * methods of github.com/shenwei356/bio/seq and shenwei356/util/byteutil once per
 * record from a serial loop.  This ABI replaces those per-record calls with
 * per-BATCH calls.  A batch is a CSR pair: all sequence bytes of the batch back to
 * back in `seq` (no separators, no line breaks), and `offsets[nrec+1]` (uint64,
 * offsets[0] = 0, non-decreasing) such that record r is seq[offsets[r]..offsets[r+1]).
 * FASTQ qualities live in a second byte buffer that shares the same offsets.
 *
 * Conventions
 *  - every function returns 0 on success or a negative SKH_E* code; a message for
 *    the calling thread is available from skh_last_error();
 *  - no function throws, aborts, or calls exit(); nothing is printed;
 *  - tables (complement LUT, class table) are host memory, read during the call;
 *  - one context drives one GPU; use one process (or one context) per GPU (see "several GPUs");
 *  - the *_dev entry points take DEVICE pointers and enqueue on `stream`
 *    (a hipStream_t passed as void*, NULL = the context's own stream) without
 *    synchronising; the batch entry points are synchronous.
 *  - device buffers handed to *_dev functions must be 16-byte aligned and readable
 *    (inputs) / writable (outputs) up to skh_padded_size(total_bytes).
 */
#ifndef SEQKIT_HIP_H
#define SEQKIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKH_OK           0
#define SKH_EINVAL      -1   /* bad argument (NULL pointer, nclass out of range, ...) */
#define SKH_ENODEV      -2   /* no usable gfx950 device / device ordinal out of range */
#define SKH_ENOMEM      -3   /* host or device allocation failed */
#define SKH_EHIP        -4   /* a HIP runtime call failed; see skh_last_error() */
#define SKH_ECAPACITY   -5   /* batch larger than its allocated capacity */

#define SKH_MAX_CLASSES  8   /* classes per counting pass */
#define SKH_NO_CLASS  0xFF   /* class_of[] value for "do not count this byte" */

typedef struct skh_ctx skh_ctx;
typedef struct skh_batch skh_batch;

/* Per-file totals of `seqkit stats` (replaces the accumulators of
 * seqkit/cmd/stat.go:78,91,106-125).  len_min is INT64_MAX and len_max 0 for
 * an empty batch, exactly like the Go initial values (stat.go:91). */
typedef struct skh_stats_partial {
    int64_t num;       /* records                      (stat.go:106) */
    int64_t len_sum;   /* sum of lengths               (stat.go:118) */
    int64_t gap_sum;   /* bytes that are gap letters   (stat.go:125) */
    int64_t len_min;   /*                              (stat.go:119-121) */
    int64_t len_max;   /*                              (stat.go:122-124) */
} skh_stats_partial;

/* Message describing the calling thread's most recent failure ("" if none). */
const char *skh_last_error(void);
/* Library version string, e.g. "seqkit-hip 0.1 gfx950". */
const char *skh_version(void);
/* Bytes a device/pinned sequence buffer must span for `total_bytes` of sequence. */
uint64_t skh_padded_size(uint64_t total_bytes);

/* Creates a context on HIP device `device` (ordinal as seen by this process). */
int skh_init(int device, skh_ctx **out);
void skh_destroy(skh_ctx *ctx);
/* Blocks until everything enqueued on the context's own stream has finished. */
int skh_sync(skh_ctx *ctx);

/* ---- table builders (host only; replace bio/seq alphabet lookups) ------------------
 * skh_pair_table: the PairLetter table ComplementInplace applies
 * (call site seqkit/cmd/seq.go:194; alphabet chosen at seqkit/cmd/helper.go:138-156
 * or guessed by the reader).  alphabet: 0 DNA, 1 DNAredundant, 2 RNA, 3 RNAredundant,
 * 4 Protein, 5 Unlimit.  Bytes outside the alphabet map to themselves. */
int skh_pair_table(int alphabet, uint8_t lut[256]);
/* skh_class_table: class table for (*Seq).BaseContent(list) letters
 * (seqkit/cmd/fx2tab.go:128-141): for each i < nlists, every letter of lists[i] in
 * either case maps to class i.  Fails with SKH_EINVAL if two lists share a letter
 * (the caller must then use separate passes) or nlists > SKH_MAX_CLASSES. */
int skh_class_table(const char *const *lists, uint32_t nlists, uint8_t class_of[256]);

/* ---- device-resident entry points -------------------------------------------------- */

/* Reverse and/or complement every record of a batch, out of place.
 * Replaces (*Seq).ReverseInplace / ComplementInplace / RevCom
 * (call sites seqkit/cmd/seq.go:187-195, locate.go:154, subseq.go:399,491).
 *   d_out[offsets[r] + L-1-j] = lut[d_in[offsets[r] + j]]   (do_reverse)
 *   d_out[offsets[r] + j]     = lut[d_in[offsets[r] + j]]   (!do_reverse)
 * lut == NULL means identity (quality strings: reverse only).  d_in and d_out must
 * not overlap.  Bytes of d_out at and beyond total_bytes are not written. */
int skh_revcomp_dev(skh_ctx *ctx, const uint8_t *d_in, uint8_t *d_out,
                    const uint64_t *d_offsets, uint64_t nrec, uint64_t total_bytes,
                    const uint8_t *lut, int do_reverse, void *stream);

/* Which kernel serves the records up to the long-record threshold for a batch of this shape on this context
 * (the selection the entry points make from the number of records, the bytes and the context's settings): its
 * name as the profiler shows it, e.g. "skh::k_reverse_flat<true>".  op: SKH_OP_REVCOMP (out of place) and
 * SKH_OP_REVCOMP_INPLACE, with `variant` != 0 when a table is applied (complement); SKH_OP_COUNT, with
 * `variant` = the number of classes.  For tools that label measurements; returns SKH_EINVAL for an unknown op
 * or a buffer that is too small.  No counterpart in the reference. */
#define SKH_OP_REVCOMP         1
#define SKH_OP_REVCOMP_INPLACE 2
#define SKH_OP_COUNT           3
int skh_kernel_for(skh_ctx *ctx, int op, uint64_t nrec, uint64_t total_bytes, int variant,
                   char *name, uint64_t name_bytes);

/* FASTQ form of the above in one pass: bases are reversed (if do_reverse) and mapped
 * through lut, qualities are reversed (if do_reverse) and otherwise copied
 * (seq.go:188 reverses Seq and Qual together; seq.go:194 touches Seq only). */
int skh_revcomp_fastq_dev(skh_ctx *ctx, const uint8_t *d_seq_in, uint8_t *d_seq_out,
                          const uint8_t *d_qual_in, uint8_t *d_qual_out,
                          const uint64_t *d_offsets, uint64_t nrec, uint64_t total_bytes,
                          const uint8_t *lut, int do_reverse, void *stream);

/* The same with ONE plane per kind, overwritten: what the reference's methods do to a record
 * ((*Seq).ReverseInplace reverses Seq.Seq and Seq.Qual where they lie, ComplementInplace maps
 * Seq.Seq where it lies; call sites seqkit/cmd/seq.go:188,194).  d_qual may be NULL; it is reversed
 * with the bases (do_reverse) and otherwise left alone.  Half the device memory of the
 * out-of-place form, the same traffic, and the same speed on fixed-length reads; ragged short
 * reads run about a tenth slower than out of place (the ends of a wavefront's records leave
 * as misaligned stores). */
int skh_revcomp_inplace_dev(skh_ctx *ctx, uint8_t *d_seq, uint8_t *d_qual,
                            const uint64_t *d_offsets, uint64_t nrec, uint64_t total_bytes,
                            const uint8_t *lut, int do_reverse, void *stream);

/* Per-record byte-class histogram.  Replaces the bytes.Count scans inside
 * (*Seq).BaseContent (call sites seqkit/cmd/fx2tab.go:128,129,141) and
 * byteutil.CountBytes (seqkit/cmd/stat.go:125) by one pass:
 *   d_counts[r*nclass + c] = #{ j : class_of[seq[offsets[r]+j]] == c },  c < nclass.
 * class_of[b] is a class index < nclass or SKH_NO_CLASS.  1 <= nclass <= SKH_MAX_CLASSES. */
int skh_count_dev(skh_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_offsets,
                  uint64_t nrec, uint64_t total_bytes, const uint8_t class_of[256],
                  uint32_t nclass, uint64_t *d_counts, void *stream);

/* Whole-batch totals for `stats` (accumulators of seqkit/cmd/stat.go:106-125):
 * writes one skh_stats_partial to DEVICE memory d_out. */
int skh_stats_dev(skh_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_offsets,
                  uint64_t nrec, uint64_t total_bytes, const uint8_t is_gap[256],
                  skh_stats_partial *d_out, void *stream);

/* skh_stats_dev for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc): same arguments, same result, but the kernel's
 * scratch (512 partial gap sums, 4 KiB) lies in a static array inside the library's device image
 * instead of memory the library allocated, and nothing on this path allocates or frees.  It has no
 * counterpart in the reference.  Limits: total_bytes <= SKH_STATS_STATIC_MAX_BYTES and
 * nrec <= SKH_STATS_STATIC_MAX_RECORDS, else SKH_ECAPACITY.  Every call on a GPU shares that GPU's
 * one array (cleared by each call on its stream): calls that overlap on one GPU (different streams,
 * or different contexts on the same device) race and must be serialised by the caller. */
#define SKH_STATS_STATIC_MAX_BYTES   (1ull << 40)
#define SKH_STATS_STATIC_MAX_RECORDS (1ull << 36)
int skh_stats_static_dev(skh_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_offsets,
                         uint64_t nrec, uint64_t total_bytes, const uint8_t is_gap[256],
                         skh_stats_partial *d_out, void *stream);

/* Sequence validation: seq.NewSeq's check with seq.ValidateSeq set (bio/seq; settings at
 * seqkit/cmd/seq.go:84-93, reached through the reader at seq.go:115-127).  The first `limit`
 * bytes of every record (all of them when limit == 0) must be members of the alphabet:
 * valid[b] != 0.  skh_valid_table builds `valid` for an alphabet id (letters + gap +
 * ambiguous letters; Unlimit accepts everything).  The result names the FIRST offending
 * record of the batch and its first offending byte; the caller prints
 * "error when parsing seq: <id> (invalid <Alphabet> letter: <c>)". */
typedef struct skh_validate_result {
    uint64_t record;     /* index of the first record with an invalid byte   */
    uint64_t offset;     /* position of that byte inside the record          */
    uint32_t letter;     /* the byte                                         */
    uint32_t found;      /* 0: every record is valid (other fields are 0)    */
} skh_validate_result;
int skh_valid_table(int alphabet, uint8_t valid[256]);
int skh_validate_dev(skh_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_offsets, uint64_t nrec,
                     uint64_t total_bytes, const uint8_t valid[256], uint64_t limit,
                     skh_validate_result *d_out, void *stream);

/* Gap removal: (*Seq).RemoveGaps(letters) of bio/seq (call site seqkit/cmd/seq.go:196-198, flags
 * -g / -G at seq.go:297-298).  Every byte b of a record with is_gap[b] != 0 is dropped, and the
 * quality byte at the same position with it (d_qual may be NULL).  The output is a new batch:
 * d_out_seq / d_out_qual (room for total_bytes, batch layout rules), d_out_offsets[nrec + 1];
 * *d_out_bytes (DEVICE memory) = bytes kept.  d_workspace: 256-byte aligned scratch of
 * skh_remove_gaps_workspace_size(nrec, total_bytes) bytes. */
uint64_t skh_remove_gaps_workspace_size(uint64_t nrec, uint64_t total_bytes);
int skh_remove_gaps_dev(skh_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual, const uint64_t *d_offsets,
                        uint64_t nrec, uint64_t total_bytes, const uint8_t is_gap[256], uint8_t *d_out_seq,
                        uint8_t *d_out_qual, uint64_t *d_out_offsets, void *d_workspace, uint64_t workspace_bytes,
                        uint64_t *d_out_bytes, void *stream);

/* FASTQ quality re-encoding (`seqkit convert`): seq.QualityConvert(from, to, qual, force)
 * (bio/seq; call sites seqkit/cmd/convert.go:161,245,270) is a per-byte function, so it is
 * handed to the GPU as a 256-entry table: skh_quality_table fills lut (converted byte) and
 * valid (0 where QualityConvert fails: a score below the source encoding's minimum, reported
 * by the caller as "seq: invalid Phred quality" / "seq: invalid Solexa quality"); *identity
 * (may be NULL) is set when nothing changes (from == to, or Sanger <-> Illumina-1.8+ without
 * force, convert.go:313-326).  On the device: skh_validate_dev(d_qual, ..., valid, 0, ...)
 * finds the first record QualityConvert would reject, skh_revcomp_dev(d_qual, d_qual_out, ...,
 * lut, 0, ...) converts the plane.  Offsets: Sanger and Illumina-1.8+ 33, the others 64;
 * Illumina-1.5+ scores <= 2 ('B') read as 0; force cuts scores above 40 going to Sanger. */
#define SKH_Q_SANGER     1
#define SKH_Q_SOLEXA     2
#define SKH_Q_ILLUMINA13 3
#define SKH_Q_ILLUMINA15 4
#define SKH_Q_ILLUMINA18 5
int skh_quality_table(int from, int to, int force, uint8_t lut[256], uint8_t valid[256], int *identity);

/* Subsequences: what `seqkit subseq` does to a record (seqkit/cmd/subseq.go:352-355 for
 * -r, :378-420 for GTF features, :470-512 for BED features): record.Seq.SubSeq(start, end),
 * followed by RevComInplace() for a feature on the minus strand (subseq.go:399,491).  A slice
 * names a record of the batch and SubSeq's arguments: 1-based, inclusive, negative positions
 * counting from the end, resolved with seq.SubLocation's rules (region table of
 * doc/docs/usage.md:408-419).  The output is a new batch with one record per slice, in slice
 * order: d_out_seq (and d_out_qual when d_qual is given; qualities are reversed with their
 * bases, never complemented) packed back to back, d_out_offsets[n+1] its offsets.
 * d_slices == NULL is the region mode: slice i is record i (n must equal nrec) with the
 * uniform start / end / minus arguments.  lut: complement table for minus slices (NULL:
 * reverse only).  The outcome goes to DEVICE memory d_result: out_bytes = total length of
 * the slices; status SKH_SUBSEQ_ECAPACITY when that exceeds out_capacity (nothing valid was
 * written; retry with larger planes), SKH_SUBSEQ_ERECORD when slice bad_slice names a record
 * >= nrec.  Output planes follow the batch layout rule (16-byte aligned, allocated to
 * skh_padded_size(out_capacity)); d_workspace is 256-byte aligned scratch of
 * skh_subseq_workspace_size(n) bytes. */
#define SKH_SUBSEQ_OK        0
#define SKH_SUBSEQ_ERECORD   1
#define SKH_SUBSEQ_ECAPACITY 2
typedef struct skh_slice {
    uint64_t record;       /* source record in the batch               */
    int64_t  start, end;   /* SubSeq(start, end)                        */
    uint32_t minus;        /* != 0: reverse-complement the subsequence  */
    uint32_t reserved;
} skh_slice;
typedef struct skh_subseq_result {
    uint64_t out_bytes;
    uint32_t status;       /* SKH_SUBSEQ_* */
    uint32_t reserved;
    uint64_t bad_slice;
} skh_subseq_result;
uint64_t skh_subseq_workspace_size(uint64_t n);
int skh_subseq_dev(skh_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual, const uint64_t *d_offsets,
                   uint64_t nrec, uint64_t total_bytes, const skh_slice *d_slices, uint64_t n,
                   int64_t start, int64_t end, int minus, const uint8_t *lut,
                   uint8_t *d_out_seq, uint8_t *d_out_qual, uint64_t out_capacity, uint64_t *d_out_offsets,
                   void *d_workspace, uint64_t workspace_bytes, skh_subseq_result *d_result, void *stream);

/* ---- FASTA/FASTQ text -> batch on the device ----------------------------------------
 * The step before the hot path: replaces the serial record scanner behind
 * fastx.NewReader / reader.Read() (call sites seqkit/cmd/seq.go:115-127,
 * fx2tab.go:96-106, stat.go:87-104) for a block of file text that already lies in
 * device memory.  Rules: FASTA records start at lines beginning with '>' and run to the
 * next such line, line ends (LF or CRLF) and blank lines are dropped from the sequence;
 * FASTQ records are four lines ('@'name, sequence, '+'..., quality of equal length),
 * blank lines only after the last record (multi-line FASTQ is reported as a format
 * error).  The name of record r is text[name_pos[r] .. name_pos[r]+name_len[r]).
 *
 * `final` != 0 says the text ends at the end of the file; otherwise the last record of
 * the block is taken to be incomplete and is left for the next call: result.consumed is
 * where it starts.  The outcome is written to DEVICE memory d_result (read it after
 * synchronising the stream): status SKH_PARSE_OK, SKH_PARSE_EFORMAT (error_line = first
 * offending 0-based line of the block) or SKH_PARSE_ECAPACITY (more lines than
 * max_lines, more records than max_records, or more sequence than seq_capacity: retry
 * with larger buffers; no output is valid).
 * d_text must be 16-byte aligned and readable up to skh_padded_size(text_bytes);
 * d_seq / d_qual (FASTQ only) hold seq_capacity bytes, d_offsets max_records+1 entries,
 * d_name_pos / d_name_len max_records entries; d_workspace is 256-byte aligned scratch of
 * skh_parse_workspace_size(text_bytes, max_lines) bytes. */
#define SKH_FASTA 1
#define SKH_FASTQ 2
/* FASTQ as the reference's reader accepts it (README.md:88, doc/docs/usage.md:357-359): sequence and quality of a
 * record may be wrapped over several lines -- sequence lines up to the first line that starts with '+', quality
 * lines until they hold as many bytes as the sequence -- and blank lines may stand between records.  Four-line
 * records are a special case, but SKH_FASTQ parses them several times faster; result.wrapped tells whether the
 * block needed this format.  Which '@' lines start records is found by following, from the first line, the line
 * each record would end at (pointer doubling over the line table).  Text that is not such FASTQ -- and a record cut
 * short by the end of the file -- is SKH_PARSE_EFORMAT with error_line = -1: this format does not name the line.
 * (A head line with nothing but blank lines between it and the end of the file is a record without bases, as a head
 * followed by a '+' line is.)
 * The workspace is skh_parse_wrapped_workspace_size(text_bytes, max_lines, max_records) bytes. */
#define SKH_FASTQ_WRAPPED 3
#define SKH_PARSE_OK        0
#define SKH_PARSE_EFORMAT   1
#define SKH_PARSE_ECAPACITY 2
typedef struct skh_parse_result {
    uint64_t nrec;        /* complete records found                         */
    uint64_t seq_bytes;   /* = offsets[nrec]                                */
    uint64_t consumed;    /* text bytes those records span                  */
    uint64_t lines;       /* text lines looked at                           */
    int64_t  error_line;  /* first offending line, or -1                    */
    uint32_t status;      /* SKH_PARSE_*                                    */
    uint32_t wrapped;     /* SKH_FASTQ_WRAPPED: 1 = some record was not in the four-line form, or blank lines
                           * stood between records; 0 otherwise and for the other formats */
} skh_parse_result;
uint64_t skh_parse_workspace_size(uint64_t text_bytes, uint64_t max_lines);
uint64_t skh_parse_wrapped_workspace_size(uint64_t text_bytes, uint64_t max_lines, uint64_t max_records);
int skh_parse_dev(skh_ctx *ctx, const uint8_t *d_text, uint64_t text_bytes, int format, int final,
                  uint8_t *d_seq, uint8_t *d_qual, uint64_t seq_capacity, uint64_t *d_offsets,
                  uint64_t *d_name_pos, uint32_t *d_name_len, uint64_t max_records,
                  void *d_workspace, uint64_t workspace_bytes, uint64_t max_lines,
                  skh_parse_result *d_result, void *stream);

/* ---- batch -> FASTA/FASTQ output text on the device ---------------------------------
 * The step after the hot path: the bytes `seqkit seq` writes for every record
 * (seqkit/cmd/seq.go:163-276): head line, sequence wrapped by byteutil.WrapByteSlice
 * ('\n' after every line_width bytes, none after the last chunk, nothing for
 * line_width 0 or an empty sequence), and for FASTQ "+" and the quality string.
 * Per record, in this order and each part only if selected:
 *     [mark if print_seq] name-or-ID '\n'            (print_name; only_id cuts the name
 *                                                     at its first white space)
 *     wrapped(sequence) '\n'                         (print_seq)
 *     ["+\n" if plus_line] wrapped(quality) '\n'     (print_qual)
 * A batch without a quality plane (d_qual == NULL) printed with print_qual gets an empty quality
 * string: "+" and an empty line, what the command prints for the records of a FASTA file that
 * follows a FASTQ file (the layout is decided once, seq.go:140-149,258-276).
 * Records shorter than min_len or longer than max_len (when >= 0) are skipped.
 * The caller maps the command's flags (-n -s -q -i, FASTQ forces line_width 0,
 * seq.go:136-138,163-241) onto these fields.  Names are read from the text the batch was
 * parsed from.  The result (bytes written, or SKH_PARSE_ECAPACITY when out_capacity is too
 * small: nothing valid was written) goes to DEVICE memory d_result. */
typedef struct skh_seq_format {
    int64_t min_len, max_len;   /* < 0 = no limit (seq.go:165-172) */
    int32_t line_width;
    uint8_t print_name, print_seq, print_qual, plus_line, only_id;
    char    mark;               /* '>' or '@' */
    uint8_t empty_as_fasta;     /* != 0: a record without bases is printed as FASTA ('>' name, an empty
                                 * line, no quality), as fastx Record.FormatToWriter chooses the layout
                                 * per record by "has qualities" (subseq.go:354) */
} skh_seq_format;
typedef struct skh_format_result {
    uint64_t out_bytes;
    uint32_t status;            /* SKH_PARSE_OK or SKH_PARSE_ECAPACITY */
    uint32_t reserved;
} skh_format_result;
uint64_t skh_format_workspace_size(uint64_t nrec);
int skh_format_seq_dev(skh_ctx *ctx, const uint8_t *d_text, const uint64_t *d_name_pos,
                       const uint32_t *d_name_len, const uint8_t *d_seq, const uint8_t *d_qual,
                       const uint64_t *d_offsets, uint64_t nrec, const skh_seq_format *fmt,
                       uint8_t *d_out, uint64_t out_capacity, void *d_workspace,
                       uint64_t workspace_bytes, skh_format_result *d_result, void *stream);

/* fx2tab rows on the device (seqkit/cmd/fx2tab.go:112-148): per record
 *     name-or-ID '\t' [sequence] '\t' [quality] { '\t' number } '\n'
 * with one number per column: the length ("%d"), GC content and GC skew ("%.2f" of
 * (g+c)*100 and (g-c)/(g+c)*100 where g and c are divided by the length separately,
 * fx2tab.go:128-133), or the content of a letter list ("%.2f" of sum/len*100; 0 for an
 * empty sequence; "NaN" prints as Go prints it).  The letter counts come from up to
 * SKH_TAB_MAX_PASSES skh_count_dev results (d_counts[p] is nrec x nclass[p]); a term names
 * a letter's count as pass * 8 + class. */
#define SKH_TAB_MAX_COLS   16
#define SKH_TAB_MAX_TERMS  30
#define SKH_TAB_MAX_PASSES 4
#define SKH_TAB_LENGTH  0
#define SKH_TAB_GC      1   /* term[0] = G, term[1] = C */
#define SKH_TAB_GC_SKEW 2   /* term[0] = G, term[1] = C */
#define SKH_TAB_CONTENT 3   /* nterm letters, repeated letters count again */
typedef struct skh_tab_column {
    uint8_t kind, nterm;
    uint8_t term[SKH_TAB_MAX_TERMS];
} skh_tab_column;
typedef struct skh_tab_format {
    uint32_t ncols;
    uint8_t only_id, print_seq, print_qual;
    uint8_t alphabet;              /* != 0: a last column with the byte values present in the record, ascending
                                    * (alphabetStr, fx2tab.go:145-147,167-180), read from d_letters */
    skh_tab_column cols[SKH_TAB_MAX_COLS];
    const uint32_t *d_letters;     /* DEVICE memory, nrec x 8 words from skh_letters_dev; the text-block form
                                    * fills it in from skh_text_letters */
} skh_tab_format;
/* Which byte values occur in each record: d_present[r * 8 + (b >> 5)] has bit (b & 31) set when byte
 * value b occurs in record r (the set alphabetStr builds with a map, fx2tab.go:167-180). */
int skh_letters_dev(skh_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_offsets, uint64_t nrec,
                    uint64_t total_bytes, uint32_t *d_present, void *stream);
int skh_format_tab_dev(skh_ctx *ctx, const uint8_t *d_text, const uint64_t *d_name_pos,
                       const uint32_t *d_name_len, const uint8_t *d_seq, const uint8_t *d_qual,
                       const uint64_t *d_offsets, uint64_t nrec,
                       const uint64_t *const d_counts[SKH_TAB_MAX_PASSES],
                       const uint32_t nclass[SKH_TAB_MAX_PASSES], const skh_tab_format *fmt,
                       uint8_t *d_out, uint64_t out_capacity, void *d_workspace,
                       uint64_t workspace_bytes, skh_format_result *d_result, void *stream);

/* ---- text blocks: file text in, file text out ----------------------------------------
 * The library-owned form of the two entry points above, for callers that hold file text
 * in host memory (what a cgo shim or the C++ driver uses instead of a record reader):
 * a block owns pinned staging for the text, the device planes and tables, and a pinned
 * output buffer.  Per block of a file: fill skh_text_buffer() with up to
 * skh_text_capacity() bytes (the unconsumed tail of the previous block first), call
 * skh_text_parse (upload + parse; grows its tables as needed), then any of revcomp /
 * stats / format.  All calls are synchronous. */
typedef struct skh_text skh_text;
int  skh_text_alloc(skh_ctx *ctx, uint64_t block_bytes, skh_text **out);
void skh_text_free(skh_text *t);
uint8_t *skh_text_buffer(skh_text *t);
uint64_t skh_text_capacity(const skh_text *t);
/* Uploads text_bytes of the buffer and parses them (skh_parse_dev); *out is host memory.
 * A format error is reported in out->status, not in the return code. */
int skh_text_parse(skh_ctx *ctx, skh_text *t, uint64_t text_bytes, int format, int final,
                   skh_parse_result *out);
/* The same for text that starts text_offset bytes into the buffer (any offset): a reader that
 * runs ahead of the parser fills a block from some headroom on and the unconsumed tail of the
 * block before it is put in front afterwards.  skh_text_name(s) keep reporting positions in
 * skh_text_buffer(). */
int skh_text_parse_at(skh_ctx *ctx, skh_text *t, uint64_t text_offset, uint64_t text_bytes, int format,
                      int final, skh_parse_result *out);
/* Starts the upload of buffer bytes [offset, offset + nbytes) and returns at once: a reader that
 * runs ahead calls it as soon as it has filled a block, so the bulk of the block crosses the link
 * while the block before it is still being parsed.  The next skh_text_parse_at on the block whose
 * text covers that range uploads only the bytes around it (another text range uploads everything,
 * as without this call).  The range must not be written to between this call and the return of
 * that parse. */
int skh_text_upload(skh_ctx *ctx, skh_text *t, uint64_t offset, uint64_t nbytes);
/* Copies offsets[first .. first+count) of the parsed block to host memory. */
int skh_text_offsets(skh_ctx *ctx, skh_text *t, uint64_t first, uint64_t count, uint64_t *offsets);
/* Copies sequence bytes [first_byte, first_byte+count) of the block to host memory
 * (e.g. the head of the first record, for the alphabet guess of seq.go:118-127). */
int skh_text_sequence(skh_ctx *ctx, skh_text *t, uint64_t first_byte, uint64_t count, uint8_t *dst);
/* skh_revcomp_dev on the block's planes (qualities reversed too when reverse_qual). */
int skh_text_revcomp(skh_ctx *ctx, skh_text *t, const uint8_t *lut, int do_reverse, int reverse_qual);
/* skh_format_seq_dev + download: *out points at out_bytes of pinned memory owned by the
 * block, valid until its next call. */
int skh_text_format_seq(skh_ctx *ctx, skh_text *t, const skh_seq_format *fmt,
                        const uint8_t **out, uint64_t *out_bytes);
/* Split forms for overlap: *_submit run the formatter, start the download of the output and
 * return; skh_text_output_wait blocks until the output has arrived (*out as above; 0 bytes when
 * nothing was submitted).  Between the two the caller works on another block, whose kernels then
 * run beside this block's download. */
int skh_text_format_seq_submit(skh_ctx *ctx, skh_text *t, const skh_seq_format *fmt);
int skh_text_format_tab_submit(skh_ctx *ctx, skh_text *t, const skh_tab_format *fmt);
int skh_text_output_wait(skh_ctx *ctx, skh_text *t, const uint8_t **out, uint64_t *out_bytes);
/* skh_count_dev on the block; the counts stay on the device in slot `pass`
 * (< SKH_TAB_MAX_PASSES) for skh_text_format_tab. */
int skh_text_count(skh_ctx *ctx, skh_text *t, const uint8_t class_of[256], uint32_t nclass, uint32_t pass);
/* skh_letters_dev on the block; the result stays on the device for skh_text_format_tab (fmt->alphabet). */
int skh_text_letters(skh_ctx *ctx, skh_text *t);
/* skh_format_tab_dev on the block with the counts of the passes run so far + download. */
int skh_text_format_tab(skh_ctx *ctx, skh_text *t, const skh_tab_format *fmt,
                        const uint8_t **out, uint64_t *out_bytes);
/* skh_validate_dev on the block; *out is host memory. */
int skh_text_validate(skh_ctx *ctx, skh_text *t, const uint8_t valid[256], uint64_t limit,
                      skh_validate_result *out);
/* Keeps only the first nrec records of the parsed block (e.g. those before an invalid one). */
int skh_text_truncate(skh_ctx *ctx, skh_text *t, uint64_t nrec);
/* Where record `record`'s name lies in the block's text buffer. */
int skh_text_name(skh_ctx *ctx, skh_text *t, uint64_t record, uint64_t *pos, uint32_t *len);
/* Quality bytes [first_byte, first_byte+count) of a FASTQ block copied to host memory (e.g. the
 * leading records, for the encoding guess of convert.go:177-183). */
int skh_text_quality(skh_ctx *ctx, skh_text *t, uint64_t first_byte, uint64_t count, uint8_t *dst);
/* skh_validate_dev on the block's quality plane (all bytes of every record); *out is host memory. */
int skh_text_validate_quality(skh_ctx *ctx, skh_text *t, const uint8_t valid[256], skh_validate_result *out);
/* The block's qualities looked up in lut (convert.go:270); bases are kept. */
int skh_text_map_quality(skh_ctx *ctx, skh_text *t, const uint8_t lut[256]);
/* name_pos / name_len of records [first, first+count) copied to host memory; the names
 * themselves lie in skh_text_buffer(). */
int skh_text_names(skh_ctx *ctx, skh_text *t, uint64_t first, uint64_t count, uint64_t *pos, uint32_t *len);
/* Replaces the name spans of records [first, first+count) by spans the caller picked inside
 * skh_text_buffer() (inside the parsed text): how a caller prints record IDs cut out by a rule of its
 * own, e.g. the capture group of --id-regexp (seqkit/cmd/root.go:68, helper.go:125-135; handed to the
 * reader at seq.go:115, fx2tab.go:96), with only_id off. */
int skh_text_set_names(skh_ctx *ctx, skh_text *t, uint64_t first, uint64_t count, const uint64_t *pos,
                       const uint32_t *len);
/* skh_subseq_dev on the block: afterwards the block holds the n slices as its records, and
 * every later call on it (format, count, stats, validate) sees those.  `slices` is host memory
 * (NULL = region mode with start / end / minus, n ignored).  names == NULL keeps the names of
 * the source records (region mode only); otherwise record i of the result is named
 * names[name_pos[i] .. name_pos[i]+name_len[i]) (host memory, names_bytes in all), as the
 * feature modes rename their records (subseq.go:453,542). */
int skh_text_subseq(skh_ctx *ctx, skh_text *t, const skh_slice *slices, uint64_t n, int64_t start, int64_t end,
                    int minus, const uint8_t *lut, const uint8_t *names, uint64_t names_bytes,
                    const uint64_t *name_pos, const uint32_t *name_len);
/* skh_remove_gaps_dev on the block (before skh_text_revcomp): afterwards the block's records are
 * the records without their gap bytes.  The length filter of skh_text_format_seq keeps looking at
 * the lengths before the removal, as the command does (seq.go:129-134 come before seq.go:196). */
int skh_text_remove_gaps(skh_ctx *ctx, skh_text *t, const uint8_t is_gap[256]);
/* skh_stats_dev on the block, merged into *inout. */
int skh_text_stats(skh_ctx *ctx, skh_text *t, const uint8_t is_gap[256], skh_stats_partial *inout);

/* The context a text block was allocated on (a driver that deals blocks to several GPUs passes it
 * to the calls on that block; every skh_text_* call checks that block and context belong together). */
skh_ctx *skh_text_ctx(const skh_text *t);

/* ---- several GPUs ---------------------------------------------------------------------
 * The reference's only parallel knob on this path is -j / seq.ComplementThreads
 * (seqkit/cmd/seq.go:88, root.go:60-64) and its per-file accumulators (stat.go:106-125); here records
 * are dealt to GPUs instead.  One context per GPU.  Reverse-complement and per-record counts of
 * different records never meet, so sharding needs no exchange:
 *   skh_split_by_bytes  cuts[0..nparts]: contiguous record ranges [cuts[k], cuts[k+1]) of about equal
 *                       byte volume (nearest record boundary to every k/nparts of the bytes);
 *   skh_split_record    cuts[0..nparts]: byte positions (multiples of 16, then `length`) that cut one
 *                       long record into slices; input slice [cuts[k], cuts[k+1]) is a record of its
 *                       own whose reverse(-complement) belongs at [length - cuts[k+1], length - cuts[k])
 *                       of the result.
 * The one collective is the merge of the per-GPU `stats` totals, done by RCCL over xGMI:
 * sum of num / len_sum / gap_sum, min of len_min, max of len_max (skh_stats_merge, element-wise).
 * A communicator is made either over several contexts of ONE process (skh_comm_init_all: the
 * driver's --gpus N; contexts that share a device are merged on the host first) or as one rank of a
 * multi-process job, one GPU per process (rank 0 calls skh_comm_unique_id, hands the bytes to the
 * other ranks by any means it has, every rank calls skh_comm_init_rank).  RCCL is loaded on first use
 * (the copy the process already holds, if any).  RCCL itself may print a line about its version on
 * standard output while a communicator is made (skh_comm_unique_id, skh_comm_init_all,
 * skh_comm_init_rank, skh_multi_comm_init and the first skh_stats_multi): a caller whose standard
 * output carries data must not have another thread writing to it during those calls, and may point
 * file descriptor 1 elsewhere around them.  skh_comm_rccl_version reports ncclGetVersion() of the
 * RCCL in use, the NCCL_VERSION_CODE of the header the library was compiled against, and whether
 * the RCCL in use was already loaded by the process (1) or opened by this library (0). */
typedef struct skh_comm skh_comm;
#define SKH_COMM_ID_BYTES 128
int skh_split_by_bytes(const uint64_t *offsets, uint64_t nrec, int nparts, uint64_t *cuts);
int skh_split_record(uint64_t length, int nparts, uint64_t *cuts);
int skh_comm_init_all(skh_ctx *const *ctxs, int n, skh_comm **out);
int skh_comm_unique_id(uint8_t id[SKH_COMM_ID_BYTES]);
int skh_comm_init_rank(skh_ctx *ctx, const uint8_t id[SKH_COMM_ID_BYTES], int nranks, int rank,
                       skh_comm **out);
void skh_comm_destroy(skh_comm *c);
/* GPUs of this process in the communicator (distinct devices of skh_comm_init_all; 1 for init_rank). */
int skh_comm_local_ranks(const skh_comm *c);
int skh_comm_rccl_version(int *runtime_version, int *header_version, int *shared_with_process);
/* Host form: partials[i] belongs to context i of skh_comm_init_all (one entry for init_rank); on
 * return every entry holds the totals over all GPUs and processes.  Synchronous. */
int skh_stats_allreduce(skh_comm *c, skh_stats_partial *partials);
/* Device form: d_partials[r] is DEVICE memory on local GPU r (e.g. what skh_stats_dev wrote),
 * reduced in place on streams[r] (NULL array or entry: the communicator's own stream), not synchronised. */
int skh_stats_allreduce_dev(skh_comm *c, skh_stats_partial *const *d_partials, void *const *streams);

/* ---- pinned-host batches (what a cgo / C++ reader fills) ---------------------------- */

/* Allocates pinned host staging plus matching device buffers for up to max_bytes of
 * sequence and max_records records (with_qual != 0 adds the quality plane). */
int skh_batch_alloc(skh_ctx *ctx, uint64_t max_bytes, uint64_t max_records, int with_qual,
                    skh_batch **out);
void skh_batch_free(skh_batch *b);
uint8_t  *skh_batch_seq(skh_batch *b);       /* max_bytes (padded) bytes            */
uint8_t  *skh_batch_qual(skh_batch *b);      /* NULL when allocated without quality */
uint64_t *skh_batch_offsets(skh_batch *b);   /* max_records + 1 entries             */
uint64_t  skh_batch_max_bytes(const skh_batch *b);
uint64_t  skh_batch_max_records(const skh_batch *b);

/* In place on the batch (host view): upload, transform on the GPU, download.
 * lut NULL = no complement; reverse_qual reverses the quality plane too
 * (ignored when the batch has none). */
int skh_revcomp(skh_ctx *ctx, skh_batch *b, uint64_t nrec, const uint8_t *lut,
                int do_reverse, int reverse_qual);
/* counts: host array of nrec*nclass uint64, row-major by record. */
int skh_count(skh_ctx *ctx, skh_batch *b, uint64_t nrec, const uint8_t class_of[256],
              uint32_t nclass, uint64_t *counts);
/* Adds this batch's totals into *inout (sum for num/len_sum/gap_sum, min/max for the
 * extremes); initialise *inout with skh_stats_init before the first batch of a file. */
void skh_stats_init(skh_stats_partial *p);
int skh_stats_reduce(skh_ctx *ctx, skh_batch *b, uint64_t nrec, const uint8_t is_gap[256],
                     skh_stats_partial *inout);
/* skh_validate_dev on a pinned batch; *out is host memory. */
int skh_validate(skh_ctx *ctx, skh_batch *b, uint64_t nrec, const uint8_t valid[256], uint64_t limit,
                 skh_validate_result *out);
/* Merge rule for per-GPU partials (what an all-reduce applies element-wise). */
void skh_stats_merge(skh_stats_partial *into, const skh_stats_partial *from);

/* Split-phase forms for overlap: skh_*_submit enqueue upload + kernels + download on the
 * batch's own stream and return; skh_batch_wait blocks until that batch is done.  Two
 * batches submitted back to back overlap one's transfers with the other's kernels. */
int skh_revcomp_submit(skh_ctx *ctx, skh_batch *b, uint64_t nrec, const uint8_t *lut,
                       int do_reverse, int reverse_qual);
int skh_count_submit(skh_ctx *ctx, skh_batch *b, uint64_t nrec, const uint8_t class_of[256],
                     uint32_t nclass, uint64_t *counts);
int skh_batch_wait(skh_batch *b);

/* ---- one pinned batch, several GPUs at once ---------------------------------------------
 * The product form of the sharding rules above, for a caller that holds ONE batch in pinned host
 * memory (a skh_batch filled by its reader) and several GPUs: what the reference does with -j inside one
 * long record (seq.ComplementThreads = config.Threads, seqkit/cmd/seq.go:88, README.md:285-295) and with
 * its per-file accumulators (stat.go:106-125), done across GPUs.  skh_multi_init starts one worker thread
 * per context.  Every call cuts the batch into one share of about equal byte volume per context
 * (skh_split_plane), each worker uploads its share in chunks on two streams of its GPU, runs the
 * single-GPU kernels on it and downloads the results to the share's own place in the batch, all
 * workers at once; nothing moves between GPUs.  The batch may belong to any context (its host planes
 * are pinned for every GPU); its contents are as after the single-GPU call of the same name:
 *   skh_revcomp_multi  in place in the batch; the pieces of a record that was cut land at the mirrored
 *                      places [L-b, L-a) of that record once every share has been read;
 *   skh_count_multi    counts[nrec * nclass]; a record that was cut gets the sum of its pieces' rows;
 *   skh_stats_multi    per-share totals (a record counts where it starts) merged by the path's one
 *                      collective, skh_stats_allreduce over a communicator of the contexts, into *inout.
 * skh_split_plane: cuts[0..nparts] are byte positions of the plane; a cut lies at the record boundary
 * nearest to k/nparts of the bytes when that boundary is within 1/16 of a share of it
 * (skh_split_by_bytes' rule), else inside the record at a multiple of 16 bytes from the record's start
 * (skh_split_record's rule).  skh_multi_shares reports what every context did in the last call
 * (n = skh_multi_size entries).  Calls on one skh_multi must not overlap. */
typedef struct skh_multi skh_multi;
typedef struct skh_share_info {
    uint64_t first_byte, last_byte;   /* the share: bytes [first_byte, last_byte) of the plane          */
    uint64_t first_record, nrecords;  /* records it touches, the cut ones at either end included        */
    uint64_t head_cut, tail_cut;      /* bytes of its first / last record that lie outside the share    */
    uint32_t chunks, reserved;        /* upload-kernel-download rounds the share was worked in          */
    double   seconds;                 /* wall time of the share on its worker thread                    */
} skh_share_info;
int  skh_split_plane(const uint64_t *offsets, uint64_t nrec, int nparts, uint64_t *cuts);
int  skh_multi_init(skh_ctx *const *ctxs, int n, skh_multi **out);
void skh_multi_destroy(skh_multi *m);
int  skh_multi_size(const skh_multi *m);
/* Makes the communicator skh_stats_multi needs ahead of its first call (see "several GPUs" about
 * RCCL and standard output); without it the first skh_stats_multi makes it. */
int  skh_multi_comm_init(skh_multi *m);
int  skh_revcomp_multi(skh_multi *m, skh_batch *b, uint64_t nrec, const uint8_t *lut, int do_reverse,
                       int reverse_qual);
int  skh_count_multi(skh_multi *m, skh_batch *b, uint64_t nrec, const uint8_t class_of[256], uint32_t nclass,
                     uint64_t *counts);
int  skh_stats_multi(skh_multi *m, skh_batch *b, uint64_t nrec, const uint8_t is_gap[256],
                     skh_stats_partial *inout);
int  skh_multi_shares(const skh_multi *m, skh_share_info *info);

#ifdef __cplusplus
}
#endif
#endif
