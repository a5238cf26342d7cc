// kernels.h -- launch interface between the C ABI (capi.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SKH_BLOCK 256          // threads per workgroup (4 wavefronts)
#define SKH_LONG_T 32768ull    // records longer than this are served tile-wise
#define SKH_TILE 32768ull      // bytes of plane per tile; must be >= SKH_LONG_T

namespace skh {

struct Table256 { uint8_t v[256]; };   // passed by value in the kernel argument block

struct StatsPartial { int64_t num, len_sum, gap_sum, len_min, len_max; };

struct LaunchCfg {
    int logG;            // log2 of lanes per record in the short kernels (0..6), from the batch's mean length
    int logG_count;      // the same for the counting kernel (a lane takes more chunks there: count_u of them per round)
    int count_u;
    int use_hint;        // 1: the short kernels lower logG to what the sampled short records ask for
                         // (batches of short reads with a few very long records among them)
    uint32_t max_grid;   // optional cap on workgroups of a grid-stride launch (0 = none)
    uint32_t n_cu;       // compute units of the device
    int staged;          // reverse, short records: 1 = LDS-staged kernel, 0 = direct (lane exchange) kernel
    int flat;            // reverse out of place: 1 = the flat kernel (pieces of all records of a step dealt over the lanes)
                         // instead of either
    int flat_inplace;    // the same choice for the in-place form
    int inplace_run;     // in place, long records: log2 of the 32-KiB tiles per workgroup run (0, 1 or 2)
    uint32_t staged_steps;   // staged kernel: steps per wavefront (0 = default)
    uint32_t flat_steps;     // flat kernel: steps per wavefront (0 = default)
    uint32_t short_steps;    // direct reverse kernel: steps per wavefront (0 = about 32 KiB of plane per workgroup)
    uint32_t count_steps;    // short count kernel: steps per wavefront (0 = default)
    uint32_t stats_steps;    // stats kernel: runs of chunks per workgroup (0 = default)
    int fuse_fastq;      // reverse: 1 = bases and qualities in one launch, 0 = one launch per plane
    int span_u;          // reverse: 0 = lane-group kernels, 2 or 4 = span kernel with that many chunks per lane
};

// What a launch learns about its batch on the device, in one small pass before the kernels proper
// (launch_prepare): 12 KiB of device scratch per launch in flight.
constexpr uint32_t LONG_LIST = 512;      // long records a launch can list; with more of them the long-record kernels walk the offsets
struct LongRec { uint64_t s, e, rec; }; // plane range and index of a record longer than SKH_LONG_T
struct BatchHint {
    uint32_t n_long;       // records longer than SKH_LONG_T met by the short kernel (the long-record kernel
                           // returns at once when it is 0)
    uint32_t mean_short;   // mean length of the sampled records of at most SKH_LONG_T bytes (~0u: none sampled)
    uint32_t stage_check;  // checked build only: wavefront steps of the staged kernel that met an incomplete stage buffer
    uint32_t max_short;    // the longest of those sampled records
    LongRec list[LONG_LIST];   // the first LONG_LIST of them, in no particular order: with a few long records among
                               // millions of short ones the long-record kernels look here instead of searching and
                               // walking the offsets of every run of the plane
};

// log2 of the lanes that serve one record in the short kernels, from a mean record length.
// Reverse: about three 16-byte chunks per lane (150-base reads -> 4 lanes per record).  Counting: the
// kernel loops over a lane's chunks and has nothing to hold per chunk, so four to five chunks per
// lane (measured: reads of ~100 bases 10-17 % faster with 2 lanes per record than with 4; 150 and
// 250 bases fastest with 4).
__host__ __device__ inline int lanes_log2_reverse(uint64_t mean)
{
    const uint64_t chunks = (mean + 15) / 16 + 1, lanes = (chunks + 2) / 3;
    int lg = 0;
    while (lg < 6 && (1ull << lg) < lanes) lg++;
    return lg;
}
// Counting kernel: a lane takes U chunks per round (the loads of the slots that are in use go out before the first
// lookup), G = 2^k lanes serve a record.  Measured on fixed and +-20 % lengths, every G x U (% of HBM peak, best / what
// "one round with the fewest idle slots" chose before): 350 bases 4 x 5: 59 / 8 x 3: 56; 700: 8 x 5: 66 / 16 x 4: 55;
// 1000: 8 x 5: 63-71 / 16 x 4: 58; 4000: 32 x 5: 74 / 64 x 3: 63; 150 and 250 bases as before (4 x 3: 65, 4 x 5: 62).
// Few lanes with several rounds beat many lanes with one: a wavefront then holds more records, and what it has in
// flight does not depend on one record's length.  So: G = chunks / 8 rounded up to a power of two (two to three
// rounds of five chunks for records beyond 20 chunks), doubled for a short read that would miss its single round by a
// chunk or two; U = what one round needs, between 3 and 5.
__host__ __device__ inline int lanes_log2_count(uint64_t mean, int U)
{
    (void)U;
    const uint64_t chunks = (mean + 15) / 16 + 1, lanes = (chunks + 7) / 8;
    int lg = 1;
    while (lg < 6 && (1ull << lg) < lanes) lg++;
    if (chunks <= 20 && (5ull << lg) < chunks) lg++;
    return lg;
}
inline int count_chunks_per_lane(uint64_t mean)
{
    const uint64_t chunks = (mean + 15) / 16 + 1;
    const uint64_t G = 1ull << lanes_log2_count(mean, 0), u = (chunks + G - 1) / G;
    return u < 3 ? 3 : (u > 5 ? 5 : (int)u);
}

// lut == nullptr: reverse only.  qin/qout == nullptr: no quality plane.  hint: device scratch of this
// launch (written by the launch itself).
hipError_t launch_reverse(const LaunchCfg& cfg, const uint8_t* in, uint8_t* out, const uint8_t* qin,
                          uint8_t* qout, const uint64_t* offs, uint64_t nrec, uint64_t total,
                          const Table256* lut, BatchHint* hint, hipStream_t st);
// The same with one plane for input and output (reverse, or reverse + map through lut).
hipError_t launch_reverse_inplace(const LaunchCfg& cfg, uint8_t* plane, const uint64_t* offs, uint64_t nrec,
                                  uint64_t total, const Table256* lut, BatchHint* hint, hipStream_t st);
hipError_t launch_map_plane(const LaunchCfg& cfg, const uint8_t* in, uint8_t* out, uint64_t total,
                            const Table256& lut, hipStream_t st);
// neutral: a byte value of no class, or 256 if every byte value has a class.
hipError_t launch_count(const LaunchCfg& cfg, const uint8_t* seq, const uint64_t* offs, uint64_t nrec,
                        uint64_t total, const Table256& cls, uint32_t nclass, uint32_t neutral,
                        uint64_t* counts, BatchHint* hint, hipStream_t st);
// present: nrec x 8 words, bit (b & 31) of word (b >> 5) set when byte value b occurs in the record
hipError_t launch_letters(const LaunchCfg& cfg, const uint8_t* seq, const uint64_t* offs, uint64_t nrec, uint64_t total,
                          uint32_t* present, BatchHint* hint, hipStream_t st);
// slots: STATS_SLOT_WORDS words of device scratch for this launch (partial gap sums)
constexpr uint32_t STATS_SLOT_WORDS = 512;
hipError_t launch_stats(const LaunchCfg& cfg, const uint8_t* seq, const uint64_t* offs, uint64_t nrec,
                        uint64_t total, const Table256& is_gap, StatsPartial* out, uint64_t* slots, hipStream_t st);

struct ValidateResult { uint64_t record, offset; uint32_t letter, found; };   // mirrors skh_validate_result
hipError_t launch_validate(const LaunchCfg& cfg, const uint8_t* seq, const uint64_t* offs, uint64_t nrec,
                           uint64_t total, const Table256& valid, uint64_t limit, ValidateResult* out, hipStream_t st);

// ---- text -> batch (parse.hip) ----------------------------------------------------------

enum : uint32_t { PARSE_OK = 0, PARSE_EFORMAT = 1, PARSE_ECAPACITY = 2 };

struct ParseResult {          // mirrors skh_parse_result of the C ABI
    uint64_t nrec;            // complete records found
    uint64_t seq_bytes;       // = offsets[nrec]
    uint64_t consumed;        // text bytes those records span; the next block starts there
    uint64_t lines;           // text lines looked at
    int64_t error_line;       // 0-based line of the first format violation, or -1
    uint32_t status;          // PARSE_*
    uint32_t pad;
};

uint64_t parse_workspace_size(uint64_t text_bytes, uint64_t max_lines);
// FASTQ with records wrapped over several lines (and four-line records among them): the same outputs, by a pass
// sequence of its own; workspace = parse_wrapped_workspace_size (it contains the parser's own).  ParseResult.pad
// is 1 when some record was not in the four-line form.
uint64_t parse_wrapped_workspace_size(uint64_t text_bytes, uint64_t max_lines, uint64_t max_records);
hipError_t launch_parse_wrapped(const uint8_t* text, uint64_t n, int final, uint8_t* seq, uint8_t* qual, uint64_t seq_capacity,
                                uint64_t* offsets, uint64_t* name_pos, uint32_t* name_len, uint64_t max_records, void* workspace,
                                uint64_t max_lines, ParseResult* res, hipStream_t s);
hipError_t launch_parse(int fastq, const uint8_t* text, uint64_t n, int final, uint8_t* seq, uint8_t* qual,
                        uint64_t seq_capacity, uint64_t* offsets, uint64_t* name_pos, uint32_t* name_len,
                        uint64_t max_records, void* workspace, uint64_t max_lines, ParseResult* res, hipStream_t s);

// ---- batch -> output text (format.hip) ---------------------------------------------------

enum : uint32_t { FORMAT_OK = 0, FORMAT_ECAPACITY = 2 };

struct FormatSeqArgs {
    const uint8_t* text;          // names are text[name_pos[r] .. +name_len[r])
    const uint64_t* name_pos;
    const uint32_t* name_len;
    const uint8_t* seq;
    const uint8_t* qual;          // only read when print_qual
    const uint64_t* offsets;
    const uint64_t* filter_offsets;   // lengths the min_len / max_len filter looks at (nullptr: offsets)
    uint64_t nrec;
    int64_t min_len, max_len;     // < 0 = no limit
    uint32_t width;               // line width of sequence and quality (0 = no wrapping)
    uint8_t print_name, print_seq, print_qual, plus_line, only_id, mark;
    uint8_t empty_fasta;          // a record without bases: '>' head, no quality part
};

struct FormatResult { uint64_t out_bytes; uint32_t status; uint32_t pad; };

// fx2tab rows: name \t seq \t qual, then one "\t<number>" per column
constexpr int TAB_MAX_COLS = 16, TAB_MAX_TERMS = 30, TAB_MAX_PASSES = 4;
enum : uint8_t { TAB_LENGTH = 0, TAB_GC = 1, TAB_GC_SKEW = 2, TAB_CONTENT = 3 };
struct TabColumn {
    uint8_t kind;                      // TAB_*
    uint8_t nterm;                     // TAB_CONTENT: letters summed; TAB_GC / TAB_GC_SKEW: term[0] = G, term[1] = C
    uint8_t term[TAB_MAX_TERMS];   // pass * 8 + class of a letter's count
};
struct FormatTabArgs {
    const uint8_t* text; const uint64_t* name_pos; const uint32_t* name_len;
    const uint8_t* seq; const uint8_t* qual; const uint64_t* offsets; uint64_t nrec;
    const uint64_t* counts[TAB_MAX_PASSES];    // per counting pass: nrec x nclass[pass]
    uint32_t nclass[TAB_MAX_PASSES];
    const uint32_t* letters;                   // alphabet column: nrec x 8 words from launch_letters (nullptr: no column)
    uint32_t ncols;
    uint8_t only_id, print_seq, print_qual, pad;
    TabColumn cols[TAB_MAX_COLS];
};
hipError_t launch_format_tab(const FormatTabArgs& a, uint8_t* out, uint64_t out_capacity, void* workspace,
                             FormatResult* res, hipStream_t s);

uint64_t format_workspace_size(uint64_t nrec);
hipError_t launch_format_seq(const FormatSeqArgs& a, uint8_t* out, uint64_t out_capacity, void* workspace,
                             FormatResult* res, hipStream_t s);

// exclusive scan of n uint64 values (format.hip): out[i] = v[0] + ... + v[i-1], *total = the sum.
// `part` holds scan64_partials(n) values of scratch.
uint64_t scan64_partials(uint64_t n);
hipError_t launch_scan64(const uint64_t* v, uint64_t n, uint64_t* out, uint64_t* part, uint64_t* total, hipStream_t s);

// ---- gap removal (gaps.hip) -------------------------------------------------------------------

uint64_t remove_gaps_workspace_size(uint64_t nrec, uint64_t total);
// out planes need room for `total` bytes; out_offs nrec + 1 entries; *result = bytes kept.  hint:
// device scratch for the counting pass.
hipError_t launch_remove_gaps(const LaunchCfg& cfg, const uint8_t* seq, const uint8_t* qual, const uint64_t* offs,
                              uint64_t nrec, uint64_t total, const Table256& is_gap, uint8_t* out_seq, uint8_t* out_qual,
                              uint64_t* out_offs, void* workspace, uint64_t* result, BatchHint* hint, hipStream_t s);

// ---- subsequences (slices.hip) --------------------------------------------------------------

enum : uint32_t { SUBSEQ_OK = 0, SUBSEQ_ERECORD = 1, SUBSEQ_ECAPACITY = 2 };
struct Slice { uint64_t record; int64_t start, end; uint32_t minus, pad; };             // mirrors skh_slice
struct SubseqResult { uint64_t out_bytes; uint32_t status, pad; uint64_t bad_slice; };  // mirrors skh_subseq_result

uint64_t subseq_workspace_size(uint64_t n);
// slices == nullptr: slice i = record i with the uniform (start, end, minus); n must equal nrec then.
// lut: complement table applied to minus-strand slices (nullptr: reverse only).  qual/out_qual may be nullptr.
hipError_t launch_subseq(const LaunchCfg& cfg, const uint8_t* seq, const uint8_t* qual, const uint64_t* offs,
                         uint64_t nrec, uint64_t total_in, const Slice* slices, uint64_t n, int64_t start, int64_t end, uint32_t minus,
                         const Table256* lut, uint8_t* out_seq, uint8_t* out_qual, uint64_t capacity,
                         uint64_t* out_offs, void* workspace, SubseqResult* res, hipStream_t st);

}  // namespace skh
