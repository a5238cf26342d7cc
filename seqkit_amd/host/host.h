// host.h -- host side of the `seqkit-hip` driver: the SeqKit commands `seq`, `fx2tab`,
// `stats`, `subseq` and `convert` (flags as in the reference: seqkit/cmd/seq.go:291-306, fx2tab.go:157-164,
// stat.go:292-294, subseq.go:549-564, convert.go:280-291, root.go:65-72) with the per-record arithmetic replaced by batch calls
// into libseqkit_hip.so.  The reader packs records into pinned CSR batches; the GPU does
// reverse / complement / counting; everything textual (wrapping, %.2f, tables) is done here
// from the integer results.
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <regex>
#include <string>
#include <vector>

#include "seqkit_hip.h"

namespace host {

// ---- alphabets (what the reference's reader guesses from the first record) -------------
enum Alphabet { DNA = 0, DNAredundant = 1, RNA = 2, RNAredundant = 3, Protein = 4, Unlimit = 5, Auto = -1 };
const char* alphabet_name(int a);
// subset test of the distinct bytes of seq[0..n) (first `threshold` bytes, 0 = all) against
// DNA, RNA, DNAredundant, RNAredundant, Protein in that order; DNA/RNA promoted to redundant
int guess_alphabet(const uint8_t* seq, size_t n, size_t threshold);

// ---- output sink: file, stdout, or gzip (suffix .gz), buffered --------------------------
class Sink {
public:
    explicit Sink(const std::string& path);
    ~Sink();
    bool ok() const { return ok_; }
    void write(const void* p, size_t n);
    void put(char c) { if (len_ == cap_) flush(); buf_[len_++] = c; }
    void puts(const char* s);
    void flush();
private:
    FILE* fp_ = nullptr; void* gz_ = nullptr; bool ok_ = false;
    char* buf_; size_t len_ = 0, cap_ = 1 << 22;
};

// ---- one batch of records: CSR planes in pinned memory + names ---------------------------
struct Batch {
    skh_batch* b = nullptr;
    uint64_t nrec = 0;
    std::vector<char> names;            // head lines back to back
    std::vector<uint64_t> name_off;     // nrec + 1
    std::vector<uint32_t> id_len;       // length of the ID prefix of each name
    std::vector<uint64_t> counts;       // results of a counting pass
    uint8_t* seq() const { return skh_batch_seq(b); }
    uint8_t* qual() const { return skh_batch_qual(b); }
    uint64_t* offs() const { return skh_batch_offsets(b); }
    uint64_t bytes() const { return offs()[nrec]; }
};

// ---- FASTA/FASTQ reader (plain or gzip, "-" = stdin) filling batches ----------------------
class FastxReader {
public:
    FastxReader(skh_ctx* ctx, const std::string& path, int forced_alphabet, size_t guess_len);
    ~FastxReader();
    bool ok() const { return fh_ != nullptr; }
    const std::string& error() const { return err_; }
    // Fills `bt` (allocating / growing its pinned batch as needed) with up to the batch limits.
    // Returns false when no record is left.  Parse errors set error().
    bool next_batch(Batch& bt, uint64_t max_bytes, uint64_t max_records);
    bool is_fastq() const { return is_fastq_; }
    int alphabet() const { return alphabet_; }     // -1 until the first record was seen
private:
    bool fill();                                     // refill the input buffer
    bool read_line(const uint8_t*& p, size_t& n);    // next line without EOL; false at EOF
    void unread_line() { have_line_ = true; }
    skh_ctx* ctx_;
    void* fh_ = nullptr;                             // gzFile
    std::vector<uint8_t> buf_; size_t pos_ = 0, end_ = 0; bool eof_ = false;
    std::vector<uint8_t> carry_;                     // a line that straddles two buffer fills
    const uint8_t* line_ = nullptr; size_t line_len_ = 0; bool have_line_ = false;
    bool format_known_ = false, is_fastq_ = false;
    int alphabet_ = -1, forced_alphabet_; size_t guess_len_;
    std::string err_;
};

// grows the pinned batch of `bt` so it can hold the given totals, preserving its first `used`
// bytes and its offsets
bool ensure_capacity(skh_ctx* ctx, Batch& bt, uint64_t used, uint64_t bytes, uint64_t records, bool with_qual,
                     std::string& err);

// ---- formatting helpers ------------------------------------------------------------------
void fmt_fixed(double v, int prec, std::string& out);      // Go "%.<prec>f" incl. NaN spelling
std::string comma(int64_t v);                                // humanize.Comma
std::string commaf(double v);                                // humanize.Commaf
double round_to(double f, int n);                            // util/math.Round
void write_wrapped(Sink& out, const uint8_t* s, size_t n, int width);   // byteutil.WrapByteSlice

// ---- commands -----------------------------------------------------------------------------
struct GlobalOpts {
    int alphabet = Auto;          // -t
    int threads = 2;              // -j (accepted; the GPU path does not use it)
    int line_width = 60;          // -w
    std::string out_file = "-";   // -o
    bool quiet = false;
    size_t guess_len = 10000;     // --alphabet-guess-seq-length
    int device = 0;               // --device (extension: which GPU)
    std::vector<int> devices;     // --gpus N (devices device .. device+N-1) or --devices a,b,c: the GPUs the text
                                  // blocks of a file are dealt to in turn (always at least one entry)
    uint64_t batch_bytes = 256ull << 20, batch_records = 4ull << 20;
    uint64_t text_block_bytes = 32ull << 20;   // text path: file bytes per block (--batch-bytes overrides)
    // --id-regexp / --id-ncbi (root.go:68-69, helper.go:125-135): the record ID is capture group 1 of the
    // expression matched against the head line, the whole head line when it does not match.  Empty: the
    // default `^([^\s]+)\s?`, which needs no regular-expression engine (head line up to the first white space).
    std::string id_regexp;
    std::shared_ptr<std::regex> id_re;
    bool host_parse = false;      // --host-parse: records are scanned and printed by the host
                                  // (needed for multi-line FASTQ); default: file text is parsed
                                  // and, for seq, formatted on the GPU
};
struct SeqOpts { bool reverse = false, complement = false, only_name = false, only_seq = false,
                 only_qual = false, only_id = false; int min_len = -1, max_len = -1;
                 bool validate = false;            // -v, or implied by -t (seq.go:84-93)
                 uint64_t validate_len = 10000;    // -V: bytes of each record that are checked (0 = all)
                 bool remove_gaps = false; std::string gap_letters = "- .";     // -g -G (seq.go:196-198)
                 bool lower_case = false, upper_case = false, dna2rna = false, rna2dna = false; };   // seq.go:200-241
struct Fx2tabOpts { bool only_id = false, length = false, gc = false, gc_skew = false, only_name = false,
                    header_line = false, alphabet_col = false; std::vector<std::string> base; };
struct StatsOpts { bool all = false, tabular = false; std::string gap_letters = "- ."; };

struct SubseqOpts { std::string region, gtf, bed, gtf_tag = "gene_id"; std::vector<std::string> chr, feature;
                    int up_stream = 0, down_stream = 0; bool only_flank = false; };

struct ConvertOpts { int from = 0, to = SKH_Q_SANGER; bool dry_run = false, force = false; int nrecords = 1000,
                     n_most_common = 4; double illumina15_frac = 0.1; };

int cmd_seq(const GlobalOpts& g, const SeqOpts& o, const std::vector<std::string>& files);
int cmd_convert(const GlobalOpts& g, const ConvertOpts& o, const std::vector<std::string>& files);
int cmd_subseq(const GlobalOpts& g, const SubseqOpts& o, const std::vector<std::string>& files);
int cmd_fx2tab(const GlobalOpts& g, const Fx2tabOpts& o, const std::vector<std::string>& files);
int cmd_stats(const GlobalOpts& g, const StatsOpts& o, const std::vector<std::string>& files);

// "[ERRO] msg" on stderr and the reference's exit status (os.Exit(-1) -> 255)
int fatal(const std::string& msg);

// Where the ID lies inside a head line: [*off, *off + *len) of name[0, n).
void id_span(const GlobalOpts& g, const uint8_t* name, uint32_t n, uint32_t* off, uint32_t* len);

}  // namespace host
