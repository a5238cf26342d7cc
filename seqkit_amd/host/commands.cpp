// commands.cpp -- `seq`, `fx2tab`, `stats`, `subseq`, `convert` on top of the C ABI.
//
// Each command mirrors the record loop of its Go counterpart (seqkit/cmd/seq.go:114-280,
// fx2tab.go:95-150, stat.go:86-161, subseq.go, convert.go) but works a block of file text at a
// time: the text is parsed, transformed or counted, and formatted on the GPU (skh_text_*), with
// a reader thread, a parser thread and the command itself working on different blocks at once
// (BlockPipe).  With --host-parse the host scans and prints the records instead and only the
// per-record work goes to the GPU, a batch at a time: while the GPU transforms or counts batch i,
// the host formats batch i-1 and parses batch i+1 (two pinned batches, skh_*_submit /
// skh_batch_wait).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <atomic>
#include <climits>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <type_traits>

#include "host.h"

namespace host {

void drop_cached_blocks();          // text blocks kept between the files of a command (BlockPipe)

namespace {

// The contexts of a command: one per entry of --devices / --gpus (c = the first, which also serves
// everything that is not dealt out: --host-parse batches, the guess pass of convert).
struct Ctx {
    skh_ctx* c = nullptr;
    std::vector<skh_ctx*> all;
    skh_comm* comm = nullptr;                 // made on first use (stats over several GPUs)
    skh_multi* multi = nullptr;               // several GPUs: pinned batches are cut over all of them (skh_*_multi)
    int init(const GlobalOpts& g)
    {
        for (int d : g.devices) {
            skh_ctx* x = nullptr;
            if (skh_init(d, &x) != SKH_OK) return -1;
            all.push_back(x);
        }
        c = all[0];
        if (all.size() > 1 && skh_multi_init(all.data(), (int)all.size(), &multi) != SKH_OK) return -1;
        return 0;
    }
    // The communicators of `stats` over several GPUs.  RCCL may announce itself on standard output while
    // one is made, and standard output is the command's data: file descriptor 1 points at standard error
    // meanwhile (stats prints its table after the last file: nothing else writes to it at this point).
    int make_comms()
    {
        if (all.size() < 2 || comm) return 0;
        fflush(stdout);
        const int saved = dup(1);
        if (saved >= 0) (void)dup2(2, 1);
        int rc = skh_comm_init_all(all.data(), (int)all.size(), &comm);
        if (rc == SKH_OK && multi) rc = skh_multi_comm_init(multi);
        fflush(stdout);
        if (saved >= 0) { (void)dup2(saved, 1); close(saved); }
        return rc == SKH_OK ? 0 : -1;
    }
    // SKH_SHARES=1: what every GPU did in the last skh_*_multi call, on stderr
    void report_shares(const char* what) const
    {
        if (!multi || !getenv("SKH_SHARES")) return;
        std::vector<skh_share_info> info((size_t)skh_multi_size(multi));
        if (skh_multi_shares(multi, info.data()) != SKH_OK) return;
        for (size_t k = 0; k < info.size(); k++)
            fprintf(stderr, "[shares] %s: context %zu bytes [%llu, %llu) records %llu..%llu head_cut %llu tail_cut %llu chunks %u\n", what, k,
                    (unsigned long long)info[k].first_byte, (unsigned long long)info[k].last_byte,
                    (unsigned long long)info[k].first_record, (unsigned long long)(info[k].first_record + info[k].nrecords),
                    (unsigned long long)info[k].head_cut, (unsigned long long)info[k].tail_cut, info[k].chunks);
    }
    ~Ctx()
    {
        drop_cached_blocks();
        if (multi) skh_multi_destroy(multi);
        if (comm) skh_comm_destroy(comm);
        for (skh_ctx* x : all) skh_destroy(x);
    }
};

struct TwoBatches {
    Batch bt[2];
    ~TwoBatches() { for (auto& b : bt) if (b.b) skh_batch_free(b.b); }
};

// ---- phase timing (SKH_TIMING=1: totals on stderr at exit) ---------------------------------
struct Phases {
    double t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool on = getenv("SKH_TIMING") != nullptr;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    ~Phases()
    {
        if (on) fprintf(stderr, "[timing] read %.3f s, upload+parse %.3f s, kernels %.3f s, format+download %.3f s, "
                                "write %.3f s, buffer alloc/free %.3f s; pipeline: tail copies %.3f s, reader waits %.3f s, "
                                "parser waits %.3f s, caller waits %.3f s\n", t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9]);
    }
};
Phases g_phases;
struct PhaseTimer {
    int k; double t0;
    explicit PhaseTimer(int k_) : k(k_), t0(g_phases.on ? Phases::now() : 0) {}
    ~PhaseTimer() { if (g_phases.on) g_phases.t[k] += Phases::now() - t0; }
};

// GPU work on a pinned batch, in flight: with one GPU it is a skh_*_submit on the batch's own stream,
// with several a skh_*_multi call (synchronous: all GPUs work on the one batch) on a helper thread,
// so the host can print the batch before while it runs.
struct BatchWork {
    std::future<int> fut;
    std::string err;
    bool active = false;
    template <typename F> void start(F&& call)
    {
        active = true;
        fut = std::async(std::launch::async, [this, call] { const int rc = call(); if (rc != SKH_OK) err = skh_last_error(); return rc; });
    }
    int wait()
    {
        if (!active) return SKH_OK;
        active = false;
        return fut.get();
    }
    ~BatchWork() { if (active) (void)fut.get(); }
};

// A plain FASTA file whose first record is longer than a text block (a chromosome, say).  With several
// GPUs such a file goes through pinned batches, where ONE record is cut over all GPUs (skh_*_multi);
// as text blocks a record lives on the one GPU its block was dealt to.
bool first_record_exceeds_block(const std::string& path, uint64_t block_bytes)
{
    if (path == "-") return false;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    bool yes = false;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && (uint64_t)st.st_size > block_bytes) {
        std::vector<uint8_t> buf(block_bytes);
        uint64_t have = 0;
        while (have < block_bytes) {
            const ssize_t n = pread(fd, buf.data() + have, (size_t)(block_bytes - have), (off_t)have);
            if (n <= 0) break;
            have += (uint64_t)n;
        }
        uint64_t k = 0;
        while (k < have && (buf[k] == '\n' || buf[k] == '\r')) k++;
        if (have == block_bytes && k < have && buf[k] == '>') {
            yes = true;
            for (const uint8_t* p = buf.data() + k + 1; yes;) {
                p = static_cast<const uint8_t*>(memchr(p, '>', (size_t)(buf.data() + have - p)));
                if (!p) break;
                if (p[-1] == '\n') yes = false;
                p++;
            }
        }
    }
    close(fd);
    return yes;
}

// Reads `want` bytes at `pos` of a regular file: the stretch is cut into equal parts read with
// pread(2) by a few threads at once (one thread copies page-cache pages at 10-25 GB/s, a PCIe
// link takes 50).  Returns the bytes read from `pos` on without a hole, or -1.
long pread_parallel(int fd, uint8_t* dst, uint64_t want, uint64_t pos)
{
    unsigned nthr = std::thread::hardware_concurrency();
    if (const char* e = getenv("SKH_READ_THREADS")) nthr = (unsigned)atoi(e);
    if (nthr > 8) nthr = 8;
    if (nthr < 1 || want < (8u << 20)) nthr = 1;
    const uint64_t part = ((want + nthr - 1) / nthr + 4095) & ~uint64_t(4095);
    std::vector<long> got(nthr, 0);
    auto work = [&](unsigned k) {
        uint64_t a = (uint64_t)k * part, b = a + part < want ? a + part : want;
        while (a < b) {
            const ssize_t n = pread(fd, dst + a, (size_t)(b - a), (off_t)(pos + a));
            if (n < 0) { got[k] = -1; return; }
            if (n == 0) break;                       // the end of the file
            a += (uint64_t)n; got[k] += (long)n;
        }
    };
    std::vector<std::thread> thr;
    for (unsigned k = 1; k < nthr && (uint64_t)k * part < want; k++) thr.emplace_back(work, k);
    work(0);
    for (auto& t : thr) t.join();
    long total = 0;
    for (unsigned k = 0; k < nthr; k++) {
        if (got[k] < 0) return -1;
        total += got[k];
        const uint64_t a = (uint64_t)k * part, b = a + part < want ? a + part : want;
        if (a < want && (uint64_t)got[k] < b - a) break;    // short part: what follows it is not contiguous
    }
    return total;
}

// ---- multi-line FASTQ ------------------------------------------------------------------------
//
// The reference's reader accepts FASTQ records whose sequence and quality are wrapped over several
// lines (README.md:88, doc/docs/usage.md:357-359): sequence lines run up to the first line that
// starts with '+', quality lines follow until they hold as many bytes as the sequence.  Where a
// record ends then depends on every record before it ('@' may start a quality line), so such text
// cannot be cut into records by a data-parallel scan.  The GPU parser reports it as a format
// error; the block is then rewritten here, serially, into the four-line form of the same records
// and parsed again, and so is every later block of the file.  Everything after the parse (kernels,
// output formatting) is the same path as for four-line files.
struct FastqNorm {
    uint64_t nrec = 0;          // complete records rewritten
    uint64_t consumed = 0;      // bytes of the input they span
    uint64_t lines = 0;         // input lines they span
    int64_t error_line = -1;    // 0-based input line of a format violation, or -1
    bool wrapped = false;       // some record was not in the plain four-line form (sequence or quality on several
                                // lines or on none, blank lines before a record)
};

// text[0, n) starts between records.  final: the text ends where the file ends.  The four-line
// form is appended to `out` (never more than n + 2 bytes per record).
FastqNorm normalize_fastq(const uint8_t* text, uint64_t n, bool final, std::vector<uint8_t>& out)
{
    FastqNorm r;
    out.clear();
    uint64_t p = 0, line_no = 0;
    // next line without its line end; false: no complete line left
    const uint8_t* lp = nullptr; uint64_t ll = 0;
    auto next_line = [&]() -> bool {
        if (p >= n) return false;
        const uint8_t* nl = static_cast<const uint8_t*>(memchr(text + p, '\n', n - p));
        uint64_t end;
        if (nl) end = (uint64_t)(nl - text);
        else if (final) end = n;
        else return false;
        lp = text + p; ll = end - p;
        while (ll && lp[ll - 1] == '\r') ll--;
        p = nl ? end + 1 : n;
        line_no++;
        return true;
    };
    for (;;) {
        const uint64_t rec_out = out.size();
        bool got = false, blank = false;
        while (next_line()) { if (ll) { got = true; break; } blank = true; }      // blank lines between records
        if (got && blank) r.wrapped = true;
        if (!got) {
            if (p >= n && final) { r.consumed = n; r.lines = line_no; }
            out.resize(rec_out);
            return r;
        }
        if (lp[0] != '@') { r.error_line = (int64_t)line_no - 1; out.resize(rec_out); return r; }
        out.insert(out.end(), lp, lp + ll);
        out.push_back('\n');
        uint64_t sl = 0, ql = 0, seq_lines = 0, qual_lines = 0;
        bool cut = false;
        while (true) {
            if (!next_line()) { cut = !(final && p >= n); break; }
            if (ll && lp[0] == '+') break;
            out.insert(out.end(), lp, lp + ll);
            sl += ll; seq_lines++;
        }
        if (cut) { out.resize(rec_out); return r; }
        out.push_back('\n'); out.push_back('+'); out.push_back('\n');
        while (ql < sl) {
            if (!next_line()) { cut = !(final && p >= n); break; }
            out.insert(out.end(), lp, lp + ll);
            ql += ll; qual_lines++;
        }
        if (cut) { out.resize(rec_out); return r; }
        if (ql != sl) { r.error_line = (int64_t)line_no - 1; out.resize(rec_out); return r; }
        out.push_back('\n');
        if (seq_lines != 1 || qual_lines != 1) r.wrapped = true;
        r.nrec++; r.consumed = p; r.lines = line_no;
    }
}

// Wrapped FASTQ is parsed on the GPU (SKH_FASTQ_WRAPPED) once the four-line parser has refused a block; the
// serial rewrite above remains for what that parser refuses too -- it names the offending line.
// SKH_FASTQ_REWRITE=host: the serial rewrite only (as before the GPU form existed); =off: never the serial
// rewrite (format errors of wrapped text then come without a line number).
enum { REWRITE_AUTO = 0, REWRITE_HOST = 1, REWRITE_OFF = 2 };
static int fastq_rewrite_mode()
{
    static const int mode = [] {
        const char* v = getenv("SKH_FASTQ_REWRITE");
        if (v && !strcmp(v, "host")) return (int)REWRITE_HOST;
        if (v && !strcmp(v, "off")) return (int)REWRITE_OFF;
        return (int)REWRITE_AUTO;
    }();
    return mode;
}

// ---- file text in blocks, parsed on the GPU ------------------------------------------------
//
// Reads a file (plain or gzip, "-" = stdin) block by block into the pinned buffer of a text
// block, parses each block on the GPU and hands it to `on_block`; the unconsumed tail of a
// block (its last, incomplete record) is moved to the front of the next one.  The format is
// decided by the first non-blank byte; leading blank lines are dropped.
class TextBlocks {
public:
    TextBlocks(skh_ctx* ctx, const std::string& path, uint64_t block_bytes) : ctx_(ctx), path_(path)
    {
        // plain files are read with read(2) straight into the pinned block; gzip (and stdin) through zlib
        if (path != "-") {
            fd_ = open(path.c_str(), O_RDONLY);
            if (fd_ < 0) { err_ = "open " + path + ": no such file or directory"; return; }
            unsigned char magic[2] = {0, 0};
            const ssize_t got = pread(fd_, magic, 2, 0);
            if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) { fh_ = gzdopen(fd_, "rb"); fd_ = -1; }
            struct stat st;
            if (fd_ >= 0 && fstat(fd_, &st) == 0 && S_ISREG(st.st_mode)) file_size_ = (uint64_t)st.st_size;
        } else {
            fh_ = gzdopen(0, "rb");
        }
        if (fd_ < 0 && !fh_) { err_ = "open " + path + ": no such file or directory"; return; }
        if (fh_) gzbuffer((gzFile)fh_, 1 << 20);
        PhaseTimer pt(5);
        if (skh_text_alloc(ctx, block_bytes, &t_) != SKH_OK) err_ = skh_last_error();
    }
    ~TextBlocks() {
        PhaseTimer pt(5); if (fh_) gzclose((gzFile)fh_); if (fd_ >= 0) close(fd_); if (t_) skh_text_free(t_); }
    bool ok() const { return err_.empty(); }
    const std::string& error() const { return err_; }
    bool is_fastq() const { return fastq_; }
    bool any_record() const { return any_; }
    skh_text* block() const { return t_; }

    // Parses the next block.  Returns false at the end of the file or on error (see error()).
    bool next(skh_parse_result& res)
    {
        if (!ok() || done_) return false;
        for (;;) {
            uint8_t* buf = skh_text_buffer(t_);
            const uint64_t cap = skh_text_capacity(t_);
            while (have_ < cap && !eof_) {
                PhaseTimer pt(0);
                const uint64_t want = cap - have_ > (1u << 30) ? (1u << 30) : cap - have_;
                const long n = fh_ ? (long)gzread((gzFile)fh_, buf + have_, (unsigned)want)
                             : file_size_ != UINT64_MAX ? read_regular(buf + have_, want)
                                                        : (long)read(fd_, buf + have_, (size_t)want);
                if (n < 0) { err_ = "read " + path_ + ": input error"; return false; }
                if (n == 0) eof_ = true;
                have_ += (uint64_t)n;
            }
            if (!format_known_) {
                uint64_t k = 0;
                while (k < have_ && (buf[k] == '\n' || buf[k] == '\r')) k++;
                if (k == have_ && !eof_) { have_ = 0; continue; }          // a block of blank lines
                if (k) { memmove(buf, buf + k, have_ - k); have_ -= k; }
                if (have_ == 0) { done_ = true; return false; }            // empty file
                if (buf[0] == '>') fastq_ = false;
                else if (buf[0] == '@') fastq_ = true;
                else { err_ = "error when parsing " + path_ + ": invalid FASTA/Q format"; return false; }
                format_known_ = true;
                if (have_ < cap && !eof_) continue;                        // top the block up again
            }
            PhaseTimer pt(1);
            const bool was_multiline = multiline_;
            block_wrapped_ = false;
            if (!multiline_) {
                const int fmt = fastq_ ? (wrapped_gpu_ ? SKH_FASTQ_WRAPPED : SKH_FASTQ) : SKH_FASTA;
                if (skh_text_parse(ctx_, t_, have_, fmt, eof_, &res) != SKH_OK) {
                    err_ = skh_last_error();
                    return false;
                }
                bool refused = fastq_ && (res.status == SKH_PARSE_EFORMAT || (res.nrec == 0 && (buf[0] == '\n' || buf[0] == '\r')));
                if (fmt == SKH_FASTQ_WRAPPED && res.status == SKH_PARSE_OK) { block_wrapped_ = true; refused = false; }
                else if (refused && fmt == SKH_FASTQ && fastq_rewrite_mode() != REWRITE_HOST) {
                    // records wrapped over several lines, blank lines between records: the GPU's other parser
                    skh_parse_result r2;
                    if (skh_text_parse(ctx_, t_, have_, SKH_FASTQ_WRAPPED, eof_, &r2) != SKH_OK) { err_ = skh_last_error(); return false; }
                    if (r2.status == SKH_PARSE_OK) { res = r2; block_wrapped_ = true; refused = false; }
                }
                if (block_wrapped_) wrapped_gpu_ = res.wrapped != 0 || res.nrec == 0;      // plain blocks go back to the fast parser
                if (refused && fastq_rewrite_mode() != REWRITE_OFF) multiline_ = true;
                else if (refused || res.status == SKH_PARSE_EFORMAT) {
                    err_ = "error when parsing " + path_ + ": invalid FASTA/Q format" +
                           (res.error_line >= 0 ? " near line " + std::to_string(lines_before_ + (uint64_t)res.error_line + 1) : std::string());
                    return false;
                }
            }
            bool no_record = !multiline_ && res.nrec == 0;
            if (multiline_) {
                // wrapped FASTQ records: the block's text is kept aside, its four-line form is parsed
                orig_.assign(buf, buf + have_);
                const FastqNorm nr = normalize_fastq(orig_.data(), have_, eof_, norm_);
                if (!was_multiline && !nr.wrapped && res.status == SKH_PARSE_EFORMAT && res.error_line >= 0) {
                    // plain four-line text as far as the serial scan got: the GPU parser's finding stands
                    err_ = "error when parsing " + path_ + ": invalid FASTA/Q format near line " +
                           std::to_string(lines_before_ + (uint64_t)res.error_line + 1);
                    return false;
                }
                if (nr.error_line >= 0) {
                    err_ = "error when parsing " + path_ + ": invalid FASTA/Q format near line " +
                           std::to_string(lines_before_ + (uint64_t)nr.error_line + 1);
                    return false;
                }
                no_record = nr.nrec == 0;
                if (!no_record && norm_.size() <= cap) {
                    memcpy(buf, norm_.data(), norm_.size());
                    if (skh_text_parse(ctx_, t_, norm_.size(), SKH_FASTQ, 1, &res) != SKH_OK) { err_ = skh_last_error(); return false; }
                    if (res.status != SKH_PARSE_OK) { err_ = "error when parsing " + path_ + ": invalid FASTA/Q format"; return false; }
                    res.consumed = nr.consumed; res.lines = nr.lines;
                } else if (!no_record) {
                    no_record = true;                              // the rewritten text needs a larger block
                    eof_hold_ = true;
                } else {
                    memset(&res, 0, sizeof res);
                }
            }
            if (no_record && (!eof_ || eof_hold_)) {
                // not one complete record in a full block: double the block and read on
                eof_hold_ = false;
                skh_text* bigger = nullptr;
                if (skh_text_alloc(ctx_, cap * 2, &bigger) != SKH_OK) { err_ = skh_last_error(); return false; }
                memcpy(skh_text_buffer(bigger), multiline_ ? orig_.data() : buf, have_);
                skh_text_free(t_);
                t_ = bigger;
                continue;
            }
            consumed_ = res.consumed;
            if (eof_) done_ = true;
            if (res.nrec) any_ = true;
            return res.nrec != 0 || !done_;
        }
    }
    bool at_eof() const { return eof_; }
    // Instead of advance(): the block just parsed is given up and the next call of next() parses a
    // block twice as large that starts at the same place (for callers that need more records at once).
    bool grow()
    {
        skh_text* bigger = nullptr;
        if (skh_text_alloc(ctx_, skh_text_capacity(t_) * 2, &bigger) != SKH_OK) { err_ = skh_last_error(); return false; }
        memcpy(skh_text_buffer(bigger), multiline_ ? orig_.data() : skh_text_buffer(t_), have_);
        skh_text_free(t_);
        t_ = bigger;
        consumed_ = 0;
        done_ = false;
        return true;
    }
    // Call after the block has been used: keeps its unconsumed tail for the next one.
    void advance(const skh_parse_result& res)
    {
        uint8_t* buf = skh_text_buffer(t_);
        lines_before_ += (fastq_ && !multiline_ && !block_wrapped_) ? res.nrec * 4 : res.lines;      // for error messages ("near line")
        if (multiline_) memcpy(buf, orig_.data() + consumed_, have_ - consumed_);  // the buffer holds the rewritten text
        else memmove(buf, buf + consumed_, have_ - consumed_);
        have_ -= consumed_;
        consumed_ = 0;
    }

private:
    long read_regular(uint8_t* dst, uint64_t want)
    {
        const long n = pread_parallel(fd_, dst, want, file_pos_);
        if (n > 0) file_pos_ += (uint64_t)n;
        return n;
    }

    skh_ctx* ctx_; std::string path_, err_;
    void* fh_ = nullptr; int fd_ = -1; skh_text* t_ = nullptr;
    uint64_t file_size_ = UINT64_MAX, file_pos_ = 0;       // regular plain files only
    uint64_t have_ = 0, consumed_ = 0, lines_before_ = 0;
    bool eof_ = false, done_ = false, format_known_ = false, fastq_ = false, any_ = false;
    bool multiline_ = false, eof_hold_ = false;            // FASTQ with wrapped records (normalize_fastq)
    bool wrapped_gpu_ = false, block_wrapped_ = false;     // ... parsed on the GPU: the next block starts with that parser / this block was
    std::vector<uint8_t> orig_, norm_;                     // the block's own text / its four-line form
};

// ---- the same, as a pipeline -----------------------------------------------------------------
//
// Three stages on three threads, a few text blocks going round: the READER fills a block's
// pinned buffer from the file, starting `head` bytes in; the PARSER puts the unconsumed tail of
// the block before in front of that (so a block never waits for its predecessor to be read),
// uploads and parses it on the block's own stream; the CALLER (next / block / advance, as with
// TextBlocks) runs kernels, formats, downloads and writes.  Block k+1 is read while block k is
// uploaded and block k-1 is downloaded: the file system, the PCIe link in both directions and
// the GPU work at once, and the slowest of them sets the pace.
class BlockPipe {
    struct Block {
        skh_text* t = nullptr;
        skh_ctx* ctx = nullptr;          // the GPU the block lives on
        int dev_index = 0;               // its index in the command's device list
        uint64_t cap = 0, head = 0;      // buffer size; where the reader starts to fill
        uint64_t filled = 0;             // bytes the reader put at buffer + head
        bool eof = false;                // the file ended in (or right after) this block
        bool grown = false;              // enlarged for long records: half of it is headroom
        bool head_line = false;          // FASTA: what the reader put in holds a line that starts with '>'
        uint64_t ordinary_for = 0;       // the block size it was allocated for (0: enlarged); such blocks are kept for the next file
        skh_parse_result res;
    };
    // a queue of block pointers; nullptr from pop() means "closed and empty"
    struct Queue {
        std::mutex mu; std::condition_variable cv; std::deque<Block*> q; bool closed = false;
        void push(Block* b) { { std::lock_guard<std::mutex> l(mu); q.push_back(b); } cv.notify_all(); }
        void close() { { std::lock_guard<std::mutex> l(mu); closed = true; } cv.notify_all(); }
        Block* pop()
        {
            std::unique_lock<std::mutex> l(mu);
            cv.wait(l, [&] { return !q.empty() || closed; });
            if (q.empty()) return nullptr;
            Block* b = q.front(); q.pop_front();
            return b;
        }
    };
    // everything the threads touch; outlives the object when the reader has to be left behind
    struct State {
        std::vector<skh_ctx*> ctxs; std::string path;
        size_t nblocks = 4;              // blocks in the round: 4, or two per GPU when that is more
        void* fh = nullptr; int fd = -1; uint64_t file_size = UINT64_MAX, file_pos = 0;
        Queue free_q, filled_q, parsed_q;
        std::vector<Block*> all;
        std::mutex err_mu; std::string err;
        std::atomic<bool> stop{false}, reader_done{false};
        bool fastq = false;              // written by the parser before the first block is handed on
        uint64_t chunk = 0;              // most the reader puts into a block at once (what an ordinary block takes)
        uint64_t block_bytes = 0;        // size of an ordinary block
        uint64_t largest_cap = 0;        // largest enlarged block so far (parser thread only)
        void fail(const std::string& m) { std::lock_guard<std::mutex> l(err_mu); if (err.empty()) err = m; }
        std::string error() { std::lock_guard<std::mutex> l(err_mu); return err; }
    };

    // Headroom in front of what the reader fills: an eighth of a block of the ordinary size (the
    // tail is less than one record), half of a block that was enlarged for records longer than that.
    static uint64_t headroom(const Block& b) { return ((b.grown ? b.cap / 2 : b.cap / 8)) & ~uint64_t(4095); }

    // Blocks of the ordinary size outlive their file: the next file of the command takes them over
    // instead of pinning fresh memory (a few milliseconds per block).
    struct Cache { std::mutex mu; std::vector<Block*> blocks; };
    static Cache& cache() { static Cache c; return c; }

public:
    static void drop_cache()
    {
        Cache& c = cache();
        std::lock_guard<std::mutex> l(c.mu);
        for (Block* b : c.blocks) { skh_text_free(b->t); delete b; }
        c.blocks.clear();
    }

private:
    static Block* new_block(State& s, uint64_t cap, bool grown, int dev_index)
    {
        skh_ctx* ctx = s.ctxs[(size_t)dev_index];
        if (!grown) {
            Cache& c = cache();
            std::lock_guard<std::mutex> l(c.mu);
            for (size_t k = 0; k < c.blocks.size(); k++) {
                if (c.blocks[k]->ordinary_for != cap || c.blocks[k]->ctx != ctx) continue;
                Block* b = c.blocks[k];
                c.blocks.erase(c.blocks.begin() + (long)k);
                b->head = headroom(*b);
                b->filled = 0; b->eof = false; b->head_line = false;
                return b;
            }
        }
        Block* b = new Block;
        b->ordinary_for = grown ? 0 : cap;
        b->ctx = ctx; b->dev_index = dev_index;
        PhaseTimer pt(5);
        if (skh_text_alloc(ctx, cap, &b->t) != SKH_OK) { s.fail(skh_last_error()); delete b; return nullptr; }
        b->cap = skh_text_capacity(b->t);
        b->grown = grown;
        b->head = headroom(*b);
        return b;
    }

    static void reader(std::shared_ptr<State> sp)
    {
        State& s = *sp;
        bool kind_known = false, fasta = false;
        uint8_t prev = 0;                // the byte before the block (0 at the start of the file)
        for (;;) {
            Block* b;
            { PhaseTimer pt(7); b = s.free_q.pop(); }
            if (!b || s.stop) break;
            uint8_t* buf = skh_text_buffer(b->t) + b->head;
            // an enlarged block is not filled further than an ordinary one: what follows the last
            // complete record of a block has to be moved to the next one
            const uint64_t room = b->cap - b->head < s.chunk ? b->cap - b->head : s.chunk;
            b->filled = 0; b->eof = false;
            bool bad = false;
            while (b->filled < room && !b->eof && !s.stop) {
                PhaseTimer pt(0);
                const uint64_t want = room - b->filled > (1u << 30) ? (1u << 30) : room - b->filled;
                long n;
                if (s.fh) n = (long)gzread((gzFile)s.fh, buf + b->filled, (unsigned)want);
                else if (s.file_size != UINT64_MAX) {
                    n = pread_parallel(s.fd, buf + b->filled, want, s.file_pos);
                    if (n > 0) s.file_pos += (uint64_t)n;
                } else n = (long)read(s.fd, buf + b->filled, (size_t)want);
                if (n < 0) { s.fail("read " + s.path + ": input error"); bad = true; break; }
                if (n == 0) b->eof = true;
                b->filled += (uint64_t)n;
            }
            if (bad || s.stop) break;
            if (!kind_known && b->filled) {
                uint64_t k = 0;
                while (k < b->filled && (buf[k] == '\n' || buf[k] == '\r')) k++;
                if (k < b->filled) { kind_known = true; fasta = buf[k] == '>'; }
            }
            b->head_line = true;                                   // FASTQ: the parser decides
            if (kind_known && fasta) {
                // a head line after the first byte of the file: the record before it is complete
                b->head_line = false;
                uint64_t p = 0;
                while (p < b->filled && !b->head_line) {
                    const uint8_t* q = static_cast<const uint8_t*>(memchr(buf + p, '>', b->filled - p));
                    if (!q) break;
                    p = (uint64_t)(q - buf);
                    b->head_line = (p ? buf[p - 1] : prev) == '\n';
                    p++;
                }
            }
            if (b->filled) prev = buf[b->filled - 1];
            // the block starts to cross the link now, while the block before it is parsed (not when
            // it can only be appended to a record that is still growing)
            if (skh_text_upload(b->ctx, b->t, b->head, b->head_line ? b->filled : 0) != SKH_OK) {
                s.fail(skh_last_error());
                break;
            }
            const bool last = b->eof;
            s.filled_q.push(b);
            if (last) break;
        }
        s.filled_q.close();
        s.reader_done = true;
    }

    // memcpy by a few threads (a single one moves 10 GB/s or less, page faults included)
    static void copy_bytes(uint8_t* dst, const uint8_t* src, uint64_t n)
    {
        if (n < (16u << 20)) { memcpy(dst, src, n); return; }
        const unsigned nthr = 8;
        const uint64_t part = ((n + nthr - 1) / nthr + 4095) & ~uint64_t(4095);
        std::vector<std::thread> thr;
        for (unsigned k = 1; k < nthr && (uint64_t)k * part < n; k++) {
            const uint64_t a = (uint64_t)k * part, len = a + part < n ? part : n - a;
            thr.emplace_back([=] { memcpy(dst + a, src + a, len); });
        }
        memcpy(dst, src, part < n ? part : n);
        for (auto& t : thr) t.join();
    }

    // `b` replaced in the round by a new block of at least `cap` bytes (contents are not copied)
    static Block* replace_block(State& s, Block* b, uint64_t cap)
    {
        // a file that made one block grow will make the others grow too: they go to the largest size
        // seen so far at once instead of doubling their way up (every step pins fresh memory)
        if (cap < s.largest_cap) cap = s.largest_cap;
        Block* nb = new_block(s, cap, true, b->dev_index);
        if (!nb) return nullptr;
        if (nb->cap > s.largest_cap) s.largest_cap = nb->cap;
        for (auto& x : s.all) if (x == b) x = nb;
        return nb;
    }
    static void drop_block(Block* b)
    {
        { PhaseTimer pt(5); skh_text_free(b->t); }
        delete b;
    }

    static void parser(std::shared_ptr<State> sp)
    {
        State& s = *sp;
        std::vector<uint8_t> tail;       // text after the last complete record of the block handed on last
        // A block without one complete record (a record longer than the block) is kept here and what
        // the reader delivers next is appended to it, so every byte of a long record is moved once.
        Block* acc = nullptr;
        uint64_t acc_off = 0, acc_have = 0;                 // where its text lies
        bool format_known = false;
        bool multiline = false;                             // FASTQ with wrapped records: blocks are rewritten (normalize_fastq)
        bool wrapped_gpu = false;                           // ... or parsed by the GPU's parser of wrapped FASTQ, from the next block on
        std::vector<uint8_t> norm;
        uint64_t lines_before = 0;
        // the blocks go round in the order they are made: dealt to the GPUs in turn
        for (size_t k = 1; k < s.nblocks && !s.stop; k++) {
            Block* b = new_block(s, s.block_bytes, false, (int)(k % s.ctxs.size()));
            if (!b) { s.stop = true; break; }
            s.all.push_back(b);
            s.free_q.push(b);
        }
        for (;;) {
            Block* b;
            { PhaseTimer pt(8); b = s.filled_q.pop(); }
            if (!b || s.stop) break;
            const double t_tail0 = g_phases.on ? Phases::now() : 0;
            uint64_t off, have;
            const bool eof = b->eof, later_head = b->head_line;
            if (acc) {
                const uint64_t need = acc_off + acc_have + b->filled;
                if (need > acc->cap) {
                    Block* nb = replace_block(s, acc, 2 * (acc_have + b->filled));
                    if (!nb) break;
                    copy_bytes(skh_text_buffer(nb->t), skh_text_buffer(acc->t) + acc_off, acc_have);
                    drop_block(acc);
                    acc = nb; acc_off = 0;
                }
                copy_bytes(skh_text_buffer(acc->t) + acc_off + acc_have, skh_text_buffer(b->t) + b->head, b->filled);
                acc_have += b->filled;
                s.free_q.push(b);                                  // back to the reader at once
                b = acc; off = acc_off; have = acc_have;
                acc = nullptr;
            } else {
                if (tail.size() > b->head && tail.size() + b->filled <= b->cap) {
                    // the tail does not fit in front of what was read, but the block holds both
                    memmove(skh_text_buffer(b->t) + tail.size(), skh_text_buffer(b->t) + b->head, b->filled);
                    b->head = tail.size();
                    skh_text_upload(b->ctx, b->t, 0, 0);           // what the reader sent ahead lies elsewhere now
                } else if (tail.size() > b->head) {
                    // both go into a block of twice the size (half of it headroom), which stays in the round
                    Block* nb = replace_block(s, b, 2 * (tail.size() + b->filled));
                    if (!nb) break;
                    nb->head = tail.size();
                    copy_bytes(skh_text_buffer(nb->t) + nb->head, skh_text_buffer(b->t) + b->head, b->filled);
                    nb->filled = b->filled; nb->eof = b->eof;
                    drop_block(b);
                    b = nb;
                }
                off = b->head - tail.size(); have = tail.size() + b->filled;
                if (!tail.empty()) copy_bytes(skh_text_buffer(b->t) + off, tail.data(), tail.size());
                tail.clear();
            }
            b->head = headroom(*b);                                // for its next round with the reader
            uint8_t* buf = skh_text_buffer(b->t);
            if (g_phases.on) g_phases.t[6] += Phases::now() - t_tail0;
            if (have == 0) {                                       // the file ended with the block before
                if (eof) break;
                s.free_q.push(b);
                continue;
            }
            if (!format_known) {
                uint64_t k = 0;
                while (k < have && (buf[off + k] == '\n' || buf[off + k] == '\r')) k++;
                off += k; have -= k;
                if (have == 0) {                                   // blank lines only so far
                    if (eof) break;
                    s.free_q.push(b);
                    continue;
                }
                if (buf[off] == '>') s.fastq = false;
                else if (buf[off] == '@') s.fastq = true;
                else { s.fail("error when parsing " + s.path + ": invalid FASTA/Q format"); break; }
                format_known = true;
            }
            // FASTA before the end of the file: a record is complete when a later head line is in sight;
            // without one the upload and the parse would be for nothing
            if (!s.fastq && !eof && !later_head) {
                acc = b; acc_off = off; acc_have = have;
                continue;
            }
            skh_parse_result res;
            bool rewrite = multiline;                // this block goes through normalize_fastq
            bool block_wrapped = false;              // ... or was parsed by the GPU's parser of wrapped FASTQ
            if (!multiline) {
                PhaseTimer pt(1);
                const int fmt = s.fastq ? (wrapped_gpu ? SKH_FASTQ_WRAPPED : SKH_FASTQ) : SKH_FASTA;
                if (skh_text_parse_at(b->ctx, b->t, off, have, fmt, eof, &res) != SKH_OK) {
                    s.fail(skh_last_error());
                    break;
                }
                // (the host's copy of the text is intact)  A block that starts with a blank line holds no
                // four-line record for the four-line parser either; wrapped records and blank lines between
                // records go to the GPU's parser of wrapped FASTQ, and what that one refuses to the serial scan,
                // which names the line.
                bool refused = s.fastq && (res.status == SKH_PARSE_EFORMAT ||
                                           (res.nrec == 0 && (buf[off] == '\n' || buf[off] == '\r')));
                if (fmt == SKH_FASTQ_WRAPPED && res.status == SKH_PARSE_OK) { block_wrapped = true; refused = false; }
                else if (refused && fmt == SKH_FASTQ && fastq_rewrite_mode() != REWRITE_HOST) {
                    skh_parse_result r2;
                    if (skh_text_parse_at(b->ctx, b->t, off, have, SKH_FASTQ_WRAPPED, eof, &r2) != SKH_OK) { s.fail(skh_last_error()); break; }
                    if (r2.status == SKH_PARSE_OK) { res = r2; block_wrapped = true; refused = false; }
                }
                if (block_wrapped) wrapped_gpu = res.wrapped != 0 || res.nrec == 0;    // plain blocks go back to the fast parser
                if (refused) {
                    if (fastq_rewrite_mode() != REWRITE_OFF) rewrite = true;
                    else {
                        s.fail("error when parsing " + s.path + ": invalid FASTA/Q format" +
                               (res.error_line >= 0 ? " near line " + std::to_string(lines_before + (uint64_t)res.error_line + 1) : std::string()));
                        break;
                    }
                }
            }
            uint64_t consumed = 0, lines = 0;
            if (rewrite) {
                const FastqNorm nr = normalize_fastq(buf + off, have, eof, norm);
                if (!multiline && !nr.wrapped && res.status == SKH_PARSE_EFORMAT && res.error_line >= 0) {
                    // plain four-line text as far as the serial scan got: the GPU parser's finding stands, a
                    // malformed record does not send the rest of the file through the serial rewrite
                    s.fail("error when parsing " + s.path + ": invalid FASTA/Q format near line " +
                           std::to_string(lines_before + (uint64_t)res.error_line + 1));
                    break;
                }
                if (nr.error_line >= 0 || (eof && nr.consumed != have)) {
                    s.fail("error when parsing " + s.path + ": invalid FASTA/Q format near line " +
                           std::to_string(lines_before + (uint64_t)(nr.error_line >= 0 ? nr.error_line : (int64_t)nr.lines) + 1));
                    break;
                }
                if (nr.nrec == 0) {
                    if (eof) break;
                    multiline = true;
                    acc = b; acc_off = off; acc_have = have;       // not one whole record yet: more text is needed
                    continue;
                }
                // the blocks after one that was four-line text throughout go back to the GPU parser
                multiline = nr.wrapped;
                consumed = nr.consumed; lines = nr.lines;
            } else {
                if (res.status == SKH_PARSE_EFORMAT) {
                    s.fail("error when parsing " + s.path + ": invalid FASTA/Q format near line " +
                           std::to_string(lines_before + (uint64_t)res.error_line + 1));
                    break;
                }
                if (res.nrec == 0) {
                    if (eof) break;
                    acc = b; acc_off = off; acc_have = have;       // nothing complete yet: more text is needed
                    continue;
                }
                consumed = res.consumed; lines = (s.fastq && !block_wrapped) ? res.nrec * 4 : res.lines;
            }
            if (!eof) {
                PhaseTimer pt(6);
                const uint64_t keep = have - consumed;
                // (capacity doubles: a fresh allocation of this size is paid for in page faults)
                if (keep > tail.capacity()) { tail.clear(); tail.shrink_to_fit(); tail.reserve(2 * keep); }
                tail.resize(keep);
                copy_bytes(tail.data(), buf + off + consumed, keep);
            }
            if (rewrite) {
                // the four-line form of the block's records takes the place of its text and is parsed
                if (norm.size() > b->cap) {                        // (records without any sequence line grow by two bytes)
                    Block* nb = replace_block(s, b, 2 * norm.size());
                    if (!nb) break;
                    nb->eof = b->eof;
                    drop_block(b);
                    b = nb;
                    b->head = headroom(*b);
                }
                memcpy(skh_text_buffer(b->t), norm.data(), norm.size());
                PhaseTimer pt(1);
                if (skh_text_upload(b->ctx, b->t, 0, 0) != SKH_OK ||
                    skh_text_parse_at(b->ctx, b->t, 0, norm.size(), SKH_FASTQ, 1, &res) != SKH_OK) {
                    s.fail(skh_last_error());
                    break;
                }
                if (res.status != SKH_PARSE_OK || res.nrec == 0) {
                    s.fail("error when parsing " + s.path + ": invalid FASTA/Q format");
                    break;
                }
            }
            lines_before += lines;
            b->res = res;
            s.parsed_q.push(b);
            if (eof) break;
        }
        s.stop = true;                   // nothing more will be parsed: the reader may stop too
        s.free_q.close();
        s.parsed_q.close();
    }

public:
    BlockPipe(const std::vector<skh_ctx*>& ctxs, const std::string& path, uint64_t block_bytes) : s_(std::make_shared<State>())
    {
        State& s = *s_;
        s.ctxs = ctxs; s.path = path;
        s.nblocks = std::max<size_t>(4, 2 * ctxs.size() + 2);       // two per GPU with their workers, two with the reader and the parser
        if (path != "-") {
            s.fd = open(path.c_str(), O_RDONLY);
            if (s.fd < 0) { s.err = "open " + path + ": no such file or directory"; return; }
            struct stat st;
            regular_ = fstat(s.fd, &st) == 0 && S_ISREG(st.st_mode);
            unsigned char magic[2] = {0, 0};
            const ssize_t got = pread(s.fd, magic, 2, 0);
            if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) { s.fh = gzdopen(s.fd, "rb"); s.fd = -1; }
            else if (regular_) s.file_size = (uint64_t)st.st_size;
        } else {
            s.fh = gzdopen(0, "rb");
        }
        if (s.fd < 0 && !s.fh) { s.err = "open " + path + ": no such file or directory"; return; }
        if (s.fh) gzbuffer((gzFile)s.fh, 1 << 20);
        // the first block here, the other three by the parser thread while the reader fills this one
        Block* b = new_block(s, block_bytes, false, 0);
        if (!b) return;
        s.chunk = b->cap - b->head;
        s.block_bytes = block_bytes;
        s.all.push_back(b);
        s.free_q.push(b);
        reader_ = std::thread(reader, s_);
        parser_ = std::thread(parser, s_);
    }
    ~BlockPipe()
    {
        State& s = *s_;
        s.stop = true;
        s.free_q.close(); s.filled_q.close(); s.parsed_q.close();
        if (parser_.joinable()) parser_.join();
        bool leave = false;
        if (reader_.joinable()) {
            // a reader blocked on a pipe that stays silent cannot be woken: it is left behind with the
            // state it uses (the process is on its way out when this happens)
            if (regular_ || s.reader_done) reader_.join();
            else { reader_.detach(); leave = true; }
        }
        if (leave) return;
        PhaseTimer pt(5);
        for (Block* b : s.all) {
            if (b->ordinary_for) {
                Cache& c = cache();
                std::lock_guard<std::mutex> l(c.mu);
                if (c.blocks.size() < s.nblocks) { c.blocks.push_back(b); continue; }
            }
            if (b->t) skh_text_free(b->t);
            delete b;
        }
        s.all.clear();
        if (s.fh) gzclose((gzFile)s.fh);
        if (s.fd >= 0) close(s.fd);
    }
    BlockPipe(const BlockPipe&) = delete;
    BlockPipe& operator=(const BlockPipe&) = delete;

    bool ok() { return s_->error().empty(); }
    std::string error() { return s_->error(); }
    bool is_fastq() const { return s_->fastq; }
    skh_text* block() const { return cur_ ? cur_->t : nullptr; }
    int device_index() const { return cur_ ? cur_->dev_index : 0; }     // of the current block, in the device list

    // The next parsed block (at least one record), in file order.  false: end of the file or error.
    bool next(skh_parse_result& res)
    {
        if (cur_) { s_->free_q.push(cur_); cur_ = nullptr; }
        { PhaseTimer pt(9); cur_ = s_->parsed_q.pop(); }
        if (!cur_) return false;
        res = cur_->res;
        return true;
    }
    // The caller is done with the current block: it goes back to the reader.
    void advance(const skh_parse_result&)
    {
        if (cur_) { s_->free_q.push(cur_); cur_ = nullptr; }
    }
    // The caller keeps the current block beyond the next call of next() (its output is still on its
    // way to the host); release() hands it back.
    void* hold() { Block* b = cur_; cur_ = nullptr; return b; }
    static skh_text* text_of(void* held) { return static_cast<Block*>(held)->t; }
    void release(void* held) { if (held) s_->free_q.push(static_cast<Block*>(held)); }

private:
    std::shared_ptr<State> s_;
    std::thread reader_, parser_;
    Block* cur_ = nullptr;
    bool regular_ = false;
};

}  // namespace
void drop_cached_blocks() { BlockPipe::drop_cache(); }
namespace {

// ---- one worker thread per GPU behind a BlockPipe --------------------------------------------
//
// The blocks of a file are dealt to the command's GPUs in turn.  The command hands every parsed
// block to the worker of the block's GPU, which runs the block's kernels, formats its output and
// waits for the download, and takes the finished blocks back in file order to write them: the
// kernels and transfers of different GPUs run side by side (what -j / runtime.GOMAXPROCS is to the
// reference's commands, seqkit/cmd/root.go:60-64, seq.go:52), and the writes overlap all of it.
struct BlockJob {
    void* held = nullptr;               // the block, out of the pipe's round until release()
    skh_text* t = nullptr;
    skh_parse_result res;
    int dev = 0;                        // index of its GPU in the command's device list
    // what the step made of it
    int rc = 0; std::string err;        // a failure of the step itself
    const uint8_t* text = nullptr; uint64_t nbytes = 0;    // output text, in the block's pinned memory
    std::string record_error;           // seq -v: the block ends before a record with an invalid letter
    std::vector<uint64_t> offs;         // stats -a: the offsets of its records
    bool done = false;
};

class BlockWorkers {
public:
    BlockWorkers(size_t ndev, std::function<void(BlockJob&)> step) : step_(std::move(step)), q_(ndev)
    {
        for (size_t d = 0; d < ndev; d++) thr_.emplace_back([this, d] { run(d); });
    }
    ~BlockWorkers()
    {
        { std::lock_guard<std::mutex> l(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : thr_) t.join();
    }
    BlockWorkers(const BlockWorkers&) = delete;
    BlockWorkers& operator=(const BlockWorkers&) = delete;
    void submit(BlockJob* j)
    {
        { std::lock_guard<std::mutex> l(mu_); q_[(size_t)j->dev].push_back(j); }
        cv_.notify_all();
    }
    bool is_done(BlockJob* j) { std::lock_guard<std::mutex> l(mu_); return j->done; }
    void wait(BlockJob* j)
    {
        PhaseTimer pt(9);
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return j->done; });
    }

private:
    void run(size_t d)
    {
        for (;;) {
            BlockJob* j;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [&] { return stop_ || !q_[d].empty(); });
                if (stop_) return;                   // what is still queued belongs to a command that gave up
                j = q_[d].front(); q_[d].pop_front();
            }
            step_(*j);
            { std::lock_guard<std::mutex> l(mu_); j->done = true; }
            cv_.notify_all();
        }
    }
    std::function<void(BlockJob&)> step_;
    std::mutex mu_; std::condition_variable cv_;
    std::vector<std::deque<BlockJob*>> q_;
    std::vector<std::thread> thr_;
    bool stop_ = false;
};

// The blocks in flight of one file, in file order: `emit` gets each finished block once, in order.
// Returns the first non-zero result of emit (or of a step), after which nothing more is emitted.
class OrderedBlocks {
public:
    OrderedBlocks(BlockPipe& tb, BlockWorkers& w, size_t limit, std::function<int(BlockJob&)> emit)
        : tb_(tb), w_(w), limit_(limit), emit_(std::move(emit)) {}
    // takes the pipe's current block; finished blocks at the head of the line are emitted
    int submit(const skh_parse_result& res, bool run_here, const std::function<void(BlockJob&)>& step)
    {
        auto j = std::make_unique<BlockJob>();
        j->t = tb_.block(); j->res = res; j->dev = tb_.device_index();
        j->held = tb_.hold();
        if (run_here) { step(*j); j->done = true; }
        else w_.submit(j.get());
        line_.push_back(std::move(j));
        return drain(false);
    }
    int finish() { return drain(true); }
    ~OrderedBlocks() { for (auto& j : line_) w_.wait(j.get()); }     // (the workers still own what they were handed)

private:
    int drain(bool all)
    {
        while (!line_.empty() && (all || line_.size() > limit_ || w_.is_done(line_.front().get()))) {
            BlockJob& j = *line_.front();
            w_.wait(&j);
            int rc = j.rc ? fatal(j.err) : emit_(j);
            tb_.release(j.held);
            line_.pop_front();
            if (rc) return rc;
        }
        return 0;
    }
    BlockPipe& tb_; BlockWorkers& w_; size_t limit_;
    std::function<int(BlockJob&)> emit_;
    std::deque<std::unique_ptr<BlockJob>> line_;
};

// Record IDs by a rule other than the default one: the name spans of the block are replaced by the ID
// spans (the formatter then prints "names" that are the IDs).  The head lines are on the host already
// (the block's text buffer); the expression runs here, record by record.
int custom_ids_as_names(const GlobalOpts& g, skh_text* t, uint64_t nrec, std::vector<uint64_t>& npos, std::vector<uint32_t>& nlen,
                        std::string& err)
{
    skh_ctx* ctx = skh_text_ctx(t);
    npos.resize(nrec); nlen.resize(nrec);
    if (skh_text_names(ctx, t, 0, nrec, npos.data(), nlen.data()) != SKH_OK) { err = skh_last_error(); return -1; }
    const uint8_t* text = skh_text_buffer(t);
    for (uint64_t r = 0; r < nrec; r++) {
        uint32_t off = 0, len = 0;
        id_span(g, text + npos[r], nlen[r], &off, &len);
        npos[r] += off; nlen[r] = len;
    }
    if (skh_text_set_names(ctx, t, 0, nrec, npos.data(), nlen.data()) != SKH_OK) { err = skh_last_error(); return -1; }
    return 0;
}

// alphabet of a file as the reference's reader guesses it: from the first record
int first_record_alphabet(skh_ctx* ctx, skh_text* t, const GlobalOpts& g, int forced, std::string& err)
{
    if (forced >= 0) return forced;
    uint64_t off[2];
    if (skh_text_offsets(ctx, t, 0, 2, off) != SKH_OK) { err = skh_last_error(); return -1; }
    uint64_t n = off[1] - off[0];
    if (g.guess_len != 0 && n > g.guess_len) n = g.guess_len;
    std::vector<uint8_t> head(n ? n : 1);
    if (n && skh_text_sequence(ctx, t, off[0], n, head.data()) != SKH_OK) { err = skh_last_error(); return -1; }
    return guess_alphabet(head.data(), n, g.guess_len);
}

// `seq` with parse and output formatting on the GPU (seq.go:114-280 a block at a time)
int seq_text_path(const GlobalOpts& g, const SeqOpts& o, const std::vector<std::string>& files, const std::vector<skh_ctx*>& ctxs, Sink& out,
                  bool& is_fastq, bool& check_type)      // seq.go:105-106: decided once per command, by the first printed record
{
    std::vector<uint64_t> offs;
    for (const std::string& file : files) {
        bool warn_once = true;                       // seq.go:118: one "already RNA / DNA" warning per file
        BlockPipe tb(ctxs, file, g.text_block_bytes);
        if (!tb.ok()) return fatal(tb.error());
        uint8_t lut[256], valid[256], gap_pre[256];
        bool lut_ready = false, any_map = false;
        int ab = -1;                                  // the file's alphabet: -t, or guessed from its first record
        skh_parse_result res;
        // What a block goes through, in two steps.  The first (validation) and everything decided from the
        // first records of a file or of the command (alphabet, tables, output layout) happen in file order
        // on this thread until they are settled; from then on both steps of a block run on the worker of
        // the block's GPU.
        auto step_validate = [&](BlockJob& j) {
            if (!o.validate || j.res.nrec == 0) return;
            // the reader validates every record it builds; the first invalid one ends the command
            skh_text* t = j.t;
            skh_ctx* ctx = skh_text_ctx(t);
            skh_validate_result vr;
            if (skh_text_validate(ctx, t, valid, o.validate_len, &vr) != SKH_OK) { j.rc = -1; j.err = skh_last_error(); return; }
            if (!vr.found) return;
            uint64_t npos = 0; uint32_t nlen = 0;
            if (skh_text_name(ctx, t, vr.record, &npos, &nlen) != SKH_OK) { j.rc = -1; j.err = skh_last_error(); return; }
            const uint8_t* name = skh_text_buffer(t) + npos;
            uint32_t ido = 0, idl = 0;
            id_span(g, name, nlen, &ido, &idl);
            j.record_error = "error when parsing seq: " + std::string((const char*)name + ido, idl) + " (invalid " +
                             alphabet_name(ab) + " letter: " + std::string(1, (char)vr.letter) + ")";
            if (skh_text_truncate(ctx, t, vr.record) != SKH_OK) { j.rc = -1; j.err = skh_last_error(); return; }
            j.res.nrec = vr.record;
        };
        auto step_transform = [&](BlockJob& j) {
            if (j.rc || j.res.nrec == 0) return;
            skh_text* t = j.t;
            skh_ctx* ctx = skh_text_ctx(t);
            auto failed = [&](const std::string& m) { j.rc = -1; j.err = m; };
            if (o.complement && ((!g.quiet && ab == Protein) || ab == Unlimit)) {
                // seq.go:190-193: one warning per record that gets as far as the complement (the condition
                // reads `!config.Quiet && alphabet == Protein || alphabet == Unlimit`)
                std::vector<uint64_t> lo(j.res.nrec + 1);
                if (skh_text_offsets(ctx, t, 0, j.res.nrec + 1, lo.data()) != SKH_OK) return failed(skh_last_error());
                for (uint64_t r = 0; r < j.res.nrec; r++) {
                    const uint64_t len = lo[r + 1] - lo[r];
                    if (o.min_len >= 0 && len < (uint64_t)o.min_len) continue;
                    if (o.max_len >= 0 && len > (uint64_t)o.max_len) continue;
                    fprintf(stderr, "[WARN] Complement does no take effect on protein/unlimit sequence\n");
                }
            }
            if (o.remove_gaps) {
                // seq.go:196-198 (the byte-wise table commutes with dropping bytes and with the reversal)
                PhaseTimer pt(2);
                if (skh_text_remove_gaps(ctx, t, gap_pre) != SKH_OK) return failed(skh_last_error());
            }
            if (o.reverse || any_map) {
                // seq.go:187-195: ReverseInplace (bases and qualities), then ComplementInplace
                PhaseTimer pt(2);
                if (skh_text_revcomp(ctx, t, any_map ? lut : nullptr, o.reverse, 1) != SKH_OK) return failed(skh_last_error());
            }
            if (check_type) return;                  // no record has passed the length filters yet: nothing is printed
            skh_seq_format f;
            memset(&f, 0, sizeof f);
            f.min_len = o.min_len; f.max_len = o.max_len;
            f.line_width = tb.is_fastq() ? 0 : g.line_width;           // seq.go:136-138
            f.print_name = 1; f.print_seq = 1; f.print_qual = is_fastq; f.plus_line = 1;
            if (o.only_name && o.only_seq) {}
            else if (o.only_name) { f.print_seq = 0; f.print_qual = 0; }
            else if (o.only_seq) { f.print_name = 0; f.print_qual = 0; }
            else if (o.only_qual) {
                if (!is_fastq) return failed("FASTA format has no quality. So do not just use flag -q (--qual)");
                f.print_name = 0; f.print_seq = 0; f.print_qual = 1; f.plus_line = 0;
            }
            // (a FASTA file after a FASTQ one keeps the FASTQ layout: "+" and an empty quality line,
            // seq.go:140-149,258-276; the formatter prints that for a block without a quality plane)
            f.only_id = o.only_id;
            if (o.only_id && g.id_re && f.print_name) {
                std::vector<uint64_t> id_pos;
                std::vector<uint32_t> id_len;
                std::string err;
                if (custom_ids_as_names(g, t, j.res.nrec, id_pos, id_len, err)) return failed(err);
                f.only_id = 0;
            }
            f.mark = is_fastq ? '@' : '>';
            PhaseTimer pt(3);
            if (skh_text_format_seq_submit(ctx, t, &f) != SKH_OK) return failed(skh_last_error());
            if (skh_text_output_wait(ctx, t, &j.text, &j.nbytes) != SKH_OK) return failed(skh_last_error());
        };
        // what needs the file's records in order: the alphabet and the tables from the first record of the
        // file, the output layout from the first record of the command that passes the length filters
        auto step_in_order = [&](BlockJob& j) {
            skh_text* t = j.t;
            skh_ctx* ctx = skh_text_ctx(t);
            auto failed = [&](const std::string& m) { j.rc = -1; j.err = m; };
            if (j.res.nrec == 0) return;
            if (ab < 0 && (o.complement || o.validate || o.dna2rna || o.rna2dna)) {
                std::string err;
                ab = first_record_alphabet(ctx, t, g, g.alphabet, err);
                if (ab < 0) return failed(err);
                skh_valid_table(ab, valid);
            }
            step_validate(j);
            if (j.rc || j.res.nrec == 0) return;
            if (!lut_ready) {
                // One table for everything seq.go:190-241 does to a base: complement, then (when the
                // sequence is printed) T -> U or U -> T, then the case.  Gaps are removed before the table
                // is applied, so the gap test asks what the complement of a byte is.
                uint8_t comp[256];
                for (int b = 0; b < 256; b++) comp[b] = (uint8_t)b;
                if (o.complement) skh_pair_table(ab, comp);
                const bool shown = !(o.only_name && !o.only_seq) && !(o.only_qual && !o.only_seq && !o.only_name);
                bool to_rna = false, to_dna = false;
                if (shown && o.dna2rna) {
                    if (ab == RNA || ab == RNAredundant) {
                        if (warn_once) { fprintf(stderr, "[WARN] it's already RNA, no need to convert\n"); warn_once = false; }
                    } else to_rna = true;
                }
                if (shown && o.rna2dna) {
                    if (ab == DNA || ab == DNAredundant) {
                        if (warn_once) { fprintf(stderr, "[WARN] it's already DNA, no need to convert\n"); warn_once = false; }
                    } else to_dna = true;
                }
                any_map = o.complement;
                for (int b = 0; b < 256; b++) {
                    uint8_t c = comp[b];
                    if (to_rna) c = c == 't' ? 'u' : c == 'T' ? 'U' : c;
                    if (to_dna) c = c == 'u' ? 't' : c == 'U' ? 'T' : c;
                    if (shown && o.lower_case && c >= 'A' && c <= 'Z') c = (uint8_t)(c + 32);
                    else if (shown && o.upper_case && c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);
                    lut[b] = c;
                    if (c != (uint8_t)b) any_map = true;
                    gap_pre[b] = 0;
                }
                for (unsigned char gl : o.gap_letters)
                    for (int b = 0; b < 256; b++) if (comp[b] == gl) gap_pre[b] = 1;
                lut_ready = true;
            }
            if (check_type) {
                // the first record that passes the length filters decides how everything is printed
                // (seq.go:140-145: looked at as the record was read)
                offs.resize(j.res.nrec + 1);
                if (skh_text_offsets(ctx, t, 0, j.res.nrec + 1, offs.data()) != SKH_OK) return failed(skh_last_error());
                for (uint64_t r = 0; r < j.res.nrec && check_type; r++) {
                    const uint64_t len = offs[r + 1] - offs[r];
                    if (o.min_len >= 0 && len < (uint64_t)o.min_len) continue;
                    if (o.max_len >= 0 && len > (uint64_t)o.max_len) continue;
                    is_fastq = tb.is_fastq() && len > 0;
                    check_type = false;
                }
            }
            step_transform(j);
        };
        auto step_on_worker = [&](BlockJob& j) { step_validate(j); step_transform(j); };
        std::string pending_error;                    // an invalid record: reported after the records before it
        auto emit = [&](BlockJob& j) -> int {
            if (j.nbytes) { PhaseTimer pt(4); out.write(j.text, j.nbytes); }
            if (!j.record_error.empty()) { pending_error = j.record_error; return -1; }
            return 0;
        };
        {
            BlockWorkers workers(ctxs.size(), step_on_worker);
            OrderedBlocks line(tb, workers, 2 * ctxs.size(), emit);
            int rc = 0;
            while (rc == 0 && tb.next(res)) {
                // settled: the alphabet is known (or not needed), the tables are made, the layout is decided
                const bool settled = lut_ready && !check_type;
                rc = line.submit(res, !settled, step_in_order);
            }
            if (rc == 0) rc = line.finish();
            if (!pending_error.empty()) { out.flush(); return fatal(pending_error); }
            if (rc) return rc;
        }
        if (!tb.error().empty()) return fatal(tb.error());
    }
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------ seq
int cmd_seq(const GlobalOpts& g, const SeqOpts& o, const std::vector<std::string>& files)
{
    Ctx ctx;
    const double t_init = Phases::now();
    if (ctx.init(g) != 0) return fatal(skh_last_error());
    if (g_phases.on) fprintf(stderr, "[timing] skh_init %.3f s\n", Phases::now() - t_init);
    Sink out(g.out_file);
    if (!out.ok()) return fatal("open " + g.out_file + ": cannot create");
    const bool text_only = o.remove_gaps || o.lower_case || o.upper_case || o.dna2rna || o.rna2dna;
    if (g.host_parse && text_only)
        return fatal("--host-parse does not know -g, -l, -u, --dna2rna, --rna2dna (they are done on the GPU text path)");
    bool is_fastq = false, check_type = true;        // seq.go:105-106: decided once, by the first record
    TwoBatches tb;
    BatchWork work[2];

    for (const std::string& file : files) {
        // File text goes to the GPUs a block at a time.  Records are scanned and printed by the host, and
        // only the per-record work is done on the GPUs, with --host-parse -- and, when there are several
        // GPUs, for a file whose first record is longer than a text block: a pinned batch is cut over all
        // GPUs, inside its records where they are long (skh_revcomp_multi), while a text block and the
        // records in it belong to one GPU.
        const bool batches = g.host_parse || (ctx.multi && !text_only && !g.id_re && first_record_exceeds_block(file, g.text_block_bytes));
        if (!batches) {
            if (int rc = seq_text_path(g, o, {file}, ctx.all, out, is_fastq, check_type)) return rc;
            continue;
        }
        FastxReader rd(ctx.c, file, g.alphabet, g.guess_len);
        if (!rd.ok()) return fatal(rd.error());
        uint8_t lut[256];
        bool lut_ready = false;

        // prints a finished batch (seq.go:163-276)
        auto emit = [&](Batch& bt) -> int {
            const int width = rd.is_fastq() ? 0 : g.line_width;        // seq.go:136-138
            const uint64_t* off = bt.offs();
            for (uint64_t r = 0; r < bt.nrec; r++) {
                const uint64_t len = off[r + 1] - off[r];
                if (o.min_len >= 0 && len < (uint64_t)o.min_len) continue;
                if (o.max_len >= 0 && len > (uint64_t)o.max_len) continue;
                if (check_type) { is_fastq = rd.is_fastq() && len > 0; check_type = false; }
                bool pq = is_fastq, pn = true, ps = true;
                if (o.only_name && o.only_seq) { pn = ps = true; }
                else if (o.only_name) { ps = false; pq = false; }
                else if (o.only_seq) { pn = false; pq = false; }
                else if (o.only_qual) {
                    if (!is_fastq) return fatal("FASTA format has no quality. So do not just use flag -q (--qual)");
                    pn = false; ps = false; pq = true;
                }
                if (pn) {
                    const char* name = bt.names.data() + bt.name_off[r];
                    const size_t nl = o.only_id ? bt.id_len[r] : (size_t)(bt.name_off[r + 1] - bt.name_off[r]);
                    if (ps) out.put(is_fastq ? '@' : '>');
                    out.write(name, nl);
                    out.put('\n');
                }
                if (ps) { write_wrapped(out, bt.seq() + off[r], len, width); out.put('\n'); }
                if (pq) {
                    if (!o.only_qual) out.puts("+\n");
                    // (a FASTA file after a FASTQ one has no qualities, though the batch, kept from that file, has a plane)
                    if (bt.qual() && rd.is_fastq()) write_wrapped(out, bt.qual() + off[r], len, width);
                    out.put('\n');
                }
            }
            return 0;
        };

        // the GPU work of batch k: one GPU = the batch's own stream, several = the helper thread of work[k]
        auto gpu_wait = [&](int k) -> int {
            if (!ctx.multi) return skh_batch_wait(tb.bt[k].b);
            const int rc = work[k].wait();
            ctx.report_shares("seq");
            return rc;
        };
        int cur = 0;
        bool have_prev = false;
        std::string pending_error;                    // an invalid record: reported after the records before it
        uint8_t valid[256];
        bool valid_ready = false;
        for (;;) {
            Batch& bt = tb.bt[cur];
            bool got = rd.next_batch(bt, g.batch_bytes, g.batch_records);
            if (!rd.error().empty()) return fatal("error when parsing " + file + ": " + rd.error());
            if (got && o.validate) {
                if (!valid_ready) { skh_valid_table(rd.alphabet(), valid); valid_ready = true; }
                skh_validate_result vr;
                if (skh_validate(ctx.c, bt.b, bt.nrec, valid, o.validate_len, &vr) != SKH_OK) return fatal(skh_last_error());
                if (vr.found) {
                    pending_error = "error when parsing seq: " +
                                    std::string(bt.names.data() + bt.name_off[vr.record], bt.id_len[vr.record]) +
                                    " (invalid " + alphabet_name(rd.alphabet()) + " letter: " +
                                    std::string(1, (char)vr.letter) + ")";
                    bt.nrec = vr.record;              // the records before it are still transformed and printed
                    if (bt.nrec == 0) got = false;
                }
            }
            if (got) {
                if (o.complement && !lut_ready) {
                    skh_pair_table(rd.alphabet(), lut);
                    lut_ready = true;
                }
                if (o.complement && ((!g.quiet && rd.alphabet() == Protein) || rd.alphabet() == Unlimit)) {
                    // seq.go:190-193: one warning per record that gets as far as the complement
                    const uint64_t* off = bt.offs();
                    for (uint64_t r = 0; r < bt.nrec; r++) {
                        const uint64_t len = off[r + 1] - off[r];
                        if (o.min_len >= 0 && len < (uint64_t)o.min_len) continue;
                        if (o.max_len >= 0 && len > (uint64_t)o.max_len) continue;
                        fprintf(stderr, "[WARN] Complement does no take effect on protein/unlimit sequence\n");
                    }
                }
            }
            // with several GPUs all of them work on one batch at a time: the batch before must be back
            // before this one goes out (it is printed while this one is on the GPUs)
            if (have_prev && ctx.multi && gpu_wait(cur ^ 1) != SKH_OK) return fatal(work[cur ^ 1].err);
            if (got && (o.reverse || o.complement)) {
                // seq.go:187-195: ReverseInplace (bases and qualities), then ComplementInplace
                if (ctx.multi) {
                    skh_multi* m = ctx.multi; skh_batch* b = bt.b; const uint64_t n = bt.nrec;
                    std::array<uint8_t, 256> table;
                    memcpy(table.data(), lut, 256);
                    const bool comp = o.complement;
                    const int rev = o.reverse;
                    work[cur].start([m, b, n, table, comp, rev] { return skh_revcomp_multi(m, b, n, comp ? table.data() : nullptr, rev, 1); });
                } else if (skh_revcomp_submit(ctx.c, bt.b, bt.nrec, o.complement ? lut : nullptr, o.reverse, 1) != SKH_OK)
                    return fatal(skh_last_error());
            }
            if (have_prev) {
                Batch& pv = tb.bt[cur ^ 1];
                if (!ctx.multi && gpu_wait(cur ^ 1) != SKH_OK) return fatal(skh_last_error());
                if (int rc = emit(pv)) return rc;
            }
            if (!pending_error.empty()) {
                if (got) {
                    if (gpu_wait(cur) != SKH_OK) return fatal(ctx.multi ? work[cur].err : std::string(skh_last_error()));
                    if (int rc = emit(bt)) return rc;
                }
                out.flush();
                return fatal(pending_error);
            }
            if (!got) break;
            have_prev = true;
            cur ^= 1;
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------ fx2tab
int cmd_fx2tab(const GlobalOpts& g, const Fx2tabOpts& o, const std::vector<std::string>& files)
{
    Ctx ctx;
    if (ctx.init(g) != 0) return fatal(skh_last_error());
    Sink out(g.out_file);
    if (!out.ok()) return fatal("open " + g.out_file + ": cannot create");

    // BaseContent(list) = sum over the bytes of `list` (duplicates count again) of the
    // occurrences of the byte in either case (fx2tab.go:128,129,141).  One GPU class per
    // distinct case-folded byte that any requested list mentions; the lists are then sums of
    // class counts.  More than SKH_MAX_CLASSES distinct bytes take further counting passes.
    std::vector<uint8_t> letters;                    // distinct folded bytes, in first-use order
    auto fold = [](uint8_t b) { return (uint8_t)((b >= 'a' && b <= 'z') ? b - 32 : b); };
    auto want = [&](const std::string& list) {
        for (unsigned char ch : list) {
            const uint8_t f = fold(ch);
            if (std::find(letters.begin(), letters.end(), f) == letters.end()) letters.push_back(f);
        }
    };
    if (o.gc || o.gc_skew) { want("G"); want("C"); }
    for (const auto& bc : o.base) want(bc);
    const size_t npass = (letters.size() + SKH_MAX_CLASSES - 1) / SKH_MAX_CLASSES;
    std::vector<std::vector<uint8_t>> pass_table(npass, std::vector<uint8_t>(256, SKH_NO_CLASS));
    std::vector<uint32_t> pass_nclass(npass, 0);
    for (size_t i = 0; i < letters.size(); i++) {
        const size_t ps = i / SKH_MAX_CLASSES;
        const uint8_t cls = (uint8_t)(i % SKH_MAX_CLASSES), up = letters[i];
        pass_table[ps][up] = cls;
        if (up >= 'A' && up <= 'Z') pass_table[ps][up + 32] = cls;
        pass_nclass[ps] = cls + 1u;
    }
    auto letter_index = [&](uint8_t ch) {
        return (size_t)(std::find(letters.begin(), letters.end(), fold(ch)) - letters.begin());
    };

    if (o.header_line) {                             // fx2tab.go:68-89
        out.puts("#name\tseq\tqual");
        if (o.length) out.puts("\tlength");
        if (o.gc) out.puts("\tGC");
        if (o.gc_skew) out.puts("\tGC-Skew");
        for (const auto& bc : o.base) { out.put('\t'); out.write(bc.data(), bc.size()); }
        if (o.alphabet_col) out.puts("\talphabet");
        out.put('\n');
    }

    // Rows built on the GPU from file text parsed on the GPU, whenever the request fits the
    // device formatter (everything but very long letter lists).
    bool device_rows = !g.host_parse && npass <= SKH_TAB_MAX_PASSES;
    skh_tab_format tf;
    memset(&tf, 0, sizeof tf);
    if (device_rows) {
        auto add = [&](uint8_t kind, const std::string& list) {
            if (tf.ncols >= SKH_TAB_MAX_COLS || list.size() > SKH_TAB_MAX_TERMS) { device_rows = false; return; }
            skh_tab_column& c = tf.cols[tf.ncols++];
            c.kind = kind;
            c.nterm = (uint8_t)list.size();
            for (size_t k = 0; k < list.size(); k++) c.term[k] = (uint8_t)letter_index((uint8_t)list[k]);
        };
        if (o.length) add(SKH_TAB_LENGTH, "");
        if (o.gc) add(SKH_TAB_GC, "GC");
        if (o.gc_skew) add(SKH_TAB_GC_SKEW, "GC");
        for (const auto& bc : o.base) add(SKH_TAB_CONTENT, bc);
        tf.only_id = o.only_id;
    }
    TwoBatches tb;
    BatchWork work[2];
    std::vector<uint64_t> cnt[2];                    // per batch: nrec x letters.size(), letter-major per record
    std::vector<uint64_t> scratch;
    std::string num;
    // one file through pinned batches: records scanned and rows printed by the host, counts on the GPUs
    auto batch_file = [&](const std::string& file) -> int {
        FastxReader rd(ctx.c, file, Auto, g.guess_len);
        if (!rd.ok()) return fatal(rd.error());

        // the counting passes of a batch: with several GPUs the batch is cut over all of them (skh_count_multi)
        auto count_batch = [&](Batch& bt, std::vector<uint64_t>& dst, int k) -> int {
            const size_t nl = letters.size();
            dst.assign(bt.nrec * nl, 0);
            for (size_t ps = 0; ps < npass; ps++) {
                const uint32_t nc = pass_nclass[ps];
                scratch.resize(bt.nrec * nc);
                // all but the last pass run synchronously; the last one overlaps with the host
                if (ps + 1 < npass) {
                    const int rc = ctx.multi ? skh_count_multi(ctx.multi, bt.b, bt.nrec, pass_table[ps].data(), nc, scratch.data())
                                             : skh_count(ctx.c, bt.b, bt.nrec, pass_table[ps].data(), nc, scratch.data());
                    if (rc != SKH_OK) return fatal(skh_last_error());
                    for (uint64_t r = 0; r < bt.nrec; r++)
                        for (uint32_t c = 0; c < nc; c++) dst[r * nl + ps * SKH_MAX_CLASSES + c] = scratch[r * nc + c];
                } else {
                    bt.counts.resize(bt.nrec * nc);
                    if (ctx.multi) {
                        skh_multi* m = ctx.multi; skh_batch* b = bt.b; const uint64_t n = bt.nrec;
                        const uint8_t* table = pass_table[ps].data();          // lives as long as the command
                        uint64_t* out_counts = bt.counts.data();
                        work[k].start([m, b, n, table, nc, out_counts] { return skh_count_multi(m, b, n, table, nc, out_counts); });
                    } else if (skh_count_submit(ctx.c, bt.b, bt.nrec, pass_table[ps].data(), nc, bt.counts.data()) != SKH_OK)
                        return fatal(skh_last_error());
                }
            }
            return 0;
        };
        auto finish_counts = [&](Batch& bt, std::vector<uint64_t>& dst) {
            if (!npass) return;
            const size_t nl = letters.size(), ps = npass - 1;
            const uint32_t nc = pass_nclass[ps];
            for (uint64_t r = 0; r < bt.nrec; r++)
                for (uint32_t c = 0; c < nc; c++) dst[r * nl + ps * SKH_MAX_CLASSES + c] = bt.counts[r * nc + c];
        };
        auto base_content = [&](const uint64_t* row, const std::string& list, uint64_t len) -> double {
            if (len == 0) return 0.0;                // BaseContent of an empty sequence
            uint64_t sum = 0;
            for (unsigned char ch : list) sum += row[letter_index(ch)];
            return (double)sum / (double)len;
        };
        auto emit = [&](Batch& bt, const std::vector<uint64_t>& c) {
            const uint64_t* off = bt.offs();
            const size_t nl = letters.size();
            for (uint64_t r = 0; r < bt.nrec; r++) {
                const uint64_t len = off[r + 1] - off[r];
                const char* name = bt.names.data() + bt.name_off[r];
                out.write(name, o.only_id ? bt.id_len[r] : (size_t)(bt.name_off[r + 1] - bt.name_off[r]));
                out.put('\t');
                if (!o.only_name) out.write(bt.seq() + off[r], len);
                out.put('\t');
                if (!o.only_name && bt.qual() && rd.is_fastq()) out.write(bt.qual() + off[r], len);
                if (o.length) { out.put('\t'); num = std::to_string(len); out.write(num.data(), num.size()); }
                const uint64_t* row = nl ? &c[r * nl] : nullptr;
                double gg = 0, cc = 0;
                if (o.gc || o.gc_skew) {             // two separate divisions, then the sum (fx2tab.go:128-133)
                    gg = base_content(row, "G", len);
                    cc = base_content(row, "C", len);
                }
                if (o.gc) { fmt_fixed((gg + cc) * 100, 2, num); out.put('\t'); out.write(num.data(), num.size()); }
                if (o.gc_skew) { fmt_fixed((gg - cc) / (gg + cc) * 100, 2, num); out.put('\t'); out.write(num.data(), num.size()); }
                for (const auto& bc : o.base) {
                    fmt_fixed(base_content(row, bc, len) * 100, 2, num);
                    out.put('\t'); out.write(num.data(), num.size());
                }
                if (o.alphabet_col) {                // fx2tab.go:167-180 (host side; not part of the GPU path)
                    bool present[256] = {false};
                    for (uint64_t i = off[r]; i < off[r + 1]; i++) present[bt.seq()[i]] = true;
                    out.put('\t');
                    for (int b = 0; b < 256; b++) if (present[b]) out.put((char)b);
                }
                out.put('\n');
            }
        };

        int cur = 0;
        bool have_prev = false;
        for (;;) {
            Batch& bt = tb.bt[cur];
            const bool got = rd.next_batch(bt, g.batch_bytes, g.batch_records);
            if (!rd.error().empty()) return fatal("error when parsing " + file + ": " + rd.error());
            // (several GPUs: all of them work on one batch at a time, so the batch before comes back first)
            if (have_prev && ctx.multi) {
                if (work[cur ^ 1].wait() != SKH_OK) return fatal(work[cur ^ 1].err);
                ctx.report_shares("fx2tab");
            }
            if (got && npass) if (int rc = count_batch(bt, cnt[cur], cur)) return rc;
            if (have_prev) {
                Batch& pv = tb.bt[cur ^ 1];
                if (!ctx.multi && skh_batch_wait(pv.b) != SKH_OK) return fatal(skh_last_error());
                finish_counts(pv, cnt[cur ^ 1]);
                emit(pv, cnt[cur ^ 1]);
            }
            if (!got) break;
            have_prev = true;
            cur ^= 1;
        }
        return 0;
    };

    if (device_rows) {
        for (const std::string& file : files) {
            // (several GPUs and a first record longer than a text block: pinned batches, cut over all GPUs)
            if (ctx.multi && !g.id_re && first_record_exceeds_block(file, g.text_block_bytes)) {
                if (int rc = batch_file(file)) return rc;
                continue;
            }
            BlockPipe tb(ctx.all, file, g.text_block_bytes);
            if (!tb.ok()) return fatal(tb.error());
            skh_parse_result res;
            // every block is counted, formatted and downloaded by the worker of its GPU; the rows are
            // written here in file order
            auto step = [&](BlockJob& j) {
                skh_text* t = j.t;
                skh_ctx* c = skh_text_ctx(t);
                auto failed = [&](const std::string& m) { j.rc = -1; j.err = m; };
                if (j.res.nrec == 0) return;
                {
                    PhaseTimer pt(2);
                    for (size_t ps = 0; ps < npass; ps++)
                        if (skh_text_count(c, t, pass_table[ps].data(), pass_nclass[ps], (uint32_t)ps) != SKH_OK)
                            return failed(skh_last_error());
                }
                if (o.alphabet_col) {                          // fx2tab.go:145-147
                    PhaseTimer pt(2);
                    if (skh_text_letters(c, t) != SKH_OK) return failed(skh_last_error());
                }
                skh_tab_format f = tf;
                f.alphabet = o.alphabet_col ? 1 : 0;
                f.only_id = o.only_id;
                if (o.only_id && g.id_re) {
                    std::vector<uint64_t> id_pos;
                    std::vector<uint32_t> id_len;
                    std::string err;
                    if (custom_ids_as_names(g, t, j.res.nrec, id_pos, id_len, err)) return failed(err);
                    f.only_id = 0;
                }
                f.print_seq = !o.only_name;
                f.print_qual = !o.only_name && tb.is_fastq();
                PhaseTimer pt(3);
                if (skh_text_format_tab_submit(c, t, &f) != SKH_OK) return failed(skh_last_error());
                if (skh_text_output_wait(c, t, &j.text, &j.nbytes) != SKH_OK) return failed(skh_last_error());
            };
            auto emit = [&](BlockJob& j) -> int {
                if (j.nbytes) { PhaseTimer pt(4); out.write(j.text, j.nbytes); }
                return 0;
            };
            {
                BlockWorkers workers(ctx.all.size(), step);
                OrderedBlocks line(tb, workers, 2 * ctx.all.size(), emit);
                int rc = 0;
                while (rc == 0 && tb.next(res)) rc = line.submit(res, false, step);
                if (rc == 0) rc = line.finish();
                if (rc) return rc;
            }
            if (!tb.error().empty()) return fatal(tb.error());
        }
        return 0;
    }

    for (const std::string& file : files)
        if (int rc = batch_file(file)) return rc;
    return 0;
}

// ------------------------------------------------------------------------------------ stats
namespace {

struct StatInfo {
    std::string file, format, type;
    int64_t num = 0, len_sum = 0, gap_sum = 0, len_min = 0, len_max = 0, n50 = 0, l50 = 0, q1 = 0, q2 = 0, q3 = 0;
    double len_avg = 0;
};

int64_t median(const int64_t* sorted, size_t l)      // stat.go:297-306
{
    if (l == 0) return 0;
    if (l % 2 == 0) return (sorted[l / 2 - 1] + sorted[l / 2]) / 2;
    return sorted[l / 2];
}

// one row of the table from a file's totals (stat.go:126-160); sorts `lens` when all is set
StatInfo make_stat_info(const std::string& file, bool no_record, bool fastq_first, int ab, const skh_stats_partial& tot,
                        std::vector<int64_t>& lens, bool all)
{
    StatInfo s;
    s.file = file;
    s.format = no_record ? "" : (fastq_first ? "FASTQ" : "FASTA");
    s.type = ab == DNAredundant ? "DNA" : (ab == RNAredundant ? "RNA" : alphabet_name(ab));
    s.len_max = tot.len_max;
    if (tot.num != 0) {
        s.num = tot.num; s.len_sum = tot.len_sum; s.gap_sum = tot.gap_sum; s.len_min = tot.len_min;
        s.len_avg = round_to((double)tot.len_sum / (double)tot.num, 1);        // stat.go:158
        if (all) {
            std::sort(lens.begin(), lens.end());
            int64_t sum = 0;                                                     // stat.go:138-145
            for (size_t k = lens.size(); k-- > 0;) {
                sum += lens[k];
                if ((sum << 1) >= s.len_sum) { s.n50 = lens[k]; s.l50 = (int64_t)(lens.size() - k); break; }
            }
            const size_t l = lens.size();                                        // stat.go:308-326
            const size_t c1 = l % 2 == 0 ? l / 2 : (l - 1) / 2, c2 = l % 2 == 0 ? l / 2 : c1 + 1;
            s.q1 = median(lens.data(), c1);
            s.q2 = median(lens.data(), l);
            s.q3 = median(lens.data() + c2, l - c2);
        }
    }
    return s;
}

}  // namespace

int cmd_stats(const GlobalOpts& g, const StatsOpts& o, const std::vector<std::string>& files)
{
    if (o.gap_letters.empty()) return fatal("value of flag -G (--gap-letters) should not be empty");
    uint8_t is_gap[256] = {0};
    for (unsigned char ch : o.gap_letters) {
        if (ch > 127) return fatal("value of -G (--gap-letters) contains non-ASCII characters");
        is_gap[ch] = 1;
    }
    Ctx ctx;
    if (ctx.init(g) != 0) return fatal(skh_last_error());
    Sink out(g.out_file);
    if (!out.ok()) return fatal("open " + g.out_file + ": cannot create");

    std::vector<StatInfo> infos;
    Batch bt;
    struct Free { Batch& b; ~Free() { if (b.b) skh_batch_free(b.b); } } guard{bt};
    for (const std::string& file : files) {
        // (several GPUs and a first record longer than a text block: pinned batches, cut over all GPUs)
        const bool batches = g.host_parse || (ctx.multi && first_record_exceeds_block(file, g.text_block_bytes));
        if (!batches) {
            // file text parsed on the GPU: per block the totals kernel, and for -a the lengths
            BlockPipe tb(ctx.all, file, g.text_block_bytes);
            if (!tb.ok()) return fatal(tb.error());
            // one partial per GPU the blocks are dealt to (each GPU's worker adds to its own); merged at the
            // end of the file
            std::vector<skh_stats_partial> part(ctx.all.size());
            for (auto& p : part) skh_stats_init(&p);
            std::vector<int64_t> lens;
            bool fastq_first = false, first = true;
            int ab = -1;
            skh_parse_result res;
            auto step = [&](BlockJob& j) {
                skh_text* t = j.t;
                skh_ctx* c = skh_text_ctx(t);
                auto failed = [&](const std::string& m) { j.rc = -1; j.err = m; };
                if (j.res.nrec == 0) return;
                if (o.all) {
                    j.offs.resize(j.res.nrec + 1);
                    if (skh_text_offsets(c, t, 0, j.res.nrec + 1, j.offs.data()) != SKH_OK) return failed(skh_last_error());
                }
                if (skh_text_stats(c, t, is_gap, &part[(size_t)j.dev]) != SKH_OK) return failed(skh_last_error());
            };
            // the first block with records also tells the file's format and alphabet (stat.go:87,107-113)
            auto step_first = [&](BlockJob& j) {
                if (j.res.nrec == 0) return;
                skh_text* t = j.t;
                uint64_t two[2];
                if (skh_text_offsets(skh_text_ctx(t), t, 0, 2, two) != SKH_OK) { j.rc = -1; j.err = skh_last_error(); return; }
                fastq_first = tb.is_fastq() && two[1] > two[0];
                std::string err;
                ab = first_record_alphabet(skh_text_ctx(t), t, g, g.alphabet, err);     // stat.go:87: config.Alphabet goes to the reader
                if (ab < 0) { j.rc = -1; j.err = err; return; }
                first = false;
                step(j);
            };
            auto emit = [&](BlockJob& j) -> int {
                if (o.all) for (uint64_t r = 0; r < j.res.nrec; r++) lens.push_back((int64_t)(j.offs[r + 1] - j.offs[r]));
                return 0;
            };
            {
                BlockWorkers workers(ctx.all.size(), step);
                OrderedBlocks line(tb, workers, 2 * ctx.all.size(), emit);
                int rc = 0;
                while (rc == 0 && tb.next(res)) rc = line.submit(res, first, step_first);
                if (rc == 0) rc = line.finish();
                if (rc) return rc;
            }
            if (!tb.error().empty()) return fatal(tb.error());
            if (part.size() > 1) {
                // the path's one collective: the per-GPU totals all-reduced over RCCL (stat.go:106-125 across GPUs)
                if (ctx.make_comms() != 0) return fatal(skh_last_error());
                if (skh_stats_allreduce(ctx.comm, part.data()) != SKH_OK) return fatal(skh_last_error());
            }
            const skh_stats_partial tot = part[0];
            infos.push_back(make_stat_info(file, first, fastq_first, ab, tot, lens, o.all));
            continue;
        }
        FastxReader rd(ctx.c, file, g.alphabet, g.guess_len);
        if (!rd.ok()) return fatal(rd.error());
        skh_stats_partial tot;
        skh_stats_init(&tot);
        std::vector<int64_t> lens;
        bool fastq_first = false, first = true;
        while (rd.next_batch(bt, g.batch_bytes, g.batch_records)) {
            if (first) { fastq_first = rd.is_fastq() && bt.offs()[1] > 0; first = false; }   // stat.go:107-113
            // num, len_sum, gap_sum, len_min, len_max of the batch on the GPU (stat.go:106-125)
            // (several GPUs: the batch is cut over all of them and their totals are all-reduced, skh_stats_multi)
            if (ctx.multi && ctx.make_comms() != 0) return fatal(skh_last_error());
            const int rc = ctx.multi ? skh_stats_multi(ctx.multi, bt.b, bt.nrec, is_gap, &tot)
                                     : skh_stats_reduce(ctx.c, bt.b, bt.nrec, is_gap, &tot);
            if (rc != SKH_OK) return fatal(skh_last_error());
            ctx.report_shares("stats");
            if (o.all) {
                const uint64_t* off = bt.offs();
                for (uint64_t r = 0; r < bt.nrec; r++) lens.push_back((int64_t)(off[r + 1] - off[r]));
            }
        }
        if (!rd.error().empty()) return fatal("error when parsing " + file + ": " + rd.error());
        infos.push_back(make_stat_info(file, first, fastq_first, rd.alphabet(), tot, lens, o.all));
    }

    char line[8192];
    if (o.tabular) {                                 // stat.go:164-208
        out.puts("file\tformat\ttype\tnum_seqs\tsum_len\tmin_len\tavg_len\tmax_len");
        if (o.all) out.puts("\tQ1\tQ2\tQ3\tsum_gap\tN50");
        out.put('\n');
        for (const StatInfo& s : infos) {
            snprintf(line, sizeof line, "%s\t%s\t%s\t%lld\t%lld\t%lld\t%.1f\t%lld", s.file.c_str(), s.format.c_str(),
                     s.type.c_str(), (long long)s.num, (long long)s.len_sum, (long long)s.len_min, s.len_avg,
                     (long long)s.len_max);
            out.puts(line);
            if (o.all) {
                snprintf(line, sizeof line, "\t%lld\t%lld\t%lld\t%lld\t%lld", (long long)s.q1, (long long)s.q2,
                         (long long)s.q3, (long long)s.gap_sum, (long long)s.n50);
                out.puts(line);
            }
            out.put('\n');
        }
        return 0;
    }
    // pretty table (stat.go:211-268): columns padded to their widest cell, numeric columns
    // right-aligned, two spaces between columns
    static const char* hdr[13] = {"file", "format", "type", "num_seqs", "sum_len", "min_len", "avg_len",
                                  "max_len", "Q1", "Q2", "Q3", "sum_gap", "N50"};
    const size_t ncol = o.all ? 13 : 8;
    std::vector<std::vector<std::string>> rows;
    rows.emplace_back(hdr, hdr + ncol);
    for (const StatInfo& s : infos) {
        std::vector<std::string> r = {s.file, s.format, s.type, comma(s.num), comma(s.len_sum), comma(s.len_min),
                                      commaf(s.len_avg), comma(s.len_max)};
        if (o.all) { r.push_back(comma(s.q1)); r.push_back(comma(s.q2)); r.push_back(comma(s.q3));
                     r.push_back(comma(s.gap_sum)); r.push_back(comma(s.n50)); }
        rows.push_back(r);
    }
    std::vector<size_t> width(ncol, 0);
    for (const auto& r : rows) for (size_t c = 0; c < ncol; c++) width[c] = std::max(width[c], r[c].size());
    for (const auto& r : rows) {
        for (size_t c = 0; c < ncol; c++) {
            if (c) out.puts("  ");
            const size_t pad = width[c] - r[c].size();
            if (c >= 3) for (size_t i = 0; i < pad; i++) out.put(' ');
            out.write(r[c].data(), r[c].size());
            if (c < 3) for (size_t i = 0; i < pad; i++) out.put(' ');
        }
        out.put('\n');
    }
    return 0;
}

// ------------------------------------------------------------------------------------ subseq
namespace {

struct Feature {
    std::string chr, type, label;      // chr and type lower-cased (subseq.go:156,160,189); label = tag value / BED name
    std::string strand;                // "+", "-", or "" when not given
    long long start = 0, end = 0;      // 1-based, inclusive
};

std::string lower(std::string s)
{
    for (auto& c : s) c = (char)tolower((unsigned char)c);
    return s;
}

std::vector<std::string> split_tabs(const std::string& line, size_t max_items)
{
    std::vector<std::string> it;
    size_t a = 0;
    while (it.size() + 1 < max_items) {
        const size_t t = line.find('\t', a);
        if (t == std::string::npos) break;
        it.push_back(line.substr(a, t - a));
        a = t + 1;
    }
    it.push_back(line.substr(a));
    return it;
}

// value of `tag` in a GTF attribute column: tag "value"; tag2 "value2";
std::string gtf_tag_value(const std::string& attrs, const std::string& tag)
{
    size_t p = 0;
    while (p < attrs.size()) {
        while (p < attrs.size() && (attrs[p] == ' ' || attrs[p] == ';')) p++;
        size_t q = p;
        while (q < attrs.size() && attrs[q] != ';') {
            if (attrs[q] == '"') { q++; while (q < attrs.size() && attrs[q] != '"') q++; if (q < attrs.size()) q++; }
            else q++;
        }
        if (q - p > tag.size() && attrs.compare(p, tag.size(), tag) == 0 && attrs[p + tag.size()] == ' ') {
            size_t v = p + tag.size() + 1, ve = q;
            while (ve > v && attrs[ve - 1] == ' ') ve--;
            if (ve - v >= 2 && attrs[v] == '"' && attrs[ve - 1] == '"') { v++; ve--; }
            return attrs.substr(v, ve - v);
        }
        p = q;
    }
    return "";
}

bool in_list_ci(const std::string& v, const std::vector<std::string>& list)
{
    const std::string l = lower(v);
    for (auto& x : list) if (lower(x) == l) return true;
    return false;
}

// GTF (gtf.ReadFilteredFeatures, subseq.go:146-152) or BED (cmd/bed.go:52-121) features, plain or gzip
bool read_features(const SubseqOpts& o, bool is_bed, std::vector<Feature>& out, std::string& err)
{
    const std::string& path = is_bed ? o.bed : o.gtf;
    gzFile fh = gzopen(path.c_str(), "rb");
    if (!fh) { err = "open " + path + ": no such file or directory"; return false; }
    gzbuffer(fh, 1 << 20);
    std::string line;
    std::vector<char> buf(1 << 16);
    bool more = true;
    while (more) {
        line.clear();
        for (;;) {
            if (!gzgets(fh, buf.data(), (int)buf.size())) { more = false; break; }
            line += buf.data();
            if (!line.empty() && line.back() == '\n') break;
        }
        if (!more && line.empty()) break;
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        Feature f;
        if (is_bed) {
            if ((line.size() > 7 && line.compare(0, 7, "browser") == 0) || (line.size() > 5 && line.compare(0, 5, "track") == 0)) continue;
            const auto it = split_tabs(line, 64);
            if (it.size() < 3) continue;
            if (!o.chr.empty() && std::find(o.chr.begin(), o.chr.end(), it[0]) == o.chr.end()) continue;   // bed.go:75-79
            char *e1 = nullptr, *e2 = nullptr;
            const long long a = strtoll(it[1].c_str(), &e1, 10), b = strtoll(it[2].c_str(), &e2, 10);
            if (it[1].empty() || *e1) { err = it[0] + ": bad start: " + it[1]; gzclose(fh); return false; }
            if (it[2].empty() || *e2) { err = it[0] + ": bad end: " + it[2]; gzclose(fh); return false; }
            if (a >= b) {
                err = it[0] + ": start (" + std::to_string(a) + ") must be <= end (" + std::to_string(b) + ")";
                gzclose(fh); return false;
            }
            if (it.size() >= 6 && it[5] != "+" && it[5] != "-") { err = "bad strand: " + it[5]; gzclose(fh); return false; }
            f.chr = lower(it[0]); f.start = a + 1; f.end = b;
            if (it.size() >= 4) f.label = it[3];
            if (it.size() >= 6) f.strand = it[5];
        } else {
            const auto it = split_tabs(line, 9);
            if (it.size() < 9) continue;
            if (!o.chr.empty() && !in_list_ci(it[0], o.chr)) continue;
            if (!o.feature.empty() && !in_list_ci(it[2], o.feature)) continue;
            f.chr = lower(it[0]); f.type = lower(it[2]);
            f.start = atoll(it[3].c_str()); f.end = atoll(it[4].c_str());
            if (it[6] == "+" || it[6] == "-") f.strand = it[6];
            f.label = gtf_tag_value(it[8], o.gtf_tag);
        }
        out.push_back(std::move(f));
    }
    gzclose(fh);
    return true;
}

}  // namespace

// subseq.go:53-347 on its streaming path (:301-343): per record SubSeq(start, end) for -r, or one
// output record per GTF / BED feature of that sequence, reverse-complemented on the minus strand.
// The slices are cut, reverse-complemented and printed on the GPU; the host only turns features
// into (record, start, end, strand) and their names.  Features of one sequence are emitted by
// type in order of first appearance in the file, then in file order (the reference walks a Go
// map of types there).
int cmd_subseq(const GlobalOpts& g, const SubseqOpts& o, const std::vector<std::string>& files)
{
    if (g.host_parse) return fatal("subseq has no --host-parse mode");
    if (o.only_flank) {                                                       // subseq.go:94-102
        if (o.up_stream > 0 && o.down_stream > 0)
            return fatal("when flag -f (--only-flank) given, only one of flags -u (--up-stream) and -d (--down-stream) is allowed");
        if (o.up_stream == 0 && o.down_stream == 0)
            return fatal("when flag -f (--only-flank) given, one of flags -u (--up-stream) and -d (--down-stream) should be given");
    }
    const bool by_region = !o.region.empty(), by_gtf = !by_region && !o.gtf.empty(), by_bed = !by_region && !by_gtf && !o.bed.empty();
    long long start = 0, end = 0;
    if (by_region) {                                                          // subseq.go:103-138
        if (o.up_stream > 0 || o.down_stream > 0 || o.only_flank)
            return fatal("when flag -r (--region) given, any of flags -u (--up-stream), -d (--down-stream) and -f (--only-flank) is not allowed");
        char *e1 = nullptr, *e2 = nullptr;
        start = strtoll(o.region.c_str(), &e1, 10);
        bool ok = e1 != o.region.c_str() && *e1 == ':';
        if (ok) { end = strtoll(e1 + 1, &e2, 10); ok = e2 != e1 + 1 && *e2 == 0; }
        if (!ok) return fatal("invalid region: " + o.region + ". type \"seqkit subseq -h\" for more examples");
        if (start == 0 || end == 0) return fatal("both start and end should not be 0");
        if (start < 0 && end > 0) return fatal("when start < 0, end should not > 0");
    }
    std::vector<Feature> feats;
    std::vector<std::pair<std::string, std::vector<size_t>>> by_chr;          // features of a sequence in output order
    if (by_gtf || by_bed) {
        if (by_bed && !o.feature.empty()) return fatal("when given flag -b (--bed), flag -f (--feature) is not allowed");
        if (!g.quiet) fprintf(stderr, by_gtf ? "[INFO] read GTF file ...\n" : "[INFO] read BED file ...\n");
        std::string err;
        if (!read_features(o, by_bed, feats, err)) return fatal(err);
        if (!g.quiet) fprintf(stderr, "[INFO] %zu %s features loaded\n", feats.size(), by_gtf ? "GTF" : "BED");
        std::vector<std::string> types;
        for (auto& f : feats) if (std::find(types.begin(), types.end(), f.type) == types.end()) types.push_back(f.type);
        std::vector<size_t> order(feats.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        auto rank = [&](size_t i) { return (size_t)(std::find(types.begin(), types.end(), feats[i].type) - types.begin()); };
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return rank(a) < rank(b); });
        for (size_t i : order) {
            auto it = std::find_if(by_chr.begin(), by_chr.end(), [&](const auto& kv) { return kv.first == feats[i].chr; });
            if (it == by_chr.end()) { by_chr.emplace_back(feats[i].chr, std::vector<size_t>()); it = by_chr.end() - 1; }
            it->second.push_back(i);
        }
        std::sort(by_chr.begin(), by_chr.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    }
    std::string flank;                                                        // subseq.go:434-452
    if (o.up_stream > 0) {
        if (o.only_flank) flank = "_usf:" + std::to_string(o.up_stream);
        else if (o.down_stream > 0) flank = "_us:" + std::to_string(o.up_stream) + "_ds:" + std::to_string(o.down_stream);
        else flank = "_us:" + std::to_string(o.up_stream);
    } else if (o.down_stream > 0) {
        flank = (o.only_flank ? "_dsf:" : "_ds:") + std::to_string(o.down_stream);
    }

    Ctx ctx;
    if (ctx.init(g) != 0) return fatal(skh_last_error());
    Sink out(g.out_file);
    if (!out.ok()) return fatal("open " + g.out_file + ": cannot create");
    std::vector<uint64_t> offs, npos, out_npos;
    std::vector<uint32_t> nlen, out_nlen;
    std::vector<skh_slice> slices;
    std::string names;
    for (const std::string& file : files) {
        BlockPipe tb(ctx.all, file, g.text_block_bytes);
        if (!tb.ok()) return fatal(tb.error());
        uint8_t lut[256];
        int ab = -1;
        skh_parse_result res;
        while (tb.next(res)) {
            skh_text* t = tb.block();
            if (res.nrec == 0) { tb.advance(res); continue; }
            skh_seq_format f;
            memset(&f, 0, sizeof f);
            f.min_len = f.max_len = -1;
            f.line_width = tb.is_fastq() ? 0 : g.line_width;                 // subseq.go:316-318
            f.print_name = 1; f.print_seq = 1; f.plus_line = 1;
            if (by_region) {
                PhaseTimer pt(2);
                if (skh_text_subseq(skh_text_ctx(t), t, nullptr, 0, start, end, 0, nullptr, nullptr, 0, nullptr, nullptr) != SKH_OK)
                    return fatal(skh_last_error());
                f.print_qual = tb.is_fastq(); f.mark = tb.is_fastq() ? '@' : '>';
                f.empty_as_fasta = 1;                                         // Record.FormatToWriter: layout per record
            } else if (by_gtf || by_bed) {
                if (ab < 0) {
                    std::string err;
                    ab = first_record_alphabet(skh_text_ctx(t), t, g, g.alphabet, err);
                    if (ab < 0) return fatal(err);
                    skh_pair_table(ab, lut);
                }
                offs.resize(res.nrec + 1); npos.resize(res.nrec); nlen.resize(res.nrec);
                if (skh_text_offsets(skh_text_ctx(t), t, 0, res.nrec + 1, offs.data()) != SKH_OK) return fatal(skh_last_error());
                if (skh_text_names(skh_text_ctx(t), t, 0, res.nrec, npos.data(), nlen.data()) != SKH_OK) return fatal(skh_last_error());
                slices.clear(); names.clear(); out_npos.clear(); out_nlen.clear();
                const uint8_t* text = skh_text_buffer(t);
                for (uint64_t r = 0; r < res.nrec; r++) {
                    const uint8_t* name = text + npos[r];
                    uint32_t ido = 0, idl = 0;
                    id_span(g, name, nlen[r], &ido, &idl);
                    const std::string id((const char*)name + ido, idl), key = lower(id);
                    auto it = std::lower_bound(by_chr.begin(), by_chr.end(), key, [](const auto& kv, const std::string& k) { return kv.first < k; });
                    if (it == by_chr.end() || it->first != key) continue;
                    const long long len = (long long)(offs[r + 1] - offs[r]);
                    for (size_t fi : it->second) {
                        const Feature& ft = feats[fi];
                        long long s = ft.start, e = ft.end;                   // subseq.go:379-420, 471-512
                        const bool minus = ft.strand == "-";
                        if (minus) {
                            if (o.only_flank) {
                                if (o.up_stream > 0) { s = ft.end + 1; e = ft.end + o.up_stream; }
                                else { s = ft.start - o.down_stream; e = ft.start - 1; }
                            } else { s = ft.start - o.down_stream; e = ft.end + o.up_stream; }
                        } else {
                            if (o.only_flank) {
                                if (o.up_stream > 0) { s = ft.start - o.up_stream; e = ft.start - 1; }
                                else { s = e + 1; e = e + o.down_stream; }
                            } else { s = ft.start - o.up_stream; e = ft.end + o.down_stream; }
                        }
                        if (s < 1) s = 1;
                        if (e > len) e = len;
                        skh_slice sl;
                        sl.record = r; sl.start = s; sl.end = e; sl.minus = minus ? 1 : 0; sl.reserved = 0;
                        slices.push_back(sl);
                        out_npos.push_back(names.size());
                        names += id; names += '_'; names += std::to_string(ft.start); names += '-';
                        names += std::to_string(ft.end); names += ':'; names += ft.strand.empty() ? "." : ft.strand;
                        names += flank; names += ' '; names += ft.label;
                        out_nlen.push_back((uint32_t)(names.size() - out_npos.back()));
                    }
                }
                if (slices.empty()) { tb.advance(res); continue; }
                PhaseTimer pt(2);
                if (skh_text_subseq(skh_text_ctx(t), t, slices.data(), slices.size(), 0, 0, 0, lut, (const uint8_t*)names.data(),
                                    names.size(), out_npos.data(), out_nlen.data()) != SKH_OK)
                    return fatal(skh_last_error());
                f.print_qual = 0; f.mark = '>';                               // fastx.NewRecord without qualities
            } else {
                tb.advance(res);                                              // no selection: nothing is printed
                continue;
            }
            const uint8_t* text = nullptr;
            uint64_t nbytes = 0;
            {
                PhaseTimer pt(3);
                if (skh_text_format_seq(skh_text_ctx(t), t, &f, &text, &nbytes) != SKH_OK) return fatal(skh_last_error());
            }
            { PhaseTimer pt(4); out.write(text, nbytes); }
            tb.advance(res);
        }
        if (!tb.error().empty()) return fatal(tb.error());
    }
    return 0;
}

// ------------------------------------------------------------------------------------ convert
namespace {

const char* quality_name(int e)
{
    static const char* names[] = {"Unknown", "Sanger", "Solexa", "Illumina-1.3+", "Illumina-1.5+", "Illumina-1.8+"};
    return e >= 0 && e < 6 ? names[e] : "Unknown";
}
int quality_offset(int e) { return (e == SKH_Q_SANGER || e == SKH_Q_ILLUMINA18) ? 33 : 64; }

// seq.GuessQualityEncoding: the encodings whose byte range holds the string, as a bit set; only
// Illumina-1.5+ when it fits and 'B' is among the n_most_common most frequent bytes
unsigned guess_quality_encoding(const uint8_t* q, size_t n, int n_most_common)
{
    if (n == 0) return 0;
    size_t cnt[256] = {0};
    int mn = 255, mx = 0;
    for (size_t i = 0; i < n; i++) { cnt[q[i]]++; mn = std::min<int>(mn, q[i]); mx = std::max<int>(mx, q[i]); }
    const bool fits15 = mn >= 66 && mx <= 105;
    if (fits15 && cnt['B'] > 0) {
        int more = 0;
        for (int c = 0; c < 256; c++) if (cnt[c] > cnt['B']) more++;
        if (more < n_most_common) return 1u << SKH_Q_ILLUMINA15;
    }
    unsigned set = 0;
    if (mn >= 33 && mx <= 73) set |= 1u << SKH_Q_SANGER;
    if (mn >= 59 && mx <= 104) set |= 1u << SKH_Q_SOLEXA;
    if (mn >= 64 && mx <= 104) set |= 1u << SKH_Q_ILLUMINA13;
    if (fits15) set |= 1u << SKH_Q_ILLUMINA15;
    if (mn >= 33 && mx <= 74) set |= 1u << SKH_Q_ILLUMINA18;
    return set;
}

// convert.go:100-139 / 185-225: the source encoding from the marks of the n leading records; 0 on error
int decide_encoding(const int marks[6], int n, int nrecords, double frac, std::string& err)
{
    std::vector<int> guessed;
    for (int e = 0; e < 6; e++) if (marks[e] == n) guessed.push_back(e);
    if ((double)(marks[SKH_Q_ILLUMINA15] / (nrecords + 1)) > frac) guessed.assign(1, SKH_Q_ILLUMINA15);   // integer division, as written
    std::string list;
    for (size_t i = 0; i < guessed.size(); i++) { if (i) list += ' '; list += quality_name(guessed[i]); }
    fprintf(stderr, "[INFO] possible quality encodings: [%s]\n", list.c_str());
    if (guessed.empty()) { err = "quality encoding not consistent"; return 0; }
    int from = guessed[0];
    if (guessed.size() > 1) {
        for (size_t i = 1; i < guessed.size(); i++)
            if ((guessed[i] == SKH_Q_SOLEXA) != (guessed[0] == SKH_Q_SOLEXA) || quality_offset(guessed[i]) != quality_offset(guessed[0])) {
                err = "fail to guess the source quality encoding, please specify it";
                return 0;
            }
        from = guessed.back();
        if (from == SKH_Q_ILLUMINA18) from = SKH_Q_SANGER;
    }
    fprintf(stderr, "[INFO] guessed quality encoding: %s\n", quality_name(from));
    return from;
}

bool encodings_match(int a, int b, bool force)                                  // convert.go:313-326
{
    if (a == b) return true;
    return !force && ((a == SKH_Q_SANGER && b == SKH_Q_ILLUMINA18) || (a == SKH_Q_ILLUMINA18 && b == SKH_Q_SANGER));
}

}  // namespace

// convert.go:42-275.  The source encoding is guessed from the first -n records of the input, then
// every block's quality plane is checked and looked up on the GPU and printed as FASTQ.  A file
// that can be opened twice is opened once for the guess (a small block that is grown until it
// holds the records) and then goes through the block pipeline; standard input is read once, its
// first block grown until it holds the records.  A file shorter than -n records is decided at its
// end, as in the reference; unlike the reference, its records are not printed again with a later file.
int cmd_convert(const GlobalOpts& g, const ConvertOpts& o, const std::vector<std::string>& files)
{
    if (g.host_parse) return fatal("convert has no --host-parse mode");
    bool guessing = o.from <= 0;
    int from = o.from, n = 0;
    int marks[6] = {0, 0, 0, 0, 0, 0};
    uint8_t lut[256], valid[256];
    int identity = 0;
    if (!guessing) {
        fprintf(stderr, "[INFO] converting %s -> %s\n", quality_name(from), quality_name(o.to));
        if (skh_quality_table(from, o.to, o.force, lut, valid, &identity) != SKH_OK) return fatal(skh_last_error());
    }
    Ctx ctx;
    if (ctx.init(g) != 0) return fatal(skh_last_error());
    Sink out(g.out_file);
    if (!out.ok()) return fatal("open " + g.out_file + ": cannot create");
    std::vector<uint64_t> offs;
    std::vector<uint8_t> qual;

    // convert.go:177-238 on the leading records of a parsed block
    auto needs_larger_block = [&](uint64_t nrec, bool at_eof) { return nrec <= (uint64_t)(o.nrecords - n) && !at_eof; };
    // `skip`: the input ended with exactly -n records looked at -- the reference decides on the record
    // after those (convert.go:184) or at the end of a shorter input (convert.go:99), so such a file
    // is neither decided nor printed
    auto guess_from = [&](skh_text* t, uint64_t nrec, bool at_eof, bool& skip) -> int {
        const uint64_t look = std::min<uint64_t>(nrec, (uint64_t)(o.nrecords - n));
        offs.resize(look + 1);
        if (skh_text_offsets(skh_text_ctx(t), t, 0, look + 1, offs.data()) != SKH_OK) return fatal(skh_last_error());
        qual.resize(offs[look] ? offs[look] : 1);
        if (offs[look] && skh_text_quality(skh_text_ctx(t), t, 0, offs[look], qual.data()) != SKH_OK) return fatal(skh_last_error());
        for (uint64_t r = 0; r < look; r++) {
            const unsigned set = guess_quality_encoding(qual.data() + offs[r], offs[r + 1] - offs[r], o.n_most_common);
            for (int e = 0; e < 6; e++) if (set & (1u << e)) marks[e]++;
        }
        n += (int)look;
        skip = at_eof && look == nrec && n == o.nrecords;
        if (skip) return 0;
        std::string err;
        from = decide_encoding(marks, n, o.nrecords, o.illumina15_frac, err);
        if (from == 0) return fatal(err);
        fprintf(stderr, "[INFO] converting %s -> %s\n", quality_name(from), quality_name(o.to));
        if (skh_quality_table(from, o.to, o.force, lut, valid, &identity) != SKH_OK) return fatal(skh_last_error());
        if (n >= o.nrecords) guessing = false;                             // else: decided at the end of a short file
        return 0;
    };

    // the blocks of one file, through either reader
    auto convert_blocks = [&](auto& tb) -> int {
        constexpr bool one_pass = std::is_same<typename std::decay<decltype(tb)>::type, TextBlocks>::value;
        bool once = true;
        std::string pending_error;
        skh_parse_result res;
        void* held = nullptr;                                               // pipeline: the block whose output is on its way
        auto flush_held = [&]() -> int {
            if constexpr (!one_pass) {
                if (!held) return 0;
                const uint8_t* text = nullptr;
                uint64_t nbytes = 0;
                {
                    PhaseTimer pt(3);
                    if (skh_text_output_wait(skh_text_ctx(BlockPipe::text_of(held)), BlockPipe::text_of(held), &text, &nbytes) != SKH_OK) return fatal(skh_last_error());
                }
                { PhaseTimer pt(4); out.write(text, nbytes); }
                tb.release(held);
                held = nullptr;
            }
            return 0;
        };
        auto one_block = [&](bool& stop) -> int {
            skh_text* t = tb.block();
            if (res.nrec == 0) { tb.advance(res); return 0; }
            if (!tb.is_fastq()) return fatal("this command only works for FASTQ format");
            if constexpr (one_pass) {
                if (guessing) {
                    if (needs_larger_block(res.nrec, tb.at_eof())) {
                        // records seen so far are looked at again in the larger block
                        if (!tb.grow()) return fatal(tb.error());
                        return 0;
                    }
                    bool skip = false;
                    if (int rc = guess_from(t, res.nrec, tb.at_eof(), skip)) return rc;
                    if (skip) { stop = true; return 0; }
                }
            }
            if (once && encodings_match(from, o.to, o.force)) {
                once = false;
                fprintf(stderr, "[WARN] source and target quality encoding match.\n");
            }
            if (o.dry_run) { stop = true; return 0; }
            if (!identity) {
                skh_validate_result vr;
                if (skh_text_validate_quality(skh_text_ctx(t), t, valid, &vr) != SKH_OK) return fatal(skh_last_error());
                if (vr.found) {
                    pending_error = std::string("seq: invalid ") + (from == SKH_Q_SOLEXA ? "Solexa" : "Phred") + " quality";
                    if (skh_text_truncate(skh_text_ctx(t), t, vr.record) != SKH_OK) return fatal(skh_last_error());
                    res.nrec = vr.record;
                    if (res.nrec == 0) { stop = true; return 0; }
                }
                PhaseTimer pt(2);
                if (skh_text_map_quality(skh_text_ctx(t), t, lut) != SKH_OK) return fatal(skh_last_error());
            }
            skh_seq_format f;
            memset(&f, 0, sizeof f);
            f.min_len = f.max_len = -1;
            f.line_width = 0;                                               // convert.go:171-172
            f.print_name = 1; f.print_seq = 1; f.print_qual = 1; f.plus_line = 1; f.mark = '@';
            f.empty_as_fasta = 1;                                           // Record.FormatToWriter: layout per record
            if constexpr (one_pass) {
                const uint8_t* text = nullptr;
                uint64_t nbytes = 0;
                {
                    PhaseTimer pt(3);
                    if (skh_text_format_seq(skh_text_ctx(t), t, &f, &text, &nbytes) != SKH_OK) return fatal(skh_last_error());
                }
                { PhaseTimer pt(4); out.write(text, nbytes); }
                if (pending_error.empty()) tb.advance(res);
            } else {
                // the download of this block's output runs beside the next block's kernels
                {
                    PhaseTimer pt(3);
                    if (skh_text_format_seq_submit(skh_text_ctx(t), t, &f) != SKH_OK) return fatal(skh_last_error());
                }
                if (int rc = flush_held()) return rc;
                held = tb.hold();
            }
            return 0;
        };
        bool stop = false;
        while (!stop && pending_error.empty() && tb.next(res)) {
            if (int rc = one_block(stop)) { flush_held(); return rc; }
        }
        if (int rc = flush_held()) return rc;
        if (!pending_error.empty()) { out.flush(); return fatal(pending_error); }
        if (!tb.error().empty()) return fatal(tb.error());
        return 0;
    };

    for (const std::string& file : files) {
        struct stat st;
        const bool opens_twice = file != "-" && stat(file.c_str(), &st) == 0 && S_ISREG(st.st_mode);
        if (!opens_twice) {
            TextBlocks tb(ctx.c, file, g.text_block_bytes);
            if (!tb.ok()) return fatal(tb.error());
            if (int rc = convert_blocks(tb)) return rc;
            continue;
        }
        bool skip = false;
        if (guessing) {
            TextBlocks tb(ctx.c, file, std::min<uint64_t>(g.text_block_bytes, 4u << 20));
            if (!tb.ok()) return fatal(tb.error());
            skh_parse_result res;
            while (tb.next(res)) {
                if (res.nrec == 0) { tb.advance(res); continue; }
                if (!tb.is_fastq()) return fatal("this command only works for FASTQ format");
                if (needs_larger_block(res.nrec, tb.at_eof())) {
                    if (!tb.grow()) return fatal(tb.error());
                    continue;
                }
                if (int rc = guess_from(tb.block(), res.nrec, tb.at_eof(), skip)) return rc;
                break;
            }
            if (!tb.error().empty()) return fatal(tb.error());
        }
        if (skip) continue;
        BlockPipe tb(ctx.all, file, g.text_block_bytes);
        if (!tb.ok()) return fatal(tb.error());
        if (int rc = convert_blocks(tb)) return rc;
    }
    return 0;
}

}  // namespace host
