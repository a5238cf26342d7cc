// fastx_reader.cpp -- FASTA/FASTQ text -> CSR batches in pinned memory.
//
// Behaviour follows what the reference's commands observe from their reader
// (call sites seqkit/cmd/seq.go:115-127, fx2tab.go:96-106, stat.go:87-104): the format is
// decided by the first non-blank byte, a FASTA record runs to the next line starting with
// '>', line ends (and only line ends) are dropped from the sequence, the head line keeps
// everything after '>' / '@', the ID is the head up to the first white space, and the
// alphabet is guessed once, from the first record.
#include <zlib.h>

#include <cstring>

#include "host.h"

namespace host {

static const struct { const char* name; const char* letters; } kAlpha[6] = {
    {"DNA", "acgtACGT -.nN"},
    {"DNAredundant", "acgtryswkmbdhvACGTRYSWKMBDHV -.nN"},
    {"RNA", "acguACGU -.nN"},
    {"RNAredundant", "acguryswkmbdhvACGURYSWKMBDHV -.nN"},
    {"Protein", "abcdefghijklmnpqrstvwyzABCDEFGHIJKLMNPQRSTVWYZ*_. -xX"},
    {"Unlimit", ""},
};

const char* alphabet_name(int a) { return (a >= 0 && a < 6) ? kAlpha[a].name : ""; }

int guess_alphabet(const uint8_t* seq, size_t n, size_t threshold)
{
    if (n == 0) return Unlimit;
    if (threshold != 0 && n > threshold) n = threshold;
    bool present[256] = {false};
    for (size_t i = 0; i < n; i++) present[seq[i]] = true;
    static const int order[5] = {DNA, RNA, DNAredundant, RNAredundant, Protein};
    for (int a : order) {
        bool member[256] = {false};
        for (const char* p = kAlpha[a].letters; *p; p++) member[(uint8_t)*p] = true;
        bool subset = true;
        for (int b = 0; b < 256 && subset; b++) subset = !present[b] || member[b];
        if (subset) return a == DNA ? DNAredundant : (a == RNA ? RNAredundant : a);
    }
    return Unlimit;
}

bool ensure_capacity(skh_ctx* ctx, Batch& bt, uint64_t used, uint64_t bytes, uint64_t records, bool with_qual,
                     std::string& err)
{
    const uint64_t have_b = bt.b ? skh_batch_max_bytes(bt.b) : 0, have_r = bt.b ? skh_batch_max_records(bt.b) : 0;
    const bool have_q = bt.b && skh_batch_qual(bt.b) != nullptr;
    if (bt.b && bytes <= have_b && records <= have_r && (!with_qual || have_q)) return true;
    uint64_t nb = have_b ? have_b : (1u << 20), nr = have_r ? have_r : 4096;
    while (nb < bytes) nb *= 2;
    while (nr < records) nr *= 2;
    skh_batch* fresh = nullptr;
    if (skh_batch_alloc(ctx, nb, nr, with_qual || have_q, &fresh) != SKH_OK) {
        err = skh_last_error();
        return false;
    }
    if (bt.b) {
        memcpy(skh_batch_seq(fresh), bt.seq(), used);
        if (have_q) memcpy(skh_batch_qual(fresh), bt.qual(), used);
        memcpy(skh_batch_offsets(fresh), bt.offs(), (bt.nrec + 1) * sizeof(uint64_t));
        skh_batch_free(bt.b);
    }
    bt.b = fresh;
    return true;
}

FastxReader::FastxReader(skh_ctx* ctx, const std::string& path, int forced_alphabet, size_t guess_len)
    : ctx_(ctx), forced_alphabet_(forced_alphabet), guess_len_(guess_len)
{
    gzFile f = path == "-" ? gzdopen(0, "rb") : gzopen(path.c_str(), "rb");
    if (!f) { err_ = "open " + path + ": no such file or directory"; return; }
    gzbuffer(f, 1 << 20);
    fh_ = f;
    buf_.resize(8 << 20);
}

FastxReader::~FastxReader() { if (fh_) gzclose((gzFile)fh_); }

bool FastxReader::fill()
{
    if (eof_) return false;
    const int n = gzread((gzFile)fh_, buf_.data(), (unsigned)buf_.size());
    pos_ = 0;
    end_ = n > 0 ? (size_t)n : 0;
    if (n <= 0) eof_ = true;
    return n > 0;
}

bool FastxReader::read_line(const uint8_t*& p, size_t& n)
{
    if (have_line_) { have_line_ = false; p = line_; n = line_len_; return true; }
    carry_.clear();
    bool in_carry = false;
    for (;;) {
        if (pos_ == end_ && !fill()) {
            if (!in_carry) return false;
            break;                                   // last line without '\n'
        }
        const uint8_t* start = buf_.data() + pos_;
        const uint8_t* nl = (const uint8_t*)memchr(start, '\n', end_ - pos_);
        if (nl) {
            const size_t len = (size_t)(nl - start);
            pos_ += len + 1;
            if (!in_carry) { line_ = start; line_len_ = len; }
            else { carry_.insert(carry_.end(), start, start + len); }
            goto done;
        }
        carry_.insert(carry_.end(), start, (const uint8_t*)buf_.data() + end_);
        in_carry = true;
        pos_ = end_;
    }
done:
    if (in_carry) { line_ = carry_.data(); line_len_ = carry_.size(); }
    while (line_len_ && line_[line_len_ - 1] == '\r') line_len_--;
    p = line_; n = line_len_;
    return true;
}

static uint32_t id_length(const uint8_t* name, size_t n)
{
    size_t i = 0;
    while (i < n && !(name[i] == ' ' || (name[i] >= '\t' && name[i] <= '\r' && name[i] != '\v'))) i++;
    return (uint32_t)(i == 0 ? n : i);
}

bool FastxReader::next_batch(Batch& bt, uint64_t max_bytes, uint64_t max_records)
{
    bt.nrec = 0;
    bt.names.clear();
    bt.name_off.assign(1, 0);
    bt.id_len.clear();
    if (!fh_) return false;
    if (!bt.b && !ensure_capacity(ctx_, bt, 0, 1 << 20, 4096, false, err_)) return false;
    bt.offs()[0] = 0;
    uint64_t used = 0;
    const uint8_t* l; size_t n;
    for (;;) {
        if (bt.nrec && (used >= max_bytes || bt.nrec >= max_records)) break;
        // head line (blank lines between records are skipped)
        bool got = false;
        while ((got = read_line(l, n)) && n == 0) {}
        if (!got) break;
        if (!format_known_) {
            if (l[0] == '>') is_fastq_ = false;
            else if (l[0] == '@') is_fastq_ = true;
            else { err_ = "invalid FASTA/Q format"; return false; }
            format_known_ = true;
        }
        if (l[0] != (is_fastq_ ? '@' : '>')) { err_ = "invalid FASTA/Q format"; return false; }
        if (!ensure_capacity(ctx_, bt, used, used + 1, bt.nrec + 2, is_fastq_, err_)) return false;
        bt.names.insert(bt.names.end(), l + 1, l + n);
        bt.name_off.push_back(bt.names.size());
        bt.id_len.push_back(id_length(l + 1, n - 1));
        const uint64_t start = used;
        if (!is_fastq_) {
            while (read_line(l, n)) {
                if (n && l[0] == '>') { unread_line(); break; }
                if (!n) continue;
                if (used + n > skh_batch_max_bytes(bt.b)) {
                    // the line may live in the reader's buffers: growing the batch does not move it
                    if (!ensure_capacity(ctx_, bt, used, used + n, bt.nrec + 2, false, err_)) return false;
                }
                memcpy(bt.seq() + used, l, n);
                used += n;
            }
        } else {
            while (read_line(l, n)) {
                if (n && l[0] == '+') break;
                if (!n) continue;
                if (used + n > skh_batch_max_bytes(bt.b) && !ensure_capacity(ctx_, bt, used, used + n, bt.nrec + 2, true, err_))
                    return false;
                memcpy(bt.seq() + used, l, n);
                used += n;
            }
            uint64_t q = start;
            while (q < used && read_line(l, n)) {
                if (q + n > used) { err_ = "quality longer than sequence"; return false; }
                memcpy(bt.qual() + q, l, n);
                q += n;
            }
            if (q != used) { err_ = "quality shorter than sequence"; return false; }
        }
        if (alphabet_ < 0)
            alphabet_ = forced_alphabet_ >= 0 ? forced_alphabet_
                                              : guess_alphabet(bt.seq() + start, used - start, guess_len_);
        bt.nrec++;
        bt.offs()[bt.nrec] = used;
    }
    return bt.nrec != 0;
}

}  // namespace host
