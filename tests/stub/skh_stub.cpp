// skh_stub.cpp -- TEST INFRASTRUCTURE, never shipped: the C ABI of include/seqkit_hip.h restated on the CPU.
//
// Exists for one purpose: to run the host side of the driver (seqkit_amd/host/*.cpp: block pipeline,
// tail carry, multi-line FASTQ rewrite, multi-device dealing, output ordering) under AddressSanitizer /
// UBSan / ThreadSanitizer and in the CPU test suite, where no GPU exists.  Every entry point the driver
// calls is written as plain serial code with the semantics the header documents; the per-record
// arithmetic is the oracle's (oracle/seqkit_oracle.c).  Nothing under seqkit_amd/ links, loads or
// knows about this file; the product has no CPU path.
//
// "Devices": skh_init accepts ordinals below SKH_STUB_DEVICES (environment, default 1).
#include "seqkit_hip.h"

#include <math.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" {
#include "../../oracle/seqkit_oracle.h"
}

namespace {

thread_local std::string g_err;
int fail(int code, const char* msg) { g_err = msg; return code; }

uint32_t id_length(const uint8_t* name, uint32_t n)
{
    uint32_t i = 0;
    while (i < n && !(name[i] == ' ' || (name[i] >= '\t' && name[i] <= '\r' && name[i] != '\v'))) i++;
    return i == 0 ? n : i;
}


void put_wrapped(std::vector<uint8_t>& out, const uint8_t* s, uint64_t n, uint32_t width)
{
    if (width < 1 || n == 0) { out.insert(out.end(), s, s + n); return; }
    for (uint64_t a = 0; a < n; a += width) {
        if (a) out.push_back('\n');
        out.insert(out.end(), s + a, s + std::min<uint64_t>(n, a + width));
    }
}

}  // namespace

struct skh_ctx { int device = 0; };

struct skh_batch {
    skh_ctx* ctx = nullptr;
    uint64_t max_bytes = 0, max_records = 0;
    std::vector<uint8_t> seq, qual;
    std::vector<uint64_t> offs;
    bool with_qual = false;
};

struct skh_text {
    skh_ctx* ctx = nullptr;
    uint64_t cap = 0;
    std::vector<uint8_t> h_text;                 // what skh_text_buffer() hands out
    std::vector<uint8_t> stage;                  // skh_text_upload: "device" copy at the same offsets
    uint64_t up_off = 0, up_bytes = 0;
    // the parsed block ("device" side)
    std::vector<uint8_t> text, seq, qual;
    std::vector<uint64_t> offs, npos;
    std::vector<uint32_t> nlen;
    std::vector<uint64_t> filter_offs;           // lengths before a gap removal (empty: offs)
    bool names_external = false;                 // names of a subseq result: positions are not in h_text
    int fastq = 0;
    uint64_t nrec = 0, text_off = 0;
    std::vector<uint64_t> counts[SKH_TAB_MAX_PASSES];
    uint32_t nclass[SKH_TAB_MAX_PASSES] = {0, 0, 0, 0};
    std::vector<uint32_t> letters;
    bool letters_ready = false, transformed = false, gaps_removed = false;
    std::vector<uint8_t> out;
    uint64_t out_pending = 0;
};

extern "C" {

const char* skh_last_error(void) { return g_err.c_str(); }
const char* skh_version(void) { return "seqkit-hip test stub (CPU)"; }
uint64_t skh_padded_size(uint64_t n) { return (n + 15) & ~15ull; }

int skh_init(int device, skh_ctx** out)
{
    if (!out) return fail(SKH_EINVAL, "skh_init: out is NULL");
    const char* e = getenv("SKH_STUB_DEVICES");
    const int n = e && *e ? atoi(e) : 1;
    if (device < 0 || device >= n) return fail(SKH_ENODEV, "device ordinal out of range");
    *out = new skh_ctx;
    (*out)->device = device;
    return SKH_OK;
}
void skh_destroy(skh_ctx* c) { delete c; }
int skh_sync(skh_ctx*) { return SKH_OK; }

int skh_pair_table(int alphabet, uint8_t lut[256])
{
    if (alphabet < 0 || alphabet > 5) return fail(SKH_EINVAL, "unknown alphabet id");
    ora_pair_table(alphabet, lut);
    return SKH_OK;
}
int skh_valid_table(int alphabet, uint8_t valid[256])
{
    if (alphabet < 0 || alphabet > 5) return fail(SKH_EINVAL, "unknown alphabet id");
    ora_alphabet_members(alphabet, valid);
    return SKH_OK;
}
int skh_quality_table(int from, int to, int force, uint8_t lut[256], uint8_t valid[256], int* identity)
{
    const int same = ora_quality_table(from, to, force, lut, valid);
    if (same < 0) return fail(SKH_EINVAL, "unknown quality encoding");
    if (identity) *identity = same;
    return SKH_OK;
}
int skh_class_table(const char* const* lists, uint32_t nlists, uint8_t class_of[256])
{
    if (nlists > SKH_MAX_CLASSES) return fail(SKH_EINVAL, "more than SKH_MAX_CLASSES lists");
    memset(class_of, SKH_NO_CLASS, 256);
    for (uint32_t i = 0; i < nlists; i++)
        for (const char* p = lists[i]; *p; p++) {
            const uint8_t b = (uint8_t)*p;
            const uint8_t both[2] = {(uint8_t)toupper(b), (uint8_t)tolower(b)};
            for (uint8_t v : both) {
                if (class_of[v] != SKH_NO_CLASS && class_of[v] != i) return fail(SKH_EINVAL, "a letter appears in two class lists");
                class_of[v] = (uint8_t)i;
            }
        }
    return SKH_OK;
}

void skh_stats_init(skh_stats_partial* p)
{
    p->num = p->len_sum = p->gap_sum = p->len_max = 0;
    p->len_min = INT64_MAX;
}
void skh_stats_merge(skh_stats_partial* into, const skh_stats_partial* from)
{
    into->num += from->num; into->len_sum += from->len_sum; into->gap_sum += from->gap_sum;
    into->len_min = std::min(into->len_min, from->len_min);
    into->len_max = std::max(into->len_max, from->len_max);
}

}  // extern "C"

namespace {

void stats_of(const uint8_t* seq, const uint64_t* offs, uint64_t nrec, const uint8_t is_gap[256], skh_stats_partial* out)
{
    std::vector<uint8_t> letters;
    for (int b = 0; b < 256; b++) if (is_gap[b]) letters.push_back((uint8_t)b);
    int64_t v[5];
    ora_batch_stats(seq, offs, nrec, letters.data(), letters.size(), v);
    out->num = v[0]; out->len_sum = v[1]; out->gap_sum = v[2]; out->len_min = v[3]; out->len_max = v[4];
}

void validate_of(const uint8_t* seq, const uint64_t* offs, uint64_t nrec, const uint8_t valid[256], uint64_t limit,
                 skh_validate_result* out)
{
    memset(out, 0, sizeof *out);
    for (uint64_t r = 0; r < nrec; r++) {
        uint64_t n = offs[r + 1] - offs[r];
        if (limit && n > limit) n = limit;
        for (uint64_t i = 0; i < n; i++)
            if (!valid[seq[offs[r] + i]]) {
                out->record = r; out->offset = i; out->letter = seq[offs[r] + i]; out->found = 1;
                return;
            }
    }
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------- pinned batches (--host-parse)

int skh_batch_alloc(skh_ctx* ctx, uint64_t max_bytes, uint64_t max_records, int with_qual, skh_batch** out)
{
    if (!ctx || !out) return fail(SKH_EINVAL, "NULL argument");
    skh_batch* b = new skh_batch;
    b->ctx = ctx;
    b->max_bytes = skh_padded_size(max_bytes ? max_bytes : 16);
    b->max_records = max_records ? max_records : 1;
    b->seq.assign(b->max_bytes, 0);
    b->with_qual = with_qual != 0;
    if (b->with_qual) b->qual.assign(b->max_bytes, 0);
    b->offs.assign(b->max_records + 1, 0);
    *out = b;
    return SKH_OK;
}
void skh_batch_free(skh_batch* b) { delete b; }
uint8_t* skh_batch_seq(skh_batch* b) { return b ? b->seq.data() : nullptr; }
uint8_t* skh_batch_qual(skh_batch* b) { return b && b->with_qual ? b->qual.data() : nullptr; }
uint64_t* skh_batch_offsets(skh_batch* b) { return b ? b->offs.data() : nullptr; }
uint64_t skh_batch_max_bytes(const skh_batch* b) { return b ? b->max_bytes : 0; }
uint64_t skh_batch_max_records(const skh_batch* b) { return b ? b->max_records : 0; }

static int batch_check(const skh_batch* b, uint64_t nrec)
{
    if (!b) return fail(SKH_EINVAL, "batch is NULL");
    if (nrec > b->max_records) return fail(SKH_ECAPACITY, "more records than the batch was allocated for");
    if (b->offs[nrec] > b->max_bytes) return fail(SKH_ECAPACITY, "more sequence bytes than the batch was allocated for");
    return SKH_OK;
}

int skh_revcomp_submit(skh_ctx*, skh_batch* b, uint64_t nrec, const uint8_t* lut, int do_reverse, int reverse_qual)
{
    if (int rc = batch_check(b, nrec)) return rc;
    ora_batch_transform(b->seq.data(), (b->with_qual && reverse_qual && do_reverse) ? b->qual.data() : nullptr,
                        b->offs.data(), nrec, lut, do_reverse);
    return SKH_OK;
}
int skh_revcomp(skh_ctx* c, skh_batch* b, uint64_t nrec, const uint8_t* lut, int do_reverse, int reverse_qual)
{
    return skh_revcomp_submit(c, b, nrec, lut, do_reverse, reverse_qual);
}
int skh_count_submit(skh_ctx*, skh_batch* b, uint64_t nrec, const uint8_t class_of[256], uint32_t nclass, uint64_t* counts)
{
    if (int rc = batch_check(b, nrec)) return rc;
    if (nclass < 1 || nclass > SKH_MAX_CLASSES) return fail(SKH_EINVAL, "nclass must be 1..SKH_MAX_CLASSES");
    if (nrec) ora_batch_count(b->seq.data(), b->offs.data(), nrec, class_of, nclass, counts);
    return SKH_OK;
}
int skh_count(skh_ctx* c, skh_batch* b, uint64_t nrec, const uint8_t class_of[256], uint32_t nclass, uint64_t* counts)
{
    return skh_count_submit(c, b, nrec, class_of, nclass, counts);
}
int skh_batch_wait(skh_batch* b) { return b ? SKH_OK : fail(SKH_EINVAL, "batch is NULL"); }
int skh_stats_reduce(skh_ctx*, skh_batch* b, uint64_t nrec, const uint8_t is_gap[256], skh_stats_partial* inout)
{
    if (int rc = batch_check(b, nrec)) return rc;
    if (nrec == 0) return SKH_OK;
    skh_stats_partial p;
    stats_of(b->seq.data(), b->offs.data(), nrec, is_gap, &p);
    skh_stats_merge(inout, &p);
    return SKH_OK;
}
int skh_validate(skh_ctx*, skh_batch* b, uint64_t nrec, const uint8_t valid[256], uint64_t limit, skh_validate_result* out)
{
    if (int rc = batch_check(b, nrec)) return rc;
    validate_of(b->seq.data(), b->offs.data(), nrec, valid, limit, out);
    return SKH_OK;
}

// ---------------------------------------------------------------- text blocks

int skh_text_alloc(skh_ctx* ctx, uint64_t block_bytes, skh_text** out)
{
    if (!ctx || !out) return fail(SKH_EINVAL, "NULL argument");
    skh_text* t = new skh_text;
    t->ctx = ctx;
    t->cap = skh_padded_size(block_bytes < 4096 ? 4096 : block_bytes);
    t->h_text.assign(t->cap, 0);
    *out = t;
    return SKH_OK;
}
void skh_text_free(skh_text* t) { delete t; }
uint8_t* skh_text_buffer(skh_text* t) { return t ? t->h_text.data() : nullptr; }
uint64_t skh_text_capacity(const skh_text* t) { return t ? t->cap : 0; }
skh_ctx* skh_text_ctx(const skh_text* t) { return t ? t->ctx : nullptr; }

int skh_text_upload(skh_ctx* cx, skh_text* t, uint64_t offset, uint64_t nbytes)
{
    if (!t) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (offset > t->cap || nbytes > t->cap - offset) return fail(SKH_ECAPACITY, "range beyond the block's buffer");
    t->up_bytes = 0;
    if (nbytes == 0) return SKH_OK;
    if (t->stage.size() < t->cap) t->stage.assign(t->cap, 0);
    memcpy(t->stage.data() + offset, t->h_text.data() + offset, nbytes);      // the bytes as they were at this moment
    t->up_off = offset; t->up_bytes = nbytes;
    return SKH_OK;
}

int skh_text_parse_at(skh_ctx* cx, skh_text* t, uint64_t off, uint64_t n, int format, int final, skh_parse_result* res)
{
    if (!t || !res) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (off > t->cap || n > t->cap - off) return fail(SKH_ECAPACITY, "text block larger than its buffer");
    if (format != SKH_FASTA && format != SKH_FASTQ && format != SKH_FASTQ_WRAPPED)
        return fail(SKH_EINVAL, "format must be SKH_FASTA, SKH_FASTQ or SKH_FASTQ_WRAPPED");
    // the "device" text: what was uploaded ahead plus the bytes around it, as the library composes it
    t->text.assign(n, 0);
    const uint64_t end = off + n;
    if (t->up_bytes && t->up_off >= off && t->up_off + t->up_bytes <= end) {
        memcpy(t->text.data(), t->h_text.data() + off, t->up_off - off);
        memcpy(t->text.data() + (t->up_off - off), t->stage.data() + t->up_off, t->up_bytes);
        const uint64_t up_end = t->up_off + t->up_bytes;
        memcpy(t->text.data() + (up_end - off), t->h_text.data() + up_end, end - up_end);
    } else if (n) {
        memcpy(t->text.data(), t->h_text.data() + off, n);
    }
    t->up_bytes = 0;
    t->text_off = off;
    const uint8_t* x = t->text.data();
    // line table: start, length without trailing CRs
    struct Line { uint64_t start; uint32_t len; };
    std::vector<Line> lines;
    for (uint64_t p = 0; p < n;) {
        const uint8_t* nl = static_cast<const uint8_t*>(memchr(x + p, '\n', n - p));
        if (!nl && !final) break;
        uint64_t e = nl ? (uint64_t)(nl - x) : n;
        const uint64_t next = nl ? e + 1 : n;
        while (e > p && x[e - 1] == '\r') e--;
        lines.push_back({p, (uint32_t)(e - p)});
        p = next;
    }
    memset(res, 0, sizeof *res);
    res->error_line = -1;
    res->lines = lines.size();
    t->seq.clear(); t->qual.clear(); t->offs.clear(); t->npos.clear(); t->nlen.clear();
    uint64_t nrec = 0, consumed = 0;
    int64_t bad = -1;
    const uint64_t L = lines.size();
    auto start_of_line = [&](uint64_t k) -> uint64_t {
        if (k < L) return lines[k].start;
        if (L && lines[L - 1].start + lines[L - 1].len == n && (n == 0 || x[n - 1] != '\n')) return n;   // unterminated last line
        if (!L) return 0;
        uint64_t e = lines[L - 1].start + lines[L - 1].len;
        while (e < n && x[e] == '\r') e++;
        return e < n ? e + 1 : n;
    };
    if (format == SKH_FASTA) {
        std::vector<uint64_t> heads;
        for (uint64_t i = 0; i < L; i++) {
            if (lines[i].len && x[lines[i].start] == '>') heads.push_back(i);
            else if (heads.empty() && lines[i].len && bad < 0) bad = (int64_t)i;
        }
        const uint64_t H = heads.size();
        uint64_t last_line = L;
        if (final) { nrec = H; consumed = n; }
        else if (H > 1) { nrec = H - 1; last_line = heads[H - 1]; consumed = lines[last_line].start; }
        if (bad >= 0 && nrec == 0 && !final) bad = -1;                // nothing accepted: nothing reported yet
        if (bad < 0 || nrec) {
            for (uint64_t r = 0; r < nrec; r++) {
                const uint64_t h = heads[r], stop = r + 1 < H ? heads[r + 1] : L;
                t->offs.push_back(t->seq.size());
                t->npos.push_back(lines[h].start + 1);
                t->nlen.push_back(lines[h].len - 1);
                for (uint64_t i = h + 1; i < stop && i < last_line; i++)
                    t->seq.insert(t->seq.end(), x + lines[i].start, x + lines[i].start + lines[i].len);
            }
        }
    } else if (format == SKH_FASTQ_WRAPPED) {
        // records wrapped over several lines, blank lines between records: a serial walk over the line table by the
        // rules of the library's header; what is not such text is a format error without a line number
        uint64_t i = 0;
        bool broken = false, wrapped = false;
        consumed = final ? n : start_of_line(L);
        uint64_t lines_used = L;
        for (;;) {
            const uint64_t i_before = i;
            while (i < L && lines[i].len == 0) i++;
            if (i == L) break;
            const uint64_t start = i;
            auto cut = [&]() { if (final) broken = true; else { consumed = lines[start].start; lines_used = start; } };
            if (x[lines[i].start] != '@') { broken = true; break; }
            uint64_t j = i + 1;
            while (j < L && !(lines[j].len && x[lines[j].start] == '+')) j++;
            if (j == L) {
                uint64_t rest = 0;
                for (uint64_t a = i + 1; a < L; a++) rest += lines[a].len;
                if (!(final && rest == 0)) { cut(); break; }
                // a head with nothing but blank lines up to the end of the file: a record without bases
                t->offs.push_back(t->seq.size());
                t->npos.push_back(lines[start].start + 1);
                t->nlen.push_back(lines[start].len - 1);
                wrapped = true;
                nrec++;
                break;
            }
            uint64_t S = 0, Q = 0, k = j;
            for (uint64_t a = i + 1; a < j; a++) S += lines[a].len;
            while (Q < S && k + 1 < L) { k++; Q += lines[k].len; }
            if (Q < S) { cut(); break; }
            if (Q != S) { broken = true; break; }
            t->offs.push_back(t->seq.size());
            t->npos.push_back(lines[start].start + 1);
            t->nlen.push_back(lines[start].len - 1);
            for (uint64_t a = i + 1; a < j; a++) t->seq.insert(t->seq.end(), x + lines[a].start, x + lines[a].start + lines[a].len);
            for (uint64_t a = j + 1; a <= k; a++) t->qual.insert(t->qual.end(), x + lines[a].start, x + lines[a].start + lines[a].len);
            if (j != start + 2 || k != j + 1 || start != i_before || (nrec == 0 && start != 0)) wrapped = true;
            nrec++;
            i = k + 1;
        }
        if (broken) { bad = 0; }
        res->wrapped = wrapped ? 1 : 0;
        res->lines = lines_used;
    } else {
        uint64_t Leff = L, fbh = ~0ull, last_nonblank = 0;
        bool any_nonblank = false;
        int64_t err = -1;
        for (uint64_t i = 0; i < L; i++) {
            const uint32_t role = (uint32_t)(i & 3);
            const uint8_t c = lines[i].len ? x[lines[i].start] : 0;
            bool b = false;
            if (role == 0) { if (lines[i].len == 0) { if (fbh == ~0ull) fbh = i; } else b = c != '@'; }
            else if (role == 2) b = c != '+';
            else if (role == 1) b = lines[i].len != 0 && c == '+';
            else if (role == 3) b = lines[i].len != lines[i - 2].len;
            if (lines[i].len) { last_nonblank = i; any_nonblank = true; }
            if (b && err < 0) err = (int64_t)i;
        }
        if (fbh != ~0ull && fbh < Leff) {
            Leff = fbh;
            if (final && any_nonblank && last_nonblank > fbh) bad = (int64_t)fbh;
        }
        nrec = Leff / 4;
        if (final && (Leff & 3) != 0 && bad < 0) bad = (int64_t)(nrec * 4);
        if (err >= 0 && ((uint64_t)err < nrec * 4 || final) && !(bad >= 0 && bad < err)) bad = err;
        if (bad >= 0) nrec = 0;
        else {
            consumed = final ? n : start_of_line(nrec * 4);
            for (uint64_t r = 0; r < nrec; r++) {
                const Line &h = lines[4 * r], &s = lines[4 * r + 1], &q = lines[4 * r + 3];
                t->offs.push_back(t->seq.size());
                t->npos.push_back(h.start + 1);
                t->nlen.push_back(h.len - 1);
                t->seq.insert(t->seq.end(), x + s.start, x + s.start + s.len);
                t->qual.insert(t->qual.end(), x + q.start, x + q.start + q.len);
            }
        }
    }
    if (bad >= 0) {
        res->status = SKH_PARSE_EFORMAT; res->error_line = format == SKH_FASTQ_WRAPPED ? -1 : bad;
        t->seq.clear(); t->qual.clear(); t->offs.clear(); t->npos.clear(); t->nlen.clear();
        nrec = 0;
    }
    t->offs.push_back(t->seq.size());
    res->nrec = nrec; res->seq_bytes = t->seq.size(); res->consumed = consumed;
    t->fastq = format != SKH_FASTA;
    t->nrec = nrec;
    t->transformed = t->gaps_removed = t->letters_ready = t->names_external = false;
    t->filter_offs.clear();
    return SKH_OK;
}
int skh_text_parse(skh_ctx* c, skh_text* t, uint64_t n, int format, int final, skh_parse_result* out)
{
    return skh_text_parse_at(c, t, 0, n, format, final, out);
}

int skh_text_offsets(skh_ctx* cx, skh_text* t, uint64_t first, uint64_t count, uint64_t* offsets)
{
    if (!t || !offsets) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (first + count > t->nrec + 1) return fail(SKH_EINVAL, "offset range beyond the block's records");
    memcpy(offsets, t->offs.data() + first, count * 8);
    return SKH_OK;
}
int skh_text_sequence(skh_ctx* cx, skh_text* t, uint64_t first_byte, uint64_t count, uint8_t* dst)
{
    if (!t || (!dst && count)) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (first_byte + count > t->offs[t->nrec]) return fail(SKH_EINVAL, "byte range beyond the block's sequence");
    if (count) memcpy(dst, t->seq.data() + first_byte, count);
    return SKH_OK;
}
int skh_text_quality(skh_ctx* cx, skh_text* t, uint64_t first_byte, uint64_t count, uint8_t* dst)
{
    if (!t || (!dst && count)) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (!t->fastq) return fail(SKH_EINVAL, "a FASTA block has no qualities");
    if (first_byte + count > t->offs[t->nrec]) return fail(SKH_EINVAL, "byte range beyond the block's qualities");
    if (count) memcpy(dst, t->qual.data() + first_byte, count);
    return SKH_OK;
}
int skh_text_revcomp(skh_ctx* cx, skh_text* t, const uint8_t* lut, int do_reverse, int reverse_qual)
{
    if (!t) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (t->transformed) return fail(SKH_EINVAL, "block was already transformed");
    if (t->nrec == 0 || (!lut && !do_reverse)) return SKH_OK;
    ora_batch_transform(t->seq.data(), (t->fastq && do_reverse && reverse_qual) ? t->qual.data() : nullptr,
                        t->offs.data(), t->nrec, lut, do_reverse);
    t->transformed = true;
    return SKH_OK;
}
int skh_text_map_quality(skh_ctx* cx, skh_text* t, const uint8_t lut[256])
{
    if (!t || !lut) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (!t->fastq) return fail(SKH_EINVAL, "a FASTA block has no qualities");
    if (t->transformed) return fail(SKH_EINVAL, "block was already transformed");
    for (uint64_t i = 0; i < t->offs[t->nrec]; i++) t->qual[i] = lut[t->qual[i]];
    t->transformed = t->nrec != 0;
    return SKH_OK;
}
int skh_text_validate(skh_ctx* cx, skh_text* t, const uint8_t valid[256], uint64_t limit, skh_validate_result* out)
{
    if (!t || !valid || !out) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    validate_of(t->seq.data(), t->offs.data(), t->nrec, valid, limit, out);
    return SKH_OK;
}
int skh_text_validate_quality(skh_ctx* cx, skh_text* t, const uint8_t valid[256], skh_validate_result* out)
{
    if (!t || !valid || !out) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (!t->fastq) return fail(SKH_EINVAL, "a FASTA block has no qualities");
    validate_of(t->qual.data(), t->offs.data(), t->nrec, valid, 0, out);
    return SKH_OK;
}
int skh_text_truncate(skh_ctx* cx, skh_text* t, uint64_t nrec)
{
    if (!t) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (nrec > t->nrec) return fail(SKH_EINVAL, "cannot truncate to more records than the block holds");
    t->nrec = nrec;
    return SKH_OK;
}
int skh_text_names(skh_ctx* cx, skh_text* t, uint64_t first, uint64_t count, uint64_t* pos, uint32_t* len)
{
    if (!t || ((!pos || !len) && count)) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (first + count > t->nrec) return fail(SKH_EINVAL, "record range beyond the block");
    for (uint64_t i = 0; i < count; i++) {
        pos[i] = t->npos[first + i] + (t->names_external ? 0 : t->text_off);
        len[i] = t->nlen[first + i];
    }
    return SKH_OK;
}
int skh_text_set_names(skh_ctx* cx, skh_text* t, uint64_t first, uint64_t count, const uint64_t* pos, const uint32_t* len)
{
    if (!t || ((!pos || !len) && count)) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (first + count > t->nrec) return fail(SKH_EINVAL, "record range beyond the block");
    if (t->names_external) return fail(SKH_EINVAL, "the block's names no longer lie in its text buffer");
    for (uint64_t i = 0; i < count; i++) {
        if (pos[i] < t->text_off || pos[i] + len[i] > t->text_off + t->text.size())
            return fail(SKH_EINVAL, "a name span lies outside the parsed text");
        t->npos[first + i] = pos[i] - t->text_off;
        t->nlen[first + i] = len[i];
    }
    return SKH_OK;
}
int skh_text_name(skh_ctx* c, skh_text* t, uint64_t record, uint64_t* pos, uint32_t* len)
{
    if (!pos || !len) return fail(SKH_EINVAL, "NULL argument");
    if (!t || record >= t->nrec) return fail(SKH_EINVAL, "record beyond the block");
    return skh_text_names(c, t, record, 1, pos, len);
}
int skh_text_count(skh_ctx* cx, skh_text* t, const uint8_t class_of[256], uint32_t nclass, uint32_t pass)
{
    if (!t || !class_of) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (pass >= SKH_TAB_MAX_PASSES) return fail(SKH_EINVAL, "pass out of range");
    if (nclass < 1 || nclass > SKH_MAX_CLASSES) return fail(SKH_EINVAL, "nclass out of range");
    t->nclass[pass] = nclass;
    t->counts[pass].assign((t->nrec ? t->nrec : 1) * nclass, 0);
    if (t->nrec) ora_batch_count(t->seq.data(), t->offs.data(), t->nrec, class_of, nclass, t->counts[pass].data());
    return SKH_OK;
}
int skh_text_letters(skh_ctx* cx, skh_text* t)
{
    if (!t) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    t->letters.assign((t->nrec ? t->nrec : 1) * 8, 0);
    for (uint64_t r = 0; r < t->nrec; r++)
        for (uint64_t i = t->offs[r]; i < t->offs[r + 1]; i++) t->letters[r * 8 + (t->seq[i] >> 5)] |= 1u << (t->seq[i] & 31);
    t->letters_ready = true;
    return SKH_OK;
}
int skh_text_stats(skh_ctx* cx, skh_text* t, const uint8_t is_gap[256], skh_stats_partial* inout)
{
    if (!t || !is_gap || !inout) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (t->nrec == 0) return SKH_OK;
    skh_stats_partial p;
    stats_of(t->seq.data(), t->offs.data(), t->nrec, is_gap, &p);
    skh_stats_merge(inout, &p);
    return SKH_OK;
}
int skh_text_remove_gaps(skh_ctx* cx, skh_text* t, const uint8_t is_gap[256])
{
    if (!t || !is_gap) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (t->transformed) return fail(SKH_EINVAL, "block was already transformed");
    if (t->gaps_removed) return fail(SKH_EINVAL, "gaps were already removed from this block");
    if (t->nrec == 0) return SKH_OK;
    t->filter_offs.assign(t->offs.begin(), t->offs.begin() + t->nrec + 1);
    uint64_t w = 0;
    std::vector<uint64_t> no(t->nrec + 1);
    for (uint64_t r = 0; r < t->nrec; r++) {
        no[r] = w;
        for (uint64_t i = t->offs[r]; i < t->offs[r + 1]; i++) {
            if (is_gap[t->seq[i]]) continue;
            t->seq[w] = t->seq[i];
            if (t->fastq) t->qual[w] = t->qual[i];
            w++;
        }
    }
    no[t->nrec] = w;
    std::copy(no.begin(), no.end(), t->offs.begin());
    t->gaps_removed = true;
    return SKH_OK;
}

int skh_text_subseq(skh_ctx* cx, skh_text* t, const skh_slice* slices, uint64_t n, int64_t start, int64_t end, int minus,
                    const uint8_t* lut, const uint8_t* names, uint64_t names_bytes, const uint64_t* name_pos,
                    const uint32_t* name_len)
{
    if (!t) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (t->transformed) return fail(SKH_EINVAL, "block was already transformed");
    if (t->gaps_removed) return fail(SKH_EINVAL, "gaps were removed from this block: cut subsequences before skh_text_remove_gaps");
    if (!slices) n = t->nrec;
    if (!names && slices) return fail(SKH_EINVAL, "a slice list needs names for its records");
    if (names && (!name_pos || !name_len)) return fail(SKH_EINVAL, "NULL name table");
    std::vector<uint8_t> os, oq;
    std::vector<uint64_t> oo;
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t r = slices ? slices[i].record : i;
        if (r >= t->nrec) return fail(SKH_EINVAL, "a slice names a record beyond the block");
        const int64_t st = slices ? slices[i].start : start, en = slices ? slices[i].end : end;
        const bool mi = slices ? slices[i].minus != 0 : minus != 0;
        const int64_t len = (int64_t)(t->offs[r + 1] - t->offs[r]);
        int64_t s = 0, e = 0;
        oo.push_back(os.size());
        if (!ora_sub_location(len, st, en, &s, &e) || e < s) continue;         // an empty subsequence
        const uint64_t a = t->offs[r] + (uint64_t)(s - 1), m = (uint64_t)(e - s + 1);
        const size_t at = os.size();
        os.insert(os.end(), t->seq.begin() + (long)a, t->seq.begin() + (long)(a + m));
        if (t->fastq) oq.insert(oq.end(), t->qual.begin() + (long)a, t->qual.begin() + (long)(a + m));
        if (mi) {
            std::reverse(os.begin() + (long)at, os.end());
            if (lut) for (size_t k = at; k < os.size(); k++) os[k] = lut[os[k]];
            if (t->fastq) std::reverse(oq.begin() + (long)at, oq.end());
        }
    }
    oo.push_back(os.size());
    t->seq.swap(os); t->qual.swap(oq); t->offs.swap(oo);
    t->nrec = n;
    t->transformed = true;
    if (names) {
        t->text.assign(names, names + names_bytes);
        t->npos.assign(name_pos, name_pos + n);
        t->nlen.assign(name_len, name_len + n);
        t->names_external = true;
    }
    return SKH_OK;
}

int skh_text_format_seq_submit(skh_ctx* cx, skh_text* t, const skh_seq_format* f)
{
    if (!t || !f) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    t->out_pending = 0;
    t->out.clear();
    if (t->nrec == 0) return SKH_OK;
    const uint64_t* fo = t->filter_offs.empty() ? t->offs.data() : t->filter_offs.data();
    for (uint64_t r = 0; r < t->nrec; r++) {
        const uint64_t len = t->offs[r + 1] - t->offs[r], flen = fo[r + 1] - fo[r];
        if (f->min_len >= 0 && flen < (uint64_t)f->min_len) continue;
        if (f->max_len >= 0 && flen > (uint64_t)f->max_len) continue;
        const bool as_fasta = f->empty_as_fasta && len == 0;
        if (f->print_name) {
            const uint8_t* name = t->text.data() + t->npos[r];
            const uint32_t nl = f->only_id ? id_length(name, t->nlen[r]) : t->nlen[r];
            if (f->print_seq) t->out.push_back(as_fasta ? (uint8_t)'>' : (uint8_t)f->mark);
            t->out.insert(t->out.end(), name, name + nl);
            t->out.push_back('\n');
        }
        if (f->print_seq) { put_wrapped(t->out, t->seq.data() + t->offs[r], len, (uint32_t)f->line_width); t->out.push_back('\n'); }
        if (f->print_qual && !as_fasta) {
            if (f->plus_line) { t->out.push_back('+'); t->out.push_back('\n'); }
            if (t->fastq) put_wrapped(t->out, t->qual.data() + t->offs[r], len, (uint32_t)f->line_width);
            t->out.push_back('\n');
        }
    }
    t->out_pending = t->out.size();
    return SKH_OK;
}

int skh_text_format_tab_submit(skh_ctx* cx, skh_text* t, const skh_tab_format* f)
{
    if (!t || !f) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    t->out_pending = 0;
    t->out.clear();
    if (t->nrec == 0) return SKH_OK;
    if (f->print_qual && !t->fastq) return fail(SKH_EINVAL, "quality requested for a FASTA block");
    if (f->alphabet && !t->letters_ready) return fail(SKH_EINVAL, "alphabet column requested before skh_text_letters");
    char num[64];
    auto count_of = [&](uint64_t r, uint8_t term) -> uint64_t {
        const uint32_t pass = term >> 3, cls = term & 7;
        return t->counts[pass][r * t->nclass[pass] + cls];
    };
    for (uint64_t r = 0; r < t->nrec; r++) {
        const uint64_t len = t->offs[r + 1] - t->offs[r];
        const uint8_t* name = t->text.data() + t->npos[r];
        t->out.insert(t->out.end(), name, name + (f->only_id ? id_length(name, t->nlen[r]) : t->nlen[r]));
        t->out.push_back('\t');
        if (f->print_seq) t->out.insert(t->out.end(), t->seq.begin() + (long)t->offs[r], t->seq.begin() + (long)t->offs[r + 1]);
        t->out.push_back('\t');
        if (f->print_qual) t->out.insert(t->out.end(), t->qual.begin() + (long)t->offs[r], t->qual.begin() + (long)t->offs[r + 1]);
        for (uint32_t c = 0; c < f->ncols; c++) {
            const skh_tab_column& col = f->cols[c];
            if (col.kind == SKH_TAB_LENGTH) snprintf(num, sizeof num, "%llu", (unsigned long long)len);
            else {
                double v;
                if (col.kind == SKH_TAB_CONTENT) {
                    uint64_t sum = 0;
                    for (uint32_t k = 0; k < col.nterm; k++) sum += count_of(r, col.term[k]);
                    v = len ? (double)sum / (double)len * 100 : 0.0;
                } else {
                    const double g = len ? (double)count_of(r, col.term[0]) / (double)len : 0.0;
                    const double cc = len ? (double)count_of(r, col.term[1]) / (double)len : 0.0;
                    v = col.kind == SKH_TAB_GC ? (g + cc) * 100 : (g - cc) / (g + cc) * 100;
                }
                ora_fmt_fixed(v, 2, num);
            }
            t->out.push_back('\t');
            t->out.insert(t->out.end(), num, num + strlen(num));
        }
        if (f->alphabet) {
            t->out.push_back('\t');
            for (int b = 0; b < 256; b++) if (t->letters[r * 8 + (b >> 5)] >> (b & 31) & 1u) t->out.push_back((uint8_t)b);
        }
        t->out.push_back('\n');
    }
    t->out_pending = t->out.size();
    return SKH_OK;
}

int skh_text_output_wait(skh_ctx* cx, skh_text* t, const uint8_t** out, uint64_t* out_bytes)
{
    if (!t || !out || !out_bytes) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != cx) return fail(SKH_EINVAL, "the text block belongs to another context");
    *out = t->out_pending ? t->out.data() : nullptr;
    *out_bytes = t->out_pending;
    t->out_pending = 0;
    return SKH_OK;
}
int skh_text_format_seq(skh_ctx* c, skh_text* t, const skh_seq_format* f, const uint8_t** out, uint64_t* n)
{
    const int rc = skh_text_format_seq_submit(c, t, f);
    return rc ? rc : skh_text_output_wait(c, t, out, n);
}
int skh_text_format_tab(skh_ctx* c, skh_text* t, const skh_tab_format* f, const uint8_t** out, uint64_t* n)
{
    const int rc = skh_text_format_tab_submit(c, t, f);
    return rc ? rc : skh_text_output_wait(c, t, out, n);
}

// ---------------------------------------------------------------- several "GPUs"

}  // extern "C"

struct skh_comm { std::vector<skh_ctx*> ctx; };

extern "C" {

int skh_comm_init_all(skh_ctx* const* ctxs, int n, skh_comm** out)
{
    if (!ctxs || !out || n < 1) return fail(SKH_EINVAL, "skh_comm_init_all: bad argument");
    *out = new skh_comm;
    (*out)->ctx.assign(ctxs, ctxs + n);
    return SKH_OK;
}
int skh_comm_unique_id(uint8_t id[SKH_COMM_ID_BYTES]) { memset(id, 7, SKH_COMM_ID_BYTES); return SKH_OK; }
int skh_comm_init_rank(skh_ctx* ctx, const uint8_t*, int nranks, int rank, skh_comm** out)
{
    if (!ctx || !out || nranks != 1 || rank != 0) return fail(SKH_EINVAL, "the CPU stub has no other processes to talk to");
    return skh_comm_init_all(&ctx, 1, out);
}
void skh_comm_destroy(skh_comm* c) { delete c; }
int skh_comm_local_ranks(const skh_comm* c) { return c ? (int)c->ctx.size() : 0; }
uint64_t skh_parse_wrapped_workspace_size(uint64_t text_bytes, uint64_t max_lines, uint64_t max_records)
{
    return 256 + text_bytes + 64 * (max_lines + max_records);
}

int skh_kernel_for(skh_ctx* ctx, int op, uint64_t, uint64_t, int, char* name, uint64_t name_bytes)
{
    if (!ctx || !name || op < SKH_OP_REVCOMP || op > SKH_OP_COUNT || name_bytes < 9) return SKH_EINVAL;
    memcpy(name, "cpu stub", 9);
    return SKH_OK;
}

int skh_comm_rccl_version(int* runtime_version, int* header_version, int* shared_with_process)
{
    if (runtime_version) *runtime_version = 0;
    if (header_version) *header_version = 0;
    if (shared_with_process) *shared_with_process = 0;
    return SKH_OK;
}
int skh_stats_allreduce(skh_comm* c, skh_stats_partial* partials)
{
    if (!c || !partials) return fail(SKH_EINVAL, "NULL argument");
    skh_stats_partial tot;
    skh_stats_init(&tot);
    for (size_t i = 0; i < c->ctx.size(); i++) skh_stats_merge(&tot, &partials[i]);
    for (size_t i = 0; i < c->ctx.size(); i++) partials[i] = tot;
    return SKH_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- one batch over several "GPUs"
//
// The header's rules restated: the plane is cut by skh_split_plane, every context's share is worked on
// by a thread of its own (so the driver's use of these calls runs under ThreadSanitizer with real
// concurrency), pieces of cut records land mirrored, rows of cut records are summed.

struct skh_multi { std::vector<skh_ctx*> ctx; std::vector<skh_share_info> info; };

namespace {

template <typename F>
void for_each_piece(const uint64_t* offs, uint64_t nrec, uint64_t b0, uint64_t b1, F&& f)     // f(record, a, b): slice [a, b) of the record
{
    if (b1 <= b0) return;
    uint64_t lo = 0, hi = nrec;                  // last record that starts at or before b0
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (offs[mid + 1] > b0) hi = mid; else lo = mid + 1; }
    for (uint64_t r = lo; r < nrec && offs[r] < b1; r++) {
        const uint64_t s = offs[r], e = offs[r + 1];
        const uint64_t a = std::max(s, b0), b = std::min(e, b1);
        if (b > a) f(r, a - s, b - s);
    }
}

std::vector<uint64_t> cut_plane(skh_multi* m, const skh_batch* b, uint64_t nrec)
{
    std::vector<uint64_t> cuts(m->ctx.size() + 1);
    skh_split_plane(b->offs.data(), nrec, (int)m->ctx.size(), cuts.data());
    m->info.assign(m->ctx.size(), skh_share_info());
    for (size_t k = 0; k < m->ctx.size(); k++) {
        skh_share_info& si = m->info[k];
        si.first_byte = cuts[k]; si.last_byte = cuts[k + 1]; si.chunks = 1;
        bool first = true;
        for_each_piece(b->offs.data(), nrec, cuts[k], cuts[k + 1], [&](uint64_t r, uint64_t a, uint64_t e) {
            if (first) { si.first_record = r; si.head_cut = a; first = false; }
            si.nrecords = r + 1 - si.first_record;
            si.tail_cut = b->offs[r + 1] - b->offs[r] - e;
        });
    }
    return cuts;
}

template <typename F>
void on_every_share(skh_multi* m, F&& f)         // f(k) on a thread per context
{
    std::vector<std::thread> thr;
    for (size_t k = 1; k < m->ctx.size(); k++) thr.emplace_back([&f, k] { f(k); });
    f(0);
    for (auto& t : thr) t.join();
}

}  // namespace

extern "C" {

int skh_split_plane(const uint64_t* offsets, uint64_t nrec, int nparts, uint64_t* cuts)
{
    if (!offsets || !cuts || nparts < 1) return fail(SKH_EINVAL, "skh_split_plane: bad argument");
    const uint64_t total = offsets[nrec], tol = total / (uint64_t)nparts / 16;
    cuts[0] = 0;
    for (int k = 1; k < nparts; k++) {
        const uint64_t target = (uint64_t)((unsigned __int128)total * (unsigned)k / (unsigned)nparts);
        const uint64_t r = (uint64_t)(std::lower_bound(offsets, offsets + nrec, target) - offsets);
        uint64_t cut = offsets[r];
        if (cut != target) {
            const uint64_t s = offsets[r - 1], e = offsets[r];
            if (e - target <= tol && e - target <= target - s) cut = e;
            else if (target - s <= tol) cut = s;
            else cut = s + ((target - s) & ~15ull);
        }
        cuts[k] = std::max(cut, cuts[k - 1]);
    }
    cuts[nparts] = total;
    return SKH_OK;
}

int skh_multi_init(skh_ctx* const* ctxs, int n, skh_multi** out)
{
    if (!ctxs || !out || n < 1) return fail(SKH_EINVAL, "skh_multi_init: bad argument");
    *out = new skh_multi;
    (*out)->ctx.assign(ctxs, ctxs + n);
    (*out)->info.assign((size_t)n, skh_share_info());
    return SKH_OK;
}
void skh_multi_destroy(skh_multi* m) { delete m; }
int skh_multi_size(const skh_multi* m) { return m ? (int)m->ctx.size() : 0; }
int skh_multi_comm_init(skh_multi* m) { return m ? SKH_OK : fail(SKH_EINVAL, "multi is NULL"); }
int skh_multi_shares(const skh_multi* m, skh_share_info* info)
{
    if (!m || !info) return fail(SKH_EINVAL, "NULL argument");
    std::copy(m->info.begin(), m->info.end(), info);
    return SKH_OK;
}

int skh_revcomp_multi(skh_multi* m, skh_batch* b, uint64_t nrec, const uint8_t* lut, int do_reverse, int reverse_qual)
{
    if (!m) return fail(SKH_EINVAL, "multi is NULL");
    if (int rc = batch_check(b, nrec)) return rc;
    if (b->offs[nrec] == 0 || (!lut && !do_reverse)) return SKH_OK;
    const std::vector<uint64_t> cuts = cut_plane(m, b, nrec);
    const bool qual = b->with_qual && reverse_qual && do_reverse;
    std::vector<uint8_t> out_seq(b->seq.size()), out_qual(qual ? b->qual.size() : 0);
    on_every_share(m, [&](size_t k) {
        for_each_piece(b->offs.data(), nrec, cuts[k], cuts[k + 1], [&](uint64_t r, uint64_t a, uint64_t e) {
            const uint64_t s = b->offs[r], len = b->offs[r + 1] - s;
            for (uint64_t i = a; i < e; i++) {
                const uint64_t dst = do_reverse ? s + len - 1 - i : s + i;
                const uint8_t c = b->seq[s + i];
                out_seq[dst] = lut ? lut[c] : c;
                if (qual) out_qual[dst] = b->qual[s + i];
            }
        });
    });
    memcpy(b->seq.data(), out_seq.data(), b->offs[nrec]);
    if (qual) memcpy(b->qual.data(), out_qual.data(), b->offs[nrec]);
    return SKH_OK;
}

int skh_count_multi(skh_multi* m, skh_batch* b, uint64_t nrec, const uint8_t class_of[256], uint32_t nclass, uint64_t* counts)
{
    if (!m || !class_of) return fail(SKH_EINVAL, "NULL argument");
    if (int rc = batch_check(b, nrec)) return rc;
    if (nclass < 1 || nclass > SKH_MAX_CLASSES) return fail(SKH_EINVAL, "nclass must be 1..SKH_MAX_CLASSES");
    if (nrec == 0) return SKH_OK;
    if (!counts) return fail(SKH_EINVAL, "counts pointer is NULL");
    const std::vector<uint64_t> cuts = cut_plane(m, b, nrec);
    struct Row { uint64_t record; uint64_t v[SKH_MAX_CLASSES]; };
    std::vector<std::vector<Row>> rows(m->ctx.size());
    on_every_share(m, [&](size_t k) {
        for_each_piece(b->offs.data(), nrec, cuts[k], cuts[k + 1], [&](uint64_t r, uint64_t a, uint64_t e) {
            Row row;
            row.record = r;
            memset(row.v, 0, sizeof row.v);
            for (uint64_t i = a; i < e; i++) {
                const uint8_t c = class_of[b->seq[b->offs[r] + i]];
                if (c != SKH_NO_CLASS) row.v[c]++;
            }
            rows[k].push_back(row);
        });
    });
    memset(counts, 0, nrec * nclass * sizeof(uint64_t));
    for (const auto& share : rows)
        for (const Row& row : share)
            for (uint32_t c = 0; c < nclass; c++) counts[row.record * nclass + c] += row.v[c];
    return SKH_OK;
}

int skh_stats_multi(skh_multi* m, skh_batch* b, uint64_t nrec, const uint8_t is_gap[256], skh_stats_partial* inout)
{
    if (!m || !is_gap || !inout) return fail(SKH_EINVAL, "NULL argument");
    if (int rc = batch_check(b, nrec)) return rc;
    if (nrec == 0) return SKH_OK;
    const std::vector<uint64_t> cuts = cut_plane(m, b, nrec);
    std::vector<skh_stats_partial> part(m->ctx.size());
    on_every_share(m, [&](size_t k) {
        skh_stats_partial& p = part[k];
        skh_stats_init(&p);
        // a record counts where it starts; gap letters and bytes are those of the share
        const uint64_t b0 = cuts[k], b1 = cuts[k + 1];
        const bool last = k + 1 == m->ctx.size();
        for (uint64_t r = (uint64_t)(std::lower_bound(b->offs.data(), b->offs.data() + nrec, b0) - b->offs.data());
             r < nrec && (b->offs[r] < b1 || last); r++) {
            const int64_t len = (int64_t)(b->offs[r + 1] - b->offs[r]);
            p.num++;
            p.len_min = std::min(p.len_min, len);
            p.len_max = std::max(p.len_max, len);
        }
        p.len_sum = (int64_t)(b1 - b0);
        for (uint64_t i = b0; i < b1; i++) p.gap_sum += is_gap[b->seq[i]] ? 1 : 0;
    });
    for (const auto& p : part) skh_stats_merge(inout, &p);
    return SKH_OK;
}

}  // extern "C"
