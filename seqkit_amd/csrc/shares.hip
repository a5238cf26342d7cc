// shares.hip -- one pinned batch, several GPUs at once.
//
// The reference's only parallel knob on this path is a split INSIDE a long record
// (seq.ComplementThreads = config.Threads, seqkit/cmd/seq.go:88, README.md:285-295) and its per-file
// accumulators (stat.go:106-125).  Here one CSR batch in pinned host memory is cut into one share of
// about equal byte volume per context (skh_split_plane: at the record boundary nearest to the byte
// target, or inside the record when no boundary is near: input slice [a, b) of a record of length L
// is a record of its own whose reverse lands at [L-b, L-a)), every context has a worker thread that
// uploads its share chunk by chunk on two streams, runs the single-GPU kernels on it and downloads
// the results into the share's own place in the batch, so the links and kernels of all GPUs work at
// the same time.  No byte ever moves between GPUs; the one collective is the all-reduce of the
// `stats` totals (multi.hip).
#include "seqkit_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "ctx.h"

namespace {

int hip_fail(hipError_t e, const char* what)
{
    return skh::set_error(e == hipErrorOutOfMemory ? SKH_ENOMEM : SKH_EHIP, "%s: %s", what, hipGetErrorString(e));
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(e_, #expr); } while (0)

inline uint64_t align16(uint64_t v) { return (v + 15) & ~15ull; }

// first r in [0, nrec] with offs[r] >= pos (nrec when there is none before the end)
uint64_t first_start_at_or_after(const uint64_t* offs, uint64_t nrec, uint64_t pos)
{
    uint64_t lo = 0, hi = nrec;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (offs[mid] >= pos) hi = mid; else lo = mid + 1; }
    return lo;
}

enum Op { OP_NONE = 0, OP_REVCOMP, OP_COUNT, OP_STATS, OP_QUIT };

// One chunk of a worker's share: plane bytes [b0, b1).  Its local records are the piece of the record
// that reaches into the chunk from before it (cont = 1) followed by the records that start in it; the
// last of them may reach beyond b1.
struct Chunk {
    uint64_t b0 = 0, b1 = 0;
    uint64_t rs0 = 0, rs1 = 0;      // records that START in the chunk
    uint32_t cont = 0;              // 1: record rs0 - 1 continues into the chunk
    uint64_t dpos = 0;              // the chunk's place in the worker's device planes (multiple of 16)
    uint64_t opos = 0;              // its local offsets in the worker's offsets arrays (nloc + 1 entries)
    uint64_t cpos = 0;              // its first row in the worker's count arrays
    uint64_t r_first() const { return rs0 - cont; }
    uint64_t nloc() const { return rs1 - rs0 + cont; }
};

struct Piece { uint64_t dsrc, hdst, len; };                   // phase 2 of a reverse: device position -> batch position
struct PartialRow { uint64_t record; uint64_t v[SKH_MAX_CLASSES]; };

struct Job {
    Op op = OP_NONE;
    skh_batch* b = nullptr;
    uint64_t nrec = 0;
    const uint64_t* cuts = nullptr;         // nchunks_total + 1 byte positions
    const uint64_t* rec_cuts = nullptr;     // first record that starts at or after each of them
    const uint8_t* table = nullptr;         // lut / class_of / is_gap (nullptr: none)
    int do_reverse = 0, with_qual = 0;
    uint32_t nclass = 0;
    uint64_t* counts = nullptr;
};

}  // namespace

struct skh_multi {
    struct Worker {
        skh_multi* m = nullptr;
        skh_ctx* ctx = nullptr;
        int index = 0;
        std::thread thread;
        hipStream_t st[2] = {nullptr, nullptr};
        // device planes of the share and their capacities
        uint8_t *d_in = nullptr, *d_out = nullptr, *d_qin = nullptr, *d_qout = nullptr;
        uint64_t cap_in = 0, cap_out = 0, cap_qin = 0, cap_qout = 0;
        uint64_t *d_loffs = nullptr, *h_loffs = nullptr, cap_offs = 0;
        uint64_t *d_counts = nullptr, *h_counts = nullptr, cap_counts = 0;
        skh::StatsPartial *d_stats = nullptr, *h_stats = nullptr;
        uint64_t cap_stats = 0;
        // the call in hand
        std::vector<Chunk> chunks;
        std::vector<Piece> pieces;
        std::vector<PartialRow> rows;
        skh_stats_partial partial;
        skh_share_info info;
        int rc = SKH_OK;
        std::string err;
    };
    std::vector<Worker> w;
    uint64_t chunk_bytes = 32ull << 20;
    skh_comm* comm = nullptr;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t generation = 0;            // bumped for every job
    int pending = 0;                    // workers still working on it
    int uploaded = 0;                   // workers whose uploads have all landed (reverse: before any piece is written back)
    Job job;
    std::vector<uint64_t> cuts, rec_cuts;
    std::vector<skh_stats_partial> partials;
};

namespace {

using Worker = skh_multi::Worker;

template <typename T>
int grow_device(T** p, uint64_t* cap, uint64_t need)
{
    if (need <= *cap) return SKH_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    const uint64_t want = need + need / 8 + 256;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), want * sizeof(T)));
    *cap = want;
    return SKH_OK;
}

template <typename T>
int grow_pair(T** d, T** h, uint64_t* cap, uint64_t need)
{
    if (need <= *cap) return SKH_OK;
    if (*d) { (void)hipFree(*d); *d = nullptr; }
    if (*h) { (void)hipHostFree(*h); *h = nullptr; }
    *cap = 0;
    const uint64_t want = need + need / 8 + 256;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(d), want * sizeof(T)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(h), want * sizeof(T), hipHostMallocDefault));
    *cap = want;
    return SKH_OK;
}

// the chunks of worker k: cuts[k * S .. (k + 1) * S] of the job
void plan_chunks(Worker& w, const Job& j, uint32_t per_worker)
{
    const uint64_t* offs = j.b->h_offs;
    const uint64_t total = offs[j.nrec];
    w.chunks.clear();
    uint64_t dpos = 0, opos = 0, cpos = 0;
    for (uint32_t c = 0; c < per_worker; c++) {
        const size_t g = (size_t)w.index * per_worker + c;
        Chunk ch;
        ch.b0 = j.cuts[g]; ch.b1 = j.cuts[g + 1];
        ch.rs0 = j.rec_cuts[g]; ch.rs1 = j.rec_cuts[g + 1];
        // a record that began before the chunk reaches into it
        ch.cont = (ch.b1 > ch.b0 && ch.rs0 > 0 && (ch.rs0 == j.nrec ? ch.b0 < total : offs[ch.rs0] > ch.b0)) ? 1u : 0u;
        if (ch.b1 == ch.b0 && ch.rs1 == ch.rs0) continue;      // neither bytes nor (empty) records
        ch.dpos = dpos; ch.opos = opos; ch.cpos = cpos;
        dpos += align16(ch.b1 - ch.b0) + 16;
        opos += ch.nloc() + 1;
        cpos += ch.nloc();
        w.chunks.push_back(ch);
    }
}

// local offsets of a chunk: its records clipped to [b0, b1), relative to b0
void fill_local_offsets(const Chunk& ch, const uint64_t* offs, uint64_t* dst)
{
    dst[0] = 0;
    const uint64_t r0 = ch.r_first(), n = ch.nloc();
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t e = offs[r0 + i + 1];
        dst[i + 1] = (e < ch.b1 ? e : ch.b1) - ch.b0;
    }
}

void describe_share(Worker& w, const Job& j)
{
    skh_share_info& s = w.info;
    memset(&s, 0, sizeof s);
    if (w.chunks.empty()) return;
    const uint64_t* offs = j.b->h_offs;
    const Chunk &a = w.chunks.front(), &z = w.chunks.back();
    s.first_byte = a.b0; s.last_byte = z.b1;
    s.first_record = a.r_first();
    s.nrecords = z.rs1 - s.first_record;
    s.chunks = (uint32_t)w.chunks.size();
    if (a.cont) s.head_cut = a.b0 - offs[a.r_first()];
    if (z.nloc() && z.b1 > z.b0) {
        const uint64_t r = z.r_first() + z.nloc() - 1;            // the last record the share touches
        if (offs[r + 1] > z.b1) s.tail_cut = offs[r + 1] - z.b1;
    }
}

int sync_both(Worker& w)
{
    HIP_TRY(hipStreamSynchronize(w.st[0]));
    HIP_TRY(hipStreamSynchronize(w.st[1]));
    return SKH_OK;
}

// every worker reports here once per reverse job, whether its uploads worked or not, and then waits
// for the others: after that no byte of the batch is still to be read, and the pieces of records that
// were cut may be written to their mirrored places
void uploads_barrier(skh_multi* m)
{
    std::unique_lock<std::mutex> l(m->mu);
    m->uploaded++;
    m->cv.notify_all();
    m->cv.wait(l, [&] { return m->uploaded >= (int)m->w.size(); });
}

int run_revcomp(Worker& w, const Job& j)
{
    skh_batch* b = j.b;
    const uint64_t* offs = b->h_offs;
    uint64_t bytes = 0, nofs = 0;
    for (const Chunk& ch : w.chunks) { bytes = ch.dpos + align16(ch.b1 - ch.b0) + 16; nofs = ch.opos + ch.nloc() + 1; }
    const bool qual = j.with_qual != 0;
    int rc = SKH_OK;
    if ((rc = grow_device(&w.d_in, &w.cap_in, bytes)) || (rc = grow_device(&w.d_out, &w.cap_out, bytes)) ||
        (rc = grow_pair(&w.d_loffs, &w.h_loffs, &w.cap_offs, nofs)))
        return rc;
    if (qual && ((rc = grow_device(&w.d_qin, &w.cap_qin, bytes)) || (rc = grow_device(&w.d_qout, &w.cap_qout, bytes)))) return rc;
    w.pieces.clear();
    for (size_t c = 0; c < w.chunks.size(); c++) {
        const Chunk& ch = w.chunks[c];
        const uint64_t len = ch.b1 - ch.b0, n = ch.nloc();
        if (len == 0) continue;
        hipStream_t st = w.st[c & 1];
        uint64_t* lo = w.h_loffs + ch.opos;
        fill_local_offsets(ch, offs, lo);
        HIP_TRY(hipMemcpyAsync(w.d_loffs + ch.opos, lo, (n + 1) * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(w.d_in + ch.dpos, b->h_seq + ch.b0, len, hipMemcpyHostToDevice, st));
        if (qual) HIP_TRY(hipMemcpyAsync(w.d_qin + ch.dpos, b->h_qual + ch.b0, len, hipMemcpyHostToDevice, st));
        if ((rc = skh_revcomp_dev(w.ctx, w.d_in + ch.dpos, w.d_out + ch.dpos, w.d_loffs + ch.opos, n, len, j.table, j.do_reverse, st)))
            return rc;
        if (qual && (rc = skh_revcomp_dev(w.ctx, w.d_qin + ch.dpos, w.d_qout + ch.dpos, w.d_loffs + ch.opos, n, len, nullptr, 1, st)))
            return rc;
        // whole records go back to where they came from; a piece of a cut record belongs at the mirrored
        // place of its record, which another chunk may still have to read: those wait for the barrier
        uint64_t i0 = 0, i1 = n;
        if (j.do_reverse) {
            const uint64_t r0 = ch.r_first();
            auto cut = [&](uint64_t i) { return offs[r0 + i] < ch.b0 || offs[r0 + i + 1] > ch.b1; };
            auto defer = [&](uint64_t i) {
                const uint64_t r = r0 + i, pb = ch.b0 + lo[i + 1], plen = lo[i + 1] - lo[i];
                if (plen) w.pieces.push_back(Piece{ch.dpos + lo[i], offs[r] + offs[r + 1] - pb, plen});
            };
            if (n && cut(0)) { defer(0); i0 = 1; }
            if (n > i0 && cut(n - 1)) { defer(n - 1); i1 = n - 1; }
        }
        if (i1 > i0 && lo[i1] > lo[i0]) {
            HIP_TRY(hipMemcpyAsync(b->h_seq + ch.b0 + lo[i0], w.d_out + ch.dpos + lo[i0], lo[i1] - lo[i0], hipMemcpyDeviceToHost, st));
            if (qual) HIP_TRY(hipMemcpyAsync(b->h_qual + ch.b0 + lo[i0], w.d_qout + ch.dpos + lo[i0], lo[i1] - lo[i0], hipMemcpyDeviceToHost, st));
        }
    }
    return sync_both(w);
}

int run_revcomp_pieces(Worker& w, const Job& j)
{
    skh_batch* b = j.b;
    for (size_t k = 0; k < w.pieces.size(); k++) {
        const Piece& p = w.pieces[k];
        hipStream_t st = w.st[k & 1];
        HIP_TRY(hipMemcpyAsync(b->h_seq + p.hdst, w.d_out + p.dsrc, p.len, hipMemcpyDeviceToHost, st));
        if (j.with_qual) HIP_TRY(hipMemcpyAsync(b->h_qual + p.hdst, w.d_qout + p.dsrc, p.len, hipMemcpyDeviceToHost, st));
    }
    return w.pieces.empty() ? SKH_OK : sync_both(w);
}

int run_count(Worker& w, const Job& j)
{
    skh_batch* b = j.b;
    const uint64_t* offs = b->h_offs;
    uint64_t bytes = 0, nofs = 0, nrows = 0;
    for (const Chunk& ch : w.chunks) { bytes = ch.dpos + align16(ch.b1 - ch.b0) + 16; nofs = ch.opos + ch.nloc() + 1; nrows = ch.cpos + ch.nloc(); }
    int rc = SKH_OK;
    if ((rc = grow_device(&w.d_in, &w.cap_in, bytes)) || (rc = grow_pair(&w.d_loffs, &w.h_loffs, &w.cap_offs, nofs)) ||
        (rc = grow_pair(&w.d_counts, &w.h_counts, &w.cap_counts, nrows * j.nclass)))
        return rc;
    for (size_t c = 0; c < w.chunks.size(); c++) {
        const Chunk& ch = w.chunks[c];
        const uint64_t len = ch.b1 - ch.b0, n = ch.nloc();
        if (n == 0) continue;
        hipStream_t st = w.st[c & 1];
        uint64_t* lo = w.h_loffs + ch.opos;
        fill_local_offsets(ch, offs, lo);
        HIP_TRY(hipMemcpyAsync(w.d_loffs + ch.opos, lo, (n + 1) * 8, hipMemcpyHostToDevice, st));
        if (len) HIP_TRY(hipMemcpyAsync(w.d_in + ch.dpos, b->h_seq + ch.b0, len, hipMemcpyHostToDevice, st));
        if ((rc = skh_count_dev(w.ctx, w.d_in + ch.dpos, w.d_loffs + ch.opos, n, len, j.table, j.nclass, w.d_counts + ch.cpos * j.nclass, st)))
            return rc;
        HIP_TRY(hipMemcpyAsync(w.h_counts + ch.cpos * j.nclass, w.d_counts + ch.cpos * j.nclass, n * j.nclass * 8, hipMemcpyDeviceToHost, st));
    }
    if ((rc = sync_both(w))) return rc;
    // rows of whole records straight to the caller's array; the rows of the pieces of a cut record are
    // added up by the calling thread once every share is in
    w.rows.clear();
    for (const Chunk& ch : w.chunks) {
        const uint64_t n = ch.nloc(), r0 = ch.r_first();
        uint64_t i0 = 0, i1 = n;
        auto cut = [&](uint64_t i) { return offs[r0 + i] < ch.b0 || offs[r0 + i + 1] > ch.b1; };
        auto keep = [&](uint64_t i) {
            PartialRow pr;
            pr.record = r0 + i;
            memset(pr.v, 0, sizeof pr.v);
            memcpy(pr.v, w.h_counts + (ch.cpos + i) * j.nclass, j.nclass * 8);
            w.rows.push_back(pr);
        };
        if (n && cut(0)) { keep(0); i0 = 1; }
        if (n > i0 && cut(n - 1)) { keep(n - 1); i1 = n - 1; }
        if (i1 > i0)
            memcpy(j.counts + (r0 + i0) * j.nclass, w.h_counts + (ch.cpos + i0) * j.nclass, (i1 - i0) * j.nclass * 8);
    }
    return SKH_OK;
}

int run_stats(Worker& w, const Job& j)
{
    skh_batch* b = j.b;
    const uint64_t* offs = b->h_offs;
    skh_stats_init(&w.partial);
    uint64_t bytes = 0, nofs = 0;
    for (const Chunk& ch : w.chunks) { bytes = ch.dpos + align16(ch.b1 - ch.b0) + 16; nofs = ch.opos + ch.nloc() + 1; }
    int rc = SKH_OK;
    if ((rc = grow_device(&w.d_in, &w.cap_in, bytes)) || (rc = grow_pair(&w.d_loffs, &w.h_loffs, &w.cap_offs, nofs)) ||
        (rc = grow_pair(&w.d_stats, &w.h_stats, &w.cap_stats, (uint64_t)w.chunks.size())))
        return rc;
    skh::Table256 t;
    memcpy(t.v, j.table, 256);
    for (size_t c = 0; c < w.chunks.size(); c++) {
        const Chunk& ch = w.chunks[c];
        // a record counts where it starts (num, len_min, len_max come from the offsets of the records that
        // start in the chunk); gap letters and len_sum are counted over the chunk's own bytes
        const uint64_t len = ch.b1 - ch.b0, nstart = ch.rs1 - ch.rs0;
        hipStream_t st = w.st[c & 1];
        HIP_TRY(hipMemcpyAsync(w.d_loffs + ch.opos, offs + ch.rs0, (nstart + 1) * 8, hipMemcpyHostToDevice, st));
        if (len) HIP_TRY(hipMemcpyAsync(w.d_in + ch.dpos, b->h_seq + ch.b0, len, hipMemcpyHostToDevice, st));
        uint32_t slot;
        if ((rc = skh::slot_for(w.ctx, st, &slot))) return rc;
        uint64_t* slots = w.ctx->d_stat_slots + (uint64_t)slot * skh::STATS_SLOT_WORDS;
        HIP_TRY(skh::launch_stats(skh::cfg_for(w.ctx, nstart, len), w.d_in + ch.dpos, w.d_loffs + ch.opos, nstart, len, t,
                                  w.d_stats + c, slots, st));
        HIP_TRY(hipMemcpyAsync(w.h_stats + c, w.d_stats + c, sizeof(skh::StatsPartial), hipMemcpyDeviceToHost, st));
    }
    if ((rc = sync_both(w))) return rc;
    for (size_t c = 0; c < w.chunks.size(); c++) skh_stats_merge(&w.partial, reinterpret_cast<const skh_stats_partial*>(w.h_stats + c));
    return SKH_OK;
}

void worker_main(Worker* wp)
{
    Worker& w = *wp;
    skh_multi* m = w.m;
    (void)hipSetDevice(w.ctx->device);
    uint64_t seen = 0;
    for (;;) {
        Job j;
        {
            std::unique_lock<std::mutex> l(m->mu);
            m->cv.wait(l, [&] { return m->generation != seen; });
            seen = m->generation;
            j = m->job;
        }
        if (j.op == OP_QUIT) return;
        const auto t0 = std::chrono::steady_clock::now();
        const uint32_t per_worker = (uint32_t)((m->cuts.size() - 1) / m->w.size());
        plan_chunks(w, j, per_worker);
        describe_share(w, j);
        int rc = SKH_OK;
        switch (j.op) {
        case OP_REVCOMP:
            rc = run_revcomp(w, j);
            uploads_barrier(m);
            if (rc == SKH_OK) rc = run_revcomp_pieces(w, j);
            break;
        case OP_COUNT: rc = run_count(w, j); break;
        case OP_STATS: rc = run_stats(w, j); break;
        default: break;
        }
        w.rc = rc;
        w.err = rc ? skh_last_error() : "";
        if (rc) { (void)hipStreamSynchronize(w.st[0]); (void)hipStreamSynchronize(w.st[1]); }   // nothing of a failed job stays in flight on the caller's batch
        w.info.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        {
            std::lock_guard<std::mutex> l(m->mu);
            m->pending--;
        }
        m->cv.notify_all();
    }
}

// cuts the batch, hands the job to every worker and waits for them; the first failure is the call's
int run_job(skh_multi* m, Job j)
{
    const uint64_t* offs = j.b->h_offs;
    const uint64_t total = offs[j.nrec];
    const int n = (int)m->w.size();
    // chunks per worker: about chunk_bytes each, so that a share's transfers in both directions and its
    // kernels overlap
    uint64_t per_worker = (total / (uint64_t)n + m->chunk_bytes - 1) / m->chunk_bytes;
    per_worker = per_worker < 1 ? 1 : (per_worker > 4096 ? 4096 : per_worker);
    const int nparts = n * (int)per_worker;
    m->cuts.resize((size_t)nparts + 1);
    m->rec_cuts.resize((size_t)nparts + 1);
    if (int rc = skh_split_plane(offs, j.nrec, nparts, m->cuts.data())) return rc;
    for (int k = 0; k < nparts; k++) m->rec_cuts[(size_t)k] = first_start_at_or_after(offs, j.nrec, m->cuts[(size_t)k]);
    m->rec_cuts[(size_t)nparts] = j.nrec;
    j.cuts = m->cuts.data();
    j.rec_cuts = m->rec_cuts.data();
    {
        std::lock_guard<std::mutex> l(m->mu);
        m->job = j;
        m->pending = n;
        m->uploaded = 0;
        m->generation++;
    }
    m->cv.notify_all();
    {
        std::unique_lock<std::mutex> l(m->mu);
        m->cv.wait(l, [&] { return m->pending == 0; });
    }
    for (const Worker& w : m->w)
        if (w.rc != SKH_OK) return skh::set_error(w.rc, "%s", w.err.c_str());
    return SKH_OK;
}

// checks the host-side CSR like the single-GPU entry points do
int batch_total(const skh_batch* b, uint64_t nrec, uint64_t* total)
{
    if (!b) return skh::set_error(SKH_EINVAL, "batch is NULL");
    if (b->pending) return skh::set_error(SKH_EINVAL, "batch has an unfinished submit; call skh_batch_wait first");
    if (nrec > b->max_records) return skh::set_error(SKH_ECAPACITY, "more records than the batch was allocated for");
    if (b->h_offs[0] != 0) return skh::set_error(SKH_EINVAL, "offsets[0] must be 0");
    if (b->h_offs[nrec] > b->max_bytes) return skh::set_error(SKH_ECAPACITY, "more sequence bytes than the batch was allocated for");
    *total = b->h_offs[nrec];
    return SKH_OK;
}

}  // namespace

extern "C" {

int skh_split_plane(const uint64_t* offsets, uint64_t nrec, int nparts, uint64_t* cuts)
{
    if (!offsets || !cuts) return skh::set_error(SKH_EINVAL, "NULL argument");
    if (nparts < 1) return skh::set_error(SKH_EINVAL, "nparts must be at least 1");
    const uint64_t total = offsets[nrec];
    const uint64_t tol = total / (uint64_t)nparts / 16;         // how far a cut may move to reach a record boundary
    cuts[0] = 0;
    for (int k = 1; k < nparts; k++) {
        const uint64_t target = (uint64_t)((unsigned __int128)total * (unsigned)k / (unsigned)nparts);
        const uint64_t r = first_start_at_or_after(offsets, nrec, target);      // offsets[r] >= target (r <= nrec)
        uint64_t cut = offsets[r];
        if (cut != target) {
            // the target lies inside record r - 1
            const uint64_t s = offsets[r - 1], e = offsets[r];
            if (e - target <= tol && e - target <= target - s) cut = e;          // skh_split_by_bytes' rule: the nearer boundary
            else if (target - s <= tol) cut = s;
            else cut = s + ((target - s) & ~15ull);                              // skh_split_record's rule: a multiple of 16 of the record
        }
        if (cut < cuts[k - 1]) cut = cuts[k - 1];
        cuts[k] = cut;
    }
    cuts[nparts] = total;
    return SKH_OK;
}

int skh_multi_init(skh_ctx* const* ctxs, int n, skh_multi** out)
{
    if (!ctxs || !out || n < 1 || n > 64) return skh::set_error(SKH_EINVAL, "skh_multi_init: bad argument");
    *out = nullptr;
    for (int i = 0; i < n; i++) if (!ctxs[i]) return skh::set_error(SKH_EINVAL, "skh_multi_init: NULL context");
    skh_multi* m = new (std::nothrow) skh_multi;
    if (!m) return skh::set_error(SKH_ENOMEM, "out of host memory");
    if (const char* e = getenv("SKH_MULTI_CHUNK")) { const long long v = atoll(e); if (v >= 64) m->chunk_bytes = (uint64_t)v; }
    m->w.resize((size_t)n);
    m->partials.resize((size_t)n);
    int rc = SKH_OK;
    for (int i = 0; i < n && rc == SKH_OK; i++) {
        Worker& w = m->w[(size_t)i];
        w.m = m; w.ctx = ctxs[i]; w.index = i;
        hipError_t e = hipSetDevice(w.ctx->device);
        for (int s = 0; s < 2 && e == hipSuccess; s++) e = hipStreamCreateWithFlags(&w.st[s], hipStreamNonBlocking);
        if (e != hipSuccess) rc = hip_fail(e, "skh_multi_init");
    }
    if (rc != SKH_OK) { skh_multi_destroy(m); return rc; }
    for (Worker& w : m->w) w.thread = std::thread(worker_main, &w);
    *out = m;
    return SKH_OK;
}

void skh_multi_destroy(skh_multi* m)
{
    if (!m) return;
    {
        std::lock_guard<std::mutex> l(m->mu);
        m->job = Job();
        m->job.op = OP_QUIT;
        m->generation++;
    }
    m->cv.notify_all();
    for (Worker& w : m->w) {
        if (w.thread.joinable()) w.thread.join();
        (void)hipSetDevice(w.ctx->device);
        for (hipStream_t& s : w.st)
            if (s) { (void)hipStreamSynchronize(s); skh::slot_release(w.ctx, s); (void)hipStreamDestroy(s); }
        (void)hipFree(w.d_in); (void)hipFree(w.d_out); (void)hipFree(w.d_qin); (void)hipFree(w.d_qout);
        (void)hipFree(w.d_loffs); (void)hipFree(w.d_counts); (void)hipFree(w.d_stats);
        (void)hipHostFree(w.h_loffs); (void)hipHostFree(w.h_counts); (void)hipHostFree(w.h_stats);
    }
    if (m->comm) skh_comm_destroy(m->comm);
    delete m;
}

int skh_multi_size(const skh_multi* m) { return m ? (int)m->w.size() : 0; }

int skh_multi_comm_init(skh_multi* m)
{
    if (!m) return skh::set_error(SKH_EINVAL, "multi is NULL");
    if (m->comm || m->w.size() < 2) return SKH_OK;
    std::vector<skh_ctx*> ctxs;
    for (const Worker& w : m->w) ctxs.push_back(w.ctx);
    return skh_comm_init_all(ctxs.data(), (int)ctxs.size(), &m->comm);
}

int skh_multi_shares(const skh_multi* m, skh_share_info* info)
{
    if (!m || !info) return skh::set_error(SKH_EINVAL, "NULL argument");
    for (size_t i = 0; i < m->w.size(); i++) info[i] = m->w[i].info;
    return SKH_OK;
}

int skh_revcomp_multi(skh_multi* m, skh_batch* b, uint64_t nrec, const uint8_t* lut, int do_reverse, int reverse_qual)
{
    if (!m) return skh::set_error(SKH_EINVAL, "multi is NULL");
    uint64_t total = 0;
    if (int rc = batch_total(b, nrec, &total)) return rc;
    if (total == 0 || (!lut && !do_reverse)) return SKH_OK;
    Job j;
    j.op = OP_REVCOMP; j.b = b; j.nrec = nrec; j.table = lut; j.do_reverse = do_reverse ? 1 : 0;
    j.with_qual = (b->with_qual && reverse_qual && do_reverse) ? 1 : 0;
    return run_job(m, j);
}

int skh_count_multi(skh_multi* m, skh_batch* b, uint64_t nrec, const uint8_t class_of[256], uint32_t nclass, uint64_t* counts)
{
    if (!m) return skh::set_error(SKH_EINVAL, "multi is NULL");
    if (!class_of) return skh::set_error(SKH_EINVAL, "class_of is NULL");
    if (nclass < 1 || nclass > SKH_MAX_CLASSES) return skh::set_error(SKH_EINVAL, "nclass must be 1..SKH_MAX_CLASSES");
    uint64_t total = 0;
    if (int rc = batch_total(b, nrec, &total)) return rc;
    if (nrec == 0) return SKH_OK;
    if (!counts) return skh::set_error(SKH_EINVAL, "counts pointer is NULL");
    Job j;
    j.op = OP_COUNT; j.b = b; j.nrec = nrec; j.table = class_of; j.nclass = nclass; j.counts = counts;
    if (int rc = run_job(m, j)) return rc;
    // a record that was cut: its row is the sum of the rows of its pieces (they arrive in plane order)
    uint64_t last = ~0ull;
    for (const Worker& w : m->w)
        for (const PartialRow& pr : w.rows) {
            uint64_t* row = counts + pr.record * nclass;
            if (pr.record != last) { memset(row, 0, nclass * 8); last = pr.record; }
            for (uint32_t c = 0; c < nclass; c++) row[c] += pr.v[c];
        }
    return SKH_OK;
}

int skh_stats_multi(skh_multi* m, skh_batch* b, uint64_t nrec, const uint8_t is_gap[256], skh_stats_partial* inout)
{
    if (!m || !is_gap || !inout) return skh::set_error(SKH_EINVAL, "NULL argument");
    uint64_t total = 0;
    if (int rc = batch_total(b, nrec, &total)) return rc;
    if (nrec == 0) return SKH_OK;
    Job j;
    j.op = OP_STATS; j.b = b; j.nrec = nrec; j.table = is_gap;
    if (int rc = run_job(m, j)) return rc;
    for (size_t i = 0; i < m->w.size(); i++) m->partials[i] = m->w[i].partial;
    if (m->w.size() > 1) {
        // the path's one collective: the totals of the shares all-reduced over RCCL (contexts that share
        // a GPU are merged on the host first)
        if (int rc = skh_multi_comm_init(m)) return rc;
        if (int rc = skh_stats_allreduce(m->comm, m->partials.data())) return rc;
    }
    skh_stats_merge(inout, &m->partials[0]);
    return SKH_OK;
}

}  // extern "C"
