// capi.hip -- the extern "C" surface of libseqkit_hip.so (see include/seqkit_hip.h).
//
// Host-side only: argument checks, table building, stream/buffer ownership and the
// launches declared in kernels.h.  No torch types, no exceptions across the ABI.
#include "seqkit_hip.h"

#include <hip/hip_runtime.h>
#include <math.h>

#include <atomic>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "kernels.h"
#include "ctx.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, const char* a = "", const char* b = "")
{
    char buf[512];
    snprintf(buf, sizeof buf, fmt, a, b);
    g_err = buf;
    return code;
}

int hip_fail(hipError_t e, const char* what)
{
    return fail(e == hipErrorOutOfMemory ? SKH_ENOMEM : SKH_EHIP, "%s: %s", what, hipGetErrorString(e));
}

#define HIP_TRY(expr)                                           \
    do {                                                        \
        hipError_t e_ = (expr);                                 \
        if (e_ != hipSuccess) return hip_fail(e_, #expr);       \
    } while (0)

// Device scratch of a launch (its BatchHint, the partial sums of the stats kernel) is owned by the
// STREAM the launch goes to: launches on one stream run one after the other, so they can share a
// slot; launches on different streams get different slots however many are in flight.
constexpr uint32_t N_SLOTS = 256;


int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

}  // namespace

// Scratch of skh_stats_static_dev: the partial gap sums of one launch of the stats kernel, in the
// library's own image instead of allocated memory.  Every GPU holds its own copy at its own address
// (skh_init asks for it with hipGetSymbolAddress, with that GPU current); host code uses that
// address, never this name.
static __device__ __align__(256) unsigned long long scratch[skh::STATS_SLOT_WORDS];

int skh::set_error(int code, const char* fmt, const char* a, const char* b) { return fail(code, fmt, a, b); }


skh::LaunchCfg skh::cfg_for(const skh_ctx* ctx, uint64_t nrec, uint64_t total)
{
    skh::LaunchCfg c;
    c.max_grid = ctx->max_grid;
    c.n_cu = ctx->n_cu;
    // Reverse, records of at most LONG_T bytes.  Measured on resident batches of 2-10 Gbases (% of HBM peak,
    // staged / direct / span kernel): reads of 100-250 bases 55-61 / 51-57 / -; 350 +- 70 bases 45 / 51 / -
    // (a wavefront's stretch of 8 such records overflows the stage every other step); 700 +- 140: 50 / 56 / -;
    // 1000 +- 200: 51 / 55 / 41; 4000 +- 800: 57 / 64 / 46; 16000 +- 3200: 59 / 60 / 54; nanopore-like
    // log-normal lengths around 8 kb: 57 / 59 / 54.  So: the LDS-staged kernel for short reads, the direct
    // lane-group kernel from a mean of 300 bases on; the span kernel (lanes own consecutive output chunks)
    // only when SKH_SPAN_U asks for it.  SKH_STAGED = 0 / 1 overrides the first choice.
    const uint64_t mean_len = nrec ? total / nrec : 0;
    c.span_u = ctx->span_u >= 0 ? ctx->span_u : 0;
    c.fuse_fastq = ctx->fuse_fastq;
    c.staged = ctx->staged >= 0 ? (ctx->staged != 0) : (mean_len < 300 ? 1 : 0);
    // The flat kernel (SKH_STAGED = 2 forces it, 0 / 1 switch it off), % of HBM peak against the choice above on
    // the same machine: reads of 100 bases 64-66 / 56, 250 bases 65 / 62, 30-151 bases 53 / 48, 350 +- 70: 61 / 56,
    // 1000 +- 200: 62 / 57, 2000 +- 400: 63 / 59; level at 50, 75 and 150 bases (150: 63.1 / 63.7, left with the
    // staged kernel); behind from 4000 +- 800 on (58.5 / 61), where a step is one record, and on nanopore-like
    // lengths (57.5 / 61), whose records beyond the stage it writes straight to global memory.
    c.flat = ctx->staged >= 0 ? (ctx->staged == 2) : (mean_len < 128 || (mean_len >= 192 && mean_len < 3000));
    c.flat_steps = ctx->staged_steps;
    // in place, long records: tiles per workgroup run.  Measured (% of HBM peak, runs of 1 / 2 / 4 tiles): 24
    // chromosomes 65 / 63 / 61, short reads with 200 contigs among them 51 / 50 / 49, nanopore-like reads (too many
    // long records for the launch's list: every run searches and walks the offsets) 56 / 59 / 60.  A batch of at
    // most LONG_LIST records is sure to have its long records listed: one tile; otherwise two.
    c.inplace_run = ctx->inplace_run >= 0 ? ctx->inplace_run : (nrec <= skh::LONG_LIST ? 0 : 1);
    // in place the flat kernel is ahead or level at every mean length (% of HBM peak against the staged /
    // window-pair kernels: 100 bases 60 / 51, 150: 59 / 58, 250: 61.5 / 59, 30-151: 48 / 40, 500 +- 100: 60 / 46,
    // 1000 +- 200: 61 / 50, 4000 +- 800: 59 / 57, 6000 +- 1200: 63 / 58, 10000 +- 2000: 64 / 59, nanopore-like
    // lengths 59 / 56, hairpin-like lengths 47 / 48): its steps that do not fit the stage reverse by window pairs
    // with the whole wavefront on one record
    c.flat_inplace = ctx->staged >= 0 ? (ctx->staged == 2) : 1;
    // staged reverse kernel, steps per wavefront (SKH_STAGED_STEPS overrides).  Measured on batches of
    // 5-10 M records: 150-base reads fastest with 2; 250-base reads 58 -> 63 % of HBM peak with 3;
    // reads of about 100 bases and ragged 30-151-base reads gain 4-5 points with 6, where fewer of
    // a lane group's chunk slots hold data and a wavefront has less in flight per step.
    c.staged_steps = ctx->staged_steps ? ctx->staged_steps : mean_len < 128 ? 6 : (mean_len >= 192 && mean_len < 512) ? 3 : 2;
    c.short_steps = ctx->short_steps;
    c.count_steps = ctx->count_steps;
    c.stats_steps = ctx->stats_steps;
    c.use_hint = ctx->use_hint;
    if (ctx->force_logG >= 0) {
        c.logG = c.logG_count = ctx->force_logG > 6 ? 6 : ctx->force_logG;
        c.count_u = ctx->count_u ? ctx->count_u : 3;
        c.use_hint = 0;
        return c;
    }
    c.logG = skh::lanes_log2_reverse(mean_len);
    c.count_u = ctx->count_u ? ctx->count_u : skh::count_chunks_per_lane(mean_len);      // SKH_COUNT_U overrides
    c.logG_count = skh::lanes_log2_count(mean_len, c.count_u);
    return c;
}

// The scratch slot of stream `st` on this context.  The library's own streams give their slot back
// when they are destroyed (slot_release); with more than N_SLOTS caller streams alive at once the
// device is drained (nothing in flight can then share a slot with a later launch) and the table
// starts again.
int skh::slot_for(skh_ctx* ctx, hipStream_t st, uint32_t* slot)
{
    std::lock_guard<std::mutex> lock(ctx->slot_mu);
    auto it = ctx->slot_of.find(st);
    if (it == ctx->slot_of.end()) {
        if (ctx->slot_of.size() >= N_SLOTS) {
            HIP_TRY(hipDeviceSynchronize());
            ctx->slot_of.clear();
        }
        bool used[N_SLOTS] = {false};
        for (const auto& kv : ctx->slot_of) used[kv.second] = true;
        uint32_t free_slot = 0;
        while (used[free_slot]) free_slot++;
        it = ctx->slot_of.emplace(st, free_slot).first;
    }
    *slot = it->second;
    return SKH_OK;
}

void skh::slot_release(skh_ctx* ctx, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(ctx->slot_mu);
    ctx->slot_of.erase(st);
}

namespace {

using skh::cfg_for;
using skh::slot_for;
using skh::slot_release;

int check_plane_args(const void* a, const void* offs, uint64_t nrec, uint64_t total)
{
    if (nrec != 0 && offs == nullptr) return fail(SKH_EINVAL, "offsets pointer is NULL");
    if (total != 0 && a == nullptr) return fail(SKH_EINVAL, "sequence pointer is NULL");
    if (reinterpret_cast<uintptr_t>(a) & 15) return fail(SKH_EINVAL, "sequence plane is not 16-byte aligned");
    if (total != 0 && nrec == 0) return fail(SKH_EINVAL, "total_bytes > 0 with zero records");
    return SKH_OK;
}

}  // namespace

extern "C" {

const char* skh_last_error(void) { return g_err.c_str(); }
const char* skh_version(void) { return "seqkit-hip 0.1 gfx950"; }
uint64_t skh_padded_size(uint64_t total_bytes) { return (total_bytes + 15) & ~15ull; }

int skh_init(int device, skh_ctx** out)
{
    if (!out) return fail(SKH_EINVAL, "skh_init: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(SKH_ENODEV, "no HIP device visible (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(SKH_ENODEV, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    skh_ctx* c = new (std::nothrow) skh_ctx;
    if (!c) return fail(SKH_ENOMEM, "out of host memory");
    c->device = device;
    c->max_grid = (uint32_t)env_int("SKH_MAX_GRID", 0);
    c->n_cu = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    c->force_logG = env_int("SKH_LOGG", -1);
    c->span_u = env_int("SKH_SPAN_U", -1);
    c->fuse_fastq = env_int("SKH_FUSE_FASTQ", 0);
    c->use_hint = env_int("SKH_HINT", 1);
    c->staged = env_int("SKH_STAGED", -1);
    c->staged_steps = (uint32_t)env_int("SKH_STAGED_STEPS", 0);
    c->inplace_run = env_int("SKH_INPLACE_RUN", -1);
    if (c->inplace_run > 2) c->inplace_run = -1;
    c->short_steps = (uint32_t)env_int("SKH_SHORT_STEPS", 0);
    c->count_steps = (uint32_t)env_int("SKH_COUNT_STEPS", 0);
    c->count_u = env_int("SKH_COUNT_U", 0);
    if (c->count_u < 3 || c->count_u > 5) c->count_u = 0;
    c->stats_steps = (uint32_t)env_int("SKH_STATS_STEPS", 0);
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_hints), N_SLOTS * sizeof(skh::BatchHint));
    if (e == hipSuccess) e = hipMemset(c->d_hints, 0, N_SLOTS * sizeof(skh::BatchHint));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_stat_slots), (size_t)N_SLOTS * skh::STATS_SLOT_WORDS * 8);
    // this GPU's address of the static scratch array (the device was made current above)
    if (e == hipSuccess) e = hipGetSymbolAddress(reinterpret_cast<void**>(&c->d_static_scratch), HIP_SYMBOL(scratch));
    if (e != hipSuccess) {
        skh_destroy(c);
        return hip_fail(e, "skh_init");
    }
    *out = c;
    return SKH_OK;
}

void skh_destroy(skh_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->d_hints) (void)hipFree(c->d_hints);
    if (c->d_stat_slots) (void)hipFree(c->d_stat_slots);
    delete c;
}

int skh_sync(skh_ctx* c)
{
    if (!c) return fail(SKH_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SKH_OK;
}

// ---------------------------------------------------------------- tables

int skh_pair_table(int alphabet, uint8_t lut[256])
{
    // PairLetter tables of the reader's alphabets (bio/seq alphabet.go): letters map
    // to their pairs, gap (" -.") and ambiguous ("nN.") letters and every byte
    // outside the alphabet to themselves.  Protein pairs are the letters themselves
    // and Unlimit has no table, so both are the identity.
    if (!lut) return fail(SKH_EINVAL, "lut is NULL");
    for (int i = 0; i < 256; i++) lut[i] = (uint8_t)i;
    const char *from = "", *to = "";
    switch (alphabet) {
        case 0: from = "acgtACGT"; to = "tgcaTGCA"; break;
        case 1: from = "acgtryswkmbdhvACGTRYSWKMBDHV"; to = "tgcayrswmkvhdbTGCAYRSWMKVHDB"; break;
        case 2: from = "acguACGU"; to = "ugcaUGCA"; break;
        case 3: from = "acguryswkmbdhvACGURYSWKMBDHV"; to = "ugcayrswmkvhdbUGCAYRSWMKVHDB"; break;
        case 4: case 5: break;
        default: return fail(SKH_EINVAL, "unknown alphabet id");
    }
    for (size_t i = 0; from[i]; i++) lut[(uint8_t)from[i]] = (uint8_t)to[i];
    return SKH_OK;
}

int skh_valid_table(int alphabet, uint8_t valid[256])
{
    // members of an alphabet: its letters, gap letters and ambiguous letters (bio/seq alphabet.go);
    // Unlimit accepts every byte
    if (!valid) return fail(SKH_EINVAL, "valid is NULL");
    const char* members = nullptr;
    switch (alphabet) {
        case 0: members = "acgtACGT -.nN"; break;
        case 1: members = "acgtryswkmbdhvACGTRYSWKMBDHV -.nN"; break;
        case 2: members = "acguACGU -.nN"; break;
        case 3: members = "acguryswkmbdhvACGURYSWKMBDHV -.nN"; break;
        case 4: members = "abcdefghijklmnpqrstvwyzABCDEFGHIJKLMNPQRSTVWYZ*_. -xX"; break;
        case 5: memset(valid, 1, 256); return SKH_OK;
        default: return fail(SKH_EINVAL, "unknown alphabet id");
    }
    memset(valid, 0, 256);
    for (const char* p = members; *p; p++) valid[(uint8_t)*p] = 1;
    return SKH_OK;
}

int skh_quality_table(int from, int to, int force, uint8_t lut[256], uint8_t valid[256], int* identity)
{
    if (!lut || !valid) return fail(SKH_EINVAL, "NULL argument");
    if (from < SKH_Q_SANGER || from > SKH_Q_ILLUMINA18 || to < SKH_Q_SANGER || to > SKH_Q_ILLUMINA18)
        return fail(SKH_EINVAL, "unknown quality encoding");
    auto offset = [](int e) { return (e == SKH_Q_SANGER || e == SKH_Q_ILLUMINA18) ? 33 : 64; };
    const int off_from = offset(from), off_to = offset(to);
    const bool sol_from = from == SKH_Q_SOLEXA, sol_to = to == SKH_Q_SOLEXA;
    // nothing changes between Sanger and Illumina-1.8+ unless scores above 40 are to be cut (force)
    const bool same = from == to || (!force && off_from == 33 && off_to == 33);
    if (identity) *identity = same ? 1 : 0;
    for (int c = 0; c < 256; c++) {
        lut[c] = (uint8_t)c;
        valid[c] = 1;
        if (same) continue;
        double q = (double)(c - off_from);
        if (sol_from) {
            if (q < -5) { valid[c] = 0; continue; }                    // "seq: invalid Solexa quality"
            if (!sol_to) q = 10.0 * log10(pow(10.0, q / 10.0) + 1.0);
        } else {
            if (q < 0) { valid[c] = 0; continue; }                     // "seq: invalid Phred quality"
            if (from == SKH_Q_ILLUMINA15 && q <= 2) q = 0;             // 'B', the read segment quality control indicator
            if (sol_to) {
                q = q <= 0 ? -5.0 : 10.0 * log10(pow(10.0, q / 10.0) - 1.0);
                if (q < -5) q = -5;
            }
        }
        long r = (long)floor(q + 0.5);
        if (to == SKH_Q_ILLUMINA15 && r < 2) r = 2;
        if (force && to == SKH_Q_SANGER && r > 40) r = 40;
        lut[c] = (uint8_t)(r + off_to);
    }
    return SKH_OK;
}

int skh_class_table(const char* const* lists, uint32_t nlists, uint8_t class_of[256])
{
    if (!class_of || (nlists && !lists)) return fail(SKH_EINVAL, "NULL argument");
    if (nlists > SKH_MAX_CLASSES) return fail(SKH_EINVAL, "more than SKH_MAX_CLASSES lists");
    memset(class_of, SKH_NO_CLASS, 256);
    for (uint32_t i = 0; i < nlists; i++) {
        if (!lists[i]) return fail(SKH_EINVAL, "NULL list");
        for (const char* p = lists[i]; *p; p++) {
            uint8_t b = (uint8_t)*p, up = b, lo = b;
            if (b >= 'a' && b <= 'z') up = (uint8_t)(b - 32);
            if (b >= 'A' && b <= 'Z') lo = (uint8_t)(b + 32);
            const uint8_t both[2] = {up, lo};
            for (int k = 0; k < 2; k++) {
                if (class_of[both[k]] != SKH_NO_CLASS && class_of[both[k]] != i)
                    return fail(SKH_EINVAL, "a letter appears in two class lists");
                class_of[both[k]] = (uint8_t)i;
            }
        }
    }
    return SKH_OK;
}

// ---------------------------------------------------------------- device entry points

static int revcomp_impl(skh_ctx* ctx, const uint8_t* d_in, uint8_t* d_out, const uint8_t* d_qin,
                        uint8_t* d_qout, const uint64_t* d_offs, uint64_t nrec, uint64_t total,
                        const uint8_t* lut, int do_reverse, hipStream_t st)
{
    if (!ctx) return fail(SKH_EINVAL, "ctx is NULL");
    int rc = check_plane_args(d_in, d_offs, nrec, total);
    if (rc) return rc;
    if (total != 0 && !d_out) return fail(SKH_EINVAL, "output pointer is NULL");
    if (reinterpret_cast<uintptr_t>(d_out) & 15) return fail(SKH_EINVAL, "output plane is not 16-byte aligned");
    if ((d_qin == nullptr) != (d_qout == nullptr)) return fail(SKH_EINVAL, "quality in/out must both be given");
    if ((reinterpret_cast<uintptr_t>(d_qin) | reinterpret_cast<uintptr_t>(d_qout)) & 15)
        return fail(SKH_EINVAL, "quality plane is not 16-byte aligned");
    if (total == 0) return SKH_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    skh::Table256 t;
    if (lut) memcpy(t.v, lut, 256);
    const skh::LaunchCfg cfg = cfg_for(ctx, nrec, total);
    if (do_reverse) {
        uint32_t slot;
        if ((rc = slot_for(ctx, st, &slot))) return rc;
        HIP_TRY(skh::launch_reverse(cfg, d_in, d_out, d_qin, d_qout, d_offs, nrec, total,
                                    lut ? &t : nullptr, ctx->d_hints + slot, st));
#ifdef SKH_STAGE_CHECK
        // checked build (tests only): the staged kernel compared every stage buffer with global memory
        skh::BatchHint h;
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(&h, ctx->d_hints + slot, sizeof h, hipMemcpyDeviceToHost));
        if (h.stage_check != 0)
            return fail(SKH_EHIP, "staged reverse kernel: %s wavefront steps found their stage buffer incomplete",
                        std::to_string(h.stage_check).c_str());
#endif
    } else {
        if (lut) HIP_TRY(skh::launch_map_plane(cfg, d_in, d_out, total, t, st));
        else HIP_TRY(hipMemcpyAsync(d_out, d_in, total, hipMemcpyDeviceToDevice, st));
        if (d_qin) HIP_TRY(hipMemcpyAsync(d_qout, d_qin, total, hipMemcpyDeviceToDevice, st));
    }
    return SKH_OK;
}

int skh_revcomp_dev(skh_ctx* ctx, const uint8_t* d_in, uint8_t* d_out, const uint64_t* d_offsets,
                    uint64_t nrec, uint64_t total_bytes, const uint8_t* lut, int do_reverse, void* stream)
{
    hipStream_t st = stream ? (hipStream_t)stream : (ctx ? ctx->stream : nullptr);
    return revcomp_impl(ctx, d_in, d_out, nullptr, nullptr, d_offsets, nrec, total_bytes, lut, do_reverse, st);
}

int skh_kernel_for(skh_ctx* ctx, int op, uint64_t nrec, uint64_t total, int variant, char* name, uint64_t name_bytes)
{
    if (!ctx || !name) return fail(SKH_EINVAL, "ctx or name is NULL");
    const skh::LaunchCfg cfg = cfg_for(ctx, nrec, total);
    char buf[96];
    const char* tf = variant ? "true" : "false";
    if (op == SKH_OP_REVCOMP) {
        if (cfg.span_u) snprintf(buf, sizeof buf, "skh::k_reverse_span<%s, %d>", tf, cfg.span_u);
        else if (cfg.flat) snprintf(buf, sizeof buf, "skh::k_reverse_flat<%s>", tf);
        else if (cfg.staged) snprintf(buf, sizeof buf, "skh::k_reverse_staged<%s, false>", tf);
        else snprintf(buf, sizeof buf, "skh::k_reverse_short<%s, false>", tf);
    } else if (op == SKH_OP_REVCOMP_INPLACE) {
        if (cfg.flat_inplace) snprintf(buf, sizeof buf, "skh::k_reverse_flat_inplace<%s>", tf);
        else if (cfg.staged) snprintf(buf, sizeof buf, "skh::k_reverse_staged<%s, true>", tf);
        else snprintf(buf, sizeof buf, "skh::k_reverse_inplace_short<%s>", tf);
    } else if (op == SKH_OP_COUNT) {
        snprintf(buf, sizeof buf, "skh::k_count_short<%d, %d>", variant <= 4 ? 1 : 2, cfg.count_u == 4 || cfg.count_u == 5 ? cfg.count_u : 3);
    } else {
        return fail(SKH_EINVAL, "unknown op");
    }
    if (strlen(buf) + 1 > name_bytes) return fail(SKH_EINVAL, "name buffer too small");
    memcpy(name, buf, strlen(buf) + 1);
    return SKH_OK;
}

int skh_revcomp_fastq_dev(skh_ctx* ctx, const uint8_t* d_seq_in, uint8_t* d_seq_out,
                          const uint8_t* d_qual_in, uint8_t* d_qual_out, const uint64_t* d_offsets,
                          uint64_t nrec, uint64_t total_bytes, const uint8_t* lut, int do_reverse,
                          void* stream)
{
    if (total_bytes != 0 && (!d_qual_in || !d_qual_out)) return fail(SKH_EINVAL, "quality pointer is NULL");
    hipStream_t st = stream ? (hipStream_t)stream : (ctx ? ctx->stream : nullptr);
    return revcomp_impl(ctx, d_seq_in, d_seq_out, d_qual_in, d_qual_out, d_offsets, nrec, total_bytes, lut,
                        do_reverse, st);
}

int skh_revcomp_inplace_dev(skh_ctx* ctx, uint8_t* d_seq, uint8_t* d_qual, const uint64_t* d_offsets,
                            uint64_t nrec, uint64_t total_bytes, const uint8_t* lut, int do_reverse, void* stream)
{
    if (!ctx) return fail(SKH_EINVAL, "ctx is NULL");
    int rc = check_plane_args(d_seq, d_offsets, nrec, total_bytes);
    if (rc) return rc;
    if (reinterpret_cast<uintptr_t>(d_qual) & 15) return fail(SKH_EINVAL, "quality plane is not 16-byte aligned");
    if (d_qual && d_qual == d_seq) return fail(SKH_EINVAL, "sequence and quality planes are the same memory");
    if (total_bytes == 0) return SKH_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    skh::Table256 t;
    if (lut) memcpy(t.v, lut, 256);
    const skh::LaunchCfg cfg = cfg_for(ctx, nrec, total_bytes);
    if (!do_reverse) {
        // every 16-byte chunk is read and then written by the same lane; qualities stay as they are
        if (lut) HIP_TRY(skh::launch_map_plane(cfg, d_seq, d_seq, total_bytes, t, st));
        return SKH_OK;
    }
    uint32_t slot;
    if ((rc = slot_for(ctx, st, &slot))) return rc;
    // the launches of one stream run in order, so the planes can share the launch scratch
    HIP_TRY(skh::launch_reverse_inplace(cfg, d_seq, d_offsets, nrec, total_bytes, lut ? &t : nullptr,
                                        ctx->d_hints + slot, st));
    if (d_qual)
        HIP_TRY(skh::launch_reverse_inplace(cfg, d_qual, d_offsets, nrec, total_bytes, nullptr, ctx->d_hints + slot, st));
    return SKH_OK;
}

int skh_count_dev(skh_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offsets, uint64_t nrec,
                  uint64_t total_bytes, const uint8_t class_of[256], uint32_t nclass, uint64_t* d_counts,
                  void* stream)
{
    if (!ctx) return fail(SKH_EINVAL, "ctx is NULL");
    if (!class_of) return fail(SKH_EINVAL, "class_of is NULL");
    if (nclass < 1 || nclass > SKH_MAX_CLASSES) return fail(SKH_EINVAL, "nclass must be 1..SKH_MAX_CLASSES");
    int rc = check_plane_args(d_seq, d_offsets, nrec, total_bytes);
    if (rc) return rc;
    if (nrec == 0) return SKH_OK;
    if (!d_counts) return fail(SKH_EINVAL, "counts pointer is NULL");
    skh::Table256 t;
    uint32_t neutral = 256;
    for (int b = 0; b < 256; b++) {
        const uint8_t c = class_of[b];
        if (c != SKH_NO_CLASS && c >= nclass) return fail(SKH_EINVAL, "class_of holds a class >= nclass");
        t.v[b] = c;
        if (c == SKH_NO_CLASS && neutral == 256) neutral = (uint32_t)b;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    uint32_t slot;
    if ((rc = slot_for(ctx, st, &slot))) return rc;
    HIP_TRY(skh::launch_count(cfg_for(ctx, nrec, total_bytes), d_seq, d_offsets, nrec, total_bytes, t, nclass,
                              neutral, d_counts, ctx->d_hints + slot, st));
    return SKH_OK;
}

int skh_stats_dev(skh_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offsets, uint64_t nrec,
                  uint64_t total_bytes, const uint8_t is_gap[256], skh_stats_partial* d_out, void* stream)
{
    static_assert(sizeof(skh_stats_partial) == sizeof(skh::StatsPartial), "layout");
    if (!ctx) return fail(SKH_EINVAL, "ctx is NULL");
    if (!is_gap || !d_out) return fail(SKH_EINVAL, "NULL argument");
    int rc = check_plane_args(d_seq, d_offsets, nrec, total_bytes);
    if (rc) return rc;
    skh::Table256 t;
    memcpy(t.v, is_gap, 256);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    uint32_t slot;
    if ((rc = slot_for(ctx, st, &slot))) return rc;
    uint64_t* slots = ctx->d_stat_slots + (uint64_t)slot * skh::STATS_SLOT_WORDS;
    HIP_TRY(skh::launch_stats(cfg_for(ctx, nrec, total_bytes), d_seq, d_offsets, nrec, total_bytes, t,
                              reinterpret_cast<skh::StatsPartial*>(d_out), slots, st));
    return SKH_OK;
}

int skh_stats_static_dev(skh_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offsets, uint64_t nrec,
                         uint64_t total_bytes, const uint8_t is_gap[256], skh_stats_partial* d_out, void* stream)
{
    // skh_stats_dev with the kernel's partial sums in the static scratch array of this GPU: no
    // allocation here or on any path below (a memset node and three kernel launches)
    if (!ctx) return fail(SKH_EINVAL, "ctx is NULL");
    if (!is_gap || !d_out) return fail(SKH_EINVAL, "NULL argument");
    if (total_bytes > SKH_STATS_STATIC_MAX_BYTES)
        return fail(SKH_ECAPACITY, "skh_stats_static_dev: total_bytes exceeds SKH_STATS_STATIC_MAX_BYTES");
    if (nrec > SKH_STATS_STATIC_MAX_RECORDS)
        return fail(SKH_ECAPACITY, "skh_stats_static_dev: nrec exceeds SKH_STATS_STATIC_MAX_RECORDS");
    int rc = check_plane_args(d_seq, d_offsets, nrec, total_bytes);
    if (rc) return rc;
    if (!ctx->d_static_scratch) return fail(SKH_EHIP, "skh_stats_static_dev: the static scratch array has no device address");
    skh::Table256 t;
    memcpy(t.v, is_gap, 256);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    HIP_TRY(skh::launch_stats(cfg_for(ctx, nrec, total_bytes), d_seq, d_offsets, nrec, total_bytes, t,
                              reinterpret_cast<skh::StatsPartial*>(d_out),
                              reinterpret_cast<uint64_t*>(ctx->d_static_scratch), st));
    return SKH_OK;
}

int skh_letters_dev(skh_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offsets, uint64_t nrec, uint64_t total_bytes,
                    uint32_t* d_present, void* stream)
{
    if (!ctx) return fail(SKH_EINVAL, "ctx is NULL");
    int rc = check_plane_args(d_seq, d_offsets, nrec, total_bytes);
    if (rc) return rc;
    if (nrec == 0) return SKH_OK;
    if (!d_present) return fail(SKH_EINVAL, "output pointer is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    uint32_t slot;
    if ((rc = slot_for(ctx, st, &slot))) return rc;
    HIP_TRY(skh::launch_letters(cfg_for(ctx, nrec, total_bytes), d_seq, d_offsets, nrec, total_bytes, d_present,
                                ctx->d_hints + slot, st));
    return SKH_OK;
}

uint64_t skh_remove_gaps_workspace_size(uint64_t nrec, uint64_t total_bytes)
{
    return skh::remove_gaps_workspace_size(nrec, total_bytes);
}

int skh_remove_gaps_dev(skh_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_qual, const uint64_t* d_offsets, uint64_t nrec,
                        uint64_t total_bytes, const uint8_t is_gap[256], uint8_t* d_out_seq, uint8_t* d_out_qual,
                        uint64_t* d_out_offsets, void* d_workspace, uint64_t workspace_bytes, uint64_t* d_out_bytes,
                        void* stream)
{
    if (!ctx) return fail(SKH_EINVAL, "ctx is NULL");
    if (!is_gap || !d_out_offsets || !d_out_bytes || !d_workspace) return fail(SKH_EINVAL, "NULL argument");
    int rc = check_plane_args(d_seq, d_offsets, nrec, total_bytes);
    if (rc) return rc;
    if (total_bytes != 0 && !d_out_seq) return fail(SKH_EINVAL, "output plane is NULL");
    if (d_qual && !d_out_qual) return fail(SKH_EINVAL, "quality plane given without an output plane for it");
    if (reinterpret_cast<uintptr_t>(d_qual) & 15) return fail(SKH_EINVAL, "quality plane is not 16-byte aligned");
    if (reinterpret_cast<uintptr_t>(d_workspace) & 255) return fail(SKH_EINVAL, "workspace is not 256-byte aligned");
    if (workspace_bytes < skh::remove_gaps_workspace_size(nrec, total_bytes))
        return fail(SKH_ECAPACITY, "workspace smaller than skh_remove_gaps_workspace_size(nrec, total_bytes)");
    skh::Table256 t;
    for (int b = 0; b < 256; b++) t.v[b] = is_gap[b] ? 1 : 0;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    uint32_t slot;
    if ((rc = slot_for(ctx, st, &slot))) return rc;
    HIP_TRY(skh::launch_remove_gaps(cfg_for(ctx, nrec, total_bytes), d_seq, d_qual, d_offsets, nrec, total_bytes, t, d_out_seq,
                                    d_out_qual, d_out_offsets, d_workspace, d_out_bytes, ctx->d_hints + slot, st));
    return SKH_OK;
}

int skh_validate_dev(skh_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offsets, uint64_t nrec, uint64_t total_bytes,
                     const uint8_t valid[256], uint64_t limit, skh_validate_result* d_out, void* stream)
{
    static_assert(sizeof(skh_validate_result) == sizeof(skh::ValidateResult), "layout");
    if (!ctx) return fail(SKH_EINVAL, "ctx is NULL");
    if (!valid || !d_out) return fail(SKH_EINVAL, "NULL argument");
    int rc = check_plane_args(d_seq, d_offsets, nrec, total_bytes);
    if (rc) return rc;
    skh::Table256 t;
    memcpy(t.v, valid, 256);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    HIP_TRY(skh::launch_validate(cfg_for(ctx, nrec, total_bytes), d_seq, d_offsets, nrec, total_bytes, t, limit,
                                 reinterpret_cast<skh::ValidateResult*>(d_out), st));
    return SKH_OK;
}

// ---------------------------------------------------------------- text -> batch

uint64_t skh_subseq_workspace_size(uint64_t n) { return skh::subseq_workspace_size(n); }

int skh_subseq_dev(skh_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_qual, const uint64_t* d_offsets, uint64_t nrec,
                   uint64_t total_bytes, const skh_slice* d_slices, uint64_t n, int64_t start, int64_t end, int minus,
                   const uint8_t* lut, uint8_t* d_out_seq, uint8_t* d_out_qual, uint64_t out_capacity,
                   uint64_t* d_out_offsets, void* d_workspace, uint64_t workspace_bytes, skh_subseq_result* d_result,
                   void* stream)
{
    static_assert(sizeof(skh_slice) == sizeof(skh::Slice) && sizeof(skh_subseq_result) == sizeof(skh::SubseqResult),
                  "ABI structs mirror the kernel structs");
    if (!ctx || !d_out_offsets || !d_workspace || !d_result) return fail(SKH_EINVAL, "NULL argument");
    if (int rc = check_plane_args(d_seq, d_offsets, nrec, total_bytes)) return rc;
    if (!d_out_seq || (d_qual && !d_out_qual)) return fail(SKH_EINVAL, "NULL output plane");
    if ((reinterpret_cast<uintptr_t>(d_out_seq) & 15) || (reinterpret_cast<uintptr_t>(d_out_qual) & 15))
        return fail(SKH_EINVAL, "output planes must be 16-byte aligned");
    if (reinterpret_cast<uintptr_t>(d_workspace) & 255) return fail(SKH_EINVAL, "workspace must be 256-byte aligned");
    if (!d_slices && n != nrec && n != 0) return fail(SKH_EINVAL, "region mode needs one slice per record");
    if (n >= (1ull << 40)) return fail(SKH_EINVAL, "too many slices");
    if (workspace_bytes < skh::subseq_workspace_size(n)) return fail(SKH_ECAPACITY, "subseq workspace too small");
    HIP_TRY(hipSetDevice(ctx->device));
    skh::Table256 t;
    if (lut) memcpy(t.v, lut, 256);
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    HIP_TRY(skh::launch_subseq(cfg_for(ctx, nrec, total_bytes), d_seq, d_qual, d_offsets, nrec, total_bytes,
                               reinterpret_cast<const skh::Slice*>(d_slices), n, start, end, minus ? 1u : 0u,
                               lut ? &t : nullptr, d_out_seq, d_out_qual, out_capacity, d_out_offsets, d_workspace,
                               reinterpret_cast<skh::SubseqResult*>(d_result), st));
    return SKH_OK;
}

uint64_t skh_parse_workspace_size(uint64_t text_bytes, uint64_t max_lines)
{
    return skh::parse_workspace_size(text_bytes, max_lines ? max_lines : 1);
}

uint64_t skh_parse_wrapped_workspace_size(uint64_t text_bytes, uint64_t max_lines, uint64_t max_records)
{
    return skh::parse_wrapped_workspace_size(text_bytes, max_lines ? max_lines : 1, max_records);
}

int skh_parse_dev(skh_ctx* ctx, const uint8_t* d_text, uint64_t text_bytes, int format, int final,
                  uint8_t* d_seq, uint8_t* d_qual, uint64_t seq_capacity, uint64_t* d_offsets,
                  uint64_t* d_name_pos, uint32_t* d_name_len, uint64_t max_records, void* d_workspace,
                  uint64_t workspace_bytes, uint64_t max_lines, skh_parse_result* d_result, void* stream)
{
    static_assert(sizeof(skh_parse_result) == sizeof(skh::ParseResult), "layout");
    if (!ctx) return fail(SKH_EINVAL, "ctx is NULL");
    if (format != SKH_FASTA && format != SKH_FASTQ && format != SKH_FASTQ_WRAPPED)
        return fail(SKH_EINVAL, "format must be SKH_FASTA, SKH_FASTQ or SKH_FASTQ_WRAPPED");
    if (!d_offsets || !d_name_pos || !d_name_len || !d_workspace || !d_result) return fail(SKH_EINVAL, "NULL argument");
    if (text_bytes != 0 && (!d_text || !d_seq)) return fail(SKH_EINVAL, "NULL text or sequence buffer");
    if (format != SKH_FASTA && text_bytes != 0 && !d_qual) return fail(SKH_EINVAL, "FASTQ needs a quality buffer");
    if (reinterpret_cast<uintptr_t>(d_text) & 15) return fail(SKH_EINVAL, "text buffer is not 16-byte aligned");
    if (reinterpret_cast<uintptr_t>(d_workspace) & 255) return fail(SKH_EINVAL, "workspace is not 256-byte aligned");
    if (max_lines == 0) return fail(SKH_EINVAL, "max_lines is 0");
    if (format == SKH_FASTQ_WRAPPED) {
        if (max_lines >= (1ull << 32) - 4) return fail(SKH_EINVAL, "SKH_FASTQ_WRAPPED: at most 2^32 - 5 lines per block");
        if (workspace_bytes < skh::parse_wrapped_workspace_size(text_bytes, max_lines, max_records))
            return fail(SKH_ECAPACITY, "workspace smaller than skh_parse_wrapped_workspace_size(text_bytes, max_lines, max_records)");
        HIP_TRY(hipSetDevice(ctx->device));
        hipStream_t wst = stream ? (hipStream_t)stream : ctx->stream;
        HIP_TRY(skh::launch_parse_wrapped(d_text, text_bytes, final != 0, d_seq, d_qual, seq_capacity, d_offsets, d_name_pos,
                                          d_name_len, max_records, d_workspace, max_lines,
                                          reinterpret_cast<skh::ParseResult*>(d_result), wst));
        return SKH_OK;
    }
    if (workspace_bytes < skh::parse_workspace_size(text_bytes, max_lines))
        return fail(SKH_ECAPACITY, "workspace smaller than skh_parse_workspace_size(text_bytes, max_lines)");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    HIP_TRY(skh::launch_parse(format == SKH_FASTQ, d_text, text_bytes, final != 0, d_seq, d_qual, seq_capacity,
                              d_offsets, d_name_pos, d_name_len, max_records, d_workspace, max_lines,
                              reinterpret_cast<skh::ParseResult*>(d_result), st));
    return SKH_OK;
}

// ---------------------------------------------------------------- batch -> output text

uint64_t skh_format_workspace_size(uint64_t nrec) { return skh::format_workspace_size(nrec); }

static int format_seq_impl(skh_ctx* ctx, const uint64_t* d_filter_offsets, const uint8_t* d_text, const uint64_t* d_name_pos, const uint32_t* d_name_len,
                       const uint8_t* d_seq, const uint8_t* d_qual, const uint64_t* d_offsets, uint64_t nrec,
                       const skh_seq_format* fmt, uint8_t* d_out, uint64_t out_capacity, void* d_workspace,
                       uint64_t workspace_bytes, skh_format_result* d_result, void* stream)
{
    static_assert(sizeof(skh_format_result) == sizeof(skh::FormatResult), "layout");
    if (!ctx) return fail(SKH_EINVAL, "ctx is NULL");
    if (!fmt || !d_offsets || !d_workspace || !d_result) return fail(SKH_EINVAL, "NULL argument");
    if (nrec != 0 && !d_out) return fail(SKH_EINVAL, "output buffer is NULL");
    if (fmt->print_name && nrec != 0 && (!d_text || !d_name_pos || !d_name_len))
        return fail(SKH_EINVAL, "names requested without name spans");
    if (fmt->print_seq && nrec != 0 && !d_seq) return fail(SKH_EINVAL, "sequence requested without a sequence plane");
    if (fmt->line_width < 0) return fail(SKH_EINVAL, "line_width is negative");
    if (reinterpret_cast<uintptr_t>(d_workspace) & 255) return fail(SKH_EINVAL, "workspace is not 256-byte aligned");
    if (workspace_bytes < skh::format_workspace_size(nrec))
        return fail(SKH_ECAPACITY, "workspace smaller than skh_format_workspace_size(nrec)");
    skh::FormatSeqArgs a;
    a.text = d_text; a.name_pos = d_name_pos; a.name_len = d_name_len; a.seq = d_seq; a.qual = d_qual;
    a.offsets = d_offsets; a.filter_offsets = d_filter_offsets; a.nrec = nrec; a.min_len = fmt->min_len; a.max_len = fmt->max_len;
    a.width = (uint32_t)fmt->line_width;
    a.print_name = fmt->print_name != 0; a.print_seq = fmt->print_seq != 0; a.print_qual = fmt->print_qual != 0;
    a.plus_line = fmt->plus_line != 0; a.only_id = fmt->only_id != 0; a.mark = (uint8_t)fmt->mark;
    a.empty_fasta = fmt->empty_as_fasta != 0;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    HIP_TRY(skh::launch_format_seq(a, d_out, out_capacity, d_workspace, reinterpret_cast<skh::FormatResult*>(d_result), st));
    return SKH_OK;
}

int skh_format_seq_dev(skh_ctx* ctx, const uint8_t* d_text, const uint64_t* d_name_pos, const uint32_t* d_name_len,
                       const uint8_t* d_seq, const uint8_t* d_qual, const uint64_t* d_offsets, uint64_t nrec,
                       const skh_seq_format* fmt, uint8_t* d_out, uint64_t out_capacity, void* d_workspace,
                       uint64_t workspace_bytes, skh_format_result* d_result, void* stream)
{
    return format_seq_impl(ctx, nullptr, d_text, d_name_pos, d_name_len, d_seq, d_qual, d_offsets, nrec, fmt, d_out, out_capacity,
                           d_workspace, workspace_bytes, d_result, stream);
}

int skh_format_tab_dev(skh_ctx* ctx, const uint8_t* d_text, const uint64_t* d_name_pos, const uint32_t* d_name_len,
                       const uint8_t* d_seq, const uint8_t* d_qual, const uint64_t* d_offsets, uint64_t nrec,
                       const uint64_t* const d_counts[SKH_TAB_MAX_PASSES], const uint32_t nclass[SKH_TAB_MAX_PASSES],
                       const skh_tab_format* fmt, uint8_t* d_out, uint64_t out_capacity, void* d_workspace,
                       uint64_t workspace_bytes, skh_format_result* d_result, void* stream)
{
    static_assert(sizeof(skh_tab_column) == sizeof(skh::TabColumn), "layout");
    static_assert(SKH_TAB_MAX_COLS == skh::TAB_MAX_COLS && SKH_TAB_MAX_TERMS == skh::TAB_MAX_TERMS &&
                  SKH_TAB_MAX_PASSES == skh::TAB_MAX_PASSES, "limits");
    if (!ctx) return fail(SKH_EINVAL, "ctx is NULL");
    if (!fmt || !d_offsets || !d_workspace || !d_result || !d_counts || !nclass) return fail(SKH_EINVAL, "NULL argument");
    if (nrec != 0 && (!d_out || !d_text || !d_name_pos || !d_name_len)) return fail(SKH_EINVAL, "NULL buffer");
    if (fmt->print_seq && nrec != 0 && !d_seq) return fail(SKH_EINVAL, "sequence requested without a sequence plane");
    if (fmt->print_qual && nrec != 0 && !d_qual) return fail(SKH_EINVAL, "quality requested without a quality plane");
    if (fmt->ncols > SKH_TAB_MAX_COLS) return fail(SKH_EINVAL, "too many columns");
    if (reinterpret_cast<uintptr_t>(d_workspace) & 255) return fail(SKH_EINVAL, "workspace is not 256-byte aligned");
    if (workspace_bytes < skh::format_workspace_size(nrec))
        return fail(SKH_ECAPACITY, "workspace smaller than skh_format_workspace_size(nrec)");
    skh::FormatTabArgs a;
    memset(&a, 0, sizeof a);
    a.text = d_text; a.name_pos = d_name_pos; a.name_len = d_name_len; a.seq = d_seq; a.qual = d_qual;
    a.offsets = d_offsets; a.nrec = nrec;
    for (int p = 0; p < SKH_TAB_MAX_PASSES; p++) { a.counts[p] = d_counts[p]; a.nclass[p] = nclass[p]; }
    a.ncols = fmt->ncols;
    if (fmt->alphabet && nrec != 0 && !fmt->d_letters) return fail(SKH_EINVAL, "alphabet column requested without d_letters");
    a.letters = fmt->alphabet ? fmt->d_letters : nullptr;
    a.only_id = fmt->only_id != 0; a.print_seq = fmt->print_seq != 0; a.print_qual = fmt->print_qual != 0;
    for (uint32_t c = 0; c < fmt->ncols; c++) {
        const skh_tab_column& col = fmt->cols[c];
        if (col.kind > SKH_TAB_CONTENT) return fail(SKH_EINVAL, "unknown column kind");
        const uint32_t nterm = col.kind == SKH_TAB_LENGTH ? 0 : (col.kind == SKH_TAB_CONTENT ? col.nterm : 2);
        if (nterm > SKH_TAB_MAX_TERMS) return fail(SKH_EINVAL, "too many letters in a column");
        for (uint32_t k = 0; k < nterm; k++) {
            const uint32_t pass = col.term[k] >> 3, cls = col.term[k] & 7;
            if (pass >= SKH_TAB_MAX_PASSES || !d_counts[pass] || cls >= nclass[pass])
                return fail(SKH_EINVAL, "a column names a count that was not supplied");
        }
        a.cols[c].kind = col.kind; a.cols[c].nterm = (uint8_t)nterm;
        memcpy(a.cols[c].term, col.term, sizeof col.term);
    }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    HIP_TRY(skh::launch_format_tab(a, d_out, out_capacity, d_workspace, reinterpret_cast<skh::FormatResult*>(d_result), st));
    return SKH_OK;
}

// ---------------------------------------------------------------- pinned batches

int skh_batch_alloc(skh_ctx* ctx, uint64_t max_bytes, uint64_t max_records, int with_qual, skh_batch** out)
{
    if (!ctx || !out) return fail(SKH_EINVAL, "NULL argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    skh_batch* b = new (std::nothrow) skh_batch;
    if (!b) return fail(SKH_ENOMEM, "out of host memory");
    b->ctx = ctx;
    b->max_bytes = skh_padded_size(max_bytes ? max_bytes : 16);
    b->max_records = max_records ? max_records : 1;
    b->with_qual = with_qual != 0;
    const size_t offs_bytes = (b->max_records + 1) * sizeof(uint64_t);
    // host planes only: they are pinned for every GPU of the process (a batch may be handed to
    // skh_*_multi, whose contexts copy their shares straight out of these buffers); the device
    // planes of the batch's own context are allocated by its first single-GPU call
    hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&b->h_seq), b->max_bytes, hipHostMallocPortable);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&b->h_offs), offs_bytes, hipHostMallocPortable);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&b->h_stats), sizeof(skh::StatsPartial), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&b->d_stats), sizeof(skh::StatsPartial));
    if (e == hipSuccess && b->with_qual)
        e = hipHostMalloc(reinterpret_cast<void**>(&b->h_qual), b->max_bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
        skh_batch_free(b);
        return hip_fail(e, "skh_batch_alloc");
    }
    b->h_offs[0] = 0;
    *out = b;
    return SKH_OK;
}

void skh_batch_free(skh_batch* b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    if (b->stream) { (void)hipStreamSynchronize(b->stream); slot_release(b->ctx, b->stream); (void)hipStreamDestroy(b->stream); }
    if (b->h_seq) (void)hipHostFree(b->h_seq);
    if (b->h_qual) (void)hipHostFree(b->h_qual);
    if (b->h_offs) (void)hipHostFree(b->h_offs);
    if (b->h_counts) (void)hipHostFree(b->h_counts);
    if (b->h_stats) (void)hipHostFree(b->h_stats);
    if (b->d_seq) (void)hipFree(b->d_seq);
    if (b->d_seq_out) (void)hipFree(b->d_seq_out);
    if (b->d_qual) (void)hipFree(b->d_qual);
    if (b->d_qual_out) (void)hipFree(b->d_qual_out);
    if (b->d_offs) (void)hipFree(b->d_offs);
    if (b->d_counts) (void)hipFree(b->d_counts);
    if (b->d_stats) (void)hipFree(b->d_stats);
    delete b;
}

uint8_t* skh_batch_seq(skh_batch* b) { return b ? b->h_seq : nullptr; }
uint8_t* skh_batch_qual(skh_batch* b) { return b ? b->h_qual : nullptr; }
uint64_t* skh_batch_offsets(skh_batch* b) { return b ? b->h_offs : nullptr; }
uint64_t skh_batch_max_bytes(const skh_batch* b) { return b ? b->max_bytes : 0; }
uint64_t skh_batch_max_records(const skh_batch* b) { return b ? b->max_records : 0; }

// checks the host-side CSR and returns the batch's byte total
static int batch_total(const skh_batch* b, uint64_t nrec, uint64_t* total)
{
    if (!b) return fail(SKH_EINVAL, "batch is NULL");
    if (b->pending) return fail(SKH_EINVAL, "batch has an unfinished submit; call skh_batch_wait first");
    if (nrec > b->max_records) return fail(SKH_ECAPACITY, "more records than the batch was allocated for");
    if (b->h_offs[0] != 0) return fail(SKH_EINVAL, "offsets[0] must be 0");
    const uint64_t t = b->h_offs[nrec];
    if (t > b->max_bytes) return fail(SKH_ECAPACITY, "more sequence bytes than the batch was allocated for");
    *total = t;
    return SKH_OK;
}

static int upload(skh_batch* b, uint64_t nrec, uint64_t total, bool qual)
{
    HIP_TRY(hipSetDevice(b->ctx->device));
    if (!b->d_seq) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_seq), b->max_bytes));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_seq_out), b->max_bytes));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_offs), (b->max_records + 1) * sizeof(uint64_t)));
    }
    if (b->with_qual && !b->d_qual) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_qual), b->max_bytes));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_qual_out), b->max_bytes));
    }
    HIP_TRY(hipMemcpyAsync(b->d_offs, b->h_offs, (nrec + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, b->stream));
    const uint64_t padded = skh_padded_size(total);
    if (padded) HIP_TRY(hipMemcpyAsync(b->d_seq, b->h_seq, padded, hipMemcpyHostToDevice, b->stream));
    if (qual && padded) HIP_TRY(hipMemcpyAsync(b->d_qual, b->h_qual, padded, hipMemcpyHostToDevice, b->stream));
    return SKH_OK;
}

int skh_revcomp_submit(skh_ctx* ctx, skh_batch* b, uint64_t nrec, const uint8_t* lut, int do_reverse,
                       int reverse_qual)
{
    if (!ctx) return fail(SKH_EINVAL, "ctx is NULL");
    uint64_t total = 0;
    int rc = batch_total(b, nrec, &total);
    if (rc) return rc;
    if (b->ctx != ctx) return fail(SKH_EINVAL, "batch belongs to another context");
    if (total == 0 || (!lut && !do_reverse)) return SKH_OK;
    const bool qual = b->with_qual && reverse_qual && do_reverse;
    rc = upload(b, nrec, total, qual);
    if (rc) return rc;
    rc = revcomp_impl(ctx, b->d_seq, b->d_seq_out, qual ? b->d_qual : nullptr, qual ? b->d_qual_out : nullptr,
                      b->d_offs, nrec, total, lut, do_reverse, b->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(b->h_seq, b->d_seq_out, total, hipMemcpyDeviceToHost, b->stream));
    if (qual) HIP_TRY(hipMemcpyAsync(b->h_qual, b->d_qual_out, total, hipMemcpyDeviceToHost, b->stream));
    b->pending = true;
    return SKH_OK;
}

int skh_count_submit(skh_ctx* ctx, skh_batch* b, uint64_t nrec, const uint8_t class_of[256], uint32_t nclass,
                     uint64_t* counts)
{
    if (!ctx) return fail(SKH_EINVAL, "ctx is NULL");
    uint64_t total = 0;
    int rc = batch_total(b, nrec, &total);
    if (rc) return rc;
    if (b->ctx != ctx) return fail(SKH_EINVAL, "batch belongs to another context");
    if (nclass < 1 || nclass > SKH_MAX_CLASSES) return fail(SKH_EINVAL, "nclass must be 1..SKH_MAX_CLASSES");
    if (nrec == 0) return SKH_OK;
    if (!counts) return fail(SKH_EINVAL, "counts pointer is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    if (!b->d_counts) {
        const size_t bytes = b->max_records * SKH_MAX_CLASSES * sizeof(uint64_t);
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_counts), bytes));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&b->h_counts), bytes, hipHostMallocDefault));
    }
    rc = upload(b, nrec, total, false);
    if (rc) return rc;
    rc = skh_count_dev(ctx, b->d_seq, b->d_offs, nrec, total, class_of, nclass, b->d_counts, b->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(b->h_counts, b->d_counts, nrec * nclass * sizeof(uint64_t), hipMemcpyDeviceToHost,
                           b->stream));
    b->user_counts = counts;
    b->n_counts = nrec * nclass;
    b->pending = true;
    return SKH_OK;
}

int skh_batch_wait(skh_batch* b)
{
    if (!b) return fail(SKH_EINVAL, "batch is NULL");
    if (!b->pending) return SKH_OK;
    b->pending = false;
    uint64_t* dst = b->user_counts;
    const uint64_t n = b->n_counts;
    b->user_counts = nullptr;
    b->n_counts = 0;
    HIP_TRY(hipSetDevice(b->ctx->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (dst && n) memcpy(dst, b->h_counts, n * sizeof(uint64_t));
    return SKH_OK;
}

int skh_revcomp(skh_ctx* ctx, skh_batch* b, uint64_t nrec, const uint8_t* lut, int do_reverse, int reverse_qual)
{
    int rc = skh_revcomp_submit(ctx, b, nrec, lut, do_reverse, reverse_qual);
    return rc ? rc : skh_batch_wait(b);
}

int skh_count(skh_ctx* ctx, skh_batch* b, uint64_t nrec, const uint8_t class_of[256], uint32_t nclass,
              uint64_t* counts)
{
    int rc = skh_count_submit(ctx, b, nrec, class_of, nclass, counts);
    return rc ? rc : skh_batch_wait(b);
}

void skh_stats_init(skh_stats_partial* p)
{
    if (!p) return;
    p->num = p->len_sum = p->gap_sum = p->len_max = 0;
    p->len_min = INT64_MAX;
}

void skh_stats_merge(skh_stats_partial* into, const skh_stats_partial* from)
{
    if (!into || !from) return;
    into->num += from->num;
    into->len_sum += from->len_sum;
    into->gap_sum += from->gap_sum;
    if (from->len_min < into->len_min) into->len_min = from->len_min;
    if (from->len_max > into->len_max) into->len_max = from->len_max;
}

int skh_stats_reduce(skh_ctx* ctx, skh_batch* b, uint64_t nrec, const uint8_t is_gap[256],
                     skh_stats_partial* inout)
{
    if (!ctx || !inout) return fail(SKH_EINVAL, "NULL argument");
    uint64_t total = 0;
    int rc = batch_total(b, nrec, &total);
    if (rc) return rc;
    if (b->ctx != ctx) return fail(SKH_EINVAL, "batch belongs to another context");
    if (nrec == 0) return SKH_OK;
    rc = upload(b, nrec, total, false);
    if (rc) return rc;
    rc = skh_stats_dev(ctx, b->d_seq, b->d_offs, nrec, total, is_gap,
                       reinterpret_cast<skh_stats_partial*>(b->d_stats), b->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(b->h_stats, b->d_stats, sizeof(skh::StatsPartial), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    skh_stats_merge(inout, reinterpret_cast<const skh_stats_partial*>(b->h_stats));
    return SKH_OK;
}

int skh_validate(skh_ctx* ctx, skh_batch* b, uint64_t nrec, const uint8_t valid[256], uint64_t limit,
                 skh_validate_result* out)
{
    static_assert(sizeof(skh_validate_result) <= sizeof(skh::StatsPartial), "result fits the batch's result slot");
    if (!ctx || !out) return fail(SKH_EINVAL, "NULL argument");
    memset(out, 0, sizeof *out);
    uint64_t total = 0;
    int rc = batch_total(b, nrec, &total);
    if (rc) return rc;
    if (b->ctx != ctx) return fail(SKH_EINVAL, "batch belongs to another context");
    if (nrec == 0) return SKH_OK;
    rc = upload(b, nrec, total, false);
    if (rc) return rc;
    rc = skh_validate_dev(ctx, b->d_seq, b->d_offs, nrec, total, valid, limit,
                          reinterpret_cast<skh_validate_result*>(b->d_stats), b->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(b->h_stats, b->d_stats, sizeof(skh_validate_result), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    memcpy(out, b->h_stats, sizeof *out);
    return SKH_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- text blocks (file text in, file text out)

struct skh_text {
    skh_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    uint64_t cap = 0;                       // text bytes per block
    uint64_t max_lines = 0;                 // capacity of the per-line / per-record arrays
    uint8_t *h_text = nullptr, *d_text = nullptr;
    uint8_t *d_seq = nullptr, *d_qual = nullptr, *d_seq2 = nullptr, *d_qual2 = nullptr;
    uint64_t *d_offs = nullptr, *d_npos = nullptr;
    uint32_t* d_nlen = nullptr;
    void *d_ws = nullptr, *d_fws = nullptr;
    void* d_wws = nullptr;                  // workspace of SKH_FASTQ_WRAPPED, made when first asked for
    uint64_t wws_bytes = 0;
    uint64_t ws_bytes = 0, fws_bytes = 0, fws_records = 0;
    uint8_t *d_out = nullptr, *h_out = nullptr;
    uint64_t out_cap = 0;
    void* h_res = nullptr;                  // pinned: results come back here
    void* d_res = nullptr;
    // the block parsed last
    int fastq = 0;
    uint64_t nrec = 0, seq_bytes = 0, text_bytes = 0;
    uint64_t text_off = 0;                  // where the parsed text starts in h_text (skh_text_parse_at)
    uint32_t* d_letters = nullptr;          // skh_text_letters: nrec x 8 words
    uint64_t letters_cap = 0;
    bool letters_ready = false;
    uint64_t* d_goffs = nullptr;            // skh_text_remove_gaps: offsets after the removal
    void* d_gws = nullptr;
    uint64_t goffs_cap = 0, gws_bytes = 0;
    const uint64_t* filter_offs = nullptr;  // lengths the -m / -M filter of the formatter looks at (nullptr: v_offs)
    uint64_t out_pending = 0;               // bytes of h_out on their way (skh_text_format_*_submit)
    uint8_t* d_stage = nullptr;             // skh_text_upload: device copy of h_text at the same offsets
    uint64_t up_off = 0, up_bytes = 0;      // the range uploaded ahead of the next parse (up_bytes = 0: none)
    bool transformed = false;               // planes of the block now live in d_seq2 / d_qual2
    // what the later calls read: the parsed tables, or those of the slices after skh_text_subseq
    const uint64_t *v_offs = nullptr, *v_npos = nullptr;
    const uint32_t* v_nlen = nullptr;
    const uint8_t* v_text = nullptr;
    // skh_text_subseq: slice list, offsets and names of the result
    uint64_t plane_cap = 0;                 // bytes of d_seq2 / d_qual2
    uint64_t base_cap = 0;                  // bytes of d_seq / d_qual
    skh::Slice* d_slices = nullptr;
    uint64_t *d_offs2 = nullptr, *d_npos2 = nullptr;
    uint32_t* d_nlen2 = nullptr;
    void* d_sws = nullptr;
    uint64_t slice_cap = 0;
    uint8_t* d_names = nullptr;
    uint64_t names_cap = 0;
    uint64_t* d_counts[SKH_TAB_MAX_PASSES] = {nullptr, nullptr, nullptr, nullptr};   // results of skh_text_count
    uint64_t counts_cap[SKH_TAB_MAX_PASSES] = {0, 0, 0, 0};
    uint32_t nclass[SKH_TAB_MAX_PASSES] = {0, 0, 0, 0};
};

namespace {

void text_free_lines(skh_text* t)
{
    (void)hipFree(t->d_offs); (void)hipFree(t->d_npos); (void)hipFree(t->d_nlen); (void)hipFree(t->d_ws); (void)hipFree(t->d_fws);
    (void)hipFree(t->d_wws);
    t->d_offs = t->d_npos = nullptr; t->d_nlen = nullptr; t->d_ws = t->d_fws = t->d_wws = nullptr;
    t->wws_bytes = 0;
}

hipError_t text_alloc_lines(skh_text* t, uint64_t max_lines)
{
    text_free_lines(t);
    t->max_lines = max_lines;
    t->ws_bytes = skh::parse_workspace_size(t->cap, max_lines);
    t->fws_bytes = skh::format_workspace_size(max_lines);
    t->fws_records = max_lines;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&t->d_offs), (max_lines + 1) * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t->d_npos), max_lines * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t->d_nlen), max_lines * 4);
    if (e == hipSuccess) e = hipMalloc(&t->d_ws, t->ws_bytes);
    if (e == hipSuccess) e = hipMalloc(&t->d_fws, t->fws_bytes);
    return e;
}

hipError_t text_alloc_out(skh_text* t, uint64_t out_cap)
{
    (void)hipFree(t->d_out); (void)hipHostFree(t->h_out);
    t->d_out = t->h_out = nullptr;
    t->out_cap = out_cap;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&t->d_out), out_cap);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&t->h_out), out_cap, hipHostMallocDefault);
    return e;
}

}  // namespace

int skh_text_alloc(skh_ctx* ctx, uint64_t block_bytes, skh_text** out)
{
    if (!ctx || !out) return fail(SKH_EINVAL, "NULL argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    skh_text* t = new (std::nothrow) skh_text;
    if (!t) return fail(SKH_ENOMEM, "out of host memory");
    t->ctx = ctx;
    t->cap = skh_padded_size(block_bytes < 4096 ? 4096 : block_bytes);
    hipError_t e = hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&t->h_text), t->cap, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t->d_text), t->cap);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t->d_seq), t->cap);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t->d_qual), t->cap);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t->d_seq2), t->cap);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t->d_qual2), t->cap);
    t->plane_cap = t->cap;
    t->base_cap = t->cap;
    if (e == hipSuccess) e = hipHostMalloc(&t->h_res, 256, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&t->d_res, 256);
    if (e == hipSuccess) e = text_alloc_lines(t, t->cap / 32 + 1024);
    if (e == hipSuccess) e = text_alloc_out(t, t->cap + t->cap / 8 + 4096);
    if (e != hipSuccess) {
        const int rc = hip_fail(e, "allocating a text block");
        skh_text_free(t);
        return rc;
    }
    *out = t;
    return SKH_OK;
}

void skh_text_free(skh_text* t)
{
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    if (t->stream) { (void)hipStreamSynchronize(t->stream); slot_release(t->ctx, t->stream); (void)hipStreamDestroy(t->stream); }
    text_free_lines(t);
    (void)hipFree(t->d_stage); (void)hipFree(t->d_goffs); (void)hipFree(t->d_gws); (void)hipFree(t->d_letters);
    (void)hipFree(t->d_text); (void)hipFree(t->d_seq); (void)hipFree(t->d_qual); (void)hipFree(t->d_seq2); (void)hipFree(t->d_qual2);
    (void)hipFree(t->d_out); (void)hipFree(t->d_res);
    (void)hipFree(t->d_slices); (void)hipFree(t->d_offs2); (void)hipFree(t->d_npos2); (void)hipFree(t->d_nlen2);
    (void)hipFree(t->d_sws); (void)hipFree(t->d_names);
    for (auto& c : t->d_counts) (void)hipFree(c);
    (void)hipHostFree(t->h_text); (void)hipHostFree(t->h_out); (void)hipHostFree(t->h_res);
    delete t;
}

uint8_t* skh_text_buffer(skh_text* t) { return t ? t->h_text : nullptr; }
skh_ctx* skh_text_ctx(const skh_text* t) { return t ? t->ctx : nullptr; }
uint64_t skh_text_capacity(const skh_text* t) { return t ? t->cap : 0; }

int skh_text_parse(skh_ctx* ctx, skh_text* t, uint64_t text_bytes, int format, int final, skh_parse_result* out)
{
    return skh_text_parse_at(ctx, t, 0, text_bytes, format, final, out);
}

int skh_text_parse_at(skh_ctx* ctx, skh_text* t, uint64_t text_offset, uint64_t text_bytes, int format, int final,
                      skh_parse_result* out)
{
    if (!ctx || !t || !out) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (text_offset > t->cap || text_bytes > t->cap - text_offset) return fail(SKH_ECAPACITY, "text block larger than its buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    // the text goes to the start of the device buffer wherever it starts on the host; the bytes of the
    // last, partial 16-byte chunk beyond text_bytes are never looked at by the parser
    const uint64_t text_end = text_offset + text_bytes;
    if (t->up_bytes && t->up_off >= text_offset && t->up_off + t->up_bytes <= text_end) {
        // most of the text is on its way already (skh_text_upload): the bytes around it follow, and a
        // device copy moves the whole to the aligned start of the text buffer
        if (t->up_off > text_offset)
            HIP_TRY(hipMemcpyAsync(t->d_stage + text_offset, t->h_text + text_offset, t->up_off - text_offset,
                                   hipMemcpyHostToDevice, t->stream));
        const uint64_t up_end = t->up_off + t->up_bytes;
        if (text_end > up_end)
            HIP_TRY(hipMemcpyAsync(t->d_stage + up_end, t->h_text + up_end, text_end - up_end, hipMemcpyHostToDevice, t->stream));
        HIP_TRY(hipMemcpyAsync(t->d_text, t->d_stage + text_offset, text_bytes, hipMemcpyDeviceToDevice, t->stream));
    } else {
        uint64_t up = skh_padded_size(text_bytes);
        if (up > t->cap - text_offset) up = text_bytes;
        if (up) HIP_TRY(hipMemcpyAsync(t->d_text, t->h_text + text_offset, up, hipMemcpyHostToDevice, t->stream));
    }
    t->up_bytes = 0;
    t->text_off = text_offset;
    for (;;) {
        void* ws = t->d_ws;
        uint64_t ws_bytes = t->ws_bytes;
        if (format == SKH_FASTQ_WRAPPED) {
            if (!t->d_wws) {
                t->wws_bytes = skh::parse_wrapped_workspace_size(t->cap, t->max_lines, t->max_lines);
                hipError_t e = hipMalloc(&t->d_wws, t->wws_bytes);
                if (e != hipSuccess) { t->d_wws = nullptr; return hip_fail(e, "allocating the workspace of the wrapped-FASTQ parser"); }
            }
            ws = t->d_wws; ws_bytes = t->wws_bytes;
        }
        int rc = skh_parse_dev(ctx, t->d_text, text_bytes, format, final, t->d_seq, t->d_qual, t->cap, t->d_offs,
                               t->d_npos, t->d_nlen, t->max_lines, ws, ws_bytes, t->max_lines,
                               static_cast<skh_parse_result*>(t->d_res), t->stream);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(t->h_res, t->d_res, sizeof(skh_parse_result), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        *out = *static_cast<skh_parse_result*>(t->h_res);
        if (out->status != SKH_PARSE_ECAPACITY) break;
        // only the line / record tables can be too small here (the planes hold a whole block)
        if (t->max_lines > text_bytes + 1) return fail(SKH_EHIP, "parser reports a capacity error with full-size tables");
        hipError_t e = text_alloc_lines(t, t->max_lines * 4);
        if (e != hipSuccess) return hip_fail(e, "growing the line tables");
    }
    t->fastq = format != SKH_FASTA;
    t->nrec = out->status == SKH_PARSE_OK ? out->nrec : 0;
    t->seq_bytes = out->status == SKH_PARSE_OK ? out->seq_bytes : 0;
    t->text_bytes = text_bytes;
    t->transformed = false;
    t->v_offs = t->d_offs; t->v_npos = t->d_npos; t->v_nlen = t->d_nlen; t->v_text = t->d_text;
    t->filter_offs = nullptr;
    t->letters_ready = false;
    return SKH_OK;
}

int skh_text_upload(skh_ctx* ctx, skh_text* t, uint64_t offset, uint64_t nbytes)
{
    if (!ctx || !t) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (offset > t->cap || nbytes > t->cap - offset) return fail(SKH_ECAPACITY, "range beyond the block's buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    t->up_bytes = 0;
    if (nbytes == 0) return SKH_OK;
    if (!t->d_stage) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t->d_stage), t->cap));
    HIP_TRY(hipMemcpyAsync(t->d_stage + offset, t->h_text + offset, nbytes, hipMemcpyHostToDevice, t->stream));
    t->up_off = offset;
    t->up_bytes = nbytes;
    return SKH_OK;
}

int skh_text_offsets(skh_ctx* ctx, skh_text* t, uint64_t first, uint64_t count, uint64_t* offsets)
{
    if (!ctx || !t || !offsets) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (first + count > t->nrec + 1) return fail(SKH_EINVAL, "offset range beyond the block's records");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(offsets, t->v_offs + first, count * 8, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    return SKH_OK;
}

int skh_text_sequence(skh_ctx* ctx, skh_text* t, uint64_t first_byte, uint64_t count, uint8_t* dst)
{
    if (!ctx || !t || (!dst && count)) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (first_byte + count > t->seq_bytes) return fail(SKH_EINVAL, "byte range beyond the block's sequence");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, (t->transformed ? t->d_seq2 : t->d_seq) + first_byte, count, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    return SKH_OK;
}

int skh_text_revcomp(skh_ctx* ctx, skh_text* t, const uint8_t* lut, int do_reverse, int reverse_qual)
{
    if (!ctx || !t) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (t->transformed) return fail(SKH_EINVAL, "block was already transformed");
    if (t->nrec == 0 || (!lut && !do_reverse)) return SKH_OK;
    int rc = skh_revcomp_dev(ctx, t->d_seq, t->d_seq2, t->v_offs, t->nrec, t->seq_bytes, lut, do_reverse, t->stream);
    if (rc) return rc;
    if (t->fastq) {
        if (do_reverse && reverse_qual)
            rc = skh_revcomp_dev(ctx, t->d_qual, t->d_qual2, t->v_offs, t->nrec, t->seq_bytes, nullptr, 1, t->stream);
        else
            HIP_TRY(hipMemcpyAsync(t->d_qual2, t->d_qual, t->seq_bytes, hipMemcpyDeviceToDevice, t->stream));
        if (rc) return rc;
    }
    t->transformed = true;
    return SKH_OK;
}

int skh_text_output_wait(skh_ctx* ctx, skh_text* t, const uint8_t** out, uint64_t* out_bytes)
{
    if (!ctx || !t || !out || !out_bytes) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    *out = nullptr; *out_bytes = 0;
    if (t->out_pending == 0) return SKH_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(t->stream));
    *out = t->h_out;
    *out_bytes = t->out_pending;
    t->out_pending = 0;
    return SKH_OK;
}

int skh_text_format_seq(skh_ctx* ctx, skh_text* t, const skh_seq_format* fmt, const uint8_t** out, uint64_t* out_bytes)
{
    if (!out || !out_bytes) return fail(SKH_EINVAL, "NULL argument");
    const int rc = skh_text_format_seq_submit(ctx, t, fmt);
    return rc ? rc : skh_text_output_wait(ctx, t, out, out_bytes);
}

int skh_text_format_tab(skh_ctx* ctx, skh_text* t, const skh_tab_format* fmt, const uint8_t** out, uint64_t* out_bytes)
{
    if (!out || !out_bytes) return fail(SKH_EINVAL, "NULL argument");
    const int rc = skh_text_format_tab_submit(ctx, t, fmt);
    return rc ? rc : skh_text_output_wait(ctx, t, out, out_bytes);
}

int skh_text_format_seq_submit(skh_ctx* ctx, skh_text* t, const skh_seq_format* fmt)
{
    if (!ctx || !t || !fmt) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    t->out_pending = 0;
    if (t->nrec == 0) return SKH_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint8_t* seq = t->transformed ? t->d_seq2 : t->d_seq;
    const uint8_t* qual = t->fastq ? (t->transformed ? t->d_qual2 : t->d_qual) : nullptr;
    for (;;) {
        int rc = format_seq_impl(ctx, t->filter_offs, t->v_text, t->v_npos, t->v_nlen, seq, qual, t->v_offs, t->nrec, fmt, t->d_out,
                                 t->out_cap, t->d_fws, t->fws_bytes, static_cast<skh_format_result*>(t->d_res),
                                 t->stream);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(t->h_res, t->d_res, sizeof(skh_format_result), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        const skh_format_result r = *static_cast<skh_format_result*>(t->h_res);
        if (r.status == SKH_PARSE_OK) {
            HIP_TRY(hipMemcpyAsync(t->h_out, t->d_out, r.out_bytes, hipMemcpyDeviceToHost, t->stream));
            t->out_pending = r.out_bytes;
            return SKH_OK;
        }
        hipError_t e = text_alloc_out(t, r.out_bytes + r.out_bytes / 16 + 4096);
        if (e != hipSuccess) return hip_fail(e, "growing the output buffer");
    }
}

int skh_text_count(skh_ctx* ctx, skh_text* t, const uint8_t class_of[256], uint32_t nclass, uint32_t pass)
{
    if (!ctx || !t || !class_of) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (pass >= SKH_TAB_MAX_PASSES) return fail(SKH_EINVAL, "pass out of range");
    if (nclass < 1 || nclass > SKH_MAX_CLASSES) return fail(SKH_EINVAL, "nclass out of range");
    HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t need = (t->nrec ? t->nrec : 1) * nclass * 8;
    if (t->counts_cap[pass] < need) {
        (void)hipFree(t->d_counts[pass]);
        t->d_counts[pass] = nullptr; t->counts_cap[pass] = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t->d_counts[pass]), need + need / 4));
        t->counts_cap[pass] = need + need / 4;
    }
    t->nclass[pass] = nclass;
    if (t->nrec == 0) return SKH_OK;
    return skh_count_dev(ctx, t->transformed ? t->d_seq2 : t->d_seq, t->v_offs, t->nrec, t->seq_bytes, class_of, nclass,
                         t->d_counts[pass], t->stream);
}

int skh_text_format_tab_submit(skh_ctx* ctx, skh_text* t, const skh_tab_format* fmt)
{
    if (!ctx || !t || !fmt) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    t->out_pending = 0;
    if (t->nrec == 0) return SKH_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint8_t* seq = t->transformed ? t->d_seq2 : t->d_seq;
    const uint8_t* qual = t->fastq ? (t->transformed ? t->d_qual2 : t->d_qual) : nullptr;
    if (fmt->print_qual && !qual) return fail(SKH_EINVAL, "quality requested for a FASTA block");
    skh_tab_format with_letters;
    if (fmt->alphabet) {
        if (!t->letters_ready) return fail(SKH_EINVAL, "alphabet column requested before skh_text_letters");
        with_letters = *fmt;
        with_letters.d_letters = t->d_letters;
        fmt = &with_letters;
    }
    for (;;) {
        int rc = skh_format_tab_dev(ctx, t->v_text, t->v_npos, t->v_nlen, seq, qual, t->v_offs, t->nrec, t->d_counts,
                                    t->nclass, fmt, t->d_out, t->out_cap, t->d_fws, t->fws_bytes,
                                    static_cast<skh_format_result*>(t->d_res), t->stream);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(t->h_res, t->d_res, sizeof(skh_format_result), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        const skh_format_result r = *static_cast<skh_format_result*>(t->h_res);
        if (r.status == SKH_PARSE_OK) {
            HIP_TRY(hipMemcpyAsync(t->h_out, t->d_out, r.out_bytes, hipMemcpyDeviceToHost, t->stream));
            t->out_pending = r.out_bytes;
            return SKH_OK;
        }
        hipError_t e = text_alloc_out(t, r.out_bytes + r.out_bytes / 16 + 4096);
        if (e != hipSuccess) return hip_fail(e, "growing the output buffer");
    }
}

int skh_text_validate(skh_ctx* ctx, skh_text* t, const uint8_t valid[256], uint64_t limit, skh_validate_result* out)
{
    if (!ctx || !t || !valid || !out) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    memset(out, 0, sizeof *out);
    if (t->nrec == 0) return SKH_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = skh_validate_dev(ctx, t->transformed ? t->d_seq2 : t->d_seq, t->v_offs, t->nrec, t->seq_bytes, valid, limit,
                              static_cast<skh_validate_result*>(t->d_res), t->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(t->h_res, t->d_res, sizeof(skh_validate_result), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    memcpy(out, t->h_res, sizeof *out);
    return SKH_OK;
}

int skh_text_truncate(skh_ctx* ctx, skh_text* t, uint64_t nrec)
{
    if (!ctx || !t) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (nrec > t->nrec) return fail(SKH_EINVAL, "cannot truncate to more records than the block holds");
    if (nrec == t->nrec) return SKH_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(t->h_res, t->v_offs + nrec, 8, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    t->seq_bytes = *static_cast<uint64_t*>(t->h_res);
    t->nrec = nrec;
    return SKH_OK;
}

int skh_text_name(skh_ctx* ctx, skh_text* t, uint64_t record, uint64_t* pos, uint32_t* len)
{
    if (!ctx || !t || !pos || !len) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (record >= t->nrec) return fail(SKH_EINVAL, "record beyond the block");
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t* h = static_cast<uint8_t*>(t->h_res);
    HIP_TRY(hipMemcpyAsync(h, t->v_npos + record, 8, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipMemcpyAsync(h + 8, t->v_nlen + record, 4, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    memcpy(pos, h, 8);
    memcpy(len, h + 8, 4);
    if (t->v_text == t->d_text) *pos += t->text_off;       // positions in skh_text_buffer()
    return SKH_OK;
}

int skh_text_quality(skh_ctx* ctx, skh_text* t, uint64_t first_byte, uint64_t count, uint8_t* dst)
{
    if (!ctx || !t || (!dst && count)) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (!t->fastq) return fail(SKH_EINVAL, "a FASTA block has no qualities");
    if (first_byte + count > t->seq_bytes) return fail(SKH_EINVAL, "byte range beyond the block's qualities");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, (t->transformed ? t->d_qual2 : t->d_qual) + first_byte, count, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    return SKH_OK;
}

int skh_text_validate_quality(skh_ctx* ctx, skh_text* t, const uint8_t valid[256], skh_validate_result* out)
{
    if (!ctx || !t || !valid || !out) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    memset(out, 0, sizeof *out);
    if (!t->fastq) return fail(SKH_EINVAL, "a FASTA block has no qualities");
    if (t->nrec == 0) return SKH_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = skh_validate_dev(ctx, t->transformed ? t->d_qual2 : t->d_qual, t->v_offs, t->nrec, t->seq_bytes, valid, 0,
                              static_cast<skh_validate_result*>(t->d_res), t->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(t->h_res, t->d_res, sizeof(skh_validate_result), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    memcpy(out, t->h_res, sizeof *out);
    return SKH_OK;
}

int skh_text_map_quality(skh_ctx* ctx, skh_text* t, const uint8_t lut[256])
{
    if (!ctx || !t || !lut) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (!t->fastq) return fail(SKH_EINVAL, "a FASTA block has no qualities");
    if (t->transformed) return fail(SKH_EINVAL, "block was already transformed");
    if (t->nrec == 0) return SKH_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = skh_revcomp_dev(ctx, t->d_qual, t->d_qual2, t->v_offs, t->nrec, t->seq_bytes, lut, 0, t->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(t->d_seq2, t->d_seq, t->seq_bytes, hipMemcpyDeviceToDevice, t->stream));
    t->transformed = true;
    return SKH_OK;
}

int skh_text_names(skh_ctx* ctx, skh_text* t, uint64_t first, uint64_t count, uint64_t* pos, uint32_t* len)
{
    if (!ctx || !t || ((!pos || !len) && count)) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (first + count > t->nrec) return fail(SKH_EINVAL, "record range beyond the block");
    if (count == 0) return SKH_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(pos, t->v_npos + first, count * 8, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipMemcpyAsync(len, t->v_nlen + first, count * 4, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    if (t->v_text == t->d_text && t->text_off)             // positions in skh_text_buffer()
        for (uint64_t i = 0; i < count; i++) pos[i] += t->text_off;
    return SKH_OK;
}

int skh_text_set_names(skh_ctx* ctx, skh_text* t, uint64_t first, uint64_t count, const uint64_t* pos, const uint32_t* len)
{
    if (!ctx || !t || ((!pos || !len) && count)) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (first + count > t->nrec) return fail(SKH_EINVAL, "record range beyond the block");
    if (t->v_text != t->d_text) return fail(SKH_EINVAL, "the block's names no longer lie in its text buffer");
    if (count == 0) return SKH_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    // positions come in skh_text_buffer() coordinates; the device copy of the text starts at text_off
    std::vector<uint64_t> p(count);
    for (uint64_t i = 0; i < count; i++) {
        if (pos[i] < t->text_off || pos[i] + len[i] > t->text_off + t->text_bytes)
            return fail(SKH_EINVAL, "a name span lies outside the parsed text");
        p[i] = pos[i] - t->text_off;
    }
    HIP_TRY(hipMemcpyAsync(t->d_npos + first, p.data(), count * 8, hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipMemcpyAsync(t->d_nlen + first, len, count * 4, hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    return SKH_OK;
}

int skh_text_subseq(skh_ctx* ctx, skh_text* t, const skh_slice* slices, uint64_t n, int64_t start, int64_t end, int minus,
                    const uint8_t* lut, const uint8_t* names, uint64_t names_bytes, const uint64_t* name_pos,
                    const uint32_t* name_len)
{
    if (!ctx || !t) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (t->transformed) return fail(SKH_EINVAL, "block was already transformed");
    if (t->v_offs != t->d_offs)
        return fail(SKH_EINVAL, "gaps were removed from this block: cut subsequences before skh_text_remove_gaps");
    if (!slices) n = t->nrec;
    if (!names && slices) return fail(SKH_EINVAL, "a slice list needs names for its records");
    if (names && (!name_pos || !name_len)) return fail(SKH_EINVAL, "NULL name table");
    HIP_TRY(hipSetDevice(ctx->device));
    if (n + 1 > t->slice_cap) {
        (void)hipFree(t->d_slices); (void)hipFree(t->d_offs2); (void)hipFree(t->d_npos2); (void)hipFree(t->d_nlen2);
        (void)hipFree(t->d_sws);
        t->d_slices = nullptr; t->d_offs2 = t->d_npos2 = nullptr; t->d_nlen2 = nullptr; t->d_sws = nullptr; t->slice_cap = 0;
        const uint64_t cap = n + n / 4 + 1024;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t->d_slices), cap * sizeof(skh::Slice)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t->d_offs2), (cap + 1) * 8));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t->d_npos2), cap * 8));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t->d_nlen2), cap * 4));
        HIP_TRY(hipMalloc(&t->d_sws, skh::subseq_workspace_size(cap)));
        t->slice_cap = cap;
    }
    if (names && names_bytes + 16 > t->names_cap) {
        (void)hipFree(t->d_names); t->d_names = nullptr; t->names_cap = 0;
        const uint64_t cap = names_bytes + names_bytes / 4 + 4096;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t->d_names), cap));
        t->names_cap = cap;
    }
    if (slices && n) HIP_TRY(hipMemcpyAsync(t->d_slices, slices, n * sizeof(skh_slice), hipMemcpyHostToDevice, t->stream));
    if (names && n) {
        HIP_TRY(hipMemcpyAsync(t->d_names, names, names_bytes, hipMemcpyHostToDevice, t->stream));
        HIP_TRY(hipMemcpyAsync(t->d_npos2, name_pos, n * 8, hipMemcpyHostToDevice, t->stream));
        HIP_TRY(hipMemcpyAsync(t->d_nlen2, name_len, n * 4, hipMemcpyHostToDevice, t->stream));
    }
    for (;;) {
        int rc = skh_subseq_dev(ctx, t->d_seq, t->fastq ? t->d_qual : nullptr, t->d_offs, t->nrec, t->seq_bytes,
                                slices ? reinterpret_cast<const skh_slice*>(t->d_slices) : nullptr, n, start, end, minus, lut,
                                t->d_seq2, t->fastq ? t->d_qual2 : nullptr, t->plane_cap, t->d_offs2, t->d_sws,
                                skh::subseq_workspace_size(t->slice_cap), static_cast<skh_subseq_result*>(t->d_res),
                                t->stream);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(t->h_res, t->d_res, sizeof(skh_subseq_result), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        const skh_subseq_result r = *static_cast<skh_subseq_result*>(t->h_res);
        if (r.status == SKH_SUBSEQ_ERECORD) return fail(SKH_EINVAL, "a slice names a record beyond the block");
        if (r.status == SKH_SUBSEQ_OK) {
            t->seq_bytes = r.out_bytes;
            break;
        }
        // overlapping features can select more bases than the block holds: larger planes, again
        (void)hipFree(t->d_seq2); (void)hipFree(t->d_qual2);
        t->d_seq2 = t->d_qual2 = nullptr; t->plane_cap = 0;
        const uint64_t cap = skh_padded_size(r.out_bytes + r.out_bytes / 8 + 4096);
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t->d_seq2), cap));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t->d_qual2), cap));
        t->plane_cap = cap;
    }
    t->nrec = n;
    t->transformed = true;
    t->v_offs = t->d_offs2;
    if (names) { t->v_text = t->d_names; t->v_npos = t->d_npos2; t->v_nlen = t->d_nlen2; }
    // the per-record tables of the formatter are sized by the parser's line count
    if (n > t->fws_records) {
        (void)hipFree(t->d_fws); t->d_fws = nullptr;
        t->fws_records = n + n / 4;
        t->fws_bytes = skh::format_workspace_size(t->fws_records);
        HIP_TRY(hipMalloc(&t->d_fws, t->fws_bytes));
    }
    return SKH_OK;
}

int skh_text_letters(skh_ctx* ctx, skh_text* t)
{
    if (!ctx || !t) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t need = (t->nrec ? t->nrec : 1) * 8;
    if (need > t->letters_cap) {
        (void)hipFree(t->d_letters); t->d_letters = nullptr; t->letters_cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t->d_letters), (need + need / 4) * 4));
        t->letters_cap = need + need / 4;
    }
    t->letters_ready = true;
    if (t->nrec == 0) return SKH_OK;
    return skh_letters_dev(ctx, t->transformed ? t->d_seq2 : t->d_seq, t->v_offs, t->nrec, t->seq_bytes, t->d_letters, t->stream);
}

int skh_text_remove_gaps(skh_ctx* ctx, skh_text* t, const uint8_t is_gap[256])
{
    if (!ctx || !t || !is_gap) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (t->transformed) return fail(SKH_EINVAL, "block was already transformed");
    if (t->v_offs != t->d_offs) return fail(SKH_EINVAL, "gaps were already removed from this block");
    if (t->nrec == 0) return SKH_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (t->nrec + 1 > t->goffs_cap) {
        (void)hipFree(t->d_goffs); t->d_goffs = nullptr; t->goffs_cap = 0;
        const uint64_t cap = t->nrec + t->nrec / 4 + 1024;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t->d_goffs), cap * 8));
        t->goffs_cap = cap;
    }
    const uint64_t need = skh::remove_gaps_workspace_size(t->nrec, t->seq_bytes);
    if (need > t->gws_bytes) {
        (void)hipFree(t->d_gws); t->d_gws = nullptr; t->gws_bytes = 0;
        HIP_TRY(hipMalloc(&t->d_gws, need + need / 4));
        t->gws_bytes = need + need / 4;
    }
    int rc = skh_remove_gaps_dev(ctx, t->d_seq, t->fastq ? t->d_qual : nullptr, t->d_offs, t->nrec, t->seq_bytes, is_gap,
                                 t->d_seq2, t->fastq ? t->d_qual2 : nullptr, t->d_goffs, t->d_gws, t->gws_bytes,
                                 static_cast<uint64_t*>(t->d_res), t->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(t->h_res, t->d_res, 8, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    // the compacted planes become the block's planes (the two pairs of buffers change places: both
    // hold at least a block of text); the lengths before the removal stay available to the formatter,
    // whose length filter looks at the records as they were read (seq.go:129-134 come before :196)
    std::swap(t->d_seq, t->d_seq2);
    std::swap(t->d_qual, t->d_qual2);
    std::swap(t->base_cap, t->plane_cap);
    t->seq_bytes = *static_cast<uint64_t*>(t->h_res);
    t->v_offs = t->d_goffs;
    t->filter_offs = t->d_offs;
    return SKH_OK;
}

int skh_text_stats(skh_ctx* ctx, skh_text* t, const uint8_t is_gap[256], skh_stats_partial* inout)
{
    if (!ctx || !t || !is_gap || !inout) return fail(SKH_EINVAL, "NULL argument");
    if (t->ctx != ctx) return fail(SKH_EINVAL, "the text block belongs to another context");
    if (t->nrec == 0) return SKH_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = skh_stats_dev(ctx, t->transformed ? t->d_seq2 : t->d_seq, t->v_offs, t->nrec, t->seq_bytes, is_gap,
                           static_cast<skh_stats_partial*>(t->d_res), t->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(t->h_res, t->d_res, sizeof(skh_stats_partial), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    skh_stats_merge(inout, static_cast<skh_stats_partial*>(t->h_res));
    return SKH_OK;
}
