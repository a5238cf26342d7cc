// ctx.h -- the context object behind skh_ctx*, shared by the host-side sources of the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <unordered_map>

#include "kernels.h"

struct skh_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    skh::BatchHint* d_hints = nullptr;     // N_SLOTS launch hints, one per stream in use
    uint64_t* d_stat_slots = nullptr;      // N_SLOTS scratch areas of the stats kernel (partial gap sums)
    unsigned long long* d_static_scratch = nullptr;   // this GPU's copy of `scratch`
    std::mutex slot_mu;
    std::unordered_map<hipStream_t, uint32_t> slot_of;
    uint32_t max_grid = 0;
    uint32_t n_cu = 256;
    int force_logG = -1;
    int span_u = -1;         // -1 = off (the span kernel runs on request only)
    int fuse_fastq = 0;
    int use_hint = 1;
    int staged = -1;         // -1: by mean record length
    uint32_t staged_steps = 0;
    int inplace_run = -1;    // log2 of the tiles per workgroup run of k_reverse_inplace_long (-1: by the number of records)
    uint32_t short_steps = 0;
    uint32_t count_steps = 0;
    int count_u = 0;         // chunks per lane and round of the counting kernel (0: from the mean record length)
    uint32_t stats_steps = 0;
};

struct skh_batch {
    skh_ctx* ctx = nullptr;
    uint64_t max_bytes = 0, max_records = 0;
    bool with_qual = false;
    uint8_t *h_seq = nullptr, *h_qual = nullptr;
    uint64_t* h_offs = nullptr;
    uint8_t *d_seq = nullptr, *d_seq_out = nullptr, *d_qual = nullptr, *d_qual_out = nullptr;
    uint64_t* d_offs = nullptr;
    uint64_t *d_counts = nullptr, *h_counts = nullptr;
    skh::StatsPartial *d_stats = nullptr, *h_stats = nullptr;
    hipStream_t stream = nullptr;
    // completion of a submitted operation
    uint64_t* user_counts = nullptr;
    uint64_t n_counts = 0;
    bool pending = false;
};

namespace skh {
// sets the calling thread's skh_last_error() text and returns `code` (defined in capi.hip)
int set_error(int code, const char* fmt, const char* a = "", const char* b = "");
// launch parameters for a batch of nrec records and `total` bytes on this context (capi.hip)
LaunchCfg cfg_for(const skh_ctx* ctx, uint64_t nrec, uint64_t total);
// the launch-scratch slot of stream `st` on this context; the library's own streams give theirs back
// with slot_release before they are destroyed (capi.hip)
int slot_for(skh_ctx* ctx, hipStream_t st, uint32_t* slot);
void slot_release(skh_ctx* ctx, hipStream_t st);
}
