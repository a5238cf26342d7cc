// multi.hip -- several GPUs: sharding rules and the path's one collective.
//
// Records are independent, so a batch is cut into contiguous record ranges of about equal byte
// volume (skh_split_by_bytes), a record longer than a share is cut inside the record
// (skh_split_record: input slice [a, b) of a record of length L is a record of its own whose
// reverse-complement lands at [L-b, L-a)), and reverse-complement and per-record counts need no
// exchange.  The one collective is the element-wise reduction of the `stats` totals
// (seqkit/cmd/stat.go:106-125: sum of num / len_sum / gap_sum, min of len_min, max of len_max)
// over RCCL: three all-reduces of int64 in one group on a 40-byte buffer, latency-bound.
//
// RCCL is loaded when the first communicator is made (a process that already holds an RCCL, e.g.
// under torch.distributed, is given that one: its symbols are found in the global scope; otherwise
// the library is opened with dlopen), so single-GPU users of the library never load it.  RCCL may
// announce itself on standard output while a communicator is made; a caller whose standard output
// is its data points file descriptor 1 elsewhere around those calls (the driver does).
#include "seqkit_hip.h"

#include <dlfcn.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "ctx.h"

namespace {

struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    bool ok = false;
    bool global = false;       // the symbols came from a copy the process held already
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    // a copy the process already holds comes first (two RCCLs in one process do not share their state):
    // its symbols are looked up in the global scope; only when there is none is the library opened here
    const bool have_global = dlsym(RTLD_DEFAULT, "ncclAllReduce") != nullptr;
    void* h = nullptr;
    if (!have_global) {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return;
    }
    void* from = have_global ? RTLD_DEFAULT : h;
    Rccl r;
#define SKH_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(from, #sym))
    SKH_SYM(GetUniqueId, ncclGetUniqueId); SKH_SYM(CommInitRank, ncclCommInitRank); SKH_SYM(CommInitAll, ncclCommInitAll);
    SKH_SYM(CommDestroy, ncclCommDestroy); SKH_SYM(AllReduce, ncclAllReduce); SKH_SYM(GroupStart, ncclGroupStart);
    SKH_SYM(GroupEnd, ncclGroupEnd); SKH_SYM(GetErrorString, ncclGetErrorString); SKH_SYM(GetVersion, ncclGetVersion);
#undef SKH_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.AllReduce && r.GroupStart &&
           r.GroupEnd && r.GetErrorString;
    r.global = have_global;
    g_rccl = r;
}

int need_rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    if (g_rccl.ok) return SKH_OK;
    const char* why = dlerror();
    return skh::set_error(SKH_ENODEV, "RCCL (librccl.so) could not be loaded: %s", why ? why : "missing symbols");
}

int rccl_fail(ncclResult_t r, const char* what) { return skh::set_error(SKH_EHIP, "%s: %s", what, g_rccl.GetErrorString(r)); }
int hip_fail(hipError_t e, const char* what) { return skh::set_error(SKH_EHIP, "%s: %s", what, hipGetErrorString(e)); }

#define RCCL_TRY(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return rccl_fail(r_, #expr); } while (0)
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(e_, #expr); } while (0)

}  // namespace

// One rank of the communicator per DISTINCT device among the contexts it was made over; contexts
// that share a device (several contexts on one GPU) are merged on the host before the collective.
struct skh_comm {
    int nctx = 0;                       // contexts (entries of the host-form partial array)
    int nlocal = 0;                     // distinct devices = local ranks
    std::vector<skh_ctx*> ctx;          // nctx
    std::vector<int> rank_of_ctx;       // nctx -> local rank
    std::vector<int> device;            // nlocal
    std::vector<ncclComm_t> comm;       // nlocal
    std::vector<hipStream_t> stream;    // nlocal
    std::vector<skh_stats_partial*> d_buf, h_buf;   // nlocal: staging of the host form
};

extern "C" {

int skh_split_by_bytes(const uint64_t* offsets, uint64_t nrec, int nparts, uint64_t* cuts)
{
    if (!offsets || !cuts) return skh::set_error(SKH_EINVAL, "NULL argument");
    if (nparts < 1) return skh::set_error(SKH_EINVAL, "nparts must be at least 1");
    const uint64_t total = offsets[nrec];
    cuts[0] = 0;
    for (int k = 1; k < nparts; k++) {
        // first record boundary at or beyond the byte target, or the one before it when that is nearer
        const uint64_t target = (uint64_t)((unsigned __int128)total * (unsigned)k / (unsigned)nparts);
        uint64_t lo = 0, hi = nrec;     // first r with offsets[r] >= target
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (offsets[mid] >= target) hi = mid; else lo = mid + 1; }
        uint64_t r = lo;
        if (r > 0 && target - offsets[r - 1] <= offsets[r] - target) r--;
        if (r < cuts[k - 1]) r = cuts[k - 1];
        cuts[k] = r;
    }
    cuts[nparts] = nrec;
    return SKH_OK;
}

int skh_split_record(uint64_t length, int nparts, uint64_t* cuts)
{
    if (!cuts) return skh::set_error(SKH_EINVAL, "NULL argument");
    if (nparts < 1) return skh::set_error(SKH_EINVAL, "nparts must be at least 1");
    cuts[0] = 0;
    for (int k = 1; k < nparts; k++) {
        uint64_t c = (uint64_t)((unsigned __int128)length * (unsigned)k / (unsigned)nparts) & ~15ull;
        if (c < cuts[k - 1]) c = cuts[k - 1];
        cuts[k] = c;
    }
    cuts[nparts] = length;
    return SKH_OK;
}

int skh_comm_unique_id(uint8_t id[SKH_COMM_ID_BYTES])
{
    static_assert(SKH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id) return skh::set_error(SKH_EINVAL, "NULL argument");
    if (int rc = need_rccl()) return rc;
    ncclUniqueId u;
    RCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, SKH_COMM_ID_BYTES);
    return SKH_OK;
}

static int comm_buffers(skh_comm* c)
{
    for (int r = 0; r < c->nlocal; r++) {
        HIP_TRY(hipSetDevice(c->device[r]));
        HIP_TRY(hipStreamCreateWithFlags(&c->stream[r], hipStreamNonBlocking));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_buf[r]), sizeof(skh_stats_partial)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_buf[r]), sizeof(skh_stats_partial), hipHostMallocDefault));
    }
    return SKH_OK;
}

static skh_comm* comm_new(int nctx, int nlocal)
{
    skh_comm* c = new (std::nothrow) skh_comm;
    if (!c) return nullptr;
    c->nctx = nctx; c->nlocal = nlocal;
    c->ctx.assign(nctx, nullptr); c->rank_of_ctx.assign(nctx, 0);
    c->device.assign(nlocal, 0); c->comm.assign(nlocal, nullptr); c->stream.assign(nlocal, nullptr);
    c->d_buf.assign(nlocal, nullptr); c->h_buf.assign(nlocal, nullptr);
    return c;
}

int skh_comm_init_all(skh_ctx* const* ctxs, int n, skh_comm** out)
{
    if (!ctxs || !out || n < 1) return skh::set_error(SKH_EINVAL, "skh_comm_init_all: bad argument");
    *out = nullptr;
    for (int i = 0; i < n; i++) if (!ctxs[i]) return skh::set_error(SKH_EINVAL, "skh_comm_init_all: NULL context");
    if (int rc = need_rccl()) return rc;
    std::vector<int> devices, rank_of(n);
    for (int i = 0; i < n; i++) {
        int r = -1;
        for (size_t k = 0; k < devices.size(); k++) if (devices[k] == ctxs[i]->device) r = (int)k;
        if (r < 0) { r = (int)devices.size(); devices.push_back(ctxs[i]->device); }
        rank_of[i] = r;
    }
    skh_comm* c = comm_new(n, (int)devices.size());
    if (!c) return skh::set_error(SKH_ENOMEM, "out of host memory");
    for (int i = 0; i < n; i++) { c->ctx[i] = ctxs[i]; c->rank_of_ctx[i] = rank_of[i]; }
    c->device = devices;
    int rc = comm_buffers(c);
    if (rc == SKH_OK) {
        const ncclResult_t r = g_rccl.CommInitAll(c->comm.data(), c->nlocal, c->device.data());
        if (r != ncclSuccess) rc = rccl_fail(r, "ncclCommInitAll");
    }
    if (rc != SKH_OK) { skh_comm_destroy(c); return rc; }
    *out = c;
    return SKH_OK;
}

int skh_comm_init_rank(skh_ctx* ctx, const uint8_t id[SKH_COMM_ID_BYTES], int nranks, int rank, skh_comm** out)
{
    if (!ctx || !id || !out || nranks < 1 || rank < 0 || rank >= nranks)
        return skh::set_error(SKH_EINVAL, "skh_comm_init_rank: bad argument");
    *out = nullptr;
    if (int rc = need_rccl()) return rc;
    skh_comm* c = comm_new(1, 1);
    if (!c) return skh::set_error(SKH_ENOMEM, "out of host memory");
    c->ctx[0] = ctx; c->device[0] = ctx->device;
    int rc = comm_buffers(c);
    if (rc == SKH_OK) {
        ncclUniqueId u;
        memcpy(u.internal, id, SKH_COMM_ID_BYTES);
        const ncclResult_t r = g_rccl.CommInitRank(&c->comm[0], nranks, u, rank);     // the device is current (comm_buffers)
        if (r != ncclSuccess) rc = rccl_fail(r, "ncclCommInitRank");
    }
    if (rc != SKH_OK) { skh_comm_destroy(c); return rc; }
    *out = c;
    return SKH_OK;
}

void skh_comm_destroy(skh_comm* c)
{
    if (!c) return;
    for (int r = 0; r < c->nlocal; r++) {
        (void)hipSetDevice(c->device[r]);
        if (c->stream[r]) (void)hipStreamSynchronize(c->stream[r]);
        if (c->comm[r] && g_rccl.ok) (void)g_rccl.CommDestroy(c->comm[r]);
        if (c->stream[r]) (void)hipStreamDestroy(c->stream[r]);
        if (c->d_buf[r]) (void)hipFree(c->d_buf[r]);
        if (c->h_buf[r]) (void)hipHostFree(c->h_buf[r]);
    }
    delete c;
}

int skh_comm_local_ranks(const skh_comm* c) { return c ? c->nlocal : 0; }

int skh_comm_rccl_version(int* runtime_version, int* header_version, int* shared_with_process)
{
    if (int rc = need_rccl()) return rc;
    int v = 0;
    if (g_rccl.GetVersion && g_rccl.GetVersion(&v) != ncclSuccess) v = 0;
    if (runtime_version) *runtime_version = v;
    if (header_version) *header_version = NCCL_VERSION_CODE;
    if (shared_with_process) *shared_with_process = g_rccl.global ? 1 : 0;
    return SKH_OK;
}

// sum of {num, len_sum, gap_sum}, min of len_min, max of len_max, in place, over all ranks
static int allreduce_group(skh_comm* c, skh_stats_partial* const* d_partial, const hipStream_t* streams)
{
    RCCL_TRY(g_rccl.GroupStart());
    for (int r = 0; r < c->nlocal; r++) {
        int64_t* p = reinterpret_cast<int64_t*>(d_partial[r]);
        ncclResult_t e = g_rccl.AllReduce(p, p, 3, ncclInt64, ncclSum, c->comm[r], streams[r]);
        if (e == ncclSuccess) e = g_rccl.AllReduce(p + 3, p + 3, 1, ncclInt64, ncclMin, c->comm[r], streams[r]);
        if (e == ncclSuccess) e = g_rccl.AllReduce(p + 4, p + 4, 1, ncclInt64, ncclMax, c->comm[r], streams[r]);
        if (e != ncclSuccess) { (void)g_rccl.GroupEnd(); return rccl_fail(e, "ncclAllReduce"); }
    }
    RCCL_TRY(g_rccl.GroupEnd());
    return SKH_OK;
}

int skh_stats_allreduce_dev(skh_comm* c, skh_stats_partial* const* d_partials, void* const* streams)
{
    static_assert(sizeof(skh_stats_partial) == 5 * sizeof(int64_t), "five int64 fields, no padding");
    if (!c || !d_partials) return skh::set_error(SKH_EINVAL, "NULL argument");
    std::vector<hipStream_t> st(c->nlocal);
    for (int r = 0; r < c->nlocal; r++) {
        if (!d_partials[r]) return skh::set_error(SKH_EINVAL, "NULL partial");
        st[r] = (streams && streams[r]) ? (hipStream_t)streams[r] : c->stream[r];
    }
    return allreduce_group(c, d_partials, st.data());
}

int skh_stats_allreduce(skh_comm* c, skh_stats_partial* partials)
{
    if (!c || !partials) return skh::set_error(SKH_EINVAL, "NULL argument");
    // contexts on one device first (host), then the collective across devices / processes
    for (int r = 0; r < c->nlocal; r++) skh_stats_init(c->h_buf[r]);
    for (int i = 0; i < c->nctx; i++) skh_stats_merge(c->h_buf[c->rank_of_ctx[i]], &partials[i]);
    for (int r = 0; r < c->nlocal; r++) {
        HIP_TRY(hipSetDevice(c->device[r]));
        HIP_TRY(hipMemcpyAsync(c->d_buf[r], c->h_buf[r], sizeof(skh_stats_partial), hipMemcpyHostToDevice, c->stream[r]));
    }
    if (int rc = allreduce_group(c, c->d_buf.data(), c->stream.data())) return rc;
    for (int r = 0; r < c->nlocal; r++) {
        HIP_TRY(hipSetDevice(c->device[r]));
        HIP_TRY(hipMemcpyAsync(c->h_buf[r], c->d_buf[r], sizeof(skh_stats_partial), hipMemcpyDeviceToHost, c->stream[r]));
    }
    for (int r = 0; r < c->nlocal; r++) {
        HIP_TRY(hipSetDevice(c->device[r]));
        HIP_TRY(hipStreamSynchronize(c->stream[r]));
    }
    for (int i = 0; i < c->nctx; i++) partials[i] = *c->h_buf[c->rank_of_ctx[i]];
    return SKH_OK;
}

}  // extern "C"
