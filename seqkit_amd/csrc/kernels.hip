// kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the SeqKit hot path.
//
// Everything here is byte/integer streaming work bounded by HBM bandwidth; there is
// no contraction anywhere, so no MFMA.  The design rules are: 16-byte aligned
// global loads and stores, the 256-entry tables staged once per workgroup in LDS,
// enough independent loads in flight per lane to cover HBM latency, and no
// per-record launches or host round trips.
//
// Data layout (all kernels): `seq` is the CSR byte plane of a batch (records back
// to back, no separators), `offs[nrec+1]` its uint64 offsets.  Both planes are
// 16-byte aligned and allocated up to a multiple of 16 bytes, so every aligned
// 16-byte chunk that overlaps a record may be loaded whole.
//
// Work decomposition: records of at most LONG_T bytes are handled by the *short*
// kernels, where a group of G = 2^logG adjacent lanes owns one record (G is picked
// by the host from the mean record length; 150-base reads use G = 4, i.e. 16
// records per wavefront).  Longer records are left to the *tile* kernels, where a
// workgroup owns an aligned TILE-byte window of the plane and serves the (at most
// two) long records that overlap it, so a chromosome-length record is spread over
// thousands of workgroups.  The short kernel counts the long records it skipped in
// a device word; the tile kernel returns at once when that word is zero.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <type_traits>
#include <unordered_map>

#include "kernels.h"
#include "device_util.h"

namespace skh {

// ------------------------------------------------------------------ helpers

// Workgroups b and b+8 share an XCD (and its L2) under round-robin dispatch.  This
// bijection hands each of the 8 residue classes one contiguous range of work ids,
// so neighbouring records (which share boundary chunks and offset cache lines)
// meet in the same L2.  Speed only; any mapping is correct.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n)
{
    const uint32_t q = n >> 3, r = n & 7, x = b & 7, k = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// Offsets of the GW consecutive records a wavefront serves in one step of the short
// kernels, fetched by ONE load instruction with GW+2 active lanes (lane l holds
// offs[rec0 - 1 + l]) and handed to the lane groups by cross-lane reads.  Letting every lane
// load its record's two or three offsets instead costs the texture addresser as much as
// the sequence data of a 150-base read.  Indices outside [0, nrec] read as 0 / total, so
// records past the end look empty.  Call from wave-uniform control flow only.
struct WaveOffsets {
    uint64_t v0, v1;
    __device__ __forceinline__ static uint64_t at(const uint64_t* __restrict__ offs, uint64_t nrec,
                                                  uint64_t total, uint64_t rec0, uint32_t l)
    {
        if (rec0 == 0 && l == 0) return 0;
        const uint64_t idx = rec0 - 1 + l;
        return idx > nrec ? total : offs[idx];
    }
    __device__ __forceinline__ void load(const uint64_t* __restrict__ offs, uint64_t nrec, uint64_t total,
                                         uint64_t rec0, uint32_t GW, uint32_t lane)
    {
        v0 = 0;
        v1 = 0;
        if (lane <= GW + 1) v0 = at(offs, nrec, total, rec0, lane);
        if (GW == 64 && lane < 2) v1 = at(offs, nrec, total, rec0, 64 + lane);
    }
    // value number k (0 = offs[rec0-1]); GW is wave-uniform
    __device__ __forceinline__ uint64_t get(uint32_t k, uint32_t GW) const
    {
        const uint64_t a = __shfl(v0, (int)(k & 63));
        if (GW != 64) return a;
        const uint64_t b = __shfl(v1, (int)(k & 63));
        return k < 64 ? a : b;
    }
};

// ------------------------------------------------------------------ batch hint
//
// One tiny workgroup ahead of the kernels of a launch: zeroes the launch's long-record counter (what
// a memset node would do) and, while at it, samples up to 1024 evenly spaced records.  The host picks
// the lanes per record of the short kernels from the batch's MEAN length, which is all it knows; a few
// records of megabases among short reads inflate that mean and would leave most lanes of every group
// idle.  The mean of the sampled records of at most LONG_T bytes corrects it on the device.
__global__ __launch_bounds__(SKH_BLOCK) void k_prepare(const uint64_t* __restrict__ offs, uint64_t nrec,
                                                       BatchHint* __restrict__ hint)
{
    __shared__ unsigned long long s_sum[SKH_BLOCK / 64];
    __shared__ uint32_t s_n[SKH_BLOCK / 64], s_max[SKH_BLOCK / 64];
    constexpr uint32_t PER = 4;
    const uint64_t nsample = nrec < PER * SKH_BLOCK ? nrec : PER * SKH_BLOCK;
    unsigned long long sum = 0;
    uint32_t n = 0, longest = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint64_t i = threadIdx.x + (uint64_t)k * SKH_BLOCK;
        if (i < nsample) {
            const uint64_t r = (nrec / nsample) * i + (nrec % nsample) * i / nsample;   // = i * nrec / nsample, no overflow
            const uint64_t len = offs[r + 1] - offs[r];
            if (len <= SKH_LONG_T) { sum += len; n++; longest = (uint32_t)len > longest ? (uint32_t)len : longest; }
        }
    }
    for (uint32_t m = 1; m < 64; m <<= 1) {
        sum += __shfl_xor(sum, m);
        n += __shfl_xor(n, m);
        const uint32_t t = __shfl_xor(longest, m);
        longest = t > longest ? t : longest;
    }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = sum; s_n[threadIdx.x >> 6] = n; s_max[threadIdx.x >> 6] = longest; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SKH_BLOCK / 64; w++) { sum += s_sum[w]; n += s_n[w]; longest = s_max[w] > longest ? s_max[w] : longest; }
        hint->n_long = 0;
        hint->mean_short = n ? (uint32_t)(sum / n) : ~0u;
        hint->stage_check = 0;
        hint->max_short = longest;
    }
}

// log2 of the lanes per record a short kernel works with: the host's choice, lowered to what the
// sampled short records ask for (never raised: the grid was sized for the host's choice).
template <bool COUNT, int U = 3>
__device__ __forceinline__ int hinted_logG(int logG_host, uint32_t mean_short, int use_hint)
{
    if (!use_hint) return logG_host;
    const uint32_t mean = __builtin_amdgcn_readfirstlane(mean_short);
    if (mean == ~0u) return logG_host;
    const int lg = COUNT ? lanes_log2_count(mean, U) : lanes_log2_reverse(mean);
    return lg < logG_host ? lg : logG_host;
}

// ------------------------------------------------------------------ long records
//
// Records longer than LONG_T are served by workgroups that own one aligned RUN of the plane each
// (the grid covers the plane once, in address order).  A workgroup finds the record holding the first
// byte of its run with one 64-ary search, then walks the offsets forward 256 at a time through LDS
// (one load per thread) looking for long records: in a chromosome-scale batch the first slice settles
// it, among short reads a run of 128 KiB costs a handful of slices, and no window of the plane pays a
// search of its own.  f(r, s, e) is called by all threads together for every long record [s, e) that
// overlaps [t0, t1), in order.
constexpr uint64_t LONG_RUN = 4 * SKH_TILE;          // bytes of plane per workgroup of the long-record kernels

struct WalkLds {
    uint64_t off[SKH_BLOCK + 1];
    uint32_t first;
};

// a short kernel met record `rec` = [s, e), which it leaves to the long-record kernel of its launch
__device__ __forceinline__ void note_long(BatchHint* __restrict__ hint, uint64_t rec, uint64_t s, uint64_t e)
{
    const uint32_t k = atomicAdd(&hint->n_long, 1u);
    if (k < LONG_LIST) { hint->list[k].s = s; hint->list[k].e = e; hint->list[k].rec = rec; }
}

template <typename F>
__device__ __forceinline__ void for_each_long(const uint64_t* __restrict__ offs, uint64_t nrec, uint64_t t0,
                                              uint64_t t1, WalkLds& W, const BatchHint* __restrict__ hint, F&& f)
{
    const uint32_t n_listed = hint->n_long;              // the same in every thread (the short kernel has finished)
    if (n_listed <= LONG_LIST) {
        // every long record of the batch is in the launch's list: the ones that overlap the run are picked
        // out of it (a run of LONG_RUN bytes meets at most LONG_RUN / LONG_T + 2 of them), sorted by position
        constexpr uint32_t MAXM = 16;
        __syncthreads();
        if (threadIdx.x == 0) W.first = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_listed; i += SKH_BLOCK) {
            const LongRec lr = hint->list[i];
            if (lr.s < t1 && lr.e > t0) {
                const uint32_t k = atomicAdd(&W.first, 1u);
                if (k < MAXM) { W.off[3 * k] = lr.s; W.off[3 * k + 1] = lr.e; W.off[3 * k + 2] = lr.rec; }
            }
        }
        __syncthreads();
        const uint32_t m = W.first;
        if (m <= MAXM) {
            if (threadIdx.x == 0)
                for (uint32_t a = 1; a < m; a++)
                    for (uint32_t b = a; b > 0 && W.off[3 * b] < W.off[3 * b - 3]; b--)
                        for (uint32_t c = 0; c < 3; c++) { const uint64_t t = W.off[3 * b + c]; W.off[3 * b + c] = W.off[3 * b - 3 + c]; W.off[3 * b - 3 + c] = t; }
            __syncthreads();
            for (uint32_t k = 0; k < m; k++) f(W.off[3 * k + 2], W.off[3 * k], W.off[3 * k + 1]);
            return;
        }
        // (more matches than a run can meet: the list holds a record more than once -- walk instead)
    }
    uint64_t r = find_record(offs, nrec, t0);            // the same in every wavefront
    while (true) {
        __syncthreads();                                  // the slice of the iteration before is done with
        {
            const uint64_t idx = r + threadIdx.x;
            W.off[threadIdx.x] = idx <= nrec ? offs[idx] : ~0ull;
            if (threadIdx.x == 0) {
                const uint64_t idx2 = r + SKH_BLOCK;
                W.off[SKH_BLOCK] = idx2 <= nrec ? offs[idx2] : ~0ull;
            }
        }
        __syncthreads();
        const uint64_t a = W.off[threadIdx.x], b = W.off[threadIdx.x + 1];
        const bool is_long = b != ~0ull && a < t1 && b - a > SKH_LONG_T;
        uint32_t from = 0;
        bool done = false;
        while (true) {
            if (threadIdx.x == 0) W.first = SKH_BLOCK;
            __syncthreads();
            const unsigned long long m = __ballot(is_long && threadIdx.x >= from);
            if (m != 0 && (threadIdx.x & 63) == 0)
                atomicMin(&W.first, (threadIdx.x & ~63u) + (uint32_t)__builtin_ctzll(m));
            __syncthreads();
            const uint32_t i = W.first;
            if (i == SKH_BLOCK) break;
            const uint64_t s = W.off[i], e = W.off[i + 1];
            f(r + i, s, e);
            if (e >= t1) { done = true; break; }
            from = i + 1;
            __syncthreads();                              // W.first is rewritten
        }
        if (done) return;
        const uint64_t last = W.off[SKH_BLOCK];
        if (last == ~0ull || last >= t1) return;          // the slice reached the end of the batch or of the run
        r += SKH_BLOCK;
    }
}

// ------------------------------------------------------------------ reverse (+complement)
//
// out[p] = lut[in[s + e - 1 - p]] for every byte p of a record [s, e).
//
// Ownership rule: every aligned 16-byte chunk of the output plane is written once,
// with one 16-byte store, by the lane that serves the record holding the chunk's
// LAST byte.  A record therefore writes the chunks that end inside it; its first
// chunk, when the record starts in the middle of it, also carries the tail bytes
// of the previous record(s), which the same lane fetches from their mirrored
// positions.  Nothing is ever written bytewise except the one chunk that straddles
// the end of the plane.
//
// The 16 source bytes of an output chunk at plane offset o are the window
// [q, q+16), q = s + e - 16 - o, read with ONE unaligned 16-byte load (gfx950
// global loads take any byte address) and byte-reversed with four v_perm_b32.

struct RevLds {
    uint8_t lut[256];
    uint4 low[17];       // low[k]: 16-byte mask with the k lowest bytes set
};

template <bool COMP>
__device__ __forceinline__ void rev_lds_init(RevLds& L, const Table256& lut)
{
    const uint32_t b = threadIdx.x;
    if (COMP) L.lut[b] = lut.v[b];
    if (b < 17) {
        uint32_t w[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int k = (int)b - 4 * m;
            w[m] = k <= 0 ? 0u : (k >= 4 ? 0xffffffffu : (1u << (8 * k)) - 1u);
        }
        L.low[b] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
}

__device__ __forceinline__ uint4 select16(uint4 mask, uint4 a, uint4 b)    // mask ? a : b, bitwise
{
    return make_uint4((a.x & mask.x) | (b.x & ~mask.x), (a.y & mask.y) | (b.y & ~mask.y),
                      (a.z & mask.z) | (b.z & ~mask.z), (a.w & mask.w) | (b.w & ~mask.w));
}

// Any chunk, any mix of records (tiny records, plane ends): byte p of the chunk is
// looked up by walking the offsets down from `rec`, a record at or after the one
// holding the chunk's last byte.  Slow, and only used where the fast forms do not apply.
template <bool COMP, bool FASTQ>
__device__ __forceinline__ void reverse_chunk_bytes(const uint8_t* __restrict__ in, const uint8_t* __restrict__ qin,
                                                    const uint64_t* __restrict__ offs, uint64_t rec, uint64_t total,
                                                    uint64_t o, const uint8_t* s_lut, uint4& v16, uint4& vq16)
{
    uint32_t w[4] = {0, 0, 0, 0}, wq[4] = {0, 0, 0, 0};
    uint64_t r = rec;
    const int n = total - o < 16 ? (int)(total - o) : 16;
#pragma unroll 1
    for (int k = n - 1; k >= 0; k--) {
        const uint64_t p = o + (uint64_t)k;
        while (offs[r] > p) r--;
        const uint64_t src = offs[r] + offs[r + 1] - 1 - p;
        uint32_t b = in[src];
        if (COMP) b = s_lut[b];
        // branch-free: the loaded byte is consumed on every path, so no load is ever left
        // pending (as the compiler sees it) when this function is left
        const uint32_t sh = 8 * (k & 3), m = (uint32_t)k >> 2;
        const uint32_t v = b << sh;
        w[0] |= m == 0 ? v : 0u; w[1] |= m == 1 ? v : 0u; w[2] |= m == 2 ? v : 0u; w[3] |= m == 3 ? v : 0u;
        if (FASTQ) {
            const uint32_t vq = (uint32_t)qin[src] << sh;
            wq[0] |= m == 0 ? vq : 0u; wq[1] |= m == 1 ? vq : 0u; wq[2] |= m == 2 ? vq : 0u; wq[3] |= m == 3 ? vq : 0u;
        }
    }
    v16 = make_uint4(w[0], w[1], w[2], w[3]);
    vq16 = make_uint4(wq[0], wq[1], wq[2], wq[3]);
}

template <bool COMP, bool FASTQ>
__device__ __forceinline__ void reverse_chunk_bytewise(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                       const uint8_t* __restrict__ qin, uint8_t* __restrict__ qout,
                                                       const uint64_t* __restrict__ offs, uint64_t rec,
                                                       uint64_t total, uint64_t o, const uint8_t* s_lut)
{
    uint4 v, vq;
    reverse_chunk_bytes<COMP, FASTQ>(in, qin, offs, rec, total, o, s_lut, v, vq);
    store_clipped(out, 0, total, o, v);
    if (FASTQ) store_clipped(qout, 0, total, o, vq);
}

// The same as a real call: for kernels that reach it from several places, so that its code
// and registers are paid for once (a call parks the caller's live values in the sparse
// callee-saved registers, so kernels with one call site inline the body instead).
template <bool COMP, bool FASTQ>
__device__ __noinline__ void reverse_chunk_generic(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                   const uint8_t* __restrict__ qin, uint8_t* __restrict__ qout,
                                                   const uint64_t* __restrict__ offs, uint64_t rec,
                                                   uint64_t total, uint64_t o, const uint8_t* s_lut)
{
    reverse_chunk_bytewise<COMP, FASTQ>(in, out, qin, qout, offs, rec, total, o, s_lut);
}

struct RevPlanes {
    const uint8_t* __restrict__ in;
    uint8_t* __restrict__ out;
    const uint8_t* __restrict__ qin;
    uint8_t* __restrict__ qout;
    const uint64_t* __restrict__ offs;
    uint64_t total;
};

// The window [q, q+16) of the 32 bytes lo‖hi (two adjacent aligned chunks, window offset
// dq = q mod 16), byte-reversed: a dword barrel shift and four v_perm_b32.
__device__ __forceinline__ uint4 window_reversed(uint4 lo, uint4 hi, uint32_t dq)
{
    const bool h = (dq & 8u) != 0, l = (dq & 4u) != 0;
    const uint32_t T0 = h ? lo.z : lo.x, T1 = h ? lo.w : lo.y, T2 = h ? hi.x : lo.z,
                   T3 = h ? hi.y : lo.w, T4 = h ? hi.z : hi.x, T5 = h ? hi.w : hi.y;
    const uint32_t V0 = l ? T1 : T0, V1 = l ? T2 : T1, V2 = l ? T3 : T2,
                   V3 = l ? T4 : T3, V4 = l ? T5 : T4;
    const uint32_t sel = 0x00010203u + (dq & 3u) * 0x01010101u;
    return make_uint4(__builtin_amdgcn_perm(V4, V3, sel), __builtin_amdgcn_perm(V3, V2, sel),
                      __builtin_amdgcn_perm(V2, V1, sel), __builtin_amdgcn_perm(V1, V0, sel));
}

__device__ __forceinline__ uint4 shfl16(uint4 v, int src_lane)
{
    return make_uint4(__shfl(v.x, src_lane), __shfl(v.y, src_lane), __shfl(v.z, src_lane),
                      __shfl(v.w, src_lane));
}

// Up to U chunks of one record, chunk indices k = k_first + u * k_step (relative to the
// record's first chunk o0 = s & ~15) below k_end.
//
// Memory instructions are what this kernel is short of (the texture addresser spends
// one cycle per lane on an aligned 16-byte access and two on a misaligned one), so the
// source window of a chunk is not read directly: every lane loads the ALIGNED chunk
// holding the start of its window, and takes the next aligned chunk -- the one its
// window spills into -- from the lane that serves the previous chunk of the record,
// whose own aligned chunk it is.  Only a chunk without such a neighbour (the first of a
// record, or lane 0 of a wavefront in ROW order) loads its second chunk itself.
// ROW = false: lanes of a group are k_step = G apart (chunk k-1 sits in lane-1, or in
// lane+G-1 one step of u earlier); ROW = true: k_step = 256, chunk k-1 is lane-1 of the
// same wavefront.  All loads are issued before the first use.
template <bool COMP, bool FASTQ, int U, bool ROW>
__device__ __forceinline__ void reverse_batch(const RevPlanes& P, const RevLds& L, uint64_t rec,
                                              uint64_t sprev, uint64_t s, uint64_t e, uint32_t k_first,
                                              uint32_t k_step, uint32_t k_end)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t o0 = s & ~15ull;
    const uint32_t sp = (uint32_t)(s - o0);                 // bytes of chunk 0 before the record
    const uint64_t q_base = s + e - 16 - o0;                // window of chunk 0
    const uint32_t dq = (uint32_t)q_base & 15u;             // the same for every chunk of the record
    const uint64_t a_base = q_base - dq;
    const uint64_t padded = (P.total + 15) & ~15ull;
    constexpr int NE = ROW ? U : 1;
    uint4 W[U], Q[U], E[NE], EQ[NE];
    bool own_upper[NE];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t k = k_first + u * k_step;
        W[u] = make_uint4(0, 0, 0, 0);
        if (FASTQ) Q[u] = make_uint4(0, 0, 0, 0);
        if (k < k_end) {
            const uint64_t a = a_base - 16ull * k;
            W[u] = ld16(P.in + a);
            if (FASTQ) Q[u] = ld16(P.qin + a);
        }
        if (u < NE) {
            // group order: only the group's first lane in its first step lacks a neighbour
            own_upper[u] = k < k_end && dq != 0 && (ROW ? lane == 0 : (u == 0 && k_first % k_step == 0));
            E[u] = make_uint4(0, 0, 0, 0);
            EQ[u] = make_uint4(0, 0, 0, 0);
            if (own_upper[u]) {
                const uint64_t a = a_base - 16ull * k + 16;
                if (a + 16 <= padded) {                      // beyond the plane only bytes nobody needs
                    E[u] = ld16(P.in + a);
                    if (FASTQ) EQ[u] = ld16(P.qin + a);
                }
            }
        }
    }
    // chunk 0 of a record that starts inside it: the previous record supplies bytes [0, sp)
    const bool head = k_first == 0 && k_end != 0 && sp != 0;
    const bool head_fast = head && sprev <= o0 && sprev + sp >= 16;
    uint4 HP = make_uint4(0, 0, 0, 0), HQ = make_uint4(0, 0, 0, 0);
    if (head_fast) {
        const uint64_t qp = sprev + s - 16 - o0;            // window of chunk o0 in the previous record
        HP = ldu16(P.in + qp);                              // one lane per record: misaligned is cheap here
        if (FASTQ) HQ = ldu16(P.qin + qp);
    }
    // Every lane runs the arithmetic below (the cross-lane reads must not sit in a
    // lane-divergent branch); lanes without a chunk compute on stale registers and skip the store.
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t k = k_first + u * k_step;
        const uint64_t o = o0 + 16ull * k;
        uint4 n = shfl16(W[u], (int)lane - 1), nq;
        if (FASTQ) nq = shfl16(Q[u], (int)lane - 1);
        if (!ROW && u > 0) {
            const bool first_lane = (k_first % k_step) == 0;
            const uint4 t = shfl16(W[u - 1], (int)(lane + k_step - 1));
            if (first_lane) n = t;
            if (FASTQ) {
                const uint4 tq = shfl16(Q[u - 1], (int)(lane + k_step - 1));
                if (first_lane) nq = tq;
            }
        }
        if (u < NE) {
            if (own_upper[u]) n = E[u];
            if (FASTQ && own_upper[u]) nq = EQ[u];
        }
        uint4 r = window_reversed(W[u], n, dq), rq;
        if (FASTQ) rq = window_reversed(Q[u], nq, dq);
        if (u == 0 && head_fast) {
            const uint4 m = L.low[sp];
            r = select16(m, reverse16(HP), r);
            if (FASTQ) rq = select16(m, reverse16(HQ), rq);
        }
        if (COMP) r = map16(r, L.lut);
        if (k < k_end) {
            if (u == 0 && head && !head_fast) {
                reverse_chunk_generic<COMP, FASTQ>(P.in, P.out, P.qin, P.qout, P.offs, rec, P.total, o, L.lut);
            } else {
                *reinterpret_cast<uint4*>(P.out + o) = r;
                if (FASTQ) *reinterpret_cast<uint4*>(P.qout + o) = rq;
            }
        }
    }
}

// Short records: a group of 2^logG lanes per record, lane j takes chunks j, j+G, ...
// Nothing crosses lanes, so any G gives the same bytes.  The grid covers the batch once, in index
// order: workgroup b serves the `steps` * 4 consecutive wavefront steps from b * steps * 4 on (with
// fewer lanes per record than the host sized the grid for, a step holds more records and the last
// workgroups find nothing left to do).
template <bool COMP, bool FASTQ>
__global__ __launch_bounds__(SKH_BLOCK) void k_reverse_short(RevPlanes P, uint64_t nrec, Table256 lut, int logG_host,
                                                             uint32_t steps, BatchHint* __restrict__ hint, int use_hint)
{
    __shared__ RevLds L;
    const uint32_t mean_short = hint->mean_short;
    rev_lds_init<COMP>(L, lut);
    constexpr int U = 3;
    const int logG = hinted_logG<false>(logG_host, mean_short, use_hint);
    const uint32_t G = 1u << logG, j = threadIdx.x & (G - 1);
    const uint32_t lane = threadIdx.x & 63, GW = 64u >> logG, g = lane >> logG;
    constexpr uint32_t WAVES = SKH_BLOCK / 64;
    const uint64_t stride = (uint64_t)WAVES * GW;
    // the wavefront's first record; all its lanes run the same number of steps
    uint64_t rec0 = ((uint64_t)blockIdx.x * steps * WAVES + (threadIdx.x >> 6)) * GW;
    const uint64_t rec_stop = (uint64_t)(blockIdx.x + 1) * steps * WAVES * GW;
    const uint64_t rec_end = rec_stop < nrec ? rec_stop : nrec;      // first record not served here
    if (rec0 >= rec_end) return;
    WaveOffsets wo;
    wo.load(P.offs, nrec, P.total, rec0, GW, lane);
    while (true) {
        // the offsets of the next step are requested before this one is worked on
        const uint64_t next0 = rec0 + stride;
        WaveOffsets nwo;
        if (next0 < rec_end) nwo.load(P.offs, nrec, P.total, next0, GW, lane);
        const uint64_t rec = rec0 + g;
        const uint64_t sprev = wo.get(g, GW), s = wo.get(g + 1, GW), e = wo.get(g + 2, GW);
        const uint64_t len = rec < rec_end ? e - s : 0;
        if (len > SKH_LONG_T) {
            if (j == 0) note_long(hint, rec, s, e);
        } else if (len != 0) {
            const uint64_t o0 = s & ~15ull;
            const uint32_t n_own = (uint32_t)((e - o0) >> 4);       // chunks that end inside the record
            for (uint32_t kb = 0; kb < n_own; kb += U * G)
                reverse_batch<COMP, FASTQ, U, false>(P, L, rec, sprev, s, e, kb + j, G, n_own);
            if (e == P.total && (e & 15) != 0 && j == 0)             // the chunk straddling the plane end
                reverse_chunk_generic<COMP, FASTQ>(P.in, P.out, P.qin, P.qout, P.offs, rec, P.total,
                                                   o0 + 16ull * n_own, L.lut);
        }
        if (next0 >= rec_end) break;
        rec0 = next0;
        wo = nwo;
    }
}

// ---- in place --------------------------------------------------------------------------
//
// out == in: a byte may only be overwritten by a lane that has already read everything the
// new bytes are made of.  A record [s, e) is taken as pairs of 16-byte windows at the same
// distance from its two ends, [s + 16 i, +16) and [e - 16 i - 16, +16): one lane loads both and
// stores each, reversed, in the other's place; it shares no byte with any other lane, whatever
// the alignment of s and e (gfx950 global loads and stores take any byte address).  What is
// left between the innermost pair (fewer than 32 bytes) goes to the lane of that pair, as
// windows that overlap bytes the lane has loaded and not yet stored: two windows when at least
// 16 bytes are left, otherwise one window that reaches back into the innermost pair.
// Overlapping stores carry the same, final bytes, so their order does not matter.  Records
// shorter than 16 bytes are swapped byte by byte.

__device__ __forceinline__ void stu16(uint8_t* p, uint4 v)      // any byte address
{
    __builtin_memcpy(p, &v, 16);
}

template <bool COMP>
__device__ __forceinline__ uint4 reversed_mapped(uint4 v, const uint8_t* s_lut)
{
    v = reverse16(v);
    return COMP ? map16(v, s_lut) : v;
}

// The middle [m0, m0 + M) of a record, 0 < M < 32, between windows the caller has loaded but not
// stored yet (or, with M >= 16, on its own).  load() before the caller's stores, store() after.
template <bool COMP>
struct MiddleInPlace {
    uint4 x, y;
    __device__ __forceinline__ void load(const uint8_t* p, uint64_t m0, uint32_t M)
    {
        x = ldu16(p + m0);                            // M < 16: reaches into the back window of the innermost pair
        if (M >= 16) y = ldu16(p + m0 + M - 16);
    }
    __device__ __forceinline__ void store(uint8_t* p, uint64_t m0, uint32_t M, const uint8_t* s_lut) const
    {
        // [m0 + M - 16, m0 + M) is the mirror image of [m0, m0 + 16)
        stu16(p + m0 + M - 16, reversed_mapped<COMP>(x, s_lut));
        if (M >= 16) stu16(p + m0, reversed_mapped<COMP>(y, s_lut));
    }
};

// [a, b) reversed byte by byte (a handful of bytes)
template <bool COMP>
__device__ __forceinline__ void reverse_bytes_inplace(uint8_t* p, uint64_t a, uint64_t b, const uint8_t* s_lut)
{
    while (b - a >= 2) {
        b--;
        const uint8_t x = p[a], y = p[b];
        p[a] = COMP ? s_lut[y] : y;
        p[b] = COMP ? s_lut[x] : x;
        a++;
    }
    if (COMP && b - a == 1) p[a] = s_lut[p[a]];
}

// One record of at most LONG_T bytes by its group of G lanes: lane j takes the window pairs
// j, j + G, ...
template <bool COMP>
__device__ __forceinline__ void reverse_record_inplace(uint8_t* p, uint64_t s, uint64_t e, uint32_t j, uint32_t G,
                                                       const uint8_t* s_lut)
{
    const uint64_t len = e - s;
    if (len >= 16) {
        const uint32_t n = (uint32_t)len >> 5, M = (uint32_t)len & 31u;       // pairs, bytes between them
        const uint64_t m0 = s + 16ull * n;
        if (n == 0) {
            if (j == 0) {
                MiddleInPlace<COMP> mid;
                mid.load(p, m0, M);
                mid.store(p, m0, M, s_lut);
            }
        } else {
            for (uint32_t i = j; i < n; i += G) {
                const uint64_t f = s + 16ull * i, b = e - 16ull * i - 16;
                const bool inner = i == n - 1 && M != 0;
                const uint4 F = ldu16(p + f), B = ldu16(p + b);
                MiddleInPlace<COMP> mid;
                if (inner) mid.load(p, m0, M);
                stu16(p + f, reversed_mapped<COMP>(B, s_lut));
                stu16(p + b, reversed_mapped<COMP>(F, s_lut));
                if (inner) mid.store(p, m0, M, s_lut);
            }
        }
    } else if (len != 0 && j == 0) {
        reverse_bytes_inplace<COMP>(p, s, e, s_lut);
    }
}

// ---- short records, staged form -------------------------------------------------------------
//
// The GW consecutive records a wavefront serves in one step are one contiguous stretch of the
// plane.  That stretch (plus the tail of the record before it, whose first bytes end up in the
// first chunk of the step) and the step's slice of the offsets array are copied into LDS by
// direct global->LDS loads: whole-wavefront, aligned, 1 KiB per instruction -- the fewest
// memory instructions this work can be done with, which is what the texture addresser charges
// for.  Every window (including the previous-record window of a head chunk) and every offset is
// then read from LDS; no lane touches global memory again until its store.
//
// Each wavefront cycles through ST_NB private stage buffers: while one step is reversed, looked
// up and stored, the loads of the next ST_NB-1 steps are in flight.  The loads are issued from
// inline assembly and awaited by hand: memory operations of a wavefront complete in issue
// order, every step issues exactly ST_LOADS loads (a step past the end of the batch re-reads
// the wavefront's first step), so "at most (ST_NB-1)*ST_LOADS operations outstanding" means
// the current step's data has landed; stores issued in between only make that wait stricter.
// (Left to the compiler, every LDS read that follows a global->LDS load is preceded by a wait
// for ALL outstanding memory operations, which serialises the pipeline.)
constexpr int ST_NB = 2;                            // stage buffers per wavefront
constexpr uint32_t ST_ROWS = 3;                     // 1-KiB load rows per stage
constexpr uint32_t ST_BYTES = ST_ROWS * 1024;       // bytes staged per wavefront step
constexpr uint32_t ST_PAD = 32;                     // a window's second chunk may be read past the stretch
constexpr uint32_t ST_OFFS = 72;                    // offsets per step: GW + 2 <= 66, loaded as 16-byte pairs
constexpr int ST_WAVES = SKH_BLOCK / 64;
constexpr int ST_LOADS = ST_ROWS + 1;               // load instructions per step

struct StageLds {
    RevLds tab;
    __attribute__((aligned(16))) uint8_t bytes[ST_WAVES][ST_NB][ST_BYTES + ST_PAD];
    __attribute__((aligned(16))) uint64_t offs[ST_WAVES][ST_NB][ST_OFFS];
};

// 16 bytes per lane from global memory straight into LDS at lds_addr + 16 * lane (lds_addr is
// wave-uniform).  Not visible to the compiler's wait-count bookkeeping: pair with wait_vmcnt.
// M0 (the LDS base of such loads) is a reserved register the compiler sets right before each
// of its own uses, so overwriting it here needs no clobber.
__device__ __forceinline__ void load_to_lds16(const void* g, uint32_t lds_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ uint32_t lds_addr_of(const void* p)
{
    return (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) uint8_t*)p);
}

// plane range [a0, a0 + n) a step must have in LDS, from the offsets just outside its records
__device__ __forceinline__ void stretch_range(uint64_t first, uint64_t last, uint64_t padded, uint64_t& a0,
                                              uint64_t& n)
{
    const uint64_t lo = first & ~15ull;
    a0 = lo >= 16 ? lo - 16 : 0;
    uint64_t a1 = ((last + 15) & ~15ull) + 16;
    if (a1 > padded) a1 = padded;
    n = a1 - a0;                                     // >= 16 whenever the plane is not empty
}

// The window starting at LDS offset lq of the slot, byte-reversed.  LDS reads need only dword
// alignment: the five dwords that hold the window are read as they lie (no dword selection
// network as for the 16-byte aligned global loads), and four v_perm_b32 shift and reverse.
__device__ __forceinline__ uint4 lds_window_reversed(const uint8_t* slot, uint32_t lq)
{
    const uint32_t* d = reinterpret_cast<const uint32_t*>(slot + (lq & ~3u));
    const uint32_t V0 = d[0], V1 = d[1], V2 = d[2], V3 = d[3], V4 = d[4];
    const uint32_t sel = 0x00010203u + (lq & 3u) * 0x01010101u;
    return make_uint4(__builtin_amdgcn_perm(V4, V3, sel), __builtin_amdgcn_perm(V3, V2, sel),
                      __builtin_amdgcn_perm(V2, V1, sel), __builtin_amdgcn_perm(V1, V0, sel));
}

template <bool COMP, bool INPLACE>
__device__ __forceinline__ void reverse_staged(StageLds& S, const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                               const uint64_t* __restrict__ offs, uint64_t total, uint64_t nrec,
                                               const Table256& lut, int logG_host, uint32_t steps,
                                               BatchHint* __restrict__ hint, int use_hint)
{
    const uint32_t mean_short = hint->mean_short;        // asked for before the table set-up, needed after it
    rev_lds_init<COMP>(S.tab, lut);
    const int logG = hinted_logG<false>(logG_host, mean_short, use_hint);
    const uint32_t G = 1u << logG, j = threadIdx.x & (G - 1);
    const uint32_t lane = threadIdx.x & 63, GW = 64u >> logG, g = lane >> logG;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t padded = (total + 15) & ~15ull;
    // Workgroup b serves the `steps` * ST_WAVES consecutive steps from b * steps * ST_WAVES on,
    // its wavefronts interleaved step by step.  (With fewer lanes per record than the host sized
    // the grid for, a step holds more records and the last workgroups of the grid find nothing
    // left to do.)  The grid covers the batch once: workgroups
    // start in index order, so the chip sweeps the planes front to back in one narrow window,
    // which the memory system rewards (a grid-stride loop over resident workgroups keeps
    // thousands of distant streams open and is ~10% slower on the same work).
    const uint64_t stride = (uint64_t)ST_WAVES * GW;
    const uint64_t rec_first = ((uint64_t)blockIdx.x * steps * ST_WAVES + wave) * GW;
    const uint64_t rec_stop = (uint64_t)(blockIdx.x + 1) * steps * ST_WAVES * GW;
    const uint64_t rec_end = rec_stop < nrec ? rec_stop : nrec;      // first record not served here
    if (rec_first >= rec_end) return;                // before anything is in flight

    // The offsets never change while the kernel runs: read through the constant address space,
    // wave-uniform reads of them are scalar loads, which cost the vector memory path nothing
    // and do not queue up with the loads below.
    const auto* coffs = (const __attribute__((address_space(4))) uint64_t*)offs;

    // the offsets just outside the records of the step starting at record r0 (scalar loads:
    // asked for one iteration before they are needed)
    struct Bounds { uint64_t first, last; };
    auto bounds_of = [&](uint64_t r0) {
        r0 = r0 < rec_end ? r0 : rec_first;
        const uint64_t hi = r0 + GW < rec_end ? r0 + GW : rec_end;
        Bounds bd;
        // (in place the record before the step supplies nothing: the stretch starts at the step's own records)
        bd.first = INPLACE ? coffs[r0] : (r0 ? coffs[r0 - 1] : 0);
        bd.last = coffs[hi];
        return bd;
    };
    // puts the loads of the step starting at record r0 in flight towards stage buffer b
    auto issue = [&](uint64_t r0, const Bounds& bd, uint32_t b) {
        r0 = r0 < rec_end ? r0 : rec_first;
        const uint64_t ob = r0 ? r0 - 1 : 0;         // first offset of the slice
        const uint64_t hi = r0 + GW < rec_end ? r0 + GW : rec_end;
        uint64_t a0, n;
        stretch_range(bd.first, bd.last, padded, a0, n);
        const uint32_t nb = n < ST_BYTES ? (uint32_t)n : ST_BYTES;
        const uint32_t base = __builtin_amdgcn_readfirstlane(lds_addr_of(&S.bytes[wave][b][0]));
        const uint32_t obase = __builtin_amdgcn_readfirstlane(lds_addr_of(&S.offs[wave][b][0]));
#pragma unroll
        for (uint32_t row = 0; row < ST_ROWS; row++) {
            // lanes past the stretch re-read its last chunk: no branch, constant load count
            uint32_t off = row * 1024 + lane * 16;
            off = off < nb ? off : nb - 16;
            load_to_lds16(in + a0 + off, base + row * 1024);
        }
        // offsets ob .. hi as 16-byte pairs; a pair that would run past offs[nrec] is moved
        // back to the last whole pair (the reader knows offs[nrec] = total without it)
        const uint32_t npairs = (uint32_t)(hi - ob + 2) >> 1;
        if (lane < npairs) {
            uint64_t p = ob + 2ull * lane;
            p = p < nrec ? p : nrec - 1;
            load_to_lds16(offs + p, obase);
        }
    };

    uint64_t rec0 = rec_first, rec_issue = rec_first;
    Bounds ahead = bounds_of(rec_issue);
#pragma unroll
    for (int i = 0; i < ST_NB - 1; i++) {
        const Bounds bd = ahead;
        ahead = bounds_of(rec_issue + stride);
        issue(rec_issue, bd, (uint32_t)i);
        rec_issue += stride;
    }
    uint32_t b = 0;
    uint32_t rows_stored = 0;     // row stores the previous step issued for certain (wave-uniform)
    while (true) {
        const uint32_t bf = b + ST_NB - 1;
        const Bounds bd = ahead;
        ahead = bounds_of(rec_issue + stride);
        issue(rec_issue, bd, bf >= ST_NB ? bf - ST_NB : bf);
        rec_issue += stride;
        // this step's loads have landed once at most the loads of the ST_NB-1 steps after it
        // and the row stores of the step before it are outstanding
        switch (rows_stored) {
        case 3: wait_vmcnt<(ST_NB - 1) * ST_LOADS + 3>(); break;
        case 2: wait_vmcnt<(ST_NB - 1) * ST_LOADS + 2>(); break;
        case 1: wait_vmcnt<(ST_NB - 1) * ST_LOADS + 1>(); break;
        default: wait_vmcnt<(ST_NB - 1) * ST_LOADS>(); break;
        }
        rows_stored = 0;

        uint8_t* slot = S.bytes[wave][b];
        const uint64_t* L = S.offs[wave][b];
        const uint32_t i0 = rec0 ? 1u : 0u;          // L[i0] = offs[rec0]
        const uint32_t bi = g + i0;
        const uint64_t rec = rec0 + g;
        uint64_t sprev = 0, s = 0, e = 0;
        if (rec < rec_end) {
            s = L[bi];
            e = rec + 1 == nrec ? total : L[bi + 1];
            sprev = rec ? L[bi - 1] : 0;
        }
        // the step as issue() saw it: its records end at e_last, its stretch is [a0, a0 + nst)
        const uint64_t hi = rec0 + GW < rec_end ? rec0 + GW : rec_end;
        const uint64_t s_first = L[i0], e_last = hi == nrec ? total : L[i0 + (uint32_t)(hi - rec0)];
        uint64_t a0, nst;
        stretch_range(INPLACE ? s_first : (rec0 ? L[0] : 0), e_last, padded, a0, nst);

#ifdef SKH_STAGE_CHECK
        // Checked build: what the hand-counted wait above promises -- this step's stretch and
        // offsets slice are in LDS -- compared with global memory; mismatches are counted in the
        // launch's scratch word and reported by the entry point.  LDS is sampled first, before
        // anything else can wait for outstanding memory operations.
        if (!INPLACE) {              // (in place the edges of a stretch are rewritten by the neighbours meanwhile)
            const uint32_t nb = nst < ST_BYTES ? (uint32_t)nst : ST_BYTES;
            uint4 staged[ST_ROWS];
#pragma unroll
            for (uint32_t row = 0; row < ST_ROWS; row++) {
                const uint32_t off = row * 1024 + lane * 16;
                staged[row] = *reinterpret_cast<const uint4*>(slot + (off < nb ? off : 0));
            }
            const uint64_t ob = rec0 ? rec0 - 1 : 0;
            const uint32_t nof = (uint32_t)(hi - ob) + 1;                 // offsets ob .. hi
            const uint64_t staged_off0 = L[lane < nof ? lane : 0], staged_off1 = L[lane + 64 < nof ? lane + 64 : 0];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            uint32_t bad = 0;
#pragma unroll
            for (uint32_t row = 0; row < ST_ROWS; row++) {
                const uint32_t off = row * 1024 + lane * 16;
                if (off < nb) {
                    const uint4 y = ld16(in + a0 + off);
                    bad |= staged[row].x != y.x || staged[row].y != y.y || staged[row].z != y.z || staged[row].w != y.w;
                }
            }
            if (lane < nof && ob + lane < nrec) bad |= staged_off0 != offs[ob + lane];
            if (lane + 64 < nof && ob + lane + 64 < nrec) bad |= staged_off1 != offs[ob + lane + 64];
            if (bad) atomicAdd(&hint->stage_check, 1u);
        }
#endif
        const uint64_t len = e - s;
        const bool is_long = len > SKH_LONG_T;
        if (is_long && j == 0) note_long(hint, rec, s, e);
        const bool work = !is_long && len != 0;
        const uint64_t o0 = s & ~15ull;
        const uint32_t n_own = work ? (uint32_t)((e - o0) >> 4) : 0u;   // chunks that end inside the record
        const uint32_t sp = (uint32_t)(s - o0);
        // chunks this lane does byte by byte: bit 0 = the head chunk, when the record before it
        // is too short to supply its first bytes as one window; bit 1 = the chunk straddling
        // the plane end
        uint32_t bytewise = 0;
        if (!INPLACE && work && j == 0) {
            if (n_own != 0 && sp != 0 && !(sprev <= o0 && sprev + sp >= 16 && e + sp <= padded)) bytewise |= 1u;
            if (e == total && (e & 15) != 0) bytewise |= 2u;
        }
        constexpr uint32_t IT = 3;                   // chunks per lane the coalesced form holds
        // In place (out == in) the step may write the bytes of its own records only: the aligned
        // chunks that lie wholly inside [s_first, e_last) as below, and what is left at either end
        // inside two misaligned 16-byte stores -- the first 16 bytes of the step's first record and
        // the last 16 of its last record, each the mirror image of the record's other end, taken
        // from the stage before anything is written back into it (what these stores overlap of the
        // chunks has the same, final bytes).  A step that does not fit that picture (an edge record
        // shorter than 16 bytes, a head chunk the record before cannot supply as one window, a
        // stretch larger than the stage) goes to the window-pair form, which needs nothing but each
        // record's own bytes in global memory.
        const bool first_rec = INPLACE && work && s == s_first;          // the step's first non-empty record
        const bool last_rec = INPLACE && work && e == e_last;            // ... and its last
        bool pairs = false;
        if (INPLACE && work) {
            if (!first_rec && n_own != 0 && sp != 0 && !(sprev <= o0 && sprev + sp >= 16 && e + sp <= padded)) pairs = true;
            if (len < 16 && ((first_rec && (s & 15) != 0) || (last_rec && (e & 15) != 0))) pairs = true;
        }
        const bool use_pairs = INPLACE && (nst > ST_BYTES || __any(pairs));
        const bool edge_h = !use_pairs && first_rec && j == 0 && (s & 15) != 0,
                   edge_t = !use_pairs && last_rec && j == 0 && (e & 15) != 0;
        uint4 EH = make_uint4(0, 0, 0, 0), ET = EH;
        if (INPLACE) {
            if (edge_h) EH = lds_window_reversed(slot, (uint32_t)(e - 16 - a0));
            if (edge_t) ET = lds_window_reversed(slot, (uint32_t)(s - a0));
            if (COMP && (edge_h || edge_t)) { EH = map16(EH, S.tab.lut); ET = map16(ET, S.tab.lut); }
        }
        if (use_pairs) {
            reverse_record_inplace<COMP>(out, work ? s : 0, work ? e : 0, j, G, S.tab.lut);
        } else if (nst > ST_BYTES) {
            // stretch larger than the stage (an outlier among short records): every window
            // straight from global memory with one misaligned load -- slower, few registers
            const uint64_t q0 = s + e - 16 - o0;
            for (uint32_t k = j; k < n_own; k += G) {
                if (k == 0 && (bytewise & 1u)) continue;
                uint4 r = reverse16(ldu16(in + q0 - 16ull * k));
                if (k == 0 && sp != 0)
                    r = select16(S.tab.low[sp], reverse16(ldu16(in + (sprev + s - 16 - o0))), r);
                if (COMP) r = map16(r, S.tab.lut);
                *reinterpret_cast<uint4*>(out + o0 + 16ull * k) = r;
            }
        } else if (__any(n_own > IT * G)) {
            // some record of the step has more chunks than its lanes can hold at once: every
            // lane stores its chunks as it makes them (64-byte runs per record)
            const uint32_t lq0 = (uint32_t)(s + e - 16 - o0 - a0);      // window of chunk 0 in the slot
            for (uint32_t k = j; k < n_own; k += G) {
                if (k == 0 && (bytewise & 1u)) continue;
                if (INPLACE && k == 0 && first_rec && sp != 0) continue;  // not wholly the step's: see the edge stores
                uint4 r = lds_window_reversed(slot, lq0 - 16 * k);
                if (k == 0 && sp != 0) {
                    // the previous record supplies bytes [0, sp) of this chunk
                    const uint32_t lqp = (uint32_t)(sprev + s - 16 - o0 - a0);
                    r = select16(S.tab.low[sp], lds_window_reversed(slot, lqp), r);
                }
                if (COMP) r = map16(r, S.tab.lut);
                *reinterpret_cast<uint4*>(out + o0 + 16ull * k) = r;
            }
        } else {
            // The usual case.  The chunks the step owns are one contiguous run of the plane,
            // [s_first & ~15, e_last & ~15).  Every lane makes its chunks in registers, all of
            // them are put back into the slot at their OUTPUT positions (after the last window
            // has been read: a wavefront's LDS operations execute in order), and the run leaves
            // as whole-wavefront rows: the same bytes as lane-group stores, in a third of the
            // write requests.
            const uint32_t lq0 = (uint32_t)(s + e - 16 - o0 - a0);
            uint4 r[IT];
#pragma unroll
            for (uint32_t it = 0; it < IT; it++) {
                const uint32_t k = j + it * G;
                if (k < n_own && !(k == 0 && (bytewise & 1u))) {
                    uint4 v = lds_window_reversed(slot, lq0 - 16 * k);
                    if (k == 0 && sp != 0 && !first_rec) {
                        const uint32_t lqp = (uint32_t)(sprev + s - 16 - o0 - a0);
                        v = select16(S.tab.low[sp], lds_window_reversed(slot, lqp), v);
                    }
                    if (COMP) v = map16(v, S.tab.lut);
                    r[it] = v;
                }
            }
            if (bytewise & 1u) {
                uint4 unused;
                reverse_chunk_bytes<COMP, false>(in, nullptr, offs, rec, total, o0, S.tab.lut, r[0], unused);
                bytewise &= ~1u;
            }
            const uint32_t w0 = (uint32_t)(o0 - a0);
#pragma unroll
            for (uint32_t it = 0; it < IT; it++) {
                const uint32_t k = j + it * G;
                if (k < n_own) *reinterpret_cast<uint4*>(slot + w0 + 16 * k) = r[it];
            }
            // in place the rows hold only the chunks that lie wholly inside the step's records
            const uint32_t c0 = __builtin_amdgcn_readfirstlane((uint32_t)(((INPLACE ? s_first + 15 : s_first) & ~15ull) - a0)),
                           c1 = __builtin_amdgcn_readfirstlane((uint32_t)((e_last & ~15ull) - a0));
#pragma unroll
            for (uint32_t row = 0; row < ST_ROWS; row++) {
                if (c0 < c1 && row * 1024 < c1 && (row + 1) * 1024 > c0) {   // wave-uniform: some lane of the row stores
                    const uint32_t c = row * 1024 + lane * 16;
                    if (c >= c0 && c < c1)
                        *reinterpret_cast<uint4*>(out + a0 + c) = *reinterpret_cast<const uint4*>(slot + c);
                    rows_stored++;
                }
            }
        }
        if (INPLACE) {
            if (edge_h) stu16(out + s, EH);
            if (edge_t) stu16(out + e - 16, ET);
        }
#pragma unroll 1
        while (bytewise) {
            const uint64_t o = (bytewise & 1u) ? o0 : o0 + 16ull * n_own;
            bytewise &= (bytewise & 1u) ? ~1u : ~2u;
            reverse_chunk_bytewise<COMP, false>(in, out, nullptr, nullptr, offs, rec, total, o, S.tab.lut);
        }
        rec0 += stride;
        if (rec0 >= rec_end) break;
        b = b + 1 == ST_NB ? 0 : b + 1;
    }
    wait_vmcnt<0>();     // nothing may land in LDS after the wavefront has gone
}

template <bool COMP, bool INPLACE>
__global__ __launch_bounds__(SKH_BLOCK) void k_reverse_staged(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                              const uint64_t* __restrict__ offs, uint64_t total,
                                                              uint64_t nrec, Table256 lut, int logG_host, uint32_t steps,
                                                              BatchHint* __restrict__ hint, int use_hint)
{
    __shared__ StageLds S;
    reverse_staged<COMP, INPLACE>(S, in, out, offs, total, nrec, lut, logG_host, steps, hint, use_hint);
}

// Long records: one aligned run of LONG_RUN bytes of the plane per workgroup (see for_each_long).
// ------------------------------------------------------------------ reverse, flat form
//
// The output of consecutive records is one contiguous stretch of the plane.  A wavefront takes R records per
// step (as many as fit its LDS stage by the mean length, at most 64; lane i holds the offsets of record i) and
// deals the 16-byte pieces of ALL of them over its 64 lanes: every record gets PF piece slots (the most any
// record of the step needs), slot q = lane + 64 k is piece q mod PF of record q / PF.  A piece at offset t of
// a record of length len is the source window [len - t - 16, len - t) of the same record, byte-reversed (and
// complemented); the end of a record goes as its last 16 output bytes once more.  The pieces are put together
// in the stage, which costs nothing at any byte offset, and the stretch leaves as aligned 16-byte chunks in
// whole-wavefront rows.  The work per piece does not depend on where records begin and end inside a chunk.
// A step that does not fit the stage is written record by record by the whole wavefront, straight to global
// memory; records beyond SKH_LONG_T are left to k_reverse_tiles.
constexpr uint32_t RF_STAGE = 7 * 1024;        // (4 and 5 KiB: 3-6 points slower on 150-base reads)

__host__ __device__ inline uint32_t flat_records_per_step(uint64_t mean_len)
{
    const uint64_t r = (RF_STAGE - 64) * 7 / 8 / (mean_len + 1);
    return r < 1 ? 1u : (r > 64 ? 64u : (uint32_t)r);
}

__device__ __forceinline__ uint64_t shfl64u(uint64_t v, uint32_t src_lane)
{
    return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)src_lane) << 32) |
           (uint32_t)__shfl((int)(uint32_t)v, (int)src_lane);
}

// INPLACE: in == out.  A step reads and writes the bytes of its own records only, and all its loads have landed
// in the stage before its first store, so the staged steps are the same; a step that does not fit the stage
// reverses its records by window pairs, and records beyond SKH_LONG_T are left whole to k_reverse_inplace_long.
struct FlatLds {
    __attribute__((aligned(16))) uint8_t stage[SKH_BLOCK / 64][RF_STAGE + 32];
    uint8_t owner[SKH_BLOCK / 64][RF_STAGE / 16 + 64];       // a stage holds RF_STAGE / 16 whole pieces + one partial per record
    uint8_t lut[256];
};

template <bool COMP, bool INPLACE>
__device__ __forceinline__ void reverse_flat(FlatLds& F, const uint8_t* in, uint8_t* out, const uint64_t* __restrict__ offs, uint64_t total,
                                             uint64_t nrec, const Table256& lut, uint32_t steps, BatchHint* __restrict__ hint,
                                             int use_hint)
{
    uint8_t* s_lut = F.lut;
    auto& s_stage = F.stage;
    auto& s_owner = F.owner;
    static_assert(SKH_BLOCK == 256, "one table byte per thread");
    s_lut[threadIdx.x] = lut.v[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr uint32_t NWAVE = SKH_BLOCK / 64;
    // The grid is sized for the records per step that the batch's mean length gives.  A few very long records among
    // short reads raise that mean: the mean of the sampled short records then gives more records per step, and the
    // last workgroups of the grid find nothing left to do.
    uint64_t mean_len = total / nrec;                    // nrec != 0: the grid is not empty
    uint32_t R = flat_records_per_step(mean_len);
    if (use_hint) {
        const uint32_t mean_short = __builtin_amdgcn_readfirstlane(hint->mean_short);
        // (only when it is well below the batch mean: it comes from a sample, and records per step chosen a few per
        // cent too high make steps overflow the stage)
        if (mean_short != ~0u && 5ull * mean_short < 4 * mean_len && flat_records_per_step(mean_short) > R) {
            R = flat_records_per_step(mean_short);
            mean_len = mean_short;
        }
    }
    const uint64_t nsteps = (nrec + R - 1) / R;
    // (twice the mean length; four times for short reads, where one record in a hundred of a log-normal batch is
    // beyond twice the mean and every such record costs its step a memory round trip of its own)
    const uint32_t pf_cap = (uint32_t)(((mean_len < 256 ? 4 : 2) * mean_len + 15) >> 4) + 1;
    // workgroup b serves the steps * NWAVE consecutive steps from b * steps * NWAVE on, its wavefronts interleaved
    const uint64_t step_first = (uint64_t)blockIdx.x * steps * NWAVE + wave;
    const uint64_t step_stop = (uint64_t)(blockIdx.x + 1) * steps * NWAVE;
    const uint64_t step_end = step_stop < nsteps ? step_stop : nsteps;
    uint8_t* stage = s_stage[wave];
    for (uint64_t step = step_first; step < step_end; step += NWAVE) {
        const uint64_t i0 = step * R;
        const uint32_t cnt = nrec - i0 < R ? (uint32_t)(nrec - i0) : R;
        uint64_t t_o = 0, t_e = 0;
        if (lane < cnt) { t_o = offs[i0 + lane]; t_e = offs[i0 + lane + 1]; }
        const uint64_t p0 = shfl64u(t_o, 0), p1 = shfl64u(t_e, cnt - 1);
        if (p1 == p0) continue;
        const uint64_t a0 = p0 & ~15ull;
        if (p1 - a0 > RF_STAGE) {                        // wave-uniform: record by record, straight to global memory
            // Records of kilobytes by the whole wavefront, one after the other (rounds 0 .. cnt - 1); then the short
            // ones among them, eight at a time by groups of eight lanes.  (Most such steps are steps of short reads
            // with ONE long record among them: served one by one, forty reads cost the wavefront forty memory round
            // trips, and the last such wavefront of a launch ended 60 us after every other.)
            constexpr uint32_t GROUPED = 2048;           // records up to this length go to the lane groups
            const uint32_t rounds = cnt + ((cnt + 7) >> 3);
#pragma unroll 1
            for (uint32_t it = 0; it < rounds; it++) {
                const bool grouped = it >= cnt;          // wave-uniform
                const uint32_t i = grouped ? ((it - cnt) << 3) + (lane >> 3) : it;
                const uint32_t j = grouped ? (lane & 7u) : lane, G = grouped ? 8u : 64u;
                const uint64_t s = shfl64u(t_o, i & 63u), e = shfl64u(t_e, i & 63u);
                if (i >= cnt || (e - s <= GROUPED) != grouped) continue;
                if (e - s > SKH_LONG_T) {
                    // k_reverse_tiles writes the chunks that end inside the record: the bytes of its last, partial
                    // chunk are written here
                    if (lane == 0) note_long(hint, i0 + i, s, e);
                    if (INPLACE) continue;
                    const uint64_t x = (e & ~15ull) + lane;
                    if (x < e) { const uint8_t c = in[s + (e - 1 - x)]; out[x] = COMP ? s_lut[c] : c; }
                    continue;
                }
                if (INPLACE) {
                    reverse_record_inplace<COMP>(out, s, e, j, G, s_lut);
                    continue;
                }
                const uint32_t len = (uint32_t)(e - s);
                if (len < 16) {
                    for (uint32_t bb = j; bb < len; bb += G) { const uint8_t c = in[e - 1 - bb]; out[s + bb] = COMP ? s_lut[c] : c; }
                    continue;
                }
                // aligned 16-byte chunks of the output (the bytes before the first whole chunk go one per lane, the
                // end of the record as its last 16 bytes once more), the loads of four rounds ahead of the stores
                const uint32_t head = (uint32_t)(-s) & 15u;
                for (uint32_t bb = j; bb < head; bb += G) { const uint8_t c = in[e - 1 - bb]; out[s + bb] = COMP ? s_lut[c] : c; }      // head < 16 <= len
                constexpr int FU = 4;
                const uint32_t row = 16 * G;
                for (uint32_t pb = head + j * 16; pb < len; pb += row * FU) {
                    uint4 v[FU];
#pragma unroll
                    for (int u = 0; u < FU; u++) {
                        const uint32_t p = pb + row * u, t = p + 16 <= len ? p : len - 16;
                        v[u] = p < len ? ldu16(in + s + (len - t - 16)) : make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < FU; u++) {
                        const uint32_t p = pb + row * u, t = p + 16 <= len ? p : len - 16;
                        if (p >= len) continue;
                        uint4 w = reverse16(v[u]);
                        if (COMP) w = map16(w, s_lut);
                        __builtin_memcpy(out + s + t, &w, 16);
                    }
                }
            }
            continue;
        }
        const uint32_t t_len = (uint32_t)(t_e - t_o), d_rel = (uint32_t)(t_o - a0), s_rel = (uint32_t)(t_o - p0);
        const uint32_t packed = t_len | (d_rel << 16), head0 = (uint32_t)(p0 - a0);     // s_rel = d_rel - head0
        const uint32_t pieces = (t_len + 15) >> 4;
        uint32_t PF = pieces;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)PF, d); PF = t > PF ? t : PF; }
        // (a few records far longer than the rest would leave most slots empty: slots for a multiple of the mean
        // length, what a record has beyond them is copied after the slots)
        PF = PF < pf_cap ? PF : pf_cap;
        // Records of unequal lengths would leave many of the cnt * PF slots empty: then every record gets as many
        // slots as it has pieces (at most PF), and a byte table in LDS, filled by the records' own lanes, says
        // which record a slot belongs to.
        // (decided on the step's byte count: its pieces number between bytes / 16 and bytes / 16 + cnt)
        const uint32_t est = ((uint32_t)(p1 - p0) >> 4) + (cnt >> 1);
        const bool ragged = cnt * PF > est + (est >> 2);                // wave-uniform
        uint8_t* owner = s_owner[wave];
        uint32_t pre = 0, npieces = 0;
        if (ragged) {
            const uint32_t mine = pieces < PF ? pieces : PF;
            uint32_t incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, d); if ((int)lane >= d) incl += t; }
            npieces = (uint32_t)__shfl((int)incl, 63);
            pre = incl - mine;
            for (uint32_t k = 0; k < mine; k++) owner[pre + k] = (uint8_t)lane;
            wave_lds_handover();
        }
        const uint32_t nslots = ragged ? npieces : cnt * PF;           // PF >= 1: the step has bytes
        const uint32_t magic = PF > 1 ? (uint32_t)((0x100000000ull + PF - 1) / PF) : 0u;   // q / PF for q < 2^16
        const uint8_t* src0 = in + p0;
        constexpr int PU = 4;                            // rounds of slots whose loads go out together (2: level, 7: slower)
        for (uint32_t q0 = 0; q0 < nslots; q0 += 64 * PU) {   // every lane takes part in the cross-lane reads
            uint4 v[PU];
            uint32_t at[PU];
            bool act[PU];
#pragma unroll
            for (int u = 0; u < PU; u++) {
                const uint32_t qq = q0 + 64 * u + lane;
                const uint32_t q = qq < nslots ? qq : nslots - 1;
                uint32_t r, k;
                if (ragged) {
                    r = owner[q];
                    k = q - (uint32_t)__shfl((int)pre, (int)r);
                } else {
                    r = PF > 1 ? __umulhi(q, magic) : q;
                    k = q - r * PF;
                }
                // (length and place of the record in one cross-lane read: both are below 2^16 in a step that fits the stage)
                const uint32_t pk = (uint32_t)__shfl((int)packed, (int)r);
                const uint32_t len = pk & 0xffffu, dr = pk >> 16, sr = dr - head0;
                const uint32_t p = 16 * k;
                const bool in_rec = qq < nslots && p < len;
                act[u] = in_rec && len >= 16;
                const uint32_t t = p + 16 <= len ? p : len - 16;
                at[u] = dr + t;
                v[u] = act[u] ? ldu16(src0 + sr + (len - t - 16)) : make_uint4(0, 0, 0, 0);
                if (in_rec && len < 16)
                    for (uint32_t b = 0; b < len; b++) { const uint8_t c = src0[sr + len - 1 - b]; stage[dr + b] = COMP ? s_lut[c] : c; }
            }
#pragma unroll
            for (int u = 0; u < PU; u++) {
                if (!act[u]) continue;
                uint4 w = reverse16(v[u]);
                if (COMP) w = map16(w, s_lut);
                __builtin_memcpy(stage + at[u], &w, 16);
            }
        }
        unsigned long long more = __ballot(pieces > PF);
        while (more) {                                   // the pieces of a long record beyond its slots, by the whole wavefront
            const uint32_t i = (uint32_t)__builtin_ctzll(more);
            more &= more - 1;
            const uint32_t len = (uint32_t)__shfl((int)t_len, (int)i), dr = (uint32_t)__shfl((int)d_rel, (int)i);
            const uint32_t sr = (uint32_t)__shfl((int)s_rel, (int)i);
            for (uint32_t p = 16 * (PF + lane); p < len; p += 1024) {      // len > 16 PF >= 16
                const uint32_t t = p + 16 <= len ? p : len - 16;
                uint4 w = reverse16(ldu16(src0 + sr + (len - t - 16)));
                if (COMP) w = map16(w, s_lut);
                __builtin_memcpy(stage + dr + t, &w, 16);
            }
        }
        wave_lds_handover();
        const uint32_t nbytes = (uint32_t)(p1 - a0), head = (uint32_t)(p0 - a0);
        const uint32_t c_lo = head ? 16u : 0u, c_hi = nbytes & ~15u;
        for (uint32_t c = lane * 16; c < c_hi; c += 1024)
            if (c >= c_lo) *reinterpret_cast<uint4*>(out + a0 + c) = *reinterpret_cast<const uint4*>(stage + c);
        if (head && lane >= head && lane < 16 && lane < nbytes) out[a0 + lane] = stage[lane];
        if (c_hi < nbytes && (c_hi != 0 || !head) && lane < nbytes - c_hi) out[a0 + c_hi + lane] = stage[c_hi + lane];
        wave_lds_handover();
    }
}

template <bool COMP>
__global__ __launch_bounds__(SKH_BLOCK) void k_reverse_flat(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                            const uint64_t* __restrict__ offs, uint64_t total,
                                                            uint64_t nrec, Table256 lut, uint32_t steps,
                                                            BatchHint* __restrict__ hint, int use_hint)
{
    __shared__ FlatLds F;
    reverse_flat<COMP, false>(F, in, out, offs, total, nrec, lut, steps, hint, use_hint);
}

template <bool COMP>
__global__ __launch_bounds__(SKH_BLOCK) void k_reverse_flat_inplace(uint8_t* plane, const uint64_t* __restrict__ offs,
                                                                    uint64_t total, uint64_t nrec, Table256 lut,
                                                                    uint32_t steps, BatchHint* __restrict__ hint, int use_hint)
{
    __shared__ FlatLds F;
    reverse_flat<COMP, true>(F, plane, plane, offs, total, nrec, lut, steps, hint, use_hint);
}

template <bool COMP, bool FASTQ>
__global__ __launch_bounds__(SKH_BLOCK) void k_reverse_tiles(RevPlanes P, uint64_t nrec, Table256 lut,
                                                             const BatchHint* __restrict__ hint)
{
    if (hint->n_long == 0) return;
    __shared__ RevLds L;
    __shared__ WalkLds W;
    rev_lds_init<COMP>(L, lut);
    constexpr int U = 4;
    const uint64_t total = P.total;
    const uint64_t t0 = (uint64_t)blockIdx.x * LONG_RUN;
    if (t0 >= total) return;
    const uint64_t t1 = t0 + LONG_RUN;                           // exclusive, may exceed total
    for_each_long(P.offs, nrec, t0, t1 < total ? t1 : total, W, hint, [&](uint64_t r, uint64_t s, uint64_t e) {
        const uint64_t sprev = r ? P.offs[r - 1] : 0;
        const uint64_t o0 = s & ~15ull;
        const uint32_t n_own = (uint32_t)((e - o0) >> 4);        // < 2^28 chunks: records < 4 GiB
        // owned chunks of the record that lie in this run: indices [k0, k1)
        const uint32_t k0 = o0 >= t0 ? 0u : (uint32_t)((t0 - o0) >> 4);
        const uint64_t own_end = o0 + 16ull * n_own;
        const uint32_t k1 = own_end <= t1 ? n_own : (uint32_t)((t1 - o0) >> 4);
        for (uint32_t kb = k0; kb < k1; kb += U * SKH_BLOCK)
            reverse_batch<COMP, FASTQ, U, true>(P, L, r, sprev, s, e, kb + threadIdx.x, SKH_BLOCK, k1);
        if (e == total && (e & 15) != 0 && own_end >= t0 && own_end < t1 && threadIdx.x == 0)
            reverse_chunk_generic<COMP, FASTQ>(P.in, P.out, P.qin, P.qout, P.offs, r, total, own_end, L.lut);
    });
}

// ---- in place: kernels (the window-pair helpers are defined ahead of the staged kernel) ------

// Records of at most LONG_T bytes: a group of 2^logG lanes per record.
template <bool COMP>
__global__ __launch_bounds__(SKH_BLOCK) void k_reverse_inplace_short(uint8_t* __restrict__ p,
                                                                     const uint64_t* __restrict__ offs, uint64_t total,
                                                                     uint64_t nrec, Table256 lut, int logG_host,
                                                                     BatchHint* __restrict__ hint, int use_hint)
{
    __shared__ RevLds L;
    const uint32_t mean_short = hint->mean_short;
    rev_lds_init<COMP>(L, lut);
    const int logG = hinted_logG<false>(logG_host, mean_short, use_hint);
    const uint32_t G = 1u << logG, j = threadIdx.x & (G - 1);
    const uint32_t lane = threadIdx.x & 63, GW = 64u >> logG, g = lane >> logG;
    const uint64_t stride = (uint64_t)gridDim.x * (SKH_BLOCK / 64) * GW;
    uint64_t rec0 = ((uint64_t)xcd_remap(blockIdx.x, gridDim.x) * (SKH_BLOCK / 64) + (threadIdx.x >> 6)) * GW;
    if (rec0 >= nrec) return;
    WaveOffsets wo;
    wo.load(offs, nrec, total, rec0, GW, lane);
    while (true) {
        const uint64_t next0 = rec0 + stride;
        WaveOffsets nwo;
        if (next0 < nrec) nwo.load(offs, nrec, total, next0, GW, lane);
        const uint64_t s = wo.get(g + 1, GW), e = wo.get(g + 2, GW);
        const uint64_t len = e - s;
        if (len > SKH_LONG_T) {
            if (j == 0) note_long(hint, rec0 + g, s, e);
        } else {
            reverse_record_inplace<COMP>(p, s, e, j, G, L.lut);
        }
        if (next0 >= nrec) break;
        rec0 = next0;
        wo = nwo;
    }
}

// Longer records: a workgroup serves the window pairs whose FRONT window starts in its run of the
// plane (workgroups whose runs lie in the back halves of records find nothing to do).  The front
// windows are moved to 16-byte alignment: the h = -s mod 16 bytes at either end of the record are
// swapped byte by byte by the workgroup that holds s.
template <bool COMP, uint64_t RUN>
__global__ __launch_bounds__(SKH_BLOCK) void k_reverse_inplace_long(uint8_t* __restrict__ p,
                                                                    const uint64_t* __restrict__ offs, uint64_t total,
                                                                    uint64_t nrec, Table256 lut,
                                                                    const BatchHint* __restrict__ hint)
{
    if (hint->n_long == 0) return;
    __shared__ RevLds L;
    __shared__ WalkLds W;
    rev_lds_init<COMP>(L, lut);
    constexpr int U = 4;
    const uint64_t t0 = (uint64_t)blockIdx.x * RUN;
    if (t0 >= total) return;
    const uint64_t t1 = t0 + RUN < total ? t0 + RUN : total;
    for_each_long(offs, nrec, t0, t1, W, hint, [&](uint64_t, uint64_t s, uint64_t e) {
        const uint32_t h = (uint32_t)(-(int64_t)s) & 15u;
        const uint64_t fs = s + h, be = e - h;                      // windows cover [fs, be); fs is aligned
        const uint64_t len = be - fs;                               // > LONG_T - 30
        const uint64_t n = len >> 5;
        const uint32_t M = (uint32_t)len & 31u;
        const uint64_t m0 = fs + 16 * n;
        // pairs whose front window starts in [t0, t1)
        const uint64_t i0 = fs >= t0 ? 0 : (t0 - fs) >> 4;
        uint64_t i1 = fs >= t1 ? 0 : (t1 - fs + 15) >> 4;
        if (i1 > n) i1 = n;
        for (uint64_t ib = i0; ib < i1; ib += (uint64_t)U * SKH_BLOCK) {
            uint4 F[U], B[U];
            MiddleInPlace<COMP> mid;
            bool inner = false;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t i = ib + threadIdx.x + (uint64_t)u * SKH_BLOCK;
                if (i < i1) {
                    F[u] = ld16(p + fs + 16 * i);
                    B[u] = ldu16(p + be - 16 * i - 16);
                    if (i == n - 1 && M != 0) inner = true;
                }
            }
            if (inner) mid.load(p, m0, M);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t i = ib + threadIdx.x + (uint64_t)u * SKH_BLOCK;
                if (i < i1) {
                    *reinterpret_cast<uint4*>(p + fs + 16 * i) = reversed_mapped<COMP>(B[u], L.lut);
                    stu16(p + be - 16 * i - 16, reversed_mapped<COMP>(F[u], L.lut));
                }
            }
            if (inner) mid.store(p, m0, M, L.lut);
        }
        if (h != 0 && s >= t0 && s < t1 && threadIdx.x < h) {       // the unaligned ends, one byte pair per thread
            const uint8_t x = p[s + threadIdx.x], y = p[e - 1 - threadIdx.x];
            p[s + threadIdx.x] = COMP ? L.lut[y] : y;
            p[e - 1 - threadIdx.x] = COMP ? L.lut[x] : x;
        }
    });
}

// ---- span form: lanes own consecutive output chunks of the plane ------------------------
//
// A workgroup walks a contiguous range of the plane in spans of U*256 aligned
// 16-byte chunks; lane t owns chunks t, t+256, ... of the span, so every store
// instruction writes 1 KiB of consecutive bytes whatever the record lengths are, and
// work is balanced by bytes, not by records.  The offsets of the records that
// overlap the span are staged in LDS once per span; each lane finds the owner of
// its chunk (the record holding the chunk's last byte) by a binary search there.
// Short and long records take the same path: a chromosome-length record is simply
// the owner of many consecutive spans.
constexpr int OFF_CAP = 2048;     // offsets staged per span; more records -> the span is cut short

struct SpanLds {
    RevLds tab;
    uint64_t off[OFF_CAP];        // offs[a + t]
    int32_t rel[OFF_CAP];         // the same relative to the span start, saturated: search keys
    uint64_t cursor;              // owner of the span's last chunk, seeds the next span
};

template <bool COMP, int U>
__global__ __launch_bounds__(SKH_BLOCK) void k_reverse_span(RevPlanes P, uint64_t nrec, Table256 lut)
{
    __shared__ SpanLds S;
    rev_lds_init<COMP>(S.tab, lut);
    constexpr uint32_t SPAN_CH = U * SKH_BLOCK;          // chunks per span
    const uint32_t tid = threadIdx.x;
    const uint64_t total = P.total, padded = (total + 15) & ~15ull;
    const uint64_t nch = padded >> 4;
    const uint64_t nspan = (nch + SPAN_CH - 1) / SPAN_CH;
    const uint64_t per_block = (nspan + gridDim.x - 1) / gridDim.x;
    const uint64_t span0 = (uint64_t)xcd_remap(blockIdx.x, gridDim.x) * per_block;
    if (span0 >= nspan) return;
    uint64_t c_next = span0 * SPAN_CH;
    uint64_t c_end = (span0 + per_block) * SPAN_CH;
    if (c_end > nch) c_end = nch;
    // cursor: a record r_cur with offs[r_cur] <= first byte of the next span
    uint64_t r_cur = find_record(P.offs, nrec, c_next << 4);

    while (c_next < c_end) {
        const uint64_t B0 = c_next << 4;
        const uint32_t nc_want = c_end - c_next < SPAN_CH ? (uint32_t)(c_end - c_next) : (uint32_t)SPAN_CH;
        const uint64_t span_last = (B0 + 16ull * nc_want < total ? B0 + 16ull * nc_want : total) - 1;
        // ---- stage offsets a, a+1, ... until one exceeds the span's last byte
        const uint64_t a = r_cur ? r_cur - 1 : 0;
        // (the staged count stays a power of two so the search below can halve its step)
        uint32_t n_staged = 0;
        int found = 0;
        for (uint32_t target = SKH_BLOCK; !found && target <= OFF_CAP; target <<= 1) {
            int mine = 0;
            for (; n_staged < target; n_staged += SKH_BLOCK) {
                const uint32_t t = n_staged + tid;
                const uint64_t idx = a + t;
                const uint64_t v = idx <= nrec ? P.offs[idx] : ~0ull;
                S.off[t] = v;
                const int64_t d = (int64_t)(v - B0);  // v >= offs[a]: may be below B0 by any amount
                S.rel[t] = v == ~0ull ? INT32_MAX
                                      : (d > INT32_MAX ? INT32_MAX : (d < INT32_MIN ? INT32_MIN : (int32_t)d));
                mine |= v > span_last;
            }
            found = __syncthreads_or(mine);
        }
        // chunks whose last byte lies below the last staged offset can be resolved
        uint32_t nc = nc_want;
        if (!found) {
            const uint64_t m = S.off[n_staged - 1];
            nc = m >= B0 + 16 ? (uint32_t)((m - B0) >> 4) : 0u;
            if (nc > nc_want) nc = nc_want;
        }
        if (nc == 0) {
            // more than OFF_CAP records end inside one chunk (runs of empty records): one lane
            // resolves this single chunk from global memory
            const uint64_t pl = B0 + 15 < total ? B0 + 15 : total - 1;
            const uint64_t r = find_record(P.offs, nrec, pl);
            if (tid == 0)
                reverse_chunk_generic<COMP, false>(P.in, P.out, nullptr, nullptr, P.offs, r, total, B0, S.tab.lut);
            __syncthreads();
            r_cur = r;
            c_next += 1;
            continue;
        }

        // ---- per chunk: owner search, then the aligned source chunk (all loads before first use).
        // meta: bits 0-1 kind (0 none, 1 interior, 2 head, 3 generic), 2-5 window offset dq,
        //       6-9 sp (bytes of a head chunk before its owner), 10.. owner's index in S.off
        uint4 W[U], E[U], HP[U];
        uint32_t meta[U];
        uint64_t alo[U];
        const uint32_t lane = tid & 63;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t c = tid + u * SKH_BLOCK;
            meta[u] = 0;
            alo[u] = 0;
            W[u] = make_uint4(0, 0, 0, 0);
            HP[u] = make_uint4(0, 0, 0, 0);
            if (c < nc) {
                const uint64_t o = B0 + 16ull * c;
                const int32_t p_rel = (o + 15 < total ? 16 * (int32_t)c + 15 : (int32_t)(total - 1 - B0));
                uint32_t i = 0;
                for (uint32_t step = n_staged >> 1; step != 0; step >>= 1)
                    if (S.rel[i + step] <= p_rel) i += step;
                const uint64_t s = S.off[i], e = S.off[i + 1];
                const uint64_t sprev = i ? S.off[i - 1] : 0;
                const uint64_t q = s + e - 16 - o;
                const uint32_t dq = (uint32_t)q & 15u;
                alo[u] = q - dq;
                uint32_t kind = 3, sp = 0;
                if (o + 16 > total) {
                    kind = 3;                                     // the chunk straddling the plane end
                } else if (o >= s) {
                    kind = 1;
                    W[u] = ld16(P.in + alo[u]);
                } else {
                    sp = (uint32_t)(s - o);                       // bytes of the chunk before the owner
                    if (sprev <= o && sprev + sp >= 16) {
                        kind = 2;
                        W[u] = ld16(P.in + alo[u]);
                        HP[u] = ldu16(P.in + (sprev + s - 16 - o));   // window in the previous record
                    }
                }
                meta[u] = kind | (dq << 2) | (sp << 6) | (i << 10);
                if (c == nc - 1) S.cursor = a + i;
            }
        }
        // The chunk a window spills into is the aligned chunk of the lane before, when that lane
        // serves the previous chunk of the same record; otherwise it is loaded here.
        bool own_upper[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t nbm = __shfl(meta[u], (int)lane - 1);
            const uint32_t kind = meta[u] & 3u, nbk = nbm & 3u;
            const bool same = lane != 0 && (nbk == 1 || nbk == 2) && (nbm >> 10) == (meta[u] >> 10);
            own_upper[u] = (kind == 1 || kind == 2) && ((meta[u] >> 2) & 15u) != 0 && !same;
            E[u] = make_uint4(0, 0, 0, 0);
            if (own_upper[u] && alo[u] + 32 <= padded) E[u] = ld16(P.in + alo[u] + 16);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t kind = meta[u] & 3u, dq = (meta[u] >> 2) & 15u, sp = (meta[u] >> 6) & 15u;
            const uint64_t o = B0 + 16ull * (tid + u * SKH_BLOCK);
            uint4 n = shfl16(W[u], (int)lane - 1);          // cross-lane: every lane, no divergence
            if (own_upper[u]) n = E[u];
            uint4 r = window_reversed(W[u], n, dq);
            if (kind == 2) r = select16(S.tab.low[sp], reverse16(HP[u]), r);
            if (COMP) r = map16(r, S.tab.lut);
            if (kind == 3)
                reverse_chunk_generic<COMP, false>(P.in, P.out, nullptr, nullptr, P.offs, a + (meta[u] >> 10), total, o,
                                                   S.tab.lut);
            else if (kind != 0)
                *reinterpret_cast<uint4*>(P.out + o) = r;
        }
        __syncthreads();                     // S.off / S.rel are rewritten by the next span
        r_cur = S.cursor;
        c_next += nc;
    }
}

// No reversal: out[i] = lut[in[i]] over the whole plane (record boundaries do not matter).
// The grid covers the plane once, MAP_U chunks per lane, all loads of a lane ahead of its first
// table lookup; in == out is allowed (a chunk is read and written by the same lane).
constexpr int MAP_U = 4;

__global__ __launch_bounds__(SKH_BLOCK) void k_map_plane(const uint8_t* in, uint8_t* out, uint64_t total, Table256 lut)
{
    __shared__ uint8_t s_lut[256];
    s_lut[threadIdx.x] = lut.v[threadIdx.x];
    __syncthreads();
    const uint64_t nchunk = (total + 15) >> 4;
    const uint64_t c0 = (uint64_t)blockIdx.x * (MAP_U * SKH_BLOCK) + threadIdx.x;
    uint4 v[MAP_U];
#pragma unroll
    for (int u = 0; u < MAP_U; u++) {
        const uint64_t c = c0 + (uint64_t)u * SKH_BLOCK;
        if (c < nchunk) v[u] = ld16(in + (c << 4));
    }
#pragma unroll
    for (int u = 0; u < MAP_U; u++) {
        const uint64_t c = c0 + (uint64_t)u * SKH_BLOCK;
        if (c < nchunk) store_clipped(out, 0, total, c << 4, map16(v[u], s_lut));
    }
}

// ------------------------------------------------------------------ counting

// LDS tables of the counting kernels: one packed word per byte value holding a 1
// in the 8-bit field of the byte's class (NW = 1: four classes in a dword,
// NW = 2: eight classes in two dwords), and 17 byte masks (low k bytes set) used
// to neutralise the bytes of a boundary chunk that belong to a neighbour record.
template <int NW>
struct CountLds {
    uint32_t lut[256 * NW];
    uint4 low[17];
};

template <int NW>
__device__ __forceinline__ void count_lds_init(CountLds<NW>& L, const Table256& cls)
{
    const uint32_t b = threadIdx.x;
    const uint32_t c = cls.v[b];
    if (NW == 1) {
        L.lut[b] = c < 4 ? 1u << (8 * c) : 0u;
    } else {
        L.lut[2 * b] = c < 4 ? 1u << (8 * c) : 0u;
        L.lut[2 * b + 1] = (c >= 4 && c < 8) ? 1u << (8 * (c - 4)) : 0u;
    }
    if (b < 17) {
        uint32_t w[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int k = (int)b - 4 * m;    // low bytes of dword m that are set
            w[m] = k <= 0 ? 0u : (k >= 4 ? 0xffffffffu : (1u << (8 * k)) - 1u);
        }
        L.low[b] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
}

// acc8[NW] += one-hot class word of each of the 4 bytes of w (8-bit fields).
template <int NW>
__device__ __forceinline__ void count4(uint32_t w, const uint32_t* lut, uint32_t* acc8)
{
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const uint32_t b = (w >> (8 * t)) & 0xffu;
        if (NW == 1) {
            acc8[0] += lut[b];
        } else {
            const uint2 p = *reinterpret_cast<const uint2*>(&lut[2 * b]);
            acc8[0] += p.x;
            acc8[1] += p.y;
        }
    }
}

// Adds the class counts of the bytes of aligned chunk `o` that lie in [s, e).
// `neutral4` is a byte value of no class replicated four times; the bytes of a
// boundary chunk outside the record are replaced by it before the table lookup.
// Without such a byte (every value classified) they are skipped one by one.
template <int NW>
__device__ __forceinline__ void count_loaded(uint4 v, uint64_t s, uint64_t e, uint64_t o,
                                             const CountLds<NW>& L, uint32_t neutral4, bool has_neutral,
                                             uint32_t* acc8)
{
    if (o < s || o + 16 > e) {
        const int lo = s > o ? (int)(s - o) : 0;
        const int hi = e < o + 16 ? (int)(e - o) : 16;
        if (has_neutral) {
            const uint4 ml = L.low[lo], mh = L.low[hi];
            v.x = (v.x & mh.x & ~ml.x) | (neutral4 & ~(mh.x & ~ml.x));
            v.y = (v.y & mh.y & ~ml.y) | (neutral4 & ~(mh.y & ~ml.y));
            v.z = (v.z & mh.z & ~ml.z) | (neutral4 & ~(mh.z & ~ml.z));
            v.w = (v.w & mh.w & ~ml.w) | (neutral4 & ~(mh.w & ~ml.w));
        } else {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k >= lo && k < hi) {
                    const uint32_t b = (w[k >> 2] >> (8 * (k & 3))) & 0xffu;
                    acc8[0] += L.lut[NW * b];
                    if (NW == 2) acc8[1] += L.lut[2 * b + 1];
                }
            }
            return;
        }
    }
    count4<NW>(v.x, L.lut, acc8);
    count4<NW>(v.y, L.lut, acc8);
    count4<NW>(v.z, L.lut, acc8);
    count4<NW>(v.w, L.lut, acc8);
}

template <int NW>
__device__ __forceinline__ void count_chunk(const uint8_t* __restrict__ seq, uint64_t s, uint64_t e,
                                            uint64_t o, const CountLds<NW>& L, uint32_t neutral4,
                                            bool has_neutral, uint32_t* acc8)
{
    count_loaded<NW>(ld16(seq + o), s, e, o, L, neutral4, has_neutral, acc8);
}

// 8-bit fields -> 16-bit fields: acc16[2w] takes classes 4w+0 and 4w+2, acc16[2w+1]
// classes 4w+1 and 4w+3.
template <int NW>
__device__ __forceinline__ void widen(uint32_t* acc8, uint32_t* acc16)
{
#pragma unroll
    for (int w = 0; w < NW; w++) {
        acc16[2 * w] += acc8[w] & 0x00ff00ffu;
        acc16[2 * w + 1] += (acc8[w] >> 8) & 0x00ff00ffu;
        acc8[w] = 0;
    }
}

template <int NW>
__device__ __forceinline__ uint32_t field16(const uint32_t* acc16, uint32_t c)
{
    // class c lives in word 2*(c/4) + (c&1), half (c>>1)&1
    uint32_t w = acc16[0];
    if ((c & 5u) == 1u) w = acc16[1];
    if (NW == 2) {
        if ((c & 5u) == 4u) w = acc16[2];
        if ((c & 5u) == 5u) w = acc16[3];
    }
    return (c & 2u) ? (w >> 16) : (w & 0xffffu);
}

template <int NW, int U>
__global__ __launch_bounds__(SKH_BLOCK) void k_count_short(
    const uint8_t* __restrict__ seq, const uint64_t* __restrict__ offs, uint64_t nrec, uint64_t total,
    Table256 cls, uint32_t nclass, int logG_host, uint32_t steps, uint32_t neutral, uint64_t* __restrict__ counts,
    BatchHint* __restrict__ hint, int use_hint)
{
    __shared__ CountLds<NW> L;
    const uint32_t mean_short = hint->mean_short;        // asked for before the table set-up, needed after it
    count_lds_init<NW>(L, cls);
    const int logG = hinted_logG<true, U>(logG_host, mean_short, use_hint);
    const bool has_neutral = neutral < 256u;
    const uint32_t neutral4 = (neutral & 0xffu) * 0x01010101u;
    const uint32_t G = 1u << logG, j = threadIdx.x & (G - 1);
    const uint32_t lane = threadIdx.x & 63, GW = 64u >> logG, g = lane >> logG;
    // workgroup b serves `steps` consecutive steps per wavefront, wavefronts interleaved; the
    // grid covers the batch once, in index order (see k_reverse_staged)
    const uint64_t stride = (uint64_t)(SKH_BLOCK / 64) * GW;
    uint64_t rec0 = ((uint64_t)blockIdx.x * steps * (SKH_BLOCK / 64) + (threadIdx.x >> 6)) * GW;
    const uint64_t rec_stop = (uint64_t)(blockIdx.x + 1) * steps * (SKH_BLOCK / 64) * GW;
    const uint64_t rec_end = rec_stop < nrec ? rec_stop : nrec;
    if (rec0 >= rec_end) return;
    WaveOffsets wo;
    wo.load(offs, nrec, total, rec0, GW, lane);
    while (true) {
        const uint64_t next0 = rec0 + stride;
        WaveOffsets nwo;
        if (next0 < rec_end) nwo.load(offs, nrec, total, next0, GW, lane);
        const uint64_t rec = rec0 + g;
        const uint64_t s = wo.get(g + 1, GW), e = wo.get(g + 2, GW);
        const uint64_t len = rec < rec_end ? e - s : 0;     // records past the range belong to the next workgroup
        uint32_t acc16[2 * NW];
#pragma unroll
        for (int i = 0; i < 2 * NW; i++) acc16[i] = 0;
        if (len > SKH_LONG_T) {
            if (j == 0) note_long(hint, rec, s, e);      // counted by k_count_tiles; zeros written here
        } else if (len != 0) {
            uint32_t acc8[NW];
#pragma unroll
            for (int i = 0; i < NW; i++) acc8[i] = 0;
            const uint64_t o0 = s & ~15ull;
            const uint32_t sp = (uint32_t)(s - o0);         // bytes of chunk 0 in front of the record
            const uint32_t span = (uint32_t)(e - o0);       // from chunk 0 to the record's end: <= LONG_T + 15
            const uint32_t K = (span + 15) >> 4;
            if (has_neutral) {
                // The vector ALUs, not the memory system, are what this kernel runs out of first, so the
                // loop body is kept to the lookups: every chunk is masked the same way, without a
                // branch -- bytes [lo, hi) of it are the record's, the others are replaced by a byte of
                // no class (for a lane past the record's last chunk that is all of them) -- and the
                // 8-bit fields are widened on a count of rounds, not of chunks.
                uint32_t rounds = 0;
                for (uint32_t kb = 0; kb < K; kb += U * G) {
                    uint4 V[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {            // all loads of the round before the first use:
                        const uint32_t k = kb + j + u * G;   // unconditional (a lane without a chunk re-reads
                        V[u] = ld16(seq + o0 + 16ull * (k < K ? k : 0));   // chunk 0) so they issue back to back
                    }                                        // (skipping the loads of a slot no lane uses on a wave-uniform
                                                             // test: 10 points slower on reads of 100-150 bases, 6-8 beyond)
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const uint32_t k = kb + j + u * G;
                        if (!__any(k < K)) continue;         // wave-uniform: no lane has a chunk in this slot
                        const int rem = (int)span - 16 * (int)k;
                        const uint32_t hi = rem >= 16 ? 16u : (rem > 0 ? (uint32_t)rem : 0u);
                        uint4 m = L.low[hi];
                        if (u == 0 && __any(k == 0 && sp != 0)) {    // wave-uniform: a record starts inside its chunk 0
                            const uint4 ml = L.low[k == 0 ? sp : 0u];   // (sp < span: the record is not empty)
                            m.x ^= ml.x; m.y ^= ml.y; m.z ^= ml.z; m.w ^= ml.w;
                        }
                        uint4 v = V[u];
                        v.x = (v.x & m.x) | (neutral4 & ~m.x);
                        v.y = (v.y & m.y) | (neutral4 & ~m.y);
                        v.z = (v.z & m.z) | (neutral4 & ~m.z);
                        v.w = (v.w & m.w) | (neutral4 & ~m.w);
                        count4<NW>(v.x, L.lut, acc8);
                        count4<NW>(v.y, L.lut, acc8);
                        count4<NW>(v.z, L.lut, acc8);
                        count4<NW>(v.w, L.lut, acc8);
                    }
                    if (++rounds == 15 / U) {                // 15 chunks * 16 bytes = 240 < 256 per field
                        widen<NW>(acc8, acc16);
                        rounds = 0;
                    }
                }
            } else {
                // every byte value has a class: bytes outside the record are skipped one by one
                uint32_t pending = 0;
                for (uint32_t kb = 0; kb < K; kb += U * G) {
#pragma unroll 1
                    for (int u = 0; u < U; u++) {
                        const uint32_t k = kb + j + u * G;
                        if (k < K) {
                            count_chunk<NW>(seq, s, e, o0 + 16ull * k, L, neutral4, false, acc8);
                            if (++pending == 15) {
                                widen<NW>(acc8, acc16);
                                pending = 0;
                            }
                        }
                    }
                }
            }
            widen<NW>(acc8, acc16);
            // a lane sees at most LONG_T bytes and a group LONG_T in total: 16 bits hold it
            for (uint32_t m = 1; m < G; m <<= 1) {
#pragma unroll
                for (int i = 0; i < 2 * NW; i++) acc16[i] += __shfl_xor(acc16[i], m);
            }
        }
        if (rec < rec_end)
            for (uint32_t c = j; c < nclass; c += G) counts[rec * nclass + c] = field16<NW>(acc16, c);
        if (next0 >= rec_end) break;
        rec0 = next0;
        wo = nwo;
    }
}

// Long records: a workgroup owns one aligned run of LONG_RUN bytes of the plane and counts the part of
// every long record inside it.  All chunk loads of a batch (8 per lane = 32 KiB per workgroup) are
// issued before the first table lookup; a record's part of the run is summed in registers and leaves
// as one atomic per class.
template <int NW>
__global__ __launch_bounds__(SKH_BLOCK) void k_count_tiles(
    const uint8_t* __restrict__ seq, const uint64_t* __restrict__ offs, uint64_t nrec, uint64_t total,
    Table256 cls, uint32_t nclass, uint32_t neutral, uint64_t* __restrict__ counts,
    const BatchHint* __restrict__ hint)
{
    if (hint->n_long == 0) return;
    __shared__ CountLds<NW> L;
    __shared__ WalkLds W;
    __shared__ uint32_t s_part[SKH_BLOCK / 64][4 * NW];
    count_lds_init<NW>(L, cls);
    const bool has_neutral = neutral < 256u;
    const uint32_t neutral4 = (neutral & 0xffu) * 0x01010101u;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t t0 = (uint64_t)blockIdx.x * LONG_RUN;
    if (t0 >= total) return;
    const uint64_t t1 = t0 + LONG_RUN;
    constexpr int UB = 8;                                      // chunks per lane in flight
    static_assert(LONG_RUN / SKH_BLOCK <= 512, "16-bit fields: 64 lanes x bytes per lane must stay below 65536");
    for_each_long(offs, nrec, t0, t1 < total ? t1 : total, W, hint, [&](uint64_t r, uint64_t s, uint64_t e) {
        const uint64_t a = s & ~15ull, b = (e + 15) & ~15ull;
        const uint64_t c0 = a > t0 ? a : t0, c1 = b < t1 ? b : t1;
        uint32_t acc8[NW], acc16[2 * NW];
#pragma unroll
        for (int i = 0; i < NW; i++) acc8[i] = 0;
#pragma unroll
        for (int i = 0; i < 2 * NW; i++) acc16[i] = 0;
        for (uint64_t ob = c0 + 16ull * threadIdx.x; ob < c1; ob += 16ull * SKH_BLOCK * UB) {
            uint4 V[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {                     // a lane without a chunk re-reads its first one
                const uint64_t o = ob + 16ull * SKH_BLOCK * u;
                V[u] = ld16(seq + (o < c1 ? o : ob));
            }
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const uint64_t o = ob + 16ull * SKH_BLOCK * u;
                if (o < c1) count_loaded<NW>(V[u], s, e, o, L, neutral4, has_neutral, acc8);
            }
            widen<NW>(acc8, acc16);                            // 8 chunks = 128 bytes per 8-bit field at most
        }
        for (uint32_t m = 1; m < 64; m <<= 1) {                // <= 64 * 512 per field
#pragma unroll
            for (int i = 0; i < 2 * NW; i++) acc16[i] += __shfl_xor(acc16[i], m);
        }
        __syncthreads();                                       // s_part free again
        if (lane < nclass) s_part[wave][lane] = field16<NW>(acc16, lane);
        __syncthreads();
        if (threadIdx.x < nclass) {
            uint64_t sum = 0;
#pragma unroll
            for (int w = 0; w < SKH_BLOCK / 64; w++) sum += s_part[w][threadIdx.x];
            if (sum) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[r * nclass + threadIdx.x]),
                               (unsigned long long)sum);
        }
    });
}

// ------------------------------------------------------------------ letters present
//
// Which byte values occur in each record: 256 bits per record (8 words; bit b & 31 of word b >> 5).
// The alphabet column of fx2tab (alphabetStr, seqkit/cmd/fx2tab.go:167-180) prints the set bits in
// ascending order.  Same decomposition as the counting kernels: lane groups for records up to
// LONG_T bytes, aligned windows of the plane with one atomic OR per word for longer ones.

__device__ __forceinline__ void set_bits16(uint4 v, int lo, int hi, uint32_t* m)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k >= lo && k < hi) {
            const uint32_t b = (w[k >> 2] >> (8 * (k & 3))) & 0xffu;
            const uint32_t bit = 1u << (b & 31u), wi = b >> 5;
#pragma unroll
            for (int i = 0; i < 8; i++) m[i] |= wi == (uint32_t)i ? bit : 0u;
        }
    }
}

__global__ __launch_bounds__(SKH_BLOCK) void k_letters_short(const uint8_t* __restrict__ seq,
                                                             const uint64_t* __restrict__ offs, uint64_t nrec,
                                                             uint64_t total, int logG, uint32_t* __restrict__ present,
                                                             BatchHint* __restrict__ hint)
{
    const uint32_t G = 1u << logG, j = threadIdx.x & (G - 1);
    const uint32_t lane = threadIdx.x & 63, GW = 64u >> logG, g = lane >> logG;
    const uint64_t stride = (uint64_t)gridDim.x * (SKH_BLOCK / 64) * GW;
    for (uint64_t rec0 = ((uint64_t)blockIdx.x * (SKH_BLOCK / 64) + (threadIdx.x >> 6)) * GW; rec0 < nrec; rec0 += stride) {
        WaveOffsets wo;
        wo.load(offs, nrec, total, rec0, GW, lane);
        const uint64_t rec = rec0 + g;
        const uint64_t s = wo.get(g + 1, GW), e = wo.get(g + 2, GW);
        const uint64_t len = e - s;
        uint32_t m[8];
#pragma unroll
        for (int i = 0; i < 8; i++) m[i] = 0;
        if (len > SKH_LONG_T) {
            if (j == 0) note_long(hint, rec, s, e);       // k_letters_tiles ORs into the zeros written here
        } else if (len != 0) {
            const uint64_t o0 = s & ~15ull;
            const uint32_t K = (uint32_t)((e - o0 + 15) >> 4);
            for (uint32_t k = j; k < K; k += G) {
                const uint64_t o = o0 + 16ull * k;
                const int lo = s > o ? (int)(s - o) : 0;
                const int hi = e < o + 16 ? (int)(e - o) : 16;
                set_bits16(ld16(seq + o), lo, hi, m);
            }
            for (uint32_t d = 1; d < G; d <<= 1) {
#pragma unroll
                for (int i = 0; i < 8; i++) m[i] |= __shfl_xor(m[i], d);
            }
        }
        if (rec < nrec) {
            for (uint32_t i = j; i < 8; i += G) {
                uint32_t v = m[0];
#pragma unroll
                for (int q = 1; q < 8; q++) v = i == (uint32_t)q ? m[q] : v;
                present[rec * 8 + i] = v;
            }
        }
    }
}

__global__ __launch_bounds__(SKH_BLOCK) void k_letters_tiles(const uint8_t* __restrict__ seq,
                                                             const uint64_t* __restrict__ offs, uint64_t nrec,
                                                             uint64_t total, uint32_t* __restrict__ present,
                                                             const BatchHint* __restrict__ hint)
{
    if (hint->n_long == 0) return;
    __shared__ uint32_t s_m[8];
    __shared__ WalkLds W;
    const uint64_t t0 = (uint64_t)blockIdx.x * LONG_RUN;
    if (t0 >= total) return;
    const uint64_t t1 = t0 + LONG_RUN;
    for_each_long(offs, nrec, t0, t1 < total ? t1 : total, W, hint, [&](uint64_t r, uint64_t s, uint64_t e) {
        const uint64_t a = s & ~15ull, b = (e + 15) & ~15ull;
        const uint64_t c0 = a > t0 ? a : t0, c1 = b < t1 ? b : t1;
        uint32_t m[8];
#pragma unroll
        for (int i = 0; i < 8; i++) m[i] = 0;
        for (uint64_t o = c0 + 16ull * threadIdx.x; o < c1; o += 16ull * SKH_BLOCK) {
            const int lo = s > o ? (int)(s - o) : 0;
            const int hi = e < o + 16 ? (int)(e - o) : 16;
            set_bits16(ld16(seq + o), lo, hi, m);
        }
        for (uint32_t d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) m[i] |= __shfl_xor(m[i], d);
        }
        __syncthreads();                                   // s_m free again
        if (threadIdx.x < 8) s_m[threadIdx.x] = 0;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) if (m[i]) atomicOr(&s_m[i], m[i]);
        }
        __syncthreads();
        if (threadIdx.x < 8 && s_m[threadIdx.x]) atomicOr(&present[r * 8 + threadIdx.x], s_m[threadIdx.x]);
    });
}

// ------------------------------------------------------------------ stats totals

__global__ void k_stats_init(StatsPartial* out, uint64_t nrec, uint64_t total)
{
    out->num = (int64_t)nrec;
    out->len_sum = (int64_t)total;
    out->gap_sum = 0;
    out->len_min = INT64_MAX;
    out->len_max = 0;
}

// gap_sum over the whole plane (record boundaries are irrelevant to a file total)
// and min/max of the record lengths from the offsets, in one grid that covers the plane once, in
// address order: workgroup b reads `steps` runs of STATS_U * SKH_BLOCK consecutive chunks (the loads of a run
// issued before its first table lookup) and its share `rpb` of the records, then reduces once.
constexpr int STATS_U = 4;
constexpr uint32_t STATS_SLOTS = 512;      // = skh::STATS_SLOT_WORDS: partial gap sums of a launch
static_assert(STATS_SLOTS == STATS_SLOT_WORDS, "slot count");

__global__ __launch_bounds__(SKH_BLOCK) void k_stats(const uint8_t* __restrict__ seq,
                                                     const uint64_t* __restrict__ offs, uint64_t nrec,
                                                     uint64_t total, uint64_t rpb, uint32_t steps, Table256 is_gap,
                                                     StatsPartial* __restrict__ out,
                                                     unsigned long long* __restrict__ slots)
{
    __shared__ uint8_t s_gap[256];
    __shared__ unsigned long long s_sum[SKH_BLOCK / 64];
    __shared__ long long s_min[SKH_BLOCK / 64], s_max[SKH_BLOCK / 64];
    s_gap[threadIdx.x] = is_gap.v[threadIdx.x] ? 1 : 0;
    __syncthreads();
    uint32_t n = 0;
    for (uint32_t it = 0; it < steps; it++) {
        const uint64_t base = (((uint64_t)blockIdx.x * steps + it) * STATS_U * SKH_BLOCK + threadIdx.x) * 16;
        if (base >= total) break;
        uint4 v[STATS_U];
#pragma unroll
        for (int u = 0; u < STATS_U; u++) {
            const uint64_t o = base + (uint64_t)u * SKH_BLOCK * 16;
            v[u] = o < total ? ld16(seq + o) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < STATS_U; u++) {
            const uint64_t o = base + (uint64_t)u * SKH_BLOCK * 16;
            const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            if (o + 16 <= total) {
#pragma unroll
                for (int k = 0; k < 16; k++) n += s_gap[(w[k >> 2] >> (8 * (k & 3))) & 0xffu];
            } else if (o < total) {
                const int hi = (int)(total - o);
#pragma unroll
                for (int k = 0; k < 16; k++)
                    if (k < hi) n += s_gap[(w[k >> 2] >> (8 * (k & 3))) & 0xffu];
            }
        }
    }
    unsigned long long gaps = n;
    long long lmin = INT64_MAX, lmax = 0;
    const uint64_t r0 = (uint64_t)blockIdx.x * rpb;
    const uint64_t r1 = r0 + rpb < nrec ? r0 + rpb : nrec;
    for (uint64_t r = r0 + threadIdx.x; r < r1; r += SKH_BLOCK) {
        const long long len = (long long)(offs[r + 1] - offs[r]);
        lmin = len < lmin ? len : lmin;
        lmax = len > lmax ? len : lmax;
    }
    for (uint32_t m = 1; m < 64; m <<= 1) {
        gaps += __shfl_xor(gaps, m);
        const long long a = __shfl_xor(lmin, m), b = __shfl_xor(lmax, m);
        lmin = a < lmin ? a : lmin;
        lmax = b > lmax ? b : lmax;
    }
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[wave] = gaps; s_min[wave] = lmin; s_max[wave] = lmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SKH_BLOCK / 64; w++) {
            gaps += s_sum[w];
            lmin = s_min[w] < lmin ? s_min[w] : lmin;
            lmax = s_max[w] > lmax ? s_max[w] : lmax;
        }
        // the grid is large: only a workgroup that improves on the totals so far touches them
        // (atomics on one address are served one after the other)
        // (the gap count goes to one of STATS_SLOTS words, summed by k_stats_finish: with gaps in the
        // data every workgroup has something to add, and one word would take them one by one)
        if (gaps) atomicAdd(&slots[blockIdx.x & (STATS_SLOTS - 1)], gaps);
        if (lmin < *reinterpret_cast<volatile long long*>(&out->len_min))
            atomicMin(reinterpret_cast<long long*>(&out->len_min), lmin);
        if (lmax > *reinterpret_cast<volatile long long*>(&out->len_max))
            atomicMax(reinterpret_cast<long long*>(&out->len_max), lmax);
    }
}

__global__ __launch_bounds__(STATS_SLOTS) void k_stats_finish(const unsigned long long* __restrict__ slots,
                                                              StatsPartial* __restrict__ out)
{
    __shared__ unsigned long long s_w[STATS_SLOTS / 64];
    unsigned long long v = slots[threadIdx.x];
    for (uint32_t m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (uint32_t w = 0; w < STATS_SLOTS / 64; w++) t += s_w[w];
        out->gap_sum = (int64_t)t;
    }
}

// ------------------------------------------------------------------ validation
//
// seq.NewSeq's check with seq.ValidateSeq set (reached through the reader, seqkit/cmd/seq.go:
// 84-93,115-127): the first `limit` bytes of every record (all when limit == 0) must be members
// of the alphabet.  The answer is the first offending byte in plane order, i.e. of the first
// offending record.  One streaming pass: sixteen table lookups per chunk; only a chunk that
// holds a non-member byte looks up its record (binary search of the offsets).

__device__ __forceinline__ uint64_t record_of(const uint64_t* __restrict__ offs, uint64_t nrec, uint64_t p)
{
    uint64_t lo = 0, hi = nrec;                  // last r with offs[r] <= p; p < offs[nrec]
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (offs[mid] <= p) lo = mid; else hi = mid; }
    return lo;
}

__global__ void k_validate_init(ValidateResult* out)
{
    out->record = ~0ull;                          // holds the first offending plane position during the scan
    out->offset = 0; out->letter = 0; out->found = 0;
}

__global__ __launch_bounds__(SKH_BLOCK) void k_validate_scan(const uint8_t* __restrict__ seq,
                                                             const uint64_t* __restrict__ offs, uint64_t nrec,
                                                             uint64_t total, Table256 valid, uint64_t limit,
                                                             ValidateResult* __restrict__ out)
{
    __shared__ uint8_t s_bad[256];
    s_bad[threadIdx.x] = valid.v[threadIdx.x] ? 0 : 1;
    __syncthreads();
    constexpr int U = 4;
    const uint64_t base = ((uint64_t)blockIdx.x * U * SKH_BLOCK + threadIdx.x) * 16;
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint64_t o = base + (uint64_t)u * SKH_BLOCK * 16;
        v[u] = o < total ? ld16(seq + o) : make_uint4(0, 0, 0, 0);
    }
    unsigned long long* first = reinterpret_cast<unsigned long long*>(&out->record);
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint64_t o = base + (uint64_t)u * SKH_BLOCK * 16;
        if (o >= total) continue;
        const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        uint32_t nbad = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) nbad += s_bad[(w[k >> 2] >> (8 * (k & 3))) & 0xffu];
        if (nbad == 0) continue;                  // bytes past `total` are re-checked below
        const int hi = total - o < 16 ? (int)(total - o) : 16;
#pragma unroll 1
        for (int k = 0; k < hi; k++) {
            if (!s_bad[(w[k >> 2] >> (8 * (k & 3))) & 0xffu]) continue;
            const uint64_t p = o + (uint64_t)k;
            if (p >= *reinterpret_cast<volatile unsigned long long*>(first)) break;     // an earlier one is known
            if (limit != 0 && p - offs[record_of(offs, nrec, p)] >= limit) continue;
            atomicMin(first, (unsigned long long)p);
            break;
        }
    }
}

__global__ void k_validate_finish(const uint8_t* __restrict__ seq, const uint64_t* __restrict__ offs, uint64_t nrec,
                                  ValidateResult* out)
{
    const uint64_t p = out->record;
    if (p == ~0ull) { out->record = 0; return; }
    const uint64_t r = record_of(offs, nrec, p);
    out->record = r; out->offset = p - offs[r]; out->letter = seq[p]; out->found = 1;
}

// ------------------------------------------------------------------ launchers

static inline uint32_t grid_for(uint64_t work_items, uint64_t per_block, uint32_t cap)
{
    uint64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (uint32_t)g;
}

// Grid cap of a grid-stride kernel: exactly the workgroups the chip holds at once (CUs x
// resident workgroups per CU).  The kernels split their work statically over the grid, so
// one workgroup more per CU than fits would run as a second, nearly empty round.
template <typename K>
static uint32_t resident_cap(K kernel, const LaunchCfg& cfg)
{
    static std::mutex mu;
    static std::unordered_map<const void*, int> cache;
    const void* key = reinterpret_cast<const void*>(kernel);
    int occ;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find(key);
        if (it == cache.end()) {
            int o = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kernel, SKH_BLOCK, 0) != hipSuccess || o < 1) o = 4;
            if (o > 8) o = 8;
            it = cache.emplace(key, o).first;
        }
        occ = it->second;
    }
    uint64_t g = (uint64_t)cfg.n_cu * (uint64_t)occ;
    if (cfg.max_grid && g > cfg.max_grid) g = cfg.max_grid;
    return (uint32_t)(g ? g : 1);
}

template <bool COMP>
static hipError_t launch_reverse_span_t(const LaunchCfg& cfg, const RevPlanes& P, uint64_t nrec,
                                        const Table256& lut, hipStream_t st)
{
    const uint64_t nch = (P.total + 15) >> 4;
    if (cfg.span_u == 2) {
        const uint32_t g = grid_for(nch, 2 * SKH_BLOCK, resident_cap(k_reverse_span<COMP, 2>, cfg));
        hipLaunchKernelGGL((k_reverse_span<COMP, 2>), dim3(g), dim3(SKH_BLOCK), 0, st, P, nrec, lut);
    } else {
        const uint32_t g = grid_for(nch, 4 * SKH_BLOCK, resident_cap(k_reverse_span<COMP, 4>, cfg));
        hipLaunchKernelGGL((k_reverse_span<COMP, 4>), dim3(g), dim3(SKH_BLOCK), 0, st, P, nrec, lut);
    }
    return hipGetLastError();
}

// workgroups of a long-record kernel: one per run of the plane
static inline uint64_t long_runs(uint64_t total) { return (total + LONG_RUN - 1) / LONG_RUN; }

static hipError_t launch_prepare(const uint64_t* offs, uint64_t nrec, BatchHint* hint, hipStream_t st)
{
    hipLaunchKernelGGL(k_prepare, dim3(1), dim3(SKH_BLOCK), 0, st, offs, nrec, hint);
    return hipGetLastError();
}

template <bool COMP, bool FASTQ>
static hipError_t launch_reverse_t(const LaunchCfg& cfg, const uint8_t* in, uint8_t* out,
                                   const uint8_t* qin, uint8_t* qout, const uint64_t* offs,
                                   uint64_t nrec, uint64_t total, const Table256& lut,
                                   BatchHint* hint, hipStream_t st)
{
    const RevPlanes P{in, out, qin, qout, offs, total};
    if (cfg.span_u && !FASTQ) return launch_reverse_span_t<COMP>(cfg, P, nrec, lut, st);
    if (cfg.flat && !FASTQ) {
        // steps per wavefront: 1 to 3 measure within two points of each other on every read length (100-base
        // reads best with 1, ragged reads with 2 to 3), more are slower
        const uint32_t steps = cfg.flat_steps ? cfg.flat_steps : 2;
        const uint32_t R = flat_records_per_step(nrec ? total / nrec : 0);
        const uint64_t nsteps = (nrec + R - 1) / R, per_block = (uint64_t)steps * (SKH_BLOCK / 64);
        const uint64_t g1 = (nsteps + per_block - 1) / per_block;
        if (g1 > 0x7fffffffull) return hipErrorInvalidValue;
        if (g1) hipLaunchKernelGGL((k_reverse_flat<COMP>), dim3((uint32_t)g1), dim3(SKH_BLOCK), 0, st, P.in, P.out, P.offs,
                                   P.total, nrec, lut, steps, hint, cfg.use_hint);
    } else if (cfg.staged && !FASTQ) {
        const uint32_t steps = cfg.staged_steps ? cfg.staged_steps : 2;   // steps per wavefront
        const uint64_t per_block = (uint64_t)steps * (SKH_BLOCK >> cfg.logG);
        const uint64_t g1 = (nrec + per_block - 1) / per_block;
        if (g1 > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL((k_reverse_staged<COMP, false>), dim3((uint32_t)g1), dim3(SKH_BLOCK), 0, st, P.in, P.out, P.offs,
                           P.total, nrec, lut, cfg.logG, steps, hint, cfg.use_hint);
    } else {
        // up to 8 steps and about 128 KiB of the plane per workgroup (measured on records of 350 to 16000
        // bases: 8 steps 1-2 points ahead of 2-4 up to 1000 bases, no difference beyond)
        const uint64_t step_bytes = (uint64_t)(SKH_BLOCK >> cfg.logG) * (nrec ? total / nrec : 0) + 1;
        uint32_t steps = cfg.short_steps ? cfg.short_steps : (uint32_t)(131072 / step_bytes);
        steps = steps < 1 ? 1 : (steps > 8 ? 8 : steps);
        const uint64_t per_block = (uint64_t)steps * (SKH_BLOCK >> cfg.logG);
        const uint64_t g1 = (nrec + per_block - 1) / per_block;
        if (g1 > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL((k_reverse_short<COMP, FASTQ>), dim3((uint32_t)g1), dim3(SKH_BLOCK), 0, st, P, nrec, lut,
                           cfg.logG, steps, hint, cfg.use_hint);
    }
    const uint64_t g2 = long_runs(total);
    if (g2 > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_reverse_tiles<COMP, FASTQ>), dim3((uint32_t)g2), dim3(SKH_BLOCK), 0, st, P, nrec, lut,
                       hint);
    return hipGetLastError();
}

static hipError_t reverse_planes(const LaunchCfg& cfg, const uint8_t* in, uint8_t* out, const uint8_t* qin,
                                 uint8_t* qout, const uint64_t* offs, uint64_t nrec, uint64_t total,
                                 const Table256* lut, BatchHint* hint, hipStream_t st)
{
    static const Table256 ident = [] { Table256 t; for (int i = 0; i < 256; i++) t.v[i] = (uint8_t)i; return t; }();
    bool fastq = qin != nullptr;
    if (fastq && (!cfg.fuse_fastq || cfg.span_u)) {
        // the two planes as two launches: half the registers per lane, twice the resident
        // wavefronts; the offsets are read twice, which is cheaper than the lost occupancy
        // (the long-record counter keeps counting: only "zero or not" is looked at)
        hipError_t e = reverse_planes(cfg, qin, qout, nullptr, nullptr, offs, nrec, total, nullptr, hint, st);
        if (e != hipSuccess) return e;
        // (the short kernel of the second plane lists the long records again: from an empty list)
        if (!cfg.span_u && (e = launch_prepare(offs, nrec, hint, st)) != hipSuccess) return e;
        fastq = false;
        qin = nullptr;
        qout = nullptr;
    }
    if (lut) {
        return fastq ? launch_reverse_t<true, true>(cfg, in, out, qin, qout, offs, nrec, total, *lut, hint, st)
                     : launch_reverse_t<true, false>(cfg, in, out, qin, qout, offs, nrec, total, *lut, hint, st);
    }
    return fastq ? launch_reverse_t<false, true>(cfg, in, out, qin, qout, offs, nrec, total, ident, hint, st)
                 : launch_reverse_t<false, false>(cfg, in, out, qin, qout, offs, nrec, total, ident, hint, st);
}

hipError_t launch_reverse(const LaunchCfg& cfg, const uint8_t* in, uint8_t* out, const uint8_t* qin,
                          uint8_t* qout, const uint64_t* offs, uint64_t nrec, uint64_t total,
                          const Table256* lut, BatchHint* hint, hipStream_t st)
{
    if (!cfg.span_u) {                       // the span kernel serves every record itself
        hipError_t e = launch_prepare(offs, nrec, hint, st);
        if (e != hipSuccess) return e;
    }
    return reverse_planes(cfg, in, out, qin, qout, offs, nrec, total, lut, hint, st);
}

hipError_t launch_map_plane(const LaunchCfg& cfg, const uint8_t* in, uint8_t* out, uint64_t total,
                            const Table256& lut, hipStream_t st)
{
    (void)cfg;
    const uint64_t per_block = (uint64_t)MAP_U * SKH_BLOCK;
    const uint64_t g = (((total + 15) >> 4) + per_block - 1) / per_block;
    if (g > 0x7fffffffull) return hipErrorInvalidValue;
    if (g) hipLaunchKernelGGL(k_map_plane, dim3((uint32_t)g), dim3(SKH_BLOCK), 0, st, in, out, total, lut);
    return hipGetLastError();
}

template <bool COMP>
static hipError_t launch_reverse_inplace_t(const LaunchCfg& cfg, uint8_t* plane, const uint64_t* offs, uint64_t nrec,
                                           uint64_t total, const Table256& lut, BatchHint* hint, hipStream_t st)
{
    if (cfg.flat_inplace) {
        const uint32_t steps = cfg.flat_steps ? cfg.flat_steps : 2;
        const uint32_t R = flat_records_per_step(nrec ? total / nrec : 0);
        const uint64_t nsteps = (nrec + R - 1) / R, per_block = (uint64_t)steps * (SKH_BLOCK / 64);
        const uint64_t g1 = (nsteps + per_block - 1) / per_block;
        if (g1 > 0x7fffffffull) return hipErrorInvalidValue;
        if (g1) hipLaunchKernelGGL((k_reverse_flat_inplace<COMP>), dim3((uint32_t)g1), dim3(SKH_BLOCK), 0, st, plane, offs, total,
                                   nrec, lut, steps, hint, cfg.use_hint);
    } else if (cfg.staged) {
        // short reads: the LDS-staged kernel with one plane (aligned row loads and stores)
        const uint32_t steps = cfg.staged_steps ? cfg.staged_steps : 2;
        const uint64_t per_block = (uint64_t)steps * (SKH_BLOCK >> cfg.logG);
        const uint64_t g1 = (nrec + per_block - 1) / per_block;
        if (g1 > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL((k_reverse_staged<COMP, true>), dim3((uint32_t)g1), dim3(SKH_BLOCK), 0, st, plane, plane, offs,
                           total, nrec, lut, cfg.logG, steps, hint, cfg.use_hint);
    } else {
        const uint32_t g1 = grid_for(nrec, SKH_BLOCK >> cfg.logG, resident_cap(k_reverse_inplace_short<COMP>, cfg));
        hipLaunchKernelGGL((k_reverse_inplace_short<COMP>), dim3(g1), dim3(SKH_BLOCK), 0, st, plane, offs, total, nrec, lut,
                           cfg.logG, hint, cfg.use_hint);
    }
    // a workgroup moves the window pairs whose front window lies in its run, twice the run's bytes (the host
    // chooses the run: see cfg_for)
    const uint64_t run = SKH_TILE << cfg.inplace_run;
    const uint64_t g2 = (total + run - 1) / run;
    if (g2 > 0x7fffffffull) return hipErrorInvalidValue;
    switch (cfg.inplace_run) {
    case 0: hipLaunchKernelGGL((k_reverse_inplace_long<COMP, SKH_TILE>), dim3((uint32_t)g2), dim3(SKH_BLOCK), 0, st, plane, offs, total, nrec, lut, hint); break;
    case 2: hipLaunchKernelGGL((k_reverse_inplace_long<COMP, 4 * SKH_TILE>), dim3((uint32_t)g2), dim3(SKH_BLOCK), 0, st, plane, offs, total, nrec, lut, hint); break;
    default: hipLaunchKernelGGL((k_reverse_inplace_long<COMP, 2 * SKH_TILE>), dim3((uint32_t)g2), dim3(SKH_BLOCK), 0, st, plane, offs, total, nrec, lut, hint); break;
    }
    return hipGetLastError();
}

hipError_t launch_reverse_inplace(const LaunchCfg& cfg, uint8_t* plane, const uint64_t* offs, uint64_t nrec,
                                  uint64_t total, const Table256* lut, BatchHint* hint, hipStream_t st)
{
    if (nrec == 0 || total == 0) return hipSuccess;
    hipError_t e = launch_prepare(offs, nrec, hint, st);
    if (e != hipSuccess) return e;
    if (lut) return launch_reverse_inplace_t<true>(cfg, plane, offs, nrec, total, *lut, hint, st);
    Table256 none{};
    return launch_reverse_inplace_t<false>(cfg, plane, offs, nrec, total, none, hint, st);
}

hipError_t launch_count(const LaunchCfg& cfg, const uint8_t* seq, const uint64_t* offs, uint64_t nrec,
                        uint64_t total, const Table256& cls, uint32_t nclass, uint32_t neutral,
                        uint64_t* counts, BatchHint* hint, hipStream_t st)
{
    const uint32_t steps = cfg.count_steps ? cfg.count_steps : 3;     // steps per wavefront (2-3 measured 1-4 % ahead of 4 and of 1 on reads of 100-250 bases)
    const uint64_t per_block = (uint64_t)steps * (SKH_BLOCK >> cfg.logG_count);
    const uint64_t g1 = (nrec + per_block - 1) / per_block;
    const uint64_t g2 = long_runs(total);
    if (g1 > 0x7fffffffull || g2 > 0x7fffffffull) return hipErrorInvalidValue;
    hipError_t e = launch_prepare(offs, nrec, hint, st);
    if (e != hipSuccess) return e;
    auto short_kernel = [&](auto nw, auto u) {
        hipLaunchKernelGGL((k_count_short<decltype(nw)::value, decltype(u)::value>), dim3((uint32_t)g1), dim3(SKH_BLOCK), 0, st, seq,
                           offs, nrec, total, cls, nclass, cfg.logG_count, steps, neutral, counts, hint, cfg.use_hint);
    };
    auto with_u = [&](auto nw) {
        switch (cfg.count_u) {
        case 4: short_kernel(nw, std::integral_constant<int, 4>{}); break;
        case 5: short_kernel(nw, std::integral_constant<int, 5>{}); break;
        default: short_kernel(nw, std::integral_constant<int, 3>{}); break;
        }
    };
    if (nclass <= 4) {
        with_u(std::integral_constant<int, 1>{});
        if (g2) hipLaunchKernelGGL((k_count_tiles<1>), dim3((uint32_t)g2), dim3(SKH_BLOCK), 0, st, seq, offs, nrec, total,
                                   cls, nclass, neutral, counts, hint);
    } else {
        with_u(std::integral_constant<int, 2>{});
        if (g2) hipLaunchKernelGGL((k_count_tiles<2>), dim3((uint32_t)g2), dim3(SKH_BLOCK), 0, st, seq, offs, nrec, total,
                                   cls, nclass, neutral, counts, hint);
    }
    return hipGetLastError();
}

hipError_t launch_letters(const LaunchCfg& cfg, const uint8_t* seq, const uint64_t* offs, uint64_t nrec, uint64_t total,
                          uint32_t* present, BatchHint* hint, hipStream_t st)
{
    const uint64_t g2 = long_runs(total);
    if (g2 > 0x7fffffffull) return hipErrorInvalidValue;
    hipError_t e = launch_prepare(offs, nrec, hint, st);
    if (e != hipSuccess) return e;
    const uint32_t g1 = grid_for(nrec, SKH_BLOCK >> cfg.logG, cfg.n_cu * 8);
    hipLaunchKernelGGL(k_letters_short, dim3(g1), dim3(SKH_BLOCK), 0, st, seq, offs, nrec, total, cfg.logG, present, hint);
    if (g2) hipLaunchKernelGGL(k_letters_tiles, dim3((uint32_t)g2), dim3(SKH_BLOCK), 0, st, seq, offs, nrec, total, present, hint);
    return hipGetLastError();
}

hipError_t launch_stats(const LaunchCfg& cfg, const uint8_t* seq, const uint64_t* offs, uint64_t nrec,
                        uint64_t total, const Table256& is_gap, StatsPartial* out, uint64_t* slots, hipStream_t st)
{
    hipError_t e0 = hipMemsetAsync(slots, 0, STATS_SLOTS * 8, st);
    if (e0 != hipSuccess) return e0;
    hipLaunchKernelGGL(k_stats_init, dim3(1), dim3(1), 0, st, out, nrec, total);
    const uint32_t steps = cfg.stats_steps > 0 ? (uint32_t)cfg.stats_steps : 8u;   // measured: 8-16 are 1-5 points of HBM peak ahead of 2-4
    const uint64_t per_block = (uint64_t)STATS_U * SKH_BLOCK * steps;
    uint64_t g = (((total + 15) >> 4) + per_block - 1) / per_block;
    const uint64_t g_rec = (nrec + 16ull * SKH_BLOCK - 1) / (16ull * SKH_BLOCK);
    if (g < g_rec) g = g_rec;
    if (g < 1) g = 1;
    if (g > 0x7fffffffull) return hipErrorInvalidValue;
    const uint64_t rpb = (nrec + g - 1) / g;
    hipLaunchKernelGGL(k_stats, dim3((uint32_t)g), dim3(SKH_BLOCK), 0, st, seq, offs, nrec, total, rpb, steps, is_gap, out,
                       reinterpret_cast<unsigned long long*>(slots));
    hipLaunchKernelGGL(k_stats_finish, dim3(1), dim3(STATS_SLOTS), 0, st, reinterpret_cast<const unsigned long long*>(slots), out);
    return hipGetLastError();
}

hipError_t launch_validate(const LaunchCfg& cfg, const uint8_t* seq, const uint64_t* offs, uint64_t nrec,
                           uint64_t total, const Table256& valid, uint64_t limit, ValidateResult* out, hipStream_t st)
{
    (void)cfg;
    hipLaunchKernelGGL(k_validate_init, dim3(1), dim3(1), 0, st, out);
    if (nrec != 0 && total != 0) {
        const uint64_t per_block = 4ull * SKH_BLOCK * 16;
        const uint64_t g = (total + per_block - 1) / per_block;
        if (g > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL(k_validate_scan, dim3((uint32_t)g), dim3(SKH_BLOCK), 0, st, seq, offs, nrec, total, valid, limit,
                           out);
    }
    hipLaunchKernelGGL(k_validate_finish, dim3(1), dim3(1), 0, st, seq, offs, nrec, out);
    return hipGetLastError();
}

}  // namespace skh
