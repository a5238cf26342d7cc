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

#include "kernels.h"

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

__device__ __forceinline__ uint4 ld16(const uint8_t* p)
{
    return *reinterpret_cast<const uint4*>(p);
}

// Index of the record that contains byte `pos` (offs[r] <= pos < offs[r+1]), found
// by a 64-ary search in which every lane of the wavefront probes one splitter per
// round: log64(nrec) dependent memory round trips instead of log2(nrec).
// Requires pos < offs[nrec]; all lanes must call it together with the same pos.
__device__ __forceinline__ uint64_t find_record(const uint64_t* __restrict__ offs, uint64_t nrec,
                                                uint64_t pos)
{
    uint64_t lo = 0, hi = nrec;   // invariant: offs[lo] <= pos < offs[hi]
    const uint32_t lane = threadIdx.x & 63;
    while (hi - lo > 1) {
        const uint64_t span = hi - lo;
        const uint64_t step = (span + 63) >> 6;
        uint64_t idx = lo + (uint64_t)(lane + 1) * step;
        if (idx > hi) idx = hi;
        const bool above = offs[idx] > pos;           // true for the last lanes (idx == hi)
        const unsigned long long m = __ballot(above);
        const uint32_t f = (uint32_t)__builtin_ctzll(m);   // m != 0: lane 63 probes hi
        uint64_t nhi = lo + (uint64_t)(f + 1) * step;
        if (nhi > hi) nhi = hi;
        lo = lo + (uint64_t)f * step;
        hi = nhi;
    }
    return lo;
}

// ------------------------------------------------------------------ reverse (+complement)

// Produces the 16 output bytes of the aligned output chunk at plane offset `o` of
// the record [s, e): out[p] = in[s + e - 1 - p].  The mirrored source window
// [q, q+16), q = s + e - 16 - o, is misaligned by delta = (s + e) mod 16, the same
// for every chunk of the record; it is cut out of the two aligned chunks that
// cover it with a dword barrel shift and four v_perm_b32 (which also reverse the
// bytes).  Source chunks that hold no byte of the record are not loaded.
__device__ __forceinline__ uint4 gather_reversed(const uint8_t* __restrict__ in, uint64_t s,
                                                 uint64_t e, uint64_t o)
{
    const int64_t q = (int64_t)(s + e) - 16 - (int64_t)o;
    const uint32_t delta = (uint32_t)q & 15u;
    const int64_t q0 = q - (int64_t)delta;
    uint4 A = make_uint4(0, 0, 0, 0), B = make_uint4(0, 0, 0, 0);
    if (q0 + 16 > (int64_t)s && q0 < (int64_t)e) A = ld16(in + q0);
    if (delta != 0 && q0 + 32 > (int64_t)s && q0 + 16 < (int64_t)e) B = ld16(in + q0 + 16);

    const bool h = (delta & 8u) != 0, l = (delta & 4u) != 0;
    const uint32_t T0 = h ? A.z : A.x, T1 = h ? A.w : A.y, T2 = h ? B.x : A.z,
                   T3 = h ? B.y : A.w, T4 = h ? B.z : B.x, T5 = h ? B.w : B.y;
    const uint32_t V0 = l ? T1 : T0, V1 = l ? T2 : T1, V2 = l ? T3 : T2,
                   V3 = l ? T4 : T3, V4 = l ? T5 : T4;
    const uint32_t sel = 0x00010203u + (delta & 3u) * 0x01010101u;
    uint4 r;
    r.x = __builtin_amdgcn_perm(V4, V3, sel);
    r.y = __builtin_amdgcn_perm(V3, V2, sel);
    r.z = __builtin_amdgcn_perm(V2, V1, sel);
    r.w = __builtin_amdgcn_perm(V1, V0, sel);
    return r;
}

__device__ __forceinline__ uint32_t map4(uint32_t w, const uint8_t* s_lut)
{
    return (uint32_t)s_lut[w & 0xffu] | ((uint32_t)s_lut[(w >> 8) & 0xffu] << 8) |
           ((uint32_t)s_lut[(w >> 16) & 0xffu] << 16) | ((uint32_t)s_lut[w >> 24] << 24);
}

__device__ __forceinline__ uint4 map16(uint4 v, const uint8_t* s_lut)
{
    return make_uint4(map4(v.x, s_lut), map4(v.y, s_lut), map4(v.z, s_lut), map4(v.w, s_lut));
}

// Stores the bytes of chunk `o` that belong to the record [s, e).  Interior chunks
// are one 16-byte store; the first and last chunk of a record share their 16 bytes
// with the neighbouring records, so only the owned dwords / bytes are written.
__device__ __forceinline__ void store_owned(uint8_t* __restrict__ out, uint64_t s, uint64_t e,
                                            uint64_t o, uint4 v)
{
    if (o >= s && o + 16 <= e) {
        *reinterpret_cast<uint4*>(out + o) = v;
        return;
    }
    const int lo = s > o ? (int)(s - o) : 0;                 // first owned byte of the chunk
    const int hi = e < o + 16 ? (int)(e - o) : 16;           // one past the last owned byte
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int m = 0; m < 4; m++) {
        if (4 * m >= lo && 4 * m + 4 <= hi) {
            *reinterpret_cast<uint32_t*>(out + o + 4 * m) = w[m];
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int k = 4 * m + t;
                if (k >= lo && k < hi) out[o + k] = (uint8_t)(w[m] >> (8 * t));
            }
        }
    }
}

template <bool COMP, bool FASTQ>
__device__ __forceinline__ void reverse_chunk(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                              const uint8_t* __restrict__ qin, uint8_t* __restrict__ qout,
                                              uint64_t s, uint64_t e, uint64_t o, const uint8_t* s_lut)
{
    uint4 r = gather_reversed(in, s, e, o);
    uint4 rq;
    if (FASTQ) rq = gather_reversed(qin, s, e, o);
    if (COMP) r = map16(r, s_lut);
    store_owned(out, s, e, o, r);
    if (FASTQ) store_owned(qout, s, e, o, rq);
}

// Short records: a group of 2^logG lanes per record, lane j takes output chunks
// j, j+G, ... of the record.  Nothing crosses lanes, so any G works.
template <bool COMP, bool FASTQ>
__global__ __launch_bounds__(SKH_BLOCK) void k_reverse_short(
    const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const uint8_t* __restrict__ qin,
    uint8_t* __restrict__ qout, const uint64_t* __restrict__ offs, uint64_t nrec, Table256 lut,
    int logG, uint32_t* __restrict__ n_long)
{
    __shared__ uint8_t s_lut[256];
    if (COMP) {
        s_lut[threadIdx.x] = lut.v[threadIdx.x];
        __syncthreads();
    }
    const uint32_t G = 1u << logG, j = threadIdx.x & (G - 1);
    const uint64_t per_block = SKH_BLOCK >> logG;
    const uint64_t stride = (uint64_t)gridDim.x * per_block;
    uint64_t rec = (uint64_t)xcd_remap(blockIdx.x, gridDim.x) * per_block + (threadIdx.x >> logG);
    for (; rec < nrec; rec += stride) {
        const uint64_t s = offs[rec], e = offs[rec + 1];
        const uint64_t L = e - s;
        if (L > SKH_LONG_T) {
            if (j == 0) atomicAdd(n_long, 1u);
            continue;
        }
        if (L == 0) continue;
        const uint64_t o0 = s & ~15ull;
        const uint32_t K = (uint32_t)((e - o0 + 15) >> 4);
        for (uint32_t k = j; k < K; k += G)
            reverse_chunk<COMP, FASTQ>(in, out, qin, qout, s, e, o0 + 16ull * k, s_lut);
    }
}

// Long records: one aligned TILE-byte window of the plane per workgroup iteration.
// Only the record holding the window's first byte and the record holding its last
// byte can be longer than TILE, so two searches find every long record it touches.
template <bool COMP, bool FASTQ>
__global__ __launch_bounds__(SKH_BLOCK) void k_reverse_tiles(
    const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const uint8_t* __restrict__ qin,
    uint8_t* __restrict__ qout, const uint64_t* __restrict__ offs, uint64_t nrec, uint64_t total,
    Table256 lut, const uint32_t* __restrict__ n_long)
{
    if (*n_long == 0) return;
    __shared__ uint8_t s_lut[256];
    if (COMP) {
        s_lut[threadIdx.x] = lut.v[threadIdx.x];
        __syncthreads();
    }
    const uint64_t ntiles = (total + SKH_TILE - 1) / SKH_TILE;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t t0 = tile * SKH_TILE;
        const uint64_t t1 = t0 + SKH_TILE;                       // exclusive, may exceed total
        const uint64_t last = (t1 < total ? t1 : total) - 1;     // last real byte of the window
        const uint64_t r_lo = find_record(offs, nrec, t0);
        const uint64_t r_hi = find_record(offs, nrec, last);
#pragma unroll 1
        for (int c = 0; c < 2; c++) {
            if (c == 1 && r_hi == r_lo) break;
            const uint64_t r = c == 0 ? r_lo : r_hi;
            const uint64_t s = offs[r], e = offs[r + 1];
            if (e - s <= SKH_LONG_T) continue;
            const uint64_t a = s & ~15ull, b = (e + 15) & ~15ull;
            const uint64_t c0 = a > t0 ? a : t0, c1 = b < t1 ? b : t1;
            for (uint64_t o = c0 + 16ull * threadIdx.x; o < c1; o += 16ull * SKH_BLOCK)
                reverse_chunk<COMP, FASTQ>(in, out, qin, qout, s, e, o, s_lut);
        }
    }
}

// No reversal: out[i] = lut[in[i]] over the whole plane (record boundaries do not matter).
__global__ __launch_bounds__(SKH_BLOCK) void k_map_plane(const uint8_t* __restrict__ in,
                                                         uint8_t* __restrict__ out, uint64_t total,
                                                         Table256 lut)
{
    __shared__ uint8_t s_lut[256];
    s_lut[threadIdx.x] = lut.v[threadIdx.x];
    __syncthreads();
    const uint64_t nchunk = (total + 15) >> 4;
    const uint64_t stride = (uint64_t)gridDim.x * SKH_BLOCK;
    for (uint64_t c = (uint64_t)blockIdx.x * SKH_BLOCK + threadIdx.x; c < nchunk; c += stride) {
        const uint64_t o = c << 4;
        const uint4 v = map16(ld16(in + o), s_lut);
        store_owned(out, 0, total, o, v);
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
__device__ __forceinline__ void count_chunk(const uint8_t* __restrict__ seq, uint64_t s, uint64_t e,
                                            uint64_t o, const CountLds<NW>& L, uint32_t neutral4,
                                            bool has_neutral, uint32_t* acc8)
{
    uint4 v = ld16(seq + o);
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

template <int NW>
__global__ __launch_bounds__(SKH_BLOCK) void k_count_short(
    const uint8_t* __restrict__ seq, const uint64_t* __restrict__ offs, uint64_t nrec, Table256 cls,
    uint32_t nclass, int logG, uint32_t neutral, uint64_t* __restrict__ counts,
    uint32_t* __restrict__ n_long)
{
    __shared__ CountLds<NW> L;
    count_lds_init<NW>(L, cls);
    const bool has_neutral = neutral < 256u;
    const uint32_t neutral4 = (neutral & 0xffu) * 0x01010101u;
    const uint32_t G = 1u << logG, j = threadIdx.x & (G - 1);
    const uint64_t per_block = SKH_BLOCK >> logG;
    const uint64_t stride = (uint64_t)gridDim.x * per_block;
    // every lane of a group must stay in the loop for the cross-lane sum: the bound
    // is on the group's record, identical for its lanes; whole groups drop out together
    uint64_t rec = (uint64_t)xcd_remap(blockIdx.x, gridDim.x) * per_block + (threadIdx.x >> logG);
    for (; rec < nrec; rec += stride) {
        const uint64_t s = offs[rec], e = offs[rec + 1];
        const uint64_t len = e - s;
        uint32_t acc16[2 * NW];
#pragma unroll
        for (int i = 0; i < 2 * NW; i++) acc16[i] = 0;
        if (len > SKH_LONG_T) {
            if (j == 0) atomicAdd(n_long, 1u);      // counted by k_count_tiles; zeros written here
        } else if (len != 0) {
            uint32_t acc8[NW];
#pragma unroll
            for (int i = 0; i < NW; i++) acc8[i] = 0;
            const uint64_t o0 = s & ~15ull;
            const uint32_t K = (uint32_t)((e - o0 + 15) >> 4);
            uint32_t pending = 0;
            for (uint32_t k = j; k < K; k += G) {
                count_chunk<NW>(seq, s, e, o0 + 16ull * k, L, neutral4, has_neutral, acc8);
                if (++pending == 15) {       // 15 chunks * 16 bytes = 240 < 256 per field
                    widen<NW>(acc8, acc16);
                    pending = 0;
                }
            }
            widen<NW>(acc8, acc16);
            // a lane sees at most LONG_T bytes and a group LONG_T in total: 16 bits hold it
            for (uint32_t m = 1; m < G; m <<= 1) {
#pragma unroll
                for (int i = 0; i < 2 * NW; i++) acc16[i] += __shfl_xor(acc16[i], m);
            }
        }
        for (uint32_t c = j; c < nclass; c += G)
            counts[rec * nclass + c] = field16<NW>(acc16, c);
    }
}

template <int NW>
__global__ __launch_bounds__(SKH_BLOCK) void k_count_tiles(
    const uint8_t* __restrict__ seq, const uint64_t* __restrict__ offs, uint64_t nrec, uint64_t total,
    Table256 cls, uint32_t nclass, uint32_t neutral, uint64_t* __restrict__ counts,
    const uint32_t* __restrict__ n_long)
{
    if (*n_long == 0) return;
    __shared__ CountLds<NW> L;
    __shared__ uint32_t s_part[SKH_BLOCK / 64][4 * NW];
    count_lds_init<NW>(L, cls);
    const bool has_neutral = neutral < 256u;
    const uint32_t neutral4 = (neutral & 0xffu) * 0x01010101u;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t ntiles = (total + SKH_TILE - 1) / SKH_TILE;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t t0 = tile * SKH_TILE;
        const uint64_t t1 = t0 + SKH_TILE;
        const uint64_t last = (t1 < total ? t1 : total) - 1;
        const uint64_t r_lo = find_record(offs, nrec, t0);
        const uint64_t r_hi = find_record(offs, nrec, last);
#pragma unroll 1
        for (int c = 0; c < 2; c++) {
            if (c == 1 && r_hi == r_lo) break;
            const uint64_t r = c == 0 ? r_lo : r_hi;
            const uint64_t s = offs[r], e = offs[r + 1];
            if (e - s <= SKH_LONG_T) continue;                 // block-uniform
            const uint64_t a = s & ~15ull, b = (e + 15) & ~15ull;
            const uint64_t c0 = a > t0 ? a : t0, c1 = b < t1 ? b : t1;
            uint32_t acc8[NW], acc16[2 * NW];
#pragma unroll
            for (int i = 0; i < NW; i++) acc8[i] = 0;
#pragma unroll
            for (int i = 0; i < 2 * NW; i++) acc16[i] = 0;
            // TILE / (16 * BLOCK) = 8 chunks = 128 bytes per lane: 8-bit fields hold it
            for (uint64_t o = c0 + 16ull * threadIdx.x; o < c1; o += 16ull * SKH_BLOCK)
                count_chunk<NW>(seq, s, e, o, L, neutral4, has_neutral, acc8);
            widen<NW>(acc8, acc16);
            for (uint32_t m = 1; m < 64; m <<= 1) {            // <= 64 * 128 per field
#pragma unroll
                for (int i = 0; i < 2 * NW; i++) acc16[i] += __shfl_xor(acc16[i], m);
            }
            __syncthreads();                                   // s_part free again
            if (lane < nclass) s_part[wave][lane] = field16<NW>(acc16, lane);
            __syncthreads();
            if (threadIdx.x < nclass) {
                uint64_t sum = 0;
#pragma unroll
                for (int w = 0; w < SKH_BLOCK / 64; w++) sum += s_part[w][threadIdx.x];
                if (sum) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[r * nclass + threadIdx.x]),
                                   (unsigned long long)sum);
            }
        }
    }
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
// and min/max of the record lengths from the offsets, in one grid.
__global__ __launch_bounds__(SKH_BLOCK) void k_stats(const uint8_t* __restrict__ seq,
                                                     const uint64_t* __restrict__ offs, uint64_t nrec,
                                                     uint64_t total, Table256 is_gap,
                                                     StatsPartial* __restrict__ out)
{
    __shared__ uint8_t s_gap[256];
    __shared__ unsigned long long s_sum[SKH_BLOCK / 64];
    __shared__ long long s_min[SKH_BLOCK / 64], s_max[SKH_BLOCK / 64];
    s_gap[threadIdx.x] = is_gap.v[threadIdx.x] ? 1 : 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * SKH_BLOCK;
    const uint64_t gid = (uint64_t)blockIdx.x * SKH_BLOCK + threadIdx.x;

    unsigned long long gaps = 0;
    const uint64_t nchunk = (total + 15) >> 4;
    for (uint64_t c = gid; c < nchunk; c += stride) {
        const uint64_t o = c << 4;
        const uint4 v = ld16(seq + o);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t n = 0;
        if (o + 16 <= total) {
#pragma unroll
            for (int k = 0; k < 16; k++) n += s_gap[(w[k >> 2] >> (8 * (k & 3))) & 0xffu];
        } else {
            const int hi = (int)(total - o);
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k < hi) n += s_gap[(w[k >> 2] >> (8 * (k & 3))) & 0xffu];
        }
        gaps += n;
    }
    long long lmin = INT64_MAX, lmax = 0;
    for (uint64_t r = gid; r < nrec; r += stride) {
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
        if (gaps) atomicAdd(reinterpret_cast<unsigned long long*>(&out->gap_sum), gaps);
        if (lmin != INT64_MAX) atomicMin(reinterpret_cast<long long*>(&out->len_min), lmin);
        if (lmax != 0) atomicMax(reinterpret_cast<long long*>(&out->len_max), lmax);
    }
}

// ------------------------------------------------------------------ launchers

static inline uint32_t grid_for(uint64_t work_items, uint64_t per_block, uint32_t cap)
{
    uint64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (uint32_t)g;
}

template <bool COMP, bool FASTQ>
static hipError_t launch_reverse_t(const LaunchCfg& cfg, const uint8_t* in, uint8_t* out,
                                   const uint8_t* qin, uint8_t* qout, const uint64_t* offs,
                                   uint64_t nrec, uint64_t total, const Table256& lut,
                                   uint32_t* n_long, hipStream_t st)
{
    const uint32_t g1 = grid_for(nrec, SKH_BLOCK >> cfg.logG, cfg.max_grid);
    hipLaunchKernelGGL((k_reverse_short<COMP, FASTQ>), dim3(g1), dim3(SKH_BLOCK), 0, st, in, out, qin,
                       qout, offs, nrec, lut, cfg.logG, n_long);
    const uint32_t g2 = grid_for((total + SKH_TILE - 1) / SKH_TILE, 1, cfg.max_grid);
    hipLaunchKernelGGL((k_reverse_tiles<COMP, FASTQ>), dim3(g2), dim3(SKH_BLOCK), 0, st, in, out, qin,
                       qout, offs, nrec, total, lut, n_long);
    return hipGetLastError();
}

hipError_t launch_reverse(const LaunchCfg& cfg, const uint8_t* in, uint8_t* out, const uint8_t* qin,
                          uint8_t* qout, const uint64_t* offs, uint64_t nrec, uint64_t total,
                          const Table256* lut, uint32_t* n_long, hipStream_t st)
{
    static const Table256 ident = [] { Table256 t; for (int i = 0; i < 256; i++) t.v[i] = (uint8_t)i; return t; }();
    const bool fastq = qin != nullptr;
    if (lut) {
        return fastq ? launch_reverse_t<true, true>(cfg, in, out, qin, qout, offs, nrec, total, *lut, n_long, st)
                     : launch_reverse_t<true, false>(cfg, in, out, qin, qout, offs, nrec, total, *lut, n_long, st);
    }
    return fastq ? launch_reverse_t<false, true>(cfg, in, out, qin, qout, offs, nrec, total, ident, n_long, st)
                 : launch_reverse_t<false, false>(cfg, in, out, qin, qout, offs, nrec, total, ident, n_long, st);
}

hipError_t launch_map_plane(const LaunchCfg& cfg, const uint8_t* in, uint8_t* out, uint64_t total,
                            const Table256& lut, hipStream_t st)
{
    const uint32_t g = grid_for((total + 15) >> 4, SKH_BLOCK, cfg.max_grid);
    hipLaunchKernelGGL(k_map_plane, dim3(g), dim3(SKH_BLOCK), 0, st, in, out, total, lut);
    return hipGetLastError();
}

hipError_t launch_count(const LaunchCfg& cfg, const uint8_t* seq, const uint64_t* offs, uint64_t nrec,
                        uint64_t total, const Table256& cls, uint32_t nclass, uint32_t neutral,
                        uint64_t* counts, uint32_t* n_long, hipStream_t st)
{
    const uint32_t g1 = grid_for(nrec, SKH_BLOCK >> cfg.logG, cfg.max_grid);
    const uint32_t g2 = grid_for((total + SKH_TILE - 1) / SKH_TILE, 1, cfg.max_grid);
    if (nclass <= 4) {
        hipLaunchKernelGGL((k_count_short<1>), dim3(g1), dim3(SKH_BLOCK), 0, st, seq, offs, nrec, cls,
                           nclass, cfg.logG, neutral, counts, n_long);
        hipLaunchKernelGGL((k_count_tiles<1>), dim3(g2), dim3(SKH_BLOCK), 0, st, seq, offs, nrec, total,
                           cls, nclass, neutral, counts, n_long);
    } else {
        hipLaunchKernelGGL((k_count_short<2>), dim3(g1), dim3(SKH_BLOCK), 0, st, seq, offs, nrec, cls,
                           nclass, cfg.logG, neutral, counts, n_long);
        hipLaunchKernelGGL((k_count_tiles<2>), dim3(g2), dim3(SKH_BLOCK), 0, st, seq, offs, nrec, total,
                           cls, nclass, neutral, counts, n_long);
    }
    return hipGetLastError();
}

hipError_t launch_stats(const LaunchCfg& cfg, const uint8_t* seq, const uint64_t* offs, uint64_t nrec,
                        uint64_t total, const Table256& is_gap, StatsPartial* out, hipStream_t st)
{
    hipLaunchKernelGGL(k_stats_init, dim3(1), dim3(1), 0, st, out, nrec, total);
    const uint64_t items = ((total + 15) >> 4) > nrec ? ((total + 15) >> 4) : nrec;
    const uint32_t g = grid_for(items, SKH_BLOCK, cfg.max_grid);
    hipLaunchKernelGGL(k_stats, dim3(g), dim3(SKH_BLOCK), 0, st, seq, offs, nrec, total, is_gap, out);
    return hipGetLastError();
}

}  // namespace skh
