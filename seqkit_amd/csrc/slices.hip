// slices.hip -- subsequences of the records of a batch on the GPU (gfx950).
//
// What `seqkit subseq` does to every record: record.Seq.SubSeq(start, end), and for a feature on
// the minus strand .RevComInplace() on the result (seqkit/cmd/subseq.go:352-355, 393-399, 485-491).
// A slice names a source record and the 1-based, inclusive, possibly negative (start, end) of
// seq.SubLocation; the output is a new packed batch, one record per slice.
//
// Three streaming passes:
//   1. resolve: one thread per slice turns (record, start, end) into (source position, length,
//      flags) with SubLocation's rules;
//   2. exclusive scan of the lengths = offsets of the output batch;
//   3. copy: every lane owns one aligned 16-byte chunk of the OUTPUT plane (so slices of any
//      length, from one byte to a whole chromosome, spread evenly over the grid).  A wavefront
//      serves runs of consecutive 1-KiB spans: one 64-ary search of the offsets finds the slice
//      under the run's first byte, after that it walks forward, copying the offsets and sources
//      of the slices under each span into LDS with whole-wavefront loads; each lane bisects
//      that table.  A chunk inside one slice is one unaligned 16-byte load (byte-reversed and
//      looked up for a minus-strand slice) and one 16-byte store; a chunk that spans slice
//      borders is gathered byte by byte from the table.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "device_util.h"

namespace skh {

namespace {

constexpr int SB = 256;

// seq.SubLocation: false when the region selects nothing; s, e 1-based inclusive (e may be
// s - 1: an empty but valid range).
__device__ __forceinline__ bool sub_location(int64_t length, int64_t start, int64_t end, int64_t& s, int64_t& e)
{
    if (length == 0) return false;
    if (start < 1) {
        if (start == 0) start = 1;
        else {
            if (end < 0 && start > end) return false;
            start = -start > length ? 1 : length + start + 1;
        }
    }
    if (start > length) return false;
    if (end > length) end = length;
    if (end < 1) {
        if (end == 0) end = -1;
        end = length + end + 1;
    }
    if (start - 1 > end) return false;
    s = start; e = end;
    return true;
}

__global__ void k_slice_init(SubseqResult* res, uint32_t* fallback)
{
    res->out_bytes = 0; res->status = SUBSEQ_OK; res->pad = 0; res->bad_slice = 0;
    fallback[0] = fallback[1] = 0;
}

// slices == nullptr: slice i is record i with the uniform (start, end, minus).
__global__ __launch_bounds__(SB) void k_slice_resolve(const uint64_t* __restrict__ offs, uint64_t nrec,
                                                      const Slice* __restrict__ slices, uint64_t n, int64_t start,
                                                      int64_t end, uint32_t minus, uint64_t* __restrict__ src,
                                                      uint64_t* __restrict__ len, SubseqResult* __restrict__ res)
{
    for (uint64_t i = (uint64_t)blockIdx.x * SB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * SB) {
        uint64_t rec = i;
        int64_t a = start, b = end;
        uint32_t m = minus;
        if (slices) { rec = slices[i].record; a = slices[i].start; b = slices[i].end; m = slices[i].minus; }
        uint64_t sp = 0, sl = 0;
        if (rec >= nrec) {
            if (atomicCAS(&res->status, SUBSEQ_OK, SUBSEQ_ERECORD) == SUBSEQ_OK) res->bad_slice = i;
        } else {
            const uint64_t o = offs[rec];
            int64_t s, e;
            if (sub_location((int64_t)(offs[rec + 1] - o), a, b, s, e) && e >= s) {
                sp = o + (uint64_t)(s - 1);
                sl = (uint64_t)(e - s + 1);
            }
        }
        src[i] = sp | (m ? 1ull << 63 : 0); len[i] = sl;      // bit 63: minus strand
    }
}

__global__ void k_slice_finish(const uint64_t* __restrict__ total, uint64_t n, uint64_t* __restrict__ out_offs,
                               uint64_t capacity, SubseqResult* __restrict__ res)
{
    out_offs[n] = *total;
    res->out_bytes = *total;
    if (res->status == SUBSEQ_OK && *total > capacity) res->status = SUBSEQ_ECAPACITY;
}

constexpr uint64_t MINUS = 1ull << 63;                 // top bit of a resolved source position: minus strand
constexpr int SL_T = 256;                                // slice-table entries a wavefront keeps in LDS
constexpr int SL_RUN = 16;                               // consecutive 1-KiB spans a wavefront serves per search

struct SliceLds {
    uint8_t lut[256];
    uint64_t oo[SB / 64][SL_T];                          // output offsets of a stretch of slices
    uint64_t src[SB / 64][SL_T];
};

// Bytes [0, nbytes) of the output chunk at o, gathered from whichever slices hold them.  `oo` and
// `src` are indexed from slice r (global arrays from 0, or a wavefront's LDS table from its first
// slice); r = a slice at or before the one holding byte o.
template <bool COMP, typename TO, typename TS>
__device__ __forceinline__ uint4 gather_chunk(const uint8_t* __restrict__ in, TS src, TO oo, uint64_t r, uint64_t o,
                                              int nbytes, const uint8_t* s_lut)
{
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int t = 0; t < nbytes; t++) {
        const uint64_t p = o + (uint64_t)t;
        while (oo[r + 1] <= p) r++;                       // p < the last offset: terminates
        const uint64_t os = oo[r], l = oo[r + 1] - os, k = p - os, sv = src[r];
        const bool rev = (sv & MINUS) != 0;
        uint32_t b = in[(sv & ~MINUS) + (rev ? l - 1 - k : k)];
        if (COMP && rev) b = s_lut[b];
        const uint32_t v = b << (8 * (t & 3)), m = (uint32_t)t >> 2;
        w[0] |= m == 0 ? v : 0u; w[1] |= m == 1 ? v : 0u; w[2] |= m == 2 ? v : 0u; w[3] |= m == 3 ? v : 0u;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// the output chunk at o, which starts in slice [os, oe) with resolved source sv
template <bool COMP>
__device__ __forceinline__ uint4 whole_chunk(const uint8_t* __restrict__ in, uint64_t sv, uint64_t os, uint64_t oe,
                                             uint64_t o, const uint8_t* s_lut)
{
    const uint64_t k = o - os, l = oe - os, base = sv & ~MINUS;
    if (sv & MINUS) {
        uint4 v = reverse16(ldu16(in + base + (l - 16 - k)));
        if (COMP) v = map16(v, s_lut);
        return v;
    }
    return ldu16(in + base + k);
}

// 16-byte mask with bytes [a, b) set (0 <= a <= b <= 16)
__device__ __forceinline__ uint4 byte_range_mask(int a, int b)
{
    auto upto = [](int k, int half) -> uint64_t {          // bytes below k of the 8-byte half
        const int n = k - 8 * half;
        return n <= 0 ? 0ull : (n >= 8 ? ~0ull : (1ull << (8 * n)) - 1ull);
    };
    const uint64_t lo = upto(b, 0) & ~upto(a, 0), hi = upto(b, 1) & ~upto(a, 1);
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}

// byte t of the result is byte t - n of v (0 <= n < 16); the n lowest bytes are zero
__device__ __forceinline__ uint4 shift_up_bytes(uint4 v, int n)
{
    uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    if (n >= 8) { hi = lo; lo = 0; n -= 8; }
    if (n) { hi = (hi << (8 * n)) | (lo >> (64 - 8 * n)); lo <<= 8 * n; }
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}

constexpr int SL_SEG = 4;                                // segments of a chunk whose windows are loaded together

template <bool COMP>
__global__ __launch_bounds__(SB) void k_slice_copy(const uint8_t* __restrict__ in, const uint64_t* __restrict__ src,
                                                   const uint64_t* __restrict__ oo, uint64_t n, uint64_t in_padded,
                                                   Table256 lut, uint8_t* __restrict__ out,
                                                   const SubseqResult* __restrict__ res, const uint32_t* __restrict__ fallback)
{
    __shared__ SliceLds L;
    if (COMP) L.lut[threadIdx.x] = lut.v[threadIdx.x];
    __syncthreads();
    if (res->status != SUBSEQ_OK || *fallback == 0) return;      // k_slice_copy_flat has written everything
    const uint64_t total = res->out_bytes;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t* t_oo = L.oo[wave];
    uint64_t* t_src = L.src[wave];
    const uint64_t nspan = (total + 1023) >> 10;           // 1 KiB of output per wavefront and step
    const uint64_t nrun = (nspan + SL_RUN - 1) / SL_RUN;
    for (uint64_t run = (uint64_t)blockIdx.x * (SB / 64) + wave; run < nrun; run += (uint64_t)gridDim.x * (SB / 64)) {
        const uint64_t sp_end = (run + 1) * SL_RUN < nspan ? (run + 1) * SL_RUN : nspan;
        uint64_t r_lo = find_record(oo, n, (run * SL_RUN) << 10);      // slice holding the run's first byte
        // The table: offsets and sources of slices tbase .. tbase + tcount - 1, fetched 64 at a time with
        // whole-wavefront loads and kept while the spans of the run walk through it (a table of
        // 150-byte slices lasts for tens of spans, one of chromosome-long slices is a single entry).
        uint64_t tbase = 0;
        uint32_t tcount = 0;
        bool have = false;
        for (uint64_t sp = run * SL_RUN; sp < sp_end; sp++) {
            const uint64_t w0 = sp << 10;
            const uint64_t wlast = (w0 + 1024 < total ? w0 + 1024 : total) - 1;
            const uint64_t o = w0 + 16ull * lane;
            // first table entry after the span's first slice whose offset lies past the span
            uint32_t rs = (uint32_t)(r_lo - tbase);
            auto scan = [&]() -> int {
                for (uint32_t k = rs >> 6; k < (tcount >> 6); k++) {
                    const uint32_t idx = 64u * k + lane;
                    const unsigned long long m = __ballot(idx > rs && t_oo[idx] > wlast);
                    if (m) return (int)(64u * k) + (int)__builtin_ctzll(m);
                }
                return -1;
            };
            int j = have && r_lo - tbase < (uint64_t)tcount ? scan() : -1;
            while (j < 0) {
                if (!have || tcount == SL_T || r_lo - tbase >= (uint64_t)tcount) {
                    if (have && r_lo == tbase && tcount == SL_T) break;       // SL_T slices do not reach the span's end
                    tbase = r_lo; tcount = 0; rs = 0; have = true;
                }
                wave_lds_handover();
                const uint64_t idx = tbase + tcount + lane;
                t_oo[tcount + lane] = idx <= n ? oo[idx] : ~0ull;
                t_src[tcount + lane] = idx < n ? src[idx] : 0;
                tcount += 64;
                wave_lds_handover();
                j = scan();
            }
            if (j < 0) {
                // more than SL_T slices under one span (slices of a few bytes): global tables, bisected per lane
                const uint64_t r_hi = find_record(oo, n, wlast);
                if (o < total) {
                    uint64_t lo = r_lo, hi = r_hi + 1;     // oo[lo] <= o < oo[hi]
                    while (hi - lo > 1) {
                        const uint64_t mid = (lo + hi) >> 1;
                        if (oo[mid] <= o) lo = mid; else hi = mid;
                    }
                    const uint64_t os = oo[lo], oe = oo[lo + 1];
                    if (o + 16 <= oe) {
                        *reinterpret_cast<uint4*>(out + o) = whole_chunk<COMP>(in, src[lo], os, oe, o, L.lut);
                    } else {
                        const int nb = total - o < 16 ? (int)(total - o) : 16;
                        store_clipped(out, 0, total, o, gather_chunk<COMP>(in, src, oo, lo, o, nb, L.lut));
                    }
                }
                r_lo = r_hi;
                continue;
            }
            // A chunk is cut into segments, one per slice under it (one segment = the whole chunk for
            // most chunks of long slices).  A segment's source bytes are the 16-byte window that
            // puts them at their place in the chunk; the windows of up to SL_SEG segments are all
            // requested before the first is used, then masked and merged.  What is left (a fifth
            // slice under one chunk, a window that would reach past the ends of the plane) is gathered
            // byte by byte.
            if (o < total) {
                const int nb = total - o < 16 ? (int)(total - o) : 16;
                uint32_t ri = rs, hi = (uint32_t)j;        // t_oo[ri] <= o < t_oo[hi]
                while (hi - ri > 1) {
                    const uint32_t mid = (ri + hi) >> 1;
                    if (t_oo[mid] <= o) ri = mid; else hi = mid;
                }
                uint4 W[SL_SEG];
                int tb[SL_SEG];
                bool rev[SL_SEG];
                int ta = 0;                                 // bytes [0, ta) of the chunk are accounted for
                bool bytewise = false;
#pragma unroll
                for (int g = 0; g < SL_SEG; g++) {
                    tb[g] = ta;
                    rev[g] = false;
                    W[g] = make_uint4(0, 0, 0, 0);
                    if (ta < nb && !bytewise) {
                        while (t_oo[ri + 1] <= o + (uint64_t)ta) ri++;          // o + ta < total: stops inside the table
                        const uint64_t os = t_oo[ri], oe = t_oo[ri + 1], sv = t_src[ri], base = sv & ~MINUS;
                        rev[g] = (sv & MINUS) != 0;
                        // byte t of the chunk is in[A + t] (plus strand) or in[A + 15 - t] (minus strand)
                        const int64_t A = rev[g] ? (int64_t)(base + (oe - os) - 16) - (int64_t)(o - os)
                                                 : (int64_t)base + (int64_t)(o - os);
                        if (A < 0 || (uint64_t)A + 16 > in_padded) {
                            bytewise = true;
                        } else {
                            W[g] = ldu16(in + A);
                            tb[g] = oe - o < (uint64_t)nb ? (int)(oe - o) : nb;
                            ta = tb[g];
                        }
                    }
                }
                uint4 v = make_uint4(0, 0, 0, 0);
                int t0 = 0;
#pragma unroll
                for (int g = 0; g < SL_SEG; g++) {
                    uint4 x = W[g];
                    if (rev[g]) {
                        x = reverse16(x);
                        if (COMP) x = map16(x, L.lut);
                    }
                    const uint4 m = byte_range_mask(t0, tb[g]);
                    v.x |= x.x & m.x; v.y |= x.y & m.y; v.z |= x.z & m.z; v.w |= x.w & m.w;
                    t0 = tb[g];
                }
                if (ta < nb) {                              // the rest, one byte at a time
                    const uint4 r = gather_chunk<COMP>(in, t_src, t_oo, (uint64_t)ri, o + (uint64_t)ta, nb - ta, L.lut);
                    const uint4 sh = shift_up_bytes(r, ta);
                    v.x |= sh.x; v.y |= sh.y; v.z |= sh.z; v.w |= sh.w;
                }
                store_clipped(out, 0, total, o, v);
            }
            r_lo = tbase + (uint64_t)(j - 1);              // the slice holding the span's last byte
        }
    }
}

// Short slices: the output of consecutive slices is one contiguous stretch, so a wavefront takes as many
// slices per step as fit an LDS stage (from the mean slice length, at most 64; lane i holds slice i), puts
// them together there and writes the stretch as aligned whole-wavefront rows.  The 16-byte pieces of all
// slices of the step are dealt over the 64 lanes (every slice gets PF piece slots, the most any slice of the
// step needs; slot q = lane + 64 k is piece q mod PF of slice q / PF); a piece of a minus-strand slice is the
// mirrored window of its source, byte-reversed (and complemented).  Every window lies inside its slice's own
// source bytes.  A step that does not fit the stage raises `fallback`, and k_slice_copy then writes the
// whole output once more its own way (same bytes).
constexpr uint32_t SF_STAGE = 7 * 1024;

__device__ __forceinline__ uint64_t sl_shfl64(uint64_t v, uint32_t src_lane)
{
    return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)src_lane) << 32) |
           (uint32_t)__shfl((int)(uint32_t)v, (int)src_lane);
}

template <bool COMP>
__global__ __launch_bounds__(SB) void k_slice_copy_flat(const uint8_t* __restrict__ in, const uint64_t* __restrict__ src,
                                                        const uint64_t* __restrict__ oo, uint64_t n, Table256 lut,
                                                        uint8_t* __restrict__ out, const SubseqResult* __restrict__ res,
                                                        uint32_t* __restrict__ fallback)
{
    __shared__ uint8_t s_lut[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[SB / 64][SF_STAGE + 32];
    if (COMP) s_lut[threadIdx.x] = lut.v[threadIdx.x];
    __syncthreads();
    if (res->status != SUBSEQ_OK) return;
    const uint64_t total = res->out_bytes;
    if (n == 0 || total == 0) return;
    if (total / n > 4 * SF_STAGE || total / n < 48) {
        // slices far beyond the stage, or so short that 64 of them are a few hundred bytes per step (12-byte
        // slices: 0.72 ms here against 0.54 ms for the chunk-owner kernel): left to k_slice_copy
        if (blockIdx.x == 0 && threadIdx.x == 0) *fallback = 1u;
        return;
    }
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t R;
    {
        const uint64_t r = (SF_STAGE - 64) * 7 / 8 / (total / n + 1);
        R = r < 1 ? 1u : (r > 64 ? 64u : (uint32_t)r);
    }
    const uint64_t nsteps = (n + R - 1) / R;
    uint8_t* stage = s_stage[wave];
    for (uint64_t step = (uint64_t)blockIdx.x * (SB / 64) + wave; step < nsteps; step += (uint64_t)gridDim.x * (SB / 64)) {
        const uint64_t i0 = step * R;
        const uint32_t cnt = n - i0 < R ? (uint32_t)(n - i0) : R;
        uint64_t t_o = 0, t_e = 0, t_src = 0;
        if (lane < cnt) { t_o = oo[i0 + lane]; t_e = oo[i0 + lane + 1]; t_src = src[i0 + lane]; }
        const uint64_t p0 = sl_shfl64(t_o, 0), p1 = sl_shfl64(t_e, cnt - 1);
        const uint64_t a0 = p0 & ~15ull;
        if (p1 - a0 > SF_STAGE) {                        // wave-uniform
            if (lane == 0) *fallback = 1u;               // (every writer stores the same value)
            continue;
        }
        if (p1 == p0) continue;
        const uint32_t t_len = (uint32_t)(t_e - t_o), d_rel = (uint32_t)(t_o - a0);
        uint32_t PF = (t_len + 15) >> 4;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)PF, d); PF = t > PF ? t : PF; }
        const uint32_t nslots = cnt * PF;                // PF >= 1: the step has bytes
        const uint32_t magic = PF > 1 ? (uint32_t)((0x100000000ull + PF - 1) / PF) : 0u;   // q / PF for q < 2^16
        // PU rounds of slots at a time, the loads of all of them ahead of the first write to the stage
        constexpr int PU = 4;                                 // (2 and 6 measure the same)
        for (uint32_t q0 = 0; q0 < nslots; q0 += 64 * PU) {   // every lane takes part in the cross-lane reads
            uint4 v[PU];
            uint32_t at[PU];
            bool act[PU], minus[PU];
#pragma unroll
            for (int u = 0; u < PU; u++) {
                const uint32_t qq = q0 + 64 * u + lane;
                const uint32_t q = qq < nslots ? qq : nslots - 1;
                const uint32_t r = PF > 1 ? __umulhi(q, magic) : q, k = q - r * PF;
                const uint32_t len = (uint32_t)__shfl((int)t_len, (int)r), dr = (uint32_t)__shfl((int)d_rel, (int)r);
                const uint64_t sv = sl_shfl64(t_src, r);
                const uint32_t p = 16 * k;
                const bool in_slice = qq < nslots && p < len;
                const uint8_t* s = in + (sv & ~MINUS);
                minus[u] = (sv & MINUS) != 0;
                act[u] = in_slice && len >= 16;
                // the piece [p, p + 16), or the slice's last 16 bytes once more; on the minus strand out[j] =
                // lut[s[len - 1 - j]], so the piece at output offset t is the window [len - t - 16, len - t) reversed
                const uint32_t t = p + 16 <= len ? p : len - 16;
                at[u] = dr + t;
                v[u] = act[u] ? ldu16(s + (minus[u] ? len - t - 16 : t)) : make_uint4(0, 0, 0, 0);
                if (in_slice && len < 16) {
                    uint8_t* o = stage + dr;
                    if (!minus[u]) for (uint32_t b = 0; b < len; b++) o[b] = s[b];
                    else for (uint32_t b = 0; b < len; b++) { const uint8_t c = s[len - 1 - b]; o[b] = COMP ? s_lut[c] : c; }
                }
            }
#pragma unroll
            for (int u = 0; u < PU; u++) {
                if (!act[u]) continue;
                uint4 w = v[u];
                if (minus[u]) {
                    w = reverse16(w);
                    if (COMP) w = map16(w, s_lut);
                }
                __builtin_memcpy(stage + at[u], &w, 16);
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

static inline uint64_t up256(uint64_t x) { return (x + 255) & ~255ull; }

}  // namespace

uint64_t subseq_workspace_size(uint64_t n)
{
    return up256(8) + up256(n * 8) + up256(n * 8) + up256(scan64_partials(n) * 8);
}

hipError_t launch_subseq(const LaunchCfg& cfg, const uint8_t* seq, const uint8_t* qual, const uint64_t* offs,
                         uint64_t nrec, uint64_t total_in, const Slice* slices, uint64_t n, int64_t start, int64_t end, uint32_t minus,
                         const Table256* lut, uint8_t* out_seq, uint8_t* out_qual, uint64_t capacity,
                         uint64_t* out_offs, void* workspace, SubseqResult* res, hipStream_t st)
{
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    uint64_t* total = reinterpret_cast<uint64_t*>(ws); ws += up256(8);
    uint64_t* src = reinterpret_cast<uint64_t*>(ws); ws += up256(n * 8);
    uint64_t* len = reinterpret_cast<uint64_t*>(ws); ws += up256(n * 8);
    uint64_t* part = reinterpret_cast<uint64_t*>(ws);
    auto capped = [](uint64_t b) { return (uint32_t)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); };
    uint32_t* fallback = reinterpret_cast<uint32_t*>(total + 1);        // two flags behind the total (per plane)
    hipLaunchKernelGGL(k_slice_init, dim3(1), dim3(1), 0, st, res, fallback);
    hipLaunchKernelGGL(k_slice_resolve, dim3(capped((n + SB - 1) / SB)), dim3(SB), 0, st, offs, nrec, slices, n, start, end,
                       minus, src, len, res);
    hipError_t e = launch_scan64(len, n, out_offs, part, total, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_slice_finish, dim3(1), dim3(1), 0, st, total, n, out_offs, capacity, res);
    // the output size is only known on the device: the grid is sized for the capacity and strides
    const uint64_t want = (capacity + 1024ull * SL_RUN * (SB / 64) - 1) / (1024ull * SL_RUN * (SB / 64));
    static int occ = 0;                                  // resident workgroups per CU (the same for both instantiations)
    if (occ == 0) {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_slice_copy<true>, SB, 0) != hipSuccess || v < 1) v = 4;
        occ = v > 8 ? 8 : v;
    }
    const uint64_t cap = (uint64_t)(cfg.n_cu ? cfg.n_cu : 256) * (uint64_t)occ;
    const uint32_t g = (uint32_t)(want < 1 ? 1 : (want > cap ? cap : want));
    const Table256 none = {};
    const uint64_t in_padded = (total_in + 15) & ~15ull;
    const uint32_t gf = capped((n + 7) / 8);             // about two steps of 16 slices per wavefront; grid-stride
    if (lut) {
        hipLaunchKernelGGL(k_slice_copy_flat<true>, dim3(gf), dim3(SB), 0, st, seq, src, out_offs, n, *lut, out_seq, res, fallback);
        hipLaunchKernelGGL(k_slice_copy<true>, dim3(g), dim3(SB), 0, st, seq, src, out_offs, n, in_padded, *lut, out_seq, res, fallback);
    } else {
        hipLaunchKernelGGL(k_slice_copy_flat<false>, dim3(gf), dim3(SB), 0, st, seq, src, out_offs, n, none, out_seq, res, fallback);
        hipLaunchKernelGGL(k_slice_copy<false>, dim3(g), dim3(SB), 0, st, seq, src, out_offs, n, in_padded, none, out_seq, res, fallback);
    }
    if (qual) {
        hipLaunchKernelGGL(k_slice_copy_flat<false>, dim3(gf), dim3(SB), 0, st, qual, src, out_offs, n, none, out_qual, res, fallback + 1);
        hipLaunchKernelGGL(k_slice_copy<false>, dim3(g), dim3(SB), 0, st, qual, src, out_offs, n, in_padded, none, out_qual, res,
                           fallback + 1);
    }
    return hipGetLastError();
}

}  // namespace skh
