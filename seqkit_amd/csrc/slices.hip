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
//      finds the slices holding its first and last byte with two 64-ary searches of the offsets,
//      each lane then bisects between them.  A chunk inside one slice is one unaligned 16-byte
//      load (byte-reversed and looked up for a minus-strand slice) and one 16-byte store; a
//      chunk that spans slice borders is gathered byte by byte.
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

__global__ void k_slice_init(SubseqResult* res)
{
    res->out_bytes = 0; res->status = SUBSEQ_OK; res->pad = 0; res->bad_slice = 0;
}

// slices == nullptr: slice i is record i with the uniform (start, end, minus).
__global__ __launch_bounds__(SB) void k_slice_resolve(const uint64_t* __restrict__ offs, uint64_t nrec,
                                                      const Slice* __restrict__ slices, uint64_t n, int64_t start,
                                                      int64_t end, uint32_t minus, uint64_t* __restrict__ src,
                                                      uint64_t* __restrict__ len, uint8_t* __restrict__ flags,
                                                      SubseqResult* __restrict__ res)
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
        src[i] = sp; len[i] = sl; flags[i] = m ? 1 : 0;
    }
}

__global__ void k_slice_finish(const uint64_t* __restrict__ total, uint64_t n, uint64_t* __restrict__ out_offs,
                               uint64_t capacity, SubseqResult* __restrict__ res)
{
    out_offs[n] = *total;
    res->out_bytes = *total;
    if (res->status == SUBSEQ_OK && *total > capacity) res->status = SUBSEQ_ECAPACITY;
}

// byte t of the output chunk at o, gathered from whichever slice holds it; r = a slice at or before it
template <bool COMP>
__device__ __forceinline__ uint4 gather_chunk(const uint8_t* __restrict__ in, const uint64_t* __restrict__ src,
                                              const uint64_t* __restrict__ oo, const uint8_t* __restrict__ flags,
                                              uint64_t r, uint64_t o, int nbytes, const uint8_t* s_lut)
{
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int t = 0; t < nbytes; t++) {
        const uint64_t p = o + (uint64_t)t;
        while (oo[r + 1] <= p) r++;                       // p < oo[n]: terminates
        const uint64_t os = oo[r], l = oo[r + 1] - os, k = p - os;
        const bool rev = flags[r] != 0;
        uint32_t b = in[src[r] + (rev ? l - 1 - k : k)];
        if (COMP && rev) b = s_lut[b];
        const uint32_t v = b << (8 * (t & 3)), m = (uint32_t)t >> 2;
        w[0] |= m == 0 ? v : 0u; w[1] |= m == 1 ? v : 0u; w[2] |= m == 2 ? v : 0u; w[3] |= m == 3 ? v : 0u;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <bool COMP>
__global__ __launch_bounds__(SB) void k_slice_copy(const uint8_t* __restrict__ in, const uint64_t* __restrict__ src,
                                                   const uint64_t* __restrict__ oo, const uint8_t* __restrict__ flags,
                                                   uint64_t n, Table256 lut, uint8_t* __restrict__ out,
                                                   uint64_t capacity, const SubseqResult* __restrict__ res)
{
    __shared__ uint8_t s_lut[256];
    if (COMP) s_lut[threadIdx.x] = lut.v[threadIdx.x];
    __syncthreads();
    if (res->status != SUBSEQ_OK) return;
    const uint64_t total = res->out_bytes;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t nwave = (total + 1023) >> 10;          // 1 KiB of output per wavefront and step
    for (uint64_t wv = (uint64_t)blockIdx.x * (SB / 64) + (threadIdx.x >> 6); wv < nwave;
         wv += (uint64_t)gridDim.x * (SB / 64)) {
        const uint64_t w0 = wv << 10;
        const uint64_t wlast = (w0 + 1024 < total ? w0 + 1024 : total) - 1;
        const uint64_t r_lo = find_record(oo, n, w0);
        const uint64_t r_hi = oo[r_lo + 1] > wlast ? r_lo : find_record(oo, n, wlast);
        const uint64_t o = w0 + 16ull * lane;
        if (o >= total) continue;
        uint64_t lo = r_lo, hi = r_hi + 1;                // oo[lo] <= o < oo[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (oo[mid] <= o) lo = mid; else hi = mid;
        }
        const uint64_t os = oo[lo], oe = oo[lo + 1];
        uint4 v;
        if (o + 16 <= oe) {                                // the whole chunk lies in slice lo
            const uint64_t k = o - os, l = oe - os;
            if (flags[lo]) {
                v = reverse16(ldu16(in + src[lo] + (l - 16 - k)));
                if (COMP) v = map16(v, s_lut);
            } else {
                v = ldu16(in + src[lo] + k);
            }
            *reinterpret_cast<uint4*>(out + o) = v;
        } else {
            const int nb = total - o < 16 ? (int)(total - o) : 16;
            v = gather_chunk<COMP>(in, src, oo, flags, lo, o, nb, s_lut);
            store_clipped(out, 0, total, o, v);
        }
    }
}

static inline uint64_t up256(uint64_t x) { return (x + 255) & ~255ull; }

}  // namespace

uint64_t subseq_workspace_size(uint64_t n)
{
    return up256(8) + up256(n * 8) + up256(n * 8) + up256(n) + up256(scan64_partials(n) * 8);
}

hipError_t launch_subseq(const LaunchCfg& cfg, const uint8_t* seq, const uint8_t* qual, const uint64_t* offs,
                         uint64_t nrec, const Slice* slices, uint64_t n, int64_t start, int64_t end, uint32_t minus,
                         const Table256* lut, uint8_t* out_seq, uint8_t* out_qual, uint64_t capacity,
                         uint64_t* out_offs, void* workspace, SubseqResult* res, hipStream_t st)
{
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    uint64_t* total = reinterpret_cast<uint64_t*>(ws); ws += up256(8);
    uint64_t* src = reinterpret_cast<uint64_t*>(ws); ws += up256(n * 8);
    uint64_t* len = reinterpret_cast<uint64_t*>(ws); ws += up256(n * 8);
    uint8_t* flags = ws; ws += up256(n);
    uint64_t* part = reinterpret_cast<uint64_t*>(ws);
    auto capped = [](uint64_t b) { return (uint32_t)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); };
    hipLaunchKernelGGL(k_slice_init, dim3(1), dim3(1), 0, st, res);
    hipLaunchKernelGGL(k_slice_resolve, dim3(capped((n + SB - 1) / SB)), dim3(SB), 0, st, offs, nrec, slices, n, start, end,
                       minus, src, len, flags, res);
    hipError_t e = launch_scan64(len, n, out_offs, part, total, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_slice_finish, dim3(1), dim3(1), 0, st, total, n, out_offs, capacity, res);
    // the output size is only known on the device: the grid is sized for the capacity and strides
    const uint64_t want = (capacity + 4095) / 4096;
    const uint64_t cap = (uint64_t)(cfg.n_cu ? cfg.n_cu : 256) * 8;
    const uint32_t g = (uint32_t)(want < 1 ? 1 : (want > cap ? cap : want));
    const Table256 none = {};
    if (lut) hipLaunchKernelGGL(k_slice_copy<true>, dim3(g), dim3(SB), 0, st, seq, src, out_offs, flags, n, *lut, out_seq, capacity, res);
    else hipLaunchKernelGGL(k_slice_copy<false>, dim3(g), dim3(SB), 0, st, seq, src, out_offs, flags, n, none, out_seq, capacity, res);
    if (qual)
        hipLaunchKernelGGL(k_slice_copy<false>, dim3(g), dim3(SB), 0, st, qual, src, out_offs, flags, n, none, out_qual, capacity, res);
    return hipGetLastError();
}

}  // namespace skh
