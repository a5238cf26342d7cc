// gaps.hip -- gap removal on a batch (gfx950): (*Seq).RemoveGaps(letters) of bio/seq, called once
// per record at seqkit/cmd/seq.go:196-198.
//
// Dropping the gap bytes of every record is a stream compaction of the whole plane: bytes keep
// their order, so where a kept byte goes only depends on how many kept bytes lie before it in the
// plane, not on the record it belongs to.  Three streaming passes over 1-KiB tiles (one wavefront,
// 16 bytes per lane): kept bytes per tile, exclusive scan of the tile counts, scatter of the kept
// bytes (and of the quality bytes at the same positions).  The new offsets are the exclusive scan
// of the per-record kept counts, which the counting kernel of the hot path delivers (one class:
// "not a gap").  Algorithmic bytes: the plane read twice and the kept bytes written once
// (3 B/base, 6 B/base with qualities), 16 B per record of offsets and counts.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_util.h"
#include "kernels.h"

namespace skh {

namespace {

constexpr int GAPB = 256;                 // threads per workgroup
constexpr uint32_t GTILE = 1024;          // bytes per wavefront tile
constexpr int GTPW = 4;                   // consecutive tiles a wavefront takes (a workgroup: 16 KiB of plane)

// bit k of the result: byte k of the chunk at plane offset o is kept (inside the plane, not a gap)
__device__ __forceinline__ uint32_t keep_mask(uint4 v, uint64_t o, uint64_t total, const uint8_t* s_gap)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t b = (w[k >> 2] >> (8 * (k & 3))) & 0xffu;
        m |= (s_gap[b] ? 0u : 1u) << k;
    }
    if (o + 16 > total) m &= o < total ? (1u << (uint32_t)(total - o)) - 1u : 0u;
    return m;
}

// kept bytes per tile.  Only the number matters here, so the bytes go through a table of 0 / 1 words and
// are added up (one shift, one LDS read and half an add per byte) instead of being assembled into a mask.
__global__ __launch_bounds__(GAPB) void k_gap_tile_count(const uint8_t* __restrict__ seq, uint64_t total, Table256 is_gap,
                                                         uint64_t* __restrict__ tile_kept)
{
    __shared__ uint32_t s_keep[256];
    s_keep[threadIdx.x] = is_gap.v[threadIdx.x] ? 0u : 1u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t ntile = (total + GTILE - 1) / GTILE;
    const uint64_t tile0 = ((uint64_t)blockIdx.x * (GAPB / 64) + (threadIdx.x >> 6)) * GTPW;     // GTPW tiles per wavefront
    if (tile0 >= ntile) return;
    uint4 v[GTPW];
#pragma unroll
    for (int t = 0; t < GTPW; t++) {                      // all loads before the first lookup
        const uint64_t o = (tile0 + t) * GTILE + 16ull * lane;
        v[t] = o + 16 <= total ? ld16(seq + o) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < GTPW; t++) {
        if (tile0 + t >= ntile) break;                   // wave-uniform
        const uint64_t o = (tile0 + t) * GTILE + 16ull * lane;
        uint32_t c = 0;
        if (o + 16 <= total) {
            const uint32_t w[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
#pragma unroll
            for (int d = 0; d < 4; d++)
                c += s_keep[w[d] & 0xffu] + s_keep[(w[d] >> 8) & 0xffu] + s_keep[(w[d] >> 16) & 0xffu] + s_keep[w[d] >> 24];
        } else if (o < total) {                          // the chunk that straddles the end of the plane
            for (uint64_t p = o; p < total; p++) c += s_keep[seq[p]];
        }
        for (int d = 1; d < 64; d <<= 1) c += __shfl_xor(c, d);
        if (lane == 0) tile_kept[tile0 + t] = c;
    }
}

__device__ __forceinline__ void store_kept(uint8_t* __restrict__ dst, uint4 v, uint32_t m)
{
    if (m == 0xffffu) {
        __builtin_memcpy(dst, &v, 16);
        return;
    }
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (m & (1u << k)) dst[n++] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
}

__global__ __launch_bounds__(GAPB) void k_gap_scatter(const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual,
                                                      uint64_t total, Table256 is_gap,
                                                      const uint64_t* __restrict__ tile_base,
                                                      uint8_t* __restrict__ out_seq, uint8_t* __restrict__ out_qual)
{
    __shared__ uint8_t s_gap[256];
    s_gap[threadIdx.x] = is_gap.v[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t ntile = (total + GTILE - 1) / GTILE;
    const uint64_t tile0 = ((uint64_t)blockIdx.x * (GAPB / 64) + (threadIdx.x >> 6)) * GTPW;
    if (tile0 >= ntile) return;
    uint4 V[GTPW], Q[GTPW];
#pragma unroll
    for (int t = 0; t < GTPW; t++) {                      // all loads before the first lookup
        const uint64_t o = (tile0 + t) * GTILE + 16ull * lane;
        V[t] = Q[t] = make_uint4(0, 0, 0, 0);
        if (o < total) {
            V[t] = ld16(seq + o);
            if (qual) Q[t] = ld16(qual + o);
        }
    }
#pragma unroll
    for (int t = 0; t < GTPW; t++) {
        const uint64_t tile = tile0 + t;
        if (tile >= ntile) break;                        // wave-uniform
        const uint64_t o = tile * GTILE + 16ull * lane;
        const uint32_t m = o < total ? keep_mask(V[t], o, total, s_gap) : 0u;
        const uint32_t c = (uint32_t)__builtin_popcount(m);
        uint32_t inc = c;                               // inclusive prefix over the lanes
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = __shfl_up(inc, d);
            if ((int)lane >= d) inc += u;
        }
        if (m == 0) continue;
        const uint64_t dst = tile_base[tile] + (inc - c);
        store_kept(out_seq + dst, V[t], m);
        if (qual) store_kept(out_qual + dst, Q[t], m);
    }
}

// out_offs[nrec] = total of the kept counts; the result for the caller
__global__ void k_gap_finish(const uint64_t* __restrict__ kept_total, uint64_t nrec, uint64_t* __restrict__ out_offs,
                             uint64_t* __restrict__ result)
{
    out_offs[nrec] = *kept_total;
    *result = *kept_total;
}

static inline uint64_t up256(uint64_t x) { return (x + 255) & ~255ull; }

}  // namespace

uint64_t remove_gaps_workspace_size(uint64_t nrec, uint64_t total)
{
    const uint64_t ntile = (total + GTILE - 1) / GTILE;
    const uint64_t nmax = nrec > ntile ? nrec : ntile;
    return up256(16) + up256(nrec * 8) + 2 * up256(ntile * 8) + up256(scan64_partials(nmax) * 8);
}

hipError_t launch_remove_gaps(const LaunchCfg& cfg, const uint8_t* seq, const uint8_t* qual, const uint64_t* offs,
                              uint64_t nrec, uint64_t total, const Table256& is_gap, uint8_t* out_seq, uint8_t* out_qual,
                              uint64_t* out_offs, void* workspace, uint64_t* result, BatchHint* hint, hipStream_t s)
{
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    uint64_t* totals = reinterpret_cast<uint64_t*>(ws); ws += up256(16);       // [0] records, [1] tiles
    uint64_t* counts = reinterpret_cast<uint64_t*>(ws); ws += up256(nrec * 8);
    const uint64_t ntile = (total + GTILE - 1) / GTILE;
    uint64_t* tile_kept = reinterpret_cast<uint64_t*>(ws); ws += up256(ntile * 8);
    uint64_t* tile_base = reinterpret_cast<uint64_t*>(ws); ws += up256(ntile * 8);
    uint64_t* part = reinterpret_cast<uint64_t*>(ws);
    hipError_t e = hipMemsetAsync(totals, 0, 16, s);
    if (e != hipSuccess) return e;
    if (nrec == 0) {
        hipLaunchKernelGGL(k_gap_finish, dim3(1), dim3(1), 0, s, totals, nrec, out_offs, result);
        return hipGetLastError();
    }
    // kept bytes per record = the count of the one class "not a gap"; their scan = the new offsets
    Table256 cls;
    uint32_t neutral = 256;
    for (int b = 0; b < 256; b++) {
        cls.v[b] = is_gap.v[b] ? 0xFF : 0;
        if (is_gap.v[b] && neutral == 256) neutral = (uint32_t)b;
    }
    e = launch_count(cfg, seq, offs, nrec, total, cls, 1, neutral, counts, hint, s);
    if (e != hipSuccess) return e;
    e = launch_scan64(counts, nrec, out_offs, part, totals, s);
    if (e != hipSuccess) return e;
    if (ntile) {
        const uint64_t g = (ntile + (GAPB / 64) * GTPW - 1) / ((GAPB / 64) * GTPW);
        if (g > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL(k_gap_tile_count, dim3((uint32_t)g), dim3(GAPB), 0, s, seq, total, is_gap, tile_kept);
        e = launch_scan64(tile_kept, ntile, tile_base, part, totals + 1, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_gap_scatter, dim3((uint32_t)g), dim3(GAPB), 0, s, seq, qual, total, is_gap, tile_base, out_seq,
                           out_qual);
    }
    hipLaunchKernelGGL(k_gap_finish, dim3(1), dim3(1), 0, s, totals, nrec, out_offs, result);
    return hipGetLastError();
}

}  // namespace skh
