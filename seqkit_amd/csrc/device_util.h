// device_util.h -- small device helpers shared by the kernel sources (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace skh {

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

// Hand-over of LDS contents between the lanes of ONE wavefront: what any lane wrote before is
// visible to every lane after.  The hardware needs nothing for this (a wavefront's LDS operations
// complete in program order), so no instruction is emitted; the fences at wavefront scope tell the
// compiler not to move LDS accesses across this point, the barrier keeps the lanes' code together.
__device__ __forceinline__ void wave_lds_handover()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint4 ldu16(const uint8_t* p)     // any byte address
{
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

__device__ __forceinline__ uint4 reverse16(uint4 w)
{
    return make_uint4(__builtin_bswap32(w.w), __builtin_bswap32(w.z), __builtin_bswap32(w.y),
                      __builtin_bswap32(w.x));
}

// (a << sh) | b in one instruction
template <int SH>
__device__ __forceinline__ uint32_t lshl_or(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(SH), "v"(b));
    return r;
}

__device__ __forceinline__ uint32_t map4(uint32_t w, const uint8_t* s_lut)
{
    // four byte loads packed with three shift-or operations (spelled out: the compiler
    // re-associates the plain expression into four)
    const uint32_t b0 = s_lut[w & 0xffu], b1 = s_lut[(w >> 8) & 0xffu], b2 = s_lut[(w >> 16) & 0xffu],
                   b3 = s_lut[w >> 24];
    return lshl_or<16>(lshl_or<8>(b3, b2), lshl_or<8>(b1, b0));
}

__device__ __forceinline__ uint4 map16(uint4 v, const uint8_t* s_lut)
{
    return make_uint4(map4(v.x, s_lut), map4(v.y, s_lut), map4(v.z, s_lut), map4(v.w, s_lut));
}

// Stores the bytes [lo_limit, hi_limit) ∩ chunk of `v` at plane offset o.  Whole
// chunks are one 16-byte store; the partial form is only reached at the end of a plane.
__device__ __forceinline__ void store_clipped(uint8_t* __restrict__ out, uint64_t s, uint64_t e,
                                              uint64_t o, uint4 v)
{
    if (o >= s && o + 16 <= e) {
        *reinterpret_cast<uint4*>(out + o) = v;
        return;
    }
    const int lo = s > o ? (int)(s - o) : 0;
    const int hi = e < o + 16 ? (int)(e - o) : 16;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll 1
    for (int k = lo; k < hi; k++) {
        const uint32_t d = k < 8 ? (k < 4 ? w[0] : w[1]) : (k < 12 ? w[2] : w[3]);
        out[o + k] = (uint8_t)(d >> (8 * (k & 3)));
    }
}

}  // namespace skh
