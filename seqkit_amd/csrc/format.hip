// format.hip -- batch -> FASTA/FASTQ output text on the GPU (gfx950).
//
// The step AFTER the per-record hot path: what `seqkit seq` prints for every record
// (seqkit/cmd/seq.go:163-276): the head line, the sequence re-wrapped to the line width
// (byteutil.WrapByteSlice: '\n' after every `width` bytes, none after the last chunk, nothing
// for width < 1 or an empty sequence), and for FASTQ the "+" line and the quality string.
// Three streaming passes: output size of every record, exclusive scan of the sizes, and a
// cooperative copy (16 lanes per record, 16 bytes per lane and step) of names, bases and
// qualities to their places in the output text.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace skh {

namespace {

constexpr int FB = 256;
constexpr int FSCAN_ITEMS = 8;
constexpr uint32_t FSCAN_TILE = FB * FSCAN_ITEMS;

__device__ __forceinline__ uint64_t wrapped_len(uint64_t len, uint32_t width)
{
    return (width < 1 || len == 0) ? len : len + (len - 1) / width;
}

// ID = head line up to the first white space; the whole line if it starts with white space
// (the default --id-regexp `^([^\s]+)\s?` does not match then).
__device__ __forceinline__ uint32_t id_length(const uint8_t* __restrict__ name, uint32_t n)
{
    uint32_t i = 0;
    while (i < n && !(name[i] == ' ' || (name[i] >= '\t' && name[i] <= '\r'))) i++;
    return i == 0 ? n : i;
}

__global__ __launch_bounds__(FB) void k_fmt_sizes(const FormatSeqArgs a, uint64_t* __restrict__ size,
                                                  uint32_t* __restrict__ shown_name)
{
    for (uint64_t r = (uint64_t)blockIdx.x * FB + threadIdx.x; r < a.nrec; r += (uint64_t)gridDim.x * FB) {
        const uint64_t len = a.offsets[r + 1] - a.offsets[r];
        const bool keep = !(a.min_len >= 0 && len < (uint64_t)a.min_len) && !(a.max_len >= 0 && len > (uint64_t)a.max_len);
        uint32_t nl = 0;
        uint64_t sz = 0;
        if (keep) {
            if (a.print_name) {
                nl = a.name_len[r];
                if (a.only_id) nl = id_length(a.text + a.name_pos[r], nl);
                sz += (a.print_seq ? 1 : 0) + (uint64_t)nl + 1;
            }
            if (a.print_seq) sz += wrapped_len(len, a.width) + 1;
            if (a.print_qual) sz += (a.plus_line ? 2 : 0) + wrapped_len(len, a.width) + 1;
        }
        size[r] = sz;
        shown_name[r] = nl;
    }
}

// ---- exclusive scan of uint64 (block partials, one-workgroup scan of partials, apply) -------

__global__ __launch_bounds__(FB) void k_scan_partials64(const uint64_t* __restrict__ v, uint64_t n,
                                                        uint64_t* __restrict__ part)
{
    __shared__ uint64_t s_w[FB / 64];
    const uint64_t base = (uint64_t)blockIdx.x * FSCAN_TILE + threadIdx.x * FSCAN_ITEMS;
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < FSCAN_ITEMS; k++) if (base + k < n) s += v[base + k];
    for (int d = 1; d < 64; d <<= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < FB / 64; w++) t += s_w[w];
        part[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void k_scan_top64(uint64_t* __restrict__ part, uint64_t npart, uint64_t* __restrict__ total)
{
    __shared__ uint64_t s_p[1024];
    const uint64_t per = (npart + 1023) / 1024;
    const uint64_t a = threadIdx.x * per, b = a + per < npart ? a + per : npart;
    uint64_t s = 0;
    for (uint64_t i = a; i < b; i++) s += part[i];
    s_p[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 1024; i++) { const uint64_t t = s_p[i]; s_p[i] = run; run += t; }
        *total = run;
    }
    __syncthreads();
    uint64_t run = s_p[threadIdx.x];
    for (uint64_t i = a; i < b; i++) { const uint64_t t = part[i]; part[i] = run; run += t; }
}

__global__ __launch_bounds__(FB) void k_scan_apply64(const uint64_t* __restrict__ v, uint64_t n,
                                                     const uint64_t* __restrict__ part, uint64_t* __restrict__ out)
{
    __shared__ uint64_t s_w[FB / 64];
    const uint64_t base = (uint64_t)blockIdx.x * FSCAN_TILE + threadIdx.x * FSCAN_ITEMS;
    uint64_t x[FSCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < FSCAN_ITEMS; k++) { x[k] = base + k < n ? v[base + k] : 0; s += x[k]; }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc = s;
    for (int d = 1; d < 64; d <<= 1) { const uint64_t t = __shfl_up(inc, d); if ((int)lane >= d) inc += t; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint64_t run = part[blockIdx.x] + inc - s;
    for (uint32_t w = 0; w < wave; w++) run += s_w[w];
#pragma unroll
    for (int k = 0; k < FSCAN_ITEMS; k++) { if (base + k < n) out[base + k] = run; run += x[k]; }
}

// ---- cooperative writers (G lanes of one record) --------------------------------------------

template <uint32_t G>
__device__ __forceinline__ void copy_plain(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint64_t n, uint32_t j)
{
    for (uint64_t p = (uint64_t)j * 16; p < n; p += G * 16) {
        if (p + 16 <= n) {
            uint4 v;
            __builtin_memcpy(&v, src + p, 16);
            __builtin_memcpy(dst + p, &v, 16);
        } else {
            for (uint64_t k = p; k < n; k++) dst[k] = src[k];
        }
    }
}

// src[0..n) with a '\n' after every `width` bytes except after the last chunk
template <uint32_t G>
__device__ __forceinline__ void copy_wrapped(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint64_t n,
                                             uint32_t width, uint32_t j)
{
    if (width < 1 || n == 0) { copy_plain<G>(dst, src, n, j); return; }
    for (uint64_t p = (uint64_t)j * 16; p < n; p += G * 16) {
        const uint64_t last = p + 16 <= n ? p + 15 : n - 1;
        const uint64_t l0 = p / width, l1 = last / width;
        if (l0 == l1 && p + 16 <= n) {
            uint4 v;
            __builtin_memcpy(&v, src + p, 16);
            __builtin_memcpy(dst + p + l0, &v, 16);
        } else {
            for (uint64_t k = p; k <= last; k++) dst[k + k / width] = src[k];
        }
    }
    const uint64_t breaks = (n - 1) / width;
    for (uint64_t b = 1 + j; b <= breaks; b += G) dst[b * width + b - 1] = '\n';
}

__global__ __launch_bounds__(FB) void k_fmt_write(const FormatSeqArgs a, const uint64_t* __restrict__ size,
                                                  const uint64_t* __restrict__ pos, const uint32_t* __restrict__ shown_name,
                                                  const uint64_t* __restrict__ total, uint8_t* __restrict__ out,
                                                  uint64_t out_capacity)
{
    constexpr uint32_t G = 16;
    if (*total > out_capacity) return;
    const uint32_t j = threadIdx.x & (G - 1);
    for (uint64_t r = ((uint64_t)blockIdx.x * FB + threadIdx.x) / G; r < a.nrec; r += (uint64_t)gridDim.x * (FB / G)) {
        if (size[r] == 0) continue;
        uint8_t* o = out + pos[r];
        const uint64_t s = a.offsets[r], len = a.offsets[r + 1] - s;
        if (a.print_name) {
            const uint32_t nl = shown_name[r];
            if (a.print_seq) { if (j == 0) *o = a.mark; o++; }
            copy_plain<G>(o, a.text + a.name_pos[r], nl, j);
            if (j == 0) o[nl] = '\n';
            o += nl + 1;
        }
        if (a.print_seq) {
            copy_wrapped<G>(o, a.seq + s, len, a.width, j);
            const uint64_t w = wrapped_len(len, a.width);
            if (j == 0) o[w] = '\n';
            o += w + 1;
        }
        if (a.print_qual) {
            if (a.plus_line) { if (j == 0) { o[0] = '+'; o[1] = '\n'; } o += 2; }
            copy_wrapped<G>(o, a.qual + s, len, a.width, j);
            const uint64_t w = wrapped_len(len, a.width);
            if (j == 0) o[w] = '\n';
        }
    }
}

__global__ void k_fmt_finish(const uint64_t* __restrict__ total, uint64_t out_capacity, FormatResult* __restrict__ res)
{
    res->out_bytes = *total;
    res->status = *total > out_capacity ? FORMAT_ECAPACITY : FORMAT_OK;
    res->pad = 0;
}

static inline uint64_t up256(uint64_t x) { return (x + 255) & ~255ull; }

}  // namespace

uint64_t format_workspace_size(uint64_t nrec)
{
    const uint64_t npart = (nrec + FSCAN_TILE - 1) / FSCAN_TILE + 1;
    return up256(8) + up256(nrec * 8) + up256(nrec * 8) + up256(nrec * 4) + up256(npart * 8);
}

hipError_t launch_format_seq(const FormatSeqArgs& a, uint8_t* out, uint64_t out_capacity, void* workspace,
                             FormatResult* res, hipStream_t s)
{
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    uint64_t* total = reinterpret_cast<uint64_t*>(ws); ws += up256(8);
    uint64_t* size = reinterpret_cast<uint64_t*>(ws); ws += up256(a.nrec * 8);
    uint64_t* pos = reinterpret_cast<uint64_t*>(ws); ws += up256(a.nrec * 8);
    uint32_t* shown = reinterpret_cast<uint32_t*>(ws); ws += up256(a.nrec * 4);
    uint64_t* part = reinterpret_cast<uint64_t*>(ws);
    const uint64_t npart = (a.nrec + FSCAN_TILE - 1) / FSCAN_TILE ? (a.nrec + FSCAN_TILE - 1) / FSCAN_TILE : 1;
    auto capped = [](uint64_t b) { return (uint32_t)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); };
    hipLaunchKernelGGL(k_fmt_sizes, dim3(capped((a.nrec + FB - 1) / FB)), dim3(FB), 0, s, a, size, shown);
    hipLaunchKernelGGL(k_scan_partials64, dim3((uint32_t)npart), dim3(FB), 0, s, size, a.nrec, part);
    hipLaunchKernelGGL(k_scan_top64, dim3(1), dim3(1024), 0, s, part, npart, total);
    hipLaunchKernelGGL(k_scan_apply64, dim3((uint32_t)npart), dim3(FB), 0, s, size, a.nrec, part, pos);
    hipLaunchKernelGGL(k_fmt_write, dim3(capped((a.nrec * 16 + FB - 1) / FB)), dim3(FB), 0, s, a, size, pos, shown, total,
                       out, out_capacity);
    hipLaunchKernelGGL(k_fmt_finish, dim3(1), dim3(1), 0, s, total, out_capacity, res);
    return hipGetLastError();
}

}  // namespace skh
