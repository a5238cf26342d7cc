// Standalone access-pattern experiments (not part of the library): how fast can a plane be
// streamed in -> out when the unit of work is a wavefront-sized stretch of a few KiB, as in
// the short-record kernels?  Build: hipcc --offload-arch=gfx950 -O3 -o stream_lab stream_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// A: plain grid-stride copy, U chunks per thread per iteration
template <int U>
__global__ __launch_bounds__(256) void k_plain(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256 * U;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * 256 < n) v[u] = in[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * 256 < n) out[i + u * 256] = v[u];
    }
}

// B: each wavefront copies one stretch of S bytes (S multiple of 16) per step through registers;
// steps are strided by the number of wavefronts in the grid.  ROWS = ceil(S / 1024).
template <int ROWS, bool PREFETCH>
__global__ __launch_bounds__(256) void k_wave_reg(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                   uint64_t nsteps, uint32_t S)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t nw = (uint64_t)gridDim.x * 4;
    uint64_t step = (uint64_t)blockIdx.x * 4 + wave;
    if (step >= nsteps) return;
    uint4 v[ROWS], w[ROWS];
    auto load = [&](uint4* r, uint64_t st) {
#pragma unroll
        for (int i = 0; i < ROWS; i++) {
            const uint32_t off = i * 1024 + lane * 16;
            r[i] = make_uint4(0, 0, 0, 0);
            if (off < S) r[i] = *reinterpret_cast<const uint4*>(in + st * S + off);
        }
    };
    auto store = [&](const uint4* r, uint64_t st) {
#pragma unroll
        for (int i = 0; i < ROWS; i++) {
            const uint32_t off = i * 1024 + lane * 16;
            if (off < S) *reinterpret_cast<uint4*>(out + st * S + off) = r[i];
        }
    };
    if (!PREFETCH) {
        for (; step < nsteps; step += nw) { load(v, step); store(v, step); }
    } else {
        load(v, step);
        while (true) {
            const uint64_t nx = step + nw;
            if (nx < nsteps) load(w, nx);
            store(v, step);
            if (nx >= nsteps) break;
#pragma unroll
            for (int i = 0; i < ROWS; i++) v[i] = w[i];
            step = nx;
        }
    }
}

// C: the same through LDS with direct global->LDS loads, NB buffers per wavefront held in
// distinct LDS variables so that the compiler can tell the buffers apart when it places waits.
template <int ROWS>
struct Slot { __attribute__((aligned(16))) uint8_t b[4][ROWS * 1024]; };

template <int ROWS>
__device__ __forceinline__ void dma(const uint8_t* __restrict__ in, uint64_t st, uint32_t S, uint8_t* slot, uint32_t lane)
{
#pragma unroll
    for (int i = 0; i < ROWS; i++) {
        // every row is issued by every lane (surplus lanes re-read the last chunk): no branch
        // around the load, so the number of loads in flight is known at compile time
        uint32_t off = i * 1024 + lane * 16;
        off = off < S ? off : S - 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + st * S + off),
                                         (__attribute__((address_space(3))) void*)(slot + i * 1024), 16, 0, 0);
    }
}
template <int ROWS>
__device__ __forceinline__ void drain(uint8_t* __restrict__ out, uint64_t st, uint32_t S, const uint8_t* slot, uint32_t lane)
{
#pragma unroll
    for (int i = 0; i < ROWS; i++) {
        const uint32_t off = i * 1024 + lane * 16;
        if (off < S) *reinterpret_cast<uint4*>(out + st * S + off) = *reinterpret_cast<const uint4*>(slot + off);
    }
}

template <int ROWS, int NB>
__global__ __launch_bounds__(256) void k_wave_lds(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                   uint64_t nsteps, uint32_t S)
{
    __shared__ Slot<ROWS> s0, s1, s2, s3;
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t nw = (uint64_t)gridDim.x * 4;
    uint64_t step = (uint64_t)blockIdx.x * 4 + wave;
#define SLOT(i) ((i) == 0 ? s0.b[wave] : (i) == 1 ? s1.b[wave] : (i) == 2 ? s2.b[wave] : s3.b[wave])
    // prologue: NB-1 steps in flight.  Loads are never skipped: a step past the end re-reads
    // the last step, so the number of loads between a buffer's fill and its use is constant.
    if (step >= nsteps) return;
    const uint64_t last = nsteps - 1;
#pragma unroll
    for (int i = 0; i < NB - 1; i++) {
        const uint64_t t = step + i * nw;
        dma<ROWS>(in, t < nsteps ? t : last, S, SLOT(i), lane);
    }
    while (true) {
        bool done = false;
#pragma unroll
        for (int ph = 0; ph < NB; ph++) {
            if (!done) {
                const uint64_t ahead = step + (NB - 1) * nw;
                if (NB > 1) dma<ROWS>(in, ahead < nsteps ? ahead : last, S, SLOT((ph + NB - 1) % NB), lane);
                if (NB == 1) dma<ROWS>(in, step, S, SLOT(0), lane);
                drain<ROWS>(out, step, S, SLOT(ph), lane);
                step += nw;
                done = step >= nsteps;
            }
        }
        if (done) break;
    }
#undef SLOT
}

// D: the same pipeline with the global->LDS loads issued from inline assembly, so that the
// compiler does not know of them and places no wait before the LDS reads; the wait is written
// by hand as vmcnt(N), N = loads issued after the buffer's own (memory operations of a
// wavefront complete in issue order; stores issued in between only make the wait stricter).
__device__ __forceinline__ void dma16_asm(const uint8_t* g, uint32_t lds_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int ROWS, int NB>
__global__ __launch_bounds__(256) void k_wave_asm(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                   uint64_t nsteps, uint32_t S)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[4][NB][ROWS * 1024];
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t nw = (uint64_t)gridDim.x * 4;
    uint64_t step = (uint64_t)blockIdx.x * 4 + wave;
    if (step >= nsteps) return;
    const uint64_t last = nsteps - 1;
    auto issue = [&](uint64_t t, uint32_t b) {
        t = t < nsteps ? t : last;
        const uint32_t base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint8_t*)&stage[wave][b][0]);
#pragma unroll
        for (int i = 0; i < ROWS; i++) {
            uint32_t off = i * 1024 + lane * 16;
            off = off < S ? off : S - 16;
            dma16_asm(in + t * S + off, base + i * 1024);
        }
    };
#pragma unroll
    for (int i = 0; i < NB - 1; i++) issue(step + i * nw, i);
    uint32_t b = 0;
    while (true) {
        uint32_t bf = b + NB - 1;
        bf = bf >= NB ? bf - NB : bf;
        issue(step + (NB - 1) * nw, NB == 1 ? 0 : bf);
        wait_vmcnt<(NB == 1 ? 0 : (NB - 1) * ROWS)>();
        drain<ROWS>(out, step, S, &stage[wave][b][0], lane);
        step += nw;
        if (step >= nsteps) break;
        b = b + 1 == NB ? 0 : b + 1;
    }
    wait_vmcnt<0>();     // nothing may land in LDS after the wavefront has gone
}

// E: as D, but the plane is cut into segments of 4*K steps; a workgroup sweeps one segment at a
// time (its 4 wavefronts interleaved), segments are taken grid-stride.  With grid = number of
// segments every workgroup does exactly one segment and the hardware hands them out in order.
template <int ROWS, int NB>
__global__ __launch_bounds__(256) void k_wave_seg(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                   uint64_t nsteps, uint32_t S, uint32_t logK)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[4][NB][ROWS * 1024];
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t K = 1u << logK;
    const uint64_t nseg = (nsteps + 4 * K - 1) / (4 * K);
    if (blockIdx.x >= nseg) return;
    const uint64_t nloc = (uint64_t)K * ((nseg - blockIdx.x + gridDim.x - 1) / gridDim.x);   // local steps
    auto step_of = [&](uint64_t n) {
        const uint64_t seg = blockIdx.x + (n >> logK) * gridDim.x;
        return seg * 4 * K + 4 * (n & (K - 1)) + wave;
    };
    auto issue = [&](uint64_t n, uint32_t b) {
        uint64_t t = n < nloc ? step_of(n) : nsteps;
        t = t < nsteps ? t : nsteps - 1;
        const uint32_t base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint8_t*)&stage[wave][b][0]);
#pragma unroll
        for (int i = 0; i < ROWS; i++) {
            uint32_t off = i * 1024 + lane * 16;
            off = off < S ? off : S - 16;
            dma16_asm(in + t * S + off, base + i * 1024);
        }
    };
#pragma unroll
    for (int i = 0; i < NB - 1; i++) issue(i, i);
    uint32_t b = 0;
    for (uint64_t n = 0; n < nloc; n++) {
        uint32_t bf = b + NB - 1;
        bf = bf >= NB ? bf - NB : bf;
        issue(n + NB - 1, NB == 1 ? 0 : bf);
        wait_vmcnt<(NB == 1 ? 0 : (NB - 1) * ROWS)>();
        const uint64_t t = step_of(n);
        if (t < nsteps) drain<ROWS>(out, t, S, &stage[wave][b][0], lane);
        b = b + 1 == NB ? 0 : b + 1;
    }
    wait_vmcnt<0>();
}


// F/G: stretch bounds come from an offsets array (GW records per step), as in the real kernels:
// lanes 0..GW+1 load offs[rec0-1+lane], the first and last value give the stretch.  PERSIST:
// grid-stride loop with the next step's offsets prefetched; otherwise K consecutive steps per wave
// and a grid that covers the plane.
template <int ROWS>
__device__ __forceinline__ void stretch_copy(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                             uint64_t a0, uint32_t bytes, uint32_t lane)
{
    uint4 v[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; i++) {
        const uint32_t off = i * 1024 + lane * 16;
        v[i] = make_uint4(0, 0, 0, 0);
        if (off < bytes) v[i] = *reinterpret_cast<const uint4*>(in + a0 + off);
    }
#pragma unroll
    for (int i = 0; i < ROWS; i++) {
        const uint32_t off = i * 1024 + lane * 16;
        if (off < bytes) *reinterpret_cast<uint4*>(out + a0 + off) = v[i];
    }
}
__device__ __forceinline__ uint64_t offs_at(const uint64_t* __restrict__ offs, uint64_t nrec, uint64_t rec0, uint32_t lane, uint32_t GW)
{
    uint64_t v = 0;
    const uint64_t idx = rec0 - 1 + lane;
    if (lane <= GW + 1 && !(rec0 == 0 && lane == 0)) v = offs[idx > nrec ? nrec : idx];
    return v;
}
__device__ __forceinline__ void bounds(uint64_t v, uint32_t GW, uint64_t padded, uint64_t& a0, uint32_t& bytes)
{
    const uint64_t first = __shfl(v, 0), last = __shfl(v, (int)GW + 1);
    const uint64_t lo = first & ~15ull;
    a0 = lo >= 16 ? lo - 16 : 0;
    uint64_t a1 = ((last + 15) & ~15ull);
    if (a1 > padded) a1 = padded;
    bytes = (uint32_t)(a1 - a0);
}
template <int ROWS, int K, bool PERSIST>
__global__ __launch_bounds__(256) void k_wave_offs(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                    const uint64_t* __restrict__ offs, uint64_t nrec, uint64_t padded, uint32_t GW)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (PERSIST) {
        const uint64_t stride = (uint64_t)gridDim.x * 4 * GW;
        uint64_t rec0 = ((uint64_t)blockIdx.x * 4 + wave) * GW;
        if (rec0 >= nrec) return;
        uint64_t v = offs_at(offs, nrec, rec0, lane, GW);
        while (true) {
            const uint64_t nx = rec0 + stride;
            uint64_t nv = 0;
            if (nx < nrec) nv = offs_at(offs, nrec, nx, lane, GW);
            uint64_t a0; uint32_t bytes;
            bounds(v, GW, padded, a0, bytes);
            stretch_copy<ROWS>(in, out, a0, bytes, lane);
            if (nx >= nrec) break;
            rec0 = nx; v = nv;
        }
    } else {
        const uint64_t base = ((uint64_t)blockIdx.x * 4 + wave) * K * GW;
        uint64_t v[K];
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = base + (uint64_t)k * GW < nrec ? offs_at(offs, nrec, base + (uint64_t)k * GW, lane, GW) : 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (base + (uint64_t)k * GW < nrec) {
                uint64_t a0; uint32_t bytes;
                bounds(v[k], GW, padded, a0, bytes);
                stretch_copy<ROWS>(in, out, a0, bytes, lane);
            }
        }
    }
}

__global__ void k_compare(const uint64_t* a, const uint64_t* b, uint64_t n, unsigned long long* bad)
{
    unsigned long long c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(bad, c);
}

int main(int argc, char** argv)
{
    const uint64_t total = 1500000000ull;
    uint8_t *in, *out;
    CK(hipMalloc(&in, total + 4096));
    CK(hipMalloc(&out, total + 4096));
    {
        std::vector<uint8_t> h(total);
        uint64_t x = 88172645463325252ull;
        for (uint64_t i = 0; i < total; i += 8) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; *(uint64_t*)&h[i] = x; }
        CK(hipMemcpy(in, h.data(), total, hipMemcpyHostToDevice));
    }
    CK(hipMemset(out, 0, total));
    unsigned long long* d_bad;
    CK(hipMalloc(&d_bad, 8));
    uint64_t check_len = 0;     // bytes the current variant is expected to have copied (0 = all)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 2; i++) launch();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; i++) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 10;
        const uint64_t covered = check_len ? check_len : total;
        CK(hipMemset(d_bad, 0, 8));
        hipLaunchKernelGGL(k_compare, dim3(4096), dim3(256), 0, 0, (const uint64_t*)in, (const uint64_t*)out, covered / 8, d_bad);
        unsigned long long bad = 0;
        CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
        CK(hipMemset(out, 0, total));
        printf("%-44s %7.3f ms  %7.1f GB/s  %s\n", name, ms, 2.0 * total / ms / 1e6, bad ? "MISMATCH" : "ok");
        fflush(stdout);
    };
    char name[128];
    const uint64_t n16 = total / 16;
    const bool only_lines = argc > 1;
    for (int bpc : {4, 8, 16}) {
        if (only_lines) break;
        const int grid = 256 * bpc;
        snprintf(name, sizeof name, "plain U=1 grid=%d", grid);
        timeit(name, [&] { hipLaunchKernelGGL(k_plain<1>, dim3(grid), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, n16); });
        snprintf(name, sizeof name, "plain U=4 grid=%d", grid);
        timeit(name, [&] { hipLaunchKernelGGL(k_plain<4>, dim3(grid), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, n16); });
    }
    {
        const int grid = (int)((n16 + 255) / 256);
        timeit("plain U=1 grid=n/256 (one chunk per thread)",
               [&] { hipLaunchKernelGGL(k_plain<1>, dim3(grid), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, n16); });
    }
    for (uint32_t S : {2400u, 3072u, 4800u}) {
        if (only_lines) break;
        const uint64_t nsteps = total / S;
        check_len = nsteps * S;
        for (int bpc : {4, 6, 8}) {
            const int grid = 256 * bpc;
            if (S <= 3072) {
                snprintf(name, sizeof name, "wave/reg S=%u grid=%d", S, grid);
                timeit(name, [&] { hipLaunchKernelGGL((k_wave_reg<3, false>), dim3(grid), dim3(256), 0, 0, in, out, nsteps, S); });
                snprintf(name, sizeof name, "wave/reg prefetch S=%u grid=%d", S, grid);
                timeit(name, [&] { hipLaunchKernelGGL((k_wave_reg<3, true>), dim3(grid), dim3(256), 0, 0, in, out, nsteps, S); });
            } else {
                snprintf(name, sizeof name, "wave/reg S=%u grid=%d", S, grid);
                timeit(name, [&] { hipLaunchKernelGGL((k_wave_reg<5, false>), dim3(grid), dim3(256), 0, 0, in, out, nsteps, S); });
                snprintf(name, sizeof name, "wave/reg prefetch S=%u grid=%d", S, grid);
                timeit(name, [&] { hipLaunchKernelGGL((k_wave_reg<5, true>), dim3(grid), dim3(256), 0, 0, in, out, nsteps, S); });
            }
        }
    }
    for (uint32_t S : {2400u, 3072u}) {
        if (only_lines) break;
        const uint64_t nsteps = total / S;
        check_len = nsteps * S;
        for (int bpc : {3, 4, 5}) {
            const int grid = 256 * bpc;
            snprintf(name, sizeof name, "wave/lds NB=1 S=%u grid=%d", S, grid);
            timeit(name, [&] { hipLaunchKernelGGL((k_wave_lds<3, 1>), dim3(grid), dim3(256), 0, 0, in, out, nsteps, S); });
            snprintf(name, sizeof name, "wave/lds NB=2 S=%u grid=%d", S, grid);
            timeit(name, [&] { hipLaunchKernelGGL((k_wave_lds<3, 2>), dim3(grid), dim3(256), 0, 0, in, out, nsteps, S); });
            if (bpc <= 4) {
                snprintf(name, sizeof name, "wave/lds NB=3 S=%u grid=%d", S, grid);
                timeit(name, [&] { hipLaunchKernelGGL((k_wave_lds<3, 3>), dim3(grid), dim3(256), 0, 0, in, out, nsteps, S); });
            }
            if (bpc <= 3) {
                snprintf(name, sizeof name, "wave/lds NB=4 S=%u grid=%d", S, grid);
                timeit(name, [&] { hipLaunchKernelGGL((k_wave_lds<3, 4>), dim3(grid), dim3(256), 0, 0, in, out, nsteps, S); });
            }
        }
        if (S == 2400) {
            for (uint32_t logK : {0u, 2u, 4u, 6u}) {
                const uint64_t nseg = (nsteps + (4ull << logK) - 1) / (4ull << logK);
                for (int grid : {768, 1024, 1536, (int)nseg}) {
                    snprintf(name, sizeof name, "seg NB=1 K=%u grid=%d", 1u << logK, grid);
                    timeit(name, [&] { hipLaunchKernelGGL((k_wave_seg<3, 1>), dim3(grid), dim3(256), 0, 0, in, out, nsteps, S, logK); });
                    snprintf(name, sizeof name, "seg NB=2 K=%u grid=%d", 1u << logK, grid);
                    timeit(name, [&] { hipLaunchKernelGGL((k_wave_seg<3, 2>), dim3(grid), dim3(256), 0, 0, in, out, nsteps, S, logK); });
                    snprintf(name, sizeof name, "seg NB=3 K=%u grid=%d", 1u << logK, grid);
                    timeit(name, [&] { hipLaunchKernelGGL((k_wave_seg<3, 3>), dim3(grid), dim3(256), 0, 0, in, out, nsteps, S, logK); });
                }
            }
        }
        for (int bpc : {2, 3, 4}) {
            const int grid = 256 * bpc;
            snprintf(name, sizeof name, "wave/asm NB=1 S=%u grid=%d", S, grid);
            timeit(name, [&] { hipLaunchKernelGGL((k_wave_asm<3, 1>), dim3(grid), dim3(256), 0, 0, in, out, nsteps, S); });
            snprintf(name, sizeof name, "wave/asm NB=2 S=%u grid=%d", S, grid);
            timeit(name, [&] { hipLaunchKernelGGL((k_wave_asm<3, 2>), dim3(grid), dim3(256), 0, 0, in, out, nsteps, S); });
            snprintf(name, sizeof name, "wave/asm NB=3 S=%u grid=%d", S, grid);
            timeit(name, [&] { hipLaunchKernelGGL((k_wave_asm<3, 3>), dim3(grid), dim3(256), 0, 0, in, out, nsteps, S); });
            if (bpc <= 3) {
                snprintf(name, sizeof name, "wave/asm NB=4 S=%u grid=%d", S, grid);
                timeit(name, [&] { hipLaunchKernelGGL((k_wave_asm<3, 4>), dim3(grid), dim3(256), 0, 0, in, out, nsteps, S); });
            }
        }
    }

    // stretches that start on 128-byte lines (2304, 2432) against stretches that do not (2400, 2416): what the
    // partial lines at either end of a wavefront's run cost
    for (uint32_t S : {2304u, 2400u, 2416u, 2432u}) {
        const uint64_t nsteps = total / S;
        check_len = nsteps * S;
        const uint32_t logK = 1;
        const uint64_t nseg = (nsteps + (4ull << logK) - 1) / (4ull << logK);
        for (int rep = 0; rep < 2; rep++) {
            snprintf(name, sizeof name, "lines: seg NB=2 K=2 S=%u grid=%d", S, (int)nseg);
            timeit(name, [&] { hipLaunchKernelGGL((k_wave_seg<3, 2>), dim3((int)nseg), dim3(256), 0, 0, in, out, nsteps, S, logK); });
        }
    }
    if (argc > 1) { printf("done\n"); return 0; }
    {
        check_len = 0;
        const uint64_t nrec = 10000000, L = 150;
        std::vector<uint64_t> ho(nrec + 1);
        for (uint64_t i = 0; i <= nrec; i++) ho[i] = i * L;
        uint64_t* offs;
        CK(hipMalloc(&offs, (nrec + 1) * 8));
        CK(hipMemcpy(offs, ho.data(), (nrec + 1) * 8, hipMemcpyHostToDevice));
        const uint64_t padded = (total + 15) & ~15ull;
        const uint32_t GW = 16;
        for (int bpc : {4, 6, 8}) {
            const int grid = 256 * bpc;
            snprintf(name, sizeof name, "offs persistent GW=16 grid=%d", grid);
            timeit(name, [&] { hipLaunchKernelGGL((k_wave_offs<3, 1, true>), dim3(grid), dim3(256), 0, 0, in, out, offs, nrec, padded, GW); });
        }
        {
            const int grid = (int)((nrec + 4 * GW - 1) / (4 * GW));
            snprintf(name, sizeof name, "offs one step per wave GW=16 grid=%d", grid);
            timeit(name, [&] { hipLaunchKernelGGL((k_wave_offs<3, 1, false>), dim3(grid), dim3(256), 0, 0, in, out, offs, nrec, padded, GW); });
        }
        {
            const int grid = (int)((nrec + 8 * GW - 1) / (8 * GW));
            snprintf(name, sizeof name, "offs two steps per wave GW=16 grid=%d", grid);
            timeit(name, [&] { hipLaunchKernelGGL((k_wave_offs<3, 2, false>), dim3(grid), dim3(256), 0, 0, in, out, offs, nrec, padded, GW); });
        }
        {
            const int grid = (int)((nrec + 16 * GW - 1) / (16 * GW));
            snprintf(name, sizeof name, "offs four steps per wave GW=16 grid=%d", grid);
            timeit(name, [&] { hipLaunchKernelGGL((k_wave_offs<3, 4, false>), dim3(grid), dim3(256), 0, 0, in, out, offs, nrec, padded, GW); });
        }
    }
    printf("done\n");
    return 0;
}
