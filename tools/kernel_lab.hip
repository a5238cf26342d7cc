// Standalone timing of the library's reverse kernels without Python or torch (experiments
// only): includes the kernel source directly so that build-time switches (-D...) apply.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iseqkit_amd/csrc -o tools/kernel_lab tools/kernel_lab.hip
#include "../seqkit_amd/csrc/kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const uint64_t nrec = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000000ull;
    const uint64_t L = argc > 2 ? strtoull(argv[2], nullptr, 10) : 150;
    const int random_data = argc > 3 ? atoi(argv[3]) : 1;
    const uint64_t total = nrec * L;
    uint8_t *in, *out;
    uint64_t* offs;
    uint32_t* flag;
    CK(hipMalloc(&in, total + 4096));
    CK(hipMalloc(&out, total + 4096));
    CK(hipMalloc(&offs, (nrec + 1) * 8));
    CK(hipMalloc(&flag, 256));
    {
        std::vector<uint64_t> ho(nrec + 1);
        for (uint64_t i = 0; i <= nrec; i++) ho[i] = i * L;
        CK(hipMemcpy(offs, ho.data(), (nrec + 1) * 8, hipMemcpyHostToDevice));
        std::vector<uint8_t> hs(total);
        uint64_t x = 88172645463325252ull;
        for (uint64_t i = 0; i < total; i++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            hs[i] = random_data ? "ACGT"[x & 3] : 'A';
        }
        CK(hipMemcpy(in, hs.data(), total, hipMemcpyHostToDevice));
    }
    CK(hipMemset(flag, 0, 256));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    skh::Table256 lut;
    for (int i = 0; i < 256; i++) lut.v[i] = (uint8_t)i;
    lut.v['A'] = 'T'; lut.v['T'] = 'A'; lut.v['C'] = 'G'; lut.v['G'] = 'C';
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 2; i++) launch();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; i++) launch();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 10;
        printf("%-48s %7.3f ms  %7.1f GB/s\n", name, ms, 2.0 * total / ms / 1e6);
        fflush(stdout);
    };
    char name[128];
    for (uint32_t steps : {1u, 2u, 3u, 4u, 6u}) {
        for (int with_lut : {0, 1}) {
            skh::LaunchCfg cfg{};
            cfg.logG = 2; cfg.max_grid = 0; cfg.n_cu = (uint32_t)prop.multiProcessorCount;
            cfg.staged = 1; cfg.staged_steps = steps; cfg.fuse_fastq = 0; cfg.span_u = 0;
            snprintf(name, sizeof name, "staged steps/wave=%u lut=%d", steps, with_lut);
            timeit(name, [&] {
                CK(skh::launch_reverse(cfg, in, out, nullptr, nullptr, offs, nrec, total, with_lut ? &lut : nullptr, flag, st));
            });
        }
    }
    for (uint32_t mg : {0u, 768u}) {
        for (int with_lut : {0, 1}) {
            skh::LaunchCfg cfg{};
            cfg.logG = 2; cfg.max_grid = mg; cfg.n_cu = (uint32_t)prop.multiProcessorCount;
            cfg.staged = 0; cfg.fuse_fastq = 0; cfg.span_u = 0;
            snprintf(name, sizeof name, "direct max_grid=%u lut=%d", mg, with_lut);
            timeit(name, [&] {
                CK(skh::launch_reverse(cfg, in, out, nullptr, nullptr, offs, nrec, total, with_lut ? &lut : nullptr, flag, st));
            });
        }
    }
    // involution spot check: out -> in2 must equal in
    return 0;
}
