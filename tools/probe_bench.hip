// Calibration micro-benchmark (not part of the product): what does MI355X sustain for the access
// pattern of KmerSet::get -- independent random 4-byte loads from a table far larger than the
// 256 MiB Infinity Cache -- and for a plain streaming read of the same table?
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_bench.hip -o tools/probe_bench
// Run  : tools/probe_bench [table_GiB=16] [probes_per_thread=64]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int ILP>
__global__ __launch_bounds__(256) void rand_probe(const uint32_t *__restrict__ t, uint64_t nwords_mask, int iters,
                                                  uint32_t *sink)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t v[ILP];
#pragma unroll
        for (int j = 0; j < ILP; j++)
            v[j] = t[mix64(tid * 1315423911ull + (uint64_t)it * ILP + j) & nwords_mask];
#pragma unroll
        for (int j = 0; j < ILP; j++)
            acc ^= v[j];
    }
    if (acc == 0x12345u)
        *sink = acc;
}

__global__ __launch_bounds__(256) void stream_read(const uint4 *__restrict__ t, uint64_t nvec, uint32_t *sink)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const uint4 v = t[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u)
        *sink = acc;
}

template <int ILP>
static void run_rand(const uint32_t *d, uint64_t mask, int per_thread, uint32_t *sink, int blocks)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int iters = per_thread / ILP;
    rand_probe<ILP><<<blocks, 256>>>(d, mask, iters, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    rand_probe<ILP><<<blocks, 256>>>(d, mask, iters, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double probes = (double)blocks * 256 * iters * ILP;
    printf("rand_probe ILP=%-2d blocks=%-5d: %8.3f ms  %7.2f Gprobe/s  = %7.1f GB/s @64B/probe\n", ILP, blocks, ms,
           probes / ms / 1e6, probes * 64 / ms / 1e6);
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 16.0;
    const int per_thread = argc > 2 ? atoi(argv[2]) : 64;
    uint64_t nwords = 1;
    while (nwords * 2 * 4 <= (uint64_t)(gib * (1ull << 30)))
        nwords *= 2;
    uint32_t *d, *sink;
    CK(hipMalloc((void **)&d, nwords * 4));
    CK(hipMalloc((void **)&sink, 4));
    CK(hipMemset(d, 1, nwords * 4));
    printf("table %.2f GiB\n", nwords * 4.0 / (1ull << 30));
    for (int blocks : {2048, 4096, 8192}) {
        run_rand<1>(d, nwords - 1, per_thread, sink, blocks);
        run_rand<4>(d, nwords - 1, per_thread, sink, blocks);
        run_rand<8>(d, nwords - 1, per_thread, sink, blocks);
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    stream_read<<<4096, 256>>>((const uint4 *)d, nwords / 4, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    stream_read<<<4096, 256>>>((const uint4 *)d, nwords / 4, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("stream_read: %8.3f ms  %7.1f GB/s\n", ms, nwords * 4.0 / ms / 1e6);
    return 0;
}
