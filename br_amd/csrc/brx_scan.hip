// Exclusive scan of per-read u32 lengths into u64 offsets (n+1 entries), on device.
// Three small launches: per-block sums, scan of the block sums, per-block rescan.
#include "brx_internal.hpp"

namespace {

constexpr int SCAN_IPT = 16;                // items per thread
constexpr int SCAN_ITEMS = 256 * SCAN_IPT;  // items per 256-thread block (the single-block scan of the block sums is
                                            // serial in the number of blocks: 4096 items keep it at a few dozen rounds)

__global__ __launch_bounds__(256) void lens_block_sum_kernel(const uint32_t *__restrict__ lens, uint32_t n,
                                                             uint64_t *__restrict__ block_sums)
{
    __shared__ unsigned long long sh[4];
    const uint32_t base = blockIdx.x * SCAN_ITEMS;
    unsigned long long acc = 0;
    for (uint32_t j = threadIdx.x; j < SCAN_ITEMS; j += 256) {
        const uint32_t idx = base + j;
        if (idx < n)
            acc += lens[idx];
    }
    for (int d = 32; d > 0; d >>= 1)
        acc += __shfl_down(acc, d);
    if ((threadIdx.x & 63) == 0)
        sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        block_sums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// one workgroup; nblocks is a few thousand at most (n / 1024)
__global__ __launch_bounds__(256) void block_sums_scan_kernel(uint64_t *block_sums, uint32_t nblocks,
                                                              unsigned long long *total)
{
    __shared__ unsigned long long sh[256];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0)
        carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 256) {
        const uint32_t idx = base + threadIdx.x;
        const unsigned long long mine = idx < nblocks ? block_sums[idx] : 0ull;
        sh[threadIdx.x] = mine;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const unsigned long long t = (threadIdx.x >= (unsigned)d) ? sh[threadIdx.x - d] : 0ull;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (idx < nblocks)
            block_sums[idx] = carry + sh[threadIdx.x] - mine;
        __syncthreads();
        if (threadIdx.x == 255)
            carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        *total = carry;
}

__global__ __launch_bounds__(256) void lens_scan_kernel(const uint32_t *__restrict__ lens, uint32_t n,
                                                        const uint64_t *__restrict__ block_sums,
                                                        uint64_t *__restrict__ out_offsets)
{
    __shared__ unsigned long long sh[256];
    const uint32_t base = blockIdx.x * SCAN_ITEMS + threadIdx.x * SCAN_IPT;
    uint32_t v[SCAN_IPT];
    unsigned long long mine = 0;
#pragma unroll
    for (int q = 0; q < SCAN_IPT; q++) {
        v[q] = (base + q < n) ? lens[base + q] : 0u;
        mine += v[q];
    }
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned long long t = (threadIdx.x >= (unsigned)d) ? sh[threadIdx.x - d] : 0ull;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    unsigned long long run = block_sums[blockIdx.x] + sh[threadIdx.x] - mine;
#pragma unroll
    for (int q = 0; q < SCAN_IPT; q++) {
        if (base + q <= n) // index n receives the grand total
            out_offsets[base + q] = run;
        run += v[q];
    }
}

} // namespace

namespace brx {

uint64_t scan_tmp_bytes(uint32_t n) { return (((uint64_t)n + SCAN_ITEMS) / SCAN_ITEMS + 1) * 8; }

int exclusive_scan_lens(const uint32_t *d_lens, uint32_t n, uint64_t *d_tmp, uint64_t *d_out_offsets,
                        unsigned long long *d_total, hipStream_t s)
{
    const uint32_t nblocks = (uint32_t)(((uint64_t)n + SCAN_ITEMS) / SCAN_ITEMS); // covers index n as well
    KernelTimer t("offsets_scan", s);
    lens_block_sum_kernel<<<nblocks, 256, 0, s>>>(d_lens, n, d_tmp);
    block_sums_scan_kernel<<<1, 256, 0, s>>>(d_tmp, nblocks, d_total);
    lens_scan_kernel<<<nblocks, 256, 0, s>>>(d_lens, n, d_tmp, d_out_offsets);
    BRX_HIP(hipGetLastError());
    return BRX_OK;
}

} // namespace brx
