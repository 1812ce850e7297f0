// Feasibility micro-benchmark (not part of the product): probe rate of a LINE-bucketed set layout,
// where runs of SHARE adjacent lanes (consecutive k-mers of a read sharing a minimizer) read the SAME
// 64-byte line and every lane reads BYTES of it, against independent 4-byte probes (SHARE=1, BYTES=4).
// Build: hipcc -O3 --offload-arch=gfx950 tools/line_probe_bench.hip -o tools/line_probe_bench
// Run  : tools/line_probe_bench [table_MiB=256] [probes_per_thread=64]
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

template <int SHARE, int BYTES>
__global__ __launch_bounds__(256) void line_probe(const uint32_t *__restrict__ t, uint64_t nlines_mask, int iters, uint32_t *sink)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t run = tid / SHARE;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint64_t line = mix64(run * 1315423911ull + (uint64_t)it) & nlines_mask;
        const uint32_t *p = t + (BYTES > 64 ? (line & ~(uint64_t)(BYTES / 64 - 1)) : line) * 16;
        if (BYTES == 4) {
            acc ^= p[threadIdx.x & 15];
        } else {
            const uint4 *q = (const uint4 *)p;
#pragma unroll
            for (int j = 0; j < BYTES / 16; j++) {
                const uint4 v = q[j];
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    if (acc == 0x12345u)
        *sink = acc;
}

// the same probes with WORK dependent integer instructions between them (the lane automaton issues ~330 vector
// instructions per probe round): does the SHAPE of the requests -- one lane pulling its 64-byte line in four 16-byte
// pieces, or four neighbouring lanes pulling a quarter each -- matter once the memory pipe shares the CU with real work?
// COOP = 0: lane l pulls its own line in four 16-byte pieces (four wave instructions, 64 lines each);
// COOP = 1: in instruction e lane l pulls quarter l & 3 of the line of "owner" 16 e + (l >> 2) (four wave instructions,
//           16 lines each, the four lanes of a line side by side).  Either way a wave-round fetches 64 lines with 4
//           instructions and 64 B per lane.
template <int SHARE, int BYTES, int WORK>
__global__ __launch_bounds__(256, 7) void line_probe_mix(const uint32_t *__restrict__ t, uint64_t nlines_mask, int iters, uint32_t *sink)
{
    constexpr bool COOP = SHARE == 4;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t wave0 = tid & ~63ull;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t acc = 0, w = (uint32_t)tid | 1u;
    for (int it = 0; it < iters; it++) {
        uint4 v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint64_t owner = COOP ? wave0 + 16u * e + (lane >> 2) : tid;
            const uint64_t line = mix64(owner * 1315423911ull + (uint64_t)it) & nlines_mask;
            v[e] = ((const uint4 *)(t + line * 16))[COOP ? (lane & 3u) : (uint32_t)e];
        }
#pragma unroll
        for (int j = 0; j < WORK; j++)
            w = w * 0x9E3779B1u + (w >> 7); // (independent of the loads: issue work to overlap them)
#pragma unroll
        for (int e = 0; e < 4; e++)
            acc ^= v[e].x ^ v[e].y ^ v[e].z ^ v[e].w;
    }
    if ((acc ^ w) == 0x12345u)
        *sink = acc;
}

// The automaton's shape: ONE line per lane per round, ~WORK x 3 vector instructions that DEPEND on it before the next
// address is known (DEPTH = 1), through LDS like lane_kernel (LDS-DMA, then read back).  DEPTH = 2: the line of the round
// AFTER next is requested before this round's work starts (its address is known a round ahead: a second slot), LINE_B
// bytes per line (64: four 16-byte pieces, 32: two) -- what a compact index with two slots per lane would do.
template <int LINE_B, int DEPTH, int WORK>
__global__ __launch_bounds__(256, 7) void line_chain(const uint32_t *__restrict__ t, uint64_t nlines_mask, int iters, uint32_t *sink)
{
    __shared__ uint4 lc[DEPTH][LINE_B / 16][256];
    const uint32_t tid = threadIdx.x, wb = tid & ~63u;
    uint64_t x = ((uint64_t)blockIdx.x * 256u + tid) * 0x9E3779B97F4A7C15ull + 1u;
    uint32_t w = (uint32_t)x | 1u, acc = 0;
    auto issue = [&](int slot, uint64_t line) {
        const uint8_t *L = reinterpret_cast<const uint8_t *>(t + (line & nlines_mask) * 16ull);
#pragma unroll
        for (int e = 0; e < LINE_B / 16; e++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(L + 16 * e),
                                             (__attribute__((address_space(3))) void *)&lc[slot][e][wb], 16, 0, 0);
    };
    // the address sequence of a lane is a pure function of (lane, round): known one round ahead, like a scan's next line
    auto addr = [&](int it) { return mix64(x + (uint64_t)it); };
    if (DEPTH == 2)
        issue(0, addr(0));
    for (int it = 0; it < iters; it++) {
        const int cur = DEPTH == 2 ? (it & 1) : 0;
        if (DEPTH == 2)
            issue(cur ^ 1, addr(it + 1));
        else
            issue(0, addr(it));
        if (DEPTH == 2)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LINE_B / 16) : "memory"); // all but the youngest line's pieces
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t v = 0;
#pragma unroll
        for (int e = 0; e < LINE_B / 16; e++) {
            const uint4 q = lc[cur][e][tid];
            v ^= q.x ^ q.y ^ q.z ^ q.w;
        }
        w ^= v;
#pragma unroll
        for (int j = 0; j < WORK; j++)
            w = w * 0x9E3779B1u + (w >> 7); // (depends on the line: the round's work)
        acc += w;
    }
    if (acc == 0x12345u)
        *sink = acc;
}

// The same dependent rounds, every lane wanting a 64-byte line of its own, but FOUR NEIGHBOURING LANES pull a quarter each:
// in instruction e lane l fetches quarter l & 3 of the line of owner 16 e + (l >> 2) (the owner's address comes by
// ds_bpermute), so a wave instruction touches 16 lines as whole 64-byte requests instead of 64 lines as 16-byte pieces.
// The LDS-DMA puts lane l's 16 bytes at base + 16 l: owner o's line lies contiguous at lc[o >> 4][4 (o & 15) .. + 3], and
// the owner reads its quarters in an order rotated by its lane (bank conflicts between lanes 64 bytes apart otherwise).
template <int WORK>
__global__ __launch_bounds__(256, 7) void line_chain_coop(const uint32_t *__restrict__ t, uint64_t nlines_mask, int iters, uint32_t *sink)
{
    __shared__ uint4 lc[4][4][64]; // [wave of the block][instruction e][lane]
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    uint64_t x = ((uint64_t)blockIdx.x * 256u + tid) * 0x9E3779B97F4A7C15ull + 1u;
    uint32_t w = (uint32_t)x | 1u, acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint64_t line = mix64(x + (uint64_t)it) & nlines_mask;
        const uint32_t lo = (uint32_t)line, hi = (uint32_t)(line >> 32);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int owner = 16 * e + (int)(lane >> 2);
            const uint64_t ol = ((uint64_t)(uint32_t)__shfl((int)hi, owner) << 32) | (uint32_t)__shfl((int)lo, owner);
            const uint8_t *L = reinterpret_cast<const uint8_t *>(t + ol * 16ull) + 16u * (lane & 3u);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)L,
                                             (__attribute__((address_space(3))) void *)&lc[wv][e][0], 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t qd = ((uint32_t)j + (lane >> 2)) & 3u;
            const uint4 q = lc[wv][lane >> 4][4u * (lane & 15u) + qd];
            v ^= q.x ^ q.y ^ q.z ^ q.w;
        }
        w ^= v;
#pragma unroll
        for (int j = 0; j < WORK; j++)
            w = w * 0x9E3779B1u + (w >> 7);
        acc += w;
    }
    if (acc == 0x12345u)
        *sink = acc;
}

template <int WORK>
static void run_chain_coop(const uint32_t *d, uint64_t mask, int iters, uint32_t *sink)
{
    const int blocks = 256 * 7;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    line_chain_coop<WORK><<<blocks, 256>>>(d, mask, iters, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    line_chain_coop<WORK><<<blocks, 256>>>(d, mask, iters, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double probes = (double)blocks * 256 * iters;
    printf("chain 64-byte lines, four lanes a line,  %3d x 3 dependent instructions a round: %8.3f ms  %6.2f Gline/s  %5.2f us a round\n", WORK, ms,
           probes / ms / 1e6, ms * 1e3 / iters);
    fflush(stdout);
}

template <int LINE_B, int DEPTH, int WORK>
static void run_chain(const uint32_t *d, uint64_t mask, int iters, uint32_t *sink)
{
    const int blocks = 256 * 7; // the automaton's grid: seven waves per SIMD, every lane busy
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    line_chain<LINE_B, DEPTH, WORK><<<blocks, 256>>>(d, mask, iters, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    line_chain<LINE_B, DEPTH, WORK><<<blocks, 256>>>(d, mask, iters, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double probes = (double)blocks * 256 * iters;
    printf("chain %2d-byte lines, %d in flight per lane, %3d x 3 dependent instructions a round: %8.3f ms  %6.2f Gline/s  %5.2f us a round\n", LINE_B, DEPTH,
           WORK, ms, probes / ms / 1e6, ms * 1e3 / iters);
    fflush(stdout);
}

template <int SHARE, int BYTES, int WORK>
static void run_mix(const uint32_t *d, uint64_t mask, int iters, uint32_t *sink, int blocks)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    line_probe_mix<SHARE, BYTES, WORK><<<blocks, 256>>>(d, mask, iters, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    line_probe_mix<SHARE, BYTES, WORK><<<blocks, 256>>>(d, mask, iters, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double probes = (double)blocks * 256 * iters;
    printf("mix %s work=%-3d blocks=%-5d: %8.3f ms  %7.2f Gline/s  (%.1f G wave-instructions/s of work)\n",
           SHARE == 4 ? "four lanes per line  " : "one lane per line    ", WORK, blocks, ms, probes / ms / 1e6, probes / 64.0 * 3.0 * WORK / ms / 1e6);
    fflush(stdout);
}

template <int SHARE, int BYTES>
static void run(const uint32_t *d, uint64_t mask, int iters, uint32_t *sink, int blocks)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    line_probe<SHARE, BYTES><<<blocks, 256>>>(d, mask, iters, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    line_probe<SHARE, BYTES><<<blocks, 256>>>(d, mask, iters, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double probes = (double)blocks * 256 * iters;
    printf("share=%-2d bytes/lane=%-2d blocks=%-5d: %8.3f ms  %7.2f Gprobe/s  %7.2f Gline/s\n", SHARE, BYTES, blocks, ms,
           probes / ms / 1e6, probes / SHARE / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const double mib = argc > 1 ? atof(argv[1]) : 256.0;
    const int iters = argc > 2 ? atoi(argv[2]) : 64;
    uint64_t nlines = 1;
    while (nlines * 2 * 64 <= (uint64_t)(mib * (1ull << 20)))
        nlines *= 2;
    uint32_t *d, *sink;
    CK(hipMalloc((void **)&d, nlines * 64));
    CK(hipMalloc((void **)&sink, 4));
    CK(hipMemset(d, 1, nlines * 64));
    printf("table %.1f MiB (%llu lines)\n", nlines * 64.0 / (1 << 20), (unsigned long long)nlines);
    const int blocks = 8192;
    if (argc > 3 && argv[3][0] == 'c') { // the automaton's shape: dependent rounds
        run_chain<64, 1, 0>(d, nlines - 1, iters * 4, sink);
        run_chain<64, 1, 100>(d, nlines - 1, iters * 4, sink);
        run_chain<32, 1, 100>(d, nlines - 1, iters * 4, sink);
        run_chain<32, 2, 100>(d, nlines - 1, iters * 4, sink);
        run_chain<64, 1, 60>(d, nlines - 1, iters * 4, sink);
        run_chain<32, 1, 60>(d, nlines - 1, iters * 4, sink);
        run_chain<32, 2, 60>(d, nlines - 1, iters * 4, sink);
        run_chain<32, 2, 0>(d, nlines - 1, iters * 4, sink);
        run_chain_coop<0>(d, nlines - 1, iters * 4, sink);
        run_chain_coop<60>(d, nlines - 1, iters * 4, sink);
        run_chain_coop<100>(d, nlines - 1, iters * 4, sink);
        return 0;
    }
    if (argc > 3) { // mixed with work only
        run_mix<1, 64, 0>(d, nlines - 1, iters, sink, blocks);
        run_mix<4, 16, 0>(d, nlines - 1, iters, sink, blocks);
        run_mix<1, 64, 50>(d, nlines - 1, iters, sink, blocks);
        run_mix<4, 16, 50>(d, nlines - 1, iters, sink, blocks);
        run_mix<1, 64, 100>(d, nlines - 1, iters, sink, blocks);
        run_mix<4, 16, 100>(d, nlines - 1, iters, sink, blocks);
        run_mix<1, 64, 200>(d, nlines - 1, iters, sink, blocks);
        run_mix<4, 16, 200>(d, nlines - 1, iters, sink, blocks);
        return 0;
    }
    run<1, 4>(d, nlines - 1, iters, sink, blocks);
    run<1, 16>(d, nlines - 1, iters, sink, blocks);
    run<1, 64>(d, nlines - 1, iters, sink, blocks);
    run<2, 64>(d, nlines - 1, iters, sink, blocks);
    run<4, 4>(d, nlines - 1, iters, sink, blocks);
    run<4, 16>(d, nlines - 1, iters, sink, blocks);
    run<4, 32>(d, nlines - 1, iters, sink, blocks);
    run<4, 64>(d, nlines - 1, iters, sink, blocks);
    run<1, 128>(d, nlines - 1, iters, sink, blocks);
    run<4, 128>(d, nlines - 1, iters, sink, blocks);
    run<1, 256>(d, nlines - 1, iters, sink, blocks);
    run<4, 256>(d, nlines - 1, iters, sink, blocks);
    run<8, 16>(d, nlines - 1, iters, sink, blocks);
    run<8, 64>(d, nlines - 1, iters, sink, blocks);
    run<16, 64>(d, nlines - 1, iters, sink, blocks);
    return 0;
}
