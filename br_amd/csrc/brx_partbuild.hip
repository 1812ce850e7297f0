// Partitioned set build (BRX_COUNT_SORTED): the same bitset as count -> threshold, without the
// 2^(2k-1)-byte count table.
//
// Reference semantics: pcon Counter<u8>::count_fasta + Solid::from_count (src/main.rs:72-115):
// bit h is set iff min(255, occurrences of canonical hash h) > abundance.
//
// Why: random 1-byte RMWs into a 128 GiB table (k=19) run at the chip's random-request ceiling
// (~50 G requests/s, profiles/r1_probe_bench_calibration.txt) and drag two full passes over the
// table behind them (zeroing + thresholding).  Scattered 4-byte stores hit the same ceiling.  So
// the canonical hashes are PARTITIONED by their high bits, MSD-radix style, with streaming passes
// whose writes are staged through LDS into bucket-contiguous runs, until a bucket's hash range
// (2^12 hashes) fits a per-wavefront LDS counter array:
//
//   hash (2k-1 bits) = [ digit 1 | digit 2 | (digit 3) | low 12 bits ]      digits of <= 9 bits
//   level l   histogram pass (LDS histogram per tile, non-zero bins flushed with global atomics),
//             exclusive scan -> child bucket offsets, scatter pass: tile of 4-8 K keys, LDS
//             histogram + block scan, keys ranked into an LDS staging buffer, one global atomic
//             per (tile, bucket) reserves the run, runs written out with consecutive lanes on
//             consecutive addresses.  Level 1 reads the bases (k-mers are re-derived, never stored
//             whole); the last level writes the low 12 bits as u16.
//   final     one wavefront per fine bucket: u16 LDS counters (2 per word), bucket offsets fetched
//             64 at a time and keys two buckets ahead, count -> min(255,.) > abundance -> the
//             bucket's 512-byte slice of the bitset, written whole (no zeroing pass of the bitset).
//
// Order inside a bucket is irrelevant (we only count): no stable ranking, no per-tile prefix
// matrices.  HBM traffic for k=19: ~30 B per k-mer + one 16 GiB bitset write, instead of 129 B
// per k-mer + 2 x 128 GiB.
#include "brx_internal.hpp"
#include "brx_index.hpp"

#include <stdlib.h>
#include <string.h>

namespace brx {
uint64_t scan_tmp_bytes(uint32_t n);
int exclusive_scan_lens(const uint32_t *d_lens, uint32_t n, uint64_t *d_tmp, uint64_t *d_out_offsets,
                        unsigned long long *d_total, hipStream_t s);
}

using namespace brx;

namespace {

constexpr int F_BITS = 12;      // hashes per fine bucket = 4096 (8 KiB of u16 counters per wave)
constexpr int MAX_LEVELS = 4;
constexpr int MAX_DIGIT_BITS = 9;
// (the scans below keep per-thread partial sums in fixed-size arrays: ln_colscan_kernel tot[2] / run[2] covers 512 digits
// with 256 threads, col_scan_digits_kernel one wave's 16 digits per lane covers 1024; col_scan_columns_kernel v[8] covers
// 2048 chunks, which part_add_batch guarantees by doubling the chunk length)
static_assert(MAX_DIGIT_BITS <= 9, "the digit scans of the partition passes are written for at most 512 digits");

struct Plan {
    int k, nbits, nlev;
    int bits[MAX_LEVELS];        // digit width of each level (may be 0 = pass-through)
    int rem_in[MAX_LEVELS];      // key bits entering the level
    uint64_t nchild[MAX_LEVELS]; // number of buckets after the level
};

Plan make_plan(int k)
{
    Plan p;
    memset(&p, 0, sizeof(p));
    p.k = k;
    p.nbits = 2 * k - 1;
    const int P = p.nbits - F_BITS; // bits to partition away
    p.nlev = (P + MAX_DIGIT_BITS - 1) / MAX_DIGIT_BITS;
    if (p.nlev < 2)
        p.nlev = 2;
    int left = P, rem = p.nbits;
    uint64_t nb = 1;
    for (int l = 0; l < p.nlev; l++) {
        int b = (left + (p.nlev - l) - 1) / (p.nlev - l); // spread evenly, larger digits first
        if (l == 0 && p.nbits - b > 32)
            b = p.nbits - 32;                              // what is left after level 1 travels as u32 keys
        p.bits[l] = b;
        p.rem_in[l] = rem;
        rem -= b;
        left -= b;
        nb <<= b;
        p.nchild[l] = nb;
    }
    return p;
}

// ---- 2-bit packing of a window of bases into LDS -------------------------------------------------------
// word w holds window-local bases 16w..16w+15, first base in the top bits, so a k-mer is a 2k-bit
// field of the big-endian bit string.  pack4: ASCII bytes b0..b3 (b0 at the lowest address) ->
// (c0<<6 | c1<<4 | c2<<2 | c3), c = (b >> 1) & 3.
__device__ __forceinline__ uint32_t pack4(uint32_t w)
{
    uint32_t x = __builtin_bswap32((w >> 1) & 0x03030303u);
    x |= x >> 6;
    return (x | (x >> 12)) & 0xffu;
}

#ifndef BRX_L1_UNROLL
#define BRX_L1_UNROLL 4
#endif
#ifndef BRX_XCD_ITEMS
#define BRX_XCD_ITEMS 1
#endif
// -DBRX_DEBUG_BOUNDS=1 (tools/ab_build.sh): every scatter store of the partition passes is checked against the size of
// its destination; an index outside it is counted in bounds_violations (brx_debug_bounds_violations()) and the store
// is dropped.  For builds that leave phases out to time the rest ("results wrong, time only"): such a build has
// undefined offsets, and on a shared pool an undefined offset must not become a wild store (profiles/r2j_ab_runs.txt,
// ab13).  The shipped build compiles none of it.
#ifndef BRX_DEBUG_BOUNDS
#define BRX_DEBUG_BOUNDS 0
#endif
#if BRX_DEBUG_BOUNDS
__device__ unsigned long long bounds_violations = 0ull;
__device__ __forceinline__ bool store_ok(uint64_t idx, uint64_t cap)
{
    if (idx >= cap) {
        atomicAdd(&bounds_violations, 1ull);
        return false;
    }
    return true;
}
#else
__device__ __forceinline__ bool store_ok(uint64_t, uint64_t) { return true; }
#endif
constexpr bool XCD_ITEMS = BRX_XCD_ITEMS != 0;
constexpr uint32_t L1_TILE = 4096;                    // k-mer start positions per level-1 work item = 16 per thread (round 1: 8192 -> scatter 5.4 ms, 4096 -> 4.3, 2048 -> 5.9 at 1 Gbp; the scatter now stages 16 KB of keys + 10 KB of counters: 5 blocks per CU)
constexpr uint32_t PACK_WORDS = L1_TILE / 16 + 4;     // bases of the tile + k - 1 + slack, 16 per word
constexpr uint32_t BND_WORDS = (L1_TILE + 64) / 32 + 2; // read-boundary bitmap of the same window

// packs bases [g0, g0 + nbases) of the batch into pk[]; bases beyond `total` read as A
__device__ __forceinline__ void pack_window(const uint8_t *__restrict__ bases, uint64_t total, uint64_t g0, uint32_t nbases,
                                            uint32_t *__restrict__ pk)
{
    const uint32_t nwords = (nbases + 15u) / 16u;
    for (uint32_t w = threadIdx.x; w < nwords; w += 256) {
        const uint64_t p = g0 + 16ull * w;
        uint32_t v[4] = {0, 0, 0, 0};
        if (p + 16 <= total) {
            uint4 q;
            __builtin_memcpy(&q, bases + p, 16); // one (possibly unaligned) global_load_dwordx4
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            for (uint32_t j = 0; j < 16; j++)
                if (p + j < total)
                    v[j >> 2] |= (uint32_t)bases[p + j] << (8u * (j & 3u));
        }
        pk[w] = (pack4(v[0]) << 24) | (pack4(v[1]) << 16) | (pack4(v[2]) << 8) | pack4(v[3]);
    }
}

// forward k-mer starting at window-local base `pos`
__device__ __forceinline__ uint64_t kmer_at(const uint32_t *__restrict__ pk, uint32_t pos, int k)
{
    const uint32_t w = pos >> 4, sh = 2u * (pos & 15u);
    const uint64_t hi = ((uint64_t)pk[w] << 32) | pk[w + 1];
    const uint64_t lo = pk[w + 2];
    const uint64_t val = sh ? ((hi << sh) | (lo >> (32u - sh))) : hi;
    return val >> (64 - 2 * k);
}

// ---- which work items a workgroup takes (levels >= 2; level 1 gives a block a contiguous range: l1_range) ------------
// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one: MI355X_MICROARCH.md, Workgroup
// dispatch), and tile t + 1 of a scatter appends each digit's keys right behind tile t's -- 32-byte runs at 8 keys per
// digit.  With the plain grid-stride loop neighbouring tiles sit on different XCDs, so every 128-byte line of the output
// is written in pieces from four L2s that cannot merge them.  Here the blocks that share an XCD take one contiguous
// eighth of the items and walk it side by side, so the pieces of a line meet in one L2 and leave as whole lines.
// Speed only: every item is taken exactly once whatever the placement really is.
struct ItemRange {
    unsigned long long first, end, step;
};
__device__ __forceinline__ ItemRange xcd_items(unsigned long long n_items)
{
    const unsigned long long G = gridDim.x, b = blockIdx.x;
    if (G < 8ull || !XCD_ITEMS)
        return {b, n_items, G};
    const unsigned long long x = b & 7ull, j = b >> 3;
    const unsigned long long R = (n_items + 7ull) >> 3;
    const unsigned long long lo = x * R;
    unsigned long long hi = lo + R;
    if (hi > n_items)
        hi = n_items;
    return {lo + j, hi, (G - x + 7ull) >> 3}; // blocks with label x: x, x + 8, ... < G
}

// ---- level 1: tiles of the flat base stream ---------------------------------------------------------------
// Work item i = k-mer start positions [i*L1_TILE, (i+1)*L1_TILE) of the concatenated batch.  A k-mer is
// valid iff it does not run over a read boundary: bnd[] has a bit for every read START (and for the
// end of the batch) inside the window; position p is valid iff no bit is set in (p, p+k-1].
struct L1Args {
    const uint8_t *bases;
    const uint64_t *offsets; // n_reads + 1 (absolute indices into bases; the flat stream starts at offsets[0])
    uint32_t n_reads;
    uint64_t total;          // bases in the batch
    uint32_t n_items;
    int k, nbits, bits;      // digit = hash >> (nbits - bits)
    uint32_t *matrix;        // [n_items][B] per-tile digit counts (hist): a tile's B counters are one coalesced row ...
    const uint64_t *pos;     // ... and, same layout, where the tile's keys of each digit go (col_scan_* below)
    uint32_t *keys_out;      // hash with the digit stripped
    uint64_t out_cap;        // entries of keys_out (checked under BRX_DEBUG_BOUNDS)
};

// sh_r0[0] = the first read that starts after the tile's first position, upper_bound(offsets, g0): searched for when
// `search` (17 dependent loads by one thread for 100 000 reads -- about 10 us with everybody else waiting, most of a
// tile's time when every tile did it), otherwise carried over from the tile before, which counts in sh_r0[1] the reads
// that start at or before ITS last position while it marks its boundaries: blocks take their tiles in rising order.
__device__ __forceinline__ void l1_prepare(const L1Args &a, uint32_t item, uint32_t *pk, uint32_t *bnd, uint32_t *sh_r0,
                                           uint32_t &n_here, bool search)
{
    const uint64_t off0 = a.offsets[0];
    const uint64_t g0 = (uint64_t)item * L1_TILE;
    const uint64_t left = a.total - g0;
    n_here = (uint32_t)(left < L1_TILE ? left : L1_TILE);
    const uint32_t win = n_here + (uint32_t)a.k + 32u; // bases needed (+ slack for the 3-word extract)
    pack_window(a.bases + off0, a.total, g0, win, pk);
    for (uint32_t w = threadIdx.x; w < BND_WORDS; w += 256)
        bnd[w] = 0;
    if (threadIdx.x == 0) {
        if (search) {
            uint32_t lo = 0, hi = a.n_reads + 1;
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (a.offsets[mid] - off0 <= g0)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            sh_r0[0] = lo;
        } else {
            sh_r0[0] += sh_r0[1];
        }
        sh_r0[1] = 0;
    }
    __syncthreads();
    const uint64_t lim = g0 + n_here + (uint64_t)a.k; // boundaries at window-local positions <= n_here + k - 1 matter
    uint32_t before_next = 0;
    for (uint32_t r = sh_r0[0] + threadIdx.x; r <= a.n_reads; r += 256) {
        const uint64_t o = a.offsets[r] - off0; // offsets[n_reads] - off0 == total: the batch end is a boundary too
        if (o >= lim)
            break;
        const uint32_t p = (uint32_t)(o - g0);
        atomicOr(&bnd[p >> 5], 1u << (p & 31u));
        before_next += o <= g0 + L1_TILE ? 1u : 0u;
    }
    if (before_next)
        atomicAdd(&sh_r0[1], before_next);
    __syncthreads();
}

// a block's tiles: a contiguous range, taken in rising order
__device__ __forceinline__ void l1_range(uint32_t n_items, uint32_t &lo, uint32_t &hi)
{
    lo = (uint32_t)(((uint64_t)blockIdx.x * n_items) / gridDim.x);
    hi = (uint32_t)(((uint64_t)(blockIdx.x + 1) * n_items) / gridDim.x);
}

// The same preparation, one tile ahead.  A tile's global loads -- its bases, the first read offsets behind its start, its
// row of counts -- are three round trips to memory with everybody waiting (the second depends on sh_r0), and a block has
// only its own four waves to hide them behind: measured, most of a tile's 13 us.  l1_fetch() issues them for the NEXT
// tile into registers as soon as sh_r0 of the current one is known; l1_consume() is l1_prepare() fed from them.
struct L1Next {
    uint4 q0, q1;  // 16 bases at window word t, and at word 256 + t (the k + 32 bases behind the tile)
    uint64_t off;  // offsets[r_first + t] - offsets[0]
    uint32_t m[(1u << MAX_DIGIT_BITS) / 256u]; // matrix[item][t + 256 q] (scatter only)
};

__device__ __forceinline__ uint32_t l1_window_words(const L1Args &a, uint32_t item, uint32_t &n_here)
{
    const uint64_t left = a.total - (uint64_t)item * L1_TILE;
    n_here = (uint32_t)(left < L1_TILE ? left : L1_TILE);
    return (n_here + (uint32_t)a.k + 32u + 15u) / 16u; // (+ slack for the 3-word extract)
}

template <bool ROW>
__device__ __forceinline__ void l1_fetch(const L1Args &a, uint32_t item, uint32_t r_first, L1Next &nx)
{
    const uint64_t off0 = a.offsets[0];
    const uint8_t *bases = a.bases + off0;
    const uint64_t g0 = (uint64_t)item * L1_TILE;
    uint32_t n_here;
    const uint32_t nwords = l1_window_words(a, item, n_here);
    const uint64_t p0 = g0 + 16ull * threadIdx.x, p1 = p0 + 16ull * 256ull;
    nx.q0 = make_uint4(0, 0, 0, 0);
    nx.q1 = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < nwords && p0 + 16 <= a.total)
        __builtin_memcpy(&nx.q0, bases + p0, 16); // one (possibly unaligned) global_load_dwordx4
    if (256u + threadIdx.x < nwords && p1 + 16 <= a.total)
        __builtin_memcpy(&nx.q1, bases + p1, 16);
    const uint32_t r = r_first + threadIdx.x;
    nx.off = a.offsets[r <= a.n_reads ? r : a.n_reads] - off0;
    if (ROW) {
        const uint32_t B = 1u << a.bits;
#pragma unroll
        for (uint32_t q = 0; q < (1u << MAX_DIGIT_BITS) / 256u; q++) {
            const uint32_t b = threadIdx.x + 256u * q;
            nx.m[q] = b < B ? a.matrix[(uint64_t)item * B + b] : 0u;
        }
    }
}

__device__ __forceinline__ uint32_t l1_pack16(const L1Args &a, const uint4 &q, uint64_t p)
{
    uint32_t v[4] = {q.x, q.y, q.z, q.w};
    if (p + 16 > a.total) { // the batch ends inside these 16 bases (its last tile only): byte by byte, beyond the end reads as A
        const uint8_t *bases = a.bases + a.offsets[0];
        v[0] = v[1] = v[2] = v[3] = 0;
        for (uint32_t j = 0; j < 16; j++)
            if (p + j < a.total)
                v[j >> 2] |= (uint32_t)bases[p + j] << (8u * (j & 3u));
    }
    return (pack4(v[0]) << 24) | (pack4(v[1]) << 16) | (pack4(v[2]) << 8) | pack4(v[3]);
}

// nx = l1_fetch(item, sh_r0[0] + sh_r0[1] of the tile before); `first`: sh_r0[0] was searched for, sh_r0[1] == 0
__device__ __forceinline__ void l1_consume(const L1Args &a, uint32_t item, const L1Next &nx, uint32_t *pk, uint32_t *bnd, uint32_t *sh_r0,
                                           uint32_t &n_here, bool first)
{
    const uint64_t off0 = a.offsets[0];
    const uint64_t g0 = (uint64_t)item * L1_TILE;
    const uint32_t nwords = l1_window_words(a, item, n_here);
    if (threadIdx.x < nwords)
        pk[threadIdx.x] = l1_pack16(a, nx.q0, g0 + 16ull * threadIdx.x);
    if (256u + threadIdx.x < nwords)
        pk[256u + threadIdx.x] = l1_pack16(a, nx.q1, g0 + 16ull * (256u + threadIdx.x));
    for (uint32_t w = threadIdx.x; w < BND_WORDS; w += 256)
        bnd[w] = 0;
    if (threadIdx.x == 0 && !first) {
        sh_r0[0] += sh_r0[1];
        sh_r0[1] = 0;
    }
    __syncthreads();
    const uint64_t lim = g0 + n_here + (uint64_t)a.k; // boundaries at window-local positions <= n_here + k - 1 matter
    uint32_t before_next = 0;
    uint64_t o = nx.off;
    for (uint32_t r = sh_r0[0] + threadIdx.x; r <= a.n_reads; r += 256) {
        if (o >= lim)
            break;
        const uint32_t p = (uint32_t)(o - g0);
        atomicOr(&bnd[p >> 5], 1u << (p & 31u));
        before_next += o <= g0 + L1_TILE ? 1u : 0u;
        if (r + 256u <= a.n_reads)
            o = a.offsets[r + 256u] - off0;
    }
    if (before_next)
        atomicAdd(&sh_r0[1], before_next);
    __syncthreads();
}

// upper_bound(offsets, first position of tile `item`) into sh_r0[0], 0 into sh_r0[1]; ends with a barrier
__device__ __forceinline__ void l1_search(const L1Args &a, uint32_t item, uint32_t *sh_r0)
{
    if (threadIdx.x == 0) {
        const uint64_t off0 = a.offsets[0], g0 = (uint64_t)item * L1_TILE;
        uint32_t lo = 0, hi = a.n_reads + 1;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (a.offsets[mid] - off0 <= g0)
                lo = mid + 1;
            else
                hi = mid;
        }
        sh_r0[0] = lo;
        sh_r0[1] = 0;
    }
    __syncthreads();
}

// true iff no read boundary lies in window-local positions (p, p + k - 1]
__device__ __forceinline__ bool l1_valid(const uint32_t *bnd, uint32_t p, int k)
{
    const uint32_t q = p + 1u; // first position to test, k-1 <= 30 bits
    const uint32_t w = q >> 5, sh = q & 31u;
    const uint64_t two = ((uint64_t)bnd[w + 1] << 32) | bnd[w];
    const uint32_t bits = (uint32_t)(two >> sh) & ((k > 1) ? ((1u << (k - 1)) - 1u) : 0u);
    return bits == 0u;
}

// Every valid k-mer of the tile, rolled: thread t takes window positions 16t .. 16t+15 (L1_TILE = 256 x 16), builds the
// first k-mer and its reverse complement once and then shifts one base in per step -- f(canonical hash) -- instead of
// cutting every k-mer out of pk[] (three LDS reads, a funnel shift, a 64-bit bit reversal) and testing the boundary
// bitmap with two more LDS reads per position.  Both level-1 kernels are bound by the instructions they issue.
static_assert(L1_TILE == 256u * 16u, "l1_for_each_hash: 16 positions per thread");
template <int UNROLL = 4, typename F>
__device__ __forceinline__ void l1_for_each_hash(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ bnd, uint32_t n_here,
                                                 int k, F &&f)
{
    const uint32_t p0 = threadIdx.x * 16u;
    if (p0 >= n_here)
        return;
    const uint32_t w = threadIdx.x; // p0 / 16
    const uint64_t hi = ((uint64_t)pk[w] << 32) | pk[w + 1];
    const uint64_t lo = pk[w + 2];
    const uint64_t mask = kmask(k);
    uint64_t fwd = hi >> (64 - 2 * k);
    uint64_t rc = revcomp(fwd, k);
    // the bases behind the first k-mer, the next one in the top two bits (the 96-bit string hi:lo shifted left by 2k)
    uint64_t next = (hi << (2 * k)) | (2 * k >= 32 ? lo << (2 * k - 32) : lo >> (32 - 2 * k));
    // bit i = a read starts at window position p0 + 1 + i; position p0 + j is valid iff bits j .. j + k - 2 are clear
    const uint32_t q = p0 + 1u, wq = q >> 5, sh = q & 31u;
    uint64_t vb = ((((uint64_t)bnd[wq + 1] << 32) | bnd[wq]) >> sh) | (sh ? (uint64_t)bnd[wq + 2] << (64u - sh) : 0ull);
    const uint64_t vmask = k > 1 ? (1ull << (k - 1)) - 1ull : 0ull;
    const uint32_t left = n_here - p0 < 16u ? n_here - p0 : 16u;
    const int top = 2 * k - 2;
#pragma unroll UNROLL
    for (uint32_t j = 0; j < left; j++) {
        if ((vb & vmask) == 0ull)
            f(((popc64(fwd) & 1) ? rc : fwd) >> 1);
        const uint64_t b = next >> 62;
        next <<= 2;
        vb >>= 1;
        fwd = ((fwd << 2) | b) & mask;
        rc = (rc >> 2) | ((b ^ 2ull) << top);
    }
}

#ifndef BRX_L1HIST_WAVES
#define BRX_L1HIST_WAVES 1 // (1 = the compiler's choice: 76 registers, six waves; tools/ab_build.sh sweeps it)
#endif
__global__ __launch_bounds__(256, BRX_L1HIST_WAVES) void l1_hist_kernel(L1Args a)
{
    extern __shared__ uint32_t lds[]; // hist[B]
    __shared__ uint32_t pk[PACK_WORDS];
    __shared__ uint32_t bnd[BND_WORDS];
    __shared__ uint32_t sh_r0[2];
    const uint32_t B = 1u << a.bits;
    const int shift = a.nbits - a.bits;
    uint32_t lo, hi;
    l1_range(a.n_items, lo, hi);
    if (lo >= hi)
        return;
    l1_search(a, lo, sh_r0);
    L1Next nx;
    l1_fetch<false>(a, lo, sh_r0[0], nx);
    for (uint32_t item = lo; item < hi; item++) {
        for (uint32_t b = threadIdx.x; b < B; b += 256)
            lds[b] = 0;
        uint32_t n_here;
        l1_consume(a, item, nx, pk, bnd, sh_r0, n_here, item == lo);
        if (item + 1 < hi)
            l1_fetch<false>(a, item + 1, sh_r0[0] + sh_r0[1], nx);
        l1_for_each_hash(pk, bnd, n_here, a.k, [&](uint64_t h) { atomicAdd(&lds[(uint32_t)(h >> shift)], 1u); });
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < B; b += 256)
            a.matrix[(uint64_t)item * B + b] = lds[b];
        __syncthreads();
    }
}

// ---- positions from the [n_items][B] count matrix -------------------------------------------------------------------
// pos[item][b] = (keys of all digits < b) + (keys of digit b in the tiles before `item`): a scan down the COLUMNS of a
// tile-major matrix.  Tile-major because a tile then reads and writes its B counters as one coalesced row -- digit-major
// ([B][n_items], one flat exclusive scan) made every tile touch B separate cache lines, twice in the histogram pass and
// twice in the scatter: 0.25 random line accesses per key, which at the platform's ~55 G lines/s was most of level 1.
// Three small passes over chunks of CH tiles: column sums per chunk, one block turning them into chunk bases (and the
// digit offsets the next level needs), then the positions.
__global__ __launch_bounds__(256) void col_scan_partial_kernel(const uint32_t *__restrict__ m, uint32_t n_items, uint32_t B, uint32_t CH,
                                                               uint64_t *__restrict__ part)
{
    const uint32_t chunk = blockIdx.x;
    const uint32_t lo = chunk * CH, hi = (lo + CH < n_items) ? lo + CH : n_items;
    for (uint32_t b = threadIdx.x; b < B; b += 256) {
        uint64_t sum = 0;
        for (uint32_t it = lo; it < hi; it++)
            sum += m[(uint64_t)it * B + b];
        part[(uint64_t)chunk * B + b] = sum;
    }
}

// block b: exclusive scan of column b of part[n_chunks][B] in place (n_chunks <= 2048: 8 per thread), its total -> coltot[b]
__global__ __launch_bounds__(256) void col_scan_columns_kernel(uint64_t *__restrict__ part, uint32_t n_chunks, uint32_t B,
                                                               uint64_t *__restrict__ coltot)
{
    __shared__ uint64_t wsum[4];
    const uint32_t b = blockIdx.x;
    const uint32_t per = (n_chunks + 255u) / 256u; // <= 8
    const uint32_t c0 = threadIdx.x * per;
    uint64_t v[8];
    uint64_t mine = 0;
    for (uint32_t q = 0; q < per && q < 8u; q++) {
        const uint32_t c = c0 + q;
        v[q] = c < n_chunks ? part[(uint64_t)c * B + b] : 0;
        mine += v[q];
    }
    uint64_t incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t t = __shfl_up(incl, d);
        if ((threadIdx.x & 63) >= (unsigned)d)
            incl += t;
    }
    if ((threadIdx.x & 63) == 63)
        wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t base = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++)
        base += wsum[w];
    uint64_t run = base + incl - mine;
    for (uint32_t q = 0; q < per && q < 8u; q++) {
        const uint32_t c = c0 + q;
        if (c < n_chunks)
            part[(uint64_t)c * B + b] = run;
        run += v[q];
    }
    if (threadIdx.x == 255)
        coltot[b] = base + incl;
}

// one wave: coff[b] = keys of all digits < b, coff[B] = *total = number of keys (B <= 1024)
__global__ __launch_bounds__(64) void col_scan_digits_kernel(const uint64_t *__restrict__ coltot, uint32_t B, uint64_t *__restrict__ coff,
                                                             unsigned long long *__restrict__ total)
{
    uint64_t carry = 0;
    for (uint32_t b0 = 0; b0 < B; b0 += 64) {
        const uint32_t b = b0 + threadIdx.x;
        const uint64_t v = b < B ? coltot[b] : 0;
        uint64_t incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t t = __shfl_up(incl, d);
            if (threadIdx.x >= (unsigned)d)
                incl += t;
        }
        if (b < B)
            coff[b] = carry + incl - v;
        carry += __shfl(incl, 63);
    }
    if (threadIdx.x == 0) {
        coff[B] = carry;
        *total = carry;
    }
}

__global__ __launch_bounds__(256) void col_scan_positions_kernel(const uint32_t *__restrict__ m, uint32_t n_items, uint32_t B, uint32_t CH,
                                                                 const uint64_t *__restrict__ part, const uint64_t *__restrict__ coff,
                                                                 uint64_t *__restrict__ pos)
{
    const uint32_t chunk = blockIdx.x;
    const uint32_t lo = chunk * CH, hi = (lo + CH < n_items) ? lo + CH : n_items;
    for (uint32_t b = threadIdx.x; b < B; b += 256) {
        uint64_t run = part[(uint64_t)chunk * B + b] + coff[b];
        for (uint32_t it = lo; it < hi; it++) {
            pos[(uint64_t)it * B + b] = run;
            run += m[(uint64_t)it * B + b];
        }
    }
}

// presence-only insertion over the FLAT base stream (Pcon::from_fasta / Hash::from_fasta): the same tiles and
// read-boundary bitmap as the level-1 partition, so one 250 Mbp record keeps the whole chip busy instead of
// one workgroup.  TABLE = false: atomicOr into the bit vector; TABLE = true: find-or-insert into the chained
// table of a sparse set (n_new counts the k-mers that were not there yet).
template <bool TABLE>
__global__ __launch_bounds__(256) void flat_insert_kernel(L1Args a, uint32_t *__restrict__ bits, uint64_t *__restrict__ lines,
                                                          uint32_t line_shift, uint32_t m, uint32_t w,
                                                          unsigned long long *__restrict__ n_new)
{
    __shared__ uint32_t pk[PACK_WORDS];
    __shared__ uint32_t bnd[BND_WORDS];
    __shared__ uint32_t sh_r0[2];
    uint32_t added = 0;
    uint32_t lo, hi;
    l1_range(a.n_items, lo, hi);
    for (uint32_t item = lo; item < hi; item++) {
        uint32_t n_here;
        l1_prepare(a, item, pk, bnd, sh_r0, n_here, item == lo);
        for (uint32_t p = threadIdx.x; p < n_here; p += 256)
            if (l1_valid(bnd, p, a.k)) {
                const uint64_t kmer = kmer_at(pk, p, a.k);
                if (TABLE) {
                    added += table_find_or_insert(lines, line_shift, m, w, a.k, kmer) ? 1u : 0u;
                } else {
                    const uint64_t h = khash(kmer, a.k);
                    atomicOr(bits + (h >> 5), 1u << (h & 31u));
                }
            }
        __syncthreads();
    }
    if (TABLE) {
        for (int d = 32; d > 0; d >>= 1)
            added += __shfl_down(added, d);
        if ((threadIdx.x & 63) == 0 && added)
            atomicAdd(n_new, (unsigned long long)added);
    }
}

// shared tail of the scatter kernels: block scan of cntv -> lofs/lcur, returns nothing (all in LDS)
__device__ __forceinline__ void block_scan_bins(uint32_t B, const uint32_t *cntv, uint32_t *lofs, uint32_t *lcur, uint32_t *sh_wsum)
{
    const uint32_t BPT = (B + 255u) / 256u; // bins per thread (1 or 2)
    uint32_t mine = 0;
    for (uint32_t q = 0; q < BPT; q++) {
        const uint32_t b = threadIdx.x * BPT + q;
        if (b < B)
            mine += cntv[b];
    }
    uint32_t incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if ((threadIdx.x & 63) >= (unsigned)d)
            incl += t;
    }
    if ((threadIdx.x & 63) == 63)
        sh_wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++)
        wbase += sh_wsum[w];
    uint32_t run = wbase + incl - mine;
    for (uint32_t q = 0; q < BPT; q++) {
        const uint32_t b = threadIdx.x * BPT + q;
        if (b < B) {
            lofs[b] = run;
            lcur[b] = run;
            run += cntv[b];
        }
    }
}

// Level-1 write-out in whole 32-byte sectors.  A tile leaves T / B = 8 keys per digit on average: written where they
// belong, that is a 32-byte run at a 4-byte-aligned place, 1.87 sectors touched per run (WRITE_SIZE counted exactly that,
// profiles/r2j_*), every one of them a partial write.  But the runs of one digit from CONSECUTIVE tiles lie end to end
// (pos[t + 1][d] = pos[t][d] + count[t][d]), so a block that takes a contiguous range of tiles can hold back the keys
// behind each digit's last sector boundary (a carry of < 8 keys per digit, in LDS) and write them with the next tile's:
// only the two ends of a block's range are partial.  The first sector of a range starts with the `ph` places that belong
// to the range before (phantoms: kept as holes in the carry, never stored).
constexpr uint32_t L1_SECTOR = 8;                              // keys per 32-byte sector
constexpr uint32_t L1_OWN = ((1u << MAX_DIGIT_BITS) + 255u) / 256u; // digits per lane: t, t + 256

#ifndef BRX_L1SCAT_WAVES
#define BRX_L1SCAT_WAVES 5
#endif
__global__ __launch_bounds__(256, BRX_L1SCAT_WAVES) void l1_scatter_kernel(L1Args a)
{
    // LDS: stage_key[T] | gsm[B] (u64: sector index | keys carried << 32 | phantoms << 36) | cnt[B] | lofs[B] | lcur[B]
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    __shared__ uint32_t pk[PACK_WORDS];
    __shared__ uint32_t bnd[BND_WORDS];
    __shared__ uint32_t sh_r0[2];
    __shared__ uint32_t sh_wsum[4];
    const uint32_t T = L1_TILE;
    const uint32_t B = 1u << a.bits;
    uint32_t *stage_key = (uint32_t *)lds_raw;
    unsigned long long *gsm = (unsigned long long *)(stage_key + T);
    uint32_t *cntv = (uint32_t *)(gsm + B);
    uint32_t *lofs = cntv + B;
    uint32_t *lcur = lofs + B;
    const int shift = a.nbits - a.bits;
    const uint64_t child_mask = (1ull << shift) - 1ull;
    uint32_t lo, hi;
    l1_range(a.n_items, lo, hi);
    if (lo >= hi)
        return;
    for (uint32_t b = threadIdx.x; b < B; b += 256) {
        const unsigned long long p0 = a.pos[(uint64_t)lo * B + b];
        const unsigned long long ph = p0 & (L1_SECTOR - 1);
        gsm[b] = (p0 / L1_SECTOR) | (ph << 32) | (ph << 36);
    }
    uint32_t creg[L1_OWN][L1_SECTOR]; // the carries of this lane's digits
#pragma unroll
    for (uint32_t q = 0; q < L1_OWN; q++)
#pragma unroll
        for (uint32_t jj = 0; jj < L1_SECTOR; jj++)
            creg[q][jj] = 0;
    l1_search(a, lo, sh_r0);
    L1Next nx;
    l1_fetch<true>(a, lo, sh_r0[0], nx);
    for (uint32_t item = lo; item < hi; item++) {
#pragma unroll
        for (uint32_t q = 0; q < L1_OWN; q++)
            if (threadIdx.x + 256u * q < B)
                cntv[threadIdx.x + 256u * q] = nx.m[q]; // the histogram pass already counted this tile
        uint32_t n_here;
        l1_consume(a, item, nx, pk, bnd, sh_r0, n_here, item == lo);
        if (item + 1 < hi)
            l1_fetch<true>(a, item + 1, sh_r0[0] + sh_r0[1], nx);
        block_scan_bins(B, cntv, lofs, lcur, sh_wsum);
        __syncthreads();
        l1_for_each_hash<BRX_L1_UNROLL>(pk, bnd, n_here, a.k, [&](uint64_t h) {
            const uint32_t d = (uint32_t)(h >> shift);
            const uint32_t slot = atomicAdd(&lcur[d], 1u);
            stage_key[slot] = (uint32_t)(h & child_mask);
        });
        __syncthreads();
        // one digit per lane: carry + the tile's keys leave as whole sectors (two 16-byte stores), the rest is the new carry
#pragma unroll
        for (uint32_t q = 0; q < L1_OWN; q++) {
            const uint32_t d = threadIdx.x + 256u * q;
            if (d < B) {
                const unsigned long long m = gsm[d];
                const uint32_t n = cntv[d], base = lofs[d];
                const uint32_t gsx = (uint32_t)m, cc = (uint32_t)(m >> 32) & 15u, ph = (uint32_t)(m >> 36) & 15u;
                const uint32_t tot = cc + n, full = tot / L1_SECTOR, r = tot % L1_SECTOR;
                const uint32_t src = base - cc; // element e >= cc of the run is stage_key[src + e]
                uint32_t *out = a.keys_out + (unsigned long long)gsx * L1_SECTOR;
                for (uint32_t sct = 0; sct < full; sct++) {
                    uint32_t v[L1_SECTOR];
#pragma unroll
                    for (uint32_t jj = 0; jj < L1_SECTOR; jj++) {
                        const uint32_t e = L1_SECTOR * sct + jj;
                        v[jj] = e < cc ? creg[q][jj] : stage_key[src + e];
                    }
                    if (store_ok((unsigned long long)(gsx + sct) * L1_SECTOR + 7u, a.out_cap)) {
                        if (sct == 0 && ph) { // the first sector of the range: its first places are the range before's
#pragma unroll
                            for (uint32_t jj = 0; jj < L1_SECTOR; jj++)
                                if (jj >= ph)
                                    out[jj] = v[jj];
                        } else {
                            uint4 *o4 = reinterpret_cast<uint4 *>(out + L1_SECTOR * sct);
                            o4[0] = make_uint4(v[0], v[1], v[2], v[3]);
                            o4[1] = make_uint4(v[4], v[5], v[6], v[7]);
                        }
                    }
                }
                if (full) {
#pragma unroll
                    for (uint32_t jj = 0; jj < L1_SECTOR; jj++)
                        creg[q][jj] = jj < r ? stage_key[src + L1_SECTOR * full + jj] : 0u;
                    gsm[d] = (unsigned long long)(gsx + full) | ((unsigned long long)r << 32);
                } else {
#pragma unroll
                    for (uint32_t jj = 0; jj < L1_SECTOR; jj++)
                        creg[q][jj] = jj < cc ? creg[q][jj] : (jj < r ? stage_key[src + jj] : 0u);
                    gsm[d] = (unsigned long long)gsx | ((unsigned long long)r << 32) | ((unsigned long long)ph << 36);
                }
            }
        }
        __syncthreads();
    }
    // the end of the range: what is still held back
#pragma unroll
    for (uint32_t q = 0; q < L1_OWN; q++) {
        const uint32_t d = threadIdx.x + 256u * q;
        if (d < B) {
            const unsigned long long m = gsm[d];
            const uint32_t cc = (uint32_t)(m >> 32) & 15u, ph = (uint32_t)(m >> 36) & 15u;
            const unsigned long long g = (unsigned long long)(uint32_t)m * L1_SECTOR;
#pragma unroll
            for (uint32_t jj = 0; jj < L1_SECTOR; jj++)
                if (jj >= ph && jj < cc && store_ok(g + jj, a.out_cap))
                    a.keys_out[g + jj] = creg[q][jj];
        }
    }
}

// ---- levels >= 2: tiles inside parent buckets -----------------------------------------------------------------
__global__ void tiles_from_offsets_kernel(const uint64_t *__restrict__ poff, uint64_t n_parents, uint32_t tile,
                                          uint32_t *__restrict__ ntiles)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_parents)
        return;
    const uint64_t n = poff[p + 1] - poff[p];
    ntiles[p] = (uint32_t)((n + tile - 1) / tile);
}

// item -> parent map (instead of a binary search per work item)
__global__ void fill_item_parent_kernel(const uint64_t *__restrict__ item_off, uint64_t n_parents,
                                        uint32_t *__restrict__ item_parent)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_parents)
        return;
    const uint64_t lo = item_off[p], hi = item_off[p + 1];
    for (uint64_t i = lo; i < hi; i++)
        item_parent[i] = (uint32_t)p;
}

struct LnArgs {
    const uint32_t *keys_in;
    const uint64_t *poff;        // parent bucket offsets into keys_in
    uint64_t n_parents;
    const uint64_t *item_off;    // n_parents + 1: first work item of each parent
    const uint32_t *item_parent; // per work item
    const unsigned long long *n_items; // device scalar
    uint32_t tile;
    int rem_in, bits;            // digit = key >> (rem_in - bits)
    uint32_t *matrix;            // [parent][B][tiles of the parent] digit counts, index B*item_off[p] + b*ntiles_p + t
    const uint64_t *pos;         // exclusive scan of matrix
    void *keys_out;
    uint64_t in_cap, out_cap;    // entries of keys_in / keys_out (checked under BRX_DEBUG_BOUNDS)
};

template <int KPT>
__device__ __forceinline__ uint32_t ln_load(const LnArgs &a, uint64_t parent, uint64_t t, uint32_t (&key)[KPT])
{
    const uint64_t lo = a.poff[parent] + t * a.tile;
    const uint64_t hi_p = a.poff[parent + 1];
    const uint64_t hi = (lo + a.tile < hi_p) ? lo + a.tile : hi_p;
    uint32_t cnt = 0;
#pragma unroll
    for (int q = 0; q < KPT; q++) {
        const uint64_t i = lo + (uint64_t)q * 256 + threadIdx.x;
        if (i < hi && store_ok(i, a.in_cap)) {
            key[q] = a.keys_in[i];
            cnt = q + 1;
        }
    }
    return cnt;
}

template <int KPT>
#ifndef BRX_LNHIST_WAVES
#define BRX_LNHIST_WAVES 1 // (72 registers, seven waves)
#endif
__global__ __launch_bounds__(256, BRX_LNHIST_WAVES) void ln_hist_kernel(LnArgs a)
{
    extern __shared__ uint32_t lds[]; // hist[B]
    const uint32_t B = 1u << a.bits;
    const int shift = a.rem_in - a.bits;
    const unsigned long long n_items = *a.n_items;
    const ItemRange ir = xcd_items(n_items);
    for (unsigned long long item = ir.first; item < ir.end; item += ir.step) {
        const uint64_t parent = a.item_parent[item];
        const uint64_t io = a.item_off[parent];
        const uint64_t t = item - io;
        for (uint32_t b = threadIdx.x; b < B; b += 256)
            lds[b] = 0;
        __syncthreads();
        uint32_t key[KPT];
        const uint32_t cnt = ln_load<KPT>(a, parent, t, key);
#pragma unroll
        for (int q = 0; q < KPT; q++)
            if ((uint32_t)q < cnt)
                atomicAdd(&lds[(key[q] >> shift) & (B - 1u)], 1u);
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < B; b += 256)
            a.matrix[B * (io + t) + b] = lds[b]; // tile-major inside the parent: one coalesced row per tile
        __syncthreads();
    }
}

// positions of a level >= 2 from its tile-major counts: block p scans the columns of parent p's [ntiles][B] rows.
// A parent's keys stay inside the parent's range, so pos = poff[p] + (keys of the parent's digits < b) + (keys of digit
// b in the parent's tiles before t); coff[p*B + b] = where child (p, b) starts, coff[n_parents*B] = total.
__global__ __launch_bounds__(256) void ln_colscan_kernel(const uint32_t *__restrict__ m, const uint64_t *__restrict__ poff,
                                                         const uint64_t *__restrict__ item_off, uint64_t n_parents, uint32_t B,
                                                         uint64_t *__restrict__ pos, uint64_t *__restrict__ coff)
{
    __shared__ uint64_t wsum[4];
    for (uint64_t p = blockIdx.x; p < n_parents; p += gridDim.x) {
        const uint64_t io = item_off[p], nt = item_off[p + 1] - io;
        const uint32_t BPT = (B + 255u) / 256u; // digits per thread (1 or 2)
        uint64_t tot[2] = {0, 0};
        for (uint64_t t = 0; t < nt; t++)
            for (uint32_t q = 0; q < BPT; q++) {
                const uint32_t b = threadIdx.x * BPT + q;
                if (b < B)
                    tot[q] += m[B * (io + t) + b];
            }
        const uint64_t mine = tot[0] + tot[1];
        uint64_t incl = mine;
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t u = __shfl_up(incl, d);
            if ((threadIdx.x & 63) >= (unsigned)d)
                incl += u;
        }
        __syncthreads(); // (wsum of the previous parent has been read)
        if ((threadIdx.x & 63) == 63)
            wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint64_t base = poff[p];
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++)
            base += wsum[w];
        uint64_t run[2];
        run[0] = base + incl - mine;
        run[1] = run[0] + tot[0];
        for (uint32_t q = 0; q < BPT; q++) {
            const uint32_t b = threadIdx.x * BPT + q;
            if (b < B)
                coff[p * B + b] = run[q];
        }
        for (uint64_t t = 0; t < nt; t++)
            for (uint32_t q = 0; q < BPT; q++) {
                const uint32_t b = threadIdx.x * BPT + q;
                if (b < B) {
                    pos[B * (io + t) + b] = run[q];
                    run[q] += m[B * (io + t) + b];
                }
            }
        if (p == n_parents - 1 && threadIdx.x == 0)
            coff[n_parents * B] = poff[n_parents];
    }
}

template <int KPT, typename OUT>
__global__ __launch_bounds__(256) void ln_scatter_kernel(LnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    __shared__ uint32_t sh_wsum[4];
    const uint32_t T = a.tile;
    const uint32_t B = 1u << a.bits;
    uint32_t *stage_key = (uint32_t *)lds_raw;
    uint16_t *stage_dig = (uint16_t *)(lds_raw + (size_t)T * 4);
    unsigned long long *gbase = (unsigned long long *)(lds_raw + (size_t)T * 6);
    uint32_t *cntv = (uint32_t *)(lds_raw + (size_t)T * 6 + (size_t)B * 8);
    uint32_t *lofs = cntv + B;
    uint32_t *lcur = lofs + B;
    const int shift = a.rem_in - a.bits;
    const uint32_t child_mask = (1u << shift) - 1u;
    OUT *out = (OUT *)a.keys_out;
    const unsigned long long n_items = *a.n_items;
    const ItemRange ir = xcd_items(n_items);
    for (unsigned long long item = ir.first; item < ir.end; item += ir.step) {
        const uint64_t parent = a.item_parent[item];
        const uint64_t io = a.item_off[parent];
        const uint64_t t = item - io;
        for (uint32_t b = threadIdx.x; b < B; b += 256) {
            const uint64_t mi = B * (io + t) + b;
            cntv[b] = a.matrix[mi];
            gbase[b] = a.pos[mi];
        }
        uint32_t key[KPT];
        const uint32_t cnt = ln_load<KPT>(a, parent, t, key);
        __syncthreads();
        block_scan_bins(B, cntv, lofs, lcur, sh_wsum);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < KPT; q++)
            if ((uint32_t)q < cnt) {
                const uint32_t d = (key[q] >> shift) & (B - 1u);
                const uint32_t slot = atomicAdd(&lcur[d], 1u);
                stage_key[slot] = key[q] & child_mask;
                stage_dig[slot] = (uint16_t)d;
            }
        __syncthreads();
        const uint32_t n_tile = lofs[B - 1] + cntv[B - 1];
        for (uint32_t idx = threadIdx.x; idx < n_tile; idx += 256) {
            const uint32_t d = stage_dig[idx];
            const uint64_t at = gbase[d] + (idx - lofs[d]);
            if (store_ok(at, a.out_cap))
                out[at] = (OUT)stage_key[idx];
        }
        __syncthreads();
    }
}

// ---- final stage: one wavefront per fine bucket (4096 hashes) -----------------------------------------
constexpr int P3_WAVES = 4;
constexpr uint32_t F_SIZE = 1u << F_BITS;   // 4096 hashes
constexpr uint32_t CNT_WORDS = F_SIZE / 2;  // u16 counters, 2 per word: hash h -> half (h>>11) of word (h & 2047)
constexpr uint32_t BIT_WORDS = F_SIZE / 32; // 128
constexpr uint32_t CHUNK_MAX = 65000;       // keys counted between two clamps (the u16 halves must not carry)

constexpr uint32_t STAGE_KEYS = 704;        // keys of a group of buckets staged in LDS per wave (4 blocks per CU)

constexpr uint32_t EMIT_CAP = 256;          // solid hashes collected per wave between two appends to the key list

// EMIT: also append every solid hash to a list (emit_keys, capacity emit_cap, counter emit_n keeps counting
// past the capacity so the host can tell a truncated list).  The list feeds the probe index
// (brx_index.hpp) without a second pass over the 2^(2k-1)-bit vector.
// BITS: write the bucket's 512-byte slice of the bit vector (not for sparse sets: 2^(2k-1) bits do not fit at k = 21).
template <bool EMIT, bool BITS>
__global__ __launch_bounds__(64 * P3_WAVES) void final_count_kernel(const uint16_t *__restrict__ keys,
                                                                     const uint64_t *__restrict__ off, uint64_t n_buckets,
                                                                     uint32_t abundance, uint32_t *__restrict__ bits,
                                                                     uint64_t *__restrict__ emit_keys, uint64_t emit_cap,
                                                                     unsigned long long *__restrict__ emit_n)
{
    __shared__ uint32_t cnt_all[P3_WAVES][CNT_WORDS];
    __shared__ uint32_t bit_all[P3_WAVES][BIT_WORDS];
    __shared__ __attribute__((aligned(8))) uint16_t stage_all[P3_WAVES][STAGE_KEYS + 8];
    __shared__ uint64_t emit_all[EMIT ? P3_WAVES : 1][EMIT ? EMIT_CAP : 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t *cnt = cnt_all[wave];
    uint32_t *bmp = bit_all[wave];
    uint16_t *stage = stage_all[wave];
    uint64_t *ebuf = emit_all[EMIT ? wave : 0];
    uint32_t ecount = 0; // wave-uniform
    auto emit_flush = [&]() {
        if (ecount) {
            unsigned long long base = 0;
            if (lane == 0)
                base = atomicAdd(emit_n, (unsigned long long)ecount);
            base = __shfl(base, 0);
            asm volatile("" ::: "memory");
            for (uint32_t q = lane; q < ecount; q += 64)
                if (base + q < emit_cap)
                    emit_keys[base + q] = ebuf[q];
            asm volatile("" ::: "memory");
            ecount = 0;
        }
    };
    for (uint32_t w = lane; w < CNT_WORDS; w += 64)
        cnt[w] = 0;
    for (uint32_t w = lane; w < BIT_WORDS; w += 64)
        bmp[w] = 0;
    // appends `hkey` of every lane whose flag is set (called by the whole wave)
    auto emit_if = [&](bool flag, uint64_t hkey) {
        const uint64_t bal = __ballot(flag);
        if (bal) {
            if (ecount + 64u > EMIT_CAP)
                emit_flush();
            if (flag)
                ebuf[ecount + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = hkey;
            ecount += (uint32_t)__builtin_popcountll(bal);
        }
    };
    asm volatile("" ::: "memory");

    // consecutive buckets go to one wave, so its key stream is contiguous
    const uint64_t n_waves = (uint64_t)gridDim.x * P3_WAVES;
    const uint64_t wid = (uint64_t)blockIdx.x * P3_WAVES + wave;
    const uint64_t per = (n_buckets + n_waves - 1) / n_waves;
    const uint64_t b_lo = wid * per;
    const uint64_t b_hi = (b_lo + per < n_buckets) ? b_lo + per : n_buckets;

    // groups of up to 63 buckets: one coalesced load brings the 64 offsets that delimit them, one
    // batch of coalesced loads stages all their keys in LDS; the per-bucket loop then runs on LDS only
    uint64_t g0 = b_lo;
    while (g0 < b_hi) {
        int gmax = (int)((b_hi - g0 < 63) ? b_hi - g0 : 63);
        const uint64_t oi = g0 + (uint64_t)lane;
        const uint64_t offv = off[oi <= n_buckets ? oi : n_buckets];
        const uint64_t gs = __shfl(offv, 0);
        // largest prefix of buckets whose keys fit the staging buffer (lane l: does bucket l-1 end inside?)
        const bool fits = lane >= 1 && lane <= gmax && (offv - gs) <= (uint64_t)STAGE_KEYS;
        const uint64_t fm = __ballot(fits) >> 1; // bit j: buckets 0..j fit
        int gcount = (fm == ~0ull >> 1) ? 63 : __builtin_ctzll(~fm);
        if (gcount > gmax)
            gcount = gmax;
        const bool staged = gcount > 0;
        if (!staged)
            gcount = 1; // a single bucket larger than the staging buffer: streamed from global memory
        const uint64_t ge = __shfl(offv, gcount);
        const uint64_t base4 = gs & ~3ull; // 8-byte aligned start of the staged window
        if (staged) {
            const uint32_t nquads = (uint32_t)((ge - base4 + 3) / 4);
            const uint2 *src = reinterpret_cast<const uint2 *>(keys + base4);
            uint2 *dstq = reinterpret_cast<uint2 *>(stage);
            for (uint32_t q = lane; q < nquads; q += 64)
                dstq[q] = src[q];
            asm volatile("" ::: "memory");
        }
        const uint32_t off_lo = (uint32_t)offv, off_hi = (uint32_t)(offv >> 32);
        for (int j = 0; j < gcount; j++) {
            // j is wave-uniform: v_readlane instead of a bpermute through the LDS pipe
            // (the builtin returns a signed int: without the casts an offset with bit 31 set sign-extends over the
            // high word -- every job beyond 2^31 keys)
            const uint64_t s = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(off_hi, j) << 32) |
                               (uint64_t)(uint32_t)__builtin_amdgcn_readlane(off_lo, j);
            const uint64_t e = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(off_hi, j + 1) << 32) |
                               (uint64_t)(uint32_t)__builtin_amdgcn_readlane(off_lo, j + 1);
            const uint64_t n = e - s;
            bool any_bit = false;
            if (n == 0) {
                // empty bucket: a slice of zeros
            } else if (staged && n <= 64) {
                const bool act = (uint64_t)lane < n;
                const uint32_t key = act ? stage[(uint32_t)(s - base4) + lane] : 0u;
                const uint32_t word = key & (CNT_WORDS - 1u), half = key >> (F_BITS - 1);
                bool hot = false, cross = false;
                if (act) {
                    // the lane that increments a counter last sees its final value minus one, so
                    // "old + 1 > abundance" on every lane sets exactly the bits of the final counts
                    const uint32_t old = (atomicAdd(&cnt[word], 1u << (16u * half)) >> (16u * half)) & 0xffffu;
                    const uint32_t c = old + 1u;
                    hot = (c > 255u ? 255u : c) > abundance; // pcon's u8 counters saturate at 255
                    cross = old == abundance && abundance < 255u; // exactly one lane per solid hash sees the crossing
                    if (BITS && hot)
                        atomicOr(&bmp[key >> 5], 1u << (key & 31u));
                }
                if (BITS)
                    any_bit = __ballot(hot) != 0ull;
                else if (EMIT)
                    emit_if(cross, (g0 + (uint64_t)j) * F_SIZE + (uint64_t)key); // no slice: list the hash right away
                asm volatile("" ::: "memory");
                if (act)
                    cnt[word] = 0;
            } else {
                any_bit = true;
                // large bucket: stream it, clamping between chunks so that the packed halves never carry
                uint64_t done = 0;
                while (done < n) {
                    const uint64_t m = (n - done < CHUNK_MAX) ? n - done : CHUNK_MAX;
                    for (uint64_t i = lane; i < m; i += 64) {
                        const uint32_t kk = staged ? (uint32_t)stage[(uint32_t)(s - base4 + done + i)] : (uint32_t)keys[s + done + i];
                        atomicAdd(&cnt[kk & (CNT_WORDS - 1u)], 1u << (16u * (kk >> (F_BITS - 1))));
                    }
                    asm volatile("" ::: "memory");
                    done += m;
                    if (done < n) {
                        for (uint32_t w = lane; w < CNT_WORDS; w += 64) {
                            const uint32_t v = cnt[w];
                            const uint32_t a0 = v & 0xffffu, a1 = v >> 16;
                            cnt[w] = (a0 > 255u ? 255u : a0) | ((a1 > 255u ? 255u : a1) << 16);
                        }
                        asm volatile("" ::: "memory");
                    }
                }
                for (uint32_t w = lane; w < CNT_WORDS; w += 64) { // (wave-uniform trip count)
                    const uint32_t v = cnt[w];
                    // word w holds hashes w (low half) and w + 2048 (high half)
                    const uint32_t c0 = v & 0xffffu, c1 = v >> 16;
                    const bool s0 = (c0 > 255u ? 255u : c0) > abundance, s1 = (c1 > 255u ? 255u : c1) > abundance;
                    if (BITS) {
                        if (s0)
                            atomicOr(&bmp[w >> 5], 1u << (w & 31u));
                        if (s1)
                            atomicOr(&bmp[(w + CNT_WORDS) >> 5], 1u << (w & 31u));
                    } else if (EMIT) {
                        emit_if(s0, (g0 + (uint64_t)j) * F_SIZE + (uint64_t)w);
                        emit_if(s1, (g0 + (uint64_t)j) * F_SIZE + (uint64_t)(w + CNT_WORDS));
                    }
                    if (v)
                        cnt[w] = 0;
                }
            }
            asm volatile("" ::: "memory");
            if (!BITS)
                continue; // hashes were listed above, there is no slice to write
            // flush the slice: 128 words = 2 per lane (LDS is only touched when a bit was set)
            uint32_t *dst = bits + (g0 + (uint64_t)j) * BIT_WORDS;
            uint32_t w0 = 0, w1 = 0;
            if (any_bit) {
                w0 = bmp[2 * lane];
                w1 = bmp[2 * lane + 1];
                bmp[2 * lane] = 0;
                bmp[2 * lane + 1] = 0;
            }
            *reinterpret_cast<uint2 *>(dst + 2 * lane) = make_uint2(w0, w1);
            asm volatile("" ::: "memory");
            if (EMIT && any_bit) {
                // one hash per lane and turn (slices hold a handful of bits: usually a single turn)
                uint64_t rem = (uint64_t)w0 | ((uint64_t)w1 << 32);
                const uint64_t hbase = (g0 + (uint64_t)j) * F_SIZE + 64ull * (uint64_t)lane;
                for (;;) {
                    const uint64_t bal = __ballot(rem != 0ull);
                    if (!bal)
                        break;
                    if (ecount + 64u > EMIT_CAP)
                        emit_flush();
                    if (rem) {
                        const uint32_t pos = ecount + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                        ebuf[pos] = hbase + (uint64_t)__builtin_ctzll(rem);
                        rem &= rem - 1ull;
                    }
                    ecount += (uint32_t)__builtin_popcountll(bal);
                }
            }
        }
        g0 += (uint64_t)gcount;
    }
    if (EMIT)
        emit_flush();
}

// ---- final stage without a bit vector: hash-count the buckets one level earlier -------------------------------
// When the set is only listed (lazy bit vector / sparse set) nothing forces the last radix level: a bucket of
// 2^R hashes (R <= 20 bits left in the keys) with a few thousand keys is counted directly in an LDS hash table
// (open addressing, entry = valid | key << 11 | count), and the entries with count > abundance are listed.  That
// replaces the last level's histogram + scatter and the per-4096-hash counting pass.  A TEAM of 1024 lanes (one
// workgroup; buckets of 640 keys and more on average) or 64 lanes (one wave) takes a bucket; a bucket with more keys than the table can take is done in
// 2^lp passes over disjoint key ranges.  lp is chosen for the DISTINCT keys expected -- n x the ratio hf_sample_kernel
// measured on a few buckets: at 20x coverage a bucket of 48 000 keys has ~10 000 different ones -- and a pass whose
// keys still do not fit is repeated with the whole table, then 4x finer.  Each bucket's table has 2.5 slots per distinct key
// expected, not the launch's maximum.  What binds the kernel and what was tried: DESIGN.md section 4,
// profiles/r2_one_kernel_ab.txt.
#ifndef BRX_HF_FLAT
#define BRX_HF_FLAT 1
#endif
constexpr bool HF_FLAT = BRX_HF_FLAT != 0;
constexpr uint32_t HF_VALID = 0x80000000u;
constexpr int HF_CHUNK = 8;
constexpr uint32_t HF_MAX_TRIES = 192; // probes before a key calls the table full (a cluster that long means it nearly is)
__host__ __device__ constexpr uint32_t hf_ebuf(int team) { return team > 64 ? 1024u : 512u; }

template <int TEAM, int BLOCK>
__global__ __launch_bounds__(BLOCK) void hash_final_kernel(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ off,
                                                         uint64_t n_buckets, int R, uint32_t abundance, uint32_t log_t,
                                                         uint64_t *__restrict__ emit_keys, uint64_t emit_cap,
                                                         unsigned long long *__restrict__ emit_n,
                                                         const unsigned long long *__restrict__ dup, uint32_t min_lt, float ratio_forced)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t hf_lds[];
    constexpr int TEAMS = BLOCK / TEAM;
    constexpr uint32_t EB = hf_ebuf(TEAM);
    const uint32_t MIN_LT = min_lt; // smallest table a bucket gets (hf_min_lt: 4 slots per lane and clearing step)
    const uint32_t Tmax = 1u << log_t;
    const int team = threadIdx.x / TEAM, tl = threadIdx.x % TEAM;
    uint32_t *tab = (uint32_t *)hf_lds + (size_t)team * Tmax;
    unsigned long long *ebuf = (unsigned long long *)(hf_lds + (size_t)TEAMS * Tmax * 4) + (size_t)team * EB;
    uint32_t *ctl = (uint32_t *)(hf_lds + (size_t)TEAMS * Tmax * 4 + (size_t)TEAMS * EB * 8) + team * 4; // [0] ecount, [1] overflow
    auto team_sync = [&]() {
        if (TEAM > 64)
            __syncthreads();
        else
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    auto flush = [&]() { // whole team
        team_sync();
        const uint32_t ne = ctl[0] < EB ? ctl[0] : EB;
        __shared__ unsigned long long sh_base[TEAMS];
        if (tl == 0)
            sh_base[team] = ne ? atomicAdd(emit_n, (unsigned long long)ne) : 0ull;
        team_sync();
        const unsigned long long base = sh_base[team];
        for (uint32_t q = tl; q < ne; q += TEAM)
            if (base + q < emit_cap)
                emit_keys[base + q] = ebuf[q];
        team_sync();
        if (tl == 0)
            ctl[0] = 0;
        team_sync();
    };
    if (tl == 0) {
        ctl[0] = 0;
        ctl[1] = 0;
    }
    team_sync();
    // distinct keys expected per key of a bucket (1 when nothing was sampled), with a quarter of room on top
    float ratio = 1.0f;
    if (dup) {
        const float d = (float)dup[0], t = (float)dup[1];
        if (t > 0.0f)
            ratio = fminf(1.0f, 1.25f * d / t + 0.01f);
    }
    if (ratio_forced > 0.0f) // test hook (BRX_HF_RATIO): a wrong guess must only cost time
        ratio = ratio_forced;
    const uint64_t n_teams = (uint64_t)gridDim.x * TEAMS;
    for (uint64_t b = (uint64_t)blockIdx.x * TEAMS + team; b < n_buckets; b += n_teams) { // (team-uniform trip count)
        const uint64_t s0 = off[b], n = off[b + 1] - s0;
        if (n == 0)
            continue;
        // passes: 2^lp disjoint ranges of the R-bit key, each expected to bring <= 5/8 T DISTINCT keys (measured: 3/8 costs
        // configs[3]'s share 39.9 -> 45.7 ms in extra passes, 7/8 costs configs[2] 83 -> 117 ms in overflowing tables; 4
        // slots per key instead of 2.5 below: no difference, 1.5: +23 % -- profiles/r2_one_kernel_ab.txt)
        const uint64_t n_eff = (uint64_t)((float)n * ratio) + 1ull;
        uint32_t lp = 0;
        while (lp < (uint32_t)R && (n_eff >> lp) > (uint64_t)(Tmax / 8u * 5u))
            lp++;
        // the table of THIS bucket: 2.5 slots per distinct key expected, not the launch's maximum -- clearing and scanning
        // cost T / TEAM LDS accesses per lane whatever the bucket holds (k = 21: 745 keys, ~200 distinct, in 2048 slots)
        uint32_t lt = log_t;
        if (lp == 0) {
            const uint32_t want = (uint32_t)n_eff * 5u / 2u;
            lt = MIN_LT < log_t ? MIN_LT : log_t;
            while (lt < log_t && (1u << lt) < want)
                lt++;
        }
        uint32_t T = 1u << lt;
        for (uint32_t pass = 0; pass < (1u << lp);) {
            for (uint32_t q = tl * 4u; q < T; q += TEAM * 4u) // 16 bytes per lane and store (T >= 4 * TEAM)
                *(uint4 *)(tab + q) = make_uint4(0u, 0u, 0u, 0u);
            team_sync();
            for (uint64_t i0 = 0; i0 < n; i0 += (uint64_t)TEAM * HF_CHUNK) {
              // HF_CHUNK independent loads in flight per thread, then the inserts
              uint32_t kbuf[HF_CHUNK];
#pragma unroll
              for (int c = 0; c < HF_CHUNK; c++) {
                  const uint64_t i = i0 + (uint64_t)c * TEAM + tl;
                  kbuf[c] = i < n ? keys[s0 + i] : 0xffffffffu; // (a clamped index instead of the branch: 3.39 -> 5.24 ms)
              }
              if constexpr (HF_FLAT && TEAM > 64) {
#pragma unroll
              for (int c = 0; c < HF_CHUNK; c++) {
                // One key per lane, flat: the loop runs while any lane of the wave is still probing, every turn is the same
                // straight code -- read the slot, claim it if empty, count if it holds the key -- with the two atomics as its
                // only predicated regions (the nested ifs and breaks of the obvious form cost ~30 scalar instructions per
                // turn in exec-mask bookkeeping: the scalar unit, 68 % busy, was the kernel's busiest; profiles/r2j_sq_summary.json)
                const uint32_t key = kbuf[c];
                uint32_t pend = (key != 0xffffffffu && !(lp && (key >> ((uint32_t)R - lp)) != pass)) ? 1u : 0u;
                uint32_t h = (key * 0x9E3779B1u) >> (32u - lt);
                const uint32_t mine = HF_VALID | (key << 11);
                for (uint32_t tries = 0; __any(pend); tries++) {
                    const uint32_t e = tab[h]; // (lanes that are done read a slot they do not use)
                    uint32_t r = e;
                    if (pend & (uint32_t)(e == 0u))
                        r = atomicCAS(&tab[h], 0u, mine | 1u);
                    const uint32_t won = (uint32_t)(e == 0u) & (uint32_t)(r == 0u);
                    const uint32_t same = (uint32_t)((r & 0xfffff800u) == mine);
                    if (pend & same & (uint32_t)((r & 0x400u) == 0u)) // saturates far above 255 and far below the key bits
                        atomicAdd(&tab[h], 1u);
                    pend &= ~(won | same) & 1u;
                    if (tries >= HF_MAX_TRIES) { // (uniform) the table is as good as full: this pass needs a finer split
                        if (pend)
                            ctl[1] = 1;
                        pend = 0;
                    }
                    h = (h + 1u) & (T - 1u);
                }
              }
              } else {
              // (one wave per bucket: the form with breaks measured 7 % faster there -- 66.7 against 71.8 ms at configs[4]'s share)
#pragma unroll
              for (int c = 0; c < HF_CHUNK; c++) {
                const uint32_t key = kbuf[c];
                if (key == 0xffffffffu || (lp && (key >> ((uint32_t)R - lp)) != pass))
                    continue;
                uint32_t h = (key * 0x9E3779B1u) >> (32u - lt);
                const uint32_t mine = HF_VALID | (key << 11);
                for (uint32_t tries = 0;; tries++) {
                    uint32_t e = tab[h];
                    if (e == 0u) {
                        e = atomicCAS(&tab[h], 0u, mine | 1u);
                        if (e == 0u)
                            break;
                    }
                    if ((e & 0xfffff800u) == mine) {
                        if ((e & 0x7ffu) < 1024u) // saturates far above 255 and far below the key bits
                            atomicAdd(&tab[h], 1u);
                        break;
                    }
                    if (tries >= HF_MAX_TRIES) { // the table is (as good as) full of other keys: this pass needs a finer split
                        ctl[1] = 1;
                        break;
                    }
                    h = (h + 1u) & (T - 1u);
                }
              }
              }
            }
            team_sync();
            const uint32_t ovf = ctl[1];
            team_sync();
            if (ovf) { // redo this key range: with the whole table if it had less, else in 4 finer passes
                if (tl == 0)
                    ctl[1] = 0;
                team_sync();
                if (lt < log_t) {
                    lt = log_t;
                    T = Tmax;
                    continue;
                }
                if (lp + 2u <= (uint32_t)R) {
                    lp += 2u;
                    pass <<= 2;
                    continue;
                }
                // cannot split further: cannot happen (2^R <= T * 2^lp by then)
            }
            // list the solid entries: into the team's buffer (a pass holds at most 5/8 T / (abundance + 1) of them; what
            // does not fit goes straight to the global list), one flush per pass
            for (uint32_t q0 = tl * 4u; q0 < T; q0 += TEAM * 4u) { // four entries per lane and LDS read
                const uint4 e4 = *(const uint4 *)(tab + q0);
                const uint32_t ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t e = ev[j];
                    const uint32_t cnt = e & 0x7ffu;
                    if ((e & HF_VALID) && (cnt > 255u ? 255u : cnt) > abundance) {
                        const unsigned long long hk = (b << R) | (uint64_t)((e >> 11) & 0xfffffu);
                        const uint32_t pos = atomicAdd(&ctl[0], 1u);
                        if (pos < EB) {
                            ebuf[pos] = hk;
                        } else {
                            const unsigned long long gp = atomicAdd(emit_n, 1ull);
                            if (gp < emit_cap)
                                emit_keys[gp] = hk;
                        }
                    }
                }
            }
            team_sync();
            const uint32_t ec = ctl[0]; // read between two barriers: the same value in every thread
            team_sync();
            if (ec > EB / 2u) // one append to the global list per ~EB/2 hashes, not per bucket
                flush();
            pass++;
        }
    }
    flush();
}

// distinct keys per key on a sample of the buckets hash_final_kernel is about to count: block i counts bucket
// i * n_buckets / gridDim.x exactly (as many passes as its n asks for) -> dup[0] += distinct, dup[1] += n
__global__ __launch_bounds__(1024) void hf_sample_kernel(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ off,
                                                         uint64_t n_buckets, int R, uint32_t log_t, unsigned long long *__restrict__ dup)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t hf_lds[];
    __shared__ uint32_t sh_new;
    uint32_t *tab = (uint32_t *)hf_lds;
    const uint32_t T = 1u << log_t;
    const uint64_t b = (uint64_t)blockIdx.x * n_buckets / gridDim.x;
    const uint64_t s0 = off[b], n = off[b + 1] - s0;
    if (threadIdx.x == 0)
        sh_new = 0;
    uint32_t lp = 0;
    while (lp < (uint32_t)R && (n >> lp) > (uint64_t)(T / 8u * 5u))
        lp++;
    uint32_t fresh = 0;
    for (uint32_t pass = 0; pass < (1u << lp); pass++) {
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < T; q += 1024u)
            tab[q] = 0;
        __syncthreads();
        for (uint64_t i = threadIdx.x; i < n; i += 1024u) {
            const uint32_t key = keys[s0 + i];
            if (lp && (key >> ((uint32_t)R - lp)) != pass)
                continue;
            uint32_t h = (key * 0x9E3779B1u) >> (32u - log_t);
            const uint32_t mine = HF_VALID | (key << 11);
            for (uint32_t tries = 0; tries < T; tries++) {
                uint32_t e = tab[h];
                if (e == 0u) {
                    e = atomicCAS(&tab[h], 0u, mine);
                    if (e == 0u) {
                        fresh++;
                        break;
                    }
                }
                if (e == mine)
                    break;
                h = (h + 1u) & (T - 1u);
            }
        }
    }
    atomicAdd(&sh_new, fresh);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(dup, (unsigned long long)sh_new);
        atomicAdd(dup + 1, (unsigned long long)n);
    }
}

// ---- count spectrum without the u8 table (pcon::spectrum::Spectrum::from_count, src/main.rs:93) ----------------
// One wave per last-level bucket (2^F_BITS hashes, u16 keys): count the keys in LDS, then every key swaps its
// counter for 0 -- exactly one lane per distinct hash gets the count and bins min(count, 255), pcon's saturated
// u8.  Bin 0 (hashes never seen) is 2^(2k-1) minus the rest; the host fills it in.
constexpr int SP_WAVES = 4;
__global__ __launch_bounds__(64 * SP_WAVES) void final_spectrum_kernel(const uint16_t *__restrict__ keys, const uint64_t *__restrict__ off,
                                                                       uint64_t n_buckets, unsigned long long *__restrict__ hist)
{
    __shared__ uint32_t sp_cnt[SP_WAVES][F_SIZE];
    __shared__ unsigned long long sp_hist[256];
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    uint32_t *cnt = sp_cnt[wave];
    for (uint32_t q = lane; q < F_SIZE; q += 64)
        cnt[q] = 0;
    for (uint32_t q = threadIdx.x; q < 256; q += 64 * SP_WAVES)
        sp_hist[q] = 0;
    __syncthreads();
    const uint64_t n_waves = (uint64_t)gridDim.x * SP_WAVES;
    for (uint64_t g = (uint64_t)blockIdx.x * SP_WAVES + wave; g < n_buckets; g += n_waves) {
        const uint64_t s0 = off[g], n = off[g + 1] - s0;
        for (uint64_t i = lane; i < n; i += 64)
            atomicAdd(&cnt[keys[s0 + i]], 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // LDS operations of one wave complete in order
        for (uint64_t i = lane; i < n; i += 64) {
            const uint32_t c = atomicExch(&cnt[keys[s0 + i]], 0u);
            if (c)
                atomicAdd(&sp_hist[c > 255u ? 255u : c], 1ull);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < 256; q += 64 * SP_WAVES)
        if (sp_hist[q])
            atomicAdd(&hist[q], sp_hist[q]);
}

// concatenates, per level-1 bucket, the segments of every batch (only needed for > 1 batch)
__global__ __launch_bounds__(256) void merge_segments_kernel(const uint32_t *__restrict__ src,
                                                             const uint64_t *__restrict__ src_off,
                                                             const uint64_t *__restrict__ dst_off,
                                                             const uint64_t *__restrict__ dst_shift, uint32_t B1,
                                                             uint32_t *__restrict__ dst)
{
    for (uint32_t b = blockIdx.x; b < B1; b += gridDim.x) {
        const uint64_t lo = src_off[b], hi = src_off[b + 1];
        const uint64_t base = dst_off[b] + dst_shift[b];
        for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x)
            dst[base + (i - lo)] = src[i];
    }
}

__global__ void add_counts_kernel(const uint64_t *__restrict__ off, uint32_t B1, uint32_t *__restrict__ counts,
                                  uint64_t *__restrict__ shift_out)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B1) {
        if (shift_out)
            shift_out[b] = counts[b];
        counts[b] += (uint32_t)(off[b + 1] - off[b]);
    }
}

struct PartBatch {
    uint32_t *d_keys = nullptr;  // keys after level 1, grouped by digit 1
    uint64_t cap = 0, n = 0;
    uint64_t *d_l1off = nullptr; // B1 + 1
    bool borrowed = false;       // memory owned by the caller (segments received from other ranks)
};

} // namespace

namespace brx {

struct PartState {
    Plan pl;
    Plan pl_wide;     // the same first level, then the widest second digit (big inputs that end in hash_final_kernel)
    bool use_wide = false; // which of the two the levels >= 2 of the finish in progress follow
    std::vector<PartBatch> batches, spare;
    // generic workspace
    uint32_t *d_ntiles = nullptr;      // per parent
    uint64_t ntiles_cap = 0;
    uint64_t *d_item_off = nullptr;    // per parent + 1
    uint64_t item_off_cap = 0;
    uint32_t *d_item_parent = nullptr; // per work item
    uint64_t item_parent_cap = 0;
    uint32_t *d_matrix = nullptr;      // per-tile digit counts
    uint64_t matrix_cap = 0;
    uint64_t *d_pos = nullptr;         // exclusive scan of the matrix
    uint64_t pos_cap = 0;
    uint64_t *d_colpart = nullptr;     // level 1: per-chunk column sums / bases of the tile-major matrix
    uint64_t colpart_cap = 0;
    uint64_t *d_scan_tmp = nullptr;
    uint64_t scan_tmp_cap = 0;
    unsigned long long *d_scalars = nullptr; // [0] n_items, [1] total keys of the last scan, [2..3] hf_sample_kernel
    uint64_t *d_coff[MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr}; // child offsets (nchild + 1)
    uint32_t *d_keys_mid = nullptr;    // outputs of the middle levels (ping-pong when there are 4 levels)
    uint64_t keys_mid_cap = 0;
    uint32_t *d_keys_mid2 = nullptr;
    uint64_t keys_mid2_cap = 0;
    uint16_t *d_keys_fin = nullptr;    // last level output
    uint64_t keys_fin_cap = 0;
    uint32_t *d_merged = nullptr;      // all batches merged per level-1 bucket (only when > 1 batch)
    uint64_t merged_cap = 0;
    uint64_t *d_l1off_all = nullptr;
    uint32_t *d_cnts = nullptr;        // B1 running counts (merge)
    uint64_t *d_shift = nullptr;       // B1 (merge)
};

static int ensure_dev(void **p, uint64_t *cap, uint64_t need_bytes)
{
    if (*p && need_bytes <= *cap)
        return BRX_OK;
    if (*p)
        (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const uint64_t want = need_bytes + need_bytes / 16 + 256;
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {
        set_error("hipMalloc(%llu B partition workspace): %s", (unsigned long long)want, hipGetErrorString(e));
        return BRX_ERR_NOMEM;
    }
    *cap = want;
    return BRX_OK;
}

// presence-only insertion of a batch (see flat_insert_kernel); table == nullptr: into `bits`
int flat_presence_insert(const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n_reads, uint64_t total_bases, int k,
                         uint32_t *bits, uint64_t *table, uint32_t line_shift, uint32_t m, unsigned long long *d_new, hipStream_t s)
{
    if (!total_bases || !n_reads)
        return BRX_OK;
    const uint64_t n_items64 = (total_bases + L1_TILE - 1) / L1_TILE;
    if (n_items64 >= (1ull << 32)) {
        set_error("batch of %llu bases is too large for one insertion pass; split it", (unsigned long long)total_bases);
        return BRX_ERR_UNSUPPORTED;
    }
    L1Args a;
    memset(&a, 0, sizeof(a));
    a.bases = d_bases;
    a.offsets = d_offsets;
    a.n_reads = n_reads;
    a.total = total_bases;
    a.n_items = (uint32_t)n_items64;
    a.k = k;
    const int grid = a.n_items < 4096u ? (int)a.n_items : 4096;
    if (table)
        flat_insert_kernel<true><<<grid, 256, 0, s>>>(a, nullptr, table, line_shift, m, (uint32_t)k - m + 1u, d_new);
    else
        flat_insert_kernel<false><<<grid, 256, 0, s>>>(a, bits, nullptr, 0, 0, 0, nullptr);
    BRX_HIP(hipGetLastError());
    return BRX_OK;
}

// level-1 keys are u32 with a digit of at most MAX_DIGIT_BITS stripped: hashes of up to 41 bits, k <= 21
bool part_supported(int k)
{
    return 2 * k - 1 > F_BITS && 2 * k - 1 - F_BITS <= MAX_LEVELS * MAX_DIGIT_BITS && 2 * k - 1 - MAX_DIGIT_BITS <= 32;
}

int part_begin(brx_counter *c)
{
    PartState *st = new PartState();
    c->part = st;
    st->pl = make_plan(c->k);
    const Plan &pl = st->pl;
    st->pl_wide = pl;
    if (pl.nlev == 3 && pl.bits[1] < MAX_DIGIT_BITS) { // three levels: move bits from the last digit into the second
        Plan &w = st->pl_wide;
        const int moved = MAX_DIGIT_BITS - pl.bits[1] < pl.bits[2] ? MAX_DIGIT_BITS - pl.bits[1] : pl.bits[2];
        w.bits[1] += moved;
        w.bits[2] -= moved;
        w.rem_in[2] -= moved;
        w.nchild[1] <<= moved;
    } else if (pl.nlev == 4) {
        // four levels (k = 21: 9/7/7/6, hash count on 2^23 buckets of 18 bits): a 9-bit second digit and just enough third
        // digit to leave the 20 bits an LDS table entry can hold -- 9/9/3/8 -- ends in 2^21 buckets, four times as big:
        // from ~640 keys per bucket on they are taken by 1024-lane workgroups (3.4-5 ms per G keys) instead of one wave
        // each (6-11 ms per G keys, most of it per-bucket overhead on the scalar unit, profiles/r2j_sq_k21_summary.json)
        Plan &w = st->pl_wide;
        const int P = pl.bits[0] + pl.bits[1] + pl.bits[2] + pl.bits[3];
        const int b1 = MAX_DIGIT_BITS;
        int b2 = (pl.nbits - pl.bits[0] - b1) - 20;
        if (b2 < 0)
            b2 = 0;
        const int b3 = P - pl.bits[0] - b1 - b2;
        if (b3 >= 0 && b3 <= MAX_DIGIT_BITS && b2 <= MAX_DIGIT_BITS) {
            w.bits[1] = b1;
            w.bits[2] = b2;
            w.bits[3] = b3;
            w.rem_in[2] = w.rem_in[1] - b1;
            w.rem_in[3] = w.rem_in[2] - b2;
            w.nchild[1] = w.nchild[0] << b1;
            w.nchild[2] = w.nchild[1] << b2;
            w.nchild[3] = w.nchild[2] << b3;
        }
    }
    hipError_t e = hipMalloc((void **)&st->d_scalars, 32);
    for (int l = 0; l < pl.nlev && e == hipSuccess; l++) {
        const uint64_t nc = pl.nchild[l] > st->pl_wide.nchild[l] ? pl.nchild[l] : st->pl_wide.nchild[l];
        e = hipMalloc((void **)&st->d_coff[l], (nc + 1) * 8);
    }
    if (e == hipSuccess)
        e = hipMalloc((void **)&st->d_l1off_all, (pl.nchild[0] + 1) * 8);
    if (e == hipSuccess)
        e = hipMalloc((void **)&st->d_cnts, (pl.nchild[0] + 1) * 4);
    if (e == hipSuccess)
        e = hipMalloc((void **)&st->d_shift, (pl.nchild[0] + 1) * 8);
    if (e != hipSuccess) {
        set_error("partition state alloc: %s", hipGetErrorString(e));
        return BRX_ERR_NOMEM;
    }
    return BRX_OK;
}

void part_free(brx_counter *c)
{
    PartState *st = c->part;
    if (!st)
        return;
    for (auto *v : {&st->batches, &st->spare})
        for (auto &b : *v) {
            if (b.borrowed)
                continue;
            if (b.d_keys)
                (void)hipFree(b.d_keys);
            if (b.d_l1off)
                (void)hipFree(b.d_l1off);
        }
    for (void *p : {(void *)st->d_ntiles, (void *)st->d_item_off, (void *)st->d_item_parent, (void *)st->d_matrix,
                    (void *)st->d_pos, (void *)st->d_colpart, (void *)st->d_scan_tmp, (void *)st->d_scalars, (void *)st->d_coff[0],
                    (void *)st->d_coff[1], (void *)st->d_coff[2], (void *)st->d_coff[3], (void *)st->d_keys_mid,
                    (void *)st->d_keys_mid2, (void *)st->d_keys_fin,
                    (void *)st->d_merged, (void *)st->d_l1off_all, (void *)st->d_cnts, (void *)st->d_shift})
        if (p)
            (void)hipFree(p);
    delete st;
    c->part = nullptr;
}

int part_reset(brx_counter *c)
{
    PartState *st = c->part;
    for (auto &b : st->batches)
        if (!b.borrowed)
            st->spare.push_back(b);
    st->batches.clear();
    return BRX_OK;
}

// level-1 output of the (single) local batch: keys grouped by their first digit + bucket offsets
int part_l1_view(brx_counter *c, void **d_keys, void **d_l1off, uint32_t *n_buckets, uint64_t *n_keys)
{
    PartState *st = c->part;
    *n_buckets = (uint32_t)st->pl.nchild[0];
    if (st->batches.empty()) { // nothing counted (an empty shard): no keys, and the caller takes every offset as 0
        *d_keys = nullptr;
        *d_l1off = nullptr;
        *n_keys = 0;
        return BRX_OK;
    }
    if (st->batches.size() != 1 || st->batches[0].borrowed) {
        set_error("l1_view needs exactly one locally counted batch (have %zu)", st->batches.size());
        return BRX_ERR_ARG;
    }
    *d_keys = st->batches[0].d_keys;
    *d_l1off = st->batches[0].d_l1off;
    *n_keys = st->batches[0].n;
    return BRX_OK;
}

// registers a level-1 segment produced elsewhere (another rank); the memory stays the caller's and
// must outlive finish
int part_add_partitioned(brx_counter *c, const uint32_t *d_keys, const uint64_t *d_l1off, uint64_t n_keys)
{
    PartState *st = c->part;
    if (n_keys == 0)
        return BRX_OK;
    PartBatch b;
    b.d_keys = const_cast<uint32_t *>(d_keys);
    b.d_l1off = const_cast<uint64_t *>(d_l1off);
    b.n = n_keys;
    b.cap = n_keys;
    b.borrowed = true;
    st->batches.push_back(b);
    return BRX_OK;
}

// matrix / pos / scan scratch for `n_entries` per-tile counters
static int ensure_matrix(PartState *st, uint64_t n_entries)
{
    uint64_t capb = st->matrix_cap;
    BRX_TRY(ensure_dev((void **)&st->d_matrix, &capb, (n_entries + 1) * 4));
    st->matrix_cap = capb;
    capb = st->pos_cap;
    BRX_TRY(ensure_dev((void **)&st->d_pos, &capb, (n_entries + 2) * 8));
    st->pos_cap = capb;
    capb = st->scan_tmp_cap;
    BRX_TRY(ensure_dev((void **)&st->d_scan_tmp, &capb, scan_tmp_bytes((uint32_t)n_entries) + 64));
    st->scan_tmp_cap = capb;
    return BRX_OK;
}

static size_t scatter_lds_bytes(uint32_t tile, int bits)
{
    const size_t B = (size_t)1 << bits;
    return (size_t)tile * 6 + B * 8 + B * 12 + 64;
}

static size_t l1_scatter_lds_bytes(int bits)
{
    const size_t B = (size_t)1 << bits;
    return (size_t)L1_TILE * 4 + B * 8 + B * 12 + 64; // keys, sector index + carry state, three counters
}

int part_add_batch(brx_counter *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n_reads,
                   uint64_t total_bases, hipStream_t s)
{
    PartState *st = c->part;
    const Plan &pl = st->pl;
    if (total_bases == 0 || n_reads == 0)
        return BRX_OK;
    const uint64_t max_keys = total_bases; // upper bound of the number of k-mers of this batch
    PartBatch b;
    for (size_t i = 0; i < st->spare.size(); i++)
        if (st->spare[i].cap >= max_keys) {
            b = st->spare[i];
            st->spare.erase(st->spare.begin() + (long)i);
            break;
        }
    if (!b.d_keys) {
        uint64_t capb = 0;
        BRX_TRY(ensure_dev((void **)&b.d_keys, &capb, (max_keys + 1) * 4));
        b.cap = capb / 4;
        BRX_HIP(hipMalloc((void **)&b.d_l1off, (pl.nchild[0] + 1) * 8));
    }
    const uint32_t B = 1u << pl.bits[0];
    const uint64_t n_items64 = (total_bases + L1_TILE - 1) / L1_TILE;
    if (n_items64 * B >= (1ull << 32)) {
        set_error("batch of %llu bases is too large for one partition pass; split it", (unsigned long long)total_bases);
        return BRX_ERR_UNSUPPORTED;
    }
    const uint32_t n_items = (uint32_t)n_items64;
    const uint64_t n_entries = (uint64_t)B * n_items;
    BRX_TRY(ensure_matrix(st, n_entries));
    L1Args a;
    memset(&a, 0, sizeof(a));
    a.bases = d_bases;
    a.offsets = d_offsets;
    a.n_reads = n_reads;
    a.total = total_bases;
    a.n_items = n_items;
    a.k = pl.k;
    a.nbits = pl.nbits;
    a.bits = pl.bits[0];
    a.matrix = st->d_matrix;
    a.pos = st->d_pos;
    a.keys_out = b.d_keys;
    a.out_cap = b.cap;
    // both level-1 kernels give a block a contiguous range of tiles (BRX_L1_GRID: tests shrink the grid so that small
    // inputs give every block several tiles)
    const char *genv = getenv("BRX_L1_GRID");
    const long gev = genv ? atol(genv) : 0;
    const uint32_t hist_blocks = gev > 0 ? (uint32_t)gev : 2048u, scatter_blocks = gev > 0 ? (uint32_t)gev : 2560u;
    const int grid = (int)(n_items < hist_blocks ? n_items : hist_blocks);
    {
        KernelTimer t("part_l1_hist", s);
        l1_hist_kernel<<<grid, 256, (size_t)B * 4, s>>>(a);
    }
    trace_stage(s, "partition level 1: histograms");
    {
        // positions of every (tile, digit) run and the digit offsets, from the tile-major counts (col_scan_* above)
        KernelTimer t("offsets_scan", s);
        uint32_t CH = 64;
        while ((n_items + CH - 1) / CH > 2048u)
            CH *= 2;
        const uint32_t n_chunks = (n_items + CH - 1) / CH;
        uint64_t capb = st->colpart_cap;
        BRX_TRY(ensure_dev((void **)&st->d_colpart, &capb, ((uint64_t)n_chunks + 1) * B * 8));
        st->colpart_cap = capb;
        uint64_t *coltot = st->d_colpart + (uint64_t)n_chunks * B;
        col_scan_partial_kernel<<<n_chunks, 256, 0, s>>>(st->d_matrix, n_items, B, CH, st->d_colpart);
        col_scan_columns_kernel<<<B, 256, 0, s>>>(st->d_colpart, n_chunks, B, coltot);
        col_scan_digits_kernel<<<1, 64, 0, s>>>(coltot, B, b.d_l1off, st->d_scalars + 1);
        col_scan_positions_kernel<<<n_chunks, 256, 0, s>>>(st->d_matrix, n_items, B, CH, st->d_colpart, b.d_l1off, st->d_pos);
    }
    {
        KernelTimer t("part_l1_scatter", s);
        const size_t lds1 = l1_scatter_lds_bytes(a.bits);
        if (lds1 > 64 * 1024)
            BRX_HIP(hipFuncSetAttribute((const void *)l1_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        // contiguous tile ranges, one per block (the kernel's carries): twice the blocks the chip holds at once
        // (5 per CU at 27 KB of LDS): the ranges stay long (~100 tiles at 1 Gbp) and the last round of blocks short
        const uint32_t sgrid = n_items < scatter_blocks ? n_items : scatter_blocks;
        l1_scatter_kernel<<<sgrid, 256, lds1, s>>>(a);
    }
    trace_stage(s, "partition level 1: scatter");
    BRX_HIP(hipGetLastError());
    unsigned long long tot = 0;
    BRX_HIP(hipMemcpyAsync(&tot, st->d_scalars + 1, 8, hipMemcpyDeviceToHost, s));
    BRX_HIP(hipStreamSynchronize(s));
    b.n = tot;
    st->batches.push_back(b);
    return BRX_OK;
}

// one level >= 2: work items -> per-tile histograms -> scan -> scatter -> child offsets
template <typename OUT>
static int run_level(PartState *st, int l, const uint32_t *keys_in, const uint64_t *poff, uint64_t n_parents, uint64_t total,
                     void *keys_out, hipStream_t s, const char *tag_hist, const char *tag_scatter)
{
    const Plan &pl = st->use_wide ? st->pl_wide : st->pl;
    const uint32_t B = 1u << pl.bits[l];
    // 8192 keys per tile whatever the digit width: the (tile, digit) runs a scatter writes are then 32 keys = 128 bytes
    // on average instead of 64, and runs that straddle fewer lines cost fewer read-for-ownership fetches (level 2 of
    // the bench: 4.5 -> 2.7 ms; BRX_SMALL_TILES=1 restores 4096 for 8-bit digits)
    static const bool small_tiles = [] { const char *e = getenv("BRX_SMALL_TILES"); return e && *e == '1'; }();
    const uint32_t tile = (pl.bits[l] > 8 || !small_tiles) ? 8192u : 4096u;
    const uint64_t ub_items = total / tile + n_parents + 1;
    const uint64_t n_entries = (uint64_t)B * ub_items;
    if (n_entries >= (1ull << 32)) {
        set_error("partition level %d needs %llu tile counters; split the input into smaller counters", l + 1,
                  (unsigned long long)n_entries);
        return BRX_ERR_UNSUPPORTED;
    }
    {
        uint64_t capb = st->ntiles_cap;
        BRX_TRY(ensure_dev((void **)&st->d_ntiles, &capb, (n_parents + 1) * 4));
        st->ntiles_cap = capb;
        capb = st->item_off_cap;
        BRX_TRY(ensure_dev((void **)&st->d_item_off, &capb, (n_parents + 2) * 8));
        st->item_off_cap = capb;
        capb = st->item_parent_cap;
        BRX_TRY(ensure_dev((void **)&st->d_item_parent, &capb, (ub_items + 1) * 4));
        st->item_parent_cap = capb;
    }
    BRX_TRY(ensure_matrix(st, n_entries > n_parents ? n_entries : n_parents));
    tiles_from_offsets_kernel<<<(unsigned)((n_parents + 255) / 256), 256, 0, s>>>(poff, n_parents, tile, st->d_ntiles);
    BRX_TRY(exclusive_scan_lens(st->d_ntiles, (uint32_t)n_parents, st->d_scan_tmp, st->d_item_off, st->d_scalars, s));
    fill_item_parent_kernel<<<(unsigned)((n_parents + 255) / 256), 256, 0, s>>>(st->d_item_off, n_parents, st->d_item_parent);
    trace_stage(s, tag_hist);
    BRX_HIP(hipMemsetAsync(st->d_matrix, 0, n_entries * 4, s));
    LnArgs a;
    memset(&a, 0, sizeof(a));
    a.keys_in = keys_in;
    a.in_cap = total;
    a.poff = poff;
    a.n_parents = n_parents;
    a.item_off = st->d_item_off;
    a.item_parent = st->d_item_parent;
    a.n_items = st->d_scalars;
    a.tile = tile;
    a.rem_in = pl.rem_in[l];
    a.bits = pl.bits[l];
    a.matrix = st->d_matrix;
    a.pos = st->d_pos;
    a.keys_out = keys_out;
    a.out_cap = total + 64;
    const int grid = 256 * 8;
    {
        KernelTimer t(tag_hist, s);
        if (tile == 8192u)
            ln_hist_kernel<32><<<grid, 256, (size_t)B * 4, s>>>(a);
        else
            ln_hist_kernel<16><<<grid, 256, (size_t)B * 4, s>>>(a);
    }
    trace_stage(s, "  histograms done");
    {
        KernelTimer t("offsets_scan", s);
        const unsigned g = n_parents < 4096 ? (unsigned)n_parents : 4096u;
        ln_colscan_kernel<<<g, 256, 0, s>>>(st->d_matrix, poff, st->d_item_off, n_parents, B, st->d_pos, st->d_coff[l]);
    }
    {
        KernelTimer t(tag_scatter, s);
        const size_t lds = scatter_lds_bytes(tile, a.bits);
        if (tile == 8192u)
            ln_scatter_kernel<32, OUT><<<grid, 256, lds, s>>>(a);
        else
            ln_scatter_kernel<16, OUT><<<grid, 256, lds, s>>>(a);
    }
    trace_stage(s, tag_scatter);
    BRX_HIP(hipGetLastError());
    return BRX_OK;
}

// d_hist != nullptr: no set is produced (dst is not touched); the keys go through every level and the final buckets
// are binned into the 256-entry count spectrum (bins 1..255)
static int part_finish_impl(brx_counter *c, uint32_t abundance, hipStream_t s, brx_set *dst, unsigned long long *d_hist)
{
    PartState *st = c->part;
    uint64_t total = 0;
    for (auto &b : st->batches)
        total += b.n;
    // Buckets that end in the LDS hash table (no bit vector to write) and would average more keys than one table pass
    // takes even at high coverage: cut them finer at level 2 instead (k = 19: digits 9, 9 in place of 9, 8 -- 2^18
    // buckets with 19 bits left; configs[3]'s share per GPU then has 24 000 keys per bucket, ~5 000 of them distinct)
    // (BRX_WIDE_L2 = the average bucket size from which on the wider digit is used, 0 = never, 1 = always; read per call:
    // the fuzzers force it on small inputs)
    const char *e_wide = getenv("BRX_WIDE_L2");
    const uint64_t wide_from = e_wide && *e_wide ? strtoull(e_wide, nullptr, 10) : 16384ull;
    const bool lazy_on = [] { const char *e = getenv("BRX_LAZY_BITS"); return !(e && *e == '0'); }(); // (per call too)
    st->use_wide = wide_from != 0 && !d_hist && dst && (dst->sparse || (index_wanted(c->k) && lazy_on)) &&
                   st->pl_wide.bits[1] != st->pl.bits[1] &&
                   (wide_from == 1 || (st->pl.nlev == 3 ? total / st->pl.nchild[1] > wide_from
                                                        // four levels: as soon as the wider buckets fill a workgroup
                                                        : total / st->pl_wide.nchild[2] > 640));
    const Plan &pl = st->use_wide ? st->pl_wide : st->pl;
    const uint32_t B1 = (uint32_t)pl.nchild[0];

    const uint32_t *keys1 = nullptr;
    const uint64_t *l1off = nullptr;
    if (st->batches.empty() || total == 0) {
        if (d_hist)
            return BRX_OK; // nothing counted: bins 1..255 stay 0
        if (dst->sparse) {
            if (!dst->d_keylist_n)
                BRX_HIP(hipMalloc((void **)&dst->d_keylist_n, 8));
            BRX_HIP(hipMemsetAsync(dst->d_keylist_n, 0, 8, s));
            dst->keylist_valid = true; // the empty list
        } else {
            BRX_HIP(hipMemsetAsync(dst->d_bits, 0, dst->nwords * 4, s));
            dst->bits_stale = false;
        }
        return BRX_OK;
    } else if (st->batches.size() == 1) {
        keys1 = st->batches[0].d_keys;
        l1off = st->batches[0].d_l1off;
    } else {
        uint64_t capb = st->merged_cap;
        BRX_TRY(ensure_dev((void **)&st->d_merged, &capb, (total + 1) * 4));
        st->merged_cap = capb;
        BRX_TRY(ensure_matrix(st, B1 + 1));
        BRX_HIP(hipMemsetAsync(st->d_cnts, 0, (uint64_t)B1 * 4, s));
        for (auto &b : st->batches)
            add_counts_kernel<<<(B1 + 255) / 256, 256, 0, s>>>(b.d_l1off, B1, st->d_cnts, nullptr);
        BRX_TRY(exclusive_scan_lens(st->d_cnts, B1, st->d_scan_tmp, st->d_l1off_all, st->d_scalars + 1, s));
        BRX_HIP(hipMemsetAsync(st->d_cnts, 0, (uint64_t)B1 * 4, s));
        for (auto &b : st->batches) {
            add_counts_kernel<<<(B1 + 255) / 256, 256, 0, s>>>(b.d_l1off, B1, st->d_cnts, st->d_shift);
            merge_segments_kernel<<<B1 < 4096u ? B1 : 4096u, 256, 0, s>>>(b.d_keys, b.d_l1off, st->d_l1off_all, st->d_shift, B1,
                                                                         st->d_merged);
        }
        keys1 = st->d_merged;
        l1off = st->d_l1off_all;
    }

    {
        uint64_t capb = st->keys_fin_cap;
        BRX_TRY(ensure_dev((void **)&st->d_keys_fin, &capb, (total + 64) * 2));
        st->keys_fin_cap = capb;
    }
    // solid-key list for the probe index: a solid hash was seen more than `abundance` times, so there are at
    // most total / (abundance + 1) of them; real data is far below that, and a list that turns out too short
    // is simply not used (the index is then built from the bit vector)
    const bool emit = !d_hist && (dst->sparse || index_wanted(c->k));
    // lazy bit vector: when the solid hashes are listed anyway, the 2^(2k-4)-byte vector (16 GiB of slices at k = 19)
    // is written only if somebody asks for it later (ensure_bits); BRX_LAZY_BITS=0 writes it here as before
    const bool lazy = emit && !dst->sparse && lazy_on;
    if (emit) {
        // (a set without bits has nothing but this list: it gets the exact bound)
        const uint64_t div = (dst->sparse || lazy) ? abundance + 1u : (abundance + 1u > 8u ? abundance + 1u : 8u);
        const uint64_t want = total / div + (1ull << 20);
        if (dst->keylist_cap < want || !dst->d_keylist) {
            if (dst->d_keylist)
                (void)hipFree(dst->d_keylist);
            dst->d_keylist = nullptr;
            dst->keylist_cap = 0;
            BRX_HIP(hipMalloc((void **)&dst->d_keylist, want * 8));
            dst->keylist_cap = want;
        }
        if (!dst->d_keylist_n)
            BRX_HIP(hipMalloc((void **)&dst->d_keylist_n, 8));
        BRX_HIP(hipMemsetAsync(dst->d_keylist_n, 0, 8, s));
    }
    // levels 2 .. nlev: u32 keys through one or two middle buffers, u16 out of the last level
    static const char *hist_tag[MAX_LEVELS] = {"part_l1_hist", "part_l2_hist", "part_l3_hist", "part_l4_hist"};
    static const char *scat_tag[MAX_LEVELS] = {"part_l1_scatter", "part_l2_scatter", "part_l3_scatter", "part_l4_scatter"};
    if (pl.nlev >= 3) {
        uint64_t capb = st->keys_mid_cap;
        BRX_TRY(ensure_dev((void **)&st->d_keys_mid, &capb, (total + 64) * 4));
        st->keys_mid_cap = capb;
    }
    if (pl.nlev >= 4) {
        uint64_t capb = st->keys_mid2_cap;
        BRX_TRY(ensure_dev((void **)&st->d_keys_mid2, &capb, (total + 64) * 4));
        st->keys_mid2_cap = capb;
    }
    const uint32_t *kin = keys1;
    const uint64_t *poff = l1off;
    // no slices to write (lazy / sparse): stop one level early and hash-count the coarser buckets, if they are small
    // enough for an LDS table (k = 19 at 1 Gbp: 131 072 buckets of ~7 600 keys with 20 bits left)
    const bool hf_on = [] { const char *e = getenv("BRX_HASH_FINAL"); return !(e && *e == '0'); }(); // (per call: fuzzers)
    const int R_hf = pl.rem_in[pl.nlev - 1];
    const uint64_t nb_hf = pl.nlev >= 2 ? pl.nchild[pl.nlev - 2] : 0;
    // (up to 2^17 keys per bucket: the table then takes a bucket in a few passes over its key range, see hash_final_kernel)
    static const uint64_t hf_max_avg = [] { const char *e = getenv("BRX_HASH_FINAL_MAX"); return e && *e ? strtoull(e, nullptr, 10) : 131072ull; }();
    const bool hash_final = !d_hist && hf_on && (dst->sparse || lazy) && pl.nlev >= 3 && R_hf <= 20 && total / nb_hf <= hf_max_avg;
    for (int l = 1; l < pl.nlev; l++) {
        if (hash_final && l == pl.nlev - 1)
            break;
        if (l == pl.nlev - 1) {
            BRX_TRY(run_level<uint16_t>(st, l, kin, poff, pl.nchild[l - 1], total, st->d_keys_fin, s, hist_tag[l], scat_tag[l]));
        } else {
            uint32_t *kout = (l & 1) ? st->d_keys_mid : st->d_keys_mid2;
            BRX_TRY(run_level<uint32_t>(st, l, kin, poff, pl.nchild[l - 1], total, kout, s, hist_tag[l], scat_tag[l]));
            kin = kout;
        }
        poff = st->d_coff[l];
    }
    if (hash_final) {
        KernelTimer t("part_hash_final", s);
        const uint64_t avg = total / nb_hf;
        // test hooks, read per call: BRX_HF_LOG_T = the launch's table maximum (log2), BRX_HF_MIN_LT = the smallest table a
        // bucket gets, BRX_HF_RATIO = the share of distinct keys to assume instead of the sampled one.  Small or wrongly
        // guessed tables walk the whole ladder (bigger table -> 4x finer passes) on inputs a test can afford.
        const char *e_logt = getenv("BRX_HF_LOG_T"), *e_minlt = getenv("BRX_HF_MIN_LT"), *e_ratio = getenv("BRX_HF_RATIO");
        const float ratio_forced = e_ratio && *e_ratio ? (float)atof(e_ratio) : 0.0f;
        // how many of a bucket's keys are distinct, measured on 64 buckets: sizes each bucket's table and passes
        unsigned long long *dup = st->d_scalars + 2;
        {
            BRX_HIP(hipMemsetAsync(dup, 0, 16, s));
            BRX_HIP(hipFuncSetAttribute((const void *)hf_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
            const int ns = (int)(nb_hf < 64ull ? nb_hf : 64ull);
            hf_sample_kernel<<<ns, 1024, 65536, s>>>(kin, poff, nb_hf, R_hf, 14, dup);
        }
        if (avg > 640) { // one 1024-thread workgroup per bucket, 16384-entry table
            uint32_t log_t = 14;
            if (e_logt && *e_logt && atoi(e_logt) >= 6 && atoi(e_logt) <= 14)
                log_t = (uint32_t)atoi(e_logt);
            const uint32_t min_lt = e_minlt && *e_minlt && atoi(e_minlt) >= 4 ? (uint32_t)atoi(e_minlt) : 12u;
            const size_t lds = ((size_t)4 << log_t) + hf_ebuf(1024) * 8 + 16;
            BRX_HIP(hipFuncSetAttribute((const void *)hash_final_kernel<1024, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = (int)(nb_hf < 256ull * 4ull ? nb_hf : 256ull * 4ull);
            hash_final_kernel<1024, 1024><<<grid, 1024, lds, s>>>(kin, poff, nb_hf, R_hf, abundance, log_t, dst->d_keylist, dst->keylist_cap,
                                                          dst->d_keylist_n, dup, min_lt, ratio_forced);
        } else { // one wave per bucket; at most 256..2048 table entries, about 4x the average bucket (each bucket then uses
                 // 2.5 slots per distinct key expected: clearing and scanning cost T / 256 LDS accesses per lane).  Sizing
                 // the launch's maximum from the sampled ratio as well (1024 slots = 6 workgroups per CU instead of 3)
                 // measured the same 66.9 ms at configs[4]'s share: the kernel is not short of waves.
            uint32_t log_t = 8;
            while (log_t < 11 && (1ull << log_t) < 4 * avg)
                log_t++;
            if (e_logt && *e_logt && atoi(e_logt) >= 6 && atoi(e_logt) <= 11)
                log_t = (uint32_t)atoi(e_logt);
            const uint32_t min_lt = e_minlt && *e_minlt && atoi(e_minlt) >= 4 ? (uint32_t)atoi(e_minlt) : 8u;
            const size_t lds = 4 * (((size_t)4 << log_t) + hf_ebuf(64) * 8 + 16);
            const uint64_t want = (nb_hf + 3) / 4;
            const int grid = (int)(want < 256ull * 16ull ? want : 256ull * 16ull);
            hash_final_kernel<64, 256><<<grid, 256, lds, s>>>(kin, poff, nb_hf, R_hf, abundance, log_t, dst->d_keylist, dst->keylist_cap,
                                                         dst->d_keylist_n, dup, min_lt, ratio_forced);
        }
        BRX_HIP(hipGetLastError());
        trace_stage(s, "hash final");
        dst->keylist_valid = true;
        dst->bits_stale = lazy;
        return BRX_OK;
    }
    const uint64_t *fin_off = st->d_coff[pl.nlev - 1];
    if (d_hist) {
        KernelTimer t("part_spectrum", s);
        const uint64_t nb = pl.nchild[pl.nlev - 1];
        const uint64_t want = (nb + SP_WAVES - 1) / SP_WAVES;
        final_spectrum_kernel<<<(int)(want < 256ull * 8ull ? want : 256ull * 8ull), 64 * SP_WAVES, 0, s>>>(st->d_keys_fin, fin_off, nb, d_hist);
        BRX_HIP(hipGetLastError());
        trace_stage(s, "spectrum");
        return BRX_OK;
    }
    {
        KernelTimer t("part_final_count", s);
        const uint64_t nb = pl.nchild[pl.nlev - 1];
        const uint64_t want_waves = nb < (256ull * 16ull * 8ull) ? nb : (256ull * 16ull * 8ull);
        const int grid = (int)((want_waves + P3_WAVES - 1) / P3_WAVES);
        if (dst->sparse || lazy)
            final_count_kernel<true, false><<<grid, 64 * P3_WAVES, 0, s>>>(st->d_keys_fin, fin_off, nb, abundance, nullptr,
                                                                          dst->d_keylist, dst->keylist_cap, dst->d_keylist_n);
        else if (emit)
            final_count_kernel<true, true><<<grid, 64 * P3_WAVES, 0, s>>>(st->d_keys_fin, fin_off, nb, abundance, dst->d_bits,
                                                                         dst->d_keylist, dst->keylist_cap, dst->d_keylist_n);
        else
            final_count_kernel<false, true><<<grid, 64 * P3_WAVES, 0, s>>>(st->d_keys_fin, fin_off, nb, abundance, dst->d_bits,
                                                                          nullptr, 0, nullptr);
    }
    BRX_HIP(hipGetLastError());
    trace_stage(s, "final count");
    dst->keylist_valid = emit;
    dst->bits_stale = lazy;
    return BRX_OK;
}

int part_finish_into(brx_counter *c, uint32_t abundance, hipStream_t s, brx_set *dst)
{
    const int rc = part_finish_impl(c, abundance, s, dst, nullptr);
#if BRX_DEBUG_BOUNDS
    unsigned long long bad = 0;
    BRX_HIP(hipStreamSynchronize(s));
    BRX_HIP(hipMemcpyFromSymbol(&bad, HIP_SYMBOL(bounds_violations), 8));
    if (bad) {
        set_error("BRX_DEBUG_BOUNDS: %llu scatter stores / key loads of the partition passes were out of range and dropped", bad);
        return BRX_ERR_HIP;
    }
#endif
    return rc;
}

// bins 1..255 of the count spectrum into d_hist (256 x u64, zeroed by the caller); the counter is left as it was
int part_spectrum(brx_counter *c, hipStream_t s, unsigned long long *d_hist)
{
    return part_finish_impl(c, 255u, s, nullptr, d_hist);
}

} // namespace brx
