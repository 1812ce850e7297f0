// Solid-k-mer set: packed canonical bitset in HBM, built by counting.
//
// Reference path: src/main.rs:72-115 (Counter::new -> count_fasta -> Solid::from_count),
// src/set/pcon.rs:13-196 (Pcon wrapper, get/k), pcon::solid::Solid (bit layout).
//
// HBM layout
//   bitset : u32[2^(2k-1)/32]; bit i (= canonical>>1) is bit (i & 31) of word (i >> 5)
//            == bit (i & 7) of byte (i >> 3) of the .solid stream (Lsb0, little endian).
//   counts : u8[2^(2k-1)] packed 4 per u32 word (dense strategy), saturating at 255.
//
// Kernels (all integer, HBM-bound; no MFMA):
//   count_dense_kernel  one workgroup per read; each lane rolls a strip of consecutive
//                       k-mers and does one saturating byte increment (32-bit CAS) per k-mer.
//                       Algorithmic bytes per k-mer: 1 B base + 64 B counter line read + 64 B
//                       write-back = 129 B.
//   threshold_kernel    streams the count table once: 32 counts -> 1 bitset word.
//                       2^(2k-1) B read + 2^(2k-4) B written.
//   insert_kernel       presence-only build (Pcon::from_fasta): atomicOr of one bit per k-mer.
//   get_kernel          n independent KmerSet::get probes.
#include "brx_internal.hpp"

#include <stdlib.h>
#include <string.h>

using namespace brx;

namespace {

constexpr int STRIP = 16; // consecutive k-mers rolled by one lane

__device__ __forceinline__ void count_inc_u8(uint32_t *counts, uint64_t h)
{
    uint32_t *w = counts + (h >> 2);
    const unsigned sh = (unsigned)(h & 3u) * 8u;
    uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if (((old >> sh) & 0xffu) == 0xffu)
            return; // saturated (pcon Counter<u8>; saturate-vs-wrap unpinned)
        const uint32_t want = old + (1u << sh);
        const uint32_t seen = atomicCAS(w, old, want);
        if (seen == old)
            return;
        old = seen;
    }
}

// MODE 0: dense count; MODE 1: presence insert (atomicOr into the bitset)
template <int MODE>
__global__ __launch_bounds__(256) void kmer_scatter_kernel(const uint8_t *__restrict__ bases,
                                                            const uint64_t *__restrict__ offsets, uint32_t n_reads,
                                                            int k, uint32_t *__restrict__ table)
{
    const uint64_t mask = kmask(k);
    for (uint32_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const uint64_t s = offsets[r];
        const uint64_t len = offsets[r + 1] - s;
        if (len < (uint64_t)k)
            continue;
        const uint64_t nk = len - (uint64_t)k + 1;
        const uint8_t *seq = bases + s;
        for (uint64_t p0 = (uint64_t)threadIdx.x * STRIP; p0 < nk; p0 += (uint64_t)blockDim.x * STRIP) {
            uint64_t kmer = 0;
            for (int j = 0; j < k; j++)
                kmer = (kmer << 2) | nuc2bit(seq[p0 + j]);
            const uint64_t pend = (p0 + STRIP < nk) ? p0 + STRIP : nk;
            for (uint64_t p = p0;;) {
                const uint64_t h = khash(kmer, k);
                if (MODE == 0)
                    count_inc_u8(table, h);
                else
                    atomicOr(table + (h >> 5), 1u << (h & 31u));
                if (++p >= pend)
                    break;
                kmer = add_nuc(kmer, nuc2bit(seq[p + k - 1]), mask);
            }
        }
    }
}

// one lane: 32 consecutive counts (two 16-byte loads) -> one bitset word
__global__ __launch_bounds__(256) void threshold_kernel(const uint4 *__restrict__ counts, uint64_t nwords,
                                                        uint32_t abundance, uint32_t *__restrict__ bits)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += stride) {
        const uint4 a = counts[2 * w], b = counts[2 * w + 1];
        const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t out = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                out |= (uint32_t)(((v[q] >> (8 * j)) & 0xffu) > abundance) << (4 * q + j);
        }
        bits[w] = out;
    }
}

__global__ __launch_bounds__(256) void clamp_kernel(uint4 *__restrict__ counts, uint64_t nvec, uint32_t cap)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        uint4 a = counts[i];
        uint32_t v[4] = {a.x, a.y, a.z, a.w};
        bool dirty = false;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t o = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t c = (v[q] >> (8 * j)) & 0xffu;
                c = c > cap ? cap : c;
                o |= c << (8 * j);
            }
            dirty |= (o != v[q]);
            v[q] = o;
        }
        if (dirty)
            counts[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
}

// 256-bin histogram of the u8 counts (pcon::spectrum::Spectrum::from_count, src/main.rs:93)
__global__ __launch_bounds__(256) void spectrum_kernel(const uint32_t *__restrict__ counts, uint64_t nwords,
                                                       unsigned long long *__restrict__ hist)
{
    __shared__ unsigned int sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) {
        const uint32_t w = counts[i];
        atomicAdd(&sh[w & 0xffu], 1u);
        atomicAdd(&sh[(w >> 8) & 0xffu], 1u);
        atomicAdd(&sh[(w >> 16) & 0xffu], 1u);
        atomicAdd(&sh[w >> 24], 1u);
    }
    __syncthreads();
    if (sh[threadIdx.x])
        atomicAdd(&hist[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}

__global__ void get_kernel(const uint32_t *__restrict__ bits, const uint64_t *__restrict__ kmers, uint32_t n, int k,
                           uint8_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint64_t h = khash(kmers[i], k);
    out[i] = (uint8_t)((bits[h >> 5] >> (h & 31u)) & 1u);
}

__global__ void set_bit_kernel(uint32_t *bits, uint64_t h, int value)
{
    if (value)
        atomicOr(bits + (h >> 5), 1u << (h & 31u));
    else
        atomicAnd(bits + (h >> 5), ~(1u << (h & 31u)));
}

__global__ __launch_bounds__(256) void popcount_kernel(const uint32_t *__restrict__ bits, uint64_t nwords,
                                                       unsigned long long *__restrict__ total)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += stride)
        acc += (unsigned long long)__popc(bits[w]);
    for (int d = 32; d > 0; d >>= 1)
        acc += __shfl_down(acc, d);
    if ((threadIdx.x & 63) == 0 && acc)
        atomicAdd(total, acc);
}

// compacts the set bits of words [w0, w0+nw) into a list of absolute bit indices (any order).
// The set is sparse (k=19: 2e7 bits in 4e9 words): a wave streams 64 x 16 bytes per iteration and only
// enters the compaction when its ballot says some lane holds a non-zero word.  Found indices are
// collected in an LDS buffer per workgroup and flushed with ONE global atomic per ~2000 entries
// (a single global counter hit once per wave serialises at ~3.5 ns per atomic).
constexpr uint32_t EXTRACT_BUF = 2048;

__device__ __forceinline__ void extract_flush(uint64_t *lbuf, uint32_t *lcount, unsigned long long *counter, uint64_t *out,
                                              uint64_t cap, unsigned long long *sh_base)
{
    __syncthreads();
    const uint32_t n = *lcount < EXTRACT_BUF + 1024u ? *lcount : EXTRACT_BUF + 1024u;
    if (threadIdx.x == 0)
        *sh_base = n ? atomicAdd(counter, (unsigned long long)n) : 0ull;
    __syncthreads();
    const unsigned long long base = *sh_base;
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x)
        if (base + j < cap)
            out[base + j] = lbuf[j];
    __syncthreads();
    if (threadIdx.x == 0)
        *lcount = 0;
    __syncthreads();
}

__global__ __launch_bounds__(256) void extract_kernel(const uint32_t *__restrict__ bits, uint64_t w0, uint64_t nw,
                                                      unsigned long long *__restrict__ counter, uint64_t *__restrict__ out,
                                                      uint64_t cap)
{
    // one iteration of the block adds at most 256 lanes x 128 bits... in practice a few; the buffer has
    // 1024 entries of head-room above the flush threshold and a dense iteration flushes first
    __shared__ uint64_t lbuf[EXTRACT_BUF + 1024];
    __shared__ uint32_t lcount;
    __shared__ unsigned long long sh_base;
    if (threadIdx.x == 0)
        lcount = 0;
    __syncthreads();
    const uint64_t nq = nw / 4; // whole uint4 groups; the tail (< 4 words) is handled by block 0 below
    const uint4 *q = reinterpret_cast<const uint4 *>(bits + w0);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x; i0 < nq; i0 += stride) { // block-uniform trip count
        const uint64_t i = i0 + threadIdx.x;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i < nq)
            v = q[i];
        const uint32_t n = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
        const bool any_in_block = __syncthreads_or(n != 0);
        if (!any_in_block)
            continue;
        // dense data: make room first (n <= 128 per lane, so one iteration can add up to 32768 entries;
        // handle that by letting lanes with many bits write straight to global memory)
        if (n) {
            if (n > 4u) {
                const unsigned long long base = atomicAdd(counter, (unsigned long long)n);
                unsigned long long pos = base;
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
                for (int c = 0; c < 4; c++) {
                    uint32_t x = w[c];
                    while (x) {
                        const int b = __ffs(x) - 1;
                        x &= x - 1u;
                        if (pos < cap)
                            out[pos] = ((w0 + 4 * i + c) << 5) + (uint64_t)b;
                        pos++;
                    }
                }
            } else {
                uint32_t slot = atomicAdd(&lcount, n); // <= 4 per lane, <= 1024 per iteration
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
                for (int c = 0; c < 4; c++) {
                    uint32_t x = w[c];
                    while (x) {
                        const int b = __ffs(x) - 1;
                        x &= x - 1u;
                        lbuf[slot++] = ((w0 + 4 * i + c) << 5) + (uint64_t)b;
                    }
                }
            }
        }
        __syncthreads();
        if (lcount >= EXTRACT_BUF) // block-uniform (read after the barrier)
            extract_flush(lbuf, &lcount, counter, out, cap, &sh_base);
    }
    extract_flush(lbuf, &lcount, counter, out, cap, &sh_base);
    if (blockIdx.x == 0 && threadIdx.x < (nw & 3)) {
        const uint64_t i = nq * 4 + threadIdx.x;
        uint32_t x = bits[w0 + i];
        while (x) {
            const int b = __ffs(x) - 1;
            x &= x - 1u;
            const unsigned long long pos = atomicAdd(counter, 1ull);
            if (pos < cap)
                out[pos] = ((w0 + i) << 5) + (uint64_t)b;
        }
    }
}

__global__ void or_keys_kernel(uint32_t *__restrict__ bits, const uint64_t *__restrict__ keys, uint64_t n, uint64_t nbits)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t h = keys[i];
        if (h < nbits)
            atomicOr(bits + (h >> 5), 1u << (h & 31u));
    }
}

int check_k(int k, bool need_odd)
{
    if (k < 1 || k > 31) {
        set_error("k=%d out of range 1..31", k);
        return BRX_ERR_ARG;
    }
    if (need_odd && (k & 1) == 0) {
        set_error("k=%d must be odd (parity-canonical k-mers; the reference forces odd k, src/cli.rs:277-279)", k);
        return BRX_ERR_ARG;
    }
    return BRX_OK;
}

int alloc_set(int k, int device, bool zero, brx_set **out, bool sparse = false)
{
    BRX_TRY(check_k(k, false));
    BRX_TRY(use_device(device));
    brx_set *s = new brx_set();
    s->k = k;
    s->device = device;
    s->nwords = set_nwords(k);
    s->d_bits = nullptr;
    s->sparse = sparse || sparse_k(k);
    if (s->sparse) { // no bit vector: an empty key list is the empty set
        s->nwords = 0;
        hipError_t e = hipMalloc((void **)&s->d_keylist_n, 8);
        if (e == hipSuccess)
            e = hipMemset(s->d_keylist_n, 0, 8);
        if (e != hipSuccess) {
            set_error("sparse set alloc: %s", hipGetErrorString(e));
            delete s;
            return BRX_ERR_NOMEM;
        }
        s->keylist_valid = true;
        *out = s;
        return BRX_OK;
    }
    hipError_t e = hipMalloc((void **)&s->d_bits, s->nwords * 4);
    if (e != hipSuccess) {
        set_error("hipMalloc(%llu B bitset, k=%d): %s", (unsigned long long)(s->nwords * 4), k, hipGetErrorString(e));
        delete s;
        return BRX_ERR_NOMEM;
    }
    if (zero) {
        e = hipMemset(s->d_bits, 0, s->nwords * 4);
        if (e != hipSuccess) {
            set_error("hipMemset bitset: %s", hipGetErrorString(e));
            (void)hipFree(s->d_bits);
            delete s;
            return BRX_ERR_HIP;
        }
    }
    *out = s;
    return BRX_OK;
}

int need_bits(const brx_set *set, const char *what) { return ensure_bits(set, nullptr, what); }

int grid_for(uint64_t items, int per_block, int cap = 256 * 8)
{
    uint64_t g = (items + per_block - 1) / per_block;
    if (g < 1)
        g = 1;
    if (g > (uint64_t)cap)
        g = cap;
    return (int)g;
}

} // namespace

namespace brx {
bool part_supported(int k);
int part_begin(brx_counter *c);
void part_free(brx_counter *c);
int part_reset(brx_counter *c);
int part_add_batch(brx_counter *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n_reads,
                   uint64_t total_bases, hipStream_t s);
int part_finish_into(brx_counter *c, uint32_t abundance, hipStream_t s, brx_set *dst);
int part_spectrum(brx_counter *c, hipStream_t s, unsigned long long *d_hist);
int part_l1_view(brx_counter *c, void **d_keys, void **d_l1off, uint32_t *n_buckets, uint64_t *n_keys);
int part_add_partitioned(brx_counter *c, const uint32_t *d_keys, const uint64_t *d_l1off, uint64_t n_keys);
// used by the correction chain and the host-pointer entry points
int upload_batch(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, uint8_t **d_bases, uint64_t *bases_cap,
                 uint64_t **d_off, uint64_t *off_cap, uint64_t *total, hipStream_t stream)
{
    const uint64_t base0 = offsets[0];
    const uint64_t tot = offsets[n_reads] - base0;
    for (uint32_t r = 0; r < n_reads; r++)
        if (offsets[r + 1] < offsets[r]) {
            set_error("offsets not monotone at read %u", r);
            return BRX_ERR_ARG;
        }
    if (tot + 64 > *bases_cap) {
        if (*d_bases)
            (void)hipFree(*d_bases);
        *d_bases = nullptr;
        *bases_cap = 0;
        BRX_HIP(hipMalloc((void **)d_bases, tot + 64));
        *bases_cap = tot + 64;
    }
    if ((uint64_t)n_reads + 1 > *off_cap) {
        if (*d_off)
            (void)hipFree(*d_off);
        *d_off = nullptr;
        *off_cap = 0;
        BRX_HIP(hipMalloc((void **)d_off, ((uint64_t)n_reads + 1) * 8));
        *off_cap = (uint64_t)n_reads + 1;
    }
    if (tot)
        BRX_HIP(hipMemcpyAsync(*d_bases, bases + base0, tot, hipMemcpyHostToDevice, stream));
    if (base0 == 0) {
        BRX_HIP(hipMemcpyAsync(*d_off, offsets, ((uint64_t)n_reads + 1) * 8, hipMemcpyHostToDevice, stream));
        BRX_HIP(hipStreamSynchronize(stream));
    } else {
        std::vector<uint64_t> rel((size_t)n_reads + 1);
        for (uint32_t r = 0; r <= n_reads; r++)
            rel[r] = offsets[r] - base0;
        BRX_HIP(hipMemcpyAsync(*d_off, rel.data(), rel.size() * 8, hipMemcpyHostToDevice, stream));
        BRX_HIP(hipStreamSynchronize(stream));
    }
    *total = tot;
    return BRX_OK;
}
} // namespace brx

extern "C" {

int brx_set_new(uint8_t k, int device, brx_set_t **out)
{
    if (!out)
        return BRX_ERR_ARG;
    return alloc_set(k, device, true, out);
}

int brx_set_new_from_solid_bytes(const uint8_t *buf, size_t len, int device, brx_set_t **out)
{
    if (!buf || !out || len < 1) {
        set_error("null/empty .solid buffer");
        return BRX_ERR_ARG;
    }
    const int k = buf[0];
    if (k < 1 || k > 31 || len - 1 != set_nbytes_file(k)) {
        set_error(".solid stream: k=%d, %zu payload bytes, expected %llu", k, len - 1,
                  (unsigned long long)((k >= 1 && k <= 31) ? set_nbytes_file(k) : 0));
        return BRX_ERR_FORMAT;
    }
    if (sparse_k(k)) {
        set_error(".solid stream with k=%d: sets of that size are sparse here (no bit vector)", k);
        return BRX_ERR_UNSUPPORTED;
    }
    brx_set *s = nullptr;
    BRX_TRY(alloc_set(k, device, set_nbytes_file(k) < 4, &s));
    hipError_t e = hipMemcpy(s->d_bits, buf + 1, len - 1, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("hipMemcpy bitset H2D: %s", hipGetErrorString(e));
        brx_set_free(s);
        return BRX_ERR_HIP;
    }
    *out = s;
    return BRX_OK;
}

int brx_set_insert_batch(brx_set_t *set, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads)
{
    if (!set || !offsets || (!bases && n_reads)) {
        set_error("null argument");
        return BRX_ERR_ARG;
    }
    BRX_TRY(check_k(set->k, true));
    BRX_TRY(use_device(set->device));
    if (n_reads == 0)
        return BRX_OK;
    if (set->sparse) { // Hash::from_fasta (src/set/hash.rs:40-60): straight into the chained table
        uint8_t *d_b = nullptr;
        uint64_t *d_o = nullptr;
        uint64_t bc = 0, oc = 0, tot = 0;
        int st = upload_batch(bases, offsets, n_reads, &d_b, &bc, &d_o, &oc, &tot, 0);
        if (st == BRX_OK)
            st = index_insert_reads(set, d_b, d_o, n_reads, tot, 0);
        if (d_b)
            (void)hipFree(d_b);
        if (d_o)
            (void)hipFree(d_o);
        return st;
    }
    BRX_TRY(need_bits(set, "insert_batch"));
    index_invalidate(set);
    uint8_t *d_b = nullptr;
    uint64_t *d_o = nullptr;
    uint64_t bc = 0, oc = 0, tot = 0;
    int st = upload_batch(bases, offsets, n_reads, &d_b, &bc, &d_o, &oc, &tot, 0);
    if (st == BRX_OK) {
        KernelTimer t("insert", 0);
        st = flat_presence_insert(d_b, d_o, n_reads, tot, set->k, set->d_bits, nullptr, 0, 0, nullptr, 0);
    }
    hipError_t e = hipDeviceSynchronize();
    if (d_b)
        (void)hipFree(d_b);
    if (d_o)
        (void)hipFree(d_o);
    if (st != BRX_OK)
        return st;
    if (e != hipSuccess) {
        set_error("insert kernel: %s", hipGetErrorString(e));
        return BRX_ERR_HIP;
    }
    return BRX_OK;
}

int brx_set_insert_batch_device(brx_set_t *set, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n_reads,
                                uint64_t total_bases, void *stream)
{
    if (!set || !d_offsets || (!d_bases && total_bases)) {
        set_error("null argument");
        return BRX_ERR_ARG;
    }
    BRX_TRY(check_k(set->k, true));
    BRX_TRY(use_device(set->device));
    if (n_reads == 0)
        return BRX_OK;
    hipStream_t s = (hipStream_t)stream;
    if (set->sparse)
        return index_insert_reads(set, d_bases, d_offsets, n_reads, total_bases, s);
    BRX_TRY(ensure_bits(set, s, "insert_batch"));
    index_invalidate(set);
    {
        KernelTimer t("insert", s);
        BRX_TRY(flat_presence_insert(d_bases, d_offsets, n_reads, total_bases, set->k, set->d_bits, nullptr, 0, 0, nullptr, s));
    }
    return BRX_OK;
}

int brx_set_set(brx_set_t *set, uint64_t forward_kmer, bool value)
{
    if (!set)
        return BRX_ERR_ARG;
    BRX_TRY(need_bits(set, "set"));
    BRX_TRY(use_device(set->device));
    index_invalidate(set);
    const uint64_t h = khash(forward_kmer & kmask(set->k), set->k);
    set_bit_kernel<<<1, 1>>>(set->d_bits, h, value ? 1 : 0);
    BRX_HIP(hipDeviceSynchronize());
    return BRX_OK;
}

bool brx_set_get(const brx_set_t *set, uint64_t forward_kmer)
{
    if (!set || use_device(set->device) != BRX_OK)
        return false;
    if (no_bits(set)) {
        uint8_t o = 0;
        return brx_set_get_batch(set, &forward_kmer, 1, &o) == BRX_OK && o;
    }
    const uint64_t h = khash(forward_kmer & kmask(set->k), set->k);
    uint32_t w = 0;
    if (hipMemcpy(&w, set->d_bits + (h >> 5), 4, hipMemcpyDeviceToHost) != hipSuccess)
        return false;
    return (w >> (h & 31u)) & 1u;
}

int brx_set_get_batch(const brx_set_t *set, const uint64_t *forward_kmers, uint32_t n, uint8_t *out)
{
    if (!set || (!forward_kmers && n) || (!out && n))
        return BRX_ERR_ARG;
    BRX_TRY(use_device(set->device));
    if (!n)
        return BRX_OK;
    if (no_bits(set)) { // the probe index is all there is (right now)
        BRX_TRY(index_ensure(set, nullptr));
        return brx_set_get_batch_indexed(set, forward_kmers, n, out, nullptr);
    }
    uint64_t *d_k = nullptr;
    uint8_t *d_o = nullptr;
    BRX_HIP(hipMalloc((void **)&d_k, (uint64_t)n * 8));
    hipError_t e = hipMalloc((void **)&d_o, n);
    if (e != hipSuccess) {
        (void)hipFree(d_k);
        set_error("hipMalloc: %s", hipGetErrorString(e));
        return BRX_ERR_NOMEM;
    }
    e = hipMemcpy(d_k, forward_kmers, (uint64_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        get_kernel<<<(n + 255) / 256, 256>>>(set->d_bits, d_k, n, set->k, d_o);
        e = hipMemcpy(out, d_o, n, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_k);
    (void)hipFree(d_o);
    if (e != hipSuccess) {
        set_error("get_batch: %s", hipGetErrorString(e));
        return BRX_ERR_HIP;
    }
    return BRX_OK;
}

uint8_t brx_set_k(const brx_set_t *set) { return set ? (uint8_t)set->k : 0; }

int brx_set_device(const brx_set_t *set) { return set ? set->device : -1; }

int brx_set_export_solid_bytes(const brx_set_t *set, uint8_t *buf, size_t cap, size_t *len)
{
    if (!set || !len)
        return BRX_ERR_ARG;
    BRX_TRY(need_bits(set, "export (.solid)"));
    const size_t need = 1 + (size_t)set_nbytes_file(set->k);
    *len = need;
    if (!buf || cap < need) {
        set_error("export needs %zu bytes, buffer has %zu", need, cap);
        return BRX_ERR_OVERFLOW;
    }
    BRX_TRY(use_device(set->device));
    buf[0] = (uint8_t)set->k;
    BRX_HIP(hipMemcpy(buf + 1, set->d_bits, need - 1, hipMemcpyDeviceToHost));
    return BRX_OK;
}

int brx_set_popcount(const brx_set_t *set, uint64_t *n_set_bits)
{
    if (!set || !n_set_bits)
        return BRX_ERR_ARG;
    BRX_TRY(use_device(set->device));
    if (no_bits(set)) {
        if (set->idx_valid) {
            *n_set_bits = set->idx_keys;
            return BRX_OK;
        }
        unsigned long long nl = 0;
        if (set->keylist_valid)
            BRX_HIP(hipMemcpy(&nl, set->d_keylist_n, 8, hipMemcpyDeviceToHost));
        *n_set_bits = nl;
        return BRX_OK;
    }
    unsigned long long *d_t = nullptr;
    BRX_HIP(hipMalloc((void **)&d_t, 8));
    BRX_HIP(hipMemset(d_t, 0, 8));
    popcount_kernel<<<grid_for(set->nwords, 256 * 8), 256>>>(set->d_bits, set->nwords, d_t);
    unsigned long long t = 0;
    hipError_t e = hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_t);
    if (e != hipSuccess) {
        set_error("popcount: %s", hipGetErrorString(e));
        return BRX_ERR_HIP;
    }
    *n_set_bits = t;
    return BRX_OK;
}

int brx_set_device_bits(const brx_set_t *set, void **d_bits, uint64_t *n_bytes)
{
    if (!set || !d_bits || !n_bytes)
        return BRX_ERR_ARG;
    BRX_TRY(need_bits(set, "device_bits"));
    *d_bits = set->d_bits;
    *n_bytes = set->nwords * 4;
    return BRX_OK;
}

int brx_set_sparse(const brx_set_t *set) { return set && set->sparse ? 1 : 0; }

int brx_set_bits_state(const brx_set_t *set) { return !set ? -1 : (set->sparse ? 2 : (set->bits_stale ? 1 : 0)); }

void brx_set_free(brx_set_t *set)
{
    if (!set)
        return;
    if (use_device(set->device) == BRX_OK) {
        if (set->d_bits)
            (void)hipFree(set->d_bits);
        if (set->d_lines)
            (void)hipFree(set->d_lines);
        if (set->d_keylist)
            (void)hipFree(set->d_keylist);
        if (set->d_keylist_n)
            (void)hipFree(set->d_keylist_n);
        if (set->d_succ)
            (void)hipFree(set->d_succ);
    }
    delete set;
}

// ---- counting ---------------------------------------------------------------------------------

int brx_set_count_begin(uint8_t k, int device, int strategy, brx_counter_t **out)
{
    if (!out)
        return BRX_ERR_ARG;
    BRX_TRY(check_k(k, true));
    BRX_TRY(use_device(device));
    if (strategy == BRX_COUNT_AUTO) // the table-free path as soon as the u8 table would not fit the caches
        strategy = (part_supported(k) && k >= 15) ? BRX_COUNT_SORTED : BRX_COUNT_DENSE;
    if (strategy != BRX_COUNT_DENSE && strategy != BRX_COUNT_SORTED) {
        set_error("unknown count strategy %d", strategy);
        return BRX_ERR_ARG;
    }
    if (strategy == BRX_COUNT_SORTED && !part_supported(k)) {
        set_error("partitioned count strategy supports 7 <= k <= 19 (got %d)", (int)k);
        return BRX_ERR_UNSUPPORTED;
    }
    brx_counter *c = new brx_counter();
    c->k = k;
    c->device = device;
    c->strategy = strategy;
    c->d_counts = nullptr;
    c->part = nullptr;
    c->d_keys = nullptr;
    c->n_keys = c->cap_keys = 0;
    c->stream = nullptr;
    c->count_bytes = set_nbits(k) < 32 ? 32 : set_nbits(k); // one u8 per canonical k-mer, >= one output word
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("stream create: %s", hipGetErrorString(e));
        brx_counter_free(c);
        return BRX_ERR_HIP;
    }
    if (strategy == BRX_COUNT_SORTED) {
        int st = part_begin(c);
        if (st != BRX_OK) {
            brx_counter_free(c);
            return st;
        }
        *out = c;
        return BRX_OK;
    }
    e = hipMalloc((void **)&c->d_counts, c->count_bytes);
    if (e == hipSuccess) {
        KernelTimer t("count_zero", c->stream);
        e = hipMemsetAsync(c->d_counts, 0, c->count_bytes, c->stream);
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        set_error("counter alloc (%llu B u8 table, k=%d): %s", (unsigned long long)c->count_bytes, (int)k,
                  hipGetErrorString(e));
        brx_counter_free(c);
        return e == hipErrorOutOfMemory ? BRX_ERR_NOMEM : BRX_ERR_HIP;
    }
    *out = c;
    return BRX_OK;
}

int brx_set_count_add_batch_device(brx_counter_t *c, const uint8_t *d_bases, const uint64_t *d_offsets,
                                   uint32_t n_reads, uint64_t total_bases, void *stream)
{
    (void)total_bases;
    if (!c || !d_offsets || (!d_bases && n_reads))
        return BRX_ERR_ARG;
    BRX_TRY(use_device(c->device));
    if (!n_reads)
        return BRX_OK;
    hipStream_t s = (hipStream_t)stream;
    if (c->strategy == BRX_COUNT_SORTED)
        return part_add_batch(c, d_bases, d_offsets, n_reads, total_bases, s);
    {
        KernelTimer t("count_dense", s);
        kmer_scatter_kernel<0><<<grid_for(n_reads, 1, 1 << 20), 256, 0, s>>>(d_bases, d_offsets, n_reads, c->k,
                                                                          c->d_counts);
    }
    BRX_HIP(hipGetLastError());
    return BRX_OK;
}

int brx_set_count_add_batch(brx_counter_t *c, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads)
{
    if (!c || !offsets || (!bases && n_reads))
        return BRX_ERR_ARG;
    BRX_TRY(use_device(c->device));
    if (!n_reads)
        return BRX_OK;
    std::lock_guard<std::mutex> g(c->mu);
    uint8_t *d_b = nullptr;
    uint64_t *d_o = nullptr;
    uint64_t bc = 0, oc = 0, tot = 0;
    int st = upload_batch(bases, offsets, n_reads, &d_b, &bc, &d_o, &oc, &tot, c->stream);
    if (st == BRX_OK)
        st = brx_set_count_add_batch_device(c, d_b, d_o, n_reads, tot, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (d_b)
        (void)hipFree(d_b);
    if (d_o)
        (void)hipFree(d_o);
    if (st != BRX_OK)
        return st;
    if (e != hipSuccess) {
        set_error("count kernel: %s", hipGetErrorString(e));
        return BRX_ERR_HIP;
    }
    return BRX_OK;
}

int brx_counter_reset(brx_counter_t *c, void *stream)
{
    if (!c)
        return BRX_ERR_ARG;
    BRX_TRY(use_device(c->device));
    hipStream_t s = (hipStream_t)stream;
    if (c->strategy == BRX_COUNT_SORTED)
        return part_reset(c);
    if (c->d_counts) {
        KernelTimer t("count_zero", s);
        BRX_HIP(hipMemsetAsync(c->d_counts, 0, c->count_bytes, s));
    }
    c->n_keys = 0;
    return BRX_OK;
}

int brx_set_count_finish_into(brx_counter_t *c, uint8_t abundance, void *stream, brx_set_t *dst)
{
    if (!c || !dst)
        return BRX_ERR_ARG;
    if (dst->k != c->k || dst->device != c->device) {
        set_error("finish_into: destination set has k=%d device=%d, counter has k=%d device=%d", dst->k, dst->device,
                  c->k, c->device);
        return BRX_ERR_ARG;
    }
    BRX_TRY(use_device(c->device));
    index_invalidate(dst);
    hipStream_t s = (hipStream_t)stream; // nullptr = the legacy default stream, like any HIP API
    if (c->strategy == BRX_COUNT_SORTED)
        return part_finish_into(c, abundance, s, dst);
    if (dst->sparse) {
        set_error("the dense count strategy needs a bit vector; a sparse set (k=%d) has none", dst->k);
        return BRX_ERR_UNSUPPORTED;
    }
    dst->bits_stale = false; // every word is written below
    {
        KernelTimer t("threshold", s);
        threshold_kernel<<<grid_for(dst->nwords, 256, 256 * 16), 256, 0, s>>>((const uint4 *)c->d_counts, dst->nwords,
                                                                             abundance, dst->d_bits);
    }
    BRX_HIP(hipGetLastError());
    return BRX_OK;
}

int brx_set_count_finish(brx_counter_t *c, uint8_t abundance, void *stream, brx_set_t **out)
{
    if (!c || !out)
        return BRX_ERR_ARG;
    brx_set *set = nullptr;
    const char *fs = getenv("BRX_FORCE_SPARSE"); // tests: the sparse representation at small k
    BRX_TRY(alloc_set(c->k, c->device, false, &set, fs && *fs == '1' && c->strategy == BRX_COUNT_SORTED));
    int st = brx_set_count_finish_into(c, abundance, stream, set);
    if (st == BRX_OK) {
        hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) {
            set_error("threshold kernel: %s", hipGetErrorString(e));
            st = BRX_ERR_HIP;
        }
    }
    if (st != BRX_OK) {
        brx_set_free(set);
        return st;
    }
    *out = set;
    return BRX_OK;
}

int brx_counter_spectrum(brx_counter_t *c, uint64_t *hist256, void *stream)
{
    if (!c || !hist256)
        return BRX_ERR_ARG;
    BRX_TRY(use_device(c->device));
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *d_h = nullptr;
    BRX_HIP(hipMalloc((void **)&d_h, 256 * 8));
    hipError_t e = hipMemsetAsync(d_h, 0, 256 * 8, s);
    const uint64_t nbytes = set_nbits(c->k); // real entries (the table may be padded to 32 B)
    if (c->strategy == BRX_COUNT_SORTED) {
        // no table to read: the partitioned keys are counted bucket by bucket (bins 1..255); bin 0 is the rest
        int st = e == hipSuccess ? part_spectrum(c, s, d_h) : BRX_ERR_HIP;
        if (st == BRX_OK) {
            e = hipMemcpyAsync(hist256, d_h, 256 * 8, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess)
                e = hipStreamSynchronize(s);
            if (e != hipSuccess) {
                set_error("spectrum: %s", hipGetErrorString(e));
                st = BRX_ERR_HIP;
            }
        }
        (void)hipFree(d_h);
        if (st != BRX_OK)
            return st;
        uint64_t seen = 0;
        for (int i = 1; i < 256; i++)
            seen += hist256[i];
        hist256[0] = nbytes - seen;
        return BRX_OK;
    }
    if (e == hipSuccess) {
        KernelTimer t("spectrum", s);
        // padded tail (k <= 2) would add zeros: count whole words of real entries only
        spectrum_kernel<<<grid_for(nbytes / 4 ? nbytes / 4 : 1, 256 * 16, 2048), 256, 0, s>>>(c->d_counts, nbytes / 4 ? nbytes / 4 : 1, d_h);
        e = hipMemcpyAsync(hist256, d_h, 256 * 8, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    (void)hipFree(d_h);
    if (e != hipSuccess) {
        set_error("spectrum: %s", hipGetErrorString(e));
        return BRX_ERR_HIP;
    }
    if (nbytes < 4) // k = 1: 2 entries in a padded word; remove the 2 padding zeros of the first word
        hist256[0] -= 4 - nbytes;
    return BRX_OK;
}

int brx_counter_device_counts(brx_counter_t *c, void **d_counts, uint64_t *n_bytes)
{
    if (!c || !d_counts || !n_bytes)
        return BRX_ERR_ARG;
    if (c->strategy != BRX_COUNT_DENSE) {
        set_error("counter is not dense");
        return BRX_ERR_ARG;
    }
    *d_counts = c->d_counts;
    *n_bytes = c->count_bytes;
    return BRX_OK;
}

int brx_counter_load_counts(brx_counter_t *c, uint64_t first, const uint8_t *counts, uint64_t n)
{
    if (!c || (!counts && n))
        return BRX_ERR_ARG;
    if (c->strategy != BRX_COUNT_DENSE) {
        set_error("load_counts needs the dense count strategy");
        return BRX_ERR_UNSUPPORTED;
    }
    const uint64_t entries = set_nbits(c->k);
    if (first > entries || n > entries - first) {
        set_error("load_counts: [%llu, +%llu) is outside the %llu counters of k=%d", (unsigned long long)first,
                  (unsigned long long)n, (unsigned long long)entries, c->k);
        return BRX_ERR_ARG;
    }
    BRX_TRY(use_device(c->device));
    if (n)
        BRX_HIP(hipMemcpy((uint8_t *)c->d_counts + first, counts, n, hipMemcpyHostToDevice));
    return BRX_OK;
}

int brx_counter_clamp(brx_counter_t *c, uint8_t cap, void *stream)
{
    if (!c || c->strategy != BRX_COUNT_DENSE)
        return BRX_ERR_ARG;
    BRX_TRY(use_device(c->device));
    hipStream_t s = (hipStream_t)stream; // nullptr = the legacy default stream, like any HIP API
    {
        KernelTimer t("clamp", s);
        clamp_kernel<<<grid_for(c->count_bytes / 16, 256, 256 * 16), 256, 0, s>>>((uint4 *)c->d_counts,
                                                                                c->count_bytes / 16, cap);
    }
    BRX_HIP(hipStreamSynchronize(s));
    return BRX_OK;
}

int brx_counter_l1_view(brx_counter_t *c, void **d_keys, void **d_l1off, uint32_t *n_buckets, uint64_t *n_keys)
{
    if (!c || !d_keys || !d_l1off || !n_buckets || !n_keys)
        return BRX_ERR_ARG;
    if (c->strategy != BRX_COUNT_SORTED) {
        set_error("l1_view: counter is not partitioned");
        return BRX_ERR_ARG;
    }
    return part_l1_view(c, d_keys, d_l1off, n_buckets, n_keys);
}

int brx_counter_add_partitioned_device(brx_counter_t *c, const uint32_t *d_keys, const uint64_t *d_l1off, uint64_t n_keys)
{
    if (!c || (!d_keys && n_keys) || !d_l1off)
        return BRX_ERR_ARG;
    if (c->strategy != BRX_COUNT_SORTED) {
        set_error("add_partitioned: counter is not partitioned");
        return BRX_ERR_ARG;
    }
    return part_add_partitioned(c, d_keys, d_l1off, n_keys);
}

int brx_set_extract_keys_device(const brx_set_t *set, uint64_t first_hash, uint64_t n_hashes, uint64_t *d_out, uint64_t cap,
                                uint64_t *n_out, void *stream)
{
    if (!set || !n_out || (!d_out && cap)) {
        set_error("null argument");
        return BRX_ERR_ARG;
    }
    BRX_TRY(need_bits(set, "extract_keys (use brx_set_keylist_device)"));
    const uint64_t nbits = set->nwords * 32;
    if ((first_hash & 127u) || (n_hashes & 31u) || first_hash + n_hashes > nbits) {
        set_error("extract range must start on a multiple of 128 hashes, span a multiple of 32, and lie inside the set");
        return BRX_ERR_ARG;
    }
    BRX_TRY(use_device(set->device));
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *d_cnt = nullptr;
    BRX_HIP(hipMalloc((void **)&d_cnt, 8));
    hipError_t e = hipMemsetAsync(d_cnt, 0, 8, s);
    const uint64_t nw = n_hashes / 32;
    if (e == hipSuccess && nw) {
        KernelTimer t("extract_keys", s);
        extract_kernel<<<grid_for(nw / 4 + 1, 256, 256 * 8), 256, 0, s>>>(set->d_bits, first_hash / 32, nw, d_cnt, d_out, cap);
    }
    unsigned long long n = 0;
    if (e == hipSuccess)
        e = hipMemcpyAsync(&n, d_cnt, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    (void)hipFree(d_cnt);
    if (e != hipSuccess) {
        set_error("extract: %s", hipGetErrorString(e));
        return BRX_ERR_HIP;
    }
    *n_out = n;
    if (n > cap) {
        set_error("extract found %llu keys, buffer holds %llu", n, (unsigned long long)cap);
        return BRX_ERR_OVERFLOW;
    }
    return BRX_OK;
}

int brx_set_or_keys_device(brx_set_t *set, const uint64_t *d_keys, uint64_t n, void *stream)
{
    if (!set || (!d_keys && n))
        return BRX_ERR_ARG;
    BRX_TRY(need_bits(set, "or_keys (use brx_set_index_build_from_keys_device)"));
    BRX_TRY(use_device(set->device));
    if (!n)
        return BRX_OK;
    index_invalidate(set);
    hipStream_t s = (hipStream_t)stream;
    {
        KernelTimer t("or_keys", s);
        or_keys_kernel<<<grid_for(n, 256, 256 * 8), 256, 0, s>>>(set->d_bits, d_keys, n, set->nwords * 32);
    }
    BRX_HIP(hipGetLastError());
    return BRX_OK;
}

void brx_counter_free(brx_counter_t *c)
{
    if (!c)
        return;
    if (use_device(c->device) == BRX_OK) {
        if (c->part)
            part_free(c);
        if (c->d_counts)
            (void)hipFree(c->d_counts);
        if (c->d_keys)
            (void)hipFree(c->d_keys);
        if (c->stream)
            (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

} // extern "C"
