// Probe index of a solid set: build + test entry points (layout and probe in brx_index.hpp).
#include "brx_internal.hpp"
#include "brx_index.hpp"

#include <stdlib.h>

using namespace brx;

namespace {

// one key per thread: claim a slot of the key's line; a full line is flagged as overflowed and the key is
// left to the bit vector (CHAIN = false) or goes on to the next line (CHAIN = true: sparse sets)
template <bool CHAIN>
__global__ __launch_bounds__(256) void index_insert_kernel(const uint64_t *__restrict__ keys, uint64_t n, uint64_t *__restrict__ lines,
                                                           uint32_t line_shift, uint32_t m, uint32_t w, int k,
                                                           unsigned long long *__restrict__ n_overflow, uint32_t *__restrict__ line_bits)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t line_mask = 0xffffffffu >> line_shift;
    uint32_t lost = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t h = keys[i];
        const uint64_t canon = (h << 1) | (uint64_t)(popc64(h) & 1); // even popcount (brx_kmer.hpp)
        const uint64_t rc = revcomp(canon, k);
        uint32_t line = index_line_of(minimizer_of(canon, rc, m, w), line_shift);
        bool counted = false;
        // a chained walk visits at most every line once (the host checks that the table has room for its keys;
        // this bound keeps a mis-sized table from spinning the GPU for ever: the key is then simply lost and counted)
        for (uint32_t walked = 0; walked <= line_mask; walked++) {
            unsigned long long *L = reinterpret_cast<unsigned long long *>(lines) + (uint64_t)line * 8ull;
            const unsigned long long seen = __hip_atomic_load(L + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t slot = IDX_SLOTS;
            if ((uint32_t)seen < (uint32_t)IDX_SLOTS) // full lines are not counted further: the counter stays small
                slot = (uint32_t)atomicAdd(L + 7, 1ull);
            if (slot < (uint32_t)IDX_SLOTS) {
                L[slot] = h + 1ull;
                if (slot == 0u) // the line's first key: mark the line as occupied (brx_index.hpp: line_bits)
                    atomicOr(line_bits + (line >> 5), 1u << (line & 31u));
                break;
            }
            {   // flag the line; at the key's home line also leave the key's signature bit
                const unsigned long long want = (unsigned long long)IDX_OVERFLOW | (counted ? 0ull : (unsigned long long)idx_sig_bit(h + 1ull));
                if ((seen & want) != want)
                    atomicOr(L + 7, want);
            }
            if (!counted) {
                lost++;
                counted = true;
            }
            if (!CHAIN)
                break;
            line = (line + 1u) & line_mask; // the table has more slots than keys: this ends
        }
    }
    for (int d = 32; d > 0; d >>= 1)
        lost += __shfl_down(lost, d);
    if ((threadIdx.x & 63) == 0 && lost)
        atomicAdd(n_overflow, (unsigned long long)lost);
}

__global__ void index_get_kernel(IdxView v, const uint32_t *__restrict__ bits, const uint64_t *__restrict__ kmers, uint32_t n, int k,
                                 uint8_t *__restrict__ out, unsigned long long *__restrict__ n_fallback)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint64_t fwd = kmers[i] & kmask(k);
    int r = index_probe(v, fwd, k);
    if (r == 2) {
        atomicAdd(n_fallback, 1ull);
        if (bits) {
            const uint64_t h = khash(fwd, k);
            r = (bits[h >> 5] >> (h & 31u)) & 1u;
        } else {
            for (uint32_t hop = 1; r == 2; hop++) // sparse set: the key was chained into a following line
                r = index_probe(v, fwd, k, hop);
        }
    }
    out[i] = (uint8_t)r;
}

// moves every key of one table into another (growing an insert-built set)
__global__ __launch_bounds__(256) void table_rehash_kernel(const uint64_t *__restrict__ old_lines, uint64_t n_old_lines, int k,
                                                           uint64_t *__restrict__ lines, uint32_t line_shift, uint32_t m, uint32_t w)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_old_lines * IDX_SLOTS; i += stride) {
        const uint64_t v = old_lines[(i / IDX_SLOTS) * 8ull + (i % IDX_SLOTS)];
        if (v) {
            const uint64_t h = v - 1ull;
            (void)table_find_or_insert(lines, line_shift, m, w, k, (h << 1) | (uint64_t)(popc64(h) & 1));
        }
    }
}

__global__ __launch_bounds__(256) void table_to_bits_kernel(const uint64_t *__restrict__ lines, uint64_t n_lines, uint32_t *__restrict__ bits,
                                                            uint64_t nbits)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines * IDX_SLOTS; i += stride) {
        const uint64_t v = lines[(i / IDX_SLOTS) * 8ull + (i % IDX_SLOTS)];
        if (v && v - 1ull < nbits)
            atomicOr(bits + ((v - 1ull) >> 5), 1u << ((v - 1ull) & 31u));
    }
}

__global__ void keys_to_bits_kernel(const uint64_t *__restrict__ keys, const unsigned long long *__restrict__ n_ptr, uint32_t *__restrict__ bits,
                                    uint64_t nbits)
{
    const uint64_t n = *n_ptr;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t h = keys[i];
        if (h < nbits)
            atomicOr(bits + (h >> 5), 1u << (h & 31u));
    }
}

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

} // namespace

namespace brx {

bool index_wanted(int k)
{
    if (sparse_k(k))
        return true; // nothing else to probe

    // below k = 15 the whole bitset (<= 4 MiB at k = 13) lives in every XCD's L2 and a plain probe is cheaper
    // (BRX_INDEX_MIN_K lowers the threshold: the parity tests run every corrector through the index at small k)
    return (k & 1) && k >= env_int("BRX_INDEX_MIN_K", 15) && k >= 5 && env_int("BRX_INDEX", 1) != 0;
}

// smallest odd m whose 4^m / 2 canonical m-mers outnumber the keys 16 to 1 (a minimizer shared by many
// genome positions piles all their k-mers into one line), at least 2 windows
int index_auto_m(int k, uint64_t n_keys)
{
    int m = 3;
    while (m < IDX_AUTO_MAX_M && (1ull << (2 * m - 1)) < 16ull * n_keys)
        m += 2;
    // beyond 2^25 keys even the 15-mers (2^29 canonical ones) are fewer than 16 per key: 16-mers (the longest a 32-bit
    // window holds; no masking in the probe either).  Measured at 0.96 G keys (k = 19): 34 % of the keys overflow their
    // line with m = 15, 14 % with m = 16, and a correction pass takes 231 ms instead of 428 (2^28 lines both).
    if (m == IDX_AUTO_MAX_M && (1ull << (2 * IDX_AUTO_MAX_M - 1)) < 16ull * n_keys && k - IDX_MAX_M + 1 >= 3)
        return IDX_MAX_M;
    if (m > k - 2)
        m = k - 2;
    if (!(m & 1))
        m--;
    return m < 3 ? 3 : m;
}

static int index_build_locked(brx_set *set, const uint64_t *d_keys, uint64_t n, int m, int log_lines, hipStream_t s)
{
    const int k = set->k;
    if (!(k & 1) || k < 5 || k > 31) {
        set_error("probe index needs odd 5 <= k <= 31 (k=%d)", k);
        return BRX_ERR_UNSUPPORTED;
    }
    BRX_TRY(use_device(set->device));
    if (m <= 0)
        m = env_int("BRX_INDEX_M", 0);
    if (m <= 0)
        m = index_auto_m(k, n);
    if ((!(m & 1) && m != 16) || m < 3 || m > IDX_MAX_M || k - m + 1 < 2) {
        set_error("probe index: minimizer length %d must be odd (or 16), 3..%d and shorter than the %d-mer", m, IDX_MAX_M, k);
        return BRX_ERR_ARG;
    }
    if (log_lines <= 0)
        log_lines = env_int("BRX_INDEX_LOG_LINES", 0);
    if (log_lines <= 0) {
        // about one key per 7-slot line: a minimizer brings up to k-m+1 keys at once, and a line that
        // overflows costs its probes a second round
        log_lines = 10;
        while (log_lines < 29 && (1ull << log_lines) < n + n / 2) // (2^29 lines = 32 GiB; 0.96 G keys: 166 ms per pass against 231 at 2^28)
            log_lines++;
    }
    while (no_bits(set) && log_lines < 30 && (7ull << log_lines) < n + n / 4)
        log_lines++; // a chained table must have room for every key
    if (log_lines < 4 || log_lines > 30) {
        set_error("probe index: log2(lines)=%d out of range 4..30", log_lines);
        return BRX_ERR_ARG;
    }
    if (no_bits(set) && (7ull << log_lines) < n + n / 4) {
        // a chained table that cannot hold its keys would send index_insert_kernel<true> round the table for ever
        set_error("set without bit vector: %llu solid k-mers do not fit a chained table of 2^%d lines (7 slots each)",
                  (unsigned long long)n, log_lines);
        return BRX_ERR_NOMEM;
    }
    set->idx_valid = false;
    set->idx_gen++;
    const uint64_t n_lines = 1ull << log_lines;
    if (set->lines_alloc < n_lines) {
        if (set->d_lines)
            (void)hipFree(set->d_lines);
        set->d_lines = nullptr;
        set->lines_alloc = 0;
        hipError_t e = hipMalloc((void **)&set->d_lines, n_lines * 64ull + n_lines / 8ull + 64ull); // lines + occupancy bits
        if (e != hipSuccess) {
            set_error("hipMalloc(%llu B probe index): %s", (unsigned long long)(n_lines * 64ull), hipGetErrorString(e));
            return BRX_ERR_NOMEM;
        }
        set->lines_alloc = n_lines;
    }
    unsigned long long *d_ovf = nullptr;
    BRX_HIP(hipMalloc((void **)&d_ovf, 8));
    hipError_t e = hipMemsetAsync(d_ovf, 0, 8, s);
    if (e == hipSuccess) {
        KernelTimer t("index_zero", s);
        e = hipMemsetAsync(set->d_lines, 0, n_lines * 64ull + n_lines / 8ull + 64ull, s);
    }
    if (e == hipSuccess && n) {
        KernelTimer t("index_insert", s);
        uint64_t blocks = (n + 255) / 256;
        if (blocks > 256 * 16)
            blocks = 256 * 16;
        if (no_bits(set))
            index_insert_kernel<true><<<(int)blocks, 256, 0, s>>>(d_keys, n, set->d_lines, 32u - (uint32_t)log_lines, (uint32_t)m,
                                                                  (uint32_t)(k - m + 1), k, d_ovf, (uint32_t *)(set->d_lines + n_lines * 8ull));
        else
            index_insert_kernel<false><<<(int)blocks, 256, 0, s>>>(d_keys, n, set->d_lines, 32u - (uint32_t)log_lines, (uint32_t)m,
                                                                   (uint32_t)(k - m + 1), k, d_ovf, (uint32_t *)(set->d_lines + n_lines * 8ull));
        e = hipGetLastError();
    }
    unsigned long long ovf = 0;
    if (e == hipSuccess)
        e = hipMemcpyAsync(&ovf, d_ovf, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    (void)hipFree(d_ovf);
    if (e != hipSuccess) {
        set_error("probe index build: %s", hipGetErrorString(e));
        return BRX_ERR_HIP;
    }
    set->idx_log_lines = (uint32_t)log_lines;
    set->idx_m = (uint32_t)m;
    set->idx_keys = n;
    set->idx_overflow_keys = ovf;
    set->idx_exact = no_bits(set);
    set->idx_open = false;
    set->idx_linebits = true; // this build filled the occupancy bits behind the lines
    set->idx_valid = true;
    set->idx_gen++;
    return BRX_OK;
}

int index_build_from_keys(brx_set *set, const uint64_t *d_keys, uint64_t n, int m, int log_lines, hipStream_t s)
{
    std::lock_guard<std::mutex> g(set->idx_mu);
    return index_build_locked(set, d_keys, n, m, log_lines, s);
}

// presence-only insertion into a sparse set (the table IS the set; kept at most ~half full, regrown by rehashing)
int index_insert_reads(brx_set *set, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n_reads, uint64_t total_bases,
                       hipStream_t s)
{
    const int k = set->k;
    if (!set->sparse || !(k & 1) || k < 5 || k > 31) {
        set_error("k-mer insertion into a sparse set needs an odd 5 <= k <= 31 (k=%d%s)", k, set->sparse ? "" : ", set is not sparse");
        return BRX_ERR_UNSUPPORTED;
    }
    BRX_TRY(use_device(set->device));
    std::lock_guard<std::mutex> g(set->idx_mu);
    if (set->idx_valid && !set->idx_open) {
        set_error("the set was built by counting; k-mers cannot be added to it");
        return BRX_ERR_UNSUPPORTED;
    }
    if (!set->idx_valid) { // the empty set
        set->idx_keys = 0;
        set->idx_overflow_keys = 0;
    }
    const uint64_t need = set->idx_keys + total_bases; // every k-mer of the batch could be new
    const bool have = set->idx_valid && set->d_lines;
    if (!have || (7ull << set->idx_log_lines) < 2ull * need) {
        int log_lines = 12;
        while (log_lines < 30 && (7ull << log_lines) < 3ull * need)
            log_lines++;
        if ((7ull << log_lines) < need + need / 4) {
            set_error("sparse set: %llu k-mers do not fit the largest table", (unsigned long long)need);
            return BRX_ERR_NOMEM;
        }
        int m = env_int("BRX_INDEX_M", 0);
        if (m <= 0)
            m = index_auto_m(k, need);
        if ((!(m & 1) && m != 16) || m < 3 || m > IDX_MAX_M || m > k - 1) {
            set_error("probe index: bad minimizer length %d for k=%d", m, k);
            return BRX_ERR_ARG;
        }
        uint64_t *nl = nullptr;
        hipError_t e = hipMalloc((void **)&nl, (64ull << log_lines) + ((1ull << log_lines) >> 3) + 64ull); // (+ room for occupancy bits)
        if (e != hipSuccess) {
            set_error("hipMalloc(%llu B sparse set table): %s", (unsigned long long)(64ull << log_lines), hipGetErrorString(e));
            return BRX_ERR_NOMEM;
        }
        BRX_HIP(hipMemsetAsync(nl, 0, 64ull << log_lines, s));
        if (have && set->idx_keys) {
            KernelTimer t("index_rehash", s);
            table_rehash_kernel<<<256 * 8, 256, 0, s>>>(set->d_lines, 1ull << set->idx_log_lines, k, nl, 32u - (uint32_t)log_lines,
                                                        (uint32_t)m, (uint32_t)(k - m + 1));
        }
        BRX_HIP(hipStreamSynchronize(s));
        if (set->d_lines)
            (void)hipFree(set->d_lines);
        set->d_lines = nl;
        set->lines_alloc = 1ull << log_lines;
        set->idx_log_lines = (uint32_t)log_lines;
        set->idx_m = (uint32_t)m;
    }
    unsigned long long *d_new = nullptr;
    BRX_HIP(hipMalloc((void **)&d_new, 8));
    hipError_t e = hipMemsetAsync(d_new, 0, 8, s);
    if (e == hipSuccess && n_reads) {
        KernelTimer t("index_insert_reads", s);
        int st = flat_presence_insert(d_bases, d_offsets, n_reads, total_bases, k, nullptr, set->d_lines, 32u - set->idx_log_lines,
                                      set->idx_m, d_new, s);
        if (st != BRX_OK) {
            (void)hipFree(d_new);
            return st;
        }
    }
    unsigned long long added = 0;
    if (e == hipSuccess)
        e = hipMemcpyAsync(&added, d_new, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    (void)hipFree(d_new);
    if (e != hipSuccess) {
        set_error("sparse set insert: %s", hipGetErrorString(e));
        return BRX_ERR_HIP;
    }
    set->idx_keys += added;
    set->idx_exact = true;
    set->idx_linebits = false; // filled k-mer by k-mer: no occupancy bits
    set->idx_open = true;
    set->idx_valid = true;
    set->idx_gen++;
    set->keylist_valid = false; // the table is the set now
    return BRX_OK;
}

int ensure_bits(const brx_set *cset, hipStream_t s, const char *what)
{
    brx_set *set = const_cast<brx_set *>(cset);
    if (set->sparse) {
        set_error("%s needs the bit vector; a sparse set (k=%d) has none", what, set->k);
        return BRX_ERR_UNSUPPORTED;
    }
    if (!set->bits_stale)
        return BRX_OK;
    BRX_TRY(use_device(set->device));
    std::lock_guard<std::mutex> g(set->idx_mu);
    if (!set->bits_stale)
        return BRX_OK;
    KernelTimer t("bits_materialise", s);
    BRX_HIP(hipMemsetAsync(set->d_bits, 0, set->nwords * 4, s));
    if (set->keylist_valid) {
        keys_to_bits_kernel<<<256 * 8, 256, 0, s>>>(set->d_keylist, set->d_keylist_n, set->d_bits, set->nwords * 32);
    } else if (set->idx_valid && set->idx_exact) {
        table_to_bits_kernel<<<256 * 16, 256, 0, s>>>(set->d_lines, 1ull << set->idx_log_lines, set->d_bits, set->nwords * 32);
    } else {
        set_error("%s: the set has neither bits, key list nor an exact index", what);
        return BRX_ERR_ARG;
    }
    BRX_HIP(hipGetLastError());
    BRX_HIP(hipStreamSynchronize(s));
    set->bits_stale = false;
    return BRX_OK;
}

int index_ensure(const brx_set *cset, hipStream_t s)
{
    brx_set *set = const_cast<brx_set *>(cset);
    if (!index_wanted(set->k) && !no_bits(set))
        return BRX_OK;
    // several chains may share the set (one per host thread): exactly one of them builds, the others wait
    // here and find the index valid -- a second build would wipe lines the first chain's kernels are reading
    std::lock_guard<std::mutex> g(set->idx_mu);
    if (set->idx_valid || set->idx_declined)
        return BRX_OK;
    BRX_TRY(use_device(set->device));
    uint64_t n = 0;
    bool listed = false;
    if (set->keylist_valid) {
        unsigned long long nl = 0;
        BRX_HIP(hipMemcpyAsync(&nl, set->d_keylist_n, 8, hipMemcpyDeviceToHost, s));
        BRX_HIP(hipStreamSynchronize(s));
        listed = nl <= set->keylist_cap; // a truncated list is useless
        n = nl;
    }
    if (!listed) {
        BRX_HIP(hipStreamSynchronize(s)); // the bits may still be in flight on `s`; popcount runs on the null stream
        BRX_TRY(brx_set_popcount(set, &n));
    }
    if (no_bits(set) && !listed) {
        set_error("set without bit vector and without a complete key list (%llu solid k-mers counted, room for %llu)", (unsigned long long)n,
                  (unsigned long long)set->keylist_cap);
        return BRX_ERR_NOMEM;
    }
    if (!no_bits(set) && set->k - index_auto_m(set->k, n) + 1 < 3 && env_int("BRX_INDEX_M", 0) <= 0) {
        // so many keys that a safe minimizer is (nearly) the k-mer itself: neighbours would not share lines
        set->idx_declined = true;
        return BRX_OK;
    }
    if (listed)
        return index_build_locked(set, set->d_keylist, n, 0, 0, s);
    uint64_t *d_keys = nullptr;
    BRX_HIP(hipMalloc((void **)&d_keys, (n ? n : 1) * 8));
    uint64_t got = 0;
    int st = brx_set_extract_keys_device(set, 0, set->nwords * 32, d_keys, n, &got, s);
    if (st == BRX_OK)
        st = index_build_locked(set, d_keys, got, 0, 0, s);
    (void)hipFree(d_keys);
    return st;
}

} // namespace brx

extern "C" {

int brx_set_index_build(brx_set_t *set, int m, int log2_lines, void *stream)
{
    if (!set)
        return BRX_ERR_ARG;
    BRX_TRY(use_device(set->device));
    BRX_HIP(hipStreamSynchronize((hipStream_t)stream));
    uint64_t n = 0;
    BRX_TRY(brx_set_popcount(set, &n));
    uint64_t *d_keys = nullptr;
    BRX_HIP(hipMalloc((void **)&d_keys, (n ? n : 1) * 8));
    uint64_t got = 0;
    int st = brx_set_extract_keys_device(set, 0, set->nwords * 32, d_keys, n, &got, stream);
    if (st == BRX_OK)
        st = index_build_from_keys(set, d_keys, got, m, log2_lines, (hipStream_t)stream);
    (void)hipFree(d_keys);
    return st;
}

int brx_set_index_build_from_keys_device(brx_set_t *set, const uint64_t *d_keys, uint64_t n, int m, int log2_lines, void *stream)
{
    if (!set || (!d_keys && n))
        return BRX_ERR_ARG;
    int st = index_build_from_keys(set, d_keys, n, m, log2_lines, (hipStream_t)stream);
    if (st == BRX_OK && no_bits(set) && d_keys != set->d_keylist)
        set->keylist_valid = false; // without a bit vector the list just given IS the set now (multi-GPU exchange)
    return st;
}

int brx_set_keylist_device(const brx_set_t *set, void **d_keys, uint64_t *n, void *stream)
{
    if (!set || !d_keys || !n)
        return BRX_ERR_ARG;
    *d_keys = nullptr;
    *n = 0;
    if (!set->keylist_valid)
        return BRX_OK;
    BRX_TRY(use_device(set->device));
    unsigned long long nl = 0;
    BRX_HIP(hipMemcpyAsync(&nl, set->d_keylist_n, 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
    BRX_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (nl > set->keylist_cap)
        return BRX_OK; // truncated list: treat as absent
    *d_keys = set->d_keylist;
    *n = nl;
    return BRX_OK;
}

// ---- fingerprint: the same three numbers for the same SET, whatever holds it ------------------------------------------
namespace {
__device__ __forceinline__ void fp_add(unsigned long long *out, unsigned long long n, unsigned long long s1, unsigned long long s2)
{
    // wave sums by DPP-free shuffles, then one atomic per wave and number
    for (int o = 32; o > 0; o >>= 1) {
        n += __shfl_down(n, o);
        s1 += __shfl_down(s1, o);
        s2 += __shfl_down(s2, o);
    }
    if ((threadIdx.x & 63) == 0 && n) {
        atomicAdd(out + 0, n);
        atomicAdd(out + 1, s1);
        atomicAdd(out + 2, s2);
    }
}
__global__ __launch_bounds__(256) void fp_list_kernel(const uint64_t *__restrict__ keys, const unsigned long long *__restrict__ n_dev, uint64_t cap,
                                                      unsigned long long *out)
{
    const uint64_t n = *n_dev < cap ? *n_dev : cap;
    unsigned long long c = 0, s1 = 0, s2 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        const unsigned long long h = keys[i];
        c++;
        s1 += h;
        s2 += h * h;
    }
    fp_add(out, c, s1, s2);
}
__global__ __launch_bounds__(256) void fp_table_kernel(const uint64_t *__restrict__ lines, uint64_t n_lines, unsigned long long *out)
{
    unsigned long long c = 0, s1 = 0, s2 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_lines * 8ull; i += (uint64_t)gridDim.x * 256u) {
        const unsigned long long v = (i & 7ull) == 7ull ? 0ull : lines[i]; // (slot 7 is the line's header)
        if (v) {
            const unsigned long long h = v - 1ull;
            c++;
            s1 += h;
            s2 += h * h;
        }
    }
    fp_add(out, c, s1, s2);
}
__global__ __launch_bounds__(256) void fp_bits_kernel(const uint32_t *__restrict__ bits, uint64_t nwords, unsigned long long *out)
{
    unsigned long long c = 0, s1 = 0, s2 = 0;
    for (uint64_t w = (uint64_t)blockIdx.x * 256u + threadIdx.x; w < nwords; w += (uint64_t)gridDim.x * 256u) {
        uint32_t x = bits[w];
        while (x) {
            const unsigned long long h = w * 32ull + (unsigned long long)(__ffs(x) - 1);
            x &= x - 1u;
            c++;
            s1 += h;
            s2 += h * h;
        }
    }
    fp_add(out, c, s1, s2);
}
} // namespace

int brx_set_fingerprint(const brx_set_t *set, uint64_t *out3, void *stream)
{
    if (!set || !out3)
        return BRX_ERR_ARG;
    BRX_TRY(use_device(set->device));
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *d = nullptr;
    BRX_HIP(hipMalloc((void **)&d, 24));
    hipError_t e = hipMemsetAsync(d, 0, 24, s);
    const char *what = "list";
    if (e == hipSuccess) {
        if (set->keylist_valid) { // (a truncated list is never left valid with a set that has nothing else)
            fp_list_kernel<<<2048, 256, 0, s>>>(set->d_keylist, set->d_keylist_n, set->keylist_cap, d);
        } else if (no_bits(set)) {
            // no bit vector (sparse, or lazy and not written): the chained table is exact and holds every key once
            if (!set->idx_valid || !set->d_lines) {
                (void)hipFree(d);
                set_error("fingerprint: the set has neither a bit vector, nor a key list, nor a table");
                return BRX_ERR_UNSUPPORTED;
            }
            what = "table";
            fp_table_kernel<<<4096, 256, 0, s>>>(set->d_lines, 1ull << set->idx_log_lines, d);
        } else {
            what = "bits";
            fp_bits_kernel<<<4096, 256, 0, s>>>(set->d_bits, set->nwords, d);
        }
        e = hipGetLastError();
    }
    unsigned long long h[3] = {0, 0, 0};
    if (e == hipSuccess)
        e = hipMemcpyAsync(h, d, 24, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) {
        set_error("fingerprint (%s): %s", what, hipGetErrorString(e));
        return BRX_ERR_HIP;
    }
    out3[0] = h[0];
    out3[1] = h[1];
    out3[2] = h[2];
    return BRX_OK;
}

int brx_set_index_drop(brx_set_t *set)
{
    if (!set)
        return BRX_ERR_ARG;
    std::lock_guard<std::mutex> g(set->idx_mu);
    set->idx_valid = false;
    set->idx_gen++;
    if (set->d_lines && use_device(set->device) == BRX_OK)
        (void)hipFree(set->d_lines);
    set->d_lines = nullptr;
    set->lines_alloc = 0;
    return BRX_OK;
}

int brx_set_index_info(const brx_set_t *set, uint64_t *info8)
{
    if (!set || !info8)
        return BRX_ERR_ARG;
    for (int i = 0; i < 8; i++)
        info8[i] = 0;
    info8[0] = set->idx_valid ? 1 : 0;
    info8[6] = (index_wanted(set->k) || no_bits(set)) ? 1 : 0;
    info8[7] = set->keylist_valid ? 1 : 0;
    if (set->idx_valid) {
        info8[1] = set->idx_m;
        info8[2] = set->idx_log_lines;
        info8[3] = set->idx_keys;
        info8[4] = set->idx_overflow_keys;
        info8[5] = (1ull << set->idx_log_lines) * 64ull;
    }
    return BRX_OK;
}

int brx_set_get_batch_indexed(const brx_set_t *set, const uint64_t *forward_kmers, uint32_t n, uint8_t *out, uint64_t *n_fallback)
{
    if (!set || (!forward_kmers && n) || (!out && n))
        return BRX_ERR_ARG;
    if (!set->idx_valid) {
        set_error("get_batch_indexed: the set has no probe index (brx_set_index_build)");
        return BRX_ERR_ARG;
    }
    BRX_TRY(use_device(set->device));
    if (n_fallback)
        *n_fallback = 0;
    if (!n)
        return BRX_OK;
    uint64_t *d_k = nullptr;
    uint8_t *d_o = nullptr;
    unsigned long long *d_f = nullptr;
    BRX_HIP(hipMalloc((void **)&d_k, (uint64_t)n * 8));
    hipError_t e = hipMalloc((void **)&d_o, n);
    if (e == hipSuccess)
        e = hipMalloc((void **)&d_f, 8);
    if (e == hipSuccess)
        e = hipMemset(d_f, 0, 8);
    if (e == hipSuccess)
        e = hipMemcpy(d_k, forward_kmers, (uint64_t)n * 8, hipMemcpyHostToDevice);
    unsigned long long fb = 0;
    if (e == hipSuccess) {
        IdxView v{set->d_lines, 32u - set->idx_log_lines, set->idx_m, (uint32_t)set->k - set->idx_m + 1u};
        index_get_kernel<<<(n + 255) / 256, 256>>>(v, no_bits(set) ? nullptr : set->d_bits, d_k, n, set->k, d_o, d_f);
        e = hipMemcpy(out, d_o, n, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            e = hipMemcpy(&fb, d_f, 8, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_k);
    if (d_o)
        (void)hipFree(d_o);
    if (d_f)
        (void)hipFree(d_f);
    if (e != hipSuccess) {
        set_error("get_batch_indexed: %s", hipGetErrorString(e));
        return BRX_ERR_HIP;
    }
    if (n_fallback)
        *n_fallback = fb;
    return BRX_OK;
}

} // extern "C"
