// Probe index: an HBM-locality layout of the solid set, used by the correction kernels.
//
// The reference's KmerSet::get (src/set/pcon.rs:189-191) is one bit of a 2^(2k-1)-bit vector, and the
// canonical hashes of the k-mers a read walks through are unrelated: every get is its own 64-byte
// memory transaction, and MI355X sustains ~55 G such transactions per second whatever the kernel does
// (tools/line_probe_bench.hip).  The same tool shows that lanes of a wave reading the SAME line cost one
// transaction.  So the solid k-mers are also stored in 64-byte lines addressed by the k-mer's
// strand-symmetric MINIMIZER (smallest hashed canonical m-mer inside the k-mer): consecutive k-mers of a
// read share their minimizer for (k-m+2)/2 positions on average and the lanes that probe them share the
// line.  The index is exact: a line lists full keys; a line that could not take all its keys is flagged
// and a probe it cannot answer is re-issued against the bitset, which stays the source of truth -- or, for
// sparse sets (k >= 21, no bit vector), against the next line, where the build chained the key to.
//
// Line layout (8 x u64): [0..6] key+1 of up to 7 k-mers (0 = empty; key = canonical >> 1),
// [7] header: low 32 bits = number of insert attempts, bit 63 = overflowed.
#pragma once
#include "brx_kmer.hpp"

namespace brx {

constexpr int IDX_SLOTS = 7;
constexpr uint64_t IDX_OVERFLOW = 1ull << 63;
// bits 32..62 of a line's header: a 31-bit signature of the keys that did not fit THEIR HOME line (the line their
// minimizer addresses).  A probe that finds its home line flagged but its own signature bit clear is a definite
// "absent" -- without it every probe of a flagged line, present or not, costs its group a second round.
// (32-bit multiply of the key's low word, top 5 bits, 31 folded onto 30: five instructions in the probe instead of a
// 64-bit multiply and a modulo)
__host__ __device__ inline uint32_t idx_sig_index(uint64_t key)
{
    const uint32_t t = ((uint32_t)key * 0x9E3779B1u) >> 27;
    return t > 30u ? 30u : t;
}
__host__ __device__ inline uint64_t idx_sig_bit(uint64_t key) { return 1ull << (32u + idx_sig_index(key)); }
constexpr int IDX_MAX_M = 16; // a canonical m-mer must fit 32 bits (32-bit window arithmetic); 16 needs no masking
constexpr int IDX_AUTO_MAX_M = 15; // what index_auto_m picks by itself (odd lengths; BRX_INDEX_M / the ABI may ask for 16)

struct IdxView {
    const uint64_t *lines; // nullptr: no index, probe the bitset
    uint32_t line_shift;   // 32 - log2(number of lines)
    uint32_t m;            // minimizer length (odd, <= 15)
    uint32_t w;            // k - m + 1 windows
    // one bit per line: set = the line holds at least one key.  4 MiB for 2^25 lines, i.e. L2-resident: a probe whose
    // home line is empty is "absent" without fetching the line from HBM (79 % of the lines of the bench's set are empty,
    // and the reverse pass of run_correction probes almost only absent k-mers).  nullptr: not kept for this index.
    const uint32_t *line_bits = nullptr;
};

#if defined(__HIPCC__)

// hash of the minimizer of a k-mer: min over its W m-mers of (min(m-mer, revcomp(m-mer)) * odd) mod 2^32,
// an injective order of the canonical m-mers, so equal hashes mean equal minimizers.
// The m-mer at bit offset 2j of fwd is the revcomp of the m-mer at bit offset 2(W-1-j) of rc, so the
// value is the same for a k-mer and its reverse complement.
template <int W>
__device__ __forceinline__ uint32_t minimizer_hash_w(uint64_t fwd, uint64_t rc, uint32_t mm)
{
    const uint32_t flo = (uint32_t)fwd, fhi = (uint32_t)(fwd >> 32), rlo = (uint32_t)rc, rhi = (uint32_t)(rc >> 32);
    uint32_t best = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < W; j++) {
        const uint32_t f = __builtin_amdgcn_alignbit(fhi, flo, 2 * j) & mm;
        const uint32_t r = __builtin_amdgcn_alignbit(rhi, rlo, 2 * (W - 1 - j)) & mm;
        const uint32_t c = f < r ? f : r;
        const uint32_t h = c * 0x9E3779B1u; // (a full-rate xor+rotate scramble measured no faster: the kernel waits on memory)
        best = h < best ? h : best;
    }
    return best;
}

__device__ __forceinline__ uint32_t minimizer_of(uint64_t fwd, uint64_t rc, uint32_t m, uint32_t w);
// the same, also giving the minimum over all windows but the OLDEST one (the m-mer at the front of the k-mer): the k-mer
// that follows in a read shares exactly those W - 1 windows, so its minimizer is min(excl_oldest, hash of its one new
// m-mer) -- brx_onelane.hip asks about two consecutive positions per round that way
template <int W>
__device__ __forceinline__ uint32_t minimizer_hash_w2(uint64_t fwd, uint64_t rc, uint32_t mm, uint32_t &excl_oldest)
{
    const uint32_t flo = (uint32_t)fwd, fhi = (uint32_t)(fwd >> 32), rlo = (uint32_t)rc, rhi = (uint32_t)(rc >> 32);
    uint32_t best = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < W - 1; j++) {
        const uint32_t f = __builtin_amdgcn_alignbit(fhi, flo, 2 * j) & mm;
        const uint32_t r = __builtin_amdgcn_alignbit(rhi, rlo, 2 * (W - 1 - j)) & mm;
        const uint32_t c = f < r ? f : r;
        const uint32_t h = c * 0x9E3779B1u;
        best = h < best ? h : best;
    }
    excl_oldest = best;
    const uint32_t f = __builtin_amdgcn_alignbit(fhi, flo, 2 * (W - 1)) & mm;
    const uint32_t r = rlo & mm;
    const uint32_t c = f < r ? f : r;
    const uint32_t h = c * 0x9E3779B1u;
    return h < best ? h : best;
}

// false: more than 16 windows (no two-position rounds there)
__device__ __forceinline__ bool minimizer_pair(uint64_t fwd, uint64_t rc, uint32_t m, uint32_t w, uint32_t &best, uint32_t &excl_oldest)
{
    const uint32_t mm = m >= 16u ? 0xffffffffu : (1u << (2u * m)) - 1u;
    switch (w) { // wave-uniform
    case 2: best = minimizer_hash_w2<2>(fwd, rc, mm, excl_oldest); return true;
    case 3: best = minimizer_hash_w2<3>(fwd, rc, mm, excl_oldest); return true;
    case 4: best = minimizer_hash_w2<4>(fwd, rc, mm, excl_oldest); return true;
    case 5: best = minimizer_hash_w2<5>(fwd, rc, mm, excl_oldest); return true;
    case 6: best = minimizer_hash_w2<6>(fwd, rc, mm, excl_oldest); return true;
    case 7: best = minimizer_hash_w2<7>(fwd, rc, mm, excl_oldest); return true;
    case 8: best = minimizer_hash_w2<8>(fwd, rc, mm, excl_oldest); return true;
    case 9: best = minimizer_hash_w2<9>(fwd, rc, mm, excl_oldest); return true;
    case 10: best = minimizer_hash_w2<10>(fwd, rc, mm, excl_oldest); return true;
    case 11: best = minimizer_hash_w2<11>(fwd, rc, mm, excl_oldest); return true;
    case 12: best = minimizer_hash_w2<12>(fwd, rc, mm, excl_oldest); return true;
    case 13: best = minimizer_hash_w2<13>(fwd, rc, mm, excl_oldest); return true;
    case 14: best = minimizer_hash_w2<14>(fwd, rc, mm, excl_oldest); return true;
    case 15: best = minimizer_hash_w2<15>(fwd, rc, mm, excl_oldest); return true;
    case 16: best = minimizer_hash_w2<16>(fwd, rc, mm, excl_oldest); return true;
    default:
        best = minimizer_of(fwd, rc, m, w);
        excl_oldest = 0;
        return false;
    }
}

__device__ __forceinline__ uint32_t minimizer_of(uint64_t fwd, uint64_t rc, uint32_t m, uint32_t w)
{
    const uint32_t mm = m >= 16u ? 0xffffffffu : (1u << (2u * m)) - 1u;
    switch (w) { // wave-uniform
    case 2: return minimizer_hash_w<2>(fwd, rc, mm);
    case 3: return minimizer_hash_w<3>(fwd, rc, mm);
    case 4: return minimizer_hash_w<4>(fwd, rc, mm);
    case 5: return minimizer_hash_w<5>(fwd, rc, mm);
    case 6: return minimizer_hash_w<6>(fwd, rc, mm);
    case 7: return minimizer_hash_w<7>(fwd, rc, mm);
    case 8: return minimizer_hash_w<8>(fwd, rc, mm);
    case 9: return minimizer_hash_w<9>(fwd, rc, mm);
    case 10: return minimizer_hash_w<10>(fwd, rc, mm);
    case 11: return minimizer_hash_w<11>(fwd, rc, mm);
    case 12: return minimizer_hash_w<12>(fwd, rc, mm);
    case 13: return minimizer_hash_w<13>(fwd, rc, mm);
    case 14: return minimizer_hash_w<14>(fwd, rc, mm);
    case 15: return minimizer_hash_w<15>(fwd, rc, mm);
    case 16: return minimizer_hash_w<16>(fwd, rc, mm);
    default: { // more than 16 windows (k >= 31 with m = 15): 64-bit shifts
        uint32_t best = 0xffffffffu;
        for (uint32_t j = 0; j < w; j++) {
            const uint32_t f = (uint32_t)(fwd >> (2u * j)) & mm;
            const uint32_t r = (uint32_t)(rc >> (2u * (w - 1u - j))) & mm;
            const uint32_t c = f < r ? f : r;
            const uint32_t h = c * 0x9E3779B1u;
            best = h < best ? h : best;
        }
        return best;
    }
    }
}

// the minimum of W hashes crowds the low end of the range: multiply once more before taking the top bits
__device__ __forceinline__ uint32_t index_line_of(uint32_t mh, uint32_t line_shift)
{
    return (mh * 0x85EBCA6Bu) >> line_shift;
}

// 1 = present, 0 = absent, 2 = the line overflowed and does not hold the key: ask the bit vector, or -- sparse sets,
// whose keys chain into the following lines -- probe again with hop + 1
// the key of a forward k-mer (canonical >> 1, + 1) and its home line
__device__ __forceinline__ uint32_t index_locate(const IdxView &v, uint64_t fwd, int k, uint64_t &key)
{
    const uint64_t rc = revcomp(fwd, k);
    key = (((popc64(fwd) & 1) ? rc : fwd) >> 1) + 1ull;
    return index_line_of(minimizer_of(fwd, rc, v.m, v.w), v.line_shift);
}

__device__ __forceinline__ int index_probe_at(const IdxView &v, uint64_t key, uint32_t home, uint32_t hop)
{
    const uint32_t line = (home + hop) & (0xffffffffu >> v.line_shift);
    const ulonglong2 *L = reinterpret_cast<const ulonglong2 *>(v.lines + (uint64_t)line * 8ull);
    const ulonglong2 q0 = L[0], q1 = L[1], q2 = L[2], q3 = L[3];
    // `|`, not `||`: with short-circuit evaluation the compiler loads slot 0 first and fetches the rest of
    // the line only for the lanes that did not match it -- a second, dependent memory round trip
    const bool found = (q0.x == key) | (q0.y == key) | (q1.x == key) | (q1.y == key) | (q2.x == key) | (q2.y == key) | (q3.x == key);
    // home line (hop 0): go on only if one of the keys it turned away had this key's signature; further down a
    // chain only the flag can tell
    const uint32_t hdr_hi = (uint32_t)(q3.y >> 32); // bit 31 = overflowed, bits 0..30 = signature of the keys turned away
    const bool more = (hdr_hi >> 31) && (hop != 0u || ((hdr_hi >> idx_sig_index(key)) & 1u));
    return found ? 1 : (more ? 2 : 0);
}

__device__ __forceinline__ int index_probe(const IdxView &v, uint64_t fwd, int k, uint32_t hop = 0)
{
    uint64_t key;
    const uint32_t home = index_locate(v, fwd, k, key);
    return index_probe_at(v, key, home, hop);
}

// index_probe through the line-occupancy bits: an empty home line answers "absent" from a 4-byte load that hits L2
__device__ __forceinline__ int index_probe_filtered(const IdxView &v, uint64_t fwd, int k)
{
    uint64_t key;
    const uint32_t home = index_locate(v, fwd, k, key);
    if (!((v.line_bits[home >> 5] >> (home & 31u)) & 1u))
        return 0;
    return index_probe_at(v, key, home, 0u);
}
// KmerSet::get through the probe index with "cannot say" settled on the spot: the lane asks the bit vector when there is
// one, else -- sets without one chain their overflowed keys into the following lines -- the next lines of the chain.
// One probe in a thousand at configs[1]'s occupancy; the other lanes of the wave wait for it.
__device__ __forceinline__ bool index_get(const IdxView &v, const uint32_t *__restrict__ bits, uint64_t fwd, int k)
{
    uint64_t key;
    const uint32_t home = index_locate(v, fwd, k, key);
    if (v.line_bits && !((v.line_bits[home >> 5] >> (home & 31u)) & 1u))
        return false;
    int a = index_probe_at(v, key, home, 0u);
    if (a == 2) {
        if (bits) {
            const uint64_t h = khash(fwd, k);
            return (bits[h >> 5] >> (h & 31u)) & 1u;
        }
        const uint32_t last = 0xffffffffu >> v.line_shift;
        for (uint32_t hop = 1; a == 2 && hop <= last; hop++)
            a = index_probe_at(v, key, home, hop);
    }
    return a == 1;
}

// find-or-insert of one k-mer into a chained table (sparse sets filled k-mer by k-mer, `br large-kmer`): true if the
// key was not there.  Threads race for the first empty slot of a line with a CAS; slots never empty again, so
// all threads see the same first empty slot and a key cannot land twice.
__device__ __forceinline__ bool table_find_or_insert(uint64_t *lines, uint32_t line_shift, uint32_t m, uint32_t w, int k, uint64_t fwd)
{
    const uint64_t rc = revcomp(fwd, k);
    const unsigned long long key = (((popc64(fwd) & 1) ? rc : fwd) >> 1) + 1ull;
    const uint32_t line_mask = 0xffffffffu >> line_shift;
    uint32_t line = index_line_of(minimizer_of(fwd, rc, m, w), line_shift);
    bool first = true; // still at the key's home line
    for (;;) {
        unsigned long long *L = reinterpret_cast<unsigned long long *>(lines) + (uint64_t)line * 8ull;
        for (int j = 0; j < IDX_SLOTS; j++) {
            unsigned long long v = __hip_atomic_load(L + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == 0ull)
                v = atomicCAS(L + j, 0ull, key);
            if (v == 0ull)
                return true;
            if (v == key)
                return false;
        }
        {
            const unsigned long long want = (unsigned long long)IDX_OVERFLOW | (first ? (unsigned long long)idx_sig_bit(key) : 0ull);
            if ((__hip_atomic_load(L + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & want) != want)
                atomicOr(L + 7, want);
        }
        first = false;
        line = (line + 1u) & line_mask; // the table is kept at most half full: this ends
    }
}
#endif

} // namespace brx
