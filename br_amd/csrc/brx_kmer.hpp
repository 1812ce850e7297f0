// 2-bit k-mer codec shared by host and gfx950 device code.
// Restates cocktail::kmer (git f63f0ba, un-vendored) as used by the reference at
// src/correct/mod.rs:26-42,110-112 and src/set/pcon.rs:189-191; conventions pinned by
// tests/golden/raw.k11.a2.solid: base code (ascii>>1)&3 (A0 C1 T2 G3),
// complement = xor 0b10, canonical = even-popcount member of {kmer, revcomp},
// bitset index = canonical >> 1.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BRX_HD __host__ __device__ __forceinline__
#else
#define BRX_HD inline
#endif

namespace brx {

BRX_HD uint64_t nuc2bit(uint8_t c) { return (uint64_t)((c >> 1) & 3u); }

BRX_HD uint8_t bit2nuc(uint64_t b)
{
    // "ACTG"[b]: 0x41 0x43 0x54 0x47 packed little-endian
    return (uint8_t)((0x47544341u >> (8u * (unsigned)(b & 3u))) & 0xffu);
}

BRX_HD uint64_t kmask(int k) { return k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull); }

// correct/mod.rs:110-112
BRX_HD uint64_t add_nuc(uint64_t kmer, uint64_t nuc, uint64_t mask) { return ((kmer << 2) & mask) ^ nuc; }

// reverse the order of the 32 2-bit groups of x
BRX_HD uint64_t rev2(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t r = __brevll(x);
#else
    uint64_t r = x;
    r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
    r = ((r >> 2) & 0x3333333333333333ull) | ((r & 0x3333333333333333ull) << 2);
    r = ((r >> 4) & 0x0f0f0f0f0f0f0f0full) | ((r & 0x0f0f0f0f0f0f0f0full) << 4);
    r = __builtin_bswap64(r);
#endif
    // full bit reversal also swapped the two bits inside each group: swap them back
    return ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
}

BRX_HD uint64_t revcomp(uint64_t kmer, int k)
{
    return rev2(kmer ^ 0xAAAAAAAAAAAAAAAAull) >> (64 - 2 * k);
}

BRX_HD int popc64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

BRX_HD uint64_t canonical(uint64_t kmer, int k) { return (popc64(kmer) & 1) ? revcomp(kmer, k) : kmer; }

// index of a forward k-mer in the packed bitset / count table
BRX_HD uint64_t khash(uint64_t kmer, int k) { return canonical(kmer, k) >> 1; }

} // namespace brx
