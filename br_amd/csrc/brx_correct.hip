// Per-read correction scan on gfx950.
//
// Reference: Corrector::correct (src/correct/mod.rs:53-107), Exist<S>::correct_error
// (src/correct/exist/mod.rs:112-150), ScenarioOne (src/correct/exist/one.rs:33-74),
// run_correction's per-record body (src/lib.rs:42-55).
//
// Execution model.  The reference scan is sequential inside a read (i, kmer, previous and the
// output length are loop carried) but every decision is a KmerSet::get, i.e. a random 1-bit
// probe into a 2^(2k-4)-byte table in HBM.  A read is walked by a GROUP of G lanes of one
// wavefront (G = 16/32/64, several reads per wave) that runs a small state machine; in every
// ROUND each lane of every group issues at most one probe, all probes of the wave go out in one
// global_load, and the group then reduces its G answers with a wave ballot:
//
//   SCAN   lanes probe the next G positions speculatively (lane l rolls kmer+bases[i..i+l]);
//          the ballot gives a G-bit solidity mask; the first `!solid && previous` bit is the
//          reference's error trigger; positions before it are accepted (copied to the output).
//   ALTS   4 lanes probe the alternatives of the last base       (alt_nucs, mod.rs:114-128)
//   SCEN   3*c lanes probe the c look-ahead k-mers of I/S/D      (get_score, exist/mod.rs:21-47)
//   MORE   <=3 lanes probe the tie-break k-mer                   (one_more, exist/mod.rs:49-70)
//
// The machine reproduces the reference's results exactly; only the order in which probes are
// issued differs (probes are pure).  Output goes to a per-read slot of a staging buffer
// (capacity = input length * (1 + slack/4) + 64); the reverse pass reads its input back to
// front instead of materialising a reversed copy (src/lib.rs:111 reverses, does not complement).
#include "brx_correct.hpp"
#include <chrono>

#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <type_traits>
#include <thread>
#include <vector>

using namespace brx;

namespace brx {
uint64_t scan_tmp_bytes(uint32_t n);
int exclusive_scan_lens(const uint32_t *d_lens, uint32_t n, uint64_t *d_tmp, uint64_t *d_out_offsets,
                        unsigned long long *d_total, hipStream_t s);
int upload_batch(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, uint8_t **d_bases, uint64_t *bases_cap,
                 uint64_t **d_off, uint64_t *off_cap, uint64_t *total, hipStream_t stream);
}

namespace {

enum { ST_INIT = 0, ST_SCAN, ST_ERRLEN, ST_ALTS, ST_SCEN, ST_MORE, ST_WALK, ST_T1, ST_TSCORE, ST_TMORE, ST_GFOLLOW, ST_GVALID };
enum { MODE_ONE = 0, MODE_GRAPH = 1, MODE_INSSUB = 2, MODE_TWO = 3 };

// ---- ScenarioTwo (src/correct/exist/two.rs:34-328), ids in declaration order ---------------------
enum { T_II, T_IS, T_SS, T_SD, T_DD, T_ICI, T_ICS, T_ICD, T_SCI, T_SCS, T_SCD, T_DCI, T_DCD, T_N };

// Everything ScenarioTwo::apply needs once the 16 stage-1 probes are back:
//   K      the corrected k-mer (exist/mod.rs:129)
//   s[j]   2-bit code of seq[j], j < 4 (seq = read from the trigger position on)
//   m1     16 solidity bits: family f = bit>>2 in {B: add(K,a), C: add(add(K,s2),a),
//          D: add(add(K,s1),a), E: add(add(K,s0),a)}, a = bit&3
struct TwoCtx {
    uint64_t K;
    uint32_t s; // s0 | s1<<2 | s2<<4 | s3<<6
    uint32_t m1;
};

__device__ __forceinline__ uint32_t two_s(const TwoCtx &t, int j) { return (t.s >> (2 * j)) & 3u; }
__device__ __forceinline__ uint32_t two_fam(const TwoCtx &t, int f) { return (t.m1 >> (4 * f)) & 0xfu; }

// ScenarioTwo::apply -> (kmer, offset); only meaningful for scenarios whose validity bit is set
__device__ __forceinline__ uint64_t two_apply(const TwoCtx &t, int sc, uint64_t mask, uint32_t &off)
{
    const uint64_t K = t.K;
    const uint64_t altB = (uint64_t)(__ffs(two_fam(t, 0)) - 1) & 3u, altC = (uint64_t)(__ffs(two_fam(t, 1)) - 1) & 3u;
    const uint64_t altD = (uint64_t)(__ffs(two_fam(t, 2)) - 1) & 3u, altE = (uint64_t)(__ffs(two_fam(t, 3)) - 1) & 3u;
    switch (sc) {
    case T_II: off = 3; return K;                                                    // two.rs:96
    case T_IS: off = 2; return K;                                                    // two.rs:97
    case T_SS: off = 2; return add_nuc(K, altB, mask);                               // two.rs:98-114
    case T_SD: off = 1; return add_nuc(K, altB, mask);                               // two.rs:115-126
    case T_DD: off = 0; return add_nuc(K, altB, mask);                               // two.rs:127-134
    case T_ICI: off = 4; return add_nuc(K, two_s(t, 3), mask);                       // two.rs:135-148
    case T_ICS: off = 3; return add_nuc(K, altB, mask);                              // two.rs:149-166
    case T_ICD: off = 3; return add_nuc(add_nuc(K, two_s(t, 2), mask), altC, mask);  // two.rs:167-181
    case T_SCI:
    case T_DCI: off = 4; return add_nuc(add_nuc(K, two_s(t, 1), mask), two_s(t, 3), mask); // two.rs:182-191,231-240
    case T_SCS: off = 3; return add_nuc(add_nuc(K, two_s(t, 1), mask), altD, mask);  // two.rs:192-215
    case T_SCD: off = 2; return add_nuc(add_nuc(K, two_s(t, 1), mask), altD, mask);  // two.rs:216-230
    default: off = 1; return add_nuc(add_nuc(K, two_s(t, 0), mask), altE, mask);     // DCD two.rs:241-254
    }
}

// which scenarios return Some(..) from apply AND pass get_score's `get(kmer)` test (exist/mod.rs:22-25)
__device__ __forceinline__ uint32_t two_valid(const TwoCtx &t, uint32_t rem)
{
    const uint32_t B = two_fam(t, 0), C = two_fam(t, 1), D = two_fam(t, 2), E = two_fam(t, 3);
    const bool nB = __popc(B) == 1, nC = __popc(C) == 1, nD = __popc(D) == 1, nE = __popc(E) == 1;
    const bool b_s1 = (B >> two_s(t, 1)) & 1u, b_s3 = (B >> two_s(t, 3)) & 1u;
    const bool d_s2 = (D >> two_s(t, 2)) & 1u, d_s3 = (D >> two_s(t, 3)) & 1u;
    uint32_t v = (1u << T_II) | (1u << T_IS);
    if (rem >= 2 && !b_s1 && nB) v |= 1u << T_SS;
    if (nB) v |= (1u << T_SD) | (1u << T_DD);
    if (rem >= 4 && b_s3) v |= 1u << T_ICI;
    if (rem >= 4 && !b_s1 && nB) v |= 1u << T_ICS;
    if (rem >= 4 && nC) v |= 1u << T_ICD;
    if (rem >= 4 && d_s3) v |= (1u << T_SCI) | (1u << T_DCI);
    if (rem >= 3 && b_s1 && !d_s2 && nD) v |= 1u << T_SCS;
    if (rem >= 2 && nD) v |= 1u << T_SCD;
    if (rem >= 2 && nE) v |= 1u << T_DCD;
    return v;
}

// ScenarioTwo::correct -> (bases packed 2 bits each, first base in the high bits; count; offset) two.rs:258-325
__device__ __forceinline__ uint32_t two_correct(const TwoCtx &t, int sc, uint64_t mask, uint32_t &nc, uint32_t &offc)
{
    uint32_t off;
    const uint64_t ck = two_apply(t, sc, mask, off);
    switch (sc) {
    case T_II:
    case T_IS: nc = 1; offc = 2; return (uint32_t)(t.K & 3u);
    case T_SS:
    case T_SD:
    case T_DD: nc = 2; offc = off; return (uint32_t)(ck & 0xfu);
    case T_ICI: nc = 1; offc = 3; return (uint32_t)(t.K & 3u);
    case T_ICD: nc = 2; offc = off - 1; return (uint32_t)(ck & 0xfu);
    case T_ICS: nc = 2; offc = off + 1; return (uint32_t)(ck & 0xfu);
    case T_SCI:
    case T_SCS:
    case T_SCD:
    case T_DCD: nc = 3; offc = off; return (uint32_t)(ck & 0x3fu);
    default: nc = 0; offc = 1; return 0; // DCI, two.rs:323
    }
}

// ---- bio 1.6.0 alignment::pairwise::Aligner::global for Greedy (greedy.rs:56-89) ------------------
// Affine gaps (open -1, extend -1), match +1 / mismatch -1, no clipping; `>`-only updates in the
// order match/subst, ins, del; traceback through per-layer back pointers; Ins consumes x, Del y.
// Restated from the crate's published algorithm (source not in the container; tie-breaks unpinned,
// SURVEY H3) -- the same restatement the CPU checker uses, laid out here as a systolic
// anti-diagonal sweep: lane l of the group owns DP row band*G+l+1, row values travel to the next
// lane by shuffle, the back-pointer matrix lives in LDS.
enum { TBV_START = 0, TBV_INS = 1, TBV_DEL = 2, TBV_SUBST = 3, TBV_MATCH = 4, TBV_XCLIP = 5 };
enum { OPV_MATCH = 0, OPV_SUBST = 1, OPV_DEL = 2, OPV_INS = 3 };
constexpr int BIO_MIN = -858993459;

struct GreedyLds {
    uint8_t *x, *y, *ops, *bT; // x: before+read, y: before+path, ops: traceback (reversed), bT: band boundary tb_s
    int *bS, *bI;              // band boundary scores
    uint16_t *tb;              // (m+1) x (n+1) back pointers: s | i<<3 | d<<6
};

__device__ __forceinline__ GreedyLds greedy_carve(uint8_t *base, uint32_t dim)
{
    // dim = max row/column count; all regions 16-byte aligned
    const uint32_t a = (dim + 15u) & ~15u;
    GreedyLds L;
    L.x = base;
    L.y = base + a;
    L.ops = base + 2 * a;          // 2*a bytes
    L.bT = base + 4 * a;
    L.bS = (int *)(base + 5 * a);  // 4*a bytes
    L.bI = (int *)(base + 9 * a);  // 4*a bytes
    L.tb = (uint16_t *)(base + 13 * a);
    return L;
}

__host__ __device__ inline uint32_t greedy_lds_bytes(uint32_t dim)
{
    const uint32_t a = (dim + 15u) & ~15u;
    return 13u * a + ((2u * dim * dim + 15u) & ~15u);
}

template <int G>
__device__ bool greedy_align(const GreedyLds &L, int gl, int m, int n, int nb, int &off_out)
{
    const int go = -1, ge = -1;
    const int W = m + 1;
    for (int band = 0; band * G < m; band++) {
        const int row = band * G + gl + 1;
        const bool arow = row <= m;
        const uint8_t pch = arow ? L.x[row - 1] : (uint8_t)0;
        int S_cur = go + ge * row, I_cur = S_cur, D_cur = BIO_MIN, S_prev1 = S_cur, tbs_cur = TBV_INS; // column 0
        const bool last_lane = (gl == G - 1);
        const bool more = (band + 1) * G < m;
        for (int t = 0; t < n + G - 1; t++) {
            int upS = __shfl_up(S_cur, 1, G), upI = __shfl_up(I_cur, 1, G);
            int upT = __shfl_up(tbs_cur, 1, G), upS1 = __shfl_up(S_prev1, 1, G);
            const int j = t - gl + 1;
            const bool act = arow && j >= 1 && j <= n;
            if (act) {
                if (gl == 0) {
                    if (band == 0) { // DP row 0
                        upS = go + ge * j;
                        upI = BIO_MIN;
                        upT = TBV_DEL;
                        upS1 = (j == 1) ? 0 : go + ge * (j - 1);
                    } else {
                        upS = L.bS[j];
                        upI = L.bI[j];
                        upT = L.bT[j];
                        upS1 = L.bS[j - 1];
                    }
                }
                const uint8_t q = L.y[j - 1];
                const int m_score = upS1 + (pch == q ? 1 : -1);
                int bestI, tbi, bestD, tbd;
                const int i_score = upI + ge, s_score = upS + go + ge;
                if (i_score > s_score) { bestI = i_score; tbi = TBV_INS; } else { bestI = s_score; tbi = upT; }
                const int d_score = D_cur + ge, s2 = S_cur + go + ge;
                if (d_score > s2) { bestD = d_score; tbd = TBV_DEL; } else { bestD = s2; tbd = tbs_cur; }
                int bestS = BIO_MIN, tbs = TBV_XCLIP;
                if (m_score > bestS) { bestS = m_score; tbs = (pch == q) ? TBV_MATCH : TBV_SUBST; }
                if (bestI > bestS) { bestS = bestI; tbs = TBV_INS; }
                if (bestD > bestS) { bestS = bestD; tbs = TBV_DEL; }
                S_prev1 = S_cur;
                S_cur = bestS;
                I_cur = bestI;
                D_cur = bestD;
                tbs_cur = tbs;
                L.tb[j * W + row] = (uint16_t)(tbs | (tbi << 3) | (tbd << 6));
                if (last_lane && more) {
                    L.bS[j] = bestS;
                    L.bI[j] = bestI;
                    L.bT[j] = (uint8_t)tbs;
                }
            }
        }
        if (last_lane && more)
            L.bS[0] = go + ge * row;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    // traceback from (m, n); every lane walks the same pointers (LDS broadcast reads)
    auto tbS = [&](int i, int j) -> int {
        if (i == 0) return j == 0 ? TBV_START : TBV_DEL;
        if (j == 0) return TBV_INS;
        return L.tb[j * W + i] & 7;
    };
    auto tbI = [&](int i, int j) -> int {
        if (j == 0) return i == 1 ? TBV_START : TBV_INS;
        return (L.tb[j * W + i] >> 3) & 7; // i >= 1 whenever the Ins layer is entered
    };
    auto tbD = [&](int i, int j) -> int {
        if (i == 0) return j == 1 ? TBV_START : TBV_DEL;
        return (L.tb[j * W + i] >> 6) & 7;
    };
    int i = m, j = n, nops = 0;
    int layer = tbS(i, j);
    while (nops < m + n + 2) {
        int next, op;
        if (layer == TBV_INS) { op = OPV_INS; next = tbI(i, j); i--; }
        else if (layer == TBV_DEL) { op = OPV_DEL; next = tbD(i, j); j--; }
        else if (layer == TBV_MATCH || layer == TBV_SUBST) { op = (layer == TBV_MATCH) ? OPV_MATCH : OPV_SUBST; next = tbS(i - 1, j - 1); i--; j--; }
        else break;
        if (gl == 0)
            L.ops[nops] = (uint8_t)op;
        nops++;
        layer = next;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // operations[before_seq.len()..].windows(2), greedy.rs:67-86 (ops are stored back to front)
    int offset = 0;
    for (int w = nb; w + 1 < nops; w++) {
        const int a = L.ops[nops - 1 - w], b = L.ops[nops - 2 - w];
        if (a == OPV_DEL) offset -= 1;
        else if (a == OPV_INS) offset += 1;
        if (a == OPV_MATCH && b == OPV_MATCH) {
            int oc = 0;
            for (int e = 0; e < nops; e++) {
                const int op = L.ops[e];
                if (op == OPV_DEL) oc -= 1;
                else if (op == OPV_INS) oc += 1;
                else break;
            }
            off_out = offset - oc;
            return true;
        }
    }
    return false;
}

// look-ahead k-mers per scenario probed in the SCEN round that starts at look-ahead index `sub`
// bit of a k-mer in the 32*G-bit visited filter of a walk
__device__ __forceinline__ uint32_t walk_filter_index(uint64_t kmer, int G)
{
    const uint32_t x = (uint32_t)(kmer ^ (kmer >> 23) ^ (kmer >> 41)) * 0x9E3779B1u;
    return x >> (32 - 5 - (G == 4 ? 2 : G == 8 ? 3 : G == 16 ? 4 : G == 32 ? 5 : 6));
}

// index of the n-th (0-based) set bit of a small mask
__device__ __forceinline__ uint32_t nth_bit(uint32_t mask, uint32_t n)
{
    for (uint32_t q = 0; q < n; q++)
        mask &= mask - 1u;
    return (uint32_t)__ffs(mask) - 1u;
}

// look-aheads per surviving scenario in a TSCORE round (Two): one while every valid scenario is still in, then as
// many as the group's lanes allow
__device__ __forceinline__ uint32_t two_width(uint32_t sub, uint32_t c, int G, uint32_t alive)
{
    const uint32_t left = c - sub;
    const uint32_t na = (uint32_t)__popc(alive);
    uint32_t cap = sub == 0u ? 1u : (uint32_t)G / (na ? na : 1u);
    if (cap < 1u)
        cap = 1u;
    return left < cap ? left : cap;
}

// (the lanes are dealt to the scenarios still alive, so once the wrong ones have died the survivor's
// remaining look-aheads fit one round even in an 8-lane group)
__device__ __forceinline__ uint32_t scen_width(uint32_t sub, uint32_t c, int G, uint32_t flags, uint32_t failmask)
{
    const uint32_t left = c - sub;
    const uint32_t alive = 3u - (uint32_t)__popc(failmask & 7u);
    // G / alive without a division (a variable u32 division is ~25 VALU instructions, paid by the whole wave)
    const uint32_t share = alive <= 1u ? (uint32_t)G : (alive == 2u ? (uint32_t)G / 2u : (uint32_t)G / 3u);
    const uint32_t cap = (sub == 0u && !(flags & 4u)) ? (share < 2u ? share : 2u) : share;
    return left < cap ? left : cap;
}

// SRC: where the work comes from -- 0 every read of the batch, 1 the reads of a list made on the device (p.only), 2 the
// triggers a lean reverse pass left open (p.trig; see rev_scan_kernel): the group enters at alt_nucs, writes nothing, and
// flags the read for a redo unless the method returns None
template <int G, int M, int SRC = 0>
// One fits 80 VGPRs without spilling: 6 waves per SIMD instead of 5 (the kernel waits on memory 60 % of the
// time, measured 8 % faster), and its 64-lane form (no group shuffles to keep) fits 72: 7 waves; the other methods
// keep the compiler's own choice
// Greedy: the compiler's own choice is 110 registers = 4 waves per SIMD; its LDS (the alignment table, ~2 KB a group)
// allows 5.  Held to 96 registers (20 bytes of scratch) it runs 5: 79.7 -> 69.1 ms per Gbp fwd+rev on the same box
// (profiles/r4m_greedy_ab.txt) -- a round of a group is a dependent trip to memory, and more groups hide it.
#ifndef BRX_GREEDY_WAVES
#define BRX_GREEDY_WAVES 5
#endif
// Two: the compiler's choice is 101 registers = 5 waves; held to 85 it runs 6: 65.2 -> 54.8 ms per Gbp fwd+rev (7 waves,
// 72 registers: 64.2 -- spills).  Graph's group kernel (92 registers, 5 waves) does not move at 6: 50.95 / 51.56.
#ifndef BRX_TWO_WAVES
#define BRX_TWO_WAVES 6
#endif
#ifndef BRX_GAPSIZE_WAVES
#define BRX_GAPSIZE_WAVES 5
#endif
#ifndef BRX_GRAPH_WAVES
#define BRX_GRAPH_WAVES 1 // (92 registers = 5 waves)
#endif
__global__ __launch_bounds__(256, (M == BRX_ONE ? (G == 64 ? 7 : 6)
                                                : (M == BRX_GAP_SIZE ? BRX_GAPSIZE_WAVES : (M == BRX_GREEDY ? BRX_GREEDY_WAVES : (M == BRX_TWO ? BRX_TWO_WAVES : BRX_GRAPH_WAVES))))) void correct_kernel(PassParams p)
{
    constexpr bool HAS_ERRLEN = (M == BRX_GRAPH || M == BRX_GAP_SIZE);
    constexpr bool HAS_ONE = (M == BRX_ONE || M == BRX_GAP_SIZE);
    constexpr bool HAS_WALK = HAS_ERRLEN;
    constexpr bool HAS_TWO = (M == BRX_TWO);
    constexpr bool HAS_GREEDY = (M == BRX_GREEDY);
    constexpr bool HAS_PATH = HAS_WALK || HAS_GREEDY;
    constexpr bool LIST = SRC == 1, VERIFY = SRC == 2;
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];

    const int lane = threadIdx.x & 63;
    const int gl = lane & (G - 1);
    const int gshift = lane & ~(G - 1);
    const uint64_t GM = (G == 64) ? ~0ull : ((1ull << (G & 63)) - 1ull);
    const int k = p.k;
    const uint32_t c = (uint32_t)p.c;
    const uint64_t mask = kmask(k);
    const uint32_t gid = blockIdx.x * (256 / G) + threadIdx.x / G; // global group index (path scratch slot)
    unsigned long long *path = HAS_PATH ? (unsigned long long *)p.path_k + (uint64_t)gid * p.maxpath : nullptr;
    GreedyLds gL = {};
    if (HAS_GREEDY)
        gL = greedy_carve(dyn_lds + (size_t)(threadIdx.x / G) * p.g_lds_bytes, p.g_dim);

    // group-uniform state (replicated in every lane of the group)
    uint32_t r = 0, n = 0, cap = 0, i = 0, olen = 0;
    const uint8_t *in = nullptr;
    uint8_t *out = nullptr;
    uint64_t kmer = 0, corr = 0;
    bool prev = false, have = false;
    int st = ST_INIT, mode = HAS_TWO ? MODE_TWO : (M == BRX_GRAPH ? MODE_GRAPH : MODE_ONE);
    uint32_t sub = 0, failmask = 0, passmask = 0;
    uint8_t ch_t = 0;
    uint32_t hop = 0;  // consecutive re-runs of the current round (sparse sets: which line of the chain is probed)
    bool slow = false; // this round is the bitset re-run of a round the probe index could not answer
    bool was_unres = false, kept_sol = false; // per lane: its probe of that round was unanswered / its answer
    // error_len / walk state
    uint32_t elen = 0, ej = 0, npath = 0, gap = 0;
    // Graph / GapSize: error_len (mod.rs:130-152) has probed the k-mers that end at i + 1 .. i + elen - 1 and found none
    // solid.  When the trigger then FAILS, the scan goes on over exactly those k-mers (the read's base stays in the
    // k-mer, mod.rs:91-96), so positions below skip_until are copied through without asking again.  In a reverse pass
    // -- where a trigger is a chance hit and error_len runs to the end of the read -- that is the rest of the read.
    uint32_t skip_until = 0;
    uint64_t fc = 0, ek = 0, wk = 0;
    // Brent cycle detector of the graph walk (see ST_WALK)
    uint64_t tort = 0;
    uint32_t bpow = 1, blam = 0;
    bool brent = false;
    uint32_t wfilt = 0; // this lane's 32 bits of the walk's visited-filter (exact-scan walks)
    TwoCtx tw = {0, 0, 0};
    uint32_t tvalid = 0;
    // greedy state: iteration, path length in bases, alignment offset
    uint32_t git = 0, gnl = 0, steps = 0;
    int goff = 0;
    // ... and its viewed set (greedy.rs:136): <= max_search + 1 k-mers, kept one per lane of the group when it has that
    // many lanes (a scan of the chain's list in memory was a dependent round trip per iteration)
    uint64_t gvis = 0;
    const bool greg = HAS_GREEDY && (uint32_t)p.max_search + 1u <= (uint32_t)G;
    // statistics
    // per-wave totals (wave-uniform, so they live in scalar registers instead of four VGPRs of a kernel that is at its
    // register limit): the lanes note their events of a round in `ev`, the end of the round ballots them
    uint32_t n_rounds = 0, n_probes = 0, n_trig = 0, n_fix = 0;
    enum { EV_ROUND = 1, EV_PROBE = 2, EV_TRIG = 4, EV_FIX = 8 };

    auto fetch = [&]() {
        for (;;) {
            unsigned long long w = 0;
            if (gl == 0)
                w = atomicAdd(p.ctrl + CTL_WORK, 1ull);
            w = __shfl(w, gshift);
            // LIST: the reads of a list made on the device (what the lane-per-chunk pass handed back)
            unsigned long long n_work = (unsigned long long)p.n_reads;
            if (LIST || VERIFY)
                n_work = *p.only_n;
            if (VERIFY && n_work > (unsigned long long)p.trig_cap)
                n_work = p.trig_cap; // (triggers beyond the buffer were not recorded: their reads were handed back whole)
            if (w >= n_work) {
                have = false;
                return;
            }
            have = true;
            TrigRec tr = {0, 0, 0, 0, 0};
            if (VERIFY)
                tr = p.trig[w];
            r = VERIFY ? tr.r : (LIST ? p.only[w] : (uint32_t)w);
            const uint64_t o0 = p.offsets[r], o1 = p.offsets[r + 1];
            if (p.in_staged) {
                in = p.in + slot_of(o0, r, p.slack);
                n = p.in_lens[r];
                if (n == 0xffffffffu) {
                    // an earlier pass of this attempt gave up on this read (slot or walk-list
                    // overflow): keep it poisoned; the host redoes the batch with more workspace
                    if (gl == 0)
                        p.out_lens[r] = 0xffffffffu;
                    continue;
                }
            } else {
                in = p.in + o0;
                n = (uint32_t)(o1 - o0);
            }
            const uint64_t s0 = slot_of(o0, r, p.slack), s1 = slot_of(o1, (uint64_t)r + 1, p.slack);
            out = p.out + s0;
            cap = (uint32_t)(s1 - s0);
            st = ST_INIT;
            i = 0;
            olen = 0;
            kmer = 0;
            prev = false;
            steps = 0;
            skip_until = 0;
            if (VERIFY) {
                // enter where the scan's trigger branch and error_len would have left the group (mod.rs:73-75 with
                // mod.rs:130-152 / gap_size.rs:97-108): the read is unchanged up to the trigger
                i = tr.i;
                olen = tr.i;
                uint64_t km = 0;
                for (int j = 0; j < k; j++)
                    km = (km << 2) | nuc2bit(in[p.flip ? (n - 1u - (tr.i + 1u - (uint32_t)k + (uint32_t)j)) : (tr.i + 1u - (uint32_t)k + (uint32_t)j)]);
                kmer = km;
                ch_t = in[p.flip ? (n - 1u - tr.i) : tr.i];
                elen = tr.elen;
                fc = tr.fc;
                if (HAS_TWO)
                    mode = MODE_TWO;
                else if (M == BRX_GRAPH)
                    mode = MODE_GRAPH;
                else if (M == BRX_GAP_SIZE) {
                    mode = elen < (uint32_t)k ? MODE_GRAPH : (elen == (uint32_t)k ? MODE_ONE : MODE_INSSUB);
                    gap = elen > (uint32_t)k ? elen - (uint32_t)k : 0u;
                } else
                    mode = MODE_ONE;
                st = ST_ALTS;
            }
            return;
        }
    };
    // logical base j of the current read (the reverse pass reads the buffer back to front)
    auto ld = [&](uint32_t j) -> uint8_t { return in[p.flip ? (n - 1u - j) : j]; };
    // A read is finished (or given up) at ONE place, the end of the round: every call site of an inlined fetch() redefines
    // the whole group state and costs a block of register copies at the join (one_kernel: 7 000 -> 4 900 lines of ISA)
    int done = 0; // 0 no, 1 finished, 16 + the CTL_* counter of the reason the read was given up
    bool flag_read = false; // VERIFY: this trigger does not end in None (or outgrew a workspace): the read is redone
    auto end_read = [&](int how) {
        if (VERIFY && how != 1) {
            flag_read = true;
            done = 1;
        } else {
            done = how;
        }
    };
    auto finish = [&]() { end_read(1); };
    auto overflow = [&]() { end_read(16 + CTL_OVERFLOW); };
    auto nonterminating = [&]() { end_read(16 + CTL_NONTERM); };
    auto path_overflow = [&]() { end_read(16 + CTL_PATHOVF); };
    // k-mer of this lane after appending the 2-bit codes of lanes 0..gl of the group to `carry`
    auto lane_kmer = [&](uint64_t carry, uint64_t code) -> uint64_t {
        uint64_t val = code;
#pragma unroll
        for (int d = 1; d < G && d < 32; d <<= 1) {
            const uint64_t other = __shfl_up(val, d, G);
            if (gl >= d)
                val |= other << (2 * d);
        }
        const int nb = gl + 1;
        return (nb >= k) ? (val & mask) : (((carry << (2 * nb)) | val) & mask);
    };

    // 2-bit codes of the first WB = min(G, 32) lanes of the group, lane 0's base in the top bits
    constexpr int WB = G < 32 ? G : 32;
    auto group_window = [&](uint64_t code) -> uint64_t {
        uint64_t val = code;
#pragma unroll
        for (int d = 1; d < WB; d <<= 1) {
            const uint64_t other = __shfl_up(val, d, G);
            if (gl >= d)
                val |= other << (2 * d);
        }
        return __shfl(val, gshift + WB - 1);
    };
    // km extended by the window bases b0 .. b0+nb-1 (b0 + nb <= WB, nb <= 31)
    auto ext = [&](uint64_t km, uint64_t W, uint32_t b0, uint32_t nb) -> uint64_t {
        const uint64_t bits = (W >> (2u * ((uint32_t)WB - b0 - nb))) & ((1ull << (2u * nb)) - 1ull);
        return ((km << (2u * nb)) | bits) & mask;
    };
    uint64_t win = 0; // bases seq[i .. i+WB) at the current trigger

    fetch();

    while (__any(have)) {
        if (G == 64) {
            // trigger-free rounds of the 64-lane form (every method's reverse pass is almost only these) in a loop of
            // their own, as in one_kernel: 64 positions, one probe each, ballot, accept; a round with a trigger is
            // left to the general code below, which redoes its (pure) probes
            while (have && st == ST_SCAN && !slow && n - i >= 65u && olen + 66u <= cap) {
                const uint8_t c8 = ld(i + (uint32_t)lane);
                const uint64_t km = lane_kmer64_dpp(kmer, (uint32_t)nuc2bit(c8), lane, mask);
                bool s1 = false;
                if (HAS_ERRLEN && i + (uint32_t)lane < skip_until) {
                    // (known not solid: error_len asked about it behind the trigger that failed)
                } else if (p.idx.lines) {
                    // (an overflowed index line is settled by the lane that met it, index_get: leaving the round to the
                    // general code and its re-run cost two full rounds of the group, ten times per 10 kb read)
                    s1 = index_get(p.idx, p.bits, km, k);
                } else {
                    s1 = probe(p.bits, km, k);
                }
                const uint64_t bs = __ballot(s1);
                const uint64_t trig = ~bs & ((bs << 1) | (prev ? 1ull : 0ull)); // mod.rs:73
                if (trig)
                    break;
                out[olen + (uint32_t)lane] = c8; // mod.rs:100
                olen += 64u;
                i += 64u;
                kmer = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(km >> 32), 63) << 32) |
                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)km, 63);
                prev = (bs >> 63) & 1ull; // mod.rs:99
                n_rounds += 1u;
                n_probes += 64u;
                steps++;
            }
            // error_len (mod.rs:130-152) in a loop of its own, too: a trigger of a REVERSE pass is a chance hit, and error_len
            // then runs to the end of the read -- 64 k-mers a round, none of them solid -- through the general round's whole
            // state dispatch (Graph's reverse pass spent 5.6 G vector instructions where One's spends 2.6 G).  Full blocks
            // without a solid k-mer stay here; the block that ends the search (a hit, the end of the read) is left to the
            // general code, which redoes its (pure) probes.
            while (HAS_ERRLEN && have && st == ST_ERRLEN && !slow && n - i > ej + 1u + 64u) {
                const uint8_t c8 = ld(i + ej + 1u + (uint32_t)lane);
                const uint64_t km = lane_kmer64_dpp(ek, (uint32_t)nuc2bit(c8), lane, mask);
                const bool s1 = p.idx.lines ? index_get(p.idx, p.bits, km, k) : probe(p.bits, km, k);
                if (__ballot(s1))
                    break;
                ek = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(km >> 32), 63) << 32) |
                     (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)km, 63);
                ej += 64u;
                n_rounds += 1u;
                n_probes += 64u;
                steps++;
            }
        }
        if (HAS_WALK) {
            // Walk steps (graph.rs:61-82, gap_size.rs:57-85) in a loop of their own, too: four probes, one successor, the
            // cycle detector's bookkeeping, the list -- a fifth of a general round's instructions.  A reverse pass's walks
            // follow the genome for hundreds to thousands of steps (the k-mer in front of a chance trigger is a genome
            // k-mer), each a dependent trip to memory, and the launch that verifies or redoes them lasts as long as the
            // walkers' steps take.  A step that ends the walk (no unique successor, a revisit, the target, the last step
            // of a fixed-length walk, a full list) changes nothing here and is left to the general round, which asks the
            // same (pure) probes again.  Narrow groups: the loop runs while at least as many groups of the wave walk as do
            // anything else, and for eight steps at most while any other group waits.
            constexpr uint64_t LEADERS = G == 64 ? 1ull : (G == 32 ? 0x0000000100000001ull : (G == 16 ? 0x0001000100010001ull : (G == 8 ? 0x0101010101010101ull : 0x1111111111111111ull)));
            for (uint32_t it = 0;; it++) {
                const bool wfast = have && st == ST_WALK && !slow && (brent || mode == MODE_INSSUB);
                const uint32_t nw = (uint32_t)__builtin_popcountll(__ballot(wfast) & LEADERS);
                const uint32_t no = (uint32_t)__builtin_popcountll(__ballot(have && !wfast) & LEADERS);
                if (nw == 0u || nw < no || (no != 0u && it >= 8u))
                    break;
                bool s1 = false, u1 = false;
                if (wfast && gl < 4) {
                    const uint64_t q = add_nuc(wk, (uint64_t)gl, mask); // next_nucs(kmer), mod.rs:118-128
                    if (p.idx.lines) {
                        const int pr = index_probe(p.idx, q, k);
                        s1 = pr == 1;
                        u1 = pr == 2;
                    } else {
                        s1 = probe(p.bits, q, k);
                    }
                }
                if (__ballot(u1))
                    break; // an overflowed index line: the general round and its re-run settle it
                const uint32_t am = (uint32_t)((__ballot(s1) >> gshift) & 0xfull);
                bool simple = wfast && __popc(am) == 1;
                uint64_t nk = 0;
                if (simple) {
                    nk = add_nuc(wk, (uint64_t)(__ffs(am) - 1), mask);
                    simple = !(brent && nk == tort) && !(!brent && npath >= p.maxpath) &&
                             !(mode == MODE_GRAPH ? nk == fc : gap <= 1u) && steps < (1u << 24) + 64u * n;
                }
                if (simple) {
                    if (brent && ++blam == bpow) {
                        tort = nk;
                        bpow *= 2u;
                        blam = 0;
                    }
                    if (gl == 0 && npath < p.maxpath)
                        __hip_atomic_store(path + npath, (unsigned long long)nk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (npath < 0xfffffff0u)
                        npath++;
                    wk = nk;
                    if (mode != MODE_GRAPH)
                        gap--;
                    steps++;
                }
                n_rounds += nw;
                n_probes += 4u * nw;
                if (__ballot(wfast && !simple))
                    break;
            }
        }
        uint32_t ev = 0;
        bool do_probe = false;
        uint64_t pk = 0;
        uint8_t ch = 0;
        uint32_t sc_s = 0; // scenario index of this lane in SCEN / TSCORE
        bool sc_active = false;

        // ---------------- phase 1: choose this round's probe --------------------------------
        if (have) {
            ev |= (gl == 0) ? EV_ROUND : 0u;
            if (st == ST_INIT) {
                if (n < (uint32_t)k) {
                    // mod.rs:56-58: shorter than k, returned verbatim (cap >= n + 64 always)
                    for (uint32_t j = gl; j < n; j += G)
                        out[j] = ld(j);
                    olen = n;
                } else {
                    uint64_t km = 0;
                    for (int j = 0; j < k; j++)
                        km = (km << 2) | nuc2bit(ld((uint32_t)j));
                    kmer = km;
                    for (uint32_t j = gl; j < (uint32_t)k; j += G)
                        out[j] = ld(j);
                    olen = (uint32_t)k;
                    i = (uint32_t)k;
                    do_probe = (gl == 0);
                    pk = kmer;
                }
            } else if (st == ST_SCAN) {
                const uint32_t pos = i + (uint32_t)gl;
                const bool valid = pos < n;
                ch = valid ? ld(pos) : (uint8_t)0;
                pk = lane_kmer(kmer, nuc2bit(ch));
                do_probe = valid && !(HAS_ERRLEN && pos < skip_until);
            } else if (HAS_ERRLEN && st == ST_ERRLEN) {
                // error_len, mod.rs:130-152: probe seq[i+1..] until the first solid k-mer
                const uint32_t rem = n - i;
                const uint32_t j = ej + 1u + (uint32_t)gl;
                const bool valid = j < rem;
                const uint8_t c2 = valid ? ld(i + j) : (uint8_t)0;
                pk = lane_kmer(ek, nuc2bit(c2));
                do_probe = valid;
            } else if (st == ST_ALTS) {
                // alt_nucs: the alternative equal to the read's own base IS the trigger k-mer, already
                // known to be non-solid -> three probes instead of four
                do_probe = gl < 4 && ((p.flags & 2u) || (uint64_t)gl != (kmer & 3ull));
                pk = add_nuc(kmer >> 2, (uint64_t)gl, mask);
                if ((HAS_ONE || HAS_TWO || HAS_GREEDY) && !(p.flags & 8u)) {
                    // the bases the scenarios will look at, fetched once (one coalesced load per group)
                    const uint32_t wp = i + (uint32_t)gl;
                    win = group_window((gl < WB && wp < n) ? nuc2bit(ld(wp)) : 0ull);
                }
            } else if (HAS_ONE && st == ST_SCEN) {
                // get_score look-ahead, staged: look-ahead k-mers [sub, sub+width) of the three
                // scenarios per round (first the two nearest ones: a wrong scenario almost always dies
                // there, so its remaining probes are never issued).  Same verdict as probing all c.
                const uint32_t width = scen_width(sub, c, G, p.flags, failmask);
                const uint32_t e = (uint32_t)gl;
                const uint32_t alive = ~failmask & 7u; // never 0 here
                // which surviving scenario this lane works for: e / width by two compares (at most three scenarios)
                const uint32_t ord = (e >= width ? 1u : 0u) + (e >= 2u * width ? 1u : 0u);
                sc_active = e < 3u * width && ord < (uint32_t)__popc(alive);
                const uint32_t a1 = alive & (alive - 1u);
                sc_s = !sc_active ? 0u : (ord == 0u ? (uint32_t)__ffs(alive) - 1u : (ord == 1u ? (uint32_t)__ffs(a1) - 1u : 2u));
                const uint32_t j = sub + (sc_active ? e - ord * width : 0u);
                const uint32_t off = 2u - sc_s; // I:2 S:1 D:0 (one.rs:57-63)
                if (sc_active) {
                    if (c + 3u <= (uint32_t)WB && !(p.flags & 8u)) {
                        pk = ext(corr, win, off, j + 1u);
                    } else {
                        pk = corr;
                        for (uint32_t q = 0; q <= j; q++)
                            pk = add_nuc(pk, nuc2bit(ld(i + off + q)), mask);
                    }
                    do_probe = true;
                }
            } else if (HAS_ONE && st == ST_MORE) {
                if (gl < 3 && ((passmask >> gl) & 1u)) {
                    const uint32_t off = 2u - (uint32_t)gl;
                    const uint32_t rem = n - i;
                    if (rem > c + off + 1u) { // exist/mod.rs:54
                        if (c + 3u <= (uint32_t)WB && !(p.flags & 8u)) {
                            pk = ext(corr, win, off, c + 1u);
                        } else {
                            pk = corr;
                            for (uint32_t q = 0; q <= c; q++)
                                pk = add_nuc(pk, nuc2bit(ld(i + off + q)), mask);
                        }
                        do_probe = true;
                    }
                }
            } else if (HAS_WALK && st == ST_WALK) {
                do_probe = gl < 4; // next_nucs(kmer), mod.rs:118-128
                pk = add_nuc(wk, (uint64_t)gl, mask);
            } else if (HAS_TWO && st == ST_T1) {
                // the 16 distinct probes behind every ScenarioTwo::apply (see TwoCtx)
                const uint32_t rem = n - i;
                const int fam = gl >> 2;
                if (gl < 16) {
                    uint64_t basek = corr;
                    bool ok = true;
                    if (fam == 1) { ok = rem >= 3; basek = add_nuc(corr, two_s(tw, 2), mask); }
                    else if (fam == 2) { ok = rem >= 2; basek = add_nuc(corr, two_s(tw, 1), mask); }
                    else if (fam == 3) { basek = add_nuc(corr, two_s(tw, 0), mask); }
                    pk = add_nuc(basek, (uint64_t)(gl & 3), mask);
                    do_probe = ok;
                }
            } else if (HAS_TWO && st == ST_TSCORE) {
                // get_score of the scenarios still alive, `sub` look-aheads done so far: first the nearest k-mer of
                // every valid scenario (13 lanes; most scenarios die there), then the lanes are dealt to the survivors
                const uint32_t alive = tvalid & ~failmask;
                const uint32_t width = two_width(sub, c, G, alive);
                const uint32_t ord = (uint32_t)gl / width;
                sc_active = ord < (uint32_t)__popc(alive);
                sc_s = sc_active ? nth_bit(alive, ord) : 0u;
                const uint32_t j = sub + (uint32_t)gl % width;
                if (sc_active) {
                    uint32_t off;
                    pk = two_apply(tw, (int)sc_s, mask, off);
                    // get_score look-ahead, exist/mod.rs:33-41: seq[i+off .. i+off+j], out of the window fetched at ALTS
                    if (off + j + 1u <= (uint32_t)WB && !(p.flags & 8u)) {
                        pk = ext(pk, win, off, j + 1u);
                    } else {
                        for (uint32_t q = 0; q <= j; q++)
                            pk = add_nuc(pk, nuc2bit(ld(i + off + q)), mask);
                    }
                    do_probe = true;
                }
            } else if (HAS_TWO && st == ST_TMORE) {
                if (gl < T_N && ((passmask >> gl) & 1u)) {
                    uint32_t nc, offc;
                    const uint32_t cb = two_correct(tw, gl, mask, nc, offc);
                    const uint32_t rem = n - i;
                    if (rem > c + offc + 1u) { // exist/mod.rs:54
                        pk = corr >> 2;
                        for (uint32_t q = 0; q < nc; q++)
                            pk = add_nuc(pk, (uint64_t)((cb >> (2 * (nc - 1 - q))) & 3u), mask);
                        if (offc + c + 1u <= (uint32_t)WB && !(p.flags & 8u)) {
                            pk = ext(pk, win, offc, c + 1u);
                        } else {
                            for (uint32_t q = 0; q <= c; q++)
                                pk = add_nuc(pk, nuc2bit(ld(i + offc + q)), mask);
                        }
                        do_probe = true;
                    }
                }
            } else if (HAS_GREEDY && st == ST_GFOLLOW) {
                do_probe = gl < 4; // follow_graph -> next_nucs, greedy.rs:91-102
                pk = add_nuc(wk, (uint64_t)gl, mask);
            } else if (HAS_GREEDY && st == ST_GVALID) {
                // check_next_kmers, greedy.rs:104-117 -- and, in the lanes behind its c probes, follow_graph of the NEXT
                // iteration (next_nucs of the same k-mer, greedy.rs:91-102): an iteration of greedy's loop nearly always
                // ends in "the look-ahead does not hold, go on", and then the four answers are already there -- one round
                // per iteration instead of two (7 of the ~15 rounds of a trigger)
                const uint32_t rem2 = n - i - git;
                const uint32_t e = sub * G + (uint32_t)gl;
                if (rem2 >= c && e < c) {
                    if (git + e + 1u <= (uint32_t)WB && !(p.flags & 8u)) {
                        pk = ext(wk, win, git, e + 1u);
                    } else {
                        pk = wk;
                        for (uint32_t q = 0; q <= e; q++)
                            pk = add_nuc(pk, nuc2bit(ld(i + git + q)), mask);
                    }
                    do_probe = true;
                } else if (c + 4u <= (uint32_t)G && (uint32_t)gl >= c && (uint32_t)gl < c + 4u) {
                    pk = add_nuc(wk, (uint64_t)((uint32_t)gl - c), mask);
                    do_probe = true;
                }
            }
            // runaway guard: no read needs anywhere near this many rounds (greedy can move the read cursor
            // backwards for ever on some inputs -- the reference spins; a logic error must not hang the GPU)
            if (++steps > (1u << 24) + 64u * n)
                do_probe = false;
        }

        // ---------------- phase 2: one probe per lane, whole wave at once ---------------------
        bool sol = false;
        bool retry = false;
        if (p.idx.lines) {
            // probe index (brx_index.hpp): the line of the k-mer's minimizer, shared by the neighbouring
            // lanes.  A line that overflowed at build time cannot say "absent": the GROUP then repeats
            // this round against the bitset (phase 1 is a pure function of the group state, so the next
            // iteration recomputes the same probes); the other groups of the wave carry on.
            bool unres = false;
            if (do_probe) {
                if (!slow) {
                    const int pr = index_probe(p.idx, pk, k);
                    sol = pr == 1;
                    unres = pr == 2;
                    ev |= EV_PROBE;
                } else if (was_unres) {
                    if (p.bits) {
                        sol = probe(p.bits, pk, k);
                    } else { // sparse set: the build chained the key into the hop-th line after its own
                        const int pr = index_probe(p.idx, pk, k, hop);
                        sol = pr == 1;
                        unres = pr == 2;
                    }
                    ev |= EV_PROBE;
                } else {
                    sol = kept_sol; // answered by the index in the round being repeated
                }
            }
            retry = ((__ballot(unres) >> gshift) & GM) != 0ull;
            hop = retry ? hop + 1u : 0u;
            slow = retry;
            was_unres = unres;
            kept_sol = sol;
        } else if (do_probe) {
            sol = probe(p.bits, pk, k);
            ev |= EV_PROBE;
        }
        const uint64_t ball = __ballot(sol);
        const uint64_t gmask = (ball >> gshift) & GM;

        // ---------------- phase 3: group-uniform transitions ----------------------------------
        if (have && !retry) {
            bool fail = false;   // correct_error returned None
            int apply_s = -1;    // One scenario to apply
            int apply_t = -1;    // Two scenario to apply
            bool apply_path = false;
            uint32_t path_offset = 0;
            // Greedy: what follows the four answers `am` of next_nucs(wk) (greedy.rs:147-168, up to match_alignement)
            auto greedy_after_follow = [&](uint32_t am) {
                if (__popc(am) == 1) { // follow_graph succeeded: extend the path
                    const uint64_t a = (uint64_t)(__ffs(am) - 1);
                    if (gl == 0)
                        gL.y[k - 1 + (int)gnl] = bit2nuc(a);
                    gnl++;
                    wk = add_nuc(wk, a, mask);
                }
                // greedy.rs:153-157 (a failed follow leaves kmer in the set -> None).  The viewed set holds at most
                // max_search + 1 k-mers: one per lane of the group when it has that many (`greg`), else the chain's list
                bool hit = false;
                if (greg) {
                    hit = (uint32_t)gl < npath && gvis == wk;
                } else {
                    for (uint32_t j = gl; j < npath; j += G)
                        hit |= __hip_atomic_load(path + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == wk;
                }
                const uint64_t anyhit = (__ballot(hit) >> gshift) & GM;
                const uint32_t rem = n - i;
                if (anyhit || rem < git) { // greedy.rs:159-161
                    fail = true;
                } else if (!greg && npath >= p.maxpath) {
                    path_overflow();
                } else {
                    if (greg) {
                        if ((uint32_t)gl == npath)
                            gvis = wk;
                    } else if (gl == 0) {
                        __hip_atomic_store(path + npath, (unsigned long long)wk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (gl == 0 && git >= 1u)
                        gL.x[k - 1 + (int)git - 1] = ld(i + git - 1u);
                    npath++;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    // greedy.rs:163-168 returns Some iff match_alignement finds an offset AND check_next_kmers
                    // holds.  Both are pure, so the cheap one goes first: the c look-ahead probes (GVALID), and the
                    // (k+i)^2 alignment only for the few candidates that pass them.
                    sub = 0;
                    st = ST_GVALID;
                }
            };
            if (st == ST_INIT) {
                if (n < (uint32_t)k) {
                    finish();
                } else {
                    prev = gmask & 1ull; // mod.rs:67
                    if (i >= n)
                        finish();
                    else
                        st = ST_SCAN;
                }
            } else if (st == ST_SCAN) {
                const uint32_t left = n - i;
                const uint32_t nvalid = left < (uint32_t)G ? left : (uint32_t)G;
                const uint64_t vmask = (nvalid >= 64) ? ~0ull : ((1ull << nvalid) - 1ull);
                const uint64_t prevmask = (gmask << 1) | (prev ? 1ull : 0ull);
                const uint64_t trig = ~gmask & prevmask & vmask; // mod.rs:73
                const uint32_t nacc = trig ? (uint32_t)__builtin_ctzll(trig) : nvalid;
                if (olen + nacc + 1u > cap) {
                    overflow();
                } else {
                    if ((uint32_t)gl < nacc)
                        out[olen + (uint32_t)gl] = ch; // mod.rs:100
                    olen += nacc;
                    i += nacc;
                    if (trig) {
                        kmer = __shfl(pk, gshift + (int)nacc);
                        ch_t = (uint8_t)__shfl((int)ch, gshift + (int)nacc);
                        ev |= (gl == 0) ? EV_TRIG : 0u;
                        if (HAS_ERRLEN) {
                            st = ST_ERRLEN;
                            ej = 0;
                            ek = kmer;
                        } else {
                            st = ST_ALTS;
                        }
                    } else {
                        kmer = __shfl(pk, gshift + (int)nacc - 1);
                        prev = (gmask >> (nacc - 1)) & 1ull; // mod.rs:99
                        if (i >= n)
                            finish();
                    }
                }
            } else if (HAS_ERRLEN && st == ST_ERRLEN) {
                const uint32_t rem = n - i;
                const uint32_t first = ej + 1u;
                const uint32_t left = rem > first ? rem - first : 0u;
                const uint32_t nvalid = left < (uint32_t)G ? left : (uint32_t)G;
                const uint64_t vmask = (nvalid >= 64) ? ~0ull : ((1ull << nvalid) - 1ull);
                const uint64_t smask = gmask & vmask;
                bool done = false, hit_end = false;
                if (smask) {
                    const int t = __builtin_ctzll(smask);
                    elen = first + (uint32_t)t;
                    fc = __shfl(pk, gshift + t);
                    done = true;
                } else if (nvalid < (uint32_t)G) {
                    elen = rem; // ran off the read: the last k-mer built is not solid
                    const uint64_t last = __shfl(pk, gshift + (nvalid ? (int)nvalid - 1 : 0));
                    fc = nvalid ? last : ek;
                    done = true;
                    hit_end = true;
                } else {
                    ek = __shfl(pk, gshift + G - 1);
                    ej += G;
                }
                if (done) {
                    if (M == BRX_GRAPH) {
                        mode = MODE_GRAPH;
                    } else { // gap_size.rs:97-108
                        if (elen < (uint32_t)k)
                            mode = MODE_GRAPH;
                        else if (elen == (uint32_t)k)
                            mode = MODE_ONE;
                        else {
                            mode = MODE_INSSUB;
                            gap = elen - (uint32_t)k;
                        }
                    }
                    // Graph with a non-solid target: every walk k-mer is solid, so `kmer ==
                    // first_correct_kmer` (graph.rs:79) can never hold and the walk can only end in
                    // None (branch, dead end or revisit) -- same outcome, no walk.
                    if (mode == MODE_GRAPH && hit_end)
                        fail = true;
                    else
                        st = ST_ALTS;
                }
            } else if (st == ST_ALTS) {
                const uint32_t am = (uint32_t)(gmask & 0xfull);
                if (__popc(am) != 1) {
                    fail = true; // exist/mod.rs:123-126, graph.rs:51-55, gap_size.rs:47-50
                } else {
                    corr = add_nuc(kmer >> 2, (uint64_t)(__ffs(am) - 1), mask);
                    const uint32_t rem = n - i;
                    if (HAS_ONE && mode == MODE_ONE) {
                        failmask = 0;
                        for (uint32_t s = 0; s < 3; s++)
                            if ((2u - s) + c > rem) // exist/mod.rs:27-29
                                failmask |= 1u << s;
                        if (failmask == 7u) {
                            fail = true;
                        } else if (c == 0u) {
                            passmask = 7u & ~failmask;
                            if (__popc(passmask) == 1)
                                apply_s = __ffs(passmask) - 1;
                            else
                                st = ST_MORE;
                        } else {
                            sub = 0;
                            st = ST_SCEN;
                        }
                    } else if (HAS_WALK && (mode == MODE_GRAPH || mode == MODE_INSSUB)) {
                        // graph.rs:57-59 / gap_size.rs:52-55: path = [alt], viewed = {corr}
                        if (gl == 0)
                            __hip_atomic_store(path, (unsigned long long)corr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        npath = 1;
                        wk = corr;
                        st = ST_WALK;
                        // Graph's viewed_kmer set (graph.rs:47,71-75) only decides WHEN a walk that has
                        // entered a cycle returns None: the walk is deterministic (unique successor), so
                        // once a k-mer repeats it can never reach first_correct_kmer any more (every k-mer
                        // of the cycle was visited before and was not it) -- any cycle detector gives the
                        // same result.  Brent's needs O(1) state instead of a scan of the visited list.
                        // Not used when corr == first_correct_kmer (corr is never compared on entry) nor
                        // for the fixed-length walk of GapSize (a late detection could run past its end).
                        brent = (mode == MODE_GRAPH) && (corr != fc);
                        {
                            const uint32_t f0 = walk_filter_index(corr, G); // viewed = {corr}
                            wfilt = ((uint32_t)gl == (f0 >> 5)) ? (1u << (f0 & 31u)) : 0u;
                        }
                        tort = corr;
                        bpow = 1;
                        blam = 0;
                    } else if (HAS_GREEDY) {
                        // greedy.rs:135-145: before = kmer2seq(kmer >> 2, k-1); path = [alt]; viewed = {corr}
                        const uint64_t pre = kmer >> 2;
                        for (int j = gl; j < k - 1; j += G) {
                            const uint8_t b = bit2nuc((pre >> (2 * (k - 2 - j))) & 3ull);
                            gL.x[j] = b;
                            gL.y[j] = b;
                        }
                        if (gl == 0) {
                            gL.y[k - 1] = bit2nuc(corr & 3ull);
                            if (greg)
                                gvis = corr;
                            else
                                __hip_atomic_store(path, (unsigned long long)corr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        gnl = 1;
                        npath = 1;
                        wk = corr;
                        git = 0;
                        st = ST_GFOLLOW;
                        // (next_nucs of the three alternatives asked beside alt_nucs -- 3 + 12 probes fill the group's
                        // sixteen lanes -- saves this round too; measured 69.8 against 68.9 ms per Gbp: nothing, not kept)
                    } else if (HAS_TWO) {
                        uint32_t s4 = 0;
                        for (uint32_t j = 0; j < 4 && j < rem; j++)
                            s4 |= (uint32_t)nuc2bit(ld(i + j)) << (2 * j);
                        tw.K = corr;
                        tw.s = s4;
                        tw.m1 = 0;
                        st = ST_T1;
                    }
                }
            } else if (HAS_ONE && st == ST_SCEN) {
                const bool bad = sc_active && !sol;
                const uint32_t w_used = scen_width(sub, c, G, p.flags, failmask); // as dealt in phase 1
#pragma unroll
                for (uint32_t s = 0; s < 3; s++) {
                    const uint64_t b = (__ballot(bad && sc_s == s) >> gshift) & GM;
                    if (b)
                        failmask |= 1u << s;
                }
                sub += w_used;
                if (failmask == 7u) {
                    fail = true; // exist/mod.rs:132-134
                } else if (sub >= c) {
                    passmask = 7u & ~failmask;
                    if (__popc(passmask) == 1)
                        apply_s = __ffs(passmask) - 1; // exist/mod.rs:135-137
                    else
                        st = ST_MORE;
                }
            } else if (HAS_ONE && st == ST_MORE) {
                const uint32_t keep = (uint32_t)(gmask & 7ull) & passmask;
                if (__popc(keep) == 1)
                    apply_s = __ffs(keep) - 1; // exist/mod.rs:143-144
                else
                    fail = true;
            } else if (HAS_WALK && st == ST_WALK) {
                const uint32_t am = (uint32_t)(gmask & 0xfull);
                if (__popc(am) != 1) {
                    fail = true; // graph.rs:64-67, gap_size.rs:60-68
                } else {
                    const uint64_t nk = add_nuc(wk, (uint64_t)(__ffs(am) - 1), mask);
                    bool revisit;
                    // GapSize's fixed-length walk (gap_size.rs:57-85): the walk is deterministic -- every k-mer is the unique
                    // solid successor of the one before -- so two equal k-mers drag everything behind them along, and a repeat
                    // ANYWHERE in the walk shows as "the LAST k-mer was seen before".  The viewed set is therefore asked once,
                    // when the gap is walked, instead of at every step: a reverse pass's rare long walks (thousands of steps
                    // along the genome) had saturated the filter below and scanned their own list at every step -- quadratic,
                    // 3 us a step, a 9 ms tail on the launch.  (A walk the reference would have cut short at its first revisit
                    // goes on here to the end of the gap or to its first branch: None either way.)
                    const bool deferred = !brent && mode == MODE_INSSUB;
                    if (deferred) {
                        revisit = false;
                    } else if (brent) {
                        blam++;
                        revisit = (nk == tort);
                        if (!revisit && blam == bpow) {
                            tort = nk;
                            bpow *= 2u;
                            blam = 0;
                        }
                    } else {
                        // viewed_kmer.contains(&kmer): graph.rs:71, gap_size.rs:75.  The visited list is scanned only
                        // when a 32*G-bit filter spread over the group's lanes says the k-mer MAY have been seen (a
                        // new k-mer nearly never does; always scanning costs GapSize 9 % more time, never scanning --
                        // which would be wrong -- 4 % less)
                        const uint32_t fidx = walk_filter_index(nk, G);
                        const bool mine = (uint32_t)gl == (fidx >> 5);
                        const bool maybe = ((__ballot(mine && ((wfilt >> (fidx & 31u)) & 1u)) >> gshift) & GM) != 0ull;
                        revisit = false;
                        if (maybe) {
                            bool hit = false;
                            const uint32_t stored = npath < p.maxpath ? npath : p.maxpath;
                            for (uint32_t j = gl; j < stored; j += G)
                                hit |= __hip_atomic_load(path + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nk;
                            revisit = ((__ballot(hit) >> gshift) & GM) != 0ull;
                        }
                        if (mine)
                            wfilt |= 1u << (fidx & 31u);
                    }
                    if (revisit) {
                        fail = true;
                    } else if (npath >= p.maxpath && !brent) {
                        path_overflow(); // the exact scan needs every visited k-mer
                    } else {
                        // the visited list doubles as the corrected path; beyond maxpath a (Brent) walk goes
                        // on unrecorded and only a SUCCESS needs the batch redone with a larger list
                        if (gl == 0 && npath < p.maxpath)
                            __hip_atomic_store(path + npath, (unsigned long long)nk, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                        if (npath < 0xfffffff0u)
                            npath++;
                        wk = nk;
                        if (mode == MODE_GRAPH) {
                            if (nk == fc) { // graph.rs:79-81
                                if (npath > p.maxpath) {
                                    path_overflow();
                                } else {
                                    apply_path = true;
                                    path_offset = elen + 1u; // graph.rs:84
                                }
                            }
                        } else if (--gap == 0u) {
                            // the deferred viewed-set check: nk against every k-mer of the walk before it (the list holds
                            // them all: a walk that outgrew it was stopped above)
                            bool hit = false;
                            for (uint32_t j = gl; j + 1u < npath; j += G)
                                hit |= __hip_atomic_load(path + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nk;
                            if (((__ballot(hit) >> gshift) & GM) != 0ull) {
                                fail = true; // gap_size.rs:75-81
                            } else {
                                apply_path = true;
                                path_offset = npath; // gap_size.rs:87-88
                            }
                        }
                    }
                }
            } else if (HAS_TWO && st == ST_T1) {
                const uint32_t rem = n - i;
                tw.m1 = (uint32_t)(gmask & 0xffffull);
                tvalid = two_valid(tw, rem);
                for (int sc = 0; sc < T_N; sc++) { // exist/mod.rs:27-29
                    uint32_t off;
                    (void)two_apply(tw, sc, mask, off);
                    if (off + c > rem)
                        tvalid &= ~(1u << sc);
                }
                if (tvalid == 0u) {
                    fail = true;
                } else if (c == 0u) {
                    passmask = tvalid;
                    if (__popc(passmask) == 1)
                        apply_t = __ffs(passmask) - 1;
                    else
                        st = ST_TMORE;
                } else {
                    failmask = 0;
                    sub = 0;
                    st = ST_TSCORE;
                }
            } else if (HAS_TWO && st == ST_TSCORE) {
                const bool bad = sc_active && !sol;
                const uint64_t badmask = (__ballot(bad) >> gshift) & GM;
                const uint32_t alive = tvalid & ~failmask; // as dealt in phase 1
                const uint32_t width = two_width(sub, c, G, alive);
                const uint64_t wmask = width >= 64u ? ~0ull : (1ull << width) - 1ull;
                uint32_t rest = alive;
                for (uint32_t ord = 0; rest; ord++, rest &= rest - 1u)
                    if ((badmask >> (ord * width)) & wmask)
                        failmask |= rest & (0u - rest); // lowest set bit = the ord-th surviving scenario
                sub += width;
                passmask = tvalid & ~failmask;
                if (passmask == 0u) {
                    fail = true;
                } else if (sub >= c) {
                    if (__popc(passmask) == 1)
                        apply_t = __ffs(passmask) - 1;
                    else
                        st = ST_TMORE;
                }
            } else if (HAS_TWO && st == ST_TMORE) {
                const uint32_t keep = (uint32_t)(gmask & ((1ull << T_N) - 1ull)) & passmask;
                if (__popc(keep) == 1)
                    apply_t = __ffs(keep) - 1;
                else
                    fail = true;
            } else if (HAS_GREEDY && st == ST_GFOLLOW) {
                greedy_after_follow((uint32_t)(gmask & 0xfull));
            } else if (HAS_GREEDY && st == ST_GVALID) {
                const uint32_t rem2 = n - i - git;
                bool ok = rem2 >= c, done = false;
                if (ok && c > 0u) {
                    const uint32_t leftc = c - sub * G;
                    const uint32_t nv = leftc < (uint32_t)G ? leftc : (uint32_t)G;
                    const uint64_t vm = (nv >= 64u) ? ~0ull : ((1ull << nv) - 1ull);
                    ok = (gmask & vm) == vm;
                    sub++;
                    done = ok && sub * G >= c;
                } else if (ok) {
                    done = true;
                }
                // the next iteration: its follow_graph was asked in this very round when the lanes allow (phase 1)
                auto next_iteration = [&]() {
                    st = ST_GFOLLOW;
                    if (++git >= (uint32_t)p.max_search)
                        fail = true;
                    else if (c + 4u <= (uint32_t)G)
                        greedy_after_follow((uint32_t)(gmask >> c) & 0xfu);
                };
                if (!ok) {
                    next_iteration();
                } else if (done && !greedy_align<G>(gL, gl, k - 1 + (int)git, k - 1 + (int)gnl, k - 1, goff)) {
                    next_iteration(); // the look-ahead held but no alignment offset: next iteration (greedy.rs:163)
                } else if (done) {
                    // greedy.rs:165-167: offset = (local_corr.len() as i64 + off) as usize, wrapping add
                    if (VERIFY) {
                        flag_read = true;
                        finish();
                    } else if (olen + gnl + 1u > cap) {
                        overflow();
                    } else {
                        for (uint32_t j = gl; j < gnl; j += G)
                            out[olen + j] = gL.y[k - 1 + (int)j];
                        olen += gnl;
                        kmer = wk;
                        prev = true;
                        const long long ni = (long long)i + (long long)gnl + (long long)goff;
                        i = (ni < 0 || ni > (long long)n) ? n : (uint32_t)ni;
                        ev |= (gl == 0) ? EV_FIX : 0u;
                        if (i >= n)
                            finish();
                        else
                            st = ST_SCAN;
                    }
                }
            }
            if (have && !done && steps > (1u << 24) + 64u * n) {
                fail = false;
                apply_s = apply_t = -1;
                apply_path = false;
                nonterminating();
            }

            if (VERIFY && (fail || apply_s >= 0 || apply_t >= 0 || apply_path)) {
                flag_read |= !fail; // None is what the lean scan assumed
                finish();
            } else if (fail) {
                // mod.rs:91-96
                if (olen + 2u > cap) {
                    overflow();
                } else {
                    if (gl == 0)
                        out[olen] = ch_t;
                    olen += 1;
                    if (HAS_ERRLEN)
                        skip_until = i + elen; // (see its declaration)
                    i += 1;
                    prev = false;
                    if (i >= n) {
                        finish();
                    } else if (HAS_ERRLEN && skip_until >= n) {
                        // error_len ran off the read: no k-mer behind the trigger is solid, so no later one can trigger
                        // (`previous` stays false) and the scan copies the rest of the read through (mod.rs:99-102) -- here
                        // in one go, every lane's loads independent, instead of 64 bases a dependent round.  This is how
                        // nearly every read of a reverse pass ends.
                        const uint32_t left = n - i;
                        if (olen + left + 1u > cap) {
                            overflow();
                        } else {
                            for (uint32_t j = (uint32_t)gl; j < left; j += (uint32_t)G)
                                out[olen + j] = ld(i + j);
                            olen += left;
                            i = n;
                            finish();
                        }
                    } else {
                        st = ST_SCAN;
                    }
                }
            } else if (apply_s >= 0) {
                // mod.rs:75-89 with one.rs:65-71
                if (olen + 2u > cap) {
                    overflow();
                } else {
                    if (gl == 0)
                        out[olen] = bit2nuc(corr & 3ull);
                    olen += 1;
                    kmer = corr;
                    prev = true;
                    i += 2u - (uint32_t)apply_s;
                    ev |= (gl == 0) ? EV_FIX : 0u;
                    // The c look-ahead k-mers of the winning scenario ARE the next c scan k-mers and were
                    // all found solid: the reference's loop would copy these c bases with previous = true
                    // (mod.rs:99-102).  Accept them without probing again.
                    if (c > 0u && c + 3u <= (uint32_t)WB && olen + c + 1u <= cap && !(p.flags & 9u)) {
                        const uint32_t off = 2u - (uint32_t)apply_s;
                        if ((uint32_t)gl < c)
                            out[olen + (uint32_t)gl] = ld(i + (uint32_t)gl);
                        kmer = ext(kmer, win, off, c); // window positions are relative to the trigger
                        olen += c;
                        i += c;
                    }
                    if (i >= n)
                        finish();
                    else
                        st = ST_SCAN;
                }
            } else if (HAS_TWO && apply_t >= 0) {
                // mod.rs:75-89 with two.rs:258-325
                uint32_t nc, offc;
                const uint32_t cb = two_correct(tw, apply_t, mask, nc, offc);
                if (olen + nc + 1u > cap) {
                    overflow();
                } else {
                    uint64_t km = kmer >> 2;
                    for (uint32_t q = 0; q < nc; q++) {
                        const uint64_t b = (cb >> (2 * (nc - 1 - q))) & 3u;
                        km = add_nuc(km, b, mask);
                        if (gl == 0)
                            out[olen + q] = bit2nuc(b);
                    }
                    olen += nc;
                    kmer = km;
                    prev = true;
                    i += offc;
                    ev |= (gl == 0) ? EV_FIX : 0u;
                    if (i >= n)
                        finish();
                    else
                        st = ST_SCAN;
                }
            } else if (HAS_WALK && apply_path) {
                // mod.rs:75-89: the whole walked path replaces `path_offset` read bases
                if (olen + npath + 1u > cap) {
                    overflow();
                } else {
                    for (uint32_t j = gl; j < npath; j += G)
                        out[olen + j] = bit2nuc(__hip_atomic_load(path + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 3ull);
                    olen += npath;
                    kmer = wk;
                    prev = true;
                    i += path_offset;
                    ev |= (gl == 0) ? EV_FIX : 0u;
                    if (i >= n)
                        finish();
                    else
                        st = ST_SCAN;
                }
            }
        }
        if (done) {
            if (VERIFY) {
                if (flag_read && gl == 0 && atomicExch(p.redo_flag + r, 1u) == 0u) {
                    p.redo_list[atomicAdd(p.ctrl + CTL_REV_HANDBACK, 1ull)] = r;
                    atomicAdd(p.ctrl + CTL_REV_HANDBACK_SUM, 1ull);
                }
                flag_read = false;
            } else if (gl == 0) {
                p.out_lens[r] = done == 1 ? olen : 0xffffffffu;
                if (done != 1)
                    atomicAdd(p.ctrl + (done - 16), 1ull);
            }
            done = 0;
            fetch();
        }
        // end of the round, every lane back together: count the events
        n_rounds += (uint32_t)__builtin_popcountll(__ballot(ev & EV_ROUND));
        n_probes += (uint32_t)__builtin_popcountll(__ballot(ev & EV_PROBE));
        n_trig += (uint32_t)__builtin_popcountll(__ballot(ev & EV_TRIG));
        n_fix += (uint32_t)__builtin_popcountll(__ballot(ev & EV_FIX));
    }

    // statistics: one atomic per wave per counter
    if (lane == 0) {
        if (n_rounds)
            atomicAdd(p.ctrl + CTL_ROUNDS, (unsigned long long)n_rounds);
        if (n_probes)
            atomicAdd(p.ctrl + CTL_PROBES, (unsigned long long)n_probes);
        if (n_trig)
            atomicAdd(p.ctrl + CTL_TRIGGERS, (unsigned long long)n_trig);
        if (n_fix)
            atomicAdd(p.ctrl + CTL_FIXES, (unsigned long long)n_fix);
    }
}

// ======================================================================================================================
// correct::One, second form (the metric's kernel).  Same rounds, same probes, same bytes out as correct_kernel<G, BRX_ONE>;
// what changed is how many instructions a round costs.  rocprofv3's SQ counters showed the first form issue-bound, not
// memory-bound (VALU busy ~75 % of every SIMD's cycles, 27 of 64 lanes active: profiles/r1k_sq_summary.json), and
// tools/valu_rate.hip showed every VALU instruction costs the same issue slot while a ds_bpermute costs six:
//   - the lanes' k-mers come from a DPP row_shr scan of 2-bit codes in 32-bit registers (2 instructions per step, no
//     LDS crossbar) instead of a 64-bit __shfl_up scan (8 per step);
//   - the group's next k-mer after a SCAN round is cut from the group's code window (one 32-bit broadcast issued before
//     the probe's memory wait) instead of two 64-bit shuffles after it;
//   - k is a template parameter for the k's that matter (19, 21), which turns the variable 64-bit shifts into constants;
//   - event counters are one packed register, flushed per wave every 256 rounds.
// G = 8 / 16: groups inside one DPP row.  G = 64: one group per wave, group state in scalar registers.
// ======================================================================================================================
__device__ __forceinline__ uint32_t row_scan4(uint32_t code)
{
    uint32_t v = code;
    v |= dpp_row_shr<1>(v) << 2;
    v |= dpp_row_shr<2>(v) << 4;
    return v; // lanes of the row's later groups also carry codes of the group before: the caller keeps 2 * (gl + 1) bits
}
__device__ __forceinline__ uint32_t row_scan8(uint32_t code)
{
    uint32_t v = code;
    v |= dpp_row_shr<1>(v) << 2;
    v |= dpp_row_shr<2>(v) << 4;
    v |= dpp_row_shr<4>(v) << 8;
    return v; // lanes 8..15 of a row also carry codes of lanes 0..7: the caller keeps 2 * (gl + 1) bits
}

template <int G, int KT, bool LIST = false>
#ifndef BRX_ONE64_WAVES
#define BRX_ONE64_WAVES 7 // (the 64-lane form, every One chain's reverse pass; tools/ab_build.sh sweeps it)
#endif
__global__ __launch_bounds__(256, (G == 64 ? BRX_ONE64_WAVES : 7)) void one_kernel(PassParams p)
{
    static_assert(G == 4 || G == 8 || G == 16 || G == 64, "one_kernel: 4, 8, 16 or 64 lanes per read");
    const int lane0 = threadIdx.x & 63;
    int lane = lane0, gl = lane0 & (G - 1), gshift = lane0 & ~(G - 1); // re-derived every round from an opaque copy (see the loop)
    constexpr uint32_t GM32 = G >= 32 ? 0xffffffffu : ((1u << (G & 31)) - 1u);
    const int k = KT ? KT : p.k;
    const uint32_t c = (uint32_t)p.c;
    const uint64_t mask = kmask(k);
    uint32_t bcast_addr = (uint32_t)(gshift + G - 1) * 4u; // ds_bpermute address of the group's last lane
    uint32_t nb2 = 2u * ((uint32_t)gl + 1u);               // bits of this lane's scan value
    constexpr int WB = G == 4 ? 16 : (G < 32 ? G : 32);          // bases in the look-ahead window

    // group-uniform state
    uint32_t r = 0, n = 0, cap = 0, i = 0, olen = 0;
    const uint8_t *in = nullptr;
    uint8_t *out = nullptr;
    uint64_t kmer = 0, corr = 0;
    bool prev = false, have = false;
    int st = ST_INIT;
    uint32_t sub = 0, failmask = 0, passmask = 0;
    uint32_t hop = 0;
    bool slow = false, was_unres = false, kept_sol = false;
    typename std::conditional<(G < 32), uint32_t, uint64_t>::type win = 0; // seq[i .. i+WB) at the trigger, 2 bits per base
    uint32_t steps = 0;
    // events of this lane since the last flush: rounds | probes << 8 | triggers << 16 | fixes << 24 (flushed every
    // 255 rounds at most, so no field overflows)
    uint32_t ev = 0, since_flush = 0;
    // the base this lane probes in the next SCAN round, loaded at the end of the round before (G < 64): a round is a
    // dependent chain base load -> k-mer -> probe -> transition, and this takes the first link out of it
    uint8_t pf_ch = 0;

    auto flush = [&]() {
        // per-field sums over the wave, then one atomic per counter: every 255 rounds, so it is noise (and keeps the
        // totals out of the register file of a kernel that sits at its 80-VGPR limit)
        uint32_t a = ev & 0xffu, b = (ev >> 8) & 0xffu, cc = (ev >> 16) & 0xffu, d = ev >> 24;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o);
            b += __shfl_xor(b, o);
            cc += __shfl_xor(cc, o);
            d += __shfl_xor(d, o);
        }
        if (lane == 0) {
            if (a)
                atomicAdd(p.ctrl + CTL_ROUNDS, (unsigned long long)a);
            if (b)
                atomicAdd(p.ctrl + CTL_PROBES, (unsigned long long)b);
            if (cc)
                atomicAdd(p.ctrl + CTL_TRIGGERS, (unsigned long long)cc);
            if (d)
                atomicAdd(p.ctrl + CTL_FIXES, (unsigned long long)d);
        }
        ev = 0;
        since_flush = 0;
    };

    auto fetch = [&]() {
        for (;;) {
            unsigned long long w = 0;
            if (gl == 0)
                w = atomicAdd(p.ctrl + CTL_WORK, 1ull);
            w = __shfl(w, gshift);
            // LIST: the reads of a list made on the device (the few reads the lane-per-chunk pass handed back)
            if (w >= (LIST ? *p.only_n : (unsigned long long)p.n_reads)) {
                have = false;
                return;
            }
            have = true;
            r = LIST ? p.only[w] : (uint32_t)w;
            const uint64_t o0 = p.offsets[r], o1 = p.offsets[r + 1];
            if (p.in_staged) {
                in = p.in + slot_of(o0, r, p.slack);
                n = p.in_lens[r];
                if (n == 0xffffffffu) { // poisoned by an earlier pass of this attempt: stays poisoned
                    if (gl == 0)
                        p.out_lens[r] = 0xffffffffu;
                    continue;
                }
            } else {
                in = p.in + o0;
                n = (uint32_t)(o1 - o0);
            }
            const uint64_t s0 = slot_of(o0, r, p.slack), s1 = slot_of(o1, (uint64_t)r + 1, p.slack);
            out = p.out + s0;
            cap = (uint32_t)(s1 - s0);
            st = ST_INIT;
            i = 0;
            olen = 0;
            kmer = 0;
            prev = false;
            steps = 0;
            return;
        }
    };
    auto ld = [&](uint32_t j) -> uint8_t { return in[p.flip ? (n - 1u - j) : j]; };
    // The read is finished (or given up) at ONE place, the end of the round: every call site of an inlined fetch()
    // redefines the whole group state, and the compiler pays for each with a block of register copies at the join.
    int done = 0; // 0 no, 1 finished, 16 + the CTL_* counter of the reason the read was given up
    auto finish = [&]() { done = 1; };
    auto give_up = [&](int which) { done = 16 + which; };
    // this lane's k-mer: `carry` extended by the codes of lanes 0..gl of the group (scan = those codes, 32 bits)
    auto lane_kmer = [&](uint64_t carry, uint32_t code, uint32_t &scan_out) -> uint64_t {
        if (G == 4) {
            const uint32_t v = row_scan4(code) & ((1u << nb2) - 1u);
            scan_out = v;
            return ((carry << nb2) | v) & mask;
        } else if (G == 8) {
            const uint32_t v = row_scan8(code) & ((1u << nb2) - 1u);
            scan_out = v;
            return ((carry << nb2) | v) & mask;
        } else if (G == 16) {
            const uint32_t v = row_scan16(code);
            scan_out = v;
            return ((carry << nb2) | v) & mask; // nb2 <= 32
        } else {
            // 64 lanes = 4 rows: own row's codes + the 16-base words of the two rows before (row 0 / 1: of `carry`)
            scan_out = 0;
            return lane_kmer64_dpp(carry, code, lane, mask);
        }
    };
    // km extended by window bases b0 .. b0+nb-1 (b0 + nb <= WB; the window holds seq[i .. i+WB) at the trigger)
    auto ext = [&](uint64_t km, uint32_t b0, uint32_t nb) -> uint64_t {
        const uint64_t bits = (uint64_t)((win >> (2u * ((uint32_t)WB - b0 - nb))) & (decltype(win))((1ull << (2u * nb)) - 1ull));
        return ((km << (2u * nb)) | bits) & mask;
    };

    fetch();

    while (__any(have)) {
        // The compiler hoists every lane constant out of this loop (2*(gl+1), gl-3, (gl==0)<<16, ...) and then, at the
        // 80-register limit, spills some of them to scratch and reloads them in the SCAN path.  Deriving them from an
        // opaque copy of the lane id costs four instructions per round and keeps the register file for the state.
        lane = lane0;
        asm volatile("" : "+v"(lane));
        gl = lane & (G - 1);
        gshift = lane & ~(G - 1);
        bcast_addr = (uint32_t)(lane | (G - 1)) * 4u;
        nb2 = 2u * ((uint32_t)gl + 1u);
        if (G == 64) {
            // One group per wave: everything below is wave-uniform.  The reverse pass of run_correction (src/lib.rs:48-55)
            // reads the bases back to front WITHOUT complementing them, so hardly any of its k-mers is solid and nearly
            // every round is "64 positions, no trigger": that round in a loop of its own, without the state dispatch,
            // the trigger machinery and the register copies at their joins.  A round that does find a trigger is left to the general
            // code below, which redoes its probes (they are pure).
            while (have && st == ST_SCAN && !slow && n - i >= 65u && olen + 66u <= cap) {
                const uint8_t c8 = ld(i + (uint32_t)lane);
                uint32_t sc;
                const uint64_t km = lane_kmer(kmer, (uint32_t)nuc2bit(c8), sc);
                // (an overflowed index line is settled by the lane that met it, index_get)
                const bool s1 = p.idx.lines ? index_get(p.idx, p.bits, km, k) : probe(p.bits, km, k);
                const uint64_t bs = __ballot(s1);
                const uint64_t trig = ~bs & ((bs << 1) | (prev ? 1ull : 0ull)); // mod.rs:73
                if (trig)
                    break;
                out[olen + (uint32_t)lane] = c8; // mod.rs:100
                olen += 64u;
                i += 64u;
                kmer = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(km >> 32), 63) << 32) |
                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)km, 63);
                prev = (bs >> 63) & 1ull; // mod.rs:99
                ev += ((lane == 0) ? 1u : 0u) + (1u << 8);
                steps++;
                if (++since_flush == 255u)
                    flush();
            }
        }
        bool do_probe = false;
        uint64_t pk = 0;
        uint8_t ch = 0;
        uint32_t sc_s = 0, scan = 0, gwin = 0;
        bool sc_active = false;

        // ---------------- phase 1 -----------------------------------------------------------------------------------
        if (have) {
            ev += (gl == 0) ? 1u : 0u;
            if (st == ST_SCAN) {
                const uint32_t pos = i + (uint32_t)gl;
                const bool valid = pos < n;
                if (G < 64)
                    ch = pf_ch; // prefetched for exactly this i (every way into SCAN ends with the prefetch below)
                else
                    ch = valid ? ld(pos) : (uint8_t)0;
                pk = lane_kmer(kmer, (uint32_t)nuc2bit(ch), scan);
                do_probe = valid;
                if (G <= 16) // all G codes of the group, for the k-mer the group leaves this round with
                    gwin = (uint32_t)__builtin_amdgcn_ds_bpermute((int)bcast_addr, (int)scan);
            } else if (st == ST_INIT) {
                if (n < (uint32_t)k) {
                    for (uint32_t j = gl; j < n; j += G) // mod.rs:56-58
                        out[j] = ld(j);
                    olen = n;
                } else {
                    uint64_t km = 0;
                    for (int j = 0; j < k; j++)
                        km = (km << 2) | nuc2bit(ld((uint32_t)j));
                    kmer = km;
                    for (uint32_t j = gl; j < (uint32_t)k; j += G)
                        out[j] = ld(j);
                    olen = (uint32_t)k;
                    i = (uint32_t)k;
                    do_probe = (gl == 0);
                    pk = kmer;
                }
            } else if (st == ST_ALTS) {
                do_probe = gl < 4 && (uint64_t)gl != (kmer & 3ull); // the read's own base is the trigger k-mer: known non-solid
                pk = (kmer & ~3ull) | (uint64_t)gl;                  // add_nuc(kmer >> 2, gl)
                // the bases the scenarios will look at, fetched once: seq[i .. i+WB)
                if (G == 4) {
                    // four bases per lane: one (unaligned) dword, the four codes gathered into a byte by one multiply
                    const uint32_t w4 = i + 4u * (uint32_t)gl;
                    uint32_t raw = 0;
                    if (w4 + 4u <= n) {
                        const uint8_t *q = p.flip ? in + (n - 4u - w4) : in + w4;
                        __builtin_memcpy(&raw, q, 4);
                        if (p.flip)
                            raw = __builtin_bswap32(raw);
                    } else {
                        for (uint32_t t = 0; t < 4u && w4 + t < n; t++)
                            raw |= (uint32_t)ld(w4 + t) << (8u * t);
                    }
                    // bytes b0..b3 (b0 = first base) -> b0<<6 | b1<<4 | b2<<2 | b3: the four products land in bits 24-31, every
                    // other partial product below bit 24 or beyond bit 31, none overlapping
                    uint32_t v = ((((raw >> 1) & 0x03030303u) * 0x40100401u) >> 24);
                    v |= dpp_row_shr<1>(v) << 8;
                    v |= dpp_row_shr<2>(v) << 16;
                    if (gl != 3)
                        v &= (1u << (8u * ((uint32_t)gl + 1u))) - 1u;
                    win = (uint32_t)__builtin_amdgcn_ds_bpermute((int)bcast_addr, (int)v);
                }
                const uint32_t wp = i + (uint32_t)gl;
                const uint32_t cd = (G != 4 && gl < WB && wp < n) ? (uint32_t)nuc2bit(ld(wp)) : 0u;
                if (G == 4) {
                } else if (G == 8) {
                    const uint32_t v = row_scan8(cd) & ((1u << nb2) - 1u);
                    win = (uint32_t)__builtin_amdgcn_ds_bpermute((int)bcast_addr, (int)v);
                } else if (G == 16) {
                    win = (uint32_t)__builtin_amdgcn_ds_bpermute((int)bcast_addr, (int)row_scan16(cd));
                } else {
                    // rows 0 and 1 of the wave hold the 32 window bases
                    const uint32_t v = row_scan16(cd);
                    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), w1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
                    win = ((uint64_t)w0 << 32) | w1;
                }
            } else if (st == ST_SCEN && (G == 8 || G == 16)) {
                // get_score's look-aheads (exist/mod.rs:21-47) with a FIXED lane -> probe map (dealing the lanes to the
                // surviving scenarios at run time cost more instructions than the rounds it saved):
                //   stage A (sub == 0): lane 2s + j probes look-ahead j < 2 of scenario s -- a wrong scenario dies here;
                //   stage B: the lowest scenario still in and not yet complete gets lanes 0.. for its look-aheads sub..c-1.
                uint32_t off, j;
                if (sub == 0u) {
                    sc_s = (uint32_t)gl >> 1;
                    j = (uint32_t)gl & 1u;
                    sc_active = gl < 6 && !((failmask >> sc_s) & 1u) && j < c;
                } else {
                    sc_s = (uint32_t)__ffs(7u & ~failmask & ~passmask) - 1u;
                    j = sub + (uint32_t)gl;
                    sc_active = j < c;
                }
                off = 2u - sc_s; // I:2 S:1 D:0 (one.rs:57-63)
                if (sc_active) {
                    if (c + 3u <= (uint32_t)WB) {
                        pk = ext(corr, off, j + 1u);
                    } else {
                        pk = corr;
                        for (uint32_t q = 0; q <= j; q++)
                            pk = add_nuc(pk, nuc2bit(ld(i + off + q)), mask);
                    }
                    do_probe = true;
                }
            } else if (st == ST_SCEN) {
                const uint32_t width = scen_width(sub, c, G, 0u, failmask);
                const uint32_t e = (uint32_t)gl;
                const uint32_t alive = ~failmask & 7u;
                // ord = e / width, e % width -- at most three scenarios, so two compares do the division
                const uint32_t ord = (e >= width ? 1u : 0u) + (e >= 2u * width ? 1u : 0u);
                sc_active = e < 3u * width && ord < (uint32_t)__popc(alive);
                const uint32_t a1 = alive & (alive - 1u);
                sc_s = !sc_active ? 0u : (ord == 0u ? (uint32_t)__ffs(alive) - 1u : (ord == 1u ? (uint32_t)__ffs(a1) - 1u : 2u));
                const uint32_t j = sub + (sc_active ? e - ord * width : 0u);
                const uint32_t off = 2u - sc_s; // I:2 S:1 D:0 (one.rs:57-63)
                if (sc_active) {
                    if (c + 3u <= (uint32_t)WB) {
                        pk = ext(corr, off, j + 1u);
                    } else {
                        pk = corr;
                        for (uint32_t q = 0; q <= j; q++)
                            pk = add_nuc(pk, nuc2bit(ld(i + off + q)), mask);
                    }
                    do_probe = true;
                }
            } else { // ST_MORE
                if (gl < 3 && ((passmask >> gl) & 1u)) {
                    const uint32_t off = 2u - (uint32_t)gl;
                    const uint32_t rem = n - i;
                    if (rem > c + off + 1u) { // exist/mod.rs:54
                        if (c + 3u <= (uint32_t)WB) {
                            pk = ext(corr, off, c + 1u);
                        } else {
                            pk = corr;
                            for (uint32_t q = 0; q <= c; q++)
                                pk = add_nuc(pk, nuc2bit(ld(i + off + q)), mask);
                        }
                        do_probe = true;
                    }
                }
            }
            if (++steps > (1u << 24) + 64u * n) // runaway guard, as in correct_kernel
                do_probe = false;
        }

        // ---------------- phase 2: one probe per lane -----------------------------------------------------------------
        bool sol = false, retry = false;
        if (p.idx.lines) {
            bool unres = false;
            if (do_probe) {
                if (!slow) {
                    // (through the occupancy bits this costs the forward pass 0.5 ms: its k-mers are mostly present, and
                    // the extra dependent load only lengthens the round)
                    const int pr = index_probe(p.idx, pk, k);
                    sol = pr == 1;
                    unres = pr == 2;
                    ev += 1u << 8;
                } else if (was_unres) {
                    if (p.bits) {
                        sol = probe(p.bits, pk, k);
                    } else {
                        const int pr = index_probe(p.idx, pk, k, hop);
                        sol = pr == 1;
                        unres = pr == 2;
                    }
                    ev += 1u << 8;
                } else {
                    sol = kept_sol;
                }
            }
            const uint64_t ub = __ballot(unres);
            retry = G == 64 ? ub != 0ull : (((uint32_t)(ub >> gshift) & GM32) != 0u);
            hop = retry ? hop + 1u : 0u;
            slow = retry;
            was_unres = unres;
            kept_sol = sol;
        } else if (do_probe) {
            sol = probe(p.bits, pk, k);
            ev += 1u << 8;
        }
        const uint64_t ball = __ballot(sol);

        // ---------------- phase 3 -------------------------------------------------------------------------------------
        if (have && !retry) {
            bool fail = false;
            int apply_s = -1;
            if (st == ST_SCAN) {
                const uint32_t left = n - i;
                const uint32_t nvalid = left < (uint32_t)G ? left : (uint32_t)G;
                uint32_t nacc;
                bool trg;
                bool last_sol;
                if (G == 64) {
                    const uint64_t vmask = (nvalid >= 64) ? ~0ull : ((1ull << nvalid) - 1ull);
                    const uint64_t trig = ~ball & ((ball << 1) | (prev ? 1ull : 0ull)) & vmask; // mod.rs:73
                    trg = trig != 0ull;
                    nacc = trg ? (uint32_t)__builtin_ctzll(trig) : nvalid;
                    last_sol = nacc ? ((ball >> (nacc - 1u)) & 1ull) : prev;
                } else {
                    const uint32_t gm = (uint32_t)(ball >> gshift) & GM32;
                    const uint32_t vmask = (1u << nvalid) - 1u; // G <= 16
                    const uint32_t trig = ~gm & ((gm << 1) | (prev ? 1u : 0u)) & vmask;
                    trg = trig != 0u;
                    nacc = trg ? (uint32_t)__builtin_ctz(trig) : nvalid;
                    last_sol = nacc ? ((gm >> (nacc - 1u)) & 1u) : prev;
                }
                if (olen + nacc + 1u > cap) {
                    give_up(CTL_OVERFLOW);
                } else {
                    if ((uint32_t)gl < nacc)
                        out[olen + (uint32_t)gl] = ch; // mod.rs:100
                    olen += nacc;
                    i += nacc;
                    // the k-mer the group goes on with: after the accepted bases, plus the trigger base if there is one
                    const uint32_t ncodes = nacc + (trg ? 1u : 0u);
                    if (G == 64) {
                        const int src = (int)ncodes - 1; // ncodes >= 1: nvalid >= 1 in SCAN
                        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pk, src);
                        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pk >> 32), src);
                        kmer = ((uint64_t)hi << 32) | lo;
                    } else {
                        kmer = ((kmer << (2u * ncodes)) | (uint64_t)(gwin >> (2u * ((uint32_t)G - ncodes)))) & mask;
                    }
                    if (trg) {
                        ev += (gl == 0) ? (1u << 16) : 0u;
                        st = ST_ALTS;
                    } else {
                        prev = last_sol; // mod.rs:99
                        if (i >= n)
                            finish();
                    }
                }
            } else if (st == ST_INIT) {
                if (n < (uint32_t)k) {
                    finish();
                } else {
                    prev = (ball >> gshift) & 1ull; // mod.rs:67
                    if (i >= n)
                        finish();
                    else
                        st = ST_SCAN;
                }
            } else if (st == ST_ALTS) {
                const uint32_t am = (uint32_t)(ball >> gshift) & 0xfu;
                if (__popc(am) != 1) {
                    fail = true; // exist/mod.rs:123-126
                } else {
                    corr = (kmer & ~3ull) | (uint64_t)(__ffs(am) - 1);
                    const uint32_t rem = n - i;
                    failmask = 0;
                    for (uint32_t s = 0; s < 3; s++)
                        if ((2u - s) + c > rem) // exist/mod.rs:27-29
                            failmask |= 1u << s;
                    if (failmask == 7u) {
                        fail = true;
                    } else if (c == 0u) {
                        passmask = 7u & ~failmask;
                        if (__popc(passmask) == 1)
                            apply_s = __ffs(passmask) - 1;
                        else
                            st = ST_MORE;
                    } else {
                        sub = 0;
                        st = ST_SCEN;
                    }
                }
            } else if (st == ST_SCEN && (G == 8 || G == 16)) {
                const uint32_t badm = (uint32_t)(__ballot(sc_active && !sol) >> gshift) & GM32;
                const uint32_t two = c < 2u ? c : 2u;
                bool finished = false;
                if (sub == 0u) {
                    failmask |= ((badm & 3u) ? 1u : 0u) | ((badm & 12u) ? 2u : 0u) | ((badm & 48u) ? 4u : 0u);
                    passmask = 0; // from here on: the scenarios whose c look-aheads all held
                    sub = two;
                    if (c <= 2u) {
                        passmask = 7u & ~failmask;
                        finished = true;
                    }
                } else {
                    const uint32_t cur = (uint32_t)__ffs(7u & ~failmask & ~passmask) - 1u; // as in phase 1
                    if (badm) {
                        failmask |= 1u << cur;
                        sub = two;
                    } else {
                        const uint32_t left = c - sub;
                        sub += left < (uint32_t)G ? left : (uint32_t)G;
                        if (sub >= c) {
                            passmask |= 1u << cur;
                            sub = two;
                        }
                    }
                }
                if (failmask == 7u) {
                    fail = true; // exist/mod.rs:132-134
                } else if (finished || (7u & ~failmask & ~passmask) == 0u) {
                    if (passmask == 0u)
                        fail = true;
                    else if (__popc(passmask) == 1)
                        apply_s = __ffs(passmask) - 1; // exist/mod.rs:135-137
                    else
                        st = ST_MORE;
                }
            } else if (st == ST_SCEN) {
                const bool bad = sc_active && !sol;
                const uint32_t w_used = scen_width(sub, c, G, 0u, failmask);
                // lanes [ord*w, (ord+1)*w) worked for the ord-th surviving scenario
                const uint64_t bb = __ballot(bad);
                const uint32_t badm = G == 64 ? 0u : ((uint32_t)(bb >> gshift) & GM32);
                uint32_t alive = ~failmask & 7u, ord = 0;
                for (uint32_t s = 0; s < 3; s++)
                    if ((alive >> s) & 1u) {
                        bool any;
                        if (G == 64) {
                            const uint64_t wm = (w_used >= 64u ? ~0ull : ((1ull << w_used) - 1ull)) << (ord * w_used);
                            any = (bb & wm) != 0ull;
                        } else {
                            any = ((badm >> (ord * w_used)) & ((1u << w_used) - 1u)) != 0u;
                        }
                        if (any)
                            failmask |= 1u << s;
                        ord++;
                    }
                sub += w_used;
                if (failmask == 7u) {
                    fail = true; // exist/mod.rs:132-134
                } else if (sub >= c) {
                    passmask = 7u & ~failmask;
                    if (__popc(passmask) == 1)
                        apply_s = __ffs(passmask) - 1; // exist/mod.rs:135-137
                    else
                        st = ST_MORE;
                }
            } else { // ST_MORE
                const uint32_t keep = (uint32_t)(ball >> gshift) & 7u & passmask;
                if (__popc(keep) == 1)
                    apply_s = __ffs(keep) - 1; // exist/mod.rs:143-144
                else
                    fail = true;
            }
            if (have && steps > (1u << 24) + 64u * n) {
                fail = false;
                apply_s = -1;
                give_up(CTL_NONTERM);
            }
            if (fail) {
                // mod.rs:91-96: the trigger base is copied through
                if (olen + 2u > cap) {
                    give_up(CTL_OVERFLOW);
                } else {
                    if (gl == 0)
                        out[olen] = ld(i);
                    olen += 1;
                    i += 1;
                    prev = false;
                    if (i >= n)
                        finish();
                    else
                        st = ST_SCAN;
                }
            } else if (apply_s >= 0) {
                // mod.rs:75-89 with one.rs:65-71
                if (olen + 2u > cap) {
                    give_up(CTL_OVERFLOW);
                } else {
                    if (gl == 0)
                        out[olen] = bit2nuc(corr & 3ull);
                    olen += 1;
                    kmer = corr;
                    prev = true;
                    const uint32_t off = 2u - (uint32_t)apply_s;
                    i += off;
                    ev += (gl == 0) ? (1u << 24) : 0u;
                    // the c look-ahead k-mers of the winning scenario ARE the next c scan k-mers, all found solid: the
                    // reference's loop would copy these bases with previous = true (mod.rs:99-102)
                    if (c > 0u && c + 3u <= (uint32_t)WB && olen + c + 1u <= cap) {
                        for (uint32_t q = (uint32_t)gl; q < c; q += (uint32_t)G)
                            out[olen + q] = ld(i + q);
                        kmer = ext(kmer, off, c);
                        olen += c;
                        i += c;
                    }
                    if (i >= n)
                        finish();
                    else
                        st = ST_SCAN;
                }
            }
        }
        if (G < 64 && have && !done && !retry && st == ST_SCAN) {
            const uint32_t pos = i + (uint32_t)gl;
            pf_ch = pos < n ? ld(pos) : (uint8_t)0;
        }
        if (done) {
            if (gl == 0) {
                p.out_lens[r] = done == 1 ? olen : 0xffffffffu;
                if (done != 1)
                    atomicAdd(p.ctrl + (done - 16), 1ull);
            }
            done = 0;
            fetch();
        }
        if (++since_flush == 255u)
            flush();
    }
    flush();
}

// ======================================================================================================================
// Reverse passes of Graph / GapSize, lean form.  run_correction's second direction (src/lib.rs:48-55) reads
// the read back to front WITHOUT complementing it: hardly a k-mer is solid, a trigger (mod.rs:73) is a chance hit, and what
// follows it is all but always one of
//   error_len (mod.rs:130-152) runs off the read        -> Graph: None (the target is not solid, no walk can reach it);
//                                                          GapSize with error_len < k: the same
//   alt_nucs does not name exactly one alternative      -> None for every method (exist/mod.rs:123-126, graph.rs:51-55,
//                                                          greedy.rs, gap_size.rs:47-50)
//   one alternative (the k-mer in front of the trigger is a genome k-mer, the alternative its successor in the genome)
//                                                       -> the method goes to work and returns None after all: the walk
//                                                          follows the genome and never meets a target that lies in a
//                                                          reversed read; the scenarios' look-aheads do not hold
// i.e. the read leaves the pass as it came.  This kernel scans on that assumption -- scan, error_len, alt_nucs, one wave
// per read, 64 k-mers a round, nothing of the methods' state -- and NOTES every trigger of the third kind (TrigRec).  A
// second launch, correct_kernel<8, M, 2> (<4, M, 2> on request), takes the notes as its work items: a narrow group enters the method at
// alt_nucs with the state the scan would have had there, runs it, writes nothing, and flags the read unless the result
// is None; flagged reads (none for Graph, 4 956 of 100 000 for GapSize over the bench's data) are then corrected from
// scratch by the 64-lane group kernel over their list (correct_kernel<64, M, 1>).  A read none of whose triggers is flagged is exactly the copy the reference makes:
// by induction over its triggers, each was reached with the reference's state and returned None like the reference's.
// What it saves is the group kernel's register file (5 waves per SIMD against 7 here) and the instructions of its state
// dispatch around rounds that are nearly all "64 positions, nothing".
// ======================================================================================================================
#ifndef BRX_REV_WAVES
#define BRX_REV_WAVES 7 // (8 / 7 / 6 waves measure the same: profiles/r4o_rev_lean_ab.txt; 62 registers for k = 19 / 21)
#endif
template <int KT, int M>
__global__ __launch_bounds__(256, BRX_REV_WAVES) void rev_scan_kernel(const PassParams pp, uint32_t *__restrict__ handback, uint32_t *__restrict__ redo_flag,
                                                                          TrigRec *__restrict__ notes, uint32_t trig_cap)
{
    // (the fields the loop needs, by value: a lambda that captures the parameter block by reference makes the compiler
    // keep a copy of all of it in scratch)
    const IdxView idx = pp.idx;
    const uint32_t *const bits = pp.bits;
    unsigned long long *const ctrl = pp.ctrl;
    const uint32_t n_reads = pp.n_reads, slack = pp.slack;
    const uint64_t *const offsets = pp.offsets;
    const uint8_t *const in_base = pp.in;
    const uint32_t *const in_lens = pp.in_lens;
    const bool in_staged = pp.in_staged != 0, flip = pp.flip != 0;
    uint8_t *const out_base = pp.out;
    uint32_t *const out_lens = pp.out_lens;
    constexpr bool HAS_ERRLEN = (M == BRX_GRAPH || M == BRX_GAP_SIZE);
    const int lane = threadIdx.x & 63;
    const int k = KT ? KT : pp.k;
    const uint64_t mask = kmask(k);
    uint32_t n_rounds = 0, n_probes = 0, n_trig = 0;

    // KmerSet::get.  An index line that overflowed at build time cannot say "absent" (one probe in a thousand, i.e. nearly
    // every 10 kb read meets one): the lane asks the bit vector, or -- sparse sets, whose keys chain into the following
    // lines -- the next lines of the chain, while its wave waits.
    // (always_inline: left to itself the compiler CALLS the probe from the three places that ask -- s_swappc, registers saved
    // through lanes and scratch -- and the scan ran at 9.1 ms per Gbp where one_kernel<64>'s identical loop takes 5.7)
    auto ask = [&](uint64_t km) __attribute__((always_inline)) -> bool { return idx.lines ? index_get(idx, bits, km, k) : probe(bits, km, k); };
    auto last_of = [&](uint64_t km, uint32_t l) __attribute__((always_inline)) -> uint64_t {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(km >> 32), (int)l) << 32) |
               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)km, (int)l);
    };

    for (;;) {
        unsigned long long w = 0;
        if (lane == 0)
            w = atomicAdd(ctrl + CTL_WORK, 1ull);
        // (lane 0's value into a scalar register: everything derived from it -- the read's pointers, its length, the
        // loop state -- then lives in scalar registers, too)
        const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w);
        if (__builtin_amdgcn_readfirstlane((int)(uint32_t)(w >> 32)) != 0 || r >= n_reads)
            break;
        const uint64_t o0 = offsets[r], o1 = offsets[r + 1];
        const uint8_t *in;
        uint32_t n;
        if (in_staged) {
            in = in_base + slot_of(o0, r, slack);
            n = in_lens[r];
            if (n == 0xffffffffu) { // poisoned by an earlier pass of this attempt: stays poisoned
                if (lane == 0)
                    out_lens[r] = 0xffffffffu;
                continue;
            }
        } else {
            in = in_base + o0;
            n = (uint32_t)(o1 - o0);
        }
        const uint64_t s0 = slot_of(o0, r, slack), s1 = slot_of(o1, (uint64_t)r + 1, slack);
        uint8_t *out = out_base + s0;
        const uint32_t cap = (uint32_t)(s1 - s0);
        auto ld = [&](uint32_t j) __attribute__((always_inline)) -> uint8_t { return in[flip ? (n - 1u - j) : j]; };

        if (n < (uint32_t)k) { // mod.rs:56-58
            for (uint32_t j = (uint32_t)lane; j < n; j += 64u)
                out[j] = ld(j);
            if (lane == 0)
                out_lens[r] = n;
            continue;
        }
        bool hb = n + 1u > cap; // (the group kernel's slot check; it poisons the read)
        uint64_t kmer = 0;
        uint32_t i = (uint32_t)k, skip_until = 0, trig_read = 0;
        bool prev = false;
        if (!hb) {
            for (int j = 0; j < k; j++)
                kmer = (kmer << 2) | nuc2bit(ld((uint32_t)j));
            if (lane < k)
                out[lane] = ld((uint32_t)lane);
            prev = ask(kmer); // mod.rs:67
            n_rounds++;
            n_probes++;
        }
        while (!hb && i < n) {
            // ---- scan: 64 positions (mod.rs:68-104 while nothing triggers) -------------------------------------------
            const uint32_t pos = i + (uint32_t)lane;
            const bool valid = pos < n;
            const uint8_t c8 = valid ? ld(pos) : (uint8_t)0;
            const uint64_t km = lane_kmer64_dpp(kmer, (uint32_t)nuc2bit(c8), lane, mask);
            // (positions below skip_until: error_len asked about them behind a trigger that failed -- not solid)
            const bool asks = valid && !(HAS_ERRLEN && pos < skip_until);
            const bool a1 = asks && ask(km);
            n_rounds++;
            n_probes += (uint32_t)__builtin_popcountll(__ballot(asks));
            const uint32_t left = n - i;
            const uint32_t cnt = left < 64u ? left : 64u;
            const uint64_t vmask = cnt >= 64u ? ~0ull : ((1ull << cnt) - 1ull);
            const uint64_t bs = __ballot(a1);
            const uint64_t trig = ~bs & ((bs << 1) | (prev ? 1ull : 0ull)) & vmask; // mod.rs:73
            if (!trig) {
                if (valid)
                    out[pos] = c8; // mod.rs:100 (nothing changes the read here: output position == input position)
                kmer = last_of(km, cnt - 1u);
                prev = (bs >> (cnt - 1u)) & 1ull; // mod.rs:99
                i += cnt;
                continue;
            }
            const uint32_t t = (uint32_t)__builtin_ctzll(trig);
            if ((uint32_t)lane < t)
                out[pos] = c8;
            i += t;
            kmer = last_of(km, t); // the k-mer that triggered; the read's base stays in it when the method fails (mod.rs:91-96)
            trig_read++;
            uint32_t elen = 0;
            uint64_t fc = 0;
            bool hit_end = false;
            if (HAS_ERRLEN) {
                // ---- error_len, mod.rs:130-152: the k-mers behind the trigger until the first solid one ---------------
                uint64_t ek = kmer;
                uint32_t ej = 0;
                const uint32_t rem = n - i;
                for (;;) {
                    const uint32_t j = ej + 1u + (uint32_t)lane;
                    const bool v2 = j < rem;
                    const uint8_t c2 = v2 ? ld(i + j) : (uint8_t)0;
                    const uint64_t km2 = lane_kmer64_dpp(ek, (uint32_t)nuc2bit(c2), lane, mask);
                    const bool a2 = v2 && ask(km2);
                    n_rounds++;
                    n_probes += (uint32_t)__builtin_popcountll(__ballot(v2));
                    const uint64_t sm = __ballot(a2);
                    if (sm) {
                        const uint32_t t2 = (uint32_t)__builtin_ctzll(sm);
                        elen = ej + 1u + t2;
                        fc = last_of(km2, t2);
                        break;
                    }
                    if (rem <= ej + 1u + 64u) { // ran off the read: the last k-mer built is not solid
                        const uint32_t nv = rem - (ej + 1u);
                        elen = rem;
                        fc = nv ? last_of(km2, nv - 1u) : ek;
                        hit_end = true;
                        break;
                    }
                    ek = last_of(km2, 63u);
                    ej += 64u;
                }
            }
            // Graph: a target that is not solid cannot be reached (every walk k-mer is solid, graph.rs:79); GapSize sends
            // error_len < k the same way (gap_size.rs:97-108)
            const bool none_already = HAS_ERRLEN && hit_end && (M == BRX_GRAPH || elen < (uint32_t)k);
            if (!none_already) {
                // ---- alt_nucs: the read's own base is the trigger k-mer, known not solid ----------------------------
                const bool asks3 = lane < 4 && (uint64_t)lane != (kmer & 3ull);
                const bool a3 = asks3 && ask(add_nuc(kmer >> 2, (uint64_t)(lane & 3), mask));
                n_rounds++;
                n_probes += 3u;
                if (__builtin_popcountll(__ballot(a3)) == 1) {
                    // A unique alternative: the method goes to work -- in a reverse pass all but always to return None
                    // after all (a walk along the genome that never meets the target, scenarios that do not hold).  The
                    // scan goes on as if it had; the group kernel's verify pass checks this trigger (SRC == 2).
                    unsigned long long at = 0;
                    if (lane == 0)
                        at = atomicAdd(ctrl + CTL_REV_TRIGS, 1ull);
                    const uint32_t at_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)at);
                    if (__builtin_amdgcn_readfirstlane((int)(uint32_t)(at >> 32)) != 0 || at_lo >= trig_cap) {
                        hb = true; // (no room to note it: the whole read goes to the group kernel)
                        break;
                    }
                    if (lane == 0) {
                        notes[at_lo] = TrigRec{r, i, elen, 0u, fc};
                        atomicAdd(ctrl + CTL_REV_TRIGS_SUM, 1ull);
                    }
                }
            }
            // ---- None: mod.rs:91-96 ------------------------------------------------------------------------------------
            if (lane == 0)
                out[i] = ld(i);
            if (HAS_ERRLEN)
                skip_until = i + elen;
            i += 1u;
            prev = false;
            if (HAS_ERRLEN && skip_until >= n) {
                // no k-mer behind the trigger is solid: nothing triggers any more (`previous` stays false)
                for (uint32_t j = i + (uint32_t)lane; j < n; j += 64u)
                    out[j] = ld(j);
                i = n;
            }
        }
        if (lane == 0) {
            if (hb) {
                if (atomicExch(redo_flag + r, 1u) == 0u) {
                    handback[atomicAdd(ctrl + CTL_REV_HANDBACK, 1ull)] = r;
                    atomicAdd(ctrl + CTL_REV_HANDBACK_SUM, 1ull);
                }
            } else {
                out_lens[r] = n;
            }
        }
        if (!hb)
            n_trig += trig_read; // (a read handed back counts its events in the group kernel)
    }
    if (lane == 0) {
        if (n_rounds)
            atomicAdd(ctrl + CTL_ROUNDS, (unsigned long long)n_rounds);
        if (n_probes)
            atomicAdd(ctrl + CTL_PROBES, (unsigned long long)n_probes);
        if (n_trig)
            atomicAdd(ctrl + CTL_TRIGGERS, (unsigned long long)n_trig);
    }
}

// one workgroup per read (grid-stride): staged slot -> compact output, reversing if needed
__global__ __launch_bounds__(256) void compact_kernel(const uint8_t *__restrict__ stage, const uint32_t *__restrict__ lens,
                                                      const uint64_t *__restrict__ offsets, uint32_t n_reads,
                                                      uint32_t slack, int reversed, const uint64_t *__restrict__ out_offsets,
                                                      uint8_t *__restrict__ out)
{
    for (uint32_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const uint64_t s0 = slot_of(offsets[r], r, slack);
        const uint8_t *src = stage + s0;
        uint32_t n = lens[r];
        {   // a read redone outside the batch that does not fit its slot has its length here but its bytes elsewhere (they
            // are copied over this read's place afterwards): copy no more than the slot holds
            const uint64_t slot = slot_of(offsets[r + 1], (uint64_t)r + 1, slack) - s0;
            if ((uint64_t)n > slot)
                n = (uint32_t)slot;
        }
        uint8_t *dst = out + out_offsets[r];
        // bytes up to the first 16-byte boundary of dst, then 16 bytes per lane (unaligned load, byte order reversed
        // in registers when the staged read is stored back to front, aligned store), then the tail
        uint32_t head = (uint32_t)((16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u);
        if (head > n)
            head = n;
        const uint32_t nv = (n - head) / 16u;
        for (uint32_t j = threadIdx.x; j < head; j += blockDim.x)
            dst[j] = reversed ? src[n - 1u - j] : src[j];
        for (uint32_t v = threadIdx.x; v < nv; v += blockDim.x) {
            const uint32_t j = head + 16u * v;
            uint4 q;
            if (reversed) {
                __builtin_memcpy(&q, src + (n - 16u - j), 16);
                q = make_uint4(__builtin_bswap32(q.w), __builtin_bswap32(q.z), __builtin_bswap32(q.y), __builtin_bswap32(q.x));
            } else {
                __builtin_memcpy(&q, src + j, 16);
            }
            *reinterpret_cast<uint4 *>(dst + j) = q;
        }
        for (uint32_t j = head + 16u * nv + threadIdx.x; j < n; j += blockDim.x)
            dst[j] = reversed ? src[n - 1u - j] : src[j];
    }
}

int ensure(void **p, uint64_t *cap, uint64_t need)
{
    if (need <= *cap && *p)
        return BRX_OK;
    if (*p)
        (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    uint64_t want = need + need / 8 + 256;
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {
        set_error("hipMalloc(%llu B workspace): %s", (unsigned long long)want, hipGetErrorString(e));
        return BRX_ERR_NOMEM;
    }
    *cap = want;
    return BRX_OK;
}

constexpr uint64_t RESIDENT_LANES = 256ull * 4ull * 6ull * 64ull; // CUs x SIMDs x waves x lanes at occupancy 6

int group_width(bool reverse_pass = false, bool indexed = false, uint32_t n_reads = 0xffffffffu)
{
    // lanes per read; BRX_GROUP / BRX_GROUP_REV override (read on every launch so tests can sweep them).
    // The reverse pass of run_correction (src/lib.rs:48-55) sees almost only non-solid k-mers and hardly
    // ever triggers, so wide groups waste nothing there and save rounds.
    const char *e = getenv(reverse_pass ? "BRX_GROUP_REV" : "BRX_GROUP");
    if (!e && reverse_pass)
        e = getenv("BRX_GROUP");
    // measured (tools/ab_correct.py): bitset probes 16/64 (+3 % over 16/16); with the probe index the shared
    // probe code is the larger part of a round and 8 groups per wave amortise it better: 8/64 (+5 % over 16/64)
    const int dflt = reverse_pass ? 64 : (indexed ? 8 : 16);
    int g = e ? atoi(e) : dflt;
    if (!e) {
        // small batches (the reference's 8192 records): with fewer reads than resident groups the GPU idles and
        // the time is one read's chain of rounds -- wider groups advance further per round
        while (g < 64 && (uint64_t)n_reads * (uint64_t)g * 2u <= RESIDENT_LANES)
            g *= 2;
    }
    return (g == 4 || g == 8 || g == 16 || g == 32 || g == 64) ? g : dflt;
}

constexpr uint32_t MAX_BLOCKS = 256u * 8u;
// Visited lists a chain always has, however few reads its batch holds: one block of the narrowest kernel that indexes them
// by its global group number (4-lane groups: 64 a block).  The list and verify launches size their grids from the lists
// the chain has (sized_path_lists), never from a width of their own.
constexpr uint32_t MIN_PATH_LISTS = 64u;

// lanes per read of Graph's and GapSize's forward passes (BRX_GROUP_WALK = 4 / 8 / 16)
int walk_group()
{
    // measured at 1 Gbp, k = 19 (fwd+rev ms, tools/method_bench.py): 16 lanes + bit vector 135 / 158 (graph / gap_size),
    // 8 lanes + bit vector 140 / 145, 8 lanes + probe index 120 / 142, 4 lanes + probe index 123 / 151
    const char *e = getenv("BRX_GROUP_WALK"); // (read on every use: the fuzzers sweep it)
    const int v = e ? atoi(e) : 8;
    return (v == 4 || v == 8 || v == 16) ? v : 8;
}

uint32_t pass_blocks(uint32_t n_reads, int G, bool balanced = false)
{
    const uint32_t groups_per_block = 256u / (uint32_t)G;
    // persistent-ish grid: enough groups to fill the chip, reads pulled from a work counter
    uint64_t want = ((uint64_t)n_reads + groups_per_block - 1) / groups_per_block;
    if (balanced) {
        // Reads of a batch take about the same time, and a wave costs the same instructions whether 8 of its groups
        // are busy or 3: 100 000 reads over 49 152 resident 8-lane groups is two reads each and a third for a few,
        // i.e. a last third of the kernel run by mostly idle waves.  Fewer groups with the same whole number of reads
        // each end together.
        // Measured (profiles/r2_one_kernel_ab.txt): NOT a win -- 4 168 waves instead of 8 192 leave the SIMDs waiting on
        // memory (VALU busy 65 % instead of 79 %, 32.6 ms against 27.2); kept behind BRX_BALANCE=1 for the record.
        static const bool on = [] { const char *e = getenv("BRX_BALANCE"); return e && *e == '1'; }();
        const uint64_t resident_blocks = 256ull * 7; // CUs x (waves per SIMD x 4 SIMDs / 4 waves per block): one_kernel runs 7 per SIMD
        const uint64_t resident_groups = resident_blocks * groups_per_block;
        if (on && (uint64_t)n_reads > resident_groups) {
            const uint64_t per_group = ((uint64_t)n_reads + resident_groups - 1) / resident_groups;
            const uint64_t groups = ((uint64_t)n_reads + per_group - 1) / per_group;
            want = (groups + groups_per_block - 1) / groups_per_block;
        }
    }
    if (want > MAX_BLOCKS)
        want = MAX_BLOCKS;
    if (want < 1)
        want = 1;
    return (uint32_t)want;
}

// visited lists of a chain for a batch of n_reads (the walking methods' group kernels index them by global group number)
uint64_t sized_path_lists(uint32_t n_reads)
{
    const uint64_t g = (uint64_t)pass_blocks(n_reads, walk_group()) * (256u / (uint32_t)walk_group());
    return g < MIN_PATH_LISTS ? MIN_PATH_LISTS : g;
}

template <int G>
void launch_one(const PassParams &p, uint32_t blocks, hipStream_t s)
{
    if (p.k == 19)
        one_kernel<G, 19><<<blocks, 256, 0, s>>>(p);
    else if (p.k == 21)
        one_kernel<G, 21><<<blocks, 256, 0, s>>>(p);
    else
        one_kernel<G, 0><<<blocks, 256, 0, s>>>(p);
}

} // namespace

namespace brx {
// Graph's / GapSize's group kernel over the reads p.only[0 .. *p.only_n) (8-lane groups; the visited lists of the chain
// are sized for pass_blocks(n_reads, walk_group()) blocks of groups at least as wide, so the grid stays inside them)
int launch_walk_list(const PassParams &p, int method, hipStream_t s)
{
    const uint64_t sized_groups = sized_path_lists(p.n_reads);
    uint32_t bl = (uint32_t)(sized_groups / 32u); // 32 eight-lane groups per block (the chain sizes at least 64 lists)
    bl = bl < 1u ? 1u : (bl > 128u ? 128u : bl);
    if (method == BRX_GRAPH)
        correct_kernel<8, BRX_GRAPH, 1><<<bl, 256, 0, s>>>(p);
    else
        correct_kernel<8, BRX_GAP_SIZE, 1><<<bl, 256, 0, s>>>(p);
    BRX_HIP(hipGetLastError());
    return BRX_OK;
}

// One's group kernel over the reads p.only[0 .. *p.only_n): what the lane-per-chunk pass could not stitch
int launch_one_list(const PassParams &p, hipStream_t s)
{
    const uint32_t blocks = 128; // 4096 eight-lane groups; the list is a few reads, and the grid loops over it
    if (p.k == 19)
        one_kernel<8, 19, true><<<blocks, 256, 0, s>>>(p);
    else if (p.k == 21)
        one_kernel<8, 21, true><<<blocks, 256, 0, s>>>(p);
    else
        one_kernel<8, 0, true><<<blocks, 256, 0, s>>>(p);
    BRX_HIP(hipGetLastError());
    return BRX_OK;
}
} // namespace brx

namespace {

// lanes per open trigger of the verify pass (BRX_REV_VERIFY_G = 4 / 8)
int verify_group()
{
    const char *e = getenv("BRX_REV_VERIFY_G"); // (read on every use: the fuzzers sweep it)
    return e && atoi(e) == 4 ? 4 : 8;
}

template <int M>
int launch_method(const PassParams &p, int G, uint32_t blocks, size_t lds, hipStream_t s)
{
    if constexpr (M == BRX_ONE) {
        if (p.trig) { // the triggers One's lean reverse pass left open (BRX_REV_LEAN_ONE), one 8-lane group each
            correct_kernel<8, BRX_ONE, 2><<<blocks, 256, lds, s>>>(p);
            return BRX_OK;
        }
        if (p.only)
            return launch_one_list(p, s);
    }
    if (M == BRX_ONE && G != 32) {
        // (4-lane groups exist in this form only)
        // BRX_ONE_V1=1: the first form of the kernel (A/B runs; results are identical)
        static const bool v1 = [] { const char *e = getenv("BRX_ONE_V1"); return e && *e == '1'; }();
        if (!v1) {
            if (G == 4)
                launch_one<4>(p, blocks, s);
            else if (G == 8)
                launch_one<8>(p, blocks, s);
            else if (G == 16)
                launch_one<16>(p, blocks, s);
            else
                launch_one<64>(p, blocks, s);
            return BRX_OK;
        }
    }
    if constexpr (M == BRX_GRAPH || M == BRX_GAP_SIZE) {
        if (p.trig) {
            // the triggers a lean reverse pass left open, one narrow group each (a walk step is four probes)
            if (verify_group() == 4)
                correct_kernel<4, M, 2><<<blocks, 256, lds, s>>>(p);
            else
                correct_kernel<8, M, 2><<<blocks, 256, lds, s>>>(p);
            return BRX_OK;
        }
        if (p.only) {
            // the reads whose trigger did not end in None: a reverse pass over them, 64 lanes a read as usual
            correct_kernel<64, M, 1><<<blocks, 256, lds, s>>>(p);
            return BRX_OK;
        }
    }
    if (lds > 64 * 1024) {
        const void *fn = G == 16 ? (const void *)correct_kernel<16, M>
                       : G == 32 ? (const void *)correct_kernel<32, M> : (const void *)correct_kernel<64, M>;
        BRX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (G <= 8 && M == BRX_ONE)
        correct_kernel<8, BRX_ONE><<<blocks, 256, lds, s>>>(p);
    else if (G == 4 && (M == BRX_GRAPH || M == BRX_GAP_SIZE))
        correct_kernel<4, (M == BRX_GRAPH || M == BRX_GAP_SIZE) ? M : BRX_GRAPH><<<blocks, 256, lds, s>>>(p);
    else if (G == 8 && (M == BRX_GRAPH || M == BRX_GAP_SIZE))
        correct_kernel<8, (M == BRX_GRAPH || M == BRX_GAP_SIZE) ? M : BRX_GRAPH><<<blocks, 256, lds, s>>>(p);
    else if (G <= 16)
        correct_kernel<16, M><<<blocks, 256, lds, s>>>(p);
    else if (G == 32)
        correct_kernel<32, M><<<blocks, 256, lds, s>>>(p);
    else
        correct_kernel<64, M><<<blocks, 256, lds, s>>>(p);
    return BRX_OK;
}

// Greedy needs LDS for the alignment back pointers: pick the narrowest group width that fits
int greedy_group(int G, int k, int max_search, uint32_t *dim, uint32_t *per_group)
{
    *dim = (uint32_t)(k + max_search + 2);
    *per_group = greedy_lds_bytes(*dim);
    for (; G <= 64; G *= 2)
        if ((uint64_t)(256 / G) * *per_group <= 160u * 1024u)
            return G;
    return 0;
}

int launch_pass(PassParams p, const brx_method_t &md, int G, hipStream_t s)
{
    static const char *names[5] = {"correct_pass", "correct_pass_two", "correct_pass_graph", "correct_pass_greedy",
                                   "correct_pass_gap_size"};
    const int method = md.method;
    size_t lds = 0;
    p.max_search = md.max_search;
    {
        const char *e = getenv("BRX_TUNE");
        p.flags = e ? (uint32_t)atoi(e) : 0u;
    }
    p.g_dim = 0;
    p.g_lds_bytes = 0;
    if (method == BRX_GREEDY) {
        G = greedy_group(G, p.k, md.max_search, &p.g_dim, &p.g_lds_bytes);
        if (!G) {
            set_error("greedy: k=%d with max_search=%d needs a %u x %u alignment table that does not fit LDS", p.k,
                      (int)md.max_search, p.g_dim, p.g_dim);
            return BRX_ERR_UNSUPPORTED;
        }
        lds = (size_t)(256 / G) * p.g_lds_bytes;
    }
    uint32_t blocks = pass_blocks(p.n_reads, G, method == BRX_ONE);
    if (p.only || p.trig) {
        // a list is a few reads and the grid loops over it (the open triggers are many: as many groups as the visited lists
        // the chain has sized allow); stay inside those lists (see launch_walk_list)
        const uint64_t sized_groups = sized_path_lists(p.n_reads);
        const uint32_t per_block = p.trig ? 256u / (uint32_t)verify_group() : 4u; // (narrow groups for the triggers, 64-lane groups for the reads)
        blocks = (uint32_t)(sized_groups / per_block);
        blocks = blocks < 1u ? 1u : (blocks > MAX_BLOCKS ? MAX_BLOCKS : blocks);
    }
    KernelTimer t(p.trig ? "rev_verify" : (p.only ? "rev_redo" : names[method]), s);
    switch (method) {
    case BRX_ONE: BRX_TRY(launch_method<BRX_ONE>(p, G, blocks, lds, s)); break;
    case BRX_TWO: BRX_TRY(launch_method<BRX_TWO>(p, G, blocks, lds, s)); break;
    case BRX_GRAPH: BRX_TRY(launch_method<BRX_GRAPH>(p, G, blocks, lds, s)); break;
    case BRX_GREEDY: BRX_TRY(launch_method<BRX_GREEDY>(p, G, blocks, lds, s)); break;
    case BRX_GAP_SIZE: BRX_TRY(launch_method<BRX_GAP_SIZE>(p, G, blocks, lds, s)); break;
    default: break;
    }
    BRX_HIP(hipGetLastError());
    return BRX_OK;
}

// BRX_REV_LEAN=0: reverse passes of Graph / GapSize through the 64-lane group kernel as before (A/B runs,
// and the fuzzer sweeps it)
bool rev_lean_on()
{
    const char *e = getenv("BRX_REV_LEAN");
    return !(e && *e == '0');
}

template <int M>
void launch_rev_scan(const PassParams &p, uint32_t *list, uint32_t *flag, TrigRec *trig, uint32_t trig_cap, uint32_t blocks, hipStream_t s)
{
    if (p.k == 19)
        rev_scan_kernel<19, M><<<blocks, 256, 0, s>>>(p, list, flag, trig, trig_cap);
    else if (p.k == 21)
        rev_scan_kernel<21, M><<<blocks, 256, 0, s>>>(p, list, flag, trig, trig_cap);
    else
        rev_scan_kernel<0, M><<<blocks, 256, 0, s>>>(p, list, flag, trig, trig_cap);
}

// a reverse pass in lean form (rev_scan_kernel), then the group kernel over the reads it handed back
int launch_rev_lean(brx_chain *ch, PassParams p, const brx_method_t &md, hipStream_t s)
{
    static const char *names[5] = {"correct_pass", "correct_pass_two", "correct_pass_graph", "correct_pass_greedy",
                                   "correct_pass_gap_size"};
    // room for four open triggers per read on average (the bench's data: 1.2); a scan that finds no room hands its read back
    const uint64_t trig_cap64 = 4ull * p.n_reads + 4096ull;
    const uint32_t trig_cap = trig_cap64 > 0x7fffffffull ? 0x7fffffffu : (uint32_t)trig_cap64;
    BRX_TRY(ensure((void **)&ch->d_rev_list, &ch->rev_list_bytes, (uint64_t)p.n_reads * 4ull));
    BRX_TRY(ensure((void **)&ch->d_rev_flag, &ch->rev_flag_bytes, (uint64_t)p.n_reads * 4ull));
    BRX_TRY(ensure(&ch->d_rev_trig, &ch->rev_trig_bytes, (uint64_t)trig_cap * sizeof(TrigRec)));
    BRX_HIP(hipMemsetAsync(p.ctrl + CTL_REV_HANDBACK, 0, 8, s));
    BRX_HIP(hipMemsetAsync(p.ctrl + CTL_REV_TRIGS, 0, 8, s));
    BRX_HIP(hipMemsetAsync(ch->d_rev_flag, 0, (uint64_t)p.n_reads * 4ull, s));
    TrigRec *trig = (TrigRec *)ch->d_rev_trig;
    {
        const uint32_t blocks = pass_blocks(p.n_reads, 64);
        KernelTimer t(names[md.method], s);
        switch (md.method) {
        case BRX_ONE: launch_rev_scan<BRX_ONE>(p, ch->d_rev_list, ch->d_rev_flag, trig, trig_cap, blocks, s); break;
        case BRX_GRAPH: launch_rev_scan<BRX_GRAPH>(p, ch->d_rev_list, ch->d_rev_flag, trig, trig_cap, blocks, s); break;
        case BRX_GAP_SIZE: launch_rev_scan<BRX_GAP_SIZE>(p, ch->d_rev_list, ch->d_rev_flag, trig, trig_cap, blocks, s); break;
        default: set_error("launch_rev_lean: method %u", md.method); return BRX_ERR_ARG;
        }
        BRX_HIP(hipGetLastError());
    }
    // the open triggers, one narrow group each
    PassParams v = p;
    v.trig = trig;
    v.trig_cap = trig_cap;
    v.only_n = p.ctrl + CTL_REV_TRIGS;
    v.redo_list = ch->d_rev_list;
    v.redo_flag = ch->d_rev_flag;
    BRX_HIP(hipMemsetAsync(p.ctrl + CTL_WORK, 0, 8, s));
    BRX_TRY(launch_pass(v, md, 16, s));
    // ... and the reads whose trigger did not end in None (or that found no room for a note), from scratch
    p.only = ch->d_rev_list;
    p.only_n = p.ctrl + CTL_REV_HANDBACK;
    BRX_HIP(hipMemsetAsync(p.ctrl + CTL_WORK, 0, 8, s));
    return launch_pass(p, md, 16, s);
}

} // namespace

extern "C" {

int brx_chain_new(const brx_set_t *set, const brx_method_t *methods, uint32_t n_methods, bool two_side,
                  brx_chain_t **out)
{
    if (!set || !out || (!methods && n_methods)) {
        set_error("null argument");
        return BRX_ERR_ARG;
    }
    for (uint32_t m = 0; m < n_methods; m++)
        if (methods[m].method > BRX_GAP_SIZE) {
            set_error("unknown correction method %u", methods[m].method);
            return BRX_ERR_ARG;
        }
    BRX_TRY(use_device(set->device));
    brx_chain *ch = new brx_chain();
    ch->set = set;
    ch->device = set->device;
    ch->methods.assign(methods, methods + n_methods);
    ch->two_side = two_side;
    ch->stream = nullptr;
    ch->d_stage[0] = ch->d_stage[1] = nullptr;
    ch->stage_bytes = 0;
    ch->d_lens[0] = ch->d_lens[1] = nullptr;
    ch->lens_cap = 0;
    ch->d_scan_tmp = nullptr;
    ch->scan_tmp_cap = 0;
    ch->d_ctrl = nullptr;
    ch->h_ctrl = nullptr;
    ch->d_path = nullptr;
    ch->path_bytes = 0;
    ch->d_in = nullptr;
    ch->d_in_cap = 0;
    ch->d_off = nullptr;
    ch->d_off_cap = 0;
    ch->d_out = nullptr;
    ch->d_out_cap = 0;
    ch->d_out_off = nullptr;
    ch->d_out_off_cap = 0;
    memset(ch->last_stats, 0, sizeof(ch->last_stats));
    if (const char *e = getenv("BRX_MAXPATH")) { // tests: a short visited list, so that walks outgrow it
        const int v = atoi(e);
        if (v >= 1 && v <= (1 << 20))
            ch->maxpath_seen = (uint32_t)v;
    }
    hipError_t e = hipStreamCreateWithFlags(&ch->stream, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipMalloc((void **)&ch->d_ctrl, CTL_N * 8);
    if (e == hipSuccess)
        e = hipHostMalloc((void **)&ch->h_ctrl, CTL_N * 8, hipHostMallocDefault);
    if (e != hipSuccess) {
        set_error("chain alloc: %s", hipGetErrorString(e));
        brx_chain_free(ch);
        return BRX_ERR_HIP;
    }
    *out = ch;
    return BRX_OK;
}

// A handful of reads that outgrew their workspace -- a graph walk longer than the visited list (GapSize at BASELINE
// configs[4]'s per-GPU share: 18 of 625 000) or a correction that grows the read beyond its output slot -- should not
// cost the whole batch a second run.  Everything else of the batch is final and sits in its own slot of the last
// staging buffer, so the poisoned reads are taken out as a small batch of their own and corrected by a second chain
// with a longer list / more slack (same set, methods and direction rule; all on the GPU).  A redone read that fits its
// slot is written back into it; one that does not goes to `side` and is copied over its place in the compact output
// after the compaction (its final length is entered in `lens`, so every other read lands where it should).
// BRX_OK: lens / slots patched, the batch is complete.  BRX_ERR_UNSUPPORTED: not applicable (too many reads) -- the
// caller redoes the whole batch with a larger workspace as before.
struct SideRead {
    uint32_t r;
    std::vector<uint8_t> bytes;
};
static uint32_t redo_max_reads()
{
    const char *e = getenv("BRX_REDO_MAX"); // tests: 0 = never (read on every use: only overflowing batches get here)
    return e ? (uint32_t)atoi(e) : 4096u;
}
static int redo_poisoned_reads(brx_chain *ch, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n_reads,
                               uint8_t *d_stage, uint32_t *d_lens, int stage_reversed, uint32_t slack, uint32_t sub_slack,
                               uint32_t maxpath, std::vector<SideRead> &side, hipStream_t s)
{
    if (ch->is_sub)
        return BRX_ERR_UNSUPPORTED;
    std::vector<uint32_t> lens(n_reads);
    std::vector<uint64_t> offs((size_t)n_reads + 1);
    BRX_HIP(hipMemcpyAsync(lens.data(), d_lens, (size_t)n_reads * 4, hipMemcpyDeviceToHost, s));
    BRX_HIP(hipMemcpyAsync(offs.data(), d_offsets, ((size_t)n_reads + 1) * 8, hipMemcpyDeviceToHost, s));
    BRX_HIP(hipStreamSynchronize(s));
    std::vector<uint32_t> ids;
    for (uint32_t r = 0; r < n_reads; r++)
        if (lens[r] == 0xffffffffu) {
            if (ids.size() >= redo_max_reads())
                return BRX_ERR_UNSUPPORTED;
            ids.push_back(r);
        }
    if (ids.empty())
        return BRX_ERR_UNSUPPORTED;
    std::vector<uint64_t> moff(ids.size() + 1, 0);
    for (size_t j = 0; j < ids.size(); j++)
        moff[j + 1] = moff[j] + (offs[ids[j] + 1] - offs[ids[j]]);
    std::vector<uint8_t> mb(moff.back() ? moff.back() : 1);
    for (size_t j = 0; j < ids.size(); j++)
        if (moff[j + 1] > moff[j])
            BRX_HIP(hipMemcpyAsync(mb.data() + moff[j], d_bases + offs[ids[j]], moff[j + 1] - moff[j], hipMemcpyDeviceToHost, s));
    BRX_HIP(hipStreamSynchronize(s));
    if (!ch->sub) {
        brx_chain *sub = nullptr;
        BRX_TRY(brx_chain_new(ch->set, ch->methods.data(), (uint32_t)ch->methods.size(), ch->two_side, &sub));
        sub->is_sub = true;
        ch->sub = sub;
    }
    if (ch->sub->maxpath_seen < maxpath)
        ch->sub->maxpath_seen = maxpath;
    if (ch->sub->slack_seen < sub_slack)
        ch->sub->slack_seen = sub_slack;
    uint8_t *ob = nullptr;
    uint64_t *oo = nullptr;
    BRX_TRY(brx_chain_correct_batch(ch->sub, mb.data(), moff.data(), (uint32_t)ids.size(), &ob, &oo));
    int st = BRX_OK;
    std::vector<uint8_t> rev;
    for (size_t j = 0; j < ids.size() && st == BRX_OK; j++) {
        const uint32_t r = ids[j];
        const uint64_t s0 = slot_of(offs[r], r, slack), s1 = slot_of(offs[r + 1], (uint64_t)r + 1, slack);
        const uint64_t len = oo[j + 1] - oo[j];
        if (len >= 0xffffffffull) {
            st = BRX_ERR_UNSUPPORTED;
            break;
        }
        const uint8_t *src = ob + oo[j];
        if (len + 1 > s1 - s0) {
            // does not fit its slot: keep the bytes aside (forward order), enter the length, patch after compaction
            side.push_back(SideRead{r, std::vector<uint8_t>(src, src + len)});
            const uint32_t l32s = (uint32_t)len;
            if (hipMemcpy(d_lens + r, &l32s, 4, hipMemcpyHostToDevice) != hipSuccess) {
                set_error("redo of poisoned reads: length patch failed");
                st = BRX_ERR_HIP;
            }
            continue;
        }
        if (stage_reversed) { // the last pass stored its reads back to front
            rev.assign(src, src + len);
            std::reverse(rev.begin(), rev.end());
            src = rev.data();
        }
        const uint32_t l32 = (uint32_t)len;
        hipError_t e = len ? hipMemcpy(d_stage + s0, src, len, hipMemcpyHostToDevice) : hipSuccess;
        if (e == hipSuccess)
            e = hipMemcpy(d_lens + r, &l32, 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("redo of poisoned reads: %s", hipGetErrorString(e));
            st = BRX_ERR_HIP;
        }
    }
    brx_buf_free(ob);
    brx_buf_free(oo);
    return st;
}

// the batch entry proper; the caller holds ch->mu (a chain owns ONE workspace: staging buffers, control block, visited
// lists -- two batches at once on one chain would share them)
static int correct_batch_device_locked(brx_chain_t *ch, const uint8_t *d_bases, const uint64_t *d_offsets,
                                       uint32_t n_reads, uint64_t total_bases, uint8_t *d_out, uint64_t out_cap,
                                       uint64_t *d_out_offsets, uint64_t *out_total, void *stream)
{
    hipStream_t s = (hipStream_t)stream; // nullptr = the legacy default stream
    *out_total = 0;
    if (n_reads == 0) {
        BRX_HIP(hipMemsetAsync(d_out_offsets, 0, 8, s));
        BRX_HIP(hipStreamSynchronize(s));
        return BRX_OK;
    }
    const int k = ch->set->k;
    if ((k & 1) == 0) {
        set_error("correction needs an odd k (parity-canonical set), got %d", k);
        return BRX_ERR_ARG;
    }

    // (lens, scan scratch) do not depend on slack
    {
        uint64_t cap_b = ch->lens_cap * 4;
        const uint64_t need = (uint64_t)n_reads * 4;
        if (need > cap_b || !ch->d_lens[0]) {
            for (int q = 0; q < 2; q++) {
                if (ch->d_lens[q])
                    (void)hipFree(ch->d_lens[q]);
                ch->d_lens[q] = nullptr;
            }
            const uint64_t want = (uint64_t)n_reads + n_reads / 8 + 64;
            BRX_HIP(hipMalloc((void **)&ch->d_lens[0], want * 4));
            BRX_HIP(hipMalloc((void **)&ch->d_lens[1], want * 4));
            ch->lens_cap = want;
        }
        uint64_t tmp_bytes = ch->scan_tmp_cap;
        BRX_TRY(ensure((void **)&ch->d_scan_tmp, &tmp_bytes, scan_tmp_bytes(n_reads)));
        ch->scan_tmp_cap = tmp_bytes;
    }

    // walks (Two/Graph/Greedy/GapSize forward passes) are faster on the bit vector: materialise a lazy one now
    if (ch->set->bits_stale && !ch->set->sparse)
        for (const brx_method_t &md : ch->methods)
            if (md.method != BRX_ONE) {
                BRX_TRY(ensure_bits(ch->set, s, "correction"));
                break;
            }
    // probe index of the set (built here when the set has none yet, e.g. after a finish_into)
    BRX_TRY(index_ensure(ch->set, s));
    IdxView idx{nullptr, 0, 0, 0};
    if (ch->set->idx_valid && (index_wanted(k) || no_bits(ch->set)))
        idx = IdxView{ch->set->d_lines, 32u - ch->set->idx_log_lines, ch->set->idx_m, (uint32_t)k - ch->set->idx_m + 1u,
                      ch->set->idx_linebits ? (const uint32_t *)(ch->set->d_lines + (8ull << ch->set->idx_log_lines)) : nullptr};
    if (const char *e = getenv("BRX_LINE_BITS")) // (A/B and fuzzers: 0 = do not consult the occupancy bits)
        if (*e == '0')
            idx.line_bits = nullptr;

    const int n_dirs = ch->two_side ? 1 : 2;
    const int n_methods = (int)ch->methods.size();
    uint64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<SideRead> side; // redone reads that do not fit their staging slot (redo_poisoned_reads)
    const int G = group_width(false, idx.lines != nullptr, n_reads); // forward width (8 exists for One only, else 16)
    bool needs_path = false;
    for (int m = 0; m < n_methods; m++)
        needs_path |= (ch->methods[m].method == BRX_GRAPH || ch->methods[m].method == BRX_GAP_SIZE ||
                       ch->methods[m].method == BRX_GREEDY);
    uint32_t maxpath = ch->maxpath_seen; // visited-list capacity per group (4096: 1 GiB of scratch at 32768 groups); grows x8 on overflow

    for (uint32_t slack = ch->slack_seen, attempt = 0;; attempt++) {
        if (attempt > 12) {
            set_error("correction output / graph walks do not fit the workspace after 12 enlargements; giving up");
            return BRX_ERR_OVERFLOW;
        }
        if (needs_path) {
            // one visited list per group that can be resident: walking methods run 16-lane groups or wider, so a grid of
            // pass_blocks(n_reads, 16) blocks of 16 groups bounds every width (a handful of redone reads needs a
            // handful of lists, not 32768 of them)
            const uint64_t n_groups = sized_path_lists(n_reads); // (at least one block of the narrowest list / verify kernel)
            BRX_TRY(ensure((void **)&ch->d_path, &ch->path_bytes, n_groups * maxpath * 8ull));
        }
        const uint64_t stage_need = total_bases + (total_bases >> 2) * slack + 64ull * ((uint64_t)n_reads + 1) + 64;
        if (stage_need > ch->stage_bytes || !ch->d_stage[0]) {
            for (int q = 0; q < 2; q++) {
                if (ch->d_stage[q])
                    (void)hipFree(ch->d_stage[q]);
                ch->d_stage[q] = nullptr;
            }
            ch->stage_bytes = 0;
            BRX_HIP(hipMalloc((void **)&ch->d_stage[0], stage_need));
            BRX_HIP(hipMalloc((void **)&ch->d_stage[1], stage_need));
            ch->stage_bytes = stage_need;
        }

        // the chain: src/lib.rs:44-55.  buffer "rev" = stored back to front w.r.t. the original read
        const uint8_t *cur = d_bases;
        const uint32_t *cur_lens = nullptr;
        int cur_staged = 0, cur_rev = 0, pp = 0;
        bool any_overflow = false;
        BRX_HIP(hipMemsetAsync(ch->d_ctrl, 0, CTL_N * 8, s));
        for (int dir = 0; dir < n_dirs && !any_overflow; dir++) {
            for (int m = 0; m < n_methods; m++) {
                PassParams p;
                p.bits = no_bits(ch->set) ? nullptr : ch->set->d_bits;
                // the index pays where whole groups probe neighbouring k-mers: One's passes and the (nearly
                // trigger-free) reverse scans of every method; walks probe 4 successors of one k-mer per round and
                // gain nothing from shared lines.
                const int mth = ch->methods[m].method;
                // BRX_INDEX_FWD: bit mask (1 << method) of the methods whose FORWARD pass probes the index; measured per
                // method (tools/method_bench.py, 1 Gbp): One 1.4x faster through it; Graph / GapSize 10-12 % faster with
                // 8-lane groups (walk_group()); Two / Greedy (16-lane groups) see profiles/r2_one_kernel_ab.txt
                const char *e_fwd = getenv("BRX_INDEX_FWD");
                const unsigned idx_fwd = e_fwd ? (unsigned)atoi(e_fwd) : 29u; // One, Graph, Greedy, GapSize (Two: 69 ms through the bit vector, 72 through the index)
                const bool use_idx = no_bits(ch->set) || ((idx_fwd >> mth) & 1u) || (dir == 1 && mth != BRX_GREEDY);
                p.idx = use_idx ? idx : IdxView{nullptr, 0, 0, 0};
                p.k = k;
                p.c = ch->methods[m].confirm;
                p.n_reads = n_reads;
                p.offsets = d_offsets;
                p.in = cur;
                p.in_lens = cur_lens;
                p.in_staged = cur_staged;
                p.flip = (cur_rev != dir);
                p.out = ch->d_stage[pp];
                p.out_lens = ch->d_lens[pp];
                p.slack = slack;
                p.ctrl = (unsigned long long *)ch->d_ctrl;
                p.path_k = ch->d_path;
                p.maxpath = maxpath;
                BRX_HIP(hipMemsetAsync(ch->d_ctrl + CTL_WORK, 0, 8, s));
                p.flags = 0;
                p.max_search = 0;
                p.g_dim = 0;
                p.g_lds_bytes = 0;
                // BRX_LANE_REV (bit 0: Graph, bit 1: GapSize; default: by index size): REVERSE passes in lane form too.
                // Their scan meets hardly a solid k-mer (the read is read back to front, not complemented), so over the
                // solidity mask it is bit runs of sixteen positions a round -- but the rare trigger runs error_len to the
                // END OF THE READ (hundreds of rounds of one lane, whose wave waits for it), and GapSize then walks a gap
                // of thousands of fixed steps along the genome (a deep-coverage set has long unbranched paths), which no
                // unit's edit list holds: the lanes hand the read back.  Measured (profiles/r4c_*, r4e_*):
                //   2^25 lines (configs[1]'s set)     graph 65.6 -> 75.4 ms per Gbp fwd+rev, gap_size 92 -> 211 (1 845 reads redone)
                //   2^28 lines (configs[4]'s share)   graph + gap_size 714 -> 1 598 ms per step with both (770 ms of redo)
                //   2^29 lines, 8x coverage, -a 1     graph + gap_size 932 -> 748 ms per 8 Gbp (walks die young in a set
                //                                     with holes: nothing is handed back)
                //   2^28 lines (configs[4]'s share)   Graph alone in lane form: 836 -> 877 ms per step (r4f)
                // Only the last case gains (its 64-lane group kernels take ~180 ms per 8 Gbp pass at 2^29 lines), and a
                // deep-coverage set of that size would hand reads back like the second.  So the form is OFF unless asked
                // for; the switch stays for measurements and for the fuzzer, which sweeps it.
                const unsigned lane_rev_default = 0u;
                const char *e_rev = getenv("BRX_LANE_REV");
                const unsigned lane_rev = e_rev && *e_rev ? (unsigned)atoi(e_rev) : lane_rev_default;
                const bool lane_dir = dir == 0 ? !p.flip
                                               : ((mth == BRX_GRAPH && (lane_rev & 1u)) || (mth == BRX_GAP_SIZE && (lane_rev & 2u)));
                if ((mth == BRX_ONE || mth == BRX_GRAPH || mth == BRX_GAP_SIZE) && lane_dir) {
                    // forward passes of One, Graph and GapSize (and the reverse ones of the latter two): one lane per chunk
                    // of a read (brx_onelane.hip) where that form applies
                    const LanePassInfo info{cur_staged ? ch->stage_bytes : total_bases, mth};
                    const int lst = lane_pass(ch, p, info, s);
                    if (lst != BRX_OK && lst != BRX_ERR_UNSUPPORTED)
                        return lst;
                    if (lst == BRX_OK) {
                        cur = ch->d_stage[pp];
                        cur_lens = ch->d_lens[pp];
                        cur_staged = 1;
                        cur_rev = dir;
                        pp ^= 1;
                        continue;
                    }
                }
                int gw = dir ? group_width(true, false, n_reads) : G;
                if (gw < 16 && mth != BRX_ONE) {
                    // the walking methods spend most of their rounds on one walk step = 4 probes: narrow groups keep the
                    // lanes busy there (BRX_GROUP_WALK, default below); Two and Greedy need 16 lanes for their stage-1 probes
                    gw = (mth == BRX_GRAPH || mth == BRX_GAP_SIZE) ? walk_group() : 16;
                }
                // (Two's and Greedy's reverse passes cost 4 ms per Gbp of their 55 / 69 in the group kernel: the lean form
                // was measured for them, too, and bought nothing -- profiles/r4o_rev_lean_ab.txt)
                // One's reverse pass in the same form: BRX_REV_LEAN_ONE=1 (measured: profiles/r4o_rev_lean_ab.txt)
                const char *e_lo = getenv("BRX_REV_LEAN_ONE");
                const bool lean_one = mth == BRX_ONE && e_lo && *e_lo == '1';
                const bool lean = dir == 1 && (mth == BRX_GRAPH || mth == BRX_GAP_SIZE || lean_one) && gw == 64 && rev_lean_on();
                if (lean)
                    BRX_TRY(launch_rev_lean(ch, p, ch->methods[m], s));
                else
                    BRX_TRY(launch_pass(p, ch->methods[m], gw, s));
                cur = ch->d_stage[pp];
                cur_lens = ch->d_lens[pp];
                cur_staged = 1;
                cur_rev = dir;
                pp ^= 1;
            }
        }
        if (n_methods == 0) {
            // no corrector: identity (a reverse of a reverse).  Copy through one staged pass-less path.
            set_error("empty method list");
            return BRX_ERR_ARG;
        }
        static const bool tr_attempts = [] { const char *e = getenv("BRX_TRACE"); return e && *e == '1'; }();
        if (tr_attempts)
            fprintf(stderr, "[brx correct] attempt %u launched: slack %u maxpath %u n_reads %u\n", attempt, slack, maxpath, n_reads);
        BRX_HIP(hipMemcpyAsync(ch->h_ctrl, ch->d_ctrl, CTL_N * 8, hipMemcpyDeviceToHost, s));
        BRX_HIP(hipStreamSynchronize(s));
        if (tr_attempts)
            fprintf(stderr, "[brx correct] attempt %u done: slot overflows %llu, walk list overflows %llu, nonterminating %llu, "
                            "lean reverse passes: %llu open triggers verified, %llu reads redone\n", attempt,
                    (unsigned long long)ch->h_ctrl[CTL_OVERFLOW], (unsigned long long)ch->h_ctrl[CTL_PATHOVF],
                    (unsigned long long)ch->h_ctrl[CTL_NONTERM], (unsigned long long)ch->h_ctrl[CTL_REV_TRIGS_SUM],
                    (unsigned long long)ch->h_ctrl[CTL_REV_HANDBACK_SUM]);
        stats[4] = attempt;
        if (ch->h_ctrl[CTL_NONTERM] != 0) {
            set_error("%llu read(s): the scan does not terminate (e.g. greedy moving the read cursor backwards for ever; "
                      "the reference spins on such input)", (unsigned long long)ch->h_ctrl[CTL_NONTERM]);
            return BRX_ERR_UNSUPPORTED;
        }
        if (ch->h_ctrl[CTL_OVERFLOW] != 0 || ch->h_ctrl[CTL_PATHOVF] != 0) {
            // some read outgrew its output slot / a graph walk outgrew its visited list: redo the
            // batch on the GPU with a larger workspace (never on the CPU)
            stats[5] += ch->h_ctrl[CTL_OVERFLOW]; // reads that outgrew their slot / walks that outgrew the list,
            stats[6] += ch->h_ctrl[CTL_PATHOVF];  // summed over the attempts that were thrown away or patched
            bool patched = false;
            if (ch->h_ctrl[CTL_OVERFLOW] + ch->h_ctrl[CTL_PATHOVF] <= redo_max_reads()) {
                side.clear();
                const int rst = redo_poisoned_reads(ch, d_bases, d_offsets, n_reads, const_cast<uint8_t *>(cur),
                                                    const_cast<uint32_t *>(cur_lens), cur_rev, slack,
                                                    ch->h_ctrl[CTL_OVERFLOW] ? slack * 4u : slack,
                                                    ch->h_ctrl[CTL_PATHOVF] ? maxpath * 8u : maxpath, side, s);
                if (rst == BRX_OK) {
                    // the few long walks were redone by the sub chain with its own longer list; this chain keeps its
                    // list size (growing it x8 for every group of the grid is 8 GiB per chain after one event)
                    patched = true;
                    if (ch->sub) // events of the redone reads, which the first attempt gave up on
                        for (int q = 0; q < 4; q++)
                            stats[q] += ch->sub->last_stats[q];
                }
                else if (rst != BRX_ERR_UNSUPPORTED)
                    return rst;
            }
            if (!patched) {
                side.clear();
                if (ch->h_ctrl[CTL_OVERFLOW] != 0)
                    slack *= 4;
                if (ch->h_ctrl[CTL_PATHOVF] != 0)
                    maxpath *= 8;
                ch->slack_seen = slack;
                ch->maxpath_seen = maxpath;
                continue;
            }
        }
        {
            const uint64_t unw = ch->h_ctrl[CTL_LANE_UNWRITTEN] + (ch->sub ? ch->sub->last_stats[7] >> 56 : 0);
            stats[7] = (ch->h_ctrl[CTL_LANE_UNITS] & 0xffffffffull) | ((ch->h_ctrl[CTL_LANE_FAIL] & 0xffffffull) << 32) |
                       ((unw > 255 ? 255ull : unw) << 56);
            if (unw) // an invariant of the lane form is broken (the output is still right: those reads went to the group kernel)
                fprintf(stderr, "[brx correct] WARNING: %llu unit record(s) were never written by the lane pass\n", (unsigned long long)unw);
        }
        stats[0] += ch->h_ctrl[CTL_ROUNDS];
        stats[1] += ch->h_ctrl[CTL_PROBES];
        stats[2] += ch->h_ctrl[CTL_TRIGGERS];
        stats[3] += ch->h_ctrl[CTL_FIXES];

        // out_offsets = exclusive scan of final lengths
        BRX_TRY(exclusive_scan_lens(cur_lens, n_reads, ch->d_scan_tmp, d_out_offsets,
                                    (unsigned long long *)(ch->d_ctrl + CTL_TOTAL), s));
        BRX_HIP(hipMemcpyAsync(ch->h_ctrl, ch->d_ctrl, CTL_N * 8, hipMemcpyDeviceToHost, s));
        BRX_HIP(hipStreamSynchronize(s));
        const uint64_t total = ch->h_ctrl[CTL_TOTAL];
        *out_total = total;
        memcpy(ch->last_stats, stats, sizeof(stats));
        if (total > out_cap) {
            set_error("corrected batch needs %llu bytes, output buffer has %llu", (unsigned long long)total,
                      (unsigned long long)out_cap);
            return BRX_ERR_OVERFLOW;
        }
        {
            KernelTimer t("compact", s);
            const uint32_t grid = n_reads < (1u << 20) ? n_reads : (1u << 20);
            compact_kernel<<<grid, 256, 0, s>>>(cur, cur_lens, d_offsets, n_reads, slack, cur_rev, d_out_offsets, d_out);
        }
        BRX_HIP(hipStreamSynchronize(s));
        for (const SideRead &sr : side) { // the few reads that outgrew their slot: their bytes go straight to their place
            uint64_t at = 0;
            BRX_HIP(hipMemcpy(&at, d_out_offsets + sr.r, 8, hipMemcpyDeviceToHost));
            if (!sr.bytes.empty())
                BRX_HIP(hipMemcpy(d_out + at, sr.bytes.data(), sr.bytes.size(), hipMemcpyHostToDevice));
        }
        return BRX_OK;
    }
}

int brx_chain_correct_batch_device(brx_chain_t *ch, const uint8_t *d_bases, const uint64_t *d_offsets,
                                   uint32_t n_reads, uint64_t total_bases, uint8_t *d_out, uint64_t out_cap,
                                   uint64_t *d_out_offsets, uint64_t *out_total, void *stream)
{
    if (!ch || !d_offsets || !d_out_offsets || !out_total || (!d_bases && total_bases) || (!d_out && out_cap)) {
        set_error("null argument");
        return BRX_ERR_ARG;
    }
    BRX_TRY(use_device(ch->device));
    std::lock_guard<std::mutex> g(ch->mu);
    return correct_batch_device_locked(ch, d_bases, d_offsets, n_reads, total_bases, d_out, out_cap, d_out_offsets, out_total,
                                       stream);
}

int brx_chain_correct_batch(brx_chain_t *ch, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                            uint8_t **out_bases, uint64_t **out_offsets)
{
    if (!ch || !offsets || !out_bases || !out_offsets || (!bases && n_reads)) {
        set_error("null argument");
        return BRX_ERR_ARG;
    }
    *out_bases = nullptr;
    *out_offsets = nullptr;
    BRX_TRY(use_device(ch->device));
    // BRX_TRACE=2: host wall time of the call's stages on stderr (where a batch from host memory spends its time)
    static const bool tr_host = [] { const char *e = getenv("BRX_TRACE"); return e && *e == '2'; }();
    const auto t_in = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    // ONE lock from the upload to the download: the chain's own d_in / d_off / d_out staging is part of the workspace
    // a concurrent call on the same chain would overwrite (include/brx.h: "a chain serialises concurrent calls")
    std::lock_guard<std::mutex> g(ch->mu);
    uint64_t total = 0;
    {
        // The batch goes up at PCIe speed only from page-locked memory.  A caller that filled a brx_host_alloc buffer is
        // copied from directly; anything else is first moved into the chain's own page-locked block by four threads
        // (the runtime's built-in staging of pageable memory managed 6-10 GB/s: 8-13 ms of a 82 MB batch).
        const uint64_t base0 = offsets[0], tot = offsets[n_reads] - base0;
        const uint8_t *src = bases + base0;
        if (tot >= (4u << 20) && !host_buf_is_pinned(src)) {
            if (ch->h_in_cap < tot) {
                host_buf_release(ch->h_in);
                ch->h_in = (uint8_t *)host_buf_acquire(tot + tot / 8);
                ch->h_in_cap = ch->h_in ? tot + tot / 8 : 0;
            }
            if (ch->h_in) {
                const int nt = 4;
                std::vector<std::thread> th;
                for (int t = 1; t < nt; t++)
                    th.emplace_back([&, t] { memcpy(ch->h_in + tot * t / nt, src + tot * t / nt, tot * (t + 1) / nt - tot * t / nt); });
                memcpy(ch->h_in, src, tot / nt);
                for (auto &x : th)
                    x.join();
                src = ch->h_in;
            }
        }
        // (upload_batch takes `bases` as the start of the stream offsets[] index into)
        BRX_TRY(upload_batch(src - base0, offsets, n_reads, &ch->d_in, &ch->d_in_cap, &ch->d_off, &ch->d_off_cap, &total, ch->stream));
    }
    {
        uint64_t cap_b = ch->d_out_off_cap * 8;
        BRX_TRY(ensure((void **)&ch->d_out_off, &cap_b, ((uint64_t)n_reads + 1) * 8));
        ch->d_out_off_cap = cap_b / 8;
    }
    const double t_up = ms_since(t_in);
    uint64_t out_total = 0;
    uint64_t want = total + total / 16 + 4096;
    for (int attempt = 0; attempt < 3; attempt++) {
        BRX_TRY(ensure((void **)&ch->d_out, &ch->d_out_cap, want));
        int st = correct_batch_device_locked(ch, ch->d_in, ch->d_off, n_reads, total, ch->d_out, ch->d_out_cap, ch->d_out_off,
                                             &out_total, ch->stream);
        if (st == BRX_ERR_OVERFLOW && out_total > ch->d_out_cap) {
            want = out_total + 64;
            continue;
        }
        if (st != BRX_OK)
            return st;
        break;
    }
    const double t_dev = ms_since(t_in);
    // the corrected batch comes back into pooled page-locked blocks (released by brx_buf_free like any other)
    uint8_t *hb = (uint8_t *)host_buf_acquire(out_total ? out_total : 1);
    uint64_t *ho = (uint64_t *)host_buf_acquire(((size_t)n_reads + 1) * 8);
    if (!hb || !ho) {
        host_buf_release(hb);
        host_buf_release(ho);
        set_error("host malloc failed");
        return BRX_ERR_NOMEM;
    }
    hipError_t e = hipSuccess;
    if (out_total)
        e = hipMemcpyAsync(hb, ch->d_out, out_total, hipMemcpyDeviceToHost, ch->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(ho, ch->d_out_off, ((size_t)n_reads + 1) * 8, hipMemcpyDeviceToHost, ch->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ch->stream);
    if (e != hipSuccess) {
        host_buf_release(hb);
        host_buf_release(ho);
        set_error("D2H: %s", hipGetErrorString(e));
        return BRX_ERR_HIP;
    }
    *out_bases = hb;
    *out_offsets = ho;
    if (tr_host)
        fprintf(stderr, "[brx batch %p] %u reads %llu bases: upload %.2f ms, passes %.2f, download %.2f (at %.2f ms of the clock)\n", (void *)ch,
                n_reads, (unsigned long long)total, t_up, t_dev - t_up, ms_since(t_in) - t_dev,
                std::chrono::duration<double, std::milli>(t_in.time_since_epoch()).count());
    return BRX_OK;
}

// ---- one batch in flight per chain -----------------------------------------------------------------------------------
// brx_chain_correct_batch is upload -> passes -> download, back to back on one stream: 82 MB up, ~3 ms of kernels,
// 82 MB down per 8192-record batch, nothing overlapping (9 Gbases/s from page-locked memory against 36 for the passes
// alone).  The overlap a host needs is BETWEEN batches -- the next one's upload under this one's kernels under the last
// one's download -- and chains are the unit that owns a stream and a workspace.  So the asynchronous form is per chain:
// _async starts the batch on a thread of the library and returns, _wait hands out its result; a host keeps two or three
// chains of the same set busy in turn from ONE thread (INTEGRATION.md has the loop run_correction would use).
namespace {
struct AsyncJob {
    std::thread th;
    int status = BRX_OK;
    uint8_t *ob = nullptr;
    uint64_t *oo = nullptr;
    std::string err;
};
} // namespace

int brx_chain_correct_batch_async(brx_chain_t *ch, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads)
{
    if (!ch || !offsets || (!bases && n_reads)) {
        set_error("null argument");
        return BRX_ERR_ARG;
    }
    if (ch->async_job) {
        set_error("chain already has a batch in flight: brx_chain_correct_batch_wait first (one per chain; use several chains)");
        return BRX_ERR_ARG;
    }
    AsyncJob *job = new AsyncJob();
    ch->async_job = job;
    job->th = std::thread([ch, job, bases, offsets, n_reads] {
        job->status = brx_chain_correct_batch(ch, bases, offsets, n_reads, &job->ob, &job->oo);
        if (job->status != BRX_OK)
            job->err = brx_last_error(); // (the message lives in this thread: carried over to the waiter)
    });
    return BRX_OK;
}

int brx_chain_correct_batch_wait(brx_chain_t *ch, uint8_t **out_bases, uint64_t **out_offsets)
{
    if (!ch || !out_bases || !out_offsets) {
        set_error("null argument");
        return BRX_ERR_ARG;
    }
    *out_bases = nullptr;
    *out_offsets = nullptr;
    AsyncJob *job = (AsyncJob *)ch->async_job;
    if (!job) {
        set_error("no batch in flight on this chain");
        return BRX_ERR_ARG;
    }
    job->th.join();
    ch->async_job = nullptr;
    const int st = job->status;
    if (st == BRX_OK) {
        *out_bases = job->ob;
        *out_offsets = job->oo;
    } else {
        set_error("%s", job->err.c_str());
    }
    delete job;
    return st;
}

int brx_chain_last_stats(const brx_chain_t *ch, uint64_t *stats8)
{
    if (!ch || !stats8)
        return BRX_ERR_ARG;
    memcpy(stats8, ch->last_stats, sizeof(ch->last_stats));
    return BRX_OK;
}

void brx_chain_free(brx_chain_t *ch)
{
    if (!ch)
        return;
    if (ch->async_job) { // a batch still in flight: let it finish, drop its result
        uint8_t *ob = nullptr;
        uint64_t *oo = nullptr;
        (void)brx_chain_correct_batch_wait(ch, &ob, &oo);
        brx_buf_free(ob);
        brx_buf_free(oo);
    }
    if (ch->sub)
        brx_chain_free(ch->sub);
    ch->sub = nullptr;
    if (use_device(ch->device) == BRX_OK)
        lane_ws_free(ch);
    host_buf_release(ch->h_in);
    ch->h_in = nullptr;
    if (use_device(ch->device) == BRX_OK) {
        for (int q = 0; q < 2; q++) {
            if (ch->d_stage[q])
                (void)hipFree(ch->d_stage[q]);
            if (ch->d_lens[q])
                (void)hipFree(ch->d_lens[q]);
        }
        if (ch->d_scan_tmp)
            (void)hipFree(ch->d_scan_tmp);
        if (ch->d_ctrl)
            (void)hipFree(ch->d_ctrl);
        if (ch->d_path)
            (void)hipFree(ch->d_path);
        if (ch->d_rev_list)
            (void)hipFree(ch->d_rev_list);
        if (ch->d_rev_flag)
            (void)hipFree(ch->d_rev_flag);
        if (ch->d_rev_trig)
            (void)hipFree(ch->d_rev_trig);
        if (ch->h_ctrl)
            (void)hipHostFree(ch->h_ctrl);
        if (ch->d_in)
            (void)hipFree(ch->d_in);
        if (ch->d_off)
            (void)hipFree(ch->d_off);
        if (ch->d_out)
            (void)hipFree(ch->d_out);
        if (ch->d_out_off)
            (void)hipFree(ch->d_out_off);
        if (ch->stream)
            (void)hipStreamDestroy(ch->stream);
    }
    delete ch;
}

} // extern "C"
