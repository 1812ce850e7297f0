// correct::One, forward pass, ONE LANE PER CHUNK OF A READ.
//
// Reference: Corrector::correct (src/correct/mod.rs:53-107), Exist<ScenarioOne>::correct_error
// (src/correct/exist/mod.rs:112-150, get_score :21-47, one_more :49-70), ScenarioOne (src/correct/exist/one.rs:33-74).
//
// Why.  The group kernel (brx_correct.hip: one_kernel, 8 lanes per read) is instruction-issue-bound: ~440 vector
// instructions per wave-round with 29 of 64 lanes live, because the eight groups of a wave sit in different states and
// every wave runs the union of the states' code every round (profiles/r2h_sq_summary.json).  Here every lane is its own
// scalar state machine over its own stretch of a read, written as ONE straight-line round -- build the k-mer this state
// asks about, one KmerSet::get, one table-like transition, at most one fix out -- so that all 64 lanes of a wave do
// useful work in every instruction whatever states they are in.
//
// Where the parallelism comes from.  A scan is sequential inside a read (i, kmer, previous are loop carried), and 1e5
// reads are far fewer than the ~4.6e5 lanes the chip keeps resident.  But One's loop-top state is (i, kmer) alone --
// `previous` always equals get(kmer): after mod.rs:99 by definition, after a fix the k-mer is the corrected one that
// alt_nucs just found solid, after a failed fix it is the trigger k-mer, not solid -- so the scan of a read can be cut
// wherever that state can be predicted.  A read is cut into UNITS at sync points: positions q behind R consecutive
// solid ORIGINAL k-mers (lane_sync_kernel finds the first one after every nominal chunk boundary).  A scan that walks
// into such a run has, with all but negligible probability, kmer == the original k-mer at q and arrives at q exactly;
// unit u+1 therefore starts at (q, original k-mer) speculatively while unit u is still running, and unit u CHECKS the
// prediction when it gets there.  If its state at q is the predicted one, its output ends at q and unit u+1's begins
// there: the concatenation is byte for byte the sequential scan.  If not (it jumped over q, or carries a corrected
// base), it keeps scanning through the next unit's stretch, writing a second list, and checks again at the sync point
// after that; three misses in a row hand the read back to the group kernel.  No unit ever waits for another one: all
// coupling is resolved afterwards by lane_apply_kernel, which walks the chain of units of a read (which unit's output
// is the truth up to where) and writes the read's ordinary staging slot, so that everything downstream (reverse
// pass, other methods, compaction, the redo of overflowing reads) is unchanged.
//
// What a lane touches.  458 752 resident lanes that each stream their own bytes thrash every cache level (first form of
// this file: 70 ms; every round waited for an input line, an output line and an index line of its own).  So a lane
// reads its stretch as 2-BIT CODES, 16 bases per dword, from a packed copy of the batch (lane_pack_kernel) through a
// 32-base window it keeps in two registers, refilled one round ahead; and it writes no bases at all: its output is the
// list of its FIXES (position, bases of the read consumed, corrected base: 4 bytes each, one per ~40 bases), which
// lane_apply_kernel replays over the original bytes -- everything the scan copies through verbatim (mod.rs:91,100),
// lowercase and non-ACGT bytes included, is copied from the input by that kernel, coalesced.
#include "brx_correct.hpp"

#include <stdlib.h>
#include <algorithm>
#include <vector>

using namespace brx;

namespace brx {
uint64_t scan_tmp_bytes(uint32_t n);
int exclusive_scan_lens(const uint32_t *d_lens, uint32_t n, uint64_t *d_tmp, uint64_t *d_out_offsets,
                        unsigned long long *d_total, hipStream_t s);
}


namespace {

constexpr uint32_t U_VOID = 0xffffffffu;   // u_q: the unit found no sync point (its predecessor scans through it)
constexpr uint32_t U_END = 0xffffffffu;    // a target: the end of the read
constexpr uint32_t C_FAIL = 0xfffffff1u;   // res[6]: three misses in a row, or an edit list overflowed: back to the group kernel
constexpr uint32_t C_VOID = 0xfffffff2u;   // res[6]: nothing produced
constexpr uint32_t U_UNWRITTEN = 0xffffffffu; // res[6] as lane_pass leaves it before the automaton: no lane has been here
constexpr int MAX_DEPTH = 3;               // edit lists: own stretch + two stretches scanned after a miss
constexpr uint32_t MAX_LANE_READ = 1u << 28; // positions are kept in 28 bits of an edit
// (a 10 kb read has ~250 fixes in ~19 units: batches of 256 fixes / 128 pieces take most reads in one go and leave the
// replay kernel's LDS at 9.5 KB a block -- 512 / 256 measured 2.37 against 2.22 ms per pass, profiles/r4m_greedy_ab.txt)
#ifndef BRX_AP_EDITS
#define BRX_AP_EDITS 256
#endif
#ifndef BRX_AP_PIECES
#define BRX_AP_PIECES 128
#endif
constexpr uint32_t AP_EDITS = BRX_AP_EDITS;   // fixes replayed per batch
constexpr uint32_t AP_PIECES = BRX_AP_PIECES; // pieces (unit, depth) gathered per batch
#ifndef BRX_AP_BS
#define BRX_AP_BS 128
#endif
constexpr uint32_t AP_BS = BRX_AP_BS; // threads of One's replay kernel per read
constexpr uint32_t APW_EDITS = 512; // ... by the walking correctors' replay kernel
constexpr uint32_t AP_VERIFY = 1024; // longest fixed-length walk whose k-mers the replay kernel checks for a repeat
constexpr uint32_t LANE_GRAB = 64;           // units a wave draws from the global counter at a time
// (Round 4 measured guided self-scheduling of the draws -- fewer units at a time as the pass runs out, so that no wave
// sits on units another could take -- with finer units for the last reads of the batch: no change / 0.9 ms slower,
// profiles/r4i_lane_tail_ab.txt.  The draws are a constant again; BRX_LANE_TAIL keeps the graded chunks for measurements.)
#ifndef BRX_LANE_WAVES
#define BRX_LANE_WAVES 7 // waves per SIMD the automaton is compiled for (tools/ab_build.sh sweeps it)
#endif

struct UnitDesc;
struct LaneArgs {
    PassParams p;
    uint32_t C;                // nominal chunk length
    uint32_t r_half, r_quarter; // reads from r_half on are cut at C / 2, from r_quarter on at C / 4 (chunk_of)
    uint32_t R;                // solid original k-mers in a row that make a sync point
    uint32_t *nu;              // units per read                            [n_reads]
    uint64_t *ubase;           // exclusive scan of nu                      [n_reads + 1]
    uint32_t *u_read;          // read of a unit                            [units]
    uint4 *u_in;               // where its read is: in_at lo, hi, n, j     [units]
    uint32_t *u_q;             // its sync position (0 for a read's first)  [units]
    uint64_t *u_qk;            // the original k-mer in front of it         [units]
    uint32_t *u_res;           // n0 t1 n1 t2 n2 t3 code -                  [8 x units]
    UnitDesc *u_desc;          // what a lane loads to start on a unit      [units]
    uint32_t *P;               // the batch as 2-bit codes, 16 bases per dword, first base in the top bits
    uint16_t *M;               // beside every dword of P: bit j = the ORIGINAL k-mer that ends at its base j is solid (or null)
    uint32_t *E[MAX_DEPTH];    // edit lists: pos << 4 | consumed << 2 | base
    uint2 *EW[MAX_DEPTH];      // the walking correctors' lists: pos, consumed << 16 | bases written (same memory as E)
    uint32_t *BW[MAX_DEPTH];   // ... and the written bases, 16 per word, a fix starting a word
    uint32_t *fail_list;       // reads handed back to the group kernel     [n_reads]
    const uint64_t *succ;      // per index line, a byte per slot: the unique solid successor of either orientation (or null)
    unsigned long long *dbg;   // BRX_LANE_TIMING: [0] waves recorded, then per wave: start, end (100 MHz ticks), loop iterations
};

// Chunk length of read r.  Units are handed out in read order and a lane keeps a unit to its end, so the pass ends
// when the LAST units do: with 545-base units (~3 ms each at configs[1]) the waves of the automaton finished between
// 10.3 and 15.8 ms (BRX_LANE_TIMING, profiles/r4h_lane_wave_timing.txt) -- the launch is as long as its slowest wave.
// BRX_LANE_TAIL=1 cuts the reads at the end of the batch finer (what is dealt last is small); see lane_pass for what it gave.
__device__ __forceinline__ uint32_t chunk_of(const LaneArgs &a, uint32_t r)
{
    const uint32_t c = r >= a.r_quarter ? a.C >> 2 : (r >= a.r_half ? a.C >> 1 : a.C);
    return c < 64u ? 64u : c;
}

// byte offset of read r's input in its buffer, its length (0 for a poisoned read)
__device__ __forceinline__ uint64_t read_view(const PassParams &p, uint32_t r, const uint8_t *&in, uint32_t &n, bool &poisoned)
{
    const uint64_t o0 = p.offsets[r], o1 = p.offsets[r + 1];
    poisoned = false;
    uint64_t at;
    if (p.in_staged) {
        at = slot_of(o0, r, p.slack);
        n = p.in_lens[r];
        if (n == 0xffffffffu) {
            poisoned = true;
            n = 0;
        }
    } else {
        at = o0;
        n = (uint32_t)(o1 - o0);
    }
    in = p.in + at;
    return at;
}
// A pass over reads stored back to front (p.flip: run_correction's reverse pass reads the forward pass's output
// backwards, src/lib.rs:48-55; no reversed copy is ever made) sees LOGICAL base j at in[n - 1 - j].  Everything between
// the packed copy and the replay works on logical positions; these three are the only places that touch the bytes.
__device__ __forceinline__ uint8_t ld_logical(const uint8_t *in, uint32_t n, uint32_t j, bool flip) { return in[flip ? n - 1u - j : j]; }
// logical bases j .. j+15 (all inside the read), first one in the lowest byte
__device__ __forceinline__ uint4 ld16_logical(const uint8_t *in, uint32_t n, uint32_t j, bool flip)
{
    uint4 q;
    if (!flip) {
        __builtin_memcpy(&q, in + j, 16);
        return q;
    }
    __builtin_memcpy(&q, in + (n - 16u - j), 16);
    return make_uint4(__builtin_bswap32(q.w), __builtin_bswap32(q.z), __builtin_bswap32(q.y), __builtin_bswap32(q.x));
}
// first dword of read r in P: 16 bases per dword and five dwords of padding per read (the window prefetches ahead)
__device__ __forceinline__ uint64_t pack_start(uint64_t in_at, uint32_t r) { return (in_at >> 4) + 5ull * r; }
// first entry of unit jj (sync position q) of a read in an edit list: a quarter entry per base + 16 per unit.  Monotone
// in (read, jj, q), so the distance to the next unit's start is this unit's capacity.
__device__ __forceinline__ uint64_t edit_start(uint64_t in_at, uint32_t r, uint64_t ub, uint32_t jj, uint32_t q)
{
    return (in_at >> 2) + 16ull * (ub + r + jj) + (uint64_t)(q >> 2);
}

// ---- unit tables, packed copy ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lane_units_kernel(LaneArgs a)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= a.p.n_reads)
        return;
    const uint8_t *in;
    uint32_t n;
    bool poisoned;
    (void)read_view(a.p, r, in, n, poisoned);
    const uint32_t C = chunk_of(a, r);
    a.nu[r] = n < 2u * C ? 1u : n / C; // the last unit takes the remainder
}

__global__ __launch_bounds__(256) void lane_pack_kernel(LaneArgs a)
{
    for (uint32_t r = blockIdx.x; r < a.p.n_reads; r += gridDim.x) {
        const uint8_t *in;
        uint32_t n;
        bool poisoned;
        const uint64_t at = read_view(a.p, r, in, n, poisoned);
        uint32_t *dst = a.P + pack_start(at, r);
        const uint32_t ndw = (n >> 4) + 5u;
        for (uint32_t d = threadIdx.x; d < ndw; d += 256) {
            const uint32_t b = 16u * d;
            uint32_t w[4] = {0, 0, 0, 0};
            if (b + 16u <= n) {
                const uint4 q = ld16_logical(in, n, b, a.p.flip);
                w[0] = q.x;
                w[1] = q.y;
                w[2] = q.z;
                w[3] = q.w;
            } else {
                for (uint32_t t = 0; t < 16u && b + t < n; t++)
                    w[t >> 2] |= (uint32_t)ld_logical(in, n, b + t, a.p.flip) << (8u * (t & 3u));
            }
            // bytes b0..b3 of a word (b0 = first base) -> b0<<6 | b1<<4 | b2<<2 | b3 by one multiply
            uint32_t v = 0;
#pragma unroll
            for (int t = 0; t < 4; t++)
                v = (v << 8) | ((((w[t] >> 1) & 0x03030303u) * 0x40100401u) >> 24);
            dst[d] = v;
        }
        // and the read's units
        const uint64_t ub = a.ubase[r], ue = a.ubase[r + 1];
        for (uint64_t u = ub + threadIdx.x; u < ue; u += 256) {
            a.u_read[u] = r;
            a.u_in[u] = make_uint4((uint32_t)at, (uint32_t)(at >> 32), n, (uint32_t)(u - ub));
        }
    }
}

// KmerSet::get of one forward k-mer, whatever holds the set (src/set/pcon.rs:189-191)
template <bool IDX>
__device__ __forceinline__ bool set_get(const PassParams &p, uint64_t km, int k)
{
    if (!IDX)
        return probe(p.bits, km, k);
    uint64_t key;
    const uint32_t home = index_locate(p.idx, km, k, key);
    for (uint32_t hop = 0;; hop++) {
        const int pr = index_probe_at(p.idx, key, home, hop);
        if (pr != 2)
            return pr == 1;
        if (p.bits) { // the line overflowed at build time and does not hold the key: the bit vector knows
            const uint64_t h = key - 1ull;
            return (p.bits[h >> 5] >> (h & 31u)) & 1u;
        }
    }
}

// Which ORIGINAL k-mers of the batch are solid, one bit per position (a.M, laid out like P): asked once, in parallel, 64
// neighbouring positions per wave -- neighbours share their index lines, so this is the cheap way to ask.  The automata
// then scan stretches the reads' own bases cover (no fix inside the last k - 1 positions) from these bits, many
// positions per round and without a probe.  A wave per unit, the unit's nominal stretch cut at multiples of 16.
template <bool IDX>
__global__ __launch_bounds__(256) void lane_mask_kernel(LaneArgs a)
{
    const PassParams &p = a.p;
    const int lane = threadIdx.x & 63;
    const int k = p.k;
    const uint64_t mask = kmask(k);
    const unsigned long long n_units = p.ctrl[CTL_LANE_UNITS];
    const unsigned long long wave = (unsigned long long)blockIdx.x * 4ull + (threadIdx.x >> 6);
    const unsigned long long n_waves = (unsigned long long)gridDim.x * 4ull;
    const bool filter = p.idx.line_bits != nullptr && p.idx.line_shift >= 6u; // (line_shift = 32 - log2(lines))
    uint4 ui_next = wave < n_units ? a.u_in[wave] : make_uint4(0, 0, 0, 0);
    uint32_t r_next = wave < n_units ? a.u_read[wave] : 0u;
    for (unsigned long long u = wave; u < n_units; u += n_waves) {
        const uint4 ui = ui_next;
        const uint32_t r = r_next;
        if (u + n_waves < n_units) {
            ui_next = a.u_in[u + n_waves];
            r_next = a.u_read[u + n_waves];
        }
        const uint32_t j = ui.w, n = ui.z;
        const uint64_t at = ((uint64_t)ui.y << 32) | ui.x;
        const bool last = (uint64_t)j + 1ull == a.ubase[r + 1] - a.ubase[r];
        const uint32_t C = chunk_of(a, r);
        const uint32_t s = (j * C) & ~15u; // (>= 64 for j >= 1: C >= 64)
        const uint32_t lim = last ? n : ((j + 1u) * C) & ~15u;
        const uint32_t *row = a.P + pack_start(at, r);
        uint16_t *mrow = a.M + pack_start(at, r);
        // the 31 bases in front of s: what the first k-mers of the stretch reach back into
        uint64_t carry = 0;
        if (s) {
            const uint32_t b = s - 31u, w = b >> 4, sh = 2u * (b & 15u);
            const uint64_t hi = ((uint64_t)row[w] << 32) | row[w + 1];
            const uint64_t lo = row[w + 2];
            carry = (sh ? ((hi << sh) | (lo >> (32u - sh))) : hi) >> 2;
        }
        for (uint32_t e0 = s; e0 < lim; e0 += 64u) {
            const uint32_t e = e0 + (uint32_t)lane;
            const uint32_t code = (row[e >> 4] >> (30u - 2u * (e & 15u))) & 3u; // (the padding behind the read packs as A)
            const uint64_t km = lane_kmer64_dpp(carry, code, lane, mask);
            bool sol = false;
            if (e < lim && e + 1u >= (uint32_t)k) {
                if (IDX) {
                    uint64_t key;
                    const uint32_t home = index_locate(p.idx, km, k, key);
                    // an empty home line answers from its occupancy bit, and most k-mers with an error point at one -- while
                    // the bits live in the L2 (4 MiB at 2^25 lines).  Past 2^26 lines every lookup is a request of its own on
                    // top of the line's (BASELINE configs[4]'s share, 2^28 lines: this kernel 40.1 ms with the bits, 35.1 without)
                    if (!filter || ((p.idx.line_bits[home >> 5] >> (home & 31u)) & 1u)) {
                        int pr = index_probe_at(p.idx, key, home, 0u);
                        for (uint32_t hop = 1; pr == 2 && !p.bits; hop++)
                            pr = index_probe_at(p.idx, key, home, hop);
                        if (pr == 2) {
                            const uint64_t h = key - 1ull;
                            pr = (p.bits[h >> 5] >> (h & 31u)) & 1u;
                        }
                        sol = pr == 1;
                    }
                } else {
                    sol = probe(p.bits, km, k);
                }
            }
            const uint64_t ball = __ballot(sol);
            if ((lane & 15) == 0 && e < lim)
                mrow[e >> 4] = (uint16_t)(ball >> lane);
            carry = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(km >> 32), 63) << 32) |
                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)km, 63);
        }
    }
}

// Half a wave per unit boundary: the first position q > j*C + k behind R solid original k-mers in a row, inside the unit's
// own nominal stretch; none -> the unit is void.  q is the loop-top position (mod.rs:68), u_qk the k-mer in front of it.
// 32 positions per step and unit, every one of them a k-mer that lies inside the stretch (they are cut from the packed
// copy, three dwords each, so no lane is spent on the k - 1 bases in front of the first one): a run of four turns up
// within the first 32 nearly always, and the first form -- a wave per unit, 64 positions per step of which the first
// k - 1 could not count -- made twice the probes (1.4 -> 0.9 ms at configs[1]).
template <bool IDX>
__global__ __launch_bounds__(256) void lane_sync_kernel(LaneArgs a)
{
    const PassParams &p = a.p;
    const uint32_t lane = threadIdx.x & 63u, half = lane >> 5, hl = lane & 31u;
    const uint32_t k = (uint32_t)p.k;
    const unsigned long long n_units = p.ctrl[CTL_LANE_UNITS];
    const unsigned long long wave = (unsigned long long)blockIdx.x * 4ull + (threadIdx.x >> 6);
    const unsigned long long n_waves = (unsigned long long)gridDim.x * 4ull;
    for (unsigned long long u0 = 2ull * wave; u0 < n_units; u0 += 2ull * n_waves) {
        const unsigned long long u = u0 + half;
        const bool mine = u < n_units;
        const uint4 ui = mine ? a.u_in[u] : make_uint4(0, 0, 0, 0);
        const uint32_t j = ui.w, n = ui.z;
        const uint32_t r = mine ? a.u_read[u] : 0u;
        const uint64_t at = ((uint64_t)ui.y << 32) | ui.x;
        const uint32_t *row = a.P + pack_start(at, r);
        const uint32_t C = chunk_of(a, r);
        const uint32_t s = j * C;
        const uint32_t lim = (s + C < n) ? s + C : n; // k-mers ending at e < lim ...
        const uint32_t e0 = s + k - 1u;                    // ... and starting at or behind s
        bool open_ = mine && j != 0u;                     // this half still looks for its unit's sync point
        uint32_t q = (mine && j == 0u) ? 0u : U_VOID;
        uint64_t qk = 0;
        uint32_t prev32 = 0;
        for (uint32_t t = 0; __any(open_ && e0 + 32u * t < lim); t++) {
            const uint32_t e = e0 + 32u * t + hl;
            const bool valid = open_ && e < lim;
            uint64_t km = 0;
            bool sol = false;
            if (valid) {
                const uint32_t b = e + 1u - k, wd = b >> 4, sh = 2u * (b & 15u);
                const uint64_t hi = ((uint64_t)row[wd] << 32) | row[wd + 1];
                const uint64_t lo = row[wd + 2];
                km = (sh ? ((hi << sh) | (lo >> (32u - sh))) : hi) >> (64u - 2u * k);
                sol = set_get<IDX>(p, km, (int)k);
            }
            const uint64_t ball = __ballot(sol);
            const uint32_t cur32 = (uint32_t)(ball >> (32u * half));
            const uint64_t w = ((uint64_t)cur32 << 32) | prev32; // the half's last 64 answers, this step's in the top half
            uint64_t x = w;
            for (uint32_t rr = 1; rr < a.R; rr++)
                x &= w << rr;
            const uint32_t hit = (uint32_t)(x >> 32);
            const bool found = open_ && hit != 0u;
            const uint32_t el_l = found ? (uint32_t)__builtin_ctz(hit) : 0u; // lane of the half whose k-mer ends the run
            // (every lane takes part in the shuffles; the source lane is in the asker's own half)
            const uint32_t src = 32u * half + el_l;
            const uint32_t klo = (uint32_t)__shfl((int)(uint32_t)km, (int)src), khi = (uint32_t)__shfl((int)(uint32_t)(km >> 32), (int)src);
            if (found) {
                const uint32_t qq = e0 + 32u * t + el_l + 1u;
                if (qq < n) {
                    q = qq;
                    qk = ((uint64_t)khi << 32) | klo;
                }
                open_ = false;
            } else if (open_ && e0 + 32u * (t + 1u) >= lim) {
                open_ = false;
            }
            prev32 = cur32;
        }
        if (mine && hl == 0u) {
            a.u_q[u] = q;
            if (j != 0u)
                a.u_qk[u] = qk;
        }
    }
}

// The same sync points read off the solidity mask (when lane_mask_kernel has run): one THREAD per unit, sixteen
// positions per step -- a run of R set bits that lies inside the unit's nominal stretch.
__global__ __launch_bounds__(256) void lane_sync_mask_kernel(LaneArgs a)
{
    const PassParams &p = a.p;
    const uint32_t k = (uint32_t)p.k;
    const unsigned long long n_units = p.ctrl[CTL_LANE_UNITS];
    for (unsigned long long u = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; u < n_units; u += (unsigned long long)gridDim.x * 256ull) {
        const uint4 ui = a.u_in[u];
        const uint32_t j = ui.w, n = ui.z;
        if (j == 0) {
            a.u_q[u] = 0;
            continue;
        }
        const uint32_t r = a.u_read[u];
        const uint64_t at = ((uint64_t)ui.y << 32) | ui.x;
        const uint32_t *row = a.P + pack_start(at, r);
        const uint16_t *mrow = a.M + pack_start(at, r);
        const uint32_t C = chunk_of(a, r);
        const uint32_t s = j * C;
        const uint32_t lim = (s + C < n) ? s + C : n; // k-mers ending at e < lim ...
        const uint32_t e0 = s + k - 1u;                    // ... and starting at or behind s
        uint32_t q = U_VOID;
        uint64_t qk = 0;
        uint64_t w = 0; // the last 64 answers, the newest sixteen in the top bits
        for (uint32_t g = e0 >> 4; 16u * g < lim; g++) {
            uint32_t bits = mrow[g];
            const uint32_t base = 16u * g;
            if (base < e0)
                bits &= 0xffffu << (e0 - base);
            if (base + 16u > lim)
                bits &= 0xffffu >> (base + 16u - lim);
            w = (w >> 16) | ((uint64_t)bits << 48);
            uint64_t t = w;
            for (uint32_t rr = 1; rr < a.R; rr++)
                t &= w << rr;
            const uint32_t hit = (uint32_t)(t >> 48);
            if (hit) {
                const uint32_t el = base + (uint32_t)__builtin_ctz(hit); // the run's last k-mer ends here
                if (el + 1u < n) {
                    q = el + 1u;
                    const uint32_t b = el + 1u - k, wd = b >> 4, sh = 2u * (b & 15u);
                    const uint64_t hi = ((uint64_t)row[wd] << 32) | row[wd + 1];
                    const uint64_t lo = row[wd + 2];
                    qk = (sh ? ((hi << sh) | (lo >> (32u - sh))) : hi) >> (64u - 2u * k);
                }
                break;
            }
        }
        a.u_q[u] = q;
        a.u_qk[u] = qk;
    }
}

// Everything a lane needs to start on a unit, or to go on into it after a missed prediction, in ONE 48-byte record: a
// lane that takes a unit stalls its whole wave while the loads it depends on come back, so the chain is kept at
// counter -> record -> window (the first form of the hand-out walked ten dependent loads and cost as much as the scan
// of short units itself).
struct __attribute__((aligned(16))) UnitDesc {
    uint32_t n;     // length of the read
    uint32_t q;     // sync position (0 for a read's first unit; U_VOID: no sync point, nothing to do)
    uint32_t t1;    // the next unit of the read that has a sync point (U_END: none)
    uint32_t tgt;   // its sync position (the read's length for U_END)
    uint64_t tgtk;  // the original k-mer in front of it
    uint64_t eat;   // first entry of this unit's stretch in an edit list
    uint64_t pw;    // dword of P that holds position q
    uint32_t ecap;  // entries up to the target's stretch
    uint32_t first; // 1: the read's first unit (the scan starts at position k, mod.rs:60-67)
};
// (Round 4 tried an 80-byte record that also carried the window's three dwords, the mask words and the k-mer in front of
// the sync point -- one trip to memory per unit start instead of three dependent ones: One's launch did not move (11.05
// against 11.09 ms per pass) and the walking automata, at their register limit, spilled and lost 10 % (graph 65.6 ->
// 75.5 ms per Gbp): profiles/r4z_methods_1gbp.jsonl against r4c_methods_1gbp_lane_rev_off.jsonl.  Taken out again.)

__global__ __launch_bounds__(256) void lane_link_kernel(LaneArgs a)
{
    const PassParams &p = a.p;
    const unsigned long long n_units = p.ctrl[CTL_LANE_UNITS];
    for (unsigned long long u = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; u < n_units; u += (unsigned long long)gridDim.x * 256ull) {
        const uint32_t r = a.u_read[u];
        const uint32_t ub = (uint32_t)a.ubase[r], uend = (uint32_t)a.ubase[r + 1];
        const uint8_t *in;
        uint32_t n;
        bool poisoned;
        const uint64_t in_at = read_view(p, r, in, n, poisoned);
        UnitDesc d;
        d.n = n;
        d.q = n >= MAX_LANE_READ ? U_VOID - 1u : a.u_q[u]; // (U_VOID - 1: the read is too long for the edits' 28-bit positions)
        uint32_t t = (uint32_t)u + 1u;
        while (t < uend && a.u_q[t] == U_VOID)
            t++;
        const uint32_t q0 = d.q >= U_VOID - 1u ? 0u : d.q;
        d.eat = edit_start(in_at, r, ub, (uint32_t)u - ub, q0);
        uint64_t end_at;
        if (t < uend) {
            d.t1 = t;
            d.tgt = a.u_q[t];
            d.tgtk = a.u_qk[t];
            end_at = edit_start(in_at, r, ub, t - ub, d.tgt);
        } else {
            d.t1 = U_END;
            d.tgt = n;
            d.tgtk = 0;
            end_at = edit_start(in_at, r, ub, uend - ub, n);
        }
        d.ecap = (uint32_t)(end_at - d.eat);
        d.pw = pack_start(in_at, r) + (q0 >> 4);
        d.first = (uint32_t)u == ub ? 1u : 0u;
        a.u_desc[u] = d;
    }
}

// ---- the automaton ----------------------------------------------------------------------------------------------------
// One round of one lane = one KmerSet::get and the transition it decides.  The round is written as straight-line code
// over a PACKED state word, so that the 64 lanes of a wave -- each in a state of its own -- share every instruction;
// what is rare (a unit ends, a new one is fetched, a read ends within eight bases) sits in branches a wave skips.
//   st    0 SCAN  probe add(kmer, seq[i])                   mod.rs:69-73     (`first`: probe kmer itself, mod.rs:67)
//         1 ALTS  probe the trigger k-mer with last base cur   mod.rs:114-128 (the read's own base is known not solid)
//         2 SCEN  probe corr + seq[off .. off+jj]            exist/mod.rs:33-41  (scenario cur: I / S / D, off = 2 - cur)
//         3 MORE  probe corr + seq[off .. off+c]             exist/mod.rs:57-66
enum { S_ST = 0, S_CUR = 2, S_JJ = 4, S_ACC = 8, S_PASS = 12, S_DIRTY = 16, S_SKIP = 24, S_PREV = 27, S_FIRST = 28, S_SLOW = 29 };
constexpr uint32_t MASK_STEP = 16; // positions a clean SCAN round takes at most: what a round's refill puts back into the window

template <bool IDX, int KT, bool MASK>
__global__ __launch_bounds__(256, MASK ? 6 : BRX_LANE_WAVES) void lane_kernel(LaneArgs a)
{
    const PassParams &p = a.p;
    const int k = KT ? KT : p.k;
    const uint32_t c = (uint32_t)p.c; // 1 .. 5
    const uint64_t mask = kmask(k);
    const unsigned long long n_units = p.ctrl[CTL_LANE_UNITS];
    // the index line a lane probed last, kept in LDS (element e of lane t at [e][t]: conflict-free 16-byte accesses):
    // consecutive k-mers of a read share their minimizer for ~3 positions, and a line fetched per lane per round is what
    // made the first form of this kernel wait for the L2 fabric
    // (Round 4 measured two other ways of getting the line: a one-round-ahead fetch of the NEXT position's line into a sink,
    // and four neighbouring lanes pulling a quarter each of one lane's line -- 54 G lines/s against 38-45 in
    // tools/line_probe_bench.hip's chain of dependent rounds where EVERY lane fetches EVERY round.  In this kernel, where
    // about half the lanes fetch in a round, both lose: 11.2-11.5 against 10.6-10.9 ms per pass, and 28.2 (with the
    // quarters read back in a turned order: spills) / 22.6 (without) / 22.0 (six waves) against 21.6 ms for both passes:
    // profiles/r4k_lane_prefetch_ab.txt, profiles/r4l_lane_coop_fetch_ab.txt.)
    __shared__ uint4 lc[4][256];
    const uint32_t tid = threadIdx.x;

    // the unit
    bool have = false, want = true;
    uint32_t u = 0, n = 0, i = 0, tgt = 0, t1 = U_END;
    uint64_t tgtk = 0;
    uint32_t depth = 0, ne = 0, ecap = 0;
    uint64_t eat = 0; // first entry of the edit list being written
    // the bases ahead: positions i .. i+wcnt of the read, 2 bits each, position i in the top bits; the next 16 in nextw
    uint64_t wreg = 0;
    uint32_t wcnt = 0, nextw = 0, pidx = 0;
    // ... and whether the ORIGINAL k-mers that end at them are solid (lane_mask_kernel): bit j of mw for position i + j,
    // wcnt bits of it, the next 16 in nextm.  A stretch of the scan whose k-mers hold no corrected base (S_DIRTY == 0)
    // is read off these bits, up to MASK_STEP positions per round, without a probe.
    constexpr bool use_mask = MASK; // (a variant of its own: the masked form needs 6 registers more than 7 waves leave)
    uint32_t mw = 0, nextm = 0;
    // the scan (mod.rs:60-67) and the trigger in progress
    uint64_t kmer = 0, corr = 0;
    uint32_t S = 0, hop = 0, cline = 0xffffffffu;
    // (fixes are not counted here: a unit that started on a wrong prediction, or a stretch scanned a second time after
    // a miss, makes fixes nobody keeps -- the replay kernel counts the ones it commits)
    uint32_t n_rounds = 0, n_probes = 0, n_trig = 0, n_miss = 0; // wave-uniform (scalar registers)
    uint32_t wnext = 0, wend = 0; // the units this wave has drawn and not yet dealt out (wave-uniform)
#ifdef BRX_LANE_TIMING // (a build of its own, -DBRX_LANE_TIMING: the counters cost the automaton its last free registers)
    const unsigned long long t_begin = a.dbg ? wall_clock64() : 0ull;
    uint32_t w_iters = 0;
#endif

    for (;;) {
#ifdef BRX_LANE_TIMING
        w_iters++;
#endif
        // ---- rare: a unit ends where the scan reaches its target (loop top, mod.rs:68); units are handed out --------
        uint32_t ev = 0; // this lane's events of the round: 1 probe, 2 trigger, 8 missed prediction, 16 a second probe
        const bool at_end = have && (S & 3u) == 0u && i >= tgt;
        if (__any(at_end || want)) {
            if (at_end) {
                uint32_t *res = a.u_res + 8ull * u;
                res[2 * depth] = ne;
                res[2 * depth + 1] = t1;
                if (t1 == U_END || (i == tgt && kmer == tgtk)) {
                    res[6] = depth;
                    want = true;
                } else if (depth + 1 < (uint32_t)MAX_DEPTH) {
                    // the prediction did not hold: keep scanning through the next unit's stretch, into the next list
                    ev |= 8u;
                    depth++;
                    const uint4 *dp = reinterpret_cast<const uint4 *>(a.u_desc + t1);
                    const uint4 d0 = dp[0], d1 = dp[1], d2 = dp[2];
                    t1 = d0.z;
                    tgt = d0.w;
                    tgtk = ((uint64_t)d1.y << 32) | d1.x;
                    eat = ((uint64_t)d1.w << 32) | d1.z;
                    ecap = d2.z;
                    ne = 0;
                } else {
                    res[6] = C_FAIL;
                    want = true;
                }
            }
            // ONE place where lanes take new units.  The wave draws LANE_GRAB units at a time from the global counter (one
            // atomic per 64 units: 7 168 waves incrementing one address once per unit measured ~8 ns per unit, i.e. the whole
            // kernel's time at short chunks) and deals them to its lanes from a wave-uniform cursor.
            for (;;) {
                const uint64_t wm = __ballot(want);
                if (!wm)
                    break;
                if (wnext == wend) {
                    unsigned long long base = 0;
                    if ((tid & 63u) == 0u)
                        base = atomicAdd(p.ctrl + CTL_LANE_WORK, (unsigned long long)LANE_GRAB);
                    base = __shfl(base, 0);
                    wnext = base < n_units ? (uint32_t)base : (uint32_t)n_units;
                    wend = base + LANE_GRAB < n_units ? (uint32_t)(base + LANE_GRAB) : (uint32_t)n_units;
                    if (wnext == wend) { // the pass has no units left
                        if (want)
                            have = false;
                        want = false;
                        break;
                    }
                }
                const uint32_t rank = (uint32_t)__builtin_popcountll(wm & ((1ull << (tid & 63u)) - 1ull));
                const uint32_t avail = wend - wnext;
                const uint32_t asked = (uint32_t)__builtin_popcountll(wm);
                const bool take = want && rank < avail;
                const uint32_t my = wnext + rank;
                wnext += asked < avail ? asked : avail;
                if (take) {
                    want = false;
                    have = true;
                    u = my;
                    const uint4 *dp = reinterpret_cast<const uint4 *>(a.u_desc + u);
                    const uint4 d0 = dp[0], d1 = dp[1], d2 = dp[2];
                    n = d0.x;
                    const uint32_t q = d0.y;
                    if (q >= U_VOID - 1u) {
                        a.u_res[8ull * u + 6] = q == U_VOID ? C_VOID : C_FAIL;
                        have = false;
                        want = true;
                    } else {
                        t1 = d0.z;
                        tgt = d0.w;
                        tgtk = ((uint64_t)d1.y << 32) | d1.x;
                        eat = ((uint64_t)d1.w << 32) | d1.z;
                        const uint64_t pw = ((uint64_t)d2.y << 32) | d2.x;
                        ecap = d2.z;
                        depth = 0;
                        ne = 0;
                        hop = 0;
                        // the window at q
                        wreg = (((uint64_t)a.P[pw] << 32) | a.P[pw + 1]) << ((q & 15u) * 2u);
                        wcnt = 32u - (q & 15u);
                        nextw = a.P[pw + 2];
                        pidx = (uint32_t)pw + 3u;
                        if (use_mask) {
                            mw = ((uint32_t)a.M[pw] | ((uint32_t)a.M[pw + 1] << 16)) >> (q & 15u);
                            nextm = a.M[pw + 2];
                        }
                        if (d2.w) { // the read's first unit
                            if (n < (uint32_t)k) { // mod.rs:56-58: returned verbatim, i.e. no edits
                                uint32_t *res = a.u_res + 8ull * u;
                                res[0] = 0;
                                res[1] = U_END;
                                res[6] = 0;
                                have = false;
                                want = true;
                            } else {
                                kmer = wreg >> (64 - 2 * k);
                                wreg <<= 2 * k; // k <= 31
                                mw >>= k;
                                wcnt -= (uint32_t)k;
                                i = (uint32_t)k;
                                S = 1u << S_FIRST; // previous = get(kmer), mod.rs:67: SCAN that probes kmer itself; prev = false accepts it
                            }
                        } else {
                            i = q;
                            kmer = a.u_qk[u];
                            S = 1u << S_PREV; // R >= 1 solid k-mers end in front of q
                        }
                    }
                }
            }
            if (!__any(have))
                break;
        }
        // (a lane that just missed its target takes part in this round only if it still is in front of the next one)
        const bool act = have && !((S & 3u) == 0u && i >= tgt);
        n_rounds += (uint32_t)__builtin_popcountll(__ballot(act));
        if (act) {
            if (wcnt <= 16u) { // (the dword was loaded at the refill before this one)
                wreg |= (uint64_t)nextw << (32u - 2u * wcnt);
                mw |= nextm << wcnt;
                wcnt += 16u;
                nextw = a.P[pidx];
                if (use_mask)
                    nextm = a.M[pidx];
                pidx++;
            }
            const uint32_t cw = (uint32_t)(wreg >> 48); // seq[i .. i+8), first base in bits 15:14
            const uint32_t c0 = cw >> 14;               // seq[i]
            const uint32_t st = S & 3u, cur = (S >> S_CUR) & 3u, jj = (S >> S_JJ) & 7u, skip = (S >> S_SKIP) & 7u;
            const bool prev = (S >> S_PREV) & 1u, first = (S >> S_FIRST) & 1u, slow = (S >> S_SLOW) & 1u;
            const bool is0 = st == 0u, is1 = st == 1u, is2 = st == 2u, is3 = st == 3u;
            const uint32_t rem = n - i;

            // ---- the k-mer this state asks about ------------------------------------------------------------------------
            const uint32_t off = 2u - cur; // I:2 S:1 D:0 (one.rs:57-63); meaningful in SCEN / MORE only
            const uint32_t nb = is0 ? (first ? 0u : 1u) : (is2 ? jj + 1u : (is3 ? c + 1u : 0u));
            const uint32_t b0 = (is2 || is3) ? off : 0u;
            const uint32_t wbits = (cw >> (16u - 2u * (b0 + nb))) & ((1u << (2u * nb)) - 1u);
            uint64_t pk = (((is0 ? kmer : corr) << (2u * nb)) | (uint64_t)wbits) & mask;
            pk = is1 ? ((pk & ~3ull) | (uint64_t)cur) : pk;
            // a base accepted behind a fix is known solid (it was a look-ahead of the winning scenario); a tie-break
            // that cannot read one base more is false without a probe (exist/mod.rs:54)
            // SCAN over the read's own bases: the answer is in the mask
            const uint32_t dirty = (S >> S_DIRTY) & 63u;
            const bool clean = use_mask && is0 && !first && skip == 0u && dirty == 0u && !slow && hop == 0u;
            const bool need = !(is0 && skip != 0u) && !(is3 && !(rem > c + off + 1u)) && !clean;
            ev |= need ? 1u : 0u;

            // ---- KmerSet::get -------------------------------------------------------------------------------------------
            bool sol = clean ? (mw & 1u) != 0u : (is0 && !need), unres = false;
            // SCAN asks about TWO positions when it can: add(pk, seq[i+1]) shares its reverse complement and all but one of its
            // minimizer windows with pk, and two k-mers in three share their index line -- then the second answer comes out
            // of the line already in hand, and a round moves the scan two bases.  Not across the unit's target, not behind a
            // fix (skip), not when the second k-mer lives in another line: the next round asks about it then, as before.
            bool two = false, sol2 = false;
            uint64_t pk2 = 0;
            const uint32_t c1 = (cw >> 12) & 3u; // seq[i + 1]
            if (IDX) {
                const uint64_t rc = revcomp(pk, k);
                const uint64_t key = (((popc64(pk) & 1) ? rc : pk) >> 1) + 1ull;
                uint32_t best, best_excl;
                const bool pairable = minimizer_pair(pk, rc, p.idx.m, p.idx.w, best, best_excl);
                const uint32_t home = index_line_of(best, p.idx.line_shift);
                const uint32_t line = (home + hop) & (0xffffffffu >> p.idx.line_shift);
                pk2 = ((pk << 2) | (uint64_t)c1) & mask;
                const uint64_t rc2 = (rc >> 2) | ((uint64_t)(c1 ^ 2u) << (2 * k - 2));
                const uint64_t key2 = (((popc64(pk2) & 1) ? rc2 : pk2) >> 1) + 1ull;
                uint32_t home2;
                {
                    const uint32_t mm = p.idx.m >= 16u ? 0xffffffffu : (1u << (2u * p.idx.m)) - 1u;
                    const uint32_t f2 = (uint32_t)pk2 & mm, r2 = (uint32_t)(rc2 >> (2u * (p.idx.w - 1u))) & mm;
                    const uint32_t h2 = (f2 < r2 ? f2 : r2) * 0x9E3779B1u;
                    home2 = index_line_of(h2 < best_excl ? h2 : best_excl, p.idx.line_shift);
                }
                two = pairable && is0 && !first && need && skip == 0u && !slow && hop == 0u && home2 == home && tgt - i >= 2u && rem >= 2u;
                if (need && !slow && line != cline) { // not the line this lane holds: fetch it, keep it
                    // straight from global memory into LDS (global_load_lds_dwordx4: lane l of the wave writes at the
                    // wave-uniform base + 16 l, lanes switched off write nothing: tools/lds_dma_test.hip): no registers
                    // in between, no ds_write
                    const uint8_t *L = reinterpret_cast<const uint8_t *>(p.idx.lines + (uint64_t)line * 8ull);
                    const uint32_t wb = tid & ~63u;
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(L + 16 * e),
                                                         (__attribute__((address_space(3))) void *)&lc[e][wb], 16, 0, 0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    cline = line;
                }
                if (need && slow) { // the home line overflowed at build time and does not hold the key: the bit vector knows
                    const uint64_t h = key - 1ull;
                    sol = (p.bits[h >> 5] >> (h & 31u)) & 1u;
                } else if (need) {
                    const uint4 q0 = lc[0][tid], q1 = lc[1][tid], q2 = lc[2][tid], q3 = lc[3][tid];
                    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
                    const bool found = ((q0.x == klo) & (q0.y == khi)) | ((q0.z == klo) & (q0.w == khi)) | ((q1.x == klo) & (q1.y == khi)) |
                                       ((q1.z == klo) & (q1.w == khi)) | ((q2.x == klo) & (q2.y == khi)) | ((q2.z == klo) & (q2.w == khi)) |
                                       ((q3.x == klo) & (q3.y == khi));
                    // home line (hop 0): go on only if one of the keys it turned away had this key's signature; further
                    // down a chain only the flag can tell (brx_index.hpp: index_probe_at)
                    const uint32_t hdr_hi = q3.w;
                    const bool more = (hdr_hi >> 31) && (hop != 0u || ((hdr_hi >> idx_sig_index(key)) & 1u));
                    sol = found;
                    unres = !found && more;
                    // the second position, against the same line
                    const uint32_t k2lo = (uint32_t)key2, k2hi = (uint32_t)(key2 >> 32);
                    const bool found2 = ((q0.x == k2lo) & (q0.y == k2hi)) | ((q0.z == k2lo) & (q0.w == k2hi)) | ((q1.x == k2lo) & (q1.y == k2hi)) |
                                        ((q1.z == k2lo) & (q1.w == k2hi)) | ((q2.x == k2lo) & (q2.y == k2hi)) | ((q2.z == k2lo) & (q2.w == k2hi)) |
                                        ((q3.x == k2lo) & (q3.y == k2hi));
                    const bool more2 = (hdr_hi >> 31) && ((hdr_hi >> idx_sig_index(key2)) & 1u);
                    sol2 = found2;
                    two = two && !(!found2 && more2); // (a flagged line that may have turned the second k-mer away: ask next round)
                } else {
                    two = false;
                }
            } else if (need) {
                sol = probe(p.bits, pk, k);
            }
            ev |= (two && !unres) ? 16u : 0u;
            if (unres) {
                // ask again next round: the bit vector, or (sparse sets) the next line of the chain
                if (p.bits)
                    S |= 1u << S_SLOW;
                else
                    hop++;
            } else {
                hop = 0;
                // ---- transition: every state's successor computed side by side, one select per field ------------------------
                // The three probing states are one loop over CANDIDATES in rising order -- ALTS: the three bases that are not
                // the read's own (mod.rs:114-128; its own base is the trigger k-mer, known not solid); SCEN: the scenarios
                // I, S, D from the first one that can still read its c look-aheads (exist/mod.rs:27-29, :97-109); MORE: the
                // scenarios that scored c (exist/mod.rs:138-147) -- with one result mask, and one verdict when they run out.
                const uint32_t solb = sol ? 1u : 0u, above = ~((2u << cur) - 1u);
                const uint32_t acc = (S >> S_ACC) & 15u, passm = (S >> S_PASS) & 7u;
                const bool accept = is0 && (sol || !prev), trigA = is0 && !accept; // mod.rs:73, :99-102 (:67 for `first`)
                // the second position: `previous` is the first one's answer by then
                const bool acceptB = accept && two && (sol2 || !sol), trigB = accept && two && !(sol2 || !sol);
                const bool trig = trigA || trigB;
                const uint32_t ctrig = trigB ? c1 : c0; // the trigger base: ALTS starts at the first base that is not it
                const uint32_t cands = (is1 ? (15u & ~(1u << c0)) : (is2 ? 7u : passm)) & above;
                const bool s_pass = is2 && sol && jj + 1u == c;  // get_score == c
                const bool s_over = is2 && (!sol || s_pass);     // ... or it stops below c (exist/mod.rs:38-42)
                const uint32_t acc1 = acc | ((is1 || is3) ? solb << cur : 0u);
                const uint32_t pass1 = passm | (s_pass ? 1u << cur : 0u);
                const bool adv_c = is1 || is3 || s_over;
                const bool end = adv_c && (cands == 0u || (is1 && __popc(acc1) > 1));
                const uint32_t dm = is2 ? pass1 : acc1; // what the verdict looks at: alternatives / scenarios that held
                const int pc = __popc(dm);
                const uint32_t win = (uint32_t)__ffs(dm) - 1u;
                // the first scenario that can read its c look-aheads: I needs c + 2 bases, S c + 1, D c (3: none)
                const uint32_t smin = rem >= c + 2u ? 0u : (rem == c + 1u ? 1u : (rem == c ? 2u : 3u));
                const bool to_scen = end && is1 && pc == 1 && smin < 3u; // exist/mod.rs:121-129
                const bool to_more = end && is2 && pc > 1;
                const bool apply_b = end && !is1 && pc == 1;             // exist/mod.rs:135-137, :143-144
                const bool fail = end && !to_scen && !to_more && !apply_b; // exist/mod.rs:123-126, :132-134, :146
                const uint32_t used = 2u - win; // bases of the read the applied scenario consumes (one.rs:65-71)
                // The c look-ahead k-mers of the winning scenario ARE the next c scan k-mers, all found solid: the reference's
                // loop copies these bases with previous = true (mod.rs:99-102).  Jump over them -- unless the unit's target
                // lies among them: then they are walked one by one (without a probe), so that the state at the target is seen
                const bool jump = tgt - i > used + c;
                const uint32_t cb = (cw >> (16u - 2u * (used + c))) & ((1u << (2u * c)) - 1u);
                const bool room = ne < ecap;
                const bool fix = apply_b && room;
                // the new state word
                const uint32_t st1 = trig ? 1u : (to_scen ? 2u : (to_more ? 3u : ((fail || apply_b) ? 0u : st)));
                const uint32_t cur1 = trig ? (ctrig == 0u ? 1u : 0u)
                                           : (to_scen ? smin : (to_more ? win : ((adv_c && !end) ? (uint32_t)__ffs(cands) - 1u : cur)));
                const uint32_t jj1 = (is2 && !s_over) ? jj + 1u : 0u;
                const uint32_t accn = (trig || to_more) ? 0u : acc1, passn = to_scen ? 0u : pass1;
                const uint32_t skipn = accept ? (skip ? skip - 1u : 0u) : ((fix && !jump) ? c : skip);
                // a clean SCAN takes the whole run of equal answers (all solid, or all not: `previous` ends as the run's
                // answer either way), as far as the window, the target and a 64-bit shift of the k-mer allow
                uint32_t run = 1u;
                if (clean) {
                    const uint32_t same = (mw & 1u) ? ~mw : mw;
                    run = same ? (uint32_t)__builtin_ctz(same) : 32u;
                    const uint32_t room_t = tgt - i;
                    run = run < MASK_STEP ? run : MASK_STEP;
                    run = run < room_t ? run : room_t;
                    run = run < wcnt ? run : wcnt;
                }
                const uint32_t prevn = accept ? (acceptB ? (sol2 ? 1u : 0u) : solb) : (fail ? 0u : (apply_b ? 1u : (prev ? 1u : 0u)));
                const uint32_t adv = accept ? (first ? 0u : (clean ? run : (acceptB ? 2u : 1u))) : (fail ? 1u : (fix ? used + (jump ? c : 0u) : 0u));
                // positions ahead whose k-mers hold a corrected base: k - 1 behind a fix (its jump taken off)
                const uint32_t kd = (uint32_t)k - 1u, jd = jump ? c : 0u;
                const uint32_t dirtyn = fix ? (kd > jd ? kd - jd : 0u) : ((accept || fail) ? (dirty > adv ? dirty - adv : 0u) : dirty);
                S = st1 | (cur1 << S_CUR) | (jj1 << S_JJ) | (accn << S_ACC) | (passn << S_PASS) | (skipn << S_SKIP) | (prevn << S_PREV) |
                    (dirtyn << S_DIRTY);
                // the k-mers: a failed fix copies the trigger base through and keeps it in the k-mer (mod.rs:91-96)
                const uint64_t kfix = jump ? (((corr << (2u * c)) | (uint64_t)cb) & mask) : corr;
                const uint64_t krun = ((kmer << (2u * run)) | (wreg >> (64u - 2u * run))) & mask; // (run >= 1)
                kmer = accept ? (clean ? krun : (acceptB ? pk2 : pk)) : (fail ? ((corr & ~3ull) | (uint64_t)c0) : (fix ? kfix : kmer));
                corr = trigA ? pk : (trigB ? pk2 : (to_scen ? ((corr & ~3ull) | (uint64_t)win) : corr));
                ev |= trig ? 2u : 0u;
                if (apply_b) { // mod.rs:75-89: one base out, `used` bases of the read consumed -- the lane's only output
                    if (room) {
                        a.E[depth][eat + ne] = (i << 4) | (used << 2) | (uint32_t)(corr & 3ull);
                        ne++;
                    } else { // more fixes than the list holds: the read goes back to the group kernel
                        a.u_res[8ull * u + 6] = C_FAIL;
                        have = false;
                        want = true;
                    }
                }
                i += adv;
                wreg <<= 2u * adv;
                mw >>= adv;
                wcnt -= adv;
            }
        }
        // every lane back together: count the round's events
        n_probes += (uint32_t)__builtin_popcountll(__ballot(ev & 1u)) + (uint32_t)__builtin_popcountll(__ballot(ev & 16u));
        n_trig += (uint32_t)__builtin_popcountll(__ballot(ev & 2u));
        n_miss += (uint32_t)__builtin_popcountll(__ballot(ev & 8u));
    }
    // statistics: one atomic per wave per counter
    if ((threadIdx.x & 63) == 0) {
#ifdef BRX_LANE_TIMING
        if (a.dbg) {
            const unsigned long long at = atomicAdd(a.dbg, 1ull);
            if (at < 16384ull) {
                a.dbg[1 + 3 * at] = t_begin;
                a.dbg[2 + 3 * at] = wall_clock64();
                a.dbg[3 + 3 * at] = ((unsigned long long)n_rounds << 32) | w_iters;
            }
        }
#endif
        if (n_rounds)
            atomicAdd(p.ctrl + CTL_ROUNDS, (unsigned long long)n_rounds);
        if (n_probes)
            atomicAdd(p.ctrl + CTL_PROBES, (unsigned long long)n_probes);
        if (n_trig)
            atomicAdd(p.ctrl + CTL_TRIGGERS, (unsigned long long)n_trig);
        if (n_miss)
            atomicAdd(p.ctrl + CTL_LANE_MISS, (unsigned long long)n_miss);
    }
}

// logical bases [from, from + len) of a read of n bases -> dst
__device__ __forceinline__ void copy_bytes(uint8_t *dst, const uint8_t *in, uint32_t n_read, uint32_t from, uint32_t len, bool flip)
{
    // bytes up to the first 16-byte boundary of dst, then 16 bytes per lane (unaligned load, aligned store), then the tail
    uint32_t head = (uint32_t)((16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u);
    if (head > len)
        head = len;
    const uint32_t nv = (len - head) / 16u;
    for (uint32_t j = threadIdx.x; j < head; j += blockDim.x)
        dst[j] = ld_logical(in, n_read, from + j, flip);
    for (uint32_t v = threadIdx.x; v < nv; v += blockDim.x) {
        const uint32_t j = head + 16u * v;
        *reinterpret_cast<uint4 *>(dst + j) = ld16_logical(in, n_read, from + j, flip);
    }
    for (uint32_t j = head + 16u * nv + threadIdx.x; j < len; j += blockDim.x)
        dst[j] = ld_logical(in, n_read, from + j, flip);
}

// ---- successor table: next_nucs (mod.rs:118-128) of every indexed k-mer, worked out once --------------------------------
// A walk step asks for the FOUR successors of a solid k-mer (graph.rs:62, gap_size.rs:58) -- four rounds of a lane.  Which
// of them are solid is a property of the set alone, so it is tabulated beside the index: one byte per slot of every line,
// bits 0-2 for the slot's canonical k-mer as it is stored, bits 4-6 for its reverse complement: 4 | base if exactly one
// of add(x, 0..3) is solid, else 0.  A walk then needs ONE probe per step: find the slot of its k-mer, read the byte.
// k-mers that live outside the lines (turned away by a full line into the bit vector) have no byte: their steps are
// asked the long way.
__global__ __launch_bounds__(256) void succ_build_kernel(PassParams p, uint64_t n_lines, uint64_t *__restrict__ succ)
{
    const int k = p.k;
    const uint64_t mask = kmask(k);
    for (uint64_t line = (uint64_t)blockIdx.x * 256ull + threadIdx.x; line < n_lines; line += (uint64_t)gridDim.x * 256ull) {
        uint64_t out = 0;
        if (!p.idx.line_bits || ((p.idx.line_bits[line >> 5] >> (line & 31u)) & 1u)) {
            const uint64_t *L = p.idx.lines + line * 8ull;
            for (int sl = 0; sl < IDX_SLOTS; sl++) {
                const uint64_t v = L[sl];
                if (!v)
                    continue;
                const uint64_t half = (v - 1ull) << 1;                       // canonical >> 1, back in place
                const uint64_t cano = half | (uint64_t)(popc64(half) & 1);   // the even-popcount member (brx_kmer.hpp)
                uint32_t byte = 0;
                for (int o = 0; o < 2; o++) {
                    const uint64_t x = o ? revcomp(cano, k) : cano;
                    uint32_t m4 = 0;
                    for (uint64_t b = 0; b < 4; b++)
                        m4 |= (set_get<true>(p, add_nuc(x, b, mask), k) ? 1u : 0u) << b;
                    if (__popc(m4) == 1)
                        byte |= (4u | ((uint32_t)__ffs(m4) - 1u)) << (4 * o);
                }
                out |= (uint64_t)byte << (8 * sl);
            }
        }
        succ[line] = out;
    }
}

// ---- the automaton for the walking correctors: correct::Graph and correct::GapSize -----------------------------------
// Reference: Graph::correct_error (src/correct/graph.rs:44-85), error_len / alt_nucs / next_nucs (src/correct/mod.rs:
// 114-152), GapSize::correct_error and ins_sub_correction (src/correct/gap_size.rs:44-108; its `one` branch is
// Exist<ScenarioOne>, as in lane_kernel above).
//
// The same round -- one KmerSet::get, one transition -- with three more kinds of question:
//   ERRLEN  add(ek, seq[i0 + j]): how far the next solid k-mer is (mod.rs:130-152); the window is read on past the
//           trigger, a copy of it as it stood at the trigger is kept for what comes back to i0 (ALTS, the One branch, a
//           failed fix);
//   WALK    add(wk, b) for the four bases b in turn = next_nucs(wk) (mod.rs:118-128); exactly one solid successor makes a
//           step: the base joins the path, and the walk ends at first_correct_kmer (Graph) or after gap steps (GapSize);
//   and SCAN / ALTS / SCEN / MORE as before.
// Loop-top state.  All k-mers a walk emits are solid (they come out of next_nucs), so after any fix `previous` is
// get(kmer) here too and the scan can be cut into units at the same sync points.
// Graph's visited set (graph.rs:47, 71-75) only decides WHEN a walk that entered a cycle gives up: the walk is
// deterministic, so any cycle detector returns the same None -- Brent's here, O(1) state; a walk whose corrected k-mer
// already IS first_correct_kmer can only come back to it as a revisit, i.e. end in None, and is not walked at all (nor
// is one whose error_len ran off the read: its target is not solid and can never be reached).  GapSize's fixed-length
// walk needs the exact rule (a late detection could run past its end): its <= 31 steps are kept as 2-bit codes in a
// register and every earlier k-mer is rebuilt from them for the comparison; longer gaps go back to the group kernel.
// What a fix writes: an entry (position, bases of the read consumed, bases written) and the written bases, 16 per word.
enum { W_ST = 0, W_CUR = 3, W_JJ = 5, W_ACC = 8, W_PASS = 12, W_MODE = 15, W_HITEND = 17, W_DIRTY = 18, W_SKIP = 24, W_PREV = 27, W_FIRST = 28, W_SLOW = 29, W_ECLEAN = 30 };
enum { WS_SCAN = 0, WS_ALTS = 1, WS_SCEN = 2, WS_MORE = 3, WS_ERRLEN = 4, WS_WALK = 5, WS_WALK4 = 6 };
enum { WM_GRAPH = 0, WM_ONE = 1, WM_INSSUB = 2 };
// waves per SIMD the walking automata are compiled for: Graph needs 89 registers, GapSize (its One branch and the exact
// visited rule on top) more than the 102 of five waves
#ifndef BRX_WALK_WAVES_GRAPH
#define BRX_WALK_WAVES_GRAPH 5
#endif
#ifndef BRX_WALK_WAVES_GAP
#define BRX_WALK_WAVES_GAP 5
#endif
constexpr int walk_waves(int m) { return m == BRX_GRAPH ? BRX_WALK_WAVES_GRAPH : BRX_WALK_WAVES_GAP; }

template <bool IDX, int KT, int M>
__global__ __launch_bounds__(256, walk_waves(M)) void lane_walk_kernel(LaneArgs a)
{
    const PassParams &p = a.p;
    const int k = KT ? KT : p.k;
    const uint32_t c = (uint32_t)p.c; // 1 .. 5 (GapSize's One branch; Graph ignores it)
    const uint64_t mask = kmask(k);
    const unsigned long long n_units = p.ctrl[CTL_LANE_UNITS];
    __shared__ uint4 lc[4][256];
    const uint32_t tid = threadIdx.x;

    bool have = false, want = true;
    uint32_t u = 0, n = 0, i = 0, tgt = 0, t1 = U_END;
    uint64_t tgtk = 0;
    uint32_t depth = 0, ne = 0, ecap = 0, bw = 0, bw0 = 0;
    uint64_t eat = 0;
    uint64_t wreg = 0, wsave = 0;
    uint32_t wcnt = 0, nextw = 0, pidx = 0, wcnt_s = 0, nextw_s = 0, pidx_s = 0, pbase = 0;
    // the solidity of the ORIGINAL k-mers that end at the window's positions (lane_mask_kernel; lane_kernel has the
    // same): bit j of mw for position i + j, the next 16 in the low half of nextm (the high half: the copy kept at a
    // trigger, beside mw_s).  SCAN reads clean stretches off it, and error_len -- whose k-mers are all original when the
    // trigger was clean -- the distance to the next solid one.
    const bool use_mask = a.M != nullptr;
    uint32_t mw = 0, mw_s = 0, nextm = 0;
    uint64_t kmer = 0, corr = 0, wk = 0, fc = 0, tort = 0; // tort: Brent's tortoise (Graph) / the walked bases (GapSize)
    uint32_t S = 0, hop = 0, cline = 0xffffffffu;
    uint64_t csucc = 0; // the successor bytes of the line held in LDS
    uint32_t elen = 0, bpow = 1, blam = 0, np = 0, pacc = 0;
    uint32_t n_rounds = 0, n_probes = 0, n_trig = 0, n_miss = 0; // (fixes: counted by the replay kernel, as for One)
    uint32_t wnext = 0, wend = 0;

    for (;;) {
        uint32_t ev = 0;
        const bool at_end = have && (S & 7u) == 0u && i >= tgt;
        if (__any(at_end || want)) {
            if (at_end) {
                uint32_t *res = a.u_res + 8ull * u;
                res[2 * depth] = ne;
                res[2 * depth + 1] = t1;
                if (t1 == U_END || (i == tgt && kmer == tgtk)) {
                    res[6] = depth;
                    want = true;
                } else if (depth + 1 < (uint32_t)MAX_DEPTH) {
                    ev |= 8u;
                    depth++;
                    const uint4 *dp = reinterpret_cast<const uint4 *>(a.u_desc + t1);
                    const uint4 d0 = dp[0], d1 = dp[1], d2 = dp[2];
                    t1 = d0.z;
                    tgt = d0.w;
                    tgtk = ((uint64_t)d1.y << 32) | d1.x;
                    eat = ((uint64_t)d1.w << 32) | d1.z;
                    ecap = d2.z;
                    ne = 0;
                    bw = 0;
                } else {
                    res[6] = C_FAIL;
                    want = true;
                }
            }
            for (;;) {
                const uint64_t wm = __ballot(want);
                if (!wm)
                    break;
                if (wnext == wend) {
                    unsigned long long base = 0;
                    if ((tid & 63u) == 0u)
                        base = atomicAdd(p.ctrl + CTL_LANE_WORK, (unsigned long long)LANE_GRAB);
                    base = __shfl(base, 0);
                    wnext = base < n_units ? (uint32_t)base : (uint32_t)n_units;
                    wend = base + LANE_GRAB < n_units ? (uint32_t)(base + LANE_GRAB) : (uint32_t)n_units;
                    if (wnext == wend) {
                        if (want)
                            have = false;
                        want = false;
                        break;
                    }
                }
                const uint32_t rank = (uint32_t)__builtin_popcountll(wm & ((1ull << (tid & 63u)) - 1ull));
                const uint32_t avail = wend - wnext;
                const uint32_t asked = (uint32_t)__builtin_popcountll(wm);
                const bool take = want && rank < avail;
                const uint32_t my = wnext + rank;
                wnext += asked < avail ? asked : avail;
                if (take) {
                    want = false;
                    have = true;
                    u = my;
                    const uint4 *dp = reinterpret_cast<const uint4 *>(a.u_desc + u);
                    const uint4 d0 = dp[0], d1 = dp[1], d2 = dp[2];
                    n = d0.x;
                    const uint32_t q = d0.y;
                    if (q >= U_VOID - 1u) {
                        a.u_res[8ull * u + 6] = q == U_VOID ? C_VOID : C_FAIL;
                        have = false;
                        want = true;
                    } else {
                        t1 = d0.z;
                        tgt = d0.w;
                        tgtk = ((uint64_t)d1.y << 32) | d1.x;
                        eat = ((uint64_t)d1.w << 32) | d1.z;
                        const uint64_t pw = ((uint64_t)d2.y << 32) | d2.x;
                        ecap = d2.z;
                        depth = 0;
                        ne = 0;
                        bw = 0;
                        hop = 0;
                        pbase = (uint32_t)pw - (q >> 4); // the read's first dword in P
                        wreg = (((uint64_t)a.P[pw] << 32) | a.P[pw + 1]) << ((q & 15u) * 2u);
                        wcnt = 32u - (q & 15u);
                        nextw = a.P[pw + 2];
                        pidx = (uint32_t)pw + 3u;
                        if (use_mask) {
                            mw = ((uint32_t)a.M[pw] | ((uint32_t)a.M[pw + 1] << 16)) >> (q & 15u);
                            nextm = a.M[pw + 2];
                        }
                        if (d2.w) {
                            if (n < (uint32_t)k) {
                                uint32_t *res = a.u_res + 8ull * u;
                                res[0] = 0;
                                res[1] = U_END;
                                res[6] = 0;
                                have = false;
                                want = true;
                            } else {
                                kmer = wreg >> (64 - 2 * k);
                                wreg <<= 2 * k;
                                mw >>= k;
                                wcnt -= (uint32_t)k;
                                i = (uint32_t)k;
                                S = 1u << W_FIRST;
                            }
                        } else {
                            i = q;
                            kmer = a.u_qk[u];
                            S = 1u << W_PREV;
                        }
                    }
                }
            }
            if (!__any(have))
                break;
        }
        const bool act = have && !((S & 7u) == 0u && i >= tgt);
        n_rounds += (uint32_t)__builtin_popcountll(__ballot(act));
        if (act) {
            if (wcnt <= 16u) {
                wreg |= (uint64_t)nextw << (32u - 2u * wcnt);
                mw |= (nextm & 0xffffu) << wcnt;
                wcnt += 16u;
                nextw = a.P[pidx];
                if (use_mask)
                    nextm = (nextm & 0xffff0000u) | a.M[pidx];
                pidx++;
            }
            const uint32_t st = S & 7u, cur = (S >> W_CUR) & 3u, jj = (S >> W_JJ) & 7u, skip = (S >> W_SKIP) & 7u;
            // (Graph knows its mode at compile time: the One branch and the fixed-length walk fall out of its kernel)
            const uint32_t mode = M == BRX_GRAPH ? (uint32_t)WM_GRAPH : ((S >> W_MODE) & 3u);
            const bool prev = (S >> W_PREV) & 1u, first = (S >> W_FIRST) & 1u, slow = (S >> W_SLOW) & 1u;
            const bool is_scan = st == WS_SCAN, is_alts = st == WS_ALTS, is_scen = st == WS_SCEN, is_more = st == WS_MORE,
                       is_errlen = st == WS_ERRLEN, is_wfast = st == WS_WALK, is_walk = st == WS_WALK4;
            // SCAN and ERRLEN read the window where it is; everything else looks at the bases from the trigger on
            const bool live = is_scan || is_errlen;
            const uint32_t cw = (uint32_t)((live ? wreg : wsave) >> 48);
            const uint32_t c0 = cw >> 14;
            const uint32_t rem = n - i; // (i stays at the trigger until it is resolved)

            // ---- the k-mer this state asks about ------------------------------------------------------------------------
            const uint32_t off = 2u - cur;
            const uint32_t nb = (is_scan ? (first ? 0u : 1u) : (is_errlen ? 1u : (is_scen ? jj + 1u : (is_more ? c + 1u : 0u))));
            const uint32_t b0 = (is_scen || is_more) ? off : 0u;
            const uint32_t wbits = (cw >> (16u - 2u * (b0 + nb))) & ((1u << (2u * nb)) - 1u);
            const uint64_t base_k = is_scan ? kmer : ((is_errlen || is_walk || is_wfast) ? wk : corr);
            uint64_t pk = ((base_k << (2u * nb)) | (uint64_t)wbits) & mask;
            pk = is_alts ? ((pk & ~3ull) | (uint64_t)cur) : (is_walk ? (((pk << 2) & mask) | (uint64_t)cur) : pk);
            // error_len stops without a probe where the read ends (mod.rs:137-139)
            const bool el_end = is_errlen && elen + 1u >= rem;
            // SCAN over the read's own bases, and error_len behind a trigger that was: the answer is in the mask
            const uint32_t dirty = (S >> W_DIRTY) & 63u;
            const bool masked = use_mask && !slow && hop == 0u &&
                                ((is_scan && !first && skip == 0u && dirty == 0u) || (is_errlen && ((S >> W_ECLEAN) & 1u)));
            const bool need = !(is_scan && skip != 0u) && !(is_more && !(rem > c + off + 1u)) && !el_end && !masked;
            ev |= need ? 1u : 0u;

            // ---- KmerSet::get -------------------------------------------------------------------------------------------
            bool sol = masked ? (mw & 1u) != 0u : (is_scan && !need), unres = false;
            uint32_t slot = 7u; // the slot of the line the k-mer was found in (7: not in a line's slot)
            if (IDX) {
                uint64_t key;
                const uint32_t home = index_locate(p.idx, pk, k, key);
                const uint32_t line = (home + hop) & (0xffffffffu >> p.idx.line_shift);
                if (need && !slow && line != cline) {
                    const uint8_t *L = reinterpret_cast<const uint8_t *>(p.idx.lines + (uint64_t)line * 8ull);
                    const uint32_t wb = tid & ~63u;
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(L + 16 * e),
                                                         (__attribute__((address_space(3))) void *)&lc[e][wb], 16, 0, 0);
                    if (a.succ)
                        csucc = a.succ[line];
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    cline = line;
                }
                if (need && slow) {
                    const uint64_t h = key - 1ull;
                    sol = (p.bits[h >> 5] >> (h & 31u)) & 1u;
                } else if (need) {
                    const uint4 q0 = lc[0][tid], q1 = lc[1][tid], q2 = lc[2][tid], q3 = lc[3][tid];
                    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
                    const bool f0 = (q0.x == klo) & (q0.y == khi), f1 = (q0.z == klo) & (q0.w == khi), f2 = (q1.x == klo) & (q1.y == khi),
                               f3 = (q1.z == klo) & (q1.w == khi), f4 = (q2.x == klo) & (q2.y == khi), f5 = (q2.z == klo) & (q2.w == khi),
                               f6 = (q3.x == klo) & (q3.y == khi);
                    const bool found = f0 | f1 | f2 | f3 | f4 | f5 | f6;
                    slot = f0 ? 0u : (f1 ? 1u : (f2 ? 2u : (f3 ? 3u : (f4 ? 4u : (f5 ? 5u : (f6 ? 6u : 7u))))));
                    const uint32_t hdr_hi = q3.w;
                    const bool more = (hdr_hi >> 31) && (hop != 0u || ((hdr_hi >> idx_sig_index(key)) & 1u));
                    sol = found;
                    unres = !found && more;
                }
            } else if (need) {
                sol = probe(p.bits, pk, k);
            }
            if (unres) {
                if (p.bits)
                    S |= 1u << W_SLOW;
                else
                    hop++;
            } else {
                hop = 0;
                // ---- transition -------------------------------------------------------------------------------------------------
                const uint32_t solb = sol ? 1u : 0u, above = ~((2u << cur) - 1u);
                const uint32_t acc = (S >> W_ACC) & 15u, passm = (S >> W_PASS) & 7u;
                const bool hit_end0 = (S >> W_HITEND) & 1u;
                const bool accept = is_scan && (sol || !prev), trig = is_scan && !accept; // mod.rs:73, :99-102
                // masked rounds take a whole run: SCAN the positions with the same answer (`previous` ends as that answer),
                // error_len the positions that are not solid -- as far as the window, the unit's target (SCAN) and the end
                // of the read (error_len stops there without a probe, mod.rs:137-139) allow
                uint32_t run = 1u;
                if (masked) {
                    const uint32_t same = (mw & 1u) ? ~mw : mw;
                    run = same ? (uint32_t)__builtin_ctz(same) : 32u;
                    const uint32_t lim_r = is_scan ? tgt - i : rem - 1u - elen; // (error_len: >= 1 unless el_end)
                    run = run < MASK_STEP ? run : MASK_STEP;
                    run = run < lim_r ? run : lim_r;
                    run = run < wcnt ? run : wcnt;
                    run = run ? run : 1u;
                }
                const uint64_t krun = (((is_scan ? kmer : wk) << (2u * run)) | (wreg >> (64u - 2u * run))) & mask;
                // error_len (mod.rs:130-152): the first solid k-mer behind the trigger, or the end of the read
                const bool el_found = is_errlen && !el_end && sol;
                const bool el_done = el_end || el_found;
                const bool el_run = is_errlen && masked && !sol && !el_end; // a run of k-mers that are not solid
                const uint32_t elen1 = is_errlen ? elen + (el_run ? run : 1u) : elen; // j of this iteration (of its last step)
                const uint32_t mode_new = M == BRX_GRAPH ? (uint32_t)WM_GRAPH
                                                         : (elen1 < (uint32_t)k ? (uint32_t)WM_GRAPH : (elen1 == (uint32_t)k ? (uint32_t)WM_ONE : (uint32_t)WM_INSSUB)); // gap_size.rs:97-108
                // a graph walk towards a k-mer that is not solid can only end in None (see above)
                const bool el_fail = el_done && mode_new == (uint32_t)WM_GRAPH && el_end;
                const bool to_alts = el_done && !el_fail;
                // the candidate loop of ALTS / SCEN / MORE / WALK
                // WALK, the short way: the k-mer's own byte of the successor table says whether exactly one successor is solid
                // and which; a k-mer found outside the lines' slots (or no table) asks the four successors one by one (WALK4)
                const uint32_t sb = (uint32_t)(csucc >> (8u * slot + ((popc64(wk) & 1) ? 4u : 0u))) & 7u;
                const bool wf_known = is_wfast && a.succ != nullptr && slot < 7u;
                const bool wf_step = wf_known && (sb & 4u) != 0u, wf_dead = wf_known && (sb & 4u) == 0u;
                const bool to_walk4 = is_wfast && !wf_known;
                const uint32_t cands = (is_alts ? (15u & ~(1u << c0)) : (is_scen ? 7u : (is_walk ? 15u : passm))) & above;
                const bool s_pass = is_scen && sol && jj + 1u == c;
                const bool s_over = is_scen && (!sol || s_pass);
                const uint32_t acc1 = acc | ((is_alts || is_more || is_walk) ? solb << cur : 0u);
                const uint32_t pass1 = passm | (s_pass ? 1u << cur : 0u);
                const bool adv_c = is_alts || is_more || is_walk || s_over;
                const bool end = adv_c && (cands == 0u || ((is_alts || is_walk) && __popc(acc1) > 1));
                const uint32_t dm = is_scen ? pass1 : acc1;
                const int pc = __popc(dm);
                const uint32_t win = wf_step ? (sb & 3u) : (uint32_t)__ffs(dm) - 1u;
                const uint32_t smin = rem >= c + 2u ? 0u : (rem == c + 1u ? 1u : (rem == c ? 2u : 3u));
                const bool alts_ok = end && is_alts && pc == 1;
                const uint64_t corr_alt = (corr & ~3ull) | (uint64_t)win;
                const bool to_scen = alts_ok && mode == (uint32_t)WM_ONE && smin < 3u;
                // graph.rs:57-59 / gap_size.rs:52-55: path = [alt]; a corrected k-mer that already is the target is never
                // compared with it on entry and can only be met again as a revisit: None
                const bool to_walk = alts_ok && mode != (uint32_t)WM_ONE && !(mode == (uint32_t)WM_GRAPH && corr_alt == fc);
                const bool to_more = end && is_scen && pc > 1;
                const bool apply_one = end && (is_scen || is_more) && pc == 1;
                // one walk step (graph.rs:61-82, gap_size.rs:57-85)
                const bool step = (end && is_walk && pc == 1) || wf_step;
                const uint64_t nk = ((wk << 2) & mask) | (uint64_t)win;
                // GapSize's fixed-length walk (gap_size.rs:57-85) is not checked for revisits HERE: it ends after `gap` steps
                // whatever it meets, a walk that would have been cut short by its visited set either fails later anyway (None
                // both ways) or comes out with a repeated k-mer in its path -- which the replay kernel finds (every such fix
                // is marked and carries its first k-mer), handing the read back to the group kernel
                bool revisit = false;
                const uint32_t blam1 = blam + 1u;
                if (step && mode == (uint32_t)WM_GRAPH)
                    revisit = nk == tort;
                const bool step_ok = step && !revisit;
                const bool graph_done = step_ok && mode == (uint32_t)WM_GRAPH && nk == fc;          // graph.rs:79-81
                const bool gap_done = step_ok && mode == (uint32_t)WM_INSSUB && elen == 1u;        // (elen counts the steps left)
                const bool too_long = step_ok && ((mode == (uint32_t)WM_INSSUB && np >= AP_VERIFY - 1u) || np >= 65000u);
                const bool walk_done = graph_done || gap_done;
                const bool fail = el_fail || (end && is_alts && !to_scen && !to_walk) || (end && is_scen && pc == 0) ||
                                  (end && is_more && pc != 1) || (end && is_walk && !step_ok) || wf_dead || (wf_step && !step_ok);
                // ---- what a successful fix writes -------------------------------------------------------------------------
                const uint32_t used_one = 2u - win;
                const bool jump = tgt - i > used_one + c;
                const uint32_t cb = (cw >> (16u - 2u * (used_one + c))) & ((1u << (2u * c)) - 1u);
                // bases of the read a walk consumes: graph.rs:84 error_len + 1; gap_size.rs:87-88 the path's own length
                const uint32_t np1 = np + 1u;
                const uint32_t used_walk = mode == (uint32_t)WM_GRAPH ? elen + 1u : np1;
                const uint32_t words_fix = apply_one ? 1u : (np1 + 15u) / 16u + (mode == (uint32_t)WM_INSSUB ? 2u : 0u);
                const bool room = ne < (ecap >> 1) && (apply_one ? bw : bw0) + words_fix <= ecap && used_walk < 65536u;
                const bool fix_one = apply_one && room, fix_walk = walk_done && room;
                const bool give_up = ((apply_one || walk_done) && !room) || too_long;
                // ---- the path: 2-bit codes, 16 per word, first base in the top bits ------------------------------------------
                if (to_walk) { // path = [alt]
                    np = 1;
                    pacc = win;
                    bw0 = bw;
                    if (mode == (uint32_t)WM_INSSUB) { // the first k-mer of the path, for the replay kernel's check
                        if (bw + 2u <= ecap) {
                            a.BW[depth][eat + bw] = (uint32_t)corr_alt;
                            a.BW[depth][eat + bw + 1u] = (uint32_t)(corr_alt >> 32);
                        }
                        bw += 2u;
                    }
                }
                if (step_ok) {
                    pacc = (pacc << 2) | win;
                    np = np1;
                    if ((np1 & 15u) == 0u) {
                        if (bw < ecap)
                            a.BW[depth][eat + bw] = pacc;
                        bw++;
                    }
                }
                if (fix_walk || fix_one) {
                    const uint32_t cnt = fix_one ? 1u : np;
                    const uint32_t usd = fix_one ? used_one : used_walk;
                    if (fix_one) {
                        a.BW[depth][eat + bw] = (uint32_t)(corr & 3ull) << 30;
                        bw++;
                    } else if (np & 15u) {
                        a.BW[depth][eat + bw] = pacc << (2u * (16u - (np & 15u)));
                        bw++;
                    }
                    a.EW[depth][(eat >> 1) + ne] = make_uint2(i | ((fix_walk && mode == (uint32_t)WM_INSSUB) ? 0x80000000u : 0u), (usd << 16) | cnt);
                    ne++;
                }
                if (fail && (is_walk || is_wfast))
                    bw = bw0; // a failed walk leaves no bases behind
                if (give_up) { // more than a list holds / a gap the register cannot remember: back to the group kernel
                    a.u_res[8ull * u + 6] = C_FAIL;
                    have = false;
                    want = true;
                }
                // ---- the new state ---------------------------------------------------------------------------------------------
                const bool resolved = fail || fix_one || fix_walk;
                const uint32_t st1 = trig ? (uint32_t)WS_ERRLEN
                                          : (to_alts ? (uint32_t)WS_ALTS
                                                     : (to_scen ? (uint32_t)WS_SCEN
                                                                : (to_more ? (uint32_t)WS_MORE
                                                                           : (resolved ? (uint32_t)WS_SCAN
                                                                                       : (to_walk4 ? (uint32_t)WS_WALK4 : ((to_walk || step_ok) ? (uint32_t)WS_WALK : st))))));
                const uint32_t c0s = (uint32_t)(wsave >> 62); // the trigger base (ALTS starts at the first base that is not it)
                const uint32_t cur1 = to_alts ? ((trig ? c0 : c0s) == 0u ? 1u : 0u)
                                              : (to_scen ? smin : (to_more ? win : ((to_walk || step_ok || to_walk4) ? 0u : ((adv_c && !end) ? (uint32_t)__ffs(cands) - 1u : cur))));
                const uint32_t jj1 = (is_scen && !s_over) ? jj + 1u : 0u;
                const uint32_t accn = (to_alts || to_more || to_walk || step_ok || to_walk4) ? 0u : acc1, passn = to_scen ? 0u : pass1;
                const uint32_t skipn = accept ? (skip ? skip - 1u : 0u) : ((fix_one && !jump) ? c : skip);
                const uint32_t prevn = accept ? solb : (fail ? 0u : ((fix_one || fix_walk) ? 1u : (prev ? 1u : 0u)));
                const uint32_t moden = el_done ? mode_new : mode;
                const uint32_t hitn = el_done ? (el_end ? 1u : 0u) : (hit_end0 ? 1u : 0u);
                // positions ahead whose k-mers hold a corrected base.  One's fix: k - 1 (its jump taken off).  A finished
                // Graph walk ends ON first_correct_kmer, the k-mer error_len found: what was dirty in front of the trigger
                // is as dirty behind it, less the bases passed.  GapSize's fixed-length walk ends on a walked k-mer: k - 1.
                const bool clean_scan = is_scan && masked;
                const uint32_t kd = (uint32_t)k - 1u, jd = jump ? c : 0u;
                const uint32_t passed = accept ? (first ? 0u : (clean_scan ? run : 1u)) : (fail ? 1u : (fix_walk ? used_walk : 0u));
                const uint32_t dirtyn = fix_one ? (kd > jd ? kd - jd : 0u)
                                                : ((fix_walk && mode == (uint32_t)WM_INSSUB) ? kd : (dirty > passed ? dirty - passed : 0u));
                const uint32_t ecleann = trig ? (masked ? 1u : 0u) : ((S >> W_ECLEAN) & 1u);
                S = st1 | (cur1 << W_CUR) | (jj1 << W_JJ) | (accn << W_ACC) | (passn << W_PASS) | (moden << W_MODE) | (hitn << W_HITEND) |
                    (dirtyn << W_DIRTY) | (skipn << W_SKIP) | (prevn << W_PREV) | (ecleann << W_ECLEAN);
                // Brent's tortoise (Graph) / the walked bases (GapSize's fixed-length walk)
                if (to_walk) {
                    tort = mode == (uint32_t)WM_GRAPH ? corr_alt : 0ull;
                    bpow = 1;
                    blam = 0;
                }
                if (step_ok) {
                    if (mode == (uint32_t)WM_GRAPH) {
                        const bool hop_t = blam1 == bpow;
                        tort = hop_t ? nk : tort;
                        bpow = hop_t ? bpow * 2u : bpow;
                        blam = hop_t ? 0u : blam1;
                    }
                }
                // error_len's counter, then (GapSize's long gaps) the steps left: gap_size.rs:107, for i in 0..gap_size
                elen = trig ? 0u : (is_errlen ? (el_done ? (mode_new == (uint32_t)WM_INSSUB ? elen1 - (uint32_t)k : elen1) : elen1)
                                             : ((step_ok && mode == (uint32_t)WM_INSSUB) ? elen - 1u : elen));
                fc = el_done ? (el_end ? wk : pk) : fc;
                // the k-mers
                const uint64_t kfix = jump ? (((corr << (2u * c)) | (uint64_t)cb) & mask) : corr;
                kmer = accept ? (clean_scan ? krun : pk) : (fail ? ((corr & ~3ull) | (uint64_t)(trig ? c0 : c0s)) : (fix_one ? kfix : (fix_walk ? nk : kmer)));
                wk = trig ? pk : ((is_errlen && !el_done) ? (el_run ? krun : pk) : (to_walk ? corr_alt : (step_ok ? nk : wk)));
                corr = trig ? pk : (alts_ok ? corr_alt : corr);
                ev |= trig ? 2u : 0u;
                // ---- the window ---------------------------------------------------------------------------------------------------
                // SCAN accepts and ERRLEN steps read on; a trigger keeps a copy of the window as it stands (at i0) and steps past
                // the trigger base; what resolves a trigger goes back to the copy -- except a finished Graph walk, which goes on
                // exactly where error_len stopped reading (i0 + error_len + 1)
                if (trig) {
                    wsave = wreg;
                    wcnt_s = wcnt;
                    nextw_s = nextw;
                    pidx_s = pidx;
                    mw_s = mw;
                    nextm = (nextm & 0xffffu) | (nextm << 16);
                }
                const bool back = fail || fix_one; // back to i0, then forward
                if (back) {
                    wreg = wsave;
                    wcnt = wcnt_s;
                    nextw = nextw_s;
                    pidx = pidx_s;
                    mw = mw_s;
                    nextm = (nextm & 0xffff0000u) | (nextm >> 16);
                }
                uint32_t adv = accept ? (first ? 0u : (clean_scan ? run : 1u))
                                      : (trig ? 1u : ((is_errlen && !el_end) ? (el_run ? run : 1u) : (fail ? 1u : (fix_one ? used_one + (jump ? c : 0u) : 0u))));
                const uint32_t iadv = accept ? adv : (fail ? 1u : (fix_one ? adv : (fix_walk ? used_walk : 0u)));
                i += iadv;
                if (fix_walk && mode == (uint32_t)WM_INSSUB) {
                    // gap_size.rs:87-88: the scan goes on at i0 + path length, which lies BEHIND what error_len read: reload
                    const uint32_t pw = pbase + (i >> 4);
                    wreg = (((uint64_t)a.P[pw] << 32) | a.P[pw + 1]) << ((i & 15u) * 2u);
                    wcnt = 32u - (i & 15u);
                    nextw = a.P[pw + 2];
                    pidx = pw + 3u;
                    if (use_mask) {
                        mw = ((uint32_t)a.M[pw] | ((uint32_t)a.M[pw + 1] << 16)) >> (i & 15u);
                        nextm = (nextm & 0xffff0000u) | a.M[pw + 2];
                    }
                    adv = 0;
                }
                wreg <<= 2u * adv;
                mw >>= adv;
                wcnt -= adv;
            }
        }
        n_probes += (uint32_t)__builtin_popcountll(__ballot(ev & 1u));
        n_trig += (uint32_t)__builtin_popcountll(__ballot(ev & 2u));
        n_miss += (uint32_t)__builtin_popcountll(__ballot(ev & 8u));
    }
    if ((threadIdx.x & 63) == 0) {
        if (n_rounds)
            atomicAdd(p.ctrl + CTL_ROUNDS, (unsigned long long)n_rounds);
        if (n_probes)
            atomicAdd(p.ctrl + CTL_PROBES, (unsigned long long)n_probes);
        if (n_trig)
            atomicAdd(p.ctrl + CTL_TRIGGERS, (unsigned long long)n_trig);
        if (n_miss)
            atomicAdd(p.ctrl + CTL_LANE_MISS, (unsigned long long)n_miss);
    }
}

// ---- apply: the fixes of a read's chain of units replayed over the input -> the read's staging slot -----------------
// Which unit's list is the truth up to where follows from the records the units left (lane 0 walks the chain); the
// lists, concatenated, are the read's fixes in scan order, and the output is the input with them applied (mod.rs:75-102:
// a fix writes one base and consumes `used` bases of the read; everything else is copied through).  Copying is
// OUTPUT-centric: every thread produces 16 aligned output bytes, finds the fix its first byte lies behind by bisection
// of the fixes' output offsets (LDS), and in the common case -- no fix inside its 16 bytes -- moves them as one vector.

__global__ __launch_bounds__(AP_BS) void lane_apply_kernel(LaneArgs a)
{
    const PassParams &p = a.p;
    __shared__ uint32_t e_raw[AP_EDITS];     // the fixes of the batch
    __shared__ uint32_t e_os[AP_EDITS + 1];  // output offset (inside the batch) of the copied stretch in front of fix m
    __shared__ uint32_t e_in[AP_EDITS];      // input position where that stretch starts
    __shared__ uint64_t pc_src[AP_PIECES];   // pieces: where their fixes are (list index << 60 | entry)
    __shared__ uint32_t pc_off[AP_PIECES + 1]; // ... and how many came before
    __shared__ uint32_t sh_part[AP_BS / 64];
    __shared__ uint32_t sh_np, sh_next_u, sh_state;
    // the records of the read's first 256 units, loaded side by side: the chain is walked by ONE lane, and every
    // dependent trip to global memory it makes is a microsecond the other 255 wait
    __shared__ uint4 sh_ra[AP_BS], sh_rb[AP_BS];
    __shared__ uint64_t sh_eat[AP_BS];
    for (uint32_t r = blockIdx.x; r < p.n_reads; r += gridDim.x) {
        if (p.in_staged && p.in_lens[r] == 0xffffffffu) { // given up by an earlier pass of this attempt: stays poisoned
            if (threadIdx.x == 0)
                p.out_lens[r] = 0xffffffffu;
            continue;
        }
        const uint8_t *in;
        uint32_t n;
        bool poisoned;
        const uint64_t in_at = read_view(p, r, in, n, poisoned);
        const uint64_t o0 = p.offsets[r], o1 = p.offsets[r + 1];
        const uint64_t s0 = slot_of(o0, r, p.slack), s1 = slot_of(o1, (uint64_t)r + 1, p.slack);
        uint8_t *dst = p.out + s0;
        const uint64_t slot = s1 - s0;
        const uint32_t ub = (uint32_t)a.ubase[r];
        uint64_t total = 0;  // output bytes so far (uniform)
        uint32_t cur_in = 0; // input bases consumed so far (uniform)
        bool failed = false, more = true;
        uint32_t u_next = ub, d_next = 0; // where the chain walk goes on (uniform)
        uint32_t committed = 0;           // fixes replayed so far (uniform)
        __syncthreads();
        if (ub + threadIdx.x < (uint32_t)a.ubase[r + 1]) {
            const uint32_t uu = ub + threadIdx.x;
            sh_ra[threadIdx.x] = *reinterpret_cast<const uint4 *>(a.u_res + 8ull * uu);
            sh_rb[threadIdx.x] = *reinterpret_cast<const uint4 *>(a.u_res + 8ull * uu + 4);
            const uint4 d1 = reinterpret_cast<const uint4 *>(a.u_desc + uu)[1];
            sh_eat[threadIdx.x] = ((uint64_t)d1.w << 32) | d1.z;
        }
        while (more && !failed) {
            // ---- gather pieces until the batch is full (lane 0; the others wait) ----------------------------------------
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t np = 0, cnt = 0, u = u_next, d = d_next, state = 1; // state: 0 chain done, 1 more to come, 2 failed
                pc_off[0] = 0;
                for (;;) {
                    const bool near = u - ub < AP_BS;
                    const uint4 ra = near ? sh_ra[u - ub] : *reinterpret_cast<const uint4 *>(a.u_res + 8ull * u);
                    const uint4 rb = near ? sh_rb[u - ub] : *reinterpret_cast<const uint4 *>(a.u_res + 8ull * u + 4);
                    const uint32_t res[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
                    const uint32_t code = res[6];
                    if (code >= (uint32_t)MAX_DEPTH) { // FAIL, or a void unit on the chain (cannot be)
                        if (code == U_UNWRITTEN)
                            atomicAdd(p.ctrl + CTL_LANE_UNWRITTEN, 1ull);
                        state = 2;
                        break;
                    }
                    bool full = false;
                    for (; d <= code; d++) {
                        const uint32_t ne = res[2 * d];
                        if (np == AP_PIECES || (cnt + ne > AP_EDITS && cnt != 0u)) {
                            full = true;
                            break;
                        }
                        const uint32_t xu = d == 0 ? u : res[2 * d - 1];
                        pc_src[np] = ((uint64_t)d << 60) | (xu - ub < AP_BS ? sh_eat[xu - ub] : edit_start(in_at, r, ub, xu - ub, a.u_q[xu]));
                        cnt += ne; // (a single piece may hold more than a batch: it is then replayed in several)
                        np++;
                        pc_off[np] = cnt;
                    }
                    if (full)
                        break;
                    const uint32_t next = res[2 * code + 1];
                    if (next == U_END) {
                        state = 0;
                        break;
                    }
                    u = next;
                    d = 0;
                }
                sh_np = np;
                sh_next_u = u;
                sh_state = state | (d << 8);
            }
            __syncthreads();
            const uint32_t np = sh_np;
            u_next = sh_next_u;
            d_next = sh_state >> 8;
            failed = (sh_state & 0xffu) == 2u;
            more = (sh_state & 0xffu) == 1u;
            if (failed)
                break;
            const uint32_t n_all = pc_off[np];
            // ---- replay the gathered fixes, AP_EDITS at a time ---------------------------------------------------------
            for (uint32_t f0 = 0; f0 < n_all; f0 += AP_EDITS) {
                const uint32_t nb = n_all - f0 < AP_EDITS ? n_all - f0 : AP_EDITS;
                __syncthreads();
                for (uint32_t t = threadIdx.x; t < nb; t += AP_BS) {
                    const uint32_t f = f0 + t;
                    uint32_t lo = 0, hi = np; // the piece of fix f: pc_off[lo] <= f < pc_off[lo + 1]
                    while (hi - lo > 1u) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (pc_off[mid] <= f)
                            lo = mid;
                        else
                            hi = mid;
                    }
                    const uint64_t src = pc_src[lo];
                    e_raw[t] = a.E[src >> 60][(src & 0x0fffffffffffffffull) + (f - pc_off[lo])];
                }
                __syncthreads();
                // stretch in front of fix t: from the end of the fix before (cur_in for the first) to its position
                const uint32_t per = (nb + AP_BS - 1u) / AP_BS; // consecutive fixes per thread
                const uint32_t t0 = threadIdx.x * per, t1 = t0 + per < nb ? t0 + per : nb;
                uint32_t part = 0;
                for (uint32_t t = t0; t < t1; t++) {
                    const uint32_t e = e_raw[t];
                    const uint32_t prev_end = t == 0 ? cur_in : (e_raw[t - 1] >> 4) + ((e_raw[t - 1] >> 2) & 3u);
                    e_in[t] = prev_end;
                    part += ((e >> 4) - prev_end) + 1u;
                }
                uint32_t inc = part; // inclusive scan of the 256 partial sums: inside the wave by shuffles, then the four waves
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t v = __shfl_up(inc, o);
                    if ((int)(threadIdx.x & 63u) >= o)
                        inc += v;
                }
                if ((threadIdx.x & 63u) == 63u)
                    sh_part[threadIdx.x >> 6] = inc;
                __syncthreads();
                uint32_t run = inc - part;
                for (uint32_t q = 0; q < (threadIdx.x >> 6); q++)
                    run += sh_part[q];
                uint32_t batch_total = 0;
                for (uint32_t q = 0; q < AP_BS / 64u; q++)
                    batch_total += sh_part[q];
                for (uint32_t t = t0; t < t1; t++) {
                    e_os[t] = run;
                    run += ((e_raw[t] >> 4) - e_in[t]) + 1u;
                }
                if (threadIdx.x == 0)
                    e_os[nb] = batch_total;
                __syncthreads();
                // output bytes [total, total + batch_total), 16 per thread, aligned to the destination
                if (total + batch_total <= slot) {
                    uint8_t *ob = dst + total;
                    const uint32_t head = (uint32_t)((16u - (uint32_t)((uintptr_t)ob & 15u)) & 15u);
                    const uint32_t n_chunks = (batch_total + (16u - head) % 16u + 15u) / 16u + 1u;
                    for (uint32_t ch = threadIdx.x; ch < n_chunks; ch += AP_BS) {
                        // chunk 0 = the bytes in front of the first 16-byte boundary
                        const uint32_t x0 = ch == 0 ? 0u : head + 16u * (ch - 1u);
                        uint32_t x1 = ch == 0 ? head : x0 + 16u;
                        if (x1 > batch_total)
                            x1 = batch_total;
                        if (x0 >= x1)
                            continue;
                        uint32_t lo = 0, hi = nb; // the fix whose stretch holds byte x0: e_os[lo] <= x0 < e_os[lo + 1]
                        while (hi - lo > 1u) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (e_os[mid] <= x0)
                                lo = mid;
                            else
                                hi = mid;
                        }
                        uint32_t m = lo;
                        const uint32_t copied_end = e_os[m + 1] - 1u; // the fix's own base sits here
                        if (x1 - x0 == 16u && x1 <= copied_end) {
                            *reinterpret_cast<uint4 *>(ob + x0) = ld16_logical(in, n, e_in[m] + (x0 - e_os[m]), p.flip);
                        } else {
                            for (uint32_t x = x0; x < x1; x++) {
                                while (x >= e_os[m + 1])
                                    m++;
                                const uint32_t rel = x - e_os[m];
                                ob[x] = x + 1u == e_os[m + 1] ? bit2nuc(e_raw[m] & 3u) : ld_logical(in, n, e_in[m] + rel, p.flip);
                            }
                        }
                    }
                }
                total += batch_total;
                committed += nb;
                cur_in = (e_raw[nb - 1] >> 4) + ((e_raw[nb - 1] >> 2) & 3u);
            }
        }
        // the bases behind the last fix
        if (!failed && n > cur_in) {
            const uint32_t len = n - cur_in;
            if (total + len <= slot)
                copy_bytes(dst + total, in, n, cur_in, len, p.flip);
            total += len;
        }
        if (threadIdx.x == 0) {
            if (failed) {
                const unsigned long long at = atomicAdd(p.ctrl + CTL_LANE_FAIL, 1ull);
                a.fail_list[at] = r;
            } else if (total + 1u > slot) { // the read outgrew its slot: poisoned, redone with more slack (brx_correct.hip)
                p.out_lens[r] = 0xffffffffu;
                atomicAdd(p.ctrl + CTL_OVERFLOW, 1ull);
            } else {
                p.out_lens[r] = (uint32_t)total;
                // the fixes of the pass (brx_chain_last_stats): the ones written into a read that is kept.  A read handed
                // back to the group kernel is counted there, one redone with more slack by the chain that redoes it.
                if (committed)
                    atomicAdd(p.ctrl + CTL_FIXES, (unsigned long long)committed);
            }
        }
    }
}

// The same replay for the walking correctors' lists: a fix consumes `used` bases of the read and writes `cnt` bases that
// sit, 16 per word, in the list of bases of its unit (a fix starts a word; the word it starts at is the running sum of the
// words of the unit's earlier fixes, taken here from a scan over the batch).
__global__ __launch_bounds__(256) void lane_apply_walk_kernel(LaneArgs a)
{
    const PassParams &p = a.p;
    __shared__ uint32_t e_pos[APW_EDITS], e_uc[APW_EDITS], e_pi[APW_EDITS]; // position, used << 16 | cnt, piece of the fix
    __shared__ uint32_t e_ws[APW_EDITS + 1];  // words of the batch's fixes in front of fix m
    __shared__ uint32_t e_os[APW_EDITS + 1];  // output offset (inside the batch) of the copied stretch in front of fix m
    __shared__ uint32_t e_in[APW_EDITS];      // input position where that stretch starts
    __shared__ uint64_t pc_src[AP_PIECES];   // pieces: where their lists are (list index << 60 | entry of the u32 geometry)
    __shared__ uint32_t pc_off[AP_PIECES + 1];
    __shared__ uint32_t sh_part[4], sh_partw[4];
    __shared__ uint32_t sh_np, sh_next_u, sh_state;
    __shared__ uint4 sh_rr[512]; // the records of the read's first 256 units: [t] and [256 + t]
    __shared__ uint64_t sh_eat[256];
    uint4 *sh_ra = sh_rr, *sh_rb = sh_rr + 256;
    const int k_ = p.k;
    const uint64_t kmask_ = kmask(k_);
    for (uint32_t r = blockIdx.x; r < p.n_reads; r += gridDim.x) {
        if (p.in_staged && p.in_lens[r] == 0xffffffffu) {
            if (threadIdx.x == 0)
                p.out_lens[r] = 0xffffffffu;
            continue;
        }
        const uint8_t *in;
        uint32_t n;
        bool poisoned;
        const uint64_t in_at = read_view(p, r, in, n, poisoned);
        const uint64_t o0 = p.offsets[r], o1 = p.offsets[r + 1];
        const uint64_t s0 = slot_of(o0, r, p.slack), s1 = slot_of(o1, (uint64_t)r + 1, p.slack);
        uint8_t *dst = p.out + s0;
        const uint64_t slot = s1 - s0;
        const uint32_t ub = (uint32_t)a.ubase[r];
        uint64_t total = 0;
        uint32_t cur_in = 0;
        bool failed = false, more = true;
        uint32_t u_next = ub, d_next = 0;
        uint32_t committed = 0; // fixes replayed so far (uniform)
        __syncthreads();
        if (ub + threadIdx.x < (uint32_t)a.ubase[r + 1]) {
            const uint32_t uu = ub + threadIdx.x;
            sh_ra[threadIdx.x] = *reinterpret_cast<const uint4 *>(a.u_res + 8ull * uu);
            sh_rb[threadIdx.x] = *reinterpret_cast<const uint4 *>(a.u_res + 8ull * uu + 4);
            const uint4 d1 = reinterpret_cast<const uint4 *>(a.u_desc + uu)[1];
            sh_eat[threadIdx.x] = ((uint64_t)d1.w << 32) | d1.z;
        }
        while (more && !failed) {
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t np = 0, cnt = 0, u = u_next, d = d_next, state = 1;
                pc_off[0] = 0;
                for (;;) {
                    const bool near = u - ub < 256u;
                    const uint4 ra = near ? sh_ra[u - ub] : *reinterpret_cast<const uint4 *>(a.u_res + 8ull * u);
                    const uint4 rb = near ? sh_rb[u - ub] : *reinterpret_cast<const uint4 *>(a.u_res + 8ull * u + 4);
                    const uint32_t res[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
                    const uint32_t code = res[6];
                    if (code >= (uint32_t)MAX_DEPTH) {
                        if (code == U_UNWRITTEN)
                            atomicAdd(p.ctrl + CTL_LANE_UNWRITTEN, 1ull);
                        state = 2;
                        break;
                    }
                    bool full = false;
                    for (; d <= code; d++) {
                        const uint32_t ne = res[2 * d];
                        if (ne > APW_EDITS) { // (a piece is never split over two batches here: hand the read back)
                            state = 2;
                            full = true;
                            break;
                        }
                        if (np == AP_PIECES || cnt + ne > APW_EDITS) {
                            full = true;
                            break;
                        }
                        const uint32_t xu = d == 0 ? u : res[2 * d - 1];
                        pc_src[np] = ((uint64_t)d << 60) | (xu - ub < 256u ? sh_eat[xu - ub] : edit_start(in_at, r, ub, xu - ub, a.u_q[xu]));
                        cnt += ne;
                        np++;
                        pc_off[np] = cnt;
                    }
                    if (full)
                        break;
                    const uint32_t next = res[2 * code + 1];
                    if (next == U_END) {
                        state = 0;
                        break;
                    }
                    u = next;
                    d = 0;
                }
                sh_np = np;
                sh_next_u = u;
                sh_state = state | (d << 8);
            }
            __syncthreads();
            const uint32_t np = sh_np;
            u_next = sh_next_u;
            d_next = sh_state >> 8;
            failed = (sh_state & 0xffu) == 2u;
            more = (sh_state & 0xffu) == 1u;
            if (failed)
                break;
            const uint32_t nb = pc_off[np];
            if (nb == 0u)
                continue;
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < nb; t += 256) {
                uint32_t lo = 0, hi = np;
                while (hi - lo > 1u) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (pc_off[mid] <= t)
                        lo = mid;
                    else
                        hi = mid;
                }
                const uint64_t src = pc_src[lo];
                const uint2 e = a.EW[src >> 60][((src & 0x0fffffffffffffffull) >> 1) + (t - pc_off[lo])];
                e_pos[t] = e.x; // (bit 31: a fixed-length walk, to be checked for a repeated k-mer below)
                e_uc[t] = e.y;
                e_pi[t] = lo;
            }
            __syncthreads();
            const uint32_t per = (nb + 255u) / 256u;
            const uint32_t t0 = threadIdx.x * per, t1 = t0 + per < nb ? t0 + per : nb;
            uint32_t part = 0, partw = 0;
            for (uint32_t t = t0; t < t1; t++) {
                const uint32_t prev_end = t == 0 ? cur_in : (e_pos[t - 1] & 0x7fffffffu) + (e_uc[t - 1] >> 16);
                e_in[t] = prev_end;
                part += ((e_pos[t] & 0x7fffffffu) - prev_end) + (e_uc[t] & 0xffffu);
                partw += ((e_uc[t] & 0xffffu) + 15u) / 16u + ((e_pos[t] >> 31) ? 2u : 0u);
            }
            uint32_t inc = part, incw = partw;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(inc, o), vw = __shfl_up(incw, o);
                if ((int)(threadIdx.x & 63u) >= o) {
                    inc += v;
                    incw += vw;
                }
            }
            if ((threadIdx.x & 63u) == 63u) {
                sh_part[threadIdx.x >> 6] = inc;
                sh_partw[threadIdx.x >> 6] = incw;
            }
            __syncthreads();
            uint32_t run = inc - part, runw = incw - partw;
            for (uint32_t q = 0; q < (threadIdx.x >> 6); q++) {
                run += sh_part[q];
                runw += sh_partw[q];
            }
            const uint32_t batch_total = sh_part[0] + sh_part[1] + sh_part[2] + sh_part[3];
            for (uint32_t t = t0; t < t1; t++) {
                e_os[t] = run;
                e_ws[t] = runw;
                run += ((e_pos[t] & 0x7fffffffu) - e_in[t]) + (e_uc[t] & 0xffffu);
                runw += ((e_uc[t] & 0xffffu) + 15u) / 16u + ((e_pos[t] >> 31) ? 2u : 0u);
            }
            if (threadIdx.x == 0)
                e_os[nb] = batch_total;
            __syncthreads();
            // GapSize's fixed-length walks (gap_size.rs:57-85) were not checked for revisits by the lanes: the reference returns
            // None the moment a k-mer of the walk repeats (viewed_kmer, :75-81).  A fix whose gap + 1 k-mers -- its first one
            // and one more per walked base -- hold a repeat would not have been made: the read goes back to the group kernel.
            // The walk is deterministic -- every k-mer is the unique solid successor of the one before --, so two equal k-mers
            // x_a = x_b drag everything behind them along: x_{a+t} = x_{b+t}, and the LAST k-mer of the path equals an earlier
            // one.  A repeat anywhere therefore shows as "the last k-mer was seen before": one thread per fix rolls the path
            // twice, once to its last k-mer and once comparing (the first form compared every pair behind a 128-bit
            // filter that long paths saturate, and needed a cooperative path for walks over 48 bases).
            {
                bool bad = false;
                for (uint32_t m = threadIdx.x; m < nb; m += 256) {
                    if (!(e_pos[m] >> 31))
                        continue;
                    const uint32_t cnt = e_uc[m] & 0xffffu;
                    if (cnt > AP_VERIFY) {
                        bad = true;
                        continue;
                    }
                    const uint64_t src = pc_src[e_pi[m]];
                    const uint32_t *W = a.BW[src >> 60] + (src & 0x0fffffffffffffffull) + (e_ws[m] - e_ws[pc_off[e_pi[m]]]);
                    const uint64_t first_k = ((uint64_t)W[1] << 32) | W[0]; // the k-mer after 0 walked bases (base 0 of the fix is its last)
                    uint64_t kl = first_k;
                    uint32_t wcur = W[2];
                    for (uint32_t j = 1; j < cnt; j++) {
                        if ((j & 15u) == 0u)
                            wcur = W[2u + (j >> 4)];
                        kl = ((kl << 2) | (uint64_t)((wcur >> (30u - 2u * (j & 15u))) & 3u)) & kmask_;
                    }
                    uint64_t kt = first_k;
                    wcur = W[2];
                    for (uint32_t j = 1; j < cnt; j++) { // kt = the k-mer after j - 1 walked bases
                        bad |= kt == kl;
                        if ((j & 15u) == 0u)
                            wcur = W[2u + (j >> 4)];
                        kt = ((kt << 2) | (uint64_t)((wcur >> (30u - 2u * (j & 15u))) & 3u)) & kmask_;
                    }
                }
                if (__syncthreads_or(bad ? 1 : 0))
                    failed = true;
            }
            if (failed)
                break;
            if (total + batch_total <= slot) {
                uint8_t *ob = dst + total;
                const uint32_t head = (uint32_t)((16u - (uint32_t)((uintptr_t)ob & 15u)) & 15u);
                const uint32_t n_chunks = (batch_total + 31u) / 16u + 1u;
                for (uint32_t ch = threadIdx.x; ch < n_chunks; ch += 256) {
                    const uint32_t x0 = ch == 0 ? 0u : head + 16u * (ch - 1u);
                    uint32_t x1 = ch == 0 ? head : x0 + 16u;
                    if (x1 > batch_total)
                        x1 = batch_total;
                    if (x0 >= x1)
                        continue;
                    uint32_t lo = 0, hi = nb;
                    while (hi - lo > 1u) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (e_os[mid] <= x0)
                            lo = mid;
                        else
                            hi = mid;
                    }
                    uint32_t m = lo;
                    const uint32_t copied_end = e_os[m + 1] - (e_uc[m] & 0xffffu); // the fix's own bases start here
                    if (x1 - x0 == 16u && x1 <= copied_end) {
                        *reinterpret_cast<uint4 *>(ob + x0) = ld16_logical(in, n, e_in[m] + (x0 - e_os[m]), p.flip);
                    } else {
                        // (the word of written bases in hand is kept from byte to byte: a fix of a walking corrector writes
                        // several bases, and a load per byte was most of what this kernel cost beyond One's)
                        uint32_t w = 0, w_m = 0xffffffffu, w_i = 0;
                        for (uint32_t x = x0; x < x1; x++) {
                            while (x >= e_os[m + 1])
                                m++;
                            const uint32_t rel = x - e_os[m], seglen = (e_pos[m] & 0x7fffffffu) - e_in[m];
                            if (rel < seglen) {
                                ob[x] = ld_logical(in, n, e_in[m] + rel, p.flip);
                            } else {
                                const uint32_t tb = rel - seglen; // base tb of the fix
                                if (w_m != m || w_i != (tb >> 4)) {
                                    const uint64_t src = pc_src[e_pi[m]];
                                    const uint32_t first_fix = pc_off[e_pi[m]];
                                    w_m = m;
                                    w_i = tb >> 4;
                                    w = a.BW[src >> 60][(src & 0x0fffffffffffffffull) + (e_ws[m] - e_ws[first_fix]) + ((e_pos[m] >> 31) ? 2u : 0u) + w_i];
                                }
                                ob[x] = bit2nuc((w >> (30u - 2u * (tb & 15u))) & 3u);
                            }
                        }
                    }
                }
            }
            total += batch_total;
            committed += nb;
            cur_in = (e_pos[nb - 1] & 0x7fffffffu) + (e_uc[nb - 1] >> 16);
        }
        if (!failed && n > cur_in) {
            const uint32_t len = n - cur_in;
            if (total + len <= slot)
                copy_bytes(dst + total, in, n, cur_in, len, p.flip);
            total += len;
        }
        if (threadIdx.x == 0) {
            if (failed) {
                const unsigned long long at = atomicAdd(p.ctrl + CTL_LANE_FAIL, 1ull);
                a.fail_list[at] = r;
            } else if (total + 1u > slot) {
                p.out_lens[r] = 0xffffffffu;
                atomicAdd(p.ctrl + CTL_OVERFLOW, 1ull);
            } else {
                p.out_lens[r] = (uint32_t)total;
                if (committed)
                    atomicAdd(p.ctrl + CTL_FIXES, (unsigned long long)committed);
            }
        }
    }
}

struct LaneWork {
    uint32_t *nu = nullptr, *u_read = nullptr, *u_q = nullptr, *u_res = nullptr, *fail_list = nullptr, *P = nullptr;
    uint16_t *M = nullptr;
    UnitDesc *u_desc = nullptr;
    uint4 *u_in = nullptr;
    uint32_t *E[MAX_DEPTH] = {nullptr, nullptr, nullptr};
    uint32_t *BW[MAX_DEPTH] = {nullptr, nullptr, nullptr};
    uint64_t *ubase = nullptr, *u_qk = nullptr;
    uint64_t reads_cap = 0, units_cap = 0, p_cap = 0, e_cap = 0, bw_cap = 0;
};

int grow_dev(void **ptr, uint64_t bytes)
{
    if (*ptr)
        (void)hipFree(*ptr);
    *ptr = nullptr;
    hipError_t e = hipMalloc(ptr, bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%llu B, lane pass workspace): %s", (unsigned long long)bytes, hipGetErrorString(e));
        return BRX_ERR_NOMEM;
    }
    return BRX_OK;
}

uint32_t env_u32(const char *name, uint32_t dflt)
{
    const char *e = getenv(name); // (read per call: the fuzzers sweep these)
    return e && *e ? (uint32_t)strtoul(e, nullptr, 10) : dflt;
}

template <bool IDX, int M>
void launch_lane_walk(const LaneArgs &a, uint32_t blocks, hipStream_t s)
{
    if (a.p.k == 19)
        lane_walk_kernel<IDX, 19, M><<<blocks, 256, 0, s>>>(a);
    else if (a.p.k == 21)
        lane_walk_kernel<IDX, 21, M><<<blocks, 256, 0, s>>>(a);
    else
        lane_walk_kernel<IDX, 0, M><<<blocks, 256, 0, s>>>(a);
}

template <bool IDX, bool MASK>
void launch_lane(const LaneArgs &a, uint32_t blocks, hipStream_t s)
{
    if (a.p.k == 19)
        lane_kernel<IDX, 19, MASK><<<blocks, 256, 0, s>>>(a);
    else if (a.p.k == 21)
        lane_kernel<IDX, 21, MASK><<<blocks, 256, 0, s>>>(a);
    else
        lane_kernel<IDX, 0, MASK><<<blocks, 256, 0, s>>>(a);
}

} // namespace

namespace brx {

void lane_ws_free(brx_chain *ch)
{
    LaneWork *w = (LaneWork *)ch->lane_ws;
    if (!w)
        return;
    for (void *q : {(void *)w->u_in, (void *)w->u_desc, (void *)w->nu, (void *)w->u_read, (void *)w->u_q, (void *)w->u_res, (void *)w->fail_list, (void *)w->P, (void *)w->M,
                    (void *)w->E[0], (void *)w->E[1], (void *)w->E[2], (void *)w->BW[0], (void *)w->BW[1], (void *)w->BW[2], (void *)w->ubase,
                    (void *)w->u_qk})
        if (q)
            (void)hipFree(q);
    delete w;
    ch->lane_ws = nullptr;
}

int lane_pass(brx_chain *ch, const PassParams &p, const LanePassInfo &info, hipStream_t s)
{
    // BRX_LANE=0: the group kernel only.  The window of a round shows 8 bases: look-aheads up to off + c + 1 <= 8.
    const bool walk = info.method != BRX_ONE;
    if (env_u32("BRX_LANE", 1u) == 0u || p.k > 31 || p.n_reads == 0)
        return BRX_ERR_UNSUPPORTED;
    if (info.method != BRX_GRAPH && (p.c < 1 || p.c > 5)) // (Graph has no look-aheads; One and GapSize's One branch do)
        return BRX_ERR_UNSUPPORTED;
    if (walk && env_u32("BRX_LANE_WALK", 1u) == 0u)
        return BRX_ERR_UNSUPPORTED;
    // One against a LARGE index stays with the group kernel: every lane of a wave probes a line of a different read, 64
    // unrelated pages per load, and past 2^26 lines (4 GiB) that costs more than the lanes save -- same box, correction of
    // 2 / 3.5 / 6.25 Gbp against sets of 41 / 73 / 130 M k-mers (2^26 / 2^27 / 2^28 lines): lanes 56.6 / 132.5 / 256 ms,
    // groups 61.0 / 107.9 / 197 ms.  (The walking methods gain at 2^28 lines all the same: their group kernels are
    // latency-bound, BASELINE configs[4]'s share 5.1 -> 6.9 Gbases/s.)
    if (!walk && p.idx.lines && 32u - p.idx.line_shift > env_u32("BRX_LANE_MAX_LOG_LINES", 26u))
        return BRX_ERR_UNSUPPORTED;
    const bool idx = p.idx.lines != nullptr;
    if (!idx && !p.bits)
        return BRX_ERR_UNSUPPORTED;
    // chunk length: enough units to keep ~4.6e5 lanes busy a few times over, not so short that the stretch a unit's
    // predecessor re-scans in front of its sync point (a few dozen bases) becomes the larger part
    constexpr uint64_t RESIDENT = 256ull * 4ull * BRX_LANE_WAVES * 64ull;
    uint32_t C = env_u32("BRX_LANE_CHUNK", 0u);
    if (C == 0u) {
        // (measured at configs[1], same box: 256 -> 11.7 ms per launch, 512 -> 11.65, 768 -> 12.4, 1090 -> 12.6, 2048 -> 16.7:
        // a lane should get about four units, or the lanes that got one more than the others run the tail alone)
        const uint64_t want = info.in_total_bound / (4ull * RESIDENT);
        C = want < 256ull ? 256u : (want > 2048ull ? 2048u : (uint32_t)want);
    }
    if (C < 64u)
        C = 64u;
    uint32_t R = env_u32("BRX_LANE_SYNC", 4u);
    if (R < 1u)
        R = 1u;
    if (R > 32u)
        R = 32u;

    LaneWork *w = (LaneWork *)ch->lane_ws;
    if (!w) {
        w = new LaneWork();
        ch->lane_ws = w;
    }
    // BRX_LANE_TAIL=1: the last fifth of the batch's reads is cut at C / 2, the last twentieth at C / 4 (chunk_of).  OFF by
    // default: measured at configs[1] (profiles/r4i_lane_tail_ab.txt) the pass got 0.9 ms SLOWER (2.35 M units instead of
    // 1.8 M: every unit start stalls its wave, and the sync / replay kernels grow with the units), as did guided draws
    // (no change) -- the thinning tail of the launch is not where its time goes.
    // (One only: a walking corrector's fix has to fit its unit's list, and short units hand more reads back)
    const bool graded = !walk && env_u32("BRX_LANE_TAIL", 0u) != 0u && p.n_reads >= 64u;
    const uint32_t r_half = graded ? p.n_reads - p.n_reads / 5u : 0xffffffffu;
    const uint32_t r_quarter = graded ? p.n_reads - p.n_reads / 20u : 0xffffffffu;
    const uint32_t c_min = graded ? ((C >> 2) < 64u ? 64u : (C >> 2)) : C;
    const uint64_t units_bound = info.in_total_bound / c_min + (uint64_t)p.n_reads + 1ull;
    if (units_bound >= 0xfffffff0ull)
        return BRX_ERR_UNSUPPORTED;
    const uint64_t p_bound = (info.in_total_bound >> 4) + 5ull * p.n_reads + 16ull;                        // dwords
    const uint64_t e_bound = (info.in_total_bound >> 2) + 16ull * (units_bound + 2ull * p.n_reads) + 64ull; // entries
    if (w->reads_cap < p.n_reads) {
        w->reads_cap = 0; // (a growth that fails half-way leaves freed pointers: nothing here may be trusted then)
        const uint64_t cap = (uint64_t)p.n_reads + p.n_reads / 8 + 64;
        BRX_TRY(grow_dev((void **)&w->nu, cap * 4));
        BRX_TRY(grow_dev((void **)&w->fail_list, cap * 4));
        BRX_TRY(grow_dev((void **)&w->ubase, (cap + 1) * 8));
        w->reads_cap = cap;
    }
    if (w->units_cap < units_bound) {
        w->units_cap = 0; // (a growth that fails half-way leaves freed pointers: nothing here may be trusted then)
        const uint64_t cap = units_bound + units_bound / 8 + 64;
        BRX_TRY(grow_dev((void **)&w->u_read, cap * 4));
        BRX_TRY(grow_dev((void **)&w->u_q, cap * 4));
        BRX_TRY(grow_dev((void **)&w->u_qk, cap * 8));
        BRX_TRY(grow_dev((void **)&w->u_res, cap * 32));
        BRX_TRY(grow_dev((void **)&w->u_desc, cap * sizeof(UnitDesc)));
        BRX_TRY(grow_dev((void **)&w->u_in, cap * 16));
        w->units_cap = cap;
    }
    if (w->p_cap < p_bound) {
        w->p_cap = 0; // (a growth that fails half-way leaves freed pointers: nothing here may be trusted then)
        const uint64_t cap = p_bound + p_bound / 16;
        BRX_TRY(grow_dev((void **)&w->P, cap * 4));
        BRX_TRY(grow_dev((void **)&w->M, cap * 2));
        w->p_cap = cap;
    }
    if (w->e_cap < e_bound) {
        w->e_cap = 0; // (a growth that fails half-way leaves freed pointers: nothing here may be trusted then)
        const uint64_t cap = e_bound + e_bound / 16;
        for (int d = 0; d < MAX_DEPTH; d++)
            BRX_TRY(grow_dev((void **)&w->E[d], cap * 4));
        w->e_cap = cap;
    }
    if (walk && w->bw_cap < e_bound) {
        w->bw_cap = 0;
        const uint64_t cap = e_bound + e_bound / 16;
        for (int d = 0; d < MAX_DEPTH; d++)
            BRX_TRY(grow_dev((void **)&w->BW[d], cap * 4));
        w->bw_cap = cap;
    }
    if (scan_tmp_bytes(p.n_reads) > ch->scan_tmp_cap) {
        set_error("lane pass: scan scratch smaller than the batch");
        return BRX_ERR_ARG;
    }
    LaneArgs a;
    a.p = p;
    a.C = C;
    a.r_half = r_half;
    a.r_quarter = r_quarter;
    a.R = R;
    a.nu = w->nu;
    a.ubase = w->ubase;
    a.u_read = w->u_read;
    a.u_q = w->u_q;
    a.u_qk = w->u_qk;
    a.u_res = w->u_res;
    a.u_desc = w->u_desc;
    a.u_in = w->u_in;
    a.P = w->P;
    // BRX_LANE_MASK: the solidity mask of the original k-mers (lane_mask_kernel) -- 0: never, every SCAN / error_len
    // position is probed; 1: for the walking correctors, whose error_len it shortens to a bit scan (rounds 1.97 G -> 0.96 G
    // per Gbp for Graph; both launches 65.0 -> 52.6 ms with 10.2 ms of mask kernel to pay for it; BASELINE configs[4]'s
    // share, 2^28 index lines: 6.58 -> 7.13 Gbases/s on one box); 2: for One as well (there the mask kernel costs what the
    // shorter scan saves: rounds 1.24 G -> 0.59 G, launch 16 -> 12 ms, mask 8.7 ms).  Either way the index lines fetched
    // are the same ones: the mask kernel fetches them for 64 neighbouring positions at a time instead of lane by lane.
    const uint32_t mask_mode = env_u32("BRX_LANE_MASK", 1u);
    const bool use_mask = walk ? mask_mode >= 1u : mask_mode >= 2u;
    a.M = use_mask ? w->M : nullptr;
    for (int d = 0; d < MAX_DEPTH; d++) {
        a.E[d] = w->E[d];
        a.EW[d] = reinterpret_cast<uint2 *>(w->E[d]); // (8-byte entries at half the index: the same bytes)
        a.BW[d] = w->BW[d];
    }
    if (info.method == BRX_GRAPH)
        a.p.c = 1; // (unused by Graph; keeps the One branch's shifts in range)
    a.fail_list = w->fail_list;
    a.succ = nullptr;
    a.dbg = nullptr;
    unsigned long long *d_dbg = nullptr;
#ifdef BRX_LANE_TIMING
    if (!walk && env_u32("BRX_LANE_TIMING", 0u) != 0u) { // (debug: when do the waves of the automaton start and end?)
        BRX_HIP(hipMalloc((void **)&d_dbg, (1 + 3 * 16384) * 8));
        BRX_HIP(hipMemsetAsync(d_dbg, 0, (1 + 3 * 16384) * 8, s));
        a.dbg = d_dbg;
    }
#endif
    if (walk && idx && env_u32("BRX_LANE_SUCC", 1u) != 0u) {
        // the successor table of the set's index: built the first time a walking corrector runs on it, again when the
        // index has changed since
        brx_set *set = const_cast<brx_set *>(ch->set);
        std::lock_guard<std::mutex> g(set->idx_mu);
        const uint64_t n_lines = 1ull << (32u - p.idx.line_shift);
        if (!set->d_succ || set->succ_lines != n_lines || set->succ_gen != set->idx_gen) {
            if (set->d_succ && set->succ_lines != n_lines) {
                (void)hipFree(set->d_succ);
                set->d_succ = nullptr;
            }
            if (!set->d_succ) {
                BRX_TRY(grow_dev((void **)&set->d_succ, n_lines * 8ull));
                set->succ_lines = n_lines;
            }
            KernelTimer t("succ_build", s);
            const uint64_t blocks = (n_lines + 255ull) / 256ull;
            succ_build_kernel<<<(uint32_t)(blocks < 65536ull ? blocks : 65536ull), 256, 0, s>>>(p, n_lines, set->d_succ);
            BRX_HIP(hipGetLastError());
            // the table belongs to the SET: another chain on it (its own stream, another host thread) takes the generation
            // below as "built" the moment the lock is released, so the kernel has to be through by then (the index build
            // does the same, brx_index.hip)
            BRX_HIP(hipStreamSynchronize(s));
            set->succ_gen = set->idx_gen;
        }
        a.succ = set->d_succ;
    }

    const uint32_t rb = (p.n_reads + 255u) / 256u;
    {
        KernelTimer t("lane_units", s);
        BRX_HIP(hipMemsetAsync(p.ctrl + CTL_LANE_UNITS, 0, (CTL_LANE_FAIL + 1 - CTL_LANE_UNITS) * 8, s));
        // every unit's record starts as "unwritten" (all ones): lane_apply hands a read with such a record to the group
        // kernel and counts it, instead of replaying whatever the memory held (32 B per unit: microseconds)
        BRX_HIP(hipMemsetAsync(w->u_res, 0xff, units_bound * 32ull, s));
        lane_units_kernel<<<rb, 256, 0, s>>>(a);
        BRX_TRY(exclusive_scan_lens(w->nu, p.n_reads, ch->d_scan_tmp, w->ubase, p.ctrl + CTL_LANE_UNITS, s));
        const uint32_t grid = p.n_reads < (1u << 16) ? p.n_reads : (1u << 16);
        lane_pack_kernel<<<grid, 256, 0, s>>>(a);
    }
    if (use_mask) {
        KernelTimer t("lane_mask", s);
        BRX_HIP(hipMemsetAsync(w->M, 0, p_bound * 2ull, s)); // (the padding behind every read: no solid k-mers there)
        const uint64_t waves = units_bound < 256ull * 32ull ? units_bound : 256ull * 32ull;
        const uint32_t blocks = (uint32_t)((waves + 3) / 4);
        if (idx)
            lane_mask_kernel<true><<<blocks, 256, 0, s>>>(a);
        else
            lane_mask_kernel<false><<<blocks, 256, 0, s>>>(a);
    }
    {
        KernelTimer t("lane_sync", s);
        const uint64_t waves = units_bound < 256ull * 32ull ? units_bound : 256ull * 32ull;
        const uint32_t blocks = (uint32_t)((waves + 3) / 4);
        if (use_mask) {
            const uint64_t ub = (units_bound + 255ull) / 256ull;
            lane_sync_mask_kernel<<<(uint32_t)(ub < 16384ull ? ub : 16384ull), 256, 0, s>>>(a);
        } else if (idx)
            lane_sync_kernel<true><<<blocks, 256, 0, s>>>(a);
        else
            lane_sync_kernel<false><<<blocks, 256, 0, s>>>(a);
        const uint64_t lb = (units_bound + 255ull) / 256ull;
        lane_link_kernel<<<(uint32_t)(lb < 4096ull ? lb : 4096ull), 256, 0, s>>>(a);
    }
    {
        static const char *names[5] = {"correct_pass", "correct_pass_two", "correct_pass_graph", "correct_pass_greedy", "correct_pass_gap_size"};
        KernelTimer t(names[info.method], s);
        const uint64_t want = (units_bound + 255ull) / 256ull;
        const uint64_t waves = walk ? (uint64_t)walk_waves(info.method) : (use_mask ? 6ull : (uint64_t)BRX_LANE_WAVES);
        const uint32_t blocks = (uint32_t)(want < 256ull * waves ? want : 256ull * waves);
        if (info.method == BRX_ONE) {
            if (idx && use_mask)
                launch_lane<true, true>(a, blocks, s);
            else if (idx)
                launch_lane<true, false>(a, blocks, s);
            else if (use_mask)
                launch_lane<false, true>(a, blocks, s);
            else
                launch_lane<false, false>(a, blocks, s);
        } else if (info.method == BRX_GRAPH) {
            if (idx)
                launch_lane_walk<true, BRX_GRAPH>(a, blocks, s);
            else
                launch_lane_walk<false, BRX_GRAPH>(a, blocks, s);
        } else {
            if (idx)
                launch_lane_walk<true, BRX_GAP_SIZE>(a, blocks, s);
            else
                launch_lane_walk<false, BRX_GAP_SIZE>(a, blocks, s);
        }
    }
    if (d_dbg) {
        std::vector<unsigned long long> h(1 + 3 * 16384);
        BRX_HIP(hipMemcpyAsync(h.data(), d_dbg, h.size() * 8, hipMemcpyDeviceToHost, s));
        BRX_HIP(hipStreamSynchronize(s));
        (void)hipFree(d_dbg);
        const size_t nw = (size_t)(h[0] < 16384ull ? h[0] : 16384ull);
        if (nw) {
            std::vector<double> st(nw), en(nw), it(nw), lr(nw);
            unsigned long long t0 = ~0ull;
            for (size_t i = 0; i < nw; i++)
                t0 = h[1 + 3 * i] < t0 ? h[1 + 3 * i] : t0;
            for (size_t i = 0; i < nw; i++) {
                st[i] = (double)(h[1 + 3 * i] - t0) * 0.01; // microseconds
                en[i] = (double)(h[2 + 3 * i] - t0) * 0.01;
                it[i] = (double)(h[3 + 3 * i] & 0xffffffffull);
                lr[i] = (double)(h[3 + 3 * i] >> 32);
            }
            auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (double)(v.size() - 1))]; };
            fprintf(stderr, "[lane timing] %zu waves, %u reads; start us p0/p50/p100 %.0f/%.0f/%.0f; end us p0/p10/p50/p90/p99/p100 %.0f/%.0f/%.0f/%.0f/%.0f/%.0f; "
                            "loop iterations per wave p0/p50/p100 %.0f/%.0f/%.0f; lane-rounds per wave p50/p100 %.0f/%.0f\n",
                    nw, p.n_reads, pct(st, 0), pct(st, .5), pct(st, 1), pct(en, 0), pct(en, .1), pct(en, .5), pct(en, .9), pct(en, .99), pct(en, 1),
                    pct(it, 0), pct(it, .5), pct(it, 1), pct(lr, .5), pct(lr, 1));
        }
    }
    {
        KernelTimer t("lane_apply", s);
        // (BRX_AP_GRID: blocks of the replay kernels; each block loops over reads)
        const uint32_t gcap = env_u32("BRX_AP_GRID", 1u << 20);
        const uint32_t grid = p.n_reads < gcap ? p.n_reads : gcap;
        if (walk)
            lane_apply_walk_kernel<<<grid, 256, 0, s>>>(a);
        else
            lane_apply_kernel<<<grid, AP_BS, 0, s>>>(a);
    }
    {
        // the reads the units could not settle (three misses in a row, more fixes than a list holds): the group kernel
        KernelTimer t("lane_redo", s);
        PassParams q = p;
        q.only = w->fail_list;
        q.only_n = p.ctrl + CTL_LANE_FAIL;
        BRX_HIP(hipMemsetAsync(p.ctrl + CTL_WORK, 0, 8, s));
        if (walk)
            BRX_TRY(launch_walk_list(q, info.method, s));
        else
            BRX_TRY(launch_one_list(q, s));
    }
    BRX_HIP(hipGetLastError());
    return BRX_OK;
}

} // namespace brx
