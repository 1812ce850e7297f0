// correct::One, forward pass, ONE LANE PER CHUNK OF A READ.
//
// Reference: Corrector::correct (src/correct/mod.rs:53-107), Exist<ScenarioOne>::correct_error
// (src/correct/exist/mod.rs:112-150, get_score :21-47, one_more :49-70), ScenarioOne (src/correct/exist/one.rs:33-74).
//
// Why.  The group kernel (brx_correct.hip: one_kernel, 8 lanes per read) is instruction-issue-bound: ~440 vector
// instructions per wave-round with 29 of 64 lanes live, because the eight groups of a wave sit in different states and
// every wave runs the union of the states' code every round (profiles/r2h_sq_summary.json).  Here every lane is its own
// scalar state machine over its own stretch of a read, written as ONE straight-line round -- build the k-mer this state
// asks about, one KmerSet::get, one table-like transition, at most one byte out -- so that all 64 lanes of a wave do
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
// base), it keeps scanning through the next unit's stretch into a second staging area and checks again at the sync
// point after that; two misses in a row hand the read back to the group kernel.  No unit ever waits for another one:
// all coupling is resolved afterwards by lane_stitch_kernel, which walks the chain of units of a read (which unit's
// output is the truth up to where) and copies the pieces into the read's ordinary staging slot, so that everything
// downstream (reverse pass, other methods, compaction, the redo of overflowing reads) is unchanged.
#include "brx_correct.hpp"

#include <stdlib.h>

using namespace brx;

namespace brx {
uint64_t scan_tmp_bytes(uint32_t n);
int exclusive_scan_lens(const uint32_t *d_lens, uint32_t n, uint64_t *d_tmp, uint64_t *d_out_offsets,
                        unsigned long long *d_total, hipStream_t s);
}

namespace {

constexpr uint32_t U_VOID = 0xffffffffu;   // u_q: the unit found no sync point (its predecessor scans through it)
constexpr uint32_t U_END = 0xffffffffu;    // res[2] / res[3]: the target was the end of the read
constexpr uint32_t C_MATCH1 = 0xfffffff0u; // res[3]: the unit's state matched at its first target
constexpr uint32_t C_FAIL = 0xfffffff1u;   // res[3]: two misses, or a staging region overflowed: back to the group kernel
constexpr uint32_t C_VOID = 0xfffffff2u;   // res[3]: nothing produced

struct LaneArgs {
    PassParams p;
    uint32_t C;                // nominal chunk length
    uint32_t R;                // solid original k-mers in a row that make a sync point
    uint32_t *nu;              // units per read                            [n_reads]
    uint32_t *xsz;             // bytes of unit staging per read            [n_reads]
    uint64_t *ubase;           // exclusive scan of nu                      [n_reads + 1]
    uint64_t *xbase;           // exclusive scan of xsz                     [n_reads + 1]
    uint32_t *u_read;          // read of a unit                            [units]
    uint32_t *u_q;             // its sync position (0 for a read's first)  [units]
    uint64_t *u_qk;            // the original k-mer in front of it         [units]
    uint32_t *u_res;           // len_own, len_cont, t1, code               [4 x units]
    uint8_t *X, *Y;            // unit staging: own stretch / the stretch after a miss
    uint32_t *fail_list;       // reads handed back to the group kernel     [n_reads]
};

// where unit jj of a read, starting at position q, writes inside the read's share of X / Y.  Monotone in (jj, q); the
// distance between two units' starts leaves the stretch between them room to grow by slack/4 and 64 bytes per unit.
__device__ __forceinline__ uint64_t lane_region(uint64_t xb, uint32_t jj, uint32_t q, uint32_t slack)
{
    return xb + (uint64_t)q + (uint64_t)(q >> 2) * slack + 64ull * jj;
}

__device__ __forceinline__ void read_view(const PassParams &p, uint32_t r, const uint8_t *&in, uint32_t &n, bool &poisoned)
{
    const uint64_t o0 = p.offsets[r], o1 = p.offsets[r + 1];
    poisoned = false;
    if (p.in_staged) {
        in = p.in + slot_of(o0, r, p.slack);
        n = p.in_lens[r];
        if (n == 0xffffffffu) {
            poisoned = true;
            n = 0;
        }
    } else {
        in = p.in + o0;
        n = (uint32_t)(o1 - o0);
    }
}

// ---- unit tables ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lane_units_kernel(LaneArgs a)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= a.p.n_reads)
        return;
    const uint8_t *in;
    uint32_t n;
    bool poisoned;
    read_view(a.p, r, in, n, poisoned);
    const uint32_t nu = n < 2u * a.C ? 1u : n / a.C; // the last unit takes the remainder
    a.nu[r] = nu;
    a.xsz[r] = n + (n >> 2) * a.p.slack + 64u * nu + 64u;
}

__global__ __launch_bounds__(256) void lane_fill_kernel(LaneArgs a)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= a.p.n_reads)
        return;
    const uint64_t b = a.ubase[r], e = a.ubase[r + 1];
    for (uint64_t u = b; u < e; u++)
        a.u_read[u] = r;
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

// One wave per unit boundary: the first position q > j*C + k behind R solid original k-mers in a row, inside the unit's
// own nominal stretch; none -> the unit is void.  q is the loop-top position (mod.rs:68), u_qk the k-mer in front of it.
template <bool IDX>
__global__ __launch_bounds__(256) void lane_sync_kernel(LaneArgs a)
{
    const PassParams &p = a.p;
    const int lane = threadIdx.x & 63;
    const int k = p.k;
    const uint64_t mask = kmask(k);
    const unsigned long long n_units = p.ctrl[CTL_LANE_UNITS];
    const unsigned long long wave = (unsigned long long)blockIdx.x * 4ull + (threadIdx.x >> 6);
    const unsigned long long n_waves = (unsigned long long)gridDim.x * 4ull;
    for (unsigned long long u = wave; u < n_units; u += n_waves) {
        const uint32_t r = a.u_read[u];
        const uint32_t j = (uint32_t)(u - a.ubase[r]);
        if (j == 0) {
            if (lane == 0)
                a.u_q[u] = 0;
            continue;
        }
        const uint8_t *in;
        uint32_t n;
        bool poisoned;
        read_view(p, r, in, n, poisoned);
        const uint32_t s = j * a.C;
        const uint32_t lim = (s + a.C < n) ? s + a.C : n; // k-mers ending at e < lim
        uint32_t q = U_VOID;
        uint64_t qk = 0, carry = 0, prev_ball = 0;
        for (uint32_t t = 0; s + 64u * t < lim; t++) {
            const uint32_t e = s + 64u * t + (uint32_t)lane;
            const uint32_t code = e < n ? (uint32_t)nuc2bit(in[e]) : 0u;
            const uint64_t km = lane_kmer64_dpp(carry, code, lane, mask);
            const bool valid = e < lim && e + 1u >= s + (uint32_t)k; // the k-mer lies inside [s, lim)
            const bool sol = valid && set_get<IDX>(p, km, k);
            const uint64_t ball = __ballot(sol);
            uint64_t x = ball;
            for (uint32_t rr = 1; rr < a.R; rr++)
                x &= (ball << rr) | (prev_ball >> (64u - rr));
            if (x) {
                const int el = __builtin_ctzll(x);
                const uint32_t qq = s + 64u * t + (uint32_t)el + 1u;
                if (qq < n) {
                    q = qq;
                    qk = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(km >> 32), el) << 32) | (uint32_t)__shfl((int)(uint32_t)km, el);
                }
                break;
            }
            prev_ball = ball;
            carry = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(km >> 32), 63) << 32) |
                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)km, 63);
        }
        if (lane == 0) {
            a.u_q[u] = q;
            a.u_qk[u] = qk;
        }
    }
}

// ---- the automaton ----------------------------------------------------------------------------------------------------
enum { L_SCAN = 0, L_ALTS = 1, L_SCEN = 2, L_MORE = 3, L_FIRST = 4 };

template <bool IDX, int KT>
__global__ __launch_bounds__(256, 7) void lane_kernel(LaneArgs a)
{
    const PassParams &p = a.p;
    const int k = KT ? KT : p.k;
    const uint32_t c = (uint32_t)p.c;
    const uint64_t mask = kmask(k);
    const unsigned long long n_units = p.ctrl[CTL_LANE_UNITS];

    // the unit
    bool have = false;
    uint32_t u = 0, r = 0, n = 0, i = 0, tgt = 0, t1 = U_END, t1_first = U_END, uend = 0, nu_r = 0;
    uint64_t ub = 0, xb = 0, tgtk = 0;
    const uint8_t *in = nullptr;
    uint8_t *out = nullptr;
    uint32_t olen = 0, cap = 0, len_own = 0, phase = 0;
    // the scan (mod.rs:60-67) and the trigger in progress
    uint64_t kmer = 0, corr = 0;
    bool prev = false;
    uint32_t st = L_SCAN, cur = 0 /* alternative / scenario being asked about */, jj = 0, am = 0, passm = 0, failm = 0, keep = 0, skip = 0;
    uint32_t hop = 0;
    bool slowbits = false;
    uint32_t n_rounds = 0, n_probes = 0, n_trig = 0, n_fix = 0;

    auto set_target = [&](uint32_t from) {
        // the next unit of this read that has a sync point; none: the end of the read
        uint32_t t = from;
        while (t < uend && a.u_q[t] == U_VOID)
            t++;
        uint64_t end_at;
        if (t < uend) {
            t1 = t;
            tgt = a.u_q[t];
            tgtk = a.u_qk[t];
            end_at = lane_region(xb, t - (uint32_t)ub, tgt, p.slack);
        } else {
            t1 = U_END;
            tgt = n;
            tgtk = 0;
            end_at = lane_region(xb, nu_r, n, p.slack);
        }
        return end_at;
    };
    auto record = [&](uint32_t l0, uint32_t l1, uint32_t tt, uint32_t code) {
        uint4 v = make_uint4(l0, l1, tt, code);
        *reinterpret_cast<uint4 *>(a.u_res + 4ull * u) = v;
    };
    auto fetch = [&]() {
        for (;;) {
            const unsigned long long w = atomicAdd(p.ctrl + CTL_LANE_WORK, 1ull);
            if (w >= n_units) {
                have = false;
                return;
            }
            have = true;
            u = (uint32_t)w;
            r = a.u_read[u];
            ub = a.ubase[r];
            uend = (uint32_t)a.ubase[r + 1];
            nu_r = uend - (uint32_t)ub;
            xb = a.xbase[r];
            const uint32_t j = u - (uint32_t)ub;
            bool poisoned;
            read_view(p, r, in, n, poisoned);
            const uint32_t q = a.u_q[u];
            if (q == U_VOID) {
                record(0, 0, U_END, C_VOID);
                continue;
            }
            const uint64_t start = lane_region(xb, j, q, p.slack);
            out = a.X + start;
            const uint64_t end_at = set_target(u + 1u);
            cap = (uint32_t)(end_at - start);
            phase = 0;
            olen = 0;
            hop = 0;
            slowbits = false;
            skip = 0;
            if (j == 0) {
                if (n < (uint32_t)k) {
                    // mod.rs:56-58: shorter than k, returned verbatim (a read's only unit: n < 2C)
                    for (uint32_t t = 0; t < n; t++)
                        out[t] = in[t];
                    record(n, 0, U_END, C_MATCH1);
                    continue;
                }
                uint64_t km = 0;
                for (int t = 0; t < k; t++) {
                    const uint8_t b = in[t];
                    km = (km << 2) | nuc2bit(b);
                    out[t] = b;
                }
                kmer = km;
                olen = (uint32_t)k;
                i = (uint32_t)k;
                st = L_FIRST; // previous = get(kmer), mod.rs:67
            } else {
                i = q;
                kmer = a.u_qk[u];
                prev = true; // R >= 1 solid k-mers end in front of q
                st = L_SCAN;
            }
            return;
        }
    };

    fetch();
    while (__any(have)) {
        // ---- a unit ends where the scan reaches its target (loop top, mod.rs:68) -------------------------------------
        if (have && st == L_SCAN && i >= tgt) {
            if (t1 == U_END || (i == tgt && kmer == tgtk)) {
                if (phase == 0)
                    record(olen, 0, t1, C_MATCH1);
                else
                    record(len_own, olen, t1_first, t1);
                fetch();
            } else if (phase == 0) {
                // the prediction did not hold: keep scanning, through the next unit's stretch, into Y
                phase = 1;
                len_own = olen;
                t1_first = t1;
                const uint64_t start = lane_region(xb, t1 - (uint32_t)ub, tgt, p.slack);
                out = a.Y + start;
                olen = 0;
                const uint64_t end_at = set_target(t1 + 1u);
                cap = (uint32_t)(end_at - start);
            } else {
                record(len_own, olen, t1_first, C_FAIL);
                fetch();
            }
        }
        // (a lane that just missed its target takes part in this round only if it still is in front of the next one)
        const bool act = have && !(st == L_SCAN && i >= tgt);
        if (act) {
            n_rounds++;
            // ---- the next bases: in[i .. i+8), 2 bits each, first base in bits 15:14 ------------------------------------
            uint64_t w8 = 0;
            if (i + 8u <= n) {
                __builtin_memcpy(&w8, in + i, 8);
            } else {
                for (uint32_t t = 0; t < 8u && i + t < n; t++)
                    w8 |= (uint64_t)in[i + t] << (8u * t);
            }
            const uint32_t wlo = (uint32_t)w8, whi = (uint32_t)(w8 >> 32);
            const uint32_t cw = (((((wlo >> 1) & 0x03030303u) * 0x40100401u) >> 24) << 8) | ((((whi >> 1) & 0x03030303u) * 0x40100401u) >> 24);
            const uint32_t c0 = cw >> 14; // code of in[i]
            const uint32_t rem = n - i;

            // ---- the k-mer this state asks about ------------------------------------------------------------------------
            // SCAN: add(kmer, seq[i]); FIRST: kmer; ALTS: the trigger k-mer with its last base replaced (mod.rs:114-128);
            // SCEN: corr + seq[off .. off+j] (exist/mod.rs:33-41); MORE: corr + seq[off .. off+c] (exist/mod.rs:57-66)
            const uint32_t off = 2u - cur; // I:2 S:1 D:0 (one.rs:57-63); meaningful in SCEN / MORE only
            const bool on_corr = st == L_ALTS || st == L_SCEN || st == L_MORE;
            const uint32_t nb = st == L_SCAN ? 1u : (st == L_SCEN ? jj + 1u : (st == L_MORE ? c + 1u : 0u));
            const uint32_t b0 = (st == L_SCEN || st == L_MORE) ? off : 0u;
            const uint32_t wbits = (cw >> (16u - 2u * (b0 + nb))) & ((1u << (2u * nb)) - 1u);
            uint64_t pk = (((on_corr ? corr : kmer) << (2u * nb)) | (uint64_t)wbits) & mask;
            if (st == L_ALTS)
                pk = (pk & ~3ull) | (uint64_t)cur;
            // a base accepted behind a fix is known solid (it was a look-ahead of the winning scenario); a tie-break
            // that cannot read one base more is false without a probe (exist/mod.rs:54)
            const bool need = !(st == L_SCAN && skip != 0u) && !(st == L_MORE && !(rem > c + off + 1u));

            // ---- KmerSet::get -------------------------------------------------------------------------------------------
            bool sol = (st == L_SCAN) && !need, unres = false;
            if (need) {
                n_probes++;
                if (IDX) {
                    uint64_t key;
                    const uint32_t home = index_locate(p.idx, pk, k, key);
                    if (!slowbits) {
                        const int pr = index_probe_at(p.idx, key, home, hop);
                        sol = pr == 1;
                        unres = pr == 2;
                    } else { // the home line overflowed at build time and does not hold the key: the bit vector knows
                        const uint64_t h = key - 1ull;
                        sol = (p.bits[h >> 5] >> (h & 31u)) & 1u;
                    }
                } else {
                    sol = probe(p.bits, pk, k);
                }
            }
            if (unres) {
                // ask again next round: the bit vector, or (sparse sets) the next line of the chain
                if (p.bits)
                    slowbits = true;
                else
                    hop++;
            } else {
                slowbits = false;
                hop = 0;
                // ---- transition ---------------------------------------------------------------------------------------------
                bool push = false, fail = false;
                int apply = -1;
                uint8_t pb = (uint8_t)w8; // seq[i]
                uint32_t adv = 0;
                if (st == L_SCAN) {
                    if (sol || !prev) { // mod.rs:99-102
                        push = true;
                        adv = 1;
                        kmer = pk;
                        prev = sol;
                        skip = skip ? skip - 1u : 0u;
                    } else { // mod.rs:73: the first k-mer that is not solid after a solid one
                        corr = pk;
                        am = 0;
                        cur = c0 == 0u ? 1u : 0u; // the read's own base IS the trigger k-mer: known not solid
                        st = L_ALTS;
                        n_trig++;
                    }
                } else if (st == L_FIRST) {
                    prev = sol;
                    st = L_SCAN;
                } else if (st == L_ALTS) {
                    am |= (sol ? 1u : 0u) << cur;
                    uint32_t an = cur + 1u;
                    an += (an == c0) ? 1u : 0u;
                    if (an >= 4u || __popc(am) > 1) {
                        if (__popc(am) == 1) { // exist/mod.rs:121-129
                            corr = (corr & ~3ull) | (uint64_t)(__ffs(am) - 1);
                            failm = 0;
                            for (uint32_t s = 0; s < 3u; s++)
                                if ((2u - s) + c > rem) // exist/mod.rs:27-29
                                    failm |= 1u << s;
                            passm = 0;
                            if (failm == 7u) {
                                fail = true;
                            } else if (c == 0u) { // every scenario scores 0 == c
                                passm = 7u & ~failm;
                                if (__popc(passm) == 1) {
                                    apply = __ffs(passm) - 1;
                                } else {
                                    st = L_MORE;
                                    cur = (uint32_t)__ffs(passm) - 1u;
                                    keep = 0;
                                }
                            } else {
                                cur = (uint32_t)__ffs(7u & ~failm) - 1u;
                                jj = 0;
                                st = L_SCEN;
                            }
                        } else {
                            fail = true; // exist/mod.rs:123-126
                        }
                    } else {
                        cur = an;
                    }
                } else if (st == L_SCEN) {
                    bool over = false;
                    if (sol) {
                        jj++;
                        if (jj == c) { // get_score == c
                            passm |= 1u << cur;
                            over = true;
                        }
                    } else { // exist/mod.rs:38-42: the score stops below c
                        failm |= 1u << cur;
                        over = true;
                    }
                    if (over) {
                        const uint32_t rest = 7u & ~failm & ~passm & ~((2u << cur) - 1u);
                        if (rest) {
                            cur = (uint32_t)__ffs(rest) - 1u;
                            jj = 0;
                        } else if (passm == 0u) {
                            fail = true; // exist/mod.rs:132-134
                        } else if (__popc(passm) == 1) {
                            apply = __ffs(passm) - 1; // exist/mod.rs:135-137
                        } else {
                            st = L_MORE;
                            cur = (uint32_t)__ffs(passm) - 1u;
                            keep = 0;
                        }
                    }
                } else { // L_MORE, exist/mod.rs:138-147
                    keep |= (sol ? 1u : 0u) << cur;
                    const uint32_t rest = passm & ~((2u << cur) - 1u);
                    if (rest)
                        cur = (uint32_t)__ffs(rest) - 1u;
                    else if (__popc(keep) == 1)
                        apply = __ffs(keep) - 1;
                    else
                        fail = true;
                }
                if (fail) { // mod.rs:91-96: the trigger base is copied through, the k-mer keeps it
                    push = true;
                    adv = 1;
                    prev = false;
                    kmer = (corr & ~3ull) | (uint64_t)c0;
                    st = L_SCAN;
                }
                if (apply >= 0) { // mod.rs:75-89 with one.rs:65-71
                    push = true;
                    pb = bit2nuc(corr & 3ull);
                    kmer = corr;
                    prev = true;
                    adv = 2u - (uint32_t)apply;
                    // the c look-ahead k-mers of the winning scenario ARE the next c scan k-mers, all found solid: the
                    // reference's loop copies these bases with previous = true (mod.rs:99-102); no second probe
                    skip = c;
                    st = L_SCAN;
                    n_fix++;
                }
                if (push) {
                    if (olen < cap) {
                        out[olen] = pb;
                        olen++;
                    } else { // the stretch outgrew its staging region: the read goes back to the group kernel
                        record(phase ? len_own : olen, phase ? olen : 0u, phase ? t1_first : t1, C_FAIL);
                        fetch();
                        adv = 0;
                    }
                }
                i += adv;
            }
        }
    }
    // statistics: one atomic per wave per counter
    for (int o = 32; o > 0; o >>= 1) {
        n_rounds += __shfl_xor(n_rounds, o);
        n_probes += __shfl_xor(n_probes, o);
        n_trig += __shfl_xor(n_trig, o);
        n_fix += __shfl_xor(n_fix, o);
    }
    if ((threadIdx.x & 63) == 0) {
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

// ---- stitch: the chain of units of a read -> its staging slot --------------------------------------------------------
__device__ __forceinline__ void copy_bytes(uint8_t *dst, const uint8_t *src, uint32_t n)
{
    // bytes up to the first 16-byte boundary of dst, then 16 bytes per lane (unaligned load, aligned store), then the tail
    uint32_t head = (uint32_t)((16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u);
    if (head > n)
        head = n;
    const uint32_t nv = (n - head) / 16u;
    for (uint32_t j = threadIdx.x; j < head; j += blockDim.x)
        dst[j] = src[j];
    for (uint32_t v = threadIdx.x; v < nv; v += blockDim.x) {
        const uint32_t j = head + 16u * v;
        uint4 q;
        __builtin_memcpy(&q, src + j, 16);
        *reinterpret_cast<uint4 *>(dst + j) = q;
    }
    for (uint32_t j = head + 16u * nv + threadIdx.x; j < n; j += blockDim.x)
        dst[j] = src[j];
}

__global__ __launch_bounds__(256) void lane_stitch_kernel(LaneArgs a)
{
    const PassParams &p = a.p;
    for (uint32_t r = blockIdx.x; r < p.n_reads; r += gridDim.x) {
        if (p.in_staged && p.in_lens[r] == 0xffffffffu) { // given up by an earlier pass of this attempt: stays poisoned
            if (threadIdx.x == 0)
                p.out_lens[r] = 0xffffffffu;
            continue;
        }
        const uint64_t o0 = p.offsets[r], o1 = p.offsets[r + 1];
        const uint64_t s0 = slot_of(o0, r, p.slack), s1 = slot_of(o1, (uint64_t)r + 1, p.slack);
        uint8_t *dst = p.out + s0;
        const uint64_t slot = s1 - s0;
        const uint64_t ub = a.ubase[r], xb = a.xbase[r];
        uint64_t total = 0;
        bool failed = false;
        // every thread walks the chain (uniform loads), all of them copy each piece
        for (uint64_t u = ub;;) {
            const uint4 res = *reinterpret_cast<const uint4 *>(a.u_res + 4ull * u);
            const uint32_t q = a.u_q[u];
            if (res.w == C_VOID) { // cannot be on the chain
                failed = true;
                break;
            }
            const uint8_t *src = a.X + lane_region(xb, (uint32_t)(u - ub), q, p.slack);
            if (total + res.x + 1u <= slot)
                copy_bytes(dst + total, src, res.x);
            total += res.x;
            if (res.w == C_FAIL) {
                failed = true;
                break;
            }
            uint32_t next;
            if (res.w == C_MATCH1) {
                next = res.z;
            } else { // missed its first target t1 = res.z, went on into Y and matched at res.w
                const uint8_t *src2 = a.Y + lane_region(xb, res.z - (uint32_t)ub, a.u_q[res.z], p.slack);
                if (total + res.y + 1u <= slot)
                    copy_bytes(dst + total, src2, res.y);
                total += res.y;
                next = res.w;
            }
            if (next == U_END)
                break;
            u = next;
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
            }
        }
    }
}

struct LaneWork {
    uint32_t *nu = nullptr, *xsz = nullptr, *u_read = nullptr, *u_q = nullptr, *u_res = nullptr, *fail_list = nullptr;
    uint64_t *ubase = nullptr, *xbase = nullptr, *u_qk = nullptr;
    uint8_t *X = nullptr, *Y = nullptr;
    uint64_t reads_cap = 0, units_cap = 0, x_cap = 0;
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

template <bool IDX>
void launch_lane(const LaneArgs &a, uint32_t blocks, hipStream_t s)
{
    if (a.p.k == 19)
        lane_kernel<IDX, 19><<<blocks, 256, 0, s>>>(a);
    else if (a.p.k == 21)
        lane_kernel<IDX, 21><<<blocks, 256, 0, s>>>(a);
    else
        lane_kernel<IDX, 0><<<blocks, 256, 0, s>>>(a);
}

} // namespace

namespace brx {

void lane_ws_free(brx_chain *ch)
{
    LaneWork *w = (LaneWork *)ch->lane_ws;
    if (!w)
        return;
    for (void *q : {(void *)w->nu, (void *)w->xsz, (void *)w->u_read, (void *)w->u_q, (void *)w->u_res, (void *)w->fail_list,
                    (void *)w->ubase, (void *)w->xbase, (void *)w->u_qk, (void *)w->X, (void *)w->Y})
        if (q)
            (void)hipFree(q);
    delete w;
    ch->lane_ws = nullptr;
}

int lane_pass(brx_chain *ch, const PassParams &p, const LanePassInfo &info, hipStream_t s)
{
    // BRX_LANE=0: the group kernel only.  The window of a round holds 8 bases: look-aheads up to off + c + 1 <= 8.
    if (env_u32("BRX_LANE", 1u) == 0u || p.flip || p.c > 5 || p.k > 32 || p.n_reads == 0)
        return BRX_ERR_UNSUPPORTED;
    const bool idx = p.idx.lines != nullptr;
    if (!idx && !p.bits)
        return BRX_ERR_UNSUPPORTED;
    // chunk length: enough units to keep ~4.6e5 lanes busy a few times over, not so short that the stretch a unit's
    // predecessor re-scans in front of its sync point (a few dozen bases) becomes the larger part
    constexpr uint64_t RESIDENT = 256ull * 4ull * 7ull * 64ull;
    uint32_t C = env_u32("BRX_LANE_CHUNK", 0u);
    if (C == 0u) {
        const uint64_t want = info.in_total_bound / (2ull * RESIDENT);
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
    const uint64_t units_bound = info.in_total_bound / C + (uint64_t)p.n_reads + 1ull;
    const uint64_t x_bound = info.in_total_bound + (info.in_total_bound >> 2) * p.slack + 64ull * (units_bound + p.n_reads) + 256ull;
    if (units_bound >= 0xfffffff0ull)
        return BRX_ERR_UNSUPPORTED;
    if (w->reads_cap < p.n_reads) {
        const uint64_t cap = (uint64_t)p.n_reads + p.n_reads / 8 + 64;
        BRX_TRY(grow_dev((void **)&w->nu, cap * 4));
        BRX_TRY(grow_dev((void **)&w->xsz, cap * 4));
        BRX_TRY(grow_dev((void **)&w->fail_list, cap * 4));
        BRX_TRY(grow_dev((void **)&w->ubase, (cap + 1) * 8));
        BRX_TRY(grow_dev((void **)&w->xbase, (cap + 1) * 8));
        w->reads_cap = cap;
    }
    if (w->units_cap < units_bound) {
        const uint64_t cap = units_bound + units_bound / 8 + 64;
        BRX_TRY(grow_dev((void **)&w->u_read, cap * 4));
        BRX_TRY(grow_dev((void **)&w->u_q, cap * 4));
        BRX_TRY(grow_dev((void **)&w->u_qk, cap * 8));
        BRX_TRY(grow_dev((void **)&w->u_res, cap * 16));
        w->units_cap = cap;
    }
    if (w->x_cap < x_bound) {
        const uint64_t cap = x_bound + x_bound / 16;
        BRX_TRY(grow_dev((void **)&w->X, cap));
        BRX_TRY(grow_dev((void **)&w->Y, cap));
        w->x_cap = cap;
    }
    {
        uint64_t tmp_bytes = ch->scan_tmp_cap;
        if (scan_tmp_bytes(p.n_reads) > tmp_bytes) {
            set_error("lane pass: scan scratch smaller than the batch");
            return BRX_ERR_ARG;
        }
    }
    LaneArgs a;
    a.p = p;
    a.C = C;
    a.R = R;
    a.nu = w->nu;
    a.xsz = w->xsz;
    a.ubase = w->ubase;
    a.xbase = w->xbase;
    a.u_read = w->u_read;
    a.u_q = w->u_q;
    a.u_qk = w->u_qk;
    a.u_res = w->u_res;
    a.X = w->X;
    a.Y = w->Y;
    a.fail_list = w->fail_list;

    const uint32_t rb = (p.n_reads + 255u) / 256u;
    {
        KernelTimer t("lane_units", s);
        BRX_HIP(hipMemsetAsync(p.ctrl + CTL_LANE_UNITS, 0, (CTL_N - CTL_LANE_UNITS) * 8, s));
        lane_units_kernel<<<rb, 256, 0, s>>>(a);
        BRX_TRY(exclusive_scan_lens(w->nu, p.n_reads, ch->d_scan_tmp, w->ubase, p.ctrl + CTL_LANE_UNITS, s));
        BRX_TRY(exclusive_scan_lens(w->xsz, p.n_reads, ch->d_scan_tmp, w->xbase, p.ctrl + CTL_LANE_XBYTES, s));
        lane_fill_kernel<<<rb, 256, 0, s>>>(a);
    }
    {
        KernelTimer t("lane_sync", s);
        const uint64_t waves = units_bound < 256ull * 32ull ? units_bound : 256ull * 32ull;
        const uint32_t blocks = (uint32_t)((waves + 3) / 4);
        if (idx)
            lane_sync_kernel<true><<<blocks, 256, 0, s>>>(a);
        else
            lane_sync_kernel<false><<<blocks, 256, 0, s>>>(a);
    }
    {
        KernelTimer t("correct_pass", s);
        const uint64_t want = (units_bound + 255ull) / 256ull;
        const uint32_t blocks = (uint32_t)(want < 256ull * 7ull ? want : 256ull * 7ull);
        if (idx)
            launch_lane<true>(a, blocks, s);
        else
            launch_lane<false>(a, blocks, s);
    }
    {
        KernelTimer t("lane_stitch", s);
        const uint32_t grid = p.n_reads < (1u << 16) ? p.n_reads : (1u << 16);
        lane_stitch_kernel<<<grid, 256, 0, s>>>(a);
    }
    {
        // the reads the units could not settle (two misses in a row, a stretch that outgrew its region): the group kernel
        KernelTimer t("lane_redo", s);
        PassParams q = p;
        q.only = w->fail_list;
        q.only_n = p.ctrl + CTL_LANE_FAIL;
        BRX_HIP(hipMemsetAsync(p.ctrl + CTL_WORK, 0, 8, s));
        BRX_TRY(launch_one_list(q, s));
    }
    BRX_HIP(hipGetLastError());
    return BRX_OK;
}

} // namespace brx
