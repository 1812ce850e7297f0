// Shared by the correction kernels (brx_correct.hip: groups of lanes per read; brx_onelane.hip: one lane per chunk of a
// read): the pass description, the staging-slot layout, the control block and the wave-level k-mer scans.
#pragma once
#include "brx_internal.hpp"
#include "brx_index.hpp"

namespace brx {

// control block layout (u64 words)
enum { CTL_WORK = 0, CTL_OVERFLOW = 1, CTL_ROUNDS = 2, CTL_PROBES = 3, CTL_TRIGGERS = 4, CTL_FIXES = 5, CTL_TOTAL = 6, CTL_PATHOVF = 7, CTL_NONTERM = 8,
       // the lane-per-chunk form of One's forward pass (brx_onelane.hip): units of the pass, predictions that missed,
       // its work counter, reads handed back to the group kernel
       CTL_LANE_UNITS = 9, CTL_LANE_MISS = 10, CTL_LANE_WORK = 11, CTL_LANE_FAIL = 12,
       // (not reset per pass: summed over the attempt) unit records the replay kernel found UNWRITTEN -- every unit is
       // taken by exactly one lane and leaves a record, so this stays 0; the records are filled with ones before the
       // automaton runs so that a unit nobody scanned could not pass for a result (its read goes to the group kernel)
       // the lean reverse pass (rev_scan_kernel, brx_correct.hip): reads handed back to the group kernel, this pass / summed
       CTL_REV_HANDBACK = 13, CTL_REV_HANDBACK_SUM = 14,
       CTL_LANE_UNWRITTEN = 15,
       // ... and the triggers it left to the verify pass (this pass / summed)
       CTL_REV_TRIGS = 16, CTL_REV_TRIGS_SUM = 17,
       CTL_N = 20 };

// A trigger of a reverse pass that the lean scan could not settle itself (alt_nucs named exactly one alternative): the scan
// goes on as if the method returned None, and the group kernel checks that it does (SRC == 2 of correct_kernel).
struct TrigRec {
    uint32_t r;    // read
    uint32_t i;    // position of the trigger (the base that made the k-mer non-solid)
    uint32_t elen; // error_len's result (Graph / GapSize; the rest of the read when it ran off the end)
    uint32_t pad;
    uint64_t fc;   // first_correct_kmer
};

struct PassParams {
    const uint32_t *bits;
    IdxView idx;      // lines == nullptr: every probe goes to the bitset
    int k;
    int c;            // confirm
    uint32_t n_reads;
    const uint64_t *offsets; // original batch offsets (n_reads+1), relative to batch start
    // input view
    const uint8_t *in;       // original bases or staging buffer
    const uint32_t *in_lens; // nullptr => lengths from offsets (original batch)
    int in_staged;           // 0: read r starts at offsets[r]; 1: at slot(r)
    int flip;                // read the input back to front
    // output view (always staged)
    uint8_t *out;
    uint32_t *out_lens;
    uint32_t slack;          // slot(r) = o + (o>>2)*slack + 64*r
    unsigned long long *ctrl;
    // graph walks (Graph, GapSize): per-group list of visited k-mers, maxpath entries each
    uint64_t *path_k;
    uint32_t maxpath;
    // greedy (greedy.rs): max_search and the per-group LDS carve-up for the alignment
    uint32_t flags;       // tuning switches for A/B runs (BRX_TUNE): 1 no look-ahead reuse, 2 four ALTS probes, 4 unstaged SCEN
    int max_search;
    uint32_t g_dim;       // max (m+1), (n+1) of the DP = k + max_search + 2
    uint32_t g_lds_bytes; // bytes of dynamic LDS per group
    // list mode (one_kernel<G, K, true>): the kernel takes reads only[0 .. *only_n) instead of 0 .. n_reads
    const uint32_t *only = nullptr;
    const unsigned long long *only_n = nullptr;
    // verify mode (correct_kernel<G, M, 2>): the work items are the triggers trig[0 .. min(*only_n, trig_cap)); a read
    // whose trigger does NOT end in None is entered once into redo_list (count at ctrl[CTL_REV_HANDBACK], redo_flag dedupes)
    const TrigRec *trig = nullptr;
    uint32_t trig_cap = 0;
    uint32_t *redo_list = nullptr;
    uint32_t *redo_flag = nullptr;
};

__host__ __device__ __forceinline__ uint64_t slot_of(uint64_t o, uint64_t r, uint32_t slack)
{
    return o + (o >> 2) * (uint64_t)slack + 64ull * r;
}

__device__ __forceinline__ bool probe(const uint32_t *__restrict__ bits, uint64_t fwd, int k)
{
    const uint64_t h = khash(fwd, k);
    return (bits[h >> 5] >> (h & 31u)) & 1u;
}


#if defined(__HIPCC__)
template <int D>
__device__ __forceinline__ uint32_t dpp_row_shr(uint32_t v)
{
    // lane l of a 16-lane row reads lane l - D of the same row; lanes without such a source read 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + D, 0xf, 0xf, true);
}

// codes of the row's lanes first..this one, this lane's code in bits 0-1, the lane before in bits 2-3, ...
__device__ __forceinline__ uint32_t row_scan16(uint32_t code)
{
    uint32_t v = code;
    v |= dpp_row_shr<1>(v) << 2;
    v |= dpp_row_shr<2>(v) << 4;
    v |= dpp_row_shr<4>(v) << 8;
    v |= dpp_row_shr<8>(v) << 16;
    return v;
}
// this lane's k-mer in a 64-lane group: `carry` extended by the codes of lanes 0..lane (own row by the DPP scan, the two
// rows before by row_bcast:15; rows 0 / 1 take those words from `carry`)
__device__ __forceinline__ uint64_t lane_kmer64_dpp(uint64_t carry, uint32_t code, int lane, uint64_t mask)
{
    const uint32_t v = row_scan16(code);
    const uint32_t clo = (uint32_t)carry, chi = (uint32_t)(carry >> 32);
    const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp((int)clo, (int)v, 0x142 /* row_bcast:15 */, 0xe, 0xf, false);
    const uint32_t old2 = (lane < 16) ? chi : clo;
    const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp((int)old2, (int)b1, 0x142, 0xc, 0xf, false);
    const uint32_t jb = 2u * ((uint32_t)(lane & 15) + 1u);
    return (((((uint64_t)b2 << 32) | b1) << jb) | v) & mask; // jb <= 32
}

#endif

struct LanePassInfo {
    uint64_t in_total_bound; // upper bound of the pass's input bases (sizes the unit tables and the unit staging)
    int method;              // BRX_ONE, BRX_GRAPH or BRX_GAP_SIZE
};
// One's forward pass, one lane per chunk of a read (brx_onelane.hip).  BRX_ERR_UNSUPPORTED: not applicable to this
// pass (the caller runs the group kernel instead).
int lane_pass(brx_chain *ch, const PassParams &p, const LanePassInfo &info, hipStream_t s);
void lane_ws_free(brx_chain *ch);
// the group kernel over a list of reads (p.only / p.only_n), brx_correct.hip
int launch_one_list(const PassParams &p, hipStream_t s);
int launch_walk_list(const PassParams &p, int method, hipStream_t s);

} // namespace brx
