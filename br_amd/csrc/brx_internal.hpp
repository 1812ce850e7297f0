// Internal declarations shared by the translation units of libbrx.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/brx.h"
#include "brx_kmer.hpp"

namespace brx {

void set_error(const char *fmt, ...);

#define BRX_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            brx::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));   \
            return (_e == hipErrorOutOfMemory) ? BRX_ERR_NOMEM : BRX_ERR_HIP;                      \
        }                                                                                          \
    } while (0)

#define BRX_TRY(expr)                                                                              \
    do {                                                                                           \
        int _s = (expr);                                                                           \
        if (_s != BRX_OK)                                                                          \
            return _s;                                                                             \
    } while (0)

// BRX_OK if `device` is a usable GPU and has been made current
int use_device(int device);
// pooled page-locked host memory (brx_api.hip): what brx_host_alloc and the callee-allocated outputs are made of
void *host_buf_acquire(size_t bytes);
void host_buf_release(void *p);
bool host_buf_is_pinned(const void *p);

// device memory through the block pool (brx_devpool.hip); every hipMalloc / hipFree below this header is one of these
hipError_t dev_alloc(void **out, size_t bytes);
hipError_t dev_free(void *p);
void dev_pool_trim();
size_t dev_pool_bytes();

// BRX_TRACE=1: synchronise `s` and print a time-stamped stage name on stderr (finding where a big job stalls)
void trace_stage(hipStream_t s, const char *what);

// ---- per-kernel HIP-event timers -----------------------------------------------------------
// Usage: { KernelTimer t("correct_pass", stream); launch<<<...,stream>>>(); }
// Events are recorded on the launch stream and resolved lazily by brx_profile_get.
class KernelTimer {
  public:
    KernelTimer(const char *name, hipStream_t s);
    ~KernelTimer();

  private:
    int slot_;
    hipStream_t s_;
    hipEvent_t start_, stop_;
    bool on_;
};

// number of 32-bit words of the bitset for a given k (at least 1)
inline uint64_t set_nbits(int k) { return 1ull << (2 * k - 1); }
inline uint64_t set_nwords(int k) { uint64_t b = set_nbits(k); return b < 32 ? 1 : b / 32; }
inline uint64_t set_nbytes_file(int k) { return (set_nbits(k) + 7) / 8; }

} // namespace brx

struct brx_set {
    int k;
    int device;
    uint64_t nwords;   // u32 words
    uint32_t *d_bits;  // nwords, zero padded; nullptr for a sparse set
    // sparse set (k >= 21: 2^(2k-1) bits do not fit): the solid hashes live in the key list and in the probe
    // index only, whose overflowing lines then chain into the next line instead of falling back to the bits
    bool sparse;
    // lazy bit vector (k <= 19): a partitioned finish leaves only the key list; the bits are materialised from it
    // (or from the chained index) by the first entry point that really needs them -- correction with One does not
    bool bits_stale;
    bool idx_exact;          // the index was built with chaining (answers without the bit vector)
    bool idx_open;           // ... and k-mer by k-mer (brx_set_insert_batch on a sparse set): more can be added
    // probe index over the same set (brx_index.hpp): built on demand, invalidated by every mutation
    // of the bits that goes through the ABI
    uint64_t *d_lines;       // 8 u64 per line
    uint64_t lines_alloc;    // lines allocated (the allocation also holds lines_alloc / 8 bytes of occupancy bits behind the lines)
    bool idx_linebits = false; // the occupancy bits are current (indexes built from a key list)
    uint32_t idx_log_lines;
    uint32_t idx_m;
    bool idx_valid;
    bool idx_declined;       // index_ensure looked at the set and found an index would not help
    // successor table beside the index (brx_onelane.hip): per line, one byte per slot -- for either orientation of the
    // slot's k-mer, whether it has exactly one solid successor and which.  Built on first use by a walking corrector,
    // stale whenever the index changes (idx_gen counts the changes).
    uint64_t *d_succ = nullptr;
    uint64_t succ_lines = 0;     // lines the table was allocated for
    uint64_t succ_gen = ~0ull;   // idx_gen it was built at
    uint64_t idx_gen = 0;
    uint64_t idx_keys, idx_overflow_keys;
    // solid hashes of the current bits, when the builder produced them on the side (partitioned finish)
    uint64_t *d_keylist;
    uint64_t keylist_cap;                 // entries
    unsigned long long *d_keylist_n;      // device counter (may exceed keylist_cap: list truncated)
    bool keylist_valid;
    std::mutex idx_mu;
};

namespace brx {
struct IdxView;
// (re)builds the index from a device list of keys (bit indices = canonical >> 1); m / log_lines 0 = auto
int index_build_from_keys(brx_set *set, const uint64_t *d_keys, uint64_t n, int m, int log_lines, hipStream_t s);
int index_insert_reads(brx_set *set, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n_reads, uint64_t total_bases,
                       hipStream_t s);
// presence-only insertion over the flat base stream (brx_partbuild.hip); table == nullptr: atomicOr into `bits`
int flat_presence_insert(const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n_reads, uint64_t total_bases, int k,
                         uint32_t *bits, uint64_t *table, uint32_t line_shift, uint32_t m, unsigned long long *d_new, hipStream_t s);
// sets of this k have no bit vector
inline bool sparse_k(int k) { return k >= 21; }
// the set must be probed through its (chained) index: it has no bit vector, or not right now
inline bool no_bits(const brx_set *set) { return set->sparse || set->bits_stale; }
// materialises a stale bit vector (BRX_ERR_UNSUPPORTED for a sparse set)
int ensure_bits(const brx_set *set, hipStream_t s, const char *what);
// builds the index from the bitset when the set has none (no-op for k outside the indexed range)
int index_ensure(const brx_set *set, hipStream_t s);
inline void index_invalidate(brx_set *set) { set->idx_valid = false; set->idx_declined = false; set->keylist_valid = false; }
int index_auto_m(int k, uint64_t n_keys);
bool index_wanted(int k);
}

namespace brx { struct PartState; }

struct brx_counter {
    int k;
    int device;
    int strategy;
    // dense: u8 table packed in u32 words (2^(2k-1) bytes, padded to 32 B)
    uint32_t *d_counts;
    uint64_t count_bytes;
    // partitioned strategy (brx_partbuild.hip)
    brx::PartState *part;
    uint64_t *d_keys;
    uint64_t n_keys, cap_keys;
    hipStream_t stream; // owned, used by host-pointer entry points
    std::mutex mu;
};

struct brx_chain {
    const brx_set *set;
    int device;
    std::vector<brx_method_t> methods;
    bool two_side;
    hipStream_t stream; // owned
    // workspace (grown on demand)
    uint8_t *d_stage[2];
    uint64_t stage_bytes;
    uint32_t *d_lens[2];
    uint64_t lens_cap;
    uint64_t *d_scan_tmp;
    uint64_t scan_tmp_cap;
    uint64_t *d_path;   // graph-walk visited lists (Graph / GapSize)
    uint32_t *d_rev_list = nullptr; // reads the lean reverse pass handed back (rev_scan_kernel)
    uint64_t rev_list_bytes = 0;
    uint32_t *d_rev_flag = nullptr; // ... one word per read: entered into that list already
    uint64_t rev_flag_bytes = 0;
    void *d_rev_trig = nullptr;     // triggers the lean reverse pass left to the verify pass (TrigRec)
    uint64_t rev_trig_bytes = 0;
    uint64_t path_bytes;
    uint64_t *d_ctrl;   // device control block (work counter, overflow, stats)
    uint64_t *h_ctrl;   // pinned mirror
    // host-entry staging
    uint8_t *d_in;
    uint64_t d_in_cap;
    uint64_t *d_off;
    uint64_t d_off_cap;
    uint8_t *d_out;
    uint64_t d_out_cap;
    uint64_t *d_out_off;
    uint64_t d_out_off_cap;
    uint64_t last_stats[8];
    // workspace sizes that a batch of this chain has needed so far: the next batch starts from them, so that one
    // long walk or one read that grows a lot costs its own batch a second run, not every batch after it
    uint32_t slack_seen = 1, maxpath_seen = 4096;
    // second chain (same set, methods and direction rule, its own workspace) that redoes the few reads of a batch
    // whose graph walks outgrew the visited list; created on first need
    brx_chain *sub = nullptr;
    bool is_sub = false;
    void *lane_ws = nullptr; // workspace of the lane-per-chunk pass (brx_onelane.hip)
    uint8_t *h_in = nullptr; // page-locked bounce block for batches handed over in pageable memory
    uint64_t h_in_cap = 0;
    void *async_job = nullptr; // a batch in flight (brx_chain_correct_batch_async), owned by the chain
    std::mutex mu;
};

// The library's device memory comes from the block pool (brx_devpool.hip, which does not include this header).
#define hipMalloc(ptr, bytes) brx::dev_alloc((void **)(ptr), (bytes))
#define hipFree(ptr) brx::dev_free((void *)(ptr))
