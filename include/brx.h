/*
 * brx.h -- C ABI of the MI355X-native replacement for natir/br's hot path:
 * solid-k-mer set build (count -> threshold -> packed canonical bitset) and the
 * per-read correction scan (br::correct::{One,Two,Graph,Greedy,GapSize} against
 * br::set::KmerSet).
 *
 * The reference (Rust, /root/reference) has no FFI layer; its seam is two traits
 * and two free functions.  Each entry point below names the reference interface it
 * replaces (file:line under /root/reference).  INTEGRATION.md shows the Rust-side
 * binding (`impl KmerSet for HipSet`, `run_correction` calling the batch entry).
 *
 * Conventions
 *   - every function returns 0 (BRX_OK) or a negative brx_status; nothing unwinds;
 *     brx_last_error() gives a thread-local message for the last failure.
 *   - "bases/offsets" batches: reads concatenated as ASCII bytes, read r =
 *     bases[offsets[r] .. offsets[r+1]); offsets has n_reads+1 entries.
 *   - k-mers are FORWARD 2-bit k-mers (A=0 C=1 T=2 G=3, first base most
 *     significant, 2k low bits) exactly as `KmerSet::get` receives them; the
 *     library canonicalises (even-popcount member of {kmer, revcomp}) and indexes
 *     bit (canonical >> 1), like pcon::solid::Solid.
 *   - `_device` variants take device pointers (HIP, same device as the handle) and a
 *     hipStream_t passed as void*; the plain variants take host pointers and do the
 *     transfers themselves.
 *   - a brx_set_t is immutable once built/loaded and may be shared by any number of
 *     chains; a brx_chain_t owns its workspace and serialises concurrent calls.
 *   - what a set IS is the reference's: the canonical k-mers with `KmerSet::get` = true.
 *     How it is HELD in HBM varies (bit vector for k <= 19, possibly lazy; key list +
 *     probe index; sparse table for k >= 21) and never changes an answer -- see
 *     brx_set_bits_state / brx_set_sparse / brx_set_index_* below and DESIGN.md 3, 5.
 *   - there is NO CPU fallback: without a usable gfx950 device every compute entry
 *     returns BRX_ERR_NODEVICE.
 */
#ifndef BRX_H
#define BRX_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct brx_set brx_set_t;         /* Box<dyn KmerSet> backed by set::Pcon   src/set.rs:17-23, src/set/pcon.rs:13-15 */
typedef struct brx_counter brx_counter_t; /* pcon::counter::Counter<u8>             src/main.rs:73-78 */
typedef struct brx_chain brx_chain_t;     /* Vec<Box<dyn Corrector>> of build_methods  src/lib.rs:141-164 */

typedef enum brx_status {
    BRX_OK = 0,
    BRX_ERR_ARG = -1,         /* bad argument (k even/out of range, null pointer, ...) */
    BRX_ERR_NOMEM = -2,       /* host or device allocation failed                      */
    BRX_ERR_HIP = -3,         /* a HIP runtime call failed                             */
    BRX_ERR_NODEVICE = -4,    /* no usable GPU: the product path has no CPU fallback   */
    BRX_ERR_FORMAT = -5,      /* malformed .solid stream (error.rs: wraps pcon errors) */
    BRX_ERR_UNSUPPORTED = -6, /* valid in the reference, not yet implemented here      */
    BRX_ERR_OVERFLOW = -7     /* caller-provided output buffer too small               */
} brx_status;

/* CorrectionMethod, src/cli.rs:11-17; order = build_methods' match, src/lib.rs:150-160 */
enum { BRX_ONE = 0, BRX_TWO = 1, BRX_GRAPH = 2, BRX_GREEDY = 3, BRX_GAP_SIZE = 4 };

/* one entry of the method list handed to build_methods (src/lib.rs:141-146):
 * confirm = -C (default 5, src/cli.rs:135-137), max_search = -M (default 7, :140-142) */
typedef struct brx_method {
    uint8_t method;
    uint8_t confirm;
    uint8_t max_search;
} brx_method_t;

/* ---- diagnostics -------------------------------------------------------------------- */
const char *brx_strerror(int status);
const char *brx_last_error(void);
int brx_version(void);
int brx_device_count(int *n); /* BRX_OK and *n == 0 when no GPU is visible */

/* per-kernel HIP-event timers (events recorded on the launch stream).  Names:
 * "count_dense", "threshold", "correct_pass", "compact", ...          */
int brx_profile_enable(int on);
int brx_profile_reset(void);
int brx_profile_get(const char *kernel, double *total_ms, uint64_t *launches);
int brx_profile_names(char *buf, size_t cap); /* comma-separated list of timer names */

/* ---- KmerSet: src/set.rs:17-21 --------------------------------------------------------- */
/* Pcon::new(Solid::new(k)): empty set                              src/set/pcon.rs:183-185 */
int brx_set_new(uint8_t k, int device, brx_set_t **out);
/* Pcon::from_pcon_solid: [k:u8][bits Lsb0], already decompressed   src/set/pcon.rs:18-25   */
int brx_set_new_from_solid_bytes(const uint8_t *buf, size_t len, int device, brx_set_t **out);
/* Pcon::from_fasta (presence only): OR in every canonical k-mer of every read of length >= k
 *                                                                  src/set/pcon.rs:47-112  */
int brx_set_insert_batch(brx_set_t *set, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads);
/* same on device buffers, asynchronous on `stream` for bit-vector sets                          */
int brx_set_insert_batch_device(brx_set_t *set, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n_reads,
                                uint64_t total_bases, void *stream);
/* Solid::set(kmer, value) -- host-side slow path (tests, csv input) src/set/pcon.rs:38      */
int brx_set_set(brx_set_t *set, uint64_t forward_kmer, bool value);
/* KmerSet::get                                                     src/set/pcon.rs:189-191 */
bool brx_set_get(const brx_set_t *set, uint64_t forward_kmer);
/* n independent KmerSet::get calls in one launch; out[i] = 0/1 */
int brx_set_get_batch(const brx_set_t *set, const uint64_t *forward_kmers, uint32_t n, uint8_t *out);
/* KmerSet::k                                                       src/set/pcon.rs:193-195 */
uint8_t brx_set_k(const brx_set_t *set);
int brx_set_device(const brx_set_t *set);
/* .solid writer ([k][bits]); len = 1 + 2^(2k-4) (at least 2)                               */
int brx_set_export_solid_bytes(const brx_set_t *set, uint8_t *buf, size_t cap, size_t *len);
int brx_set_popcount(const brx_set_t *set, uint64_t *n_set_bits);
/* 1 for a SPARSE set: k >= 21, where the 2^(2k-1)-bit vector (256 GiB at k = 21) does not fit next to the data.
 * Such a set holds its solid k-mers only as a key list and as the probe index (exact, overflowing lines chain
 * into the next line).  get / get_batch / popcount / count_finish[_into] / the correction entry points work;
 * entries that need the bit vector (.solid import/export, set, insert_batch, device_bits, extract/or_keys)
 * return BRX_ERR_UNSUPPORTED.  Sets are built by the partitioned counter (k <= 21).                          */
int brx_set_sparse(const brx_set_t *set);
/* 0: the bit vector is in HBM and current; 1: LAZY -- a partitioned brx_set_count_finish[_into] listed the solid
 * hashes and left the 2^(2k-4)-byte vector unwritten (16 GiB at k = 19); it is materialised by the first entry
 * that needs it (export, device_bits, set, insert_batch, extract/or_keys, correction with a walking method);
 * get / get_batch / popcount / correction with One answer from the key list / probe index (BRX_LAZY_BITS=0
 * writes the vector at finish time as the reference does); 2: sparse set, no bit vector ever.                */
int brx_set_bits_state(const brx_set_t *set);
/* device view of the packed bit array (for RCCL all-gather / OR across ranks)              */
int brx_set_device_bits(const brx_set_t *set, void **d_bits, uint64_t *n_bytes);
/* sparse replication of the set: list the set bits of [first_hash, first_hash+n_hashes) (both
 * multiples of 32) as absolute bit indices, any order; OR a list of bit indices into a set      */
int brx_set_extract_keys_device(const brx_set_t *set, uint64_t first_hash, uint64_t n_hashes, uint64_t *d_out, uint64_t cap,
                                uint64_t *n_out, void *stream);
int brx_set_or_keys_device(brx_set_t *set, const uint64_t *d_keys, uint64_t n, void *stream);
/* ---- probe index: no counterpart in the reference (an HBM-locality copy of the same set) ------
 * KmerSet::get is one bit of a 2^(2k-1)-bit vector; the k-mers a read walks through hit unrelated
 * 64-byte lines.  The index stores the solid k-mers a second time in 64-byte lines addressed by the
 * k-mer's strand-symmetric minimizer, so that neighbouring k-mers of a read share a line.  It is exact
 * (full keys; a line that overflowed sends the probe back to the bitset) and only changes where
 * `get` reads.  The correction entry points build it on first use for 15 <= k <= 19 (BRX_INDEX=0
 * disables); every mutation of the bits through this ABI drops it.  m = minimizer length (odd, <= 15),
 * log2_lines = table size; 0 = automatic.                                                          */
int brx_set_index_build(brx_set_t *set, int m, int log2_lines, void *stream);
/* same, from a device list of the set's bit indices (e.g. the keys just exchanged between ranks)     */
int brx_set_index_build_from_keys_device(brx_set_t *set, const uint64_t *d_keys, uint64_t n, int m, int log2_lines,
                                         void *stream);
int brx_set_index_drop(brx_set_t *set);
/* the list of the set's bit indices (any order) when the builder produced one on the side (partitioned
 * brx_set_count_finish[_into]); *d_keys == NULL when there is none.  Valid until the set is mutated.
 * Lets the multi-GPU exchange ship the solid k-mers of a rank's range without scanning the bit vector.  */
int brx_set_keylist_device(const brx_set_t *set, void **d_keys, uint64_t *n, void *stream);
/* Three numbers that depend on the SET alone, not on what holds it (bit vector, solid-hash list, chained table):
 * out3[0] = members, out3[1] = sum of their hashes (canonical >> 1) mod 2^64, out3[2] = sum of the squares mod 2^64.
 * What the ranks of a multi-GPU job compare after the exchange (bench.py `checks.set_agree`): the reference has one
 * process and one set (src/main.rs:35-47); here every rank must end with the same one.                              */
int brx_set_fingerprint(const brx_set_t *set, uint64_t *out3, void *stream);
/* info8: [0] valid, [1] m, [2] log2_lines, [3] keys, [4] keys left to the bitset (overflow), [5] bytes,
 * [6] the correction entry points would use an index for this k, [7] a key list is attached          */
int brx_set_index_info(const brx_set_t *set, uint64_t *info8);
/* brx_set_get_batch answered through the index (+ bitset for overflowed lines); *n_fallback = probes
 * that needed the bitset                                                                             */
int brx_set_get_batch_indexed(const brx_set_t *set, const uint64_t *forward_kmers, uint32_t n, uint8_t *out,
                              uint64_t *n_fallback);
void brx_set_free(brx_set_t *set);

/* ---- set build by counting: src/main.rs:72-115 -------------------------------------------
 * Counter::<u8>::new(k) -> count_fasta -> Solid::from_count(k, counts, abundance).
 * k must be odd (Fasta::kmer_size forces it, src/cli.rs:277-279).
 * strategy: BRX_COUNT_DENSE keeps the reference's 2^(2k-1)-byte u8 table in HBM (k <= 19: 128 GiB);
 * BRX_COUNT_SORTED radix-partitions the canonical hashes and counts them bucket by bucket (same
 * set, no table; k <= 21); BRX_COUNT_AUTO = SORTED for 15 <= k <= 21, else DENSE.                 */
enum { BRX_COUNT_AUTO = 0, BRX_COUNT_DENSE = 1, BRX_COUNT_SORTED = 2 };
int brx_set_count_begin(uint8_t k, int device, int strategy, brx_counter_t **out);
int brx_set_count_add_batch(brx_counter_t *c, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads);
int brx_set_count_add_batch_device(brx_counter_t *c, const uint8_t *d_bases, const uint64_t *d_offsets,
                                   uint32_t n_reads, uint64_t total_bases, void *stream);
/* solid iff count > abundance (strict; pinned by tests/data/raw.k11.a2.solid).  The counter
 * stays valid (finish may be called again with another abundance) until freed.             */
int brx_set_count_finish(brx_counter_t *c, uint8_t abundance, void *stream, brx_set_t **out);
/* same, into an existing set of the same k/device (no allocation; asynchronous on `stream`)  */
int brx_set_count_finish_into(brx_counter_t *c, uint8_t abundance, void *stream, brx_set_t *dst);
/* forget everything counted so far (Counter::new without re-allocating; async on `stream`)   */
int brx_counter_reset(brx_counter_t *c, void *stream);
/* 256-bin histogram of the counts = pcon::spectrum::Spectrum::from_count (src/main.rs:93):
 * hist256[v] = number of canonical k-mers seen exactly v times (255 = 255 or more, 0 = never).  Either
 * strategy; the counter is left as it was, so brx_set_count_finish with the chosen threshold follows.   */
int brx_counter_spectrum(brx_counter_t *c, uint64_t *hist256, void *stream);
/* dense strategy only: device view of the u8 table, for the RCCL reduction of SURVEY 8(e)   */
int brx_counter_device_counts(brx_counter_t *c, void **d_counts, uint64_t *n_bytes);
/* dense strategy only: overwrite counters [first, first + n) with counts computed elsewhere -- `br count`
 * (src/main.rs:59-70: pcon::counter::Counter::from_stream) hands a whole table over this way        */
int brx_counter_load_counts(brx_counter_t *c, uint64_t first, const uint8_t *counts, uint64_t n);
/* clamp every count to min(count, cap) in place (the exact-sum trick before an u8 all-reduce) */
int brx_counter_clamp(brx_counter_t *c, uint8_t cap, void *stream);
/* partitioned strategy, multi-GPU exchange (done on keys instead of the count vector):
 * after ONE add_batch the canonical hashes sit grouped by their first radix digit.  l1_view exposes
 * that layout (u32 keys with the digit stripped, u64 offsets[n_buckets+1]); a rank owns a contiguous
 * range of buckets, receives the other ranks' segments of that range and registers them with
 * add_partitioned (the memory stays the caller's until the counter is finished/reset).           */
int brx_counter_l1_view(brx_counter_t *c, void **d_keys, void **d_l1off, uint32_t *n_buckets, uint64_t *n_keys);
int brx_counter_add_partitioned_device(brx_counter_t *c, const uint32_t *d_keys, const uint64_t *d_l1off, uint64_t n_keys);
void brx_counter_free(brx_counter_t *c);

/* ---- multi-GPU: reads shard over the GPUs, the set is exchanged ONCE -----------------------------
 * The reference is one process with rayon threads over shared memory (src/main.rs:30-33, src/lib.rs:72-139); on N
 * GPUs every rank counts and later corrects its own block of the reads, and the single exchange step is the k-mer
 * counts.  It runs inside this library over RCCL (xGMI), so a host needs no collective library of its own:
 *   - one process per GPU: rank 0 calls brx_comm_unique_id and hands the 128 bytes to the other ranks by any means
 *     it has (file, socket, MPI, torch.distributed); every rank then calls brx_comm_init(id, world, rank, device);
 *   - one process driving several GPUs (the reference's shape): brx_comm_init_all(n, devices, comms[n]), then one
 *     host thread per GPU.
 * librccl is dlopen'ed by soname on first use (BRX_RCCL_PATH overrides); BRX_ERR_UNSUPPORTED if it cannot be found. */
typedef struct brx_comm brx_comm_t;
#define BRX_COMM_ID_BYTES 128
int brx_comm_unique_id(uint8_t *id128);
int brx_comm_init(const uint8_t *id128, int world, int rank, int device, brx_comm_t **out);
int brx_comm_init_all(int n_devices, const int *devices, brx_comm_t **out /* n_devices handles */);
int brx_comm_info(const brx_comm_t *comm, int *world, int *rank, int *device);
/* `c` (partitioned strategy) has counted this rank's reads in ONE add_batch.  Every rank calls this once, together:
 * the canonical hashes go to the rank owning their first radix digit (all-to-all of u32 keys), the owner finishes its
 * digit range with `count > abundance`, and every owner's solid hashes go to everybody (all-gather of u64 lists).  On
 * return `dst` holds the solid set of ALL ranks' reads -- what the reference's all-reduce of the count vector followed
 * by Solid::from_count gives (src/main.rs:112-114) -- its probe index is built, and `c` is empty again.  k=19, 1 Gbp
 * per GPU: ~3.5 GB of keys + 0.15 GB of lists per rank over xGMI instead of 120 GB + 15 GB for the u8 vector.   */
int brx_exchange_build_partitioned(brx_comm_t *comm, brx_counter_t *c, uint8_t abundance, brx_set_t *dst, void *stream);
/* dense strategy, north_star's form: in-place all-reduce of the 2^(2k-1)-byte u8 table, exact for `count >
 * abundance` (counts are clamped to abundance+1 first; int32 slices once world*(abundance+1) > 255); the caller
 * then runs brx_set_count_finish[_into] on every rank.                                                          */
int brx_exchange_reduce_counts(brx_comm_t *comm, brx_counter_t *c, uint8_t abundance, void *stream);
/* The host-side arithmetic of brx_exchange_build_partitioned, a pure function (no GPU, no communicator): from the
 * `world` level-1 offset tables (each n_buckets+1 entries, non-decreasing from 0) it gives, for `rank`, the owner
 * bounds (world+1: rank r owns first digits [bound[r], bound[r+1]) = [r*n_buckets/world, ...)), the keys it sends to
 * and receives from every rank, optionally the per-source segment tables (world x (n_buckets+1): the source's table
 * clamped to the owned range and rebased to the received segment) and the largest single message of the job (what
 * fixes the number of capped rounds on every rank alike).  Exposed so that hosts and tests can check the layout for
 * any world size; there is no reference counterpart (one process, src/main.rs:30-33).                            */
int brx_exchange_plan(const uint64_t *tables, int world, uint32_t n_buckets, int rank, uint32_t *bound, uint64_t *send_counts,
                      uint64_t *recv_counts, uint64_t *seg /* may be NULL */, uint64_t *largest_message /* may be NULL */);
/* last build_partitioned: [0] key bytes sent, [1] received, [2] keys counted by this owner, [3] its solid k-mers,
 * [4] solid k-mers of the job, [5] all-to-all us, [6] whole call us, [7] largest single message (keys)          */
int brx_comm_last_stats(const brx_comm_t *comm, uint64_t *stats8);
void brx_comm_free(brx_comm_t *comm);

/* ---- correction: src/lib.rs:22-139 (run_correction) + src/correct/mod.rs:44-108 ----------
 * brx_chain_new = build_methods (order kept, duplicates allowed).  two_side=true means the
 * -s flag was given, i.e. the reverse pass is SKIPPED (src/lib.rs:48,110).                  */
int brx_chain_new(const brx_set_t *set, const brx_method_t *methods, uint32_t n_methods, bool two_side,
                  brx_chain_t **out);
/* one batch of records through the whole per-record body of run_correction
 * (src/lib.rs:42-55): every method in order, then (unless two_side) reverse, every method,
 * reverse back.  Output in input order.  *out_bases / *out_offsets are malloc'd by the
 * library; release with brx_buf_free.                                                       */
int brx_chain_correct_batch(brx_chain_t *chain, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                            uint8_t **out_bases, uint64_t **out_offsets);
/* The same batch, started and collected in two calls, so that ONE host thread can keep several chains of a set busy:
 * the next batch's upload runs under this one's kernels under the last one's download (a chain owns a stream and a
 * workspace; batches on different chains overlap, a chain takes one batch at a time).  `bases` / `offsets` must stay
 * valid until _wait returns; _wait blocks, returns what brx_chain_correct_batch would have returned and hands out the
 * same callee-allocated buffers.  This is the loop run_correction's 8192-record batches map to (src/lib.rs:84-132):
 * INTEGRATION.md shows it.  brx_chain_free waits for a batch still in flight.                                          */
int brx_chain_correct_batch_async(brx_chain_t *chain, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads);
int brx_chain_correct_batch_wait(brx_chain_t *chain, uint8_t **out_bases, uint64_t **out_offsets);
/* same on device buffers: d_out (capacity out_cap bytes) and d_out_offsets (n_reads+1) are
 * caller-allocated; *out_total receives the number of corrected bytes.  Returns
 * BRX_ERR_OVERFLOW (with *out_total = needed) if out_cap is too small.  The call returns
 * after its kernels have completed on `stream`.                                            */
int brx_chain_correct_batch_device(brx_chain_t *chain, const uint8_t *d_bases, const uint64_t *d_offsets,
                                   uint32_t n_reads, uint64_t total_bases, uint8_t *d_out, uint64_t out_cap,
                                   uint64_t *d_out_offsets, uint64_t *out_total, void *stream);
/* counters of the last batch: [0] scan rounds, [1] probes issued, [2] triggers, [3] fixes -- the fixes WRITTEN into the
 * reads that came back (Some(..) returns of correct_error, src/correct/mod.rs:75: the oracle's count), whereas
 * [0]-[2] count work done, speculative stretches included --,
 * [4] overflow retries, [5] reads that outgrew their output slot and [6] graph walks that outgrew the
 * visited list in the attempts that were redone; [7] the last lane-per-chunk pass (forward scan cut into units at
 * predictable states): units in bits 0-31, reads handed back to the group kernel in bits 32-55
 * (three predictions in a row missed, a list overflowed), and in bits 56-63 the unit records the replay found
 * unwritten (an invariant: always 0; the suites assert it); 0 = that form did not run                          */
int brx_chain_last_stats(const brx_chain_t *chain, uint64_t *stats8);
void brx_chain_free(brx_chain_t *chain);
void brx_buf_free(void *p);
/* Page-locked host memory for batch buffers.  brx_chain_correct_batch / brx_set_count_add_batch accept ANY host
 * pointer, but a batch in pageable memory is first copied into a page-locked block by the library (the DMA engines
 * read nothing else at full speed); a caller that fills buffers from brx_host_alloc -- the reference's
 * populate_buffer, src/lib.rs:168-188, is the natural place -- skips that copy.  The blocks are pooled; release with
 * brx_host_free (or brx_buf_free: the callee-allocated outputs of the batch entries are the same kind of block).      */
void *brx_host_alloc(size_t bytes);
void brx_host_free(void *p);
/* Device memory of the library is pooled too: blocks of 32 MiB and more are kept when a counter, set or chain is freed
 * (up to BRX_DEVPOOL_GB GiB per process, default 48; 0 = no pool) and reused by the next one -- the reference builds and
 * drops its 2^(2k-1)-byte counter table and its set once per run (src/main.rs:60-115), a host that runs job after job
 * would otherwise pay the driver's unmap / map of gigabytes every time.  brx_devpool_trim gives everything that is parked
 * back to the runtime (a host that needs the HBM for something else); brx_devpool_bytes says how much that is.            */
void brx_devpool_trim(void);
uint64_t brx_devpool_bytes(void);

/* ---- host pipelines over file descriptors ----------------------------------------
 * run_correction (src/lib.rs:22-139) for ONE input/output pair: FASTA records are parsed from in_fd (plain
 * text; decompress upstream), corrected by build_methods(methods) against `set` (reverse pass unless
 * two_side), and written to out_fd in input order as '>name[ description]' + the sequence wrapped at 80
 * columns (noodles' reader / writer conventions, restated; unpinned by the reference's tests).  Parsing, the
 * GPU (two chains on two streams, each worker also lays out the 80-column text of its batch) and writing run on
 * their own threads.  A malformed record ends the stream silently after the records before it (src/lib.rs:35).
 * max_batch_records 0 = default (batches of 32 MB of bases, BRX_PIPE_BATCH_MB); the batch size does not change
 * a byte of the output.
 * stats8: [0] records, [1] bases in, [2] bases out, [3] batches, [4..7] ns parsing / GPU + formatting (summed
 * over the workers) / writing / wall.                                                                        */
int brx_run_correction_fd(const brx_set_t *set, const brx_method_t *methods, uint32_t n_methods, bool two_side, int in_fd,
                          int out_fd, uint32_t max_batch_records, uint64_t *stats8);
/* Counter::count_fasta(reader, record_buffer) (src/main.rs:73-78): counts every record of the FASTA stream */
int brx_count_fasta_fd(brx_counter_t *c, int in_fd, uint32_t max_batch_records, uint64_t *stats8);
/* Pcon::from_fasta / Hash::from_fasta (src/set/pcon.rs:47-112, src/set/hash.rs:40-60): every record of the stream
 * inserted presence-only (`br solid -f fasta`, `br large-kmer -f fasta`)                                          */
int brx_set_insert_fasta_fd(brx_set_t *set, int in_fd, uint32_t max_batch_records, uint64_t *stats8);

/* ---- synthetic reads: deterministic, identical on host and device ----------
 * genome: i.i.d. uniform ACGT of length genome_len (seed); read r: window of read_len
 * reference bases at a uniform start, strand +/- with p=1/2, per-reference-base errors
 * sub/ins/del at the given rates (parts per 10 000).  Output offsets are fixed-stride
 * capacity slots compacted by the call; *total receives the number of bases.               */
typedef struct brx_synth {
    uint64_t seed;
    uint64_t genome_len;
    uint32_t read_len;
    uint32_t sub_e4, ins_e4, del_e4;
} brx_synth_t;
int brx_synth_genome_device(const brx_synth_t *cfg, int device, uint8_t *d_genome, void *stream);
int brx_synth_reads_device(const brx_synth_t *cfg, int device, const uint8_t *d_genome, uint64_t first_read,
                           uint32_t n_reads, uint8_t *d_bases, uint64_t bases_cap, uint64_t *d_offsets,
                           uint64_t *total, void *stream);
int brx_synth_genome_host(const brx_synth_t *cfg, uint8_t *genome);
int brx_synth_reads_host(const brx_synth_t *cfg, const uint8_t *genome, uint64_t first_read, uint32_t n_reads,
                         uint8_t *bases, uint64_t bases_cap, uint64_t *offsets, uint64_t *total);

#ifdef __cplusplus
}
#endif
#endif /* BRX_H */
