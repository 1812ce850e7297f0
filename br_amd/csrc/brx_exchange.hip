// Multi-GPU set exchange behind the C ABI: RCCL over xGMI, called from this library.
//
// Reference: the path has ONE exchange step -- the k-mer counts, once per job.  Reads shard
// over the GPUs (src/lib.rs:72-139 is the loop being sharded), every rank counts its own block, then
//
//   partitioned strategy (k >= 15)  the canonical hashes a rank has radix-partitioned by their first digit travel
//       to the rank that owns the digit (one all-to-all of u32 keys = ncclSend/ncclRecv in one group), the owner
//       finishes its digit range, and the solid hashes come back to everybody (all-gather-v of u64 lists);
//   dense strategy                  the reference's own u8 table is all-reduced (north_star's form): counts are
//       clamped to min(c, a+1) first so that an u8 SUM over `world` ranks cannot wrap while world*(a+1) <= 255;
//       beyond that the table is reduced in int32 slices.
//
// librccl is NOT a link-time dependency: it is dlopen'ed by soname on first use, so that a process which
// already carries a copy (PyTorch bundles one under the same soname) shares it, and hosts that never go
// multi-GPU never load it.  A communicator is made from a 128-byte unique id the host passes between its
// ranks by whatever means it has (one process per GPU: a file, a socket, torch.distributed; one process
// driving several GPUs: brx_comm_init_all).
#include "brx_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>
#include <stdlib.h>
#include <chrono>
#include <thread>
#include <vector>

using namespace brx;

namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr; // optional
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_rccl_mu;
Rccl g_rccl;

int rccl_load()
{
    std::lock_guard<std::mutex> g(g_rccl_mu);
    if (g_rccl.h)
        return BRX_OK;
    const char *names[] = {getenv("BRX_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)))
            break;
    if (!h) {
        set_error("multi-GPU exchange: cannot load librccl (%s)", dlerror());
        return BRX_ERR_UNSUPPORTED;
    }
    Rccl r;
    r.h = h;
#define SYM(field, name)                                                    \
    *(void **)(&r.field) = dlsym(h, name);                                  \
    if (!r.field) {                                                         \
        set_error("librccl: symbol %s missing", name);                      \
        return BRX_ERR_UNSUPPORTED;                                         \
    }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(AllGather, "ncclAllGather")
    SYM(AllReduce, "ncclAllReduce")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    *(void **)(&r.CommAbort) = dlsym(h, "ncclCommAbort"); // (not required: without it a failed rank can only leave)
    g_rccl = r;
    return BRX_OK;
}

#define BRX_NCCL(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t _r = (expr);                                                                        \
        if (_r != ncclSuccess) {                                                                         \
            set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r));          \
            return BRX_ERR_HIP;                                                                          \
        }                                                                                                \
    } while (0)

// per-peer message cap in elements (BRX_A2A_CHUNK; 0 = one message per peer whatever its size).  A single 4 GB
// message per peer arrived truncated at 2^31 bytes through torch.distributed's all_to_all_single on this stack
// (round 1); whether ncclSend/ncclRecv themselves carry one has never been measured (one-GPU boxes), so the cap stays.
uint64_t a2a_chunk_elems()
{
    const char *e = getenv("BRX_A2A_CHUNK");
    if (e && *e)
        return strtoull(e, nullptr, 10);
    return 1ull << 28; // 1 GiB of u32 keys
}

__global__ void widen_u8_kernel(const uint8_t *__restrict__ in, int *__restrict__ out, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = in[i];
}

__global__ void narrow_sat_kernel(const int *__restrict__ in, uint8_t *__restrict__ out, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = (uint8_t)(in[i] > 255 ? 255 : in[i]);
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

struct brx_comm {
    int world = 1, rank = 0, device = 0;
    ncclComm_t comm = nullptr;
    // workspace kept between jobs
    uint32_t *d_recv = nullptr;    // keys received from every rank for the owned digit range
    uint64_t recv_cap = 0;         // bytes
    uint64_t *d_tables = nullptr;  // world x (B1+1) gathered level-1 offset tables, then the owner's segment tables
    uint64_t tables_cap = 0;
    uint64_t *d_gather = nullptr;  // every rank's solid-hash list, rank after rank
    uint64_t gather_cap = 0;
    uint64_t *d_counts = nullptr;  // world u64
    int *d_wide = nullptr;         // int32 slice of the dense reduction
    uint64_t wide_cap = 0;
    uint64_t *d_zero_tab = nullptr; // B1 + 1 zeros: the level-1 offsets of a rank that counted nothing
    uint64_t zero_tab_cap = 0;
    uint64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t rounds_a2a = 0, rounds_gather = 0;
    // how the current call failed, if it did: a failure every rank KNOWS about (it went through a status exchange) lets
    // all of them return together; one that only this rank saw -- a HIP call or a collective that failed between two
    // exchanges -- would leave the peers waiting in their next collective for ever, so the communicator is aborted
    // (ncclCommAbort: the peers' pending and later collectives return an error) and is dead from then on
    bool agreed_failure = false, dead = false;
    std::mutex mu;
};

namespace {

int grow(void **p, uint64_t *cap, uint64_t need, const char *what)
{
    if (*p && need <= *cap)
        return BRX_OK;
    if (*p)
        (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const uint64_t want = need + need / 16 + 256;
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {
        set_error("hipMalloc(%llu B, %s): %s", (unsigned long long)want, what, hipGetErrorString(e));
        return BRX_ERR_NOMEM;
    }
    *cap = want;
    return BRX_OK;
}

// every rank's `send` (send_counts[r] elements for rank r, back to back) to its owner; `recv` likewise.
// One group of ncclSend / ncclRecv per round, messages capped at a2a_chunk_elems().  A failing call never leaves
// the thread's group open: the first error is kept, the group is closed, then the error is returned.
int all_to_all_v(brx_comm *cm, const uint32_t *send, const uint64_t *send_counts, uint32_t *recv, const uint64_t *recv_counts,
                 uint64_t job_biggest, hipStream_t s)
{
    // tests on one GPU: route the own share through ncclSend/ncclRecv to self instead of a device copy
    const char *ess = getenv("BRX_EXCHANGE_SELF_SEND");
    const bool self_send = ess && *ess == '1';
    const int W = cm->world;
    std::vector<uint64_t> so(W + 1, 0), ro(W + 1, 0);
    for (int r = 0; r < W; r++) {
        so[r + 1] = so[r] + send_counts[r];
        ro[r + 1] = ro[r] + recv_counts[r];
    }
    // every rank must run the same number of rounds: the largest message of the JOB decides (every rank holds every
    // offset table, so all of them compute the same `job_biggest`)
    const uint64_t chunk = a2a_chunk_elems() ? a2a_chunk_elems() : (job_biggest ? job_biggest : 1);
    const uint64_t rounds = (job_biggest + chunk - 1) / chunk;
    cm->rounds_a2a = rounds;
    for (uint64_t c = 0; c < rounds; c++) {
        ncclResult_t first = ncclSuccess;
        hipError_t hfirst = hipSuccess;
        BRX_NCCL(g_rccl.GroupStart());
        for (int r = 0; r < W && first == ncclSuccess && hfirst == hipSuccess; r++) {
            const uint64_t slo = std::min(c * chunk, send_counts[r]), shi = std::min((c + 1) * chunk, send_counts[r]);
            const uint64_t rlo = std::min(c * chunk, recv_counts[r]), rhi = std::min((c + 1) * chunk, recv_counts[r]);
            if (r == cm->rank && !self_send) { // own share: a device copy, not a message
                if (shi > slo)
                    hfirst = hipMemcpyAsync(recv + ro[r] + rlo, send + so[r] + slo, (shi - slo) * 4, hipMemcpyDeviceToDevice, s);
                continue;
            }
            if (shi > slo)
                first = g_rccl.Send(send + so[r] + slo, shi - slo, ncclUint32, r, cm->comm, s);
            if (rhi > rlo && first == ncclSuccess)
                first = g_rccl.Recv(recv + ro[r] + rlo, rhi - rlo, ncclUint32, r, cm->comm, s);
        }
        const ncclResult_t ge = g_rccl.GroupEnd();
        if (hfirst != hipSuccess) {
            set_error("exchange: device copy of the own share: %s", hipGetErrorString(hfirst));
            return BRX_ERR_HIP;
        }
        BRX_NCCL(first);
        BRX_NCCL(ge);
    }
    return BRX_OK;
}

// Every rank reports the status of the LOCAL work it did since the last collective; all of them get the same verdict
// (one in-place all-gather of a word), so that a rank whose allocation failed, or which holds a counter the call
// cannot use, does not leave its peers waiting in the next collective: either all go on or all return.
int agree(brx_comm *cm, int local, hipStream_t s)
{
    const int W = cm->world, me = cm->rank;
    if (W == 1) {
        cm->agreed_failure = local != BRX_OK;
        return local;
    }
    const uint64_t mine = (uint64_t)(int64_t)local;
    std::vector<uint64_t> all(W, 0);
    hipError_t e = hipMemcpy(cm->d_counts + me, &mine, 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const ncclResult_t r = g_rccl.AllGather(cm->d_counts + me, cm->d_counts, 1, ncclUint64, cm->comm, s);
        if (r != ncclSuccess) {
            set_error("exchange: status all-gather: %s", g_rccl.GetErrorString(r));
            return local != BRX_OK ? local : BRX_ERR_HIP;
        }
        e = hipMemcpyAsync(all.data(), cm->d_counts, (size_t)W * 8, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        set_error("exchange: status exchange: %s", hipGetErrorString(e));
        return local != BRX_OK ? local : BRX_ERR_HIP;
    }
    if (local != BRX_OK) {
        cm->agreed_failure = true; // (every peer has read this rank's status and returns too)
        return local; // (its message is already set)
    }
    for (int r = 0; r < W; r++)
        if ((int64_t)all[r] != BRX_OK) {
            set_error("exchange: rank %d failed (%s); rank %d leaves the exchange with it", r, brx_strerror((int)(int64_t)all[r]), me);
            cm->agreed_failure = true;
            return (int)(int64_t)all[r];
        }
    return BRX_OK;
}

} // namespace

extern "C" {

int brx_comm_unique_id(uint8_t *id128)
{
    if (!id128)
        return BRX_ERR_ARG;
    BRX_TRY(rccl_load());
    static_assert(sizeof(ncclUniqueId) == BRX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    BRX_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return BRX_OK;
}

int brx_comm_init(const uint8_t *id128, int world, int rank, int device, brx_comm_t **out)
{
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) {
        set_error("comm_init: bad argument (world %d, rank %d)", world, rank);
        return BRX_ERR_ARG;
    }
    BRX_TRY(rccl_load());
    BRX_TRY(use_device(device));
    brx_comm *cm = new brx_comm();
    cm->world = world;
    cm->rank = rank;
    cm->device = device;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&cm->comm, world, id, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(world %d, rank %d, device %d): %s", world, rank, device, g_rccl.GetErrorString(r));
        delete cm;
        return BRX_ERR_HIP;
    }
    hipError_t e = hipMalloc((void **)&cm->d_counts, (size_t)world * 8 * 2);
    if (e != hipSuccess) {
        set_error("comm alloc: %s", hipGetErrorString(e));
        brx_comm_free(cm);
        return BRX_ERR_NOMEM;
    }
    *out = cm;
    return BRX_OK;
}

int brx_comm_init_all(int n_devices, const int *devices, brx_comm_t **out)
{
    if (n_devices < 1 || !devices || !out)
        return BRX_ERR_ARG;
    uint8_t id[BRX_COMM_ID_BYTES];
    BRX_TRY(brx_comm_unique_id(id));
    // ncclCommInitRank blocks until every rank has joined: one host thread per device
    std::vector<int> st(n_devices, BRX_OK);
    std::vector<std::string> msg(n_devices);
    std::vector<std::thread> th;
    for (int r = 0; r < n_devices; r++) {
        out[r] = nullptr;
        th.emplace_back([&, r] {
            st[r] = brx_comm_init(id, n_devices, r, devices[r], &out[r]);
            if (st[r] != BRX_OK)
                msg[r] = brx_last_error();
        });
    }
    for (auto &t : th)
        t.join();
    for (int r = 0; r < n_devices; r++)
        if (st[r] != BRX_OK) {
            set_error("comm_init_all: rank %d (device %d): %s", r, devices[r], msg[r].c_str());
            for (int q = 0; q < n_devices; q++) {
                brx_comm_free(out[q]);
                out[q] = nullptr;
            }
            return st[r];
        }
    return BRX_OK;
}

int brx_comm_info(const brx_comm_t *cm, int *world, int *rank, int *device)
{
    if (!cm)
        return BRX_ERR_ARG;
    if (world)
        *world = cm->world;
    if (rank)
        *rank = cm->rank;
    if (device)
        *device = cm->device;
    return BRX_OK;
}

int brx_comm_last_stats(const brx_comm_t *cm, uint64_t *stats8)
{
    if (!cm || !stats8)
        return BRX_ERR_ARG;
    memcpy(stats8, cm->stats, sizeof(cm->stats));
    return BRX_OK;
}

void brx_comm_free(brx_comm_t *cm)
{
    if (!cm)
        return;
    if (use_device(cm->device) == BRX_OK) {
        for (void *p : {(void *)cm->d_recv, (void *)cm->d_tables, (void *)cm->d_gather, (void *)cm->d_counts, (void *)cm->d_wide,
                        (void *)cm->d_zero_tab})
            if (p)
                (void)hipFree(p);
    }
    if (cm->comm && g_rccl.h)
        (void)g_rccl.CommDestroy(cm->comm);
    delete cm;
}

// The host-side arithmetic of the key exchange, a pure function of the gathered level-1 offset tables -- exported
// so that it can be checked for any world size without a GPU (tests/test_boundary.py).  Rank r owns the first digits
// [bound[r], bound[r+1]) = [r*B/W, (r+1)*B/W); rank `rank` sends its keys of owner r's range to r and receives from
// every rank s the keys of its own range; seg[s] is s's offset table clamped to the owned range and rebased to the
// start of the segment received from s (flat outside the range), which is what the finish of borrowed segments reads.
int brx_exchange_plan(const uint64_t *tables, int world, uint32_t n_buckets, int rank, uint32_t *bound, uint64_t *send_counts,
                      uint64_t *recv_counts, uint64_t *seg, uint64_t *largest_message)
{
    if (!tables || world < 1 || rank < 0 || rank >= world || n_buckets == 0 || !bound || !send_counts || !recv_counts) {
        set_error("exchange_plan: bad argument (world %d, rank %d, %u buckets)", world, rank, n_buckets);
        return BRX_ERR_ARG;
    }
    const size_t T = (size_t)n_buckets + 1;
    for (int r = 0; r < world; r++) {
        const uint64_t *t = tables + (size_t)r * T;
        if (t[0] != 0) {
            set_error("exchange_plan: offset table of rank %d does not start at 0", r);
            return BRX_ERR_ARG;
        }
        for (size_t b = 1; b < T; b++)
            if (t[b] < t[b - 1]) {
                set_error("exchange_plan: offset table of rank %d decreases at bucket %zu", r, b);
                return BRX_ERR_ARG;
            }
    }
    for (int r = 0; r <= world; r++)
        bound[r] = (uint32_t)((uint64_t)r * n_buckets / (uint64_t)world);
    const uint64_t *mine = tables + (size_t)rank * T;
    uint64_t biggest = 0;
    for (int r = 0; r < world; r++) {
        send_counts[r] = mine[bound[r + 1]] - mine[bound[r]];
        const uint64_t *tr = tables + (size_t)r * T;
        const uint64_t lo = tr[bound[rank]], hi = tr[bound[rank + 1]];
        recv_counts[r] = hi - lo;
        if (seg)
            for (size_t b = 0; b < T; b++)
                seg[(size_t)r * T + b] = (tr[b] < lo ? lo : (tr[b] > hi ? hi : tr[b])) - lo;
    }
    for (int a = 0; a < world; a++) { // the largest message anywhere in the job: every rank computes the same value
        const uint64_t *ta = tables + (size_t)a * T;
        for (int b = 0; b < world; b++)
            biggest = std::max(biggest, ta[bound[b + 1]] - ta[bound[b]]);
    }
    if (largest_message)
        *largest_message = biggest;
    return BRX_OK;
}

} // extern "C"

namespace {

// the owners' solid-hash lists to everybody: every rank's list lands at off[r] of d_gather (send/recv groups, capped
// messages; the group is always closed before an error is returned)
int gather_lists(brx_comm *cm, const std::vector<uint64_t> &n_of, const std::vector<uint64_t> &off, hipStream_t s)
{
    const int W = cm->world, me = cm->rank;
    const uint64_t n_mine = n_of[me];
    uint64_t chunk = a2a_chunk_elems() ? a2a_chunk_elems() / 2 : ~0ull; // u64 elements
    if (chunk == 0)
        chunk = 1;
    uint64_t maxn = 0;
    for (int r = 0; r < W; r++)
        maxn = std::max(maxn, n_of[r]);
    const uint64_t rounds = chunk == ~0ull ? (maxn ? 1 : 0) : (maxn + chunk - 1) / chunk;
    cm->rounds_gather = rounds;
    for (uint64_t q = 0; q < rounds; q++) {
        ncclResult_t first = ncclSuccess;
        BRX_NCCL(g_rccl.GroupStart());
        for (int r = 0; r < W && first == ncclSuccess; r++) {
            if (r == me)
                continue;
            const uint64_t slo = chunk == ~0ull ? 0 : std::min(q * chunk, n_mine), shi = chunk == ~0ull ? n_mine : std::min((q + 1) * chunk, n_mine);
            const uint64_t rlo = chunk == ~0ull ? 0 : std::min(q * chunk, n_of[r]), rhi = chunk == ~0ull ? n_of[r] : std::min((q + 1) * chunk, n_of[r]);
            if (shi > slo)
                first = g_rccl.Send(cm->d_gather + off[me] + slo, shi - slo, ncclUint64, r, cm->comm, s);
            if (rhi > rlo && first == ncclSuccess)
                first = g_rccl.Recv(cm->d_gather + off[r] + rlo, rhi - rlo, ncclUint64, r, cm->comm, s);
        }
        const ncclResult_t ge = g_rccl.GroupEnd();
        BRX_NCCL(first);
        BRX_NCCL(ge);
    }
    return BRX_OK;
}

// a failure only this rank saw: the peers must not be left in their next collective (see brx_comm::agreed_failure)
void abort_after_local_failure(brx_comm *cm)
{
    if (cm->world == 1 || cm->agreed_failure || cm->dead)
        return;
    cm->dead = true;
    if (cm->comm && g_rccl.CommAbort) {
        (void)g_rccl.CommAbort(cm->comm); // (frees the communicator)
        cm->comm = nullptr;
    }
}

struct ExchangeJob {
    uint64_t *d_extracted = nullptr; // owned: freed by the caller whatever happens
    bool counter_borrows = false;    // the counter refers to cm->d_recv: reset it before returning
};

// steps 4a: the owner's finish of the segments it received (all local work, one status)
int finish_owned(brx_comm *cm, brx_counter_t *c, uint8_t abundance, brx_set_t *dst, const std::vector<uint64_t> &seg,
                 const uint64_t *recv_counts, uint32_t T, ExchangeJob &job, void **d_list, uint64_t *n_mine, hipStream_t s)
{
    const int W = cm->world;
    BRX_TRY(brx_counter_reset(c, s));
    uint64_t *d_seg = cm->d_tables + (size_t)W * T;
    BRX_HIP(hipMemcpyAsync(d_seg, seg.data(), (size_t)W * T * 8, hipMemcpyHostToDevice, s));
    BRX_HIP(hipStreamSynchronize(s));
    uint64_t pos = 0;
    job.counter_borrows = true;
    for (int r = 0; r < W; r++) {
        if (recv_counts[r])
            BRX_TRY(brx_counter_add_partitioned_device(c, cm->d_recv + pos, d_seg + (size_t)r * T, recv_counts[r]));
        pos += recv_counts[r];
    }
    BRX_TRY(brx_set_count_finish_into(c, abundance, s, dst));
    // The partitioned finish lists the solid hashes on the side (only owned buckets were counted, so the list IS this
    // rank's share of the set).
    BRX_TRY(brx_set_keylist_device(dst, d_list, n_mine, s));
    if (!*d_list) {
        // no list (it did not fit, or BRX_LAZY_BITS=0 builds that keep none): take it from the bit vector
        if (dst->sparse || dst->bits_stale) {
            set_error("exchange: the owner's set has neither a key list nor a bit vector");
            return BRX_ERR_NOMEM;
        }
        BRX_HIP(hipStreamSynchronize(s));
        uint64_t pc = 0;
        BRX_TRY(brx_set_popcount(dst, &pc));
        BRX_HIP(hipMalloc((void **)&job.d_extracted, (pc + 64) * 8));
        BRX_TRY(brx_set_extract_keys_device(dst, 0, dst->nwords * 32, job.d_extracted, pc + 64, n_mine, s));
        *d_list = job.d_extracted;
    }
    BRX_HIP(hipStreamSynchronize(s));
    return BRX_OK;
}

int exchange_body(brx_comm *cm, brx_counter_t *c, uint8_t abundance, brx_set_t *dst, hipStream_t s, ExchangeJob &job)
{
    const int W = cm->world, me = cm->rank;
    const double t_begin = now_ms();

    // 1. this rank's level-1 layout: keys grouped by first digit + the B1+1 bucket offsets.  A rank that counted
    //    nothing (an empty shard) still joins every collective, with an all-zero offset table.
    void *pk = nullptr, *po = nullptr;
    uint32_t B1 = 0;
    uint64_t nk = 0;
    int local = brx_counter_l1_view(c, &pk, &po, &B1, &nk);
    const uint32_t T = B1 + 1;
    if (local == BRX_OK && !po) {
        local = grow((void **)&cm->d_zero_tab, &cm->zero_tab_cap, (uint64_t)T * 8, "empty offset table");
        if (local == BRX_OK && hipMemsetAsync(cm->d_zero_tab, 0, (uint64_t)T * 8, s) != hipSuccess)
            local = BRX_ERR_HIP;
        po = cm->d_zero_tab;
        nk = 0;
    }
    if (local == BRX_OK)
        local = grow((void **)&cm->d_tables, &cm->tables_cap, (uint64_t)W * T * 8 * 2, "offset tables");
    BRX_TRY(agree(cm, local, s));

    // 2. everybody's offset table to everybody (small), then to the host: all message sizes follow from them
    BRX_NCCL(g_rccl.AllGather(po, cm->d_tables, T, ncclUint64, cm->comm, s));
    std::vector<uint64_t> tab((size_t)W * T), seg((size_t)W * T), send_counts(W), recv_counts(W);
    std::vector<uint32_t> bound(W + 1);
    BRX_HIP(hipMemcpyAsync(tab.data(), cm->d_tables, (size_t)W * T * 8, hipMemcpyDeviceToHost, s));
    BRX_HIP(hipStreamSynchronize(s));
    uint64_t n_recv = 0, biggest = 0;
    local = brx_exchange_plan(tab.data(), W, B1, me, bound.data(), send_counts.data(), recv_counts.data(), seg.data(), &biggest);
    for (int r = 0; r < W; r++)
        n_recv += recv_counts[r];
    cm->stats[7] = biggest;

    // 3. keys to their owners
    if (local == BRX_OK)
        local = grow((void **)&cm->d_recv, &cm->recv_cap, (n_recv + 1) * 4, "received keys");
    BRX_TRY(agree(cm, local, s));
    const double t_a2a = now_ms();
    BRX_TRY(all_to_all_v(cm, (const uint32_t *)pk, send_counts.data(), cm->d_recv, recv_counts.data(), biggest, s));
    BRX_HIP(hipStreamSynchronize(s)); // the local level-1 buffer has been read: the counter may forget it now
    const double t_a2a_done = now_ms();

    // 4. the owner finishes its digit range: one borrowed segment per source rank, each with a full-size offset
    //    table that is flat outside the owned range
    void *d_list = nullptr;
    uint64_t n_mine = 0;
    local = finish_owned(cm, c, abundance, dst, seg, recv_counts.data(), T, job, &d_list, &n_mine, s);

    // 5. the owners' solid hashes to everybody; the size exchange carries the status of step 4
    uint64_t pair[2] = {n_mine, (uint64_t)(int64_t)local};
    std::vector<uint64_t> got((size_t)W * 2);
    BRX_HIP(hipMemcpy(cm->d_counts + 2 * me, pair, 16, hipMemcpyHostToDevice)); // (a stack variable: not an async copy)
    BRX_NCCL(g_rccl.AllGather(cm->d_counts + 2 * me, cm->d_counts, 2, ncclUint64, cm->comm, s)); // in place
    BRX_HIP(hipMemcpyAsync(got.data(), cm->d_counts, (size_t)W * 16, hipMemcpyDeviceToHost, s));
    BRX_HIP(hipStreamSynchronize(s));
    if (local != BRX_OK) {
        cm->agreed_failure = true; // (the size exchange carried it to everybody)
        return local;
    }
    std::vector<uint64_t> n_of(W), off(W + 1, 0);
    for (int r = 0; r < W; r++) {
        if ((int64_t)got[2 * r + 1] != BRX_OK) {
            set_error("exchange: rank %d failed to finish its digit range (%s)", r, brx_strerror((int)(int64_t)got[2 * r + 1]));
            cm->agreed_failure = true;
            return (int)(int64_t)got[2 * r + 1];
        }
        n_of[r] = got[2 * r];
        off[r + 1] = off[r] + n_of[r];
    }
    const uint64_t n_all = off[W];
    local = grow((void **)&cm->d_gather, &cm->gather_cap, (n_all + 1) * 8, "gathered solid k-mers");
    BRX_TRY(agree(cm, local, s));
    if (n_mine)
        BRX_HIP(hipMemcpyAsync(cm->d_gather + off[me], d_list, n_mine * 8, hipMemcpyDeviceToDevice, s));
    if (W > 1)
        BRX_TRY(gather_lists(cm, n_of, off, s));

    // 6. the whole job's set on this rank: OR the lists into a current bit vector, and hand the complete list to
    //    the probe index (for lazy and sparse sets the chained index built from it IS the set).  No collective
    //    follows, so a failure here is this rank's alone.
    if (!dst->sparse && !dst->bits_stale && W > 1) {
        if (off[me])
            BRX_TRY(brx_set_or_keys_device(dst, cm->d_gather, off[me], s));
        if (n_all > off[me + 1])
            BRX_TRY(brx_set_or_keys_device(dst, cm->d_gather + off[me + 1], n_all - off[me + 1], s));
    }
    if (index_wanted(dst->k) || no_bits(dst))
        BRX_TRY(brx_set_index_build_from_keys_device(dst, cm->d_gather, n_all, 0, 0, s));
    BRX_HIP(hipStreamSynchronize(s));
    cm->stats[0] = (nk - send_counts[me]) * 4;             // key bytes sent over the links
    cm->stats[1] = (n_recv - recv_counts[me]) * 4;         // key bytes received
    cm->stats[2] = n_recv;                                 // keys this owner counted
    cm->stats[3] = n_mine;                                 // solid k-mers of the owned range
    cm->stats[4] = n_all;                                  // solid k-mers of the job
    cm->stats[5] = (uint64_t)((t_a2a_done - t_a2a) * 1e3); // all-to-all, microseconds of host wall time
    cm->stats[6] = (uint64_t)((now_ms() - t_begin) * 1e3); // whole exchange + finish
    return BRX_OK;
}

} // namespace

extern "C" {

int brx_exchange_build_partitioned(brx_comm_t *cm, brx_counter_t *c, uint8_t abundance, brx_set_t *dst, void *stream)
{
    if (!cm || !c || !dst) {
        set_error("null argument");
        return BRX_ERR_ARG;
    }
    // (argument faults are this rank's alone, yet the peers must not be left waiting: they go into the first agreement)
    int local = BRX_OK;
    if (c->device != cm->device || dst->device != cm->device || dst->k != c->k) {
        set_error("exchange: communicator on device %d, counter on %d (k=%d), set on %d (k=%d)", cm->device, c->device, c->k,
                  dst->device, dst->k);
        local = BRX_ERR_ARG;
    }
    BRX_TRY(use_device(cm->device));
    std::lock_guard<std::mutex> g(cm->mu);
    if (cm->dead) {
        set_error("exchange: this communicator was aborted after a failure on this rank; make a new one");
        return BRX_ERR_UNSUPPORTED;
    }
    cm->agreed_failure = false;
    hipStream_t s = (hipStream_t)stream;
    if (local != BRX_OK)
        return agree(cm, local, s);
    ExchangeJob job;
    const int rc = exchange_body(cm, c, abundance, dst, s, job);
    if (rc != BRX_OK)
        abort_after_local_failure(cm);
    // ONE way out: whatever happened, the counter is empty again and no longer refers to the receive buffer
    if (job.d_extracted)
        (void)hipFree(job.d_extracted);
    const std::string msg = rc != BRX_OK ? brx_last_error() : "";
    const int rst = brx_counter_reset(c, s);
    if (rc != BRX_OK) {
        set_error("%s", msg.c_str());
        return rc;
    }
    return rst;
}

static int reduce_counts_body(brx_comm_t *cm, brx_counter_t *c, uint8_t abundance, hipStream_t s);

int brx_exchange_reduce_counts(brx_comm_t *cm, brx_counter_t *c, uint8_t abundance, void *stream)
{
    if (!cm || !c)
        return BRX_ERR_ARG;
    BRX_TRY(use_device(cm->device));
    std::lock_guard<std::mutex> g(cm->mu);
    if (cm->dead) {
        set_error("exchange: this communicator was aborted after a failure on this rank; make a new one");
        return BRX_ERR_UNSUPPORTED;
    }
    cm->agreed_failure = false;
    const int rc = reduce_counts_body(cm, c, abundance, (hipStream_t)stream);
    if (rc != BRX_OK)
        abort_after_local_failure(cm);
    return rc;
}

static int reduce_counts_body(brx_comm_t *cm, brx_counter_t *c, uint8_t abundance, hipStream_t s)
{
    void *d_counts = nullptr;
    uint64_t nbytes = 0;
    int local = BRX_OK;
    if (c->device != cm->device) {
        set_error("exchange: communicator on device %d, counter on %d", cm->device, c->device);
        local = BRX_ERR_ARG;
    }
    if (local == BRX_OK)
        local = brx_counter_device_counts(c, &d_counts, &nbytes);
    // BRX_EXCHANGE_FORCE_WIDE=1 (test hook): take the int32 path whatever world*(a+1) is, also with a world of one
    const char *efw = getenv("BRX_EXCHANGE_FORCE_WIDE");
    const bool force_wide = efw && *efw == '1';
    if (cm->world == 1 && !force_wide)
        return local;
    const uint32_t cap = (uint32_t)abundance + 1u;
    const bool exact_u8 = (uint64_t)cm->world * cap <= 255u && !force_wide;
    // sum_r min(c_r, a+1) > a  <=>  sum_r c_r > a, and the clamped sum cannot wrap an u8
    if (local == BRX_OK)
        local = brx_counter_clamp(c, exact_u8 ? (uint8_t)cap : 255, s);
    const uint64_t slice = 1ull << 30;
    uint8_t *p = (uint8_t *)d_counts;
    if (local == BRX_OK && !exact_u8)
        local = grow((void **)&cm->d_wide, &cm->wide_cap, std::min(slice, nbytes) * 4, "int32 reduction slice");
    BRX_TRY(agree(cm, local, s));
    for (uint64_t lo = 0; lo < nbytes; lo += slice) {
        const uint64_t n = std::min(slice, nbytes - lo);
        if (exact_u8) {
            BRX_NCCL(g_rccl.AllReduce(p + lo, p + lo, n, ncclUint8, ncclSum, cm->comm, s));
        } else {
            widen_u8_kernel<<<2048, 256, 0, s>>>(p + lo, cm->d_wide, n);
            BRX_NCCL(g_rccl.AllReduce(cm->d_wide, cm->d_wide, n, ncclInt32, ncclSum, cm->comm, s));
            narrow_sat_kernel<<<2048, 256, 0, s>>>(cm->d_wide, p + lo, n);
        }
    }
    BRX_HIP(hipStreamSynchronize(s));
    return BRX_OK;
}

} // extern "C"
