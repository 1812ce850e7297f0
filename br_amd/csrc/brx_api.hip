// Diagnostics, device selection and the per-kernel event timers of libbrx.so.
#include "brx_internal.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <string.h>
#include <map>

namespace brx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int use_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no usable HIP device (hipGetDeviceCount: %s); libbrx has no CPU fallback",
                  e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return BRX_ERR_NODEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (0..%d)", device, n - 1);
        return BRX_ERR_ARG;
    }
    BRX_HIP(hipSetDevice(device));
    return BRX_OK;
}

void trace_stage(hipStream_t s, const char *what)
{
    static const bool on = [] { const char *e = getenv("BRX_TRACE"); return e && *e && *e != '0'; }();
    if (!on)
        return;
    static const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipStreamSynchronize(s);
    fprintf(stderr, "[brx %9.3f s] %s%s%s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), what,
            e == hipSuccess ? "" : " -> ", e == hipSuccess ? "" : hipGetErrorString(e));
    fflush(stderr);
}

// ---- timers ---------------------------------------------------------------------------------
struct TimerSlot {
    std::string name;
    double total_ms = 0;
    uint64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

static std::mutex g_tmu;
static bool g_prof_on = false;
static std::vector<TimerSlot> g_slots;
static std::vector<hipEvent_t> g_pool;

static int slot_of(const char *name)
{
    for (size_t i = 0; i < g_slots.size(); i++)
        if (g_slots[i].name == name)
            return (int)i;
    g_slots.emplace_back();
    g_slots.back().name = name;
    return (int)g_slots.size() - 1;
}

static hipEvent_t get_event()
{
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess)
        return nullptr;
    return e;
}

static void resolve(TimerSlot &s)
{
    for (auto &p : s.pending) {
        float ms = 0;
        if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
            s.total_ms += ms;
            s.launches++;
        }
        g_pool.push_back(p.first);
        g_pool.push_back(p.second);
    }
    s.pending.clear();
}

KernelTimer::KernelTimer(const char *name, hipStream_t s) : slot_(-1), s_(s), start_(nullptr), stop_(nullptr), on_(false)
{
    std::lock_guard<std::mutex> g(g_tmu);
    if (!g_prof_on)
        return;
    slot_ = slot_of(name);
    start_ = get_event();
    stop_ = get_event();
    if (!start_ || !stop_)
        return;
    on_ = hipEventRecord(start_, s_) == hipSuccess;
}

KernelTimer::~KernelTimer()
{
    if (!on_)
        return;
    (void)hipEventRecord(stop_, s_);
    std::lock_guard<std::mutex> g(g_tmu);
    g_slots[slot_].pending.emplace_back(start_, stop_);
}

// ---- page-locked host buffers: what the host-buffer entry points hand back, and what callers may fill ---------------
// A copy between HBM and PAGEABLE host memory goes through the runtime's own bounce buffers at 6-10 GB/s: 82 MB each
// way per 8192-record batch was 2/3 of brx_chain_correct_batch's 31 ms (tools/host_rate.py, round 2).  Page-locked memory
// moves at PCIe speed and asynchronously; hipHostMalloc itself costs milliseconds, so the blocks are pooled: a block
// released by brx_buf_free / brx_host_free is kept (up to BRX_HOSTPOOL_GB) for the next batch.
namespace {
struct HostBlock {
    void *p;
    size_t cap;
    bool used;
};
std::mutex g_hostpool_mu;
std::vector<HostBlock> g_hostpool;
// idle page-locked bytes kept for reuse (BRX_HOSTPOOL_GB, default 4): three chains in flight with 32768-record batches
// hold ~2 GiB of blocks between them -- at the first 1 GiB limit every batch paid a hipHostMalloc of its 328 MB output
// again (67 ms per batch instead of 25: profiles/r4h_bench_n1.json records_32768 before the change)
size_t pool_keep()
{
    static const size_t v = [] {
        const char *e = getenv("BRX_HOSTPOOL_GB");
        const double gb = e && *e ? atof(e) : 4.0;
        return gb <= 0 ? (size_t)0 : (size_t)(gb * (double)(1ull << 30));
    }();
    return v;
}
}

void *host_buf_acquire(size_t bytes)
{
    {
        std::lock_guard<std::mutex> g(g_hostpool_mu);
        int best = -1;
        for (int i = 0; i < (int)g_hostpool.size(); i++)
            if (!g_hostpool[i].used && g_hostpool[i].cap >= bytes && g_hostpool[i].cap <= 2 * bytes + (1u << 20) &&
                (best < 0 || g_hostpool[i].cap < g_hostpool[best].cap))
                best = i;
        if (best >= 0) {
            g_hostpool[best].used = true;
            return g_hostpool[best].p;
        }
    }
    void *p = nullptr;
    const size_t cap = bytes + bytes / 8 + 4096; // batches of a stream differ by a few percent: let the next one fit
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess || !p) {
        // no GPU runtime / no lockable memory left: plain memory still works, only slower
        p = malloc(bytes);
        return p;
    }
    std::lock_guard<std::mutex> g(g_hostpool_mu);
    g_hostpool.push_back(HostBlock{p, cap, true});
    return p;
}

void host_buf_release(void *p)
{
    if (!p)
        return;
    void *drop = nullptr;
    {
        std::lock_guard<std::mutex> g(g_hostpool_mu);
        size_t idle = 0;
        int at = -1;
        for (int i = 0; i < (int)g_hostpool.size(); i++) {
            if (g_hostpool[i].p == p)
                at = i;
            else if (!g_hostpool[i].used)
                idle += g_hostpool[i].cap;
        }
        if (at < 0) { // not one of ours: it came from malloc
            drop = nullptr;
        } else if (idle + g_hostpool[at].cap > pool_keep()) {
            drop = g_hostpool[at].p;
            g_hostpool.erase(g_hostpool.begin() + at);
            (void)hipHostFree(drop);
            return;
        } else {
            g_hostpool[at].used = false;
            return;
        }
    }
    free(p);
}

bool host_buf_is_pinned(const void *p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError(); // (an ordinary malloc'ed pointer: the query fails and leaves a sticky-looking error behind)
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

} // namespace brx

using namespace brx;

extern "C" {

const char *brx_strerror(int status)
{
    switch (status) {
    case BRX_OK: return "ok";
    case BRX_ERR_ARG: return "bad argument";
    case BRX_ERR_NOMEM: return "out of memory";
    case BRX_ERR_HIP: return "HIP runtime error";
    case BRX_ERR_NODEVICE: return "no usable GPU (no CPU fallback)";
    case BRX_ERR_FORMAT: return "malformed input";
    case BRX_ERR_UNSUPPORTED: return "not implemented";
    case BRX_ERR_OVERFLOW: return "output buffer too small";
    default: return "unknown status";
    }
}

const char *brx_last_error(void) { return g_err; }

int brx_version(void) { return 100; }

int brx_device_count(int *n)
{
    if (!n)
        return BRX_ERR_ARG;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *n = c;
    return BRX_OK;
}

int brx_profile_enable(int on)
{
    std::lock_guard<std::mutex> g(g_tmu);
    g_prof_on = on != 0;
    return BRX_OK;
}

int brx_profile_reset(void)
{
    std::lock_guard<std::mutex> g(g_tmu);
    for (auto &s : g_slots) {
        resolve(s);
        s.total_ms = 0;
        s.launches = 0;
    }
    return BRX_OK;
}

int brx_profile_get(const char *kernel, double *total_ms, uint64_t *launches)
{
    if (!kernel)
        return BRX_ERR_ARG;
    std::lock_guard<std::mutex> g(g_tmu);
    for (auto &s : g_slots)
        if (s.name == kernel) {
            resolve(s);
            if (total_ms)
                *total_ms = s.total_ms;
            if (launches)
                *launches = s.launches;
            return BRX_OK;
        }
    if (total_ms)
        *total_ms = 0;
    if (launches)
        *launches = 0;
    return BRX_OK;
}

int brx_profile_names(char *buf, size_t cap)
{
    if (!buf || cap == 0)
        return BRX_ERR_ARG;
    std::lock_guard<std::mutex> g(g_tmu);
    std::string s;
    for (auto &t : g_slots) {
        if (!s.empty())
            s += ",";
        s += t.name;
    }
    snprintf(buf, cap, "%s", s.c_str());
    return BRX_OK;
}

void brx_buf_free(void *p) { brx::host_buf_release(p); }

void *brx_host_alloc(size_t bytes) { return brx::host_buf_acquire(bytes ? bytes : 1); }

void brx_host_free(void *p) { brx::host_buf_release(p); }
void brx_devpool_trim(void) { brx::dev_pool_trim(); }
uint64_t brx_devpool_bytes(void) { return (uint64_t)brx::dev_pool_bytes(); }

} // extern "C"
