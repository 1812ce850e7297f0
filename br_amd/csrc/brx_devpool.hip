// Device memory of libbrx.so: large blocks are kept when they are freed and handed out again.
//
// A job file -> file creates a counter, a set and a chain, and drops them again: 4 GB of keys, the 2 GiB probe index, the
// staging buffers.  hipFree of such a block usually takes microseconds and now and then a third of a second (the driver
// unmaps it), hipMalloc pays for the page tables again -- measured with tools/e2e_repeat.py: one repetition in eight lost
// 0.7 s to two frees, and bench.py's `e2e` block showed build legs of 1.4 s instead of 0.13 s.  Every hipMalloc / hipFree
// of the library's translation units goes through dev_alloc / dev_free (the macros at the end of brx_internal.hpp);
// blocks of at least POOL_MIN bytes are parked here when freed, up to BRX_DEVPOOL_GB (default 48) GiB per process and only
// while at least an eighth of the card is free, and a
// request takes the smallest parked block that is large enough and at most twice its size.
//
// Same contract as the runtime's: a freed block may still be in use by queued kernels (hipFree waits for the device), so
// dev_free synchronises the device before the block can go to anybody else; fresh blocks hold whatever they held.  When
// the runtime is out of memory the pool is emptied and the request tried again.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <mutex>
#include <unordered_map>
#include <vector>

namespace brx {

namespace {
struct Parked {
    void *p;
    size_t bytes;
    int device;
};
std::mutex g_mu;
std::unordered_map<void *, size_t> g_live; // blocks handed out: their sizes
std::vector<Parked> g_parked;
size_t g_parked_bytes = 0;
constexpr size_t POOL_MIN = 32ull << 20;

// BRX_DEBUG_POISON=1 (read once): every block handed out -- fresh from the runtime or recycled from the pool -- is filled
// with 0xA5 first, so that a kernel that reads memory nobody wrote in ITS job meets the same garbage on every run and on
// every box instead of whatever the block's last user left (tools/fuzz_parity.py and the GPU suite are run under it).
bool poison_on()
{
    static const bool on = [] {
        const char *e = getenv("BRX_DEBUG_POISON");
        return e && *e && *e != '0';
    }();
    return on;
}

hipError_t poison(void *p, size_t bytes)
{
    if (!poison_on())
        return hipSuccess;
    hipError_t e = hipMemset(p, 0xA5, bytes);
    if (e == hipSuccess)
        e = hipDeviceSynchronize(); // (hipMemset on device memory does not wait; streams here are non-blocking)
    return e;
}

size_t pool_cap()
{
    static const size_t cap = [] {
        const char *e = getenv("BRX_DEVPOOL_GB");
        const double gb = e && *e ? atof(e) : 48.0;
        return gb <= 0 ? (size_t)0 : (size_t)(gb * (double)(1ull << 30));
    }();
    return cap;
}

void drop_all_locked(std::vector<void *> &out)
{
    for (const Parked &b : g_parked)
        out.push_back(b.p);
    g_parked.clear();
    g_parked_bytes = 0;
}
} // namespace

void dev_pool_trim();

hipError_t dev_alloc(void **out, size_t bytes)
{
    *out = nullptr;
    if (bytes == 0)
        bytes = 1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (bytes >= POOL_MIN) {
        std::lock_guard<std::mutex> g(g_mu);
        int best = -1;
        for (int i = 0; i < (int)g_parked.size(); i++)
            if (g_parked[i].device == dev && g_parked[i].bytes >= bytes && g_parked[i].bytes / 2 <= bytes &&
                (best < 0 || g_parked[i].bytes < g_parked[best].bytes))
                best = i;
        if (best >= 0) {
            const Parked b = g_parked[best];
            g_parked.erase(g_parked.begin() + best);
            g_parked_bytes -= b.bytes;
            g_live[b.p] = b.bytes;
            *out = b.p;
        }
    }
    if (*out)
        return poison(*out, bytes);
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> g(g_mu);
            drop_all_locked(drop);
        }
        for (void *p : drop)
            (void)hipFree(p);
        e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess && bytes >= POOL_MIN) {
        std::lock_guard<std::mutex> g(g_mu);
        g_live[*out] = bytes;
    }
    if (e == hipSuccess)
        e = poison(*out, bytes);
    return e;
}

hipError_t dev_free(void *p)
{
    if (!p)
        return hipSuccess;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto it = g_live.find(p);
        if (it != g_live.end()) {
            bytes = it->second;
            g_live.erase(it);
        }
    }
    if (bytes == 0 || pool_cap() == 0)
        return hipFree(p); // small, or not one of the pool's
    // what hipFree would have waited for: kernels that still use the block -- on the device the BLOCK lives on, which
    // need not be the calling thread's current one (brx_comm_init_all's per-device threads)
    int cur = 0, dev = 0;
    (void)hipGetDevice(&cur);
    dev = cur;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) == hipSuccess)
        dev = at.device;
    else
        (void)hipGetLastError();
    if (dev != cur)
        (void)hipSetDevice(dev);
    const hipError_t se = hipDeviceSynchronize();
    if (dev != cur)
        (void)hipSetDevice(cur);
    if (se != hipSuccess)
        return hipFree(p);
    // The pool's memory is invisible to every other allocator on the card (torch's, RCCL's, another rank's pool when
    // several processes share a device): when the card runs short -- less than an eighth of it free -- nothing is parked
    // and what is parked goes back to the runtime.
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot != 0 && fr < tot / 8) {
            dev_pool_trim();
            return hipFree(p);
        }
        (void)hipGetLastError();
    }
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> g(g_mu);
        // make room: the oldest parked blocks go first
        while (!g_parked.empty() && g_parked_bytes + bytes > pool_cap()) {
            drop.push_back(g_parked.front().p);
            g_parked_bytes -= g_parked.front().bytes;
            g_parked.erase(g_parked.begin());
        }
        if (bytes <= pool_cap()) {
            g_parked.push_back(Parked{p, bytes, dev});
            g_parked_bytes += bytes;
            p = nullptr;
        }
    }
    for (void *q : drop)
        (void)hipFree(q);
    return p ? hipFree(p) : hipSuccess;
}

// everything parked goes back to the runtime (brx_devpool_trim: a host that wants the memory for something else)
void dev_pool_trim()
{
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> g(g_mu);
        drop_all_locked(drop);
    }
    for (void *p : drop)
        (void)hipFree(p);
}

size_t dev_pool_bytes()
{
    std::lock_guard<std::mutex> g(g_mu);
    return g_parked_bytes;
}

} // namespace brx
