// Host pipeline of run_correction (src/lib.rs:22-139) and of count_fasta (src/main.rs:72-78) over file
// descriptors: FASTA parse -> pinned batch -> H2D -> kernels -> FASTA text made on the device -> D2H -> write, the
// three stages on their own threads so that parsing, the GPU and the writes of consecutive batches overlap.
//
// FASTA conventions = noodles::fasta::{Reader, Writer} as the reference uses them (src/lib.rs:30-31,57-60,
// 80-81,123-131; noodles-fasta 0.38, not vendored, UNPINNED by the reference's tests), the same rules as
// br_amd/fasta.py (tests compare the two byte for byte):
//   reader  '>' definition: name = up to the first ASCII whitespace, description = the rest, trimmed;
//           sequence = the following lines with their line ends (\r, \n) removed; a malformed record ends
//           the stream silently (`while let Some(Ok(record))`, src/lib.rs:35), records before it are kept.
//   writer  '>name[ description]\n', then the sequence wrapped at 80 columns.
// Batches are larger than the reference's 8192 records (reads are independent units and the output keeps
// the input order, so the batch size cannot change a byte); they are what fills the GPU.
#include "brx_internal.hpp"

#include <condition_variable>
#include <deque>
#include <thread>
#include <chrono>
#include <errno.h>
#include <string.h>
#include <unistd.h>
#include <poll.h>
#include <fcntl.h>
#include <sys/stat.h>

using namespace brx;

namespace {

constexpr size_t LINE_BASES = 80;          // noodles fasta::Writer default line_base_count
constexpr size_t READ_CHUNK = 8u << 20;    // bytes per read() call
// bases per batch.  Measured on 1 Gbp of 10 kb reads (tools/pipe_sweep.sh): 16-32 MB batches run the whole
// FASTA -> FASTA job 1.7x faster than 128 MB ones -- the kernels lose a little on 3 000-read batches, but the
// pinned and device buffers (allocated and first touched once per slot) are a quarter of the size and the writer
// starts after 5 ms of parsing instead of 30; below 16 MB the per-batch overheads take over.
inline uint64_t batch_bases()
{
    static const uint64_t v = [] {
        const char *e = getenv("BRX_PIPE_BATCH_MB");
        const long mb = e ? atol(e) : 0;
        return (uint64_t)(mb >= 1 && mb <= 4096 ? mb : 32) << 20;
    }();
    return v;
}
constexpr int N_SLOTS = 6;

inline bool is_ws(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// BRX_PIPE_TRACE=1: per-allocation and per-batch timings on stderr
inline bool pipe_trace()
{
    static const bool on = [] { const char *e = getenv("BRX_PIPE_TRACE"); return e && *e == '1'; }();
    return on;
}

// growable pinned host buffer (hipHostMalloc: H2D / D2H at link speed, async capable)
struct Pinned {
    uint8_t *p = nullptr;
    uint64_t cap = 0;
    int reserve(uint64_t need, uint64_t keep)
    {
        if (need <= cap)
            return BRX_OK;
        uint64_t want = cap ? cap : (1u << 20);
        while (want < need)
            want += want / 2 + (1u << 20);
        // from the library's pool of page-locked blocks (brx_api.hip): hipHostMalloc costs milliseconds per block, and a
        // job is two pipeline runs (count, then correct) -- or many, when a host calls them file after file -- that want the
        // same handful of sizes
        const double t0 = now_s();
        uint8_t *q = (uint8_t *)host_buf_acquire(want);
        if (pipe_trace())
            fprintf(stderr, "[brx pipe] page-locked block %.1f MB: %.2f ms\n", (double)want / 1e6, (now_s() - t0) * 1e3);
        if (!q) {
            set_error("page-locked host block of %llu B: out of memory", (unsigned long long)want);
            return BRX_ERR_NOMEM;
        }
        if (p) {
            if (keep)
                memcpy(q, p, keep);
            host_buf_release(p);
        }
        p = q;
        cap = want;
        return BRX_OK;
    }
    ~Pinned()
    {
        if (p)
            host_buf_release(p);
    }
};

struct Batch {
    uint64_t seq = 0;             // position in the stream (the writer emits in this order)
    Pinned bases;                 // concatenated sequences
    std::vector<uint64_t> offsets; // n + 1
    std::string defs;             // re-emitted definition lines, concatenated
    std::vector<uint32_t> def_end; // end offset of record r's definition in `defs`
    uint64_t total = 0;
    Pinned text;                  // the batch as FASTA text, formatted on the GPU and copied here by the worker that corrected it
    uint64_t text_len = 0;
    uint64_t out_total = 0;       // corrected bases of the batch
    bool last = false;
    void clear()
    {
        offsets.assign(1, 0);
        defs.clear();
        def_end.clear();
        total = 0;
        last = false;
    }
    uint32_t n() const { return (uint32_t)def_end.size(); }
};

template <class T>
class Queue {
  public:
    void push(T v)
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            q_.push_back(v);
        }
        cv_.notify_all();
    }
    // false when the queue was closed and is empty
    bool pop(T &v)
    {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return !q_.empty() || closed_; });
        if (q_.empty())
            return false;
        v = q_.front();
        q_.pop_front();
        return true;
    }
    void close()
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            closed_ = true;
        }
        cv_.notify_all();
    }

  private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<T> q_;
    bool closed_ = false;
};

// ---- FASTA reader: whole lines out of blocks that a read-ahead thread fills ---------------------------
// read() (a kernel copy of the whole stream) runs on its own thread, READ_CHUNK bytes at a time into a small pool of
// blocks; next() hands out lines in place -- only a line that straddles two blocks is assembled in `carry_`.
class LineReader {
  public:
    explicit LineReader(int fd) : fd_(fd)
    {
        for (auto &b : pool_)
            free_.push_back(&b);
        io_ = std::thread([this] { pump(); });
    }
    ~LineReader()
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        io_.join();
    }
    LineReader(const LineReader &) = delete;
    LineReader &operator=(const LineReader &) = delete;
    // next line without its trailing run of \r / \n (python's rstrip(b"\r\n")); false at end of input.
    // The line stays valid until the next call.
    bool next(const char *&line, size_t &len)
    {
        if (carry_served_) {
            carry_.clear();
            carry_served_ = false;
        }
        for (;;) {
            if (!cur_) {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return !full_.empty() || eof_; });
                if (full_.empty()) { // end of input: what is left in carry_ is the last, unterminated line
                    if (carry_.empty())
                        return false;
                    line = carry_.data();
                    len = carry_.size();
                    carry_served_ = true;
                    strip(line, len);
                    return true;
                }
                cur_ = full_.front();
                full_.pop_front();
                pos_ = 0;
            }
            const char *base = cur_->data.data();
            const char *nl = (const char *)memchr(base + pos_, '\n', cur_->len - pos_);
            if (nl) {
                if (carry_.empty()) {
                    line = base + pos_;
                    len = (size_t)(nl - line);
                } else {
                    carry_.append(base + pos_, (size_t)(nl - (base + pos_)));
                    line = carry_.data();
                    len = carry_.size();
                    carry_served_ = true;
                }
                pos_ = (size_t)(nl - base) + 1;
                strip(line, len);
                return true;
            }
            carry_.append(base + pos_, cur_->len - pos_); // no line end in the rest of this block
            {
                std::lock_guard<std::mutex> g(mu_);
                free_.push_back(cur_);
            }
            cv_.notify_all();
            cur_ = nullptr;
        }
    }
    int error() const { return err_; }

  private:
    struct Block {
        std::vector<char> data;
        size_t len = 0;
    };
    static void strip(const char *line, size_t &len)
    {
        while (len && (line[len - 1] == '\r' || line[len - 1] == '\n'))
            len--;
    }
    void pump()
    {
        for (;;) {
            Block *b = nullptr;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return !free_.empty() || stop_; });
                if (stop_)
                    break;
                b = free_.front();
                free_.pop_front();
            }
            if (b->data.size() < READ_CHUNK)
                b->data.resize(READ_CHUNK);
            // wait for data in slices, so that a reader that gave up on the stream (parse error) is not held by a
            // pipe whose writer is still alive
            struct pollfd pf = {fd_, POLLIN, 0};
            int pr = 0;
            while ((pr = ::poll(&pf, 1, 50)) == 0 || (pr < 0 && errno == EINTR)) {
                std::lock_guard<std::mutex> g(mu_);
                if (stop_)
                    break;
            }
            ssize_t r = 0;
            {
                std::lock_guard<std::mutex> g(mu_);
                if (stop_)
                    break;
            }
            do {
                r = ::read(fd_, b->data.data(), READ_CHUNK);
            } while (r < 0 && errno == EINTR);
            std::lock_guard<std::mutex> g(mu_);
            if (r <= 0) {
                if (r < 0)
                    err_ = errno;
                eof_ = true;
                cv_.notify_all();
                break;
            }
            b->len = (size_t)r;
            full_.push_back(b);
            cv_.notify_all();
        }
        std::lock_guard<std::mutex> g(mu_);
        eof_ = true;
        cv_.notify_all();
    }

    int fd_;
    Block pool_[3];
    std::deque<Block *> free_, full_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::thread io_;
    bool stop_ = false, eof_ = false;
    Block *cur_ = nullptr;
    size_t pos_ = 0;
    std::string carry_;
    bool carry_served_ = false;
    int err_ = 0;
};

// fills batches from the stream; one record is never split between batches
class FastaBatcher {
  public:
    FastaBatcher(int fd, uint32_t max_records) : lr_(fd), max_records_(max_records ? max_records : 16384u) {}
    // BRX_OK; b.last is set when the stream ended (a batch may be empty then)
    int fill(Batch &b)
    {
        b.clear();
        if (done_) {
            b.last = true;
            return BRX_OK;
        }
        const uint64_t target = batch_bases();
        for (;;) {
            if (have_def_) { // definition read while closing the previous record
                if (!open_record(b))
                    break;
                have_def_ = false;
            }
            const char *line;
            size_t len;
            if (!lr_.next(line, len)) {
                close_record(b);
                done_ = true;
                break;
            }
            if (len && line[0] == '>') {
                close_record(b);
                if (!parse_def(line + 1, len - 1)) { // missing name: the stream ends here (src/lib.rs:35)
                    done_ = true;
                    break;
                }
                have_def_ = true;
                if (b.n() >= max_records_ || b.total >= target)
                    break; // batch full: the pending definition opens the next one
                continue;
            }
            if (!in_record_) { // data before the first definition: parse error
                done_ = true;
                break;
            }
            if (b.total + len + 64 > b.bases.cap) { // one allocation per slot in the common case: the batch + one long read
                const uint64_t room = target + target / 4 + (1ull << 20);
                BRX_TRY(b.bases.reserve(b.total + len + 64 > room ? b.total + len + 64 : room, b.total));
            }
            memcpy(b.bases.p + b.total, line, len);
            b.total += len;
        }
        if (lr_.error()) {
            set_error("read: %s", strerror(lr_.error()));
            return BRX_ERR_ARG;
        }
        b.last = done_;
        return BRX_OK;
    }

  private:
    bool parse_def(const char *body, size_t len)
    {
        if (len == 0 || is_ws((unsigned char)body[0]))
            return false;
        size_t i = 0;
        while (i < len && !is_ws((unsigned char)body[i]))
            i++;
        def_.assign(body, i);
        size_t j = i;
        while (j < len && is_ws((unsigned char)body[j]))
            j++;
        size_t e = len;
        while (e > j && is_ws((unsigned char)body[e - 1]))
            e--;
        if (e > j) {
            def_.push_back(' ');
            def_.append(body + j, e - j);
        }
        return true;
    }
    bool open_record(Batch &b)
    {
        b.defs += def_;
        in_record_ = true;
        cur_open_ = true;
        return true;
    }
    void close_record(Batch &b)
    {
        if (cur_open_) {
            b.def_end.push_back((uint32_t)b.defs.size());
            b.offsets.push_back(b.total);
            cur_open_ = false;
        }
    }
    LineReader lr_;
    uint32_t max_records_;
    std::string def_;
    bool have_def_ = false, in_record_ = false, cur_open_ = false, done_ = false;
};

int write_all(int fd, const char *p, size_t n)
{
    while (n) {
        ssize_t w = ::write(fd, p, n);
        if (w < 0) {
            if (errno == EINTR)
                continue;
            set_error("write: %s", strerror(errno));
            return BRX_ERR_ARG;
        }
        p += w;
        n -= (size_t)w;
    }
    return BRX_OK;
}

// ---- the writer's text, made on the device ------------------------------------------------------------------------------
// noodles' fasta::Writer layout (see the top of the file): '>' definition '\n', then the sequence in lines of 80.  The
// host used to build it with one 80-byte memcpy per line (0.36 s per Gbp over two threads, more than the kernels of the
// batch); here a batch costs the host one prefix-sum read-back and the D2H copy it needed anyway, 1.25 % longer.
// text_off[r] = where record r starts in the text; text_off[n] = its length.  One block: every thread sums a
// contiguous run of records, the runs are scanned through LDS.
__global__ __launch_bounds__(1024) void text_offsets_kernel(const uint64_t *__restrict__ out_off, const uint32_t *__restrict__ def_end,
                                                            uint32_t n, uint64_t *__restrict__ text_off)
{
    __shared__ uint64_t run_sum[1024];
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t r0 = threadIdx.x * per, r1 = r0 + per < n ? r0 + per : n;
    auto size_of = [&](uint32_t r) -> uint64_t {
        const uint64_t len = out_off[r + 1] - out_off[r];
        const uint64_t dl = def_end[r] - (r ? def_end[r - 1] : 0u);
        return 2ull + dl + len + (len + LINE_BASES - 1) / LINE_BASES;
    };
    uint64_t mine = 0;
    for (uint32_t r = r0; r < r1; r++)
        mine += size_of(r);
    run_sum[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) { // inclusive scan
        const uint64_t v = threadIdx.x >= d ? run_sum[threadIdx.x - d] : 0ull;
        __syncthreads();
        run_sum[threadIdx.x] += v;
        __syncthreads();
    }
    uint64_t run = run_sum[threadIdx.x] - mine;
    for (uint32_t r = r0; r < r1; r++) {
        text_off[r] = run;
        run += size_of(r);
    }
    if (threadIdx.x == 1023)
        text_off[n] = run_sum[1023];
}

// one workgroup per record (grid-stride)
__global__ __launch_bounds__(256) void format_kernel(const uint8_t *__restrict__ seqs, const uint64_t *__restrict__ out_off,
                                                     const uint8_t *__restrict__ defs, const uint32_t *__restrict__ def_end,
                                                     const uint64_t *__restrict__ text_off, uint32_t n, uint8_t *__restrict__ text)
{
    for (uint32_t r = blockIdx.x; r < n; r += gridDim.x) {
        const uint32_t d0 = r ? def_end[r - 1] : 0u, dl = def_end[r] - d0;
        uint8_t *w = text + text_off[r];
        if (threadIdx.x == 0) {
            w[0] = '>';
            w[1 + dl] = '\n';
        }
        for (uint32_t i = threadIdx.x; i < dl; i += 256)
            w[1 + i] = defs[d0 + i];
        w += 2 + dl;
        const uint8_t *seq = seqs + out_off[r];
        const uint64_t len = out_off[r + 1] - out_off[r];
        const uint64_t tl = len + (len + LINE_BASES - 1) / LINE_BASES;
        for (uint64_t j = threadIdx.x; j < tl; j += 256) {
            const uint64_t line = j / (LINE_BASES + 1), col = j - line * (LINE_BASES + 1);
            const uint64_t src = line * LINE_BASES + col;
            w[j] = (col == LINE_BASES || src >= len) ? (uint8_t)'\n' : seq[src];
        }
    }
}

int pwrite_all(int fd, const char *p, size_t n, uint64_t off)
{
    while (n) {
        ssize_t w = ::pwrite(fd, p, n, (off_t)off);
        if (w < 0) {
            if (errno == EINTR)
                continue;
            set_error("pwrite: %s", strerror(errno));
            return BRX_ERR_ARG;
        }
        p += w;
        n -= (size_t)w;
        off += (uint64_t)w;
    }
    return BRX_OK;
}

struct Shared {
    std::mutex mu;
    int status = BRX_OK;
    std::string message;
    void fail(int st)
    {
        std::lock_guard<std::mutex> g(mu);
        if (status == BRX_OK) {
            status = st;
            message = brx_last_error();
        }
    }
    bool failed()
    {
        std::lock_guard<std::mutex> g(mu);
        return status != BRX_OK;
    }
};

// device-side buffers of one GPU worker
struct DevBufs {
    uint8_t *d_in = nullptr, *d_out = nullptr;
    uint64_t *d_off = nullptr, *d_out_off = nullptr;
    uint64_t in_cap = 0, out_cap = 0, off_cap = 0;
    // the FASTA text of the batch (format_kernel): definitions, their ends, record starts, the text
    uint8_t *d_defs = nullptr, *d_text = nullptr;
    uint32_t *d_def_end = nullptr;
    uint64_t *d_text_off = nullptr;
    uint64_t defs_cap = 0, text_cap = 0, tmeta_cap = 0;
    int ensure_text_meta(uint64_t defs_bytes, uint32_t n)
    {
        if (defs_bytes + 1 > defs_cap) {
            if (d_defs)
                (void)hipFree(d_defs);
            d_defs = nullptr;
            defs_cap = defs_bytes + defs_bytes / 4 + 4096;
            BRX_HIP(hipMalloc((void **)&d_defs, defs_cap));
        }
        if ((uint64_t)n + 1 > tmeta_cap) {
            if (d_def_end)
                (void)hipFree(d_def_end);
            if (d_text_off)
                (void)hipFree(d_text_off);
            d_def_end = nullptr;
            d_text_off = nullptr;
            tmeta_cap = (uint64_t)n + n / 8 + 64;
            BRX_HIP(hipMalloc((void **)&d_def_end, tmeta_cap * 4));
            BRX_HIP(hipMalloc((void **)&d_text_off, tmeta_cap * 8));
        }
        return BRX_OK;
    }
    int ensure_text(uint64_t bytes)
    {
        if (bytes + 64 > text_cap) {
            if (d_text)
                (void)hipFree(d_text);
            d_text = nullptr;
            text_cap = bytes + bytes / 8 + 4096;
            BRX_HIP(hipMalloc((void **)&d_text, text_cap));
        }
        return BRX_OK;
    }
    int ensure(uint64_t bases, uint32_t n)
    {
        if (bases + 64 > in_cap) {
            if (d_in)
                (void)hipFree(d_in);
            d_in = nullptr;
            in_cap = bases + bases / 8 + 4096;
            BRX_HIP(hipMalloc((void **)&d_in, in_cap));
        }
        const uint64_t want_out = bases + bases / 16 + 4096;
        if (want_out > out_cap) {
            if (d_out)
                (void)hipFree(d_out);
            d_out = nullptr;
            out_cap = want_out + want_out / 8;
            BRX_HIP(hipMalloc((void **)&d_out, out_cap));
        }
        if ((uint64_t)n + 1 > off_cap) {
            if (d_off)
                (void)hipFree(d_off);
            if (d_out_off)
                (void)hipFree(d_out_off);
            d_off = d_out_off = nullptr;
            off_cap = (uint64_t)n + n / 8 + 64;
            BRX_HIP(hipMalloc((void **)&d_off, off_cap * 8));
            BRX_HIP(hipMalloc((void **)&d_out_off, off_cap * 8));
        }
        return BRX_OK;
    }
    ~DevBufs()
    {
        if (d_in)
            (void)hipFree(d_in);
        if (d_out)
            (void)hipFree(d_out);
        if (d_off)
            (void)hipFree(d_off);
        if (d_out_off)
            (void)hipFree(d_out_off);
        for (void *q : {(void *)d_defs, (void *)d_text, (void *)d_def_end, (void *)d_text_off})
            if (q)
                (void)hipFree(q);
    }
};

int correct_one_batch(brx_chain_t *chain, DevBufs &dv, hipStream_t s, Batch &b)
{
    const uint32_t n = b.n();
    b.text_len = 0;
    b.out_total = 0;
    if (n == 0)
        return BRX_OK;
    const bool tr = pipe_trace();
    const double t0 = now_s();
    BRX_TRY(dv.ensure(b.total, n));
    const double t1 = now_s();
    if (b.total)
        BRX_HIP(hipMemcpyAsync(dv.d_in, b.bases.p, b.total, hipMemcpyHostToDevice, s));
    BRX_HIP(hipMemcpyAsync(dv.d_off, b.offsets.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, s));
    if (tr)
        BRX_HIP(hipStreamSynchronize(s));
    const double t2 = now_s();
    uint64_t out_total = 0;
    int st = brx_chain_correct_batch_device(chain, dv.d_in, dv.d_off, n, b.total, dv.d_out, dv.out_cap, dv.d_out_off,
                                            &out_total, s);
    if (st == BRX_ERR_OVERFLOW && out_total > dv.out_cap) { // corrections grew the batch beyond the estimate
        (void)hipFree(dv.d_out);
        dv.d_out = nullptr;
        dv.out_cap = out_total + out_total / 16 + 4096;
        BRX_HIP(hipMalloc((void **)&dv.d_out, dv.out_cap));
        st = brx_chain_correct_batch_device(chain, dv.d_in, dv.d_off, n, b.total, dv.d_out, dv.out_cap, dv.d_out_off,
                                            &out_total, s);
    }
    BRX_TRY(st);
    b.out_total = out_total;
    const double t3 = now_s();
    // the FASTA text, on the device: record starts from the corrected lengths and the definitions' lengths, then the text
    BRX_TRY(dv.ensure_text_meta(b.defs.size(), n));
    if (!b.defs.empty())
        BRX_HIP(hipMemcpyAsync(dv.d_defs, b.defs.data(), b.defs.size(), hipMemcpyHostToDevice, s));
    BRX_HIP(hipMemcpyAsync(dv.d_def_end, b.def_end.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    text_offsets_kernel<<<1, 1024, 0, s>>>(dv.d_out_off, dv.d_def_end, n, dv.d_text_off);
    uint64_t text_len = 0;
    BRX_HIP(hipMemcpyAsync(&text_len, dv.d_text_off + n, 8, hipMemcpyDeviceToHost, s));
    BRX_HIP(hipStreamSynchronize(s));
    BRX_TRY(dv.ensure_text(text_len));
    BRX_TRY(b.text.reserve(text_len + 64, 0));
    const double t4 = now_s();
    format_kernel<<<n < 4096u ? n : 4096u, 256, 0, s>>>(dv.d_out, dv.d_out_off, dv.d_defs, dv.d_def_end, dv.d_text_off, n, dv.d_text);
    BRX_HIP(hipGetLastError());
    if (text_len)
        BRX_HIP(hipMemcpyAsync(b.text.p, dv.d_text, text_len, hipMemcpyDeviceToHost, s));
    BRX_HIP(hipStreamSynchronize(s));
    b.text_len = text_len;
    if (tr)
        fprintf(stderr, "[brx pipe] batch %llu (%u records, %.1f MB): dev alloc %.2f, h2d %.2f, correct %.2f, text offsets + buffers %.2f, format + d2h %.2f ms\n",
                (unsigned long long)b.seq, n, (double)b.total / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3,
                (now_s() - t4) * 1e3);
    return BRX_OK;
}

// shared by brx_count_fasta_fd / brx_set_insert_fasta_fd: parse on a thread, hand every batch (already on the
// device) to `consume(d_bases, d_offsets, n_reads, total_bases, stream)`
template <class F>
int stream_fasta_batches(int device, int in_fd, uint32_t max_batch_records, hipStream_t s, uint64_t *stats8, F consume)
{
    const double t_start = now_s();
    std::vector<Batch> slots(3);
    Queue<Batch *> q_free, q_ready;
    Shared sh;
    double t_read = 0, t_gpu = 0;
    uint64_t n_records = 0, bases_in = 0, n_batches = 0;
    for (auto &sl : slots)
        q_free.push(&sl);
    std::thread reader([&] {
        if (hipSetDevice(device) != hipSuccess) {
            set_error("hipSetDevice in reader thread failed");
            sh.fail(BRX_ERR_HIP);
            q_ready.close();
            return;
        }
        FastaBatcher fb(in_fd, max_batch_records);
        for (;;) {
            Batch *b = nullptr;
            if (!q_free.pop(b) || sh.failed())
                break;
            const double t0 = now_s();
            int st = fb.fill(*b);
            t_read += now_s() - t0;
            if (st != BRX_OK) {
                sh.fail(st);
                break;
            }
            const bool last = b->last;
            q_ready.push(b);
            if (last)
                break;
        }
        q_ready.close();
    });
    DevBufs dv;
    Batch *b = nullptr;
    while (q_ready.pop(b)) {
        if (!sh.failed() && b->n()) {
            const double t0 = now_s();
            int st = dv.ensure(b->total, b->n());
            hipError_t e = hipSuccess;
            if (st == BRX_OK && b->total)
                e = hipMemcpyAsync(dv.d_in, b->bases.p, b->total, hipMemcpyHostToDevice, s);
            if (st == BRX_OK && e == hipSuccess)
                e = hipMemcpyAsync(dv.d_off, b->offsets.data(), ((size_t)b->n() + 1) * 8, hipMemcpyHostToDevice, s);
            if (st == BRX_OK && e == hipSuccess)
                st = consume(dv.d_in, dv.d_off, b->n(), b->total, s);
            if (st == BRX_OK && e == hipSuccess)
                e = hipStreamSynchronize(s); // the batch's device buffers are reused by the next one
            if (e != hipSuccess) {
                set_error("fasta stream: %s", hipGetErrorString(e));
                st = BRX_ERR_HIP;
            }
            t_gpu += now_s() - t0;
            if (st != BRX_OK)
                sh.fail(st);
            n_records += b->n();
            bases_in += b->total;
            n_batches++;
        }
        q_free.push(b);
    }
    reader.join();
    q_free.close();
    if (stats8) {
        for (int i = 0; i < 8; i++)
            stats8[i] = 0;
        stats8[0] = n_records;
        stats8[1] = bases_in;
        stats8[3] = n_batches;
        stats8[4] = (uint64_t)(t_read * 1e9);
        stats8[5] = (uint64_t)(t_gpu * 1e9);
        stats8[7] = (uint64_t)((now_s() - t_start) * 1e9);
    }
    if (sh.status != BRX_OK) {
        set_error("%s", sh.message.c_str());
        return sh.status;
    }
    return BRX_OK;
}

} // namespace

extern "C" {

int brx_run_correction_fd(const brx_set_t *set, const brx_method_t *methods, uint32_t n_methods, bool two_side, int in_fd,
                          int out_fd, uint32_t max_batch_records, uint64_t *stats8)
{
    if (!set || (!methods && n_methods) || n_methods == 0) {
        set_error("null argument / empty method list");
        return BRX_ERR_ARG;
    }
    BRX_TRY(use_device(set->device));
    const int device = set->device;
    const double t_start = now_s();
    std::vector<Batch> slots(N_SLOTS);
    Queue<Batch *> q_free, q_ready;
    // the writer needs the batches in stream order: done batches wait in a small map keyed by sequence number
    std::mutex done_mu;
    std::condition_variable done_cv;
    std::vector<Batch *> done; // unordered, few entries
    bool gpu_finished = false;
    Shared sh;
    uint64_t n_records = 0, bases_in = 0, bases_out = 0, n_batches = 0;
    double t_read = 0, t_gpu = 0, t_write = 0;
    for (auto &s : slots)
        q_free.push(&s);

    std::thread reader([&] {
        if (hipSetDevice(device) != hipSuccess) { // pinned allocations belong to this device's context
            set_error("hipSetDevice in reader thread failed");
            sh.fail(BRX_ERR_HIP);
            q_ready.close();
            return;
        }
        FastaBatcher fb(in_fd, max_batch_records);
        uint64_t seq = 0;
        for (;;) {
            Batch *b = nullptr;
            if (!q_free.pop(b) || sh.failed())
                break;
            const double t0 = now_s();
            int st = fb.fill(*b);
            t_read += now_s() - t0;
            if (st != BRX_OK) {
                sh.fail(st);
                break;
            }
            b->seq = seq++;
            if (pipe_trace())
                fprintf(stderr, "[brx pipe] t=%.1f ms: batch %llu parsed (%.1f ms)\n", (now_s() - t_start) * 1e3,
                        (unsigned long long)b->seq, (now_s() - t0) * 1e3);
            const bool last = b->last;
            q_ready.push(b);
            if (last)
                break;
        }
        q_ready.close();
    });

    auto gpu_worker = [&]() {
        brx_chain_t *chain = nullptr;
        hipStream_t s = nullptr;
        DevBufs dv;
        int st = use_device(device);
        if (st == BRX_OK)
            st = brx_chain_new(set, methods, n_methods, two_side, &chain);
        if (st == BRX_OK && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            set_error("hipStreamCreate failed");
            st = BRX_ERR_HIP;
        }
        if (st != BRX_OK)
            sh.fail(st);
        Batch *b = nullptr;
        while (q_ready.pop(b)) {
            if (!sh.failed()) {
                const double t0 = now_s();
                st = correct_one_batch(chain, dv, s, *b); // (leaves the batch's FASTA text in b->text)
                const double dt = now_s() - t0;
                if (st != BRX_OK)
                    sh.fail(st);
                if (pipe_trace())
                    fprintf(stderr, "[brx pipe] t=%.1f ms: batch %llu corrected + formatted (%.1f ms)\n", (now_s() - t_start) * 1e3,
                            (unsigned long long)b->seq, dt * 1e3);
                std::lock_guard<std::mutex> g(done_mu);
                t_gpu += dt;
            }
            {
                std::lock_guard<std::mutex> g(done_mu);
                done.push_back(b);
            }
            done_cv.notify_all();
        }
        if (chain)
            brx_chain_free(chain);
        if (s)
            (void)hipStreamDestroy(s);
    };
    // two workers keep H2D, kernels, D2H and formatting of consecutive batches overlapped (BRX_PIPE_WORKERS: 1..3)
    static const int n_workers = [] {
        const char *e = getenv("BRX_PIPE_WORKERS");
        const int v = e ? atoi(e) : 2;
        return v < 1 ? 1 : (v > 3 ? 3 : v);
    }();
    std::vector<std::thread> gpu_threads;
    for (int i = 0; i < n_workers; i++)
        gpu_threads.emplace_back(gpu_worker);

    // The writer: the batches in stream order from one thread.  With BRX_PIPE_WRITERS=2..4 a seekable regular file (not
    // O_APPEND) gets them from that many threads with pwrite at the offset the batch has in the stream, which is known as
    // soon as every batch before it has been formatted (its length is then on record), not written.  Off by default:
    // buffered writes to one file serialise on its inode lock (tmpfs, 1 Gbp: two writers took 0.35 s of write time
    // between them against 0.16 s for one, the job 0.27 s against 0.25), so it only pays on a file system without that.
    const off_t base_off = lseek(out_fd, 0, SEEK_CUR);
    struct stat st_out;
    const int fl = fcntl(out_fd, F_GETFL);
    const int n_writers_env = [] { // read per call
        const char *e = getenv("BRX_PIPE_WRITERS");
        const int v = e ? atoi(e) : 1;
        return v < 1 ? 1 : (v > 4 ? 4 : v);
    }();
    const bool positional = n_writers_env > 1 && base_off >= 0 && fl >= 0 && !(fl & O_APPEND) && fstat(out_fd, &st_out) == 0 &&
                            S_ISREG(st_out.st_mode);
    const int n_writers = positional ? n_writers_env : 1;
    // stream offsets: len_of[seq] is recorded when the batch arrives in `done`; [0, frontier_seq) have their offsets
    std::vector<uint64_t> len_of, off_of;
    std::vector<char> have_len;
    uint64_t frontier_seq = 0, frontier_off = 0;
    auto advance_frontier = [&]() { // done_mu held
        for (auto *d : done) {
            if (d->seq >= have_len.size()) {
                have_len.resize(d->seq + 64, 0);
                len_of.resize(d->seq + 64, 0);
                off_of.resize(d->seq + 64, 0);
            }
            if (!have_len[d->seq]) {
                have_len[d->seq] = 1;
                len_of[d->seq] = d->text_len;
            }
        }
        while (frontier_seq < have_len.size() && have_len[frontier_seq]) {
            off_of[frontier_seq] = frontier_off;
            frontier_off += len_of[frontier_seq];
            frontier_seq++;
        }
    };
    uint64_t next_in_order = 0; // (one writer, no offsets: the next batch of the stream)
    auto writer_fn = [&]() {
        for (;;) {
            Batch *b = nullptr;
            uint64_t off = 0;
            {
                std::unique_lock<std::mutex> g(done_mu);
                done_cv.wait(g, [&] {
                    advance_frontier();
                    for (auto *d : done)
                        if (positional ? d->seq < frontier_seq : d->seq == next_in_order)
                            return true;
                    return gpu_finished;
                });
                for (size_t i = 0; i < done.size(); i++)
                    if (positional ? done[i]->seq < frontier_seq : done[i]->seq == next_in_order) {
                        b = done[i];
                        done.erase(done.begin() + (long)i);
                        break;
                    }
                if (!b)
                    return; // GPU side finished and nothing writable is left
                off = positional ? off_of[b->seq] : 0;
                next_in_order++;
            }
            int st = BRX_OK;
            const double t0 = now_s();
            if (!sh.failed())
                st = positional ? pwrite_all(out_fd, (const char *)b->text.p, b->text_len, (uint64_t)base_off + off)
                                : write_all(out_fd, (const char *)b->text.p, b->text_len);
            const double dt = now_s() - t0;
            if (pipe_trace())
                fprintf(stderr, "[brx pipe] t=%.1f ms: batch %llu written (%.1f ms)\n", (now_s() - t_start) * 1e3,
                        (unsigned long long)b->seq, dt * 1e3);
            if (st != BRX_OK)
                sh.fail(st);
            {
                std::lock_guard<std::mutex> g(done_mu);
                t_write += dt;
                if (st == BRX_OK && !sh.failed()) {
                    n_records += b->n();
                    bases_in += b->total;
                    bases_out += b->out_total;
                    n_batches++;
                }
            }
            q_free.push(b);
        }
    };
    std::vector<std::thread> writers;
    for (int i = 0; i < n_writers; i++)
        writers.emplace_back(writer_fn);

    reader.join();
    for (auto &t : gpu_threads)
        t.join();
    {
        std::lock_guard<std::mutex> g(done_mu);
        gpu_finished = true;
    }
    done_cv.notify_all();
    for (auto &t : writers)
        t.join();
    q_free.close();
    if (positional && lseek(out_fd, base_off + (off_t)frontier_off, SEEK_SET) < 0 && sh.status == BRX_OK) {
        set_error("lseek to the end of the output: %s", strerror(errno));
        return BRX_ERR_ARG;
    }
    if (stats8) {
        stats8[0] = n_records;
        stats8[1] = bases_in;
        stats8[2] = bases_out;
        stats8[3] = n_batches;
        stats8[4] = (uint64_t)(t_read * 1e9);
        stats8[5] = (uint64_t)(t_gpu * 1e9);
        stats8[6] = (uint64_t)(t_write * 1e9);
        stats8[7] = (uint64_t)((now_s() - t_start) * 1e9);
    }
    if (sh.status != BRX_OK) {
        set_error("%s", sh.message.c_str());
        return sh.status;
    }
    return BRX_OK;
}

int brx_set_insert_fasta_fd(brx_set_t *set, int in_fd, uint32_t max_batch_records, uint64_t *stats8)
{
    if (!set) {
        set_error("null set");
        return BRX_ERR_ARG;
    }
    BRX_TRY(use_device(set->device));
    hipStream_t s = nullptr;
    BRX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int st = stream_fasta_batches(set->device, in_fd, max_batch_records, s, stats8,
                                  [&](const uint8_t *d_b, const uint64_t *d_o, uint32_t n, uint64_t total, hipStream_t st_) {
                                      return brx_set_insert_batch_device(set, d_b, d_o, n, total, st_);
                                  });
    (void)hipStreamDestroy(s);
    return st;
}

int brx_count_fasta_fd(brx_counter_t *c, int in_fd, uint32_t max_batch_records, uint64_t *stats8)
{
    if (!c) {
        set_error("null counter");
        return BRX_ERR_ARG;
    }
    BRX_TRY(use_device(c->device));
    return stream_fasta_batches(c->device, in_fd, max_batch_records, c->stream, stats8,
                                [&](const uint8_t *d_b, const uint64_t *d_o, uint32_t n, uint64_t total, hipStream_t st_) {
                                    return brx_set_count_add_batch_device(c, d_b, d_o, n, total, st_);
                                });
}

} // extern "C"
