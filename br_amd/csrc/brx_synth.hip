// Deterministic synthetic ONT-like reads: the same
// counter-based generator on host and device, so tests can compare the two bit for bit and the
// CPU baseline can regenerate any slice of a GPU-resident data set without a transfer.
#include "brx_internal.hpp"

namespace brx {
uint64_t scan_tmp_bytes(uint32_t n);
int exclusive_scan_lens(const uint32_t *d_lens, uint32_t n, uint64_t *d_tmp, uint64_t *d_out_offsets,
                        unsigned long long *d_total, hipStream_t s);
}

using namespace brx;

namespace {

BRX_HD uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

BRX_HD uint8_t genome_code(uint64_t seed, uint64_t g)
{
    const uint64_t w = mix64(seed ^ ((g >> 5) * 0xD1B54A32D192ED03ull));
    return (uint8_t)((w >> (2 * (g & 31))) & 3u);
}

struct ReadGen {
    uint64_t s;
    BRX_HD uint64_t next()
    {
        s += 0x9E3779B97F4A7C15ull;
        return mix64(s);
    }
};

// generates read r; writes at most cap bytes to out (nullptr: count only); returns the length
BRX_HD uint32_t gen_read(const brx_synth_t &cfg, const uint8_t *genome, uint64_t r, uint8_t *out, uint64_t cap)
{
    ReadGen g;
    g.s = mix64((cfg.seed + 1) ^ (r * 0xA24BAED4963EE407ull));
    const uint64_t span = cfg.genome_len - cfg.read_len + 1;
    const uint64_t start = g.next() % span;
    const bool minus = g.next() & 1ull;
    const uint32_t t_sub = cfg.sub_e4, t_ins = t_sub + cfg.ins_e4, t_del = t_ins + cfg.del_e4;
    uint32_t n = 0;
    for (uint32_t j = 0; j < cfg.read_len; j++) {
        const uint64_t pos = minus ? (start + cfg.read_len - 1 - j) : (start + j);
        uint64_t code = nuc2bit(genome[pos]);
        if (minus)
            code ^= 2ull; // complement
        const uint64_t u = g.next();
        const uint32_t e = (uint32_t)(u % 10000ull);
        const uint32_t r2 = (uint32_t)(u >> 32);
        if (e < t_sub) {
            const uint8_t b = bit2nuc((code + 1 + (r2 % 3u)) & 3ull);
            if (out && n < cap)
                out[n] = b;
            n++;
        } else if (e < t_ins) {
            if (out && n < cap)
                out[n] = bit2nuc(r2 & 3u);
            n++;
            if (out && n < cap)
                out[n] = bit2nuc(code);
            n++;
        } else if (e < t_del) {
            // deleted
        } else {
            if (out && n < cap)
                out[n] = bit2nuc(code);
            n++;
        }
    }
    return n;
}

__global__ void genome_kernel(brx_synth_t cfg, uint8_t *genome)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < cfg.genome_len; g += stride)
        genome[g] = bit2nuc(genome_code(cfg.seed, g));
}

__global__ void read_lens_kernel(brx_synth_t cfg, const uint8_t *genome, uint64_t first, uint32_t n, uint32_t *lens)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n)
        lens[t] = gen_read(cfg, genome, first + t, nullptr, 0);
}

__global__ void read_write_kernel(brx_synth_t cfg, const uint8_t *genome, uint64_t first, uint32_t n,
                                  const uint64_t *offsets, uint8_t *bases)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n)
        (void)gen_read(cfg, genome, first + t, bases + offsets[t], offsets[t + 1] - offsets[t]);
}

int check_cfg(const brx_synth_t *cfg)
{
    if (!cfg || cfg->read_len == 0 || cfg->genome_len < cfg->read_len ||
        (uint64_t)cfg->sub_e4 + cfg->ins_e4 + cfg->del_e4 > 10000ull) {
        set_error("bad synth config");
        return BRX_ERR_ARG;
    }
    return BRX_OK;
}

} // namespace

extern "C" {

int brx_synth_genome_host(const brx_synth_t *cfg, uint8_t *genome)
{
    BRX_TRY(check_cfg(cfg));
    if (!genome)
        return BRX_ERR_ARG;
    for (uint64_t g = 0; g < cfg->genome_len; g++)
        genome[g] = bit2nuc(genome_code(cfg->seed, g));
    return BRX_OK;
}

int brx_synth_reads_host(const brx_synth_t *cfg, const uint8_t *genome, uint64_t first_read, uint32_t n_reads,
                         uint8_t *bases, uint64_t bases_cap, uint64_t *offsets, uint64_t *total)
{
    BRX_TRY(check_cfg(cfg));
    if (!genome || !offsets || !total)
        return BRX_ERR_ARG;
    uint64_t run = 0;
    offsets[0] = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        const uint32_t n = gen_read(*cfg, genome, first_read + r, nullptr, 0);
        run += n;
        offsets[r + 1] = run;
    }
    *total = run;
    if (!bases || bases_cap < run) {
        set_error("synth reads need %llu bytes, buffer has %llu", (unsigned long long)run,
                  (unsigned long long)bases_cap);
        return BRX_ERR_OVERFLOW;
    }
    for (uint32_t r = 0; r < n_reads; r++)
        (void)gen_read(*cfg, genome, first_read + r, bases + offsets[r], offsets[r + 1] - offsets[r]);
    return BRX_OK;
}

int brx_synth_genome_device(const brx_synth_t *cfg, int device, uint8_t *d_genome, void *stream)
{
    BRX_TRY(check_cfg(cfg));
    if (!d_genome)
        return BRX_ERR_ARG;
    BRX_TRY(use_device(device));
    hipStream_t s = (hipStream_t)stream;
    genome_kernel<<<2048, 256, 0, s>>>(*cfg, d_genome);
    BRX_HIP(hipGetLastError());
    BRX_HIP(hipStreamSynchronize(s));
    return BRX_OK;
}

int brx_synth_reads_device(const brx_synth_t *cfg, int device, const uint8_t *d_genome, uint64_t first_read,
                           uint32_t n_reads, uint8_t *d_bases, uint64_t bases_cap, uint64_t *d_offsets,
                           uint64_t *total, void *stream)
{
    BRX_TRY(check_cfg(cfg));
    if (!d_genome || !d_offsets || !total)
        return BRX_ERR_ARG;
    BRX_TRY(use_device(device));
    hipStream_t s = (hipStream_t)stream;
    *total = 0;
    if (n_reads == 0) {
        BRX_HIP(hipMemsetAsync(d_offsets, 0, 8, s));
        BRX_HIP(hipStreamSynchronize(s));
        return BRX_OK;
    }
    uint32_t *d_lens = nullptr;
    uint64_t *d_tmp = nullptr;
    unsigned long long *d_total = nullptr;
    BRX_HIP(hipMalloc((void **)&d_lens, (uint64_t)n_reads * 4));
    hipError_t e = hipMalloc((void **)&d_tmp, scan_tmp_bytes(n_reads) + 8);
    if (e != hipSuccess) {
        (void)hipFree(d_lens);
        set_error("hipMalloc: %s", hipGetErrorString(e));
        return BRX_ERR_NOMEM;
    }
    d_total = (unsigned long long *)((uint8_t *)d_tmp + scan_tmp_bytes(n_reads));
    int st = BRX_OK;
    read_lens_kernel<<<(n_reads + 63) / 64, 64, 0, s>>>(*cfg, d_genome, first_read, n_reads, d_lens);
    st = exclusive_scan_lens(d_lens, n_reads, d_tmp, d_offsets, d_total, s);
    unsigned long long tot = 0;
    if (st == BRX_OK) {
        e = hipMemcpyAsync(&tot, d_total, 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess)
            e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            set_error("synth: %s", hipGetErrorString(e));
            st = BRX_ERR_HIP;
        }
    }
    if (st == BRX_OK) {
        *total = tot;
        if (!d_bases || bases_cap < tot) {
            set_error("synth reads need %llu bytes, buffer has %llu", tot, (unsigned long long)bases_cap);
            st = BRX_ERR_OVERFLOW;
        }
    }
    if (st == BRX_OK) {
        read_write_kernel<<<(n_reads + 63) / 64, 64, 0, s>>>(*cfg, d_genome, first_read, n_reads, d_offsets, d_bases);
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            set_error("synth write: %s", hipGetErrorString(e));
            st = BRX_ERR_HIP;
        }
    }
    (void)hipFree(d_lens);
    (void)hipFree(d_tmp);
    return st;
}

} // extern "C"
