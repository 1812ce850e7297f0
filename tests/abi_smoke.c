/*
 * abi_smoke.c -- a host with no Python and no torch in the process, doing what a Rust `br` would do through
 * include/brx.h: load a .solid set (Pcon::from_pcon_solid, src/set/pcon.rs:18-25), ask KmerSet::get for a few
 * k-mers (src/set.rs:17-21), build the method chain (build_methods, src/lib.rs:141-164) and push one batch of
 * records through run_correction's per-record body (src/lib.rs:42-55).  The corrected bytes are compared with a
 * file the CPU oracle wrote for the same input (tests/test_gpu_parity.py::test_abi_smoke_c_host writes the three
 * files and runs this program as a child process).
 *
 * Plain C11, compiled with gcc against the header only, linked against br_amd/lib/libbrx.so -- libamdhip64 comes
 * in through libbrx's own RUNPATH.  Exit code 0 = byte-identical.
 *
 * usage: abi_smoke SOLID_BYTES READS_BIN EXPECTED_BIN METHODS [two_side]
 *   SOLID_BYTES   raw [k][bits] stream (already decompressed)
 *   READS_BIN     u32 n_reads, u64 offsets[n+1], bases
 *   EXPECTED_BIN  u64 out_offsets[n+1], bases
 *   METHODS       comma list of method ids 0..4 (One,Two,Graph,Greedy,GapSize), confirm 5, max_search 7
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "brx.h"

static uint8_t *slurp(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) {
        perror(path);
        exit(2);
    }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc(n > 0 ? (size_t)n : 1);
    if (n > 0 && fread(buf, 1, (size_t)n, f) != (size_t)n) {
        fprintf(stderr, "%s: short read\n", path);
        exit(2);
    }
    fclose(f);
    *len = (size_t)n;
    return buf;
}

#define CHECK(expr)                                                                              \
    do {                                                                                         \
        int st_ = (expr);                                                                        \
        if (st_ != BRX_OK) {                                                                     \
            fprintf(stderr, "abi_smoke: %s -> %d (%s): %s\n", #expr, st_, brx_strerror(st_), brx_last_error()); \
            return 3;                                                                            \
        }                                                                                        \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s SOLID_BYTES READS_BIN EXPECTED_BIN METHODS [two_side]\n", argv[0]);
        return 2;
    }
    int n_dev = 0;
    CHECK(brx_device_count(&n_dev));
    if (n_dev < 1) {
        fprintf(stderr, "abi_smoke: no GPU visible (the library has no CPU fallback)\n");
        return 4;
    }
    size_t solid_len, reads_len, exp_len;
    uint8_t *solid = slurp(argv[1], &solid_len);
    uint8_t *reads = slurp(argv[2], &reads_len);
    uint8_t *expect = slurp(argv[3], &exp_len);

    brx_set_t *set = NULL;
    CHECK(brx_set_new_from_solid_bytes(solid, solid_len, 0, &set));
    const int k = brx_set_k(set);
    if (k != solid[0]) {
        fprintf(stderr, "abi_smoke: KmerSet::k = %d, file says %d\n", k, solid[0]);
        return 1;
    }

    uint32_t n_reads;
    memcpy(&n_reads, reads, 4);
    const uint64_t *offsets = (const uint64_t *)(reads + 4); /* 4-byte aligned is fine for memcpy'd use below */
    uint64_t *offs = (uint64_t *)malloc(((size_t)n_reads + 1) * 8);
    memcpy(offs, offsets, ((size_t)n_reads + 1) * 8);
    const uint8_t *bases = reads + 4 + ((size_t)n_reads + 1) * 8;

    /* KmerSet::get on the first k-mers of the first read against the file's own bits */
    if (n_reads && offs[1] - offs[0] >= (uint64_t)k) {
        uint64_t kmer = 0, mask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
        for (uint64_t i = 0; i < offs[1] - offs[0] && i < 200; i++) {
            kmer = ((kmer << 2) & mask) ^ (uint64_t)((bases[offs[0] + i] >> 1) & 3u);
            if (i + 1 < (uint64_t)k)
                continue;
            /* canonical = even-popcount member of {kmer, revcomp}; bit index = canonical >> 1 */
            uint64_t rc = 0, x = kmer;
            for (int j = 0; j < k; j++) {
                rc = (rc << 2) | ((x & 3u) ^ 2u);
                x >>= 2;
            }
            const uint64_t cano = (__builtin_popcountll(kmer) & 1) ? rc : kmer;
            const uint64_t h = cano >> 1;
            const int want = (solid[1 + (h >> 3)] >> (h & 7)) & 1;
            if ((int)brx_set_get(set, kmer) != want) {
                fprintf(stderr, "abi_smoke: KmerSet::get differs from the .solid bits at k-mer %llu\n", (unsigned long long)i);
                return 1;
            }
        }
    }

    brx_method_t methods[16];
    uint32_t n_methods = 0;
    for (char *tok = strtok(argv[4], ","); tok && n_methods < 16; tok = strtok(NULL, ",")) {
        methods[n_methods].method = (uint8_t)atoi(tok);
        methods[n_methods].confirm = 5;
        methods[n_methods].max_search = 7;
        n_methods++;
    }
    const bool two_side = argc > 5 && atoi(argv[5]) != 0;
    brx_chain_t *chain = NULL;
    CHECK(brx_chain_new(set, methods, n_methods, two_side, &chain));
    uint8_t *out = NULL;
    uint64_t *out_off = NULL;
    CHECK(brx_chain_correct_batch(chain, bases, offs, n_reads, &out, &out_off));

    const uint64_t *exp_off = (const uint64_t *)expect;
    uint64_t *eo = (uint64_t *)malloc(((size_t)n_reads + 1) * 8);
    memcpy(eo, exp_off, ((size_t)n_reads + 1) * 8);
    const uint8_t *exp_bases = expect + ((size_t)n_reads + 1) * 8;
    int rc = 0;
    if (memcmp(out_off, eo, ((size_t)n_reads + 1) * 8) != 0) {
        fprintf(stderr, "abi_smoke: corrected lengths differ from the oracle's\n");
        rc = 1;
    } else if (memcmp(out, exp_bases, (size_t)eo[n_reads]) != 0) {
        fprintf(stderr, "abi_smoke: corrected bases differ from the oracle's\n");
        rc = 1;
    }
    uint64_t st[8];
    CHECK(brx_chain_last_stats(chain, st));
    printf("abi_smoke: k=%d reads=%u bases_in=%llu bases_out=%llu fixes=%llu probes=%llu %s\n", k, n_reads,
           (unsigned long long)offs[n_reads], (unsigned long long)out_off[n_reads], (unsigned long long)st[3],
           (unsigned long long)st[1], rc ? "MISMATCH" : "identical to the oracle");
    brx_buf_free(out);
    brx_buf_free(out_off);
    brx_chain_free(chain);
    brx_set_free(set);
    free(offs);
    free(eo);
    free(solid);
    free(reads);
    free(expect);
    return rc;
}
