// Issue cost of the VALU / LDS-crossbar instructions the correction kernel is made of, on the card in hand.
// The kernel is VALU-issue-bound (SQ_ACTIVE_INST_VALU ~ 75 % of the SIMD cycles, profiles/r1k_sq_summary.json), so
// what an instruction costs the SIMD decides how its instruction mix should be changed.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate && tools/valu_rate
// Output: cycles of one SIMD per wave-instruction (4 = full rate for wave64 on a 16-lane SIMD), measured with 8
// waves per SIMD and 8 independent chains per wave, clock from s_memtime / wall time.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define CHECK(e)                                                                                   \
    do {                                                                                           \
        hipError_t _e = (e);                                                                       \
        if (_e != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e));              \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

constexpr int ITERS = 4096;
constexpr int CH = 8;

enum Op { OP_ADD32, OP_MUL_LO, OP_MUL_HI, OP_MUL_U24, OP_MAD_U24, OP_LSHL64, OP_ADD64, OP_CMP64, OP_BFREV, OP_ALIGNBIT, OP_BCNT,
          OP_MIN, OP_DPP_SHR, OP_DPP_OR, OP_DPP_WAVE_SHR, OP_BPERMUTE, OP_SWIZZLE, OP_READLANE, OP_BFE, OP_LSHL_OR, OP_PERM,
          OP_MAD64, OP_CNDMASK, OP_BALLOT, OP_N };
static const char *names[OP_N] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_lshlrev_b64",
                                  "add u64 (2 instr)", "v_cmp_eq_u64 (+s_or)", "v_bfrev_b32", "v_alignbit_b32", "v_bcnt_u32_b32", "v_min_u32",
                                  "v_mov_dpp row_shr:1", "v_or_b32_dpp row_shr:1", "v_mov_dpp wave_shr:1", "ds_bpermute_b32", "ds_swizzle",
                                  "v_readlane", "v_bfe_u32", "v_lshl_or_b32", "v_perm_b32", "v_mad_u64_u32", "v_cndmask", "ballot(cmp)"};

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t *out, uint32_t seed)
{
    uint32_t a[CH];
    uint64_t q[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        a[c] = seed * (threadIdx.x + 1) + c;
        q[c] = ((uint64_t)a[c] << 32) | (a[c] * 7u);
    }
    uint32_t acc = 0;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (OP == OP_ADD32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));
            if (OP == OP_MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));
            if (OP == OP_MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));
            if (OP == OP_MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[c]) : "v"(seed));
            if (OP == OP_MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[c]) : "v"(seed));
            if (OP == OP_LSHL64) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(q[c]) : "v"(seed & 3u));
            if (OP == OP_ADD64) q[c] += ((uint64_t)seed << 20) | 1u;
            if (OP == OP_CMP64) acc += (uint32_t)__builtin_popcountll(__ballot(q[c] == (uint64_t)(it + c)));
            if (OP == OP_BFREV) asm volatile("v_bfrev_b32 %0, %0" : "+v"(a[c]));
            if (OP == OP_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 6" : "+v"(a[c]) : "v"(seed));
            if (OP == OP_BCNT) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));
            if (OP == OP_MIN) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));
            if (OP == OP_DPP_SHR) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[c]));
            if (OP == OP_DPP_OR) asm volatile("s_nop 1\n v_or_b32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[c]));
            if (OP == OP_DPP_WAVE_SHR) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[c]));
            if (OP == OP_BPERMUTE) a[c] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((threadIdx.x * 4 + 4) & 255), (int)a[c]);
            if (OP == OP_SWIZZLE) a[c] = (uint32_t)__builtin_amdgcn_ds_swizzle((int)a[c], 0x041f);
            if (OP == OP_READLANE) acc += (uint32_t)__builtin_amdgcn_readlane((int)a[c], 5) + (a[c] += acc, 0u);
            if (OP == OP_BFE) asm volatile("v_bfe_u32 %0, %0, 1, 20" : "+v"(a[c]));
            if (OP == OP_LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(a[c]) : "v"(seed));
            if (OP == OP_PERM) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(seed));
            if (OP == OP_MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(q[c]) : "v"(seed) : "vcc");
            if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(seed) : "vcc");
            if (OP == OP_BALLOT) acc += (uint32_t)__ballot(a[c] > (uint32_t)it), a[c] += 3;
        }
    }
#pragma unroll
    for (int c = 0; c < CH; c++)
        acc += a[c] + (uint32_t)q[c] + (uint32_t)(q[c] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
static double run(uint32_t *d_out, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rate_kernel<OP><<<blocks, 256>>>(d_out, 3u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<OP><<<blocks, 256>>>(d_out, 5u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    const int blocks = cus * 8; // 8 blocks x 4 waves per CU = 8 waves per SIMD
    uint32_t *d_out = nullptr;
    CHECK(hipMalloc((void **)&d_out, (size_t)blocks * 256 * 4));
    printf("%s: %d CUs, %.2f GHz nominal; %d chains x %d iterations, 8 waves per SIMD\n", prop.name, cus, ghz, CH, ITERS);
    double ms[OP_N];
#define R(op) ms[op] = run<op>(d_out, blocks);
    R(OP_ADD32) R(OP_MUL_LO) R(OP_MUL_HI) R(OP_MUL_U24) R(OP_MAD_U24) R(OP_LSHL64) R(OP_ADD64) R(OP_CMP64) R(OP_BFREV) R(OP_ALIGNBIT)
    R(OP_BCNT) R(OP_MIN) R(OP_DPP_SHR) R(OP_DPP_OR) R(OP_DPP_WAVE_SHR) R(OP_BPERMUTE) R(OP_SWIZZLE) R(OP_READLANE) R(OP_BFE) R(OP_LSHL_OR)
    R(OP_PERM) R(OP_MAD64) R(OP_CNDMASK) R(OP_BALLOT)
    // per SIMD: 8 waves x CH x ITERS wave-instructions (of the measured op; loop overhead is scalar)
    const double n = 8.0 * CH * ITERS;
    for (int op = 0; op < OP_N; op++)
        printf("%-26s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.2f GHz)  x%.2f of v_add_u32\n", names[op], ms[op],
               ms[op] * 1e-3 * ghz * 1e9 / n, ghz, ms[op] / ms[OP_ADD32]);
    hipFree(d_out);
    return 0;
}
