// Checks what global_load_lds_dwordx4 does on gfx950 before brx_onelane.hip relies on it: lane l of a wave writes its
// 16 bytes at  M0_base + 16 * l  (a wave-uniform LDS base per instruction), lanes masked off by exec write nothing.
// build: hipcc -O3 --offload-arch=gfx950 tools/lds_dma_test.hip -o tools/lds_dma_test ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const uint4 *g, uint4 *out, const int *idx)
{
    __shared__ uint4 lc[4][256];
    const int tid = threadIdx.x, wv = tid >> 6;
    for (int e = 0; e < 4; e++)
        lc[e][tid] = make_uint4(0xdeadu, tid, e, 0);
    __syncthreads();
    const uint4 *L = g + (size_t)idx[tid] * 4;
    if (idx[tid] & 1) {
#pragma unroll
        for (int e = 0; e < 4; e++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(L + e),
                                             (__attribute__((address_space(3))) void *)&lc[e][wv * 64], 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int e = 0; e < 4; e++)
        out[e * 256 + tid] = lc[e][tid];
}
int main()
{
    const int N = 4096;
    std::vector<uint4> h(N * 4);
    for (int i = 0; i < N * 4; i++)
        h[i] = make_uint4(i, i * 3 + 1, 0xabc, i ^ 0x55);
    std::vector<int> idx(256);
    for (int t = 0; t < 256; t++)
        idx[t] = (t * 37 + 11) % N;
    uint4 *dg, *dout;
    int *didx;
    hipMalloc(&dg, h.size() * 16);
    hipMalloc(&dout, 1024 * 16);
    hipMalloc(&didx, 256 * 4);
    hipMemcpy(dg, h.data(), h.size() * 16, hipMemcpyHostToDevice);
    hipMemcpy(didx, idx.data(), 256 * 4, hipMemcpyHostToDevice);
    k<<<1, 256>>>(dg, dout, didx);
    std::vector<uint4> o(1024);
    if (hipMemcpy(o.data(), dout, 1024 * 16, hipMemcpyDeviceToHost) != hipSuccess) {
        printf("FAIL: kernel error\n");
        return 1;
    }
    int bad = 0;
    for (int e = 0; e < 4; e++)
        for (int t = 0; t < 256; t++) {
            const uint4 v = o[e * 256 + t];
            uint4 exp = (idx[t] & 1) ? h[(size_t)idx[t] * 4 + e] : make_uint4(0xdeadu, t, e, 0);
            if (v.x != exp.x || v.y != exp.y || v.z != exp.z || v.w != exp.w) {
                if (bad < 5)
                    printf("mismatch e=%d t=%d got %x %x %x %x exp %x %x %x %x\n", e, t, v.x, v.y, v.z, v.w, exp.x, exp.y, exp.z, exp.w);
                bad++;
            }
        }
    printf(bad ? "FAIL: %d mismatches\n" : "OK: global_load_lds_dwordx4 writes lane l at base + 16 l, masked lanes untouched (%d)\n", bad);
    return bad != 0;
}
