// Phase timing of k_prep_lists (wall_clock64 stamps of thread 0 of every workgroup), cfg3 shape.
#include <hip/hip_runtime.h>
__device__ long long g_stamps[64 * 8];
#define PREP_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0) g_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#include "../../non-parallel-text-style-transfer-using-self-attn-discriminator_amd/csrc/kernels.hpp"
#include "../../non-parallel-text-style-transfer-using-self-attn-discriminator_amd/csrc/wgrad.hpp"
#include <cstdio>
#include <vector>
using namespace sad;
int main() {
    const int B = 1024, T = 32, M = B * T;
    Shape s{};
    s.B = B; s.T = T; s.M = M;
    std::vector<int> lens(B);
    std::vector<unsigned char> mask(M);
    for (int b = 0; b < B; ++b) {
        lens[b] = 3 + (b * 29) / B;
        for (int t = 0; t < T; ++t) mask[b * T + t] = !(t >= 1 && t < lens[b] - 1);
    }
    int *dl, *lensd, *rows, *nrows, *tiles; unsigned char *dm, *need; unsigned* sync;
    hipMalloc(&dl, B * 4); hipMalloc(&lensd, B * 4); hipMalloc(&rows, M * 12); hipMalloc(&nrows, 64); hipMalloc(&tiles, (M / 128 + 4) * 4);
    hipMalloc(&dm, M); hipMalloc(&need, M + 256); hipMalloc(&sync, 8192 * 4);
    hipMemcpy(dl, lens.data(), B * 4, hipMemcpyHostToDevice);
    hipMemcpy(dm, mask.data(), M, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k_prep_lists, hipFuncAttributeMaxDynamicSharedMemorySize, (int)prepl_lds_bytes(PREPL_MAX_M, PREPL_MAX_B));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_prep_lists, dim3((M + 1023) / 1024), dim3(1024), prepl_lds_bytes(M, B), 0, s, dl, dm, (long long)T, 1LL, 1,
                           need, lensd, sync, 8192, rows, nrows, tiles);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[64 * 8];
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h));
        printf("launch %d: %.1f us (events)\n", it, ms * 1e3);
        for (int w : {0, 1, 15, 31}) {
            printf("  wg %2d:", w);
            for (int i = 1; i < 7; ++i) printf(" %6.2f", (h[w * 8 + i] - h[w * 8 + i - 1]) * 0.01);
            printf("  | start-to-end %.2f us, start offset %.2f\n", (h[w * 8 + 6] - h[w * 8]) * 0.01, (h[w * 8] - h[0]) * 0.01);
        }
    }
    int hn[4]; hipMemcpy(hn, nrows, 16, hipMemcpyDeviceToHost);
    printf("nrows %d %d %d\n", hn[0], hn[1], hn[2]);
    return 0;
}
