// What limits the W loaders of K1 (k_softemb_fwd_ws3 / _sk)?  Four waves of a 1024-thread workgroup bring a 320 x 64 bf16
// tile (40 KB) per k-tile into LDS with LDS-direct loads (10 instructions of 1 KB per wave), three stages, two tiles in
// flight, one workgroup barrier per tile -- here WITHOUT the compute and S-loader roles (the other 12 waves only meet
// the barrier), for two source layouts of the same 6 MB matrix:
//   rows   : W^T as packed today, [320][Vp] row-major: an instruction gathers 8 rows x 128 B (row pitch 18.7 KB)
//   tiles  : k-tile-major, [Vp/64][320][64]: the 40 KB of a tile are contiguous, an instruction reads 1 KB linear
// One workgroup per CU, every workgroup sweeps all k-tiles starting at its own offset (as stream-K does).
// Prints us per k-tile and GB/s per CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/ldsdma tools/micro/ldsdma.hip && ./tools/micro/ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(1))) const void gmem_cv;
typedef __attribute__((address_space(3))) void lds_v;

template <int TILED, int NW>
__global__ __launch_bounds__(1024) void k_fill(const unsigned short* __restrict__ w, int Vp, int nk, int tiles_per_wg, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int start = (int)((long long)blockIdx.x * 37 % nk);
    if (wave < 16 - NW) {
        for (int t = 0; t <= tiles_per_wg; ++t) __builtin_amdgcn_s_barrier();
        if (tid == 0 && lds[lane] == 0x5a && lds[40960 + lane] == 0x77) sink[0] = 1;
        return;
    }
    const int ww = wave - (16 - NW), r = lane >> 3, sl = lane & 7;
    constexpr int NI = 40 / NW;                     // instructions per wave and tile
    auto issue = [&](int kt, int st) {
        unsigned char* dst = lds + st * 40960 + ww * (NI * 1024);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = (ww * NI + i) * 8 + r;
            const unsigned short* src = TILED ? w + ((long long)kt * 320 + row) * 64 + sl * 8
                                              : w + (long long)row * Vp + kt * 64 + sl * 8;
            __builtin_amdgcn_global_load_lds((gmem_cv*)src, (lds_v*)(dst + i * 1024), 16, 0, 0);
        }
    };
    auto kt_of = [&](int t) { int k = start + t; return k >= nk ? k - nk : k; };
    issue(kt_of(0), 0);
    issue(kt_of(1), 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    __builtin_amdgcn_s_barrier();
    int st2 = 2;
    for (int t = 0; t < tiles_per_wg; ++t) {
        if (t + 2 < tiles_per_wg) {
            issue(kt_of(t + 2), st2);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        st2 = st2 == 2 ? 0 : st2 + 1;
        __builtin_amdgcn_s_barrier();
    }
}

template <int TILED, int NW>
static void run(const char* name, const unsigned short* w, int Vp, unsigned* sink) {
    const int nk = Vp / 64, tiles = nk;             // every workgroup streams the whole matrix once (6 MB)
    hipFuncSetAttribute((const void*)k_fill<TILED, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 40960);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_fill<TILED, NW>), dim3(256), dim3(1024), 3 * 40960, 0, w, Vp, nk, tiles, sink);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_fill<TILED, NW>), dim3(256), dim3(1024), 3 * 40960, 0, w, Vp, nk, tiles, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, per = us / tiles;
    printf("%-6s %d loader waves: %8.1f us per launch, %.3f us per 40 KB tile, %.1f GB/s per CU (%s)\n", name, NW, us, per, 40960 / per / 1e3,
           hipGetErrorString(hipGetLastError()));
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    const int Vp = 9408;                            // V = 9353 padded to 64
    unsigned short* w;
    unsigned* sink;
    hipMalloc(&w, (size_t)320 * Vp * 2 + 4096);
    hipMalloc(&sink, 16);
    hipMemset(w, 0, (size_t)320 * Vp * 2 + 4096);
    run<0, 4>("rows", w, Vp, sink);
    run<1, 4>("tiles", w, Vp, sink);
    run<0, 8>("rows", w, Vp, sink);
    run<1, 8>("tiles", w, Vp, sink);
    run<0, 2>("rows", w, Vp, sink);
    run<1, 2>("tiles", w, Vp, sink);
    return 0;
}
