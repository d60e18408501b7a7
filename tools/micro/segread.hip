// What does the LENGTH of the contiguous run per row cost a strided row-segment read stream (K1's S operand)?
// 256 workgroups of 256 threads, one 128-row tile each (rows at cfg3's pitch of V = 9353 bf16 = 18706 bytes, or an
// aligned pitch), sweeping the rows from left to right in visits of SEG bytes per row, with 16 sixteen-byte loads
// (256 bytes) in flight per lane = 64 KB per CU whatever SEG is -- the in-flight budget of K1's S loader waves.
// SEG = 128 is what K1 does today (one 64-element k-tile per visit); 256 / 512 fetch two / four k-tiles' worth of a row
// at once.  No arithmetic, no LDS: prints the time of the 613 MB sweep and TB/s.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/segread tools/micro/segread.hip && ./tools/micro/segread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4h __attribute__((aligned(2)));      // a 16-byte load at a 2-byte boundary, as K1 issues them

template <int SEG, int ALIGNED>
__global__ __launch_bounds__(256) void k_seg(const unsigned char* S, long long pitch, int row_bytes, unsigned* sink) {
    constexpr int CPR = SEG / 16;                 // 16-byte chunks per row and visit
    constexpr int CPT = 128 * CPR / 256;          // chunks per thread and visit (4, 8, 16)
    constexpr int VIF = 16 / CPT;                 // visits in flight (4, 2, 1)
    const int tid = threadIdx.x;
    const unsigned char* base = S + (long long)blockIdx.x * 128 * pitch;
    const int nvisit = row_bytes / SEG;
    u32x4 r[16];
    u32x4 acc = u32x4{0u, 0u, 0u, 0u};
    auto addr = [&](int visit, int i) {
        const int c = i * 256 + tid, row = c / CPR, col = c % CPR;
        return base + (long long)row * pitch + (long long)visit * SEG + col * 16;
    };
    auto ld = [&](const unsigned char* p) {
        if (ALIGNED) return *(const u32x4*)p;
        return (u32x4)*(const u32x4h*)p;
    };
#pragma unroll
    for (int v = 0; v < VIF; ++v)
#pragma unroll
        for (int i = 0; i < CPT; ++i) r[v * CPT + i] = ld(addr(v < nvisit ? v : 0, i));
    for (int visit = 0; visit < nvisit; visit += VIF) {
#pragma unroll
        for (int v = 0; v < VIF; ++v) {
            const int nx = visit + VIF + v;
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                acc ^= r[v * CPT + i];
                r[v * CPT + i] = ld(addr(nx < nvisit ? nx : 0, i));
            }
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

template <class F>
static double time_us(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / reps;
}

int main() {
    const int rows = 32768;
    unsigned* sink;
    hipMalloc(&sink, 64);
    for (int pass = 0; pass < 2; ++pass) {
        const long long pitch = pass == 0 ? 18706 : 18816;          // V = 9353 (2-byte-aligned rows) / V = 9408 (128-byte-aligned)
        const int row_bytes = 18432;                                  // 144 k-tiles of 128 bytes, the same sweep for every SEG
        unsigned char* S;
        hipMalloc(&S, (size_t)rows * pitch + 4096);
        hipMemset(S, 1, (size_t)rows * pitch + 4096);
        const double mb = (double)rows * row_bytes / 1e6;
        double t;
        if (pass == 0) {
            t = time_us([&] { hipLaunchKernelGGL((k_seg<128, 0>), dim3(256), dim3(256), 0, 0, S, pitch, row_bytes, sink); });
            printf("pitch %lld  SEG 128: %7.1f us  %.2f TB/s\n", pitch, t, mb / t);
            t = time_us([&] { hipLaunchKernelGGL((k_seg<256, 0>), dim3(256), dim3(256), 0, 0, S, pitch, row_bytes, sink); });
            printf("pitch %lld  SEG 256: %7.1f us  %.2f TB/s\n", pitch, t, mb / t);
            t = time_us([&] { hipLaunchKernelGGL((k_seg<512, 0>), dim3(256), dim3(256), 0, 0, S, pitch, row_bytes, sink); });
            printf("pitch %lld  SEG 512: %7.1f us  %.2f TB/s\n", pitch, t, mb / t);
        } else {
            t = time_us([&] { hipLaunchKernelGGL((k_seg<128, 1>), dim3(256), dim3(256), 0, 0, S, pitch, row_bytes, sink); });
            printf("pitch %lld  SEG 128: %7.1f us  %.2f TB/s\n", pitch, t, mb / t);
            t = time_us([&] { hipLaunchKernelGGL((k_seg<256, 1>), dim3(256), dim3(256), 0, 0, S, pitch, row_bytes, sink); });
            printf("pitch %lld  SEG 256: %7.1f us  %.2f TB/s\n", pitch, t, mb / t);
            t = time_us([&] { hipLaunchKernelGGL((k_seg<512, 1>), dim3(256), dim3(256), 0, 0, S, pitch, row_bytes, sink); });
            printf("pitch %lld  SEG 512: %7.1f us  %.2f TB/s\n", pitch, t, mb / t);
        }
        hipFree(S);
    }
    hipFree(sink);
    return 0;
}
