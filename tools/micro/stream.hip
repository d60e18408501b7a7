// HBM streaming roofs that the store-bound / load-bound kernels are priced against (DESIGN.md, K1' and K1):
// a write-only fill and a read-only sweep of a buffer the size of cfg3's d(soft tokens) (T*B*V bf16 = 613 MB), as a
// plain grid-stride kernel with 16-byte accesses -- no arithmetic, no LDS.  Prints TB/s for several grid sizes and
// for default vs non-temporal stores.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/stream tools/micro/stream.hip && ./tools/micro/stream
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT_STORE>
__global__ __launch_bounds__(256) void k_fill(u32x4* p, size_t n) {
    const u32x4 v = u32x4{1u, 2u, 3u, 4u};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (NT_STORE) __builtin_nontemporal_store(v, p + i);
        else p[i] = v;
    }
}
// each workgroup writes a contiguous 32 KB piece at a time (the shape of a tile-row store)
__global__ __launch_bounds__(256) void k_fill_pieces(u32x4* p, size_t n) {
    const u32x4 v = u32x4{1u, 2u, 3u, 4u};
    const size_t piece = 2048;                       // 16-byte chunks per piece
    for (size_t b = blockIdx.x; b * piece < n; b += gridDim.x)
        for (size_t i = threadIdx.x; i < piece && b * piece + i < n; i += 256) p[b * piece + i] = v;
}
__global__ __launch_bounds__(256) void k_read(const u32x4* p, size_t n, unsigned* sink) {
    u32x4 a = u32x4{0u, 0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a ^= p[i];
    if ((a[0] ^ a[1] ^ a[2] ^ a[3]) == 0x12345u) sink[0] = 1;
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
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3 / reps;
}

int main() {
    const size_t bytes = (size_t)32 * 1024 * 9353 * 2;      // cfg3: T * B * V bf16
    const size_t n = bytes / 16;
    u32x4* buf;
    unsigned* sink;
    hipMalloc(&buf, bytes);
    hipMalloc(&sink, 16);
    hipMemset(buf, 0, bytes);
    printf("# buffer %.1f MB; us per pass and TB/s\n", bytes / 1e6);
    const int grids[5] = {256, 512, 1024, 2048, 8192};
    for (int g : grids) {
        double t = time_us([&] { hipLaunchKernelGGL(k_fill<0>, dim3(g), dim3(256), 0, 0, buf, n); });
        printf("fill   default   grid %5d: %8.1f us  %.2f TB/s\n", g, t, bytes / t / 1e6);
        t = time_us([&] { hipLaunchKernelGGL(k_fill<1>, dim3(g), dim3(256), 0, 0, buf, n); });
        printf("fill   nontemp   grid %5d: %8.1f us  %.2f TB/s\n", g, t, bytes / t / 1e6);
        t = time_us([&] { hipLaunchKernelGGL(k_fill_pieces, dim3(g), dim3(256), 0, 0, buf, n); });
        printf("fill   32KB-piece grid %5d: %8.1f us  %.2f TB/s\n", g, t, bytes / t / 1e6);
        t = time_us([&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, (const u32x4*)buf, n, sink); });
        printf("read             grid %5d: %8.1f us  %.2f TB/s\n", g, t, bytes / t / 1e6);
    }
    double t = time_us([&] { hipMemsetAsync(buf, 0, bytes, 0); });
    printf("hipMemsetAsync              : %8.1f us  %.2f TB/s\n", t, bytes / t / 1e6);
    return 0;
}
