// Where do the workgroups of a 2-per-CU persistent grid land?  Prints blockIdx -> (XCC, SE, CU) for a grid of 512
// workgroups of 256 threads with 66 KB of LDS each (the footprint of k_lstm_fwd_persist2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void k_where(unsigned* out, int hold) {
    extern __shared__ unsigned char lds[];
    lds[threadIdx.x] = (unsigned char)threadIdx.x;
    unsigned xcc = 0, hw = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
    // stay resident for a while so that the whole grid is placed at once
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < hold) __builtin_amdgcn_s_sleep(8);
    if (lds[threadIdx.x] == 255 && hold < 0) out[0] = 0;
}
int main() {
    const int G = 512;
    unsigned* d;
    hipMalloc(&d, G * 2 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 66 * 1024);
    hipLaunchKernelGGL(k_where, dim3(G), dim3(256), 66 * 1024, 0, d, 20000 /* 200 us at 100 MHz */);
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 2);
    hipMemcpy(h.data(), d, G * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int b = 0; b < G; ++b) {
        const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7, simd = (hw >> 4) & 3, wave = hw & 0xf;
        printf("b=%3d xcc=%u se=%u sh=%u cu=%2u simd=%u wave=%u hw=%08x\n", b, xcc, se, sh, cu, simd, wave, hw);
    }
    return 0;
}
