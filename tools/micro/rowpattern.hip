// Micro-benchmark (measurement tool, not product code): how fast can a [M][V] bf16 matrix with an
// ODD row pitch (V = 9353 -> 18706-byte rows) be written / read with 16-byte accesses, as a function
// of how a wavefront's 64 lanes are spread over rows?  RPW = rows per wave instruction; each row gets
// 64/RPW lanes x 16 B contiguous.  A 512-thread workgroup owns 256 rows and walks the column range.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4h __attribute__((ext_vector_type(4), aligned(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

template <int RPW, bool STORE>
__global__ __launch_bounds__(512) void k_rows(u16* __restrict__ p, int M, int V, long long pitch, int csplit, unsigned* sink) {
    constexpr int LPR = 64 / RPW;            // lanes per row
    constexpr int CW = LPR * 8;              // columns per wave instruction
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 256 + wave * 32;       // this wave's 32 rows
    const int cper = (V + csplit - 1) / csplit;
    const int c0 = blockIdx.y * cper, c1 = min(V, c0 + cper);
    const int r = lane / LPR, cl = (lane % LPR) * 8;
    unsigned acc = 0;
    for (int c = c0; c < c1; c += CW) {
#pragma unroll
        for (int rr = 0; rr < 32; rr += RPW) {
            const int m = m0 + rr + r, col = c + cl;
            if (m < M && col + 8 <= c1) {
                u32x4h* q = (u32x4h*)(p + (long long)m * pitch + col);
                if (STORE) *q = u32x4h{(unsigned)m, (unsigned)col, 1u, 2u};
                else { u32x4h v = *q; acc += v[0] ^ v[1] ^ v[2] ^ v[3]; }
            }
        }
    }
    if (!STORE && acc == 0x12345u) *sink = acc;
}

// Row-tile sweep: the 32 workgroups of an XCD (blockIdx = xcd + 8 * slot) share ONE 256-row tile at a time and cut its
// columns into 32 strips; the XCD's tiles are xcd, xcd + 8, ...  At any moment the device writes 8 row tiles = 8
// contiguous 4.8 MB regions instead of a few hundred bytes of every row of the matrix.  SUB = rows of the tile a
// workgroup writes before moving to the next column tile of 64 (256: the K1' tile order; smaller: not used).
template <int RPW>
__global__ __launch_bounds__(512) void k_sweep(u16* __restrict__ p, int M, int V, long long pitch, int strips) {
    constexpr int LPR = 64 / RPW;
    constexpr int CW = LPR * 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cper = ((V + strips - 1) / strips + 7) / 8 * 8;
    const int c0 = slot * cper, c1 = min(V, c0 + cper);
    const int r = lane / LPR, cl = (lane % LPR) * 8;
    const int tiles = (M + 255) / 256;
    for (int rt = xcd; rt < tiles; rt += 8) {
        const int m0 = rt * 256 + wave * 32;
        for (int c = c0; c < c1; c += CW) {
#pragma unroll
            for (int rr = 0; rr < 32; rr += RPW) {
                const int m = m0 + rr + r, col = c + cl;
                if (m < M && col + 8 <= c1) *(u32x4h*)(p + (long long)m * pitch + col) = u32x4h{(unsigned)m, (unsigned)col, 1u, 2u};
            }
        }
    }
}
template <int RPW>
float run_sweep(u16* p, int M, int V, long long pitch, int strips) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k_sweep<RPW><<<8 * strips, 512>>>(p, M, V, pitch, strips);
    hipEventRecord(a);
    const int it = 20;
    for (int i = 0; i < it; ++i) k_sweep<RPW><<<8 * strips, 512>>>(p, M, V, pitch, strips);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / it * 1000.f;
}

// Rows per workgroup as the knob: a 256-thread workgroup owns ROWS rows (its 4 waves ROWS/4 each) and walks its column range
// in steps of one wave instruction per RPW rows; fewer rows per workgroup = fewer rows "open" device-wide.
// MODE 0: grid (row tiles, csplit); 1: grid (csplit, row tiles) -- the column ranges of one row tile get neighbouring
// workgroup ids (different XCDs); 2: as 0, odd column ranges sweep their columns in descending order
template <int RPW, int ROWS, int MODE = 0>
__global__ __launch_bounds__(256) void k_rows2(u16* __restrict__ p, int M, int V, long long pitch, int csplit) {
    constexpr int LPR = 64 / RPW;
    constexpr int CW = LPR * 8;
    constexpr int RW = ROWS / 4;                      // rows per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx = MODE == 1 ? blockIdx.y : blockIdx.x, by = MODE == 1 ? blockIdx.x : blockIdx.y;
    if (MODE == 3 || MODE == 4) {
        // K1' store form: every row's 128-byte piece is moved down to a 64-byte boundary (what the compute waves'
        // shift achieves); MODE 4: only two of the four waves store, 128 rows each (K1' has two store waves)
        const int nw = MODE == 4 ? 2 : 4;
        if (wave >= nw) return;
        const int rw = ROWS / nw;
        const int mm0 = bx * ROWS + wave * rw;
        const int cper = ((V + csplit - 1) / csplit + 63) / 64 * 64;
        const int c0 = by * cper, c1 = min(V, c0 + cper);
        const int r = lane / LPR, cl = (lane % LPR) * 8;
        for (int c = c0; c < c1; c += CW) {
            for (int rr = 0; rr < rw; rr += RPW) {
                const int m = mm0 + rr + r;
                const long long e = ((long long)m * pitch + c) & ~31ll;          // 64-byte aligned element offset
                if (m < M && c + CW <= c1) *(u32x4*)(p + e + cl) = u32x4{(unsigned)m, (unsigned)c, 1u, 2u};
            }
        }
        return;
    }
    const int m0 = bx * ROWS + wave * RW;
    const int cper = ((V + csplit - 1) / csplit + 7) / 8 * 8;
    const int c0 = by * cper, c1 = min(V, c0 + cper);
    if (MODE == 2 && (by & 1)) {
        const int r = lane / LPR, cl = (lane % LPR) * 8;
        for (int c = c0 + (c1 - c0 - 1) / CW * CW; c >= c0; c -= CW) {
#pragma unroll
            for (int rr = 0; rr < RW; rr += RPW) {
                const int m = m0 + rr + r, col = c + cl;
                if (m < M && col + 8 <= c1) *(u32x4h*)(p + (long long)m * pitch + col) = u32x4h{(unsigned)m, (unsigned)col, 1u, 2u};
            }
        }
        return;
    }
    const int r = lane / LPR, cl = (lane % LPR) * 8;
    for (int c = c0; c < c1; c += CW) {
#pragma unroll
        for (int rr = 0; rr < RW; rr += RPW) {
            const int m = m0 + rr + r, col = c + cl;
            if (m < M && col + 8 <= c1) *(u32x4h*)(p + (long long)m * pitch + col) = u32x4h{(unsigned)m, (unsigned)col, 1u, 2u};
        }
    }
}
template <int RPW, int ROWS, int MODE = 0>
float run_rows2(u16* p, int M, int V, long long pitch, int csplit) {
    dim3 grid((M + ROWS - 1) / ROWS, csplit);
    if (MODE == 1) grid = dim3(csplit, (M + ROWS - 1) / ROWS);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k_rows2<RPW, ROWS, MODE><<<grid, 256>>>(p, M, V, pitch, csplit);
    hipEventRecord(a);
    const int it = 20;
    for (int i = 0; i < it; ++i) k_rows2<RPW, ROWS, MODE><<<grid, 256>>>(p, M, V, pitch, csplit);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / it * 1000.f;
}

template <int RPW, bool STORE>
float run(u16* p, int M, int V, long long pitch, int csplit, unsigned* sink) {
    dim3 grid((M + 255) / 256, csplit);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k_rows<RPW, STORE><<<grid, 512>>>(p, M, V, pitch, csplit, sink);
    hipEventRecord(a);
    const int it = 20;
    for (int i = 0; i < it; ++i) k_rows<RPW, STORE><<<grid, 512>>>(p, M, V, pitch, csplit, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / it * 1000.f;
}

int main() {
    const int M = 32768;
    u16* p; unsigned* sink;
    hipMalloc(&p, (size_t)M * 9408 * 2 + 4096); hipMalloc(&sink, 4);
    hipMemset(p, 0, (size_t)M * 9408 * 2);
    {
        const double gb = (double)M * 9353 * 2 / 1e9;
        for (int strips = 8; strips <= 64; strips *= 2) {
            float t;
#define SW(RPW) t = run_sweep<RPW>(p, M, 9353, 9353, strips); \
            printf("sweep V=9353 strips/XCD=%2d store rows/wave-instr=%2d (%4d B/row): %7.1f us %7.0f GB/s\n", strips, RPW, 64 / RPW * 16, t, gb / t * 1e6);
            SW(8) SW(4) SW(2) SW(1)
        }
    }
    if (getenv("MATRIX")) {
        const int Vm = getenv("VM") ? atoi(getenv("VM")) : 9353;
        const double gb = (double)M * Vm * 2 / 1e9;
        const int css[6] = {1, 2, 3, 4, 6, 8};
        printf("# V=%d, 128 B per row and wave instruction, 256-thread workgroups; GB/s by rows per workgroup x column ranges\n", Vm);
        printf("%8s", "rows\\cs");
        for (int c = 0; c < 6; ++c) printf(" %7d", css[c]);
        printf("\n");
#define ROWLINE(ROWS) { printf("%8d", ROWS); for (int c = 0; c < 6; ++c) { float t = run_rows2<8, ROWS, 0>(p, M, Vm, Vm, css[c]); printf(" %7.0f", gb / t * 1e6); } printf("\n"); }
        ROWLINE(32) ROWLINE(64) ROWLINE(128) ROWLINE(256) ROWLINE(512)
        return 0;
    }
    if (getenv("K1B")) {
        const double gb = (double)M * 9353 * 2 / 1e9;
        float t;
#define KB(ROWS, MODE, CS) t = run_rows2<8, ROWS, MODE>(p, M, 9353, 9353, CS); \
        printf("rows/workgroup=%3d csplit=%d %s: %7.1f us %7.0f GB/s\n", ROWS, CS, MODE == 3 ? "64-byte-aligned pieces, 4 waves store" : MODE == 4 ? "64-byte-aligned pieces, 2 waves store" : "unaligned pieces", t, gb / t * 1e6);
        KB(256, 0, 2) KB(256, 3, 2) KB(256, 4, 2) KB(128, 0, 2) KB(128, 3, 2) KB(128, 4, 2) KB(128, 3, 4) KB(128, 4, 4) KB(256, 3, 1) KB(256, 4, 1)
        return 0;
    }
    if (getenv("MODES")) {
        const double gb = (double)M * 9353 * 2 / 1e9;
        for (int cs = 2; cs <= 4; cs *= 2) {
            float t;
#define RM(MODE) t = run_rows2<8, 256, MODE>(p, M, 9353, 9353, cs); \
            printf("rows/workgroup=256 csplit=%d mode=%d (0 grid (tiles, csplit), 1 grid (csplit, tiles), 2 odd ranges descending): %7.1f us %7.0f GB/s\n", cs, MODE, t, gb / t * 1e6);
            RM(0) RM(1) RM(2)
        }
        for (int cs = 3; cs <= 3; ++cs) {
            float t;
            RM(0) RM(1)
        }
        return 0;
    }
    if (getenv("ROWS2")) {
        const double gb = (double)M * 9353 * 2 / 1e9;
        for (int cs = 1; cs <= 4; cs *= 2) {
            float t;
#define R2(RPW, ROWS) t = run_rows2<RPW, ROWS>(p, M, 9353, 9353, cs); \
            printf("rows/workgroup=%3d csplit=%d store rows/wave-instr=%2d (%4d B/row): %7.1f us %7.0f GB/s\n", ROWS, cs, RPW, 64 / RPW * 16, t, gb / t * 1e6);
            R2(8, 32) R2(8, 64) R2(8, 128) R2(8, 256) R2(4, 16) R2(4, 32) R2(4, 64) R2(1, 4) R2(1, 8) R2(1, 16) R2(1, 32)
        }
        return 0;
    }
    if (getenv("SWEEP_ONLY")) return 0;
    const int Vs[3] = {9353, 9352, 9344};
    for (int vi = 0; vi < 3; ++vi) {
        const int V = Vs[vi];
        const double gb = (double)M * V * 2 / 1e9;
        for (int cs = 4; cs <= 16; cs *= 2) {
            float t;
#define ONE(RPW, ST) t = run<RPW, ST>(p, M, V, V, cs, sink); \
            printf("V=%d csplit=%2d %s rows/wave-instr=%2d (%4d B/row): %7.1f us %7.0f GB/s\n", V, cs, ST ? "store" : "load ", RPW, 64 / RPW * 16, t, gb / t * 1e6);
            ONE(16, true) ONE(8, true) ONE(4, true) ONE(2, true) ONE(1, true)
            ONE(16, false) ONE(8, false) ONE(4, false) ONE(2, false) ONE(1, false)
        }
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return 0;
}
