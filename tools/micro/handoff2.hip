// Model of a forward recurrence at the operating point the round-2 review asks about:
//   one 512-thread workgroup per CU, groups of G = 8 workgroups on one XCD, 64 sentences per group, W_hh in VGPRs
//   (every wave owns 32 gate columns for all sentences), h shared through LDS -- and TWO independent chains per
//   workgroup (sentence halves A and B, own counters and images) interleaved in one instruction stream, so that the
//   hand-off of one chain (signal -> visible -> LDS-direct ingest) runs under the other chain's MFMAs:
//     poll A | LDS-direct loads of h_A (sc1) | cell update + publish + drain + signal of B | wait A, barrier |
//     MFMAs of A (fragments from LDS) | poll B | loads of h_B | cell update + publish of A | wait B, barrier | MFMAs of B
// No model arithmetic: stand-in MFMAs (B operand in registers), fragment reads of the whole chain payload from LDS by
// every wave, stand-in cell updates; every word read from LDS is checked against the tag its producer wrote, so
// a hand-off that returns stale bytes (e.g. LDS-direct loads that ignore sc1) shows up as a non-zero error count.
// Prints microseconds per time step (both chains) from the slope between two launch lengths.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/handoff2 tools/micro/handoff2.hip && ./tools/micro/handoff2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) const void gmem_cv;
typedef __attribute__((address_space(3))) void lds_v;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Params {
    int G;          // workgroups per group
    int P;          // payload bytes of ONE chain per step (whole group's publishes)
    int steps;
    int nch;        // chains per workgroup: 1 or 2
    int n_mfma;     // stand-in MFMAs per wave and chain-step
    int n_cells;    // stand-in cell updates per lane and chain-step
    int frag_reads; // 1: every wave reads the whole chain payload from LDS and checks it
    int groups;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

__global__ __launch_bounds__(512) void k_handoff2(Params p, unsigned* image, unsigned* sync, unsigned* errs, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int group = xcd + 8 * (slot / p.G), member = slot % p.G;
    if (group >= p.groups) return;
    const unsigned pub = (unsigned)(p.P / p.G);            // bytes this workgroup publishes per chain-step
    const unsigned pubw = pub / 8;                         // ... per wave
    const size_t img_words = (size_t)p.P / 4;
    unsigned bad = 0;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cst = 0.25f + lane * 1e-3f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)1.0f; }
    const int share = p.P / 8;                             // bytes of a chain's payload each wave brings into LDS
    auto counter = [&](int ch) { return (gu32*)(sync + ((size_t)group * 2 + ch) * 32); };
    auto img = [&](int ch, int parity) { return image + (((size_t)group * 2 + ch) * 2 + parity) * img_words; };

    auto poll = [&](int ch, int step) {                    // step >= 1: the group's publishes of step - 1 are visible
        if (lane == 0) {
            const unsigned target = (unsigned)step * (unsigned)p.G;
            unsigned spins = 0;
            while (__hip_atomic_load(counter(ch), RLX_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { atomicAdd(errs + 1, 1u); break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto ingest = [&](int ch, int step) {                  // this wave's share of the chain's image -> LDS, sc1
        const unsigned char* src = (const unsigned char*)img(ch, (step - 1) & 1) + wave * share;
        unsigned char* dst = lds + ch * p.P + wave * share;
        for (int o = 0; o < share; o += 1024)
            __builtin_amdgcn_global_load_lds((gmem_cv*)(src + o + lane * 16), (lds_v*)(dst + o), 16, 0, 16 /* sc1 */);
    };
    auto mfma = [&](int ch, int step) {
        if (p.frag_reads && step > 0) {
            const unsigned want = (unsigned)step;
            unsigned x = 0;
            // eight 16-byte reads in flight per wave (inline asm: an ordinary LDS load after LDS-direct loads makes
            // hipcc wait for all of them; the barrier in front has done that already)
            const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(lds + ch * p.P) + lane * 16;
            for (int c = 0; c < p.P; c += 8192) {
                u32x4 f0, f1, f2, f3, f4, f5, f6, f7;
                const unsigned ad = base + c;
                asm volatile(
                    "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\t"
                    "ds_read_b128 %3, %8 offset:3072\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\t"
                    "ds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\ts_waitcnt lgkmcnt(0)"
                    : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3), "=&v"(f4), "=&v"(f5), "=&v"(f6), "=&v"(f7)
                    : "v"(ad));
                const u32x4 o = (f0 ^ want) | (f1 ^ want) | (f2 ^ want) | (f3 ^ want) | (f4 ^ want) | (f5 ^ want) | (f6 ^ want) | (f7 ^ want);
                x += (o[0] | o[1] | o[2] | o[3]) != 0;
            }
            bad += x;
        }
        for (int i = 0; i < p.n_mfma; i += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
        }
    };
    auto publish = [&](int ch, int step) {                 // cell updates, 16-byte sc1 stores of this wave's piece
        for (int i = 0; i < p.n_cells; ++i) {
            float z = cst + acc[i & 3][0] * 1e-30f;
#pragma unroll
            for (int k = 0; k < 5; ++k) z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44f * z)) + 0.1f * z;
            cst = z;
        }
        const __amdgpu_buffer_rsrc_t ws = rsrc(img(ch, step & 1), (unsigned)p.P);
        const unsigned tag = (unsigned)step + 1u;
        const u32x4 v = u32x4{tag, tag, tag, tag};
        const unsigned off = (unsigned)member * pub + (unsigned)wave * pubw + (unsigned)lane * 16u;
        __builtin_amdgcn_raw_buffer_store_b128(v, ws, (unsigned)lane * 16u < pubw ? off : 0x7ffffff0u, 0, 16 /* sc1 */);
    };
    auto signal = [&](int ch) {                            // every wave has drained its stores
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(counter(ch), 1u, RLX_AGENT);
    };
    auto landed = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    if (p.nch == 1) {
        for (int step = 0; step < p.steps; ++step) {
            if (step > 0) { poll(0, step); ingest(0, step); landed(); }
            mfma(0, step);
            publish(0, step);
            signal(0);
        }
    } else {
        // prologue: step 0 of both chains has no input
        mfma(0, 0);
        mfma(1, 0);
        publish(0, 0);
        signal(0);
        for (int step = 1; step <= p.steps; ++step) {
            // --- chain A's hand-off under chain B's tail
            if (step < p.steps) { poll(0, step); ingest(0, step); }
            publish(1, step - 1);
            signal(1);                                     // (drains A's LDS-direct loads as well: vmcnt is one counter)
            if (step < p.steps) {
                mfma(0, step);
                // --- chain B's hand-off under chain A's tail
                poll(1, step);
                ingest(1, step);
                publish(0, step);
                signal(0);
                mfma(1, step);
            }
        }
    }
    if (bad) atomicAdd(errs, bad);
    if (cst == 123.f) sink[0] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + cst;
}

struct Result { double us; unsigned errs, timeouts; };

static Result run(Params p, int cus, unsigned* image, unsigned* sync, unsigned* errs, float* sink) {
    p.groups = cus / p.G / 8 * 8;
    const int grid = p.groups * p.G;
    const size_t ldsb = 100 * 1024 > p.nch * p.P ? 100 * 1024 : (size_t)p.nch * p.P;      // > 80 KB: one workgroup per CU
    hipFuncSetAttribute((const void*)k_handoff2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double t[2];
    const int T[2] = {40, 200};
    Result r = {0, 0, 0};
    for (int rep = 0; rep < 3; ++rep)
        for (int i = 0; i < 2; ++i) {
            p.steps = T[i];
            hipMemsetAsync(sync, 0, 65536, 0);
            hipMemsetAsync(errs, 0, 16, 0);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_handoff2, dim3(grid), dim3(512), ldsb, 0, p, image, sync, errs, sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            t[i] = ms * 1e3;
            unsigned h[2];
            hipMemcpy(h, errs, 8, hipMemcpyDeviceToHost);
            r.errs += h[0]; r.timeouts += h[1];
        }
    r.us = (t[1] - t[0]) / (T[1] - T[0]);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return r;
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned *image, *sync, *errs;
    float* sink;
    hipMalloc(&image, (size_t)64 << 20);
    hipMalloc(&sync, 65536);
    hipMalloc(&errs, 16);
    hipMalloc(&sink, 16);
    hipMemset(image, 0, (size_t)64 << 20);
    printf("# CUs=%d; one 512-thread workgroup per CU; us per TIME STEP (all chains of a workgroup), slope of 40 vs 200 steps\n", cus);
    printf("# G=workgroups per group, P=payload KB of one chain per step, nch=chains per workgroup\n");
    printf("%3s %5s %4s %6s %6s %5s | %8s %6s %4s\n", "G", "P_KB", "nch", "mfma", "cells", "frag", "us/step", "errs", "t/o");
    auto line = [&](int G, int PKB, int nch, int n_mfma, int n_cells, int frag) {
        Params p;
        p.G = G; p.P = PKB * 1024; p.nch = nch; p.n_mfma = n_mfma; p.n_cells = n_cells; p.frag_reads = frag; p.steps = 0; p.groups = 0;
        const Result r = run(p, cus, image, sync, errs, sink);
        printf("%3d %5d %4d %6d %6d %5d | %8.2f %6u %4u\n", G, PKB, nch, n_mfma, n_cells, frag, r.us, r.errs, r.timeouts);
        fflush(stdout);
    };
    // forward recurrence, 64 sentences per group, 8 workgroups of 256 gate columns: h of a chain = 32 x 512 bf16 = 32 KB
    line(8, 64, 1, 0, 0, 0);        // one chain of 64 sentences, bare
    line(8, 64, 1, 0, 0, 1);        //   + fragment reads
    line(8, 64, 1, 128, 8, 1);      //   + work (128 MFMAs per wave, 8 cells per lane)
    line(8, 32, 2, 0, 0, 0);        // two chains of 32 sentences, bare
    line(8, 32, 2, 0, 0, 1);
    line(8, 32, 2, 64, 4, 1);       //   + work (64 MFMAs per wave and chain-step, 4 cells per lane)
    line(8, 32, 2, 96, 4, 1);       //   MFMA at 2/3 efficiency
    line(8, 32, 2, 128, 4, 1);      //   MFMA at half efficiency
    // 128 sentences per group (16 workgroups of 128 gate columns), two chains of 64
    line(16, 64, 2, 64, 4, 1);
    line(16, 64, 2, 0, 0, 0);
    return 0;
}
