// Floor of the in-launch hand-off chain of the persistent recurrences ("next round" item 2a).
// No arithmetic of the model: groups of G workgroups on one XCD repeat, per step,
//     poll (one lane, sc1) -> barrier -> sc1 loads of the group's payload image (P bytes, the whole group's
//     publishes of the step before) -> [optional LDS staging + fragment reads] -> [optional stand-in work:
//     MFMAs, cell-update transcendentals] -> 16-byte sc1 stores of this workgroup's P/G bytes -> drain ->
//     barrier -> counter add,
// i.e. the protocol of csrc/lstm_persist.hpp with the payload size, the group size, the loads in flight, the
// workgroups per CU and the staging form as knobs.  Every loaded word is checked against the step tag its
// producer wrote, so a protocol that returns stale bytes shows up as a non-zero error count.
// Prints microseconds per step from the slope between two launch lengths.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/handoff tools/micro/handoff.hip
//   ./tools/micro/handoff            (full sweep)      ./tools/micro/handoff quick
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr unsigned OOB = 0x7ffffff0u;

struct Params {
    int G;            // workgroups per group
    int P;            // payload bytes a workgroup ingests per step (= G * publish bytes per workgroup)
    int steps;
    int stage_lds;    // 0: loads stay in registers (each wave its share); 1: -> LDS, barrier, every wave reads all P bytes from LDS
    int n_mfma;       // stand-in MFMAs (16x16x32 bf16) per wave and step
    int n_cells;      // stand-in LSTM cell updates per lane and step (5 exp + 5 rcp + a few FMAs each)
    int per_wave_signal;  // 1: every storing wave adds to the counter after its own drain (no second barrier)
    int groups;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

template <int NF, int NT>
__global__ __launch_bounds__(NT) void k_handoff(Params p, unsigned* image, unsigned* sync, unsigned* errs, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = NT / 64;
    // blocks b, b + 8, ... share an XCD: keep a group on one XCD (as the product kernels do)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int group = xcd + 8 * (slot / p.G), member = slot % p.G;
    if (group >= p.groups) return;
    gu32* counter = (gu32*)(sync + (size_t)group * 32);
    const unsigned pub = (unsigned)(p.P / p.G);                    // bytes this workgroup publishes per step
    const size_t img_words = (size_t)p.P / 4;                      // one parity of one group's image
    unsigned* gimg = image + (size_t)group * 2 * img_words;
    const int chunks = p.P / 16;                                   // 16-byte chunks of the payload
    const int my_chunks = (int)pub / 16;
    const int storing_waves = (my_chunks + 63) / 64 < nwaves ? (my_chunks + 63) / 64 : nwaves;
    unsigned bad = 0;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cst = 0.25f + lane * 1e-3f;
    unsigned x = 0;

    for (int step = 0; step < p.steps; ++step) {
        if (step > 0) {
            if (tid == 0) {
                const unsigned target = (unsigned)step * (unsigned)(p.per_wave_signal ? p.G * storing_waves : p.G);
                unsigned spins = 0;
                while (__hip_atomic_load(counter, RLX_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) { atomicAdd(errs + 1, 1u); break; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __syncthreads();
            const __amdgpu_buffer_rsrc_t rs = rsrc(gimg + ((step - 1) & 1) * img_words, (unsigned)p.P);
            const unsigned want = (unsigned)step;                  // tag of the step before + 1
            // this lane's share: chunks tid, tid + NT, ...; NF of them in flight
            for (int c0 = tid; c0 < chunks; c0 += NT * NF) {
                u32x4 v[NF];
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const int c = c0 + i * NT;
                    v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, c < chunks ? (unsigned)c * 16u : OOB, 0, 16 /* sc1 */);
                }
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const int c = c0 + i * NT;
                    if (c < chunks) {
                        bad += (v[i][0] != want) + (v[i][1] != want) + (v[i][2] != want) + (v[i][3] != want);
                        if (p.stage_lds) *(u32x4*)(lds + (size_t)c * 16) = v[i];
                    }
                }
            }
            if (p.stage_lds) {
                __syncthreads();
                // every wave reads the whole payload as "fragments"
                for (int c = lane; c < chunks; c += 64) {
                    const u32x4 f = *(const u32x4*)(lds + (size_t)c * 16);
                    x ^= f[0] ^ f[3];
                }
            }
        }
        // stand-in work
        {
            bf16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)1.0f; }
            for (int i = 0; i < p.n_mfma; i += 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
            }
            for (int i = 0; i < p.n_cells; ++i) {
                float z = cst + acc[i & 3][0] * 1e-30f;
#pragma unroll
                for (int k = 0; k < 5; ++k) z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44f * z)) + 0.1f * z;
                cst = z;
            }
        }
        // publish: 16-byte sc1 stores of this workgroup's share, tagged with step + 1
        {
            const __amdgpu_buffer_rsrc_t ws = rsrc(gimg + (step & 1) * img_words, (unsigned)p.P);
            const unsigned tag = (unsigned)step + 1u;
            const u32x4 v = u32x4{tag, tag, tag, tag};
            for (int c = tid; c < my_chunks; c += NT)
                __builtin_amdgcn_raw_buffer_store_b128(v, ws, (unsigned)member * pub + (unsigned)c * 16u, 0, 16 /* sc1 */);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (p.per_wave_signal) {
            if (lane == 0 && wave < storing_waves) __hip_atomic_fetch_add(counter, 1u, RLX_AGENT);
        } else {
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(counter, 1u, RLX_AGENT);
        }
    }
    if (bad) atomicAdd(errs, bad);
    if (x == 0x12345678u || cst == 123.f) sink[0] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + cst;
}

typedef void (*kern_t)(Params, unsigned*, unsigned*, unsigned*, float*);
template <int NF, int NT> kern_t get() { return k_handoff<NF, NT>; }

static kern_t pick(int nf, int nt) {
    if (nt == 256) return nf == 2 ? get<2, 256>() : nf == 4 ? get<4, 256>() : nf == 8 ? get<8, 256>() : get<16, 256>();
    return nf == 2 ? get<2, 512>() : nf == 4 ? get<4, 512>() : nf == 8 ? get<8, 512>() : get<16, 512>();
}

struct Result { double us_per_step; unsigned errs, timeouts; };

static Result run(int G, int P, int nf, int nt, int per_cu, int stage, int n_mfma, int n_cells, int pws, unsigned* image,
                  unsigned* sync, unsigned* errs, float* sink, int cus) {
    Params p;
    p.G = G; p.P = P; p.stage_lds = stage; p.n_mfma = n_mfma; p.n_cells = n_cells; p.per_wave_signal = pws;
    const int slots = cus * per_cu;
    p.groups = slots / G / 8 * 8;                                  // whole groups, a multiple of 8 (one XCD each)
    if (p.groups < 8) p.groups = 8;
    const int grid = p.groups * G;
    // dynamic LDS pins the residency: > 80 KB -> one per CU, <= 80 KB and > 53 KB -> two per CU
    size_t ldsb = per_cu == 1 ? 100 * 1024 : 66 * 1024;
    if (stage && (size_t)P > ldsb) ldsb = P;
    kern_t k = pick(nf, nt);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k, nt, ldsb);
    Result r = {0, 0, 0};
    if (occ < per_cu || grid > occ * cus) { r.us_per_step = -1; return r; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double t[2];
    const int T[2] = {40, 200};
    for (int rep = 0; rep < 3; ++rep)
        for (int i = 0; i < 2; ++i) {
            p.steps = T[i];
            hipMemsetAsync(sync, 0, 4096 * 4, 0);
            hipMemsetAsync(errs, 0, 16, 0);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(nt), ldsb, 0, p, image, sync, errs, sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            t[i] = ms * 1e3;
            unsigned h[2];
            hipMemcpy(h, errs, 8, hipMemcpyDeviceToHost);
            r.errs += h[0]; r.timeouts += h[1];
        }
    r.us_per_step = (t[1] - t[0]) / (T[1] - T[0]);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return r;
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned *image, *sync, *errs;
    float* sink;
    hipMalloc(&image, (size_t)256 << 20);                          // groups * 2 * P <= 128 * 2 * 256 KB = 64 MB
    hipMalloc(&sync, 4096 * 4);
    hipMalloc(&errs, 16);
    hipMalloc(&sink, 16);
    hipMemset(image, 0, (size_t)256 << 20);
    printf("# CUs=%d; us per step = slope between launches of 40 and 200 steps (last of 3 repetitions)\n", cus);
    printf("# G=workgroups per group, P=payload KB ingested per workgroup and step, NT=threads, perCU=workgroups per CU,\n");
    printf("# NF=16-byte loads in flight per lane, lds=1: staged through LDS + every wave reads all of it, pws=per-wave signal\n");
    printf("%4s %5s %4s %5s %3s %3s %3s %6s %6s | %8s %6s %4s\n", "G", "P_KB", "NT", "perCU", "NF", "lds", "pws", "mfma", "cells", "us/step", "errs", "t/o");
    auto line = [&](int G, int PKB, int nt, int per_cu, int nf, int stage, int pws, int n_mfma, int n_cells) {
        if (PKB * 1024 / G < 16) return;
        const Result r = run(G, PKB * 1024, nf, nt, per_cu, stage, n_mfma, n_cells, pws, image, sync, errs, sink, cus);
        printf("%4d %5d %4d %5d %3d %3d %3d %6d %6d | %8.2f %6u %4u\n", G, PKB, nt, per_cu, nf, stage, pws, n_mfma, n_cells,
               r.us_per_step, r.errs, r.timeouts);
        fflush(stdout);
    };
    // (1) the bare chain: protocol only, registers
    const int Gs[4] = {4, 8, 16, 32}, Ps[5] = {4, 16, 64, 128, 256}, NFs[3] = {2, 4, 8};
    for (int per_cu = 1; per_cu <= 2; ++per_cu)
        for (int gi = 0; gi < 4; ++gi)
            for (int pi = 0; pi < 5; ++pi)
                for (int ni = 0; ni < 3; ++ni) {
                    if (quick && (ni != 2 || (pi != 2 && pi != 4))) continue;
                    line(Gs[gi], Ps[pi], 256, per_cu, NFs[ni], 0, 0, 0, 0);
                }
    // (2) 512-thread workgroups, one per CU
    for (int gi = 0; gi < 4; ++gi)
        for (int pi = 0; pi < 5; ++pi) line(Gs[gi], Ps[pi], 512, 1, 8, 0, 0, 0, 0);
    // (3) staging through LDS (payload shared by the waves of a workgroup) and per-wave signalling
    for (int per_cu = 1; per_cu <= 2; ++per_cu)
        for (int gi = 1; gi < 4; ++gi)
            for (int pi = 1; pi < 4; ++pi) {
                if (per_cu == 2 && Ps[pi] > 64) continue;
                line(Gs[gi], Ps[pi], 256, per_cu, 8, 1, 0, 0, 0);
                line(Gs[gi], Ps[pi], 256, per_cu, 16, 1, 0, 0, 0);
                line(Gs[gi], Ps[pi], 256, per_cu, 8, 1, 1, 0, 0);
            }
    // (4) candidate operating points with stand-in work (MFMAs per wave, cell updates per lane):
    //     current forward kernel: G=32, 128 KB, 2/CU, registers, 128 MFMA, 8 cells
    line(32, 128, 256, 2, 8, 0, 0, 128, 8);
    line(32, 128, 256, 2, 8, 0, 0, 0, 0);
    //     64-sentence tile, 128 gate columns per workgroup, W in registers, h shared through LDS, 2/CU
    line(16, 64, 256, 2, 8, 1, 0, 128, 8);
    line(16, 64, 256, 2, 16, 1, 0, 128, 8);
    line(16, 64, 256, 2, 16, 1, 1, 128, 8);
    line(16, 64, 256, 2, 16, 1, 0, 0, 0);
    //     32-sentence tile, 128 gate columns, 2/CU (needs 64 groups: more than the 512 slots / 16 -> uses what fits)
    line(16, 32, 256, 2, 8, 1, 0, 64, 4);
    //     64-sentence tile, 256 gate columns per workgroup, one 256-thread workgroup per CU (W in 256 VGPRs per wave)
    line(8, 64, 256, 1, 16, 1, 0, 256, 16);
    line(8, 64, 256, 1, 16, 1, 0, 0, 0);
    //     the same with 512 threads
    line(8, 64, 512, 1, 8, 1, 0, 128, 8);
    //     BPTT: current (128 sentences x 32 units, 512 KB, 512 threads, 1/CU), and 64-sentence tiles (256 KB)
    line(16, 512, 512, 1, 8, 0, 0, 64, 8);
    line(16, 512, 512, 1, 8, 0, 0, 0, 0);
    line(16, 256, 256, 2, 8, 0, 0, 128, 8);
    line(16, 256, 256, 2, 8, 0, 0, 0, 0);
    line(8, 256, 512, 1, 8, 0, 0, 128, 8);
    line(8, 256, 512, 1, 8, 0, 0, 0, 0);
    return 0;
}
