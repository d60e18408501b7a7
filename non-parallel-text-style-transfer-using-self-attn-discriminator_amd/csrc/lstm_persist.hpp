// Persistent, weight-stationary BiLSTM recurrence (K2b): ONE launch runs all T time steps of both
// directions (reference: nn.LSTM over a packed sequence, transfer/models.py:779-785,904-908).
//
// Why: one launch per step pays a kernel boundary, a cold W_hh read and a G/Cs round trip per
// step; the steps are tiny (4.3 GFLOP) and strictly sequential, so those fixed costs dominate.
//
// Decomposition (B = 1024, H = 512: 8 x 2 x 16 = 256 workgroups = one per CU):
//   * a workgroup owns 128 sentences x 128 permuted gate columns (= 32 hidden units x 4 gates) of
//     one direction for the WHOLE sequence.  Its W_hh slice [128][Hp] bf16 (128 KB) is loaded into
//     LDS once; the cell state of its (sentence, unit) pairs never leaves registers.
//   * per step only h_{t-1} of its 128 sentences (all Hp units, written by the 16 workgroups of the
//     same sentence tile and direction = a "group") is read: 128 KB from L2 / fabric.
//   * the only synchronisation is inside a group (16 workgroups), never grid-wide; sentence tiles
//     advance independently, and a tile of short sentences simply stops early.
//
// Hand-off of h between workgroups INSIDE the launch follows cdna_hip_programming.md Guideline 16 /
// MI355X_MICROARCH.md "Valid forms" row 1, in the form that needs no cache-wide fence:
//   producer: every h store is an 8-byte agent-scope (sc1, write-through) store; every storing wave
//             drains (s_waitcnt vmcnt(0)); workgroup barrier; ONE lane adds 1 to the group counter
//             (agent-scope atomic);
//   consumer: ONE lane polls the counter (relaxed agent-scope = sc1 load) with s_sleep, bounded;
//             workgroup barrier; EVERY load of h is an 8-byte agent-scope (sc1) load to registers.
// No other load in the kernel touches bytes another workgroup writes during the launch (G tiles and
// Cs are private to the workgroup; W_hh and lens are read-only).
// Counters are zeroed by a hipMemsetAsync in front of every launch.  All workgroups must be
// co-resident: the host checks the grid against the occupancy query and otherwise runs the
// one-launch-per-step kernels.  Spins are bounded: a timeout sets the error word and ends waiting.
#pragma once
#include "kernels.hpp"

namespace sad {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
#define SAD_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int LP_BM = 128, LP_BN = 128, LP_KT_MAX = 8;       // tile; Hp <= 64 * LP_KT_MAX
constexpr int LP_SYNC_PITCH = 32;                            // counter pitch in 4-byte words (128-B lines)
constexpr unsigned LP_SPIN_LIMIT = 1u << 22;

__host__ __device__ inline size_t lstm_persist_lds_bytes(int Hp) { return (size_t)LP_BN * Hp * 2 + 2 * LP_BM * 64 * 2; }

// 16-byte chunk swizzle of the resident W image (rows of Hp bf16): conflict-free b128 fragment
// reads when a row has a multiple of 16 chunks, 4-way otherwise.
__device__ __forceinline__ int lp_wswz(int col, int chunk, int cmask) { return chunk ^ (col & cmask); }

__global__ __launch_bounds__(512) void k_lstm_fwd_persist(Shape s, const int* __restrict__ lens,
                                                         const bf16raw* __restrict__ whh, float* __restrict__ G,
                                                         float* __restrict__ Cs, bf16raw* Hs, unsigned* sync) {
    typedef Cfg<LP_BM, LP_BN, 4, 2, 1> C;                   // 8 waves: 32 rows x one 64-column gate block each
    extern __shared__ __attribute__((aligned(16))) unsigned char lp_lds[];
    const int Hp = s.Hp, Gp = s.Gp, B = s.B, T = s.T;
    bf16raw* Wl = (bf16raw*)lp_lds;                          // [128][Hp], chunk-swizzled
    bf16raw* As = (bf16raw*)(lp_lds + (size_t)LP_BN * Hp * 2);   // 2 x [128][64], gemm_core swizzle
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ctiles = Gp / LP_BN, rtiles = (B + LP_BM - 1) / LP_BM, groups = rtiles * 2;
    int group, ct;
    if (groups % 8 == 0) {        // blocks b, b+8, ... share an XCD (speed only): keep a group on one XCD
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        group = xcd + 8 * (slot / ctiles);
        ct = slot % ctiles;
    } else {
        group = blockIdx.x / ctiles;
        ct = blockIdx.x % ctiles;
    }
    const int dir = group & 1, b0 = (group >> 1) * LP_BM, n0 = ct * LP_BN;
    gu32* counter = (gu32*)(sync + (size_t)group * LP_SYNC_PITCH);
    gu32* errword = (gu32*)(sync + (size_t)groups * LP_SYNC_PITCH);
    const int cpr = Hp / 8;                                  // 16-byte chunks per W row
    const int cmask = (cpr % 16 == 0) ? 15 : 3;

    // ---- once: W_hh slice -> LDS; longest sentence of the tile ------------------------------------
    {
        const bf16raw* w = whh + ((long long)dir * Gp + n0) * Hp;
        for (int idx = tid; idx < LP_BN * cpr; idx += 512) {
            const int r = idx / cpr, c = idx - r * cpr;
            *(u32x4*)(Wl + (long long)r * Hp + lp_wswz(r, c, cmask) * 8) = *(const u32x4*)(w + (long long)r * Hp + c * 8);
        }
    }
    int maxlen = 0;
    for (int r = tid; r < LP_BM; r += 512) maxlen = (b0 + r < B) ? lens[b0 + r] : 0;
    {
        int* red_max = (int*)As;             // all LDS is dynamic (16-B aligned base); As is not in use yet
        if (tid == 0) *red_max = 0;
        __syncthreads();
        if (maxlen > 0) atomicMax(red_max, maxlen);
        __syncthreads();
        maxlen = *red_max;
    }
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int col0 = n0 + wn * 64;                           // first permuted column of this wave's unit block
    const int u4 = (lane >> 4) * 4;
    const int j = (col0 >> 6) * 16 + u4;                     // first of this lane's 4 hidden units
    int brow[C::TM], blen[C::TM];
    f32x4 creg[C::TM];
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
        brow[tm] = b0 + frag_row<C>(wm, tm, lane);
        blen[tm] = brow[tm] < B ? lens[brow[tm]] : 0;
        creg[tm] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // staging rows of this thread: chunk g = tid + i*512 -> row g/8, 16-byte chunk g%8 of a 64-wide k-tile
    int srow[2], slen[2];
    const int skc = (tid & 7) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        srow[i] = b0 + (tid >> 3) + i * 64;
        slen[i] = srow[i] < B ? lens[srow[i]] : 0;
    }
    const int nkt = (Hp + 63) / 64;
    unsigned done = 0;                                       // active steps this group has completed
    bool dead = false;                                       // a spin timed out: stop waiting

    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        if (!(maxlen > t)) continue;                         // group-uniform: no sentence of the tile is alive
        const int tp = dir ? t + 1 : t - 1;
        // gate pre-activations of this step: private to the workgroup, independent of h -> issued first
        f32x4 pre[C::TM][4];
        bool act[C::TM];
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) {
            act[tm] = blen[tm] > t;
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[tm][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (act[tm]) {
                const float* g = G + ((long long)t * B + brow[tm]) * s.Gld + dir * Gp + col0 + u4;
#pragma unroll
                for (int q = 0; q < 4; ++q) pre[tm][q] = *(const f32x4*)(g + 16 * q);
            }
        }
        f32x4 acc[C::TM][C::TN];
        zero_acc<C>(acc);
        if (done > 0) {
            // ---- wait for h_{tp} of the whole group -------------------------------------------------
            if (tid == 0 && !dead) {
                const unsigned target = done * (unsigned)ctiles;
                unsigned spins = 0;
                while (__hip_atomic_load(counter, SAD_RLX_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > LP_SPIN_LIMIT) {
                        __hip_atomic_store(errword, 1u, SAD_RLX_AGENT);
                        dead = true;
                        break;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no code motion across the poll
            __syncthreads();
            // ---- all of this tile's h_{tp} in flight at once (sc1 loads, 8 bytes each) ----------------
            unsigned long long a[LP_KT_MAX][2][2];
#pragma unroll
            for (int kt = 0; kt < LP_KT_MAX; ++kt)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[kt][i][0] = a[kt][i][1] = 0ull;
                    const int k = kt * 64 + skc;
                    if (slen[i] > tp && k < Hp) {
                        gu64* p = (gu64*)(Hs + (((long long)tp * B + srow[i]) * 2 + dir) * Hp + k);
                        a[kt][i][0] = __hip_atomic_load(p, SAD_RLX_AGENT);
                        a[kt][i][1] = __hip_atomic_load(p + 1, SAD_RLX_AGENT);
                    }
                }
            // ---- gates += h_{tp} . W_hh^T against the resident slice -----------------------------------
            // k-tile kt+1 is written into the other As buffer while kt is multiplied: one barrier per k-tile.
            auto stage = [&](int kt, const unsigned long long (&ak)[2][2]) {
                bf16raw* buf = As + (kt & 1) * (LP_BM * 64);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = (tid >> 3) + i * 64;
                    u32x4 v;
                    v[0] = (unsigned)ak[i][0];
                    v[1] = (unsigned)(ak[i][0] >> 32);
                    v[2] = (unsigned)ak[i][1];
                    v[3] = (unsigned)(ak[i][1] >> 32);
                    *(u32x4*)&buf[lds_off<C>(r, skc >> 3)] = v;
                }
            };
            __syncthreads();                                 // As free (previous step / prologue done with it)
            stage(0, a[0]);
            __syncthreads();
#pragma unroll
            for (int kt = 0; kt < LP_KT_MAX; ++kt) {
                if (kt < nkt) {
                    if (kt + 1 < LP_KT_MAX && kt + 1 < nkt) stage(kt + 1, a[kt + 1 < LP_KT_MAX ? kt + 1 : 0]);
                    const bf16raw* buf = As + (kt & 1) * (LP_BM * 64);
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        if (kt * 64 + kk * 32 < Hp) {
                            bf16x8 af[C::TM], bfr[C::TN];
                            const int ch = kk * 4 + (lane >> 4);
#pragma unroll
                            for (int tm = 0; tm < C::TM; ++tm)
                                af[tm] = *(const bf16x8*)&buf[lds_off<C>(frag_row<C>(wm, tm, lane), ch)];
#pragma unroll
                            for (int tn = 0; tn < C::TN; ++tn) {
                                const int col = wn * 64 + tn * 16 + (lane & 15);
                                bfr[tn] = *(const bf16x8*)(Wl + (long long)col * Hp + lp_wswz(col, kt * 8 + ch, cmask) * 8);
                            }
#pragma unroll
                            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                                for (int tn = 0; tn < C::TN; ++tn)
                                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[tn], af[tm], acc[tm][tn], 0, 0, 0);
                        }
                    }
                    __syncthreads();                         // kt consumed by all; kt+1 written by all
                }
            }
        }
        // ---- cell update; publish h_t first, the private stores (gates, c) after the signal -----------
        f32x4 gi[C::TM], gf[C::TM], gg[C::TM], go[C::TM];
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) {
            if (!act[tm]) continue;
            const f32x4 pi = acc[tm][0] + pre[tm][0], pf = acc[tm][1] + pre[tm][1];
            const f32x4 pg = acc[tm][2] + pre[tm][2], po = acc[tm][3] + pre[tm][3];
            const f32x4 cprev = creg[tm];     // still zero at a sentence's first step in this direction
            f32x4 c, h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gi[tm][e] = sigmoidf_(pi[e]);
                gf[tm][e] = sigmoidf_(pf[e]);
                gg[tm][e] = tanhf_(pg[e]);
                go[tm][e] = sigmoidf_(po[e]);
                c[e] = gf[tm][e] * cprev[e] + gi[tm][e] * gg[tm][e];
                h[e] = go[tm][e] * tanhf_(c[e]);
            }
            creg[tm] = c;
            const long long row = ((long long)t * B + brow[tm]) * 2 + dir;
            const u32x2 hp = pack_bf16x4(h);
            __hip_atomic_store((gu64*)&Hs[row * Hp + j], (unsigned long long)hp[0] | ((unsigned long long)hp[1] << 32),
                               SAD_RLX_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // EVERY storing wave drains its sc1 stores
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(counter, 1u, SAD_RLX_AGENT);
        ++done;
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) {
            if (!act[tm]) continue;
            float* g = G + ((long long)t * B + brow[tm]) * s.Gld + dir * Gp + col0 + u4;
            *(f32x4*)(g + 0) = gi[tm];
            *(f32x4*)(g + 16) = gf[tm];
            *(f32x4*)(g + 32) = gg[tm];
            *(f32x4*)(g + 48) = go[tm];
            *(f32x4*)&Cs[(((long long)t * B + brow[tm]) * 2 + dir) * Hp + j] = creg[tm];
        }
    }
}

}  // namespace sad
