// Persistent, weight-stationary BiLSTM recurrence (K2b): ONE launch runs all T time steps of both
// directions (reference: nn.LSTM over a packed sequence, transfer/models.py:779-785,904-908).
//
// Why: one launch per step pays a kernel boundary, a cold W_hh read and a G/Cs round trip per
// step; the steps are tiny (4.3 GFLOP) and strictly sequential, so those fixed costs dominate.
//
// Decomposition: see the two kernels below (forward: 128 sentences x 16 hidden units per workgroup, two
// workgroups per CU; BPTT: 128 sentences x 32 units, one per CU).  The workgroups that share a
// sentence tile and a direction form a "group"; the only synchronisation is inside a group, never
// grid-wide; sentence tiles advance independently, and a tile of short sentences simply stops early.
//
// Hand-off of h between workgroups INSIDE the launch follows cdna_hip_programming.md Guideline 16 /
// MI355X_MICROARCH.md "Valid forms" row 1, in the form that needs no cache-wide fence:
//   producer: every h store is a 16-byte agent-scope (sc1, write-through) store; every storing wave
//             drains (s_waitcnt vmcnt(0)); workgroup barrier; ONE lane adds 1 to the group counter
//             (agent-scope atomic);
//   consumer: ONE lane polls the counter (relaxed agent-scope = sc1 load) with s_sleep, bounded;
//             workgroup barrier; EVERY load of h is a 16-byte agent-scope (sc1) buffer load to registers.
// No other load in the kernel touches bytes another workgroup writes during the launch (G tiles and
// Cs are private to the workgroup; W_hh and lens are read-only).
// Counters are zeroed in front of every launch by a kernel of the same call that runs earlier on the
// stream (k_prep_lists / k_prep in the forward, k_attn_bwd in the backward: a separate memset node cost 4.5 us).  All workgroups must be
// co-resident: the host checks the grid against the occupancy query and otherwise runs the
// one-launch-per-step kernels.  Spins are bounded: a timeout raises the error word (checked by
// k_attn_fwd, which then writes NaN logits / loss for the whole batch, and by the host helpers) and
// the workgroup publishes NaN from then on, so logits and dS cannot come out silently wrong.  No
// other kernel may occupy CUs of the device while a persistent launch is in flight.
#pragma once
#include "kernels.hpp"

namespace sad {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
// the workgroup's "a wait timed out" flag lives in LDS and is read through an LDS pointer: through a generic pointer it
// is a FLAT load, which counts as a vector-memory operation too -- every read of the flag was a wait for all of the
// wave's outstanding global loads and stores (in front of the poll, and in front of the hand-off stores)
typedef volatile __attribute__((address_space(3))) unsigned lds_flag;
#define SAD_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// 16-byte sc1 (L1-bypassing, agent-coherent) load through a buffer descriptor: visible to the
// compiler's waitcnt pass (counted vmcnt), and an offset beyond `bytes` returns zeros, which serves
// as a branch-free row predicate.
constexpr unsigned LP_OOB = 0x7ffffff0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t lp_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
// Lanes l and l+16 (quad-rows 2p, 2p+1 of the wave) exchange halves: an even quad-row ends up with
// {its own x, x of lane+16}, an odd one with {y of lane-16, its own y} (v_permlane16_swap: odd rows of
// the first operand <-> even rows of the second).  Must be executed by both partners.
__device__ __forceinline__ u32x4 lp_pair16(u32x2 x, u32x2 y) {
    const auto s0 = __builtin_amdgcn_permlane16_swap(x[0], y[0], false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(x[1], y[1], false, false);
    return u32x4{s0[0], s1[0], s0[1], s1[1]};
}
// 16-byte agent-scope (sc1, write-through) store of handed-off payload, as a buffer store the compiler can see; an
// offset beyond the descriptor's range drops the store (branch-free row predicate).  It used to be inline asm
// (`global_store_dwordx4 ... sc1`): the hazard recognizer and the waitcnt pass know nothing of inline asm, and in an
// experiment with two such stores back to back the compiler put the second one's operand moves right behind the
// first store, inside the window in which a 16-byte store still reads its registers -- stores went astray.
__device__ __forceinline__ void lp_store16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16 /* sc1 */);
}
__device__ __forceinline__ u32x4 lp_load16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16 /* sc1 */);
}

// 1: poll the group counter with a scalar load when the group sits on one XCD (see lp_counter).  Measured twice in
// round 3, correct both times and never faster than the vector poll beyond the noise (forward 231 us either way before
// the gate records were kept raw across the K loop, 203-206 against 205-207 us after; BPTT 330 -> 338 us).  Build
// switches, off: the vector poll does not depend on where the workgroups of a group are placed.
#ifndef SAD_SCALAR_POLL_FWD
#define SAD_SCALAR_POLL_FWD 0
#endif
#ifndef SAD_SCALAR_POLL_BWD
#define SAD_SCALAR_POLL_BWD 0
#endif
constexpr int LP_BM = 128;                                   // sentences per tile (= LF_BM = LB_BM)
constexpr int LP_SYNC_PITCH = 32;                            // counter pitch in 4-byte words (128-B lines)
constexpr int LP_BM_MIN = 32;                                // smallest sentence tile (BPTT at small batches)
// SAD_WS_SYNC: one counter line per (sentence tile, direction) at the smallest tile, then the line of the timeout words
__host__ __device__ inline int lp_err_line(int B) { return 2 * ((B + LP_BM_MIN - 1) / LP_BM_MIN); }
constexpr unsigned LP_SPIN_LIMIT = 1u << 22;                 // default bound of a hand-off wait (~seconds)
constexpr unsigned LP_NAN2 = 0x7fc07fc0u;                    // two bf16 quiet NaNs

// 16-byte chunk swizzle of the resident W image (rows of Hp bf16): conflict-free b128 fragment
// reads when a row has a multiple of 16 chunks, 4-way otherwise.
__device__ __forceinline__ int lp_wswz(int col, int chunk, int cmask) { return chunk ^ (col & cmask); }

// Bounded wait of ONE lane for `counter >= target`.  On a timeout the error word is raised and the
// workgroup's LDS flag `dead` is set: from then on the workgroup stops waiting and publishes NaN
// instead of results computed from stale hand-off data, so the failure reaches logits / loss / dS
// (a NaN h or dZ propagates through every later product) instead of passing silently.
// `hostflag` (or NULL): a word of host-mapped memory the library owns per device; raised too, so that the NEXT
// sad_forward on the device can refuse with the reason without anybody synchronising (sad_api.hip).
// `scalar` (the group sits on one XCD: its counter lives in that XCD's L2): the counter is read with a SCALAR load
// (glc: past the scalar cache).  A vector load's result is waited for with vmcnt, which retires in order: the poll
// of a step stood behind the polling wave's own plain stores of the step before and behind the private operand
// loads issued in front of it, whatever the counter said.
__device__ __forceinline__ unsigned lp_counter(gu32* counter, bool scalar) {
    if (scalar) {
        unsigned v;
        asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(counter) : "memory");
        return v;
    }
    return __hip_atomic_load(counter, SAD_RLX_AGENT);
}
__device__ __forceinline__ void lp_wait(gu32* counter, unsigned target, unsigned spin_limit, gu32* errword,
                                        lds_flag* dead, unsigned* hostflag, unsigned code, bool scalar = false) {
    if (*dead) return;
    unsigned spins = 0;
    while (lp_counter(counter, scalar) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > spin_limit) {
            __hip_atomic_store(errword, 1u, SAD_RLX_AGENT);
            if (hostflag != nullptr) __hip_atomic_store(hostflag, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            *dead = 1u;
            break;
        }
    }
}

}  // namespace sad

namespace sad {

// ------------------------------------------------------------------------------------------------
// K2b: HALF-WIDTH workgroups, TWO per CU.
// One step of the recurrence is a chain of waits (hand-off visible -> h loads -> MFMA -> cell update
// -> write-through stores drained -> signal); with one workgroup per CU the CU idles through most of
// it (13.4 us per step at cfg3 for 0.9 us of MFMA).  Here a workgroup has 256 threads and owns 128
// sentences x 64 permuted gate columns (16 units) of one direction; its W_hh slice is 64 KB, so two
// workgroups (of any two groups) share a CU and one computes while the other waits.
//   * wave w owns sentences 32w..32w+31 (two MFMA row fragments): B fragments from LDS feed two
//     MFMAs each, A fragments (h_{t-1}) come straight from global memory -- no LDS staging and no
//     barrier inside the K loop;
//   * h is handed over through an image in MFMA-fragment order (lstm_fwd_xchg_bytes), two step
//     parities, like dZ in the BPTT: the 16 lanes of a quarter wave read 256 contiguous bytes;
//   * the row-major Hs (hidden bridge, weight gradients), the activated gates and c are written with
//     plain stores after the signal.
// Hand-off protocol: the header of this file (16-byte sc1 stores, drain, barrier, counter add;
// one-lane bounded poll, barrier, 16-byte sc1 loads).
// ------------------------------------------------------------------------------------------------
#ifndef SAD_LF_NS
#define SAD_LF_NS 2
#endif
constexpr int LF_BM = 128, LF_BN = 64, LF_KS_MAX = 16, LF_NS = SAD_LF_NS;   // Hp <= 32 * LF_KS_MAX

// Sentence tile of the forward recurrence: the smallest of 32 / 64 / 128 whose grid still fits `slots` co-resident
// workgroups (two per CU), so that small batches spread over the CUs (see lstm_bwd_tile).
__host__ __device__ inline int lstm_fwd_tile(int B, int Gp, int slots) {
    const int ctiles = Gp / LF_BN;
    for (int bm = LP_BM_MIN; bm < LF_BM; bm *= 2)
        if (2 * ((B + bm - 1) / bm) * ctiles <= slots) return bm;
    return LF_BM;
}
__host__ __device__ inline size_t lstm_fwd2_lds_bytes(int Hp) { return (size_t)LF_BN * Hp * 2 + 16; }
__host__ __device__ inline size_t lstm_fwd_xchg_bytes(int B, int Hp) {
    return (size_t)2 * 2 * ((B + LF_BM - 1) / LF_BM * LF_BM) * Hp * 2;
}

#ifdef SAD_LF_PROF
// debug build only: phase timers (100 MHz wall clock) of one workgroup -> the free words after the timeout words
#define LF_PROF_DECL unsigned long long pt[7] = {0,0,0,0,0,0,0}, pa[7] = {0,0,0,0,0,0,0}; unsigned psteps = 0;
#define LF_PROF_T(i) pt[i] = wall_clock64();
#define LF_PROF_ACC { float tmp_; asm volatile("v_add_f32 %0, %1, %2" : "=v"(tmp_) : "v"(acc[0][0][0]), "v"(acc[TM - 1][3][0])); }
#define LF_PROF_END if (done > 1) { for (int i = 1; i < 7; ++i) pa[i] += pt[i] - pt[i - 1]; ++psteps; }
#define LF_PROF_OUT if (tid == 0 && group == SAD_LF_PROF && ct == 3) { unsigned long long* o = (unsigned long long*)(sync + (size_t)lp_err_line(s.B) * LP_SYNC_PITCH + 8); for (int i = 1; i < 7; ++i) o[i - 1] = pa[i]; o[6] = psteps; }
#else
#define LF_PROF_DECL
#define LF_PROF_T(i)
#define LF_PROF_ACC
#define LF_PROF_END
#define LF_PROF_OUT
#endif
// MS (tiles of 32 / 64 sentences, small batches): a wave takes ONE 16-sentence block instead of two, so a tile of 64
// keeps all four waves busy and a tile of 32 two of them (it ran on one: 128 MFMAs and two round trips of loads per step
// on one SIMD), and all of a block's h (16 KB at H = 512) is requested at once.  The sum over K keeps its order: the
// results are bit-identical to the full-tile form's.
template <bool MS>
__global__ __launch_bounds__(256, 2) void k_lstm_fwd_persist2(Shape s, const int* __restrict__ lens,
                                                          const bf16raw* __restrict__ whh, float* __restrict__ G,
                                                          float* __restrict__ Cs, bf16raw* __restrict__ Hs, bf16raw* HX,
                                                          unsigned short* __restrict__ GA, unsigned* sync,
                                                          unsigned spin_limit, int withhold, int bm, unsigned* hostflag,
                                                          int ga_pre) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lp_lds[];
    const int Hp = s.Hp, Gp = s.Gp, B = s.B, T = s.T;
    bf16raw* Wl = (bf16raw*)lp_lds;                          // [64][Hp], chunk-swizzled
    if ((int)blockIdx.x == withhold) return;                 // test hook: this workgroup never signals
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // bm: sentences per tile (LF_BM, or 64 / 32 at small batches: waves 32 w >= bm idle -- see lstm_fwd_tile)
    const int ctiles = Gp / LF_BN, rtiles = (B + bm - 1) / bm, groups = rtiles * 2;
    int group, ct;
    const bool xcd_local = (gridDim.x / ctiles) % 8 == 0;
    if (xcd_local) {
        // blocks b, b+8, ... share an XCD (speed only): keep a group on one XCD, so that its hand-offs are served
        // by that XCD's L2.  The host pads the grid to a multiple of 8 groups when that still fits the device;
        // the workgroups of the padding leave at once.
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        group = xcd + 8 * (slot / ctiles);
        ct = slot % ctiles;
        if (group >= groups) return;
    } else {
        group = blockIdx.x / ctiles;
        ct = blockIdx.x % ctiles;
    }
    const int dir = group & 1, b0 = (group >> 1) * bm, n0 = ct * LF_BN;
    gu32* counter = (gu32*)(sync + (size_t)group * LP_SYNC_PITCH);
    gu32* errword = (gu32*)(sync + (size_t)lp_err_line(B) * LP_SYNC_PITCH);
    const int cpr = Hp / 8;                                  // 16-byte chunks per W row
    const int cmask = (cpr % 16 == 0) ? 15 : 3;
    {
        // eight 16-byte loads per lane in flight (one load -> LDS store per iteration made the prologue a chain of
        // 16 global-load latencies, ~12 us of every launch)
        const bf16raw* w = whh + ((long long)dir * Gp + n0) * Hp;
        const int total = LF_BN * cpr;
        for (int base = 0; base < total; base += 256 * 8) {
            u32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = base + i * 256 + tid, ic = idx < total ? idx : 0;
                v[i] = *(const u32x4*)(w + (long long)ic * 8);            // rows are contiguous: element offset = idx * 8
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = base + i * 256 + tid;
                const int r = idx / cpr, c = idx - r * cpr;
                if (idx < total) *(u32x4*)(Wl + (long long)r * Hp + lp_wswz(r, c, cmask) * 8) = v[i];
            }
        }
    }
    int maxlen = 0;
    lds_flag* dead = (lds_flag*)(lp_lds + (size_t)LF_BN * Hp * 2 + 4);   // a wait timed out
    {
        int* red_max = (int*)(lp_lds + (size_t)LF_BN * Hp * 2);
        if (tid == 0) { *red_max = 0; *dead = 0u; }
        __syncthreads();
        if (tid < bm && b0 + tid < B) atomicMax(red_max, lens[b0 + tid]);
        __syncthreads();
        maxlen = *red_max;
    }
    const int q = lane >> 4, u4 = q * 4;
    const int j = ct * 16 + u4;                              // first of this lane's 4 hidden units
    const int sblocks = rtiles * (bm / 16);
    const size_t xpar = (size_t)2 * sblocks * Hp * 16;       // elements of one parity of the image
    constexpr int TM = MS ? 1 : 2;                           // 16-sentence blocks (MFMA row fragments) of a wave
    constexpr int NS = MS ? 2 * LF_NS : LF_NS;               // fragment sets in flight (a set: 4 k-steps of TM blocks)
    const int sblk = MS ? wave : wave * 2;                   // first block of this wave
    int brow[TM], blen[TM];
    unsigned xrow[TM];                                       // image offset (elements) of this lane's sentence, chunk 0
    f32x4 creg[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        brow[tm] = b0 + (sblk + tm) * 16 + (lane & 15);
        blen[tm] = ((sblk + tm) * 16 < bm && brow[tm] < B) ? lens[brow[tm]] : 0;
        creg[tm] = f32x4{0.f, 0.f, 0.f, 0.f};
        xrow[tm] = (unsigned)(((size_t)dir * sblocks + (b0 >> 4) + sblk + tm) * cpr * 128 + (lane & 15) * 8);
    }
    const int nks = Hp / 32;
    const int nsets = (nks + 3) / 4;
    const bool spoll = xcd_local && SAD_SCALAR_POLL_FWD;
    unsigned done = 0;
    LF_PROF_DECL

    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        if (!(maxlen > t)) continue;                         // group-uniform: no sentence of the tile is alive
        const int tp = dir ? t + 1 : t - 1;
        // accumulators start from the gate pre-activations of this step (private to the workgroup,
        // independent of h: issued before the wait)
        f32x4 acc[TM][4];
        u32x4 praw[TM][2];
        bool act[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            act[tm] = blen[tm] > t;
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[tm][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            praw[tm][0] = praw[tm][1] = u32x4{0u, 0u, 0u, 0u};
            if (act[tm]) {
                if (ga_pre) {
                    // the in-projection left the pre-activations as fp16 records in GA, in this lane's own record
                    // (the activated gates overwrite it below): 32 bytes instead of 64 of fp32 G.  Kept raw until
                    // the K loop is over: unpacking here is a wait for these loads in front of the hand-off wait
                    const unsigned short* ga = GA + ga_index(t, dir, brow[tm], j, B, Hp);
                    praw[tm][0] = *(const u32x4*)ga;
                    praw[tm][1] = *(const u32x4*)(ga + 8);
                } else {
                    const float* g = G + ((long long)t * B + brow[tm]) * s.Gld + dir * Gp + n0 + u4;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) acc[tm][gq] = *(const f32x4*)(g + 16 * gq);
                }
            }
        }
        LF_PROF_T(0)
        if (done > 0) {
            // vector poll: its result is waited for with vmcnt, in order behind the record loads anyway.  EVERY wave waits
            // for its records here, in front of the barrier (measured: with only the polling wave waiting, 236 us at cfg3
            // against 205 us)
            if (ga_pre && !spoll) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) lp_wait(counter, done * (unsigned)ctiles, spin_limit, errword, dead, hostflag, 1u, spoll);
            LF_PROF_T(1)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no code motion across the poll
            __syncthreads();
            LF_PROF_T(2)
            // ---- h_{tp} of this wave's 32 sentences: LF_NS sets of 4 k-steps (4 KB per wave each) in flight,
            //      16-byte sc1 loads straight into MFMA fragment registers; gates += h_{tp} . W_hh^T
            //      (a wave without sentences skips it)
            if (sblk * 16 < bm && b0 + sblk * 16 < B) {
            const __amdgpu_buffer_rsrc_t rs = lp_rsrc(HX + ((done - 1) & 1) * xpar, (unsigned)(xpar * 2));
            unsigned voff[TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) voff[tm] = (blen[tm] > tp && act[tm]) ? (xrow[tm] + (unsigned)q * 128u) * 2u : LP_OOB;
            u32x4 a[NS][TM][4];
            auto load_set = [&](auto slot, int sg) {
                constexpr int S = decltype(slot)::value;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
                        a[S][tm][kk] = lp_load16_sc1(rs, sg * 4 + kk < nks ? voff[tm] + (unsigned)(sg * 4 + kk) * 1024u : LP_OOB);
                    __builtin_amdgcn_sched_barrier(0);   // issue order = use order: the waits stay counted
                }
            };
            auto mma_set = [&](auto slot, int sg) {
                constexpr int S = decltype(slot)::value;
                if (sg >= nsets) return;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    {
                        const int ch = min((sg * 4 + kk) * 4 + q, cpr - 1);
                        bf16x8 bfr[4];
#pragma unroll
                        for (int tn = 0; tn < 4; ++tn) {
                            const int col = tn * 16 + (lane & 15);
                            bfr[tn] = *(const bf16x8*)(Wl + (long long)col * Hp + lp_wswz(col, ch, cmask) * 8);
                        }
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) {
                            const bf16x8 af = __builtin_bit_cast(bf16x8, a[S][tm][kk]);
#pragma unroll
                            for (int tn = 0; tn < 4; ++tn)
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[tn], af, acc[tm][tn], 0, 0, 0);
                        }
                    }
                }
            };
            // a real loop (not unrolled): straight-line code lets the scheduler hoist every LDS fragment
            // read to the top and spill
#pragma unroll
            for (int i = 0; i < NS; ++i) static_for<NS>([&](auto sl) { if (decltype(sl)::value == i) load_set(sl, i); });
#pragma nounroll
            for (int sg = 0; sg < nsets; sg += NS) {
                static_for<NS>([&](auto sl) {
                    constexpr int I = decltype(sl)::value;
                    mma_set(sl, sg + I);
                    load_set(sl, sg + NS + I);
                });
            }
            }
        }
        if (ga_pre) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                f32x4 p[4];
                ga_unpack(praw[tm][0], praw[tm][1], p[0], p[1], p[2], p[3]);
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[tm][g] += p[g];
            }
        }
        LF_PROF_ACC
        LF_PROF_T(3)
        // ---- cell update; publish h_t first, the private stores (Hs, gates, c) after the signal -------
        f32x4 gi[TM], gf[TM], gg[TM], go[TM];
        u32x2 hp[2];
        hp[1] = u32x2{0u, 0u};
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            hp[tm] = u32x2{0u, 0u};
            if (!act[tm]) continue;
            const f32x4 pi = acc[tm][0], pf = acc[tm][1], pg = acc[tm][2], po = acc[tm][3];
            const f32x4 cprev = creg[tm];     // still zero at a sentence's first step in this direction
            f32x4 c, h;
            if (s.cell == CELL_GRU) {
                // slots [r, z, n_input, n_hidden]; the carried state is the fp32 hidden state; the record keeps
                // [r, z, n, q = W_hn h + b_hn] for the BPTT (same layout as the per-step kernel k_lstm_fwd)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gi[tm][e] = sigmoidf_(pi[e]);
                    gf[tm][e] = sigmoidf_(pf[e]);
                    go[tm][e] = po[e];
                    gg[tm][e] = tanhf_(pg[e] + gi[tm][e] * po[e]);
                    c[e] = (1.f - gf[tm][e]) * gg[tm][e] + gf[tm][e] * cprev[e];
                    h[e] = c[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gi[tm][e] = sigmoidf_(pi[e]);
                    gf[tm][e] = sigmoidf_(pf[e]);
                    gg[tm][e] = tanhf_(pg[e]);
                    go[tm][e] = sigmoidf_(po[e]);
                    c[e] = gf[tm][e] * cprev[e] + gi[tm][e] * gg[tm][e];
                    h[e] = go[tm][e] * tanhf_(c[e]);
                }
            }
            creg[tm] = c;
            hp[tm] = pack_bf16x4(h);
        }
        // lanes 16 apart hold neighbouring unit quads of the same sentence: after the row swap an even
        // quad-row holds 8 consecutive units of row fragment 0, an odd one of row fragment 1
        u32x4 hv = lp_pair16(hp[0], hp[1]);
        if (*dead) hv = u32x4{LP_NAN2, LP_NAN2, LP_NAN2, LP_NAN2};   // timed out: never publish results of stale h
        const int tmq = q & 1;                               // MS: odd quad-rows hold the (absent) second block
        const bool actq = tmq ? (TM == 2 && act[TM - 1]) : act[0];
        lp_store16_sc1(lp_rsrc(HX + (done & 1) * xpar, (unsigned)(xpar * 2)),
                       actq ? ((tmq ? xrow[TM - 1] : xrow[0]) + (unsigned)(ct * 2 + (q >> 1)) * 128u) * 2u : LP_OOB, hv);
        LF_PROF_T(4)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // EVERY storing wave drains its sc1 stores
        LF_PROF_T(5)
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(counter, 1u, SAD_RLX_AGENT);
        ++done;
        if (actq) {
            const long long row = ((long long)t * B + (tmq ? brow[TM - 1] : brow[0])) * 2 + dir;
            *(u32x4*)&Hs[row * Hp + ct * 16 + (q >> 1) * 8] = hv;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            if (!act[tm]) continue;
            if (GA != nullptr) {
                ga_store(GA + ga_index(t, dir, brow[tm], j, B, Hp), gi[tm], gf[tm], gg[tm], go[tm]);
            } else {
                float* g = G + ((long long)t * B + brow[tm]) * s.Gld + dir * Gp + n0 + u4;
                *(f32x4*)(g + 0) = gi[tm];
                *(f32x4*)(g + 16) = gf[tm];
                *(f32x4*)(g + 32) = gg[tm];
                *(f32x4*)(g + 48) = go[tm];
            }
            *(f32x4*)&Cs[(((long long)t * B + brow[tm]) * 2 + dir) * Hp + j] = creg[tm];
        }
        LF_PROF_T(6)
        LF_PROF_END
    }
    LF_PROF_OUT
}

}  // namespace sad

namespace sad {

// ------------------------------------------------------------------------------------------------
// Persistent BPTT (K2'): ONE launch for all T steps of both directions, frozen weights.
//   dh_t = dZ_{next} . W_hh (+ d h_final at a sentence's last step), gate derivatives -> dZ_t.
// A workgroup owns 128 sentences x 32 hidden units of one direction; its W_hh^T slice [32][Gp] bf16
// (128 KB) stays in LDS, dc of its (sentence, unit) pairs stays in registers.  Per step it needs
// dZ_{next} of its 128 sentences over ALL Gp gate columns (written by the 16 workgroups of its
// group), 512 KB.  That operand never passes through LDS: wave w owns sentences 16w..16w+15 and
// loads their fragments straight from global memory (sc1) into registers in MFMA layout, three
// fragment sets in flight; there is no barrier inside the K loop, no cross-wave reduction (every
// dh is summed by one wave in a fixed order: results are reproducible bit for bit), and the
// accumulator layout (lane = sentence, 4 consecutive units) is exactly the epilogue's.
// Hand-off protocol: the header of this file, as in k_lstm_fwd_persist2.
// ------------------------------------------------------------------------------------------------
#ifndef SAD_LB_NS
#define SAD_LB_NS 2
#endif
#ifndef SAD_LB_ABL
#define SAD_LB_ABL 0      // timing-only ablation (debug builds): 1 no row-major dZ copy
#endif
constexpr int LB_BM = 128, LB_BN = 32, LB_SG_MAX = 16, LB_NS = SAD_LB_NS;       // Gp <= 128 * LB_SG_MAX

// Sentence tile of the BPTT.  A workgroup reads tile x Gp x 2 bytes of handed-off dZ per step at the ~70 GB/s a CU
// gets from sc1-stored payload (B = 64: 256 KB, 3.7 us of a 5.8 us step), and at B < 1024 the 128-sentence tiling
// leaves CUs empty: take the smallest tile whose grid still fits one workgroup per CU.
__host__ __device__ inline int lstm_bwd_tile(int B, int Hp, int cus) {
    const int ctiles = Hp / LB_BN;
    for (int bm = LP_BM_MIN; bm < LB_BM; bm *= 2)
        if (2 * ((B + bm - 1) / bm) * ctiles <= cus) return bm;
    return LB_BM;
}
__host__ __device__ inline size_t lstm_bwd_persist_lds_bytes(int Gp) { return (size_t)LB_BN * Gp * 2 + 16; }
// K-split form (tiles of 32 / 64 sentences): partial dh of up to 6 helper waves, 2 x f32x4 per lane
constexpr size_t LB_RED_BYTES = 6 * 2 * 64 * 16;
// wave-private strip of the row-major dZ copy: 16 sentences x (128 + 16) bytes
constexpr size_t LB_STRIP_BYTES = 16 * 144;
// Hand-off image of dZ between the workgroups of a group, two step parities: laid out so that the 16
// lanes of a quarter wave (16 sentences, one 8-column chunk = one MFMA fragment piece each) touch 256
// contiguous bytes -- fragments loaded straight from the row-major DZ touched 16 cache lines per
// quarter wave and ran at ~17 B/clk per CU.
__host__ __device__ inline size_t lstm_bwd_xchg_bytes(int B, int Gp) {
    return (size_t)2 * 2 * ((B + LB_BM - 1) / LB_BM * LB_BM) * Gp * 2;
}

#ifdef SAD_LB_PROF
// debug build only: phase timers (100 MHz wall clock) of one workgroup -> the free words after the timeout words
#define LB_PROF_ENTRY const unsigned long long p_entry = wall_clock64();
#define LB_PROF_DECL unsigned long long pt[7] = {0,0,0,0,0,0,0}, pa[7] = {0,0,0,0,0,0,0}, p_last6 = 0, p_gap = 0; unsigned psteps = 0; const unsigned long long p_loop = wall_clock64();
#define LB_PROF_T(i) pt[i] = wall_clock64();
#define LB_PROF_ACC { float tmp_; asm volatile("v_add_f32 %0, %1, %2" : "=v"(tmp_) : "v"(acc[0][0]), "v"(acc[1][0])); }
#define LB_PROF_END if (done > 1) { for (int i = 1; i < 7; ++i) pa[i] += pt[i] - pt[i - 1]; ++psteps; if (p_last6) p_gap += pt[0] - p_last6; } p_last6 = pt[6];
#define LB_PROF_OUT if (tid == 0 && group == SAD_LB_PROF && ct == 3) { unsigned long long* o = (unsigned long long*)(sync + (size_t)lp_err_line(s.B) * LP_SYNC_PITCH + 8); for (int i = 1; i < 7; ++i) o[i - 1] = pa[i]; o[6] = psteps; o[7] = p_loop - p_entry; o[8] = wall_clock64() - p_entry; o[9] = p_gap; }
#else
#define LB_PROF_ENTRY
#define LB_PROF_DECL
#define LB_PROF_T(i)
#define LB_PROF_ACC
#define LB_PROF_END
#define LB_PROF_OUT
#endif
// KS (tiles of 32 / 64 sentences, small batches): the tile has only sb = bm / 16 sentence blocks for the 8 waves, and one
// wave keeps just 8 KB of hand-off loads in flight -- a lone wave per block read its 64 KB per step as a chain of eight
// round trips.  Wave w takes block w % sb and part w / sb of ks = 8 / sb parts of the K range; the waves of part 0 own the
// block (operands, gate derivatives, stores), the others leave their partial dh in LDS, and the owner adds them in a fixed
// order (results stay reproducible bit for bit).  One more workgroup barrier per step.
// Canonical order of the sum over the gate columns, the same in every form of the kernel (so that a batch gives the same
// bits whatever tile its size selects): four parts with boundaries at whole rounds of LB_NS fragment sets, each summed
// front to back into its own accumulator, then (p0 + p1) + (p2 + p3).
__device__ __forceinline__ int lb_part(int p, int nsg) { return LB_NS * (p * ((nsg + LB_NS - 1) / LB_NS) / 4); }
template <bool KS>
__global__ __launch_bounds__(512) void k_lstm_bwd_persist(Shape s, const int* __restrict__ lens,
                                                         const bf16raw* __restrict__ whhT, const float* __restrict__ G,
                                                         const float* __restrict__ Cs, const float* __restrict__ DHFIN,
                                                         bf16raw* DZ, bf16raw* DZX, const unsigned short* __restrict__ GA, unsigned* sync,
                                                         unsigned spin_limit, int withhold,
                                                         const bf16raw* __restrict__ DHS, int ldh, int bm, unsigned* hostflag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lp_lds[];
    const int Hp = s.Hp, Gp = s.Gp, B = s.B, T = s.T, Ncat = s.Ncat;
    LB_PROF_ENTRY
    bf16raw* Wl = (bf16raw*)lp_lds;                                  // [32][Gp], chunk-swizzled
    if ((int)blockIdx.x == withhold) return;                         // test hook: this workgroup never signals
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // bm: sentences per tile (LB_BM, or 64 / 32 at small batches: waves 16 w >= bm idle -- see lstm_bwd_tile)
    const int ctiles = Hp / LB_BN, rtiles = (B + bm - 1) / bm, groups = rtiles * 2;
    int group, ct;
    const bool xcd_local = (gridDim.x / ctiles) % 8 == 0;
    if (xcd_local) {                            // blocks b, b+8, ... share an XCD (speed only; see the forward kernel)
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        group = xcd + 8 * (slot / ctiles);
        ct = slot % ctiles;
        if (group >= groups) return;
    } else {
        group = blockIdx.x / ctiles;
        ct = blockIdx.x % ctiles;
    }
    const int dir = group & 1, b0 = (group >> 1) * bm, j0 = ct * LB_BN;
    gu32* counter = (gu32*)(sync + (size_t)group * LP_SYNC_PITCH);
    gu32* errword = (gu32*)(sync + (size_t)lp_err_line(B) * LP_SYNC_PITCH + 1);
    const int cpr = Gp / 8;                                          // 16-byte chunks per W row (multiple of 16)

    // ---- once: W_hh^T slice -> LDS; longest sentence of the tile -----------------------------------
    {
        // eight 16-byte loads per lane in flight (see the forward kernel)
        const bf16raw* w = whhT + ((long long)dir * Hp + j0) * Gp;
        const int total = LB_BN * cpr;
        for (int base = 0; base < total; base += 512 * 8) {
            u32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = base + i * 512 + tid, ic = idx < total ? idx : 0;
                v[i] = *(const u32x4*)(w + (long long)ic * 8);            // rows are contiguous: element offset = idx * 8
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = base + i * 512 + tid;
                const int r = idx / cpr, c = idx - r * cpr;
                if (idx < total) *(u32x4*)(Wl + (long long)r * Gp + lp_wswz(r, c, 15) * 8) = v[i];
            }
        }
    }
    int maxlen = 0;
    lds_flag* dead = (lds_flag*)(lp_lds + (size_t)LB_BN * Gp * 2 + 4);   // a wait timed out
    {
        int* red_max = (int*)(lp_lds + (size_t)LB_BN * Gp * 2);
        if (tid == 0) { *red_max = 0; *dead = 0u; }
        __syncthreads();
        if (tid < bm && b0 + tid < B) atomicMax(red_max, lens[b0 + tid]);
        __syncthreads();
        maxlen = *red_max;
    }
    // this lane: sentence b0 + 16*sblk + (lane & 15); unit quads j0 + 16*tn + 4*(lane >> 4), tn = 0, 1
    const int sb = KS ? bm / 16 : 8, ks = 8 / sb;
    const int sblk = KS ? wave % sb : wave, kq = KS ? wave / sb : 0;
    const int b = b0 + sblk * 16 + (lane & 15);
    const int rlen = (sblk * 16 < bm && b < B) ? lens[b] : 0;   // K loop: every wave of the block
    const bool mine = kq == 0 && sblk * 16 < bm && b < B;        // owner of the sentence: operands, derivatives, stores
    const int len = mine ? rlen : 0;
    int len8[2];                                                 // the row-major copy's lane -> sentence mapping (8 lanes a row)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int b8 = b0 + sblk * 16 + 8 * i + (lane >> 3);
        len8[i] = (kq == 0 && sblk * 16 < bm && b8 < B) ? lens[b8] : 0;
    }
    int ju[2], pcu[2];
    f32x4 dcreg[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        ju[tn] = j0 + 16 * tn + 4 * (lane >> 4);
        pcu[tn] = (ju[tn] >> 4) * 64 + (ju[tn] & 15);
        dcreg[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nsg = Gp / 128;                                // fragment sets of 4 x 32 k
    // hand-off image of dZ (see lstm_bwd_xchg_bytes): [parity][dir][16-sentence block][8-column chunk][16][8]
    const int sblocks = rtiles * (bm / 16);
    const size_t xpar = (size_t)2 * sblocks * Gp * 16;       // elements of one parity
    const unsigned xrow = (unsigned)(((size_t)dir * sblocks + (b0 >> 4) + sblk) * (Gp / 8) * 128 + (lane & 15) * 8);
    // this wave's share of the K range (all of it unless KS): 4 / ks of the four parts
    const int sg0 = lb_part(kq * (4 / ks), nsg), sg1r = lb_part((kq + 1) * (4 / ks), nsg), sg1 = min(sg1r, nsg);
    unsigned done = 0;
    int t_carry = -1;                                        // step whose c is held in eccarry (per lane)
    f32x4 eccarry[2];
    eccarry[0] = eccarry[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    LB_PROF_DECL

    for (int step = 0; step < T; ++step) {
        const int t = dir ? step : T - 1 - step;             // reverse of the forward order
        if (!(maxlen > t)) continue;                         // group-uniform
        const int tn_ = dir ? t - 1 : t + 1;                 // step whose dZ feeds dh_t
        const int tpf = dir ? t + 1 : t - 1;                 // forward-previous step (source of c_prev)
        const bool act = len > t;
        // epilogue operands: private to the workgroup, independent of the hand-off -> issued first
        f32x4 egi[2], egf[2], egg[2], ego[2], ec[2], ecp[2], edhf[2];
        u32x4 graw[2][2];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            egi[tn] = egf[tn] = egg[tn] = ego[tn] = ec[tn] = ecp[tn] = edhf[tn] = z;
            graw[tn][0] = graw[tn][1] = u32x4{0u, 0u, 0u, 0u};
            if (!act) continue;
            if (GA != nullptr) {
                // raw fp16 records now, converted after the K loop: converting here would wait for the
                // loads in front of the poll (measured: 2 us per step)
                const unsigned short* ga = GA + ga_index(t, dir, b, ju[tn], B, Hp);
                graw[tn][0] = *(const u32x4*)ga;
                graw[tn][1] = *(const u32x4*)(ga + 8);
            } else {
                const float* g = G + ((long long)t * B + b) * s.Gld + dir * Gp + pcu[tn];
                egi[tn] = *(const f32x4*)(g + 0);
                egf[tn] = *(const f32x4*)(g + 16);
                egg[tn] = *(const f32x4*)(g + 32);
                ego[tn] = *(const f32x4*)(g + 48);
            }
            // c_t was this lane's c_prev of the step before (t == the tpf of that step): carried in registers
            if (t_carry == t) ec[tn] = eccarry[tn];
            else ec[tn] = *(const f32x4*)&Cs[(((long long)t * B + b) * 2 + dir) * Hp + ju[tn]];
            if (tpf >= 0 && tpf < T && len > tpf) ecp[tn] = *(const f32x4*)&Cs[(((long long)tpf * B + b) * 2 + dir) * Hp + ju[tn]];
            eccarry[tn] = ecp[tn];
            const bool last = dir ? (t == 0) : (t == len - 1);
            if (last) edhf[tn] = *(const f32x4*)&DHFIN[(long long)b * 2 * Hp + dir * Hp + ju[tn]];
            if (DHS != nullptr) {
                // gradient of this step's output from the RNN layer above: d(its input)[t][b][dir*H + unit]
                const bf16raw* q = DHS + ((long long)t * B + b) * ldh + dir * s.H + ju[tn];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ju[tn] + e < s.H) edhf[tn][e] += bf16_to_f32(q[e]);
            }
        }
        t_carry = (act && tpf >= 0 && tpf < T && len > tpf) ? tpf : -1;
        f32x4 acc[2];
        acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        LB_PROF_T(0)
        if (done > 0) {
            // (waiting here for the private operands issued above, as the forward does, changes nothing: 321 us either way)
            if (tid == 0) lp_wait(counter, done * (unsigned)ctiles, spin_limit, errword, dead, hostflag, 2u, xcd_local && SAD_SCALAR_POLL_BWD);
            LB_PROF_T(1)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __syncthreads();
            LB_PROF_T(2)
            // ---- dh = dZ_{tn} . W_hh for this wave's 16 sentences (a wave without sentences skips it) --
            if (sblk * 16 < bm && b0 + sblk * 16 < B) {
            const bool rowok = rlen > tn_ && rlen > t;
            const __amdgpu_buffer_rsrc_t rs = lp_rsrc(DZX + ((done - 1) & 1) * xpar, (unsigned)(xpar * 2));
            const unsigned voff = rowok ? (xrow + (unsigned)(lane >> 4) * 128u) * 2u : LP_OOB;
            auto load_set = [&](int sg, u32x4 (&f)[4]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) f[q] = lp_load16_sc1(rs, sg < sg1 ? voff + (unsigned)(sg * 16 + q * 4) * 256u : LP_OOB);
            };
            auto mma_set = [&](int sg, const u32x4 (&f)[4], f32x4 (&c)[2]) {
                if (sg >= sg1) return;
                // all eight W fragments of the set are requested from LDS before the first wait for dZ: one fragment
                // read per MFMA, issued behind the MFMA before it, left the LDS latency in the chain 128 times a step
                bf16x8 bfr[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = sg * 16 + q * 4 + (lane >> 4);
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        const int n = tn * 16 + (lane & 15);
                        bfr[q][tn] = *(const bf16x8*)(Wl + (long long)n * Gp + lp_wswz(n, ch, 15) * 8);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bf16x8 af = __builtin_bit_cast(bf16x8, f[q]);
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) c[tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[q][tn], af, c[tn], 0, 0, 0);
                }
            };
            // LB_NS fragment sets (LB_NS x 4 KB per wave) in flight: the loads come through the fabric
            // (sc1), so the bytes in flight per CU set the rate.  Sets beyond sg1 are OOB loads (zeros).
            // The K range is summed in FOUR parts, each into an accumulator of its own, in every form of the
            // kernel (see lb_part): the stream of loads runs on across the part boundaries.
            u32x4 f[LB_NS][4];
            f32x4 c[4][2];
#pragma unroll
            for (int i = 0; i < LB_NS; ++i) load_set(sg0 + i, f[i]);
            static_for<4>([&](auto pc) {
                constexpr int P = decltype(pc)::value;
                c[P][0] = c[P][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int lo = max(lb_part(P, nsg), sg0), hi = min(lb_part(P + 1, nsg), sg1r);
#pragma nounroll
                for (int sg = lo; sg < hi; sg += LB_NS) {
#pragma unroll
                    for (int i = 0; i < LB_NS; ++i) {
                        mma_set(sg + i, f[i], c[P]);
                        load_set(sg + LB_NS + i, f[i]);
                    }
                }
            });
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) acc[tn] = (c[0][tn] + c[1][tn]) + (c[2][tn] + c[3][tn]);
            }
            if (KS) {
                // partial dh of the helper waves -> the block's owner.  A wave's parts it did not run are zero, so with
                // two waves per block the owner holds p0 + p1 and receives p2 + p3, with four it holds p0 and receives
                // p1, p2, p3: the same tree either way
                f32x4* red = (f32x4*)(lp_lds + (size_t)LB_BN * Gp * 2 + 16);
                if (kq > 0) {
                    red[((wave - sb) * 2 + 0) * 64 + lane] = acc[0];
                    red[((wave - sb) * 2 + 1) * 64 + lane] = acc[1];
                }
                __syncthreads();
                if (kq == 0) {
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        const f32x4 r1 = red[((sblk) * 2 + tn) * 64 + lane];
                        if (ks == 2) {
                            acc[tn] = acc[tn] + r1;
                        } else {
                            const f32x4 r2 = red[((sb + sblk) * 2 + tn) * 64 + lane], r3 = red[((2 * sb + sblk) * 2 + tn) * 64 + lane];
                            acc[tn] = (acc[tn] + r1) + (r2 + r3);
                        }
                    }
                }
            }
        }
        LB_PROF_ACC
        LB_PROF_T(3)
        // ---- gate derivatives; publish dZ_t ----------------------------------------------------------
        u32x2 pk[2][4];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g) pk[tn][g] = u32x2{0u, 0u};
        if (act) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                if (GA != nullptr) ga_unpack(graw[tn][0], graw[tn][1], egi[tn], egf[tn], egg[tn], ego[tn]);
                const f32x4 dh = acc[tn] + edhf[tn];
                const f32x4 dcin = (tn_ >= 0 && tn_ < T && len > tn_) ? dcreg[tn] : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 zi, zf, zg, zo, dcn;
                if (s.cell == CELL_GRU) {
                    // record = [r, z, n, q], ecp = h_prev, the carried register holds dh * z (as in k_lstm_bwd)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dht = dh[e] + dcin[e];
                        const float r = egi[tn][e], z = egf[tn][e], n = egg[tn][e], q = ego[tn][e];
                        const float da = dht * (1.f - z) * (1.f - n * n);
                        zi[e] = da * q * r * (1.f - r);
                        zf[e] = dht * (ecp[tn][e] - n) * z * (1.f - z);
                        zg[e] = da;
                        zo[e] = da * r;
                        dcn[e] = dht * z;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float tc = tanhf_(ec[tn][e]);
                        const float d_o = dh[e] * tc;
                        const float dct = dcin[e] + dh[e] * ego[tn][e] * (1.f - tc * tc);
                        zi[e] = dct * egg[tn][e] * egi[tn][e] * (1.f - egi[tn][e]);
                        zf[e] = dct * ecp[tn][e] * egf[tn][e] * (1.f - egf[tn][e]);
                        zg[e] = dct * egi[tn][e] * (1.f - egg[tn][e] * egg[tn][e]);
                        zo[e] = d_o * ego[tn][e] * (1.f - ego[tn][e]);
                        dcn[e] = dct * egf[tn][e];
                    }
                }
                dcreg[tn] = dcn;
                const f32x4 zz[4] = {zi, zf, zg, zo};
#pragma unroll
                for (int g = 0; g < 4; ++g) pk[tn][g] = pack_bf16x4(zz[g]);
            }
        }
        // 16-byte pieces: lanes 16 apart hold neighbouring unit quads of the same sentence; after the row
        // swap an even quad-row holds 8 consecutive units of gate 2p, an odd one of gate 2p+1.  A wave's
        // store covers 4 consecutive chunks x 16 sentences = 1 KB of the hand-off image (sc1).
        u32x4 pv[2][2];
        const bool poisoned = *dead != 0u;                   // timed out: never publish results of stale dZ
        {
            const __amdgpu_buffer_rsrc_t xout = lp_rsrc(DZX + (done & 1) * xpar, (unsigned)(xpar * 2));
            const int q = lane >> 4;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    pv[tn][p2] = lp_pair16(pk[tn][2 * p2], pk[tn][2 * p2 + 1]);
                    if (poisoned) pv[tn][p2] = u32x4{LP_NAN2, LP_NAN2, LP_NAN2, LP_NAN2};
                    const int col = (pcu[tn] - 4 * q) + (2 * p2 + (q & 1)) * 16 + (q >> 1) * 8;
                    lp_store16_sc1(xout, act ? (xrow + (unsigned)(col >> 3) * 128u) * 2u : LP_OOB, pv[tn][p2]);
                }
        }
        LB_PROF_T(4)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // EVERY storing wave drains its sc1 stores
        LB_PROF_T(5)
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(counter, 1u, SAD_RLX_AGENT);
        ++done;
        const bool act_t_copy = act;
        if (KS && act_t_copy && SAD_LB_ABL != 1) {
            // small tiles: straight from the registers (few rows per workgroup; the LDS turn below costs more than it saves)
            bf16raw* zout = DZ + ((long long)t * B + b) * Ncat + (long long)dir * Gp;
            const int q = lane >> 4;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2)
                    *(u32x4*)&zout[(pcu[tn] - 4 * q) + (2 * p2 + (q & 1)) * 16 + (q >> 1) * 8] = pv[tn][p2];
        }
        if (!KS && SAD_LB_ABL != 1) {
            // row-major copy for the kernels after the recurrence: not part of the hand-off, plain stores.  A sentence's
            // 256 bytes sit in the registers of its four lanes, 64 bytes per (tn, p2): stored from there, a wave
            // instruction touches 16 rows x half a line, and these stores stand in the vector-memory queue in front of
            // the next step's operand loads (without the copy the launch runs 295 us instead of 327).  Through a
            // wave-private LDS strip (pitch 128 + 16 bytes: both sides conflict-free) the wave turns each tn half into
            // two stores of 8 rows x one whole line: 324 -> 312 us at cfg3.
            unsigned char* strip = lp_lds + (size_t)LB_BN * Gp * 2 + 16 + (size_t)wave * LB_STRIP_BYTES;
            const int q = lane >> 4, srow = lane & 15;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2)
                    *(u32x4*)(strip + srow * 144 + (p2 * 4 + (q & 1) * 2 + (q >> 1)) * 16) = pv[tn][p2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int s2 = 8 * i + (lane >> 3), pc = lane & 7;
                    const u32x4 v = *(const u32x4*)(strip + s2 * 144 + pc * 16);
                    if (len8[i] > t)
                        *(u32x4*)&DZ[((long long)t * B + b0 + sblk * 16 + s2) * Ncat + (long long)dir * Gp + ((j0 >> 4) + tn) * 64 + pc * 8] = v;
                }
            }
        }
        LB_PROF_T(6)
        LB_PROF_END
    }
    // dZ rows of padded steps (t >= len) are exact zeros: later kernels read every row of DZ.  Done
    // after the recurrence: tiles of short sentences have the most of these rows and finish their
    // steps first, so the stores no longer sit in front of the first hand-offs of the long tiles.
    if (mine) {
        for (int t = len; t < T; ++t) {
            bf16raw* zrow = DZ + ((long long)t * B + b) * Ncat + (long long)dir * Gp;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int g = 0; g < 4; ++g) *(u32x2*)&zrow[pcu[tn] + g * 16] = u32x2{0u, 0u};
        }
    }
    LB_PROF_OUT
}

}  // namespace sad
