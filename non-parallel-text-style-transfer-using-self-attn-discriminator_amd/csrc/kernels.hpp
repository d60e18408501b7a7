// Hand-written gfx950 kernels of the discriminator hot path.  Reference lines each kernel replaces
// (paths relative to the reference repo) are cited at the kernel.
//
// Buffers (all time-major, token row m = t*B + b; see DESIGN.md "Data layout in HBM"):
//   need [M]           u8    bit0: t < len[b] (LSTM-active), bit1: !mask'[b,t] (attention-active)
//   X    [M][Ep]       bf16  embedded tokens
//   G    [M][Gld]      fp32  [fwd Gp | rev Gp | 64 pad]: gate pre-activations (input part) -> overwritten by
//                            activated i,f,g,o
//   Cs   [M][2][Hp]    fp32  cell state after step t
//   Hs   [M][2][Hp]    bf16  hidden state after step t (MFMA operand of the next step)
//   Kk   [M][Hp]       fp32  attention keys/values  linear_emb(X)
//   DZ   [M][Ncat]     bf16  d(pre-activations): [fwd gates | rev gates | dKk]
//   DX   [M][Ep]       bf16  d(embedded tokens)
#pragma once
#include "gemm_core.hpp"

namespace sad {

enum { SRC_IDS = 0, SRC_F32 = 1, SRC_BF16 = 2 };

// ------------------------------------------------------------------------------------------------
// prep: clamp lengths, apply the all-masked-row guard IN PLACE on the caller's mask
// (transfer/models.py:865-867), derive the per-token need bits.
// ------------------------------------------------------------------------------------------------
__global__ void k_prep(Shape s, const int* __restrict__ lens_in, unsigned char* mask, long long msb, long long mst,
                       int guard, unsigned char* __restrict__ need, int* __restrict__ lens, unsigned* __restrict__ sync,
                       int sync_words) {
    // hand-off counters + timeout words of the persistent recurrence start from zero every call (a
    // separate memset node cost 4.5 us)
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < sync_words; i += blockDim.x) sync[i] = 0u;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= s.B) return;
    int len = lens_in[b];
    len = len < 0 ? 0 : (len > s.T ? s.T : len);
    lens[b] = len;
    unsigned char* mrow = mask + (long long)b * msb;
    // one pass over the mask row in blocks of 64 steps (independent loads, kept as bits); the guard
    // needs the whole row, so the need bytes of a block are written once the next one is known
    bool all = true;
    for (int t0 = 0; t0 < s.T; t0 += 64) {
        unsigned long long bits = 0;
        const int n = s.T - t0 < 64 ? s.T - t0 : 64;
#pragma unroll 8
        for (int i = 0; i < n; ++i) bits |= (unsigned long long)(mrow[(long long)(t0 + i) * mst] != 0) << i;
        all = all && (n == 64 ? ~bits == 0ull : bits == ((1ull << n) - 1ull));
    }
    const bool unmask1 = guard && s.T > 1 && all;        // all-masked row: the reference unmasks t = 1
    if (unmask1) mrow[mst] = 0;
    for (int t0 = 0; t0 < s.T; t0 += 64) {
        unsigned long long bits = 0;
        const int n = s.T - t0 < 64 ? s.T - t0 : 64;
#pragma unroll 8
        for (int i = 0; i < n; ++i) bits |= (unsigned long long)(mrow[(long long)(t0 + i) * mst] != 0) << i;
        if (unmask1 && t0 == 0) bits &= ~2ull;
#pragma unroll 8
        for (int i = 0; i < n; ++i) {
            const int t = t0 + i;
            need[(long long)t * s.B + b] = (unsigned char)((t < len ? 1 : 0) | (((bits >> i) & 1ull) ? 0 : 2));
        }
    }
}

template <class C>
__device__ __forceinline__ int tile_any_need(const unsigned char* __restrict__ need, int m0, int M) {
    int any = 0;
    for (int r = threadIdx.x; r < C::BM; r += C::NT) {
        const int m = m0 + r;
        if (m < M && need[m]) any = 1;
    }
    return __syncthreads_or(any);
}

// ------------------------------------------------------------------------------------------------
// K1  soft-embedding contraction  X[m][e] = sum_v S[m][v] * W_emb[v][e]   (transfer/models.py:898)
// S is read ONCE from HBM (fp32 or bf16, V-contiguous rows at arbitrary (t,b) strides, rows only
// 4-/2-byte aligned because V is odd) and converted to bf16 on the way into LDS.
// ------------------------------------------------------------------------------------------------
struct RawF32x8 {
    f32x4 lo, hi;
};
struct CvtF32x8 {
    __device__ __forceinline__ u32x4 operator()(const RawF32x8& r) const { return pack_bf16x8(r.lo, r.hi); }
};

template <int KIND>
struct SoftRow;   // loads 8 consecutive vocabulary entries of one token row
template <>
struct SoftRow<SRC_F32> {
    typedef RawF32x8 Raw;
    typedef CvtF32x8 Cvt;
    typedef float Elem;
    // full chunk, row may be invalid (p == nullptr -> zeros); branch-free
    static __device__ __forceinline__ Raw load_fast(const float* p, const float* dummy, int k) {
        const float* q = p != nullptr ? p + k : dummy;
        Raw r;
        r.lo = *(const f32x4u*)q;
        r.hi = *(const f32x4u*)(q + 4);
        if (p == nullptr) r.lo = r.hi = f32x4{0.f, 0.f, 0.f, 0.f};
        return r;
    }
    // unconditional load; the caller zeroes invalid rows when it USES the value (a select right after the load is a
    // use: in a loop unrolled over register sets hipcc collects such selects at the end of the body and drains every
    // load in flight in front of them)
    static __device__ __forceinline__ Raw load_raw(const float* q) {
        Raw r;
        r.lo = *(const f32x4u*)q;
        r.hi = *(const f32x4u*)(q + 4);
        return r;
    }
    static __device__ __forceinline__ Raw load(const float* p, int k, int V) {
        Raw r;
        if (p != nullptr && k + 8 <= V) {
            r.lo = *(const f32x4u*)(p + k);
            r.hi = *(const f32x4u*)(p + k + 4);
        } else {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = (p != nullptr && k + j < V) ? p[k + j] : 0.f;
            r.lo = f32x4{t[0], t[1], t[2], t[3]};
            r.hi = f32x4{t[4], t[5], t[6], t[7]};
        }
        return r;
    }
};
template <>
struct SoftRow<SRC_BF16> {
    typedef u32x4 Raw;
    typedef Ident Cvt;
    typedef bf16raw Elem;
    static __device__ __forceinline__ Raw load_fast(const bf16raw* p, const bf16raw* dummy, int k) {
        const u32x4 v = *(const u32x4h*)(p != nullptr ? p + k : dummy);
        return p != nullptr ? v : u32x4{0u, 0u, 0u, 0u};
    }
    static __device__ __forceinline__ Raw load_raw(const bf16raw* q) { return *(const u32x4h*)q; }   // (non-temporal: K1 166 -> 182 us)
    static __device__ __forceinline__ Raw load(const bf16raw* p, int k, int V) {
        if (p != nullptr && k + 8 <= V) return *(const u32x4h*)(p + k);
        unsigned t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = (p != nullptr && k + j < V) ? (unsigned)p[k + j] : 0u;
        return u32x4{t[0] | (t[1] << 16), t[2] | (t[3] << 16), t[4] | (t[5] << 16), t[6] | (t[7] << 16)};
    }
};

template <class C, int KIND>
__global__ __launch_bounds__(C::NT) void k_softemb_fwd(Shape s, const void* __restrict__ src, long long sT, long long sB,
                                                      const unsigned char* __restrict__ need,
                                                      const bf16raw* __restrict__ wembT, bf16raw* __restrict__ X,
                                                      float* __restrict__ accf, int ksplit) {
    __shared__ __attribute__((aligned(16))) bf16raw As[C::BM * C::LDT];
    __shared__ __attribute__((aligned(16))) bf16raw Bs[C::BN * C::LDT];
    typedef SoftRow<KIND> SR;
    typedef typename SR::Elem Elem;
    const int m0 = blockIdx.x * C::BM, n0 = blockIdx.y * C::BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[C::TM][C::TN];
    zero_acc<C>(acc);
    if (tile_any_need<C>(need, m0, s.M)) {
        const Elem* arow[C::ACH];
#pragma unroll
        for (int i = 0; i < C::ACH; ++i) {
            const int m = m0 + tid / C::CH + i * C::RSTEP;
            arow[i] = nullptr;
            if (m < s.M) {
                const int t = m / s.B, b = m - t * s.B;
                arow[i] = (const Elem*)src + (long long)t * sT + (long long)b * sB;
            }
        }
        const int V = s.V, Vp = s.Vp, Ep = s.Ep;
        // split-K (small batches: few row tiles, K = V is long): blockIdx.z owns whole K-tiles
        const int tps = ceil_div(ceil_div(V, C::BK), ksplit);
        const int k0 = blockIdx.z * tps * C::BK;
        const int k1 = k0 + tps * C::BK < V ? k0 + tps * C::BK : V;
        const int Vmain = V / C::BK * C::BK;     // whole K-tiles: unguarded, branch-free loads
        const int kme = k1 < Vmain ? k1 : Vmain; // end of this split's whole tiles
        const Elem* dummy = (const Elem*)src;
        if (kme > k0)
            gemm_core<C, typename SR::Raw>(
                kme - k0, [&](int i, int, int k) { return SR::load_fast(arow[i], dummy, k0 + k); }, typename SR::Cvt(),
                [&](int, int r, int k) { return load_packed(wembT, Vp, n0 + r, Ep, k0 + k, Vp); }, acc, As, Bs);
        if (k1 > Vmain && k1 > k0)               // ragged last tile (V is odd): element-guarded loads
            gemm_core<C, typename SR::Raw>(
                V - Vmain, [&](int i, int, int k) { return SR::load(arow[i], Vmain + k, V); }, typename SR::Cvt(),
                [&](int, int r, int k) { return load_packed(wembT, Vp, n0 + r, Ep, Vmain + k, Vp); }, acc, As, Bs);
    }
    const int wm = wave / C::WN, wn = wave % C::WN;
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
        const int m = m0 + frag_row<C>(wm, tm, lane);
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            const int n = n0 + frag_col<C>(wn, tn, lane);
            if (m < s.M && n < s.Ep) {
                if (ksplit > 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(&accf[(long long)m * s.Ep + n + e], acc[tm][tn][e]);
                } else {
                    *(u32x2*)&X[(long long)m * s.Ep + n] = pack_bf16x4(acc[tm][tn]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1, three wave roles (large batches, Ep = 320): the two operand streams have different latencies
// (S comes from HBM, ~2.3 us under load; W_emb^T from L2), and vmcnt retires in order inside a wave,
// so one loader wave cannot keep different depths in flight for them.
//   waves 0-7   compute: fragments from LDS, 40 MFMAs per 64-deep k-tile, no vector-memory
//               instruction until the epilogue;
//   waves 8-11  S loaders: register-staged (dtype conversion, 2-/4-byte aligned rows), D k-tiles of
//               loads in flight (4 for bf16 S, 2 for fp32 S), two LDS stages;
//   waves 12-15 W loaders: LDS-direct loads (global_load_lds_dwordx4, no registers), three LDS
//               stages, two k-tiles in flight, own vmcnt counting and raw barriers.  The LDS image is
//               lane-linear per instruction (8 rows x 128 B); the chunk swizzle is applied to the
//               SOURCE address.
// One workgroup barrier per k-tile, 1 + nk barriers in every role.
// Measured (cfg3, tools/k1prof.py): the W loaders spend 0.8 us per k-tile inside the issue of their
// LDS-direct loads whether 2 or 4 waves share them -- the CU takes in its 56 KB per k-tile at
// ~50-70 GB/s (the per-CU L2 read rate), which is what bounds this kernel now (compute: 0.47 us).
// ------------------------------------------------------------------------------------------------
#ifdef SAD_K1F_PROF
// debug build only: phase timers of one workgroup, printed by one lane per role
#define K1P_DECL long long pt_[4] = {0, 0, 0, 0}, pa_[4] = {0, 0, 0, 0}; int pn_ = 0; const long long t_begin_ = wall_clock64();
#define K1P_T(i) pt_[i] = wall_clock64();
#define K1P_ACC { for (int i_ = 1; i_ < 4; ++i_) pa_[i_] += pt_[i_] - pt_[i_ - 1]; ++pn_; }
#define K1P_MFMA { float t_; asm volatile("v_add_f32 %0, %1, %2" : "=v"(t_) : "v"(acc[0][0][0]), "v"(acc[C::TM - 1][C::TN - 1][0])); }
#define K1P_OUT(what) if (blockIdx.x == SAD_K1F_PROF && lane == 0 && (wave == 0 || wave == 8 || wave == 12)) printf("%s: n=%d  %.3f %.3f %.3f us per k-tile, role total %.1f us\n", what, pn_, pa_[1] / 100.0 / pn_, pa_[2] / 100.0 / pn_, pa_[3] / 100.0 / pn_, (wall_clock64() - t_begin_) / 100.0);
#define K1P_OUT2(what) if (blockIdx.x == SAD_K1F_PROF && blockIdx.y == 1 && lane == 0 && (wave == 0 || wave == 8 || wave == 10)) printf("%s: n=%d  %.3f %.3f %.3f us per tile, role total %.1f us\n", what, pn_, pa_[1] / 100.0 / pn_, pa_[2] / 100.0 / pn_, pa_[3] / 100.0 / pn_, (wall_clock64() - t_begin_) / 100.0);
#else
#define K1P_DECL
#define K1P_T(i)
#define K1P_ACC
#define K1P_MFMA
#define K1P_OUT(what)
#define K1P_OUT2(what)
#endif
#ifndef SAD_K1F_D
#define SAD_K1F_D 4
#endif
typedef __attribute__((address_space(1))) const void gmem_cv;
typedef __attribute__((address_space(3))) void lds_v;

template <int KIND>
__global__ __launch_bounds__(1024) void k_softemb_fwd_ws3(Shape s, const void* __restrict__ src, long long sT, long long sB,
                                                        const unsigned char* __restrict__ need,
                                                        const bf16raw* __restrict__ wembT, bf16raw* __restrict__ X) {
    typedef Cfg<128, 320, 2, 4, 1> C;
    typedef SoftRow<KIND> SR;
    typedef typename SR::Elem Elem;
    constexpr int ABYTES = 128 * 128, BBYTES = 320 * 128, BOFF = 2 * ABYTES;   // LDS: A0 A1 B0 B1 B2
    constexpr int D = sizeof(typename SR::Raw) > 16 ? 2 : SAD_K1F_D;
    extern __shared__ __attribute__((aligned(16))) unsigned char fw_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 128;
    const int M = s.M, V = s.V, Vp = s.Vp;
    {   // tiles nobody reads are skipped (every wave takes the same decision)
        int any = 0;
        for (int r = tid; r < 128; r += 1024)
            if (m0 + r < M && need[m0 + r]) any = 1;
        if (!__syncthreads_or(any)) return;
    }
    const int nk = (V + 63) / 64;
    if (wave >= 12) {
        // ---------------------------------- W loader waves -------------------------------------
        const int ww = wave - 12, r = lane >> 3, sl = lane & 7;
        K1P_DECL
        // instruction i covers tile rows R0 = (10 ww + i) * 8 .. + 7; lane -> row R0 + r, LDS slot sl,
        // which holds global chunk sl ^ ((row >> 1) & 7); (R0 >> 1) & 7 is 0 for even i, 4 for odd i
        const bf16raw* be = wembT + (long long)(ww * 80 + r) * Vp + ((sl ^ (r >> 1)) << 3);
        const bf16raw* bo = wembT + (long long)(ww * 80 + 8 + r) * Vp + ((sl ^ (4 + (r >> 1))) << 3);
        const long long step2 = 16ll * Vp;
        auto issue = [&](int kt, int st) {
            unsigned char* dst = fw_lds + BOFF + st * BBYTES + ww * (80 * 128);
            const int k0 = kt * 64;
#pragma unroll
            for (int y = 0; y < 5; ++y) {
                __builtin_amdgcn_global_load_lds((gmem_cv*)(be + y * step2 + k0), (lds_v*)(dst + (2 * y) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gmem_cv*)(bo + y * step2 + k0), (lds_v*)(dst + (2 * y + 1) * 1024), 16, 0, 0);
            }
        };
        issue(0, 0);
        if (nk > 1) {
            issue(1, 1);
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                          // k-tile 0 ready
        int st2 = 2;                                           // stage of k-tile kt + 2
        for (int kt = 0; kt < nk; ++kt) {
            // stage (kt + 2) % 3 was last read in iteration kt - 1: free
            K1P_T(0)
            if (kt + 2 < nk) {
                issue(kt + 2, st2);
                K1P_T(1)
                asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // k-tile kt + 1 has landed
            } else {
                K1P_T(1)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            K1P_T(2)
            st2 = st2 == 2 ? 0 : st2 + 1;
            __builtin_amdgcn_s_barrier();
            K1P_T(3)
            K1P_ACC
        }
        K1P_OUT("W loader: issue, wait, barrier")
        return;
    }
    if (wave >= 8) {
        // ---------------------------------- S loader waves -------------------------------------
        const int lt = tid - 512;                              // 0..255: tile row lt/8 + 32 i, 16-byte chunk lt%8
        K1P_DECL
        const int r0 = lt >> 3, c0 = lt & 7;
        const Elem* arow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + r0 + 32 * i;
            arow[i] = nullptr;
            if (m < M && need[m]) {
                const int t = m / s.B, b = m - t * s.B;
                arow[i] = (const Elem*)src + (long long)t * sT + (long long)b * sB + c0 * 8;
            }
        }
        const int aoff0 = lds_off<C>(r0, c0) * 2;              // row + 32 i keeps the swizzle: + i * 4096 bytes
        typename SR::Raw ra[D][4];
        const Elem* aeff[4];                                   // rows nobody needs load row 0 and are zeroed in to_lds
#pragma unroll
        for (int i = 0; i < 4; ++i) aeff[i] = arow[i] != nullptr ? arow[i] : (const Elem*)src;
        const int nkf = V / 64;                                // whole k-tiles: unguarded, branch-free loads
        auto load_tile = [&](auto par, int kt) {               // clamped, never skipped: static operation counts
            constexpr int P = decltype(par)::value;
            const int k0 = (kt < nkf ? kt : nkf - 1) * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[P][i] = SR::load_raw(aeff[i] + k0);
        };
        auto to_lds = [&](auto par, int kt) {
            constexpr int P = decltype(par)::value;
            unsigned char* buf = fw_lds + (kt & 1) * ABYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 v = typename SR::Cvt()(ra[P][i]);
                *(u32x4*)&buf[aoff0 + i * 4096] = arow[i] != nullptr ? v : u32x4{0u, 0u, 0u, 0u};
            }
        };
        auto tail_to_lds = [&](int kt) {                       // the ragged last k-tile: element-guarded loads
            unsigned char* buf = fw_lds + (kt & 1) * ABYTES;
            const int k0 = kt * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *(u32x4*)&buf[aoff0 + i * 4096] =
                    typename SR::Cvt()(SR::load(arow[i] != nullptr ? arow[i] - c0 * 8 : nullptr, k0 + c0 * 8, V));
        };
        int kt = 0;
        if (nkf >= 1) {
            static_for<D>([&](auto d) { load_tile(d, decltype(d)::value); });
            to_lds(std::integral_constant<int, 0>{}, 0);
            load_tile(std::integral_constant<int, 0>{}, D);
        } else {
            tail_to_lds(0);
        }
        __syncthreads();                                       // k-tile 0 ready
        // steady state, branch-free: iteration j writes whole k-tile j + 1 (set (j+1) % D) and requests j + 1 + D
        for (; kt + D < nkf; kt += D) {
            static_for<D>([&](auto u) {
                constexpr int U = decltype(u)::value;
                using P = std::integral_constant<int, (U + 1) % D>;
                K1P_T(0)
                to_lds(P{}, kt + U + 1);
                K1P_T(1)
                load_tile(P{}, kt + U + 1 + D);
                K1P_T(2)
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);             // keeps the next tile's zeroing selects (uses of loads in flight) below
                K1P_T(3)
                K1P_ACC
            });
        }
        for (int j = kt; j < nk; ++j) {
            if (j + 1 < nkf) {
                static_for<D>([&](auto d) {
                    if ((j + 1) % D == decltype(d)::value) to_lds(d, j + 1);
                });
            } else if (j + 1 < nk) {
                tail_to_lds(j + 1);
            }
            __syncthreads();
        }
        K1P_OUT("S loader: to_lds (wait + write), issue, barrier")
        return;
    }
    // -------------------------------------- compute waves ----------------------------------------
    const int wm = wave / C::WN, wn = wave % C::WN;
    f32x4 acc[C::TM][C::TN];
    zero_acc<C>(acc);
    __syncthreads();                                           // k-tile 0 ready
    int st = 0;
    K1P_DECL
    for (int kt = 0; kt < nk; ++kt) {
        K1P_T(0)
        K1P_T(1)
        const bf16raw* As = (const bf16raw*)(fw_lds + (kt & 1) * ABYTES);
        const bf16raw* Bs = (const bf16raw*)(fw_lds + BOFF + st * BBYTES);
        st = st == 2 ? 0 : st + 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[C::TM], bfr[C::TN];
            const int ch = kk * 4 + (lane >> 4);
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm) af[tm] = *(const bf16x8*)&As[lds_off<C>(frag_row<C>(wm, tm, lane), ch)];
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn)
                bfr[tn] = *(const bf16x8*)&Bs[lds_off<C>(wn * (C::BN / C::WN) + tn * 16 + (lane & 15), ch)];
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[tn], af[tm], acc[tm][tn], 0, 0, 0);
        }
        K1P_MFMA
        K1P_T(2)
        __syncthreads();                                       // k-tile kt consumed; kt+1 written by the loaders
        K1P_T(3)
        K1P_ACC
    }
    K1P_OUT("compute: -, reads + MFMA, barrier")
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
        const int m = m0 + frag_row<C>(wm, tm, lane);
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            const int n = frag_col<C>(wn, tn, lane);
            if (m < M && n < s.Ep) *(u32x2*)&X[(long long)m * s.Ep + n] = pack_bf16x4(acc[tm][tn]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1 as STREAM-K over (row tile, k-tile) units: a ragged batch leaves ~45 % of the 128-row tiles
// empty, so one workgroup per tile keeps only ~141 of 256 CUs busy, each streaming all of W_emb^T
// (6 MB) at the per-CU L2 read rate.  Here the non-empty tiles are listed (k_tile_list), the
// ntiles * nk units are cut into gridDim.x equal contiguous ranges, and a workgroup runs the
// three-role pipeline of k_softemb_fwd_ws3 over each (tile, k-range) segment of its range.  A segment
// that covers a whole tile stores bf16 X directly; the (at most two) partial segments of a workgroup
// go to fp32 slabs in accumulator-fragment order (1 KB per wave store) and k_softemb_fwd_fix adds the
// slabs of a tile in workgroup order: the result does not depend on timing.
// ------------------------------------------------------------------------------------------------
constexpr int K1_SLAB_FLOATS = 128 * 320;                   // one partial tile
constexpr int K1_MAX_WGS = 256;

// Unit ranges of the workgroups.  Equal ranges of ceil(total / G) units start every workgroup at a
// different k, so the 32 workgroups of an XCD sweep W_emb^T (6 MB > the 4 MB L2) out of step and nearly
// every W tile comes from the Infinity Cache (measured: FETCH_SIZE 1.05 GB per launch, 1.9 us per unit).
// Instead workgroup g takes units [g*a*nk/16, (g+1)*a*nk/16): ranges of a/16 of a tile, so there are
// only 16 distinct start offsets, and because blocks g, g+8, ... share an XCD (speed only) an XCD
// hosts two cohorts of 16 workgroups that advance through k in step and share their W tiles in L2.
__host__ __device__ inline int k1_sixteenths(int ntiles, int G) {           // a: smallest with ceil(16 ntiles / a) <= G
    const int a = (16 * ntiles + G - 1) / G;
    return a < 1 ? 1 : a;
}
__host__ __device__ inline long long k1_u0(int g, int a, int nk) { return (long long)g * a * nk / 16; }
// the workgroup whose range holds unit u
__host__ __device__ inline int k1_wg_of(long long u, int a, int nk) {
    const long long d = (long long)a * nk;
    return (int)(((u + 1) * 16 + d - 1) / d) - 1;
}

// tiles[0] = number of 128-row tiles with a needed row, tiles[1..] = their indices (ascending).
__global__ __launch_bounds__(1024) void k_tile_list(int M, const unsigned char* __restrict__ need, int* __restrict__ tiles) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = (M + 127) / 128;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int t0 = 0; t0 < nt; t0 += 1024) {
        const int ti = t0 + tid;
        int f = 0;
        if (ti < nt) {
            const int m0 = ti * 128, rows = M - m0 < 128 ? M - m0 : 128;
            if (rows == 128 && ((unsigned long long)(need + m0) & 15) == 0) {
                u32x4 o = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const u32x4 v = *(const u32x4*)(need + m0 + 16 * i);
                    o[0] |= v[0]; o[1] |= v[1]; o[2] |= v[2]; o[3] |= v[3];
                }
                f = (o[0] | o[1] | o[2] | o[3]) != 0;
            } else {
                for (int r = 0; r < rows; ++r) f |= need[m0 + r] != 0;
            }
        }
        const unsigned long long bal = __ballot(f);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (f) tiles[1 + off + before] = ti;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wsum[w];
            base += tot;
        }
        __syncthreads();
    }
    if (tid == 0) tiles[0] = base;
}

// Timing-only ablations of the stream-K kernel's roles (tools/k1abl.py builds variant libraries with -DSAD_K1_ABL=<bits>;
// results are garbage): 1 compute waves skip the MFMAs, 2 ... and their LDS fragment reads, 4 S loaders skip the global
// loads, 8 ... and the LDS writes, 16 W loaders skip the LDS-direct loads.
#ifndef SAD_K1_ABL
#define SAD_K1_ABL 0
#endif
template <int KIND>
__global__ __launch_bounds__(1024) void k_softemb_fwd_sk(Shape s, const void* __restrict__ src, long long sT, long long sB,
                                                       const unsigned char* __restrict__ need, const int* __restrict__ tiles,
                                                       const bf16raw* __restrict__ wembT, bf16raw* __restrict__ X,
                                                       float* __restrict__ slabs) {
    typedef Cfg<128, 320, 2, 4, 1> C;
    typedef SoftRow<KIND> SR;
    typedef typename SR::Elem Elem;
    constexpr int ABYTES = 128 * 128, BBYTES = 320 * 128, BOFF = 2 * ABYTES;   // LDS: A0 A1 B0 B1 B2
    constexpr int D = sizeof(typename SR::Raw) > 16 ? 2 : SAD_K1F_D;
    extern __shared__ __attribute__((aligned(16))) unsigned char fw_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = s.M, V = s.V, Vp = s.Vp;
    const int nk = (V + 63) / 64, nkf = V / 64;
    // this workgroup's range of units (every wave computes the same numbers)
    const int total = tiles[0] * nk;
    const int sx = k1_sixteenths(tiles[0], (int)gridDim.x);
    const long long u0l = k1_u0((int)blockIdx.x, sx, nk), u1l = k1_u0((int)blockIdx.x + 1, sx, nk);
    if (u0l >= total) return;
    const int u0 = (int)u0l, u1 = u1l < total ? (int)u1l : total;
    if (u0 >= u1) return;
    const int first_tile = u0 / nk;

    K1P_DECL
    if (wave >= 12) {
        // ---------------------------------- W loader waves -------------------------------------
        for (int u = u0; u < u1;) {
        const int ti = u / nk, ka = u - ti * nk;
        const int kb = ka + (u1 - u) < nk ? ka + (u1 - u) : nk;
        const int n = kb - ka;                                 // k-tiles of this segment, >= 1
        u += n;
            const int ww = wave - 12, r = lane >> 3, sl = lane & 7;
            const bf16raw* be = wembT + (long long)(ww * 80 + r) * Vp + ((sl ^ (r >> 1)) << 3) + ka * 64;
            const bf16raw* bo = wembT + (long long)(ww * 80 + 8 + r) * Vp + ((sl ^ (4 + (r >> 1))) << 3) + ka * 64;
            const long long step2 = 16ll * Vp;
            auto issue = [&](int j, int st) {
                if (SAD_K1_ABL & 16) return;
                unsigned char* dst = fw_lds + BOFF + st * BBYTES + ww * (80 * 128);
                const int k0 = j * 64;
#pragma unroll
                for (int y = 0; y < 5; ++y) {
                    __builtin_amdgcn_global_load_lds((gmem_cv*)(be + y * step2 + k0), (lds_v*)(dst + (2 * y) * 1024), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((gmem_cv*)(bo + y * step2 + k0), (lds_v*)(dst + (2 * y + 1) * 1024), 16, 0, 0);
                }
            };
            issue(0, 0);
            if (n > 1) {
                issue(1, 1);
                asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                      // k-tile 0 of the segment ready
            int st2 = 2;
            for (int j = 0; j < n; ++j) {
                K1P_T(0)
                if (j + 2 < n) {
                    issue(j + 2, st2);
                    K1P_T(1)
                    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                } else {
                    K1P_T(1)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                K1P_T(2)
                st2 = st2 == 2 ? 0 : st2 + 1;
                __builtin_amdgcn_s_barrier();
                K1P_T(3)
                K1P_ACC
            }
        }
        K1P_OUT("W loader: issue, wait, barrier")
        return;
    }
    if (wave >= 8) {
        // ---------------------------------- S loader waves -------------------------------------
        for (int u = u0; u < u1;) {
        const int ti = u / nk, ka = u - ti * nk;
        const int kb = ka + (u1 - u) < nk ? ka + (u1 - u) : nk;
        const int n = kb - ka;                                 // k-tiles of this segment, >= 1
        const int m0 = tiles[1 + ti] * 128;
        u += n;
            const int lt = tid - 512;                          // 0..255: tile row lt/8 + 32 i, 16-byte chunk lt%8
            const int r0 = lt >> 3, c0 = lt & 7;
            const Elem* arow[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + r0 + 32 * i;
                arow[i] = nullptr;
                if (m < M && need[m]) {
                    const int t = m / s.B, b = m - t * s.B;
                    arow[i] = (const Elem*)src + (long long)t * sT + (long long)b * sB + c0 * 8;
                }
            }
            const int aoff0 = lds_off<C>(r0, c0) * 2;          // row + 32 i keeps the swizzle: + i * 4096 bytes
            typename SR::Raw ra[D][4];
            if (SAD_K1_ABL & 4) memset(ra, 0, sizeof(ra));
            const Elem* aeff[4];                               // rows nobody needs load row 0 and are zeroed in to_lds
#pragma unroll
            for (int i = 0; i < 4; ++i) aeff[i] = arow[i] != nullptr ? arow[i] : (const Elem*)src;
            // whole k-tiles of this segment: relative j < nw (absolute ka + j < nkf); loads are clamped to them
            const int nw = (nkf < kb ? nkf : kb) - ka;         // may be <= 0: the segment is the ragged tile alone
            auto load_tile = [&](auto par, int j) {
                constexpr int P = decltype(par)::value;
                if (SAD_K1_ABL & 4) return;
                const int k0 = (ka + (j < nw ? j : nw - 1)) * 64;
#pragma unroll
                for (int i = 0; i < 4; ++i) ra[P][i] = SR::load_raw(aeff[i] + k0);
            };
            auto to_lds = [&](auto par, int j) {
                constexpr int P = decltype(par)::value;
                if (SAD_K1_ABL & 8) return;
                unsigned char* buf = fw_lds + (j & 1) * ABYTES;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4 v = typename SR::Cvt()(ra[P][i]);
                    *(u32x4*)&buf[aoff0 + i * 4096] = arow[i] != nullptr ? v : u32x4{0u, 0u, 0u, 0u};
                }
            };
            auto tail_to_lds = [&](int j) {                    // the ragged last k-tile: element-guarded loads
                unsigned char* buf = fw_lds + (j & 1) * ABYTES;
                const int k0 = (ka + j) * 64;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *(u32x4*)&buf[aoff0 + i * 4096] =
                        typename SR::Cvt()(SR::load(arow[i] != nullptr ? arow[i] - c0 * 8 : nullptr, k0 + c0 * 8, V));
            };
            int j = 0;
            if (nw >= 1) {
                static_for<D>([&](auto d) { load_tile(d, decltype(d)::value); });
                to_lds(std::integral_constant<int, 0>{}, 0);
                load_tile(std::integral_constant<int, 0>{}, D);
            } else {
                tail_to_lds(0);
            }
            __syncthreads();                                   // k-tile 0 of the segment ready
            for (; j + D < nw; j += D) {                       // steady state, branch-free
                static_for<D>([&](auto uu) {
                    constexpr int U = decltype(uu)::value;
                    using P = std::integral_constant<int, (U + 1) % D>;
                    K1P_T(0)
                    to_lds(P{}, j + U + 1);
                    K1P_T(1)
                    load_tile(P{}, j + U + 1 + D);
                    K1P_T(2)
                    __syncthreads();
                    __builtin_amdgcn_sched_barrier(0);         // keeps the next tile's zeroing selects (uses of loads in flight) below
                    K1P_T(3)
                    K1P_ACC
                });
            }
            for (; j < n; ++j) {
                if (j + 1 < nw) {
                    static_for<D>([&](auto d) {
                        if ((j + 1) % D == decltype(d)::value) to_lds(d, j + 1);
                    });
                } else if (j + 1 < n) {
                    tail_to_lds(j + 1);
                }
                __syncthreads();
            }
        }
        K1P_OUT("S loader: to_lds (wait + write), issue, barrier")
        return;
    }
    // -------------------------------------- compute waves ------------------------------------
    for (int u = u0; u < u1;) {
        const int ti = u / nk, ka = u - ti * nk;
        const int kb = ka + (u1 - u) < nk ? ka + (u1 - u) : nk;
        const int n = kb - ka;                                 // k-tiles of this segment, >= 1
        const int m0 = tiles[1 + ti] * 128;
        u += n;
        const int wm = wave / C::WN, wn = wave % C::WN;
        f32x4 acc[C::TM][C::TN];
        zero_acc<C>(acc);
        __syncthreads();                                       // k-tile 0 of the segment ready
        int st = 0;
        for (int j = 0; j < n; ++j) {
            K1P_T(0)
            K1P_T(1)
            const bf16raw* As = (const bf16raw*)(fw_lds + (j & 1) * ABYTES);
            const bf16raw* Bs = (const bf16raw*)(fw_lds + BOFF + st * BBYTES);
            st = st == 2 ? 0 : st + 1;
#pragma unroll
            for (int kk = 0; kk < ((SAD_K1_ABL & 2) ? 0 : 2); ++kk) {
                bf16x8 af[C::TM], bfr[C::TN];
                const int ch = kk * 4 + (lane >> 4);
#pragma unroll
                for (int tm = 0; tm < C::TM; ++tm) af[tm] = *(const bf16x8*)&As[lds_off<C>(frag_row<C>(wm, tm, lane), ch)];
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn)
                    bfr[tn] = *(const bf16x8*)&Bs[lds_off<C>(wn * (C::BN / C::WN) + tn * 16 + (lane & 15), ch)];
                if (SAD_K1_ABL & 1) {
#pragma unroll
                    for (int tm = 0; tm < C::TM; ++tm) asm volatile("" ::"v"(af[tm]));
#pragma unroll
                    for (int tn = 0; tn < C::TN; ++tn) asm volatile("" ::"v"(bfr[tn]));
                    continue;
                }
#pragma unroll
                for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < C::TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[tn], af[tm], acc[tm][tn], 0, 0, 0);
            }
            K1P_MFMA
            K1P_T(2)
            __syncthreads();                                   // k-tile consumed; the next one written by the loaders
            K1P_T(3)
            K1P_ACC
        }
        if (n == nk) {
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm) {
                const int m = m0 + frag_row<C>(wm, tm, lane);
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn) {
                    const int nn = frag_col<C>(wn, tn, lane);
                    if (m < M && nn < s.Ep) *(u32x2*)&X[(long long)m * s.Ep + nn] = pack_bf16x4(acc[tm][tn]);
                }
            }
        } else {
            const int slot = ti == first_tile ? 0 : 1;
            float* slab = slabs + ((long long)blockIdx.x * 2 + slot) * K1_SLAB_FLOATS + (long long)wave * (C::TM * C::TN * 256);
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn) *(f32x4*)&slab[((tm * C::TN + tn) * 64 + lane) * 4] = acc[tm][tn];
        }
    }
    K1P_OUT("compute: -, reads + MFMA, barrier")
}

// Adds the slabs of every row tile that k_softemb_fwd_sk split over workgroups (in workgroup order) and
// writes its bf16 X rows.  Same unit partition (nwg = the stream-K grid).  One wave per block:
// block (ti, w, tm) takes the accumulator fragments of row fragment tm of compute wave w (many small
// blocks: the kernel is a latency chain of a few dependent loads).
__global__ __launch_bounds__(64) void k_softemb_fwd_fix(Shape s, const int* __restrict__ tiles, int nwg,
                                                       const float* __restrict__ slabs, bf16raw* __restrict__ X) {
    typedef Cfg<128, 320, 2, 4, 1> C;
    static_assert(C::TN == 5 && C::TM == 4, "one block per (tile, compute wave, row fragment)");
    const int ti = blockIdx.x;
    if (ti >= tiles[0]) return;
    const int nk = (s.V + 63) / 64;
    const int sx = k1_sixteenths(tiles[0], nwg);
    const int g0 = k1_wg_of((long long)ti * nk, sx, nk), g1 = k1_wg_of((long long)(ti + 1) * nk - 1, sx, nk);
    if (g0 == g1) return;                                      // one workgroup had the whole tile: stored directly
    const int lane = threadIdx.x, wave = blockIdx.y, tm = blockIdx.z;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int m0 = tiles[1 + ti] * 128;
    f32x4 v[C::TN];
#pragma unroll
    for (int i = 0; i < C::TN; ++i) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // four slabs' loads in flight at a time (small batches: up to 16 slabs per tile); added in workgroup order
    for (int g = g0; g <= g1; g += 4) {
        f32x4 w[4][C::TN];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gj = g + j <= g1 ? g + j : g1;           // clamped: loads are never skipped
            const int slot = (int)(k1_u0(gj, sx, nk) / nk) == ti ? 0 : 1;
            const float* slab = slabs + ((long long)gj * 2 + slot) * K1_SLAB_FLOATS + (long long)wave * (C::TM * C::TN * 256);
#pragma unroll
            for (int i = 0; i < C::TN; ++i) w[j][i] = *(const f32x4*)&slab[((tm * C::TN + i) * 64 + lane) * 4];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            // short k ranges (nk < 16) leave some workgroups without units: they wrote no slab
            if (g + j <= g1 && k1_u0(g + j, sx, nk) < k1_u0(g + j + 1, sx, nk)) {
#pragma unroll
                for (int i = 0; i < C::TN; ++i) v[i] += w[j][i];
            }
    }
    const int m = m0 + frag_row<C>(wm, tm, lane);
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) {
        const int nn = frag_col<C>(wn, tn, lane);
        if (m < s.M && nn < s.Ep) *(u32x2*)&X[(long long)m * s.Ep + nn] = pack_bf16x4(v[tn]);
    }
}

// fp32 split-K accumulator -> bf16 operand buffer (8 elements per thread)
__global__ void k_cvt_bf16(const float* __restrict__ src, bf16raw* __restrict__ dst, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 a = *(const f32x4*)(src + i * 8), b = *(const f32x4*)(src + i * 8 + 4);
        *(u32x4*)(dst + i * 8) = pack_bf16x8(a, b);
    }
}

// Hard tokens: X[m] = bf16(W_emb[ids[b,t]])  (transfer/models.py:900).  One 16-byte chunk per thread.
__global__ void k_embed_gather(Shape s, const long long* __restrict__ ids, long long ib, long long it,
                               const bf16raw* __restrict__ wemb, bf16raw* __restrict__ X) {
    const int cpr = s.Ep / 8;
    const long long total = (long long)s.M * cpr;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(c / cpr), kc = (int)(c - (long long)m * cpr) * 8;
        const int t = m / s.B, b = m - t * s.B;
        const long long id = ids[(long long)b * ib + (long long)t * it];
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (id >= 0 && id < s.V) v = *(const u32x4*)(wemb + id * s.Ep + kc);
        *(u32x4*)(X + (long long)m * s.Ep + kc) = v;
    }
}

// Inverted dropout on the embedded tokens (classifier copy in training mode, classifier/models.py:217),
// with the caller's keep mask [M][E]:  X[m][e] = keep ? X[m][e] * scale : 0.  The same kernel masks
// d(embedded tokens) in the backward.  One 16-byte chunk per thread.
// Input of RNN layer l+1 = per-step outputs of layer l (nn.LSTM with num_layers > 1, reference
// classifier/models.py:104-110):  X[m][dir*H + j] = Hs[m][dir][j], zeros for padded steps and padded columns,
// times the inter-layer dropout mask when given.  sl = lower layer's shape, su = upper layer's (su.E == 2*sl.H).
__global__ void k_hs_to_x(Shape sl, Shape su, const int* __restrict__ lens, const bf16raw* __restrict__ Hs,
                          const unsigned char* __restrict__ keep, float scale, bf16raw* __restrict__ X) {
    const long long total = (long long)su.M * su.Ep;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / su.Ep), c = (int)(i - (long long)m * su.Ep);
        const int t = m / sl.B, b = m - t * sl.B;
        float v = 0.f;
        if (c < 2 * sl.H && t < lens[b]) {
            const int dir = c / sl.H, j = c - dir * sl.H;
            v = bf16_to_f32(Hs[((long long)m * 2 + dir) * sl.Hp + j]);
            if (keep != nullptr) v = keep[(long long)m * su.E + c] ? v * scale : 0.f;
        }
        X[i] = f32_to_bf16(v);
    }
}

__global__ void k_apply_keep(Shape s, bf16raw* __restrict__ X, const unsigned char* __restrict__ keep, float scale) {
    const int cpr = s.Ep / 8;
    const long long total = (long long)s.M * cpr;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(c / cpr), e0 = (int)(c - (long long)m * cpr) * 8;
        bf16raw* p = X + (long long)m * s.Ep + e0;
        const u32x4 v = *(const u32x4*)p;
        bf16raw out[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16raw x = (bf16raw)((v[i >> 1] >> ((i & 1) * 16)) & 0xffffu);
            const int e = e0 + i;
            const bool k = e < s.E && keep[(long long)m * s.E + e] != 0;
            out[i] = k ? f32_to_bf16(bf16_to_f32(x) * scale) : (bf16raw)0;
        }
        *(u32x4*)p = u32x4{out[0] | ((unsigned)out[1] << 16), out[2] | ((unsigned)out[3] << 16),
                           out[4] | ((unsigned)out[5] << 16), out[6] | ((unsigned)out[7] << 16)};
    }
}

// ------------------------------------------------------------------------------------------------
// K2a + K3  fused input projection:  [G | Kk][m] = X[m] . Wcat^T + bcat
// (nn.LSTM input GEMMs of both directions, transfer/models.py:779-785,906, and linear_emb, :925)
// ------------------------------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(C::NT) void k_inproj_fwd(Shape s, const bf16raw* __restrict__ X,
                                                     const unsigned char* __restrict__ need,
                                                     const bf16raw* __restrict__ wcat, const float* __restrict__ bcat,
                                                     float* __restrict__ G, float* __restrict__ Kk) {
    __shared__ __attribute__((aligned(16))) bf16raw As[C::BM * C::LDT];
    __shared__ __attribute__((aligned(16))) bf16raw Bs[C::BN * C::LDT];
    const int n0 = blockIdx.x * C::BN, m0 = blockIdx.y * C::BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!tile_any_need<C>(need, m0, s.M)) return;   // rows nobody reads (t >= len and masked)
    f32x4 acc[C::TM][C::TN];
    zero_acc<C>(acc);
    const int Ep = s.Ep, M = s.M, Ncat = s.Ncat;
    gemm_core<C, u32x4>(
        Ep, [&](int, int r, int k) { return load_packed_nb(X, Ep, m0 + r, M, k, Ep); }, Ident(),
        [&](int, int r, int k) { return load_packed_nb(wcat, Ep, n0 + r, Ncat, k, Ep); }, acc, As, Bs);
    const int wm = wave / C::WN, wn = wave % C::WN, G2 = 2 * s.Gp;
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
        const int m = m0 + frag_row<C>(wm, tm, lane);
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            const int n = n0 + frag_col<C>(wn, tn, lane);
            if (m < M && n < Ncat) {
                const f32x4 v = acc[tm][tn] + *(const f32x4*)&bcat[n];
                if (n < G2)
                    *(f32x4*)&G[(long long)m * s.Gld + n] = v;
                else
                    *(f32x4*)&Kk[(long long)m * s.Hp + (n - G2)] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2b  one time step of the bidirectional LSTM recurrence (both directions in one launch):
//   gates = G[t] + h_prev . W_hh^T ; i,f,o = sigmoid, g = tanh ; c = f*c_prev + i*g ; h = o*tanh(c)
// Only rows with t < len[b] are touched (packed-sequence semantics, transfer/models.py:904-906).
// Requires C::BN / C::WN == 64: a wave's four n-tiles are the four gates of the same 16 units.
// ------------------------------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(C::NT) void k_lstm_fwd(Shape s, int step, const int* __restrict__ lens,
                                                   const bf16raw* __restrict__ whh, float* __restrict__ G,
                                                   float* __restrict__ Cs, bf16raw* __restrict__ Hs, int dbg) {
    static_assert(C::BN / C::WN == 64 && C::TN == 4, "one 64-column gate block per wave");
    __shared__ __attribute__((aligned(16))) bf16raw As[C::BM * C::LDT];
    __shared__ __attribute__((aligned(16))) bf16raw Bs[C::BN * C::LDT];
    const int dir = blockIdx.z;
    const int t = dir ? s.T - 1 - step : step;
    const int tp = dir ? t + 1 : t - 1;              // time index of the previous recurrence step
    const bool tp_ok = tp >= 0 && tp < s.T;
    const int n0 = blockIdx.x * C::BN, b0 = blockIdx.y * C::BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        int any = 0;
        for (int r = tid; r < C::BM; r += C::NT) {
            const int b = b0 + r;
            if (b < s.B && lens[b] > t) any = 1;
        }
        if (!__syncthreads_or(any)) return;
    }
    f32x4 acc[C::TM][C::TN];
    zero_acc<C>(acc);
    const int Hp = s.Hp, Gp = s.Gp, B = s.B;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int col0 = n0 + wn * 64;                   // first permuted column of this wave's unit block
    const int u4 = (lane >> 4) * 4;
    const int j = (col0 >> 6) * 16 + u4;             // first of this lane's 4 hidden units
    // Epilogue operands (gate pre-activations, c_prev) are requested BEFORE the GEMM so their HBM
    // latency hides under the K loop instead of trailing it.
    f32x4 pre[C::TM][4], cp[C::TM];
    bool act[C::TM];
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
        const int b = b0 + frag_row<C>(wm, tm, lane);
        act[tm] = b < B && col0 < Gp && lens[b] > t;
        cp[tm] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[tm][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (act[tm] && !(dbg & 4)) {
            const float* g = G + ((long long)t * B + b) * s.Gld + dir * Gp + col0 + u4;
#pragma unroll
            for (int q = 0; q < 4; ++q) pre[tm][q] = *(const f32x4*)(g + 16 * q);
            if (tp_ok && lens[b] > tp) cp[tm] = *(const f32x4*)&Cs[(((long long)tp * B + b) * 2 + dir) * Hp + j];
        }
    }
    if (tp_ok && !(dbg & 1)) {
        const bf16raw* arow[C::ACH];
#pragma unroll
        for (int i = 0; i < C::ACH; ++i) {
            const int b = b0 + tid / C::CH + i * C::RSTEP;
            arow[i] = (b < B && lens[b] > tp) ? Hs + (((long long)tp * B + b) * 2 + dir) * Hp : nullptr;
        }
        const bf16raw* w = whh + (long long)dir * Gp * Hp;
        gemm_core<C, u32x4>(
            Hp, [&](int i, int, int k) { return load_row(arow[i], arow[i] != nullptr && k < Hp, Hs, k); },
            Ident(), [&](int, int r, int k) { return load_packed(w, Hp, n0 + r, Gp, k, Hp); }, acc, As, Bs);
    }
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
        if (!act[tm]) continue;
        const int b = b0 + frag_row<C>(wm, tm, lane);
        const long long row = ((long long)t * B + b) * 2 + dir;
        float* g = G + ((long long)t * B + b) * s.Gld + dir * Gp + col0 + u4;
        const f32x4 pi = acc[tm][0] + pre[tm][0];
        const f32x4 pf = acc[tm][1] + pre[tm][1];
        const f32x4 pg = acc[tm][2] + pre[tm][2];
        const f32x4 po = acc[tm][3] + pre[tm][3];
        f32x4 gi, gf, gg, go, c, h;
        if (s.cell == CELL_GRU) {
            // slots [r, z, n_input, n_hidden]; Cs carries the fp32 hidden state (cp = h_prev); G keeps
            // [r, z, n, W_hn h + b_hn] for the BPTT (torch.nn.GRU; reference classifier/models.py:102-110)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gi[e] = sigmoidf_(pi[e]);
                gf[e] = sigmoidf_(pf[e]);
                go[e] = po[e];
                gg[e] = tanhf_(pg[e] + gi[e] * po[e]);
                c[e] = (1.f - gf[e]) * gg[e] + gf[e] * cp[tm][e];
                h[e] = c[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gi[e] = sigmoidf_(pi[e]);
                gf[e] = sigmoidf_(pf[e]);
                gg[e] = tanhf_(pg[e]);
                go[e] = sigmoidf_(po[e]);
                c[e] = gf[e] * cp[tm][e] + gi[e] * gg[e];
                h[e] = go[e] * tanhf_(c[e]);
            }
        }
        if (!(dbg & 2)) {
            *(f32x4*)(g + 0) = gi;
            *(f32x4*)(g + 16) = gf;
            *(f32x4*)(g + 32) = gg;
            *(f32x4*)(g + 48) = go;
        }
        *(f32x4*)&Cs[row * Hp + j] = c;
        *(u32x2*)&Hs[row * Hp + j] = pack_bf16x4(h);
    }
}

// ------------------------------------------------------------------------------------------------
// Small dense layers around the recurrence (K4 and its backward):
//   out[m][n] = sum_k A[m][k] * Bt[n][k] + bias[n]
// AKIND 0: A is fp32 [Mrows][lda].  AKIND 1: A is the [h_fwd(len-1) | h_rev(0)] gather out of Hs
// (hidden.view(-1,2,B,H)[-1].transpose(0,1).reshape(-1,2H), transfer/models.py:929-934).
// The shapes are B x {Hp, 2Hp} x {Hp, 2Hp}: a GFLOP and a few MB, so the kernel is a chain of latencies.  The
// LDS-tiled form (gemm_core: 8-16 k-tiles, each global -> registers -> LDS -> two barriers -> MFMA, 64 x 64 tiles on
// 128 workgroups) took 9.4 us per product (13.6 with the gather).  Here a workgroup owns a 32 x 64 output tile (256 workgroups at B = 1024) and its eight waves
// split K: every wave loads its MFMA fragments straight from global memory (16 rows x 64..128 contiguous bytes
// per load), no LDS and no barrier inside the k loop, and the loads of two k-steps per wave (= 16 of the workgroup) are in flight at once; the
// eight partial tiles meet in LDS once at the end.  Needs N % 32 == 0 and K % 32 == 0 (Hp is a multiple of 32).
// ------------------------------------------------------------------------------------------------
constexpr int LKS_BM = 32, LKS_BN = 64, LKS_PITCH = 68, LKS_WAVES = 8;
constexpr size_t LKS_LDS = (size_t)LKS_WAVES * LKS_BM * LKS_PITCH * 4;
template <int AKIND>
__global__ __launch_bounds__(64 * LKS_WAVES) void k_linear_ks(Shape s, int Mrows, int N, int K, const void* __restrict__ A, int lda,
                                                            const int* __restrict__ lens, const bf16raw* __restrict__ Bt,
                                                            const float* __restrict__ bias, float* __restrict__ out, int ldc) {
    extern __shared__ __attribute__((aligned(16))) float lks_part[];      // [LKS_WAVES][LKS_BM * LKS_PITCH]
    const int n0 = blockIdx.x * LKS_BN, m0 = blockIdx.y * LKS_BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kq = lane >> 4;
    const int nks = K >> 5, ks0 = wave * nks / LKS_WAVES, ks1 = (wave + 1) * nks / LKS_WAVES;
    const int Hp = s.Hp;
    // fragment rows of this lane: two of A (m), four of Bt (n); rows outside the matrix read row 0 and count as zero
    const float* apf[2];
    const bf16raw* aph[2][2];
    bool aok[2], bok[4];
    const bf16raw* bp[4];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int m = m0 + tm * 16 + r;
        if constexpr (AKIND == 0) {
            aok[tm] = m < Mrows;
            apf[tm] = (const float*)A + (aok[tm] ? (long long)m * lda : 0ll) + kq * 8;
        } else {
            const int len = m < Mrows ? lens[m] : 0;
            aok[tm] = len > 0;
            const int bb = aok[tm] ? m : 0, ts = aok[tm] ? len - 1 : 0;
            aph[tm][0] = (const bf16raw*)A + (((long long)ts * s.B + bb) * 2 + 0) * Hp + kq * 8;
            aph[tm][1] = (const bf16raw*)A + (((long long)0 * s.B + bb) * 2 + 1) * Hp + kq * 8;
        }
    }
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
        const int n = n0 + tn * 16 + r;
        bok[tn] = n < N;
        bp[tn] = Bt + (bok[tn] ? (long long)n * K : 0ll) + kq * 8;
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    // two k-steps per round, all 16 loads of a round requested before the first MFMA (branch-free: an odd last
    // step is loaded twice and its second copy zeroed)
    for (int ks = ks0; ks < ks1; ks += 2) {
        const bool two = ks + 1 < ks1;
        const int kk[2] = {ks << 5, (two ? ks + 1 : ks) << 5};
        u32x4 af[2][2], bfr[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                if constexpr (AKIND == 0) {
                    af[h][tm] = pack_bf16x8(*(const f32x4*)(apf[tm] + kk[h]), *(const f32x4*)(apf[tm] + kk[h] + 4));
                } else {
                    const int d = kk[h] >= Hp ? 1 : 0;            // [h_fwd(len-1) | h_rev(0)]
                    af[h][tm] = *(const u32x4*)((d ? aph[tm][1] : aph[tm][0]) + (kk[h] - d * Hp));
                }
                if (!aok[tm] || (h == 1 && !two)) af[h][tm] = u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                bfr[h][tn] = *(const u32x4*)(bp[tn] + kk[h]);
                if (!bok[tn]) bfr[h][tn] = u32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bfr[h][tn]),
                                                                          __builtin_bit_cast(bf16x8, af[h][tm]), acc[tm][tn], 0, 0, 0);
    }
    float* mine = lks_part + wave * (LKS_BM * LKS_PITCH);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) *(f32x4*)&mine[(tm * 16 + r) * LKS_PITCH + tn * 16 + kq * 4] = acc[tm][tn];
    __syncthreads();
    {
        const int ml = tid >> 4, nl = (tid & 15) * 4;             // 512 threads: one float4 of the 32 x 64 tile each
        const int m = m0 + ml, n = n0 + nl;
        if (m < Mrows && n < N) {
            f32x4 v = *(const f32x4*)&lks_part[ml * LKS_PITCH + nl];
#pragma unroll
            for (int w = 1; w < LKS_WAVES; ++w) v += *(const f32x4*)&lks_part[w * (LKS_BM * LKS_PITCH) + ml * LKS_PITCH + nl];
            if (bias != nullptr) v += *(const f32x4*)&bias[n];
            *(f32x4*)&out[(long long)m * ldc + n] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K5  fused additive attention + masked softmax + pooling + 2-way head + per-sentence CE
// (transfer/models.py:855-880, :949; nn.CrossEntropyLoss(reduction='none'), :990,:1212).
// One 256-thread workgroup per sentence; wave w owns time steps t = w, w+4, ...; lanes split H;
// wavefront reductions over H, online softmax over T, cross-wave combine through LDS.
// NV = float4 groups per lane (Hp <= NV*256).
// ------------------------------------------------------------------------------------------------
constexpr int ATT_MAX_T = 1024;
constexpr int ATT_MAX_L = 64;

template <int NV>
__global__ __launch_bounds__(256) void k_attn_fwd(Shape s, const float* __restrict__ Kk, const float* __restrict__ U,
                                                 const unsigned char* __restrict__ need, const float* __restrict__ wa,
                                                 const float* __restrict__ ba, const float* __restrict__ wo,
                                                 const float* __restrict__ bo, const long long* __restrict__ labels,
                                                 float* __restrict__ weights, float* __restrict__ ctx_out,
                                                 float* __restrict__ logits, float* __restrict__ loss,
                                                 const unsigned* __restrict__ errword) {
    __shared__ float sc[ATT_MAX_T];
    __shared__ float wm_[4], wl_[4];
    __shared__ __attribute__((aligned(16))) float cx[4][NV * 256];
    __shared__ float lg[ATT_MAX_L];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Hp = s.Hp, B = s.B, T = s.T;
    f32x4 u[NV], w[NV], cacc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int h = v * 256 + lane * 4;
        u[v] = w[v] = cacc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (h < Hp) {
            u[v] = *(const f32x4*)&U[(long long)b * Hp + h];
            w[v] = *(const f32x4*)&wa[h];
        }
    }
    const float bias_a = ba[0];
    float mx = -INFINITY, l = 0.f;
    // The unmasked steps of this wave as a bit mask (lane j looks at t = tb + wave + 4 j; masked positions get
    // weight exactly 0 and are skipped), then the key rows of up to four steps in flight: a step-by-step loop is
    // a chain of two dependent global loads (need byte, key row) per step, 16 latencies at T = 32.
    for (int tb = 0; tb < T; tb += 256) {
        const int tl = tb + wave + 4 * lane;
        unsigned long long todo = __ballot(tl < T && (need[(long long)tl * B + b] & 2));
        while (todo) {
            int tt[4];
            f32x4 kk[4][NV];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tt[k] = -1;
                if (todo) {
                    tt[k] = tb + wave + 4 * (int)__builtin_ctzll(todo);
                    todo &= todo - 1;
                    const long long m = (long long)tt[k] * B + b;
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const int h = v * 256 + lane * 4;
                        kk[k][v] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (h < Hp) kk[k][v] = *(const f32x4*)&Kk[m * Hp + h];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (tt[k] < 0) break;
                float part = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) part += w[v][e] * tanhf_(kk[k][v][e] + u[v][e]);   // w = 0 at h >= Hp
                const float sc_t = wave_sum(part) + bias_a;
                if (lane == 0) sc[tt[k]] = sc_t;
                const float mn = fmaxf(mx, sc_t);
                const float scale = __expf(mx - mn), p = __expf(sc_t - mn);   // exp(-inf) = 0 on the first hit
                l = l * scale + p;
#pragma unroll
                for (int v = 0; v < NV; ++v) cacc[v] = cacc[v] * scale + kk[k][v] * p;
                mx = mn;
            }
        }
    }
    if (lane == 0) {
        wm_[wave] = mx;
        wl_[wave] = l;
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) *(f32x4*)&cx[wave][v * 256 + lane * 4] = cacc[v];
    __syncthreads();
    float gm = fmaxf(fmaxf(wm_[0], wm_[1]), fmaxf(wm_[2], wm_[3]));
    float gl = 0.f, sw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sw[k] = (wm_[k] == -INFINITY) ? 0.f : __expf(wm_[k] - gm);
        gl += wl_[k] * sw[k];
    }
    // gl == 0: every position masked -> softmax of all -inf is NaN in the reference; mirror it.
    // A hand-off wait of the persistent recurrence timed out (errword, zeroed by k_prep of this call):
    // h may be stale for some sentences -> every sentence's weights / logits / loss come out NaN.
    if (errword != nullptr && *errword != 0u) gl = 0.f;
    const float inv = gl > 0.f ? 1.0f / gl : __builtin_nanf("");
    for (int t = tid; t < T; t += 256) {
        const long long m = (long long)t * B + b;
        float a = (need[m] & 2) ? __expf(sc[t] - gm) * inv : 0.f;
        if (!(gl > 0.f)) a = inv;
        weights[(long long)b * T + t] = a;
    }
    __syncthreads();   // cx[0] is rewritten below
    for (int h = tid; h < Hp; h += 256) {
        const float c = (cx[0][h] * sw[0] + cx[1][h] * sw[1] + cx[2][h] * sw[2] + cx[3][h] * sw[3]) * inv;
        ctx_out[(long long)b * Hp + h] = c;
        cx[0][h] = c;
    }
    __syncthreads();
    for (int c = wave; c < s.L; c += 4) {
        float part = 0.f;
        for (int h = lane; h < Hp; h += 64) part += wo[(long long)c * Hp + h] * cx[0][h];
        const float v = wave_sum(part) + bo[c];
        if (lane == 0) {
            logits[(long long)b * s.L + c] = v;
            lg[c] = v;
        }
    }
    if (loss != nullptr && labels != nullptr) {
        __syncthreads();
        if (tid == 0) {
            float m2 = -INFINITY;
            for (int c = 0; c < s.L; ++c) m2 = fmaxf(m2, lg[c]);
            float se = 0.f;
            for (int c = 0; c < s.L; ++c) se += __expf(lg[c] - m2);
            const long long lab = labels[b];
            loss[b] = (lab >= 0 && lab < s.L) ? (__logf(se) + m2 - lg[lab]) : __builtin_nanf("");
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K5'  backward of K5: dlogits (given, or CE(logits,labels)*dloss) -> dKk (bf16, written into the
// key columns of DZ for EVERY t, zeros where masked) and du = d linear_q(q).  Also leaves the
// per-sentence pieces of the head's parameter gradients (used only by sad_param_grads):
// DL = dlogits, DWA[b] = sum_t ds_t * tanh(Kk_t + u)  (d linear_attn.weight), DSSUM[b] = sum_t ds_t.
// ------------------------------------------------------------------------------------------------
constexpr int ATT_BWD_INFLIGHT = 2;
template <int NV>
__global__ __launch_bounds__(256) void k_attn_bwd(Shape s, const float* __restrict__ Kk, const float* __restrict__ U,
                                                 const float* __restrict__ wa, const float* __restrict__ wo,
                                                 const float* __restrict__ weights, const float* __restrict__ ctx,
                                                 const float* __restrict__ logits, const long long* __restrict__ labels,
                                                 const float* __restrict__ dlogits, const float* __restrict__ dloss,
                                                 float dloss_scale, const float* __restrict__ dweights,
                                                 bf16raw* __restrict__ DZ, float* __restrict__ DU,
                                                 float* __restrict__ DL, float* __restrict__ DWA,
                                                 float* __restrict__ DSSUM, unsigned* __restrict__ sync, int sync_words) {
    __shared__ float dl[ATT_MAX_L];
    // hand-off counters of the persistent BPTT start from zero (the timeout words behind them keep
    // what the forward left)
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < sync_words; i += 256) sync[i] = 0u;
    __shared__ __attribute__((aligned(16))) float dux[4][NV * 256];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Hp = s.Hp, B = s.B, T = s.T, L = s.L;
    // everything that does not depend on dlogits is requested before thread 0 works them out (the kernel is a
    // chain of global-load latencies: ~1 us each, ~10 per sentence)
    f32x4 u[NV], w[NV], cv[NV], wo0[NV], wo1[NV], dctx[NV], du[NV], dwa[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int h = v * 256 + lane * 4;
        u[v] = w[v] = cv[v] = wo0[v] = wo1[v] = dctx[v] = du[v] = dwa[v] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (h < Hp) {
            u[v] = *(const f32x4*)&U[(long long)b * Hp + h];
            w[v] = *(const f32x4*)&wa[h];
            cv[v] = *(const f32x4*)&ctx[(long long)b * Hp + h];
            wo0[v] = *(const f32x4*)&wo[h];
            if (L > 1) wo1[v] = *(const f32x4*)&wo[Hp + h];
        }
    }
    if (tid == 0) {
        for (int c = 0; c < L; ++c) dl[c] = dlogits != nullptr ? dlogits[(long long)b * L + c] : 0.f;
        if (labels != nullptr && (dlogits == nullptr || dloss != nullptr || dloss_scale != 0.f)) {
            // + d CE(logits, labels) scaled by the upstream gradient of the per-sentence loss
            const float up = dloss != nullptr ? dloss[b] : dloss_scale;
            float m2 = -INFINITY;
            for (int c = 0; c < L; ++c) m2 = fmaxf(m2, logits[(long long)b * L + c]);
            float se = 0.f;
            for (int c = 0; c < L; ++c) se += __expf(logits[(long long)b * L + c] - m2);
            const long long lab = labels[b];
            for (int c = 0; c < L; ++c)
                dl[c] += (__expf(logits[(long long)b * L + c] - m2) / se - (c == lab ? 1.f : 0.f)) * up;
        }
    }
    __syncthreads();
    if (tid < L) DL[(long long)b * L + tid] = dl[tid];
    float dpart = 0.f, dssum = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int h = v * 256 + lane * 4;
        if (h < Hp) {
            dctx[v] = wo0[v] * dl[0];
            if (L > 1) dctx[v] += wo1[v] * dl[1];
            for (int c = 2; c < L; ++c) dctx[v] += *(const f32x4*)&wo[(long long)c * Hp + h] * dl[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) dpart += dctx[v][e] * cv[v][e];
        }
    }
    // delta = sum_t a_t * da_t = dctx . ctx (+ sum_t a_t * dweights_t); every wave computes it.
    if (dweights != nullptr)
        for (int t = lane; t < T; t += 64) {
            const float a = weights[(long long)b * T + t];
            if (a > 0.f) dpart += a * dweights[(long long)b * T + t];
        }
    const float delta = wave_sum(dpart);
    // Lane j holds the attention weight (and its upstream gradient) of step t = tb + wave + 4 j; the steps with a
    // positive weight are worked through with the key rows of ATT_BWD_INFLIGHT of them in flight (see k_attn_fwd;
    // four in flight need 134 VGPRs: three workgroups per CU instead of four, two rounds at B = 1024: 22 us vs 20).
    for (int tb = 0; tb < T; tb += 256) {
        const int tl = tb + wave + 4 * lane;
        const float a_l = tl < T ? weights[(long long)b * T + tl] : 0.f;
        const float dw_l = tl < T && dweights != nullptr ? dweights[(long long)b * T + tl] : 0.f;
        unsigned long long todo = __ballot(a_l > 0.f);
        unsigned long long zero = __ballot(tl < T && !(a_l > 0.f));   // masked (exact 0), underflowed, or NaN row
        while (zero) {
            const int t = tb + wave + 4 * (int)__builtin_ctzll(zero);
            zero &= zero - 1;
            bf16raw* dz = DZ + ((long long)t * B + b) * s.Ncat + 2 * s.Gp;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int h = v * 256 + lane * 4;
                if (h < Hp) *(u32x2*)&dz[h] = u32x2{0u, 0u};
            }
        }
        while (todo) {
            int jj[ATT_BWD_INFLIGHT];
            f32x4 kk[ATT_BWD_INFLIGHT][NV];
#pragma unroll
            for (int k = 0; k < ATT_BWD_INFLIGHT; ++k) {
                jj[k] = -1;
                if (todo) {
                    jj[k] = (int)__builtin_ctzll(todo);
                    todo &= todo - 1;
                    const long long m = (long long)(tb + wave + 4 * jj[k]) * B + b;
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const int h = v * 256 + lane * 4;
                        kk[k][v] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (h < Hp) kk[k][v] = *(const f32x4*)&Kk[m * Hp + h];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < ATT_BWD_INFLIGHT; ++k) {
                if (jj[k] < 0) break;
                const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a_l), jj[k]));
                const float dw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dw_l), jj[k]));
                bf16raw* dz = DZ + ((long long)(tb + wave + 4 * jj[k]) * B + b) * s.Ncat + 2 * s.Gp;
                f32x4 mid[NV];
                float part = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        mid[v][e] = tanhf_(kk[k][v][e] + u[v][e]);
                        part += dctx[v][e] * kk[k][v][e];          // dctx = 0 at h >= Hp
                    }
                const float da = wave_sum(part) + dw;
                const float ds = a * (da - delta);
                dssum += ds;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int h = v * 256 + lane * 4;
                    if (h < Hp) {
                        const f32x4 dpre = (1.f - mid[v] * mid[v]) * w[v] * ds;
                        du[v] += dpre;
                        dwa[v] += mid[v] * ds;
                        *(u32x2*)&dz[h] = pack_bf16x4(dctx[v] * a + dpre);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) *(f32x4*)&dux[wave][v * 256 + lane * 4] = du[v];
    __syncthreads();
    for (int h = tid; h < Hp; h += 256) DU[(long long)b * Hp + h] = dux[0][h] + dux[1][h] + dux[2][h] + dux[3][h];
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v) *(f32x4*)&dux[wave][v * 256 + lane * 4] = dwa[v];
    if (lane == 0) dl[wave] = dssum;          // dl is dead by now (L <= 64, 4 waves)
    __syncthreads();
    for (int h = tid; h < Hp; h += 256) DWA[(long long)b * Hp + h] = dux[0][h] + dux[1][h] + dux[2][h] + dux[3][h];
    if (tid == 0) DSSUM[b] = dl[0] + dl[1] + dl[2] + dl[3];
}

// ------------------------------------------------------------------------------------------------
// K2'  one BPTT step of the LSTM / GRU (both directions), frozen weights:
//   dh_t = dZ_{next} . W_hh  (+ d h_final at the row's last forward step) (+ DHS[t]: the gradient of the step's
//   output when another RNN layer consumes it), gate derivatives,
//   dZ_t -> DZ (bf16, zeros for t >= len[b]); dc carried in DC.
// ------------------------------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(C::NT) void k_lstm_bwd(Shape s, int step, const int* __restrict__ lens,
                                                   const bf16raw* __restrict__ whhT, const float* __restrict__ G,
                                                   const float* __restrict__ Cs, const float* __restrict__ DHFIN,
                                                   float* __restrict__ DC, bf16raw* __restrict__ DZ,
                                                   const bf16raw* __restrict__ DHS, int ldh) {
    __shared__ __attribute__((aligned(16))) bf16raw As[C::BM * C::LDT];
    __shared__ __attribute__((aligned(16))) bf16raw Bs[C::BN * C::LDT];
    const int dir = blockIdx.z;
    const int t = dir ? step : s.T - 1 - step;       // reverse of the forward order
    const int tn = dir ? t - 1 : t + 1;              // step whose dZ feeds dh_t (processed one launch ago)
    const int tpf = dir ? t + 1 : t - 1;             // forward-previous step (source of c_prev)
    const bool tn_ok = tn >= 0 && tn < s.T, tpf_ok = tpf >= 0 && tpf < s.T;
    const int n0 = blockIdx.x * C::BN, b0 = blockIdx.y * C::BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Hp = s.Hp, Gp = s.Gp, B = s.B, Ncat = s.Ncat;
    int any = 0;
    for (int r = tid; r < C::BM; r += C::NT) {
        const int b = b0 + r;
        if (b < B && lens[b] > t) any = 1;
    }
    any = __syncthreads_or(any);
    f32x4 acc[C::TM][C::TN];
    zero_acc<C>(acc);
    const int wm = wave / C::WN, wn = wave % C::WN;
    // Epilogue operands are requested before the GEMM (their latency hides under the K loop).
    struct Pre {
        f32x4 gi, gf, gg, go, c, cp, dc, dhf;
    } pre[C::TM][C::TN];
    bool act[C::TM];
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
        const int b = b0 + frag_row<C>(wm, tm, lane);
        const int len = b < B ? lens[b] : 0;
        act[tm] = len > t;
        const bool last = dir ? (t == 0) : (t == len - 1);
        const long long row = ((long long)t * B + b) * 2 + dir;
#pragma unroll
        for (int tnn = 0; tnn < C::TN; ++tnn) {
            Pre& p = pre[tm][tnn];
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            p.gi = p.gf = p.gg = p.go = p.c = p.cp = p.dc = p.dhf = z;
            const int j = n0 + frag_col<C>(wn, tnn, lane);   // 4 consecutive hidden units
            if (!act[tm] || j >= Hp) continue;
            const int pc = (j >> 4) * 64 + (j & 15);         // permuted column of gate 0
            const float* g = G + ((long long)t * B + b) * s.Gld + dir * Gp + pc;
            p.gi = *(const f32x4*)(g + 0);
            p.gf = *(const f32x4*)(g + 16);
            p.gg = *(const f32x4*)(g + 32);
            p.go = *(const f32x4*)(g + 48);
            p.c = *(const f32x4*)&Cs[row * Hp + j];
            if (tpf_ok && len > tpf) p.cp = *(const f32x4*)&Cs[(((long long)tpf * B + b) * 2 + dir) * Hp + j];
            if (tn_ok && len > tn) p.dc = *(const f32x4*)&DC[(long long)b * 2 * Hp + dir * Hp + j];
            if (last) p.dhf = *(const f32x4*)&DHFIN[(long long)b * 2 * Hp + dir * Hp + j];
            if (DHS != nullptr) {
                // gradient of this step's output from the layer above: d(input of layer l+1)[t][b][dir*H + unit]
                const bf16raw* q = DHS + ((long long)t * B + b) * ldh + dir * s.H + j;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j + e < s.H) p.dhf[e] += bf16_to_f32(q[e]);
            }
        }
    }
    if (any && tn_ok) {
        const bf16raw* arow[C::ACH];
#pragma unroll
        for (int i = 0; i < C::ACH; ++i) {
            const int b = b0 + tid / C::CH + i * C::RSTEP;
            arow[i] = (b < B && lens[b] > tn && lens[b] > t) ? DZ + ((long long)tn * B + b) * Ncat + (long long)dir * Gp : nullptr;
        }
        const bf16raw* w = whhT + (long long)dir * Hp * Gp;
        gemm_core<C, u32x4>(
            Gp, [&](int i, int, int k) { return load_row(arow[i], arow[i] != nullptr && k < Gp, DZ, k); },
            Ident(), [&](int, int r, int k) { return load_packed(w, Gp, n0 + r, Hp, k, Gp); }, acc, As, Bs);
    }
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
        const int b = b0 + frag_row<C>(wm, tm, lane);
        if (b >= B) continue;
        bf16raw* zrow = DZ + ((long long)t * B + b) * Ncat + (long long)dir * Gp;
#pragma unroll
        for (int tnn = 0; tnn < C::TN; ++tnn) {
            const int j = n0 + frag_col<C>(wn, tnn, lane);
            if (j >= Hp) continue;
            const int pc = (j >> 4) * 64 + (j & 15);
            if (!act[tm]) {
#pragma unroll
                for (int g = 0; g < 4; ++g) *(u32x2*)&zrow[pc + g * 16] = u32x2{0u, 0u};
                continue;
            }
            const Pre& p = pre[tm][tnn];
            const f32x4 dh = acc[tm][tnn] + p.dhf;
            f32x4 zi, zf, zg, zo, dcn;
            if (s.cell == CELL_GRU) {
                // G = [r, z, n, q = W_hn h + b_hn], cp = h_prev, DC carries dh * z to the previous step
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dht = dh[e] + p.dc[e];
                    const float r = p.gi[e], z = p.gf[e], n = p.gg[e], q = p.go[e];
                    const float da = dht * (1.f - z) * (1.f - n * n);
                    zi[e] = da * q * r * (1.f - r);
                    zf[e] = dht * (p.cp[e] - n) * z * (1.f - z);
                    zg[e] = da;
                    zo[e] = da * r;
                    dcn[e] = dht * z;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float tc = tanhf_(p.c[e]);
                    const float d_o = dh[e] * tc;
                    const float dct = p.dc[e] + dh[e] * p.go[e] * (1.f - tc * tc);
                    zi[e] = dct * p.gg[e] * p.gi[e] * (1.f - p.gi[e]);
                    zf[e] = dct * p.cp[e] * p.gf[e] * (1.f - p.gf[e]);
                    zg[e] = dct * p.gi[e] * (1.f - p.gg[e] * p.gg[e]);
                    zo[e] = d_o * p.go[e] * (1.f - p.go[e]);
                    dcn[e] = dct * p.gf[e];
                }
            }
            *(u32x2*)&zrow[pc + 0] = pack_bf16x4(zi);
            *(u32x2*)&zrow[pc + 16] = pack_bf16x4(zf);
            *(u32x2*)&zrow[pc + 32] = pack_bf16x4(zg);
            *(u32x2*)&zrow[pc + 48] = pack_bf16x4(zo);
            *(f32x4*)&DC[(long long)b * 2 * Hp + dir * Hp + j] = dcn;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2a' + K3'  dX[m] = DZ[m] . Wcat   (dgrad of both LSTM input projections and of linear_emb)
// ------------------------------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(C::NT) void k_inproj_bwd(Shape s, const bf16raw* __restrict__ DZ,
                                                     const unsigned char* __restrict__ need,
                                                     const bf16raw* __restrict__ wcatT, bf16raw* __restrict__ DX,
                                                     float* __restrict__ accf, int ksplit) {
    __shared__ __attribute__((aligned(16))) bf16raw As[C::BM * C::LDT];
    __shared__ __attribute__((aligned(16))) bf16raw Bs[C::BN * C::LDT];
    const int m0 = blockIdx.x * C::BM, n0 = blockIdx.y * C::BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[C::TM][C::TN];
    zero_acc<C>(acc);
    const int Ep = s.Ep, M = s.M, Ncat = s.Ncat;
    if (tile_any_need<C>(need, m0, M)) {
        const int tps = ceil_div(ceil_div(Ncat, C::BK), ksplit);
        const int k0 = blockIdx.z * tps * C::BK;
        const int k1 = k0 + tps * C::BK < Ncat ? k0 + tps * C::BK : Ncat;
        if (k1 > k0)
            gemm_core<C, u32x4>(
                k1 - k0, [&](int, int r, int k) { return load_packed(DZ, Ncat, m0 + r, M, k0 + k, Ncat); }, Ident(),
                [&](int, int r, int k) { return load_packed(wcatT, Ncat, n0 + r, Ep, k0 + k, Ncat); }, acc, As, Bs);
    }
    const int wm = wave / C::WN, wn = wave % C::WN;
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
        const int m = m0 + frag_row<C>(wm, tm, lane);
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            const int n = n0 + frag_col<C>(wn, tn, lane);
            if (m < M && n < Ep) {
                if (ksplit > 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(&accf[(long long)m * Ep + n + e], acc[tm][tn][e]);
                } else {
                    *(u32x2*)&DX[(long long)m * Ep + n] = pack_bf16x4(acc[tm][tn]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1'  d(soft tokens)  dS[m][v] = sum_e dX[m][e] * W_emb[v][e]   (autograd of transfer/models.py:898)
// Every (m, v) is written (exact zeros for t >= len[b]), in the caller's dtype and strides.
// ------------------------------------------------------------------------------------------------
template <class C, int KIND>
__global__ __launch_bounds__(C::NT) void k_softemb_bwd(Shape s, const bf16raw* __restrict__ DX,
                                                      const unsigned char* __restrict__ need,
                                                      const bf16raw* __restrict__ wemb, void* __restrict__ dS,
                                                      long long sT, long long sB) {
    __shared__ __attribute__((aligned(16))) bf16raw As[C::BM * C::LDT];
    __shared__ __attribute__((aligned(16))) bf16raw Bs[C::BN * C::LDT];
    const int n0 = blockIdx.x * C::BN, m0 = blockIdx.y * C::BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[C::TM][C::TN];
    zero_acc<C>(acc);
    const int Ep = s.Ep, M = s.M, V = s.V, Vp = s.Vp;
    if (tile_any_need<C>(need, m0, M)) {
        gemm_core<C, u32x4>(
            Ep, [&](int, int r, int k) { return load_packed(DX, Ep, m0 + r, M, k, Ep); }, Ident(),
            [&](int, int r, int k) { return load_packed(wemb, Ep, n0 + r, Vp, k, Ep); }, acc, As, Bs);
    }
    const int wm = wave / C::WN, wn = wave % C::WN;
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
        const int m = m0 + frag_row<C>(wm, tm, lane);
        if (m >= M) continue;
        const int t = m / s.B, b = m - t * s.B;
        const long long base = (long long)t * sT + (long long)b * sB;
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            const int n = n0 + frag_col<C>(wn, tn, lane);
            if (n >= V) continue;
            const f32x4 v = acc[tm][tn];
            if constexpr (KIND == SRC_F32) {
                float* p = (float*)dS + base + n;
                if (n + 4 <= V) {
                    *(f32x4u*)p = v;
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (n + e < V) p[e] = v[e];
                }
            } else {
                bf16raw* p = (bf16raw*)dS + base + n;
                if (n + 4 <= V) {
                    *(u32x2h*)p = pack_bf16x4(v);
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (n + e < V) p[e] = f32_to_bf16(v[e]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// A-stationary GEMM for K = Ep = 32*KS (the embedding size): used by K1' (dS = dX . W_emb^T) and by
// the fused in-projection (G|Kk = X . Wcat^T).  A workgroup owns 256 token rows x a range of
// 64-column tiles of Bt.  Each of its 8 waves keeps the A fragments of its 32 rows for the WHOLE K
// in registers (loaded once); the Bt tile [64][Ep] (one contiguous 64*Ep*2-byte block of the packed
// matrix) is staged through LDS and shared by all 8 waves, so per output byte the kernel moves
// ~Ep/256 bytes of operands instead of re-reading both operands per 128x128 tile.
//   * LDS image: unpadded rows, 16-byte chunks XOR-swizzled with (row>>1)&7 -> conflict-free
//     ds_read_b128 fragment reads (MI355X bank rule: 64 banks, 16-lane groups).
//   * PAIR (2-byte outputs): LDS row rho holds Bt row sigma(rho) (a bit permutation), so that a
//     lane's accumulator tile pairs are 8 CONSECUTIVE output columns -> one 16-byte store per lane.
//     Either way every store instruction writes 64 CONTIGUOUS bytes per token row.
//   * Pipeline per tile (ONE LDS buffer): compute(nt) | barrier | LDS <- tile nt+1 | barrier |
//     stores(nt) | issue loads(nt+2).  vmcnt retires in order, so the wait in front of the LDS write
//     also covers the stores of tile nt-1, which by then had a whole MFMA phase to drain.
// PAIR: epi(tm, n, lo, hi) = columns [n, n+8) of this lane's row tm; else epi(tm, n, v) = [n, n+4).
// ------------------------------------------------------------------------------------------------
// Timing-only ablations of the in-projection forward (tools/k1abl.py, -DSAD_IP_ABL=<bits>; results are garbage):
// 1 no MFMA, 2 no LDS fragment reads and no MFMA, 4 no epilogue (no output stores), 8 no W tile loads, 16 no W tile LDS writes.
#ifndef SAD_IP_ABL
#define SAD_IP_ABL 0
#endif
template <int KS, int PAIR, bool DB = false, class Pre = void, class Epi = void>     // DB: two LDS tile buffers, one barrier per tile; PAIR: 0 = epi per accumulator tile, 1 = per pair of tiles, 2 = per 64-column tile (all of acc)
__device__ __forceinline__ void rs_gemm(const bf16raw* __restrict__ A, int M, const int (&mrow)[2], bool wave_any,
                                        int block_any, const bf16raw* __restrict__ Bt, int nt0, int nt1,
                                        unsigned char* Bs, Pre pre, Epi epi) {
    // pre(nt): called in front of tile nt's MFMAs (DB form): global loads the epilogue needs (bias) go HERE.  vmcnt
    // retires in order, so a load issued inside the epilogue is waited for behind the previous tile's output stores --
    // every tile drained its predecessor's stores before it could store (and nothing overlapped: the ablations'
    // phases added up).  Issued before the MFMAs, both have a whole MFMA phase to complete.
    constexpr int EP = KS * 32, BN = 64, NTH = 512;
    constexpr int ROWB = EP * 2;              // bytes per Bt row
    constexpr int CPR = EP / 8;               // 16-byte chunks per row
    constexpr int SCH = BN * CPR / NTH;       // staging chunks per thread
    static_assert((BN * CPR) % NTH == 0, "tile must split evenly over the workgroup");
    const int tid = threadIdx.x, lane = tid & 63;

    bf16x8 a[2][KS];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (wave_any && mrow[tm] < M) v = *(const u32x4*)(A + (long long)mrow[tm] * EP + ks * 32 + (lane >> 4) * 8);
            a[tm][ks] = __builtin_bit_cast(bf16x8, v);
        }
    // Retire the fragment loads HERE.  Otherwise they are still "pending" at the loop header and
    // hipcc's waitcnt pass, merging that state over the back edge, puts s_waitcnt vmcnt(0) in front
    // of the first MFMA of EVERY tile — draining the previous tile's stores and the prefetch.
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(a[tm][ks]));

    // staging map of this thread: global chunk g -> (Bt row vloc, chunk) -> swizzled LDS offset
    int soff[SCH], goff[SCH];
#pragma unroll
    for (int i = 0; i < SCH; ++i) {
        const int g = tid + i * NTH;
        const int vloc = g / CPR, ch = g - vloc * CPR;
        // PAIR: vloc = 32p + 8g + 4tau + r  <->  rho = 32p + 16tau + 4g + r ; else identity
        const int rho = PAIR == 1 ? (vloc & 32) + 16 * ((vloc >> 2) & 1) + 4 * ((vloc >> 3) & 3) + (vloc & 3) : vloc;
        soff[i] = rho * ROWB + ((ch ^ ((rho >> 1) & 7)) << 4);
        goff[i] = g * 8;                    // elements from the tile's first row (tile is contiguous)
    }
    u32x4 st[SCH];
    if (DB && (SAD_IP_ABL & 8)) memset(st, 0, sizeof(st));
    auto load_tile = [&](int nt) {
        if (DB && (SAD_IP_ABL & 8)) return;
        const bf16raw* base = Bt + (long long)nt * BN * EP;
#pragma unroll
        for (int i = 0; i < SCH; ++i) st[i] = *(const u32x4*)(base + goff[i]);
    };
    // fragment read offsets: LDS row rho = tn*16 + (lane&15), chunk = ks*4 + (lane>>4)
    int roff[4], rsw[4];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
        const int rho = tn * 16 + (lane & 15);
        roff[tn] = rho * ROWB;
        rsw[tn] = (rho >> 1) & 7;
    }
    constexpr int TILEB = BN * ROWB;
    if constexpr (DB) {
        // two tile buffers: tile nt+1 is written (from the registers loaded one iteration earlier) while
        // tile nt is multiplied; one workgroup barrier per tile
        if (block_any) {
            load_tile(nt0);
#pragma unroll
            for (int i = 0; i < SCH; ++i) *(u32x4*)&Bs[soff[i]] = st[i];
            if (nt0 + 1 < nt1) load_tile(nt0 + 1);
            __syncthreads();
        }
        // Role ablations (profiles/r03_inproj_fwd_ablation.txt): the phases of a tile add up -- staging 9 + fragment reads
        // 8 + MFMA 15 + output stores 36 us of an 82 us launch (fixed part 14).  Tried and reverted: the epilogue of tile
        // nt-1 issued beside the MFMAs of tile nt, waves 0-3 before and waves 4-7 (their SIMD mates) after their own
        // MFMAs, accumulators double-buffered (256 VGPRs): 76 -> 81-82 us.
        for (int nt = nt0; nt < nt1; ++nt) {
            f32x4 acc[2][4];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (block_any) {
                const unsigned char* cur = Bs + ((nt - nt0) & 1) * TILEB;
                if (nt + 1 < nt1) {
                    unsigned char* nxt = Bs + ((nt + 1 - nt0) & 1) * TILEB;
                    if (!(SAD_IP_ABL & 16)) {
#pragma unroll
                        for (int i = 0; i < SCH; ++i) *(u32x4*)&nxt[soff[i]] = st[i];
                    }
                    if (nt + 2 < nt1) load_tile(nt + 2);
                }
                pre(nt);
                if (wave_any && !(SAD_IP_ABL & 2)) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        bf16x8 bfr[4];
#pragma unroll
                        for (int tn = 0; tn < 4; ++tn)
                            bfr[tn] = *(const bf16x8*)&cur[roff[tn] + (((ks * 4 + (lane >> 4)) ^ rsw[tn]) << 4)];
                        if (SAD_IP_ABL & 1) {
#pragma unroll
                            for (int tn = 0; tn < 4; ++tn) asm volatile("" ::"v"(bfr[tn]));
                            continue;
                        }
#pragma unroll
                        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                            for (int tn = 0; tn < 4; ++tn)
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[tn], a[tm][ks], acc[tm][tn], 0, 0, 0);
                    }
                }
            }
            static_assert(!DB || PAIR == 2, "the double-buffered form is used with the per-tile epilogue");
            if (!(SAD_IP_ABL & 4)) epi(nt, acc);
            if (block_any) __syncthreads();       // tile nt consumed by all, tile nt+1 written by all
        }
        return;
    }
    if (block_any) {
        load_tile(nt0);
#pragma unroll
        for (int i = 0; i < SCH; ++i) *(u32x4*)&Bs[soff[i]] = st[i];
        __syncthreads();
        if (nt0 + 1 < nt1) load_tile(nt0 + 1);
    }
    for (int nt = nt0; nt < nt1; ++nt) {
        f32x4 acc[2][4];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (block_any) {
            if (wave_any) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    bf16x8 bfr[4];
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        bfr[tn] = *(const bf16x8*)&Bs[roff[tn] + (((ks * 4 + (lane >> 4)) ^ rsw[tn]) << 4)];
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 4; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[tn], a[tm][ks], acc[tm][tn], 0, 0, 0);
                }
            }
            __syncthreads();                       // all waves finished reading tile nt
            if (nt + 1 < nt1) {
#pragma unroll
                for (int i = 0; i < SCH; ++i) *(u32x4*)&Bs[soff[i]] = st[i];
            }
            __syncthreads();
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            if constexpr (PAIR == 1) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
                    epi(tm, nt * BN + 32 * pr + 8 * (lane >> 4), acc[tm][2 * pr], acc[tm][2 * pr + 1]);
            } else if constexpr (PAIR == 0) {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) epi(tm, nt * BN + 16 * tn + 4 * (lane >> 4), acc[tm][tn]);
            }
        }
        if constexpr (PAIR == 2) epi(nt, acc);
        if (block_any && nt + 2 < nt1) load_tile(nt + 2);
    }
}

// ------------------------------------------------------------------------------------------------
// Wave-specialised form of rs_gemm (K1'): 8 COMPUTE waves (A fragments in registers, MFMAs,
// epilogue stores) + 4 LOADER waves (Bt tiles -> LDS) in one 768-thread workgroup.
// vmcnt retires in order per wave, so a wave that both prefetches tiles and stores results has to
// drain its stores whenever it waits for a tile; here the compute waves issue nothing but stores
// after their one-time fragment loads and never wait on vector memory, and the loaders keep two
// tiles of loads in flight (two register sets) while one tile is being written to the spare LDS
// buffer.  Two LDS tile buffers, one workgroup barrier per tile.
// PAIR / epi: as in rs_gemm.  Bs: 2 * 64 * KS * 64 bytes.
// ------------------------------------------------------------------------------------------------
template <int KS, bool PAIR, class Epi>
__device__ __forceinline__ void rs_gemm_ws(const bf16raw* __restrict__ A, int M, const int (&mrow)[2], bool wave_any,
                                           const bf16raw* __restrict__ Bt, int nt0, int nt1, unsigned char* Bs, Epi epi) {
    constexpr int EP = KS * 32, BN = 64, NLOAD = 256;
    constexpr int ROWB = EP * 2;              // bytes per Bt row
    constexpr int CPR = EP / 8;               // 16-byte chunks per row
    constexpr int SCH = BN * CPR / NLOAD;     // staging chunks per loader thread and tile
    constexpr int TILEB = BN * ROWB;
    static_assert((BN * CPR) % NLOAD == 0, "tile must split evenly over the loader waves");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave >= 8) {
        // ---------------------------------- loader waves ---------------------------------------
        const int lt = tid - 512;
        int soff[SCH], goff[SCH];
#pragma unroll
        for (int i = 0; i < SCH; ++i) {
            const int g = lt + i * NLOAD;
            const int vloc = g / CPR, ch = g - vloc * CPR;
            const int rho = PAIR == 1 ? (vloc & 32) + 16 * ((vloc >> 2) & 1) + 4 * ((vloc >> 3) & 3) + (vloc & 3) : vloc;
            soff[i] = rho * ROWB + ((ch ^ ((rho >> 1) & 7)) << 4);
            goff[i] = g * 8;
        }
        u32x4 st[2][SCH];
        auto load_tile = [&](auto par, int nt) {      // clamped, never skipped: static operation counts
            constexpr int P = decltype(par)::value;
            const bf16raw* base = Bt + (long long)(nt < nt1 ? nt : nt1 - 1) * BN * EP;
#pragma unroll
            for (int i = 0; i < SCH; ++i) st[P][i] = *(const u32x4*)(base + goff[i]);
        };
        auto to_lds = [&](auto par, int nt) {         // tile nt (register set nt&1) -> LDS buffer nt&1
            constexpr int P = decltype(par)::value;
            unsigned char* buf = Bs + ((nt - nt0) & 1) * TILEB;
#pragma unroll
            for (int i = 0; i < SCH; ++i) *(u32x4*)&buf[soff[i]] = st[P][i];
        };
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        load_tile(P0{}, nt0);
        load_tile(P1{}, nt0 + 1);
        to_lds(P0{}, nt0);
        load_tile(P0{}, nt0 + 2);
        __syncthreads();                              // tile nt0 ready
        for (int nt = nt0; nt < nt1; nt += 2) {
            // during compute(nt): tile nt+1 -> other buffer, request nt+3
            to_lds(P1{}, nt + 1);
            load_tile(P1{}, nt + 3);
            __syncthreads();
            if (nt + 1 < nt1) {
                to_lds(P0{}, nt + 2);
                load_tile(P0{}, nt + 4);
                __syncthreads();
            }
        }
        return;
    }
    // -------------------------------------- compute waves ----------------------------------------
    bf16x8 a[2][KS];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (wave_any && mrow[tm] < M) v = *(const u32x4*)(A + (long long)mrow[tm] * EP + ks * 32 + (lane >> 4) * 8);
            a[tm][ks] = __builtin_bit_cast(bf16x8, v);
        }
    int roff[4], rsw[4];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
        const int rho = tn * 16 + (lane & 15);
        roff[tn] = rho * ROWB;
        rsw[tn] = (rho >> 1) & 7;
    }
    __syncthreads();                                  // tile nt0 ready
    for (int nt = nt0; nt < nt1; ++nt) {
        const unsigned char* buf = Bs + ((nt - nt0) & 1) * TILEB;
        f32x4 acc[2][4];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wave_any) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 bfr[4];
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    bfr[tn] = *(const bf16x8*)&buf[roff[tn] + (((ks * 4 + (lane >> 4)) ^ rsw[tn]) << 4)];
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[tn], a[tm][ks], acc[tm][tn], 0, 0, 0);
            }
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            if constexpr (PAIR) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
                    epi(tm, nt * BN + 32 * pr + 8 * (lane >> 4), acc[tm][2 * pr], acc[tm][2 * pr + 1]);
            } else {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) epi(tm, nt * BN + 16 * tn + 4 * (lane >> 4), acc[tm][tn]);
            }
        }
        __syncthreads();                              // tile nt consumed; tile nt+1 written by the loaders
    }
}

// K1' on rs_gemm_ws (Ep == 32*KS): dS[m][v] = sum_e dX[m][e] * W_emb[v][e], every (m, v) written in the
// caller's dtype/strides (rows with t >= len come out as exact zeros because their dX rows are zero).
template <int KS, int KIND>
__global__ __launch_bounds__(768) void k_softemb_bwd_ws(Shape s, const bf16raw* __restrict__ DX,
                                                       const unsigned char* __restrict__ need,
                                                       const bf16raw* __restrict__ wemb, void* __restrict__ dS,
                                                       long long sT, long long sB, int nt_per) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_lds[];   // 2 * 64 * KS * 64 bytes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 256;
    const int ntiles = s.Vp / 64;
    const int nt0 = blockIdx.y * nt_per;
    const int nt1 = nt0 + nt_per < ntiles ? nt0 + nt_per : ntiles;
    const int M = s.M, V = s.V;
    int mrow[2];
    long long obase[2];
    int rneed = 0;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int m = m0 + (wave & 7) * 32 + tm * 16 + (lane & 15);
        mrow[tm] = m;
        obase[tm] = 0;
        if (m < M) {
            const int t = m / s.B, b = m - t * s.B;
            obase[tm] = (long long)t * sT + (long long)b * sB;
            if (wave < 8) rneed |= need[m];
        }
    }
    const bool wave_any = __any(rneed != 0);
    if constexpr (KIND == SRC_F32) {
        rs_gemm_ws<KS, false>(DX, M, mrow, wave_any, wemb, nt0, nt1, ws_lds, [&](int tm, int n, f32x4 v) {
            if (mrow[tm] >= M || n >= V) return;
            float* p = (float*)dS + obase[tm] + n;
            if (n + 4 <= V) {
                *(f32x4u*)p = v;
            } else {
                for (int e = 0; e < 4; ++e)
                    if (n + e < V) p[e] = v[e];
            }
        });
    } else {
        rs_gemm_ws<KS, true>(DX, M, mrow, wave_any, wemb, nt0, nt1, ws_lds, [&](int tm, int n, f32x4 lo, f32x4 hi) {
            if (mrow[tm] >= M || n >= V) return;
            bf16raw* p = (bf16raw*)dS + obase[tm] + n;
            if (n + 8 <= V) {
                *(u32x4h*)p = pack_bf16x8(lo, hi);
            } else {
                for (int e = 0; e < 4; ++e) {
                    if (n + e < V) p[e] = f32_to_bf16(lo[e]);
                    if (n + 4 + e < V) p[4 + e] = f32_to_bf16(hi[e]);
                }
            }
        });
    }
}

// ------------------------------------------------------------------------------------------------
// K1' for bf16 dS with LOADER, COMPUTE and STORE waves (768 threads):
//   waves 0-7  compute: dX fragments in registers, 80 MFMAs per 64-column tile, results -> LDS
//   waves 8-9  loaders: W_emb tile nt+1 -> LDS while tile nt is multiplied
//   waves 10-11 storers: tile nt-1 from LDS -> dS, 128 contiguous bytes per row and 8 rows per
//              instruction (the accumulator layout only gives 64-byte pieces of 16 rows)
// A wave that multiplies and also stores alternates between the two (store issue stalls when the
// memory pipe is full, and all waves meet at the tile barrier), so the kernel ran at store time +
// MFMA time; with the roles split the tile time is the maximum of the three.
// LDS: 2 W_emb tile buffers (2 x 40 KB at E = 300) + 2 output tile buffers (2 x 32 KB), one
// workgroup barrier per tile.
// 64-byte-aligned stores (round 2): the rows of dS start at arbitrary 2-byte offsets (V = 9353: pitch 18706 B), so a
// tile's 128-byte row segment used to straddle 64-byte blocks that the neighbouring tile completed an iteration
// later, and the two partial writes reached HBM separately (WRITE_SIZE 1.3 x the tensor; 276 us at V = 9353
// against 187 us at V = 9344).  Now the compute waves write their results into the LDS tile shifted by the
// row's misalignment (a register funnel between lanes 16 apart, so LDS writes stay aligned 16-byte chunks), the
// wrapped tail of a tile waits in registers for the next tile's buffer, and the store waves issue plain aligned
// stores of two whole 64-byte blocks per row: 280 -> 237 us back to back, 222 -> 210 us inside the step.
// Tried on the way (all measured, all slower): re-aligning in the store waves (two LDS reads + funnel per chunk:
// 342 us), element-wise shifted LDS writes from the compute waves (48 ds_write_b16 per lane: 302 us in the step),
// every wave storing a share of the tile (321 us), four compute waves of 64 rows (no change).
// ------------------------------------------------------------------------------------------------
// Timing-only ablations (tools/k1abl.py ABL_KERNEL=k1b, -DSAD_K1B_ABL=<bits>; results are garbage):
// 1 store waves skip the global stores, 2 compute waves skip the MFMAs, 4 loader waves skip the W_emb loads.
#ifndef SAD_K1B_ABL
#define SAD_K1B_ABL 0
#endif
template <int KS>
__global__ __launch_bounds__(768) void k_softemb_bwd_ws2(Shape s, const bf16raw* __restrict__ DX,
                                                        const unsigned char* __restrict__ need,
                                                        const bf16raw* __restrict__ wemb, bf16raw* __restrict__ dS,
                                                        long long sT, long long sB, int nt_per) {
    constexpr int EP = KS * 32, BN = 64;
    constexpr int ROWB = EP * 2, CPR = EP / 8, TILEB = BN * ROWB;
    constexpr int NLOAD = 128, SCH = BN * CPR / NLOAD;        // loader threads, chunks per thread and tile
    constexpr int OUTB = 256 * BN * 2;                        // one output tile: [256][64] bf16
    static_assert((BN * CPR) % NLOAD == 0, "tile must split evenly over the loader waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char w2_lds[];
    unsigned char* Bs = w2_lds;                                // 2 x TILEB
    unsigned char* Os = w2_lds + 2 * TILEB;                    // 2 x OUTB, chunk c of row r at r*128 + ((c ^ (r&7)) << 4)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 256;
    const int ntiles = s.Vp / 64;
    const int nt0 = blockIdx.y * nt_per;
    const int nt1 = nt0 + nt_per < ntiles ? nt0 + nt_per : ntiles;
    const int ntl = nt1 - nt0;
    const int M = s.M, V = s.V;

    if (wave >= 10) {
        // -------------------------------------- store waves --------------------------------------
        // The LDS tile of iteration j holds, for row r, the columns [64 j - phi_r, 64 j + 64 - phi_r) of the row,
        // phi_r = the row's offset from a 64-byte boundary in elements (the compute waves write their results
        // shifted, see below): lane (r, c) stores chunk c to an address that is 16-byte aligned, and the 8 lanes of
        // a row cover two whole 64-byte blocks.  Columns outside [64 nt0, min(64 nt1, V)) are clipped: the head of
        // the first tile has no data, the flush tile (iteration ntl + 1) only holds the last phi_r columns.
        const int ls = tid - 640;
        bf16raw* prow[16];                                     // row start minus the row's misalignment (64-byte aligned), or null
        int phs[16];                                           // that misalignment, elements
        const int c = ls & 7;
        const unsigned dS_lo = (unsigned)((uintptr_t)dS >> 1);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + (ls >> 3) + 16 * i;
            const int t = m / s.B, b = m - t * s.B;
            const long long ob = (long long)t * sT + (long long)b * sB;
            phs[i] = (int)((dS_lo + (unsigned)ob) & 31u);
            prow[i] = m < M ? dS + ob - phs[i] + c * 8 : nullptr;
        }
        const int col_lo = nt0 * BN, col_hi = nt1 * BN < V ? nt1 * BN : V;
        K1P_DECL
        __syncthreads();                                       // prologue barrier
        for (int it = 0; it <= ntl + 1; ++it) {
            K1P_T(0)
            K1P_T(1)
            if (it >= 1) {
                const unsigned char* buf = Os + ((it - 1) & 1) * OUTB;
                // all 16 LDS reads first: behind the per-row branches below they are not hoisted, and sixteen
                // read -> wait -> store rounds cost 2.7 us per tile
                u32x4 vv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = (ls >> 3) + 16 * i;
                    vv[i] = *(const u32x4*)&buf[r * 128 + ((c ^ (r & 7)) << 4)];
                }
                const int seg = (nt0 + it - 1) * BN;             // first stream position of this tile
                if (SAD_K1B_ABL & 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(vv[i]));
                } else if (seg - 31 >= col_lo && seg + BN <= col_hi) {
                    // interior tile (workgroup-uniform): every chunk lies inside the column range
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (prow[i] != nullptr) *(u32x4*)(prow[i] + seg) = vv[i];
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        if (prow[i] == nullptr) continue;
                        bf16raw* p = prow[i] + seg;            // columns col0 .. col0 + 7 of the row
                        const int col0 = seg + c * 8 - phs[i];
                        const int lo = col0 > col_lo ? col0 : col_lo, hi = col0 + 8 < col_hi ? col0 + 8 : col_hi;
                        if (hi <= lo) continue;
                        const u32x4 v = vv[i];
                        if (hi - lo == 8) {
                            *(u32x4*)p = v;
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (col0 + e >= lo && col0 + e < hi) p[e] = (bf16raw)((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
                        }
                    }
                }
            }
            K1P_T(2)
            __syncthreads();
            K1P_T(3)
            K1P_ACC
        }
        K1P_OUT2("storers: -, LDS reads + store issue, barrier")
        return;
    }
    if (wave >= 8) {
        // -------------------------------------- loader waves -------------------------------------
        const int lt = tid - 512;
        int soff[SCH], goff[SCH];
#pragma unroll
        for (int i = 0; i < SCH; ++i) {
            const int g = lt + i * NLOAD;
            const int vloc = g / CPR, ch = g - vloc * CPR;
            // pair permutation of rs_gemm: LDS row rho holds W_emb row vloc
            const int rho = (vloc & 32) + 16 * ((vloc >> 2) & 1) + 4 * ((vloc >> 3) & 3) + (vloc & 3);
            soff[i] = rho * ROWB + ((ch ^ ((rho >> 1) & 7)) << 4);
            goff[i] = g * 8;
        }
        u32x4 st[SCH];
        if (SAD_K1B_ABL & 4) memset(st, 0, sizeof(st));
        auto load_tile = [&](int k) {                          // clamped, never skipped
            if (SAD_K1B_ABL & 4) return;
            const bf16raw* base = wemb + (long long)(nt0 + (k < ntl ? k : ntl - 1)) * BN * EP;
#pragma unroll
            for (int i = 0; i < SCH; ++i) st[i] = *(const u32x4*)(base + goff[i]);
        };
        auto to_lds = [&](int k) {
            unsigned char* buf = Bs + (k & 1) * TILEB;
#pragma unroll
            for (int i = 0; i < SCH; ++i) *(u32x4*)&buf[soff[i]] = st[i];
        };
        load_tile(0);
        to_lds(0);
        load_tile(1);
        __syncthreads();                                       // prologue barrier: tile 0 ready
        K1P_DECL
        for (int it = 0; it <= ntl + 1; ++it) {
            K1P_T(0)
            if (it + 1 < ntl) {
                to_lds(it + 1);
                K1P_T(1)
                load_tile(it + 2);
            } else {
                K1P_T(1)
            }
            K1P_T(2)
            __syncthreads();
            K1P_T(3)
            K1P_ACC
        }
        K1P_OUT2("loaders: to_lds (wait + write), issue, barrier")
        return;
    }
    // ---------------------------------------- compute waves --------------------------------------
    int mrow[2];
    int rneed = 0;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        mrow[tm] = m0 + wave * 32 + tm * 16 + (lane & 15);
        if (mrow[tm] < M) rneed |= need[mrow[tm]];
    }
    const bool wave_any = __any(rneed != 0);
    bf16x8 a[2][KS];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (wave_any && mrow[tm] < M) v = *(const u32x4*)(DX + (long long)mrow[tm] * EP + ks * 32 + (lane >> 4) * 8);
            a[tm][ks] = __builtin_bit_cast(bf16x8, v);
        }
    // B-fragment addresses: W row rho = 16 tn + (lane & 15), chunk (4 ks + (lane >> 4)) ^ ((rho >> 1) & 7).  The
    // swizzle term does not depend on tn, its low two bits only touch the lane's quarter index and its bit 2 swaps
    // k-steps 2j and 2j + 1: two base registers and immediate offsets instead of eight address registers.
    const int rswz = ((lane & 15) >> 1) & 7, sflip = (rswz >> 2) & 1;
    const int b_even = (lane & 15) * ROWB + (((lane >> 4) ^ (rswz & 3)) << 4) + sflip * 64;   // + 64 ks          (ks even)
    const int b_odd = b_even + 64 - sflip * 128;                                              // + 64 (ks - 1)    (ks odd)
    // Shifted output: element (row r, tile column x) goes to LDS position y = x + phi_r of the row's stream.  The
    // shift is done in registers so that every LDS write stays an aligned 16-byte chunk: output chunk O_i =
    // last f elements of raw chunk i-1 ++ first 8-f of raw chunk i (f = phi_r & 7; raw chunk -1 = chunk 7 of the
    // previous tile), written at chunk position i + (phi_r >> 3).  The left neighbour of a lane's chunk lives 16
    // lanes below (lanes 16 apart hold neighbouring 8-column groups of one row): one ds_bpermute per dword.
    // Positions 8 and above belong to the head of the NEXT tile's buffer, which the store waves are reading during
    // this iteration: those chunks (upper half of the tile only, phi_r < 32) wait in registers until the top of
    // the next iteration.
    const int q4 = lane >> 4;
    int phi[2];                                                // misalignment of this lane's two rows, elements (0..31)
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int m = mrow[tm] < M ? mrow[tm] : 0;
        const int t = m / s.B, b = m - t * s.B;
        phi[tm] = (int)(((unsigned)((uintptr_t)dS >> 1) + (unsigned)((long long)t * sT + (long long)b * sB)) & 31u);
    }
    // window of 16 bytes at byte offset 16 - 2 f in (left, cur): barrel shift by dwords with bit-selects, then by
    // 0 or 16 bits (the masks are recomputed per use: kept across the tile loop they cost registers that spill)
    auto funnel = [&](const u32x4& L, const u32x4& C, int f) {
        const int sb = 16 - 2 * f, d = (sb >> 2) & 3;
        const unsigned m1 = (d & 1) ? 0xffffffffu : 0u, m2 = (d & 2) ? 0xffffffffu : 0u, sh16 = (unsigned)(sb & 2) * 8u;
        const unsigned x[8] = {L[0], L[1], L[2], L[3], C[0], C[1], C[2], C[3]};
        unsigned y[7], z[5];
#pragma unroll
        for (int i = 0; i < 7; ++i) y[i] = (x[i + 1] & m1) | (x[i] & ~m1);
#pragma unroll
        for (int i = 0; i < 5; ++i) z[i] = (y[i + 2] & m2) | (y[i] & ~m2);
        u32x4 o;
#pragma unroll
        for (int w = 0; w < 4; ++w) o[w] = f == 0 ? C[w] : __builtin_amdgcn_alignbit(z[w + 1], z[w], sh16);
        return o;
    };
    auto from_below = [&](const u32x4& v) {                    // the value of the lane 16 below (lane 0..15: of lane + 48)
        u32x4 o;
        const int src = ((lane - 16) & 63) << 2;
#pragma unroll
        for (int w = 0; w < 4; ++w) o[w] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)v[w]);
        return o;
    };
    u32x4 prev_hi[2], held[2];                                 // raw upper chunk of the previous tile; shifted chunk waiting to wrap
    prev_hi[0] = prev_hi[1] = held[0] = held[1] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();                                           // prologue barrier
    K1P_DECL
    for (int it = 0; it <= ntl + 1; ++it) {
        K1P_T(0)
        asm volatile("" : "+v"(phi[0]), "+v"(phi[1]));         // nothing derived from phi is hoisted out of the loop
        int fsh[2], qsh[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            fsh[tm] = phi[tm] & 7;
            qsh[tm] = phi[tm] >> 3;
        }
        if (it >= 1 && it <= ntl) {                            // wrapped chunks of tile it - 1 -> head of buffer it & 1
            unsigned char* ob = Os + (it & 1) * OUTB;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const int r = wave * 32 + tm * 16 + (lane & 15);
                const int g = 4 + q4 + qsh[tm];
                if (g >= 8) *(u32x4*)&ob[r * 128 + (((g - 8) ^ (r & 7)) << 4)] = held[tm];
            }
        }
        K1P_T(1)
        if (it <= ntl) {
            const unsigned char* buf = Bs + (it & 1) * TILEB;
            unsigned char* ob = Os + (it & 1) * OUTB;
            f32x4 acc[2][4];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
            // The tile is multiplied in two 32-column halves; the pack / funnel / LDS write of the lower half sits
            // between them in one basic block, so that it overlaps the MFMAs of the upper half (both compute waves
            // of a SIMD meet the tile barrier in phase: without this the MFMA pipe idles through the epilogue).
            auto half = [&](int pr) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    bf16x8 bfr[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        bfr[h] = *(const bf16x8*)&buf[((ks & 1) ? b_odd + 64 * (ks - 1) : b_even + 64 * ks) + (2 * pr + h) * 16 * ROWB];
                    if (SAD_K1B_ABL & 2) {
                        asm volatile("" ::"v"(bfr[0]), "v"(bfr[1]));
                        continue;
                    }
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            acc[tm][2 * pr + h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[h], a[tm][ks], acc[tm][2 * pr + h], 0, 0, 0);
                }
            };
            u32x4 lo[2];
            // raw chunk q4 (columns 8 q4 .. + 7): left neighbour = chunk q4 - 1 of the lane 16 below; for q4 == 0 the
            // lane 48 above holds chunk 7 of the PREVIOUS tile (its upper chunk)
            auto epi_lo = [&]() {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    const int r = wave * 32 + tm * 16 + (lane & 15);
                    lo[tm] = pack_bf16x8(acc[tm][0], acc[tm][1]);
                    const u32x4 left_lo = from_below(q4 == 3 ? prev_hi[tm] : lo[tm]);
                    const u32x4 o_lo = funnel(left_lo, lo[tm], fsh[tm]);
                    const int g_lo = q4 + qsh[tm];                                 // <= 6
                    *(u32x4*)&ob[r * 128 + ((g_lo ^ (r & 7)) << 4)] = o_lo;
                }
            };
            // raw chunk 4 + q4 (columns 32 + 8 q4 .. + 7): left neighbour = chunk 3 + q4 (q4 == 0: the lower chunk
            // of the lane 48 above)
            auto epi_hi = [&]() {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    const int r = wave * 32 + tm * 16 + (lane & 15);
                    const u32x4 hi = pack_bf16x8(acc[tm][2], acc[tm][3]);
                    const u32x4 left_hi = from_below(q4 == 3 ? lo[tm] : hi);
                    prev_hi[tm] = hi;
                    const u32x4 o_hi = funnel(left_hi, hi, fsh[tm]);
                    const int g_hi = 4 + q4 + qsh[tm];
                    if (g_hi < 8) *(u32x4*)&ob[r * 128 + ((g_hi ^ (r & 7)) << 4)] = o_hi;
                    held[tm] = o_hi;
                }
            };
            if (wave_any && it < ntl) {
                half(0);
                epi_lo();
                half(1);
                epi_hi();
            } else {
                // rows none of whose tokens is needed, and the flush iteration ntl (zero "raw" chunks push the last f
                // elements of the last tile out)
                epi_lo();
                epi_hi();
            }
        }
        K1P_T(2)
        __syncthreads();
        K1P_T(3)
        K1P_ACC
    }
    K1P_OUT2("compute: -, MFMA + LDS write, barrier")
}

// K2a + K3 over the COMPACTED list of needed token rows (k_row_lists, list 0), work split evenly:
// a ragged batch leaves ~45 % of the time-major rows un-needed, and with one workgroup per 256
// consecutive rows the launch ran at the speed of its full tiles while the workgroups of late time
// steps idled (128 us at cfg3 for 54 % of the rows).  Here the units (256-row tile of the list, 64-column
// tile of Wcat) are ordered tile-major and cut into gridDim.x equal contiguous ranges; a workgroup
// runs the A-stationary GEMM (rs_gemm) over each (tile, column range) segment of its range (at most a few
// tiles: the X fragments are re-loaded per segment).  N is split, not K: no partial sums.
// The output tile is turned through a wave-private LDS scratch: the accumulator layout gives 16-byte pieces
// of 16 different rows per quarter wave; after the turn a store instruction writes full 256-byte row segments
// of 4 rows.  No extra workgroup barrier: a wave's LDS operations execute in program order.
template <int KS>
__global__ __launch_bounds__(512) void k_inproj_fwd_rs3(Shape s, const bf16raw* __restrict__ X,
                                                       const int* __restrict__ rows, const int* __restrict__ nrows,
                                                       const bf16raw* __restrict__ wcat, const float* __restrict__ bcat,
                                                       float* __restrict__ G, float* __restrict__ Kk,
                                                       unsigned short* __restrict__ GA) {
    // GA != NULL (the persistent recurrences run, sad_uses_half_gates): the gate pre-activations leave as fp16
    // RECORDS in SAD_WS_GA -- the record the forward recurrence's lane for (sentence, unit quad) reads and then
    // overwrites with its activated gates -- instead of fp32 rows of G: half the bytes of this store-bound kernel,
    // and in the accumulator layout a lane already holds exactly one record per row (4 gate tiles x 4 units), so the
    // gate tiles skip the LDS turn.  The attention keys (last Hp columns) stay fp32 rows of Kk.
    extern __shared__ __attribute__((aligned(16))) unsigned char ip_lds[];
    unsigned char* Bs = ip_lds;                                          // 2 x 64 * KS * 64 bytes (two W tile buffers)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* scr = ip_lds + 2 * 64 * KS * 64 + wave * 8192;        // [32 rows][64 cols] fp32, 256-byte rows
    const int M = s.M, G2 = 2 * s.Gp, ntn = s.Ncat / 64;
    const long long gld = s.Gld, hp = s.Hp;
    const int n = nrows[0] < M ? nrows[0] : M;                           // list 0: rows with need != 0
    const long long U = (long long)((n + 255) / 256) * ntn;
    long long u = U * blockIdx.x / gridDim.x;
    const long long u1 = U * (blockIdx.x + 1) / gridDim.x;
    const int sc = lane & 15;
    while (u < u1) {
        const int tile = (int)(u / ntn), nt0 = (int)(u - (long long)tile * ntn);
        const int nt1 = (long long)(ntn - nt0) < u1 - u ? ntn : nt0 + (int)(u1 - u);
        u += nt1 - nt0;
        const int j0 = tile * 256 + wave * 32;
        int mrow[2];
        int any = 0;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const int j = j0 + tm * 16 + (lane & 15);
            mrow[tm] = j < n ? rows[j] : M;
            any |= j < n;
        }
        const bool wave_any = __any(any);
        // store side: lane -> list rows (lane >> 4) + 4 i of the wave's 32, 16-byte chunk lane & 15 of the 256-byte row
        int srow[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = j0 + (lane >> 4) + 4 * i;
            srow[i] = j < n ? rows[j] : -1;
        }
        // record base of this lane's two rows (fragment side): everything of ga_index but direction and unit quad
        long long garow[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const int m = mrow[tm] < M ? mrow[tm] : 0, t = m / s.B, b = m - t * s.B;
            garow[tm] = ga_index(t, 0, b, 0, s.B, s.Hp);
        }
        const long long gadir = (long long)((s.B + 15) >> 4) * (s.Hp >> 2) * 256;     // ga_index(t, 1, ..) - ga_index(t, 0, ..)
        f32x4 bias[4];                                                   // this tile's bias, loaded by pre()
        rs_gemm<KS, 2, true>(X, M, mrow, wave_any, 1, wcat, nt0, nt1, Bs, [&](int nt) {
            if (!wave_any) return;
            if (GA != nullptr && nt * 64 < G2) {                             // record layout: columns tn * 16 + 4 q ..
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) bias[tn] = *(const f32x4*)&bcat[nt * 64 + tn * 16 + 4 * (lane >> 4)];
            } else {                                                         // row layout (after the LDS turn): columns 4 sc ..
                bias[0] = *(const f32x4*)&bcat[nt * 64 + sc * 4];
                bias[1] = bias[2] = bias[3] = bias[0];
            }
        }, [&](int nt, const f32x4 (&acc)[2][4]) {
            if (!wave_any) return;
            // retire the bias loads HERE, unconditionally: used first inside the row predicates below, they stay
            // "pending" on the paths around them and hipcc puts s_waitcnt vmcnt(0) in front of EVERY store -- each
            // store then waits for the one before it to complete
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) asm volatile("" : "+v"(bias[tn]));
            if (GA != nullptr && nt * 64 < G2) {
                const int dir = nt * 64 >= s.Gp ? 1 : 0, ub = nt - dir * (s.Gp >> 6), q = lane >> 4;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
                    if (mrow[tm] < M)
                        ga_store(GA + garow[tm] + dir * gadir + (long long)(ub * 4 + q) * 256, acc[tm][0] + bias[0],
                                 acc[tm][1] + bias[1], acc[tm][2] + bias[2], acc[tm][3] + bias[3]);
                return;
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const int r = tm * 16 + (lane & 15);
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    const int c = tn * 4 + (lane >> 4);
                    *(f32x4*)&scr[r * 256 + ((c ^ (r & 15)) << 4)] = acc[tm][tn];
                }
            }
            const int nn = nt * 64 + sc * 4;
            float* base = nn < G2 ? G + nn : Kk + (nn - G2);
            const long long ld = nn < G2 ? gld : hp;       // (selected between the Shape fields themselves, hipcc fetched
                                                           // the kernel argument with a per-lane global load: vmcnt(0) here)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = (lane >> 4) + 4 * i;
                const f32x4 v = *(const f32x4*)&scr[r * 256 + ((sc ^ (r & 15)) << 4)];
                // written once, read once by the recurrence much later: keep it out of the L2's way
                if (srow[i] >= 0) __builtin_nontemporal_store(v + bias[0], (f32x4*)(base + (long long)srow[i] * ld));
            }
        });
    }
}

// ------------------------------------------------------------------------------------------------
// Weight packing: fp32 state_dict tensors -> padded / permuted / transposed bf16 (or fp32) images.
// dst[r][c] = src[sr][sc] with (sr, sc) = TR ? (cmap(c), rmap(r)) : (rmap(r), cmap(c)); -1 -> 0.
// ------------------------------------------------------------------------------------------------
enum { MAP_ID = 0, MAP_GATE = 1, MAP_HCAT = 2 };
struct IdxMap {
    int kind, limit, H, Hp;
};
__device__ __forceinline__ int map_idx(const IdxMap& m, int i) {
    if (m.kind == MAP_ID) return i < m.limit ? i : -1;
    if (m.kind == MAP_GATE) {
        const int ub = i >> 6, g = (i & 63) >> 4, u = i & 15, unit = ub * 16 + u;
        return unit < m.H ? g * m.H + unit : -1;
    }
    const int d = i / m.Hp, j = i - d * m.Hp;
    return j < m.H ? d * m.H + j : -1;
}
template <typename OutT>
__global__ void k_pack(OutT* __restrict__ dst, int rows, int cols, long long ldd, const float* __restrict__ src,
                       const float* __restrict__ src2, long long lds, IdxMap rmap, IdxMap cmap, int tr) {
    const long long total = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
        const int a = map_idx(rmap, r), bb = map_idx(cmap, c);
        float v = 0.f;
        if (a >= 0 && bb >= 0) {
            const long long si = tr ? (long long)bb * lds + a : (long long)a * lds + bb;
            v = src[si];
            if (src2 != nullptr) v += src2[si];
        }
        if constexpr (sizeof(OutT) == 2)
            dst[(long long)r * ldd + c] = f32_to_bf16(v);
        else
            dst[(long long)r * ldd + c] = v;
    }
}

// All packing jobs of sad_pack_weights in ONE launch (28 separate launches cost ~140 us per weight
// re-pack, which the unfrozen mode pays every optimizer step).  blockIdx.y = job.
struct PackJob {
    void* dst;
    const float* src;
    const float* src2;
    long long ldd, lds;
    IdxMap rmap, cmap;
    int rows, cols, tr, bf16;
};
constexpr int PACK_MAX_JOBS = 32;
struct PackJobs {
    PackJob j[PACK_MAX_JOBS];
};
__global__ __launch_bounds__(256) void k_pack_all(PackJobs jobs) {
    const PackJob& J = jobs.j[blockIdx.y];
    if (J.tr) {
        // transposing jobs go through 64 x 64 LDS tiles: lanes run along the SOURCE's fast axis when they load and
        // along the destination's when they store (element-wise, one side touched 64 cache lines per wave
        // instruction: 174 MB of HBM traffic and 50 us per re-pack, which the unfrozen mode pays every step)
        __shared__ float tile[64][65];
        const int tr_ = (J.rows + 63) >> 6, tc_ = (J.cols + 63) >> 6;
        const int lo = threadIdx.x & 63, hi = threadIdx.x >> 6;
        for (int t = blockIdx.x; t < tr_ * tc_; t += gridDim.x) {
            const int r0 = (t / tc_) << 6, c0 = (t % tc_) << 6;
            const int r = r0 + lo, a = r < J.rows ? map_idx(J.rmap, r) : -1;
#pragma unroll 4
            for (int cc = hi; cc < 64; cc += 4) {
                const int c = c0 + cc, bb = c < J.cols ? map_idx(J.cmap, c) : -1;
                float v = 0.f;
                if (a >= 0 && bb >= 0) {
                    const long long si = (long long)bb * J.lds + a;
                    v = J.src[si];
                    if (J.src2 != nullptr) v += J.src2[si];
                }
                tile[cc][lo] = v;
            }
            __syncthreads();
            const int c = c0 + lo;
#pragma unroll 4
            for (int rr = hi; rr < 64; rr += 4) {
                const int r2 = r0 + rr;
                if (r2 < J.rows && c < J.cols) {
                    const float v = tile[lo][rr];
                    if (J.bf16)
                        ((bf16raw*)J.dst)[(long long)r2 * J.ldd + c] = f32_to_bf16(v);
                    else
                        ((float*)J.dst)[(long long)r2 * J.ldd + c] = v;
                }
            }
            __syncthreads();
        }
        return;
    }
    const long long total = (long long)J.rows * J.cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / J.cols), c = (int)(i - (long long)r * J.cols);
        const int a = map_idx(J.rmap, r), bb = map_idx(J.cmap, c);
        float v = 0.f;
        if (a >= 0 && bb >= 0) {
            const long long si = J.tr ? (long long)bb * J.lds + a : (long long)a * J.lds + bb;
            v = J.src[si];
            if (J.src2 != nullptr) v += J.src2[si];
        }
        if (J.bf16)
            ((bf16raw*)J.dst)[(long long)r * J.ldd + c] = f32_to_bf16(v);
        else
            ((float*)J.dst)[(long long)r * J.ldd + c] = v;
    }
}

}  // namespace sad
