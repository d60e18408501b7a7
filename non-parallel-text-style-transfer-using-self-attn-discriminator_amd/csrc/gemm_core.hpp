// LDS-tiled bf16 MFMA GEMM core for gfx950:  C[m][n] = sum_k A[m][k] * Bt[n][k]   ("NT": both
// operands K-contiguous, which is the natural layout of torch Linear/LSTM weights [out,in]).
//
//  * 64-wide wavefronts, v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//  * The MFMA is issued with the operands SWAPPED (Bt fragment as the A operand, A fragment as the
//    B operand), so the accumulator holds C^T: lane l owns output row  m = l & 15  and the FOUR
//    CONSECUTIVE columns n = (l >> 4) * 4 + {0..3} of each 16x16 tile.  Epilogues therefore store
//    8/16 contiguous bytes per lane along n, and the LSTM epilogue finds the four gates of one
//    hidden unit in acc[tm][0..3] of the same lane (see gate_col()).
//  * global -> register -> LDS staging with NS K-tiles of global loads in flight (register ring,
//    statically indexed); LDS rows unpadded with XOR-swizzled 16-byte chunks (conflict-free).
//  * Operand access goes through loader functors so each op decides dtype conversion (fp32 soft
//    tokens -> bf16), row predicates (sequence length masks), gathers and bounds.
#pragma once
#include "common.hpp"
#include <type_traits>

namespace sad {

template <int BM_, int BN_, int WM_, int WN_, int NS_ = 1>
struct Cfg {
    static constexpr int BM = BM_, BN = BN_, BK = 64, WM = WM_, WN = WN_;
    static constexpr int NS = NS_;                          // K-tiles of global loads kept in flight
    static constexpr int NT = 64 * WM_ * WN_;               // threads per workgroup
    static constexpr int TM = BM_ / WM_ / 16;               // 16x16 tiles per wave along m
    static constexpr int TN = BN_ / WN_ / 16;               // ... along n
    static constexpr int LDT = BK;                          // LDS row stride, bf16 elements (128 B, unpadded)
    static constexpr int CH = BK / 8;                       // 16-byte chunks per tile row
    static constexpr int RSTEP = NT / CH;                   // tile rows covered per staging pass
    static constexpr int ACH = BM_ * CH / NT;               // A chunks per thread
    static constexpr int BCH = BN_ * CH / NT;               // B chunks per thread
    static_assert(BM_ % (WM_ * 16) == 0 && BN_ % (WN_ * 16) == 0, "tile/wave mismatch");
    static_assert((BM_ * CH) % NT == 0 && (BN_ * CH) % NT == 0, "staging must divide evenly");
};

// Row (within the block tile) and first column of this lane's 4-column group for tile (tm, tn).
template <class C>
__device__ __forceinline__ int frag_row(int wm, int tm, int lane) { return wm * (C::BM / C::WM) + tm * 16 + (lane & 15); }
template <class C>
__device__ __forceinline__ int frag_col(int wn, int tn, int lane) { return wn * (C::BN / C::WN) + tn * 16 + (lane >> 4) * 4; }

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// LDS image of a tile: unpadded 128-byte rows (8 chunks of 16 B), chunk index XOR-swizzled with
// (row>>1)&7.  With the MI355X bank rule for ds_read_b128 (64 banks; 16-lane groups of rows
// {0-3,12-15}@chunk c + {4-11}@chunk c^1) every fragment read is conflict-free, and so are the
// staging writes (8 consecutive threads fill the 8 chunks of one row).  Returns a bf16-element offset.
template <class C>
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * C::LDT + ((chunk ^ ((row >> 1) & 7)) << 3); }

// load_a(i, r, k) -> ARaw : chunk i of this thread = tile row r, 8 elements starting at absolute k
// cvt_a(ARaw)     -> u32x4 (8 bf16)
// load_b(i, r, k) -> u32x4
template <class C, class ARaw, class LA, class CA, class LB>
__device__ __forceinline__ void gemm_core(int K, LA load_a, CA cvt_a, LB load_b,
                                          f32x4 (&acc)[C::TM][C::TN], bf16raw* As, bf16raw* Bs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int r0 = tid / C::CH, kc = (tid % C::CH) * 8;
    ARaw ra[C::NS][C::ACH];
    u32x4 rb[C::NS][C::BCH];
    const int nk = (K + C::BK - 1) / C::BK;

    auto issue = [&](auto stage, int kt) {
        constexpr int s = decltype(stage)::value;
        const int k0 = kt * C::BK + kc;
#pragma unroll
        for (int i = 0; i < C::ACH; ++i) ra[s][i] = load_a(i, r0 + i * C::RSTEP, k0);
#pragma unroll
        for (int i = 0; i < C::BCH; ++i) rb[s][i] = load_b(i, r0 + i * C::RSTEP, k0);
    };
    auto stage_to_lds = [&](auto stage) {
        constexpr int s = decltype(stage)::value;
        __syncthreads();                       // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < C::ACH; ++i) *(u32x4*)&As[lds_off<C>(r0 + i * C::RSTEP, kc >> 3)] = cvt_a(ra[s][i]);
#pragma unroll
        for (int i = 0; i < C::BCH; ++i) *(u32x4*)&Bs[lds_off<C>(r0 + i * C::RSTEP, kc >> 3)] = rb[s][i];
        __syncthreads();
    };
    auto mfma_tile = [&]() {
#pragma unroll
        for (int kk = 0; kk < C::BK / 32; ++kk) {
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
    };

    // Prologue: NS K-tiles of global loads in flight (registers are the staging ring; the small
    // latency-bound GEMMs of the recurrence need several tiles outstanding to cover L2/HBM latency).
    static_for<C::NS>([&](auto stage) {
        if (decltype(stage)::value < nk) issue(stage, decltype(stage)::value);
    });
    // Steady state: branch-free (the loaders are branch-free too), so hipcc's waitcnt pass keeps
    // COUNTED s_waitcnt vmcnt(N): only the oldest stage is waited for, NS-1 tiles stay in flight.
    // (Any control flow around the loads makes it fall back to vmcnt(0) = a one-deep pipeline.)
    const int steady = nk > C::NS ? (nk - C::NS) / C::NS : 0;      // groups of NS tiles that all refill
    for (int g = 0; g < steady; ++g) {
        static_for<C::NS>([&](auto stage) {
            const int kt = g * C::NS + decltype(stage)::value;
            stage_to_lds(stage);
            issue(stage, kt + C::NS);
            mfma_tile();
        });
    }
    // Drain: at most 2*NS-1 tiles, refilling while tiles remain.
    for (int kt0 = steady * C::NS; kt0 < nk; kt0 += C::NS) {
        static_for<C::NS>([&](auto stage) {
            const int kt = kt0 + decltype(stage)::value;
            if (kt < nk) {                     // block-uniform
                stage_to_lds(stage);
                if (kt + C::NS < nk) issue(stage, kt + C::NS);
                mfma_tile();
            }
        });
    }
}

template <class C>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[C::TM][C::TN]) {
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// Aligned 16-byte load of 8 bf16 from a packed (padded, 16-B aligned) matrix; zero outside
// [rows) x [K).  Two forms, chosen per kernel by measurement (profiles/r01_ab_loaders.txt): the
// predicated form is faster everywhere except the store-heavy in-projection, where the
// branch-free one (clamped address + select) lets hipcc keep counted vmcnt waits.
__device__ __forceinline__ u32x4 load_packed(const bf16raw* base, long long ld, int row, int rows, int k, int K) {
    if (row < rows && k < K) return *(const u32x4*)(base + (long long)row * ld + k);
    return u32x4{0u, 0u, 0u, 0u};
}
__device__ __forceinline__ u32x4 load_packed_nb(const bf16raw* base, long long ld, int row, int rows, int k, int K) {
    const bool ok = row < rows && k < K;
    const u32x4 v = *(const u32x4*)(base + (ok ? (long long)row * ld + k : 0ll));
    return ok ? v : u32x4{0u, 0u, 0u, 0u};
}
// Row pointer that may be invalid (`ok` false).
__device__ __forceinline__ u32x4 load_row(const bf16raw* row, bool ok, const bf16raw* dummy, int k) {
    (void)dummy;
    if (ok) return *(const u32x4*)(row + k);
    return u32x4{0u, 0u, 0u, 0u};
}

struct Ident {
    __device__ __forceinline__ u32x4 operator()(u32x4 v) const { return v; }
};

}  // namespace sad
