// Parameter gradients of the discriminator ("unfrozen" mode: classifier training, and the mode the
// data-parallel all-reduce of SURVEY.md §8 d1/e1 refers to).  The reference gets these from autograd
// (loss.backward(), classifier/trainers.py:371-373) for the 19 live tensors of
// transfer/models.py:757-829; here each is one reduction over the token rows m = t*B + b:
//
//   dW[n1][n2] += sum_m A[m][n1] * B[m][n2]                         ("TN" GEMM, K = token rows)
//
// Both operands are stored row-major in m (the reduced index is the SLOW index), which is the
// opposite of what an MFMA fragment wants (8 consecutive k per lane).  gfx950's
// ds_read_b64_tr_b16 does the transpose on the way out of LDS: tiles are staged row-major exactly
// as they sit in HBM (coalesced 16-byte loads, no scatter) and read back column-major.
//
//  * LDS tile: [64 token rows][BN columns] bf16, row pitch 2*BN + 32 bytes.  pitch/32 is odd, so
//    the 8 rows a 32-lane half touches in one transposed read start 8 dwords apart mod 64 banks:
//    conflict-free.
//  * k assignment inside a 32-row slab: lane group g = lane>>4 holds rows {4g..4g+3} u {16+4g..16+4g+3}
//    (two transposed reads).  Any assignment works as long as both operands use the same one.
//  * Results are ACCUMULATED (+=) with fp32 atomics straight into the caller's state_dict-layout
//    tensors (gate rows un-permuted, padding dropped in the epilogue): split-K over token rows needs
//    no staging buffer, and gradient accumulation over micro-batches / the flat all-reduce bucket
//    come for free.
#pragma once
#include "kernels.hpp"

namespace sad {

typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int BN1_, int BN2_, int W1_, int W2_>
struct WCfg {
    static constexpr int BN1 = BN1_, BN2 = BN2_, W1 = W1_, W2 = W2_, BK = 64;
    static constexpr int LDS_BYTES = 2 * BK * (BN1_ * 2 + 32 + BN2_ * 2 + 32);   // two tile buffers
    static constexpr int NT = 64 * W1_ * W2_;
    static constexpr int T1 = BN1_ / W1_ / 16, T2 = BN2_ / W2_ / 16;
    static constexpr int P1 = BN1_ * 2 + 32, P2 = BN2_ * 2 + 32;   // LDS row pitch, bytes
    static constexpr int CPR1 = BN1_ / 8, CPR2 = BN2_ / 8;         // 16-byte chunks per tile row
    static constexpr int CH1 = BK * CPR1 / NT, CH2 = BK * CPR2 / NT;
    static_assert(BN1_ % 32 == 0 && BN2_ % 32 == 0, "pitch/32 must be odd");
    static_assert(BN1_ % (W1_ * 16) == 0 && BN2_ % (W2_ * 16) == 0, "tile/wave mismatch");
    static_assert((BK * CPR1) % NT == 0 && (BK * CPR2) % NT == 0, "staging must divide evenly");
};

// 8 k-values (rows r .. r+3 and r+16 .. r+19 of the slab) of one column; see file header.
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p, int pitch) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * pitch));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// The reduction runs over a COMPACTED list of contributing token rows (k_row_lists): 45 % of the
// rows of a ragged batch are padding and are never touched, and every 64-row tile is full, so the
// loop has no predicates and no control flow.
//
// row_of(j) -> int      : token row of list position j (j already clamped into the list); a memory
//                         read whose value is not looked at until two tiles later (latency hidden)
// load_a(row, c, ok) / load_b(row, c, ok) -> u32x4 : 8 bf16 of columns [c, c+8) of the block tile;
//                         zeros when !ok; BRANCH-FREE (load from a dummy address, then select)
// acc[t1][t2][e]: dW[n1 = w1*(BN1/W1) + 16*t1 + 4*(lane>>4) + e][n2 = w2*(BN2/W2) + 16*t2 + (lane&15)]
// (plain operand order: 16 lanes hold 16 CONSECUTIVE n2 of one row -> 64-byte atomic segments).
//
// Pipeline (tile = 64 list positions):
//   * two register slots hold the operand loads of tiles k+1 / k+2 (later k+3); the row-index reads
//     of the two tiles after those are in flight as well;
//   * two LDS buffers: while tile k is multiplied out of buffer k&1, tile k+1 is written into the
//     other one: ONE barrier per tile, and loads / LDS writes / MFMAs of a step share one basic
//     block, so the scheduler interleaves them;
//   * the loop issues a STATIC number of memory operations, so hipcc keeps counted
//     s_waitcnt vmcnt(N) and only the oldest tile is waited for.
// lds: W::LDS_BYTES = 2 * 64 * (P1 + P2) bytes, 16-byte aligned.
// fix_b(v, c) -> u32x4  : applied to a B chunk (columns [c, c+8) of the block tile) on its way from the register
//                         slot into LDS, i.e. once the load has long landed (touching the value where it is
//                         loaded would wait for the load there and end the pipelining)
template <class W, class RO, class LA, class LB, class FB>
__device__ __forceinline__ void wgrad_core(int j_begin, int j_end, RO row_of, LA load_a, LB load_b, FB fix_b,
                                           f32x4 (&acc)[W::T1][W::T2], unsigned char* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w1 = wave / W::W2, w2 = wave % W::W2;
    constexpr int ABYTES = W::BK * W::P1, BBYTES = W::BK * W::P2, BUF = ABYTES + BBYTES;
    u32x4 ra[2][W::CH1], rb[2][W::CH2];
    struct Rows {
        int v[W::CH1 + W::CH2];
    };
    auto rows_of = [&](int kt) {              // issue the row-index reads of tile kt (values unused here)
        const int j0 = j_begin + kt * W::BK;
        Rows p;
#pragma unroll
        for (int i = 0; i < W::CH1; ++i) {
            const int j = j0 + (tid + i * W::NT) / W::CPR1;
            p.v[i] = row_of(j < j_end ? j : j_end - 1);
        }
#pragma unroll
        for (int i = 0; i < W::CH2; ++i) {
            const int j = j0 + (tid + i * W::NT) / W::CPR2;
            p.v[W::CH1 + i] = row_of(j < j_end ? j : j_end - 1);
        }
        return p;
    };
    auto issue = [&](auto slot, int kt, const Rows& p) {
        constexpr int sl = decltype(slot)::value;
        const int j0 = j_begin + kt * W::BK;
#pragma unroll
        for (int i = 0; i < W::CH1; ++i) {
            const int g = tid + i * W::NT, r = g / W::CPR1, c = g - r * W::CPR1;
            ra[sl][i] = load_a(p.v[i], c * 8, j0 + r < j_end);
        }
#pragma unroll
        for (int i = 0; i < W::CH2; ++i) {
            const int g = tid + i * W::NT, r = g / W::CPR2, c = g - r * W::CPR2;
            rb[sl][i] = load_b(p.v[W::CH1 + i], c * 8, j0 + r < j_end);
        }
    };
    auto to_lds = [&](auto slot) {             // register slot s -> LDS buffer s
        constexpr int sl = decltype(slot)::value;
        unsigned char* As = lds + sl * BUF;
        unsigned char* Bs = As + ABYTES;
#pragma unroll
        for (int i = 0; i < W::CH1; ++i) {
            const int g = tid + i * W::NT, r = g / W::CPR1, c = g - r * W::CPR1;
            *(u32x4*)(As + r * W::P1 + c * 16) = ra[sl][i];
        }
#pragma unroll
        for (int i = 0; i < W::CH2; ++i) {
            const int g = tid + i * W::NT, r = g / W::CPR2, c = g - r * W::CPR2;
            *(u32x4*)(Bs + r * W::P2 + c * 16) = fix_b(rb[sl][i], c * 8);
        }
    };
    const int rowsel = 4 * (lane >> 4) + ((lane >> 2) & 3), p4 = (lane & 3) * 4;
    const int aoff = rowsel * W::P1 + (w1 * (W::BN1 / W::W1) + p4) * 2;
    const int boff = ABYTES + rowsel * W::P2 + (w2 * (W::BN2 / W::W2) + p4) * 2;
    auto mfma_tile = [&](auto buf) {
        const unsigned char* abase = lds + decltype(buf)::value * BUF + aoff;
        const unsigned char* bbase = lds + decltype(buf)::value * BUF + boff;
#pragma unroll
        for (int ks = 0; ks < W::BK / 32; ++ks) {
            bf16x8 fa[W::T1], fb[W::T2];
#pragma unroll
            for (int t1 = 0; t1 < W::T1; ++t1) fa[t1] = tr_frag(abase + ks * 32 * W::P1 + t1 * 32, W::P1);
#pragma unroll
            for (int t2 = 0; t2 < W::T2; ++t2) fb[t2] = tr_frag(bbase + ks * 32 * W::P2 + t2 * 32, W::P2);
#pragma unroll
            for (int t1 = 0; t1 < W::T1; ++t1)
#pragma unroll
                for (int t2 = 0; t2 < W::T2; ++t2)
                    acc[t1][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t1], fb[t2], acc[t1][t2], 0, 0, 0);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int nk = (j_end - j_begin + W::BK - 1) / W::BK;
    // Prologue.  Tiles past the end load nothing (ok = false: dummy address, zeros), so the loop
    // body needs no tail case.
    issue(S0{}, 0, rows_of(0));
    issue(S1{}, 1, rows_of(1));
    Rows p0 = rows_of(2), p1 = rows_of(3);
    to_lds(S0{});
    issue(S0{}, 2, p0);
    p0 = p1;
    p1 = rows_of(4);
    __syncthreads();
    // Tile k is multiplied out of buffer k&1 while tile k+1 moves from slot (k+1)&1 into the other
    // buffer and that slot is refilled with tile k+3.
    auto step = [&](auto par, int k) {
        constexpr int P = decltype(par)::value, Q = P ^ 1;
        to_lds(std::integral_constant<int, Q>{});
        issue(std::integral_constant<int, Q>{}, k + 3, p0);
        p0 = p1;
        p1 = rows_of(k + 5);
        mfma_tile(std::integral_constant<int, P>{});
        __syncthreads();
    };
    for (int k = 0; k < nk; k += 2) {
        step(S0{}, k);
        step(S1{}, k + 1);
    }
}

template <class W, class RO, class LA, class LB>
__device__ __forceinline__ void wgrad_core(int j_begin, int j_end, RO row_of, LA load_a, LB load_b,
                                           f32x4 (&acc)[W::T1][W::T2], unsigned char* lds) {
    wgrad_core<W>(j_begin, j_end, row_of, load_a, load_b, [](u32x4 v, int) { return v; }, acc, lds);
}

// This workgroup's slice [j_begin, j_end) of a row list of `count` entries split `zs` ways in whole
// 64-row tiles.  False: nothing to do.
__device__ __forceinline__ bool wgrad_slice(int count, int z, int zs, int* j_begin, int* j_end) {
    const int tiles = (count + 63) / 64;
    const int per = (tiles + zs - 1) / zs * 64;
    *j_begin = z * per;
    *j_end = min(count, *j_begin + per);
    return *j_begin < *j_end;
}

// Branch-free 16-byte load: `ok ? *p : 0`, reading `dummy` instead when !ok.
__device__ __forceinline__ u32x4 load16_if(const void* p, const void* dummy, bool ok) {
    const u32x4 v = *(const u32x4*)(ok ? p : dummy);
    return ok ? v : u32x4{0u, 0u, 0u, 0u};
}
__device__ __forceinline__ u32x4 load8f_if(const float* p, const float* dummy, bool ok) {
    const float* q = ok ? p : dummy;
    const f32x4 lo = *(const f32x4*)q, hi = *(const f32x4*)(q + 4);
    const u32x4 v = pack_bf16x8(lo, hi);
    return ok ? v : u32x4{0u, 0u, 0u, 0u};
}

template <class W>
__device__ __forceinline__ void wgrad_zero(f32x4 (&acc)[W::T1][W::T2]) {
#pragma unroll
    for (int t1 = 0; t1 < W::T1; ++t1)
#pragma unroll
        for (int t2 = 0; t2 < W::T2; ++t2) acc[t1][t2] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// epi(n1, n2, v): v = dW[n1][n2] (block-tile-relative indices); per call the 16 lanes of a group
// hold 16 consecutive n2 of one row n1.
template <class W, class Epi>
__device__ __forceinline__ void wgrad_epilogue(const f32x4 (&acc)[W::T1][W::T2], Epi epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w1 = wave / W::W2, w2 = wave % W::W2;
#pragma unroll
    for (int t1 = 0; t1 < W::T1; ++t1)
#pragma unroll
        for (int t2 = 0; t2 < W::T2; ++t2)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                epi(w1 * (W::BN1 / W::W1) + 16 * t1 + 4 * (lane >> 4) + e, w2 * (W::BN2 / W::W2) + 16 * t2 + (lane & 15),
                    acc[t1][t2][e]);
}

// dst[0..3] += v[0..3] for the elements below `limit` (hardware fp32 atomics)
__device__ __forceinline__ void atomic_add4(float* dst, int col, int limit, f32x4 v) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (col + e < limit) unsafeAtomicAdd(dst + e, v[e]);
}

// ------------------------------------------------------------------------------------------------
// Compacted lists of the token rows that contribute to a weight gradient, ascending, from the need
// bits:  list 0: need != 0 (embedding / input-projection / linear_emb gradients);
//        list 1: t >= 1 and t < len      (forward-direction  W_hh: rows that have an h_{t-1});
//        list 2: t + 1 < len             (reverse-direction  W_hh: rows that have an h_{t+1}).
// rows: int32 [3][M], nrows: int32 [3].  One 1024-thread workgroup (M is a few 10^4..10^5).
// ------------------------------------------------------------------------------------------------
enum { ROWS_ANY = 0, ROWS_FWD = 1, ROWS_REV = 2 };
// Each 256-thread workgroup compacts one 1024-row segment.  Its output offsets are the counts over
// all earlier rows, which it recounts itself from the need bytes (a few 10 KB, L2-resident) instead
// of waiting for its predecessors:  with C1(x) = #{m < x : bit0}, Cany(x) = #{m < x : need != 0},
//   offset_any = Cany(m0),  offset_fwd = C1(m0) - C1(min(B, m0)),  offset_rev = C1(min(M, m0+B)) - C1(min(M, B)).
__global__ __launch_bounds__(256) void k_row_lists(Shape s, const unsigned char* __restrict__ need, int* __restrict__ rows,
                                                  int* __restrict__ nrows) {
    __shared__ int red[4][4];
    __shared__ int wsum[3][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, M = s.M, B = s.B;
    const int m0 = blockIdx.x * 1024;
    const bool last = m0 + 1024 >= M;
    // limits: [0] Cany(m0)  [1] C1(m0)  [2] C1(min(B, M))  [3] C1(min(M, m0 + B));  the last segment
    // also produces the totals, so it counts up to M instead of m0.
    const int mend = last ? M : m0;
    const int lim[4] = {mend, mend, min(B, M), min(M, mend + B)};
    int cnt[4] = {0, 0, 0, 0};
    const int scan_to = lim[3] > lim[0] ? lim[3] : lim[0];
    for (int c = tid * 16; c < scan_to; c += 256 * 16) {
        const u32x4 w = *(const u32x4*)(need + c);            // need is padded to a multiple of 256 bytes
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int base = c + 4 * q;
            const unsigned anyb = (w[q] | (w[q] >> 1)) & 0x01010101u, b0 = w[q] & 0x01010101u;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const unsigned bits = l == 0 ? anyb : b0;
                const int left = lim[l] - base;               // rows of this dword below the limit
                const unsigned m = left >= 4 ? 0xffffffffu : (left <= 0 ? 0u : ((1u << (8 * left)) - 1u));
                cnt[l] += __popc(bits & m);
            }
        }
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        int v = cnt[l];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[l][wave] = v;
    }
    __syncthreads();
    int C[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) C[l] = red[l][0] + red[l][1] + red[l][2] + red[l][3];
    if (last && tid == 0) {
        nrows[ROWS_ANY] = C[0];
        nrows[ROWS_FWD] = C[1] - C[2];
        nrows[ROWS_REV] = C[3] - C[2];                        // rows m with m + B < M and bit0 of need[m + B]
    }
    if (last) {                                               // recount this segment's own offsets up to m0
        __syncthreads();
        int c2[4] = {0, 0, 0, 0};
        const int lim2[4] = {m0, m0, min(B, M), min(M, m0 + B)};
        for (int c = tid * 16; c < lim2[3] || c < lim2[0]; c += 256 * 16) {
            const u32x4 w = *(const u32x4*)(need + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int base = c + 4 * q;
                const unsigned anyb = (w[q] | (w[q] >> 1)) & 0x01010101u, b0 = w[q] & 0x01010101u;
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    const unsigned bits = l == 0 ? anyb : b0;
                    const int left = lim2[l] - base;
                    const unsigned m = left >= 4 ? 0xffffffffu : (left <= 0 ? 0u : ((1u << (8 * left)) - 1u));
                    c2[l] += __popc(bits & m);
                }
            }
        }
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            int v = c2[l];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) red[l][wave] = v;
        }
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 4; ++l) C[l] = red[l][0] + red[l][1] + red[l][2] + red[l][3];
    }
    const int off[3] = {C[0], m0 >= B ? C[1] - C[2] : 0, C[3] - C[2]};
    // own segment: 4 consecutive rows per thread
    int flag[3][4], mine[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 4 * tid + i;
        const int nd = m < M ? need[m] : 0;
        const int nn = m + B < M ? need[m + B] : 0;
        flag[ROWS_ANY][i] = nd != 0;
        flag[ROWS_FWD][i] = m >= B && (nd & 1);
        flag[ROWS_REV][i] = nn & 1;
#pragma unroll
        for (int l = 0; l < 3; ++l) mine[l] += flag[l][i];
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        int incl = mine[l];                                   // inclusive scan over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[l][wave] = incl;
        __syncthreads();
        int pos = off[l] + incl - mine[l];
        for (int wv = 0; wv < wave; ++wv) pos += wsum[l][wv];
        int* out = rows + (long long)l * M;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (flag[l][i]) out[pos++] = m0 + 4 * tid + i;
    }
}

// ------------------------------------------------------------------------------------------------
// k_prep + k_row_lists + k_tile_list in ONE launch (the three launches cost 7 + 13 + 4 us of a 1.27 ms
// step, and 24 of 396 us at batch 64), for M <= PREPL_MAX_M, B <= PREPL_MAX_B.  Nothing here waits on another workgroup: every workgroup rebuilds ALL need bytes in
// LDS from the mask and the lengths (a few 10 KB, L2-resident after the first reader), counts its own
// output offsets from them, and writes the need bytes, the three row lists and the tile-list entries of
// its own 1024-row segment.  Workgroup 0 also writes the clamped lengths, the list totals, applies the
// all-masked-row guard to the caller's mask and zeroes the recurrence's hand-off words.  A workgroup that
// reads mask[b][1] after workgroup 0 unmasked it sees the value the guard produces anyway (idempotent).
// The work is a chain of short dependent phases, so it is written for latency: 16 mask bytes per thread
// (one 16-byte load when the mask rows allow it), no branches per byte, wave scans on DPP row shifts (a
// ds_bpermute butterfly costs ~2 us per block reduction here), cross-wave sums as one LDS read + a row scan.
// LDS: [Mp] need bytes in (t, b) order | 16 bytes that swallow the stores of rows past M | [B] u8 guard flags.
// ------------------------------------------------------------------------------------------------
constexpr int PREPL_MAX_M = 65536, PREPL_MAX_B = 2048;
#ifndef PREP_STAMP            // phase time stamps of tools/micro/prep_phases.hip; nothing in the library
#define PREP_STAMP(i)
#endif
__host__ __device__ inline size_t prepl_lds_bytes(int M, int B) { return (((size_t)M + 15) & ~(size_t)15) + 16 + (((size_t)B + 15) & ~(size_t)15); }

// inclusive prefix sums over the 16 lanes of each DPP row / over the wave (total in lane 15 of a row / lane 63)
__device__ __forceinline__ int row_scan16(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);          // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);          // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);          // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);          // row_shr:8
    return v;
}
__device__ __forceinline__ int wave_scan64(int v) {
    v = row_scan16(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);          // row_bcast:15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);          // row_bcast:31 into rows 2, 3
    return v;
}
// number of set 0x01 bits among the first n (0..16) bytes of a 16-byte chunk
__device__ __forceinline__ int popc_low_bytes(u32x4 bits, int n) {
    int r = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int nq = n - 4 * q;
        nq = nq < 0 ? 0 : (nq > 4 ? 4 : nq);
        const unsigned keep = ~((0xffffffffu << (4 * nq)) << (4 * nq));
        r += __popc(bits[q] & keep);
    }
    return r;
}

__global__ __launch_bounds__(1024) void k_prep_lists(Shape s, const int* __restrict__ lens_in, unsigned char* mask, long long msb,
                                                    long long mst, int guard, unsigned char* __restrict__ need,
                                                    int* __restrict__ lens,
                                                    unsigned* __restrict__ sync, int sync_words, int* __restrict__ rows,
                                                    int* __restrict__ nrows, int* __restrict__ tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_lds[];
    __shared__ int red[6][16];
    __shared__ int wsum[4][16];
    __shared__ unsigned char tflag[PREPL_MAX_M / 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, M = s.M, B = s.B, T = s.T;
    const int Mp = (M + 15) & ~15;
    unsigned char* nl = pl_lds;
    unsigned char* allm = pl_lds + Mp + 16;
    const bool first = blockIdx.x == 0;
    PREP_STAMP(0);
    if (first)
        for (int i = tid; i < sync_words; i += 1024) sync[i] = 0u;
    for (int b = tid; b < B; b += 1024) allm[b] = guard && T > 1;
    if (tid < 32) nl[Mp - 16 + tid] = 0;                      // rows past M count as not needed
    __syncthreads();
    // ---- need bytes in (t, b) order, without the guard; rows that have an unmasked step lose their guard flag.
    // A thread takes 16 steps of one sentence, consecutive lanes take consecutive sentences: the byte stores of a
    // wave then fall into 64 consecutive bytes (sentence-major chunks put them 8 deep on 8 banks: 4.7 us vs 1.x).
    PREP_STAMP(1);
    const int nch = (T + 15) >> 4;
    const bool vec = mst == 1 && (msb & 15) == 0 && ((unsigned long long)mask & 15) == 0;
    for (int u = tid; u < nch * B; u += 1024) {
        const int tc = u / B, b = u - tc * B, t0 = tc * 16;
        const unsigned char* mrow = mask + (long long)b * msb;
        u32x4 w;
        if (vec && t0 + 16 <= T) w = *(const u32x4*)(mrow + t0);
        else {
            unsigned char by[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) by[i] = t0 + i < T ? mrow[(long long)(t0 + i) * mst] : (unsigned char)1;
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = by[4 * q] | (by[4 * q + 1] << 8) | (by[4 * q + 2] << 16) | ((unsigned)by[4 * q + 3] << 24);
        }
        int len = lens_in[b];
        len = len < 0 ? 0 : (len > T ? T : len);
        int z = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned byte = (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
            const bool valid = t0 + i < T;
            nl[valid ? (t0 + i) * B + b : Mp] = (unsigned char)((t0 + i < len ? 1 : 0) | (byte != 0u ? 0 : 2));
            z |= byte == 0u && valid;
        }
        if (z) allm[b] = 0;
    }
    __syncthreads();
    // ---- the guard: an all-masked row gets t = 1 unmasked (transfer/models.py:865-867), in LDS and in the caller's mask
    PREP_STAMP(2);
    for (int b = tid; b < B; b += 1024) {
        if (first) {
            const int len = lens_in[b];
            lens[b] = len < 0 ? 0 : (len > T ? T : len);
        }
        if (allm[b]) {
            nl[B + b] |= 2;
            if (first) mask[(long long)b * msb + mst] = 0;
        }
    }
    __syncthreads();
    // ---- counts over all rows (see k_row_lists), flags of the 128-row tiles, own need bytes out
    PREP_STAMP(3);
    const int m0 = blockIdx.x * 1024;
    if (tid < 64 && m0 + tid * 16 < Mp) *(u32x4*)(need + m0 + tid * 16) = *(const u32x4*)(nl + m0 + tid * 16);
    // [0] Cany(m0) [1] C1(m0) [2] C1(min(B, M)) [3] C1(min(M, m0 + B)) [4] Cany(M) [5] C1(M), with
    // C1(x) = #{m < x : bit0}, Cany(x) = #{m < x : need != 0}; whole 16-row chunks below the limit, then the partial one
    const int limB = min(B, M), limN = min(M, m0 + B);
    int cnt[6] = {0, 0, 0, 0, 0, 0};
    for (int c = tid * 16; c < Mp; c += 1024 * 16) {
        const u32x4 w = *(const u32x4*)(nl + c);
        const unsigned long long bal = __ballot((w[0] | w[1] | w[2] | w[3]) != 0u);
        if ((lane & 7) == 0) tflag[c >> 7] = ((bal >> lane) & 0xffull) != 0ull;
        u32x4 b0v;
        int a16 = 0, b16 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            b0v[q] = w[q] & 0x01010101u;
            a16 += __popc((w[q] | (w[q] >> 1)) & 0x01010101u);
            b16 += __popc(b0v[q]);
        }
        cnt[0] += c < m0 ? a16 : 0;
        cnt[1] += c < m0 ? b16 : 0;
        cnt[2] += c + 16 <= limB ? b16 : 0;
        cnt[3] += c + 16 <= limN ? b16 : 0;
        cnt[4] += a16;
        cnt[5] += b16;
        if (c < limB && c + 16 > limB) cnt[2] += popc_low_bytes(b0v, limB - c);
        if (c < limN && c + 16 > limN) cnt[3] += popc_low_bytes(b0v, limN - c);
    }
#pragma unroll
    for (int l = 0; l < 6; ++l) {
        const int v = wave_scan64(cnt[l]);
        if (lane == 63) red[l][wave] = v;
    }
    __syncthreads();
    PREP_STAMP(4);
    // sums over the 16 waves: lists 0..3 sit in the four DPP rows of one read, lists 4..5 in a second
    const int* redf = &red[0][0];
    const int s03 = row_scan16(redf[lane]), s45 = row_scan16(redf[64 + (lane & 31)]);
    const int C0 = __builtin_amdgcn_readlane(s03, 15), C1 = __builtin_amdgcn_readlane(s03, 31);
    const int C2 = __builtin_amdgcn_readlane(s03, 47), C3 = __builtin_amdgcn_readlane(s03, 63);
    const int C4 = __builtin_amdgcn_readlane(s45, 15), C5 = __builtin_amdgcn_readlane(s45, 31);
    if (first && tid == 0) {
        nrows[ROWS_ANY] = C4;
        nrows[ROWS_FWD] = C5 - C2;
        nrows[ROWS_REV] = C5 - C2;                            // rows m with m + B < M and bit0 of need[m + B]
    }
    const int off[3] = {C0, m0 >= B ? C1 - C2 : 0, C3 - C2};
    const int m = m0 + tid;
    const int nd = m < M ? nl[m] : 0;
    const int nn = m + B < M ? nl[m + B] : 0;
    const int nt = (M + 127) >> 7;
    const bool flag[4] = {nd != 0, m >= B && (nd & 1), (nn & 1) != 0, tid < nt && tflag[tid < nt ? tid : 0] != 0};
    int before[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const unsigned long long bal = __ballot(flag[l]);
        before[l] = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[l][wave] = __popcll(bal);
    }
    __syncthreads();
    PREP_STAMP(5);
    // flags in the waves below this one: the four lists again in the four DPP rows
    const int wv = (&wsum[0][0])[lane];
    const int below = row_scan16((lane & 15) < wave ? wv : 0);
    const int pre[4] = {__builtin_amdgcn_readlane(below, 15), __builtin_amdgcn_readlane(below, 31),
                        __builtin_amdgcn_readlane(below, 47), __builtin_amdgcn_readlane(below, 63)};
#pragma unroll
    for (int l = 0; l < 3; ++l)
        if (flag[l]) rows[(long long)l * M + off[l] + pre[l] + before[l]] = m;
    if (tiles != nullptr) {
        if (flag[3] && (tid >> 3) == (int)blockIdx.x) tiles[1 + pre[3] + before[3]] = tid;
        const int tot = __builtin_amdgcn_readlane(row_scan16(wv), 63);
        if (first && tid == 0) tiles[0] = tot;
    }
    PREP_STAMP(6);
}

// Packed (permuted) gate column -> row of a torch [4H, *] LSTM parameter, or -1 for padding.
__device__ __forceinline__ int gate_row(int pc, int H) {
    const int unit = (pc >> 6) * 16 + (pc & 15), gate = (pc & 63) >> 4;
    return unit < H ? gate * H + unit : -1;
}

// ------------------------------------------------------------------------------------------------
// d rnn.weight_ih_l0{,_reverse} [4H,E] and d linear_emb.weight [H,E]:  dWcat = DZ^T . X
// (autograd of nn.LSTM's input GEMMs, transfer/models.py:906, and of linear_emb, :925)
// grid: (Ncat/BN1, Ep/BN2, row splits)
// The bias gradients ride along when E < Ep (CatBias non-NULL): they are the column sums of DZ, i.e. dWcat's
// column for an input that is 1 in every row.  X's padding columns E..Ep-1 are zero, so the B tile that holds
// column E gets a bf16 1.0 there on its way into LDS and the epilogue sends output column E to the biases:
// k_colsum_dz's 63 us (cfg3) for no extra MFMA.
// ------------------------------------------------------------------------------------------------
struct CatBias {
    float *bih0, *bhh0, *bih1, *bhh1, *blin;     // d rnn.bias_ih / bias_hh of both directions, d linear_emb.bias
};
template <class W>
__global__ __launch_bounds__(W::NT) void k_wgrad_cat(Shape s, const bf16raw* __restrict__ DZ,
                                                    const bf16raw* __restrict__ X,
                                                    const int* __restrict__ rows, const int* __restrict__ nrows,
                                                    float* __restrict__ gih0, float* __restrict__ gih1,
                                                    float* __restrict__ glin, CatBias cb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_lds[];   // 2 * 64 * (P1 + P2) bytes
    const int n10 = blockIdx.x * W::BN1, n20 = blockIdx.y * W::BN2;
    int j_begin, j_end;
    if (!wgrad_slice(nrows[ROWS_ANY], blockIdx.z, gridDim.z, &j_begin, &j_end)) return;
    const int Ncat = s.Ncat, Ep = s.Ep;
    // skip column blocks whose destination is absent (frozen tensors)
    {
        const int lo = n10, hi = min(Ncat, n10 + W::BN1) - 1;
        const bool w0 = gih0 != nullptr && lo < s.Gp, w1 = gih1 != nullptr && hi >= s.Gp && lo < 2 * s.Gp;
        const bool w2 = glin != nullptr && hi >= 2 * s.Gp;
        const bool b0 = (cb.bih0 != nullptr || cb.bhh0 != nullptr) && lo < s.Gp;
        const bool b1 = (cb.bih1 != nullptr || cb.bhh1 != nullptr) && hi >= s.Gp && lo < 2 * s.Gp;
        const bool b2 = cb.blin != nullptr && hi >= 2 * s.Gp;
        const bool has_e = s.E < Ep && n20 <= s.E && s.E < n20 + W::BN2;      // this block's X tile holds the ones column
        if (!(w0 || w1 || w2) && !((b0 || b1 || b2) && has_e)) return;
    }
    const bool ones = cb.bih0 != nullptr || cb.bhh0 != nullptr || cb.bih1 != nullptr || cb.bhh1 != nullptr || cb.blin != nullptr;
    const int ecol = ones && s.E < Ep ? s.E : -1;
    f32x4 acc[W::T1][W::T2];
    wgrad_zero<W>(acc);
    wgrad_core<W>(
        j_begin, j_end, [&](int j) { return rows[j]; },
        [&](int m, int c, bool ok) {
            const int n = n10 + c;
            return load16_if(DZ + (long long)m * Ncat + n, DZ, ok && n < Ncat);
        },
        [&](int m, int c, bool ok) {
            const int n = n20 + c;
            return load16_if(X + (long long)m * Ep + n, X, ok && n < Ep);
        },
        [&](u32x4 v, int c) {
            // the ones column (rows past the end of the list carry zeros in DZ, so they may get the 1 too)
            const int d = ecol - (n20 + c);
            const unsigned one = d >= 0 && d < 8 ? 0x3f80u << (16 * (d & 1)) : 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] |= (d >> 1) == q ? one : 0u;
            return v;
        },
        acc, wg_lds);
    const int E = s.E, H = s.H, Gp = s.Gp;
    wgrad_epilogue<W>(acc, [&](int r1, int r2, float v) {
        const int n1 = n10 + r1, n2 = n20 + r2;
        if (n1 >= Ncat) return;
        if (n2 == ecol) {                                        // column sums of DZ -> biases
            if (n1 < 2 * Gp) {
                const int dir = n1 >= Gp ? 1 : 0;
                const int row = gate_row(n1 - dir * Gp, H);
                float* bi = dir ? cb.bih1 : cb.bih0;
                float* bh = dir ? cb.bhh1 : cb.bhh0;
                if (row >= 0 && bi != nullptr) unsafeAtomicAdd(bi + row, v);
                if (row >= 0 && bh != nullptr) unsafeAtomicAdd(bh + row, v);
            } else if (n1 - 2 * Gp < H && cb.blin != nullptr) {
                unsafeAtomicAdd(cb.blin + (n1 - 2 * Gp), v);
            }
            return;
        }
        if (n2 >= E) return;
        float* dst = nullptr;
        if (n1 < 2 * Gp) {
            const int dir = n1 >= Gp ? 1 : 0;
            const int row = gate_row(n1 - dir * Gp, H);
            float* g = dir ? gih1 : gih0;
            if (row >= 0 && g != nullptr) dst = g + (long long)row * E + n2;
        } else {
            const int j = n1 - 2 * Gp;
            if (j < H && glin != nullptr) dst = glin + (long long)j * E + n2;
        }
        if (dst != nullptr) unsafeAtomicAdd(dst, v);
    });
}

// ------------------------------------------------------------------------------------------------
// d rnn.weight_hh_l0{,_reverse} [4H,H]:  dWhh[d] = sum_t dZ_t[d]^T . h_{prev(t)}[d]
// h_prev of token row m = (t,b): forward direction Hs[t-1][b] (zero state at t = 0), reverse
// direction Hs[t+1][b] (zero state at t = len-1).  grid: (Gp/BN1, Hp/BN2, 2 * row splits)
// ------------------------------------------------------------------------------------------------
template <class W>
__global__ __launch_bounds__(W::NT) void k_wgrad_hh(Shape s, const bf16raw* __restrict__ DZ,
                                                   const bf16raw* __restrict__ Hs,
                                                   const int* __restrict__ rows, const int* __restrict__ nrows,
                                                   float* __restrict__ ghh0, float* __restrict__ ghh1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_lds[];   // 2 * 64 * (P1 + P2) bytes
    const int dir = blockIdx.z & 1;
    float* g = dir ? ghh1 : ghh0;
    if (g == nullptr) return;
    const int n10 = blockIdx.x * W::BN1, n20 = blockIdx.y * W::BN2;
    int j_begin, j_end;
    if (!wgrad_slice(nrows[ROWS_FWD + dir], blockIdx.z >> 1, gridDim.z >> 1, &j_begin, &j_end)) return;
    const int* list = rows + (long long)(ROWS_FWD + dir) * s.M;
    const int Ncat = s.Ncat, Hp = s.Hp, Gp = s.Gp, B = s.B;
    const long long shift = dir ? B : -B;         // token-row offset of the previous recurrence step
    f32x4 acc[W::T1][W::T2];
    wgrad_zero<W>(acc);
    wgrad_core<W>(
        j_begin, j_end, [&](int j) { return list[j]; },
        [&](int m, int c, bool ok) {
            const int n = n10 + c;
            return load16_if(DZ + (long long)m * Ncat + dir * Gp + n, DZ, ok && n < Gp);
        },
        [&](int m, int c, bool ok) {
            const int n = n20 + c;
            return load16_if(Hs + ((m + shift) * 2 + dir) * Hp + n, Hs, ok && n < Hp);
        },
        acc, wg_lds);
    const int H = s.H;
    wgrad_epilogue<W>(acc, [&](int r1, int r2, float v) {
        const int n1 = n10 + r1, n2 = n20 + r2;
        if (n1 >= Gp || n2 >= H) return;
        const int row = gate_row(n1, H);
        if (row >= 0) unsafeAtomicAdd(g + (long long)row * H + n2, v);
    });
}

// ------------------------------------------------------------------------------------------------
// d embedding.weight [V,E], soft tokens:  dW_emb = S^T . dX   (autograd of `src @ embedding.weight`,
// transfer/models.py:898; every vocabulary row including the pad row receives gradient).
// grid: (Vp/BN1, Ep/BN2, row splits)
// ------------------------------------------------------------------------------------------------
template <class W, int KIND>
__global__ __launch_bounds__(W::NT) void k_wgrad_emb(Shape s, const void* __restrict__ src, long long sT, long long sB,
                                                    const bf16raw* __restrict__ DX,
                                                    const int* __restrict__ rows, const int* __restrict__ nrows,
                                                    float* __restrict__ gemb, int xcd_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_lds[];   // 2 * 64 * (P1 + P2) bytes
    typedef SoftRow<KIND> SR;
    typedef typename SR::Elem Elem;
    // xcd_tiles != 0 (grid x = 8 * xcd_tiles): workgroups b, b+8, ... share an XCD and take NEIGHBOURING vocabulary
    // tiles.  A tile's 128-byte row segments of S are misaligned (V = 9353), so every line it fetches is half used
    // and the other half belongs to the neighbouring tile: on one XCD, at the same time, that half comes out of L2
    // (FETCH_SIZE 2.3 x the needed rows of S with consecutive tiles spread over the eight XCDs).
    const int tile1 = xcd_tiles ? (int)(blockIdx.x & 7) * xcd_tiles + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (xcd_tiles && (int)(blockIdx.x >> 3) >= xcd_tiles) return;
    const int n10 = tile1 * W::BN1, n20 = blockIdx.y * W::BN2;
    if (n10 >= s.V) return;
    int j_begin, j_end;
    if (!wgrad_slice(nrows[ROWS_ANY], blockIdx.z, gridDim.z, &j_begin, &j_end)) return;
    const int V = s.V, Ep = s.Ep, B = s.B;
    f32x4 acc[W::T1][W::T2];
    wgrad_zero<W>(acc);
    auto row_of = [&](int j) { return rows[j]; };
    auto load_b = [&](int m, int c, bool ok) {
        const int n = n20 + c;
        return load16_if(DX + (long long)m * Ep + n, DX, ok && n < Ep);
    };
    auto s_row = [&](int m, bool ok) -> const Elem* {
        const int t = m / B, b = m - t * B;
        return ok ? (const Elem*)src + (long long)t * sT + (long long)b * sB : nullptr;
    };
    if (n10 + W::BN1 <= V) {     // interior vocabulary tile: whole chunks, branch-free loads
        wgrad_core<W>(
            j_begin, j_end, row_of,
            [&](int m, int c, bool ok) {
                return typename SR::Cvt()(SR::load_fast(s_row(m, ok), (const Elem*)src, n10 + c));
            },
            load_b, acc, wg_lds);
    } else {                     // the tile holding the ragged end of V: element-guarded loads
        wgrad_core<W>(
            j_begin, j_end, row_of,
            [&](int m, int c, bool ok) { return typename SR::Cvt()(SR::load(s_row(m, ok), n10 + c, V)); }, load_b, acc,
            wg_lds);
    }
    const int E = s.E;
    wgrad_epilogue<W>(acc, [&](int r1, int r2, float v) {
        const int n1 = n10 + r1, n2 = n20 + r2;
        if (n1 < V && n2 < E) unsafeAtomicAdd(gemb + (long long)n1 * E + n2, v);
    });
}

// d embedding.weight, hard tokens: scatter-add of dX rows (autograd of nn.Embedding,
// transfer/models.py:900; the padding_idx row gets no gradient, :770-775).
__global__ void k_wgrad_emb_ids(Shape s, const long long* __restrict__ ids, long long ib, long long it,
                                const bf16raw* __restrict__ DX, const unsigned char* __restrict__ need, int pad_id,
                                float* __restrict__ gemb) {
    const int cpr = s.Ep / 4;
    const long long total = (long long)s.M * cpr;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(c / cpr), e = (int)(c - (long long)m * cpr) * 4;
        if (!need[m] || e >= s.E) continue;
        const int t = m / s.B, b = m - t * s.B;
        const long long id = ids[(long long)b * ib + (long long)t * it];
        if (id < 0 || id >= s.V || id == pad_id) continue;
        const u32x2 raw = *(const u32x2*)(DX + (long long)m * s.Ep + e);
        const f32x4 v = f32x4{bf16_to_f32((bf16raw)(raw[0] & 0xffffu)), bf16_to_f32((bf16raw)(raw[0] >> 16)),
                              bf16_to_f32((bf16raw)(raw[1] & 0xffffu)), bf16_to_f32((bf16raw)(raw[1] >> 16))};
        atomic_add4(gemb + id * s.E + e, e, s.E, v);
    }
}

// ------------------------------------------------------------------------------------------------
// Per-sentence dense layers: d hidden_bridges.0.weight [H,2H] = dq^T . [h_fwd | h_rev]
// (transfer/models.py:929-934) and d linear_q.weight [H,H] = du^T . q (:855).  K = B sentences.
// BKIND 0: B operand fp32 [rows][ldb];  1: the [h_fwd(len-1) | h_rev(0)] gather out of Hs.
// grid: (Hp/BN1, N2/BN2, row splits)
// ------------------------------------------------------------------------------------------------
template <class W, int BKIND>
__global__ __launch_bounds__(W::NT) void k_wgrad_dense(Shape s, const float* __restrict__ A, const void* __restrict__ Bop,
                                                      int N2p, const int* __restrict__ lens, float* __restrict__ g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_lds[];   // 2 * 64 * (P1 + P2) bytes
    const int n10 = blockIdx.x * W::BN1, n20 = blockIdx.y * W::BN2;
    int j_begin, j_end;
    if (!wgrad_slice(s.B, blockIdx.z, gridDim.z, &j_begin, &j_end)) return;
    const int Hp = s.Hp, B = s.B;
    f32x4 acc[W::T1][W::T2];
    wgrad_zero<W>(acc);
    wgrad_core<W>(
        j_begin, j_end, [](int j) { return j; },
        [&](int b, int c, bool ok) {
            const int n = n10 + c;
            return load8f_if(A + (long long)b * Hp + n, A, ok && n < Hp);
        },
        [&](int b, int c, bool ok) {
            const int n = n20 + c;
            const bool ok2 = ok && n < N2p;
            if constexpr (BKIND == 0) {
                return load8f_if((const float*)Bop + (long long)b * N2p + n, (const float*)Bop, ok2);
            } else {
                const int d = n >= Hp ? 1 : 0;
                const int len = lens[ok2 ? b : 0];
                const int ts = d ? 0 : len - 1;
                return load16_if((const bf16raw*)Bop + (((long long)ts * B + b) * 2 + d) * Hp + (n - d * Hp), Bop,
                                 ok2 && len > 0);
            }
        },
        acc, wg_lds);
    const int H = s.H;
    wgrad_epilogue<W>(acc, [&](int r1, int r2, float v) {
        const int n1 = n10 + r1, n2 = n20 + r2;
        if (n1 >= H || n2 >= N2p) return;
        if constexpr (BKIND == 0) {
            if (n2 < H) unsafeAtomicAdd(g + (long long)n1 * H + n2, v);
        } else {
            const int d = n2 >= Hp ? 1 : 0, j = n2 - d * Hp;
            if (j < H) unsafeAtomicAdd(g + (long long)n1 * 2 * H + d * H + j, v);
        }
    });
}

// ------------------------------------------------------------------------------------------------
// Bias gradients = column sums.
// k_colsum_dz: DZ [M][Ncat] bf16 -> d rnn.bias_ih/hh (both get the same sum, nn.LSTM adds the two
// biases) and d linear_emb.bias.  grid: (Ncat/256, row splits), 128 threads x 2 columns.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_colsum_dz(Shape s, const bf16raw* __restrict__ DZ,
                                                  const int* __restrict__ rows, const int* __restrict__ nrows,
                                                  float* __restrict__ bih0, float* __restrict__ bhh0,
                                                  float* __restrict__ bih1, float* __restrict__ bhh1,
                                                  float* __restrict__ blin) {
    // 8 columns per lane (16-byte loads), only the contributing rows (compacted list), 8 rows in flight;
    // few row splits: every split costs an atomic add per column onto the same few addresses
    const int n = blockIdx.x * 1024 + threadIdx.x * 8;
    if (n >= s.Ncat) return;
    int j_begin, j_end;
    if (!wgrad_slice(nrows[ROWS_ANY], blockIdx.y, gridDim.y, &j_begin, &j_end)) return;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto add = [&](u32x4 v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[2 * e] += bf16_to_f32((bf16raw)(v[e] & 0xffffu));
            a[2 * e + 1] += bf16_to_f32((bf16raw)(v[e] >> 16));
        }
    };
    int j = j_begin;
    for (; j + 7 < j_end; j += 8) {
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *(const u32x4*)(DZ + (long long)rows[j + i] * s.Ncat + n);
#pragma unroll
        for (int i = 0; i < 8; ++i) add(v[i]);
    }
    for (; j < j_end; ++j) add(*(const u32x4*)(DZ + (long long)rows[j] * s.Ncat + n));
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int col = n + e;
        if (col < 2 * s.Gp) {
            const int dir = col >= s.Gp ? 1 : 0;
            const int row = gate_row(col - dir * s.Gp, s.H);
            if (row < 0) continue;
            float* bi = dir ? bih1 : bih0;
            float* bh = dir ? bhh1 : bhh0;
            if (bi != nullptr) unsafeAtomicAdd(bi + row, a[e]);
            if (bh != nullptr) unsafeAtomicAdd(bh + row, a[e]);
        } else {
            const int j2 = col - 2 * s.Gp;
            if (j2 < s.H && blin != nullptr) unsafeAtomicAdd(blin + j2, a[e]);
        }
    }
}

// The small reductions over sentences of sad_param_grads in ONE launch (six dependent launches of ~4.6 us each
// before): jobs 0..n-1: dst[c] += sum_b src[b][c], c < cols; the last grid slice (when gwo != NULL):
// d output_layer.weight [L,H] += dl^T . ctx (transfer/models.py:949).  grid: (ceil(max cols / 256), row splits, n + 1)
struct ColsumJobs {
    const float* src[5];
    float* dst[5];
    int cols[5], ld[5];
    int n;
};
__global__ __launch_bounds__(256) void k_colsum_small(Shape s, ColsumJobs jobs, int rchunk, const float* __restrict__ DL,
                                                     const float* __restrict__ CTX, float* __restrict__ gwo) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * rchunk, r1 = min(s.B, r0 + rchunk);
    const int z = blockIdx.z;
    if (z < jobs.n) {
        const float* src = jobs.src[z];
        float* dst = jobs.dst[z];
        const int cols = jobs.cols[z], ld = jobs.ld[z];
        if (dst == nullptr || c >= cols) return;
        float a = 0.f;
#pragma unroll 8
        for (int r = r0; r < r1; ++r) a += src[(long long)r * ld + c];
        unsafeAtomicAdd(dst + c, a);
    } else {
        if (gwo == nullptr || c >= s.H) return;
        for (int l = 0; l < s.L; ++l) {
            float a = 0.f;
#pragma unroll 8
            for (int b = r0; b < r1; ++b) a += DL[(long long)b * s.L + l] * CTX[(long long)b * s.Hp + c];
            unsafeAtomicAdd(gwo + (long long)l * s.H + c, a);
        }
    }
}

}  // namespace sad
