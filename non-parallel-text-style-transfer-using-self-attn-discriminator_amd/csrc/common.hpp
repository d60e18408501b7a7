// Common device/host definitions for libsad (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sad {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));     // dword-aligned 16-B access
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4h __attribute__((ext_vector_type(4), aligned(2)));  // 2-byte-aligned 16-B access
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2h __attribute__((ext_vector_type(2), aligned(2)));  // 2-byte-aligned 8-B access
typedef unsigned short bf16raw;                                           // bf16 bit pattern in memory

__host__ __device__ constexpr int ceil_to(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ constexpr int ceil_div(int x, int m) { return (x + m - 1) / m; }

// Derived (padded) sizes.  Time-major token row m = t*B + b.
struct Shape {
    int B, T, V, E, H, L;
    int Ep;    // E rounded up to 32   (K of the in-projection, N of the soft-embedding GEMM)
    int Hp;    // H rounded up to 32   (K of the recurrent GEMM; unit dimension)
    int Gp;    // 4*Hp                 (permuted gate columns per direction)
    int Vp;    // V rounded up to 64   (K of the soft-embedding GEMM, rows of the packed embedding)
    int Ncat;  // 2*Gp + Hp            (fused in-projection columns: fwd gates | rev gates | attention keys)
    int M;     // T*B token rows
    int cell;  // 0 = LSTM, 1 = GRU (four-slot layout [r, z, n_input, n_hidden], include/sad.h)
    int Gld;   // row pitch of G in floats: 2*Gp + 64 (breaks the power-of-two 16 KB pitch: 16-row store
               // instructions at a 2^k pitch camp on one HBM channel)
};

constexpr int CELL_LSTM = 0, CELL_GRU = 1;   // == SAD_CELL_* of include/sad.h

__host__ __device__ inline Shape make_shape(int B, int T, int V, int E, int H, int L, int cell = 0) {
    Shape s;
    s.B = B; s.T = T; s.V = V; s.E = E; s.H = H; s.L = L;
    s.Ep = ceil_to(E, 32);
    s.Hp = ceil_to(H, 32);
    s.Gp = 4 * s.Hp;
    s.Vp = ceil_to(V, 64);
    s.Ncat = 2 * s.Gp + s.Hp;
    s.M = T * B;
    s.Gld = 2 * s.Gp + 64;
    s.cell = cell;
    return s;
}

// Gate-column permutation: packed column pc = ub*64 + gate*16 + u  <->  torch row gate*H + ub*16 + u
// (torch gate order i,f,g,o).  One 64-column block holds all four gates of 16 hidden units, so the
// 16x16 MFMA tiles tn = 0..3 of a wave are the four gates of the SAME units.
__host__ __device__ inline int gate_col(int unit, int gate) { return (unit >> 4) * 64 + gate * 16 + (unit & 15); }

#ifdef __HIPCC__
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    bf16x2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ u32x2 pack_bf16x4(f32x4 v) {
    u32x2 r;
    r[0] = pack_bf16x2(v[0], v[1]);
    r[1] = pack_bf16x2(v[2], v[3]);
    return r;
}
__device__ __forceinline__ u32x4 pack_bf16x8(f32x4 a, f32x4 b) {
    u32x4 r;
    r[0] = pack_bf16x2(a[0], a[1]);
    r[1] = pack_bf16x2(a[2], a[3]);
    r[2] = pack_bf16x2(b[0], b[1]);
    r[3] = pack_bf16x2(b[2], b[3]);
    return r;
}
__device__ __forceinline__ float bf16_to_f32(bf16raw x) { return __builtin_bit_cast(float, (unsigned)x << 16); }
__device__ __forceinline__ bf16raw f32_to_bf16(float f) { return __builtin_bit_cast(bf16raw, (__bf16)f); }

// v_exp_f32 + v_rcp_f32 (1 ulp each): an IEEE division expands to ~10 instructions, and the cell update of
// the recurrence evaluates 5 of these per hidden unit on the critical path of every time step.
// exp2 overflow -> inf -> rcp -> 0 and underflow -> 0 give the right limits.
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// Wave reductions on DPP row shifts + row broadcasts, result in every lane (v_readlane of lane 63).  A
// __shfl_xor butterfly is six dependent ds_bpermute round trips through the LDS pipe (~2 us for a block
// reduction of six counters in k_prep_lists); the small kernels that use these are chains of such latencies.
// All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or(float v, float identity) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_or<0x111, 0xf>(v, 0.f);          // row_shr:1
    v += dpp_or<0x112, 0xf>(v, 0.f);          // row_shr:2
    v += dpp_or<0x114, 0xf>(v, 0.f);          // row_shr:4
    v += dpp_or<0x118, 0xf>(v, 0.f);          // row_shr:8   -> lane 15 of each row: the row's sum
    v += dpp_or<0x142, 0xa>(v, 0.f);          // row_bcast:15 into rows 1 and 3
    v += dpp_or<0x143, 0xc>(v, 0.f);          // row_bcast:31 into rows 2 and 3 -> lane 63: the wave's sum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_or<0x111, 0xf>(v, -INFINITY));
    v = fmaxf(v, dpp_or<0x112, 0xf>(v, -INFINITY));
    v = fmaxf(v, dpp_or<0x114, 0xf>(v, -INFINITY));
    v = fmaxf(v, dpp_or<0x118, 0xf>(v, -INFINITY));
    v = fmaxf(v, dpp_or<0x142, 0xa>(v, -INFINITY));
    v = fmaxf(v, dpp_or<0x143, 0xc>(v, -INFINITY));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Activated gates of the persistent recurrences: fp16 records of [unit][i, f, g, o] (SAD_WS_GA).  The
// values lie in [-1, 1] (the GRU's fourth slot, W_hn h + b_hn, is O(1)), so fp16 keeps 11 significant bits (the BPTT rounds its results to bf16
// anyway); a lane's 4 units x 4 gates are 32 contiguous bytes.  Halves the gate stores of the forward
// recurrence (47 us of its 305 at cfg3 were these stores) and the gate loads of the BPTT.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// Record (4 units x 4 gates, 32 bytes) of sentence b, unit quad j/4 at step t: [t][dir][16-sentence
// block][quad][16 sentences][16 halves] -- the lane mapping of both recurrences is lane & 15 = sentence,
// lane >> 4 = quad, so a wave instruction covers 2 KB of contiguous bytes (row-major records gave 16
// cache lines per quarter wave, the fragment-shaped access the hand-off images avoid).
__host__ __device__ inline size_t ga_bytes(int T, int B, int Hp) { return (size_t)T * 2 * ((B + 15) / 16 * 16) * Hp * 4 * 2; }
__device__ __forceinline__ long long ga_index(int t, int dir, int b, int j, int B, int Hp) {
    const int sblocks = (B + 15) >> 4;
    return (((((long long)t * 2 + dir) * sblocks + (b >> 4)) * (Hp >> 2) + (j >> 2)) * 16 + (b & 15)) * 16;
}
__device__ __forceinline__ void ga_store(unsigned short* p, const f32x4& gi, const f32x4& gf, const f32x4& gg, const f32x4& go) {
    f16x8 lo, hi;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        lo[4 * e + 0] = (_Float16)gi[e]; lo[4 * e + 1] = (_Float16)gf[e]; lo[4 * e + 2] = (_Float16)gg[e]; lo[4 * e + 3] = (_Float16)go[e];
        hi[4 * e + 0] = (_Float16)gi[2 + e]; hi[4 * e + 1] = (_Float16)gf[2 + e]; hi[4 * e + 2] = (_Float16)gg[2 + e]; hi[4 * e + 3] = (_Float16)go[2 + e];
    }
    *(f16x8*)p = lo;
    *(f16x8*)(p + 8) = hi;
}
__device__ __forceinline__ void ga_unpack(const u32x4& rlo, const u32x4& rhi, f32x4& gi, f32x4& gf, f32x4& gg, f32x4& go) {
    const f16x8 lo = __builtin_bit_cast(f16x8, rlo), hi = __builtin_bit_cast(f16x8, rhi);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        gi[e] = (float)lo[4 * e + 0]; gf[e] = (float)lo[4 * e + 1]; gg[e] = (float)lo[4 * e + 2]; go[e] = (float)lo[4 * e + 3];
        gi[2 + e] = (float)hi[4 * e + 0]; gf[2 + e] = (float)hi[4 * e + 1]; gg[2 + e] = (float)hi[4 * e + 2]; go[2 + e] = (float)hi[4 * e + 3];
    }
}
#endif

}  // namespace sad
