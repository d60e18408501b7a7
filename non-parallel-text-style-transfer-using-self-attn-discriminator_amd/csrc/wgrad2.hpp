// Parameter gradients, second generation (round 3): the three large TN GEMMs of the unfrozen mode
//   d rnn.weight_ih / d linear_emb.weight (dWcat = DZ^T . X), d rnn.weight_hh (DZ^T . h_prev), d embedding.weight (S^T . dX)
// (autograd of transfer/models.py:898,906,925; what loss.backward() leaves in .grad, classifier/trainers.py:371-373).
//
// Why a second core (wgrad.hpp's wgrad_core stays for the small / odd shapes): at cfg3 the first generation ran
// 64-column tiles of the long operand (48 KB of operands per 64-row stage for 2.6 MFLOP: a CU takes in ~60 GB/s, so
// 0.8 us per stage against 0.27 us of MFMA), staged every tile through two register slots (228 VGPRs for 40
// accumulators; 128-column tiles spilled) and took 354 / 197 / 191 us (0.11-0.15 of the MFMA peak).  Here
//   * tiles are 128 x 320 (x 256 for W_hh): 56 KB per stage for 5.2 MFLOP;
//   * three wave roles in a 1024-thread workgroup (wgrad2_core below): the short operand (X, dX, Hs: L2-resident,
//     re-read by every column tile) goes global -> LDS by LDS-direct loads (global_load_lds_dwordx4: no registers, no
//     ds_write) issued by 4 loader waves, three LDS stages; the long operand (DZ, S: streamed from HBM once) passes
//     through FOUR static register sets of 4 other loader waves (a wave that also holds the 80 accumulators of its
//     share of the tile has room for one set only, and one tile in flight ran at HBM latency: 2-3 us per stage);
//     8 compute waves multiply;
//   * the LDS tiles are UNPADDED row-major images (an LDS-direct instruction writes 64 consecutive 16-byte chunks)
//     with the 32-byte column pairs XOR-swizzled per row, so that the transposing fragment reads
//     (ds_read_b64_tr_b16, see wgrad.hpp) stay conflict-free: a half wave reads 8 rows x 32 bytes, which must fall
//     into 8 different 8-bank groups; with pitch P the row term is (P/32 * r) mod 8 -- 0 for P = 256 / 512,
//     4r mod 8 for P = 640 -- and the swizzle supplies the missing bits (r & 7, resp. (r >> 1) & 3);
//   * every workgroup stores its partial tile to a slab and a second launch reduces the row splits of a tile in fixed
//     order (no fp32 atomics: 8-10 M atomic adds per launch took ~150 us whatever stood in front of them); atomics
//     remain the fallback when the partial tiles of a launch do not fit the slab (SAD_DBG_WGRAD_ATOMICS tests it).
// What hipcc's waitcnt pass does to such loops, and how each case is avoided here: DESIGN.md, "Parameter gradients".
#pragma once
#include "wgrad.hpp"

namespace sad {

// SW: 0 -> pair swizzle r & 7 (pitch/32 = 0 mod 8), 1 -> (r >> 1) & 3 (pitch/32 = 4 mod 8)
template <int BN> struct W2Swz { static constexpr int kind = ((BN * 2 / 32) % 8 == 0) ? 0 : 1; };
template <int KIND> __device__ __forceinline__ int w2_swz(int r) { return KIND == 0 ? (r & 7) : ((r >> 1) & 3); }

// LDS writes of a wave that has LDS-direct loads in flight go through inline asm: an ordinary LDS access while
// LDS-direct loads are pending makes hipcc's waitcnt pass (which cannot tell the LDS regions apart) wait for ALL of
// them -- s_waitcnt vmcnt(0).  Inline asm is invisible to that pass; the lgkmcnt wait is the caller's.
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(lds_u8*)p; }
__device__ __forceinline__ void lds_store16b(void* p, unsigned v) {
    asm volatile("ds_write_b16 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory");
}

template <int BN1_, int BN2_, int W1_, int W2_>
struct W2Cfg {
    static constexpr int BN1 = BN1_, BN2 = BN2_, W1 = W1_, W2 = W2_, BK = 64;
    static constexpr int NT = 1024;                                  // 8 compute waves, 4 A-loader waves, 4 B-loader waves
    static constexpr int T1 = BN1_ / W1_ / 16, T2 = BN2_ / W2_ / 16;
    static constexpr int P1 = BN1_ * 2, P2 = BN2_ * 2;               // LDS row pitch (unpadded), bytes
    static constexpr int CPR1 = BN1_ / 8, CPR2 = BN2_ / 8;           // 16-byte chunks per tile row
    static constexpr int ABYTES = BK * P1, BBYTES = BK * P2;
    static constexpr int BOFF = 2 * ABYTES;                          // LDS: A0 A1 B0 B1 B2
    static constexpr int LDS_BYTES = 2 * ABYTES + 3 * BBYTES;
    static constexpr int NIB = BBYTES / 1024 / 4;                    // LDS-direct instructions per B-loader wave and tile
    static constexpr int ACH = BK * CPR1 / 256;                      // A chunks per loader thread and tile
    static constexpr int SW1 = W2Swz<BN1_>::kind, SW2 = W2Swz<BN2_>::kind;
    static_assert(W1_ * W2_ == 8, "eight compute waves");
    static_assert(BN1_ == 128, "the A loaders map 16 lanes to a tile row");
    static_assert(BBYTES % 4096 == 0, "LDS-direct instructions must split evenly over the four B-loader waves");
    static_assert(BN1_ % (W1_ * 16) == 0 && BN2_ % (W2_ * 16) == 0, "tile/wave mismatch");
    static_assert((BN1_ / 16) % (SW1 == 0 ? 8 : 4) == 0 && (BN2_ / 16) % (SW2 == 0 ? 8 : 4) == 0, "swizzle groups");
};

// Three wave roles, as in K1 (k_softemb_fwd_ws3): the accumulators of a 128 x 320 tile leave a wave that also loads
// no room for more than one tile of the long operand in flight (two or three register sets spill), and with one tile
// in flight every k-tile ran at HBM latency (2-3 us per 64-row stage, 285 / 127 / 97 us per launch).  Here
//   waves 0-7    multiply: transposing fragment reads + MFMAs, no vector-memory instruction until the results;
//   waves 8-11   A loaders (the long operand: DZ resp. the soft tokens S, streamed from HBM once): register-staged
//                (dtype conversion, 2-/4-byte aligned rows), D tiles of loads in flight in D static register sets
//                (loop unrolled D times: hipcc's own counted vmcnt waits), two LDS stages;
//   waves 12-15  B loaders (X, dX, Hs: a few 10 MB, L2-resident, re-read by every column tile): LDS-direct loads, three
//                LDS stages, two tiles in flight, own vmcnt counting, raw barriers.
// One workgroup barrier per tile, 1 + nk barriers in every role.  Token rows come from the compacted row list: a
// loader lane fetches ONE list entry per tile (a tile ahead in the B loaders, D tiles ahead in the A loaders) and the
// lanes of a wave exchange them with ds_bpermute (no LDS memory, no memory operand: the waitcnt pass leaves the
// LDS-direct loads alone).  Entries past the slice are clamped to its last row: B rows stay finite, A rows are zeroed
// on their way to LDS.
//
// a_load(row, col) -> RAW              8 entries of columns [col, col + 8) of the block tile; BRANCH-FREE
// cvt(RAW) -> u32x4                    the 8 bf16 that go to LDS
// b_src(row, col) -> const void*       global address of that 16-byte chunk of B; 16-byte aligned
// patch(lds_chunk_ptr, col)            called for every B chunk a lane has loaded, once it has landed (ones column)
// acc[t1][t2][e]: as in wgrad.hpp (plain operand order: 16 lanes hold 16 consecutive n2 of one row n1)
// Returns true in the compute waves (which hold the results), false in the loader waves (which are done).
template <class W, class RAW, int D, class AL, class CV, class BS, class PB>
__device__ __forceinline__ bool wgrad2_core(const int* __restrict__ list, int j_begin, int j_end, AL a_load, CV cvt, BS b_src,
                                            PB patch, f32x4 (&acc)[W::T1][W::T2], unsigned char* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nk = (j_end - j_begin + W::BK - 1) / W::BK;
    if (wave >= 12) {
        // ------------------------------------- B loader waves ------------------------------------------------------
        const int ww = wave - 12;
        // instruction i of this wave covers chunks 64 (NIB ww + i) .. + 63 of the tile (row-major, CPR2 per row):
        // tile rows 16 ww .. 16 ww + 15
        int rsel[W::NIB], scol[W::NIB];
#pragma unroll
        for (int i = 0; i < W::NIB; ++i) {
            const int g = 64 * (W::NIB * ww + i) + lane, r = g / W::CPR2, p = g - r * W::CPR2;
            rsel[i] = (r - 16 * ww) * 4;
            scol[i] = (p ^ (w2_swz<W::SW2>(r) << 1)) * 8;
        }
        // list entries through inline asm: a compiler-tracked load that is pending across the loop's branches gets
        // s_waitcnt vmcnt(0) in front of its use -- behind the LDS-direct loads of the tile before.  The waits that
        // cover these loads are the counted ones below; they carry the register so that no use moves above them.
        auto ldidx = [&](int t) {
            const int j = j_begin + t * W::BK + 16 * ww + (lane & 15);
            const int* q = list + (j < j_end ? j : j_end - 1);
            int v;
            asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(q) : "memory");
            return v;
        };
        auto issue = [&](int st, int xi) {
            unsigned char* base = lds + W::BOFF + st * W::BBYTES + ww * (W::NIB * 1024);
#pragma unroll
            for (int i = 0; i < W::NIB; ++i) {
                const int row = __builtin_amdgcn_ds_bpermute(rsel[i], xi);
                __builtin_amdgcn_global_load_lds((gmem_cv*)b_src(row, scol[i]), (lds_v*)(base + i * 1024), 16, 0, 0);
            }
        };
        auto landed = [&](int st) {                    // B chunks this lane loaded into stage st have landed
            unsigned char* base = lds + W::BOFF + st * W::BBYTES + ww * (W::NIB * 1024) + lane * 16;
#pragma unroll
            for (int i = 0; i < W::NIB; ++i) patch(base + i * 1024, scol[i]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        int x0 = ldidx(0), x1 = ldidx(1), xc = ldidx(2);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(xc)::"memory");
        issue(0, x0);
        if (nk > 1) {
            issue(1, x1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W::NIB) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        landed(0);
        __builtin_amdgcn_s_barrier();                  // tile 0 ready
        int st1 = 1, st2 = 2;                          // stages of tiles kt + 1, kt + 2
        for (int kt = 0; kt < nk; ++kt) {
            // stage (kt + 2) % 3 was last read in iteration kt - 1: free
            if (kt + 2 < nk) {
                int xn = ldidx(kt + 3);                // BEFORE this tile's LDS-direct loads: vmcnt retires in order
                issue(st2, xc);
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(xn) : "n"(W::NIB) : "memory");   // tile kt + 1 (and xn) has landed
                xc = xn;
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (kt + 1 < nk) landed(st1);
            st1 = st1 == 2 ? 0 : st1 + 1;
            st2 = st2 == 2 ? 0 : st2 + 1;
            __builtin_amdgcn_s_barrier();
        }
        return false;
    }
    if (wave >= 8) {
        // ------------------------------------- A loader waves ------------------------------------------------------
        const int w = wave - 8, q4 = lane >> 4, pc = lane & 15;
        // chunk i of this lane: tile row 16 i + 4 w + q4, column chunk pc; the lane FETCHES the list entry of tile row
        // 16 ((lane >> 2) & 3) + 4 w + (lane & 3), so chunk i's entry sits in lane 4 i + q4 of the wave
        const int ridx = 16 * ((lane >> 2) & 3) + 4 * w + (lane & 3);
        auto ldidx = [&](int t) {
            const int j = j_begin + t * W::BK + ridx;
            return list[j < j_end ? j : j_end - 1];
        };
        RAW ra[D][W::ACH];
        int xi[D];
        // Rows past the end of the slice are loaded like any other (their list entry is clamped: a valid row) and
        // zeroed on the way to LDS.  (Zeroed right after the load, the select is a USE of the loaded value: hipcc put
        // the selects of all register sets at the end of the unrolled loop body, and the wait in front of them
        // drained every load in flight once per D tiles.)
        auto load_tile = [&](auto par) {
            constexpr int P = decltype(par)::value;
#pragma unroll
            for (int i = 0; i < W::ACH; ++i) {
                const int row = __builtin_amdgcn_ds_bpermute((4 * i + q4) * 4, xi[P]);
                ra[P][i] = a_load(row, pc * 8);
            }
        };
        auto to_lds = [&](auto par, int t) {
            constexpr int P = decltype(par)::value;
            unsigned char* As = lds + (t & 1) * W::ABYTES;
            const int left = j_end - (j_begin + t * W::BK);
#pragma unroll
            for (int i = 0; i < W::ACH; ++i) {
                const int r = 16 * i + 4 * w + q4;
                const u32x4 v = cvt(ra[P][i]);
                *(u32x4*)(As + r * W::P1 + ((pc ^ (w2_swz<W::SW1>(r) << 1)) << 4)) = r < left ? v : u32x4{0u, 0u, 0u, 0u};
            }
        };
        using S0 = std::integral_constant<int, 0>;
        static_for<D>([&](auto d) { xi[decltype(d)::value] = ldidx(decltype(d)::value); });
        static_for<D>([&](auto d) {
            load_tile(d);
            xi[decltype(d)::value] = ldidx(decltype(d)::value + D);
        });
        to_lds(S0{}, 0);
        load_tile(S0{});
        xi[0] = ldidx(2 * D);
        __syncthreads();                               // tile 0 ready
        // steady state, branch-free: iteration j writes tile j + 1 (set (j + 1) % D), requests tile j + 1 + D into the
        // same set and that set's list entries of tile j + 1 + 2 D
        int kt = 0;
        for (; kt + D < nk; kt += D) {
            static_for<D>([&](auto u) {
                constexpr int U = decltype(u)::value;
                using P = std::integral_constant<int, (U + 1) % D>;
                to_lds(P{}, kt + U + 1);
                load_tile(P{});
                xi[P::value] = ldidx(kt + U + 1 + 2 * D);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);     // keeps the next tile's selects (uses of loads in flight) below
            });
        }
        for (int j = kt; j < nk; ++j) {                // the last tiles: everything they need has been requested
            if (j + 1 < nk) {
                static_for<D>([&](auto d) {
                    if ((j + 1) % D == decltype(d)::value) to_lds(d, j + 1);
                });
            }
            __syncthreads();
        }
        return false;
    }
    // ----------------------------------------- compute waves ------------------------------------------------------
    const int w1 = wave / W::W2, w2 = wave % W::W2;
    const int rowsel = 4 * (lane >> 4) + ((lane >> 2) & 3), p4 = (lane & 3) * 4;
    const int s1 = w2_swz<W::SW1>(rowsel), s2 = w2_swz<W::SW2>(rowsel);
    __syncthreads();                                   // tile 0 ready
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned char* As = lds + (kt & 1) * W::ABYTES + rowsel * W::P1 + p4 * 2;
        const unsigned char* Bs = lds + W::BOFF + st * W::BBYTES + rowsel * W::P2 + p4 * 2;
        st = st == 2 ? 0 : st + 1;
#pragma unroll
        for (int ks = 0; ks < W::BK / 32; ++ks) {
            bf16x8 fa[W::T1];
#pragma unroll
            for (int t1 = 0; t1 < W::T1; ++t1)
                fa[t1] = tr_frag(As + ks * 32 * W::P1 + (((w1 * (W::BN1 / W::W1) / 16 + t1) ^ s1) << 5), W::P1);
#pragma unroll
            for (int t2 = 0; t2 < W::T2; ++t2) {
                const bf16x8 fb = tr_frag(Bs + ks * 32 * W::P2 + (((w2 * (W::BN2 / W::W2) / 16 + t2) ^ s2) << 5), W::P2);
#pragma unroll
                for (int t1 = 0; t1 < W::T1; ++t1)
                    acc[t1][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t1], fb, acc[t1][t2], 0, 0, 0);
            }
        }
        __syncthreads();                               // tile kt consumed; tile kt + 1 written by the loaders
    }
    return true;
}

template <class W, class Epi>
__device__ __forceinline__ void wgrad2_epilogue(const f32x4 (&acc)[W::T1][W::T2], Epi epi) {
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

template <class W>
__device__ __forceinline__ void wgrad2_zero(f32x4 (&acc)[W::T1][W::T2]) {
#pragma unroll
    for (int t1 = 0; t1 < W::T1; ++t1)
#pragma unroll
        for (int t2 = 0; t2 < W::T2; ++t2) acc[t1][t2] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ---- results without atomics ---------------------------------------------------------------------------------
// The first version of these kernels added every partial tile to the gradient tensors with fp32 atomics like
// wgrad.hpp does, and was bound by them: 8-10 M atomic adds per launch took ~150 us, whatever the GEMM in front of
// them did (354 -> 326, 197 -> 154, 191 -> 172 us).  Instead every workgroup stores its partial tile with plain
// 16-byte stores, in accumulator-fragment order (1 KB per wave instruction), into a scratch slab
// [row split z][tile][wave][t1][t2][lane][4] and a second launch adds the row splits of a tile in FIXED order and
// does the one read-modify-write per gradient element (single writer: the results no longer depend on timing).
// The slab is the stream-K slab of K1 (SAD_WS_K1SLABS), idle while the parameter gradients are computed.
template <class W>
__device__ __forceinline__ void wgrad2_to_slab(const f32x4 (&acc)[W::T1][W::T2], float* slab, long long part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* base = slab + (part * 8 + wave) * (long long)(W::T1 * W::T2 * 256);
#pragma unroll
    for (int t1 = 0; t1 < W::T1; ++t1)
#pragma unroll
        for (int t2 = 0; t2 < W::T2; ++t2) *(f32x4*)&base[((t1 * W::T2 + t2) * 64 + lane) * 4] = acc[t1][t2];
}
template <class W>
__host__ __device__ constexpr long long wgrad2_tile_floats() { return (long long)W::BN1 * W::BN2; }

// d embedding.weight: the vocabulary tile (128 columns) that holds the ragged end of V starts at V - 128 instead
// (V >= 128): every load is a whole in-row chunk, and the rows it shares with the tile before are dropped when the
// results are written.
__host__ __device__ __forceinline__ int w2_emb_first(int tile, int V) { return tile * 128 + 128 <= V ? tile * 128 : V - 128; }

enum { W2_EMB = 0, W2_CAT = 1, W2_HH = 2 };
struct W2Out {
    float* gemb;
    float *gih0, *gih1, *glin;
    CatBias cb;
    float *ghh0, *ghh1;
};
// grid: (tiles, 8 * T1 * T2 * 64 / 256); tile = n1 tile (emb, cat) or (dir * t1n + n1 tile) * t2n + n2 tile (hh);
// `list`: ROWS_ANY, or ROWS_FWD (+ dir for hh): which row splits hold rows is recomputed from the list length.
template <class W, int KIND>
__global__ __launch_bounds__(256) void k_wgrad2_reduce(Shape s, const float* __restrict__ slab, const int* __restrict__ nrows,
                                                      int ntiles, int zs, W2Out o, int t2n, int ecol) {
    const int tile = blockIdx.x;
    const int idx = blockIdx.y * 256 + threadIdx.x;                  // (wave, t1, t2, lane)
    constexpr int PER_WAVE = W::T1 * W::T2 * 64;
    const int wave = idx / PER_WAVE, rem = idx - wave * PER_WAVE;
    const int t12 = rem >> 6, lane = rem & 63, t1 = t12 / W::T2, t2 = t12 - t1 * W::T2;
    int dir = 0, n1t = tile, n2t = 0;
    if (KIND == W2_HH) {
        n2t = tile % t2n;
        const int r = tile / t2n, t1n = s.Gp / W::BN1;
        dir = r / t1n;
        n1t = r - dir * t1n;
    }
    const int count = nrows[KIND == W2_HH ? ROWS_FWD + dir : ROWS_ANY];
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < zs; ++z) {
        int jb, je;
        if (!wgrad_slice(count, z, zs, &jb, &je)) break;
        v += *(const f32x4*)&slab[(((long long)z * ntiles + tile) * 8 * PER_WAVE + idx) * 4];
    }
    const int w1 = wave / W::W2, w2 = wave % W::W2;
    const int n2 = n2t * W::BN2 + w2 * (W::BN2 / W::W2) + 16 * t2 + (lane & 15);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n1 = n1t * W::BN1 + w1 * (W::BN1 / W::W1) + 16 * t1 + 4 * (lane >> 4) + e;
        const float x = v[e];
        if (KIND == W2_EMB) {
            const int n1s = n1 - n1t * W::BN1 + w2_emb_first(n1t, s.V);      // the last tile starts at V - 128
            if (n1s >= n1t * W::BN1 && n1s < s.V && n2 < s.E) o.gemb[(long long)n1s * s.E + n2] += x;
        } else if (KIND == W2_HH) {
            float* g = dir ? o.ghh1 : o.ghh0;
            const int row = gate_row(n1, s.H);
            if (g != nullptr && row >= 0 && n2 < s.H) g[(long long)row * s.H + n2] += x;
        } else {
            const int Gp = s.Gp, H = s.H, E = s.E;
            if (n2 == ecol) {                                    // column sums of DZ -> biases
                if (n1 < 2 * Gp) {
                    const int d = n1 >= Gp ? 1 : 0;
                    const int row = gate_row(n1 - d * Gp, H);
                    float* bi = d ? o.cb.bih1 : o.cb.bih0;
                    float* bh = d ? o.cb.bhh1 : o.cb.bhh0;
                    if (row >= 0 && bi != nullptr) bi[row] += x;
                    if (row >= 0 && bh != nullptr) bh[row] += x;
                } else if (n1 - 2 * Gp < H && o.cb.blin != nullptr) {
                    o.cb.blin[n1 - 2 * Gp] += x;
                }
            } else if (n2 < E) {
                float* dst = nullptr;
                if (n1 < 2 * Gp) {
                    const int d = n1 >= Gp ? 1 : 0;
                    const int row = gate_row(n1 - d * Gp, H);
                    float* g = d ? o.gih1 : o.gih0;
                    if (row >= 0 && g != nullptr) dst = g + (long long)row * E + n2;
                } else {
                    const int j = n1 - 2 * Gp;
                    if (j < H && o.glin != nullptr) dst = o.glin + (long long)j * E + n2;
                }
                if (dst != nullptr) *dst += x;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dWcat = DZ^T . X (+ the bias gradients as the ones column, see k_wgrad_cat).  Ep == 320, Ncat % 128 == 0.
// grid: (Ncat / 128, 1, row splits)
// ------------------------------------------------------------------------------------------------
typedef W2Cfg<128, 320, 2, 4> W2Cat;
__global__ __launch_bounds__(1024) void k_wgrad2_cat(Shape s, const bf16raw* __restrict__ DZ, const bf16raw* __restrict__ X,
                                                   const int* __restrict__ rows, const int* __restrict__ nrows,
                                                   float* __restrict__ gih0, float* __restrict__ gih1,
                                                   float* __restrict__ glin, CatBias cb, float* __restrict__ slab) {
    typedef W2Cat W;
    extern __shared__ __attribute__((aligned(1024))) unsigned char wg_lds[];
    const int n10 = blockIdx.x * W::BN1;
    int j_begin, j_end;
    if (!wgrad_slice(nrows[ROWS_ANY], blockIdx.z, gridDim.z, &j_begin, &j_end)) return;
    const int Ncat = s.Ncat, Ep = s.Ep;
    const bool ones = cb.bih0 != nullptr || cb.bhh0 != nullptr || cb.bih1 != nullptr || cb.bhh1 != nullptr || cb.blin != nullptr;
    {   // skip column blocks whose destinations are all absent (frozen tensors)
        const int lo = n10, hi = n10 + W::BN1 - 1;
        const bool w0 = gih0 != nullptr && lo < s.Gp, w1 = gih1 != nullptr && hi >= s.Gp && lo < 2 * s.Gp;
        const bool w2 = glin != nullptr && hi >= 2 * s.Gp;
        const bool b0 = (cb.bih0 != nullptr || cb.bhh0 != nullptr) && lo < s.Gp;
        const bool b1 = (cb.bih1 != nullptr || cb.bhh1 != nullptr) && hi >= s.Gp && lo < 2 * s.Gp;
        const bool b2 = cb.blin != nullptr && hi >= 2 * s.Gp;
        if (!(w0 || w1 || w2) && !((b0 || b1 || b2) && s.E < Ep)) return;
    }
    const int ecol = ones && s.E < Ep ? s.E : -1;
    f32x4 acc[W::T1][W::T2];
    wgrad2_zero<W>(acc);
    if (!wgrad2_core<W, u32x4, 4>(
        rows, j_begin, j_end,
        [&](int m, int c) { return *(const u32x4*)(DZ + (long long)m * Ncat + n10 + c); }, Ident(),
        [&](int m, int c) { return (const void*)(X + (long long)m * Ep + c); },
        [&](unsigned char* chunk, int c) {
            // X's padding column E carries a 1 in every row: dWcat's column E is the column sum of DZ
            if (ecol >= c && ecol < c + 8) lds_store16b(chunk + (ecol - c) * 2, 0x3f80u);
        },
        acc, wg_lds))
        return;
    if (slab != nullptr) {
        wgrad2_to_slab<W>(acc, slab, (long long)blockIdx.z * gridDim.x + blockIdx.x);
        return;
    }
    const int E = s.E, H = s.H, Gp = s.Gp;
    wgrad2_epilogue<W>(acc, [&](int r1, int n2, float v) {
        const int n1 = n10 + r1;
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
// dWhh[d] = sum_t dZ_t[d]^T . h_prev(t)[d]  (see k_wgrad_hh).  Gp % 128 == 0, Hp % 256 == 0.
// grid: (Gp / 128, Hp / 256, 2 * row splits)
// ------------------------------------------------------------------------------------------------
typedef W2Cfg<128, 256, 2, 4> W2Hh;
__global__ __launch_bounds__(1024) void k_wgrad2_hh(Shape s, const bf16raw* __restrict__ DZ, const bf16raw* __restrict__ Hs,
                                                  const int* __restrict__ rows, const int* __restrict__ nrows,
                                                  float* __restrict__ ghh0, float* __restrict__ ghh1, float* __restrict__ slab) {
    typedef W2Hh W;
    extern __shared__ __attribute__((aligned(1024))) unsigned char wg_lds[];
    const int dir = blockIdx.z & 1;
    float* g = dir ? ghh1 : ghh0;
    if (g == nullptr) return;
    const int n10 = blockIdx.x * W::BN1, n20 = blockIdx.y * W::BN2;
    int j_begin, j_end;
    if (!wgrad_slice(nrows[ROWS_FWD + dir], blockIdx.z >> 1, gridDim.z >> 1, &j_begin, &j_end)) return;
    const int* list = rows + (long long)(ROWS_FWD + dir) * s.M;
    const int Ncat = s.Ncat, Hp = s.Hp, Gp = s.Gp;
    const long long shift = dir ? s.B : -s.B;         // token-row offset of the previous recurrence step
    f32x4 acc[W::T1][W::T2];
    wgrad2_zero<W>(acc);
    if (!wgrad2_core<W, u32x4, 4>(
        list, j_begin, j_end,
        [&](int m, int c) { return *(const u32x4*)(DZ + (long long)m * Ncat + dir * Gp + n10 + c); }, Ident(),
        [&](int m, int c) { return (const void*)(Hs + ((m + shift) * 2 + dir) * Hp + n20 + c); },
        [](unsigned char*, int) {}, acc, wg_lds))
        return;
    if (slab != nullptr) {
        const long long tile = ((long long)dir * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y;
        wgrad2_to_slab<W>(acc, slab, (long long)(blockIdx.z >> 1) * (2 * gridDim.x * gridDim.y) + tile);
        return;
    }
    const int H = s.H;
    wgrad2_epilogue<W>(acc, [&](int r1, int r2, float v) {
        const int n1 = n10 + r1, n2 = n20 + r2;
        if (n2 >= H) return;
        const int row = gate_row(n1, H);
        if (row >= 0) unsafeAtomicAdd(g + (long long)row * H + n2, v);
    });
}

// ------------------------------------------------------------------------------------------------
// dW_emb = S^T . dX for soft tokens (see k_wgrad_emb).  Ep == 320.  S passes through registers (its rows are 2- or
// 4-byte aligned), dX by LDS-direct loads.  grid: (8 * xcd_tiles, 1, row splits), vocabulary tiles of 128.
// ------------------------------------------------------------------------------------------------
typedef W2Cfg<128, 320, 2, 4> W2Emb;
template <int KIND>
__global__ __launch_bounds__(1024) void k_wgrad2_emb(Shape s, const void* __restrict__ src, long long sT, long long sB,
                                                    const bf16raw* __restrict__ DX, const int* __restrict__ rows,
                                                    const int* __restrict__ nrows, float* __restrict__ gemb, int xcd_tiles,
                                                    float* __restrict__ slab, int ntiles) {
    typedef W2Emb W;
    typedef SoftRow<KIND> SR;
    typedef typename SR::Elem Elem;
    extern __shared__ __attribute__((aligned(1024))) unsigned char wg_lds[];
    // workgroups b, b + 8, ... share an XCD and take NEIGHBOURING vocabulary tiles (see k_wgrad_emb)
    const int tile1 = (int)(blockIdx.x & 7) * xcd_tiles + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= xcd_tiles) return;
    if (tile1 * W::BN1 >= s.V) return;
    const int n10 = w2_emb_first(tile1, s.V);
    int j_begin, j_end;
    if (!wgrad_slice(nrows[ROWS_ANY], blockIdx.z, gridDim.z, &j_begin, &j_end)) return;
    const int V = s.V, Ep = s.Ep, B = s.B;
    f32x4 acc[W::T1][W::T2];
    wgrad2_zero<W>(acc);
    if (!wgrad2_core<W, typename SR::Raw, (sizeof(typename SR::Raw) > 16 ? 2 : 4)>(
        rows, j_begin, j_end,
        [&](int m, int c) {
            const int t = m / B, b = m - t * B;
            return SR::load_raw((const Elem*)src + (long long)t * sT + (long long)b * sB + n10 + c);
        },
        typename SR::Cvt(), [&](int m, int c) { return (const void*)(DX + (long long)m * Ep + c); },
        [](unsigned char*, int) {}, acc, wg_lds))
        return;
    if (slab != nullptr) {
        wgrad2_to_slab<W>(acc, slab, (long long)blockIdx.z * ntiles + tile1);
        return;
    }
    const int E = s.E;
    wgrad2_epilogue<W>(acc, [&](int r1, int n2, float v) {
        const int n1 = n10 + r1;
        if (n1 >= tile1 * W::BN1 && n1 < V && n2 < E) unsafeAtomicAdd(gemb + (long long)n1 * E + n2, v);
    });
}

}  // namespace sad
