// Parameter gradients, second generation (round 3): the three large TN GEMMs of the unfrozen mode
//   d rnn.weight_ih / d linear_emb.weight (dWcat = DZ^T . X), d rnn.weight_hh (DZ^T . h_prev), d embedding.weight (S^T . dX)
// (autograd of transfer/models.py:898,906,925; what loss.backward() leaves in .grad, classifier/trainers.py:371-373).
//
// Why a second core (wgrad.hpp's wgrad_core stays for the small / odd shapes): at cfg3 the first generation ran
// 64-column tiles of the long operand (48 KB of operands per 64-row stage for 2.6 MFLOP: a CU takes in ~60 GB/s, so
// 0.8 us per stage against 0.27 us of MFMA), staged every tile through two register slots (228 VGPRs for 40
// accumulators; 128-column tiles spilled) and took 354 / 197 / 191 us (0.11-0.15 of the MFMA peak).  Here
//   * tiles are 128 x 320 (x 256 for W_hh): 56 KB per stage for 5.2 MFLOP;
//   * the short operand (X, dX, Hs: L2-resident, re-read by every column tile) goes global -> LDS by LDS-direct
//     loads (global_load_lds_dwordx4: no registers, no ds_write); the long operand (DZ, S: streamed from HBM once)
//     passes through three register sets (tiles k+2, k+3 in flight: HBM latency, not bandwidth, is what a single
//     tile in flight ran at);
//   * the LDS tiles are UNPADDED row-major images (an LDS-direct instruction writes 64 consecutive 16-byte chunks)
//     with the 32-byte column pairs XOR-swizzled per row, so that the transposing fragment reads
//     (ds_read_b64_tr_b16, see wgrad.hpp) stay conflict-free: a half wave reads 8 rows x 32 bytes, which must fall
//     into 8 different 8-bank groups; with pitch P the row term is (P/32 * r) mod 8 -- 0 for P = 256 / 512,
//     4r mod 8 for P = 640 -- and the swizzle supplies the missing bits (r & 7, resp. (r >> 1) & 3);
//   * two LDS stages, loads of tile k+2 issued right after tile k is consumed, counted vmcnt waits (only the oldest
//     tile is waited for).  The workgroup's slice of the compacted row list is copied into LDS once: fetched per
//     tile by vector loads, the pending index register crossed the loop's back edge and hipcc put s_waitcnt
//     vmcnt(0) in front of the copy that carries it over -- behind the LDS-direct loads issued a moment before,
//     i.e. no load was ever in flight during the MFMAs.
// Results are accumulated with fp32 atomics into the caller's state_dict-layout tensors exactly like wgrad.hpp does.
#pragma once
#include "wgrad.hpp"

namespace sad {

// SW: 0 -> pair swizzle r & 7 (pitch/32 = 0 mod 8), 1 -> (r >> 1) & 3 (pitch/32 = 4 mod 8)
template <int BN> struct W2Swz { static constexpr int kind = ((BN * 2 / 32) % 8 == 0) ? 0 : 1; };
template <int KIND> __device__ __forceinline__ int w2_swz(int r) { return KIND == 0 ? (r & 7) : ((r >> 1) & 3); }

// Row indices come out of LDS through inline asm: an ordinary LDS load while LDS-direct loads are pending makes
// hipcc's waitcnt pass (which cannot tell the two LDS regions apart) wait for ALL of them -- s_waitcnt vmcnt(0) in
// front of every tile's MFMAs.  Inline asm is invisible to that pass; the lgkmcnt wait is ours.
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(lds_u8*)p; }
__device__ __forceinline__ void lds_zero16(void* p) {
    const u32x4 z = u32x4{0u, 0u, 0u, 0u};
    asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr(p)), "v"(z) : "memory");
}
__device__ __forceinline__ void lds_store16b(void* p, unsigned v) {
    asm volatile("ds_write_b16 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_read_ints(int (&v)[N], const unsigned (&addr)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("ds_read_b32 %0, %1" : "=v"(v[i]) : "v"(addr[i]));
    static_assert(N >= 1 && N <= 8, "1..8 values");
    if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]));
    if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]));
    if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
    if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
    if constexpr (N == 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]));
    if constexpr (N == 6)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
    if constexpr (N == 7)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]));
    if constexpr (N == 8)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}

template <int BN1_, int BN2_, int W1_, int W2_, int D_ = 1>
struct W2Cfg {
    static constexpr int BN1 = BN1_, BN2 = BN2_, W1 = W1_, W2 = W2_, BK = 64, NT = 512;
    static constexpr int T1 = BN1_ / W1_ / 16, T2 = BN2_ / W2_ / 16;
    static constexpr int P1 = BN1_ * 2, P2 = BN2_ * 2;               // LDS row pitch (unpadded), bytes
    static constexpr int CPR1 = BN1_ / 8, CPR2 = BN2_ / 8;           // 16-byte chunks per tile row
    static constexpr int ABYTES = BK * P1, BBYTES = BK * P2, STAGE = ABYTES + BBYTES;
    static constexpr int LDS_BYTES = 2 * STAGE;                      // + 4 bytes per list position of the workgroup's slice (row indices)
    static constexpr int NI = CPR2 / 8;                              // LDS-direct instructions (64 chunks each) per wave and stage
    static constexpr int ACH = BK * CPR1 / NT;                       // register-staged A chunks per thread and stage
    static constexpr int D = D_;                                     // A register sets (tiles in flight besides the two LDS stages)
    static constexpr int SW1 = W2Swz<BN1_>::kind, SW2 = W2Swz<BN2_>::kind;
    static_assert(W1_ * W2_ == 8, "eight waves");
    static_assert(CPR2 % 8 == 0, "LDS-direct instructions must split evenly over the waves");
    static_assert(BN1_ % (W1_ * 16) == 0 && BN2_ % (W2_ * 16) == 0, "tile/wave mismatch");
    static_assert((BN1_ / 16) % (SW1 == 0 ? 8 : 4) == 0 && (BN2_ / 16) % (SW2 == 0 ? 8 : 4) == 0, "swizzle groups");
    static_assert((BK * CPR1) % NT == 0, "A staging must divide evenly");
};

// The A operand (the long one: DZ resp. the soft tokens S, streamed from HBM once) passes through registers, W::D
// tiles deep: HBM latency under load is ~2 us and an iteration ~1 us.  (With A on LDS-direct loads as well hipcc put
// s_waitcnt vmcnt(0) in front of every tile's first barrier, and LDS holds only two stages anyway.)  The B operand
// (X, dX, Hs: a few 10 MB, L2-resident, re-read by every column tile) goes global -> LDS by LDS-direct loads.
//
// a_load(row, col, ok) -> u32x4        8 bf16 of columns [col, col + 8) of the block tile; zeros when !ok; BRANCH-FREE
// b_src(row, col) -> const void*       global address of that 16-byte chunk of B; 16-byte aligned
// patch(lds_chunk_ptr, col)            called for every B chunk a lane has loaded, once it has landed (ones column)
// AVM                                  vector-memory instructions of one tile's a_load calls per lane; < 0: not static
// acc[t1][t2][e]: as in wgrad.hpp (plain operand order: 16 lanes hold 16 consecutive n2 of one row n1)
template <class W, int AVM, class RO, class AL, class BS, class PB>
__device__ __forceinline__ void wgrad2_core(int j_begin, int j_end, RO row_of, AL a_load, BS b_src, PB patch,
                                            f32x4 (&acc)[W::T1][W::T2], unsigned char* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w1 = wave / W::W2, w2 = wave % W::W2;
    const int nk = (j_end - j_begin + W::BK - 1) / W::BK;
    // ---- static map of this lane's LDS-direct slots: instruction q = wave + 8 i covers chunks 64 q .. 64 q + 63 of the B tile
    int srow[W::NI], scol[W::NI];
#pragma unroll
    for (int i = 0; i < W::NI; ++i) {
        const int g = 64 * (wave + 8 * i) + lane, r = g / W::CPR2, p = g - r * W::CPR2;
        srow[i] = r;
        scol[i] = (p ^ (w2_swz<W::SW2>(r) << 1)) * 8;
    }
    auto slot_off = [&](int i) { return W::ABYTES + (wave + 8 * i) * 1024; };     // wave-uniform
    // the slice's row indices, clamped into the list, one int per list position of the nk tiles
    int* idxl = (int*)(lds + W::LDS_BYTES);
    for (int i = tid; i < nk * W::BK; i += W::NT) {
        const int j = j_begin + i;
        idxl[i] = row_of(j < j_end ? j : j_end - 1);
    }
    __syncthreads();
    auto issue = [&](int st, int kt) {                 // B tile kt -> stage st (rows past the end: a clamped, finite row)
        unsigned char* base = lds + st * W::STAGE;
        const unsigned ix = lds_addr(idxl + kt * W::BK);
        int rowv[W::NI];
        unsigned ad[W::NI];
#pragma unroll
        for (int i = 0; i < W::NI; ++i) ad[i] = ix + 4u * (unsigned)srow[i];
        lds_read_ints<W::NI>(rowv, ad);
#pragma unroll
        for (int i = 0; i < W::NI; ++i)
            __builtin_amdgcn_global_load_lds((gmem_cv*)b_src(rowv[i], scol[i]), (lds_v*)(base + slot_off(i)), 16, 0, 0);
    };
    // register-staged A: chunk g = tid + 512 i -> tile row g / CPR1, column chunk g % CPR1
    u32x4 ra[W::D][W::ACH];
    auto a_fetch = [&](auto set, int kt) {             // kt may lie past the last tile: nothing valid, dummy loads
        constexpr int S = decltype(set)::value;
        const int ktc = kt < nk ? kt : nk - 1;
        const int left = kt < nk ? j_end - (j_begin + kt * W::BK) : 0;
        const unsigned ix = lds_addr(idxl + ktc * W::BK);
        int rowv[W::ACH];
        unsigned ad[W::ACH];
#pragma unroll
        for (int i = 0; i < W::ACH; ++i) ad[i] = ix + 4u * (unsigned)((tid + i * W::NT) / W::CPR1);
        lds_read_ints<W::ACH>(rowv, ad);
#pragma unroll
        for (int i = 0; i < W::ACH; ++i) {
            const int g = tid + i * W::NT, r = g / W::CPR1, p = g - r * W::CPR1;
            ra[S][i] = a_load(rowv[i], p * 8, r < left);
        }
    };
    auto a_store = [&](auto set, int st) {
        constexpr int S = decltype(set)::value;
        unsigned char* As = lds + st * W::STAGE;
#pragma unroll
        for (int i = 0; i < W::ACH; ++i) {
            const int g = tid + i * W::NT, r = g / W::CPR1, p = g - r * W::CPR1;
            *(u32x4*)(As + r * W::P1 + ((p ^ (w2_swz<W::SW1>(r) << 1)) << 4)) = ra[S][i];
        }
    };
    const int rowsel = 4 * (lane >> 4) + ((lane >> 2) & 3), p4 = (lane & 3) * 4;
    const int s1 = w2_swz<W::SW1>(rowsel), s2 = w2_swz<W::SW2>(rowsel);
    auto mfma_stage = [&](int st) {
        const unsigned char* As = lds + st * W::STAGE + rowsel * W::P1 + p4 * 2;
        const unsigned char* Bs = lds + st * W::STAGE + W::ABYTES + rowsel * W::P2 + p4 * 2;
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
    };
    auto patch_landed = [&](int st) {                  // B chunks this lane loaded into stage st have landed
        unsigned char* base = lds + st * W::STAGE;
#pragma unroll
        for (int i = 0; i < W::NI; ++i) patch(base + slot_off(i) + lane * 16, scol[i]);
    };
    // Vector-memory operations of a wave in issue order (steady state): ... | Afetch(k+3) DMA_B(k+1) | Afetch(k+4) DMA_B(k+2) | ...
    // At the top of iteration k tile k's B has landed once at most what the end of iteration k-1 issued (AVM + NI
    // operations) is outstanding; its A chunks were written to LDS before that.  AVM < 0 (element-guarded loads, no
    // static count): wait for everything.
    constexpr int ALLOW = W::NI + (AVM > 0 ? AVM : 0) * W::D;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    int k = 0;
    if constexpr (W::D == 2) {
        // ---- two register sets (tiles k + 2, k + 3 in flight), loop unrolled by two so each set is static.  Only
        // the configurations whose accumulators leave room use it (hipcc spills the 80-register tiles).
        a_fetch(S0{}, 0);
        a_fetch(S1{}, 1);
        a_store(S0{}, 0);
        issue(0, 0);
        a_store(S1{}, 1);
        if (nk > 1) issue(1, 1);
        a_fetch(S0{}, 2);
        a_fetch(S1{}, 3);
        // iteration k (k + 2 < nk): multiply tile k; then tile k + 2 (set k % 2) -> the stage just freed, that set
        // refilled with tile k + 4.  No branch inside: hipcc keeps counted vmcnt waits for the register sets.
        auto body = [&](auto set, int kk) {
            const int st = decltype(set)::value;       // kk & 1
            if constexpr (AVM >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ALLOW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            patch_landed(st);
            __syncthreads();
            mfma_stage(st);
            __syncthreads();                           // stage st consumed by every wave
            a_store(set, st);
            issue(st, kk + 2);                         // B first: vmcnt retires in order, so the wait for B(k) at
            a_fetch(set, kk + 4);                      // the top of iteration k must not cover A(k + 2)
        };
        while (true) {
            if (!(k + 2 < nk)) break;
            body(S0{}, k++);
            if (!(k + 2 < nk)) break;
            body(S1{}, k++);
        }
    } else {
        // ---- one register set: tile k + 2 is in flight while tile k is multiplied
        a_fetch(S0{}, 0);
        a_store(S0{}, 0);
        issue(0, 0);
        a_fetch(S0{}, 1);
        a_store(S0{}, 1);
        a_fetch(S0{}, 2);
        if (nk > 1) issue(1, 1);
        for (; k + 2 < nk; ++k) {
            const int st = k & 1;
            if constexpr (AVM >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ALLOW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            patch_landed(st);
            __syncthreads();
            mfma_stage(st);
            __syncthreads();                           // stage st consumed by every wave
            a_store(S0{}, st);
            a_fetch(S0{}, k + 3);
            issue(st, k + 2);
        }
    }
    for (; k < nk; ++k) {                              // the last two tiles: nothing left to issue
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        patch_landed(k & 1);
        __syncthreads();
        mfma_stage(k & 1);
        __syncthreads();
    }
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
            if (n1 < s.V && n2 < s.E) o.gemb[(long long)n1 * s.E + n2] += x;
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
__global__ __launch_bounds__(512) void k_wgrad2_cat(Shape s, const bf16raw* __restrict__ DZ, const bf16raw* __restrict__ X,
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
    wgrad2_core<W, W::ACH>(
        j_begin, j_end, [&](int j) { return rows[j]; },
        [&](int m, int c, bool ok) { return load16_if(DZ + (long long)m * Ncat + n10 + c, DZ, ok); },
        [&](int m, int c) { return (const void*)(X + (long long)m * Ep + c); },
        [&](unsigned char* chunk, int c) {
            // X's padding column E carries a 1 in every row: dWcat's column E is the column sum of DZ
            if (ecol >= c && ecol < c + 8) lds_store16b(chunk + (ecol - c) * 2, 0x3f80u);
        },
        acc, wg_lds);
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
typedef W2Cfg<128, 256, 2, 4, 2> W2Hh;
__global__ __launch_bounds__(512) void k_wgrad2_hh(Shape s, const bf16raw* __restrict__ DZ, const bf16raw* __restrict__ Hs,
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
    wgrad2_core<W, W::ACH>(
        j_begin, j_end, [&](int j) { return list[j]; },
        [&](int m, int c, bool ok) { return load16_if(DZ + (long long)m * Ncat + dir * Gp + n10 + c, DZ, ok); },
        [&](int m, int c) { return (const void*)(Hs + ((m + shift) * 2 + dir) * Hp + n20 + c); },
        [](unsigned char*, int) {}, acc, wg_lds);
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
__global__ __launch_bounds__(512) void k_wgrad2_emb(Shape s, const void* __restrict__ src, long long sT, long long sB,
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
    const int n10 = tile1 * W::BN1;
    if (n10 >= s.V) return;
    int j_begin, j_end;
    if (!wgrad_slice(nrows[ROWS_ANY], blockIdx.z, gridDim.z, &j_begin, &j_end)) return;
    const int V = s.V, Ep = s.Ep, B = s.B;
    f32x4 acc[W::T1][W::T2];
    wgrad2_zero<W>(acc);
    auto s_row = [&](int m, bool ok) -> const Elem* {
        const int t = m / B, b = m - t * B;
        return ok ? (const Elem*)src + (long long)t * sT + (long long)b * sB : nullptr;
    };
    auto row_of = [&](int j) { return rows[j]; };
    auto b_src = [&](int m, int c) { return (const void*)(DX + (long long)m * Ep + c); };
    auto none = [](unsigned char*, int) {};
    if (n10 + W::BN1 <= V) {     // interior vocabulary tile: whole chunks, branch-free loads
        wgrad2_core<W, W::ACH * (KIND == SRC_F32 ? 2 : 1)>(
            j_begin, j_end, row_of,
            [&](int m, int c, bool ok) { return typename SR::Cvt()(SR::load_fast(s_row(m, ok), (const Elem*)src, n10 + c)); },
            b_src, none, acc, wg_lds);
    } else {                     // the tile holding the ragged end of V: element-guarded loads
        wgrad2_core<W, -1>(
            j_begin, j_end, row_of,
            [&](int m, int c, bool ok) { return typename SR::Cvt()(SR::load(s_row(m, ok), n10 + c, V)); }, b_src, none, acc,
            wg_lds);
    }
    if (slab != nullptr) {
        wgrad2_to_slab<W>(acc, slab, (long long)blockIdx.z * ntiles + tile1);
        return;
    }
    const int E = s.E;
    wgrad2_epilogue<W>(acc, [&](int r1, int n2, float v) {
        const int n1 = n10 + r1;
        if (n1 < V && n2 < E) unsafeAtomicAdd(gemb + (long long)n1 * E + n2, v);
    });
}

}  // namespace sad
