// libsad host side: C-ABI entry points (include/sad.h), layouts, argument checks, launches.
// gfx950 only.  Nothing here allocates, frees or synchronises.
#include "../../include/sad.h"
#include "kernels.hpp"
#include "wgrad.hpp"
#include "wgrad2.hpp"
#include "lstm_persist.hpp"
#include "producer.hpp"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

using namespace sad;

// Workspace of sad_forward_noalloc: lives in the code object, one copy per GPU the library is loaded
// on, never freed.  Host code must not use this name as an address (it is a host-side stand-in):
// noalloc_scratch() resolves the device address once per device with hipGetSymbolAddress.
static __device__ __align__(256) unsigned char g_noalloc_scratch[SAD_NOALLOC_MAX_BYTES];

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define SAD_CHECK_LAUNCH(name)                                                          \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) return fail(-2, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- debug / test settings (sad_debug_set) ------------------------------------------------------
std::atomic<int> g_no_persist{-1};                       // -1: not set -> SAD_NO_PERSIST of the environment, read once
std::atomic<unsigned> g_spin_limit{LP_SPIN_LIMIT};       // bound of a hand-off wait inside the persistent recurrences
std::atomic<int> g_withhold_fwd{-1}, g_withhold_bwd{-1}; // workgroup that exits at once (forces a hand-off timeout)
std::atomic<int> g_wgrad_atomics{0};                     // 1: no slab for the gen-2 weight gradients (test hook)
std::atomic<int> g_withhold_upper_only{0};               // 1: only in the launches of RNN layers above the first

bool no_persist() {
    int v = g_no_persist.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* env = getenv("SAD_NO_PERSIST");
        v = (env != nullptr && atoi(env) != 0) ? 1 : 0;
        g_no_persist.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}

// ---- per-device facts, queried once (the occupancy / attribute calls cost microseconds each and the
// step is ~1 ms; they used to run 4-6 times per forward + backward) --------------------------------
std::mutex g_cache_mu;
constexpr int MAX_DEVICES = 64;
int g_cus[MAX_DEVICES];                                  // 0 = not queried yet
struct KernelFacts {
    int dev;
    const void* kernel;
    size_t lds;
    int threads;
    int per_cu;                                          // co-resident workgroups per CU at (threads, lds); -1 = query failed
};
constexpr int MAX_KERNEL_FACTS = 48;                     // per device; a full table evicts its oldest entry
KernelFacts g_facts[MAX_DEVICES][MAX_KERNEL_FACTS];
int g_nfacts[MAX_DEVICES], g_fact_next[MAX_DEVICES];

int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return 0;
    return dev;
}
int device_cus() {
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(g_cache_mu);
    if (g_cus[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 1;
        g_cus[dev] = cus;
    }
    return g_cus[dev];
}
// Allows `lds` bytes of dynamic LDS for `kernel` on the current device (once) and returns how many of
// its workgroups fit one CU.
int kernel_per_cu(const void* kernel, size_t lds, int threads) {
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(g_cache_mu);
    KernelFacts* facts = g_facts[dev];
    for (int i = 0; i < g_nfacts[dev]; ++i)
        if (facts[i].kernel == kernel && facts[i].lds == lds && facts[i].threads == threads) return facts[i].per_cu;
    int per_cu = -1;
    if (lds <= 65536 || hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess) per_cu = -1;
    }
    (void)hipGetLastError();
    if (g_nfacts[dev] < MAX_KERNEL_FACTS) {
        facts[g_nfacts[dev]++] = KernelFacts{dev, kernel, lds, threads, per_cu};
    } else {
        facts[g_fact_next[dev]] = KernelFacts{dev, kernel, lds, threads, per_cu};
        g_fact_next[dev] = (g_fact_next[dev] + 1) % MAX_KERNEL_FACTS;
    }
    return per_cu;
}
// Kernels with more than 64 KB of dynamic LDS: the limit has to be raised once per device.
void allow_lds(const void* kernel, size_t lds, int threads) { (void)kernel_per_cu(kernel, lds, threads); }

// ---- hand-off time-outs surfaced without a synchronisation ------------------------------------------
// One word of host-mapped (pinned) memory per device, created by the first sad_pack_weights on that device (the
// library's set-up call; the compute entry points never allocate).  A persistent recurrence whose hand-off wait
// times out raises it from the device (system-scope store: 1 = forward recurrence, 2 = BPTT) besides poisoning
// its outputs with NaN; the next sad_forward on the device reads it with a plain host load, refuses (-4) with the
// reason and clears it.  Without it a time-out in production is a NaN loss whose reason only Workspace.check()
// (a synchronising read of the workspace) could tell.
unsigned* g_hostflag[MAX_DEVICES];        // host address
unsigned* g_hostflag_dev[MAX_DEVICES];    // the same word as the device sees it
void ensure_hostflag() {
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(g_cache_mu);
    if (g_hostflag[dev] != nullptr) return;
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess || h == nullptr) { (void)hipGetLastError(); return; }
    memset(h, 0, 64);
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || d == nullptr) { (void)hipGetLastError(); (void)hipHostFree(h); return; }
    g_hostflag_dev[dev] = (unsigned*)d;
    g_hostflag[dev] = (unsigned*)h;
}
unsigned* hostflag_dev() {
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(g_cache_mu);
    return g_hostflag[dev] != nullptr ? g_hostflag_dev[dev] : nullptr;
}
// returns the raised code (and clears it), 0 if none
unsigned take_hostflag() {
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(g_cache_mu);
    volatile unsigned* f = g_hostflag[dev];
    if (f == nullptr) return 0;
    const unsigned v = *f;
    if (v != 0) *f = 0;
    return v;
}

struct PackedLayout {
    size_t wembT, wemb, wcat, wcatT, bcat, whh, whhT, whb0, whb0T, bhb0, wq, wqT, bq, wa, ba, wo, bo, total;
};
PackedLayout packed_layout(const Shape& s) {
    PackedLayout p;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t r = o;
        o = align_up(o + bytes);
        return r;
    };
    p.wembT = take((size_t)s.Ep * s.Vp * 2);
    p.wemb = take((size_t)s.Vp * s.Ep * 2);
    p.wcat = take((size_t)s.Ncat * s.Ep * 2);
    p.wcatT = take((size_t)s.Ep * s.Ncat * 2);
    p.bcat = take((size_t)s.Ncat * 4);
    p.whh = take((size_t)2 * s.Gp * s.Hp * 2);
    p.whhT = take((size_t)2 * s.Hp * s.Gp * 2);
    p.whb0 = take((size_t)s.Hp * 2 * s.Hp * 2);
    p.whb0T = take((size_t)2 * s.Hp * s.Hp * 2);
    p.bhb0 = take((size_t)s.Hp * 4);
    p.wq = take((size_t)s.Hp * s.Hp * 2);
    p.wqT = take((size_t)s.Hp * s.Hp * 2);
    p.bq = take((size_t)s.Hp * 4);
    p.wa = take((size_t)s.Hp * 4);
    p.ba = take(4);
    p.wo = take((size_t)s.L * s.Hp * 4);
    p.bo = take((size_t)s.L * 4);
    p.total = o;
    return p;
}

struct WsLayout {
    size_t off[SAD_WS_NFIELDS], bytes[SAD_WS_NFIELDS], total;
};
WsLayout ws_layout(const Shape& s) {
    WsLayout w;
    const size_t M = (size_t)s.M, B = (size_t)s.B;
    w.bytes[SAD_WS_LENS] = B * 4;
    w.bytes[SAD_WS_NEED] = M;
    w.bytes[SAD_WS_X] = M * s.Ep * 2;
    w.bytes[SAD_WS_G] = M * s.Gld * 4;
    w.bytes[SAD_WS_CS] = M * 2 * s.Hp * 4;
    w.bytes[SAD_WS_HS] = M * 2 * s.Hp * 2;
    w.bytes[SAD_WS_KK] = M * s.Hp * 4;
    w.bytes[SAD_WS_Q] = B * s.Hp * 4;
    w.bytes[SAD_WS_U] = B * s.Hp * 4;
    w.bytes[SAD_WS_CTX] = B * s.Hp * 4;
    w.bytes[SAD_WS_DZ] = M * s.Ncat * 2;
    w.bytes[SAD_WS_DX] = M * s.Ep * 2;
    w.bytes[SAD_WS_DU] = B * s.Hp * 4;
    w.bytes[SAD_WS_DQ] = B * s.Hp * 4;
    w.bytes[SAD_WS_DHFIN] = B * 2 * s.Hp * 4;
    w.bytes[SAD_WS_DC] = B * 2 * s.Hp * 4;
    w.bytes[SAD_WS_ACCF] = M * s.Ep * 4;
    w.bytes[SAD_WS_DL] = B * s.L * 4;
    w.bytes[SAD_WS_DWA] = B * s.Hp * 4;
    w.bytes[SAD_WS_DSSUM] = B * 4;
    w.bytes[SAD_WS_ROWS] = M * 3 * 4;
    w.bytes[SAD_WS_NROWS] = 4 * 4;
    w.bytes[SAD_WS_SYNC] = ((size_t)lp_err_line(s.B) + 1) * LP_SYNC_PITCH * 4;
    w.bytes[SAD_WS_DZX] = lstm_bwd_xchg_bytes(s.B, s.Gp);
    w.bytes[SAD_WS_HX] = lstm_fwd_xchg_bytes(s.B, s.Hp);
    w.bytes[SAD_WS_GA] = ga_bytes(s.T, s.B, s.Hp);
    const bool k1_sk = s.Ep == 320;                          // the stream-K form of K1 (run_softemb_fwd) and of K2a'+K3'
    w.bytes[SAD_WS_K1TILES] = k1_sk ? ((size_t)ceil_div(s.M, 128) + 4) * 4 : 0;
    w.bytes[SAD_WS_K1SLABS] = k1_sk ? (size_t)K1_MAX_WGS * 2 * K1_SLAB_FLOATS * 4 : 0;
    size_t o = 0;
    for (int i = 0; i < SAD_WS_NFIELDS; ++i) {
        w.off[i] = o;
        o = align_up(o + w.bytes[i]);
    }
    w.total = o;
    return w;
}

int check_dims(const sad_dims* d, Shape* s) {
    if (d == nullptr) return fail(-1, "dims is NULL");
    if (d->B < 1 || d->T < 1 || d->V < 1 || d->E < 1 || d->H < 1 || d->L < 1)
        return fail(-1, "dims must be positive (B=%d T=%d V=%d E=%d H=%d L=%d)", d->B, d->T, d->V, d->E, d->H, d->L);
    if (d->T > ATT_MAX_T) return fail(-1, "T=%d exceeds the supported maximum %d", d->T, ATT_MAX_T);
    if (d->L > ATT_MAX_L) return fail(-1, "L=%d exceeds the supported maximum %d", d->L, ATT_MAX_L);
    if (d->H > 2048) return fail(-1, "H=%d exceeds the supported maximum 2048", d->H);
    if ((long long)d->B * d->T > (1ll << 30)) return fail(-1, "B*T too large");
    if (d->B > 65535 * 64) return fail(-1, "B too large");
    if (d->cell != SAD_CELL_LSTM && d->cell != SAD_CELL_GRU) return fail(-1, "cell=%d is neither SAD_CELL_LSTM nor SAD_CELL_GRU", d->cell);
    *s = make_shape(d->B, d->T, d->V, d->E, d->H, d->L, d->cell);
    return 0;
}

template <class T>
T* at(void* base, size_t off) { return (T*)((char*)base + off); }
template <class T>
const T* at(const void* base, size_t off) { return (const T*)((const char*)base + off); }

int check_ptr(const void* p, const char* name) {
    if (p == nullptr) return fail(-1, "%s is NULL", name);
    return 0;
}
int check_aligned(const void* p, size_t a, const char* name) {
    if (((uintptr_t)p) % a) return fail(-1, "%s must be %zu-byte aligned", name, a);
    return 0;
}

typedef Cfg<128, 320, 2, 4, 2> CfgWideN;   // 512 threads; all of Ep=320 in one tile: S is read once
typedef Cfg<64, 320, 2, 4, 2> CfgWideN64;  // K1 forward at small batches: half-height tiles, twice the workgroups
typedef Cfg<128, 64, 2, 2, 2> CfgNarrowN;
typedef Cfg<128, 128, 2, 2, 2> CfgSquare;
// LSTM steps are small latency-bound GEMMs: 4 K-tiles of loads in flight per workgroup.
typedef Cfg<128, 128, 4, 2, 4> CfgStepBig;   // 512 threads: 32 rows x one 64-col gate block per wave
typedef Cfg<64, 64, 4, 1, 4> CfgStepSmall;   // small batch: 16 rows x one gate block per wave
typedef Cfg<64, 64, 4, 2, 4> CfgStepBwd;     // 512 threads: 16 rows x 32 units per wave

// Split-K factor for GEMMs with few output tiles and a long K (small batches): aim at ~1 workgroup
// per CU, keep >= 2 K-tiles per split.  Partial sums meet in an fp32 buffer via atomics.
int split_k(int tiles, int ktiles) {
    if (tiles >= 128 || ktiles < 4) return 1;
    int ks = 256 / tiles;
    if (ks > ktiles / 2) ks = ktiles / 2;
    if (ks > 64) ks = 64;
    return ks < 1 ? 1 : ks;
}
void launch_cvt(const float* src, bf16raw* dst, long long n, hipStream_t st) {
    const long long n8 = n / 8;
    int blocks = (int)((n8 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_cvt_bf16, dim3(blocks), dim3(256), 0, st, src, dst, n8);
}

template <class C, int KIND>
void launch_softemb_fwd(const Shape& s, const void* src, long long sT, long long sB, const unsigned char* need,
                        const bf16raw* wembT, bf16raw* X, float* accf, hipStream_t st) {
    const int tiles = ceil_div(s.M, C::BM) * ceil_div(s.Ep, C::BN);
    const int ksplit = split_k(tiles, ceil_div(s.V, C::BK));
    dim3 grid(ceil_div(s.M, C::BM), ceil_div(s.Ep, C::BN), ksplit);
    if (ksplit > 1) (void)hipMemsetAsync(accf, 0, (size_t)s.M * s.Ep * 4, st);
    hipLaunchKernelGGL((k_softemb_fwd<C, KIND>), grid, dim3(C::NT), 0, st, s, src, sT, sB, need, wembT, X, accf, ksplit);
    if (ksplit > 1) launch_cvt(accf, X, (long long)s.M * s.Ep, st);
}
template <class C, int KIND>
void launch_softemb_bwd(const Shape& s, const bf16raw* DX, const unsigned char* need, const bf16raw* wemb, void* dS,
                        long long sT, long long sB, hipStream_t st) {
    dim3 grid(ceil_div(s.V, C::BN), ceil_div(s.M, C::BM));
    hipLaunchKernelGGL((k_softemb_bwd<C, KIND>), grid, dim3(C::NT), 0, st, s, DX, need, wemb, dS, sT, sB);
}

int run_softemb_fwd(const Shape& s, const PackedLayout& pl, const WsLayout& wl, const void* packed, void* ws,
                    const void* src, int kind, long long sB, long long sT, hipStream_t st) {
    const unsigned char* need = at<unsigned char>(ws, wl.off[SAD_WS_NEED]);
    const bf16raw* wembT = at<bf16raw>(packed, pl.wembT);
    bf16raw* X = at<bf16raw>(ws, wl.off[SAD_WS_X]);
    float* accf = at<float>(ws, wl.off[SAD_WS_ACCF]);
    if (kind != SAD_SRC_F32 && kind != SAD_SRC_BF16)
        return fail(-1, "soft-embedding stage needs SAD_SRC_F32 or SAD_SRC_BF16, got %d", kind);
    if (check_aligned(src, kind == SAD_SRC_F32 ? 4 : 2, "src")) return -1;
    if (s.Ep == 320 && s.V >= 64) {
        // The shipped embedding size (E = 289..320): stream-K over the non-empty 128-row tiles, one
        // workgroup per CU with an equal share of (tile, k-tile) units; three wave roles per workgroup.
        constexpr size_t lds3 = 2 * 128 * 128 + 3 * 320 * 128;
        int* tiles = at<int>(ws, wl.off[SAD_WS_K1TILES]);
        float* slabs = at<float>(ws, wl.off[SAD_WS_K1SLABS]);
        const int cus = device_cus();
        const int nwg = cus > K1_MAX_WGS ? K1_MAX_WGS : cus;
        if (kind == SAD_SRC_F32) {
            allow_lds((const void*)k_softemb_fwd_sk<SRC_F32>, lds3, 1024);
            hipLaunchKernelGGL(k_softemb_fwd_sk<SRC_F32>, dim3(nwg), dim3(1024), lds3, st, s, src, sT, sB, need, tiles, wembT, X, slabs);
        } else {
            allow_lds((const void*)k_softemb_fwd_sk<SRC_BF16>, lds3, 1024);
            hipLaunchKernelGGL(k_softemb_fwd_sk<SRC_BF16>, dim3(nwg), dim3(1024), lds3, st, s, src, sT, sB, need, tiles, wembT, X, slabs);
        }
        hipLaunchKernelGGL(k_softemb_fwd_fix, dim3(ceil_div(s.M, 128), 8, 4), dim3(64), 0, st, s, tiles, nwg, slabs, X);
        SAD_CHECK_LAUNCH("k_softemb_fwd_sk");
        return 0;
    }
    // any other embedding size: the generic LDS-tiled kernel (split-K with fp32 atomics at small batches)
    const bool wide = s.Ep > 128;
    const bool small_m = s.M <= 16384;      // few row tiles: half-height tiles
    if (kind == SAD_SRC_F32) {
        if (wide && small_m) launch_softemb_fwd<CfgWideN64, SRC_F32>(s, src, sT, sB, need, wembT, X, accf, st);
        else if (wide) launch_softemb_fwd<CfgWideN, SRC_F32>(s, src, sT, sB, need, wembT, X, accf, st);
        else launch_softemb_fwd<CfgNarrowN, SRC_F32>(s, src, sT, sB, need, wembT, X, accf, st);
    } else {
        if (wide && small_m) launch_softemb_fwd<CfgWideN64, SRC_BF16>(s, src, sT, sB, need, wembT, X, accf, st);
        else if (wide) launch_softemb_fwd<CfgWideN, SRC_BF16>(s, src, sT, sB, need, wembT, X, accf, st);
        else launch_softemb_fwd<CfgNarrowN, SRC_BF16>(s, src, sT, sB, need, wembT, X, accf, st);
    }
    SAD_CHECK_LAUNCH("k_softemb_fwd");
    return 0;
}

int run_softemb_bwd(const Shape& s, const PackedLayout& pl, const WsLayout& wl, const void* packed, void* ws,
                    void* dsrc, int kind, long long sB, long long sT, hipStream_t st) {
    const unsigned char* need = at<unsigned char>(ws, wl.off[SAD_WS_NEED]);
    const bf16raw* wemb = at<bf16raw>(packed, pl.wemb);
    const bf16raw* DX = at<bf16raw>(ws, wl.off[SAD_WS_DX]);
    if (kind != SAD_SRC_F32 && kind != SAD_SRC_BF16)
        return fail(-1, "d(soft tokens) needs SAD_SRC_F32 or SAD_SRC_BF16, got %d", kind);
    if (check_aligned(dsrc, kind == SAD_SRC_F32 ? 4 : 2, "dsrc")) return -1;
    if (s.Ep == 320) {
        // A-stationary fast path for the shipped embedding size (E = 289..320 -> Ep = 320 = 32*10).
        // One workgroup per CU is resident at a time (LDS), and every workgroup pays a prologue (dX
        // fragments, first W tile): one round of workgroups (measured: 1024 -> 257 us, 256 -> 246 us).
        const int mtiles = ceil_div(s.M, 256), ntiles = s.Vp / 64;
        int vsplit = device_cus() / mtiles;
        vsplit = vsplit < 1 ? 1 : (vsplit > ntiles ? ntiles : vsplit);
        const int nt_per = ceil_div(ntiles, vsplit);
        const dim3 grid(mtiles, ceil_div(ntiles, nt_per));
        if (kind == SAD_SRC_F32) {          // loader + compute waves
            constexpr size_t lds = 2 * 64 * 10 * 64;
            allow_lds((const void*)k_softemb_bwd_ws<10, SRC_F32>, lds, 768);
            hipLaunchKernelGGL((k_softemb_bwd_ws<10, SRC_F32>), grid, dim3(768), lds, st, s, DX, need, wemb, dsrc, sT, sB, nt_per);
        } else {                            // loader + compute + store waves
            constexpr size_t lds2 = 2 * 64 * 10 * 64 + 2 * 256 * 64 * 2;
            allow_lds((const void*)k_softemb_bwd_ws2<10>, lds2, 768);
            hipLaunchKernelGGL((k_softemb_bwd_ws2<10>), grid, dim3(768), lds2, st, s, DX, need, wemb, (bf16raw*)dsrc, sT, sB, nt_per);
        }
        SAD_CHECK_LAUNCH("k_softemb_bwd_ws");
        return 0;
    }
    if (kind == SAD_SRC_F32) launch_softemb_bwd<CfgSquare, SRC_F32>(s, DX, need, wemb, dsrc, sT, sB, st);
    else launch_softemb_bwd<CfgSquare, SRC_BF16>(s, DX, need, wemb, dsrc, sT, sB, st);
    SAD_CHECK_LAUNCH("k_softemb_bwd");
    return 0;
}

int run_prep(const Shape& s, const WsLayout& wl, void* ws, const int32_t* lengths, uint8_t* mask, long long msb,
             long long mst, int guard, hipStream_t st) {
    unsigned char* need = at<unsigned char>(ws, wl.off[SAD_WS_NEED]);
    int* lens = at<int>(ws, wl.off[SAD_WS_LENS]);
    unsigned* sync = at<unsigned>(ws, wl.off[SAD_WS_SYNC]);
    const int sync_words = (int)(wl.bytes[SAD_WS_SYNC] / 4);
    int* rows = at<int>(ws, wl.off[SAD_WS_ROWS]);
    int* nrows = at<int>(ws, wl.off[SAD_WS_NROWS]);
    int* tiles = wl.bytes[SAD_WS_K1TILES] != 0 ? at<int>(ws, wl.off[SAD_WS_K1TILES]) : nullptr;
    if (s.M <= PREPL_MAX_M && s.B <= PREPL_MAX_B) {
        // one launch: need bits, compacted row lists, list of non-empty 128-row tiles
        allow_lds((const void*)k_prep_lists, prepl_lds_bytes(PREPL_MAX_M, PREPL_MAX_B), 1024);
        hipLaunchKernelGGL(k_prep_lists, dim3(ceil_div(s.M, 1024)), dim3(1024), prepl_lds_bytes(s.M, s.B), st, s, lengths, mask,
                           msb, mst, guard, need, lens, sync, sync_words, rows, nrows, tiles);
        SAD_CHECK_LAUNCH("k_prep_lists");
        return 0;
    }
    hipLaunchKernelGGL(k_prep, dim3(ceil_div(s.B, 64)), dim3(64), 0, st, s, lengths, mask, msb, mst, guard, need, lens, sync,
                       sync_words);
    // compacted lists of the needed token rows (in-projection, weight gradients) and of the non-empty row tiles (K1)
    hipLaunchKernelGGL(k_row_lists, dim3(ceil_div(s.M, 1024)), dim3(256), 0, st, s, need, rows, nrows);
    if (tiles != nullptr) hipLaunchKernelGGL(k_tile_list, dim3(1), dim3(1024), 0, st, s.M, need, tiles);
    SAD_CHECK_LAUNCH("k_prep");
    return 0;
}

// Weight-gradient kernels use two LDS tile buffers (up to 120 KB): dynamic LDS above the 64 KB
// static limit has to be allowed once per kernel and device.
template <class W, class K, class... Args>
void launch_wg(K kernel, dim3 grid, hipStream_t st, Args... args) {
    allow_lds((const void*)kernel, W::LDS_BYTES, W::NT);
    hipLaunchKernelGGL(kernel, grid, dim3(W::NT), W::LDS_BYTES, st, args...);
}

// The persistent recurrences need every workgroup co-resident (they wait for each other): the grid
// must fit the device at the kernel's own occupancy (queried once per device and LDS size).
// sad_debug_set(SAD_DBG_NO_PERSIST, 1) / SAD_NO_PERSIST=1 force the per-step kernels.
bool persist_fits(const void* kernel, size_t lds, int grid, int threads) {
    if (no_persist()) return false;
    const int per_cu = kernel_per_cu(kernel, lds, threads);
    return per_cu >= 1 && grid <= per_cu * device_cus();
}
// Sentence tile and kernel form of a persistent recurrence: tiles below 128 sentences (small batches) run the split
// forms of the kernels (lstm_persist.hpp: MS / KS), which need a little LDS for the partial sums.
struct PersistPlan { int bm; bool split; const void* kernel; size_t lds; };
// forward recurrence: half-width workgroups, two per CU
PersistPlan lstm_fwd_plan(const Shape& s) {
    const size_t lds_ms = lstm_fwd2_lds_bytes(s.Hp);
    const int per_cu = kernel_per_cu((const void*)k_lstm_fwd_persist2<true>, lds_ms, 256);
    const int bm = lstm_fwd_tile(s.B, s.Gp, (per_cu < 1 ? 1 : per_cu) * device_cus());
    if (bm < LF_BM) return PersistPlan{bm, true, (const void*)k_lstm_fwd_persist2<true>, lds_ms};
    return PersistPlan{LF_BM, false, (const void*)k_lstm_fwd_persist2<false>, lstm_fwd2_lds_bytes(s.Hp)};
}
bool lstm_fwd_persist_fits(const Shape& s, int* grid) {
    if (s.Hp > 32 * LF_KS_MAX || s.Hp % 32 != 0 || s.Gp % LF_BN != 0) return false;
    if (lstm_fwd_xchg_bytes(s.B, s.Hp) >= (1ull << 31)) return false;
    // groups padded to a multiple of 8 (one XCD per group, see the kernel) when that still fits
    const PersistPlan p = lstm_fwd_plan(s);
    const int groups = 2 * ceil_div(s.B, p.bm), ctiles = s.Gp / LF_BN;
    *grid = ceil_to(groups, 8) * ctiles;
    if (persist_fits(p.kernel, p.lds, *grid, 256)) return true;
    *grid = groups * ctiles;
    return persist_fits(p.kernel, p.lds, *grid, 256);
}
PersistPlan lstm_bwd_plan(const Shape& s) {
    const int bm = lstm_bwd_tile(s.B, s.Hp, device_cus());
    if (bm < LB_BM) return PersistPlan{bm, true, (const void*)k_lstm_bwd_persist<true>, lstm_bwd_persist_lds_bytes(s.Gp) + LB_RED_BYTES};
    return PersistPlan{LB_BM, false, (const void*)k_lstm_bwd_persist<false>, lstm_bwd_persist_lds_bytes(s.Gp) + 8 * LB_STRIP_BYTES};
}
bool lstm_bwd_persist_fits(const Shape& s, int* grid) {
    if (s.Gp > 128 * LB_SG_MAX || s.Hp % LB_BN != 0 || (long long)s.B * s.Ncat * 2 >= (1ll << 31)) return false;
    const PersistPlan p = lstm_bwd_plan(s);
    const int groups = 2 * ceil_div(s.B, p.bm), ctiles = s.Hp / LB_BN;
    *grid = ceil_to(groups, 8) * ctiles;
    if (persist_fits(p.kernel, p.lds, *grid, 512)) return true;
    *grid = groups * ctiles;
    return persist_fits(p.kernel, p.lds, *grid, 512);
}
// fp16 activated gates (SAD_WS_GA) are written by the persistent forward recurrence and read by the
// persistent BPTT: used when both run (same decision in sad_forward and sad_backward as long as the
// debug settings do not change between the two calls).
bool half_gates(const Shape& s) {
    int g1 = 0, g2 = 0;
    return lstm_fwd_persist_fits(s, &g1) && lstm_bwd_persist_fits(s, &g2);
}

// The in-projection writes the gate pre-activations as fp16 records straight into SAD_WS_GA (and the forward
// recurrence reads them there) when the A-stationary in-projection kernel runs AND both recurrences are persistent;
// otherwise they are fp32 rows of SAD_WS_G.
bool ga_preact(const Shape& s) { return s.Ep == 320 && s.Ncat % 64 == 0 && half_gates(s); }

// sad_pack_weights collects its jobs here and launches them together (k_pack_all)
struct PackBatch {
    PackJobs jobs;
    int n = 0;
    long long biggest = 0;
};
template <typename OutT>
void launch_pack(PackBatch& pb, OutT* dst, int rows, int cols, long long ldd, const float* src, const float* src2,
                 long long lds, IdxMap rmap, IdxMap cmap, int tr) {
    PackJob& J = pb.jobs.j[pb.n++];
    J.dst = dst;
    J.src = src;
    J.src2 = src2;
    J.ldd = ldd;
    J.lds = lds;
    J.rmap = rmap;
    J.cmap = cmap;
    J.rows = rows;
    J.cols = cols;
    J.tr = tr;
    J.bf16 = sizeof(OutT) == 2;
    const long long total = (long long)rows * cols;
    if (total > pb.biggest) pb.biggest = total;
}
void run_pack(PackBatch& pb, hipStream_t st) {
    int blocks = (int)((pb.biggest + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_pack_all, dim3(blocks, pb.n), dim3(256), 0, st, pb.jobs);
}

}  // namespace

extern "C" {

int sad_version(void) { return SAD_VERSION; }
const char* sad_last_error(void) { return g_err; }

int sad_uses_half_gates(const sad_dims* d) {
    Shape s;
    if (check_dims(d, &s)) return -1;
    return half_gates(s) ? 1 : 0;
}

int sad_padded_dims(const sad_dims* d, int32_t out[5]) {
    Shape s;
    if (check_dims(d, &s)) return -1;
    out[0] = s.Ep; out[1] = s.Hp; out[2] = s.Gp; out[3] = s.Vp; out[4] = s.Ncat;
    return 0;
}

size_t sad_packed_bytes(const sad_dims* d) {
    Shape s;
    if (check_dims(d, &s)) return 0;
    return packed_layout(s).total;
}

size_t sad_workspace_bytes(const sad_dims* d) {
    Shape s;
    if (check_dims(d, &s)) return 0;
    return ws_layout(s).total;
}

int sad_workspace_field(const sad_dims* d, int field, size_t* offset, size_t* bytes) {
    Shape s;
    if (check_dims(d, &s)) return -1;
    if (field < 0 || field >= SAD_WS_NFIELDS) return fail(-1, "unknown workspace field %d", field);
    WsLayout w = ws_layout(s);
    if (offset) *offset = w.off[field];
    if (bytes) *bytes = w.bytes[field];
    return 0;
}

int sad_pack_weights(const sad_dims* d, const sad_weights* w, void* packed, void* stream) {
    Shape s;
    if (check_dims(d, &s)) return -1;
    if (check_ptr(w, "weights") || check_ptr(packed, "packed") || check_aligned(packed, 256, "packed")) return -1;
    const void* need[] = {w->embedding_weight, w->rnn_weight_ih[0], w->rnn_weight_ih[1], w->rnn_weight_hh[0],
                          w->rnn_weight_hh[1], w->rnn_bias_ih[0], w->rnn_bias_ih[1], w->rnn_bias_hh[0],
                          w->rnn_bias_hh[1], w->hidden_bridge0_weight, w->hidden_bridge0_bias, w->linear_emb_weight,
                          w->linear_emb_bias, w->linear_q_weight, w->linear_q_bias, w->linear_attn_weight,
                          w->linear_attn_bias, w->output_layer_weight, w->output_layer_bias};
    for (size_t i = 0; i < sizeof(need) / sizeof(need[0]); ++i)
        if (need[i] == nullptr) return fail(-1, "sad_weights field #%zu is NULL", i);
    hipStream_t st = (hipStream_t)stream;
    ensure_hostflag();                                       // set-up call: the time-out word of this device
    const PackedLayout p = packed_layout(s);
    const IdxMap idE{MAP_ID, s.E, s.H, s.Hp}, idV{MAP_ID, s.V, s.H, s.Hp}, idH{MAP_ID, s.H, s.H, s.Hp};
    const IdxMap idL{MAP_ID, s.L, s.H, s.Hp}, id1{MAP_ID, 1, s.H, s.Hp};
    const IdxMap gate{MAP_GATE, 0, s.H, s.Hp}, hcat{MAP_HCAT, 0, s.H, s.Hp};
    PackBatch pb;
    // embedding: [Ep][Vp] (K1's Bt) and [Vp][Ep] (K1' Bt, and the hard-token gather table)
    launch_pack(pb, at<bf16raw>(packed, p.wembT), s.Ep, s.Vp, s.Vp, w->embedding_weight, nullptr, s.E, idE, idV, 1);
    launch_pack(pb, at<bf16raw>(packed, p.wemb), s.Vp, s.Ep, s.Ep, w->embedding_weight, nullptr, s.E, idV, idE, 0);
    // fused in-projection [Ncat][Ep], its transpose [Ep][Ncat], bias [Ncat]
    for (int dir = 0; dir < 2; ++dir) {
        launch_pack(pb, at<bf16raw>(packed, p.wcat) + (size_t)dir * s.Gp * s.Ep, s.Gp, s.Ep, s.Ep, w->rnn_weight_ih[dir],
                    nullptr, s.E, gate, idE, 0);
        launch_pack(pb, at<bf16raw>(packed, p.wcatT) + (size_t)dir * s.Gp, s.Ep, s.Gp, s.Ncat, w->rnn_weight_ih[dir],
                    nullptr, s.E, idE, gate, 1);
        launch_pack(pb, at<float>(packed, p.bcat) + (size_t)dir * s.Gp, 1, s.Gp, s.Gp, w->rnn_bias_ih[dir],
                    w->rnn_bias_hh[dir], 4 * s.H, id1, gate, 0);
        launch_pack(pb, at<bf16raw>(packed, p.whh) + (size_t)dir * s.Gp * s.Hp, s.Gp, s.Hp, s.Hp, w->rnn_weight_hh[dir],
                    nullptr, s.H, gate, idH, 0);
        launch_pack(pb, at<bf16raw>(packed, p.whhT) + (size_t)dir * s.Hp * s.Gp, s.Hp, s.Gp, s.Gp, w->rnn_weight_hh[dir],
                    nullptr, s.H, idH, gate, 1);
    }
    launch_pack(pb, at<bf16raw>(packed, p.wcat) + (size_t)2 * s.Gp * s.Ep, s.Hp, s.Ep, s.Ep, w->linear_emb_weight, nullptr,
                s.E, idH, idE, 0);
    launch_pack(pb, at<bf16raw>(packed, p.wcatT) + (size_t)2 * s.Gp, s.Ep, s.Hp, s.Ncat, w->linear_emb_weight, nullptr, s.E,
                idE, idH, 1);
    launch_pack(pb, at<float>(packed, p.bcat) + (size_t)2 * s.Gp, 1, s.Hp, s.Hp, w->linear_emb_bias, nullptr, s.H, id1, idH,
                0);
    // hidden_bridges[0]: [Hp][2Hp] over k = [h_fwd | h_rev], and its transpose
    launch_pack(pb, at<bf16raw>(packed, p.whb0), s.Hp, 2 * s.Hp, 2 * s.Hp, w->hidden_bridge0_weight, nullptr, 2 * s.H, idH,
                hcat, 0);
    launch_pack(pb, at<bf16raw>(packed, p.whb0T), 2 * s.Hp, s.Hp, s.Hp, w->hidden_bridge0_weight, nullptr, 2 * s.H, hcat,
                idH, 1);
    launch_pack(pb, at<float>(packed, p.bhb0), 1, s.Hp, s.Hp, w->hidden_bridge0_bias, nullptr, s.H, id1, idH, 0);
    launch_pack(pb, at<bf16raw>(packed, p.wq), s.Hp, s.Hp, s.Hp, w->linear_q_weight, nullptr, s.H, idH, idH, 0);
    launch_pack(pb, at<bf16raw>(packed, p.wqT), s.Hp, s.Hp, s.Hp, w->linear_q_weight, nullptr, s.H, idH, idH, 1);
    launch_pack(pb, at<float>(packed, p.bq), 1, s.Hp, s.Hp, w->linear_q_bias, nullptr, s.H, id1, idH, 0);
    launch_pack(pb, at<float>(packed, p.wa), 1, s.Hp, s.Hp, w->linear_attn_weight, nullptr, s.H, id1, idH, 0);
    launch_pack(pb, at<float>(packed, p.ba), 1, 1, 1, w->linear_attn_bias, nullptr, 1, id1, id1, 0);
    launch_pack(pb, at<float>(packed, p.wo), s.L, s.Hp, s.Hp, w->output_layer_weight, nullptr, s.H, idL, idH, 0);
    launch_pack(pb, at<float>(packed, p.bo), 1, s.L, s.L, w->output_layer_bias, nullptr, s.L, id1, idL, 0);
    static_assert(PACK_MAX_JOBS >= 28, "job table too small");
    run_pack(pb, st);
    SAD_CHECK_LAUNCH("k_pack");
    return 0;
}

int sad_prep(const sad_dims* d, const int32_t* lengths, uint8_t* mask, int64_t msb, int64_t mst, int32_t guard,
             void* workspace, void* stream) {
    Shape s;
    if (check_dims(d, &s)) return -1;
    if (check_ptr(lengths, "lengths") || check_ptr(mask, "mask") || check_ptr(workspace, "workspace") ||
        check_aligned(workspace, 256, "workspace"))
        return -1;
    return run_prep(s, ws_layout(s), workspace, lengths, mask, msb, mst, guard, (hipStream_t)stream);
}

int sad_softemb_fwd(const sad_dims* d, const void* packed, const void* src, int32_t kind, int64_t sB, int64_t sT,
                    void* workspace, void* stream) {
    Shape s;
    if (check_dims(d, &s)) return -1;
    if (check_ptr(packed, "packed") || check_ptr(src, "src") || check_ptr(workspace, "workspace") ||
        check_aligned(workspace, 256, "workspace") || check_aligned(packed, 256, "packed"))
        return -1;
    return run_softemb_fwd(s, packed_layout(s), ws_layout(s), packed, workspace, src, kind, sB, sT, (hipStream_t)stream);
}

int sad_softemb_bwd(const sad_dims* d, const void* packed, void* dsrc, int32_t kind, int64_t sB, int64_t sT,
                    void* workspace, void* stream) {
    Shape s;
    if (check_dims(d, &s)) return -1;
    if (check_ptr(packed, "packed") || check_ptr(dsrc, "dsrc") || check_ptr(workspace, "workspace") ||
        check_aligned(workspace, 256, "workspace") || check_aligned(packed, 256, "packed"))
        return -1;
    return run_softemb_bwd(s, packed_layout(s), ws_layout(s), packed, workspace, dsrc, kind, sB, sT, (hipStream_t)stream);
}

}  // extern "C"

namespace {

// ---- stages shared by sad_forward / sad_backward and the single-stage entry points -------------------
struct Ctx {
    Shape s;
    PackedLayout pl;
    WsLayout wl;
    const void* packed;
    void* ws;
    hipStream_t st;
    bool upper = false;                                      // an RNN layer above the first (sad_rnn_layer)
};

// K2a + K3: [G | Kk] = X . Wcat^T + bcat
int run_inproj_fwd(const Ctx& c) {
    const Shape& s = c.s;
    const unsigned char* need = at<unsigned char>(c.ws, c.wl.off[SAD_WS_NEED]);
    const bf16raw* X = at<bf16raw>(c.ws, c.wl.off[SAD_WS_X]);
    float* G = at<float>(c.ws, c.wl.off[SAD_WS_G]);
    float* Kk = at<float>(c.ws, c.wl.off[SAD_WS_KK]);
    if (s.Ep == 320 && s.Ncat % 64 == 0) {
        // A-stationary + wave-private output turn (E = 289..320), over the compacted list of needed rows
        // (k_row_lists of this call), one workgroup per CU with an equal share of the (row tile, column tile) units
        constexpr size_t lds = 2 * 64 * 10 * 64 + 8 * 8192;
        allow_lds((const void*)k_inproj_fwd_rs3<10>, lds, 512);
        const long long units = (long long)ceil_div(s.M, 256) * (s.Ncat / 64);
        const int cus = device_cus();
        const int grid = units < cus ? (int)units : cus;
        hipLaunchKernelGGL((k_inproj_fwd_rs3<10>), dim3(grid), dim3(512), lds, c.st, s, X,
                           at<int>(c.ws, c.wl.off[SAD_WS_ROWS]), at<int>(c.ws, c.wl.off[SAD_WS_NROWS]),
                           at<bf16raw>(c.packed, c.pl.wcat), at<float>(c.packed, c.pl.bcat), G, Kk,
                           ga_preact(s) ? at<unsigned short>(c.ws, c.wl.off[SAD_WS_GA]) : (unsigned short*)nullptr);
    } else {
        typedef CfgSquare C;
        dim3 grid(ceil_div(s.Ncat, C::BN), ceil_div(s.M, C::BM));
        hipLaunchKernelGGL((k_inproj_fwd<C>), grid, dim3(C::NT), 0, c.st, s, X, need, at<bf16raw>(c.packed, c.pl.wcat),
                           at<float>(c.packed, c.pl.bcat), G, Kk);
    }
    SAD_CHECK_LAUNCH("k_inproj_fwd");
    return 0;
}

// K2b: the recurrence.  Persistent weight-stationary kernel (one launch for all T steps) when its grid
// is co-resident, else T dependent launches (both directions per launch).  The hand-off counters and
// the timeout words in SAD_WS_SYNC must be zero (k_prep of the same call, or run_zero_sync).
int run_lstm_fwd(const Ctx& c) {
    const Shape& s = c.s;
    const int* lens = at<int>(c.ws, c.wl.off[SAD_WS_LENS]);
    float* G = at<float>(c.ws, c.wl.off[SAD_WS_G]);
    float* Cs = at<float>(c.ws, c.wl.off[SAD_WS_CS]);
    bf16raw* Hs = at<bf16raw>(c.ws, c.wl.off[SAD_WS_HS]);
    const bf16raw* whh = at<bf16raw>(c.packed, c.pl.whh);
    int pgrid = 0;
    if (lstm_fwd_persist_fits(s, &pgrid)) {
        const PersistPlan plan = lstm_fwd_plan(s);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(plan.split ? k_lstm_fwd_persist2<true> : k_lstm_fwd_persist2<false>), dim3(pgrid), dim3(256),
                           plan.lds, c.st, s, lens, whh, G, Cs, Hs, at<bf16raw>(c.ws, c.wl.off[SAD_WS_HX]),
                           half_gates(s) ? at<unsigned short>(c.ws, c.wl.off[SAD_WS_GA]) : (unsigned short*)nullptr,
                           at<unsigned>(c.ws, c.wl.off[SAD_WS_SYNC]), g_spin_limit.load(std::memory_order_relaxed),
                           (c.upper || !g_withhold_upper_only.load(std::memory_order_relaxed)) ? g_withhold_fwd.load(std::memory_order_relaxed) : -1,
                           plan.bm, hostflag_dev(), ga_preact(s) ? 1 : 0);
    } else {
        for (int step = 0; step < s.T; ++step) {
            if (s.B >= 512) {
                typedef CfgStepBig C;
                dim3 grid(ceil_div(s.Gp, C::BN), ceil_div(s.B, C::BM), 2);
                hipLaunchKernelGGL((k_lstm_fwd<C>), grid, dim3(C::NT), 0, c.st, s, step, lens, whh, G, Cs, Hs, 0);
            } else {
                typedef CfgStepSmall C;
                dim3 grid(ceil_div(s.Gp, C::BN), ceil_div(s.B, C::BM), 2);
                hipLaunchKernelGGL((k_lstm_fwd<C>), grid, dim3(C::NT), 0, c.st, s, step, lens, whh, G, Cs, Hs, 0);
            }
        }
    }
    SAD_CHECK_LAUNCH("k_lstm_fwd");
    return 0;
}

// K4: q = hidden_bridges[0]([h_fwd | h_rev]), u = linear_q(q)
// (`top`: context whose HS holds the final hidden states — the last layer's for a multi-layer RNN)
int run_bridge_fwd(const Ctx& c, const Ctx& top) {
    const Shape& s = c.s;
    const int* lens = at<int>(c.ws, c.wl.off[SAD_WS_LENS]);
    const bf16raw* Hs = at<bf16raw>(top.ws, top.wl.off[SAD_WS_HS]);
    float* Q = at<float>(c.ws, c.wl.off[SAD_WS_Q]);
    float* U = at<float>(c.ws, c.wl.off[SAD_WS_U]);
    dim3 grid(ceil_div(s.Hp, LKS_BN), ceil_div(s.B, LKS_BM));
    allow_lds((const void*)k_linear_ks<1>, LKS_LDS, 64 * LKS_WAVES);
    allow_lds((const void*)k_linear_ks<0>, LKS_LDS, 64 * LKS_WAVES);
    hipLaunchKernelGGL(k_linear_ks<1>, grid, dim3(64 * LKS_WAVES), LKS_LDS, c.st, s, s.B, s.Hp, 2 * s.Hp, (const void*)Hs, 0, lens,
                       at<bf16raw>(c.packed, c.pl.whb0), at<float>(c.packed, c.pl.bhb0), Q, s.Hp);
    hipLaunchKernelGGL(k_linear_ks<0>, grid, dim3(64 * LKS_WAVES), LKS_LDS, c.st, s, s.B, s.Hp, s.Hp, (const void*)Q, s.Hp, lens,
                       at<bf16raw>(c.packed, c.pl.wq), at<float>(c.packed, c.pl.bq), U, s.Hp);
    SAD_CHECK_LAUNCH("k_linear_ks");
    return 0;
}

// K5: attention + pooling + head (+ CE)
int run_attn_fwd(const Ctx& c, const int64_t* labels, float* logits, float* weights, float* loss) {
    const Shape& s = c.s;
    const unsigned char* need = at<unsigned char>(c.ws, c.wl.off[SAD_WS_NEED]);
    const float* Kk = at<float>(c.ws, c.wl.off[SAD_WS_KK]);
    const float* U = at<float>(c.ws, c.wl.off[SAD_WS_U]);
    float* CTX = at<float>(c.ws, c.wl.off[SAD_WS_CTX]);
    const float* wa = at<float>(c.packed, c.pl.wa);
    const float* ba = at<float>(c.packed, c.pl.ba);
    const float* wo = at<float>(c.packed, c.pl.wo);
    const float* bo = at<float>(c.packed, c.pl.bo);
    const long long* lab = (const long long*)labels;
    // timeout word of the forward recurrence (behind the hand-off counters)
    const unsigned* err = at<unsigned>(c.ws, c.wl.off[SAD_WS_SYNC]) + (size_t)lp_err_line(s.B) * LP_SYNC_PITCH;
#define SAD_ATT_FWD(NV)                                                                                             \
    hipLaunchKernelGGL((k_attn_fwd<NV>), dim3(s.B), dim3(256), 0, c.st, s, Kk, U, need, wa, ba, wo, bo, lab, weights, CTX, \
                       logits, loss, err)
    if (s.Hp <= 256) SAD_ATT_FWD(1);
    else if (s.Hp <= 512) SAD_ATT_FWD(2);
    else if (s.Hp <= 1024) SAD_ATT_FWD(4);
    else SAD_ATT_FWD(8);
#undef SAD_ATT_FWD
    SAD_CHECK_LAUNCH("k_attn_fwd");
    return 0;
}

// K5': d logits -> dKk (key columns of DZ), du; zeroes the hand-off counters for the BPTT
int run_attn_bwd(const Ctx& c, const sad_bwd_args* a) {
    const Shape& s = c.s;
    const float* Kk = at<float>(c.ws, c.wl.off[SAD_WS_KK]);
    const float* U = at<float>(c.ws, c.wl.off[SAD_WS_U]);
    const float* CTX = at<float>(c.ws, c.wl.off[SAD_WS_CTX]);
    bf16raw* DZ = at<bf16raw>(c.ws, c.wl.off[SAD_WS_DZ]);
    float* DU = at<float>(c.ws, c.wl.off[SAD_WS_DU]);
    const float* wa = at<float>(c.packed, c.pl.wa);
    const float* wo = at<float>(c.packed, c.pl.wo);
    const long long* lab = (const long long*)a->labels;
#define SAD_ATT_BWD(NV)                                                                                              \
    hipLaunchKernelGGL((k_attn_bwd<NV>), dim3(s.B), dim3(256), 0, c.st, s, Kk, U, wa, wo, a->weights, CTX, a->logits, \
                       lab, a->dlogits, a->dloss, a->dloss_scale, a->dweights, DZ, DU,                                \
                       at<float>(c.ws, c.wl.off[SAD_WS_DL]), at<float>(c.ws, c.wl.off[SAD_WS_DWA]),                   \
                       at<float>(c.ws, c.wl.off[SAD_WS_DSSUM]), at<unsigned>(c.ws, c.wl.off[SAD_WS_SYNC]),            \
                       lp_err_line(s.B) * LP_SYNC_PITCH)
    if (s.Hp <= 256) SAD_ATT_BWD(1);
    else if (s.Hp <= 512) SAD_ATT_BWD(2);
    else if (s.Hp <= 1024) SAD_ATT_BWD(4);
    else SAD_ATT_BWD(8);
#undef SAD_ATT_BWD
    SAD_CHECK_LAUNCH("k_attn_bwd");
    return 0;
}

// K4': dq = du . W_q ; d[h_fwd | h_rev] = dq . W_hb0
// (`top`: context that receives d(final hidden) — the last layer's for a multi-layer RNN)
int run_bridge_bwd(const Ctx& c, const Ctx& top) {
    const Shape& s = c.s;
    const int* lens = at<int>(c.ws, c.wl.off[SAD_WS_LENS]);
    float* DU = at<float>(c.ws, c.wl.off[SAD_WS_DU]);
    float* DQ = at<float>(c.ws, c.wl.off[SAD_WS_DQ]);
    float* DHFIN = at<float>(top.ws, top.wl.off[SAD_WS_DHFIN]);
    dim3 g1(ceil_div(s.Hp, LKS_BN), ceil_div(s.B, LKS_BM));
    allow_lds((const void*)k_linear_ks<0>, LKS_LDS, 64 * LKS_WAVES);
    hipLaunchKernelGGL(k_linear_ks<0>, g1, dim3(64 * LKS_WAVES), LKS_LDS, c.st, s, s.B, s.Hp, s.Hp, (const void*)DU, s.Hp, lens,
                       at<bf16raw>(c.packed, c.pl.wqT), (const float*)nullptr, DQ, s.Hp);
    dim3 g2(ceil_div(2 * s.Hp, LKS_BN), ceil_div(s.B, LKS_BM));
    hipLaunchKernelGGL(k_linear_ks<0>, g2, dim3(64 * LKS_WAVES), LKS_LDS, c.st, s, s.B, 2 * s.Hp, s.Hp, (const void*)DQ, s.Hp, lens,
                       at<bf16raw>(c.packed, c.pl.whb0T), (const float*)nullptr, DHFIN, 2 * s.Hp);
    SAD_CHECK_LAUNCH("k_linear_ks(bwd)");
    return 0;
}

// K2': BPTT.  Persistent kernel (one launch) when its grid is co-resident, else T launches.  The
// hand-off counters must be zero (k_attn_bwd of the same call, or run_zero_sync); the timeout words
// behind them keep what the forward left.
// `dhs` (bf16 [T*B][ldh], column dir*H + unit, or NULL): gradient of every step's output when a layer above
// consumed it.
int run_lstm_bwd(const Ctx& c, const bf16raw* dhs = nullptr, int ldh = 0) {
    const Shape& s = c.s;
    const int* lens = at<int>(c.ws, c.wl.off[SAD_WS_LENS]);
    const float* G = at<float>(c.ws, c.wl.off[SAD_WS_G]);
    const float* Cs = at<float>(c.ws, c.wl.off[SAD_WS_CS]);
    const float* DHFIN = at<float>(c.ws, c.wl.off[SAD_WS_DHFIN]);
    float* DC = at<float>(c.ws, c.wl.off[SAD_WS_DC]);
    bf16raw* DZ = at<bf16raw>(c.ws, c.wl.off[SAD_WS_DZ]);
    const bf16raw* whhT = at<bf16raw>(c.packed, c.pl.whhT);
    int pgrid = 0;
    if (lstm_bwd_persist_fits(s, &pgrid)) {
        const PersistPlan plan = lstm_bwd_plan(s);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(plan.split ? k_lstm_bwd_persist<true> : k_lstm_bwd_persist<false>), dim3(pgrid), dim3(512),
                           plan.lds, c.st, s, lens, whhT, G, Cs, DHFIN, DZ, at<bf16raw>(c.ws, c.wl.off[SAD_WS_DZX]),
                           half_gates(s) ? at<unsigned short>(c.ws, c.wl.off[SAD_WS_GA]) : (const unsigned short*)nullptr,
                           at<unsigned>(c.ws, c.wl.off[SAD_WS_SYNC]), g_spin_limit.load(std::memory_order_relaxed),
                           (c.upper || !g_withhold_upper_only.load(std::memory_order_relaxed)) ? g_withhold_bwd.load(std::memory_order_relaxed) : -1,
                           dhs, ldh, plan.bm, hostflag_dev());
    } else {
        for (int step = 0; step < s.T; ++step) {
            typedef CfgStepBwd C;
            dim3 grid(ceil_div(s.Hp, C::BN), ceil_div(s.B, C::BM), 2);
            hipLaunchKernelGGL((k_lstm_bwd<C>), grid, dim3(C::NT), 0, c.st, s, step, lens, whhT, G, Cs, DHFIN, DC, DZ, dhs, ldh);
        }
    }
    SAD_CHECK_LAUNCH("k_lstm_bwd");
    return 0;
}

// K2a' + K3': dX = dZ . Wcat
int run_inproj_bwd(const Ctx& c) {
    const Shape& s = c.s;
    const unsigned char* need = at<unsigned char>(c.ws, c.wl.off[SAD_WS_NEED]);
    const bf16raw* DZ = at<bf16raw>(c.ws, c.wl.off[SAD_WS_DZ]);
    bf16raw* DX = at<bf16raw>(c.ws, c.wl.off[SAD_WS_DX]);
    float* accf = at<float>(c.ws, c.wl.off[SAD_WS_ACCF]);
    const bf16raw* wT = at<bf16raw>(c.packed, c.pl.wcatT);
    if (s.Ep == 320 && s.Ncat % 64 == 0 && c.wl.bytes[SAD_WS_K1SLABS] != 0) {
        // Ep = 320: the same GEMM shape as K1 (A = dZ rows, B = Wcat^T, K = Ncat), so it runs on K1's
        // pipelined kernels with a patched shape.
        Shape sp = s;
        sp.V = s.Ncat;
        sp.Vp = s.Ncat;
        constexpr size_t lds3 = 2 * 128 * 128 + 3 * 320 * 128;
        if (s.M < 16384) {       // few row tiles: stream-K fills the CUs (deterministic, no atomics)
            int* tiles = at<int>(c.ws, c.wl.off[SAD_WS_K1TILES]);
            float* slabs = at<float>(c.ws, c.wl.off[SAD_WS_K1SLABS]);
            const int cus = device_cus();
            const int nwg = cus > K1_MAX_WGS ? K1_MAX_WGS : cus;
            allow_lds((const void*)k_softemb_fwd_sk<SRC_BF16>, lds3, 1024);
            hipLaunchKernelGGL(k_softemb_fwd_sk<SRC_BF16>, dim3(nwg), dim3(1024), lds3, c.st, sp, (const void*)DZ,
                               (long long)s.B * s.Ncat, (long long)s.Ncat, need, tiles, wT, DX, slabs);
            hipLaunchKernelGGL(k_softemb_fwd_fix, dim3(ceil_div(s.M, 128), 8, 4), dim3(64), 0, c.st, sp, tiles, nwg, slabs, DX);
        } else {                 // three wave roles, one 128-row tile per workgroup (stream-K does not pay: 71 + 24 us vs 86)
            allow_lds((const void*)k_softemb_fwd_ws3<SRC_BF16>, lds3, 1024);
            hipLaunchKernelGGL(k_softemb_fwd_ws3<SRC_BF16>, dim3(ceil_div(s.M, 128)), dim3(1024), lds3, c.st, sp,
                               (const void*)DZ, (long long)s.B * s.Ncat, (long long)s.Ncat, need, wT, DX);
        }
    } else {
        const bool wide = s.Ep > 128;
        const int bm = 128, bn = wide ? 320 : 64;
        const int ksplit = split_k(ceil_div(s.M, bm) * ceil_div(s.Ep, bn), ceil_div(s.Ncat, 64));
        dim3 grid(ceil_div(s.M, bm), ceil_div(s.Ep, bn), ksplit);
        if (ksplit > 1) (void)hipMemsetAsync(accf, 0, (size_t)s.M * s.Ep * 4, c.st);
        if (wide)
            hipLaunchKernelGGL((k_inproj_bwd<CfgWideN>), grid, dim3(CfgWideN::NT), 0, c.st, s, DZ, need, wT, DX, accf, ksplit);
        else
            hipLaunchKernelGGL((k_inproj_bwd<CfgNarrowN>), grid, dim3(CfgNarrowN::NT), 0, c.st, s, DZ, need, wT, DX, accf, ksplit);
        if (ksplit > 1) launch_cvt(accf, DX, (long long)s.M * s.Ep, c.st);
    }
    SAD_CHECK_LAUNCH("k_inproj_bwd");
    return 0;
}

int run_apply_keep(const Ctx& c, bf16raw* x, const uint8_t* keep, float scale) {
    const long long chunks = (long long)c.s.M * (c.s.Ep / 8);
    int blocks = (int)((chunks + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_apply_keep, dim3(blocks), dim3(256), 0, c.st, c.s, x, keep, scale);
    SAD_CHECK_LAUNCH("k_apply_keep");
    return 0;
}

// Stage entry points run without the kernel that normally zeroes the hand-off counters in front of a
// recurrence: a 256-byte-multiple memset on the same stream does it (counters only, or + timeout words).
int run_zero_sync(const Ctx& c, bool with_timeout_words) {
    const size_t counters = (size_t)lp_err_line(c.s.B) * LP_SYNC_PITCH * 4;
    const size_t n = with_timeout_words ? c.wl.bytes[SAD_WS_SYNC] : counters;
    if (hipMemsetAsync(at<unsigned char>(c.ws, c.wl.off[SAD_WS_SYNC]), 0, n, c.st) != hipSuccess)
        return fail(-2, "hipMemsetAsync(SYNC) failed");
    return 0;
}

int make_ctx(const sad_dims* d, const void* packed, void* ws, void* stream, Ctx* c) {
    if (check_dims(d, &c->s)) return -1;
    if (check_ptr(packed, "packed") || check_ptr(ws, "workspace") || check_aligned(ws, 256, "workspace") ||
        check_aligned(packed, 256, "packed"))
        return -1;
    c->pl = packed_layout(c->s);
    c->wl = ws_layout(c->s);
    c->packed = packed;
    c->ws = ws;
    c->st = (hipStream_t)stream;
    return 0;
}

// Device address of g_noalloc_scratch on the current device, resolved on the first call per device.
unsigned char* g_noalloc_ptr[MAX_DEVICES];
int noalloc_scratch(unsigned char** out) {
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(g_cache_mu);
    if (g_noalloc_ptr[dev] == nullptr) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch));
        if (e != hipSuccess || p == nullptr)
            return fail(-2, "hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
        g_noalloc_ptr[dev] = (unsigned char*)p;
    }
    *out = g_noalloc_ptr[dev];
    return 0;
}

// Context of RNN layer i above the first: dims {B, T, 1, 2H, H, L, cell}, the caller's image and workspace.
int make_upper_ctx(const Ctx& c0, const sad_rnn_layer* upper, int n_upper, int i, Ctx* cu) {
    if (n_upper < 0 || n_upper > SAD_MAX_UPPER_LAYERS) return fail(-1, "n_upper=%d outside 0..%d", n_upper, SAD_MAX_UPPER_LAYERS);
    if (upper == nullptr) return fail(-1, "n_upper=%d but upper is NULL", n_upper);
    sad_dims du = {c0.s.B, c0.s.T, 1, 2 * c0.s.H, c0.s.H, c0.s.L, c0.s.cell};
    if (make_ctx(&du, upper[i].packed, upper[i].workspace, (void*)c0.st, cu)) return -1;
    cu->upper = true;
    return 0;
}

// Layers above the first: input = per-step outputs of the layer below, in-projection, recurrence.
int run_upper_fwd(const Ctx& c0, const sad_fwd_args* a) {
    const Ctx* below = &c0;
    Ctx cu[SAD_MAX_UPPER_LAYERS];
    for (int i = 0; i < a->n_upper; ++i) {
        int rc = make_upper_ctx(c0, a->upper, a->n_upper, i, &cu[i]);
        if (rc) return rc;
        const Ctx& c = cu[i];
        // lengths / need bits / row lists / zeroed counters of this layer's workspace (the mask was already
        // mutated by layer 0's k_prep; the guard is idempotent)
        if ((rc = run_prep(c.s, c.wl, c.ws, a->lengths, a->mask, a->mask_stride_b, a->mask_stride_t, a->guard, c.st)) != 0) return rc;
        const long long total = (long long)c.s.M * c.s.Ep;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(k_hs_to_x, dim3(blocks), dim3(256), 0, c.st, below->s, c.s, at<int>(c.ws, c.wl.off[SAD_WS_LENS]),
                           at<bf16raw>(below->ws, below->wl.off[SAD_WS_HS]), a->upper[i].drop_keep, a->upper[i].drop_scale,
                           at<bf16raw>(c.ws, c.wl.off[SAD_WS_X]));
        SAD_CHECK_LAUNCH("k_hs_to_x");
        if ((rc = run_inproj_fwd(c)) != 0) return rc;
        if ((rc = run_lstm_fwd(c)) != 0) return rc;
        below = &cu[i];
    }
    return run_bridge_fwd(c0, *below);
}

int check_upper(const sad_rnn_layer* upper, int n_upper) {
    if (n_upper < 0 || n_upper > SAD_MAX_UPPER_LAYERS) return fail(-1, "n_upper=%d outside 0..%d", n_upper, SAD_MAX_UPPER_LAYERS);
    if (n_upper > 0 && upper == nullptr) return fail(-1, "n_upper=%d but upper is NULL", n_upper);
    return 0;
}

int forward_impl(const Ctx& c_in, const sad_fwd_args* a) {
    if (check_ptr(a, "args")) return -1;
    if (check_upper(a->upper, a->n_upper)) return -1;
    const Ctx& c = c_in;
    const Shape& s = c.s;
    if (const unsigned code = take_hostflag())
        return fail(-4, "a hand-off wait inside the persistent %s of an EARLIER call on this device timed out (its workgroups "
                        "were not co-resident: another kernel on the device?); logits / loss / d(soft tokens) of that call "
                        "are NaN.  This call was not started; the next one will be",
                    code == 2 ? "BPTT" : "forward recurrence");
    if (check_ptr(a->src, "src") || check_ptr(a->lengths, "lengths") || check_ptr(a->mask, "mask") ||
        check_ptr(a->logits, "logits") || check_ptr(a->weights, "weights"))
        return -1;
    if (a->loss != nullptr && a->labels == nullptr) return fail(-1, "loss requested without labels");
    bf16raw* X = at<bf16raw>(c.ws, c.wl.off[SAD_WS_X]);
    // need bits, row lists, tile list (the same launch zeroes the hand-off counters and timeout words of the recurrence)
    int rc = run_prep(s, c.wl, c.ws, a->lengths, a->mask, a->mask_stride_b, a->mask_stride_t, a->guard, c.st);
    if (rc) return rc;
    // K1 / gather
    if (a->src_kind == SAD_SRC_IDS) {
        if (check_aligned(a->src, 8, "src")) return -1;
        const long long chunks = (long long)s.M * (s.Ep / 8);
        int blocks = (int)((chunks + 255) / 256);
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(k_embed_gather, dim3(blocks), dim3(256), 0, c.st, s, (const long long*)a->src,
                           (long long)a->src_stride_b, (long long)a->src_stride_t, at<bf16raw>(c.packed, c.pl.wemb), X);
        SAD_CHECK_LAUNCH("k_embed_gather");
    } else if (a->onehot_ids != nullptr) {
        // one-hot soft tokens with their (index, value) side channel: K1 forward is a scaled gather
        if (a->src_kind != SAD_SRC_F32 && a->src_kind != SAD_SRC_BF16)
            return fail(-1, "soft tokens need SAD_SRC_F32 or SAD_SRC_BF16, got %d", a->src_kind);
        if (check_ptr(a->onehot_val, "onehot_val") || check_aligned(a->onehot_ids, 8, "onehot_ids")) return -1;
        const long long chunks = (long long)s.M * (s.Ep / 8);
        int blocks = (int)((chunks + 255) / 256);
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(k_embed_gather_scaled, dim3(blocks), dim3(256), 0, c.st, s, (const long long*)a->onehot_ids,
                           (long long)a->onehot_stride_b, (long long)a->onehot_stride_t, a->onehot_val,
                           (long long)a->onehot_stride_b, (long long)a->onehot_stride_t, at<bf16raw>(c.packed, c.pl.wemb), X);
        SAD_CHECK_LAUNCH("k_embed_gather_scaled");
    } else {
        rc = run_softemb_fwd(s, c.pl, c.wl, c.packed, c.ws, a->src, a->src_kind, a->src_stride_b, a->src_stride_t, c.st);
        if (rc) return rc;
    }
    // dropout on the embedded tokens (classifier copy, training mode)
    if (a->drop_keep != nullptr && (rc = run_apply_keep(c, X, a->drop_keep, a->drop_scale)) != 0) return rc;
    if ((rc = run_inproj_fwd(c)) != 0) return rc;
    if ((rc = run_lstm_fwd(c)) != 0) return rc;
    if ((rc = a->n_upper ? run_upper_fwd(c, a) : run_bridge_fwd(c, c)) != 0) return rc;
    return run_attn_fwd(c, a->labels, a->logits, a->weights, a->loss);
}

int check_bwd_args(const sad_bwd_args* a) {
    if (check_ptr(a, "args")) return -1;
    if (check_ptr(a->weights, "weights") || check_ptr(a->logits, "logits")) return -1;
    if (a->dlogits == nullptr && a->labels == nullptr) return fail(-1, "need dlogits, or labels (+ dloss / dloss_scale)");
    return 0;
}

}  // namespace

extern "C" {

int sad_forward(const sad_dims* d, const void* packed, const sad_fwd_args* a, void* ws, void* stream) {
    Ctx c;
    if (make_ctx(d, packed, ws, stream, &c)) return -1;
    return forward_impl(c, a);
}

int sad_backward(const sad_dims* d, const void* packed, const sad_bwd_args* a, void* ws, void* stream) {
    Ctx c;
    if (make_ctx(d, packed, ws, stream, &c)) return -1;
    if (check_bwd_args(a) || check_upper(a->upper, a->n_upper)) return -1;
    int rc;
    if ((rc = run_attn_bwd(c, a)) != 0) return rc;
    if (a->n_upper == 0) {
        if ((rc = run_bridge_bwd(c, c)) != 0) return rc;
        if ((rc = run_lstm_bwd(c)) != 0) return rc;
    } else {
        // top-down through the layers above the first: d(final hidden) enters the last layer only; every
        // layer's d(input) is the per-step output gradient of the layer below it
        Ctx cu[SAD_MAX_UPPER_LAYERS];
        for (int i = 0; i < a->n_upper; ++i)
            if ((rc = make_upper_ctx(c, a->upper, a->n_upper, i, &cu[i])) != 0) return rc;
        const int top = a->n_upper - 1;
        if ((rc = run_bridge_bwd(c, cu[top])) != 0) return rc;
        const bf16raw* dhs = nullptr;
        for (int i = top; i >= 0; --i) {
            const Ctx& u = cu[i];
            // the key columns of this layer's dZ have no producer (no attention on upper layers): all zero
            if (hipMemsetAsync(at<unsigned char>(u.ws, u.wl.off[SAD_WS_DZ]), 0, u.wl.bytes[SAD_WS_DZ], u.st) != hipSuccess)
                return fail(-2, "hipMemsetAsync(DZ) failed");
            if (i != top && hipMemsetAsync(at<unsigned char>(u.ws, u.wl.off[SAD_WS_DHFIN]), 0, u.wl.bytes[SAD_WS_DHFIN], u.st) != hipSuccess)
                return fail(-2, "hipMemsetAsync(DHFIN) failed");
            if ((rc = run_zero_sync(u, false)) != 0) return rc;      // hand-off counters of this layer's BPTT
            if ((rc = run_lstm_bwd(u, dhs, u.s.Ep)) != 0) return rc;
            if ((rc = run_inproj_bwd(u)) != 0) return rc;
            if (a->upper[i].drop_keep != nullptr &&
                (rc = run_apply_keep(u, at<bf16raw>(u.ws, u.wl.off[SAD_WS_DX]), a->upper[i].drop_keep, a->upper[i].drop_scale)) != 0)
                return rc;
            dhs = at<bf16raw>(u.ws, u.wl.off[SAD_WS_DX]);
        }
        if (hipMemsetAsync(at<unsigned char>(c.ws, c.wl.off[SAD_WS_DHFIN]), 0, c.wl.bytes[SAD_WS_DHFIN], c.st) != hipSuccess)
            return fail(-2, "hipMemsetAsync(DHFIN) failed");
        if ((rc = run_lstm_bwd(c, dhs, cu[0].s.Ep)) != 0) return rc;
    }
    if ((rc = run_inproj_bwd(c)) != 0) return rc;
    // d(embedded tokens) in front of the dropout
    if (a->drop_keep != nullptr &&
        (rc = run_apply_keep(c, at<bf16raw>(c.ws, c.wl.off[SAD_WS_DX]), a->drop_keep, a->drop_scale)) != 0)
        return rc;
    // K1'
    if (a->dsrc != nullptr)
        return run_softemb_bwd(c.s, c.pl, c.wl, packed, ws, a->dsrc, a->dsrc_kind, a->dsrc_stride_b, a->dsrc_stride_t, c.st);
    return 0;
}

int sad_forward_noalloc(const sad_dims* d, const void* packed, const sad_fwd_args* a, void* stream) {
    Shape s;
    if (check_dims(d, &s)) return -1;
    const size_t need = ws_layout(s).total;
    if (need > (size_t)SAD_NOALLOC_MAX_BYTES)
        return fail(-3, "sad_forward_noalloc: B=%d T=%d V=%d E=%d H=%d needs a workspace of %zu bytes, the static scratch "
                        "holds %zu (use sad_forward with a caller-provided workspace)",
                    d->B, d->T, d->V, d->E, d->H, need, (size_t)SAD_NOALLOC_MAX_BYTES);
    unsigned char* scratch = nullptr;
    if (noalloc_scratch(&scratch)) return -2;
    Ctx c;
    if (make_ctx(d, packed, scratch, stream, &c)) return -1;
    return forward_impl(c, a);
}

// ---- single stages (unit tests, timing the kernel classes in isolation) ---------------------------
int sad_lstm_fwd(const sad_dims* d, const void* packed, void* ws, void* stream) {
    Ctx c;
    if (make_ctx(d, packed, ws, stream, &c)) return -1;
    int rc = run_zero_sync(c, true);
    return rc ? rc : run_lstm_fwd(c);
}

int sad_lstm_bwd(const sad_dims* d, const void* packed, void* ws, void* stream) {
    Ctx c;
    if (make_ctx(d, packed, ws, stream, &c)) return -1;
    int rc = run_zero_sync(c, false);
    return rc ? rc : run_lstm_bwd(c);
}

int sad_inproj_fwd(const sad_dims* d, const void* packed, void* ws, void* stream) {
    Ctx c;
    if (make_ctx(d, packed, ws, stream, &c)) return -1;
    return run_inproj_fwd(c);
}

int sad_inproj_bwd(const sad_dims* d, const void* packed, void* ws, void* stream) {
    Ctx c;
    if (make_ctx(d, packed, ws, stream, &c)) return -1;
    return run_inproj_bwd(c);
}

int sad_attnpool_fwd(const sad_dims* d, const void* packed, const int64_t* labels, float* logits, float* weights,
                     float* loss, void* ws, void* stream) {
    Ctx c;
    if (make_ctx(d, packed, ws, stream, &c)) return -1;
    if (check_ptr(logits, "logits") || check_ptr(weights, "weights")) return -1;
    if (loss != nullptr && labels == nullptr) return fail(-1, "loss requested without labels");
    return run_attn_fwd(c, labels, logits, weights, loss);
}

int sad_attnpool_bwd(const sad_dims* d, const void* packed, const sad_bwd_args* a, void* ws, void* stream) {
    Ctx c;
    if (make_ctx(d, packed, ws, stream, &c)) return -1;
    if (check_bwd_args(a)) return -1;
    return run_attn_bwd(c, a);
}

int sad_gumbel_st_step(const sad_gumbel_args* a, void* stream) {
    if (check_ptr(a, "args")) return -1;
    if (a->B < 1 || a->V < 1) return fail(-1, "B and V must be positive (B=%d V=%d)", a->B, a->V);
    if (check_ptr(a->logits, "logits") || check_ptr(a->noise, "noise") || check_ptr(a->s_row, "s_row") ||
        check_ptr(a->idx, "idx") || check_ptr(a->val, "val"))
        return -1;
    if (!(a->tau > 0.f)) return fail(-1, "tau must be positive");
    hipStream_t st = (hipStream_t)stream;
#define SAD_GUMBEL(KIND)                                                                                              \
    hipLaunchKernelGGL((k_gumbel_st_fwd<KIND>), dim3(a->B), dim3(256), 0, st, a->B, a->V, a->logits,                   \
                       (long long)a->logits_stride, a->noise, (long long)a->noise_stride, a->tau, a->alive, a->eos_id, \
                       a->s_row, (long long)a->s_stride_b, a->rnn_in, a->y_soft, (long long*)a->idx, a->val, a->lens,  \
                       a->alive_out)
    if (a->s_kind == SAD_SRC_F32) SAD_GUMBEL(SRC_F32);
    else if (a->s_kind == SAD_SRC_BF16) SAD_GUMBEL(SRC_BF16);
    else return fail(-1, "s_kind must be SAD_SRC_F32 or SAD_SRC_BF16, got %d", a->s_kind);
#undef SAD_GUMBEL
    SAD_CHECK_LAUNCH("k_gumbel_st_fwd");
    return 0;
}

int sad_gumbel_st_step_bwd(int32_t B, int32_t V, const void* d_s_row, int32_t s_kind, int64_t s_stride_b,
                           const float* d_rnn_in, const uint8_t* alive, const float* y_soft, float tau, float* dlogits,
                           int64_t dlogits_stride, void* stream) {
    if (B < 1 || V < 1) return fail(-1, "B and V must be positive (B=%d V=%d)", B, V);
    if (check_ptr(y_soft, "y_soft") || check_ptr(dlogits, "dlogits")) return -1;
    if (d_s_row == nullptr && d_rnn_in == nullptr) return fail(-1, "need d_s_row or d_rnn_in");
    hipStream_t st = (hipStream_t)stream;
    if (s_kind == SAD_SRC_F32)
        hipLaunchKernelGGL((k_gumbel_st_bwd<SRC_F32>), dim3(B), dim3(256), 0, st, B, V, d_s_row, (long long)s_stride_b, d_rnn_in,
                           alive, y_soft, tau, dlogits, (long long)dlogits_stride);
    else if (s_kind == SAD_SRC_BF16)
        hipLaunchKernelGGL((k_gumbel_st_bwd<SRC_BF16>), dim3(B), dim3(256), 0, st, B, V, d_s_row, (long long)s_stride_b, d_rnn_in,
                           alive, y_soft, tau, dlogits, (long long)dlogits_stride);
    else return fail(-1, "s_kind must be SAD_SRC_F32 or SAD_SRC_BF16, got %d", s_kind);
    SAD_CHECK_LAUNCH("k_gumbel_st_bwd");
    return 0;
}

int sad_soft_finalize(int32_t B, int32_t T, int32_t V, int32_t steps, int32_t over_max, int32_t pad_id, int32_t sos_id,
                      int32_t eos_id, void* S, int32_t s_kind, int64_t sT, int64_t sB, int64_t* ids, float* val,
                      const int32_t* lens_steps, const uint8_t* alive, int32_t* lens_out, uint8_t* mask, void* stream) {
    if (B < 1 || V < 1 || T < 1 || steps < 0) return fail(-1, "bad sizes (B=%d T=%d V=%d steps=%d)", B, T, V, steps);
    if (T < steps + 1) return fail(-1, "T=%d is too small for %d steps + [sos]", T, steps);
    if (B > 65535) return fail(-1, "B too large for this entry point");
    if (check_ptr(S, "S") || check_ptr(ids, "ids") || check_ptr(val, "val") || check_ptr(lens_steps, "lens_steps") ||
        check_ptr(alive, "alive") || check_ptr(lens_out, "lens_out") || check_ptr(mask, "mask"))
        return -1;
    hipStream_t st = (hipStream_t)stream;
    if (s_kind == SAD_SRC_F32)
        hipLaunchKernelGGL((k_soft_finalize<SRC_F32>), dim3(T, B), dim3(256), 0, st, B, T, V, steps, over_max, pad_id, sos_id, eos_id, S,
                           (long long)sT, (long long)sB, (long long*)ids, val, lens_steps, alive, lens_out, mask);
    else if (s_kind == SAD_SRC_BF16)
        hipLaunchKernelGGL((k_soft_finalize<SRC_BF16>), dim3(T, B), dim3(256), 0, st, B, T, V, steps, over_max, pad_id, sos_id, eos_id, S,
                           (long long)sT, (long long)sB, (long long*)ids, val, lens_steps, alive, lens_out, mask);
    else return fail(-1, "s_kind must be SAD_SRC_F32 or SAD_SRC_BF16, got %d", s_kind);
    SAD_CHECK_LAUNCH("k_soft_finalize");
    return 0;
}

int sad_debug_set(int32_t key, int64_t value) {
    switch (key) {
        case SAD_DBG_NO_PERSIST: g_no_persist.store(value != 0 ? 1 : 0); return 0;
        case SAD_DBG_SPIN_LIMIT: g_spin_limit.store(value <= 0 ? LP_SPIN_LIMIT : (unsigned)value); return 0;
        case SAD_DBG_WITHHOLD_FWD: g_withhold_fwd.store((int)value); return 0;
        case SAD_DBG_WITHHOLD_BWD: g_withhold_bwd.store((int)value); return 0;
        case SAD_DBG_WITHHOLD_UPPER_ONLY: g_withhold_upper_only.store(value != 0 ? 1 : 0); return 0;
        case SAD_DBG_WGRAD_ATOMICS: g_wgrad_atomics.store(value != 0 ? 1 : 0); return 0;
        default: return fail(-1, "unknown debug key %d", key);
    }
}

int sad_param_grads_accumulate(const sad_dims* d, const sad_param_grads* g, const void* src, int32_t src_kind,
                               int64_t sB, int64_t sT, int32_t pad_id, void* ws, void* stream) {
    Shape s;
    if (check_dims(d, &s)) return -1;
    if (check_ptr(g, "grads") || check_ptr(ws, "workspace") || check_aligned(ws, 256, "workspace")) return -1;
    hipStream_t st = (hipStream_t)stream;
    const WsLayout wl = ws_layout(s);
    const int* lens = at<int>(ws, wl.off[SAD_WS_LENS]);
    const unsigned char* need = at<unsigned char>(ws, wl.off[SAD_WS_NEED]);
    const bf16raw* X = at<bf16raw>(ws, wl.off[SAD_WS_X]);
    const bf16raw* Hs = at<bf16raw>(ws, wl.off[SAD_WS_HS]);
    const bf16raw* DZ = at<bf16raw>(ws, wl.off[SAD_WS_DZ]);
    const bf16raw* DX = at<bf16raw>(ws, wl.off[SAD_WS_DX]);
    const float* Q = at<float>(ws, wl.off[SAD_WS_Q]);
    const float* CTX = at<float>(ws, wl.off[SAD_WS_CTX]);
    const float* DU = at<float>(ws, wl.off[SAD_WS_DU]);
    const float* DQ = at<float>(ws, wl.off[SAD_WS_DQ]);
    const float* DL = at<float>(ws, wl.off[SAD_WS_DL]);
    const float* DWA = at<float>(ws, wl.off[SAD_WS_DWA]);
    const float* DSSUM = at<float>(ws, wl.off[SAD_WS_DSSUM]);

    // Compacted lists of the contributing token rows (padding rows are never touched): left in the
    // workspace by the sad_prep / sad_forward that wrote the need bits.
    const int* rows = at<int>(ws, wl.off[SAD_WS_ROWS]);
    const int* nrows = at<int>(ws, wl.off[SAD_WS_NROWS]);
    // Row splits: ~2 workgroups per CU (every split costs one atomic add per output element).  The
    // list lengths live on the device, so the split is sized for the worst case (every row active).
    auto row_split = [](int rows_max, int tiles) {
        int want = ceil_div(512, tiles < 1 ? 1 : tiles);
        const int stages = ceil_div(rows_max, 64);
        if (want > stages) want = stages;
        return want < 1 ? 1 : want;
    };

    // Second-generation kernels (wgrad2.hpp: 128-column tiles, three wave roles, LDS-direct loads) at the shipped sizes
    const bool gen2 = s.Ep == 320 && s.Hp % 256 == 0 && s.V >= 128;
    // row splits of a gen-2 launch: about one workgroup per CU
    auto split2 = [&](int tiles, int stage_bytes, size_t* lds) {
        const int cus = device_cus();
        int zs = cus / (tiles < 1 ? 1 : tiles);
        if (zs < 1) zs = 1;
        const int stages = ceil_div(s.M, 64);
        if (zs > stages) zs = stages;
        *lds = (size_t)stage_bytes;
        return zs;
    };

    // partial tiles go to the stream-K slab of K1 (idle here) and a second launch adds them up in fixed order; when a
    // launch's partial tiles do not fit the slab (very large M: many row splits) it falls back to fp32 atomics
    float* slab_all = wl.bytes[SAD_WS_K1SLABS] != 0 && !g_wgrad_atomics.load(std::memory_order_relaxed)
                          ? at<float>(ws, wl.off[SAD_WS_K1SLABS]) : nullptr;
    auto slab_for = [&](long long tiles, int zs, long long tile_floats) {
        return (size_t)tiles * zs * tile_floats * 4 <= wl.bytes[SAD_WS_K1SLABS] ? slab_all : (float*)nullptr;
    };
    W2Out wo = {g->embedding_weight, g->rnn_weight_ih[0], g->rnn_weight_ih[1], g->linear_emb_weight,
                CatBias{nullptr, nullptr, nullptr, nullptr, nullptr}, g->rnn_weight_hh[0], g->rnn_weight_hh[1]};

    // d embedding.weight
    if (g->embedding_weight != nullptr && gen2 && (src_kind == SAD_SRC_F32 || src_kind == SAD_SRC_BF16)) {
        if (check_ptr(src, "src") || check_aligned(src, src_kind == SAD_SRC_F32 ? 4 : 2, "src")) return -1;
        const int t1 = ceil_div(s.V, W2Emb::BN1), xt = ceil_div(t1, 8);
        size_t lds = 0;
        const int zs = split2(t1, W2Emb::LDS_BYTES, &lds);
        const dim3 grid(8 * xt, 1, zs);
        float* slab = slab_for(t1, zs, wgrad2_tile_floats<W2Emb>());
        if (src_kind == SAD_SRC_F32) {
            allow_lds((const void*)k_wgrad2_emb<SRC_F32>, lds, 1024);
            hipLaunchKernelGGL(k_wgrad2_emb<SRC_F32>, grid, dim3(1024), lds, st, s, src, (long long)sT, (long long)sB, DX, rows,
                               nrows, g->embedding_weight, xt, slab, t1);
        } else {
            allow_lds((const void*)k_wgrad2_emb<SRC_BF16>, lds, 1024);
            hipLaunchKernelGGL(k_wgrad2_emb<SRC_BF16>, grid, dim3(1024), lds, st, s, src, (long long)sT, (long long)sB, DX, rows,
                               nrows, g->embedding_weight, xt, slab, t1);
        }
        if (slab != nullptr)
            hipLaunchKernelGGL((k_wgrad2_reduce<W2Emb, W2_EMB>), dim3(t1, 8 * W2Emb::T1 * W2Emb::T2 * 64 / 256), dim3(256), 0, st, s,
                               slab, nrows, t1, zs, wo, 1, -1);
        SAD_CHECK_LAUNCH("k_wgrad2_emb");
    } else if (g->embedding_weight != nullptr) {
        if (check_ptr(src, "src")) return -1;
        if (src_kind == SAD_SRC_IDS) {
            if (check_aligned(src, 8, "src")) return -1;
            const long long chunks = (long long)s.M * (s.Ep / 4);
            int blocks = (int)((chunks + 255) / 256);
            if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(k_wgrad_emb_ids, dim3(blocks), dim3(256), 0, st, s, (const long long*)src, (long long)sB,
                               (long long)sT, DX, need, pad_id, g->embedding_weight);
        } else if (src_kind == SAD_SRC_F32 || src_kind == SAD_SRC_BF16) {
            if (check_aligned(src, src_kind == SAD_SRC_F32 ? 4 : 2, "src")) return -1;
            if (s.Ep == 320) {          // one n2 tile: S is read exactly once
                typedef WCfg<64, 320, 2, 4> W;                   // (128 vocabulary columns per tile: 66 VGPRs spilled, 460 us)
                const int t1 = ceil_div(s.V, W::BN1);
                const int xt = ceil_div(t1, 8);                  // vocabulary tiles per XCD (see the kernel)
                const dim3 grid(8 * xt, 1, row_split(s.M, t1));
                if (src_kind == SAD_SRC_F32)
                    launch_wg<W>(k_wgrad_emb<W, SRC_F32>, grid, st, s, src, (long long)sT, (long long)sB, DX, rows, nrows,
                                 g->embedding_weight, xt);
                else
                    launch_wg<W>(k_wgrad_emb<W, SRC_BF16>, grid, st, s, src, (long long)sT, (long long)sB, DX, rows, nrows,
                                 g->embedding_weight, xt);
            } else {
                typedef WCfg<128, 64, 4, 1> W;
                const int t1 = ceil_div(s.V, W::BN1), t2 = ceil_div(s.Ep, W::BN2);
                const dim3 grid(t1, t2, row_split(s.M, t1 * t2));
                if (src_kind == SAD_SRC_F32)
                    launch_wg<W>(k_wgrad_emb<W, SRC_F32>, grid, st, s, src, (long long)sT, (long long)sB, DX, rows, nrows,
                                 g->embedding_weight, 0);
                else
                    launch_wg<W>(k_wgrad_emb<W, SRC_BF16>, grid, st, s, src, (long long)sT, (long long)sB, DX, rows, nrows,
                                 g->embedding_weight, 0);
            }
        } else {
            return fail(-1, "unknown src_kind %d", src_kind);
        }
        SAD_CHECK_LAUNCH("k_wgrad_emb");
    }
    // d rnn.weight_ih (both directions), d linear_emb.weight; with a padding column in X (E < Ep) also the LSTM
    // biases and linear_emb.bias (column sums of DZ: see k_wgrad_cat)
    const bool want_bias = g->rnn_bias_ih[0] != nullptr || g->rnn_bias_ih[1] != nullptr || g->rnn_bias_hh[0] != nullptr ||
                           g->rnn_bias_hh[1] != nullptr || g->linear_emb_bias != nullptr;
    bool bias_done = false;
    if (g->rnn_weight_ih[0] != nullptr || g->rnn_weight_ih[1] != nullptr || g->linear_emb_weight != nullptr) {
        CatBias cb = {nullptr, nullptr, nullptr, nullptr, nullptr};
        if (want_bias && s.E < s.Ep) {
            cb = CatBias{g->rnn_bias_ih[0], g->rnn_bias_hh[0], g->rnn_bias_ih[1], g->rnn_bias_hh[1], g->linear_emb_bias};
            bias_done = true;
        }
        if (gen2) {
            const int t1 = s.Ncat / W2Cat::BN1;
            size_t lds = 0;
            const int zs = split2(t1, W2Cat::LDS_BYTES, &lds);
            allow_lds((const void*)k_wgrad2_cat, lds, 1024);
            float* slab = slab_for(t1, zs, wgrad2_tile_floats<W2Cat>());
            hipLaunchKernelGGL(k_wgrad2_cat, dim3(t1, 1, zs), dim3(1024), lds, st, s, DZ, X, rows, nrows, g->rnn_weight_ih[0],
                               g->rnn_weight_ih[1], g->linear_emb_weight, cb, slab);
            if (slab != nullptr) {
                wo.cb = cb;
                const bool ones = cb.bih0 != nullptr || cb.bhh0 != nullptr || cb.bih1 != nullptr || cb.bhh1 != nullptr || cb.blin != nullptr;
                hipLaunchKernelGGL((k_wgrad2_reduce<W2Cat, W2_CAT>), dim3(t1, 8 * W2Cat::T1 * W2Cat::T2 * 64 / 256), dim3(256), 0, st,
                                   s, slab, nrows, t1, zs, wo, 1, ones && s.E < s.Ep ? s.E : -1);
            }
        } else if (s.Ep == 320) {
            typedef WCfg<64, 320, 2, 4> W;
            const int t1 = ceil_div(s.Ncat, W::BN1);
            launch_wg<W>(k_wgrad_cat<W>, dim3(t1, 1, row_split(s.M, t1)), st, s, DZ, X, rows, nrows, g->rnn_weight_ih[0],
                         g->rnn_weight_ih[1], g->linear_emb_weight, cb);
        } else {
            typedef WCfg<128, 64, 4, 1> W;
            const int t1 = ceil_div(s.Ncat, W::BN1), t2 = ceil_div(s.Ep, W::BN2);
            launch_wg<W>(k_wgrad_cat<W>, dim3(t1, t2, row_split(s.M, t1 * t2)), st, s, DZ, X, rows, nrows,
                         g->rnn_weight_ih[0], g->rnn_weight_ih[1], g->linear_emb_weight, cb);
        }
        SAD_CHECK_LAUNCH("k_wgrad_cat");
    }
    // d rnn.weight_hh (both directions)
    if ((g->rnn_weight_hh[0] != nullptr || g->rnn_weight_hh[1] != nullptr) && gen2) {
        const int t1 = s.Gp / W2Hh::BN1, t2 = s.Hp / W2Hh::BN2;
        size_t lds = 0;
        const int zs = split2(2 * t1 * t2, W2Hh::LDS_BYTES, &lds);
        allow_lds((const void*)k_wgrad2_hh, lds, 1024);
        float* slab = slab_for(2 * t1 * t2, zs, wgrad2_tile_floats<W2Hh>());
        hipLaunchKernelGGL(k_wgrad2_hh, dim3(t1, t2, 2 * zs), dim3(1024), lds, st, s, DZ, Hs, rows, nrows, g->rnn_weight_hh[0],
                           g->rnn_weight_hh[1], slab);
        if (slab != nullptr)
            hipLaunchKernelGGL((k_wgrad2_reduce<W2Hh, W2_HH>), dim3(2 * t1 * t2, 8 * W2Hh::T1 * W2Hh::T2 * 64 / 256), dim3(256), 0, st,
                               s, slab, nrows, 2 * t1 * t2, zs, wo, t2, -1);
        SAD_CHECK_LAUNCH("k_wgrad2_hh");
    } else if (g->rnn_weight_hh[0] != nullptr || g->rnn_weight_hh[1] != nullptr) {
        typedef WCfg<128, 256, 2, 4> W;
        const int t1 = ceil_div(s.Gp, W::BN1), t2 = ceil_div(s.Hp, W::BN2);
        launch_wg<W>(k_wgrad_hh<W>, dim3(t1, t2, 2 * row_split(s.M, 2 * t1 * t2)), st, s, DZ, Hs, rows, nrows,
                     g->rnn_weight_hh[0], g->rnn_weight_hh[1]);
        SAD_CHECK_LAUNCH("k_wgrad_hh");
    }
    // LSTM biases and linear_emb.bias: column sums of DZ (when they did not ride along above)
    if (want_bias && !bias_done) {
        const int cb = ceil_div(s.Ncat, 1024);
        int zs = ceil_div(512, cb);                         // row splits (whole 64-row pieces of the compacted list)
        if (zs > ceil_div(s.M, 64)) zs = ceil_div(s.M, 64);
        hipLaunchKernelGGL(k_colsum_dz, dim3(cb, zs), dim3(128), 0, st, s, DZ, rows, nrows, g->rnn_bias_ih[0],
                           g->rnn_bias_hh[0], g->rnn_bias_ih[1], g->rnn_bias_hh[1], g->linear_emb_bias);
        SAD_CHECK_LAUNCH("k_colsum_dz");
    }
    // per-sentence dense layers
    {
        typedef WCfg<64, 64, 2, 2> W;
        if (g->hidden_bridge0_weight != nullptr) {
            const int t1 = ceil_div(s.Hp, W::BN1), t2 = ceil_div(2 * s.Hp, W::BN2);
            const bf16raw* Hfin = Hs;
            if (g->final_hidden_workspace != nullptr)       // the last RNN layer's workspace (dims of a sad_rnn_layer)
                Hfin = at<bf16raw>(g->final_hidden_workspace,
                                   ws_layout(make_shape(s.B, s.T, 1, 2 * s.H, s.H, s.L, s.cell)).off[SAD_WS_HS]);
            launch_wg<W>(k_wgrad_dense<W, 1>, dim3(t1, t2, row_split(s.B, t1 * t2)), st, s, DQ, (const void*)Hfin, 2 * s.Hp,
                         lens, g->hidden_bridge0_weight);
        }
        if (g->linear_q_weight != nullptr) {
            const int t1 = ceil_div(s.Hp, W::BN1), t2 = ceil_div(s.Hp, W::BN2);
            launch_wg<W>(k_wgrad_dense<W, 0>, dim3(t1, t2, row_split(s.B, t1 * t2)), st, s, DU, (const void*)Q, s.Hp, lens,
                         g->linear_q_weight);
        }
        SAD_CHECK_LAUNCH("k_wgrad_dense");
    }
    // small reductions over sentences: one launch (short serial sums: the kernel is a latency chain); skipped when
    // none of its six outputs is wanted (e.g. the embedding-only stage of a data-parallel backward)
    if (g->hidden_bridge0_bias != nullptr || g->linear_q_bias != nullptr || g->linear_attn_weight != nullptr ||
        g->linear_attn_bias != nullptr || g->output_layer_bias != nullptr || g->output_layer_weight != nullptr) {
        ColsumJobs jobs;
        const float* srcs[5] = {DQ, DU, DWA, DSSUM, DL};
        float* dsts[5] = {g->hidden_bridge0_bias, g->linear_q_bias, g->linear_attn_weight, g->linear_attn_bias, g->output_layer_bias};
        const int cols[5] = {s.H, s.H, s.H, 1, s.L}, lds[5] = {s.Hp, s.Hp, s.Hp, 1, s.L};
        int maxc = s.H;
        for (int i = 0; i < 5; ++i) {
            jobs.src[i] = srcs[i];
            jobs.dst[i] = dsts[i];
            jobs.cols[i] = cols[i];
            jobs.ld[i] = lds[i];
            if (cols[i] > maxc) maxc = cols[i];
        }
        jobs.n = 5;
        int zs = ceil_div(s.B, 16);
        if (zs > 256) zs = 256;
        const int chunk = ceil_div(s.B, zs);
        hipLaunchKernelGGL(k_colsum_small, dim3(ceil_div(maxc, 256), ceil_div(s.B, chunk), 6), dim3(256), 0, st, s, jobs, chunk,
                           DL, CTX, g->output_layer_weight);
        SAD_CHECK_LAUNCH("k_colsum_small");
    }
    return 0;
}

}  // extern "C"
