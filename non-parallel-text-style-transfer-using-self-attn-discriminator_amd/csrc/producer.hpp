// Fused producer of the discriminator's soft tokens: the straight-through
// Gumbel-softmax sample of one decode step written STRAIGHT into its row of the time-major soft-token
// tensor S[T][B][V] (bf16 or fp32), plus an exact (index, value) side channel per token, the alive /
// length bookkeeping of soft_decode, and the [sos] / [eos] / [pad] rows and the pad+sos+eos mask.
//
// Reference lines (relative to the reference repo):
//   transfer/models.py:531      rnn_in = F.gumbel_softmax(logits, tau=1.0, hard=True)
//   transfer/models.py:533-543  argmax id, output_list.append(rnn_in * mask), outputs_len += mask,
//                               mask = (id != eos) * mask
//   transfer/models.py:546-571  [sos] row in front, optional [eos] row, rows t >= len := one-hot [pad]
//   transfer/models.py:1180-1183 mask = arange(T) >= len - 1 ; mask[:, 0] = True
// What the reference materialises per step (a dense fp32 [B,V] sample, its masked copy, the cat of all
// steps, the padded clone) is written once, in the dtype the discriminator consumes; the side channel
// lets K1 forward gather  X[t,b] = val * W_emb[idx]  instead of contracting a one-hot row with the whole
// embedding matrix.  d(soft tokens) stays dense (K1'): the straight-through estimator back-propagates
// through y_soft = softmax((logits + noise) / tau), which is dense.
#pragma once
#include "kernels.hpp"

namespace sad {

struct ArgMax {
    float v;
    int i;
};
__device__ __forceinline__ ArgMax argmax_combine(ArgMax a, ArgMax b) {
    // larger value wins; on equal values the smaller index (first occurrence, as torch.max on a row)
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ArgMax wave_argmax(ArgMax a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax b;
        b.v = __shfl_xor(a.v, o, 64);
        b.i = __shfl_xor(a.i, o, 64);
        a = argmax_combine(a, b);
    }
    return a;
}

template <int KIND>
__device__ __forceinline__ void store_soft(void* row, long long v, float x) {
    if constexpr (KIND == SRC_F32) ((float*)row)[v] = x;
    else ((bf16raw*)row)[v] = f32_to_bf16(x);
}
template <int KIND>
__device__ __forceinline__ float load_soft(const void* row, long long v) {
    if constexpr (KIND == SRC_F32) return ((const float*)row)[v];
    else return bf16_to_f32(((const bf16raw*)row)[v]);
}

// One decode step.  One 256-thread workgroup per sentence, three passes over its vocabulary row
// (arg-max of z = (logits + noise) / tau; sum of exp; outputs).  `alive` (nullable = every sentence
// alive) is soft_decode's `mask` BEFORE this step; the row written into S is the sample times alive.
template <int KIND>
__global__ __launch_bounds__(256) void k_gumbel_st_fwd(int B, int V, const float* __restrict__ logits, long long ldl,
                                                      const float* __restrict__ noise, long long ldn, float tau,
                                                      const unsigned char* alive, int eos_id,
                                                      void* __restrict__ S_row, long long ldS, float* __restrict__ rnn_in,
                                                      float* __restrict__ y_soft, long long* __restrict__ idx_out,
                                                      float* __restrict__ val_out, int* __restrict__ lens,
                                                      unsigned char* alive_out) {
    // `alive_out` may be the same buffer as `alive` (no __restrict__ on either): every thread reads its sentence's
    // byte here, in front of the first barrier, and thread 0 writes the new one behind the last.
    const bool live = alive == nullptr || alive[blockIdx.x] != 0;
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    __shared__ float red_s[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lg = logits + (long long)b * ldl;
    const float* nz = noise + (long long)b * ldn;
    ArgMax am{-INFINITY, 0x7fffffff};
    for (int v = tid; v < V; v += 256) {
        const float z = (lg[v] + nz[v]) / tau;
        if (z > am.v) am = ArgMax{z, v};          // ascending v inside a thread: keeps the first occurrence
    }
    am = wave_argmax(am);
    if (lane == 0) { red_v[wave] = am.v; red_i[wave] = am.i; }
    __syncthreads();
    am = ArgMax{red_v[0], red_i[0]};
#pragma unroll
    for (int w = 1; w < 4; ++w) am = argmax_combine(am, ArgMax{red_v[w], red_i[w]});
    const float zmax = am.v;
    const int idx = am.i;
    float part = 0.f;
    for (int v = tid; v < V; v += 256) part += expf((lg[v] + nz[v]) / tau - zmax);
    part = wave_sum(part);
    if (lane == 0) red_s[wave] = part;
    __syncthreads();
    const float sum = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
    const float y_idx = expf((lg[idx] + nz[idx]) / tau - zmax) / sum;
    const float val = (1.0f - y_idx) + y_idx;      // y_hard - y_soft + y_soft at the arg-max; exactly 0 elsewhere
    void* srow = (char*)S_row + (long long)b * ldS * (KIND == SRC_F32 ? 4 : 2);
    for (int v = tid; v < V; v += 256) {
        const float y = expf((lg[v] + nz[v]) / tau - zmax) / sum;
        if (y_soft != nullptr) y_soft[(long long)b * V + v] = y;
        const float r = v == idx ? val : 0.f;
        if (rnn_in != nullptr) rnn_in[(long long)b * V + v] = r;
        store_soft<KIND>(srow, v, live ? r : 0.f);
    }
    if (tid == 0) {
        idx_out[b] = idx;
        val_out[b] = live ? val : 0.f;
        if (lens != nullptr && live) lens[b] += 1;
        if (alive_out != nullptr) alive_out[b] = (live && idx != eos_id) ? 1 : 0;
    }
}

// Backward of one step: g = alive * dS_row (+ d rnn_in);  dlogits = y * (g - <g, y>) / tau.
template <int KIND>
__global__ __launch_bounds__(256) void k_gumbel_st_bwd(int B, int V, const void* __restrict__ gS_row, long long ldS,
                                                      const float* __restrict__ g_rnn, const unsigned char* __restrict__ alive,
                                                      const float* __restrict__ y_soft, float tau,
                                                      float* __restrict__ dlogits, long long ldd) {
    __shared__ float red_s[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool live = gS_row != nullptr && (alive == nullptr || alive[b] != 0);
    const void* grow = gS_row == nullptr ? nullptr : (const char*)gS_row + (long long)b * ldS * (KIND == SRC_F32 ? 4 : 2);
    const float* y = y_soft + (long long)b * V;
    float part = 0.f;
    for (int v = tid; v < V; v += 256) {
        float g = live ? load_soft<KIND>(grow, v) : 0.f;
        if (g_rnn != nullptr) g += g_rnn[(long long)b * V + v];
        part += g * y[v];
    }
    part = wave_sum(part);
    if (lane == 0) red_s[wave] = part;
    __syncthreads();
    const float dot = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
    for (int v = tid; v < V; v += 256) {
        float g = live ? load_soft<KIND>(grow, v) : 0.f;
        if (g_rnn != nullptr) g += g_rnn[(long long)b * V + v];
        dlogits[(long long)b * ldd + v] = y[v] * (g - dot) / tau;
    }
}

// After the last step: [sos] row, optional [eos] row for sentences still alive at the length limit, one-hot
// [pad] for t >= len, side channel of those rows, final lengths, pad+sos+eos mask.  Grid (T, B); a
// workgroup whose row keeps its sampled tokens only fills in the mask byte.
//   steps     decode steps taken (rows 1..steps of S hold samples)
//   over_max  the decode loop stopped at the length limit (transfer/models.py:556)
//   lens      in: tokens sampled per sentence (outputs_len before ":554 += 1"); out: final length
template <int KIND>
__global__ __launch_bounds__(256) void k_soft_finalize(int B, int T, int V, int steps, int over_max, int pad_id, int sos_id,
                                                      int eos_id, void* __restrict__ S, long long sT, long long sB,
                                                      long long* __restrict__ ids, float* __restrict__ val,
                                                      const int* __restrict__ lens_steps, const unsigned char* __restrict__ alive,
                                                      int* __restrict__ lens_out, unsigned char* __restrict__ mask) {
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const bool extra = over_max != 0 && alive[b] != 0;                 // gets an [eos] row appended (:559-561)
    const int len = lens_steps[b] + 1 + (extra ? 1 : 0);
    if (tid == 0) {
        if (t == 0) lens_out[b] = len;
        mask[(long long)b * T + t] = (t == 0 || t >= len - 1) ? 1 : 0;  // transfer/models.py:1180-1183
    }
    int hot;
    if (t == 0) hot = sos_id;
    else if (extra && t == steps + 1) hot = eos_id;
    else if (t >= len) hot = pad_id;
    else return;                                                       // a sampled row: keep
    void* row = (char*)S + ((long long)t * sT + (long long)b * sB) * (KIND == SRC_F32 ? 4 : 2);
    for (int v = tid; v < V; v += 256) store_soft<KIND>(row, v, v == hot ? 1.f : 0.f);
    if (tid == 0) {
        ids[(long long)t * B + b] = hot;
        val[(long long)t * B + b] = 1.f;
    }
}

// K1 forward from the side channel: X[m] = bf16( round_S(val) * W_emb[idx] ) — bit-identical to what the
// GEMM computes on a row whose only non-zero is `val` at `idx` (every other product is an exact zero).
// The GEMM rounds S to bf16 on its way into LDS (fp32 and bf16 S alike), so val is rounded to bf16 here too.
__global__ void k_embed_gather_scaled(Shape s, const long long* __restrict__ ids, long long ib, long long it,
                                      const float* __restrict__ val, long long vb, long long vt,
                                      const bf16raw* __restrict__ wemb, bf16raw* __restrict__ X) {
    const int cpr = s.Ep / 8;
    const long long total = (long long)s.M * cpr;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(c / cpr), kc = (int)(c - (long long)m * cpr) * 8;
        const int t = m / s.B, b = m - t * s.B;
        const long long id = ids[(long long)b * ib + (long long)t * it];
        const float x = bf16_to_f32(f32_to_bf16(val[(long long)b * vb + (long long)t * vt]));
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (id >= 0 && id < s.V) {
            const u32x4 w = *(const u32x4*)(wemb + id * s.Ep + kc);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                v[i] = pack_bf16x2(x * bf16_to_f32((bf16raw)(w[i] & 0xffffu)), x * bf16_to_f32((bf16raw)(w[i] >> 16)));
        }
        *(u32x4*)(X + (long long)m * s.Ep + kc) = v;
    }
}

}  // namespace sad
