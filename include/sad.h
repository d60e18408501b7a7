/* libsad — C-ABI of the MI355X-native self-attention style discriminator (hot path only).
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  The
 * reference has no FFI for this path (it is pure PyTorch); each entry point below cites the
 * reference Python lines whose arithmetic it replaces (paths relative to the reference repo).
 * The host-side mirror of the reference interface (`SelfAttnRNNClassifier`, same constructor /
 * forward / state_dict) lives in the Python package and binds these symbols with ctypes; see
 * INTEGRATION.md for the binding a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the current HIP device unless stated otherwise;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued
 *     asynchronously on it, nothing here synchronises or allocates;
 *   - functions return 0 on success, <0 on error; `sad_last_error()` (thread-local, host string)
 *     describes the last failure of the calling thread (-1 bad argument, -2 HIP error, -3 shape beyond
 *     sad_forward_noalloc's static scratch, -4 an EARLIER call's persistent recurrence timed out: see sad_forward);
 *   - token rows are addressed through explicit element strides, so both the transfer loop's
 *     time-major soft tokens ([T,B,V] storage viewed as [B,T,V]) and batch-major tensors work
 *     without copies; the vocabulary axis must be contiguous.
 */
#ifndef SAD_H_
#define SAD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAD_VERSION 300 /* 0.3.0 */

/* src_kind / dsrc_kind */
#define SAD_SRC_IDS 0  /* int64 token ids [B,T]                  (reference: self.embedding(src)) */
#define SAD_SRC_F32 1  /* fp32 soft tokens [B,T,V]               (reference: src @ embedding.weight) */
#define SAD_SRC_BF16 2 /* bf16 soft tokens [B,T,V] */

typedef struct sad_dims {
    int32_t B; /* sentences in this call */
    int32_t T; /* max length incl. sos/eos (<= 1024) */
    int32_t V; /* vocabulary */
    int32_t E; /* embedding size */
    int32_t H; /* LSTM hidden size per direction */
    int32_t L; /* labels (<= 64) */
    int32_t cell; /* SAD_CELL_LSTM / SAD_CELL_GRU (reference classifier/models.py:102-110: getattr(nn, cell)) */
} sad_dims;

/* Recurrent cell.  The GRU runs in the LSTM's four-slot gate layout: the caller hands in
 *   weight_ih = [W_ir; W_iz; W_in; 0]   bias_ih = [b_ir; b_iz; b_in; 0]
 *   weight_hh = [W_hr; W_hz; 0; W_hn]   bias_hh = [b_hr; b_hz; 0; b_hn]        (each [4H, .])
 * so that slot 2 is the input half and slot 3 the hidden half of the candidate gate
 *   n = tanh(slot2 + r * slot3),  h' = (1 - z) * n + z * h      (torch.nn.GRU);
 * gradients come back in the same four-slot layout (the zero blocks receive values to be dropped). */
#define SAD_CELL_LSTM 0
#define SAD_CELL_GRU 1

/* The reference module's fp32 parameters (state_dict layout, reference transfer/models.py:757-829;
 * output_bridge.* and hidden_bridges.1.* never reach the logits and are not needed). Index 0 =
 * forward direction (`*_l0`), 1 = reverse (`*_l0_reverse`). */
typedef struct sad_weights {
    const float* embedding_weight;       /* [V,E]  */
    const float* rnn_weight_ih[2];       /* [4H,E] */
    const float* rnn_weight_hh[2];       /* [4H,H] */
    const float* rnn_bias_ih[2];         /* [4H]   */
    const float* rnn_bias_hh[2];         /* [4H]   */
    const float* hidden_bridge0_weight;  /* [H,2H] */
    const float* hidden_bridge0_bias;    /* [H]    */
    const float* linear_emb_weight;      /* [H,E]  */
    const float* linear_emb_bias;        /* [H]    */
    const float* linear_q_weight;        /* [H,H]  */
    const float* linear_q_bias;          /* [H]    */
    const float* linear_attn_weight;     /* [1,H]  */
    const float* linear_attn_bias;       /* [1]    */
    const float* output_layer_weight;    /* [L,H]  */
    const float* output_layer_bias;      /* [L]    */
} sad_weights;

/* One RNN layer ABOVE the first (enc_num_layers > 1; reference classifier/models.py:104-110 -> nn.LSTM / nn.GRU
 * with num_layers: layer l consumes the per-step outputs [h_fwd | h_rev] of layer l-1).  Each such layer runs
 * with its own weight image and workspace, both built by the caller with the dims
 *     {B, T, V = 1, E = 2H, H, L, cell}
 * where only the rnn_* tensors of sad_weights are this layer's (`weight_ih_l<l>` [4H, 2H], ...) and every other
 * tensor is zero.  Every layer runs the same recurrence kernels as a one-layer call (persistent when co-resident);
 * the layers below the last receive the gradient of their per-step outputs from the layer above. */
#define SAD_MAX_UPPER_LAYERS 7
typedef struct sad_rnn_layer {
    const void* packed;       /* sad_pack_weights image of this layer */
    void* workspace;          /* sad_workspace_bytes of this layer's dims; 256-byte aligned */
    const uint8_t* drop_keep; /* [T,B,2H] bytes, 1 = keep, or NULL: inverted dropout on this layer's INPUT (nn.LSTM's
                                 inter-layer dropout in training mode, classifier/models.py:108) */
    float drop_scale;
} sad_rnn_layer;

typedef struct sad_fwd_args {
    const void* src;          /* ids / soft tokens, see src_kind */
    int32_t src_kind;
    int64_t src_stride_b;     /* element strides of the batch and time axes */
    int64_t src_stride_t;
    const int32_t* lengths;   /* [B], sentence lengths incl. sos/eos */
    uint8_t* mask;            /* [B,T] bool bytes, 1 = pad/sos/eos; MUTATED in place when guard != 0 */
    int64_t mask_stride_b;
    int64_t mask_stride_t;
    int32_t guard;            /* 1: transfer copy (all-masked rows get position 1 unmasked,
                                 reference transfer/models.py:865-867); 0: classifier copy */
    const int64_t* labels;    /* [B] or NULL; with `loss` gives per-sentence CE */
    float* logits;            /* out [B,L] */
    float* weights;           /* out [B,T] attention weights */
    float* loss;              /* out [B] or NULL */
    const uint8_t* drop_keep; /* [T,B,E] bytes (time-major), 1 = keep, or NULL: inverted dropout on the embedded
                                 tokens, X *= keep * drop_scale (classifier copy in training mode, reference
                                 classifier/models.py:217; the caller draws the mask) */
    float drop_scale;         /* 1 / (1 - p) */
    /* Optional one-hot side channel of soft tokens (what sad_gumbel_st_step /
     * sad_soft_finalize emit): when `src` holds straight-through one-hot rows (reference
     * transfer/models.py:531, hard=True) the caller may pass, per token, the index of the non-zero entry
     * and its value; K1 forward then gathers X = value * W_emb[index] instead of contracting the row with the
     * whole embedding matrix — bit-identical to the GEMM on such rows.  `src` is still required (parameter
     * gradients read it) and d(soft tokens) stays dense.  NULL = no side channel. */
    const int64_t* onehot_ids;  /* [B,T] through the strides below */
    const float* onehot_val;    /* [B,T] */
    int64_t onehot_stride_b;
    int64_t onehot_stride_t;
    const sad_rnn_layer* upper; /* layers above the first, bottom-up, or NULL */
    int32_t n_upper;            /* 0 .. SAD_MAX_UPPER_LAYERS */
} sad_fwd_args;

typedef struct sad_bwd_args {
    /* d logits = dlogits (if given) + dCE(logits, labels)/dlogits * upstream (if labels are given and an
     * upstream gradient of the per-sentence loss is: dloss, or dloss_scale != 0, or dlogits == NULL). */
    const float* dlogits;     /* [B,L] upstream gradient of the logits, or NULL */
    const int64_t* labels;    /* [B] labels of the fused CE, or NULL */
    const float* dloss;       /* [B] upstream gradient of the per-sentence loss, or NULL -> */
    float dloss_scale;        /*   this scalar for every sentence (1/B for `.mean()`)     */
    const float* dweights;    /* [B,T] upstream gradient of the attention weights, or NULL */
    const float* logits;      /* [B,L] as written by sad_forward */
    const float* weights;     /* [B,T] as written by sad_forward */
    void* dsrc;               /* out: d(soft tokens) in dsrc_kind (F32/BF16) and strides, or NULL */
    int32_t dsrc_kind;
    int64_t dsrc_stride_b;
    int64_t dsrc_stride_t;
    const uint8_t* drop_keep; /* the forward's dropout mask (or NULL) and scale: applied to d(embedded tokens) */
    float drop_scale;
    const sad_rnn_layer* upper; /* exactly what the matching sad_forward got */
    int32_t n_upper;
} sad_bwd_args;

/* Gradient tensors of the live parameters, fp32, state_dict layout (same shapes as sad_weights).
 * A NULL field is skipped (that tensor is frozen: `requires_grad = False`, e.g. after
 * freeze_emb(), reference classifier/models.py:151-154). */
typedef struct sad_param_grads {
    float* embedding_weight;
    float* rnn_weight_ih[2];
    float* rnn_weight_hh[2];
    float* rnn_bias_ih[2];
    float* rnn_bias_hh[2];
    float* hidden_bridge0_weight;
    float* hidden_bridge0_bias;
    float* linear_emb_weight;
    float* linear_emb_bias;
    float* linear_q_weight;
    float* linear_q_bias;
    float* linear_attn_weight;
    float* linear_attn_bias;
    float* output_layer_weight;
    float* output_layer_bias;
    const void* final_hidden_workspace; /* multi-layer only: workspace of the LAST layer (its final hidden states are
                                           the input of hidden_bridges[0]); NULL = the workspace of the call */
} sad_param_grads;

/* Named regions of the workspace, for tests / profiling (sad_workspace_field). */
enum sad_ws_field {
    SAD_WS_LENS = 0, /* int32 [B] clamped lengths */
    SAD_WS_NEED,     /* u8   [T*B] */
    SAD_WS_X,        /* bf16 [T*B][Ep] */
    SAD_WS_G,        /* fp32 [T*B][2*Gp+64] activated gates [fwd | rev | pad], permuted columns */
    SAD_WS_CS,       /* fp32 [T*B][2][Hp] */
    SAD_WS_HS,       /* bf16 [T*B][2][Hp] */
    SAD_WS_KK,       /* fp32 [T*B][Hp] */
    SAD_WS_Q,        /* fp32 [B][Hp] */
    SAD_WS_U,        /* fp32 [B][Hp] */
    SAD_WS_CTX,      /* fp32 [B][Hp] */
    SAD_WS_DZ,       /* bf16 [T*B][Ncat] */
    SAD_WS_DX,       /* bf16 [T*B][Ep] */
    SAD_WS_DU,       /* fp32 [B][Hp] */
    SAD_WS_DQ,       /* fp32 [B][Hp] */
    SAD_WS_DHFIN,    /* fp32 [B][2][Hp] */
    SAD_WS_DC,       /* fp32 [B][2][Hp] */
    SAD_WS_ACCF,     /* fp32 [T*B][Ep] split-K accumulator (small batches) */
    SAD_WS_DL,       /* fp32 [B][L]  dlogits as used by the backward */
    SAD_WS_DWA,      /* fp32 [B][Hp] per-sentence d linear_attn.weight */
    SAD_WS_DSSUM,    /* fp32 [B]     per-sentence d linear_attn.bias */
    SAD_WS_ROWS,     /* int32 [3][T*B] compacted lists of contributing token rows (in-projection, weight gradients) */
    SAD_WS_NROWS,    /* int32 [4]    their lengths */
    SAD_WS_SYNC,     /* u32: hand-off counters of the persistent recurrence (one 128-B line per sentence
                        tile and direction; 2 * ceil(B / 32) lines), then one line with the timeout words: [0] forward recurrence,
                        [1] BPTT (non-zero = a hand-off wait timed out; the outputs are NaN then) */
    SAD_WS_DZX,      /* bf16 [2][2][B/16][Gp/8][16][8] hand-off image of dZ inside the persistent BPTT */
    SAD_WS_HX,       /* bf16 [2][2][B/16][Hp/8][16][8] hand-off image of h inside the persistent recurrence */
    SAD_WS_K1TILES,  /* int32: count + list of the 128-row tiles with a needed row (stream-K form of K1) */
    SAD_WS_K1SLABS,  /* fp32 [256][2][128*320] partial tiles of the stream-K form of K1, accumulator-fragment order */
    SAD_WS_GA,       /* fp16 [T][2][B/16][Hp/4][16 sentences][4 units][i, f, g, o] activated gates when
                        sad_uses_half_gates() */
    SAD_WS_NFIELDS
};

int sad_version(void);
const char* sad_last_error(void);

/* Padded sizes derived from dims: out[0..4] = Ep, Hp, Gp, Vp, Ncat. */
int sad_padded_dims(const sad_dims* d, int32_t out[5]);

/* Bytes of the packed (bf16, padded, gate-permuted, pre-transposed) weight image. */
size_t sad_packed_bytes(const sad_dims* d);
/* Bytes of the per-call workspace (saved activations + backward scratch). 256-byte aligned base. */
size_t sad_workspace_bytes(const sad_dims* d);
int sad_workspace_field(const sad_dims* d, int field, size_t* offset, size_t* bytes);
/* 1 when, for these dims on the current device, the recurrences run as persistent kernels and keep
 * the activated LSTM gates as fp16 records in SAD_WS_GA; 0 when the activated gates are fp32 in SAD_WS_G.
 * In the first case, at the shipped embedding size (E = 289..320), the gate pre-activations of
 * transfer/models.py:906's nn.LSTM input GEMM are fp16 records in SAD_WS_GA too (the input projection writes the
 * record that the forward recurrence reads and then overwrites with the activated gates) and SAD_WS_G is unused;
 * at other sizes (e.g. the 2H-wide input of an upper RNN layer) they are fp32 rows of SAD_WS_G. */
int sad_uses_half_gates(const sad_dims* d);

/* state_dict fp32 -> packed image.  Replaces nothing in the reference (it consumes fp32 parameters
 * directly, transfer/models.py:770-827); call again whenever the parameters change. */
int sad_pack_weights(const sad_dims* d, const sad_weights* w, void* packed, void* stream);

/* SelfAttnRNNClassifier.forward (reference transfer/models.py:885-953 incl. attention :841-883;
 * classifier copy classifier/models.py:200-266 with guard = 0) + per-sentence
 * nn.CrossEntropyLoss(reduction='none') (transfer/models.py:990,1212) when labels/loss are given.
 * Saves what the backward needs in `workspace`.
 * Returns -4 (and starts nothing) when a hand-off wait inside a persistent recurrence of an EARLIER call on the
 * current device timed out: that call's outputs are NaN by construction, the recurrence raised a word of
 * host-mapped memory (created by the first sad_pack_weights on the device) and this call reports and clears it --
 * no synchronisation involved; the call after it runs normally.  (The reference has no counterpart: its callers
 * catch RuntimeError around the step, classifier/trainers.py:375-382.) */
int sad_forward(const sad_dims* d, const void* packed, const sad_fwd_args* a, void* workspace, void* stream);

/* sad_forward without a workspace argument and without any allocation.  It has NO counterpart in the
 * reference; it exists for callers that cannot allocate device memory around the call: code under HIP
 * graph capture, tight serving loops that score hard tokens (calls #1 / #3 of the transfer step,
 * reference transfer/models.py:1065-1074, 1233-1262), harnesses that forbid hipMalloc.  The workspace is
 * a static device array of SAD_NOALLOC_MAX_BYTES inside the library (each GPU the library runs on holds
 * it until the process ends); shapes whose sad_workspace_bytes() exceeds it are refused (-3) with a
 * message.  Calls on one GPU share that array: calls that overlap on one GPU (two streams, two host
 * threads) race — serialise them.  Forward only: nothing is saved for a sad_backward. */
#define SAD_NOALLOC_MAX_BYTES (192u << 20)
int sad_forward_noalloc(const sad_dims* d, const void* packed, const sad_fwd_args* a, void* stream);

/* Autograd backward of the above down to d(soft tokens) with frozen parameters — what
 * loss.backward() (reference transfer/trainers.py:376) runs through transfer/models.py:1185-1187.
 * `workspace` must be the one the matching sad_forward filled. */
int sad_backward(const sad_dims* d, const void* packed, const sad_bwd_args* a, void* workspace, void* stream);

/* Parameter gradients ("unfrozen" mode): what loss.backward() leaves in `.grad` of the module's
 * parameters when they require grad (classifier training, reference classifier/trainers.py:371-373;
 * transfer/models.py:757-829 lists the tensors).  Call AFTER sad_backward on the same workspace
 * (sad_backward may be called with dsrc = NULL when only parameter gradients are wanted).
 * Every non-NULL tensor of `g` is ACCUMULATED into (+=, fp32 atomics): zero-fill for plain
 * gradients, or keep accumulating over micro-batches.  `src` / strides are the forward's input
 * (needed for d embedding.weight: S^T . dX for soft tokens, scatter-add for ids, where the
 * `pad_id` row gets no gradient as with nn.Embedding(padding_idx=pad_id); pass -1 for none). */
int sad_param_grads_accumulate(const sad_dims* d, const sad_param_grads* g, const void* src, int32_t src_kind,
                               int64_t src_stride_b, int64_t src_stride_t, int32_t pad_id, void* workspace,
                               void* stream);

/* Single stages, for unit tests and for timing the dominant kernels in isolation (bench.py).
 * sad_prep: length clamp + mask guard + need bits (reference transfer/models.py:865-867), and from the
 *           need bits the compacted row lists (ROWS / NROWS), the list of non-empty 128-row tiles
 *           (K1TILES) and zeroed hand-off words (SYNC): the same launch(es) that open sad_forward.
 * sad_softemb_fwd:  X = S @ W_emb            (reference transfer/models.py:898), S -> workspace X.
 * sad_softemb_bwd:  dS = dX @ W_emb^T        (its autograd), workspace DX -> dsrc.
 * Both (and sad_param_grads_accumulate) read what a previous sad_prep / sad_forward left in `workspace`:
 * need bits, row lists, tile list. */
int sad_prep(const sad_dims* d, const int32_t* lengths, uint8_t* mask, int64_t mask_stride_b,
             int64_t mask_stride_t, int32_t guard, void* workspace, void* stream);
int sad_softemb_fwd(const sad_dims* d, const void* packed, const void* src, int32_t src_kind,
                    int64_t src_stride_b, int64_t src_stride_t, void* workspace, void* stream);
int sad_softemb_bwd(const sad_dims* d, const void* packed, void* dsrc, int32_t dsrc_kind,
                    int64_t dsrc_stride_b, int64_t dsrc_stride_t, void* workspace, void* stream);
/* sad_inproj_fwd:   input projection of both LSTM directions + linear_emb keys in one GEMM (reference
 *                   transfer/models.py:906 nn.LSTM's W_ih products, :925): workspace X -> G (or GA records, see
 *                   sad_uses_half_gates), KK.
 * sad_inproj_bwd:   its autograd: workspace DZ -> DX.
 * sad_lstm_fwd:     the packed BiLSTM recurrence (reference transfer/models.py:904-906, ctor :779-785):
 *                   workspace G / GA (gate pre-activations of the input projection) + LENS -> HS, CS and the
 *                   activated gates (GA, or G in place when sad_uses_half_gates() == 0).  With pre-activation
 *                   records in GA the stage consumes them: run sad_inproj_fwd again before repeating it.
 * sad_lstm_bwd:     its BPTT (autograd of :906): workspace DHFIN + saved gates / cell states -> gate columns
 *                   of DZ.
 * sad_attnpool_fwd: additive attention + masked softmax + pooling + head (+ CE) (reference
 *                   transfer/models.py:855-880, :949, CE :990,:1212): workspace KK, U, NEED -> logits,
 *                   weights, loss (NULL to skip), CTX.
 * sad_attnpool_bwd: its backward: -> key columns of DZ, DU and the per-sentence pieces of the head's
 *                   parameter gradients; `a` as for sad_backward (dsrc / drop_keep ignored).
 * All six work on a workspace that a sad_forward (and, for the backward stages, the stages in front
 * of them in sad_backward's order) has filled. */
int sad_inproj_fwd(const sad_dims* d, const void* packed, void* workspace, void* stream);
int sad_inproj_bwd(const sad_dims* d, const void* packed, void* workspace, void* stream);
int sad_lstm_fwd(const sad_dims* d, const void* packed, void* workspace, void* stream);
int sad_lstm_bwd(const sad_dims* d, const void* packed, void* workspace, void* stream);
int sad_attnpool_fwd(const sad_dims* d, const void* packed, const int64_t* labels, float* logits, float* weights,
                     float* loss, void* workspace, void* stream);
int sad_attnpool_bwd(const sad_dims* d, const void* packed, const sad_bwd_args* a, void* workspace, void* stream);

/* ---- fused producer of the soft tokens ---------------------------------------------
 * One decode step of AttnRNNDecoder.soft_decode (reference transfer/models.py:531-543): the straight-through
 * Gumbel-softmax sample  y_hard - y_soft + y_soft,  y_soft = softmax((logits + noise) / tau),  written
 * straight into its row of the time-major soft-token tensor in the dtype the discriminator consumes, plus
 * the (index, value) side channel and the alive / length bookkeeping.  The caller draws `noise`
 * (-log(Exp(1)), as F.gumbel_softmax does) with its own generator. */
typedef struct sad_gumbel_args {
    int32_t B, V;
    const float* logits;      /* [B,V], row stride logits_stride */
    int64_t logits_stride;
    const float* noise;       /* [B,V] Gumbel(0,1) samples, row stride noise_stride */
    int64_t noise_stride;
    float tau;
    const uint8_t* alive;     /* [B] soft_decode's `mask` before this step (1 = still decoding), NULL = all */
    int32_t eos_id;
    void* s_row;              /* out: row t of S for every sentence: sample * alive */
    int32_t s_kind;           /* SAD_SRC_F32 / SAD_SRC_BF16 */
    int64_t s_stride_b;       /* element stride between sentences inside that row (V for dense time-major S) */
    float* rnn_in;            /* out [B,V] fp32: the un-masked sample (next decoder input), or NULL */
    float* y_soft;            /* out [B,V] fp32: saved for the backward, or NULL */
    int64_t* idx;             /* out [B]: arg-max index */
    float* val;               /* out [B]: value at the arg-max times alive */
    int32_t* lens;            /* in/out [B] or NULL: += alive        (outputs_len += mask, :542) */
    uint8_t* alive_out;       /* out [B] or NULL: alive && idx != eos (:543); may alias `alive` */
} sad_gumbel_args;
int sad_gumbel_st_step(const sad_gumbel_args* a, void* stream);
/* Its autograd: dlogits = y_soft * (g - <g, y_soft>) / tau with g = alive * d s_row (+ d rnn_in); either
 * gradient source may be NULL. */
int sad_gumbel_st_step_bwd(int32_t B, int32_t V, const void* d_s_row, int32_t s_kind, int64_t s_stride_b,
                           const float* d_rnn_in, const uint8_t* alive, const float* y_soft, float tau,
                           float* dlogits, int64_t dlogits_stride, void* stream);
/* After the last step (reference transfer/models.py:546-571, 1180-1183): [sos] row 0, an [eos] row for
 * sentences still alive when the length limit stopped the loop (`over_max`), one-hot [pad] for t >= len, the
 * side channel of those rows, final lengths and the pad+sos+eos mask.  S is [T][B][V] through element
 * strides; `steps` decode steps filled rows 1..steps; T = steps + 1, + 1 if a sentence gets the [eos] row
 * (over_max and still alive: the caller knows from the loop's own `any(alive)` test). */
int sad_soft_finalize(int32_t B, int32_t T, int32_t V, int32_t steps, int32_t over_max, int32_t pad_id, int32_t sos_id,
                      int32_t eos_id, void* S, int32_t s_kind, int64_t s_stride_t, int64_t s_stride_b, int64_t* ids,
                      float* val, const int32_t* lens_steps, const uint8_t* alive, int32_t* lens_out, uint8_t* mask,
                      void* stream);

/* Test / diagnosis settings (process-wide; no counterpart in the reference).  Do not change them
 * between a sad_forward and its sad_backward.
 *   SAD_DBG_NO_PERSIST    1: run the recurrences as one launch per time step (also: SAD_NO_PERSIST=1 in
 *                         the environment at the first call), 0: persistent kernels when they fit;
 *   SAD_DBG_SPIN_LIMIT    bound (poll iterations) of a hand-off wait inside the persistent recurrences,
 *                         <= 0 restores the default;
 *   SAD_DBG_WITHHOLD_FWD  index of a workgroup of the persistent forward recurrence that exits at once,
 *   SAD_DBG_WITHHOLD_BWD  the same for the persistent BPTT (-1: none): forces the hand-off timeout path,
 *                         which must end in NaN logits / loss / d(soft tokens) and a raised timeout word
 *                         in SAD_WS_SYNC, never in silently wrong numbers. */
#define SAD_DBG_NO_PERSIST 0
#define SAD_DBG_SPIN_LIMIT 1
#define SAD_DBG_WITHHOLD_FWD 2
#define SAD_DBG_WITHHOLD_BWD 3
#define SAD_DBG_WITHHOLD_UPPER_ONLY 4 /* 1: the two settings above act only on RNN layers above the first */
#define SAD_DBG_WGRAD_ATOMICS 5 /* 1: the large weight-gradient kernels add their partial tiles with fp32 atomics (the
                                  * fallback for launches whose partial tiles do not fit the slab) instead of slab + reduce */
int sad_debug_set(int32_t key, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* SAD_H_ */
