"""Drop-in ``SelfAttnRNNClassifier`` — host-side mirror of the reference interface.

Same constructor, same 23 ``state_dict`` tensors (names, shapes, fp32; a reference
``model_state_dict.pt`` loads unchanged), same ``forward`` signatures and return values as the
reference's two copies:

* transfer copy  — reference transfer/models.py:753-953
  ``forward(src, src_len, pad_sos_eos_mask, hidden=None, soft=False)``; the mask argument is
  mutated in place for all-masked rows (:865-867).
* classifier copy — reference classifier/models.py:76-299
  ``forward(src, src_len, pad_mask=None, hidden=None, soft=False)``; derives the pad+sos+eos mask
  from the pad mask (:183-187), no all-masked guard; ``set_mask_id`` / ``get_masked_src``.

Every configuration the reference constructor accepts is covered (reference
classifier/models.py:102-124): ``cell`` LSTM / GRU, ``enc_bidirectional`` True / False, ``enc_num_layers`` >= 1.
The kernels run ONE canonical layout — bidirectional, four gate slots per unit — and the variants are
embedded into it here, with autograd carrying the gradients back to the real parameters:

* GRU: ``weight_ih -> [W_ir; W_iz; W_in; 0]``, ``weight_hh -> [W_hr; W_hz; 0; W_hn]`` (biases alike), cell
  code ``CELL_GRU`` (include/sad.h);
* uni-directional: the reverse direction gets zero weights (its hidden state is exactly 0) and the bridge /
  upper-layer input matrices are zero-extended from ``H`` to ``2H`` columns;
* layers above the first: one weight image + workspace each (``sad_rnn_layer``), input = the per-step
  outputs of the layer below; nn.LSTM's inter-layer dropout (classifier copy, training mode) uses masks drawn
  here.

The shipped configuration (LSTM, bidirectional, one layer) passes its parameters through untouched.  Every
configuration runs the same kernels, the persistent recurrences included (GRU: a second cell epilogue; layers
above the first: their own launch per layer).

The submodules (``nn.Embedding``, ``nn.LSTM`` / ``nn.GRU``, ``nn.Linear``) exist only as PARAMETER HOLDERS so
that ``named_parameters()`` filtering, ``requires_grad`` freezing, ``clip_grad_norm_`` and the
optimizers behave exactly as with the reference; they are constructed in the
reference's order, so seeding gives the same initial values.  Their ``forward`` is never called:
all arithmetic runs in libsad's HIP kernels, and nothing falls back to stock torch ops.
"""
import torch
import torch.nn as nn

from . import functional as Fn
from . import lib as _lib


class _SelfAttnBase(nn.Module):
    _flavour = None

    def __init__(self, config, embedding=None):
        super().__init__()
        self.cell = getattr(config, "cell", "LSTM").upper()
        self.bidirectional = getattr(config, "enc_bidirectional", True)
        pad_id = getattr(config, "pad_id", 0)
        num_layers = getattr(config, "enc_num_layers", 1)
        self.mask_id = getattr(config, "mask_id", None)
        self.hidden_size = config.hidden_size
        self.dropout_p = float(getattr(config, "dropout", 0.0)) if self._flavour == "classifier" else 0.0
        assert self.cell == "LSTM" or self.cell == "GRU"
        if not 1 <= num_layers <= 1 + _lib.MAX_UPPER_LAYERS:
            raise ValueError("enc_num_layers must be 1..%d, got %d" % (1 + _lib.MAX_UPPER_LAYERS, num_layers))
        self.num_layers = num_layers
        D = 2 if self.bidirectional else 1

        # parameter holders, created in the reference's order (transfer/models.py:770-827)
        self.embedding = nn.Embedding.from_pretrained(
            torch.FloatTensor(embedding), freeze=config.freeze_embedding, padding_idx=pad_id
        ) if embedding is not None else nn.Embedding(config.vocab_size, config.emb_size, padding_idx=pad_id)
        # (inter-layer dropout: classifier copy only, classifier/models.py:108 vs transfer/models.py:783)
        self.rnn = getattr(nn, self.cell)(input_size=self.embedding.embedding_dim, hidden_size=self.hidden_size,
                                          num_layers=num_layers, dropout=(0.0 if num_layers == 1 else self.dropout_p),
                                          bidirectional=self.bidirectional)
        self.output_bridge = nn.Linear(D * self.hidden_size, self.hidden_size)      # dead in forward
        self.hidden_bridges = nn.ModuleList([nn.Linear(D * self.hidden_size, self.hidden_size)
                                             for _ in range(2 if self.cell == "LSTM" else 1)])   # [1] is dead
        if self._flavour == "classifier":
            self.dropout_layer = nn.Dropout(self.dropout_p)
        self.linear_emb = nn.Linear(self.embedding.embedding_dim, self.hidden_size)
        self.linear_q = nn.Linear(self.hidden_size, self.hidden_size)
        self.linear_attn = nn.Linear(self.hidden_size, 1)
        self.output_layer = nn.Linear(self.hidden_size, config.num_labels)
        self.device = config.device
        self._pw = None
        self._pw_key = None

    # -- reference API ----------------------------------------------------------------------------
    def freeze_emb(self):
        for param in self.embedding.parameters():
            param.requires_grad = False

    def unfreeze_emb(self):
        for param in self.embedding.parameters():
            param.requires_grad = True

    # -- packed weights ---------------------------------------------------------------------------
    def _real(self):
        """The parameters that reach the logits (everything but output_bridge.* / hidden_bridges.1.*), read from
        the sub-modules' own tables: `named_parameters()` walks the module tree with a de-duplication set, ~90 us of
        host time per call, which a batch-64 step through this module waits for."""
        out = {}
        for prefix, mod in (("embedding.", self.embedding), ("rnn.", self.rnn), ("hidden_bridges.0.", self.hidden_bridges[0]),
                            ("linear_emb.", self.linear_emb), ("linear_q.", self.linear_q),
                            ("linear_attn.", self.linear_attn), ("output_layer.", self.output_layer)):
            for name, p in mod._parameters.items():
                if p is not None:
                    out[prefix + name] = p
        return out

    def _slots4(self, w, hidden_half):
        """GRU [3H, .] -> the kernels' four gate slots [4H, .]: input half [r, z, n, 0], hidden half [r, z, 0, n]."""
        if self.cell == "LSTM":
            return w
        H = self.hidden_size
        z = w.new_zeros((H,) + tuple(w.shape[1:]))
        return torch.cat([w[:2 * H], z, w[2 * H:]], 0) if hidden_half else torch.cat([w, z], 0)

    def _canonical(self):
        """(the 19 tensors of layer 0, [the 8 rnn.* tensors of every layer above]) in the kernels' canonical
        layout, built from the real parameters with differentiable torch ops (cat with zeros): the shipped
        configuration returns its parameters themselves."""
        sd = dict(self.named_parameters())

        def widen(w):            # uni-directional: [., H] -> [., 2H] (the reverse half is exactly zero)
            return w if self.bidirectional else torch.cat([w, torch.zeros_like(w)], 1)

        layers = []
        for l in range(self.num_layers):
            out = {}
            for sfx in ("", "_reverse"):
                k = "_l%d%s" % (l, sfx)
                if sfx and not self.bidirectional:
                    for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                        out["rnn.%s_l0%s" % (n, sfx)] = torch.zeros_like(out["rnn.%s_l0" % n])
                    continue
                w_ih = self._slots4(sd["rnn.weight_ih" + k], False)
                out["rnn.weight_ih_l0" + sfx] = widen(w_ih) if l > 0 else w_ih
                out["rnn.weight_hh_l0" + sfx] = self._slots4(sd["rnn.weight_hh" + k], True)
                out["rnn.bias_ih_l0" + sfx] = self._slots4(sd["rnn.bias_ih" + k], False)
                out["rnn.bias_hh_l0" + sfx] = self._slots4(sd["rnn.bias_hh" + k], True)
            layers.append(out)
        live = dict(layers[0])
        for k in Fn._LIVE_PARAMS:
            if k not in live:
                live[k] = widen(sd[k]) if k == "hidden_bridges.0.weight" else sd[k]
        return live, layers[1:]

    def packed_weights(self, real=None):
        """Packed device image of the current parameters; re-packed whenever a live parameter's
        storage or version counter changes (optimizer step, load_state_dict, .to())."""
        if real is None:
            real = self._real()
        key = tuple((p.data_ptr(), p._version, p.device) for p in real.values())
        if self._pw is None or key != self._pw_key:
            with torch.no_grad():
                live, upper = self._canonical()
                self._pw = Fn.pack_weights(live, cell=_lib.CELL_GRU if self.cell == "GRU" else _lib.CELL_LSTM, upper=upper)
            self._pw_key = key
        return self._pw

    def _trainable(self, real=None):
        """The live parameters (canonical layout) when the call must be differentiable w.r.t. them (classifier
        training, reference classifier/trainers.py:367-373), else (None, ()): the frozen transfer-loop mode
        (reference transfer/trainers.py:216-223) skips every weight-gradient kernel."""
        if real is None:
            real = self._real()
        if torch.is_grad_enabled() and any(p.requires_grad for p in real.values()):
            return self._canonical()
        return None, ()

    def _draw_layer_keep(self, src):
        """Keep masks of nn.LSTM's inter-layer dropout [T,B,2H] per layer boundary (classifier copy in training
        mode with enc_num_layers > 1, reference classifier/models.py:108); None otherwise."""
        if self._flavour != "classifier" or not self.training or self.dropout_p <= 0.0 or self.num_layers == 1:
            return None
        B, T = src.shape[0], src.shape[1]
        return [(torch.rand((T, B, 2 * self.hidden_size), device=src.device) >= self.dropout_p, 1.0 / (1.0 - self.dropout_p))
                for _ in range(self.num_layers - 1)]

    def _run(self, src, src_len, mask, hidden, soft, guard, drop=None, onehot=None):
        if hidden is not None:
            raise NotImplementedError("an initial hidden state is never passed by the reference callers")
        if soft and src.dim() != 3:
            raise RuntimeError("soft=True expects [B,T,V] soft tokens")
        if not soft and src.dim() != 2:
            raise RuntimeError("soft=False expects [B,T] token ids")
        real = self._real()
        params, upper = self._trainable(real)
        return Fn.discriminator(src, src_len, mask, self.packed_weights(real), guard=guard, params=params,
                                pad_id=self.embedding.padding_idx, drop=drop, onehot=onehot,
                                layer_drop=self._draw_layer_keep(src), upper_params=upper)

    def _draw_keep(self, src):
        """Inverted-dropout keep mask on the embedded tokens [T,B,E] (classifier copy in training mode,
        reference classifier/models.py:126,217): drawn with torch's generator (plumbing), applied by the
        kernels in the forward and in the backward.  None for the transfer copy / eval mode / p = 0."""
        if self._flavour != "classifier" or not self.training or self.dropout_p <= 0.0:
            return None
        B, T = src.shape[0], src.shape[1]
        keep = torch.rand((T, B, self.embedding.embedding_dim), device=src.device) >= self.dropout_p
        return keep, 1.0 / (1.0 - self.dropout_p)

    def loss(self, src, src_len, mask, labels, soft=True):
        """Fused variant: per-sentence CE vs ``labels`` computed inside the attention kernels
        (reference transfer/models.py:1185-1212). Returns (loss [B], logits, weights).  ``mask`` is the
        pad+sos+eos mask (both flavours); the classifier flavour applies the same training-mode dropout
        as its ``forward``."""
        if soft and src.dim() != 3:
            raise RuntimeError("soft=True expects [B,T,V] soft tokens")
        if not soft and src.dim() != 2:
            raise RuntimeError("soft=False expects [B,T] token ids")
        real = self._real()
        params, upper = self._trainable(real)
        return Fn.discriminator_loss(src, src_len, mask, labels, self.packed_weights(real),
                                     guard=self._flavour == "transfer", params=params,
                                     pad_id=self.embedding.padding_idx, drop=self._draw_keep(src),
                                     layer_drop=self._draw_layer_keep(src), upper_params=upper)


class TransferSelfAttnRNNClassifier(_SelfAttnBase):
    """Reference transfer/models.py:753-953."""
    _flavour = "transfer"

    def forward(self, src, src_len, pad_sos_eos_mask, hidden=None, soft=False, onehot=None):
        """``onehot`` (optional, not in the reference): (ids, val) side channel of straight-through one-hot
        soft tokens from producer.SoftTokenAssembler — the soft-embedding contraction becomes a gather."""
        return self._run(src, src_len, pad_sos_eos_mask, hidden, soft, guard=True, onehot=onehot)


class ClassifierSelfAttnRNNClassifier(_SelfAttnBase):
    """Reference classifier/models.py:76-299."""
    _flavour = "classifier"

    def set_mask_id(self, mask_id):
        self.mask_id = mask_id

    @staticmethod
    def _pad_sos_eos_mask(pad_mask):
        # classifier/models.py:183-187: [True | pad[:, 2:] | True]
        ones = torch.ones((pad_mask.shape[0], 1), dtype=torch.bool, device=pad_mask.device)
        return torch.cat([ones, pad_mask[:, 2:], ones], dim=1)

    def forward(self, src, src_len, pad_mask=None, hidden=None, soft=False):
        drop = self._draw_keep(src)
        if pad_mask is None:
            mask = torch.zeros(src.shape[:2], dtype=torch.bool, device=src.device)
        else:
            mask = self._pad_sos_eos_mask(pad_mask)
        return self._run(src, src_len, mask, hidden, soft, guard=False, drop=drop)

    def get_masked_src(self, src, src_len, pad_mask):
        """classifier/models.py:269-299 (thin post-processing around forward)."""
        if self.mask_id is None:
            self.mask_id = 0
        pad_sos_eos_mask = self._pad_sos_eos_mask(pad_mask)
        cls_logits, style_alpha = self.forward(src, src_len, pad_mask)
        src_len = torch.as_tensor(src_len).to(style_alpha.device)
        threshold = (1.0 / (src_len - 2).float()).unsqueeze(1)
        inv = torch.logical_not(pad_sos_eos_mask)
        style_mask = (style_alpha <= threshold) * inv
        content_mask = (style_alpha > threshold) * inv
        style_src = src.clone()
        content_src = src.clone()
        style_src[style_mask] = self.mask_id
        content_src[content_mask] = self.mask_id
        return cls_logits, style_alpha, style_src, content_src


# The hot path is the transfer copy (the one embedded in TransferModel as `.classifier`,
# reference transfer/models.py:961).
SelfAttnRNNClassifier = TransferSelfAttnRNNClassifier


class CNNClassifier(nn.Module):
    """Drop-in for the reference's CNN metric classifier (reference classifier/models.py:8-73; used as the
    transfer-accuracy metric, transfer/trainers.py:247-248,523).  Same constructor, state_dict and
    ``forward(src, soft=False)``.  SURVEY.md §8 f4 scopes its SOFT branch (:53-54): the contraction
    ``src @ embedding.weight`` runs in libsad's K1 / K1' kernels (``functional.soft_embedding``, differentiable
    in both operands, bf16 MFMA inputs with fp32 accumulation).  The convolution stack behind it is the eval
    metric's own model (out of scope) and stays on torch's library ops."""

    def __init__(self, config, embedding=None):
        super().__init__()
        pad_id = getattr(config, "pad_id", 0)
        dropout = getattr(config, "dropout", 0.0)
        self.num_kernels_each_size = config.num_kernels_each_size
        self.kernel_sizes = config.kernel_sizes
        hidden_size = config.num_kernels_each_size * len(config.kernel_sizes)
        self.embedding = nn.Embedding.from_pretrained(
            torch.FloatTensor(embedding), freeze=config.freeze_embedding, padding_idx=pad_id
        ) if embedding is not None else nn.Embedding(config.vocab_size, config.emb_size, padding_idx=pad_id)
        self.cnns = nn.ModuleList([nn.Conv2d(1, config.num_kernels_each_size, (ks, config.emb_size))
                                   for ks in config.kernel_sizes])
        self.ffn = nn.Linear(hidden_size, config.num_labels)
        self.dropout_layer = nn.Dropout(dropout)
        self.device = config.device

    def forward(self, src, soft=False):
        if soft:
            if src.dim() != 3:
                raise RuntimeError("soft=True expects [B,T,V] soft tokens")
            embedded = Fn.soft_embedding(src, self.embedding.weight)      # classifier/models.py:54
        else:
            embedded = self.embedding(src)
        embedded = self.dropout_layer(embedded)
        return self.ffn(self._pooled_features(embedded))

    def _pooled_features(self, embedded):
        """[B, T, E] -> [B, len(kernel_sizes) * num_kernels_each_size]: per window size the strongest response of
        every filter over the sentence (the reference's relu -> max_pool1d over all positions, classifier/models.py:
        58-70, written as ONE reduction: relu is monotone, so max_t relu(y_t) == relu(max_t y_t)); a window longer
        than the sentence leaves its block of features at zero, as there."""
        B, T = embedded.shape[0], embedded.shape[1]
        feats = embedded.new_zeros((B, len(self.cnns), self.num_kernels_each_size))
        image = embedded.unsqueeze(1)                                      # one input channel
        for slot, (width, conv) in enumerate(zip(self.kernel_sizes, self.cnns)):
            if width <= T:
                feats[:, slot] = torch.relu(conv(image).squeeze(3).amax(dim=2))
        return feats.flatten(1)
