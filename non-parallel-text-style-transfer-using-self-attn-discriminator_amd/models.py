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

The submodules (``nn.Embedding``, ``nn.LSTM``, ``nn.Linear``) exist only as PARAMETER HOLDERS so
that ``named_parameters()`` filtering, ``requires_grad`` freezing, ``clip_grad_norm_`` and the
optimizers behave exactly as with the reference; they are constructed in the
reference's order, so seeding gives the same initial values.  Their ``forward`` is never called:
all arithmetic runs in libsad's HIP kernels, and nothing falls back to stock torch ops.
"""
import torch
import torch.nn as nn

from . import functional as Fn


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
        if self.cell != "LSTM" or not self.bidirectional or num_layers != 1:
            # GRU / uni-directional / multi-layer variants are not on the hot path.
            raise NotImplementedError(
                "sad_amd implements the shipped configuration only: cell=LSTM, enc_bidirectional=True, "
                "enc_num_layers=1 (got cell=%s bidirectional=%s layers=%d)" % (self.cell, self.bidirectional, num_layers))

        # parameter holders, created in the reference's order (transfer/models.py:770-827)
        self.embedding = nn.Embedding.from_pretrained(
            torch.FloatTensor(embedding), freeze=config.freeze_embedding, padding_idx=pad_id
        ) if embedding is not None else nn.Embedding(config.vocab_size, config.emb_size, padding_idx=pad_id)
        self.rnn = nn.LSTM(input_size=self.embedding.embedding_dim, hidden_size=self.hidden_size,
                           num_layers=1, dropout=0.0, bidirectional=True)
        self.output_bridge = nn.Linear(2 * self.hidden_size, self.hidden_size)      # dead in forward
        self.hidden_bridges = nn.ModuleList([nn.Linear(2 * self.hidden_size, self.hidden_size)
                                             for _ in range(2)])                    # [1] is dead
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
    def _live(self):
        sd = dict(self.named_parameters())
        return {k: sd[k] for k in Fn._LIVE_PARAMS}

    def packed_weights(self):
        """Packed device image of the current parameters; re-packed whenever a live parameter's
        storage or version counter changes (optimizer step, load_state_dict, .to())."""
        live = self._live()
        key = tuple((p.data_ptr(), p._version, p.device) for p in live.values())
        if self._pw is None or key != self._pw_key:
            self._pw = Fn.pack_weights(live)
            self._pw_key = key
        return self._pw

    def _trainable(self):
        """The live parameters when the call must be differentiable w.r.t. them (classifier training,
        reference classifier/trainers.py:367-373), else None: the frozen transfer-loop mode
        (reference transfer/trainers.py:216-223) skips every weight-gradient kernel."""
        live = self._live()
        if torch.is_grad_enabled() and any(p.requires_grad for p in live.values()):
            return live
        return None

    def _run(self, src, src_len, mask, hidden, soft, guard, drop=None, onehot=None):
        if hidden is not None:
            raise NotImplementedError("an initial hidden state is never passed by the reference callers")
        if soft and src.dim() != 3:
            raise RuntimeError("soft=True expects [B,T,V] soft tokens")
        if not soft and src.dim() != 2:
            raise RuntimeError("soft=False expects [B,T] token ids")
        return Fn.discriminator(src, src_len, mask, self.packed_weights(), guard=guard, params=self._trainable(),
                                pad_id=self.embedding.padding_idx, drop=drop, onehot=onehot)

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
        return Fn.discriminator_loss(src, src_len, mask, labels, self.packed_weights(),
                                     guard=self._flavour == "transfer", params=self._trainable(),
                                     pad_id=self.embedding.padding_idx, drop=self._draw_keep(src))


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
