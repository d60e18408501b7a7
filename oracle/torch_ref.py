"""ORACLE — test infrastructure only. NOT part of the product path.

Stock-``torch.nn`` restatement of the reference discriminator's op sequence (matmul ->
``pack_padded_sequence`` -> ``nn.LSTM`` -> ``nn.Linear``s -> masked softmax -> ``bmm`` -> CE, then
autograd), written from SURVEY.md §3.2.  It is (a) the "reference CPU path" that ``bench.py``'s
``cpu_baseline`` leg times on the GPU box's host cores (kind "port": the reference itself cannot
travel to the GPU box) and (b) a second checker for the numpy oracle.

Parity status: PINNED by ``tests/test_oracle_golden.py`` against ``tests/golden/*.npz``
(generated from the real reference by ``tools/gen_golden.py``).

Reference lines restated: transfer/models.py:757-829 (constructor), :841-883 (attention),
:885-953 (forward); classifier/models.py:161-266 for the classifier flavour.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence


class TorchRefDiscriminator(nn.Module):
    def __init__(self, V, E, H, L=2, pad_id=0, dead_compute=True):
        super().__init__()
        self.H = H
        self.dead_compute = dead_compute  # the reference also evaluates two bridges nobody reads
        self.embedding = nn.Embedding(V, E, padding_idx=pad_id)
        self.rnn = nn.LSTM(input_size=E, hidden_size=H, num_layers=1, bidirectional=True)
        self.output_bridge = nn.Linear(2 * H, H)
        self.hidden_bridges = nn.ModuleList([nn.Linear(2 * H, H) for _ in range(2)])
        self.linear_emb = nn.Linear(E, H)
        self.linear_q = nn.Linear(H, H)
        self.linear_attn = nn.Linear(H, 1)
        self.output_layer = nn.Linear(H, L)

    def load_numpy(self, params):
        sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
        self.load_state_dict(sd)
        return self

    def forward(self, src, src_len, mask, soft=False, guard=True):
        """src [B,T,V] float or [B,T] int64; src_len CPU int tensor sorted desc; mask bool [B,T]
        (mutated in place when ``guard``, as the reference does)."""
        x = src.transpose(0, 1)
        emb = x @ self.embedding.weight if soft else self.embedding(x)
        packed = pack_padded_sequence(emb, src_len.cpu())
        outputs, (h_n, c_n) = self.rnn(packed)
        B = emb.shape[1]
        if self.dead_compute:
            outputs, _ = pad_packed_sequence(outputs)
            _ = self.output_bridge(outputs)
            _ = self.hidden_bridges[1](c_n.transpose(0, 1).reshape(B, 2 * self.H))
        k = self.linear_emb(emb).transpose(0, 1)
        q = self.hidden_bridges[0](h_n.transpose(0, 1).reshape(B, 2 * self.H))
        mid = torch.tanh(k + self.linear_q(q).unsqueeze(1))
        s = self.linear_attn(mid).squeeze(2)
        if guard:
            allm = mask.all(dim=1)
            if mask.shape[1] > 1:
                mask[allm, 1] = False
        s = s.masked_fill(mask, float("-inf"))
        a = F.softmax(s, dim=1)
        ctx = torch.bmm(a.unsqueeze(1), k).squeeze(1)
        return self.output_layer(ctx), a


def fwd_bwd_step(model, S, src_len, mask, labels):
    """One 'step' as bench.py defines it: forward on soft tokens, per-sentence CE vs labels,
    mean, backward to dS (frozen parameters).  Returns (loss, dS)."""
    S = S.detach().requires_grad_(True)
    logits, _ = model(S, src_len, mask.clone(), soft=True)
    loss = F.cross_entropy(logits, labels, reduction="none").mean()
    (dS,) = torch.autograd.grad(loss, S)
    return loss.detach(), dS
