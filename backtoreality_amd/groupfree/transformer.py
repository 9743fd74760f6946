"""The decoder layer of GroupFree3D (detection/GroupFree3D/models/transformer.py:11-76):
post-norm self-attention over the query points, cross-attention onto the seed points, FFN;
position embeddings are added to queries and keys AND to the values (the reference passes
the position-augmented tensors as `value`).  `torch.nn.MultiheadAttention` is the module the
reference vendors a copy of (models/multi_head_attention.py), same parameter names; on the
GPU the whole layer is one library call (groupfree/fused_decoder.py -> csrc/decoder.hip); the
op-by-op form below (attention core through groupfree/fused_attention.py) is the CPU path and
what `BTR_FUSED_DECODER=0` selects."""
import torch.nn as nn
import torch.nn.functional as F

from . import fused_attention, fused_decoder


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu",
                 self_posembed=None, cross_posembed=None):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        if activation not in ("relu", "gelu", "glu"):
            raise RuntimeError("activation should be relu/gelu, not %s." % activation)
        self.activation = getattr(F, activation)
        self.self_posembed = self_posembed
        self.cross_posembed = cross_posembed

    def forward(self, query, key, query_pos, key_pos):
        """query (B,C,Pq), key (B,C,Pk), query_pos (B,Pq,3|6), key_pos (B,Pk,3) -> (B,C,Pq)."""
        q_pos = self.self_posembed(query_pos) if self.self_posembed is not None else None
        k_pos = self.cross_posembed(key_pos) if self.cross_posembed is not None else None
        # the whole layer as one library call (csrc/decoder.hip) when it covers the configuration
        out = fused_decoder.layer_forward(self, query, key, q_pos, k_pos)
        if out is not None:
            return out
        q_pos = q_pos.permute(2, 0, 1) if q_pos is not None else None
        k_pos = k_pos.permute(2, 0, 1) if k_pos is not None else None
        query = query.permute(2, 0, 1)
        key = key.permute(2, 0, 1)

        # attention core: the fused kernels (csrc/attention.hip) when they cover the call, else
        # the stock module (CPU, BTR_FUSED_ATTENTION=0)
        qp = query if q_pos is None else query + q_pos
        att = fused_attention.mha_forward(self.self_attn, qp, qp)
        if att is None:
            att = self.self_attn(qp, qp, value=qp)[0]
        query = self.norm1(query + self.dropout1(att))

        qp = query if q_pos is None else query + q_pos
        kp = key if k_pos is None else key + k_pos
        att = fused_attention.mha_forward(self.multihead_attn, qp, kp)
        if att is None:
            att = self.multihead_attn(query=qp, key=kp, value=kp)[0]
        query = self.norm2(query + self.dropout2(att))

        ffn = self.linear2(self.dropout(self.activation(self.linear1(query))))
        query = self.norm3(query + self.dropout3(ffn))
        return query.permute(1, 2, 0)
