"""Dual-encoder model for polus/ir/training.py: frozen BERT encoders (forward only, run in
`forward_without_grads`, :69-75) + trainable query/document projections (:82-83).

`encode_*` return the [CLS] hidden state (TFBertSplited's pooler_output convention,
polus/models.py:215-216).  One encoder may be shared by both towers."""
import torch

from ..layers import Dense
from ..models import BertModel, PolusModel, COMPUTE_DTYPES
from ..tensor import ParamArena, to_device


class DualEncoder(PolusModel):
    def __init__(self, query_encoder, document_encoder=None, projection_dim=128, compute_dtype="bf16", name="dual_encoder"):
        super().__init__(name)
        self.query_encoder = query_encoder
        self.document_encoder = document_encoder or query_encoder
        self.compute_dtype = COMPUTE_DTYPES[compute_dtype]
        H = query_encoder.config.hidden_size
        self.arena = ParamArena(self.compute_dtype)       # only the projections are trainable
        self.qp = Dense(projection_dim, name="query_projection")
        self.dp = Dense(projection_dim, name="document_projection")
        self.qp.build(self.arena, H, "query_projection")
        self.dp.build(self.arena, self.document_encoder.config.hidden_size, "document_projection")
        self.arena.finalize()

    def _cls(self, encoder, x, training):
        out = encoder(**x, training=False) if isinstance(x, dict) else encoder(x, training=False)
        return out.pooler_output.contiguous()            # [B, H], a fresh buffer (the encoder reuses its own)

    def encode_query(self, x, training=False):
        return self._cls(self.query_encoder, x, training)

    def encode_document(self, x, training=False):
        return self._cls(self.document_encoder, x, training)

    def query_projection(self, rep, training=False):
        return self.qp.forward(to_device(rep, self.compute_dtype, self.arena.device), training)

    def document_projection(self, rep, training=False):
        return self.dp.forward(to_device(rep, self.compute_dtype, self.arena.device), training)

    def backward_projections(self, dq, dd, accumulate=False):
        self.qp.backward(dq, accumulate, need_dx=False)
        self.dp.backward(dd, accumulate, need_dx=False)
        self._notify(self.dp.variables() + self.qp.variables())
