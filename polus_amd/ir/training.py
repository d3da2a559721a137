"""polus/ir/training.py drop-in: EfficientDenseRetrievalTrainer.

The two BERT encoders run in forward_without_grads (no gradient through BERT, :47-75); only
query_projection / document_projection and compute_scores are differentiated (:77-117).
Because there is no tape, `compute_scores` and the loss must be objects that can run
backward: `InBatchDotScores` + `ContrastiveLoss` below are the stock pair (in-batch negatives:
sample i's positive document is every other sample's negative)."""
import torch

from .. import ops
from ..tensor import DeviceScalar, to_device
from ..training import BaseTrainer


class EfficientDenseRetrievalTrainer(BaseTrainer):
    def __init__(self, model, compute_scores, k_negatives=0, trainable_weights=None, *args, **kwargs):
        self.compute_scores = compute_scores
        self.k_negatives = k_negatives
        self.trainable_weights = model.trainable_weights if trainable_weights is None else trainable_weights
        super().__init__(model, *args, **kwargs)

    def __str__(self):
        return "SimilarityTrainer"

    def forward_without_grads(self, question, positive_doc, negative_doc=None):
        q = self.model.encode_query(question, training=True)
        d = self.model.encode_document(positive_doc, training=True)
        if negative_doc is None:
            return q, d
        self.k_negatives = negative_doc["input_ids"].shape[1]
        negs = [self.model.encode_document({"input_ids": negative_doc["input_ids"][:, i, :],
                                            "attention_mask": negative_doc["attention_mask"][:, i, :]},
                                           training=True).clone() for i in range(self.k_negatives)]
        return q, d, torch.stack(negs, 0)

    def forward_with_grads(self, question, positive_doc, negative_doc=None):
        q = self.model.query_projection(question, training=True)
        d = self.model.document_projection(positive_doc, training=True)
        if self.post_process_logits is not None:
            q, d = self.post_process_logits(q), self.post_process_logits(d)
        if negative_doc is None:
            return self.compute_scores(q, d)
        negs = [self.model.document_projection(negative_doc[i], training=True) for i in range(self.k_negatives)]
        if self.post_process_logits is not None:
            negs = [self.post_process_logits(n) for n in negs]
        return self.compute_scores(q, d, *negs)

    def backward_from_loss(self, accumulate=False):
        dpos, dneg = self.loss.backward(accumulate)
        dq, dd = self.compute_scores.backward(dpos, dneg)
        self.model.backward_projections(dq, dd, accumulate=accumulate)


class InBatchDotScores:
    """scores = Q D^T [B,B]; positives on the diagonal, the rest of each row are negatives."""

    def __call__(self, q, d, *negs):
        assert not negs, "explicit negatives are scored by a user-supplied compute_scores"
        self.q, self.d = q.contiguous(), d.contiguous()
        B = q.shape[0]
        self.scores = torch.empty((B, B), dtype=torch.float32, device=q.device)
        ops.gemm(self.q, self.d, self.scores)
        return self.scores, None

    def backward(self, dscores, _):
        ds = dscores if dscores.dtype == self.q.dtype else dscores.to(self.q.dtype)
        dq, dd = torch.empty_like(self.q), torch.empty_like(self.d)
        ops.gemm(ds, self.d, dq, b_layout=ops.K_STRIDED)                       # dQ = dS D
        ops.gemm(ds, self.q, dd, a_layout=ops.K_STRIDED, b_layout=ops.K_STRIDED)  # dD = dS^T Q
        return dq, dd


class ContrastiveLoss:
    """softmax CE over each row of the in-batch score matrix with the diagonal as label."""

    def __call__(self, pos_scores, neg_scores=None):
        s = pos_scores
        B = s.shape[0]
        labels = torch.arange(B, dtype=torch.int32, device=s.device)
        self.loss = torch.empty(1, dtype=torch.float32, device=s.device)
        self.d = torch.empty_like(s)
        ops.softmax_xent(s, labels, self.loss, self.d)
        return DeviceScalar(self.loss)

    def backward(self, accumulate=False):
        return self.d, None
