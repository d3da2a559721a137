"""polus/ir/training.py drop-in: EfficientDenseRetrievalTrainer.

The two BERT encoders run in forward_without_grads (no gradient through BERT, :47-75); only
query_projection / document_projection and compute_scores are differentiated (:77-117).
Because there is no tape, `compute_scores` and the loss must be objects that can run
backward: `InBatchDotScores` + `ContrastiveLoss` below are the stock pair.  With explicit negatives
(:59-67, :94-107) the positive and the k negative documents of a batch go through
`document_projection` as ONE [(k+1)B, H] call -- one stashed input, one backward -- and
`compute_scores.backward` returns (dq, dd, dnegs)."""
import torch

from .. import ops
from ..tensor import DeviceScalar, to_device
from ..training import BaseTrainer


def _adjacent_rows(first, rest):
    """True when `rest` [n, E] starts right after `first` [m, E] in one contiguous buffer."""
    return (first.is_contiguous() and rest.is_contiguous() and first.dtype == rest.dtype and
            rest.data_ptr() == first.data_ptr() + first.numel() * first.element_size())


class EfficientDenseRetrievalTrainer(BaseTrainer):
    def __init__(self, model, compute_scores, k_negatives=0, trainable_weights=None, *args, **kwargs):
        self.compute_scores = compute_scores
        self.k_negatives = k_negatives
        self.trainable_weights = model.trainable_weights if trainable_weights is None else trainable_weights
        super().__init__(model, *args, **kwargs)

    def __str__(self):
        return "SimilarityTrainer"

    def forward_without_grads(self, question, positive_doc, negative_doc=None):
        """polus/ir/training.py:47-75.  The document representations land in one [(1+k)B, H] buffer
        (positives first) so that forward_with_grads can project them in one call without a copy."""
        q = self.model.encode_query(question, training=True)
        if negative_doc is None:
            return q, self.model.encode_document(positive_doc, training=True)
        ids, am = negative_doc["input_ids"], negative_doc["attention_mask"]
        self.k_negatives = k = ids.shape[1]
        d = self.model.encode_document(positive_doc, training=True)
        B, H = d.shape
        reps = torch.empty((1 + k, B, H), dtype=d.dtype, device=d.device)
        reps[0].copy_(d)
        for i in range(k):
            reps[1 + i].copy_(self.model.encode_document({"input_ids": ids[:, i, :], "attention_mask": am[:, i, :]}, training=True))
        return q, reps[0], reps[1:]

    def forward_with_grads(self, question, positive_doc, negative_doc=None):
        """polus/ir/training.py:77-117."""
        q = self.model.query_projection(question, training=True)
        if negative_doc is None:
            d = self.model.document_projection(positive_doc, training=True)
            if self.post_process_logits is not None:
                q, d = self.post_process_logits(q), self.post_process_logits(d)
            self._n_neg = 0
            return self.compute_scores(q, d)
        negative_doc = to_device(negative_doc, None, q.device)
        positive_doc = to_device(positive_doc, None, q.device)
        k, B = negative_doc.shape[0], positive_doc.shape[0]
        self.k_negatives = self._n_neg = k
        flat_neg = negative_doc.reshape(k * B, -1)
        if _adjacent_rows(positive_doc, flat_neg):
            docs = torch.as_strided(positive_doc, ((k + 1) * B, positive_doc.shape[1]), (positive_doc.shape[1], 1))
        else:
            docs = torch.cat([positive_doc, flat_neg], 0)
        dall = self.model.document_projection(docs, training=True)        # [(k+1)B, E]: one stash, one backward
        d, negs = dall[:B], [dall[B * (i + 1):B * (i + 2)] for i in range(k)]
        if self.post_process_logits is not None:
            q, d = self.post_process_logits(q), self.post_process_logits(d)
            negs = [self.post_process_logits(n) for n in negs]
        return self.compute_scores(q, d, *negs)

    def backward_from_loss(self, accumulate=False):
        dpos, dneg = self.loss.backward(accumulate)
        out = self.compute_scores.backward(dpos, dneg)
        k = getattr(self, "_n_neg", 0)
        if k == 0:
            dq, dd = out[0], out[1]
        else:
            if len(out) < 3 or out[2] is None:
                raise ValueError("compute_scores.backward must return (dq, dd, dnegs) when it was called with negatives")
            dq, dd, dnegs = out
            dn = dnegs if torch.is_tensor(dnegs) else torch.stack(list(dnegs), 0)
            dn = dn.reshape(-1, dd.shape[-1])
            if _adjacent_rows(dd, dn):
                dd = torch.as_strided(dd, (dd.shape[0] + dn.shape[0], dd.shape[1]), (dd.shape[1], 1))
            else:
                dd = torch.cat([dd, dn], 0)
        self.model.backward_projections(dq, dd, accumulate=accumulate)


class InBatchDotScores:
    """Dot-product scores of every query against every document of the batch:
    pos_scores = Q D^T [B, B] (the positives on the diagonal, the other B-1 of a row are in-batch
    negatives); with explicit negatives N_1..N_k also neg_scores = Q [N_1; ..; N_k]^T [B, kB].  Both are
    column blocks of ONE [B, (k+1)B] matrix (one GEMM), which is what ContrastiveLoss takes its softmax over."""

    def __call__(self, q, d, *negs):
        self.q = q.contiguous()
        B, E = self.q.shape
        k = len(negs)
        if k:
            if all(_adjacent_rows(a.reshape(-1, E), b.reshape(-1, E)) for a, b in zip((d,) + negs[:-1], negs)):
                docs = torch.as_strided(d, ((k + 1) * B, E), (E, 1))
            else:
                docs = torch.cat([d] + [n.reshape(-1, E) for n in negs], 0)
        else:
            docs = d.contiguous()
        self.docs, self.k = docs, k
        self.scores = torch.empty((B, (k + 1) * B), dtype=torch.float32, device=q.device)
        ops.gemm(self.q, docs, self.scores)
        return (self.scores[:, :B], self.scores[:, B:]) if k else (self.scores, None)

    def backward(self, dpos, dneg):
        B, E = self.q.shape
        ds = dpos if self.k == 0 else _whole(dpos, dneg)
        ds = ds if ds.dtype == self.q.dtype else ds.to(self.q.dtype)
        dq, ddocs = torch.empty_like(self.q), torch.empty_like(self.docs)
        ops.gemm(ds, self.docs, dq, b_layout=ops.K_STRIDED)                         # dQ = dS Docs
        ops.gemm(ds, self.q, ddocs, a_layout=ops.K_STRIDED, b_layout=ops.K_STRIDED)  # dDocs = dS^T Q
        if self.k == 0:
            return dq, ddocs
        return dq, ddocs[:B], ddocs[B:].view(self.k, B, E)


def _whole(pos, neg):
    """The [B, (k+1)B] matrix whose column blocks `pos` and `neg` are (no copy), else their concatenation."""
    B = pos.shape[0]
    if (neg is not None and pos.stride(0) == neg.stride(0) and pos.stride(1) == 1 and neg.stride(1) == 1 and
            neg.data_ptr() == pos.data_ptr() + pos.shape[1] * pos.element_size() and
            pos.stride(0) == pos.shape[1] + neg.shape[1]):
        return torch.as_strided(pos, (B, pos.shape[1] + neg.shape[1]), (pos.stride(0), 1))
    return pos if neg is None else torch.cat([pos, neg], 1)


class ContrastiveLoss:
    """softmax cross-entropy over each query's row of scores [in-batch positives | explicit negatives]
    with the diagonal (the query's own positive document) as the label."""

    def __call__(self, pos_scores, neg_scores=None):
        s = _whole(pos_scores, neg_scores)
        B = s.shape[0]
        self._split = pos_scores.shape[1]
        labels = torch.arange(B, dtype=torch.int32, device=s.device)
        self.loss = torch.empty(1, dtype=torch.float32, device=s.device)
        self.d = torch.empty_like(s)
        ops.softmax_xent(s, labels, self.loss, self.d)
        return DeviceScalar(self.loss)

    def backward(self, accumulate=False):
        if self.d.shape[1] == self._split:
            return self.d, None
        return self.d[:, :self._split], self.d[:, self._split:]
