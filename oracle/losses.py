"""NumPy restatement of the losses on the Polus hot path (test infrastructure only).

  * sparse softmax CE from logits, mean over all leading dims — Keras
    ``SparseCategoricalCrossentropy(from_logits=True)`` as used at
    tutorials/classifier_example.py:55 (third-party; parity unpinned).
  * polus/losses.py:5-18   class-weighted softmax CE
  * polus/losses.py:21-41  class-weighted sigmoid CE with negative-sample weight
  * polus/layers.py:58-63, 86-126  CRF transitions mask, NLL, sample-weighted NLL;
    the log-likelihood itself is tensorflow-addons ``crf_log_likelihood``
    (third-party, unpinned in requirements.txt:5): restated from its published
    algorithm — unary + binary path score minus the forward-algorithm log-norm,
    both masked by ``sequence_lengths``.  Parity unpinned.
"""
import numpy as np


def log_softmax(x):
    m = x.max(-1, keepdims=True)
    z = x - m
    return z - np.log(np.exp(z).sum(-1, keepdims=True))


def sparse_softmax_xent_fwd(logits, labels):
    """Returns (mean loss, dlogits)."""
    C = logits.shape[-1]
    lp = log_softmax(logits)
    flat = lp.reshape(-1, C)
    lab = labels.reshape(-1)
    n = flat.shape[0]
    loss = -flat[np.arange(n), lab].mean()
    d = np.exp(flat)
    d[np.arange(n), lab] -= 1.0
    d /= n
    return logits.dtype.type(loss), d.reshape(logits.shape).astype(logits.dtype)


def weighted_softmax_xent_fwd(class_weights, y_true, logits):
    """polus/losses.py:8-18; y_true one-hot [..., C]."""
    cw = np.asarray(class_weights, logits.dtype)
    w = (cw * y_true).sum(-1)
    lp = log_softmax(logits)
    unweighted = -(y_true * lp).sum(-1)
    n = unweighted.size
    loss = (unweighted * w).mean()
    ysum = y_true.sum(-1, keepdims=True)
    d = (np.exp(lp) * ysum - y_true) * w[..., None] / n
    return logits.dtype.type(loss), d.astype(logits.dtype)


def weighted_sigmoid_xent_fwd(class_weights, negative_weight, y_true, logits):
    """polus/losses.py:25-41."""
    cw = np.asarray(class_weights, logits.dtype)
    mask = np.all(y_true == 0, axis=-1)
    w = (cw * y_true).sum(-1) + mask.astype(logits.dtype) * negative_weight
    x = logits
    # tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log(1+exp(-|x|))
    per = np.maximum(x, 0) - x * y_true + np.log1p(np.exp(-np.abs(x)))
    unweighted = per.sum(-1)
    n = unweighted.size
    loss = (unweighted * w).mean()
    sig = 1.0 / (1.0 + np.exp(-x))
    d = (sig - y_true) * w[..., None] / n
    return logits.dtype.type(loss), d.astype(logits.dtype)


# ----------------------------------------------------------------------------- CRF

def crf_transitions(transitions, mask_impossible=None):
    """polus/layers.py:58-63."""
    if mask_impossible is None:
        return transitions
    m = np.asarray(mask_impossible, transitions.dtype)
    return transitions * m + ((1 - m).astype(np.int32) * -10000).astype(transitions.dtype)


def _lse(x, axis):
    m = x.max(axis, keepdims=True)
    return (m + np.log(np.exp(x - m).sum(axis, keepdims=True))).squeeze(axis)


def crf_log_likelihood(potentials, tags, lengths, trans):
    """Per-sample log-likelihood [B] and its gradient wrt potentials [B,S,C] and
    transitions [C,C] (gradient of sum_b ll_b)."""
    B, S, C = potentials.shape
    x = potentials.astype(np.float64)
    T = trans.astype(np.float64)
    ll = np.zeros(B)
    dx = np.zeros_like(x)
    dT = np.zeros_like(T)
    for b in range(B):
        L = int(lengths[b])
        if L <= 0:
            continue
        t = tags[b]
        score = x[b, np.arange(L), t[:L]].sum() + T[t[:L - 1], t[1:L]].sum()
        alpha = np.zeros((L, C))
        alpha[0] = x[b, 0]
        for s in range(1, L):
            alpha[s] = _lse(alpha[s - 1][:, None] + T, 0) + x[b, s]
        logz = _lse(alpha[L - 1], 0)
        ll[b] = score - logz
        beta = np.zeros((L, C))
        for s in range(L - 2, -1, -1):
            beta[s] = _lse(T + (x[b, s + 1] + beta[s + 1])[None, :], 1)
        marg = np.exp(alpha + beta - logz)
        dx[b, :L] -= marg
        dx[b, np.arange(L), t[:L]] += 1.0
        for s in range(L - 1):
            pair = np.exp(alpha[s][:, None] + T + (x[b, s + 1] + beta[s + 1])[None, :] - logz)
            dT -= pair
            dT[t[s], t[s + 1]] += 1.0
    return ll, dx, dT


def crf_nll_fwd(y_true_onehot, potentials, lengths, transitions, mask_impossible=None,
                sample_weights=None):
    """polus/layers.py:89-98 (and :106-124 with sample weights): -mean(ll * w).
    Returns loss, dpotentials, dtransitions (wrt the raw transitions weight)."""
    tags = y_true_onehot.argmax(-1).astype(np.int32)
    T = crf_transitions(transitions, mask_impossible)
    B = potentials.shape[0]
    w = np.ones(B) if sample_weights is None else np.asarray(sample_weights, np.float64)
    ll_b, _, _ = crf_log_likelihood(potentials, tags, lengths, T)
    loss = np.mean(-ll_b * w)
    dx = np.zeros(potentials.shape)
    dT = np.zeros(T.shape)
    for b in range(B):
        _, dxb, dTb = crf_log_likelihood(potentials[b:b + 1], tags[b:b + 1], lengths[b:b + 1], T)
        dx[b] = -dxb[0] * w[b] / B
        dT += -dTb * w[b] / B
    if mask_impossible is not None:
        dT = dT * np.asarray(mask_impossible, np.float64)
    dt = potentials.dtype
    return dt.type(loss), dx.astype(dt), dT.astype(dt)


def crf_sample_weights(y_true_onehot, mask_positive_classes, negative_weight):
    """polus/layers.py:113-119."""
    pos = y_true_onehot * np.asarray(mask_positive_classes, y_true_onehot.dtype)
    neg_mask = np.all(pos == 0, axis=(-2, -1))
    return np.any(pos == 1, axis=(-2, -1)).astype(np.float32) + neg_mask.astype(np.float32) * negative_weight


def crf_viterbi(potentials, lengths, trans):
    """tfa crf_decode (third-party, restated): best tag path; positions >= length are 0."""
    B, S, C = potentials.shape
    out = np.zeros((B, S), np.int32)
    for b in range(B):
        L = int(lengths[b])
        if L <= 0:
            continue
        score = potentials[b, 0].astype(np.float64)
        back = np.zeros((L, C), np.int32)
        for s in range(1, L):
            cand = score[:, None] + trans.astype(np.float64)
            back[s] = cand.argmax(0)
            score = cand.max(0) + potentials[b, s]
        best = int(score.argmax())
        out[b, L - 1] = best
        for s in range(L - 1, 0, -1):
            best = int(back[s, best])
            out[b, s - 1] = best
    return out
