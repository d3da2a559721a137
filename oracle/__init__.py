"""CPU oracle for the Polus data-parallel training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``polus_amd/`` may import this package:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker / reported baseline.

Parity status: the reference (bioinformatics-ua/polus @ 0.2.1) delegates the
arithmetic of this path to third-party wheels that are not vendored and not
installed here (tensorflow >=2.6, transformers TF-BERT, tensorflow-addons CRF,
horovod 0.24.2), and its own tests hold no numeric golden vectors for it
(tests/utils.py:3-5 is a vacuous one-sided cosine check).  The BERT
forward/backward restated here is pinned instead against the PyTorch twin of the
model the reference loads with ``from_pt=True`` (polus/models.py:225-229):
``transformers`` 5.15.0 ``BertModel`` on CPU, eager attention, dropout 0 —
see ``tests/golden/make_golden.py``.  CRF / Keras-Adam / HF AdamWeightDecay /
WarmUp are restated from their published algorithms: "parity unpinned" for
those (no reference fixture covers them).
"""
