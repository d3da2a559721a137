"""polus/schedulers.py:5-23 drop-in."""
from .optimizers import WarmUpLinearDecay, AdamWeightDecay  # noqa: F401 (the reference re-exports it too)


def warmup_scheduler(num_train_steps, max_lr, warmup_percentage=0.1, end_lr=1e-7):
    """Linear warm-up then linear decay to 1e-7 (``end_lr`` is ignored, as in the reference)."""
    return WarmUpLinearDecay(num_train_steps, max_lr, warmup_percentage, end_lr)
