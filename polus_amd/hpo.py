"""polus/hpo.py:8-28: HPOContext singleton + parameter().  The optuna-driven HPO_Objective
(:30-146) is orchestration above train() and out of scope; the context
is kept because callbacks and model factories consult it."""
from .context import Singleton


class TrialPruned(Exception):
    pass


class HPOContext(metaclass=Singleton):
    def __init__(self):
        self.hpo_backend = None

    def is_hpo_enable(self):
        return self.hpo_backend is not None

    def add_hpo_backend(self, hpo_backend):
        self.hpo_backend = hpo_backend

    def reset(self):
        self.hpo_backend = None

    def prune(self, message=""):
        try:
            from optuna.exceptions import TrialPruned as _TP
        except ImportError:
            _TP = TrialPruned
        raise _TP(message)


def parameter(real_value, hpo_lambda):
    """polus/hpo.py:22-28: the real value without an HPO context, hpo_lambda(backend) with."""
    ctx = HPOContext()
    if ctx.is_hpo_enable():
        return hpo_lambda(ctx.hpo_backend)
    return real_value
