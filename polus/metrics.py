"""polus.metrics -> polus_amd.metrics (re-export)."""
from polus_amd.metrics import *  # noqa: F401,F403
from polus_amd import metrics as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
