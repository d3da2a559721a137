"""polus.training -> polus_amd.training (re-export)."""
from polus_amd.training import *  # noqa: F401,F403
from polus_amd import training as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
