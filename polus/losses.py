"""polus.losses -> polus_amd.losses (re-export)."""
from polus_amd.losses import *  # noqa: F401,F403
from polus_amd import losses as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
