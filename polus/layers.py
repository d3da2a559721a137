"""polus.layers -> polus_amd.layers (re-export)."""
from polus_amd.layers import *  # noqa: F401,F403
from polus_amd import layers as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
