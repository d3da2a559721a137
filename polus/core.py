"""polus.core -> polus_amd.core (re-export)."""
from polus_amd.core import *  # noqa: F401,F403
from polus_amd import core as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
