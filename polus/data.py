"""polus.data -> polus_amd.data (re-export)."""
from polus_amd.data import *  # noqa: F401,F403
from polus_amd import data as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
