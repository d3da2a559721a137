"""polus.utils -> polus_amd.utils (re-export)."""
from polus_amd.utils import *  # noqa: F401,F403
from polus_amd import utils as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
