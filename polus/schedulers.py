"""polus.schedulers -> polus_amd.schedulers (re-export)."""
from polus_amd.schedulers import *  # noqa: F401,F403
from polus_amd import schedulers as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
