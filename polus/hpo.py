"""polus.hpo -> polus_amd.hpo (re-export)."""
from polus_amd.hpo import *  # noqa: F401,F403
from polus_amd import hpo as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
