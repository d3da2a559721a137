"""polus.models -> polus_amd.models (re-export)."""
from polus_amd.models import *  # noqa: F401,F403
from polus_amd import models as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
