"""polus.ir.models -> polus_amd.ir.models (re-export)."""
from polus_amd.ir import models as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
