"""polus.ir.training -> polus_amd.ir.training (re-export)."""
from polus_amd.ir import training as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
