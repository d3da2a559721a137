"""polus.ner.models -> polus_amd.ner.models (re-export)."""
from polus_amd.ner import models as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
