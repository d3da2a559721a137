"""polus.callbacks -> polus_amd.callbacks (re-export)."""
from polus_amd.callbacks import *  # noqa: F401,F403
from polus_amd import callbacks as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
