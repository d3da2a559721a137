"""polus.checkpoint -> polus_amd.checkpoint (re-export; no counterpart module in the reference: the Keras / HF objects it stood for came from TensorFlow)."""
from polus_amd import checkpoint as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
