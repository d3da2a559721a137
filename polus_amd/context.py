"""PolusContext (polus/__init__.py:102-127): decides whether data parallelism is on and pins
this process to its GPU.  The reference turns DP on when >1 GPU is visible, Horovod imports
and hvd.size() > 1; here: when torchrun's WORLD_SIZE > 1 (one process per GPU)."""
import logging
import os

from . import comm

logger = logging.getLogger("polus_amd")
if not logger.handlers:
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter("%(asctime)s - polus_amd - %(levelname)s - %(message)s"))
    logger.addHandler(_h)
    logger.setLevel(getattr(logging, os.environ.get("POLUS_LOGGER_LEVEL", "WARNING").upper(), logging.WARNING))


class Singleton(type):
    """polus/utils.py:84-96."""
    _instances = {}

    def __call__(cls, *args, **kwargs):
        if cls not in cls._instances:
            cls._instances[cls] = super().__call__(*args, **kwargs)
        return cls._instances[cls]


class PolusContext(metaclass=Singleton):
    def __init__(self):
        self.use_horovod = False
        self.backend = comm.init()
        if comm.size() > 1:
            self.use_horovod = True
            logger.info(f"MultiGPU training enabled: rank {comm.rank()} of {comm.size()} ({self.backend})")
        else:
            logger.info("Single process / single GPU training")

    def is_horovod_enabled(self):
        return self.use_horovod

    @classmethod
    def reset(cls):
        """Test hook: forget the singleton (and the process group)."""
        Singleton._instances.pop(cls, None)
        comm.shutdown()
