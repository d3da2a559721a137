"""polus/core.py drop-in: the environment-flag helpers user scripts and tests touch.

`POLUS_JIT` (polus/core.py:34-56) selected XLA compilation in the reference (default False,
tests/test_core.py:5-11); here every kernel is ahead-of-time HIP, so the flag is kept for API
compatibility and read by nothing on the compute path."""
import os

import numpy as np


def set_jit_compile(mode: bool):
    os.environ["POLUS_JIT"] = str(mode)


def get_jit_compile():
    if os.environ.get("POLUS_JIT") is None:
        set_jit_compile(False)
    return os.environ.get("POLUS_JIT") == "True"


def find_dtype_and_shapes(data_generator, k=10):
    """polus/core.py:58-111: dtype and (possibly partly dynamic) shape of every key of the dict samples,
    from the first k samples (k = -1: all).  A dimension that differs between samples comes back None."""
    if k == -1:
        samples = list(data_generator)
    else:
        it = iter(data_generator)
        samples = [next(it) for _ in range(k)]
    if not isinstance(samples[0], dict):
        raise ValueError("The find_dtype_and_shapes only supports when the sample came from generator are dict "
                         f"but found {type(samples[0])}")
    dtypes, shapes = {}, {}
    for key, v in samples[0].items():
        a = np.asarray(v)
        dtypes[key], shapes[key] = a.dtype, list(a.shape)
    for prev, cur in zip(samples, samples[1:]):
        assert len(set(prev.keys()) - set(cur.keys())) == 0
        for key, v in cur.items():
            shp = np.asarray(v).shape
            assert len(shp) == len(shapes[key])
            shapes[key] = [None if (d is None or d != s) else d for d, s in zip(shapes[key], shp)]
    return dtypes, {k2: tuple(v) for k2, v in shapes.items()}


def execute_if(condition_var, error_message="", on=True):
    """polus/core.py:114-142: run the decorated method only while getattr(self, condition_var) == on."""
    def decorator(func):
        def function_wrapper(self, *args, **kwargs):
            if getattr(self, condition_var) == on:
                return func(self, *args, **kwargs)
            if error_message != "":
                print(error_message)
        return function_wrapper
    return decorator
