"""Shared helpers for the GPU parity tests (test infrastructure)."""
import numpy as np
import torch

TOL = {torch.float32: 2e-5, torch.bfloat16: 1.5e-2}


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def host(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


def rounded(a, dtype):
    """The values the device actually sees for an input array of `dtype`."""
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).float().numpy().astype(np.float64)


def relerr(actual, ref):
    ref = np.asarray(ref, np.float64)
    actual = np.asarray(actual, np.float64)
    denom = np.abs(ref).max() + 1e-30
    return float(np.abs(actual - ref).max() / denom)


def assert_close(actual, ref, tol, what=""):
    e = relerr(actual, ref)
    assert np.isfinite(e) and e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
