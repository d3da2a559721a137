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


def elem_err(actual, ref, rtol, atol_rms):
    """Worst per-element violation ratio of |a - r| <= rtol*|r| + atol_rms*rms(r) (<= 1 passes).
    Unlike `relerr` (one scale, max|ref|, for the whole tensor) small-magnitude elements are held to
    rtol of their own value plus a floor tied to the tensor's RMS, not to its largest entry."""
    ref = np.asarray(ref, np.float64)
    actual = np.asarray(actual, np.float64).reshape(ref.shape)
    nz = ref[ref != 0]                                   # sparse tensors (embedding-table gradients): RMS of the touched entries
    rms = float(np.sqrt(np.mean(nz ** 2))) + 1e-300 if nz.size else 1e-300
    bound = rtol * np.abs(ref) + atol_rms * rms
    return float((np.abs(actual - ref) / bound).max())


def assert_close_elem(actual, ref, rtol, atol_rms, what=""):
    e = elem_err(actual, ref, rtol, atol_rms)
    assert np.isfinite(e) and e <= 1.0, f"{what}: per-element error {e:.3f} x (rtol {rtol:.1e}, floor {atol_rms:.1e} rms)"


def cosine(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
