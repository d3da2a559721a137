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


def dropout_keep_np(seed, drop_p, idx0, n):
    """numpy restatement of the engine's counter-based dropout mask (polus_amd/csrc/common.h polus_keep): element idx
    takes 16-bit field (idx & 3) of the pair (h1, h2), h1 = murmur3 finaliser of (idx >> 2) * 0x9E3779B1 + seed,
    h2 = xs15(h1 * 0x27D4EB2F); kept when the field is >= round(p * 65536)."""
    U, M = np.uint64, np.uint64(0xFFFFFFFF)
    idx = (np.arange(n, dtype=np.uint64) + U(idx0)) & M
    x = ((idx >> U(2)) * U(0x9E3779B1) + U(int(seed) & 0xFFFFFFFF)) & M
    x ^= x >> U(16); x = (x * U(0x85EBCA6B)) & M
    x ^= x >> U(13); x = (x * U(0xC2B2AE35)) & M
    h1 = x ^ (x >> U(16))
    y = (h1 * U(0x27D4EB2F)) & M
    h2 = y ^ (y >> U(15))
    h = np.where((idx & U(2)) != 0, h2, h1)
    field = np.where((idx & U(1)) != 0, h >> U(16), h & U(0xFFFF))
    thresh = min(max(int(drop_p * 65536.0 + 0.5), 0), 65535)
    return (field >= U(thresh)).astype(np.uint8)
