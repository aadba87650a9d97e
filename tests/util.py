"""Shared helpers of the parity tests."""
import torch


def rel_err(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def assert_close(a, b, tol, what=""):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    e = rel_err(a, b)
    assert e <= tol, "{}: max-rel error {:.3e} > {:.1e} (|ref|max={:.3e})".format(what, e, tol, float(b.abs().max()))
    return e
