"""Shared helpers of the parity tests.

``assert_close`` holds a tensor to three bars at once (round-2 verdict: a max-norm metric alone leaves small-magnitude
channels and gradients unchecked element-wise):
  max-norm   max|a-b| / max|b|                         <= tol
  RMS        ||a-b||_2 / ||b||_2                       <= tol / 2
  element    |a-b| <= tol * (|b| + rms_c(b))           for every element, rms_c = RMS of the element's own channel
                                                        (last axis) -- a small-magnitude channel is judged on its own scale
"""
import torch


def rel_err(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rms_rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


def elem_excess(a, b, tol):
    """max over elements of |a-b| / (tol * (|b| + rms of b's channel)); <= 1 passes."""
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    if b.ndim >= 2:
        rms_c = b.reshape(-1, b.shape[-1]).pow(2).mean(dim=0).sqrt()
        floor = 1e-3 * b.pow(2).mean().sqrt()          # an all-zero channel is judged on the tensor's scale / 1000
        rms_c = torch.maximum(rms_c, floor.expand_as(rms_c))
    else:
        rms_c = b.pow(2).mean().sqrt()
    bound = tol * (b.abs() + rms_c) + 1e-30
    return float(((a - b).abs() / bound).max())


def assert_close(a, b, tol, what="", elementwise=True):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    e = rel_err(a, b)
    assert e <= tol, "{}: max-rel error {:.3e} > {:.1e} (|ref|max={:.3e})".format(what, e, tol, float(b.abs().max()))
    r = rms_rel(a, b)
    assert r <= 0.5 * tol, "{}: RMS-rel error {:.3e} > {:.1e}".format(what, r, 0.5 * tol)
    if elementwise:
        x = elem_excess(a, b, tol)
        assert x <= 1.0, "{}: element-wise bound |a-b| <= {:.1e} * (|b| + rms_channel) exceeded {:.2f}x".format(what, tol, x)
    return e
