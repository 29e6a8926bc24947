"""Analytic-vs-numeric Jacobian check - the method by which every backend's
backward is validated in the reference (`lightgrad/autograd/utils/gradcheck.py:5-63`).

`jacobian` back-propagates one output element at a time (`y[j].backward()`),
`numerical_jacobian` uses central differences with a one-hot perturbation
built through `zeros` + `__setitem__`, so both exercise the backend's own
getitem/setitem/reshape kernels.
"""
import numpy as np
from ..tensor import AbstractTensor
from ..grads import Gradients


def jacobian(f, x: AbstractTensor) -> np.ndarray:
    assert isinstance(x, AbstractTensor) and x.requires_grad
    y = f(x)
    assert isinstance(y, AbstractTensor) and y.requires_grad
    n_in, n_out = x.numel(), y.numel()
    y = y.reshape(-1)
    J = np.empty((n_in, n_out), dtype=x.dtype)
    for j in range(n_out):
        y.zero_grad(traverse_graph=True)
        y[j].backward()
        J[:, j] = x.grad.reshape(-1).numpy()
    return J


@Gradients.no_grad()
def numerical_jacobian(f, x: AbstractTensor, eps=1e-4) -> np.ndarray:
    assert isinstance(x, AbstractTensor)
    y = f(x)
    assert isinstance(y, AbstractTensor)
    n_in, n_out = x.numel(), y.numel()
    NJ = np.empty((n_in, n_out), dtype=x.dtype)
    for i, idx in enumerate(np.ndindex(x.shape)):
        h = x.__class__.zeros(x.shape)
        h[idx] = eps
        y_hi = f(x + h).reshape(-1)
        y_lo = f(x - h).reshape(-1)
        NJ[i, :] = (y_hi - y_lo).numpy() / (2 * eps)
    return NJ


def gradcheck(f, x, eps=1e-3, atol=5e-4, rtol=5e-4) -> bool:
    return np.allclose(jacobian(f, x), numerical_jacobian(f, x, eps), atol=atol, rtol=rtol)


def assert_gradcheck(f, x, eps=1e-3, atol=5e-4, rtol=5e-4):
    return np.testing.assert_allclose(jacobian(f, x), numerical_jacobian(f, x, eps), atol=atol, rtol=rtol)
