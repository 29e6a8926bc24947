"""Analytic-vs-numeric Jacobian check - the method by which every backend's backward is validated in the reference
(same four entry points and default tolerances as `lightgrad/autograd/utils/gradcheck.py:5-63`).

Both Jacobians have one ROW per input element and one COLUMN per output element.  The analytic one is assembled
column by column: back-propagating output element j alone (`y[j].backward()` after clearing every gradient of the graph)
leaves d y[j] / d x in `x.grad`.  The numeric one is assembled row by row from central differences, the one-hot
perturbation being built with the backend's own `zeros` + `__setitem__` - so a check exercises the backend's getitem /
setitem / reshape kernels as well as the op under test.
"""
import numpy as np
from ..tensor import AbstractTensor
from ..grads import Gradients


def _flat_output(f, x, needs_grad):
    assert isinstance(x, AbstractTensor) and (x.requires_grad or not needs_grad), "gradcheck works on one tensor argument"
    y = f(x)
    assert isinstance(y, AbstractTensor) and (y.requires_grad or not needs_grad), "the function must return a tensor of the tape"
    return y.reshape(-1)


def _analytic_columns(y_flat, x):
    for j in range(y_flat.numel()):
        y_flat.zero_grad(traverse_graph=True)
        y_flat[j].backward()
        yield x.grad.reshape(-1).numpy()


def _numeric_rows(f, x, eps):
    for position in np.ndindex(x.shape):
        bump = type(x).zeros(x.shape)
        bump[position] = eps
        above, below = f(x + bump).reshape(-1), f(x - bump).reshape(-1)
        yield (above - below).numpy() / (2 * eps)


def jacobian(f, x: AbstractTensor) -> np.ndarray:
    y_flat = _flat_output(f, x, needs_grad=True)
    table = np.empty((x.numel(), y_flat.numel()), dtype=x.dtype)
    for j, column in enumerate(_analytic_columns(y_flat, x)):
        table[:, j] = column
    return table


@Gradients.no_grad()
def numerical_jacobian(f, x: AbstractTensor, eps=1e-4) -> np.ndarray:
    n_out = _flat_output(f, x, needs_grad=False).numel()
    table = np.empty((x.numel(), n_out), dtype=x.dtype)
    for i, row in enumerate(_numeric_rows(f, x, eps)):
        table[i, :] = row
    return table


def _both(f, x, eps):
    return jacobian(f, x), numerical_jacobian(f, x, eps)


def gradcheck(f, x, eps=1e-3, atol=5e-4, rtol=5e-4) -> bool:
    analytic, numeric = _both(f, x, eps)
    return bool(np.allclose(analytic, numeric, atol=atol, rtol=rtol))


def assert_gradcheck(f, x, eps=1e-3, atol=5e-4, rtol=5e-4):
    analytic, numeric = _both(f, x, eps)
    return np.testing.assert_allclose(analytic, numeric, atol=atol, rtol=rtol)
