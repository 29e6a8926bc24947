"""Losses.  `mse` restates the reference's `lightgrad/loss.py:4-12`: value is
`mean((y - y_hat)^2) / 2`, gradient is `(y - y_hat) * out_grad` - WITHOUT the 1/N
of the mean (reference quirk kept for parity; it makes the data-parallel
gradient of a concatenated batch the SUM of per-rank gradients).
`cross_entropy` restates loss.py:14-24 (softmax, pick the label's probability with an index pair, -log, mean; backward
`(softmax - onehot) / N * out_grad`); the generic form needs fancy indexing (CpuTensor has it), HipTensor provides it
as one fused kernel through the optional `_fused_cross_entropy` hook."""
from .autograd import Function


class mse(Function):
    """ Mean Squared Error """

    def forward(ctx, y, y_hat):
        fused = getattr(y, "_fused_mse", None)
        if fused is not None and y.shape == getattr(y_hat, "shape", None):
            # optional backend hook: the expression below in one kernel (same roundings at the scaling steps)
            loss, err = fused(y_hat)
            ctx.save_for_backward(err)
            return loss
        err = y - y_hat
        ctx.save_for_backward(err)
        return (err ** 2).mean() / 2

    def backward(ctx, out_grad):
        err, = ctx.get_saved_tensors()
        if getattr(out_grad, "_is_unit_constant", False):
            return err          # the loss is the root of backward(): its seed is the backend's constant 1.0, and x * 1.0 is x
        return err * out_grad


class cross_entropy(Function):
    """ Cross Entropy Loss of softmax(y) against integer class labels y_hat of shape (N,) """

    def forward(ctx, y, y_hat, axis: int = -1):
        fused = getattr(y, "_fused_cross_entropy", None)
        if fused is not None and len(y.shape) == 2 and axis in (-1, 1):
            loss, dlogits = fused(y_hat)                   # dlogits = (softmax - onehot) / N
            ctx.save_for_backward(dlogits, None, axis)
            return loss
        p = y.softmax(axis=axis)
        ctx.save_for_backward(p, y_hat, axis)
        n = y_hat.shape[0]
        return -p[range(n), y_hat].log().mean()

    def backward(ctx, out_grad):
        p, y_hat, axis = ctx.get_saved_tensors()
        if y_hat is None:
            return p if getattr(out_grad, "_is_unit_constant", False) else p * out_grad
        n = y_hat.shape[0]
        p[range(n), y_hat] -= 1
        p /= n
        return p * out_grad
