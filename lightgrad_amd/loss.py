"""Losses.  `mse` restates the reference's `lightgrad/loss.py:4-12`: value is
`mean((y - y_hat)^2) / 2`, gradient is `(y - y_hat) * out_grad` - WITHOUT the 1/N
of the mean (reference quirk kept for parity; it makes the data-parallel
gradient of a concatenated batch the SUM of per-rank gradients).
`cross_entropy` (loss.py:14-24) needs fancy indexing and is out of scope."""
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
        return err * out_grad
