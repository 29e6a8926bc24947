"""Gradient-mode switch and the backward graph walk.

Mirrors the reference interface `lightgrad/autograd/grads.py:4-42`
(`Gradients.disable/enable/_is_enabled/no_grad/backward`).

Deliberate divergence: the reference pops nodes LIFO from an
OrderedDict (grads.py:36), which double-counts when a non-leaf tensor feeds two
consumers in the "wrong" insertion order. Here the walk is a reverse post-order
(topological) traversal.  For chains, trees and re-used *leaf* tensors - every
graph of BASELINE configs #1-#4 - the visiting order is exactly the
reference's (last parent first), so results are identical there; on diamonds
this walk is the mathematically correct one.
"""
from functools import wraps


class _NoGrad(object):
    """Context manager *and* decorator, like the reference's handler (grads.py:7-14)."""

    def __enter__(self, *args):
        Gradients.disable()

    def __exit__(self, *args):
        Gradients.enable()

    def __call__(self, fn):
        @wraps(fn)
        def guarded(*args, **kwargs):
            Gradients.disable()
            try:
                return fn(*args, **kwargs)
            finally:
                Gradients.enable()
        return guarded


class Gradients(object):

    # depth counter: gradients are recorded only while it is zero (grads.py:6)
    _disable_depth = 0

    @staticmethod
    def disable():
        Gradients._disable_depth += 1

    @staticmethod
    def enable():
        Gradients._disable_depth = max(0, Gradients._disable_depth - 1)

    @staticmethod
    def _is_enabled() -> bool:
        return Gradients._disable_depth == 0

    @staticmethod
    def no_grad():
        return _NoGrad()

    @staticmethod
    def _schedule(root_ctx):
        """Reverse post-order of the ctx graph reachable from `root_ctx`.

        Parents are expanded in positional order, so for a tree the reversed
        post-order visits the *last* parent's subtree first - the same order
        the reference's LIFO queue produces (grads.py:36-42).
        """
        order, seen = [], {id(root_ctx)}
        stack = [(root_ctx, iter([t.ctx for t in root_ctx.parent_tensors if t.ctx is not None]))]
        while stack:
            node, it = stack[-1]
            advanced = False
            for parent_ctx in it:
                if id(parent_ctx) not in seen:
                    seen.add(id(parent_ctx))
                    stack.append((parent_ctx, iter([t.ctx for t in parent_ctx.parent_tensors if t.ctx is not None])))
                    advanced = True
                    break
            if not advanced:
                order.append(node)
                stack.pop()
        order.reverse()
        return order

    @staticmethod
    def backward(ctx, grad):
        """Propagate `grad` (gradient of the tensor produced by `ctx`) to all ancestors."""
        out_grads = {id(ctx): grad}
        order = Gradients._schedule(ctx)
        # a backend may bracket the whole pass (HipTensor: parameter-gradient kernels on a second stream, joined at the end)
        begin = getattr(grad, "_backward_pass_begins", None)
        finish = begin(len(order)) if begin is not None else None
        try:
            for node in order:
                out_grad = out_grads.pop(id(node), None)
                if out_grad is None:
                    # reachable only through tensors that do not require gradients
                    continue
                Gradients.disable()
                try:
                    node._backpropagate(out_grad)
                finally:
                    Gradients.enable()
                # the accumulated .grad of each parent is the out-grad of the node that made it
                for t in node.parent_tensors:
                    if t.ctx is not None and t.grad is not None:
                        out_grads[id(t.ctx)] = t.grad
        finally:
            if finish is not None:
                finish()
