"""Optimizers written as tensor expressions, backend-agnostic.

Restates the reference's `lightgrad/optim.py`: `step` adds `compute_delta(grad, i)`
to each parameter in place under no_grad (optim.py:10-13); SGD with momentum
(:15-25); Adam (:27-41) and AdaBelief (:43-52).  Note the reference quirk that
is kept on purpose: `self.t` advances once per *parameter*, not once per step
(optim.py:36, :48), so bias correction differs between parameters of one step.

The expression form is the definition and runs on every backend.  Optional
backend hooks (HipTensor implements them) replace it by
kernels that evaluate the SAME expression sequence per element:
  fused=True        one kernel per parameter instead of ~14 elementwise launches
  device_step=True  the step number behind the bias corrections lives in device
                    memory, so the whole training step can be captured in a
                    hipGraph and replayed (autograd/hip/graph.py)
  use_flat_buckets  parameters / gradients / moments live in flat buckets
                    (dist.DataParallel(flatten=True)): zero_grad is one fill and
                    the update of ALL parameters is one launch
"""
from .autograd import Gradients, AbstractTensor


class Optimizer(object):
    """A tuple of parameters and the rule that turns a gradient into an additive update.

    Subclasses implement `compute_delta(grad, index)`; `step` adds the result to parameter `index` in place (the `+=` of
    the parameter's backend, so views handed out earlier keep seeing the parameter)."""

    def __init__(self, parameters) -> None:
        self.parameters = tuple(parameters)
        strangers = [type(p).__name__ for p in self.parameters if not isinstance(p, AbstractTensor)]
        assert not strangers, "optimizers update tensors, got %s" % strangers
        self._flat_grad = None          # the gradient bucket, once a flat-bucket backend took over (use_flat_buckets)

    def zero_grad(self) -> None:
        if self._flat_grad is None:
            for p in self.parameters:
                p.zero_grad()
            return
        # every p.grad is a view into the bucket, written by the first backward kernel that reaches it
        for p in self.parameters:
            p._grad_zero_pending = True

    def compute_delta(self, grad: AbstractTensor, idx: int) -> AbstractTensor:
        raise NotImplementedError("%s does not say how a gradient becomes an update" % type(self).__name__)

    def step(self) -> None:
        with Gradients.no_grad():
            for index, parameter in enumerate(self.parameters):
                parameter += self.compute_delta(parameter.grad, index)


class SGD(Optimizer):
    """gradient descent with (heavy-ball) momentum: the update is -lr * grad plus `momentum` times the previous update"""

    def __init__(self, parameters, lr: float, momentum: float = 0.0):
        Optimizer.__init__(self, parameters)
        self.lr = lr
        self.momentum = momentum
        self.last_update = {}           # parameter index -> the update applied last (missing: none yet)

    def compute_delta(self, grad, i):
        update = -self.lr * grad + self.momentum * self.last_update.get(i, 0)
        self.last_update[i] = update
        return update


class Adam(Optimizer):
    """Adam (Kingma & Ba): running means of the gradient (`m`) and of its square (`v`), both bias-corrected by the number of
    updates `t` - which, as in the reference (optim.py:36), counts every PARAMETER's update, not every step."""
    belief = False

    def __init__(self, parameters, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                 fused: bool = False, grad_scale: float = 1.0, device_step: bool = False):
        Optimizer.__init__(self, parameters)
        self.lr, self.eps = lr, eps
        self.b1, self.b2 = beta1, beta2
        count = len(self.parameters)
        self.t, self.m, self.v = 0, [0] * count, [0] * count
        # grad_scale: factor applied to every gradient first (1/world_size in data-parallel training)
        self.fused, self.grad_scale, self.device_step = fused, grad_scale, device_step
        self._step_counter = None
        self._flat = None               # (flat_params, flat_m, flat_v, offsets) once use_flat_buckets() was called
        self._peer_exchange = None
        self._backward_update = None    # fuse_update_into_backward(): the backend object that arms / finishes every step
        self.last_step_launched_update = None

    def second_moment_input(self, grad, m):
        return grad

    def compute_delta(self, grad, i):
        self.t += 1
        self.m[i] = self.b1 * self.m[i] + (1 - self.b1) * grad
        self.v[i] = self.b2 * self.v[i] + (1 - self.b2) * self.second_moment_input(grad, self.m[i])**2
        m, v = self.m[i] / (1 - self.b1**self.t), self.v[i] / (1 - self.b2**self.t)
        return -self.lr * m / (v**0.5 + self.eps)

    def use_flat_buckets(self, flat_params, flat_grads, offsets) -> None:
        """called by dist.DataParallel(flatten=True).attach(optimizer): parameter i is
        flat_params[offsets[i]:offsets[i+1]] and its gradient the same slice of flat_grads"""
        assert self.t == 0, "switch to flat buckets before the first step"
        assert self.fused and self.device_step and hasattr(flat_params, "_fused_adam_multi_dev"), \
            "flat buckets need fused=True, device_step=True and a backend with a multi-tensor kernel"
        cls = flat_params.__class__
        self._flat_grad = flat_grads
        self._flat = (flat_params, cls.zeros(flat_params.shape, requires_grad=False),
                      cls.zeros(flat_params.shape, requires_grad=False), tuple(int(o) for o in offsets))
        # the update launch advances the device step number itself: one private copy per workgroup of its grid
        longest = max(b - a for a, b in zip(self._flat[3][:-1], self._flat[3][1:]))
        self._step_counter = flat_params._new_step_counter(0, slots=max(1, len(self.parameters) * -(-longest // 1024)))

    def use_peer_exchange(self, comm) -> None:
        """data parallel with a communicator whose exchange rides in the optimizer launch (dist.PeerWindowCommunicator):
        step() first sums the flat gradient bucket over the ranks, in the same kernel"""
        assert self._flat is not None and hasattr(self._flat[0], "_fused_adam_multi_p2p"), "use_flat_buckets() first"
        assert self.t == 0, "switch to the exchange inside the optimizer launch before the first step"
        self._peer_exchange = comm
        offsets = self._flat[3]
        chunks = sum(-(-(b - a) // 1024) for a, b in zip(offsets[:-1], offsets[1:]))
        self._step_counter = self._flat[0]._new_step_counter(0, slots=chunks)     # one private copy of the step per exchange workgroup

    def fuse_update_into_backward(self) -> None:
        """let the kernels that PRODUCE the parameter gradients apply this optimizer's update to them (HipTensor: the weight
        gradient GEMM's epilogue, its bias row sums, the skinny-head backward) - `step()` then launches nothing when every
        gradient was produced that way, and one kernel for the rest otherwise.  Same values as the update launch, bit for bit.
        Needs flat buckets; not with a data-parallel exchange (the update needs the SUMMED gradient); every parameter's
        gradient must be written by one kernel per step (no weight shared between layers: a second writer raises).  The
        parameters alternate between two buckets: `zero_grad(); backward(); step()` in that order, one backward per step, and
        an even number of steps inside a captured hipGraph (lightgrad_amd/autograd/hip/tensor.py: BackwardUpdate)."""
        assert self._flat is not None and hasattr(self._flat[0], "_new_backward_update"), "use_flat_buckets() first (dist.DataParallel(flatten=True).attach)"
        assert self._peer_exchange is None, "the update cannot ride in the backward kernels when the gradients are exchanged first"
        assert self.t % max(1, len(self.parameters)) == 0
        flat_p, flat_m, flat_v, offsets = self._flat
        self._backward_update = flat_p._new_backward_update(self.parameters, self._flat_grad, flat_m, flat_v, offsets, self.lr, self.b1, self.b2,
                                                            self.eps, self.grad_scale, self.belief, steps_done=self.t // max(1, len(self.parameters)))

    def zero_grad(self) -> None:
        Optimizer.zero_grad(self)
        if self._backward_update is not None:
            self._backward_update.arm()

    @Gradients.no_grad()
    def step(self) -> None:
        n_params = len(self.parameters)
        if self._backward_update is not None:
            for p in self.parameters:
                p._materialize_zero_grad()        # parameters no gradient reached since zero_grad
            _, here = self._backward_update.finish()
            self.last_step_launched_update = here > 0
            self.t += n_params
            return
        if self._flat is not None:
            for p in self.parameters:
                p._materialize_zero_grad()        # parameters no gradient reached since zero_grad
            flat_p, flat_m, flat_v, offsets = self._flat
            update = flat_p._fused_adam_multi_dev if self._peer_exchange is None else flat_p._fused_adam_multi_p2p
            update(self._flat_grad, flat_m, flat_v, offsets, self.lr, self.b1, self.b2, self.eps,
                                         self._step_counter, self.grad_scale, self.belief)   # advances the device counter too
            self.t += n_params
            return
        for i, p in enumerate(self.parameters):
            kernel = getattr(p, "_fused_adam_step", None) if self.fused else None
            if kernel is None:
                g = p.grad if self.grad_scale == 1.0 else p.grad * self.grad_scale
                p += self.compute_delta(g, i)
                continue
            self.t += 1
            if not isinstance(self.m[i], AbstractTensor):
                self.m[i] = p.__class__.zeros(p.shape, requires_grad=False)
                self.v[i] = p.__class__.zeros(p.shape, requires_grad=False)
            if self.device_step:
                if self._step_counter is None:
                    self._step_counter = p._new_step_counter((self.t - 1) // n_params)
                p._fused_adam_step_dev(p.grad, self.m[i], self.v[i], self.lr, self.b1, self.b2, self.eps,
                                       self._step_counter, n_params, i + 1, self.grad_scale, self.belief)
            else:
                kernel(p.grad, self.m[i], self.v[i], self.lr, self.b1, self.b2, self.eps,
                       (1 - self.b1**self.t) ** -1, (1 - self.b2**self.t) ** -1, self.grad_scale, self.belief)
        if self._step_counter is not None:
            self.parameters[0]._advance_step_counter(self._step_counter)      # these kernels only read the counter: one tiny launch

    def on_graph_replay(self, n: int = 1) -> None:
        """keep the host-side step count in line after `n` replays of a captured step"""
        self.t += n * len(self.parameters)


class AdaBelief(Adam):
    """AdaBelief (arXiv:2010.07468): Adam whose second running mean follows (grad - m)^2, the squared surprise, instead
    of grad^2"""
    belief = True

    def second_moment_input(self, grad, m):
        return grad - m
