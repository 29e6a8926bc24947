"""Data-parallel training: one process per GPU, ONE gradient all-reduce per step.

The reference has no distributed code; this module is the MI355X-native
addition for BASELINE config #4.  Design:
  * every rank holds a full replica and draws its own batch;
  * parameter gradients are VIEWS into one flat bucket (MLP: 407 050 fp32 = 1.63 MB), so the
    exchange is a single in-place all-reduce with no pack/unpack kernels.  Over xGMI a message
    of this size is latency-bound, hence exactly one collective per step, enqueued on the
    compute stream between backward() and optim.step() with no host synchronisation;
  * `mse.backward` carries no 1/N (loss.py:12), so the all-reduce SUM equals the gradient of the
    concatenated batch; the mean (x 1/world) is folded into the optimizer (`grad_scale`).

When a rank is lost (the peer-window forms).  A wait inside an exchange launch gives up after 20 s; the next synchronising call
raises HipError (LG_ECOMM) naming the chunk, the rank and the exchange it waited for.  From then on `comm.failed()` is True: the
gradient bucket and the parameters of this rank are NOT to be trusted (the step that failed may have applied a partial sum), every
further collective on the communicator is refused, graphs that recorded its launches report again when replayed.  What a training
loop can do: `comm.close()` (returns at once, no barrier with the lost peer), then either end the job (bench.py: exit code 3) or,
with the surviving ranks renumbered by the launcher, open a new communicator (`open_communicators`), build a new DataParallel
around the model - its constructor broadcasts rank 0's parameters, which are from the last COMPLETED step on any rank that raised
before its own update ran - and re-capture any hipGraph that held an exchange.  Optimizer moments are per rank and stay valid.

Communicators with the same interface:
  PeerWindowCommunicator  HipTensor buckets, hand-written exchange through peer-mapped device memory (include/lghip_p2p.h):
                    ordinary kernels, and the gradient exchange fused into the optimizer launch
  RcclCommunicator  HipTensor buckets, RCCL through liblghip_comm.so (include/lghip_comm.h)
  GlooCommunicator  CpuTensor buckets, torch.distributed/gloo - lets the bucket / scaling /
                    determinism logic be tested with world_size 2 on a CPU-only machine
"""
import ctypes
import os
import time
import numpy as np

from .autograd import CpuTensor, HipTensor, Gradients


class Communicator(object):
    rank, world_size = 0, 1

    def allreduce_sum_(self, flat):
        raise NotImplementedError()

    def allreduce_max_(self, flat):
        raise NotImplementedError()

    def broadcast_(self, flat, root=0):
        raise NotImplementedError()

    def barrier(self):
        raise NotImplementedError()

    # overlap protocol (DataParallel(overlap=True)): `fork` lets the communicator's own stream wait for the gradient
    # kernels enqueued so far, `allreduce_sum_(flat, forked=True)` runs there, `join` makes the compute stream wait
    # for it.  Communicators without streams (host collectives) run the collective at once and ignore fork / join.
    def fork(self):
        pass

    def join(self):
        pass

    def close(self):
        pass


class SingleProcess(Communicator):
    """world_size 1: every collective is the identity"""

    def allreduce_sum_(self, flat, forked=False):
        return flat

    def allreduce_max_(self, flat):
        return flat

    def broadcast_(self, flat, root=0):
        return flat

    def barrier(self):
        pass


class GlooCommunicator(Communicator):
    """CPU ranks over torch.distributed (gloo); the process group must already be initialised"""

    def __init__(self):
        import torch.distributed as dist
        assert dist.is_initialized(), "call torch.distributed.init_process_group('gloo', ...) first"
        self._dist = dist
        self.rank, self.world_size = dist.get_rank(), dist.get_world_size()

    def _tensor(self, flat):
        import torch
        assert isinstance(flat, CpuTensor) and flat.data.flags["C_CONTIGUOUS"]
        return torch.from_numpy(flat.data)     # shares memory: the reduction lands in the bucket

    def allreduce_sum_(self, flat, forked=False):
        self._dist.all_reduce(self._tensor(flat), op=self._dist.ReduceOp.SUM)
        return flat

    def allreduce_max_(self, flat):
        self._dist.all_reduce(self._tensor(flat), op=self._dist.ReduceOp.MAX)
        return flat

    def broadcast_(self, flat, root=0):
        self._dist.broadcast(self._tensor(flat), src=root)
        return flat

    def barrier(self):
        self._dist.barrier()


class HostStagedCommunicator(GlooCommunicator):
    """GPU ranks whose collectives travel through host memory: device -> host copy, gloo, host -> device copy.

    Not a fast path and not used by bench.py - RCCL is (RcclCommunicator).  It exists so that everything ABOVE the collective
    (flat HipTensor bucket, gradient hooks, the fused multi-tensor optimizer with grad_scale = 1 / world, replica identity) can
    be exercised with world_size > 1 on a box with ONE GPU, where RCCL refuses a second rank on the same device
    (tests/test_hip_dist.py).  Every call synchronises the stream; nothing here can be captured into a hipGraph."""

    def _staged(self, flat, reduce):
        assert not isinstance(flat, CpuTensor), "HostStagedCommunicator is for device tensors; CPU ranks use GlooCommunicator"
        assert flat.is_contiguous(), "collectives run on dense buckets"
        import torch
        host = np.ascontiguousarray(flat.numpy())                 # device -> host (synchronises)
        reduce(torch.from_numpy(host))
        flat.upload_(host)                                        # host -> device, ordered on the stream
        return flat

    def allreduce_sum_(self, flat, forked=False):
        return self._staged(flat, lambda t: self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM))

    def allreduce_max_(self, flat):
        return self._staged(flat, lambda t: self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX))

    def broadcast_(self, flat, root=0):
        return self._staged(flat, lambda t: self._dist.broadcast(t, src=root))


def _exchange_unique_id(rank, make_id, path, timeout=300.0):
    """rank 0 writes the 128-byte RCCL id to `path` atomically, the others poll for it (single node)"""
    if rank == 0:
        blob = make_id()
        tmp = "%s.tmp.%d" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(blob)
        os.replace(tmp, path)
        return blob
    deadline = time.time() + timeout
    while time.time() < deadline:
        try:
            with open(path, "rb") as f:
                blob = f.read()
            if len(blob) == 128:
                return blob
        except FileNotFoundError:
            pass
        time.sleep(0.01)
    raise TimeoutError("rank %d: no RCCL id at %s after %.0f s" % (rank, path, timeout))


def _exchange_blobs(rank, world, blob, prefix, timeout=300.0, accept=None):
    """every rank publishes `blob` as <prefix>.<rank> (atomic rename) and reads the others'; returns the list by rank.
    accept(data) -> bool: a file that fails it counts as not yet written (a leftover of an earlier attempt under the same name)"""
    tmp = "%s.%d.tmp.%d" % (prefix, rank, os.getpid())
    with open(tmp, "wb") as f:
        f.write(blob)
    os.replace(tmp, "%s.%d" % (prefix, rank))
    out, deadline = [], time.time() + timeout
    for r in range(world):
        while True:
            try:
                with open("%s.%d" % (prefix, r), "rb") as f:
                    data = f.read()
                if len(data) == len(blob) and (accept is None or accept(data)):
                    out.append(data)
                    break
            except FileNotFoundError:
                pass
            if time.time() > deadline:
                raise TimeoutError("rank %d: nothing from rank %d at %s.%d after %.0f s" % (rank, r, prefix, r, timeout))
            time.sleep(0.005)
    return out


def _job_rendezvous_path():
    """a path unique to this job that every rank of it derives alike (lightgrad_amd.launch sets it; torch.distributed.run: its pid)"""
    path = os.environ.get("LIGHTGRAD_RCCL_ID_FILE")
    if path is None:
        tag = "%s_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "norun"), os.getppid())
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "lightgrad_rccl_%s.id" % tag)
    return path


class _c_stdout_to_stderr(object):
    """redirect file descriptor 1 to file descriptor 2 for the duration of a `with` block (C libraries included)"""

    @staticmethod
    def _flush_all():
        import sys
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)      # the C library's own buffers: a banner printf()ed into a pipe sits there until exit
        except (OSError, AttributeError):
            pass

    def __enter__(self):
        self._flush_all()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        self._flush_all()                       # ... so push it out while descriptor 1 still points at stderr
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


class RcclCommunicator(Communicator):
    """GPU ranks over RCCL/xGMI; collectives run on liblghip's compute stream"""

    def __init__(self, rank=None, world_size=None, id_path=None, rendezvous_timeout=300.0, selftest_timeout=60.0):
        from .autograd.hip import lib as L
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else world_size
        self._L = L
        self._lib = L.comm_lib()
        if id_path is None:
            # set per job by lightgrad_amd.launch; under torch.distributed.run all ranks of one launch are children of the same
            # launcher process, whose pid makes the rendezvous file unique per launch: a stale file of a crashed run is never read
            id_path = _job_rendezvous_path()
        self._id_path = id_path

        def make_id():
            buf = ctypes.create_string_buffer(128)
            L.comm_check(self._lib.lg_comm_get_unique_id(buf))
            return buf.raw
        # RCCL prints a version banner on the C-level stdout when a communicator is created; programs that
        # print machine-readable results on stdout (bench.py) must not get it mixed in: send it to stderr
        with _c_stdout_to_stderr():
            blob = _exchange_unique_id(self.rank, make_id, id_path, timeout=rendezvous_timeout)
            L.comm_check(self._lib.lg_comm_init(self.rank, self.world_size, ctypes.create_string_buffer(blob, 128)))
            # first collective: on the library's COMMUNICATION stream and awaited by polling an event, so that a collective
            # that never completes (a peer that did not make it through init) leaves the compute stream usable for the other
            # forms of the exchange; whatever RCCL prints lazily also lands on stderr
            L.comm_check(self._lib.lg_comm_selftest(float(selftest_timeout)))
        if self.rank == 0:
            try:
                os.remove(id_path)
            except OSError:
                pass

    def _check_flat(self, flat):
        assert isinstance(flat, HipTensor) and flat.is_contiguous() and flat.dtype == np.float32
        from .autograd.hip.tensor import flush_lazy_readers
        flush_lazy_readers(flat)        # collectives write in place
        return flat

    def allreduce_sum_(self, flat, forked=False):
        self._check_flat(flat)
        fn = self._lib.lg_comm_allreduce_forked_f32 if forked else self._lib.lg_comm_allreduce_f32
        self._L.comm_check(fn(flat.ptr, flat.numel(), 0))
        return flat

    def fork(self):
        self._L.comm_check(self._lib.lg_comm_fork())

    def join(self):
        self._L.comm_check(self._lib.lg_comm_join())

    def ranks_seen(self) -> int:
        """number of ranks the RCCL communicator itself reports (lg_comm_rank)"""
        r, n = ctypes.c_int(-1), ctypes.c_int(0)
        self._L.comm_check(self._lib.lg_comm_rank(ctypes.byref(r), ctypes.byref(n)))
        assert r.value == self.rank
        return n.value

    def allreduce_max_(self, flat):
        self._check_flat(flat)
        self._L.comm_check(self._lib.lg_comm_allreduce_f32(flat.ptr, flat.numel(), 1))
        return flat

    def broadcast_(self, flat, root=0):
        self._check_flat(flat)
        self._L.comm_check(self._lib.lg_comm_broadcast_f32(flat.ptr, flat.numel(), root))
        return flat

    def barrier(self):
        token = HipTensor.zeros((1,), requires_grad=False)
        self.allreduce_sum_(token)
        self._L.check(self._L.lib().lg_sync())

    def close(self):
        if getattr(self, "_abandoned", False):
            return                      # open_communicators dropped this form: a peer did not make it, ncclCommDestroy would wait for it
        self._L.comm_check(self._lib.lg_comm_destroy())


class PeerWindowCommunicator(Communicator):
    """GPU ranks of ONE node that exchange through peer-mapped device memory (include/lghip_p2p.h, csrc/p2p.hip): every
    collective is one ordinary kernel launch on the compute stream - capturable in a hipGraph, no collective library, no
    second stream - and the gradient exchange can ride INSIDE the optimizer launch (`fused_optimizer_exchange`:
    DataParallel.attach hands the optimizer this communicator, sync_gradients() then launches nothing).

    Works over xGMI between the GPUs of a node and between rank processes that SHARE one GPU (hipIpc handles are per
    process, not per device) - which is how the multi-rank device path is tested on one-GPU machines; ranks that share a
    GPU must each run on CUs of their own (`shared_gpu_environment`)."""
    fused_optimizer_exchange = True
    # Rendezvous files carry the number of the communicator within its job (every rank constructs its communicators in the same
    # order, so the numbers agree): a second communicator of a job never reads the handle or the "bye" of the first, and the
    # blob carries the exporter's pid, so a handle left behind by a crashed attempt of a restarted job (same launcher pid and
    # port under an elastic agent) is told from a live one by `os.kill(pid, 0)`.
    _generation = 0

    def __init__(self, rank=None, world_size=None, id_path=None, capacity_floats=1 << 22, rendezvous_timeout=120.0):
        from .autograd.hip import lib as L
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else world_size
        self._L, self._lib = L, L.lib()
        PeerWindowCommunicator._generation += 1
        self._prefix = "%s.p2p.%d" % (_job_rendezvous_path() if id_path is None else id_path, PeerWindowCommunicator._generation)
        self._timeout = rendezvous_timeout
        self._open = False
        handle = ctypes.create_string_buffer(L.P2P_HANDLE_BYTES)
        rc = self._lib.lg_p2p_export(self.rank, self.world_size, int(capacity_floats), handle)
        export_error = None if rc == 0 else self._lib.lg_last_error().decode()
        # a rank that could not export still publishes (64 zero bytes): its peers fail at once instead of waiting for it
        pid = ("%-16d" % os.getpid()).encode()

        def exporter_alive(data):
            try:
                os.kill(int(data[L.P2P_HANDLE_BYTES:].decode()), 0)
            except ProcessLookupError:
                return False                     # the file of an earlier, crashed attempt: wait for the live peer to replace it
            except (PermissionError, ValueError):
                pass                             # alive under another user / no pid: let the mapping decide
            return True
        try:
            blobs = _exchange_blobs(self.rank, self.world_size, (handle.raw if rc == 0 else bytes(L.P2P_HANDLE_BYTES)) + pid, self._prefix,
                                    timeout=rendezvous_timeout, accept=exporter_alive)
        except TimeoutError:
            if rc == 0:
                self._lib.lg_p2p_free()
            raise
        blobs = [b[:L.P2P_HANDLE_BYTES] for b in blobs]
        failed = [r for r, b in enumerate(blobs) if not any(b)]
        if failed:
            if rc == 0:
                self._lib.lg_p2p_free()
            raise L.HipError("peer-window exchange: rank(s) %s could not export a window%s" % (failed, ": " + export_error if export_error else ""))
        self._open = True
        try:
            L.check(self._lib.lg_p2p_connect(ctypes.create_string_buffer(b"".join(blobs), len(blobs) * L.P2P_HANDLE_BYTES)))
        except L.HipError:
            self._open = False
            self._lib.lg_p2p_free()
            raise
        try:
            self.barrier()              # every rank has mapped every window ...
        except L.HipError:              # ... or one could not, and this rank's wait for it gave up (lghip_p2p.h: LG_P2P_TIMEOUT_S)
            self._open = False
            self._lib.lg_p2p_disconnect()
            self._lib.lg_p2p_free()
            raise
        try:
            os.remove("%s.%d" % (self._prefix, self.rank))       # ... so nobody reads this file any more
        except OSError:
            pass

    def _check_flat(self, flat):
        assert isinstance(flat, HipTensor) and flat.is_contiguous() and flat.dtype == np.float32
        from .autograd.hip.tensor import flush_lazy_readers
        flush_lazy_readers(flat)        # collectives write in place
        return flat

    def failed(self) -> bool:
        """True once a wait of this communicator gave up and a synchronising call reported it (HipError, LG_ECOMM): every later
        collective raises, an exchange that did not happen never passes for one that did.  Recovery = close() (which then does
        not wait for the lost peer) and a new communicator - or a new job: after a lost peer the replicas are out of step."""
        if not self._open:
            return False
        f = ctypes.c_int(0)
        self._L.check(self._lib.lg_p2p_state(ctypes.byref(f), None))
        return bool(f.value)

    def memory_kind(self) -> str:
        kind = ctypes.create_string_buffer(32)
        self._L.check(self._lib.lg_p2p_state(None, kind))
        return kind.value.decode()

    def allreduce_sum_(self, flat, forked=False):
        self._check_flat(flat)
        self._L.check(self._lib.lg_p2p_allreduce_f32(flat.ptr, flat.numel(), self._L.P2P_SUM))
        return flat

    def allreduce_max_(self, flat):
        self._check_flat(flat)
        self._L.check(self._lib.lg_p2p_allreduce_f32(flat.ptr, flat.numel(), self._L.P2P_MAX))
        return flat

    def broadcast_(self, flat, root=0):
        self._check_flat(flat)
        if self.rank != root:
            with Gradients.no_grad():
                flat.fill(0.0)          # x + 0 + ... + 0: the root's values (a -0.0 arrives as +0.0)
        return self.allreduce_sum_(flat)

    def barrier(self):
        token = HipTensor.zeros((1,), requires_grad=False)
        self.allreduce_sum_(token)
        self._L.check(self._lib.lg_sync())

    def ranks_seen(self) -> int:
        r, n, cap = ctypes.c_int(-1), ctypes.c_int(0), ctypes.c_int64(0)
        self._L.check(self._lib.lg_p2p_rank(ctypes.byref(r), ctypes.byref(n), ctypes.byref(cap)))
        assert r.value == self.rank
        return n.value

    def close(self):
        if not self._open:
            return
        lost = self.failed() or getattr(self, "_abandoned", False)     # _abandoned: open_communicators dropped the form, peers may be gone
        self._open = False
        if not lost:
            try:
                self.barrier()                                    # every rank has finished its launches
            except self._L.HipError:
                lost = True                                       # the peer went away between the last step and now
        try:
            self._lib.lg_sync()                                   # (a failed communicator's stream may still hold reported launches)
        finally:
            self._L.check(self._lib.lg_p2p_disconnect())
        if not lost:
            # nobody maps my window any more once every rank has said so (the files carry the communicator's number and are
            # tiny: a rank that removed its own could be gone before a slower peer has read it; lightgrad_amd.launch clears the
            # job's directory).  With a lost peer there is nobody to wait for: unmap and free at once.
            try:
                _exchange_blobs(self.rank, self.world_size, b"bye", self._prefix + ".bye", timeout=self._timeout)
            except TimeoutError:
                pass
        self._L.check(self._lib.lg_p2p_free())


# ---- opening the communicators of a job so that no rank is left waiting on a form that does not work ------------------------
def _run_with_timeout(fn, seconds):
    """fn() on a helper thread (ctypes calls release the GIL); returns (value, None) or (None, "reason").  A call that does not
    come back is ABANDONED - its thread stays where it is (daemon) - and the caller goes on without that form."""
    import threading
    box = {}

    def body():
        try:
            box["value"] = fn()
        except BaseException as e:                      # the reason travels to the vote
            box["error"] = "%s: %s" % (type(e).__name__, (str(e).splitlines() or [""])[0])
    t = threading.Thread(target=body, daemon=True, name="lightgrad-open-communicator")
    t.start()
    t.join(seconds)
    if t.is_alive():
        return None, "no answer within %.0f s (abandoned)" % seconds
    if "error" in box:
        return None, box["error"]
    return box.get("value"), None


def _vote(rank, world, name, reason, timeout):
    """every rank publishes how its attempt at `name` went (None = fine); returns the reasons by rank, alike on every rank"""
    blob = ("ok" if reason is None else "no: " + reason)[:500].encode("utf-8", "replace").ljust(512)
    blobs = _exchange_blobs(rank, world, blob, "%s.vote.%s" % (_job_rendezvous_path(), name), timeout=timeout)
    out = []
    for b in blobs:
        text = b.decode("utf-8", "replace").rstrip()
        out.append(None if text == "ok" else text[4:])
    return out


_vote_round = 0


def open_communicators(rank, world, openers, timeout=120.0, vote_timeout=None):
    """Try the forms of the gradient exchange in the order given and keep those that work on EVERY rank.

        openers   [(name, callable -> Communicator)], e.g. [("rccl", ...), ("peer", ...), ("host", ...)]; a callable may be
                  None on a rank (the form is then skipped everywhere) and raises or hangs when its form does not work
        returns   ({name: Communicator}, {name: "why not"}), identical keys on every rank

    Each attempt runs under a timeout on a helper thread, then the ranks VOTE through the job's rendezvous files (no GPU, no
    collective library involved): a form that failed or did not answer on any rank is dropped by all of them - a rank whose own
    attempt succeeded closes nothing collectively (the peer may be gone) and simply does not use it.  The first real multi-GPU
    node a job meets may refuse any of the forms; the job then runs on the next one instead of dying in the first."""
    global _vote_round
    vote_timeout = max(2.0 * timeout, 60.0) if vote_timeout is None else vote_timeout
    opened, why_not = {}, {}
    for name, opener in openers:
        _vote_round += 1
        if opener is None:
            comm, reason = None, "not attempted on rank %d" % rank
        else:
            comm, reason = _run_with_timeout(opener, timeout)
        reasons = _vote(rank, world, "%d.%s" % (_vote_round, name), reason, vote_timeout)
        bad = [(r, w) for r, w in enumerate(reasons) if w is not None]
        if bad:
            why_not[name] = "; ".join("rank %d: %s" % rw for rw in bad[:3]) + (" (+%d more)" % (len(bad) - 3) if len(bad) > 3 else "")
            if comm is not None:
                comm._abandoned = True                      # never close()d collectively: its peers did not make it
        else:
            opened[name] = comm
    return opened, why_not


def peer_window_selftest(comm, n=70001):
    """an all-reduce whose result every rank can compute alone (per-rank seeds): the windows of all ranks carry data"""
    parts = [np.random.RandomState(4242 + r).uniform(-1, 1, n).astype(np.float32) for r in range(comm.world_size)]
    want = parts[0].copy()
    for q in parts[1:]:
        want = want + q                                      # rank order, fp32: the owner's bits
    t = HipTensor.from_numpy(parts[comm.rank], requires_grad=False)
    comm.allreduce_sum_(t)
    got = t.numpy()
    if not np.array_equal(got, want):
        raise AssertionError("peer-window all-reduce self-test: %d of %d values differ (max |diff| %.3g)"
                             % (int((got != want).sum()), n, float(np.abs(got - want).max())))
    return comm


def shared_gpu_environment(rank: int, world_size: int) -> dict:
    """environment of a rank process that shares ONE GPU with the other ranks of its job (tests, rehearsals): every rank
    binds device 0 and is confined to its own 1/world_size of the CUs (csrc/runtime.hip, LG_CU_MASK) - a rank whose exchange
    kernel waits for a peer then cannot keep that peer's kernels off the device.  Set before the library initialises."""
    return {"LIGHTGRAD_HIP_DEVICE": "0", "LG_CU_MASK": "%d/%d" % (rank, world_size)}


def _flat_and_views(cls, shapes):
    """one dense fp32 bucket of class `cls` and a view of it per shape (no copies)"""
    sizes = [int(np.prod(s, dtype=np.int64)) if len(s) else 1 for s in shapes]
    total = sum(sizes)
    views, off = [], 0
    if cls is HipTensor or issubclass(cls, HipTensor):
        flat = HipTensor.zeros((total,), requires_grad=False)
        for s, n in zip(shapes, sizes):
            views.append(HipTensor(flat.data, s, None, flat.offset + off, flat.dtype, requires_grad=False))
            off += n
    elif issubclass(cls, CpuTensor):
        flat = cls(np.zeros((total,), dtype=np.float32), requires_grad=False)
        for s, n in zip(shapes, sizes):
            views.append(cls(flat.data[off:off + n].reshape(s), requires_grad=False))
            off += n
    else:
        raise TypeError("no flat-bucket support for %s" % cls.__name__)
    return flat, views


class DataParallel(object):
    """Wraps a model's parameters for synchronous data-parallel SGD.

        dp = DataParallel(model.parameters(), comm)      # grads become views into one bucket
        optim = AdaBelief(model.parameters(), lr=1e-3, grad_scale=dp.grad_scale)
        ...
        optim.zero_grad(); loss.backward(); dp.sync_gradients(); optim.step()
    """

    def __init__(self, parameters, comm: Communicator, broadcast_parameters: bool = True, flatten: bool = False,
                 overlap: bool = False):
        self.parameters = tuple(parameters)
        self.comm = comm
        # overlap=True: start the all-reduce on the communicator's own stream as soon as the LAST parameter gradient of
        # the step has been enqueued, instead of after backward() - on the MLP that is before the input-gradient GEMM
        # of the first layer, which then runs concurrently with the exchange.  "Last" is learnt, not guessed: the first
        # step counts the gradient-written notifications of a whole backward pass (a static graph writes the same number
        # every step) and exchanges the plain way; from then on the count reaching that number triggers the exchange.
        # A step that writes fewer falls back to the plain way at sync_gradients() and re-learns; one that writes
        # MORE - a gradient landing in the bucket while it is being reduced - is an error and raises.
        self.overlap = bool(overlap)
        self._writes_expected, self._writes_seen, self._exchange_started = None, 0, False
        assert len(self.parameters) > 0
        cls = self.parameters[0].__class__
        shapes = [p.shape for p in self.parameters]
        self.bucket, views = _flat_and_views(cls, shapes)
        for p, g in zip(self.parameters, views):
            assert p.requires_grad and p.dtype == np.float32
            p._grad = g           # zero_grad -> fill(0) and add_grad -> += both act in place on the view
        self.grad_scale = 1.0 / comm.world_size
        self.offsets = tuple(int(o) for o in np.concatenate([[0], np.cumsum([p.numel() for p in self.parameters])]))
        self.flat_parameters = None
        self._exchange_in_optimizer = False
        if flatten:
            # re-home the parameter VALUES into one bucket too (the tensor objects the model holds stay the same):
            # lets the optimizer update every parameter with a single launch
            assert issubclass(cls, HipTensor), "flatten=True needs the HipTensor backend"
            self.flat_parameters, pviews = _flat_and_views(cls, shapes)
            with Gradients.no_grad():
                for p, v in zip(self.parameters, pviews):
                    v[...] = p
                    p._data, p._offset, p._byte_offset, p._strides, p._dense = v._data, v._offset, v._byte_offset, v._strides, v._dense      # ... except here
        if broadcast_parameters and comm.world_size > 1:
            self.broadcast_parameters()
        if self.overlap:
            for p in self.parameters:
                p._grad_written_hook = self._on_grad_written

    def _exchange_needed(self) -> bool:
        return self.comm.world_size > 1 or getattr(self, "always_sync", False)

    def _on_grad_written(self, p) -> None:
        if self._exchange_started:
            raise RuntimeError("DataParallel(overlap=True): a parameter gradient was written after the gradient exchange of "
                               "this step had started (%d writes expected) - the backward graph changed; call "
                               "reset_overlap() before a step with a different graph" % self._writes_expected)
        self._writes_seen += 1
        if self._writes_seen == self._writes_expected and self._exchange_needed():
            for q in self.parameters:
                q._materialize_zero_grad()    # a lazily zeroed gradient no kernel has written (optim.zero_grad)
            self.comm.fork()
            self.comm.allreduce_sum_(self.bucket, forked=True)
            self._exchange_started = True

    def set_overlap(self, enabled: bool) -> None:
        """switch the overlapped exchange on or off (off: one all-reduce on the compute stream in sync_gradients())"""
        self.overlap = bool(enabled)
        for p in self.parameters:
            p._grad_written_hook = self._on_grad_written if self.overlap else None
        self.reset_overlap()

    def reset_overlap(self) -> None:
        """forget the learnt write count (the next step exchanges after backward and learns again)"""
        self._writes_expected, self._writes_seen, self._exchange_started = None, 0, False

    def attach(self, optimizer, exchange_in_optimizer=None):
        """hand the flat buckets to an optimizer that can use them (one fill for zero_grad, one launch for step).
        exchange_in_optimizer: None = whenever the communicator offers it, False = keep the exchange in sync_gradients()"""
        assert self.flat_parameters is not None, "DataParallel(..., flatten=True) first"
        assert tuple(optimizer.parameters) == self.parameters
        optimizer.use_flat_buckets(self.flat_parameters, self.bucket, self.offsets)
        if exchange_in_optimizer is None:
            exchange_in_optimizer = getattr(self.comm, "fused_optimizer_exchange", False)
        assert not exchange_in_optimizer or getattr(self.comm, "fused_optimizer_exchange", False), \
            "%s cannot exchange inside the optimizer launch" % type(self.comm).__name__
        if exchange_in_optimizer and self._exchange_needed():
            # the gradient exchange happens INSIDE the optimizer's launch (lg_p2p_adam_multi_dev_f32): nothing to overlap,
            # nothing for sync_gradients() to launch
            optimizer.use_peer_exchange(self.comm)
            self._exchange_in_optimizer = True
            self.set_overlap(False)
        return optimizer

    def broadcast_parameters(self, root: int = 0):
        """make every replica start from rank `root`'s weights (one flat broadcast)"""
        cls = self.parameters[0].__class__
        flat, views = _flat_and_views(cls, [p.shape for p in self.parameters])
        with Gradients.no_grad():      # setitem must not become the parameters' tape context
            for p, v in zip(self.parameters, views):
                v[...] = p
            self.comm.broadcast_(flat, root)
            for p, v in zip(self.parameters, views):
                p[...] = v

    def sync_gradients(self):
        """sum the gradient bucket over all ranks, in place, asynchronously"""
        started, seen = self._exchange_started, self._writes_seen
        self._writes_seen, self._exchange_started = 0, False
        if not self._exchange_needed():
            return
        if self._exchange_in_optimizer:
            return                                # optimizer.step() sums the bucket over the ranks in its own launch
        if started:
            self.comm.join()                      # the optimizer (next on the compute stream) waits for the exchange
            return
        if self.overlap:
            self._writes_expected = seen if seen > 0 else None      # learn (first step) or re-learn (fewer writes than expected)
        for p in self.parameters:
            p._materialize_zero_grad()            # a lazily zeroed gradient no kernel has written yet (optim.zero_grad)
        self.comm.allreduce_sum_(self.bucket)

    def zero_grad(self) -> None:
        """zero the gradient bucket LAZILY (what `Optimizer.zero_grad` does once it holds the flat buckets): every parameter's
        gradient view is marked "zero pending" - the first backward kernel that reaches it overwrites instead of adding, readers
        and accumulating writers fill first - so no pass over the bucket happens here"""
        for p in self.parameters:
            p._grad_zero_pending = True

    def parameter_digest(self) -> float:
        """sum of |w| over all parameters: equal on every rank iff the replicas are in sync"""
        return float(sum(np.abs(p.numpy().astype(np.float64)).sum() for p in self.parameters))
