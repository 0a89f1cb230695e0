"""P virtual ranks of ONE process on ONE GPU, for the tests: test infrastructure, not product.

RCCL refuses two ranks per device, so the multi-process tests stage their collectives through gloo -- which takes the
`backend == "gloo"` branches of mpreid/distributed.py (host tensors, all_gather instead of all_to_all_single).  The branches
an 8-GPU RCCL run takes -- device-tensor all_gather_into_tensor, the ragged all_to_all_single of column_to_row_blocks, the
padded dist.gather into the pinned host matrix, the CSR / index all-gathers of the sharded re-ranking -- execute with
P > 1 only here: every virtual rank is a thread that routes its collectives through an EmulatedGroup
(mpreid.distributed.use_group), which reports backend "nccl", delivers the data by device copies, and ASSERTS what RCCL
requires but does not check (it hangs or corrupts instead): device tensors only, contiguous buffers, a dtype NCCL knows, equal
sizes where the collective needs them, and -- all_to_all_single -- that what rank s sends to rank r is exactly what r expects
from s.

The threads take turns (a baton): exactly one of them runs between two collectives, so the virtual ranks may share the
process-wide caches (workspaces keyed by stream, pinned host buffers) the way separate processes never would have to.
"""
import threading

import torch
import torch.distributed as dist

NCCL_DTYPES = {torch.int8, torch.uint8, torch.int32, torch.int64, torch.float16, torch.bfloat16, torch.float32, torch.float64,
               torch.bool}


class EmulatedWorld:
    def __init__(self, world: int, backend: str = "nccl", require_cuda: bool = True):
        # require_cuda=False: the CPU tests run the same "nccl" branches on host tensors (index arithmetic only)
        self.world, self.backend, self.require_cuda = world, backend, require_cuda
        self.slots = [None] * world
        self.barrier = threading.Barrier(world)
        self.baton = threading.Lock()
        self.log = []                       # (collective name, per-rank detail) in call order, appended by rank 0
        self.failed = None

    def group(self, rank: int) -> "EmulatedGroup":
        return EmulatedGroup(self, rank)

    def run(self, fn, timeout: float = 600.0):
        """fn(rank) on every virtual rank; returns the list of results; the first exception of any rank is re-raised"""
        from mpreid import distributed as D
        results, errors = [None] * self.world, [None] * self.world

        def body(r):
            if self.require_cuda:
                torch.cuda.set_device(0)
            self.baton.acquire()
            try:
                with D.use_group(self.group(r)):
                    results[r] = fn(r)
                    if self.require_cuda:
                        torch.cuda.synchronize()
            except BaseException as e:   # noqa: BLE001
                errors[r] = e
                self.failed = e
                self.barrier.abort()
            finally:
                self.baton.release()

        threads = [threading.Thread(target=body, args=(r,), name=f"vrank{r}") for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout)
            assert not t.is_alive(), "a virtual rank is stuck (a collective some rank never entered?)"
        for e in errors:
            if e is not None and not isinstance(e, threading.BrokenBarrierError):
                raise e
        for e in errors:
            if e is not None:
                raise e
        return results


class EmulatedGroup:
    """the functions of torch.distributed that mpreid uses, for one virtual rank"""
    ReduceOp = dist.ReduceOp

    def __init__(self, w: EmulatedWorld, rank: int):
        self.w, self.rank = w, rank

    # -- queries ------------------------------------------------------------------------------------------------------
    def is_initialized(self):
        return True

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.w.world

    def get_backend(self):
        return self.w.backend

    # -- rendezvous -----------------------------------------------------------------------------------------------------
    def _exchange(self, name, payload):
        """deposit this rank's payload, wait for all, return every rank's payload (rank order).  The baton is handed over
        while waiting, so exactly one virtual rank computes at a time."""
        w = self.w
        w.slots[self.rank] = (name, payload)
        w.baton.release()
        try:
            w.barrier.wait()
            got = list(w.slots)
            w.barrier.wait()                 # nobody deposits the next payload before everyone has read this one
        finally:
            w.baton.acquire()
        names = {g[0] for g in got}
        assert len(names) == 1, f"ranks are in different collectives: {[g[0] for g in got]}"
        if self.rank == 0:
            w.log.append(name)
        return [g[1] for g in got]

    def _check(self, t, what):
        assert torch.is_tensor(t), what
        if self.w.backend == "nccl" and self.w.require_cuda:
            assert t.is_cuda, f"{what}: RCCL needs device tensors"
        assert t.is_contiguous(), f"{what}: not contiguous"
        assert t.dtype in NCCL_DTYPES, f"{what}: dtype {t.dtype} has no NCCL type"

    # -- collectives ----------------------------------------------------------------------------------------------------
    def barrier(self):
        self._exchange("barrier", None)

    def all_gather(self, out_list, x):
        self._check(x, "all_gather input")
        assert len(out_list) == self.w.world
        got = self._exchange("all_gather", x)
        for o, g in zip(out_list, got):
            self._check(o, "all_gather output")
            assert o.shape == g.shape and o.dtype == g.dtype, ("all_gather: ranks passed different shapes", o.shape, g.shape)
            o.copy_(g)

    def all_gather_into_tensor(self, out, x):
        self._check(x, "all_gather_into_tensor input")
        self._check(out, "all_gather_into_tensor output")
        got = self._exchange("all_gather_into_tensor", x)
        assert all(g.shape == x.shape and g.dtype == x.dtype for g in got), "all_gather_into_tensor: unequal inputs"
        assert out.numel() == self.w.world * x.numel() and out.dtype == x.dtype, (out.shape, x.shape)
        flat = out.view(-1)
        for r, g in enumerate(got):
            flat[r * x.numel():(r + 1) * x.numel()].copy_(g.reshape(-1))

    def all_to_all_single(self, recv, send, output_split_sizes=None, input_split_sizes=None):
        self._check(send, "all_to_all_single send")
        self._check(recv, "all_to_all_single recv")
        W = self.w.world
        assert output_split_sizes is not None and input_split_sizes is not None
        assert len(output_split_sizes) == W and len(input_split_sizes) == W
        assert sum(input_split_sizes) == send.shape[0], (sum(input_split_sizes), send.shape)
        assert sum(output_split_sizes) == recv.shape[0], (sum(output_split_sizes), recv.shape)
        got = self._exchange("all_to_all_single", (send, list(input_split_sizes), list(output_split_sizes)))
        off = 0
        for s, (sbuf, s_in, s_out) in enumerate(got):
            # what rank s sends to me must be what I expect from s -- RCCL would hang or scribble
            assert s_in[self.rank] == output_split_sizes[s], (f"rank {s} sends {s_in[self.rank]} to rank {self.rank}, which "
                                                              f"expects {output_split_sizes[s]}")
            assert sbuf.dtype == recv.dtype
            lo = sum(s_in[:self.rank])
            n = s_in[self.rank]
            recv[off:off + n].copy_(sbuf[lo:lo + n])
            off += n

    def gather(self, x, gather_list=None, dst=0):
        self._check(x, "gather input")
        got = self._exchange("gather", x)
        assert all(g.shape == x.shape and g.dtype == x.dtype for g in got), "gather: ranks passed different shapes"
        if self.rank == dst:
            assert gather_list is not None and len(gather_list) == self.w.world
            for o, g in zip(gather_list, got):
                self._check(o, "gather output")
                o.copy_(g)
        else:
            assert gather_list is None, "gather: only the destination passes a list"

    def all_reduce(self, t, op=dist.ReduceOp.SUM):
        self._check(t, "all_reduce")
        got = self._exchange("all_reduce", t.clone())
        st = torch.stack(got)
        if op == dist.ReduceOp.MAX:
            t.copy_(st.max(dim=0).values)
        elif op == dist.ReduceOp.SUM:
            t.copy_(st.sum(dim=0))
        else:
            raise NotImplementedError(op)

    def all_gather_object(self, out_list, obj):
        got = self._exchange("all_gather_object", obj)
        out_list[:] = got
