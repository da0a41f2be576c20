"""One flat gradient bucket for every trainable policy of a process.

The reference wraps each policy in its own DistributedDataParallel (trainer_a2c.py:137-146), i.e. at
least one latency-bound collective per policy per iteration; the tag_continuous run has two policies of
95 275 parameters each.  Here the `.grad` of every trainable parameter of every policy is a VIEW into one
contiguous float32 buffer (762 KB for that run), the backward passes accumulate straight into it, and a
training iteration issues exactly ONE all-reduce -- RCCL over xGMI on a GPU node
(backend "nccl"), gloo in the CPU tests.  Rank 0's initial parameters are broadcast once, as DDP does.
"""
import torch
import torch.distributed as dist


class GradientBucket:
    def __init__(self, modules, device=None):
        self.params = [p for m in modules for p in m.parameters() if p.requires_grad]
        assert self.params, "no trainable parameters"
        device = device if device is not None else self.params[0].device
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=device)
        offset = 0
        for p in self.params:
            assert p.dtype == torch.float32
            p.grad = self.flat[offset: offset + p.numel()].view_as(p)  # autograd accumulates into the view in place
            offset += p.numel()
        self.collectives = 0  # all-reduces issued so far (tests / metrics)
        # bench.py's `trainer` object: time the collective where it happens (events on the stream it is enqueued on for
        # RCCL, wall clock for gloo); off by default -- the events cost nothing but are nobody's business otherwise
        self.time_collectives = False
        self._timed = None
        self.reattached = 0   # times a detached .grad had to be put back into the bucket (see _reattach)

    @staticmethod
    def _group_active():
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def zero(self):
        """instead of optimizer.zero_grad(): the views must stay attached to the bucket"""
        self._reattach()
        self.flat.zero_()

    def attached(self):
        return all(p.grad is not None and p.grad.data_ptr() >= self.flat.data_ptr()
                   and p.grad.data_ptr() < self.flat.data_ptr() + 4 * self.flat.numel() for p in self.params)

    def _reattach(self):
        """`optimizer.zero_grad()` (set_to_none=True by default) or any code that replaces a `.grad` detaches it
        from the bucket; the collective would then average a stale buffer and the ranks would diverge silently
        (DistributedDataParallel raises in that situation, a plain bucket does not).  A detached gradient is
        copied into its slice and the view is restored, so the bucket is always what the optimizers see."""
        if self.attached():
            return
        offset = 0
        for p in self.params:
            view = self.flat[offset: offset + p.numel()].view_as(p)
            lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.flat.numel()
            if p.grad is None:
                view.zero_()
            elif not (lo <= p.grad.data_ptr() < hi):
                view.copy_(p.grad)
            p.grad = view
            offset += p.numel()
        self.reattached += 1

    def all_reduce_mean(self):
        """average the gradients of all policies over the ranks: one collective"""
        self._reattach()
        if not self._group_active():
            return
        if not self.time_collectives:
            dist.all_reduce(self.flat)
        elif self.flat.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(self.flat)
            e1.record()
            self._timed = (e0, e1)
        else:
            import time

            t0 = time.perf_counter()
            dist.all_reduce(self.flat)
            self._timed = (time.perf_counter() - t0) * 1e6
        self.flat.div_(dist.get_world_size())
        self.collectives += 1

    def read_allreduce_us(self):
        """duration (us) of the last all-reduce issued while `time_collectives` was set, or None"""
        if self._timed is None:
            return None
        if isinstance(self._timed, tuple):
            e0, e1 = self._timed
            e1.synchronize()
            return e0.elapsed_time(e1) * 1e3
        return float(self._timed)

    @torch.no_grad()
    def broadcast_parameters(self, src=0):
        """every rank starts from rank `src`'s parameters (one broadcast of the flattened parameters)"""
        if not self._group_active():
            return
        flat = torch.cat([p.detach().reshape(-1) for p in self.params])
        dist.broadcast(flat, src=src)
        offset = 0
        for p in self.params:
            p.copy_(flat[offset: offset + p.numel()].view_as(p))
            offset += p.numel()
