"""Data-parallel host logic: one process per GPU over torch.distributed
(backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).

  * every rank runs the IDENTICAL sampler for the GLOBAL batch (the reference's sampler is one
    sequential libc-rand() stream with a cursor and a negative buffer, so it cannot be sharded without
    changing the indices) and keeps items [rank*B, (rank+1)*B);
  * the local loss gradient is scaled by the GLOBAL count B_global*Nn, so the sum over ranks is the
    single-process gradient of the reference at the global batch;
  * ONE fp32 all-reduce(sum) of the flat [dW | db] buffer; every rank then applies the same update,
    so parameters stay bit-identical across ranks without a broadcast.

The compute backend is injected: the GPU backend below drives the C ABI (Engine); the CPU tests
inject an oracle-backed stand-in to exercise exactly this file under gloo.
"""
import numpy as np


def shard_items(global_idx, rank, world):
    """Items of this rank out of a global batch [B_global][C+Nn]."""
    bg = global_idx.shape[0]
    assert bg % world == 0, "global batch must divide evenly over the ranks"
    b = bg // world
    return np.ascontiguousarray(global_idx[rank * b:(rank + 1) * b])


class GpuBackend:
    """vv_forward_backward / vv_apply_update with the gradient buffer bound to a torch tensor, all
    on torch's current stream so that the collective is ordered with the kernels.  Two gradient
    buffers ("slots") let one be all-reduced while the next iteration fills the other."""

    def __init__(self, engine, cfg, stream=None):
        import torch
        self.torch, self.eng, self.cfg = torch, engine, cfg
        # one explicit non-default stream for kernels and collective (the default stream's raw handle
        # is 0, which vv_set_stream reads as "the context's own stream")
        self.stream = stream if stream is not None else torch.cuda.Stream()
        torch.cuda.set_stream(self.stream)
        self.eng.set_stream(self.stream.cuda_stream)
        n = engine.D * engine.F + engine.D
        self.slots = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(2)]
        self.grads = self.slots[0]
        self.eng.grads_bind(self.grads.data_ptr())

    def forward_backward(self, idx, global_count, slot=0, idx_dev_ptr=None):
        self.cfg.set("global_count", int(global_count))
        self.grads = self.slots[slot]
        self.eng.grads_bind(self.grads.data_ptr())
        self.eng.forward_backward(self.cfg, idx, idx_dev_ptr)
        return self.grads

    def apply(self, lr, slot=0):
        self.cfg.set("lr", float(lr))
        self.eng.grads_bind(self.slots[slot].data_ptr())
        self.eng.apply_update(self.cfg)

    def loss_terms(self):
        loss, viol = self.eng.loss()       # local mean over the local count
        return loss, viol


class DataParallelTrainer:
    def __init__(self, backend, sampler, Nn, dist=None, rank=0, world=1):
        self.be, self.sampler, self.Nn = backend, sampler, Nn
        self.dist, self.rank, self.world = dist, rank, world

    def step(self, lr):
        g = self.sampler.next()
        g = g[0] if isinstance(g, tuple) else g
        idx = shard_items(g, self.rank, self.world)
        grads = self.be.forward_backward(idx, g.shape[0] * self.Nn)
        if self.dist is not None and self.world > 1:
            self.dist.all_reduce(grads)                      # sum, fp32
        self.be.apply(lr)

    def global_loss(self):
        """Mean loss / total violations over the global batch (2-float all-reduce; display only)."""
        loss, viol = self.be.loss_terms()
        if self.dist is None or self.world == 1:
            return loss, viol
        import torch
        t = torch.tensor([loss / self.world, viol], dtype=torch.float64,
                         device=self.be.grads.device if hasattr(self.be, "grads") else "cpu")
        self.dist.all_reduce(t)
        return float(t[0]), float(t[1])


class PipelinedTrainer(DataParallelTrainer):
    """The overlap the north-star asks for: the all-reduce of iteration t's gradients runs (RCCL's own
    stream, `async_op=True`) while iteration t+1's forward/backward executes; the update with g_t is
    applied after that forward/backward, before iteration t+2.  Gradients are therefore one update
    stale -- g_{t+1} is taken at the weights that do not yet contain g_t -- the classic delayed-
    gradient pipeline; `flush()` applies the last pending update.  With world == 1 the collective is
    skipped and the same schedule runs (used to test the bookkeeping on one GPU).

    DataParallelTrainer (above) is the exact synchronous alternative."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.pending = None          # (work handle or None, slot, lr)
        self.t = 0

    def step(self, lr, idx_local=None, global_batch=None, idx_dev_ptr=None):
        if idx_local is None and idx_dev_ptr is None:
            g = self.sampler.next()
            g = g[0] if isinstance(g, tuple) else g
            idx_local, global_batch = shard_items(g, self.rank, self.world), g.shape[0]
        slot = self.t & 1
        if idx_dev_ptr is not None:
            grads = self.be.forward_backward(None, global_batch * self.Nn, slot, idx_dev_ptr)
        else:
            grads = self.be.forward_backward(idx_local, global_batch * self.Nn, slot)
        work = None
        if self.dist is not None and self.world > 1:
            work = self.dist.all_reduce(grads, async_op=True)
        prev, self.pending = self.pending, (work, slot, lr)
        if prev is not None:
            self._apply(prev)
        self.t += 1

    def _apply(self, item):
        work, slot, lr = item
        if work is not None:
            work.wait()              # GPU: the current stream waits for the collective; host does not block
        self.be.apply(lr, slot)

    def flush(self):
        if self.pending is not None:
            self._apply(self.pending)
            self.pending = None
