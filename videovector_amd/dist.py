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
    on torch's current stream so that the collective is ordered with the kernels."""

    def __init__(self, engine, cfg):
        import torch
        self.torch, self.eng, self.cfg = torch, engine, cfg
        # one explicit non-default stream for kernels and collective (the default stream's raw handle
        # is 0, which vv_set_stream reads as "the context's own stream")
        self.stream = torch.cuda.Stream()
        torch.cuda.set_stream(self.stream)
        self.eng.set_stream(self.stream.cuda_stream)
        n = engine.D * engine.F + engine.D
        self.grads = torch.zeros(n, dtype=torch.float32, device="cuda")
        self.eng.grads_bind(self.grads.data_ptr())

    def forward_backward(self, idx, global_count):
        self.cfg.set("global_count", int(global_count))
        self.eng.forward_backward(self.cfg, idx)
        return self.grads

    def apply(self, lr):
        self.cfg.set("lr", float(lr))
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
