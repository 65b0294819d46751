"""World-size-2 gloo test of the data-parallel host logic (videovector_amd/dist.py): sharded sampling,
global-count gradient scaling, one all-reduce, identical updates -- must reproduce the single-process
oracle at the global batch.  The compute backend here is the CPU oracle (test stand-in for the HIP
engine; the GPU backend drives the same trainer class)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import videovector_amd as vv
from videovector_amd.dist import DataParallelTrainer, PipelinedTrainer, shard_items
from videovector_amd.synth import SyntheticVideos, init_weights

B_LOCAL, C, NN, F, D, STEPS = 8, 5, 4, 32, 16, 3
SAMPLER_KW = dict(context_size=C, num_negative_samples=NN, max_buffer_size=120, negative_swap_percentage=50)


class OracleBackend:
    def __init__(self, orc, table, W, b):
        self.orc, self.table = orc, table
        self.W, self.b = W.copy(), b.copy()
        self.hW, self.hb = np.zeros_like(W), np.zeros_like(b)
        self.slots = [torch.zeros(D * F + D, dtype=torch.float32) for _ in range(2)]
        self.grads = self.slots[0]
        self.last = (0.0, 0.0)

    def forward_backward(self, idx, global_count, slot=0, idx_dev_ptr=None):
        r = self.orc.forward_backward(self.table, idx, self.W, self.b, C_=C, Nn=NN, global_count=global_count,
                                      want=("dW", "db"))
        self.grads = self.slots[slot]
        self.grads[:D * F] = torch.from_numpy(r["dW"].reshape(-1))
        self.grads[D * F:] = torch.from_numpy(r["db"])
        self.last = (r["loss"], r["violations"])
        return self.grads

    def apply(self, lr, slot=0):
        g = self.slots[slot].numpy()
        dW, db = g[:D * F].reshape(D, F).copy(), g[D * F:].copy()
        self.orc.sgd_update(self.W, dW, self.hW, lr, 1.0, 0.9, 5e-4, 1.0)
        self.orc.sgd_update(self.b, db, self.hb, lr, 2.0, 0.9, 5e-4, 0.0)

    def loss_terms(self):
        return self.last


def _worker(rank, world, port, out_dir, pipelined=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    orc.set_threads(1)
    ds = SyntheticVideos(seed=3, n_videos=40)
    W, b = init_weights(3, D, F, std=0.05)
    smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B_LOCAL * world, **SAMPLER_KW)
    cls = PipelinedTrainer if pipelined else DataParallelTrainer
    tr = cls(OracleBackend(orc, ds.table(F), W, b), smp, NN, dist=dist, rank=rank, world=world)
    losses = []
    for it in range(STEPS):
        tr.step(0.05)
        losses.append(tr.global_loss())
    if pipelined:
        tr.flush()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), W=tr.be.W, b=tr.be.b, hW=tr.be.hW, losses=np.array(losses))
    dist.destroy_process_group()


def test_two_ranks_equal_single_process_global_batch(oracle, tmp_path):
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # every rank holds bit-identical parameters without any broadcast
    assert np.array_equal(r0["W"], r1["W"]) and np.array_equal(r0["b"], r1["b"]) and np.array_equal(r0["hW"], r1["hW"])
    # single process, global batch
    ds = SyntheticVideos(seed=3, n_videos=40)
    W, b = init_weights(3, D, F, std=0.05)
    hW, hb = np.zeros_like(W), np.zeros_like(b)
    smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B_LOCAL * world, **SAMPLER_KW)
    table = ds.table(F)
    for it in range(STEPS):
        idx = smp.next()
        r = oracle.forward_backward(table, idx, W, b, C_=C, Nn=NN, want=("dW", "db"))
        oracle.sgd_update(W, r["dW"], hW, 0.05, 1.0, 0.9, 5e-4, 1.0)
        oracle.sgd_update(b, r["db"], hb, 0.05, 2.0, 0.9, 5e-4, 0.0)
        assert abs(r0["losses"][it][0] - r["loss"]) <= 1e-5 * abs(r["loss"])
        assert r0["losses"][it][1] == r["violations"]
    assert np.abs(r0["W"] - W).max() <= 1e-5 * np.abs(W).max()
    assert np.abs(r0["hW"] - hW).max() <= 1e-4 * np.abs(hW).max()
    assert np.abs(r0["b"] - b).max() <= 1e-5 * max(np.abs(b).max(), 1e-6)


def test_two_ranks_pipelined_allreduce_equals_delayed_gradient_sgd(oracle, tmp_path):
    # PipelinedTrainer: async all-reduce of g_t overlapped with forward/backward t+1, update applied
    # afterwards == single-process SGD with a one-update gradient delay at the global batch
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path), True), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["W"], r1["W"]) and np.array_equal(r0["hW"], r1["hW"])
    ds = SyntheticVideos(seed=3, n_videos=40)
    W, b = init_weights(3, D, F, std=0.05)
    hW, hb = np.zeros_like(W), np.zeros_like(b)
    smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B_LOCAL * world, **SAMPLER_KW)
    table = ds.table(F)
    pending = None

    def apply(r):
        oracle.sgd_update(W, r["dW"], hW, 0.05, 1.0, 0.9, 5e-4, 1.0)
        oracle.sgd_update(b, r["db"], hb, 0.05, 2.0, 0.9, 5e-4, 0.0)
    for it in range(STEPS):
        r = oracle.forward_backward(table, smp.next(), W, b, C_=C, Nn=NN, want=("dW", "db"))   # at not-yet-updated weights
        assert abs(r0["losses"][it][0] - r["loss"]) <= 1e-5 * abs(r["loss"])
        if pending is not None:
            apply(pending)
        pending = r
    apply(pending)
    assert np.abs(r0["W"] - W).max() <= 1e-5 * np.abs(W).max()
    assert np.abs(r0["hW"] - hW).max() <= 1e-4 * np.abs(hW).max()


def test_shard_items():
    g = np.arange(24).reshape(8, 3)
    assert np.array_equal(np.concatenate([shard_items(g, r, 4) for r in range(4)]), g)
    with pytest.raises(AssertionError):
        shard_items(g, 0, 3)
