"""The data-parallel exchange through the C ABI (vv_comm_init / vv_allreduce_grads / vv_comm_overlap / vv_comm_schedule):

  * two ranks = two PROCESSES on the one visible GPU, host-staged shared-memory transport (RCCL refuses two ranks on one
    device): both schedules -- whole buffer after the backward pass ("sync") and the update F-chunk by F-chunk on the
    communication stream while the NEXT step's forward GEMM already runs and waits per chunk inside the kernel ("overlap")
    -- give bit-identical parameters on both ranks, equal to each other, and the same step as ONE process on the global
    batch (sums reassociated: <= 5e-4);
  * the one-shot DIRECT exchange (VV_COMM_PEER: hipIpc peer mappings, the reduce-scatter and the all-gather one kernel each, meeting
    points in host-coherent flags) between 2 and 8 processes on the one GPU: bit for bit the shared-memory transport's results in all
    three schedules; a missing rank ends in an error, not in a hang;
  * one rank over real RCCL: the dlopen'ed library, the communication stream, the chunk gates run on the GPU box;
  * the gates under a slow exchange (test hook: the communication stream is held 300 us in front of every chunk): the
    forward GEMM really waits where its K loop reaches a chunk that has not arrived; results unchanged bit for bit.
"""
import multiprocessing as mp
import os
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

B, C, Nn, F, D, ITERS = 64, 5, 10, 2048, 512, 6          # F = 2048: 8 K-tiles per F-chunk (the gates sit inside the K loop)


def _case():
    from videovector_amd.synth import SyntheticVideos, init_weights
    ds = SyntheticVideos(seed=21, n_videos=300)
    W, b = init_weights(21, D, F, std=0.02)
    return ds, W, b


def _batches(ds, world):
    import videovector_amd as vv
    s = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=world * B, context_size=C, num_negative_samples=Nn,
                   max_buffer_size=1000, negative_swap_percentage=50)
    out = [s.next() for _ in range(ITERS)]
    s.close()
    return out


def _rank_main(rank, world, id_path, overlap, transport, q, env=None):
    os.environ.update(env or {})
    import videovector_amd as vv
    ds, W, b = _case()
    eng = vv.Engine(0, "f16")
    eng.table_synth(ds.seed, ds.n_rows, F)
    eng.params_set(W, b)
    eng.comm_init(world, rank, id_path, transport)
    mix = isinstance(overlap, str) and overlap.startswith("mix_")
    if overlap in ("sharded", "mix_sharded_overlap", "mix_sharded_exposed"): eng.comm_schedule("sharded")
    elif overlap == "mix_overlap_sync": eng.comm_overlap(True)
    else: eng.comm_overlap(overlap)
    cfg = vv.StepConfig(B, C, Nn, global_count=world * B * Nn, lr=0.05)
    losses = []
    for k, g in enumerate(_batches(ds, world)):
        if mix and k == ITERS // 2:                # the schedule changes in mid-run: every rank makes the same call
            if overlap == "mix_sharded_overlap": eng.comm_overlap(True)        # vv_comm_overlap(ctx, 1) leaves the sharded schedule
            elif overlap == "mix_sharded_exposed": eng.grads_device()          # a holder of the gradient buffer: whole-matrix updates from here on
            elif overlap == "mix_overlap_sync": eng.comm_overlap(False)        # the next update runs on the compute stream
        eng.forward_backward(cfg, g[rank * B:(rank + 1) * B])
        eng.apply_update(cfg)                      # all-reduces first
        losses.append(eng.loss()[0])
    Wn, bn, hW, hb = eng.params_get()              # (sharded schedule: a collective -- every rank is here)
    eng.comm_destroy()
    q.put((rank, Wn, bn, hW, losses))


def _run_world(world, overlap, transport="shm", env=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    id_path = os.path.join(tempfile.gettempdir(), "vv_comm_test_%d_%d_%s_%s" % (os.getpid(), world, str(overlap), transport))
    if os.path.exists(id_path):
        os.unlink(id_path)
    procs = [ctx.Process(target=_rank_main, args=(r, world, id_path, overlap, transport, q, env)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=300)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def rel(a, r):
    return float(np.linalg.norm(a.astype(np.float64) - r) / max(np.linalg.norm(r), 1e-30))


def test_two_ranks_on_one_gpu_match_the_global_batch():
    import videovector_amd as vv
    ds, W, b = _case()
    # the single-process reference: the same three global batches in one call each
    eng = vv.Engine(0, "f16")
    eng.table_synth(ds.seed, ds.n_rows, F)
    eng.params_set(W, b)
    cfg = vv.StepConfig(2 * B, C, Nn, lr=0.05)
    ref_losses = []
    for g in _batches(ds, 2):
        eng.step(cfg, g)
        ref_losses.append(eng.loss()[0])
    W1, b1, hW1, _ = eng.params_get()
    eng.close()
    out = {}
    for overlap in (False, True):
        res = _run_world(2, overlap)
        (Wa, ba, ha, la), (Wb, bb, hb_, lb) = res[0], res[1]
        assert np.array_equal(Wa, Wb) and np.array_equal(ba, bb) and np.array_equal(ha, hb_), "ranks diverged"
        e = rel(Wa - W, W1 - W)
        print("COMM overlap=%s: %d-step parameter change vs one process on the global batch %.3e; losses %s / %s vs %s"
              % (overlap, ITERS, e, la, lb, ref_losses))
        assert e <= 3e-3 and rel(ba - b, b1 - b) <= 3e-3        # (sums reassociated, then six free-running iterations)
        for k in range(ITERS):                     # global loss = mean of the shard losses
            assert abs(0.5 * (la[k] + lb[k]) - ref_losses[k]) <= 2e-4 * ref_losses[k]
        out[overlap] = (Wa, ba)
    # the two schedules reduce the same numbers in the same order: identical results
    assert np.array_equal(out[False][0], out[True][0]) and np.array_equal(out[False][1], out[True][1])


@pytest.mark.parametrize("world,inline", [(2, "1"), (2, "0"), (8, "1")])
def test_sharded_update_is_the_synchronous_update_bit_for_bit(world, inline):
    """reduce-scatter -> the solver's rule on this rank's D / world rows -> all-gather of the 16-bit copy, the bias and the per-block
    maxima (vv_comm_schedule 2), on `world` processes sharing the one GPU: every rank ends with the parameters, the bias and the history
    (gathered by the collective vv_params_get) that the synchronous all-reduce schedule gives -- the shard's sums are the all-reduce's
    sums in the same rank order, the rule is elementwise, the W -> half scale comes from the same maxima -- bit for bit, on every rank."""
    sync = _run_world(world, False)
    shard = _run_world(world, "sharded", env={"VV_COMM_INLINE": inline})      # 1 (default): in the compute stream; 0: on the communication stream, one gate
    for r in range(world):
        for k in range(3):
            assert np.array_equal(shard[r][k], shard[0][k]), "ranks diverged (rank %d, array %d)" % (r, k)
            assert np.array_equal(shard[r][k], sync[r][k]), "sharded differs from sync (rank %d, array %d)" % (r, k)
        assert shard[r][3] == sync[r][3]                       # the per-rank losses of all six iterations: the forward passes read the same bits
    assert np.isfinite(shard[0][0]).all() and np.abs(shard[0][2]).max() > 0


@pytest.mark.parametrize("mode", ["mix_sharded_overlap", "mix_sharded_exposed", "mix_overlap_sync"])
def test_schedule_changes_in_mid_run_keep_the_synchronous_trajectory(mode):
    """(a) a step that leaves the SHARDED schedule without vv_comm_schedule -- vv_comm_overlap(ctx, 1), or the gradient buffer
    handed out by vv_grads_device -- found the other ranks' rows of the fp32 master W and of the history stale and applied the whole-matrix
    rule to them; (b) an update queued on the COMPUTE stream right after an overlapped step did not wait for the end of the previous update's
    kernels on the communication stream.  Three steps in one schedule, three in the other, two ranks: bit for bit the all-synchronous run."""
    ref = _run_world(2, False)
    res = _run_world(2, mode)
    for r in (0, 1):
        for i, what in enumerate(("W", "b", "hW")):
            assert np.array_equal(res[r][i], ref[r][i]), (mode, "rank %d %s differs from the synchronous run" % (r, what))
        assert res[r][3] == ref[r][3], (mode, res[r][3], ref[r][3])


@pytest.mark.parametrize("overlap", [False, True, "sharded"])
def test_one_rank_over_real_rccl(overlap):
    import videovector_amd as vv
    ds, W, b = _case()
    g = _batches(ds, 1)
    ref = vv.Engine(0, "f16")
    ref.table_synth(ds.seed, ds.n_rows, F); ref.params_set(W, b)
    cfg = vv.StepConfig(B, C, Nn, lr=0.05)
    for x in g:
        ref.step(cfg, x)
    W0 = ref.params_get()[0]
    ref.close()
    res = _run_world(1, overlap, transport="rccl")
    assert np.array_equal(res[0][0], W0), "a one-rank RCCL all-reduce must leave the gradients unchanged"


def test_gated_forward_waits_for_a_slow_exchange():
    """One rank over RCCL, overlapped schedule, the communication stream held 300 us in front of every chunk's all-reduce:
    the host queues the next forward GEMM at once, its waves find chunks missing at their gates and wait there (the
    update's own kernels run beside them).  Parameters after six iterations: bit for bit those of a plain engine."""
    import videovector_amd as vv
    ds, W, b = _case()
    g = _batches(ds, 1)
    ref = vv.Engine(0, "f16")
    ref.table_synth(ds.seed, ds.n_rows, F); ref.params_set(W, b)
    cfg = vv.StepConfig(B, C, Nn, lr=0.05)
    for x in g:
        ref.step(cfg, x)
    W0, b0, h0, _ = ref.params_get()
    ref.close()
    res = _run_world(1, True, transport="rccl", env={"VV_COMM_TEST_DELAY_US": "300"})
    assert np.array_equal(res[0][0], W0) and np.array_equal(res[0][1], b0) and np.array_equal(res[0][2], h0)
    # the sharded schedule's one gate (in front of the forward GEMM's first W tile), the exchange held 300 us
    res = _run_world(1, "sharded", transport="rccl", env={"VV_COMM_TEST_DELAY_US": "300", "VV_COMM_INLINE": "0"})
    assert np.array_equal(res[0][0], W0) and np.array_equal(res[0][1], b0) and np.array_equal(res[0][2], h0)
    # ... and the default form of the sharded schedule (its three steps in the compute stream, no gate), held the same way
    res = _run_world(1, "sharded", transport="rccl", env={"VV_COMM_TEST_DELAY_US": "300"})
    assert np.array_equal(res[0][0], W0) and np.array_equal(res[0][1], b0) and np.array_equal(res[0][2], h0)


@pytest.mark.parametrize("world,schedule", [(2, False), (2, True), (2, "sharded"), (8, False), (8, "sharded")])
def test_direct_peer_exchange_is_the_shared_memory_sum_bit_for_bit(world, schedule):
    """VV_COMM_PEER between `world` processes on the one GPU (the peer mappings are then local; the code path -- handle exchange,
    meeting kernels, one-shot reduce and gather kernels -- is the one an xGMI node runs): the sums are taken in rank order like the
    shared-memory transport's, so parameters, bias, history and every iteration's loss are the same bits, on every rank."""
    shm = _run_world(world, schedule)
    peer = _run_world(world, schedule, transport="peer")
    for r in range(world):
        for k in range(3):
            assert np.array_equal(peer[r][k], peer[0][k]), "ranks diverged (rank %d, array %d)" % (r, k)
            assert np.array_equal(peer[r][k], shm[r][k]), "peer differs from shm (rank %d, array %d)" % (r, k)
        assert peer[r][3] == shm[r][3]
    assert np.isfinite(peer[0][0]).all() and np.abs(peer[0][2]).max() > 0


def _one_step_rank(rank, id_path, q):
    """both ranks take one step together (the buffers get mapped); rank 1 then leaves, rank 0 takes another"""
    os.environ["VV_COMM_TIMEOUT"] = "3"
    import videovector_amd as vv
    ds, W, b = _case()
    eng = vv.Engine(0, "f16")
    eng.table_synth(ds.seed, ds.n_rows, F)
    eng.params_set(W, b)
    eng.comm_init(2, rank, id_path, "peer")
    cfg = vv.StepConfig(B, C, Nn, global_count=2 * B * Nn, lr=0.05)
    g = _batches(ds, 2)
    eng.forward_backward(cfg, g[0][rank * B:(rank + 1) * B])
    eng.apply_update(cfg)
    eng.params_get()
    if rank == 1:
        os._exit(0)
    try:
        eng.forward_backward(cfg, g[1][:B])
        eng.apply_update(cfg)                      # rank 1 never raises its flag again: the meeting kernel gives up after 3 s
        eng.params_get()                           # waits for the update in flight, then reports what it ran into
        q.put("no error")
    except Exception as e:                         # noqa: BLE001
        q.put(str(e))


def test_direct_peer_exchange_a_missing_rank_is_an_error_not_a_hang():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    id_path = os.path.join(tempfile.gettempdir(), "vv_comm_test_%d_lonely" % os.getpid())
    procs = [ctx.Process(target=_one_step_rank, args=(r, id_path, q)) for r in range(2)]
    for p in procs:
        p.start()
    msg = q.get(timeout=180)
    for p in procs:
        p.join(60)
    assert "did not reach the exchange in time" in msg, msg
