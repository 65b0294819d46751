"""bench.py's N > 1 code path on ONE GPU: two ranks share device 0 and reduce through gloo (RCCL refuses two ranks
on one device).  Exercises what the driver's multi-GPU runs execute -- rendezvous, the shared sampler sliced per
rank out of rank 0's shared-memory prefetch ring, the global loss count, the all-reduce schedules, max-over-ranks
timing, the rank-0 JSON line -- except RCCL itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout, env):
    """subprocess.run(capture_output=True, text=True) whose time limit takes the WHOLE process tree down (the launcher's ranks are
    grandchildren: killed by process group, by PID -- a rank left behind would keep the GPU busy under every later test)."""
    import signal
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        out, err = p.communicate()
        raise AssertionError("timed out after %d s: %s\n--- stderr tail ---\n%s" % (timeout, " ".join(cmd[-14:]), err[-3000:]))
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


@pytest.mark.parametrize("mode", ["sync", "overlap", "sharded", "torch-sync", "stale", "peer-sharded", "peer-overlap", "peer-auto"])
def test_bench_two_ranks_on_one_device(mode):
    env = dict(os.environ, VV_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = "295%02d" % {"sync": 17, "overlap": 19, "torch-sync": 21, "stale": 23, "sharded": 25, "peer-sharded": 27, "peer-overlap": 29, "peer-auto": 37}[mode]
    extra = ["--comm", "torch"] if mode == "torch-sync" else (["--comm", "peer"] if mode.startswith("peer-") else [])   # peer: the one-shot direct exchange (hipIpc mappings)
    ar = "sync" if mode == "torch-sync" else mode.replace("peer-", "")          # (peer-auto: --allreduce auto = sharded over this transport)
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
                        "--allreduce", ar, "--sampler", "node", "--settle-ms", "2"] + extra + (["--no-extra-legs"] if mode != "stale" else []), 600, env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 2048 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - 2048 * 50 * 4 / (d["ms_per_step"] * 4e-3)) <= 1e-6 * d["value"]
    assert 0 < d["final_loss"] < 16 and d["roofline"]["frac"] > 0
    if mode != "stale":
        assert d["value_scope"].startswith("end to end") and "shared-memory ring" in d["config"]["sampler"]
        assert ("torch" in d["config"]["comm"]) == (mode == "torch-sync")
        assert ("direct peer exchange" in d["config"]["comm"]) == mode.startswith("peer-")
        if mode == "peer-auto":
            assert "reduce-scatter" in d["config"]["allreduce"]
    else:
        assert d["gpu_path_only"]["value"] > 0 and d["step_ms_stats"]["n"] == 4      # 4 steps (kernels are individually timed on every 5th step at the earliest)


def test_bench_fallbacks_keep_the_run_alive():
    """If a rank cannot attach the node's batch ring, every rank runs the identical global sampler for itself (same
    indices, hence the same loss as the shared ring gives), and the JSON line says so."""
    losses = {}
    for tag, extra_env in (("ring", {}), ("private", {"VV_BENCH_PRIVATE_SAMPLERS": "1"})):
        env = dict(os.environ, VV_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
        r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", "29531" if tag == "ring" else "29533",
                            os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
                            "--no-extra-legs", "--sampler", "node", "--settle-ms", "2"], 600, env)
        assert r.returncode == 0, r.stderr[-3000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert ("fallback" in d["config"]["sampler"]) == (tag == "private")
        losses[tag] = d["final_loss"]
    assert losses["ring"] == losses["private"]


def test_bench_per_rank_samplers_is_the_default_for_two_ranks():
    """N > 1 default: every rank runs the reference's sampler for its own batch (own draw stream and starting record);
    the line names the mode and carries the measured bound of the one-logical-sampler form beside it."""
    env = dict(os.environ, VV_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29535", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--settle-ms", "2"], 600, env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2048 and d["scaling"] == "weak"
    assert d["config"]["sampler"].startswith("one per rank") and "srand(1 + rank)" in d["config"]["sampler"]
    assert d["one_logical_sampler"]["ms_per_global_batch"] > 0 and d["one_logical_sampler"]["bound_triplets_per_s"] > 0
    assert d["value"] > 0 and 0 < d["final_loss"] < 16 and d["settle"]["steps"] == 8
    assert d["gpu_path_only"]["value"] > 0 and d["dense_execution"]["value"] > 0
    assert "logical CPUs" in d["config"]["cpu_binding_rank0"] or d["config"]["cpu_binding_rank0"].startswith("not bound")
    # the driver's launch form (torch.distributed.run): the schedule legs and the fresh-process legs come back here too (their own rendezvous:
    # the launcher's agent does not host a store for them)
    # N > 1 default: --allreduce auto = the fastest of the three exact schedules AS MEASURED BY THIS INVOCATION, before
    # the timed run; the two conformant forms ride beside `value` as first-class numbers and the deviation is stated in one line
    legs = d["schedules"]["legs"]
    assert set(legs) == {"sync", "overlap", "sharded"}
    auto = d["schedules"]["auto"]
    fastest = min(legs, key=lambda k: legs[k]["ms_per_step"])
    assert auto["picked"] == fastest and legs[fastest].get("default") and sum(1 for v in legs.values() if v.get("default")) == 1
    key = {"sync": "synchronous", "overlap": "F-chunk by F-chunk", "sharded": "reduce-scatter"}[fastest]
    assert key in d["config"]["allreduce"]
    for name, leg in legs.items():
        assert leg["final_loss"] == d["final_loss"], (name, leg)
    assert d["value_overlap_schedule"]["value"] == legs["overlap"]["value"] or fastest == "overlap"
    assert d["value_reference_batch"]["value"] == d["node_sampler_leg"]["value"] and d["value_reference_batch"]["value"] > 0
    assert "per-rank samplers" in d["config"]["deviation"]
    assert ("schedule " + fastest in d["config"]["deviation"]) == (fastest != "overlap")
    assert "error" not in d["peer_transport_leg"] and d["peer_transport_leg"]["final_loss"] == d["final_loss"], d["peer_transport_leg"]
    assert "error" not in d["node_sampler_leg"] and d["node_sampler_leg"]["wall_s"] < 200, d["node_sampler_leg"]


def test_bench_bare_command_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the process starts its two ranks itself
    (videovector_amd/launch.py) and rank 0's ONE line comes out of its stdout -- the form the driver's 1-GPU command has."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(VV_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--no-cpu-baseline", "--settle-ms", "2"], 900, env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 2048 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["allreduce"] and d["config"]["sampler"] and d["config"]["comm"]
    assert d["roofline"]["frac"] > 0 and d["roofline"]["kernel"]
    assert d["gpu_path_only"]["value"] > 0
    assert d["value"] > 0 and abs(d["value"] - 2048 * 50 * 6 / (d["ms_per_step"] * 6e-3)) <= 1e-6 * d["value"]
    # ONE invocation answers DESIGN.md 9.1 -- the three exact schedules on the library's transport, the sharded schedule
    # over the direct peer transport and the one-logical-sampler arrangement, each with its step time and final loss; the schedules (and the
    # other transport) reduce the same numbers in the same order: the same loss, bit for bit, as the line's own
    legs = d["schedules"]["legs"]
    assert set(legs) == {"sync", "overlap", "sharded"} and sum(1 for v in legs.values() if v.get("default")) == 1
    for name, leg in legs.items():
        assert "error" not in leg, (name, leg)
        assert leg["ms_per_step"] > 0 and leg["final_loss"] == d["final_loss"], (name, leg["final_loss"], d["final_loss"])
    peer = d["peer_transport_leg"]
    assert "error" not in peer, peer
    assert peer["ms_per_step"] > 0 and peer["final_loss"] == d["final_loss"] and "direct peer exchange" in peer["comm"]
    node = d["node_sampler_leg"]
    assert "error" not in node, node
    assert node["ms_per_step"] > 0 and 0 < node["final_loss"] < 16 and "one per node" in node["sampler"]


def test_bench_leg_process_that_fails_leaves_an_error_string_not_a_dead_job():
    """The direct peer transport's leg runs as fresh processes under a time limit: here its communicator is made to fail (an unknown transport
    name reaches vv_comm_init through VV_BENCH_TEST_PEER_AS) -- the line still comes out, with the error in the leg's place."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(VV_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", VV_BENCH_TEST_PEER_AS="no-such-transport")
    r = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline", "--settle-ms", "2"], 900, env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert "error" in d["peer_transport_leg"] and d["peer_transport_leg"]["error"], d["peer_transport_leg"]
    assert d["value"] > 0 and "error" not in d["node_sampler_leg"]


def test_bench_bare_command_a_dead_rank_fails_the_job():
    """rank 1 dies right after the rendezvous while rank 0 waits for it: rc != 0 and a message, not a hang."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(VV_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", VV_BENCH_TEST_FAIL_RANK="1", VV_LAUNCH_GRACE="5")
    r = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline", "--no-extra-legs", "--settle-ms", "2"], 300, env)
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert "rank 1 exited with code 7" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_eight_ranks_overlap_node_sampler_matches_one_process_at_the_global_batch():
    """What the driver's N = 8 run executes by default in the library -- eight ranks, the launcher, the eight-rank chunk plan,
    k_sgd's publication, the gated forward GEMM, the node's ONE sampler sliced per rank -- on the one-device hook, against ONE
    process that takes the same global batches (B = 8192) through a single engine: after the same number of iterations the
    parameters are the same to rounding, so rank 0's loss on its 1024 items of the last batch is (<= 1e-5)."""
    import numpy as np
    import videovector_amd as vv
    from videovector_amd.synth import SyntheticVideos, init_weights
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(VV_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    K, Wm, settle_ms = 3, 1, 0.5                                   # settle steps = ceil(0.5 / 0.25) = 2 -> 6 iterations
    r = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", str(K), "--warmup", str(Wm),
                        "--no-cpu-baseline", "--no-extra-legs", "--allreduce", "overlap", "--sampler", "node",
                        "--settle-ms", str(settle_ms), "--sampler-threads", "2"], 1500, env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 8192 and d["config"]["parallelism"] == "dp8"
    assert "F-chunk by F-chunk" in d["config"]["allreduce"] and "shared-memory ring" in d["config"]["sampler"]
    n_it = d["settle"]["steps"] + Wm + K
    assert n_it == 6
    # ---- the same iterations in ONE process at the global batch
    B, C, NN, F, D, BG = 1024, 5, 50, 4096, 512, 8192
    ds = SyntheticVideos(seed=1701, n_videos=2048)
    smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=BG, context_size=C, num_negative_samples=NN,
                     max_buffer_size=5000, negative_swap_percentage=50, max_same_video_negs=0)
    W0, b0 = init_weights(1701, D, F)
    eng = vv.Engine(0, "f16")
    eng.table_synth(ds.seed, ds.n_rows, F)
    eng.params_set(W0, b0)
    cfg_g = vv.StepConfig(BG, C, NN)
    cfg_r = vv.StepConfig(B, C, NN, global_count=BG * NN)
    lr_at = lambda it: 1e-3 * (1.0 + 1e-3 * it) ** -0.75
    for it in range(n_it - 1):
        cfg_g.set("lr", lr_at(it))
        eng.forward_backward(cfg_g, smp.next())
        eng.apply_update(cfg_g)
    eng.forward_backward(cfg_r, smp.next()[:B])                    # rank 0's items of the last batch, at the parameters of 5 global updates
    loss, viol = eng.loss()
    print("DIST8 rank-0 loss after %d iterations: 8 ranks %.7f, one process %.7f" % (n_it, d["final_loss"], loss))
    assert abs(d["final_loss"] - loss) <= 1e-5 * loss
