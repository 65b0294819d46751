#!/usr/bin/env python3
"""bench.py -- triplets/sec of the videovec_embedding training step on N MI355X (one process per GPU).

  python bench.py --gpus N --steps K --warmup W       (N > 1 without a launcher: starts its own N ranks, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1] per GPU: synthetic fc7 features 4096-d -> 512-d embedding, batch 1024 per GPU
(global batch N*1024, weak scaling), context window +-2 (context_size 5), 50 negatives.

A step = one full training iteration on one batch: (row de-duplication,) gather-GEMM forward, fused score/loss
forward+backward, (per-row gradient sums,) gather-GEMM^T weight gradient, fused SGD update; for N > 1 the exact exchange of
[dW|db]: by default the SHARDED update in the compute stream (reduce-scatter, the rule on D / N rows, all-gather of the 16-bit copy:
--allreduce sharded); or chunk by chunk on the library's communication stream while the next step's forward GEMM already runs and
waits per chunk inside the kernel (overlap); or the whole all-reduce between backward and update (sync).  An N > 1 line carries all
three as `schedules` legs, the sharded schedule over the direct peer transport and the one-logical-sampler arrangement as legs run in
fresh processes (`peer_transport_leg`, `node_sampler_leg`), each with its step time and final loss.
What is timed, and reported as what:

  value            END TO END, sustainable: the triplet sampler (the reference's sequential libc-rand() sampler,
                   bit-exact) runs on prefetch threads INSIDE the timed region, as BasePrefetchingDataLayer's thread does
                   in the reference (base_data_layer.cpp:69-95); every step takes its batch out of the prefetch ring,
                   hands the 225 KB of indices to the GPU (pinned, device-mapped staging) and runs the iteration.  For N > 1
                   every rank runs the reference's sampler for its own 1024 items (--sampler rank, the default: srand(1 + rank),
                   its own starting record); --sampler node: ONE sampler per node (rank 0) draws the global batch and
                   publishes it in POSIX shared memory, every rank takes its 1024 items (sampler-bound).
  gpu_path_only    the same K steps with the index batches already resident in HBM (sampler excluded) -- the kernel path
                   alone, what round 1 reported as `value`.
  dense_execution  gpu_path_only with row de-duplication off: every sampled row projected separately, as the reference does.
  dropout_execution  the same with the shipped dropout ratio 0.9 ("a dropout-0.9 number reported separately"): de-duplicated (the
                   mask per instance on the shared projection); dropout_dense_execution: the dense form beside it.  Formerly: dense
                   kernels (every instance has its own mask), the mask from a counter-based generator in the forward epilogue.
  bf16_execution   end to end with bf16 MFMA operands (the north star's operand type; the default is f16: same MFMA rate,
                   and only f16 meets the 1e-3 embedding tolerance).
  step_ms_stats    min / median / p95 / max of the individual step times of a further run of the K steps with one HIP event
                   per step.
  settle           every leg runs --settle-ms (50) ms of untimed steps in front of its W warm-up steps: after an idle spell the
                   device needs ~25 ms of continuous work to reach a steady step time; the timed region is exactly K steps.
Kernel durations (kernels_ms, roofline): HIP events on the kernels' own dispatch packets, on every max(5, K/8)-th step; inside the
timed region only the forward GEMM (the dominant kernel, the roofline's) carries them -- a timed dispatch cannot be pipelined behind
its predecessor, ~5 us each --, the other kernels are timed on the same steps of the step_ms_stats leg.
roofline.traffic / roofline.step_traffic (HBM bytes per launch / per step): the full one-GPU line measures them ITSELF -- two child processes
of this script under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, 8 steps each) before this process touches the GPU
(live_traffic(); VV_BENCH_LIVE_TRAFFIC=0, --no-extra-legs or N > 1: the committed passes of profiles/pmc_latest.json instead, labelled as such).
Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

B_PER_GPU, C, NN, F, D = 1024, 5, 50, 4096, 512
N_VIDEOS, SEED = 2048, 1701
MFMA_PEAK_TFLOPS = 2500.0     # dense bf16/f16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def cpu_baseline(ds, idx, W, b, items, iters, threads=0):
    """The oracle (CPU restatement of the reference path, layer by layer with an sgemm for fc7) timed like `caffe time`
    (tools/caffe.cpp:193-266): whole ForwardBackward + ComputeUpdateValue + Update iterations on `items` batch items.
    threads = 0: all host threads OpenMP offers; 1: the single-thread figure SURVEY 8(d) also asks for.  The reference
    links an external BLAS for its sgemm (Makefile.config:34): when the machine has one (MKL ships in the image) it is
    timed against the oracle's own blocked sgemm on one warm-up iteration each and the faster is used and named.  The
    GFLOP/s of the two fc7 GEMMs alone is reported next to the whole-iteration rate."""
    # OpenMP placement of the oracle's own runtime (read when liboracle.so loads -- torch's bundled OpenMP is another library): threads on
    # neighbouring cores of one socket.  profiles/r05_cpu_scaling.txt: 32 threads 0.273 s bound `close` against 0.358 s unbound.
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    # an external BLAS must not bring a second, spinning thread pool onto the same bound cores (MKL's own OpenMP runtime did: the iterations
    # after its trial ran 20x slower): its GNU threading layer shares the oracle's runtime, and idle workers of any Intel runtime sleep at once
    os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
    os.environ.setdefault("KMP_BLOCKTIME", "0")
    from oracle import oracle as orc
    orc.set_threads(threads)
    # What this process may use of the host: the MI355X boxes' containers carry a CPU-time quota (cgroup cpu.max: 16 CPUs' worth on a 256-thread
    # host) -- more runnable threads than ~2x the quota are throttled, not run (64 threads 0.47-0.63 s, 128 0.74-0.97 s, 256 4-20 s per iteration)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = os.cpu_count() or 1
    allow = max(1, min(affinity, int(np.ceil(quota)) if quota else affinity))
    sh = idx[:items]
    uniq, inv = np.unique(sh.reshape(-1), return_inverse=True)
    table = ds.table(F, uniq)
    idx_local = inv.reshape(sh.shape).astype(np.int32)
    Wo, bo = W.copy(), b.copy()
    hW, hb = np.zeros_like(W), np.zeros_like(b)

    def one():
        t0 = time.perf_counter()
        r = orc.forward_backward(table, idx_local, Wo, bo, C_=C, Nn=NN, want=("dW", "db"))
        orc.sgd_update(Wo, r["dW"], hW, 1e-3, 1.0, 0.9, 5e-4, 1.0)
        orc.sgd_update(bo, r["db"], hb, 1e-3, 2.0, 0.9, 5e-4, 0.0)
        return time.perf_counter() - t0
    blas = orc.find_blas()
    # Thread count: every hardware thread is not the fastest choice on a two-socket SMT host (128 threads ran this step
    # 2x slower than 64 on the 2 x EPYC 9575F box); like the BLAS, it is picked on one warm-up iteration each.
    tried = {}
    if threads == 0:
        all_t = orc.get_threads()
        one()                                    # first touch of the buffer pool
        cand = {min(all_t, allow), min(all_t, 2 * allow), min(all_t, 3 * allow)} if quota else {all_t, max(1, all_t // 2), max(1, all_t // 4)}
        for t_ in sorted(cand, reverse=True):
            orc.set_threads(t_)
            tried[t_] = one()
        orc.set_threads(min(tried, key=tried.get))
    # the oracle's own sgemm first, completely (timed iterations and its GEMM rates) -- then the external BLAS, if any, the same way: whatever a
    # foreign library leaves behind (threads, bindings) cannot touch the first measurement; the faster of the two is reported
    orc.set_blas(None)
    one()
    ts = [one() for _ in range(iters)]
    t = float(np.mean(ts))
    gemm = orc.gemm_gflops(items * (C + NN), D, F) if hasattr(orc, "gemm_gflops") else None
    use_ext = False
    if blas and orc.set_blas(blas):
        one()
        ts_e = [one() for _ in range(iters)]
        if float(np.mean(ts_e)) < t:
            use_ext, ts, t = True, ts_e, float(np.mean(ts_e))
            gemm = orc.gemm_gflops(items * (C + NN), D, F) if hasattr(orc, "gemm_gflops") else None
    orc.set_blas(None)
    out = {"value": items * NN / t, "unit": "triplets/s", "cores": orc.get_threads(), "kind": "port",
           "blas": ("cblas_sgemm of " + os.path.basename(blas)) if use_ext else "the oracle's own blocked OpenMP sgemm",
           "sample": "%d of %d batch items (%d rows) of the same 4096->%d, C5, Nn%d step, "
                     "%d timed iterations after warm-up, %.2f s each" % (items, B_PER_GPU, items * (C + NN), D, NN, iters, t),
           "iteration_gflops": 4.0 * items * (C + NN) * F * D / t / 1e9,
           "host_logical_cpus": os.cpu_count(), "cpu_quota_cpus": quota, "cpus_in_affinity_mask": affinity,
           "omp_binding": "%s / %s" % (os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES")),
           "note": ("this process may use %.0f CPUs' worth of time (cgroup cpu.max) of the host's %d logical CPUs: the thread count is picked among "
                    "1x / 2x / 3x that allowance on one warm-up iteration each, threads bound to neighbouring cores (profiles/r05_cpu_scaling.txt: "
                    "every count above ~2x the allowance runs slower -- throttling, not NUMA); `cores` = the threads used.  A stated baseline for "
                    "THIS allowance, never the target" % (quota, os.cpu_count())) if quota else
                   ("thread count picked on one warm-up iteration each (all / half / a quarter of the host's threads), threads bound to "
                    "neighbouring cores; a stated baseline, never the target")}
    if tried:
        out["threads_tried_s_per_iteration"] = {str(k): round(v, 3) for k, v in tried.items()}
    if gemm is not None:
        out["fc7_gemm_gflops"] = gemm
    return out


BOX_REF_GEMM_TFLOPS = 1200.0   # the reference rate `ms_per_step_at_ref` is quoted at (a round number near the pool's median box)


def box_record(box, ms_per_step, gemm_ms_in_step):
    """The line's `box` object: vv_box_probe right before the settle steps of the timed leg (a cold device: noisy) and right after the timed
    leg (steady state: repeatable to ~1 % on one box) -- the second is the calibration.  ms_per_step_at_ref scales the step's GEMM share
    (the two GEMM kernels' own durations) to a reference box and leaves the rest as measured: ms - gemm_ms x (1 - gemm_tflops / ref)."""
    if not box:
        return None
    b, a = box["before"], box.get("after") or box["before"]
    keys = ("gemm_tflops", "gemm_ms", "gemm_clock_mhz", "copy_tbs", "copy_ms")
    return {"gemm_tflops": a["gemm_tflops"], "copy_tbs": a["copy_tbs"], "gemm_clock_mhz": a["gemm_clock_mhz"],
            "after_timed_leg": {k: a[k] for k in keys}, "before_settle_steps_cold": {k: b[k] for k in keys},
            "ref_gemm_tflops": BOX_REF_GEMM_TFLOPS, "gemm_ms_in_step": gemm_ms_in_step,
            "ms_per_step_at_ref": ms_per_step - gemm_ms_in_step * (1.0 - a["gemm_tflops"] / BOX_REF_GEMM_TFLOPS),
            "probes": "vv_box_probe (include/videovec.h): gemm = the benchmark's forward instantiation (k_fwd_gemm_ph, f16, 192-row tiles, 216 "
                      "workgroups) on contiguous rows of a random table, %d x %d x %d, operands uniform in [-1, 1), 24 launches behind 8 warm-up "
                      "launches, clock = s_memtime / s_memrealtime inside the kernel (median over workgroups); copy = 1 GiB device-to-device, "
                      "read + written bytes per second, 6 copies behind 2" % tuple(b["gemm_shape"]),
            "note": "a calibration of the box, not of the code: the top-level figures are the probe right AFTER the timed leg (steady state); "
                    "ms_per_step_at_ref = ms_per_step - gemm_ms_in_step x (1 - gemm_tflops / ref_gemm_tflops), gemm_ms_in_step = the two GEMM "
                    "kernels' durations"}


def live_traffic(cfg_argv, first_timeout_s=300.0, timeout_s=150.0):
    """HBM bytes per kernel launch MEASURED BY THIS INVOCATION (roofline.traffic, roofline.step_traffic): two child processes of this script,
    each under `rocprofv3 --pmc <counter> --kernel-trace` (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, as /opt/skills/guides/MI355X_MICROARCH.md
    prescribes; no other trace domain), 8 steps of the same workload each (no box probe: its launches carry the product kernel's name; no extra
    legs, no CPU baseline), started BEFORE this process touches the GPU and killed by process group on a time limit.  Returns (per-kernel dict
    as tools/make_pmc_json.fold gives it, note) or (None, why not) -- the caller then falls back to the committed profile."""
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process runs under a profiler itself: no nested passes"
    tmp = tempfile.mkdtemp(prefix="vv_bench_pmc_", dir="/tmp")
    t0 = time.perf_counter()
    try:
        for i, ctr in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, ctr), "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "6", "--warmup", "2", "--settle-ms", "0", "--no-cpu-baseline",
                   "--no-extra-legs"] + cfg_argv
            env = dict(os.environ, VV_BENCH_NO_BOX="1", VV_BENCH_CHILD="1", TMPDIR="/tmp")
            p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=first_timeout_s if i == 0 else timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)            # the child is a session leader: it and everything it started, by PID
                except OSError:
                    pass
                p.wait()
                return None, "the %s pass did not finish within its time limit: killed" % ctr
            if rc != 0:
                return None, "the %s pass exited with %d" % (ctr, rc)
        from tools.make_pmc_json import SOURCE, fold
        d = fold(tmp)
        if not d:
            return None, "the passes left no counter rows"
        return d, "measured by this invocation (%.0f s): %s; 8 steps per pass (2 warm-up + 6), the average over all launches of a kernel" % (
            time.perf_counter() - t0, SOURCE)
    except Exception as e:                                      # noqa: BLE001 -- the line falls back to the committed profile
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def helper_main():
    """`bench.py --helper`: a process that never touches the GPU.  Every rank of an N > 1 run starts one BEFORE its own first GPU call; later it
    asks it -- one JSON line in, one out -- to run a further bench.py rank (a leg that may not come back: the direct peer transport on a
    node it has never run on) as a FRESH process, under a time limit.  The asking rank is never replaced and never exec()s; a leg that
    wedges is killed by process group (this helper is a session leader: the rank kills it and everything it started, by PID)."""
    import signal
    import subprocess
    from videovector_amd import launch as _launch

    def die_with_parent():                                      # (between fork and exec: the pre-bound prctl only, videovector_amd/launch.py)
        if _launch._PRCTL is not None:
            _launch._PRCTL(1, int(signal.SIGTERM), 0, 0, 0)     # PR_SET_PDEATHSIG
    die_with_parent()                                           # this helper goes when its rank goes, and a leg goes when the helper goes
    for line in sys.stdin:
        try:
            cmd = json.loads(line)
        except ValueError:
            break
        reply = {"rc": None, "line": None, "err_tail": None}
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + cmd["argv"], env=dict(os.environ, **cmd.get("env", {})),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=cmd.get("timeout", 300), text=True,
                               preexec_fn=die_with_parent)
            reply["rc"] = p.returncode
            js = [x for x in p.stdout.splitlines() if x.startswith("{")]
            reply["line"] = js[-1] if js else None
            reply["err_tail"] = p.stderr[-600:]
        except subprocess.TimeoutExpired as e:
            reply["err_tail"] = "timed out after %.0f s" % e.timeout
        except Exception as e:                                  # noqa: BLE001 -- whatever it was, the asking rank gets a line back
            reply["err_tail"] = "%s: %s" % (type(e).__name__, e)
        sys.stdout.write(json.dumps(reply) + "\n")
        sys.stdout.flush()


class LegHelper:
    """The asking side of helper_main()."""

    def __init__(self):
        import subprocess
        self.p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--helper"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                  text=True, start_new_session=True)

    def run(self, argv, env, timeout_s):
        import select
        import signal
        if self.p is None or self.p.poll() is not None:
            return {"rc": None, "line": None, "err_tail": "the helper process is gone"}
        try:
            self.p.stdin.write(json.dumps({"argv": argv, "env": env, "timeout": timeout_s}) + "\n")
            self.p.stdin.flush()
            r, _, _ = select.select([self.p.stdout], [], [], timeout_s + 30.0)
            if r:
                return json.loads(self.p.stdout.readline())
        except (OSError, ValueError) as e:
            return {"rc": None, "line": None, "err_tail": "helper: %s" % e}
        try:                                                    # no answer: the helper and whatever it started, by process group
            os.killpg(self.p.pid, signal.SIGKILL)
        except OSError:
            pass
        self.p = None
        return {"rc": None, "line": None, "err_tail": "no answer within %.0f s: killed" % (timeout_s + 30.0)}

    def close(self):
        if self.p is not None and self.p.poll() is None:
            try:
                self.p.stdin.close()
                self.p.wait(timeout=10)
            except Exception:                                   # noqa: BLE001
                self.p.kill()
        self.p = None


def main():
    t_process_start = time.perf_counter()
    if len(sys.argv) > 1 and sys.argv[1] == "--helper":
        return helper_main()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prec", default=os.environ.get("VV_PREC", "f16"), choices=["f16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dedup", default="on", choices=["on", "off"],
                    help="row de-duplication of the batch (results identical up to the rounding of reassociated "
                         "sums); 'off' executes the reference-equivalent dense work")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="only the end-to-end leg (skips gpu_path_only, dense_execution, bf16_execution, step_ms_stats)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg5", "shipped"],
                    help="cfg2 (default, the metric's configuration): batch 1024/GPU, 50 negatives, 4096->512.  cfg5: the "
                         "per-GPU work of BASELINE configs[4] (batch 4096, 200 negatives, 4096->1024, quoted for bf16).  shipped: "
                         "the reference's own project files (mednet_embedding_train.prototxt:13-23,200,226: batch 128, window 5, "
                         "10 negatives of which up to 6 from the same video -- quirk Q1 --, 4096->4096, dropout 0.9).  The last two "
                         "are informational, not the contract's bench line")
    ap.add_argument("--allreduce", default="auto", choices=["auto", "sync", "overlap", "sharded", "stale"],
                    help="N>1.  'auto' (default): the three exact schedules below are timed first, on this node, K steps each on resident "
                         "batches, and the fastest becomes the line's schedule (with --no-extra-legs: 'sharded').  'overlap': exact synchronous "
                         "SGD with the exchange hidden behind the NEXT step's forward GEMM -- the update runs F-chunk by F-chunk on the library's communication stream "
                         "(all-reduce of the chunk, SGD on its columns, publish) while the forward GEMM already runs and "
                         "waits per chunk inside the kernel.  'sync': the same update, the whole all-reduce of [dW|db] "
                         "between backward and update, exposed.  'sharded': reduce-scatter of the gradients, the solver's rule on this rank's "
                         "D / N rows of W, all-gather of the 16-bit copy of W and the bias (3/4 of the wire bytes, 1/N of the update's "
                         "memory traffic; the same parameters bit for bit).  'stale': the all-reduce of iteration t overlaps "
                         "iteration t+1 and gradients are applied one update late (NOT the reference's algorithm; opt-in, "
                         "labelled)")
    ap.add_argument("--comm", default="lib", choices=["lib", "torch", "peer"],
                    help="N>1: who runs the gradient all-reduce.  'lib' (default): the product library's own RCCL "
                         "communicator on its communication stream (vv_comm_init / vv_allreduce_grads).  'torch': "
                         "torch.distributed.all_reduce on the tensor the gradients are bound to (cross-check; no overlap mode).  "
                         "'peer': the library's one-shot direct exchange over hipIpc peer mappings (VV_COMM_PEER: reduce-scatter and "
                         "all-gather as one kernel each over all xGMI links; opt-in -- RCCL stays the default until this has run on an "
                         "eight-GPU node)")
    ap.add_argument("--sampler", default="auto", choices=["auto", "node", "rank"],
                    help="N>1: who draws the batches.  'rank' (auto for N>1): every rank runs the reference's sampler for its own "
                         "batch of 1024 (its own draw stream: srand(1 + rank), and its own starting record) -- the global batch is "
                         "N independent reference batches.  'node': ONE logical sampler draws the global batch of N*1024 exactly as "
                         "a single reference process with that batch size would (SURVEY 8e; the parity-tested form) and every rank "
                         "takes its slice out of a shared-memory ring -- bound by the sampler's serial walk (DESIGN.md 8)")
    ap.add_argument("--cpu-bind", default="auto", choices=["auto", "off"],
                    help="N>1: give every rank its own block of physical cores on its GPU's NUMA node (videovector_amd/hostbind.py)")
    ap.add_argument("--sampler-threads", type=int, default=int(os.environ.get("VV_SAMPLER_THREADS", "4")))
    ap.add_argument("--prefetch-depth", type=int, default=128,
                    help="batches the sampler keeps ahead of the consumer (the reference: 1).  128 (29 MB of index batches) covers the stretches of up to several hundred steps in "
                         "which the sampler -- 0.09 ms per batch at the median against a 0.21 ms step since round 5 (0.13 before), with the host's occasional slow stretches -- falls behind (profiles/r02_long_run.txt, r05_sampler_rates.txt)")
    ap.add_argument("--settle-ms", type=float, default=50.0,
                    help="untimed steps of the same workload, this many ms of them, in front of the W warm-up steps of every "
                         "leg: the device needs ~25 ms of continuous work after an idle spell (set-up, a host-side leg) before "
                         "its step time is steady (profiles/r02_step_ablations.txt, 3.).  0 = none: W warm-up steps only")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process -- which has not imported torch or touched HIP, and
        # never exec()s -- becomes the launcher of N ranks of itself (videovector_amd/launch.py) and relays rank 0's line.
        from videovector_amd.launch import launch_ranks
        sys.exit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    # N > 1, the full line: the helper for the legs that run as fresh processes -- started HERE, before this rank's first GPU call
    leg_helper = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not args.no_extra_legs and args.workload == "cfg2" and os.environ.get("VV_BENCH_CHILD") != "1" \
            and args.allreduce != "stale" and args.comm != "torch":
        leg_helper = LegHelper()

    # roofline.traffic measured by this invocation (one GPU, the full line only): the two PMC passes run as child processes HERE, before this
    # process imports torch or touches HIP (VV_BENCH_LIVE_TRAFFIC=0: the committed profile instead)
    live_pmc, live_pmc_note = None, "not attempted"
    if args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_extra_legs and os.environ.get("VV_BENCH_CHILD") != "1" \
            and os.environ.get("VV_BENCH_LIVE_TRAFFIC", "1") != "0":
        live_pmc, live_pmc_note = live_traffic(["--workload", args.workload, "--dedup", args.dedup, "--prec", args.prec])

    global B_PER_GPU, NN, D
    if args.workload == "cfg5":
        B_PER_GPU, NN, D = 4096, 200, 1024
    shipped = args.workload == "shipped"
    if shipped:
        B_PER_GPU, NN, D = 128, 10, 4096
        if args.gpus != 1:
            raise SystemExit("--workload shipped is a one-GPU informational run")
        args.no_extra_legs = True
    DROPOUT = 0.9 if shipped else 0.0        # mednet_embedding_train.prototxt:226
    MAX_SAME = 6 if shipped else 0           # :19 (same-video negatives: quirk Q1)
    import torch
    import videovector_amd as vv
    from videovector_amd.synth import SyntheticVideos, init_weights

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one process per GPU, or none: a bare "
                         "`python bench.py --gpus N` starts its own ranks)" % (args.gpus, world))
    dist = None
    cpu_bind = "none (one rank)"
    if world > 1 and args.cpu_bind == "auto":
        # before the sampler's threads and the collective library's helper threads exist: they inherit the mask
        try:
            from videovector_amd import hostbind
            ndev = max(1, torch.cuda.device_count())
            n_local = int(os.environ.get("LOCAL_WORLD_SIZE", world))
            pci = []
            for r in range(n_local):
                pr = torch.cuda.get_device_properties(r % ndev)
                pci.append((pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id))
            cpu_bind = hostbind.bind(local_rank, pci)
        except Exception as e:       # an unexpected topology must not cost the run: unbound ranks are merely slower
            cpu_bind = "not bound (%s: %s)" % (type(e).__name__, e)
            print("rank %d: cpu binding skipped: %s" % (rank, e), file=sys.stderr)
    elif world > 1:
        cpu_bind = "off"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # VV_DIST_BACKEND=gloo with several ranks on ONE device is a test hook for the N > 1 code path on a 1-GPU box
        # (RCCL refuses two ranks on one device); the benchmark proper is nccl = RCCL, one rank per GPU.
        backend = os.environ.get("VV_DIST_BACKEND", "nccl")
        if backend != "nccl":
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        if os.environ.get("VV_BENCH_TEST_FAIL_RANK") == str(rank):
            # test hook: this rank dies while the others wait for it in a collective (tests/test_gpu_dist.py)
            print("rank %d: exiting on VV_BENCH_TEST_FAIL_RANK" % rank, file=sys.stderr)
            os._exit(7)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    K, Wm = args.steps, args.warmup
    # settle steps: a fixed count per workload (every rank must run the same number), ~args.settle_ms of GPU work
    S = int(np.ceil(args.settle_ms / (2.3 if args.workload == "cfg5" else 0.25))) if args.settle_ms > 0 else 0
    Bg = B_PER_GPU * world
    ds = SyntheticVideos(seed=SEED, n_videos=N_VIDEOS)
    skw = dict(batch_size=Bg, context_size=C, num_negative_samples=NN, max_buffer_size=5000, negative_swap_percentage=50,
               max_same_video_negs=MAX_SAME)

    smode = args.sampler if args.sampler != "auto" else ("rank" if world > 1 else "node")
    if world == 1:
        smode = "node"                                  # one rank: the two are the same sampler
    ring_name = "vv_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", str(os.getppid())))
    sampler = None
    sampler_note = ""
    one_logical = None
    ring_attached = False
    if shipped:
        # quirk-Q1 batches carry a second index array (which row each slot's LAST feature comes from): drawn on the calling
        # thread, step by step, as the facade does for such batches (caffe_facade/src/net.cpp:505-517)
        sampler = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **skw)
        ring = None; ring_consumer = 0; item_begin = 0
    elif smode == "rank":
        # ---- per-rank samplers: the reference's sampler, one instance per rank, batch B_PER_GPU, its own rand() stream and
        # starting record.  Beside it rank 0 measures what ONE logical sampler of the global batch delivers (the bound of 'node').
        if rank == 0:
            sn = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **skw)
            sn.prefetch_start(depth=args.prefetch_depth, threads=args.sampler_threads, shm_name=None, consumers=1)
            for _ in range(3): sn.next()
            nb = max(4, 32 // world)
            t0 = time.perf_counter()
            for _ in range(nb): sn.next()
            ms_g = (time.perf_counter() - t0) / nb * 1e3
            sn.prefetch_stop(); sn.close()
            one_logical = {"ms_per_global_batch": ms_g, "bound_triplets_per_s": Bg * NN / (ms_g * 1e-3),
                           "note": "--sampler node: one logical sampler of the global batch (%d stage thread(s)), measured here "
                                   "without the GPUs: the node's rate in that mode cannot exceed it" % args.sampler_threads}
        skw = dict(skw, batch_size=B_PER_GPU, rand_seed=1 + rank, initial_cursor=(rank * N_VIDEOS) // world)
        sampler = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **skw)
        sampler.prefetch_start(depth=args.prefetch_depth, threads=args.sampler_threads, shm_name=None, consumers=1)
        ring = sampler.ring()
        ring_consumer = 0
        item_begin = 0
        if dist: dist.barrier()
    else:
        # ---- the node's ONE sampler: rank 0 draws global batches ahead on prefetch threads and publishes them through a
        # ring (POSIX shared memory when there are other ranks); every rank takes items [rank*B, (rank+1)*B) of each batch.
        item_begin = rank * B_PER_GPU
        if rank == 0:
            sampler = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **skw)
            sampler.prefetch_start(depth=args.prefetch_depth, threads=args.sampler_threads,
                                   shm_name=ring_name if world > 1 else None, consumers=world)
            ring = sampler.ring()
        ring_consumer = rank
        if dist:
            dist.barrier()
            ok = 1
            if rank != 0:
                try:
                    if os.environ.get("VV_BENCH_PRIVATE_SAMPLERS") == "1":
                        raise vv.VVError("forced by VV_BENCH_PRIVATE_SAMPLERS")
                    ring = vv.BatchRing.attach(ring_name, timeout_s=120.0)
                    ring_attached = True
                except vv.VVError as e:
                    ok = 0
                    print("rank %d: cannot attach the node's batch ring (%s)" % (rank, e), file=sys.stderr)
            flag = torch.tensor([ok], dtype=torch.int32)
            flag_dev = flag.to(torch.device("cuda", local_rank))
            dist.all_reduce(flag_dev, op=dist.ReduceOp.MIN)
            if int(flag_dev.item()) == 0:
                # the shared-memory ring is not usable on this node: every rank runs the identical sampler of the global
                # batch for itself (the same indices; N samplers' worth of host work) and the line says so
                if rank == 0:
                    sampler.prefetch_stop()
                    sampler.close()
                elif ok:
                    ring.close()
                    ring_attached = False
                sampler = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **skw)
                sampler.prefetch_start(depth=args.prefetch_depth, threads=args.sampler_threads, shm_name=None, consumers=1)
                ring = sampler.ring()
                ring_consumer = 0
                sampler_note = " (fallback: one identical sampler per rank, the shared-memory ring could not be attached)"

    # batches for the resident-indices legs come from a second, identical sampler (rank-local slice of the global batch)
    n_res = 0 if (args.no_extra_legs or shipped) else S + Wm + K
    batches = None
    if n_res:
        smp2 = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **skw)
        batches = np.stack([smp2.next()[item_begin:item_begin + B_PER_GPU] for _ in range(n_res)])
        smp2.close()
    # raw rate of the sampler by itself (calling thread, no pipeline), for the record
    smp3 = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **skw)
    smp3.next()
    t0 = time.perf_counter()
    n3 = max(3, 24 // world)
    for _ in range(n3):
        smp3.next()
    sampler_serial_ms = (time.perf_counter() - t0) / n3 * 1e3
    smp3.close()

    W0, b0 = init_weights(SEED, D, F)
    # Everything (kernels of the context and the collective) is issued under ONE explicit, non-default torch stream:
    # torch orders the RCCL all-reduce after the kernels already queued on the current stream and the following kernels
    # after the all-reduce.  (The default stream's raw handle is 0, which vv_set_stream reads as "the context's own".)
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    assert work_stream.cuda_stream != 0
    stride = B_PER_GPU * (C + NN) * 4
    idx_dev = torch.from_numpy(batches).to(dev) if batches is not None else None
    want_peer = args.comm == "peer"                 # the library's communicator, over its direct peer transport
    if want_peer:
        args.comm = "lib"
    # auto (round 5): the SHARDED update in the compute stream on either transport -- on one rank over real RCCL it costs what the synchronous
    # schedule costs (0.2195 against 0.2197 ms) while `overlap` pays 20-26 us before it hides anything (profiles/r05_overlap_cost.txt), and it
    # moves 3/4 of the all-reduce's bytes; the line's `schedules` legs time all three on the node it runs on.  torch's collective is synchronous.
    mode = args.allreduce if args.allreduce != "auto" else (("sharded" if args.comm == "lib" else "sync") if world > 1 else "none")
    if world == 1 and mode in ("sync", "overlap", "sharded"):
        mode = "none"
    comm = args.comm if mode in ("sync", "overlap", "sharded") else "none"
    if mode in ("overlap", "sharded") and comm == "torch":
        raise SystemExit("--allreduce %s needs --comm lib (the chunked all-reduce / the sharded update live in the library)" % mode)
    # the library's communicator: RCCL, or the shared-memory transport under the one-device test hook
    comm_transport = "peer" if want_peer else ("rccl" if os.environ.get("VV_DIST_BACKEND", "nccl") == "nccl" else "shm")
    if want_peer and os.environ.get("VV_BENCH_TEST_PEER_AS"):        # test hook: the peer leg's communicator fails to come up (tests/test_gpu_dist.py)
        comm_transport = os.environ["VV_BENCH_TEST_PEER_AS"]
    comm_id_path = "/tmp/vv_comm_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", str(os.getppid())))

    def lr_at(it):     # shipped solver: inv policy, base 1e-3, gamma 1e-3, power .75
        return 1e-3 * (1.0 + 1e-3 * it) ** -0.75

    no_hint = os.environ.get("VV_BENCH_NO_HINT") == "1"      # (A/B: vv_update_hint off -- the update always as its own launch)
    KERNELS = ("dedup", "fwd_gemm", "score_loss", "segsum", "guard", "wgrad_gemm", "reduce", "sgd", "reduce_sgd")
    GEMMS = ("fwd_gemm",)      # timed INSIDE the timed region: the dominant kernel (the roofline's); the others on the same steps of the per-step leg

    class Run:
        """One engine + its step function; source = 'ring' (end to end) or 'resident'."""
        n_comm = 0
        comm_kind = None                          # "torch" once a run had to fall back from the library's communicator

        def __init__(self, prec, dedup, sched=None, transport=None):
            # sched / transport: a schedule leg (N > 1) -- another exact schedule, or the other transport, than the line's default
            self.prec = prec
            self.mode = mode if sched is None else sched
            self.comm = comm if sched is None else "lib"
            self.transport = comm_transport if transport is None else transport
            self.comm_error = None
            self.eng = vv.Engine(local_rank, prec)
            self.eng.set_stream(work_stream.cuda_stream)
            self.eng.table_synth(ds.seed, ds.n_rows, F)
            self.eng.params_set(W0, b0)
            self.eng.set_dedup(dedup)
            self.cfg = vv.StepConfig(B_PER_GPU, C, NN, global_count=Bg * NN)
            if DROPOUT > 0:
                self.cfg.set("dropout_ratio", DROPOUT); self.cfg.set("dropout_seed", 1701)
            self.grads = None
            self.trainer = None
            if self.mode == "stale":
                from videovector_amd.dist import GpuBackend, PipelinedTrainer
                be = GpuBackend(self.eng, self.cfg, stream=work_stream)
                self.trainer = PipelinedTrainer(be, None, NN, dist=dist, rank=rank, world=world)
            elif self.comm == "torch":
                self.grads = torch.zeros(D * F + D, dtype=torch.float32, device=dev)
                self.eng.grads_bind(self.grads.data_ptr())
            elif self.comm == "lib":
                Run.n_comm += 1                   # one communicator per engine: its own id file
                if rank == 0 and os.path.exists(comm_id_path + "_%d" % Run.n_comm):
                    os.unlink(comm_id_path + "_%d" % Run.n_comm)
                dist.barrier()
                ok = 1
                try:
                    self.eng.comm_init(world, rank, comm_id_path + "_%d" % Run.n_comm, self.transport)
                    self.eng.comm_schedule(self.mode)
                except vv.VVError as e:
                    ok = 0
                    self.comm_error = str(e)
                    print("rank %d: library communicator failed (%s)" % (rank, e), file=sys.stderr)
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    if ok:
                        self.eng.comm_destroy()
                    if sched is not None:
                        # a schedule leg whose communicator did not come up on some rank: the leg is skipped (every rank alike), the line says why
                        self.comm_error = self.comm_error or "the communicator did not come up on another rank"
                        self.comm = "failed"
                    elif os.environ.get("VV_BENCH_CHILD") == "1":
                        raise SystemExit("bench.py (leg process): the %s communicator did not come up: %s" % (self.transport, self.comm_error or "another rank failed"))
                    else:
                        # some rank could not bring the library's communicator up: every rank falls back to the collective
                        # of torch.distributed (same RCCL, same exact synchronous schedule) and the line says so
                        Run.comm_kind = "torch"
                        self.grads = torch.zeros(D * F + D, dtype=torch.float32, device=dev)
                        self.eng.grads_bind(self.grads.data_ptr())
            self.it = 0

        def reset(self, dedup):
            self.eng.set_dedup(dedup)
            self.eng.params_set(W0, b0)
            self.it = 0

        def step(self, source, i):
            eng, cfg = self.eng, self.cfg
            cfg.set("lr", lr_at(self.it))
            if self.trainer is not None:
                assert source == "resident", "--allreduce stale runs on resident indices"
                self.trainer.step(lr_at(self.it), global_batch=Bg, idx_dev_ptr=idx_dev.data_ptr() + i * stride)
            else:
                if self.comm == "none" and not no_hint:
                    eng.update_hint(cfg)           # Solver::Step as one unit (include/videovec.h): with one split of K the update rides in the weight-gradient GEMM
                if source == "q1":
                    idx_h, last_h, _ = sampler.next(want_last=True)
                    eng.forward_backward_q1(cfg, idx_h, last_h)
                elif source == "ring":
                    th0 = time.perf_counter()
                    eng.forward_backward_ring(cfg, ring, consumer=ring_consumer, item_begin=item_begin)
                    if diag: host_ms.append((time.perf_counter() - th0) * 1e3)
                else:
                    eng.forward_backward(cfg, idx_dev_ptr=idx_dev.data_ptr() + i * stride, idx_ready=True)   # uploaded and synchronised at set-up
                if self.comm == "torch" or (self.grads is not None and Run.comm_kind == "torch"):
                    dist.all_reduce(self.grads)
                elif self.comm == "lib":
                    eng.allreduce_grads()          # (vv_apply_update would call it too; explicit for the reader)
                eng.apply_update(cfg)
            self.it += 1

        def timed(self, source, per_step_events=False, profile=True):
            """S settle steps, W warm-up steps, then exactly K steps between barrier + synchronize; max over ranks."""
            eng = self.eng
            # Everything that takes host time happens HERE, before the settle steps: between the synchronisation that ends the
            # warm-up and the first timed step the device must not sit idle (tens of ms of idling -- a collector run, thousands of
            # event objects -- and it is back at the bottom of its clock ramp: a 20-step region then measures 0.26 ms, not 0.23).
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)] if per_step_events else None
            gc.collect()
            gc.disable()             # (as timeit does: a collector pause of the Python driver loop is not the path's time)
            cold = [torch.cuda.Event(enable_timing=True) for _ in range(min(S, 25))]
            for i in range(S):
                if i < len(cold): cold[i].record(work_stream)
                self.step(source, i)
            for i in range(S, S + Wm):
                self.step(source, i)
            if self.trainer is not None:
                self.trainer.flush()
            if dist: dist.barrier()
            torch.cuda.synchronize()
            # Kernel durations: HIP events stamped by the kernels' own dispatch packets inside the timed region, on every
            # prof_every-th step (a timed dispatch cannot be pipelined behind its predecessor, ~5 us each).
            prof_every = int(os.environ.get("VV_BENCH_PROF_EVERY", "0")) or max(5, K // 8)
            if profile:
                eng.profile_select(GEMMS if profile == "gemm" else None)
                eng.profile_enable(prof_every)
            t0 = time.perf_counter()
            if evs: evs[0].record(work_stream)
            for i in range(S + Wm, S + Wm + K):
                self.step(source, i)
                if evs: evs[i - S - Wm + 1].record(work_stream)
            if self.trainer is not None:
                self.trainer.flush()
            torch.cuda.synchronize()
            if dist: dist.barrier()
            el = time.perf_counter() - t0
            gc.enable()
            if dist:
                t = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            kern = {k: eng.profile_get(k) for k in KERNELS} if profile else {}
            if profile:
                eng.profile_enable(False)
            # (with kernel timing on, the steps that carried timed dispatches are left out of the per-step series)
            steps_ms = [evs[j].elapsed_time(evs[j + 1]) for j in range(K)
                        if not (profile and j % prof_every == prof_every - 1)] if evs else None
            self.cold_ms = [cold[j].elapsed_time(cold[j + 1]) for j in range(len(cold) - 1)]
            return el, kern, steps_ms

    def stats(ms):
        a = np.sort(np.asarray(ms))
        return {"min": float(a[0]), "median": float(np.median(a)), "p95": float(a[int(0.95 * (len(a) - 1))]), "max": float(a[-1]),
                "mean": float(a.mean()), "n": len(a)}

    diag = os.environ.get("VV_BENCH_DIAG") == "1"
    host_ms = []
    t_setup_done = time.perf_counter()
    main_source = "resident" if mode == "stale" else ("q1" if shipped else "ring")
    if os.environ.get("VV_BENCH_SOURCE") == "resident":       # debugging aid: the main leg on resident indices (needs the extra legs' batches)
        main_source = "resident"
    # --allreduce auto, N > 1: the schedule is chosen from what THIS node measures, in this invocation -- the three exact
    # schedules, a fresh engine and communicator each, the same K steps on the same resident batches, max over ranks -- not from one-rank costs.
    # Every rank sees the same (all-reduced) times, so every rank picks the same schedule.
    pre_legs, auto_pick = None, None
    if world > 1 and args.allreduce == "auto" and comm == "lib" and not args.no_extra_legs and args.workload == "cfg2" and batches is not None:
        pre_legs = {}
        for sched in ("sync", "overlap", "sharded"):
            r2 = Run(args.prec, args.dedup == "on", sched=sched)
            if r2.comm == "failed":
                pre_legs[sched] = {"error": r2.comm_error}
            else:
                l_el, _, _ = r2.timed("resident", profile=False)
                pre_legs[sched] = {"ms_per_step": l_el / K * 1e3, "value": Bg * NN * K / l_el, "final_loss": r2.eng.loss()[0], "source": "resident indices"}
            r2.eng.close()
        ok_legs = {k: v["ms_per_step"] for k, v in pre_legs.items() if "error" not in v}
        if ok_legs:
            mode = min(ok_legs, key=ok_legs.get)
            auto_pick = {"picked": mode, "ms_per_step_of_the_legs": ok_legs,
                         "rule": "--allreduce auto: the fastest of the exact schedules as measured by this invocation, before the timed run"}
    run = Run(args.prec, args.dedup == "on")
    # The timed leg times the two GEMMs only when the other kernels' durations come from the per-step leg below (cfg 2: a GEMM
    # is the dominant kernel by 3x; a timed dispatch costs ~5 us of stream time, 7 of them on every 4th of 20 steps 4 %).
    main_prof = "gemm" if (not args.no_extra_legs and args.workload == "cfg2") else "all"
    if shipped:
        S = max(1, int(args.settle_ms / 0.5))
    # Box calibration: two fixed probes of the library -- the benchmark's forward GEMM instantiation on contiguous random
    # rows, a 1 GiB streaming copy -- right before the settle steps of the timed leg and right after it: what THIS device delivers now.
    box = None
    if rank == 0 and os.environ.get("VV_BENCH_NO_BOX") != "1":
        box = {"before": run.eng.box_probe()}
    elapsed, kern, diag_ms = run.timed(main_source, per_step_events=diag, profile=main_prof)
    if box is not None:
        box["after"] = run.eng.box_probe()
    main_cold_ms = run.cold_ms
    if diag and rank == 0:
        print("main-leg step ms: " + " ".join("%.3f" % x for x in diag_ms), file=sys.stderr)
        print("host ms in forward_backward_ring (warm-up included): " + " ".join("%.3f" % x for x in host_ms), file=sys.stderr)
    t_main_done = time.perf_counter()
    loss, viol = run.eng.loss()
    extra = {}
    if not args.no_extra_legs:
        run.reset(args.dedup == "on")
        g_el, g_kern, _ = run.timed("resident", profile=False)
        extra["gpu_path_only"] = {"value": Bg * NN * K / g_el, "unit": "triplets/s", "ms_per_step": g_el / K * 1e3,
                                  "note": "index batches resident in HBM before the timed region: sampler, ring and H2D excluded"}
        extra["gpu_path_only"]["final_loss"] = run.eng.loss()[0]
        run.reset(args.dedup == "on")
        _, s_kern, steps_ms = run.timed(main_source, per_step_events=True, profile="all")
        extra["step_ms_stats"] = dict(stats(steps_ms), note="end-to-end steps, one HIP event per step (a separate run of the same K "
                                                             "steps; the steps whose kernels were individually timed left out)")
        if main_prof == "gemm":
            for k, v in s_kern.items():
                if k not in GEMMS:
                    kern[k] = v
        if args.dedup == "on" and (run.eng.get_option("h16") or run.eng.get_option("slab16")):
            # round 6: two 16-bit intermediates (options "h16": ip2 between the forward GEMM and the segment-wise pair as f16; "slab16": the
            # weight gradient's split-K partial products as f16 x a power of two per tile) -- the same steps with both as
            # fp32, the rounds 1-5 arithmetic, beside the line
            h16_was, s16_was = run.eng.get_option("h16"), run.eng.get_option("slab16")
            run.reset(True)
            run.eng.set_option("h16", 0); run.eng.set_option("slab16", 0)
            f_el, f_kern, _ = run.timed("resident")
            run.eng.set_option("h16", h16_was); run.eng.set_option("slab16", s16_was)
            extra["fp32_intermediates_execution"] = {"value": Bg * NN * K / f_el, "unit": "triplets/s", "ms_per_step": f_el / K * 1e3,
                                                    "source": "resident indices; options h16 = 0 and slab16 = 0 (VV_H16=0 VV_SLAB16=0): ip2 rows and split-K partial products as fp32",
                                                    "kernels_ms": {k: round(v[0], 4) for k, v in f_kern.items() if v[1] > 0}}
        if args.dedup == "on":
            run.reset(False)
            d_el, d_kern, _ = run.timed("resident")
            extra["dense_execution"] = {"value": Bg * NN * K / d_el, "unit": "triplets/s", "ms_per_step": d_el / K * 1e3,
                                        "source": "resident indices",
                                        "kernels_ms": {k: round(v[0], 4) for k, v in d_kern.items() if v[1] > 0}}
        if args.workload == "cfg2":
            # "a dropout-0.9 number reported separately" -- the same steps with the shipped dropout ratio, mask from a
            # counter-based generator.  drop2 sits behind fc7 + ReLU (mednet_embedding_train.prototxt:190-230): the projection of equal
            # rows is equal, only the mask is per instance -- so dropout rides the de-duplicated path (every instance masks its shared row
            # in the score kernel, the segment kernel sums mask-weighted terms); the dense execution (mask in the forward GEMM's epilogue,
            # per-instance gradient rows: what the reference's schedule amounts to) is timed beside it.
            run.reset(args.dedup == "on")
            run.cfg.set("dropout_ratio", 0.9); run.cfg.set("dropout_seed", 1701)
            p_el, p_kern, _ = run.timed("resident")
            p_rows, p_uniq = run.eng.dedup_stats()
            extra["dropout_execution"] = {"value": Bg * NN * K / p_el, "unit": "triplets/s", "ms_per_step": p_el / K * 1e3,
                                          "dropout_ratio": 0.9, "source": "resident indices; " + ("de-duplicated execution (%.2f rows per distinct row): the shared "
                                          "projection once per distinct row, the mask per instance" % (p_rows / max(p_uniq, 1)) if p_uniq < p_rows else "dense execution"),
                                          "kernels_ms": {k: round(v[0], 4) for k, v in p_kern.items() if v[1] > 0}}
            if args.dedup == "on":
                run.reset(False)
                q_el, q_kern, _ = run.timed("resident")
                extra["dropout_dense_execution"] = {"value": Bg * NN * K / q_el, "unit": "triplets/s", "ms_per_step": q_el / K * 1e3, "dropout_ratio": 0.9,
                                                    "source": "resident indices; dense execution (every sampled row projected, mask in the GEMM's epilogue)",
                                                    "kernels_ms": {k: round(v[0], 4) for k, v in q_kern.items() if v[1] > 0}}
            run.cfg.set("dropout_ratio", 0.0)
        first_inline, sharded_inline = bool(run.eng.get_option("comm_first_inline")), bool(run.eng.get_option("comm_inline"))
        run.eng.close()
        if world > 1 and comm == "lib" and mode in ("sync", "overlap", "sharded") and args.workload == "cfg2":
            # the three exact schedules (and, below, the other transport) in ONE invocation -- the same K steps on the same
            # resident batches, a fresh engine and communicator each.  `value` stays the default schedule's; every leg's final loss must be the
            # default's, bit for bit (the schedules reduce the same numbers in the same order).
            if pre_legs is not None:
                legs = pre_legs                                   # (timed up front: --allreduce auto picked from them)
                legs[mode]["default"] = True
            else:
                legs = {mode: {"ms_per_step": extra["gpu_path_only"]["ms_per_step"], "value": extra["gpu_path_only"]["value"],
                               "final_loss": extra["gpu_path_only"]["final_loss"], "default": True, "source": "resident indices (= gpu_path_only)"}}
            for sched in ("sync", "overlap", "sharded"):
                if sched in legs:
                    continue
                r2 = Run(args.prec, args.dedup == "on", sched=sched)
                if r2.comm == "failed":
                    legs[sched] = {"error": r2.comm_error}
                else:
                    l_el, _, _ = r2.timed("resident", profile=False)
                    legs[sched] = {"ms_per_step": l_el / K * 1e3, "value": Bg * NN * K / l_el, "final_loss": r2.eng.loss()[0], "source": "resident indices"}
                r2.eng.close()
            extra["schedules"] = {"transport": comm_transport, "legs": legs, "auto": auto_pick,
                                  "note": "the exact schedules of DESIGN.md 8 on the same batches, K steps each behind the same settle + warm-up steps: "
                                          "sync = whole-buffer all-reduce between backward and update; overlap = the update F-chunk by F-chunk on the "
                                          "communication stream beside the next forward GEMM (gated per chunk)%s; sharded = reduce-scatter, the rule on "
                                          "D / N rows, all-gather of the 16-bit copy, %s"
                                          % (", the first chunk queued in the compute stream" if first_inline else "",
                                             "in the compute stream" if sharded_inline else "on the communication stream, the next forward GEMM gated on one flag")}
    # the other operand type, end to end (configs[4] is quoted for bf16 operands while the product defaults to f16: --workload cfg5 always
    # shows both; their parity bounds against the fp32 CPU path are tests/test_gpu_cfg5.py's -- f16 1e-3, bf16 4e-3 on the embeddings)
    if (not args.no_extra_legs or args.workload == "cfg5") and not shipped:
        other = "bf16" if args.prec == "f16" else "f16"
        if mode != "stale":
            run2 = Run(other, args.dedup == "on")
            o_el, o_kern, _ = run2.timed("ring")
            extra[other + "_execution"] = {"value": Bg * NN * K / o_el, "unit": "triplets/s", "ms_per_step": o_el / K * 1e3,
                                           "source": "end to end (sampler prefetch + H2D inside the timed region)",
                                           "kernels_ms": {k: round(v[0], 4) for k, v in o_kern.items() if v[1] > 0}}
            if args.workload == "cfg5":
                extra[other + "_execution"]["parity_vs_fp32_cpu_path"] = ("tests/test_gpu_cfg5.py::test_cfg5_whole_per_gpu_batch_against_the_oracle (all 104 960 rows of one rank's "
                                                                          "batch): embeddings <= 1e-3 and scores <= 1e-3 with f16 operands (the north star's bound), <= 4e-3 / 2e-3 with bf16")
            run2.eng.close()
    # N > 1: the legs that run as FRESH processes (one per rank, started by the helper each rank launched before its first GPU call), under a
    # time limit: the direct peer transport -- which on a node it has never run on may fail to map, or not come back -- and the ONE-logical-
    # sampler arrangement (SURVEY 8e's batch) beside the per-rank samplers.  A leg that fails leaves its error string, not a dead job.
    child = {}
    if leg_helper is not None:
        base_port = int(os.environ.get("MASTER_PORT", "29500"))
        rid = os.environ.get("TORCHELASTIC_RUN_ID", str(os.getppid()))
        common = ["--gpus", str(world), "--steps", str(K), "--warmup", str(Wm), "--no-extra-legs", "--no-cpu-baseline", "--prec", args.prec,
                  "--dedup", args.dedup, "--settle-ms", str(args.settle_ms), "--sampler-threads", str(args.sampler_threads),
                  "--prefetch-depth", str(args.prefetch_depth), "--cpu-bind", args.cpu_bind]
        want = []
        if comm == "lib" and not want_peer and mode in ("sync", "overlap", "sharded"):
            want.append(("peer_transport_leg", common + ["--comm", "peer", "--allreduce", "sharded", "--sampler", smode], 101))
        if smode == "rank" and mode in ("sync", "overlap", "sharded"):
            want.append(("node_sampler_leg", common + ["--comm", "peer" if want_peer else args.comm, "--allreduce", mode, "--sampler", "node"], 202))
        limit = float(os.environ.get("VV_BENCH_CHILD_TIMEOUT", "240"))
        for name, argv, off in want:
            if dist: dist.barrier()
            t_leg = time.perf_counter()
            # (under torch.distributed.run the AGENT hosts the rendezvous store of the job; the leg's own job has no agent: its rank 0 hosts one)
            child[name] = leg_helper.run(argv, {"MASTER_PORT": str(base_port + off), "TORCHELASTIC_RUN_ID": rid + "_" + name, "VV_BENCH_CHILD": "1",
                                                "TORCHELASTIC_USE_AGENT_STORE": "False",
                                                "VV_COMM_TIMEOUT": os.environ.get("VV_COMM_TIMEOUT", "30")}, limit)
            child[name]["wall_s"] = round(time.perf_counter() - t_leg, 2)
        leg_helper.close()
    if shipped and not no_hint:
        # vv_update_hint + option "wgrad_update" (default): the solver's rule applied in the weight-gradient GEMM's epilogue (one split of K at this
        # shape) -- bit-identical parameters.  Timed beside the line: the update as its own launch (option 0), the form of rounds 3-4.
        run3 = Run(args.prec, args.dedup == "on")
        run3.eng.set_option("wgrad_update", 0)
        w_el, w_kern, _ = run3.timed("q1")
        extra["update_as_own_launch_execution"] = {"value": Bg * NN * K / w_el, "unit": "triplets/s", "ms_per_step": w_el / K * 1e3,
                                                   "kernels_ms": {k: round(v[0], 4) for k, v in w_kern.items() if v[1] > 0},
                                                   "final_loss": run3.eng.loss()[0],
                                                   "note": "option wgrad_update = 0 (VV_WGRAD_UPDATE=0): weight-gradient GEMM stores dW, k_reduce_sgd applies the rule (the line itself: the update in the GEMM's epilogue)"}
        run3.eng.close()
    t_legs_done = time.perf_counter()

    if rank == 0:
        ms = elapsed / K * 1e3
        value = Bg * NN * K / elapsed
        R = B_PER_GPU * (C + NN)
        dense_flop = 2.0 * R * F * D         # per launch of either GEMM kernel, every sampled row (SURVEY 8d figure)
        # rows the GEMMs really processed: distinct table rows per timed batch of this rank (host recount on the identical
        # stream of the second sampler; the ring's batches are the same batches)
        if DROPOUT > 0:
            U = float(R)                      # the shipped shape under dropout runs dense (D = 4096)
        elif args.dedup == "on" and batches is not None:
            U = float(np.mean([len(np.unique(batches[i])) for i in range(S + Wm, S + Wm + K)]))
        elif args.dedup == "on":
            U = float(run.eng.dedup_stats()[1])
        else:
            U = float(R)
        gemm_flop = 2.0 * U * F * D
        live = {k: v[0] for k, v in kern.items() if v[1] > 0} or {"wgrad_gemm": 1e-9}
        dom = max(live, key=live.get)
        dom_ms = live[dom]
        if dom in ("fwd_gemm", "wgrad_gemm"):
            ach = gemm_flop / (dom_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / MFMA_PEAK_TFLOPS, "algorithmic_flop_per_launch": gemm_flop,
                    "dense_equivalent_flop_per_launch": dense_flop,
                    "dense_equivalent_achieved": dense_flop / (dom_ms * 1e-3) / 1e12}
        else:
            if args.dedup == "on" and DROPOUT == 0 and dom == "segsum":
                # k_seg_bwd: reads every distinct row's ip2 row (fp32) and the 16-byte record of every instance once, the two
                # vectors of every item once; writes a 16-bit gradient row per distinct row
                nbytes = U * D * 4.0 + R * 16.0 + B_PER_GPU * 2 * D * 4.0 + U * D * 2.0
            elif args.dedup == "on" and DROPOUT == 0:
                # k_score_fwd / k_score_stream: reads every instance's ip2 row (fp32) once; writes a record per instance and two
                # vectors per item
                nbytes = R * D * 4.0 + R * 16.0 + B_PER_GPU * 2 * D * 4.0
            else:
                # k_score_loss: reads every instance's ip2 row (fp32) once, writes its 16-bit gradient row once
                nbytes = R * D * 4.0 + R * D * 2.0
            ach = nbytes / (dom_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes}
        roof["avg_launch_ms"] = dom_ms
        roof["dedup_factor"] = R / U
        pmc, pmc_src, pk, pmc_live = None, None, {}, False
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        committed = {}
        if os.path.exists(pmc_path) and args.workload == "cfg2":      # (the committed PMC passes are of the cfg-2 kernels)
            try:
                pj = json.load(open(pmc_path))
                committed = pj.get("dedup_" + args.dedup, {})
                pmc_src = "profiles/pmc_latest.json (" + pj.get("_source", pj.get("source", "rocprofv3 PMC passes")) + ")"
            except Exception:
                committed = {}
        if live_pmc and dom in live_pmc:
            pk, pmc_live = live_pmc, True
        else:
            pk = committed
        pmc = pk.get(dom, {}).get("hbm_bytes_per_launch")
        roof["traffic"] = pmc
        # the whole step's HBM bytes by the same PMC passes beside what the de-duplicated algorithm needs
        if pk and args.workload == "cfg2" and args.dedup == "on" and DROPOUT == 0:
            try:
                step_k = ("fwd_gemm", "score_loss", "segsum", "wgrad_gemm", "reduce_sgd")
                pmc_step = sum(pk[k]["hbm_bytes_per_launch"] for k in step_k)
                h_b = 2.0 if (os.environ.get("VV_H16", "1") != "0") else 4.0
                s_b = 2.0 if (os.environ.get("VV_SLAB16", "1") != "0") else 4.0
                alg = (2.0 * U * F * 2 + 3.0 * U * D * h_b + 2.0 * U * D * 2 + 2.0 * 8 * D * F * s_b + 4.0 * D * F * 4 + 2.0 * D * F * 2)
                roof["step_traffic"] = {"pmc_bytes_per_step": pmc_step, "algorithmic_bytes_per_step": alg, "ratio": pmc_step / alg if alg else None,
                                        "kernels": {k: pk[k]["hbm_bytes_per_launch"] for k in step_k},
                                        "algorithmic": "2 U F 2 (rows gathered by both GEMMs) + 3 U D h (ip2 written once, read by the score and the segment "
                                                       "kernel; h = 2 B as f16) + 2 U D 2 (dYu) + 2 S D F s (split-K partial products, s = 2 B as f16) + "
                                                       "4 D F 4 (W and history, read and written) + 2 D F 2 (the 16-bit copy of W)",
                                        "note": "pmc: the sum over the step's five kernels, " + ("measured by this invocation's own PMC passes" if pmc_live else
                                                "the committed passes of profiles/pmc_latest.json, not measured by this run")}
            except Exception:
                pass
        if pmc is None:
            roof["traffic_note"] = None
        elif pmc_live:
            roof["traffic_note"] = "HBM bytes per launch of this kernel, " + live_pmc_note
            if committed.get(dom):
                roof["traffic_committed_profile"] = committed[dom].get("hbm_bytes_per_launch")
        else:
            roof["traffic_note"] = ("HBM bytes per launch from rocprofv3 PMC passes recorded in %s -- a committed profile of this kernel, NOT measured "
                                    "by this run (live passes: %s)" % (pmc_src, live_pmc_note))
        out = {
            "metric": "triplets/sec (whole node), 4096->%d-d embed, batch %d/GPU, C5, Nn%d" % (D, B_PER_GPU, NN),
            "value": value, "unit": "triplets/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.prec,
            "dtype_note": args.prec + " MFMA operands with fp32 accumulation; fp32 master weights, activations, loss and update",
            "data": "synthetic",
            "settle": {"steps": S, "cold_start_step_ms": [round(x, 4) for x in main_cold_ms],
                       "note": "untimed steps of the same workload in front of the W warm-up steps of every leg (--settle-ms, "
                               "default 50 ms of them; 0 = none).  After an idle spell the device's step time takes ~100 steps "
                               "(~25 ms) to become steady -- cold_start_step_ms are the first of them, one HIP event per step; "
                               "the timed region is exactly `steps` full training steps either way"},
            "value_scope": ("end to end: bit-exact reference sampler on %d prefetch thread(s) inside the timed region, "
                            "225 KB index batch per step through pinned staging + async H2D, full training iteration"
                            % args.sampler_threads) if main_source == "ring" else
                           ("end to end: bit-exact reference sampler on the calling thread (quirk-Q1 batches: index + last-feature "
                            "arrays), composite rows patched on the device, full training iteration" if main_source == "q1"
                            else "resident indices (stale-gradient schedule)"),
            "config": {"workload": ("the reference's shipped project files (mednet_embedding_train.prototxt:13-23,200,226; INFORMATIONAL, not "
                                    "BASELINE's metric configuration): synthetic fc7 4096-d -> %d-d, batch %d, context_size 5, %d negatives "
                                    "(max_same_video_negs 6: quirk Q1), dropout 0.9, max_buffer 5000, swap 50%%, margin 2 L2, SGD momentum .9 "
                                    "wd 5e-4 inv lr" % (D, B_PER_GPU, NN)) if shipped else
                                   "BASELINE configs[%d] per GPU: synthetic fc7 4096-d -> %d-d, batch "
                                   "%d/GPU (global %d), context_size 5 (window +-2), %d negatives, "
                                   "max_buffer 5000, swap 50%%, margin 2 L2, SGD momentum .9 wd 5e-4 inv lr"
                                   % (1 if args.workload == "cfg2" else 4, D, B_PER_GPU, Bg, NN),
                       "global_batch": Bg, "triplets_per_step": Bg * NN,
                       "parallelism": "dp%d" % world, "items_per_s": value / NN,
                       "ip2_rows": ("f16 between the forward GEMM and the score / segment kernels (option h16; fp32 accumulate, one f16 rounding of the "
                                    "stored embedding: rows 3.5e-4, scores 5e-5 off the fp32 CPU path on whole batches, tests/test_gpu_fullsize.py)"
                                    if (args.dedup == "on" and DROPOUT == 0 and os.environ.get("VV_H16", "1") != "0" and D in (512, 1024)) else "fp32"),
                       "split_k_partials": ("f16 x one power of two per (split, 256 x 256 tile), summed in fp32 (option slab16; dW 3.5e-4 -> 4.6e-4 off the "
                                            "oracle on the same operands)" if (os.environ.get("VV_SLAB16", "1") != "0" and not shipped) else "fp32"),
                       "dedup": args.dedup if DROPOUT == 0 else "off (dropout at D = 4096: dense kernels -- the de-duplicated path carries the per-instance masks at D = 512 only, and 128-item batches hardly repeat a row)",
                       "sampler_host": ({"avx512_forms": bool(sampler.stat(7) == 1), "cores_held_for_stage_threads": int(sampler.stat(9))}
                                        if sampler is not None else None),
                       "cpu_binding_rank0": cpu_bind,
                       "sampler": ("one per rank: the reference's sampler at batch %d with srand(1 + rank) and its own starting record, "
                                   "%d stage thread(s), prefetch depth %d" % (B_PER_GPU, args.sampler_threads, args.prefetch_depth))
                                  if smode == "rank" else
                                  ("one per node (rank 0), %d stage thread(s), prefetch depth %d%s"
                                   % (args.sampler_threads, args.prefetch_depth, ", POSIX shared-memory ring" if world > 1 else "") + sampler_note),
                       "comm": {"none": "none", "lib": "the library's RCCL communicator on its own communication stream (vv_comm_*)"
                                if comm_transport == "rccl" else ("the library's direct peer exchange (hipIpc mappings, one-shot reduce-scatter / all-gather kernels)"
                                                                  if comm_transport == "peer" else "the library's shared-memory test transport (one-device hook)"),
                                "torch": "torch.distributed.all_reduce"}[Run.comm_kind or comm]
                               + (" (fallback: the library's communicator did not come up)" if Run.comm_kind == "torch" and comm == "lib" else ""),
                       "allreduce": {"none": "none (1 GPU)", "sync": "synchronous (exact SGD), exposed",
                                     "overlap": "exact SGD; the update (all-reduce, SGD, publish) runs F-chunk by F-chunk on the communication "
                                                "stream while the next step's forward GEMM runs and waits per chunk inside the kernel",
                                     "sharded": "exact SGD; reduce-scatter of the gradients, the solver's rule on this rank's D / N rows, all-gather of "
                                                "the 16-bit copy of W + the bias (3/4 of the all-reduce's wire bytes, 1/N of the update's "
                                                "traffic), queued in the compute stream",
                                     "stale": "overlapped with the next iteration's forward/backward "
                                              "(one-update delayed gradients: NOT the reference's algorithm)"}[mode]},
            "roofline": roof,
            "box": box_record(box, ms, live.get("fwd_gemm", 0.0) + live.get("wgrad_gemm", 0.0)),
            "kernels_ms": {k: round(v, 4) for k, v in live.items()},
            "kernel_timing": ("HIP events on the kernels' dispatch packets, every %d-th of the %d timed steps (%d samples per kernel)"
                              % (max(5, K // 8), K, max([v[1] for v in kern.values()] or [0])))
                             + ("; the forward GEMM (the dominant kernel: the roofline's) inside the timed region itself, the other kernels "
                                "on the same steps of the per-step leg (step_ms_stats)" if main_prof == "gemm" else ""),
            "dedup": {"mode": args.dedup, "rows_per_step": R, "distinct_rows_per_step": U, "factor": R / U,
                      "note": "the reference sampler draws all negatives of a batch from one shared 5000-frame "
                              "buffer, so sampled rows repeat; each distinct row is projected once and its "
                              "instances' gradient rows are summed before the weight-gradient GEMM"},
            "step_tflops_executed": 2 * gemm_flop / (ms * 1e-3) / 1e12,
            "step_tflops_dense_equivalent": 2 * dense_flop / (ms * 1e-3) / 1e12,
            "gather_GBs_dense_equivalent": 2.0 * R * F * 2 / (ms * 1e-3) / 1e9,
            "gather_GBs_executed": 2.0 * U * F * 2 / (ms * 1e-3) / 1e9,
            "gather_note": "feature rows read by the two GEMMs per second: dense-equivalent = every sampled row (2 R F 2 B per step, "
                           "SURVEY 8d's reference-equivalent figure); executed = the distinct rows the kernels really gathered (2 U F 2 B)",
            "sampler_ms_per_global_batch_one_thread": sampler_serial_ms if smode == "node" else None,
            "sampler_ms_per_rank_batch_one_thread": sampler_serial_ms if smode == "rank" else None,
            "final_loss": loss, "final_violations": viol,
        }
        out.update(extra)
        for name, rep in child.items():
            if rep.get("line"):
                cj = json.loads(rep["line"])
                out[name] = {"ms_per_step": cj["ms_per_step"], "value": cj["value"], "final_loss": cj["final_loss"],
                             "allreduce": cj["config"]["allreduce"], "comm": cj["config"]["comm"], "sampler": cj["config"]["sampler"],
                             "wall_s": rep.get("wall_s"),
                             "source": "a fresh process per rank (started by each rank's helper before the rank's first GPU call), end to end like `value`"}
            else:
                out[name] = {"error": (rep.get("err_tail") or "no line").strip()[-400:], "rc": rep.get("rc"), "wall_s": rep.get("wall_s")}
        if one_logical is not None:
            out["one_logical_sampler"] = one_logical
        if world > 1:
            # the two forms SURVEY 8(e) / the north star describe, as first-class numbers beside `value`, and a one-line
            # statement of how the line's own configuration deviates from either
            dev = []
            ov = (out.get("schedules") or {}).get("legs", {}).get("overlap")
            if mode == "overlap":
                out["value_overlap_schedule"] = {"value": value, "ms_per_step": ms, "source": "this line's own schedule (end to end)"}
            elif ov and "error" not in ov:
                out["value_overlap_schedule"] = {"value": ov["value"], "ms_per_step": ov["ms_per_step"], "final_loss": ov["final_loss"],
                                                 "source": "schedules.legs.overlap: the same K steps on resident batches, a fresh engine and communicator"}
            else:
                out["value_overlap_schedule"] = {"value": None, "error": (ov or {}).get("error", "not run (--no-extra-legs, or another --comm)")}
            if mode != "overlap":
                dev.append("schedule %s, not the north star's second-stream overlap (value_overlap_schedule)%s"
                           % (mode, "; picked by --allreduce auto as the fastest exact schedule measured here" if auto_pick else ""))
            ns = out.get("node_sampler_leg")
            if smode == "node":
                out["value_reference_batch"] = {"value": value, "ms_per_step": ms, "source": "this line's own sampler arrangement (one logical sampler per node)"}
            elif ns and "error" not in ns:
                out["value_reference_batch"] = {"value": ns["value"], "ms_per_step": ns["ms_per_step"], "final_loss": ns["final_loss"],
                                                "source": "node_sampler_leg: ONE logical sampler draws the global batch of %d as a single reference process would "
                                                          "(video_sampled_shots_data_layer.cpp:768-909), end to end, fresh processes" % Bg}
            else:
                out["value_reference_batch"] = {"value": None, "error": (ns or {}).get("error", "not run (--no-extra-legs)")}
            if smode != "node":
                dev.append("per-rank samplers (N independent reference batches of %d), not SURVEY 8(e)'s one logical batch of %d "
                           "(value_reference_batch; bound by the sampler's serial walk)" % (B_PER_GPU, Bg))
            out["config"]["deviation"] = "; ".join(dev) if dev else None
        # where the wall-clock time of this process goes besides the K timed steps (for whoever times the whole command)
        out["wall_s"] = {"imports_setup_presampling_table": round(t_setup_done - t_process_start, 3),
                         "warmup_plus_timed_steps": round(t_main_done - t_setup_done, 3),
                         "extra_legs": round(t_legs_done - t_main_done, 3)}
        if world == 1 and not args.no_cpu_baseline and not shipped:
            b0_idx = batches[0] if batches is not None else None
            if b0_idx is None:
                s4 = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **skw); b0_idx = s4.next(); s4.close()
            full = B_PER_GPU if args.workload == "cfg2" else 256
            out["cpu_baseline"] = cpu_baseline(ds, b0_idx, W0, b0, items=full, iters=3)
            out["cpu_baseline_1_thread"] = cpu_baseline(ds, b0_idx, W0, b0, items=32 if args.workload == "cfg2" else 8,
                                                         iters=2, threads=1)
            out["wall_s"]["cpu_baselines"] = round(time.perf_counter() - t_legs_done, 3)
        print(json.dumps(out))
    if dist:
        dist.barrier()
    if ring_attached:
        ring.close()
    if sampler is not None:
        sampler.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
