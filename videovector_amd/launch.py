"""One process per GPU without an external launcher.

`python bench.py --gpus N` (no WORLD_SIZE in the environment) calls launch_ranks(): the calling process -- which has NOT
imported torch or touched HIP, and never exec()s -- starts N child processes of the same script with the environment
torch.distributed.run would give them (RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT /
TORCHELASTIC_RUN_ID), lets rank 0 write to its own stdout (the ONE JSON line), sends the other ranks' stdout to stderr,
and returns the worst exit code.  A rank that dies takes the job down: the remaining ranks (normally stuck in a
collective waiting for it) get a grace period, then SIGTERM, then SIGKILL -- by PID, never by pattern.  A launcher that is
itself told to stop (SIGTERM / SIGHUP / SIGINT: `timeout 600 python bench.py --gpus 8` ends that way) stops its ranks
first: the handlers turn the signal into SystemExit so that the `finally` below runs; every rank is the leader of its own
session (what it started dies with it: killpg) and asks the kernel for SIGTERM should the launcher vanish without running
anything (SIGKILL, a crash: prctl PR_SET_PDEATHSIG).  The reference has nothing to mirror here (single process, single
device: src/caffe/common.cpp:127-145).
"""
import ctypes
import os
import signal
import socket
import subprocess
import sys
import time


def _free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, run_id, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCHELASTIC_RUN_ID=run_id,
               HSA_ENABLE_IPC_MODE_LEGACY=env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return env


# prctl is bound ONCE, at import time in the launcher: between fork and exec the child must not dlopen or allocate (the launcher may have
# other threads holding the loader's or malloc's lock at the moment of the fork) -- it only calls the function already resolved
try:
    _PRCTL = ctypes.CDLL(None, use_errno=True).prctl
    _PRCTL.argtypes = [ctypes.c_int, ctypes.c_ulong, ctypes.c_ulong, ctypes.c_ulong, ctypes.c_ulong]
    _PRCTL.restype = ctypes.c_int
except Exception:                                 # no libc prctl (not Linux): ranks are then stopped by the launcher's own handlers only
    _PRCTL = None
_PDEATHSIG_MSG = b"[vv launch] prctl(PR_SET_PDEATHSIG) failed in a rank: it will not be signalled should the launcher be killed\n"


def _child_setup():
    """In the rank, between fork and exec: own session; SIGTERM when the launcher dies."""
    os.setsid()
    if _PRCTL is not None and _PRCTL(1, int(signal.SIGTERM), 0, 0, 0) != 0:      # PR_SET_PDEATHSIG = 1
        os.write(2, _PDEATHSIG_MSG)               # (the launcher's stderr: a raw write, nothing that allocates or locks)


class _Stopped(SystemExit):
    pass


def _raise_stopped(signum, frame):
    raise _Stopped(128 + signum)


def launch_ranks(script, argv, world, timeout_s=None, grace_s=None, log=sys.stderr):
    """Runs `python script argv...` as `world` ranks; -> exit code (0 only if every rank exited 0)."""
    timeout_s = float(os.environ.get("VV_LAUNCH_TIMEOUT", "1500")) if timeout_s is None else timeout_s
    grace_s = float(os.environ.get("VV_LAUNCH_GRACE", "15")) if grace_s is None else grace_s
    port = _free_port()
    run_id = "vv%d_%d" % (os.getpid(), port)
    procs = []
    old_handlers = {}
    try:
        for sig in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
            try:
                old_handlers[sig] = signal.signal(sig, _raise_stopped)
            except ValueError:                    # not the main thread: the caller keeps its own handling
                pass
        for r in range(world):
            procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=rank_env(r, world, port, run_id),
                                          stdout=None if r == 0 else log, preexec_fn=_child_setup))
        t0 = time.monotonic()
        first_bad = None              # (rank, code) of the first rank seen to fail
        t_bad = None
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            if first_bad is None:
                for r, c in enumerate(codes):
                    if c not in (None, 0):
                        first_bad, t_bad = (r, c), time.monotonic()
                        print("launch: rank %d exited with code %d; giving the other ranks %.0f s" % (r, c, grace_s),
                              file=log, flush=True)
                        break
            now = time.monotonic()
            if first_bad is not None and now - t_bad > grace_s:
                break
            if now - t0 > timeout_s:
                print("launch: %d ranks did not finish within %.0f s" % (world, timeout_s), file=log, flush=True)
                first_bad = first_bad or (-1, 124)
                break
            time.sleep(0.05)
    except _Stopped as e:
        print("launch: signal %d: stopping %d rank(s)" % (e.code - 128, sum(p.poll() is None for p in procs)), file=log, flush=True)
        raise
    finally:
        for sig in old_handlers:                  # a second signal while the ranks are being stopped must not abort the stopping
            signal.signal(sig, signal.SIG_IGN)
        _stop(procs, log)
        for sig, h in old_handlers.items():
            signal.signal(sig, h)
    codes = [p.returncode for p in procs]
    if first_bad is not None:
        return first_bad[1] if first_bad[1] > 0 else 1
    worst = 0
    for r, c in enumerate(codes):
        if c != 0:
            print("launch: rank %d exited with code %s" % (r, c), file=log, flush=True)
            worst = worst or (c if c and c > 0 else 1)
    return worst


def _signal_group(p, sig):
    """The rank and whatever it started (it leads its own session / process group)."""
    try:
        os.killpg(p.pid, sig)
    except OSError:
        try:
            p.send_signal(sig)
        except OSError:
            pass


def _stop(procs, log):
    live = [p for p in procs if p.poll() is None]
    for p in live:
        _signal_group(p, signal.SIGTERM)
    t0 = time.monotonic()
    while any(p.poll() is None for p in live) and time.monotonic() - t0 < 5.0:
        time.sleep(0.05)
    for p in live:
        if p.poll() is None:
            _signal_group(p, signal.SIGKILL)
            p.wait()
    if live:
        print("launch: stopped %d rank(s) still running: pids %s" % (len(live), [p.pid for p in live]), file=log, flush=True)
