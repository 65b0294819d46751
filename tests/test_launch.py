"""videovector_amd/launch.py -- `python bench.py --gpus N` without a launcher starts its own ranks.  CPU tests of the
launcher itself (environment of the ranks, relay of rank 0's stdout, exit codes, a dead rank does not hang the job);
the real thing on a GPU is tests/test_gpu_dist.py::test_bench_bare_command_launches_its_own_ranks."""
import io
import json
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from videovector_amd.launch import launch_ranks, rank_env  # noqa: E402


def _script(tmp_path, body):
    p = tmp_path / "child.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def _run(tmp_path, body, world, argv=(), **kw):
    """launch_ranks in a subprocess (it hands rank 0 the caller's real stdout): -> (rc, stdout, stderr)"""
    script = _script(tmp_path, body)
    code = ("import sys; sys.path.insert(0, %r); from videovector_amd.launch import launch_ranks; "
            "sys.exit(launch_ranks(%r, %r, %d, **%r))" % (ROOT, script, list(argv), world, kw))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    return r.returncode, r.stdout, r.stderr


def test_rank_env_matches_what_torchrun_sets():
    e = rank_env(3, 8, 29400, "job", base={})
    assert e["RANK"] == "3" and e["LOCAL_RANK"] == "3" and e["WORLD_SIZE"] == "8" and e["LOCAL_WORLD_SIZE"] == "8"
    assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29400" and e["TORCHELASTIC_RUN_ID"] == "job"
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_every_rank_runs_and_only_rank0_reaches_stdout(tmp_path):
    rc, out, err = _run(tmp_path, """
        import json, os, sys
        print(json.dumps({k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")} | {"argv": sys.argv[1:]}))
        """, 4, argv=["--gpus", "4", "--steps", "3"])
    assert rc == 0, err
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["RANK"] == "0" and d["WORLD_SIZE"] == "4" and d["MASTER_ADDR"] == "127.0.0.1" and d["argv"] == ["--gpus", "4", "--steps", "3"]
    others = sorted(json.loads(l)["RANK"] for l in err.splitlines() if l.startswith("{"))
    assert others == ["1", "2", "3"]                      # the other ranks' stdout is on stderr
    ports = {json.loads(l)["MASTER_PORT"] for l in (out + err).splitlines() if l.startswith("{")}
    assert len(ports) == 1


def test_a_failing_rank_fails_the_job_with_its_code(tmp_path):
    rc, out, err = _run(tmp_path, """
        import os, sys
        sys.exit(5 if os.environ["RANK"] == "1" else 0)
        """, 2)
    assert rc == 5 and "rank 1 exited with code 5" in err


def test_a_dead_rank_does_not_hang_the_job(tmp_path):
    """rank 1 dies; rank 0 'waits in a collective' for ever: after the grace period it is stopped by PID."""
    t0 = time.monotonic()
    rc, out, err = _run(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            os._exit(7)
        time.sleep(600)
        """, 2, grace_s=1.0)
    assert rc == 7 and time.monotonic() - t0 < 60
    assert "rank 1 exited with code 7" in err and "stopped 1 rank(s)" in err


def test_overall_timeout(tmp_path):
    rc, out, err = _run(tmp_path, "import time; time.sleep(600)", 2, timeout_s=1.0)
    assert rc == 124 and "did not finish" in err


def _alive(pid):
    try:
        os.kill(pid, 0)
    except OSError:
        return False
    try:                                           # a zombie still answers kill(0)
        return open("/proc/%d/stat" % pid).read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def _launcher_with_sleeping_ranks(tmp_path, world):
    script = _script(tmp_path, """
        import os, sys, time
        open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pid%s" % os.environ["RANK"]), "w").write(str(os.getpid()))
        time.sleep(600)
        """)
    code = ("import sys; sys.path.insert(0, %r); from videovector_amd.launch import launch_ranks; "
            "sys.exit(launch_ranks(%r, [], %d))" % (ROOT, script, world))
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    pids = []
    t0 = time.monotonic()
    while len(pids) < world and time.monotonic() - t0 < 60:
        pids = []
        for r in range(world):
            f = tmp_path / ("pid%d" % r)
            if f.exists() and f.read_text().strip():
                pids.append(int(f.read_text()))
        time.sleep(0.05)
    assert len(pids) == world
    return p, pids


def test_a_terminated_launcher_takes_its_ranks_down(tmp_path):
    """`timeout N python bench.py --gpus 8` ends the launcher with SIGTERM; the ranks must not outlive it."""
    import signal
    p, pids = _launcher_with_sleeping_ranks(tmp_path, 3)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err)
    assert "signal 15: stopping 3 rank(s)" in err
    t0 = time.monotonic()
    while any(_alive(x) for x in pids) and time.monotonic() - t0 < 10:
        time.sleep(0.05)
    assert not any(_alive(x) for x in pids)


def test_a_killed_launcher_takes_its_ranks_down(tmp_path):
    """SIGKILL runs no handler: the ranks asked the kernel for SIGTERM at their parent's death."""
    import signal
    p, pids = _launcher_with_sleeping_ranks(tmp_path, 2)
    p.send_signal(signal.SIGKILL)
    p.communicate(timeout=60)
    t0 = time.monotonic()
    while any(_alive(x) for x in pids) and time.monotonic() - t0 < 10:
        time.sleep(0.05)
    assert not any(_alive(x) for x in pids)


def test_bench_refuses_a_world_that_contradicts_gpus():
    """(no GPU needed: the check sits in front of anything that touches the device)"""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
