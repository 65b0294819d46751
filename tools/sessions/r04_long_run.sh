#!/bin/bash
# Sustained rate: 40 000 end-to-end steps of bench.py (per-step HIP-event times by window), twice 6000 steps for
# bit-reproducibility, and a 6000-iteration caffe train of the cfg-2 example.
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
VV_BENCH_DIAG=1 timeout 600 python3 bench.py --no-cpu-baseline --no-extra-legs --steps 40000 --warmup 20 > $O/long.json 2> $O/long.err
python3 - > $O/r04_long_run.txt <<'PY'
import json, statistics
d = json.loads(open("gpurun_out/long.json").read().strip().splitlines()[-1])
line = [l for l in open("gpurun_out/long.err") if l.startswith("main-leg step ms:")][0]
x = [float(t) for t in line.split(":")[1].split()]
print("Sustained rate (1 x MI355X, defaults): VV_BENCH_DIAG=1 python3 bench.py --no-cpu-baseline --no-extra-legs --steps 40000 --warmup 20")
print("ms_per_step %.5f  (%.1f M triplets/s over 40 000 end-to-end steps = %.2e triplets), final loss %.6f" % (d["ms_per_step"], d["value"] / 1e6, 40000 * 1024 * 50, d["final_loss"]))
print("per-step HIP-event times by window of 5000 steps (steps carrying timed kernels left out):")
n = len(x)
for a in range(0, n, n // 8):
    w = sorted(x[a:a + n // 8])
    print("%6d median %.4f mean %.4f p95 %.4f max %.3f" % (a * 40000 // n, statistics.median(w), sum(w) / len(w), w[int(0.95 * (len(w) - 1))], w[-1]))
PY
for i in 1 2; do
  timeout 300 python3 bench.py --no-cpu-baseline --no-extra-legs --steps 6000 --warmup 20 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('6000-step run $i: ms_per_step %.4f final loss %.10f violations %.1f' % (d['ms_per_step'], d['final_loss'], d['final_violations']))" >> $O/r04_long_run.txt
done
sed 's/max_iter: [0-9]*/max_iter: 6000/; s/display: [0-9]*/display: 1000/' examples/videovec_cfg2_solver.prototxt > /tmp/long_solver.prototxt
( cd $GRAFT_REPO_ROOT && timeout 600 caffe_facade/build/caffe train --solver=/tmp/long_solver.prototxt 2>&1 | grep -E "Iteration [0-9]+, loss" | head -12 ) >> $O/r04_long_run.txt
cat $O/r04_long_run.txt
