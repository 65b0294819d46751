cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do VV_BENCH_DIAG=1 python3 bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/dg_$i.json 2> gpurun_out/dg_$i.err; python3 - $i <<'PY'
import sys, json
i = sys.argv[1]
d = json.loads(open("gpurun_out/dg_%s.json" % i).read().strip().splitlines()[-1])
L = open("gpurun_out/dg_%s.err" % i).read().splitlines()
x = [float(t) for t in [l for l in L if l.startswith("main-leg step ms")][0].split(":")[1].split()]
h = [float(t) for t in [l for l in L if l.startswith("host ms")][0].split(":")[1].split()][-len(x):]
big = [(j, round(t, 3)) for j, t in enumerate(x) if t > 0.28]
bigh = [(j, round(t, 3)) for j, t in enumerate(h) if t > 0.35]
print(i, "ms/step %.4f" % d["ms_per_step"], "sum(step ev) %.2f ms" % sum(x), "n", len(x), "slow steps", big[:8], "slow host calls", bigh[:8])
PY
done
