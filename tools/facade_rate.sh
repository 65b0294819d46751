#!/bin/bash
# iterations/s of `caffe train` on the cfg-2 example (facade end to end: sampler thread + H2D of indices + fused step)
cd $GRAFT_REPO_ROOT
sed 's/max_iter: 200/max_iter: 1000/; s/display: 20/display: 100/' examples/videovec_cfg2_solver.prototxt > /tmp/s.prototxt
VV_FACADE_PROFILE=1 caffe_facade/build/caffe train --solver=/tmp/s.prototxt --log_file=/tmp/train.log 2>&1 >/dev/null | grep "facade host profile"
python3 - <<'PY'
import re, datetime
ts = {}
for l in open('/tmp/train.log'):
    m = re.match(r"I\d{4} (\d\d:\d\d:\d\d\.\d+) .*Iteration (\d+), loss", l)
    if m:
        t = datetime.datetime.strptime(m.group(1), "%H:%M:%S.%f")
        ts[int(m.group(2))] = t
ks = sorted(ts)
a, b = ks[1], ks[-2]
dt = (ts[b] - ts[a]).total_seconds()
print("caffe train cfg2: %d iterations in %.3f s -> %.3f ms/iter = %.1f M triplets/s" % (b - a, dt, dt / (b - a) * 1e3, (b - a) * 51200 / dt / 1e6))
PY
