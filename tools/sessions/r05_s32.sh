#!/bin/bash
# Round-5 session 32: sustained run on the round's final code: 100 000 steps of bench.py (sampler pipeline, hand-over, GPU step), twice (bit-identical
# final loss expected), and the caffe-train soak.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for i in 1 2; do
timeout 900 python bench.py --steps 100000 --warmup 100 --no-cpu-baseline --no-extra-legs > $O/r05_s32_long_$i.json 2> $O/r05_s32.err
python3 -c "
import json
d=json.loads([x for x in open('gpurun_out/r05_s32_long_$i.json') if x.startswith('{')][-1]); s=d.get('step_ms_stats',{}); print('run $i: 100000 steps', round(d['ms_per_step'],4), 'ms/step', round(d['value']/1e6,1), 'M triplets/s; final loss', d['final_loss'], 'violations', d['final_violations'], '; step stats', {k:s[k] for k in list(s)[:8]})"
done
bash tools/soak.sh
