#!/bin/bash
# Round-5 session 27: what the chip reports (power, clocks) while the step runs.
cd $GRAFT_REPO_ROOT
O=gpurun_out
{
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -30
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --json 2>/dev/null | head -c 1500; echo; sleep 0.25; done ) > $O/r05_s27_smi_during.txt 2>&1 &
SMI=$!
timeout 300 python bench.py --steps 30000 --warmup 100 --no-cpu-baseline --no-extra-legs > $O/r05_s27_bench.json 2> $O/r05_s27_bench.err
wait $SMI
python3 - <<'PY'
import json,re
rows=[]
for line in open('gpurun_out/r05_s27_smi_during.txt'):
    line=line.strip()
    if not line.startswith('{'): continue
    try: d=json.loads(line)
    except Exception: continue
    for card,v in d.items():
        rows.append({k:v[k] for k in v if 'ower' in k or 'clk' in k.lower() or 'clock' in k.lower()})
for r in rows[:40]: print(r)
PY
python3 -c "
import json
d=json.loads([x for x in open('gpurun_out/r05_s27_bench.json') if x.startswith('{')][-1]); print('bench:', round(d['ms_per_step'],4), d.get('kernels_ms'))"
} > $O/r05_s27_power.txt 2>&1
cat $O/r05_s27_power.txt | cut -c1-400
