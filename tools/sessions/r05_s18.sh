#!/bin/bash
# Round-5 session 18: sampler rates with a core per stage thread (default), bench N = 1 with it.
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python3 tools/samp_rates.py 10 $O/r05_s18_sampler_rates.json > $O/r05_s18_sampler_rates.txt 2>&1
echo "== VV_SAMPLER_SPREAD=0 (common set)" >> $O/r05_s18_sampler_rates.txt
VV_SAMPLER_SPREAD=0 timeout 600 python3 tools/samp_rates.py 5 >> $O/r05_s18_sampler_rates.txt 2>&1
timeout 300 python3 tools/lab/samp_stages.py >> $O/r05_s18_sampler_rates.txt 2>&1
cat $O/r05_s18_sampler_rates.txt
timeout 600 python bench.py --steps 300 --warmup 30 > $O/r05_s18_bench.json 2> $O/r05_s18_bench.err
python3 -c "
import json
d=json.loads([x for x in open('gpurun_out/r05_s18_bench.json') if x.startswith('{')][-1]); print(round(d['ms_per_step'],4), d.get('kernels_ms'), d['cpu_baseline'])"
