#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python3 tools/samp_rates.py 10 $O/r05_s30_sampler_rates.json > $O/r05_s30_sampler_rates.txt 2>&1
timeout 300 python3 tools/lab/samp_stages.py >> $O/r05_s30_sampler_rates.txt 2>&1
cat $O/r05_s30_sampler_rates.txt
