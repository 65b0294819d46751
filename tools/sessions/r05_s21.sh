#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
{
timeout 600 python3 tools/samp_rates.py 10 $O/r05_s21_sampler_rates.json
echo "== VV_SAMPLER_REC_PTRS=0 (records carry copies of the stream words, as before)"
VV_SAMPLER_REC_PTRS=0 timeout 600 python3 tools/samp_rates.py 5
timeout 300 python3 tools/lab/samp_stages.py | grep "threads 4"
} > $O/r05_s21_sampler_rates.txt 2>&1
cat $O/r05_s21_sampler_rates.txt
timeout 600 python -m pytest tests/test_product_host.py -q -x 2>&1 | tail -2
