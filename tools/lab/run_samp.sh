lscpu | grep -E "Model name|Thread|Core|Socket|NUMA node[0-9]" | head -12
echo "--- A default"; python tools/lab/samp_stages.py 2>&1 | grep "B  1024 threads 3"
echo "--- B caller on cpu 8, stages on 9,10,11"; VV_SAMPLER_CPUS=9,10,11 taskset -c 8 python tools/lab/samp_stages.py 2>&1 | grep "B  1024 threads 3"
echo "--- C caller on cpu 8, stages on 9-15"; VV_SAMPLER_CPUS=9,10,11,12,13,14,15 taskset -c 8 python tools/lab/samp_stages.py 2>&1 | grep "B  1024 threads 3"
echo "--- D no pin"; VV_SAMPLER_PIN=0 python tools/lab/samp_stages.py 2>&1 | grep "B  1024 threads 3"
echo "--- E caller 8, stages 9,10 (2 threads)"; VV_SAMPLER_CPUS=9,10 taskset -c 8 python tools/lab/samp_stages.py 2>&1 | grep "B  1024 threads 2"
