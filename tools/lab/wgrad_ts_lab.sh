# builds tools/lab/wgrad_ts_lab (run here, the binary travels with gpurun)
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -Wno-unused-function wgrad_ts_lab.hip -o wgrad_ts_lab
