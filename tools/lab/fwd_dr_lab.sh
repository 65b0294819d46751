# builds tools/lab/fwd_dr_lab (run here, the binary travels with gpurun)
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -Wno-unused-function fwd_dr_lab.hip -o fwd_dr_lab
