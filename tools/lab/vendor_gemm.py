"""What do the vendor GEMMs (hipBLASLt / rocBLAS through torch.matmul) reach on the two GEMM shapes of the step,
WITHOUT the gather (contiguous operands)?  A yardstick for k_fwd_gemm / k_wgrad_gemm, not part of the product."""
import torch
dev = torch.device("cuda:0")
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for R in (56320, 20736):
    for dt in (torch.float16, torch.bfloat16):
        X = (torch.randn(R, 4096, device=dev) * 0.5).to(dt)
        W = (torch.randn(512, 4096, device=dev) * 0.02).to(dt)
        dY = (torch.randn(R, 512, device=dev) * 0.1).to(dt)
        fl = 2.0 * R * 4096 * 512
        t = bench(lambda: X @ W.t())
        print("R=%d %s fwd   X[R,4096] @ W^T[4096,512]  : %.4f ms  %.0f TFLOP/s" % (R, dt, t, fl / t / 1e9))
        t = bench(lambda: dY.t() @ X)
        print("R=%d %s wgrad dY^T[512,R] @ X[R,4096]     : %.4f ms  %.0f TFLOP/s" % (R, dt, t, fl / t / 1e9))
