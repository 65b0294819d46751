"""k_score_fwd's phase stamps at BASELINE configs[1] (lab library, VV_LAB_SCORE_TS=1): per workgroup the shader clocks of
[start -> context rows + first block sum -> all rows reduced -> scores -> dAh / records -> end], the 100 MHz real time of its start and end
and the compute unit it ran on: how long a workgroup's chain is, how many rounds a CU runs, what share of the kernel is the wait for rows.
Usage: VV_LIB=videovector_amd/lib/libvideovec_lab.so VV_LAB_SCORE_TS=1 python tools/lab/score_ts.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos, init_weights

B, Cc, Nn, F, D = 1024, 5, 50, 4096, 512
ds = SyntheticVideos(seed=1701, n_videos=2048)
smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=Cc, num_negative_samples=Nn, max_buffer_size=5000, negative_swap_percentage=50)
n = 260
idx = torch.from_numpy(np.stack([smp.next() for _ in range(n)])).to("cuda:0")
torch.cuda.synchronize()
W, b = init_weights(1701, D, F)
eng = vv.Engine(0, "f16"); eng.table_synth(ds.seed, ds.n_rows, F); eng.params_set(W, b)
cfg = vv.StepConfig(B, Cc, Nn)
stride = B * (Cc + Nn) * 4
for i in range(n):
    eng.forward_backward(cfg, idx_dev_ptr=idx.data_ptr() + i * stride, idx_ready=True)
    eng.apply_update(cfg)
eng.synchronize()
out = np.zeros((B, 16), np.uint32)
rc = eng.L.vv_lab_score_ts(eng.h, out.ctypes.data_as(C.c_void_p), B)
assert rc == 0, rc
ts = out[:, :6].astype(np.int64)
d = np.diff(ts, axis=1) & 0xffffffff
rt0, rt1 = out[:, 6].astype(np.int64), out[:, 7].astype(np.int64)
t0 = rt0.min()
start_us, end_us = (rt0 - t0) / 100.0, (rt1 - t0) / 100.0
names = ["rows of the context + 1st sum", "all rows reduced", "scores + block sums", "dAh + records", "dA + tail"]
print("k_score_fwd, %d workgroups: kernel span %.2f us (first start -> last end, 100 MHz clock)" % (B, end_us.max()))
print("  per workgroup (shader clocks, mean / p10 / p90):")
for i, nm in enumerate(names):
    print("    %-32s %7.0f %7.0f %7.0f" % (nm, d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
tot = (ts[:, 5] - ts[:, 0]) & 0xffffffff
dur_us = end_us - start_us
print("    %-32s %7.0f clocks = %.2f us mean (p10 %.2f, p90 %.2f); clock %.0f MHz" % ("whole workgroup", tot.mean(), dur_us.mean(), np.percentile(dur_us, 10), np.percentile(dur_us, 90), tot.mean() / dur_us.mean()))
order = np.argsort(start_us)
print("  start times (us) of the workgroups, sorted, every 64th:", " ".join("%.1f" % start_us[order[i]] for i in range(0, B, 64)), "last %.1f" % start_us[order[-1]])
print("  end times, every 64th:", " ".join("%.1f" % np.sort(end_us)[i] for i in range(0, B, 64)), "last %.1f" % end_us.max())
hw = out[:, 9]
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (out[:, 10].astype(np.int64) << 8)      # cu_id, sh_id, se_id, xcc
u, cnt = np.unique(cu, return_counts=True)
print("  distinct (xcc, se, sh, cu) seen: %d; workgroups per CU: min %d max %d" % (len(u), cnt.min(), cnt.max()))
conc = [(np.sum((start_us <= t) & (end_us > t))) for t in np.linspace(0, end_us.max(), 21)[:-1]]
print("  workgroups in flight at 5 %% steps of the span:", conc)
